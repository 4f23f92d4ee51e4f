"""Per-launch HBM traffic of the hot kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units).
gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md "HBM"); WRITE_SIZE is uncalibrated (raw)."""
import csv, sys, json, collections
def mean_by_kernel(path):
    agg, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(path)):
        agg[row["Kernel_Name"]] += float(row["Counter_Value"]); cnt[row["Kernel_Name"]] += 1
    return {k: agg[k] / cnt[k] for k in agg}, cnt
fetch, n = mean_by_kernel(sys.argv[1]); write, _ = mean_by_kernel(sys.argv[2])
classes = {"attn3": ("attn3_kernel",), "chain": ("chain_kernel", "chain4_kernel"), "chain_tall": ("chain4_kernel",), "chain_gen1": ("chain_kernel",), "attn": ("attn_kernel",), "gemm": ("gemm_kernel",), "ln_rope": ("ln_rope_kernel",)}
out, lines = {}, []
for cls, pat in classes.items():
    ks = [k for k in fetch if any(q in k for q in pat)]
    if not ks: continue
    tot = sum(n[k] for k in ks)
    f = sum(fetch[k] * n[k] for k in ks) / tot * 1024 * 2
    w = sum(write.get(k, 0) * n[k] for k in ks) / tot * 1024
    out[cls] = {"fetch_bytes": round(f), "write_bytes": round(w), "total_bytes": round(f + w), "launches_profiled": tot}
for k in sorted(fetch, key=lambda k: -fetch[k] * n[k])[:12]:
    lines.append(f"{k[:60]:60s} n={n[k]:4d} FETCH_SIZE(KiB, raw)={fetch[k]:10.0f} WRITE_SIZE(KiB, raw)={write.get(k, 0):10.0f}")
out["attn_cross"] = out.get("attn3") or out.get("attn"); out["attn_self"] = out.get("attn3") or out.get("attn")
# the chain class of ONE step per kernel family (bench.py roofline.chain_family = 10 x MID + POST; 1 = kernels_chain.h, 4 = kernels_chain4.h):
# 17 launches = the input / PRE kernel (chain4_kernel<3, 4, 0>: input projection + layer 0's PRE work) + 8 MID + 7 POST + the last layer's POST
# (always the tall kernel with final_layer inside: chain4_kernel<3, 2, 2>).  face model, B = 8 (48-row panels) names.
def one(name_part):
    ks = [k for k in fetch if name_part in k]
    if not ks: return None
    k = ks[0]
    return fetch[k] * 1024 * 2, write.get(k, 0) * 1024
names = {"pre": "chain4_kernel<3, 4, 0>", "mid1": "chain_kernel<512, 3, 1, 0, 8>", "post1": "chain_kernel<512, 3, 2, 0, 8>",
         "mid4": "chain4_kernel<3, 1, 0>", "post4": "chain4_kernel<3, 2, 0>", "last4": "chain4_kernel<3, 2, 2>"}
v = {k: one(n) for k, n in names.items()}
fam = {}
for mid in (1, 4):
    for post in (1, 4):
        parts = [(v["pre"], 1), (v[f"mid{mid}"], 8), (v[f"post{post}"], 7), (v["last4"], 1)]
        if any(x is None for x, _ in parts): continue
        f = sum(x[0] * n for x, n in parts) / 17; w = sum(x[1] * n for x, n in parts) / 17
        fam[str(10 * mid + post)] = {"fetch_bytes": round(f), "write_bytes": round(w), "total_bytes": round(f + w), "what": "per chain launch, 17 launches of one step"}
out["chain_by_family"] = fam
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("\n".join(lines)); print(json.dumps(out))
