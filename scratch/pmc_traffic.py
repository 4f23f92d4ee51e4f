"""Per-launch HBM traffic of the hot kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units).
gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md "HBM"); WRITE_SIZE is uncalibrated (raw)."""
import csv, sys, json, collections
def mean_by_kernel(path):
    agg, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(path)):
        agg[row["Kernel_Name"]] += float(row["Counter_Value"]); cnt[row["Kernel_Name"]] += 1
    return {k: agg[k] / cnt[k] for k in agg}, cnt
fetch, n = mean_by_kernel(sys.argv[1]); write, _ = mean_by_kernel(sys.argv[2])
classes = {"chain": ("chain_kernel", "chain4_kernel"), "chain_tall": ("chain4_kernel",), "chain_gen1": ("chain_kernel",), "attn": ("attn_kernel",), "gemm": ("gemm_kernel",), "ln_rope": ("ln_rope_kernel",)}
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
out["attn_cross"] = out.get("attn"); out["attn_self"] = out.get("attn")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("\n".join(lines)); print(json.dumps(out))
