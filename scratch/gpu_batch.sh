#!/bin/bash
# round 6, call 27: library = attn3_kernel v5 (MFMA row sums) + family calibration v2 (MID tall by rule, POST fastest of four): full GPU suite, then the default bench line (timed)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
T0=$(date +%s); timeout -k 5 900 env A2P_TUNE_VERBOSE=1 python bench.py > $O/r06_c26_bench_default.json 2> $O/r06_c26_bench_default.err
echo "default bench wall: $(( $(date +%s) - T0 )) s"; grep "a2p\]" $O/r06_c26_bench_default.err | head
python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c26_bench_default.json") if l.startswith("{")][-1])
k=j["kernels"]; sub=k.get("_sub_classes",{})
print("headline", j["value"], {a:k[a]["avg_launch_us"] for a in ("chain","attn_self","attn_cross")}, {a:v["avg_launch_us"] for a,v in sub.items()}, "family", j["roofline"].get("chain_family"), "frac", j["roofline"]["frac"], "decoder", j.get("decoder_mfma_frac"))
for n,l in j["legs"].items():
    print(n, l["value"], l.get("decoder_mfma_frac"), l.get("chain_family"), {a:v["avg_launch_us"] for a,v in l["kernels"].items() if isinstance(v,dict) and "avg_launch_us" in v})
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["all_host_cpus_as_threads"])
print("parity bar", j["parity"]["bar"]); print("b8", j["parity"]["b8"]["fp16"]["last_step_worst_single_sample_rel_l2"], "chain_vs_oracle fp16", j["parity"].get("chain_vs_oracle",{}).get("fp16"))
PY
