R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 900 python bench.py --write-parity $O/r02_parity.json > $O/b5_bench.json 2> $O/b5_bench.err; tail -2 $O/b5_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/b5_bench.json").read().strip().splitlines()[-1])
print(d["dtype"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["decoder_mfma_frac"])
for k,v in (d.get("legs") or {}).items(): print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["decoder_mfma_frac"])
print(json.dumps(d["parity"]["chain"]))
PY
timeout 300 python bench.py --pipeline 2>/dev/null | tail -1
timeout 300 python bench.py --pipeline --batch 32 2>/dev/null | tail -1
timeout 300 python bench.py --batch 1 --frames 240 --no-legs --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config0 shape', d['value'], d['ms_per_step'], {k:(v['launches_per_step'], v['avg_launch_us']) for k,v in d['kernels'].items()})"
