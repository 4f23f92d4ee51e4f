R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_round2.py -m gpu -q -x -k T600 2>&1 | tail -4
python - <<'PY'
import json; d=json.load(open("gpurun_out/parity_tests.json")); print({k:v for k,v in d.items() if k.startswith("T600")})
PY
timeout 600 python bench.py --no-legs --write-parity $O/r02_parity.json > $O/b4_bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/b4_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['parity'], indent=1)); print(d['value'], d['cpu_baseline'])"
timeout 300 python bench.py --no-cpu-baseline --no-legs --precision fp16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16', d['value'], d['ms_per_step'], {k:(v['avg_launch_us'], v.get('tflops')) for k,v in d['kernels'].items()})"
