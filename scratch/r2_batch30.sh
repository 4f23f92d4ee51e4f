R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_frontend_hip.py -m gpu -q -x 2>&1 | tail -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_tests.json'))
for k,v in d.items():
    if 'frontend' in k: print(k, v)
PY
timeout 600 python bench.py --pipeline 2>/dev/null | tail -1 | cut -c1-900
timeout 600 python bench.py --pipeline --batch 32 2>/dev/null | tail -1 | cut -c1-900
