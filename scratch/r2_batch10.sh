R=$GRAFT_REPO_ROOT; cd $R
for v in "" "A2P_NO_SIDE_STREAM=1" "A2P_NO_SHARED_HALF=1" "A2P_NO_SIDE_STREAM=1 A2P_NO_SHARED_HALF=1" "A2P_ATTN_NO_REMAP=1"; do
  echo "== $v"; env $v timeout 300 python -m pytest tests/test_hip_round2.py -m gpu -q -k "pose" 2>&1 | grep -E "passed|failed|max .diff"
done
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
