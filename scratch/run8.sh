R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "chain|passed|failed|Error|error" | tail -20
