R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^FAILED|passed|failed|AssertionError|max .diff" | head -6; done
