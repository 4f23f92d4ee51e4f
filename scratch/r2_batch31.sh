R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python bench.py --pipeline 2>&1 | tail -2 | cut -c1-1100
timeout 600 python bench.py --pipeline --batch 32 2>&1 | tail -1 | cut -c1-1100
