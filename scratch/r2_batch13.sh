R=$GRAFT_REPO_ROOT; cd $R
timeout 200 scratch/chain_bench 2>&1 | grep -E "M=|full|phases|no " 
