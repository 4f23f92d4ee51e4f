#!/bin/bash
# round 5, call 30: evidence of the final kernels (scratch/run_evidence_r05.sh: rocprofv3 stats, PMC traffic B=8, SQ counters, power, default bench) + PMC traffic at B=32 + full GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
( time timeout -k 5 1200 python -m pytest tests -m gpu -q -x ) > $O/r05_c30_tests.log 2>&1; tail -5 $O/r05_c30_tests.log
bash scratch/run_evidence_r05.sh > $O/r05_c30_evidence.log 2>&1; tail -c 1500 $O/r05_c30_evidence.log
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-legs --no-parity --repeats 1 --batch 32"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_b32 -o p -- $B > $O/pmcf_b32.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_b32 -o p -- $B > $O/pmcw_b32.log 2>&1
cd $R
python scratch/pmc_traffic.py $O/pmcf_b32/p_counter_collection.csv $O/pmcw_b32/p_counter_collection.csv $O/r05_pmc_traffic_b32.json > $O/r05_pmc_traffic_b32.txt 2>&1
rm -rf $O/pmcf_b32 $O/pmcw_b32
head -8 $O/r05_pmc_traffic_b32.txt | cut -c1-160
