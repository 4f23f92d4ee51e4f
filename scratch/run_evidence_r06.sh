#!/bin/bash
# Round-6 evidence for profiles/ (run under gpurun): rocprofv3 kernel stats of the default bench command (B=8), of B=32 and of the body
# leg; HBM traffic counters (separate --pmc passes, FETCH doubled per the gfx950 note in pmc_traffic.py); one SQ counter pass (matrix
# pipe busy); board power / clocks sampled with rocm-smi while the B=8 and B=32 steps run; then the un-profiled default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=r06
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-legs --no-parity --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o p -- $B > $O/prof_$TAG.log 2>&1
cp $O/prof_$TAG/p_kernel_stats.csv $O/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_b32 -o p -- $B --batch 32 > $O/prof_${TAG}_b32.log 2>&1
cp $O/prof_${TAG}_b32/p_kernel_stats.csv $O/${TAG}_b32_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_body -o p -- $B --model pose --batch 16 > $O/prof_${TAG}_body.log 2>&1
cp $O/prof_${TAG}_body/p_kernel_stats.csv $O/${TAG}_body_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$TAG -o p -- $B > $O/pmcf_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$TAG -o p -- $B > $O/pmcw_$TAG.log 2>&1
cd $R
python scratch/pmc_traffic.py $O/pmcf_$TAG/p_counter_collection.csv $O/pmcw_$TAG/p_counter_collection.csv $O/pmc_traffic_$TAG.json > $O/pmc_traffic_$TAG.txt 2>&1
cd /tmp
: > $O/${TAG}_pmc_sq_counters.txt
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf $O/pmcs_$TAG
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmcs_$TAG -o p -- $B > $O/pmcs_$TAG.log 2>&1
  echo "## --pmc $SET   (mean per dispatch; SQ_* activity counters are in quad-cycles summed over SIMDs / XCDs as the guide describes)" >> $O/${TAG}_pmc_sq_counters.txt
  if [ -f $O/pmcs_$TAG/p_counter_collection.csv ]; then python $R/scratch/pmc_summary.py $O/pmcs_$TAG/p_counter_collection.csv | grep -E 'attn3_kernel|attn_kernel|chain_kernel|chain4_kernel|gemm_kernel|step_tail|split3' >> $O/${TAG}_pmc_sq_counters.txt
  else echo "(pass failed: $(tail -2 $O/pmcs_$TAG.log | tr '\n' ' '))" >> $O/${TAG}_pmc_sq_counters.txt; fi
done
rm -rf $O/pmcs_$TAG $O/pmcf_$TAG $O/pmcw_$TAG $O/prof_$TAG $O/prof_${TAG}_b32 $O/prof_${TAG}_body
cd $R
head -8 $O/${TAG}_kernel_stats.csv | cut -c1-140; tail -3 $O/pmc_traffic_$TAG.txt; cat $O/${TAG}_pmc_sq_counters.txt | cut -c1-260
# un-profiled bench line
timeout -k 5 900 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -c 600 $O/${TAG}_bench_default.json
