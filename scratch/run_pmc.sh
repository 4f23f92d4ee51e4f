# rocprofv3 evidence for profiles/: kernel stats + HBM traffic counters of the default bench command (run under gpurun)
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-legs --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- $B > $R/gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_$TAG -o p -- $B > $R/gpurun_out/pmcf_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_$TAG -o p -- $B > $R/gpurun_out/pmcw_$TAG.log 2>&1
cd $R
python scratch/pmc_traffic.py gpurun_out/pmcf_$TAG/p_counter_collection.csv gpurun_out/pmcw_$TAG/p_counter_collection.csv gpurun_out/pmc_traffic_$TAG.json > gpurun_out/pmc_traffic_$TAG.txt
rm -f gpurun_out/pmc?_$TAG/*counter_collection.csv gpurun_out/pmc?_$TAG/*kernel_trace.csv gpurun_out/prof_$TAG/*kernel_trace.csv
cp gpurun_out/prof_$TAG/p_kernel_stats.csv gpurun_out/kernel_stats_$TAG.csv
head -8 gpurun_out/kernel_stats_$TAG.csv | cut -c1-120; cat gpurun_out/pmc_traffic_$TAG.txt | tail -3
