R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pipe -o p -- python $R/bench.py --pipeline > $R/gpurun_out/prof_pipe.log 2>&1
cd $R; head -30 gpurun_out/prof_pipe/p_kernel_stats.csv | cut -c1-160
rm -f gpurun_out/prof_pipe/*kernel_trace.csv
