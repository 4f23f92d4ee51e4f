R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -o p -- $B > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -o p -- $B > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc3 -o p -- $B > $R/gpurun_out/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc4 -o p -- $B > $R/gpurun_out/pmc4.log 2>&1
cd $R
for i in 1 2 3 4; do f=$(ls gpurun_out/pmc$i/*counter_collection.csv 2>/dev/null | head -1); echo "== pmc$i $f"; [ -n "$f" ] && python scratch/pmc_summary.py $f > gpurun_out/pmc${i}_summary.txt; head -12 gpurun_out/pmc${i}_summary.txt; rm -f gpurun_out/pmc$i/*counter_collection.csv gpurun_out/pmc$i/*kernel_trace.csv; done
