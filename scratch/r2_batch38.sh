R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-legs --repeats 1"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_a -o p -- $B > $O/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_b -o p -- $B > $O/pmc_b.log 2>&1
cd $R
for t in a b; do f=$(ls $O/pmc_$t/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python scratch/pmc_summary.py $f | grep -E "chain_kernel|attn_kernel" | cut -c1-420 > $O/final_pmc_$t.txt; rm -rf $O/pmc_$t; done
cat $O/final_pmc_a.txt $O/final_pmc_b.txt 2>/dev/null; tail -2 $O/pmc_a.log $O/pmc_b.log
