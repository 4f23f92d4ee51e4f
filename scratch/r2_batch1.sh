# Round-2 GPU batch 1 (run under gpurun): test suite, open-item bisect, issue-cost probe, clocks, full bench, PMC pass.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_round2.py > $O/b1_tests_r1.log 2>&1; tail -5 $O/b1_tests_r1.log
timeout 600 python -m pytest tests/test_hip_round2.py -m gpu -q > $O/b1_tests_r2.log 2>&1; tail -30 $O/b1_tests_r2.log
echo "== bisect"; timeout 300 python scratch/pose_nw_bisect.py > $O/b1_bisect.log 2>&1; tail -40 $O/b1_bisect.log
echo "== issue probe"; timeout 120 scratch/issue_probe > $O/b1_issue_probe.txt 2>&1; cat $O/b1_issue_probe.txt
echo "== clocks"; timeout 300 python scratch/clk_probe.py > $O/b1_clk.txt 2>&1; cat $O/b1_clk.txt
echo "== bench"; timeout 600 python bench.py --write-parity $O/r02_parity.json > $O/b1_bench.json 2> $O/b1_bench.err; tail -c 6000 $O/b1_bench.json; tail -5 $O/b1_bench.err
echo "== counters"; rocprofv3 -L > $O/b1_counters.txt 2>&1; grep -c . $O/b1_counters.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-timing --no-legs"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_a -o p -- $B > $O/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b -o p -- $B > $O/pmc_b.log 2>&1
cd $R
for t in a b; do f=$(ls $O/pmc_$t/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python scratch/pmc_summary.py $f | grep -E "chain_kernel|attn_kernel" | cut -c1-400 > $O/b1_pmc_$t.txt; rm -rf $O/pmc_$t; done
cat $O/b1_pmc_a.txt $O/b1_pmc_b.txt 2>/dev/null; tail -3 $O/pmc_a.log $O/pmc_b.log
