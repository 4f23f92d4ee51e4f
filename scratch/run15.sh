R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['kernels']['chain']['avg_launch_us'], d['kernels']['attn_cross']['avg_launch_us'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcg -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmcg.log 2>&1
cd $R
python - <<'PY'
import csv, collections
cc = list(csv.DictReader(open('gpurun_out/pmcg/p_counter_collection.csv')))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in cc:
    k = r['Kernel_Name'][:44]
    dur = int(r['End_Timestamp']) - int(r['Start_Timestamp']) if 'End_Timestamp' in r else 0
    agg[k][0] += float(r['Counter_Value']); agg[k][1] += dur; agg[k][2] += 1
for k, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    if ns > 0: print(f"{k:44s} n={n:4d} avg {ns/n/1e3:7.1f} us  GUI_ACTIVE/launch {cyc/n:10.0f}  => {cyc/ns:.3f} GHz")
print(list(cc[0].keys()))
PY
rm -rf gpurun_out/pmcg
