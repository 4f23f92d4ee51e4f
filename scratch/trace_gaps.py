"""Per-step timeline from a rocprofv3 kernel trace: busy time vs wall, largest gaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find step boundaries: map_timesteps_kernel starts a step
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('map_timesteps_kernel')]
for a, b in list(zip(idx, idx[1:]))[-3:]:
    seg = rows[a:b]
    t0, t1 = int(seg[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    print(f"step: {len(seg)} kernels wall={(t1 - t0) / 1e3:.1f}us busy={busy / 1e3:.1f}us")
    gaps = []
    for p, q in zip(seg, seg[1:] + [rows[b]]):
        gaps.append(((int(q['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3, p['Kernel_Name'][:40], q['Kernel_Name'][:40]))
    gaps.sort(reverse=True)
    for g in gaps[:8]:
        print(f"   gap {g[0]:7.1f}us after {g[1]} before {g[2]}")
    print("   total gap %.1fus, median gap %.2fus" % (sum(g[0] for g in gaps), sorted(g[0] for g in gaps)[len(gaps) // 2]))
