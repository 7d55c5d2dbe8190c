import csv, sys, glob
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'argmax' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
span = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3
print(len(step), 'kernels in last step; span us', span, 'busy us', busy)
att = [i for i, r in enumerate(step) if 'paged_attention_kernel' in r['Kernel_Name']]
l0 = step[att[3] + 1:att[4] + 1]
prev = None
for r in l0:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0
    prev = e
    print(f"{r['Kernel_Name'][:78]:78s} grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])},{r['Grid_Size_Y']},{r['Grid_Size_Z']} dur={(e-s)/1e3:7.2f} gap={gap:6.2f} vgpr={r.get('VGPR_Count')} lds={r.get('LDS_Block_Size')}")
agg = {}
for r in step:
    k = r['Kernel_Name'][:60]
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} n={v[0]:4d} total_us={v[1]:9.1f} avg={v[1]/v[0]:7.2f}")
