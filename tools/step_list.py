"""One steady-state training step of a rocprofv3 kernel trace, launch by launch in launch order:
   python tools/step_list.py trace.csv [--agg]    (--agg: per kernel name + grid, summed)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3
print('kernels', len(step), 'kernel us %.1f' % tot, 'span us %.1f' % ((int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3))
agg = {}
for r in step:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)[:64]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    if '--agg' in sys.argv:
        k = (n, g)
        agg[k] = (agg.get(k, (0, 0))[0] + d, agg.get(k, (0, 0))[1] + 1)
    else:
        print(f"{d:8.1f} {g:7d}x{r['Workgroup_Size_X']:>4} v{r['VGPR_Count']:>3} lds{r['LDS_Block_Size']:>6} {n}")
for (n, g), (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{d:9.1f} us {c:4d} x {d / c:8.1f}  grid {g:7d}  {n}")
