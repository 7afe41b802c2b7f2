"""Every kernel of one steady-state bench step, in launch order (from a rocprofv3 kernel trace)."""
import csv, glob, re, sys
path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'synth_mfma_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n = r['Kernel_Name'].strip('"')
    n = re.sub(r'\(.*', '', n)[:90]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  grid {r['Grid_Size_X']:>10s}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']:>4s}  {n}")
