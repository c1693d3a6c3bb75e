import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
sel = rows[n - 40:n - 20]
prev = None
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"gap {((s - prev) / 1e3 if prev else 0):7.2f}  dur {(e - s) / 1e3:8.2f}us  grid={r['Grid_Size_X']:>8} wg={r['Workgroup_Size_X']:>4} {r['Kernel_Name'][:40]}")
    prev = e
