"""Developer tool: print the kernel sequence of one minibatch from a rocprofv3
kernel trace csv (durations in us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
i0 = len(rows) // 2
while 'adam' not in rows[i0]['Kernel_Name']:
    i0 += 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for r in rows[i0 + 1:i0 + 1 + n]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:64]
    print('%-66s grid=%s,%s,%s dur=%7.1f' % (nm, r['Grid_Size_X'], r['Grid_Size_Y'],
                                            r['Grid_Size_Z'], (e - s) / 1e3))
