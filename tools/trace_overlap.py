"""Developer tool: start / end of the kernels around the middle of a rocprofv3 kernel
trace (csv), relative to the first one shown, with the stream each ran on -- how the
two update chains of the overlapped mode actually interleave."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
i0 = len(rows) // 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0 + n]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:40]
    q = r.get('Queue_Id', r.get('Stream_Id', '?'))
    print('%-42s q=%-4s start=%8.1f end=%8.1f dur=%6.1f' % (nm, q, (s - t0) / 1e3,
                                                            (e - t0) / 1e3, (e - s) / 1e3))
