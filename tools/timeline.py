"""Prints the kernel timeline of the last proof in a rocprofv3 --kernel-trace CSV (one line per kernel > min_ms)."""
import csv, sys
path = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('zk::k_assemble(')]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['End_Timestamp'])
print("proof span ms: %.3f" % ((int(rows[b]['End_Timestamp']) - t0) / 1e6))
for r in rows[a + 1:b + 1]:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zk::', '').replace('Fp<FqParams> ', 'Fq')
    s = (int(r['Start_Timestamp']) - t0) / 1e6
    e = (int(r['End_Timestamp']) - t0) / 1e6
    if e - s > min_ms or 'assemble' in n:
        print("%7.3f -> %7.3f  (%6.3f)  %s" % (s, e, e - s, n[:60]))
