"""How much of the GAMG solve is GPU idle time between dependent kernels (launch-bound) rather than kernel time?
Reads a rocprofv3 --kernel-trace csv: gaps between consecutive kernels shorter than 100 us are launch gaps."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
busy = sum(e - s for s, e, _ in rows)
gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2][:40], rows[i + 1][2][:40]) for i in range(len(rows) - 1)]
small = [g for g in gaps if 0 < g[0] < 100000]
print("kernels", len(rows), "busy %.1f ms" % (busy / 1e6), "small gaps %.1f ms (n=%d, mean %.1f us)" % (
    sum(g[0] for g in small) / 1e6, len(small), sum(g[0] for g in small) / max(1, len(small)) / 1e3))
import collections
by = collections.Counter()
for g in small:
    by[g[2]] += g[0]
for k, v in by.most_common(12):
    print("  gap before %-42s %.2f ms" % (k, v / 1e6))
