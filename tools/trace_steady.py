"""Per-kernel statistics of the STEADY-STATE part of a rocprofv3 kernel trace.
python tools/trace_steady.py <..._kernel_trace.csv> [marker=xcc_census_kernel] > stats.csv
python tools/trace_steady.py <..._kernel_trace.csv> --longest <kernel name part> <count>   (the longest dispatches, ms)

tools/gamg_profile.py creates a second library context after its warm-up solve; the context's placement census
(`xcc_census_kernel`) is the only launch of that kernel outside set-up, so everything after its LAST occurrence is
steady state (no agglomeration, no plan uploads).  Output: the columns of rocprofv3's own kernel_stats.csv."""
import csv, sys, collections, math
path = sys.argv[1]
longest = None
hist = None
if len(sys.argv) > 2 and sys.argv[2] == "--hist":
    # dispatch durations of one kernel name after the marker, grouped (a name that serves several GAMG levels)
    hist = sys.argv[3]
    marker = "xcc_census_kernel"
elif len(sys.argv) > 2 and sys.argv[2] == "--longest":
    longest = (sys.argv[3], int(sys.argv[4]))
    marker = "xcc_census_kernel"
else:
    marker = sys.argv[2] if len(sys.argv) > 2 else "xcc_census_kernel"
rows = list(csv.DictReader(open(path)))
kn = "Kernel_Name"; ts = "Start_Timestamp"; te = "End_Timestamp"
t0 = max((int(r[ts]) for r in rows if marker in r[kn]), default=None)
if t0 is None:
    sys.exit("marker kernel %s not in the trace" % marker)
agg = collections.defaultdict(list)
for r in rows:
    if int(r[ts]) > t0 and marker not in r[kn]:
        agg[r[kn]].append(int(r[te]) - int(r[ts]))
if hist:
    d = sorted(x * 1e-6 for k, v in agg.items() if hist in k for x in v)
    print("%s: %d dispatches after the marker; durations [ms] in groups of equal launches (count x mean):" % (hist, len(d)))
    groups, cur = [], []
    for x in d:
        if cur and x > 1.12 * cur[0] + 0.01:
            groups.append(cur); cur = []
        cur.append(x)
    if cur: groups.append(cur)
    print("  " + "  ".join("%d x %.4f" % (len(g), sum(g) / len(g)) for g in groups))
    sys.exit(0)
if longest:
    d = sorted((x for k, v in agg.items() if longest[0] in k for x in v), reverse=True)[:longest[1]]
    print("%s: the %d longest dispatches after the marker [ms]" % (longest[0], len(d)))
    print([round(x * 1e-6, 6) for x in d])
    print("mean %.4f ms" % (sum(d) / max(1, len(d)) * 1e-6))
    sys.exit(0)
tot = sum(sum(v) for v in agg.values())
w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    m = sum(v) / len(v)
    sd = math.sqrt(sum((x - m) ** 2 for x in v) / len(v))
    w.writerow([k, len(v), sum(v), round(m, 1), round(100.0 * sum(v) / tot, 3), min(v), max(v), round(sd, 1)])
