"""per-kernel launch times out of a rocprofv3 --kernel-trace output directory: python tools/ktrace_summary.py <dir>"""
import collections, csv, glob, re, sys
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if m:
            d[m.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:34s} launches {len(v):4d}  avg {sum(v) / len(v):9.1f} us  min {min(v):9.1f}  max {max(v):9.1f}")
