#!/bin/bash
# kernel-trace + stats of bench.py; usage: tools/ktrace.sh <outdir> [bench args...]
out=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf $R/$out; mkdir -p $R/$out
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/$out/bench.log 2>&1
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("$R/$out/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0][-60:]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
with open("$R/$out/kernel_summary.txt","w") as o:
    for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
        line=f"{k:62s} n={len(v):4d} total {sum(v):10.1f} us  mean {sum(v)/len(v):9.1f}  min {min(v):9.1f}  max {max(v):9.1f}"
        print(line); o.write(line+"\n")
PY
