#!/bin/bash
# Collect the round's profile evidence on the GPU box into gpurun_out/profiles_rNN/ (copy the summaries to profiles/).
# usage: tools/make_profiles.sh r01
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/profiles_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for wl in sparse clustered dense; do
  # (a) kernel trace + stats of the exact bench command
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$wl -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $wl > $out/bench_profiled_$wl.json 2> $out/kt_$wl.err
  # (b) HBM traffic: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${wl}_$c -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $wl > /dev/null 2> $out/pmc_${wl}_$c.err
  done
  # (c) the un-profiled bench line
  python3 $R/bench.py --steps 20 --warmup 3 --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
  echo "done $wl"
done
python3 - <<PY
import csv, glob, json, re, collections, os
out = "$out"
summary = {}
lines = []
for wl in ("sparse", "clustered", "dense"):
    # kernel-trace: per-kernel durations
    d = collections.defaultdict(list)
    for f in glob.glob(f"{out}/kt_{wl}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            k = m.group(1) if m else r["Kernel_Name"][:48]
            d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines.append(f"== {wl}: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload {wl}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"  {k:36s} calls {len(v):4d}  avg {sum(v)/len(v):10.1f} us  min {min(v):10.1f}  max {max(v):10.1f}  total {sum(v):12.1f}")
    # PMC: bytes per launch
    tr = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for f in glob.glob(f"{out}/pmc_{wl}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
                if m and r["Counter_Name"] == c:
                    agg[m.group(1)].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            tr.setdefault(k, {})[c] = sum(v) / len(v)
    lines.append(f"  PMC (KiB per launch, raw counters): " + json.dumps(tr))
    def hbm(k):
        # FETCH_SIZE counts 64 B per 128-B request on gfx950 for wide coalesced streams: double it (MI355X_MICROARCH.md, HBM)
        t = tr.get(k, {})
        return (2.0 * t.get("FETCH_SIZE", 0) + t.get("WRITE_SIZE", 0)) * 1024.0
    summary[wl] = {
        "compress_bytes_per_launch": hbm("compress_tile_kernel"),
        "decompress_bytes_per_launch": hbm("decode_sums_kernel") + hbm("decode_expand_kernel"),
        "decompress_indexed_bytes_per_launch": hbm("decode_segments_kernel"),
        "source": f"profiles/{os.path.basename(out).replace('profiles_', '')}_summary.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md",
        "raw_KiB": tr,
    }
    try:
        lines.append("  bench line: " + open(f"{out}/bench_{wl}.json").read().strip())
    except Exception:
        pass
open(f"{out}/summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(summary, open(f"{out}/traffic.json", "w"), indent=1)
print("\n".join(lines))
PY
