#!/bin/bash
# Collect the round's profile evidence on the GPU box into gpurun_out/profiles_rNN/ (copy the summaries to profiles/).
# usage: tools/make_profiles.sh r01
tag=${1:-r03}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/profiles_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for wl in sparse clustered dense; do
  # (a) kernel trace + stats of the exact bench command
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$wl -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-columns --no-traffic --workload $wl > $out/bench_profiled_$wl.json 2> $out/kt_$wl.err
  # (b) HBM traffic: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${wl}_$c -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-columns --no-traffic --workload $wl > /dev/null 2> $out/pmc_${wl}_$c.err
  done
  # (c) the un-profiled bench line
  python3 $R/bench.py --steps 20 --warmup 3 --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
  echo "done $wl"
done
# (d) SQ counters of the three kernels on the headline workload (separate passes: 8 SQ slots per pass), same build, same run
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/sq_p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-columns --no-traffic --workload sparse > /dev/null 2> $out/sq_p$i.err || echo "SQ pass $i failed"
done
# (e) the kernels of SURVEY 8(f) on the sparse and the clustered GiB (tools/next_rows_time.py): kernel trace per workload
for wl in sparse clustered; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_next_$wl -- python3 $R/tools/next_rows_time.py $wl > $out/next_rows_$wl.log 2> $out/kt_next_$wl.err || echo "next rows $wl failed"
done
python3 - <<PY
import csv, glob, json, re, collections, os
out = "$out"
# ---- 8(f) rows: kernel-trace average, algorithmic bytes, fraction of 8 TB/s
with open(f"{out}/next_rows.txt", "w") as o:
    o.write("rocprofv3 --kernel-trace --stats -- python3 tools/next_rows_time.py <workload>   (1 GiB bitmaps, N = 268435200 words; every call 1 + 5 times)\n")
    o.write("algorithmic bytes: unsegmented encoder 4N + 4C_u; checker / sums pass 4C; index builder 4C + 8 per segment; merge count 4C, scatter 4C + 4C_u;\n")
    o.write("fused AND of two indexed streams 4C_A + 4C_B + 4C_out; four-operand combining pass 4(C_A + .. + C_D) + 4N (one decoded bitmap written);\n")
    o.write("run merge (operands of few words per segment): merge pass 4(C_A + ..) + 4C_out, placement pass 8C_out\n")
    for wl in ("sparse", "clustered"):
        meta = None
        try:
            for ln in open(f"{out}/next_rows_{wl}.log"):
                if ln.startswith("NEXT_ROWS "):
                    meta = json.loads(ln[len("NEXT_ROWS "):])
        except Exception:
            pass
        if not meta:
            o.write(f"== {wl}: no NEXT_ROWS line\n")
            continue
        d = collections.defaultdict(list)
        for f in glob.glob(f"{out}/kt_next_{wl}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"(\w+_kernel)(<\d+)?", r["Kernel_Name"])
                if m:
                    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    d[m.group(1)].append(us)
                    if m.group(2):  # (template kernels by their first argument too: bitop_runs_kernel<2, <4 -- the operand count)
                        d[m.group(1) + m.group(2)].append(us)
        o.write(f"== {wl}: C = {meta['c_words']} words (C/N {meta['c_words']/meta['n_words']:.4f}), unsegmented {meta['c_unsegmented']} words\n")
        for k, b in meta["algorithmic_bytes"].items():
            v = d.get(k)
            if not v:
                o.write(f"  {k:30s} (not in the trace)\n")
                continue
            if k == "compress_pair_kernel":  # the set-up launches (making the operands) run the same kernel on other bytes
                v = v[-6:]
            avg = sum(v) / len(v)
            o.write(f"  {k:30s} launches {len(v):3d}  avg {avg:9.1f} us  algorithmic {b/1e9:7.4f} GB  {b/avg/1e3:7.0f} GB/s = {b/avg/1e3/8000.0:.3f} of 8 TB/s\n")
        o.write("  calls (ms between two events, everything the call launches): " + json.dumps(meta["call_ms"]) + "\n")
print(open(f"{out}/next_rows.txt").read())
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
    lines.append(f"== {wl}: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-columns --workload {wl}")
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
        "compress_bytes_per_launch": hbm("compress_pair_kernel"),
        # the general decoder: one pass (decode_tile_kernel + the launch over its list of deferred tiles) or, for highly
        # compressed streams, sums + expand -- whichever kernels ran
        "decompress_bytes_per_launch": hbm("decode_tile_kernel") + hbm("decode_expand_list_kernel") + hbm("decode_sums_kernel") + hbm("decode_expand_kernel"),
        "decompress_indexed_bytes_per_launch": hbm("decode_segments_kernel"),
        "source": f"profiles/{os.path.basename(out).replace('profiles_', '')}_summary.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md",
        "raw_KiB": tr,
    }
    try:
        lines.append("  bench line: " + open(f"{out}/bench_{wl}.json").read().strip())
    except Exception:
        pass
# SQ counters, mean per dispatch, the three kernels of the round trip (sparse)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/sq_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(compress_pair_kernel|decode_tile_kernel|decode_sums_kernel|decode_expand_kernel)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/pmc_sq_counters_sparse.txt", "w") as o:
    o.write("rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-columns --workload sparse\n")
    for k in agg:
        o.write(f"== {k}\n")
        for c, v in sorted(agg[k].items()):
            o.write(f"  {c:28s} mean/dispatch {sum(v)/len(v):16.1f}  (n={len(v)})\n")
    c = agg.get("compress_pair_kernel", {})
    if c.get("SQ_INSTS_VALU") and c.get("GRBM_GUI_ACTIVE"):
        mean = lambda k: sum(c[k]) / len(c[k])
        segs = 270600.0
        cycles = mean("GRBM_GUI_ACTIVE") / 8.0  # the counter sums the 8 XCDs
        o.write(f"== compress_pair_kernel, derived (1 GiB = {int(segs)} segments, 1024 SIMDs)\n")
        o.write(f"  instructions per segment: vector {mean('SQ_INSTS_VALU')/segs:.0f}, scalar {mean('SQ_INSTS_SALU')/segs:.0f}, LDS {mean('SQ_INSTS_LDS')/segs:.0f}, "
                f"branch {mean('SQ_INSTS_BRANCH')/segs if c.get('SQ_INSTS_BRANCH') else 0:.0f}, vector memory {(mean('SQ_INSTS_VMEM_RD')+mean('SQ_INSTS_VMEM_WR'))/segs:.1f}\n")
        o.write(f"  kernel cycles (GRBM_GUI_ACTIVE / 8): {cycles:.0f}\n")
        per_simd = mean("SQ_INSTS_VALU") / 1024.0
        o.write(f"  vector instructions per SIMD: {per_simd:.0f}; VALU pipe busy at 2 cycles per wave64 instruction (SIMD-32): {2*per_simd/cycles:.2f}, "
                f"at the 4 cycles one wave alone needs per instruction: {4*per_simd/cycles:.2f}\n")
        if c.get("SQ_WAVE_CYCLES"):
            o.write(f"  of the wave cycles: issuing {mean('SQ_ACTIVE_INST_ANY')/mean('SQ_WAVE_CYCLES'):.2f}, issue-stalled {mean('SQ_WAIT_INST_ANY')/mean('SQ_WAVE_CYCLES'):.2f}, "
                    f"waiting (s_waitcnt / barrier) {mean('SQ_WAIT_ANY')/mean('SQ_WAVE_CYCLES'):.2f}\n")
        if c.get("SQ_LDS_IDX_ACTIVE"):
            o.write(f"  LDS bank conflict cycles / LDS active cycles: {mean('SQ_LDS_BANK_CONFLICT')/mean('SQ_LDS_IDX_ACTIVE'):.2f}\n")
    t = agg.get("decode_tile_kernel", {})
    if t.get("SQ_INSTS_VALU") and t.get("GRBM_GUI_ACTIVE"):
        mean = lambda k: sum(t[k]) / len(t[k])
        segs = 270600.0
        cycles = mean("GRBM_GUI_ACTIVE") / 8.0
        o.write(f"== decode_tile_kernel, derived (1 GiB = {int(segs)} output segments, 1024 SIMDs)\n")
        o.write(f"  instructions per output segment: vector {mean('SQ_INSTS_VALU')/segs:.0f}, scalar {mean('SQ_INSTS_SALU')/segs:.0f}, LDS {mean('SQ_INSTS_LDS')/segs:.0f}, "
                f"branch {mean('SQ_INSTS_BRANCH')/segs:.0f}, vector memory {(mean('SQ_INSTS_VMEM_RD')+mean('SQ_INSTS_VMEM_WR'))/segs:.1f}\n")
        o.write(f"  kernel cycles (GRBM_GUI_ACTIVE / 8): {cycles:.0f}\n")
        per_simd = mean("SQ_INSTS_VALU") / 1024.0
        o.write(f"  vector instructions per SIMD: {per_simd:.0f}; VALU pipe busy at the 4 cycles a wave64 instruction takes: {4*per_simd/cycles:.2f}\n")
        o.write(f"  of the wave cycles: issuing {mean('SQ_ACTIVE_INST_ANY')/mean('SQ_WAVE_CYCLES'):.2f}, issue-stalled {mean('SQ_WAIT_INST_ANY')/mean('SQ_WAVE_CYCLES'):.2f}, "
                f"waiting (s_waitcnt / barrier) {mean('SQ_WAIT_ANY')/mean('SQ_WAVE_CYCLES'):.2f}\n")
    for dk in ("decode_tile_kernel", "decode_expand_kernel"):
      e = agg.get(dk, {})
      if e.get("SQ_LDS_IDX_ACTIVE"):
        o.write(f"== {dk}: LDS bank conflict cycles / LDS active cycles: {sum(e['SQ_LDS_BANK_CONFLICT'])/len(e['SQ_LDS_BANK_CONFLICT'])/(sum(e['SQ_LDS_IDX_ACTIVE'])/len(e['SQ_LDS_IDX_ACTIVE'])):.2f}\n")
print(open(f"{out}/pmc_sq_counters_sparse.txt").read())
open(f"{out}/summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(summary, open(f"{out}/traffic.json", "w"), indent=1)
print("\n".join(lines))
PY
