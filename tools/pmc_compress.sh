#!/bin/bash
# PMC passes for the compress kernel (separate runs, --pmc only with kernel-trace as the guide prescribes).
# usage: tools/pmc_compress.sh <outdir> [workload]
out=${1:-gpurun_out/pmc}; wl=${2:-sparse}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/$out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_I8" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  [ -n "$PMC_PASSES" ] && [ $i -gt $PMC_PASSES ] && break
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$out/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $wl $BENCH_EXTRA > $R/$out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, os
R="$R/$out"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row.get("Kernel_Name","")
        short = "compress" if ("compress_tile_kernel" in k or "compress_pair_kernel" in k) else ("sums" if "decode_sums" in k else ("expand" if "decode_expand" in k else None))
        if short: agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(R+"/summary.txt","w") as o:
    for k in agg:
        o.write(f"== {k}\n")
        for c,v in sorted(agg[k].items()):
            o.write(f"  {c:28s} mean/dispatch {sum(v)/len(v):16.1f}  (n={len(v)})\n")
print(open(R+"/summary.txt").read())
PY
