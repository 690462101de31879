# round-trip A/B through bench.py: shipped two-launch decoder against the one-pass decoder #7 (tools/experiments), 1 / 2 tiles per workgroup
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for cfg in "shipped:" "onepass1:tools/libwah_onepass.so:1" "onepass2:tools/libwah_onepass.so:2"; do
    IFS=: read name lib batch <<< "$cfg"
    for wl in sparse dense; do
      WAH_DT_BATCH=${batch:-2} WAH_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-columns --workload $wl 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '$wl', j['value'], j['compress_ms']['avg'], j['decompress_ms']['avg'], j['ms_per_step'])"
    done
  done
done
