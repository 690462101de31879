# same-box A/B of two builds through bench.py (compress in the round trip): usage: tools/ab_bench.sh <other .so>
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for lib in "" "$1"; do
    for wl in sparse clustered dense; do
      WAH_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-columns --no-traffic --workload $wl 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('${lib:-shipped}', '$wl', j['value'], j['compress_ms']['avg'], j['decompress_ms']['avg'], r['frac'], r['frac_isolated'])"
    done
  done
done
