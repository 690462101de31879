# the look-ahead decoder's lag (batches of 16 384 words counted ahead; 0 = the form that counts what it expands): isolated decode
# loops (tools/decode_ab.py) and the bench's round trip, same box.  usage: tools/lag_sweep.sh "<lags>" "<kinds>"
cd $GRAFT_REPO_ROOT
lags=${1:-"0 128 256 384 512 768"}; kinds=${2:-"sparse dense"}
for i in 1 2; do
  for lag in $lags; do
    WAH_DT_LAG=$lag timeout -k 10 200 python tools/decode_ab.py $kinds || exit 1
  done
done
for lag in $lags; do
  for w in $kinds; do
    echo "== bench lag $lag $w"
    WAH_DT_LAG=$lag timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --no-columns --no-traffic 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('decompress_ms', r['decompress_ms'], 'compress_ms', r['compress_ms'], 'value', r['value'])" || exit 1
  done
done
