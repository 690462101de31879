# same-box A/B of the general decoder of two builds (tools/decode_ab.py): usage: tools/ab_lib_decode.sh <other .so> [kinds...]
cd $GRAFT_REPO_ROOT
other=$1; shift
for i in 1 2; do
  for lib in "" "$other"; do
    echo "== ${lib:-shipped}"
    WAH_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 200 python tools/decode_ab.py "$@" || exit 1
  done
done
