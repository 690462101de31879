# same-box timing of the general decoder over several builds (tools/decode_ab.py): usage: tools/ab_libs.sh "<kinds>" lib1.so lib2.so ...  ("-" = the shipped library)
cd $GRAFT_REPO_ROOT
kinds=$1; shift
for i in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    if [ "$lib" = "-" ]; then timeout -k 10 200 python tools/decode_ab.py $kinds || exit 1
    else WAH_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/decode_ab.py $kinds || exit 1; fi
  done
done
