cd $GRAFT_REPO_ROOT
export KINDS="sparse clustered dense p4 p8 p16 p32 half_dense periodic2 periodic4"
for lib in gpu-wah_amd/libwah_hip.so tools/libwah_swz1984.so tools/libwah_swz4096.so; do
  echo "== $lib"
  WAH_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/compress_time.py 1024 || exit 1
done
