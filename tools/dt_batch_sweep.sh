# the one-pass decoder with 1..6 tiles per workgroup (WAH_DT_BATCH; 3..6 need tools/experiments/decode_tile_batches.diff), tools/decode_ab.py
cd $GRAFT_REPO_ROOT
export WAH_LIB_PATH=$PWD/gpu-wah_amd/libwah_hip_exp.so  # the experiment build (make -C gpu-wah_amd exp): the shipped library reads none of these switches
for b in ${BATCHES:-2 3 4 5 6 2}; do
  WAH_DT_BATCH=$b timeout -k 10 200 python tools/decode_ab.py ${KINDS:-sparse dense} || exit 1
done
