# A/B of the general decoder's routes on the 1 GiB bitmaps (tools/decode_ab.py): two launches, one pass with 1 / 2 tiles per workgroup
cd $GRAFT_REPO_ROOT
export WAH_LIB_PATH=$PWD/gpu-wah_amd/libwah_hip_exp.so  # the experiment build (make -C gpu-wah_amd exp): the shipped library reads none of these switches
WAH_DECODE_TWO_PASS=1 timeout -k 10 120 python tools/decode_ab.py sparse dense || exit 1
for b in 1 2; do
  WAH_DT_BATCH=$b timeout -k 10 120 python tools/decode_ab.py sparse dense || exit 1
done
