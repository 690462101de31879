# tile shapes of a compress launch below two rounds of the chip (WAH_SHAPE = "body tiles of 3 pairs per wave, pairs per wave of the rest")
cd $GRAFT_REPO_ROOT
export WAH_LIB_PATH=$PWD/gpu-wah_amd/libwah_hip_exp.so  # the experiment build (make -C gpu-wah_amd exp): the shipped library reads none of these switches
export KINDS="sparse dense"
echo "== default"; timeout -k 10 100 python tools/compress_time.py 128 64 2>&1 | grep -v amdgpu.ids
for sh in 66,2 0,2 200,2 340,2 0,1 512,1 33,2 176,2 100,1; do
  echo "== WAH_SHAPE=$sh"; WAH_SHAPE=$sh timeout -k 10 100 python tools/compress_time.py 128 64 2>&1 | grep -v amdgpu.ids || exit 1
done
