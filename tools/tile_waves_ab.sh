cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo "== shipped"; timeout -k 10 100 python tools/compress_time.py 1024 2>&1 | grep -v amdgpu
  echo "== exp8 pairs=2"; WAH_LIB_PATH=$PWD/gpu-wah_amd/libwah_hip_exp.so WAH_WAVE_PAIRS=2 timeout -k 10 100 python tools/compress_time.py 1024 2>&1 | grep -v amdgpu
  echo "== exp6 pairs=2"; WAH_LIB_PATH=$PWD/gpu-wah_amd/libwah_hip_exp6.so WAH_WAVE_PAIRS=2 timeout -k 10 100 python tools/compress_time.py 1024 2>&1 | grep -v amdgpu
  echo "== exp6 pairs=3"; WAH_LIB_PATH=$PWD/gpu-wah_amd/libwah_hip_exp6.so WAH_WAVE_PAIRS=3 timeout -k 10 100 python tools/compress_time.py 1024 2>&1 | grep -v amdgpu
  echo "== exp6 pairs=1"; WAH_LIB_PATH=$PWD/gpu-wah_amd/libwah_hip_exp6.so WAH_WAVE_PAIRS=1 timeout -k 10 100 python tools/compress_time.py 1024 2>&1 | grep -v amdgpu
done
