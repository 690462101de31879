# decode_expand_kernel's workgroups per launch (WAH_EXPAND_WANT, default 4096) on the clustered GiB, tools/decode_ab.py
cd $GRAFT_REPO_ROOT
export WAH_LIB_PATH=$PWD/gpu-wah_amd/libwah_hip_exp.so  # the experiment build (make -C gpu-wah_amd exp): the shipped library reads none of these switches
for w in ${WANTS:-4096 1536 2304 3072 6144 8192 12288 16384 4096}; do
  WAH_EXPAND_WANT=$w timeout -k 10 200 python tools/decode_ab.py clustered || exit 1
done
