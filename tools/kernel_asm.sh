#!/bin/bash
# Split the device assembly of `make -C gpu-wah_amd asm` into one file per kernel (build/<name>.s).
set -e
cd "$(dirname "$0")/../gpu-wah_amd"
cat build/wah_compress-hip-amdgcn-amd-amdhsa-gfx950.s build/wah_decode-hip-amdgcn-amd-amdhsa-gfx950.s build/wah_aux-hip-amdgcn-amd-amdhsa-gfx950.s > build/all_kernels.s
S=build/all_kernels.s
for k in compress_kernelILi15E compress_kernelILi7E decode_sums_kernel decode_expand_kernel; do
  sym=$(grep -o "^_ZN[A-Za-z0-9_]*${k}[A-Za-z0-9_]*:" $S | head -1 | tr -d ':')
  [ -n "$sym" ] || continue
  awk -v s="$sym:" '$1==s{p=1} p{print} p&&/s_endpgm/{exit}' $S > build/${k}.s
  echo "$k: $(wc -l < build/${k}.s) lines, valu $(grep -c '^\s*v_' build/${k}.s), salu $(grep -c '^\s*s_' build/${k}.s)"
done
