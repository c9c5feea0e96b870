#!/bin/bash
# MX_EXP=8: the 256x256 persistent GEMM (gemm_v4) and the 256x160/128 ping-pong GEMM (gemm_v5) with wall-clock stamps per workgroup
# (diagnostic only) -> build/exp/libmx_exp8.so; read by tools/exp/timeline_v4.py
set -e
cd "$(dirname "$0")/../.."
make -C sduss_amd/csrc -j4 > /dev/null
mkdir -p build/exp
OBJ=build/obj
for f in gemm_bf16_v4 gemm_bf16_v5; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMX_EXP=8 -c sduss_amd/csrc/$f.hip -o build/exp/${f}_exp8.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libmx_exp8.so $OBJ/gemm_bf16_v2.o $OBJ/gemm_bf16.o build/exp/gemm_bf16_v4_exp8.o \
  build/exp/gemm_bf16_v5_exp8.o $OBJ/attention.o $OBJ/norm.o $OBJ/elementwise.o $OBJ/gn_halo_nchw.o $OBJ/unet_sdxl.o $OBJ/mmdit_sd3.o $OBJ/capi.o \
  $OBJ/clip_text.o $OBJ/t5_text.o $OBJ/vae_sdxl.o $OBJ/patch_cache.o
# the same stamps in the two-segment experiment kernel (A/B of the K loop and of the shader clock under it)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMX_EXP=8 -Isduss_amd/csrc -c tools/exp/gemm_bf16_v4_twophase.hip -o build/exp/gemm_bf16_v4_twophase_exp8.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libmx_exp8_twophase.so $OBJ/gemm_bf16_v2.o $OBJ/gemm_bf16.o build/exp/gemm_bf16_v4_twophase_exp8.o \
  build/exp/gemm_bf16_v5_exp8.o $OBJ/attention.o $OBJ/norm.o $OBJ/elementwise.o $OBJ/gn_halo_nchw.o $OBJ/unet_sdxl.o $OBJ/mmdit_sd3.o $OBJ/capi.o \
  $OBJ/clip_text.o $OBJ/t5_text.o $OBJ/vae_sdxl.o $OBJ/patch_cache.o
ls -la build/exp/libmx_exp8*.so
