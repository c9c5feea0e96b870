#!/bin/bash
# MX_EXP=7: the 256x256 GEMM with per-tile wall-clock stamps (diagnostic only) -> build/exp/libmx_exp7.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/exp
OBJ=build/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMX_EXP=7 -c sduss_amd/csrc/gemm_bf16_v3.hip -o build/exp/gemm_v3_exp7.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libmx_exp7.so $OBJ/gemm_bf16_v2.o $OBJ/gemm_bf16.o build/exp/gemm_v3_exp7.o \
  $OBJ/attention.o $OBJ/norm.o $OBJ/elementwise.o $OBJ/gn_halo_nchw.o $OBJ/unet_sdxl.o $OBJ/mmdit_sd3.o $OBJ/capi.o
ls -la build/exp/libmx_exp7.so
