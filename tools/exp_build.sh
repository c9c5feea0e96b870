#!/bin/bash
# Diagnostic builds of the GEMM kernel with one pipeline component removed (results are garbage; timing only):
#   build/exp/libmx_exp1.so no MFMA | exp2 no LDS-DMA inside the K loop | exp3 no LDS fragment reads
# Use: MXDENOISE_LIB=build/exp/libmx_exp1.so python tools/gemm_bench.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp
OBJ=build/obj
for v in 1 2 4 5 6; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMX_EXP=$v -c sduss_amd/csrc/gemm_bf16_v2.hip -o build/exp/gemm_v2_exp$v.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMX_EXP=$v -c sduss_amd/csrc/gemm_bf16_v3.hip -o build/exp/gemm_v3_exp$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libmx_exp$v.so build/exp/gemm_v2_exp$v.o \
    $OBJ/gemm_bf16.o build/exp/gemm_v3_exp$v.o $OBJ/attention.o $OBJ/norm.o $OBJ/elementwise.o $OBJ/gn_halo_nchw.o $OBJ/unet_sdxl.o $OBJ/mmdit_sd3.o $OBJ/capi.o
done
ls -la build/exp/*.so
