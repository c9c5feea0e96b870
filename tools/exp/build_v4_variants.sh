#!/bin/bash
# Ablation builds of the 8-phase GEMM (results garbage, timing only) -> build/exp/libmx_v4e<N>.so
#   1 no MFMA | 2 no LDS-DMA | 3 no fragment reads | 12 = 1+2 (reads + barriers only) | 13 = 1+3 (DMA + barriers only)
set -e
VARIANTS=${VARIANTS:-"1 2 3 12 13 23 123"}
cd "$(dirname "$0")/../.."
mkdir -p build/exp
OBJ=build/obj
for v in $VARIANTS; do :; done
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMX_EXP=$v -c sduss_amd/csrc/gemm_bf16_v4.hip -o build/exp/gemm_v4_e$v.o &
done
wait
for v in $VARIANTS; do :; done
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libmx_v4e$v.so $OBJ/gemm_bf16_v2.o $OBJ/gemm_bf16.o $OBJ/gemm_bf16_v3.o build/exp/gemm_v4_e$v.o \
    $OBJ/attention.o $OBJ/norm.o $OBJ/elementwise.o $OBJ/gn_halo_nchw.o $OBJ/unet_sdxl.o $OBJ/mmdit_sd3.o $OBJ/capi.o
done
ls build/exp/libmx_v4e*.so
