#!/bin/bash
# Ablation builds of attn_fwd_dma_kernel (results garbage, timing only) -> build/exp/libmx_attn_e<N>.so
#   bits: 1 no exp2 | 2 no LDS-DMA after the prologue | 4 no barrier | 8 no QK MFMAs | 16 no PV MFMAs | 32 one workgroup per CU | 64 two per CU
set -e
VARIANTS=${VARIANTS:-"1 2 4 6 8 16 24 32 64"}
cd "$(dirname "$0")/../.."
mkdir -p build/exp
OBJ=build/obj
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMX_AEXP=$v -c sduss_amd/csrc/attention.hip -o build/exp/attention_e$v.o &
done
wait
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libmx_attn_e$v.so $OBJ/gemm_bf16_v2.o $OBJ/gemm_bf16.o $OBJ/gemm_bf16_v3.o $OBJ/gemm_bf16_v4.o \
    build/exp/attention_e$v.o $OBJ/norm.o $OBJ/elementwise.o $OBJ/gn_halo_nchw.o $OBJ/unet_sdxl.o $OBJ/mmdit_sd3.o $OBJ/vae_sdxl.o $OBJ/capi.o
done
ls build/exp/libmx_attn_e*.so
