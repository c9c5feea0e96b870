#!/bin/bash
# A/B: the library with the TWO-segment form of the 256x256 ping-pong kernel (tools/exp/gemm_bf16_v4_twophase.hip: L | M per K tile, the
# second X sub-tile in its own registers) in place of the shipped four-segment one -> build/exp/libmx_v4_twophase.so.  Measured round 3
# (profiles/r03_h_*): correct (all GEMM / UNet / MMDiT tests), 13 % fewer shader cycles per K tile would be expected from halving the
# barriers, but 4-8 % SLOWER in wall time at every shape: the chip is power-limited under MFMA load and runs the denser schedule at a lower
# shader clock.  Use: MXDENOISE_LIB=build/exp/libmx_v4_twophase.so python tools/gemm_bench.py
set -e
cd "$(dirname "$0")/../.."
make -C sduss_amd/csrc -j4 > /dev/null
mkdir -p build/exp
OBJ=build/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Isduss_amd/csrc -c tools/exp/gemm_bf16_v4_twophase.hip -o build/exp/gemm_bf16_v4_twophase.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libmx_v4_twophase.so $OBJ/gemm_bf16_v2.o $OBJ/gemm_bf16.o build/exp/gemm_bf16_v4_twophase.o \
  $OBJ/gemm_bf16_v5.o $OBJ/attention.o $OBJ/norm.o $OBJ/elementwise.o $OBJ/gn_halo_nchw.o $OBJ/unet_sdxl.o $OBJ/mmdit_sd3.o $OBJ/capi.o \
  $OBJ/clip_text.o $OBJ/t5_text.o $OBJ/vae_sdxl.o
ls -la build/exp/libmx_v4_twophase.so
