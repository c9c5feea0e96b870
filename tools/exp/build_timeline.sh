#!/bin/bash
# MX_EXP=8: the 256x256 persistent GEMM (gemm_v4) and the 256x160/128 ping-pong GEMM (gemm_v5) with wall-clock stamps per workgroup
# (diagnostic only) -> sduss_amd/libmxdenoise_exp8.so (in-tree, git-ignored: build/ does not travel to a GPU lease); read by tools/exp/timeline_v4.py
#   MXDENOISE_LIB=sduss_amd/libmxdenoise_exp8.so python tools/exp/timeline_v4.py [v4]
set -e
cd "$(dirname "$0")/../.."
make -C sduss_amd/csrc -j4 > /dev/null
mkdir -p build/exp
for f in gemm_bf16_v4 gemm_bf16_v5; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMX_EXP=8 -c sduss_amd/csrc/$f.hip -o build/exp/${f}_exp8.o
done
OBJS=$(ls build/obj/*.o | grep -v -e gemm_bf16_v4.o -e gemm_bf16_v5.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sduss_amd/libmxdenoise_exp8.so build/exp/gemm_bf16_v4_exp8.o build/exp/gemm_bf16_v5_exp8.o $OBJS
ls -la sduss_amd/libmxdenoise_exp8.so
