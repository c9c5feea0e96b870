#!/bin/bash
# Diagnostic build of the chained attention-tail launch with wall-clock stamps per work item (attn_tail.hip, MX_TAIL_STAMPS): the library with that one object
# replaced, left IN-TREE as sduss_amd/libmxdenoise_tailstamps.so (git-ignored; build/ does not travel to a GPU lease).  Use:
#   MXDENOISE_LIB=sduss_amd/libmxdenoise_tailstamps.so python tools/exp/tail_timeline.py
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMX_TAIL_STAMPS -c sduss_amd/csrc/attn_tail.hip -o build/exp/attn_tail_stamps.o
OBJS=$(ls build/obj/*.o | grep -v attn_tail.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sduss_amd/libmxdenoise_tailstamps.so build/exp/attn_tail_stamps.o $OBJS
ls -la sduss_amd/libmxdenoise_tailstamps.so
