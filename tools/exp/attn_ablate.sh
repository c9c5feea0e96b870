#!/bin/bash
# run tools/attn_bench.py over the ablation builds (GPU box): bash tools/exp/attn_ablate.sh > gpurun_out/attn_ablate.log
echo "== base"; ITERS=20 python tools/attn_bench.py 2>&1 | grep prescaled
for v in ${VARIANTS:-1 2 4 6 8 16 24 32 64}; do
  echo "== MX_AEXP=$v"; MXDENOISE_LIB=build/exp/libmx_attn_e$v.so ITERS=20 python tools/attn_bench.py 2>&1 | grep prescaled
done
