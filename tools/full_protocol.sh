#!/bin/bash
# BASELINE.md section 4's protocol at size, once per round (the default bench's stream legs are 80 / 16 / 16 requests to fit the driver's run):
# 500 requests, 100 % 1024^2, 50 steps, Poisson arrivals (numpy seed 10086) at the offered load given (requests/s per GPU).
# usage: bash tools/full_protocol.sh <rate> <out.json>     e.g. 1.2 (the reference sweep's top) or 2.0 (past saturation)
set -e
RATE=${1:-1.2}; OUT=${2:-gpurun_out/stream_full_$RATE.json}
python bench.py --steps 5 --warmup 2 --stream-requests 500 --stream-rates $RATE --mix 0 --no-cpu-baseline --no-stages --no-roofline --no-sd3 --no-parity --no-cached-mix --no-two-model > $OUT.log 2>&1
tail -1 $OUT.log > $OUT
python -c "
import json,sys
d=json.load(open('$OUT'))
s=d.get('stream') or d.get('streams') or {k:v for k,v in d.items() if 'stream' in k}
print(json.dumps(s)[:1500])
"
