set -e
Q="--stream-requests 0 --mix 0 --no-cpu-baseline --no-stages --no-roofline --no-sd3 --no-parity --no-cached-mix --no-two-model --steps 20 --warmup 5"
for i in 1 2; do
  MX_ATTN_TAIL=0 python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail OFF ms/step', d['ms_per_step'])"
  MX_ATTN_TAIL=1 python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail ON  ms/step', d['ms_per_step'])"
done
