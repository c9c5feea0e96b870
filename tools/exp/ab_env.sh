# same-lease A/B of the headline step between environment settings: usage  bash tools/exp/ab_env.sh "<label>:<ENV=V ...>" ...   (first = baseline)
# e.g. bash tools/exp/ab_env.sh "old:MX_GN_FOLD=0 MX_TEMB_SIDE=0" "fold:MX_TEMB_SIDE=0" "both:"   -- three interleaved rounds, 20 timed steps each
set -e
Q="--stream-requests 0 --mix 0 --no-cpu-baseline --no-stages --no-roofline --no-sd3 --no-parity --no-cached-mix --no-two-model --steps ${STEPS:-20} --warmup 5 ${BENCH_EXTRA:-}"
for i in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    env $envs python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label ms/step', d['ms_per_step'])"
  done
done
