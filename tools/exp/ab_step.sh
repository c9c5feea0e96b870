# same-box A/B of the headline step: $1 = environment switch that selects variant A (an experiment switch of the build under test)
B="python bench.py --stream-requests 0 --mix 0 --no-sd3 --no-stages --no-cached-mix --no-two-model --no-cpu-baseline --no-parity --no-roofline --steps 40"
for i in 1 2 3; do
  env $1=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A ($1=1)', d['ms_per_step'])"
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B        ', d['ms_per_step'])"
done
