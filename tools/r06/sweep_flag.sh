# the default bench with and without one environment switch (set to 1 / unset), alternating, REPS rounds: $1 = variable
V=$1
B="python bench.py --no-cpu-baseline --no-extras"
$B --steps 2 --warmup 1 > /dev/null 2>&1
p() { python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['value']), round(j['ms_per_step'],2), round(j['ms_per_step_min'],2), round(j['ms_per_step_max'],2))"; }
for rep in $(seq 1 ${REPS:-3}); do
  env $V=1 $B --steps 12 --warmup 2 2>/dev/null | tail -1 | p "$V=1"
  $B --steps 12 --warmup 2 2>/dev/null | tail -1 | p "$V unset"
done
