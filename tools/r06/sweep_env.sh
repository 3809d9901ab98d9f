# the default bench under values of one environment switch, alternating, REPS rounds: $1 = variable, $2.. = values
V=$1; shift
B="python bench.py --no-cpu-baseline --no-extras"
$B --steps 2 --warmup 1 > /dev/null 2>&1
for rep in $(seq 1 ${REPS:-2}); do for x in "$@"; do
  env $V=$x $B --steps 12 --warmup 2 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$V=$x', round(j['value']), round(j['ms_per_step'],2), round(j['ms_per_step_min'],2), round(j['ms_per_step_max'],2))"
done; done
