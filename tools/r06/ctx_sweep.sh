# the default bench with 3 / 4 / 5 / 6 batches in flight, alternating, REPS rounds
B="python bench.py --no-cpu-baseline --no-extras"
$B --steps 2 --warmup 1 > /dev/null 2>&1
for rep in $(seq 1 ${REPS:-2}); do for c in ${CTXS:-4 5 6}; do
  $B --steps 20 --warmup 5 --contexts $c 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('contexts $c', round(j['value']), round(j['ms_per_step'],2), round(j['ms_per_step_min'],2), round(j['ms_per_step_max'],2))"
done; done
