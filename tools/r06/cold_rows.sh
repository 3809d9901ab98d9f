# the first rows launch of a starting pipeline at full size (YGPU_ROWS_COLD_MS, default 100; 0 = off): the bench in the driver's form and with blocks of 6 steps, alternating
B="python bench.py --no-cpu-baseline --no-extras"
$B --steps 2 --warmup 1 > /dev/null 2>&1
p() { python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['value']), round(j['ms_per_step'],2), j['ms_per_step_blocks'])"; }
for rep in 1 2 3; do for c in 0 100; do
  YGPU_ROWS_COLD_MS=$c $B --steps 20 --warmup 5 2>/dev/null | tail -1 | p "K=20 cold_ms=$c"
done; done
for rep in 1 2; do for c in 0 100; do
  YGPU_ROWS_COLD_MS=$c $B --steps 6 --warmup 1 2>/dev/null | tail -1 | p "K=6 cold_ms=$c"
done; done
