# the post-filter stage's launches held to a few workgroups a CU (YGPU_OQC_GRID="a,b,c,d" per class): hot path / + D2H / + device post-filter + D2H, reads/s, four contexts
# (the kernel loop this needs was measured and not kept: see profiles/r06_postfilter_grid_cap.txt)
B="python bench.py --no-cpu-baseline --quick-extras --steps 10 --warmup 2"
$B > /dev/null 2>&1
for rep in 1 2 3; do for g in "0,0,0,0" "0,0,1,1" "4,2,1,1" "2,1,1,1"; do
  YGPU_OQC_GRID=$g $B 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('grid $g', round(j['value']), round(j.get('value_with_d2h',0)), round(j.get('value_with_postfilter',0)), round(j.get('value_with_postfilter',0)/j['value'],3))"
done; done
