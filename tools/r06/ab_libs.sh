# the default bench (four contexts) with two builds of the library, alternating, REPS rounds: $1, $2 = library files in yaha_amd/csrc
L=$PWD/yaha_amd/csrc
B="python bench.py --no-cpu-baseline --no-extras"
$B --steps 2 --warmup 1 > /dev/null 2>&1
p() { python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['value']), round(j['ms_per_step'],2), round(j['ms_per_step_min'],2), round(j['ms_per_step_max'],2))"; }
for rep in $(seq 1 ${REPS:-3}); do
  YAHA_HIP_LIB=$L/$1 $B --steps ${STEPS:-20} --warmup 5 2>/dev/null | tail -1 | p "$1"
  YAHA_HIP_LIB=$L/$2 $B --steps ${STEPS:-20} --warmup 5 2>/dev/null | tail -1 | p "$2"
done
