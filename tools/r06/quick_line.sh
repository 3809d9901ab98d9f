# a short bench run (no side legs) and the keys of its line that a change is usually about: $@ = extra bench flags
python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-8} --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('value %.0f  ms/step %.2f [%.2f..%.2f]' % (j['value'], j['ms_per_step'], j['ms_per_step_min'], j['ms_per_step_max']))
print({k: r.get(k) for k in ('frac','frac_in_run','kernel_ms_in_run','kernel_ms_unshared','trace_bytes_useful_frac','trace_bytes_written','trace_bytes_visitable','trace_calls','trace_calls_walking')})
print(j['config']['workload'])"
