#!/bin/bash
# One script for the measurements DESIGN.md quotes (run on the GPU box through gpurun; output under gpurun_out/<tag>...).
#   tools/measure.sh step   TAG ["pytest -k expression"]     stage tests, default bench x3, one context, per-kernel times of one context
#   tools/measure.sh ab     VAR A B [contexts]                A/B of an environment switch on the default bench, alternating, four pairs of 12 steps
#   tools/measure.sh kstats TAG CONTEXTS [bench flags]        rocprofv3 --kernel-trace --stats of the bench with CONTEXTS batches in flight
#   tools/measure.sh cli    TAG ["opts" ...]                  the command line on 1 M reads of 1 kbp, one run per option string, 25 s apart (YAHA_STATS line)
#   tools/measure.sh cli10k TAG ["opts" ...]                  the same on 32 768 reads of 10 kbp
#   tools/measure.sh clienv TAG VAR v1 v2 ...                 the command line on 1 M reads under VAR=v, runs 25 s apart
#   tools/measure.sh n2     TAG                               bench.py --gpus 2 as two ranks on one GPU over gloo (100 Mbp genome, one context each)
#   tools/measure.sh oqc    TAG                               post-filter stage: in-kernel timers, kernel stats with the stage in the run
# (the earlier rounds' one-off scripts -- r03_step.sh, r03_ab.sh, r03_cli_sweep.sh, r03_cli_ctx4.sh, r03_cli_10k.sh, r03_first.sh, r03_oqcmax.sh, r03_upload.sh,
#  r03_n2_dryrun.sh, r03_oqc_prof.sh, ctx_sweep.sh, cli_probe.sh -- are modes of this one)
MODE=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-extras"
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['value']), round(j['ms_per_step'],2))"; }
warm() { $B --steps 2 --warmup 1 > /dev/null 2>&1; }
reads1k() { R1=$C/e2e_n1048576_l1000_s3000.fa; [ -f $R1 ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R1 --seed 3000 --n 1048576 --len 1000 --div 0.017; }
reads10k() { R10=$C/cli10k.fa; [ -f $R10 ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R10 --seed 77 --n 32768 --len 10000 --div 0.034; }
kernel_table() {   # $1 = directory with the rocprofv3 output, $2 = steps in the run (timed + warm-up)
python3 - "$1" "$2" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f))); n = float(sys.argv[2])
tot = sum(float(r['TotalDurationNs']) for r in rows); lib = sum(float(r['TotalDurationNs']) for r in rows if 'rocprim' in r['Name'] or 'hipcub' in r['Name'])
print("sum of kernel durations per step: %.2f ms (library kernels %.2f) over %g steps (the first batch's extra passes included)" % (tot / 1e6 / n, lib / 1e6 / n, n))
for r in rows[:45]: print("%-72s calls %5d  avg %9.3f ms  per step %8.3f ms" % (r['Name'][:72], int(r['Calls']), float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6 / n))
PY
}
case $MODE in
step)
  TAG=${1:-step}; K=${2:-"seed or sort or stage or golden or bench_scale or chain"}
  python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -3
  warm
  for rep in 1 2 3; do $B --steps 12 --warmup 2 2>/dev/null | line "default contexts"; done
  $B --steps 6 --warmup 2 --contexts 1 2>/dev/null | line "one context"
  O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 8 --warmup 2 --contexts 1 --blocks 1 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
  kernel_table $O/stats 10; cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/stats ;;
ab)
  V=$1; A=$2; Bv=$3; CT=${4:-}; warm
  for rep in 1 2 3 4; do for x in $A $Bv; do env $V=$x $B --steps 12 --warmup 2 ${CT:+--contexts $CT} 2>/dev/null | line "$V=$x"; done; done
  for x in $A $Bv; do env $V=$x $B --steps 6 --warmup 2 --contexts 1 2>/dev/null | line "one context $V=$x"; done ;;
kstats)
  TAG=${1:-kstats}; CT=${2:-1}; shift; shift; warm
  O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 8 --warmup 2 --contexts $CT --blocks 1 --no-cpu-baseline --no-extras "$@" > $O/bench.json 2> $O/bench.err
  kernel_table $O/stats $((8 + 3 * CT)); cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; cp $(find $O/stats -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv; rm -rf $O/stats; tail -1 $O/bench.json | line "under the profiler" ;;      # 8 timed steps + (first pass + 2 warm-up) per context
cli|cli10k)
  TAG=${1:-cli}; shift; warm
  if [ $MODE = cli ]; then reads1k; Q=$R1; else reads10k; Q=$R10; fi
  [ $# -eq 0 ] && set -- "" ""
  yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam 2>/dev/null
  for opts in "$@"; do
    sleep 25          # (the driver scrubs what the previous process freed; a run started right away pays seconds for its first allocations)
    YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam $opts 2>&1 | grep "stats" | sed "s/^/[$opts] /" | cut -c1-700 | tee -a gpurun_out/$TAG.txt
  done
  rm -f /dev/shm/o.sam ;;
clienv)
  TAG=${1:-clienv}; V=$2; shift; shift; warm; reads1k
  yaha_amd/csrc/yaha -x $X -q $R1 -osh /dev/shm/o.sam 2>/dev/null
  for x in "$@"; do sleep 25; env $V=$x YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $R1 -osh /dev/shm/o.sam 2>&1 | grep "stats" | sed "s/^/[$V=$x] /" | cut -c1-700 | tee -a gpurun_out/$TAG.txt; done
  rm -f /dev/shm/o.sam ;;
n2)
  TAG=${1:-n2}
  YAHA_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 1 --genome-mbp 100 --contexts 1 --e2e-reads 65536 > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err; echo "rc $?"
  tail -5 gpurun_out/$TAG.err | cut -c1-300
  python - gpurun_out/$TAG.json <<'PY'
import json, sys; j = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print({k: j.get(k) for k in ('n_gpus', 'value', 'ms_per_step', 'e2e_reads_per_s', 'steady_reads_per_s')}); print(j.get('end_to_end'))
PY
  ;;
oqc)
  TAG=${1:-oqc}
  YGPU_OQC_PROF=1 python bench.py --steps 2 --warmup 1 --contexts 1 --no-cpu-baseline --e2e-reads 16384 2>&1 >/dev/null | grep "post-filter class" | tail -5
  O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 4 --warmup 1 --contexts 1 --no-cpu-baseline --e2e-reads 16384 > $O/bench.json 2> $O/bench.err
  kernel_table $O/stats 5 | grep -E "sum of|k_oqc" ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
