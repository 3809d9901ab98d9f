# the workgroup sort's ranking variants, one after the other: per-kernel times (one context) and the sum over k_seg_sort; $@ = library file names in yaha_amd/csrc
L=$PWD/yaha_amd/csrc
for v in "$@"; do
  echo "== $v"
  YAHA_HIP_LIB=$L/$v python -m pytest tests -m gpu -x -q -k "sort" 2>&1 | tail -1
  YAHA_HIP_LIB=$L/$v tools/measure.sh kstats sortv_${v%.so} 1 > gpurun_out/sortv_${v%.so}.txt 2>&1
  grep -E "sum of|under the profiler" gpurun_out/sortv_${v%.so}.txt
  python3 - gpurun_out/sortv_${v%.so}/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); n = 11.0
s = sum(float(r['TotalDurationNs']) for r in rows if 'k_seg_sort' in r['Name'])
print("sum over k_seg_sort: %.3f ms a step" % (s / 1e6 / n))
PY
done
