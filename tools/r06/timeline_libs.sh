# four-context timelines for several builds of the library: tests of the seed stage first, then per build the kernels that stretch most under contention. $@ = library files in yaha_amd/csrc
L=$PWD/yaha_amd/csrc
for v in "$@"; do
  echo "== $v"
  YAHA_HIP_LIB=$L/$v python -m pytest tests -m gpu -x -q -k "seed or stage or golden or sort" 2>&1 | tail -1
  YAHA_HIP_LIB=$L/$v tools/measure.sh kstats tl_$v 4 2>&1 | grep -E "under the profiler"
  python3 tools/overlap.py gpurun_out/tl_$v/kernel_trace.csv | head -14 | cut -c1-150
  rm -f gpurun_out/tl_$v/kernel_trace.csv
done
