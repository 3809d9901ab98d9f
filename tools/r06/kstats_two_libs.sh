# per-kernel times (rocprofv3 --kernel-trace --stats, one context) of two builds of the library, one after the other: $1 = tag, $2 = first library, $3 = second
L=$PWD/yaha_amd/csrc
for v in ${2:-libyaha_hip_var.so} ${3:-libyaha_hip.so}; do
  echo "== $v"; YAHA_HIP_LIB=$L/$v tools/measure.sh kstats ${1}_${v%.so} 1 2>&1 | grep -E "sum of|k_ext_rows_pk|k_ext_trace_pk|under the profiler"
done
