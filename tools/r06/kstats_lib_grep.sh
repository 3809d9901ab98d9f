# per-kernel times (one context) of one build of the library, lines matching a pattern: $1 = tag, $2 = library file in yaha_amd/csrc, $3 = egrep pattern
L=$PWD/yaha_amd/csrc
YAHA_HIP_LIB=$L/${2:-libyaha_hip.so} tools/measure.sh kstats ${1:-r06} 1 2>&1 | grep -E "sum of|under the profiler|${3:-k_gap}"
