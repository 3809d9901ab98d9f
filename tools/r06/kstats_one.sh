# DP / stage parity tests, then per-kernel times (one context) of the built library: $1 = tag
python -m pytest tests -m gpu -x -q -k "dp or stage or golden or parity or seed" 2>&1 | tail -2
tools/measure.sh kstats ${1:-r06} 1 2>&1 | grep -E "sum of|k_ext_rows|k_ext_trace|under the profiler"
