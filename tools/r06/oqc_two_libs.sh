# the post-filter stage's in-kernel timers and kernel times for two builds of the library: $1 first, $2 second
L=$PWD/yaha_amd/csrc
python -m pytest tests -m gpu -x -q -k "postfilter or oqc or filter" 2>&1 | tail -2
for v in ${1:-libyaha_hip_var.so} ${2:-libyaha_hip.so}; do echo "== $v"; YAHA_HIP_LIB=$L/$v tools/measure.sh oqc r06_oqc_${v%.so} 2>&1 | cut -c1-400; done
