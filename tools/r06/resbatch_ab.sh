python -m pytest tests -m gpu -x -q -k "dp or stage or golden or parity or seed" 2>&1 | tail -3
L=$PWD/yaha_amd/csrc
tools/measure.sh ab YAHA_HIP_LIB $L/libyaha_hip_var.so $L/libyaha_hip.so
YAHA_HIP_LIB=$L/libyaha_hip_prof.so python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --contexts 1 --blocks 1 2>&1 | grep "YD_PROF" | tail -4
