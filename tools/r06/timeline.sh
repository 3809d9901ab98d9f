# how four batches in flight share the device: kernel traces of the bench with four contexts and with one, tools/overlap.py over both: $1 = tag
T=${1:-r06_timeline}
tools/measure.sh kstats ${T}_ctx1 1 > /dev/null 2>&1
tools/measure.sh kstats ${T}_ctx4 4 2>&1 | grep -E "sum of|under the profiler"
python3 tools/overlap.py gpurun_out/${T}_ctx4/kernel_trace.csv gpurun_out/${T}_ctx1/kernel_trace.csv 8 > gpurun_out/${T}.txt; head -45 gpurun_out/${T}.txt
cp gpurun_out/${T}_ctx1/kernel_stats.csv gpurun_out/${T}_kernel_stats_one_context.csv; cp gpurun_out/${T}_ctx4/kernel_stats.csv gpurun_out/${T}_kernel_stats_four_contexts.csv
rm -f gpurun_out/${T}_ctx1/kernel_trace.csv gpurun_out/${T}_ctx4/kernel_trace.csv
