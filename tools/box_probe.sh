#!/bin/bash
# what the GPU box's host gives this user: CPUs, cgroup quota, NUMA, memory, disk
mkdir -p gpurun_out
{
echo "== nproc: $(nproc)   cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|MHz|L3'
free -g | head -2
df -h /tmp /dev/shm | cat
cat /sys/fs/cgroup/memory.max 2>/dev/null
cat /sys/fs/cgroup/cpu.stat 2>/dev/null
./tools/cpu_probe 1 8 16 32 64 128 256
cat /sys/fs/cgroup/cpu.stat 2>/dev/null
} > gpurun_out/r03_box_probe.txt 2>&1
cat gpurun_out/r03_box_probe.txt
