#!/bin/bash
# CPU topology of the box: logical CPUs, their hyper-thread siblings, the cgroup's CPU quota
nproc; lscpu | grep -E "Model name|Thread|Core|Socket|NUMA|^CPU\(s\)"
for c in /sys/devices/system/cpu/cpu[0-9]*; do echo "$(basename $c): siblings $(cat $c/topology/thread_siblings_list 2>/dev/null) core $(cat $c/topology/core_id 2>/dev/null)"; done | head -40
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
python -c "import os; print('affinity', sorted(os.sched_getaffinity(0)))"
