#!/bin/bash
lscpu | grep -i "numa\|model name\|^CPU(s)\|socket"
for d in /sys/class/drm/card*/device; do echo $d $(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null); done
ls /sys/class/kfd/kfd/topology/nodes/ 2>/dev/null | head
for n in /sys/class/kfd/kfd/topology/nodes/*; do echo $n; grep -E "cpu_cores_count|simd_count|domain|location_id" $n/properties 2>/dev/null | tr '\n' ' '; echo; done 2>/dev/null | head -20
python - <<'PY'
import os
print("affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
PY
which numactl taskset
cat /proc/self/status | grep -i "mems_allowed_list\|cpus_allowed_list"
