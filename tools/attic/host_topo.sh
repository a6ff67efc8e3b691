#!/bin/bash
# tools/host_topo.sh -- what the GPU box's host looks like to the I/O path: sockets, NUMA nodes, memory per node, the GPU's node
lscpu | grep -E "Model name|Socket|NUMA|Thread|Core|^CPU\(s\)" 
for n in /sys/devices/system/node/node*; do echo "$(basename $n): cpus $(cat $n/cpulist) mem $(grep MemTotal $n/meminfo | awk '{print $4/1048576 " GB"}') free $(grep MemFree $n/meminfo | awk '{print $4/1048576 " GB"}')"; done
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) $(cat $d/uevent 2>/dev/null | grep PCI_SLOT)"; done
cat /proc/meminfo | head -5
which numactl 2>/dev/null; numactl -H 2>/dev/null | head -20
nproc; taskset -p $$
df -h /tmp | tail -1; mount | grep -E " /tmp | / " | head -3
