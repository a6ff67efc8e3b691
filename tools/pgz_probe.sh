#!/bin/bash
# tools/pgz_probe.sh <file.gz> -- on the GPU box: how the block-parallel inflater alone scales with threads, and what the box gives a process
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) ; cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null) ; affinity: $(taskset -p $$ 2>/dev/null)"
g++ -O2 -std=c++17 -Ibitmapperbs_amd/csrc -o /tmp/pgz_test tools/pgz_test.cpp -lz -lpthread || exit 1
for t in 1 4 8 16 32 64 128; do /tmp/pgz_test $1 $t 1048576 q; done
# two files side by side, as the paired-end driver does
for t in 16 32 64; do ( /tmp/pgz_test $1 $t 1048576 q & /tmp/pgz_test ${2:-$1} $t 1048576 q & wait ) 2>&1 | tr '\n' ' '; echo; done
