#!/bin/bash
# tools/inflate_probe.sh [pairs] [lib ...] -- on the GPU box: (1) k_bgzf_inflate alone on the bench's FASTQ sample, for the library in
# the tree and every other build of it given (same box, same clocks: the only comparison that holds); (2) BGZF FASTQ pairs through
# the driver's device inflater, with the phase times of every window
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
N=${1:-2500000}; shift
read FA F1 F2 NP < <(python3 tools/e2e_setup.py $N 2 | tail -1)
for lib in "" "$@"; do
  echo "== kernel alone: ${lib:-bitmapperbs_amd/libbmbs_hip.so}"
  python3 tools/inflate_bench.py $F1 $lib 2>&1 | grep -vE "amdgpu.ids" | tail -${TAIL:-2}
done
python3 - <<PY
import sys; sys.path.insert(0, ".")
import bench
for k in (1, 2):
    with open("$W/e2e_%d.fq" % k, "rb") as f: data = f.read()
    bench.write_bgzf("$W/b_%d.fq.gz" % k, data, level=1, threads=16)
PY
for lib in "" "$@"; do
  echo "== driver: ${lib:-bitmapperbs_amd/libbmbs_hip.so}"
  if [ -n "$lib" ]; then mkdir -p $W/lib_$$ && cp $lib $W/lib_$$/libbmbs_hip.so; export LD_LIBRARY_PATH=$W/lib_$$; fi
  for rep in 1 2; do
  BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $W/b_1.fq.gz --seq2 $W/b_2.fq.gz -o /dev/null -t 32 --verbose 2>&1 | grep -E "text open|mapping wall" | tail -4
  done
  unset LD_LIBRARY_PATH
done
