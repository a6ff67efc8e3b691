#!/bin/bash
# tools/e2e_probe.sh <outdir> [pairs=10000000] -- file-to-file rates of bmbs_search on the GPU box for a few driver settings (batch size,
# contexts, output parts, sink).  Needs the bench's 3.1 Gb index and FASTQ sample under $BMBS_BENCH_DIR (python bench.py writes both).
O=${1:-gpurun_out/e2e}; N=${2:-10000000}
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
mkdir -p $O
read FA F1 F2 NP < <(python3 tools/e2e_setup.py $((N / 4)) 4 | tail -1)
echo "inputs: $FA $F1 $F2 pairs=$NP"
run() { # label, args...
  local label=$1; shift
  ./bitmapperbs_amd/bmbs_search --search $FA --seq1 $F1 --seq2 $F2 -e 0.08 -t 32 --verbose "$@" 2> $O/$label.err > /dev/null
  grep "mapping wall" $O/$label.err | sed "s/^/$label: /" | cut -c1-420
  rm -f $W/o.sam $W/o.sam.part*
}
run null_c3 -o /dev/null --contexts 3
run null_c2 -o /dev/null --contexts 2
run null_c4 -o /dev/null --contexts 4
run null_c3_b250 -o /dev/null --contexts 3 --batch 250000
run null_c3_b1m -o /dev/null --contexts 3 --batch 1000000
run null_c3_p2 -o /dev/null --contexts 3 --out-parts 2
run file_c3 -o $W/o.sam --contexts 3
run file_c3_p2 -o $W/o.sam --contexts 3 --out-parts 2
run file_c3_p4 -o $W/o.sam --contexts 3 --out-parts 4
run file_c4_p8 -o $W/o.sam --contexts 4 --out-parts 8
