#!/bin/bash
# tools/e2e_quick.sh [keys...] -- on the GPU box: the bench's file-to-file keys alone (null bam_null bgzf file bam pgz), each over seconds of
# mapping (--loop-input), with the driver's own busy fractions.  Needs nothing but the tree: index and sample are made here.
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
read FA F1 F2 NP < <(python3 tools/e2e_setup.py 5000000 4 | tail -1)
D="./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 -t 32 --verbose"
show() { grep -E "mapping wall|busy fractions" | sed -E 's/.*mapping wall ([0-9.]+)s.*/wall \1/; s/.*busy fractions of the mapping wall: //' | tr '\n' ' '; echo; }
for k in ${@:-null bam_null bgzf}; do
  case $k in
    null)     echo -n "null_sink (x8, $((NP*2*8)) reads): "; $D --seq1 $F1 --seq2 $F2 -o /dev/null --loop-input 8 2>&1 | show ;;
    bam_null) echo -n "bam_null_sink (x8): "; $D --seq1 $F1 --seq2 $F2 -o /dev/null --bam --loop-input 8 2>&1 | show ;;
    file)     echo -n "file (x1): "; $D --seq1 $F1 --seq2 $F2 -o $W/o.sam 2>&1 | show; rm -f $W/o.sam ;;
    bam)      echo -n "bam (x3): "; $D --seq1 $F1 --seq2 $F2 -o $W/o.bam --bam --loop-input 3 2>&1 | show; rm -f $W/o.bam ;;
    bgzf)
      python3 - <<PY
import sys; sys.path.insert(0, ".")
import bench
for k in (1, 2):
    with open("$W/e2e_%d.fq" % k, "rb") as f: data = f.read()
    bench.write_bgzf("$W/b_%d.fq.gz" % k, data, level=1, threads=16)
PY
      echo -n "bgzf (x6): "; $D --seq1 $W/b_1.fq.gz --seq2 $W/b_2.fq.gz -o /dev/null --loop-input 6 2>&1 | show ;;
    pgz)
      python3 - <<PY
import sys; sys.path.insert(0, ".")
import bench
for k in (1, 2):
    with open("$W/e2e_%d.fq" % k, "rb") as f: data = f.read()
    with open("$W/e2e2_%d.fq" % k, "wb") as f: f.write(data); f.write(data)            # (twice the sample: the run takes about a second)
    bench.write_gzip_one_member("$W/g_%d.fq.gz" % k, "$W/e2e2_%d.fq" % k)
PY
      echo -n "plain_gzip (one member per file, host inflater): "; $D --seq1 $W/g_1.fq.gz --seq2 $W/g_2.fq.gz -o /dev/null 2>&1 | show; rm -f $W/e2e2_?.fq $W/g_?.fq.gz ;;
  esac
done
