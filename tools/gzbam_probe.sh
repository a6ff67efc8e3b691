#!/bin/bash
# tools/gzbam_probe.sh <outdir> [pairs per rep=2500000] -- on the GPU box: gzip-in / BAM-out rates of bmbs_search for a few thread counts.
# Needs the bench's 3.1 Gb index and FASTQ sample under $BMBS_BENCH_DIR (tools/e2e_setup.py writes both).
O=${1:-gpurun_out/gzbam}; N=${2:-2500000}
W=${BMBS_BENCH_DIR:-/tmp/bmbs_bench}
mkdir -p $O
echo "host: $(nproc) cpus; $(lscpu | grep -E 'Model name|Socket|Core' | tr -s ' ' | tr '\n' ';')"
read FA F1 F2 NP < <(python3 tools/e2e_setup.py $N 4 | tail -1)
echo "inputs: $FA $F1 $F2 pairs=$NP"
REC=315
GZN=$((N * 2))
( head -c $((GZN * REC)) $F1 | gzip -1 -c > $W/g_1.fq.gz ) &
( head -c $((GZN * REC)) $F2 | gzip -1 -c > $W/g_2.fq.gz ) &
wait
ls -la $W/g_1.fq.gz $W/g_2.fq.gz
run() { # label, reads, args...
  local label=$1; shift; local reads=$1; shift
  ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --verbose "$@" 2> $O/$label.err > /dev/null
  local wall=$(grep "mapping wall" $O/$label.err | sed 's/.*mapping wall \([0-9.]*\)s.*/\1/')
  echo "$label: wall $wall s -> $(python3 -c "print(round($reads/$wall/1e6,1))") M reads/s | $(grep 'stage busy' $O/$label.err | sed 's/.*(pipeline/(pipeline/' | cut -c1-260)"
}
ALL=$((NP * 2)); GZR=$((GZN * 2))
python3 - <<PY
import sys; sys.path.insert(0, ".")
import bench
for k in (1, 2):
    with open("$W/e2e_%d.fq" % k, "rb") as f: data = f.read($GZN * $REC)
    bench.write_bgzf("$W/b_%d.fq.gz" % k, data, level=1, threads=16)
PY
ls -la $W/b_1.fq.gz
run text_null $ALL --seq1 $F1 --seq2 $F2 -o /dev/null -t 32
for t in 16 32; do run gz_t$t $GZR --seq1 $W/g_1.fq.gz --seq2 $W/g_2.fq.gz -o /dev/null -t $t; done
run bgzf_device_t32 $GZR --seq1 $W/b_1.fq.gz --seq2 $W/b_2.fq.gz -o /dev/null -t 32
BMBS_GZ_DEVICE=0 run bgzf_host_t32 $GZR --seq1 $W/b_1.fq.gz --seq2 $W/b_2.fq.gz -o /dev/null -t 32
run bgzf_device_bam $GZR --seq1 $W/b_1.fq.gz --seq2 $W/b_2.fq.gz -o /dev/null -t 32 --bam
run file_single $ALL --seq1 $F1 --seq2 $F2 -o $W/o.sam -t 32
rm -f $W/o.sam
run bam_null $ALL --seq1 $F1 --seq2 $F2 -o /dev/null -t 32 --bam
run bam_file $ALL --seq1 $F1 --seq2 $F2 -o $W/o.bam -t 32 --bam
ls -la $W/o.bam; python3 - <<PY
import gzip, time
t=time.time(); n=0
with gzip.open("$W/o.bam","rb") as f:
    while True:
        b=f.read(1<<24)
        if not b: break
        n+=len(b)
        if n > (1<<30): break
print("inflated %d bytes of o.bam ok (%.1fs)" % (n, time.time()-t))
PY
rm -f $W/o.bam
run gz_bam_t64 $GZR --seq1 $W/g_1.fq.gz --seq2 $W/g_2.fq.gz -o /dev/null -t 64 --bam
BMBS_TEXT_TRACE=1 ./bitmapperbs_amd/bmbs_search --search $FA -e 0.08 --seq1 $F1 --seq2 $F2 -o /dev/null -t 32 --bam --contexts 1 2>&1 | grep "text/bam" | head -5
rm -f $W/g_1.fq.gz $W/g_2.fq.gz $W/b_1.fq.gz $W/b_2.fq.gz
