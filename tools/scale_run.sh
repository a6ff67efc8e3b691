#!/bin/bash
# tools/scale_run.sh <genome_bp> -- GRCh38-scale check on the GPU box: build the index with bmbs_index_build, bench, then a
# file-to-file run next to the reference binary (which loads the same index files).
G=${1:-3100000000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/scale_$G
mkdir -p $O
cd $R
date > $O/log.txt
( time python bench.py --genome $G --reads 10000000 --no-cpu --steps 2 > $O/bench.json 2> $O/bench.err ) 2>> $O/log.txt
tail -3 $O/bench.err >> $O/log.txt
free -g | head -2 >> $O/log.txt
( time python tools/e2e_bench.py --genome $G --reads 2000000 --io-threads 32 --ref-threads 32 --out $O/e2e.json > $O/e2e.out 2>&1 ) 2>> $O/log.txt
tail -5 $O/e2e.out >> $O/log.txt
date >> $O/log.txt
cat $O/log.txt
