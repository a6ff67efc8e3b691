#!/bin/bash
# tools/r3_ab.sh <outdir> <label=ENV1=v,ENV2=v ...> -- bench.py --no-cpu --no-secondary under different A/B environments (GPU box)
O=$1; shift
mkdir -p $O
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; python bench.py --no-cpu --no-secondary --steps ${STEPS:-5} ${BARGS:-} > $O/bench_$label.json 2> $O/bench_$label.err )
  python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])" $O/bench_$label.json $label
done
