#!/bin/bash
# tools/r4_ab.sh <outdir> <label:ENV1=v,ENV2=v ...> -- bench.py --no-cpu --no-secondary under different A/B environments (GPU box);
# prints value and the single-lane times of the seeding / vote / DP kernels.  STEPS (default 3), BARGS (extra bench arguments)
O=$1; shift
mkdir -p $O
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; python bench.py --no-cpu --no-secondary --steps ${STEPS:-3} ${BARGS:-} > $O/bench_$label.json 2> $O/bench_$label.err )
  python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); s=d.get('kernels_ms_per_launch_single_lane') or {}
ks=['k_seed_first','k_seed_decide','k_seed_second','k_seed_extra','k_vote_pe_fused','k_align_sw','k_finalize_pe','k_pe_prepare']
print(sys.argv[2], 'value', d['value'], 'single-lane value', (d['roofline'].get('single_lane') or {}).get('value_M_reads_s'), ' '.join('%s=%.3f' % (k, s.get(k, 0)) for k in ks), 'sum=%.2f' % sum(v for k, v in s.items() if k.startswith('k_')))" $O/bench_$label.json $label
done
