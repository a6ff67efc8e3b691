run() { echo "== $1"; env $1 python bench.py --no-cpu --no-secondary --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_launch']
print(d['value'], {x:k[x] for x in ('k_seed_first','k_seed_decide','k_seed_second','k_seed_extra','k_align_sw','k_vote_pe_fused') if x in k})"; }
run A=1
run BMBS_EXTRA_NOLDS=1
run BMBS_DECIDE=vec8
run BMBS_SEED_WAVES=16384
run BMBS_SEED_WAVES=262144
