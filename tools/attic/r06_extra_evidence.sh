cd $GRAFT_REPO_ROOT
timeout 1500 python tools/fuzz_e2e.py --big --trials 240 --seed 6641 > gpurun_out/r06_final_fuzz_e2e_big_240.txt 2>&1; tail -n 1 gpurun_out/r06_final_fuzz_e2e_big_240.txt
timeout 600 python tools/fuzz_parity.py --trials 300 --seed 6643 > gpurun_out/r06_final_fuzz_parity_300.txt 2>&1; tail -n 1 gpurun_out/r06_final_fuzz_parity_300.txt
for c in 1 3 4; do python bench.py --config $c --no-cpu --no-secondary > gpurun_out/r06_c${c}_bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_c${c}_bench.json').read().strip().splitlines()[-1]); print('config $c', d['value'], d['metric'], d['library']['build_id'])"; done
