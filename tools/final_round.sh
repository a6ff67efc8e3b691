#!/bin/bash
# tools/final_round.sh [tag prefix, default r06_final] -- GPU box: the round's closing evidence on ONE build: the GPU test suite, the
# soaks against the oracle and the real reference binary (small and repeat-rich genome, the bound program), then the default bench line.
P=${1:-r06_final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${P}_gpu_tests.txt 2>&1; tail -n 2 gpurun_out/${P}_gpu_tests.txt
timeout 1200 python tools/fuzz_e2e.py --big --trials 120 --seed 6631 > gpurun_out/${P}_fuzz_e2e_big.txt 2>&1; tail -n 1 gpurun_out/${P}_fuzz_e2e_big.txt
timeout 900 python tools/fuzz_parity.py --trials 200 --seed 6633 > gpurun_out/${P}_fuzz_parity.txt 2>&1; tail -n 1 gpurun_out/${P}_fuzz_parity.txt
timeout 900 python tools/fuzz_e2e.py --trials 80 --seed 6632 > gpurun_out/${P}_fuzz_e2e.txt 2>&1; tail -n 1 gpurun_out/${P}_fuzz_e2e.txt
timeout 900 python tools/fuzz_e2e.py --bound --big --trials 60 --seed 6634 > gpurun_out/${P}_fuzz_bound_big.txt 2>&1; tail -n 1 gpurun_out/${P}_fuzz_bound_big.txt
python bench.py > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err; tail -c 300 gpurun_out/${P}_bench.json
