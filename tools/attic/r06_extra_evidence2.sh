cd $GRAFT_REPO_ROOT
timeout 900 python tools/fuzz_e2e.py --bound --trials 120 --seed 6651 > gpurun_out/r06_final_fuzz_bound.txt 2>&1; tail -n 1 gpurun_out/r06_final_fuzz_bound.txt
timeout 900 python tools/fuzz_e2e.py --trials 200 --seed 6652 > gpurun_out/r06_final_fuzz_e2e_200.txt 2>&1; tail -n 1 gpurun_out/r06_final_fuzz_e2e_200.txt
