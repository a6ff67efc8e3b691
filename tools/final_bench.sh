set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
python bench.py --grch38-like --no-cpu --no-secondary > gpurun_out/r06_final_grch38_like_pe.json 2>> gpurun_out/r06_final_bench.err
python bench.py --grch38-like --se --no-cpu --no-secondary > gpurun_out/r06_final_grch38_like_se.json 2>> gpurun_out/r06_final_bench.err
python bench.py --grch38-like --sensitive --units 5000000 --no-cpu --no-secondary > gpurun_out/r06_final_grch38_like_sensitive.json 2>> gpurun_out/r06_final_bench.err
tail -c 600 gpurun_out/r06_final_bench.json
