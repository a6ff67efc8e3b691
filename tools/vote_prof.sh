#!/bin/bash
# tools/vote_prof.sh -- here: builds bitmapperbs_amd/libbmbs_hip_voteprof.so (the mapping translation unit with -DVOTE_PROF: phase cycle
# counters in k_vote_long, printed by bmbs_destroy).  On the GPU box:
#   BMBS_LANES=1 BMBS_LIB=$PWD/bitmapperbs_amd/libbmbs_hip_voteprof.so python3 bench.py --grch38-like --se --launches 1 --steps 2 --no-cpu --no-secondary --no-single-lane
cd "$(dirname "$0")/../bitmapperbs_amd/csrc" || exit 1
make -s ../libbmbs_hip.so || exit 1
ID=$(cat $(make -s -p -n 2>/dev/null | sed -n "s/^LIB_SRCS := //p") 2>/dev/null | sha256sum | cut -c1-16); g++ -O2 -fPIC -DBMBS_BUILD_ID="\"${ID:-unknown}+voteprof\"" -c -o build/build_id_voteprof.o build_id.cpp && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DVOTE_PROF -c -o build/bmbs_api_voteprof.o bmbs_api.hip && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o ../libbmbs_hip_voteprof.so build/bmbs_api_voteprof.o build/bmbs_textpath.o build/index_build_gpu.o build/index_io.o build/build_id_voteprof.o -lpthread
