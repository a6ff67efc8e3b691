#!/bin/bash
# tools/bgzf_prof.sh -- here: builds bitmapperbs_amd/libbmbs_hip_bzprof.so (the text-path translation unit with -DBGZF_PROFILE: phase cycle
# counters in k_bgzf_block).  On the GPU box: BMBS_LIB=$PWD/bitmapperbs_amd/libbmbs_hip_bzprof.so python3 tools/text_bench.py
cd "$(dirname "$0")/../bitmapperbs_amd/csrc" || exit 1
make -s ../libbmbs_hip.so || exit 1
ID=$(cat $(make -s -p -n 2>/dev/null | sed -n "s/^LIB_SRCS := //p") 2>/dev/null | sha256sum | cut -c1-16); g++ -O2 -fPIC -DBMBS_BUILD_ID="\"${ID:-unknown}+bzprof\"" -c -o build/build_id_bzprof.o build_id.cpp && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DBGZF_PROFILE -c -o build/bmbs_textpath_bzprof.o bmbs_textpath.hip && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o ../libbmbs_hip_bzprof.so build/bmbs_api.o build/bmbs_textpath_bzprof.o build/index_build_gpu.o build/index_io.o build/build_id_bzprof.o -lpthread
