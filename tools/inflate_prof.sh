#!/bin/bash
# tools/inflate_prof.sh -- here: builds bitmapperbs_amd/libbmbs_hip_prof.so (the library with -DINF_PROFILE: phase cycle counters in
# k_bgzf_inflate).  On the GPU box: python3 tools/inflate_bench.py 160 bitmapperbs_amd/libbmbs_hip_prof.so
cd "$(dirname "$0")/../bitmapperbs_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -DINF_PROFILE -DBMBS_BUILD_ID='"profile"' -shared \
    -o ../libbmbs_hip_prof.so bmbs_api.hip index_build_gpu.hip index_io.cpp -lpthread
