#!/usr/bin/env bash
# Build the reference's external suffix sorter `psascan` (vendored pSAscan-0.1.0 + libdivsufsort-2.0.1) from the sources where
# they lie under /root/reference, outputs only into oracle/_ref/ (git-ignored).  TEST INFRASTRUCTURE ONLY: it exists so that
# tests/golden/make_golden.py can let the REFERENCE's own `--index` (bwt.cpp:1031-1041 shells out to ./psascan) write the index
# whose sha256 pins our builders (tests/golden/index_ref_sha256.json).  Recipe = SURVEY.md §8c step 4, out of source:
#   libdivsufsort: cmake -DBUILD_DIVSUFSORT64=ON -DBUILD_SHARED_LIBS=OFF (generates divsufsort.h / divsufsort64.h), make
#   pSAscan:       the one g++ line of pSAscan-0.1.0/src/Makefile against those headers / libraries
set -euo pipefail
REF=${BMBS_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/pSAscan-0.1.0" ]; then
  echo "build_psascan: $REF/pSAscan-0.1.0 absent - nothing to do" >&2
  exit 0
fi
if [ -x "$OUT/psascan" ] && [ "$OUT/psascan" -nt "$REF/pSAscan-0.1.0/src/main.cpp" ]; then
  echo "build_psascan: up to date -> $OUT/psascan"; exit 0
fi
B="$OUT/divsufsort"
mkdir -p "$B"
cmake -S "$REF/libdivsufsort-2.0.1" -B "$B" -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -DCMAKE_BUILD_TYPE=Release \
      -DBUILD_DIVSUFSORT64=ON -DBUILD_SHARED_LIBS=OFF -DBUILD_EXAMPLES=OFF > "$B/cmake.log" 2>&1
make -C "$B" -j"${JOBS:-4}" > "$B/make.log" 2>&1
SRC="$REF/pSAscan-0.1.0/src"
g++ -w -funroll-loops -pthread -std=c++0x -DNDEBUG -O3 -o "$OUT/psascan" "$SRC/psascan_src/utils.cpp" "$SRC/main.cpp" \
    -I "$B/include" -L "$B/lib" -ldivsufsort -ldivsufsort64 -fopenmp
echo "build_psascan: OK -> $OUT/psascan"
