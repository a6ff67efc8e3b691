#!/usr/bin/env bash
# Control experiment for oracle/build_ref.sh: the four reference sources that receive `return 0;` splices there (bwt, Schema,
# Process_sam_out, Process_Reads: 33 non-void functions fall off their end) compiled here WITHOUT any splice at -O0, where
# g++ does not exploit the undefined behaviour, and linked with the ten untouched objects of the regular build ->
# oracle/_ref/bitmapperBS_unpatched_O0.  tests/test_oracle.py asserts that this binary writes the same SAM as the committed
# goldens: the splices are behaviour-neutral.  TEST INFRASTRUCTURE ONLY.
set -euo pipefail
REF=${BMBS_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
[ -d "$REF" ] || { echo "build_ref_unpatched: $REF absent" >&2; exit 0; }
[ -x "$OUT/bitmapperBS" ] || "$HERE/build_ref.sh"
if [ -x "$OUT/bitmapperBS_unpatched_O0" ] && [ "$OUT/bitmapperBS_unpatched_O0" -nt "$REF/Schema.cpp" ]; then exit 0; fi
mkdir -p "$OUT/obj_O0"
one() { g++ -w -mavx2 -mpopcnt -fomit-frame-pointer -O0 -D__AVX2__ -iquote "$REF" -I"$REF" -I"$REF/htslib" -c "$REF/$1.cpp" -o "$OUT/obj_O0/$1.o"; }
export -f one; export REF OUT
echo bwt Schema Process_sam_out Process_Reads | tr ' ' '\n' | xargs -P "${JOBS:-4}" -I{} bash -c 'one {}'
OBJS=""
for s in saca-k Bitmapper_main Process_CommandLines Auxiliary Index Ref_Genome Levenshtein_Cal SAM_queue bam_prase ksw; do OBJS="$OBJS $OUT/obj/$s.o"; done
for s in bwt Schema Process_sam_out Process_Reads; do OBJS="$OBJS $OUT/obj_O0/$s.o"; done
g++ -o "$OUT/bitmapperBS_unpatched_O0" $OBJS "$OUT/libhts.a" -lm -lz -lpthread -Wl,--allow-multiple-definition
echo "build_ref_unpatched: OK -> $OUT/bitmapperBS_unpatched_O0"
