#!/usr/bin/env bash
# oracle/_ref/bitmapperBS_hip: the REFERENCE program (its own main, command line, Load_Index, FASTQ reader, SAM / BAM writers --
# the objects oracle/build_ref.sh compiled from /root/reference) with its per-read mapping loops replaced by batch calls into
# bitmapperbs_amd/libbmbs_hip.so through oracle/bind_check.cpp -- the binding INTEGRATION.md sections 2-4 describe.
#
# TEST INFRASTRUCTURE ONLY.  Outputs only under oracle/_ref/ (git-ignored, travels to the GPU box).  tests/test_binding.py diffs the
# SAM this binary writes against the committed goldens (-m gpu).
#
# How the loops are replaced without touching a reference source: the six entry points main calls (Map_Single_Seq[_pbat][_muti_thread],
# Map_Pair_Seq[_muti_thread]; Schema.h:375-388) are made WEAK in a copy of the reference's Schema.o (objcopy --weaken-symbol), so the
# strong definitions in bind_check.o win at link time; every other symbol of Schema.o -- Prepare_alignment, the emitters, the
# statistics -- is used as compiled.
set -euo pipefail
REF=${BMBS_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
LIBDIR="$HERE/../bitmapperbs_amd"
if [ ! -d "$REF" ]; then
  echo "build_ref_hip: $REF absent (GPU box?) - keeping prebuilt oracle/_ref/bitmapperBS_hip" >&2
  exit 0
fi
[ -f "$OUT/obj/Schema.o" ] || "$HERE/build_ref.sh"
[ -f "$LIBDIR/libbmbs_hip.so" ] || { echo "build_ref_hip: build bitmapperbs_amd/libbmbs_hip.so first (make -C bitmapperbs_amd/csrc)" >&2; exit 1; }
mkdir -p "$OUT/obj_hip"
WEAK=""
SYMS="$(nm "$OUT/obj/Schema.o")"
for s in _Z14Map_Single_Seqi _Z26Map_Single_Seq_muti_threadi _Z19Map_Single_Seq_pbati _Z31Map_Single_Seq_pbat_muti_threadi \
         _Z12Map_Pair_Seqi _Z24Map_Pair_Seq_muti_threadi; do
  grep -q " T $s\$" <<<"$SYMS" || { echo "build_ref_hip: $s is not a defined function of the reference's Schema.o" >&2; exit 1; }
  WEAK="$WEAK --weaken-symbol=$s"
done
objcopy $WEAK "$OUT/obj/Schema.o" "$OUT/obj_hip/Schema.o"
g++ -w -O2 -mavx2 -mpopcnt -D__AVX2__ -iquote "$REF" -I"$REF" -I"$REF/htslib" -I"$HERE/../include" \
    -c "$HERE/bind_check.cpp" -o "$OUT/obj_hip/bind_check.o"
SRCS="saca-k bwt Bitmapper_main Process_CommandLines Auxiliary Index Process_sam_out Process_Reads Ref_Genome Levenshtein_Cal SAM_queue bam_prase ksw"
g++ -o "$OUT/bitmapperBS_hip" "$OUT/obj_hip/bind_check.o" "$OUT/obj_hip/Schema.o" $(for s in $SRCS; do echo "$OUT/obj/$s.o"; done) \
    "$OUT/libhts.a" -L"$LIBDIR" -lbmbs_hip -lm -lz -lpthread -Wl,--allow-multiple-definition \
    -Wl,-rpath,'$ORIGIN/../../bitmapperbs_amd'
# the strong definitions must be the binding's
LINKED="$(nm "$OUT/bitmapperBS_hip")"
for s in _Z14Map_Single_Seqi _Z12Map_Pair_Seqi; do
  grep -q " T $s\$" <<<"$LINKED" || { echo "build_ref_hip: $s did not resolve to the binding" >&2; exit 1; }
done
echo "build_ref_hip: OK -> $OUT/bitmapperBS_hip"
