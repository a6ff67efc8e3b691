#!/usr/bin/env bash
# Build the REAL reference (chhylp123/BitMapperBS v1.0.2.3) from its own sources where they lie
# under /root/reference, outputs only into oracle/_ref/ (git-ignored, but travels to the GPU box).
#
# TEST INFRASTRUCTURE ONLY.  Nothing under bitmapperbs_amd/ may call, link or execute this.
#
# Recipe, no reference build system is run, no reference source is copied:
#  * each of the 14 SOURCES of the reference Makefile (Makefile:26) is compiled straight from
#    /root/reference with the reference's own AVX2 flags (Makefile:20-24);
#  * 33 non-void functions in bwt.cpp / Schema.cpp / Process_sam_out.cpp / Process_Reads.cpp lack a
#    `return` (UB: with g++ >= 8 -O3 the binary segfaults after index load).  The sites are found
#    with `g++ -fsyntax-only -Wreturn-type` and `return 0;` is spliced in front of the closing
#    brace IN THE COMPILER'S INPUT STREAM (sed | g++ -x c++ -); no patched copy is written;
#  * the vendored htslib (needed only because bam_prase.cpp links it) is compiled from
#    /root/reference/htslib/*.c; the two files its own Makefile would `echo` into existence
#    (config.h without bz2/lzma, version.h; htslib/Makefile:116-117,205-211) are emitted into
#    oracle/_ref/inc/ by this script.
set -euo pipefail
REF=${BMBS_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then
  echo "build_ref: $REF absent (GPU box?) - keeping prebuilt oracle/_ref" >&2
  exit 0
fi
mkdir -p "$OUT/obj/cram" "$OUT/inc"
JOBS=${JOBS:-4}

# ---- htslib ------------------------------------------------------------------------------------
printf '/* default config.h as htslib/Makefile:205 would write it, minus bz2/lzma (headers absent) */\n#define HAVE_FSEEKO 1\n#define HAVE_DRAND48 1\n' > "$OUT/inc/config.h"
printf '#define HTS_VERSION "1.9"\n' > "$OUT/inc/version.h"
HTS_OBJS="kfunc knetfile kstring bcf_sr_sort bgzf errmod faidx hfile hfile_net hts hts_os md5 multipart probaln realn regidx sam synced_bcf_reader vcf_sweep tbx textutils thread_pool vcf vcfutils cram/cram_codecs cram/cram_decode cram/cram_encode cram/cram_external cram/cram_index cram/cram_io cram/cram_samtools cram/cram_stats cram/files cram/mFILE cram/open_trace_file cram/pooled_alloc cram/rANS_static cram/sam_header cram/string_alloc"
hts_one() {
  local o=$1
  [ "$OUT/obj/$o.o" -nt "$REF/htslib/$o.c" ] && return 0
  gcc -w -O2 -fPIC -I"$OUT/inc" -I"$REF/htslib" -c "$REF/htslib/$o.c" -o "$OUT/obj/$o.o"
}
export -f hts_one; export OUT REF
echo $HTS_OBJS | tr ' ' '\n' | xargs -P "$JOBS" -I{} bash -c 'hts_one {}'
rm -f "$OUT/libhts.a"
ar rcs "$OUT/libhts.a" $(for o in $HTS_OBJS; do echo "$OUT/obj/$o.o"; done)

# ---- the 14 reference sources ------------------------------------------------------------------
SRCS="saca-k bwt Bitmapper_main Process_CommandLines Auxiliary Index Schema Process_sam_out Process_Reads Ref_Genome Levenshtein_Cal SAM_queue bam_prase ksw"
CXXFLAGS="-w -mavx2 -mpopcnt -fomit-frame-pointer -O3 -D__AVX2__"
ref_one() {
  local s=$1 src="$REF/$1.cpp" obj="$OUT/obj/$1.o"
  [ "$obj" -nt "$src" ] && return 0
  # lines of the closing braces of non-void functions that fall off their end
  local lines
  lines=$( (g++ -fsyntax-only -Wreturn-type -mavx2 -mpopcnt -D__AVX2__ -I"$REF" -I"$REF/htslib" "$src" 2>&1 || true) \
            | grep -E "no return statement in function returning non-void" \
            | sed -E 's/^[^:]+:([0-9]+):.*/\1/' | sort -un)
  local sedf="$OUT/obj/$s.sed"
  : > "$sedf"
  for l in $lines; do echo "${l}s/}/return 0;}/" >> "$sedf"; done
  (cd "$REF" && sed -f "$sedf" "$src" | g++ $CXXFLAGS -x c++ - -iquote "$REF" -I"$REF" -I"$REF/htslib" -c -o "$obj")
  echo "  ref: $s ($(echo $lines | wc -w) missing-return sites spliced)"
}
export -f ref_one; export CXXFLAGS
echo $SRCS | tr ' ' '\n' | xargs -P "$JOBS" -I{} bash -c 'ref_one {}'
g++ -o "$OUT/bitmapperBS" $(for s in $SRCS; do echo "$OUT/obj/$s.o"; done) \
    "$OUT/libhts.a" -lm -lz -lpthread -Wl,--allow-multiple-definition

echo "build_ref: OK -> $OUT/bitmapperBS"
