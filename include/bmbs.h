/* include/bmbs.h -- C-ABI of the MI355X-native BitMapperBS mapping hot path (libbmbs_hip.so).
 *
 * BitMapperBS has no plugin/FFI layer: this is the boundary *cut* at L4 -> {L5,L6,L7}.
 * A host driver that keeps BitMapperBS's CLI, FASTQ reader and SAM writer calls these entry points
 * where the reference calls (per read, inline) the routines cited at each declaration; paths are
 * relative to the reference tree.  Conventions kept from the reference side of the cut:
 *   - index arrays are read-only after attach (Load_Index, Index.cpp:940);
 *   - errors are status codes / sentinels, never exceptions: err = 0xFFFFFFFF, end_site = -1
 *     (Levenshtein_Cal.h:354,512), calls return 0 or a negative BMBS_E*;
 *   - the caller owns host buffers, the library owns device buffers; one bmbs_ctx per GPU;
 *     calls on one ctx are serialised by the caller, different ctxs are fully concurrent.
 * Plain C types only (no torch / HIP types in any signature).
 */
#ifndef BMBS_H
#define BMBS_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMBS_OK            0
/* the longest read the path takes.  The reference sizes its per-read buffers for 1000 characters (Auxiliary.h:14) and has no room for
 * their terminators at 999 and 1000: it prints 222 of 300 records at those lengths, one with bytes outside ASCII
 * (tests/golden/make_golden.py) -- there is nothing to be identical to, so reads of 999 and 1000 bases are refused (BMBS_EINVAL)
 * rather than mapped unpinned.  998 is pinned by goldens (se_g998, pe_p998).                                                    */
#define BMBS_MAX_READ    998
#define BMBS_EINVAL      (-22)
#define BMBS_ENOMEM      (-12)
#define BMBS_ENODEV      (-19)   /* no HIP device / runtime error: the product never falls back to CPU */
#define BMBS_ESTATE       (-1)   /* index not attached, batch too large, ... */

typedef struct bmbs_ctx bmbs_ctx;

/* mapping parameters == the reference's option globals (Process_CommandLines.cpp:40-75,88-132) */
typedef struct bmbs_params {
    double  e_f;        /* -e, thread_e_f: k = (uint64)(e_f * L) capped at 31 (Schema.cpp:24546)   */
    int32_t mp_max;     /* --mp_max 6  */
    int32_t mp_min;     /* --mp_min 2  */
    int32_t np;         /* --np 1      */
    int32_t gap_open;   /* --gap_open 5 */
    int32_t gap_ext;    /* --gap_extension 3 */
    int32_t q_base;     /* 33 | 64     */
    int32_t seed_len;   /* --seed (over_all_seed_length) 30 */
    int32_t min_ins;    /* --min 0     */
    int32_t max_ins;    /* --max 500   */
    int32_t sensitive;  /* --sensitive */
    int32_t ambiguous_out; /* --ambiguous_out: one hit of every ambiguous read / pair is aligned and returned with status
                            * BMBS_ST_AMBIG (Schema.cpp:24639, 25095, 19345); 0 = ambiguous reads carry no alignment.
                            * --unmapped_out and --pbat need nothing from the device: the former prints the records whose
                            * status is BMBS_ST_UNMAPPED / BMBS_ST_OFFEND, the latter maps the reverse complement of each
                            * read with mirrored qualities (Process_Reads.cpp:986-1075; see bmbs_search.cpp) */
} bmbs_params;

void bmbs_default_params(bmbs_params* p);

/* host view of the reference's loaded index: global `bitmapper_index_params` (bwt.h:34-163, filled by
 * load_index, bwt.cpp:2563-2643), the 2-bit genome `_ih_refGen` and the chromosome table
 * (Load_Index, Index.cpp:940-1045).  All pointers are host memory in the on-disk layouts. */
typedef struct bmbs_index_view {
    uint64_t        ref_len;            /* refGenLength (one strand) */
    const uint8_t*  pac;                /* .bs.pac payload, 4 bases/byte MSB first */
    uint64_t        pac_bytes;
    uint64_t        sa_length;          /* rows = 2*ref_len + 1 */
    uint64_t        shapline;
    uint64_t        nacgt[5];
    const uint64_t* bwt;                /* 5 x u64 per 128 rows */
    uint64_t        bwt_words;
    const uint64_t* high_occ;           /* 2 x u64 per 65536 rows */
    uint64_t        high_occ_words;
    const uint32_t* hash_hi;            /* 16-mer table, 3^16+1 entries */
    const uint8_t*  hash_lo;
    uint64_t        hash_entries;
    const uint32_t* sa;                 /* sampled SA (every 8th text position) */
    uint64_t        sa_entries;
    const uint64_t* sa_flag;            /* 5 x u64 per 256 rows */
    uint64_t        sa_flag_words;
    int32_t         n_chrom;
    const uint64_t* chrom_len;          /* _rg_chrome_length[n_chrom] */
} bmbs_index_view;

/* one mapped read; 32 bytes.  What output_sam_end_to_end (Schema.cpp:11928) is handed, after the
 * chromosome lookup and the off-end check it performs (11962-11986). */
typedef struct bmbs_result {
    uint64_t pos;          /* 1-based position on `chrom`                                    */
    uint32_t cigar_off;    /* first op in the cigar pool (only when n_cigar > 0; else "<L>M") */
    int32_t  chrom;        /* chromosome id (32 bits: scaffold-level assemblies have > 32 767 sequences) */
    uint16_t flag;         /* 0 | 16 (SE); 99/147/83/163 (PE)                                */
    uint16_t nm;           /* NM:i                                                           */
    int16_t  score;        /* alignment score (<= 0)                                         */
    uint8_t  status;       /* BMBS_ST_*                                                      */
    uint8_t  mapq;
    uint8_t  n_cigar;      /* ops in the pool; 0 means the single op <L>M                    */
    uint8_t  path;         /* 1 exact-unique exit, 2 one-mismatch exit, 3 general, 4 exact-ambiguous */
    uint16_t n_cand;       /* candidate sites located for this read (diagnostic, saturates at 65 535) */
    uint32_t tlen;         /* paired end: |TLEN| of the pair; 0 for single-end records       */
} bmbs_result;

#define BMBS_ST_UNMAPPED  0
#define BMBS_ST_UNIQUE    1   /* a SAM record is emitted                                       */
#define BMBS_ST_AMBIG     2   /* counted as ambiguous, nothing emitted (no --ambiguous_out)     */
#define BMBS_ST_OFFEND    3   /* alignment crosses a chromosome end: rejected at emit           */

/* cigar op = len << 4 | op, op 0 M, 1 D, 2 I, already in SAM (left-to-right on the forward strand)
 * order (ksw.cpp:2785-2857 prints them forward or reversed by strand) */

/* ---- launch sequence ----------------------------------------------------------------------------
 * A context owns BMBS_LANES (environment, default 3) lanes: a stream, work buffers and counters each, on one attached index.  A
 * mapping call of 500 000 units and more is cut into one chunk per lane, so that the issue-bound kernels of one chunk (DP, Myers,
 * row preparation) and the list kernels, which wait on their own chains, run beside the memory-bound seeding kernels of another; the host-pointer calls also overlap the copies of one
 * chunk with the kernels of another.  After the first call of a context no call waits for its stage counts: buffers and grids
 * are sized from what earlier calls needed per read (+25 %), guard kernels compare the real counts on the device, and a call that
 * did not fit is issued again with exact sizes when the context is next synchronised -- results and statistics are the same either
 * way.  BMBS_EXACT=1 makes every call wait for its counts (the round-2 sequence), BMBS_LANES=1 turns the split off.            */

/* ---- lifecycle -------------------------------------------------------------------------------- */
/* replaces Prepare_alignment (Schema.cpp:639: LUTs, score matrices) for one GPU                 */
bmbs_ctx* bmbs_create(int device_id, const bmbs_params* params);
void      bmbs_destroy(bmbs_ctx*);
const char* bmbs_last_error(const bmbs_ctx*);
/* replaces the in-memory result of Load_Index + load_index: uploads once, re-packs for HBM        */
int bmbs_index_attach(bmbs_ctx*, const bmbs_index_view*);
/* takes device memory for the work buffers of later calls now (bytes per lane; about 2.5 KB per read of the largest batch for the
 * text calls): optional, saves the first calls the device-wide allocation                                           */
int bmbs_reserve(bmbs_ctx*, uint64_t bytes_per_lane);
/* a further context on the index `owner` has attached (same device): shares the index in HBM, owns its stream and work buffers.
 * For keeping two batches in flight from two host threads; `owner` has to outlive it.                                  */
int bmbs_index_share(bmbs_ctx*, const bmbs_ctx* owner);

/* ---- stage entry points (host buffers in, host buffers out; used by the parity tests) ---------- */
/* reads: n rows of `stride` bytes ASCII upper-case (as produced by inputReads_single_directly,
 * Process_Reads.cpp:810), all of length L. */

/* K7 alone: get_actuall_genome / get_actuall_rc_genome (Schema.cpp:4998-5115): the `len` bases (len <= 1024) of the doubled
 * genome that start at doubled coordinate site[i], as ASCII into out + i*len; an out-of-strand request gives the reference's
 * all-zero window (bytes 0).  Used by the parity tests to check the HBM genome against the index files.                 */
int bmbs_window_batch(bmbs_ctx*, const uint64_t* site, int64_t n_sites, int32_t len, char* out);

/* K5 alone: locate_one_position_direct (bwt.h:2585) without the seed adjustment: the text position SA[row] of each
 * suffix-array row (0 <= row <= 2*ref_len).                                                                            */
int bmbs_locate_batch(bmbs_ctx*, const uint64_t* row, int64_t n_rows, uint64_t* pos);

/* a9 alone: the order in which `std::sort(votes, votes + n, compare_seed_votes)` (Schema.cpp:560-563, 24986 -- an unstable sort:
 * libstdc++'s introsort) visits vote lists.  List s is vote[seg_off[s] .. seg_off[s+1]) (counts 1..255); perm[seg_off[s] + j] =
 * index within list s of the entry visited j-th.  form 0: the one-wave kernel (lists of up to 256 entries), form 1: the
 * one-block kernel (up to 4096) -- the two forms k_vote_long runs on the candidate lists of repeat reads.               */
int bmbs_vote_order_batch(bmbs_ctx*, const uint8_t* vote, const int64_t* seg_off, int64_t n_seg, int32_t form, uint32_t* perm);

/* K7+K8: get_actuall_[rc_]genome + BS_Reserve_Banded_BPM{,_4_SSE,_8_SSE}
 * (Schema.cpp:4998-5115; Levenshtein_Cal.h:351,1678,2093): candidate i = (read_of[i], site[i]).   */
int bmbs_filter_batch(bmbs_ctx*, const char* seq, int32_t L, int32_t stride, int64_t n_reads,
                      const uint32_t* read_of, const uint64_t* site, int64_t n_cand,
                      uint32_t* err, int32_t* end_site);

/* the same call on the form the mapping calls run: the rows are packed on the device first (2 bits per base + a not-ACGT bit plane, what
 * k_filter / k_filter_pe read since round 4) and the Myers rows take their characters from the packed words.  Same results as
 * bmbs_filter_batch for every input, characters outside ACGT included (tests/test_gpu_parity.py).                                  */
int bmbs_filter_batch_packed(bmbs_ctx*, const char* seq, int32_t L, int32_t stride, int64_t n_reads,
                             const uint32_t* read_of, const uint64_t* site, int64_t n_cand,
                             uint32_t* err, int32_t* end_site);

/* K11-K13: fast_recalculate_bs_Cigar (ksw.cpp:2578) for job i = (read_of[i], site[i], end_site[i],
 * err[i]); cigar ops: max_ops per job, SAM order.                                                  */
int bmbs_align_batch(bmbs_ctx*, const char* seq, const char* qual, int32_t L, int32_t stride,
                     int64_t n_reads, const uint32_t* read_of, const uint64_t* site,
                     const int32_t* end_site_in, const uint32_t* err_in, int64_t n_jobs,
                     int32_t* start_site, int32_t* end_site, uint32_t* nm, int32_t* score,
                     uint32_t* cigar_ops, int32_t* n_ops, int32_t max_ops);

/* K1-K6 (+a8-a10): the seeding state machine of Map_Single_Seq_end_to_end (Schema.cpp:24588-24986).
 * verdict[i]: 0 no candidate, 1 exact-unique exit (exit_site), 2 one-mismatch exit (exit_site),
 * 3 general path, 4 exact but ambiguous.  For verdict 3 the read's votes, in the reference's visiting
 * order (std::sort by vote, Schema.cpp:24986), are vote_site/vote_cnt[seg_off[i] .. seg_off[i]+n_votes[i]). */
int bmbs_seed_batch(bmbs_ctx*, const char* seq, int32_t L, int32_t stride, int64_t n_reads,
                    uint8_t* verdict, uint64_t* exit_site, uint64_t* seg_off, uint32_t* n_votes,
                    uint64_t* vote_site, uint32_t* vote_cnt, int64_t vote_cap, int64_t* total_slots);

/* Slots per read (per mate) the cigar pool of the mapping calls needs for reads of length L under these parameters (NULL: the
 * defaults): 2k + 8 with the default penalties, more when --gap_open / --gap_extension (Process_CommandLines.cpp:129-130) make
 * gaps cheaper than mismatches; -1 for L outside 1..BMBS_MAX_READ.  A pool of n_reads * bmbs_max_cigar_ops(params, L_max) entries (twice
 * that for pairs) is always enough; parameter sets that allow more than 254 operations per alignment are refused by the mapping
 * calls (BMBS_EINVAL): n_cigar of a record is 8 bits. */
int32_t bmbs_max_cigar_ops(const bmbs_params* params, int32_t L);

/* ---- fused single-end mapping (Map_Single_Seq_end_to_end loop body, Schema.cpp:24488-25119) ---- */
/* host buffers: copies in, maps, copies results out.  cigar_pool[cigar_cap] receives the ops.      */
int bmbs_map_se(bmbs_ctx*, const char* seq, const char* qual, int32_t L, int32_t stride,
                int64_t n_reads, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap,
                int64_t* n_cigar_used);
/* device-resident variant: d_seq/d_qual/d_results/d_cigar_pool are device addresses (e.g. from
 * torch tensors); nothing crosses PCIe; asynchronous on the ctx stream until bmbs_sync(): the input
 * buffers are read throughout the call (also by the paired-end variants) and have to stay untouched until then. */
int bmbs_map_se_device(bmbs_ctx*, uint64_t d_seq, uint64_t d_qual, int32_t L, int32_t stride,
                       int64_t n_reads, uint64_t d_results, uint64_t d_cigar_pool, int64_t cigar_cap);
int bmbs_sync(bmbs_ctx*);

/* ---- reads of different lengths in ONE batch (reads_soa.len of SURVEY.md section 8b; a trimmed library).  The reference maps
 * read by read and derives everything from current_read.length (error_threshold1 = thread_e_f * length, Schema.cpp:24546;
 * seed count L/10-1; p_length = L + 2k; MAP_Calculation over that read's threshold).  len[i] in [1, L_max], rows are still
 * `stride` bytes apart (stride >= L_max), bytes past len[i] are ignored.  The CIGAR pool reserves 2*k(L_max)+8 ops per read.
 * Paired-end: mates of one pair may have different lengths (the insert window uses the larger threshold and the longer
 * mate, Schema.cpp:18900-18935).  d_len of the device form = u16[2n]: the n first mates, then the n second mates.     */
int bmbs_map_se_var(bmbs_ctx*, const char* seq, const char* qual, const uint16_t* len, int32_t L_max, int32_t stride,
                    int64_t n_reads, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used);
int bmbs_map_se_var_device(bmbs_ctx*, uint64_t d_seq, uint64_t d_qual, uint64_t d_len, int32_t L_max, int32_t stride,
                           int64_t n_reads, uint64_t d_results, uint64_t d_cigar_pool, int64_t cigar_cap);
int bmbs_map_pe_var(bmbs_ctx*, const char* seq1, const char* qual1, const char* seq2, const char* qual2,
                    const uint16_t* len1, const uint16_t* len2, int32_t L_max, int32_t stride, int64_t n_pairs,
                    bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used);
int bmbs_map_pe_var_device(bmbs_ctx*, uint64_t d_seq1, uint64_t d_qual1, uint64_t d_seq2, uint64_t d_qual2, uint64_t d_len,
                           int32_t L_max, int32_t stride, int64_t n_pairs, uint64_t d_results, uint64_t d_cigar_pool,
                           int64_t cigar_cap);

/* ---- packed reads (round 5): the same calls with 2 bits per base going over the link instead of a byte ---------------------------------
 * What the reference's reader hands a mapping thread is the upper-cased sequence line (Process_Reads.cpp:810-890; mate 2 reverse-
 * complemented, :262-267); the host-buffer entry points above move that as it is, 160 bytes of ASCII per 150-base read and mate, and are
 * bound by the link (SURVEY 8d counts H2D / D2H).  Here the caller packs the sequence once per batch (bmbs_pack_rows, a few GB/s on the
 * host's threads, or its own reader writes the format directly):
 *   row = W = ceil(L_max / 32) u64 words of bases -- base j in bits 2 (j % 32), 2 (j % 32) + 1 of word j / 32: A 0, C 1, G 2, T 3 --
 *         followed by M = ceil(L_max / 64) words of marks -- bit j % 64 of word j / 64 set: the character at j is 'N' (its base bits 0);
 *         bits at and beyond a read's length are ignored; rows `pwords` (>= W + M) words apart.  150 bases: 8 words = 64 bytes.
 * Characters other than A C G T N cannot be expressed (bmbs_pack_rows says BMBS_EINVAL and which row; the same for a length of 0 or
 * beyond L_max): such batches take the ASCII calls.  BOTH mates are given in FASTQ orientation (what bmbs_map_pe takes as seq2): the device reverse-complements mate 2 on the
 * packed words.  Qualities stay bytes (`stride` apart, as above); len NULL: every read has length L_max.  Results, CIGAR pool, errors:
 * exactly those of bmbs_map_se[_var] / bmbs_map_pe[_var] on the ASCII rows the packed ones stand for (tests/test_gpu_parity.py).     */
int bmbs_pack_rows(const char* seq, int32_t L_max, int32_t stride, int64_t n, const uint16_t* len /* NULL: uniform */, uint64_t* rows, int32_t pwords,
                   int32_t threads, int64_t* bad_row /* may be NULL */);
int bmbs_map_se_packed(bmbs_ctx*, const uint64_t* rows, int32_t pwords, const char* qual, const uint16_t* len, int32_t L_max, int32_t stride,
                       int64_t n_reads, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used);
int bmbs_map_pe_packed(bmbs_ctx*, const uint64_t* rows1, const uint64_t* rows2, int32_t pwords, const char* qual1, const char* qual2,
                       const uint16_t* len1, const uint16_t* len2, int32_t L_max, int32_t stride, int64_t n_pairs, bmbs_result* results,
                       uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used);

/* ---- fused paired-end mapping, default (fast) mode: Map_Pair_Seq_end_to_end_fast (Schema.cpp:18570-19546)
 * = get_candidates x2 (18172), filter_pairs (16052), verify_candidate_locations (18130) on the smaller side,
 * filter_pairs_single_side (16186), new_faster_verify_pairs (15773), calculate_best_map_cigar_end_to_end_return
 * (14602), TLEN / insert / chromosome-end checks and MAPQ over k1+k2 (19400-19440).
 * seq2/qual2 = mate 2 exactly as in the FASTQ file (the library builds the reverse complement the reference's
 * reader builds, Process_Reads.cpp:262-267).  Both mates have length L.  results[2*i], results[2*i+1] = mate 1,
 * mate 2 of pair i: flag 99/83 and 147/163, `tlen` = |TLEN|, status BMBS_ST_* for the PAIR
 * (BMBS_ST_OFFEND also covers the insert-size rejection).  Stats count pairs (Schema.cpp:19531-19537).
 * With bmbs_params.sensitive = 1 the same entry points run --sensitive: Map_Pair_Seq_end_to_end (Schema.cpp:19953-21459)
 * = first seeds of both mates, process_rest_seed_debug (17574) on the mate with fewer first-seed candidates,
 * process_rest_seed_filter_debug (16298) on the other, reseed_filter / select_best_seeds (16678 / 16630) for a mate
 * left without a hit, then the same pairing and post-processing.                                          */
int bmbs_map_pe(bmbs_ctx*, const char* seq1, const char* qual1, const char* seq2, const char* qual2, int32_t L,
                int32_t stride, int64_t n_pairs, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap,
                int64_t* n_cigar_used);
int bmbs_map_pe_device(bmbs_ctx*, uint64_t d_seq1, uint64_t d_qual1, uint64_t d_seq2, uint64_t d_qual2, int32_t L,
                       int32_t stride, int64_t n_pairs, uint64_t d_results, uint64_t d_cigar_pool, int64_t cigar_cap);

/* ---- FASTQ text in: the reads are cut out of the text ON THE DEVICE -------------------------------------------------------------
 * What the reference's reader does per record on one host thread (inputReads_single_directly / inputReads_paired_directly,
 * Process_Reads.cpp:810-890, 155-317: kseq line splitting, toupper, `qual.resize(seq.size(), ' ')`, the reverse complement of
 * mate 2 at :262-267 and of every --pbat read with mirrored qualities at :986-1075) is done by one kernel over the batch; the host
 * hands over the text window as it came from the file plus the line starts it found.  A window is smaller than 4 GiB.
 * Records keep their own lengths (1..L_max, L_max <= BMBS_MAX_READ); `uniform` != 0 promises that every read has length L_max (the
 * fixed-length kernels are used).  results / cigar_pool as for bmbs_map_se / bmbs_map_pe.  Page-locked text buffers
 * (bmbs_host_alloc) are copied at link speed.                                                                                  */
typedef struct bmbs_fastq_view {
    const char*     text;        /* FASTQ text window (host memory)                                                   */
    uint64_t        text_bytes;
    const uint32_t* seq_off;     /* [n] offset of the first base of record i                                          */
    const uint32_t* qual_off;    /* [n] offset of its first quality character                                         */
    const uint16_t* seq_len;     /* [n] bases of record i                                                             */
    const uint16_t* qual_len;    /* [n] quality characters present (a shorter line is padded with ' ')                */
} bmbs_fastq_view;
int bmbs_map_se_fastq(bmbs_ctx*, const bmbs_fastq_view* reads, int64_t n_reads, int32_t L_max, int32_t uniform, int32_t pbat,
                      bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used);
int bmbs_map_pe_fastq(bmbs_ctx*, const bmbs_fastq_view* mate1, const bmbs_fastq_view* mate2, int64_t n_pairs, int32_t L_max,
                      int32_t uniform, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used);

/* ---- FASTQ text in, SAM text out: both ends of the file-to-file path on the device -------------------------------------------------
 * The reference's reader and its SAM sink are one host thread each (Process_Reads.cpp:2057-2260, Process_sam_out.cpp:954-1006) and
 * bound the whole program.  Here the host hands over the text window exactly as it came from the file(s) together with the number
 * of complete records it holds (it only has to count newlines), and gets the finished SAM lines of those records back: the
 * newline index, the read rows, the mapping and the formatting of output_sam_end_to_end / directly_output_read1 / _read2 /
 * output_sam_unmapped / directly_output_unmapped_PE (Schema.cpp:11989-12039, 10537-10640, 11494-11590, 23955-23975, 10392-10430)
 * all happen on the device; QNAME, SEQ and QUAL are taken from the FASTQ text that is resident there anyway.
 * text = n_records * 4 lines (a window may hold more: the rest is ignored); paired end: record i of text1 and of text2 are a pair.
 * sam receives the lines in input order, *sam_bytes their total size; BMBS_ENOMEM with *sam_bytes = the size needed when sam_cap is
 * too small.  Page-locked buffers (bmbs_host_alloc) move at link speed.  bmbs_sam_refs sets the RNAME strings (the names of the
 * index's sequences, in index order: bmbs_index_file_chrom_name) once per context.                                           */
#define BMBS_TEXT_PBAT      1   /* single end --pbat: reads are mapped as their reverse complement (Process_Reads.cpp:986-1075)   */
#define BMBS_TEXT_UNMAPPED  2   /* --unmapped_out: unmapped reads / pairs are printed with flag 4 / 77 + 141                      */
#define BMBS_TEXT_BAM      16   /* --bam: `sam` receives the batch's records as BAM inside complete BGZF blocks (a piece of a .bam file
                                 * behind its header) instead of SAM text: what the reference gets from htslib's sam_parse1 + bam_write1
                                 * per line (bam_prase.cpp:201-221), built and deflated on the device; the INFLATED bytes are the
                                 * reference's, block boundaries and compressed bytes are not.  *sam_bytes = compressed bytes        */
int bmbs_sam_refs(bmbs_ctx*, const char* const* names, int32_t n_names);
int bmbs_map_se_text(bmbs_ctx*, const char* text, uint64_t text_bytes, int64_t n_records, int32_t flags,
                     char* sam, uint64_t sam_cap, uint64_t* sam_bytes, int64_t* n_lines);
int bmbs_map_pe_text(bmbs_ctx*, const char* text1, uint64_t bytes1, const char* text2, uint64_t bytes2, int64_t n_pairs, int32_t flags,
                     char* sam, uint64_t sam_cap, uint64_t* sam_bytes, int64_t* n_lines);

/* ---- bgzip'ed FASTQ inflated on the device -------------------------------------------------------------------------------------------
 * The reference reads .gz input through zlib's gzread on its reader thread (Process_Reads.cpp:1455-1514).  A BGZF file (bgzip) is a
 * series of independent gzip members of at most 64 KiB of text whose compressed size stands in the header: a window of them is
 * inflated here by one wave per block, CRC-32 and ISIZE of every block checked.  comp = the bytes of n_blocks consecutive blocks
 * (block i at comp + blk_off[i], blk_off[n_blocks] = their end), text receives block i's bytes at text + out_off[i]
 * (out_off[i + 1] - out_off[i] = the ISIZE in block i's trailer).  Needs no index.  BMBS_EINVAL (bmbs_last_error says which
 * block) for anything zlib's inflate would refuse or a CRC that does not match.  Page-locked `text` moves at link speed.
 * nl_per_64k (optional): what the driver's reader would otherwise count on the host -- the newlines of the inflated text per 64 KiB
 * of the caller's WINDOW, in which text byte i sits at offset window_shift + i: nl_per_64k[j] = newlines among the bytes at window
 * offsets [j * 65536, (j + 1) * 65536); (window_shift + text length + 65535) / 65536 entries.                                  */
int bmbs_inflate_bgzf(bmbs_ctx*, const void* comp, uint64_t comp_bytes, const uint64_t* blk_off, const uint64_t* out_off, int64_t n_blocks,
                      char* text, uint64_t text_bytes, uint32_t* nl_per_64k, uint64_t window_shift);

/* ... and the same blocks as the INPUT of a text call, so that the inflated text never leaves the device.  Two phases, because a
 * reader has to know what a window left over before it can cut the next one:
 *   bmbs_text_open_bgzf   the window(s) are assembled on the device -- `prefix` (uncompressed bytes: what the previous window left
 *                         over) followed by the text of the blocks --, their lines are indexed, *n_records = the complete records
 *                         (pairs: of both mates) they hold, at most max_records; tail1 / tail2 receive the text BEHIND those records
 *                         (the next window's prefix; BMBS_ENOMEM with *tail_bytes = the size needed when tail_cap is too small);
 *                         last1 / last2: the window ends its file (an unterminated last line is closed, as the reader does)
 *   bmbs_text_map_open    maps the open batch: records / SAM text / BAM blocks exactly as bmbs_map_*_text returns them
 * The context holds one open batch at a time.                                                                                    */
typedef struct bmbs_ztext {
    const char*     prefix;        /* host; NULL when prefix_bytes == 0 */
    uint64_t        prefix_bytes;
    const void*     comp;          /* the bytes of n_blocks consecutive BGZF blocks (page-locked memory moves at link speed) */
    uint64_t        comp_bytes;
    const uint64_t* blk_off;       /* [n_blocks + 1] */
    const uint64_t* out_off;       /* [n_blocks + 1]: ISIZE prefix sums */
    int64_t         n_blocks;      /* may be 0 (a window made of the prefix alone) */
} bmbs_ztext;
int bmbs_text_open_bgzf(bmbs_ctx*, const bmbs_ztext* mate1, const bmbs_ztext* mate2 /* NULL: single end */, int64_t max_records,
                        int32_t last1, int32_t last2, int64_t* n_records, char* tail1, uint64_t tail_cap, uint64_t* tail1_bytes,
                        char* tail2, uint64_t* tail2_bytes);
/* (An ordinary .gz file -- ONE deflate stream per member -- is inflated by the driver's block-parallel host inflater, csrc/pgz.h.  Round 4
 * also carried a device form of that scheme, bmbs_inflate_gzip / bmbs_text_open_gzip: exact, and slower than the host's; removed in round 5.) */
int bmbs_text_map_open(bmbs_ctx*, int32_t flags, char* sam, uint64_t sam_cap, uint64_t* sam_bytes, int64_t* n_lines);

/* a21: per-ctx counters of the batches mapped so far = {reads, unique, ambiguous, mapped bases,
 * error bases} (Schema.cpp:25141-25146); bmbs_stats_allreduce sums them over the ctxs one process
 * drives (get_mapping_informations, Schema.cpp:451-476).  Multi-process jobs sum the five int64 with
 * one RCCL all-reduce in the host layer (bitmapperbs_amd.distributed).                             */
int bmbs_stats_get(bmbs_ctx*, int64_t stats[5]);
int bmbs_stats_reset(bmbs_ctx*);
int bmbs_stats_allreduce(bmbs_ctx** ctxs, int n, int64_t stats[5]);

/* ---- measurement ------------------------------------------------------------------------------- */
/* per-kernel HIP-event timings of the last bmbs_map_se[_device] call, on the ctx stream.
 * names/ms arrays of length >= *n (in: capacity, out: count).                                     */
int bmbs_profile_last(bmbs_ctx*, const char** names, float* ms, int* n);
/* the same HIP-event timings summed over every mapping call since bmbs_profile_reset (all lanes), and the number of calls (a
 * call split over two lanes counts twice): per-kernel averages of a timed region without a wait after every call           */
int bmbs_profile_total(bmbs_ctx*, const char** names, double* ms, int* n, int64_t* calls);
int bmbs_profile_reset(bmbs_ctx*);
/* event counters of the last call for the algorithmic-byte model:
 * c[0]=n_hash c[1]=n_ext(LF pairs) c[2]=n_sa c[3]=n_cand(windows filtered) c[4]=n_sw(jobs that ran the DP) c[5]=n_ungapped
 * c[6]=window bytes                                                                               */
int bmbs_counters_last(bmbs_ctx*, uint64_t c[8]);
/* all 32 words: c[0..7] as above, c[8] three-letter index steps taken (each counts three in c[1]); what the list kernels handled:
 * c[9] / c[10] / c[11] candidates located by the kernels for lists of 17..32 / 33..256 / more entries, c[12] lists of more than 32,
 * c[13] list entries filter_pairs read, c[14] candidates of re-seeded mates (--sensitive), c[15] located sites the paired-end vote
 * kernels dropped before sorting (no partner on the mate's list); c[16+4*kid+{0,1,2,3}] =
 * n_hash, n_ext, n_sa, n_ungapped of seeding kernel kid (0 k_seed_first, 1 k_seed_second, 2 k_seed_extra)          */
int bmbs_counters_all(bmbs_ctx*, uint64_t c[32]);
/* calls that were issued a second time with exact buffer sizes because a stage count (candidate slots, DP jobs, re-seeded
 * candidates) exceeded the capacity learned from earlier calls (see "launch sequence" below); diagnostic                      */
int64_t bmbs_retries(bmbs_ctx*);
/* diagnostic (bmbs_search --verbose, bench.py's e2e keys): out = { seconds the context's bmbs_map_*_text / bmbs_text_map_open calls kept
 * the link busy with uploads, with downloads (a copy and the wait for its end; waiting for another context's copy excluded), seconds
 * spent inside those calls, number of calls }                                                                                     */
int bmbs_text_times(bmbs_ctx*, double out[4]);
/* diagnostic: the Huffman code lengths the device's BGZF deflater (--bam) gives a table of symbol frequencies -- n <= 320 symbols,
 * maxbits <= 15; every used symbol gets a length, the lengths form a complete prefix code (tests/test_gpu_parity.py)              */
int bmbs_debug_huff_lengths(bmbs_ctx*, const uint32_t* freq, int32_t n, int32_t maxbits, uint8_t* len_out);

/* ---- index files (next-row (f)2: reader/writer of the reference's on-disk formats) ------------- */
typedef struct bmbs_index_file bmbs_index_file;
/* Load_Index/load_index equivalents: reads <prefix>.index, .index.bs.pac, .index.bs.index{,.bwt,.sa,.occ} */
bmbs_index_file* bmbs_index_file_load(const char* prefix);
void bmbs_index_file_view(const bmbs_index_file*, bmbs_index_view* out);
const char* bmbs_index_file_chrom_name(const bmbs_index_file*, int i);
void bmbs_index_file_free(bmbs_index_file*);
/* createIndex equivalent (Index.cpp:832-938) without the psascan dependency: FASTA -> the six files */
int bmbs_index_build(const char* fasta, const char* prefix, int n_threads);
/* the same builder with the suffix sort and every array-sized derivation (BWT planes, Occ counters, SA_flag, samples, 16-mer
 * rows) on HIP device `device_id`: the same six files byte for byte; a GRCh38-size genome (6.2 G suffixes) in well under a
 * minute instead of six on 64 host cores.  Needs about 60 bytes of device memory per base.  n_threads: host threads of the
 * FASTA / .pac preparation.  BMBS_ENODEV without a device (no silent fall-back to the host builder).                       */
int bmbs_index_build_device(int device_id, const char* fasta, const char* prefix, int n_threads);

/* first 16 hex digits of the sha256 over the library's sources (in the Makefile's LIB_SRCS order) at build time: lets a run show
 * that the .so it loaded was built from the sources it sits next to (bitmapperbs_amd.capi.sources_id() recomputes it)          */
const char* bmbs_build_id(void);

/* Page-locked host buffers for the host-pointer entry points (bmbs_map_se / bmbs_map_pe copy from and to them at
 * full link speed).  The reference's per-thread scratch is plain malloc (Schema.cpp:24344-24362); a caller that
 * keeps malloc'ed buffers still works, only slower.  NULL on failure.                                     */
void* bmbs_host_alloc(uint64_t bytes);
/* the same with the buffer's role stated: 1 = the host writes it and the device reads it (FASTQ windows), 2 = the device writes it
 * and the host reads it (SAM text), 0 = either                                                             */
void* bmbs_host_alloc_kind(uint64_t bytes, int32_t kind);
void  bmbs_host_free(void* p);
/* lets the device touch a page-locked buffer once (kind 1: reads it, 2: writes it, 0: both): the first copy to or from a fresh
 * buffer is several times slower than the later ones; a driver does this while it loads                                  */
int   bmbs_host_prefault(bmbs_ctx*, void* p, uint64_t bytes, int32_t kind);

#ifdef __cplusplus
}
#endif
#endif
