/* oracle/bmbs_oracle.h -- CPU restatement of the BitMapperBS --search hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * import, link or execute anything under oracle/.  The product (bitmapperbs_amd/) never does.
 *
 * Parity pin: the SAM text and mapstats this restatement produces are compared byte-for-byte with
 * the real reference binary (oracle/_ref/bitmapperBS, built by oracle/build_ref.sh from
 * /root/reference) in tests/test_oracle_vs_reference.py, and with the committed golden SAM
 * fixtures under tests/golden/ that the same binary generated (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 */
#ifndef BMBS_ORACLE_H
#define BMBS_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_index orc_index;

/* scoring / mode parameters == the reference's CLI globals (Process_CommandLines.cpp:40-75) */
typedef struct orc_params {
    double  e_f;          /* -e  (thread_e_f, default 0.08)                         */
    int     mp_max;       /* --mp_max 6   */
    int     mp_min;       /* --mp_min 2   */
    int     np;           /* --np 1       */
    int     gap_open;     /* --gap_open 5 */
    int     gap_ext;      /* --gap_extension 3 */
    int     q_base;       /* 33 (--phred33) or 64 */
    int     seed_len;     /* --seed, over_all_seed_length = 30 */
    int     min_ins;      /* --min 0   */
    int     max_ins;      /* --max 500 */
    int     sensitive;    /* --sensitive */
    int     unmapped_out; /* --unmapped_out  (search entry points only: extra SAM records) */
    int     ambiguous_out;/* --ambiguous_out (one hit of each ambiguous read / pair is aligned and reported) */
    int     pbat;         /* --pbat */
} orc_params;

void orc_default_params(orc_params* p);

/* ---- index (Index.cpp:514-1045, bwt.cpp:1108-2355 formats; naive SA) -------------------------- */
/* build all index files <prefix>.index, .index.bs.pac, .index.bs.index{,.bwt,.sa,.occ} from FASTA */
int        orc_index_build(const char* fasta, const char* prefix);
orc_index* orc_index_load(const char* prefix);
void       orc_index_free(orc_index*);
uint64_t   orc_index_genome_len(const orc_index*);

/* ---- stage-level entry points (used by the GPU parity tests as the checker) ------------------- */
/* K7: window fetch, Schema.cpp:4998-5115.  out[len] ASCII (all-zero when out of range)           */
void orc_window(const orc_index*, uint64_t site, int len, char* out);
/* K8: BS banded Myers, Levenshtein_Cal.h:351-567.  returns end_site (-1 if > k), *err            */
int  orc_bpm(const char* window, int p_len, const char* read, int t_len, int k, unsigned* err);
/* K11-K13: fast_recalculate_bs_Cigar, ksw.cpp:2578-2876.  cigar gets the text; returns 0        */
int  orc_align(const orc_params*, const char* window, int p_len, const char* read, const char* qual,
               int t_len, int k, int end_site, unsigned err, int is_forward, int reverse_quality,
               int* start_site, int* new_end_site, unsigned* nm, int* score, char* cigar);
/* a19: MAP_Calculation, Schema.cpp:168-405 */
int  orc_mapq(const orc_params*, unsigned second_best_diff, unsigned k, int score);
/* K3/K4: count_backward_as_much_1_terminate / count_hash_table (bwt.h:2081, 1848) on bsSeq       */
uint64_t orc_count_terminate(const orc_index*, const char* bsseq, uint64_t len, uint64_t* sp,
                             uint64_t* ep, uint64_t* match_len);
/* K5: locate one row -> text position SA[row] (bwt.h:2449-2560) */
uint64_t orc_sa_at(const orc_index*, uint64_t row);

/* per-read mapping record: what the SAM line is printed from */
typedef struct orc_rec {
    int32_t  status;        /* 0 unmapped/none, 1 unique (emitted), 2 ambiguous (fields valid only with ambiguous_out), 3 off-end rejected */
    int32_t  chrom;         /* chromosome id                                  */
    uint64_t pos;           /* 1-based position                               */
    uint64_t site;          /* doubled-coordinate window start (candidate.site) */
    int32_t  start_site, end_site;
    int32_t  flag, mapq, nm, score;
    int32_t  path;          /* 1 exit A (exact unique), 2 exit C (1-mismatch), 3 general, 4 exit B */
    int32_t  n_cand, n_votes;
    char     cigar[1024];    /* 254 operations (what a product record can hold) of up to 4 characters */
} orc_rec;

/* event counters for SURVEY.md §8d algorithmic-byte accounting */
typedef struct orc_counters {
    uint64_t n_reads, n_hash, n_ext, n_lf, n_sa1, n_locate_rows, n_cand, n_sw, n_ungapped;
} orc_counters;

/* SE mapping of n reads held as SoA (seq/qual: n rows of stride bytes, len[i] valid).  recs[n].
 * stats[5] = reads, unique, ambiguous, mapped_bases, error_bases (Schema.cpp:25141-25146).      */
int orc_map_se(const orc_index*, const orc_params*, const char* seq, const char* qual,
               const int32_t* len, int stride, int64_t n, orc_rec* recs, int64_t stats[5],
               orc_counters* counters);

/* orc_map_se plus the vote lists (site, count) of the general-path reads in the reference's visiting order (a9/a10) */
int64_t orc_map_se_votes(const orc_index*, const orc_params*, const char* seq, const char* qual, const int32_t* len, int stride,
                         int64_t n, orc_rec* recs, int64_t stats[5], uint64_t* vote_site, uint32_t* vote_cnt, uint64_t* vote_off,
                         int64_t cap);

/* paired-end record (fast mode, Map_Pair_Seq_end_to_end_fast, Schema.cpp:18570) */
typedef struct orc_pe_rec {
    int32_t  status;        /* 0 unmapped, 1 unique pair (emitted), 2 ambiguous, 3 rejected by TLEN / chromosome-end check */
    int32_t  n_pairs;       /* mapping_pair */
    int32_t  mapq, tlen;
    int32_t  flag1, flag2, chrom1, chrom2;
    uint64_t pos1, pos2;
    int32_t  nm1, nm2, score1, score2, matched1, matched2;
    char     cigar1[1024], cigar2[1024];
} orc_pe_rec;

/* PE mapping of n pairs (SoA, common stride).  seq2 = mate 2 as the reference's reader hands it on, i.e.
 * the REVERSE COMPLEMENT of the FASTQ record (Process_Reads.cpp:262-267); qual2 in FASTQ order.
 * stats[5] = pairs, unique, ambiguous, mapped bases, error bases (Schema.cpp:19531-19537).          */
int orc_map_pe(const orc_index*, const orc_params*, const char* seq1, const char* qual1, const char* seq2,
               const char* qual2, int L1, int L2, int stride, int64_t n, orc_pe_rec* recs, int64_t stats[5],
               orc_counters* counters);

int orc_map_pe_var(const orc_index*, const orc_params*, const char* seq1, const char* qual1, const char* seq2,
                   const char* qual2, const int32_t* len1, const int32_t* len2, int stride, int64_t n, orc_pe_rec* recs,
                   int64_t stats[5], orc_counters* counters);

/* whole-program equivalents (FASTQ -> SAM); return 0 on success.  argv_line is printed in @PG CL */
int orc_search_se(const orc_index*, const orc_params*, const char* fastq, const char* out_sam,
                  const char* argv_line, int64_t stats[5]);
int orc_search_pe(const orc_index*, const orc_params*, const char* fq1, const char* fq2,
                  const char* out_sam, const char* argv_line, int64_t stats[5]);

#ifdef __cplusplus
}
#endif
#endif
