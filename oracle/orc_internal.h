// oracle/orc_internal.h -- TEST INFRASTRUCTURE (see bmbs_oracle.h)
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "bmbs_oracle.h"
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t  u8;

struct orc_chrom { std::string name; u64 len, start, end; };

// the reference's global `bitmapper_index_params` (bwt.h:34-163) + chromosome table + .pac
struct orc_index {
    std::vector<orc_chrom> chroms;
    u64 G = 0;                       // _msf_refGenLength
    std::vector<u8> pac; u64 pac_bytes = 0;
    u64 SA_length = 0, shapline = 0, nacgt[5] = {0, 0, 0, 0, 0};
    u32 compress_sa = 8, compress_occ = 64, high_compress_occ = 128;
    std::vector<u64> bwt; u64 bwt_len = 0;
    std::vector<u32> hash_hi; std::vector<u8> hash_lo; u64 hash_size = 0;
    std::vector<u32> sa;
    std::vector<u64> sa_flag; u64 sa_flag_len = 0;
    std::vector<u64> high_occ; u64 high_occ_len = 0;
    u64 total = 0;                   // total_SA_length = 2G
};

int  orc_build_from_genome(const std::vector<orc_chrom>& chroms, const std::string& gen, const char* prefix);
void orc_hash_query(const orc_index* ix, u64 key, u64* sp, u64* ep);
u64  orc_lf(const orc_index* ix, u64 row, int c);
u64  orc_sa_row(const orc_index* ix, u64 row);
u64  orc_sa_row_counted(const orc_index* ix, u64 row, u64* n_lf);

// ---- shared by orc_map.cpp (SE) and orc_pe.cpp (PE) ---------------------------------------------
struct seed_res { u64 hits, sp, ep, match_len; };
struct vote_t { u64 site, vote; unsigned err; u64 end_site; };      // seed_votes, Schema.h:169-176
struct fq_rec { std::string name, seq, rseq, qual; };
void window_at(const orc_index* ix, u64 site, u64 len, char* out);
int  mismatch_penalty(const orc_params* P, int Q);
seed_res count_terminate(const orc_index* ix, const char* pat, u64 length, orc_counters* C);
seed_res count_fixed(const orc_index* ix, const char* pat, u64 length, orc_counters* C);
void locate_rows(const orc_index* ix, u64 sp, u64 ep, u64 seed_len, u64 seed_off, std::vector<u64>& out, orc_counters* C);
int  seed_offset_unmatch(int readLen, int pre, const char* read, int step);
void make_votes(const std::vector<u64>& cand, u64 k, std::vector<vote_t>& votes);
bool getline_(FILE* f, std::string& s);
void sam_header(FILE* o, const orc_index* ix, const char* argv_line);
#endif
