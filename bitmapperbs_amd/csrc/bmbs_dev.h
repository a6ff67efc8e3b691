// bitmapperbs_amd/csrc/bmbs_dev.h -- device-side data layouts of the gfx950 mapping path.
#ifndef BMBS_DEV_H
#define BMBS_DEV_H
#include <stdint.h>
#include <hip/hip_runtime.h>

typedef uint64_t u64; typedef uint32_t u32; typedef uint16_t u16; typedef uint8_t u8;

// ---- index as laid out in HBM ------------------------------------------------------------------
// The reference streams 40-byte blocks (4 x 16-bit relative counters + two 64-symbol bit-plane
// pairs) plus a 65 536-row super-block table, and keeps every 8th SA entry (bwt.h:1007-1136,
// 2449-2560).  On MI355X the dependent random gather is what costs, and HBM capacity (288 GB) is not
// a constraint, so the index is re-packed at attach time into:
//   occ   : one 16-byte block per 32 BWT symbols = { u32 count(T) before the block, u32 count(A) before the
//           block, u32 plane_T, u32 plane_A } (symbol j of a word = bit 31-j, absolute counts while the
//           text has < 2^32 symbols) -> every rank query is exactly ONE 16-byte lane load, no second-level
//           table; with one read per lane a wave-wide load costs one cache-line request per lane whatever
//           its width, so what matters is the number of pieces per query, not bytes (4 bit/symbol);
//   hash  : the 3^16+1 16-mer entries fused to one u64 each (36-bit row | gap nibble << 60), so a
//           lookup (entries key, key+1) is one 16-byte read;
//   sa    : the FULL suffix array (u32 per row), expanded once on the GPU from the sampled SA, so
//           locate is one 4-byte read instead of <=7 dependent LF steps;
//   t20   : 3^16 * 81 u64 (27.9 GB): for every 16-mer and every 4-letter continuation, what count_backward_as_much_1_terminate
//           (bwt.h:2081-2209) has decided after those four backward extensions -- stopped unique at depth 16..19, stopped
//           because the next letter does not occur, or still going with the depth-20 interval.  One lookup replaces the 16-mer
//           lookup plus up to four dependent Occ gathers of a seed; 288 GB of HBM is what makes the table affordable;
//   gen2  : the doubled genome (forward ++ reverse complement) 2 bits/base, 32 bases per u64
//           LSB-first (A0 C1 G2 T3), so both strands' windows are forward reads;
//   gen2p : the same bases as two bit planes -- word w = { low half: bit 0 of the letters of bases 32 w .. 32 w + 31, high half: their
//           bit 1 } -- which is the form the Myers filter works on (k_filter.hip): same footprint, same sectors per window, and no
//           de-interleaving per candidate (1.55 GB more at GRCh38 size).
struct DevIndex {
    const uint4* occ;
    const u64*   hash;
    const u32*   sa;
    const u64*   sa64;          // texts of >= 2^32 symbols: 64-bit suffix array instead of sa
    // ... and the counts of occ are relative to super-blocks of 2^sup_shift symbols (2^31: relative counts fit 32 bits, and a text of
    // up to 2^33 symbols has at most four super-blocks) whose {T, A} sums travel IN this struct, i.e. in scalar registers: the
    // reference's 65 536-symbol super-block table in memory cost one more request per rank query.  sup_shift == 0: absolute counts.
    int          sup_shift;
    u64          supT[4], supA[4];
    const u64*   gen2;
    const u64*   gen2p;
    const u64*   t20;           // optional: outcome of the first t_e extensions of every (16 + t_e)-mer (k_build_t20), else nullptr
    int          t_e;           // letters the table looks ahead: 4 (3^20 entries, 27.9 GB) or 5 (3^21 entries, 83.7 GB; GRCh38-size texts)
    // three backward extensions in one step (round 4): occ3[g * nb3 + row / 96] = { rows before the block whose three preceding text
    // letters are the trigram g, one bit per row of the block for "this row's are" }, g = d1 + 3 d2 + 9 d3 in extension order;
    // c3[g] = first row of the suffixes that begin with the trigram.  LF_d3(LF_d2(LF_d1(row))) = c3[g] + rank_g(row): one 16-byte
    // gather per interval end for three letters.  27 x 16 B per 96 rows = 4.5 B per row (27.9 GB at GRCh38 size); nullptr: off
    const uint4* occ3;
    const u64*   c3;
    u64          nb3;
    const u64*   chrom_start;   // n_chrom+1 cumulative starts (single strand)
    u64 G;                      // one-strand length
    u64 total;                  // 2G = total_SA_length
    u64 shapline;
    u64 C[3];                   // nacgt[0..2]: first row of G / T / A suffixes
    int n_chrom;
};

// one seed as recorded by the seeding kernel: SA interval [sp, sp+hits), seed length and read offset
struct SeedRec { u64 sp; u32 hits; u16 len; u16 off; };
#define BMBS_MAX_SEEDS 28

// per-read state carried between the stage kernels (SoA arrays indexed by read)
struct ReadState {
    u8*   verdict;      // 0 none, 1 exit A, 2 exit C, 3 general, 4 exact-ambiguous
    u8*   n_seeds;
    u8*   multi;        // is_mutiple_map
    u16*  mm_site;      // one_mismatch_site (exit C)
    u64*  exit_site;    // site for exit A / C
    SeedRec* seeds;     // [n][BMBS_MAX_SEEDS]
    u32*  n_cand;       // candidates to locate (general path)
    u64*  cand_off;     // exclusive scan of n_cand, n+1 entries
    u32*  n_votes;
    // reduction output
    u64*  best_site;
    int32_t* best_end;
    u32*  best_err;
    u32*  sbd;          // second_best_diff
    u8*   red_status;   // 0 none, 1 unique, 2 ambiguous
    u32*  job_flag;     // 1 if the winner needs K11-K13
    u64*  job_off;      // exclusive scan of job_flag
};

// read lengths of a batch.  len == nullptr: every read has length L and threshold k (the fixed-length entry points);
// otherwise read r has length len[r] <= L and threshold klut[len[r]] = min(31, (u64)(e_f * len)) tabulated on the host
// in the reference's double arithmetic (Schema.cpp:24546), L = the longest length and k = klut[L] (the largest).
struct ReadGeom {
    const u16* len; const u8* klut; int L; int k;
    __device__ __forceinline__ int rl(long r) const { return len ? (int)len[r] : L; }
    __device__ __forceinline__ int rk(int Lr) const { return len ? (int)klut[Lr] : k; }
};

struct ScoreParams {
    int mp_max, mp_min, np, gap_open, gap_ext, q_base;
    int seed_len;
};
struct bmbs_result_dev {   // == bmbs_result (include/bmbs.h), 32 bytes
    u64 pos; u32 cigar_off; int32_t chrom; u16 flag; u16 nm; int16_t score; u8 status; u8 mapq; u8 n_cigar; u8 path; u16 n_cand; u32 tlen;
};

// Event / stats counters are sharded: 64 shards of 32 u64 (256 B apart), shard = blockIdx & 63, summed on the host.
// One global word sustains only ~90 atomics/us on MI355X; with >100 k blocks per launch un-sharded counters
// cost more than the kernels themselves (k_seed_decide: 4.0 ms -> 1.2 ms).
#define BMBS_SHARDS 64
#define BMBS_SHARD_WORDS 32
#endif
