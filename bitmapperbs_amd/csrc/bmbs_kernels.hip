// bitmapperbs_amd/csrc/bmbs_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the
// bisulfite mapping hot path.  Integer / bit work only: no MFMA, HBM-gather bound.
//
// Stage split (one kernel per work granularity, exclusive scans / compacted lists in between so that every kernel runs
// dense lanes):
//   k_seed_first / k_seed_decide / k_seed_second / k_seed_extra
//                      read per lane (wave-owned chunks)  K1-K5 + the seeding state machine of Map_Single_Seq_end_to_end
//   k_vote_fused / k_vote_mid / k_vote_long
//                      read (<= 16, <= 32 candidates) per lane, wave or block per read beyond
//                                                          K5/K6 + a9/a10: locate, sort, run-length votes, std::sort-exact order
//   k_filter           candidate per lane                  K7+K8: window fetch + BS banded Myers on bit planes
//   k_reduce           read per lane                       K9: ordered min / ambiguity / second_best_diff scan
//   k_align_ungapped / k_align_sw<KB>
//                      winner per lane                     K11-K13: un-gapped recheck, banded affine SW + CIGAR + NM
//   k_finalize         read per lane                       MAPQ LUT, doubled -> chromosome coordinates, off-end, stats
//   k_pe_* / k_pes_* / k_*_pe                              the paired-end (fast and --sensitive) counterparts
#include "bmbs_dev.h"
#include "bmbs_sort.h"
#include <type_traits>

#define DEVI __device__ __forceinline__

// Event / stats counters are sharded: 64 shards of 32 u64 (256 B apart), shard = blockIdx & 63, summed on the host.
// One global word sustains only ~90 atomics/us on MI355X; with >100 k blocks per launch un-sharded counters
// cost more than the kernels themselves (k_seed_decide: 4.0 ms -> 1.2 ms).
#define BMBS_SHARDS 64
#define BMBS_SHARD_WORDS 32
#define SHARD(p) ((p) + (size_t)(blockIdx.x & (BMBS_SHARDS - 1)) * BMBS_SHARD_WORDS)

// ---- per-wave timeline (diagnostic, BMBS_WAVELOG=<file>) ------------------------------------------------------------------------
// A kernel that calls wavelog_begin / wavelog_end records for each of its working waves where it ran (XCC, SE, CU, SIMD), when
// (100 MHz wall clock) and for how many shader cycles: 4 u64 per wave, dumped by bmbs_destroy and read by tools/wavelog.py.
// That is the only way to see occupancy rounds, dispatch imbalance and the actual shader clock of one kernel; with the variable
// unset g_wavelog.buf is null and the cost is one scalar load per wave.
struct WaveLogDev { unsigned long long* buf; unsigned int* count; unsigned int cap; };
__device__ WaveLogDev g_wavelog;
struct WaveLogT { u64 wall, cyc; bool on; };
DEVI WaveLogT wavelog_begin()
{
    WaveLogT t; t.on = g_wavelog.buf != nullptr; t.wall = 0; t.cyc = 0;
    if (t.on) { t.wall = wall_clock64(); t.cyc = __builtin_readcyclecounter(); }
    return t;
}
DEVI void wavelog_end(const WaveLogT& t, int kernel_id)
{
    if (!t.on) return;
    const u64 w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) != 0) return;
    const u32 hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const u32 idx = atomicAdd(g_wavelog.count, 1u);
    if (idx >= g_wavelog.cap) return;
    unsigned long long* o = g_wavelog.buf + (size_t)idx * 4;
    o[0] = ((u64)kernel_id << 56) | ((u64)(xcc & 0xffu) << 32) | hw;
    o[1] = t.wall; o[2] = w1; o[3] = c1 - t.cyc;
}

// ================================================================================================
// index primitives
// ================================================================================================

// rank of T and of A in BWT stream [0, line): ONE aligned 16-byte load (see bmbs_dev.h).
// Replaces get_occ_value* + the popcount tail of find_occ_fm_index (bwt.h:1007-1136, 1373-1465).
// Texts of 2^32 symbols and more (GRCh38: 2G = 6.2 G): the block counts are relative to super-blocks of 2^31 symbols whose
// sums sit in the DevIndex itself (scalar registers, picked by compare + select: no memory request), the suffix array is 64-bit
// (ix.sa64); the branches are wave-uniform.
DEVI void super_add(const DevIndex& ix, u64 line, u64& cT, u64& cA)
{
    const u32 S = (u32)(line >> ix.sup_shift);
    cT += S == 0 ? ix.supT[0] : S == 1 ? ix.supT[1] : S == 2 ? ix.supT[2] : ix.supT[3];
    cA += S == 0 ? ix.supA[0] : S == 1 ? ix.supA[1] : S == 2 ? ix.supA[2] : ix.supA[3];
}
DEVI void occ_TA(const DevIndex& ix, u64 line, u64& cT, u64& cA)
{
    const uint4 h = ix.occ[line >> 5];
    const u32 r = (u32)line & 31u;
    const u32 m = r ? (~0u << (32 - r)) : 0u;
    cT = (u64)h.x + __popc(h.z & m);
    cA = (u64)h.y + __popc(h.w & m);
    if (ix.sup_shift) super_add(ix, line, cT, cA);
}
DEVI u64 sa_at(const DevIndex& ix, u64 row) { return ix.sa64 ? ix.sa64[row] : (u64)ix.sa[row]; }

// one LF / backward-extension step: nacgt[c] + Occ(c, row), '$' row removed (bwt.h:1373-1465)
DEVI u64 lf_step(const DevIndex& ix, u64 row, int c)
{
    const u64 line = row - (row > ix.shapline ? 1 : 0);
    u64 cT, cA;
    occ_TA(ix, line, cT, cA);
    const u64 cnt = c == 1 ? cT : (c == 2 ? cA : line - cT - cA);
    return ix.C[c] + cnt;
}

// both ends of an SA interval in one go (find_occ_fm_index_combine, bwt.h:1473-1596): when top and
// bot fall into the same 32-symbol block -- the usual case once the interval is small -- one load serves both
DEVI void lf_pair(const DevIndex& ix, u64& top, u64& bot, int c)
{
    const u64 lt = top - (top > ix.shapline ? 1 : 0), lb = bot - (bot > ix.shapline ? 1 : 0);
    const uint4 ht = ix.occ[lt >> 5];
    uint4 hb = ht;
    if ((lb >> 5) != (lt >> 5)) hb = ix.occ[lb >> 5];
    const u32 rt = (u32)lt & 31u, rb = (u32)lb & 31u;
    const u32 mt = rt ? (~0u << (32 - rt)) : 0u, mb = rb ? (~0u << (32 - rb)) : 0u;
    u64 tT = (u64)ht.x + __popc(ht.z & mt), tA = (u64)ht.y + __popc(ht.w & mt);
    u64 bT = (u64)hb.x + __popc(hb.z & mb), bA = (u64)hb.y + __popc(hb.w & mb);
    if (ix.sup_shift) { super_add(ix, lt, tT, tA); super_add(ix, lb, bT, bA); }
    const u64 ct = c == 1 ? tT : (c == 2 ? tA : lt - tT - tA);
    const u64 cb = c == 1 ? bT : (c == 2 ? bA : lb - bT - bA);
    top = ix.C[c] + ct; bot = ix.C[c] + cb;
}

// ---- three letters per step (DevIndex::occ3) ------------------------------------------------------------------------------------
DEVI u32 occ3_in_block(const uint4& h, u32 r)        // rows 0 .. r-1 of the block that carry the trigram
{
    const u32 m0 = r >= 32 ? ~0u : ((1u << r) - 1u);
    const u32 m1 = r >= 64 ? ~0u : (r > 32 ? ((1u << (r - 32)) - 1u) : 0u);
    const u32 m2 = r > 64 ? ((1u << (r - 64)) - 1u) : 0u;
    return h.x + __popc(h.y & m0) + __popc(h.z & m1) + __popc(h.w & m2);
}
// both ends of an interval through the trigram g: c3g + rank_g(row) (c3g = c3[g], from the block's LDS copy)
DEVI void lf3_pair(const DevIndex& ix, int g, u64 c3g, u64& top, u64& bot)
{
    const u64 bt = top / 96, bb = bot / 96;
    const uint4* base = ix.occ3 + (u64)g * ix.nb3;
    const uint4 ht = base[bt];
    uint4 hb = ht;
    if (bb != bt) hb = base[bb];
    top = c3g + occ3_in_block(ht, (u32)(top - bt * 96));
    bot = c3g + occ3_in_block(hb, (u32)(bot - bb * 96));
}
// the block's copy of c3 (27 words of LDS): a per-lane load from global memory would be one more request per step
DEVI const u64* kgram_c3(const DevIndex& ix, u64* lds)
{
    if (!ix.occ3) return nullptr;
    if (threadIdx.x < 27) lds[threadIdx.x] = ix.c3[threadIdx.x];
    __syncthreads();
    return lds;
}

// BWT symbol of a row (access_bwt_delta, bwt.h:2413-2447); only used while expanding the SA
DEVI int bwt_sym(const DevIndex& ix, u64 row)
{
    const u64 line = row - (row > ix.shapline ? 1 : 0);
    const uint4 h = ix.occ[line >> 5];
    const int sh = 31 - (int)(line & 31);
    if ((h.z >> sh) & 1) return 1;
    if ((h.w >> sh) & 1) return 2;
    return 0;
}

// query_16_mer_hash_table (bwt.h:284-306) on the fused entries
DEVI void hash_lookup(const DevIndex& ix, u64 key, u64& sp, u64& ep)
{
    const u64 e0 = ix.hash[key], e1 = ix.hash[key + 1];
    const u64 m36 = (1ull << 36) - 1;
    sp = e0 & m36;
    ep = (e1 & m36) - (e1 >> 60);
}

// base code (A0 C1 G2 T3) at doubled coordinate d
DEVI int gbase(const DevIndex& ix, u64 d) { return (int)((ix.gen2[d >> 5] >> ((d & 31) * 2)) & 3); }

// chromosome of a forward-strand coordinate: the c with chrom_start[c] <= loc < chrom_start[c + 1], n_chrom when there is none.
// Binary search: an assembly with its alternate contigs and decoys has thousands of sequences, and this runs once per read.
DEVI int chrom_of(const u64* chrom_start, int n_chrom, u64 loc)
{
    if (loc >= chrom_start[n_chrom]) return n_chrom;
    int lo = 0, hi = n_chrom;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chrom_start[mid] <= loc) lo = mid; else hi = mid;
    }
    return lo;
}
DEVI int chrom_of(const DevIndex& ix, u64 loc) { return chrom_of(ix.chrom_start, ix.n_chrom, loc); }
// The finalize kernels place every read by a binary search over the chromosome starts: five dependent loads plus two for the
// chromosome's bounds, in kernels that are nothing but chains of dependent loads (k_finalize_pe: 61 % of its wave cycles parked
// on memory).  Assemblies of up to BMBS_CS_LDS - 1 sequences get the table copied into LDS by every block first.
#define BMBS_CS_LDS 1025
DEVI const u64* chrom_table(const DevIndex& ix, u64* lds)
{
    if (ix.n_chrom + 1 > BMBS_CS_LDS) return ix.chrom_start;
    for (int i = threadIdx.x; i <= ix.n_chrom; i += blockDim.x) lds[i] = ix.chrom_start[i];
    return lds;
}

// window validity: get_actuall_genome / get_actuall_rc_genome return an all-zero window when the
// request leaves the strand (Schema.cpp:5013-5019, 5076-5084; u64 wrap-around as in the reference)
DEVI bool window_valid(const DevIndex& ix, u64 start, u64 len, bool fwd_strand)
{
    return fwd_strand ? (start + len <= ix.G) : (start - ix.G + len <= ix.G && start - ix.G < ix.G);
}

// window base with the all-zero-window rule (out-of-strand request: every base compares unequal and
// scores as N, nt4[0] = 4)
struct WinReader {
    // the word after the current one is requested as soon as the current one is taken into use, so that its latency
    // overlaps the 32 bases of work in between (one read per lane: nothing else would hide it)
    const u64* g; u64 pos, w, nxt; int left; bool valid;
    DEVI void init(const DevIndex& ix, u64 start, bool v)
    {
        g = ix.gen2; valid = v; pos = start;
        if (v) { w = g[pos >> 5] >> ((pos & 31) * 2); nxt = g[(pos >> 5) + 1]; left = 32 - (int)(pos & 31); } else { w = 0; nxt = 0; left = 32; }
    }
    DEVI int next()
    {
        if (!valid) return 4;
        const int b = (int)(w & 3);
        w >>= 2; pos++; left--;
        if (left == 0) { w = nxt; nxt = g[(pos >> 5) + 1]; left = 32; }      // gen2 carries two spare words at its end
        return b;
    }
};

// The window of one banded alignment, staged in LDS (the DP kernels).  WinReader's prefetch does not survive the compiler: the
// reload sits in a branch that some lane of the wave takes in nearly every row (each lane has its own phase, start & 31), and
// the s_waitcnt for the loaded word is placed at the join right behind the load -- one exposed HBM latency per row, 53 % of
// the wave cycles of k_align_sw2 (SQ_WAIT_ANY).  Here the lane copies the (L + 2k) / 32 + 2 words of its window to LDS once,
// all loads in flight together, and a reload is a ds_read (lgkmcnt, ~100 cycles, independent of the trace stores' vmcnt).
// Word m of lane l lives at base[m * 64 + l]: consecutive lanes, consecutive banks.
struct LdsWin {
    const u64* p; u64 w; int left; bool valid;
    DEVI void init(const DevIndex& ix, u64 start, bool v, u64* lane_base, int nww)       // nww = (window length + 62) / 32
    {
        valid = v; w = 0; left = 32; p = lane_base;
        if (!v) return;
        const u64* g = ix.gen2 + (start >> 5);
#pragma unroll 4
        for (int m = 0; m < nww; m++) lane_base[m * 64] = g[m];                         // gen2 carries two spare words at its end
        const int off = (int)(start & 31);
        w = lane_base[0] >> (2 * off); left = 32 - off; p = lane_base + 64;
    }
    DEVI int next()
    {
        if (!valid) return 4;
        const int b = (int)(w & 3);
        w >>= 2;
        if (--left == 0) { w = *p; p += 64; left = 32; }
        return b;
    }
};
// bytes of a 16-byte aligned row from position p downwards (the qualities of a reverse-strand read), 16 per global load
struct RevCur {
    const char* rd; u64 lo, hi; int pos;
    DEVI void load(int at) { const uint4 v = *reinterpret_cast<const uint4*>(rd + at); lo = ((u64)v.y << 32) | v.x; hi = ((u64)v.w << 32) | v.z; }
    DEVI void seek(const char* r, int p) { rd = r; pos = p; lo = 0; hi = 0; if (p >= 0) load(p & ~15); }
    DEVI unsigned char next()
    {
        const int o = pos & 15;
        const unsigned char c = (unsigned char)(((o & 8) ? hi : lo) >> (8 * (o & 7)));
        pos--;
        if (o == 0 && pos >= 0) load(pos & ~15);
        return c;
    }
};

// bisulfite 3-letter code of a read character after C->T: G0 T1 A2, anything else 4
// (C_to_T_forward, Schema.h:1534; ctoi, bwt.cpp:2376-2381)
DEVI int code3(char ch) { return ch == 'G' ? 0 : (ch == 'T' || ch == 'C') ? 1 : ch == 'A' ? 2 : 4; }
DEVI int code4(char ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4; }

// sequential reader of a read's characters, 8 bytes per global load (rows are 16-byte aligned and
// padded to a multiple of 16, so the aligned u64 that holds any position < stride is in the row).
// One read per lane means a byte load touches 64 different cache lines per wave instruction; wide
// loads cut the number of such instructions by 8.
struct ReadCur {
    // 16 characters per global load (rows are 16-byte aligned); the second half waits in `hi` until the first eight
    // characters are used up
    const char* rd; u64 buf, hi; int pos, lim;
    DEVI void fill(int at)                        // at: multiple of 16, < lim
    {
        const uint4 v = *reinterpret_cast<const uint4*>(rd + at);
        buf = ((u64)v.y << 32) | v.x; hi = ((u64)v.w << 32) | v.z;
    }
    DEVI void seek(const char* r, int p, int L)
    {
        rd = r; pos = p; lim = L; buf = 0; hi = 0;
        if (p < L) {
            fill(p & ~15);
            if (p & 8) buf = hi;
            buf >>= 8 * (p & 7);
        }
    }
    // seek with the 16-byte piece that holds position p already in registers
    DEVI void seek_with(const char* r, int p, int L, const uint4& v)
    {
        rd = r; pos = p; lim = L; buf = 0; hi = 0;
        if (p < L) {
            buf = ((u64)v.y << 32) | v.x; hi = ((u64)v.w << 32) | v.z;
            if (p & 8) buf = hi;
            buf >>= 8 * (p & 7);
        }
    }
    DEVI char next()
    {
        const char c = (char)(buf & 0xff);
        pos++;
        if ((pos & 7) == 0) {
            if (pos & 8) buf = hi;
            else if (pos < lim) fill(pos);        // never past the read's last 16-byte piece
            else buf = 0;
        } else buf >>= 8;
        return c;
    }
};

// ---- 8 read characters against 8 window bases at a time ---------------------------------------
// 8 bases of the doubled 2-bit genome starting at doubled coordinate d, as 16 bits
DEVI u64 win16(const DevIndex& ix, u64 d)
{
    const int sh = (int)(d & 31) * 2;
    u64 w = ix.gen2[d >> 5] >> sh;
    if (sh > 48) w |= ix.gen2[(d >> 5) + 1] << (64 - sh);
    return w & 0xffff;
}
// 0x80 in every byte j of the result where read character j (byte j of rw, ASCII) does NOT match window base j
// (2-bit code j of w16) under the bisulfite rule: equal letters match, and read 'T' matches window 'C'
// (Schema.cpp:15212-15216; everything else, 'N' included, is a mismatch).
// Four positions at a time in 32-bit registers: the four 2-bit codes become a byte selector, v_perm_b32 turns the selector into
// the letters the read may show there -- the window letter itself, and 'T' where the window has 'C' -- and a position
// mismatches when the read byte differs from both.
template <u32 LUT1, u32 LUT2>
DEVI u32 mism4(u32 rw, u32 w8)
{
    u32 sel = (w8 | (w8 << 12)) & 0x000f000fu;
    sel = (sel | (sel << 6)) & 0x03030303u;
    const u32 d1 = rw ^ __builtin_amdgcn_perm(0u, LUT1, sel);
    const u32 d2 = rw ^ __builtin_amdgcn_perm(0u, LUT2, sel);
    const u32 K7F = 0x7f7f7f7fu;
    return (((d1 & K7F) + K7F) | d1) & (((d2 & K7F) + K7F) | d2) & 0x80808080u;      // byte != 0, both times
}
DEVI u64 mism8(u64 rw, u64 w16)
{
    // "ACGT" and "ATGT"
    return (u64)mism4<0x54474341u, 0x54475441u>((u32)rw, (u32)w16 & 0xffu) |
           ((u64)mism4<0x54474341u, 0x54475441u>((u32)(rw >> 32), ((u32)w16 >> 8) & 0xffu) << 32);
}

// 0x80 in byte j where read character j does NOT equal window base j in the 3-letter (C->T) alphabet the FM index
// is built over: A=A, G=G, {C,T}={C,T}; any other read character never matches (ctoi > 2, bwt.h:1894).
// Same scheme: the read may show the window letter with C folded into T ("ATGT") or with T folded into C ("ACGC").
DEVI u64 mism8_3letter(u64 rw, u64 w16)
{
    return (u64)mism4<0x54475441u, 0x43474341u>((u32)rw, (u32)w16 & 0xffu) |
           ((u64)mism4<0x54475441u, 0x43474341u>((u32)(rw >> 32), ((u32)w16 >> 8) & 0xffu) << 32);
}

// sequential reader of the doubled 2-bit genome, 16 bases (32 bits) per step; two words are kept in registers and one
// global load is issued per 32 bases (gen2 carries two spare words at its end)
struct Win32Cur {
    const u64* g; u64 lo, hi; long idx;
    DEVI void init(const DevIndex& ix, u64 d) { g = ix.gen2; idx = (long)(d >> 5); lo = g[idx]; hi = g[idx + 1]; }
    DEVI u32 at(u64 d)                          // the 16 bases starting at d; d never decreases and advances by <= 32 per call
    {
        const long id2 = (long)(d >> 5);
        if (id2 != idx) { lo = hi; hi = g[id2 + 1]; idx = id2; }
        const int sh = (int)(d & 31) * 2;
        u64 w = lo >> sh;
        if (sh) w |= hi << (64 - sh);
        return (u32)w;
    }
};
// number of positions j < len (len <= 8) where read character rd[ts + j] does not match window base d + j (mism8's rule);
// rd + (ts & ~7) is an aligned u64 inside the read's row, the following one is touched only when the span crosses into it
DEVI int mism_span(const DevIndex& ix, const char* rd, int ts, u64 d, int len)
{
    const u64* p = reinterpret_cast<const u64*>(rd + (ts & ~7));
    const int sh = (ts & 7) * 8;
    u64 rw = p[0] >> sh;
    if (sh && (ts & 7) + len > 8) rw |= p[1] << (64 - sh);
    const u64 keep = len >= 8 ? ~0ull : ((1ull << (8 * len)) - 1);
    return __popcll(mism8(rw, win16(ix, d)) & keep & 0x8080808080808080ull);
}

// ================================================================================================
// packed read rows (round 2)
// ================================================================================================
// Every kernel of the seeding engine starts a seed by reading 21 characters at a new offset of its read, and every such
// per-lane load is a request of its own to the memory pipeline (a row load costs 3/4 of a random gather, tools/gather_bench):
// with ASCII rows that is two or three 16-byte requests per seed start plus one per 16 characters walked.  k_pack_rows
// therefore writes, once per batch, a packed copy of every row -- 2 bits per base (A0 C1 G2 T3, LSB first like gen2) followed by
// one bit per base that says "this character is not one of ACGT" (then the base bits are 0): 64 bytes for a 150-base read,
// ONE sector -- and a `dirty` byte per read that tells whether any such character exists at all.  A seed start is then one
// 16-byte request (64 bases from the word that holds its first base), the walk one 8-byte request per 32 bases, the 16-mer key
// falls out of the 2-bit codes with a few logic ops, and the comparisons with the 2-bit genome are XORs of whole words.  Only
// what asks for the letter 'N' itself (penalty np, determine_seed_offset_unmatch) still looks at the ASCII row, and only for
// dirty reads.
struct PackedRows {
    const u64* base;      // row r at base + r * pwords
    const u8*  dirty;     // [n] 1 = the row holds a character outside ACGT
    int pwords;           // u64 words per row (even: rows are 16-byte aligned)
    int W;                // words of bases; the mask words follow
};
__host__ __device__ inline int pack_base_words(int L) { return (L + 31) / 32; }
__host__ __device__ inline int pack_words(int L) { const int w = (L + 31) / 32 + (L + 63) / 64 + 1; return (w + 1) & ~1; }   // + one spare word for two-word loads at the end

// two consecutive words from an 8-byte aligned address: one 16-byte request
DEVI void load2(const u64* p, u64& a, u64& b)
{
    uint4 v;
    __builtin_memcpy(&v, __builtin_assume_aligned(p, 8), 16);
    a = ((u64)v.y << 32) | v.x; b = ((u64)v.w << 32) | v.z;
}
// 32 bases (or 64 mask bits shifted) starting at an arbitrary position
DEVI u64 prow_bases32(const u64* row, int pos)
{
    u64 a, b; load2(row + (pos >> 5), a, b);
    const int sh = 2 * (pos & 31);
    return sh ? (a >> sh) | (b << (64 - sh)) : a;
}
DEVI u32 prow_mask32(const u64* row, int W, int pos)      // mask bits of positions pos .. pos+31
{
    u64 a, b; load2(row + W + (pos >> 6), a, b);
    const int sh = pos & 63;
    return (u32)(sh ? (a >> sh) | (b << (64 - sh)) : a);
}
// 32 bases of the doubled genome starting at doubled coordinate d (gen2 carries spare words at its end)
DEVI u64 gen_bases32(const DevIndex& ix, u64 d)
{
    u64 a, b; load2(ix.gen2 + (d >> 5), a, b);
    const int sh = 2 * (int)(d & 31);
    return sh ? (a >> sh) | (b << (64 - sh)) : a;
}
// the doubled genome from coordinate d on, 32 bases per call, ONE 16-byte request per 64 bases (gen_bases32 takes one per 32:
// every load is a request to the memory pipeline whether it hits or not, and the window was 5 of the 7 divergent requests a read
// costs in k_seed_decide).  Reads one pair of words past the last one used: gen2 carries spare pieces at its end.
struct GenStream {
    const u64* p; u64 a, b; int sh; bool second;
    DEVI void init(const DevIndex& ix, u64 d) { p = ix.gen2 + (d >> 5); sh = 2 * (int)(d & 31); load2(p, a, b); p += 2; second = false; }
    DEVI u64 next32()
    {
        u64 lo = a, hi = b;
        if (second) { u64 na, nb; load2(p, na, nb); p += 2; lo = b; hi = na; a = na; b = nb; }
        second = !second;
        return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    }
};
// 32 bits -> the even bit positions of 64 bits
DEVI u64 spread32(u32 m)
{
    u64 x = m;
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
#define PK_EVEN 0x5555555555555555ull
// mismatches (bit 2j set = position j differs) of 32 read bases against 32 window bases
//   mism_bs:  the alignment rule -- equal letters match, and read T matches window C (Schema.cpp:15212-15216)
//   mism_3l:  the index alphabet -- A=A, G=G, {C,T}={C,T}
DEVI u64 mism_bs(u64 r, u64 g) { const u64 x = r ^ g; return (x | ((x >> 1) & ~(r & (r >> 1)))) & PK_EVEN; }
DEVI u64 mism_3l(u64 r, u64 g) { const u64 x = r ^ g; return (x | ((x >> 1) & ~r)) & PK_EVEN; }
// fields lo .. hi-1 (hi <= 32) as a mask of even bits
DEVI u64 field_range(int lo, int hi)
{
    const u64 up = hi >= 32 ? ~0ull : ((1ull << (2 * hi)) - 1);
    const u64 dn = lo <= 0 ? 0ull : ((1ull << (2 * lo)) - 1);
    return up & ~dn & PK_EVEN;
}

// cursor over the bases of a packed row: 3-letter digit of the next base (G0 T1 A2, C folded into T; 4 = outside ACGT)
struct PCur {
    const u64* row; u64 buf; int W, pos, have; bool dirty;
    DEVI int next3()
    {
        if (have == 0) { buf = row[pos >> 5] >> (2 * (pos & 31)); have = 32 - (pos & 31); }
        const int c = (int)(buf & 3);
        buf >>= 2; have--;
        int d = (0x46 >> (2 * c)) & 3;
        if (dirty && ((row[W + (pos >> 6)] >> (pos & 63)) & 1)) d = 4;
        pos++;
        return d;
    }
};

// cursor over a packed row for the alignment kernels: next4() = A0 C1 G2 T3, 4 for a character outside ACGT (code4's values)
struct PCode {
    const u64* row; u64 buf; int W, pos, have; bool dirty;
    DEVI void seek(const u64* r, int W_, bool d) { row = r; W = W_; dirty = d; pos = 0; have = 0; buf = 0; }
    DEVI int next4()
    {
        if (have == 0) { buf = row[pos >> 5]; have = 32; }
        int c = (int)(buf & 3);
        buf >>= 2; have--;
        if (dirty && ((row[W + (pos >> 6)] >> (pos & 63)) & 1)) c = 4;
        pos++;
        return c;
    }
};
// positions j < len where read[ts + j] does not match the window base at doubled coordinate d + j (mism8's rule), on a packed row
DEVI int count_mism_p(const DevIndex& ix, const u64* row, int W, bool dirty, int ts, u64 d, int len)
{
    int c = 0;
    GenStream gs; gs.init(ix, d);
    for (int o = 0; o < len; o += 32) {
        u64 mm = mism_bs(prow_bases32(row, ts + o), gs.next32());
        if (dirty) mm |= spread32(prow_mask32(row, W, ts + o));
        c += __popcll(mm & field_range(0, len - o));
    }
    return c;
}

// 16 characters (one 16-byte piece of an ASCII row) -> 32 bits of bases + 16 mask bits; characters at and beyond `valid` count as A.
// Fast path (every byte one of A C G T, the piece inside the read): a dozen 32-bit ops per four characters.
DEVI void pack_piece(const uint4& v, int valid, u32& bases, u32& mask)
{
    // branch-free: the last piece of every row is ragged (L = 150: six characters), so every wave would take a per-character
    // path beside the whole-piece one; the mask of the bad bytes is formed by SWAR instead and applied to all sixteen at once
    const u32 w[4] = {v.x, v.y, v.z, v.w};
    u32 out = 0, bad16 = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const u32 x = w[q];
        // (c >> 1) & 3 is A0 C1 T2 G3; swapping 2 and 3 gives A0 C1 G2 T3
        u32 c = (x >> 1) & 0x03030303u;
        c ^= (c >> 1) & 0x01010101u;
        // a byte that is not one of A C G T: rebuild the letter its bits 1-2 stand for and compare (as swar_code3)
        const u32 isT = (x >> 2) & ~(x >> 1) & 0x01010101u;
        const u32 b = x ^ (0x41414141u | (x & 0x06060606u)) ^ (isT * 0x11u);
        u32 y = ((((b & 0x7f7f7f7fu) + 0x7f7f7f7fu) | b) & 0x80808080u) >> 7;      // 1 in bit 0 of every byte that differs
        y = (y | (y >> 7) | (y >> 14) | (y >> 21)) & 0xfu;                         // the four bits side by side
        bad16 |= y << (4 * q);
        u32 t = (c | (c >> 6)) & 0x000f000fu;
        t = (t | (t >> 12)) & 0xffu;
        out |= t << (8 * q);
    }
    const u32 in16 = valid >= 16 ? 0xffffu : valid > 0 ? (1u << valid) - 1u : 0u;  // characters beyond the read's end are dropped
    mask = bad16 & in16;
    u32 k = in16 & ~bad16;                                                         // real bases -> both bits of their pair
    k = (k | (k << 8)) & 0x00ff00ffu; k = (k | (k << 4)) & 0x0f0f0f0fu; k = (k | (k << 2)) & 0x33333333u; k = (k | (k << 1)) & 0x55555555u;
    bases = out & (k | (k << 1));
}

// one thread per 16-byte piece of an ASCII row
__global__ void __launch_bounds__(256)
k_pack_rows(const char* __restrict__ seq, ReadGeom gm, int stride, long n, u64* __restrict__ prow, int pwords, int W, u32* __restrict__ dirty32)
{
    const long i16 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = stride / 16;
    if (i16 >= n * per_row) return;
    const long r = i16 / per_row;
    const int piece = (int)(i16 - r * per_row);
    const int L = gm.rl(r);
    if (piece * 16 >= ((L + 63) & ~63)) return;                 // beyond the row's last mask word: nothing to write
    const uint4 v = reinterpret_cast<const uint4*>(seq)[i16];
    u32 bases, mask;
    pack_piece(v, L - piece * 16, bases, mask);
    u64* row = prow + (size_t)r * pwords;
    if (piece * 16 < ((L + 31) & ~31)) reinterpret_cast<u32*>(row)[piece] = bases;
    reinterpret_cast<u16*>(row + W)[piece] = (u16)mask;
    if (mask) atomicOr(&dirty32[r >> 2], 1u << (8 * (int)(r & 3)));
}

// ================================================================================================
// attach-time re-pack kernels
// ================================================================================================
struct RefIndexDev {            // reference on-disk layouts, uploaded verbatim
    const u64* bwt; const u64* high_occ; const u32* hash_hi; const u8* hash_lo;
    const u32* sa; const u64* sa_flag; const u8* pac;
};

// rank in the reference layout at a 64-aligned stream position (bwt.h:1007-1081)
DEVI void ref_rank64(const RefIndexDev& R, u64 line, u64& cT, u64& cA)
{
    const u64 base = (line >> 7) * 5, half = (line & 127) >> 6, sb = (line >> 16) << 1;
    const u64 w0 = R.bwt[base];
    cT = R.high_occ[sb] + ((w0 >> (48 - 32 * half)) & 0xffff);
    cA = R.high_occ[sb + 1] + ((w0 >> (32 - 32 * half)) & 0xffff);
}

// one 16-byte block per 32 BWT symbols: { u32 count(T) before, u32 count(A) before, u32 plane_T, u32 plane_A }
struct SuperSums { int shift; u64 T[4], A[4]; };
__global__ void k_repack_occ(RefIndexDev R, u64 n_stream, u64 n_blk, SuperSums sup, uint4* out)
{
    const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blk) return;
    const u64 s0 = b * 32;
    const u64 s64 = s0 & ~63ull;          // the reference stores counters at every 64-boundary it reached (bwt.cpp:1437-1490)
    u64 cT = 0, cA = 0;
    ref_rank64(R, s64, cT, cA);
    if (sup.shift) { const u32 S = (u32)(s64 >> sup.shift); cT -= S == 0 ? sup.T[0] : S == 1 ? sup.T[1] : S == 2 ? sup.T[2] : sup.T[3];
                     cA -= S == 0 ? sup.A[0] : S == 1 ? sup.A[1] : S == 2 ? sup.A[2] : sup.A[3]; }     // relative to the super-block
    u32 pT = 0, pA = 0;
    if (s64 < n_stream) {
        const u64 wi = (s64 >> 7) * 5 + 1 + 2 * ((s64 & 127) >> 6);
        const u64 wT = R.bwt[wi], wA = R.bwt[wi + 1];
        if (s0 & 32) { cT += __popcll(wT >> 32); cA += __popcll(wA >> 32); pT = (u32)wT; pA = (u32)wA; }
        else { pT = (u32)(wT >> 32); pA = (u32)(wA >> 32); }
    }
    out[b] = make_uint4((u32)cT, (u32)cA, pT, pA);
}

__global__ void k_repack_hash(RefIndexDev R, u64 n, u64* out)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 hi = R.hash_hi[i];
    out[i] = (((u64)(hi & 0x0fffffffu)) << 8) | R.hash_lo[i] | ((u64)(hi >> 28) << 60);
}

// ---- (16 + E)-mer outcome table, E = ix.t_e = 4 or 5 ---------------------------------------------------------------
// entry = row (36 bits) | hits (24 bits) << 36 | tag << 60.  With I16 the interval of the 16-mer and c16, c17, ... the next E
// letters, count_backward_as_much_1_terminate does, for s = 0, 1, ...: stop if |I| == 1 (match length 16+s, 1 hit);
// extend by c(16+s); stop if that is empty (match length 16+s, hits of the interval before).  Tags:
//   1..E        stopped unique before consuming c(15 + tag)          (match length 15 + tag; the field holds SA[row], the TEXT
//               POSITION of that single row: whoever gets a unique seed needs nothing else from the row, and the suffix-array
//               gather -- one 64-byte sector for 4 or 8 bytes, per read -- is paid once, here, instead of per lookup)
//   E+1..2E     stopped because c(15 + tag - E) does not occur       (row, hits = interval before, match length 15 - E + tag)
//   0           all E letters consumed: row, hits = depth-(16+E) interval (the caller carries on with s = E)
//   2E+1        all E letters consumed and that interval is one row: the next iteration would stop there (match length 16 + E,
//               1 hit); the field holds the text position, as for tags 1..E
//   15          the 16-mer itself does not occur;   14  hits do not fit 24 bits: use the 16-mer path
// E = 4: 3^20 entries = 27.9 GB.  E = 5: 3^21 entries = 83.7 GB -- chosen for texts of 2^32 symbols and more, where a 20-mer
// still has ~2 occurrences (6.2 G suffixes / 3.5 G 20-mers) and every seed would walk 2-3 more dependent Occ gathers.
#define T20_MAX_E 5
DEVI u64 t20_entry(u64 row, u64 hits, int tag) { return hits >= (1ull << 24) ? (14ull << 60) : (row | (hits << 36) | ((u64)tag << 60)); }
__host__ __device__ inline u64 t20_width(int e) { return e == 5 ? 243ull : 81ull; }

__global__ void __launch_bounds__(256)
k_build_t20(DevIndex ix, u64 n_keys, u64* __restrict__ t20)
{
    const u64 key = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= n_keys) return;
    const int E = ix.t_e;
    const int W = (int)t20_width(E);
    u64* o = t20 + key * (u64)W;
    u64 t0, b0;
    hash_lookup(ix, key, t0, b0);
    if (b0 <= t0) { for (int e = 0; e < W; e++) o[e] = 15ull << 60; return; }
    // depth-first over c16 .. c(15+E): level l consumes c(16+l) = digit l of the entry index (least significant first); a stopped
    // prefix decides all of its continuations.  tp/bt[l] = interval after l letters, val[l] / stop[l] = outcome once decided.
    u64 tp[T20_MAX_E + 1], bt[T20_MAX_E + 1], val[T20_MAX_E + 1];
    bool stop[T20_MAX_E + 1];
    int dig[T20_MAX_E];
    int pw[T20_MAX_E];
    { int x = 1; for (int l = 0; l < E; l++) { pw[l] = x; x *= 3; } }
    tp[0] = t0; bt[0] = b0; val[0] = 0; stop[0] = false;
    for (int l = 0; l < E; l++) dig[l] = 0;
    int l = 0;                     // level being (re)computed
    for (;;) {
        // state after consuming digits dig[0..l] -> level l + 1
        for (; l < E; l++) {
            const int d = dig[l];
            if (stop[l]) { stop[l + 1] = true; val[l + 1] = val[l]; tp[l + 1] = tp[l]; bt[l + 1] = bt[l]; continue; }
            if (bt[l] - tp[l] == 1) { val[l + 1] = t20_entry(sa_at(ix, tp[l]), 1, 1 + l); stop[l + 1] = true; continue; }
            u64 t = tp[l], b = bt[l];
            lf_pair(ix, t, b, d);
            if (b <= t) { val[l + 1] = t20_entry(tp[l], bt[l] - tp[l], E + 1 + l); stop[l + 1] = true; }
            else { tp[l + 1] = t; bt[l + 1] = b; stop[l + 1] = false; val[l + 1] = 0; }
        }
        int idx = 0;
        for (int q = 0; q < E; q++) idx += dig[q] * pw[q];
        o[idx] = stop[E] ? val[E] : (bt[E] - tp[E] == 1 ? t20_entry(sa_at(ix, tp[E]), 1, 2 * E + 1) : t20_entry(tp[E], bt[E] - tp[E], 0));
        // next continuation: the deepest digit first (shares the longest prefix)
        int q = E - 1;
        while (q >= 0 && dig[q] == 2) { dig[q] = 0; q--; }
        if (q < 0) break;
        dig[q]++;
        l = q;
    }
}

// doubled 2-bit genome: d < G forward base, else complement of base 2G-1-d; LSB-first in u64 words
__global__ void k_build_gen2(RefIndexDev R, u64 G, u64 n_words, u64* out)
{
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    u64 v = 0;
    for (int j = 0; j < 32; j++) {
        const u64 d = w * 32 + j;
        int b = 0;
        if (d < G) b = (R.pac[d >> 2] >> (6 - 2 * (d & 3))) & 3;
        else if (d < 2 * G) { const u64 p = 2 * G - 1 - d; b = 3 - ((R.pac[p >> 2] >> (6 - 2 * (p & 3))) & 3); }
        v |= (u64)b << (2 * j);
    }
    out[w] = v;
}

// full SA from the sampled SA: LF-walk to a flagged row (bwt_get_sa_restrict_steps_more_than_3,
// bwt.h:2449-2560), done once per attach so that the mapping kernels never walk.
// (grid-stride: a launch may not exceed 2^32 threads, GRCh38 has 6.2 G rows)
__global__ void k_expand_sa(DevIndex ix, RefIndexDev R, u64 rows, u32* out, u64* out64)
{
  for (u64 row = (u64)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (u64)gridDim.x * blockDim.x) {
    u64 l = row, steps = 0, val = 0;
    if (l == ix.shapline) { if (out64) out64[row] = 0; else out[row] = 0; continue; }
    for (;;) {
        const u64 blk = (l >> 8) * 5, last = l & 255;
        const u64 w = R.sa_flag[blk + 1 + (last >> 6)];
        if ((w << (last & 63)) >> 63) {
            u64 rank = R.sa_flag[blk];
            for (u64 j = 0; j < (last >> 6); j++) rank += __popcll(R.sa_flag[blk + 1 + j]);
            if (last & 63) rank += __popcll(w >> (64 - (last & 63)));
            val = (u64)(R.sa[rank] & 0x3fffffffu) * 8 + steps;
            break;
        }
        const int c = bwt_sym(ix, l);
        l = lf_step(ix, l, c);
        steps++;
        if (l == ix.shapline) { val = steps; break; }
    }
    if (out64) out64[row] = val; else out[row] = (u32)val;
  }
}

// ---- the trigram rank table (DevIndex::occ3) ---------------------------------------------------------------------------------------------
// trigram of a row = the three letters in front of its suffix in extension order (index alphabet G0 T1 A2, C folded into T);
// 27 = none (the suffix starts less than three letters into the indexed text)
DEVI int row_trigram(const DevIndex& ix, u64 row)
{
    // the index is built over the REVERSED doubled text (a backward extension of the pattern is a step forward along the genome:
    // site = 2G - pos - ..., Schema.cpp:4657), so the letters in front of suffix p are the doubled-genome bases at 2G - p, + 1, + 2
    const u64 p = sa_at(ix, row);
    if (p < 3) return 27;
    const u64 q = ix.total - p;
    const int sh = 2 * (int)(q & 31);
    u64 w = ix.gen2[q >> 5] >> sh;
    if (sh > 58) w |= ix.gen2[(q >> 5) + 1] << (64 - sh);
    const int b1 = (int)(w & 3), b2 = (int)((w >> 2) & 3), b3 = (int)((w >> 4) & 3);            // first, second, third extension letter
    return ((0x46 >> (2 * b1)) & 3) + 3 * ((0x46 >> (2 * b2)) & 3) + 9 * ((0x46 >> (2 * b3)) & 3);
}
// one wave per block of 96 rows: the 27 bit planes by ballots, lane g keeps and stores trigram g's; .x = rows of the block that carry it
__global__ void __launch_bounds__(256)
k_occ3_planes(DevIndex ix, u64 rows, u64 nb, uint4* __restrict__ out)
{
    const u64 n_waves = ((u64)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (u64 blk = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; blk < nb; blk += n_waves) {
        int gA = 27, gB = 27;
        u64 r = blk * 96 + (u64)lane;
        if (r < rows) gA = row_trigram(ix, r);
        r += 64;
        if (lane < 32 && r < rows) gB = row_trigram(ix, r);
        u32 w0 = 0, w1 = 0, w2 = 0;
        for (int g = 0; g < 27; g++) {
            const unsigned long long mA = __ballot(gA == g), mB = __ballot(gB == g);
            if (lane == g) { w0 = (u32)mA; w1 = (u32)(mA >> 32); w2 = (u32)mB; }
        }
        if (lane < 27) out[(u64)lane * nb + blk] = make_uint4((u32)(__popc(w0) + __popc(w1) + __popc(w2)), w0, w1, w2);
    }
}
// .x of every block -> rows before the block: sums of chunks of OCC3_CHUNK blocks, a scan of the chunk sums by one thread per
// trigram, then the running count inside every chunk.  A trigram whose total does not fit 32 bits raises *overflow (the table is
// then not used: counts are 32-bit)
#define OCC3_CHUNK 512
__global__ void __launch_bounds__(256)
k_occ3_chunk_sums(const uint4* __restrict__ t, u64 nb, u64 n_chunks, u64* __restrict__ sums)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 27 * n_chunks) return;
    const u64 g = i / n_chunks, c = i - g * n_chunks;
    const u64 a = c * OCC3_CHUNK, e = a + OCC3_CHUNK < nb ? a + OCC3_CHUNK : nb;
    u64 s = 0;
    for (u64 b = a; b < e; b++) s += t[g * nb + b].x;
    sums[i] = s;
}
__global__ void k_occ3_chunk_scan(u64* __restrict__ sums, u64 n_chunks, u32* __restrict__ overflow)
{
    const int g = threadIdx.x;
    if (g >= 27) return;
    u64 run = 0;
    for (u64 c = 0; c < n_chunks; c++) { const u64 v = sums[(u64)g * n_chunks + c]; sums[(u64)g * n_chunks + c] = run; run += v; }
    if (run >= (1ull << 32)) *overflow = 1;
}
__global__ void __launch_bounds__(256)
k_occ3_apply(uint4* __restrict__ t, u64 nb, u64 n_chunks, const u64* __restrict__ sums)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 27 * n_chunks) return;
    const u64 g = i / n_chunks, c = i - g * n_chunks;
    const u64 a = c * OCC3_CHUNK, e = a + OCC3_CHUNK < nb ? a + OCC3_CHUNK : nb;
    u64 run = sums[i];
    for (u64 b = a; b < e; b++) { const u32 v = t[g * nb + b].x; t[g * nb + b].x = (u32)run; run += v; }
}
// c3[g] = LF_d3(LF_d2(LF_d1(0))): the first row of the suffixes that begin with the trigram (letters in text order d3 d2 d1)
__global__ void k_occ3_c3(DevIndex ix, u64* __restrict__ c3)
{
    const int g = threadIdx.x;
    if (g >= 27) return;
    const int d1 = g % 3, d2 = (g / 3) % 3, d3 = g / 9;
    c3[g] = lf_step(ix, lf_step(ix, lf_step(ix, 0, d1), d2), d3);
}
// the table against three single steps on pseudo-random rows: *bad counts the differences (attach refuses the table if any)
__global__ void __launch_bounds__(256)
k_occ3_check(DevIndex ix, u64 rows, u64 n, u32* __restrict__ bad)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 row = i < 200 ? (i < 100 ? i : rows - (i - 100)) : (i * 0x9E3779B97F4A7C15ull >> 11) % (rows + 1);
    const int g = (int)(i % 27);
    const int d1 = g % 3, d2 = (g / 3) % 3, d3 = g / 9;
    const u64 want = lf_step(ix, lf_step(ix, lf_step(ix, row, d1), d2), d3);
    u64 t = row, b = row;
    lf3_pair(ix, g, ix.c3[g], t, b);
    if (t != want) atomicAdd(bad, 1u);
}


// ================================================================================================
// scan (exclusive, u32 -> u64), two launches: tile sums, then every tile adds up the sums before it (L2 hits) and writes its part
// ================================================================================================
#define SCAN_BLOCK 256
#define SCAN_ITEMS 8            // per thread
// block-wide inclusive scan of one u64 per thread (wave shuffles + one LDS hop); returns the inclusive value, *total the block sum
DEVI u64 block_scan_incl(u64 v, u64* sh_waves, u64& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(v, d); if (lane >= d) v += t; }
    if (lane == 63) sh_waves[w] = v;
    __syncthreads();
    u64 add = 0, tot = 0;
    for (int i = 0; i < nw; i++) { const u64 x = sh_waves[i]; if (i < w) add += x; tot += x; }
    __syncthreads();
    total = tot;
    return v + add;
}
// eight consecutive u32 of a thread, two 16-byte loads when whole (the inputs are hipMalloc'ed: 16-byte aligned)
DEVI void scan_load8(const u32* in, u64 base, u64 n, u32 x[SCAN_ITEMS])
{
    if (base + SCAN_ITEMS <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(in + base), b = *reinterpret_cast<const uint4*>(in + base + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) x[j] = base + j < n ? in[base + j] : 0u;
    }
}
// nz: the values count as flags (x != 0)
// n_dev != nullptr: only the first min(n, *n_dev) entries exist (a count an earlier kernel of the same stream left in device
// memory: the launch is sized for the capacity n and the host never waits for the count)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_partial(const u32* in, u64 n, u64* block_sums, int nz, const u64* __restrict__ n_dev)
{
    __shared__ u64 sh[SCAN_BLOCK / 64];
    if (n_dev) { const u64 nd = *n_dev; if (nd < n) n = nd; }
    const u64 base = (u64)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + (u64)threadIdx.x * SCAN_ITEMS;
    u32 x[SCAN_ITEMS];
    scan_load8(in, base, n, x);
    if (nz) {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) x[j] = x[j] != 0;
    }
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += x[j];
    u64 total;
    (void)block_scan_incl(s, sh, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// list != nullptr: `in` holds 0/1 flags and the positions of the ones are written, in order, to list[]; the offsets
// themselves are not stored (the work lists of the seeding stages need nothing else)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_final(const u32* in, u64 n, const u64* __restrict__ block_sums, u64* out, u32* list, int nz, const u64* __restrict__ n_dev, u64* total)
{
    __shared__ u64 sh[SCAN_BLOCK / 64];
    if (n_dev) { const u64 nd = *n_dev; if (nd < n) n = nd; }
    const u64 base = (u64)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + (u64)threadIdx.x * SCAN_ITEMS;
    u32 x[SCAN_ITEMS];
    scan_load8(in, base, n, x);
    if (nz) {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) x[j] = x[j] != 0;
    }
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += x[j];
    // what the tiles before this one hold: 256 threads over blockIdx.x sums
    u64 before = 0;
    for (u32 i = threadIdx.x; i < blockIdx.x; i += SCAN_BLOCK) before += block_sums[i];
    u64 prefix;
    (void)block_scan_incl(before, sh, prefix);
    u64 tot;
    const u64 incl = block_scan_incl(s, sh, tot);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = prefix + tot;
    u64 run = incl - s + prefix;
    if (list) {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n && x[j]) { list[run] = (u32)(base + j); run += x[j]; }
        return;
    }
    if (base + SCAN_ITEMS <= n) {
        u64 o[SCAN_ITEMS];
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) { o[j] = run; run += x[j]; }
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(out + base);        // out is 16-byte aligned, base a multiple of 8
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j += 2) dst[j >> 1] = make_ulonglong2(o[j], o[j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n) { out[base + j] = run; run += x[j]; }
    }
    if (base <= n && n < base + SCAN_ITEMS) out[n] = run;                   // the thread that owns position n writes the total
    if (n % ((u64)SCAN_BLOCK * SCAN_ITEMS) == 0 && n && base + SCAN_ITEMS == n) out[n] = run;
}

// ================================================================================================
// K1-K5: seeding
// ================================================================================================
// The seeding state machine of Map_Single_Seq_end_to_end (Schema.cpp:24588-24898) / get_candidates
// (18172-18565) is data dependent per read: ~half of the reads leave after one short seed (exact-unique
// exit), a third need one long second seed (1-mismatch path), the rest run up to 25 more seeds.  One
// read per lane in ONE kernel makes every wave as slow as its slowest read.  It is therefore split
// by read class, with scan-compacted work lists in between (no host round-trip):
//   k_seed_first   all reads          first seed (count_backward_as_much_1_terminate)
//   k_seed_decide  all reads          exact-unique / exact-ambiguous exits, 1-mismatch detection
//   k_seed_second  1-mismatch reads   second seed (count_hash_table) + 1-mismatch exit
//   k_seed_extra   everything else    the remaining seeds
// The three search kernels share one engine: a wave owns a contiguous chunk of its work list and
// every lane that finishes an item immediately takes the next one of the chunk (wave-local counter,
// no global atomics), so the lanes of a wave keep stepping in lock-step through the LF loop -- the
// only hot code -- whatever the individual seed lengths are.
struct SeedHit { u64 hits, sp, ml; };

// search state of one lane: count_backward_as_much_1_terminate (bwt.h:2081-2209) or count_hash_table
// (bwt.h:1848-1952) over read[tm, L), advanced one backward-extension at a time.  In read coordinates
// the pattern bsSeq[0, L-tm) = reverse(read[tm, L)) with C->T, so the 16-mer key is the little-endian
// base-3 number of read[tm .. tm+15] and extension consumes read[tm+16], read[tm+17], ....
struct Search { u64 top, bot, ptop, pbot; int s, steps, tm; ReadCur cur; };

// eight ASCII characters -> eight 3-letter digits (G0 T1 A2, C folded into T), one per byte.  (c >> 1) & 3 is A0 C1 T2 G3, and
// the digit is 2 - popcount of that.  `bad` is non-zero in every byte that does not hold one of A C G T: the letter a byte would
// have to be is rebuilt from its bits 1-2 (0x41 | bits 1-2; T: ^ 0x11) and compared with it.
DEVI void swar_code3(u64 w, u64& digits, u64& bad)
{
    const u64 K01 = 0x0101010101010101ull;
    digits = 0x0202020202020202ull - ((w >> 1) & K01) - ((w >> 2) & K01);
    const u64 isT = (w >> 2) & ~(w >> 1) & K01;
    bad = w ^ (0x4141414141414141ull | (w & 0x0606060606060606ull)) ^ (isT | (isT << 4));
}
// four digits (one per byte of x) -> d0 + 3 d1 + 9 d2 + 27 d3 in one multiply
DEVI u32 base3_of4(u32 x) { return (x * 0x0103091Bu) >> 24; }

// returns true when the search has to be stepped; false when it is already decided (out filled)
// LOCATED (with FIXED): the caller finishes single-row intervals against the genome and takes them with the text position
// in S.top (bit 63 set) -- k_seed_second; without it the full-length search does not touch the 20-mer table.
template <bool FIXED, bool LOCATED = false>
DEVI bool search_begin(const DevIndex& ix, const char* rd, int L, int tm, Search& S, SeedHit& out, u32& n_hash)
{
    const int len = L - tm;
    out.hits = 0; out.sp = 0; out.ml = FIXED ? (u64)len : 0;
    if (len < (FIXED ? 17 : 18)) return false;
    // key = sum code(read[tm+u]) * 3^u over the 16 characters read[tm .. tm+15].  Every per-lane load is a request of its
    // own to the memory pipeline whether it hits or not (tools/gather_bench.hip: a row load costs 3/4 of a random gather), so the
    // 20 characters of the key and the cursor's first piece come from two aligned 16-byte loads (a third one when tm sits in
    // the last quarter of its piece) instead of three unaligned loads plus the cursor's own.
    const int a16 = tm & ~15, o = tm & 15;
    const uint4 B0 = *reinterpret_cast<const uint4*>(rd + a16), B1 = *reinterpret_cast<const uint4*>(rd + a16 + 16);
    uint4 B2 = make_uint4(0, 0, 0, 0);
    if (o + 16 + ix.t_e >= 32 && a16 + 32 < L) B2 = *reinterpret_cast<const uint4*>(rd + a16 + 32);
    const u64 q0 = ((u64)B0.y << 32) | B0.x, q1 = ((u64)B0.w << 32) | B0.z, q2 = ((u64)B1.y << 32) | B1.x,
              q3 = ((u64)B1.w << 32) | B1.z, q4 = ((u64)B2.y << 32) | B2.x;
    const int sh8 = (o & 7) * 8;
    auto funnel = [&](u64 lo, u64 hi) -> u64 { return sh8 ? (lo >> sh8) | (hi << (64 - sh8)) : lo; };
    const bool up = o >= 8;
    const u64 c0 = up ? q1 : q0, c1 = up ? q2 : q1, c2 = up ? q3 : q2, c3 = up ? q4 : q3;
    const u64 w0 = funnel(c0, c1), w1 = funnel(c1, c2);          // read[tm .. tm+7], read[tm+8 .. tm+15]
    u64 d0, v0, d1, v1;
    swar_code3(w0, d0, v0);
    swar_code3(w1, d1, v1);
    if ((v0 | v1) != 0) return false;                              // get_3_letter_hash_value returned -1 (bwt.h:309-332)
    // < 3^16: 32-bit arithmetic
    const u64 key = base3_of4((u32)d0) + 81u * base3_of4((u32)(d0 >> 32)) + 6561u * base3_of4((u32)d1) + 531441u * base3_of4((u32)(d1 >> 32));
    S.steps = len - 16; S.tm = tm;
    const int E = ix.t_e;
    if ((!FIXED || LOCATED) && ix.t20 && len >= 16 + E) {
        // the 16-mer lookup and the first E extensions in one table read
        u64 d2, v2;
        const u64 cmask = E == 5 ? 0xffffffffffull : 0xffffffffull;
        swar_code3(funnel(c2, c3) & cmask, d2, v2);                  // read[tm+16 .. tm+15+E]
        if ((v2 & cmask) == 0) {
            const u64 code = (u64)base3_of4((u32)d2) + (E == 5 ? 81ull * ((d2 >> 32) & 0xffull) : 0ull);
            const u64 v = ix.t20[key * t20_width(E) + code];
            const int tag = (int)(v >> 60);
            if (tag != 14) {
                n_hash++;
                const u64 row = v & ((1ull << 36) - 1), hits = (v >> 36) & ((1ull << 24) - 1);
                if (tag == 15) return false;                                           // hits 0, match length 0
                if (FIXED) {
                    // count_hash_table goes through the whole pattern: a missing letter is 0 hits; a single row carries on --
                    // the caller finishes it against the genome (k_seed_second), from the text position the table holds
                    if (tag > E && tag <= 2 * E) return false;
                    if (tag != 0) {
                        S.top = row | (1ull << 63); S.bot = S.top + 1; S.ptop = ~0ull; S.pbot = ~0ull; S.s = tag == 2 * E + 1 ? E : tag - 1;
                        return true;
                    }
                } else {
                    if (tag >= 1 && tag <= E) { out.ml = (u64)(15 + tag); out.sp = row | (1ull << 63); out.hits = 1; return false; }     // located (bit 63): text position
                    if (tag == 2 * E + 1) { out.ml = (u64)(16 + E); out.sp = row | (1ull << 63); out.hits = 1; return false; }
                    if (tag > E) { out.ml = (u64)(15 - E + tag); out.sp = row; out.hits = hits; return false; }
                }
                S.top = row; S.bot = row + hits; S.ptop = ~0ull; S.pbot = ~0ull; S.s = E;
                if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = hits; return false; }
                S.cur.seek_with(rd, tm + 16 + E, L, o + 16 + E < 32 ? B1 : B2);
                return true;
            }
        }
        // a letter outside the alphabet among them, or an oversized interval: the 16-mer path
    }
    hash_lookup(ix, key, S.top, S.bot);
    n_hash++;
    if (S.bot <= S.top) return false;
    S.ptop = ~0ull; S.pbot = ~0ull; S.s = 0;
    S.cur.seek_with(rd, tm + 16, L, B1);
    return true;
}

// one loop iteration of the reference; returns true when the search is finished (out filled)
template <bool FIXED>
DEVI bool search_step(const DevIndex& ix, const char* rd, int L, Search& S, SeedHit& out, u32& n_ext)
{
    const int len = L - S.tm;
    if (!FIXED) {
        S.ptop = S.top; S.pbot = S.bot;
        if (S.bot - S.top == 1) { out.ml = 16 + S.s; out.sp = S.top; out.hits = 1; return true; }
        const int d = code3(S.cur.next());          // read[tm + 16 + s]
        if (d > 2) { out.ml = 16 + S.s; out.sp = S.ptop; out.hits = S.pbot - S.ptop; return true; }
        lf_pair(ix, S.top, S.bot, d);
        n_ext++;
        if (S.bot <= S.top) { out.ml = 16 + S.s; out.sp = S.ptop; out.hits = S.pbot - S.ptop; return true; }
        S.s++;
        if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = S.bot - S.top; return true; }
        return false;
    } else {
        out.ml = (u64)len;
        const int d = code3(S.cur.next());          // read[tm + 16 + s]
        if (d > 2) { out.hits = 0; out.sp = 0; return true; }
        lf_pair(ix, S.top, S.bot, d);
        n_ext++;
        if (S.bot <= S.top) { out.hits = 0; out.sp = S.top; return true; }      // the remaining iterations only break
        S.s++;
        if (S.s == S.steps) { out.sp = S.top; out.hits = S.bot - S.top; return true; }
        return false;
    }
}

// ---- the same two functions over a packed row (PackedRows) ----------------------------------------------------------------
struct SearchP { u64 top, bot; int s, steps, tm, kg; PCur cur; };      // kg: steps in a row that kept most of the interval (< 0: three-letter steps are off for this seed)

// four 2-bit digits (d0 in bits 0-1) -> d0 + 3 d1 + 9 d2 + 27 d3
DEVI u32 base3_of4x2(u32 v8)
{
    u32 t = (v8 | (v8 << 12)) & 0x000f000fu;
    t = (t | (t << 6)) & 0x03030303u;
    return base3_of4(t);
}

template <bool FIXED, bool LOCATED = false>
DEVI bool search_begin_p(const DevIndex& ix, const u64* row, int W, bool dirty, int L, int tm, SearchP& S, SeedHit& out, u32& n_hash)
{
    const int len = L - tm;
    out.hits = 0; out.sp = 0; out.ml = FIXED ? (u64)len : 0;
    if (len < (FIXED ? 17 : 18)) return false;
    // the 64 bases from the word that holds read[tm]: the 16 of the key, the table's look-ahead and the cursor's first piece in
    // ONE 16-byte request
    const u64 x = prow_bases32(row, tm);                              // bases tm .. tm+31
    const u32 mbits = dirty ? prow_mask32(row, W, tm) : 0u;
    if (mbits & 0xffffu) return false;                               // get_3_letter_hash_value returned -1 (bwt.h:309-332)
    // 3-letter digits in the 2-bit fields: A (00) -> 2, C / T (low bit) -> 1, G -> 0
    const u64 D = ((((~x) & ((~x) >> 1)) & PK_EVEN) << 1) | (x & PK_EVEN);
    const u32 d32 = (u32)D;
    const u64 key = base3_of4x2(d32 & 0xffu) + 81u * base3_of4x2((d32 >> 8) & 0xffu) + 6561u * base3_of4x2((d32 >> 16) & 0xffu) +
                    531441u * base3_of4x2(d32 >> 24);
    S.steps = len - 16; S.tm = tm; S.kg = 0;
    S.cur.row = row; S.cur.W = W; S.cur.dirty = dirty;
    const int E = ix.t_e;
    if ((!FIXED || LOCATED) && ix.t20 && len >= 16 + E) {
        // the 16-mer lookup and the first E extensions in one table read
        if (((mbits >> 16) & ((1u << E) - 1)) == 0) {
            const u32 dE = (u32)(D >> 32);
            const u64 code = (u64)base3_of4x2(dE & 0xffu) + (E == 5 ? 81ull * ((dE >> 8) & 3u) : 0ull);
            const u64 v = ix.t20[key * t20_width(E) + code];
            const int tag = (int)(v >> 60);
            if (tag != 14) {
                n_hash++;
                const u64 rowv = v & ((1ull << 36) - 1), hits = (v >> 36) & ((1ull << 24) - 1);
                if (tag == 15) return false;                                           // hits 0, match length 0
                if (FIXED) {
                    if (tag > E && tag <= 2 * E) return false;
                    if (tag != 0) {
                        S.top = rowv | (1ull << 63); S.bot = S.top + 1; S.s = tag == 2 * E + 1 ? E : tag - 1;
                        return true;
                    }
                } else {
                    if (tag >= 1 && tag <= E) { out.ml = (u64)(15 + tag); out.sp = rowv | (1ull << 63); out.hits = 1; return false; }
                    if (tag == 2 * E + 1) { out.ml = (u64)(16 + E); out.sp = rowv | (1ull << 63); out.hits = 1; return false; }
                    if (tag > E) { out.ml = (u64)(15 - E + tag); out.sp = rowv; out.hits = hits; return false; }
                }
                S.top = rowv; S.bot = rowv + hits; S.s = E;
                if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = hits; return false; }
                S.cur.pos = tm + 16 + E; S.cur.buf = x >> (2 * (16 + E)); S.cur.have = 16 - E;
                return true;
            }
        }
    }
    hash_lookup(ix, key, S.top, S.bot);
    n_hash++;
    if (S.bot <= S.top) return false;
    S.s = 0;
    S.cur.pos = tm + 16; S.cur.buf = x >> 32; S.cur.have = 16;
    return true;
}

// Three letters at once (lf3_pair) while the interval shrinks slowly -- a read inside a repeat family walks the index for most of
// its length with hundreds of rows, one dependent gather pair per letter.  Exact by construction: a jump is TAKEN only when the
// interval behind it still has two rows or more, so none of the reference's stop conditions (one row left, letter absent, letter
// outside the alphabet -- bwt.h:2081-2209, 1848-1952) fell inside it; otherwise it is dropped, the three letters are stepped one by
// one as before, and the seed makes no further attempt.  Counted as three extensions (the reference's events).
template <bool FIXED, bool KG = false>
DEVI bool search_step_p(const DevIndex& ix, int L, SearchP& S, SeedHit& out, u32& n_ext, const u64* c3 = nullptr, u32* n_jump = nullptr)
{
    const int len = L - S.tm;
    const u64 ptop = S.top, pbot = S.bot;
    if (!FIXED) {
        if (pbot - ptop == 1) { out.ml = 16 + S.s; out.sp = ptop; out.hits = 1; return true; }
    } else out.ml = (u64)len;
    const u64 before = pbot - ptop;
    // which kind of step this lane takes: three letters (trigram g) or one (digit d).  Both kinds then share ONE gather pair -- a
    // jump tried in a branch of its own made every wave with a jumping lane wait out two memory round trips per iteration
    bool jump = false; int g = 0, d = 0;
    if (KG && S.kg >= (before >= 8 ? 1 : 2) && S.steps - S.s >= 3 && S.cur.have >= 3 && before >= 2) {
        jump = true;                                       // none of the three letters outside ACGT (their mask bits inside one word)
        if (S.cur.dirty) { const int o = S.cur.pos & 63; jump = o <= 61 && ((S.cur.row[S.cur.W + (S.cur.pos >> 6)] >> o) & 7ull) == 0; }
        const u32 b6 = (u32)S.cur.buf & 63u;
        g = ((0x46 >> (2 * (b6 & 3u))) & 3) + 3 * ((0x46 >> (2 * ((b6 >> 2) & 3u))) & 3) + 9 * ((0x46 >> (2 * (b6 >> 4))) & 3);
    }
    if (!jump) {
        d = S.cur.next3();
        if (d > 2) {
            if (!FIXED) { out.ml = 16 + S.s; out.sp = ptop; out.hits = pbot - ptop; } else { out.hits = 0; out.sp = 0; }
            return true;
        }
    }
    const u64 lt = ptop - (ptop > ix.shapline ? 1 : 0), lb = pbot - (pbot > ix.shapline ? 1 : 0);       // (single steps: '$' row removed)
    // rows fit 36 bits (checked at attach), so row >> 5 fits 32: the division by 96 is a 32-bit one by 3
    const u32 it = jump ? (u32)(ptop >> 5) / 3u : (u32)(lt >> 5), ib = jump ? (u32)(pbot >> 5) / 3u : (u32)(lb >> 5);
    const u32 rt = jump ? (u32)(ptop - (u64)it * 96) : (u32)lt & 31u, rb = jump ? (u32)(pbot - (u64)ib * 96) : (u32)lb & 31u;
    const uint4* base = jump ? ix.occ3 + (u64)g * ix.nb3 : ix.occ;
    const uint4 ht = base[it];
    uint4 hb = ht;
    if (ib != it) hb = base[ib];
    if (jump) {
        const u64 c3g = c3[g];
        const u64 t2 = c3g + occ3_in_block(ht, rt), b2 = c3g + occ3_in_block(hb, rb);
        if (b2 > t2 && b2 - t2 >= 2) {
            S.top = t2; S.bot = b2; S.s += 3; n_ext += 3;
            if (n_jump) (*n_jump)++;
            S.cur.buf >>= 6; S.cur.have -= 3; S.cur.pos += 3;
            if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = S.bot - S.top; return true; }
            return false;
        }
        S.kg = -(1 << 20);                                 // dropped: nothing was consumed, the next calls step letter by letter
        return false;
    }
    {
        const u32 mt = rt ? (~0u << (32 - rt)) : 0u, mb = rb ? (~0u << (32 - rb)) : 0u;
        u64 tT = (u64)ht.x + __popc(ht.z & mt), tA = (u64)ht.y + __popc(ht.w & mt);
        u64 bT = (u64)hb.x + __popc(hb.z & mb), bA = (u64)hb.y + __popc(hb.w & mb);
        if (ix.sup_shift) { super_add(ix, lt, tT, tA); super_add(ix, lb, bT, bA); }
        const u64 ct = d == 1 ? tT : (d == 2 ? tA : lt - tT - tA);
        const u64 cb = d == 1 ? bT : (d == 2 ? bA : lb - bT - bA);
        S.top = ix.C[d] + ct; S.bot = ix.C[d] + cb;
    }
    n_ext++;
    if (S.bot <= S.top) {
        if (!FIXED) { out.ml = 16 + S.s; out.sp = ptop; out.hits = pbot - ptop; } else { out.hits = 0; out.sp = S.top; }      // (FIXED: the remaining iterations only break)
        return true;
    }
    if (KG) S.kg = 2 * (S.bot - S.top) > before ? S.kg + 1 : (S.kg < 0 ? S.kg : 0);
    S.s++;
    if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = S.bot - S.top; return true; }
    return false;
}

// does read[q0 .. L) equal the text at doubled coordinate site0 + (q - tm) in the index alphabet?  (k_seed_second's single-row
// shortcut, on a packed row: 32 bases per step)
DEVI bool rest_matches_3l(const DevIndex& ix, const u64* row, int W, bool dirty, int L, int tm, int start, u64 site)
{
    for (int q = start & ~31; q < L; q += 32) {
        const u64 rb = row[q >> 5];
        const u64 d = site + (u64)(q - tm);                  // doubled coordinate facing read[q] (u64 wrap = out of range)
        u64 mm;
        if (d + 32 <= ix.total) mm = mism_3l(rb, gen_bases32(ix, d));
        else {
            // runs off the end of the text: '$' never matches
            mm = 0;
            for (int j = 0; j < 32; j++) {
                const u64 dj = d + (u64)j;
                const int rc = (int)((rb >> (2 * j)) & 3);
                bool eq = false;
                if (dj < ix.total) { const int g = gbase(ix, dj); eq = (rc == g) || ((rc & 1) && (g & 1)); }
                if (!eq) mm |= 1ull << (2 * j);
            }
        }
        if (dirty) mm |= spread32((u32)((row[W + (q >> 6)] >> (q & 63)) & 0xffffffffull));
        if (mm & field_range(start - q, L - q)) return false;
    }
    return true;
}

// determine_seed_offset_unmatch (Schema.h:1506-1531)
DEVI int seed_offset_unmatch(int L, int pre, const char* rd, int step)
{
    if (L - pre < 18 || L - pre < step) return L;
    const int ret = pre + step;
    for (int i = 0; i < step; i++, pre++) if (rd[pre] == 'N') return pre + 1;
    return ret;
}
// ... on packed rows: `mask` is the not-ACGT bit plane of the row; the ASCII row is asked only where a bit is set (the
// paired-end rows keep their ASCII text only in the 16-byte pieces that hold such a character, k_pe_prepare)
DEVI int seed_offset_unmatch_p(int L, int pre, const char* rd, int step, const u64* mask)
{
    if (L - pre < 18 || L - pre < step) return L;
    const int ret = pre + step;
    for (int i = 0; i < step; i++, pre++) if (((mask[pre >> 6] >> (pre & 63)) & 1) && rd[pre] == 'N') return pre + 1;
    return ret;
}

// per-read state carried between the seeding kernels
struct SeedCarry {
    u64* sp0; u32* hits0; u16* ml0;        // first seed result
    u16* tm; u8* seed_id; u32* clen; u16* first_ml;
    u32* flag_c; u32* flag_d;              // needs k_seed_second / k_seed_extra (scan inputs)
    u64* off_c; u64* off_d;                // exclusive scans
    u32* list_c; u32* list_d;              // compacted read lists
};

DEVI void seed_record(SeedRec* my, int& ns, u64& ncand, u64 sp, u64 hits, u64 len, u64 off)
{
    my[ns].sp = sp; my[ns].hits = (u32)hits; my[ns].len = (u16)len; my[ns].off = (u16)off; ns++;
    ncand += hits;
}

DEVI void seed_finish(const ReadState& st, long r, int verdict, int ns, u64 ncand, int pe_mode)
{
    st.verdict[r] = (u8)verdict;
    st.n_seeds[r] = (u8)ns;
    st.n_cand[r] = verdict == 3 ? (u32)ncand : (pe_mode ? (verdict == 4 ? (u32)ncand : (verdict == 1 || verdict == 2) ? 1u : 0u) : 0u);
}

#define SEED_BATCH 16             // pending lanes that trigger a transition batch
#define SEED_CHUNK 256            // items per wave (upper bound)
#define SEED_CHUNK_MIN 64
// items per wave for a work list of `total` items: enough waves to fill the chip a few times over (the second / extra
// lists hold 10-30 % of the batch; with 256-item chunks they gave fewer waves than the 8192 wave slots of the chip).
// (Measured and rejected, round 2: persistent waves -- exactly as many as the chip holds -- walking the list in strided 64-item
// chunks.  The per-wave timeline (BMBS_WAVELOG) then shows every wave slot occupied from start to end instead of 70-87 %, and
// the kernels get SLOWER (k_seed_first 1.95 -> 2.40 ms, k_seed_extra 3.73 -> 3.85 ms): the engine is bound by the request rate of
// the memory system, not by resident waves, and the long-lived waves end 20 % apart.  Chunks from a global atomic counter balance
// perfectly and cost 4.5x: ~60 k returning device-scope atomics on one word take several ms on this chip.)
DEVI long seed_chunk(long total, int target_waves)
{
    long c = (total + target_waves - 1) / target_waves;
    return c < SEED_CHUNK_MIN ? SEED_CHUNK_MIN : (c > SEED_CHUNK ? SEED_CHUNK : c);
}

struct LaneCounters { u32 n_hash, n_ext, n_sa, n_ung, n_jump; };      // n_jump: three-letter steps taken (each also counts three in n_ext)
// totals in counters[0,1,2,5]; per-kernel copies in counters[16 + 4*kid ..] (kid 0 first, 1 second, 2 extra)
DEVI void flush_counters(unsigned long long* counters, const LaneCounters& c, int kid)
{
    // 64-thread blocks: one wave; reduce with shuffles, one atomic per wave and counter
    u32 a = c.n_hash, b = c.n_ext, d = c.n_sa, e = c.n_ung, j = c.n_jump;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); d += __shfl_down(d, o); e += __shfl_down(e, o); j += __shfl_down(j, o); }
    if ((threadIdx.x & 63) == 0 && counters) {
        counters = SHARD(counters);
        if (j) atomicAdd(&counters[8], (unsigned long long)j);
        unsigned long long* k = counters + 16 + 4 * kid;
        if (a) { atomicAdd(&counters[0], (unsigned long long)a); atomicAdd(&k[0], (unsigned long long)a); }
        if (b) { atomicAdd(&counters[1], (unsigned long long)b); atomicAdd(&k[1], (unsigned long long)b); }
        if (d) { atomicAdd(&counters[2], (unsigned long long)d); atomicAdd(&k[2], (unsigned long long)d); }
        if (e) { atomicAdd(&counters[5], (unsigned long long)e); atomicAdd(&k[3], (unsigned long long)e); }
    }
}

// five stats values of a lane -> one LDS atomic per wave and value (256 lanes hammering five LDS words with 64-bit atomics was
// 40 % of k_finalize's issue time)
DEVI void wave_stats_add(unsigned long long* sh, u32 v0, u32 v1, u32 v2, u32 v3, u32 v4)
{
    for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_down(v0, o); v1 += __shfl_down(v1, o); v2 += __shfl_down(v2, o); v3 += __shfl_down(v3, o); v4 += __shfl_down(v4, o); }
    if ((threadIdx.x & 63) == 0) {
        if (v0) atomicAdd(&sh[0], (unsigned long long)v0);
        if (v1) atomicAdd(&sh[1], (unsigned long long)v1);
        if (v2) atomicAdd(&sh[2], (unsigned long long)v2);
        if (v3) atomicAdd(&sh[3], (unsigned long long)v3);
        if (v4) atomicAdd(&sh[4], (unsigned long long)v4);
    }
}

// ---- first seed of every read ------------------------------------------------------------------
template <bool PACKED, bool KG = false>
__global__ void __launch_bounds__(64)
k_seed_first(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, long n, SeedCarry sc,
             unsigned long long* __restrict__ counters)
{
    __shared__ u64 s_c3[KG ? 27 : 1];
    const u64* c3 = KG ? kgram_c3(ix, s_c3) : nullptr;
    const WaveLogT wl_t = wavelog_begin();
    int L = gm.L;                                     // length of the lane's current read
    const long chunk_begin = (long)blockIdx.x * SEED_CHUNK;
    const long chunk_end = chunk_begin + SEED_CHUNK < n ? chunk_begin + SEED_CHUNK : n;
    long next = chunk_begin;
    LaneCounters lc = {0, 0, 0, 0, 0};
    bool active = false;
    long r = 0;
    const char* rd = seq;
    typename std::conditional<PACKED, SearchP, Search>::type S; SeedHit h;
    // lanes whose search ended wait (`pending`) until SEED_BATCH of them can run the divergent
    // store / refill / hash-lookup code together: one straggler must not stall 63 stepping lanes
    bool pending = true, have = false;
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (__popcll(pm) >= SEED_BATCH || !__any(active)) {
            if (pm == 0) break;
            const int rank = __popcll(pm & ((1ull << (threadIdx.x & 63)) - 1));
            const long it = next + rank;
            next += __popcll(pm);
            if (pending) {
                if (have) { sc.sp0[r] = h.sp; sc.hits0[r] = (u32)h.hits; sc.ml0[r] = (u16)h.ml; have = false; }
                if (it < chunk_end) {
                    r = it; have = true; L = gm.rl(r);
                    bool go;
                    if constexpr (PACKED) go = search_begin_p<false>(ix, pr.base + (size_t)r * pr.pwords, pr.W, pr.dirty[r] != 0, L, 0, S, h, lc.n_hash);
                    else { rd = seq + (size_t)r * stride; go = search_begin<false>(ix, rd, L, 0, S, h, lc.n_hash); }
                    if (go) { active = true; pending = false; }
                    // else: decided at once; stays pending, stored at the next batch
                } else pending = false;
            }
            if (!__any(active) && !__any(pending)) break;
            continue;
        }
#ifdef BMBS_UTIL
        if ((threadIdx.x & 63) == 0) atomicAdd(&counters[8], 1ull);
        if (active) atomicAdd(&counters[9], 1ull);
#endif
        if (active) {
            bool fin;
            if constexpr (PACKED) fin = search_step_p<false, KG>(ix, L, S, h, lc.n_ext, c3, &lc.n_jump); else fin = search_step<false>(ix, rd, L, S, h, lc.n_ext);
            if (fin) { active = false; pending = true; }
        }
    }
    flush_counters(counters, lc, 0);
    wavelog_end(wl_t, 3);
}

// ---- exits after the first seed (Schema.cpp:24599-24727 / 18225-18330) ---------------------------
template <bool USE_LDS, bool VEC8>
__global__ void
k_seed_decide(DevIndex ix, const char* __restrict__ seq, ReadGeom gm, int stride, long n, int seed_len, int pe_mode, ReadState st,
              SeedCarry sc, unsigned long long* __restrict__ counters)
{
    __shared__ unsigned int shc[2];
    if (threadIdx.x < 2) shc[threadIdx.x] = 0;
    __syncthreads();
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    u32 n_sa = 0, n_ung = 0;
    // the block's 256 reads are contiguous in memory: copy them into LDS with fully coalesced 16-byte loads
    // (rows re-padded by 8 bytes against bank conflicts); each lane then scans ITS row out of LDS instead of
    // issuing 64-cache-line gathers per instruction
    extern __shared__ __align__(16) char lds_rows[];
    const int lstride = stride + 8;
    if (USE_LDS) {
        const long row0 = (long)blockIdx.x * blockDim.x;
        const long rows = n - row0 < (long)blockDim.x ? n - row0 : (long)blockDim.x;
        const int per_row = stride / 16;
        const int total16 = (int)rows * per_row;
        const uint4* src = reinterpret_cast<const uint4*>(seq + (size_t)row0 * stride);
        // piece q = (row rr, column cc); q advances by the block size, so (rr, cc) advance by its quotient and remainder -- one
        // division per thread instead of one (64-bit) per piece
        const int dq = (int)blockDim.x / per_row, dr = (int)blockDim.x - dq * per_row;
        int rr = (int)threadIdx.x / per_row, cc = (int)threadIdx.x - rr * per_row;
        for (int q = threadIdx.x; q < total16; q += blockDim.x) {
            const uint4 v = src[q];
            u64* dst = reinterpret_cast<u64*>(lds_rows + rr * lstride + cc * 16);
            dst[0] = ((u64)v.y << 32) | v.x;
            dst[1] = ((u64)v.w << 32) | v.z;
            rr += dq; cc += dr;
            if (cc >= per_row) { cc -= per_row; rr++; }
        }
    }
    __syncthreads();
    if (r < n) {
        const char* rd = USE_LDS ? lds_rows + (size_t)threadIdx.x * lstride : seq + (size_t)r * stride;
        const int L = gm.rl(r);
        int firstC = L;
        for (int i = 0; i < L; i += 8) {
            // lowest byte equal to 'C' in this 8-byte word (exact for the lowest-order zero byte)
            const u64 x = *reinterpret_cast<const u64*>(rd + i) ^ 0x4343434343434343ull;
            const u64 z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
            if (z) { const int j = i + (__ffsll((long long)z) - 1) / 8; if (j < L) firstC = j; break; }
        }
        SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        int ns = 0;
        u64 ncand = 0, clen = 0;
        const int max_seed = L / 10 == 0 ? 25 : (L / 10 - 1 > 25 ? 25 : L / 10 - 1);
        int verdict = 0, multi = 0, get_error = -1, tm = 0, seed_id = 0;
        u64 mm_site = 0, c0 = 0, first_ml = 0;
        const u64 max_hits = 1000;
        bool done = false;
        u32 fc = 0, fd = 0;
        if (seed_id < max_seed && tm < L) {
            const u64 hits = sc.hits0[r], sp = sc.sp0[r];
            u64 ml = sc.ml0[r];
            first_ml = ml;
            if (hits == 1) {
                // try_process_unique_mismatch_end_to_end (Schema.cpp:15164-15282)
                u64 p;
                if (sp >> 63) p = sp & ~(1ull << 63);          // the 20-mer table had the text position
                else { p = sa_at(ix, sp); n_sa++; }
                const u64 loc = ix.total - p - ml;
                seed_record(my, ns, ncand, sp, 1, ml, 0);
                c0 = loc; clen = 1;
                int error = 0;
                if (ml > (u64)firstC) ml = (u64)firstC;
                if (ml != (u64)L) {
                    const int need = L - (int)ml;
                    const u64 start = loc + ml;
                    n_ung++;
                    if (VEC8) {
                        if (!window_valid(ix, start, (u64)need, loc < ix.G)) {
                            // all-zero window: every position mismatches; the first sets ml = read_i (= ml), the second stops
                            error = need >= 2 ? 2 : 1;
                        } else {
                            // read position q faces doubled coordinate loc + q; 16 positions per step
                            const int ml0 = (int)ml;
                            // window bases cached in registers, 64 per 16-byte global load, the next block already on its way
                            // (every per-lane load is a request of its own: 3 wide loads instead of 6 narrow ones per read)
                            const uint4* g4 = reinterpret_cast<const uint4*>(ix.gen2);
                            u64 gidx = (loc + (u64)(ml0 & ~15)) >> 6;
                            uint4 cb = g4[gidx], nb = g4[gidx + 1];
                            for (int p = ml0 & ~15; p < L && error < 2; p += 16) {
                                const u64 r0 = *reinterpret_cast<const u64*>(rd + p), r1 = *reinterpret_cast<const u64*>(rd + p + 8);
                                const u64 d = loc + (u64)p;
                                if ((d >> 6) != gidx) { gidx = d >> 6; cb = nb; nb = g4[gidx + 1]; }
                                const int o = (int)(d & 63) * 2;
                                const u64 c0 = ((u64)cb.y << 32) | cb.x, c1 = ((u64)cb.w << 32) | cb.z, n0 = ((u64)nb.y << 32) | nb.x;
                                const u64 lo64 = o < 64 ? c0 : c1, hi64 = o < 64 ? c1 : n0;
                                const int sh = o & 63;
                                u64 w = lo64 >> sh;
                                if (sh > 32) w |= hi64 << (64 - sh);
                                const u32 w32 = (u32)w;
                                u64 m0 = mism8(r0, w32 & 0xffffu), m1 = mism8(r1, w32 >> 16);
                                if (p < ml0) {                                   // positions before ml0 are matched already
                                    const int lo = ml0 - p;
                                    if (lo >= 8) { m0 = 0; m1 &= ~((1ull << (8 * (lo - 8))) - 1); } else m0 &= ~((1ull << (8 * lo)) - 1);
                                }
                                if (p + 16 > L) {                                // the read ends inside this piece
                                    const int hi = L - p;
                                    if (hi <= 8) { m1 = 0; if (hi < 8) m0 &= (1ull << (8 * hi)) - 1; } else m1 &= (1ull << (8 * (hi - 8))) - 1;
                                }
                                if (m0 | m1) {
                                    const int cnt = __popcll(m0) + __popcll(m1);
                                    if (error == 0) {
                                        ml = (u64)(m0 ? p + (__ffsll((long long)m0) - 1) / 8 : p + 8 + (__ffsll((long long)m1) - 1) / 8);
                                        error = cnt >= 2 ? 2 : 1;
                                    } else error = 2;
                                }
                            }
                        }
                    } else {
                        WinReader wr; wr.init(ix, start, window_valid(ix, start, (u64)need, loc < ix.G));
                        int read_i = (int)ml;
                        ReadCur rc; rc.seek(rd, read_i, L);
                        for (int i = 0; i < need; i++) {
                            const char a = rc.next();
                            const int b = wr.next();                 // 4 when the window leaves the strand: never equal
                            if (!(code4(a) == b || (a == 'T' && b == 1))) { error++; if (error == 1) ml = (u64)read_i; else break; }
                            read_i++;
                        }
                    }
                }
                get_error = error;
                if (error == 0) { verdict = 1; st.exit_site[r] = loc; done = true; }
            }
            if (!done) {
                mm_site = ml;
                if (!pe_mode) {
                    if (ml == (u64)L && hits > 1) {
                        multi = 1;
                        if (firstC == L) { verdict = 4; done = true; }      // exact, ambiguous, no C in the read
                    }
                } else if (ml == (u64)L && hits > 1 && hits <= 10000) {
                    // get_candidates (Schema.cpp:18260-18290): every exact hit becomes a verified candidate
                    multi = 1;
                    if (firstC == L) { seed_record(my, ns, ncand, sp, hits, ml, 0); verdict = 4; done = true; }
                }
            }
            if (!done) {
                if (hits == 1) { /* recorded */ }
                else if (ml >= (u64)seed_len && hits <= max_hits) { if (hits != 0) { seed_record(my, ns, ncand, sp, hits, ml, (u64)tm); clen += hits; } }
                if (ml == 0) tm = seed_offset_unmatch(L, tm, rd, 8); else tm = tm + (int)(ml / 2);
                seed_id++;
            }
        }
        st.multi[r] = (u8)multi;
        // bit 15: the read character there is 'N' (L <= 1000).  k_finalize prices the 1-mismatch exit with it and need not touch the
        // read row again (one sector per such read)
        st.mm_site[r] = (u16)(mm_site | ((mm_site < (u64)L && rd[mm_site] == 'N') ? 0x8000u : 0u));
        if (done) seed_finish(st, r, verdict, ns, ncand, pe_mode);
        else {
            st.exit_site[r] = c0;
            st.n_seeds[r] = (u8)ns; st.n_cand[r] = (u32)ncand;
            sc.tm[r] = (u16)tm; sc.seed_id[r] = (u8)seed_id; sc.clen[r] = (u32)clen; sc.first_ml[r] = (u16)first_ml;
            // 1-mismatch first seed: second seed over the rest of the read (Schema.cpp:24734-24801)
            if (get_error == 1 && L - (int)first_ml >= 17) fc = 1; else fd = 1;
        }
        sc.flag_c[r] = fc; sc.flag_d[r] = fd;
    }
    if (counters) {
        atomicAdd(&shc[0], n_sa); atomicAdd(&shc[1], n_ung);
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long* cs = SHARD(counters); atomicAdd(&cs[2], (unsigned long long)shc[0]); atomicAdd(&cs[5], (unsigned long long)shc[1]); }
    }
}

// ---- the same kernel over packed rows: 64 bytes per read through LDS instead of 160, comparisons by whole-word XOR -------------
__global__ void __launch_bounds__(64)
k_seed_decide_p(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, long n, int seed_len, int pe_mode,
                ReadState st, SeedCarry sc, unsigned long long* __restrict__ counters)
{
    __shared__ unsigned int shc[2];
    if (threadIdx.x < 2) shc[threadIdx.x] = 0;
    extern __shared__ __align__(16) char lds_prows[];
    const int lw = pr.pwords + 1;                                   // row stride in words, odd against bank conflicts
    u64* lrows = reinterpret_cast<u64*>(lds_prows);
    const long row0 = (long)blockIdx.x * blockDim.x;
    const long rows = n - row0 < (long)blockDim.x ? n - row0 : (long)blockDim.x;
    {
        // the block's rows are contiguous in memory: copy them with coalesced 8-byte loads
        const u64* src = pr.base + (size_t)row0 * pr.pwords;
        const int total = (int)rows * pr.pwords;
        for (int q = threadIdx.x; q < total; q += blockDim.x) { const int rr = q / pr.pwords, cc = q - rr * pr.pwords; lrows[rr * lw + cc] = src[q]; }
    }
    __syncthreads();
    const long r = row0 + threadIdx.x;
    u32 n_sa = 0, n_ung = 0;
    if (r < n) {
        const u64* row = lrows + (size_t)threadIdx.x * lw;
        const int W = pr.W;
        const bool dirty = pr.dirty[r] != 0;
        const int L = gm.rl(r);
        // first 'C' of the read (code 01)
        int firstC = L;
        for (int w = 0; w * 32 < L; w++) {
            const u64 x = row[w];
            const u64 z = x & ~(x >> 1) & PK_EVEN;
            if (z) { const int j = w * 32 + (__ffsll((long long)z) - 1) / 2; if (j < L) firstC = j; break; }
        }
        SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        int ns = 0;
        u64 ncand = 0, clen = 0;
        const int max_seed = L / 10 == 0 ? 25 : (L / 10 - 1 > 25 ? 25 : L / 10 - 1);
        int verdict = 0, multi = 0, get_error = -1, tm = 0, seed_id = 0;
        u64 mm_site = 0, c0 = 0, first_ml = 0;
        const u64 max_hits = 1000;
        bool done = false;
        u32 fc = 0, fd = 0;
        if (seed_id < max_seed && tm < L) {
            const u64 hits = sc.hits0[r], sp = sc.sp0[r];
            u64 ml = sc.ml0[r];
            first_ml = ml;
            if (hits == 1) {
                // try_process_unique_mismatch_end_to_end (Schema.cpp:15164-15282)
                u64 p;
                if (sp >> 63) p = sp & ~(1ull << 63);          // the outcome table had the text position
                else { p = sa_at(ix, sp); n_sa++; }
                const u64 loc = ix.total - p - ml;
                seed_record(my, ns, ncand, sp, 1, ml, 0);
                c0 = loc; clen = 1;
                int error = 0;
                if (ml > (u64)firstC) ml = (u64)firstC;
                if (ml != (u64)L) {
                    const int need = L - (int)ml;
                    n_ung++;
                    if (!window_valid(ix, loc + ml, (u64)need, loc < ix.G)) {
                        // all-zero window: every position mismatches; the first sets ml = read_i (= ml), the second stops
                        error = need >= 2 ? 2 : 1;
                    } else {
                        // read position q faces doubled coordinate loc + q; 32 positions per step
                        const int ml0 = (int)ml;
                        GenStream gs; gs.init(ix, loc + (u64)(ml0 & ~31));
                        for (int q = ml0 & ~31; q < L && error < 2; q += 32) {
                            u64 mm = mism_bs(row[q >> 5], gs.next32());
                            if (dirty) mm |= spread32((u32)((row[W + (q >> 6)] >> (q & 63)) & 0xffffffffull));
                            mm &= field_range(ml0 - q, L - q);
                            if (mm) {
                                const int cnt = __popcll(mm);
                                if (error == 0) { ml = (u64)(q + (__ffsll((long long)mm) - 1) / 2); error = cnt >= 2 ? 2 : 1; }
                                else error = 2;
                            }
                        }
                    }
                }
                get_error = error;
                if (error == 0) { verdict = 1; st.exit_site[r] = loc; done = true; }
            }
            if (!done) {
                mm_site = ml;
                if (!pe_mode) {
                    if (ml == (u64)L && hits > 1) {
                        multi = 1;
                        if (firstC == L) { verdict = 4; done = true; }      // exact, ambiguous, no C in the read
                    }
                } else if (ml == (u64)L && hits > 1 && hits <= 10000) {
                    // get_candidates (Schema.cpp:18260-18290): every exact hit becomes a verified candidate
                    multi = 1;
                    if (firstC == L) { seed_record(my, ns, ncand, sp, hits, ml, 0); verdict = 4; done = true; }
                }
            }
            if (!done) {
                if (hits == 1) { /* recorded */ }
                else if (ml >= (u64)seed_len && hits <= max_hits) { if (hits != 0) { seed_record(my, ns, ncand, sp, hits, ml, (u64)tm); clen += hits; } }
                if (ml == 0) {
                    if (!dirty) tm = (L - tm < 18) ? L : tm + 8;
                    else tm = seed_offset_unmatch_p(L, tm, seq + (size_t)r * stride, 8, row + W);
                } else tm = tm + (int)(ml / 2);
                seed_id++;
            }
        }
        st.multi[r] = (u8)multi;
        // bit 15: the read character there is 'N' (only a dirty row can hold one: then the ASCII row is asked)
        bool isN = false;
        if (dirty && mm_site < (u64)L && ((row[W + (mm_site >> 6)] >> (mm_site & 63)) & 1)) isN = seq[(size_t)r * stride + mm_site] == 'N';
        st.mm_site[r] = (u16)(mm_site | (isN ? 0x8000u : 0u));
        if (done) seed_finish(st, r, verdict, ns, ncand, pe_mode);
        else {
            st.exit_site[r] = c0;
            st.n_seeds[r] = (u8)ns; st.n_cand[r] = (u32)ncand;
            sc.tm[r] = (u16)tm; sc.seed_id[r] = (u8)seed_id; sc.clen[r] = (u32)clen; sc.first_ml[r] = (u16)first_ml;
            // 1-mismatch first seed: second seed over the rest of the read (Schema.cpp:24734-24801)
            if (get_error == 1 && L - (int)first_ml >= 17) fc = 1; else fd = 1;
        }
        sc.flag_c[r] = fc; sc.flag_d[r] = fd;
    }
    if (counters) {
        atomicAdd(&shc[0], n_sa); atomicAdd(&shc[1], n_ung);
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long* cs = SHARD(counters); atomicAdd(&cs[2], (unsigned long long)shc[0]); atomicAdd(&cs[5], (unsigned long long)shc[1]); }
    }
}

// ---- second seed of the 1-mismatch reads + fast exit C (Schema.cpp:24734-24801, 24894-24898) -----
template <bool PACKED, bool KG = false>
__global__ void __launch_bounds__(64)
k_seed_second(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, const u64* __restrict__ count_ptr, int target_waves,
              int pe_mode, ReadState st, SeedCarry sc, unsigned long long* __restrict__ counters)
{
    int L = gm.L;                                     // length of the lane's current read
    const long total = (long)*count_ptr;
    const long chunk = seed_chunk(total, target_waves);
    const long chunk_begin = (long)blockIdx.x * chunk;
    if (chunk_begin >= total) return;
    __shared__ u64 s_c3[KG ? 27 : 1];
    const u64* c3 = KG ? kgram_c3(ix, s_c3) : nullptr;
    const WaveLogT wl_t = wavelog_begin();
    const long chunk_end = chunk_begin + chunk < total ? chunk_begin + chunk : total;
    long next = chunk_begin;
    LaneCounters lc = {0, 0, 0, 0, 0};
    bool active = false;
    long r = 0;
    const char* rd = seq;
    const u64* prow = nullptr;
    bool dirty = false;
    typename std::conditional<PACKED, SearchP, Search>::type S; SeedHit h;
    const u64 max_hits = 1000;
    bool verify = false;          // the interval shrank to one row: finish the count against the genome itself
    auto finish = [&]() {
        // the second seed is over: record it and decide (Schema.cpp:24748-24791)
        SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        int ns = st.n_seeds[r];
        u64 ncand = st.n_cand[r], clen = sc.clen[r];
        const u64 first_ml = sc.first_ml[r];
        const u64 second_len = (u64)L - first_ml;
        const u64 c0 = st.exit_site[r];
        u64 c1 = 0;
        int extra = 1;
        if (verify) {
            // A single-row interval: every further backward extension only tests whether the text character in
            // front of that ONE occurrence equals the next read character.  Instead of one dependent random
            // Occ gather per character (count_hash_table, bwt.h:1889-1933) locate the row once and compare the
            // rest of the read with the doubled genome in the index alphabet (C folded into T), 8 bases a step.
            verify = false;
            u64 p;
            if (S.top >> 63) p = S.top & ~(1ull << 63);            // the 20-mer table had the text position
            else { p = sa_at(ix, S.top); lc.n_sa++; }
            const int done_chars = 16 + S.s;                       // read[tm, tm+done_chars) is matched at text position p
            const int tm = S.tm;
            const u64 site = ix.total - p - (u64)done_chars;       // doubled coordinate of read[tm]
            bool ok = true;
            // 16 positions per step: one 16-byte row load, the window through a two-word cursor (one load per 32 bases)
            const int start = tm + done_chars;
            if constexpr (PACKED) ok = rest_matches_3l(ix, prow, pr.W, dirty, L, tm, start, site);
            else {
            Win32Cur wc; wc.init(ix, site + (u64)((start & ~15) - tm));
            for (int q = start & ~15; q < L && ok; q += 16) {
                const uint4 v = *reinterpret_cast<const uint4*>(rd + q);
                const u64 rw[2] = {((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z};
                const u64 d = site + (u64)(q - tm);                 // doubled coordinate facing read[q] (u64 wrap = out of range)
                u64 m[2];
                if (d + 16 <= ix.total) {
                    const u32 w32 = wc.at(d);
                    m[0] = mism8_3letter(rw[0], w32 & 0xffffu); m[1] = mism8_3letter(rw[1], w32 >> 16);
                } else {                                             // runs off the end of the text: '$' never matches
                    for (int hf = 0; hf < 2; hf++) {
                        m[hf] = 0;
                        for (int j = 0; j < 8; j++) {
                            const u64 dj = d + (u64)(8 * hf + j);
                            const char a = (char)((rw[hf] >> (8 * j)) & 0xff);
                            const bool eq = dj < ix.total && code3(a) <= 2 && code3(a) == code3("ACGT"[gbase(ix, dj)]);
                            if (!eq) m[hf] |= 0x80ull << (8 * j);
                        }
                    }
                }
                if (q < start) {                                     // positions before `start` are matched already
                    const int lo = start - q;
                    if (lo >= 8) { m[0] = 0; m[1] &= ~((1ull << (8 * (lo - 8))) - 1); } else m[0] &= ~((1ull << (8 * lo)) - 1);
                }
                if (q + 16 > L) {                                    // the read ends inside this piece
                    const int hi = L - q;
                    if (hi <= 8) { m[1] = 0; if (hi < 8) m[0] &= (1ull << (8 * hi)) - 1; } else m[1] &= (1ull << (8 * (hi - 8))) - 1;
                }
                if (m[0] | m[1]) ok = false;
            }
            }
            if (ok) { h.hits = 1; h.sp = (p - (u64)(S.steps - S.s)) | (1ull << 63); }     // located: text position of the full seed
            else { h.hits = 0; h.sp = 0; }
        }
        if (h.hits == 1) {
            u64 p;
            if (h.sp >> 63) p = h.sp & ~(1ull << 63);
            else { p = sa_at(ix, h.sp); lc.n_sa++; }
            c1 = ix.total - p - second_len - first_ml;
            seed_record(my, ns, ncand, h.sp, 1, second_len, first_ml);
            clen += 1; extra = 0;
        } else if (h.hits <= max_hits) {
            if (h.hits != 0) { seed_record(my, ns, ncand, h.sp, h.hits, second_len, first_ml); clen += h.hits; }
            extra = 0;
        }
        if (extra == 0) {
            int verdict = 0;
            if (clen == 1 || (clen == 2 && c0 == c1)) verdict = 2;            // fast exit C: exit_site = c0 already stored
            else if (clen != 0) verdict = 3;
            seed_finish(st, r, verdict, ns, ncand, pe_mode);
            sc.flag_d[r] = 0;
        } else {
            st.n_seeds[r] = (u8)ns; st.n_cand[r] = (u32)ncand; sc.clen[r] = (u32)clen;
            sc.flag_d[r] = 1;
        }
    };
    bool pending = true, have = false;
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (__popcll(pm) >= SEED_BATCH || !__any(active)) {
            if (pm == 0) break;
            const int rank = __popcll(pm & ((1ull << (threadIdx.x & 63)) - 1));
            const long it = next + rank;
            next += __popcll(pm);
            if (pending) {
                if (have) { finish(); have = false; }
                if (it < chunk_end) {
                    r = sc.list_c[it]; have = true; L = gm.rl(r);
                    bool go;
                    if constexpr (PACKED) {
                        prow = pr.base + (size_t)r * pr.pwords; dirty = pr.dirty[r] != 0;
                        go = search_begin_p<true, true>(ix, prow, pr.W, dirty, L, (int)sc.first_ml[r], S, h, lc.n_hash);
                    } else { rd = seq + (size_t)r * stride; go = search_begin<true, true>(ix, rd, L, (int)sc.first_ml[r], S, h, lc.n_hash); }
                    if (go) {
                        if (S.bot - S.top == 1) verify = true;          // already a single row: stays pending, verified next batch
                        else { active = true; pending = false; }
                    }
                } else pending = false;
            }
            if (!__any(active) && !__any(pending)) break;
            continue;
        }
#ifdef BMBS_UTIL
        if ((threadIdx.x & 63) == 0) atomicAdd(&counters[10], 1ull);
        if (active) atomicAdd(&counters[11], 1ull);
#endif
        if (active) {
            bool fin;
            if constexpr (PACKED) fin = search_step_p<true, KG>(ix, L, S, h, lc.n_ext, c3, &lc.n_jump); else fin = search_step<true>(ix, rd, L, S, h, lc.n_ext);
            if (fin) { active = false; pending = true; }
            else if (S.bot - S.top == 1) { verify = true; active = false; pending = true; }
        }
    }
    flush_counters(counters, lc, 1);
    wavelog_end(wl_t, 4);
}

// ---- the remaining seeds (Schema.cpp:24809-24889) -------------------------------------------------
// ROWS_LDS: ASCII rows staged in LDS (small indexes); PACKED: packed rows read from global memory (64 bytes per read)
// PLDS (with PACKED): the packed row of a lane's read (pwords x 8 bytes) is copied into the lane's LDS slot when the lane takes the
// read -- coalesced 16-byte loads along the rows -- and every seed start and cursor refill reads LDS: one request to the memory
// pipeline per seed start less (of about five), at 64 x (pwords + 1) x 8 = 4.6 KB of LDS per wave for 150-base reads
template <bool ROWS_LDS, bool PACKED = false, bool PLDS = false, bool KG = false>
__global__ void __launch_bounds__(64)
k_seed_extra(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, const u64* __restrict__ count_ptr, int target_waves,
             int seed_len, int pe_mode_x, ReadState st, SeedCarry sc, unsigned long long* __restrict__ counters)
{
    const int pe_mode = pe_mode_x & 0xff;
    const bool exp_nostore = (pe_mode_x >> 8) & 1;    // BMBS_EXP=1 (timing experiment)
    int L = gm.L;                                     // length of the lane's current read
    const long total = (long)*count_ptr;
    const long chunk = seed_chunk(total, target_waves);
    const long chunk_begin = (long)blockIdx.x * chunk;
    if (chunk_begin >= total) return;
    __shared__ u64 s_c3[KG ? 27 : 1];
    const u64* c3 = KG ? kgram_c3(ix, s_c3) : nullptr;
    const WaveLogT wl_t = wavelog_begin();
    const long chunk_end = chunk_begin + chunk < total ? chunk_begin + chunk : total;
    long next = chunk_begin;
    LaneCounters lc = {0, 0, 0, 0, 0};
    bool active = false, have = false;
    long r = 0;
    // A lane keeps its read for all the remaining seeds (about eight), and every seed start reads the row at a new offset.
    // Each such per-lane load is a request of its own to the memory pipeline (tools/gather_bench.hip), two thirds of all the
    // requests of this kernel.  So the wave copies the rows of the lanes that take a new read into LDS, with 16-byte loads
    // that run along the rows (about three cache lines per row instead of ~18 requests), and the searches read LDS
    // (ROWS_LDS = false: rows too long for the LDS of a 64-lane block; the lanes then read global memory as before.  A
    // template flag, so that the row pointer has one address space and the reads compile to ds_read, not to flat loads.)
    extern __shared__ __align__(16) char lds_rows[];
    __shared__ u32 take_row[64];
    __shared__ u8 take_lane[64];
    const int lstride = stride + 16;                  // 16-byte aligned rows, shifted against bank conflicts
    const char* rd = ROWS_LDS ? lds_rows + (size_t)(threadIdx.x & 63) * lstride : seq;
    const int plw = pr.pwords + 1;                    // PLDS: LDS row stride in words, odd against bank conflicts
    const u64* prow = PLDS ? reinterpret_cast<const u64*>(lds_rows) + (size_t)(threadIdx.x & 63) * plw : nullptr;
    bool dirty = false;
    typename std::conditional<PACKED, SearchP, Search>::type S; SeedHit h = {0, 0, 0};
    SeedRec* my = nullptr;
    int ns = 0, tm = 0, seed_id = 0, max_seed = 0;
    u64 ncand = 0, clen = 0;
    const u64 max_hits = 1000;
    // after a seed: record, advance (Schema.cpp:24830-24882); returns false when the read is finished
    auto after_seed = [&]() -> bool {
        const int cur_len = L - tm;
        const u64 ml = h.ml;
        auto rec = [&](u64 hits) { if (exp_nostore) { ns++; ncand += hits; } else seed_record(my, ns, ncand, h.sp, hits, ml, (u64)tm); };
        if (h.hits == 1) { rec(1); clen += 1; }
        else if (ml >= (u64)seed_len && h.hits <= max_hits) { if (h.hits != 0) { rec(h.hits); clen += h.hits; } }
        else if ((u64)cur_len == ml) return false;
        if (ml == 0) {
            // only a read with a character outside ACGT can hold the 'N' determine_seed_offset_unmatch looks for
            if (PACKED && !dirty) tm = (L - tm < 18) ? L : tm + 8;
            else if constexpr (PACKED) tm = seed_offset_unmatch_p(L, tm, seq + (size_t)r * stride, 8, prow + pr.W);
            else tm = seed_offset_unmatch(L, tm, rd, 8);
        } else tm = tm + (int)(ml / 2);
        seed_id++;
        return true;
    };
    // one transition of a pending lane: book the finished seed, start the next one or the next read.
    // `pending` stays set when the new seed was decided without stepping (handled in the next batch).
    bool pending = true, seed_done = false;
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (__popcll(pm) >= SEED_BATCH || !__any(active)) {
            if (pm == 0) break;
            if (pending && have && seed_done) {
                seed_done = false;
                if (!after_seed()) { seed_finish(st, r, clen != 0 ? 3 : 0, ns, ncand, pe_mode); have = false; }
            }
            if (pending && have && !(seed_id < max_seed && tm < L)) { seed_finish(st, r, clen != 0 ? 3 : 0, ns, ncand, pe_mode); have = false; }
            // lanes without a read take the next ones of the chunk
            const unsigned long long want = __ballot(pending && !have);
            const int rank = __popcll(want & ((1ull << (threadIdx.x & 63)) - 1));
            const long it = next + rank;
            const long avail = chunk_end - next;
            const int n_take = (long)__popcll(want) < avail ? __popcll(want) : (int)(avail > 0 ? avail : 0);
            next += __popcll(want);
            if ((ROWS_LDS || PLDS) && n_take > 0) {
                // (source row, destination lane) of every taker, then the copy: lane t moves piece t % per_row of taker t / per_row
                if (pending && !have && it < chunk_end) { take_row[rank] = sc.list_d[it]; take_lane[rank] = (u8)(threadIdx.x & 63); }
                __syncthreads();
                if constexpr (PLDS) {
                    const int per_row = pr.pwords / 2, pieces = n_take * per_row;         // rows are 16-byte aligned, pwords is even
                    u64* lr = reinterpret_cast<u64*>(lds_rows);
                    for (int t = threadIdx.x & 63; t < pieces; t += 64) {
                        const int w = t / per_row, cc = t - w * per_row;
                        u64 a, b; load2(pr.base + (size_t)take_row[w] * pr.pwords + 2 * cc, a, b);
                        u64* d = lr + (size_t)take_lane[w] * plw + 2 * cc;
                        d[0] = a; d[1] = b;
                    }
                } else {
                    const int per_row = stride / 16, pieces = n_take * per_row;
                    for (int t = threadIdx.x & 63; t < pieces; t += 64) {
                        const int w = t / per_row, cc = t - w * per_row;
                        const uint4 v = *reinterpret_cast<const uint4*>(seq + (size_t)take_row[w] * stride + (size_t)cc * 16);
                        *reinterpret_cast<uint4*>(lds_rows + (size_t)take_lane[w] * lstride + cc * 16) = v;
                    }
                }
                __syncthreads();
            }
            if (pending && !have) {
                if (it < chunk_end) {
                    r = sc.list_d[it]; L = gm.rl(r);
                    if (!ROWS_LDS) rd = seq + (size_t)r * stride;
                    if constexpr (PACKED) { if constexpr (!PLDS) prow = pr.base + (size_t)r * pr.pwords; dirty = pr.dirty[r] != 0; }
                    my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
                    ns = st.n_seeds[r]; ncand = st.n_cand[r]; clen = sc.clen[r]; tm = sc.tm[r]; seed_id = sc.seed_id[r];
                    max_seed = L / 10 == 0 ? 25 : (L / 10 - 1 > 25 ? 25 : L / 10 - 1);
                    have = true;
                    if (!(seed_id < max_seed && tm < L)) { seed_finish(st, r, clen != 0 ? 3 : 0, ns, ncand, pe_mode); have = false; }
                } else pending = false;         // chunk exhausted: this lane is done
            }
            if (pending && have) {
                bool go;
                if constexpr (PACKED) go = search_begin_p<false>(ix, prow, pr.W, dirty, L, tm, S, h, lc.n_hash); else go = search_begin<false>(ix, rd, L, tm, S, h, lc.n_hash);
                if (go) { active = true; pending = false; }
                else seed_done = true;              // decided without stepping: booked in the next batch
            }
            if (!__any(active) && !__any(pending)) break;
            continue;
        }
#ifdef BMBS_UTIL
        if ((threadIdx.x & 63) == 0) atomicAdd(&counters[12], 1ull);
        if (active) atomicAdd(&counters[13], 1ull);
#endif
        if (active) {
            bool fin;
            if constexpr (PACKED) fin = search_step_p<false, KG>(ix, L, S, h, lc.n_ext, c3, &lc.n_jump); else fin = search_step<false>(ix, rd, L, S, h, lc.n_ext);
            if (fin) { active = false; pending = true; seed_done = true; }
        }
    }
    flush_counters(counters, lc, 2);
    wavelog_end(wl_t, 5);
}

// ================================================================================================
// K5/K6 + a8-a10: locate, per-read candidate sort, run-length votes, reference vote order
// ================================================================================================
// (reverse_and_adjust_site, Schema.cpp:4669; generate_candidate_votes_shift, 4687-4773; std::sort(votes, compare_seed_votes), 24986)
// locate and vote in one pass for the usual small candidate lists: up to VOTE_REG candidates are located straight into
// registers, sorted by a fixed compare-exchange network and ranked (std::sort on <= 16 elements is libstdc++'s plain
// insertion sort, i.e. stable: rank = votes larger + equal votes earlier), so the candidate array never goes through
// memory and no per-lane sort runs on global memory.  Longer lists take the two-step path inside the same kernel.
#define VOTE_REG 16
#define VOTE_MID 32
__global__ void __launch_bounds__(64)
k_vote_fused(DevIndex ix, long n, ReadGeom gm, ReadState st, u64* __restrict__ cand, bmbs_vote* __restrict__ votes,
             u32* __restrict__ slot_read, u32* __restrict__ long_flag, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
             u32* __restrict__ mid_flag)
{
    // list != nullptr: the reads that have candidates, compacted (a quarter of a batch: with one lane per read of the whole
    // batch every wave ran the sort for a few busy lanes); n_votes and long_flag of the others were zeroed by the caller
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long r = it;
    if (list) { if (it >= (long)*count_ptr) return; r = list[it]; }
    else {
        if (r >= n) return;
        if (long_flag) long_flag[r] = 0;
    }
    if (st.verdict[r] != 3) { st.n_votes[r] = 0; return; }
    const int k = gm.rk(gm.rl(r));
    const u64 off = st.cand_off[r];
    const long nc = (long)st.n_cand[r];
    const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
    const int ns = st.n_seeds[r];
    bmbs_vote* v = votes + off;
    if (nc <= VOTE_REG) {
        u64 c[VOTE_REG];
        // slot j of the list = hit h of seed s, in seed order (locate + reverse_and_adjust_site, Schema.cpp:4669)
        int sidx = 0; u32 h = 0;
        u64 sp = 0, adj = 0; u32 hits = 0;
        if (ns > 0) { sp = my[0].sp; adj = (u64)my[0].len + (u64)my[0].off; hits = my[0].hits; }
#pragma unroll
        for (int j = 0; j < VOTE_REG; j++) {
            c[j] = ~0ull;
            if (j < nc) {
                while (h == hits && sidx + 1 < ns) { sidx++; h = 0; sp = my[sidx].sp; adj = (u64)my[sidx].len + (u64)my[sidx].off; hits = my[sidx].hits; }
                c[j] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + h)) - adj;
                h++;
            }
        }
        // Batcher odd-even merge sort, 16 keys, ascending (padding ~0 sinks to the end)
#define CE(a, b) { const u64 x_ = c[a], y_ = c[b]; c[a] = x_ < y_ ? x_ : y_; c[b] = x_ < y_ ? y_ : x_; }
        CE(0,1) CE(2,3) CE(4,5) CE(6,7) CE(8,9) CE(10,11) CE(12,13) CE(14,15)
        CE(0,2) CE(1,3) CE(4,6) CE(5,7) CE(8,10) CE(9,11) CE(12,14) CE(13,15)
        CE(1,2) CE(5,6) CE(9,10) CE(13,14)
        CE(0,4) CE(1,5) CE(2,6) CE(3,7) CE(8,12) CE(9,13) CE(10,14) CE(11,15)
        CE(2,4) CE(3,5) CE(10,12) CE(11,13)
        CE(1,2) CE(3,4) CE(5,6) CE(9,10) CE(11,12) CE(13,14)
        CE(0,8) CE(1,9) CE(2,10) CE(3,11) CE(4,12) CE(5,13) CE(6,14) CE(7,15)
        CE(4,8) CE(5,9) CE(6,10) CE(7,11)
        CE(2,4) CE(3,5) CE(6,8) CE(7,9) CE(10,12) CE(11,13)
        CE(1,2) CE(3,4) CE(5,6) CE(7,8) CE(9,10) CE(11,12) CE(13,14)
#undef CE
        // generate_candidate_votes_shift (Schema.cpp:4687-4773): one vote entry per run of equal sites, at the run's end
        u32 vote[VOTE_REG];
        bool last[VOTE_REG];
        u32 run = 0;
        int nv = 0;
#pragma unroll
        for (int i = 0; i < VOTE_REG; i++) {
            run = (i > 0 && c[i] == c[i - 1]) ? run + 1 : 1;
            vote[i] = run;
            last[i] = i < nc && (i + 1 >= nc || (i + 1 < VOTE_REG && c[i + 1] != c[i]));
            nv += last[i] ? 1 : 0;
        }
        // std::sort(votes, compare_seed_votes) (Schema.cpp:24986) on <= 16 entries: stable, descending by vote
#pragma unroll
        for (int i = 0; i < VOTE_REG; i++) {
            if (last[i]) {
                int rank = 0;
#pragma unroll
                for (int j = 0; j < VOTE_REG; j++) rank += (last[j] && (vote[j] > vote[i] || (vote[j] == vote[i] && j < i))) ? 1 : 0;
                bmbs_vote o;
                o.site = c[i] < (u64)k ? 0 : c[i] - (u64)k; o.vote = vote[i]; o.pad = 0;
                v[rank] = o;
            }
        }
        st.n_votes[r] = (u32)nv;
        for (long i = 0; i < nc; i++) slot_read[off + i] = i < nv ? (u32)r : 0xffffffffu;
        return;
    }
    // 17..32 candidates -- the usual case of a long read, which places up to 25 seeds: k_vote_mid, still one lane per read
    if (mid_flag && nc <= VOTE_MID) { mid_flag[r] = 1; st.n_votes[r] = 0; return; }
    // long lists (repeats): a whole block sorts each of them out of LDS (k_vote_long)
    if (long_flag) { long_flag[r] = 1; st.n_votes[r] = 0; return; }
    // single-lane form (stage API without the list buffers, and lists beyond the LDS capacity of k_vote_long)
    u64* c = cand + off;
    {
        u64 o = 0;
        for (int s2 = 0; s2 < ns && o < (u64)nc; s2++) {
            const u64 sp = my[s2].sp, adj = (u64)my[s2].len + (u64)my[s2].off;
            const u32 hh = my[s2].hits;
            for (u32 j = 0; j < hh && o < (u64)nc; j++) c[o++] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + j)) - adj;
        }
    }
    sort_u64_asc(c, nc);
    long nv = 0;
    u64 pre = c[0];
    u32 vote = 1;
    for (long i = 1; i < nc; i++) {
        if (c[i] == pre) vote++;
        else { v[nv].site = pre < (u64)k ? 0 : pre - (u64)k; v[nv].vote = vote; v[nv].pad = 0; nv++; vote = 1; pre = c[i]; }
    }
    v[nv].site = pre >= (u64)k ? pre - (u64)k : 0; v[nv].vote = vote; v[nv].pad = 0; nv++;
    intro_sort_desc(v, nv);             // std::sort(votes, compare_seed_votes), Schema.cpp:24986
    st.n_votes[r] = (u32)nv;
    for (long i = 0; i < nc; i++) slot_read[off + i] = i < nv ? (u32)r : 0xffffffffu;
}

// ---- 17..32 candidates: one lane per read, keys in registers ---------------------------------------------------------------
// A read of 180 bases and more places up to 25 seeds, so most of its lists have 17..32 entries; giving each of them a whole wave
// (k_vote_long) made the vote stage the largest kernel of a 250-bp batch.  Same scheme as the 16-key path of k_vote_fused with a
// bitonic network of 32; above 16 DISTINCT sites std::sort is no longer an insertion sort, and the order comes from the
// introsort emulation (bmbs_sort.h) on (vote, entry) items, the sites parked in the read's own candidate segment meanwhile.
__global__ void __launch_bounds__(64)
k_vote_mid(DevIndex ix, ReadGeom gm, ReadState st, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
           u64* __restrict__ cand, bmbs_vote* __restrict__ votes, u32* __restrict__ slot_read)
{
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= (long)*count_ptr) return;
    const long r = list[it];
    const int k = gm.rk(gm.rl(r));
    const u64 off = st.cand_off[r];
    const long nc = (long)st.n_cand[r];
    const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
    const int ns = st.n_seeds[r];
    bmbs_vote* v = votes + off;
    u64 c[VOTE_MID];
    {
        int sidx = 0; u32 h = 0;
        u64 sp = 0, adj = 0; u32 hits = 0;
        if (ns > 0) { sp = my[0].sp; adj = (u64)my[0].len + (u64)my[0].off; hits = my[0].hits; }
#pragma unroll
        for (int j = 0; j < VOTE_MID; j++) {
            c[j] = ~0ull;
            if (j < nc) {
                while (h == hits && sidx + 1 < ns) { sidx++; h = 0; sp = my[sidx].sp; adj = (u64)my[sidx].len + (u64)my[sidx].off; hits = my[sidx].hits; }
                c[j] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + h)) - adj;
                h++;
            }
        }
    }
    // bitonic network, 32 keys, ascending (padding ~0 sinks to the end); every index is a compile-time constant once unrolled
#pragma unroll
    for (int size = 2; size <= VOTE_MID; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int t = 0; t < VOTE_MID / 2; t++) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool asc = (i & size) == 0;
                const u64 x_ = c[i], y_ = c[j];
                const bool sw = asc ? x_ > y_ : x_ < y_;
                c[i] = sw ? y_ : x_; c[j] = sw ? x_ : y_;
            }
        }
    }
    u32 vote[VOTE_MID];
    bool last[VOTE_MID];
    u32 run = 0;
    int nv = 0;
#pragma unroll
    for (int i = 0; i < VOTE_MID; i++) {
        run = (i > 0 && c[i] == c[i - 1]) ? run + 1 : 1;
        vote[i] = run;
        last[i] = i < nc && (i + 1 >= nc || (i + 1 < VOTE_MID && c[i + 1] != c[i]));
        nv += last[i] ? 1 : 0;
    }
    if (nv <= 16) {
        // std::sort on <= 16 entries is an insertion sort: stable, descending by vote
#pragma unroll
        for (int i = 0; i < VOTE_MID; i++) {
            if (last[i]) {
                int rank = 0;
#pragma unroll
                for (int j = 0; j < VOTE_MID; j++) rank += (last[j] && (vote[j] > vote[i] || (vote[j] == vote[i] && j < i))) ? 1 : 0;
                bmbs_vote o;
                o.site = c[i] < (u64)k ? 0 : c[i] - (u64)k; o.vote = vote[i]; o.pad = 0;
                v[rank] = o;
            }
        }
    } else {
        bmbs_vk items[VOTE_MID];
        u64* park = cand + off;
        int e = 0;
#pragma unroll
        for (int i = 0; i < VOTE_MID; i++) {
            if (last[i]) { park[e] = c[i]; items[e].x = (vote[i] << 24) | (u32)e; e++; }
        }
        intro_sort_desc(items, (long)nv);          // std::sort(votes, compare_seed_votes), Schema.cpp:24986
        for (int j = 0; j < nv; j++) {
            const u32 x = items[j].x;
            const u64 site = park[x & 0xffffffu];
            bmbs_vote o;
            o.site = site < (u64)k ? 0 : site - (u64)k; o.vote = x >> 24; o.pad = 0;
            v[j] = o;
        }
    }
    st.n_votes[r] = (u32)nv;
    for (long i = 0; i < nc; i++) slot_read[off + i] = i < nv ? (u32)r : 0xffffffffu;
}

// ---- long candidate lists (reads inside repeats: up to 25 seeds x 1000 hits) -----------------------------------------------
// One lane sorting thousands of sites in global memory holds its whole wave for milliseconds; on a repeat-rich genome that was
// 10-40 ms per batch.  Here a 256-thread block takes one such read: the sites are located straight into LDS, sorted by a
// bitonic network, run-length encoded in parallel -- and only the vote order, which must be std::sort's exact (unstable)
// permutation (bmbs_sort.h), is produced by a single lane, on 4-byte (vote, index) items in LDS.
#define VL_CAP 4096           // block form: 256 threads per read
#define VL_BLOCK 256
#define VM_CAP 256            // wave form: 64 threads per read (most repeat reads have a few dozen candidates)
#define VM_BLOCK 64
// block-wide exclusive prefix of a 0/1 flag; returns the prefix, `total` the block count.  sh_w: one word per wave
DEVI int vl_prefix(bool flag, int* sh_w, int& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) sh_w[w] = __popcll(m);
    __syncthreads();
    int add = 0, tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) { const int x = sh_w[i]; if (i < w) add += x; tot += x; }
    __syncthreads();
    total = tot;
    return add + __popcll(m & ((1ull << lane) - 1));
}
// keys[0, nc) ascending, by the whole block, in place: stable 2-bit LSD passes.  Every thread owns E = ceil(nc / threads) consecutive
// keys in registers; a pass counts its keys per digit (four 16-bit counters in one u64), one block-wide exclusive scan of that word
// gives every key its destination.  nc <= 16 * blockDim.x.
template <int EMAX>
DEVI void vl_radix_sort(u64* keys, int nc)
{
    __shared__ u64 sh_scan[18];
    const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6, nw = T >> 6;
    const int E = (nc + T - 1) / T;                     // <= EMAX = capacity / threads of the instance
    u64 mine[EMAX];
    // the bits that vary: OR of key ^ keys[0]
    u64 diff = 0;
    const u64 k0 = keys[0];
    for (int i = tid; i < nc; i += T) diff |= keys[i] ^ k0;
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    if (lane == 0) sh_scan[w] = diff;
    __syncthreads();
    diff = 0;
    for (int i = 0; i < nw; i++) diff |= sh_scan[i];
    __syncthreads();
    const int nbits = diff ? 64 - __builtin_clzll(diff) : 0;
    for (int b = 0; b < nbits; b += 2) {
        u64 cnt = 0;
#pragma unroll
        for (int e = 0; e < EMAX; e++) {
            const int idx = tid * E + e;
            if (e < E && idx < nc) { mine[e] = keys[idx]; cnt += 1ull << (16 * (int)((mine[e] >> b) & 3)); }
        }
        // block-wide exclusive scan of cnt (four packed counters: a digit's total is at most 4096 < 2^16)
        u64 incl = cnt;
        for (int o = 1; o < 64; o <<= 1) { const u64 v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) sh_scan[w] = incl;
        __syncthreads();
        u64 wbase = 0, total = 0;
        for (int i = 0; i < nw; i++) { const u64 x = sh_scan[i]; if (i < w) wbase += x; total += x; }
        const u64 excl = wbase + incl - cnt;
        // first slot of every digit: totals of the smaller digits
        const u32 t0 = (u32)(total & 0xffff), t1 = (u32)((total >> 16) & 0xffff), t2 = (u32)((total >> 32) & 0xffff);
        // (slot of the next key of digit d = totals of the smaller digits + this thread's share of the scan; kept in four scalars:
        // an array indexed by the digit would live in scratch memory)
        u32 r0 = (u32)(excl & 0xffff), r1 = t0 + (u32)((excl >> 16) & 0xffff), r2 = t0 + t1 + (u32)((excl >> 32) & 0xffff),
            r3 = t0 + t1 + t2 + (u32)((excl >> 48) & 0xffff);
        __syncthreads();                                // every key is in registers: the array may be overwritten
#pragma unroll
        for (int e = 0; e < EMAX; e++) {
            const int idx = tid * E + e;
            if (e < E && idx < nc) {
                const int d = (int)((mine[e] >> b) & 3);
                const u32 r = d == 0 ? r0++ : d == 1 ? r1++ : d == 2 ? r2++ : r3++;
                keys[r] = mine[e];
            }
        }
        __syncthreads();
    }
}
// locate candidates j0 .. j0 + cnt - 1 of a read (cnt <= the LDS capacity) into keys[0, np2) (padded with ~0) and sort them ascending;
// returns np2.  build_pref: sh_pref (the running sum of the seeds' hit counts) is filled first -- once per read.
template <int EMAX>
DEVI int vl_locate_sort_range(const DevIndex& ix, const SeedRec* my, int ns, long j0, int cnt, u64* keys, u32* sh_pref, bool build_pref)
{
    if (build_pref && threadIdx.x == 0) { u32 a = 0; for (int s2 = 0; s2 < ns; s2++) { sh_pref[s2] = a; a += my[s2].hits; } sh_pref[ns] = a; }
    __syncthreads();
    int np2 = 32;
    while (np2 < cnt) np2 <<= 1;
    for (int j = threadIdx.x; j < np2; j += blockDim.x) {
        u64 key = ~0ull;
        if (j < cnt) {
            const u32 g = (u32)(j0 + j);
            int s2 = 0;
            while (s2 + 1 < ns && sh_pref[s2 + 1] <= g) s2++;
            const u64 sp = my[s2].sp, adj = (u64)my[s2].len + (u64)my[s2].off;
            key = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + (g - sh_pref[s2]))) - adj;
        }
        keys[j] = key;
    }
    __syncthreads();
    // long lists (a read inside a repeat family: up to 25 seeds x 1000 rows): LSD radix sort, two bits a pass over the bits that
    // vary -- 17 passes of one block scan each for a 6.2 G text, where the bitonic network takes 78 stages of 8 sweeps over 4096
    // keys.  On a GRCh38-like genome these lists were most of k_vote_pe_long's 11-13 ms per 10 M pairs.
    if (np2 > 512 && blockDim.x >= 128) { vl_radix_sort<EMAX>(keys, cnt); return np2; }
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < np2 / 2; t += blockDim.x) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool asc = (i & size) == 0;
                const u64 a = keys[i], b = keys[j];
                if ((a > b) == asc) { keys[i] = b; keys[j] = a; }
            }
            __syncthreads();
        }
    return np2;
}
template <int EMAX>
DEVI int vl_locate_sort(const DevIndex& ix, const SeedRec* my, int ns, int nc, u64* keys, u32* sh_pref)
{
    return vl_locate_sort_range<EMAX>(ix, my, ns, 0, nc, keys, sh_pref, true);
}
// a list beyond the LDS capacity (a read may collect 25 seeds x 1000 rows; one lane sorting ten thousand sites in global memory took
// 50 ms and held its whole launch): tiles of CAP candidates are located and sorted in LDS and parked in tmp[0, nc), every site then
// finds its place by a binary search in each of the other tiles (ties in tile order) and goes to c[rank].  Whole block; c and tmp
// are global arrays of nc sites each.
template <int CAP, int BLOCK>
DEVI void vl_sort_huge(const DevIndex& ix, const SeedRec* my, int ns, long nc, u64* keys, u32* sh_pref, u64* tmp, u64* c)
{
    const int T = (int)((nc + CAP - 1) / CAP);
    for (int t = 0; t < T; t++) {
        const long j0 = (long)t * CAP;
        const int cnt = (int)(nc - j0 < CAP ? nc - j0 : CAP);
        vl_locate_sort_range<(CAP + BLOCK - 1) / BLOCK>(ix, my, ns, j0, cnt, keys, sh_pref, t == 0);
        for (int j = threadIdx.x; j < cnt; j += BLOCK) tmp[j0 + j] = keys[j];
        __syncthreads();
    }
    for (long g = threadIdx.x; g < nc; g += BLOCK) {
        const int t = (int)(g / CAP);
        const u64 x = tmp[g];
        long rank = g - (long)t * CAP;
        for (int u = 0; u < T; u++) {
            if (u == t) continue;
            const u64* tu = tmp + (long)u * CAP;
            const long len = nc - (long)u * CAP < CAP ? nc - (long)u * CAP : CAP;
            long lo = 0, hi = len;
            while (lo < hi) { const long mid = (lo + hi) >> 1; const u64 y = tu[mid]; if (u < t ? y <= x : y < x) lo = mid + 1; else hi = mid; }
            rank += lo;
        }
        c[rank] = x;
    }
    __syncthreads();
}
// positions of the run ends of the sorted keys[0, nc), in order, into endpos; returns their number (block-uniform)
DEVI int vl_run_ends(const u64* keys, int nc, u16* endpos, int* sh_w)
{
    int running = 0;
    for (int base = 0; base < nc; base += blockDim.x) {
        const int i = base + (int)threadIdx.x;
        const bool flag = i < nc && (i == nc - 1 || keys[i + 1] != keys[i]);
        int total;
        const int pre = vl_prefix(flag, sh_w, total);
        if (flag) endpos[running + pre] = (u16)i;
        running += total;
    }
    __syncthreads();
    return running;
}

// ---- std::sort's permutation, in parallel ------------------------------------------------------------------------------------
// The vote order must be the exact permutation of libstdc++'s introsort (bmbs_sort.h).  Its moves are data-parallel all the
// same: in one __unguarded_partition pass the left cursor stops exactly at the positions whose vote is <= the pivot's (in
// ascending order: Lpos) and the right cursor at those >= it (descending: Rpos); the pass swaps Lpos[t] <-> Rpos[t] for
// every t with Lpos[t] < Rpos[t] (a prefix, T of them, since one list ascends and the other descends) and returns
// cut = min(Lpos[T], Rpos[T-1]) (Lpos[0] when T = 0).  Ranges above SMALL elements are partitioned by the whole block that
// way; the disjoint ranges of 17..SMALL elements that remain are finished by one lane each with the serial loop; the final
// insertion sort of std::sort is the stable sort of what the loop left, done with a bitonic network on (vote, position) keys.
// tests/test_sort_order.py checks this formulation against std::sort on the CPU.
struct VlRange { u16 f, l; int d; };
DEVI void vl_prefix2(bool f0, bool f1, int* sh_w, int& p0, int& p1, int& t0, int& t1)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
    if (lane == 0) { sh_w[w] = __popcll(m0); sh_w[nw + w] = __popcll(m1); }
    __syncthreads();
    int a0 = 0, a1 = 0, s0 = 0, s1 = 0;
    for (int i = 0; i < nw; i++) { const int x = sh_w[i], y = sh_w[nw + i]; if (i < w) { a0 += x; a1 += y; } s0 += x; s1 += y; }
    __syncthreads();
    t0 = s0; t1 = s1;
    const unsigned long long below = (1ull << lane) - 1;
    p0 = a0 + __popcll(m0 & below); p1 = a1 + __popcll(m1 & below);
}
// std::__unguarded_partition_pivot on items[first, last) by the whole block; returns the cut (block-uniform)
DEVI int vl_partition(bmbs_vk* items, int first, int last, u16* Lpos, u16* Rpos, int* sh_w)
{
    if (threadIdx.x == 0) {
        using namespace bmbs_sort_detail;
        move_median_to_first(items, (long)first, (long)first + 1, (long)first + (last - first) / 2, (long)last - 1);
    }
    __syncthreads();
    const u32 pv = items[first].x >> 24;
    const int n = last - first - 1;
    int nL = 0, nR = 0;
    for (int base = 0; base < n; base += (int)blockDim.x) {
        const int j = base + (int)threadIdx.x;
        const int iL = first + 1 + j, iR = last - 1 - j;
        const bool fl = j < n && (items[iL].x >> 24) <= pv;
        const bool fr = j < n && (items[iR].x >> 24) >= pv;
        int pl, pr, tl, tr;
        vl_prefix2(fl, fr, sh_w, pl, pr, tl, tr);
        if (fl) Lpos[nL + pl] = (u16)iL;
        if (fr) Rpos[nR + pr] = (u16)iR;
        nL += tl; nR += tr;
    }
    __syncthreads();
    const int m = nL < nR ? nL : nR;
    int T = 0;
    for (int base = 0; base < m; base += (int)blockDim.x) {
        const int t = base + (int)threadIdx.x;
        const bool c = t < m && Lpos[t] < Rpos[t];
        int tot;
        vl_prefix(c, sh_w, tot);
        T += tot;
        const int span = m - base < (int)blockDim.x ? m - base : (int)blockDim.x;
        if (tot < span) break;                      // the condition is monotone in t
    }
    for (int t = threadIdx.x; t < T; t += (int)blockDim.x) {
        const int a = Lpos[t], b = Rpos[t];
        const bmbs_vk x = items[a]; items[a] = items[b]; items[b] = x;
    }
    int cut;
    if (T == 0) cut = Lpos[0];
    else { const int lt = T < nL ? (int)Lpos[T] : 0x7fffffff, rt = Rpos[T - 1]; cut = lt < rt ? lt : rt; }
    __syncthreads();
    return cut;
}
// the serial introsort loop on a short range (<= 128 elements: the pending ranges are disjoint and each above 16)
DEVI void vl_intro_small(bmbs_vk* v, int first0, int last0, int depth0)
{
    using namespace bmbs_sort_detail;
    int sf[8], sl[8], sd[8];
    int sp = 1;
    sf[0] = first0; sl[0] = last0; sd[0] = depth0;
    while (sp > 0) {
        --sp;
        int first = sf[sp], last = sl[sp], depth = sd[sp];
        while (last - first > 16) {
            if (depth == 0) { heap_sort(v, (long)first, (long)last); break; }
            --depth;
            const int cut = (int)partition_pivot(v, (long)first, (long)last);
            if (last - cut > 16) { sf[sp] = cut; sl[sp] = last; sd[sp] = depth; ++sp; }
            last = cut;
        }
    }
}
// items[0, nv) stably by vote (the top byte), descending: 2-bit LSD passes over the vote bits that vary (a site collects at most one
// vote per seed: votes stay below 32, two or three passes), thread t owning items [t E, (t + 1) E) as vl_radix_sort does
template <int EMAX>
DEVI void vl_stable_by_vote_desc(bmbs_vk* items, int nv)
{
    __shared__ u64 sh_vscan[18];
    const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6, nw = (T + 63) >> 6;
    const int E = (nv + T - 1) / T;                     // <= EMAX
    u32 mine[EMAX];
    u32 diff = 0;
    const u32 v0 = items[0].x >> 24;
    for (int i = tid; i < nv; i += T) diff |= (items[i].x >> 24) ^ v0;
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    if (lane == 0) sh_vscan[w] = diff;
    __syncthreads();
    diff = 0;
    for (int i = 0; i < nw; i++) diff |= (u32)sh_vscan[i];
    __syncthreads();
    const int nbits = diff ? 32 - __builtin_clz(diff) : 0;
    for (int b = 0; b < nbits; b += 2) {
        u64 cnt = 0;
#pragma unroll
        for (int e = 0; e < EMAX; e++) {
            const int idx = tid * E + e;
            if (e < E && idx < nv) { mine[e] = items[idx].x; cnt += 1ull << (16 * (3 - (int)((mine[e] >> (24 + b)) & 3))); }
        }
        u64 incl = cnt;
        for (int o = 1; o < 64; o <<= 1) { const u64 v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) sh_vscan[w] = incl;
        __syncthreads();
        u64 wbase = 0, total = 0;
        for (int i = 0; i < nw; i++) { const u64 x = sh_vscan[i]; if (i < w) wbase += x; total += x; }
        const u64 excl = wbase + incl - cnt;
        const u32 t0 = (u32)(total & 0xffff), t1 = (u32)((total >> 16) & 0xffff), t2 = (u32)((total >> 32) & 0xffff);
        u32 r0 = (u32)(excl & 0xffff), r1 = t0 + (u32)((excl >> 16) & 0xffff), r2 = t0 + t1 + (u32)((excl >> 32) & 0xffff),
            r3 = t0 + t1 + t2 + (u32)((excl >> 48) & 0xffff);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EMAX; e++) {
            const int idx = tid * E + e;
            if (e < E && idx < nv) {
                const int d = 3 - (int)((mine[e] >> (24 + b)) & 3);
                const u32 r = d == 0 ? r0++ : d == 1 ? r1++ : d == 2 ? r2++ : r3++;
                items[r].x = mine[e];
            }
        }
        __syncthreads();
    }
}
// items[0, nv) -> std::sort(.., vote descending)'s permutation.  scratch: 4*CAP + 512 + CAP/2 bytes.  Returns false when a
// large range ran out of depth budget (heapsort fallback): the caller then takes the serial path.
template <int CAP, int SMALL, int EMAX>
DEVI bool vl_sort_votes(bmbs_vk* items, int nv, void* scratch, int* sh_w, int* ctl)
{
    u16* Lpos = (u16*)scratch;
    u16* Rpos = Lpos + CAP;
    VlRange* big = (VlRange*)(Rpos + CAP);
    VlRange* small = big + 64;
    if (nv > 16) {
        if (threadIdx.x == 0) {
            int lg = 0;
            for (int t = nv; t > 1; t >>= 1) lg++;
            VlRange rg; rg.f = 0; rg.l = (u16)nv; rg.d = 2 * lg;
            ctl[0] = 0; ctl[1] = 0; ctl[2] = 0;
            if (nv > SMALL) { big[0] = rg; ctl[0] = 1; } else { small[0] = rg; ctl[1] = 1; }
        }
        __syncthreads();
        while (true) {
            const int nb = ctl[0];
            if (nb == 0 || ctl[2]) break;
            const VlRange rg = big[nb - 1];
            __syncthreads();
            if (threadIdx.x == 0) ctl[0] = nb - 1;
            int first = rg.f, last = rg.l, depth = rg.d;
            while (last - first > SMALL) {
                if (depth == 0) { if (threadIdx.x == 0) ctl[2] = 1; break; }
                --depth;
                const int cut = vl_partition(items, first, last, Lpos, Rpos, sh_w);
                if (threadIdx.x == 0) {
                    VlRange q; q.f = (u16)cut; q.l = (u16)last; q.d = depth;
                    if (last - cut > SMALL) big[ctl[0]++] = q;
                    else if (last - cut > 16) small[ctl[1]++] = q;
                }
                last = cut;
            }
            if (threadIdx.x == 0 && last - first > 16 && last - first <= SMALL) {
                VlRange q; q.f = (u16)first; q.l = (u16)last; q.d = depth;
                small[ctl[1]++] = q;
            }
            __syncthreads();
        }
        if (ctl[2]) return false;
        const int n_small = ctl[1];
        for (int s2 = threadIdx.x; s2 < n_small; s2 += (int)blockDim.x) vl_intro_small(items, small[s2].f, small[s2].l, small[s2].d);
        __syncthreads();
    }
    // the final insertion sort = stable sort by vote descending of the current arrangement (a bitonic network on (vote, position)
    // keys did this before: 55 stages with a barrier each for 1024 items, most of the time of a long list)
    vl_stable_by_vote_desc<EMAX>(items, nv);
    return true;
}

// CAP, BLOCK = (VM_CAP, VM_BLOCK): lists of up to 256 candidates, one wave each; (VL_CAP, VL_BLOCK): the longer ones, one
// block each (the two instances walk the same list and take the reads of their size class: LO < nc <= CAP, the last one also beyond)
template <int CAP, int BLOCK, int LO>
__global__ void __launch_bounds__(BLOCK)
k_vote_long(DevIndex ix, ReadGeom gm, ReadState st, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
            u64* __restrict__ cand, bmbs_vote* __restrict__ votes, u32* __restrict__ slot_read, u32* __restrict__ big_list,
            unsigned long long* __restrict__ big_count)
{
    __shared__ u64 keys[CAP];
    __shared__ u16 endpos[CAP];
    __shared__ bmbs_vk items[CAP];
    __shared__ u32 sh_pref[BMBS_MAX_SEEDS + 1];
    __shared__ int sh_w[2 * (BLOCK / 64) + 1];
    __shared__ int sh_ctl[4];
    const long total_items = (long)*count_ptr;
    for (long item = blockIdx.x; item < total_items; item += gridDim.x) {
        const long r = list[item];
        const long nc = (long)st.n_cand[r];
        // the wave form sees every listed read and passes the ones beyond its capacity on to a list of their own (see k_vote_pe_long)
        if (big_list && CAP != VL_CAP && nc > CAP) { if (threadIdx.x == 0) big_list[atomicAdd(big_count, 1ull)] = (u32)r; continue; }
        if (nc <= LO || (CAP != VL_CAP && nc > CAP)) continue;          // another instance's size class
        const int k = gm.rk(gm.rl(r));
        const u64 off = st.cand_off[r];
        const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        const int ns = st.n_seeds[r];
        bmbs_vote* v = votes + off;
        if (nc > CAP) {
            // beyond the LDS capacity: the sites are sorted in tiles (vl_sort_huge; the vote segment, 16 bytes per candidate, parks the
            // tiles), the run ends are listed in the slot map (one word per candidate), and when the distinct sites fit the LDS the vote
            // order is made as for any other list; otherwise one lane runs std::sort's loop on the votes
            u64* c = cand + off;
            u64* tmp = reinterpret_cast<u64*>(v);
            u32* endidx = slot_read + off;
            vl_sort_huge<CAP, BLOCK>(ix, my, ns, nc, keys, sh_pref, tmp, c);
            int nvh = 0;
            for (long base = 0; base < nc; base += BLOCK) {
                const long i = base + (long)threadIdx.x;
                const bool flag = i < nc && (i == nc - 1 || c[i + 1] != c[i]);
                int tot;
                const int pre = vl_prefix(flag, sh_w, tot);
                if (flag) endidx[nvh + pre] = (u32)i;
                nvh += tot;
            }
            __syncthreads();
            for (long e = threadIdx.x; e < nvh; e += BLOCK) tmp[e] = c[endidx[e]];        // distinct sites, in order
            __syncthreads();
            if (nvh <= CAP) {
                for (int e = threadIdx.x; e < nvh; e += BLOCK) {
                    const u32 vote = endidx[e] - (e ? endidx[e - 1] : 0xffffffffu);
                    items[e].x = (vote << 24) | (u32)e;
                    c[e] = tmp[e];
                }
                __syncthreads();
                if (!vl_sort_votes<CAP, (CAP > 256 ? 128 : 32), (CAP + BLOCK - 1) / BLOCK>(items, nvh, keys, sh_w, sh_ctl)) {
                    for (int e = threadIdx.x; e < nvh; e += BLOCK) {
                        const u32 vote = endidx[e] - (e ? endidx[e - 1] : 0xffffffffu);
                        items[e].x = (vote << 24) | (u32)e;
                    }
                    __syncthreads();
                    if (threadIdx.x == 0) intro_sort_desc(items, (long)nvh);
                    __syncthreads();
                }
                for (int j = threadIdx.x; j < nvh; j += BLOCK) {
                    const u32 it = items[j].x;
                    const u64 site = c[it & 0xffffffu];
                    bmbs_vote o; o.site = site < (u64)k ? 0 : site - (u64)k; o.vote = it >> 24; o.pad = 0;
                    v[j] = o;
                }
            } else {
                // more distinct sites than the LDS holds: the votes in site order (written back to front: v[e] covers tmp[2e], tmp[2e + 1],
                // which only entries at or beyond e still need), then std::sort's loop on one lane
                if (threadIdx.x == 0) {
                    for (long e = nvh - 1; e >= 0; e--) {
                        const u64 site = tmp[e];
                        const u32 vote = endidx[e] - (e ? endidx[e - 1] : 0xffffffffu);
                        bmbs_vote o; o.site = site < (u64)k ? 0 : site - (u64)k; o.vote = vote; o.pad = 0;
                        v[e] = o;
                    }
                    intro_sort_desc(v, (long)nvh);
                }
            }
            __syncthreads();
            for (long i = threadIdx.x; i < nc; i += BLOCK) slot_read[off + i] = i < nvh ? (u32)r : 0xffffffffu;
            if (threadIdx.x == 0) st.n_votes[r] = (u32)nvh;
            __syncthreads();
            continue;
        }
        vl_locate_sort<(CAP + BLOCK - 1) / BLOCK>(ix, my, ns, (int)nc, keys, sh_pref);
        const int nv = vl_run_ends(keys, (int)nc, endpos, sh_w);
        // (vote, entry) items in site order; a site collects at most one vote per seed, so the vote fits 8 bits
        for (int e = threadIdx.x; e < nv; e += BLOCK) {
            const u32 vote = (u32)endpos[e] - (e ? (u32)endpos[e - 1] : 0xffffffffu);
            items[e].x = (vote << 24) | (u32)e;
        }
        // the sites move to the candidate segment in global memory: the sort below needs the LDS they occupy
        u64* c = cand + off;
        for (int e = threadIdx.x; e < nv; e += BLOCK) c[e] = keys[endpos[e]];
        __syncthreads();
        // std::sort(votes, compare_seed_votes), Schema.cpp:24986
        if (!vl_sort_votes<CAP, (CAP > 256 ? 128 : 32), (CAP + BLOCK - 1) / BLOCK>(items, nv, keys, sh_w, sh_ctl)) {
            for (int e = threadIdx.x; e < nv; e += BLOCK) {
                const u32 vote = (u32)endpos[e] - (e ? (u32)endpos[e - 1] : 0xffffffffu);
                items[e].x = (vote << 24) | (u32)e;
            }
            __syncthreads();
            if (threadIdx.x == 0) intro_sort_desc(items, (long)nv);
            __syncthreads();
        }
        for (int j = threadIdx.x; j < nv; j += BLOCK) {
            const u32 it = items[j].x;
            const u64 site = c[it & 0xffffffu];
            bmbs_vote o;
            o.site = site < (u64)k ? 0 : site - (u64)k; o.vote = it >> 24; o.pad = 0;
            v[j] = o;
        }
        for (long i = threadIdx.x; i < nc; i += BLOCK) slot_read[off + i] = i < nv ? (u32)r : 0xffffffffu;
        if (threadIdx.x == 0) st.n_votes[r] = (u32)nv;
        __syncthreads();
    }
}

// a9 alone (stage API, parity tests): the visiting order of given vote lists, one block per list
template <int CAP, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_vote_order(const uint8_t* __restrict__ vote, const long* __restrict__ seg_off, long n_seg, u32* __restrict__ perm)
{
    __shared__ u64 scratch[CAP];
    __shared__ bmbs_vk items[CAP];
    __shared__ int sh_w[2 * (BLOCK / 64) + 1];
    __shared__ int sh_ctl[4];
    for (long sg = blockIdx.x; sg < n_seg; sg += gridDim.x) {
        const long a = seg_off[sg];
        const int nv = (int)(seg_off[sg + 1] - a);
        if (nv <= 0 || nv > CAP) continue;
        for (int e = threadIdx.x; e < nv; e += BLOCK) items[e].x = ((u32)vote[a + e] << 24) | (u32)e;
        __syncthreads();
        if (!vl_sort_votes<CAP, (CAP > 256 ? 128 : 32), (CAP + BLOCK - 1) / BLOCK>(items, nv, scratch, sh_w, sh_ctl)) {
            for (int e = threadIdx.x; e < nv; e += BLOCK) items[e].x = ((u32)vote[a + e] << 24) | (u32)e;
            __syncthreads();
            if (threadIdx.x == 0) intro_sort_desc(items, (long)nv);
            __syncthreads();
        }
        for (int j = threadIdx.x; j < nv; j += BLOCK) perm[a + j] = items[j].x & 0xffffffu;
        __syncthreads();
    }
}

// the vote lists are shorter than the candidate segments they were built in: pack them densely
// (vote_off = exclusive scan of n_votes) so that the filter runs on full waves
__global__ void __launch_bounds__(256)
k_vote_compact(u64 n_slots, const u64* __restrict__ n_slots_dev, ReadState st, const u64* __restrict__ vote_off, const u32* __restrict__ slot_read,
               const bmbs_vote* __restrict__ votes, bmbs_vote* __restrict__ dense, u32* __restrict__ dense_read)
{
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_slots_dev) { const u64 nd = *n_slots_dev; if (nd < n_slots) n_slots = nd; }
    if (g >= n_slots) return;
    const u32 r = slot_read[g];
    if (r == 0xffffffffu) return;
    const u64 d = vote_off[r] + (g - st.cand_off[r]);
    dense[d] = votes[g];
    dense_read[d] = r;
}

// ================================================================================================
// K7+K8: window fetch + BS banded Myers, one candidate per lane, 64-bit words
// ================================================================================================
// BS_Reserve_Banded_BPM (Levenshtein_Cal.h:351-567); the 4 x 64-bit and 8 x 32-bit AVX2 forms
// (:1678, :2093) compute the same (err, end_site) per candidate.  pattern = window (L+2k bases from
// the doubled 2-bit genome), text = read; read 'T' also matches window 'C' (:384,473).
// W = u32 when the band (2k+1 bits) fits 32 bits (k <= 15, exactly the case in which the reference runs
// its 8 x 32-bit AVX2 form), else u64.
// k_filter is bound by VALU issue, not by memory (profiles/: ~85 % of its time was VALU issue with four sliding Peq vectors), so this
// form spends fewer instructions per read character: the window is kept as two bit planes (bit 0 / bit 1 of the 2-bit letters;
// 64 or 96 bases per register set, re-filled every 32 rows), the row's Peq is derived from them with the bisulfite rule folded in
// (T: plane 0 alone = {C, T}), and the characters come 16 per load.
DEVI void planes32(const DevIndex& ix, u64 d, u32& lo, u32& hi)      // the 32 bases starting at doubled coordinate d
{
    const int sh = (int)(d & 31) * 2;
    u64 w = ix.gen2[d >> 5] >> sh;
    if (sh) w |= ix.gen2[(d >> 5) + 1] << (64 - sh);
    auto squeeze = [](u32 x) -> u32 {           // bits 0, 2, 4 .. 30 -> bits 0 .. 15
        x &= 0x55555555u;
        x = (x | (x >> 1)) & 0x33333333u;
        x = (x | (x >> 2)) & 0x0f0f0f0fu;
        x = (x | (x >> 4)) & 0x00ff00ffu;
        x = (x | (x >> 8)) & 0x0000ffffu;
        return x;
    };
    const u32 a = (u32)w, b = (u32)(w >> 32);
    lo = squeeze(a) | (squeeze(b) << 16);
    hi = squeeze(a >> 1) | (squeeze(b >> 1) << 16);
}
// W = u32: band <= 31 bits (k <= 15), 64 bases of each plane in a register pair; W = u64: band <= 63 bits (k <= 31), 96 bases.
// PACKED: the read comes from a packed row (prow, 32 bases per word; pW = its base words; dirty = it holds characters outside ACGT)
template <class W, bool PACKED = false>
DEVI void bpm_planes(const DevIndex& ix, const char* rd, int L, int k, u64 site, u32& out_err, int& out_end,
                     const u64* prow = nullptr, int pW = 0, bool dirty = false)
{
    constexpr bool WIDE = sizeof(W) == 8;
    out_err = 0xffffffffu; out_end = -1;
    const int p_len = L + 2 * k;
    if (!window_valid(ix, site, (u64)p_len, site < ix.G)) return;
    const int band = 2 * k + 1;
    const W bmask = ((W)1 << band) - 1;
    u64 loS, hiS;                                       // bit j = plane bit of base site + i0 + j
    u32 loT = 0, hiT = 0;                               // WIDE: bits 64..95
    {
        u32 l0, h0, l1, h1;
        planes32(ix, site, l0, h0); planes32(ix, site + 32, l1, h1);
        loS = ((u64)l1 << 32) | l0; hiS = ((u64)h1 << 32) | h0;
        if (WIDE) planes32(ix, site + 64, loT, hiT);
    }
    W VP = 0, VN = 0;
    int err = 0;
    const int last_high = 2 * k;
    // one read character: CHECKED also tests the read's end and that the character is one of A, C, G, T
    auto step = [&](u32 tc, int i, int i0, auto checked) {
        const int sh = i - i0;
        W lo, hi;
        if (WIDE) {
            lo = (W)(sh ? (loS >> sh) | ((u64)loT << (64 - sh)) : loS) & bmask;
            hi = (W)(sh ? (hiS >> sh) | ((u64)hiT << (64 - sh)) : hiS) & bmask;
        } else { lo = (W)(loS >> sh) & bmask; hi = (W)(hiS >> sh) & bmask; }
        const W xl = (tc == 'A' || tc == 'G') ? (lo ^ bmask) : lo;
        const W xh = tc == 'G' ? hi : ~hi;
        W eq = tc == 'T' ? lo : (xl & xh);
        if (decltype(checked)::value) {
            const u32 idx = tc ^ 0x40u;                                         // 'A' 1, 'C' 3, 'G' 7, 'T' 20
            const u32 okc = idx < 32u ? (0x0010008au >> idx) & 1u : 0u;
            eq &= (W)0 - (W)okc;
        }
        W X = eq | VN;
        const W D0 = ((VP + (X & VP)) ^ VP) | X;
        const W HN = VP & D0;
        const W HP = VN | ~(VP | D0);
        X = D0 >> 1;
        const W VN2 = X & HP, VP2 = HN | ~(X | HP);
        if (decltype(checked)::value) { if (i < L) { VN = VN2; VP = VP2; err += 1 - (int)(D0 & 1u); } }
        else { VN = VN2; VP = VP2; err += 1 - (int)(D0 & 1u); }
    };
    for (int i0 = 0; i0 < L; i0 += 32) {
        if (i0) {
            u32 nl, nh;
            planes32(ix, site + (u64)i0 + (WIDE ? 64 : 32), nl, nh);
            if (WIDE) {
                loS = (loS >> 32) | ((u64)loT << 32); hiS = (hiS >> 32) | ((u64)hiT << 32);
                loT = nl; hiT = nh;
            } else { loS = (loS >> 32) | ((u64)nl << 32); hiS = (hiS >> 32) | ((u64)nh << 32); }
        }
        if constexpr (PACKED) {
            // 32 bases per word; the per-character step sees the 2-bit code (A0 C1 G2 T3) and, for dirty rows, the not-ACGT bit
            const u64 rb = prow[i0 >> 5];
            const u32 mb = dirty ? (u32)((prow[pW + (i0 >> 6)] >> (i0 & 63)) & 0xffffffffull) : 0u;
            auto step_p = [&](u32 c, u32 bad, int i, auto checked) {
                const int sh = i - i0;
                W lo, hi;
                if (WIDE) {
                    lo = (W)(sh ? (loS >> sh) | ((u64)loT << (64 - sh)) : loS) & bmask;
                    hi = (W)(sh ? (hiS >> sh) | ((u64)hiT << (64 - sh)) : hiS) & bmask;
                } else { lo = (W)(loS >> sh) & bmask; hi = (W)(hiS >> sh) & bmask; }
                const W xl = (c & 1u) ? lo : (lo ^ bmask);
                const W xh = (c & 2u) ? hi : ~hi;
                W eq = c == 3u ? lo : (xl & xh);
                if (decltype(checked)::value) eq &= (W)0 - (W)(bad ^ 1u);
                W X = eq | VN;
                const W D0 = ((VP + (X & VP)) ^ VP) | X;
                const W HN = VP & D0;
                const W HP = VN | ~(VP | D0);
                X = D0 >> 1;
                const W VN2 = X & HP, VP2 = HN | ~(X | HP);
                if (decltype(checked)::value) { if (i < L) { VN = VN2; VP = VP2; err += 1 - (int)(D0 & 1u); } }
                else { VN = VN2; VP = VP2; err += 1 - (int)(D0 & 1u); }
            };
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int ib = i0 + 16 * half;
                if (ib >= L) break;
                const u32 c16 = (u32)(rb >> (32 * half));
                const u32 m16 = (mb >> (16 * half)) & 0xffffu;
                const bool plain = m16 == 0 && ib + 16 <= L;
                if (__all(plain)) {
#pragma unroll
                    for (int c = 0; c < 16; c++) step_p((c16 >> (2 * c)) & 3u, 0u, ib + c, std::false_type());
                } else {
#pragma unroll
                    for (int c = 0; c < 16; c++) step_p((c16 >> (2 * c)) & 3u, (m16 >> c) & 1u, ib + c, std::true_type());
                }
                if (__all(err - last_high > k)) return;
            }
            continue;
        }
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int ib = i0 + 16 * half;
            if (ib >= L) break;
            const uint4 v = *reinterpret_cast<const uint4*>(rd + ib);       // rows are 16-byte aligned and padded
            const u32 cw[4] = {v.x, v.y, v.z, v.w};
            // a wave whose lanes all hold 16 characters of A/C/G/T inside their reads runs the unchecked steps.  The letter a
            // byte would have to be, rebuilt from its bits 1-2 (A 00, C 01, G 11, T 10): 0x41 | bits 1-2, T: ^ 0x11
            u32 bad = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 x = cw[q];
                const u32 isT = (x >> 2) & ~(x >> 1) & 0x01010101u;
                bad |= x ^ ((0x41414141u | (x & 0x06060606u)) ^ (isT * 0x11u));
            }
            const bool plain = bad == 0 && ib + 16 <= L;
            if (__all(plain)) {
#pragma unroll
                for (int c = 0; c < 16; c++) step((cw[c >> 2] >> (8 * (c & 3))) & 0xffu, ib + c, i0, std::false_type());
            } else {
#pragma unroll
                for (int c = 0; c < 16; c++) step((cw[c >> 2] >> (8 * (c & 3))) & 0xffu, ib + c, i0, std::true_type());
            }
            // a candidate that cannot come back under k (Levenshtein_Cal.h:455) ends with err = ~0 below whether or not it
            // goes on; the wave stops once that is every lane
            if (__all(err - last_high > k)) return;
        }
    }
    if (err - last_high > k) return;
    // minimum over the last 2k+1 columns; later column wins ties, then the un-gapped diagonal
    // (Levenshtein_Cal.h:511-563)
    const int site_e = L - 1;
    u32 best = 0xffffffffu;
    int ret = -1;
    if (err <= k && (u32)err <= best) { best = (u32)err; ret = site_e; }
    int i = 0;
    while (i < k) {
        err += (int)((VP >> i) & 1); err -= (int)((VN >> i) & 1); ++i;
        if (err <= k && (u32)err <= best) { best = (u32)err; ret = site_e + i; }
    }
    const u32 ungap = (u32)err;
    while (i < last_high) {
        err += (int)((VP >> i) & 1); err -= (int)((VN >> i) & 1); ++i;
        if (err <= k && (u32)err <= best) { best = (u32)err; ret = site_e + i; }
    }
    if (ungap <= (u32)k && ungap == best) ret = site_e + k;
    out_err = best; out_end = ret;
}

DEVI void bpm_one(const DevIndex& ix, const char* rd, int L, int k, u64 site, u32& out_err, int& out_end)
{
    if (k <= 15) bpm_planes<u32>(ix, rd, L, k, site, out_err, out_end);          // k is wave-uniform unless lengths are mixed
    else bpm_planes<u64>(ix, rd, L, k, site, out_err, out_end);
}
// read r of a batch: from its packed row when the batch has them, else from the ASCII row
DEVI void bpm_read(const DevIndex& ix, const char* seq, int stride, const PackedRows& pr, long r, int L, int k, u64 site, u32& out_err, int& out_end)
{
    if (pr.base) {
        const u64* row = pr.base + (size_t)r * pr.pwords;
        const bool dirty = pr.dirty[r] != 0;
        if (k <= 15) bpm_planes<u32, true>(ix, nullptr, L, k, site, out_err, out_end, row, pr.W, dirty);
        else bpm_planes<u64, true>(ix, nullptr, L, k, site, out_err, out_end, row, pr.W, dirty);
    } else bpm_one(ix, seq + (size_t)r * stride, L, k, site, out_err, out_end);
}

__global__ void __launch_bounds__(256)
k_filter(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, const u64* __restrict__ n_votes_total,
         const u32* __restrict__ dense_read, const bmbs_vote* __restrict__ dense, u32* __restrict__ ferr,
         int* __restrict__ fend, unsigned long long* __restrict__ counters)
{
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= *n_votes_total) return;
    const u32 r = dense_read[g];
    u32 e; int es;
    const int L = gm.rl(r), k = gm.rk(L);
    bpm_read(ix, seq, stride, pr, (long)r, L, k, dense[g].site, e, es);
    ferr[g] = e; fend[g] = es;
    if (counters) atomicAdd(&SHARD(counters)[3], 1ull);
}

// K5 on its own (bmbs_locate_batch)
__global__ void __launch_bounds__(256)
k_locate_rows(DevIndex ix, const u64* __restrict__ row, long n, u64* __restrict__ pos)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos[i] = row[i] <= ix.total ? sa_at(ix, row[i]) : ~0ull;
}

// K7 on its own (bmbs_window_batch): one window per thread
__global__ void __launch_bounds__(256)
k_window(DevIndex ix, const u64* __restrict__ site, long n, int len, char* __restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 s0 = site[i];
    const bool valid = window_valid(ix, s0, (u64)len, s0 < ix.G);
    WinReader wr; wr.init(ix, s0, valid);
    for (int j = 0; j < len; j++) { const int b = wr.next(); out[(size_t)i * len + j] = b > 3 ? 0 : "ACGT"[b]; }
}

// standalone form for bmbs_filter_batch: explicit (read, site) pairs
__global__ void __launch_bounds__(256)
k_filter_pairs(DevIndex ix, const char* __restrict__ seq, ReadGeom gm, int stride, u64 n_cand,
               const u32* __restrict__ read_of, const u64* __restrict__ site,
               u32* __restrict__ ferr, int* __restrict__ fend)
{
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_cand) return;
    u32 e; int es;
    const int L = gm.rl(read_of[g]), k = gm.rk(L);
    bpm_one(ix, seq + (size_t)read_of[g] * stride, L, k, site[g], e, es);
    ferr[g] = e; fend[g] = es;
}

// ================================================================================================
// K9: ordered reduction over a read's votes (Schema.cpp:7847-8172 / 8335-8750)
// ================================================================================================
// What the loop below leaves behind, as a summary of an ORDERED run of votes that merges left to right (as pair_comb): the lowest
// error m, where it first occurs (i0, with its site + end t0), whether a later vote reaches m at another place (amb), and the
// lowest error before i0 (pm: second_best_diff is the drop at the moment the final best was first met).
struct RedSum { u32 m, pm; int i0; int amb; u64 t0; };          // i0 < 0: empty
DEVI RedSum red_comb(const RedSum& L, const RedSum& R)
{
    if (R.i0 < 0) return L;
    if (L.i0 < 0) return R;
    RedSum o;
    if (L.m < R.m) o = L;
    else if (L.m > R.m) { o = R; o.pm = L.m < R.pm ? L.m : R.pm; }
    else { o = L; o.amb = L.amb | R.amb | (R.t0 != L.t0 ? 1 : 0); }
    return o;
}
__global__ void __launch_bounds__(256)
k_reduce(long n, int ambiguous_out, ReadState st, const u64* __restrict__ vote_off, const bmbs_vote* __restrict__ votes,
         const u32* __restrict__ ferr, const int* __restrict__ fend, const u64* __restrict__ count_ptr, const u32* __restrict__ list)
{
    // list != nullptr: the compacted list of reads with candidates (k_vote_fused's); job_flag / red_status of the others were zeroed
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    long r = it;
    bool act = true;
    if (list) { if (it >= (long)*count_ptr) act = false; else r = list[it]; }
    else {
        if (r >= n) act = false;
        else { st.job_flag[r] = 0; st.red_status[r] = 0; }
    }
    if (act && st.verdict[r] != 3) act = false;
    const u64 off = act ? vote_off[r] : 0;
    const long nv = act ? (long)st.n_votes[r] : 0;
    u32 min_err = 0xfffffffeu, sbd = 0;
    long min_idx = -1;
    const bool coop = nv > 64;                  // a read inside a repeat family: hundreds of verified votes -- the whole wave walks them
    if (act && !coop) {
        u64 min_site = ~0ull;
        for (long i = 0; i < nv; i++) {
            const u32 e = ferr[off + i];
            const u64 tmp_site = votes[off + i].site + (u64)(long long)fend[off + i];
            if (e == min_err && min_site != tmp_site && min_idx >= 0) { sbd = 0; min_idx = -2 - min_idx; }
            else if (e < min_err) { sbd = min_err - e; min_err = e; min_idx = i; min_site = tmp_site; }
        }
    }
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long rr = (long)__shfl((long long)r, src, 64);
        const u64 o2 = vote_off[rr];
        const long nv2 = (long)st.n_votes[rr];
        RedSum tot; tot.m = 0; tot.pm = 0xfffffffeu; tot.i0 = -1; tot.amb = 0; tot.t0 = 0;
        for (long base = 0; base < nv2; base += 64) {
            const long i = base + lane;
            RedSum me; me.m = 0; me.pm = 0xfffffffeu; me.i0 = -1; me.amb = 0; me.t0 = 0;
            if (i < nv2) {
                const u32 e = ferr[o2 + i];
                if (e < 0xfffffffeu) { me.m = e; me.i0 = (int)i; me.t0 = votes[o2 + i].site + (u64)(long long)fend[o2 + i]; }
            }
            for (int d = 1; d < 64; d <<= 1) {
                RedSum o;
                o.m = __shfl_down(me.m, d, 64); o.pm = __shfl_down(me.pm, d, 64); o.i0 = __shfl_down(me.i0, d, 64);
                o.amb = __shfl_down(me.amb, d, 64); o.t0 = (u64)__shfl_down((long long)me.t0, d, 64);
                if ((lane & (2 * d - 1)) == 0) me = red_comb(me, o);
            }
            RedSum ch;
            ch.m = __shfl(me.m, 0, 64); ch.pm = __shfl(me.pm, 0, 64); ch.i0 = __shfl(me.i0, 0, 64); ch.amb = __shfl(me.amb, 0, 64);
            ch.t0 = (u64)__shfl((long long)me.t0, 0, 64);
            tot = red_comb(tot, ch);
        }
        if (lane == src && tot.i0 >= 0) {
            min_err = tot.m;
            if (tot.amb) { sbd = 0; min_idx = -2 - (long)tot.i0; }
            else { sbd = tot.pm - tot.m; min_idx = tot.i0; }
        }
    }
    if (!act) return;
    if (min_idx >= 0) {
        st.best_site[r] = votes[off + min_idx].site;
        st.best_end[r] = fend[off + min_idx];
        st.best_err[r] = min_err;
        st.sbd[r] = sbd;
        st.red_status[r] = 1;
        st.job_flag[r] = min_err != 0 ? 1u : 0u;
    } else if (min_idx != -1) {
        st.red_status[r] = 2;
        if (ambiguous_out) {
            // --ambiguous_out (Schema.cpp:25095-25118): the first candidate that reached the minimum is aligned and reported
            const long a = -2 - min_idx;
            st.best_site[r] = votes[off + a].site;
            st.best_end[r] = fend[off + a];
            st.best_err[r] = min_err;
            st.sbd[r] = 0;
            st.job_flag[r] = min_err != 0 ? 1u : 0u;
        }
    }
}

// ---- launches without host round trips ------------------------------------------------------------------------------------------
// The stage counts (candidate slots, jobs, DP jobs, re-seeded candidates) are only known on the device.  A call that does not
// wait for them sizes its buffers and grids from what earlier calls of the same shape needed (plus a margin) and lets these
// guards compare the real count with that capacity right after the scan that produced it.  On overflow the guard raises a
// flag and takes the work away from every later kernel (nothing is written out of bounds); the host sees the flag when it next
// synchronises and runs the batch again with exact sizes.  The statistics of a call are added to the context's counters by
// k_stats_commit only when no flag is up, so a repeated batch counts once.
#define BMBS_FLAG_CAND   0   // candidate slots > capacity
#define BMBS_FLAG_SW     1   // DP jobs > capacity of the launches issued
#define BMBS_FLAG_RCAND  2   // --sensitive: re-seeded candidates > capacity
#define BMBS_FLAG_CIGAR  3   // the caller's CIGAR pool is too small for the jobs of this batch (an error, not a retry)
#define BMBS_FLAG_WORDS  8
__global__ void __launch_bounds__(256)
k_guard_cand(u64* __restrict__ total, u64 cap, u32* __restrict__ flags, long n, u8* __restrict__ verdict, u32* __restrict__ n_cand,
             u64* __restrict__ cand_off)
{
    if (*total <= cap && !flags[BMBS_FLAG_CAND]) return;
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) { verdict[r] = 0; n_cand[r] = 0; cand_off[r] = 0; }
    if (r == 0) { cand_off[n] = 0; flags[BMBS_FLAG_CAND] = 1; }
    // *total is cleared by k_guard_done (one thread, after this kernel: other blocks still read it here)
}
__global__ void k_guard_done(u64* __restrict__ total, const u32* __restrict__ flags, int flag) { if (flags[flag]) *total = 0; }
// jobs: the arrays are sized for one job per read, only the caller's CIGAR pool can be too small
__global__ void __launch_bounds__(256)
k_guard_jobs(const u64* __restrict__ n_jobs, u64 max_ops, u64 cigar_cap, u32* __restrict__ flags, long n, u32* __restrict__ job_flag)
{
    if (*n_jobs * max_ops <= cigar_cap && !flags[BMBS_FLAG_CIGAR]) return;
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) job_flag[r] = 0;
    if (r == 0) flags[BMBS_FLAG_CIGAR] = 1;
}
__global__ void k_guard_count(const u64* __restrict__ count, u64 cap, u32* __restrict__ flags, int flag) { if (*count > cap) flags[flag] = 1; }
// --sensitive: candidates of the re-seeded mates
__global__ void __launch_bounds__(256)
k_guard_rcand(const u64* __restrict__ total, u64 cap, u32* __restrict__ flags, long n, u32* __restrict__ rcnt, u64* __restrict__ item_off)
{
    if (*total <= cap && !flags[BMBS_FLAG_RCAND]) return;
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) { rcnt[r] = 0; item_off[r] = 0; }
    if (r == 0) { item_off[n] = 0; flags[BMBS_FLAG_RCAND] = 1; }
}
// the five mapstats counters of this call (sharded like the context's) -> the context's, unless the call is going to be repeated
__global__ void __launch_bounds__(256)
k_stats_commit(const u32* __restrict__ flags, unsigned long long* __restrict__ call_stats, unsigned long long* __restrict__ stats, int words)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    const bool again = flags[BMBS_FLAG_CAND] | flags[BMBS_FLAG_SW] | flags[BMBS_FLAG_RCAND] | flags[BMBS_FLAG_CIGAR];
    const unsigned long long v = call_stats[i];
    call_stats[i] = 0;
    if (!again && v) stats[i] += v;
}

// 16-mer lookups and extensions of this call (summed over the counter shards) -> totals[14], [15]: the host learns from them
// whether the reads of this input walk the index in long chains (three-letter steps pay off) or not
__global__ void k_call_chain_counts(const unsigned long long* __restrict__ counters, u64* __restrict__ totals)
{
    const int j = threadIdx.x;                  // 0: lookups, 1: extensions
    if (j >= 2) return;
    u64 t = 0;
    for (int sdx = 0; sdx < BMBS_SHARDS; sdx++) t += counters[sdx * BMBS_SHARD_WORDS + j];
    totals[14 + j] = t;
}

// job arrays shared by the fused path and bmbs_align_batch
struct Jobs {
    const u32* read;      // read (row) of the job
    const u64* site;      // window start (doubled coordinate)
    const int* end;       // end_site from the filter
    const u32* err;       // err from the filter
};

__global__ void k_job_list(long n, ReadState st, u32* __restrict__ job_read, u64* __restrict__ job_site,
                           int* __restrict__ job_end, u32* __restrict__ job_err)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (st.job_flag[r]) {
        const u64 j = st.job_off[r];
        job_read[j] = (u32)r; job_site[j] = st.best_site[r]; job_end[j] = st.best_end[r]; job_err[j] = st.best_err[r];
    }
}

// ================================================================================================
// K11-K13: un-gapped recheck, banded affine-gap semi-global alignment with traceback, CIGAR + NM
// ================================================================================================
// fast_recalculate_bs_Cigar (ksw.cpp:2578-2876) = try_cigar_without_path (:2515) else
// ksw_semi_global_quality_back (:1850-2045) + leading/trailing-I folding + NM recount.
// pen_lut[q] = MismatchPenaltyByQuality(q) evaluated on the host in IEEE double (ksw.h:148-161); the
// per-cell `(int)(mat_diff * Phred)` of the reference (ksw.cpp:1950) is the same product.
//
// Two kernels so that lanes stay dense: k_align_ungapped (every job; most succeed) marks the jobs
// that really need the DP, k_align_sw runs only those (compacted by a scan, no host round-trip).
// k_align_sw keeps the whole DP band (H, E of 2k+2 cells, the 2-bit window, the row's trace nibbles)
// in REGISTERS: the band loop is fully unrolled for a compile-time bound KB >= k, cell b of row i
// reads slot b and writes slot b-1 (the band slides one column per row), and the only memory traffic
// of the DP is one packed trace word (4 bits per cell) per 16 cells per row, interleaved by job.

// quality row of read r.  Paired-end calls hand over the caller's two buffers as they are (qual: mate 1, qual2: mate 2 of pair
// r - rev_qual_from) instead of copying 2 x n rows into one: the alignment kernels touch the qualities of a few reads only.
DEVI const char* qual_row(const char* qual, const char* qual2, u32 rev_qual_from, u32 r, int stride)
{
    return (qual2 && r >= rev_qual_from) ? qual2 + (size_t)(r - rev_qual_from) * stride : qual + (size_t)r * stride;
}

__global__ void __launch_bounds__(256)
k_align_ungapped(DevIndex ix, ScoreParams sp, const int* __restrict__ pen_lut, const char* __restrict__ seq,
                 const char* __restrict__ qual, const char* __restrict__ qual2, ReadGeom gm, int stride, u64 n_jobs, const u64* __restrict__ n_jobs_dev, Jobs jb_, u32 rev_qual_from,
                 int* __restrict__ a_start, int* __restrict__ a_end, u32* __restrict__ a_nm, int* __restrict__ a_score,
                 int* __restrict__ a_nops, u32* __restrict__ need_sw, unsigned long long* __restrict__ counters)
{
    const u64 jb = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_jobs_dev) { const u64 nd = *n_jobs_dev; if (nd < n_jobs) n_jobs = nd; }
    if (jb >= n_jobs) return;
    const u32 r = jb_.read[jb];
    const u64 site = jb_.site[jb];
    const int end_site = jb_.end[jb];
    const u32 err_in = jb_.err[jb];
    const int L = gm.rl(r), k = gm.rk(L);
    need_sw[jb] = 0;
    if (err_in == 0) {          // fast_recalculate_bs_Cigar's own err == 0 branch (ksw.cpp:2607-2616)
        a_start[jb] = end_site - L + 1; a_end[jb] = end_site; a_nm[jb] = 0; a_score[jb] = 0; a_nops[jb] = 0;
        return;
    }
    const char* rd = seq + (size_t)r * stride;
    const char* ql = qual_row(qual, qual2, rev_qual_from, r, stride);
    const bool rev = r >= rev_qual_from;
    const int p_len = L + 2 * k;
    const bool wvalid = window_valid(ix, site, (u64)p_len, site < ix.G);
    const int start = end_site - L + 1;
    // an out-of-strand window compares unequal everywhere (err_in <= k < L mismatches can never account for that)
    bool ok = start >= 0 && wvalid;
    int tmp_err = 0, score = 0;
    if (ok) {
        // 16 read characters (one 16-byte load) against 16 window bases per step; the quality penalties of the few mismatching
        // positions are picked up from the mask
        Win32Cur wc; wc.init(ix, site + (u64)start);
        for (int p = 0; p < L && ok; p += 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(rd + p);
            const u64 r0 = ((u64)v.y << 32) | v.x, r1 = ((u64)v.w << 32) | v.z;
            const u32 w32 = wc.at(site + (u64)start + (u64)p);
            u64 m0 = mism8(r0, w32 & 0xffff), m1 = mism8(r1, w32 >> 16);
            const int left = L - p;
            if (left < 16) {
                if (left <= 8) { m1 = 0; if (left < 8) m0 &= (1ull << (8 * left)) - 1; }
                else m1 &= (1ull << (8 * (left - 8))) - 1;
            }
            tmp_err += __popcll(m0) + __popcll(m1);
            if (tmp_err > (int)err_in) { ok = false; break; }
            while (m0) {
                const int i = p + (__ffsll((unsigned long long)m0) - 1) / 8;
                m0 &= m0 - 1;
                score -= rd[i] == 'N' ? sp.np : pen_lut[(unsigned char)ql[rev ? L - 1 - i : i]];
            }
            while (m1) {
                const int i = p + 8 + (__ffsll((unsigned long long)m1) - 1) / 8;
                m1 &= m1 - 1;
                score -= rd[i] == 'N' ? sp.np : pen_lut[(unsigned char)ql[rev ? L - 1 - i : i]];
            }
        }
        if (ok && tmp_err != (int)err_in) ok = false;
    }
    if (ok) { a_start[jb] = start; a_end[jb] = end_site; a_nm[jb] = err_in; a_score[jb] = score; a_nops[jb] = 0; }
    else { need_sw[jb] = 1; if (counters) atomicAdd(&SHARD(counters)[4], 1ull); }
}

// the same kernel over packed rows: 64 bytes of read instead of 160, the window compare a whole-word XOR (32 positions per step),
// the quality row touched only at the mismatching positions, the ASCII row only where a dirty read might hold an 'N'
__global__ void __launch_bounds__(256)
k_align_ungapped_p(DevIndex ix, ScoreParams sp, const int* __restrict__ pen_lut, const char* __restrict__ seq, PackedRows pr,
                   const char* __restrict__ qual, const char* __restrict__ qual2, ReadGeom gm, int stride, u64 n_jobs, const u64* __restrict__ n_jobs_dev, Jobs jb_, u32 rev_qual_from,
                   int* __restrict__ a_start, int* __restrict__ a_end, u32* __restrict__ a_nm, int* __restrict__ a_score,
                   int* __restrict__ a_nops, u32* __restrict__ need_sw, unsigned long long* __restrict__ counters)
{
    const u64 jb = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_jobs_dev) { const u64 nd = *n_jobs_dev; if (nd < n_jobs) n_jobs = nd; }
    if (jb >= n_jobs) return;
    const u32 r = jb_.read[jb];
    const u64 site = jb_.site[jb];
    const int end_site = jb_.end[jb];
    const u32 err_in = jb_.err[jb];
    const int L = gm.rl(r), k = gm.rk(L);
    need_sw[jb] = 0;
    if (err_in == 0) {          // fast_recalculate_bs_Cigar's own err == 0 branch (ksw.cpp:2607-2616)
        a_start[jb] = end_site - L + 1; a_end[jb] = end_site; a_nm[jb] = 0; a_score[jb] = 0; a_nops[jb] = 0;
        return;
    }
    const u64* row = pr.base + (size_t)r * pr.pwords;
    const bool dirty = pr.dirty[r] != 0;
    const char* ql = qual_row(qual, qual2, rev_qual_from, r, stride);
    const bool rev = r >= rev_qual_from;
    const bool wvalid = window_valid(ix, site, (u64)(L + 2 * k), site < ix.G);
    const int start = end_site - L + 1;
    bool ok = start >= 0 && wvalid;
    int tmp_err = 0, score = 0;
    if (ok) {
        GenStream gs; gs.init(ix, site + (u64)start);
        for (int p = 0; p < L && ok; p += 32) {
            u64 mm = mism_bs(row[p >> 5], gs.next32());
            if (dirty) mm |= spread32((u32)((row[pr.W + (p >> 6)] >> (p & 63)) & 0xffffffffull));
            mm &= field_range(0, L - p);
            tmp_err += __popcll(mm);
            if (tmp_err > (int)err_in) { ok = false; break; }
            while (mm) {
                const int i = p + (__ffsll((unsigned long long)mm) - 1) / 2;
                mm &= mm - 1;
                const bool isN = dirty && ((row[pr.W + (i >> 6)] >> (i & 63)) & 1) && seq[(size_t)r * stride + i] == 'N';
                score -= isN ? sp.np : pen_lut[(unsigned char)ql[rev ? L - 1 - i : i]];
            }
        }
        if (ok && tmp_err != (int)err_in) ok = false;
    }
    if (ok) { a_start[jb] = start; a_end[jb] = end_site; a_nm[jb] = err_in; a_score[jb] = score; a_nops[jb] = 0; }
    else { need_sw[jb] = 1; if (counters) atomicAdd(&SHARD(counters)[4], 1ull); }
}

// UNIFORM: all reads of the launch have one length, so k is a kernel argument (a scalar register) and the band tests of the
// unrolled loop are scalar branches; with per-read lengths they are per-lane and cost an exec-mask save/restore per cell.
// EXACT (with UNIFORM): k == KB, the band width is a compile-time constant (see k_align_sw2).
template <int KB, bool UNIFORM, bool EXACT = false>
__global__ void __launch_bounds__(64)
k_align_sw(DevIndex ix, ScoreParams sp, const int* __restrict__ pen_lut, const char* __restrict__ seq,
           const char* __restrict__ qual, const char* __restrict__ qual2, ReadGeom gm, int stride, const u64* __restrict__ n_sw_ptr,
           const u32* __restrict__ sw_job, Jobs jb_, u32 rev_qual_from, u64* __restrict__ trace, u64 trace_stride, u64 job_base,
           u32* __restrict__ cigar_pool, int max_ops,
           int* __restrict__ a_start, int* __restrict__ a_end, u32* __restrict__ a_nm, int* __restrict__ a_score,
           int* __restrict__ a_nops, PackedRows pr)
{
    constexpr int BW = 2 * KB + 1;          // compile-time bound of the band width
    constexpr int NW = (BW + 15) / 16;      // trace words per row (4 bits per cell)
    // a thread owns trace slot `slot` and takes job job_base + slot: the host issues one launch per trace_stride jobs of its upper
    // bound, so the trace buffer is sized by the launch (at most 1 M slots), not by the number of jobs (which only the device
    // knows; launches beyond it find nothing to do)
    const u64 slot = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 t = job_base + slot;
    if ((u64)blockIdx.x * blockDim.x + job_base >= *n_sw_ptr) return;          // the whole wave has nothing to do
    // the mismatch penalty of a row is on the critical path of all its cells: the 256-entry table lives in LDS, not behind a
    // global gather per row
    __shared__ int s_pen[256];
    for (int q = threadIdx.x; q < 256; q += 64) s_pen[q] = pen_lut[q];
    __syncthreads();
    if (t >= *n_sw_ptr) return;
    const WaveLogT wl_t = wavelog_begin();
    const u64 jb = sw_job[t];
    const u32 r = jb_.read[jb];
    const u64 site = jb_.site[jb];
    const char* rd = seq + (size_t)r * stride;
    const char* ql = qual_row(qual, qual2, rev_qual_from, r, stride);
    const bool rev = r >= rev_qual_from;
    const bool fwd = site < ix.G;
    const int L = UNIFORM ? gm.L : gm.rl(r), k = EXACT ? KB : UNIFORM ? gm.k : gm.rk(L);          // k <= KB: the unrolled band is masked to the job's own width
    const int band = 2 * k + 1;
    const int p_len = L + 2 * k, tlen = L;
    const bool wvalid = window_valid(ix, site, (u64)p_len, fwd);
    const int MINUS_INF = -0x40000000;
    const int gapoe = sp.gap_open + sp.gap_ext, gape = sp.gap_ext;
    int RH[BW + 1], RE[BW + 1];
#pragma unroll
    for (int b = 0; b <= BW; b++) { RH[b] = b < band ? 0 : MINUS_INF; RE[b] = b < band ? -gapoe : MINUS_INF; }
    // window bases i .. i+band-1 of the current row as 3-bit codes (4 = out-of-strand) in two u64 x 2 ... keep
    // it simple: 4 bits per base, up to 63 bases -> four u64
    u64 wq[(BW + 15) / 16];
#pragma unroll
    for (int q = 0; q < NW; q++) wq[q] = 0;
    extern __shared__ u64 lds_win[];        // [(gm.L + 2 gm.k + 62) / 32 + 1][64]: sized for the longest read of the batch
    LdsWin wr; wr.init(ix, site, wvalid, lds_win + threadIdx.x, (p_len + 62) / 32);
    for (int b = 0; b < band; b++) { const u64 v = (u64)wr.next(); wq[b >> 4] |= v << (4 * (b & 15)); }
    u64* tz = trace + slot;                 // word (i*NWk + q) lives at tz[(i*NWk + q) * trace_stride]
    const int NWk = (band + 15) / 16;
    int h1_last = MINUS_INF;
    // the read's letters: from the packed row (2 bits per base) when the batch has one, else from the ASCII row
    const u64* prow = pr.base ? pr.base + (size_t)r * pr.pwords : nullptr;
    const bool pdirty = pr.base ? pr.dirty[r] != 0 : false;
    ReadCur rcur; PCode pcur;
    if (prow) pcur.seek(prow, pr.W, pdirty); else rcur.seek(rd, 0, L);
    ReadCur qcur; RevCur qrev;
    if (!rev) qcur.seek(ql, 0, L); else qrev.seek(ql, L - 1);
    // the trace words of a row are stored one row later, after the next row's loads (see k_align_sw2)
    u64 tw[NW];
#pragma unroll
    for (int q = 0; q < NW; q++) tw[q] = 0;
    for (int i = 0; i < tlen; ++i) {
        int f = MINUS_INF, h1 = MINUS_INF;
        if (i > 0) {
            // slide the window one base
#pragma unroll
            for (int q = 0; q < NW; q++) { wq[q] >>= 4; if (q + 1 < NW) wq[q] |= (wq[q + 1] & 15) << 60; }
            const u64 v = (u64)wr.next();
#pragma unroll
            for (int q = 0; q < NW; q++) if (q == ((band - 1) >> 4)) wq[q] |= v << (4 * ((band - 1) & 15));
        }
        const int ta = prow ? pcur.next4() : code4(rcur.next());
        const unsigned char qc = rev ? qrev.next() : (unsigned char)qcur.next();
        const int mis = ta == 4 ? -sp.np : -s_pen[qc];
        if (i > 0) {
#pragma unroll
            for (int q = 0; q < NW; q++) if (q < NWk) tz[((u64)(i - 1) * NWk + q) * trace_stride] = tw[q];
        }
#pragma unroll
        for (int q = 0; q < NW; q++) tw[q] = 0;
#pragma unroll
        for (int b = 0; b < BW; b++) {
            if (b < band) {
                int m = RH[b], e = RE[b], h, tt;
                const int wb = (int)((wq[b >> 4] >> (4 * (b & 15))) & 15);
                m += ((ta == wb && ta < 4) || (ta == 3 && wb == 1)) ? 0 : (wb == 4 ? -sp.np : mis);      // mat[] of Schema.cpp:830-850: N never matches
                int d = m >= e ? 0 : 1;
                h = m >= e ? m : e;
                d = h >= f ? d : 2;
                h = h >= f ? h : f;
                tt = m - gapoe;
                e -= gape;
                d |= e > tt ? 4 : 0;
                e = e > tt ? e : tt;
                f -= gape;
                d |= f > tt ? 8 : 0;
                f = f > tt ? f : tt;
                // eh[j].h = h1 (H(i, j-1)), eh[j].e = e: next row reads them one slot to the left
                if (b > 0) { RH[b - 1] = h1; RE[b - 1] = e; }
                h1 = h;
                tw[b >> 4] |= (u64)d << (4 * (b & 15));
            }
        }
        // eh[end] = { h1, -inf }
#pragma unroll
        for (int b = 0; b <= BW; b++) if (b == band - 1) { RH[b] = h1; RE[b] = MINUS_INF; }
        h1_last = h1;
    }
    if (tlen > 0) {
#pragma unroll
        for (int q = 0; q < NW; q++) if (q < NWk) tz[((u64)(tlen - 1) * NWk + q) * trace_stride] = tw[q];
    }
    (void)h1_last;
    // score: un-gapped diagonal wins ties, then the highest column (ksw.cpp:2001-2010)
    int max_i = tlen + k, score = MINUS_INF;
#pragma unroll
    for (int b = 0; b < BW; b++) if (b == k) score = RH[b];
#pragma unroll
    for (int bp = BW; bp >= 1; bp--) if (bp <= band) { const int h = RH[bp - 1]; if (h > score) { score = h; max_i = tlen - 1 + bp; } }
    int qe = max_i - 1;
    // traceback
    const int LOCAL_OPS = 256;            // >= the 254 operations a record can hold (cigar_ops_bound)
    u32 cg[LOCAL_OPS + 1];
    int nc = 0;
    bool overflow = false;
    // the run being extended stays in registers (cur_op, cur_len); it goes to cg[] only when the operation changes
    int cur_op = -1, cur_len = 0;
    auto flush = [&]() { if (cur_op >= 0) { if (nc < LOCAL_OPS) cg[nc++] = ((u32)cur_len << 4) | (u32)cur_op; else overflow = true; } };
    auto push = [&](int op, int len) {
        if (op != cur_op) { flush(); cur_op = op; cur_len = len; }
        else cur_len += len;
    };
    int i = tlen - 1, kk = max_i - 1, which = 0;
    {
        // the rows are visited in descending order and the path rarely changes its trace-word column: an 8-deep software pipeline
        // of row loads for the current column (each step would otherwise wait out a full memory round trip), refilled when the
        // column changes
        int q = (kk - i) >> 4;
        u64 tb[8];
        auto refill = [&]() {
#pragma unroll
            for (int p = 0; p < 8; p++) tb[p] = i - p >= 0 ? tz[((u64)(i - p) * NWk + q) * trace_stride] : 0;
        };
        refill();
        while (i >= 0 && kk >= 0) {
            const int b = kk - i;
            if ((b >> 4) != q) { q = b >> 4; refill(); }
            const int d = (int)((tb[0] >> (4 * (b & 15))) & 15);
            which = which == 0 ? (d & 3) : which == 1 ? ((d >> 2) & 1) : ((d >> 3) & 1) * 2;
            if (which == 2) { push(1, 1); --kk; continue; }
            if (which == 0) { push(0, 1); --kk; } else push(2, 1);
            --i;
#pragma unroll
            for (int p = 0; p < 7; p++) tb[p] = tb[p + 1];
            tb[7] = i - 7 >= 0 ? tz[((u64)(i - 7) * NWk + q) * trace_stride] : 0;
        }
    }
    if (i >= 0) push(2, i + 1);
    flush();
    for (int a2 = 0, b2 = nc - 1; a2 < b2; a2++, b2--) { const u32 x = cg[a2]; cg[a2] = cg[b2]; cg[b2] = x; }
    cg[nc] = 0;
    int qb = kk + 1;
    // K13: fold leading / trailing insertions into M (ksw.cpp:2677-2772)
    int n_cigar = nc, ii, op, opl, ins = 0;
    for (ii = 0; ii < n_cigar; ++ii) { op = cg[ii] & 0xf; opl = cg[ii] >> 4; if (op != 2) break; ins += opl; }
    if (ii != 0) {
        op = cg[ii] & 0xf; opl = cg[ii] >> 4;
        if (op == 0) opl += ins; else { op = 0; opl = ins; ii--; }
        cg[ii] = ((u32)opl << 4) | (u32)op;
        qb -= ins;
    }
    const int cigar_b = ii;
    ins = 0;
    for (ii = n_cigar - 1; ii >= cigar_b; --ii) { op = cg[ii] & 0xf; opl = cg[ii] >> 4; if (op != 2) break; ins += opl; }
    if (ii != n_cigar - 1) {
        op = cg[ii] & 0xf; opl = cg[ii] >> 4;
        if (op == 0) opl += ins; else { op = 0; opl = ins; ii++; }
        cg[ii] = ((u32)opl << 4) | (u32)op;
        qe += ins;
    }
    const int cigar_e = ii;
    // NM recount under bisulfite matching + ops in SAM order (ksw.cpp:2779-2857); a window never holds 'N'
    // mismatches of an M run: read [ts, ts + len) against window [qs, qs + len), eight positions per step
    auto m_run = [&](int ts, int qs, int len) -> int {
        if (!wvalid) return len;
        if (prow) return count_mism_p(ix, prow, pr.W, pdirty, ts, site + (u64)qs, len);
        int c = 0;
        for (int o = 0; o < len; o += 8) c += mism_span(ix, rd, ts + o, site + (u64)(qs + o), len - o < 8 ? len - o : 8);
        return c;
    };
    u32* ops_out = cigar_pool + jb * (u64)max_ops;
    int NM = 0, no = 0;
    if (fwd) {
        int qs = qb, ts = 0;
        for (ii = cigar_b; ii <= cigar_e; ++ii) {
            op = cg[ii] & 0xf; opl = cg[ii] >> 4;
            if (no < max_ops) ops_out[no] = cg[ii]; else overflow = true;
            no++;
            if (op == 0) { NM += m_run(ts, qs, opl); qs += opl; ts += opl; }
            else if (op == 1) { qs += opl; NM += opl; }
            else { ts += opl; NM += opl; }
        }
    } else {
        int qx = qe, te = tlen - 1;
        for (ii = cigar_e; ii >= cigar_b; --ii) {
            op = cg[ii] & 0xf; opl = cg[ii] >> 4;
            if (no < max_ops) ops_out[no] = cg[ii]; else overflow = true;
            no++;
            if (op == 0) { NM += m_run(te - opl + 1, qx - opl + 1, opl); qx -= opl; te -= opl; }
            else if (op == 1) { qx -= opl; NM += opl; }
            else { te -= opl; NM += opl; }
        }
    }
    a_start[jb] = qb; a_end[jb] = qe; a_nm[jb] = (u32)NM; a_score[jb] = score; a_nops[jb] = overflow ? -1 : no;
    wavelog_end(wl_t, 1);
}

// ------------------------------------------------------------------------------------------------
// k_align_sw2<KB>: k_align_sw with TWO alignments per lane in packed 16-bit arithmetic (v_pk_add/sub/max_i16).
// The register-band DP is VALU-issue bound (1.3 k wave instructions per job at k = 12: 0.65 ms per 300 k jobs is the chip's
// issue peak), so the lever is instructions per cell.  Job A lives in the low halves of every H / E / F register, job B in the
// high halves; one packed instruction advances both.  What does not pack is made cheap: the match test of a whole band row is
// a handful of 64-bit logic ops per job (window bases one-hot in nibbles, AND with the set of bases the read letter accepts,
// nibble -> bit), each cell then takes its score from a sign-extended bit field; the four comparison results of a cell are the
// sign bits of four packed differences, collected into the trace byte (low nibble job A, high nibble job B) without a compare.
// Scores are exact in 16 bits as long as L * max(penalty) + gap costs stay below 12 000 and the band's "minus infinity"
// (-16 000) cannot wrap (the host checks; otherwise, and for batches of mixed read lengths, k_align_sw runs).  Same trace
// volume as k_align_sw (8 bits per cell pair), same traceback, same results.
typedef short bmbs_s2 __attribute__((ext_vector_type(2)));
DEVI u32 pk_add(u32 a, u32 b) { bmbs_s2 x = __builtin_bit_cast(bmbs_s2, a) + __builtin_bit_cast(bmbs_s2, b); return __builtin_bit_cast(u32, x); }
DEVI u32 pk_sub(u32 a, u32 b) { bmbs_s2 x = __builtin_bit_cast(bmbs_s2, a) - __builtin_bit_cast(bmbs_s2, b); return __builtin_bit_cast(u32, x); }
DEVI u32 pk_max(u32 a, u32 b) { bmbs_s2 x = __builtin_elementwise_max(__builtin_bit_cast(bmbs_s2, a), __builtin_bit_cast(bmbs_s2, b)); return __builtin_bit_cast(u32, x); }
DEVI u32 pk_make(int lo, int hi) { return ((u32)lo & 0xffffu) | ((u32)hi << 16); }
DEVI int pk_lo(u32 x) { return (int)(short)(x & 0xffffu); }
DEVI int pk_hi(u32 x) { return (int)x >> 16; }
#define SW2_MINF (-16000)
// the trace nibble (bit 0 m < e, bit 1 h < f, bit 2 e extended, bit 3 f extended) of band cell b, job j, from the u64 trace word
// that holds cells 8 (b >> 3) .. + 7 (layout: see the cell loop of k_align_sw2)
DEVI int sw2_trace_nibble(u64 word, int b, int j)
{
    const u32 w = (u32)(word >> (32 * ((b >> 2) & 1))) >> (b & 3);
    const int nb = 8 * j;                       // job B's four nibbles sit two nibbles above job A's
    return (int)(((w >> nb) & 1u) | (((w >> (nb + 16)) & 1u) << 1) | (((w >> (nb + 4)) & 1u) << 2) | (((w >> (nb + 20)) & 1u) << 3));
}

// EXACT: the batch's threshold k equals KB, so the band width is a compile-time constant and every "is this cell inside the
// band" select / branch of the unrolled row disappears (the usual case: KB is instantiated for the thresholds the default -e
// values give).  PACKED: the read letters come from the packed rows (the product path; ASCII rows only with BMBS_ROWS=ascii).
template <int KB, bool EXACT = false, bool PACKED = true>
__global__ void __launch_bounds__(64)
k_align_sw2(DevIndex ix, ScoreParams sp, const int* __restrict__ pen_lut, const char* __restrict__ seq,
            const char* __restrict__ qual, const char* __restrict__ qual2, ReadGeom gm, int stride, const u64* __restrict__ n_sw_ptr,
            const u32* __restrict__ sw_job, Jobs jb_, u32 rev_qual_from, u64* __restrict__ trace, u64 trace_stride, u64 job_base,
            u32* __restrict__ cigar_pool, int max_ops,
            int* __restrict__ a_start, int* __restrict__ a_end, u32* __restrict__ a_nm, int* __restrict__ a_score,
            int* __restrict__ a_nops, PackedRows pr)
{
    constexpr int BW = 2 * KB + 1;          // compile-time bound of the band width
    constexpr int NW = (BW + 15) / 16;      // window words per job (4 bits per base)
    constexpr int NT = (BW + 7) / 8;        // trace words per row (8 bits per cell: both jobs)
    const u64 slot = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n_sw = *n_sw_ptr;
    const u64 tA = job_base + 2 * slot;
    if (job_base + 2 * (u64)blockIdx.x * blockDim.x >= n_sw) return;            // the whole wave has nothing to do
    __shared__ int s_pen[256];                                                  // as in k_align_sw
    for (int q = threadIdx.x; q < 256; q += 64) s_pen[q] = pen_lut[q];
    __syncthreads();
    if (tA >= n_sw) return;
    const WaveLogT wl_t = wavelog_begin();
    const bool haveB = tA + 1 < n_sw;
    const int L = gm.L, k = EXACT ? KB : gm.k;           // uniform batch: one length, one threshold
    const int band = 2 * k + 1, tlen = L;
    const int NTk = (band + 7) / 8;
    u64 jbv[2]; u32 rv[2]; u64 sitev[2]; const char* rdv[2]; const char* qlv[2]; bool revv[2], fwdv[2], wval[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        jbv[j] = sw_job[(j == 1 && haveB) ? tA + 1 : tA];
        rv[j] = jb_.read[jbv[j]];
        sitev[j] = jb_.site[jbv[j]];
        rdv[j] = seq + (size_t)rv[j] * stride;
        qlv[j] = qual_row(qual, qual2, rev_qual_from, rv[j], stride);
        revv[j] = rv[j] >= rev_qual_from;
        fwdv[j] = sitev[j] < ix.G;
        wval[j] = window_valid(ix, sitev[j], (u64)(L + 2 * k), fwdv[j]);
    }
    const int gapoe = sp.gap_open + sp.gap_ext, gape = sp.gap_ext;
    const u32 gapoeP = pk_make(gapoe, gapoe), gapeP = pk_make(gape, gape);
    const u32 MINFP = pk_make(SW2_MINF, SW2_MINF);
    u32 RH[BW + 1], RE[BW + 1];
#pragma unroll
    for (int b = 0; b <= BW; b++) { RH[b] = b < band ? 0u : MINFP; RE[b] = b < band ? pk_make(-gapoe, -gapoe) : MINFP; }
    // window bases of the current row, one-hot in nibbles (A1 C2 G4 T8, out-of-strand 0)
    u64 wq[2][NW];
    extern __shared__ u64 lds_win[];        // [2][(L + 2k + 62) / 32 + 1][64]
    const int nww = (L + 2 * k + 62) / 32;
    LdsWin wr[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int q = 0; q < NW; q++) wq[j][q] = 0;
        wr[j].init(ix, sitev[j], wval[j], lds_win + (size_t)j * (nww + 1) * 64 + threadIdx.x, nww);
        for (int b = 0; b < band; b++) { const int v = wr[j].next(); const u64 oh = v < 4 ? (1ull << v) : 0ull; wq[j][b >> 4] |= oh << (4 * (b & 15)); }
    }
    u64* tz = trace + slot;                 // word (i*NTk + q) lives at tz[(i*NTk + q) * trace_stride]
    ReadCur rcur[2], qcur[2];
    RevCur qrev[2];
    PCode pcur[2];
    constexpr bool packed_in = PACKED;
    const u64* prowv[2] = {nullptr, nullptr};
    bool pdirtyv[2] = {false, false};
#pragma unroll
    for (int j = 0; j < 2; j++) {
        if (packed_in) { prowv[j] = pr.base + (size_t)rv[j] * pr.pwords; pdirtyv[j] = pr.dirty[rv[j]] != 0; pcur[j].seek(prowv[j], pr.W, pdirtyv[j]); }
        else rcur[j].seek(rdv[j], 0, L);
        if (!revv[j]) qcur[j].seek(qlv[j], 0, L); else qrev[j].seek(qlv[j], L - 1);
    }
    // Order inside a row: slide the windows and read the row's letter / quality (global loads, each followed by a
    // wait on vmcnt, which on gfx9 counts stores too), THEN store the trace words of the PREVIOUS row, then the cells.  With the
    // stores at the end of their own row every such wait also sat out the acknowledgement of stores issued a few instructions
    // earlier: 53 % of the wave cycles of this kernel were SQ_WAIT_ANY.  Now a store has a whole row of cell arithmetic
    // (plus the other waves' turns) behind it before anything waits.
    u64 tw[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) tw[q] = 0;
    for (int i = 0; i < tlen; ++i) {
        if (i > 0) {
            // slide both windows one base
#pragma unroll
            for (int j = 0; j < 2; j++) {
#pragma unroll
                for (int q = 0; q < NW; q++) { wq[j][q] >>= 4; if (q + 1 < NW) wq[j][q] |= (wq[j][q + 1] & 15) << 60; }
                const int v = wr[j].next();
                const u64 oh = v < 4 ? (1ull << v) : 0ull;
#pragma unroll
                for (int q = 0; q < NW; q++) if (q == ((band - 1) >> 4)) wq[j][q] |= oh << (4 * ((band - 1) & 15));
            }
        }
        // per-row, per-job: mismatch penalty and the match bits of the whole band
        int mis[2];
        u64 Y[2][NW];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int ta = packed_in ? pcur[j].next4() : code4(rcur[j].next());
            const unsigned char qc = revv[j] ? qrev[j].next() : (unsigned char)qcur[j].next();
            mis[j] = (ta == 4 || !wval[j]) ? -sp.np : -s_pen[qc];
            const u64 racc = (u64)((0x0A421u >> (4 * ta)) & 15u) * 0x1111111111111111ull;      // read A C G T N accepts {A} {C} {G} {T,C} {}
#pragma unroll
            for (int q = 0; q < NW; q++) {
                u64 x = wq[j][q] & racc;
                x |= x >> 1; x |= x >> 2;
                Y[j][q] = x & 0x1111111111111111ull;
            }
        }
        const u32 misP = pk_make(mis[0], mis[1]);
        u32 f = MINFP, h1 = MINFP;
        if (i > 0) {
#pragma unroll
            for (int q = 0; q < NT; q++) if (q < NTk) tz[((u64)(i - 1) * NTk + q) * trace_stride] = tw[q];
        }
#pragma unroll
        for (int q = 0; q < NT; q++) tw[q] = 0;
#pragma unroll
        for (int b = 0; b < BW; b++) {
            if (b < band) {
                // score of the cell pair: 0 where the job's match bit is set, the row's penalty elsewhere
                const u32 yA = (u32)(Y[0][b >> 4] >> (32 * ((b & 15) >> 3))), yB = (u32)(Y[1][b >> 4] >> (32 * ((b & 15) >> 3)));
                // (asm: left to itself the compiler turns "sign-extended bit & constant" into and + compare + wait state +
                // select, eight instructions per cell pair where two v_bfe_i32, one v_bfi_b32 and one and-not do)
                int mA, mB;
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mA) : "v"(yA), "n"(4 * (b & 7)));
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mB) : "v"(yB), "n"(4 * (b & 7)));
                const u32 mk = ((u32)mA & 0xffffu) | ((u32)mB & 0xffff0000u);          // (two v_bfi_b32 in asm instead: measured no faster)
                const u32 sc = misP & ~mk;
                const u32 m = pk_add(RH[b], sc);
                u32 e = RE[b];
                const u32 t1 = pk_sub(m, e);                 // sign: m < e
                u32 h = pk_max(m, e);
                const u32 t2 = pk_sub(h, f);                 // sign: h < f
                h = pk_max(h, f);
                const u32 tt = pk_sub(m, gapoeP);
                e = pk_sub(e, gapeP);
                const u32 t3 = pk_sub(tt, e);                // sign: e > tt
                e = pk_max(e, tt);
                f = pk_sub(f, gapeP);
                const u32 t4 = pk_sub(tt, f);                // sign: f > tt
                f = pk_max(f, tt);
                if (b > 0) { RH[b - 1] = h1; RE[b - 1] = e; }
                h1 = h;
                // The four comparison results of both jobs are the sign bits of t1..t4 (bits 15 and 31).  Two v_perm_b32 bring the eight
                // bytes that hold them together, two shift + mask steps put one sign per nibble, and the cell's eight flags go to bit
                // (b & 3) of the eight nibbles of the 32-bit trace word that four consecutive cells share -- seven instructions
                // per cell pair where shifting and masking the four differences one by one took twelve.  Nibble order in the word:
                // t1A t3A t1B t3B t2A t4A t2B t4B (t1: m < e, t2: h < f, t3: e extended, t4: f extended); sw2_trace_nibble undoes it.
                const u32 P = __builtin_amdgcn_perm(t2, t1, 0x07050301u);          // bytes t1.1 t1.3 t2.1 t2.3
                const u32 Q = __builtin_amdgcn_perm(t4, t3, 0x07050301u);          // bytes t3.1 t3.3 t4.1 t4.3
                const u32 x = ((P >> 7) & 0x01010101u) | ((Q >> 3) & 0x10101010u);
                if ((b >> 2) & 1) tw[b >> 3] |= (u64)(x << (b & 3)) << 32; else tw[b >> 3] |= (u64)(x << (b & 3));
            }
        }
#pragma unroll
        for (int b = 0; b <= BW; b++) if (b == band - 1) { RH[b] = h1; RE[b] = MINFP; }
    }
    if (tlen > 0) {
#pragma unroll
        for (int q = 0; q < NT; q++) if (q < NTk) tz[((u64)(tlen - 1) * NTk + q) * trace_stride] = tw[q];
    }
    // per job: score, traceback, CIGAR, NM -- as k_align_sw, on this job's half of the registers and nibbles of the trace
    for (int j = 0; j < 2; j++) {
        if (j == 1 && !haveB) break;
        const u64 jb = jbv[j];
        const char* rd = rdv[j];
        const u64 site = sitev[j];
        const bool fwd = fwdv[j], wvalid = wval[j];
        auto half = [&](u32 x) -> int { return j ? pk_hi(x) : pk_lo(x); };
        // score: un-gapped diagonal wins ties, then the highest column (ksw.cpp:2001-2010)
        int max_i = tlen + k, score = SW2_MINF;
#pragma unroll
        for (int b = 0; b < BW; b++) if (b == k) score = half(RH[b]);
#pragma unroll
        for (int bp = BW; bp >= 1; bp--) if (bp <= band) { const int h = half(RH[bp - 1]); if (h > score) { score = h; max_i = tlen - 1 + bp; } }
        int qe = max_i - 1;
        const int LOCAL_OPS = 256;            // >= the 254 operations a record can hold (cigar_ops_bound)
        u32 cg[LOCAL_OPS + 1];
        int nc = 0;
        bool overflow = false;
        int cur_op = -1, cur_len = 0;
        auto flush = [&]() { if (cur_op >= 0) { if (nc < LOCAL_OPS) cg[nc++] = ((u32)cur_len << 4) | (u32)cur_op; else overflow = true; } };
        auto push = [&](int op, int len) {
            if (op != cur_op) { flush(); cur_op = op; cur_len = len; }
            else cur_len += len;
        };
        int i = tlen - 1, kk = max_i - 1, which = 0;
        {
            // 8-deep pipeline of row loads for the path's current trace-word column (as in k_align_sw)
            int q = (kk - i) >> 3;
            u64 tb[8];
            auto refill = [&]() {
#pragma unroll
                for (int p = 0; p < 8; p++) tb[p] = i - p >= 0 ? tz[((u64)(i - p) * NTk + q) * trace_stride] : 0;
            };
            refill();
            while (i >= 0 && kk >= 0) {
                const int b = kk - i;
                if ((b >> 3) != q) { q = b >> 3; refill(); }
                const int d = sw2_trace_nibble(tb[0], b, j);
                which = which == 0 ? ((d & 2) ? 2 : (d & 1)) : which == 1 ? ((d >> 2) & 1) : ((d >> 3) & 1) * 2;
                if (which == 2) { push(1, 1); --kk; continue; }
                if (which == 0) { push(0, 1); --kk; } else push(2, 1);
                --i;
#pragma unroll
                for (int p = 0; p < 7; p++) tb[p] = tb[p + 1];
                tb[7] = i - 7 >= 0 ? tz[((u64)(i - 7) * NTk + q) * trace_stride] : 0;
            }
        }
        if (i >= 0) push(2, i + 1);
        flush();
        for (int a2 = 0, b2 = nc - 1; a2 < b2; a2++, b2--) { const u32 x = cg[a2]; cg[a2] = cg[b2]; cg[b2] = x; }
        cg[nc] = 0;
        int qb = kk + 1;
        // K13: fold leading / trailing insertions into M (ksw.cpp:2677-2772)
        int n_cigar = nc, ii, op, opl, ins = 0;
        for (ii = 0; ii < n_cigar; ++ii) { op = cg[ii] & 0xf; opl = cg[ii] >> 4; if (op != 2) break; ins += opl; }
        if (ii != 0) {
            op = cg[ii] & 0xf; opl = cg[ii] >> 4;
            if (op == 0) opl += ins; else { op = 0; opl = ins; ii--; }
            cg[ii] = ((u32)opl << 4) | (u32)op;
            qb -= ins;
        }
        const int cigar_b = ii;
        ins = 0;
        for (ii = n_cigar - 1; ii >= cigar_b; --ii) { op = cg[ii] & 0xf; opl = cg[ii] >> 4; if (op != 2) break; ins += opl; }
        if (ii != n_cigar - 1) {
            op = cg[ii] & 0xf; opl = cg[ii] >> 4;
            if (op == 0) opl += ins; else { op = 0; opl = ins; ii++; }
            cg[ii] = ((u32)opl << 4) | (u32)op;
            qe += ins;
        }
        const int cigar_e = ii;
        auto m_run = [&](int ts, int qs, int len) -> int {
            if (!wvalid) return len;
            if (packed_in) return count_mism_p(ix, prowv[j], pr.W, pdirtyv[j], ts, site + (u64)qs, len);
            int c = 0;
            for (int o = 0; o < len; o += 8) c += mism_span(ix, rd, ts + o, site + (u64)(qs + o), len - o < 8 ? len - o : 8);
            return c;
        };
        u32* ops_out = cigar_pool + jb * (u64)max_ops;
        int NM = 0, no = 0;
        if (fwd) {
            int qs = qb, ts = 0;
            for (ii = cigar_b; ii <= cigar_e; ++ii) {
                op = cg[ii] & 0xf; opl = cg[ii] >> 4;
                if (no < max_ops) ops_out[no] = cg[ii]; else overflow = true;
                no++;
                if (op == 0) { NM += m_run(ts, qs, opl); qs += opl; ts += opl; }
                else if (op == 1) { qs += opl; NM += opl; }
                else { ts += opl; NM += opl; }
            }
        } else {
            int qx = qe, te = tlen - 1;
            for (ii = cigar_e; ii >= cigar_b; --ii) {
                op = cg[ii] & 0xf; opl = cg[ii] >> 4;
                if (no < max_ops) ops_out[no] = cg[ii]; else overflow = true;
                no++;
                if (op == 0) { NM += m_run(te - opl + 1, qx - opl + 1, opl); qx -= opl; te -= opl; }
                else if (op == 1) { qx -= opl; NM += opl; }
                else { te -= opl; NM += opl; }
            }
        }
        a_start[jb] = qb; a_end[jb] = qe; a_nm[jb] = (u32)NM; a_score[jb] = score; a_nops[jb] = overflow ? -1 : no;
    }
    wavelog_end(wl_t, 2);
}

// ------------------------------------------------------------------------------------------------
// k_align_sw_wave<LANES>: the same DP as k_align_sw, wave-cooperative with the trace in LDS (north_star (c)).
//
// One alignment per group of LANES lanes (16 / 32 / 64 for k <= 7 / 15 / 31: four, two or one alignment per wave); lane b
// owns band cell b of the current row, i.e. window column j = i + b.  Dependencies of cell (i, j) in ksw_semi_global_quality_back
// (ksw.cpp:1850-2045): the diagonal H(i-1, j-1) is the SAME lane's h of the previous row, E(i, j) was produced by lane b+1 in
// the previous row (one DPP wave shift), and F runs along the row.  The reference opens gaps from the diagonal term alone
// (t = M - gapoe with M = H(i-1,j-1) + s(i,j), not from max(M, E, F)), so the F chain  f(b+1) = max(f(b) - gape, t(b))  is a
// max-plus prefix over values every lane already has:  f(b) = max_{b' < b} (t(b') + gape * b') - gape * (b - 1)  -- one
// wave prefix-max (4-6 DPP steps) replaces the serial sweep of the anti-diagonal formulation, and a whole band row is one
// step of the wave.  H, E, F live in registers; the four trace bits of a cell (ksw.cpp:1966-1990) are packed eight cells to a
// 32-bit word by three DPP shifts and written to LDS (ceil(band/8) words per row: 2.4 KB at L = 150, k = 12; 6 KB at L = 250,
// k = 20), the traceback walks them there, the run-length CIGAR is assembled in LDS, and the only HBM traffic of a job is its
// read row, its window and the ops it emits.  (The register kernel wrote and re-read 1.25 GB of trace words per 10 M-read
// batch: 11.6x its algorithmic bytes, profiles/r01_pmc_fetch_write.csv.)
template <int CTRL, int ROW_MASK>
DEVI int dpp_i32(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false); }

template <int LANES>
DEVI int group_prefix_max(int x, int minf)
{
    x = max(x, dpp_i32<0x111, 0xf>(minf, x));          // row_shr:1
    x = max(x, dpp_i32<0x112, 0xf>(minf, x));          // row_shr:2
    x = max(x, dpp_i32<0x114, 0xf>(minf, x));          // row_shr:4
    x = max(x, dpp_i32<0x118, 0xf>(minf, x));          // row_shr:8
    if (LANES >= 32) x = max(x, dpp_i32<0x142, 0xa>(minf, x));   // row_bcast:15 into rows 1 and 3
    if (LANES >= 64) x = max(x, dpp_i32<0x143, 0xc>(minf, x));   // row_bcast:31 into rows 2 and 3
    return x;
}

#define SWW_CG_WORDS 260
// LDS words per job: row constants (u16 per read position, padded to 4) + trace + CIGAR ops; even, so that jobs stay 8-byte aligned
__host__ __device__ inline int sww_lds_words(int L, int k)
{
    const int rw = (2 * k + 1 + 7) / 8;
    return ((L + 3) / 4) * 2 + L * rw + SWW_CG_WORDS;
}

template <int LANES>
__global__ void __launch_bounds__(64)
k_align_sw_wave(DevIndex ix, ScoreParams sp, const int* __restrict__ pen_lut, const char* __restrict__ seq,
                const char* __restrict__ qual, const char* __restrict__ qual2, ReadGeom gm, int stride, const u64* __restrict__ n_sw_ptr,
                const u32* __restrict__ sw_job, Jobs jb_, u32 rev_qual_from, u32* __restrict__ cigar_pool, int max_ops,
                int* __restrict__ a_start, int* __restrict__ a_end, u32* __restrict__ a_nm, int* __restrict__ a_score,
                int* __restrict__ a_nops, PackedRows pr)
{
    extern __shared__ u32 sww_lds[];
    constexpr int JPB = 64 / LANES;
    const int lane = (int)(threadIdx.x & 63), grp = lane / LANES, b = lane % LANES;
    const u64 n_sw = *n_sw_ptr;
    const u64 t0 = (u64)blockIdx.x * JPB;
    if (t0 >= n_sw) return;                               // the whole wave has nothing to do
    const bool live = t0 + (u64)grp < n_sw;
    const u64 t = live ? t0 + (u64)grp : t0;              // idle groups shadow the block's first job (DPP needs every lane in step)
    const u64 jb = sw_job[t];
    const u32 r = jb_.read[jb];
    const u64 site = jb_.site[jb];
    const char* rd = seq + (size_t)r * stride;
    const char* ql = qual_row(qual, qual2, rev_qual_from, r, stride);
    const bool rev = r >= rev_qual_from;
    const bool fwd = site < ix.G;
    const int L = gm.rl(r), k = gm.rk(L);
    const int band = 2 * k + 1, tlen = L;
    const int RW = (2 * gm.k + 1 + 7) / 8;                // trace words per row, sized for the launch's largest band
    const bool wvalid = window_valid(ix, site, (u64)(L + 2 * k), fwd);
    const int MINUS_INF = -0x40000000;
    const int gapoe = sp.gap_open + sp.gap_ext, gape = sp.gap_ext;
    u32* base = sww_lds + (size_t)grp * sww_lds_words(gm.L, gm.k);
    u16* rc = reinterpret_cast<u16*>(base);                        // row constants: read code | penalty << 3
    u32* tr = base + ((gm.L + 3) / 4) * 2;                         // trace, RW words per row
    u32* cg = tr + (size_t)gm.L * RW;                              // CIGAR ops of the traceback
    for (int i = b; i < ((L + 3) & ~3); i += LANES) {
        u16 v = 4;
        if (i < L) {
            int ta;
            if (pr.base) {
                const u64* prow = pr.base + (size_t)r * pr.pwords;
                ta = (int)((prow[i >> 5] >> (2 * (i & 31))) & 3);
                if (pr.dirty[r] && ((prow[pr.W + (i >> 6)] >> (i & 63)) & 1)) ta = 4;
            } else ta = code4(rd[i]);
            const unsigned char qc = (unsigned char)ql[rev ? L - 1 - i : i];
            const int pen = ta == 4 ? sp.np : pen_lut[qc];
            v = (u16)(ta | (pen << 3));
        }
        rc[i] = v;
    }
    __syncthreads();
    // rows: the longest job of the wave sets the trip count, shorter ones stop updating
    int Lmax = L;
    if (JPB > 1 && gm.len) {
#pragma unroll
        for (int o = LANES; o < 64; o <<= 1) Lmax = max(Lmax, __shfl_xor(Lmax, o));
    }
    WinReader wr; wr.init(ix, site + (u64)b, wvalid);
    const int gb = gape * b;
    const bool in_band = b < band;
    int hprev = 0, e_next = -gapoe;
    u64 rc4 = 0;
    for (int i = 0; i < Lmax; ++i) {
        if ((i & 3) == 0) rc4 = *reinterpret_cast<const u64*>(rc + (i < L ? i : 0));
        const u32 x = (u32)(rc4 >> (16 * (i & 3))) & 0xffffu;
        const int ta = (int)(x & 7), mis = -(int)(x >> 3);
        const int wb = wr.next();
        const int sc = ((ta == wb && ta < 4) || (ta == 3 && wb == 1)) ? 0 : (wb == 4 ? -sp.np : mis);   // mat[] of Schema.cpp:830-850
        const int m = hprev + sc, e = e_next;
        const int tt = m - gapoe;
        const int P = group_prefix_max<LANES>(in_band ? tt + gb : MINUS_INF, MINUS_INF);
        int Pex = dpp_i32<0x138, 0xf>(MINUS_INF, P);                // wave_shr:1
        const int f = b == 0 ? MINUS_INF : Pex - gb + gape;
        int d = m >= e ? 0 : 1;
        int h = m >= e ? m : e;
        d = h >= f ? d : 2;
        h = h >= f ? h : f;
        const int e2 = e - gape;
        d |= e2 > tt ? 4 : 0;
        const int e_new = e2 > tt ? e2 : tt;
        d |= (f - gape) > tt ? 8 : 0;
        const int e_shl = dpp_i32<0x130, 0xf>(MINUS_INF, e_new);    // wave_shl:1: E(i+1, j) comes from lane b+1
        // eight cells per trace word
        int pk = d | (dpp_i32<0x101, 0xf>(0, d) << 4);              // row_shl:1
        pk |= dpp_i32<0x102, 0xf>(0, pk) << 8;                      // row_shl:2
        pk |= dpp_i32<0x104, 0xf>(0, pk) << 16;                     // row_shl:4
        if (i < L) {
            hprev = h;
            e_next = b == band - 1 ? MINUS_INF : e_shl;
            if ((b & 7) == 0 && in_band) tr[(size_t)i * RW + (b >> 3)] = (u32)pk;
        }
    }
    // score: the un-gapped diagonal wins ties, then the highest column (ksw.cpp:2001-2010)
    int best = in_band ? hprev : MINUS_INF;
#pragma unroll
    for (int o = 1; o < LANES; o <<= 1) best = max(best, __shfl_xor(best, o));
    const u64 eq = __ballot(in_band && hprev == best);
    const u64 geq = LANES == 64 ? eq : (eq >> (grp * LANES)) & ((1ull << LANES) - 1);
    const int s_col = ((geq >> k) & 1) ? k : 63 - __clzll((long long)geq);
    __syncthreads();
    if (b != 0 || !live) return;
    // ---- one lane per job from here: traceback over the LDS trace, CIGAR in LDS ----
    const int score = best;
    int qe = tlen - 1 + s_col;
    const int LOCAL_OPS = SWW_CG_WORDS - 4;
    int nc = 0;
    bool overflow = false;
    int cur_op = -1, cur_len = 0;
    auto flush = [&]() { if (cur_op >= 0) { if (nc < LOCAL_OPS) cg[nc++] = ((u32)cur_len << 4) | (u32)cur_op; else overflow = true; } };
    auto push = [&](int op, int len) {
        if (op != cur_op) { flush(); cur_op = op; cur_len = len; }
        else cur_len += len;
    };
    int i = tlen - 1, kk = qe, which = 0;
    while (i >= 0 && kk >= 0) {
        const int bb = kk - i;
        const int d = (int)((tr[(size_t)i * RW + (bb >> 3)] >> (4 * (bb & 7))) & 15);
        which = which == 0 ? (d & 3) : which == 1 ? ((d >> 2) & 1) : ((d >> 3) & 1) * 2;
        if (which == 0) { push(0, 1); --i; --kk; }
        else if (which == 1) { push(2, 1); --i; }
        else { push(1, 1); --kk; }
    }
    if (i >= 0) push(2, i + 1);
    flush();
    for (int a2 = 0, b2 = nc - 1; a2 < b2; a2++, b2--) { const u32 x = cg[a2]; cg[a2] = cg[b2]; cg[b2] = x; }
    cg[nc] = 0;
    int qb = kk + 1;
    // K13: fold leading / trailing insertions into M (ksw.cpp:2677-2772)
    int n_cigar = nc, ii, op, opl, ins = 0;
    for (ii = 0; ii < n_cigar; ++ii) { op = cg[ii] & 0xf; opl = cg[ii] >> 4; if (op != 2) break; ins += opl; }
    if (ii != 0) {
        op = cg[ii] & 0xf; opl = cg[ii] >> 4;
        if (op == 0) opl += ins; else { op = 0; opl = ins; ii--; }
        cg[ii] = ((u32)opl << 4) | (u32)op;
        qb -= ins;
    }
    const int cigar_b = ii;
    ins = 0;
    for (ii = n_cigar - 1; ii >= cigar_b; --ii) { op = cg[ii] & 0xf; opl = cg[ii] >> 4; if (op != 2) break; ins += opl; }
    if (ii != n_cigar - 1) {
        op = cg[ii] & 0xf; opl = cg[ii] >> 4;
        if (op == 0) opl += ins; else { op = 0; opl = ins; ii++; }
        cg[ii] = ((u32)opl << 4) | (u32)op;
        qe += ins;
    }
    const int cigar_e = ii;
    // NM recount under bisulfite matching + ops in SAM order (ksw.cpp:2779-2857); a window never holds 'N'
    auto m_run = [&](int ts, int qs, int len) -> int {
        if (!wvalid) return len;
        if (pr.base) return count_mism_p(ix, pr.base + (size_t)r * pr.pwords, pr.W, pr.dirty[r] != 0, ts, site + (u64)qs, len);
        int c = 0;
        for (int o = 0; o < len; o += 8) c += mism_span(ix, rd, ts + o, site + (u64)(qs + o), len - o < 8 ? len - o : 8);
        return c;
    };
    u32* ops_out = cigar_pool + jb * (u64)max_ops;
    int NM = 0, no = 0;
    if (fwd) {
        int qs = qb, ts = 0;
        for (ii = cigar_b; ii <= cigar_e; ++ii) {
            op = cg[ii] & 0xf; opl = cg[ii] >> 4;
            if (no < max_ops) ops_out[no] = cg[ii]; else overflow = true;
            no++;
            if (op == 0) { NM += m_run(ts, qs, opl); qs += opl; ts += opl; }
            else if (op == 1) { qs += opl; NM += opl; }
            else { ts += opl; NM += opl; }
        }
    } else {
        int qx = qe, te = tlen - 1;
        for (ii = cigar_e; ii >= cigar_b; --ii) {
            op = cg[ii] & 0xf; opl = cg[ii] >> 4;
            if (no < max_ops) ops_out[no] = cg[ii]; else overflow = true;
            no++;
            if (op == 0) { NM += m_run(te - opl + 1, qx - opl + 1, opl); qx -= opl; te -= opl; }
            else if (op == 1) { qx -= opl; NM += opl; }
            else { te -= opl; NM += opl; }
        }
    }
    a_start[jb] = qb; a_end[jb] = qe; a_nm[jb] = (u32)NM; a_score[jb] = score; a_nops[jb] = overflow ? -1 : no;
}

// ================================================================================================
// finalize: MAPQ, placement, stats
// ================================================================================================
struct bmbs_result_dev {   // == bmbs_result (include/bmbs.h), 32 bytes
    u64 pos; u32 cigar_off; int32_t chrom; u16 flag; u16 nm; int16_t score; u8 status; u8 mapq; u8 n_cigar; u8 path; u16 n_cand; u32 tlen;
};
DEVI u16 sat16(u32 v) { return v > 0xffffu ? (u16)0xffffu : (u16)v; }

// mapq_lut[(ed) * (range+1) + sd]: MAP_Calculation (Schema.cpp:168-405) tabulated on the host in
// IEEE double for this k: ed = min(second_best_diff, k+1), sd = clamp(score + range, 0, range).
__global__ void __launch_bounds__(256)
k_finalize(DevIndex ix, ScoreParams sp, const int* __restrict__ pen_lut, const u8* __restrict__ mapq_lut,
           const u32* __restrict__ mapq_off, int unit, const char* __restrict__ seq, const char* __restrict__ qual, ReadGeom gm,
           int stride, long n, ReadState st, const int* __restrict__ a_start, const int* __restrict__ a_end,
           const u32* __restrict__ a_nm, const int* __restrict__ a_score, const int* __restrict__ a_nops,
           int max_ops, u32 cigar_base, int ambiguous_out, const u64* __restrict__ sp0, const u32* __restrict__ hits0,
           bmbs_result_dev* __restrict__ res, unsigned long long* __restrict__ stats)
{
    __shared__ unsigned long long sh[5];
    __shared__ u64 s_cs[BMBS_CS_LDS];
    if (threadIdx.x < 5) sh[threadIdx.x] = 0;
    const u64* cs = chrom_table(ix, s_cs);
    __syncthreads();
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;          // this lane's contribution to the five counters
    if (r < n) {
        bmbs_result_dev o;
        o.pos = 0; o.cigar_off = 0; o.chrom = -1; o.status = 0; o.mapq = 0; o.flag = 0; o.nm = 0; o.score = 0;
        o.n_cigar = 0; o.path = 0; o.n_cand = sat16(st.n_cand[r]); o.tlen = 0;
        const int verdict = st.verdict[r];
        const int L = gm.rl(r), k = gm.rk(L);
        const int range = unit * k;
        mapq_lut += mapq_off[k];                  // MAP_Calculation table of this read's own threshold
        bool have = false, amb = false;
        u64 site = 0; long long start_site = 0, end_site = 0;
        u32 nm = 0; int score = 0; u32 sbd = 0; int mapq = 0;
        if (verdict == 1) { have = true; site = st.exit_site[r]; start_site = 0; end_site = L - 1; nm = 0; score = 0; mapq = 42; o.path = 1; }
        else if (verdict == 2) {
            have = true; site = st.exit_site[r]; start_site = 0; end_site = L - 1; nm = 1; o.path = 2;
            const int mv = st.mm_site[r], ms = mv & 0x7fff;          // bit 15: the read has 'N' there (k_seed_decide)
            score = (mv & 0x8000) ? -sp.np : -pen_lut[(unsigned char)qual[(size_t)r * stride + ms]];
            sbd = 0xffffffffu;
        } else if (verdict == 4) {
            o.status = 2; o.path = 4;
            if (ambiguous_out) {
                // output_ambiguous_exact_map (Schema.cpp:24072-24115): first row in SA order whose placement stays inside
                // its chromosome, MAPQ 1; none -> the read counts as unmapped
                const u64 sp_ = sp0[r];
                u32 nh = hits0[r]; if (nh > 1000u) nh = 1000u;
                o.status = 3;
                for (u32 i = 0; i < nh; i++) {
                    const u64 s_ = ix.total - sa_at(ix, sp_ + i) - (u64)L;
                    u64 loc = s_; int flag;
                    if (loc >= ix.G) { loc = ix.G * 2 - (loc + (u64)(L - 1)) - 1; flag = 16; } else flag = 0;
                    int c = 0;
                    c = chrom_of(cs, ix.n_chrom, loc);
                    if (c >= ix.n_chrom) continue;
                    const u64 pos = loc + 1 - cs[c];
                    if (pos + (u64)(L - 1) > cs[c + 1] - cs[c]) continue;
                    o.pos = pos; o.chrom = c; o.flag = (u16)flag; o.mapq = 1; o.status = 2;
                    break;
                }
            }
        }
        else if (verdict == 3) {
            o.path = 3;
            const int rs = st.red_status[r];
            if (rs == 1 || (rs == 2 && ambiguous_out)) {
                amb = rs == 2;
                have = true; site = st.best_site[r]; sbd = st.sbd[r];
                if (st.job_flag[r]) {
                    const u64 jb = st.job_off[r];
                    start_site = a_start[jb]; end_site = a_end[jb]; nm = a_nm[jb]; score = a_score[jb];
                    const int no = a_nops[jb];
                    o.cigar_off = cigar_base + (u32)(jb * (u64)max_ops);
                    o.n_cigar = no < 0 ? 255 : (u8)no;
                } else { end_site = st.best_end[r]; start_site = end_site - L + 1; nm = 0; score = 0; }
            } else if (rs == 2) o.status = 2;
        }
        if (have) {
            if (verdict != 1) {
                int sd = score + range; if (sd < 0) sd = 0; if (sd > range) sd = range;
                const u32 ed = sbd > (u32)k ? (u32)k + 1 : sbd;
                mapq = mapq_lut[(size_t)ed * (range + 1) + sd];
            }
            // output_sam_end_to_end placement (Schema.cpp:11941-11986)
            u64 loc = site; int flag;
            if (loc >= ix.G) { loc = loc + (u64)end_site; loc = ix.G * 2 - loc - 1; flag = 16; }
            else { loc = loc + (u64)start_site; flag = 0; }
            int c = 0;
            c = chrom_of(cs, ix.n_chrom, loc);
            bool ok = c < ix.n_chrom;
            u64 pos = 0;
            if (ok) {
                pos = loc + 1 - cs[c];
                const u64 clen = cs[c + 1] - cs[c];
                if (pos + (u64)end_site - (u64)start_site > clen) ok = false;
            }
            o.pos = pos; o.chrom = c < ix.n_chrom ? c : -1; o.flag = (u16)flag; o.mapq = (u8)mapq;
            o.nm = (u16)nm; o.score = (int16_t)score;
            o.status = ok ? (amb ? 2 : 1) : 3;
        }
        res[r] = o;
        s0 = 1;
        if (o.status == 1) { s1 = 1; s3 = (u32)L; s4 = (u32)nm; }
        else if (o.status == 2) s2 = 1;
    }
    wave_stats_add(sh, s0, s1, s2, s3, s4);
    __syncthreads();
    if (threadIdx.x < 5 && sh[threadIdx.x]) atomicAdd(&SHARD(stats)[threadIdx.x], sh[threadIdx.x]);
}

// ================================================================================================
// Paired-end fast mode (Map_Pair_Seq_end_to_end_fast, Schema.cpp:18570-19546)
// ================================================================================================
// Reads of a PE batch are rows [0,n) = mate 1 and [n,2n) = mate 2 AS THE REFERENCE'S READER HANDS IT ON:
// reverse complement of the FASTQ record (Process_Reads.cpp:262-267), qualities in FASTQ order.
struct PeCand { u64 site; u32 err; int32_t end; };      // seed_votes fields used by the PE path

// inner_maxDistance_pair / inner_minDistance_pair of pair p (Schema.cpp:18900-18935): the insert bounds widened by
// twice the larger threshold, the lower one also by the longer mate
struct PeIns { int min_ins, max_ins; };
DEVI void pe_bounds(const ReadGeom& gm, const PeIns& pi, long p, long n, long long& maxd, long long& mind, int& large_k)
{
    const int L1 = gm.rl(p), L2 = gm.rl(p + n);
    const int k1 = gm.rk(L1), k2 = gm.rk(L2);
    large_k = k1 > k2 ? k1 : k2;
    maxd = (long long)pi.max_ins + 2LL * large_k;
    mind = (long long)pi.min_ins - 2LL * large_k - (L1 > L2 ? L1 : L2);
}

struct PeState {
    int*  occ;        // per read: best_mapp_occ (>0 verified, -1 to verify, 0 none)
    u32*  len;        // per read: current list length
    u8*   cur;        // per read: 0 list lives in buffer A, 1 in buffer B
    u8*   vround;     // per read: verification round (0 none, 1, 2)
    u8*   dead;       // per pair
    u8*   both;       // per pair: both mates had to be verified
    int*  npair;      // per pair: mapping_pair
    u32*  sbd;        // per pair: second_best_diff
    // --sensitive only
    u8*   first;      // per pair: 0 = mate 1 is finished and verified first, 1 = mate 2
    u8*   full;       // per read: full_seed_id (seeds recorded in read?_seed_start/length)
    PeCand* R;        // lists of re-seeded mates (cur == 2), at roff[read]
    u64*  roff;       // per read
};

// mate 2: reverse complement of the FASTQ read (rc_table, Process_Reads.cpp:1603-1613: identity for non-ACGT)
__global__ void __launch_bounds__(256)
k_pe_prepare(const char* __restrict__ s1, const char* __restrict__ s2raw, ReadGeom gm, int stride, long n, char* __restrict__ seq_all,
             u64* __restrict__ prow, int pwords, int W, u32* __restrict__ dirty32, int sparse_ascii)
{
    // sparse_ascii (the packed rows are what every later kernel reads, --sensitive's re-seeding included): the ASCII text of a 16-byte piece is stored only
    // when the piece holds a character outside ACGT -- the only places the ASCII rows are asked then (is it 'N'?) sit
    // under a set bit of the mask plane, so the other 99.9 % of the 2 x n x stride bytes are never written
    // one 16-byte piece per thread (rows are 16-byte aligned, stride % 16 == 0)
    const long i16 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total16 = n * (stride / 16);
    if (i16 >= total16) return;
    const long total = n * stride;
    const long i = i16 * 16;
    const long r = i / stride;
    const int j0 = (int)(i - r * stride);
    const int L = gm.rl(n + r);                    // mate 2 of pair r
    const uint4 v1 = reinterpret_cast<const uint4*>(s1)[i16];
    if (!sparse_ascii) reinterpret_cast<uint4*>(seq_all)[i16] = v1;                           // the qualities stay where they are (qual_row)
    // the packed copy of both rows (what k_pack_rows would write), from the pieces this thread holds anyway
    auto pack_out = [&](const uint4& pv, long row_id, int Lr, uint4* ascii) {
        const int piece = j0 / 16;
        if (!prow || piece * 16 >= ((Lr + 63) & ~63)) return;
        u32 bases, mask;
        pack_piece(pv, Lr - piece * 16, bases, mask);
        u64* row = prow + (size_t)row_id * pwords;
        if (piece * 16 < ((Lr + 31) & ~31)) reinterpret_cast<u32*>(row)[piece] = bases;
        reinterpret_cast<u16*>(row + W)[piece] = (u16)mask;
        if (mask) {
            atomicOr(&dirty32[row_id >> 2], 1u << (8 * (int)(row_id & 3)));
            if (sparse_ascii) *ascii = pv;
        }
    };
    pack_out(v1, r, gm.rl(r), reinterpret_cast<uint4*>(seq_all) + i16);
    // out[j] = complement(in[L-1-j]) for j < L, 0 beyond: one reversed 16-byte piece per thread.  complement = c ^ 0x15 for
    // A/T, c ^ 0x04 for C/G, identity otherwise (rc_table), eight characters per step.
    auto comp8 = [](u64 w) -> u64 {
        const u64 K7F = 0x7f7f7f7f7f7f7f7full;
        auto zb = [&](u64 x) -> u64 { return ~(((x & K7F) + K7F) | x | K7F); };                 // 0x80 where a byte is 0
        const u64 at = zb(w ^ 0x4141414141414141ull) | zb(w ^ 0x5454545454545454ull);
        const u64 cg = zb(w ^ 0x4343434343434343ull) | zb(w ^ 0x4747474747474747ull);
        return w ^ ((at >> 7) * 0x15) ^ ((cg >> 7) * 0x04);
    };
    const char* row = s2raw + r * stride;
    const int src = L - 16 - j0;                       // in[src .. src+15] reversed = out[j0 .. j0+15]
    uint4 v;
    if (src >= 0) {
        const u64 lo = *reinterpret_cast<const u64*>(row + src), hi = *reinterpret_cast<const u64*>(row + src + 8);
        const u64 a = comp8(__builtin_bswap64(hi)), b2 = comp8(__builtin_bswap64(lo));
        v.x = (u32)a; v.y = (u32)(a >> 32); v.z = (u32)b2; v.w = (u32)(b2 >> 32);
    } else {
        unsigned char o[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int j = j0 + t;
            char c = 0;
            if (j < L) { const char a = row[L - 1 - j]; c = a == 'A' ? 'T' : a == 'T' ? 'A' : a == 'C' ? 'G' : a == 'G' ? 'C' : a; }
            o[t] = (unsigned char)c;
        }
        v.x = o[0] | (o[1] << 8) | (o[2] << 16) | ((u32)o[3] << 24);
        v.y = o[4] | (o[5] << 8) | (o[6] << 16) | ((u32)o[7] << 24);
        v.z = o[8] | (o[9] << 8) | (o[10] << 16) | ((u32)o[11] << 24);
        v.w = o[12] | (o[13] << 8) | (o[14] << 16) | ((u32)o[15] << 24);
    }
    if (!sparse_ascii) reinterpret_cast<uint4*>(seq_all + total)[i16] = v;
    pack_out(v, n + r, L, reinterpret_cast<uint4*>(seq_all + total) + i16);
}

// k_pe_prepare for the packed fast path (sparse ASCII): mate 2 is packed FORWARD first (one aligned 16-byte piece per thread,
// the same SWAR as mate 1), parked in LDS, and the reverse complement is then taken on the packed words -- a 32-bit funnel
// window of the forward row, complemented under its valid-base mask and reversed by 2-bit groups.  The byte-wise complement
// + byte swap of the ASCII form cost ~2.5x the instructions of everything else in the kernel, which was VALU-bound.
// A block takes 256 / (stride / 16) whole pairs; only a piece that holds a character outside ACGT rebuilds its ASCII text.
__global__ void __launch_bounds__(256)
k_pe_prepare_p(const char* __restrict__ s1, const char* __restrict__ s2raw, ReadGeom gm, int stride, long n, char* __restrict__ seq_all,
               u64* __restrict__ prow, int pwords, int W, u32* __restrict__ dirty32)
{
    extern __shared__ u32 lds_pp[];
    const int ppr = stride / 16, rpb = 256 / ppr;
    u32* lb = lds_pp;                                   // [rpb][ppr + 1] forward base words (+ one zero word)
    u32* lm = lds_pp + rpb * (ppr + 1);                 // [rpb][ppr + 1] forward mask pieces (16 bits each, + one zero)
    const int tid = threadIdx.x, rl = tid / ppr, piece = tid - rl * ppr;
    const long r = (long)blockIdx.x * rpb + rl;
    const bool on = rl < rpb && r < n;
    const long total = n * stride;
    int L2 = 0;
    if (on) {
        const size_t i16 = (size_t)r * ppr + piece;
        const int L1 = gm.rl(r);
        L2 = gm.rl(n + r);
        const uint4 v1 = reinterpret_cast<const uint4*>(s1)[i16];
        const uint4 v2 = reinterpret_cast<const uint4*>(s2raw)[i16];
        u32 b1, m1, b2, m2;
        pack_piece(v1, L1 - piece * 16, b1, m1);
        pack_piece(v2, L2 - piece * 16, b2, m2);
        lb[rl * (ppr + 1) + piece] = b2; lm[rl * (ppr + 1) + piece] = m2;
        if (piece == 0) { lb[rl * (ppr + 1) + ppr] = 0; lm[rl * (ppr + 1) + ppr] = 0; }
        if (piece * 16 < ((L1 + 63) & ~63)) {
            u64* row = prow + (size_t)r * pwords;
            if (piece * 16 < ((L1 + 31) & ~31)) reinterpret_cast<u32*>(row)[piece] = b1;
            reinterpret_cast<u16*>(row + W)[piece] = (u16)m1;
            if (m1) { atomicOr(&dirty32[r >> 2], 1u << (8 * (int)(r & 3))); reinterpret_cast<uint4*>(seq_all)[i16] = v1; }
        }
    }
    __syncthreads();
    if (!on || piece * 16 >= ((L2 + 63) & ~63)) return;
    const int j0 = piece * 16, lo = L2 - 16 - j0;       // forward positions lo .. lo+15, reversed, are rc positions j0 .. j0+15
    const u32* fb = lb + rl * (ppr + 1); const u32* fm = lm + rl * (ppr + 1);
    u32 win, bad, inr;
    if (lo >= 0) {
        const int idx = lo >> 4, sh = lo & 15;
        win = sh ? (fb[idx] >> (2 * sh)) | (fb[idx + 1] << (32 - 2 * sh)) : fb[idx];
        bad = ((fm[idx] | (fm[idx + 1] << 16)) >> sh) & 0xffffu;
        inr = 0xffffu;
    } else if (lo > -16) {
        win = fb[0] << (2 * -lo);
        bad = (fm[0] << -lo) & 0xffffu;
        inr = (0xffffu << -lo) & 0xffffu;
    } else { win = 0; bad = 0; inr = 0; }
    u32 x = inr & ~bad;                                 // real bases of the window -> both bits of their pair
    x = (x | (x << 8)) & 0x00ff00ffu; x = (x | (x << 4)) & 0x0f0f0f0fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    win ^= x | (x << 1);                                // complement: code -> 3 - code
    u32 rv = __brev(win);
    rv = ((rv >> 1) & 0x55555555u) | ((rv & 0x55555555u) << 1);
    const u32 om = __brev(bad & inr) >> 16;
    const long row_id = n + r;
    u64* row = prow + (size_t)row_id * pwords;
    if (j0 < ((L2 + 31) & ~31)) reinterpret_cast<u32*>(row)[piece] = rv;
    reinterpret_cast<u16*>(row + W)[piece] = (u16)om;
    if (om) {
        atomicOr(&dirty32[row_id >> 2], 1u << (8 * (int)(row_id & 3)));
        const char* src = s2raw + r * stride;
        unsigned char o[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int j = j0 + t;
            char c = 0;
            if (j < L2) { const char a = src[L2 - 1 - j]; c = a == 'A' ? 'T' : a == 'T' ? 'A' : a == 'C' ? 'G' : a == 'G' ? 'C' : a; }
            o[t] = (unsigned char)c;
        }
        uint4 v;
        v.x = o[0] | (o[1] << 8) | (o[2] << 16) | ((u32)o[3] << 24);
        v.y = o[4] | (o[5] << 8) | (o[6] << 16) | ((u32)o[7] << 24);
        v.z = o[8] | (o[9] << 8) | (o[10] << 16) | ((u32)o[11] << 24);
        v.w = o[12] | (o[13] << 8) | (o[14] << 16) | ((u32)o[15] << 24);
        reinterpret_cast<uint4*>(seq_all + total)[(size_t)r * ppr + piece] = v;
    }
}

// ---- FASTQ text -> read rows (bmbs_map_*_fastq) --------------------------------------------------------------------------------
// What inputReads_single_directly / inputReads_paired_directly (Process_Reads.cpp:810-890, 155-317) do per record on the host --
// cut the sequence and quality lines out of the text, upper-case the bases, pad short quality lines with ' ', reverse-complement
// mate 2 (and every read of a --pbat library, with mirrored qualities) -- done here for a whole batch from the FASTQ text as it was
// read from the file: the host only finds the line starts.  One thread per 16-byte piece of an output row.
__global__ void __launch_bounds__(256)
k_fastq_rows(const char* __restrict__ text, const u32* __restrict__ seq_off, const u32* __restrict__ qual_off,
             const u16* __restrict__ seq_len, const u16* __restrict__ qual_len, long n, int stride, int rc_seq, int rev_qual,
             char* __restrict__ seq_out, char* __restrict__ qual_out, u16* __restrict__ len_out)
{
    const long i16 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = stride / 16;
    if (i16 >= n * per_row) return;
    const long r = i16 / per_row;
    const int j0 = (int)(i16 - r * per_row) * 16;
    const int L = seq_len[r];
    if (len_out && j0 == 0) len_out[r] = (u16)L;
    const char* src = text + seq_off[r];
    unsigned char o[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int j = j0 + t;
        unsigned char c = 0;
        if (j < L) {
            c = (unsigned char)src[rc_seq ? L - 1 - j : j];
            if (c >= 'a' && c <= 'z') c -= 32;
            if (rc_seq) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;      // rc_table, Process_Reads.cpp:1603
        }
        o[t] = c;
    }
    uint4 v;
    v.x = o[0] | (o[1] << 8) | (o[2] << 16) | ((u32)o[3] << 24);
    v.y = o[4] | (o[5] << 8) | (o[6] << 16) | ((u32)o[7] << 24);
    v.z = o[8] | (o[9] << 8) | (o[10] << 16) | ((u32)o[11] << 24);
    v.w = o[12] | (o[13] << 8) | (o[14] << 16) | ((u32)o[15] << 24);
    reinterpret_cast<uint4*>(seq_out)[i16] = v;
    if (!qual_out) return;
    const int ql = qual_len[r];
    const char* qs = text + qual_off[r];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int j = j0 + t;
        unsigned char c = 0;
        if (j < L) { const int jj = rev_qual ? L - 1 - j : j; c = jj < ql ? (unsigned char)qs[jj] : (unsigned char)' '; }
        o[t] = c;
    }
    v.x = o[0] | (o[1] << 8) | (o[2] << 16) | ((u32)o[3] << 24);
    v.y = o[4] | (o[5] << 8) | (o[6] << 16) | ((u32)o[7] << 24);
    v.z = o[8] | (o[9] << 8) | (o[10] << 16) | ((u32)o[11] << 24);
    v.w = o[12] | (o[13] << 8) | (o[14] << 16) | ((u32)o[15] << 24);
    reinterpret_cast<uint4*>(qual_out)[i16] = v;
}

// get_candidates' list construction (Schema.cpp:18510-18545): site-sorted votes (NOT re-sorted by vote)
__global__ void __launch_bounds__(64)
k_vote_pe(long n2, ReadGeom gm, ReadState st, PeState ps, u64* __restrict__ cand, PeCand* __restrict__ A, u32* __restrict__ slot_read)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n2) return;
    const int L = gm.rl(r), k = gm.rk(L);
    const int v = st.verdict[r];
    const u64 off = st.cand_off[r];
    for (u64 g = off; g < st.cand_off[r + 1]; g++) slot_read[g] = (u32)r;
    ps.cur[r] = 0; ps.vround[r] = 0;
    if (v == 1 || v == 2) {
        A[off].site = st.exit_site[r]; A[off].err = v == 1 ? 0u : 1u; A[off].end = L - 1;
        ps.occ[r] = 1; ps.len[r] = 1;
    } else if (v == 4) {
        const long nc = (long)st.n_cand[r];
        sort_u64_asc(cand + off, nc);
        for (long i = 0; i < nc; i++) { A[off + i].site = cand[off + i]; A[off + i].err = 0; A[off + i].end = L - 1; }
        ps.occ[r] = (int)nc; ps.len[r] = (u32)nc;
    } else if (v == 3) {
        const long nc = (long)st.n_cand[r];
        u64* c = cand + off;
        sort_u64_asc(c, nc);
        PeCand* o = A + off;
        long nv = 0;
        u64 pre = c[0];
        for (long i = 1; i < nc; i++)
            if (c[i] != pre) { o[nv].site = pre < (u64)k ? 0 : pre - (u64)k; o[nv].err = 0; o[nv].end = 0; nv++; pre = c[i]; }
        o[nv].site = pre >= (u64)k ? pre - (u64)k : 0; o[nv].err = 0; o[nv].end = 0; nv++;
        ps.occ[r] = -1; ps.len[r] = (u32)nv;
    } else { ps.occ[r] = 0; ps.len[r] = 0; }
}

// k_locate + k_vote_pe for lists of up to VOTE_REG candidates: located into registers, sorted by the same network as
// k_vote_fused; general reads emit one entry per distinct site (no vote order), exact-ambiguous reads every hit
__global__ void __launch_bounds__(64)
k_vote_pe_fused(DevIndex ix, long n2, ReadGeom gm, ReadState st, PeState ps, u64* __restrict__ cand, PeCand* __restrict__ A,
                u32* __restrict__ slot_read, u32* __restrict__ long_flag, u32* __restrict__ mid_flag)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n2) return;
    long_flag[r] = 0;
    const int L = gm.rl(r), k = gm.rk(L);
    const int v = st.verdict[r];
    const u64 off = st.cand_off[r];
    (void)slot_read;             // nothing downstream of the paired-end vote stage reads the slot -> read map (49 M scattered stores per launch)
    ps.cur[r] = 0; ps.vround[r] = 0;
    if (v == 1 || v == 2) {
        A[off].site = st.exit_site[r]; A[off].err = v == 1 ? 0u : 1u; A[off].end = L - 1;
        ps.occ[r] = 1; ps.len[r] = 1;
        return;
    }
    if (v != 3 && v != 4) { ps.occ[r] = 0; ps.len[r] = 0; return; }
    const long nc = (long)st.n_cand[r];
    const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
    const int ns = st.n_seeds[r];
    PeCand* o = A + off;
    if (nc <= VOTE_REG) {
        u64 c[VOTE_REG];
        int sidx = 0; u32 h = 0;
        u64 sp = 0, adj = 0; u32 hits = 0;
        if (ns > 0) { sp = my[0].sp; adj = (u64)my[0].len + (u64)my[0].off; hits = my[0].hits; }
#pragma unroll
        for (int j = 0; j < VOTE_REG; j++) {
            c[j] = ~0ull;
            if (j < nc) {
                while (h == hits && sidx + 1 < ns) { sidx++; h = 0; sp = my[sidx].sp; adj = (u64)my[sidx].len + (u64)my[sidx].off; hits = my[sidx].hits; }
                c[j] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + h)) - adj;
                h++;
            }
        }
#define CE(a, b) { const u64 x_ = c[a], y_ = c[b]; c[a] = x_ < y_ ? x_ : y_; c[b] = x_ < y_ ? y_ : x_; }
        CE(0,1) CE(2,3) CE(4,5) CE(6,7) CE(8,9) CE(10,11) CE(12,13) CE(14,15)
        CE(0,2) CE(1,3) CE(4,6) CE(5,7) CE(8,10) CE(9,11) CE(12,14) CE(13,15)
        CE(1,2) CE(5,6) CE(9,10) CE(13,14)
        CE(0,4) CE(1,5) CE(2,6) CE(3,7) CE(8,12) CE(9,13) CE(10,14) CE(11,15)
        CE(2,4) CE(3,5) CE(10,12) CE(11,13)
        CE(1,2) CE(3,4) CE(5,6) CE(9,10) CE(11,12) CE(13,14)
        CE(0,8) CE(1,9) CE(2,10) CE(3,11) CE(4,12) CE(5,13) CE(6,14) CE(7,15)
        CE(4,8) CE(5,9) CE(6,10) CE(7,11)
        CE(2,4) CE(3,5) CE(6,8) CE(7,9) CE(10,12) CE(11,13)
        CE(1,2) CE(3,4) CE(5,6) CE(7,8) CE(9,10) CE(11,12) CE(13,14)
#undef CE
        int nv = 0;
#pragma unroll
        for (int i = 0; i < VOTE_REG; i++) {
            if (i < nc) {
                if (v == 4) { o[i].site = c[i]; o[i].err = 0; o[i].end = L - 1; }
                else if (i + 1 >= nc || (i + 1 < VOTE_REG && c[i + 1] != c[i])) {
                    o[nv].site = c[i] < (u64)k ? 0 : c[i] - (u64)k; o[nv].err = 0; o[nv].end = 0; nv++;
                }
            }
        }
        if (v == 4) { ps.occ[r] = (int)nc; ps.len[r] = (u32)nc; }
        else { ps.occ[r] = -1; ps.len[r] = (u32)nv; }
        return;
    }
    if (mid_flag && nc <= VOTE_MID) { mid_flag[r] = 1; return; }      // 17..32 candidates, the rule for reads of 180 bases and more: k_vote_pe_mid
    long_flag[r] = 1;                                         // repeats: k_vote_pe_long sorts the list out of LDS (beyond its capacity: in tiles)
}

// lists of 17..32 candidates (reads of 180 bases and more place up to 25 seeds): one lane per read over the compacted list,
// located into registers and sorted by a bitonic network -- the paired-end counterpart of k_vote_mid (no vote order here)
__global__ void __launch_bounds__(64)
k_vote_pe_mid(DevIndex ix, ReadGeom gm, ReadState st, PeState ps, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
              PeCand* __restrict__ A)
{
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= (long)*count_ptr) return;
    const long r = list[it];
    const int L = gm.rl(r), k = gm.rk(L);
    const int v = st.verdict[r];
    const long nc = (long)st.n_cand[r];
    const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
    const int ns = st.n_seeds[r];
    PeCand* o = A + st.cand_off[r];
    u64 c[VOTE_MID];
    {
        int sidx = 0; u32 h = 0;
        u64 sp = 0, adj = 0; u32 hits = 0;
        if (ns > 0) { sp = my[0].sp; adj = (u64)my[0].len + (u64)my[0].off; hits = my[0].hits; }
#pragma unroll
        for (int j = 0; j < VOTE_MID; j++) {
            c[j] = ~0ull;
            if (j < nc) {
                while (h == hits && sidx + 1 < ns) { sidx++; h = 0; sp = my[sidx].sp; adj = (u64)my[sidx].len + (u64)my[sidx].off; hits = my[sidx].hits; }
                c[j] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + h)) - adj;
                h++;
            }
        }
    }
#pragma unroll
    for (int size = 2; size <= VOTE_MID; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int t = 0; t < VOTE_MID / 2; t++) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool asc = (i & size) == 0;
                const u64 x_ = c[i], y_ = c[j];
                const bool sw = asc ? x_ > y_ : x_ < y_;
                c[i] = sw ? y_ : x_; c[j] = sw ? x_ : y_;
            }
        }
    }
    if (v == 4) {
#pragma unroll
        for (int i = 0; i < VOTE_MID; i++) if (i < nc) { PeCand e; e.site = c[i]; e.err = 0; e.end = L - 1; o[i] = e; }
        ps.occ[r] = (int)nc; ps.len[r] = (u32)nc;
    } else {
        int nv = 0;
#pragma unroll
        for (int i = 0; i < VOTE_MID; i++) {
            if (i < nc && (i + 1 >= nc || (i + 1 < VOTE_MID && c[i + 1] != c[i]))) {
                PeCand e; e.site = c[i] < (u64)k ? 0 : c[i] - (u64)k; e.err = 0; e.end = 0;
                o[nv++] = e;
            }
        }
        ps.occ[r] = -1; ps.len[r] = (u32)nv;
    }
}

// the paired-end counterpart of k_vote_long: no vote order here, so everything is parallel (general reads: one entry per
// distinct site; exact-ambiguous reads: every hit)
template <int CAP, int BLOCK, int LO>
__global__ void __launch_bounds__(BLOCK)
k_vote_pe_long(DevIndex ix, ReadGeom gm, ReadState st, PeState ps, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
               PeCand* __restrict__ A, u32* __restrict__ big_list, unsigned long long* __restrict__ big_count, u64* __restrict__ cand)
{
    __shared__ u64 keys[CAP];
    __shared__ u16 endpos[CAP];
    __shared__ u32 sh_pref[BMBS_MAX_SEEDS + 1];
    __shared__ int sh_w[2 * (BLOCK / 64) + 1];
    const long total_items = (long)*count_ptr;
    for (long item = blockIdx.x; item < total_items; item += gridDim.x) {
        const long r = list[item];
        const long nc = (long)st.n_cand[r];
        // the wave form sees every listed read and passes the ones beyond its capacity on (a list of their own: the block form used
        // to walk the whole list -- millions of reads on a repeat-rich genome, two dependent loads each -- to find its few)
        if (big_list && nc > CAP) { if (threadIdx.x == 0) big_list[atomicAdd(big_count, 1ull)] = (u32)r; continue; }
        if (nc <= LO || (CAP != VL_CAP && nc > CAP)) continue;       // another instance's size class (the largest also takes what is beyond it)
        const int L = gm.rl(r), k = gm.rk(L);
        const int v = st.verdict[r];
        PeCand* o = A + st.cand_off[r];
        if (nc > CAP) {
            // beyond the LDS capacity: sorted in tiles (vl_sort_huge; the output segment parks the tiles), then the same entries in order
            u64* c = cand + st.cand_off[r];
            vl_sort_huge<CAP, BLOCK>(ix, st.seeds + (size_t)r * BMBS_MAX_SEEDS, st.n_seeds[r], nc, keys, sh_pref, reinterpret_cast<u64*>(o), c);
            if (v == 4) {
                for (long i = threadIdx.x; i < nc; i += BLOCK) { PeCand e; e.site = c[i]; e.err = 0; e.end = L - 1; o[i] = e; }
                if (threadIdx.x == 0) { ps.occ[r] = (int)nc; ps.len[r] = (u32)nc; }
            } else {
                int running = 0;
                for (long base = 0; base < nc; base += BLOCK) {
                    const long i = base + (long)threadIdx.x;
                    bool keep = false;
                    u64 key = 0;
                    if (i < nc) { key = c[i]; keep = i == nc - 1 || c[i + 1] != key; }
                    int tot;
                    const int pre = vl_prefix(keep, sh_w, tot);
                    if (keep) { PeCand e; e.site = key < (u64)k ? 0 : key - (u64)k; e.err = 0; e.end = 0; o[running + pre] = e; }
                    running += tot;
                }
                if (threadIdx.x == 0) { ps.occ[r] = -1; ps.len[r] = (u32)running; }
            }
            __syncthreads();
            continue;
        }
        vl_locate_sort<(CAP + BLOCK - 1) / BLOCK>(ix, st.seeds + (size_t)r * BMBS_MAX_SEEDS, st.n_seeds[r], (int)nc, keys, sh_pref);
        if (v == 4) {
            for (long i = threadIdx.x; i < nc; i += BLOCK) { PeCand e; e.site = keys[i]; e.err = 0; e.end = L - 1; o[i] = e; }
            if (threadIdx.x == 0) { ps.occ[r] = (int)nc; ps.len[r] = (u32)nc; }
        } else {
            const int nv = vl_run_ends(keys, (int)nc, endpos, sh_w);
            for (int e2 = threadIdx.x; e2 < nv; e2 += BLOCK) {
                const u64 site = keys[endpos[e2]];
                PeCand e; e.site = site < (u64)k ? 0 : site - (u64)k; e.err = 0; e.end = 0;
                o[e2] = e;
            }
            if (threadIdx.x == 0) { ps.occ[r] = -1; ps.len[r] = (u32)nv; }
        }
        __syncthreads();
    }
}

DEVI PeCand* pe_list(const PeState& ps, const ReadState& st, PeCand* A, PeCand* B, long r)
{
    const int cur = ps.cur[r];
    return cur == 2 ? ps.R + ps.roff[r] : (cur ? B : A) + st.cand_off[r];
}

// filter_pairs (Schema.cpp:16052-16180) + the driver's choice of what to verify (19050-19290)
// what follows the two filtered lists (Schema.cpp:19050-19290): who is verified in which round
DEVI void pe_filter_decide(const PeState& ps, long p, long r1, long r2, int occ1, int occ2, long la, long lb)
{
    ps.cur[r1] = 1; ps.cur[r2] = 1;
    ps.len[r1] = (u32)la; ps.len[r2] = (u32)lb;
    if (la == 0 || lb == 0) { ps.dead[p] = 1; return; }
    if (occ1 == -1 && occ2 == -1) {
        ps.both[p] = 1;
        if (la <= lb) { ps.vround[r1] = 1; ps.vround[r2] = 2; } else { ps.vround[r2] = 1; ps.vround[r1] = 2; }
    } else if (occ1 != -1) {
        if (la < occ1) ps.occ[r1] = (int)la;
        ps.vround[r2] = 1;
    } else {
        if (lb < occ2) ps.occ[r2] = (int)lb;
        ps.vround[r1] = 1;
    }
}
// the reference's merge loop itself, one lane
DEVI void pe_filter_serial(const PeCand* a, long na, const PeCand* b, long nb, long long maxd, long long mind, PeCand* ra, PeCand* rb, long& la_out, long& lb_out)
{
    long la = 0, lb = 0, first = 0;
    for (long i = 0; i < na; i++) {
        for (long j = first; j < nb; j++) {
            bool hit = false;
            if (a[i].site > b[j].site) {
                const long long d = (long long)(a[i].site - b[j].site);
                if (d > maxd) first = j + 1;
                else if (d >= mind) hit = true;
            } else {
                const long long d = (long long)(b[j].site - a[i].site);
                if (d > maxd) break;
                if (d >= mind) hit = true;
            }
            if (hit) {
                if (la == 0 || a[i].site > ra[la - 1].site) ra[la++] = a[i];
                if (lb == 0 || b[j].site > rb[lb - 1].site) rb[lb++] = b[j];
            }
        }
    }
    la_out = la; lb_out = lb;
}
#define PEF_LONG 24       // pairs whose two lists hold more candidates than this go to k_pe_filter_pairs_long (one wave per pair)
__global__ void __launch_bounds__(64)
k_pe_filter_pairs(long n, ReadGeom gm, PeIns pi, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B, u32* __restrict__ long_flag)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    long long maxd, mind; int large_k;
    pe_bounds(gm, pi, p, n, maxd, mind, large_k);
    const long r1 = p, r2 = p + n;
    int occ1 = ps.occ[r1], occ2 = ps.occ[r2];
    ps.dead[p] = 0; ps.both[p] = 0; ps.npair[p] = 0; ps.sbd[p] = 0;
    if (occ1 > 0 && occ2 > 0) return;
    if (occ1 == 0 || occ2 == 0) { ps.dead[p] = 1; return; }
    const long na = ps.len[r1], nb = ps.len[r2];
    // in a repeat-rich genome one pair in a few has lists of dozens to thousands of candidates: a lane that walks them alone holds its
    // wave for as long (k_pe_filter_pairs: 0.46 ms per 10 M pairs on the uniform genome, 12.8 ms on the GRCh38-like one)
    if (long_flag && na + nb > PEF_LONG) { long_flag[p] = 1; return; }
    long la, lb;
    pe_filter_serial(A + st.cand_off[r1], na, A + st.cand_off[r2], nb, maxd, mind, B + st.cand_off[r1], B + st.cand_off[r2], la, lb);
    pe_filter_decide(ps, p, r1, r2, occ1, occ2, la, lb);
}

// One wave per pair with long lists.  With mind <= 0 (the default --min 0 makes it negative) a pair of sites hits iff they lie
// within maxd of each other, so: an entry of one list survives iff the other list holds a site within maxd of it (a binary search
// over the sorted other list), minus entries whose site repeats the one before (the reference pushes a[i] / b[j] only when its
// site is larger than the last one pushed).  Every lane takes entries of its own; ballots compact the survivors in order.
// mind > 0 (a minimum insert larger than the read + 2k): the hit rule is no longer an interval and the order of discovery
// matters -- lane 0 runs the reference's loop.
DEVI long pe_lower_bound(const PeCand* v, long n, u64 key)          // first index with site >= key
{
    long lo = 0, hi = n;
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (v[mid].site < key) lo = mid + 1; else hi = mid; }
    return lo;
}
DEVI long pe_filter_side(const PeCand* x, long nx, const PeCand* y, long ny, u64 maxd, PeCand* out)
{
    const int lane = threadIdx.x & 63;
    long w = 0;
    for (long base = 0; base < nx; base += 64) {
        const long i = base + lane;
        bool keep = false;
        PeCand e; e.site = 0; e.err = 0; e.end = 0;
        if (i < nx) {
            e = x[i];
            const long j = pe_lower_bound(y, ny, e.site > maxd ? e.site - maxd : 0);        // the first site of y not more than maxd below e
            keep = j < ny && (y[j].site <= e.site || y[j].site - e.site <= maxd);            // ... is it also not more than maxd above?
            if (keep && i > 0 && x[i - 1].site == e.site) keep = false;                      // a repeated site is pushed once
        }
        const unsigned long long m = __ballot(keep);
        if (keep) out[w + __popcll(m & ((1ull << lane) - 1))] = e;
        w += __popcll(m);
    }
    return w;
}
__global__ void __launch_bounds__(64)
k_pe_filter_pairs_long(long n, ReadGeom gm, PeIns pi, ReadState st, PeState ps, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
                       PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long total = (long)*count_ptr;
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const long p = list[item];
        long long maxd, mind; int large_k;
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        const long r1 = p, r2 = p + n;
        const int occ1 = ps.occ[r1], occ2 = ps.occ[r2];
        const PeCand* a = A + st.cand_off[r1];
        const PeCand* b = A + st.cand_off[r2];
        PeCand* ra = B + st.cand_off[r1];
        PeCand* rb = B + st.cand_off[r2];
        const long na = ps.len[r1], nb = ps.len[r2];
        long la = 0, lb = 0;
        // sites that wrapped around below zero (a seed at the very start of the text) sort last as huge unsigned values and take the
        // reference's mixed unsigned / signed comparisons: those pairs keep its loop
        const bool wrapped = (na && (a[na - 1].site >> 63)) || (nb && (b[nb - 1].site >> 63));
        if (mind <= 0 && maxd >= 0 && !wrapped) {
            la = pe_filter_side(a, na, b, nb, (u64)maxd, ra);
            lb = pe_filter_side(b, nb, a, na, (u64)maxd, rb);
        } else {
            if ((threadIdx.x & 63) == 0) pe_filter_serial(a, na, b, nb, maxd, mind, ra, rb, la, lb);
            la = __shfl((int)la, 0, 64); lb = __shfl((int)lb, 0, 64);
        }
        if ((threadIdx.x & 63) == 0) pe_filter_decide(ps, p, r1, r2, occ1, occ2, la, lb);
    }
}

// verify_candidate_locations' Myers pass for the mates scheduled in `round`: a dense work list
// (read, list index) is built by count -> scan -> scatter so that the filter runs on full waves
__global__ void __launch_bounds__(256)
k_pe_count(long n, long n2, int round, PeState ps, u32* __restrict__ cnt)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n2) return;
    const long p = r < n ? r : r - n;
    cnt[r] = (ps.vround[r] == round && !ps.dead[p]) ? ps.len[r] : 0u;
}
__global__ void __launch_bounds__(256)
k_pe_worklist(long n2, const u32* __restrict__ cnt, const u64* __restrict__ off, u32* __restrict__ work_r, u32* __restrict__ work_i)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n2) return;
    const u32 m = cnt[r];
    const u64 o = off[r];
    for (u32 i = 0; i < m; i++) { work_r[o + i] = (u32)r; work_i[o + i] = i; }
}
__global__ void __launch_bounds__(256)
k_filter_pe(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, ReadState st, PeState ps,
            PeCand* __restrict__ A, PeCand* __restrict__ B, const u64* __restrict__ n_work, const u32* __restrict__ work_r,
            const u32* __restrict__ work_i, unsigned long long* __restrict__ counters)
{
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= *n_work) return;
    const long r = (long)work_r[g];
    PeCand* e = pe_list(ps, st, A, B, r) + work_i[g];
    u32 er; int es;
    const int L = gm.rl(r), k = gm.rk(L);
    bpm_read(ix, seq, stride, pr, r, L, k, e->site, er, es);
    e->err = er; e->end = es;
    if (counters) atomicAdd(&SHARD(counters)[3], 1ull);
}

// the PE compaction (Schema.cpp:7480-7690): keep err <= k whose site+end differs from the previous candidate's
// A lane walks its read's list; a list of more than 64 entries (a read inside a repeat family: hundreds to thousands) is walked by
// the whole wave afterwards, 64 entries a step -- one such lane used to hold its wave for the length of its list.
__global__ void __launch_bounds__(64)
k_pe_compact(long n, long n2, ReadGeom gm, int round, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool act = r < n2 && ps.vround[r] == round;
    if (act) { const long p = r < n ? r : r - n; if (ps.dead[p]) act = false; }
    const long m = act ? (long)ps.len[r] : 0;
    const bool coop = m > 64;
    if (act && !coop) {
        const int k = gm.rk(gm.rl(r));
        PeCand* l = pe_list(ps, st, A, B, r);
        u64 pre = ~0ull;
        int occ = 0;
        for (long i = 0; i < m; i++) {
            const PeCand c = l[i];
            const u64 t = c.site + (u64)(long long)c.end;
            if (c.err <= (u32)k && pre != t) { l[occ] = c; occ++; }
            pre = t;
        }
        ps.occ[r] = occ;
    }
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long rr = (long)__shfl((long long)r, src, 64);
        const long mm = (long)ps.len[rr];
        const int k = gm.rk(gm.rl(rr));
        PeCand* l = pe_list(ps, st, A, B, rr);
        u64 carry = ~0ull;                              // site + end of the entry before the step's first
        int occ = 0;
        for (long base = 0; base < mm; base += 64) {
            const long i = base + lane;
            PeCand c; c.site = 0; c.err = 0; c.end = 0;
            u64 t = 0;
            if (i < mm) { c = l[i]; t = c.site + (u64)(long long)c.end; }
            u64 pre = (u64)__shfl_up((long long)t, 1, 64);
            if (lane == 0) pre = carry;
            const bool keep = i < mm && c.err <= (u32)k && pre != t;
            const unsigned long long kb = __ballot(keep);       // every entry of the step is in registers before the first is stored
            if (keep) l[occ + __popcll(kb & ((1ull << lane) - 1))] = c;
            occ += __popcll(kb);
            carry = (u64)__shfl((long long)t, 63, 64);
        }
        if (lane == src) ps.occ[rr] = occ;
    }
}

// after round 1 of a both-unverified pair: filter_pairs_single_side (Schema.cpp:16186-16270)
// The reference's merge loop, one lane per pair -- and, for a pair whose two lists hold more than 64 entries, by the whole wave:
// with a lower bound of the distance <= 0 (the default insert range) the loop keeps b[j] exactly when some a[i] lies within maxd of
// it (it drops b[j] only when the current a[i] is more than maxd above it, and then so is every later one; it leaves the scan of
// a[i] only at a b[j] more than maxd above a[i], and then so is every later one), so every b[j] is decided by one binary search in a.
__global__ void __launch_bounds__(64)
k_pe_prune(long n, ReadGeom gm, PeIns pi, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool act = p < n && !ps.dead[p] && ps.both[p];
    long long maxd = 0, mind = 0; int large_k;
    long rs = 0, ro = 0;
    int occ_s = 0;
    if (act) {
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        rs = ps.vround[p] == 1 ? p : p + n;       // verified side
        ro = ps.vround[p] == 1 ? p + n : p;       // side still to verify
        occ_s = ps.occ[rs];
        if (occ_s == 0) { ps.dead[p] = 1; act = false; }
    }
    const long nb = act ? (long)ps.len[ro] : 0;
    auto serial = [&](const PeCand* a, long na, PeCand* b, long nbb, long long mxd, long long mnd) -> long {
        long len2 = 0, first = 0;
        for (long i = 0; i < na; i++) {
            for (long j = first; j < nbb; j++) {
                if (a[i].site > b[j].site) {
                    const long long d = (long long)(a[i].site - b[j].site);
                    if (d > mxd) first = j + 1;
                    else if (d >= mnd) { b[len2] = b[j]; len2++; first = j + 1; }
                } else {
                    const long long d = (long long)(b[j].site - a[i].site);
                    if (d > mxd) break;
                    if (d >= mnd) { b[len2] = b[j]; len2++; first = j + 1; }
                }
            }
        }
        return len2;
    };
    const bool coop = act && nb + occ_s > 64 && mind <= 0 && maxd >= 0;
    if (act && !coop) ps.len[ro] = (u32)serial(pe_list(ps, st, A, B, rs), occ_s, pe_list(ps, st, A, B, ro), nb, maxd, mind);
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long pp = (long)__shfl((long long)p, src, 64);
        long long mxd, mnd; int lk;
        pe_bounds(gm, pi, pp, n, mxd, mnd, lk);
        const long rs2 = ps.vround[pp] == 1 ? pp : pp + n, ro2 = ps.vround[pp] == 1 ? pp + n : pp;
        const long na = ps.occ[rs2], nbb = (long)ps.len[ro2];
        const PeCand* a = pe_list(ps, st, A, B, rs2);
        PeCand* b = pe_list(ps, st, A, B, ro2);
        // sites that wrapped below zero (the last ones of an ascending list) make the distances negative: the loop itself decides
        if ((a[na - 1].site >> 63) || (nbb && (b[nbb - 1].site >> 63))) {
            if (lane == src) ps.len[ro2] = (u32)serial(a, na, b, nbb, mxd, mnd);
            continue;
        }
        long len2 = 0;
        for (long base = 0; base < nbb; base += 64) {
            const long j = base + lane;
            PeCand e; e.site = 0; e.err = 0; e.end = 0;
            bool keep = false;
            if (j < nbb) {
                e = b[j];
                const long i = pe_lower_bound(a, na, e.site > (u64)mxd ? e.site - (u64)mxd : 0);      // the first a not more than maxd below b[j]
                keep = i < na && (a[i].site <= e.site || a[i].site - e.site <= (u64)mxd);             // ... and not more than maxd above
            }
            const unsigned long long kb = __ballot(keep);
            if (keep) b[len2 + __popcll(kb & ((1ull << lane) - 1))] = e;
            len2 += __popcll(kb);
        }
        if (lane == src) ps.len[ro2] = (u32)len2;
    }
}

// ================================================================================================
// Paired-end sensitive mode (Map_Pair_Seq_end_to_end, Schema.cpp:19953-21459)
// ================================================================================================
// Seeding of both mates is the same state machine as fast mode (first seed, 1-mismatch second seed, remaining
// seeds; process_rest_seed[_filter]_debug, Schema.cpp:17574 / 16298), so k_seed_* + k_locate + k_vote_pe are
// shared.  What differs is the order of verification -- the mate with fewer first-seed candidates is verified
// completely (round 1), the other mate's votes are kept only where a verified hit of the first lies within the
// insert window (select_suit_candidates, 4775) and verified (round 2) -- and the rescue: a mate left without a
// hit is re-seeded (reseed_filter, 16678) with fixed segments chosen from its recorded seeds (select_best_seeds,
// 16630) plus seeds sliding by 8, filtered by the verified mate and verified (round 3).

// exists a verified hit of the mate with mind <= |distance| <= maxd; `next_start` is the reference's running
// lower bound (sites arrive in ascending order)
DEVI bool pes_suit(const PeCand* mate, int mate_occ, int& next_start, u64 site, long long maxd, long long mind)
{
    for (int i = next_start; i < mate_occ; i++) {
        const u64 ms = mate[i].site;
        if (ms > site) {
            const long long d = (long long)(ms - site);
            if (d > maxd) return false;
            if (d >= mind) return true;
        } else {
            const long long d = (long long)(site - ms);
            if (d > maxd) next_start = i + 1;
            else if (d >= mind) return true;
        }
    }
    return false;
}

// which mate goes first (Schema.cpp:20870): the one with fewer candidates after its FIRST seed
__global__ void __launch_bounds__(256)
k_pes_order(long n, ReadState st, SeedCarry sc, PeState ps)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    ps.dead[p] = 0; ps.both[p] = 0; ps.npair[p] = 0; ps.sbd[p] = 0;
    u32 c[2];
    for (int m = 0; m < 2; m++) {
        const long r = p + m * n;
        const int v = st.verdict[r], ns = st.n_seeds[r];
        const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        // first seed recorded <=> seeds[0].off == 0 (every later seed starts at an offset >= 1)
        c[m] = (v == 1 || v == 4) ? 0u : (ns >= 1 && my[0].off == 0 ? my[0].hits : 0u);
        // full_seed_id: the terminate seeds that produced candidates; the fixed second seed of a 1-mismatch read
        // is not among them, and when it was usable no further seed was run (extra_seed_flag == 0)
        ps.full[r] = (u8)((sc.flag_c[r] && !sc.flag_d[r]) ? 1 : ns);
        ps.roff[r] = 0;
    }
    const int f = c[0] <= c[1] ? 0 : 1;
    ps.first[p] = (u8)f;
    const long rF = p + (long)f * n;
    const int vF = st.verdict[rF];
    if (vF == 0) ps.dead[p] = 1;                 // best_mapp_occ == 0 -> next pair
    else if (vF == 3) ps.vround[rF] = 1;
}

// pes_suit without its running lower bound: it returns whether ANY verified hit of the mate lies mind <= |distance| <= maxd from
// the site (the bound only skips hits more than maxd below the site, which a larger site cannot use either; the scan ends at the
// first hit more than maxd above it) -- two binary searches in the ascending list.  Sites below 2^63 only (the caller checks).
DEVI bool pes_suit_any(const PeCand* a, int na, u64 s, long long maxd, long long mind)
{
    if (na == 0 || maxd < 0) return false;
    const u64 mn = mind > 0 ? (u64)mind : 0;
    if (mn > (u64)maxd) return false;
    long j = pe_lower_bound(a, na, s + mn);                                        // hits above the site: [s + mn, s + maxd]
    if (j < na && a[j].site - s <= (u64)maxd) return true;
    if (s < mn) return false;
    j = pe_lower_bound(a, na, s > (u64)maxd ? s - (u64)maxd : 0);                  // hits below (or on) it: [s - maxd, s - mn]
    return j < na && a[j].site <= s - mn;
}
// after round 1: filter the second mate's votes by the first mate's verified hits (generate_candidate_votes_shift_filter)
// One lane per pair; a pair whose two lists hold more than 64 entries is done by the whole wave afterwards (pes_suit_any per entry,
// kept entries compacted by ballot) unless a site of its lists wrapped below zero.
__global__ void __launch_bounds__(64)
k_pes_second(long n, ReadGeom gm, PeIns pi, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool act = p < n && !ps.dead[p];
    long long maxd = 0, mind = 0; int large_k;
    long rF = 0, rS = 0;
    int occF = 0;
    if (act) {
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        const int f = ps.first[p];
        rF = p + (long)f * n; rS = p + (long)(1 - f) * n;
        occF = ps.occ[rF];
        if (occF == 0) { ps.dead[p] = 1; act = false; }
        else if (st.verdict[rS] != 3) act = false;             // direct hits / 1-mismatch exit / nothing: no verification
    }
    const long nb = act ? (long)ps.len[rS] : 0;
    const bool coop = act && nb + occF > 64;
    if (act && !coop) {
        const PeCand* a = pe_list(ps, st, A, B, rF);
        PeCand* b = pe_list(ps, st, A, B, rS);
        long kept = 0;
        int next_start = 0;
        for (long j = 0; j < nb; j++) {
            const PeCand cj = b[j];
            if (pes_suit(a, occF, next_start, cj.site, maxd, mind)) b[kept++] = cj;
        }
        ps.len[rS] = (u32)kept;
        ps.vround[rS] = 2;
    }
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long pp = (long)__shfl((long long)p, src, 64);
        long long mxd, mnd; int lk;
        pe_bounds(gm, pi, pp, n, mxd, mnd, lk);
        const int f = ps.first[pp];
        const long rF2 = pp + (long)f * n, rS2 = pp + (long)(1 - f) * n;
        const int na = ps.occ[rF2];
        const long nb2 = (long)ps.len[rS2];
        const PeCand* a = pe_list(ps, st, A, B, rF2);
        PeCand* b = pe_list(ps, st, A, B, rS2);
        if ((a[na - 1].site >> 63) || (nb2 && (b[nb2 - 1].site >> 63))) {
            if (lane == src) {
                long kept = 0;
                int next_start = 0;
                for (long j = 0; j < nb2; j++) {
                    const PeCand cj = b[j];
                    if (pes_suit(a, na, next_start, cj.site, mxd, mnd)) b[kept++] = cj;
                }
                ps.len[rS2] = (u32)kept; ps.vround[rS2] = 2;
            }
            continue;
        }
        long kept = 0;
        for (long base = 0; base < nb2; base += 64) {
            const long j = base + lane;
            PeCand cj; cj.site = 0; cj.err = 0; cj.end = 0;
            bool keep = false;
            if (j < nb2) { cj = b[j]; keep = pes_suit_any(a, na, cj.site, mxd, mnd); }
            const unsigned long long kb = __ballot(keep);      // every entry of the step is in registers before the first is stored
            if (keep) b[kept + __popcll(kb & ((1ull << lane) - 1))] = cj;
            kept += __popcll(kb);
        }
        if (lane == src) { ps.len[rS2] = (u32)kept; ps.vround[rS2] = 2; }
    }
}

// pairs whose second mate has no hit are re-seeded
__global__ void __launch_bounds__(256)
k_pes_reseed_flag(long n, PeState ps, u32* __restrict__ flag)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const long rS = p + (long)(1 - ps.first[p]) * n;
    flag[p] = (!ps.dead[p] && ps.occ[rS] == 0) ? 1u : 0u;
}

// reseed_filter's seeding (Schema.cpp:16678-16900) with select_best_seeds (16630): up to three fixed segments
// (count_hash_table) and then count_backward_as_much_1_terminate seeds sliding by 8.  One re-seeded mate per lane.
template <bool PACKED, bool KG = false>
__global__ void __launch_bounds__(64)
k_pes_reseed(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, long n, const u64* __restrict__ count_ptr,
             const u32* __restrict__ plist, ReadState st, PeState ps, u32* __restrict__ rcnt, unsigned long long* __restrict__ counters)
{
    __shared__ u64 s_c3[KG ? 27 : 1];
    const u64* c3 = KG ? kgram_c3(ix, s_c3) : nullptr;
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    LaneCounters lc = {0, 0, 0, 0, 0};
    if (it < (long)*count_ptr) {
        const long p = plist[it];
        const long r = p + (long)(1 - ps.first[p]) * n;
        const char* rd = seq + (size_t)r * stride;
        const u64* prow = PACKED ? pr.base + (size_t)r * pr.pwords : nullptr;
        const bool dirty = PACKED ? pr.dirty[r] != 0 : false;
        const int L = gm.rl(r);
        SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        const int full = ps.full[r];
        int rs[3], rl[3], rn = 0;
        int s0 = 0, s1 = 0;
        if (full >= 1) s0 = my[0].off;
        if (full >= 2) {
            s1 = my[1].off;
            rn = 2;
            rs[0] = s0; rl[0] = s1 - s0;
            rs[1] = (int)my[full - 2].off + (int)my[full - 2].len; rl[1] = L - rs[1];
        } else if (full == 1) {
            rn = 2;
            rs[0] = s0; rl[0] = L / 2;
            rs[1] = rs[0] + rl[0]; rl[1] = L - rs[1];
        }
        // full == 0: the reference reads index -1 of two malloc'ed int arrays (Schema.cpp:16657), which is the zero
        // upper half of the allocator's chunk-size word: the whole read becomes one fixed seed
        const int last = full >= 1 ? (int)my[full - 1].off + (int)my[full - 1].len : 0;
        if (last < L) { rs[rn] = last; rl[rn] = L - last; rn++; }
        const int max_seed = L / 10 == 0 ? 25 : (L / 10 - 1 > 25 ? 25 : L / 10 - 1);
        const u64 max_hits = 1000, avail = 20;
        int ns = 0, seed_id = 0;
        u64 ncand = 0;
        typename std::conditional<PACKED, SearchP, Search>::type S; SeedHit h;
        while (seed_id < rn) {
            const int tm = rs[seed_id], ml = rl[seed_id];
            if constexpr (PACKED) {
                if (search_begin_p<true>(ix, prow, pr.W, dirty, tm + ml, tm, S, h, lc.n_hash))
                    while (!search_step_p<true, KG>(ix, tm + ml, S, h, lc.n_ext, c3, &lc.n_jump)) {}
            } else {
                if (search_begin<true>(ix, rd, tm + ml, tm, S, h, lc.n_hash))
                    while (!search_step<true>(ix, rd, tm + ml, S, h, lc.n_ext)) {}
            }
            if (h.hits == 1) seed_record(my, ns, ncand, h.sp, 1, (u64)ml, (u64)tm);
            else if ((u64)ml >= avail && h.hits <= max_hits) { if (h.hits != 0) seed_record(my, ns, ncand, h.sp, h.hits, (u64)ml, (u64)tm); }
            else if (L - tm == ml) break;
            seed_id++;
        }
        int tm = full > 1 ? (s0 + s1) / 2 : 4;
        while (seed_id < max_seed && tm < L) {
            if constexpr (PACKED) {
                if (search_begin_p<false>(ix, prow, pr.W, dirty, L, tm, S, h, lc.n_hash))
                    while (!search_step_p<false, KG>(ix, L, S, h, lc.n_ext, c3, &lc.n_jump)) {}
            } else {
                if (search_begin<false>(ix, rd, L, tm, S, h, lc.n_hash))
                    while (!search_step<false>(ix, rd, L, S, h, lc.n_ext)) {}
            }
            if (h.hits == 1) seed_record(my, ns, ncand, h.sp, 1, h.ml, (u64)tm);
            else if (h.ml >= avail && h.hits <= max_hits) { if (h.hits != 0) seed_record(my, ns, ncand, h.sp, h.hits, h.ml, (u64)tm); }
            else if ((u64)(L - tm) == h.ml) break;
            tm += 8;
            seed_id++;
        }
        st.n_seeds[r] = (u8)ns;
        rcnt[it] = (u32)ncand;
    }
    flush_counters(counters, lc, 2);
}

// locate + sort + filtered votes of one re-seeded mate; the list goes to the R buffer (cur = 2)
#define PESV_LONG 32        // re-seeded mates with more candidates than this go to k_pes_vote_long (a block per mate)
__global__ void __launch_bounds__(64)
k_pes_vote(DevIndex ix, long n, ReadGeom gm, PeIns pi, const u64* __restrict__ count_ptr, const u32* __restrict__ plist,
           const u64* __restrict__ roff, ReadState st, PeState ps, u64* __restrict__ rcand, PeCand* __restrict__ A, PeCand* __restrict__ B,
           u32* __restrict__ long_flag)
{
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= (long)*count_ptr) return;
    if (long_flag) long_flag[it] = 0;
    const long p = plist[it];
    const int f = ps.first[p];
    const long rF = p + (long)f * n, r = p + (long)(1 - f) * n;
    const int k = gm.rk(gm.rl(r));
    long long maxd, mind; int large_k;
    pe_bounds(gm, pi, p, n, maxd, mind, large_k);
    const u64 o0 = roff[it], o1 = roff[it + 1];
    ps.roff[r] = o0;
    const long nc = (long)(o1 - o0);
    if (nc == 0) { ps.len[r] = 0; ps.vround[r] = 0; return; }       // no candidate: best_mapp_occ stays 0
    if (long_flag && nc > PESV_LONG) { long_flag[it] = 1; return; }   // a mate inside a repeat family: hundreds of candidates, sorted by a block
    const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
    const int ns = st.n_seeds[r];
    u64* c = rcand + o0;
    long o = 0;
    for (int s = 0; s < ns; s++) {
        const u64 sp = my[s].sp, adj = (u64)my[s].len + (u64)my[s].off;
        for (u32 j = 0; j < my[s].hits; j++) c[o++] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + j)) - adj;
    }
    sort_u64_asc(c, nc);
    const PeCand* a = pe_list(ps, st, A, B, rF);
    const int occF = ps.occ[rF];
    PeCand* out = ps.R + o0;
    long nv = 0;
    int next_start = 0;
    u64 pre = c[0];
    for (long i = 1; i <= nc; i++) {
        if (i < nc && c[i] == pre) continue;
        const u64 site = (i < nc) ? (pre < (u64)k ? 0 : pre - (u64)k) : (pre >= (u64)k ? pre - (u64)k : 0);
        if (pes_suit(a, occF, next_start, site, maxd, mind)) { out[nv].site = site; out[nv].err = 0; out[nv].end = 0; nv++; }
        if (i < nc) pre = c[i];
    }
    ps.cur[r] = 2; ps.len[r] = (u32)nv; ps.vround[r] = 3;
}

// k_pes_vote for the mates it flagged: a block per mate -- candidates located into LDS and sorted (vl_locate_sort), distinct sites
// (vl_run_ends), the window test per site, kept sites compacted in order.  Lists beyond the LDS capacity and lists with sites that
// wrapped below zero take k_pes_vote's loop on one lane.
template <int CAP, int BLOCK, int LO>
__global__ void __launch_bounds__(BLOCK)
k_pes_vote_long(DevIndex ix, long n, ReadGeom gm, PeIns pi, const u64* __restrict__ count_ptr, const u32* __restrict__ items,
                const u32* __restrict__ plist, const u64* __restrict__ roff, ReadState st, PeState ps, u64* __restrict__ rcand,
                PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    __shared__ u64 keys[CAP];
    __shared__ u16 endpos[CAP];
    __shared__ u32 sh_pref[BMBS_MAX_SEEDS + 1];
    __shared__ int sh_w[2 * (BLOCK / 64) + 1];
    const long total = (long)*count_ptr;
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const long it = items[item];
        const u64 o0 = roff[it], o1 = roff[it + 1];
        const long nc = (long)(o1 - o0);
        if (nc <= LO || (CAP != VL_CAP && nc > CAP)) continue;          // another instance's size class
        const long p = plist[it];
        const int f = ps.first[p];
        const long rF = p + (long)f * n, r = p + (long)(1 - f) * n;
        const int k = gm.rk(gm.rl(r));
        long long maxd, mind; int large_k;
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        const int ns = st.n_seeds[r];
        const PeCand* a = pe_list(ps, st, A, B, rF);
        const int occF = ps.occ[rF];
        PeCand* out = ps.R + o0;
        if (nc > CAP) {
            // beyond the LDS capacity: vl_sort_huge, the tiles parked in the output segment (16 bytes per candidate: room for the 8-byte
            // sites); the merged list -- in the candidate segment -- is made distinct and filtered 256 sites a step
            u64* c = rcand + o0;
            vl_sort_huge<CAP, BLOCK>(ix, my, ns, nc, keys, sh_pref, reinterpret_cast<u64*>(out), c);
            if ((occF > 0 && (a[occF - 1].site >> 63)) || (c[nc - 1] >> 63)) {
                // sites that wrapped below zero: the reference's loop over the sorted list, one lane
                if (threadIdx.x == 0) {
                    long nv = 0;
                    int next_start = 0;
                    u64 pre = c[0];
                    for (long i = 1; i <= nc; i++) {
                        if (i < nc && c[i] == pre) continue;
                        const u64 site = pre < (u64)k ? 0 : pre - (u64)k;
                        if (pes_suit(a, occF, next_start, site, maxd, mind)) { out[nv].site = site; out[nv].err = 0; out[nv].end = 0; nv++; }
                        if (i < nc) pre = c[i];
                    }
                    ps.cur[r] = 2; ps.len[r] = (u32)nv; ps.vround[r] = 3;
                }
                __syncthreads();
                continue;
            }
            int running = 0;
            for (long base = 0; base < nc; base += BLOCK) {
                const long i = base + (long)threadIdx.x;
                bool keep = false;
                u64 site = 0;
                if (i < nc) {
                    const u64 key = c[i];
                    if (i == nc - 1 || c[i + 1] != key) { site = key < (u64)k ? 0 : key - (u64)k; keep = pes_suit_any(a, occF, site, maxd, mind); }
                }
                int tot;
                const int pre = vl_prefix(keep, sh_w, tot);
                if (keep) { PeCand c2; c2.site = site; c2.err = 0; c2.end = 0; out[running + pre] = c2; }
                running += tot;
            }
            if (threadIdx.x == 0) { ps.cur[r] = 2; ps.len[r] = (u32)running; ps.vround[r] = 3; }
            __syncthreads();
            continue;
        }
        bool serial = occF > 0 && (a[occF - 1].site >> 63);
        if (!serial) {
            vl_locate_sort<(CAP + BLOCK - 1) / BLOCK>(ix, my, ns, (int)nc, keys, sh_pref);
            serial = (keys[nc - 1] >> 63) != 0;
        }
        if (serial) {
            if (threadIdx.x == 0) {
                u64* c = rcand + o0;
                long o = 0;
                for (int s2 = 0; s2 < ns; s2++) {
                    const u64 sp = my[s2].sp, adj = (u64)my[s2].len + (u64)my[s2].off;
                    for (u32 j = 0; j < my[s2].hits; j++) c[o++] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + j)) - adj;
                }
                sort_u64_asc(c, nc);
                long nv = 0;
                int next_start = 0;
                u64 pre = c[0];
                for (long i = 1; i <= nc; i++) {
                    if (i < nc && c[i] == pre) continue;
                    const u64 site = pre < (u64)k ? 0 : pre - (u64)k;
                    if (pes_suit(a, occF, next_start, site, maxd, mind)) { out[nv].site = site; out[nv].err = 0; out[nv].end = 0; nv++; }
                    if (i < nc) pre = c[i];
                }
                ps.cur[r] = 2; ps.len[r] = (u32)nv; ps.vround[r] = 3;
            }
            __syncthreads();
            continue;
        }
        const int nd = vl_run_ends(keys, (int)nc, endpos, sh_w);
        int running = 0;
        for (int base = 0; base < nd; base += BLOCK) {
            const int e = base + (int)threadIdx.x;
            bool keep = false;
            u64 site = 0;
            if (e < nd) {
                const u64 key = keys[endpos[e]];
                site = key < (u64)k ? 0 : key - (u64)k;
                keep = pes_suit_any(a, occF, site, maxd, mind);
            }
            int tot;
            const int pre = vl_prefix(keep, sh_w, tot);
            if (keep) { PeCand c2; c2.site = site; c2.err = 0; c2.end = 0; out[running + pre] = c2; }
            running += tot;
        }
        if (threadIdx.x == 0) { ps.cur[r] = 2; ps.len[r] = (u32)running; ps.vround[r] = 3; }
        __syncthreads();
    }
}

// new_faster_verify_pairs (Schema.cpp:15773-15900) + hand-over of the winning candidates to K11-K13
// What the reference's loop leaves behind, as a summary of an ORDERED run of (i, j) hits that can be merged left to right:
// the lowest error sum m, where it first occurs, how often it occurs (c), and the lowest sum among the hits before that first
// occurrence (pm) -- the loop's `second` is the running best at the moment the final best was first met (not the true runner-up),
// or the best itself when it was met again afterwards.
struct PairSum { int m, c, pm; long i, j; };
DEVI PairSum pair_comb(const PairSum& L, const PairSum& R)
{
    if (R.c == 0) return L;
    if (L.c == 0) return R;
    PairSum o;
    if (L.m < R.m) o = L;
    else if (L.m > R.m) { o = R; o.pm = L.m < R.pm ? L.m : R.pm; }
    else { o = L; o.c = L.c + R.c; }
    return o;
}
__global__ void __launch_bounds__(64)
k_pe_pair(DevIndex ix, long n, ReadGeom gm, PeIns pi, int ambiguous_out, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool act = p < n;
    long long maxd = 0, mind = 0; int large_k = 0;
    const long r1 = p, r2 = p + n;
    if (act) {
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        st.job_flag[r1] = 0; st.job_flag[r2] = 0;
        st.red_status[r1] = 0; st.red_status[r2] = 0;
        if (ps.dead[p]) act = false;
    }
    const int n1 = act ? ps.occ[r1] : 0, n2 = act ? ps.occ[r2] : 0;
    const PeCand* a = act ? pe_list(ps, st, A, B, r1) : nullptr;
    const PeCand* b = act ? pe_list(ps, st, A, B, r2) : nullptr;
    int mapping_pair = 0;
    long long bi = 0, bj = 0;
    u32 sbd = 0;
    // the reference's loop itself, one lane
    auto serial = [&](const PeCand* a_, long n1_, const PeCand* b_, long n2_, long long mxd, long long mnd, int lk, int& mp_out, u32& sbd_out,
                      long long& bi_out, long long& bj_out) {
        int mp = 0;
        int best_sum = 4 * lk + 2;
        long long second = (long long)best_sum * 2, bi_ = 0, bj_ = 0;
        bool early = false;
        long first = 0;
        for (long i = 0; i < n1_ && !early; i++) {
            for (long j = first; j < n2_; j++) {
                bool hit = false;
                if (a_[i].site > b_[j].site) {
                    const long long d = (long long)(a_[i].site - b_[j].site);
                    if (d > mxd) first = j + 1;
                    else if (d >= mnd) hit = true;
                } else {
                    const long long d = (long long)(b_[j].site - a_[i].site);
                    if (d > mxd) break;
                    if (d >= mnd) hit = true;
                }
                if (hit) {
                    const long long cur = (long long)a_[i].err + (long long)b_[j].err;
                    if (cur < best_sum) { second = best_sum; best_sum = (int)cur; bi_ = i; bj_ = j; mp = 1; }
                    else if (cur == best_sum) {
                        second = best_sum; mp++;
                        if (best_sum == 0) { early = true; break; }
                    }
                }
            }
        }
        mp_out = mp; bi_out = bi_; bj_out = bj_;
        sbd_out = early ? 0u : (mp != 0 ? (u32)(second - best_sum) : 0u);
    };
    // two long lists (both mates inside a repeat family: hundreds of verified copies each) go to the whole wave: with a lower bound of
    // the distance <= 0 the hits of a[i] are exactly the b[j] within maxd of it, in order -- a lane takes an a[i], finds its window in b
    // by binary search and sums it up, and the lanes' summaries merge in order (pair_comb)
    const bool coop = act && n1 > 0 && n2 > 0 && (long)n1 + n2 > 64 && mind <= 0 && maxd >= 0;
    if (act && !coop && n1 > 0 && n2 > 0) serial(a, n1, b, n2, maxd, mind, large_k, mapping_pair, sbd, bi, bj);
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long pp = (long)__shfl((long long)p, src, 64);
        long long mxd, mnd; int lk;
        pe_bounds(gm, pi, pp, n, mxd, mnd, lk);
        const long m1 = ps.occ[pp], m2 = ps.occ[pp + n];
        const PeCand* a2 = pe_list(ps, st, A, B, pp);
        const PeCand* b2 = pe_list(ps, st, A, B, pp + n);
        if ((a2[m1 - 1].site >> 63) || (b2[m2 - 1].site >> 63)) {     // sites that wrapped below zero: the loop itself decides
            if (lane == src) serial(a2, m1, b2, m2, mxd, mnd, lk, mapping_pair, sbd, bi, bj);
            continue;
        }
        PairSum tot; tot.m = 0; tot.c = 0; tot.pm = 0x7fffffff; tot.i = 0; tot.j = 0;
        for (long base = 0; base < m1; base += 64) {
            const long i = base + lane;
            PairSum me; me.m = 0; me.c = 0; me.pm = 0x7fffffff; me.i = i; me.j = 0;
            if (i < m1) {
                const PeCand e = a2[i];
                const u64 hi = e.site + (u64)mxd;
                for (long j = pe_lower_bound(b2, m2, e.site > (u64)mxd ? e.site - (u64)mxd : 0); j < m2; j++) {
                    const PeCand f = b2[j];
                    if (f.site > hi) break;
                    const int cur = (int)(e.err + f.err);
                    if (me.c == 0 || cur < me.m) { if (me.c) me.pm = me.m < me.pm ? me.m : me.pm; me.m = cur; me.c = 1; me.j = j; }
                    else if (cur == me.m) me.c++;
                }
            }
            // ordered reduction over the lanes: lane l collects lanes l .. l + 2 off - 1
            for (int off = 1; off < 64; off <<= 1) {
                PairSum o;
                o.m = __shfl_down(me.m, off, 64); o.c = __shfl_down(me.c, off, 64); o.pm = __shfl_down(me.pm, off, 64);
                o.i = (long)__shfl_down((long long)me.i, off, 64); o.j = (long)__shfl_down((long long)me.j, off, 64);
                if ((lane & (2 * off - 1)) == 0 && lane + off < 64) me = pair_comb(me, o);
            }
            PairSum ch;
            ch.m = __shfl(me.m, 0, 64); ch.c = __shfl(me.c, 0, 64); ch.pm = __shfl(me.pm, 0, 64);
            ch.i = (long)__shfl((long long)me.i, 0, 64); ch.j = (long)__shfl((long long)me.j, 0, 64);
            tot = pair_comb(tot, ch);
        }
        if (lane == src) {
            const int init = 4 * lk + 2;
            if (tot.c == 0) { mapping_pair = 0; sbd = 0; }
            else {
                bi = tot.i; bj = tot.j;
                if (tot.c >= 2) { mapping_pair = tot.m == 0 ? 2 : tot.c; sbd = 0; }     // met again: second = best (sum 0: the loop stops at the second)
                else { mapping_pair = 1; sbd = (u32)((tot.pm < init ? tot.pm : init) - tot.m); }
            }
        }
    }
    if (!act) return;
    ps.npair[p] = mapping_pair; ps.sbd[p] = sbd;
    if (mapping_pair == 1 || (ambiguous_out && mapping_pair > 1)) {          // Schema.cpp:19342-19347
        st.best_site[r1] = a[bi].site; st.best_end[r1] = a[bi].end; st.best_err[r1] = a[bi].err;
        st.best_site[r2] = b[bj].site; st.best_end[r2] = b[bj].end; st.best_err[r2] = b[bj].err;
        st.red_status[r1] = 1; st.red_status[r2] = 1;
        // a mate that left through the 1-mismatch exit has exactly one mismatch, at mm_site, on the un-gapped diagonal:
        // fast_recalculate_bs_Cigar (ksw.cpp:2578) would only re-derive NM 1 / <L>M / minus one penalty, which k_finalize_pe
        // writes directly -- unless its window leaves the strand, where the reference aligns against an all-zero window
        const long rr[2] = {r1, r2};
        const PeCand w2[2] = {a[bi], b[bj]};
        for (int m = 0; m < 2; m++) {
            const int Lm = gm.rl(rr[m]), km = gm.rk(Lm);
            const bool direct = st.verdict[rr[m]] == 2 && window_valid(ix, w2[m].site, (u64)(Lm + 2 * km), w2[m].site < ix.G);
            st.job_flag[rr[m]] = (w2[m].err != 0 && !direct) ? 1u : 0u;
        }
    }
}

// per-pair post-processing (Schema.cpp:19330-19480): placement of both mates (output_sam_end_to_end_return,
// 9188), TLEN (Schema.h:1587), insert/chromosome-end checks, MAPQ over k1+k2, flags 99/83/147/163, stats
__global__ void __launch_bounds__(256)
k_finalize_pe(DevIndex ix, ScoreParams sp, const int* __restrict__ pen_lut, const char* __restrict__ seq, const char* __restrict__ qual,
              const char* __restrict__ qual2,
              int stride, const u8* __restrict__ mapq_lut, const u32* __restrict__ mapq_off, int unit, ReadGeom gm, int min_ins, int max_ins,
              int ambiguous_out, long n,
              ReadState st, PeState ps, const int* __restrict__ a_start, const int* __restrict__ a_end,
              const u32* __restrict__ a_nm, const int* __restrict__ a_score, const int* __restrict__ a_nops, int max_ops, u32 cigar_base,
              bmbs_result_dev* __restrict__ res, unsigned long long* __restrict__ stats)
{
    __shared__ unsigned long long sh[5];
    __shared__ u64 s_cs[BMBS_CS_LDS];
    if (threadIdx.x < 5) sh[threadIdx.x] = 0;
    const u64* cs = chrom_table(ix, s_cs);
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;          // this lane's contribution to the five counters
    if (p < n) {
        bmbs_result_dev o[2];
        for (int m = 0; m < 2; m++) {
            o[m].pos = 0; o[m].cigar_off = 0; o[m].chrom = -1; o[m].status = 0; o[m].mapq = 0; o[m].flag = 0; o[m].nm = 0;
            o[m].score = 0; o[m].n_cigar = 0; o[m].path = 0; o[m].n_cand = sat16(st.n_cand[p + m * n]); o[m].tlen = 0;
        }
        const int np = ps.dead[p] ? 0 : ps.npair[p];
        int status = 0;
        if (np > 1 && !ambiguous_out) status = 2;
        else if (np >= 1) {
            long long site_pos[2], matched[2];
            int rflag[2], chrom[2], score[2]; u32 nm[2];
            bool inrange = true;
            for (int m = 0; m < 2; m++) {
                const long r = p + m * n;
                const u64 site = st.best_site[r];
                long long start_site, end_site;
                if (st.job_flag[r]) {
                    const u64 jb = st.job_off[r];
                    start_site = a_start[jb]; end_site = a_end[jb]; nm[m] = a_nm[jb]; score[m] = a_score[jb];
                    const int no = a_nops[jb];
                    o[m].cigar_off = cigar_base + (u32)(jb * (u64)max_ops);
                    o[m].n_cigar = no < 0 ? 255 : (u8)no;
                } else {
                    const int Lm = gm.rl(r);
                    end_site = st.best_end[r]; start_site = end_site - Lm + 1; nm[m] = 0; score[m] = 0;
                    if (st.best_err[r] != 0) {
                        // 1-mismatch exit (see k_pe_pair): NM 1, score = minus the penalty at mm_site; mate 2 rows carry their
                        // qualities in FASTQ order for a reverse-complemented read (need_reverse_quality = 1)
                        const int mv = st.mm_site[r], ms = mv & 0x7fff;          // bit 15: the read has 'N' there (k_seed_decide)
                        const int qi = m == 1 ? Lm - 1 - ms : ms;
                        nm[m] = 1;
                        score[m] = (mv & 0x8000) ? -sp.np : -pen_lut[(unsigned char)qual_row(qual, qual2, (u32)n, (u32)r, stride)[qi]];
                    }
                }
                u64 loc = site;
                if (loc >= ix.G) { loc = loc + (u64)end_site; loc = ix.G * 2 - loc - 1; rflag[m] = 16; }
                else { loc = loc + (u64)start_site; rflag[m] = 0; }
                int c = 0;
                c = chrom_of(cs, ix.n_chrom, loc);
                if (c >= ix.n_chrom) { c = ix.n_chrom - 1; inrange = false; }
                chrom[m] = c;
                site_pos[m] = (long long)(loc + 1 - cs[c]);
                matched[m] = end_site - start_site + 1;
                const long long clen = (long long)(cs[c + 1] - cs[c]);
                if ((u64)site_pos[m] + (u64)matched[m] > (u64)clen + 1) inrange = false;
            }
            long long mn = site_pos[0], mx = site_pos[0] + matched[0] - 1;
            if (site_pos[0] > site_pos[1]) mn = site_pos[1];
            if (mx < site_pos[1] + matched[1] - 1) mx = site_pos[1] + matched[1] - 1;
            const int tlen = (int)(mx - mn + 1);
            if (tlen <= max_ins && tlen >= min_ins && inrange) {
                status = np == 1 ? 1 : 2;
                // MAP_Calculation over error_threshold1 + error_threshold2 (Schema.cpp:19445)
                const int L1 = gm.rl(p), L2 = gm.rl(p + n);
                const u32 kk = (u32)(gm.rk(L1) + gm.rk(L2)), sb = ps.sbd[p];
                const int range = unit * (int)kk;
                int sd = score[0] + score[1] + range; if (sd < 0) sd = 0; if (sd > range) sd = range;
                const u32 ed = sb > kk ? kk + 1 : sb;
                const int mapq = mapq_lut[mapq_off[kk] + (size_t)ed * (range + 1) + sd];
                for (int m = 0; m < 2; m++) {
                    o[m].pos = (u64)site_pos[m]; o[m].chrom = chrom[m]; o[m].mapq = (u8)mapq; o[m].nm = (u16)nm[m];
                    o[m].score = (int16_t)score[m]; o[m].tlen = (u32)tlen; o[m].path = 3;
                }
                o[0].flag = (u16)(rflag[0] == 0 ? (1 | 2 | 32 | 64) : (1 | 2 | 16 | 64));
                o[1].flag = (u16)(rflag[1] == 0 ? (1 | 2 | 16 | 128) : (1 | 2 | 32 | 128));
                if (np == 1) s1 = 1;
                s3 = (u32)(L1 + L2);
                s4 = nm[0] + nm[1];
            } else status = 3;
        }
        if (status == 2) s2 = 1;
        s0 = 1;
        o[0].status = (u8)status; o[1].status = (u8)status;
        res[2 * p] = o[0]; res[2 * p + 1] = o[1];
    }
    wave_stats_add(sh, s0, s1, s2, s3, s4);
    __syncthreads();
    if (threadIdx.x < 5 && sh[threadIdx.x]) atomicAdd(&SHARD(stats)[threadIdx.x], sh[threadIdx.x]);
}
