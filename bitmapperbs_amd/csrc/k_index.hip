// bitmapperbs_amd/csrc/k_index.hip -- index primitives
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
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
