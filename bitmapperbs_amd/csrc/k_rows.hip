// bitmapperbs_amd/csrc/k_rows.hip -- packed read rows (round 2)
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
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

// ---- packed rows handed over by the caller (bmbs_map_*_packed, round 5) ---------------------------------------------------------------
// Host format of a row (include/bmbs.h): W = ceil(L_max / 32) words of bases (base j in bits 2 (j % 32) .. +1 of word j / 32, A0 C1 G2 T3),
// then M = ceil(L_max / 64) words of marks (bit j % 64 of word j / 64: the character at j is 'N', its base bits are 0), rows `hw` words
// apart: 64 bytes for 150 bases where the ASCII row takes 160.  This kernel makes the lane's own packed rows of them (pack_words(L_max)
// words: the same two planes and the spare words the two-word loads read), sets the rows' dirty bytes, rebuilds the ASCII text of every
// 16-character piece that holds an 'N' (the only pieces whose text a kernel ever asks for) -- and, RC, reverse-complements the row on
// the way: mate 2 comes in FASTQ orientation and is mapped as its reverse complement (Process_Reads.cpp:262-267).  It replaces
// k_pe_prepare / k_pack_rows for such callers: 2 x 64 bytes read per pair instead of 2 x 160.
// One thread per 32-base word of an output row; bits at and beyond a read's length come out 0 whatever the caller left there.
DEVI u64 pk_win_bases(const u64* w, int nw, int s)          // bases s .. s+31 of a row of nw words (0 outside it)
{
    if (s <= -32 || s >= 32 * nw) return 0;
    if (s < 0) return w[0] << (2 * -s);
    const int i = s >> 5, sh = 2 * (s & 31);
    u64 v = w[i] >> sh;
    if (sh && i + 1 < nw) v |= w[i + 1] << (64 - sh);
    return v;
}
DEVI u32 pk_win_marks(const u64* m, int nm, int s)          // marks s .. s+31
{
    if (s <= -32 || s >= 64 * nm) return 0;
    if (s < 0) return (u32)(m[0] << -s);
    const int i = s >> 6, sh = s & 63;
    u64 v = m[i] >> sh;
    if (sh && i + 1 < nm) v |= m[i + 1] << (64 - sh);
    return (u32)v;
}
template <bool RC>
__global__ void __launch_bounds__(256)
k_rows_from_packed(const u64* __restrict__ src, int hw, ReadGeom gm, long row0, long n, u64* __restrict__ prow, int pwords, int W,
                   u32* __restrict__ dirty32, char* __restrict__ ascii, int stride)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = t / W;
    const int k = (int)(t - r * W);
    if (r >= n) return;
    const long row_id = row0 + r;
    const int L = gm.rl(row_id);
    const int M = (gm.L + 63) / 64;
    const u64* in = src + (size_t)r * hw;
    u64* out = prow + (size_t)row_id * pwords;
    const int j0 = 32 * k;                                   // first base of this thread's word
    // the 32 bases and marks of output positions j0 .. j0+31
    u64 b = 0; u32 mk = 0;
    const u32 inr = j0 >= L ? 0u : L - j0 >= 32 ? 0xffffffffu : (1u << (L - j0)) - 1u;      // output positions inside the read
    if (inr) {
        if (!RC) { b = in[k]; mk = pk_win_marks(in + W, M, j0); }
        else {
            const int s = L - 32 - j0;                       // input positions s .. s+31, reversed, are output positions j0 .. j0+31
            const u64 x = pk_win_bases(in, W, s);
            u64 rv = __brevll(x);
            rv = ((rv >> 1) & PK_EVEN) | ((rv & PK_EVEN) << 1);
            b = rv;
            mk = __brev(pk_win_marks(in + W, M, s));
        }
        mk &= inr;
        const u64 v = spread32(inr & ~mk);                   // real bases -> both bits of their pair
        const u64 vv = v | (v << 1);
        b = RC ? (b ^ vv) & vv : b & vv;                     // complement: code -> 3 - code; everything else 0
    }
    out[k] = b;
    // marks: this thread's 32 bits are one half of mask word k / 2
    reinterpret_cast<u32*>(out + W)[k] = mk;
    if ((W & 1) && k == W - 1) reinterpret_cast<u32*>(out + W)[W] = 0;                       // (the upper half of the last mask word when W is odd)
    if (k == 0) for (int q = W + M; q < pwords; q++) out[q] = 0;                             // the spare words
    if (mk) {
        atomicOr(&dirty32[row_id >> 2], 1u << (8 * (int)(row_id & 3)));
        // the text of the pieces that hold an 'N'
        char* arow = ascii + (size_t)row_id * stride;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const u32 pm = (mk >> (16 * h)) & 0xffffu;
            if (!pm || j0 + 16 * h >= stride) continue;
            u32 wv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u32 x = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int p = 16 * h + 4 * q + e;        // position inside this thread's 32
                    const u32 code = (u32)(b >> (2 * p)) & 3u;
                    u32 c = (0x54474341u >> (8 * code)) & 0xffu;                             // 'A' 'C' 'G' 'T'
                    if ((mk >> p) & 1u) c = 'N';
                    if (!((inr >> p) & 1u)) c = 0;
                    x |= c << (8 * e);
                }
                wv[q] = x;
            }
            *reinterpret_cast<uint4*>(arow + j0 + 16 * h) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
    }
}
