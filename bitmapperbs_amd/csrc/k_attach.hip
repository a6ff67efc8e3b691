// bitmapperbs_amd/csrc/k_attach.hip -- attach-time re-pack kernels
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
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

// the same words as two bit planes (DevIndex::gen2p)
__global__ void k_build_gen2p(const u64* __restrict__ gen2, u64 n_words, u64* __restrict__ out)
{
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const u64 v = gen2[w];
    auto squeeze = [](u64 x) -> u64 {              // bits 0, 2, 4 .. 62 -> bits 0 .. 31
        x &= 0x5555555555555555ull;
        x = (x | (x >> 1)) & 0x3333333333333333ull;
        x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
        x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
        x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
        x = (x | (x >> 16)) & 0x00000000ffffffffull;
        return x;
    };
    out[w] = squeeze(v) | (squeeze(v >> 1) << 32);
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
