// bitmapperbs_amd/csrc/k_filter.hip -- K7+K8: window fetch + BS banded Myers, one candidate per lane, 64-bit words
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
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
