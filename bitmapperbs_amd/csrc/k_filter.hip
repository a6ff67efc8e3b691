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
// form spends fewer instructions per read character: the window is kept as two bit planes (bit 0 / bit 1 of the 2-bit letters, read
// from the index's planar copy of the genome, DevIndex::gen2p), the row's Peq is derived from them with the bisulfite rule folded in
// (T: plane 0 alone = {C, T}), and the characters come 16 per load.
DEVI void planes32(const DevIndex& ix, u64 d, u32& lo, u32& hi)      // the 32 bases starting at doubled coordinate d
{
    const u64 a = ix.gen2p[d >> 5], b = ix.gen2p[(d >> 5) + 1];             // (the array carries spare words at its end)
    const u32 s = (u32)(d & 31);
    lo = __builtin_amdgcn_alignbit((u32)b, (u32)a, s);
    hi = __builtin_amdgcn_alignbit((u32)(b >> 32), (u32)(a >> 32), s);
}
// W = u32: band <= 31 bits (k <= 15), 64 bases of each plane in a register pair; W = u64: band <= 63 bits (k <= 31), 96 bases.
// PACKED: the read comes from a packed row (prow, 32 bases per word; pW = its base words; dirty = it holds characters outside ACGT)
// Two forms of the window: a wave of u32 candidates whose windows all fit 192 bases (reads up to ~165 bases) fetches each window
// ONCE, up front (every 32-byte sector of the genome a candidate needs is then asked for exactly once: the kernel sits at the
// chip's rate of divergent DRAM requests, and a window streamed 32 bases at a time finds its sectors evicted again between
// visits); any other wave streams the window 32 bases per 32 rows.
constexpr int BPM_HELD = 6;                             // plane words (32 bases each) of a window held in registers
template <class W, bool PACKED = false>
DEVI void bpm_planes(const DevIndex& ix, const char* rd, int L, int k, u64 site, u32& out_err, int& out_end,
                     const u64* prow = nullptr, int pW = 0, bool dirty = false)
{
    constexpr bool WIDE = sizeof(W) == 8;
    out_err = 0xffffffffu; out_end = -1;
    const int p_len = L + 2 * k;
    const bool valid = window_valid(ix, site, (u64)p_len, site < ix.G);
    const int band = 2 * k + 1;
    const W bmask = ((W)1 << band) - 1;
    W VP = 0, VN = 0;
    int err = 0;
    u32 acc = 0;
    const int last_high = 2 * k;
    // every active lane of the wave reads the same packed row (one read's candidates fill the wave)
    [[maybe_unused]] bool uniform = false;
    if constexpr (PACKED && !WIDE) {
        const u64 pa = (u64)prow;
        const u32 p0 = __builtin_amdgcn_readfirstlane((u32)pa), p1 = __builtin_amdgcn_readfirstlane((u32)(pa >> 32));
        uniform = __all((u32)pa == p0 && (u32)(pa >> 32) == p1) != 0;
    }
    // One read character (row i of the matrix) as its 2-bit code (A0 C1 G2 T3): c0 / c1 = the code's bits spread over the word;
    // lo / hi = the two planes of the window bases i .. i + 2k.  A window base matches when both of its plane bits equal the
    // code's; a read T also matches a window C (plane 0 alone).  The row is 16 VALU instructions for W = u32 (two funnel shifts,
    // two bit-field extracts, 3-input boolean ops); the D0 bit that the error count needs is funnelled into `acc` and counted
    // once per 16 rows.  CHECKED also tests the read's end and the character's not-ACGT mark.
    auto row = [&](W lo, W hi, u32 c16, u32 m16, int c, int i, auto checked) {
        if constexpr (!WIDE) {
            // (the 3-input boolean instruction, truth tables with a = 0xF0, b = 0xCC, c = 0xAA)
            constexpr u32 A = 0xF0u, B = 0xCCu, C = 0xAAu;
            const u32 c0 = (u32)__builtin_amdgcn_sbfe((int)c16, 2 * c, 1), c1 = (u32)__builtin_amdgcn_sbfe((int)c16, 2 * c + 1, 1);
            const u32 t = __builtin_amdgcn_bitop3_b32(hi, c1, c0, (~(A ^ B) | (B & C)) & 0xffu);
            u32 eq = __builtin_amdgcn_bitop3_b32(lo, c0, t, (~(A ^ B) & C) & 0xffu);
            if (decltype(checked)::value) eq &= ((m16 >> c) & 1u) - 1u;
            u32 X = __builtin_amdgcn_bitop3_b32(eq, bmask, VN, ((A & B) | C) & 0xffu);
            const u32 D0 = __builtin_amdgcn_bitop3_b32(VP + (X & VP), VP, X, ((A ^ B) | C) & 0xffu);
            const u32 HN = VP & D0;
            const u32 HP = __builtin_amdgcn_bitop3_b32(VN, VP, D0, (A | ~(B | C)) & 0xffu);
            X = D0 >> 1;
            const u32 VN2 = X & HP, VP2 = __builtin_amdgcn_bitop3_b32(HN, X, HP, (A | ~(B | C)) & 0xffu);
            if (decltype(checked)::value) { if (i < L) { VN = VN2; VP = VP2; err += 1 - (int)(D0 & 1u); } }
            else { VN = VN2; VP = VP2; acc = __builtin_amdgcn_alignbit(D0, acc, 1); }
        } else {
            // the same row on the two halves of a 64-bit band: 31 instructions (the add carries from the low half into the high one)
            constexpr u32 A = 0xF0u, B = 0xCCu, C = 0xAAu;
            constexpr u32 T_HI = (~(A ^ B) | (B & C)) & 0xffu, T_EQ = (~(A ^ B) & C) & 0xffu, T_X = ((A & B) | C) & 0xffu,
                          T_D0 = ((A ^ B) | C) & 0xffu, T_P = (A | ~(B | C)) & 0xffu;
            const u32 c0 = (u32)__builtin_amdgcn_sbfe((int)c16, 2 * c, 1), c1 = (u32)__builtin_amdgcn_sbfe((int)c16, 2 * c + 1, 1);
            const u32 lol = (u32)lo, loh = (u32)(lo >> 32), hil = (u32)hi, hih = (u32)(hi >> 32);
            const u32 VPl = (u32)VP, VPh = (u32)(VP >> 32), VNl = (u32)VN, VNh = (u32)(VN >> 32);
            u32 eql = __builtin_amdgcn_bitop3_b32(lol, c0, __builtin_amdgcn_bitop3_b32(hil, c1, c0, T_HI), T_EQ);
            u32 eqh = __builtin_amdgcn_bitop3_b32(loh, c0, __builtin_amdgcn_bitop3_b32(hih, c1, c0, T_HI), T_EQ);
            if (decltype(checked)::value) { const u32 ok = ((m16 >> c) & 1u) - 1u; eql &= ok; eqh &= ok; }
            const u32 Xl = __builtin_amdgcn_bitop3_b32(eql, (u32)bmask, VNl, T_X), Xh = __builtin_amdgcn_bitop3_b32(eqh, (u32)(bmask >> 32), VNh, T_X);
            const u64 sum = VP + (((u64)(Xh & VPh) << 32) | (u64)(Xl & VPl));
            const u32 D0l = __builtin_amdgcn_bitop3_b32((u32)sum, VPl, Xl, T_D0), D0h = __builtin_amdgcn_bitop3_b32((u32)(sum >> 32), VPh, Xh, T_D0);
            const u32 HNl = VPl & D0l, HNh = VPh & D0h;
            const u32 HPl = __builtin_amdgcn_bitop3_b32(VNl, VPl, D0l, T_P), HPh = __builtin_amdgcn_bitop3_b32(VNh, VPh, D0h, T_P);
            const u32 Sl = __builtin_amdgcn_alignbit(D0h, D0l, 1), Sh = D0h >> 1;
            const u32 VN2l = Sl & HPl, VN2h = Sh & HPh;
            const u32 VP2l = __builtin_amdgcn_bitop3_b32(HNl, Sl, HPl, T_P), VP2h = __builtin_amdgcn_bitop3_b32(HNh, Sh, HPh, T_P);
            const W VN2 = ((W)VN2h << 32) | VN2l, VP2 = ((W)VP2h << 32) | VP2l;
            if (decltype(checked)::value) { if (i < L) { VN = VN2; VP = VP2; err += 1 - (int)(D0l & 1u); } }
            else { VN = VN2; VP = VP2; acc = __builtin_amdgcn_alignbit(D0l, acc, 1); }
        }
    };
    // The same row when every lane of the wave verifies a candidate of ONE read (the dense list keeps a read's candidates together, and
    // on a repeat-rich genome most candidates belong to reads with hundreds of them): the character is a scalar, the wave branches
    // on it, and the row's match vector is ONE funnel shift of the window's match plane for that letter (made once per 32 rows from
    // the held planes: A = ~lo & ~hi, C = lo & ~hi, G = ~lo & hi; a read T matches window C and T: plane 0 alone) -- 11
    // instructions instead of 16, in a kernel that runs at the VALU's issue rate.
    auto row_eq = [&](u32 eq) {
        if constexpr (!WIDE) {
            constexpr u32 A = 0xF0u, B = 0xCCu, C = 0xAAu;
            u32 X = __builtin_amdgcn_bitop3_b32(eq, bmask, VN, ((A & B) | C) & 0xffu);
            const u32 D0 = __builtin_amdgcn_bitop3_b32(VP + (X & VP), VP, X, ((A ^ B) | C) & 0xffu);
            const u32 HN = VP & D0;
            const u32 HP = __builtin_amdgcn_bitop3_b32(VN, VP, D0, (A | ~(B | C)) & 0xffu);
            X = D0 >> 1;
            VN = X & HP; VP = __builtin_amdgcn_bitop3_b32(HN, X, HP, (A | ~(B | C)) & 0xffu);
            acc = __builtin_amdgcn_alignbit(D0, acc, 1);
        }
    };
    // the 16 characters of rows ib .. ib + 15 as codes (c16) and not-ACGT marks (m16); i0 = ib & ~31
    auto codes = [&](int i0, int half, u32& c16, u32& m16) {
        if constexpr (PACKED) {
            // 32 bases per word; for dirty rows the not-ACGT bit of each
            c16 = (u32)(prow[i0 >> 5] >> (32 * half));
            m16 = dirty ? (u32)((prow[pW + (i0 >> 6)] >> ((i0 & 63) + 16 * half)) & 0xffffull) : 0u;
        } else {
            const uint4 v = *reinterpret_cast<const uint4*>(rd + i0 + 16 * half);       // rows are 16-byte aligned and padded
            const u32 cw[4] = {v.x, v.y, v.z, v.w};
            // the letter a byte would have to be, rebuilt from its bits 1-2 (A 00, C 01, G 11, T 10): 0x41 | bits 1-2, T: ^ 0x11;
            // the code is those two bits with G and T exchanged, four characters gathered into a byte by one multiplication
            c16 = 0; m16 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 x = cw[q];
                const u32 isT = (x >> 2) & ~(x >> 1) & 0x01010101u;
                const u32 bad = x ^ ((0x41414141u | (x & 0x06060606u)) ^ (isT * 0x11u));
                const u32 nz = ((bad | ((bad & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u;
                const u32 b = (x >> 1) & 0x03030303u;
                const u32 cd = b ^ ((b >> 1) & 0x01010101u);
                c16 |= ((cd * 0x01041040u) >> 24) << (8 * q);
                m16 |= ((nz * 0x01020408u) >> 24) << (4 * q);
            }
        }
    };
    // 16 rows from row ib on; win(sh, lo, hi) = the planes at offset sh of the current 32 rows.  true: no lane of the wave can
    // come back under k (Levenshtein_Cal.h:455: such a candidate ends with err = ~0 whether or not it goes on)
    auto rows16 = [&](int ib, int half, u32 c16, u32 m16, auto&& win) -> bool {
        // a wave whose lanes all hold 16 characters of A/C/G/T inside their reads runs the unchecked rows
        const bool plain = m16 == 0 && ib + 16 <= L;
        if (__all(plain)) {
#pragma unroll
            for (int c = 0; c < 16; c++) { W lo, hi; win(16 * half + c, lo, hi); row(lo, hi, c16, 0u, c, ib + c, std::false_type()); }
            err += 16 - __popc(acc >> 16);
        } else {
#pragma unroll
            for (int c = 0; c < 16; c++) { W lo, hi; win(16 * half + c, lo, hi); row(lo, hi, c16, m16, c, ib + c, std::true_type()); }
        }
        return __all(err - last_high > k);
    };
    bool held = false;
    if constexpr (!WIDE) held = __all(!valid || p_len <= 32 * BPM_HELD) != 0;
    if (held) {
        if constexpr (!WIDE) {
            if (!valid) return;
            // the words that hold the window's bases, all requested before the first is used
            const u64* gp = ix.gen2p + (site >> 5);
            const int nw = (int)(((site & 31) + (u64)p_len + 31) >> 5);
            u64 raw[BPM_HELD + 1];
#pragma unroll
            for (int j = 0; j <= BPM_HELD; j++) raw[j] = j < nw ? gp[j] : 0ull;
            u32 pl[BPM_HELD + 1], ph[BPM_HELD + 1];       // the planes of the words as they lie
#pragma unroll
            for (int j = 0; j <= BPM_HELD; j++) { pl[j] = (u32)raw[j]; ph[j] = (u32)(raw[j] >> 32); }
            const u32 s = (u32)(site & 31);
            u32 lo[BPM_HELD], hi[BPM_HELD];               // bit j of word w = plane bit of base site + 32 w + j
#pragma unroll
            for (int j = 0; j < BPM_HELD; j++) { lo[j] = __builtin_amdgcn_alignbit(pl[j + 1], pl[j], s); hi[j] = __builtin_amdgcn_alignbit(ph[j + 1], ph[j], s); }
            for (int i0 = 0; i0 < L; i0 += 32) {
                [[maybe_unused]] u32 m0[4], m1[4];          // the match planes of the four letters over the held pair of words
                if constexpr (PACKED) if (uniform) {
                    constexpr u32 A = 0xF0u, B = 0xCCu;
                    m0[0] = __builtin_amdgcn_bitop3_b32(lo[0], hi[0], 0u, (~A & ~B) & 0xffu); m1[0] = __builtin_amdgcn_bitop3_b32(lo[1], hi[1], 0u, (~A & ~B) & 0xffu);
                    m0[1] = __builtin_amdgcn_bitop3_b32(lo[0], hi[0], 0u, (A & ~B) & 0xffu);  m1[1] = __builtin_amdgcn_bitop3_b32(lo[1], hi[1], 0u, (A & ~B) & 0xffu);
                    m0[2] = __builtin_amdgcn_bitop3_b32(lo[0], hi[0], 0u, (~A & B) & 0xffu);  m1[2] = __builtin_amdgcn_bitop3_b32(lo[1], hi[1], 0u, (~A & B) & 0xffu);
                    m0[3] = lo[0]; m1[3] = lo[1];
                }
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int ib = i0 + 16 * half;
                    if (ib >= L) break;
                    u32 c16, m16;
                    codes(i0, half, c16, m16);
                    if constexpr (PACKED) if (uniform && __all(m16 == 0 && ib + 16 <= L)) {
                        const u32 cs = __builtin_amdgcn_readfirstlane(c16);
#pragma unroll
                        for (int c = 0; c < 16; c++) {
                            const u32 sh = (u32)(16 * half + c);
                            u32 eq;
                            switch ((cs >> (2 * c)) & 3u) {                       // a scalar: the wave branches
                                case 0: eq = __builtin_amdgcn_alignbit(m1[0], m0[0], sh); break;
                                case 1: eq = __builtin_amdgcn_alignbit(m1[1], m0[1], sh); break;
                                case 2: eq = __builtin_amdgcn_alignbit(m1[2], m0[2], sh); break;
                                default: eq = __builtin_amdgcn_alignbit(m1[3], m0[3], sh); break;
                            }
                            row_eq(eq);
                        }
                        err += 16 - __popc(acc >> 16);
                        if (__all(err - last_high > k)) return;
                        continue;
                    }
                    if (rows16(ib, half, c16, m16, [&](int sh, u32& a, u32& b) { a = __builtin_amdgcn_alignbit(lo[1], lo[0], (u32)sh); b = __builtin_amdgcn_alignbit(hi[1], hi[0], (u32)sh); })) return;
                }
#pragma unroll
                for (int j = 0; j + 1 < BPM_HELD; j++) { lo[j] = lo[j + 1]; hi[j] = hi[j + 1]; }
                lo[BPM_HELD - 1] = 0; hi[BPM_HELD - 1] = 0;
            }
        }
    } else {
        if (!valid) return;
        u64 loS, hiS;                                       // bit j = plane bit of base site + i0 + j
        u32 loT = 0, hiT = 0;                               // WIDE: bits 64..95
        {
            u32 l0, h0, l1, h1;
            planes32(ix, site, l0, h0); planes32(ix, site + 32, l1, h1);
            loS = ((u64)l1 << 32) | l0; hiS = ((u64)h1 << 32) | h0;
            if (WIDE) planes32(ix, site + 64, loT, hiT);
        }
        for (int i0 = 0; i0 < L; i0 += 32) {
            if (i0) {
                u32 nl, nh;
                planes32(ix, site + (u64)i0 + (WIDE ? 64 : 32), nl, nh);
                if (WIDE) {
                    loS = (loS >> 32) | ((u64)loT << 32); hiS = (hiS >> 32) | ((u64)hiT << 32);
                    loT = nl; hiT = nh;
                } else { loS = (loS >> 32) | ((u64)nl << 32); hiS = (hiS >> 32) | ((u64)nh << 32); }
            }
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int ib = i0 + 16 * half;
                if (ib >= L) break;
                u32 c16, m16;
                codes(i0, half, c16, m16);
                if (rows16(ib, half, c16, m16, [&](int sh, W& a, W& b) {
                        if (WIDE) {         // (three 32-bit funnel shifts' worth: sh < 32)
                            a = (W)(((u64)__builtin_amdgcn_alignbit(loT, (u32)(loS >> 32), (u32)sh) << 32) | __builtin_amdgcn_alignbit((u32)(loS >> 32), (u32)loS, (u32)sh));
                            b = (W)(((u64)__builtin_amdgcn_alignbit(hiT, (u32)(hiS >> 32), (u32)sh) << 32) | __builtin_amdgcn_alignbit((u32)(hiS >> 32), (u32)hiS, (u32)sh));
                        } else { a = (W)(loS >> sh); b = (W)(hiS >> sh); }
                    })) return;
            }
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
k_filter_pairs(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, u64 n_cand,
               const u32* __restrict__ read_of, const u64* __restrict__ site,
               u32* __restrict__ ferr, int* __restrict__ fend)
{
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_cand) return;
    u32 e; int es;
    const int L = gm.rl(read_of[g]), k = gm.rk(L);
    // pr.base: the read comes from its packed row (bpm_planes<W, true>: what k_filter / k_filter_pe run inside the mapping calls)
    bpm_read(ix, seq, stride, pr, (long)read_of[g], L, k, site[g], e, es);
    ferr[g] = e; fend[g] = es;
}
