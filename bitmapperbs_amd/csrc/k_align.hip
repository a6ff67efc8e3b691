// bitmapperbs_amd/csrc/k_align.hip -- K11-K13: un-gapped recheck, banded affine-gap semi-global alignment with traceback, CIGAR + NM
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
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
