// bitmapperbs_amd/csrc/bmbs_api.hip -- C-ABI (include/bmbs.h) over the gfx950 kernels.
// Owns the device buffers, the stream, the HIP-event profile and the per-ctx counters.  There is no
// CPU fallback anywhere in this library: without a HIP device every entry point fails with
// BMBS_ENODEV.
#include "bmbs_host.h"
#include "bmbs_kernels.hip"



// zero: the buffer is cleared when it is (re)allocated (packed rows: the mask words behind a row's last piece are never written)
int ensure(Lane* c, DevBuf& b, size_t bytes, bool zero)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return BMBS_OK;
    size_t want = bytes + bytes / 8 + 256;
    if (c->kn.arena && want <= ((size_t)256 << 20)) {              // larger ones keep a hipMalloc of their own (freed when they grow)
        // an arena buffer leaves its old region behind when it grows: growing by half bounds what a stream of slowly rising needs strands
        if (b.cap) want = std::min<size_t>(std::max(want, b.cap + b.cap / 2), (size_t)256 << 20);
        void* p = c->arena.alloc(want);
        if (!p) { c->err = "out of device memory (lane arena)"; return BMBS_ENOMEM; }
        if (b.p && !b.arena) (void)hipFree(b.p);
        b.p = p; b.cap = want; b.arena = true;
    } else {
        if (b.p && !b.arena) (void)hipFree(b.p);
        b.p = nullptr; b.cap = 0; b.arena = false;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) { c->err = std::string("hipMalloc: ") + hipGetErrorString(e); b.p = nullptr; return BMBS_ENOMEM; }
        b.cap = want;
    }
    if (zero && hipMemsetAsync(b.p, 0, want, c->stream) != hipSuccess) { c->err = "hipMemsetAsync failed"; return BMBS_ESTATE; }
    return BMBS_OK;
}
void release(DevBuf& b) { if (b.p && !b.arena) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; b.arena = false; }



void prof_begin(Lane* c, const char* name)
{
    std::vector<Prof>& set = c->profset[c->cur_slot];
    int& used = c->prof_used[c->cur_slot];
    if ((size_t)used == set.size()) {
        Prof p; p.name = name; p.used = false;
        (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b);
        set.push_back(p);
    }
    Prof& p = set[(size_t)used];
    p.name = name; p.used = true;
    (void)hipEventRecord(p.a, c->stream);
}
void prof_end(Lane* c) { (void)hipEventRecord(c->profset[c->cur_slot][(size_t)c->prof_used[c->cur_slot]].b, c->stream); c->prof_used[c->cur_slot]++; }
// the events of a call whose stream work has completed -> `last` and the running sums
void prof_collect(Lane* c, int slot)
{
    c->last.clear();
    for (int i = 0; i < c->prof_used[slot]; i++) {
        const Prof& p = c->profset[slot][(size_t)i];
        float t = 0;
        if (hipEventElapsedTime(&t, p.a, p.b) != hipSuccess) t = 0.f;
        c->last.push_back({p.name, (double)t});
        bool found = false;
        for (auto& a : c->acc) if (a.name == p.name || !strcmp(a.name, p.name)) { a.ms += t; found = true; break; }
        if (!found) c->acc.push_back({p.name, (double)t});
    }
    c->acc_calls++;
}

// exclusive scan u32[n] -> u64[n+1], total left in c->totals[slot]; n_dev: a device-side count that bounds n (see k_scan_partial)
// Two launches (tile sums, offsets) by default; BMBS_SCAN_CHAIN=1: one (k_scan_chain).
int scan_u32(Lane* c, const u32* in, u64 n, u64* out, int slot, u32* list, int nz, const u64* n_dev)
{
    const u64 per = (u64)SCAN_BLOCK * SCAN_ITEMS;
    const u64 nb = n ? (n + per - 1) / per : 1;
    if (!c->kn.scan_chain) {
        ENS(c, c->scan_tmp, (nb + 1) * 8);
        u64* bs = c->scan_tmp.as<u64>();
        hipLaunchKernelGGL(k_scan_partial, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, c->stream, in, n, bs, nz, n_dev);
        hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, c->stream, in, n, bs, out, list, nz, n_dev, c->totals.as<u64>() + slot);
        return BMBS_OK;
    }
    // status words: zeroed when the buffer is (re)allocated and when the epoch wraps, so no word of an earlier scan carries this scan's epoch
    const u64 nt = (nb + SCAN_SUB - 1) / SCAN_SUB;                  // a ticket per SCAN_SUB tiles
    { int rc = ensure(c, c->scan_tmp, (nb + 1) * 8, true); if (rc) return rc; }
    { int rc = ensure(c, c->scan_ticket, 64, true); if (rc) return rc; }
    if (++c->scan_epoch >= SCAN_EPOCH_MAX) {
        if (hipMemsetAsync(c->scan_tmp.p, 0, c->scan_tmp.cap, c->stream) != hipSuccess) { c->err = "hipMemsetAsync failed"; return BMBS_ESTATE; }
        c->scan_epoch = 1;
    }
    hipLaunchKernelGGL(k_scan_chain, dim3((unsigned)nt), dim3(SCAN_BLOCK), 0, c->stream, in, n, out, list, nz, n_dev, c->totals.as<u64>() + slot,
                       c->scan_ticket.as<unsigned int>(), c->scan_ticket_base, c->scan_tmp.as<u64>(), c->scan_epoch);
    c->scan_ticket_base += (u32)nt;
    return BMBS_OK;
}

// capacity for a device-side count of a call that does not wait for it: what earlier calls needed per read, plus a margin
inline u64 cap_from(double need_per_read, u64 n_reads, u64 floor_, double scale = 1.0)
{
    const u64 c = (u64)(scale * (need_per_read * 1.25 * (double)n_reads + 64.0 * std::sqrt((double)n_reads + 1.0) + 1024.0));
    return c > floor_ ? c : floor_;
}

// MismatchPenaltyByQuality (ksw.h:148-161): mp_min + (int)(min(Q - q_base, 40) / 40 * (mp_max - mp_min)), every step in IEEE double
// as the reference evaluates it (the product is tabulated once per context: pen_lut)
int mismatch_penalty(const bmbs_params& P, int Q)
{
    const double phred = std::min<double>(Q - P.q_base, 40.0) / 40;
    return (int)(phred * (P.mp_max - P.mp_min)) + P.mp_min;
}

// MAP_Calculation (Schema.cpp:168-405) as data.  The reference's ladder of `if`s compares two ratios -- the runner-up's distance in
// edits over the threshold (rank_error) and the winner's score above the worst admissible score over that range (rank) -- with
// fixed cut points; the tables below are those cut points and the MAPQ each cell returns.  The ratios are formed and compared in
// double exactly as there; the result is tabulated per threshold on the host (prepare_luts) and only looked up on the device.
int map_calculation(const bmbs_params& P, unsigned second_best_diff, unsigned error_threshold, int best_score)
{
    const int unit = std::max(P.gap_open + P.gap_ext, P.mp_max);
    const int worst = -unit * (int)error_threshold;                 // scoreMax of the reference
    const double rank = (double)std::max(best_score - worst, 0) / (double)(-worst);
    const unsigned ediff = second_best_diff > error_threshold ? error_threshold + 1 : second_best_diff;
    if (ediff > error_threshold) {
        // no runner-up within the threshold: the score alone decides
        static const struct { double at; int q; } alone[] = {{0.8, 42}, {0.7, 40}, {0.6, 24}, {0.5, 23}, {0.4, 8}, {0.3, 3}};
        for (const auto& r : alone) if (rank >= r.at) return r.q;
        return 0;
    }
    const double rerr = (double)(int)ediff / (double)error_threshold;
    const bool perfect = best_score == 0;
    // rows: rank_error >= at; `zero` for a perfect winner, otherwise the first column whose rank cut is reached (hi, mid), else lo
    static const struct { double at; int zero; double hi_cut; int hi; double mid_cut; int mid; int lo; } rows[] = {
        {0.9, 39, 0.0, 33, 0.0, 33, 33}, {0.8, 38, 0.0, 27, 0.0, 27, 27}, {0.7, 37, 0.0, 26, 0.0, 26, 26}, {0.6, 36, 0.0, 22, 0.0, 22, 22},
        {0.5, 35, 0.84, 25, 0.68, 16, 5}, {0.4, 34, 0.84, 21, 0.68, 14, 4},
        {0.3, 32, 0.88, 18, 0.67, 15, 3}, {0.2, 31, 0.88, 17, 0.67, 11, 0}, {0.1, 30, 0.88, 12, 0.67, 7, 0},
    };
    for (const auto& r : rows)
        if (rerr >= r.at) return perfect ? r.zero : rank >= r.hi_cut ? r.hi : rank >= r.mid_cut ? r.mid : r.lo;
    // runner-up (almost) as good as the winner
    return ediff == 0 ? (rank >= 0.67 ? 1 : 0) : (rank >= 0.67 ? 6 : 2);
}

int threshold_k(const bmbs_params& P, int L)
{
    u64 k = (u64)(P.e_f * L);            // error_threshold1 = thread_e_f * length (Schema.cpp:24546)
    if (k >= 31) k = 31;
    return (int)k;
}

// CIGAR operations one alignment can have, + slack: the slots per job in the cigar pool.  The DP (ksw.cpp:1850-2045) maximises the
// score inside the band; the alignment with <= k edits that the Myers filter found lies inside the band and scores at least
// -k * max(mp_max, np, gap_open + gap_ext), and every gap run of any alignment costs at least gap_open + gap_ext, so the optimum
// holds at most k * max(...) / (gap_open + gap_ext) gap runs and twice that + 1 operations.  With the default penalties (6, 1,
// 5 + 3) that is the 2k + 8 this code used for every parameter set -- until a random-parameter soak (tools/fuzz_parity.py) ran
// --gap_open 1 with mismatches at 8-9: gaps cheaper than mismatches, alignments with more runs than slots.
int cigar_ops_bound(const bmbs_params& P, int L, int k)
{
    const long goe = (long)P.gap_open + P.gap_ext;
    const long worst = std::max<long>(std::max<long>(P.mp_max, P.np), goe);
    long gaps = (long)L + 2 * k;
    if (goe > 0) gaps = std::min<long>(gaps, (long)k * worst / goe);
    return (int)std::min<long>(2 * gaps + 8, 1 << 20);
}
// n_cigar of a record is 8 bits, 255 marks "more operations than slots" (cannot happen with the bound above)
#define BMBS_MAX_RECORD_OPS 254

// MAP_Calculation tables for every error_threshold 0..62 (k for single-end, k1+k2 for pairs), concatenated:
// table t = (t+2) x (unit*t+1) bytes at mapq_off[t]; plus klut[L] = the threshold of a read of length L.  Built once.
int prepare_luts(Lane* c)
{
    if (c->luts_ready) return BMBS_OK;
    int unit = c->prm.gap_open + c->prm.gap_ext;
    if (unit < c->prm.mp_max) unit = c->prm.mp_max;
    c->mapq_unit = unit;
    std::vector<u32> off(64, 0);
    std::vector<u8> lut;
    for (int k = 0; k <= 62; k++) {
        const int range = unit * k;
        off[k] = (u32)lut.size();
        lut.resize(lut.size() + (size_t)(k + 2) * (range + 1));
        u8* t = lut.data() + off[k];
        for (int ed = 0; ed <= k + 1; ed++)
            for (int sd = 0; sd <= range; sd++)
                t[(size_t)ed * (range + 1) + sd] = (u8)map_calculation(c->prm, (unsigned)ed, (unsigned)k, sd - range);
    }
    std::vector<u8> kl(1002);
    for (int L = 0; L <= 1001; L++) kl[L] = (u8)threshold_k(c->prm, L);
    ENS(c, c->mapq_lut, lut.size()); ENS(c, c->mapq_off, off.size() * 4); ENS(c, c->klut, kl.size());
    HIPCHK(c, hipMemcpyAsync(c->mapq_lut.p, lut.data(), lut.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->mapq_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->klut.p, kl.data(), kl.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->luts_ready = true;
    return BMBS_OK;
}

// geometry of a batch: fixed length (d_len == nullptr) or per-read lengths with L = the longest
ReadGeom geom(Lane* c, int L, const u16* d_len)
{
    ReadGeom g;
    g.len = d_len; g.klut = c->klut.as<u8>(); g.L = L; g.k = threshold_k(c->prm, L);
    return g;
}

int per_read_workspace(Lane* c, u64 n)
{
    ENS(c, c->verdict, n); ENS(c, c->n_seeds, n); ENS(c, c->multi, n); ENS(c, c->mm_site, n * 2);
    ENS(c, c->exit_site, n * 8); ENS(c, c->seeds, n * BMBS_MAX_SEEDS * sizeof(SeedRec));
    ENS(c, c->n_cand, n * 4); ENS(c, c->cand_off, (n + 1) * 8); ENS(c, c->n_votes, n * 4);
    ENS(c, c->best_site, n * 8); ENS(c, c->best_end, n * 4); ENS(c, c->best_err, n * 4); ENS(c, c->sbd, n * 4);
    ENS(c, c->red_status, n); ENS(c, c->job_flag, n * 4); ENS(c, c->job_off, (n + 1) * 8);
    return BMBS_OK;
}

ReadState read_state(Lane* c)
{
    ReadState s;
    s.verdict = c->verdict.as<u8>(); s.n_seeds = c->n_seeds.as<u8>(); s.multi = c->multi.as<u8>();
    s.mm_site = c->mm_site.as<u16>(); s.exit_site = c->exit_site.as<u64>(); s.seeds = c->seeds.as<SeedRec>();
    s.n_cand = c->n_cand.as<u32>(); s.cand_off = c->cand_off.as<u64>(); s.n_votes = c->n_votes.as<u32>();
    s.best_site = c->best_site.as<u64>(); s.best_end = c->best_end.as<int32_t>(); s.best_err = c->best_err.as<u32>();
    s.sbd = c->sbd.as<u32>(); s.red_status = c->red_status.as<u8>(); s.job_flag = c->job_flag.as<u32>();
    s.job_off = c->job_off.as<u64>();
    return s;
}

// trace slots of the register-band DP kernel: one per launched thread (<= 1 M: 16 k waves fill the chip several times over)
inline u64 sw_trace_slots(u64 n_jobs) { const u64 t = (u64)nblk(n_jobs, 64) * 64; return t < (1ull << 20) ? t : (1ull << 20); }

// LDS of the register-band DP kernels: the window words of every lane's alignment(s) (LdsWin)
inline size_t sw_window_lds(const ReadGeom& gm, int jobs_per_lane) { return (size_t)jobs_per_lane * ((gm.L + 2 * gm.k + 62) / 32 + 1) * 64 * 8; }

template <int KB>
void launch_sw(Lane* c, const char* d_seq, const char* d_qual, const char* d_qual2, const ReadGeom& gm, int stride, u64 n_jobs,
               const Jobs& jobs, u32 rev_from, u32* d_cigar_pool, int max_ops, const PackedRows& prw)
{
    const u64 slots = sw_trace_slots(n_jobs);
    const size_t lds = sw_window_lds(gm, 1);
    for (u64 base = 0; base < n_jobs; base += slots) {
        if (!gm.len)
            hipLaunchKernelGGL((gm.k == KB ? k_align_sw<KB, true, true> : k_align_sw<KB, true, false>), dim3((unsigned)(slots / 64)), dim3(64), lds, c->stream, c->ix, c->sp, c->pen_lut.as<int>(), d_seq,
                               d_qual, d_qual2, gm, stride, c->totals.as<u64>() + 2, c->sw_job.as<u32>(), jobs, rev_from, c->trace.as<u64>(),
                               slots, base, d_cigar_pool, max_ops, c->a_start.as<int>(), c->a_end.as<int>(), c->a_nm.as<u32>(),
                               c->a_score.as<int>(), c->a_nops.as<int>(), prw);
        else
            hipLaunchKernelGGL((k_align_sw<KB, false>), dim3((unsigned)(slots / 64)), dim3(64), lds, c->stream, c->ix, c->sp, c->pen_lut.as<int>(), d_seq,
                               d_qual, d_qual2, gm, stride, c->totals.as<u64>() + 2, c->sw_job.as<u32>(), jobs, rev_from, c->trace.as<u64>(),
                               slots, base, d_cigar_pool, max_ops, c->a_start.as<int>(), c->a_end.as<int>(), c->a_nm.as<u32>(),
                               c->a_score.as<int>(), c->a_nops.as<int>(), prw);
    }
}

// the packed form: two jobs per lane, so a launch of `slots` threads covers 2 * slots jobs
template <int KB>
void launch_sw2(Lane* c, const char* d_seq, const char* d_qual, const char* d_qual2, const ReadGeom& gm, int stride, u64 n_jobs,
                const Jobs& jobs, u32 rev_from, u32* d_cigar_pool, int max_ops, const PackedRows& prw)
{
    const u64 slots = sw_trace_slots((n_jobs + 1) / 2);
    auto kern = !prw.base ? k_align_sw2<KB, false, false> : gm.k == KB ? k_align_sw2<KB, true, true> : k_align_sw2<KB, false, true>;
    for (u64 base = 0; base < n_jobs; base += 2 * slots)
        hipLaunchKernelGGL(kern, dim3((unsigned)(slots / 64)), dim3(64), sw_window_lds(gm, 2), c->stream, c->ix, c->sp, c->pen_lut.as<int>(), d_seq,
                           d_qual, d_qual2, gm, stride, c->totals.as<u64>() + 2, c->sw_job.as<u32>(), jobs, rev_from, c->trace.as<u64>(),
                           slots, base, d_cigar_pool, max_ops, c->a_start.as<int>(), c->a_end.as<int>(), c->a_nm.as<u32>(),
                           c->a_score.as<int>(), c->a_nops.as<int>(), prw);
}

// K11-K13 over n_jobs jobs: un-gapped recheck for all, scan-compact the ones that need the DP, run the
// register-band DP kernel instantiated for the smallest KB >= k.  No host round-trip inside.
// n_jobs bounds the job arrays and grids; n_jobs_dev (or nullptr) is where the device keeps the real count; sw_bound = the DP jobs the
// DP launches cover (a call that does not wait for its counts passes capacities here and lets k_guard_count check the last one)
int run_align(Lane* c, const char* d_seq, const char* d_qual, const ReadGeom& gm, int stride, u64 n_jobs, const Jobs& jobs,
              u32 rev_from, u32* d_cigar_pool, int max_ops, const char* d_qual2 = nullptr, const PackedRows* pr = nullptr,
              const u64* n_jobs_dev = nullptr, u64 sw_bound = ~0ull)
{
    if (sw_bound > n_jobs) sw_bound = n_jobs;
    const int L = gm.L, k = gm.k;              // the longest read and the largest threshold size the workspace
    const u64 nj = n_jobs ? n_jobs : 1;
    const u64 nwk = (u64)((2 * k + 1 + 15) / 16);
    ENS(c, c->a_start, nj * 4); ENS(c, c->a_end, nj * 4); ENS(c, c->a_nm, nj * 4); ENS(c, c->a_score, nj * 4); ENS(c, c->a_nops, nj * 4);
    ENS(c, c->need_sw, nj * 4); ENS(c, c->sw_off, (nj + 1) * 8); ENS(c, c->sw_job, nj * 4);
    if (!n_jobs) return BMBS_OK;
    // Three forms of the DP (BMBS_SW=reg2|reg|wave), all with identical results:
    //  reg2 (default for k >= 5; it was k >= 9 until the row loop lost a third of its instructions: k = 6 now 0.69 vs 0.73 ms):
    //        one PAIR of jobs per lane in packed 16-bit arithmetic, band in registers, trace bytes in HBM
    //        (k_align_sw2) -- the DP is VALU-issue bound, so instructions per cell is what counts: 30 per cell against 52.  Needs
    //        one read length per batch and scores that fit 16 bits.  Measured against `reg` on 10 M-pair / 10 M-read batches:
    //        k = 20 (250 bp) 2.63 vs 3.50 ms, k = 12 2.53 vs 2.69 ms, k = 6 1.02 vs 0.83 ms (narrow bands: the per-row work of two
    //        jobs outweighs the packed cells, and half as many waves balance worse) -- hence the threshold;
    //  reg:  one job per lane, 32-bit (k_align_sw) -- batches of mixed lengths, extreme penalties;
    //  wave: one job per 16 / 32 / 64 lanes, a band row per step, trace + CIGAR in LDS, nothing but the ops written to HBM
    //        (north_star (c)); measured 6.4x the instructions per job of `reg` (lanes beyond the band idle, the F prefix scan and
    //        its DPP wait states), so it only pays when there are too few jobs to fill the lanes.
    const int swm = c->kn.sw_form;
    const bool wave_form = swm == 3;
    int maxpen = std::max(std::max(c->prm.mp_max, c->prm.np), c->prm.gap_open + c->prm.gap_ext);
    const bool packed = !wave_form && swm != 1 && (k >= 5 || swm == 2) && !gm.len && c->prm.gap_ext < 256 && (L + 64) * maxpen < 12000 &&
                        c->prm.mp_min >= 0 && c->prm.gap_ext >= 0 && c->prm.gap_open >= 0 && c->prm.np >= 0;
    const bool reg_form = !wave_form;
    if (reg_form) {
        const u64 nwords = packed ? (u64)((2 * k + 1 + 7) / 8) : nwk;
        const u64 slots = packed ? sw_trace_slots((sw_bound + 1) / 2) : sw_trace_slots(sw_bound);
        ENS(c, c->trace, slots * (u64)L * nwords * 8);
    }
    unsigned long long* cnt = c->counters.as<unsigned long long>();
    const PackedRows prw_ = (pr && pr->base) ? *pr : PackedRows{nullptr, nullptr, 0, 0};
    prof_begin(c, "k_align_ungapped");
    if (pr && pr->base)
        hipLaunchKernelGGL(k_align_ungapped_p, dim3(nblk(n_jobs, 256)), dim3(256), 0, c->stream, c->ix, c->sp, c->pen_lut.as<int>(), d_seq, *pr,
                           d_qual, d_qual2, gm, stride, n_jobs, n_jobs_dev, jobs, rev_from, c->a_start.as<int>(), c->a_end.as<int>(), c->a_nm.as<u32>(),
                           c->a_score.as<int>(), c->a_nops.as<int>(), c->need_sw.as<u32>(), cnt);
    else
        hipLaunchKernelGGL(k_align_ungapped, dim3(nblk(n_jobs, 256)), dim3(256), 0, c->stream, c->ix, c->sp, c->pen_lut.as<int>(), d_seq,
                           d_qual, d_qual2, gm, stride, n_jobs, n_jobs_dev, jobs, rev_from, c->a_start.as<int>(), c->a_end.as<int>(), c->a_nm.as<u32>(),
                           c->a_score.as<int>(), c->a_nops.as<int>(), c->need_sw.as<u32>(), cnt);
    prof_end(c);
    prof_begin(c, "scan_sw");
    int rc = scan_u32(c, c->need_sw.as<u32>(), n_jobs, c->sw_off.as<u64>(), 2, c->sw_job.as<u32>(), 0, n_jobs_dev);
    if (rc) return rc;
    prof_end(c);
    if (n_jobs_dev && reg_form && sw_bound < n_jobs)
        hipLaunchKernelGGL(k_guard_count, dim3(1), dim3(1), 0, c->stream, c->totals.as<u64>() + 2, sw_bound, c->flags.as<u32>(), BMBS_FLAG_SW);
    prof_begin(c, "k_align_sw");
    if (!reg_form) {
        // one alignment per 16 / 32 / 64 lanes (band of 2k+1 cells), trace + CIGAR in LDS
        const size_t words = (size_t)sww_lds_words(L, k);
#define SWW_LAUNCH(LANES)                                                                                                             \
        hipLaunchKernelGGL((k_align_sw_wave<LANES>), dim3(nblk(n_jobs, 64 / LANES)), dim3(64), words * 4 * (64 / LANES), c->stream, c->ix, \
                           c->sp, c->pen_lut.as<int>(), d_seq, d_qual, d_qual2, gm, stride, c->totals.as<u64>() + 2, c->sw_job.as<u32>(),   \
                           jobs, rev_from, d_cigar_pool, max_ops, c->a_start.as<int>(), c->a_end.as<int>(), c->a_nm.as<u32>(),          \
                           c->a_score.as<int>(), c->a_nops.as<int>(), prw_)
        if (k <= 7) SWW_LAUNCH(16);
        else if (k <= 15) SWW_LAUNCH(32);
        else SWW_LAUNCH(64);
#undef SWW_LAUNCH
        prof_end(c);
        return BMBS_OK;
    }
    if (packed) {
        if (k <= 2) launch_sw2<2>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        else if (k <= 4) launch_sw2<4>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        else if (k <= 6) launch_sw2<6>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        else if (k <= 8) launch_sw2<8>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        else if (k <= 10) launch_sw2<10>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        else if (k <= 12) launch_sw2<12>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        else if (k <= 16) launch_sw2<16>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        else if (k <= 20) launch_sw2<20>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        else if (k <= 24) launch_sw2<24>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        else launch_sw2<31>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
        prof_end(c);
        return BMBS_OK;
    }
    // the band loop is unrolled for KB: a tighter bound wastes fewer masked cells (k = 6 in a KB = 8 kernel idles 4 of 17)
    if (k <= 2) launch_sw<2>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    else if (k <= 4) launch_sw<4>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    else if (k <= 6) launch_sw<6>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    else if (k <= 8) launch_sw<8>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    else if (k <= 10) launch_sw<10>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    else if (k <= 12) launch_sw<12>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    else if (k <= 16) launch_sw<16>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    else if (k <= 20) launch_sw<20>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    else if (k <= 24) launch_sw<24>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    else launch_sw<31>(c, d_seq, d_qual, d_qual2, gm, stride, sw_bound, jobs, rev_from, d_cigar_pool, max_ops, prw_);
    prof_end(c);
    return BMBS_OK;
}

// K1-K5: the four seeding kernels with their two work-list compactions (no host round-trip)
SeedCarry seed_carry(Lane* c)
{
    SeedCarry sc;
    sc.sp0 = c->sd_sp0.as<u64>(); sc.hits0 = c->sd_hits0.as<u32>(); sc.ml0 = c->sd_ml0.as<u16>(); sc.tm = c->sd_tm.as<u16>();
    sc.seed_id = c->sd_seed_id.as<u8>(); sc.clen = c->sd_clen.as<u32>(); sc.first_ml = c->sd_first_ml.as<u16>();
    sc.flag_c = c->sd_flag_c.as<u32>(); sc.flag_d = c->sd_flag_d.as<u32>(); sc.off_c = c->sd_off_c.as<u64>(); sc.off_d = c->sd_off_d.as<u64>();
    sc.list_c = c->sd_list_c.as<u32>(); sc.list_d = c->sd_list_d.as<u32>();
    return sc;
}

// packed copy of the read rows for the seeding kernels, k_seed_decide and the un-gapped recheck (default); BMBS_ROWS=ascii keeps the
// round-1 forms for A/B runs.  Paired end: the copy is a by-product of k_pe_prepare (+2.5 %); single end: a kernel of its own
// (k_pack_rows, 0.53 ms per 10 M reads) that the consumers win back (+1.8 %: 1463 -> 1489 M reads/s on configs[1]).
bool use_packed_rows(const Lane* c) { return !c->kn.rows_ascii; }

// trigram rank table: three backward extensions per gather pair (bmbs_dev.h: occ3).  4.5 bytes per row; built from the full SA and
// the 2-bit text on a stream of its own, checked against three single steps on a million rows before it is used.
// margin: HBM that has to stay free behind the table (the work buffers of calls still to come)
// -> false: there was no room for it this time (the caller may ask again behind a later call, when other contexts have gone)
bool occ3_build(Occ3Shared& o, u64 margin)
{
    const u64 rows = o.rows, nb = rows / 96 + 2, need = 27 * nb * 16;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (free_b <= need + margin) return false;
    const u64 n_chunks = (nb + OCC3_CHUNK - 1) / OCC3_CHUNK;
    void *t3 = nullptr, *c3 = nullptr, *sums = nullptr;
    hipStream_t st = nullptr;
    bool ok = hipMalloc(&t3, need) == hipSuccess && hipMalloc(&c3, 64 * 8) == hipSuccess && hipMalloc(&sums, 27 * n_chunks * 8 + 64) == hipSuccess &&
              hipStreamCreate(&st) == hipSuccess;
    if (ok) {
        u32* flag = reinterpret_cast<u32*>(c3) + 2 * 60;                       // two spare words behind the 27 entries: overflow, mismatches
        (void)hipMemsetAsync(c3, 0, 64 * 8, st);
        hipLaunchKernelGGL(k_occ3_planes, dim3(8192), dim3(256), 0, st, o.base, rows, nb, reinterpret_cast<uint4*>(t3));
        hipLaunchKernelGGL(k_occ3_chunk_sums, dim3(nblk(27 * n_chunks, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(t3), nb, n_chunks, reinterpret_cast<u64*>(sums));
        hipLaunchKernelGGL(k_occ3_chunk_scan, dim3(1), dim3(64), 0, st, reinterpret_cast<u64*>(sums), n_chunks, flag);
        hipLaunchKernelGGL(k_occ3_apply, dim3(nblk(27 * n_chunks, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(t3), nb, n_chunks, reinterpret_cast<u64*>(sums));
        hipLaunchKernelGGL(k_occ3_c3, dim3(1), dim3(64), 0, st, o.base, reinterpret_cast<u64*>(c3));
        DevIndex probe = o.base;
        probe.occ3 = reinterpret_cast<const uint4*>(t3); probe.c3 = reinterpret_cast<const u64*>(c3); probe.nb3 = nb;
        const u64 n_check = 1u << 20;
        hipLaunchKernelGGL(k_occ3_check, dim3(nblk(n_check, 256)), dim3(256), 0, st, probe, rows, n_check, flag + 1);
        u32 h[2] = {1, 1};
        ok = hipMemcpyAsync(h, flag, 8, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess && h[0] == 0 && h[1] == 0;
        if (!ok && getenv("BMBS_VERBOSE")) fprintf(stderr, "[bmbs] trigram table not used: count overflow %u, differences from three single steps %u\n", h[0], h[1]);
    }
    if (st) (void)hipStreamDestroy(st);
    if (sums) (void)hipFree(sums);
    if (ok) { o.occ3 = t3; o.c3 = c3; o.nb3 = nb; }
    else { if (t3) (void)hipFree(t3); if (c3) (void)hipFree(c3); }
    return true;
}
// called where a lane decides which seeding kernels to launch, and when a settled call has told it how long its chains are: builds the
// table once for all users of the index when this lane would use it, and adopts it when someone has built it
void occ3_want(Lane* c, bool behind_a_call = false)
{
    if (c->ix.occ3 || !c->o3 || c->kn.kgram < 1 || !use_packed_rows(c)) return;
    const bool wants = c->kn.kgram >= 2 || c->lr_chain >= 3.0;
    std::lock_guard<std::mutex> l(c->o3->mu);
    // (behind a settled call the lane's work buffers exist already -- the table needs little room beside itself; in front of a context's
    // first call, BMBS_KGRAM=2, room for them is left)
    // (`tried` stays clear when the build was skipped for lack of room -- at most eight such attempts, so a device that stays full is not
    // asked for its free memory behind every call.  A lane that sizes its work buffers AFTER the table was built can still meet
    // BMBS_ENOMEM on a device this full: bmbs_reserve() in front of the first call takes the buffers first.)
    if (!c->o3->occ3 && wants && !c->o3->tried) {
        (void)hipSetDevice(c->dev);
        const bool attempted = occ3_build(*c->o3, behind_a_call ? (8ull << 30) : (40ull << 30));
        if (attempted || ++c->o3->skipped >= 8) c->o3->tried = true;
    }
    if (c->o3->occ3) { c->ix.occ3 = reinterpret_cast<const uint4*>(c->o3->occ3); c->ix.c3 = reinterpret_cast<const u64*>(c->o3->c3); c->ix.nb3 = c->o3->nb3; }
}

int launch_seeding(Lane* c, const char* d_seq, const ReadGeom& gm, int stride, u64 n, int pe_mode, bool prepacked = false)
{
    ENS(c, c->sd_sp0, n * 8); ENS(c, c->sd_hits0, n * 4); ENS(c, c->sd_ml0, n * 2); ENS(c, c->sd_tm, n * 2); ENS(c, c->sd_seed_id, n);
    ENS(c, c->sd_clen, n * 4); ENS(c, c->sd_first_ml, n * 2); ENS(c, c->sd_flag_c, n * 4); ENS(c, c->sd_flag_d, n * 4);
    ENS(c, c->sd_off_c, (n + 1) * 8); ENS(c, c->sd_off_d, (n + 1) * 8); ENS(c, c->sd_list_c, n * 4); ENS(c, c->sd_list_d, n * 4);
    SeedCarry sc = seed_carry(c);
    ReadState st = read_state(c);
    unsigned long long* cnt = c->counters.as<unsigned long long>();
    const unsigned chunks = nblk(n, SEED_CHUNK);
    const unsigned chunks_min = nblk(n, SEED_CHUNK_MIN);      // grid for the device-sized chunks of the two work lists
    const int target_waves = c->kn.seed_waves;
    // packed copy of the rows (2 bits per base + a not-ACGT bit plane, 64 bytes for 150 bases): what the seeding engine and
    // k_seed_decide read instead of the ASCII rows; BMBS_ROWS=ascii keeps the round-1 forms (A/B runs)
    const bool packed_rows = use_packed_rows(c);
    PackedRows pr = {nullptr, nullptr, 0, 0};
    if (packed_rows) {
        const int pwords = pack_words(gm.L), W = pack_base_words(gm.L);
        if (!prepacked) {
            { int rz_ = ensure(c, c->prow, n * (u64)pwords * 8 + 64, true); if (rz_) return rz_; } ENS(c, c->prow_dirty, n + 64);
            HIPCHK(c, hipMemsetAsync(c->prow_dirty.p, 0, n + 64, c->stream));
            prof_begin(c, "k_pack_rows");
            hipLaunchKernelGGL(k_pack_rows, dim3(nblk(n * (u64)(stride / 16), 256)), dim3(256), 0, c->stream, d_seq, gm, stride, (long)n,
                               c->prow.as<u64>(), pwords, W, c->prow_dirty.as<u32>());
            prof_end(c);
        }
        pr.base = c->prow.as<u64>(); pr.dirty = c->prow_dirty.as<u8>(); pr.pwords = pwords; pr.W = W;
    }
    prof_begin(c, "k_seed_first");
    // three-letter index steps (DevIndex::occ3): their kernels hold more registers (one or two waves per SIMD fewer), which costs a
    // few per cent where chains are short -- so they are used once the context has seen long ones (BMBS_KGRAM=2: always, 0: never)
    occ3_want(c);
    const bool kg = c->ix.occ3 && packed_rows && (c->kn.kgram >= 2 || (c->kn.kgram == 1 && c->lr_chain >= 3.0));
    if (packed_rows && kg) hipLaunchKernelGGL((k_seed_first<true, true>), dim3(chunks), dim3(64), 0, c->stream, c->ix, d_seq, pr, gm, stride, (long)n, sc, cnt);
    else if (packed_rows) hipLaunchKernelGGL(k_seed_first<true>, dim3(chunks), dim3(64), 0, c->stream, c->ix, d_seq, pr, gm, stride, (long)n, sc, cnt);
    else hipLaunchKernelGGL(k_seed_first<false>, dim3(chunks), dim3(64), 0, c->stream, c->ix, d_seq, pr, gm, stride, (long)n, sc, cnt);
    prof_end(c);
    prof_begin(c, "k_seed_decide");
    // rows staged through LDS (each read fetched from HBM exactly once, coalesced) + 8 characters per compare step.  (An
    // 8-lanes-per-read form without staging was measured at 2.07 ms against 1.36 ms; the un-staged and the one-character forms of
    // round 1 were removed in round 4.)
    const bool lds_ok = (size_t)64 * (stride + 8) <= 48 * 1024;
    if (packed_rows)
        hipLaunchKernelGGL(k_seed_decide_p, dim3(nblk(n, 64)), dim3(64), (size_t)64 * (pr.pwords + 1) * 8, c->stream, c->ix, d_seq, pr, gm, stride,
                           (long)n, c->prm.seed_len, pe_mode, st, sc, cnt);
    else if (lds_ok)
        hipLaunchKernelGGL((k_seed_decide<true, true>), dim3(nblk(n, 64)), dim3(64), (size_t)64 * (stride + 8), c->stream, c->ix, d_seq, gm,
                           stride, (long)n, c->prm.seed_len, pe_mode, st, sc, cnt);
    else
        hipLaunchKernelGGL((k_seed_decide<false, true>), dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->ix, d_seq, gm, stride, (long)n,
                           c->prm.seed_len, pe_mode, st, sc, cnt);
    prof_end(c);
    prof_begin(c, "list_second");
    int rc = scan_u32(c, sc.flag_c, n, sc.off_c, 3, sc.list_c);          // scan + list in one pass
    if (rc) return rc;
    prof_end(c);
    prof_begin(c, "k_seed_second");
    if (packed_rows && kg) hipLaunchKernelGGL((k_seed_second<true, true>), dim3(chunks_min), dim3(64), 0, c->stream, c->ix, d_seq, pr, gm, stride, c->totals.as<u64>() + 3, target_waves, pe_mode, st, sc, cnt);
    else if (packed_rows) hipLaunchKernelGGL(k_seed_second<true>, dim3(chunks_min), dim3(64), 0, c->stream, c->ix, d_seq, pr, gm, stride, c->totals.as<u64>() + 3, target_waves, pe_mode, st, sc, cnt);
    else hipLaunchKernelGGL(k_seed_second<false>, dim3(chunks_min), dim3(64), 0, c->stream, c->ix, d_seq, pr, gm, stride, c->totals.as<u64>() + 3, target_waves, pe_mode, st, sc, cnt);
    prof_end(c);
    prof_begin(c, "list_extra");
    rc = scan_u32(c, sc.flag_d, n, sc.off_d, 4, sc.list_d);
    if (rc) return rc;
    prof_end(c);
    prof_begin(c, "k_seed_extra");
    {
        // the rows of a wave's 64 reads staged in LDS (k_seed_extra); rows too long for 64 KB stay in global memory
        const size_t lds = 64 * (size_t)(stride + 16);
        // ... and, measured, only while the index gathers are short: on a GRCh38-size index (wide forms) every gather is an HBM
        // round trip and the 11 KB of LDS per wave cost more occupancy (3.5 instead of 8 waves per SIMD) than the coalesced rows
        // save: 4.20 ms with LDS rows, 3.93 ms without on the configs[2] batch
        const bool wide_ix = c->ix.sa64 != nullptr;
        const int rows_in_lds = lds <= 48 * 1024 && !wide_ix;
        // packed rows: the lane's row in its LDS slot (4.6 KB per wave at 150 bases: no occupancy lost); rows too long for 16 KB: from global memory
        const size_t plds = (size_t)64 * (pr.pwords + 1) * 8;
        if (packed_rows && plds <= 16 * 1024 && kg)
            hipLaunchKernelGGL((k_seed_extra<false, true, true, true>), dim3(chunks_min), dim3(64), plds, c->stream, c->ix, d_seq, pr, gm, stride, c->totals.as<u64>() + 4,
                               target_waves, c->prm.seed_len, pe_mode, st, sc, cnt);
        else if (packed_rows && plds <= 16 * 1024)
            hipLaunchKernelGGL((k_seed_extra<false, true, true>), dim3(chunks_min), dim3(64), plds, c->stream, c->ix, d_seq, pr, gm, stride, c->totals.as<u64>() + 4,
                               target_waves, c->prm.seed_len, pe_mode, st, sc, cnt);
        else if (packed_rows)
            hipLaunchKernelGGL((k_seed_extra<false, true>), dim3(chunks_min), dim3(64), 0, c->stream, c->ix, d_seq, pr, gm, stride, c->totals.as<u64>() + 4,
                               target_waves, c->prm.seed_len, pe_mode, st, sc, cnt);
        else if (rows_in_lds)
            hipLaunchKernelGGL((k_seed_extra<true, false>), dim3(chunks_min), dim3(64), lds, c->stream, c->ix, d_seq, pr, gm, stride, c->totals.as<u64>() + 4,
                               target_waves, c->prm.seed_len, pe_mode, st, sc, cnt);
        else
            hipLaunchKernelGGL((k_seed_extra<false, false>), dim3(chunks_min), dim3(64), 0, c->stream, c->ix, d_seq, pr, gm, stride, c->totals.as<u64>() + 4,
                               target_waves, c->prm.seed_len, pe_mode, st, sc, cnt);
    }
    prof_end(c);
    return BMBS_OK;
}

// the candidate total behind scan_cand: waited for (exact), or bounded by a capacity with a guard behind the scan (k_guard_cand)
int cand_total(Lane* c, const ReadState& st, u64 n, bool exact, u64* tot_out)
{
    if (exact) {
        u64 tot = 0;
        HIPCHK(c, hipMemcpyAsync(&tot, c->totals.as<u64>(), 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *tot_out = tot;
        return BMBS_OK;
    }
    const u64 cap = cap_from(c->lr_cand, n, n, c->kn.cap_scale);
    hipLaunchKernelGGL(k_guard_cand, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->totals.as<u64>(), cap, c->flags.as<u32>(), (long)n, st.verdict,
                       st.n_cand, st.cand_off);
    hipLaunchKernelGGL(k_guard_done, dim3(1), dim3(1), 0, c->stream, c->totals.as<u64>(), c->flags.as<u32>(), BMBS_FLAG_CAND);
    *tot_out = cap;
    return BMBS_OK;
}

// stages K1-K6 + votes; leaves the vote segments in c->votes / c->slot_read
int run_seed_stages(Lane* c, const char* d_seq, const ReadGeom& gm, int stride, u64 n, u64* total_cand, int pe_mode = 0, bool exact = true, bool prepacked = false)
{
    ReadState st = read_state(c);
    {
        int rcs = launch_seeding(c, d_seq, gm, stride, n, pe_mode, prepacked);
        if (rcs) return rcs;
    }
    prof_begin(c, "scan_cand");
    int rc = scan_u32(c, st.n_cand, n, st.cand_off, 0);
    if (rc) return rc;
    prof_end(c);
    u64 tot = 0;
    rc = cand_total(c, st, n, exact, &tot);
    if (rc) return rc;
    *total_cand = tot;
    const u64 t1 = exact ? std::max<u64>(tot, cap_from((double)tot / (double)n, n, n)) : tot;     // buffers as the next call (which does not wait) will want them
    ENS(c, c->cand, t1 * 8); ENS(c, c->votes, t1 * sizeof(bmbs_vote)); ENS(c, c->slot_read, t1 * 4);
    ENS(c, c->ferr, t1 * 4); ENS(c, c->fend, t1 * 4);
    // locate + sort + votes in one kernel per list-size class (the two-kernel form of round 1, k_locate + k_vote, was removed in round 4)
    {
        ENS(c, c->long_flag, n * 4); ENS(c, c->long_off, (n + 1) * 8); ENS(c, c->long_list, n * 4); ENS(c, c->vote_list, n * 4);
        prof_begin(c, "k_vote_fused");
        // the reads that have candidates, compacted (the scan's list mode on n_cand != 0), so that the vote kernel's waves are dense
        HIPCHK(c, hipMemsetAsync(c->long_flag.p, 0, n * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(st.n_votes, 0, n * 4, c->stream));
        {
            int rv = scan_u32(c, st.n_cand, n, c->long_off.as<u64>(), 10, c->vote_list.as<u32>(), 1);
            if (rv) return rv;
        }
        // long reads (up to 25 seeds): lists of 17..32 candidates are the rule, not the repeat case -- they get a kernel of their
        // own (k_vote_mid); its flag and list live in the seeding work-list buffers, free by now
        // ... and on a repeat-rich genome also for short reads (seeds with many hits): taken when the last call left more than 0.5 % of
        // its reads to the long-list kernels (a wave per read)
        const bool use_mid = gm.L / 10 - 1 > VOTE_REG || c->lr_long > 0.005;
        u32* mid_flag = use_mid ? c->sd_flag_c.as<u32>() : nullptr;
        if (use_mid) HIPCHK(c, hipMemsetAsync(mid_flag, 0, n * 4, c->stream));
        hipLaunchKernelGGL(k_vote_fused, dim3(nblk(n, 64)), dim3(64), 0, c->stream, c->ix, (long)n, gm, st, c->cand.as<u64>(),
                           c->votes.as<bmbs_vote>(), c->slot_read.as<u32>(), c->long_flag.as<u32>(), c->totals.as<u64>() + 10,
                           c->vote_list.as<u32>(), mid_flag);
        prof_end(c);
        if (use_mid) {
            prof_begin(c, "k_vote_mid");
            int rm = scan_u32(c, mid_flag, n, c->long_off.as<u64>(), 11, c->sd_list_c.as<u32>());
            if (rm) return rm;
            hipLaunchKernelGGL(k_vote_mid, dim3(nblk(n, 64)), dim3(64), 0, c->stream, c->ix, gm, st, c->totals.as<u64>() + 11, c->sd_list_c.as<u32>(),
                               c->cand.as<u64>(), c->votes.as<bmbs_vote>(), c->slot_read.as<u32>(), c->counters.as<unsigned long long>());
            prof_end(c);
        }
        // reads with more than 16 candidates (repeats): one block per read
        prof_begin(c, "k_vote_long");
        int rl = scan_u32(c, c->long_flag.as<u32>(), n, c->long_off.as<u64>(), 9, c->long_list.as<u32>());
        if (rl) return rl;
        // the wave form (lists of up to 256 candidates) walks the list and hands the longer ones to a list of their own (slot 13)
        ENS(c, c->big_list, n * 4 + 64);
        unsigned long long* big_count = c->totals.as<unsigned long long>() + 13;
        HIPCHK(c, hipMemsetAsync(big_count, 0, 8, c->stream));
        hipLaunchKernelGGL((k_vote_long<VM_CAP, VM_BLOCK, VOTE_REG>), dim3(32768), dim3(VM_BLOCK), 0, c->stream, c->ix, gm, st, c->totals.as<u64>() + 9,
                           c->long_list.as<u32>(), c->cand.as<u64>(), c->votes.as<bmbs_vote>(), c->slot_read.as<u32>(), c->big_list.as<u32>(), big_count, c->counters.as<unsigned long long>());
        prof_end(c);
        prof_begin(c, "k_vote_big");
        // the handed-over lists in two size classes (as k_vote_pe_long): the <= 1024-key form needs 14 KB of LDS instead of 57 KB, so five
        // times as many reads are in flight -- the vote order (std::sort's permutation, one partition pass after the other) is a
        // chain of barriers, not work
        hipLaunchKernelGGL((k_vote_long<1024, 128, VM_CAP>), dim3(8192), dim3(128), 0, c->stream, c->ix, gm, st, c->totals.as<u64>() + 13,
                           c->big_list.as<u32>(), c->cand.as<u64>(), c->votes.as<bmbs_vote>(), c->slot_read.as<u32>(), (u32*)nullptr, (unsigned long long*)nullptr, c->counters.as<unsigned long long>());
        hipLaunchKernelGGL((k_vote_long<2048, 256, 1024>), dim3(4096), dim3(256), 0, c->stream, c->ix, gm, st, c->totals.as<u64>() + 13,
                           c->big_list.as<u32>(), c->cand.as<u64>(), c->votes.as<bmbs_vote>(), c->slot_read.as<u32>(), (u32*)nullptr, (unsigned long long*)nullptr, c->counters.as<unsigned long long>());
        hipLaunchKernelGGL((k_vote_long<VL_CAP, VL_BLOCK, 2048>), dim3(2048), dim3(VL_BLOCK), 0, c->stream, c->ix, gm, st, c->totals.as<u64>() + 13,
                           c->big_list.as<u32>(), c->cand.as<u64>(), c->votes.as<bmbs_vote>(), c->slot_read.as<u32>(), (u32*)nullptr, (unsigned long long*)nullptr, c->counters.as<unsigned long long>());
        prof_end(c);
    }
    return BMBS_OK;
}

// host rows (stride bytes apart) -> device rows padded to a multiple of 16 bytes
int upload_rows(Lane* c, DevBuf& dst, const char* src, int L, int stride, u64 n, int* dstride, hipStream_t st = nullptr)
{
    if (!st) st = c->stream;
    const int ds = (L + 15) / 16 * 16;
    *dstride = ds;
    ENS(c, dst, n * (u64)ds + 64);
    // rows that are already `ds` bytes apart go over as ONE block (the bytes past L are ignored by every kernel); a 2-D copy of
    // 150-byte rows out of a 160-byte pitch ran at half the link rate
    if (n && stride == ds) { HIPCHK(c, hipMemcpyAsync(dst.p, src, n * (u64)ds, hipMemcpyHostToDevice, st)); return BMBS_OK; }
    HIPCHK(c, hipMemsetAsync(dst.p, 0, n * (u64)ds + 64, st));
    if (n) HIPCHK(c, hipMemcpy2DAsync(dst.p, ds, src, stride, L, n, hipMemcpyHostToDevice, st));
    return BMBS_OK;
}


// ================================================================================================
extern "C" void bmbs_default_params(bmbs_params* p)
{
    p->e_f = 0.08; p->mp_max = 6; p->mp_min = 2; p->np = 1; p->gap_open = 5; p->gap_ext = 3; p->q_base = 33;
    p->seed_len = 30; p->min_ins = 0; p->max_ins = 500; p->sensitive = 0; p->ambiguous_out = 0;
}

void lane_destroy(Lane* c);

Lane* lane_create(int device_id, const bmbs_params& prm, const Knobs& kn, bool first)
{
    Lane* c = new Lane();
    c->dev = device_id;
    c->prm = prm; c->kn = kn;
    c->sp.mp_max = c->prm.mp_max; c->sp.mp_min = c->prm.mp_min; c->sp.np = c->prm.np;
    c->sp.gap_open = c->prm.gap_open; c->sp.gap_ext = c->prm.gap_ext; c->sp.q_base = c->prm.q_base;
    c->sp.seed_len = c->prm.seed_len;
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return nullptr; }
    // (the copy streams are created by lane_copy_streams once every lane of the context has its kernel stream: see there)
    (void)hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_k, hipEventDisableTiming);
    int lut[256];
    for (int q = 0; q < 256; q++) lut[q] = mismatch_penalty(c->prm, q);
    const size_t shard_bytes = BMBS_SHARDS * BMBS_SHARD_WORDS * 8;
    if (ensure(c, c->pen_lut, sizeof(lut)) || ensure(c, c->stats, shard_bytes) || ensure(c, c->call_stats, shard_bytes) || ensure(c, c->counters, shard_bytes) ||
        ensure(c, c->totals, 32 * 8) || ensure(c, c->flags, BMBS_FLAG_WORDS * 4) || ensure(c, c->tx_info, 64) ||
        hipHostMalloc((void**)&c->h_info, 128, hipHostMallocPortable) != hipSuccess ||
        hipHostMalloc((void**)&c->h_tot, 8 * 20 * 8, hipHostMallocPortable) != hipSuccess) { lane_destroy(c); return nullptr; }
    memset(c->h_tot, 0, 8 * 20 * 8);
    (void)hipMemcpy(c->pen_lut.p, lut, sizeof(lut), hipMemcpyHostToDevice);
    (void)hipMemset(c->stats.p, 0, shard_bytes);
    (void)hipMemset(c->call_stats.p, 0, shard_bytes);
    (void)hipMemset(c->counters.p, 0, shard_bytes);
    textpath_device_init(c);
    (void)hipMemset(c->totals.p, 0, 32 * 8);
    (void)hipMemset(c->flags.p, 0, BMBS_FLAG_WORDS * 4);
    // diagnostic: per-wave timeline of the kernels that call wavelog_begin/_end (one context at a time; tools/wavelog.py)
    if (first)
    if (const char* wl = getenv("BMBS_WAVELOG")) {
        const unsigned cap = 1u << 21;
        if (*wl && !ensure(c, c->wavelog_buf, (u64)cap * 32) && !ensure(c, c->wavelog_count, 64)) {
            (void)hipMemset(c->wavelog_count.p, 0, 64);
            WaveLogDev d = {c->wavelog_buf.as<unsigned long long>(), c->wavelog_count.as<unsigned int>(), cap};
            if (hipMemcpyToSymbol(HIP_SYMBOL(g_wavelog), &d, sizeof(d)) == hipSuccess) c->wavelog_path = wl;
        }
    }
    return c;
}

// The runtime multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 unless the environment says otherwise),
// each new stream taking the queue with the fewest users, and two streams on one queue run one behind the other.  With a lane's three
// streams created together, the kernel streams of a context's two lanes ended up on ONE queue in some processes and not in others --
// the lanes did not overlap there (round 5 took it for slow hosts; round 6: the same binary 1108 or 1248 M reads/s from one process
// to the next, 1255 every time with GPU_MAX_HW_QUEUES=8; profiles/HISTORY.md).  So: all kernel streams first, then the upload streams,
// then the download streams (a lane's download follows its kernels anyway: sharing a queue with them costs nothing), and bmbs_env_init
// asks for eight queues unless the process has set the variable itself.
void lane_copy_streams(std::vector<Lane*>& lanes)
{
    for (Lane* c : lanes) (void)hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking);
    for (Lane* c : lanes) (void)hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking);
}
// runs when the library is loaded: before the HIP runtime reads its environment, if nothing in the process has touched the GPU yet
__attribute__((constructor)) static void bmbs_env_init() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

void lane_destroy(Lane* c)
{
    if (!c) return;
    (void)hipSetDevice(c->dev);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (!c->wavelog_path.empty()) {
        (void)hipDeviceSynchronize();
        unsigned cnt = 0;
        (void)hipMemcpy(&cnt, c->wavelog_count.p, 4, hipMemcpyDeviceToHost);
        if (cnt > (1u << 21)) cnt = 1u << 21;
        std::vector<unsigned long long> h((size_t)cnt * 4);
        if (cnt) (void)hipMemcpy(h.data(), c->wavelog_buf.p, (size_t)cnt * 32, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(c->wavelog_path.c_str(), "wb")) { fwrite(h.data(), 32, cnt, f); fclose(f); }
        WaveLogDev d = {nullptr, nullptr, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wavelog), &d, sizeof(d));
        release(c->wavelog_buf); release(c->wavelog_count);
    }
    DevBuf* all[] = {&c->occ, &c->hash, &c->sa, &c->gen2, &c->gen2p, &c->chrom_start, &c->t20, &c->pen_lut, &c->mapq_lut, &c->verdict,
                     &c->n_seeds, &c->multi, &c->mm_site, &c->exit_site, &c->seeds, &c->n_cand, &c->cand_off,
                     &c->n_votes, &c->best_site, &c->best_end, &c->best_err, &c->sbd, &c->red_status, &c->job_flag,
                     &c->job_off, &c->scan_tmp, &c->scan_ticket, &c->totals, &c->cand, &c->votes, &c->slot_read, &c->vote_off, &c->votes_dense, &c->dense_read, &c->ferr, &c->fend,
                     &c->job_read, &c->job_site, &c->job_end, &c->job_err, &c->need_sw, &c->sw_off, &c->sw_job, &c->trace, &c->a_start, &c->a_end, &c->a_nm, &c->a_score, &c->a_nops,
                     &c->in_seq, &c->in_qual, &c->out_res, &c->cig_pool, &c->in_a, &c->in_b, &c->in_c, &c->in_d,
                     &c->stats, &c->call_stats, &c->flags, &c->counters, &c->pe_seq, &c->pe_B, &c->pe_occ, &c->pe_len, &c->pe_cur,
                     &c->sd_sp0, &c->sd_hits0, &c->sd_ml0, &c->sd_tm, &c->sd_seed_id, &c->sd_clen, &c->sd_first_ml, &c->sd_flag_c, &c->sd_flag_d,
                     &c->sd_off_c, &c->sd_off_d, &c->sd_list_c, &c->sd_list_d, &c->pe_vround, &c->pe_dead, &c->pe_both, &c->pe_npair, &c->pe_sbd, &c->in_seq2, &c->in_qual2,
                     &c->pe_first, &c->pe_full, &c->pe_R, &c->pe_roff, &c->pe_rflag, &c->pe_rscan, &c->pe_rlist, &c->pe_rcnt, &c->pe_ritem_off, &c->pe_rcand, &c->long_flag, &c->long_off, &c->long_list, &c->vote_list,
                     &c->mapq_off, &c->klut, &c->in_len, &c->fq_text1, &c->fq_text2, &c->fq_idx, &c->pk_in1, &c->pk_in2, &c->pk_ascii, &c->prow, &c->prow_dirty, &c->pe_mid_flag, &c->pe_mid_list};
    for (DevBuf* b : all) release(*b);
    for (auto& set : c->profset) for (auto& p : set) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    c->arena.free_all();
    if (c->h_tot) (void)hipHostFree(c->h_tot);
    if (c->h_info) (void)hipHostFree(c->h_info);
    { DevBuf* tx[] = {&c->tx_tilecnt, &c->tx_tileoff, &c->tx_nl[0], &c->tx_nl[1], &c->tx_rec[0], &c->tx_rec[1], &c->tx_info, &c->sam_len, &c->sam_off, &c->sam_out, &c->chrom_chars, &c->chrom_off, &c->big_list,
                     &c->bam_raw, &c->bam_slots, &c->bam_slot_len, &c->bam_off, &c->stats_snap, &c->z_comp, &c->z_off, &c->z_text, &c->z_err, &c->z_nl, &c->z_comp2, &c->z_off2, &c->z_err2};
      for (DevBuf* b : tx) release(*b); }
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    if (c->ev_k) (void)hipEventDestroy(c->ev_k);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// the lane a stage entry point runs on; a failing call leaves its message in the context

extern "C" bmbs_ctx* bmbs_create(int device_id, const bmbs_params* params)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return nullptr;
    if (hipSetDevice(device_id) != hipSuccess) return nullptr;
    bmbs_ctx* X = new bmbs_ctx();
    X->dev = device_id;
    if (params) X->prm = *params; else bmbs_default_params(&X->prm);
    X->kn.read();                                  // the A/B environment is read here, once; the launch path never calls getenv
    for (int i = 0; i < X->kn.host_lanes; i++) {          // (the first kn.lanes of them take the chunks of the *_device calls)
        Lane* c = lane_create(device_id, X->prm, X->kn, i == 0);
        if (!c) { bmbs_destroy(X); return nullptr; }
        X->lanes.push_back(c);
    }
    lane_copy_streams(X->lanes);
    return X;
}

extern "C" void bmbs_destroy(bmbs_ctx* X)
{
#ifdef VOTE_PROF
    {
        unsigned long long h[4][8];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_vote_prof), sizeof h) == hipSuccess)
            for (int q = 0; q < 4; q++)
                fprintf(stderr, "[vote_prof] class %d: lists %llu candidates %llu sites %llu | Mcycles: locate+sort %.1f run-ends %.1f vote-order %.1f write %.1f | one-lane fallbacks %llu\n",
                        q, h[q][0], h[q][1], h[q][2], h[q][3] / 1e6, h[q][4] / 1e6, h[q][5] / 1e6, h[q][6] / 1e6, h[q][7]);
    }
#endif
    if (!X) return;
    (void)hipSetDevice(X->dev);
    for (size_t i = X->lanes.size(); i-- > 0;) lane_destroy(X->lanes[i]);      // lane 0 (the index owner) last
    delete X;
}

extern "C" int32_t bmbs_max_cigar_ops(const bmbs_params* params, int32_t L)
{
    if (L <= 0 || L > BMBS_MAX_READ) return -1;
    bmbs_params P;
    if (params) P = *params; else bmbs_default_params(&P);
    return cigar_ops_bound(P, L, threshold_k(P, L));
}

extern "C" const char* bmbs_last_error(const bmbs_ctx* X) { return X ? X->err.c_str() : "no context (no HIP device?)"; }

static int lane_index_attach(Lane* c, const bmbs_index_view* v)
{
    if (!c || !v) return BMBS_EINVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    const u64 G = v->ref_len, n = 2 * G, rows = n + 1;
    if (v->sa_length != rows) { c->err = "index view: sa_length != 2*ref_len + 1"; return BMBS_EINVAL; }
    if (rows >= (1ull << 36)) { c->err = "genome too large: rows must fit the 36-bit fields of the 16-mer table"; return BMBS_EINVAL; }
    if (v->n_chrom < 1) { c->err = "index view: n_chrom must be at least 1"; return BMBS_EINVAL; }
    // texts of 2^32 symbols and more (GRCh38) take the wide forms: 64-bit suffix array, Occ counts relative to super-blocks of 2^31
    // symbols whose sums travel in the DevIndex; BMBS_WIDE=1 forces them on a small index and BMBS_SUPER_SHIFT=s (>= 16) makes the
    // super-blocks small enough for such an index to have several (tests)
    const char* wide_env = getenv("BMBS_WIDE");
    const bool wide = rows >= (1ull << 32) || (wide_env && !strcmp(wide_env, "1"));
    SuperSums sup; sup.shift = 0;
    for (int q = 0; q < 4; q++) { sup.T[q] = 0; sup.A[q] = 0; }
    if (wide) {
        const char* ss = getenv("BMBS_SUPER_SHIFT");
        sup.shift = ss ? atoi(ss) : 31;
        if (sup.shift < 16 || sup.shift > 31 || (n >> sup.shift) >= 4) { c->err = "index attach: the text needs more than four Occ super-blocks"; return BMBS_EINVAL; }
        for (u64 S = 0; S < 4 && (S << sup.shift) <= n; S++) {
            const u64 e = ((S << sup.shift) >> 16) * 2;
            if (e + 1 >= v->high_occ_words) break;
            sup.T[S] = v->high_occ[e]; sup.A[S] = v->high_occ[e + 1];
        }
    }
    // upload the reference layouts verbatim, re-pack on the device, drop the originals
    DevBuf t_bwt, t_ho, t_hh, t_hl, t_sa, t_fl, t_pac;
    // the staging copies are released on every path out of this function (also the HIPCHK early returns)
    struct Staging {
        DevBuf* b[7];
        ~Staging() { for (DevBuf* x : b) release(*x); }
    } staging = {{&t_bwt, &t_ho, &t_hh, &t_hl, &t_sa, &t_fl, &t_pac}};
    (void)staging;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        int rc = ensure(c, b, bytes + 64);
        if (rc) return rc;
        if (hipMemsetAsync(b.p, 0, b.cap, c->stream) != hipSuccess) return BMBS_ENODEV;
        if (hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return BMBS_ENODEV;
        return BMBS_OK;
    };
    int rc = 0;
    rc |= up(t_bwt, v->bwt, v->bwt_words * 8);
    rc |= up(t_ho, v->high_occ, v->high_occ_words * 8);
    rc |= up(t_hh, v->hash_hi, v->hash_entries * 4);
    rc |= up(t_hl, v->hash_lo, v->hash_entries);
    rc |= up(t_sa, v->sa, v->sa_entries * 4);
    rc |= up(t_fl, v->sa_flag, v->sa_flag_words * 8);
    rc |= up(t_pac, v->pac, v->pac_bytes);
    if (rc) { c->err = "index upload failed"; return BMBS_ENOMEM; }
    RefIndexDev R;
    R.bwt = t_bwt.as<u64>(); R.high_occ = t_ho.as<u64>(); R.hash_hi = t_hh.as<u32>(); R.hash_lo = t_hl.as<u8>();
    R.sa = t_sa.as<u32>(); R.sa_flag = t_fl.as<u64>(); R.pac = t_pac.as<u8>();
    const u64 n_blk = n / 32 + 2;
    const u64 gen_words = ((n + 63) / 64 + 3) * 2;          // whole 16-byte pieces plus spare ones for the look-ahead loads
    std::vector<u64> cs(v->n_chrom + 1, 0);
    for (int i = 0; i < v->n_chrom; i++) cs[i + 1] = cs[i] + v->chrom_len[i];
    if (ensure(c, c->occ, n_blk * 16) || ensure(c, c->hash, v->hash_entries * 8) || ensure(c, c->sa, rows * (wide ? 8 : 4)) ||
        ensure(c, c->gen2, gen_words * 8) || ensure(c, c->gen2p, gen_words * 8) || ensure(c, c->chrom_start, cs.size() * 8)) return BMBS_ENOMEM;
    HIPCHK(c, hipMemcpyAsync(c->chrom_start.p, cs.data(), cs.size() * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_repack_occ, dim3(nblk(n_blk, 256)), dim3(256), 0, c->stream, R, n, n_blk, sup, c->occ.as<uint4>());
    hipLaunchKernelGGL(k_repack_hash, dim3(nblk(v->hash_entries, 256)), dim3(256), 0, c->stream, R, v->hash_entries, c->hash.as<u64>());
    hipLaunchKernelGGL(k_build_gen2, dim3(nblk(gen_words, 256)), dim3(256), 0, c->stream, R, G, gen_words, c->gen2.as<u64>());
    hipLaunchKernelGGL(k_build_gen2p, dim3(nblk(gen_words, 256)), dim3(256), 0, c->stream, c->gen2.as<u64>(), gen_words, c->gen2p.as<u64>());
    DevIndex ix;
    ix.occ = c->occ.as<uint4>(); ix.hash = c->hash.as<u64>(); ix.gen2 = c->gen2.as<u64>(); ix.gen2p = c->gen2p.as<u64>();
    ix.sa = wide ? nullptr : c->sa.as<u32>(); ix.sa64 = wide ? c->sa.as<u64>() : nullptr;
    ix.sup_shift = sup.shift;
    for (int q = 0; q < 4; q++) { ix.supT[q] = sup.T[q]; ix.supA[q] = sup.A[q]; }
    ix.chrom_start = c->chrom_start.as<u64>(); ix.G = G; ix.total = n; ix.shapline = v->shapline;
    ix.C[0] = v->nacgt[0]; ix.C[1] = v->nacgt[1]; ix.C[2] = v->nacgt[2]; ix.n_chrom = v->n_chrom;
    hipLaunchKernelGGL(k_expand_sa, dim3(nblk(std::min<u64>(rows, 1ull << 30), 256)), dim3(256), 0, c->stream, ix, R, rows, wide ? nullptr : c->sa.as<u32>(),
                       wide ? c->sa.as<u64>() : nullptr);
    // (16 + E)-mer outcome table: E = 5 (3^21 entries, 83.7 GB) for texts of 2^32 symbols and more when the device has the room,
    // else E = 4 (3^20 entries, 27.9 GB), else none; BMBS_T20=0 turns it off, BMBS_TDEPTH=20|21 forces a depth (A/B runs, tests)
    ix.t20 = nullptr; ix.t_e = 4;
    {
        const char* t20_env = getenv("BMBS_T20");
        const char* td_env = getenv("BMBS_TDEPTH");
        const u64 n_keys = v->hash_entries - 1;                 // 3^16
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        int want_e = wide ? 5 : 4;
        if (td_env) want_e = atoi(td_env) == 21 ? 5 : 4;
        if (!(t20_env && !strcmp(t20_env, "0")) && n_keys == 43046721ull) {
            for (int e5 = want_e; e5 >= 4; e5--) {
                const u64 need = n_keys * t20_width(e5) * 8;
                if (free_b > need + (48ull << 30) && ensure(c, c->t20, need) == BMBS_OK) {
                    ix.t_e = e5;
                    hipLaunchKernelGGL(k_build_t20, dim3(nblk(n_keys, 256)), dim3(256), 0, c->stream, ix, n_keys, c->t20.as<u64>());
                    ix.t20 = c->t20.as<u64>();
                    break;
                }
            }
        }
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { c->err = std::string("index re-pack: ") + hipGetErrorString(e); return BMBS_ENODEV; }
    // the trigram rank table is built when it is wanted (occ3_want): BMBS_KGRAM=2 at the first call, 1 (default) once long chains were seen
    ix.occ3 = nullptr; ix.c3 = nullptr; ix.nb3 = 0;
    c->o3 = std::make_shared<Occ3Shared>();
    c->o3->dev = c->dev; c->o3->base = ix; c->o3->rows = rows;
    c->ix = ix; c->rows = rows; c->attached = true;
    return BMBS_OK;
}

// ------------------------------------------------------------------------------------------------
// A mapping call on a lane: call_begin, the launch sequence, call_end.  call_end adds the call's five counters to the lane's
// (unless a guard flag says the call is going to be repeated) and copies the stage counts and the flags into page-locked
// words; lane_settle waits for the lane and reads them.
int call_begin(Lane* c, int slot)
{
    c->cur_slot = slot; c->prof_used[slot] = 0;
    HIPCHK(c, hipMemsetAsync(c->counters.p, 0, BMBS_SHARDS * BMBS_SHARD_WORDS * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->flags.p, 0, BMBS_FLAG_WORDS * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->totals.p, 0, 16 * 8, c->stream));
    return BMBS_OK;
}
#define BMBS_SLOTS 8                // calls a lane may have in flight: one set of page-locked words each
#define BMBS_SLOT_WORDS 20          // 16 totals + the flag words
int call_end(Lane* c, int slot)
{
    u64* h = c->h_tot + (size_t)slot * BMBS_SLOT_WORDS;
    const int words = BMBS_SHARDS * BMBS_SHARD_WORDS;
    hipLaunchKernelGGL(k_stats_commit, dim3(nblk(words, 256)), dim3(256), 0, c->stream, c->flags.as<u32>(), c->call_stats.as<unsigned long long>(),
                       c->stats.as<unsigned long long>(), words);
    hipLaunchKernelGGL(k_call_chain_counts, dim3(1), dim3(64), 0, c->stream, c->counters.as<unsigned long long>(), c->totals.as<u64>());
    HIPCHK(c, hipMemcpyAsync(h, c->totals.p, 16 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h + 16, c->flags.p, BMBS_FLAG_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
    return BMBS_OK;
}

// the caller's packed rows (bmbs_map_*_packed) -> the lane's: prow / prow_dirty, and `ascii` (rows `stride` apart) gets the text of the
// pieces that hold an 'N'.  rc: the rows are reverse-complemented on the way (mate 2).  row0: where these rows sit in the batch
int rows_from_packed(Lane* c, const u64* src, int hw, const ReadGeom& gm, u64 row0, u64 n, bool rc, char* ascii, int stride)
{
    const int pwords = pack_words(gm.L), W = pack_base_words(gm.L);
    if (hw < W + (gm.L + 63) / 64) { c->err = "packed rows: pwords is smaller than a row of this length takes"; return BMBS_EINVAL; }
    const unsigned g = nblk(n * (u64)W, 256);
    if (rc) hipLaunchKernelGGL(k_rows_from_packed<true>, dim3(g), dim3(256), 0, c->stream, src, hw, gm, (long)row0, (long)n, c->prow.as<u64>(), pwords, W, c->prow_dirty.as<u32>(), ascii, stride);
    else hipLaunchKernelGGL(k_rows_from_packed<false>, dim3(g), dim3(256), 0, c->stream, src, hw, gm, (long)row0, (long)n, c->prow.as<u64>(), pwords, W, c->prow_dirty.as<u32>(), ascii, stride);
    return BMBS_OK;
}

// packed_hw > 0: d_seq_ holds the caller's packed rows (packed_hw words apart) instead of ASCII rows
int map_se_dev(Lane* c, uint64_t d_seq_, uint64_t d_qual_, const u16* d_len, int32_t L, int32_t stride,
               int64_t n_reads, uint64_t d_results, uint64_t d_cigar_pool, int64_t cigar_cap, bool exact = true, u32 cigar_base = 0, int slot = 0, int packed_hw = 0)
{
    if (!c) return BMBS_EINVAL;
    if (!c->attached) { c->err = "no index attached"; return BMBS_ESTATE; }
    if (L <= 0 || L > BMBS_MAX_READ || stride < L || n_reads < 0) { c->err = "bad read geometry"; return BMBS_EINVAL; }
    if ((stride & 15) || (d_seq_ & 15) || (d_qual_ & 15)) { c->err = "device read buffers must be 16-byte aligned with a stride that is a multiple of 16"; return BMBS_EINVAL; }
    HIPCHK(c, hipSetDevice(c->dev));
    const char* d_seq = reinterpret_cast<const char*>(d_seq_);
    const char* d_qual = reinterpret_cast<const char*>(d_qual_);
    const u64 n = (u64)n_reads;
    if (n == 0) return BMBS_OK;
    int rc = prepare_luts(c);
    if (rc) return rc;
    const ReadGeom gm = geom(c, L, d_len);
    const int k = gm.k;
    const int max_ops = cigar_ops_bound(c->prm, L, k);
    c->last_max_ops = max_ops;
    if (max_ops > BMBS_MAX_RECORD_OPS) { c->err = "these gap / mismatch penalties allow alignments with more CIGAR operations than a record holds (254)"; return BMBS_EINVAL; }
    rc = per_read_workspace(c, n);
    if (rc) return rc;
    if (packed_hw && !use_packed_rows(c)) { c->err = "packed reads need the packed-row kernels (BMBS_LEGACY is set)"; return BMBS_EINVAL; }
    if (packed_hw && packed_hw < pack_base_words(gm.L) + (gm.L + 63) / 64) { c->err = "packed rows: pwords is smaller than a row of this length takes"; return BMBS_EINVAL; }
    rc = call_begin(c, slot);
    if (rc) return rc;
    if (packed_hw) {
        // the ASCII rows exist only where a piece holds an 'N' (written by k_rows_from_packed into the lane's own buffer)
        ENS(c, c->pk_ascii, n * (u64)stride + 64);
        { int rz_ = ensure(c, c->prow, n * (u64)pack_words(gm.L) * 8 + 64, true); if (rz_) return rz_; } ENS(c, c->prow_dirty, n + 64);
        HIPCHK(c, hipMemsetAsync(c->prow_dirty.p, 0, n + 64, c->stream));
        prof_begin(c, "k_rows_from_packed");
        rc = rows_from_packed(c, reinterpret_cast<const u64*>(d_seq_), packed_hw, gm, 0, n, false, c->pk_ascii.as<char>(), stride);
        if (rc) return rc;
        prof_end(c);
        d_seq = c->pk_ascii.as<char>();
    }
    ReadState st = read_state(c);
    unsigned long long* cnt = c->counters.as<unsigned long long>();
    u64 tot = 0;                     // candidate slots: the count itself (exact) or the capacity the buffers and grids are sized for
    rc = run_seed_stages(c, d_seq, gm, stride, n, &tot, 0, exact, packed_hw > 0);
    if (rc) return rc;
    c->last_total_cand = tot;
    ENS(c, c->vote_off, (n + 1) * 8);
    {
        const u64 t1 = exact ? std::max<u64>(tot, cap_from((double)tot / (double)n, n, n)) : (tot ? tot : 1);
        ENS(c, c->votes_dense, t1 * sizeof(bmbs_vote)); ENS(c, c->dense_read, t1 * 4);
    }
    prof_begin(c, "vote_compact");
    rc = scan_u32(c, st.n_votes, n, c->vote_off.as<u64>(), 5);
    if (rc) return rc;
    if (tot)
        hipLaunchKernelGGL(k_vote_compact, dim3(nblk(tot, 256)), dim3(256), 0, c->stream, tot, exact ? (const u64*)nullptr : c->totals.as<u64>(), st,
                           c->vote_off.as<u64>(), c->slot_read.as<u32>(), c->votes.as<bmbs_vote>(), c->votes_dense.as<bmbs_vote>(), c->dense_read.as<u32>());
    prof_end(c);
    if (tot) {
        prof_begin(c, "k_filter");
        PackedRows prf = {nullptr, nullptr, 0, 0};
        if (use_packed_rows(c)) { prf.base = c->prow.as<u64>(); prf.dirty = c->prow_dirty.as<u8>(); prf.pwords = pack_words(gm.L); prf.W = pack_base_words(gm.L); }
        hipLaunchKernelGGL(k_filter, dim3(nblk(tot, 256)), dim3(256), 0, c->stream, c->ix, d_seq, prf, gm, stride, c->totals.as<u64>() + 5,
                           c->dense_read.as<u32>(), c->votes_dense.as<bmbs_vote>(), c->ferr.as<u32>(), c->fend.as<int>(), cnt);
        prof_end(c);
    }
    prof_begin(c, "k_reduce");
    {
        // over the compacted list of reads with candidates the vote stage built
        const bool listed = true;
        HIPCHK(c, hipMemsetAsync(st.job_flag, 0, n * 4, c->stream));
        HIPCHK(c, hipMemsetAsync(st.red_status, 0, n, c->stream));
        hipLaunchKernelGGL(k_reduce, dim3(nblk(n, 256)), dim3(256), 0, c->stream, (long)n, c->prm.ambiguous_out, st, c->vote_off.as<u64>(),
                           c->votes_dense.as<bmbs_vote>(), c->ferr.as<u32>(), c->fend.as<int>(), c->totals.as<u64>() + 10,
                           listed ? c->vote_list.as<u32>() : (const u32*)nullptr);
    }
    prof_end(c);
    prof_begin(c, "scan_jobs");
    rc = scan_u32(c, st.job_flag, n, st.job_off, 1);
    if (rc) return rc;
    prof_end(c);
    // jobs: the count (exact), or one slot per read with the device-side count bounding every kernel and a guard on the caller's pool
    u64 n_jobs = n, sw_bound = ~0ull;
    const u64* n_jobs_dev = nullptr;
    if (exact) {
        HIPCHK(c, hipMemcpyAsync(&n_jobs, c->totals.as<u64>() + 1, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if ((u64)cigar_cap < n_jobs * (u64)max_ops) { c->err = "cigar pool too small (it takes bmbs_max_cigar_ops() slots per read)"; return BMBS_ENOMEM; }
    } else {
        n_jobs_dev = c->totals.as<u64>() + 1;
        sw_bound = cap_from(c->lr_sw, n, c->kn.cap_scale < 1.0 ? 64 : 16384, c->kn.cap_scale);
        if ((u64)cigar_cap < n * (u64)max_ops) {
            hipLaunchKernelGGL(k_guard_jobs, dim3(nblk(n, 256)), dim3(256), 0, c->stream, n_jobs_dev, (u64)max_ops, (u64)cigar_cap, c->flags.as<u32>(), (long)n, st.job_flag);
            hipLaunchKernelGGL(k_guard_done, dim3(1), dim3(1), 0, c->stream, c->totals.as<u64>() + 1, c->flags.as<u32>(), BMBS_FLAG_CIGAR);
        }
    }
    c->last_n_jobs = n_jobs;
    {
        const u64 nj = n_jobs ? n_jobs : 1;
        ENS(c, c->job_read, nj * 4); ENS(c, c->job_site, nj * 8); ENS(c, c->job_end, nj * 4); ENS(c, c->job_err, nj * 4);
        if (n_jobs) {
            prof_begin(c, "k_job_list");
            hipLaunchKernelGGL(k_job_list, dim3(nblk(n, 256)), dim3(256), 0, c->stream, (long)n, st, c->job_read.as<u32>(),
                               c->job_site.as<u64>(), c->job_end.as<int>(), c->job_err.as<u32>());
            prof_end(c);
        }
        Jobs jobs = {c->job_read.as<u32>(), c->job_site.as<u64>(), c->job_end.as<int>(), c->job_err.as<u32>()};
        PackedRows prw = {nullptr, nullptr, 0, 0};
        if (use_packed_rows(c)) { prw.base = c->prow.as<u64>(); prw.dirty = c->prow_dirty.as<u8>(); prw.pwords = pack_words(gm.L); prw.W = pack_base_words(gm.L); }
        rc = run_align(c, d_seq, d_qual, gm, stride, n_jobs, jobs, 0xffffffffu, reinterpret_cast<u32*>(d_cigar_pool), max_ops, nullptr, &prw, n_jobs_dev, sw_bound);
        if (rc) return rc;
    }
    prof_begin(c, "k_finalize");
    hipLaunchKernelGGL(k_finalize, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->ix, c->sp, c->pen_lut.as<int>(),
                       c->mapq_lut.as<u8>(), c->mapq_off.as<u32>(), c->mapq_unit, d_seq, d_qual, gm, stride, (long)n, st, c->a_start.as<int>(),
                       c->a_end.as<int>(), c->a_nm.as<u32>(), c->a_score.as<int>(), c->a_nops.as<int>(), max_ops, cigar_base,
                       c->prm.ambiguous_out, c->sd_sp0.as<u64>(), c->sd_hits0.as<u32>(),
                       reinterpret_cast<bmbs_result_dev*>(d_results), c->call_stats.as<unsigned long long>());
    prof_end(c);
    return call_end(c, slot);
}

// host lengths -> device (u16 per read)
int upload_lens(Lane* c, const uint16_t* len, u64 n, u64 at, u64 total, int L, hipStream_t st = nullptr)
{
    if (!st) st = c->stream;
    for (u64 i = 0; i < n; i++) if (len[i] == 0 || len[i] > L) { c->err = "read length 0 or longer than L_max"; return BMBS_EINVAL; }
    ENS(c, c->in_len, total * 2 + 16);           // sized for the whole batch up front: growing would drop the first part
    HIPCHK(c, hipMemcpyAsync(c->in_len.as<u16>() + at, len, n * 2, hipMemcpyHostToDevice, st));
    return BMBS_OK;
}


// ------------------------------------------------------------------------------------------------
// paired-end fast mode (Map_Pair_Seq_end_to_end_fast, Schema.cpp:18570)
// d_len: NULL, or u16[2n] = the lengths of the n first mates followed by those of the n second mates
// prepared: c->pe_seq already holds the 2n rows (mate 1, then reverse-complemented mate 2) -- the FASTQ-text entry point writes
// them there straight from the text and k_pe_prepare is not run
int map_pe_dev(Lane* c, uint64_t d_seq1, uint64_t d_qual1, uint64_t d_seq2, uint64_t d_qual2, const u16* d_len,
               int32_t L, int32_t stride, int64_t n_pairs, uint64_t d_results, uint64_t d_cigar_pool, int64_t cigar_cap, bool prepared = false,
               bool exact = true, u32 cigar_base = 0, int slot = 0, int packed_hw = 0)
{
    if (!c) return BMBS_EINVAL;
    if (!c->attached) { c->err = "no index attached"; return BMBS_ESTATE; }
    if (L <= 0 || L > BMBS_MAX_READ || stride < L || n_pairs < 0) { c->err = "bad read geometry"; return BMBS_EINVAL; }
    if ((stride & 15) || ((d_seq1 | d_qual1 | d_seq2 | d_qual2) & 15)) { c->err = "device read buffers must be 16-byte aligned with a stride that is a multiple of 16"; return BMBS_EINVAL; }
    HIPCHK(c, hipSetDevice(c->dev));
    const u64 n = (u64)n_pairs, n2 = 2 * n;
    if (prepared && c->pe_seq.cap < n2 * (u64)stride + 64) { c->err = "internal: prepared rows missing"; return BMBS_ESTATE; }
    if (n == 0) return BMBS_OK;
    int rc = prepare_luts(c);
    if (rc) return rc;
    const ReadGeom gm = geom(c, L, d_len);
    const PeIns pi = {c->prm.min_ins, c->prm.max_ins};
    const int k = gm.k;
    const int max_ops = cigar_ops_bound(c->prm, L, k);
    c->last_max_ops = max_ops;
    if (max_ops > BMBS_MAX_RECORD_OPS) { c->err = "these gap / mismatch penalties allow alignments with more CIGAR operations than a record holds (254)"; return BMBS_EINVAL; }
    rc = per_read_workspace(c, n2);
    if (rc) return rc;
    ENS(c, c->pe_seq, n2 * (u64)stride + 64);
    ENS(c, c->pe_occ, n2 * 4); ENS(c, c->pe_len, n2 * 4); ENS(c, c->pe_cur, n2); ENS(c, c->pe_vround, n2);
    ENS(c, c->pe_dead, n); ENS(c, c->pe_both, n); ENS(c, c->pe_npair, n * 4); ENS(c, c->pe_sbd, n * 4);
    if (packed_hw && !use_packed_rows(c)) { c->err = "packed reads need the packed-row kernels (BMBS_LEGACY is set)"; return BMBS_EINVAL; }
    if (packed_hw && packed_hw < pack_base_words(gm.L) + (gm.L + 63) / 64) { c->err = "packed rows: pwords is smaller than a row of this length takes"; return BMBS_EINVAL; }
    rc = call_begin(c, slot);
    if (rc) return rc;
    char* seq_all = c->pe_seq.as<char>();
    // the qualities are read where the caller put them (qual_row): mate 1 rows in d_qual1, mate 2 rows in d_qual2
    const char* qual_1 = reinterpret_cast<const char*>(d_qual1);
    const char* qual_2 = reinterpret_cast<const char*>(d_qual2);
    bool prepacked = false;
    if (packed_hw) {
        // the caller's packed rows: mate 1 as it is, mate 2 reverse-complemented (bmbs_map_pe_packed); no ASCII rows to read at all
        { int rz_ = ensure(c, c->prow, n2 * (u64)pack_words(gm.L) * 8 + 64, true); if (rz_) return rz_; } ENS(c, c->prow_dirty, n2 + 64);
        HIPCHK(c, hipMemsetAsync(c->prow_dirty.p, 0, n2 + 64, c->stream));
        prof_begin(c, "k_rows_from_packed");
        rc = rows_from_packed(c, reinterpret_cast<const u64*>(d_seq1), packed_hw, gm, 0, n, false, seq_all, stride);
        if (!rc) rc = rows_from_packed(c, reinterpret_cast<const u64*>(d_seq2), packed_hw, gm, n, n, true, seq_all, stride);
        if (rc) return rc;
        prof_end(c);
        prepacked = true;
    } else if (!prepared) {
        u64* prow = nullptr; u32* pdirty = nullptr;
        const int pwords = pack_words(gm.L), W = pack_base_words(gm.L);
        if (use_packed_rows(c)) {
            { int rz_ = ensure(c, c->prow, n2 * (u64)pwords * 8 + 64, true); if (rz_) return rz_; } ENS(c, c->prow_dirty, n2 + 64);
            HIPCHK(c, hipMemsetAsync(c->prow_dirty.p, 0, n2 + 64, c->stream));
            prow = c->prow.as<u64>(); pdirty = c->prow_dirty.as<u32>(); prepacked = true;
        }
        // on packed rows nothing reads the ASCII rows except under a set bit of the mask plane, so only those pieces are written
        // (BMBS_LEGACY=1: no packed rows, the complete ASCII copy)
        const int sparse = prow != nullptr;
        prof_begin(c, "k_pe_prepare");
        const int ppr = stride / 16;
        if (sparse && ppr <= 256) {
            const int rpb = 256 / ppr;
            hipLaunchKernelGGL(k_pe_prepare_p, dim3(std::min<u64>(nblk(n, rpb), 2048)), dim3(256), (size_t)rpb * (ppr + 1) * 16, c->stream,
                               reinterpret_cast<const char*>(d_seq1), reinterpret_cast<const char*>(d_seq2), gm, stride, (long)n, seq_all, prow,
                               pwords, W, pdirty);
        } else
        hipLaunchKernelGGL(k_pe_prepare, dim3(nblk(n * (stride / 16), 256)), dim3(256), 0, c->stream, reinterpret_cast<const char*>(d_seq1),
                           reinterpret_cast<const char*>(d_seq2), gm, stride, (long)n, seq_all, prow, pwords, W, pdirty, sparse);
        prof_end(c);
    }
    ReadState st = read_state(c);
    PeState ps;
    ps.occ = c->pe_occ.as<int>(); ps.len = c->pe_len.as<u32>(); ps.cur = c->pe_cur.as<u8>(); ps.vround = c->pe_vround.as<u8>();
    ps.dead = c->pe_dead.as<u8>(); ps.both = c->pe_both.as<u8>(); ps.npair = c->pe_npair.as<int>(); ps.sbd = c->pe_sbd.as<u32>();
    const bool sensitive = c->prm.sensitive != 0;
    if (sensitive) {
        ENS(c, c->pe_first, n); ENS(c, c->pe_full, n2); ENS(c, c->pe_roff, n2 * 8); ENS(c, c->pe_rflag, n * 4); ENS(c, c->pe_rscan, (n + 1) * 8);
        ENS(c, c->pe_rlist, n * 4); ENS(c, c->pe_rcnt, n * 4); ENS(c, c->pe_ritem_off, (n + 1) * 8);
    }
    ps.first = c->pe_first.as<u8>(); ps.full = c->pe_full.as<u8>(); ps.R = c->pe_R.as<PeCand>(); ps.roff = c->pe_roff.as<u64>();
    c->last_reseeded = 0; c->last_reseed_cand = 0;
    unsigned long long* cnt = c->counters.as<unsigned long long>();
    // seeding of all 2n reads; candidate slots by scan; locate
    rc = launch_seeding(c, seq_all, gm, stride, n2, 1, prepacked);
    if (rc) return rc;
    prof_begin(c, "scan_cand");
    rc = scan_u32(c, st.n_cand, n2, st.cand_off, 0);
    if (rc) return rc;
    prof_end(c);
    u64 tot = 0;                     // candidate slots: the count itself (exact) or the capacity the buffers and grids are sized for
    rc = cand_total(c, st, n2, exact, &tot);
    if (rc) return rc;
    c->last_total_cand = tot;
    const u64 t1 = exact ? std::max<u64>(tot, cap_from((double)tot / (double)n2, n2, n2)) : (tot ? tot : 1);
    ENS(c, c->cand, t1 * 8); ENS(c, c->votes, t1 * sizeof(PeCand)); ENS(c, c->pe_B, t1 * sizeof(PeCand)); ENS(c, c->slot_read, t1 * 4);
    ENS(c, c->dense_read, t1 * 4); ENS(c, c->ferr, t1 * 4);
    PeCand* A = c->votes.as<PeCand>();
    PeCand* B = c->pe_B.as<PeCand>();
    ENS(c, c->long_flag, n2 * 4); ENS(c, c->long_off, (n2 + 1) * 8); ENS(c, c->long_list, n2 * 4);
    // reads of 180 bases and more place up to 25 seeds: lists of 17..32 candidates are the rule there and get a kernel of their own
    // (k_vote_pe_mid; buffers of its own: --sensitive still reads the seeding flags afterwards)
    // (on a repeat-rich genome also for short reads: when the last call left more than 0.5 % of its reads to the long-list kernels)
    const bool use_mid = gm.L / 10 - 1 > VOTE_REG || c->lr_long > 0.005;
    if (use_mid) { ENS(c, c->pe_mid_flag, n2 * 4); ENS(c, c->pe_mid_list, n2 * 4); }
    u32* mid_flag = use_mid ? c->pe_mid_flag.as<u32>() : nullptr;
    if (use_mid) HIPCHK(c, hipMemsetAsync(mid_flag, 0, n2 * 4, c->stream));
    prof_begin(c, "k_vote_pe_fused");
    hipLaunchKernelGGL(k_vote_pe_fused, dim3(nblk(n2, 64)), dim3(64), 0, c->stream, c->ix, (long)n2, gm, st, ps, c->cand.as<u64>(), A,
                       c->slot_read.as<u32>(), c->long_flag.as<u32>(), mid_flag);
    prof_end(c);
    if (use_mid) {
        prof_begin(c, "k_vote_pe_mid");
        rc = scan_u32(c, mid_flag, n2, c->long_off.as<u64>(), 11, c->pe_mid_list.as<u32>());
        if (rc) return rc;
        hipLaunchKernelGGL(k_vote_pe_mid, dim3(nblk(n2, 64)), dim3(64), 0, c->stream, c->ix, gm, st, ps, c->totals.as<u64>() + 11, c->pe_mid_list.as<u32>(), A, cnt);
        prof_end(c);
    }
    prof_begin(c, "k_vote_pe_long");
    // fast mode: located sites without a partner on the mate's finished list are dropped before the sort (k_pe_fast.hip; --sensitive
    // uses the lists differently, Schema.cpp:19953-21459).  BMBS_PREFILTER=0: off (A/B runs, tests)
    const int prefilter = (!sensitive && c->kn.prefilter) ? 1 : 0;
    rc = scan_u32(c, c->long_flag.as<u32>(), n2, c->long_off.as<u64>(), 9, c->long_list.as<u32>());
    if (rc) return rc;
    ENS(c, c->big_list, n2 * 4 + 64);
    unsigned long long* big_count = c->totals.as<unsigned long long>() + 13;
    HIPCHK(c, hipMemsetAsync(big_count, 0, 8, c->stream));
    hipLaunchKernelGGL((k_vote_pe_long<VM_CAP, VM_BLOCK, VOTE_REG>), dim3(32768), dim3(VM_BLOCK), 0, c->stream, c->ix, gm, st, ps,
                       c->totals.as<u64>() + 9, c->long_list.as<u32>(), A, c->big_list.as<u32>(), big_count, c->cand.as<u64>(), cnt, (long)n, pi, prefilter, c->long_flag.as<u32>());
    prof_end(c);
    prof_begin(c, "k_vote_pe_big");
    // the handed-over lists in two size classes: up to 1024 candidates (10 KB of LDS per block: twice the blocks per CU of the
    // 4096-key form; on the GRCh38-like genome 88 % of the handed-over lists), and the rest
    hipLaunchKernelGGL((k_vote_pe_long<1024, 128, VM_CAP>), dim3(8192), dim3(128), 0, c->stream, c->ix, gm, st, ps,
                       c->totals.as<u64>() + 13, c->big_list.as<u32>(), A, (u32*)nullptr, (unsigned long long*)nullptr, c->cand.as<u64>(), cnt, (long)n, pi, prefilter, c->long_flag.as<u32>());
    hipLaunchKernelGGL((k_vote_pe_long<2048, 256, 1024>), dim3(4096), dim3(256), 0, c->stream, c->ix, gm, st, ps,
                       c->totals.as<u64>() + 13, c->big_list.as<u32>(), A, (u32*)nullptr, (unsigned long long*)nullptr, c->cand.as<u64>(), cnt, (long)n, pi, prefilter, c->long_flag.as<u32>());
    hipLaunchKernelGGL((k_vote_pe_long<VL_CAP, VL_BLOCK, 2048>), dim3(2048), dim3(VL_BLOCK), 0, c->stream, c->ix, gm, st, ps,
                       c->totals.as<u64>() + 13, c->big_list.as<u32>(), A, (u32*)nullptr, (unsigned long long*)nullptr, c->cand.as<u64>(), cnt, (long)n, pi, prefilter, c->long_flag.as<u32>());
    prof_end(c);
    // one verification round: dense (read, list index) work list of the mates scheduled in `round`, Myers, compaction
    auto verify_round = [&](int round, u64 cap, const char* name_f, const char* name_c) -> int {
        if (cap) {
            prof_begin(c, name_f);
            u32* wcnt = c->sd_flag_c.as<u32>();
            u64* woff = c->sd_off_c.as<u64>();
            hipLaunchKernelGGL(k_pe_count, dim3(nblk(n2, 256)), dim3(256), 0, c->stream, (long)n, (long)n2, round, ps, wcnt);
            int r_ = scan_u32(c, wcnt, n2, woff, 6);
            if (r_) return r_;
            hipLaunchKernelGGL(k_pe_worklist, dim3(nblk(n2, 256)), dim3(256), 0, c->stream, (long)n2, wcnt, woff, c->dense_read.as<u32>(),
                               c->ferr.as<u32>());
            PackedRows prf = {nullptr, nullptr, 0, 0};
            if (use_packed_rows(c)) { prf.base = c->prow.as<u64>(); prf.dirty = c->prow_dirty.as<u8>(); prf.pwords = pack_words(gm.L); prf.W = pack_base_words(gm.L); }
            hipLaunchKernelGGL(k_filter_pe, dim3(nblk(cap, 256)), dim3(256), 0, c->stream, c->ix, seq_all, prf, gm, stride, st, ps, A, B,
                               c->totals.as<u64>() + 6, c->dense_read.as<u32>(), c->ferr.as<u32>(), cnt);
            prof_end(c);
        }
        prof_begin(c, name_c);
        hipLaunchKernelGGL(k_pe_compact, dim3(nblk(n2, 64)), dim3(64), 0, c->stream, (long)n, (long)n2, gm, round, st, ps, A, B);
        prof_end(c);
        return BMBS_OK;
    };
    if (!sensitive) {
        prof_begin(c, "k_pe_filter_pairs");
        // pairs with long candidate lists (repeats) are left to a second kernel, one wave per pair -- on repeat-rich input only (the last
        // call left more than 0.5 % of its reads to the long-list vote kernels): flag array, scan and the extra launch cost 0.3 ms per
        // 10 M pairs, 4 % of a launch on a repeat-poor genome.  BMBS_PEF_LONG=0: never, =2: always
        u32* pef_flag = (c->kn.pef_long == 2 || (c->kn.pef_long == 1 && c->lr_long > 0.005)) ? c->long_flag.as<u32>() : nullptr;
        if (pef_flag) HIPCHK(c, hipMemsetAsync(pef_flag, 0, n * 4, c->stream));
        hipLaunchKernelGGL(k_pe_filter_pairs, dim3(nblk(n, 64)), dim3(64), 0, c->stream, (long)n, gm, pi, st, ps, A, B, pef_flag, cnt);
        if (pef_flag) {
            rc = scan_u32(c, pef_flag, n, c->long_off.as<u64>(), 12, c->long_list.as<u32>());
            if (rc) return rc;
            hipLaunchKernelGGL(k_pe_filter_pairs_long, dim3((unsigned)std::min<u64>(n, 65536)), dim3(64), 0, c->stream, (long)n, gm, pi, st, ps, c->totals.as<u64>() + 12,
                               c->long_list.as<u32>(), A, B);
        }
        prof_end(c);
        rc = verify_round(1, tot, "k_filter_pe_r1", "k_pe_compact_r1");
        if (rc) return rc;
        prof_begin(c, "k_pe_prune");
        hipLaunchKernelGGL(k_pe_prune, dim3(nblk(n, 64)), dim3(64), 0, c->stream, (long)n, gm, pi, st, ps, A, B);
        prof_end(c);
        rc = verify_round(2, tot, "k_filter_pe_r2", "k_pe_compact_r2");
        if (rc) return rc;
    } else {
        // Map_Pair_Seq_end_to_end: first mate verified in full, second mate filtered by it, rescue by re-seeding
        prof_begin(c, "k_pes_order");
        hipLaunchKernelGGL(k_pes_order, dim3(nblk(n, 256)), dim3(256), 0, c->stream, (long)n, st, seed_carry(c), ps);
        prof_end(c);
        rc = verify_round(1, tot, "k_filter_pe_r1", "k_pe_compact_r1");
        if (rc) return rc;
        prof_begin(c, "k_pes_second");
        hipLaunchKernelGGL(k_pes_second, dim3(nblk(n, 64)), dim3(64), 0, c->stream, (long)n, gm, pi, st, ps, A, B);
        prof_end(c);
        rc = verify_round(2, tot, "k_filter_pe_r2", "k_pe_compact_r2");
        if (rc) return rc;
        prof_begin(c, "k_pes_reseed");
        u32* rflag = c->pe_rflag.as<u32>();
        u32* rlist = c->pe_rlist.as<u32>();
        u32* rcnt = c->pe_rcnt.as<u32>();
        u64* n_reseed = c->totals.as<u64>() + 7;
        hipLaunchKernelGGL(k_pes_reseed_flag, dim3(nblk(n, 256)), dim3(256), 0, c->stream, (long)n, ps, rflag);
        rc = scan_u32(c, rflag, n, c->pe_rscan.as<u64>(), 7, rlist);
        if (rc) return rc;
        HIPCHK(c, hipMemsetAsync(rcnt, 0, n * 4, c->stream));
        {
            PackedRows prs = {nullptr, nullptr, 0, 0};
            if (use_packed_rows(c)) { prs.base = c->prow.as<u64>(); prs.dirty = c->prow_dirty.as<u8>(); prs.pwords = pack_words(gm.L); prs.W = pack_base_words(gm.L); }
            const bool kg = c->ix.occ3 && prs.base && (c->kn.kgram >= 2 || (c->kn.kgram == 1 && c->lr_chain >= 3.0));
            hipLaunchKernelGGL((kg ? k_pes_reseed<true, true> : prs.base ? k_pes_reseed<true> : k_pes_reseed<false>), dim3(nblk(n, 64)), dim3(64), 0, c->stream, c->ix, seq_all, prs, gm, stride,
                               (long)n, n_reseed, rlist, st, ps, rcnt, cnt);
        }
        rc = scan_u32(c, rcnt, n, c->pe_ritem_off.as<u64>(), 8);
        if (rc) return rc;
        prof_end(c);
        u64 rt[2] = {0, 0};          // pairs to re-seed, their candidates: counts (exact) or capacities with a guard
        if (exact) {
            HIPCHK(c, hipMemcpyAsync(rt, c->totals.as<u64>() + 7, 16, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        } else {
            rt[0] = n; rt[1] = cap_from(c->lr_rcand, n2, c->kn.cap_scale < 1.0 ? 64 : 65536, c->kn.cap_scale);
            hipLaunchKernelGGL(k_guard_rcand, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->totals.as<u64>() + 8, rt[1], c->flags.as<u32>(), (long)n, rcnt,
                               c->pe_ritem_off.as<u64>());
            hipLaunchKernelGGL(k_guard_done, dim3(1), dim3(1), 0, c->stream, c->totals.as<u64>() + 8, c->flags.as<u32>(), BMBS_FLAG_RCAND);
        }
        c->last_reseeded = rt[0]; c->last_reseed_cand = rt[1];
        if (rt[0]) {
            const u64 rtot = exact ? std::max<u64>(rt[1], cap_from((double)rt[1] / (double)n2, n2, 65536)) : (rt[1] ? rt[1] : 1);
            ENS(c, c->pe_R, rtot * sizeof(PeCand)); ENS(c, c->pe_rcand, rtot * 8);
            ENS(c, c->dense_read, rtot * 4); ENS(c, c->ferr, rtot * 4);
            ps.R = c->pe_R.as<PeCand>();
            prof_begin(c, "k_pes_vote");
            // mates with more than PESV_LONG candidates are flagged, listed (slot 14) and sorted by a block each
            u32* pv_flag = c->long_flag.as<u32>();                                  // n2 words: free again after the vote stage
            hipLaunchKernelGGL(k_pes_vote, dim3(nblk(rt[0], 64)), dim3(64), 0, c->stream, c->ix, (long)n, gm, pi, n_reseed, rlist,
                               c->pe_ritem_off.as<u64>(), st, ps, c->pe_rcand.as<u64>(), A, B, pv_flag, cnt);
            if (pv_flag) {
                rc = scan_u32(c, pv_flag, rt[0], c->long_off.as<u64>(), 14, c->long_list.as<u32>(), 0, n_reseed);
                if (rc) return rc;
                hipLaunchKernelGGL((k_pes_vote_long<1024, 128, PESV_LONG>), dim3(8192), dim3(128), 0, c->stream, c->ix, (long)n, gm, pi, c->totals.as<u64>() + 14,
                                   c->long_list.as<u32>(), rlist, c->pe_ritem_off.as<u64>(), st, ps, c->pe_rcand.as<u64>(), A, B);
                // (three size classes, as the vote kernels: a re-seeded mate of 1 025 .. 2 048 candidates in the 4 096-key form held 41 KB of
                // LDS -- three lists per CU; r6)
                hipLaunchKernelGGL((k_pes_vote_long<2048, 256, 1024>), dim3(4096), dim3(256), 0, c->stream, c->ix, (long)n, gm, pi, c->totals.as<u64>() + 14,
                                   c->long_list.as<u32>(), rlist, c->pe_ritem_off.as<u64>(), st, ps, c->pe_rcand.as<u64>(), A, B);
                hipLaunchKernelGGL((k_pes_vote_long<VL_CAP, VL_BLOCK, 2048>), dim3(2048), dim3(VL_BLOCK), 0, c->stream, c->ix, (long)n, gm, pi, c->totals.as<u64>() + 14,
                                   c->long_list.as<u32>(), rlist, c->pe_ritem_off.as<u64>(), st, ps, c->pe_rcand.as<u64>(), A, B);
            }
            prof_end(c);
            rc = verify_round(3, rt[1], "k_filter_pe_r3", "k_pe_compact_r3");
            if (rc) return rc;
        }
    }
    prof_begin(c, "k_pe_pair");
    hipLaunchKernelGGL(k_pe_pair, dim3(nblk(n, 64)), dim3(64), 0, c->stream, c->ix, (long)n, gm, pi, c->prm.ambiguous_out, st, ps, A, B);
    prof_end(c);
    prof_begin(c, "scan_jobs");
    rc = scan_u32(c, st.job_flag, n2, st.job_off, 1);
    if (rc) return rc;
    prof_end(c);
    u64 n_jobs = n2, sw_bound = ~0ull;
    const u64* n_jobs_dev = nullptr;
    if (exact) {
        HIPCHK(c, hipMemcpyAsync(&n_jobs, c->totals.as<u64>() + 1, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if ((u64)cigar_cap < n_jobs * (u64)max_ops) { c->err = "cigar pool too small (it takes bmbs_max_cigar_ops() slots per read)"; return BMBS_ENOMEM; }
    } else {
        n_jobs_dev = c->totals.as<u64>() + 1;
        sw_bound = cap_from(c->lr_sw, n2, c->kn.cap_scale < 1.0 ? 64 : 16384, c->kn.cap_scale);
        if ((u64)cigar_cap < n2 * (u64)max_ops) {
            hipLaunchKernelGGL(k_guard_jobs, dim3(nblk(n2, 256)), dim3(256), 0, c->stream, n_jobs_dev, (u64)max_ops, (u64)cigar_cap, c->flags.as<u32>(), (long)n2, st.job_flag);
            hipLaunchKernelGGL(k_guard_done, dim3(1), dim3(1), 0, c->stream, c->totals.as<u64>() + 1, c->flags.as<u32>(), BMBS_FLAG_CIGAR);
        }
    }
    c->last_n_jobs = n_jobs;
    {
        const u64 nj = n_jobs ? n_jobs : 1;
        ENS(c, c->job_read, nj * 4); ENS(c, c->job_site, nj * 8); ENS(c, c->job_end, nj * 4); ENS(c, c->job_err, nj * 4);
        if (n_jobs) {
            prof_begin(c, "k_job_list");
            hipLaunchKernelGGL(k_job_list, dim3(nblk(n2, 256)), dim3(256), 0, c->stream, (long)n2, st, c->job_read.as<u32>(),
                               c->job_site.as<u64>(), c->job_end.as<int>(), c->job_err.as<u32>());
            prof_end(c);
        }
        Jobs jobs = {c->job_read.as<u32>(), c->job_site.as<u64>(), c->job_end.as<int>(), c->job_err.as<u32>()};
        // mate 2 rows (>= n) carry FASTQ-order qualities for a reverse-complemented read: need_reverse_quality = 1
        PackedRows prw = {nullptr, nullptr, 0, 0};
        if (use_packed_rows(c)) { prw.base = c->prow.as<u64>(); prw.dirty = c->prow_dirty.as<u8>(); prw.pwords = pack_words(gm.L); prw.W = pack_base_words(gm.L); }
        rc = run_align(c, seq_all, qual_1, gm, stride, n_jobs, jobs, (u32)n, reinterpret_cast<u32*>(d_cigar_pool), max_ops, qual_2, &prw, n_jobs_dev, sw_bound);
        if (rc) return rc;
    }
    prof_begin(c, "k_finalize_pe");
    hipLaunchKernelGGL(k_finalize_pe, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->ix, c->sp, c->pen_lut.as<int>(), seq_all, qual_1, qual_2, stride,
                       c->mapq_lut.as<u8>(), c->mapq_off.as<u32>(), c->mapq_unit, gm,
                       c->prm.min_ins, c->prm.max_ins, c->prm.ambiguous_out, (long)n, st, ps, c->a_start.as<int>(), c->a_end.as<int>(), c->a_nm.as<u32>(),
                       c->a_score.as<int>(), c->a_nops.as<int>(), max_ops, cigar_base, reinterpret_cast<bmbs_result_dev*>(d_results),
                       c->call_stats.as<unsigned long long>());
    prof_end(c);
    return call_end(c, slot);
}

// ---- calls in flight on a lane ------------------------------------------------------------------------------------------------
int lane_issue(Lane* c, const Pending& P)
{
    return P.pe ? map_pe_dev(c, P.a[0], P.a[1], P.a[2], P.a[3], P.d_len, P.L, P.stride, P.n, P.d_results, P.d_cigar_pool, P.cigar_cap, P.prepared, P.exact,
                             P.cigar_base, P.slot, P.packed_hw)
                : map_se_dev(c, P.a[0], P.a[1], P.d_len, P.L, P.stride, P.n, P.d_results, P.d_cigar_pool, P.cigar_cap, P.exact, P.cigar_base, P.slot, P.packed_hw);
}

// Wait for the lane, then read what its calls left in their page-locked words: the stage counts (they size the next calls) and the
// guard flags.  A call whose counts did not fit its capacities has done no harm (the guards took its work away and its counters
// were not committed): it is issued again with exact sizes.  The caller's buffers of a *_device call have to stay untouched until
// bmbs_sync() for exactly this reason.
int lane_settle(Lane* c)
{
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc_all = BMBS_OK;
    while (!c->inflight.empty()) {
        Pending P = c->inflight.front();
        c->inflight.pop_front();
        for (int attempt = 0; attempt < 2; attempt++) {
            const u64* t = c->h_tot + (size_t)P.slot * BMBS_SLOT_WORDS;
            const u32* f = reinterpret_cast<const u32*>(t + 16);
            if (f[BMBS_FLAG_CIGAR]) { c->err = "cigar pool too small (it takes bmbs_max_cigar_ops() slots per read)"; rc_all = BMBS_ENOMEM; break; }
            if (!(f[BMBS_FLAG_CAND] | f[BMBS_FLAG_SW] | f[BMBS_FLAG_RCAND])) {
                const double nr = (double)(P.pe ? 2 * P.n : P.n);
                if (nr > 0) {
                    c->lr_cand = std::max(c->lr_cand, std::max(1e-9, (double)t[0] / nr));
                    c->lr_sw = std::max(c->lr_sw, (double)t[2] / nr);
                    c->lr_rcand = std::max(c->lr_rcand, (double)t[8] / nr);
                    c->lr_long = ((double)t[9] + (double)t[11]) / nr;             // reads whose candidate lists went to the mid / long kernels
                    if (t[14]) { c->lr_chain = (double)t[15] / (double)t[14]; occ3_want(c, true); }      // (the table is built here, behind the call that showed the chains)
                }
                c->last_total_cand = t[0]; c->last_n_jobs = t[1];
                prof_collect(c, P.slot);
                break;
            }
            if (attempt == 1) { c->err = "internal error: a call issued with exact sizes raised a capacity flag"; rc_all = BMBS_ESTATE; break; }
            c->n_retries++;
            P.exact = true;
            const int rc = lane_issue(c, P);
            if (rc) { rc_all = rc; break; }
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    return rc_all;
}

// P.staged: the call reads lane-owned staging buffers (host entry points, re-laid length arrays), which the next call overwrites
int lane_enqueue(Lane* c, Pending P, bool staged)
{
    if (!c->inflight.empty() && (c->lr_cand == 0 || (int)c->inflight.size() >= BMBS_SLOTS || c->inflight.back().staged || c->inflight.back().prepared || staged)) {
        const int rc = lane_settle(c);
        if (rc) return rc;
    }
    P.exact = c->kn.exact || c->lr_cand == 0;          // nothing learned yet: wait for the counts this once
    P.slot = c->next_slot;
    c->next_slot = (c->next_slot + 1) % BMBS_SLOTS;
    const int rc = lane_issue(c, P);
    if (rc) return rc;
    P.staged = staged;
    c->inflight.push_back(P);
    return BMBS_OK;
}

// what earlier calls needed per read is shared by the lanes of a context
void share_needs(bmbs_ctx* X)
{
    double a = 0, b = 0, d = 0, g = 0;
    double ch = 0;
    for (Lane* c : X->lanes) { a = std::max(a, c->lr_cand); b = std::max(b, c->lr_sw); d = std::max(d, c->lr_rcand); g = std::max(g, c->lr_long); ch = std::max(ch, c->lr_chain); }
    for (Lane* c : X->lanes) { c->lr_cand = a; c->lr_sw = b; c->lr_rcand = d; c->lr_long = g; c->lr_chain = ch; }
}

int settle_all(bmbs_ctx* X)
{
    int rc_all = BMBS_OK;
    for (Lane* c : X->lanes) { const int rc = lane_settle(c); if (rc) { X->err = c->err; rc_all = rc; } }
    return rc_all;
}

// units per chunk of a call: the whole call on lane 0 when it is small (or the caller's CIGAR pool has no room for every chunk's
// worst case), otherwise n / lanes (BMBS_CHUNK overrides) so that every lane gets one chunk per call
// host_copies: the chunks carry their own H2D / D2H copies; smaller ones (n / 8, within 250 k .. 500 k units) leave a shorter tail behind the last upload
// lanes a device call is dealt to.  Measured on one box, lanes overlapping (round 6; M reads/s at 2 / 3 / 4 lanes): uniform-random genome
// 1249 / 1211-vs-1201 / 1179 -- the seeding kernels there sit at the gather ceiling and a fourth lane only adds contention --;
// GRCh38-like pairs 622 / 640 / 646, single-end 412 / 419 / 421, --sensitive 428 / 440 / 450 -- the list kernels wait on their own
// chains and fill with whatever else is resident.  Three it is.  (Four on repeat-rich input, chosen from what
// settled calls had shown, was built and taken out again: a call's chunk boundaries -- and with them the records' cigar_off -- then
// differed between the first and the second call of a context on the same input.)
int device_lanes(const bmbs_ctx* X) { return std::min<int>((int)X->lanes.size(), X->kn.lanes); }
int64_t chunk_units(const bmbs_ctx* X, int64_t n, int64_t cigar_cap, int rpu, int max_ops, bool host_copies = false, int n_lanes = 0)
{
    const int64_t nl = host_copies ? (int64_t)X->lanes.size() : (int64_t)(n_lanes > 0 ? n_lanes : device_lanes(X));
    if (nl < 2 || n < 2 * X->kn.split_min || cigar_cap < n * rpu * (int64_t)max_ops) return n;
    int64_t ch = X->kn.chunk > 0 ? X->kn.chunk : (n + nl - 1) / nl;
    // (a call of 2 M pairs: 126 M reads/s in chunks of 500 k, 133-138 in chunks of 250 k -- the first upload and the last chunk's
    // kernels and download are not hidden behind anything; tools/hostbuf_ab.py)
    if (host_copies && X->kn.chunk <= 0) ch = std::min<int64_t>(ch, std::min<int64_t>(500000, std::max<int64_t>(250000, n / 8)));
    if (ch < X->kn.split_min) ch = X->kn.split_min;
    return ch < n ? ch : n;
}

// *_device entry points: the call is cut into chunks dealt to the lanes in turn; nothing waits (bmbs_sync does)
int dispatch_device(bmbs_ctx* X, bool pe, uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3, const u16* d_len, int32_t L, int32_t stride, int64_t n,
                    uint64_t d_results, uint64_t d_cigar_pool, int64_t cigar_cap)
{
    if (!X || X->lanes.empty()) return BMBS_EINVAL;
    Lane* c0 = X->lanes[0];
    if (L <= 0 || L > BMBS_MAX_READ || stride < L || n < 0) { X->err = "bad read geometry"; return BMBS_EINVAL; }
    if (n == 0) return BMBS_OK;
    const int rpu = pe ? 2 : 1;
    const int max_ops = cigar_ops_bound(X->prm, L, threshold_k(X->prm, L));
    share_needs(X);
    const int n_lanes = device_lanes(X);                  // (fixed for the call: a lane that settles meanwhile may learn a new lr_long)
    const int64_t ch = chunk_units(X, n, cigar_cap, rpu, max_ops, false, n_lanes);
    int li = 0, used = 0;
    for (int64_t off = 0; off < n; off += ch) {
        const int64_t m = std::min(ch, n - off);
        Lane* c = ch == n ? c0 : X->lanes[(size_t)li];
        Pending P;
        P.pe = pe; P.L = L; P.stride = stride; P.n = m;
        const uint64_t ro = (uint64_t)off * (uint64_t)stride;
        P.a[0] = a0 + ro; P.a[1] = a1 + ro; P.a[2] = pe ? a2 + ro : 0; P.a[3] = pe ? a3 + ro : 0;
        P.d_results = d_results + (uint64_t)off * rpu * 32;
        bool staged = false;
        P.d_len = d_len;
        if (d_len && ch != n) {
            if (!pe) P.d_len = d_len + off;
            else {
                // a chunk of pairs wants its lengths as [m first mates][m second mates]: re-laid into the lane's own array
                HIPCHK(c, hipSetDevice(c->dev));
                if (!c->inflight.empty()) { const int rs = lane_settle(c); if (rs) { X->err = c->err; return rs; } }
                ENS(c, c->in_len, (u64)m * 4 + 16);
                HIPCHK(c, hipMemcpyAsync(c->in_len.as<u16>(), d_len + off, (size_t)m * 2, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->in_len.as<u16>() + m, d_len + n + off, (size_t)m * 2, hipMemcpyDeviceToDevice, c->stream));
                P.d_len = c->in_len.as<u16>();
                staged = true;
            }
        }
        if (ch == n) { P.d_cigar_pool = d_cigar_pool; P.cigar_cap = cigar_cap; P.cigar_base = 0; }
        else {
            const uint64_t base = (uint64_t)off * rpu * (uint64_t)max_ops;
            P.d_cigar_pool = d_cigar_pool + base * 4; P.cigar_cap = m * rpu * (int64_t)max_ops; P.cigar_base = (u32)base;
        }
        const int rc = lane_enqueue(c, P, staged);
        if (rc) { X->err = c->err; return rc; }
        used = std::max(used, (ch == n ? 0 : li) + 1);
        li = (li + 1) % n_lanes;
    }
    X->used_lanes = used;
    return BMBS_OK;
}

// host entry points: chunk i is uploaded, mapped and read back on lane i % lanes -- its copies run beside the kernels of the
// chunks on the other lanes (one direction of the link each: 48 GB/s both ways at once on the MI355X boxes, tools/pcie_probe)
struct HostIn { const char *seq1, *qual1, *seq2, *qual2; const uint16_t *len1, *len2; const uint64_t *rows1 = nullptr, *rows2 = nullptr; int hw = 0; };
int dispatch_host(bmbs_ctx* X, bool pe, const HostIn& in, int32_t L, int32_t stride, int64_t n, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap,
                  int64_t* n_cigar_used)
{
    static_assert(sizeof(bmbs_result) == 32 && sizeof(bmbs_result_dev) == 32, "result record is 32 bytes");
    if (!X || X->lanes.empty()) return BMBS_EINVAL;
    if (n_cigar_used) *n_cigar_used = 0;
    if (n <= 0) return BMBS_OK;
    if (L <= 0 || L > BMBS_MAX_READ || stride < L) { X->err = "bad read geometry"; return BMBS_EINVAL; }
    const int rpu = pe ? 2 : 1;
    const int max_ops = cigar_ops_bound(X->prm, L, threshold_k(X->prm, L));
    const int64_t ch = chunk_units(X, n, cigar_cap, rpu, max_ops, true);
    const int nl = (int)X->lanes.size();
    const int used_lanes_ = ch == n ? 1 : (int)std::min<int64_t>(nl, (n + ch - 1) / ch);
    share_needs(X);
    struct Open { bool on = false; int64_t off = 0, m = 0; };
    std::vector<Open> open((size_t)nl);
    std::atomic<int64_t> extent(0);
    auto finish = [&](int li) -> int {            // the chunk in flight on lane li: counts, then exactly the CIGAR operations it produced
        Open& o = open[(size_t)li];
        if (!o.on) return BMBS_OK;
        o.on = false;
        Lane* c = X->lanes[(size_t)li];
        const u64 retries0 = c->n_retries;
        int rc = lane_settle(c);
        if (rc) return rc;
        const bool cs = c->kn.copy_streams && c->up_stream && c->down_stream && c->ev_up && c->ev_k;
        hipStream_t dsn = cs ? c->down_stream : c->stream;
        if (c->n_retries != retries0)               // the chunk was issued again with exact sizes: the records copied behind the first attempt are stale
            HIPCHK(c, hipMemcpyAsync(results + o.off * rpu, c->out_res.p, (u64)o.m * rpu * 32, hipMemcpyDeviceToHost, dsn));
        const u64 used = c->last_n_jobs * (u64)max_ops;
        const u64 base = ch == n ? 0 : (u64)o.off * rpu * (u64)max_ops;
        if (base + used > (u64)cigar_cap) { c->err = "host cigar pool too small"; return BMBS_ENOMEM; }
        if (used) {
            HIPCHK(c, hipMemcpyAsync(cigar_pool + base, c->cig_pool.p, used * 4, hipMemcpyDeviceToHost, dsn));
            int64_t seen = extent.load();
            while ((int64_t)(base + used) > seen && !extent.compare_exchange_weak(seen, (int64_t)(base + used))) {}
        }
        HIPCHK(c, hipStreamSynchronize(dsn));
        return BMBS_OK;
    };
    // upload, kernels and the records' download of chunk [off, off + m) on lane lane_i (its previous chunk has been finished)
    auto issue = [&](int lane_i, int64_t off, int64_t m) -> int {
        Lane* c = X->lanes[(size_t)lane_i];
            HIPCHK(c, hipSetDevice(c->dev));
            const u64 um = (u64)m;
            ENS(c, c->out_res, um * rpu * 32);
            const u64 pool = um * rpu * (u64)max_ops;                     // worst case: every read needs K12
            ENS(c, c->cig_pool, pool * 4);
            int ds = 0;
            const size_t ro = (size_t)off * (size_t)stride;
            const bool cs = c->kn.copy_streams && c->up_stream && c->down_stream && c->ev_up && c->ev_k;
            hipStream_t us = cs ? c->up_stream : c->stream, dsn = cs ? c->down_stream : c->stream;
            int r1 = BMBS_OK;
            // One upload at a time per device (the text calls' rule, bmbs_textpath.hip): four lanes that upload side by side share the
            // link, finish together, compute together and leave the link idle meanwhile -- in turn, each chunk goes up at the full
            // rate and its kernels run beside the next lane's upload.  The lock is held until this chunk's copies have ended; the
            // kernels are queued behind them meanwhile.  BMBS_UP_TURNS=0: the round-5 form.  (ONE upload stream shared by the lanes --
            // copies back to back with no host in between -- measured no better than side-by-side uploads: 172 against 174 / 165 M
            // reads/s, turns 186 / 192; 2 M pairs per call, profiles/HISTORY.md)
            std::unique_lock<std::mutex> up_turn(g_h2d_mu[c->dev & 15], std::defer_lock);
            if (cs && used_lanes_ > 1 && c->kn.up_turns) up_turn.lock();
            if (in.hw) {
                // packed rows: hw words per read go over as they are (one block per mate)
                const u64 pb = um * (u64)in.hw * 8;
                ENS(c, c->pk_in1, pb + 64);
                HIPCHK(c, hipMemcpyAsync(c->pk_in1.p, in.rows1 + (size_t)off * (size_t)in.hw, pb, hipMemcpyHostToDevice, us));
                if (pe) { ENS(c, c->pk_in2, pb + 64); HIPCHK(c, hipMemcpyAsync(c->pk_in2.p, in.rows2 + (size_t)off * (size_t)in.hw, pb, hipMemcpyHostToDevice, us)); }
            } else { r1 = upload_rows(c, c->in_seq, in.seq1 + ro, L, stride, um, &ds, us); if (r1) return r1; }
            r1 = upload_rows(c, c->in_qual, in.qual1 + ro, L, stride, um, &ds, us); if (r1) return r1;
            if (pe) {
                if (!in.hw) { r1 = upload_rows(c, c->in_seq2, in.seq2 + ro, L, stride, um, &ds, us); if (r1) return r1; }
                r1 = upload_rows(c, c->in_qual2, in.qual2 + ro, L, stride, um, &ds, us); if (r1) return r1;
            }
            if (in.len1) {
                r1 = upload_lens(c, in.len1 + off, um, 0, um * rpu, L, us); if (r1) return r1;
                if (pe) { r1 = upload_lens(c, in.len2 + off, um, um, um * rpu, L, us); if (r1) return r1; }
            }
            if (cs) { HIPCHK(c, hipEventRecord(c->ev_up, us)); HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_up, 0)); }
            Pending P;
            P.pe = pe; P.L = L; P.stride = ds; P.n = m;
            P.a[0] = (uint64_t)c->in_seq.p; P.a[1] = (uint64_t)c->in_qual.p; P.a[2] = pe ? (uint64_t)c->in_seq2.p : 0; P.a[3] = pe ? (uint64_t)c->in_qual2.p : 0;
            if (in.hw) { P.packed_hw = in.hw; P.a[0] = (uint64_t)c->pk_in1.p; P.a[2] = pe ? (uint64_t)c->pk_in2.p : 0; }
            P.d_len = in.len1 ? c->in_len.as<u16>() : nullptr;
            P.d_results = (uint64_t)c->out_res.p; P.d_cigar_pool = (uint64_t)c->cig_pool.p; P.cigar_cap = (int64_t)pool;
            P.cigar_base = ch == n ? 0u : (u32)((u64)off * rpu * (u64)max_ops);
            r1 = lane_enqueue(c, P, true);
            if (r1) return r1;
            if (cs) { HIPCHK(c, hipEventRecord(c->ev_k, c->stream)); HIPCHK(c, hipStreamWaitEvent(dsn, c->ev_k, 0)); }
            HIPCHK(c, hipMemcpyAsync(results + off * rpu, c->out_res.p, um * rpu * 32, hipMemcpyDeviceToHost, dsn));
            if (up_turn.owns_lock()) HIPCHK(c, hipEventSynchronize(c->ev_up));          // this chunk is up: the next lane's turn
            return BMBS_OK;
    };
    // One host thread per lane: a chunk is ~110 kernel launches and a handful of waits, and with all of them issued by one thread the
    // call was bound by that thread on boxes with slower cores (2 M pairs through bmbs_map_pe_packed: 137 M reads/s where the link
    // would give 200).  Chunk k goes to lane k % lanes whoever issues it: the results do not depend on the threads.
    const int64_t n_chunks = (n + ch - 1) / ch;
    const int used_lanes = ch == n ? 1 : (int)std::min<int64_t>(nl, n_chunks);
    int rc = BMBS_OK;
    std::mutex err_mu;
    std::atomic<int> stop(0);
    auto worker = [&](int lane_i) {
        Lane* c = X->lanes[(size_t)lane_i];
        (void)hipSetDevice(c->dev);
        int r = BMBS_OK;
        for (int64_t k = lane_i; k < n_chunks && !r && !stop.load(); k += used_lanes) {
            const int64_t off = k * ch, m = std::min(ch, n - off);
            r = finish(lane_i);
            if (!r) r = issue(lane_i, off, m);
            if (!r) { open[(size_t)lane_i].on = true; open[(size_t)lane_i].off = off; open[(size_t)lane_i].m = m; }
        }
        if (!r) r = finish(lane_i);
        if (r) { stop.store(1); std::lock_guard<std::mutex> l(err_mu); if (!rc) { rc = r; X->err = c->err; } }
    };
    if (used_lanes == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int t = 1; t < used_lanes; t++) th.emplace_back(worker, t);
        worker(0);
        for (auto& t : th) t.join();
    }
    if (rc) { for (Lane* c : X->lanes) { (void)hipStreamSynchronize(c->stream); if (c->down_stream) (void)hipStreamSynchronize(c->down_stream); c->inflight.clear(); } return rc; }
    X->used_lanes = used_lanes;
    if (n_cigar_used) *n_cigar_used = extent.load();
    return BMBS_OK;
}

extern "C" int bmbs_sync(bmbs_ctx* X)
{
    if (!X) return BMBS_EINVAL;
    return settle_all(X);
}

extern "C" int bmbs_map_se_device(bmbs_ctx* X, uint64_t d_seq, uint64_t d_qual, int32_t L, int32_t stride,
                                  int64_t n_reads, uint64_t d_results, uint64_t d_cigar_pool, int64_t cigar_cap)
{
    return dispatch_device(X, false, d_seq, d_qual, 0, 0, nullptr, L, stride, n_reads, d_results, d_cigar_pool, cigar_cap);
}
extern "C" int bmbs_map_se_var_device(bmbs_ctx* X, uint64_t d_seq, uint64_t d_qual, uint64_t d_len, int32_t L_max, int32_t stride,
                                      int64_t n_reads, uint64_t d_results, uint64_t d_cigar_pool, int64_t cigar_cap)
{
    if (X && !d_len) { X->err = "d_len is NULL"; return BMBS_EINVAL; }
    return dispatch_device(X, false, d_seq, d_qual, 0, 0, reinterpret_cast<const u16*>(d_len), L_max, stride, n_reads, d_results, d_cigar_pool, cigar_cap);
}
extern "C" int bmbs_map_se(bmbs_ctx* X, const char* seq, const char* qual, int32_t L, int32_t stride, int64_t n_reads,
                           bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used)
{
    const HostIn in = {seq, qual, nullptr, nullptr, nullptr, nullptr};
    return dispatch_host(X, false, in, L, stride, n_reads, results, cigar_pool, cigar_cap, n_cigar_used);
}
extern "C" int bmbs_map_se_var(bmbs_ctx* X, const char* seq, const char* qual, const uint16_t* len, int32_t L_max, int32_t stride,
                               int64_t n_reads, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used)
{
    if (X && !len) { X->err = "len is NULL"; return BMBS_EINVAL; }
    const HostIn in = {seq, qual, nullptr, nullptr, len, nullptr};
    return dispatch_host(X, false, in, L_max, stride, n_reads, results, cigar_pool, cigar_cap, n_cigar_used);
}
extern "C" int bmbs_map_pe_device(bmbs_ctx* X, uint64_t d_seq1, uint64_t d_qual1, uint64_t d_seq2, uint64_t d_qual2,
                                  int32_t L, int32_t stride, int64_t n_pairs, uint64_t d_results, uint64_t d_cigar_pool,
                                  int64_t cigar_cap)
{
    return dispatch_device(X, true, d_seq1, d_qual1, d_seq2, d_qual2, nullptr, L, stride, n_pairs, d_results, d_cigar_pool, cigar_cap);
}
extern "C" int bmbs_map_pe_var_device(bmbs_ctx* X, uint64_t d_seq1, uint64_t d_qual1, uint64_t d_seq2, uint64_t d_qual2, uint64_t d_len,
                                      int32_t L_max, int32_t stride, int64_t n_pairs, uint64_t d_results, uint64_t d_cigar_pool,
                                      int64_t cigar_cap)
{
    if (X && !d_len) { X->err = "d_len is NULL"; return BMBS_EINVAL; }
    return dispatch_device(X, true, d_seq1, d_qual1, d_seq2, d_qual2, reinterpret_cast<const u16*>(d_len), L_max, stride, n_pairs, d_results,
                           d_cigar_pool, cigar_cap);
}
extern "C" int bmbs_map_pe(bmbs_ctx* X, const char* seq1, const char* qual1, const char* seq2, const char* qual2, int32_t L,
                           int32_t stride, int64_t n_pairs, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap,
                           int64_t* n_cigar_used)
{
    const HostIn in = {seq1, qual1, seq2, qual2, nullptr, nullptr};
    return dispatch_host(X, true, in, L, stride, n_pairs, results, cigar_pool, cigar_cap, n_cigar_used);
}
extern "C" int bmbs_map_pe_var(bmbs_ctx* X, const char* seq1, const char* qual1, const char* seq2, const char* qual2,
                               const uint16_t* len1, const uint16_t* len2, int32_t L_max, int32_t stride, int64_t n_pairs,
                               bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used)
{
    if (X && (!len1 || !len2)) { X->err = "len1 / len2 is NULL"; return BMBS_EINVAL; }
    const HostIn in = {seq1, qual1, seq2, qual2, len1, len2};
    return dispatch_host(X, true, in, L_max, stride, n_pairs, results, cigar_pool, cigar_cap, n_cigar_used);
}

// ---- packed reads (include/bmbs.h): 2 bits per base + an 'N' plane instead of a byte per base ---------------------------------------------
extern "C" int bmbs_map_se_packed(bmbs_ctx* X, const uint64_t* rows, int32_t pwords, const char* qual, const uint16_t* len, int32_t L_max, int32_t stride,
                                  int64_t n_reads, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used)
{
    if (X && (!rows || !qual || pwords <= 0)) { X->err = "packed reads: NULL rows / qualities or pwords <= 0"; return BMBS_EINVAL; }
    HostIn in = {nullptr, qual, nullptr, nullptr, len, nullptr};
    in.rows1 = rows; in.hw = pwords;
    return dispatch_host(X, false, in, L_max, stride, n_reads, results, cigar_pool, cigar_cap, n_cigar_used);
}
extern "C" int bmbs_map_pe_packed(bmbs_ctx* X, const uint64_t* rows1, const uint64_t* rows2, int32_t pwords, const char* qual1, const char* qual2,
                                  const uint16_t* len1, const uint16_t* len2, int32_t L_max, int32_t stride, int64_t n_pairs, bmbs_result* results,
                                  uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used)
{
    if (X && (!rows1 || !rows2 || !qual1 || !qual2 || pwords <= 0 || (len1 != nullptr) != (len2 != nullptr))) { X->err = "packed reads: NULL rows / qualities, pwords <= 0, or only one of len1 / len2"; return BMBS_EINVAL; }
    HostIn in = {nullptr, qual1, nullptr, qual2, len1, len2};
    in.rows1 = rows1; in.rows2 = rows2; in.hw = pwords;
    return dispatch_host(X, true, in, L_max, stride, n_pairs, results, cigar_pool, cigar_cap, n_cigar_used);
}
// ASCII rows -> packed rows on the host's threads (what a caller's reader does once per batch).  -> BMBS_OK, or BMBS_EINVAL with *bad_row =
// the first row that holds a character other than A C G T N
extern "C" int bmbs_pack_rows(const char* seq, int32_t L_max, int32_t stride, int64_t n, const uint16_t* len, uint64_t* rows, int32_t pwords, int32_t threads,
                              int64_t* bad_row)
{
    const int W = (L_max + 31) / 32, M = (L_max + 63) / 64;
    if (bad_row) *bad_row = -1;
    if (!seq || !rows || L_max <= 0 || L_max > BMBS_MAX_READ || stride < L_max || pwords < W + M || n < 0) return BMBS_EINVAL;
    static unsigned char code[256]; static std::once_flag once;
    std::call_once(once, [] { for (int i = 0; i < 256; i++) code[i] = 5; code[(int)'A'] = 0; code[(int)'C'] = 1; code[(int)'G'] = 2; code[(int)'T'] = 3; code[(int)'N'] = 4; });
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(threads > 0 ? threads : 1, n / 4096 + 1));
    std::vector<int64_t> bad((size_t)T, -1);
    auto work = [&](int t) {
        const int64_t a = n * t / T, e = n * (t + 1) / T;
        for (int64_t r = a; r < e; r++) {
            const unsigned char* s = reinterpret_cast<const unsigned char*>(seq) + (size_t)r * (size_t)stride;
            uint64_t* o = rows + (size_t)r * (size_t)pwords;
            const int Lr = len ? (int)len[r] : L_max;
            for (int q = 0; q < pwords; q++) o[q] = 0;
            // (the rule upload_lens applies later: a length of 0 or beyond L_max would read past the row and write past its words)
            if (Lr <= 0 || Lr > L_max) { if (bad[(size_t)t] < 0) bad[(size_t)t] = r; continue; }
            for (int j = 0; j < Lr; j++) {
                const unsigned c = code[s[j]];
                if (c < 4) o[j >> 5] |= (uint64_t)c << (2 * (j & 31));
                else if (c == 4) o[W + (j >> 6)] |= 1ull << (j & 63);
                else if (bad[(size_t)t] < 0) bad[(size_t)t] = r;
            }
        }
    };
    if (T == 1) work(0);
    else { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(work, t); for (auto& x : th) x.join(); }
    for (int t = 0; t < T; t++) if (bad[(size_t)t] >= 0) { if (bad_row) *bad_row = bad[(size_t)t]; return BMBS_EINVAL; }
    return BMBS_OK;
}


// ------------------------------------------------------------------------------------------------
static int lane_locate_batch(Lane* c, const uint64_t* row, int64_t n_rows, uint64_t* pos)
{
    if (!c) return BMBS_EINVAL;
    if (!c->attached) { c->err = "no index attached"; return BMBS_ESTATE; }
    if (n_rows <= 0) return BMBS_OK;
    HIPCHK(c, hipSetDevice(c->dev));
    const u64 m = (u64)n_rows;
    ENS(c, c->in_a, m * 8); ENS(c, c->in_b, m * 8);
    HIPCHK(c, hipMemcpyAsync(c->in_a.p, row, m * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_locate_rows, dim3(nblk(m, 256)), dim3(256), 0, c->stream, c->ix, c->in_a.as<u64>(), (long)m, c->in_b.as<u64>());
    HIPCHK(c, hipMemcpyAsync(pos, c->in_b.p, m * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BMBS_OK;
}

// A second context on the index of another one (same device): it takes the device pointers, owns none of the index memory and
// has its own stream and work buffers.  Two such contexts driven by two host threads keep two batches in flight: the kernels of
// one hide the memory waits and the low-occupancy phases of the other's.
extern "C" int bmbs_index_share(bmbs_ctx* X, const bmbs_ctx* owner)
{
    if (!X || !owner || X->lanes.empty() || owner->lanes.empty()) return BMBS_EINVAL;
    const Lane* o = owner->lanes[0];
    if (!o->attached) { X->err = "index share: the owner has no index attached"; return BMBS_ESTATE; }
    if (X->dev != owner->dev) { X->err = "index share: contexts on different devices"; return BMBS_EINVAL; }
    if (X->lanes[0]->attached) { X->err = "index share: this context already has an index"; return BMBS_ESTATE; }
    for (Lane* c : X->lanes) { c->ix = o->ix; c->rows = o->rows; c->o3 = o->o3; c->attached = true; }
    return BMBS_OK;
}

static int lane_vote_order_batch(Lane* c, const uint8_t* vote, const int64_t* seg_off, int64_t n_seg, int32_t form,
                                     uint32_t* perm)
{
    if (!c) return BMBS_EINVAL;
    if (n_seg <= 0) return BMBS_OK;
    if (form != 0 && form != 1) { c->err = "vote order: form is 0 (wave) or 1 (block)"; return BMBS_EINVAL; }
    const int64_t cap = form ? VL_CAP : VM_CAP;
    for (int64_t s = 0; s < n_seg; s++)
        if (seg_off[s + 1] < seg_off[s] || seg_off[s + 1] - seg_off[s] > cap) { c->err = "vote order: list longer than the form's capacity"; return BMBS_EINVAL; }
    const u64 m = (u64)seg_off[n_seg] - (u64)seg_off[0];
    if (m == 0) return BMBS_OK;
    HIPCHK(c, hipSetDevice(c->dev));
    ENS(c, c->in_a, ((u64)n_seg + 1) * 8); ENS(c, c->in_b, m * 4); ENS(c, c->in_seq, (u64)seg_off[n_seg] + 64);
    HIPCHK(c, hipMemcpyAsync(c->in_a.p, seg_off, ((u64)n_seg + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->in_seq.p, vote, (u64)seg_off[n_seg], hipMemcpyHostToDevice, c->stream));
    const u64 grid = (u64)n_seg < 65536 ? (u64)n_seg : 65536;
    // perm is indexed like vote (seg_off[0] may be > 0): the kernel writes perm[a + j], so shift the base
    u32* dperm = c->in_b.as<u32>() - seg_off[0];
    if (form) hipLaunchKernelGGL((k_vote_order<VL_CAP, VL_BLOCK>), dim3(grid), dim3(VL_BLOCK), 0, c->stream, c->in_seq.as<uint8_t>(), c->in_a.as<long>(), (long)n_seg, dperm);
    else hipLaunchKernelGGL((k_vote_order<VM_CAP, VM_BLOCK>), dim3(grid), dim3(VM_BLOCK), 0, c->stream, c->in_seq.as<uint8_t>(), c->in_a.as<long>(), (long)n_seg, dperm);
    HIPCHK(c, hipMemcpyAsync(perm + seg_off[0], c->in_b.p, m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BMBS_OK;
}

static int lane_window_batch(Lane* c, const uint64_t* site, int64_t n_sites, int32_t len, char* out)
{
    if (!c) return BMBS_EINVAL;
    if (!c->attached) { c->err = "no index attached"; return BMBS_ESTATE; }
    if (n_sites <= 0) return BMBS_OK;
    if (len <= 0 || len > 1024) { c->err = "window length out of range"; return BMBS_EINVAL; }
    HIPCHK(c, hipSetDevice(c->dev));
    const u64 m = (u64)n_sites;
    ENS(c, c->in_b, m * 8); ENS(c, c->in_seq, m * (u64)len + 64);
    HIPCHK(c, hipMemcpyAsync(c->in_b.p, site, m * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_window, dim3(nblk(m, 256)), dim3(256), 0, c->stream, c->ix, c->in_b.as<u64>(), (long)m, len, c->in_seq.as<char>());
    HIPCHK(c, hipMemcpyAsync(out, c->in_seq.p, m * (u64)len, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BMBS_OK;
}

static int lane_filter_batch(Lane* c, const char* seq, int32_t L, int32_t stride, int64_t n_reads,
                                 const uint32_t* read_of, const uint64_t* site, int64_t n_cand, uint32_t* err,
                                 int32_t* end_site, bool packed)
{
    if (!c) return BMBS_EINVAL;
    if (!c->attached) { c->err = "no index attached"; return BMBS_ESTATE; }
    HIPCHK(c, hipSetDevice(c->dev));
    if (n_cand <= 0) return BMBS_OK;
    if (L <= 0 || L > BMBS_MAX_READ || stride < L || n_reads < 0) { c->err = "bad read geometry"; return BMBS_EINVAL; }
    if (!seq || !read_of || !site || !err || !end_site) { c->err = "filter batch: NULL buffer"; return BMBS_EINVAL; }
    for (int64_t i = 0; i < n_cand; i++) if ((int64_t)read_of[i] >= n_reads) { c->err = "filter batch: read_of out of range"; return BMBS_EINVAL; }
    const u64 bytes = (u64)n_reads * stride, m = (u64)n_cand;
    { const int r0 = prepare_luts(c); if (r0) return r0; }
    ENS(c, c->in_a, m * 4); ENS(c, c->in_b, m * 8); ENS(c, c->ferr, m * 4); ENS(c, c->fend, m * 4);
    { int ds = 0; int r1 = upload_rows(c, c->in_seq, seq, L, stride, (u64)n_reads, &ds); if (r1) return r1; stride = ds; (void)bytes; }
    HIPCHK(c, hipMemcpyAsync(c->in_a.p, read_of, m * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->in_b.p, site, m * 8, hipMemcpyHostToDevice, c->stream));
    PackedRows pr = {nullptr, nullptr, 0, 0};
    const ReadGeom gm = geom(c, L, nullptr);
    if (packed) {
        // the rows packed on the device exactly as the mapping calls pack them (k_pack_rows: 2 bits per base + the not-ACGT plane and the
        // per-row dirty byte), then the Myers form those calls run (bpm_planes<W, true>)
        const u64 nrd = (u64)n_reads;
        const int pwords = pack_words(gm.L), W = pack_base_words(gm.L);
        { int rz_ = ensure(c, c->prow, nrd * (u64)pwords * 8 + 64, true); if (rz_) return rz_; } ENS(c, c->prow_dirty, nrd + 64);
        HIPCHK(c, hipMemsetAsync(c->prow_dirty.p, 0, nrd + 64, c->stream));
        hipLaunchKernelGGL(k_pack_rows, dim3(nblk(nrd * (u64)(stride / 16), 256)), dim3(256), 0, c->stream, c->in_seq.as<char>(), gm, stride, (long)nrd,
                           c->prow.as<u64>(), pwords, W, c->prow_dirty.as<u32>());
        pr.base = c->prow.as<u64>(); pr.dirty = c->prow_dirty.as<u8>(); pr.pwords = pwords; pr.W = W;
    }
    hipLaunchKernelGGL(k_filter_pairs, dim3(nblk(m, 256)), dim3(256), 0, c->stream, c->ix, c->in_seq.as<char>(), pr, gm, stride, m,
                       c->in_a.as<u32>(), c->in_b.as<u64>(), c->ferr.as<u32>(), c->fend.as<int>());
    HIPCHK(c, hipMemcpyAsync(err, c->ferr.p, m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(end_site, c->fend.p, m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BMBS_OK;
}

static int lane_align_batch(Lane* c, const char* seq, const char* qual, int32_t L, int32_t stride, int64_t n_reads,
                                const uint32_t* read_of, const uint64_t* site, const int32_t* end_site_in,
                                const uint32_t* err_in, int64_t n_jobs, int32_t* start_site, int32_t* end_site,
                                uint32_t* nm, int32_t* score, uint32_t* cigar_ops, int32_t* n_ops, int32_t max_ops)
{
    if (!c) return BMBS_EINVAL;
    if (!c->attached) { c->err = "no index attached"; return BMBS_ESTATE; }
    HIPCHK(c, hipSetDevice(c->dev));
    if (n_jobs <= 0) return BMBS_OK;
    if (L <= 0 || L > BMBS_MAX_READ || stride < L || n_reads < 0) { c->err = "bad read geometry"; return BMBS_EINVAL; }
    if (max_ops < cigar_ops_bound(c->prm, L, threshold_k(c->prm, L))) { c->err = "align batch: max_ops must be at least bmbs_max_cigar_ops(L)"; return BMBS_EINVAL; }
    for (int64_t i = 0; i < n_jobs; i++) if ((int64_t)read_of[i] >= n_reads) { c->err = "align batch: read_of out of range"; return BMBS_EINVAL; }
    const u64 bytes = (u64)n_reads * stride, m = (u64)n_jobs;
    { const int r0 = prepare_luts(c); if (r0) return r0; }
    ENS(c, c->in_a, m * 4); ENS(c, c->in_b, m * 8); ENS(c, c->in_c, m * 4); ENS(c, c->in_d, m * 4);
    ENS(c, c->cig_pool, m * (u64)max_ops * 4);
    {
        int ds = 0;
        int r1 = upload_rows(c, c->in_seq, seq, L, stride, (u64)n_reads, &ds); if (r1) return r1;
        r1 = upload_rows(c, c->in_qual, qual, L, stride, (u64)n_reads, &ds); if (r1) return r1;
        stride = ds; (void)bytes;
    }
    HIPCHK(c, hipMemcpyAsync(c->in_a.p, read_of, m * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->in_b.p, site, m * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->in_c.p, end_site_in, m * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->in_d.p, err_in, m * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->cig_pool.p, 0, m * (u64)max_ops * 4, c->stream));
    c->cur_slot = 0; c->prof_used[0] = 0;
    Jobs jobs = {c->in_a.as<u32>(), c->in_b.as<u64>(), c->in_c.as<int>(), c->in_d.as<u32>()};
    int rc = run_align(c, c->in_seq.as<char>(), c->in_qual.as<char>(), geom(c, L, nullptr), stride, m, jobs, 0xffffffffu, c->cig_pool.as<u32>(), max_ops);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(start_site, c->a_start.p, m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(end_site, c->a_end.p, m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(nm, c->a_nm.p, m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(score, c->a_score.p, m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(n_ops, c->a_nops.p, m * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cigar_ops, c->cig_pool.p, m * (u64)max_ops * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c, 0);
    return BMBS_OK;
}

static int lane_seed_batch(Lane* c, const char* seq, int32_t L, int32_t stride, int64_t n_reads, uint8_t* verdict,
                               uint64_t* exit_site, uint64_t* seg_off, uint32_t* n_votes, uint64_t* vote_site,
                               uint32_t* vote_cnt, int64_t vote_cap, int64_t* total_slots)
{
    if (!c) return BMBS_EINVAL;
    if (!c->attached) { c->err = "no index attached"; return BMBS_ESTATE; }
    HIPCHK(c, hipSetDevice(c->dev));
    const u64 n = (u64)n_reads, bytes = n * (u64)stride;
    if (total_slots) *total_slots = 0;
    if (n == 0) return BMBS_OK;
    if (L <= 0 || L > BMBS_MAX_READ || stride < L || n_reads < 0) { c->err = "bad read geometry"; return BMBS_EINVAL; }
    { const int r0 = prepare_luts(c); if (r0) return r0; }
    c->cur_slot = 0; c->prof_used[0] = 0;
    int rc = per_read_workspace(c, n);
    if (rc) return rc;
    { int ds = 0; int r1 = upload_rows(c, c->in_seq, seq, L, stride, n, &ds); if (r1) return r1; stride = ds; (void)bytes; }
    HIPCHK(c, hipMemsetAsync(c->counters.p, 0, BMBS_SHARDS * BMBS_SHARD_WORDS * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->exit_site.p, 0, n * 8, c->stream));
    u64 tot = 0;
    rc = run_seed_stages(c, c->in_seq.as<char>(), geom(c, L, nullptr), stride, n, &tot);
    if (rc) return rc;
    if (total_slots) *total_slots = (int64_t)tot;
    if ((u64)vote_cap < tot) { c->err = "vote buffers too small"; (void)hipStreamSynchronize(c->stream); return BMBS_ENOMEM; }
    HIPCHK(c, hipMemcpyAsync(verdict, c->verdict.p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(exit_site, c->exit_site.p, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(seg_off, c->cand_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(n_votes, c->n_votes.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    std::vector<bmbs_vote> hv(tot ? tot : 1);
    if (tot) HIPCHK(c, hipMemcpyAsync(hv.data(), c->votes.p, tot * sizeof(bmbs_vote), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (u64 i = 0; i < tot; i++) { vote_site[i] = hv[i].site; vote_cnt[i] = hv[i].vote; }
    return BMBS_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int bmbs_stats_get(bmbs_ctx* X, int64_t stats[5])
{
    if (!X) return BMBS_EINVAL;
    { const int rc = settle_all(X); if (rc) return rc; }
    static_assert(sizeof(unsigned long long) == 8, "");
    for (int j = 0; j < 5; j++) stats[j] = 0;
    for (Lane* c : X->lanes) {
        uint64_t all[BMBS_SHARDS * BMBS_SHARD_WORDS];
        HIPCHK(c, hipSetDevice(c->dev));
        HIPCHK(c, hipMemcpy(all, c->stats.p, sizeof(all), hipMemcpyDeviceToHost));
        for (int j = 0; j < 5; j++) { uint64_t t = 0; for (int sdx = 0; sdx < BMBS_SHARDS; sdx++) t += all[sdx * BMBS_SHARD_WORDS + j]; stats[j] += (int64_t)t; }
    }
    return BMBS_OK;
}
extern "C" int bmbs_stats_reset(bmbs_ctx* X)
{
    if (!X) return BMBS_EINVAL;
    { const int rc = settle_all(X); if (rc) return rc; }
    for (Lane* c : X->lanes) {
        HIPCHK(c, hipSetDevice(c->dev));
        HIPCHK(c, hipMemsetAsync(c->stats.p, 0, BMBS_SHARDS * BMBS_SHARD_WORDS * 8, c->stream));
    }
    return BMBS_OK;
}
extern "C" int bmbs_stats_allreduce(bmbs_ctx** ctxs, int n, int64_t stats[5])
{
    for (int j = 0; j < 5; j++) stats[j] = 0;
    for (int i = 0; i < n; i++) {
        int64_t s[5];
        int rc = bmbs_stats_get(ctxs[i], s);
        if (rc) return rc;
        for (int j = 0; j < 5; j++) stats[j] += s[j];
    }
    return BMBS_OK;
}

// the kernels of the last mapping call, lane by lane (a split call has every kernel once per lane it ran on: the caller adds them up)
extern "C" int bmbs_profile_last(bmbs_ctx* X, const char** names, float* ms, int* n)
{
    if (!X || !n) return BMBS_EINVAL;
    { const int rc = settle_all(X); if (rc) return rc; }
    int cap = *n, cntp = 0;
    for (int li = 0; li < X->used_lanes && li < (int)X->lanes.size(); li++) {
        Lane* c = X->lanes[(size_t)li];
        for (size_t i = 0; i < c->last.size() && cntp < cap; i++) { names[cntp] = c->last[i].name; ms[cntp] = (float)c->last[i].ms; cntp++; }
    }
    *n = cntp;
    return BMBS_OK;
}

// sums over every mapping call settled since the last bmbs_profile_reset (all lanes): the bench reads per-kernel averages from
// here once per timed region instead of waiting for the device after every call
extern "C" int bmbs_profile_total(bmbs_ctx* X, const char** names, double* ms, int* n, int64_t* calls)
{
    if (!X || !n) return BMBS_EINVAL;
    { const int rc = settle_all(X); if (rc) return rc; }
    int cap = *n, cntp = 0;
    int64_t nc = 0;
    for (Lane* c : X->lanes) {
        nc += (int64_t)c->acc_calls;
        for (const auto& a : c->acc) {
            int at = -1;
            for (int j = 0; j < cntp; j++) if (!strcmp(names[j], a.name)) { at = j; break; }
            if (at < 0) { if (cntp >= cap) continue; at = cntp++; names[at] = a.name; ms[at] = 0; }
            ms[at] += a.ms;
        }
    }
    *n = cntp;
    if (calls) *calls = nc;
    return BMBS_OK;
}
extern "C" int bmbs_profile_reset(bmbs_ctx* X)
{
    if (!X) return BMBS_EINVAL;
    { const int rc = settle_all(X); if (rc) return rc; }
    for (Lane* c : X->lanes) { c->acc.clear(); c->acc_calls = 0; }
    return BMBS_OK;
}

extern "C" int bmbs_counters_last(bmbs_ctx* X, uint64_t out[8])
{
    uint64_t all[32];
    int rc = bmbs_counters_all(X, all);
    if (rc) return rc;
    for (int i = 0; i < 8; i++) out[i] = all[i];
    return BMBS_OK;
}

extern "C" int bmbs_counters_all(bmbs_ctx* X, uint64_t out[32])
{
    if (!X) return BMBS_EINVAL;
    { const int rc = settle_all(X); if (rc) return rc; }
    for (int j = 0; j < 32; j++) out[j] = 0;
    uint64_t cand = 0, jobs = 0;
    for (int li = 0; li < X->used_lanes && li < (int)X->lanes.size(); li++) {
        Lane* c = X->lanes[(size_t)li];
        uint64_t all[BMBS_SHARDS * BMBS_SHARD_WORDS];
        HIPCHK(c, hipSetDevice(c->dev));
        HIPCHK(c, hipMemcpy(all, c->counters.p, sizeof(all), hipMemcpyDeviceToHost));
        for (int j = 0; j < 32; j++) { uint64_t t = 0; for (int sdx = 0; sdx < BMBS_SHARDS; sdx++) t += all[sdx * BMBS_SHARD_WORDS + j]; out[j] += t; }
        cand += c->last_total_cand; jobs += c->last_n_jobs;
    }
    out[6] = cand;
    out[7] = jobs;
    return BMBS_OK;
}

// room for the work buffers of later calls, taken now (a lane's buffers are carved out of large slabs; the first slab costs a
// device-wide hipMalloc): a driver calls this while it loads, so that its first batches do not pay for it
extern "C" int bmbs_reserve(bmbs_ctx* X, uint64_t bytes_per_lane)
{
    if (!X) return BMBS_EINVAL;
    for (Lane* c : X->lanes) {
        if (!c->kn.arena) continue;
        HIPCHK(c, hipSetDevice(c->dev));
        size_t have = 0;
        for (const auto& sl : c->arena.slabs) have += sl.cap - sl.used;
        if (have >= bytes_per_lane) continue;
        const size_t cap = (size_t)bytes_per_lane;
        void* p = nullptr;
        if (hipMalloc(&p, cap) != hipSuccess) { X->err = "bmbs_reserve: out of device memory"; return BMBS_ENOMEM; }
        c->arena.slabs.push_back({(char*)p, cap, 0});
        c->arena.total += cap;
    }
    return BMBS_OK;
}

// first use of a page-locked buffer by the device is slow (the first 361 MB download into a fresh buffer took 44 ms instead of 7 on
// the MI355X boxes): a driver lets every staging buffer be touched once in each direction while it loads
extern "C" int bmbs_host_prefault(bmbs_ctx* X, void* p, uint64_t bytes, int32_t kind)
{
    Lane* c = lane0(X);
    if (!c || !p) return BMBS_EINVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    const u64 piece = 64ull << 20;
    DevBuf tmp;
    void* d = nullptr;
    if (hipMalloc(&d, piece) != hipSuccess) return BMBS_ENOMEM;
    hipStream_t st = c->down_stream ? c->down_stream : c->stream;
    hipError_t e = hipSuccess;
    for (u64 o = 0; o < bytes && e == hipSuccess; o += piece) {
        const u64 m = std::min(piece, bytes - o);
        if (kind != 2) e = hipMemcpyAsync(d, (char*)p + o, m, hipMemcpyHostToDevice, st);
        if (kind != 1 && e == hipSuccess) e = hipMemcpyAsync((char*)p + o, d, m, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    (void)tmp;
    return e == hipSuccess ? BMBS_OK : BMBS_ESTATE;
}

// diagnostic: calls that were issued again with exact sizes because a stage count did not fit the capacity learned so far
extern "C" int64_t bmbs_retries(bmbs_ctx* X)
{
    int64_t t = 0;
    if (X) for (Lane* c : X->lanes) t += (int64_t)c->n_retries;
    return t;
}

// ---- the stage entry points and the FASTQ-text calls run on lane 0 -----------------------------------------------------------
extern "C" int bmbs_index_attach(bmbs_ctx* X, const bmbs_index_view* v)
{
    Lane* c = lane0(X);
    if (!c) return BMBS_EINVAL;
    const int rc = lane_index_attach(c, v);
    if (rc) return fin(X, c, rc);
    for (size_t i = 1; i < X->lanes.size(); i++) { Lane* o = X->lanes[i]; o->ix = c->ix; o->rows = c->rows; o->o3 = c->o3; o->attached = true; }
    return BMBS_OK;
}
extern "C" int bmbs_locate_batch(bmbs_ctx* X, const uint64_t* row, int64_t n_rows, uint64_t* pos) { ON_LANE0(lane_locate_batch(c, row, n_rows, pos)); }
extern "C" int bmbs_vote_order_batch(bmbs_ctx* X, const uint8_t* vote, const int64_t* seg_off, int64_t n_seg, int32_t form, uint32_t* perm)
{ ON_LANE0(lane_vote_order_batch(c, vote, seg_off, n_seg, form, perm)); }
extern "C" int bmbs_window_batch(bmbs_ctx* X, const uint64_t* site, int64_t n_sites, int32_t len, char* out) { ON_LANE0(lane_window_batch(c, site, n_sites, len, out)); }
extern "C" int bmbs_filter_batch(bmbs_ctx* X, const char* seq, int32_t L, int32_t stride, int64_t n_reads, const uint32_t* read_of, const uint64_t* site,
                                 int64_t n_cand, uint32_t* err, int32_t* end_site)
{ ON_LANE0(lane_filter_batch(c, seq, L, stride, n_reads, read_of, site, n_cand, err, end_site, false)); }
extern "C" int bmbs_filter_batch_packed(bmbs_ctx* X, const char* seq, int32_t L, int32_t stride, int64_t n_reads, const uint32_t* read_of, const uint64_t* site,
                                        int64_t n_cand, uint32_t* err, int32_t* end_site)
{ ON_LANE0(lane_filter_batch(c, seq, L, stride, n_reads, read_of, site, n_cand, err, end_site, true)); }
extern "C" int bmbs_align_batch(bmbs_ctx* X, const char* seq, const char* qual, int32_t L, int32_t stride, int64_t n_reads, const uint32_t* read_of,
                                const uint64_t* site, const int32_t* end_site_in, const uint32_t* err_in, int64_t n_jobs, int32_t* start_site,
                                int32_t* end_site, uint32_t* nm, int32_t* score, uint32_t* cigar_ops, int32_t* n_ops, int32_t max_ops)
{ ON_LANE0(lane_align_batch(c, seq, qual, L, stride, n_reads, read_of, site, end_site_in, err_in, n_jobs, start_site, end_site, nm, score, cigar_ops, n_ops, max_ops)); }
extern "C" int bmbs_seed_batch(bmbs_ctx* X, const char* seq, int32_t L, int32_t stride, int64_t n_reads, uint8_t* verdict, uint64_t* exit_site,
                               uint64_t* seg_off, uint32_t* n_votes, uint64_t* vote_site, uint32_t* vote_cnt, int64_t vote_cap, int64_t* total_slots)
{ ON_LANE0(lane_seed_batch(c, seq, L, stride, n_reads, verdict, exit_site, seg_off, n_votes, vote_site, vote_cnt, vote_cap, total_slots)); }

// BMBS_HOST_INTERLEAVE=1: the pages of page-locked buffers are spread over all NUMA nodes of the host (set_mempolicy around the
// allocation): a file-to-file run moves ~1.3 KB per read through host memory, and one node of a 4-nodes-per-socket host has a
// quarter of the socket's bandwidth
static long set_policy_interleave(bool on)
{
#ifdef SYS_set_mempolicy
    unsigned long mask[16];
    memset(mask, 0, sizeof mask);
    int nodes = 0;
    if (FILE* f = fopen("/sys/devices/system/node/online", "r")) {
        int a = 0, b = 0;
        const int k = fscanf(f, "%d-%d", &a, &b);
        fclose(f);
        nodes = k == 2 ? b + 1 : 1;
    }
    if (nodes < 2) return -1;
    for (int i = 0; i < nodes && i < 1024; i++) mask[i / (8 * sizeof(long))] |= 1ul << (i % (8 * sizeof(long)));
    return syscall(SYS_set_mempolicy, on ? 3 /* MPOL_INTERLEAVE */ : 0 /* MPOL_DEFAULT */, on ? mask : nullptr, on ? (unsigned long)(nodes + 1) : 0ul);
#else
    (void)on; return -1;
#endif
}
static void* host_alloc_flags(uint64_t bytes, unsigned extra);
extern "C" void* bmbs_host_alloc(uint64_t bytes) { return host_alloc_flags(bytes, 0); }
// kind 1: written by the host, read by the device (FASTQ windows); kind 2: written by the device, read by the host (SAM text).
// The HIP flags of either kind can be set from the environment for experiments (BMBS_PIN_IN_FLAGS / BMBS_PIN_OUT_FLAGS, numeric:
// 0x40000000 hipHostMallocNonCoherent, 0x4 hipHostMallocWriteCombined)
extern "C" void* bmbs_host_alloc_kind(uint64_t bytes, int32_t kind)
{
    static const unsigned in_flags = [] { const char* e = getenv("BMBS_PIN_IN_FLAGS"); return e ? (unsigned)strtoul(e, nullptr, 0) : 0u; }();
    static const unsigned out_flags = [] { const char* e = getenv("BMBS_PIN_OUT_FLAGS"); return e ? (unsigned)strtoul(e, nullptr, 0) : 0u; }();
    return host_alloc_flags(bytes, kind == 1 ? in_flags : kind == 2 ? out_flags : 0u);
}
static void* host_alloc_flags(uint64_t bytes, unsigned extra)
{
    static const bool interleave = [] { const char* e = getenv("BMBS_HOST_INTERLEAVE"); return e && !strcmp(e, "1"); }();
    void* p = nullptr;
    unsigned flags = hipHostMallocPortable | extra;                           // portable: contexts on several devices copy from it
    bool pol = false;
    if (interleave && set_policy_interleave(true) == 0) { pol = true; flags |= hipHostMallocNumaUser; }
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, flags);
    if (pol) (void)set_policy_interleave(false);
    if (e != hipSuccess) return nullptr;
    return p;
}
extern "C" void bmbs_host_free(void* p) { if (p) (void)hipHostFree(p); }
