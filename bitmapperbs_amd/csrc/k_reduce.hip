// bitmapperbs_amd/csrc/k_reduce.hip -- K9: ordered reduction over a read's votes (Schema.cpp:7847-8172 / 8335-8750)
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
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
