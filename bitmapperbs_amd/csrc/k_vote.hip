// bitmapperbs_amd/csrc/k_vote.hip -- K5/K6 + a8-a10: locate, per-read candidate sort, run-length votes, reference vote order
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
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
           u64* __restrict__ cand, bmbs_vote* __restrict__ votes, u32* __restrict__ slot_read, unsigned long long* __restrict__ counters)
{
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = it < (long)*count_ptr;
    const long r = act ? (long)list[it] : 0;
    const long nc = act ? (long)st.n_cand[r] : 0;
    wave_count_add(counters, CNT_CAND_MID, (u32)nc);
    if (!act) return;
    const int k = gm.rk(gm.rl(r));
    const u64 off = st.cand_off[r];
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
// the same sort with 4-bit digits (round 6): sixteen 16-bit counters in four u64 words, one block-wide scan of those four words per
// pass -- 9 passes for the 33 varying bits of a GRCh38-size text where the 2-bit form takes 17 (each pass is a chain of three barriers
// and a scan: the passes, not the keys, are what a list of a few hundred sites pays for)
template <int EMAX>
DEVI void vl_radix_sort4(u64* keys, int nc)
{
    __shared__ u64 sh_scan4[4 * 4 + 4];
    const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6, nw = (T + 63) >> 6;
    const int E = (nc + T - 1) / T;                     // <= EMAX
    u64 mine[EMAX];
    u64 diff = 0;
    const u64 k0 = keys[0];
    for (int i = tid; i < nc; i += T) diff |= keys[i] ^ k0;
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    if (lane == 0) sh_scan4[w] = diff;
    __syncthreads();
    diff = 0;
    for (int i = 0; i < nw; i++) diff |= sh_scan4[i];
    __syncthreads();
    const int nbits = diff ? 64 - __builtin_clzll(diff) : 0;
    for (int b = 0; b < nbits; b += 4) {
        u64 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
        for (int e = 0; e < EMAX; e++) {
            const int idx = tid * E + e;
            if (e < E && idx < nc) {
                mine[e] = keys[idx];
                const int d = (int)((mine[e] >> b) & 15);
                const u64 inc = 1ull << (16 * (d & 3));
                const int q = d >> 2;
                c0 += q == 0 ? inc : 0; c1 += q == 1 ? inc : 0; c2 += q == 2 ? inc : 0; c3 += q == 3 ? inc : 0;
            }
        }
        u64 i0 = c0, i1 = c1, i2 = c2, i3 = c3;
        for (int o = 1; o < 64; o <<= 1) {
            const u64 v0 = __shfl_up(i0, o, 64), v1 = __shfl_up(i1, o, 64), v2 = __shfl_up(i2, o, 64), v3 = __shfl_up(i3, o, 64);
            if (lane >= o) { i0 += v0; i1 += v1; i2 += v2; i3 += v3; }
        }
        if (lane == 63) { sh_scan4[4 * w] = i0; sh_scan4[4 * w + 1] = i1; sh_scan4[4 * w + 2] = i2; sh_scan4[4 * w + 3] = i3; }
        __syncthreads();
        u64 b0 = 0, b1 = 0, b2 = 0, b3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        for (int i = 0; i < nw; i++) {
            const u64 x0 = sh_scan4[4 * i], x1 = sh_scan4[4 * i + 1], x2 = sh_scan4[4 * i + 2], x3 = sh_scan4[4 * i + 3];
            if (i < w) { b0 += x0; b1 += x1; b2 += x2; b3 += x3; }
            t0 += x0; t1 += x1; t2 += x2; t3 += x3;
        }
        // first slot of every digit = the totals of the smaller digits, packed like the counters (a total is at most 4096 < 2^16)
        auto starts = [](u64 t, u32& run) -> u64 {
            u64 o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { o |= (u64)run << (16 * j); run += (u32)((t >> (16 * j)) & 0xffff); }
            return o;
        };
        u32 run = 0;
        u64 r0 = starts(t0, run); u64 r1 = starts(t1, run); u64 r2 = starts(t2, run); u64 r3 = starts(t3, run);
        r0 += b0 + i0 - c0; r1 += b1 + i1 - c1; r2 += b2 + i2 - c2; r3 += b3 + i3 - c3;       // + this thread's exclusive share
        __syncthreads();                                // every key is in registers: the array may be overwritten
#pragma unroll
        for (int e = 0; e < EMAX; e++) {
            const int idx = tid * E + e;
            if (e < E && idx < nc) {
                const int d = (int)((mine[e] >> b) & 15);
                const int q = d >> 2, sh = 16 * (d & 3);
                const u64 word = q == 0 ? r0 : q == 1 ? r1 : q == 2 ? r2 : r3;
                const u32 r = (u32)((word >> sh) & 0xffff);
                const u64 inc = 1ull << sh;
                r0 += q == 0 ? inc : 0; r1 += q == 1 ? inc : 0; r2 += q == 2 ? inc : 0; r3 += q == 3 ? inc : 0;
                keys[r] = mine[e];
            }
        }
        __syncthreads();
    }
}
// candidates j0 .. j0 + cnt - 1 of a read located into keys[0, cnt).  build_pref: sh_pref (the running sum of the seeds' hit counts)
// is filled first -- once per read.
DEVI void vl_locate_range(const DevIndex& ix, const SeedRec* my, int ns, long j0, int cnt, u64* keys, u32* sh_pref, bool build_pref)
{
    if (build_pref && threadIdx.x == 0) { u32 a = 0; for (int s2 = 0; s2 < ns; s2++) { sh_pref[s2] = a; a += my[s2].hits; } sh_pref[ns] = a; }
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
        const u32 g = (u32)(j0 + j);
        int s2 = 0;
        while (s2 + 1 < ns && sh_pref[s2 + 1] <= g) s2++;
        const u64 sp = my[s2].sp, adj = (u64)my[s2].len + (u64)my[s2].off;
        keys[j] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + (g - sh_pref[s2]))) - adj;
    }
    __syncthreads();
}
// keys[0, cnt) ascending, by the whole block: returns np2 (keys[cnt, np2) = ~0 padding)
template <int EMAX>
DEVI int vl_sort_keys(u64* keys, int cnt)
{
    int np2 = 32;
    while (np2 < cnt) np2 <<= 1;
    for (int j = cnt + (int)threadIdx.x; j < np2; j += blockDim.x) keys[j] = ~0ull;
    __syncthreads();
    // long lists (a read inside a repeat family: up to 25 seeds x 1000 rows): LSD radix sort over the bits that vary -- 9 four-bit
    // passes of one block scan each for a 6.2 G text, where the bitonic network takes 78 stages of 8 sweeps over 4096 keys.  On a
    // GRCh38-like genome these lists were most of k_vote_pe_long's 11-13 ms per 10 M pairs.
    if (np2 > 512 && blockDim.x >= 128) { vl_radix_sort4<EMAX>(keys, cnt); return np2; }
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
// locate candidates j0 .. j0 + cnt - 1 of a read (cnt <= the LDS capacity) into keys[0, np2) (padded with ~0) and sort them ascending;
// returns np2
template <int EMAX>
DEVI int vl_locate_sort_range(const DevIndex& ix, const SeedRec* my, int ns, long j0, int cnt, u64* keys, u32* sh_pref, bool build_pref)
{
    vl_locate_range(ix, my, ns, j0, cnt, keys, sh_pref, build_pref);
    return vl_sort_keys<EMAX>(keys, cnt);
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
// cut = min(Lpos[T], Rpos[T-1]) (Lpos[0] when T = 0).  The final insertion sort of std::sort is the stable sort of what the
// loop left.  tests/test_sort_order.py checks this formulation -- one range at a time, and level by level as vl_sort_votes_lv
// below runs it -- against std::sort on the CPU.
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
// ---- every pending range of a recursion level in ONE block pass (round 6) ------------------------------------------------------
// Round 3-5 partitioned one range after the other (vl_partition / vl_sort_votes, removed) -- 660 all-equal votes: seven block-wide partitions (660 -> 330 -> 165),
// each a chain of scans and barriers, then eight lanes running the serial loop on ~82 elements while 120 wait: 37-60 % of a long
// list's time (profiles/HISTORY.md, round 5).  The ranges of a level are disjoint, tile [0, nv) and all carry the same depth budget
// (both halves of a partition inherit depth - 1), so a level is one pass over the positions:
//   A  the head of every range of more than 16 elements moves the median of three to its first position (its owner thread)
//   B  every position reads its range's pivot: flags "left cursor stops here" (vote <= pivot) and "right cursor stops here"
//      (vote >= pivot); one block-wide inclusive scan of the two counts, packed in one word, kept per position (Pin)
//   C  a stop's rank within its range is a difference of two Pin entries: Lbuf[f + 1 + rank] = position (ascending),
//      Rbuf[f + 1 + rank] = position (descending)
//   D  slot s of a range swaps items[Lbuf[s]] <-> items[Rbuf[s]] while Lbuf[s] < Rbuf[s] (a prefix of the slots); the head finds the
//      number of swaps T by a binary search and the cut = min(Lbuf[T], Rbuf[T - 1]) (vl_partition's rule)
//   E  positions at and beyond the cut start a new range
// Five barriers per level, ~log2(nv / 16) levels, nothing serial but the medians.  rs[i] = first position of i's range; a range's end
// sits in Rbuf[first] and its cut in Lbuf[first] (slot `first` of a range holds no stop: the pivot is there).
// scratch: 8 * CAP bytes (Pin u32[CAP], Lbuf u16[CAP], Rbuf u16[CAP]); rs: u16[CAP].  Returns false when a range of more than 16
// elements is left at depth 0 (std::sort would heap-sort it): the caller restores the items and takes the serial path.
template <int CAP, int EMAX>
DEVI bool vl_sort_votes_lv(bmbs_vk* items, int nv, u16* rs, void* scratch, int* sh_w)
{
    u32* Pin = (u32*)scratch;
    u16* Lbuf = (u16*)(Pin + CAP);
    u16* Rbuf = Lbuf + CAP;
    const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6, nw = (T + 63) >> 6;
    const int E = (nv + T - 1) / T;                     // <= EMAX
    const int i0 = tid * E;
    if (nv > 16) {
        int depth = 0;
        for (int t = nv; t > 1; t >>= 1) depth += 2;
        for (int e = 0; e < E; e++) { const int i = i0 + e; if (i < nv) rs[i] = 0; }
        if (tid == 0) Rbuf[0] = (u16)nv;
        __syncthreads();
        while (true) {
            // ---- A: medians (and: is any range of more than 16 elements left?)
            bool any = false;
            for (int e = 0; e < E; e++) {
                const int f = i0 + e;
                if (f < nv && rs[f] == f) {
                    const int l = Rbuf[f];
                    if (l - f > 16) {
                        any = true;
                        if (depth > 0) bmbs_sort_detail::move_median_to_first(items, (long)f, (long)f + 1, (long)f + (l - f) / 2, (long)l - 1);
                    }
                }
            }
            if (!__syncthreads_or(any ? 1 : 0)) break;
            if (depth == 0) return false;
            --depth;
            // ---- B: stop flags, inclusive prefix of both counts per position
            u32 loc[EMAX];
            u32 sum = 0;
#pragma unroll
            for (int e = 0; e < EMAX; e++) {
                const int i = i0 + e;
                u32 fl = 0;
                if (e < E && i < nv) {
                    const int f = rs[i], l = Rbuf[f];
                    if (l - f > 16 && i != f) {
                        const u32 pv = items[f].x >> 24, v = items[i].x >> 24;
                        fl = (v <= pv ? 1u : 0u) | (v >= pv ? 0x10000u : 0u);
                    }
                }
                sum += fl;
                loc[e] = sum;
            }
            u32 incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
            if (lane == 63) sh_w[w] = (int)incl;
            __syncthreads();
            u32 base = incl - sum;
            for (int q = 0; q < w; q++) base += (u32)sh_w[q];
#pragma unroll
            for (int e = 0; e < EMAX; e++) { const int i = i0 + e; if (e < E && i < nv) Pin[i] = base + loc[e]; }
            __syncthreads();
            // ---- C: the two stop lists of every range
#pragma unroll
            for (int e = 0; e < EMAX; e++) {
                const int i = i0 + e;
                if (e < E && i < nv) {
                    const u32 mine = Pin[i], prev = e ? base + loc[e - 1] : base;          // prev = Pin[i - 1] (0 in front of the list)
                    const u32 fl = mine - prev;
                    if (fl) {
                        const int f = rs[i], l = Rbuf[f];
                        const u32 pf = Pin[f], pl = Pin[l - 1];
                        if (fl & 1u) Lbuf[f + 1 + (int)((mine & 0xffffu) - 1 - (pf & 0xffffu))] = (u16)i;
                        if (fl >> 16) Rbuf[f + 1 + (int)((pl >> 16) - (mine >> 16))] = (u16)i;
                    }
                }
            }
            __syncthreads();
            // ---- D: swaps, and the cut of every range (into Lbuf[first])
#pragma unroll
            for (int e = 0; e < EMAX; e++) {
                const int i = i0 + e;
                if (e < E && i < nv) {
                    const int f = rs[i], l = Rbuf[f];
                    if (l - f > 16) {
                        const u32 pf = Pin[f], pl = Pin[l - 1];
                        const int nL = (int)((pl & 0xffffu) - (pf & 0xffffu)), nR = (int)((pl >> 16) - (pf >> 16));
                        const int m = nL < nR ? nL : nR;
                        if (i == f) {
                            // T = slots of the prefix that swap (the condition is monotone: one list ascends, the other descends)
                            int lo = 0, hi = m;
                            while (lo < hi) { const int mid = (lo + hi) >> 1; if (Lbuf[f + 1 + mid] < Rbuf[f + 1 + mid]) lo = mid + 1; else hi = mid; }
                            const int Tn = lo;
                            int cut;
                            if (Tn == 0) cut = Lbuf[f + 1];
                            else { const int lt = Tn < nL ? (int)Lbuf[f + 1 + Tn] : 0x7fffffff, rt = Rbuf[f + Tn]; cut = lt < rt ? lt : rt; }
                            Lbuf[f] = (u16)cut;
                        } else {
                            const int t = i - f - 1;
                            if (t < m) {
                                const int a = Lbuf[i], b = Rbuf[i];
                                if (a < b) { const bmbs_vk x = items[a]; items[a] = items[b]; items[b] = x; }
                            }
                        }
                    }
                }
            }
            __syncthreads();
            // ---- E: the positions at and beyond a cut start a new range; the head records both ends
            int nf[EMAX];
#pragma unroll
            for (int e = 0; e < EMAX; e++) {
                const int i = i0 + e;
                nf[e] = -1;
                if (e < E && i < nv) {
                    const int f = rs[i], l = Rbuf[f];
                    if (l - f > 16) {
                        const int cut = Lbuf[f];
                        nf[e] = i >= cut ? cut : f;
                        if (i == f) nf[e] = -2 - cut;            // the head: writes the ends once every thread has read the old ones
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EMAX; e++) {
                const int i = i0 + e;
                if (e < E && i < nv) {
                    if (nf[e] >= 0) rs[i] = (u16)nf[e];
                    else if (nf[e] <= -2) { const int cut = -2 - nf[e]; const int l = Rbuf[i]; Rbuf[cut] = (u16)l; Rbuf[i] = (u16)cut; }
                }
            }
            __syncthreads();
        }
    }
    vl_stable_by_vote_desc<EMAX>(items, nv);
    return true;
}

// items[] back in entry order (every item carries its entry index in the low 24 bits): what the serial fallback starts from
template <int EMAX>
DEVI void vl_restore_items(bmbs_vk* items, int nv)
{
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    const int E = (nv + T - 1) / T;
    u32 mine[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; e++) { const int i = tid * E + e; if (e < E && i < nv) mine[e] = items[i].x; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EMAX; e++) { const int i = tid * E + e; if (e < E && i < nv) items[mine[e] & 0xffffffu].x = mine[e]; }
    __syncthreads();
}

// CAP, BLOCK = (VM_CAP, VM_BLOCK): lists of up to 256 candidates, one wave each; (VL_CAP, VL_BLOCK): the longer ones, one
// block each (the two instances walk the same list and take the reads of their size class: LO < nc <= CAP, the last one also beyond)
// -DVOTE_PROF (tools/vote_prof.sh): cycles per phase of k_vote_long as thread 0 of a block sees them, summed per size class
// [class][0 lists, 1 candidates, 2 distinct sites, 3 locate + site sort, 4 run ends + items, 5 vote order, 6 write-out, 7 vote order fell back to one lane]
#ifdef VOTE_PROF
__device__ unsigned long long g_vote_prof[4][8];
#define VP_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define VP_ADD(cls, slot, val) do { if (threadIdx.x == 0) atomicAdd(&g_vote_prof[cls][slot], (unsigned long long)(val)); } while (0)
#else
#define VP_T(v)
#define VP_ADD(cls, slot, val)
#endif
template <int CAP, int BLOCK, int LO>
__global__ void __launch_bounds__(BLOCK)
k_vote_long(DevIndex ix, ReadGeom gm, ReadState st, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
            u64* __restrict__ cand, bmbs_vote* __restrict__ votes, u32* __restrict__ slot_read, u32* __restrict__ big_list,
            unsigned long long* __restrict__ big_count, unsigned long long* __restrict__ counters)
{
    __shared__ u64 keys[CAP];
    __shared__ u16 endpos[CAP];
    __shared__ bmbs_vk items[CAP];
    __shared__ u32 sh_pref[BMBS_MAX_SEEDS + 1];
    __shared__ int sh_w[2 * (BLOCK / 64) + 1];
    const long total_items = (long)*count_ptr;
    for (long item = blockIdx.x; item < total_items; item += gridDim.x) {
        const long r = list[item];
        const long nc = (long)st.n_cand[r];
        // the wave form sees every listed read and passes the ones beyond its capacity on to a list of their own (see k_vote_pe_long)
        if (big_list && CAP != VL_CAP && nc > CAP) { if (threadIdx.x == 0) big_list[atomicAdd(big_count, 1ull)] = (u32)r; continue; }
        if (nc <= LO || (CAP != VL_CAP && nc > CAP)) continue;          // another instance's size class
        if (counters && threadIdx.x == 0) {
            atomicAdd(&SHARD(counters)[CAP == VM_CAP ? CNT_CAND_LONG : CNT_CAND_BIG], (unsigned long long)nc);
            atomicAdd(&SHARD(counters)[CNT_LISTS_LONG], 1ull);
        }
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
                if (!vl_sort_votes_lv<CAP, (CAP + BLOCK - 1) / BLOCK>(items, nvh, endpos, keys, sh_w)) {
                    vl_restore_items<(CAP + BLOCK - 1) / BLOCK>(items, nvh);
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
        constexpr int VP_CLS = CAP == VM_CAP ? 0 : CAP == 1024 ? 1 : CAP == 2048 ? 2 : 3;
        VP_T(vp0);
        vl_locate_sort<(CAP + BLOCK - 1) / BLOCK>(ix, my, ns, (int)nc, keys, sh_pref);
        VP_T(vp1);
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
        VP_T(vp2);
        // std::sort(votes, compare_seed_votes), Schema.cpp:24986
        // (endpos is free from here on: it becomes the range map of the level-synchronous replay)
        if (!vl_sort_votes_lv<CAP, (CAP + BLOCK - 1) / BLOCK>(items, nv, endpos, keys, sh_w)) {
            vl_restore_items<(CAP + BLOCK - 1) / BLOCK>(items, nv);
            if (threadIdx.x == 0) intro_sort_desc(items, (long)nv);
            __syncthreads();
            VP_ADD(VP_CLS, 7, 1);
        }
        VP_T(vp3);
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
        VP_T(vp4);
        VP_ADD(VP_CLS, 0, 1); VP_ADD(VP_CLS, 1, nc); VP_ADD(VP_CLS, 2, nv);
        VP_ADD(VP_CLS, 3, vp1 - vp0); VP_ADD(VP_CLS, 4, vp2 - vp1); VP_ADD(VP_CLS, 5, vp3 - vp2); VP_ADD(VP_CLS, 6, vp4 - vp3);
    }
}

// a9 alone (stage API, parity tests): the visiting order of given vote lists, one block per list
template <int CAP, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_vote_order(const uint8_t* __restrict__ vote, const long* __restrict__ seg_off, long n_seg, u32* __restrict__ perm)
{
    __shared__ u64 scratch[CAP];
    __shared__ bmbs_vk items[CAP];
    __shared__ u16 rs[CAP];
    __shared__ int sh_w[2 * (BLOCK / 64) + 1];
    for (long sg = blockIdx.x; sg < n_seg; sg += gridDim.x) {
        const long a = seg_off[sg];
        const int nv = (int)(seg_off[sg + 1] - a);
        if (nv <= 0 || nv > CAP) continue;
        for (int e = threadIdx.x; e < nv; e += BLOCK) items[e].x = ((u32)vote[a + e] << 24) | (u32)e;
        __syncthreads();
        if (!vl_sort_votes_lv<CAP, (CAP + BLOCK - 1) / BLOCK>(items, nv, rs, scratch, sh_w)) {
            vl_restore_items<(CAP + BLOCK - 1) / BLOCK>(items, nv);
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
