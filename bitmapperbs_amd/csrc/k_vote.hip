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
    // long lists (a read inside a repeat family: up to 25 seeds x 1000 rows): LSD radix sort, two bits a pass over the bits that
    // vary -- 17 passes of one block scan each for a 6.2 G text, where the bitonic network takes 78 stages of 8 sweeps over 4096
    // keys.  On a GRCh38-like genome these lists were most of k_vote_pe_long's 11-13 ms per 10 M pairs.  (Four bits a pass -- nine
    // passes, sixteen counters in four words -- was measured in round 6: the same time at equal occupancy, more registers.)
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
// (amdgpu_waves_per_eu(5): 96 registers instead of 114-128 -- one more wave per SIMD, which the 15 KB of LDS of the 1024-key class
// allow: the kernel issues 0.2 instructions per cycle and SIMD at four waves, every wave waiting on its own chain of LDS round
// trips and barriers; tools/vote_kbench.hip: 3.06 -> 2.66 ms per 60 000 lists of 660.  The wave form (4 KB of LDS) takes seven: 72
// registers; 400 000 lists of 60 / 120 on the library's 32 768 waves: 1.83 / 2.92 ms at five, 1.68 / 2.67 at six, 1.58 / 2.47 at
// seven, 1.64 / 2.45 at eight)
template <int CAP, int BLOCK, int LO>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(CAP == 256 ? 7 : 5)))
k_vote_long(DevIndex ix, ReadGeom gm, ReadState st, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
            u64* __restrict__ cand, bmbs_vote* __restrict__ votes, u32* __restrict__ slot_read, u32* __restrict__ big_list,
            unsigned long long* __restrict__ big_count, unsigned long long* __restrict__ counters)
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
        if (counters && threadIdx.x == 0) {
            atomicAdd(&SHARD(counters)[CAP == VM_CAP ? CNT_CAND_LONG : CNT_CAND_BIG], (unsigned long long)nc);
            atomicAdd(&SHARD(counters)[CNT_LISTS_LONG], 1ull);
        }
        const int k = gm.rk(gm.rl(r));
        const u64 off = st.cand_off[r];
        const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        const int ns = st.n_seeds[r];
        bmbs_vote* v = votes + off;
        if constexpr (CAP == VL_CAP) if (nc > CAP) {      // (only the largest class takes what is beyond its capacity: the others do not carry this path's registers)
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
        [[maybe_unused]] constexpr int VP_CLS = CAP == VM_CAP ? 0 : CAP == 1024 ? 1 : CAP == 2048 ? 2 : 3;
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
        if (!vl_sort_votes<CAP, (CAP > 256 ? 128 : 32), (CAP + BLOCK - 1) / BLOCK>(items, nv, keys, sh_w, sh_ctl)) {
            for (int e = threadIdx.x; e < nv; e += BLOCK) {
                const u32 vote = (u32)endpos[e] - (e ? (u32)endpos[e - 1] : 0xffffffffu);
                items[e].x = (vote << 24) | (u32)e;
            }
            __syncthreads();
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
