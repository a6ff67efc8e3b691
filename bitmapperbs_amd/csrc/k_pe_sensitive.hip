// bitmapperbs_amd/csrc/k_pe_sensitive.hip -- Paired-end sensitive mode (Map_Pair_Seq_end_to_end, Schema.cpp:19953-21459)
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
// ================================================================================================
// Paired-end sensitive mode (Map_Pair_Seq_end_to_end, Schema.cpp:19953-21459)
// ================================================================================================
// Seeding of both mates is the same state machine as fast mode (first seed, 1-mismatch second seed, remaining
// seeds; process_rest_seed[_filter]_debug, Schema.cpp:17574 / 16298), so k_seed_* + k_locate + k_vote_pe are
// shared.  What differs is the order of verification -- the mate with fewer first-seed candidates is verified
// completely (round 1), the other mate's votes are kept only where a verified hit of the first lies within the
// insert window (select_suit_candidates, 4775) and verified (round 2) -- and the rescue: a mate left without a
// hit is re-seeded (reseed_filter, 16678) with fixed segments chosen from its recorded seeds (select_best_seeds,
// 16630) plus seeds sliding by 8, filtered by the verified mate and verified (round 3).

// exists a verified hit of the mate with mind <= |distance| <= maxd; `next_start` is the reference's running
// lower bound (sites arrive in ascending order)
DEVI bool pes_suit(const PeCand* mate, int mate_occ, int& next_start, u64 site, long long maxd, long long mind)
{
    for (int i = next_start; i < mate_occ; i++) {
        const u64 ms = mate[i].site;
        if (ms > site) {
            const long long d = (long long)(ms - site);
            if (d > maxd) return false;
            if (d >= mind) return true;
        } else {
            const long long d = (long long)(site - ms);
            if (d > maxd) next_start = i + 1;
            else if (d >= mind) return true;
        }
    }
    return false;
}

// which mate goes first (Schema.cpp:20870): the one with fewer candidates after its FIRST seed
__global__ void __launch_bounds__(256)
k_pes_order(long n, ReadState st, SeedCarry sc, PeState ps)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    ps.dead[p] = 0; ps.both[p] = 0; ps.npair[p] = 0; ps.sbd[p] = 0;
    u32 c[2];
    for (int m = 0; m < 2; m++) {
        const long r = p + m * n;
        const int v = st.verdict[r], ns = st.n_seeds[r];
        const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        // first seed recorded <=> seeds[0].off == 0 (every later seed starts at an offset >= 1)
        c[m] = (v == 1 || v == 4) ? 0u : (ns >= 1 && my[0].off == 0 ? my[0].hits : 0u);
        // full_seed_id: the terminate seeds that produced candidates; the fixed second seed of a 1-mismatch read
        // is not among them, and when it was usable no further seed was run (extra_seed_flag == 0)
        ps.full[r] = (u8)((sc.flag_c[r] && !sc.flag_d[r]) ? 1 : ns);
        ps.roff[r] = 0;
    }
    const int f = c[0] <= c[1] ? 0 : 1;
    ps.first[p] = (u8)f;
    const long rF = p + (long)f * n;
    const int vF = st.verdict[rF];
    if (vF == 0) ps.dead[p] = 1;                 // best_mapp_occ == 0 -> next pair
    else if (vF == 3) ps.vround[rF] = 1;
}

// pes_suit without its running lower bound: it returns whether ANY verified hit of the mate lies mind <= |distance| <= maxd from
// the site (the bound only skips hits more than maxd below the site, which a larger site cannot use either; the scan ends at the
// first hit more than maxd above it) -- two binary searches in the ascending list.  Sites below 2^63 only (the caller checks).
DEVI bool pes_suit_any(const PeCand* a, int na, u64 s, long long maxd, long long mind)
{
    if (na == 0 || maxd < 0) return false;
    const u64 mn = mind > 0 ? (u64)mind : 0;
    if (mn > (u64)maxd) return false;
    long j = pe_lower_bound(a, na, s + mn);                                        // hits above the site: [s + mn, s + maxd]
    if (j < na && a[j].site - s <= (u64)maxd) return true;
    if (s < mn) return false;
    j = pe_lower_bound(a, na, s > (u64)maxd ? s - (u64)maxd : 0);                  // hits below (or on) it: [s - maxd, s - mn]
    return j < na && a[j].site <= s - mn;
}
// after round 1: filter the second mate's votes by the first mate's verified hits (generate_candidate_votes_shift_filter)
// One lane per pair; a pair whose two lists hold more than 64 entries is done by the whole wave afterwards (pes_suit_any per entry,
// kept entries compacted by ballot) unless a site of its lists wrapped below zero.
__global__ void __launch_bounds__(64)
k_pes_second(long n, ReadGeom gm, PeIns pi, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool act = p < n && !ps.dead[p];
    long long maxd = 0, mind = 0; int large_k;
    long rF = 0, rS = 0;
    int occF = 0;
    if (act) {
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        const int f = ps.first[p];
        rF = p + (long)f * n; rS = p + (long)(1 - f) * n;
        occF = ps.occ[rF];
        if (occF == 0) { ps.dead[p] = 1; act = false; }
        else if (st.verdict[rS] != 3) act = false;             // direct hits / 1-mismatch exit / nothing: no verification
    }
    const long nb = act ? (long)ps.len[rS] : 0;
    const bool coop = act && nb + occF > 64;
    if (act && !coop) {
        const PeCand* a = pe_list(ps, st, A, B, rF);
        PeCand* b = pe_list(ps, st, A, B, rS);
        long kept = 0;
        int next_start = 0;
        for (long j = 0; j < nb; j++) {
            const PeCand cj = b[j];
            if (pes_suit(a, occF, next_start, cj.site, maxd, mind)) b[kept++] = cj;
        }
        ps.len[rS] = (u32)kept;
        ps.vround[rS] = 2;
    }
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long pp = (long)__shfl((long long)p, src, 64);
        long long mxd, mnd; int lk;
        pe_bounds(gm, pi, pp, n, mxd, mnd, lk);
        const int f = ps.first[pp];
        const long rF2 = pp + (long)f * n, rS2 = pp + (long)(1 - f) * n;
        const int na = ps.occ[rF2];
        const long nb2 = (long)ps.len[rS2];
        const PeCand* a = pe_list(ps, st, A, B, rF2);
        PeCand* b = pe_list(ps, st, A, B, rS2);
        if ((a[na - 1].site >> 63) || (nb2 && (b[nb2 - 1].site >> 63))) {
            if (lane == src) {
                long kept = 0;
                int next_start = 0;
                for (long j = 0; j < nb2; j++) {
                    const PeCand cj = b[j];
                    if (pes_suit(a, na, next_start, cj.site, mxd, mnd)) b[kept++] = cj;
                }
                ps.len[rS2] = (u32)kept; ps.vround[rS2] = 2;
            }
            continue;
        }
        long kept = 0;
        for (long base = 0; base < nb2; base += 64) {
            const long j = base + lane;
            PeCand cj; cj.site = 0; cj.err = 0; cj.end = 0;
            bool keep = false;
            if (j < nb2) { cj = b[j]; keep = pes_suit_any(a, na, cj.site, mxd, mnd); }
            const unsigned long long kb = __ballot(keep);      // every entry of the step is in registers before the first is stored
            if (keep) b[kept + __popcll(kb & ((1ull << lane) - 1))] = cj;
            kept += __popcll(kb);
        }
        if (lane == src) { ps.len[rS2] = (u32)kept; ps.vround[rS2] = 2; }
    }
}

// pairs whose second mate has no hit are re-seeded
__global__ void __launch_bounds__(256)
k_pes_reseed_flag(long n, PeState ps, u32* __restrict__ flag)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const long rS = p + (long)(1 - ps.first[p]) * n;
    flag[p] = (!ps.dead[p] && ps.occ[rS] == 0) ? 1u : 0u;
}

// reseed_filter's seeding (Schema.cpp:16678-16900) with select_best_seeds (16630): up to three fixed segments
// (count_hash_table) and then count_backward_as_much_1_terminate seeds sliding by 8.  One re-seeded mate per lane.
template <bool PACKED, bool KG = false>
__global__ void __launch_bounds__(64)
k_pes_reseed(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, long n, const u64* __restrict__ count_ptr,
             const u32* __restrict__ plist, ReadState st, PeState ps, u32* __restrict__ rcnt, unsigned long long* __restrict__ counters)
{
    __shared__ u64 s_c3[KG ? 27 : 1];
    const u64* c3 = KG ? kgram_c3(ix, s_c3) : nullptr;
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    LaneCounters lc = {0, 0, 0, 0, 0};
    if (it < (long)*count_ptr) {
        const long p = plist[it];
        const long r = p + (long)(1 - ps.first[p]) * n;
        const char* rd = seq + (size_t)r * stride;
        const u64* prow = PACKED ? pr.base + (size_t)r * pr.pwords : nullptr;
        const bool dirty = PACKED ? pr.dirty[r] != 0 : false;
        const int L = gm.rl(r);
        SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        const int full = ps.full[r];
        int rs[3], rl[3], rn = 0;
        int s0 = 0, s1 = 0;
        if (full >= 1) s0 = my[0].off;
        if (full >= 2) {
            s1 = my[1].off;
            rn = 2;
            rs[0] = s0; rl[0] = s1 - s0;
            rs[1] = (int)my[full - 2].off + (int)my[full - 2].len; rl[1] = L - rs[1];
        } else if (full == 1) {
            rn = 2;
            rs[0] = s0; rl[0] = L / 2;
            rs[1] = rs[0] + rl[0]; rl[1] = L - rs[1];
        }
        // full == 0: the reference reads index -1 of two malloc'ed int arrays (Schema.cpp:16657), which is the zero
        // upper half of the allocator's chunk-size word: the whole read becomes one fixed seed
        const int last = full >= 1 ? (int)my[full - 1].off + (int)my[full - 1].len : 0;
        if (last < L) { rs[rn] = last; rl[rn] = L - last; rn++; }
        const int max_seed = L / 10 == 0 ? 25 : (L / 10 - 1 > 25 ? 25 : L / 10 - 1);
        const u64 max_hits = 1000, avail = 20;
        int ns = 0, seed_id = 0;
        u64 ncand = 0;
        typename std::conditional<PACKED, SearchP, Search>::type S; SeedHit h;
        while (seed_id < rn) {
            const int tm = rs[seed_id], ml = rl[seed_id];
            if constexpr (PACKED) {
                if (search_begin_p<true>(ix, prow, pr.W, dirty, tm + ml, tm, S, h, lc.n_hash))
                    while (!search_step_p<true, KG>(ix, tm + ml, S, h, lc.n_ext, c3, &lc.n_jump)) {}
            } else {
                if (search_begin<true>(ix, rd, tm + ml, tm, S, h, lc.n_hash))
                    while (!search_step<true>(ix, rd, tm + ml, S, h, lc.n_ext)) {}
            }
            if (h.hits == 1) seed_record(my, ns, ncand, h.sp, 1, (u64)ml, (u64)tm);
            else if ((u64)ml >= avail && h.hits <= max_hits) { if (h.hits != 0) seed_record(my, ns, ncand, h.sp, h.hits, (u64)ml, (u64)tm); }
            else if (L - tm == ml) break;
            seed_id++;
        }
        int tm = full > 1 ? (s0 + s1) / 2 : 4;
        while (seed_id < max_seed && tm < L) {
            if constexpr (PACKED) {
                if (search_begin_p<false>(ix, prow, pr.W, dirty, L, tm, S, h, lc.n_hash))
                    while (!search_step_p<false, KG>(ix, L, S, h, lc.n_ext, c3, &lc.n_jump)) {}
            } else {
                if (search_begin<false>(ix, rd, L, tm, S, h, lc.n_hash))
                    while (!search_step<false>(ix, rd, L, S, h, lc.n_ext)) {}
            }
            if (h.hits == 1) seed_record(my, ns, ncand, h.sp, 1, h.ml, (u64)tm);
            else if (h.ml >= avail && h.hits <= max_hits) { if (h.hits != 0) seed_record(my, ns, ncand, h.sp, h.hits, h.ml, (u64)tm); }
            else if ((u64)(L - tm) == h.ml) break;
            tm += 8;
            seed_id++;
        }
        st.n_seeds[r] = (u8)ns;
        rcnt[it] = (u32)ncand;
    }
    flush_counters(counters, lc, 2);
}

// locate + sort + filtered votes of one re-seeded mate; the list goes to the R buffer (cur = 2)
#define PESV_LONG 32        // re-seeded mates with more candidates than this go to k_pes_vote_long (a block per mate)
__global__ void __launch_bounds__(64)
k_pes_vote(DevIndex ix, long n, ReadGeom gm, PeIns pi, const u64* __restrict__ count_ptr, const u32* __restrict__ plist,
           const u64* __restrict__ roff, ReadState st, PeState ps, u64* __restrict__ rcand, PeCand* __restrict__ A, PeCand* __restrict__ B,
           u32* __restrict__ long_flag, unsigned long long* __restrict__ counters)
{
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act_ = it < (long)*count_ptr;
    wave_count_add(counters, CNT_CAND_RESEED, act_ ? (u32)(roff[it + 1] - roff[it]) : 0u);
    if (!act_) return;
    if (long_flag) long_flag[it] = 0;
    const long p = plist[it];
    const int f = ps.first[p];
    const long rF = p + (long)f * n, r = p + (long)(1 - f) * n;
    const int k = gm.rk(gm.rl(r));
    long long maxd, mind; int large_k;
    pe_bounds(gm, pi, p, n, maxd, mind, large_k);
    const u64 o0 = roff[it], o1 = roff[it + 1];
    ps.roff[r] = o0;
    const long nc = (long)(o1 - o0);
    if (nc == 0) { ps.len[r] = 0; ps.vround[r] = 0; return; }       // no candidate: best_mapp_occ stays 0
    if (long_flag && nc > PESV_LONG) { long_flag[it] = 1; return; }   // a mate inside a repeat family: hundreds of candidates, sorted by a block
    const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
    const int ns = st.n_seeds[r];
    u64* c = rcand + o0;
    long o = 0;
    for (int s = 0; s < ns; s++) {
        const u64 sp = my[s].sp, adj = (u64)my[s].len + (u64)my[s].off;
        for (u32 j = 0; j < my[s].hits; j++) c[o++] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + j)) - adj;
    }
    sort_u64_asc(c, nc);
    const PeCand* a = pe_list(ps, st, A, B, rF);
    const int occF = ps.occ[rF];
    PeCand* out = ps.R + o0;
    long nv = 0;
    int next_start = 0;
    u64 pre = c[0];
    for (long i = 1; i <= nc; i++) {
        if (i < nc && c[i] == pre) continue;
        const u64 site = (i < nc) ? (pre < (u64)k ? 0 : pre - (u64)k) : (pre >= (u64)k ? pre - (u64)k : 0);
        if (pes_suit(a, occF, next_start, site, maxd, mind)) { out[nv].site = site; out[nv].err = 0; out[nv].end = 0; nv++; }
        if (i < nc) pre = c[i];
    }
    ps.cur[r] = 2; ps.len[r] = (u32)nv; ps.vround[r] = 3;
}

// k_pes_vote for the mates it flagged: a block per mate -- candidates located into LDS and sorted (vl_locate_sort), distinct sites
// (vl_run_ends), the window test per site, kept sites compacted in order.  Lists beyond the LDS capacity and lists with sites that
// wrapped below zero take k_pes_vote's loop on one lane.
template <int CAP, int BLOCK, int LO>
__global__ void __launch_bounds__(BLOCK)
k_pes_vote_long(DevIndex ix, long n, ReadGeom gm, PeIns pi, const u64* __restrict__ count_ptr, const u32* __restrict__ items,
                const u32* __restrict__ plist, const u64* __restrict__ roff, ReadState st, PeState ps, u64* __restrict__ rcand,
                PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    __shared__ u64 keys[CAP];
    __shared__ u16 endpos[CAP];
    __shared__ u32 sh_pref[BMBS_MAX_SEEDS + 1];
    __shared__ int sh_w[2 * (BLOCK / 64) + 1];
    const long total = (long)*count_ptr;
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const long it = items[item];
        const u64 o0 = roff[it], o1 = roff[it + 1];
        const long nc = (long)(o1 - o0);
        if (nc <= LO || (CAP != VL_CAP && nc > CAP)) continue;          // another instance's size class
        const long p = plist[it];
        const int f = ps.first[p];
        const long rF = p + (long)f * n, r = p + (long)(1 - f) * n;
        const int k = gm.rk(gm.rl(r));
        long long maxd, mind; int large_k;
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        const int ns = st.n_seeds[r];
        const PeCand* a = pe_list(ps, st, A, B, rF);
        const int occF = ps.occ[rF];
        PeCand* out = ps.R + o0;
        if (nc > CAP) {
            // beyond the LDS capacity: vl_sort_huge, the tiles parked in the output segment (16 bytes per candidate: room for the 8-byte
            // sites); the merged list -- in the candidate segment -- is made distinct and filtered 256 sites a step
            u64* c = rcand + o0;
            vl_sort_huge<CAP, BLOCK>(ix, my, ns, nc, keys, sh_pref, reinterpret_cast<u64*>(out), c);
            if ((occF > 0 && (a[occF - 1].site >> 63)) || (c[nc - 1] >> 63)) {
                // sites that wrapped below zero: the reference's loop over the sorted list, one lane
                if (threadIdx.x == 0) {
                    long nv = 0;
                    int next_start = 0;
                    u64 pre = c[0];
                    for (long i = 1; i <= nc; i++) {
                        if (i < nc && c[i] == pre) continue;
                        const u64 site = pre < (u64)k ? 0 : pre - (u64)k;
                        if (pes_suit(a, occF, next_start, site, maxd, mind)) { out[nv].site = site; out[nv].err = 0; out[nv].end = 0; nv++; }
                        if (i < nc) pre = c[i];
                    }
                    ps.cur[r] = 2; ps.len[r] = (u32)nv; ps.vround[r] = 3;
                }
                __syncthreads();
                continue;
            }
            int running = 0;
            for (long base = 0; base < nc; base += BLOCK) {
                const long i = base + (long)threadIdx.x;
                bool keep = false;
                u64 site = 0;
                if (i < nc) {
                    const u64 key = c[i];
                    if (i == nc - 1 || c[i + 1] != key) { site = key < (u64)k ? 0 : key - (u64)k; keep = pes_suit_any(a, occF, site, maxd, mind); }
                }
                int tot;
                const int pre = vl_prefix(keep, sh_w, tot);
                if (keep) { PeCand c2; c2.site = site; c2.err = 0; c2.end = 0; out[running + pre] = c2; }
                running += tot;
            }
            if (threadIdx.x == 0) { ps.cur[r] = 2; ps.len[r] = (u32)running; ps.vround[r] = 3; }
            __syncthreads();
            continue;
        }
        bool serial = occF > 0 && (a[occF - 1].site >> 63);
        if (!serial) {
            vl_locate_sort<(CAP + BLOCK - 1) / BLOCK>(ix, my, ns, (int)nc, keys, sh_pref);
            serial = (keys[nc - 1] >> 63) != 0;
        }
        if (serial) {
            if (threadIdx.x == 0) {
                u64* c = rcand + o0;
                long o = 0;
                for (int s2 = 0; s2 < ns; s2++) {
                    const u64 sp = my[s2].sp, adj = (u64)my[s2].len + (u64)my[s2].off;
                    for (u32 j = 0; j < my[s2].hits; j++) c[o++] = ix.total - ((sp >> 63) ? (sp & ~(1ull << 63)) : sa_at(ix, sp + j)) - adj;
                }
                sort_u64_asc(c, nc);
                long nv = 0;
                int next_start = 0;
                u64 pre = c[0];
                for (long i = 1; i <= nc; i++) {
                    if (i < nc && c[i] == pre) continue;
                    const u64 site = pre < (u64)k ? 0 : pre - (u64)k;
                    if (pes_suit(a, occF, next_start, site, maxd, mind)) { out[nv].site = site; out[nv].err = 0; out[nv].end = 0; nv++; }
                    if (i < nc) pre = c[i];
                }
                ps.cur[r] = 2; ps.len[r] = (u32)nv; ps.vround[r] = 3;
            }
            __syncthreads();
            continue;
        }
        const int nd = vl_run_ends(keys, (int)nc, endpos, sh_w);
        int running = 0;
        for (int base = 0; base < nd; base += BLOCK) {
            const int e = base + (int)threadIdx.x;
            bool keep = false;
            u64 site = 0;
            if (e < nd) {
                const u64 key = keys[endpos[e]];
                site = key < (u64)k ? 0 : key - (u64)k;
                keep = pes_suit_any(a, occF, site, maxd, mind);
            }
            int tot;
            const int pre = vl_prefix(keep, sh_w, tot);
            if (keep) { PeCand c2; c2.site = site; c2.err = 0; c2.end = 0; out[running + pre] = c2; }
            running += tot;
        }
        if (threadIdx.x == 0) { ps.cur[r] = 2; ps.len[r] = (u32)running; ps.vround[r] = 3; }
        __syncthreads();
    }
}

// new_faster_verify_pairs (Schema.cpp:15773-15900) + hand-over of the winning candidates to K11-K13
// What the reference's loop leaves behind, as a summary of an ORDERED run of (i, j) hits that can be merged left to right:
// the lowest error sum m, where it first occurs, how often it occurs (c), and the lowest sum among the hits before that first
// occurrence (pm) -- the loop's `second` is the running best at the moment the final best was first met (not the true runner-up),
// or the best itself when it was met again afterwards.
struct PairSum { int m, c, pm; long i, j; };
DEVI PairSum pair_comb(const PairSum& L, const PairSum& R)
{
    if (R.c == 0) return L;
    if (L.c == 0) return R;
    PairSum o;
    if (L.m < R.m) o = L;
    else if (L.m > R.m) { o = R; o.pm = L.m < R.pm ? L.m : R.pm; }
    else { o = L; o.c = L.c + R.c; }
    return o;
}
__global__ void __launch_bounds__(64)
k_pe_pair(DevIndex ix, long n, ReadGeom gm, PeIns pi, int ambiguous_out, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool act = p < n;
    long long maxd = 0, mind = 0; int large_k = 0;
    const long r1 = p, r2 = p + n;
    if (act) {
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        st.job_flag[r1] = 0; st.job_flag[r2] = 0;
        st.red_status[r1] = 0; st.red_status[r2] = 0;
        if (ps.dead[p]) act = false;
    }
    const int n1 = act ? ps.occ[r1] : 0, n2 = act ? ps.occ[r2] : 0;
    const PeCand* a = act ? pe_list(ps, st, A, B, r1) : nullptr;
    const PeCand* b = act ? pe_list(ps, st, A, B, r2) : nullptr;
    int mapping_pair = 0;
    long long bi = 0, bj = 0;
    u32 sbd = 0;
    // the reference's loop itself, one lane
    auto serial = [&](const PeCand* a_, long n1_, const PeCand* b_, long n2_, long long mxd, long long mnd, int lk, int& mp_out, u32& sbd_out,
                      long long& bi_out, long long& bj_out) {
        int mp = 0;
        int best_sum = 4 * lk + 2;
        long long second = (long long)best_sum * 2, bi_ = 0, bj_ = 0;
        bool early = false;
        long first = 0;
        for (long i = 0; i < n1_ && !early; i++) {
            for (long j = first; j < n2_; j++) {
                bool hit = false;
                if (a_[i].site > b_[j].site) {
                    const long long d = (long long)(a_[i].site - b_[j].site);
                    if (d > mxd) first = j + 1;
                    else if (d >= mnd) hit = true;
                } else {
                    const long long d = (long long)(b_[j].site - a_[i].site);
                    if (d > mxd) break;
                    if (d >= mnd) hit = true;
                }
                if (hit) {
                    const long long cur = (long long)a_[i].err + (long long)b_[j].err;
                    if (cur < best_sum) { second = best_sum; best_sum = (int)cur; bi_ = i; bj_ = j; mp = 1; }
                    else if (cur == best_sum) {
                        second = best_sum; mp++;
                        if (best_sum == 0) { early = true; break; }
                    }
                }
            }
        }
        mp_out = mp; bi_out = bi_; bj_out = bj_;
        sbd_out = early ? 0u : (mp != 0 ? (u32)(second - best_sum) : 0u);
    };
    // two long lists (both mates inside a repeat family: hundreds of verified copies each) go to the whole wave: with a lower bound of
    // the distance <= 0 the hits of a[i] are exactly the b[j] within maxd of it, in order -- a lane takes an a[i], finds its window in b
    // by binary search and sums it up, and the lanes' summaries merge in order (pair_comb)
    const bool coop = act && n1 > 0 && n2 > 0 && (long)n1 + n2 > 64 && mind <= 0 && maxd >= 0;
    if (act && !coop && n1 > 0 && n2 > 0) serial(a, n1, b, n2, maxd, mind, large_k, mapping_pair, sbd, bi, bj);
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long pp = (long)__shfl((long long)p, src, 64);
        long long mxd, mnd; int lk;
        pe_bounds(gm, pi, pp, n, mxd, mnd, lk);
        const long m1 = ps.occ[pp], m2 = ps.occ[pp + n];
        const PeCand* a2 = pe_list(ps, st, A, B, pp);
        const PeCand* b2 = pe_list(ps, st, A, B, pp + n);
        if ((a2[m1 - 1].site >> 63) || (b2[m2 - 1].site >> 63)) {     // sites that wrapped below zero: the loop itself decides
            if (lane == src) serial(a2, m1, b2, m2, mxd, mnd, lk, mapping_pair, sbd, bi, bj);
            continue;
        }
        PairSum tot; tot.m = 0; tot.c = 0; tot.pm = 0x7fffffff; tot.i = 0; tot.j = 0;
        for (long base = 0; base < m1; base += 64) {
            const long i = base + lane;
            PairSum me; me.m = 0; me.c = 0; me.pm = 0x7fffffff; me.i = i; me.j = 0;
            if (i < m1) {
                const PeCand e = a2[i];
                const u64 hi = e.site + (u64)mxd;
                for (long j = pe_lower_bound(b2, m2, e.site > (u64)mxd ? e.site - (u64)mxd : 0); j < m2; j++) {
                    const PeCand f = b2[j];
                    if (f.site > hi) break;
                    const int cur = (int)(e.err + f.err);
                    if (me.c == 0 || cur < me.m) { if (me.c) me.pm = me.m < me.pm ? me.m : me.pm; me.m = cur; me.c = 1; me.j = j; }
                    else if (cur == me.m) me.c++;
                }
            }
            // ordered reduction over the lanes: lane l collects lanes l .. l + 2 off - 1
            for (int off = 1; off < 64; off <<= 1) {
                PairSum o;
                o.m = __shfl_down(me.m, off, 64); o.c = __shfl_down(me.c, off, 64); o.pm = __shfl_down(me.pm, off, 64);
                o.i = (long)__shfl_down((long long)me.i, off, 64); o.j = (long)__shfl_down((long long)me.j, off, 64);
                if ((lane & (2 * off - 1)) == 0 && lane + off < 64) me = pair_comb(me, o);
            }
            PairSum ch;
            ch.m = __shfl(me.m, 0, 64); ch.c = __shfl(me.c, 0, 64); ch.pm = __shfl(me.pm, 0, 64);
            ch.i = (long)__shfl((long long)me.i, 0, 64); ch.j = (long)__shfl((long long)me.j, 0, 64);
            tot = pair_comb(tot, ch);
        }
        if (lane == src) {
            const int init = 4 * lk + 2;
            if (tot.c == 0) { mapping_pair = 0; sbd = 0; }
            else {
                bi = tot.i; bj = tot.j;
                if (tot.c >= 2) { mapping_pair = tot.m == 0 ? 2 : tot.c; sbd = 0; }     // met again: second = best (sum 0: the loop stops at the second)
                else { mapping_pair = 1; sbd = (u32)((tot.pm < init ? tot.pm : init) - tot.m); }
            }
        }
    }
    if (!act) return;
    ps.npair[p] = mapping_pair; ps.sbd[p] = sbd;
    if (mapping_pair == 1 || (ambiguous_out && mapping_pair > 1)) {          // Schema.cpp:19342-19347
        st.best_site[r1] = a[bi].site; st.best_end[r1] = a[bi].end; st.best_err[r1] = a[bi].err;
        st.best_site[r2] = b[bj].site; st.best_end[r2] = b[bj].end; st.best_err[r2] = b[bj].err;
        st.red_status[r1] = 1; st.red_status[r2] = 1;
        // a mate that left through the 1-mismatch exit has exactly one mismatch, at mm_site, on the un-gapped diagonal:
        // fast_recalculate_bs_Cigar (ksw.cpp:2578) would only re-derive NM 1 / <L>M / minus one penalty, which k_finalize_pe
        // writes directly -- unless its window leaves the strand, where the reference aligns against an all-zero window
        const long rr[2] = {r1, r2};
        const PeCand w2[2] = {a[bi], b[bj]};
        for (int m = 0; m < 2; m++) {
            const int Lm = gm.rl(rr[m]), km = gm.rk(Lm);
            const bool direct = st.verdict[rr[m]] == 2 && window_valid(ix, w2[m].site, (u64)(Lm + 2 * km), w2[m].site < ix.G);
            st.job_flag[rr[m]] = (w2[m].err != 0 && !direct) ? 1u : 0u;
        }
    }
}

// per-pair post-processing (Schema.cpp:19330-19480): placement of both mates (output_sam_end_to_end_return,
// 9188), TLEN (Schema.h:1587), insert/chromosome-end checks, MAPQ over k1+k2, flags 99/83/147/163, stats
__global__ void __launch_bounds__(256)
k_finalize_pe(DevIndex ix, ScoreParams sp, const int* __restrict__ pen_lut, const char* __restrict__ seq, const char* __restrict__ qual,
              const char* __restrict__ qual2,
              int stride, const u8* __restrict__ mapq_lut, const u32* __restrict__ mapq_off, int unit, ReadGeom gm, int min_ins, int max_ins,
              int ambiguous_out, long n,
              ReadState st, PeState ps, const int* __restrict__ a_start, const int* __restrict__ a_end,
              const u32* __restrict__ a_nm, const int* __restrict__ a_score, const int* __restrict__ a_nops, int max_ops, u32 cigar_base,
              bmbs_result_dev* __restrict__ res, unsigned long long* __restrict__ stats)
{
    __shared__ unsigned long long sh[5];
    __shared__ u64 s_cs[BMBS_CS_LDS];
    if (threadIdx.x < 5) sh[threadIdx.x] = 0;
    const u64* cs = chrom_table(ix, s_cs);
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;          // this lane's contribution to the five counters
    if (p < n) {
        bmbs_result_dev o[2];
        for (int m = 0; m < 2; m++) {
            o[m].pos = 0; o[m].cigar_off = 0; o[m].chrom = -1; o[m].status = 0; o[m].mapq = 0; o[m].flag = 0; o[m].nm = 0;
            o[m].score = 0; o[m].n_cigar = 0; o[m].path = 0; o[m].n_cand = sat16(st.n_cand[p + m * n]); o[m].tlen = 0;
        }
        const int np = ps.dead[p] ? 0 : ps.npair[p];
        int status = 0;
        if (np > 1 && !ambiguous_out) status = 2;
        else if (np >= 1) {
            long long site_pos[2], matched[2];
            int rflag[2], chrom[2], score[2]; u32 nm[2];
            bool inrange = true;
            for (int m = 0; m < 2; m++) {
                const long r = p + m * n;
                const u64 site = st.best_site[r];
                long long start_site, end_site;
                if (st.job_flag[r]) {
                    const u64 jb = st.job_off[r];
                    start_site = a_start[jb]; end_site = a_end[jb]; nm[m] = a_nm[jb]; score[m] = a_score[jb];
                    const int no = a_nops[jb];
                    o[m].cigar_off = cigar_base + (u32)(jb * (u64)max_ops);
                    o[m].n_cigar = no < 0 ? 255 : (u8)no;
                } else {
                    const int Lm = gm.rl(r);
                    end_site = st.best_end[r]; start_site = end_site - Lm + 1; nm[m] = 0; score[m] = 0;
                    if (st.best_err[r] != 0) {
                        // 1-mismatch exit (see k_pe_pair): NM 1, score = minus the penalty at mm_site; mate 2 rows carry their
                        // qualities in FASTQ order for a reverse-complemented read (need_reverse_quality = 1)
                        const int mv = st.mm_site[r], ms = mv & 0x7fff;          // bit 15: the read has 'N' there (k_seed_decide)
                        const int qi = m == 1 ? Lm - 1 - ms : ms;
                        nm[m] = 1;
                        score[m] = (mv & 0x8000) ? -sp.np : -pen_lut[(unsigned char)qual_row(qual, qual2, (u32)n, (u32)r, stride)[qi]];
                    }
                }
                u64 loc = site;
                if (loc >= ix.G) { loc = loc + (u64)end_site; loc = ix.G * 2 - loc - 1; rflag[m] = 16; }
                else { loc = loc + (u64)start_site; rflag[m] = 0; }
                int c = 0;
                c = chrom_of(cs, ix.n_chrom, loc);
                if (c >= ix.n_chrom) { c = ix.n_chrom - 1; inrange = false; }
                chrom[m] = c;
                site_pos[m] = (long long)(loc + 1 - cs[c]);
                matched[m] = end_site - start_site + 1;
                const long long clen = (long long)(cs[c + 1] - cs[c]);
                if ((u64)site_pos[m] + (u64)matched[m] > (u64)clen + 1) inrange = false;
            }
            long long mn = site_pos[0], mx = site_pos[0] + matched[0] - 1;
            if (site_pos[0] > site_pos[1]) mn = site_pos[1];
            if (mx < site_pos[1] + matched[1] - 1) mx = site_pos[1] + matched[1] - 1;
            const int tlen = (int)(mx - mn + 1);
            if (tlen <= max_ins && tlen >= min_ins && inrange) {
                status = np == 1 ? 1 : 2;
                // MAP_Calculation over error_threshold1 + error_threshold2 (Schema.cpp:19445)
                const int L1 = gm.rl(p), L2 = gm.rl(p + n);
                const u32 kk = (u32)(gm.rk(L1) + gm.rk(L2)), sb = ps.sbd[p];
                const int range = unit * (int)kk;
                int sd = score[0] + score[1] + range; if (sd < 0) sd = 0; if (sd > range) sd = range;
                const u32 ed = sb > kk ? kk + 1 : sb;
                const int mapq = mapq_lut[mapq_off[kk] + (size_t)ed * (range + 1) + sd];
                for (int m = 0; m < 2; m++) {
                    o[m].pos = (u64)site_pos[m]; o[m].chrom = chrom[m]; o[m].mapq = (u8)mapq; o[m].nm = (u16)nm[m];
                    o[m].score = (int16_t)score[m]; o[m].tlen = (u32)tlen; o[m].path = 3;
                }
                o[0].flag = (u16)(rflag[0] == 0 ? (1 | 2 | 32 | 64) : (1 | 2 | 16 | 64));
                o[1].flag = (u16)(rflag[1] == 0 ? (1 | 2 | 16 | 128) : (1 | 2 | 32 | 128));
                if (np == 1) s1 = 1;
                s3 = (u32)(L1 + L2);
                s4 = nm[0] + nm[1];
            } else status = 3;
        }
        if (status == 2) s2 = 1;
        s0 = 1;
        o[0].status = (u8)status; o[1].status = (u8)status;
        res[2 * p] = o[0]; res[2 * p + 1] = o[1];
    }
    wave_stats_add(sh, s0, s1, s2, s3, s4);
    __syncthreads();
    if (threadIdx.x < 5 && sh[threadIdx.x]) atomicAdd(&SHARD(stats)[threadIdx.x], sh[threadIdx.x]);
}
