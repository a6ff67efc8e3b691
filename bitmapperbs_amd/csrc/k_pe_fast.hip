// bitmapperbs_amd/csrc/k_pe_fast.hip -- Paired-end fast mode (Map_Pair_Seq_end_to_end_fast, Schema.cpp:18570-19546)
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
// ================================================================================================
// Paired-end fast mode (Map_Pair_Seq_end_to_end_fast, Schema.cpp:18570-19546)
// ================================================================================================
// Reads of a PE batch are rows [0,n) = mate 1 and [n,2n) = mate 2 AS THE REFERENCE'S READER HANDS IT ON:
// reverse complement of the FASTQ record (Process_Reads.cpp:262-267), qualities in FASTQ order.
struct PeCand { u64 site; u32 err; int32_t end; };      // seed_votes fields used by the PE path

// inner_maxDistance_pair / inner_minDistance_pair of pair p (Schema.cpp:18900-18935): the insert bounds widened by
// twice the larger threshold, the lower one also by the longer mate
struct PeIns { int min_ins, max_ins; };
DEVI void pe_bounds(const ReadGeom& gm, const PeIns& pi, long p, long n, long long& maxd, long long& mind, int& large_k)
{
    const int L1 = gm.rl(p), L2 = gm.rl(p + n);
    const int k1 = gm.rk(L1), k2 = gm.rk(L2);
    large_k = k1 > k2 ? k1 : k2;
    maxd = (long long)pi.max_ins + 2LL * large_k;
    mind = (long long)pi.min_ins - 2LL * large_k - (L1 > L2 ? L1 : L2);
}

struct PeState {
    int*  occ;        // per read: best_mapp_occ (>0 verified, -1 to verify, 0 none)
    u32*  len;        // per read: current list length
    u8*   cur;        // per read: 0 list lives in buffer A, 1 in buffer B
    u8*   vround;     // per read: verification round (0 none, 1, 2)
    u8*   dead;       // per pair
    u8*   both;       // per pair: both mates had to be verified
    int*  npair;      // per pair: mapping_pair
    u32*  sbd;        // per pair: second_best_diff
    // --sensitive only
    u8*   first;      // per pair: 0 = mate 1 is finished and verified first, 1 = mate 2
    u8*   full;       // per read: full_seed_id (seeds recorded in read?_seed_start/length)
    PeCand* R;        // lists of re-seeded mates (cur == 2), at roff[read]
    u64*  roff;       // per read
};

// mate 2: reverse complement of the FASTQ read (rc_table, Process_Reads.cpp:1603-1613: identity for non-ACGT)
__global__ void __launch_bounds__(256)
k_pe_prepare(const char* __restrict__ s1, const char* __restrict__ s2raw, ReadGeom gm, int stride, long n, char* __restrict__ seq_all,
             u64* __restrict__ prow, int pwords, int W, u32* __restrict__ dirty32, int sparse_ascii)
{
    // sparse_ascii (the packed rows are what every later kernel reads, --sensitive's re-seeding included): the ASCII text of a 16-byte piece is stored only
    // when the piece holds a character outside ACGT -- the only places the ASCII rows are asked then (is it 'N'?) sit
    // under a set bit of the mask plane, so the other 99.9 % of the 2 x n x stride bytes are never written
    // one 16-byte piece per thread (rows are 16-byte aligned, stride % 16 == 0)
    const long i16 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total16 = n * (stride / 16);
    if (i16 >= total16) return;
    const long total = n * stride;
    const long i = i16 * 16;
    const long r = i / stride;
    const int j0 = (int)(i - r * stride);
    const int L = gm.rl(n + r);                    // mate 2 of pair r
    const uint4 v1 = reinterpret_cast<const uint4*>(s1)[i16];
    if (!sparse_ascii) reinterpret_cast<uint4*>(seq_all)[i16] = v1;                           // the qualities stay where they are (qual_row)
    // the packed copy of both rows (what k_pack_rows would write), from the pieces this thread holds anyway
    auto pack_out = [&](const uint4& pv, long row_id, int Lr, uint4* ascii) {
        const int piece = j0 / 16;
        if (!prow || piece * 16 >= ((Lr + 63) & ~63)) return;
        u32 bases, mask;
        pack_piece(pv, Lr - piece * 16, bases, mask);
        u64* row = prow + (size_t)row_id * pwords;
        if (piece * 16 < ((Lr + 31) & ~31)) reinterpret_cast<u32*>(row)[piece] = bases;
        reinterpret_cast<u16*>(row + W)[piece] = (u16)mask;
        if (mask) {
            atomicOr(&dirty32[row_id >> 2], 1u << (8 * (int)(row_id & 3)));
            if (sparse_ascii) *ascii = pv;
        }
    };
    pack_out(v1, r, gm.rl(r), reinterpret_cast<uint4*>(seq_all) + i16);
    // out[j] = complement(in[L-1-j]) for j < L, 0 beyond: one reversed 16-byte piece per thread.  complement = c ^ 0x15 for
    // A/T, c ^ 0x04 for C/G, identity otherwise (rc_table), eight characters per step.
    auto comp8 = [](u64 w) -> u64 {
        const u64 K7F = 0x7f7f7f7f7f7f7f7full;
        auto zb = [&](u64 x) -> u64 { return ~(((x & K7F) + K7F) | x | K7F); };                 // 0x80 where a byte is 0
        const u64 at = zb(w ^ 0x4141414141414141ull) | zb(w ^ 0x5454545454545454ull);
        const u64 cg = zb(w ^ 0x4343434343434343ull) | zb(w ^ 0x4747474747474747ull);
        return w ^ ((at >> 7) * 0x15) ^ ((cg >> 7) * 0x04);
    };
    const char* row = s2raw + r * stride;
    const int src = L - 16 - j0;                       // in[src .. src+15] reversed = out[j0 .. j0+15]
    uint4 v;
    if (src >= 0) {
        const u64 lo = *reinterpret_cast<const u64*>(row + src), hi = *reinterpret_cast<const u64*>(row + src + 8);
        const u64 a = comp8(__builtin_bswap64(hi)), b2 = comp8(__builtin_bswap64(lo));
        v.x = (u32)a; v.y = (u32)(a >> 32); v.z = (u32)b2; v.w = (u32)(b2 >> 32);
    } else {
        unsigned char o[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int j = j0 + t;
            char c = 0;
            if (j < L) { const char a = row[L - 1 - j]; c = a == 'A' ? 'T' : a == 'T' ? 'A' : a == 'C' ? 'G' : a == 'G' ? 'C' : a; }
            o[t] = (unsigned char)c;
        }
        v.x = o[0] | (o[1] << 8) | (o[2] << 16) | ((u32)o[3] << 24);
        v.y = o[4] | (o[5] << 8) | (o[6] << 16) | ((u32)o[7] << 24);
        v.z = o[8] | (o[9] << 8) | (o[10] << 16) | ((u32)o[11] << 24);
        v.w = o[12] | (o[13] << 8) | (o[14] << 16) | ((u32)o[15] << 24);
    }
    if (!sparse_ascii) reinterpret_cast<uint4*>(seq_all + total)[i16] = v;
    pack_out(v, n + r, L, reinterpret_cast<uint4*>(seq_all + total) + i16);
}

// k_pe_prepare for the packed fast path (sparse ASCII): mate 2 is packed FORWARD first (one aligned 16-byte piece per thread,
// the same SWAR as mate 1), parked in LDS, and the reverse complement is then taken on the packed words -- a 32-bit funnel
// window of the forward row, complemented under its valid-base mask and reversed by 2-bit groups.  The byte-wise complement
// + byte swap of the ASCII form cost ~2.5x the instructions of everything else in the kernel, which was VALU-bound.
// A block takes 256 / (stride / 16) whole pairs; only a piece that holds a character outside ACGT rebuilds its ASCII text.
// The grid is what the chip holds at once and every block walks the pairs in strides: one thread per 16-byte piece per launch made
// 1.6 M waves that lived 5 microseconds each -- alone on the chip that streams at 4 TB/s, but beside another lane's kernel those
// waves queue behind its workgroups (event time with two lanes 1.9 -> 1.55 ms per 10 M pairs; alone 1.2 ms either way).  The pieces
// of the block's next group of pairs are requested before the current group is worked on, and the two LDS buffers alternate so
// that a group costs one barrier.
__global__ void __launch_bounds__(256)
k_pe_prepare_p(const char* __restrict__ s1, const char* __restrict__ s2raw, ReadGeom gm, int stride, long n, char* __restrict__ seq_all,
               u64* __restrict__ prow, int pwords, int W, u32* __restrict__ dirty32)
{
    extern __shared__ u32 lds_pp[];
    const int ppr = stride / 16, rpb = 256 / ppr;
    const int bufw = rpb * (ppr + 1);                   // words of one plane of one buffer
    const int tid = threadIdx.x, rl = tid / ppr, piece = tid - rl * ppr;
    const long total = n * stride;
    const long step = (long)gridDim.x * rpb;
    long r = (long)blockIdx.x * rpb + rl;
    bool on = rl < rpb && r < n;
    uint4 v1 = {0, 0, 0, 0}, v2 = {0, 0, 0, 0};
    if (on) { const size_t i16 = (size_t)r * ppr + piece; v1 = reinterpret_cast<const uint4*>(s1)[i16]; v2 = reinterpret_cast<const uint4*>(s2raw)[i16]; }
    int buf = 0;
    for (long base = (long)blockIdx.x * rpb; base < n; base += step, buf ^= 1) {
        const long rn = r + step;
        const bool onn = rl < rpb && rn < n;
        uint4 n1 = {0, 0, 0, 0}, n2 = {0, 0, 0, 0};
        if (onn) { const size_t j16 = (size_t)rn * ppr + piece; n1 = reinterpret_cast<const uint4*>(s1)[j16]; n2 = reinterpret_cast<const uint4*>(s2raw)[j16]; }
        u32* lb = lds_pp + (size_t)buf * 2 * bufw;      // [rpb][ppr + 1] forward base words (+ one zero word)
        u32* lm = lb + bufw;                            // [rpb][ppr + 1] forward mask pieces (16 bits each, + one zero)
        int L2 = 0;
        if (on) {
            const size_t i16 = (size_t)r * ppr + piece;
            const int L1 = gm.rl(r);
            L2 = gm.rl(n + r);
            u32 b1, m1, b2, m2;
            pack_piece(v1, L1 - piece * 16, b1, m1);
            pack_piece(v2, L2 - piece * 16, b2, m2);
            lb[rl * (ppr + 1) + piece] = b2; lm[rl * (ppr + 1) + piece] = m2;
            if (piece == 0) { lb[rl * (ppr + 1) + ppr] = 0; lm[rl * (ppr + 1) + ppr] = 0; }
            if (piece * 16 < ((L1 + 63) & ~63)) {
                u64* row = prow + (size_t)r * pwords;
                if (piece * 16 < ((L1 + 31) & ~31)) reinterpret_cast<u32*>(row)[piece] = b1;
                reinterpret_cast<u16*>(row + W)[piece] = (u16)m1;
                if (m1) { atomicOr(&dirty32[r >> 2], 1u << (8 * (int)(r & 3))); reinterpret_cast<uint4*>(seq_all)[i16] = v1; }
            }
        }
        __syncthreads();
        if (on && piece * 16 < ((L2 + 63) & ~63)) {
            const int j0 = piece * 16, lo = L2 - 16 - j0;       // forward positions lo .. lo+15, reversed, are rc positions j0 .. j0+15
            const u32* fb = lb + rl * (ppr + 1); const u32* fm = lm + rl * (ppr + 1);
            u32 win, bad, inr;
            if (lo >= 0) {
                const int idx = lo >> 4, sh = lo & 15;
                win = sh ? (fb[idx] >> (2 * sh)) | (fb[idx + 1] << (32 - 2 * sh)) : fb[idx];
                bad = ((fm[idx] | (fm[idx + 1] << 16)) >> sh) & 0xffffu;
                inr = 0xffffu;
            } else if (lo > -16) {
                win = fb[0] << (2 * -lo);
                bad = (fm[0] << -lo) & 0xffffu;
                inr = (0xffffu << -lo) & 0xffffu;
            } else { win = 0; bad = 0; inr = 0; }
            u32 x = inr & ~bad;                                 // real bases of the window -> both bits of their pair
            x = (x | (x << 8)) & 0x00ff00ffu; x = (x | (x << 4)) & 0x0f0f0f0fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
            win ^= x | (x << 1);                                // complement: code -> 3 - code
            u32 rv = __brev(win);
            rv = ((rv >> 1) & 0x55555555u) | ((rv & 0x55555555u) << 1);
            const u32 om = __brev(bad & inr) >> 16;
            const long row_id = n + r;
            u64* row = prow + (size_t)row_id * pwords;
            if (j0 < ((L2 + 31) & ~31)) reinterpret_cast<u32*>(row)[piece] = rv;
            reinterpret_cast<u16*>(row + W)[piece] = (u16)om;
            if (om) {
                atomicOr(&dirty32[row_id >> 2], 1u << (8 * (int)(row_id & 3)));
                const char* src = s2raw + r * stride;
                unsigned char o[16];
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const int j = j0 + t;
                    char c = 0;
                    if (j < L2) { const char a = src[L2 - 1 - j]; c = a == 'A' ? 'T' : a == 'T' ? 'A' : a == 'C' ? 'G' : a == 'G' ? 'C' : a; }
                    o[t] = (unsigned char)c;
                }
                uint4 v;
                v.x = o[0] | (o[1] << 8) | (o[2] << 16) | ((u32)o[3] << 24);
                v.y = o[4] | (o[5] << 8) | (o[6] << 16) | ((u32)o[7] << 24);
                v.z = o[8] | (o[9] << 8) | (o[10] << 16) | ((u32)o[11] << 24);
                v.w = o[12] | (o[13] << 8) | (o[14] << 16) | ((u32)o[15] << 24);
                reinterpret_cast<uint4*>(seq_all + total)[(size_t)r * ppr + piece] = v;
            }
        }
        r = rn; on = onn; v1 = n1; v2 = n2;
    }
}


// get_candidates' list construction (Schema.cpp:18510-18545): site-sorted votes (NOT re-sorted by vote)
__global__ void __launch_bounds__(64)
k_vote_pe(long n2, ReadGeom gm, ReadState st, PeState ps, u64* __restrict__ cand, PeCand* __restrict__ A, u32* __restrict__ slot_read)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n2) return;
    const int L = gm.rl(r), k = gm.rk(L);
    const int v = st.verdict[r];
    const u64 off = st.cand_off[r];
    for (u64 g = off; g < st.cand_off[r + 1]; g++) slot_read[g] = (u32)r;
    ps.cur[r] = 0; ps.vround[r] = 0;
    if (v == 1 || v == 2) {
        A[off].site = st.exit_site[r]; A[off].err = v == 1 ? 0u : 1u; A[off].end = L - 1;
        ps.occ[r] = 1; ps.len[r] = 1;
    } else if (v == 4) {
        const long nc = (long)st.n_cand[r];
        sort_u64_asc(cand + off, nc);
        for (long i = 0; i < nc; i++) { A[off + i].site = cand[off + i]; A[off + i].err = 0; A[off + i].end = L - 1; }
        ps.occ[r] = (int)nc; ps.len[r] = (u32)nc;
    } else if (v == 3) {
        const long nc = (long)st.n_cand[r];
        u64* c = cand + off;
        sort_u64_asc(c, nc);
        PeCand* o = A + off;
        long nv = 0;
        u64 pre = c[0];
        for (long i = 1; i < nc; i++)
            if (c[i] != pre) { o[nv].site = pre < (u64)k ? 0 : pre - (u64)k; o[nv].err = 0; o[nv].end = 0; nv++; pre = c[i]; }
        o[nv].site = pre >= (u64)k ? pre - (u64)k : 0; o[nv].err = 0; o[nv].end = 0; nv++;
        ps.occ[r] = -1; ps.len[r] = (u32)nv;
    } else { ps.occ[r] = 0; ps.len[r] = 0; }
}

// k_locate + k_vote_pe for lists of up to VOTE_REG candidates: located into registers, sorted by the same network as
// k_vote_fused; general reads emit one entry per distinct site (no vote order), exact-ambiguous reads every hit
__global__ void __launch_bounds__(64)
k_vote_pe_fused(DevIndex ix, long n2, ReadGeom gm, ReadState st, PeState ps, u64* __restrict__ cand, PeCand* __restrict__ A,
                u32* __restrict__ slot_read, u32* __restrict__ long_flag, u32* __restrict__ mid_flag)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n2) return;
    long_flag[r] = 0;
    const int L = gm.rl(r), k = gm.rk(L);
    const int v = st.verdict[r];
    const u64 off = st.cand_off[r];
    (void)slot_read;             // nothing downstream of the paired-end vote stage reads the slot -> read map (49 M scattered stores per launch)
    ps.cur[r] = 0; ps.vround[r] = 0;
    if (v == 1 || v == 2) {
        A[off].site = st.exit_site[r]; A[off].err = v == 1 ? 0u : 1u; A[off].end = L - 1;
        ps.occ[r] = 1; ps.len[r] = 1;
        return;
    }
    if (v != 3 && v != 4) { ps.occ[r] = 0; ps.len[r] = 0; return; }
    const long nc = (long)st.n_cand[r];
    const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
    const int ns = st.n_seeds[r];
    PeCand* o = A + off;
    if (nc <= VOTE_REG) {
        u64 c[VOTE_REG];
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
        int nv = 0;
#pragma unroll
        for (int i = 0; i < VOTE_REG; i++) {
            if (i < nc) {
                if (v == 4) { o[i].site = c[i]; o[i].err = 0; o[i].end = L - 1; }
                else if (i + 1 >= nc || (i + 1 < VOTE_REG && c[i + 1] != c[i])) {
                    o[nv].site = c[i] < (u64)k ? 0 : c[i] - (u64)k; o[nv].err = 0; o[nv].end = 0; nv++;
                }
            }
        }
        if (v == 4) { ps.occ[r] = (int)nc; ps.len[r] = (u32)nc; }
        else { ps.occ[r] = -1; ps.len[r] = (u32)nv; }
        return;
    }
    if (mid_flag && nc <= VOTE_MID) { mid_flag[r] = 1; return; }      // 17..32 candidates, the rule for reads of 180 bases and more: k_vote_pe_mid
    long_flag[r] = 1;                                         // repeats: k_vote_pe_long sorts the list out of LDS (beyond its capacity: in tiles)
}

// lists of 17..32 candidates (reads of 180 bases and more place up to 25 seeds): one lane per read over the compacted list,
// located into registers and sorted by a bitonic network -- the paired-end counterpart of k_vote_mid (no vote order here)
__global__ void __launch_bounds__(64)
k_vote_pe_mid(DevIndex ix, ReadGeom gm, ReadState st, PeState ps, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
              PeCand* __restrict__ A, unsigned long long* __restrict__ counters)
{
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = it < (long)*count_ptr;
    const long r = act ? (long)list[it] : 0;
    const long nc = act ? (long)st.n_cand[r] : 0;
    wave_count_add(counters, CNT_CAND_MID, (u32)nc);
    if (!act) return;
    const int L = gm.rl(r), k = gm.rk(L);
    const int v = st.verdict[r];
    const SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
    const int ns = st.n_seeds[r];
    PeCand* o = A + st.cand_off[r];
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
    if (v == 4) {
#pragma unroll
        for (int i = 0; i < VOTE_MID; i++) if (i < nc) { PeCand e; e.site = c[i]; e.err = 0; e.end = L - 1; o[i] = e; }
        ps.occ[r] = (int)nc; ps.len[r] = (u32)nc;
    } else {
        int nv = 0;
#pragma unroll
        for (int i = 0; i < VOTE_MID; i++) {
            if (i < nc && (i + 1 >= nc || (i + 1 < VOTE_MID && c[i + 1] != c[i]))) {
                PeCand e; e.site = c[i] < (u64)k ? 0 : c[i] - (u64)k; e.err = 0; e.end = 0;
                o[nv++] = e;
            }
        }
        ps.occ[r] = -1; ps.len[r] = (u32)nv;
    }
}

// ---- filter_pairs in front of the sort (round 6) ---------------------------------------------------------------------------------------
// A read inside a repeat family locates hundreds to thousands of sites; filter_pairs (Schema.cpp:16052-16164) then keeps the few that
// have a partner on the mate's list within the insert window, and nothing else ever reads the unfiltered list (get_candidates hands it
// straight to filter_pairs, Schema.cpp:19055-19110; fast mode does not order the list by votes).  So when the mate's list is FINAL
// before this read's is sorted -- it came from a kernel that ran earlier (a smaller size class), or from this very block (both mates in
// one class: the one with fewer candidates goes first) -- every located site is tested against it first (a binary search for an entry
// within maxd of the entry the site becomes) and only the survivors are sorted and made distinct.  Exact under the conditions
// k_pe_filter_pairs_long already relies on (lower distance bound <= 0, no site wrapped below zero): an entry survives filter_pairs
// iff the other list holds a site within maxd of it, so dropping the others early changes neither list's filtered form; filter_pairs
// itself still runs afterwards, on the short list.
DEVI long pe_lower_bound_u64(const u64* v, long n, u64 key)
{
    long lo = 0, hi = n;
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (v[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
#define PREF_STAGE 256          // sites are tested against mate lists of up to this many entries (an LDS copy); longer ones: no pre-filter
// keys[0, cnt): located raw sites -> the ones with a partner on `mate` (ascending sites), compacted in place, in order; returns their
// number, or -1 when the list holds a site that wrapped below zero (the reference's mixed comparisons decide there: no pre-filter).
// nm <= PREF_STAGE.
template <int EMAX>
DEVI int pe_prefilter(u64* keys, int cnt, int k, const PeCand* mate, long nm, u64 maxd, u64* sh_mate, int* sh_w)
{
    const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int E = (cnt + T - 1) / T;
    for (int i = tid; i < (int)nm; i += T) sh_mate[i] = mate[i].site;           // nm <= PREF_STAGE (the caller's condition)
    __syncthreads();
    u64 mine[EMAX];
    u32 keep = 0;
    bool wrapped = false;
    int kept = 0;
#pragma unroll
    for (int e = 0; e < EMAX; e++) {
        const int idx = tid * E + e;
        if (e < E && idx < cnt) {
            const u64 c = keys[idx];
            mine[e] = c;
            wrapped |= (c >> 63) != 0;
            const u64 x = c < (u64)k ? 0 : c - (u64)k;
            const u64 lo = x > maxd ? x - maxd : 0;
            const long j = pe_lower_bound_u64(sh_mate, nm, lo);
            const bool ok = j < nm && (sh_mate[j] <= x || sh_mate[j] - x <= maxd);
            if (ok) { keep |= 1u << e; kept++; }
        }
    }
    if (__syncthreads_or(wrapped ? 1 : 0)) return -1;           // (also: every key is in registers)
    int incl = kept;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) sh_w[w] = incl;
    __syncthreads();
    int base = incl - kept, total = 0;
    for (int q = 0; q < (T + 63) / 64; q++) { const int x = sh_w[q]; if (q < w) base += x; total += x; }
#pragma unroll
    for (int e = 0; e < EMAX; e++) if (keep & (1u << e)) keys[base++] = mine[e];
    __syncthreads();
    return total;
}

// the paired-end counterpart of k_vote_long: no vote order here, so everything is parallel (general reads: one entry per
// distinct site; exact-ambiguous reads: every hit)
template <int CAP, int BLOCK, int LO>
__global__ void __launch_bounds__(BLOCK)
k_vote_pe_long(DevIndex ix, ReadGeom gm, ReadState st, PeState ps, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
               PeCand* __restrict__ A, u32* __restrict__ big_list, unsigned long long* __restrict__ big_count, u64* __restrict__ cand,
               unsigned long long* __restrict__ counters, long n_pairs, PeIns pi, int prefilter, const u32* __restrict__ long_flag)
{
    __shared__ u64 keys[CAP];
    __shared__ u16 endpos[CAP];
    __shared__ u32 sh_pref[BMBS_MAX_SEEDS + 1];
    __shared__ int sh_w[2 * (BLOCK / 64) + 1];
    __shared__ u64 sh_mate[PREF_STAGE];
    constexpr int EMAX = (CAP + BLOCK - 1) / BLOCK;
    auto in_class = [](long nc) { return nc > LO && (CAP == VL_CAP || nc <= CAP); };
    // is read x's list built by one of these size-class kernels at all (k_vote_pe_fused / _mid wrote the others before this launch)
    auto listed = [&](long x) { return long_flag[x] != 0; };
    const long total_items = (long)*count_ptr;
    for (long item = blockIdx.x; item < total_items; item += gridDim.x) {
        const long r0 = list[item];
        const long nc0 = (long)st.n_cand[r0];
        // the wave form sees every listed read and passes the ones beyond its capacity on (a list of their own: the block form used
        // to walk the whole list -- millions of reads on a repeat-rich genome, two dependent loads each -- to find its few)
        if (big_list && nc0 > CAP) { if (threadIdx.x == 0) big_list[atomicAdd(big_count, 1ull)] = (u32)r0; continue; }
        if (!in_class(nc0)) continue;                               // another instance's size class (the largest also takes what is beyond it)
        // both mates in this class: ONE block takes them, the one with fewer candidates first (ties: mate 1), so that the second is
        // filtered against a finished list
        const long m0 = r0 < n_pairs ? r0 + n_pairs : r0 - n_pairs;
        const bool mate_here = prefilter && listed(m0) && in_class((long)st.n_cand[m0]);
        if (mate_here) {
            const long ncm = (long)st.n_cand[m0];
            if (ncm < nc0 || (ncm == nc0 && m0 < r0)) continue;    // the mate's block does both
        }
        for (int turn = 0; turn < (mate_here ? 2 : 1); turn++) {
            const long r = turn ? m0 : r0;
            const long m = turn ? r0 : m0;
            const long nc = (long)st.n_cand[r];
            if (counters && threadIdx.x == 0) {
                atomicAdd(&SHARD(counters)[CAP == VM_CAP ? CNT_CAND_LONG : CNT_CAND_BIG], (unsigned long long)nc);
                atomicAdd(&SHARD(counters)[CNT_LISTS_LONG], 1ull);
            }
            const int L = gm.rl(r), k = gm.rk(L);
            const int v = st.verdict[r];
            PeCand* o = A + st.cand_off[r];
            if (nc > CAP) {
                // beyond the LDS capacity: sorted in tiles (vl_sort_huge; the output segment parks the tiles), then the same entries in order
                u64* c = cand + st.cand_off[r];
                vl_sort_huge<CAP, BLOCK>(ix, st.seeds + (size_t)r * BMBS_MAX_SEEDS, st.n_seeds[r], nc, keys, sh_pref, reinterpret_cast<u64*>(o), c);
                if (v == 4) {
                    for (long i = threadIdx.x; i < nc; i += BLOCK) { PeCand e; e.site = c[i]; e.err = 0; e.end = L - 1; o[i] = e; }
                    if (threadIdx.x == 0) { ps.occ[r] = (int)nc; ps.len[r] = (u32)nc; }
                } else {
                    int running = 0;
                    for (long base = 0; base < nc; base += BLOCK) {
                        const long i = base + (long)threadIdx.x;
                        bool keep = false;
                        u64 key = 0;
                        if (i < nc) { key = c[i]; keep = i == nc - 1 || c[i + 1] != key; }
                        int tot;
                        const int pre = vl_prefix(keep, sh_w, tot);
                        if (keep) { PeCand e; e.site = key < (u64)k ? 0 : key - (u64)k; e.err = 0; e.end = 0; o[running + pre] = e; }
                        running += tot;
                    }
                    if (threadIdx.x == 0) { ps.occ[r] = -1; ps.len[r] = (u32)running; }
                }
                if (mate_here && turn == 0) __threadfence();   // (the second turn reads this list from memory)
                __syncthreads();
                continue;
            }
            vl_locate_range(ix, st.seeds + (size_t)r * BMBS_MAX_SEEDS, st.n_seeds[r], 0, (int)nc, keys, sh_pref, true);
            int cnt = (int)nc;
            // the mate's list is final when an earlier kernel wrote it (verdicts 1 / 2 / none, lists of a smaller class) or this block
            // just did (turn 1)
            if (prefilter && v == 3) {
                // (only against a mate list short enough for the LDS copy: a binary search per site through a list in memory -- both
                // mates inside repeats -- costs more than the sort it saves; measured, round 6)
                // (a listed mate of a smaller class: its kernel ran earlier in the stream -- the instances are launched one behind the other)
                const bool earlier = !listed(m) || (long)st.n_cand[m] <= LO;
                const bool mate_final = (turn == 1 || earlier) && ps.len[m] <= PREF_STAGE;
                long long maxd, mind; int large_k;
                pe_bounds(gm, pi, r < n_pairs ? r : r - n_pairs, n_pairs, maxd, mind, large_k);
                if (mate_final && mind <= 0 && maxd >= 0) {
                    const int occm = ps.occ[m];
                    const long nm = occm == 0 ? 0 : (long)ps.len[m];
                    const PeCand* ml = A + st.cand_off[m];
                    if (nm == 0) cnt = 0;                           // the pair is dead whatever this list holds (Schema.cpp:19084-19090)
                    else if (!(ml[nm - 1].site >> 63)) {
                        const int kept = pe_prefilter<EMAX>(keys, cnt, k, ml, nm, (u64)maxd, sh_mate, sh_w);
                        if (kept >= 0) cnt = kept;
                    }
                    if (cnt < (int)nc && counters && threadIdx.x == 0) atomicAdd(&SHARD(counters)[CNT_PREFILTER_DROP], (unsigned long long)(nc - cnt));
                }
            }
            if (cnt == 0) {
                if (threadIdx.x == 0) { ps.occ[r] = -1; ps.len[r] = 0; }
                __syncthreads();
                continue;
            }
            vl_sort_keys<EMAX>(keys, cnt);
            if (v == 4) {
                for (long i = threadIdx.x; i < cnt; i += BLOCK) { PeCand e; e.site = keys[i]; e.err = 0; e.end = L - 1; o[i] = e; }
                if (threadIdx.x == 0) { ps.occ[r] = (int)cnt; ps.len[r] = (u32)cnt; }
            } else {
                const int nv = vl_run_ends(keys, cnt, endpos, sh_w);
                for (int e2 = threadIdx.x; e2 < nv; e2 += BLOCK) {
                    const u64 site = keys[endpos[e2]];
                    PeCand e; e.site = site < (u64)k ? 0 : site - (u64)k; e.err = 0; e.end = 0;
                    o[e2] = e;
                }
                if (threadIdx.x == 0) { ps.occ[r] = -1; ps.len[r] = (u32)nv; }
            }
            if (mate_here && turn == 0) __threadfence();       // (an agent-scope fence is microseconds: only where a second turn follows)
            __syncthreads();
        }
    }
}

DEVI PeCand* pe_list(const PeState& ps, const ReadState& st, PeCand* A, PeCand* B, long r)
{
    const int cur = ps.cur[r];
    return cur == 2 ? ps.R + ps.roff[r] : (cur ? B : A) + st.cand_off[r];
}

// filter_pairs (Schema.cpp:16052-16180) + the driver's choice of what to verify (19050-19290)
// what follows the two filtered lists (Schema.cpp:19050-19290): who is verified in which round
DEVI void pe_filter_decide(const PeState& ps, long p, long r1, long r2, int occ1, int occ2, long la, long lb)
{
    ps.cur[r1] = 1; ps.cur[r2] = 1;
    ps.len[r1] = (u32)la; ps.len[r2] = (u32)lb;
    if (la == 0 || lb == 0) { ps.dead[p] = 1; return; }
    if (occ1 == -1 && occ2 == -1) {
        ps.both[p] = 1;
        if (la <= lb) { ps.vround[r1] = 1; ps.vround[r2] = 2; } else { ps.vround[r2] = 1; ps.vround[r1] = 2; }
    } else if (occ1 != -1) {
        if (la < occ1) ps.occ[r1] = (int)la;
        ps.vround[r2] = 1;
    } else {
        if (lb < occ2) ps.occ[r2] = (int)lb;
        ps.vround[r1] = 1;
    }
}
// the reference's merge loop itself, one lane
DEVI void pe_filter_serial(const PeCand* a, long na, const PeCand* b, long nb, long long maxd, long long mind, PeCand* ra, PeCand* rb, long& la_out, long& lb_out)
{
    long la = 0, lb = 0, first = 0;
    for (long i = 0; i < na; i++) {
        for (long j = first; j < nb; j++) {
            bool hit = false;
            if (a[i].site > b[j].site) {
                const long long d = (long long)(a[i].site - b[j].site);
                if (d > maxd) first = j + 1;
                else if (d >= mind) hit = true;
            } else {
                const long long d = (long long)(b[j].site - a[i].site);
                if (d > maxd) break;
                if (d >= mind) hit = true;
            }
            if (hit) {
                if (la == 0 || a[i].site > ra[la - 1].site) ra[la++] = a[i];
                if (lb == 0 || b[j].site > rb[lb - 1].site) rb[lb++] = b[j];
            }
        }
    }
    la_out = la; lb_out = lb;
}
#define PEF_LONG 24       // pairs whose two lists hold more candidates than this go to k_pe_filter_pairs_long (one wave per pair)
__global__ void __launch_bounds__(64)
k_pe_filter_pairs(long n, ReadGeom gm, PeIns pi, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B, u32* __restrict__ long_flag,
                  unsigned long long* __restrict__ counters)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r1 = p, r2 = p + n;
    int occ1 = 1, occ2 = 1;
    if (p < n) {
        occ1 = ps.occ[r1]; occ2 = ps.occ[r2];
        ps.dead[p] = 0; ps.both[p] = 0; ps.npair[p] = 0; ps.sbd[p] = 0;
    }
    const bool runs = p < n && !(occ1 > 0 && occ2 > 0) && occ1 != 0 && occ2 != 0;       // filter_pairs is reached (Schema.cpp:19084-19110)
    const long na = runs ? (long)ps.len[r1] : 0, nb = runs ? (long)ps.len[r2] : 0;
    wave_count_add(counters, CNT_PEF_ENTRIES, (u32)(na + nb));
    if (p >= n) return;
    long long maxd, mind; int large_k;
    pe_bounds(gm, pi, p, n, maxd, mind, large_k);
    if (occ1 > 0 && occ2 > 0) return;
    if (occ1 == 0 || occ2 == 0) { ps.dead[p] = 1; return; }
    // in a repeat-rich genome one pair in a few has lists of dozens to thousands of candidates: a lane that walks them alone holds its
    // wave for as long (k_pe_filter_pairs: 0.46 ms per 10 M pairs on the uniform genome, 12.8 ms on the GRCh38-like one)
    if (long_flag && na + nb > PEF_LONG) { long_flag[p] = 1; return; }
    long la, lb;
    pe_filter_serial(A + st.cand_off[r1], na, A + st.cand_off[r2], nb, maxd, mind, B + st.cand_off[r1], B + st.cand_off[r2], la, lb);
    pe_filter_decide(ps, p, r1, r2, occ1, occ2, la, lb);
}

// One wave per pair with long lists.  With mind <= 0 (the default --min 0 makes it negative) a pair of sites hits iff they lie
// within maxd of each other, so: an entry of one list survives iff the other list holds a site within maxd of it (a binary search
// over the sorted other list), minus entries whose site repeats the one before (the reference pushes a[i] / b[j] only when its
// site is larger than the last one pushed).  Every lane takes entries of its own; ballots compact the survivors in order.
// mind > 0 (a minimum insert larger than the read + 2k): the hit rule is no longer an interval and the order of discovery
// matters -- lane 0 runs the reference's loop.
DEVI long pe_lower_bound(const PeCand* v, long n, u64 key)          // first index with site >= key
{
    long lo = 0, hi = n;
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (v[mid].site < key) lo = mid + 1; else hi = mid; }
    return lo;
}
DEVI long pe_filter_side(const PeCand* x, long nx, const PeCand* y, long ny, u64 maxd, PeCand* out)
{
    const int lane = threadIdx.x & 63;
    long w = 0;
    for (long base = 0; base < nx; base += 64) {
        const long i = base + lane;
        bool keep = false;
        PeCand e; e.site = 0; e.err = 0; e.end = 0;
        if (i < nx) {
            e = x[i];
            const long j = pe_lower_bound(y, ny, e.site > maxd ? e.site - maxd : 0);        // the first site of y not more than maxd below e
            keep = j < ny && (y[j].site <= e.site || y[j].site - e.site <= maxd);            // ... is it also not more than maxd above?
            if (keep && i > 0 && x[i - 1].site == e.site) keep = false;                      // a repeated site is pushed once
        }
        const unsigned long long m = __ballot(keep);
        if (keep) out[w + __popcll(m & ((1ull << lane) - 1))] = e;
        w += __popcll(m);
    }
    return w;
}
__global__ void __launch_bounds__(64)
k_pe_filter_pairs_long(long n, ReadGeom gm, PeIns pi, ReadState st, PeState ps, const u64* __restrict__ count_ptr, const u32* __restrict__ list,
                       PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long total = (long)*count_ptr;
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const long p = list[item];
        long long maxd, mind; int large_k;
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        const long r1 = p, r2 = p + n;
        const int occ1 = ps.occ[r1], occ2 = ps.occ[r2];
        const PeCand* a = A + st.cand_off[r1];
        const PeCand* b = A + st.cand_off[r2];
        PeCand* ra = B + st.cand_off[r1];
        PeCand* rb = B + st.cand_off[r2];
        const long na = ps.len[r1], nb = ps.len[r2];
        long la = 0, lb = 0;
        // sites that wrapped around below zero (a seed at the very start of the text) sort last as huge unsigned values and take the
        // reference's mixed unsigned / signed comparisons: those pairs keep its loop
        const bool wrapped = (na && (a[na - 1].site >> 63)) || (nb && (b[nb - 1].site >> 63));
        if (mind <= 0 && maxd >= 0 && !wrapped) {
            la = pe_filter_side(a, na, b, nb, (u64)maxd, ra);
            lb = pe_filter_side(b, nb, a, na, (u64)maxd, rb);
        } else {
            if ((threadIdx.x & 63) == 0) pe_filter_serial(a, na, b, nb, maxd, mind, ra, rb, la, lb);
            la = __shfl((int)la, 0, 64); lb = __shfl((int)lb, 0, 64);
        }
        if ((threadIdx.x & 63) == 0) pe_filter_decide(ps, p, r1, r2, occ1, occ2, la, lb);
    }
}

// verify_candidate_locations' Myers pass for the mates scheduled in `round`: a dense work list
// (read, list index) is built by count -> scan -> scatter so that the filter runs on full waves
__global__ void __launch_bounds__(256)
k_pe_count(long n, long n2, int round, PeState ps, u32* __restrict__ cnt)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n2) return;
    const long p = r < n ? r : r - n;
    cnt[r] = (ps.vround[r] == round && !ps.dead[p]) ? ps.len[r] : 0u;
}
__global__ void __launch_bounds__(256)
k_pe_worklist(long n2, const u32* __restrict__ cnt, const u64* __restrict__ off, u32* __restrict__ work_r, u32* __restrict__ work_i)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n2) return;
    const u32 m = cnt[r];
    const u64 o = off[r];
    for (u32 i = 0; i < m; i++) { work_r[o + i] = (u32)r; work_i[o + i] = i; }
}
__global__ void __launch_bounds__(256)
k_filter_pe(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, ReadState st, PeState ps,
            PeCand* __restrict__ A, PeCand* __restrict__ B, const u64* __restrict__ n_work, const u32* __restrict__ work_r,
            const u32* __restrict__ work_i, unsigned long long* __restrict__ counters)
{
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= *n_work) return;
    const long r = (long)work_r[g];
    PeCand* e = pe_list(ps, st, A, B, r) + work_i[g];
    u32 er; int es;
    const int L = gm.rl(r), k = gm.rk(L);
    bpm_read(ix, seq, stride, pr, r, L, k, e->site, er, es);
    e->err = er; e->end = es;
    if (counters) atomicAdd(&SHARD(counters)[3], 1ull);
}

// the PE compaction (Schema.cpp:7480-7690): keep err <= k whose site+end differs from the previous candidate's
// A lane walks its read's list; a list of more than 64 entries (a read inside a repeat family: hundreds to thousands) is walked by
// the whole wave afterwards, 64 entries a step -- one such lane used to hold its wave for the length of its list.
__global__ void __launch_bounds__(64)
k_pe_compact(long n, long n2, ReadGeom gm, int round, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool act = r < n2 && ps.vround[r] == round;
    if (act) { const long p = r < n ? r : r - n; if (ps.dead[p]) act = false; }
    const long m = act ? (long)ps.len[r] : 0;
    const bool coop = m > 64;
    if (act && !coop) {
        const int k = gm.rk(gm.rl(r));
        PeCand* l = pe_list(ps, st, A, B, r);
        u64 pre = ~0ull;
        int occ = 0;
        for (long i = 0; i < m; i++) {
            const PeCand c = l[i];
            const u64 t = c.site + (u64)(long long)c.end;
            if (c.err <= (u32)k && pre != t) { l[occ] = c; occ++; }
            pre = t;
        }
        ps.occ[r] = occ;
    }
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long rr = (long)__shfl((long long)r, src, 64);
        const long mm = (long)ps.len[rr];
        const int k = gm.rk(gm.rl(rr));
        PeCand* l = pe_list(ps, st, A, B, rr);
        u64 carry = ~0ull;                              // site + end of the entry before the step's first
        int occ = 0;
        for (long base = 0; base < mm; base += 64) {
            const long i = base + lane;
            PeCand c; c.site = 0; c.err = 0; c.end = 0;
            u64 t = 0;
            if (i < mm) { c = l[i]; t = c.site + (u64)(long long)c.end; }
            u64 pre = (u64)__shfl_up((long long)t, 1, 64);
            if (lane == 0) pre = carry;
            const bool keep = i < mm && c.err <= (u32)k && pre != t;
            const unsigned long long kb = __ballot(keep);       // every entry of the step is in registers before the first is stored
            if (keep) l[occ + __popcll(kb & ((1ull << lane) - 1))] = c;
            occ += __popcll(kb);
            carry = (u64)__shfl((long long)t, 63, 64);
        }
        if (lane == src) ps.occ[rr] = occ;
    }
}

// after round 1 of a both-unverified pair: filter_pairs_single_side (Schema.cpp:16186-16270)
// The reference's merge loop, one lane per pair -- and, for a pair whose two lists hold more than 64 entries, by the whole wave:
// with a lower bound of the distance <= 0 (the default insert range) the loop keeps b[j] exactly when some a[i] lies within maxd of
// it (it drops b[j] only when the current a[i] is more than maxd above it, and then so is every later one; it leaves the scan of
// a[i] only at a b[j] more than maxd above a[i], and then so is every later one), so every b[j] is decided by one binary search in a.
__global__ void __launch_bounds__(64)
k_pe_prune(long n, ReadGeom gm, PeIns pi, ReadState st, PeState ps, PeCand* __restrict__ A, PeCand* __restrict__ B)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool act = p < n && !ps.dead[p] && ps.both[p];
    long long maxd = 0, mind = 0; int large_k;
    long rs = 0, ro = 0;
    int occ_s = 0;
    if (act) {
        pe_bounds(gm, pi, p, n, maxd, mind, large_k);
        rs = ps.vround[p] == 1 ? p : p + n;       // verified side
        ro = ps.vround[p] == 1 ? p + n : p;       // side still to verify
        occ_s = ps.occ[rs];
        if (occ_s == 0) { ps.dead[p] = 1; act = false; }
    }
    const long nb = act ? (long)ps.len[ro] : 0;
    auto serial = [&](const PeCand* a, long na, PeCand* b, long nbb, long long mxd, long long mnd) -> long {
        long len2 = 0, first = 0;
        for (long i = 0; i < na; i++) {
            for (long j = first; j < nbb; j++) {
                if (a[i].site > b[j].site) {
                    const long long d = (long long)(a[i].site - b[j].site);
                    if (d > mxd) first = j + 1;
                    else if (d >= mnd) { b[len2] = b[j]; len2++; first = j + 1; }
                } else {
                    const long long d = (long long)(b[j].site - a[i].site);
                    if (d > mxd) break;
                    if (d >= mnd) { b[len2] = b[j]; len2++; first = j + 1; }
                }
            }
        }
        return len2;
    };
    const bool coop = act && nb + occ_s > 64 && mind <= 0 && maxd >= 0;
    if (act && !coop) ps.len[ro] = (u32)serial(pe_list(ps, st, A, B, rs), occ_s, pe_list(ps, st, A, B, ro), nb, maxd, mind);
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long pp = (long)__shfl((long long)p, src, 64);
        long long mxd, mnd; int lk;
        pe_bounds(gm, pi, pp, n, mxd, mnd, lk);
        const long rs2 = ps.vround[pp] == 1 ? pp : pp + n, ro2 = ps.vround[pp] == 1 ? pp + n : pp;
        const long na = ps.occ[rs2], nbb = (long)ps.len[ro2];
        const PeCand* a = pe_list(ps, st, A, B, rs2);
        PeCand* b = pe_list(ps, st, A, B, ro2);
        // sites that wrapped below zero (the last ones of an ascending list) make the distances negative: the loop itself decides
        if ((a[na - 1].site >> 63) || (nbb && (b[nbb - 1].site >> 63))) {
            if (lane == src) ps.len[ro2] = (u32)serial(a, na, b, nbb, mxd, mnd);
            continue;
        }
        long len2 = 0;
        for (long base = 0; base < nbb; base += 64) {
            const long j = base + lane;
            PeCand e; e.site = 0; e.err = 0; e.end = 0;
            bool keep = false;
            if (j < nbb) {
                e = b[j];
                const long i = pe_lower_bound(a, na, e.site > (u64)mxd ? e.site - (u64)mxd : 0);      // the first a not more than maxd below b[j]
                keep = i < na && (a[i].site <= e.site || a[i].site - e.site <= (u64)mxd);             // ... and not more than maxd above
            }
            const unsigned long long kb = __ballot(keep);
            if (keep) b[len2 + __popcll(kb & ((1ull << lane) - 1))] = e;
            len2 += __popcll(kb);
        }
        if (lane == src) ps.len[ro2] = (u32)len2;
    }
}
