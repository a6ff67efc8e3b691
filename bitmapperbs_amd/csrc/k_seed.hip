// bitmapperbs_amd/csrc/k_seed.hip -- K1-K5: seeding
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
// ================================================================================================
// K1-K5: seeding
// ================================================================================================
// The seeding state machine of Map_Single_Seq_end_to_end (Schema.cpp:24588-24898) / get_candidates
// (18172-18565) is data dependent per read: ~half of the reads leave after one short seed (exact-unique
// exit), a third need one long second seed (1-mismatch path), the rest run up to 25 more seeds.  One
// read per lane in ONE kernel makes every wave as slow as its slowest read.  It is therefore split
// by read class, with scan-compacted work lists in between (no host round-trip):
//   k_seed_first   all reads          first seed (count_backward_as_much_1_terminate)
//   k_seed_decide  all reads          exact-unique / exact-ambiguous exits, 1-mismatch detection
//   k_seed_second  1-mismatch reads   second seed (count_hash_table) + 1-mismatch exit
//   k_seed_extra   everything else    the remaining seeds
// The three search kernels share one engine: a wave owns a contiguous chunk of its work list and
// every lane that finishes an item immediately takes the next one of the chunk (wave-local counter,
// no global atomics), so the lanes of a wave keep stepping in lock-step through the LF loop -- the
// only hot code -- whatever the individual seed lengths are.
struct SeedHit { u64 hits, sp, ml; };

// search state of one lane: count_backward_as_much_1_terminate (bwt.h:2081-2209) or count_hash_table
// (bwt.h:1848-1952) over read[tm, L), advanced one backward-extension at a time.  In read coordinates
// the pattern bsSeq[0, L-tm) = reverse(read[tm, L)) with C->T, so the 16-mer key is the little-endian
// base-3 number of read[tm .. tm+15] and extension consumes read[tm+16], read[tm+17], ....
struct Search { u64 top, bot, ptop, pbot; int s, steps, tm; ReadCur cur; };

// eight ASCII characters -> eight 3-letter digits (G0 T1 A2, C folded into T), one per byte.  (c >> 1) & 3 is A0 C1 T2 G3, and
// the digit is 2 - popcount of that.  `bad` is non-zero in every byte that does not hold one of A C G T: the letter a byte would
// have to be is rebuilt from its bits 1-2 (0x41 | bits 1-2; T: ^ 0x11) and compared with it.
DEVI void swar_code3(u64 w, u64& digits, u64& bad)
{
    const u64 K01 = 0x0101010101010101ull;
    digits = 0x0202020202020202ull - ((w >> 1) & K01) - ((w >> 2) & K01);
    const u64 isT = (w >> 2) & ~(w >> 1) & K01;
    bad = w ^ (0x4141414141414141ull | (w & 0x0606060606060606ull)) ^ (isT | (isT << 4));
}
// four digits (one per byte of x) -> d0 + 3 d1 + 9 d2 + 27 d3 in one multiply
DEVI u32 base3_of4(u32 x) { return (x * 0x0103091Bu) >> 24; }

// returns true when the search has to be stepped; false when it is already decided (out filled)
// LOCATED (with FIXED): the caller finishes single-row intervals against the genome and takes them with the text position
// in S.top (bit 63 set) -- k_seed_second; without it the full-length search does not touch the 20-mer table.
template <bool FIXED, bool LOCATED = false>
DEVI bool search_begin(const DevIndex& ix, const char* rd, int L, int tm, Search& S, SeedHit& out, u32& n_hash)
{
    const int len = L - tm;
    out.hits = 0; out.sp = 0; out.ml = FIXED ? (u64)len : 0;
    if (len < (FIXED ? 17 : 18)) return false;
    // key = sum code(read[tm+u]) * 3^u over the 16 characters read[tm .. tm+15].  Every per-lane load is a request of its
    // own to the memory pipeline whether it hits or not (tools/gather_bench.hip: a row load costs 3/4 of a random gather), so the
    // 20 characters of the key and the cursor's first piece come from two aligned 16-byte loads (a third one when tm sits in
    // the last quarter of its piece) instead of three unaligned loads plus the cursor's own.
    const int a16 = tm & ~15, o = tm & 15;
    const uint4 B0 = *reinterpret_cast<const uint4*>(rd + a16), B1 = *reinterpret_cast<const uint4*>(rd + a16 + 16);
    uint4 B2 = make_uint4(0, 0, 0, 0);
    if (o + 16 + ix.t_e >= 32 && a16 + 32 < L) B2 = *reinterpret_cast<const uint4*>(rd + a16 + 32);
    const u64 q0 = ((u64)B0.y << 32) | B0.x, q1 = ((u64)B0.w << 32) | B0.z, q2 = ((u64)B1.y << 32) | B1.x,
              q3 = ((u64)B1.w << 32) | B1.z, q4 = ((u64)B2.y << 32) | B2.x;
    const int sh8 = (o & 7) * 8;
    auto funnel = [&](u64 lo, u64 hi) -> u64 { return sh8 ? (lo >> sh8) | (hi << (64 - sh8)) : lo; };
    const bool up = o >= 8;
    const u64 c0 = up ? q1 : q0, c1 = up ? q2 : q1, c2 = up ? q3 : q2, c3 = up ? q4 : q3;
    const u64 w0 = funnel(c0, c1), w1 = funnel(c1, c2);          // read[tm .. tm+7], read[tm+8 .. tm+15]
    u64 d0, v0, d1, v1;
    swar_code3(w0, d0, v0);
    swar_code3(w1, d1, v1);
    if ((v0 | v1) != 0) return false;                              // get_3_letter_hash_value returned -1 (bwt.h:309-332)
    // < 3^16: 32-bit arithmetic
    const u64 key = base3_of4((u32)d0) + 81u * base3_of4((u32)(d0 >> 32)) + 6561u * base3_of4((u32)d1) + 531441u * base3_of4((u32)(d1 >> 32));
    S.steps = len - 16; S.tm = tm;
    const int E = ix.t_e;
    if ((!FIXED || LOCATED) && ix.t20 && len >= 16 + E) {
        // the 16-mer lookup and the first E extensions in one table read
        u64 d2, v2;
        const u64 cmask = E == 5 ? 0xffffffffffull : 0xffffffffull;
        swar_code3(funnel(c2, c3) & cmask, d2, v2);                  // read[tm+16 .. tm+15+E]
        if ((v2 & cmask) == 0) {
            const u64 code = (u64)base3_of4((u32)d2) + (E == 5 ? 81ull * ((d2 >> 32) & 0xffull) : 0ull);
            const u64 v = ix.t20[key * t20_width(E) + code];
            const int tag = (int)(v >> 60);
            if (tag != 14) {
                n_hash++;
                const u64 row = v & ((1ull << 36) - 1), hits = (v >> 36) & ((1ull << 24) - 1);
                if (tag == 15) return false;                                           // hits 0, match length 0
                if (FIXED) {
                    // count_hash_table goes through the whole pattern: a missing letter is 0 hits; a single row carries on --
                    // the caller finishes it against the genome (k_seed_second), from the text position the table holds
                    if (tag > E && tag <= 2 * E) return false;
                    if (tag != 0) {
                        S.top = row | (1ull << 63); S.bot = S.top + 1; S.ptop = ~0ull; S.pbot = ~0ull; S.s = tag == 2 * E + 1 ? E : tag - 1;
                        return true;
                    }
                } else {
                    if (tag >= 1 && tag <= E) { out.ml = (u64)(15 + tag); out.sp = row | (1ull << 63); out.hits = 1; return false; }     // located (bit 63): text position
                    if (tag == 2 * E + 1) { out.ml = (u64)(16 + E); out.sp = row | (1ull << 63); out.hits = 1; return false; }
                    if (tag > E) { out.ml = (u64)(15 - E + tag); out.sp = row; out.hits = hits; return false; }
                }
                S.top = row; S.bot = row + hits; S.ptop = ~0ull; S.pbot = ~0ull; S.s = E;
                if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = hits; return false; }
                S.cur.seek_with(rd, tm + 16 + E, L, o + 16 + E < 32 ? B1 : B2);
                return true;
            }
        }
        // a letter outside the alphabet among them, or an oversized interval: the 16-mer path
    }
    hash_lookup(ix, key, S.top, S.bot);
    n_hash++;
    if (S.bot <= S.top) return false;
    S.ptop = ~0ull; S.pbot = ~0ull; S.s = 0;
    S.cur.seek_with(rd, tm + 16, L, B1);
    return true;
}

// one loop iteration of the reference; returns true when the search is finished (out filled)
template <bool FIXED>
DEVI bool search_step(const DevIndex& ix, const char* rd, int L, Search& S, SeedHit& out, u32& n_ext)
{
    const int len = L - S.tm;
    if (!FIXED) {
        S.ptop = S.top; S.pbot = S.bot;
        if (S.bot - S.top == 1) { out.ml = 16 + S.s; out.sp = S.top; out.hits = 1; return true; }
        const int d = code3(S.cur.next());          // read[tm + 16 + s]
        if (d > 2) { out.ml = 16 + S.s; out.sp = S.ptop; out.hits = S.pbot - S.ptop; return true; }
        lf_pair(ix, S.top, S.bot, d);
        n_ext++;
        if (S.bot <= S.top) { out.ml = 16 + S.s; out.sp = S.ptop; out.hits = S.pbot - S.ptop; return true; }
        S.s++;
        if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = S.bot - S.top; return true; }
        return false;
    } else {
        out.ml = (u64)len;
        const int d = code3(S.cur.next());          // read[tm + 16 + s]
        if (d > 2) { out.hits = 0; out.sp = 0; return true; }
        lf_pair(ix, S.top, S.bot, d);
        n_ext++;
        if (S.bot <= S.top) { out.hits = 0; out.sp = S.top; return true; }      // the remaining iterations only break
        S.s++;
        if (S.s == S.steps) { out.sp = S.top; out.hits = S.bot - S.top; return true; }
        return false;
    }
}

// ---- the same two functions over a packed row (PackedRows) ----------------------------------------------------------------
struct SearchP { u64 top, bot; int s, steps, tm, kg; PCur cur; };      // kg: steps in a row that kept most of the interval (< 0: three-letter steps are off for this seed)

// four 2-bit digits (d0 in bits 0-1) -> d0 + 3 d1 + 9 d2 + 27 d3
DEVI u32 base3_of4x2(u32 v8)
{
    u32 t = (v8 | (v8 << 12)) & 0x000f000fu;
    t = (t | (t << 6)) & 0x03030303u;
    return base3_of4(t);
}

template <bool FIXED, bool LOCATED = false>
DEVI bool search_begin_p(const DevIndex& ix, const u64* row, int W, bool dirty, int L, int tm, SearchP& S, SeedHit& out, u32& n_hash)
{
    const int len = L - tm;
    out.hits = 0; out.sp = 0; out.ml = FIXED ? (u64)len : 0;
    if (len < (FIXED ? 17 : 18)) return false;
    // the 64 bases from the word that holds read[tm]: the 16 of the key, the table's look-ahead and the cursor's first piece in
    // ONE 16-byte request
    const u64 x = prow_bases32(row, tm);                              // bases tm .. tm+31
    const u32 mbits = dirty ? prow_mask32(row, W, tm) : 0u;
    if (mbits & 0xffffu) return false;                               // get_3_letter_hash_value returned -1 (bwt.h:309-332)
    // 3-letter digits in the 2-bit fields: A (00) -> 2, C / T (low bit) -> 1, G -> 0
    const u64 D = ((((~x) & ((~x) >> 1)) & PK_EVEN) << 1) | (x & PK_EVEN);
    const u32 d32 = (u32)D;
    const u64 key = base3_of4x2(d32 & 0xffu) + 81u * base3_of4x2((d32 >> 8) & 0xffu) + 6561u * base3_of4x2((d32 >> 16) & 0xffu) +
                    531441u * base3_of4x2(d32 >> 24);
    S.steps = len - 16; S.tm = tm; S.kg = 0;
    S.cur.row = row; S.cur.W = W; S.cur.dirty = dirty;
    const int E = ix.t_e;
    if ((!FIXED || LOCATED) && ix.t20 && len >= 16 + E) {
        // the 16-mer lookup and the first E extensions in one table read
        if (((mbits >> 16) & ((1u << E) - 1)) == 0) {
            const u32 dE = (u32)(D >> 32);
            const u64 code = (u64)base3_of4x2(dE & 0xffu) + (E == 5 ? 81ull * ((dE >> 8) & 3u) : 0ull);
            const u64 v = ix.t20[key * t20_width(E) + code];
            const int tag = (int)(v >> 60);
            if (tag != 14) {
                n_hash++;
                const u64 rowv = v & ((1ull << 36) - 1), hits = (v >> 36) & ((1ull << 24) - 1);
                if (tag == 15) return false;                                           // hits 0, match length 0
                if (FIXED) {
                    if (tag > E && tag <= 2 * E) return false;
                    if (tag != 0) {
                        S.top = rowv | (1ull << 63); S.bot = S.top + 1; S.s = tag == 2 * E + 1 ? E : tag - 1;
                        return true;
                    }
                } else {
                    if (tag >= 1 && tag <= E) { out.ml = (u64)(15 + tag); out.sp = rowv | (1ull << 63); out.hits = 1; return false; }
                    if (tag == 2 * E + 1) { out.ml = (u64)(16 + E); out.sp = rowv | (1ull << 63); out.hits = 1; return false; }
                    if (tag > E) { out.ml = (u64)(15 - E + tag); out.sp = rowv; out.hits = hits; return false; }
                }
                S.top = rowv; S.bot = rowv + hits; S.s = E;
                if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = hits; return false; }
                S.cur.pos = tm + 16 + E; S.cur.buf = x >> (2 * (16 + E)); S.cur.have = 16 - E;
                return true;
            }
        }
    }
    hash_lookup(ix, key, S.top, S.bot);
    n_hash++;
    if (S.bot <= S.top) return false;
    S.s = 0;
    S.cur.pos = tm + 16; S.cur.buf = x >> 32; S.cur.have = 16;
    return true;
}

// Three letters at once (lf3_pair) while the interval shrinks slowly -- a read inside a repeat family walks the index for most of
// its length with hundreds of rows, one dependent gather pair per letter.  Exact by construction: a jump is TAKEN only when the
// interval behind it still has two rows or more, so none of the reference's stop conditions (one row left, letter absent, letter
// outside the alphabet -- bwt.h:2081-2209, 1848-1952) fell inside it; otherwise it is dropped, the three letters are stepped one by
// one as before, and the seed makes no further attempt.  Counted as three extensions (the reference's events).
template <bool FIXED, bool KG = false>
DEVI bool search_step_p(const DevIndex& ix, int L, SearchP& S, SeedHit& out, u32& n_ext, const u64* c3 = nullptr, u32* n_jump = nullptr)
{
    const int len = L - S.tm;
    const u64 ptop = S.top, pbot = S.bot;
    if (!FIXED) {
        if (pbot - ptop == 1) { out.ml = 16 + S.s; out.sp = ptop; out.hits = 1; return true; }
    } else out.ml = (u64)len;
    const u64 before = pbot - ptop;
    // which kind of step this lane takes: three letters (trigram g) or one (digit d).  Both kinds then share ONE gather pair -- a
    // jump tried in a branch of its own made every wave with a jumping lane wait out two memory round trips per iteration
    bool jump = false; int g = 0, d = 0;
    if (KG && S.kg >= (before >= 8 ? 1 : 2) && S.steps - S.s >= 3 && S.cur.have >= 3 && before >= 2) {
        jump = true;                                       // none of the three letters outside ACGT (their mask bits inside one word)
        if (S.cur.dirty) { const int o = S.cur.pos & 63; jump = o <= 61 && ((S.cur.row[S.cur.W + (S.cur.pos >> 6)] >> o) & 7ull) == 0; }
        const u32 b6 = (u32)S.cur.buf & 63u;
        g = ((0x46 >> (2 * (b6 & 3u))) & 3) + 3 * ((0x46 >> (2 * ((b6 >> 2) & 3u))) & 3) + 9 * ((0x46 >> (2 * (b6 >> 4))) & 3);
    }
    if (!jump) {
        d = S.cur.next3();
        if (d > 2) {
            if (!FIXED) { out.ml = 16 + S.s; out.sp = ptop; out.hits = pbot - ptop; } else { out.hits = 0; out.sp = 0; }
            return true;
        }
    }
    const u64 lt = ptop - (ptop > ix.shapline ? 1 : 0), lb = pbot - (pbot > ix.shapline ? 1 : 0);       // (single steps: '$' row removed)
    // rows fit 36 bits (checked at attach), so row >> 5 fits 32: the division by 96 is a 32-bit one by 3
    const u32 it = jump ? (u32)(ptop >> 5) / 3u : (u32)(lt >> 5), ib = jump ? (u32)(pbot >> 5) / 3u : (u32)(lb >> 5);
    const u32 rt = jump ? (u32)(ptop - (u64)it * 96) : (u32)lt & 31u, rb = jump ? (u32)(pbot - (u64)ib * 96) : (u32)lb & 31u;
    const uint4* base = jump ? ix.occ3 + (u64)g * ix.nb3 : ix.occ;
    const uint4 ht = base[it];
    uint4 hb = ht;
    if (ib != it) hb = base[ib];
    if (jump) {
        const u64 c3g = c3[g];
        const u64 t2 = c3g + occ3_in_block(ht, rt), b2 = c3g + occ3_in_block(hb, rb);
        if (b2 > t2 && b2 - t2 >= 2) {
            S.top = t2; S.bot = b2; S.s += 3; n_ext += 3;
            if (n_jump) (*n_jump)++;
            S.cur.buf >>= 6; S.cur.have -= 3; S.cur.pos += 3;
            if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = S.bot - S.top; return true; }
            return false;
        }
        S.kg = -(1 << 20);                                 // dropped: nothing was consumed, the next calls step letter by letter
        return false;
    }
    {
        const u32 mt = rt ? (~0u << (32 - rt)) : 0u, mb = rb ? (~0u << (32 - rb)) : 0u;
        u64 tT = (u64)ht.x + __popc(ht.z & mt), tA = (u64)ht.y + __popc(ht.w & mt);
        u64 bT = (u64)hb.x + __popc(hb.z & mb), bA = (u64)hb.y + __popc(hb.w & mb);
        if (ix.sup_shift) { super_add(ix, lt, tT, tA); super_add(ix, lb, bT, bA); }
        const u64 ct = d == 1 ? tT : (d == 2 ? tA : lt - tT - tA);
        const u64 cb = d == 1 ? bT : (d == 2 ? bA : lb - bT - bA);
        S.top = ix.C[d] + ct; S.bot = ix.C[d] + cb;
    }
    n_ext++;
    if (S.bot <= S.top) {
        if (!FIXED) { out.ml = 16 + S.s; out.sp = ptop; out.hits = pbot - ptop; } else { out.hits = 0; out.sp = S.top; }      // (FIXED: the remaining iterations only break)
        return true;
    }
    if (KG) S.kg = 2 * (S.bot - S.top) > before ? S.kg + 1 : (S.kg < 0 ? S.kg : 0);
    S.s++;
    if (S.s == S.steps) { out.ml = (u64)len; out.sp = S.top; out.hits = S.bot - S.top; return true; }
    return false;
}

// does read[q0 .. L) equal the text at doubled coordinate site0 + (q - tm) in the index alphabet?  (k_seed_second's single-row
// shortcut, on a packed row: 32 bases per step)
DEVI bool rest_matches_3l(const DevIndex& ix, const u64* row, int W, bool dirty, int L, int tm, int start, u64 site)
{
    for (int q = start & ~31; q < L; q += 32) {
        const u64 rb = row[q >> 5];
        const u64 d = site + (u64)(q - tm);                  // doubled coordinate facing read[q] (u64 wrap = out of range)
        u64 mm;
        if (d + 32 <= ix.total) mm = mism_3l(rb, gen_bases32(ix, d));
        else {
            // runs off the end of the text: '$' never matches
            mm = 0;
            for (int j = 0; j < 32; j++) {
                const u64 dj = d + (u64)j;
                const int rc = (int)((rb >> (2 * j)) & 3);
                bool eq = false;
                if (dj < ix.total) { const int g = gbase(ix, dj); eq = (rc == g) || ((rc & 1) && (g & 1)); }
                if (!eq) mm |= 1ull << (2 * j);
            }
        }
        if (dirty) mm |= spread32((u32)((row[W + (q >> 6)] >> (q & 63)) & 0xffffffffull));
        if (mm & field_range(start - q, L - q)) return false;
    }
    return true;
}

// determine_seed_offset_unmatch (Schema.h:1506-1531)
DEVI int seed_offset_unmatch(int L, int pre, const char* rd, int step)
{
    if (L - pre < 18 || L - pre < step) return L;
    const int ret = pre + step;
    for (int i = 0; i < step; i++, pre++) if (rd[pre] == 'N') return pre + 1;
    return ret;
}
// ... on packed rows: `mask` is the not-ACGT bit plane of the row; the ASCII row is asked only where a bit is set (the
// paired-end rows keep their ASCII text only in the 16-byte pieces that hold such a character, k_pe_prepare)
DEVI int seed_offset_unmatch_p(int L, int pre, const char* rd, int step, const u64* mask)
{
    if (L - pre < 18 || L - pre < step) return L;
    const int ret = pre + step;
    for (int i = 0; i < step; i++, pre++) if (((mask[pre >> 6] >> (pre & 63)) & 1) && rd[pre] == 'N') return pre + 1;
    return ret;
}

// per-read state carried between the seeding kernels
struct SeedCarry {
    u64* sp0; u32* hits0; u16* ml0;        // first seed result
    u16* tm; u8* seed_id; u32* clen; u16* first_ml;
    u32* flag_c; u32* flag_d;              // needs k_seed_second / k_seed_extra (scan inputs)
    u64* off_c; u64* off_d;                // exclusive scans
    u32* list_c; u32* list_d;              // compacted read lists
};

DEVI void seed_record(SeedRec* my, int& ns, u64& ncand, u64 sp, u64 hits, u64 len, u64 off)
{
    my[ns].sp = sp; my[ns].hits = (u32)hits; my[ns].len = (u16)len; my[ns].off = (u16)off; ns++;
    ncand += hits;
}

DEVI void seed_finish(const ReadState& st, long r, int verdict, int ns, u64 ncand, int pe_mode)
{
    st.verdict[r] = (u8)verdict;
    st.n_seeds[r] = (u8)ns;
    st.n_cand[r] = verdict == 3 ? (u32)ncand : (pe_mode ? (verdict == 4 ? (u32)ncand : (verdict == 1 || verdict == 2) ? 1u : 0u) : 0u);
}

#define SEED_BATCH 16             // pending lanes that trigger a transition batch
#define SEED_CHUNK 256            // items per wave (upper bound)
#define SEED_CHUNK_MIN 64
// items per wave for a work list of `total` items: enough waves to fill the chip a few times over (the second / extra
// lists hold 10-30 % of the batch; with 256-item chunks they gave fewer waves than the 8192 wave slots of the chip).
// (Measured and rejected, round 2: persistent waves -- exactly as many as the chip holds -- walking the list in strided 64-item
// chunks.  The per-wave timeline (BMBS_WAVELOG) then shows every wave slot occupied from start to end instead of 70-87 %, and
// the kernels get SLOWER (k_seed_first 1.95 -> 2.40 ms, k_seed_extra 3.73 -> 3.85 ms): the engine is bound by the request rate of
// the memory system, not by resident waves, and the long-lived waves end 20 % apart.  Chunks from a global atomic counter balance
// perfectly and cost 4.5x: ~60 k returning device-scope atomics on one word take several ms on this chip.)
DEVI long seed_chunk(long total, int target_waves)
{
    long c = (total + target_waves - 1) / target_waves;
    return c < SEED_CHUNK_MIN ? SEED_CHUNK_MIN : (c > SEED_CHUNK ? SEED_CHUNK : c);
}

struct LaneCounters { u32 n_hash, n_ext, n_sa, n_ung, n_jump; };      // n_jump: three-letter steps taken (each also counts three in n_ext)
// totals in counters[0,1,2,5]; per-kernel copies in counters[16 + 4*kid ..] (kid 0 first, 1 second, 2 extra)
DEVI void flush_counters(unsigned long long* counters, const LaneCounters& c, int kid)
{
    // 64-thread blocks: one wave; reduce with shuffles, one atomic per wave and counter
    u32 a = c.n_hash, b = c.n_ext, d = c.n_sa, e = c.n_ung, j = c.n_jump;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); d += __shfl_down(d, o); e += __shfl_down(e, o); j += __shfl_down(j, o); }
    if ((threadIdx.x & 63) == 0 && counters) {
        counters = SHARD(counters);
        if (j) atomicAdd(&counters[8], (unsigned long long)j);
        unsigned long long* k = counters + 16 + 4 * kid;
        if (a) { atomicAdd(&counters[0], (unsigned long long)a); atomicAdd(&k[0], (unsigned long long)a); }
        if (b) { atomicAdd(&counters[1], (unsigned long long)b); atomicAdd(&k[1], (unsigned long long)b); }
        if (d) { atomicAdd(&counters[2], (unsigned long long)d); atomicAdd(&k[2], (unsigned long long)d); }
        if (e) { atomicAdd(&counters[5], (unsigned long long)e); atomicAdd(&k[3], (unsigned long long)e); }
    }
}

// five stats values of a lane -> one LDS atomic per wave and value (256 lanes hammering five LDS words with 64-bit atomics was
// 40 % of k_finalize's issue time)
DEVI void wave_stats_add(unsigned long long* sh, u32 v0, u32 v1, u32 v2, u32 v3, u32 v4)
{
    for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_down(v0, o); v1 += __shfl_down(v1, o); v2 += __shfl_down(v2, o); v3 += __shfl_down(v3, o); v4 += __shfl_down(v4, o); }
    if ((threadIdx.x & 63) == 0) {
        if (v0) atomicAdd(&sh[0], (unsigned long long)v0);
        if (v1) atomicAdd(&sh[1], (unsigned long long)v1);
        if (v2) atomicAdd(&sh[2], (unsigned long long)v2);
        if (v3) atomicAdd(&sh[3], (unsigned long long)v3);
        if (v4) atomicAdd(&sh[4], (unsigned long long)v4);
    }
}

// ---- first seed of every read ------------------------------------------------------------------
template <bool PACKED, bool KG = false>
__global__ void __launch_bounds__(64)
k_seed_first(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, long n, SeedCarry sc,
             unsigned long long* __restrict__ counters)
{
    __shared__ u64 s_c3[KG ? 27 : 1];
    const u64* c3 = KG ? kgram_c3(ix, s_c3) : nullptr;
    const WaveLogT wl_t = wavelog_begin();
    int L = gm.L;                                     // length of the lane's current read
    const long chunk_begin = (long)blockIdx.x * SEED_CHUNK;
    const long chunk_end = chunk_begin + SEED_CHUNK < n ? chunk_begin + SEED_CHUNK : n;
    long next = chunk_begin;
    LaneCounters lc = {0, 0, 0, 0, 0};
    bool active = false;
    long r = 0;
    const char* rd = seq;
    typename std::conditional<PACKED, SearchP, Search>::type S; SeedHit h;
    // lanes whose search ended wait (`pending`) until SEED_BATCH of them can run the divergent
    // store / refill / hash-lookup code together: one straggler must not stall 63 stepping lanes
    bool pending = true, have = false;
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (__popcll(pm) >= SEED_BATCH || !__any(active)) {
            if (pm == 0) break;
            const int rank = __popcll(pm & ((1ull << (threadIdx.x & 63)) - 1));
            const long it = next + rank;
            next += __popcll(pm);
            if (pending) {
                if (have) { sc.sp0[r] = h.sp; sc.hits0[r] = (u32)h.hits; sc.ml0[r] = (u16)h.ml; have = false; }
                if (it < chunk_end) {
                    r = it; have = true; L = gm.rl(r);
                    bool go;
                    if constexpr (PACKED) go = search_begin_p<false>(ix, pr.base + (size_t)r * pr.pwords, pr.W, pr.dirty[r] != 0, L, 0, S, h, lc.n_hash);
                    else { rd = seq + (size_t)r * stride; go = search_begin<false>(ix, rd, L, 0, S, h, lc.n_hash); }
                    if (go) { active = true; pending = false; }
                    // else: decided at once; stays pending, stored at the next batch
                } else pending = false;
            }
            if (!__any(active) && !__any(pending)) break;
            continue;
        }
#ifdef BMBS_UTIL
        if ((threadIdx.x & 63) == 0) atomicAdd(&counters[8], 1ull);
        if (active) atomicAdd(&counters[9], 1ull);
#endif
        if (active) {
            bool fin;
            if constexpr (PACKED) fin = search_step_p<false, KG>(ix, L, S, h, lc.n_ext, c3, &lc.n_jump); else fin = search_step<false>(ix, rd, L, S, h, lc.n_ext);
            if (fin) { active = false; pending = true; }
        }
    }
    flush_counters(counters, lc, 0);
    wavelog_end(wl_t, 3);
}

// ---- exits after the first seed (Schema.cpp:24599-24727 / 18225-18330) ---------------------------
template <bool USE_LDS, bool VEC8>
__global__ void
k_seed_decide(DevIndex ix, const char* __restrict__ seq, ReadGeom gm, int stride, long n, int seed_len, int pe_mode, ReadState st,
              SeedCarry sc, unsigned long long* __restrict__ counters)
{
    __shared__ unsigned int shc[2];
    if (threadIdx.x < 2) shc[threadIdx.x] = 0;
    __syncthreads();
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    u32 n_sa = 0, n_ung = 0;
    // the block's 256 reads are contiguous in memory: copy them into LDS with fully coalesced 16-byte loads
    // (rows re-padded by 8 bytes against bank conflicts); each lane then scans ITS row out of LDS instead of
    // issuing 64-cache-line gathers per instruction
    extern __shared__ __align__(16) char lds_rows[];
    const int lstride = stride + 8;
    if (USE_LDS) {
        const long row0 = (long)blockIdx.x * blockDim.x;
        const long rows = n - row0 < (long)blockDim.x ? n - row0 : (long)blockDim.x;
        const int per_row = stride / 16;
        const int total16 = (int)rows * per_row;
        const uint4* src = reinterpret_cast<const uint4*>(seq + (size_t)row0 * stride);
        // piece q = (row rr, column cc); q advances by the block size, so (rr, cc) advance by its quotient and remainder -- one
        // division per thread instead of one (64-bit) per piece
        const int dq = (int)blockDim.x / per_row, dr = (int)blockDim.x - dq * per_row;
        int rr = (int)threadIdx.x / per_row, cc = (int)threadIdx.x - rr * per_row;
        for (int q = threadIdx.x; q < total16; q += blockDim.x) {
            const uint4 v = src[q];
            u64* dst = reinterpret_cast<u64*>(lds_rows + rr * lstride + cc * 16);
            dst[0] = ((u64)v.y << 32) | v.x;
            dst[1] = ((u64)v.w << 32) | v.z;
            rr += dq; cc += dr;
            if (cc >= per_row) { cc -= per_row; rr++; }
        }
    }
    __syncthreads();
    if (r < n) {
        const char* rd = USE_LDS ? lds_rows + (size_t)threadIdx.x * lstride : seq + (size_t)r * stride;
        const int L = gm.rl(r);
        int firstC = L;
        for (int i = 0; i < L; i += 8) {
            // lowest byte equal to 'C' in this 8-byte word (exact for the lowest-order zero byte)
            const u64 x = *reinterpret_cast<const u64*>(rd + i) ^ 0x4343434343434343ull;
            const u64 z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
            if (z) { const int j = i + (__ffsll((long long)z) - 1) / 8; if (j < L) firstC = j; break; }
        }
        SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        int ns = 0;
        u64 ncand = 0, clen = 0;
        const int max_seed = L / 10 == 0 ? 25 : (L / 10 - 1 > 25 ? 25 : L / 10 - 1);
        int verdict = 0, multi = 0, get_error = -1, tm = 0, seed_id = 0;
        u64 mm_site = 0, c0 = 0, first_ml = 0;
        const u64 max_hits = 1000;
        bool done = false;
        u32 fc = 0, fd = 0;
        if (seed_id < max_seed && tm < L) {
            const u64 hits = sc.hits0[r], sp = sc.sp0[r];
            u64 ml = sc.ml0[r];
            first_ml = ml;
            if (hits == 1) {
                // try_process_unique_mismatch_end_to_end (Schema.cpp:15164-15282)
                u64 p;
                if (sp >> 63) p = sp & ~(1ull << 63);          // the 20-mer table had the text position
                else { p = sa_at(ix, sp); n_sa++; }
                const u64 loc = ix.total - p - ml;
                seed_record(my, ns, ncand, sp, 1, ml, 0);
                c0 = loc; clen = 1;
                int error = 0;
                if (ml > (u64)firstC) ml = (u64)firstC;
                if (ml != (u64)L) {
                    const int need = L - (int)ml;
                    const u64 start = loc + ml;
                    n_ung++;
                    if (VEC8) {
                        if (!window_valid(ix, start, (u64)need, loc < ix.G)) {
                            // all-zero window: every position mismatches; the first sets ml = read_i (= ml), the second stops
                            error = need >= 2 ? 2 : 1;
                        } else {
                            // read position q faces doubled coordinate loc + q; 16 positions per step
                            const int ml0 = (int)ml;
                            // window bases cached in registers, 64 per 16-byte global load, the next block already on its way
                            // (every per-lane load is a request of its own: 3 wide loads instead of 6 narrow ones per read)
                            const uint4* g4 = reinterpret_cast<const uint4*>(ix.gen2);
                            u64 gidx = (loc + (u64)(ml0 & ~15)) >> 6;
                            uint4 cb = g4[gidx], nb = g4[gidx + 1];
                            for (int p = ml0 & ~15; p < L && error < 2; p += 16) {
                                const u64 r0 = *reinterpret_cast<const u64*>(rd + p), r1 = *reinterpret_cast<const u64*>(rd + p + 8);
                                const u64 d = loc + (u64)p;
                                if ((d >> 6) != gidx) { gidx = d >> 6; cb = nb; nb = g4[gidx + 1]; }
                                const int o = (int)(d & 63) * 2;
                                const u64 c0 = ((u64)cb.y << 32) | cb.x, c1 = ((u64)cb.w << 32) | cb.z, n0 = ((u64)nb.y << 32) | nb.x;
                                const u64 lo64 = o < 64 ? c0 : c1, hi64 = o < 64 ? c1 : n0;
                                const int sh = o & 63;
                                u64 w = lo64 >> sh;
                                if (sh > 32) w |= hi64 << (64 - sh);
                                const u32 w32 = (u32)w;
                                u64 m0 = mism8(r0, w32 & 0xffffu), m1 = mism8(r1, w32 >> 16);
                                if (p < ml0) {                                   // positions before ml0 are matched already
                                    const int lo = ml0 - p;
                                    if (lo >= 8) { m0 = 0; m1 &= ~((1ull << (8 * (lo - 8))) - 1); } else m0 &= ~((1ull << (8 * lo)) - 1);
                                }
                                if (p + 16 > L) {                                // the read ends inside this piece
                                    const int hi = L - p;
                                    if (hi <= 8) { m1 = 0; if (hi < 8) m0 &= (1ull << (8 * hi)) - 1; } else m1 &= (1ull << (8 * (hi - 8))) - 1;
                                }
                                if (m0 | m1) {
                                    const int cnt = __popcll(m0) + __popcll(m1);
                                    if (error == 0) {
                                        ml = (u64)(m0 ? p + (__ffsll((long long)m0) - 1) / 8 : p + 8 + (__ffsll((long long)m1) - 1) / 8);
                                        error = cnt >= 2 ? 2 : 1;
                                    } else error = 2;
                                }
                            }
                        }
                    } else {
                        WinReader wr; wr.init(ix, start, window_valid(ix, start, (u64)need, loc < ix.G));
                        int read_i = (int)ml;
                        ReadCur rc; rc.seek(rd, read_i, L);
                        for (int i = 0; i < need; i++) {
                            const char a = rc.next();
                            const int b = wr.next();                 // 4 when the window leaves the strand: never equal
                            if (!(code4(a) == b || (a == 'T' && b == 1))) { error++; if (error == 1) ml = (u64)read_i; else break; }
                            read_i++;
                        }
                    }
                }
                get_error = error;
                if (error == 0) { verdict = 1; st.exit_site[r] = loc; done = true; }
            }
            if (!done) {
                mm_site = ml;
                if (!pe_mode) {
                    if (ml == (u64)L && hits > 1) {
                        multi = 1;
                        if (firstC == L) { verdict = 4; done = true; }      // exact, ambiguous, no C in the read
                    }
                } else if (ml == (u64)L && hits > 1 && hits <= 10000) {
                    // get_candidates (Schema.cpp:18260-18290): every exact hit becomes a verified candidate
                    multi = 1;
                    if (firstC == L) { seed_record(my, ns, ncand, sp, hits, ml, 0); verdict = 4; done = true; }
                }
            }
            if (!done) {
                if (hits == 1) { /* recorded */ }
                else if (ml >= (u64)seed_len && hits <= max_hits) { if (hits != 0) { seed_record(my, ns, ncand, sp, hits, ml, (u64)tm); clen += hits; } }
                if (ml == 0) tm = seed_offset_unmatch(L, tm, rd, 8); else tm = tm + (int)(ml / 2);
                seed_id++;
            }
        }
        st.multi[r] = (u8)multi;
        // bit 15: the read character there is 'N' (L <= 1000).  k_finalize prices the 1-mismatch exit with it and need not touch the
        // read row again (one sector per such read)
        st.mm_site[r] = (u16)(mm_site | ((mm_site < (u64)L && rd[mm_site] == 'N') ? 0x8000u : 0u));
        if (done) seed_finish(st, r, verdict, ns, ncand, pe_mode);
        else {
            st.exit_site[r] = c0;
            st.n_seeds[r] = (u8)ns; st.n_cand[r] = (u32)ncand;
            sc.tm[r] = (u16)tm; sc.seed_id[r] = (u8)seed_id; sc.clen[r] = (u32)clen; sc.first_ml[r] = (u16)first_ml;
            // 1-mismatch first seed: second seed over the rest of the read (Schema.cpp:24734-24801)
            if (get_error == 1 && L - (int)first_ml >= 17) fc = 1; else fd = 1;
        }
        sc.flag_c[r] = fc; sc.flag_d[r] = fd;
    }
    if (counters) {
        atomicAdd(&shc[0], n_sa); atomicAdd(&shc[1], n_ung);
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long* cs = SHARD(counters); atomicAdd(&cs[2], (unsigned long long)shc[0]); atomicAdd(&cs[5], (unsigned long long)shc[1]); }
    }
}

// ---- the same kernel over packed rows: 64 bytes per read through LDS instead of 160, comparisons by whole-word XOR -------------
__global__ void __launch_bounds__(64)
k_seed_decide_p(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, long n, int seed_len, int pe_mode,
                ReadState st, SeedCarry sc, unsigned long long* __restrict__ counters)
{
    __shared__ unsigned int shc[2];
    if (threadIdx.x < 2) shc[threadIdx.x] = 0;
    extern __shared__ __align__(16) char lds_prows[];
    const int lw = pr.pwords + 1;                                   // row stride in words, odd against bank conflicts
    u64* lrows = reinterpret_cast<u64*>(lds_prows);
    const long row0 = (long)blockIdx.x * blockDim.x;
    const long rows = n - row0 < (long)blockDim.x ? n - row0 : (long)blockDim.x;
    {
        // the block's rows are contiguous in memory: copy them with coalesced 8-byte loads
        const u64* src = pr.base + (size_t)row0 * pr.pwords;
        const int total = (int)rows * pr.pwords;
        for (int q = threadIdx.x; q < total; q += blockDim.x) { const int rr = q / pr.pwords, cc = q - rr * pr.pwords; lrows[rr * lw + cc] = src[q]; }
    }
    __syncthreads();
    const long r = row0 + threadIdx.x;
    u32 n_sa = 0, n_ung = 0;
    if (r < n) {
        const u64* row = lrows + (size_t)threadIdx.x * lw;
        const int W = pr.W;
        const bool dirty = pr.dirty[r] != 0;
        const int L = gm.rl(r);
        // first 'C' of the read (code 01)
        int firstC = L;
        for (int w = 0; w * 32 < L; w++) {
            const u64 x = row[w];
            const u64 z = x & ~(x >> 1) & PK_EVEN;
            if (z) { const int j = w * 32 + (__ffsll((long long)z) - 1) / 2; if (j < L) firstC = j; break; }
        }
        SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        int ns = 0;
        u64 ncand = 0, clen = 0;
        const int max_seed = L / 10 == 0 ? 25 : (L / 10 - 1 > 25 ? 25 : L / 10 - 1);
        int verdict = 0, multi = 0, get_error = -1, tm = 0, seed_id = 0;
        u64 mm_site = 0, c0 = 0, first_ml = 0;
        const u64 max_hits = 1000;
        bool done = false;
        u32 fc = 0, fd = 0;
        if (seed_id < max_seed && tm < L) {
            const u64 hits = sc.hits0[r], sp = sc.sp0[r];
            u64 ml = sc.ml0[r];
            first_ml = ml;
            if (hits == 1) {
                // try_process_unique_mismatch_end_to_end (Schema.cpp:15164-15282)
                u64 p;
                if (sp >> 63) p = sp & ~(1ull << 63);          // the outcome table had the text position
                else { p = sa_at(ix, sp); n_sa++; }
                const u64 loc = ix.total - p - ml;
                seed_record(my, ns, ncand, sp, 1, ml, 0);
                c0 = loc; clen = 1;
                int error = 0;
                if (ml > (u64)firstC) ml = (u64)firstC;
                if (ml != (u64)L) {
                    const int need = L - (int)ml;
                    n_ung++;
                    if (!window_valid(ix, loc + ml, (u64)need, loc < ix.G)) {
                        // all-zero window: every position mismatches; the first sets ml = read_i (= ml), the second stops
                        error = need >= 2 ? 2 : 1;
                    } else {
                        // read position q faces doubled coordinate loc + q; 32 positions per step
                        const int ml0 = (int)ml;
                        GenStream gs; gs.init(ix, loc + (u64)(ml0 & ~31));
                        for (int q = ml0 & ~31; q < L && error < 2; q += 32) {
                            u64 mm = mism_bs(row[q >> 5], gs.next32());
                            if (dirty) mm |= spread32((u32)((row[W + (q >> 6)] >> (q & 63)) & 0xffffffffull));
                            mm &= field_range(ml0 - q, L - q);
                            if (mm) {
                                const int cnt = __popcll(mm);
                                if (error == 0) { ml = (u64)(q + (__ffsll((long long)mm) - 1) / 2); error = cnt >= 2 ? 2 : 1; }
                                else error = 2;
                            }
                        }
                    }
                }
                get_error = error;
                if (error == 0) { verdict = 1; st.exit_site[r] = loc; done = true; }
            }
            if (!done) {
                mm_site = ml;
                if (!pe_mode) {
                    if (ml == (u64)L && hits > 1) {
                        multi = 1;
                        if (firstC == L) { verdict = 4; done = true; }      // exact, ambiguous, no C in the read
                    }
                } else if (ml == (u64)L && hits > 1 && hits <= 10000) {
                    // get_candidates (Schema.cpp:18260-18290): every exact hit becomes a verified candidate
                    multi = 1;
                    if (firstC == L) { seed_record(my, ns, ncand, sp, hits, ml, 0); verdict = 4; done = true; }
                }
            }
            if (!done) {
                if (hits == 1) { /* recorded */ }
                else if (ml >= (u64)seed_len && hits <= max_hits) { if (hits != 0) { seed_record(my, ns, ncand, sp, hits, ml, (u64)tm); clen += hits; } }
                if (ml == 0) {
                    if (!dirty) tm = (L - tm < 18) ? L : tm + 8;
                    else tm = seed_offset_unmatch_p(L, tm, seq + (size_t)r * stride, 8, row + W);
                } else tm = tm + (int)(ml / 2);
                seed_id++;
            }
        }
        st.multi[r] = (u8)multi;
        // bit 15: the read character there is 'N' (only a dirty row can hold one: then the ASCII row is asked)
        bool isN = false;
        if (dirty && mm_site < (u64)L && ((row[W + (mm_site >> 6)] >> (mm_site & 63)) & 1)) isN = seq[(size_t)r * stride + mm_site] == 'N';
        st.mm_site[r] = (u16)(mm_site | (isN ? 0x8000u : 0u));
        if (done) seed_finish(st, r, verdict, ns, ncand, pe_mode);
        else {
            st.exit_site[r] = c0;
            st.n_seeds[r] = (u8)ns; st.n_cand[r] = (u32)ncand;
            sc.tm[r] = (u16)tm; sc.seed_id[r] = (u8)seed_id; sc.clen[r] = (u32)clen; sc.first_ml[r] = (u16)first_ml;
            // 1-mismatch first seed: second seed over the rest of the read (Schema.cpp:24734-24801)
            if (get_error == 1 && L - (int)first_ml >= 17) fc = 1; else fd = 1;
        }
        sc.flag_c[r] = fc; sc.flag_d[r] = fd;
    }
    if (counters) {
        atomicAdd(&shc[0], n_sa); atomicAdd(&shc[1], n_ung);
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long* cs = SHARD(counters); atomicAdd(&cs[2], (unsigned long long)shc[0]); atomicAdd(&cs[5], (unsigned long long)shc[1]); }
    }
}

// ---- second seed of the 1-mismatch reads + fast exit C (Schema.cpp:24734-24801, 24894-24898) -----
template <bool PACKED, bool KG = false>
__global__ void __launch_bounds__(64)
k_seed_second(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, const u64* __restrict__ count_ptr, int target_waves,
              int pe_mode, ReadState st, SeedCarry sc, unsigned long long* __restrict__ counters)
{
    int L = gm.L;                                     // length of the lane's current read
    const long total = (long)*count_ptr;
    const long chunk = seed_chunk(total, target_waves);
    const long chunk_begin = (long)blockIdx.x * chunk;
    if (chunk_begin >= total) return;
    __shared__ u64 s_c3[KG ? 27 : 1];
    const u64* c3 = KG ? kgram_c3(ix, s_c3) : nullptr;
    const WaveLogT wl_t = wavelog_begin();
    const long chunk_end = chunk_begin + chunk < total ? chunk_begin + chunk : total;
    long next = chunk_begin;
    LaneCounters lc = {0, 0, 0, 0, 0};
    bool active = false;
    long r = 0;
    const char* rd = seq;
    const u64* prow = nullptr;
    bool dirty = false;
    typename std::conditional<PACKED, SearchP, Search>::type S; SeedHit h;
    const u64 max_hits = 1000;
    bool verify = false;          // the interval shrank to one row: finish the count against the genome itself
    auto finish = [&]() {
        // the second seed is over: record it and decide (Schema.cpp:24748-24791)
        SeedRec* my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
        int ns = st.n_seeds[r];
        u64 ncand = st.n_cand[r], clen = sc.clen[r];
        const u64 first_ml = sc.first_ml[r];
        const u64 second_len = (u64)L - first_ml;
        const u64 c0 = st.exit_site[r];
        u64 c1 = 0;
        int extra = 1;
        if (verify) {
            // A single-row interval: every further backward extension only tests whether the text character in
            // front of that ONE occurrence equals the next read character.  Instead of one dependent random
            // Occ gather per character (count_hash_table, bwt.h:1889-1933) locate the row once and compare the
            // rest of the read with the doubled genome in the index alphabet (C folded into T), 8 bases a step.
            verify = false;
            u64 p;
            if (S.top >> 63) p = S.top & ~(1ull << 63);            // the 20-mer table had the text position
            else { p = sa_at(ix, S.top); lc.n_sa++; }
            const int done_chars = 16 + S.s;                       // read[tm, tm+done_chars) is matched at text position p
            const int tm = S.tm;
            const u64 site = ix.total - p - (u64)done_chars;       // doubled coordinate of read[tm]
            bool ok = true;
            // 16 positions per step: one 16-byte row load, the window through a two-word cursor (one load per 32 bases)
            const int start = tm + done_chars;
            if constexpr (PACKED) ok = rest_matches_3l(ix, prow, pr.W, dirty, L, tm, start, site);
            else {
            Win32Cur wc; wc.init(ix, site + (u64)((start & ~15) - tm));
            for (int q = start & ~15; q < L && ok; q += 16) {
                const uint4 v = *reinterpret_cast<const uint4*>(rd + q);
                const u64 rw[2] = {((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z};
                const u64 d = site + (u64)(q - tm);                 // doubled coordinate facing read[q] (u64 wrap = out of range)
                u64 m[2];
                if (d + 16 <= ix.total) {
                    const u32 w32 = wc.at(d);
                    m[0] = mism8_3letter(rw[0], w32 & 0xffffu); m[1] = mism8_3letter(rw[1], w32 >> 16);
                } else {                                             // runs off the end of the text: '$' never matches
                    for (int hf = 0; hf < 2; hf++) {
                        m[hf] = 0;
                        for (int j = 0; j < 8; j++) {
                            const u64 dj = d + (u64)(8 * hf + j);
                            const char a = (char)((rw[hf] >> (8 * j)) & 0xff);
                            const bool eq = dj < ix.total && code3(a) <= 2 && code3(a) == code3("ACGT"[gbase(ix, dj)]);
                            if (!eq) m[hf] |= 0x80ull << (8 * j);
                        }
                    }
                }
                if (q < start) {                                     // positions before `start` are matched already
                    const int lo = start - q;
                    if (lo >= 8) { m[0] = 0; m[1] &= ~((1ull << (8 * (lo - 8))) - 1); } else m[0] &= ~((1ull << (8 * lo)) - 1);
                }
                if (q + 16 > L) {                                    // the read ends inside this piece
                    const int hi = L - q;
                    if (hi <= 8) { m[1] = 0; if (hi < 8) m[0] &= (1ull << (8 * hi)) - 1; } else m[1] &= (1ull << (8 * (hi - 8))) - 1;
                }
                if (m[0] | m[1]) ok = false;
            }
            }
            if (ok) { h.hits = 1; h.sp = (p - (u64)(S.steps - S.s)) | (1ull << 63); }     // located: text position of the full seed
            else { h.hits = 0; h.sp = 0; }
        }
        if (h.hits == 1) {
            u64 p;
            if (h.sp >> 63) p = h.sp & ~(1ull << 63);
            else { p = sa_at(ix, h.sp); lc.n_sa++; }
            c1 = ix.total - p - second_len - first_ml;
            seed_record(my, ns, ncand, h.sp, 1, second_len, first_ml);
            clen += 1; extra = 0;
        } else if (h.hits <= max_hits) {
            if (h.hits != 0) { seed_record(my, ns, ncand, h.sp, h.hits, second_len, first_ml); clen += h.hits; }
            extra = 0;
        }
        if (extra == 0) {
            int verdict = 0;
            if (clen == 1 || (clen == 2 && c0 == c1)) verdict = 2;            // fast exit C: exit_site = c0 already stored
            else if (clen != 0) verdict = 3;
            seed_finish(st, r, verdict, ns, ncand, pe_mode);
            sc.flag_d[r] = 0;
        } else {
            st.n_seeds[r] = (u8)ns; st.n_cand[r] = (u32)ncand; sc.clen[r] = (u32)clen;
            sc.flag_d[r] = 1;
        }
    };
    bool pending = true, have = false;
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (__popcll(pm) >= SEED_BATCH || !__any(active)) {
            if (pm == 0) break;
            const int rank = __popcll(pm & ((1ull << (threadIdx.x & 63)) - 1));
            const long it = next + rank;
            next += __popcll(pm);
            if (pending) {
                if (have) { finish(); have = false; }
                if (it < chunk_end) {
                    r = sc.list_c[it]; have = true; L = gm.rl(r);
                    bool go;
                    if constexpr (PACKED) {
                        prow = pr.base + (size_t)r * pr.pwords; dirty = pr.dirty[r] != 0;
                        go = search_begin_p<true, true>(ix, prow, pr.W, dirty, L, (int)sc.first_ml[r], S, h, lc.n_hash);
                    } else { rd = seq + (size_t)r * stride; go = search_begin<true, true>(ix, rd, L, (int)sc.first_ml[r], S, h, lc.n_hash); }
                    if (go) {
                        if (S.bot - S.top == 1) verify = true;          // already a single row: stays pending, verified next batch
                        else { active = true; pending = false; }
                    }
                } else pending = false;
            }
            if (!__any(active) && !__any(pending)) break;
            continue;
        }
#ifdef BMBS_UTIL
        if ((threadIdx.x & 63) == 0) atomicAdd(&counters[10], 1ull);
        if (active) atomicAdd(&counters[11], 1ull);
#endif
        if (active) {
            bool fin;
            if constexpr (PACKED) fin = search_step_p<true, KG>(ix, L, S, h, lc.n_ext, c3, &lc.n_jump); else fin = search_step<true>(ix, rd, L, S, h, lc.n_ext);
            if (fin) { active = false; pending = true; }
            else if (S.bot - S.top == 1) { verify = true; active = false; pending = true; }
        }
    }
    flush_counters(counters, lc, 1);
    wavelog_end(wl_t, 4);
}

// ---- the remaining seeds (Schema.cpp:24809-24889) -------------------------------------------------
// ROWS_LDS: ASCII rows staged in LDS (small indexes); PACKED: packed rows read from global memory (64 bytes per read)
// PLDS (with PACKED): the packed row of a lane's read (pwords x 8 bytes) is copied into the lane's LDS slot when the lane takes the
// read -- coalesced 16-byte loads along the rows -- and every seed start and cursor refill reads LDS: one request to the memory
// pipeline per seed start less (of about five), at 64 x (pwords + 1) x 8 = 4.6 KB of LDS per wave for 150-base reads
template <bool ROWS_LDS, bool PACKED = false, bool PLDS = false, bool KG = false>
__global__ void __launch_bounds__(64)
k_seed_extra(DevIndex ix, const char* __restrict__ seq, PackedRows pr, ReadGeom gm, int stride, const u64* __restrict__ count_ptr, int target_waves,
             int seed_len, int pe_mode_x, ReadState st, SeedCarry sc, unsigned long long* __restrict__ counters)
{
    const int pe_mode = pe_mode_x & 0xff;
    const bool exp_nostore = (pe_mode_x >> 8) & 1;    // BMBS_EXP=1 (timing experiment)
    int L = gm.L;                                     // length of the lane's current read
    const long total = (long)*count_ptr;
    const long chunk = seed_chunk(total, target_waves);
    const long chunk_begin = (long)blockIdx.x * chunk;
    if (chunk_begin >= total) return;
    __shared__ u64 s_c3[KG ? 27 : 1];
    const u64* c3 = KG ? kgram_c3(ix, s_c3) : nullptr;
    const WaveLogT wl_t = wavelog_begin();
    const long chunk_end = chunk_begin + chunk < total ? chunk_begin + chunk : total;
    long next = chunk_begin;
    LaneCounters lc = {0, 0, 0, 0, 0};
    bool active = false, have = false;
    long r = 0;
    // A lane keeps its read for all the remaining seeds (about eight), and every seed start reads the row at a new offset.
    // Each such per-lane load is a request of its own to the memory pipeline (tools/gather_bench.hip), two thirds of all the
    // requests of this kernel.  So the wave copies the rows of the lanes that take a new read into LDS, with 16-byte loads
    // that run along the rows (about three cache lines per row instead of ~18 requests), and the searches read LDS
    // (ROWS_LDS = false: rows too long for the LDS of a 64-lane block; the lanes then read global memory as before.  A
    // template flag, so that the row pointer has one address space and the reads compile to ds_read, not to flat loads.)
    extern __shared__ __align__(16) char lds_rows[];
    __shared__ u32 take_row[64];
    __shared__ u8 take_lane[64];
    const int lstride = stride + 16;                  // 16-byte aligned rows, shifted against bank conflicts
    const char* rd = ROWS_LDS ? lds_rows + (size_t)(threadIdx.x & 63) * lstride : seq;
    const int plw = pr.pwords + 1;                    // PLDS: LDS row stride in words, odd against bank conflicts
    const u64* prow = PLDS ? reinterpret_cast<const u64*>(lds_rows) + (size_t)(threadIdx.x & 63) * plw : nullptr;
    bool dirty = false;
    typename std::conditional<PACKED, SearchP, Search>::type S; SeedHit h = {0, 0, 0};
    SeedRec* my = nullptr;
    int ns = 0, tm = 0, seed_id = 0, max_seed = 0;
    u64 ncand = 0, clen = 0;
    const u64 max_hits = 1000;
    // after a seed: record, advance (Schema.cpp:24830-24882); returns false when the read is finished
    auto after_seed = [&]() -> bool {
        const int cur_len = L - tm;
        const u64 ml = h.ml;
        auto rec = [&](u64 hits) { if (exp_nostore) { ns++; ncand += hits; } else seed_record(my, ns, ncand, h.sp, hits, ml, (u64)tm); };
        if (h.hits == 1) { rec(1); clen += 1; }
        else if (ml >= (u64)seed_len && h.hits <= max_hits) { if (h.hits != 0) { rec(h.hits); clen += h.hits; } }
        else if ((u64)cur_len == ml) return false;
        if (ml == 0) {
            // only a read with a character outside ACGT can hold the 'N' determine_seed_offset_unmatch looks for
            if (PACKED && !dirty) tm = (L - tm < 18) ? L : tm + 8;
            else if constexpr (PACKED) tm = seed_offset_unmatch_p(L, tm, seq + (size_t)r * stride, 8, prow + pr.W);
            else tm = seed_offset_unmatch(L, tm, rd, 8);
        } else tm = tm + (int)(ml / 2);
        seed_id++;
        return true;
    };
    // one transition of a pending lane: book the finished seed, start the next one or the next read.
    // `pending` stays set when the new seed was decided without stepping (handled in the next batch).
    bool pending = true, seed_done = false;
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (__popcll(pm) >= SEED_BATCH || !__any(active)) {
            if (pm == 0) break;
            if (pending && have && seed_done) {
                seed_done = false;
                if (!after_seed()) { seed_finish(st, r, clen != 0 ? 3 : 0, ns, ncand, pe_mode); have = false; }
            }
            if (pending && have && !(seed_id < max_seed && tm < L)) { seed_finish(st, r, clen != 0 ? 3 : 0, ns, ncand, pe_mode); have = false; }
            // lanes without a read take the next ones of the chunk
            const unsigned long long want = __ballot(pending && !have);
            const int rank = __popcll(want & ((1ull << (threadIdx.x & 63)) - 1));
            const long it = next + rank;
            const long avail = chunk_end - next;
            const int n_take = (long)__popcll(want) < avail ? __popcll(want) : (int)(avail > 0 ? avail : 0);
            next += __popcll(want);
            if ((ROWS_LDS || PLDS) && n_take > 0) {
                // (source row, destination lane) of every taker, then the copy: lane t moves piece t % per_row of taker t / per_row
                if (pending && !have && it < chunk_end) { take_row[rank] = sc.list_d[it]; take_lane[rank] = (u8)(threadIdx.x & 63); }
                __syncthreads();
                if constexpr (PLDS) {
                    const int per_row = pr.pwords / 2, pieces = n_take * per_row;         // rows are 16-byte aligned, pwords is even
                    u64* lr = reinterpret_cast<u64*>(lds_rows);
                    for (int t = threadIdx.x & 63; t < pieces; t += 64) {
                        const int w = t / per_row, cc = t - w * per_row;
                        u64 a, b; load2(pr.base + (size_t)take_row[w] * pr.pwords + 2 * cc, a, b);
                        u64* d = lr + (size_t)take_lane[w] * plw + 2 * cc;
                        d[0] = a; d[1] = b;
                    }
                } else {
                    const int per_row = stride / 16, pieces = n_take * per_row;
                    for (int t = threadIdx.x & 63; t < pieces; t += 64) {
                        const int w = t / per_row, cc = t - w * per_row;
                        const uint4 v = *reinterpret_cast<const uint4*>(seq + (size_t)take_row[w] * stride + (size_t)cc * 16);
                        *reinterpret_cast<uint4*>(lds_rows + (size_t)take_lane[w] * lstride + cc * 16) = v;
                    }
                }
                __syncthreads();
            }
            if (pending && !have) {
                if (it < chunk_end) {
                    r = sc.list_d[it]; L = gm.rl(r);
                    if (!ROWS_LDS) rd = seq + (size_t)r * stride;
                    if constexpr (PACKED) { if constexpr (!PLDS) prow = pr.base + (size_t)r * pr.pwords; dirty = pr.dirty[r] != 0; }
                    my = st.seeds + (size_t)r * BMBS_MAX_SEEDS;
                    ns = st.n_seeds[r]; ncand = st.n_cand[r]; clen = sc.clen[r]; tm = sc.tm[r]; seed_id = sc.seed_id[r];
                    max_seed = L / 10 == 0 ? 25 : (L / 10 - 1 > 25 ? 25 : L / 10 - 1);
                    have = true;
                    if (!(seed_id < max_seed && tm < L)) { seed_finish(st, r, clen != 0 ? 3 : 0, ns, ncand, pe_mode); have = false; }
                } else pending = false;         // chunk exhausted: this lane is done
            }
            if (pending && have) {
                bool go;
                if constexpr (PACKED) go = search_begin_p<false>(ix, prow, pr.W, dirty, L, tm, S, h, lc.n_hash); else go = search_begin<false>(ix, rd, L, tm, S, h, lc.n_hash);
                if (go) { active = true; pending = false; }
                else seed_done = true;              // decided without stepping: booked in the next batch
            }
            if (!__any(active) && !__any(pending)) break;
            continue;
        }
#ifdef BMBS_UTIL
        if ((threadIdx.x & 63) == 0) atomicAdd(&counters[12], 1ull);
        if (active) atomicAdd(&counters[13], 1ull);
#endif
        if (active) {
            bool fin;
            if constexpr (PACKED) fin = search_step_p<false, KG>(ix, L, S, h, lc.n_ext, c3, &lc.n_jump); else fin = search_step<false>(ix, rd, L, S, h, lc.n_ext);
            if (fin) { active = false; pending = true; seed_done = true; }
        }
    }
    flush_counters(counters, lc, 2);
    wavelog_end(wl_t, 5);
}
