// oracle/orc_map.cpp -- TEST INFRASTRUCTURE (see bmbs_oracle.h).
// Scalar CPU restatement of the per-read mapping machine of BitMapperBS (single-end path):
// seeding (bwt.h), candidate voting (Schema.cpp), BS banded Myers filter (Levenshtein_Cal.h),
// quality-aware banded affine-gap alignment (ksw.cpp), MAPQ and SAM record formatting.
#include "orc_internal.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" void orc_default_params(orc_params* p)
{
    p->e_f = 0.08; p->mp_max = 6; p->mp_min = 2; p->np = 1; p->gap_open = 5; p->gap_ext = 3;
    p->q_base = 33; p->seed_len = 30; p->min_ins = 0; p->max_ins = 500; p->sensitive = 0;
    p->unmapped_out = 0; p->ambiguous_out = 0; p->pbat = 0;
}

// ------------------------------------------------------------------------------------------------
// K7  get_actuall_genome / get_actuall_rc_genome  (Schema.cpp:4998-5115)
static void window_fwd(const orc_index* ix, u64 start, u64 len, char* out)
{
    if (start + len > ix->G) { memset(out, 0, len); return; }
    for (u64 i = 0; i < len; i++) {
        u64 p = start + i;
        out[i] = "ACGT"[(ix->pac[p >> 2] >> (6 - 2 * (p & 3))) & 3];
    }
}
static void window_rc(const orc_index* ix, u64 rc_start, u64 len, char* out)
{
    u64 end_site = ix->G - rc_start - 1;            // u64 wrap-around intended (as in the reference)
    if (end_site < len - 1 || (end_site >> 2) >= ix->pac_bytes) { memset(out, 0, len); return; }
    for (u64 i = 0; i < len; i++) {
        u64 p = end_site - i;
        out[i] = "TGCA"[(ix->pac[p >> 2] >> (6 - 2 * (p & 3))) & 3];
    }
}
void window_at(const orc_index* ix, u64 site, u64 len, char* out)
{
    if (site < ix->G) window_fwd(ix, site, len, out);
    else window_rc(ix, site - ix->G, len, out);
}
extern "C" void orc_window(const orc_index* ix, uint64_t site, int len, char* out) { window_at(ix, site, len, out); }

// ------------------------------------------------------------------------------------------------
// K8  BS_Reserve_Banded_BPM  (Levenshtein_Cal.h:351-567), 64-bit words.
// pattern = reference window (L+2k), text = read (L).  Band of 2k+1 bits slides down the diagonal;
// read 'T' matches window 'C' or 'T' (Peq['T'] |= Peq['C'], :384,473).
extern "C" int orc_bpm(const char* pattern, int p_length, const char* text, int t_length, int k, unsigned* return_err)
{
    (void)p_length;
    *return_err = (unsigned)-1;
    u64 PA = 0, PC = 0, PG = 0, PT = 0;
    const int band = 2 * k + 1;
    auto setbit = [&](char ch, u64 bit) {
        if (ch == 'A') PA |= bit; else if (ch == 'C') PC |= bit; else if (ch == 'G') PG |= bit; else if (ch == 'T') PT |= bit;
    };
    for (int i = 0; i < band; i++) setbit(pattern[i], (u64)1 << i);
    PT |= PC;
    const u64 Mask = (u64)1 << (2 * k);
    u64 VP = 0, VN = 0, X, D0, HN, HP;
    int err = 0, i_bd = 2 * k;
    const int last_high = 2 * k;
    for (int i = 0; i < t_length; i++) {
        char tc = text[i];
        u64 eq = tc == 'A' ? PA : tc == 'C' ? PC : tc == 'G' ? PG : tc == 'T' ? PT : 0;
        X = eq | VN;
        D0 = ((VP + (X & VP)) ^ VP) | X;
        HN = VP & D0;
        HP = VN | ~(VP | D0);
        X = D0 >> 1;
        VN = X & HP;
        VP = HN | ~(X | HP);
        if (!(D0 & 1)) {
            ++err;
            if (err - last_high > k) return -1;
        }
        if (i + 1 < t_length) {
            PA >>= 1; PC >>= 1; PG >>= 1; PT >>= 1;
            ++i_bd;
            setbit(pattern[i_bd], Mask);
            PT |= PC;
        }
    }
    const int site = t_length - 1;
    int return_site = -1;
    if (err <= k && (unsigned)err <= *return_err) { *return_err = err; return_site = site; }
    int i = 0;
    while (i < k) {
        err += (int)((VP >> i) & 1);
        err -= (int)((VN >> i) & 1);
        ++i;
        if (err <= k && (unsigned)err <= *return_err) { *return_err = err; return_site = site + i; }
    }
    unsigned ungap_err = (unsigned)err;
    while (i < last_high) {
        err += (int)((VP >> i) & 1);
        err -= (int)((VN >> i) & 1);
        ++i;
        if (err <= k && (unsigned)err <= *return_err) { *return_err = err; return_site = site + i; }
    }
    if (ungap_err <= (unsigned)k && ungap_err == *return_err) return_site = site + k;
    return return_site;
}

// ------------------------------------------------------------------------------------------------
// MismatchPenaltyByQuality (ksw.h:148-161): IEEE double arithmetic, truncation toward zero
int mismatch_penalty(const orc_params* P, int Q)
{
    double Phred = Q - P->q_base;
    if (Phred > 40) Phred = 40;
    Phred = Phred / 40;
    int r = Phred * (P->mp_max - P->mp_min);
    return r + P->mp_min;
}

static const unsigned char nt4[256] = {
#define X4 4,4,4,4
#define X16 X4,X4,X4,X4
    X16, X16, X16, X16,
    4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4,
    4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4,
    X16, X16, X16, X16, X16, X16, X16, X16
#undef X4
#undef X16
};

// K11  try_cigar_without_path  (ksw.cpp:2515-2570)
static int try_ungapped(const orc_params* P, const char* pattern, const char* text, int t_length,
                        int end_site, int err, int* start_out, int* score, const char* qual)
{
    int tmp_err = 0, start_site = end_site - t_length + 1;
    *score = 0;
    if (start_site >= 0) {
        for (int i = 0; i < t_length; i++) {
            char a = text[i], b = pattern[i + start_site];
            if (a != b && !(a == 'T' && b == 'C')) {
                if (++tmp_err > err) return 0;
                if (a == 'N' || b == 'N') *score -= P->np;
                else *score -= mismatch_penalty(P, (unsigned char)qual[i]);
            }
        }
        if (tmp_err == err) { *start_out = start_site; return 6; }
    }
    return 0;
}

// K12  ksw_semi_global_quality_back  (ksw.cpp:1850-2045).  query = window (columns), target = read
// (rows); band [i, i+2w] on row i; trace byte = h | e<<2 | f<<4.  Returns ops (len<<4|op, op 0 M,
// 1 D (window only), 2 I (read only)) in forward order.
struct sw_out { int score, qb, qe; std::vector<u32> cig; };
static void sw_banded(const orc_params* P, int qlen, const char* query, int tlen, const char* target,
                      int w, const char* qual, sw_out* out)
{
    const int MINUS_INF = -0x40000000;
    const int gapoe = P->gap_open + P->gap_ext, gape = P->gap_ext;
    const int band = 2 * w + 1;
    // mat / mat_diff (Schema.cpp:830-850): [read base][window base]
    int mat[25], mat_diff[25];
    {
        int k = 0;
        for (int i = 0; i < 4; i++) {
            for (int j = 0; j < 4; j++) { mat_diff[k] = i == j ? 0 : (P->mp_max - P->mp_min); mat[k++] = i == j ? 0 : -P->mp_min; }
            mat_diff[k] = 0; mat[k++] = -P->np;
        }
        for (int j = 0; j < 5; j++) { mat_diff[k] = 0; mat[k++] = -P->np; }
        mat_diff[16] = 0; mat[16] = 0;           // read T vs window C is a match
    }
    std::vector<int> H(qlen + 2), E(qlen + 2);
    std::vector<u8> z((size_t)band * tlen);
    int j;
    for (j = 0; j < band; ++j) { H[j] = 0; E[j] = -gapoe; }
    for (; j <= qlen; ++j) H[j] = E[j] = MINUS_INF;
    int beg = 0, end = 0;
    for (int i = 0; i < tlen; ++i) {
        int f = MINUS_INF, h1 = MINUS_INF;
        const int tc = nt4[(unsigned char)target[i]];
        double Phred = (unsigned char)qual[i] - P->q_base;
        if (Phred > 40) Phred = 40;
        Phred = Phred / 40;
        beg = i; end = i + band;
        u8* zi = &z[(size_t)i * band];
        for (j = beg; j < end; ++j) {
            int m = H[j], e = E[j], h, t;
            u8 d;
            H[j] = h1;
            const int qc = nt4[(unsigned char)query[j]];
            m = m + mat[tc * 5 + qc] - (int)(mat_diff[tc * 5 + qc] * Phred);
            d = m >= e ? 0 : 1;
            h = m >= e ? m : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            t = m - gapoe;
            e -= gape;
            d |= e > t ? 1 << 2 : 0;
            e = e > t ? e : t;
            E[j] = e;
            f -= gape;
            d |= f > t ? 2 << 4 : 0;
            f = f > t ? f : t;
            zi[j - beg] = d;
        }
        H[end] = h1; E[end] = MINUS_INF;
    }
    int max_i = tlen + w, score = H[max_i];
    for (int i = end; i > beg; i--)
        if (H[i] > score) { score = H[i]; max_i = i; }
    out->score = score;
    out->qe = max_i - 1;
    // backtrack
    std::vector<u32>& cg = out->cig;
    cg.clear();
    auto push = [&](int op, int len) {
        if (cg.empty() || op != (int)(cg.back() & 0xf)) cg.push_back((u32)len << 4 | op);
        else cg.back() += (u32)len << 4;
    };
    int i = tlen - 1, k = max_i - 1, which = 0;
    while (i >= 0 && k >= 0) {
        which = z[(size_t)i * band + (k - i)] >> (which << 1) & 3;
        if (which == 0) { push(0, 1); --i; --k; }
        else if (which == 1) { push(2, 1); --i; }
        else { push(1, 1); --k; }
    }
    if (i >= 0) push(2, i + 1);
    std::reverse(cg.begin(), cg.end());
    out->qb = k + 1;
}

// K11-K13  fast_recalculate_bs_Cigar  (ksw.cpp:2578-2876)
extern "C" int orc_align(const orc_params* P, const char* pattern, int p_length, const char* text,
                         const char* qual_in, int t_length, int k, int end_site, unsigned error,
                         int is_forward, int reverse_quality, int* start_site, int* new_end_site,
                         unsigned* nm, int* score, char* cigar)
{
    std::string qual(qual_in, qual_in + t_length);
    if (reverse_quality) std::reverse(qual.begin(), qual.end());
    if (try_ungapped(P, pattern, text, t_length, end_site, (int)error, start_site, score, qual.data())) {
        *new_end_site = end_site; *nm = error;
        sprintf(cigar, "%dM", t_length);
        return 6;
    }
    sw_out so;
    sw_banded(P, p_length, pattern, t_length, text, k, qual.data(), &so);
    std::vector<u32>& cg = so.cig;
    int n_cigar = (int)cg.size(), i, op, opl, ins = 0;
    cg.push_back(0);                               // the reference may read one slot past the end
    // leading I -> M
    for (i = 0; i < n_cigar; ++i) { op = cg[i] & 0xf; opl = cg[i] >> 4; if (op != 2) break; ins += opl; }
    if (i != 0) {
        op = cg[i] & 0xf; opl = cg[i] >> 4;
        if (op == 0) opl += ins; else { op = 0; opl = ins; i--; }
        cg[i] = ((u32)opl << 4) | op;
        so.qb -= ins;
    }
    int cigar_b = i;
    ins = 0;
    for (i = n_cigar - 1; i >= cigar_b; --i) { op = cg[i] & 0xf; opl = cg[i] >> 4; if (op != 2) break; ins += opl; }
    if (i != n_cigar - 1) {
        op = cg[i] & 0xf; opl = cg[i] >> 4;
        if (op == 0) opl += ins; else { op = 0; opl = ins; i++; }
        cg[i] = ((u32)opl << 4) | op;
        so.qe += ins;
    }
    int cigar_e = i, NM = 0;
    cigar[0] = 0;
    char* cp = cigar;
    if (is_forward) {
        int qs = so.qb, ts = 0;
        for (i = cigar_b; i <= cigar_e; ++i) {
            op = cg[i] & 0xf; opl = cg[i] >> 4;
            cp += sprintf(cp, "%d%c", opl, "MDISH"[op]);
            if (op == 0) {
                for (int q = 0; q < opl; q++) {
                    if (pattern[qs] != text[ts] && !(pattern[qs] == 'C' && text[ts] == 'T')) NM++;
                    qs++; ts++;
                }
            } else if (op == 1) { qs += opl; NM += opl; }
            else { ts += opl; NM += opl; }
        }
    } else {
        int qe = so.qe, te = t_length - 1;
        for (i = cigar_e; i >= cigar_b; --i) {
            op = cg[i] & 0xf; opl = cg[i] >> 4;
            cp += sprintf(cp, "%d%c", opl, "MDISH"[op]);
            if (op == 0) {
                for (int q = 0; q < opl; q++) {
                    if (pattern[qe] != text[te] && !(pattern[qe] == 'C' && text[te] == 'T')) NM++;
                    qe--; te--;
                }
            } else if (op == 1) { qe -= opl; NM += opl; }
            else { te -= opl; NM += opl; }
        }
    }
    *score = so.score; *start_site = so.qb; *new_end_site = so.qe; *nm = (unsigned)NM;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// a19  MAP_Calculation  (Schema.cpp:168-405)
extern "C" int orc_mapq(const orc_params* P, unsigned second_best_diff, unsigned error_threshold, int best_score)
{
    int scoreMax = P->gap_open + P->gap_ext;
    if (scoreMax < P->mp_max) scoreMax = P->mp_max;
    scoreMax = -scoreMax * error_threshold;
    int scoreMaxRange = -scoreMax;
    int score_diff = best_score - scoreMax;
    if (score_diff < 0) score_diff = 0;
    int error_diff = second_best_diff;
    if (second_best_diff > error_threshold) error_diff = error_threshold + 1;
    double rank, rank_error;
    if ((unsigned)error_diff > error_threshold) {
        rank = (double)score_diff / (double)scoreMaxRange;
        if (rank >= 0.8) return 42;
        if (rank >= 0.7) return 40;
        if (rank >= 0.6) return 24;
        if (rank >= 0.5) return 23;
        if (rank >= 0.4) return 8;
        if (rank >= 0.3) return 3;
        return 0;
    }
    rank_error = (double)error_diff / (double)error_threshold;
    rank = (double)score_diff / (double)scoreMaxRange;
    const bool z = best_score == 0;
    if (rank_error >= 0.9) return z ? 39 : 33;
    if (rank_error >= 0.8) return z ? 38 : 27;
    if (rank_error >= 0.7) return z ? 37 : 26;
    if (rank_error >= 0.6) return z ? 36 : 22;
    if (rank_error >= 0.5) return z ? 35 : rank >= 0.84 ? 25 : rank >= 0.68 ? 16 : 5;
    if (rank_error >= 0.4) return z ? 34 : rank >= 0.84 ? 21 : rank >= 0.68 ? 14 : 4;
    if (rank_error >= 0.3) return z ? 32 : rank >= 0.88 ? 18 : rank >= 0.67 ? 15 : 3;
    if (rank_error >= 0.2) return z ? 31 : rank >= 0.88 ? 17 : rank >= 0.67 ? 11 : 0;
    if (rank_error >= 0.1) return z ? 30 : rank >= 0.88 ? 12 : rank >= 0.67 ? 7 : 0;
    if (error_diff == 0) return rank >= 0.67 ? 1 : 0;
    return rank >= 0.67 ? 6 : 2;
}

// ------------------------------------------------------------------------------------------------
// seeding.  bsSeq = reverse(read) with C->T (C_to_T_forward, Schema.h:1534); ctoi G0 T1 A2 else 4
static inline int ctoi3(char c) { return c == 'G' ? 0 : c == 'T' ? 1 : c == 'A' ? 2 : 4; }

// get_3_letter_hash_value, bwt.h:309-332
static u64 hash16(const char* p)
{
    u64 h = 0;
    for (int i = 0; i < 16; i++) { int d = ctoi3(p[i]); if (d > 2) return (u64)-1; h = h * 3 + d; }
    return h;
}


// K3  count_backward_as_much_1_terminate  (bwt.h:2081-2209)
seed_res count_terminate(const orc_index* ix, const char* pat, u64 length, orc_counters* C)
{
    seed_res r = {0, 0, 0, 0};
    if (length < 18) return r;
    long long j = (long long)length - 16;
    u64 hv = hash16(pat + j);
    if (hv == (u64)-1) return r;
    u64 top, bot;
    orc_hash_query(ix, hv, &top, &bot);
    if (C) C->n_hash++;
    if (bot <= top) return r;
    u64 pre_top = (u64)-1, pre_bot = (u64)-1;
    for (j = j - 1; j >= 0; j--) {
        pre_top = top; pre_bot = bot;
        if (bot - top == 1) { r.match_len = length - j - 1; break; }
        int d = ctoi3(pat[j]);
        if (d > 2) { r.match_len = length - j - 1; bot = top; break; }
        u64 nt = orc_lf(ix, top, d), nb = orc_lf(ix, bot, d);
        if (C) C->n_ext++;
        top = nt; bot = nb;
        if (bot <= top) { r.match_len = length - j - 1; break; }
    }
    if (bot <= top) { r.sp = pre_top; r.ep = pre_bot; }
    else { r.match_len = length - j - 1; r.sp = top; r.ep = bot; }
    r.hits = r.ep - r.sp;
    return r;
}

// K4  count_hash_table  (bwt.h:1848-1952): fixed-length count, no early stop
seed_res count_fixed(const orc_index* ix, const char* pat, u64 length, orc_counters* C)
{
    seed_res r = {0, 0, 0, length};
    if (length < 17) return r;
    long long j = (long long)length - 16;
    u64 hv = hash16(pat + j);
    if (hv == (u64)-1) return r;
    u64 top, bot;
    orc_hash_query(ix, hv, &top, &bot);
    if (C) C->n_hash++;
    if (bot <= top) return r;
    for (j = j - 1; j >= 0; j--) {
        if (bot <= top) break;
        int d = ctoi3(pat[j]);
        if (d > 2) return r;
        u64 nt = orc_lf(ix, top, d), nb = orc_lf(ix, bot, d);
        if (C) C->n_ext++;
        top = nt; bot = nb;
    }
    r.sp = top; r.ep = bot;
    r.hits = bot <= top ? 0 : bot - top;
    return r;
}

extern "C" uint64_t orc_count_terminate(const orc_index* ix, const char* bsseq, uint64_t len, uint64_t* sp,
                                        uint64_t* ep, uint64_t* match_len)
{
    seed_res r = count_terminate(ix, bsseq, len, nullptr);
    *sp = r.sp; *ep = r.ep; *match_len = r.match_len;
    return r.hits;
}

// K5/K6: positions of the rows [sp,ep) as doubled-coordinate read-start sites
// (locate + reverse_and_adjust_site, bwt.cpp:4620 / Schema.cpp:4657; locate_one_position_direct,
// bwt.h:2585).  The FMtree traversal of the reference enumerates exactly {SA[r]}; callers sort.
void locate_rows(const orc_index* ix, u64 sp, u64 ep, u64 seed_len, u64 seed_off,
                        std::vector<u64>& out, orc_counters* C)
{
    for (u64 r = sp; r < ep; r++) {
        u64 nlf = 0;
        u64 p = orc_sa_row_counted(ix, r, &nlf);
        if (C) { C->n_lf += nlf; C->n_sa1++; C->n_locate_rows++; }
        out.push_back(ix->total - p - seed_len - seed_off);
    }
}

// determine_seed_offset_unmatch, Schema.h:1506-1531
int seed_offset_unmatch(int readLen, int pre, const char* read, int step)
{
    if (readLen - pre < 18 || readLen - pre < step) return readLen;
    int ret = pre + step;
    for (int i = 0; i < step; i++, pre++)
        if (read[pre] == 'N') return pre + 1;
    return ret;
}

// the reference's vote record (Schema.h:169-176)
static bool vote_gt(const vote_t& a, const vote_t& b) { return a.vote > b.vote; }   // compare_seed_votes, Schema.cpp:560

// generate_candidate_votes_shift, Schema.cpp:4687-4773
void make_votes(const std::vector<u64>& cand, u64 k, std::vector<vote_t>& votes)
{
    votes.clear();
    size_t n = cand.size(), i = 1;
    u64 pre = cand[0], vote = 1;
    auto emit = [&](u64 site, u64 v) { vote_t x; x.site = site; x.vote = v; x.err = 0; x.end_site = 0; votes.push_back(x); };
    while (i < n) {
        if (cand[i] == pre) { i++; vote++; }
        else { emit(pre < k ? 0 : pre - k, vote); vote = 1; pre = cand[i]; i++; }
    }
    emit(pre >= k ? pre - k : 0, vote);
}

// chromosome lookup + off-end rejection (output_sam_end_to_end, Schema.cpp:11941-11986)
static bool place(const orc_index* ix, u64 site, u64 start_site, u64 end_site, int* chrom, u64* pos, int* flag)
{
    u64 loc = site;
    if (loc >= ix->G) { loc = loc + end_site; loc = ix->G * 2 - loc - 1; *flag = 16; }
    else { loc = loc + start_site; *flag = 0; }
    size_t c = 0;
    for (; c < ix->chroms.size(); ++c)
        if (loc >= ix->chroms[c].start && loc <= ix->chroms[c].end) break;
    if (c == ix->chroms.size()) c = ix->chroms.size() - 1;   // reference reads past the table (UB); never hit by valid sites
    loc = loc + 1 - ix->chroms[c].start;
    *chrom = (int)c; *pos = loc;
    return !(loc + end_site - start_site > ix->chroms[c].len);
}

struct read_t { const char* seq; const char* qual; int len; };

// one read through Map_Single_Seq_end_to_end's loop body (Schema.cpp:24488-25119)
// orc_map_se_votes: when set, every general-path read appends its votes (site, count) in the visiting order of a9/a10
static std::vector<std::pair<u64, u64>>* g_vote_sink = nullptr;

static void map_one_se(const orc_index* ix, const orc_params* P, const read_t& rd, orc_rec* rec,
                       int64_t st[5], orc_counters* C, std::vector<u64>& cand, std::vector<vote_t>& votes,
                       std::vector<char>& win)
{
    const int L = rd.len;
    const char* read = rd.seq;
    memset(rec, 0, sizeof(*rec));
    st[0]++;
    if (C) C->n_reads++;
    // C_to_T_forward (Schema.h:1534)
    std::string bs(L, 0);
    int C_site = -1;
    for (int i = 0; i < L; i++) { bs[i] = read[L - 1 - i]; if (bs[i] == 'C') { C_site = i; bs[i] = 'T'; } }
    u64 k = (u64)(P->e_f * L);
    if (k >= 31) k = 31;
    u64 total_match = 0, seed_id = 0, max_seed = (u64)L / 10 - 1;
    if (max_seed > 25) max_seed = 25;
    cand.clear();
    int is_multiple = 0, get_error = -1, extra_seed_flag = 1;
    u64 first_seed_match = 0, one_mismatch_site = 0;
    const u64 max_hits = 1000, avail_len = (u64)P->seed_len;

    auto finish_unique = [&](u64 site, u64 end_site, u64 start_site, unsigned err, const char* cigar, int mapq, int score, int path) {
        int chrom, flag; u64 pos;
        bool ok = place(ix, site, start_site, end_site, &chrom, &pos, &flag);
        rec->site = site; rec->start_site = (int)start_site; rec->end_site = (int)end_site;
        rec->chrom = chrom; rec->pos = pos; rec->flag = flag; rec->mapq = mapq; rec->nm = (int)err;
        rec->score = score; rec->path = path;
        snprintf(rec->cigar, sizeof(rec->cigar), "%s", cigar);
        rec->status = ok ? 1 : 3;
        return ok;
    };

    if (seed_id < max_seed && total_match < (u64)L) {
        u64 cur_len = L - total_match;
        seed_res s = count_terminate(ix, bs.data(), cur_len, C);
        u64 match_length = s.match_len;
        first_seed_match = match_length;
        if (s.hits == 1) {
            // try_process_unique_mismatch_end_to_end (Schema.cpp:15164-15282)
            u64 nlf = 0;
            u64 p = orc_sa_row_counted(ix, s.sp, &nlf);
            if (C) { C->n_lf += nlf; C->n_sa1++; }
            u64 loc = ix->total - p - match_length - 0;
            cand.push_back(loc);
            int first_C_site = L - C_site - 1, error = 0;
            if (match_length > (u64)first_C_site) match_length = first_C_site;
            if (match_length != (u64)L) {
                int need = L - (int)match_length;
                win.assign(need + 8, 0);
                window_at(ix, loc + match_length, need, win.data());
                if (C) C->n_ungapped++;
                int read_i = (int)match_length;
                for (int i = 0; i < need; i++) {
                    if (read[read_i] != win[i] && !(read[read_i] == 'T' && win[i] == 'C')) {
                        error++;
                        if (error == 1) match_length = read_i; else break;
                    }
                    read_i++;
                }
            }
            get_error = error;
            if (error == 0) {
                char cg[32]; sprintf(cg, "%dM", L);
                if (finish_unique(loc, L - 1, 0, 0, cg, 42, 0, 1)) { st[1]++; st[3] += L; }
                return;
            }
        }
        one_mismatch_site = match_length;
        if (match_length == (u64)L && s.hits > 1) {
            is_multiple = 1;
            if (C_site == -1) {                                                      // ambiguous exact (fast exit B)
                rec->path = 4;
                if (!P->ambiguous_out) { rec->status = 2; st[2]++; return; }
                // output_ambiguous_exact_map (Schema.cpp:24072-24115): the first row, in SA order, whose placement
                // does not cross a chromosome end; MAPQ 1
                u64 nh = s.hits > max_hits ? max_hits : s.hits;
                char cg[32]; sprintf(cg, "%dM", L);
                rec->status = 3;
                for (u64 i = 0; i < nh; i++) {
                    u64 p = orc_sa_row(ix, s.sp + i);
                    if (finish_unique(ix->total - p - match_length, L - 1, 0, 0, cg, 1, 0, 4)) { rec->status = 2; st[2]++; break; }
                }
                return;
            }
        }
        if (s.hits == 1) { /* already located into cand[0] */ }
        else if (match_length >= avail_len && s.hits <= max_hits) {
            if (s.hits != 0) locate_rows(ix, s.sp, s.ep, match_length, total_match, cand, C);
        }
        if (match_length == 0) total_match = seed_offset_unmatch(L, (int)total_match, read, 8);
        else total_match = total_match + match_length / 2;
        seed_id++;
    }
    // 1-mismatch shortcut: second seed over the remaining suffix (Schema.cpp:24734-24801)
    if (get_error == 1) {
        u64 second_len = L - first_seed_match;
        if (second_len >= 17) {
            seed_res s = count_fixed(ix, bs.data(), second_len, C);
            if (s.hits == 1) { locate_rows(ix, s.sp, s.ep, second_len, first_seed_match, cand, C); extra_seed_flag = 0; }
            else if (s.hits <= max_hits) { if (s.hits != 0) locate_rows(ix, s.sp, s.ep, second_len, first_seed_match, cand, C); extra_seed_flag = 0; }
            else extra_seed_flag = 1;
        } else extra_seed_flag = 1;
    }
    if (extra_seed_flag == 1) {
        while (seed_id < max_seed && total_match < (u64)L) {
            u64 cur_len = L - total_match;
            seed_res s = count_terminate(ix, bs.data(), cur_len, C);
            u64 match_length = s.match_len;
            if (s.hits == 1) locate_rows(ix, s.sp, s.ep, match_length, total_match, cand, C);
            else if (match_length >= avail_len && s.hits <= max_hits) {
                if (s.hits != 0) locate_rows(ix, s.sp, s.ep, match_length, total_match, cand, C);
            } else if (cur_len == match_length) break;
            if (match_length == 0) total_match = seed_offset_unmatch(L, (int)total_match, read, 8);
            else total_match = total_match + match_length / 2;
            seed_id++;
        }
    }
    rec->n_cand = (int)cand.size();
    if (extra_seed_flag == 0 && (cand.size() == 1 || (cand.size() == 2 && cand[0] == cand[1]))) {
        // fast exit C (Schema.cpp:24894-24974)
        int score = 0;
        if (read[one_mismatch_site] == 'N') score -= P->np;
        else score -= mismatch_penalty(P, (unsigned char)rd.qual[one_mismatch_site]);
        int mapq = orc_mapq(P, (unsigned)-1, (unsigned)k, score);
        char cg[32]; sprintf(cg, "%dM", L);
        if (finish_unique(cand[0], L - 1, 0, 1, cg, mapq, score, 2)) { st[1]++; st[3] += L; st[4] += 1; }
        return;
    }
    if (cand.empty()) return;
    std::sort(cand.begin(), cand.end());
    make_votes(cand, k, votes);
    std::sort(votes.begin(), votes.end(), vote_gt);          // unstable, as the reference (Schema.cpp:24986)
    rec->n_votes = (int)votes.size();
    if (g_vote_sink) for (const vote_t& v : votes) g_vote_sink->push_back({v.site, v.vote});
    if (C) C->n_cand += votes.size();
    // K7+K8+K9: map_candidate_votes_mutiple_[cut_]end_to_end_* (Schema.cpp:7707-8183, 8202-8750)
    const int p_length = L + 2 * (int)k;
    unsigned min_err = ((unsigned)-1) - 1, second_best_diff = 0;
    long long min_err_index = -1;
    u64 min_err_site = (u64)-1;
    win.assign(p_length + 40, 0);
    for (size_t i = 0; i < votes.size(); i++) {
        window_at(ix, votes[i].site, p_length, win.data());
        unsigned e; int es = orc_bpm(win.data(), p_length, read, L, (int)k, &e);
        votes[i].err = e; votes[i].end_site = (u64)(long long)es;
        u64 tmp_site = votes[i].site + votes[i].end_site;
        if (e == min_err && min_err_site != tmp_site && min_err_index >= 0) {
            second_best_diff = 0; min_err_index = -2 - min_err_index;
            if (is_multiple && min_err == 0) {
                // the non-"cut" variant (used when the read matched exactly at >1 places) stops at
                // the second exact hit at a different end coordinate (Schema.cpp:8361 ... 8705)
                break;
            }
        } else if (e < min_err) {
            second_best_diff = min_err - e; min_err = e; min_err_index = (long long)i; min_err_site = tmp_site;
        }
    }
    if (min_err_index >= 0) {
        vote_t best = votes[min_err_index];
        int score = 0, start_site;
        char cigar[1024];
        unsigned nm = best.err;
        int end_site = (int)best.end_site;
        if (best.err != 0) {
            window_at(ix, best.site, p_length, win.data());
            if (C) C->n_sw++;
            orc_align(P, win.data(), p_length, read, rd.qual, L, (int)k, end_site, best.err, best.site < ix->G, 0,
                      &start_site, &end_site, &nm, &score, cigar);
        } else { start_site = end_site - L + 1; sprintf(cigar, "%dM", L); }
        int mapq = orc_mapq(P, second_best_diff, (unsigned)k, score);
        if (finish_unique(best.site, (u64)(long long)end_site, (u64)(long long)start_site, nm, cigar, mapq, score, 3)) {
            st[1]++; st[3] += L; st[4] += nm;
        }
    } else if (min_err_index != -1) {
        rec->path = 3;
        if (!P->ambiguous_out) { rec->status = 2; st[2]++; return; }
        // Schema.cpp:25095-25118: the first candidate that reached the minimum is aligned and reported (second_best_diff = 0)
        vote_t best = votes[-2 - min_err_index];
        int score = 0, start_site;
        char cigar[1024];
        unsigned nm = best.err;
        int end_site = (int)best.end_site;
        if (best.err != 0) {
            window_at(ix, best.site, p_length, win.data());
            if (C) C->n_sw++;
            orc_align(P, win.data(), p_length, read, rd.qual, L, (int)k, end_site, best.err, best.site < ix->G, 0,
                      &start_site, &end_site, &nm, &score, cigar);
        } else { start_site = end_site - L + 1; sprintf(cigar, "%dM", L); }
        int mapq = orc_mapq(P, second_best_diff, (unsigned)k, score);
        if (finish_unique(best.site, (u64)(long long)end_site, (u64)(long long)start_site, nm, cigar, mapq, score, 3)) { rec->status = 2; st[2]++; }
    }
}

extern "C" int orc_map_se(const orc_index* ix, const orc_params* P, const char* seq, const char* qual,
                          const int32_t* len, int stride, int64_t n, orc_rec* recs, int64_t stats[5],
                          orc_counters* counters)
{
    std::vector<u64> cand; std::vector<vote_t> votes; std::vector<char> win;
    int64_t st[5] = {0, 0, 0, 0, 0};
    if (counters) memset(counters, 0, sizeof(*counters));
    for (int64_t i = 0; i < n; i++) {
        read_t rd = {seq + (size_t)i * stride, qual + (size_t)i * stride, len[i]};
        map_one_se(ix, P, rd, &recs[i], st, counters, cand, votes, win);
    }
    // stats: reads, unique, ambiguous, mapped bases, error bases
    for (int j = 0; j < 5; j++) stats[j] = st[j];
    return 0;
}

/* orc_map_se plus the vote lists of the general-path reads (a9/a10: generate_candidate_votes_shift + std::sort by vote,
 * Schema.cpp:4687, 24978-24986): read i's votes are vote_site/vote_cnt[vote_off[i] .. vote_off[i+1]) in visiting order.
 * Returns the number of votes, or -1 when `cap` is too small. */
extern "C" int64_t orc_map_se_votes(const orc_index* ix, const orc_params* P, const char* seq, const char* qual, const int32_t* len,
                                    int stride, int64_t n, orc_rec* recs, int64_t stats[5], uint64_t* vote_site, uint32_t* vote_cnt,
                                    uint64_t* vote_off, int64_t cap)
{
    std::vector<u64> cand; std::vector<vote_t> votes; std::vector<char> win;
    std::vector<std::pair<u64, u64>> sink;
    int64_t st[5] = {0, 0, 0, 0, 0};
    g_vote_sink = &sink;
    for (int64_t i = 0; i < n; i++) {
        vote_off[i] = sink.size();
        read_t rd = {seq + (size_t)i * stride, qual + (size_t)i * stride, len[i]};
        map_one_se(ix, P, rd, &recs[i], st, nullptr, cand, votes, win);
    }
    vote_off[n] = sink.size();
    g_vote_sink = nullptr;
    for (int j = 0; j < 5; j++) stats[j] = st[j];
    if ((int64_t)sink.size() > cap) return -1;
    for (size_t j = 0; j < sink.size(); j++) { vote_site[j] = sink[j].first; vote_cnt[j] = (uint32_t)sink[j].second; }
    return (int64_t)sink.size();
}

// ------------------------------------------------------------------------------------------------
// FASTQ reader (inputReads_single_directly, Process_Reads.cpp:810-890) and SAM text
bool getline_(FILE* f, std::string& s)
{
    s.clear();
    int c;
    bool any = false;
    while ((c = fgetc(f)) != EOF) { any = true; if (c == '\n') break; s.push_back((char)c); }
    return any;
}
static inline char rc_char(char c) { return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c; }
// pbat (inputReads_single_directly_pbat, Process_Reads.cpp:986-1075): seq = reverse complement of the record, rseq = the
// record; every quality access of the pbat path is mirrored (Schema.cpp:15102 need_reverse_quality = 1, 26066).  That is
// the normal path run on the reverse-complemented read with reversed qualities, which is how it is restated here.
static bool read_fastq(FILE* f, fq_rec& r, bool cut_name, bool pbat = false)
{
    std::string plus;
    if (!getline_(f, r.name)) return false;
    getline_(f, r.seq); getline_(f, plus); getline_(f, r.qual);
    if (cut_name) for (size_t j = 0; j < r.name.size(); j++) if (r.name[j] == ' ' || r.name[j] == '/') { r.name.resize(j); break; }
    for (auto& ch : r.seq) ch = (char)toupper(ch);
    r.qual.resize(r.seq.size(), ' ');
    if (pbat) {
        std::string t(r.seq.rbegin(), r.seq.rend());
        for (auto& ch : t) ch = rc_char(ch);
        r.seq.swap(t);
        std::reverse(r.qual.begin(), r.qual.end());
    }
    r.rseq.assign(r.seq.rbegin(), r.seq.rend());
    for (auto& ch : r.rseq) ch = rc_char(ch);
    return true;
}

// OutPutSAM_Nounheader, Process_sam_out.cpp:1137-1153
void sam_header(FILE* o, const orc_index* ix, const char* argv_line)
{
    fprintf(o, "@HD\tVN:1.4\tSO:unsorted\n");
    for (auto& c : ix->chroms) fprintf(o, "@SQ\tSN:%s\tLN:%llu\n", c.name.c_str(), (unsigned long long)c.len);
    fprintf(o, "@PG\tID:BitMapperBS\tVN:1.0.2.3\tCL:%s\n", argv_line ? argv_line : "");
}

// output_sam_end_to_end text branch, Schema.cpp:11989-12039
static void sam_se(FILE* o, const orc_index* ix, const fq_rec& r, const orc_rec& m)
{
    const char* nm = r.name.c_str();
    if (nm[0] == '@') nm++;
    fprintf(o, "%s\t%d\t%s\t%llu\t%d\t%s\t*\t0\t0\t", nm, m.flag, ix->chroms[m.chrom].name.c_str(),
            (unsigned long long)m.pos, m.mapq, m.cigar);
    if (m.flag == 0) fprintf(o, "%s\t%s\t", r.seq.c_str(), r.qual.c_str());
    else { std::string q(r.qual.rbegin(), r.qual.rend()); fprintf(o, "%s\t%s\t", r.rseq.c_str(), q.c_str()); }
    fprintf(o, "NM:i:%d\n", m.nm);
}

extern "C" int orc_search_se(const orc_index* ix, const orc_params* P, const char* fastq, const char* out_sam,
                             const char* argv_line, int64_t stats[5])
{
    FILE* f = fopen(fastq, "rb");
    if (!f) return -1;
    FILE* o = fopen(out_sam, "wb");
    if (!o) { fclose(f); return -2; }
    sam_header(o, ix, argv_line);
    std::vector<u64> cand; std::vector<vote_t> votes; std::vector<char> win;
    int64_t st[5] = {0, 0, 0, 0, 0};
    fq_rec r; orc_rec m;
    while (read_fastq(f, r, true, P->pbat != 0)) {
        read_t rd = {r.seq.data(), r.qual.data(), (int)r.seq.size()};
        map_one_se(ix, P, rd, &m, st, nullptr, cand, votes, win);
        if (m.status == 1 || (m.status == 2 && P->ambiguous_out)) sam_se(o, ix, r, m);
        else if (P->unmapped_out && m.status != 2) {
            // output_sam_unmapped (Schema.cpp:23955-23975); the pbat path hands over the record as read from the file
            // (rseq and the un-mirrored qualities, Schema.cpp:25538-25543)
            const char* nm = r.name.c_str();
            if (nm[0] == '@') nm++;
            std::string q = r.qual;
            if (P->pbat) std::reverse(q.begin(), q.end());
            fprintf(o, "%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n", nm, P->pbat ? r.rseq.c_str() : r.seq.c_str(), q.c_str());
        }
    }
    fclose(f); fclose(o);
    for (int j = 0; j < 5; j++) stats[j] = st[j];
    return 0;
}
