// oracle/orc_pe.cpp -- TEST INFRASTRUCTURE (see bmbs_oracle.h).
// Paired-end restatement: Map_Pair_Seq_end_to_end_fast (Schema.cpp:18570-19546) and its helpers
// get_candidates (18172), verify_candidate_locations (18130) with the PE compaction of
// map_candidate_votes_mutiple_cut_end_to_end_8_for_paired_end (7334-7700), filter_pairs (16052),
// filter_pairs_single_side (16186), new_faster_verify_pairs (15773),
// calculate_best_map_cigar_end_to_end_return (14602), directly_output_read1/2 (10537, 11494).
#include "orc_internal.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

namespace {

struct cand_t { u64 site; unsigned err; u64 end_site; };

// get_candidates (Schema.cpp:18172-18565).  Returns best_mapp_occ: >0 direct hits (list holds them),
// -1 unverified candidates (list holds sites, sorted), 0 nothing.
int get_candidates(const orc_index* ix, const orc_params* P, const char* read, int L, u64 k,
                   std::vector<cand_t>& list, std::vector<u64>& cand, std::vector<vote_t>& votes, orc_counters* C)
{
    std::string bs(L, 0);
    int C_site = -1;
    for (int i = 0; i < L; i++) { bs[i] = read[L - 1 - i]; if (bs[i] == 'C') { C_site = i; bs[i] = 'T'; } }
    const u64 max_candidates_occ = 10000, max_hits = 1000, avail_len = (u64)P->seed_len;
    u64 total_match = 0, seed_id = 0, max_seed = (u64)L / 10 - 1;
    if (max_seed > 25) max_seed = 25;
    int get_error = -1, extra_seed_flag = 1;
    u64 first_seed_match = 0;
    cand.clear(); list.clear();
    std::vector<char> win;
    if (seed_id < max_seed && total_match < (u64)L) {
        u64 cur_len = L - total_match;
        seed_res s = count_terminate(ix, bs.data(), cur_len, C);
        u64 match_length = s.match_len;
        first_seed_match = match_length;
        if (s.hits == 1) {
            // try_process_unique_mismatch_end_to_end_return_site (Schema.cpp:15661-15760)
            u64 p = orc_sa_row(ix, s.sp);
            u64 loc = ix->total - p - match_length;
            cand.push_back(loc);
            int first_C_site = L - C_site - 1, error = 0;
            if (match_length > (u64)first_C_site) match_length = first_C_site;
            if (match_length != (u64)L) {
                int need = L - (int)match_length;
                win.assign(need + 8, 0);
                window_at(ix, loc + match_length, need, win.data());
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
            if (error == 0) { list.push_back({loc, 0u, (u64)(L - 1)}); return 1; }
        }
        if (match_length == (u64)L && s.hits > 1 && s.hits <= max_candidates_occ) {
            if (C_site == -1) {
                std::vector<u64> loc;
                locate_rows(ix, s.sp, s.ep, match_length, total_match, loc, C);
                std::sort(loc.begin(), loc.end());
                for (u64 x : loc) list.push_back({x, 0u, (u64)(L - 1)});
                return (int)loc.size();
            }
        }
        if (s.hits == 1) { /* cand[0] already holds it */ }
        else if (match_length >= avail_len && s.hits <= max_hits) { if (s.hits != 0) locate_rows(ix, s.sp, s.ep, match_length, total_match, cand, C); }
        if (match_length == 0) total_match = seed_offset_unmatch(L, (int)total_match, read, 8);
        else total_match = total_match + match_length / 2;
        seed_id++;
    }
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
            else if (match_length >= avail_len && s.hits <= max_hits) { if (s.hits != 0) locate_rows(ix, s.sp, s.ep, match_length, total_match, cand, C); }
            else if (cur_len == match_length) break;
            if (match_length == 0) total_match = seed_offset_unmatch(L, (int)total_match, read, 8);
            else total_match = total_match + match_length / 2;
            seed_id++;
        }
    }
    if (extra_seed_flag == 0 && (cand.size() == 1 || (cand.size() == 2 && cand[0] == cand[1]))) {
        list.push_back({cand[0], 1u, (u64)(L - 1)});
        return 1;
    }
    if (!cand.empty()) {
        std::sort(cand.begin(), cand.end());
        make_votes(cand, k, votes);                 // site-sorted; the PE path does NOT re-sort by vote
        for (auto& v : votes) list.push_back({v.site, 0u, 0});
        return -1;
    }
    return 0;
}

// verify_candidate_locations + the PE compaction (Schema.cpp:7334-7700): keep err <= k whose
// site+end_site differs from the PREVIOUS candidate's (kept or not)
int verify(const orc_index* ix, const char* read, int L, u64 k, std::vector<cand_t>& list, u64 n, orc_counters* C)
{
    const int p_length = L + 2 * (int)k;
    std::vector<char> win(p_length + 40, 0);
    u64 pre = (u64)-1;
    int occ = 0;
    for (u64 i = 0; i < n; i++) {
        window_at(ix, list[i].site, p_length, win.data());
        unsigned e; int es = orc_bpm(win.data(), p_length, read, L, (int)k, &e);
        if (C) C->n_cand++;
        list[i].err = e; list[i].end_site = (u64)(long long)es;
        u64 t = list[i].site + list[i].end_site;
        if (e <= k && pre != t) { list[occ] = list[i]; occ++; }
        pre = t;
    }
    return occ;
}

// filter_pairs (Schema.cpp:16052-16180)
void filter_pairs(std::vector<cand_t>& a, u64& na, std::vector<cand_t>& b, u64& nb, long long maxd, long long mind)
{
    std::vector<cand_t> ra, rb;
    long long first = 0;
    for (long long i = 0; i < (long long)na; i++) {
        for (long long j = first; j < (long long)nb; j++) {
            bool hit = false;
            if (a[i].site > b[j].site) {
                long long d = (long long)(a[i].site - b[j].site);
                if (d > maxd) first = j + 1;
                else if (d >= mind) hit = true;
            } else {
                long long d = (long long)(b[j].site - a[i].site);
                if (d > maxd) break;
                if (d <= maxd && d >= mind) hit = true;
            }
            if (hit) {
                if (ra.empty() || a[i].site > ra.back().site) ra.push_back(a[i]);
                if (rb.empty() || b[j].site > rb.back().site) rb.push_back(b[j]);
            }
        }
    }
    a.swap(ra); b.swap(rb);
    na = a.size(); nb = b.size();
}

// filter_pairs_single_side (Schema.cpp:16186-16270): prune b by the verified list a (in place)
void filter_single(const std::vector<cand_t>& a, u64 na, std::vector<cand_t>& b, u64& nb, long long maxd, long long mind)
{
    u64 len2 = 0;
    long long first = 0;
    for (long long i = 0; i < (long long)na; i++) {
        for (long long j = first; j < (long long)nb; j++) {
            if (a[i].site > b[j].site) {
                long long d = (long long)(a[i].site - b[j].site);
                if (d > maxd) first = j + 1;
                else if (d >= mind) { b[len2] = b[j]; len2++; first = j + 1; }
            } else {
                long long d = (long long)(b[j].site - a[i].site);
                if (d > maxd) break;
                if (d <= maxd && d >= mind) { b[len2] = b[j]; len2++; first = j + 1; }
            }
        }
    }
    nb = len2;
}

// new_faster_verify_pairs (Schema.cpp:15773-15900)
int verify_pairs(const std::vector<cand_t>& r1, int n1, const std::vector<cand_t>& r2, int n2, int error_threshold,
                 long long* b1, long long* b2, long long maxd, long long mind, unsigned* sbd)
{
    int mapping_pair = 0;
    int best_sum_err = 4 * error_threshold + 2;
    long long second_best_err = (long long)best_sum_err * 2, bi = 0, bj = 0;
    *sbd = 0;
    if (n1 > 0 && n2 > 0) {
        long long first = 0;
        for (int i = 0; i < n1; i++) {
            for (long long j = first; j < n2; j++) {
                bool hit = false;
                if (r1[i].site > r2[j].site) {
                    long long d = (long long)(r1[i].site - r2[j].site);
                    if (d > maxd) first = j + 1;
                    else if (d >= mind) hit = true;
                } else {
                    long long d = (long long)(r2[j].site - r1[i].site);
                    if (d > maxd) break;
                    if (d <= maxd && d >= mind) hit = true;
                }
                if (hit) {
                    long long cur = (long long)r1[i].err + (long long)r2[j].err;
                    if (cur < best_sum_err) { second_best_err = best_sum_err; best_sum_err = (int)cur; bi = i; bj = j; mapping_pair = 1; }
                    else if (cur == best_sum_err) {
                        second_best_err = best_sum_err; mapping_pair++;
                        if (best_sum_err == 0) { *sbd = 0; *b1 = bi; *b2 = bj; return mapping_pair; }
                    }
                }
            }
        }
    }
    if (mapping_pair != 0) *sbd = (unsigned)(second_best_err - best_sum_err);
    *b1 = bi; *b2 = bj;
    return mapping_pair;
}

struct mate_res { u64 origin_site, end_site, site; unsigned err; int score, flag, chrom, matched; char cigar[1024]; };

// output_sam_end_to_end_return (Schema.cpp:9188-9225)
void place_return(const orc_index* ix, u64 site, u64 end_site, u64 start_site, mate_res* r)
{
    u64 loc = site;
    if (loc >= ix->G) { loc = loc + end_site; loc = ix->G * 2 - loc - 1; r->flag = 16; }
    else { loc = loc + start_site; r->flag = 0; }
    size_t c = 0;
    for (; c < ix->chroms.size(); ++c) if (loc >= ix->chroms[c].start && loc <= ix->chroms[c].end) break;
    if (c == ix->chroms.size()) c = ix->chroms.size() - 1;
    r->chrom = (int)c; r->site = loc + 1 - ix->chroms[c].start;
}

// the per-mate post-processing of Schema.cpp:19330-19400
void finish_mate(const orc_index* ix, const orc_params* P, const char* read, const char* qual, int L, u64 k,
                 const cand_t& c, int reverse_quality, mate_res* r, orc_counters* C)
{
    r->err = c.err; r->origin_site = c.site; r->end_site = c.end_site;
    int start_site;
    if (r->err != 0) {
        const int p_length = L + 2 * (int)k;
        std::vector<char> win(p_length + 40, 0);
        window_at(ix, c.site, p_length, win.data());
        int ne; unsigned nm;
        if (C) C->n_sw++;
        orc_align(P, win.data(), p_length, read, qual, L, (int)k, (int)c.end_site, c.err, c.site < ix->G, reverse_quality,
                  &start_site, &ne, &nm, &r->score, r->cigar);
        r->end_site = (u64)(long long)ne; r->err = nm;
    } else {
        r->score = 0;
        start_site = (int)c.end_site - L + 1;
        sprintf(r->cigar, "%dM", L);
    }
    place_return(ix, r->origin_site, r->end_site, (u64)(long long)start_site, r);
    r->matched = (int)r->end_site - start_site + 1;
}

// ---- --sensitive: Map_Pair_Seq_end_to_end (Schema.cpp:19953-21459) -------------------------------
// state of one mate after its first seed (Schema.cpp:20395-20600 / 20602-20862)
struct mate_t {
    const char* read; int L; u64 k;
    std::string bs; int C_site;
    int jump = 0, occ = 0, get_error = -1;
    u64 total_match = 0, seed_id = 0, max_seed = 0, first_seed_match = 0;
    std::vector<u64> cand;                 // candidates1 after the first seed
    std::vector<cand_t> list;              // candidates_votes
    std::vector<int> seed_start, seed_len; // read1_seed_start / read1_seed_length[0 .. full_seed_id)
};

void first_seed(const orc_index* ix, const orc_params* P, mate_t& m, orc_counters* C)
{
    const int L = m.L; const char* read = m.read;
    m.bs.assign(L, 0); m.C_site = -1;
    for (int i = 0; i < L; i++) { m.bs[i] = read[L - 1 - i]; if (m.bs[i] == 'C') { m.C_site = i; m.bs[i] = 'T'; } }
    m.max_seed = (u64)L / 10 - 1; if (m.max_seed > 25) m.max_seed = 25;
    const u64 max_candidates_occ = 10000, max_hits = 1000, avail_len = (u64)P->seed_len;
    std::vector<char> win;
    if (m.seed_id < m.max_seed && m.total_match < (u64)L) {
        u64 cur_len = L - m.total_match;
        seed_res s = count_terminate(ix, m.bs.data(), cur_len, C);
        u64 match_length = s.match_len;
        m.first_seed_match = match_length;
        m.seed_start.push_back((int)m.total_match); m.seed_len.push_back((int)match_length);
        if (s.hits == 1) {
            u64 p = orc_sa_row(ix, s.sp);
            u64 loc = ix->total - p - match_length;
            m.cand.push_back(loc);
            int first_C_site = L - m.C_site - 1, error = 0;
            if (match_length > (u64)first_C_site) match_length = first_C_site;
            if (match_length != (u64)L) {
                int need = L - (int)match_length;
                win.assign(need + 8, 0);
                window_at(ix, loc + match_length, need, win.data());
                int read_i = (int)match_length;
                for (int i = 0; i < need; i++) {
                    if (read[read_i] != win[i] && !(read[read_i] == 'T' && win[i] == 'C')) { error++; if (error == 1) match_length = read_i; else break; }
                    read_i++;
                }
            }
            m.get_error = error;
            if (error == 0) { m.list.push_back({loc, 0u, (u64)(L - 1)}); m.occ = 1; m.jump = 1; m.cand.clear(); return; }
        }
        if (match_length == (u64)L && s.hits > 1 && s.hits <= max_candidates_occ && m.C_site == -1) {
            std::vector<u64> loc;
            locate_rows(ix, s.sp, s.ep, match_length, m.total_match, loc, C);
            std::sort(loc.begin(), loc.end());
            for (u64 x : loc) m.list.push_back({x, 0u, (u64)(L - 1)});
            m.occ = (int)loc.size(); m.jump = 1; m.cand.clear();
            return;
        }
        if (s.hits == 1) { /* cand[0] holds it */ }
        else if (match_length >= avail_len && s.hits <= max_hits) { if (s.hits != 0) locate_rows(ix, s.sp, s.ep, match_length, m.total_match, m.cand, C); }
        else { m.seed_start.pop_back(); m.seed_len.pop_back(); }           // full_seed_id--
        if (match_length == 0) m.total_match = seed_offset_unmatch(L, (int)m.total_match, read, 8);
        else m.total_match = m.total_match + match_length / 2;
        m.seed_id++;
    }
}

// votes of a sorted candidate array, optionally kept only when a verified hit of the mate lies within the insert
// window (generate_candidate_votes_shift[_filter] + select_suit_candidates, Schema.cpp:4687, 4884, 4775)
void votes_filtered(const std::vector<u64>& cand, u64 k, const std::vector<cand_t>* mate, int mate_occ, long long maxd, long long mind,
                    std::vector<cand_t>& out)
{
    out.clear();
    int next_start = 0;
    auto suit = [&](u64 site) -> bool {
        if (!mate) return true;
        for (int i = next_start; i < mate_occ; i++) {
            if ((*mate)[i].site > site) {
                long long d = (long long)((*mate)[i].site - site);
                if (d > maxd) return false;
                if (d <= maxd && d >= mind) return true;
            } else {
                long long d = (long long)(site - (*mate)[i].site);
                if (d > maxd) next_start = i + 1;
                else if (d >= mind) return true;
            }
        }
        return false;
    };
    size_t n = cand.size(), i = 1;
    u64 pre = cand[0];
    while (i < n) {
        if (cand[i] == pre) i++;
        else { u64 site = pre < k ? 0 : pre - k; if (suit(site)) out.push_back({site, 0u, 0}); pre = cand[i]; i++; }
    }
    u64 site = pre >= k ? pre - k : 0;
    if (suit(site)) out.push_back({site, 0u, 0});
}

// process_rest_seed_debug / process_rest_seed_filter_debug (Schema.cpp:17574, 16298)
int process_rest(const orc_index* ix, const orc_params* P, mate_t& m, const std::vector<cand_t>* mate, int mate_occ,
                 long long maxd, long long mind, orc_counters* C)
{
    const int L = m.L; const char* read = m.read;
    const u64 max_hits = 1000, avail_len = (u64)P->seed_len;
    int extra = 1;
    u64 total_match = m.total_match, seed_id = m.seed_id;      // by-value parameters of the reference
    if (m.get_error == 1) {
        u64 second_len = L - m.first_seed_match;
        if (second_len >= 17) {
            seed_res s = count_fixed(ix, m.bs.data(), second_len, C);
            if (s.hits == 1) { locate_rows(ix, s.sp, s.ep, second_len, m.first_seed_match, m.cand, C); extra = 0; }
            else if (s.hits <= max_hits) { if (s.hits != 0) locate_rows(ix, s.sp, s.ep, second_len, m.first_seed_match, m.cand, C); extra = 0; }
        }
    }
    if (extra == 1) {
        while (seed_id < m.max_seed && total_match < (u64)L) {
            u64 cur_len = L - total_match;
            seed_res s = count_terminate(ix, m.bs.data(), cur_len, C);
            u64 ml = s.match_len;
            m.seed_start.push_back((int)total_match); m.seed_len.push_back((int)ml);
            if (s.hits == 1) locate_rows(ix, s.sp, s.ep, ml, total_match, m.cand, C);
            else if (ml >= avail_len && s.hits <= max_hits) { if (s.hits != 0) locate_rows(ix, s.sp, s.ep, ml, total_match, m.cand, C); }
            else { m.seed_start.pop_back(); m.seed_len.pop_back(); if (cur_len == ml) break; }
            if (ml == 0) total_match = seed_offset_unmatch(L, (int)total_match, read, 8);
            else total_match = total_match + ml / 2;
            seed_id++;
        }
    }
    if (extra == 0 && (m.cand.size() == 1 || (m.cand.size() == 2 && m.cand[0] == m.cand[1]))) {
        m.list.clear(); m.list.push_back({m.cand[0], 1u, (u64)(L - 1)});
        return 1;
    }
    if (!m.cand.empty()) {
        std::sort(m.cand.begin(), m.cand.end());
        votes_filtered(m.cand, m.k, mate, mate_occ, maxd, mind, m.list);
        return verify(ix, read, L, m.k, m.list, m.list.size(), C);
    }
    return 0;
}

// reseed_filter + select_best_seeds (Schema.cpp:16678, 16630)
int reseed(const orc_index* ix, const orc_params* P, mate_t& m, const std::vector<cand_t>& mate, int mate_occ, long long maxd,
           long long mind, orc_counters* C)
{
    (void)P;
    const int L = m.L; const char* read = m.read;
    const u64 avail_len = 20, max_hits = 1000, match_step = 8;
    const int full = (int)m.seed_start.size();
    int rs[4], rl[4], rn = 0;
    // select_best_seeds
    if (full >= 2) {
        rn = 2;
        rs[0] = m.seed_start[0]; rl[0] = m.seed_start[1] - m.seed_start[0];
        rs[1] = m.seed_start[full - 2] + m.seed_len[full - 2]; rl[1] = L - rs[1];
    } else if (full == 1) {
        rn = 2;
        rs[0] = m.seed_start[0]; rl[0] = L / 2;
        rs[1] = rs[0] + rl[0]; rl[1] = L - rs[1];
    }
    // With no recorded seed (full == 0) the reference indexes both malloc'ed int arrays at [-1] (Schema.cpp:16657;
    // full_seed_id is unsigned).  That word is the upper half of glibc's chunk-size field, i.e. 0, so the reference
    // deterministically adds the whole read (0, L) as one fixed seed -- which also consumes one seed_id.  Reproduced.
    const int last_start = full >= 1 ? m.seed_start[full - 1] : 0, last_len = full >= 1 ? m.seed_len[full - 1] : 0;
    if (last_start + last_len < L) { rs[rn] = last_start + last_len; rl[rn] = L - rs[rn]; rn++; }
    std::vector<u64> cand;
    u64 seed_id = 0, total_match = 0;
    while ((int)seed_id < rn) {
        total_match = (u64)rs[seed_id];
        const u64 cur_len = (u64)L - total_match, ml = (u64)rl[seed_id];
        seed_res s = count_fixed(ix, m.bs.data() + (cur_len - ml), ml, C);
        if (s.hits == 1) locate_rows(ix, s.sp, s.ep, ml, total_match, cand, C);
        else if (ml >= avail_len && s.hits <= max_hits) { if (s.hits != 0) locate_rows(ix, s.sp, s.ep, ml, total_match, cand, C); }
        else if (cur_len == ml) break;
        seed_id++;
    }
    total_match = full > 1 ? (u64)((m.seed_start[0] + m.seed_start[1]) / 2) : match_step / 2;
    while (seed_id < m.max_seed && total_match < (u64)L) {
        const u64 cur_len = (u64)L - total_match;
        seed_res s = count_terminate(ix, m.bs.data(), cur_len, C);
        const u64 ml = s.match_len;
        if (s.hits == 1) locate_rows(ix, s.sp, s.ep, ml, total_match, cand, C);
        else if (ml >= avail_len && s.hits <= max_hits) { if (s.hits != 0) locate_rows(ix, s.sp, s.ep, ml, total_match, cand, C); }
        else if (cur_len == ml) break;
        total_match += match_step;
        seed_id++;
    }
    if (cand.empty()) return 0;
    std::sort(cand.begin(), cand.end());
    votes_filtered(cand, m.k, &mate, mate_occ, maxd, mind, m.list);
    return verify(ix, read, L, m.k, m.list, m.list.size(), C);
}

static inline char rc_char(char c) { return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c; }

}  // namespace

// one pair through the loop body of Map_Pair_Seq_end_to_end_fast.  seq2 is mate 2 as the reader
// hands it over (= reverse complement of the FASTQ record, Process_Reads.cpp:262-267); qual2 is in
// FASTQ order.  st: pairs, unique, ambiguous, bases, errors.
static void map_one_pe(const orc_index* ix, const orc_params* P, const char* seq1, const char* qual1, int L1,
                       const char* seq2, const char* qual2, int L2, orc_pe_rec* rec, int64_t st[5], orc_counters* C)
{
    memset(rec, 0, sizeof(*rec));
    st[0]++;
    u64 k1 = (u64)(P->e_f * L1); if (k1 >= 31) k1 = 31;
    u64 k2 = (u64)(P->e_f * L2); if (k2 >= 31) k2 = 31;
    const u64 large = k1 > k2 ? k1 : k2;
    const int max_length = L1 > L2 ? L1 : L2;
    const long long inner_max = P->max_ins + (long long)large * 2;
    const long long inner_min = P->min_ins - (long long)large * 2 - max_length;
    std::vector<cand_t> l1, l2;
    int occ1, occ2;
    if (P->sensitive) {
        mate_t m1, m2;
        m1.read = seq1; m1.L = L1; m1.k = k1; m2.read = seq2; m2.L = L2; m2.k = k2;
        first_seed(ix, P, m1, C);
        first_seed(ix, P, m2, C);
        occ1 = m1.occ; occ2 = m2.occ;
        // the mate with fewer first-seed candidates is finished and verified first (Schema.cpp:20870-21030)
        if (m1.cand.size() <= m2.cand.size()) {
            if (!m1.jump) occ1 = process_rest(ix, P, m1, nullptr, 0, inner_max, inner_min, C);
            if (occ1 == 0) return;
            if (!m2.jump) occ2 = process_rest(ix, P, m2, &m1.list, occ1, inner_max, inner_min, C);
        } else {
            if (!m2.jump) occ2 = process_rest(ix, P, m2, nullptr, 0, inner_max, inner_min, C);
            if (occ2 == 0) return;
            if (!m1.jump) occ1 = process_rest(ix, P, m1, &m2.list, occ2, inner_max, inner_min, C);
        }
        if (occ2 == 0) occ2 = reseed(ix, P, m2, m1.list, occ1, inner_max, inner_min, C);
        else if (occ1 == 0) occ1 = reseed(ix, P, m1, m2.list, occ2, inner_max, inner_min, C);
        l1.swap(m1.list); l2.swap(m2.list);
    } else {
    std::vector<u64> cand; std::vector<vote_t> votes;
    occ1 = get_candidates(ix, P, seq1, L1, k1, l1, cand, votes, C);
    occ2 = get_candidates(ix, P, seq2, L2, k2, l2, cand, votes, C);
    u64 n1 = l1.size(), n2 = l2.size();
    if (occ1 > 0 && occ2 > 0) { occ1 = (int)n1; occ2 = (int)n2; }
    else {
        if (occ1 == 0 || occ2 == 0) return;
        filter_pairs(l1, n1, l2, n2, inner_max, inner_min);
        if (n1 == 0 || n2 == 0) return;
        if (occ1 == -1 && occ2 == -1) {
            if (n1 <= n2) {
                occ1 = verify(ix, seq1, L1, k1, l1, n1, C);
                if (occ1 == 0) return;
                filter_single(l1, (u64)occ1, l2, n2, inner_max, inner_min);
                occ2 = verify(ix, seq2, L2, k2, l2, n2, C);
            } else {
                occ2 = verify(ix, seq2, L2, k2, l2, n2, C);
                if (occ2 == 0) return;
                filter_single(l2, (u64)occ2, l1, n1, inner_max, inner_min);
                occ1 = verify(ix, seq1, L1, k1, l1, n1, C);
            }
        } else if (occ1 != -1) {
            if ((long long)n1 < occ1) occ1 = (int)n1;
            if (occ2 == -1) occ2 = verify(ix, seq2, L2, k2, l2, n2, C);
        } else if (occ2 != -1) {
            if ((long long)n2 < occ2) occ2 = (int)n2;
            if (occ1 == -1) occ1 = verify(ix, seq1, L1, k1, l1, n1, C);
        }
    }
    }
    long long b1 = 0, b2 = 0;
    unsigned sbd = 0;
    int mapping_pair = verify_pairs(l1, occ1, l2, occ2, (int)large, &b1, &b2, inner_max, inner_min, &sbd);
    rec->n_pairs = mapping_pair;
    if (mapping_pair == 1 || (P->ambiguous_out && mapping_pair > 1)) {        // Schema.cpp:19342-19347
        mate_res r1, r2;
        finish_mate(ix, P, seq1, qual1, L1, k1, l1[b1], 0, &r1, C);
        finish_mate(ix, P, seq2, qual2, L2, k2, l2[b2], 1, &r2, C);
        // calculate_TLEN (Schema.h:1587)
        long long mn = (long long)r1.site, mx = (long long)r1.site + r1.matched - 1;
        if ((long long)r1.site > (long long)r2.site) mn = (long long)r2.site;
        if (mx < (long long)r2.site + r2.matched - 1) mx = (long long)r2.site + r2.matched - 1;
        const int tlen = (int)(mx - mn + 1);
        if (tlen <= P->max_ins && tlen >= P->min_ins &&
            r1.site + (u64)r1.matched <= ix->chroms[r1.chrom].len + 1 && r2.site + (u64)r2.matched <= ix->chroms[r2.chrom].len + 1) {
            if (mapping_pair == 1) { st[1]++; rec->status = 1; } else { st[2]++; rec->status = 2; }
            st[3] += L1 + L2; st[4] += r1.err + r2.err;
            rec->mapq = orc_mapq(P, sbd, (unsigned)(k1 + k2), r1.score + r2.score);
            rec->tlen = tlen;
            // directly_output_read1 / read2 flags (Schema.cpp:10552-10560, 11503-11511)
            rec->flag1 = r1.flag == 0 ? (1 | 2 | 32 | 64) : (1 | 2 | 16 | 64);
            rec->flag2 = r2.flag == 0 ? (1 | 2 | 16 | 128) : (1 | 2 | 32 | 128);
            rec->chrom1 = r1.chrom; rec->chrom2 = r2.chrom; rec->pos1 = r1.site; rec->pos2 = r2.site;
            rec->nm1 = (int)r1.err; rec->nm2 = (int)r2.err; rec->score1 = r1.score; rec->score2 = r2.score;
            rec->matched1 = r1.matched; rec->matched2 = r2.matched;
            snprintf(rec->cigar1, sizeof(rec->cigar1), "%s", r1.cigar);
            snprintf(rec->cigar2, sizeof(rec->cigar2), "%s", r2.cigar);
        } else rec->status = 3;
    } else if (mapping_pair > 1) { st[2]++; rec->status = 2; }
}

extern "C" int orc_map_pe(const orc_index* ix, const orc_params* P, const char* seq1, const char* qual1, const char* seq2,
                          const char* qual2, int L1, int L2, int stride, int64_t n, orc_pe_rec* recs, int64_t stats[5],
                          orc_counters* counters)
{
    int64_t st[5] = {0, 0, 0, 0, 0};
    if (counters) memset(counters, 0, sizeof(*counters));
    for (int64_t i = 0; i < n; i++)
        map_one_pe(ix, P, seq1 + (size_t)i * stride, qual1 + (size_t)i * stride, L1, seq2 + (size_t)i * stride,
                   qual2 + (size_t)i * stride, L2, &recs[i], st, counters);
    for (int j = 0; j < 5; j++) stats[j] = st[j];
    return 0;
}

// the same with per-pair mate lengths (mates of one pair may differ, Schema.cpp:18900-18935 uses both lengths)
extern "C" int orc_map_pe_var(const orc_index* ix, const orc_params* P, const char* seq1, const char* qual1, const char* seq2,
                              const char* qual2, const int32_t* len1, const int32_t* len2, int stride, int64_t n,
                              orc_pe_rec* recs, int64_t stats[5], orc_counters* counters)
{
    int64_t st[5] = {0, 0, 0, 0, 0};
    if (counters) memset(counters, 0, sizeof(*counters));
    for (int64_t i = 0; i < n; i++)
        map_one_pe(ix, P, seq1 + (size_t)i * stride, qual1 + (size_t)i * stride, len1[i], seq2 + (size_t)i * stride,
                   qual2 + (size_t)i * stride, len2[i], &recs[i], st, counters);
    for (int j = 0; j < 5; j++) stats[j] = st[j];
    return 0;
}

// inputReads_paired_directly (Process_Reads.cpp:155-317) + the two record writers
extern "C" int orc_search_pe(const orc_index* ix, const orc_params* P, const char* fq1, const char* fq2,
                             const char* out_sam, const char* argv_line, int64_t stats[5])
{
    if (P->pbat) { const char* t = fq1; fq1 = fq2; fq2 = t; }               // exchange_two_reads (Process_Reads.cpp:1628)
    FILE* f1 = fopen(fq1, "rb"); FILE* f2 = fopen(fq2, "rb");
    if (!f1 || !f2) return -1;
    FILE* o = fopen(out_sam, "wb");
    if (!o) return -2;
    sam_header(o, ix, argv_line);
    int64_t st[5] = {0, 0, 0, 0, 0};
    std::string n1, s1, p1, q1, n2, s2, p2, q2;
    orc_pe_rec m;
    while (getline_(f1, n1) && getline_(f2, n2)) {
        getline_(f1, s1); getline_(f1, p1); getline_(f1, q1);
        getline_(f2, s2); getline_(f2, p2); getline_(f2, q2);
        for (auto& c : s1) c = (char)toupper(c);
        for (auto& c : s2) c = (char)toupper(c);
        std::string r1(s1.rbegin(), s1.rend()); for (auto& c : r1) c = rc_char(c);           // rseq of mate 1
        std::string seq2(s2.rbegin(), s2.rend()); for (auto& c : seq2) c = rc_char(c);       // seq of mate 2 = revcomp(raw)
        q1.resize(s1.size(), ' '); q2.resize(s2.size(), ' ');
        // name: cut at the first differing char, ' ' or '/'
        size_t j = 0;
        for (; j < n1.size() && j < n2.size(); j++) if (n1[j] != n2[j] || n1[j] == ' ' || n1[j] == '/') break;
        std::string nm = n1.substr(0, j);
        if (j == n1.size() || j == n2.size()) nm = n1.substr(0, std::min(n1.size(), n2.size()));
        const char* name = nm.c_str(); if (name[0] == '@') name++;
        map_one_pe(ix, P, s1.data(), q1.data(), (int)s1.size(), seq2.data(), q2.data(), (int)s2.size(), &m, st, nullptr);
        if (m.status == 0 || m.status == 3) {
            // directly_output_unmapped_PE (Schema.cpp:10392-10430): flags 77 / 141, mate 2 as in its FASTQ record
            if (P->unmapped_out) {
                fprintf(o, "%s\t77\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n", name, s1.c_str(), q1.c_str());
                fprintf(o, "%s\t141\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n", name, s2.c_str(), q2.c_str());
            }
            continue;
        }
        if (m.status == 2 && !P->ambiguous_out) continue;
        // TLEN sign: Schema.cpp:10575-10600 (read 1) and 11530-11555 (read 2)
        const char* t1 = m.pos2 < m.pos1 ? "-" : "";          // read 1: negative only when the mate lies to the left
        const char* t2 = m.pos1 > m.pos2 ? "" : "-";          // read 2: positive only when the mate lies to the right
        std::string rq1(q1.rbegin(), q1.rend()), rq2(q2.rbegin(), q2.rend());
        fprintf(o, "%s\t%d\t%s\t%llu\t%d\t%s\t=\t%llu\t%s%d\t%s\t%s\tNM:i:%d\n", name, m.flag1, ix->chroms[m.chrom1].name.c_str(),
                (unsigned long long)m.pos1, m.mapq, m.cigar1, (unsigned long long)m.pos2, t1, m.tlen,
                (m.flag1 & 32) ? s1.c_str() : r1.c_str(), (m.flag1 & 32) ? q1.c_str() : rq1.c_str(), m.nm1);
        fprintf(o, "%s\t%d\t%s\t%llu\t%d\t%s\t=\t%llu\t%s%d\t%s\t%s\tNM:i:%d\n", name, m.flag2, ix->chroms[m.chrom2].name.c_str(),
                (unsigned long long)m.pos2, m.mapq, m.cigar2, (unsigned long long)m.pos1, t2, m.tlen,
                (m.flag2 & 16) ? seq2.c_str() : s2.c_str(), (m.flag2 & 16) ? rq2.c_str() : q2.c_str(), m.nm2);
    }
    fclose(f1); fclose(f2); fclose(o);
    for (int j = 0; j < 5; j++) stats[j] = st[j];
    return 0;
}
