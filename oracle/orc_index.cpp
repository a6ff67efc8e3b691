// oracle/orc_index.cpp -- TEST INFRASTRUCTURE (see bmbs_oracle.h).
// Index build (naive suffix sort) + load in the reference's on-disk formats, and the FM-index
// primitives restated from bwt.h.  File formats: SURVEY.md §2b.
#include "orc_internal.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------------
// FASTA -> chromosome table + upper-cased forward genome.
// Restates loadRefGenome (Ref_Genome.cpp:28-96): name = header up to the first ' ' or '\n';
// every non-space character is kept, upper-cased.
static bool read_fasta(const char* path, std::vector<orc_chrom>& chroms, std::string& gen)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    std::string line;
    int c;
    bool in_header = false;
    std::string hdr;
    while ((c = fgetc(f)) != EOF) {
        if (in_header) {
            if (c == '\n') {
                in_header = false;
                size_t k = 0;
                while (k < hdr.size() && hdr[k] != ' ') k++;
                orc_chrom ch; ch.name = hdr.substr(0, k); ch.len = 0; ch.start = ch.end = 0;
                chroms.push_back(ch);
            } else hdr.push_back((char)c);
        } else if (c == '>') {
            in_header = true; hdr.clear();
        } else if (!isspace(c)) {
            gen.push_back((char)toupper(c));
            if (!chroms.empty()) chroms.back().len++;
        }
    }
    fclose(f);
    return !chroms.empty();
}

// replace_N (Index.cpp:696-729) uses srand(time(0)): non-deterministic by construction.  We use a
// fixed LCG so that our builds are reproducible; N-free inputs (all synthetic genomes) are
// unaffected, which is the only case where byte parity with a reference-built index is defined.
static void replace_non_acgt(std::string& g)
{
    uint64_t s = 0x2545F4914F6CDD1DULL;
    for (size_t i = 0; i < g.size(); i++) {
        char ch = g[i];
        if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            g[i] = "ACGT"[(s >> 33) & 3];
        }
    }
}

template <class T> static void wr(FILE* f, const T& v) { fwrite(&v, sizeof(T), 1, f); }

// ------------------------------------------------------------------------------------------------
int orc_build_from_genome(const std::vector<orc_chrom>& chroms_in, const std::string& gen_in,
                          const char* prefix)
{
    std::vector<orc_chrom> chroms = chroms_in;
    std::string gen = gen_in;
    const u64 G = gen.size();
    replace_non_acgt(gen);
    std::string base = std::string(prefix) + ".index";

    // <prefix>.index : chromosome table (initSavingIHashTable, Index.cpp:134-159)
    {
        FILE* f = fopen(base.c_str(), "wb");
        if (!f) return -1;
        wr<u64>(f, chroms.size());
        for (auto& c : chroms) {
            wr<u64>(f, c.name.size());
            fwrite(c.name.data(), 1, c.name.size(), f);
            wr<u64>(f, c.len);
        }
        wr<u64>(f, G);
        fclose(f);
    }
    // <prefix>.index.bs.pac : 2-bit forward genome, A0 C1 G2 T3, 4 per byte MSB first
    // (convert_to_2bit, Index.cpp:731-830)
    {
        u64 nb = (G + 3) / 4;
        std::vector<u8> pac(nb, 0);
        for (u64 i = 0; i < G; i++) {
            u8 v = gen[i] == 'A' ? 0 : gen[i] == 'C' ? 1 : gen[i] == 'G' ? 2 : 3;
            pac[i >> 2] |= v << (6 - 2 * (i & 3));
        }
        FILE* f = fopen((base + ".bs.pac").c_str(), "wb");
        wr<u64>(f, nb);
        fwrite(pac.data(), 1, nb, f);
        fclose(f);
    }
    // 3-letter text: complement(fwd) with C->T, then reverse(fwd) with C->T
    // (generate_directional_BS_genome_to_disk, Index.cpp:569-682); codes G=0 T=1 A=2
    // (indenpendent_creadte_index, bwt.cpp:1135-1140).
    const u64 n = 2 * G;
    std::vector<u8> T(n + 32, 0);
    auto code = [](char b) -> u8 { return b == 'G' ? 0 : b == 'T' ? 1 : 2; };
    for (u64 i = 0; i < G; i++) {
        char b = gen[i];
        char cb = b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : 'A';
        if (cb == 'C') cb = 'T';
        T[i] = code(cb);
    }
    for (u64 i = 0; i < G; i++) {
        char b = gen[G - 1 - i];
        if (b == 'C') b = 'T';
        T[G + i] = code(b);
    }
    // ---- suffix array of T (shorter suffix first), rows = n+1 with SA[0] = n  (bwt.cpp:881-905)
    std::vector<u64> SA(n + 1);
    {
        // key = first 20 symbols (digit = code+1, 0 past the end) in 40 bits | index in 24.. no:
        // keep it simple: sort (key32 of 16 symbols, index) pairs, then finish ties by comparison.
        if (n >= (1ULL << 32)) { fprintf(stderr, "orc_index_build: genome too large for the naive oracle builder\n"); return -2; }
        std::vector<u64> ks(n);
        for (u64 i = 0; i < n; i++) {
            u64 k = 0;
            for (int j = 0; j < 16; j++) k = (k << 2) | (i + j < n ? (u64)(T[i + j] + 1) : 0);
            ks[i] = (k << 32) | i;
        }
        std::sort(ks.begin(), ks.end());
        auto cmp_tail = [&](u64 a, u64 b) {           // both share the first 16 symbols
            u64 ia = (a & 0xffffffffULL) + 16, ib = (b & 0xffffffffULL) + 16;
            while (ia < n && ib < n) {
                if (T[ia] != T[ib]) return T[ia] < T[ib];
                ia++; ib++;
            }
            return ia >= n && ib < n;                 // the one that ends first is smaller
        };
        u64 a = 0;
        while (a < n) {
            u64 b = a + 1;
            while (b < n && (ks[b] >> 32) == (ks[a] >> 32)) b++;
            if (b - a > 1) std::sort(ks.begin() + a, ks.begin() + b, cmp_tail);
            a = b;
        }
        SA[0] = n;
        for (u64 i = 0; i < n; i++) SA[i + 1] = ks[i] & 0xffffffffULL;
    }
    const u64 rows = n + 1;

    // ---- BWT bit-planes + interleaved Occ (bwt.cpp:1290-1500) -----------------------------------
    u64 bwt_len = 1 + 2 * (n / 64) + (n / 128) + 2;
    std::vector<u64> bwt(bwt_len + 8, 0);
    std::vector<u64> high_occ;
    high_occ.push_back(0); high_occ.push_back(0);
    u64 cnt[3] = {0, 0, 0};
    u64 shapline = 0, t = 0;
    for (u64 r = 0; r < rows; r++) {
        if (SA[r] == 0) { shapline = r; continue; }
        u8 ch = T[SA[r] - 1];
        u64 w = (t >> 7) * 5 + 1 + 2 * ((t & 127) >> 6);
        u64 sh = 63 - (t & 63);
        bwt[w]     |= (u64)(ch & 1) << sh;
        bwt[w + 1] |= (u64)((ch >> 1) & 1) << sh;
        cnt[ch]++;
        t++;
        if ((t & 65535) == 0) { high_occ.push_back(cnt[1]); high_occ.push_back(cnt[2]); }
        if ((t & 63) == 0) {
            u64 w0 = (t >> 7) * 5, half = (t & 127) >> 6, sb = (t >> 16) * 2;
            u64 cT = cnt[1] - high_occ[sb], cA = cnt[2] - high_occ[sb + 1];
            bwt[w0] |= cT << (48 - 32 * half);
            bwt[w0] |= cA << (32 - 32 * half);
        }
    }
    u64 nacgt[5] = {1, 1 + cnt[0], 1 + cnt[0] + cnt[1], 1 + cnt[0] + cnt[1] + cnt[2],
                    1 + cnt[0] + cnt[1] + cnt[2]};

    // ---- SA_flag (1 counter word + 4 flag words per 256 rows) and SA samples (bwt.cpp:1580-1800)
    std::vector<u64> sa_flag((rows / 256 + 2) * 5 + 8, 0);
    std::vector<u32> sa_samp;
    u64 it = 0, bits = 0, sparse = 0;
    sa_flag[0] = 0; bits = 64; it = 1;
    for (u64 r = 0; r < rows; r++) {
        u64 fl = (SA[r] % 8 == 0) ? 1 : 0;
        if (fl) sparse++;
        sa_flag[it] |= fl << (63 - (bits & 63));
        bits++;
        if ((bits & 63) == 0) it++;
        if (((r + 1) & 255) == 0) { sa_flag[it] = sparse; bits += 64; it++; }
    }
    if (bits & 63) it++;
    it++;
    const u64 sa_flag_len = it;
    for (u64 r = 0; r < rows; r++)
        if (SA[r] % 8 == 0) {
            u32 ch = SA[r] != 0 ? T[SA[r] - 1] : 1;
            sa_samp.push_back((ch << 30) | (u32)(SA[r] / 8));
        }

    // ---- 16-mer table (bwt.cpp:1866-2010): entry[i] = first row of key i (36 bit) with the gap to
    // the previous key's end in the top nibble; keys are base-3 numbers over G=0,T=1,A=2.
    const u64 HS = 43046721ULL + 1;
    std::vector<u32> hh(HS, 0);
    std::vector<u8>  hl(HS, 0);
    {
        std::vector<u32> key16(n, 0xffffffffu);      // key of T[p..p+16) or "short"
        if (n >= 16) {
            u64 k = 0, p3_15 = 14348907ULL;
            for (int j = 0; j < 16; j++) k = k * 3 + T[j];
            key16[0] = (u32)k;
            for (u64 p = 1; p + 16 <= n; p++) {
                k = (k - (u64)T[p - 1] * p3_15) * 3 + T[p + 15];
                key16[p] = (u32)k;
            }
        }
        u64 run = 1;                 // hash_table[0] = 1 (row 0 is the empty suffix)
        hh[0] = 0; hl[0] = 1;
        u64 r = 1, key = 0;
        // rows of full-length suffixes appear in non-decreasing key order; short suffixes and the
        // running pointer produce the gaps
        while (key < HS - 1) {
            // skip short suffixes sitting before the next full-length one
            u64 rr = r;
            while (rr < rows && key16[SA[rr]] == 0xffffffffu) rr++;
            if (rr < rows && key16[SA[rr]] == key) {
                u64 top = rr, bot = rr;
                while (bot < rows && (key16[SA[bot]] == key)) bot++;
                // short suffixes cannot sit inside a key's range (they would share the 16-mer)
                u32 diff = (u32)(top - run) << 28;
                hh[key] = (u32)(top >> 8) | diff;
                hl[key] = (u8)(top & 255);
                hh[key + 1] = (u32)(bot >> 8);
                hl[key + 1] = (u8)(bot & 255);
                run = bot; r = bot;
            } else {
                hh[key] = (u32)(run >> 8);            // top = bot = running value, gap nibble 0
                hl[key] = (u8)(run & 255);
                hh[key + 1] = (u32)(run >> 8);
                hl[key + 1] = (u8)(run & 255);
            }
            key++;
        }
    }

    // ---- write --------------------------------------------------------------------------------
    {
        FILE* f = fopen((base + ".bs.index").c_str(), "wb");
        wr<u64>(f, rows); wr<u64>(f, shapline);
        for (int j = 0; j < 5; j++) wr<u64>(f, nacgt[j]);
        wr<u32>(f, 8); wr<u32>(f, 64); wr<u32>(f, 128);
        fclose(f);
        f = fopen((base + ".bs.index.bwt").c_str(), "wb");
        wr<u64>(f, bwt_len);
        fwrite(bwt.data(), 8, bwt_len, f);
        wr<u64>(f, HS);
        fwrite(hh.data(), 4, HS, f);
        fwrite(hl.data(), 1, HS, f);
        fclose(f);
        f = fopen((base + ".bs.index.sa").c_str(), "wb");
        wr<u64>(f, (u64)sa_samp.size());
        fwrite(sa_samp.data(), 4, sa_samp.size(), f);
        wr<u64>(f, sa_flag_len);
        fwrite(sa_flag.data(), 8, sa_flag_len, f);
        fclose(f);
        f = fopen((base + ".bs.index.occ").c_str(), "wb");
        wr<u64>(f, (u64)high_occ.size());
        fwrite(high_occ.data(), 8, high_occ.size(), f);
        fclose(f);
    }
    return 0;
}

extern "C" int orc_index_build(const char* fasta, const char* prefix)
{
    std::vector<orc_chrom> chroms;
    std::string gen;
    if (!read_fasta(fasta, chroms, gen)) return -1;
    return orc_build_from_genome(chroms, gen, prefix);
}

// ------------------------------------------------------------------------------------------------
template <class T> static bool rd(FILE* f, T& v) { return fread(&v, sizeof(T), 1, f) == 1; }

// Load_Index (Index.cpp:940-1045) + load_index (bwt.cpp:2458-2650)
extern "C" orc_index* orc_index_load(const char* prefix)
{
    orc_index* ix = new orc_index();
    std::string base = std::string(prefix) + ".index";
    FILE* f = fopen(base.c_str(), "rb");
    if (!f) { delete ix; return nullptr; }
    u64 nch = 0; rd(f, nch);
    u64 start = 0;
    for (u64 i = 0; i < nch; i++) {
        u64 len = 0; rd(f, len);
        std::string nm(len, 0);
        if (len) (void)!fread(&nm[0], 1, len, f);
        orc_chrom c; c.name = nm; rd(f, c.len);
        c.start = start; c.end = start + c.len - 1; start = c.end + 1;
        ix->chroms.push_back(c);
    }
    rd(f, ix->G);
    fclose(f);
    f = fopen((base + ".bs.pac").c_str(), "rb");
    if (!f) { delete ix; return nullptr; }
    rd(f, ix->pac_bytes);
    ix->pac.assign(ix->pac_bytes + 1024, 0);
    (void)!fread(ix->pac.data(), 1, ix->pac_bytes, f);
    fclose(f);
    f = fopen((base + ".bs.index").c_str(), "rb");
    if (!f) { delete ix; return nullptr; }
    rd(f, ix->SA_length); rd(f, ix->shapline);
    for (int j = 0; j < 5; j++) rd(f, ix->nacgt[j]);
    rd(f, ix->compress_sa); rd(f, ix->compress_occ); rd(f, ix->high_compress_occ);
    fclose(f);
    f = fopen((base + ".bs.index.bwt").c_str(), "rb");
    if (!f) { delete ix; return nullptr; }
    u64 bl = 0; rd(f, bl);
    ix->bwt.assign(bl + 8, 0);
    (void)!fread(ix->bwt.data(), 8, bl, f);
    ix->bwt_len = bl;
    rd(f, ix->hash_size);
    ix->hash_hi.assign(ix->hash_size, 0);
    ix->hash_lo.assign(ix->hash_size, 0);
    (void)!fread(ix->hash_hi.data(), 4, ix->hash_size, f);
    (void)!fread(ix->hash_lo.data(), 1, ix->hash_size, f);
    fclose(f);
    f = fopen((base + ".bs.index.sa").c_str(), "rb");
    if (!f) { delete ix; return nullptr; }
    u64 ns = 0; rd(f, ns);
    ix->sa.assign(ns, 0);
    (void)!fread(ix->sa.data(), 4, ns, f);
    u64 nf = 0; rd(f, nf);
    ix->sa_flag.assign(nf + 8, 0);
    (void)!fread(ix->sa_flag.data(), 8, nf, f);
    ix->sa_flag_len = nf;
    fclose(f);
    f = fopen((base + ".bs.index.occ").c_str(), "rb");
    if (!f) { delete ix; return nullptr; }
    u64 no = 0; rd(f, no);
    ix->high_occ.assign(no + 2, 0);
    (void)!fread(ix->high_occ.data(), 8, no, f);
    ix->high_occ_len = no;
    fclose(f);
    // total_SA_length = SA[row 0] (Index.cpp:1033-1037)
    ix->total = orc_sa_row(ix, 0);
    return ix;
}

extern "C" void orc_index_free(orc_index* ix) { delete ix; }
extern "C" uint64_t orc_index_genome_len(const orc_index* ix) { return ix->G; }

// ------------------------------------------------------------------------------------------------
// FM primitives

// query_16_mer_hash_table, bwt.h:284-306
void orc_hash_query(const orc_index* ix, u64 key, u64* sp, u64* ep)
{
    u64 a = ((u64)(ix->hash_hi[key] & 0x0fffffffu) << 8) | ix->hash_lo[key];
    u64 b = ((u64)(ix->hash_hi[key + 1] & 0x0fffffffu) << 8) | ix->hash_lo[key + 1];
    b -= ix->hash_hi[key + 1] >> 28;
    *sp = a; *ep = b;
}

// rank of symbol c in the BWT stream [0, line)  (get_occ_value + find_occ_fm_index, bwt.h:1007-1465)
static inline u64 occ_stream(const orc_index* ix, u64 line, int c)
{
    const u64* bwt = ix->bwt.data();
    u64 base = (line >> 7) * 5, half = (line & 127) >> 6, sb = (line >> 16) << 1, r = line & 63;
    u64 w0 = bwt[base];
    u64 cT = ix->high_occ[sb] + ((w0 >> (48 - 32 * half)) & 0xffff);
    u64 cA = ix->high_occ[sb + 1] + ((w0 >> (32 - 32 * half)) & 0xffff);
    if (r) {
        cT += __builtin_popcountll(bwt[base + 1 + 2 * half] >> (64 - r));
        cA += __builtin_popcountll(bwt[base + 2 + 2 * half] >> (64 - r));
    }
    if (c == 1) return cT;
    if (c == 2) return cA;
    return line - cT - cA;
}

// LF step: nacgt[c] + Occ(c, row) with the '$' row removed from the stream (bwt.h:1373-1465)
u64 orc_lf(const orc_index* ix, u64 row, int c)
{
    if (row > ix->shapline) row--;
    return ix->nacgt[c] + occ_stream(ix, row, c);
}

// access_bwt_delta, bwt.h:2413-2447
static inline int bwt_symbol(const orc_index* ix, u64 row)
{
    if (row > ix->shapline) row--;
    u64 w = (row >> 7) * 5 + 1 + 2 * ((row & 127) >> 6), sh = 63 - (row & 63);
    if ((ix->bwt[w] >> sh) & 1) return 1;
    if ((ix->bwt[w + 1] >> sh) & 1) return 2;
    return 0;
}

// bwt_get_sa_restrict_steps_more_than_3, bwt.h:2449-2560: LF-walk to a sampled row.
u64 orc_sa_row_counted(const orc_index* ix, u64 row, u64* n_lf)
{
    u64 l = row, steps = 0;
    if (l == ix->shapline) return 0;
    for (;;) {
        u64 blk = (l >> 8) * 5, last = l & 255;
        u64 w = ix->sa_flag[blk + 1 + (last >> 6)];
        if ((w << (last & 63)) >> 63) {
            u64 rank = ix->sa_flag[blk];
            for (u64 j = 0; j < (last >> 6); j++) rank += __builtin_popcountll(ix->sa_flag[blk + 1 + j]);
            if (last & 63) rank += __builtin_popcountll(w >> (64 - (last & 63)));
            return (u64)(ix->sa[rank] & 0x3fffffffu) * 8 + steps;
        }
        int c = bwt_symbol(ix, l);
        l = orc_lf(ix, l, c);
        steps++;
        if (n_lf) (*n_lf)++;
        if (l == ix->shapline) return steps;
    }
}
u64 orc_sa_row(const orc_index* ix, u64 row) { return orc_sa_row_counted(ix, row, nullptr); }
extern "C" uint64_t orc_sa_at(const orc_index* ix, uint64_t row) { return orc_sa_row(ix, row); }
