// bitmapperbs_amd/csrc/bmbs_search.cpp -- C++ host driver over the C-ABI: the `bitmapperBS --search`
// command line (Process_CommandLines.cpp:88-132), FASTQ in (Process_Reads.cpp:155-317, 810-890), SAM out
// (Process_sam_out.cpp:1137-1153, Schema.cpp:11989-12039, 10537-10640, 11494-11590), mapstats
// (Bitmapper_main.cpp:266-308) -- with the per-read mapping loops of Schema.cpp replaced by batch calls into
// libbmbs_hip.so.  Record order is the input order (== the reference at -t 1).
//
//   bmbs_search --search <index prefix | dir> --seq r.fq[.gz] [-o out.sam] [-e 0.08] [--mapstats f]
//   bmbs_search --search <index> --seq1 a.fq --seq2 b.fq [--min 0] [--max 500] ...
//   extra: --device N, --batch N (reads per GPU batch, default 4 M)
#include "../../include/bmbs.h"
#include <zlib.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace {

struct Reader {                       // .fq / .fastq / .fq.gz / .fastq.gz (gzread handles plain files too)
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, len = 0;
    bool open(const char* path) { f = gzopen(path, "rb"); if (f) { gzbuffer(f, 1 << 20); buf.resize(1 << 22); } return f != nullptr; }
    bool getline(std::string& s)
    {
        s.clear();
        bool any = false;
        for (;;) {
            if (pos == len) { int n = gzread(f, buf.data(), (unsigned)buf.size()); if (n <= 0) return any; len = (size_t)n; pos = 0; }
            any = true;
            char* p = (char*)memchr(buf.data() + pos, '\n', len - pos);
            if (p) { s.append(buf.data() + pos, p - (buf.data() + pos)); pos = p - buf.data() + 1; return true; }
            s.append(buf.data() + pos, len - pos); pos = len;
        }
    }
    void close() { if (f) gzclose(f); f = nullptr; }
};

struct Rec { std::string name, seq, qual; };

bool next_record(Reader& r, Rec& x)
{
    std::string plus;
    if (!r.getline(x.name)) return false;
    r.getline(x.seq); r.getline(plus); r.getline(x.qual);
    for (auto& c : x.seq) c = (char)toupper((unsigned char)c);
    x.qual.resize(x.seq.size(), ' ');
    return true;
}

inline char rc_char(char c) { return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c; }   // rc_table, Process_Reads.cpp:1603
std::string revcomp(const std::string& s) { std::string r(s.rbegin(), s.rend()); for (auto& c : r) c = rc_char(c); return r; }

std::string cigar_text(const bmbs_result& r, const uint32_t* pool, int L)
{
    if (r.n_cigar == 0) return std::to_string(L) + "M";
    std::string s;
    for (int i = 0; i < r.n_cigar; i++) { uint32_t o = pool[r.cigar_off + i]; s += std::to_string(o >> 4); s += "MDISH"[o & 15]; }
    return s;
}

void print_stats(FILE* o, const int64_t st[5])
{
    long long reads = st[0], uniq = st[1], amb = st[2], unm = st[0] - st[1] - st[2];
    fprintf(o, "%-48s%lld\n", "No. of Reads:", reads);
    fprintf(o, "%-48s%lld (%0.2f%%)\n", "No. of Unique Mapped Reads:", uniq, ((double)uniq / (double)reads) * 100);
    fprintf(o, "%-48s%lld (%0.2f%%)\n", "No. of Ambiguous Mapped Reads:", amb, ((double)amb / (double)reads) * 100);
    fprintf(o, "%-48s%lld (%0.2f%%)\n", "No. of Unmapped Reads:", unm, ((double)unm / (double)reads) * 100);
    fprintf(o, "%-47s %0.2f%%\n", "Mismatch and Indel Rate:", ((double)st[4] / (double)st[3]) * 100);
}

bool is_dir(const std::string& p) { struct stat sb; return stat(p.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode); }

}  // namespace

int main(int argc, char** argv)
{
    bmbs_params P; bmbs_default_params(&P);
    std::string index, seq, seq1, seq2, out = "output", mapstats;
    int device = 0;
    long batch = 4000000;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--search") index = val();
        else if (a == "--seq") seq = val();
        else if (a == "--seq1") seq1 = val();
        else if (a == "--seq2") seq2 = val();
        else if (a == "-o") out = val();
        else if (a == "-e") P.e_f = atof(val());
        else if (a == "--min") P.min_ins = atoi(val());
        else if (a == "--max") P.max_ins = atoi(val());
        else if (a == "--mp_max") P.mp_max = atoi(val());
        else if (a == "--mp_min") P.mp_min = atoi(val());
        else if (a == "--np") P.np = atoi(val());
        else if (a == "--gap_open") P.gap_open = atoi(val());
        else if (a == "--gap_extension") P.gap_ext = atoi(val());
        else if (a == "--seed") P.seed_len = atoi(val());
        else if (a == "--phred33") P.q_base = 33;
        else if (a == "--phred64") P.q_base = 64;
        else if (a == "--sensitive") P.sensitive = 1;
        else if (a == "--fast") P.sensitive = 0;
        else if (a == "--pe") {}
        else if (a == "-t") val();                       // CPU threads: not used, the GPU maps
        else if (a == "--mapstats") mapstats = val();
        else if (a == "--device") device = atoi(val());
        else if (a == "--batch") batch = atol(val());
        else { fprintf(stderr, "bmbs_search: unsupported option %s\n", a.c_str()); return 2; }
    }
    if (index.empty() || (seq.empty() && (seq1.empty() || seq2.empty()))) {
        fprintf(stderr, "usage: bmbs_search --search <index> (--seq r.fq | --seq1 a.fq --seq2 b.fq) [-o out.sam] [-e f] [--min n] [--max n] [--mapstats f]\n");
        return 2;
    }
    if (is_dir(index)) index += "/genome";           // Index.cpp:1048-1069
    bmbs_index_file* ixf = bmbs_index_file_load(index.c_str());
    if (!ixf) { fprintf(stderr, "Cannot open index %s.index*\n", index.c_str()); return 1; }
    bmbs_index_view view; bmbs_index_file_view(ixf, &view);
    bmbs_ctx* ctx = bmbs_create(device, &P);
    if (!ctx) { fprintf(stderr, "bmbs_search: no usable HIP device %d (this driver has no CPU mapping path)\n", device); return 1; }
    if (bmbs_index_attach(ctx, &view)) { fprintf(stderr, "%s\n", bmbs_last_error(ctx)); return 1; }
    FILE* o = fopen(out.c_str(), "wb");
    if (!o) { fprintf(stderr, "Cannot open %s\n", out.c_str()); return 1; }
    std::vector<char> obuf(1 << 24);
    setvbuf(o, obuf.data(), _IOFBF, obuf.size());
    // OutPutSAM_Nounheader (Process_sam_out.cpp:1137-1153)
    fprintf(o, "@HD\tVN:1.4\tSO:unsorted\n");
    for (int i = 0; i < view.n_chrom; i++) fprintf(o, "@SQ\tSN:%s\tLN:%llu\n", bmbs_index_file_chrom_name(ixf, i), (unsigned long long)view.chrom_len[i]);
    fprintf(o, "@PG\tID:BitMapperBS\tVN:1.0.2.3\tCL:");
    for (int i = 0; i < argc; i++) fprintf(o, "%s ", argv[i]);
    fprintf(o, "\n");
    const bool pe = seq.empty();
    Reader r1, r2;
    if (!r1.open(pe ? seq1.c_str() : seq.c_str()) || (pe && !r2.open(seq2.c_str()))) { fprintf(stderr, "Cannot open the read file(s)\n"); return 1; }
    std::vector<Rec> a(batch), b(pe ? batch : 0);
    std::vector<char> s1, q1, s2, q2;
    std::vector<bmbs_result> res;
    std::vector<uint32_t> pool;
    std::vector<std::string> lines;
    for (;;) {
        long n = 0;
        while (n < batch && next_record(r1, a[n]) && (!pe || next_record(r2, b[n]))) n++;
        if (n == 0) break;
        // equal-length sub-batches (k = (uint64)(e*L) is per length, Schema.cpp:24546); output keeps the input order
        std::map<std::pair<int, int>, std::vector<long>> groups;
        for (long i = 0; i < n; i++) groups[{(int)a[i].seq.size(), pe ? (int)b[i].seq.size() : 0}].push_back(i);
        lines.assign(n, std::string());
        for (auto& g : groups) {
            const int L = g.first.first;
            const std::vector<long>& ids = g.second;
            const long m = (long)ids.size();
            if (pe && g.first.second != L) { fprintf(stderr, "bmbs_search: mates of different lengths (%d/%d) are not supported by the device path yet; %ld pairs skipped\n", L, g.first.second, m); continue; }
            if (L <= 0) continue;
            const int stride = (L + 15) / 16 * 16;
            s1.assign((size_t)m * stride, 0); q1.assign((size_t)m * stride, 0);
            for (long j = 0; j < m; j++) { memcpy(&s1[(size_t)j * stride], a[ids[j]].seq.data(), L); memcpy(&q1[(size_t)j * stride], a[ids[j]].qual.data(), L); }
            int k = (int)(uint64_t)(P.e_f * L); if (k > 31) k = 31;
            int64_t used = 0;
            if (!pe) {
                res.resize(m); pool.resize((size_t)m * (2 * k + 8));
                if (bmbs_map_se(ctx, s1.data(), q1.data(), L, stride, m, res.data(), pool.data(), (int64_t)pool.size(), &used)) { fprintf(stderr, "%s\n", bmbs_last_error(ctx)); return 1; }
                for (long j = 0; j < m; j++) {
                    const bmbs_result& r = res[j];
                    if (r.status != BMBS_ST_UNIQUE) continue;
                    const Rec& x = a[ids[j]];
                    std::string nm = x.name;                       // cut at the first ' ' or '/' (Process_Reads.cpp:843-850)
                    size_t c = nm.find_first_of(" /"); if (c != std::string::npos) nm.resize(c);
                    const char* nmp = nm.c_str(); if (nmp[0] == '@') nmp++;
                    std::string& ln = lines[ids[j]];
                    char head[512];
                    snprintf(head, sizeof(head), "\t%d\t%s\t%llu\t%d\t", (int)r.flag, bmbs_index_file_chrom_name(ixf, r.chrom), (unsigned long long)r.pos, (int)r.mapq);
                    ln = nmp; ln += head; ln += cigar_text(r, pool.data(), L); ln += "\t*\t0\t0\t";
                    if (r.flag & 16) { ln += revcomp(x.seq); ln += '\t'; ln.append(x.qual.rbegin(), x.qual.rend()); }
                    else { ln += x.seq; ln += '\t'; ln += x.qual; }
                    ln += "\tNM:i:" + std::to_string((int)r.nm) + "\n";
                }
            } else {
                s2.assign((size_t)m * stride, 0); q2.assign((size_t)m * stride, 0);
                for (long j = 0; j < m; j++) { memcpy(&s2[(size_t)j * stride], b[ids[j]].seq.data(), L); memcpy(&q2[(size_t)j * stride], b[ids[j]].qual.data(), L); }
                res.resize(2 * m); pool.resize((size_t)2 * m * (2 * k + 8));
                if (bmbs_map_pe(ctx, s1.data(), q1.data(), s2.data(), q2.data(), L, stride, m, res.data(), pool.data(), (int64_t)pool.size(), &used)) { fprintf(stderr, "%s\n", bmbs_last_error(ctx)); return 1; }
                for (long j = 0; j < m; j++) {
                    const bmbs_result &x1 = res[2 * j], &x2 = res[2 * j + 1];
                    if (x1.status != BMBS_ST_UNIQUE) continue;
                    const Rec &m1 = a[ids[j]], &m2 = b[ids[j]];
                    size_t c = 0;                               // name: first differing char, ' ' or '/' (Process_Reads.cpp:296-307)
                    while (c < m1.name.size() && c < m2.name.size() && m1.name[c] == m2.name[c] && m1.name[c] != ' ' && m1.name[c] != '/') c++;
                    std::string nm = m1.name.substr(0, c);
                    const char* nmp = nm.c_str(); if (nmp[0] == '@') nmp++;
                    const unsigned tlen = x1.reserved;
                    char buf[512];
                    std::string& ln = lines[ids[j]];
                    snprintf(buf, sizeof(buf), "%s\t%d\t%s\t%llu\t%d\t", nmp, (int)x1.flag, bmbs_index_file_chrom_name(ixf, x1.chrom), (unsigned long long)x1.pos, (int)x1.mapq);
                    ln = buf; ln += cigar_text(x1, pool.data(), L);
                    snprintf(buf, sizeof(buf), "\t=\t%llu\t%s%u\t", (unsigned long long)x2.pos, x2.pos < x1.pos ? "-" : "", tlen);
                    ln += buf;
                    if (x1.flag & 32) { ln += m1.seq; ln += '\t'; ln += m1.qual; } else { ln += revcomp(m1.seq); ln += '\t'; ln.append(m1.qual.rbegin(), m1.qual.rend()); }
                    ln += "\tNM:i:" + std::to_string((int)x1.nm) + "\n";
                    snprintf(buf, sizeof(buf), "%s\t%d\t%s\t%llu\t%d\t", nmp, (int)x2.flag, bmbs_index_file_chrom_name(ixf, x2.chrom), (unsigned long long)x2.pos, (int)x2.mapq);
                    ln += buf; ln += cigar_text(x2, pool.data(), L);
                    snprintf(buf, sizeof(buf), "\t=\t%llu\t%s%u\t", (unsigned long long)x1.pos, x1.pos > x2.pos ? "" : "-", tlen);
                    ln += buf;
                    if (x2.flag & 16) { ln += revcomp(m2.seq); ln += '\t'; ln.append(m2.qual.rbegin(), m2.qual.rend()); } else { ln += m2.seq; ln += '\t'; ln += m2.qual; }
                    ln += "\tNM:i:" + std::to_string((int)x2.nm) + "\n";
                }
            }
        }
        for (long i = 0; i < n; i++) if (!lines[i].empty()) fwrite(lines[i].data(), 1, lines[i].size(), o);
        if (n < batch) break;
    }
    fclose(o);
    r1.close(); if (pe) r2.close();
    int64_t st[5];
    bmbs_stats_get(ctx, st);
    print_stats(stderr, st);
    if (!mapstats.empty()) { FILE* m = fopen(mapstats.c_str(), "w"); if (m) { print_stats(m, st); fclose(m); } }
    bmbs_destroy(ctx);
    bmbs_index_file_free(ixf);
    return 0;
}
