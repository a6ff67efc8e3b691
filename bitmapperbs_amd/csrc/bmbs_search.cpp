// bitmapperbs_amd/csrc/bmbs_search.cpp -- C++ host driver over the C-ABI: the `bitmapperBS --search`
// command line (Process_CommandLines.cpp:88-132), FASTQ in (Process_Reads.cpp:155-317, 810-890), SAM out
// (Process_sam_out.cpp:1137-1153, Schema.cpp:11989-12039, 10537-10640, 11494-11590), mapstats
// (Bitmapper_main.cpp:266-308) -- with the per-read mapping loops of Schema.cpp replaced by batch calls into
// libbmbs_hip.so.  Record order is the input order (== the reference at -t 1).
//
//   bmbs_search --search <index prefix | dir> --seq r.fq[.gz] [-o out.sam] [-e 0.08] [--mapstats f]
//   bmbs_search --search <index> --seq1 a.fq --seq2 b.fq [--min 0] [--max 500] [--sensitive] ...
//   output variants (Process_CommandLines.cpp:93-105): --pbat, --unmapped_out, --ambiguous_out, --bam (BGZF-compressed BAM)
//   extra: --device N | --devices a,b,... (one index copy per listed device, batches dealt to whichever context is free, output
//          order kept), --contexts S (contexts per device sharing its index: S batches in flight per GPU so that the copies of
//          one overlap the kernels of another; default 2), --batch N (records per GPU batch, default 500 k), -t N (host I/O
//          threads), --verbose
//
// The reference has ONE reader thread and ONE fprintf sink (Process_Reads.cpp / Schema.cpp:26336-26633), which is
// what limits it.  Here the host side is a three-stage, order-preserving pipeline so that
// the GPU is fed at memory speed:
//   stage R  window of the FASTQ read (parallel pread; .gz: gzread) into a page-locked buffer -> newline index built by the
//            I/O threads -> per-record line starts and lengths.  The records are NOT packed on the host: the text window goes
//            to the GPU as it is and the read rows are cut out of it there (bmbs_map_*_fastq), every read with its own length
//            (k = (uint64)(e*L) is per read, Schema.cpp:24546)
//   stage G  one bmbs_map_se[_var] / bmbs_map_pe[_var] call per batch (the only stage that touches the GPU); one worker thread
//            per context -- the reference's parallel axis (N pthreads over sub-blocks, Schema.cpp:26336-26633) becomes
//            N contexts over batches, and the five counters are summed over them at the end (Schema.cpp:451-476)
//   stage F  SAM text formatted by the I/O threads into per-slice buffers (SEQ / QUAL straight from the FASTQ text)
//   stage W  pwrite of the slices at their prefix offsets (buffered writes to ONE file serialise on its inode lock at the speed of
//            one memcpy -- 10.5 GB/s on the MI355X box -- which is the ceiling of the whole pipeline)
// Batches circulate through hand-over queues, so stage R of batch i+1 and stage W of batch i-1 overlap stage G of i.
#include "../../include/bmbs.h"
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---- a small persistent thread pool: run(n, f) executes f(0..n-1) and returns when all are done -------------
class Pool {
public:
    explicit Pool(int extra_threads)
    {
        for (int i = 0; i < extra_threads; i++) th_.emplace_back([this] { loop(); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return (int)th_.size() + 1; }
    void run(int n, const std::function<void(int)>& f)
    {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int i = 0; i < n; i++) f(i); return; }
        {
            std::lock_guard<std::mutex> l(m_);
            fn_ = &f; ntask_ = n; next_ = 0; pending_ = n; gen_++;
        }
        cv_.notify_all();
        work();                                   // the caller helps
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    void work()
    {
        for (;;) {
            int i;
            const std::function<void(int)>* f;
            {
                std::lock_guard<std::mutex> l(m_);
                if (!fn_ || next_ >= ntask_) return;
                i = next_++; f = fn_;
            }
            (*f)(i);
            {
                std::lock_guard<std::mutex> l(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    void loop()
    {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    bool stop_ = false;
    unsigned long gen_ = 0;
    int pending_ = 0, next_ = 0, ntask_ = 0;
    const std::function<void(int)>* fn_ = nullptr;
};

template <class T> class Chan {                  // hand-over queue between the pipeline stages
public:
    void put(T v) { { std::lock_guard<std::mutex> l(m_); q_.push(v); } cv_.notify_one(); }
    T get() { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [this] { return !q_.empty(); }); T v = q_.front(); q_.pop(); return v; }
private:
    std::mutex m_; std::condition_variable cv_; std::queue<T> q_;
};

template <class T> class OrderedChan {           // hands items out in sequence-number order whatever order they arrive in
public:
    void put(long seq, T v) { { std::lock_guard<std::mutex> l(m_); q_[seq] = v; } cv_.notify_all(); }
    T get()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return q_.count(next_) != 0; });
        T v = q_[next_]; q_.erase(next_); next_++;
        return v;
    }
private:
    std::mutex m_; std::condition_variable cv_; std::map<long, T> q_; long next_ = 0;
};

// ---- FASTQ text source: plain files are read with parallel pread()s straight into the batch's page-locked window, .gz goes
// through gzread ---------------------------------------------------------------------------------------------------------------
struct Source {
    bool gz = false;
    int fd = -1;
    size_t size = 0, off = 0;
    gzFile gzf = nullptr;
    std::vector<char> carry;
    bool gz_eof = false;

    bool open(const char* path)
    {
        FILE* f = fopen(path, "rb");
        if (!f) return false;
        unsigned char mg[2] = {0, 0};
        const size_t got = fread(mg, 1, 2, f);
        fclose(f);
        gz = got == 2 && mg[0] == 0x1f && mg[1] == 0x8b;
        if (gz) { gzf = gzopen(path, "rb"); if (gzf) gzbuffer(gzf, 1 << 20); return gzf != nullptr; }
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        if (fstat(fd, &sb)) return false;
        size = (size_t)sb.st_size;
        (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
        return true;
    }
    // up to `cap` bytes of text starting at the current record boundary into dst; `last` when they reach the end of the input
    size_t window(Pool& pool, char* dst, size_t cap, bool& last)
    {
        if (!gz) {
            const size_t len = std::min(cap, size - off);
            const int T = pool.size() * 2;
            const size_t per = ((len + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
            std::atomic<bool> bad(false);
            pool.run(T, [&](int t) {
                size_t a = std::min(len, per * (size_t)t);
                const size_t e = std::min(len, a + per);
                while (a < e) {
                    const ssize_t g = pread(fd, dst + a, e - a, (off_t)(off + a));
                    if (g <= 0) { bad = true; return; }
                    a += (size_t)g;
                }
            });
            last = off + len == size;
            return bad ? 0 : len;
        }
        size_t have = carry.size();
        if (have) memcpy(dst, carry.data(), have);
        carry.clear();
        while (!gz_eof && have < cap) {
            const int n = gzread(gzf, dst + have, (unsigned)std::min<size_t>(cap - have, 1u << 30));
            if (n <= 0) { gz_eof = true; break; }
            have += (size_t)n;
        }
        last = gz_eof;
        return have;
    }
    void consumed(const char* p, size_t len, size_t used)
    {
        if (!gz) { off += used; return; }
        carry.assign(p + used, p + len);
    }
    void close() { if (gzf) gzclose(gzf); if (fd >= 0) ::close(fd); }
};

struct Lines {                                   // newline index of one window
    const char* p = nullptr;
    size_t len = 0, used = 0;                    // used = bytes consumed by the batch's records
    std::vector<size_t> nl;                      // position of the '\n' ending line i (or len for an unterminated last line)
    size_t start(size_t line) const { return line == 0 ? 0 : nl[line - 1] + 1; }
    size_t end(size_t line) const { return nl[line]; }
};

// `part` is caller-owned scratch that keeps its capacity from batch to batch (fresh allocations of this size
// are mmap'ed and page-faulted in again on every call)
void index_lines(Pool& pool, const char* p, size_t len, bool last, Lines& out, std::vector<std::vector<size_t>>& part)
{
    out.p = p; out.len = len; out.used = 0;
    const int T = pool.size() * 4;
    part.resize((size_t)T);
    const size_t per = (len + (size_t)T - 1) / (size_t)T;
    pool.run(T, [&](int t) {
        const size_t a = std::min(len, per * (size_t)t), b = std::min(len, a + per);
        std::vector<size_t>& v = part[(size_t)t];
        v.clear();
        v.reserve((b - a) / 64 + 16);
        const char* q = p + a;
        const char* e = p + b;
        while (q < e) {
            const char* h = (const char*)memchr(q, '\n', (size_t)(e - q));
            if (!h) break;
            v.push_back((size_t)(h - p));
            q = h + 1;
        }
    });
    std::vector<size_t> base((size_t)T + 1, 0);
    for (int t = 0; t < T; t++) base[(size_t)t + 1] = base[(size_t)t] + part[(size_t)t].size();
    const size_t total = base[(size_t)T];
    const bool open_tail = last && len > 0 && p[len - 1] != '\n';
    out.nl.resize(total + (open_tail ? 1 : 0));
    pool.run(T, [&](int t) {
        if (!part[(size_t)t].empty()) memcpy(&out.nl[base[(size_t)t]], part[(size_t)t].data(), part[(size_t)t].size() * sizeof(size_t));
    });
    if (open_tail) out.nl[total] = len;
}

struct Pinned {                                  // page-locked staging (bmbs_host_alloc)
    char* p = nullptr; size_t cap = 0;
    bool need(size_t bytes)
    {
        if (bytes <= cap) return true;
        if (p) bmbs_host_free(p);
        cap = bytes + bytes / 4 + 4096;
        p = (char*)bmbs_host_alloc(cap);
        if (!p) { cap = 0; return false; }
        return true;
    }
    void release() { if (p) bmbs_host_free(p); p = nullptr; cap = 0; }
};

// line index of one FASTQ file's records in a batch: offsets into the text window + lengths (what bmbs_fastq_view wants)
struct RecIdx {
    Pinned mem;
    uint32_t *seq_off = nullptr, *qual_off = nullptr;
    uint16_t *seq_len = nullptr, *qual_len = nullptr;
    bool need(size_t n)
    {
        if (!mem.need(n * 12 + 64)) return false;
        seq_off = (uint32_t*)mem.p; qual_off = seq_off + n; seq_len = (uint16_t*)(qual_off + n); qual_len = seq_len + n;
        return true;
    }
};

struct Batch {
    long seq = 0;                                // position in the input (output order)
    long n = 0;                                  // records
    bool end = false;                            // no more input after this batch
    Pinned text1, text2;                         // the FASTQ text windows, as read from the files
    Lines l1, l2;
    RecIdx r1, r2;
    int maxL = 0, k = 0, max_ops = 8;            // longest read of the batch, its threshold, CIGAR pool slots per read (bmbs_max_cigar_ops)
    bool uniform = true;                         // every read of the batch has the same length
    Pinned res, pool;
    std::vector<std::vector<char>> text;         // SAM text per formatter slice (capacity kept from batch to batch)
    std::vector<size_t> text_len;
    std::vector<std::vector<char>> bam_rec, bam_z;   // --bam: records and BGZF blocks of a slice
};

inline char rc_char(char c) { return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c; }   // rc_table, Process_Reads.cpp:1603

// eight characters at a time: reverse complement (A<->T = ^0x15, C<->G = ^0x04, everything else unchanged) and upper-casing
inline uint64_t zero_bytes(uint64_t x) { const uint64_t K7F = 0x7f7f7f7f7f7f7f7full; return ~(((x & K7F) + K7F) | x | K7F); }   // 0x80 where a byte is 0
inline uint64_t comp8(uint64_t w)
{
    const uint64_t at = zero_bytes(w ^ 0x4141414141414141ull) | zero_bytes(w ^ 0x5454545454545454ull);
    const uint64_t cg = zero_bytes(w ^ 0x4343434343434343ull) | zero_bytes(w ^ 0x4747474747474747ull);
    return w ^ ((at >> 7) * 0x15) ^ ((cg >> 7) * 0x04);
}
// dst[i] = complement(src[L-1-i])
inline void revcomp_copy(char* dst, const char* src, int L)
{
    int i = 0;
    for (; i + 8 <= L; i += 8) { uint64_t w; memcpy(&w, src + L - 8 - i, 8); w = comp8(__builtin_bswap64(w)); memcpy(dst + i, &w, 8); }
    for (; i < L; i++) dst[i] = rc_char(src[L - 1 - i]);
}
inline void reverse_copy(char* dst, const char* src, int L)
{
    int i = 0;
    for (; i + 8 <= L; i += 8) { uint64_t w; memcpy(&w, src + L - 8 - i, 8); w = __builtin_bswap64(w); memcpy(dst + i, &w, 8); }
    for (; i < L; i++) dst[i] = src[L - 1 - i];
}
// toupper over a buffer: a character with bit 0x20 set is rare in FASTQ sequence lines, so test 8 at a time
inline void upper_inplace(char* d, int L)
{
    int i = 0;
    for (; i + 8 <= L; i += 8) {
        uint64_t w; memcpy(&w, d + i, 8);
        if (w & 0x2020202020202020ull) for (int j = 0; j < 8; j++) { const char c = d[i + j]; if (c >= 'a' && c <= 'z') d[i + j] = (char)(c - 32); }
    }
    for (; i < L; i++) { const char c = d[i]; if (c >= 'a' && c <= 'z') d[i] = (char)(c - 32); }
}

inline void put_uint(std::string& s, unsigned long long v)
{
    char b[24]; int i = 24;
    do { b[--i] = (char)('0' + v % 10); v /= 10; } while (v);
    s.append(b + i, (size_t)(24 - i));
}

// raw cursor into a pre-sized slice buffer (the formatter computes an upper bound first)
struct Out {
    char* p;
    void ch(char c) { *p++ = c; }
    void mem(const char* s, size_t n) { memcpy(p, s, n); p += n; }
    void str(const std::string& s) { mem(s.data(), s.size()); }
    template <size_t N> void lit(const char (&s)[N]) { memcpy(p, s, N - 1); p += N - 1; }
    void num(unsigned long long v)
    {
        char b[24]; int i = 24;
        do { b[--i] = (char)('0' + v % 10); v /= 10; } while (v);
        mem(b + i, (size_t)(24 - i));
    }
    void cigar(const bmbs_result& r, const uint32_t* pool, int L)
    {
        if (r.n_cigar == 0) { num((unsigned)L); ch('M'); return; }
        if (r.n_cigar == 255) {               // "more operations than pool slots": bmbs_max_cigar_ops rules it out; never print garbage
            fprintf(stderr, "bmbs_search: internal error: an alignment overflowed its CIGAR slots\n");
            std::_Exit(3);
        }
        for (int i = 0; i < r.n_cigar; i++) { const uint32_t o = pool[r.cigar_off + i]; num(o >> 4); ch("MDISH"[o & 15]); }
    }
    // SEQ \t QUAL from the FASTQ text: bases upper-cased (Process_Reads.cpp:836), a quality line shorter than the sequence
    // padded with ' ' (qual.resize); as read, or reverse-complemented with reversed qualities
    void seq(const char* sq, const char* ql, int L, int qlen, bool rc)
    {
        char* qd = p + L + 1;
        if (!rc) {
            memcpy(p, sq, (size_t)L); upper_inplace(p, L);
            memcpy(qd, ql, (size_t)qlen);
            for (int i = qlen; i < L; i++) qd[i] = ' ';
        } else {
            memcpy(qd, sq, (size_t)L); upper_inplace(qd, L);            // the quality area as scratch
            revcomp_copy(p, qd, L);
            if (qlen == L) reverse_copy(qd, ql, L);
            else for (int i = 0; i < L; i++) { const int jj = L - 1 - i; qd[i] = jj < qlen ? ql[jj] : ' '; }
        }
        p[L] = '\t';
        p += 2 * (size_t)L + 1;
    }
};

// ---- --bam (Process_CommandLines.cpp:94; the reference hands each SAM line to htslib's sam_parse1 and bam_write1,
// bam_prase.cpp:201-221): the same conversion here, line by line, then BGZF blocks compressed by the I/O threads ----------
int reg2bin(long long beg, long long end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
inline void put_le32(std::vector<char>& o, uint32_t v) { char b[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)}; o.insert(o.end(), b, b + 4); }
inline void put_le16(std::vector<char>& o, uint16_t v) { o.push_back((char)v); o.push_back((char)(v >> 8)); }

struct BamNames { std::vector<std::string> names; int id(const char* s, size_t n) const { for (size_t i = 0; i < names.size(); i++) if (names[i].size() == n && !memcmp(names[i].data(), s, n)) return (int)i; return -1; } };

// one SAM line of ours (11 mandatory columns + optional NM:i) -> one BAM record appended to o
void sam_line_to_bam(const char* l, size_t n, const BamNames& refs, std::vector<char>& o)
{
    const char* f[13]; size_t fl[13]; int nf = 0;
    const char* s = l; const char* e = l + n;
    while (nf < 13) {
        const char* t = (const char*)memchr(s, '\t', (size_t)(e - s));
        f[nf] = s; fl[nf] = t ? (size_t)(t - s) : (size_t)(e - s); nf++;
        if (!t) break;
        s = t + 1;
    }
    auto num = [&](int i) -> long long { long long v = 0; bool neg = false; size_t j = 0; if (fl[i] && f[i][0] == '-') { neg = true; j = 1; } for (; j < fl[i]; j++) v = v * 10 + (f[i][j] - '0'); return neg ? -v : v; };
    const int flag = (int)num(1);
    const int refid = (fl[2] == 1 && f[2][0] == '*') ? -1 : refs.id(f[2], fl[2]);
    const long long pos = num(3) - 1;
    const int mapq = (int)num(4);
    // CIGAR
    std::vector<uint32_t> cig;
    long long reflen = 0;
    if (!(fl[5] == 1 && f[5][0] == '*')) {
        uint32_t len = 0;
        for (size_t j = 0; j < fl[5]; j++) {
            const char c = f[5][j];
            if (c >= '0' && c <= '9') { len = len * 10 + (uint32_t)(c - '0'); continue; }
            const int op = c == 'M' ? 0 : c == 'I' ? 1 : c == 'D' ? 2 : c == 'N' ? 3 : c == 'S' ? 4 : c == 'H' ? 5 : c == 'P' ? 6 : c == '=' ? 7 : 8;
            cig.push_back((len << 4) | (uint32_t)op);
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += len;
            len = 0;
        }
    }
    const int next_ref = (fl[6] == 1 && f[6][0] == '=') ? refid : (fl[6] == 1 && f[6][0] == '*') ? -1 : refs.id(f[6], fl[6]);
    const long long next_pos = num(7) - 1;
    const long long tlen = num(8);
    const size_t lseq = (fl[9] == 1 && f[9][0] == '*') ? 0 : fl[9];
    const size_t at = o.size();
    put_le32(o, 0);                                                  // block_size, patched below
    put_le32(o, (uint32_t)refid); put_le32(o, (uint32_t)pos);
    o.push_back((char)(fl[0] + 1)); o.push_back((char)mapq);
    put_le16(o, (uint16_t)reg2bin(pos, pos + (cig.empty() || reflen == 0 ? 1 : reflen)));
    put_le16(o, (uint16_t)cig.size()); put_le16(o, (uint16_t)flag);
    put_le32(o, (uint32_t)lseq); put_le32(o, (uint32_t)next_ref); put_le32(o, (uint32_t)next_pos); put_le32(o, (uint32_t)tlen);
    o.insert(o.end(), f[0], f[0] + fl[0]); o.push_back(0);
    for (uint32_t c : cig) put_le32(o, c);
    static const unsigned char nt16[256] = {
        15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
        15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 1,2,4,8,15,15,15,15,15,15,15,15,15,0,15,15,
        15,1,14,2,13,15,15,4,11,15,15,12,15,3,15,15, 15,15,5,6,8,15,7,9,15,10,15,15,15,15,15,15,
        15,1,14,2,13,15,15,4,11,15,15,12,15,3,15,15, 15,15,5,6,8,15,7,9,15,10,15,15,15,15,15,15,
        15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
        15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
        15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
        15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15};
    for (size_t j = 0; j < lseq; j += 2) {
        const unsigned char hi = nt16[(unsigned char)f[9][j]], lo = j + 1 < lseq ? nt16[(unsigned char)f[9][j + 1]] : 0;
        o.push_back((char)((hi << 4) | lo));
    }
    if (fl[10] == 1 && f[10][0] == '*') o.insert(o.end(), lseq, (char)0xff);
    else for (size_t j = 0; j < lseq; j++) o.push_back((char)(f[10][j] - 33));
    if (nf > 11 && fl[11] > 5 && !memcmp(f[11], "NM:i:", 5)) {
        long long v = 0;
        for (size_t j = 5; j < fl[11]; j++) v = v * 10 + (f[11][j] - '0');
        o.push_back('N'); o.push_back('M');
        if (v <= 0xff) { o.push_back('C'); o.push_back((char)v); }
        else if (v <= 0xffff) { o.push_back('S'); put_le16(o, (uint16_t)v); }
        else { o.push_back('I'); put_le32(o, (uint32_t)v); }
    }
    const uint32_t bs = (uint32_t)(o.size() - at - 4);
    o[at] = (char)bs; o[at + 1] = (char)(bs >> 8); o[at + 2] = (char)(bs >> 16); o[at + 3] = (char)(bs >> 24);
}

// BGZF: independent gzip members of at most 0xff00 input bytes with the BC extra field (SAM spec 4.1)
void bgzf_append(const char* in, size_t n, std::vector<char>& out)
{
    size_t done = 0;
    do {
        const size_t chunk = std::min<size_t>(n - done, 0xff00);
        const size_t at = out.size();
        out.resize(at + 18 + compressBound((uLong)chunk) + 8);
        z_stream zs; memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef*)(in + done); zs.avail_in = (uInt)chunk;
        zs.next_out = (Bytef*)(out.data() + at + 18); zs.avail_out = (uInt)(out.size() - at - 18 - 8);
        deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(out.data() + at, hdr, 16);
        const uint16_t bsize = (uint16_t)(clen + 25);
        out[at + 16] = (char)bsize; out[at + 17] = (char)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)(in + done), (uInt)chunk);
        char* t = out.data() + at + 18 + clen;
        t[0] = (char)crc; t[1] = (char)(crc >> 8); t[2] = (char)(crc >> 16); t[3] = (char)(crc >> 24);
        t[4] = (char)chunk; t[5] = (char)(chunk >> 8); t[6] = (char)(chunk >> 16); t[7] = (char)(chunk >> 24);
        out.resize(at + 18 + clen + 8);
        done += chunk;
    } while (done < n);
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

double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

}  // namespace

int main(int argc, char** argv)
{
    bmbs_params P; bmbs_default_params(&P);
    std::string index, seq, seq1, seq2, out = "output", mapstats, build_fasta, index_folder;
    int device = 0, io_threads = 0, contexts = 2;
    std::vector<int> devices;
    long batch = 500000;                         // 20 M SE reads to a file: 27.3 M reads/s with 250-500 k, 23.3-24.2 with 1 M (pipeline fill)
    bool verbose = false, unmapped_out = false, pbat = false, bam = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--search") index = val();
        else if (a == "--index") build_fasta = val();                 // Process_CommandLines.cpp:107, 364-381
        else if (a == "--index_folder") index_folder = val();
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
        else if (a == "-t") io_threads = atoi(val());    // the reference's mapping threads; here: host I/O threads (the GPU maps)
        else if (a == "--mapstats") mapstats = val();
        else if (a == "--device") device = atoi(val());
        else if (a == "--devices") {                                  // comma-separated device ids, one index copy each
            const char* v = val();
            devices.clear();
            for (const char* q = v; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
        }
        else if (a == "--contexts") contexts = atoi(val());
        else if (a == "--batch") batch = atol(val());
        else if (a == "--verbose") verbose = true;
        else if (a == "--unmapped_out") unmapped_out = true;          // Process_CommandLines.cpp:104-105
        else if (a == "--ambiguous_out") P.ambiguous_out = 1;
        else if (a == "--pbat") pbat = true;                          // Process_CommandLines.cpp:93
        else if (a == "--bam") bam = true;                            // Process_CommandLines.cpp:94-95
        else if (a == "--sam") bam = false;
        else { fprintf(stderr, "bmbs_search: unsupported option %s\n", a.c_str()); return 2; }
    }
    if (!build_fasta.empty()) {
        // bitmapperBS --index <fasta> [--index_folder <dir>]: <fasta>.index* or <dir>/genome.index* (Index.cpp:832-938)
        std::string prefix = build_fasta;
        if (!index_folder.empty()) {
            while (index_folder.size() > 1 && index_folder.back() == '/') index_folder.pop_back();
            ::mkdir(index_folder.c_str(), 0755);
            prefix = index_folder + "/genome";
        }
        if (io_threads <= 0) io_threads = (int)std::thread::hardware_concurrency();
        const double t0 = now();
        const int rc = bmbs_index_build(build_fasta.c_str(), prefix.c_str(), io_threads < 1 ? 1 : io_threads);
        if (rc) { fprintf(stderr, "bmbs_search: index build failed (%d)\n", rc); return 1; }
        fprintf(stderr, "index written to %s.index* in %.1f s\n", prefix.c_str(), now() - t0);
        return 0;
    }
    if (index.empty() || (seq.empty() && (seq1.empty() || seq2.empty()))) {
        fprintf(stderr, "usage: bmbs_search --index <genome.fa> [--index_folder dir] [-t threads]\n       bmbs_search --search <index> (--seq r.fq | --seq1 a.fq --seq2 b.fq) [-o out.sam] [-e f] [--min n] [--max n] [--sensitive] [--pbat] [--unmapped_out] [--ambiguous_out] [--bam] [--mapstats f] [-t io_threads]\n");
        return 2;
    }
    if (batch < 1) batch = 1;
    if (io_threads <= 0) { io_threads = (int)std::thread::hardware_concurrency(); if (io_threads > 32) io_threads = 32; }
    if (io_threads < 1) io_threads = 1;
    const double t_start = now();
    if (devices.empty()) devices.push_back(device);
    if (contexts < 1) contexts = 1;
    const bool pe = seq.empty();
    const int n_batches = 4 + (int)devices.size() * contexts;
    std::vector<Batch> batches((size_t)n_batches);
    // page-locking memory costs ~0.2 ms per MB: the staging of every circulating batch is allocated once, here, by a few threads
    // at a time, sized from the first records of the (plain-text) input -- a batch that needs more grows its own, which for 250-base
    // reads and the former fixed 400 bytes per record re-allocated every window once: 1.2 s of a 1.6 s run
    size_t est0 = 400;
    {
        const std::string& f0 = pe ? seq1 : seq;
        char head[1 << 16];
        FILE* fp = fopen(f0.c_str(), "rb");
        const size_t got = fp ? fread(head, 1, sizeof head, fp) : 0;
        if (fp) fclose(fp);
        if (got > 2 && !((unsigned char)head[0] == 0x1f && (unsigned char)head[1] == 0x8b)) {
            size_t lines = 0, last = 0;
            for (size_t i = 0; i < got; i++) if (head[i] == '\n') { lines++; if (lines % 4 == 0) last = i + 1; }
            if (lines >= 4) est0 = std::max<size_t>(est0, last / (lines / 4) + 32);
        }
    }
    std::thread prealloc([&] {
        const size_t want = std::min<size_t>((size_t)batch * est0 + (1u << 16), (size_t)4000 << 20) + 64;
        const int ops250 = std::max(8, (int)bmbs_max_cigar_ops(&P, 250));
        std::vector<std::thread> th;
        for (auto& b : batches)
            th.emplace_back([&, want, ops250] {
                Batch* bb = &b;
                bb->text1.need(want); if (pe) bb->text2.need(want);
                bb->r1.need((size_t)batch); if (pe) bb->r2.need((size_t)batch);
                bb->res.need((size_t)batch * sizeof(bmbs_result) * (pe ? 2 : 1) + 64);
                bb->pool.need((size_t)batch * (size_t)ops250 * (pe ? 2 : 1) * 4 + 64);
                bb->l1.nl.reserve((size_t)batch * 4 + 16); if (pe) bb->l2.nl.reserve((size_t)batch * 4 + 16);
            });
        for (auto& t : th) t.join();
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } prealloc_guard{prealloc};     // error returns below must not leave it running
    if (is_dir(index)) index += "/genome";           // Index.cpp:1048-1069
    bmbs_index_file* ixf = bmbs_index_file_load(index.c_str());
    if (!ixf) { fprintf(stderr, "Cannot open index %s.index*\n", index.c_str()); return 1; }
    bmbs_index_view view; bmbs_index_file_view(ixf, &view);
    // one owner context per listed device (attached in parallel: each uploads and re-packs its own index copy), plus
    // contexts-1 further contexts per device on the owner's index (bmbs_index_share)
    std::vector<bmbs_ctx*> ctxs;                  // owners first
    for (int d : devices) {
        bmbs_ctx* c = bmbs_create(d, &P);
        if (!c) { fprintf(stderr, "bmbs_search: no usable HIP device %d (this driver has no CPU mapping path)\n", d); return 1; }
        ctxs.push_back(c);
    }
    {
        std::vector<int> rcs(ctxs.size(), 0);
        std::vector<std::thread> th;
        for (size_t i = 0; i < ctxs.size(); i++) th.emplace_back([&, i] { rcs[i] = bmbs_index_attach(ctxs[i], &view); });
        for (auto& t : th) t.join();
        for (size_t i = 0; i < ctxs.size(); i++) if (rcs[i]) { fprintf(stderr, "%s\n", bmbs_last_error(ctxs[i])); return 1; }
    }
    const size_t n_owner = ctxs.size();
    for (size_t i = 0; i < n_owner; i++)
        for (int s = 1; s < contexts; s++) {
            bmbs_ctx* c = bmbs_create(devices[i], &P);
            if (!c || bmbs_index_share(c, ctxs[i])) { fprintf(stderr, "bmbs_search: cannot create a shared context on device %d\n", devices[i]); return 1; }
            ctxs.push_back(c);
        }
    const double t_loaded = now();
    std::vector<std::string> chrom_names;
    for (int i = 0; i < view.n_chrom; i++) chrom_names.push_back(bmbs_index_file_chrom_name(ixf, i));
    const int ofd = ::open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (ofd < 0) { fprintf(stderr, "Cannot open %s\n", out.c_str()); return 1; }
    size_t out_off = 0;
    {
        // OutPutSAM_Nounheader (Process_sam_out.cpp:1137-1153)
        std::string h = "@HD\tVN:1.4\tSO:unsorted\n";
        for (int i = 0; i < view.n_chrom; i++) { h += "@SQ\tSN:" + chrom_names[(size_t)i] + "\tLN:"; put_uint(h, view.chrom_len[i]); h += '\n'; }
        h += "@PG\tID:BitMapperBS\tVN:1.0.2.3\tCL:";
        for (int i = 0; i < argc; i++) { h += argv[i]; h += ' '; }
        h += '\n';
        if (bam) {
            // BAM header: magic, the same text, the reference dictionary; one BGZF block series
            std::vector<char> hb = {'B', 'A', 'M', 1}, z;
            put_le32(hb, (uint32_t)h.size()); hb.insert(hb.end(), h.begin(), h.end());
            put_le32(hb, (uint32_t)view.n_chrom);
            for (int i = 0; i < view.n_chrom; i++) {
                const std::string& nm = chrom_names[(size_t)i];
                put_le32(hb, (uint32_t)nm.size() + 1); hb.insert(hb.end(), nm.begin(), nm.end()); hb.push_back(0);
                put_le32(hb, (uint32_t)view.chrom_len[i]);
            }
            bgzf_append(hb.data(), hb.size(), z);
            h.assign(z.begin(), z.end());
        }
        if (pwrite(ofd, h.data(), h.size(), 0) != (ssize_t)h.size()) { fprintf(stderr, "write error on %s\n", out.c_str()); return 1; }
        out_off = h.size();
    }
    BamNames bam_refs; bam_refs.names = chrom_names;
    prealloc.join();
    const bool ambiguous_out = P.ambiguous_out != 0;
    // --pbat: single-end reads are mapped as their reverse complement with mirrored qualities (inputReads_single_directly_pbat,
    // Process_Reads.cpp:986-1075; Schema.cpp:15102 need_reverse_quality = 1); paired-end input files swap roles
    // (exchange_two_reads, Process_Reads.cpp:1628, called from Bitmapper_main.cpp:169)
    if (pbat && pe) std::swap(seq1, seq2);
    const bool pbat_se = pbat && !pe;
    Source src1, src2;
    if (!src1.open(pe ? seq1.c_str() : seq.c_str()) || (pe && !src2.open(seq2.c_str()))) { fprintf(stderr, "Cannot open the read file(s)\n"); return 1; }

    Chan<Batch*> free_q, gpu_q, wr_q;
    OrderedChan<Batch*> out_q;                    // the G workers finish in any order; formatting and writing follow the input order
    long next_seq = 0;
    for (auto& b : batches) free_q.put(&b);
    std::atomic<bool> failed(false);
    double t_read = 0, t_gpu = 0, t_write = 0, t_index = 0, t_format = 0, t_window = 0, t_wait_r = 0, t_wait_f = 0, t_wait_w = 0, t_wait_g = 0;
    struct Ev { char stage; long n; double a, b; };
    std::vector<Ev> ev_r, ev_g, ev_f, ev_w;

    // ---------------- stage R: text window -> line index -> per-record line starts and lengths ---------------------
    std::thread reader([&] {
        Pool pool(std::max(1, io_threads / 4) - 1);
        size_t est = est0;                                              // bytes per record, refined from the data
        std::vector<std::vector<size_t>> part;                          // scratch reused by every batch
        for (;;) {
            const double tw0 = now();
            Batch* b = free_q.get();
            const double t0 = now();
            t_wait_r += t0 - tw0;
            b->seq = next_seq++;
            b->n = 0; b->end = false; b->maxL = 0; b->k = 0; b->uniform = true;
            auto bail = [&](const char* why) {
                if (why) fprintf(stderr, "bmbs_search: %s\n", why);
                failed = true; b->end = true; b->n = 0; gpu_q.put(b);
            };
            // a window is handed to the library with 32-bit offsets: keep it below 4 GiB
            const size_t want = std::min<size_t>((size_t)batch * est + (1u << 16), (size_t)4000 << 20);
            if (!b->text1.need(want + 64) || (pe && !b->text2.need(want + 64))) { bail("cannot allocate page-locked staging memory"); return; }
            bool last1 = true, last2 = true;
            const size_t n1 = src1.window(pool, b->text1.p, want, last1);
            t_window += now() - t0;
            index_lines(pool, b->text1.p, n1, last1, b->l1, part);
            t_index += now() - t0;
            long avail1 = (long)(b->l1.nl.size() / 4), avail2 = 0;
            long nrec = avail1;
            size_t n2 = 0;
            if (pe) {
                n2 = src2.window(pool, b->text2.p, want, last2);
                index_lines(pool, b->text2.p, n2, last2, b->l2, part);
                avail2 = (long)(b->l2.nl.size() / 4);
                nrec = std::min(nrec, avail2);
            }
            if (nrec > batch) nrec = batch;
            // the input ends with this batch when a file has no complete record left after it (PE: the shorter file decides)
            b->end = (last1 && avail1 == nrec) || (pe && last2 && avail2 == nrec);
            if (nrec == 0) {
                if (!b->end) { fprintf(stderr, "bmbs_search: FASTQ record larger than the %zu-byte window\n", want); failed = true; b->end = true; }
                gpu_q.put(b);
                return;
            }
            const size_t used1 = std::min(n1, b->l1.nl[(size_t)nrec * 4 - 1] + 1);
            src1.consumed(b->text1.p, n1, used1);
            b->l1.used = used1;
            if (pe) { b->l2.used = std::min(n2, b->l2.nl[(size_t)nrec * 4 - 1] + 1); src2.consumed(b->text2.p, n2, b->l2.used); }
            est = std::max<size_t>(64, used1 / (size_t)nrec + 16);
            b->n = nrec;
            if (!b->r1.need((size_t)nrec) || (pe && !b->r2.need((size_t)nrec))) { bail("cannot allocate page-locked staging memory"); return; }
            // ---- one batch = one library call: every record keeps its own length, rows in input order
            const int T = pool.size();
            const long per = (nrec + T - 1) / T;
            std::vector<int> mx((size_t)T, 0), mn((size_t)T, 1 << 30);
            std::vector<long> bad((size_t)T, -1);
            pool.run(T, [&](int t) {
                const long a = std::min<long>(nrec, per * t), e = std::min<long>(nrec, a + per);
                int m = 0, lo = 1 << 30;
                auto one = [&](const Lines& ln, RecIdx& ri, long r) -> bool {
                    const size_t s0 = ln.start((size_t)r * 4 + 1), e0 = ln.end((size_t)r * 4 + 1);
                    const size_t q0 = ln.start((size_t)r * 4 + 3), q1 = ln.end((size_t)r * 4 + 3);
                    const size_t L = e0 - s0;
                    if (L < 1 || L > 1000) return false;
                    ri.seq_off[r] = (uint32_t)s0; ri.qual_off[r] = (uint32_t)q0;
                    ri.seq_len[r] = (uint16_t)L; ri.qual_len[r] = (uint16_t)std::min(L, q1 - q0);
                    m = std::max(m, (int)L); lo = std::min(lo, (int)L);
                    return true;
                };
                for (long r = a; r < e; r++)
                    if (!one(b->l1, b->r1, r) || (pe && !one(b->l2, b->r2, r))) { if (bad[(size_t)t] < 0) bad[(size_t)t] = r; break; }
                mx[(size_t)t] = m; mn[(size_t)t] = lo;
            });
            int maxL = 0, minL = 1 << 30;
            for (int t = 0; t < T; t++) {
                maxL = std::max(maxL, mx[(size_t)t]); minL = std::min(minL, mn[(size_t)t]);
                if (bad[(size_t)t] >= 0) {
                    // the reference's reader has no such limit on paper, its kernels do (SEQ_MAX_LENGTH 1000, Schema.h); a silently
                    // dropped record would change the mapstats totals, so stop instead
                    fprintf(stderr, "bmbs_search: record %ld of this batch has an empty or longer-than-1000-character sequence line: not supported\n", bad[(size_t)t]);
                    bail(nullptr); return;
                }
            }
            b->maxL = maxL; b->uniform = minL == maxL;
            int k = (int)(uint64_t)(P.e_f * maxL); if (k > 31) k = 31;
            b->k = k;
            b->max_ops = std::max(8, (int)bmbs_max_cigar_ops(&P, maxL));
            const size_t pool_ops = (size_t)nrec * (size_t)b->max_ops * (pe ? 2 : 1);
            if (!b->res.need((size_t)nrec * sizeof(bmbs_result) * (pe ? 2 : 1) + 64) || !b->pool.need(pool_ops * 4 + 64)) { bail("cannot allocate page-locked staging memory"); return; }
            t_read += now() - t0;
            ev_r.push_back({'R', nrec, t0, now()});
            const bool end = b->end;
            gpu_q.put(b);
            if (end) return;
        }
    });

    // ---------------- stage F: SAM text, formatted by the I/O threads into per-slice buffers ---------------------
    std::thread formatter([&] {
        Pool pool(std::max(1, io_threads - io_threads / 4 - 2) - 1);
        size_t max_chrom = 0;
        for (const auto& c : chrom_names) max_chrom = std::max(max_chrom, c.size());
        for (;;) {
            const double tw0 = now();
            Batch* b = out_q.get();
            const double t0 = now();
            t_wait_f += t0 - tw0;
            const long nrec = b->n;
            const bool end = b->end;
            const int T = pool.size() * 2;
            b->text.resize((size_t)T); b->text_len.assign((size_t)T, 0);
            if (bam) { b->bam_rec.resize((size_t)T); b->bam_z.resize((size_t)T); }
            if (nrec && !failed) {
                const long per = (nrec + T - 1) / T;
                const bmbs_result* res = (const bmbs_result*)b->res.p;
                const uint32_t* cpool = (const uint32_t*)b->pool.p;
                pool.run(T, [&](int t) {
                    const long a = std::min<long>(nrec, per * t), e = std::min<long>(nrec, a + per);
                    // upper bound of this slice's text: name + fixed columns + CIGAR + SEQ + QUAL per line
                    size_t bound = 64;
                    for (long r = a; r < e; r++) {
                        const size_t nl = b->l1.end((size_t)r * 4) - b->l1.start((size_t)r * 4);
                        bound += ((size_t)(pe ? 2 : 1)) * (nl + max_chrom + 2 * (size_t)b->maxL + 6 * (size_t)b->max_ops + 128);
                    }
                    std::vector<char>& buf = b->text[(size_t)t];
                    if (buf.size() < bound) buf.resize(bound + bound / 8);
                    Out o{buf.data()};
                    const char* t1 = b->l1.p;
                    const char* t2 = b->l2.p;
                    for (long r = a; r < e; r++) {
                        const int L = (int)b->r1.seq_len[r], L2 = pe ? (int)b->r2.seq_len[r] : 0;
                        const char* s1 = t1 + b->r1.seq_off[r];
                        const char* q1 = t1 + b->r1.qual_off[r];
                        const int ql1 = (int)b->r1.qual_len[r];
                        const char* nm = t1 + b->l1.start((size_t)r * 4);
                        size_t nl = b->l1.end((size_t)r * 4) - b->l1.start((size_t)r * 4);
                        if (!pe) {
                            const bmbs_result& x = res[r];
                            const bool mapped = x.status == BMBS_ST_UNIQUE || (x.status == BMBS_ST_AMBIG && ambiguous_out);
                            if (!mapped && !(unmapped_out && x.status != BMBS_ST_AMBIG)) continue;
                            size_t c = 0;                               // cut at the first ' ' or '/' (Process_Reads.cpp:843-850)
                            while (c < nl && nm[c] != ' ' && nm[c] != '/') c++;
                            nl = c;
                            if (nl && nm[0] == '@') { nm++; nl--; }
                            o.mem(nm, nl); o.ch('\t');
                            if (!mapped) {
                                // output_sam_unmapped (Schema.cpp:23955-23975); pbat prints the record as it was read (25538-25543)
                                o.lit("4\t*\t0\t0\t*\t*\t0\t0\t");
                                o.seq(s1, q1, L, ql1, false);
                                o.ch('\n');
                                continue;
                            }
                            o.num(x.flag); o.ch('\t'); o.str(chrom_names[(size_t)x.chrom]); o.ch('\t'); o.num(x.pos); o.ch('\t');
                            o.num(x.mapq); o.ch('\t'); o.cigar(x, cpool, L); o.lit("\t*\t0\t0\t");
                            // a --pbat read was mapped as the reverse complement of the text: flag 16 prints the text as it is
                            o.seq(s1, q1, L, ql1, ((x.flag & 16) != 0) != pbat_se);
                            o.lit("\tNM:i:"); o.num(x.nm); o.ch('\n');
                        } else {
                            const bmbs_result &x1 = res[2 * r], &x2 = res[2 * r + 1];
                            const bool mapped = x1.status == BMBS_ST_UNIQUE || (x1.status == BMBS_ST_AMBIG && ambiguous_out);
                            if (!mapped && !(unmapped_out && x1.status != BMBS_ST_AMBIG)) continue;
                            const char* s2 = t2 + b->r2.seq_off[r];
                            const char* q2 = t2 + b->r2.qual_off[r];
                            const int ql2 = (int)b->r2.qual_len[r];
                            const char* nm2 = t2 + b->l2.start((size_t)r * 4);
                            const size_t nl2 = b->l2.end((size_t)r * 4) - b->l2.start((size_t)r * 4);
                            size_t c = 0;                               // first differing char, ' ' or '/' (Process_Reads.cpp:296-307)
                            while (c < nl && c < nl2 && nm[c] == nm2[c] && nm[c] != ' ' && nm[c] != '/') c++;
                            nl = c;
                            if (nl && nm[0] == '@') { nm++; nl--; }
                            if (!mapped) {
                                // directly_output_unmapped_PE (Schema.cpp:10392-10430)
                                o.mem(nm, nl); o.lit("\t77\t*\t0\t0\t*\t*\t0\t0\t");
                                o.seq(s1, q1, L, ql1, false); o.ch('\n');
                                o.mem(nm, nl); o.lit("\t141\t*\t0\t0\t*\t*\t0\t0\t");
                                o.seq(s2, q2, L2, ql2, false); o.ch('\n');
                                continue;
                            }
                            const unsigned tlen = x1.tlen;
                            o.mem(nm, nl); o.ch('\t');
                            o.num(x1.flag); o.ch('\t'); o.str(chrom_names[(size_t)x1.chrom]); o.ch('\t'); o.num(x1.pos); o.ch('\t');
                            o.num(x1.mapq); o.ch('\t'); o.cigar(x1, cpool, L); o.lit("\t=\t"); o.num(x2.pos); o.ch('\t');
                            if (x2.pos < x1.pos) o.ch('-');             // TLEN sign, Schema.cpp:10575-10600
                            o.num(tlen); o.ch('\t');
                            o.seq(s1, q1, L, ql1, !(x1.flag & 32));
                            o.lit("\tNM:i:"); o.num(x1.nm); o.ch('\n');
                            o.mem(nm, nl); o.ch('\t');
                            o.num(x2.flag); o.ch('\t'); o.str(chrom_names[(size_t)x2.chrom]); o.ch('\t'); o.num(x2.pos); o.ch('\t');
                            o.num(x2.mapq); o.ch('\t'); o.cigar(x2, cpool, L2); o.lit("\t=\t"); o.num(x1.pos); o.ch('\t');
                            if (!(x1.pos > x2.pos)) o.ch('-');           // Schema.cpp:11530-11555
                            o.num(tlen); o.ch('\t');
                            o.seq(s2, q2, L2, ql2, (x2.flag & 16) != 0);
                            o.lit("\tNM:i:"); o.num(x2.nm); o.ch('\n');
                        }
                    }
                    size_t tl = (size_t)(o.p - buf.data());
                    if (bam && tl) {
                        // the slice's SAM lines -> BAM records -> BGZF blocks (records may straddle blocks)
                        std::vector<char>& rec = b->bam_rec[(size_t)t];
                        std::vector<char>& z = b->bam_z[(size_t)t];
                        rec.clear(); z.clear();
                        const char* q = buf.data();
                        const char* e2 = q + tl;
                        while (q < e2) {
                            const char* nlp = (const char*)memchr(q, '\n', (size_t)(e2 - q));
                            const size_t ll = nlp ? (size_t)(nlp - q) : (size_t)(e2 - q);
                            sam_line_to_bam(q, ll, bam_refs, rec);
                            q += ll + 1;
                        }
                        bgzf_append(rec.data(), rec.size(), z);
                        if (buf.size() < z.size()) buf.resize(z.size());
                        memcpy(buf.data(), z.data(), z.size());
                        tl = z.size();
                    }
                    b->text_len[(size_t)t] = tl;
                });
            }
            t_format += now() - t0;
            ev_f.push_back({'F', nrec, t0, now()});
            wr_q.put(b);
            if (end) return;
        }
    });

    // ---------------- stage W: the slices go into the output file at their prefix offsets -----------------------------
    // write()/pwrite() to ONE file take its inode lock exclusively: N threads writing N slices run one after the other at the
    // speed of one memcpy (tools/host_mem_probe on the MI355X box: 10.5 GB/s with 1, 8, 16 or 32 threads; tmpfs 5.4 GB/s) -- that,
    // 30 M SAM records of 350 bytes per second, is the ceiling of a file-to-file run.  The alternative, a shared mapping of the
    // file filled by several threads (BMBS_MMAP_OUT=1), measured WORSE on that box: 8.4 GB/s with one thread, 2.3 GB/s with 8-16
    // (page-fault path of the overlay file system), so pwrite stays the default.
    struct stat ost;
    const bool out_mappable = fstat(ofd, &ost) == 0 && S_ISREG(ost.st_mode) && getenv("BMBS_MMAP_OUT");
    std::thread writer([&] {
        Pool pool(out_mappable ? std::max(1, io_threads / 4) - 1 : 1);     // pwrite()s to one file run one at a time anyway
        for (;;) {
            const double tw0 = now();
            Batch* b = wr_q.get();
            const double t0 = now();
            t_wait_w += t0 - tw0;
            const bool end = b->end;
            const int T = (int)b->text.size();
            if (b->n && !failed) {
                std::vector<size_t> at((size_t)T + 1, out_off);
                for (int t = 0; t < T; t++) at[(size_t)t + 1] = at[(size_t)t] + b->text_len[(size_t)t];
                const size_t total = at[(size_t)T] - out_off;
                char* map = nullptr;
                size_t map_lo = 0, map_len = 0;
                if (out_mappable && total) {
                    map_lo = out_off & ~(size_t)4095; map_len = at[(size_t)T] - map_lo;
                    if (ftruncate(ofd, (off_t)at[(size_t)T]) == 0) {
                        void* m = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, ofd, (off_t)map_lo);
                        if (m != MAP_FAILED) map = (char*)m;
                    }
                }
                if (map) {
                    // equal byte ranges, not slices: the copy is balanced whatever the slices' sizes are
                    const int W = pool.size() * 2;
                    const size_t per = (total + (size_t)W - 1) / (size_t)W;
                    pool.run(W, [&](int w) {
                        size_t lo = out_off + std::min(total, per * (size_t)w);
                        const size_t hi = out_off + std::min(total, per * (size_t)(w + 1));
                        int t = (int)(std::upper_bound(at.begin(), at.end(), lo) - at.begin()) - 1;
                        while (lo < hi && t < T) {
                            const size_t stop = std::min(hi, at[(size_t)t + 1]);
                            if (stop > lo) memcpy(map + (lo - map_lo), b->text[(size_t)t].data() + (lo - at[(size_t)t]), stop - lo);
                            lo = std::max(lo, stop);
                            t++;
                        }
                    });
                    munmap(map, map_len);
                } else {
                    pool.run(T, [&](int t) {
                        const char* d = b->text[(size_t)t].data();
                        const size_t len = b->text_len[(size_t)t];
                        size_t done = 0;
                        while (done < len) {
                            const ssize_t w = pwrite(ofd, d + done, len - done, (off_t)(at[(size_t)t] + done));
                            if (w <= 0) { failed = true; break; }
                            done += (size_t)w;
                        }
                    });
                }
                out_off = at[(size_t)T];
            }
            t_write += now() - t0;
            ev_w.push_back({'W', b->n, t0, now()});
            free_q.put(b);
            if (end) return;
        }
    });

    // ---------------- stage G: one worker per context, one library call per batch ---------------------------------------
    std::atomic<long> total_records_a(0);
    std::mutex g_mu;
    auto g_worker = [&](bmbs_ctx* ctx) {
        for (;;) {
            const double tw0 = now();
            Batch* b = gpu_q.get();
            if (!b) return;                                                // another worker has seen the last batch
            const double t0 = now();
            { std::lock_guard<std::mutex> l(g_mu); t_wait_g += t0 - tw0; }
            if (!failed && b->n) {
                int64_t used = 0;
                int rc;
                bmbs_fastq_view v1 = {b->text1.p, (uint64_t)b->l1.used, b->r1.seq_off, b->r1.qual_off, b->r1.seq_len, b->r1.qual_len};
                const int64_t cap = (int64_t)b->n * b->max_ops * (pe ? 2 : 1);
                if (!pe)
                    rc = bmbs_map_se_fastq(ctx, &v1, b->n, b->maxL, b->uniform ? 1 : 0, pbat_se ? 1 : 0, (bmbs_result*)b->res.p, (uint32_t*)b->pool.p, cap, &used);
                else {
                    bmbs_fastq_view v2 = {b->text2.p, (uint64_t)b->l2.used, b->r2.seq_off, b->r2.qual_off, b->r2.seq_len, b->r2.qual_len};
                    rc = bmbs_map_pe_fastq(ctx, &v1, &v2, b->n, b->maxL, b->uniform ? 1 : 0, (bmbs_result*)b->res.p, (uint32_t*)b->pool.p, cap, &used);
                }
                if (rc) { fprintf(stderr, "%s\n", bmbs_last_error(ctx)); failed = true; }
            }
            total_records_a += b->n;
            {
                std::lock_guard<std::mutex> l(g_mu);
                t_gpu += now() - t0;
                ev_g.push_back({'G', b->n, t0, now()});
            }
            const bool end = b->end;
            out_q.put(b->seq, b);
            if (end) { for (size_t i = 1; i < ctxs.size(); i++) gpu_q.put(nullptr); return; }
        }
    };
    {
        std::vector<std::thread> workers;
        for (size_t i = 1; i < ctxs.size(); i++) workers.emplace_back(g_worker, ctxs[i]);
        g_worker(ctxs[0]);
        for (auto& t : workers) t.join();
    }
    const long total_records = total_records_a.load();
    reader.join();
    formatter.join();
    writer.join();
    if (bam && !failed) {
        static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (pwrite(ofd, eof_block, 28, (off_t)out_off) != 28) failed = true;
        out_off += 28;
    }
    const double t_joined = now();
    ::close(ofd);
    src1.close();
    if (pe) src2.close();
    if (failed) { fprintf(stderr, "bmbs_search: failed\n"); return 1; }
    int64_t st[5];
    bmbs_stats_allreduce(ctxs.data(), (int)ctxs.size(), st);      // get_mapping_informations: the counters of every worker summed
    print_stats(stderr, st);
    if (!mapstats.empty()) { FILE* m = fopen(mapstats.c_str(), "w"); if (m) { print_stats(m, st); fclose(m); } }
    const double t_end = now();
    if (verbose)
        fprintf(stderr, "[bmbs_search] records %ld  load+attach %.3fs  mapping wall %.3fs  (pipeline %.3fs; stage busy: read/pack %.3fs of which line index %.3fs, gpu %.3fs, format %.3fs, write %.3fs)  %d I/O threads, batch %ld, %zu device(s) x %d context(s)\n",
                total_records, t_loaded - t_start, t_end - t_loaded, t_joined - t_loaded, t_read, t_index, t_gpu, t_format, t_write, io_threads, batch,
                n_owner, contexts);
    if (verbose)
        fprintf(stderr, "[bmbs_search] read stage: window %.3fs, line index %.3fs; stage idle (waiting for a batch): read %.3fs, gpu workers %.3fs (summed), format %.3fs, write %.3fs\n",
                t_window, t_index - t_window, t_wait_r, t_wait_g, t_wait_f, t_wait_w);
    if (verbose && getenv("BMBS_TRACE"))
        for (const auto* v : {&ev_r, &ev_g, &ev_f, &ev_w})
            for (const Ev& e : *v) fprintf(stderr, "[trace] %c n=%ld %.4f .. %.4f\n", e.stage, e.n, e.a - t_loaded, e.b - t_loaded);
    const double t0 = now();
    for (auto& b : batches) { b.text1.release(); b.text2.release(); b.r1.mem.release(); b.r2.mem.release(); b.res.release(); b.pool.release(); }
    const double t1 = now();
    for (size_t i = ctxs.size(); i-- > 0;) bmbs_destroy(ctxs[i]);     // the sharing contexts go before their owners
    const double t2 = now();
    bmbs_index_file_free(ixf);
    if (verbose) fprintf(stderr, "[bmbs_search] teardown: unpin %.3fs, destroy ctx %.3fs, free index %.3fs\n", t1 - t0, t2 - t1, now() - t2);
    return 0;
}
