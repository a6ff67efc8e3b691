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
//          one overlap the kernels of another; default 4), --batch N (records per GPU batch, default 500 k), -t N (host I/O
//          threads), --out-parts N (the input is cut into N contiguous record ranges, each written to its own file
//          <out>.part000 ... concurrently; `cat` of the parts in order == the one-file output), --verbose
//
// The reference has ONE reader thread and ONE fprintf sink (Process_Reads.cpp:2057-2260, Process_sam_out.cpp:954-1006), which is
// what limits it.  Round 2 of this driver indexed the lines and formatted the SAM text with the host's
// I/O threads; those two stages were as long as the GPU stage.  Now the host touches file bytes only to move them:
//   stage R  pread of a window of the FASTQ file(s) into a page-locked buffer (parallel; .gz: one inflate thread per file); the
//            reader COUNTS the newlines (SWAR over the bytes it has just read) to know how many whole records the window holds
//   stage G  one bmbs_map_se_text / bmbs_map_pe_text call per batch: text up, newline index + rows + mapping + SAM formatting on
//            the device (bmbs_text.hip), finished SAM text down into a page-locked buffer; one worker thread per context
//   stage W  pwrite of the SAM text at the part's running offset (--bam: the text is converted to BAM records and BGZF blocks by
//            the I/O threads first, bam_prase.cpp:201-221)
// Buffered writes to ONE file serialise on its inode lock at the speed of one memcpy (~10 GB/s = 28 M SAM records/s on the MI355X
// boxes): --out-parts N gives N inodes.  Every part is a pipeline of its own (reader -> shared GPU workers -> writer) over its own
// record range of the input; for pairs the ranges are cut at the same record in both files (found by the read names, by counting
// lines when the names do not tell).
#include "../../include/bmbs.h"
#include "pgz.h"
#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
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


// ---- newline counting: 64 bytes per step (SSE2: compare, move mask, one population count per 64 bytes) -----------------------------
// (the 8-bytes-per-step SWAR form this replaces ran at 2.4 GB/s per core -- without -mpopcnt every population count is a library
// call -- and sixteen cores' worth of it was what the readers of a FASTQ -> SAM run were busy with; this form: 18 GB/s per core)
inline size_t count_nl(const char* p, size_t n)
{
    size_t c = 0, i = 0;
#if defined(__x86_64__)
    const __m128i nl = _mm_set1_epi8('\n');
    for (; i + 64 <= n; i += 64) {
        const uint64_t a = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i)), nl));
        const uint64_t b = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i + 16)), nl));
        const uint64_t d = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i + 32)), nl));
        const uint64_t e = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i + 48)), nl));
        c += (size_t)__builtin_popcountll(a | (b << 16) | (d << 32) | (e << 48));
    }
#endif
    for (; i < n; i++) c += p[i] == '\n';                 // (the whole buffer on a host without SSE2)
    return c;
}
// offset just behind the k-th newline (k >= 1) of p[0, n), n when there are fewer
inline size_t after_kth_nl(const char* p, size_t n, size_t k)
{
    const char* q = p; const char* e = p + n;
    while (k && q < e) { const char* h = (const char*)memchr(q, '\n', (size_t)(e - q)); if (!h) return n; q = h + 1; k--; }
    return k ? n : (size_t)(q - p);
}

struct Pinned {                                  // page-locked staging (bmbs_host_alloc_kind)
    char* p = nullptr; size_t cap = 0; int kind = 0;
    bool need(size_t bytes)
    {
        if (bytes <= cap) return true;
        if (p) bmbs_host_free(p);
        cap = bytes + bytes / 4 + 4096;
        p = (char*)bmbs_host_alloc_kind(cap, kind);
        if (!p) { cap = 0; return false; }
        return true;
    }
    void release() { if (p) bmbs_host_free(p); p = nullptr; cap = 0; }
};

// ---- FASTQ text source over a byte range of a file.  Plain files: parallel pread()s straight into the batch's page-locked window,
// the newlines counted per 64 KiB block by the thread that has just read it.  .gz: a thread of its own inflates into a queue of
// chunks (so that the two files of a paired-end run inflate side by side) and the window is assembled from them. ---------------
#define SUB_BLOCK ((size_t)1 << 16)
// --loop-input N (measurement aid; plain FASTQ, and BGZF input that is inflated on the device): a part's byte range is read N times over, so that a run lasts seconds on an input
// that fits the page cache (the pipeline's fill and the contexts' first calls then weigh what they weigh in a real run)
static int g_loop_input = 1;
struct Source {
    bool gz = false;
    size_t lo0 = 0; int loops_left = 0;
    int fd = -1;
    size_t size = 0, off = 0, end = 0;           // plain: the part's byte range [off, end)
    std::string err;
    // .gz: inflated text arrives as numbered chunks; `ready` hands them to window() in order
    std::vector<std::thread> inflaters;
    std::mutex m; std::condition_variable cv_data, cv_room;
    std::map<long, std::vector<char>> ready;     // chunk number -> text
    int zdev = -1;                               // >= 0: BGZF blocks are inflated on this HIP device (bmbs_inflate_bgzf)
    // ... a window at a time, straight into the batch's page-locked window: no inflater threads, no chunk queue, and the newline counts
    // come back with the text -- the host moves the compressed bytes into a staging buffer and nothing else
    bool zdirect = false;
    bmbs_ctx* zc = nullptr;
    Pinned zstage;
    std::vector<uint64_t> zblk, zout;
    long next_chunk = 0;                         // the chunk window() takes next
    long want_chunk = 0;                         // the chunk window() is waiting for (always admitted by push_chunk)
    size_t queued = 0, front_used = 0;
    bool gz_done = false, gz_stop = false;
    int live_inflaters = 0;
    std::vector<char> carry;
    // BGZF (bgzip): independent deflate blocks of <= 64 KiB with their compressed size in the header -- inflated by several threads
    bool bgzf = false;
    const unsigned char* zmap = nullptr; size_t zsize = 0, znext = 0;     // the compressed file, mapped; next unassigned block
    long zjob = 0;                                                        // number of the next job
    // ordinary gzip (one deflate stream per member): block-parallel inflate, pgz.h
    std::unique_ptr<pgz::Engine> pgz_eng;
    std::thread pgz_watch;
    int gz_threads_ = 1;

    static bool is_gz(const char* path)
    {
        FILE* f = fopen(path, "rb");
        if (!f) return false;
        unsigned char mg[2] = {0, 0};
        const size_t got = fread(mg, 1, 2, f);
        fclose(f);
        return got == 2 && mg[0] == 0x1f && mg[1] == 0x8b;
    }
    // compressed size of the BGZF block at p (0: not a BGZF block header)
    static size_t bgzf_block(const unsigned char* p, size_t avail)
    {
        if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4) || p[10] != 6 || p[11] != 0 || p[12] != 'B' || p[13] != 'C' || p[14] != 2 || p[15] != 0) return 0;
        const size_t bs = (size_t)(p[16] | (p[17] << 8)) + 1;
        return bs >= 26 && bs <= avail ? bs : 0;
    }
    static size_t bgzf_isize(const unsigned char* p, size_t bs) { return (size_t)p[bs - 4] | ((size_t)p[bs - 3] << 8) | ((size_t)p[bs - 2] << 16) | ((size_t)p[bs - 1] << 24); }
    // the gzip members from byte `at` of the mapped file on, chunk numbers from `first_id` (called with `m` held)
    void start_pgz(size_t at, long first_id)
    {
        if (gz_stop || pgz_eng) return;
        pgz::Options o; o.threads = gz_threads_; o.span = (size_t)1 << 20;
        if (const char* sp = getenv("BMBS_GZ_SPAN")) { const long v = atol(sp); if (v >= 1024) o.span = (size_t)v; }      // tests: many spans in a small file
        pgz_eng.reset(new pgz::Engine(zmap, zsize, at, first_id, o, [this](long id, std::vector<char>&& c) { push_chunk(id, std::move(c)); }));
        live_inflaters++;
        pgz_eng->start();
        pgz_watch = std::thread([this] {
            const std::string e = pgz_eng->wait();
            if (!e.empty()) { std::lock_guard<std::mutex> l(m); if (err.empty()) err = e; }
            inflater_exit();
        });
    }
    void push_chunk(long id, std::vector<char>&& c)
    {
        std::unique_lock<std::mutex> l(m);
        // the chunk window() is waiting for always gets in; the others wait for room (text inflated ahead of its turn is bounded)
        cv_room.wait(l, [&] { return id == want_chunk || queued < ((size_t)768 << 20) || gz_stop; });
        if (gz_stop) return;
        queued += c.size();
        ready[id] = std::move(c);
        cv_data.notify_all();
    }
    void inflater_exit()
    {
        std::lock_guard<std::mutex> l(m);
        if (--live_inflaters == 0) { gz_done = true; cv_data.notify_all(); }
    }
    // a one-member .gz file that is not taken by the device path after all: the host's block-parallel inflater from its first byte
    void host_stream() { std::lock_guard<std::mutex> l(m); start_pgz(0, 0); }
    bool open(const char* path, size_t lo, size_t hi, int gz_threads = 1, int device = -1)
    {
        zdev = device;
        gz = is_gz(path);
        if (gz) {
            const int zfd = ::open(path, O_RDONLY);
            struct stat zsb;
            bool zmap_keep = false;
            if (zfd >= 0 && fstat(zfd, &zsb) == 0 && zsb.st_size >= 18) {
                void* mp = mmap(nullptr, (size_t)zsb.st_size, PROT_READ, MAP_PRIVATE, zfd, 0);
                if (mp != MAP_FAILED) {
                    zmap = (const unsigned char*)mp; zsize = (size_t)zsb.st_size; zmap_keep = true;
                    (void)madvise(mp, zsize, MADV_SEQUENTIAL);
                    bgzf = bgzf_block(zmap, zsize) != 0;
                }
            }
            if (zfd >= 0) ::close(zfd);
            if (!zmap_keep) return false;
            gz_threads_ = std::max(1, gz_threads);
            if (bgzf) {
                const char* zd = getenv("BMBS_GZ_DEVICE");
                if (zd && !strcmp(zd, "0")) zdev = -1;
                if (zdev >= 0) { zdirect = true; zstage.kind = 1; loops_left = g_loop_input - 1; return true; }      // inflated on the device, a window at a time (no threads here)
                const int n_inflaters = gz_threads_;
                live_inflaters = n_inflaters;                   // (the threads count it down as they finish: not the loop bound)
                for (int t = 0; t < n_inflaters; t++)
                    inflaters.emplace_back([this] {
                        // every block is a gzip member of its own: the driver's own inflater (pgz.h) on known bytes -- no window in
                        // front of a block, no markers -- and the member's CRC-32 checked by carry-less multiplication
                        pgz::OutBuf<pgz::u8> ob;
                        std::unique_ptr<pgz::Tables> dyn(new pgz::Tables);
                        std::vector<pgz::MemberEnd> ends;
                        for (;;) {
                            // a job = the blocks of the next ~2 MiB of the compressed file
                            size_t a, e; long id;
                            {
                                std::lock_guard<std::mutex> l(m);
                                if (gz_stop || znext >= zsize) break;
                                a = znext; id = zjob;
                                size_t q = a;
                                while (q < zsize && q - a < ((size_t)2 << 20)) { const size_t bs = bgzf_block(zmap + q, zsize - q); if (!bs) break; q += bs; }
                                if (q == a) {
                                    // not a BGZF block: a file whose later members are ordinary gzip goes on through the stream inflater
                                    znext = zsize;
                                    if (pgz::gzip_header(zmap, zsize, a)) start_pgz(a, id);
                                    else err = "corrupt BGZF block header in the .gz input";
                                    break;
                                }
                                zjob++; e = q; znext = q;
                            }
                            bool bad = false;
                            std::vector<char> out;
                            try {
                                size_t total = 0;
                                for (size_t q = a; q < e;) { const size_t bs = bgzf_block(zmap + q, zsize - q); const size_t isz = bgzf_isize(zmap + q, bs); if (isz > 65536) bad = true; total += isz; q += bs; }
                                if (!bad) out.resize(total);
                                size_t at = 0;
                                for (size_t q = a; q < e && !bad;) {
                                    const size_t bs = bgzf_block(zmap + q, zsize - q);
                                    const size_t isz = bgzf_isize(zmap + q, bs);
                                    if (isz) {
                                        if (!ob.mem) { if (!ob.reserve(70000)) throw std::bad_alloc(); memset(ob.mem, 0, pgz::WIN); }
                                        ob.n = 0; ob.mstart = 0; ob.reach = 0; ends.clear();      // (a member of its own: nothing in front of it)
                                        // (the deflate data starts behind the WHOLE member header: a block may carry a name, a comment or a header CRC too)
                                        const size_t body = pgz::gzip_header(zmap, q + bs, q);
                                        pgz::DecodeResult r; r.st = pgz::ST_ERROR; r.end_bit = 0; r.why = "";
                                        if (body) r = pgz::decode_blocks<pgz::u8>(zmap, q + bs, (pgz::u64)body * 8, ~(pgz::u64)0, ob, ends, *dyn);
                                        if (r.st != pgz::ST_END || ob.n != isz || ends.size() != 1 || ends[0].isize != (pgz::u32)isz ||
                                            ends[0].crc != pgz::crc32_fast(0, ob.out(), isz)) bad = true;
                                        else memcpy(out.data() + at, ob.out(), isz);
                                    }
                                    at += isz; q += bs;
                                }
                            } catch (const std::exception&) { bad = true; }
                            if (bad) { std::lock_guard<std::mutex> l(m); err = "corrupt BGZF block in the .gz input"; znext = zsize; break; }
                            push_chunk(id, std::move(out));
                        }
                        inflater_exit();
                    });
                return true;
            }
            host_stream();
            return true;
        }
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        if (fstat(fd, &sb)) return false;
        size = (size_t)sb.st_size;
        off = std::min(lo, size); end = std::min(hi, size);
        lo0 = off; loops_left = g_loop_input - 1;
        (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
        return true;
    }
    // up to `cap` bytes of text starting at the current record boundary into dst (which has 64 spare bytes behind cap); `last` when
    // they reach the end of the range; counts[i] = newlines of dst[i * SUB_BLOCK ...).  false: I/O error (err says which)
    bool window(Pool& pool, char* dst, size_t cap, size_t& len_out, bool& last, std::vector<uint32_t>& counts)
    {
        size_t len = 0;
        if (!gz) {
            if (off == end && loops_left > 0) { off = lo0; loops_left--; }
            len = std::min(cap, end - off);
            const size_t nsb = (len + SUB_BLOCK - 1) / SUB_BLOCK;
            counts.assign(nsb, 0);
            const int T = pool.size() * 2;
            const size_t per = ((nsb + (size_t)T - 1) / (size_t)T) * SUB_BLOCK;
            std::atomic<int> bad(0);
            pool.run(T, [&](int t) {
                size_t a = std::min(len, per * (size_t)t);
                const size_t e = std::min(len, a + per);
                while (a < e) {
                    const size_t stop = std::min(e, a + SUB_BLOCK);            // read one block, count it while it is in cache
                    size_t at = a;
                    while (at < stop) {
                        const ssize_t g = pread(fd, dst + at, stop - at, (off_t)(off + at));
                        if (g <= 0) { bad = g < 0 ? errno : EIO; return; }
                        at += (size_t)g;
                    }
                    counts[a / SUB_BLOCK] = (uint32_t)count_nl(dst + a, stop - a);
                    a = stop;
                }
            });
            if (bad) { err = std::string("read error on the FASTQ input: ") + strerror(bad); return false; }
            last = off + len == end && loops_left == 0;
        } else if (zdirect) {
            if (!zc) {
                // (only when the driver's two-phase path is not in use: a context of this source's own inflates into the host window)
                bmbs_params P0; bmbs_default_params(&P0);
                zc = bmbs_create(zdev, &P0);
                if (!zc) { err = "cannot create a context on the device for the BGZF input (BMBS_GZ_DEVICE=0 inflates on the host)"; return false; }
            }
            size_t have = carry.size();
            if (have > cap) { err = "internal: carried text larger than the window"; return false; }
            if (have) memcpy(dst, carry.data(), have);
            carry.clear();
            // the BGZF blocks whose text fits behind the carried bytes
            const size_t a = znext;
            size_t q = a; uint64_t text = 0;
            zblk.clear(); zout.clear(); zblk.push_back(0); zout.push_back(0);
            bool foreign = false;
            while (q < zsize) {
                const size_t bs = bgzf_block(zmap + q, zsize - q);
                if (!bs) { foreign = true; break; }
                const size_t isz = bgzf_isize(zmap + q, bs);
                if (isz > 65536) { err = "corrupt BGZF block in the .gz input"; return false; }
                if (have + text + isz > cap) break;
                q += bs; text += isz;
                zblk.push_back(q - a); zout.push_back(text);
            }
            if (foreign && q == a) {
                // a member that is not a BGZF block: the rest of the file goes through the host's stream inflater (chunks)
                if (!pgz::gzip_header(zmap, zsize, a)) { err = "corrupt BGZF block header in the .gz input"; return false; }
                zdirect = false;
                { std::lock_guard<std::mutex> l(m); znext = zsize; start_pgz(a, 0); }
                carry.assign(dst, dst + have);
                return window(pool, dst, cap, len_out, last, counts);
            }
            if (q == a && q < zsize) { err = "a BGZF block larger than the window"; return false; }
            znext = q;
            len = have + (size_t)text;
            last = znext >= zsize;
            const size_t nsb = (len + SUB_BLOCK - 1) / SUB_BLOCK;
            counts.assign(nsb + 1, 0);
            if (q > a) {
                const size_t zbytes = q - a;
                if (!zstage.need(zbytes + 64)) { err = "cannot allocate page-locked staging memory"; return false; }
                const int T = pool.size() * 2;
                const size_t per = ((zbytes + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
                pool.run(T, [&](int t) { const size_t x = std::min(zbytes, per * (size_t)t), y = std::min(zbytes, x + per); if (x < y) memcpy(zstage.p + x, zmap + a + x, y - x); });
                const int rc = bmbs_inflate_bgzf(zc, zstage.p, zbytes, zblk.data(), zout.data(), (int64_t)zblk.size() - 1, dst + have, (uint64_t)text, counts.data(), (uint64_t)have);
                if (rc) { err = bmbs_last_error(zc); return false; }
            }
            counts.resize(nsb);
            for (size_t i = 0; i * SUB_BLOCK < have; i++) counts[i] += (uint32_t)count_nl(dst + i * SUB_BLOCK, std::min(SUB_BLOCK, have - i * SUB_BLOCK));
        } else {
            size_t have = std::min(carry.size(), cap);
            if (carry.size() > cap) { err = "internal: carried text larger than the window"; return false; }
            // which pieces of which chunks make up the window (waiting for the inflaters as needed) ...
            struct Piece { const char* src; size_t len, dst; };
            std::vector<Piece> pieces;
            if (have) pieces.push_back(Piece{carry.data(), have, 0});
            bool done = false;
            long chunk = next_chunk; size_t used = front_used;
            std::vector<long> finished;
            while (have < cap) {
                std::unique_lock<std::mutex> l(m);
                if (want_chunk != chunk) { want_chunk = chunk; cv_room.notify_all(); }
                cv_data.wait(l, [&] { return ready.count(chunk) != 0 || gz_done; });
                auto it = ready.find(chunk);
                if (it == ready.end()) { done = true; if (!err.empty()) return false; break; }
                std::vector<char>& f = it->second;                                  // (only this thread erases: the chunk stays put)
                l.unlock();
                const size_t take = std::min(cap - have, f.size() - used);
                pieces.push_back(Piece{f.data() + used, take, have});
                have += take; used += take;
                if (used == f.size()) { finished.push_back(chunk); chunk++; used = 0; }
            }
            len = have;
            last = done;
            // ... then every thread copies its share of the window and counts the newlines of what it has just written (one thread
            // copying 300 MB windows of two files was the whole run time of gzipped input once the inflate ran on many threads)
            const size_t nsb = (len + SUB_BLOCK - 1) / SUB_BLOCK;
            counts.assign(nsb, 0);
            const int T = (int)std::min<size_t>(std::max<size_t>(nsb, 1), (size_t)pool.size() * 2);
            const size_t per = ((nsb + (size_t)T - 1) / (size_t)T) * SUB_BLOCK;
            pool.run(T, [&](int t) {
                const size_t a = std::min(len, per * (size_t)t), e = std::min(len, a + per);
                if (a >= e) return;
                size_t lo = 0, hi = pieces.size();                                  // first piece that reaches beyond a
                while (lo < hi) { const size_t mid = (lo + hi) / 2; if (pieces[mid].dst + pieces[mid].len <= a) lo = mid + 1; else hi = mid; }
                for (size_t i = lo; i < pieces.size() && pieces[i].dst < e; i++) {
                    const size_t x = std::max(a, pieces[i].dst), y = std::min(e, pieces[i].dst + pieces[i].len);
                    if (x < y) memcpy(dst + x, pieces[i].src + (x - pieces[i].dst), y - x);
                }
                for (size_t q = a; q < e; q += SUB_BLOCK) counts[q / SUB_BLOCK] = (uint32_t)count_nl(dst + q, std::min(SUB_BLOCK, e - q));
            });
            carry.clear();
            {
                std::lock_guard<std::mutex> l(m);
                for (long id : finished) { auto it = ready.find(id); if (it != ready.end()) { queued -= it->second.size(); ready.erase(it); } }
                next_chunk = chunk; want_chunk = chunk; front_used = used;
                cv_room.notify_all();
            }
        }
        // an unterminated last line counts as a line: the device wants every line closed
        if (last && len && dst[len - 1] != '\n') { dst[len] = '\n'; len++; if ((len - 1) / SUB_BLOCK >= counts.size()) counts.push_back(0); counts[(len - 1) / SUB_BLOCK]++; }
        len_out = len;
        return true;
    }
    void consumed(const char* p, size_t len, size_t used)
    {
        if (!gz) { off += std::min(used, end - off); return; }
        carry.assign(p + used, p + len);
    }
    void close()
    {
        if (gz) {
            { std::lock_guard<std::mutex> l(m); gz_stop = true; }
            cv_room.notify_all();
            if (pgz_eng) pgz_eng->stop();
            for (auto& t : inflaters) t.join();
            inflaters.clear();
            if (pgz_watch.joinable()) pgz_watch.join();
            pgz_eng.reset();
        }
        ready.clear();
        if (zmap) munmap((void*)zmap, zsize);
        zmap = nullptr;
        if (fd >= 0) ::close(fd);
        fd = -1; gz = false;
    }
    // the next BGZF blocks whose text fits into `room` bytes (at least one): their bytes are zmap[a, q); tables relative to a.
    // foreign: the file goes on with a member that is not a BGZF block.  false: corrupt header
    bool next_blocks(size_t room, size_t& a, size_t& q, bool& foreign)
    {
        a = znext; q = a; foreign = false;
        uint64_t text = 0;
        zblk.clear(); zout.clear(); zblk.push_back(0); zout.push_back(0);
        while (q < zsize) {
            const size_t bs = bgzf_block(zmap + q, zsize - q);
            if (!bs) { foreign = true; break; }
            const size_t isz = bgzf_isize(zmap + q, bs);
            if (isz > 65536) { err = "corrupt BGZF block in the .gz input"; return false; }
            if (text + isz > room && q > a) break;
            q += bs; text += isz;
            zblk.push_back(q - a); zout.push_back(text);
            if (text >= room) break;
        }
        if (foreign && q == a && !pgz::gzip_header(zmap, zsize, a)) { err = "corrupt BGZF block header in the .gz input"; return false; }
        return true;
    }
    // the device side of a compressed source (context, staging): released apart from close(), outside a driver's timed region
    void release_device()
    {
        if (zc) bmbs_destroy(zc);
        zc = nullptr;
        zstage.release();
    }
    ~Source() { close(); release_device(); }
};

// offset just behind the k-th newline of a window whose blocks have been counted
size_t after_kth_nl_blocks(const char* p, size_t len, const std::vector<uint32_t>& counts, size_t k)
{
    size_t acc = 0;
    for (size_t i = 0; i < counts.size(); i++) {
        if (acc + counts[i] >= k) { const size_t a = i * SUB_BLOCK; return a + after_kth_nl(p + a, std::min(SUB_BLOCK, len - a), k - acc); }
        acc += counts[i];
    }
    return len;
}

// ---- where the parts begin: record boundaries of plain FASTQ files ----------------------------------------------------------------
// first record start at or after `guess`: a line that begins with '@' whose second successor begins with '+' (a quality line may
// begin with '@', but then the line two further on is a sequence line, which cannot begin with '+')
size_t record_start_at(int fd, size_t size, size_t guess)
{
    if (guess == 0) return 0;
    if (guess >= size) return size;
    for (size_t span = (size_t)1 << 20; ; span *= 4) {
        const size_t a = guess - 1, n = std::min(span, size - a);           // from the byte before: a newline there makes `guess` a line start
        std::vector<char> buf(n);
        size_t got = 0;
        while (got < n) { const ssize_t g = pread(fd, buf.data() + got, n - got, (off_t)(a + got)); if (g <= 0) break; got += (size_t)g; }
        std::vector<size_t> ls;                                                // line starts inside the buffer
        for (size_t i = 0; i + 1 < got; i++) if (buf[i] == '\n') ls.push_back(i + 1);
        for (size_t i = 0; i + 2 < ls.size(); i++)
            if (buf[ls[i]] == '@' && buf[ls[i + 2]] == '+' && (i + 4 >= ls.size() || buf[ls[i + 4]] == '@')) return a + ls[i];
        if (a + n >= size) return size;
    }
}
// name of the record at `at`, cut like the paired-end reader does (first ' ' or '/')
std::string cut_name_at(int fd, size_t size, size_t at)
{
    char b[4096];
    const size_t n = std::min(sizeof b, size - at);
    const ssize_t g = pread(fd, b, n, (off_t)at);
    std::string s;
    for (ssize_t i = 0; i < g && b[i] != '\n' && b[i] != ' ' && b[i] != '/'; i++) s += b[i];
    return s;
}
size_t next_record(int fd, size_t size, size_t at)       // start of the record after the one at `at`
{
    size_t pos = at; int lines = 0;
    char b[1 << 16];
    while (pos < size && lines < 4) {
        const ssize_t g = pread(fd, b, sizeof b, (off_t)pos);
        if (g <= 0) break;
        for (ssize_t i = 0; i < g; i++) if (b[i] == '\n' && ++lines == 4) return pos + (size_t)i + 1;
        pos += (size_t)g;
    }
    return size;
}
size_t count_lines(Pool& pool, int fd, size_t a, size_t b)                 // newlines of file bytes [a, b)
{
    const size_t blk = (size_t)16 << 20, nb = (b - a + blk - 1) / blk;
    std::vector<size_t> c(nb, 0);
    const int T = (int)std::min<size_t>(nb, (size_t)pool.size() * 2);
    pool.run(T, [&](int t) {
        std::vector<char> buf(blk);
        for (size_t i = (size_t)t; i < nb; i += (size_t)T) {
            const size_t lo = a + i * blk, n = std::min(blk, b - lo);
            size_t got = 0;
            while (got < n) { const ssize_t g = pread(fd, buf.data() + got, n - got, (off_t)(lo + got)); if (g <= 0) break; got += (size_t)g; }
            c[i] = count_nl(buf.data(), got);
        }
    });
    size_t s = 0;
    for (size_t x : c) s += x;
    return s;
}
size_t offset_of_line(Pool& pool, int fd, size_t size, size_t line)          // offset of the first byte of line `line` (0-based)
{
    if (line == 0) return 0;
    const size_t blk = (size_t)16 << 20, nb = (size + blk - 1) / blk;
    std::vector<size_t> c(nb, 0);
    const int T = (int)std::min<size_t>(nb, (size_t)pool.size() * 2);
    pool.run(T, [&](int t) {
        std::vector<char> buf(blk);
        for (size_t i = (size_t)t; i < nb; i += (size_t)T) {
            const size_t lo = i * blk, n = std::min(blk, size - lo);
            size_t got = 0;
            while (got < n) { const ssize_t g = pread(fd, buf.data() + got, n - got, (off_t)(lo + got)); if (g <= 0) break; got += (size_t)g; }
            c[i] = count_nl(buf.data(), got);
        }
    });
    size_t acc = 0;
    for (size_t i = 0; i < nb; i++) {
        if (acc + c[i] >= line) {
            const size_t lo = i * blk, n = std::min(blk, size - lo);
            std::vector<char> buf(n);
            size_t got = 0;
            while (got < n) { const ssize_t g = pread(fd, buf.data() + got, n - got, (off_t)(lo + got)); if (g <= 0) break; got += (size_t)g; }
            return lo + after_kth_nl(buf.data(), got, line - acc);
        }
        acc += c[i];
    }
    return size;
}
// the record of file 2 that pairs with the record starting at b1 of file 1: looked for by name around the proportional offset
// (two consecutive names have to agree and the match has to be the only one in the window); counted when the names do not tell
size_t mate_boundary(Pool& pool, int fd1, size_t size1, size_t b1, int fd2, size_t size2)
{
    if (b1 == 0) return 0;
    if (b1 >= size1) return size2;
    const std::string n0 = cut_name_at(fd1, size1, b1);
    const size_t b1n = next_record(fd1, size1, b1);
    const std::string n1 = b1n < size1 ? cut_name_at(fd1, size1, b1n) : std::string();
    const size_t g2 = (size_t)((double)size2 * ((double)b1 / (double)size1));
    for (size_t W = (size_t)2 << 20; W <= ((size_t)64 << 20) && !n0.empty(); W *= 4) {
        size_t lo = g2 > W ? record_start_at(fd2, size2, g2 - W) : 0;
        const size_t hi = std::min(size2, g2 + W);
        size_t found = size2 + 1; int hits = 0;
        for (size_t at = lo; at < hi && at < size2; at = next_record(fd2, size2, at)) {
            if (cut_name_at(fd2, size2, at) != n0) continue;
            const size_t nx = next_record(fd2, size2, at);
            if (!n1.empty() && (nx >= size2 || cut_name_at(fd2, size2, nx) != n1)) continue;
            hits++; found = at;
        }
        if (hits == 1) return found;
        if (hits > 1) break;                                                  // names repeat: they do not identify a record
    }
    const size_t lines = count_lines(pool, fd1, 0, b1);                       // b1 is a record start: lines % 4 == 0
    return offset_of_line(pool, fd2, size2, lines);
}

inline void put_uint(std::string& s, unsigned long long v)
{
    char b[24]; int i = 24;
    do { b[--i] = (char)('0' + v % 10); v /= 10; } while (v);
    s.append(b + i, (size_t)(24 - i));
}

// ---- --bam (Process_CommandLines.cpp:94; the reference hands each SAM line to htslib's sam_parse1 and bam_write1,
// bam_prase.cpp:201-221): the records are built and deflated into BGZF blocks ON THE DEVICE (BMBS_TEXT_BAM, bmbs_bam.hip); the host
// writes the BAM header (below) and the end-of-file block, and moves the device's bytes into the file ----------
inline void put_le32(std::vector<char>& o, uint32_t v) { char b[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)}; o.insert(o.end(), b, b + 4); }

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

struct Part;
struct Batch {
    Part* part = nullptr;
    long seq = 0;                                // position within the part (output order)
    long n = 0;                                  // records (pairs)
    bool end = false;                            // the part's input ends with this batch
    Pinned text1, text2, sam;                    // FASTQ windows in, SAM text out
    size_t used1 = 0, used2 = 0;
    uint64_t sam_bytes = 0;
    std::vector<uint32_t> counts1, counts2;
    bmbs_ctx* open_ctx = nullptr;                // compressed input kept on the device: the context that holds this batch's open window
};

struct Part {                                    // one contiguous record range of the input -> one output file
    int id = 0;
    Source s1, s2;
    int ofd = -1;
    size_t out_off = 0;
    size_t alloc_end = 0;                        // the file's blocks are reserved up to here (fallocate ahead of the writers)
    bool can_alloc = true, regular = false;
    OrderedChan<Batch*> out_q;
    long next_seq = 0;
    std::thread reader, writer;
    double t_read = 0, t_write = 0, t_format = 0, t_wait_r = 0, t_wait_w = 0;
    long records = 0;
};

}  // namespace

#ifdef BMBS_SOURCE_TEST
// reader self-test (no GPU): bmbs_reader_test <file> <window bytes> <threads> -- the text the driver's reader hands on, window by
// window, to stdout; what is left of a window behind its last complete record is carried into the next one as in the real pipeline
int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    Source s;
    const size_t cap = (size_t)atol(argv[2]);
    if (!s.open(argv[1], 0, ~(size_t)0, atoi(argv[3]))) { fprintf(stderr, "cannot open\n"); return 1; }
    Pool pool(3);
    std::vector<char> buf(cap + 64 + ((size_t)64 << 20));
    std::vector<uint32_t> counts;
    for (;;) {
        size_t n = 0; bool last = false;
        const size_t want = std::max(cap, s.carry.size() + 1024);
        if (!s.window(pool, buf.data(), want, n, last, counts)) { fprintf(stderr, "%s\n", s.err.c_str()); return 1; }
        size_t lines = 0;
        for (uint32_t c : counts) lines += c;
        const size_t nrec = lines / 4;
        if (nrec == 0 && !last) { fprintf(stderr, "record larger than the window\n"); return 1; }
        const size_t used = nrec ? after_kth_nl_blocks(buf.data(), n, counts, nrec * 4) : 0;
        fwrite(buf.data(), 1, used, stdout);
        s.consumed(buf.data(), n, used);
        if (last && used == n) break;
        if (last && nrec == 0) break;
    }
    s.close();
    return 0;
}
#else
int main(int argc, char** argv)
{
    bmbs_params P; bmbs_default_params(&P);
    std::string index, seq, seq1, seq2, out = "output", mapstats, build_fasta, index_folder;
    int device = 0, io_threads = 0, contexts = 4, parts = 1, reader_threads = 0;
    std::vector<int> devices;
    long batch = 500000;
    bool verbose = false, unmapped_out = false, pbat = false, bam = false, print_parts = false, print_plan = false;
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
        else if (a == "--loop-input") g_loop_input = std::max(1, atoi(val()));
        else if (a == "--out-parts") parts = atoi(val());
        else if (a == "--print-plan") print_parts = print_plan = true;           // ... and how the parts are worked off: devices, contexts, workers (no GPU needed: tests)
        else if (a == "--print-parts") print_parts = true;                   // the record ranges --out-parts would use, then exit (no GPU needed: tests)
        else if (a == "--reader-threads") reader_threads = atoi(val());     // pread threads per part (default: -t / (2 x parts))
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
        fprintf(stderr, "usage: bmbs_search --index <genome.fa> [--index_folder dir] [-t threads]\n       bmbs_search --search <index> (--seq r.fq | --seq1 a.fq --seq2 b.fq) [-o out.sam] [-e f] [--min n] [--max n] [--sensitive] [--pbat] [--unmapped_out] [--ambiguous_out] [--bam] [--mapstats f] [-t io_threads] [--out-parts n]\n");
        return 2;
    }
    if (batch < 1) batch = 1;
    if (io_threads <= 0) {
        // plain text needs few threads to read at memory speed; compressed input is inflated by them (csrc/pgz.h): more pay off
        const bool zin = Source::is_gz(seq.empty() ? seq1.c_str() : seq.c_str());
        const int hw = (int)std::thread::hardware_concurrency();
        // what the process may actually use: a container's CPU quota (cgroup v2 cpu.max "quota period") is often far below the hardware
        // threads it can see -- 16 cores' worth of 256 on the MI355X boxes -- and inflating on four times as many threads as that is slower
        int eff = hw;
        if (FILE* cf = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0;
            if (fscanf(cf, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) eff = (int)std::max<long long>(1, std::min<long long>(hw, (quota + period - 1) / period));
            fclose(cf);
        }
        io_threads = zin ? std::min(std::max(1, eff), 64) : std::min(hw, 32);
    }
    if (io_threads < 1) io_threads = 1;
    if (parts < 1) parts = 1;
    if (parts > 64) parts = 64;
    const double t_start = now();
    if (devices.empty()) devices.push_back(device);
    if (contexts < 1) contexts = 1;
    const bool pe = seq.empty();
    // --pbat: single-end reads are mapped as their reverse complement with mirrored qualities (inputReads_single_directly_pbat,
    // Process_Reads.cpp:986-1075; Schema.cpp:15102 need_reverse_quality = 1); paired-end input files swap roles
    // (exchange_two_reads, Process_Reads.cpp:1628, called from Bitmapper_main.cpp:169)
    if (pbat && pe) std::swap(seq1, seq2);
    const std::string& in1 = pe ? seq1 : seq;
    const bool gz_in = Source::is_gz(in1.c_str()) || (pe && Source::is_gz(seq2.c_str()));
    // a .gz stream cannot be entered in the middle: everything goes through part 0, the other part files stay empty
    const int live_parts = gz_in ? 1 : parts;
    // where the parts begin: byte offsets of record starts, the same record in both files of a pair (plain files; a .gz stream cannot
    // be entered in the middle)
    auto compute_cuts = [&](std::vector<size_t>& cut1, std::vector<size_t>& cut2) -> bool {
        cut1.assign((size_t)live_parts + 1, 0); cut2.assign((size_t)live_parts + 1, 0);
        if (gz_in) { cut1[1] = cut2[1] = ~(size_t)0; return true; }
        Pool pool(std::max(1, io_threads / 2) - 1);
        const int fd1 = ::open(in1.c_str(), O_RDONLY), fd2 = pe ? ::open(seq2.c_str(), O_RDONLY) : -1;
        struct stat sb1, sb2;
        if (fd1 < 0 || fstat(fd1, &sb1) || (pe && (fd2 < 0 || fstat(fd2, &sb2)))) return false;
        const size_t size1 = (size_t)sb1.st_size, size2 = pe ? (size_t)sb2.st_size : 0;
        cut1[(size_t)live_parts] = size1; cut2[(size_t)live_parts] = size2;
        for (int p = 1; p < live_parts; p++) {
            cut1[(size_t)p] = std::max(cut1[(size_t)p - 1], record_start_at(fd1, size1, (size_t)((double)size1 * p / live_parts)));
            if (pe) cut2[(size_t)p] = std::max(cut2[(size_t)p - 1], mate_boundary(pool, fd1, size1, cut1[(size_t)p], fd2, size2));
        }
        ::close(fd1); if (fd2 >= 0) ::close(fd2);
        return true;
    };
    if (print_parts) {
        std::vector<size_t> cut1, cut2;
        if (!compute_cuts(cut1, cut2)) { fprintf(stderr, "Cannot open the read file(s)\n"); return 1; }
        for (int p = 0; p <= live_parts; p++) printf("%d\t%zu\t%zu\n", p, cut1[(size_t)p], cut2[(size_t)p]);
        if (print_plan) {
            // record ranges of the input -> GPUs, index replicated, no exchange step, output concatenated in range order.
            // Here a range is a part; its batches go to whichever context is free (one worker thread per context, `contexts` contexts
            // on every listed device sharing that device's index copy) and come back in order through the part's own queue.
            printf("plan\tdevices\t%zu\tcontexts_per_device\t%d\tworkers\t%zu\tparts\t%d\tbatches_in_flight\t%zu\tbatch\t%ld\n",
                   devices.size(), contexts, devices.size() * (size_t)contexts, live_parts, (size_t)live_parts + devices.size() * (size_t)contexts + 2, batch);
            for (size_t d = 0; d < devices.size(); d++) printf("device\t%d\tindex_copy\t1\tcontexts\t%d\n", devices[d], contexts);
            printf("stats\tsum over %zu contexts (bmbs_stats_allreduce)\n", devices.size() * (size_t)contexts);
        }
        return 0;
    }
    // the drivers' contexts run one batch at a time each: one lane per context is enough (BMBS_LANES is only read by bmbs_create)
    setenv("BMBS_LANES", "1", 0);
    const int n_ctx = (int)devices.size() * contexts;
    const int n_batches = live_parts + n_ctx + 2;
    std::vector<Batch> batches((size_t)n_batches);
    for (auto& b : batches) { b.text1.kind = 1; b.text2.kind = 1; b.sam.kind = 2; }
    // bytes per record of the input, from its first records (plain text): sizes the page-locked windows, which cost ~0.2 ms per MB
    // to pin and are therefore allocated once, by several threads, while the index loads
    size_t est0 = 400;
    {
        char head[1 << 16];
        FILE* fp = fopen(in1.c_str(), "rb");
        const size_t got = fp ? fread(head, 1, sizeof head, fp) : 0;
        if (fp) fclose(fp);
        if (got > 2 && !((unsigned char)head[0] == 0x1f && (unsigned char)head[1] == 0x8b)) {
            size_t lines = 0, last = 0;
            for (size_t i = 0; i < got; i++) if (head[i] == '\n') { lines++; if (lines % 4 == 0) last = i + 1; }
            if (lines >= 4) est0 = std::max<size_t>(est0, last / (lines / 4) + 32);
        }
    }
    if (is_dir(index)) index += "/genome";           // Index.cpp:1048-1069
    bmbs_index_file* ixf = bmbs_index_file_load(index.c_str());
    if (!ixf) { fprintf(stderr, "Cannot open index %s.index*\n", index.c_str()); return 1; }
    bmbs_index_view view; bmbs_index_file_view(ixf, &view);
    std::vector<std::string> chrom_names;
    size_t max_chrom = 0;
    for (int i = 0; i < view.n_chrom; i++) { chrom_names.push_back(bmbs_index_file_chrom_name(ixf, i)); max_chrom = std::max(max_chrom, chrom_names.back().size()); }
    // SAM bytes a batch can need: QNAME + SEQ + QUAL come out of the text, the other columns are bounded per line
    const int L_est = (int)std::min<size_t>(1000, (est0 / 2) + (est0 / 4));
    auto sam_bound = [&](size_t text_bytes, size_t lines, int L) {
        return text_bytes + lines * (max_chrom + 5 * (size_t)std::max(8, (int)bmbs_max_cigar_ops(&P, std::max(1, std::min(1000, L)))) + 96) + 4096;
    };
    auto window_bytes = [&](size_t est) { return std::min<size_t>((size_t)batch * est + (1u << 16), (size_t)4000 << 20); };
    // BGZF input that is inflated on the device: the compressed bytes of a window, staged (block tables and a page-locked copy per
    // file).  Two of them: while a context opens one window, a helper thread stages the next
    struct Staged {
        Pinned buf[2]; std::vector<uint64_t> blk[2], out[2];
        size_t a[2] = {0, 0}, q[2] = {0, 0};
        bool foreign_any = false, ok = true; std::string err;
        bool last[2] = {false, false};               // the window ends its file (--loop-input: for the last time)
    } zst[2];
    for (auto& x : zst) { x.buf[0].kind = 1; x.buf[1].kind = 1; }
    std::thread prealloc([&] {
        const size_t want = window_bytes(est0) + 64;
        std::vector<std::thread> th;
        if (gz_in)
            for (auto& x : zst)
                for (int f = 0; f < (pe ? 2 : 1); f++) {
                    struct stat sb;
                    const size_t fsize = stat((f ? seq2 : in1).c_str(), &sb) == 0 ? (size_t)sb.st_size : 0;
                    Pinned* pb = &x.buf[f];
                    th.emplace_back([pb, fsize, want] { pb->need(std::min(fsize + 64, want / 2)); });
                }
        for (auto& b : batches)
            th.emplace_back([&, want] {
                Batch* bb = &b;
                bb->text1.need(want); if (pe) bb->text2.need(want);
                bb->sam.need(sam_bound(want * (pe ? 2 : 1), (size_t)batch * (pe ? 2 : 1), L_est));
            });
        for (auto& t : th) t.join();
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } prealloc_guard{prealloc};     // error returns below must not leave it running
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
    {
        std::vector<const char*> nm;
        for (const auto& s : chrom_names) nm.push_back(s.c_str());
        for (bmbs_ctx* c : ctxs) if (bmbs_sam_refs(c, nm.data(), (int32_t)nm.size())) { fprintf(stderr, "%s\n", bmbs_last_error(c)); return 1; }
    }
    // device memory for the work buffers of a batch, taken while the index loads instead of inside the first calls
    for (bmbs_ctx* c : ctxs) (void)bmbs_reserve(c, (uint64_t)batch * (pe ? 2 : 1) * (1200 + 3 * (uint64_t)std::max<size_t>(est0 / 2, 100)));
    // ---- the parts: record ranges of the input, one output file each
    std::vector<std::unique_ptr<Part>> P_(static_cast<size_t>(parts));
    for (int p = 0; p < parts; p++) { P_[(size_t)p].reset(new Part()); P_[(size_t)p]->id = p; }
    {
        std::vector<size_t> cut1, cut2;
        if (!compute_cuts(cut1, cut2)) { fprintf(stderr, "Cannot open the read file(s)\n"); return 1; }
        for (int p = 0; p < live_parts; p++) {
            Part& pt = *P_[(size_t)p];
            const int zt = std::max(1, io_threads / (pe ? 2 : 1));          // compressed input: inflate threads per file
            if (!pt.s1.open(in1.c_str(), cut1[(size_t)p], cut1[(size_t)p + 1], zt, devices[(size_t)p % devices.size()]) ||
                (pe && !pt.s2.open(seq2.c_str(), cut2[(size_t)p], cut2[(size_t)p + 1], zt, devices[(size_t)p % devices.size()]))) {
                fprintf(stderr, "Cannot open the read file(s)\n"); return 1;
            }
        }
    }
    for (int p = 0; p < parts; p++) {
        Part& pt = *P_[(size_t)p];
        std::string path = out;
        if (parts > 1) { char suf[32]; snprintf(suf, sizeof suf, ".part%03d", p); path += suf; }
        pt.ofd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (pt.ofd < 0) { fprintf(stderr, "Cannot open %s\n", path.c_str()); return 1; }
        {
            struct stat osb; pt.regular = fstat(pt.ofd, &osb) == 0 && S_ISREG(osb.st_mode);
            // blocks are reserved ahead only where that is a table entry (extent file systems): tmpfs answers fallocate by taking and
            // clearing the pages (4 GiB ahead of eight parts: slower than the writes, then ENOSPC), overlayfs refuses it
            struct statfs fsb;
            pt.can_alloc = pt.regular && fstatfs(pt.ofd, &fsb) == 0 &&
                           ((unsigned long)fsb.f_type == 0xEF53ul || (unsigned long)fsb.f_type == 0x58465342ul || (unsigned long)fsb.f_type == 0x9123683Eul);
        }
        if (p == 0) {
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
            if (pwrite(pt.ofd, h.data(), h.size(), 0) != (ssize_t)h.size()) { fprintf(stderr, "write error on %s\n", path.c_str()); return 1; }
            pt.out_off = h.size();
        }
    }
    prealloc.join();
    // the device touches every staging buffer once now (BMBS_NO_PREFAULT=1 skips it)
    if (!getenv("BMBS_NO_PREFAULT")) {
        std::vector<std::thread> th;
        for (size_t i = 0; i < batches.size(); i++)
            th.emplace_back([&, i] {
                Batch& b = batches[i];
                bmbs_ctx* c = ctxs[i % ctxs.size()];
                if (b.text1.p) (void)bmbs_host_prefault(c, b.text1.p, b.text1.cap, 1);
                if (b.text2.p) (void)bmbs_host_prefault(c, b.text2.p, b.text2.cap, 1);
                if (b.sam.p) (void)bmbs_host_prefault(c, b.sam.p, b.sam.cap, 2);
            });
        for (auto& t : th) t.join();
    }
    const int32_t flags = (pbat && !pe ? BMBS_TEXT_PBAT : 0) | (unmapped_out ? BMBS_TEXT_UNMAPPED : 0) | (bam ? BMBS_TEXT_BAM : 0);
    const double t_loaded = now();

    Chan<Batch*> free_q, gpu_q;
    for (auto& b : batches) free_q.put(&b);
    std::atomic<bool> failed(false);
    std::mutex err_mu;
    auto fail = [&](const std::string& why) { std::lock_guard<std::mutex> l(err_mu); if (!failed.exchange(true)) fprintf(stderr, "bmbs_search: %s\n", why.c_str()); };
    // the writers only pwrite (SAM text or finished BGZF blocks), every I/O thread reads
    const int r_threads = reader_threads > 0 ? reader_threads : std::max(1, io_threads / live_parts);

    // ---------------- stage R (one per part): text window + newline count -> how many whole records ----------------------------
    // BGZF input that stays on the device (bmbs_text_open_bgzf / bmbs_text_map_open): the reader hands the compressed blocks of a window
    // to a context, learns how many whole records they held and what is left over (the next window's prefix), and passes the context on
    // to a worker for the mapping.  The inflated text never crosses the link -- the host moves compressed bytes and tails only.
    Chan<bmbs_ctx*> ctx_pool;
    auto z_mode = [&](Part* pt) { return pt->s1.zdirect && (!pe || pt->s2.zdirect); };
    if (live_parts == 1 && z_mode(P_[0].get())) for (bmbs_ctx* c : ctxs) ctx_pool.put(c);
    // true: the input is finished (or failed); false: go on with the host-window reader (a member that is not a BGZF block turned up)
    auto reader_z = [&](Part* pt, Pool& pool, Pool& pool2) -> bool {
        size_t est = est0;
        Pinned tails[2]; tails[0].kind = 2; tails[1].kind = 2;
        // (what a window leaves over is a partial record, plus -- when the mates' records differ in size -- the surplus of one file, which the
        // next window's smaller block count evens out: never more than a window)
        // in practice a fraction of a record; the buffers grow when a call says so (page-locked memory is slow to get: ~0.1 ms per MB)
        size_t tail_cap = (size_t)4 << 20;
        if (const char* tv = getenv("BMBS_Z_TAIL")) { const long v = atol(tv); if (v >= 1) tail_cap = (size_t)v; }      // tests: the growth path
        if (!tails[0].need(tail_cap) || (pe && !tails[1].need(tail_cap))) { fail("cannot allocate page-locked staging memory"); return true; }
        Source* S[2] = {&pt->s1, &pt->s2};
        const int nf = pe ? 2 : 1;
        // (zst: the two staged windows -- while a context opens one (upload, inflate, index: the reader waits for the device), a helper
        // thread stages the next)
        Staged* st = zst;
        auto stage = [&](Staged& g, size_t room0, size_t room1) {
            g.foreign_any = false; g.ok = true;
            const size_t room[2] = {room0, room1};
            for (int f = 0; f < nf && g.ok; f++) {
                Source& s = *S[f];
                bool foreign = false;
                if (!s.next_blocks(room[f], g.a[f], g.q[f], foreign)) { g.ok = false; g.err = s.err; break; }
                if (foreign && g.q[f] == g.a[f]) { g.foreign_any = true; continue; }
                g.blk[f] = s.zblk; g.out[f] = s.zout;
                const size_t zbytes = g.q[f] - g.a[f];
                if (zbytes) {
                    if (!g.buf[f].need(zbytes + 64)) { g.ok = false; g.err = "cannot allocate page-locked staging memory"; break; }
                    Pool& pl = f ? pool2 : pool;
                    const int T = pl.size() * 2;
                    const size_t per = ((zbytes + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
                    char* dst = g.buf[f].p; const unsigned char* src = s.zmap + g.a[f];
                    pl.run(T, [&](int t) { const size_t x = std::min(zbytes, per * (size_t)t), y = std::min(zbytes, x + per); if (x < y) memcpy(dst + x, src + x, y - x); });
                }
                s.znext = g.q[f];
                g.last[f] = g.q[f] >= s.zsize && s.loops_left == 0;
                if (g.q[f] >= s.zsize && s.loops_left > 0) { s.znext = 0; s.loops_left--; }      // --loop-input: the file once more
            }
        };
        auto room_for = [&](size_t est_now, int f) {
            const size_t target = std::min<size_t>((size_t)batch * est_now + (1u << 16), (size_t)3500 << 20);
            return target > S[f]->carry.size() ? target - S[f]->carry.size() : (size_t)0;
        };
        int cur = 0;
        stage(st[0], room_for(est, 0), pe ? room_for(est, 1) : 0);
        std::thread ahead;
        auto join_ahead = [&] { if (ahead.joinable()) ahead.join(); };
        struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{ahead};
        for (;;) {
            const double tw0 = now();
            Batch* b = free_q.get();
            bmbs_ctx* ctx = ctx_pool.get();
            const double t0 = now();
            pt->t_wait_r += t0 - tw0;
            b->part = pt; b->seq = pt->next_seq++; b->n = 0; b->end = false; b->used1 = b->used2 = 0; b->sam_bytes = 0; b->open_ctx = nullptr;
            auto bail = [&](const std::string& why) { join_ahead(); fail(why); b->end = true; b->n = 0; ctx_pool.put(ctx); gpu_q.put(b); };
            if (failed) { join_ahead(); b->end = true; ctx_pool.put(ctx); gpu_q.put(b); return true; }
            join_ahead();
            Staged& g = st[cur];
            if (!g.ok) { bail(g.err); return true; }
            const size_t target = std::min<size_t>((size_t)batch * est + (1u << 16), (size_t)3500 << 20);
            bmbs_ztext z[2]; memset(z, 0, sizeof z);
            if (g.foreign_any) {
                // the rest of such a file goes through the host's stream inflater; blocks already taken for this window are given back
                for (int f = 0; f < nf; f++) {
                    Source& s = *S[f];
                    s.znext = g.a[f];
                    if (s.znext < s.zsize && !s.bgzf_block(s.zmap + s.znext, s.zsize - s.znext)) { std::lock_guard<std::mutex> l(s.m); s.zdirect = false; const size_t at = s.znext; s.znext = s.zsize; s.start_pgz(at, 0); }
                }
                pt->next_seq--;                                     // (the batch was not used)
                free_q.put(b);
                // every window opened so far has to be mapped before the workers go back to contexts of their own
                for (size_t i = 1; i < ctxs.size(); i++) (void)ctx_pool.get();
                return false;
            }
            size_t text_bytes = 0;
            for (int f = 0; f < nf; f++) {
                Source& s = *S[f];
                z[f].prefix = s.carry.empty() ? nullptr : s.carry.data(); z[f].prefix_bytes = s.carry.size();
                z[f].comp = g.buf[f].p; z[f].comp_bytes = g.q[f] - g.a[f]; z[f].blk_off = g.blk[f].data(); z[f].out_off = g.out[f].data(); z[f].n_blocks = (int64_t)g.blk[f].size() - 1;
                text_bytes += s.carry.size() + (size_t)g.out[f].back();
            }
            const bool last1 = g.last[0], last2 = pe && g.last[1];
            // the window behind this one is staged while the device works on this one (what this window will leave over is not known
            // yet: the last window's leftover stands in for it when the room is measured)
            {
                const size_t r0 = room_for(est, 0), r1 = pe ? room_for(est, 1) : 0;
                Staged* nx = &st[cur ^ 1];
                ahead = std::thread([&stage, nx, r0, r1] { stage(*nx, r0, r1); });
                cur ^= 1;
            }
            int64_t nrec = 0; uint64_t tb[2] = {0, 0};
            // every whole record of the window is taken (the window is what bounds a batch here: --batch records by the running estimate of a record's size)
            int rc = bmbs_text_open_bgzf(ctx, &z[0], pe ? &z[1] : nullptr, 4 * (int64_t)batch, last1 ? 1 : 0, last2 ? 1 : 0, &nrec, tails[0].p, tail_cap, &tb[0],
                                         pe ? tails[1].p : nullptr, &tb[1]);
            if (rc == BMBS_ENOMEM && std::max(tb[0], tb[1]) > tail_cap) {
                // more text behind the window's records than the tail buffers hold (the mates' records differ in size): once more with room
                tail_cap = (size_t)std::max(tb[0], tb[1]) + ((size_t)4 << 20);
                if (!tails[0].need(tail_cap) || (pe && !tails[1].need(tail_cap))) { bail("cannot allocate page-locked staging memory"); return true; }
                rc = bmbs_text_open_bgzf(ctx, &z[0], pe ? &z[1] : nullptr, 4 * (int64_t)batch, last1 ? 1 : 0, last2 ? 1 : 0, &nrec, tails[0].p, tail_cap, &tb[0],
                                         pe ? tails[1].p : nullptr, &tb[1]);
            }
            if (rc) { bail(bmbs_last_error(ctx)); return true; }
            for (int f = 0; f < nf; f++) S[f]->carry.assign(tails[f].p, tails[f].p + tb[f]);
            // the part's input ends with this batch when a file has nothing left behind it (PE: the shorter file decides)
            b->end = (last1 && tb[0] == 0) || (pe && last2 && tb[1] == 0);
            if (nrec == 0) {
                if (!b->end && (last1 || (pe && last2))) b->end = true;            // a trailing fragment that is not a whole record
                if (!b->end) { bail("FASTQ record larger than the " + std::to_string(target) + "-byte window"); return true; }
                ctx_pool.put(ctx); gpu_q.put(b);
                return true;
            }
            est = std::max<size_t>(64, text_bytes / (size_t)nf / (size_t)nrec + 16);
            b->n = nrec; b->open_ctx = ctx;
            const int Lg = (int)std::min<size_t>(1000, est / 2);
            if (!b->sam.need(sam_bound(text_bytes, (size_t)nrec * (pe ? 2 : 1), Lg))) { bail("cannot allocate page-locked staging memory"); return true; }
            pt->t_read += now() - t0;
            pt->records += nrec;
            const bool end = b->end;
            gpu_q.put(b);
            if (end) return true;
        }
    };
    auto reader_fn = [&](Part* pt) {
        Pool pool(r_threads - 1);
        Pool pool2(pe && pt->s2.gz ? std::max(1, r_threads / 2) - 1 : 0);          // second mate's window of compressed input
        if (live_parts == 1 && z_mode(pt) && reader_z(pt, pool, pool2)) return;
        size_t est = est0;
        for (;;) {
            const double tw0 = now();
            Batch* b = free_q.get();
            const double t0 = now();
            pt->t_wait_r += t0 - tw0;
            b->part = pt; b->seq = pt->next_seq++; b->n = 0; b->end = false; b->used1 = b->used2 = 0; b->sam_bytes = 0;
            auto bail = [&](const std::string& why) { fail(why); b->end = true; b->n = 0; gpu_q.put(b); };
            if (failed) { b->end = true; gpu_q.put(b); return; }
            // (.gz: what the previous window left over is copied in first and must fit whatever the new estimate says)
            const size_t want = std::max(window_bytes(est), std::max(pt->s1.carry.size(), pt->s2.carry.size()) + (1u << 16));
            if (!b->text1.need(want + 64) || (pe && !b->text2.need(want + 64))) { bail("cannot allocate page-locked staging memory"); return; }
            bool last1 = true, last2 = true;
            size_t n1 = 0, n2 = 0;
            if (pe && pt->s2.gz) {
                // compressed input: the two windows are assembled side by side (each waits for its own inflaters)
                bool ok2 = true;
                std::thread w2([&] { ok2 = pt->s2.window(pool2, b->text2.p, want, n2, last2, b->counts2); });
                const bool ok1 = pt->s1.window(pool, b->text1.p, want, n1, last1, b->counts1);
                w2.join();
                if (!ok1) { bail(pt->s1.err); return; }
                if (!ok2) { bail(pt->s2.err); return; }
            } else {
                if (!pt->s1.window(pool, b->text1.p, want, n1, last1, b->counts1)) { bail(pt->s1.err); return; }
                if (pe && !pt->s2.window(pool, b->text2.p, want, n2, last2, b->counts2)) { bail(pt->s2.err); return; }
            }
            size_t l1 = 0, l2 = 0;
            for (uint32_t c : b->counts1) l1 += c;
            for (uint32_t c : b->counts2) l2 += c;
            const long avail1 = (long)(l1 / 4), avail2 = pe ? (long)(l2 / 4) : 0;
            long nrec = pe ? std::min(avail1, avail2) : avail1;
            if (nrec > batch) nrec = batch;
            // the part's input ends with this batch when a file has no complete record left after it (PE: the shorter file decides)
            b->end = (last1 && avail1 == nrec) || (pe && last2 && avail2 == nrec);
            if (nrec == 0) {
                if (!b->end) { bail("FASTQ record larger than the " + std::to_string(want) + "-byte window"); return; }
                gpu_q.put(b);
                return;
            }
            b->used1 = after_kth_nl_blocks(b->text1.p, n1, b->counts1, (size_t)nrec * 4);
            pt->s1.consumed(b->text1.p, n1, b->used1);
            if (pe) { b->used2 = after_kth_nl_blocks(b->text2.p, n2, b->counts2, (size_t)nrec * 4); pt->s2.consumed(b->text2.p, n2, b->used2); }
            est = std::max<size_t>(64, std::max(b->used1, b->used2) / (size_t)nrec + 16);
            b->n = nrec;
            const int Lg = (int)std::min<size_t>(1000, est / 2);
            if (!b->sam.need(sam_bound(b->used1 + b->used2, (size_t)nrec * (pe ? 2 : 1), Lg))) { bail("cannot allocate page-locked staging memory"); return; }
            pt->t_read += now() - t0;
            pt->records += nrec;
            const bool end = b->end;
            gpu_q.put(b);
            if (end) return;
        }
    };

    // ---------------- stage W (one per part): the SAM text (or its BAM form) goes into the part's file, in order -----------------
    auto writer_fn = [&](Part* pt) {
        Pool wpool(std::max(1, std::min(8, io_threads / (2 * live_parts))) - 1);       // slices of a batch written side by side
        for (;;) {
            const double tw0 = now();
            Batch* b = pt->out_q.get();
            const double t0 = now();
            pt->t_wait_w += t0 - tw0;
            const bool end = b->end;
            if (b->n && !failed && b->sam_bytes) {
                const char* text = b->sam.p;
                const size_t len = (size_t)b->sam_bytes;
                // One buffered pwrite per batch extends the file under its inode lock at the speed of one memcpy (10 GB/s = 28 M SAM
                // records/s on the MI355X boxes).  With the blocks reserved ahead (fallocate in 4 GiB steps, cut back to the true size
                // at the end) several slices of a batch go into the page cache side by side: 17 GB/s (profiles/r02_write_probe.txt)
                if (pt->can_alloc && pt->out_off + len > pt->alloc_end) {
                    const size_t step = std::max<size_t>((size_t)4 << 30, 2 * len);
                    if (fallocate(pt->ofd, 0, (off_t)pt->alloc_end, (off_t)(pt->out_off + len + step - pt->alloc_end)) == 0) pt->alloc_end = pt->out_off + len + step;
                    else { pt->can_alloc = false; if (verbose && pt->alloc_end == 0) fprintf(stderr, "[bmbs_search] part %d: fallocate not available on the output (%s): one writer per batch\n", pt->id, strerror(errno)); }      // not a regular file (/dev/null, a pipe), or a file system without it
                }
                // (no fallocate -- overlayfs says ENODEV: four slices still extend a regular file a little faster than one, 12.8 against 10 GB/s)
                const int T = pt->can_alloc ? wpool.size() : (pt->regular ? std::min(4, wpool.size()) : 1);
                const size_t per = ((len + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
                wpool.run(T, [&](int t) {
                    size_t done = std::min(len, per * (size_t)t);
                    const size_t stop = std::min(len, done + per);
                    while (done < stop) {
                        const ssize_t w = pwrite(pt->ofd, text + done, stop - done, (off_t)(pt->out_off + done));
                        if (w <= 0) { fail(std::string("write error: ") + strerror(errno)); break; }
                        done += (size_t)w;
                    }
                });
                pt->out_off += len;
            }
            pt->t_write += now() - t0;
            free_q.put(b);
            if (end) return;
        }
    };

    // ---------------- stage G: one worker per context, one library call per batch ---------------------------------------
    double t_gpu = 0, t_wait_g = 0;
    std::mutex g_mu;
    auto g_worker = [&](bmbs_ctx* own_ctx) {
        for (;;) {
            bmbs_ctx* ctx = own_ctx;
            const double tw0 = now();
            Batch* b = gpu_q.get();
            if (!b) return;
            const double t0 = now();
            bmbs_ctx* octx = b->open_ctx;                                        // compressed input on the device: the batch's window is open on this context
            b->open_ctx = nullptr;
            if (octx) ctx = octx;
            if (!failed && b->n) {
                for (int attempt = 0; attempt < 2; attempt++) {
                    uint64_t bytes = 0; int64_t lines = 0;
                    const int rc = octx ? bmbs_text_map_open(ctx, flags, b->sam.p, b->sam.cap, &bytes, &lines)
                                 : pe ? bmbs_map_pe_text(ctx, b->text1.p, b->used1, b->text2.p, b->used2, b->n, flags, b->sam.p, b->sam.cap, &bytes, &lines)
                                      : bmbs_map_se_text(ctx, b->text1.p, b->used1, b->n, flags, b->sam.p, b->sam.cap, &bytes, &lines);
                    if (rc == BMBS_ENOMEM && bytes > b->sam.cap && attempt == 0 && b->sam.need((size_t)bytes + 64)) continue;   // reads longer than guessed
                    if (rc) fail(bmbs_last_error(ctx));
                    b->sam_bytes = rc ? 0 : bytes;
                    break;
                }
            }
            { std::lock_guard<std::mutex> l(g_mu); t_wait_g += t0 - tw0; t_gpu += now() - t0; }
            if (octx) ctx_pool.put(octx);
            b->part->out_q.put(b->seq, b);
        }
    };
    std::vector<std::thread> workers;
    for (bmbs_ctx* c : ctxs) workers.emplace_back(g_worker, c);
    for (int p = 0; p < live_parts; p++) { Part* pt = P_[(size_t)p].get(); pt->writer = std::thread(writer_fn, pt); pt->reader = std::thread(reader_fn, pt); }
    for (int p = 0; p < live_parts; p++) P_[(size_t)p]->reader.join();
    for (int p = 0; p < live_parts; p++) P_[(size_t)p]->writer.join();           // every batch has passed its writer: the workers are idle
    for (size_t i = 0; i < workers.size(); i++) gpu_q.put(nullptr);
    for (auto& t : workers) t.join();
    long total_records = 0;
    double t_read = 0, t_write = 0, t_format = 0, t_wait_r = 0, t_wait_w = 0;
    for (int p = 0; p < live_parts; p++) {
        const Part& pt = *P_[(size_t)p];
        total_records += pt.records; t_read += pt.t_read; t_write += pt.t_write; t_format += pt.t_format; t_wait_r += pt.t_wait_r; t_wait_w += pt.t_wait_w;
    }
    if (bam && !failed) {
        static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        Part& lastp = *P_[(size_t)parts - 1];
        if (pwrite(lastp.ofd, eof_block, 28, (off_t)lastp.out_off) != 28) failed = true;
        lastp.out_off += 28;
    }
    const double t_joined = now();
    for (int p = 0; p < parts; p++) { Part& pt = *P_[(size_t)p]; if (pt.alloc_end > pt.out_off && ftruncate(pt.ofd, (off_t)pt.out_off) != 0) failed = true; ::close(pt.ofd); if (p < live_parts) { pt.s1.close(); if (pe) pt.s2.close(); } }
    if (failed) { fprintf(stderr, "bmbs_search: failed\n"); return 1; }
    int64_t st[5];
    bmbs_stats_allreduce(ctxs.data(), (int)ctxs.size(), st);      // get_mapping_informations: the counters of every worker summed
    print_stats(stderr, st);
    if (!mapstats.empty()) { FILE* m = fopen(mapstats.c_str(), "w"); if (m) { print_stats(m, st); fclose(m); } }
    const double t_end = now();
    for (int p = 0; p < live_parts; p++) { P_[(size_t)p]->s1.release_device(); P_[(size_t)p]->s2.release_device(); }
    if (verbose)
        fprintf(stderr, "[bmbs_search] records %ld  load+attach %.3fs  mapping wall %.3fs  (pipeline %.3fs; stage busy, summed over %d part(s): read + newline count %.3fs, gpu calls %.3fs over %d context(s), host format %.3fs, write %.3fs)  %d I/O threads, batch %ld, %zu device(s) x %d context(s), %d output part(s)\n",
                total_records, t_loaded - t_start, t_end - t_loaded, t_joined - t_loaded, live_parts, t_read, t_gpu, n_ctx, t_format, t_write, io_threads, batch,
                n_owner, contexts, parts);
    if (verbose)
        fprintf(stderr, "[bmbs_search] stage idle (waiting for a batch, summed): readers %.3fs, gpu workers %.3fs, writers %.3fs\n", t_wait_r, t_wait_g, t_wait_w);
    if (verbose) {
        // what bounds the run: the link's busy time per direction (copies of the text calls, summed over the contexts: one copy per direction
        // and device at a time) and the workers' busy time against the mapping wall
        double up = 0, down = 0, in_calls = 0, calls = 0;
        for (bmbs_ctx* c : ctxs) { double t4[4]; if (bmbs_text_times(c, t4) == 0) { up += t4[0]; down += t4[1]; in_calls += t4[2]; calls += t4[3]; } }
        const double wall_s = t_end - t_loaded, nd = (double)std::max<size_t>(1, n_owner);
        fprintf(stderr, "[bmbs_search] busy fractions of the mapping wall: link up %.3f, link down %.3f (per device), gpu workers %.3f, readers %.3f, writers %.3f  (text calls %.0f, %.3fs inside them)\n",
                up / nd / wall_s, down / nd / wall_s, t_gpu / std::max(1, n_ctx) / wall_s, t_read / std::max(1, live_parts) / wall_s, t_write / std::max(1, live_parts) / wall_s, calls, in_calls);
    }
    const double t0 = now();
    for (auto& b : batches) { b.text1.release(); b.text2.release(); b.sam.release(); }
    const double t1 = now();
    for (size_t i = ctxs.size(); i-- > 0;) bmbs_destroy(ctxs[i]);     // the sharing contexts go before their owners
    const double t2 = now();
    bmbs_index_file_free(ixf);
    if (verbose) fprintf(stderr, "[bmbs_search] teardown: unpin %.3fs, destroy ctx %.3fs, free index %.3fs\n", t1 - t0, t2 - t1, now() - t2);
    return 0;
}
#endif
