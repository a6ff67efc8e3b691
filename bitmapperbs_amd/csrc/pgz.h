// bitmapperbs_amd/csrc/pgz.h -- block-parallel inflate of ordinary gzip files (one deflate stream per member, no block index).
//
// The reference reads .fastq.gz through zlib's gzread on its one reader thread (Process_Reads.cpp:1455-1514, kseq over gzFile); a
// deflate stream is sequential by construction (back-references reach 32 KiB into whatever came before), so one stream = one thread
// = ~0.7 GB/s of text, twenty times less than the mapping path takes.  This file inflates ONE stream on many threads:
//
//   A  (parallel)   the compressed file is cut every `span` bytes.  The task of a cut looks for the first position at or behind it
//                   that parses as the header of a dynamic-Huffman block (BFINAL = 0, BTYPE = 2, a complete code-length code, a
//                   valid literal/length and distance code with an end-of-block symbol) and decodes from there to the first block
//                   boundary at or behind the next cut.  It does not know the 32 KiB of text in front of its first block, so it
//                   decodes to 16-bit symbols: a byte, or "the byte at position w of the unknown window" (0x8000 | w).
//   B  (in order)   the start a task found has to be the boundary the task before it stopped at (otherwise its symbols are thrown
//                   away and that stretch is decoded again from the known boundary with the known window); the window behind a
//                   task = the window in front of it pushed through the task's last 32 Ki symbols.
//   C  (parallel)   symbols -> bytes with the task's window, CRC-32 of the bytes.
//   D  (in order)   member CRCs and lengths are checked against the gzip trailers (crc32_combine over the tasks' pieces) and the
//                   text is handed on in stream order.
//
// Everything zlib's inflate refuses is refused here (over-subscribed or incomplete codes, missing end-of-block code, distance too
// far back, reserved block type, stored-block length check, trailer mismatch), multi-member files and trailing garbage are handled
// as gzread does.  One case is caught later than zlib catches it: a match of a LATER member of a multi-member file that reaches back
// past its member's start is refused at once when the member began inside the stretch being decoded (OutBuf::mstart), and by the
// member's CRC-32 / length check when it began in an earlier stretch (the 32 KiB window handed from stretch to stretch does not
// carry where the member began).  The scheme is the published one of pugz / rapidgzip (two-pass decoding with window markers); no code of either.
#ifndef BMBS_PGZ_H
#define BMBS_PGZ_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace pgz {
typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;

static const u32 WIN = 32768;

// ---- CRC-32 (the gzip polynomial, reflected) by carry-less multiplication: 64 bytes folded per step, then 16, then a Barrett
// reduction (the scheme of Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ", with the constants every
// implementation of it shares).  zlib 1.2.11's table-driven crc32() runs at ~1 GB/s, a third of what a thread inflates: this one
// at 6+.  Checked once against zlib at start-up; a CPU without PCLMULQDQ (or a failed check) keeps zlib's.
#if defined(__x86_64__)
}  // namespace pgz
#include <immintrin.h>
namespace pgz {
__attribute__((target("pclmul,sse4.1")))
static inline uint32_t crc32_clmul_raw(uint32_t crc, const unsigned char* p, size_t len)      // len >= 64, len % 16 == 0; raw register in / out
{
    const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0000000163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x00000001F7011641ll, 0x00000001DB710641ll);
    const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
    __m128i x1 = _mm_loadu_si128((const __m128i*)(p)), x2 = _mm_loadu_si128((const __m128i*)(p + 16)),
            x3 = _mm_loadu_si128((const __m128i*)(p + 32)), x4 = _mm_loadu_si128((const __m128i*)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    p += 64; len -= 64;
    while (len >= 64) {
        const __m128i t1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), t2 = _mm_clmulepi64_si128(x2, k1k2, 0x00),
                      t3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), t4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, t1), _mm_loadu_si128((const __m128i*)(p)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, t2), _mm_loadu_si128((const __m128i*)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, t3), _mm_loadu_si128((const __m128i*)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, t4), _mm_loadu_si128((const __m128i*)(p + 48)));
        p += 64; len -= 64;
    }
    __m128i t;
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00); x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, t), x2);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00); x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, t), x3);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00); x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, t), x4);
    while (len >= 16) {
        t = _mm_clmulepi64_si128(x1, k3k4, 0x00); x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, t), _mm_loadu_si128((const __m128i*)p));
        p += 16; len -= 16;
    }
    __m128i y = _mm_clmulepi64_si128(x1, k3k4, 0x10);                          // 128 -> 64
    x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, y);
    y = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, mask32); x1 = _mm_clmulepi64_si128(x1, k5, 0x00); x1 = _mm_xor_si128(x1, y);     // 64 -> 32
    y = _mm_and_si128(x1, mask32); y = _mm_clmulepi64_si128(y, poly, 0x10); y = _mm_and_si128(y, mask32); y = _mm_clmulepi64_si128(y, poly, 0x00);
    x1 = _mm_xor_si128(x1, y);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
static inline bool crc_clmul_ok()
{
    static const bool ok = [] {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        unsigned char b[400];
        uint32_t x = 0x12345678u;
        for (int i = 0; i < 400; i++) { x = x * 1664525u + 1013904223u; b[i] = (unsigned char)(x >> 24); }
        for (size_t n : {64u, 80u, 144u, 400u})
            if (~crc32_clmul_raw(~0x9e3779b9u, b, n) != (uint32_t)crc32(0x9e3779b9u, b, (uInt)n)) return false;
        return true;
    }();
    return ok;
}
#else
static inline bool crc_clmul_ok() { return false; }
static inline uint32_t crc32_clmul_raw(uint32_t, const unsigned char*, size_t) { return 0; }
#endif
// zlib's crc32() semantics
static inline uint32_t crc32_fast(uint32_t crc, const unsigned char* p, size_t len)
{
    if (len >= 80 && crc_clmul_ok()) {
        const size_t body = len & ~(size_t)15;
        crc = ~crc32_clmul_raw(~crc, p, body);
        p += body; len -= body;
    }
    while (len) { const size_t q = len < ((size_t)1 << 30) ? len : ((size_t)1 << 30); crc = (uint32_t)crc32(crc, p, (uInt)q); p += q; len -= q; }
    return crc;
}

// ---- bit input: LSB-first, 64-bit buffer; bits past the end read as zero and are caught by `over()` -----------------------------
struct BitIn {
    const u8* base; const u8* next; const u8* end; u64 buf; int cnt;
    void init(const u8* b, size_t n, u64 bitpos)
    {
        base = b; end = b + n; next = b + (size_t)(bitpos >> 3); buf = 0; cnt = 0;
        if (next > end) next = end;
        refill();
        const int sh = (int)(bitpos & 7);
        buf >>= sh; cnt -= sh;
    }
    inline void refill()
    {
        if (next + 8 <= end) {
            u64 v; memcpy(&v, next, 8);
            buf |= v << cnt;
            next += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56) { buf |= (next < end ? (u64)*next : 0ull) << cnt; next++; cnt += 8; }      // `next` may pass `end`: zeros
        }
    }
    inline u32 peek(int n) const { return (u32)(buf & ((1ull << n) - 1)); }
    inline void drop(int n) { buf >>= n; cnt -= n; }
    inline u32 take(int n) { const u32 v = peek(n); drop(n); return v; }
    inline u64 pos() const { return (u64)(next - base) * 8 - (u64)cnt; }
    inline bool over() const { return pos() > (u64)(end - base) * 8; }
    void align() { drop(cnt & 7); }
};

// ---- Huffman tables: two levels, entry = value << 16 | extra-or-subtable-bits << 8 | kind << 5 | code bits to drop -------------
enum { K_BAD = 0, K_LIT = 1, K_BASE = 2, K_EOB = 3, K_SUB = 4 };      // K_LIT: `extra` literals (1 or 2) in the entry, value = first | second << 8
static const int LIT_ROOT = 11, DIST_ROOT = 8;
static const int LIT_CAP = 2048 + 2560, DIST_CAP = 256 + 2304;
static inline u32 mk(u32 value, u32 extra, u32 kind, u32 bits) { return value << 16 | extra << 8 | kind << 5 | bits; }
static inline u32 e_bits(u32 e) { return e & 31u; }
static inline u32 e_kind(u32 e) { return (e >> 5) & 7u; }
static inline u32 e_extra(u32 e) { return (e >> 8) & 31u; }
static inline u32 e_val(u32 e) { return e >> 16; }

static const u16 LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const u8 LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const u16 DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const u8 DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline u32 rev_bits(u32 v, int n) { u32 r = 0; for (int i = 0; i < n; i++) { r = (r << 1) | (v & 1); v >>= 1; } return r; }

// Kraft sum of a length set the way zlib's inflate_table judges it: -1 over-subscribed, > 0 incomplete, 0 complete
static inline int kraft_left(const u16 count[16])
{
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = left * 2 - (int)count[l]; if (left < 0) return -1; }
    return left;
}

// lens[0, n) -> table.  dist: distance code (30 symbols) else literal/length (286).  false = a set zlib refuses.
static bool build_table(const u8* lens, int n, bool dist, u32* tab, int cap)
{
    const int root = dist ? DIST_ROOT : LIT_ROOT;
    u16 count[16]; memset(count, 0, sizeof count);
    for (int s = 0; s < n; s++) count[lens[s]]++;
    int maxl = 15; while (maxl > 0 && !count[maxl]) maxl--;
    count[0] = 0;
    const int left = kraft_left(count);
    if (left < 0) return false;
    if (left > 0 && maxl != 1 && maxl != 0) return false;                  // incomplete: only "one code of one bit" / "no code at all" pass
    const int nroot = 1 << root;
    for (int i = 0; i < nroot; i++) tab[i] = 0;                             // K_BAD
    if (maxl == 0) return true;
    u16 first[17]; first[1] = 0;
    for (int l = 1; l <= 15; l++) first[l + 1] = (u16)((first[l] + count[l]) << 1);
    auto entry = [&](int s, u32 bits) -> u32 {
        if (dist) return s < 30 ? mk(DIST_BASE[s], DIST_EXTRA[s], K_BASE, bits) : mk(0, 0, K_BAD, bits);
        if (s < 256) return mk((u32)s, 1, K_LIT, bits);
        if (s == 256) return mk(0, 0, K_EOB, bits);
        return s < 286 ? mk(LEN_BASE[s - 257], LEN_EXTRA[s - 257], K_BASE, bits) : mk(0, 0, K_BAD, bits);
    };
    // codes of up to `root` bits fill the root table directly
    u16 nextc[16]; for (int l = 1; l <= 15; l++) nextc[l] = first[l];
    // pass 1: the longest code under every root prefix that needs a subtable
    u8 sub_bits[1 << LIT_ROOT];
    if (maxl > root) memset(sub_bits, 0, (size_t)nroot);
    u32 code_of[320];
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const u32 c = rev_bits(nextc[l]++, l);
        code_of[s] = c;
        if (l > root) { u8& b = sub_bits[c & (u32)(nroot - 1)]; if (l - root > b) b = (u8)(l - root); }
    }
    int used = nroot;
    if (maxl > root)
        for (int p = 0; p < nroot; p++)
            if (sub_bits[p]) {
                const int sz = 1 << sub_bits[p];
                if (used + sz > cap) return false;
                tab[p] = mk((u32)used, sub_bits[p], K_SUB, (u32)root);
                for (int i = 0; i < sz; i++) tab[used + i] = 0;
                used += sz;
            }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const u32 c = code_of[s];
        if (l <= root) {
            const u32 e = entry(s, (u32)l);
            for (u32 i = c; i < (u32)nroot; i += 1u << l) tab[i] = e;
        } else {
            const u32 pe = tab[c & (u32)(nroot - 1)];
            const u32 sb = e_extra(pe), so = e_val(pe);
            const u32 e = entry(s, (u32)(l - root));
            for (u32 i = c >> root; i < (1u << sb); i += 1u << (l - root)) tab[so + i] = e;
        }
    }
    if (!dist) {
        // two literals per look-up where both codes fit into the root index: base calls code in 2-3 bits, and a record is 150 of them
        u32 orig[1 << LIT_ROOT];
        memcpy(orig, tab, sizeof orig);
        for (int i = 0; i < nroot; i++) {
            const u32 e1 = orig[i];
            if (e_kind(e1) != K_LIT || (int)e_bits(e1) >= root) continue;
            const u32 e2 = orig[(u32)i >> e_bits(e1)];
            if (e_kind(e2) != K_LIT || e_bits(e1) + e_bits(e2) > (u32)root) continue;
            tab[i] = mk(e_val(e1) | e_val(e2) << 8, 2, K_LIT, e_bits(e1) + e_bits(e2));
        }
    }
    return true;
}

struct Tables { u32 lit[LIT_CAP]; u32 dist[DIST_CAP]; };

static const Tables& fixed_tables()
{
    static const Tables* t = [] {
        Tables* x = new Tables;
        u8 l[288]; for (int i = 0; i < 144; i++) l[i] = 8; for (int i = 144; i < 256; i++) l[i] = 9; for (int i = 256; i < 280; i++) l[i] = 7; for (int i = 280; i < 288; i++) l[i] = 8;
        build_table(l, 288, false, x->lit, LIT_CAP);
        u8 d[32]; for (int i = 0; i < 32; i++) d[i] = 5;
        build_table(d, 32, true, x->dist, DIST_CAP);
        return x;
    }();
    return *t;
}

static const u8 CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// the header of a dynamic block behind its three type bits: code lengths -> tables.  false = invalid.
static bool read_dynamic(BitIn& in, Tables& T)
{
    in.refill();
    const int hlit = (int)in.take(5) + 257, hdist = (int)in.take(5) + 1, hclen = (int)in.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    u8 cl[19]; memset(cl, 0, sizeof cl);
    in.refill();
    for (int i = 0; i < hclen; i++) { if (in.cnt < 3) in.refill(); cl[CL_ORDER[i]] = (u8)in.take(3); }
    u16 count[16]; memset(count, 0, sizeof count);
    for (int i = 0; i < 19; i++) count[cl[i]]++;
    count[0] = 0;
    if (kraft_left(count) != 0) return false;                                 // zlib: the code-length code has to be complete
    // 7-bit direct table of the code-length code
    u8 sym7[128], len7[128];
    {
        u16 first[9]; first[1] = 0;
        for (int l = 1; l <= 7; l++) first[l + 1] = (u16)((first[l] + count[l]) << 1);
        memset(len7, 0, sizeof len7);
        for (int s = 0; s < 19; s++) {
            const int l = cl[s];
            if (!l) continue;
            const u32 c = rev_bits(first[l]++, l);
            for (u32 i = c; i < 128; i += 1u << l) { sym7[i] = (u8)s; len7[i] = (u8)l; }
        }
    }
    u8 lens[286 + 30 + 138];
    int have = 0;
    const int total = hlit + hdist;
    while (have < total) {
        in.refill();
        const u32 x = in.peek(7);
        const int l = len7[x];
        if (!l) return false;
        in.drop(l);
        const int s = sym7[x];
        if (s < 16) { lens[have++] = (u8)s; continue; }
        int rep; u8 v = 0;
        if (s == 16) { if (!have) return false; v = lens[have - 1]; rep = 3 + (int)in.take(2); }
        else if (s == 17) rep = 3 + (int)in.take(3);
        else rep = 11 + (int)in.take(7);
        if (have + rep > total) return false;
        while (rep--) lens[have++] = v;
    }
    if (in.over()) return false;
    if (!lens[256]) return false;                                             // zlib: "invalid code -- missing end-of-block"
    return build_table(lens, hlit, false, T.lit, LIT_CAP) && build_table(lens + hlit, hdist, true, T.dist, DIST_CAP);
}

// ---- output of a decoding task: symbols of type S behind a 32 Ki prefix (the window in front of the task) ------------------------
template <class S> struct OutBuf {
    S* mem = nullptr; size_t cap = 0, n = 0;          // mem[0, WIN) = the window, mem[WIN + i] = output symbol i
    // how far back a match may reach: to output symbol `mstart` (where the member being decoded began, if it began in this buffer),
    // and `reach` symbols into the window in front of the buffer while no member has begun in it (zlib: "invalid distance too far back")
    size_t mstart = 0, reach = WIN;
    ~OutBuf() { free(mem); }
    OutBuf() {}
    OutBuf(const OutBuf&) = delete; OutBuf& operator=(const OutBuf&) = delete;
    bool reserve(size_t want)
    {
        if (want <= cap) return true;
        size_t c = std::max<size_t>(want, cap + cap / 2 + (1 << 16));
        S* m = (S*)realloc(mem, (WIN + c) * sizeof(S));
        if (!m) return false;
        mem = m; cap = c;
        return true;
    }
    S* out() { return mem + WIN; }
};

enum Status { ST_BOUNDARY = 0, ST_END = 1, ST_ERROR = 2 };
struct MemberEnd { u64 out_off; u32 crc; u32 isize; };
struct DecodeResult { Status st; u64 end_bit; const char* why; };

// a gzip member header at byte `at`: returns the byte offset of the deflate data, 0 when there is no header there
static size_t gzip_header(const u8* p, size_t n, size_t at)
{
    if (at + 10 > n || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8 || (p[at + 3] & 0xe0)) return 0;
    const int flg = p[at + 3];
    size_t q = at + 10;
    if (flg & 4) { if (q + 2 > n) return 0; const size_t xl = (size_t)p[q] | ((size_t)p[q + 1] << 8); q += 2 + xl; if (q > n) return 0; }
    if (flg & 8) { while (q < n && p[q]) q++; if (q >= n) return 0; q++; }
    if (flg & 16) { while (q < n && p[q]) q++; if (q >= n) return 0; q++; }
    if (flg & 2) { q += 2; if (q > n) return 0; }
    return q;
}

// Decodes blocks from `start_bit` (a block boundary) until the first block boundary at or behind `stop_bit`, the end of the last
// member, or an error.  ob.mem[0, WIN) has to hold the window (bytes, or markers 0x8000 | i when S = u16).  max_blocks: give up
// (as a boundary) after that many blocks -- used to try out a candidate start.
template <class S>
static DecodeResult decode_blocks(const u8* data, size_t size, u64 start_bit, u64 stop_bit, OutBuf<S>& ob, std::vector<MemberEnd>& ends,
                                  Tables& dyn, long max_blocks = -1)
{
    BitIn in; in.init(data, size, start_bit);
    DecodeResult r; r.st = ST_ERROR; r.end_bit = start_bit; r.why = "";
    long blocks = 0;
    for (;;) {
        const u64 here = in.pos();
        if (here >= stop_bit || (max_blocks >= 0 && blocks >= max_blocks)) { r.st = ST_BOUNDARY; r.end_bit = here; return r; }
        blocks++;
        in.refill();
        const u32 bfinal = in.take(1), btype = in.take(2);
        if (btype == 3) { r.why = "reserved block type"; return r; }
        if (btype == 0) {
            in.align();
            in.refill();
            const u32 len = in.take(16), nlen = in.take(16);
            if ((len ^ 0xffffu) != nlen) { r.why = "stored block length check"; return r; }
            const u64 byte = in.pos() >> 3;
            if (byte + len > size) { r.why = "stored block runs off the file"; return r; }
            if (!ob.reserve(ob.n + len + 512)) { r.why = "out of memory"; return r; }
            S* o = ob.out() + ob.n;
            for (u32 i = 0; i < len; i++) o[i] = (S)data[byte + i];
            ob.n += len;
            in.init(data, size, (byte + len) * 8);
        } else {
            const Tables* T;
            if (btype == 1) T = &fixed_tables();
            else { if (!read_dynamic(in, dyn)) { r.why = "invalid dynamic block header"; return r; } T = &dyn; }
            const u32* lt = T->lit; const u32* dt = T->dist;
            size_t n = ob.n;
            for (;;) {
                if (n + 600 > ob.cap) {
                    ob.n = n;
                    if (in.over()) { r.why = "deflate data runs off the file"; return r; }      // zeros behind the end decode for ever
                    if (!ob.reserve(n + (1 << 20))) { r.why = "out of memory"; return r; }
                }
                S* const o = ob.out();
                in.refill();
                u32 e = lt[in.buf & ((1u << LIT_ROOT) - 1)];
                if (e_kind(e) == K_SUB) { in.drop(LIT_ROOT); e = lt[e_val(e) + in.peek((int)e_extra(e))]; }
                if (e_kind(e) == K_LIT) {
                    // a run of literals from one refill: 56 bits held, the first code may take 15, each further root look-up 11 at most.
                    // Both bytes of an entry are stored whatever it holds (the second is overwritten when it was not one)
#define PGZ_LITERALS  { in.drop((int)e_bits(e)); o[n] = (S)(e_val(e) & 0xffu); o[n + 1] = (S)(e_val(e) >> 8); n += e_extra(e); }
                    PGZ_LITERALS
                    e = lt[in.buf & ((1u << LIT_ROOT) - 1)]; if (e_kind(e) != K_LIT) continue;
                    PGZ_LITERALS
                    e = lt[in.buf & ((1u << LIT_ROOT) - 1)]; if (e_kind(e) != K_LIT) continue;
                    PGZ_LITERALS
                    e = lt[in.buf & ((1u << LIT_ROOT) - 1)]; if (e_kind(e) != K_LIT) continue;
                    PGZ_LITERALS
#undef PGZ_LITERALS
                    continue;
                }
                in.drop((int)e_bits(e));
                if (e_kind(e) == K_BASE) {
                    const u32 len = e_val(e) + in.take((int)e_extra(e));
                    u32 d = dt[in.buf & ((1u << DIST_ROOT) - 1)];
                    if (e_kind(d) == K_SUB) { in.drop(DIST_ROOT); d = dt[e_val(d) + in.peek((int)e_extra(d))]; }
                    if (e_kind(d) != K_BASE) { r.why = "invalid distance code"; return r; }
                    in.drop((int)e_bits(d));
                    const u32 dist = e_val(d) + in.take((int)e_extra(d));
                    if ((u64)dist > n - ob.mstart + ob.reach) { r.why = "distance too far back"; return r; }
                    S* dst = o + n; const S* src = dst - dist;
                    const u32 step = 16 / sizeof(S);                         // elements per 16-byte move
                    if (dist >= step) {
                        // 16 bytes at a time; up to 15 bytes behind the match are written too (the buffer has the room, and they are
                        // overwritten by what follows)
                        for (u32 i = 0; i < len; i += step) memcpy(dst + i, src + i, 16);
                    } else if (dist == 1) {
                        const S v = src[0];
                        for (u32 i = 0; i < len; i++) dst[i] = v;
                    } else for (u32 i = 0; i < len; i++) dst[i] = src[i];
                    n += len;
                    continue;
                }
                if (e_kind(e) == K_EOB) break;
                r.why = "invalid literal/length code"; return r;
            }
            ob.n = n;
            if (in.over()) { r.why = "deflate data runs off the file"; return r; }
        }
        if (bfinal) {
            in.align();
            const u64 byte = in.pos() >> 3;
            if (byte + 8 > size) { r.why = "gzip trailer missing"; return r; }
            MemberEnd me; me.out_off = ob.n;
            me.crc = (u32)data[byte] | (u32)data[byte + 1] << 8 | (u32)data[byte + 2] << 16 | (u32)data[byte + 3] << 24;
            me.isize = (u32)data[byte + 4] | (u32)data[byte + 5] << 8 | (u32)data[byte + 6] << 16 | (u32)data[byte + 7] << 24;
            ends.push_back(me);
            const size_t nxt = gzip_header(data, size, (size_t)byte + 8);        // no further member: the rest is ignored, as gzread does
            if (!nxt) { r.st = ST_END; r.end_bit = (byte + 8) * 8; return r; }
            in.init(data, size, (u64)nxt * 8);
            ob.mstart = ob.n; ob.reach = 0;                                      // the next member's matches stay inside the member
        }
    }
}

// ---- looking for a block start -------------------------------------------------------------------------------------------------------
// the cheap part of the test: type bits, HLIT / HDIST in range, the code-length code complete
static inline bool header_plausible(const u8* data, size_t size, u64 bit)
{
    const size_t by = (size_t)(bit >> 3);
    if (by + 16 > size) return false;
    u64 v; memcpy(&v, data + by, 8);
    v >>= (bit & 7);
    if ((v & 7) != 4) return false;                                             // BFINAL 0, BTYPE 2
    if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) return false;
    const int hclen = (int)((v >> 13) & 15) + 4;
    // the 3-bit lengths start at bit + 17
    const u64 b2 = bit + 17;
    u64 w; memcpy(&w, data + (size_t)(b2 >> 3), 8);
    w >>= (b2 & 7);                                                             // >= 57 bits = 19 lengths
    int left = 1 << 7;                                                          // Kraft in units of 2^-7
    for (int i = 0; i < hclen; i++) { const int l = (int)(w & 7); w >>= 3; if (l) left -= 128 >> l; }
    return left == 0;
}

// ---- the engine ---------------------------------------------------------------------------------------------------------------------------
struct Options { size_t span = (size_t)1 << 20; int threads = 8; int max_ahead = 0; bool verify_crc = true; };

class Engine {
public:
    // sink(id, text): called from worker threads, ids 0, 1, 2 ... each exactly once, NOT necessarily in order (the caller orders them);
    // blocks while the caller wants back-pressure.  done(error or ""): called once after the last sink call returned.
    typedef std::function<void(long, std::vector<char>&&)> Sink;
    Engine(const u8* data, size_t size, size_t first_member_at, long first_id, const Options& o, Sink sink)
        : d_(data), n_(size), opt_(o), sink_(std::move(sink)), id0_(first_id)
    {
        if (opt_.threads < 1) opt_.threads = 1;
        if (opt_.max_ahead < 1) opt_.max_ahead = 2 * opt_.threads + 2;
        const size_t body = gzip_header(d_, n_, first_member_at);
        if (!body) { err_ = "not a gzip member header"; failed_ = true; finished_ = true; return; }
        body_bit_ = (u64)body * 8; known_bit_ = body_bit_;
        // cuts: task t covers compressed bytes [cut(t), cut(t + 1))
        const size_t len = n_ - body;
        n_tasks_ = (long)std::max<size_t>(1, (len + opt_.span - 1) / opt_.span);
        body_ = body;
        tasks_.resize((size_t)n_tasks_);
        window_.assign(WIN, 0);
    }
    ~Engine() { stop(); }
    void start() { if (finished_) return; for (int t = 0; t < opt_.threads; t++) th_.emplace_back([this] { work(); }); }
    void stop()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all(); cv_done_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
        th_.clear();
    }
    // blocks until everything has been handed to the sink (or an error / stop); returns the error text ("" = none)
    std::string wait()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_done_.wait(l, [this] { return finished_ || stop_; });
        return err_;
    }
    bool finished() { std::lock_guard<std::mutex> l(m_); return finished_; }
    std::string error() { std::lock_guard<std::mutex> l(m_); return err_; }
    // statistics for the tests: tasks whose speculative start was wrong or missing and had to be decoded again in order
    long redone() { std::lock_guard<std::mutex> l(m_); return redone_; }

private:
    struct Task {
        int state = 0;                       // 0 new, 1 A running, 2 A done, 3 B done (C may run), 4 C running, 5 C done
        bool found = false;
        u64 start_bit = 0, end_bit = 0;
        Status st = ST_BOUNDARY;
        const char* why = "";
        std::unique_ptr<OutBuf<u16>> sym;    // A's output
        std::unique_ptr<OutBuf<u8>> redo;    // B's output when the stretch was decoded again in order
        std::vector<MemberEnd> ends;
        std::vector<u8> window;              // the window in front of the task (set by B)
        bool skip = false;                   // B: the stretch was covered by the task before (nothing to emit)
        // C's output
        std::vector<char> text;
        std::vector<std::pair<u64, u32>> piece_crc;      // (length, crc) of the pieces between member ends
    };
    // symbol buffers are reused: a fresh 10 MB allocation per task is 2 500 page faults, and first-touch faults are what a
    // virtual machine is slowest at
    std::unique_ptr<OutBuf<u16>> sym_get()
    {
        std::lock_guard<std::mutex> l(pool_m_);
        if (pool_.empty()) return std::unique_ptr<OutBuf<u16>>(new OutBuf<u16>);
        std::unique_ptr<OutBuf<u16>> b = std::move(pool_.back()); pool_.pop_back();
        return b;
    }
    void sym_put(std::unique_ptr<OutBuf<u16>>& b)
    {
        if (!b) return;
        std::lock_guard<std::mutex> l(pool_m_);
        pool_.push_back(std::move(b));
    }
    u64 cut_bit(long t) const { return t >= n_tasks_ ? (u64)n_ * 8 + 64 : (t == 0 ? body_bit_ : (u64)(body_ + (size_t)t * opt_.span) * 8); }

    void fail(const std::string& e) { if (err_.empty()) err_ = e; failed_ = true; }

    // ---- A ----
    void run_a(long t, Task& k)
    {
        const u64 from = cut_bit(t), stop = cut_bit(t + 1);
        k.sym = sym_get();
        std::unique_ptr<Tables> dyn(new Tables);
        auto prime = [&] { if (!k.sym->reserve(opt_.span * 5)) return false; u16* w = k.sym->mem; for (u32 i = 0; i < WIN; i++) w[i] = (u16)(0x8000u | i); k.sym->n = 0; return true; };
        if (t == 0) {
            if (!prime()) { k.st = ST_ERROR; k.why = "out of memory"; return; }
            k.found = true; k.start_bit = from;
            const DecodeResult r = decode_blocks<u16>(d_, n_, from, stop, *k.sym, k.ends, *dyn);
            k.st = r.st; k.end_bit = r.end_bit; k.why = r.why;
            return;
        }
        const u64 last = std::min<u64>(stop, (u64)n_ * 8);
        for (u64 bit = from; bit < last; bit++) {
            if (!header_plausible(d_, n_, bit)) continue;
            {
                // the expensive part of the test: the whole header
                BitIn in; in.init(d_, n_, bit + 3);
                if (!read_dynamic(in, *dyn)) continue;
            }
            if (!prime()) { k.st = ST_ERROR; k.why = "out of memory"; return; }
            k.ends.clear();
            const DecodeResult r = decode_blocks<u16>(d_, n_, bit, stop, *k.sym, k.ends, *dyn);
            if (r.st == ST_ERROR) continue;                                      // not a block start after all
            k.found = true; k.start_bit = bit; k.st = r.st; k.end_bit = r.end_bit;
            return;
        }
        k.found = false; sym_put(k.sym);
    }

    // ---- B (one task at a time, in order; called without the lock) ----
    // prev_end: boundary the stream is known to have reached; st_prev: ST_END when the stream ended before this task
    void run_b(long t, Task& k)
    {
        k.window = window_;
        if (stream_ended_) { k.skip = true; sym_put(k.sym); return; }
        const u64 stop = cut_bit(t + 1);
        if (known_bit_ >= stop && t + 1 < n_tasks_) { k.skip = true; sym_put(k.sym); return; }     // the task before ran past this whole span
        if (k.found && k.start_bit == known_bit_ && k.st != ST_ERROR) {
            // the speculative decode stands: push the window through its last 32 Ki symbols
            const size_t n = k.sym->n; const u16* s = k.sym->out();
            advance_window(s, n);
            known_bit_ = k.end_bit; if (k.st == ST_END) stream_ended_ = true;
            return;
        }
        // decode this stretch again from the known boundary with the known window
        redone_count_++;
        k.redo.reset(new OutBuf<u8>);
        if (!k.redo->reserve(opt_.span * 5)) { k.st = ST_ERROR; k.why = "out of memory"; return; }
        memcpy(k.redo->mem, window_.data(), WIN);
        k.ends.clear();
        std::unique_ptr<Tables> dyn(new Tables);
        // up to the task's own start when that lies ahead (its symbols then follow), else to the boundary behind the next cut
        const bool join = k.found && k.st != ST_ERROR && k.start_bit > known_bit_ && k.start_bit < stop;
        std::vector<MemberEnd> ends;
        DecodeResult r;
        if (join) {
            // block by block, so that the task's start is recognised when the decode lands on it
            u64 at = known_bit_;
            for (;;) {
                r = decode_blocks<u8>(d_, n_, at, stop, *k.redo, ends, *dyn, 1);
                if (r.st != ST_BOUNDARY || r.end_bit >= k.start_bit || r.end_bit >= stop) break;
                at = r.end_bit;
            }
            if (r.st == ST_BOUNDARY && r.end_bit == k.start_bit) {
                // joined: the bytes decoded here, then the task's symbols resolved against the window as it is now
                std::vector<u8> w(WIN);
                window_after(k.redo->mem, k.redo->n, w);
                const size_t pre = k.redo->n;
                if (!k.redo->reserve(pre + k.sym->n + 64)) { k.st = ST_ERROR; k.why = "out of memory"; return; }
                u8* o = k.redo->out() + pre; const u16* s = k.sym->out();
                for (size_t i = 0; i < k.sym->n; i++) { const u16 v = s[i]; o[i] = v < 256 ? (u8)v : w[v & 0x7fffu]; }
                k.redo->n = pre + k.sym->n;
                for (auto& e : k.ends) { e.out_off += pre; ends.push_back(e); }
                k.ends = ends;
                sym_put(k.sym);
                window_after(k.redo->mem, k.redo->n, window_);
                known_bit_ = k.end_bit; if (k.st == ST_END) stream_ended_ = true;
                return;
            }
            if (r.st == ST_BOUNDARY && r.end_bit < stop) r = decode_blocks<u8>(d_, n_, r.end_bit, stop, *k.redo, ends, *dyn);
        } else r = decode_blocks<u8>(d_, n_, known_bit_, stop, *k.redo, ends, *dyn);
        sym_put(k.sym);
        k.ends = ends;
        k.st = r.st; k.why = r.why; k.end_bit = r.end_bit;
        if (r.st == ST_ERROR) return;
        window_after(k.redo->mem, k.redo->n, window_);
        known_bit_ = r.end_bit; if (r.st == ST_END) stream_ended_ = true;
    }
    // window_ <- the last 32 KiB of (window_ ++ resolved symbols)
    void advance_window(const u16* s, size_t n)
    {
        std::vector<u8> w(WIN);
        if (n >= WIN) { const u16* t = s + (n - WIN); for (u32 i = 0; i < WIN; i++) { const u16 v = t[i]; w[i] = v < 256 ? (u8)v : window_[v & 0x7fffu]; } }
        else {
            memcpy(w.data(), window_.data() + n, WIN - n);
            for (size_t i = 0; i < n; i++) { const u16 v = s[i]; w[WIN - n + i] = v < 256 ? (u8)v : window_[v & 0x7fffu]; }
        }
        window_.swap(w);
    }
    // w <- the last 32 KiB of mem[0, WIN + n) (mem = window ++ output, bytes)
    static void window_after(const u8* mem, size_t n, std::vector<u8>& w) { std::vector<u8> t(mem + n, mem + n + WIN); w.swap(t); }

    // ---- C ----
    void run_c(Task& k)
    {
        if (k.skip) return;
        size_t n;
        if (k.redo) { n = k.redo->n; k.text.assign((const char*)k.redo->out(), (const char*)k.redo->out() + n); k.redo.reset(); }
        else {
            n = k.sym->n; k.text.resize(n);
            const u16* s = k.sym->out(); const u8* w = k.window.data(); u8* o = (u8*)k.text.data();
            // A symbol is a byte or a marker 0x8000 | w; on FASTQ text about half of them stay markers to the end of a stretch (a
            // quality run copies the run before it, which copied the one before ...), so "v < 256 ? v : window[v & 0x7fff]" was an
            // unpredictable branch per symbol and 35-40 % of a thread's time.  One table serves both kinds -- the identity in its first
            // 256 entries, the window at 0x8000 -- and a group of sixteen symbols without a marker is one pack instruction.
            std::vector<u8> tab((size_t)65536);
            for (u32 v = 0; v < 256; v++) tab[v] = (u8)v;
            memcpy(tab.data() + 0x8000, w, WIN);
            const u8* T = tab.data();
            size_t i = 0;
#if defined(__x86_64__)
            const __m128i zero = _mm_setzero_si128();
            for (; i + 16 <= n; i += 16) {
                const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i)), b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i + 8));
                const __m128i hi = _mm_srli_epi16(_mm_or_si128(a, b), 8);
                if (_mm_movemask_epi8(_mm_cmpeq_epi8(hi, zero)) == 0xffff) _mm_storeu_si128(reinterpret_cast<__m128i*>(o + i), _mm_packus_epi16(a, b));
                else for (size_t j = i; j < i + 16; j++) o[j] = T[s[j]];
            }
#endif
            for (; i < n; i++) o[i] = T[s[i]];
            sym_put(k.sym);
        }
        k.window.clear(); k.window.shrink_to_fit();
        if (opt_.verify_crc) {
            u64 at = 0;
            for (size_t e = 0; e <= k.ends.size(); e++) {
                const u64 to = e < k.ends.size() ? k.ends[e].out_off : (u64)n;
                const u32 c = crc32_fast((u32)crc32(0L, Z_NULL, 0), (const unsigned char*)k.text.data() + at, (size_t)(to - at));
                k.piece_crc.push_back(std::make_pair(to - at, c));
                at = to;
            }
        }
    }

    // ---- D (in order, under the lock except for the sink call) ----
    bool check_members(Task& k)
    {
        if (!opt_.verify_crc || k.skip) return true;
        for (size_t e = 0; e < k.piece_crc.size(); e++) {
            run_crc_ = (u32)crc32_combine(run_crc_, k.piece_crc[e].second, (z_off_t)k.piece_crc[e].first);
            run_len_ += k.piece_crc[e].first;
            if (e < k.ends.size()) {
                if (run_crc_ != k.ends[e].crc || (u32)run_len_ != k.ends[e].isize) return false;
                run_crc_ = (u32)crc32(0L, Z_NULL, 0); run_len_ = 0;
            }
        }
        return true;
    }

    void work()
    {
        std::unique_lock<std::mutex> l(m_);
        for (;;) {
            if (stop_ || finished_) return;
            // D: hand finished tasks on in order (one thread at a time)
            if (!emitting_ && next_d_ < n_tasks_ && tasks_[(size_t)next_d_].state == 5) {
                emitting_ = true;
                const long t = next_d_;
                Task& k = tasks_[(size_t)t];
                bool ok = !failed_;
                if (ok && !check_members(k)) { fail("gzip trailer does not match the inflated data (corrupt .gz input?)"); ok = false; }
                std::vector<char> text; text.swap(k.text);
                const bool has = ok && !text.empty();
                const long id = id0_ + emitted_;                                  // ids are dense over the non-empty pieces
                if (has) emitted_++;
                l.unlock();
                if (has) sink_(id, std::move(text));
                l.lock();
                emitting_ = false;
                next_d_++;
                if (next_d_ == n_tasks_ || failed_) {
                    if (!failed_ && !stream_ended_) fail("the deflate stream ends before its last block (truncated .gz input?)");
                    finished_ = true;
                }
                // (a thread that changes the state goes round the loop itself and takes what has become possible: the others are woken
                // one at a time, for the second piece of work an event may have made -- notify_all here woke every thread for every
                // step of every task, and 64 threads queueing for this mutex were slower than 8)
                if (finished_) { cv_.notify_all(); cv_done_.notify_all(); } else cv_.notify_one();
                continue;
            }
            // B: the next task in order whose A has finished
            if (!sequencing_ && next_b_ < n_tasks_ && tasks_[(size_t)next_b_].state == 2) {
                sequencing_ = true;
                const long t = next_b_;
                Task& k = tasks_[(size_t)t];
                l.unlock();
                if (!failed_) run_b(t, k);
                l.lock();
                if (k.st == ST_ERROR && !k.skip && !failed_) fail(std::string("corrupt .gz input: ") + (k.why && *k.why ? k.why : "invalid deflate data"));
                redone_ = redone_count_;
                k.state = 3; next_b_++; sequencing_ = false;
                cv_.notify_one();
                continue;
            }
            // C: any task whose window is known
            {
                long pick = -1;
                for (long t = next_d_; t < next_b_; t++) if (tasks_[(size_t)t].state == 3) { pick = t; break; }
                if (pick >= 0) {
                    Task& k = tasks_[(size_t)pick];
                    k.state = 4;
                    l.unlock();
                    if (!failed_) run_c(k); else { sym_put(k.sym); k.redo.reset(); }
                    l.lock();
                    k.state = 5;
                    continue;
                }
            }
            // A: the next span, while the text inflated ahead of its turn stays bounded
            if (next_a_ < n_tasks_ && next_a_ - next_d_ < opt_.max_ahead && !failed_) {
                const long t = next_a_++;
                Task& k = tasks_[(size_t)t];
                k.state = 1;
                l.unlock();
                run_a(t, k);
                l.lock();
                k.state = 2;
                continue;
            }
            if (failed_ && next_a_ < n_tasks_) {
                // nothing new is started after an error: let the tasks in flight drain through B / C / D as skips
                for (long t = next_a_; t < n_tasks_; t++) { tasks_[(size_t)t].state = 2; tasks_[(size_t)t].found = false; }
                next_a_ = n_tasks_;
                cv_.notify_all();
                continue;
            }
            cv_.wait(l);
        }
    }

    const u8* d_; size_t n_; Options opt_; Sink sink_; long id0_;
    size_t body_ = 0; u64 body_bit_ = 0; long n_tasks_ = 0;
    std::vector<Task> tasks_;
    std::mutex m_; std::condition_variable cv_, cv_done_;
    std::mutex pool_m_; std::vector<std::unique_ptr<OutBuf<u16>>> pool_;
    std::vector<std::thread> th_;
    long next_a_ = 0, next_b_ = 0, next_d_ = 0, emitted_ = 0, redone_ = 0, redone_count_ = 0;
    bool sequencing_ = false, emitting_ = false, stop_ = false, finished_ = false;
    std::atomic<bool> failed_{false};
    std::string err_;
    // B's state (touched by the one sequencing thread only)
    std::vector<u8> window_; u64 known_bit_ = 0; bool stream_ended_ = false;
    // D's state
    u32 run_crc_ = 0; u64 run_len_ = 0;
};
}  // namespace pgz
#endif
