// bitmapperbs_amd/csrc/index_build_gpu.hip -- bmbs_index_build_device: the index builder of index_io.cpp with every
// array-sized step on the GPU ("a deterministic GPU/CPU rebuild ... to drop the psascan dependency").
// Same six files, byte for byte, as bmbs_index_build and as the reference's --index on an N-free genome (tests).
//
// Why it exists: BASELINE configs[2..4] map against a GRCh38-size index.  The host builder needs 6 minutes for 6.2 G
// suffixes on 64 cores; the device sorts them in seconds, so bench.py can build that index on a fresh box.
//
// Suffix sort (text of n = 2G symbols over {G,T,A} = {0,1,2}, shorter suffix first -- the order psascan's output has once
// '$' is row 0, bwt.cpp:1135-1290):
//   1. text packed 2 bits per symbol (symbol+1, 0 beyond the end), 32 symbols per u64, most significant first: the first 32
//      symbols of suffix i are one funnel shift of two words and compare as an integer;
//   2. suffixes are dealt into the 12 buckets of their first two symbols (a 25 % bucket of GRCh38 = 1.55 G pairs of 16 bytes,
//      sorted with a double buffer in 50 GB); each bucket: (key, index) pairs -> LSD radix sort (rocPRIM) on the 60 low key
//      bits -> SA range written, rank = 1 + position of the first suffix with the same key (max-scan);
//   3. prefix doubling (Larsson-Sadakane) over the still-tied suffixes only: key = rank[suffix + h], two stable radix passes
//      (by key, then by group), new ranks, compaction; h = 32, 64, ... -- a random genome leaves ~10^4 ties after step 2,
//      a repeat-rich one log2(longest repeat / 32) rounds.
// BWT planes + interleaved counters, super-block table, SA_flag + sampled SA and the first/last rows of every 16-mer are
// then one pass each over the suffix array (wave ballots give the 64-row bit words).
#include "../../include/bmbs.h"
#include "index_io.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

using namespace bmbs_io;

namespace {

// A launch cannot carry 2^32 threads and GRCh38 has 6.2 G suffixes: every per-element kernel is a grid-stride loop over a
// capped grid.  IB_FOR_WAVE keeps whole waves in step (ballots inside): lanes beyond the end run the body with their index
// >= the bound.
#define IB_FOR(i, m) for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < (m); i += (u64)gridDim.x * blockDim.x)
#define IB_FOR_WAVE(i, m) for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; (i & ~63ull) < (m); i += (u64)gridDim.x * blockDim.x)

struct IbText {
    const u64* tx;   // packed text, n/32 + 3 words
    u64 n;
};

__device__ __forceinline__ u64 ib_key32(const IbText& t, u64 i)
{
    const u64 w = i >> 5;
    const unsigned s = 2u * (unsigned)(i & 31);
    const u64 w0 = t.tx[w], w1 = t.tx[w + 1];
    return s ? (w0 << s) | (w1 >> (64 - s)) : w0;
}
// symbol 0..2 at position i (i < n)
__device__ __forceinline__ unsigned ib_sym(const IbText& t, u64 i)
{
    return (unsigned)((t.tx[i >> 5] >> (62 - 2 * (i & 31))) & 3) - 1u;
}

// text = complement(fwd) C->T ++ reverse(fwd) C->T, recoded G0 T1 A2 (Index.cpp:645-682, bwt.cpp:1135); pac: A0 C1 G2 T3
__global__ void k_ib_text(const u8* __restrict__ pac, u64 G, u64 n_words, u64* __restrict__ tx)
{
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    u64 out = 0;
    for (int j = 0; j < 32; j++) {
        const u64 i = w * 32 + j;
        unsigned code = 0;
        if (i < G) {
            const unsigned v = (pac[i >> 2] >> (6 - 2 * (i & 3))) & 3;
            code = 1 + (v == 0 ? 1u : v == 1 ? 0u : v == 2 ? 1u : 2u);
        } else if (i < 2 * G) {
            const u64 q = 2 * G - 1 - i;
            const unsigned v = (pac[q >> 2] >> (6 - 2 * (q & 3))) & 3;
            code = 1 + (v == 2 ? 0u : v == 0 ? 2u : 1u);
        }
        out = (out << 2) | code;
    }
    tx[w] = out;
}

// suffixes per two-symbol bucket (16 codes)
__global__ void k_ib_hist(IbText t, u64 n_words, unsigned long long* __restrict__ hist)
{
    __shared__ unsigned int h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) {
        const u64 w0 = t.tx[w], w1 = t.tx[w + 1];
        unsigned int loc[16];
        for (int b = 0; b < 16; b++) loc[b] = 0;
        for (int j = 0; j < 32; j++) {
            const unsigned c = j < 31 ? (unsigned)((w0 >> (60 - 2 * j)) & 15) : (unsigned)(((w0 & 3) << 2) | (w1 >> 62));
            if (w * 32 + j < t.n) loc[c]++;
        }
        for (int b = 0; b < 16; b++) if (loc[b]) atomicAdd(&h[b], loc[b]);
    }
    __syncthreads();
    if (threadIdx.x < 16 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// (key, index) pairs of the suffixes of one bucket; order inside the bucket is arbitrary (the sort fixes it)
__global__ void k_ib_emit(IbText t, u64 n_words, unsigned bucket, unsigned long long* __restrict__ cursor,
                          u64* __restrict__ keys, u64* __restrict__ vals)
{
    __shared__ unsigned int cnt[256];
    __shared__ unsigned long long base;
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 w0 = 0, w1 = 0;
    unsigned mask = 0;
    if (w < n_words) {
        w0 = t.tx[w]; w1 = t.tx[w + 1];
        for (int j = 0; j < 32; j++) {
            const unsigned c = j < 31 ? (unsigned)((w0 >> (60 - 2 * j)) & 15) : (unsigned)(((w0 & 3) << 2) | (w1 >> 62));
            if (c == bucket && w * 32 + j < t.n) mask |= 1u << j;
        }
    }
    const unsigned mine = __popc(mask);
    cnt[threadIdx.x] = mine;
    __syncthreads();
    // exclusive scan of 256 counts (Hillis-Steele in LDS)
    for (int d = 1; d < 256; d <<= 1) {
        unsigned v = threadIdx.x >= (unsigned)d ? cnt[threadIdx.x - d] : 0;
        __syncthreads();
        cnt[threadIdx.x] += v;
        __syncthreads();
    }
    if (threadIdx.x == 255) base = cnt[255] ? atomicAdd(cursor, (unsigned long long)cnt[255]) : 0ull;
    __syncthreads();
    u64 o = base + cnt[threadIdx.x] - mine;
    while (mask) {
        const int j = __ffs(mask) - 1;
        mask &= mask - 1;
        const unsigned s = 2u * (unsigned)j;
        keys[o] = s ? (w0 << s) | (w1 >> (64 - s)) : w0;
        vals[o] = w * 32 + (u64)j;
        o++;
    }
}

// h[j] = j where a new key starts (else 0) -> max-scan gives the position of the first suffix with the same key
__global__ void k_ib_head(const u64* __restrict__ keys, u64 m, u64* __restrict__ h)
{
    IB_FOR(j, m) h[j] = (j == 0 || keys[j] != keys[j - 1]) ? j : 0;
}
__global__ void k_ib_place(const u64* __restrict__ vals, const u64* __restrict__ headpos, u64 m, u64 base,
                           u64* __restrict__ sa, u64* __restrict__ isa)
{
    IB_FOR(j, m) {
        const u64 idx = vals[j];
        sa[base + j] = idx;
        isa[idx] = base + headpos[j] + 1;
    }
}

// tied suffix = member of a group of two or more (rank = 1 + first position of its group)
__global__ void k_ib_tieflag(const u64* __restrict__ sa, const u64* __restrict__ isa, u64 n, u32* __restrict__ flag)
{
    IB_FOR(p, n) {
        const bool head = isa[sa[p]] == p + 1;
        const bool next_head = p + 1 >= n || isa[sa[p + 1]] == p + 2;
        flag[p] = (head && next_head) ? 0u : 1u;
    }
}
__global__ void k_ib_tielist(const u64* __restrict__ sa, const u64* __restrict__ isa, u64 n, const u32* __restrict__ flag,
                             const u64* __restrict__ off, u64* __restrict__ l_pos, u64* __restrict__ l_grp, u64* __restrict__ l_idx)
{
    IB_FOR(p, n) {
        if (!flag[p]) continue;
        const u64 o = off[p], idx = sa[p];
        l_pos[o] = p; l_idx[o] = idx; l_grp[o] = isa[idx] - 1;
    }
}
__global__ void k_ib_dkey(const u64* __restrict__ l_idx, const u64* __restrict__ isa, u64 n, u64 h, u64 m, u64* __restrict__ key,
                          u64* __restrict__ val)
{
    IB_FOR(j, m) {
        const u64 q = l_idx[j] + h;
        key[j] = q < n ? isa[q] : 0;
        if (val) val[j] = j;
    }
}
__global__ void k_ib_gather(const u64* __restrict__ src, const u64* __restrict__ perm, u64 m, u64* __restrict__ dst)
{
    IB_FOR(j, m) dst[j] = src[perm[j]];
}
__global__ void k_ib_dhead(const u64* __restrict__ l_grp, const u64* __restrict__ l_pos, const u64* __restrict__ key, u64 m,
                           u64* __restrict__ h)
{
    IB_FOR(j, m) h[j] = (j == 0 || l_grp[j] != l_grp[j - 1] || key[j] != key[j - 1]) ? l_pos[j] + 1 : 0;     // +1: position 0 is a valid head
}
// writes the re-ordered suffixes and their new ranks; flags the ones that are still tied
__global__ void k_ib_dplace(const u64* __restrict__ l_pos, const u64* __restrict__ n_idx, const u64* __restrict__ ngrp1, u64 m,
                            u64* __restrict__ sa, u64* __restrict__ isa, u32* __restrict__ flag)
{
    IB_FOR(j, m) {
        const u64 idx = n_idx[j], g1 = ngrp1[j];            // g1 = group head position + 1 = the rank
        sa[l_pos[j]] = idx;
        isa[idx] = g1;
        const bool head = g1 == l_pos[j] + 1;
        const bool next_head = j + 1 >= m || ngrp1[j + 1] == l_pos[j + 1] + 1;
        flag[j] = (head && next_head) ? 0u : 1u;
    }
}
__global__ void k_ib_dcompact(const u32* __restrict__ flag, const u64* __restrict__ off, u64 m, const u64* __restrict__ l_pos,
                              const u64* __restrict__ ngrp1, const u64* __restrict__ n_idx, u64* __restrict__ o_pos,
                              u64* __restrict__ o_grp, u64* __restrict__ o_idx)
{
    IB_FOR(j, m) {
        if (!flag[j]) continue;
        const u64 o = off[j];
        o_pos[o] = l_pos[j]; o_grp[o] = ngrp1[j] - 1; o_idx[o] = n_idx[j];
    }
}

// ---- outputs ------------------------------------------------------------------------------------------------------------
struct IbSa { const u64* sa; u64 n; };                   // row r: r == 0 -> n ('$'), else sa[r-1]
__device__ __forceinline__ u64 ib_row(const IbSa& s, u64 r) { return r == 0 ? s.n : s.sa[r - 1]; }

// BWT bit planes (bwt.cpp:1290-1500): stream position t = row minus the '$' row; 64 rows per wave -> two ballot words
__global__ void k_ib_bwt(IbText t, IbSa s, u64 shap, u64* __restrict__ bw)
{
    IB_FOR_WAVE(t0, s.n) {
        unsigned ch = 0;
        if (t0 < s.n) {
            const u64 row = t0 < shap ? t0 : t0 + 1;
            ch = ib_sym(t, ib_row(s, row) - 1);
        }
        const u64 b0 = __ballot(ch & 1), b1 = __ballot(ch >> 1);
        if ((threadIdx.x & 63) == 0 && t0 < s.n) {
            const u64 w = (t0 >> 7) * 5 + 1 + 2 * ((t0 & 127) >> 6);
            bw[w] = __brevll(b0);
            bw[w + 1] = __brevll(b1);
        }
    }
}
// in-block counters (relative to the super-block of 65 536 positions) + the super-block sums; one block = one super-block
__global__ void __launch_bounds__(1024) k_ib_occ(u64 n, u64* __restrict__ bw, u64* __restrict__ tot1, u64* __restrict__ tot2)
{
    __shared__ unsigned int s1[1024], s2[1024];
    const u64 wi = (u64)blockIdx.x * 1024 + threadIdx.x;          // 64-position word
    unsigned c1 = 0, c2 = 0;
    if (wi * 64 < n) {
        const u64 w = (wi >> 1) * 5 + 1 + 2 * (wi & 1);
        const u64 p0 = bw[w], p1 = bw[w + 1];
        c1 = __popcll(p0 & ~p1); c2 = __popcll(p1 & ~p0);
    }
    s1[threadIdx.x] = c1; s2[threadIdx.x] = c2;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        unsigned a = 0, b = 0;
        if (threadIdx.x >= (unsigned)d) { a = s1[threadIdx.x - d]; b = s2[threadIdx.x - d]; }
        __syncthreads();
        s1[threadIdx.x] += a; s2[threadIdx.x] += b;
        __syncthreads();
    }
    // counts in front of position t = 64 (wi + 1), written into the block that holds t, unless t starts a super-block
    const u64 tt = 64 * (wi + 1);
    if (tt <= n && threadIdx.x != 1023) {
        const u64 w0 = (tt >> 7) * 5, half = (tt & 127) >> 6;
        const unsigned long long v = ((unsigned long long)s1[threadIdx.x] << (48 - 32 * half)) | ((unsigned long long)s2[threadIdx.x] << (32 - 32 * half));
        atomicOr(reinterpret_cast<unsigned long long*>(bw + w0), v);
    }
    if (threadIdx.x == 1023) { tot1[blockIdx.x] = s1[1023]; tot2[blockIdx.x] = s2[1023]; }
}

// SA_flag bit words (bwt.cpp:1580-1800) and the number of samples per 64 rows
__global__ void k_ib_flag(IbSa s, u64 rows, u64 words, u64* __restrict__ fl, u32* __restrict__ cnt64)
{
    IB_FOR_WAVE(r, rows) {
        const bool f = r < rows && (ib_row(s, r) & 7) == 0;
        const u64 b = __ballot(f);
        if ((threadIdx.x & 63) == 0 && r < rows) {
            const u64 w = 5 * (r >> 8) + 1 + ((r & 255) >> 6);
            if (w < words + 8) fl[w] = __brevll(b);
            cnt64[r >> 6] = __popcll(b);
        }
    }
}
__global__ void k_ib_samples(IbText t, IbSa s, u64 rows, u64 words, const u64* __restrict__ off64, u64* __restrict__ fl,
                             u32* __restrict__ samp)
{
    IB_FOR_WAVE(r, rows) {
        u64 p = 0;
        bool f = false;
        if (r < rows) { p = ib_row(s, r); f = (p & 7) == 0; }
        const u64 b = __ballot(f);
        if (r < rows) {
            if ((r & 255) == 0 && 5 * (r >> 8) < words) fl[5 * (r >> 8)] = off64[r >> 6];      // samples in front of the block
            if (f) {
                const unsigned lane = threadIdx.x & 63;
                const u64 o = off64[r >> 6] + __popcll(b & ((1ull << lane) - 1));
                const u32 ch = p != 0 ? ib_sym(t, p - 1) : 1u;
                samp[o] = (ch << 30) | (u32)(p >> 3);
            }
        }
    }
}

// 16-mer of every row (bwt.cpp:1866-2010): base-3 number of the first 16 symbols, NONE for suffixes shorter than 16
__global__ void k_ib_key16(IbText t, IbSa s, u64 rows, u32* __restrict__ kr)
{
    IB_FOR(r, rows) {
        u32 k = 0xffffffffu;
        if (r >= 1) {
            const u64 p = s.sa[r - 1];
            if (p + 16 <= s.n) {
                const u64 key = ib_key32(t, p);
                u32 v = 0;
                for (int j = 0; j < 16; j++) v = v * 3 + ((u32)((key >> (62 - 2 * j)) & 3) - 1u);
                k = v;
            }
        }
        kr[r] = k;
    }
}
__global__ void k_ib_toprow(const u32* __restrict__ kr, u64 rows, u64* __restrict__ top, u64* __restrict__ bot)
{
    IB_FOR(r, rows) {
        if (r < 1) continue;
        const u32 k = kr[r];
        if (k == 0xffffffffu) continue;
        if (r == 1 || kr[r - 1] != k) top[k] = r;
        if (r + 1 == rows || kr[r + 1] != k) bot[k] = r + 1;
    }
}

struct Dev {
    std::vector<void*> all;
    std::string err;
    template <class T> T* get(u64 count)
    {
        void* p = nullptr;
        const size_t bytes = (size_t)(count ? count : 1) * sizeof(T);
        if (hipMalloc(&p, bytes) != hipSuccess) { err = "hipMalloc of " + std::to_string(bytes) + " bytes failed"; return nullptr; }
        all.push_back(p);
        return reinterpret_cast<T*>(p);
    }
    void drop(void* p)
    {
        if (!p) return;
        for (auto& q : all) if (q == p) { (void)hipFree(p); q = nullptr; }
    }
    ~Dev() { for (void* p : all) if (p) (void)hipFree(p); }
};

inline unsigned nb(u64 n, unsigned bs) { const u64 b = (n + bs - 1) / bs; return (unsigned)(b < (1u << 22) ? (b ? b : 1) : (1u << 22)); }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// out of device memory is BMBS_ENOMEM (a smaller GPU needs ~60 B per base), any other runtime failure BMBS_ESTATE; BMBS_ENODEV is kept
// for "there is no such device"
#define IB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "bmbs_index_build_device: %s: %s\n", #call, hipGetErrorString(e_)); return e_ == hipErrorOutOfMemory ? BMBS_ENOMEM : BMBS_ESTATE; } } while (0)
#define IB_PTR(p) do { if (!(p)) { fprintf(stderr, "bmbs_index_build_device: %s\n", D.err.c_str()); return BMBS_ENOMEM; } } while (0)

struct MaxOp { __device__ __host__ u64 operator()(const u64& a, const u64& b) const { return a > b ? a : b; } };

// stable LSD radix sort of (key, value) pairs on key bits [0, bits); temp storage grown on demand
struct Sorter {
    Dev& D;
    void* tmp = nullptr; size_t cap = 0;
    explicit Sorter(Dev& d) : D(d) {}
    int need(size_t bytes)
    {
        if (bytes <= cap) return 0;
        D.drop(tmp);
        tmp = D.get<u8>(bytes + bytes / 8);
        cap = tmp ? bytes + bytes / 8 : 0;
        return tmp ? 0 : BMBS_ENOMEM;
    }
    int pairs(const u64* kin, u64* kout, const u64* vin, u64* vout, u64 m, unsigned bits)
    {
        size_t bytes = 0;
        if (rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)m, 0u, bits, (hipStream_t)0) != hipSuccess) return BMBS_ENODEV;
        if (need(bytes)) return BMBS_ENOMEM;
        if (rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, (size_t)m, 0u, bits, (hipStream_t)0) != hipSuccess) return BMBS_ENODEV;
        return 0;
    }
    int maxscan(const u64* in, u64* out, u64 m)
    {
        size_t bytes = 0;
        if (rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)m, MaxOp(), (hipStream_t)0) != hipSuccess) return BMBS_ENODEV;
        if (need(bytes)) return BMBS_ENOMEM;
        if (rocprim::inclusive_scan(tmp, bytes, in, out, (size_t)m, MaxOp(), (hipStream_t)0) != hipSuccess) return BMBS_ENODEV;
        return 0;
    }
    int exscan(const u32* in, u64* out, u64 m)
    {
        size_t bytes = 0;
        if (rocprim::exclusive_scan(nullptr, bytes, in, out, (u64)0, (size_t)m, rocprim::plus<u64>(), (hipStream_t)0) != hipSuccess) return BMBS_ENODEV;
        if (need(bytes)) return BMBS_ENOMEM;
        if (rocprim::exclusive_scan(tmp, bytes, in, out, (u64)0, (size_t)m, rocprim::plus<u64>(), (hipStream_t)0) != hipSuccess) return BMBS_ENODEV;
        return 0;
    }
};

}  // namespace

extern "C" int bmbs_index_build_device(int device_id, const char* fasta, const char* prefix, int n_threads)
{
    if (!fasta || !prefix) return BMBS_EINVAL;
    if (n_threads < 1) n_threads = 1;
    const bool verbose = getenv("BMBS_BUILD_VERBOSE") != nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) {
        fprintf(stderr, "bmbs_index_build_device: no HIP device %d (the host builder is bmbs_index_build)\n", device_id);
        return BMBS_ENODEV;
    }
    // the caller's current device is put back on every way out
    struct DeviceGuard { int prev = -1; DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; } ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); } } device_guard;
    IB_HIP(hipSetDevice(device_id));
    // (an error another library in this process has already handled -- an allocator that frees its cache and tries again after
    // "out of memory" -- stays the thread's last error until it is read: not to be mistaken for one of this build's launches)
    (void)hipGetLastError();
    double t_last = now_s();
    auto lap = [&](const char* what) {
        if (!verbose) return;
        (void)hipDeviceSynchronize();
        const double t = now_s();
        fprintf(stderr, "[index_build_device] %-28s %.2f s\n", what, t - t_last);
        t_last = t;
    };
    Built B;
    u64 G = 0;
    {
        std::vector<char> gen;
        if (!slurp_fasta(fasta, B.chroms, gen)) return BMBS_EINVAL;
        G = gen.size();
        if (2 * G + 1 >= (1ULL << 33)) return BMBS_EINVAL;   // the sampled SA stores position / 8 in 30 bits (bwt.cpp:1793)
        lap("FASTA read");
        prepare_genome(B, gen, n_threads);
        lap("N replacement + pac");
    }
    const u64 n = 2 * G, rows = n + 1;
    B.sa_length = rows;
    Dev D;
    Sorter S(D);

    // ---- text ----
    const u64 n_words = (n + 31) / 32;
    u8* d_pac = D.get<u8>(B.pac.size() + 8); IB_PTR(d_pac);
    u64* d_tx = D.get<u64>(n_words + 3); IB_PTR(d_tx);
    IB_HIP(hipMemcpy(d_pac, B.pac.data(), B.pac.size(), hipMemcpyHostToDevice));
    IB_HIP(hipMemset(d_tx, 0, (n_words + 3) * 8));
    hipLaunchKernelGGL(k_ib_text, dim3(nb(n_words, 256)), dim3(256), 0, 0, d_pac, G, n_words, d_tx);
    IB_HIP(hipGetLastError());
    IB_HIP(hipDeviceSynchronize());
    D.drop(d_pac);
    IbText T = {d_tx, n};
    lap("text");

    // ---- buckets of the first two symbols ----
    unsigned long long* d_hist = D.get<unsigned long long>(17); IB_PTR(d_hist);
    IB_HIP(hipMemset(d_hist, 0, 17 * 8));
    hipLaunchKernelGGL(k_ib_hist, dim3(nb(n_words, 256)), dim3(256), 0, 0, T, n_words, d_hist);
    IB_HIP(hipGetLastError());
    unsigned long long hist[16];
    IB_HIP(hipMemcpy(hist, d_hist, 16 * 8, hipMemcpyDeviceToHost));
    u64 max_bucket = 1;
    for (int b = 0; b < 16; b++) max_bucket = std::max<u64>(max_bucket, hist[b]);

    u64* d_sa = D.get<u64>(n + 1); IB_PTR(d_sa);
    u64* d_isa = D.get<u64>(n + 1); IB_PTR(d_isa);
    {
        u64* ka = D.get<u64>(max_bucket); IB_PTR(ka);
        u64* kb = D.get<u64>(max_bucket); IB_PTR(kb);
        u64* va = D.get<u64>(max_bucket); IB_PTR(va);
        u64* vb = D.get<u64>(max_bucket); IB_PTR(vb);
        u64 base = 0;
        for (unsigned b = 0; b < 16; b++) {
            const u64 m = hist[b];
            if (!m) continue;
            IB_HIP(hipMemset(d_hist + 16, 0, 8));
            hipLaunchKernelGGL(k_ib_emit, dim3(nb(n_words, 256)), dim3(256), 0, 0, T, n_words, b, d_hist + 16, ka, va);
            int rc = S.pairs(ka, kb, va, vb, m, 60);
            if (rc) { fprintf(stderr, "bmbs_index_build_device: radix sort failed (%s)\n", D.err.c_str()); return rc; }
            hipLaunchKernelGGL(k_ib_head, dim3(nb(m, 256)), dim3(256), 0, 0, kb, m, ka);
            rc = S.maxscan(ka, va, m);
            if (rc) return rc;
            hipLaunchKernelGGL(k_ib_place, dim3(nb(m, 256)), dim3(256), 0, 0, vb, va, m, base, d_sa, d_isa);
            base += m;
        }
        IB_HIP(hipGetLastError());
        IB_HIP(hipDeviceSynchronize());
        if (base != n) { fprintf(stderr, "bmbs_index_build_device: bucket sizes do not add up\n"); return BMBS_ESTATE; }
        D.drop(ka); D.drop(kb); D.drop(va); D.drop(vb);
    }
    lap("bucket sort (32 symbols)");

    // ---- prefix doubling over the tied suffixes ----
    {
        u32* flag = D.get<u32>(n + 1); IB_PTR(flag);
        u64* off = D.get<u64>(n + 1); IB_PTR(off);
        hipLaunchKernelGGL(k_ib_tieflag, dim3(nb(n, 256)), dim3(256), 0, 0, d_sa, d_isa, n, flag);
        int rc = S.exscan(flag, off, n);
        if (rc) return rc;
        u64 m = 0;
        {
            u64 lo = 0; u32 lf = 0;
            IB_HIP(hipMemcpy(&lo, off + (n - 1), 8, hipMemcpyDeviceToHost));
            IB_HIP(hipMemcpy(&lf, flag + (n - 1), 4, hipMemcpyDeviceToHost));
            m = lo + lf;
        }
        if (verbose) fprintf(stderr, "[index_build_device] tied after 32 symbols: %llu\n", (unsigned long long)m);
        if (m) {
            u64* l_pos = D.get<u64>(m); IB_PTR(l_pos);
            u64* l_grp = D.get<u64>(m); IB_PTR(l_grp);
            u64* l_idx = D.get<u64>(m); IB_PTR(l_idx);
            hipLaunchKernelGGL(k_ib_tielist, dim3(nb(n, 256)), dim3(256), 0, 0, d_sa, d_isa, n, flag, off, l_pos, l_grp, l_idx);
            IB_HIP(hipDeviceSynchronize());
            D.drop(flag); D.drop(off);
            u64* ka = D.get<u64>(m); IB_PTR(ka);
            u64* kb = D.get<u64>(m); IB_PTR(kb);
            u64* va = D.get<u64>(m); IB_PTR(va);
            u64* vb = D.get<u64>(m); IB_PTR(vb);
            u64* o_pos = D.get<u64>(m); IB_PTR(o_pos);
            u64* o_grp = D.get<u64>(m); IB_PTR(o_grp);
            u64* o_idx = D.get<u64>(m); IB_PTR(o_idx);
            u32* tflag = D.get<u32>(m); IB_PTR(tflag);
            for (u64 h = 32; m; h *= 2) {
                const unsigned g = nb(m, 256);
                hipLaunchKernelGGL(k_ib_dkey, dim3(g), dim3(256), 0, 0, l_idx, d_isa, n, h, m, ka, va);
                rc = S.pairs(ka, kb, va, vb, m, 34);                       // by key ...
                if (rc) return rc;
                hipLaunchKernelGGL(k_ib_gather, dim3(g), dim3(256), 0, 0, l_grp, vb, m, ka);
                rc = S.pairs(ka, kb, vb, va, m, 34);                       // ... then, stably, by group: va = the permutation
                if (rc) return rc;
                hipLaunchKernelGGL(k_ib_gather, dim3(g), dim3(256), 0, 0, l_idx, va, m, o_idx);           // suffixes in their new order
                hipLaunchKernelGGL(k_ib_dkey, dim3(g), dim3(256), 0, 0, o_idx, d_isa, n, h, m, ka, (u64*)nullptr);
                hipLaunchKernelGGL(k_ib_dhead, dim3(g), dim3(256), 0, 0, l_grp, l_pos, ka, m, kb);
                rc = S.maxscan(kb, vb, m);                                 // vb = new group head position + 1 = new rank
                if (rc) return rc;
                hipLaunchKernelGGL(k_ib_dplace, dim3(g), dim3(256), 0, 0, l_pos, o_idx, vb, m, d_sa, d_isa, tflag);
                rc = S.exscan(tflag, ka, m);
                if (rc) return rc;
                u64 lo = 0; u32 lf = 0;
                IB_HIP(hipMemcpy(&lo, ka + (m - 1), 8, hipMemcpyDeviceToHost));
                IB_HIP(hipMemcpy(&lf, tflag + (m - 1), 4, hipMemcpyDeviceToHost));
                const u64 m2 = lo + lf;
                if (m2) {
                    hipLaunchKernelGGL(k_ib_dcompact, dim3(g), dim3(256), 0, 0, tflag, ka, m, l_pos, vb, o_idx, o_pos, o_grp, kb);
                    IB_HIP(hipMemcpyAsync(l_idx, kb, m2 * 8, hipMemcpyDeviceToDevice, 0));
                    std::swap(l_pos, o_pos); std::swap(l_grp, o_grp);
                }
                if (verbose) fprintf(stderr, "[index_build_device] h = %llu: %llu tied -> %llu\n", (unsigned long long)h, (unsigned long long)m, (unsigned long long)m2);
                m = m2;
            }
            IB_HIP(hipDeviceSynchronize());
            D.drop(l_pos); D.drop(l_grp); D.drop(l_idx); D.drop(ka); D.drop(kb); D.drop(va); D.drop(vb);
            D.drop(o_pos); D.drop(o_grp); D.drop(o_idx); D.drop(tflag);
        } else { D.drop(flag); D.drop(off); }
    }
    IB_HIP(hipGetLastError());
    lap("prefix doubling");

    // the row of the whole text ('$' precedes it)
    u64 shap = 0;
    IB_HIP(hipMemcpy(&shap, d_isa, 8, hipMemcpyDeviceToHost));
    B.shapline = shap;
    D.drop(d_isa);
    IbSa SA = {d_sa, n};

    // ---- BWT planes, counters, super-blocks ----
    {
        const u64 bw_words = 1 + 2 * (n / 64) + (n / 128) + 2;
        u64* d_bw = D.get<u64>(bw_words + 16); IB_PTR(d_bw);
        IB_HIP(hipMemset(d_bw, 0, (bw_words + 16) * 8));
        const u64 n_chunk = (n + 65535) / 65536;
        u64* d_t1 = D.get<u64>(n_chunk + 1); IB_PTR(d_t1);
        u64* d_t2 = D.get<u64>(n_chunk + 1); IB_PTR(d_t2);
        hipLaunchKernelGGL(k_ib_bwt, dim3(nb(n, 256)), dim3(256), 0, 0, T, SA, shap, d_bw);
        hipLaunchKernelGGL(k_ib_occ, dim3((unsigned)n_chunk), dim3(1024), 0, 0, n, d_bw, d_t1, d_t2);
        B.bwt.assign(bw_words, 0);
        IB_HIP(hipMemcpy(B.bwt.data(), d_bw, bw_words * 8, hipMemcpyDeviceToHost));
        std::vector<u64> t1(n_chunk), t2(n_chunk);
        IB_HIP(hipMemcpy(t1.data(), d_t1, n_chunk * 8, hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(t2.data(), d_t2, n_chunk * 8, hipMemcpyDeviceToHost));
        B.high_occ.assign(2 * (n / 65536 + 1), 0);
        u64 c1 = 0, c2 = 0;
        for (u64 c = 0; c < n_chunk; c++) {
            c1 += t1[c]; c2 += t2[c];
            if (c + 1 <= n / 65536) { B.high_occ[2 * (c + 1)] = c1; B.high_occ[2 * (c + 1) + 1] = c2; }
        }
        const u64 c0 = n - c1 - c2;
        B.nacgt[0] = 1; B.nacgt[1] = 1 + c0; B.nacgt[2] = B.nacgt[1] + c1; B.nacgt[3] = B.nacgt[2] + c2; B.nacgt[4] = B.nacgt[3];
        D.drop(d_bw); D.drop(d_t1); D.drop(d_t2);
    }
    IB_HIP(hipGetLastError());
    lap("BWT + Occ");

    // ---- SA_flag + sampled SA ----
    {
        const u64 q = rows / 256, rem = rows % 256;
        const u64 words = 1 + 5 * q + rem / 64 + ((rem % 64) ? 1 : 0) + 1;
        const u64 n_blk = (rows + 255) / 256, n64 = (rows + 63) / 64;
        u64* d_fl = D.get<u64>(words + 16); IB_PTR(d_fl);
        IB_HIP(hipMemset(d_fl, 0, (words + 16) * 8));
        u32* d_c64 = D.get<u32>(n64 + 1); IB_PTR(d_c64);
        u64* d_o64 = D.get<u64>(n64 + 1); IB_PTR(d_o64);
        IB_HIP(hipMemset(d_c64, 0, (n64 + 1) * 4));
        hipLaunchKernelGGL(k_ib_flag, dim3(nb(rows, 256)), dim3(256), 0, 0, SA, rows, words, d_fl, d_c64);
        int rc = S.exscan(d_c64, d_o64, n64 + 1);
        if (rc) return rc;
        u64 n_samp = 0;
        IB_HIP(hipMemcpy(&n_samp, d_o64 + n64, 8, hipMemcpyDeviceToHost));
        u32* d_samp = D.get<u32>(n_samp + 1); IB_PTR(d_samp);
        hipLaunchKernelGGL(k_ib_samples, dim3(nb(rows, 256)), dim3(256), 0, 0, T, SA, rows, words, d_o64, d_fl, d_samp);
        B.sa.assign(n_samp, 0);
        IB_HIP(hipMemcpy(B.sa.data(), d_samp, n_samp * 4, hipMemcpyDeviceToHost));
        std::vector<u64> fl(words + 8, 0);
        IB_HIP(hipMemcpy(fl.data(), d_fl, (words + 8) * 8, hipMemcpyDeviceToHost));
        if (rem == 0 && 5 * n_blk < words) fl[5 * n_blk] = n_samp;          // count word written after a complete last block
        fl.resize(words);
        B.sa_flag.swap(fl);
        D.drop(d_fl); D.drop(d_c64); D.drop(d_o64); D.drop(d_samp);
    }
    IB_HIP(hipGetLastError());
    lap("SA_flag + samples");

    // ---- 16-mer table ----
    {
        const u64 HS = 43046721ULL + 1;
        u32* d_kr = D.get<u32>(rows + 1); IB_PTR(d_kr);
        u64* d_top = D.get<u64>(HS); IB_PTR(d_top);
        u64* d_bot = D.get<u64>(HS); IB_PTR(d_bot);
        IB_HIP(hipMemset(d_top, 0, HS * 8));
        IB_HIP(hipMemset(d_bot, 0, HS * 8));
        hipLaunchKernelGGL(k_ib_key16, dim3(nb(rows, 256)), dim3(256), 0, 0, T, SA, rows, d_kr);
        hipLaunchKernelGGL(k_ib_toprow, dim3(nb(rows, 256)), dim3(256), 0, 0, d_kr, rows, d_top, d_bot);
        std::vector<u64> top(HS), bot(HS);
        IB_HIP(hipMemcpy(top.data(), d_top, HS * 8, hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(bot.data(), d_bot, HS * 8, hipMemcpyDeviceToHost));
        fill_hash_table(B, top.data(), bot.data());
    }
    IB_HIP(hipGetLastError());
    lap("16-mer table");
    const int rc = write_files(B, std::string(prefix) + ".index");
    lap("files written");
    return rc;
}
