// tools/vote_kbench.hip -- k_vote_long ALONE (the single-end long-list vote kernel: locate, site sort, run-length votes, std::sort's
// vote order, write-out), built from the library's own k_vote.hip in seconds: synthetic reads with lists of ~660 candidates over a
// random suffix array, the kernel timed with HIP events, the first lists checked against the same steps on the host (std::sort).
// build (here):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/vote_kbench tools/vote_kbench.hip
//                (-DKVOTE_FILE='"../tools/_ab/k_vote_r05.hip"' -DVB_OLD: the round-5 kernel, for same-box comparisons)
// run (GPU box): ./tools/vote_kbench [lists=60000] [mean candidates=660] [log2 of the suffix-array rows=26] [blocks=8192]
//                (-DVB_CAP=2048 -DVB_LO=1024 -DVB_BLOCK=256: the next size class)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../bitmapperbs_amd/csrc/bmbs_dev.h"
#include "../bitmapperbs_amd/csrc/bmbs_sort.h"
#define DEVI __device__ __forceinline__
#define SHARD(p) ((p) + (size_t)(blockIdx.x & (BMBS_SHARDS - 1)) * BMBS_SHARD_WORDS)
#define CNT_CAND_MID 9
#define CNT_CAND_LONG 10
#define CNT_CAND_BIG 11
#define CNT_LISTS_LONG 12
DEVI void wave_count_add(unsigned long long* counters, int word, u32 v)
{
    if (!counters) return;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&SHARD(counters)[word], (unsigned long long)v);
}
DEVI u64 sa_at(const DevIndex& ix, u64 row) { return ix.sa64 ? ix.sa64[row] : (u64)ix.sa[row]; }
#ifndef KVOTE_FILE
#define KVOTE_FILE "../bitmapperbs_amd/csrc/k_vote.hip"
#endif
#include KVOTE_FILE
#ifndef VB_BLOCK
#define VB_BLOCK 128
#endif
#ifndef VB_CAP
#define VB_CAP 1024
#define VB_LO 256
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 60000;
    const int mean = argc > 2 ? atoi(argv[2]) : 660;
    const unsigned grid = argc > 4 ? (unsigned)atoi(argv[4]) : 8192u;          // the library launches 32768 waves of the wave form, 8192 / 4096 / 2048 blocks of the others
    CK(hipSetDevice(0));
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    // a "suffix array" of 2^26 random text positions in a 6.2 G text (wide: 64-bit entries, as at GRCh38 size)
    const u64 rows = 1ull << (argc > 3 ? atoi(argv[3]) : 26), total = 6200000000ull;       // (2^26 rows = 512 MB; 2^31: 16 GB, beyond the caches)
    std::vector<u64> sa(rows);
    for (u64 i = 0; i < rows; i++) sa[i] = 1000 + rnd() % (total - 2000);
    std::vector<SeedRec> seeds((size_t)n * BMBS_MAX_SEEDS);
    std::vector<u8> verdict(n, 3), n_seeds(n);
    std::vector<u32> n_cand(n), list(n);
    std::vector<u64> cand_off(n + 1, 0);
    for (long r = 0; r < n; r++) {
        const int ns = 8 + (int)(rnd() % 17);
        int want = mean / 2 + (int)(rnd() % mean);                 // mean/2 .. 3 mean/2
        if (want > VB_CAP - 24) want = VB_CAP - 24;
        if (want < VB_LO + 1) want = VB_LO + 1;
        n_seeds[r] = (u8)ns;
        int left = want;
        u32 tot = 0;
        const u64 row0 = rnd() % (rows - 2000);
        for (int q = 0; q < ns; q++) {
            SeedRec& sr = seeds[(size_t)r * BMBS_MAX_SEEDS + q];
            const int h = q == ns - 1 ? left : std::max(1, std::min(left - (ns - 1 - q), (int)(rnd() % (2 * want / ns + 1))));
            // a seed in fifty walks the rows of the seed before it with the same adjustment: its sites coincide with that seed's (votes of 2)
            const bool again = q > 0 && rnd() % 50 == 0;
            sr.sp = again ? seeds[(size_t)r * BMBS_MAX_SEEDS + q - 1].sp : (rows > (1ull << 27) ? rnd() % (rows - 2000) : row0 + rnd() % 1000);
            sr.hits = (u32)(again ? std::min<u32>((u32)h, seeds[(size_t)r * BMBS_MAX_SEEDS + q - 1].hits) : h);
            sr.len = 30; sr.off = again ? seeds[(size_t)r * BMBS_MAX_SEEDS + q - 1].off : (u16)(rnd() % 100);
            left -= (int)sr.hits; tot += sr.hits;
            if (left <= 0) { n_seeds[r] = (u8)(q + 1); break; }
        }
        n_cand[r] = tot; cand_off[r + 1] = cand_off[r] + tot; list[r] = (u32)r;
    }
    const u64 slots = cand_off[n];
    u64* d_sa; SeedRec* d_seeds; u8 *d_verdict, *d_nseeds; u32 *d_ncand, *d_list, *d_nvotes, *d_slot; u64 *d_off, *d_cand, *d_count; bmbs_vote* d_votes;
    unsigned long long* d_cnt;
    CK(hipMalloc(&d_sa, rows * 8)); CK(hipMemcpy(d_sa, sa.data(), rows * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_seeds, seeds.size() * sizeof(SeedRec))); CK(hipMemcpy(d_seeds, seeds.data(), seeds.size() * sizeof(SeedRec), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_verdict, n)); CK(hipMemcpy(d_verdict, verdict.data(), n, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_nseeds, n)); CK(hipMemcpy(d_nseeds, n_seeds.data(), n, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ncand, n * 4)); CK(hipMemcpy(d_ncand, n_cand.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_list, n * 4)); CK(hipMemcpy(d_list, list.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_off, (n + 1) * 8)); CK(hipMemcpy(d_off, cand_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_nvotes, n * 4)); CK(hipMalloc(&d_slot, slots * 4)); CK(hipMalloc(&d_cand, slots * 8)); CK(hipMalloc(&d_votes, slots * sizeof(bmbs_vote)));
    CK(hipMalloc(&d_count, 8)); { u64 c = (u64)n; CK(hipMemcpy(d_count, &c, 8, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&d_cnt, BMBS_SHARDS * BMBS_SHARD_WORDS * 8)); CK(hipMemset(d_cnt, 0, BMBS_SHARDS * BMBS_SHARD_WORDS * 8));
    DevIndex ix; memset(&ix, 0, sizeof ix);
    ix.sa64 = d_sa; ix.total = total; ix.G = total / 2;
    ReadGeom gm = {nullptr, nullptr, 150, 12};
    ReadState st; memset(&st, 0, sizeof st);
    st.verdict = d_verdict; st.n_seeds = d_nseeds; st.seeds = d_seeds; st.n_cand = d_ncand; st.cand_off = d_off; st.n_votes = d_nvotes;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&] {
#ifdef VB_OLD
        hipLaunchKernelGGL((k_vote_long<VB_CAP, VB_BLOCK, VB_LO>), dim3(grid), dim3(VB_BLOCK), 0, 0, ix, gm, st, d_count, d_list, d_cand, d_votes, d_slot, (u32*)nullptr, (unsigned long long*)nullptr);
#else
        hipLaunchKernelGGL((k_vote_long<VB_CAP, VB_BLOCK, VB_LO>), dim3(grid), dim3(VB_BLOCK), 0, 0, ix, gm, st, d_count, d_list, d_cand, d_votes, d_slot, (u32*)nullptr, (unsigned long long*)nullptr, d_cnt);
#endif
    };
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    const int reps = 5;
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms); sum += ms;
    }
    // check the first lists against the same steps on the host
    const long chk = std::min<long>(n, 300);
    std::vector<u32> nv(n); CK(hipMemcpy(nv.data(), d_nvotes, n * 4, hipMemcpyDeviceToHost));
    std::vector<bmbs_vote> hv(cand_off[chk]); CK(hipMemcpy(hv.data(), d_votes, hv.size() * sizeof(bmbs_vote), hipMemcpyDeviceToHost));
    long bad = 0;
    for (long r = 0; r < chk; r++) {
        std::vector<u64> c;
        for (int q = 0; q < n_seeds[r]; q++) {
            const SeedRec& sr = seeds[(size_t)r * BMBS_MAX_SEEDS + q];
            for (u32 j = 0; j < sr.hits; j++) c.push_back(total - sa[sr.sp + j] - ((u64)sr.len + sr.off));
        }
        std::sort(c.begin(), c.end());
        std::vector<bmbs_vote> v;
        for (size_t i = 0; i < c.size();) { size_t j = i; while (j < c.size() && c[j] == c[i]) j++; bmbs_vote o; o.site = c[i] < 12 ? 0 : c[i] - 12; o.vote = (u32)(j - i); o.pad = 0; v.push_back(o); i = j; }
        std::sort(v.begin(), v.end(), [](const bmbs_vote& x, const bmbs_vote& y) { return x.vote > y.vote; });
        if (nv[r] != v.size()) { bad++; continue; }
        for (size_t i = 0; i < v.size(); i++) if (hv[cand_off[r] + i].site != v[i].site || hv[cand_off[r] + i].vote != v[i].vote) { bad++; break; }
    }
#ifdef VOTE_PROF
    {
        unsigned long long h[4][8];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_vote_prof), sizeof h));
        for (int q = 0; q < 4; q++) if (h[q][0])
            printf("  [vote_prof] class %d: lists %llu | K cycles per list: locate+sort %.1f run-ends+items %.1f vote-order %.1f write %.1f | one-lane fallbacks %llu\n",
                   q, h[q][0], h[q][3] / 1e3 / h[q][0], h[q][4] / 1e3 / h[q][0], h[q][5] / 1e3 / h[q][0], h[q][6] / 1e3 / h[q][0], h[q][7]);
    }
#endif
    double cands = (double)slots;
    printf("%s: %ld lists, %.0f candidates each: best %.3f ms, mean %.3f ms (%.1f us per list-slot of 8192 blocks); first %ld lists %s std::sort\n",
#ifdef VB_OLD
           "round-5 kernel",
#else
           "this tree",
#endif
           n, cands / n, best, sum / reps, 0.0, chk, bad ? "DIFFER from" : "equal");
    return bad ? 1 : 0;
}
