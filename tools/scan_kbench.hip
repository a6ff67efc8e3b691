// tools/scan_kbench.hip -- the two scans of k_scan.hip alone: the chained one-launch form against the two-launch form and a host prefix
// sum, over sizes around the tile edges and the mapping path's sizes, many scans back to back on one status buffer (the epoch is what
// keeps an earlier scan's words apart).
// build (here):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/scan_kbench tools/scan_kbench.hip
// run (GPU box): timeout 120 ./tools/scan_kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef unsigned int u32; typedef unsigned long long u64; typedef unsigned char u8;
#define DEVI __device__ __forceinline__
#include "../bitmapperbs_amd/csrc/k_scan.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main()
{
    CK(hipSetDevice(0));
    const u64 N = 24000000;
    std::vector<u32> h(N);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (u64 i = 0; i < N; i++) h[i] = (rnd() % 16 == 0) ? (u32)(rnd() % 3000) : (u32)(rnd() % 3);
    u32 *d_in, *d_list; u64 *d_out, *d_out2, *d_status, *d_total, *d_bs, *d_ndev; unsigned int* d_ticket;
    CK(hipMalloc(&d_in, N * 4)); CK(hipMemcpy(d_in, h.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (N + 1) * 8)); CK(hipMalloc(&d_out2, (N + 1) * 8)); CK(hipMalloc(&d_list, N * 4));
    const u64 per = SCAN_BLOCK * SCAN_ITEMS, NB = (N + per - 1) / per;
    CK(hipMalloc(&d_status, (NB + 1) * 8)); CK(hipMemset(d_status, 0, (NB + 1) * 8));
    CK(hipMalloc(&d_bs, (NB + 1) * 8));
    CK(hipMalloc(&d_total, 16)); CK(hipMalloc(&d_ticket, 64)); CK(hipMemset(d_ticket, 0, 64)); CK(hipMalloc(&d_ndev, 8));
    u32 base = 0, epoch = 0;
    auto chain = [&](u64 n, u64* out, u32* list, int nz, const u64* ndev) {
        const u64 nb = n ? (n + per * SCAN_SUB - 1) / (per * SCAN_SUB) : 1;
        ++epoch;
        hipLaunchKernelGGL(k_scan_chain, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, 0, d_in, n, out, list, nz, ndev, d_total, d_ticket, base, d_status, epoch);
        base += (u32)nb;
    };
    auto two = [&](u64 n, u64* out, u32* list, int nz, const u64* ndev) {
        const u64 nb = n ? (n + per - 1) / per : 1;
        hipLaunchKernelGGL(k_scan_partial, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, 0, d_in, n, d_bs, nz, ndev);
        hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, 0, d_in, n, d_bs, out, list, nz, ndev, d_total + 1);
    };
    std::vector<u64> sizes = {0, 1, 7, 8, 9, 2047, 2048, 2049, 4096, 4097, 100000, 131072, 1000003, 2048 * 64, 2048 * 64 + 1, 2048 * 65, 2048 * 129 + 5, 10000000, N};
    for (int i = 0; i < 40; i++) sizes.push_back(rnd() % N);
    std::vector<u64> ho(N + 1), hc(N + 1);
    long bad = 0;
    for (int rep = 0; rep < 3; rep++)
    for (u64 n : sizes) for (int nz = 0; nz < 2; nz++) {
        CK(hipMemset(d_out, 0xff, (n + 1) * 8));
        chain(n, d_out, nullptr, nz, nullptr);
        u64 tot; CK(hipMemcpy(&tot, d_total, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), d_out, (n + 1) * 8, hipMemcpyDeviceToHost));
        u64 run = 0; bool ok = true;
        for (u64 i = 0; i < n; i++) { if (hc[i] != run) { ok = false; break; } run += nz ? (h[i] != 0) : h[i]; }
        if (hc[n] != run || tot != run) ok = false;
        if (!ok) { printf("chain DIFFERS n=%llu nz=%d\n", n, nz); bad++; }
        // list form (flags): positions of the non-zeros, and the device-side bound
        if (nz) {
            const u64 bound = n / 2 + 1; CK(hipMemcpy(d_ndev, &bound, 8, hipMemcpyHostToDevice));
            chain(n, nullptr, d_list, 1, d_ndev);
            CK(hipMemcpy(&tot, d_total, 8, hipMemcpyDeviceToHost));
            std::vector<u32> hl(tot ? tot : 1); CK(hipMemcpy(hl.data(), d_list, tot * 4, hipMemcpyDeviceToHost));
            u64 k = 0; const u64 lim = std::min(n, bound);
            for (u64 i = 0; i < lim; i++) if (h[i]) { if (k >= tot || hl[k] != (u32)i) { ok = false; break; } k++; }
            if (k != tot) ok = false;
            if (!ok) { printf("chain list DIFFERS n=%llu\n", n); bad++; }
        }
    }
    // timing at the mapping path's sizes
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (u64 n : {(u64)200000, (u64)2500000, (u64)10000000, N}) {
        float t[2];
        for (int which = 0; which < 2; which++) {
            for (int i = 0; i < 3; i++) which ? chain(n, d_out, nullptr, 0, nullptr) : two(n, d_out2, nullptr, 0, nullptr);
            CK(hipEventRecord(a));
            for (int i = 0; i < 20; i++) which ? chain(n, d_out, nullptr, 0, nullptr) : two(n, d_out2, nullptr, 0, nullptr);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&t[which], a, b));
        }
        CK(hipMemcpy(ho.data(), d_out2, (n + 1) * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), d_out, (n + 1) * 8, hipMemcpyDeviceToHost));
        const bool same = !memcmp(ho.data(), hc.data(), (n + 1) * 8);
        if (!same) bad++;
        printf("n=%9llu: two launches %.1f us, chained %.1f us per scan (%.0f GB/s over 12 B per entry); outputs %s\n", n, t[0] * 50, t[1] * 50,
               12.0 * n / (t[1] * 50e-6) / 1e9, same ? "equal" : "DIFFER");
    }
    printf("%s (%ld differences)\n", bad ? "FAILED" : "OK", bad);
    return bad ? 1 : 0;
}
