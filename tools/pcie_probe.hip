// tools/pcie_probe.hip -- what the host link of the MI355X box gives to page-locked copies: H2D alone, D2H alone, both directions
// at once on two streams, per transfer size; and the same with the copy split over 2-4 streams.  Sizes the overlapped
// host-buffer path (bmbs_map_*: H2D of chunk i+1, kernels of chunk i, D2H of chunk i-1) and the FASTQ-in / SAM-out pipeline of
// bmbs_search.  Build: hipcc -O2 --offload-arch=gfx950 -o tools/pcie_probe tools/pcie_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <atomic>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// background host traffic: N threads copying 64 MiB blocks between buffers of their own (what a FASTQ reader's pread()s do)
static std::atomic<bool> g_bg_stop(false);
static std::atomic<size_t> g_bg_bytes(0);

int main(int argc, char** argv)
{
    const size_t total = (argc > 1 ? (size_t)atol(argv[1]) : 2048) << 20;       // bytes moved per measurement
    const int bg = argc > 2 ? atoi(argv[2]) : 0;                                // background memcpy threads
    std::vector<std::thread> bgt;
    for (int t = 0; t < bg; t++)
        bgt.emplace_back([] {
            const size_t n = (size_t)64 << 20;
            char* a = (char*)malloc(n); char* b = (char*)malloc(n);
            memset(a, 1, n); memset(b, 2, n);
            while (!g_bg_stop) { memcpy(b, a, n); g_bg_bytes += n; }
            free(a); free(b);
        });
    const double t_bg0 = now();
    CK(hipSetDevice(0));
    char *h_in, *h_out, *d_in, *d_out;
    CK(hipHostMalloc((void**)&h_in, total, hipHostMallocPortable));
    CK(hipHostMalloc((void**)&h_out, total, hipHostMallocPortable));
    CK(hipMalloc((void**)&d_in, total));
    CK(hipMalloc((void**)&d_out, total));
    {   // touch the page-locked memory from several threads (first-touch placement)
        std::vector<std::thread> th;
        for (int t = 0; t < 8; t++) th.emplace_back([&, t] { memset(h_in + total / 8 * t, t + 1, total / 8); memset(h_out + total / 8 * t, 0, total / 8); });
        for (auto& x : th) x.join();
    }
    hipStream_t s[8];
    for (auto& x : s) CK(hipStreamCreate(&x));
    CK(hipMemcpy(d_in, h_in, total, hipMemcpyHostToDevice));
    CK(hipMemcpy(h_out, d_out, total, hipMemcpyDeviceToHost));
    printf("bytes per measurement: %zu MiB\n", total >> 20);
    printf("%-44s %10s %10s %10s\n", "pattern (chunk)", "H2D GB/s", "D2H GB/s", "sum");
    for (size_t chunk : {(size_t)32 << 20, (size_t)256 << 20}) {
        const size_t nch = total / chunk;
        // one direction at a time, one stream
        double t0 = now();
        for (size_t i = 0; i < nch; i++) CK(hipMemcpyAsync(d_in + i * chunk, h_in + i * chunk, chunk, hipMemcpyHostToDevice, s[0]));
        CK(hipStreamSynchronize(s[0]));
        const double h2d = total / (now() - t0) / 1e9;
        t0 = now();
        for (size_t i = 0; i < nch; i++) CK(hipMemcpyAsync(h_out + i * chunk, d_out + i * chunk, chunk, hipMemcpyDeviceToHost, s[1]));
        CK(hipStreamSynchronize(s[1]));
        const double d2h = total / (now() - t0) / 1e9;
        printf("alone, 1 stream (%4zu MiB)                     %10.1f %10.1f\n", chunk >> 20, h2d, d2h);
        // both directions at once
        t0 = now();
        for (size_t i = 0; i < nch; i++) {
            CK(hipMemcpyAsync(d_in + i * chunk, h_in + i * chunk, chunk, hipMemcpyHostToDevice, s[0]));
            CK(hipMemcpyAsync(h_out + i * chunk, d_out + i * chunk, chunk, hipMemcpyDeviceToHost, s[1]));
        }
        CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
        const double both = total / (now() - t0) / 1e9;
        printf("both directions, 2 streams (%4zu MiB)          %10.1f %10.1f %10.1f\n", chunk >> 20, both, both, 2 * both);
        // each direction split over 2 and 3 streams
        for (int ns : {2, 3}) {
            t0 = now();
            for (size_t i = 0; i < nch; i++) {
                CK(hipMemcpyAsync(d_in + i * chunk, h_in + i * chunk, chunk, hipMemcpyHostToDevice, s[i % ns]));
                CK(hipMemcpyAsync(h_out + i * chunk, d_out + i * chunk, chunk, hipMemcpyDeviceToHost, s[4 + i % ns]));
            }
            for (auto& x : s) CK(hipStreamSynchronize(x));
            const double b2 = total / (now() - t0) / 1e9;
            printf("both directions, %d + %d streams (%4zu MiB)      %10.1f %10.1f %10.1f\n", ns, ns, chunk >> 20, b2, b2, 2 * b2);
        }
    }
    // pageable memory for comparison
    {
        char* p = (char*)malloc(total);
        memset(p, 1, total);
        double t0 = now();
        CK(hipMemcpy(d_in, p, total, hipMemcpyHostToDevice));
        const double h2d = total / (now() - t0) / 1e9;
        t0 = now();
        CK(hipMemcpy(p, d_out, total, hipMemcpyDeviceToHost));
        const double d2h = total / (now() - t0) / 1e9;
        printf("pageable malloc, hipMemcpy                     %10.1f %10.1f\n", h2d, d2h);
        free(p);
    }
    g_bg_stop = true;
    for (auto& t : bgt) t.join();
    if (bg) printf("background: %d threads copied %.1f GB/s (read + write of that each)\n", bg, (double)g_bg_bytes / (now() - t_bg0) / 1e9);
    return 0;
}
