// tools/link_probe.hip -- which neighbour slows the page-locked copies of the file-to-file path down: a copy loop per direction
// (315 MB up, 361 MB down, as a 1 M-read batch of bmbs_search moves) alone, beside busy compute kernels on a third stream, beside
// host threads that copy memory (what pread() into the staging windows does), beside both.
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/link_probe tools/link_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_busy(unsigned long long* p, size_t n, int rounds)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = i;
    for (int r = 0; r < rounds; r++) { v = v * 6364136223846793005ull + p[(i * 977 + (size_t)r * 64) % n]; }
    if (v == 42) p[0] = v;
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    CK(hipSetDevice(0));
    const size_t UP = (size_t)315 << 20, DN = (size_t)361 << 20;
    const int NB = 4;
    char *h_in[NB], *h_out[NB], *d_in, *d_out;
    for (int i = 0; i < NB; i++) { CK(hipHostMalloc((void**)&h_in[i], UP, hipHostMallocPortable)); CK(hipHostMalloc((void**)&h_out[i], DN, hipHostMallocPortable)); memset(h_in[i], i + 1, UP); memset(h_out[i], 0, DN); }
    CK(hipMalloc((void**)&d_in, UP)); CK(hipMalloc((void**)&d_out, DN));
    unsigned long long* d_work; const size_t WN = (size_t)1 << 28;
    CK(hipMalloc((void**)&d_work, WN * 8)); CK(hipMemset(d_work, 1, WN * 8));
    hipStream_t s_up, s_dn, s_k;
    CK(hipStreamCreate(&s_up)); CK(hipStreamCreate(&s_dn)); CK(hipStreamCreate(&s_k));
    printf("%-46s %10s %10s\n", "neighbours", "H2D GB/s", "D2H GB/s");
    // 'page cache': 2 GiB of ordinary memory the refill threads copy from
    const size_t PC = (size_t)2 << 30;
    char* pc = (char*)malloc(PC);
    { std::vector<std::thread> tt; for (int t = 0; t < 16; t++) tt.emplace_back([&, t] { memset(pc + PC / 16 * t, t + 3, PC / 16); }); for (auto& x : tt) x.join(); }
    for (int mode : {0, 1, 2, 3, 8, 9, 16, 17, 24}) {
        const bool kern = mode & 1, host = mode & 2, pieces = mode & 4, refill = mode & 8, refill_nt = mode & 16;
        std::atomic<bool> stop(false);
        std::atomic<size_t> up_b(0), dn_b(0), host_b(0);
        std::vector<std::thread> th;
        th.emplace_back([&] { int i = 0; while (!stop) { CK(hipMemcpyAsync(d_in, h_in[i % NB], UP, hipMemcpyHostToDevice, s_up)); CK(hipStreamSynchronize(s_up)); up_b += UP; i++; } });
        th.emplace_back([&] {
            int i = 0;
            while (!stop) {
                if (!pieces) CK(hipMemcpyAsync(h_out[i % NB], d_out, DN, hipMemcpyDeviceToHost, s_dn));
                else for (size_t o = 0; o < DN; o += (size_t)128 << 20) CK(hipMemcpyAsync(h_out[i % NB] + o, d_out + o, std::min((size_t)128 << 20, DN - o), hipMemcpyDeviceToHost, s_dn));
                CK(hipStreamSynchronize(s_dn)); dn_b += DN; i++;
            }
        });
        if (kern) th.emplace_back([&] { while (!stop) { hipLaunchKernelGGL(k_busy, dim3(1 << 16), dim3(256), 0, s_k, d_work, WN, 200); CK(hipStreamSynchronize(s_k)); } });
        if (host)
            for (int t = 0; t < 16; t++)
                th.emplace_back([&, t] {
                    const size_t n = (size_t)32 << 20;
                    char* a = (char*)malloc(n); char* b = (char*)malloc(n);
                    memset(a, t, n);
                    unsigned long long sink = 0;
                    while (!stop) { memcpy(b, a, n); sink += (unsigned char)b[(sink * 4099) % n]; a[sink % n]++; host_b += n; }
                    if (sink == 1) printf("!");
                    free(a); free(b);
                });
        // refill: like the FASTQ reader, 16 threads copy 'file' bytes into the page-locked INPUT windows that the upload loop cycles
        // through (plain memcpy, or non-temporal stores that leave no dirty lines in the caches)
        if (refill || refill_nt)
            for (int t = 0; t < 16; t++)
                th.emplace_back([&, t] {
                    size_t src = (size_t)t * (PC / 16), k = 0;
                    while (!stop) {
                        char* dst = h_in[k % NB] + UP / 16 * t;
                        const size_t n = UP / 16;
                        if (src + n > PC) src = 0;
                        if (!refill_nt) memcpy(dst, pc + src, n);
                        else for (size_t o = 0; o + 16 <= n; o += 16) { const __int128 v = *(const __int128*)(pc + src + o); __builtin_nontemporal_store(v, (__int128*)(dst + o)); }
                        src += n; k++; host_b += n;
                    }
                });
        const double t0 = now();
        std::this_thread::sleep_for(std::chrono::duration<double>(secs));
        stop = true;
        for (auto& t : th) t.join();
        const double dt = now() - t0;
        char label[128];
        snprintf(label, sizeof label, "%s%s%s%s%s", kern ? "compute kernels " : "", host ? "16 host copy threads " : "", pieces ? "(D2H in 128 MiB pieces)" : "", refill ? "16 threads refilling the input windows " : "", refill_nt ? "16 refilling with non-temporal stores " : "");
        printf("%-46s %10.1f %10.1f   host copies %.1f GB/s\n", label[0] ? label : "nothing", up_b / dt / 1e9, dn_b / dt / 1e9, host_b / dt / 1e9);
    }
    return 0;
}
