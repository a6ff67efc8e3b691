// tools/host_mem_probe.cpp -- what the host side of the file-to-file pipeline can expect from this box: pread() of a page-cached
// file into page-locked (bmbs_host_alloc) vs ordinary memory, memchr and memcpy over both, single file buffered pwrite vs a shared
// mapping, each with 1 / 8 / 16 / 32 threads.   g++ -O2 -std=c++17 -o host_mem_probe host_mem_probe.cpp -L../bitmapperbs_amd -lbmbs_hip -lpthread
#include "../include/bmbs.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> double par(int T, F f) { const double t0 = now(); std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(f, t); for (auto& x : th) x.join(); return now() - t0; }
int main(int argc, char** argv)
{
    const size_t N = (size_t)(argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30);
    const char* dir = argc > 2 ? argv[2] : "/tmp";
    std::string fn = std::string(dir) + "/hmp.bin", fo = std::string(dir) + "/hmp.out";
    char* pin = (char*)bmbs_host_alloc(N);
    char* reg = (char*)malloc(N);
    if (!pin || !reg) { fprintf(stderr, "alloc failed\n"); return 1; }
    memset(reg, 'A', N); for (size_t i = 150; i < N; i += 151) reg[i] = '\n';
    { int fd = open(fn.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); size_t d = 0; while (d < N) { ssize_t w = write(fd, reg + d, N - d); if (w <= 0) return 1; d += w; } close(fd); }
    const int fd = open(fn.c_str(), O_RDONLY);
    for (int T : {1, 8, 16, 32}) {
        const size_t per = (N / T) & ~(size_t)4095;
        for (int which = 0; which < 2; which++) {
            char* dst = which ? pin : reg;
            const double a = par(T, [&](int t) { size_t o = per * t, e = t == T - 1 ? N : o + per; while (o < e) { ssize_t g = pread(fd, dst + o, e - o, o); if (g <= 0) break; o += g; } });
            size_t cnt = 0; std::vector<size_t> c(T, 0);
            const double b = par(T, [&](int t) { size_t o = per * t, e = t == T - 1 ? N : o + per; const char* q = dst + o; size_t k = 0; while (q < dst + e) { const char* h = (const char*)memchr(q, '\n', dst + e - q); if (!h) break; k++; q = h + 1; } c[t] = k; });
            for (auto x : c) cnt += x;
            char* tmp = (char*)malloc(per + 4096);
            const double m = T <= 16 ? par(T, [&](int t) { char* mine = (char*)malloc(per); memcpy(mine, dst + per * t, per); free(mine); }) : 0;
            free(tmp);
            printf("T=%2d %-6s pread %.2f GB/s  memchr %.2f GB/s (%zu lines)  memcpy-from %.2f GB/s\n", T, which ? "pinned" : "malloc", N / a / 1e9, N / b / 1e9, cnt, m ? N / m / 1e9 : 0.0);
        }
        // output: buffered pwrite to one file vs shared mapping
        { int ofd = open(fo.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
          const double w = par(T, [&](int t) { size_t o = per * t, e = t == T - 1 ? N : o + per; while (o < e) { ssize_t g = pwrite(ofd, reg + o, e - o, o); if (g <= 0) break; o += g; } });
          close(ofd); unlink(fo.c_str());
          ofd = open(fo.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
          if (ftruncate(ofd, N)) return 1;
          char* mp = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, ofd, 0);
          const double x = par(T, [&](int t) { size_t o = per * t, e = t == T - 1 ? N : o + per; memcpy(mp + o, reg + o, e - o); });
          munmap(mp, N); close(ofd); unlink(fo.c_str());
          printf("T=%2d output: pwrite %.2f GB/s   shared mapping %.2f GB/s\n", T, N / w / 1e9, N / x / 1e9); }
    }
    unlink(fn.c_str());
    return 0;
}
