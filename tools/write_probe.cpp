// tools/write_probe.cpp -- what ONE output file takes on this box: buffered pwrite vs O_DIRECT pwrite (4 MiB aligned blocks), with and
// without fallocate, 4 / 8 / 16 / 32 threads.   g++ -O2 -std=c++17 -o write_probe write_probe.cpp -lpthread ; ./write_probe [GiB] [dir]
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/statfs.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const size_t N = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) * (1ull << 30);
    const char* dir = argc > 2 ? argv[2] : "/tmp";
    struct statfs sf; if (!statfs(dir, &sf)) printf("dir %s: f_type 0x%lx, bsize %ld, free %.1f GB\n", dir, (long)sf.f_type, (long)sf.f_bsize, (double)sf.f_bavail * sf.f_bsize / 1e9);
    const size_t BLK = 4u << 20;
    char* buf = nullptr; if (posix_memalign((void**)&buf, 4096, BLK * 32)) return 1;
    memset(buf, 'A', BLK * 32);
    const std::string fn = std::string(dir) + "/wprobe.bin";
    for (int direct = 0; direct < 2; direct++)
        for (int prealloc = 0; prealloc < 2; prealloc++)
            for (int T : {4, 8, 16, 32}) {
                unlink(fn.c_str());
                int fd = open(fn.c_str(), O_WRONLY | O_CREAT | O_TRUNC | (direct ? O_DIRECT : 0), 0644);
                if (fd < 0) { printf("open(direct=%d) failed: %s\n", direct, strerror(errno)); break; }
                if (prealloc && posix_fallocate(fd, 0, N)) { printf("fallocate failed\n"); close(fd); continue; }
                const size_t nblk = N / BLK;
                bool ok = true;
                const double t0 = now();
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++) th.emplace_back([&, t] {
                    for (size_t b = t; b < nblk; b += T) if (pwrite(fd, buf + (size_t)t * BLK, BLK, (off_t)(b * BLK)) != (ssize_t)BLK) { ok = false; break; }
                });
                for (auto& x : th) x.join();
                const double t1 = now();
                close(fd);
                printf("%s %s T=%2d: %.2f GB/s%s\n", direct ? "O_DIRECT" : "buffered", prealloc ? "fallocate" : "extending", T, N / (t1 - t0) / 1e9, ok ? "" : "  (FAILED)");
                fflush(stdout);
            }
    unlink(fn.c_str());
    return 0;
}
