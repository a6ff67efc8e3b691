// tools/pgz_test.cpp -- pgz_test <file.gz> <threads> <span bytes> [q] : the block-parallel inflater of the driver (csrc/pgz.h) on its own:
// inflated text to stdout (q: only the rate to stderr).  g++ -O2 -std=c++17 -Ibitmapperbs_amd/csrc -o /tmp/pgz_test tools/pgz_test.cpp -lz -lpthread
#include "pgz.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <time.h>
int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb; fstat(fd, &sb);
    const unsigned char* p = (const unsigned char*)mmap(nullptr, sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    pgz::Options o; o.threads = atoi(argv[2]); o.span = (size_t)atol(argv[3]);
    std::mutex m; std::map<long, std::vector<char>> got; long next = 0; size_t total = 0;
    const bool quiet = argc > 4;
    timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    pgz::Engine e(p, (size_t)sb.st_size, 0, 0, o, [&](long id, std::vector<char>&& c) {
        std::lock_guard<std::mutex> l(m);
        got[id] = std::move(c);
        while (!got.empty() && got.begin()->first == next) { auto& v = got.begin()->second; total += v.size(); if (!quiet) fwrite(v.data(), 1, v.size(), stdout); got.erase(got.begin()); next++; }
    });
    e.start();
    const std::string err = e.wait();
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    fprintf(stderr, "%s: %zu bytes in %.3f s = %.1f MB/s, redone %ld%s%s\n", argv[1], total, s, total / s / 1e6, e.redone(), err.empty() ? "" : "  ERROR: ", err.c_str());
    return err.empty() ? 0 : 1;
}
