// tests/csrc/sort_check.cpp -- checks bmbs_sort.h's intro_sort_desc against libstdc++ std::sort with
// the reference's comparator (Schema.cpp:560-563) on random, few-valued, sorted, reversed and
// organ-pipe vote arrays.  Prints "OK <cases>" or the first mismatch.
#include "../../bitmapperbs_amd/csrc/bmbs_sort.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
struct seed_votes { uint64_t site, vote; unsigned err; uint64_t end_site; };   // Schema.h:169-176
static bool cmp(const seed_votes& a, const seed_votes& b) { return a.vote > b.vote; }
int main(int argc, char** argv)
{
    unsigned long cases = argc > 1 ? strtoul(argv[1], 0, 10) : 20000;
    std::mt19937_64 rng(12345);
    unsigned long done = 0;
    for (unsigned long c = 0; c < cases; c++) {
        long n;
        int kind = c % 8;
        if (c % 97 == 0) n = 1000 + rng() % 24000; else if (c % 5 == 0) n = 17 + rng() % 300; else n = 1 + rng() % 40;
        int maxv = (c % 3 == 0) ? 2 : (c % 3 == 1) ? 6 : 25;
        std::vector<seed_votes> a(n); std::vector<bmbs_vote> b(n);
        for (long i = 0; i < n; i++) {
            uint64_t v;
            switch (kind) {
                case 0: case 1: case 2: v = 1 + rng() % maxv; break;
                case 3: v = 1 + (i * maxv) / n; break;                 // ascending
                case 4: v = maxv - (i * maxv) / n; break;              // descending
                case 5: v = 1 + (i < n / 2 ? i : n - i) % maxv; break; // organ pipe
                case 6: v = 1; break;                                  // all equal
                default: v = 1 + ((rng() % 10) ? 0 : rng() % maxv); break;
            }
            a[i].site = 1000 + i; a[i].vote = v; a[i].err = 0; a[i].end_site = 0;
            b[i].site = 1000 + i; b[i].vote = (uint32_t)v; b[i].pad = 0;
        }
        std::sort(a.begin(), a.end(), cmp);
        intro_sort_desc(b.data(), n);
        for (long i = 0; i < n; i++)
            if (a[i].site != b[i].site || a[i].vote != b[i].vote) { printf("MISMATCH case %lu n=%ld kind=%d at %ld\n", c, n, kind, i); return 1; }
        // also the plain u64 sort
        std::vector<uint64_t> x(n), y;
        for (long i = 0; i < n; i++) x[i] = rng() % (n < 50 ? 20 : 100000);
        y = x; std::sort(y.begin(), y.end()); sort_u64_asc(x.data(), n);
        if (x != y) { printf("U64 MISMATCH case %lu\n", c); return 1; }
        done++;
    }
    printf("OK %lu\n", done);
    return 0;
}
