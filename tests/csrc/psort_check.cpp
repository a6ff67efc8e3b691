// The data-parallel formulation of std::sort's partition pass that k_vote_long runs on the GPU (bmbs_kernels.hip, vl_partition /
// vl_sort_votes), restated serially and checked against std::sort itself.
#include "../../bitmapperbs_amd/csrc/bmbs_sort.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef uint32_t u32;
struct seed_votes { uint64_t site, vote; unsigned err; uint64_t end_site; };
static bool cmp(const seed_votes& a, const seed_votes& b) { return a.vote > b.vote; }
static inline u32 V(u32 x) { return x >> 24; }
// data-parallel formulation of std::__unguarded_partition_pivot for "vote descending"
static long ppartition(u32* it, long first, long last, std::vector<long>& Lpos, std::vector<long>& Rpos)
{
    long mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
    auto before = [&](long x, long y) { return V(it[x]) > V(it[y]); };
    auto swp = [&](long x, long y) { u32 t = it[x]; it[x] = it[y]; it[y] = t; };
    if (before(a, b)) { if (before(b, c)) swp(first, b); else if (before(a, c)) swp(first, c); else swp(first, a); }
    else if (before(a, c)) swp(first, a);
    else if (before(b, c)) swp(first, c);
    else swp(first, b);
    const u32 pv = V(it[first]);
    Lpos.clear(); Rpos.clear();
    for (long i = first + 1; i < last; i++) if (V(it[i]) <= pv) Lpos.push_back(i);          // ascending
    for (long i = last - 1; i > first; i--) if (V(it[i]) >= pv) Rpos.push_back(i);          // descending
    long T = 0;
    const long m = (long)std::min(Lpos.size(), Rpos.size());
    while (T < m && Lpos[T] < Rpos[T]) T++;
    for (long t = 0; t < T; t++) swp(Lpos[t], Rpos[t]);
    if (T == 0) return Lpos[0];
    const long lT = T < (long)Lpos.size() ? Lpos[T] : (1L << 60);
    return std::min(lT, Rpos[T - 1]);
}
static long g_heap = 0;
static bool pintro(std::vector<u32>& it)
{
    const long n = (long)it.size();
    std::vector<long> L, R;
    if (n > 16) {
        int lg = 0; for (long t = n; t > 1; t >>= 1) lg++;
        struct Rg { long f, l; int d; };
        std::vector<Rg> st; st.push_back({0, n, 2 * lg});
        while (!st.empty()) {
            Rg r = st.back(); st.pop_back();
            while (r.l - r.f > 16) {
                if (r.d == 0) { g_heap++; bmbs_sort_detail::heap_sort((bmbs_vk*)it.data(), r.f, r.l); break; }
                --r.d;
                long cut = ppartition(it.data(), r.f, r.l, L, R);
                st.push_back({cut, r.l, r.d});
                r.l = cut;
            }
        }
    }
    std::stable_sort(it.begin(), it.end(), [](u32 a, u32 b) { return V(a) > V(b); });
    return true;
}
int main(int argc, char** argv)
{
    unsigned long cases = argc > 1 ? strtoul(argv[1], 0, 10) : 20000;
    std::mt19937_64 rng(777);
    for (unsigned long c = 0; c < cases; c++) {
        long n; int kind = c % 8;
        if (c % 97 == 0) n = 1000 + rng() % 24000; else if (c % 5 == 0) n = 17 + rng() % 300; else n = 1 + rng() % 40;
        int maxv = (c % 3 == 0) ? 2 : (c % 3 == 1) ? 6 : 25;
        std::vector<seed_votes> a(n); std::vector<u32> b(n);
        for (long i = 0; i < n; i++) {
            uint64_t v;
            switch (kind) {
                case 0: case 1: case 2: v = 1 + rng() % maxv; break;
                case 3: v = 1 + (i * maxv) / n; break;
                case 4: v = maxv - (i * maxv) / n; break;
                case 5: v = 1 + (i < n / 2 ? i : n - i) % maxv; break;
                case 6: v = 1; break;
                default: v = 1 + ((rng() % 10) ? 0 : rng() % maxv); break;
            }
            a[i].site = i; a[i].vote = v; a[i].err = 0; a[i].end_site = 0;
            b[i] = ((u32)v << 24) | (u32)i;
        }
        std::sort(a.begin(), a.end(), cmp);
        pintro(b);
        for (long i = 0; i < n; i++)
            if (a[i].site != (b[i] & 0xffffff) || a[i].vote != V(b[i])) { printf("MISMATCH case %lu n=%ld kind=%d maxv=%d at %ld\n", c, n, kind, maxv, i); return 1; }
    }
    // the depth-limit / heapsort branch: McIlroy's adversary run against std::sort gives inputs that reach it
    {
        static int* val; static int nsolid, candidate, gas;
        for (int n : {64, 120, 200, 255}) {
            std::vector<int> vv(n, n), ptr(n);
            for (int i = 0; i < n; i++) ptr[i] = i;
            val = vv.data(); nsolid = 0; candidate = 0; gas = n;
            std::sort(ptr.begin(), ptr.end(), [](int a, int b) {
                if (val[a] == gas && val[b] == gas) { if (a == candidate) val[a] = nsolid++; else val[b] = nsolid++; }
                if (val[a] == gas) candidate = a; else if (val[b] == gas) candidate = b;
                return val[a] < val[b];
            });
            std::vector<seed_votes> a(n); std::vector<u32> b(n);
            for (int i = 0; i < n; i++) {
                const u32 v = (u32)(n - (vv[i] >= n ? n - 1 : vv[i]));
                a[i].site = i; a[i].vote = v; a[i].err = 0; a[i].end_site = 0;
                b[i] = (v << 24) | (u32)i;
            }
            const long before = g_heap;
            std::sort(a.begin(), a.end(), cmp);
            pintro(b);
            if (g_heap == before) { printf("killer n=%d did not reach the heapsort branch\n", n); return 1; }
            for (int i = 0; i < n; i++)
                if (a[i].site != (b[i] & 0xffffff)) { printf("MISMATCH killer n=%d at %d\n", n, i); return 1; }
        }
    }
    printf("OK %lu\n", cases);
    return 0;
}
