// test helper: the visiting order libstdc++'s std::sort gives vote lists under the reference's comparator
// ("vote descending", Schema.cpp:560-563, 24986), for comparisons with bmbs_vote_order_batch
#include <algorithm>
#include <cstdint>
#include <vector>
struct seed_votes { uint64_t site, vote; unsigned err; uint64_t end_site; };
static bool by_vote(const seed_votes& a, const seed_votes& b) { return a.vote > b.vote; }
extern "C" void std_order(const uint8_t* vote, const int64_t* seg_off, int64_t n_seg, uint32_t* perm)
{
    std::vector<seed_votes> v;
    for (int64_t s = 0; s < n_seg; s++) {
        const int64_t a = seg_off[s], n = seg_off[s + 1] - a;
        v.resize(n);
        for (int64_t i = 0; i < n; i++) { v[i].site = i; v[i].vote = vote[a + i]; v[i].err = 0; v[i].end_site = 0; }
        std::sort(v.begin(), v.end(), by_vote);
        for (int64_t i = 0; i < n; i++) perm[a + i] = (uint32_t)v[i].site;
    }
}

// McIlroy's adversary ("A killer adversary for quicksort", 1999) run against std::sort itself: vote[0..n) (n <= 255, values
// 1..n) on which std::sort with the vote-descending comparator degenerates and falls back to heapsort
namespace {
int* g_val; int g_nsolid, g_candidate, g_gas;
bool adversary_less(int a, int b)
{
    if (g_val[a] == g_gas && g_val[b] == g_gas) { if (a == g_candidate) g_val[a] = g_nsolid++; else g_val[b] = g_nsolid++; }
    if (g_val[a] == g_gas) g_candidate = a; else if (g_val[b] == g_gas) g_candidate = b;
    return g_val[a] < g_val[b];
}
}
extern "C" void killer_votes(int n, uint8_t* vote)
{
    std::vector<int> val(n, n), ptr(n);
    for (int i = 0; i < n; i++) ptr[i] = i;
    g_val = val.data(); g_nsolid = 0; g_candidate = 0; g_gas = n;
    std::sort(ptr.begin(), ptr.end(), adversary_less);
    for (int i = 0; i < n; i++) vote[i] = (uint8_t)(n - (val[i] >= n ? n - 1 : val[i]));      // vote a > vote b  <=>  val a < val b
}
