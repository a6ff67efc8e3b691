// bitmapperbs_amd/csrc/bmbs_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the
// bisulfite mapping hot path.  Integer / bit work only: no MFMA, HBM-gather bound.
//
// Stage split (one kernel per work granularity, exclusive scans / compacted lists in between so that every kernel runs
// dense lanes):
//   k_seed_first / k_seed_decide / k_seed_second / k_seed_extra
//                      read per lane (wave-owned chunks)  K1-K5 + the seeding state machine of Map_Single_Seq_end_to_end
//   k_vote_fused / k_vote_mid / k_vote_long
//                      read (<= 16, <= 32 candidates) per lane, wave or block per read beyond
//                                                          K5/K6 + a9/a10: locate, sort, run-length votes, std::sort-exact order
//   k_filter           candidate per lane                  K7+K8: window fetch + BS banded Myers on bit planes
//   k_reduce           read per lane                       K9: ordered min / ambiguity / second_best_diff scan
//   k_align_ungapped / k_align_sw<KB>
//                      winner per lane                     K11-K13: un-gapped recheck, banded affine SW + CIGAR + NM
//   k_finalize         read per lane                       MAPQ LUT, doubled -> chromosome coordinates, off-end, stats
//   k_pe_* / k_pes_* / k_*_pe                              the paired-end (fast and --sensitive) counterparts
#include "bmbs_dev.h"
#include "bmbs_sort.h"
#include <type_traits>

#define DEVI __device__ __forceinline__

#define SHARD(p) ((p) + (size_t)(blockIdx.x & (BMBS_SHARDS - 1)) * BMBS_SHARD_WORDS)
// event counter words beyond the seeding ones (bmbs_counters_all): what the list kernels handled, for the per-kernel byte terms of the
// bench line.  (-DBMBS_UTIL builds overload words 8..13 with lane-utilisation probes.)
#define CNT_CAND_MID   9      // candidates located by k_vote_mid / k_vote_pe_mid (lists of 17..32)
#define CNT_CAND_LONG 10      // ... by the wave form of k_vote_long / k_vote_pe_long (33..256)
#define CNT_CAND_BIG  11      // ... by their block forms (beyond 256)
#define CNT_LISTS_LONG 12     // lists the wave and block forms took
#define CNT_PEF_ENTRIES 13    // list entries filter_pairs read (both mates of the pairs that ran it)
#define CNT_CAND_RESEED 14    // candidates located by k_pes_vote[_long] (--sensitive re-seeding)
#define CNT_PREFILTER_DROP 15 // located sites the paired-end vote kernels dropped before their sort (no partner on the mate's list)
// sum of v over the 64 lanes of a wave (EVERY lane has to arrive), one atomic per wave
DEVI void wave_count_add(unsigned long long* counters, int word, u32 v)
{
    if (!counters) return;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&SHARD(counters)[word], (unsigned long long)v);
}

// ---- per-wave timeline (diagnostic, BMBS_WAVELOG=<file>) ------------------------------------------------------------------------
// A kernel that calls wavelog_begin / wavelog_end records for each of its working waves where it ran (XCC, SE, CU, SIMD), when
// (100 MHz wall clock) and for how many shader cycles: 4 u64 per wave, dumped by bmbs_destroy and read by tools/wavelog.py.
// That is the only way to see occupancy rounds, dispatch imbalance and the actual shader clock of one kernel; with the variable
// unset g_wavelog.buf is null and the cost is one scalar load per wave.
struct WaveLogDev { unsigned long long* buf; unsigned int* count; unsigned int cap; };
__device__ WaveLogDev g_wavelog;
struct WaveLogT { u64 wall, cyc; bool on; };
DEVI WaveLogT wavelog_begin()
{
    WaveLogT t; t.on = g_wavelog.buf != nullptr; t.wall = 0; t.cyc = 0;
    if (t.on) { t.wall = wall_clock64(); t.cyc = __builtin_readcyclecounter(); }
    return t;
}
DEVI void wavelog_end(const WaveLogT& t, int kernel_id)
{
    if (!t.on) return;
    const u64 w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) != 0) return;
    const u32 hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const u32 idx = atomicAdd(g_wavelog.count, 1u);
    if (idx >= g_wavelog.cap) return;
    unsigned long long* o = g_wavelog.buf + (size_t)idx * 4;
    o[0] = ((u64)kernel_id << 56) | ((u64)(xcc & 0xffu) << 32) | hw;
    o[1] = t.wall; o[2] = w1; o[3] = c1 - t.cyc;
}

#include "k_index.hip"
#include "k_rows.hip"
#include "k_attach.hip"
#include "k_scan.hip"
#include "k_seed.hip"
#include "k_vote.hip"
#include "k_filter.hip"
#include "k_reduce.hip"
#include "k_align.hip"
#include "k_finalize.hip"
#include "k_pe_fast.hip"
#include "k_pe_sensitive.hip"
