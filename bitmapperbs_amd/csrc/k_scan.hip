// bitmapperbs_amd/csrc/k_scan.hip -- scan (exclusive, u32 -> u64), two launches: tile sums, then every tile adds up the sums before it (L2 hits) and writes its part
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
// ================================================================================================
// scan (exclusive, u32 -> u64), two launches: tile sums, then every tile adds up the sums before it (L2 hits) and writes its part
// ================================================================================================
#define SCAN_BLOCK 256
#define SCAN_ITEMS 8            // per thread
// block-wide inclusive scan of one u64 per thread (wave shuffles + one LDS hop); returns the inclusive value, *total the block sum
DEVI u64 block_scan_incl(u64 v, u64* sh_waves, u64& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(v, d); if (lane >= d) v += t; }
    if (lane == 63) sh_waves[w] = v;
    __syncthreads();
    u64 add = 0, tot = 0;
    for (int i = 0; i < nw; i++) { const u64 x = sh_waves[i]; if (i < w) add += x; tot += x; }
    __syncthreads();
    total = tot;
    return v + add;
}
// eight consecutive u32 of a thread, two 16-byte loads when whole (the inputs are hipMalloc'ed: 16-byte aligned)
DEVI void scan_load8(const u32* in, u64 base, u64 n, u32 x[SCAN_ITEMS])
{
    if (base + SCAN_ITEMS <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(in + base), b = *reinterpret_cast<const uint4*>(in + base + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) x[j] = base + j < n ? in[base + j] : 0u;
    }
}
// nz: the values count as flags (x != 0)
// n_dev != nullptr: only the first min(n, *n_dev) entries exist (a count an earlier kernel of the same stream left in device
// memory: the launch is sized for the capacity n and the host never waits for the count)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_partial(const u32* in, u64 n, u64* block_sums, int nz, const u64* __restrict__ n_dev)
{
    __shared__ u64 sh[SCAN_BLOCK / 64];
    if (n_dev) { const u64 nd = *n_dev; if (nd < n) n = nd; }
    const u64 base = (u64)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + (u64)threadIdx.x * SCAN_ITEMS;
    u32 x[SCAN_ITEMS];
    scan_load8(in, base, n, x);
    if (nz) {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) x[j] = x[j] != 0;
    }
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += x[j];
    u64 total;
    (void)block_scan_incl(s, sh, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// list != nullptr: `in` holds 0/1 flags and the positions of the ones are written, in order, to list[]; the offsets
// themselves are not stored (the work lists of the seeding stages need nothing else)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_final(const u32* in, u64 n, const u64* __restrict__ block_sums, u64* out, u32* list, int nz, const u64* __restrict__ n_dev, u64* total)
{
    __shared__ u64 sh[SCAN_BLOCK / 64];
    if (n_dev) { const u64 nd = *n_dev; if (nd < n) n = nd; }
    const u64 base = (u64)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + (u64)threadIdx.x * SCAN_ITEMS;
    u32 x[SCAN_ITEMS];
    scan_load8(in, base, n, x);
    if (nz) {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) x[j] = x[j] != 0;
    }
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += x[j];
    // what the tiles before this one hold: 256 threads over blockIdx.x sums
    u64 before = 0;
    for (u32 i = threadIdx.x; i < blockIdx.x; i += SCAN_BLOCK) before += block_sums[i];
    u64 prefix;
    (void)block_scan_incl(before, sh, prefix);
    u64 tot;
    const u64 incl = block_scan_incl(s, sh, tot);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = prefix + tot;
    u64 run = incl - s + prefix;
    if (list) {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n && x[j]) { list[run] = (u32)(base + j); run += x[j]; }
        return;
    }
    if (base + SCAN_ITEMS <= n) {
        u64 o[SCAN_ITEMS];
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) { o[j] = run; run += x[j]; }
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(out + base);        // out is 16-byte aligned, base a multiple of 8
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j += 2) dst[j >> 1] = make_ulonglong2(o[j], o[j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n) { out[base + j] = run; run += x[j]; }
    }
    if (base <= n && n < base + SCAN_ITEMS) out[n] = run;                   // the thread that owns position n writes the total
    if (n % ((u64)SCAN_BLOCK * SCAN_ITEMS) == 0 && n && base + SCAN_ITEMS == n) out[n] = run;
}
// ------------------------------------------------------------------------------------------------
// The same scan in ONE launch (a chained scan with look-back): a block takes its tile from a ticket (so every tile before it is
// held by a block that is already running: nothing waits on a block that has not started), publishes the tile's sum, looks back over
// the status words of the tiles before it -- 64 at a time, one per lane -- until it meets one that already carries its inclusive
// prefix, then publishes its own.  A status word is ONE 8-byte value {epoch 24 | state 2 | value 38} written by one relaxed
// device-scope store and polled by relaxed device-scope loads: the word is the datum, so no fence orders anything around it.  The
// epoch is the host's count of scans on this lane (scan_u32 zeroes the words when it wraps), which is what makes the words of
// earlier scans read as "not there yet".
// ------------------------------------------------------------------------------------------------
#define SCAN_VAL_BITS 38
#define SCAN_VAL_MASK ((1ull << SCAN_VAL_BITS) - 1)
#define SCAN_EPOCH_MAX (1u << 24)
#define SCAN_SUB 4              // tiles of k_scan_partial's size per ticket: a returning device-scope atomic every 8192 entries (they serialise at ~11 ns each)
DEVI u64 scan_word(u32 epoch, u32 state, u64 v) { return ((u64)epoch << (SCAN_VAL_BITS + 2)) | ((u64)state << SCAN_VAL_BITS) | (v & SCAN_VAL_MASK); }
// SCAN_SUB block-wide inclusive scans at once (one pair of barriers for all of them)
DEVI void block_scan_incl_sub(u64 v[SCAN_SUB], u64 (*sh_waves)[SCAN_BLOCK / 64], u64 tot[SCAN_SUB])
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < SCAN_SUB; q++) {
        for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(v[q], d); if (lane >= d) v[q] += t; }
        if (lane == 63) sh_waves[q][w] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SCAN_SUB; q++) {
        u64 add = 0, t = 0;
#pragma unroll
        for (int i = 0; i < SCAN_BLOCK / 64; i++) { const u64 x = sh_waves[q][i]; if (i < w) add += x; t += x; }
        v[q] += add; tot[q] = t;
    }
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_chain(const u32* in, u64 n, u64* out, u32* list, int nz, const u64* __restrict__ n_dev, u64* total,
                                                           unsigned int* ticket, u32 ticket_base, u64* status, u32 epoch)
{
    __shared__ u64 sh[SCAN_SUB][SCAN_BLOCK / 64];
    __shared__ u32 sh_tile;
    __shared__ u64 sh_prefix;
    if (n_dev) { const u64 nd = *n_dev; if (nd < n) n = nd; }
    if (threadIdx.x == 0) sh_tile = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    const u32 tile = sh_tile;
    const u64 base0 = (u64)tile * SCAN_SUB * SCAN_BLOCK * SCAN_ITEMS + (u64)threadIdx.x * SCAN_ITEMS;
    u32 x[SCAN_SUB][SCAN_ITEMS];
    u64 s[SCAN_SUB], incl[SCAN_SUB], tot[SCAN_SUB];
#pragma unroll
    for (int q = 0; q < SCAN_SUB; q++) {
        scan_load8(in, base0 + (u64)q * SCAN_BLOCK * SCAN_ITEMS, n, x[q]);
        if (nz) {
#pragma unroll
            for (int j = 0; j < SCAN_ITEMS; j++) x[q][j] = x[q][j] != 0;
        }
        s[q] = 0;
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) s[q] += x[q][j];
        incl[q] = s[q];
    }
    block_scan_incl_sub(incl, sh, tot);
    u64 all = 0;
#pragma unroll
    for (int q = 0; q < SCAN_SUB; q++) all += tot[q];
    if (threadIdx.x == 0)
        __hip_atomic_store(&status[tile], scan_word(epoch, tile ? 1u : 2u, all), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tile && threadIdx.x < 64) {
        const int lane = threadIdx.x;
        u64 prefix = 0;
        long top = (long)tile - 1;                                   // lane 0 looks at the nearest tile before this one
        for (;;) {
            const long idx = top - lane;
            const u64 w = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : scan_word(epoch, 2u, 0);   // before tile 0: an inclusive 0
            const u32 state = (u32)(w >> (SCAN_VAL_BITS + 2)) == epoch ? (u32)(w >> SCAN_VAL_BITS) & 3u : 0u;
            const u64 ready = __ballot(state != 0), closed = __ballot(state == 2);
            const int fi = closed ? __ffsll((long long)closed) - 1 : 63;      // the nearest word that carries an inclusive prefix, else the whole window
            const u64 need = fi == 63 ? ~0ull : ((2ull << fi) - 1);
            if ((ready & need) != need) { __builtin_amdgcn_s_sleep(2); continue; }
            u64 v = lane <= fi ? (w & SCAN_VAL_MASK) : 0;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            prefix += v;
            if (closed) break;
            top -= 64;
        }
        if (lane == 0) {
            sh_prefix = prefix;
            __hip_atomic_store(&status[tile], scan_word(epoch, 2u, prefix + all), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    u64 prefix = tile ? sh_prefix : 0;
    if (tile == gridDim.x - 1 && threadIdx.x == 0) *total = prefix + all;
#pragma unroll
    for (int q = 0; q < SCAN_SUB; q++) {
        const u64 base = base0 + (u64)q * SCAN_BLOCK * SCAN_ITEMS;
        u64 run = incl[q] - s[q] + prefix;
        prefix += tot[q];
        if (list) {
#pragma unroll
            for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n && x[q][j]) { list[run] = (u32)(base + j); run += x[q][j]; }
            continue;
        }
        if (base + SCAN_ITEMS <= n) {
            u64 o[SCAN_ITEMS];
#pragma unroll
            for (int j = 0; j < SCAN_ITEMS; j++) { o[j] = run; run += x[q][j]; }
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(out + base);
#pragma unroll
            for (int j = 0; j < SCAN_ITEMS; j += 2) dst[j >> 1] = make_ulonglong2(o[j], o[j + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n) { out[base + j] = run; run += x[q][j]; }
        }
        if (base <= n && n < base + SCAN_ITEMS) out[n] = run;
        if (n % ((u64)SCAN_BLOCK * SCAN_ITEMS) == 0 && n && base + SCAN_ITEMS == n) out[n] = run;
    }
}
