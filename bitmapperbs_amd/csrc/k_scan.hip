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
