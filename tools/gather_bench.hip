// tools/gather_bench.hip -- micro-benchmark: ceiling of dependent random 16-byte gathers on one MI355X
// (the access pattern of the FM-index seeding kernels: one Occ block or hash entry per step per lane).
//   hipcc -O3 --offload-arch=gfx950 -o gather_bench tools/gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32;

template <int ILP, int BYTES>
__global__ void __launch_bounds__(64) k_chase(const uint4* __restrict__ tab, u64 mask, int steps, u64* __restrict__ out)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 idx[ILP];
#pragma unroll
    for (int j = 0; j < ILP; j++) idx[j] = (gid * 0x9E3779B97F4A7C15ull + (u64)j * 0xD1B54A32D192ED03ull) & mask;
    u64 acc = 0;
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            u64 v;
            if (BYTES == 16) { const uint4 h = tab[idx[j]]; v = ((u64)h.y << 32 | h.x) ^ h.z ^ ((u64)h.w << 13); }
            else if (BYTES == 8) { v = reinterpret_cast<const u64*>(tab)[idx[j] * 2]; }
            else { v = reinterpret_cast<const u32*>(tab)[idx[j] * 4]; }
            acc += v;
            idx[j] = (v * 0x9E3779B97F4A7C15ull + idx[j] + 1) & mask;       // dependent on the loaded value
        }
    }
    out[gid] = acc + idx[0];
}

// one random 16-byte gather + H 8-byte loads from the lane's own 160-byte row per step (the seeding kernels read the
// next read characters that way): how much do the cache-hitting per-lane loads cost next to the missing gather?
template <int H>
__global__ void __launch_bounds__(64) k_mixed(const uint4* __restrict__ tab, u64 mask, const char* __restrict__ rows, int steps, u64* __restrict__ out)
{
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 idx = (gid * 0x9E3779B97F4A7C15ull) & mask;
    const char* row = rows + gid * 160;
    u64 acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4 h = tab[idx];
        u64 v = ((u64)h.y << 32 | h.x) ^ h.z ^ ((u64)h.w << 13);
#pragma unroll
        for (int j = 0; j < H; j++) v += *reinterpret_cast<const u64*>(row + (((v >> 7) + j * 5) % 19) * 8);
        acc += v;
        idx = (v * 0x9E3779B97F4A7C15ull + idx + 1) & mask;
    }
    out[gid] = acc + idx;
}
template <int H>
void run_mixed(const uint4* tab, u64 entries, const char* rows, u64* out, long lanes, int steps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_mixed<H><<<dim3((unsigned)(lanes / 64)), dim3(64)>>>(tab, entries - 1, rows, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_mixed<H><<<dim3((unsigned)(lanes / 64)), dim3(64)>>>(tab, entries - 1, rows, steps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("1 gather + %d row loads/step   table %6.0f MB  lanes %9ld : %7.3f ms  %6.1f G gathers/s\n", H, entries * 16.0 / 1e6, lanes, ms, (double)lanes * steps / ms / 1e6);
}

template <int ILP, int BYTES>
void run(const uint4* tab, u64 entries, u64* out, long lanes, int steps, const char* label)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_chase<ILP, BYTES><<<dim3((unsigned)(lanes / 64)), dim3(64)>>>(tab, entries - 1, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_chase<ILP, BYTES><<<dim3((unsigned)(lanes / 64)), dim3(64)>>>(tab, entries - 1, steps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double acc = (double)lanes * ILP * steps;
    printf("%-28s table %6.0f MB  lanes %9ld  ILP %d  %2d B : %7.3f ms  %6.1f G accesses/s\n", label, entries * 16.0 / 1e6, lanes, ILP, BYTES, ms, acc / ms / 1e6);
}

__global__ void k_fill(u32* p, u64 n) { for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = (u32)(i * 2654435761ull >> 7) ^ (u32)(i >> 13) * 40503u; }

int main(int argc, char** argv)
{
    if (argc > 1) {
        // `gather_bench big`: a table of the size of the 20-mer outcome table (2^31 x 16 B = 34 GB): does the reach of the TLBs show?
        const u64 entries = 1ull << 31;
        uint4* tab; u64* out;
        if (hipMalloc(&tab, entries * 16) != hipSuccess) { printf("cannot allocate 34 GB\n"); return 1; }
        k_fill<<<dim3(65536), dim3(256)>>>((u32*)tab, entries * 4);
        hipDeviceSynchronize();
        const long lanes = 1 << 22;
        hipMalloc(&out, lanes * 8 * 4);
        run<1, 16>(tab, entries, out, lanes, 32, "dependent chain");
        run<1, 4>(tab, entries, out, lanes, 32, "dependent chain");
        run<2, 16>(tab, entries, out, lanes, 32, "2 chains per lane");
        run<1, 16>(tab, entries, out, lanes * 4, 16, "4x lanes");
        return 0;
    }
    for (u64 entries : {1ull << 22, 1ull << 25, 1ull << 28}) {         // 64 MB (fits Infinity Cache), 512 MB, 4 GB
        uint4* tab; u64* out;
        hipMalloc(&tab, entries * 16);
        std::vector<u32> h(1 << 20);
        for (auto& x : h) x = (u32)rand() * 2654435761u;
        for (u64 o = 0; o < entries * 16; o += (u64)h.size() * 4) hipMemcpy((char*)tab + o, h.data(), std::min<u64>((u64)h.size() * 4, entries * 16 - o), hipMemcpyHostToDevice);
        const long lanes = 1 << 22;
        hipMalloc(&out, lanes * 8 * 4);
        run<1, 16>(tab, entries, out, lanes, 32, "dependent chain");
        run<2, 16>(tab, entries, out, lanes, 32, "2 chains per lane");
        run<4, 16>(tab, entries, out, lanes, 16, "4 chains per lane");
        run<1, 4>(tab, entries, out, lanes, 32, "dependent chain");
        run<4, 4>(tab, entries, out, lanes, 16, "4 chains per lane");
        run<1, 16>(tab, entries, out, lanes * 4, 16, "4x lanes");
        if (entries == (1ull << 25)) {
            char* rows; hipMalloc(&rows, (size_t)lanes * 160);
            hipMemset(rows, 7, (size_t)lanes * 160);
            run_mixed<0>(tab, entries, rows, out, lanes, 32);
            run_mixed<1>(tab, entries, rows, out, lanes, 32);
            run_mixed<2>(tab, entries, rows, out, lanes, 32);
            run_mixed<4>(tab, entries, rows, out, lanes, 32);
            hipFree(rows);
        }
        hipFree(tab); hipFree(out);
    }
    return 0;
}
