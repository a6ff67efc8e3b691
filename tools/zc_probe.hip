// tools/zc_probe.hip -- on the GPU box: does a KERNEL that stores straight into page-locked host memory move bytes faster than the
// copy engine's device-to-host hipMemcpyAsync?  (Some boxes of the pool copy D2H at 26-33 GB/s and H2D at 55.)  16 bytes per lane,
// consecutive lanes consecutive addresses; grids of 64 .. 4096 workgroups; both kinds of host memory (hipHostMalloc, hipHostRegister).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/zc_probe tools/zc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// one workgroup moves whole 4 KiB pieces (what k_line_write's store phase looks like: a block writes its contiguous piece)
__global__ void __launch_bounds__(256) k_copy_piece(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    const size_t per = 1024;        // 16 KiB per block step
    for (size_t b = (size_t)blockIdx.x * per; b < n16; b += (size_t)gridDim.x * per)
        for (size_t i = threadIdx.x; i < per && b + i < n16; i += 256) dst[b + i] = src[b + i];
}

static float run(void (*k)(const uint4*, uint4*, size_t), int grid, const void* s, void* d, size_t bytes, hipStream_t st)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, st, (const uint4*)s, (uint4*)d, bytes / 16);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, st, (const uint4*)s, (uint4*)d, bytes / 16);
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return (float)(3.0 * bytes / 1e6 / ms);
}

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 384) << 20;
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    void* dev; CK(hipMalloc(&dev, bytes)); CK(hipMemset(dev, 1, bytes));
    void* hm; CK(hipHostMalloc(&hm, bytes, hipHostMallocDefault)); memset(hm, 2, bytes);
    void* hr = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); memset(hr, 3, bytes);
    CK(hipHostRegister(hr, bytes, hipHostRegisterDefault));
    void* hr_dev = nullptr; CK(hipHostGetDevicePointer(&hr_dev, hr, 0));
    void* hm_dev = nullptr; CK(hipHostGetDevicePointer(&hm_dev, hm, 0));
    printf("bytes %zu MiB; hipHostMalloc dev ptr %s host ptr, hipHostRegister dev ptr %s host ptr\n", bytes >> 20, hm_dev == hm ? "==" : "!=", hr_dev == hr ? "==" : "!=");
    // copy engine
    for (int which = 0; which < 2; which++) {
        void* h = which ? hr : hm;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int dir = 0; dir < 2; dir++) {
            CK(hipMemcpyAsync(dir ? dev : h, dir ? h : dev, bytes, dir ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(a, st));
            for (int r = 0; r < 3; r++)
                for (size_t o = 0; o < bytes; o += (size_t)128 << 20) {
                    const size_t m = bytes - o < ((size_t)128 << 20) ? bytes - o : (size_t)128 << 20;
                    CK(hipMemcpyAsync((char*)(dir ? dev : h) + o, (char*)(dir ? h : dev) + o, m, dir ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
                }
            CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            printf("hipMemcpyAsync %s %-16s %6.1f GB/s\n", dir ? "H2D" : "D2H", which ? "hipHostRegister" : "hipHostMalloc", 3.0 * bytes / 1e6 / ms);
        }
    }
    for (int which = 0; which < 2; which++) {
        void* h = which ? hr_dev : hm_dev;
        for (int grid : {64, 256, 1024, 4096}) {
            printf("kernel D2H (stores to host) %-16s grid %5d: flat %6.1f GB/s  pieces %6.1f GB/s\n", which ? "hipHostRegister" : "hipHostMalloc", grid,
                   run(k_copy16, grid, dev, h, bytes, st), run(k_copy_piece, grid, dev, h, bytes, st));
            printf("kernel H2D (loads from host) %-16s grid %5d: flat %6.1f GB/s\n", which ? "hipHostRegister" : "hipHostMalloc", grid, run(k_copy16, grid, h, dev, bytes, st));
        }
    }
    // the stores really landed
    hipLaunchKernelGGL(k_copy16, dim3(1024), dim3(256), 0, st, (const uint4*)dev, (uint4*)hr_dev, bytes / 16); CK(hipStreamSynchronize(st));
    printf("check: host byte %d (expect 1)\n", ((unsigned char*)hr)[bytes - 1]);
    return 0;
}
