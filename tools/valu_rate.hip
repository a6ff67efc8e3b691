// tools/valu_rate.hip -- issue rate of the VALU instructions the Myers filter's row is made of (gfx950): every wave runs a long chain
// of one instruction, eight waves per SIMD, all CUs; prints wave-instructions per nanosecond per SIMD (full rate at 2.4 GHz: 0.6)
// build (here): hipcc -O3 --offload-arch=gfx950 -o tools/valu_rate tools/valu_rate.hip      run (GPU box): ./tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int OP>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters)
{
    unsigned a = threadIdx.x * 2654435761u, b = blockIdx.x * 40503u + 1, c = a ^ b, d = b + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            if (OP == 0) { a = a & b; b = b ^ c; c = c | d; d = d + a; }
            if (OP == 1) { a = __builtin_amdgcn_bitop3_b32(a, b, c, 0xcb); b = __builtin_amdgcn_bitop3_b32(b, c, d, 0x82); c = __builtin_amdgcn_bitop3_b32(c, d, a, 0xbe); d = __builtin_amdgcn_bitop3_b32(d, a, b, 0xf1); }
            if (OP == 2) { a = __builtin_amdgcn_alignbit(a, b, 3); b = __builtin_amdgcn_alignbit(b, c, 5); c = __builtin_amdgcn_alignbit(c, d, 7); d = __builtin_amdgcn_alignbit(d, a, 9); }
            if (OP == 3) { a = (unsigned)__builtin_amdgcn_sbfe((int)b, 3, 1) ^ a; b = (unsigned)__builtin_amdgcn_sbfe((int)c, 5, 1) ^ b; c = (unsigned)__builtin_amdgcn_sbfe((int)d, 7, 1) ^ c; d = (unsigned)__builtin_amdgcn_sbfe((int)a, 9, 1) ^ d; }
            if (OP == 4) { a = a + (b & c); b = b + (c & d); c = c + (d & a); d = d + (a & b); }
        }
    }
    if ((a ^ b ^ c ^ d) == 0x12345u) out[0] = a;
}
template <int OP> int run(const char* name, int per_group, unsigned* out)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 4 * iters * 32 * per_group;
    printf("%-28s %8.3f ms  %.3f wave-instructions / ns / SIMD\n", name, ms, wave_instr / (ms * 1e6) / 1024.0);
    return 0;
}
int main()
{
    unsigned* out; CK(hipMalloc(&out, 64));
    run<0>("and/xor/or/add", 4, out);
    run<1>("v_bitop3_b32", 4, out);
    run<2>("v_alignbit_b32", 4, out);
    run<3>("v_bfe_i32 + xor", 8, out);
    run<4>("and + add", 8, out);
    return 0;
}
