// tools/valu_bench.hip -- issue cost of the VALU instructions the DP / bit-vector kernels are made of, on gfx950: shader cycles
// per instruction for ONE wave alone (dependent chain and 8 independent chains) and the SIMD-level rate with 1 / 3 / 8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o valu_bench valu_bench.hip && ./valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP 64
#define ITERS 1024
// OP(d, a, b): one instruction, d = a op b
#define KERNEL(NAME, ASM, CONSTR)                                                                                           \
    __global__ void __launch_bounds__(64) dep_##NAME(unsigned long long* out, unsigned seed)                                \
    {                                                                                                                       \
        unsigned x = threadIdx.x + seed, y = seed | 1;                                                                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                         \
        for (int it = 0; it < ITERS; it++) {                                                                                \
            _Pragma("unroll") for (int r = 0; r < REP; r++) asm volatile(ASM : "+v"(x) : CONSTR(y));                        \
        }                                                                                                                   \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                         \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                    \
        if (x == 0x12345) out[0] = x;                                                                                       \
    }                                                                                                                       \
    __global__ void __launch_bounds__(64) ind_##NAME(unsigned long long* out, unsigned seed)                                \
    {                                                                                                                       \
        unsigned x[8], y = seed | 1;                                                                                        \
        for (int q = 0; q < 8; q++) x[q] = threadIdx.x + seed + q;                                                          \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                         \
        for (int it = 0; it < ITERS; it++) {                                                                                \
            _Pragma("unroll") for (int r = 0; r < REP; r++) asm volatile(ASM : "+v"(x[r & 7]) : CONSTR(y));                 \
        }                                                                                                                   \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                         \
        unsigned s = 0; for (int q = 0; q < 8; q++) s ^= x[q];                                                              \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                    \
        if (s == 0x12345) out[0] = s;                                                                                       \
    }
#define V "v"
#define S "s"
KERNEL(add_u32, "v_add_u32 %0, %0, %1", V)
KERNEL(and_b32, "v_and_b32 %0, %0, %1", V)
KERNEL(max_i32, "v_max_i32 %0, %0, %1", V)
KERNEL(pk_add_u16, "v_pk_add_u16 %0, %0, %1", V)
KERNEL(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1", V)
KERNEL(pk_max_i16, "v_pk_max_i16 %0, %0, %1", V)
KERNEL(pk_add_f16, "v_pk_add_f16 %0, %0, %1", V)
KERNEL(lshrrev_b32, "v_lshrrev_b32 %0, 1, %0", V)
KERNEL(bfe_i32, "v_bfe_i32 %0, %0, 4, 1", V)
KERNEL(and_or_b32, "v_and_or_b32 %0, %0, %1, %1", V)
KERNEL(add3_u32, "v_add3_u32 %0, %0, %1, %1", V)
KERNEL(bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96", V)
KERNEL(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", V)
KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc", V)
KERNEL(cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", V)
KERNEL(cmp_cndmask, "v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", V)
KERNEL(add_sgpr, "v_add_u32 %0, s20, %0", V)
KERNEL(add_co, "v_add_co_u32 %0, vcc, %0, %1", V)
KERNEL(addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc", V)
KERNEL(xor_b32, "v_xor_b32 %0, %0, %1", V)
KERNEL(or3_b32, "v_or3_b32 %0, %0, %1, %1", V)
KERNEL(sdwa, "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1", V)
KERNEL(readlane_w, "v_readlane_b32 s20, %0, 3\n v_writelane_b32 %0, s20, 5", V)
KERNEL(mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", V)
// 64-bit
#define KERNEL64(NAME, ASM)                                                                                                 \
    __global__ void __launch_bounds__(64) dep_##NAME(unsigned long long* out, unsigned seed)                                \
    {                                                                                                                       \
        unsigned long long x = threadIdx.x + seed, y = seed | 1;                                                            \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                         \
        for (int it = 0; it < ITERS; it++) {                                                                                \
            _Pragma("unroll") for (int r = 0; r < REP; r++) asm volatile(ASM : "+v"(x) : "v"(y));                           \
        }                                                                                                                   \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                         \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                    \
        if (x == 0x12345) out[0] = x;                                                                                       \
    }                                                                                                                       \
    __global__ void __launch_bounds__(64) ind_##NAME(unsigned long long* out, unsigned seed)                                \
    {                                                                                                                       \
        unsigned long long x[8], y = seed | 1;                                                                              \
        for (int q = 0; q < 8; q++) x[q] = threadIdx.x + seed + q;                                                          \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                         \
        for (int it = 0; it < ITERS; it++) {                                                                                \
            _Pragma("unroll") for (int r = 0; r < REP; r++) asm volatile(ASM : "+v"(x[r & 7]) : "v"(y));                    \
        }                                                                                                                   \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                         \
        unsigned long long s = 0; for (int q = 0; q < 8; q++) s ^= x[q];                                                    \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                    \
        if (s == 0x12345) out[0] = s;                                                                                       \
    }
KERNEL64(lshrrev_b64, "v_lshrrev_b64 %0, 1, %0")
KERNEL64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL64(mov_b64, "v_mov_b64 %0, %1")
KERNEL64(cmp_eq_u64, "v_cmp_eq_u64 vcc, %0, %1")
KERNEL64(pk_add_f32, "v_pk_add_f32 %0, %0, %1")

typedef void (*kfn)(unsigned long long*, unsigned);
struct Ent { const char* name; kfn dep, ind; };
#define E(N) {#N, dep_##N, ind_##N}
int main()
{
    Ent ents[] = {E(add_u32), E(and_b32), E(max_i32), E(pk_add_u16), E(pk_sub_i16), E(pk_max_i16), E(pk_add_f16), E(lshrrev_b32), E(bfe_i32), E(and_or_b32),
                  E(add3_u32), E(bitop3), E(mul_lo_u32), E(cndmask), E(cndmask_e64), E(cmp_cndmask), E(add_sgpr), E(add_co), E(addc_co), E(xor_b32), E(or3_b32), E(sdwa), E(readlane_w), E(mov_dpp), E(lshrrev_b64), E(lshl_add_u64), E(mov_b64), E(cmp_eq_u64), E(pk_add_f32)};
    unsigned long long* d; hipMalloc(&d, 8 * 65536);
    std::vector<unsigned long long> h(65536);
    const double n = (double)REP * ITERS;
    printf("%-14s %8s %8s | SIMD cycles per instruction with w waves/SIMD (independent chains): %6s %6s %6s\n", "op", "dep", "ind8", "w=1", "w=3", "w=8");
    for (auto& e : ents) {
        double r[5];
        int cfg[5][2] = {{1024, 0}, {1024, 1}, {1024, 1}, {3072, 1}, {8192, 1}};
        for (int c = 0; c < 5; c++) {
            const int blocks = cfg[c][0];
            for (int w = 0; w < 2; w++) hipLaunchKernelGGL(cfg[c][1] ? e.ind : e.dep, dim3(blocks), dim3(64), 0, 0, d, 7u);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + blocks);
            const double med = (double)h[blocks / 2];
            r[c] = med / n / (blocks / 1024.0);          // wave cycles / instr / waves sharing the SIMD = SIMD cycles per instruction
            if (c < 2) r[c] = med / n;
        }
        hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
        hipEventRecord(ea, 0);
        for (int w = 0; w < 20; w++) hipLaunchKernelGGL(e.ind, dim3(8192), dim3(64), 0, 0, d, 7u);
        hipEventRecord(eb, 0); hipEventSynchronize(eb);
        float ms = 0; hipEventElapsedTime(&ms, ea, eb);
        const double tps = 20.0 * 8192 * n / (ms * 1e-3) / 1e12;
        // dependent chains, 1 / 3 / 8 waves per SIMD: SIMD cycles per instruction from the wall clock (2.4 GHz assumed)
        double depc[3];
        const int nb[3] = {1024, 3072, 8192};
        for (int q = 0; q < 3; q++) {
            hipEventRecord(ea, 0);
            for (int w = 0; w < 20; w++) hipLaunchKernelGGL(e.dep, dim3(nb[q]), dim3(64), 0, 0, d, 7u);
            hipEventRecord(eb, 0); hipEventSynchronize(eb);
            hipEventElapsedTime(&ms, ea, eb);
            depc[q] = (ms * 1e-3 / 20.0) * 2.4e9 / (n * nb[q] / 1024.0);
        }
        printf("%-14s %8.2f %8.2f | %6.2f %6.2f %6.2f   chip %.3f T wave-instr/s | dependent chains, SIMD cycles/instr at w=1,3,8: %5.2f %5.2f %5.2f\n", e.name, r[0], r[1], r[2], r[3], r[4], tps, depc[0], depc[1], depc[2]);
    }
    return 0;
}
