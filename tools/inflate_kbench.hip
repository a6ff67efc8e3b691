// tools/inflate_kbench.hip -- k_bgzf_inflate ALONE, built from the library's own source file in seconds (the library takes 90 s):
// FASTQ-like text -> BGZF blocks (zlib, level 1 and 6) -> the kernel, timed with HIP events, one window and two windows side by
// side on two streams (what a pair's files look like to the device); the text that comes back is compared with what went in.
// build (here):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/inflate_kbench tools/inflate_kbench.hip -lz
// run (GPU box): ./tools/inflate_kbench [MB=160]
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../bitmapperbs_amd/csrc/bmbs_dev.h"
#include "../bitmapperbs_amd/csrc/bmbs_inflate.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static std::string fastq_text(size_t mb, int L, unsigned seed)
{
    std::string t; t.reserve(mb * 1000000 + 1000);
    unsigned long long s = seed * 0x9e3779b97f4a7c15ull + 1;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const char* q4 = "F:,#";
    for (size_t i = 0; t.size() < mb * 1000000; i++) {
        char nm[64]; const int nl = snprintf(nm, sizeof nm, "@sim.%zu/1\n", i);
        t.append(nm, (size_t)nl);
        for (int j = 0; j < L; j++) { const unsigned r = (unsigned)(rnd() >> 33) & 3u; t.push_back("ACGT"[r == 1 && (rnd() & 127) ? 3 : r]); }   // (bisulfite: few C)
        t += "\n+\n";
        for (int j = 0; j < L;) { int run = 5 * (1 + (int)((rnd() >> 40) % 3)); const unsigned r = (unsigned)(rnd() >> 50) % 10; const char c = q4[r < 7 ? 0 : r - 6]; while (run-- && j < L) { t.push_back(c); j++; } }
        t.push_back('\n');
    }
    return t;
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 160;
    const std::string text = fastq_text(mb, 150, 5);
    CK(hipSetDevice(0));
    {
        u32 x2n[32];
        auto mult = [](u32 a, u32 b) { u32 m = 1u << 31, p = 0; for (;;) { if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; } m >>= 1; b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1; } return p; };
        u32 p = 1u << 30; x2n[0] = p;
        for (int n = 1; n < 32; n++) { p = mult(p, p); x2n[n] = p; }
        CK(hipMemcpyToSymbol(HIP_SYMBOL(c_x2n), x2n, sizeof x2n));
    }
    for (int level : {1, 6}) {
        std::vector<unsigned char> comp; std::vector<u64> blk(1, 0), out(1, 0);
        std::vector<unsigned char> tmp(compressBound(65280) + 64);
        for (size_t a = 0; a < text.size(); a += 65280) {
            const size_t n = std::min<size_t>(65280, text.size() - a);
            z_stream zs; memset(&zs, 0, sizeof zs);
            deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            zs.next_in = (Bytef*)text.data() + a; zs.avail_in = (uInt)n; zs.next_out = tmp.data(); zs.avail_out = (uInt)tmp.size();
            deflate(&zs, Z_FINISH);
            const size_t cl = tmp.size() - zs.avail_out;
            deflateEnd(&zs);
            const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            comp.insert(comp.end(), hdr, hdr + 16);
            const unsigned bs = (unsigned)(cl + 25);
            comp.push_back((unsigned char)bs); comp.push_back((unsigned char)(bs >> 8));
            comp.insert(comp.end(), tmp.begin(), tmp.begin() + (long)cl);
            const u32 crc = (u32)crc32(0, (const Bytef*)text.data() + a, (uInt)n), isz = (u32)n;
            for (int k = 0; k < 4; k++) comp.push_back((unsigned char)(crc >> (8 * k)));
            for (int k = 0; k < 4; k++) comp.push_back((unsigned char)(isz >> (8 * k)));
            blk.push_back(comp.size()); out.push_back(a + n);
        }
        const long nb = (long)blk.size() - 1;
        unsigned char* d_comp; u64 *d_blk, *d_out; char* d_text[2]; u32* d_err[2];
        CK(hipMalloc(&d_comp, comp.size() + 4096)); CK(hipMalloc(&d_blk, blk.size() * 8)); CK(hipMalloc(&d_out, out.size() * 8));
        CK(hipMemcpy(d_comp, comp.data(), comp.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_blk, blk.data(), blk.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_out, out.data(), out.size() * 8, hipMemcpyHostToDevice));
        hipStream_t st[2]; hipEvent_t ea[2], eb[2];
        for (int f = 0; f < 2; f++) { CK(hipMalloc(&d_text[f], text.size() + 4096)); CK(hipMalloc(&d_err[f], (size_t)nb * 4)); CK(hipStreamCreate(&st[f])); CK(hipEventCreate(&ea[f])); CK(hipEventCreate(&eb[f])); }
        auto launch = [&](int f) { hipLaunchKernelGGL(k_bgzf_inflate, dim3((unsigned)nb), dim3(64), 0, st[f], d_comp, d_blk, d_out, nb, d_text[f], d_err[f]); };
        launch(0); CK(hipStreamSynchronize(st[0]));
        std::vector<char> back(text.size()); std::vector<u32> err((size_t)nb);
        CK(hipMemcpy(back.data(), d_text[0], text.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(err.data(), d_err[0], (size_t)nb * 4, hipMemcpyDeviceToHost));
        long bad = 0; for (u32 e : err) bad += e != 0;
        const bool same = memcmp(back.data(), text.data(), text.size()) == 0;
        float one = 0, two = 0;
        const int R = 5;
        CK(hipEventRecord(ea[0], st[0])); for (int r = 0; r < R; r++) launch(0); CK(hipEventRecord(eb[0], st[0])); CK(hipStreamSynchronize(st[0]));
        CK(hipEventElapsedTime(&one, ea[0], eb[0]));
        for (int f = 0; f < 2; f++) CK(hipEventRecord(ea[f], st[f]));
        for (int r = 0; r < R; r++) for (int f = 0; f < 2; f++) launch(f);
        for (int f = 0; f < 2; f++) CK(hipEventRecord(eb[f], st[f]));
        for (int f = 0; f < 2; f++) CK(hipStreamSynchronize(st[f]));
        float t0 = 0, t1 = 0; CK(hipEventElapsedTime(&t0, ea[0], eb[0])); CK(hipEventElapsedTime(&t1, ea[1], eb[1])); two = t0 > t1 ? t0 : t1;
        printf("level %d: %ld blocks, %.1f MB -> %.1f MB  text %s, %ld blocks refused;  alone %.2f ms per window;  two windows side by side %.2f ms per pair (%.2f per window)\n",
               level, nb, comp.size() / 1e6, text.size() / 1e6, same ? "identical" : "DIFFERS", bad, one / R, two / R, two / R / 2);
        for (int f = 0; f < 2; f++) { (void)hipFree(d_text[f]); (void)hipFree(d_err[f]); }
        (void)hipFree(d_comp); (void)hipFree(d_blk); (void)hipFree(d_out);
    }
    return 0;
}
