// bitmapperbs_amd/csrc/bmbs_bytes.h -- byte-level helpers shared by the text-path kernels (bmbs_text.hip, bmbs_bam.hip, bmbs_inflate.hip)
// and by tools/inflate_kbench.hip, which builds the inflate kernel alone
#ifndef BMBS_BYTES_H
#define BMBS_BYTES_H
#ifndef DEVI
#define DEVI __device__ __forceinline__
#endif
// bit 8j+7 set where byte j of w is '\n'
DEVI u32 nl_mask4(u32 w)
{
    const u32 x = w ^ 0x0a0a0a0au;
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
}

__constant__ u32 c_x2n[32];                 // x^(2^n) mod P of CRC-32 (reflected), n = 0..31: set by bgzf_init_constants

#define CRC_POLY 0xedb88320u
DEVI u32 crc_multmodp(u32 a, u32 b)
{
    u32 m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
// x^(8 n) mod P
DEVI u32 crc_x8n(u32 n)
{
    u32 p = 1u << 31; int k = 3;
    while (n) { if (n & 1) p = crc_multmodp(c_x2n[k & 31], p); n >>= 1; k++; }
    return p;
}

#endif
