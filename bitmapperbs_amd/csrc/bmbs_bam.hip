// bitmapperbs_amd/csrc/bmbs_bam.hip -- `--bam` on the device: records -> BAM records -> BGZF blocks (round 4).
//
// The reference hands every SAM line to htslib (sam_parse1 + bam_write1 over a BGZF stream, bam_prase.cpp:201-221) on its output
// thread; round 3 of this tree re-parsed the device-made SAM text on the host's I/O threads and deflated it with zlib.  Here the
// host never sees a record: the BAM record of every output line is built from bmbs_result + the resident FASTQ text
// (k_bam_len / k_bam_write, the same line logic as the SAM text: sam_line), the record stream of the batch is cut into BGZF
// blocks of 0xff00 input bytes, and every block is deflated by one workgroup (k_bgzf_block): run-length matches, a dynamic
// Huffman code built from the block's own symbol counts (what compresses BAM: 4-bit base pairs and a small quality alphabet),
// CRC-32 by segments combined with x^(8n) mod P.  The blocks of a batch come back as one contiguous piece of a BGZF file.
// Parity surface: the INFLATED record stream equals the reference's (tests/common.bam_payload); block boundaries and compressed
// bytes are this design's own.
#ifndef BMBS_BAM_HIP
#define BMBS_BAM_HIP

#define BMBS_TEXT_BAM 16          // bmbs_map_*_text: `sam` receives BGZF-compressed BAM records instead of SAM text

// htslib's seq_nt16_table ("=ACMGRSVTWYHKDBN"; digits 0-3 count as ACGT), two entries per byte
__constant__ u8 c_nt16[256] = {
    15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
    15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 1,2,4,8,15,15,15,15,15,15,15,15,15,0,15,15,
    15,1,14,2,13,15,15,4,11,15,15,12,15,3,15,15, 15,15,5,6,8,15,7,9,15,10,15,15,15,15,15,15,
    15,1,14,2,13,15,15,4,11,15,15,12,15,3,15,15, 15,15,5,6,8,15,7,9,15,10,15,15,15,15,15,15,
    15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
    15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
    15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
    15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15};

// hts_reg2bin(beg, end, 14, 5) as sam_parse1 calls it (end exclusive)
DEVI int bam_reg2bin(long long beg, long long end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
DEVI void st32(char* p, u32 v) { p[0] = (char)v; p[1] = (char)(v >> 8); p[2] = (char)(v >> 16); p[3] = (char)(v >> 24); }
DEVI void st16(char* p, u32 v) { p[0] = (char)v; p[1] = (char)(v >> 8); }

// bytes of the NM tag of a mapped record: "NM" + the smallest unsigned type that holds the value (sam_parse1's choice)
DEVI int bam_tag_len(const SamLine& s) { return s.kind == 1 ? (s.x->nm <= 0xff ? 4 : 5) : 0; }
DEVI int bam_ncigar(const SamLine& s) { return s.kind == 1 ? (s.x->n_cigar ? (int)s.x->n_cigar : 1) : 0; }

// the fixed 36 bytes (block_size + the 32-byte core) of line s into p, its CIGAR words into c and its tag behind the qualities
// into g (the three sit apart in the record: core, QNAME, CIGAR, SEQ, QUAL, tag)
DEVI void bam_head(const SamIn& in, const SamLine& s, int L, int total, char* p, char* c, char* g)
{
    const bool pe = (in.flags & BMBS_TEXT_PE) != 0;
    const int ncig = bam_ncigar(s);
    st32(p, (u32)(total - 4));
    if (s.kind == 2) {
        const int flag = !pe ? 4 : s.mate == 0 ? 77 : 141;
        st32(p + 4, 0xffffffffu); st32(p + 8, 0xffffffffu);
        p[12] = (char)(s.name_len + 1); p[13] = 0; st16(p + 14, (u32)bam_reg2bin(-1, 0));
        st16(p + 16, 0); st16(p + 18, (u32)flag); st32(p + 20, (u32)L);
        st32(p + 24, 0xffffffffu); st32(p + 28, 0xffffffffu); st32(p + 32, 0);
        return;
    }
    const bmbs_result_dev& x = *s.x;
    const long long pos0 = (long long)x.pos - 1;
    long long reflen = 0;
    if (x.n_cigar == 0) { st32(c, (u32)L << 4); reflen = L; c += 4; }
    else
        for (int i = 0; i < x.n_cigar; i++) {
            const u32 o = in.cigar[x.cigar_off + i];
            const u32 k = o & 7u;                                             // ours: 0 M, 1 D, 2 I, 3 S, 4 H  ->  BAM: M 0, I 1, D 2, S 4, H 5
            const u32 bop = k == 0 ? 0u : k == 1 ? 2u : k == 2 ? 1u : k == 3 ? 4u : 5u;
            if (k <= 1) reflen += (long long)(o >> 4);
            st32(c, (o & ~15u) | bop); c += 4;
        }
    st32(p + 4, (u32)x.chrom); st32(p + 8, (u32)pos0);
    p[12] = (char)(s.name_len + 1); p[13] = (char)x.mapq; st16(p + 14, (u32)bam_reg2bin(pos0, pos0 + (reflen == 0 ? 1 : reflen)));
    st16(p + 16, (u32)ncig); st16(p + 18, (u32)x.flag); st32(p + 20, (u32)L);
    if (!pe) { st32(p + 24, 0xffffffffu); st32(p + 28, 0xffffffffu); st32(p + 32, 0); }
    else {
        const bmbs_result_dev& y = *s.mate_x;
        const bool neg = s.mate == 0 ? (y.pos < x.pos) : !(y.pos > x.pos);      // TLEN sign as in sam_head
        const u32 tl = s.mate == 0 ? x.tlen : y.tlen;
        st32(p + 24, (u32)x.chrom); st32(p + 28, (u32)((long long)y.pos - 1)); st32(p + 32, neg ? (u32)(-(int)tl) : tl);
    }
    g[0] = 'N'; g[1] = 'M';
    if (x.nm <= 0xff) { g[2] = 'C'; g[3] = (char)x.nm; }
    else { g[2] = 'S'; st16(g + 3, x.nm); }
}

// bytes of the BAM record of output line `line` (0: nothing is printed for it); info[3] = 1 + a line whose QNAME is too long
__global__ void __launch_bounds__(256)
k_bam_len(SamIn in, long n_lines, u32* __restrict__ len_out, u32* __restrict__ info)
{
    const long line = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= n_lines) return;
    const SamLine s = sam_line(in, line);
    u32 len = 0;
    if (s.kind) {
        const int L = in.rec[s.mate].seq_len[s.rec];
        len = (u32)(36 + s.name_len + 1 + 4 * bam_ncigar(s) + (L + 1) / 2 + L + bam_tag_len(s));
        if (s.name_len > 254) atomicMax(&info[3], (u32)line + 1u);         // l_read_name is one byte (htslib refuses such a line)
    }
    len_out[line] = len;
}

// ---- records -> SAM text or BAM records: one workgroup per piece of the output (round 5) ------------------------------------------------
// A workgroup of 256 threads owns `lpb` consecutive output lines = one contiguous piece of the output, and their sources are
// contiguous too: the records of consecutive lines follow each other in the resident FASTQ text (pairs: in either mate's text).
//   1. the source range(s) come into LDS with 16-byte loads, the records' fields into registers (lane i: line i);
//   2. lane i finds its QNAME in the staged text and renders its numeric columns (SAM) / core, CIGAR words and tag (BAM)
//      straight into the line's place in the LDS image of the piece;
//   3. 16 lanes per line move QNAME / SEQ / QUAL from the staged text into the image: upper case, complement, reversal, padding
//      (Process_Reads.cpp:836, 1603), 4-bit bases and qualities - 33 for BAM;
//   4. the image goes out with 16-byte stores (its first and last 16-byte unit byte-wise: the neighbouring pieces own the rest).
// The round-3/4 form built every output dword from four single-byte global loads, one wave per workgroup: 4.5 ms per 1.05 M lines
// (160 GB/s); this one moves the same bytes at the rate of its loads and stores.
// A piece or a source range that does not fit its LDS buffer (names of kilobytes, a length estimate that was off) takes the plain
// path: a wave per line, bytes straight from and to memory.
#define TXW_THREADS 256
struct TxwDesc {                 // per line, in LDS
    u32 start;                   // offset of the line inside the piece
    u32 total;                   // bytes (0: the line is not printed)
    u16 name_rel, seq_rel, qual_rel;   // where QNAME / SEQ / QUAL start in the staged text (QNAME: file 0; SEQ / QUAL: the line's own file)
    u16 nlen, hl, tl, L, qn;
    u8 rc, mate;
};

// printed SEQ character: upper case (Process_Reads.cpp:836), complemented when the line prints the other strand (rc_table, :1603)
DEVI u32 txw_base(u32 c, bool rc)
{
    if (c >= 'a' && c <= 'z') c -= 32;
    if (rc) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
    return c;
}

template <bool BAM>
DEVI int txw_line_bytes(const SamLine& s, int L, int hl, int tl) { return BAM ? 36 + s.name_len + 1 + hl + (L + 1) / 2 + L + tl : s.name_len + hl + 2 * L + 1 + tl; }

template <bool BAM>
__global__ void __launch_bounds__(TXW_THREADS)
k_line_write(SamIn in, long n_lines, const u64* __restrict__ off, int lpb, u32 out_cap, u32 src_cap, char* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char lds_txw[];        // [out_cap + 16] image of the piece, [src_cap + 16] x files: staged text
    __shared__ TxwDesc s_d[64];
    char* const img = lds_txw;
    char* const src0 = lds_txw + out_cap + 16;
    char* const src1 = src0 + src_cap + 16;
    const bool pe = (in.flags & BMBS_TEXT_PE) != 0;
    const long line0 = (long)blockIdx.x * lpb;
    const int tid = threadIdx.x;
    const int nl = (int)((n_lines - line0) < (long)lpb ? (n_lines - line0) : (long)lpb);      // lines of this workgroup
    const u64 o0 = off[line0], o1 = off[line0 + nl];
    const u32 piece = (u32)(o1 - o0);
    if (!piece) return;
    // source ranges: records r0 .. r1 of every file, from the name line of r0 to the last quality character of r1
    const long r0 = pe ? line0 >> 1 : line0, r1 = pe ? (line0 + nl - 1) >> 1 : line0 + nl - 1;
    const int nf = pe ? 2 : 1;
    u32 a0[2] = {0, 0}, span[2] = {0, 0};
    bool fits = piece <= out_cap;
    for (int f = 0; f < nf; f++) {
        const u32 s = in.rec[f].name_off[r0], e = in.rec[f].qual_off[r1] + in.rec[f].qual_len[r1];
        const u32 lead = (u32)((uintptr_t)(in.text[f] + s) & 15u);
        a0[f] = s - lead; span[f] = e - a0[f];
        fits = fits && span[f] <= src_cap;
    }
    const u32 lead = (u32)((uintptr_t)(out + o0) & 15u);
    char* const gout = out + o0 - lead;                        // 16-byte aligned; byte `lead` of the image is the piece's first
    if (fits) {
        // ---- 1. stage the text, fetch the records
        for (int f = 0; f < nf; f++) {
            const uint4* g = reinterpret_cast<const uint4*>(in.text[f] + a0[f]);
            uint4* d = reinterpret_cast<uint4*>(f ? src1 : src0);
            const int n16 = (int)((span[f] + 15u) >> 4);
#pragma unroll 2
            for (int i = tid; i < n16; i += TXW_THREADS) d[i] = g[i];
        }
        SamLine sl; sl.kind = 0;
        u32 start = 0;
        int L = 0, qn = 0;
        u32 nm0 = 0, nm1 = 0, so = 0, qo = 0; int nl0 = 0, nl1 = 0;
        if (tid < nl) {
            const long line = line0 + tid;
            sl = sam_line_core(in, line);
            start = (u32)(off[line] - o0);
            if (sl.kind) {
                const FqRec& R = in.rec[sl.mate];
                L = R.seq_len[sl.rec]; qn = R.qual_len[sl.rec];
                so = R.seq_off[sl.rec] - a0[sl.mate]; qo = R.qual_off[sl.rec] - a0[sl.mate];
                nm0 = in.rec[0].name_off[sl.rec] - a0[0]; nl0 = in.rec[0].name_len[sl.rec];
                if (pe) { nm1 = in.rec[1].name_off[sl.rec] - a0[1]; nl1 = in.rec[1].name_len[sl.rec]; }
            }
        }
        __syncthreads();
        // ---- 2. QNAME from the staged text; the columns that are computed, into the image
        if (tid < nl) {
            TxwDesc d;
            d.start = start; d.total = 0; d.name_rel = d.seq_rel = d.qual_rel = 0; d.nlen = d.hl = d.tl = d.L = d.qn = 0; d.rc = d.mate = 0;
            if (sl.kind) {
                if (!pe) qname_se(src0 + nm0, nl0, sl.name_skip, sl.name_len);
                else qname_pe(src0 + nm0, nl0, src1 + nm1, nl1, sl.name_skip, sl.name_len);
                char* const p = img + lead + start;
                int hl, tl;
                if (BAM) {
                    hl = 4 * bam_ncigar(sl); tl = bam_tag_len(sl);
                    const int total = txw_line_bytes<true>(sl, L, hl, tl);
                    bam_head(in, sl, L, total, p, p + 36 + sl.name_len + 1, p + total - tl);
                    d.total = (u32)total;
                } else {
                    hl = sam_head(in, sl, p + sl.name_len);
                    tl = sam_tail(sl, p + sl.name_len + hl + 2 * L + 1);
                    d.total = (u32)txw_line_bytes<false>(sl, L, hl, tl);
                }
                d.name_rel = (u16)(nm0 + sl.name_skip); d.seq_rel = (u16)so; d.qual_rel = (u16)qo;
                d.nlen = (u16)sl.name_len; d.hl = (u16)hl; d.tl = (u16)tl; d.L = (u16)L; d.qn = (u16)qn; d.rc = sl.rc ? 1 : 0; d.mate = (u8)sl.mate;
            }
            s_d[tid] = d;
        }
        __syncthreads();
        // ---- 3. QNAME / SEQ / QUAL: 16 lanes per line
        const int grp = tid >> 4, gl = tid & 15;
        for (int j = grp; j < nl; j += TXW_THREADS / 16) {
            const TxwDesc d = s_d[j];
            if (!d.total) continue;
            const int L = d.L, qn = d.qn, nlen = d.nlen;
            const bool rc = d.rc != 0;
            const char* const nm = src0 + d.name_rel;
            const char* const sq = (d.mate ? src1 : src0) + d.seq_rel;
            const char* const ql = (d.mate ? src1 : src0) + d.qual_rel;
            char* const p = img + lead + d.start;
            if (BAM) {
                char* const q = p + 36;
                for (int t = gl; t <= nlen; t += 16) q[t] = t < nlen ? nm[t] : (char)0;
                char* const s4 = q + nlen + 1 + d.hl;
                for (int t = gl; t < (L + 1) / 2; t += 16) {
                    const int i = 2 * t;
                    const u32 hi = c_nt16[txw_base((unsigned char)sq[rc ? L - 1 - i : i], rc)];
                    const u32 lo = i + 1 < L ? (u32)c_nt16[txw_base((unsigned char)sq[rc ? L - 2 - i : i + 1], rc)] : 0u;
                    s4[t] = (char)((hi << 4) | lo);
                }
                char* const qq = s4 + (L + 1) / 2;
                for (int i = gl; i < L; i += 16) { const int jj = rc ? L - 1 - i : i; qq[i] = (char)((jj < qn ? (u32)(unsigned char)ql[jj] : (u32)' ') - 33u); }
            } else {
                for (int t = gl; t < nlen; t += 16) p[t] = nm[t];
                char* const s1 = p + nlen + d.hl;
                for (int i = gl; i < L; i += 16) s1[i] = (char)txw_base((unsigned char)sq[rc ? L - 1 - i : i], rc);
                if (gl == 0) s1[L] = '\t';
                char* const qq = s1 + L + 1;
                for (int i = gl; i < L; i += 16) { const int jj = rc ? L - 1 - i : i; qq[i] = jj < qn ? ql[jj] : ' '; }       // qual.resize(seq.size(), ' ')
            }
        }
        __syncthreads();
        // ---- 4. the image goes out
        const u32 end = lead + piece;
        const int n16 = (int)((end + 15u) >> 4);
        for (int c = tid; c < n16; c += TXW_THREADS) {
            const u32 b = (u32)c << 4;
            if (b >= lead && b + 16u <= end) *reinterpret_cast<uint4*>(gout + b) = *reinterpret_cast<const uint4*>(img + b);
            else for (u32 i = b; i < b + 16u; i++) if (i >= lead && i < end) gout[i] = img[i];
        }
        return;
    }
    // ---- the plain path: a wave per line, its computed columns in the wave's quarter of the LDS buffer, bytes from and to memory
    const int wave = tid >> 6, lane = tid & 63;
    char* const hd = lds_txw + (size_t)wave * ((out_cap + 16) / 4);
    for (int j = wave; j < nl; j += TXW_THREADS / 64) {
        const long line = line0 + j;
        SamLine sl = sam_line(in, line);
        if (!sl.kind) continue;
        const FqRec& R = in.rec[sl.mate];
        const char* const text = in.text[sl.mate];
        const int L = R.seq_len[sl.rec], qn = R.qual_len[sl.rec];
        const char* const nm = in.text[0] + in.rec[0].name_off[sl.rec] + sl.name_skip;
        const char* const sq = text + R.seq_off[sl.rec];
        const char* const ql = text + R.qual_off[sl.rec];
        const bool rc = sl.rc;
        int hl, tl, total;
        if (BAM) { hl = 4 * bam_ncigar(sl); tl = bam_tag_len(sl); total = txw_line_bytes<true>(sl, L, hl, tl); }
        else { hl = sam_head(in, sl, nullptr); tl = sam_tail(sl, nullptr); total = txw_line_bytes<false>(sl, L, hl, tl); }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            if (BAM) bam_head(in, sl, L, total, hd, hd + 36, hd + 36 + hl);
            else { sam_head(in, sl, hd); sam_tail(sl, hd + hl); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        char* const dst = out + off[line];
        const int nlen = sl.name_len;
        for (int t = lane; t < total; t += 64) {
            u32 v;
            if (BAM) {
                const int b1 = 36, b2 = b1 + nlen + 1, b3 = b2 + hl, b4 = b3 + (L + 1) / 2, b5 = b4 + L;
                if (t < b1) v = (unsigned char)hd[t];
                else if (t < b2) v = t - b1 < nlen ? (u32)(unsigned char)nm[t - b1] : 0u;
                else if (t < b3) v = (unsigned char)hd[36 + (t - b2)];
                else if (t < b4) {
                    const int i = 2 * (t - b3);
                    const u32 hi = c_nt16[txw_base((unsigned char)sq[rc ? L - 1 - i : i], rc)];
                    const u32 lo = i + 1 < L ? (u32)c_nt16[txw_base((unsigned char)sq[rc ? L - 2 - i : i + 1], rc)] : 0u;
                    v = (hi << 4) | lo;
                }
                else if (t < b5) { const int i = t - b4, jj = rc ? L - 1 - i : i; v = ((jj < qn ? (u32)(unsigned char)ql[jj] : (u32)' ') - 33u) & 0xffu; }
                else v = (unsigned char)hd[36 + hl + (t - b5)];
            } else {
                const int b1 = nlen, b2 = b1 + hl, b3 = b2 + L, b4 = b3 + 1, b5 = b4 + L;
                if (t < b1) v = (unsigned char)nm[t];
                else if (t < b2) v = (unsigned char)hd[t - b1];
                else if (t < b3) { const int i = t - b2; v = txw_base((unsigned char)sq[rc ? L - 1 - i : i], rc); }
                else if (t < b4) v = (u32)'\t';
                else if (t < b5) { const int i = t - b4, jj = rc ? L - 1 - i : i; v = jj < qn ? (u32)(unsigned char)ql[jj] : (u32)' '; }
                else v = (unsigned char)hd[hl + (t - b5)];
            }
            dst[t] = (char)v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// -DBGZF_PROFILE (tools/bgzf_prof.sh): cycles per phase of k_bgzf_block as thread 0 sees them, summed over the blocks
#ifdef BGZF_PROFILE
__device__ unsigned long long g_bgzf_prof[8];
#define BZ_T(k) do { if (tid == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_bgzf_prof[k], t_ - bz_t); bz_t = t_; } } while (0)
#else
#define BZ_T(k)
#endif

// ---- BGZF: one workgroup deflates one block of <= 0xff00 input bytes --------------------------------------------------------------
#define BGZF_IN       0xff00                // input bytes per block (bgzf.h BGZF_BLOCK_SIZE)
#define BGZF_SLOT     65536                 // bytes of a block's slot in the scratch output (a stored block fits: 0xff00 + 5 + 26)
#define BGZF_THREADS  256
#define BGZF_SEG      256                   // input bytes parsed by one thread
#define DEF_LL        286                   // literal/length symbols
#define DEF_D         30
#define DEF_SYMS      320                   // ll at 0, distance codes at 288
#define DEF_DOFF      288

__constant__ u8 c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

DEVI void len_code(int len, int& code, int& extra, int& xval)
{
    const int v = len - 3;
    if (v < 8) { code = v; extra = 0; xval = 0; return; }
    if (len == 258) { code = 28; extra = 0; xval = 0; return; }
    const int m = 31 - __clz(v);
    code = 4 * (m - 1) + ((v >> (m - 2)) & 3); extra = m - 2; xval = v & ((1 << (m - 2)) - 1);
}
DEVI void dist_code(int dist, int& code, int& extra, int& xval)
{
    const int d = dist - 1;
    if (d < 4) { code = d; extra = 0; xval = 0; return; }
    const int m = 31 - __clz(d);
    code = 2 * m + ((d >> (m - 1)) & 1); extra = m - 1; xval = d & ((1 << (m - 1)) - 1);
}

// Huffman code lengths of n symbols limited to `maxbits`, by ONE lane over LDS arrays (the trees of a block are a few hundred
// sequential steps; the 255 other lanes wait at the barrier behind it).  zlib's construction in outline (build_tree / gen_bitlen,
// trees.c): at least two codes, overflowing leaves moved up, lengths handed out again in frequency order -- so that every decoder
// that reads zlib's streams reads these.  Work arrays: key[2n] (weights), par[2n], ord[n] (symbols by ascending frequency).
struct HuffWork { u32* key; u16* par; u16* ord; u8* dep; };
// ord[0, m) = the used symbols by ascending (frequency, symbol); m < 0: sort here (small alphabets)
DEVI void huff_lengths(const u32* freq, int n, int maxbits, u8* len, const HuffWork& w, int m)
{
    for (int i = 0; i < n; i++) len[i] = 0;
    if (m < 0) {
        m = 0;
        for (int s = 0; s < n; s++) {
            const u32 f = freq[s];
            if (!f) continue;
            int j = m++;
            while (j > 0 && freq[w.ord[j - 1]] > f) { w.ord[j] = w.ord[j - 1]; j--; }
            w.ord[j] = (u16)s;
        }
    }
    if (m == 0) { len[0] = 1; len[1] = 1; return; }                               // no symbol at all: two dummies (pkzip wants two codes)
    if (m == 1) { const int s = w.ord[0]; len[s] = 1; len[s == 0 ? 1 : 0] = 1; return; }
    // two-queue merge: leaves 0..m-1 in order, internal nodes m..2m-2 come out in ascending weight
    for (int i = 0; i < m; i++) w.key[i] = freq[w.ord[i]];
    int a = 0, b = m, e = m;                                                      // next leaf, next internal, end of internals
    for (int k = 0; k < m - 1; k++) {
        int c[2];
        for (int q = 0; q < 2; q++) {
            if (a < m && (b >= e || w.key[a] <= w.key[b])) c[q] = a++; else c[q] = b++;
        }
        w.key[e] = w.key[c[0]] + w.key[c[1]];
        w.par[c[0]] = (u16)e; w.par[c[1]] = (u16)e;
        e++;
    }
    // depths from the root down
    const int root = e - 1;
    w.dep[root] = 0;
    for (int i = root - 1; i >= 0; i--) w.dep[i] = (u8)(w.dep[w.par[i]] + 1 > 255 ? 255 : w.dep[w.par[i]] + 1);
    // bl_count with the overflow moved up (gen_bitlen).  Leaves deeper than maxbits are set to maxbits; what that adds to the Kraft sum
    // -- counted in units of 2^-maxbits: zlib counts it as half the nodes, leaves AND internal ones, whose parent sits at maxbits, which
    // the number of deep LEAVES alone is not once the tree is three levels deeper than maxbits -- is taken off one unit per step: a leaf
    // one level up moves down beside a leaf that comes up from maxbits
    int cnt[17]; for (int i = 0; i <= 16; i++) cnt[i] = 0;
    for (int i = 0; i < m; i++) { int d = w.dep[i]; if (d > maxbits) d = maxbits; cnt[d]++; }
    long kraft = 0;
    for (int d = 1; d <= maxbits; d++) kraft += (long)cnt[d] << (maxbits - d);
    long over = kraft - (1l << maxbits);
    while (over > 0) {
        int bits = maxbits - 1;
        while (cnt[bits] == 0) bits--;
        cnt[bits]--; cnt[bits + 1] += 2; cnt[maxbits]--;
        over--;
    }
    // the rarest symbols get the longest codes
    int i = 0;
    for (int bits = maxbits; bits >= 1; bits--) for (int c2 = cnt[bits]; c2 > 0; c2--) len[w.ord[i++]] = (u8)bits;
}
// canonical codes, bit-reversed for an LSB-first stream: code[s] = len << 16 | reversed code
DEVI void huff_codes(const u8* len, int n, u32* code)
{
    int cnt[16]; for (int i = 0; i < 16; i++) cnt[i] = 0;
    for (int s = 0; s < n; s++) cnt[len[s]]++;
    cnt[0] = 0;
    u32 next[16]; u32 c = 0;
    for (int b = 1; b <= 15; b++) { c = (c + (u32)cnt[b - 1]) << 1; next[b] = c; }
    for (int s = 0; s < n; s++) {
        const int l = len[s];
        if (!l) { code[s] = 0; continue; }
        const u32 v = next[l]++;
        code[s] = ((u32)l << 16) | (__brev(v) >> (32 - l));
    }
}

// ---- the same two functions by a whole workgroup (round 5): in k_bgzf_block one lane built the literal/length tree while 255 waited, ~9 000
// dependent LDS accesses = 0.3 ms per block.  Only the two-queue merge is a chain; the rest is data-parallel: depths by pointer jumping
// over the parent links, the length counts by LDS atomics, a symbol's length from its rank, a symbol's code from the number of symbols
// of its length in front of it.  Same lengths and codes as huff_lengths / huff_codes (tests: bmbs_debug_huff_lengths with form 1).
// scratch: w.key is reused once the merge is over (ancestor links and depths of the 2m - 1 nodes, 16 bits each); cnt_s: 20 words
DEVI void huff_lengths_block(const u32* freq, int n, int maxbits, u8* len, const HuffWork& w, int m, int tid, int nt, u32* cnt_s)
{
    for (int i = tid; i < n; i += nt) len[i] = 0;
    if (m < 2) {                                                                    // (m is the same in every thread)
        __syncthreads();
        if (tid == 0) { if (m == 0) { len[0] = 1; len[1] = 1; } else { const int s = w.ord[0]; len[s] = 1; len[s == 0 ? 1 : 0] = 1; } }
        __syncthreads();
        return;
    }
    for (int i = tid; i < m; i += nt) w.key[i] = freq[w.ord[i]];
    __syncthreads();
    if (tid == 0) {
        int a = 0, b = m, e = m;
        for (int k = 0; k < m - 1; k++) {
            int c[2];
            for (int q = 0; q < 2; q++) {
                if (a < m && (b >= e || w.key[a] <= w.key[b])) c[q] = a++; else c[q] = b++;
            }
            w.key[e] = w.key[c[0]] + w.key[c[1]];
            w.par[c[0]] = (u16)e; w.par[c[1]] = (u16)e;
            e++;
        }
    }
    __syncthreads();
    const int nn = 2 * m - 1, root = nn - 1;
    u16* anc = reinterpret_cast<u16*>(w.key);                                       // [nn]
    u16* dp = anc + 2 * DEF_LL;                                                     // [nn] (w.key has 2 * DEF_LL words = 4 * DEF_LL halves)
    for (int i = tid; i < nn; i += nt) { anc[i] = (u16)(i == root ? root : w.par[i]); dp[i] = (u16)(i == root ? 0 : 1); }
    __syncthreads();
    for (int span = 1; span < nn; span <<= 1) {
        u16 na[3], nd[3];                                                           // nn <= 571 nodes, nt = 256 threads: at most three each
        int q = 0;
        for (int i = tid; i < nn; i += nt, q++) { const int a = anc[i]; na[q] = anc[a]; nd[q] = (u16)(dp[i] + dp[a]); }
        __syncthreads();
        q = 0;
        for (int i = tid; i < nn; i += nt, q++) { anc[i] = na[q]; dp[i] = nd[q]; }
        __syncthreads();
    }
    if (tid < 20) cnt_s[tid] = 0;
    __syncthreads();
    for (int i = tid; i < m; i += nt) { int d = dp[i]; if (d > maxbits) d = maxbits; atomicAdd(&cnt_s[d], 1u); }
    __syncthreads();
    if (tid == 0) {
        // bl_count with the overflow moved up (gen_bitlen): see huff_lengths
        int cnt[17];
        for (int i = 0; i <= 16; i++) cnt[i] = i <= maxbits ? (int)cnt_s[i] : 0;
        long kraft = 0;
        for (int d = 1; d <= maxbits; d++) kraft += (long)cnt[d] << (maxbits - d);
        long over = kraft - (1l << maxbits);
        while (over > 0) {
            int bits = maxbits - 1;
            while (cnt[bits] == 0) bits--;
            cnt[bits]--; cnt[bits + 1] += 2; cnt[maxbits]--;
            over--;
        }
        for (int i = 0; i <= 16; i++) cnt_s[i] = (u32)cnt[i];
    }
    __syncthreads();
    // the rarest symbols get the longest codes: rank i (ascending frequency) falls into the run of its length
    for (int i = tid; i < m; i += nt) {
        int acc = 0, bits = maxbits;
        for (; bits >= 1; bits--) { acc += (int)cnt_s[bits]; if (i < acc) break; }
        len[w.ord[i]] = (u8)bits;
    }
    __syncthreads();
}
// canonical codes of n symbols by the workgroup: next_s: 16 words of scratch
DEVI void huff_codes_block(const u8* len, int n, u32* code, int tid, int nt, u32* next_s)
{
    if (tid < 16) next_s[tid] = 0;
    __syncthreads();
    for (int s = tid; s < n; s += nt) if (len[s]) atomicAdd(&next_s[len[s]], 1u);
    __syncthreads();
    if (tid == 0) {
        u32 cnt[16]; for (int b = 0; b < 16; b++) cnt[b] = next_s[b];
        cnt[0] = 0;
        u32 c = 0;
        for (int b = 1; b <= 15; b++) { c = (c + cnt[b - 1]) << 1; next_s[b] = c; }
    }
    __syncthreads();
    for (int s = tid; s < n; s += nt) {
        const int l = len[s];
        if (!l) { code[s] = 0; continue; }
        u32 r = 0;
        for (int t = 0; t < s; t++) r += len[t] == l ? 1u : 0u;                      // symbols of the same length in front of s
        const u32 v = next_s[l] + r;
        code[s] = ((u32)l << 16) | (__brev(v) >> (32 - l));
    }
    __syncthreads();
}

struct BitOut { u32* w; u32 pos; };       // LSB-first bits OR-ed into zeroed 32-bit words (one writer)
DEVI void put_bits(BitOut& o, u32 v, int n)
{
    if (!n) return;
    const u32 wi = o.pos >> 5, sh = o.pos & 31;
    o.w[wi] |= v << sh;
    if (sh + n > 32) o.w[wi + 1] |= v >> (32 - sh);
    o.pos += (u32)n;
}

// raw: the BAM record stream of the batch (total bytes); block b = raw[b * BGZF_IN ...).  slot b of `slots` receives the finished
// BGZF block (header, deflate data, CRC-32, ISIZE), slot_len[b] its length.
// Round 5: a thread's 256-byte segment sits in LDS at a stride of 65 words, so that the 64 lanes of a wave reading "their" word j hit
// 32 different banks (at 64 words apart every access of the byte loops was a 64-way bank conflict: 58 % of the kernel's time); the
// tokens -- a literal byte, or a run of the byte before it (distance 1) -- are not stored anywhere: they follow from 256 "equals the
// byte before it" bits a thread keeps in registers, and the three passes that need them (symbol counts, bit lengths, emit) walk the
// segment again; the deflate stream goes straight into the block's slot in memory (whole words stored, shared edge words OR-ed in).  (Round 4 kept 16-bit tokens in memory: 590 MB of
// scratch and 1.2 GB of 2-byte loads and stores per 300 MB of records.)
#define BGZF_PSTRIDE  65                    // words between the segments of two threads in LDS
// number of consecutive set bits of e (256 bits, LSB first) from position p on
DEVI int bz_streak(const u64 (&e)[4], int p)
{
    int n = 0;
    for (int w = p >> 6, sh = p & 63; w < 4; w++, sh = 0) {
        const u64 x = ~(e[w] >> sh);
        const int room = 64 - sh;
        const int t = x ? __builtin_ctzll(x) : 64;
        if (t < room) return n + t;
        n += room;
    }
    return n;
}
// the tokens of a segment of n bytes (seg: its words in LDS), in order: lit(byte) / run(length >= 3)
template <class FL, class FR>
DEVI void bz_walk(const u32* seg, const u64 (&e)[4], int n, FL lit, FR run)
{
    int skip = 0;
    for (int j = 0; 4 * j < n; j++) {
        const u32 w = seg[j];
        const u32 e4 = (u32)(e[j >> 4] >> (4 * (j & 15))) & 15u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int p = 4 * j + q;
            if (p >= n) break;
            if (skip > 0) { skip--; continue; }
            if ((e4 >> q) & 1u) {
                const int sl = bz_streak(e, p);
                if (sl >= 3) { run(sl); skip = sl - 1; continue; }
            }
            lit((w >> (8 * q)) & 0xffu);
        }
    }
}

__global__ void __launch_bounds__(BGZF_THREADS)
k_bgzf_block(const char* __restrict__ raw, const u64* __restrict__ total_ptr, char* __restrict__ slots, u32* __restrict__ slot_len)
{
    extern __shared__ __align__(16) u32 s_dyn[];      // [BGZF_THREADS * BGZF_PSTRIDE] the block's bytes
    u32* const s_in = s_dyn;
    __shared__ u32 s_freq[DEF_SYMS];
    __shared__ u32 s_code[DEF_SYMS];
    __shared__ u8 s_len[DEF_SYMS + 8];
    __shared__ u32 s_crc_tab[256];
    __shared__ u32 s_key[2 * DEF_LL]; __shared__ u16 s_par[2 * DEF_LL]; __shared__ u16 s_ord[DEF_LL]; __shared__ u16 s_ord_d[DEF_D + 2]; __shared__ u16 s_ord_c[20]; __shared__ u8 s_dep[2 * DEF_LL];
    __shared__ u32 s_hdr[176];                        // header bits of the dynamic block
    __shared__ u32 s_cl_freq[19]; __shared__ u32 s_cl_code[19]; __shared__ u8 s_cl_len[19];
    __shared__ u8 s_rle_sym[DEF_SYMS]; __shared__ u8 s_rle_x[DEF_SYMS];
    __shared__ u32 s_wave[BGZF_THREADS / 64];
    __shared__ u32 s_misc[8];                         // 0 xbits, 1 crc, 2 hdr bits, 3 mode (0 stored, 1 dynamic), 4 body bits, 6 / 7 used ll / distance symbols

    const u64 total = *total_ptr;
    const u64 base = (u64)blockIdx.x * BGZF_IN;
    if (base >= total) return;
    const int blen = (int)((total - base) < (u64)BGZF_IN ? (total - base) : (u64)BGZF_IN);
    const int tid = threadIdx.x;
#ifdef BGZF_PROFILE
    unsigned long long bz_t = clock64();
#endif
    // ---- load (raw + base is 16-byte aligned: BGZF_IN is a multiple of 16), tables
    {
        const uint4* src = reinterpret_cast<const uint4*>(raw + base);
        const int n16 = (blen + 15) >> 4;
        for (int i = tid; i < n16; i += BGZF_THREADS) {                          // (the raw buffer is padded past `total`)
            const uint4 v = src[i];
            u32* d = s_in + (i >> 4) * BGZF_PSTRIDE + (i & 15) * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        for (int i = tid; i < DEF_SYMS; i += BGZF_THREADS) s_freq[i] = 0;
        u32 c = (u32)tid;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1;
        s_crc_tab[tid] = c;
        if (tid < 8) s_misc[tid] = 0;
    }
    __syncthreads();
    BZ_T(0);
    // ---- parse: greedy, literals and runs (distance 1); CRC of the segment on the way
    const int s0 = tid * BGZF_SEG, s1 = s0 + BGZF_SEG < blen ? s0 + BGZF_SEG : blen;
    const int nseg = s1 > s0 ? s1 - s0 : 0;          // bytes of this thread's segment
    const u32* seg = s_in + tid * BGZF_PSTRIDE;
    u64 eq[4] = {0, 0, 0, 0};                         // bit p: byte p of the segment equals the byte before it
    u32 xbits = 0, crc = 0;
    if (nseg > 0) {
        u32 c = 0xffffffffu;
        u32 prev = tid > 0 ? s_in[(tid - 1) * BGZF_PSTRIDE + 63] >> 24 : 0x100u;  // (0x100: nothing in front of the block's first byte)
        for (int j = 0; 4 * j < nseg; j++) {
            const u32 w = seg[j];
            u32 m4 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 bq = (w >> (8 * q)) & 0xffu;
                if (4 * j + q < nseg) {
                    c = s_crc_tab[(c ^ bq) & 0xffu] ^ (c >> 8);
                    m4 |= (bq == prev ? 1u : 0u) << q;
                    prev = bq;
                }
            }
            eq[j >> 4] |= (u64)m4 << (4 * (j & 15));
        }
        c = ~c;
        crc = crc_multmodp(crc_x8n((u32)(blen - s1)), c);                       // this segment's share of the block's CRC
        bz_walk(seg, eq, nseg,
                [&](u32 b) { atomicAdd(&s_freq[b], 1u); },
                [&](int run) { int lc, le, lx; len_code(run, lc, le, lx); atomicAdd(&s_freq[257 + lc], 1u); atomicAdd(&s_freq[DEF_DOFF + 0], 1u); xbits += (u32)le; });
    }
    if (tid == 0) atomicAdd(&s_freq[256], 1u);
    // block sums of xbits and the CRC shares (xor)
    for (int o = 32; o > 0; o >>= 1) { xbits += __shfl_down(xbits, o, 64); crc ^= __shfl_down(crc, o, 64); }
    if ((tid & 63) == 0) { atomicAdd(&s_misc[0], xbits); atomicXor(&s_misc[1], crc); }
    __syncthreads();
    BZ_T(1);
    // ---- the used symbols by ascending (frequency, symbol): every thread ranks its symbols against all the others (an insertion
    // sort by one lane is ~20 000 dependent LDS steps for a block that uses all 256 byte values)
    for (int sy = tid; sy < DEF_LL + DEF_D; sy += BGZF_THREADS) {
        const bool dsym = sy >= DEF_LL;
        const u32* fr = dsym ? s_freq + DEF_DOFF : s_freq;
        const int me = dsym ? sy - DEF_LL : sy, n = dsym ? DEF_D : DEF_LL;
        const u32 f = fr[me];
        if (f) {
            int r = 0;
            for (int t = 0; t < n; t++) { const u32 g = fr[t]; r += (g && (g < f || (g == f && t < me))) ? 1 : 0; }
            (dsym ? s_ord_d : s_ord)[r] = (u16)me;
            atomicAdd(&s_misc[dsym ? 7 : 6], 1u);
        }
    }
    __syncthreads();
    BZ_T(2);
    // ---- trees: the literal/length tree by the whole workgroup, the distance tree (one used symbol at most: every match is a run at
    // distance 1) by one lane
    __shared__ u32 s_cnt[20];
    {
        HuffWork hw; hw.key = s_key; hw.par = s_par; hw.ord = s_ord; hw.dep = s_dep;
        huff_lengths_block(s_freq, DEF_LL, 15, s_len, hw, (int)s_misc[6], tid, BGZF_THREADS, s_cnt);
        if (tid == 0) {
            hw.ord = s_ord_d;
            huff_lengths(s_freq + DEF_DOFF, DEF_D, 15, s_len + DEF_DOFF, hw, (int)s_misc[7]);
            huff_codes(s_len + DEF_DOFF, DEF_D, s_code + DEF_DOFF);
        }
        huff_codes_block(s_len, DEF_LL, s_code, tid, BGZF_THREADS, s_cnt);
    }
    // ---- the header of the dynamic block: the run-length coding of the code lengths and its small tree by one lane (a chain of ~300
    // steps), the header's bits by everybody
    __shared__ u32 s_hn[4];                           // 0 nr, 1 hlit, 2 hdist, 3 hclen
    if (tid == 0) {
        HuffWork hw; hw.key = s_key; hw.par = s_par; hw.ord = s_ord_c; hw.dep = s_dep;
        int hlit = DEF_LL; while (hlit > 257 && !s_len[hlit - 1]) hlit--;
        int hdist = DEF_D; while (hdist > 1 && !s_len[DEF_DOFF + hdist - 1]) hdist--;
        // the hlit + hdist lengths as one sequence, run-length coded with 16 / 17 / 18 (send_tree)
        for (int i = 0; i < 19; i++) s_cl_freq[i] = 0;
        int nr = 0;
        const int nseq = hlit + hdist;
        auto at = [&](int i) -> int { return i < hlit ? s_len[i] : s_len[DEF_DOFF + (i - hlit)]; };
        for (int i = 0; i < nseq;) {
            const int v = at(i);
            int run = 1;
            while (i + run < nseq && at(i + run) == v) run++;
            int left = run;
            if (v == 0) {
                while (left >= 11) { const int r = left < 138 ? left : 138; s_rle_sym[nr] = 18; s_rle_x[nr++] = (u8)(r - 11); s_cl_freq[18]++; left -= r; }
                if (left >= 3) { s_rle_sym[nr] = 17; s_rle_x[nr++] = (u8)(left - 3); s_cl_freq[17]++; left = 0; }
                while (left-- > 0) { s_rle_sym[nr] = 0; s_rle_x[nr++] = 0; s_cl_freq[0]++; }
            } else {
                s_rle_sym[nr] = (u8)v; s_rle_x[nr++] = 0; s_cl_freq[v]++; left--;
                while (left >= 3) { const int r = left < 6 ? left : 6; s_rle_sym[nr] = 16; s_rle_x[nr++] = (u8)(r - 3); s_cl_freq[16]++; left -= r; }
                while (left-- > 0) { s_rle_sym[nr] = (u8)v; s_rle_x[nr++] = 0; s_cl_freq[v]++; }
            }
            i += run;
        }
        huff_lengths(s_cl_freq, 19, 7, s_cl_len, hw, -1);
        huff_codes(s_cl_len, 19, s_cl_code);
        int hclen = 19; while (hclen > 4 && !s_cl_len[c_cl_order[hclen - 1]]) hclen--;
        s_hn[0] = (u32)nr; s_hn[1] = (u32)hlit; s_hn[2] = (u32)hdist; s_hn[3] = (u32)hclen;
    }
    for (int i = tid; i < 176; i += BGZF_THREADS) s_hdr[i] = 0;
    __syncthreads();
    {
        // the header's bits: 17 bits of counts + 3 bits per code-length-code length, then the nr items, each at the bit its predecessors
        // end at (a prefix sum of the items' bit lengths over the workgroup: at most two items per thread)
        const int nr = (int)s_hn[0], hclen = (int)s_hn[3];
        const u32 fixed = 17u + 3u * (u32)hclen;
        auto item_bits = [&](int i) -> u32 { const int sy = s_rle_sym[i]; return (s_cl_code[sy] >> 16) + (sy == 16 ? 2u : sy == 17 ? 3u : sy == 18 ? 7u : 0u); };
        const int i0 = 2 * tid, i1 = i0 + 1;
        const u32 b0 = i0 < nr ? item_bits(i0) : 0u, b1 = i1 < nr ? item_bits(i1) : 0u;
        u32 incl = b0 + b1;
        for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += v; }
        if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        u32 wb = 0, all = 0;
        for (int i = 0; i < BGZF_THREADS / 64; i++) { if (i < (tid >> 6)) wb += s_wave[i]; all += s_wave[i]; }
        auto put = [&](u32 pos, u32 v, int nbits) {                                 // nbits <= 14
            if (!nbits) return;
            const u32 wi = pos >> 5, sh = pos & 31;
            atomicOr(&s_hdr[wi], v << sh);
            if (sh + (u32)nbits > 32) atomicOr(&s_hdr[wi + 1], v >> (32 - sh));
        };
        if (tid == 0) {
            put(0, 1, 1); put(1, 2, 2);                                             // BFINAL, BTYPE = dynamic
            put(3, s_hn[1] - 257u, 5); put(8, s_hn[2] - 1u, 5); put(13, (u32)hclen - 4u, 4);
        }
        if (tid < hclen) put(17u + 3u * (u32)tid, s_cl_len[c_cl_order[tid]], 3);
        u32 pos = fixed + wb + (incl - b0 - b1);
        for (int q = 0; q < 2; q++) {
            const int i = q ? i1 : i0;
            if (i >= nr) break;
            const int sy = s_rle_sym[i];
            const u32 cl = s_cl_code[sy] >> 16;
            put(pos, s_cl_code[sy] & 0xffffu, (int)cl);
            const int xb = sy == 16 ? 2 : sy == 17 ? 3 : sy == 18 ? 7 : 0;
            if (xb) put(pos + cl, s_rle_x[i], xb);
            pos += cl + (u32)xb;
        }
        // the body's bits: extra bits of the run lengths + every symbol's count times its code length
        u32 body = 0;
        for (int sy = tid; sy < DEF_SYMS; sy += BGZF_THREADS) body += s_freq[sy] * (u32)s_len[sy];
        for (int o = 32; o > 0; o >>= 1) body += __shfl_down(body, o, 64);
        __syncthreads();                                                            // (s_wave was read by everybody)
        if ((tid & 63) == 0) s_wave[tid >> 6] = body;
        __syncthreads();
        if (tid == 0) {
            u32 bsum = s_misc[0];
            for (int i = 0; i < BGZF_THREADS / 64; i++) bsum += s_wave[i];
            const u32 hb = fixed + all;
            s_misc[2] = hb; s_misc[4] = bsum;
            const u32 dyn_bytes = (hb + bsum + 7) >> 3;
            s_misc[3] = dyn_bytes < (u32)blen + 5u ? 1u : 0u;
        }
    }
    __syncthreads();
    BZ_T(3);
    const bool dynamic = s_misc[3] != 0;
    const u32 hdr_bits = s_misc[2], body_bits = s_misc[4];
    const u32 crc_all = s_misc[1];
    char* slot = slots + (size_t)blockIdx.x * BGZF_SLOT;
    u32 clen;                                                                      // bytes of deflate data
    if (!dynamic) {
        // stored: 1 byte of type bits, LEN, NLEN, the bytes
        clen = 5u + (u32)blen;
        // (bytes 18, 19 share slot word 4 with BSIZE: OR-ed in below like every shared word)
        if (tid == 0) { atomicOr(reinterpret_cast<u32*>(slot) + 4, (1u << 16) | (((u32)blen & 0xffu) << 24)); slot[20] = (char)(blen >> 8); slot[21] = (char)~blen; slot[22] = (char)(~blen >> 8); }
        for (int i = tid; i < blen; i += BGZF_THREADS) slot[23 + i] = (char)(s_in[(i >> 8) * BGZF_PSTRIDE + ((i & 255) >> 2)] >> (8 * (i & 3)));
    } else {
        clen = (hdr_bits + body_bits + 7) >> 3;
        // bits of this thread's tokens, exclusive prefix over the block
        u32 mybits = 0;
        const u32 dcode = s_code[DEF_DOFF + 0];                                    // distance 1: code 0, no extra bits
        bz_walk(seg, eq, nseg,
                [&](u32 b) { mybits += s_code[b] >> 16; },
                [&](int run) { int lc, le, lx; len_code(run, lc, le, lx); mybits += (s_code[257 + lc] >> 16) + (u32)le + (dcode >> 16); });
        u32 incl = mybits;
        for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += v; }
        if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        BZ_T(4);
        u32 wbase = 0;
        for (int i = 0; i < (tid >> 6); i++) wbase += s_wave[i];
        // The stream goes straight into the block's slot in memory (zeroed by the host before the launch), as 32-bit words of the slot:
        // bit g of the slot = bit g % 32 of word g / 32; the deflate data starts at byte 18 = slot bit 144.  A thread owns the words that
        // lie wholly inside its bit range and stores them; the word at either end it shares with a neighbour (or with the gzip header /
        // the dynamic-block header in front) and ORs its bits in.
        u32* const gw = reinterpret_cast<u32*>(slot);
        // ... the dynamic-block header first: words of s_hdr shifted by 144 % 32 = 16 bits
        for (u32 i = tid; i * 32 < hdr_bits + 16; i += BGZF_THREADS) {
            const u32 lo = i ? s_hdr[i - 1] >> 16 : 0u, hi = i < 176 ? s_hdr[i] << 16 : 0u;         // slot word 4 + i
            const u32 v = lo | hi;
            const bool whole = i > 0 && (i + 1) * 32 <= hdr_bits + 16;
            if (whole) gw[4 + i] = v; else if (v) atomicOr(&gw[4 + i], v);
        }
        const u32 pos = 144u + hdr_bits + wbase + (incl - mybits);                  // slot bit of this thread's first token
        const u32 endb = pos + mybits + ((s0 < blen && s1 == blen) ? (s_code[256] >> 16) : 0u);
        unsigned long long acc = 0; int have = (int)(pos & 31); u32 wi = pos >> 5;  // bits of word wi below `have` belong to the neighbour
        bool first = true;
        auto flush = [&](u32 v) {
            // the first word is shared unless the range starts on its boundary; a later word is whole (its 32 bits are this thread's)
            if (first && (pos & 31)) atomicOr(&gw[wi], v); else gw[wi] = v;
            first = false;
        };
        auto emit = [&](u32 v, int n) {
            acc |= (unsigned long long)v << have; have += n;
            if (have >= 32) { flush((u32)acc); wi++; acc >>= 32; have -= 32; }
        };
        bz_walk(seg, eq, nseg,
                [&](u32 b) { const u32 cd = s_code[b]; emit(cd & 0xffffu, (int)(cd >> 16)); },
                [&](int run) {
                    int lc, le, lx; len_code(run, lc, le, lx);
                    const u32 cd = s_code[257 + lc];
                    emit(cd & 0xffffu, (int)(cd >> 16));
                    if (le) emit((u32)lx, le);
                    emit(dcode & 0xffffu, (int)(dcode >> 16));
                });
        // the thread that holds the end of the input also writes the end-of-block code (exactly one: s0 < blen, s1 == blen)
        if (s0 < blen && s1 == blen) emit(s_code[256] & 0xffffu, (int)(s_code[256] >> 16));
        if (have > 0 && (u32)(wi << 5) < endb) atomicOr(&gw[wi], (u32)acc);         // the last, partial word
        BZ_T(5);
    }
    if (tid == 0) {
        // (bytes 16, 17 share a word with the first two bytes of the deflate data: OR-ed in like everything shared)
        const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        for (int i = 0; i < 16; i++) slot[i] = (char)hdr[i];
        const u32 bsize = clen + 25u;                                               // total block size - 1
        atomicOr(reinterpret_cast<u32*>(slot) + 4, bsize & 0xffffu);
    }
    __syncthreads();
    if (tid == 0) {
        // the trailer behind the stream's last byte (its bits are in memory: every thread has passed the barrier; the bytes are this
        // thread's alone -- but they may share a word with the stream's last bits, so they are OR-ed in as well)
        const u32 at = 18u + clen;
        const u32 isz = (u32)blen;
        unsigned long long t8 = (unsigned long long)crc_all | ((unsigned long long)isz << 32);
        u32* w = reinterpret_cast<u32*>(slot) + (at >> 2);
        const int sh = (int)(at & 3u) * 8;
        atomicOr(&w[0], (u32)(t8 << sh));
        atomicOr(&w[1], (u32)(sh ? t8 >> (32 - sh) : t8 >> 32));
        if (sh) atomicOr(&w[2], (u32)(t8 >> (64 - sh)));
        slot_len[blockIdx.x] = clen + 26u;
    }
    BZ_T(6);
}

// slots -> one contiguous piece: block b goes to out[off[b] ...); destination-aligned dwords, the ragged ends byte-wise
__global__ void __launch_bounds__(256)
k_bgzf_gather(const char* __restrict__ slots, const u32* __restrict__ slot_len, const u64* __restrict__ off, char* __restrict__ out)
{
    const u32 b = blockIdx.x;
    const u32 len = slot_len[b];
    const u64 o = off[b];
    const char* src = slots + (size_t)b * BGZF_SLOT;
    const u64 d0 = o & ~3ull;
    const int lead = (int)(o - d0);
    const int ndw = (int)((lead + len + 3) >> 2);
    for (int w = threadIdx.x; w < ndw; w += blockDim.x) {
        const int t0 = 4 * w - lead;
        char* dst = out + d0 + 4 * (u64)w;
        if (t0 >= 0 && t0 + 4 <= (int)len) {
            const u32 v = (u32)(unsigned char)src[t0] | ((u32)(unsigned char)src[t0 + 1] << 8) | ((u32)(unsigned char)src[t0 + 2] << 16) | ((u32)(unsigned char)src[t0 + 3] << 24);
            *reinterpret_cast<u32*>(dst) = v;
        } else {
            for (int k = 0; k < 4; k++) { const int t = t0 + k; if (t >= 0 && t < (int)len) dst[k] = src[t]; }
        }
    }
}

// diagnostic (bmbs_debug_huff_lengths): the code lengths huff_lengths gives a frequency table, as k_bgzf_block calls it (m < 0: sorted here)
// -- and, for alphabets of up to DEF_LL symbols, the lengths AND codes of the workgroup forms (huff_lengths_block / huff_codes_block on
// the ranks k_bgzf_block computes): *differ = 1 when they are not the serial forms' lengths and codes
__global__ void __launch_bounds__(256)
k_debug_huff(const u32* __restrict__ freq, int n, int maxbits, u8* __restrict__ len_out, u32* __restrict__ differ)
{
    __shared__ u32 s_f[320]; __shared__ u32 s_key[640]; __shared__ u16 s_par[640]; __shared__ u16 s_ord[320]; __shared__ u8 s_dep[640]; __shared__ u8 s_len[320];
    __shared__ u8 s_len2[320]; __shared__ u32 s_code[320]; __shared__ u32 s_code2[320]; __shared__ u32 s_cnt[20]; __shared__ u32 s_m;
    const int tid = threadIdx.x;
    if (tid == 0) {
        for (int i = 0; i < n; i++) s_f[i] = freq[i];
        HuffWork hw; hw.key = s_key; hw.par = s_par; hw.ord = s_ord; hw.dep = s_dep;
        huff_lengths(s_f, n, maxbits, s_len, hw, -1);
        huff_codes(s_len, n, s_code);
        for (int i = 0; i < n; i++) len_out[i] = s_len[i];
        s_m = 0;
    }
    __syncthreads();
    if (n > DEF_LL) return;
    for (int sy = tid; sy < n; sy += 256) {
        const u32 f = s_f[sy];
        if (f) {
            int r = 0;
            for (int t = 0; t < n; t++) { const u32 g = s_f[t]; r += (g && (g < f || (g == f && t < sy))) ? 1 : 0; }
            s_ord[r] = (u16)sy;
            atomicAdd(&s_m, 1u);
        }
    }
    __syncthreads();
    HuffWork hw; hw.key = s_key; hw.par = s_par; hw.ord = s_ord; hw.dep = s_dep;
    huff_lengths_block(s_f, n, maxbits, s_len2, hw, (int)s_m, tid, 256, s_cnt);
    huff_codes_block(s_len2, n, s_code2, tid, 256, s_cnt);
    for (int i = tid; i < n; i += 256) if (s_len2[i] != s_len[i] || s_code2[i] != s_code[i]) atomicOr(differ, 1u);
}
#endif
