// bitmapperbs_amd/csrc/bmbs_text.hip -- the two ends of the file-to-file path on the device (round 3).
//
//   FASTQ text  ->  line index          k_fq_count / k_fq_lines / k_fq_records
//   records     ->  SAM text            k_sam_len / (scan) / k_sam_write
//
// What the reference does per record on ONE host thread at each end -- kseq line splitting in inputReads_single_directly /
// inputReads_paired_directly (Process_Reads.cpp:810-890, 155-317) and the fprintf / buffer formatting of output_sam_end_to_end,
// directly_output_read1 / _read2, output_sam_unmapped, directly_output_unmapped_PE (Schema.cpp:11989-12039, 10537-10640,
// 11494-11590, 23955-23975, 10392-10430) -- and what round 2 did with the host's I/O threads, happens here for a whole batch: the
// host hands over the text window as it came from the file and gets finished SAM text back; in between nothing leaves the device.
// Byte work: one newline per ~80 bytes going in, ~350 bytes per read going out; both kernels are bound by their own stores and
// by the link behind them, not by arithmetic.
#ifndef BMBS_TEXT_HIP
#define BMBS_TEXT_HIP
#include "bmbs_bytes.h"

// ---- FASTQ text -> newline positions -----------------------------------------------------------------------------------------------
#define FQ_TILE_THREADS 256
#define FQ_BYTES_PER_THREAD 64
#define FQ_TILE_BYTES (FQ_TILE_THREADS * FQ_BYTES_PER_THREAD)

// the 64 bytes of thread `t` of a tile as a 64-bit mask of newline positions (bit j = byte j); bytes at and beyond `n` do not count
DEVI u64 nl_mask64(const char* __restrict__ text, u64 base, u64 n)
{
    u64 m = 0;
    if (base + 64 <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(text + base);           // the window is 16-byte aligned (hipMalloc), base % 64 == 0
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = p[q];
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 z = nl_mask4(w[j]);
                // 0x80 per matching byte -> one bit per byte
                const u32 b = ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
                m |= (u64)b << (16 * q + 4 * j);
            }
        }
    } else {
        for (int j = 0; j < 64; j++) if (base + j < n && text[base + j] == '\n') m |= 1ull << j;
    }
    return m;
}

__global__ void __launch_bounds__(FQ_TILE_THREADS)
k_fq_count(const char* __restrict__ text, u64 n, u32* __restrict__ tile_count)
{
    __shared__ u32 sh[FQ_TILE_THREADS / 64];
    const u64 base = (u64)blockIdx.x * FQ_TILE_BYTES + (u64)threadIdx.x * FQ_BYTES_PER_THREAD;
    u32 c = base < n ? (u32)__popcll(nl_mask64(text, base, n)) : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < FQ_TILE_THREADS / 64; i++) t += sh[i]; tile_count[blockIdx.x] = t; }
}

// nl[j] = position of the newline that ends line j, for the first max_lines lines
__global__ void __launch_bounds__(FQ_TILE_THREADS)
k_fq_lines(const char* __restrict__ text, u64 n, const u64* __restrict__ tile_off, u64 max_lines, u32* __restrict__ nl)
{
    __shared__ u32 sh[FQ_TILE_THREADS / 64];
    const u64 base = (u64)blockIdx.x * FQ_TILE_BYTES + (u64)threadIdx.x * FQ_BYTES_PER_THREAD;
    u64 m = base < n ? nl_mask64(text, base, n) : 0ull;
    const u32 c = (u32)__popcll(m);
    // exclusive prefix of c inside the block: wave scan + wave totals through LDS
    u32 incl = c;
    for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += v; }
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 wbase = 0;
    for (int i = 0; i < (int)(threadIdx.x >> 6); i++) wbase += sh[i];
    u64 line = tile_off[blockIdx.x] + wbase + (incl - c);
    while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        if (line < max_lines) nl[line] = (u32)(base + (u64)b);
        line++;
    }
}

// per-record fields from the newline positions (record r = lines 4r .. 4r+3) + the checks the host reader made:
// a sequence line of 1..998 characters (BMBS_MAX_READ).  info[0] = longest read, info[1] = ~shortest (atomicMax of the complement),
// info[2] = 1 + index of a bad record (0: none)
struct FqRec {
    u32* seq_off; u32* qual_off; u32* name_off; u16* seq_len; u16* qual_len; u16* name_len;
};
__global__ void __launch_bounds__(256)
k_fq_records(const u32* __restrict__ nl, long n, FqRec o, u32* __restrict__ info)
{
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    u32 L = 0, Lc = 0, bad = 0;
    if (r < n) {
        const u32 s0 = r == 0 ? 0u : nl[4 * r - 1] + 1u;     // start of the name line
        const u32 e0 = nl[4 * r], e1 = nl[4 * r + 1], e2 = nl[4 * r + 2], e3 = nl[4 * r + 3];
        const u32 sl = e1 - (e0 + 1u);
        u32 ql = e3 - (e2 + 1u);
        if (ql > sl) ql = sl;
        const u32 nml = e0 - s0;
        o.name_off[r] = s0; o.name_len[r] = (u16)(nml > 0xffffu ? 0xffffu : nml);
        o.seq_off[r] = e0 + 1u; o.qual_off[r] = e2 + 1u;
        if (sl < 1u || sl > (u32)BMBS_MAX_READ) { bad = (u32)r + 1u; o.seq_len[r] = 1; o.qual_len[r] = 0; }
        else { o.seq_len[r] = (u16)sl; o.qual_len[r] = (u16)ql; L = sl; Lc = ~sl; }
    }
    for (int of = 32; of > 0; of >>= 1) {
        const u32 a = __shfl_down(L, of, 64), b = __shfl_down(Lc, of, 64), c = __shfl_down(bad, of, 64);
        L = a > L ? a : L; Lc = b > Lc ? b : Lc; bad = c > bad ? c : bad;
    }
    if ((threadIdx.x & 63) == 0) {
        if (L) atomicMax(&info[0], L);
        if (Lc) atomicMax(&info[1], Lc);
        if (bad) atomicMax(&info[2], bad);
    }
}

// ---- FASTQ text -> read rows (bmbs_map_*_fastq) --------------------------------------------------------------------------------
// What inputReads_single_directly / inputReads_paired_directly (Process_Reads.cpp:810-890, 155-317) do per record on the host --
// cut the sequence and quality lines out of the text, upper-case the bases, pad short quality lines with ' ', reverse-complement
// mate 2 (and every read of a --pbat library, with mirrored qualities) -- done here for a whole batch from the FASTQ text as it was
// read from the file: the host only finds the line starts.  One thread per 16-byte piece of an output row.
__global__ void __launch_bounds__(256)
k_fastq_rows(const char* __restrict__ text, const u32* __restrict__ seq_off, const u32* __restrict__ qual_off,
             const u16* __restrict__ seq_len, const u16* __restrict__ qual_len, long n, int stride, int rc_seq, int rev_qual,
             char* __restrict__ seq_out, char* __restrict__ qual_out, u16* __restrict__ len_out)
{
    const long i16 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = stride / 16;
    if (i16 >= n * per_row) return;
    const long r = i16 / per_row;
    const int j0 = (int)(i16 - r * per_row) * 16;
    const int L = seq_len[r];
    if (len_out && j0 == 0) len_out[r] = (u16)L;
    const char* src = text + seq_off[r];
    unsigned char o[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int j = j0 + t;
        unsigned char c = 0;
        if (j < L) {
            c = (unsigned char)src[rc_seq ? L - 1 - j : j];
            if (c >= 'a' && c <= 'z') c -= 32;
            if (rc_seq) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;      // rc_table, Process_Reads.cpp:1603
        }
        o[t] = c;
    }
    uint4 v;
    v.x = o[0] | (o[1] << 8) | (o[2] << 16) | ((u32)o[3] << 24);
    v.y = o[4] | (o[5] << 8) | (o[6] << 16) | ((u32)o[7] << 24);
    v.z = o[8] | (o[9] << 8) | (o[10] << 16) | ((u32)o[11] << 24);
    v.w = o[12] | (o[13] << 8) | (o[14] << 16) | ((u32)o[15] << 24);
    reinterpret_cast<uint4*>(seq_out)[i16] = v;
    if (!qual_out) return;
    const int ql = qual_len[r];
    const char* qs = text + qual_off[r];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int j = j0 + t;
        unsigned char c = 0;
        if (j < L) { const int jj = rev_qual ? L - 1 - j : j; c = jj < ql ? (unsigned char)qs[jj] : (unsigned char)' '; }
        o[t] = c;
    }
    v.x = o[0] | (o[1] << 8) | (o[2] << 16) | ((u32)o[3] << 24);
    v.y = o[4] | (o[5] << 8) | (o[6] << 16) | ((u32)o[7] << 24);
    v.z = o[8] | (o[9] << 8) | (o[10] << 16) | ((u32)o[11] << 24);
    v.w = o[12] | (o[13] << 8) | (o[14] << 16) | ((u32)o[15] << 24);
    reinterpret_cast<uint4*>(qual_out)[i16] = v;
}

// ---- records -> SAM text ---------------------------------------------------------------------------------------------------------
#define BMBS_TEXT_PBAT      1     // single end: the reads were mapped as their reverse complement (--pbat)
#define BMBS_TEXT_UNMAPPED  2     // --unmapped_out
#define BMBS_TEXT_AMBIG     4     // --ambiguous_out (the records of ambiguous reads carry an alignment)
#define BMBS_TEXT_PE        8

struct SamIn {
    const char* text[2];                 // FASTQ text of mate 1 / mate 2 (SE: [0] only)
    FqRec rec[2];
    const bmbs_result_dev* res;          // SE: [n]; PE: [2n] (mate 1, mate 2 of pair i at 2i, 2i+1)
    const u32* cigar;
    const char* chrom_chars; const u32* chrom_off;      // RNAME table: name c = chrom_chars[chrom_off[c] .. chrom_off[c+1])
    long n;                              // records (pairs)
    int flags;
};

DEVI int dec_digits(u64 v)
{
    int d = 1;
    if (v <= 0xffffffffull) { u32 w = (u32)v; while (w >= 10u) { w /= 10u; d++; } return d; }
    while (v >= 10) { v /= 10; d++; }
    return d;
}
DEVI char* put_dec(char* p, u64 v)
{
    const int d = dec_digits(v);
    char* q = p + d;
    if (v <= 0xffffffffull) { u32 w = (u32)v; do { *--q = (char)('0' + w % 10u); w /= 10u; } while (w); }
    else do { *--q = (char)('0' + (int)(v % 10)); v /= 10; } while (v);
    return p + d;
}
DEVI char* put_str(char* p, const char* s, int n) { for (int i = 0; i < n; i++) p[i] = s[i]; return p + n; }

// what one output line is made of.  kind: 0 nothing is printed, 1 mapped, 2 unmapped (flag 4 / 77 / 141)
struct SamLine {
    int kind, mate;          // mate: 0 / 1 = which FASTQ text the SEQ / QUAL / name come from
    long rec;                // record index in that text
    int name_skip, name_len; // QNAME = text[name_off + name_skip .. + name_len)
    bool rc;                 // SEQ reverse-complemented, QUAL reversed
    const bmbs_result_dev* x; const bmbs_result_dev* mate_x;
};

// QNAME of a single-end record: cut at the first ' ' or '/', a leading '@' dropped (Process_Reads.cpp:843-850)
DEVI void qname_se(const char* nm, int nl, int& skip, int& len)
{
    int c = 0;
    while (c < nl && nm[c] != ' ' && nm[c] != '/') c++;
    skip = 0; len = c;
    if (len && nm[0] == '@') { skip = 1; len--; }
}
// ... of a pair: up to the first character where the two names differ, ' ' or '/' (Process_Reads.cpp:296-307)
DEVI void qname_pe(const char* a, int la, const char* b, int lb, int& skip, int& len)
{
    int c = 0;
    while (c < la && c < lb && a[c] == b[c] && a[c] != ' ' && a[c] != '/') c++;
    skip = 0; len = c;
    if (len && a[0] == '@') { skip = 1; len--; }
}

// everything of a line but its QNAME (name_skip / name_len): global loads of the records only
DEVI SamLine sam_line_core(const SamIn& in, long line)
{
    SamLine s; s.kind = 0; s.mate = 0; s.rec = 0; s.name_skip = 0; s.name_len = 0; s.rc = false; s.x = nullptr; s.mate_x = nullptr;
    const bool pe = (in.flags & BMBS_TEXT_PE) != 0, amb_out = (in.flags & BMBS_TEXT_AMBIG) != 0, unm_out = (in.flags & BMBS_TEXT_UNMAPPED) != 0;
    if (!pe) {
        const long r = line;
        const bmbs_result_dev* x = in.res + r;
        const int st = x->status;
        const bool mapped = st == 1 || (st == 2 && amb_out);
        if (!mapped && !(unm_out && st != 2)) return s;
        s.kind = mapped ? 1 : 2; s.rec = r; s.x = x;
        // a --pbat read was mapped as the reverse complement of the text: flag 16 prints the text as it is (Schema.cpp:25538-25543)
        s.rc = mapped && (((x->flag & 16) != 0) != ((in.flags & BMBS_TEXT_PBAT) != 0));
        return s;
    }
    const long p = line >> 1;
    const int m = (int)(line & 1);
    const bmbs_result_dev* x1 = in.res + 2 * p;
    const bmbs_result_dev* x2 = x1 + 1;
    const int st = x1->status;
    const bool mapped = st == 1 || (st == 2 && amb_out);
    if (!mapped && !(unm_out && st != 2)) return s;
    s.kind = mapped ? 1 : 2; s.mate = m; s.rec = p; s.x = m ? x2 : x1; s.mate_x = m ? x1 : x2;
    // mate 1 prints as read unless it mapped to the reverse strand (flag 83: no 0x20); mate 2's text is in FASTQ orientation
    if (mapped) s.rc = m == 0 ? !(x1->flag & 32) : (x2->flag & 16) != 0;
    return s;
}
DEVI SamLine sam_line(const SamIn& in, long line)
{
    SamLine s = sam_line_core(in, line);
    if (!s.kind) return s;
    if (!(in.flags & BMBS_TEXT_PE)) qname_se(in.text[0] + in.rec[0].name_off[s.rec], in.rec[0].name_len[s.rec], s.name_skip, s.name_len);
    else qname_pe(in.text[0] + in.rec[0].name_off[s.rec], in.rec[0].name_len[s.rec], in.text[1] + in.rec[1].name_off[s.rec], in.rec[1].name_len[s.rec], s.name_skip, s.name_len);
    return s;
}

// the columns between QNAME and SEQ, and the ones behind QUAL; returns the length (p == nullptr: count only)
DEVI int sam_head(const SamIn& in, const SamLine& s, char* p)
{
    const bool pe = (in.flags & BMBS_TEXT_PE) != 0;
    char* const p0 = p;
    int len = 0;
    if (s.kind == 2) {
        // output_sam_unmapped (Schema.cpp:23955-23975), directly_output_unmapped_PE (10392-10430)
        const char* lit = !pe ? "\t4\t*\t0\t0\t*\t*\t0\t0\t" : s.mate == 0 ? "\t77\t*\t0\t0\t*\t*\t0\t0\t" : "\t141\t*\t0\t0\t*\t*\t0\t0\t";
        const int n = (int)(!pe ? sizeof("\t4\t*\t0\t0\t*\t*\t0\t0\t") : s.mate == 0 ? sizeof("\t77\t*\t0\t0\t*\t*\t0\t0\t") : sizeof("\t141\t*\t0\t0\t*\t*\t0\t0\t")) - 1;
        if (p) put_str(p, lit, n);
        return n;
    }
    const bmbs_result_dev& x = *s.x;
    const u32 c0 = in.chrom_off[x.chrom], c1 = in.chrom_off[x.chrom + 1];
    const int L = in.rec[s.mate].seq_len[s.rec];
    // \t flag \t chrom \t pos \t mapq \t
    len = 1 + dec_digits(x.flag) + 1 + (int)(c1 - c0) + 1 + dec_digits(x.pos) + 1 + dec_digits(x.mapq) + 1;
    if (p) {
        *p++ = '\t'; p = put_dec(p, x.flag); *p++ = '\t'; p = put_str(p, in.chrom_chars + c0, (int)(c1 - c0)); *p++ = '\t';
        p = put_dec(p, x.pos); *p++ = '\t'; p = put_dec(p, x.mapq); *p++ = '\t';
    }
    // CIGAR
    if (x.n_cigar == 0) { len += dec_digits((u64)L) + 1; if (p) { p = put_dec(p, (u64)L); *p++ = 'M'; } }
    else {
        for (int i = 0; i < x.n_cigar; i++) {
            const u32 o = in.cigar[x.cigar_off + i];
            len += dec_digits(o >> 4) + 1;
            if (p) { p = put_dec(p, o >> 4); *p++ = "MDISH"[o & 7u]; }
        }
    }
    if (!pe) { len += 7; if (p) p = put_str(p, "\t*\t0\t0\t", 7); }
    else {
        // \t = \t pnext \t [-]tlen \t : TLEN sign, Schema.cpp:10575-10600 (mate 1: negative only if the mate lies to the left),
        // 11530-11555 (mate 2: positive only if the mate lies to the right)
        const bmbs_result_dev& y = *s.mate_x;
        const bool neg = s.mate == 0 ? (y.pos < x.pos) : !(y.pos > x.pos);
        const u32 tl = s.mate == 0 ? x.tlen : y.tlen;
        len += 3 + dec_digits(y.pos) + 1 + (neg ? 1 : 0) + dec_digits(tl) + 1;
        if (p) { p = put_str(p, "\t=\t", 3); p = put_dec(p, y.pos); *p++ = '\t'; if (neg) *p++ = '-'; p = put_dec(p, tl); *p++ = '\t'; }
    }
    (void)p0;
    return len;
}
DEVI int sam_tail(const SamLine& s, char* p)
{
    if (s.kind == 2) { if (p) *p = '\n'; return 1; }
    const int len = 6 + dec_digits(s.x->nm) + 1;
    if (p) { p = put_str(p, "\tNM:i:", 6); p = put_dec(p, s.x->nm); *p++ = '\n'; }
    return len;
}

// bytes of output line `line` (0: nothing is printed for it)
__global__ void __launch_bounds__(256)
k_sam_len(SamIn in, long n_lines, u32* __restrict__ len_out)
{
    const long line = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= n_lines) return;
    const SamLine s = sam_line(in, line);
    u32 len = 0;
    if (s.kind) {
        const int L = in.rec[s.mate].seq_len[s.rec];
        len = (u32)(s.name_len + sam_head(in, s, nullptr) + 2 * L + 1 + sam_tail(s, nullptr));
    }
    len_out[line] = len;
}

#endif
