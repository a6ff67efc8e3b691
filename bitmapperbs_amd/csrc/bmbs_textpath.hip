// bitmapperbs_amd/csrc/bmbs_textpath.hip -- the two ends of the file path, their kernels and their entry points (a translation unit of its
// own since round 5): FASTQ text -> line index -> read rows (bmbs_map_*_fastq, bmbs_map_*_text), BGZF blocks inflated on the device
// (bmbs_inflate_bgzf, bmbs_text_open_bgzf / bmbs_text_map_open), records -> SAM text or BAM records in BGZF blocks.  The mapping in
// between is bmbs_api.hip's (lane_enqueue / lane_settle).  Kernels: bmbs_text.hip, bmbs_bam.hip, bmbs_inflate.hip.
#include "bmbs_host.h"
#define DEVI __device__ __forceinline__
#include "bmbs_text.hip"
#include "bmbs_bam.hip"
#include "bmbs_inflate.hip"

// the constants the kernels of this file read: x^(2^n) mod P of CRC-32 for the BGZF blocks' CRCs (bmbs_bytes.h: crc_x8n); per device
void textpath_device_init(Lane* c)
{
    (void)c;
    u32 x2n[32];
    auto mult = [](u32 a, u32 b) { u32 m = 1u << 31, p = 0; for (;;) { if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; } m >>= 1; b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1; } return p; };
    u32 p = 1u << 30;
    x2n[0] = p;
    for (int n = 1; n < 32; n++) { p = mult(p, p); x2n[n] = p; }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_x2n), x2n, sizeof x2n);
}

// ------------------------------------------------------------------------------------------------
// FASTQ text in (the host only finds the line starts; the rows are cut out of the text on the device)
struct FqDev { const char* text; const u32* seq_off; const u32* qual_off; const u16* seq_len; const u16* qual_len; };

int fastq_check(Lane* c, const bmbs_fastq_view* v, int64_t n, int L_max)
{
    if (!v || !v->text || !v->seq_off || !v->qual_off || !v->seq_len || !v->qual_len) { c->err = "fastq view: NULL field"; return BMBS_EINVAL; }
    if (v->text_bytes >= (1ull << 32)) { c->err = "fastq view: a text window has to be smaller than 4 GiB (32-bit offsets)"; return BMBS_EINVAL; }
    // every line the device is going to read lies inside the window (a bad index must not become an out-of-bounds device read)
    for (int64_t i = 0; i < n; i++) {
        const uint64_t sl = v->seq_len[i], ql = v->qual_len[i];
        if (sl < 1 || sl > (uint64_t)L_max || ql > sl || (uint64_t)v->seq_off[i] + sl > v->text_bytes || (uint64_t)v->qual_off[i] + ql > v->text_bytes) {
            c->err = "fastq view: record " + std::to_string(i) + " has a line outside the text window or a length outside 1..L_max";
            return BMBS_EINVAL;
        }
    }
    return BMBS_OK;
}
// text + index arrays of one file to the device; idx_at = byte offset of this file's arrays inside c->fq_idx
int fastq_upload(Lane* c, DevBuf& dtext, const bmbs_fastq_view* v, u64 n, u64 idx_at, FqDev& out)
{
    ENS(c, dtext, v->text_bytes + 64);
    HIPCHK(c, hipMemcpyAsync(dtext.p, v->text, v->text_bytes, hipMemcpyHostToDevice, c->stream));
    char* base = c->fq_idx.as<char>() + idx_at;
    HIPCHK(c, hipMemcpyAsync(base, v->seq_off, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(base + n * 4, v->qual_off, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(base + n * 8, v->seq_len, n * 2, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(base + n * 10, v->qual_len, n * 2, hipMemcpyHostToDevice, c->stream));
    out.text = dtext.as<char>(); out.seq_off = reinterpret_cast<const u32*>(base); out.qual_off = reinterpret_cast<const u32*>(base + n * 4);
    out.seq_len = reinterpret_cast<const u16*>(base + n * 8); out.qual_len = reinterpret_cast<const u16*>(base + n * 10);
    return BMBS_OK;
}

static int lane_map_se_fastq(Lane* c, const bmbs_fastq_view* reads, int64_t n_reads, int32_t L_max, int32_t uniform, int32_t pbat,
                                 bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used)
{
    if (!c) return BMBS_EINVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    if (n_cigar_used) *n_cigar_used = 0;
    if (n_reads <= 0) return BMBS_OK;
    if (L_max <= 0 || L_max > BMBS_MAX_READ) { c->err = "bad read geometry"; return BMBS_EINVAL; }
    { const int r0 = fastq_check(c, reads, n_reads, L_max); if (r0) return r0; }
    const u64 n = (u64)n_reads;
    const int ds = (L_max + 15) / 16 * 16;
    const int k = threshold_k(c->prm, L_max);
    const u64 pool = n * (u64)cigar_ops_bound(c->prm, L_max, k);
    ENS(c, c->out_res, n * 32); ENS(c, c->cig_pool, pool * 4);
    ENS(c, c->in_seq, n * (u64)ds + 64); ENS(c, c->in_qual, n * (u64)ds + 64); ENS(c, c->in_len, n * 2 + 16);
    ENS(c, c->fq_idx, n * 12 + 64);
    FqDev f;
    { const int r1 = fastq_upload(c, c->fq_text1, reads, n, 0, f); if (r1) return r1; }
    hipLaunchKernelGGL(k_fastq_rows, dim3(nblk(n * (u64)(ds / 16), 256)), dim3(256), 0, c->stream, f.text, f.seq_off, f.qual_off, f.seq_len,
                       f.qual_len, (long)n, ds, pbat ? 1 : 0, pbat ? 1 : 0, c->in_seq.as<char>(), c->in_qual.as<char>(), c->in_len.as<u16>());
    Pending P;
    P.pe = false; P.L = L_max; P.stride = ds; P.n = n_reads;
    P.a[0] = (uint64_t)c->in_seq.p; P.a[1] = (uint64_t)c->in_qual.p; P.d_len = uniform ? nullptr : c->in_len.as<u16>();
    P.d_results = (uint64_t)c->out_res.p; P.d_cigar_pool = (uint64_t)c->cig_pool.p; P.cigar_cap = (int64_t)pool;
    int rc = lane_enqueue(c, P, true);
    if (rc) return rc;
    rc = lane_settle(c);                         // counts (and, rarely, the repeat with exact sizes) before anything is copied back
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(results, c->out_res.p, n * 32, hipMemcpyDeviceToHost, c->stream));
    const u64 used = c->last_n_jobs * (u64)c->last_max_ops;
    if (used > (u64)cigar_cap) { c->err = "host cigar pool too small"; (void)hipStreamSynchronize(c->stream); return BMBS_ENOMEM; }
    if (used) HIPCHK(c, hipMemcpyAsync(cigar_pool, c->cig_pool.p, used * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_cigar_used) *n_cigar_used = (int64_t)used;
    return BMBS_OK;
}

static int lane_map_pe_fastq(Lane* c, const bmbs_fastq_view* mate1, const bmbs_fastq_view* mate2, int64_t n_pairs, int32_t L_max,
                                 int32_t uniform, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used)
{
    if (!c) return BMBS_EINVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    if (n_cigar_used) *n_cigar_used = 0;
    if (n_pairs <= 0) return BMBS_OK;
    if (L_max <= 0 || L_max > BMBS_MAX_READ) { c->err = "bad read geometry"; return BMBS_EINVAL; }
    { int r0 = fastq_check(c, mate1, n_pairs, L_max); if (r0) return r0; r0 = fastq_check(c, mate2, n_pairs, L_max); if (r0) return r0; }
    const u64 n = (u64)n_pairs, n2 = 2 * n;
    const int ds = (L_max + 15) / 16 * 16;
    const int k = threshold_k(c->prm, L_max);
    const u64 pool = n2 * (u64)cigar_ops_bound(c->prm, L_max, k);
    ENS(c, c->out_res, n2 * 32); ENS(c, c->cig_pool, pool * 4);
    ENS(c, c->pe_seq, n2 * (u64)ds + 64); ENS(c, c->in_qual, n * (u64)ds + 64); ENS(c, c->in_qual2, n * (u64)ds + 64); ENS(c, c->in_len, n2 * 2 + 16);
    ENS(c, c->fq_idx, n2 * 12 + 128);
    FqDev f1, f2;
    { int r1 = fastq_upload(c, c->fq_text1, mate1, n, 0, f1); if (r1) return r1; r1 = fastq_upload(c, c->fq_text2, mate2, n, (n * 12 + 63) & ~63ull, f2); if (r1) return r1; }
    // rows 0..n-1 = mate 1 as read, rows n..2n-1 = mate 2 reverse-complemented; the qualities stay in FASTQ order (qual_row)
    char* seq_all = c->pe_seq.as<char>();
    const unsigned g = nblk(n * (u64)(ds / 16), 256);
    hipLaunchKernelGGL(k_fastq_rows, dim3(g), dim3(256), 0, c->stream, f1.text, f1.seq_off, f1.qual_off, f1.seq_len, f1.qual_len, (long)n, ds, 0, 0,
                       seq_all, c->in_qual.as<char>(), c->in_len.as<u16>());
    hipLaunchKernelGGL(k_fastq_rows, dim3(g), dim3(256), 0, c->stream, f2.text, f2.seq_off, f2.qual_off, f2.seq_len, f2.qual_len, (long)n, ds, 1, 0,
                       seq_all + n * (u64)ds, c->in_qual2.as<char>(), c->in_len.as<u16>() + n);
    Pending P;
    P.pe = true; P.L = L_max; P.stride = ds; P.n = n_pairs; P.prepared = true;
    P.a[0] = (uint64_t)seq_all; P.a[1] = (uint64_t)c->in_qual.p; P.a[2] = (uint64_t)(seq_all + n * (u64)ds); P.a[3] = (uint64_t)c->in_qual2.p;
    P.d_len = uniform ? nullptr : c->in_len.as<u16>();
    P.d_results = (uint64_t)c->out_res.p; P.d_cigar_pool = (uint64_t)c->cig_pool.p; P.cigar_cap = (int64_t)pool;
    int rc = lane_enqueue(c, P, true);
    if (rc) return rc;
    rc = lane_settle(c);                         // counts (and, rarely, the repeat with exact sizes) before anything is copied back
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(results, c->out_res.p, n2 * 32, hipMemcpyDeviceToHost, c->stream));
    const u64 used = c->last_n_jobs * (u64)c->last_max_ops;
    if (used > (u64)cigar_cap) { c->err = "host cigar pool too small"; (void)hipStreamSynchronize(c->stream); return BMBS_ENOMEM; }
    if (used) HIPCHK(c, hipMemcpyAsync(cigar_pool, c->cig_pool.p, used * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_cigar_used) *n_cigar_used = (int64_t)used;
    return BMBS_OK;
}

// ------------------------------------------------------------------------------------------------
// FASTQ text in, SAM text out (bmbs_text.hip): the host reads and writes files, everything between is on the device
// newline index of one text window (already on the device): positions of its first 4 * n_cap newlines; the totals slot 16 + f gets
// the number of lines found
int text_index(Lane* c, DevBuf& dtext, u64 bytes, u64 n_cap, int f)
{
    const u64 tiles = (bytes + FQ_TILE_BYTES - 1) / FQ_TILE_BYTES;
    ENS(c, c->tx_tilecnt, tiles * 4 + 64); ENS(c, c->tx_tileoff, (tiles + 1) * 8 + 64);
    ENS(c, c->tx_nl[f], (4 * n_cap + 8) * 4);
    hipLaunchKernelGGL(k_fq_count, dim3((unsigned)tiles), dim3(FQ_TILE_THREADS), 0, c->stream, dtext.as<char>(), bytes, c->tx_tilecnt.as<u32>());
    int rc = scan_u32(c, c->tx_tilecnt.as<u32>(), tiles, c->tx_tileoff.as<u64>(), 16 + f);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fq_lines, dim3((unsigned)tiles), dim3(FQ_TILE_THREADS), 0, c->stream, dtext.as<char>(), bytes, c->tx_tileoff.as<u64>(), 4 * n_cap, c->tx_nl[f].as<u32>());
    return BMBS_OK;
}
// the per-record field arrays of n records of file f
int fq_rec_setup(Lane* c, u64 n, int f, FqRec& rec)
{
    const u64 n4 = (n * 4 + 63) & ~63ull, n2 = (n * 2 + 63) & ~63ull;
    ENS(c, c->tx_rec[f], 3 * n4 + 3 * n2 + 64);
    char* b = c->tx_rec[f].as<char>();
    rec.seq_off = reinterpret_cast<u32*>(b); rec.qual_off = reinterpret_cast<u32*>(b + n4); rec.name_off = reinterpret_cast<u32*>(b + 2 * n4);
    rec.seq_len = reinterpret_cast<u16*>(b + 3 * n4); rec.qual_len = reinterpret_cast<u16*>(b + 3 * n4 + n2); rec.name_len = reinterpret_cast<u16*>(b + 3 * n4 + 2 * n2);
    return BMBS_OK;
}
// One upload and one download at a time per device, whatever the number of contexts: concurrent copies in one direction share the
// link badly (tools/pcie_probe: 48 GB/s each way with one stream per direction, 33 with three), and a context's copies are long
// enough (hundreds of MB) to fill the link on their own.  Held from the first copy of a phase until the wait that ends it.
std::mutex g_h2d_mu[16];
std::mutex g_d2h_mu[16];
// The results of a text call go down the lane's own copy stream under the device's download mutex: one copy per direction and device at
// a time (two at once share the link and both finish late).  Round 5 tried ONE download stream per device shared by all contexts, the
// copies queued back to back without the host in between: 320 M reads to /dev/null took 3.3-3.8 s instead of 2.9-3.05 s, whether the
// copy waited on the stream for its kernel or was queued once the kernel had ended (tools/e2e_quick.sh, same box, alternating runs).
// *copy_s = seconds the copy held the link.
int d2h_chunked(Lane* c, char* dst, const char* src, u64 bytes, hipStream_t st);
extern std::mutex g_d2h_mu[16];
int download_locked(Lane* c, char* dst, const char* src, u64 bytes, hipStream_t after, double* copy_s)
{
    hipStream_t ds = c->down_stream ? c->down_stream : c->stream;
    HIPCHK(c, hipStreamSynchronize(after));                  // the bytes are complete before the link is claimed
    std::lock_guard<std::mutex> down(g_d2h_mu[c->dev & 15]);
    timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = d2h_chunked(c, dst, src, bytes, ds);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(ds));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (copy_s) *copy_s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    return BMBS_OK;
}

// D2H in pieces: one 2 GiB device-to-host copy ran at a quarter of the link rate on the MI355X boxes (tools/pcie_probe)
int d2h_chunked(Lane* c, char* dst, const char* src, u64 bytes, hipStream_t st)
{
    const u64 piece = 128ull << 20;
    for (u64 o = 0; o < bytes; o += piece) HIPCHK(c, hipMemcpyAsync(dst + o, src + o, std::min(piece, bytes - o), hipMemcpyDeviceToHost, st));
    return BMBS_OK;
}

int lane_text_finish(Lane* c, bool pe, u64 bytes1, u64 bytes2, int64_t n_records, int32_t flags_in, char* sam, u64 sam_cap, u64* sam_bytes,
                     int64_t* n_lines_out, double t_start, double t_uploaded);

int lane_map_text(Lane* c, bool pe, const char* text1, u64 bytes1, const char* text2, u64 bytes2, int64_t n_records, int32_t flags_in, char* sam,
                  u64 sam_cap, u64* sam_bytes, int64_t* n_lines_out)
{
    if (!c) return BMBS_EINVAL;
    if (sam_bytes) *sam_bytes = 0;
    if (n_lines_out) *n_lines_out = 0;
    if (!c->attached) { c->err = "no index attached"; return BMBS_ESTATE; }
    if (c->n_refs != c->ix.n_chrom) { c->err = "bmbs_sam_refs has not been given the index's reference names"; return BMBS_ESTATE; }
    if (n_records <= 0) return BMBS_OK;
    if (!text1 || (pe && !text2) || !sam) { c->err = "text call: NULL buffer"; return BMBS_EINVAL; }
    if (bytes1 >= (1ull << 32) || bytes2 >= (1ull << 32)) { c->err = "a text window has to be smaller than 4 GiB (32-bit offsets)"; return BMBS_EINVAL; }
    HIPCHK(c, hipSetDevice(c->dev));
    { const int rs = lane_settle(c); if (rs) return rs; }
    c->open_text.valid = false;
    const u64 n = (u64)n_records;
    static const bool trace = getenv("BMBS_TEXT_TRACE") != nullptr;       // diagnostic: host-side phase times of every text call
    auto wall = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    const double t_start = wall();
    HIPCHK(c, hipMemsetAsync(c->tx_info.p, 0, 64, c->stream));
    int rc;
    ENS(c, c->fq_text1, bytes1 + 64);
    if (pe) ENS(c, c->fq_text2, bytes2 + 64);
    {
        // the link is claimed for the copies alone: the kernels behind them may have to queue behind other contexts' kernels
        std::unique_lock<std::mutex> up(g_h2d_mu[c->dev & 15], std::defer_lock);
        if (c->kn.copy_lock) up.lock();
        const double t_up0 = wall();
        // on a stream of their own that never carries a kernel (the lane has nothing in flight here: the previous call ended with a wait)
        hipStream_t us = c->kn.copy_streams && c->up_stream ? c->up_stream : c->stream;
        HIPCHK(c, hipMemcpyAsync(c->fq_text1.p, text1, bytes1, hipMemcpyHostToDevice, us));
        if (pe) HIPCHK(c, hipMemcpyAsync(c->fq_text2.p, text2, bytes2, hipMemcpyHostToDevice, us));
        if (c->kn.copy_lock || us != c->stream) { HIPCHK(c, hipStreamSynchronize(us)); c->link_up_s += wall() - t_up0; }
    }
    const double t_uploaded = trace ? wall() : 0;
    // diagnostic (tools/e2e_trace.sh): BMBS_TEXT_COPY_ONLY=1 moves the bytes of a batch over the link and runs nothing in between
    static const bool copy_only = getenv("BMBS_TEXT_COPY_ONLY") != nullptr;
    if (copy_only) {
        const u64 total = std::min<u64>(sam_cap, (bytes1 + bytes2) * 115 / 100);
        ENS(c, c->sam_out, total + 64);
        std::unique_lock<std::mutex> down(g_d2h_mu[c->dev & 15], std::defer_lock);
        if (c->kn.copy_lock) down.lock();
        rc = d2h_chunked(c, sam, c->sam_out.as<char>(), total, c->stream);
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (sam_bytes) *sam_bytes = 0;
        if (trace) fprintf(stderr, "[text] copy only: upload %.2f download %.2f ms\n", (t_uploaded - t_start) * 1e3, (wall() - t_uploaded) * 1e3);
        return BMBS_OK;
    }
    rc = text_index(c, c->fq_text1, bytes1, n, 0);
    if (rc) return rc;
    if (pe) { rc = text_index(c, c->fq_text2, bytes2, n, 1); if (rc) return rc; }
    return lane_text_finish(c, pe, bytes1, bytes2, n_records, flags_in, sam, sam_cap, sam_bytes, n_lines_out, t_start, t_uploaded);
}

// k_line_write's launch shape: lines per workgroup and the LDS it gets for the image of its piece of the output and for the staged
// FASTQ text, from the window's bytes per record.  `hb` bounds the computed columns of one line (the plain path keeps them in a
// quarter of the image buffer).  BMBS_TXW_TINY=1 (test aid): staging buffers too small for anything -- every workgroup takes the plain path.
struct TxwPlan { int lpb; u32 out_cap, src_cap; size_t lds; };
static TxwPlan txw_plan(bool pe, u64 text_bytes, u64 n_lines, int hb)
{
    static const bool tiny = getenv("BMBS_TXW_TINY") != nullptr;
    TxwPlan t = {0, 0, 0, 0};
    if (4 * (u64)hb > 30 * 1024) return t;
    const u64 rec = text_bytes / std::max<u64>(n_lines, 1) + 8, line = rec + 96;
    const u64 out_max = 30 * 1024, src_max = pe ? 15 * 1024 : 30 * 1024;
    int lpb = 32;
    auto need_out = [&](int l) { return ((u64)l * line * 5 / 4 + 64 + 15) & ~15ull; };
    auto need_src = [&](int l) { return ((u64)(pe ? l / 2 : l) * rec * 5 / 4 + 64 + 15) & ~15ull; };
    while (lpb > 2 && (need_out(lpb) > out_max || need_src(lpb) > src_max)) lpb /= 2;
    t.lpb = lpb;
    t.out_cap = (u32)std::min<u64>(std::max<u64>(need_out(lpb), (4 * (u64)hb + 15) & ~15ull), out_max);
    t.src_cap = tiny ? 16u : (u32)std::min<u64>(need_src(lpb), src_max);
    t.lds = (size_t)t.out_cap + 16 + (pe ? 2 : 1) * ((size_t)t.src_cap + 16);
    return t;
}

// the text window(s) are on the device (fq_text1 / fq_text2) and their newline positions are being indexed (text_index): record fields,
// rows, mapping, SAM text or BAM blocks, download
int lane_text_finish(Lane* c, bool pe, u64 bytes1, u64 bytes2, int64_t n_records, int32_t flags_in, char* sam, u64 sam_cap, u64* sam_bytes,
                     int64_t* n_lines_out, double t_start, double t_uploaded)
{
    const u64 n = (u64)n_records, n2 = pe ? 2 * n : n;
    static const bool trace = getenv("BMBS_TEXT_TRACE") != nullptr;
    auto wall = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    double tp[8] = {t_start, 0, 0, 0, 0, 0, 0, t_uploaded}, t_uplock = t_start, t_dnlock = 0, t_dnstart = 0;
    FqRec rec[2] = {};
    int rc = fq_rec_setup(c, n, 0, rec[0]);
    if (rc) return rc;
    if (pe) { rc = fq_rec_setup(c, n, 1, rec[1]); if (rc) return rc; }
    // records can only be cut out once the host knows that every one of them is complete: the line counts first
    HIPCHK(c, hipMemcpyAsync(c->h_info + 16, c->totals.as<u64>() + 16, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    tp[1] = wall();
    const u64* lines = reinterpret_cast<const u64*>(c->h_info + 16);
    if (lines[0] < 4 * n || (pe && lines[1] < 4 * n)) { c->err = "text call: a window holds fewer than 4 lines per record"; return BMBS_EINVAL; }
    hipLaunchKernelGGL(k_fq_records, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->tx_nl[0].as<u32>(), (long)n, rec[0], c->tx_info.as<u32>());
    if (pe) hipLaunchKernelGGL(k_fq_records, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->tx_nl[1].as<u32>(), (long)n, rec[1], c->tx_info.as<u32>() + 4);
    HIPCHK(c, hipMemcpyAsync(c->h_info, c->tx_info.p, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    tp[2] = wall();
    const u32* inf = c->h_info;
    for (int f = 0; f < (pe ? 2 : 1); f++)
        if (inf[4 * f + 2]) {
            c->err = "record " + std::to_string(inf[4 * f + 2] - 1) + " of this batch has an empty or longer-than-" + std::to_string(BMBS_MAX_READ) + "-character sequence line: not supported (the reference's own buffers end there)";
            return BMBS_EINVAL;
        }
    int maxL = (int)inf[0], minL = (int)~inf[1];
    if (pe) { maxL = std::max(maxL, (int)inf[4]); minL = std::min(minL, (int)~inf[5]); }
    const bool uniform = minL == maxL;
    const int ds = (maxL + 15) / 16 * 16;
    const int max_ops = cigar_ops_bound(c->prm, maxL, threshold_k(c->prm, maxL));
    const u64 pool = n2 * (u64)max_ops;
    ENS(c, c->out_res, n2 * 32); ENS(c, c->cig_pool, pool * 4); ENS(c, c->in_len, n2 * 2 + 16);
    Pending P;
    P.pe = pe; P.L = maxL; P.stride = ds; P.n = n_records;
    const int pbat = (flags_in & BMBS_TEXT_PBAT) && !pe ? 1 : 0;
    if (!pe) {
        ENS(c, c->in_seq, n * (u64)ds + 64); ENS(c, c->in_qual, n * (u64)ds + 64);
        hipLaunchKernelGGL(k_fastq_rows, dim3(nblk(n * (u64)(ds / 16), 256)), dim3(256), 0, c->stream, c->fq_text1.as<char>(), rec[0].seq_off, rec[0].qual_off,
                           rec[0].seq_len, rec[0].qual_len, (long)n, ds, pbat, pbat, c->in_seq.as<char>(), c->in_qual.as<char>(), c->in_len.as<u16>());
        P.a[0] = (uint64_t)c->in_seq.p; P.a[1] = (uint64_t)c->in_qual.p;
    } else {
        ENS(c, c->pe_seq, n2 * (u64)ds + 64); ENS(c, c->in_qual, n * (u64)ds + 64); ENS(c, c->in_qual2, n * (u64)ds + 64);
        char* seq_all = c->pe_seq.as<char>();
        const unsigned g = nblk(n * (u64)(ds / 16), 256);
        hipLaunchKernelGGL(k_fastq_rows, dim3(g), dim3(256), 0, c->stream, c->fq_text1.as<char>(), rec[0].seq_off, rec[0].qual_off, rec[0].seq_len, rec[0].qual_len,
                           (long)n, ds, 0, 0, seq_all, c->in_qual.as<char>(), c->in_len.as<u16>());
        hipLaunchKernelGGL(k_fastq_rows, dim3(g), dim3(256), 0, c->stream, c->fq_text2.as<char>(), rec[1].seq_off, rec[1].qual_off, rec[1].seq_len, rec[1].qual_len,
                           (long)n, ds, 1, 0, seq_all + n * (u64)ds, c->in_qual2.as<char>(), c->in_len.as<u16>() + n);
        P.a[0] = (uint64_t)seq_all; P.a[1] = (uint64_t)c->in_qual.p; P.a[2] = (uint64_t)(seq_all + n * (u64)ds); P.a[3] = (uint64_t)c->in_qual2.p;
        P.prepared = true;
    }
    P.d_len = uniform ? nullptr : c->in_len.as<u16>();
    P.d_results = (uint64_t)c->out_res.p; P.d_cigar_pool = (uint64_t)c->cig_pool.p; P.cigar_cap = (int64_t)pool;
    // a caller whose output buffer turns out too small repeats the call (BMBS_ENOMEM below): the batch must not be counted twice
    const size_t stats_bytes = BMBS_SHARDS * BMBS_SHARD_WORDS * 8;
    ENS(c, c->stats_snap, stats_bytes);
    HIPCHK(c, hipMemcpyAsync(c->stats_snap.p, c->stats.p, stats_bytes, hipMemcpyDeviceToDevice, c->stream));
    auto too_small = [&](const char* what) -> int {
        (void)hipMemcpyAsync(c->stats.p, c->stats_snap.p, stats_bytes, hipMemcpyDeviceToDevice, c->stream);
        (void)hipStreamSynchronize(c->stream);
        c->err = what;
        return BMBS_ENOMEM;
    };
    rc = lane_enqueue(c, P, true);
    if (rc) return rc;
    rc = lane_settle(c);
    if (rc) return rc;
    tp[3] = wall();
    // ---- records -> SAM text
    SamIn in;
    in.text[0] = c->fq_text1.as<char>(); in.text[1] = pe ? c->fq_text2.as<char>() : nullptr;
    in.rec[0] = rec[0]; in.rec[1] = rec[1];
    in.res = c->out_res.as<bmbs_result_dev>(); in.cigar = c->cig_pool.as<u32>();
    in.chrom_chars = c->chrom_chars.as<char>(); in.chrom_off = c->chrom_off.as<u32>();
    in.n = (long)n;
    in.flags = (flags_in & (BMBS_TEXT_PBAT | BMBS_TEXT_UNMAPPED)) | (c->prm.ambiguous_out ? BMBS_TEXT_AMBIG : 0) | (pe ? BMBS_TEXT_PE : 0);
    ENS(c, c->sam_len, n2 * 4 + 64); ENS(c, c->sam_off, (n2 + 1) * 8 + 64);
    if (flags_in & BMBS_TEXT_BAM) {
        // ---- records -> BAM records -> BGZF blocks, all on the device (bmbs_bam.hip)
        prof_begin(c, "k_bam_len");
        hipLaunchKernelGGL(k_bam_len, dim3(nblk(n2, 256)), dim3(256), 0, c->stream, in, (long)n2, c->sam_len.as<u32>(), c->tx_info.as<u32>());
        rc = scan_u32(c, c->sam_len.as<u32>(), n2, c->sam_off.as<u64>(), 19);
        if (rc) return rc;
        prof_end(c);
        HIPCHK(c, hipMemcpyAsync(c->h_info + 24, c->totals.as<u64>() + 19, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_info, c->tx_info.p, 32, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        tp[4] = wall();
        const u64 raw_total = *reinterpret_cast<const u64*>(c->h_info + 24);
        if (n_lines_out) *n_lines_out = (int64_t)n2;
        if (c->h_info[3]) { c->err = "output line " + std::to_string(c->h_info[3] - 1) + " of this batch has a read name of more than 254 characters: BAM cannot hold it"; return BMBS_EINVAL; }
        if (!raw_total) return BMBS_OK;
        const u64 nb = (raw_total + BGZF_IN - 1) / BGZF_IN;
        ENS(c, c->bam_raw, raw_total + 256);
        const TxwPlan tw = txw_plan(pe, bytes1 + bytes2, n2, (36 + 4 * std::max(max_ops, 1) + 8 + 15) & ~15);
        if (!tw.lpb) { c->err = "text call: CIGARs too long"; return BMBS_EINVAL; }
        prof_begin(c, "k_bam_write");
        hipLaunchKernelGGL(k_line_write<true>, dim3(nblk(n2, (unsigned)tw.lpb)), dim3(TXW_THREADS), tw.lds, c->stream, in, (long)n2, c->sam_off.as<u64>(), tw.lpb, tw.out_cap, tw.src_cap,
                           c->bam_raw.as<char>());
        prof_end(c);
        ENS(c, c->bam_slots, nb * (u64)BGZF_SLOT);
        ENS(c, c->bam_slot_len, nb * 4 + 64); ENS(c, c->bam_off, (nb + 1) * 8 + 64);
        const size_t lds = (size_t)BGZF_THREADS * BGZF_PSTRIDE * 4;                      // the block's bytes (padded segments)
        HIPCHK(c, hipMemsetAsync(c->bam_slots.p, 0, nb * (u64)BGZF_SLOT, c->stream));      // (k_bgzf_block ORs the shared words of its stream into the slots)
        static std::once_flag lds_once[16];
        std::call_once(lds_once[c->dev & 15], [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bgzf_block), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        prof_begin(c, "k_bgzf_block");
        hipLaunchKernelGGL(k_bgzf_block, dim3((unsigned)nb), dim3(BGZF_THREADS), lds, c->stream, c->bam_raw.as<char>(), c->totals.as<u64>() + 19,
                           c->bam_slots.as<char>(), c->bam_slot_len.as<u32>());
        prof_end(c);
        rc = scan_u32(c, c->bam_slot_len.as<u32>(), nb, c->bam_off.as<u64>(), 20);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->h_info + 26, c->totals.as<u64>() + 20, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const u64 ztotal = *reinterpret_cast<const u64*>(c->h_info + 26);
        if (sam_bytes) *sam_bytes = ztotal;
        if (ztotal > sam_cap) return too_small("text call: the BAM buffer is too small (sam_bytes tells what this batch needs)");
        ENS(c, c->sam_out, ztotal + 64);
        prof_begin(c, "k_bgzf_gather");
        hipLaunchKernelGGL(k_bgzf_gather, dim3((unsigned)nb), dim3(256), 0, c->stream, c->bam_slots.as<char>(), c->bam_slot_len.as<u32>(), c->bam_off.as<u64>(), c->sam_out.as<char>());
        prof_end(c);
        if (trace) { HIPCHK(c, hipStreamSynchronize(c->stream)); tp[5] = wall(); }
        {
            double cs = 0;
            rc = download_locked(c, sam, c->sam_out.as<char>(), ztotal, c->stream, &cs);
            if (rc) return rc;
            c->link_down_s += cs;
        }
        tp[6] = wall();
        c->text_call_s += tp[6] - tp[0]; c->text_calls++;
        if (trace)
            fprintf(stderr, "[text/bam] n=%ld in=%.1fMB records=%.1fMB out=%.1fMB  upload %.2f lines+records %.2f rows+map %.2f  len+scan %.2f  write+deflate %.2f  download %.2f  total %.2f ms\n",
                    (long)n2, (double)(bytes1 + bytes2) / 1e6, (double)raw_total / 1e6, (double)ztotal / 1e6, (tp[7] - tp[0]) * 1e3, (tp[2] - tp[7]) * 1e3, (tp[3] - tp[2]) * 1e3, (tp[4] - tp[3]) * 1e3,
                    (tp[5] - tp[4]) * 1e3, (tp[6] - tp[5]) * 1e3, (tp[6] - tp[0]) * 1e3);
        return BMBS_OK;
    }
    prof_begin(c, "k_sam_len");
    hipLaunchKernelGGL(k_sam_len, dim3(nblk(n2, 256)), dim3(256), 0, c->stream, in, (long)n2, c->sam_len.as<u32>());
    rc = scan_u32(c, c->sam_len.as<u32>(), n2, c->sam_off.as<u64>(), 18);
    if (rc) return rc;
    prof_end(c);
    HIPCHK(c, hipMemcpyAsync(c->h_info + 24, c->totals.as<u64>() + 18, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    tp[4] = wall();
    const u64 total = *reinterpret_cast<const u64*>(c->h_info + 24);
    if (sam_bytes) *sam_bytes = total;
    if (n_lines_out) *n_lines_out = (int64_t)n2;
    if (total > sam_cap) return too_small("text call: the SAM buffer is too small (sam_bytes tells what this batch needs)");
    if (!total) return BMBS_OK;
    ENS(c, c->sam_out, total + 64);
    const TxwPlan tw = txw_plan(pe, bytes1 + bytes2, n2, (c->max_ref_len + 5 * std::max(max_ops, 1) + 96 + 15) & ~15);
    if (!tw.lpb) { c->err = "text call: reference names too long"; return BMBS_EINVAL; }
    prof_begin(c, "k_sam_write");
    hipLaunchKernelGGL(k_line_write<false>, dim3(nblk(n2, (unsigned)tw.lpb)), dim3(TXW_THREADS), tw.lds, c->stream, in, (long)n2, c->sam_off.as<u64>(), tw.lpb, tw.out_cap, tw.src_cap,
                       c->sam_out.as<char>());
    prof_end(c);
    if (trace) { HIPCHK(c, hipStreamSynchronize(c->stream)); tp[5] = wall(); }
    {
        double cs = 0;
        t_dnstart = wall();
        rc = download_locked(c, sam, c->sam_out.as<char>(), total, c->stream, &cs);
        if (rc) return rc;
        c->link_down_s += cs;
        t_dnlock = wall() - cs;                       // (trace: what was not the copy was the wait behind other contexts' copies)
    }
    tp[6] = wall();
    c->text_call_s += tp[6] - tp[0]; c->text_calls++;
    if (trace)
        fprintf(stderr, "[text] n=%ld in=%.1fMB out=%.1fMB  (waits for the link: up %.2f, down %.2f)  upload %.2f lines %.2f  records %.2f  rows+map %.2f  len+scan %.2f  write %.2f  download %.2f  total %.2f ms\n",
                (long)n2, (double)(bytes1 + bytes2) / 1e6, (double)total / 1e6, (t_uplock - tp[0]) * 1e3, (t_dnlock - t_dnstart) * 1e3, (tp[7] - tp[0]) * 1e3, (tp[1] - tp[7]) * 1e3, (tp[2] - tp[1]) * 1e3, (tp[3] - tp[2]) * 1e3,
                (tp[4] - tp[3]) * 1e3, (tp[5] - tp[4]) * 1e3, (tp[6] - tp[5]) * 1e3, (tp[6] - tp[0]) * 1e3);
    return BMBS_OK;
}

// the part of an "open" call behind the inflated text: newline index, whole records, what is left behind them
static int lane_text_open_finish(Lane* c, bool pe, DevBuf* const* texts, u64* bytes, int64_t max_records, int64_t* n_records, char* tail1, uint64_t tail_cap,
                                 uint64_t* tail1_bytes, char* tail2, uint64_t* tail2_bytes, double* tp, const char* what, double comp_mb)
{
    auto wall = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    const u64 n_cap = (u64)max_records;
    int rc = text_index(c, c->fq_text1, bytes[0], n_cap, 0);
    if (rc) return rc;
    if (pe) { rc = text_index(c, c->fq_text2, bytes[1], n_cap, 1); if (rc) return rc; }
    HIPCHK(c, hipMemcpyAsync(c->h_info + 16, c->totals.as<u64>() + 16, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    tp[2] = wall();
    const u64* lines = reinterpret_cast<const u64*>(c->h_info + 16);
    u64 n = lines[0] / 4;
    if (pe) n = std::min(n, lines[1] / 4);
    n = std::min(n, n_cap);
    // the text behind the n records: handed back for the next window
    u64 cut[2] = {0, 0};
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->h_info + 20, c->tx_nl[0].as<u32>() + (4 * n - 1), 4, hipMemcpyDeviceToHost, c->stream));
        if (pe) HIPCHK(c, hipMemcpyAsync(c->h_info + 21, c->tx_nl[1].as<u32>() + (4 * n - 1), 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        cut[0] = (u64)c->h_info[20] + 1; if (pe) cut[1] = (u64)c->h_info[21] + 1;
    }
    hipStream_t ds = c->kn.copy_streams && c->down_stream ? c->down_stream : c->stream;
    char* tails[2] = {tail1, tail2}; uint64_t* tb[2] = {tail1_bytes, tail2_bytes};
    for (int f = 0; f < (pe ? 2 : 1); f++) *tb[f] = bytes[f] - cut[f];                   // (both sizes are known to a caller that has to come back with room)
    for (int f = 0; f < (pe ? 2 : 1); f++)
        if (*tb[f] > tail_cap) { c->err = "text open: the tail buffer is too small (" + std::to_string(*tb[f]) + " bytes behind the window's records)"; return BMBS_ENOMEM; }
    for (int f = 0; f < (pe ? 2 : 1); f++)
        if (*tb[f]) HIPCHK(c, hipMemcpyAsync(tails[f], texts[f]->as<char>() + cut[f], *tb[f], hipMemcpyDeviceToHost, ds));
    HIPCHK(c, hipStreamSynchronize(ds));
    tp[3] = wall();
    if (what)
        fprintf(stderr, "[text open %s] n=%ld comp=%.1fMB text=%.1fMB  upload+inflate %.2f  lines %.2f  tails %.2f (%.2f MB)  total %.2f ms\n", what, (long)n,
                comp_mb, (double)(bytes[0] + bytes[1]) / 1e6, (tp[1] - tp[0]) * 1e3, (tp[2] - tp[1]) * 1e3, (tp[3] - tp[2]) * 1e3,
                (double)(*tb[0] + (pe ? *tb[1] : 0)) / 1e6, (tp[3] - tp[0]) * 1e3);
    *n_records = (int64_t)n;
    c->open_text.valid = n > 0; c->open_text.pe = pe; c->open_text.bytes1 = cut[0]; c->open_text.bytes2 = cut[1]; c->open_text.n = (int64_t)n;
    return BMBS_OK;
}

// ---- compressed input that stays on the device: open (assemble + index a window from BGZF blocks) and map (everything after) -------
struct ZTextArgs { const bmbs_ztext* z; DevBuf* text; DevBuf* comp; DevBuf* off; DevBuf* err; };

int lane_text_open_bgzf(Lane* c, const bmbs_ztext* z1, const bmbs_ztext* z2, int64_t max_records, int32_t last1, int32_t last2, int64_t* n_records,
                        char* tail1, uint64_t tail_cap, uint64_t* tail1_bytes, char* tail2, uint64_t* tail2_bytes)
{
    if (!c) return BMBS_EINVAL;
    if (n_records) *n_records = 0;
    if (tail1_bytes) *tail1_bytes = 0;
    if (tail2_bytes) *tail2_bytes = 0;
    c->open_text.valid = false;
    if (!c->attached) { c->err = "no index attached"; return BMBS_ESTATE; }
    if (!z1 || max_records <= 0 || !n_records || !tail1 || !tail1_bytes || (z2 && (!tail2 || !tail2_bytes))) { c->err = "text open: NULL argument"; return BMBS_EINVAL; }
    HIPCHK(c, hipSetDevice(c->dev));
    { const int rs = lane_settle(c); if (rs) return rs; }
    const bool pe = z2 != nullptr;
    ZTextArgs A[2] = {{z1, &c->fq_text1, &c->z_comp, &c->z_off, &c->z_err}, {z2, &c->fq_text2, &c->z_comp2, &c->z_off2, &c->z_err2}};
    const int32_t last[2] = {last1, last2};
    u64 bytes[2] = {0, 0};
    static const bool trace = getenv("BMBS_TEXT_TRACE") != nullptr;
    auto wall = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    double tp[6] = {wall(), 0, 0, 0, 0, 0};
    HIPCHK(c, hipMemsetAsync(c->tx_info.p, 0, 64, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // each file on a stream of its own: upload, inflate (one wave per block: a launch of one file's blocks leaves most of the chip's
    // wave slots empty, so the two files' launches run side by side), last line
    hipStream_t fs[2] = {c->stream, c->kn.copy_streams && c->up_stream ? c->up_stream : c->stream};
    for (int f = 0; f < (pe ? 2 : 1); f++) {
        const bmbs_ztext* z = A[f].z;
        const u64 nb = (u64)std::max<int64_t>(0, z->n_blocks);
        if ((z->prefix_bytes && !z->prefix) || (nb && (!z->comp || !z->blk_off || !z->out_off))) { c->err = "text open: NULL buffer"; return BMBS_EINVAL; }
        const u64 text = nb ? z->out_off[nb] : 0;
        if (nb && z->blk_off[nb] > z->comp_bytes) { c->err = "text open: block table outside the compressed bytes"; return BMBS_EINVAL; }
        for (u64 i = 0; i < nb; i++)
            if (z->blk_off[i + 1] < z->blk_off[i] + 26 || z->out_off[i + 1] < z->out_off[i] || z->out_off[i + 1] - z->out_off[i] > 65536) { c->err = "text open: malformed block table"; return BMBS_EINVAL; }
        // (the text has to be contiguous behind the prefix: the inflate kernel takes any byte offset, and the line kernels read the
        // window from its 16-byte aligned start)
        bytes[f] = z->prefix_bytes + text;
        if (bytes[f] + 1 >= (1ull << 32)) { c->err = "a text window has to be smaller than 4 GiB (32-bit offsets)"; return BMBS_EINVAL; }
        ENS(c, *A[f].text, bytes[f] + 64 + 16);
        if (nb) { ENS(c, *A[f].comp, z->comp_bytes + 1024); ENS(c, *A[f].off, 2 * (nb + 1) * 8 + 64); ENS(c, *A[f].err, nb * 4 + 64); }
    }
    for (int f = 0; f < (pe ? 2 : 1); f++) {
        const bmbs_ztext* z = A[f].z;
        const u64 nb = (u64)std::max<int64_t>(0, z->n_blocks);
        if (z->prefix_bytes) HIPCHK(c, hipMemcpyAsync(A[f].text->p, z->prefix, z->prefix_bytes, hipMemcpyHostToDevice, fs[f]));
        if (nb) {
            HIPCHK(c, hipMemcpyAsync(A[f].comp->p, z->comp, z->comp_bytes, hipMemcpyHostToDevice, fs[f]));
            HIPCHK(c, hipMemcpyAsync(A[f].off->p, z->blk_off, (nb + 1) * 8, hipMemcpyHostToDevice, fs[f]));
            HIPCHK(c, hipMemcpyAsync(A[f].off->as<u64>() + (nb + 1), z->out_off, (nb + 1) * 8, hipMemcpyHostToDevice, fs[f]));
            hipLaunchKernelGGL(k_bgzf_inflate, dim3((unsigned)nb), dim3(64), 0, fs[f], A[f].comp->as<u8>(), A[f].off->as<u64>(), A[f].off->as<u64>() + (nb + 1), (long)nb,
                               A[f].text->as<char>() + z->prefix_bytes, A[f].err->as<u32>());
        }
        if (last[f] && bytes[f]) {
            // an unterminated last line counts as a line (the reader's rule): the newline is added here, on the device
            hipLaunchKernelGGL(k_close_last_line, dim3(1), dim3(1), 0, fs[f], A[f].text->as<char>(), bytes[f], c->totals.as<u64>() + 21 + f);
        } else HIPCHK(c, hipMemsetAsync(c->totals.as<u64>() + 21 + f, 0, 8, fs[f]));
    }
    if (pe && fs[1] != c->stream) HIPCHK(c, hipStreamSynchronize(fs[1]));
    // whether a newline was added has to be known before the lines are indexed
    HIPCHK(c, hipMemcpyAsync(c->h_info + 28, c->totals.as<u64>() + 21, 16, hipMemcpyDeviceToHost, c->stream));
    std::vector<u32> err[2];
    for (int f = 0; f < (pe ? 2 : 1); f++) {
        const u64 nb = (u64)std::max<int64_t>(0, A[f].z->n_blocks);
        err[f].resize(nb);
        if (nb) HIPCHK(c, hipMemcpyAsync(err[f].data(), A[f].err->p, nb * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int f = 0; f < (pe ? 2 : 1); f++)
        for (size_t i = 0; i < err[f].size(); i++)
            if (err[f][i]) { c->err = "corrupt BGZF block in the .gz input (file " + std::to_string(f + 1) + ", block " + std::to_string(i) + " of this window, code " + std::to_string(err[f][i]) + ")"; return BMBS_EINVAL; }
    tp[1] = wall();
    const u64* added = reinterpret_cast<const u64*>(c->h_info + 28);
    bytes[0] += added[0]; if (pe) bytes[1] += added[1];
    DevBuf* texts[2] = {&c->fq_text1, &c->fq_text2};
    return lane_text_open_finish(c, pe, texts, bytes, max_records, n_records, tail1, tail_cap, tail1_bytes, tail2, tail2_bytes, tp, trace ? "bgzf" : nullptr,
                                 (double)(z1->comp_bytes + (z2 ? z2->comp_bytes : 0)) / 1e6);
}

int lane_text_map_open(Lane* c, int32_t flags_in, char* sam, u64 sam_cap, u64* sam_bytes, int64_t* n_lines_out)
{
    if (!c) return BMBS_EINVAL;
    if (sam_bytes) *sam_bytes = 0;
    if (n_lines_out) *n_lines_out = 0;
    if (!c->open_text.valid) { c->err = "text map: no open batch (bmbs_text_open_bgzf first)"; return BMBS_ESTATE; }
    if (c->n_refs != c->ix.n_chrom) { c->err = "bmbs_sam_refs has not been given the index's reference names"; return BMBS_ESTATE; }
    if (!sam) { c->err = "text call: NULL buffer"; return BMBS_EINVAL; }
    HIPCHK(c, hipSetDevice(c->dev));
    auto wall = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    const double t0 = wall();
    // (an output buffer that turns out too small leaves the batch open: the call can be repeated)
    const int rc = lane_text_finish(c, c->open_text.pe, c->open_text.bytes1, c->open_text.bytes2, c->open_text.n, flags_in, sam, sam_cap, sam_bytes, n_lines_out, t0, t0);
    if (rc != BMBS_ENOMEM) c->open_text.valid = false;
    return rc;
}

// RNAME table of the SAM text: the names behind the index's sequences, in index order
int lane_sam_refs(Lane* c, const char* const* names, int n_names)
{
    if (!names || n_names < 1) { c->err = "sam refs: no names"; return BMBS_EINVAL; }
    HIPCHK(c, hipSetDevice(c->dev));
    std::vector<u32> off((size_t)n_names + 1, 0);
    std::string chars;
    int mx = 0;
    for (int i = 0; i < n_names; i++) {
        const size_t l = names[i] ? strlen(names[i]) : 0;
        if (l > 4096) { c->err = "sam refs: a reference name is longer than 4096 characters"; return BMBS_EINVAL; }
        chars.append(names[i] ? names[i] : "", l);
        off[(size_t)i + 1] = (u32)chars.size();
        mx = std::max(mx, (int)l);
    }
    ENS(c, c->chrom_chars, chars.size() + 64); ENS(c, c->chrom_off, off.size() * 4);
    HIPCHK(c, hipMemcpy(c->chrom_chars.p, chars.data(), chars.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->chrom_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    c->n_refs = n_names; c->max_ref_len = mx;
    return BMBS_OK;
}

// bgzip'ed input inflated on the device (bmbs_inflate.hip): needs no index
static int lane_inflate_bgzf(Lane* c, const void* comp, uint64_t comp_bytes, const uint64_t* blk_off, const uint64_t* out_off, int64_t n_blocks,
                             char* text, uint64_t text_bytes, uint32_t* nl_per_64k, uint64_t window_shift)
{
    if (n_blocks <= 0) return BMBS_OK;
    if (!comp || !blk_off || !out_off || !text) { c->err = "inflate: NULL buffer"; return BMBS_EINVAL; }
    const u64 n = (u64)n_blocks;
    if (blk_off[n] > comp_bytes || out_off[n] > text_bytes) { c->err = "inflate: block table outside the buffers"; return BMBS_EINVAL; }
    for (u64 i = 0; i < n; i++)
        if (blk_off[i + 1] < blk_off[i] + 26 || out_off[i + 1] < out_off[i] || out_off[i + 1] - out_off[i] > 65536) { c->err = "inflate: malformed block table"; return BMBS_EINVAL; }
    HIPCHK(c, hipSetDevice(c->dev));
    static const bool trace = getenv("BMBS_TEXT_TRACE") != nullptr;       // diagnostic: host-side phase times of every call
    auto wall = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    const double t0 = wall();
    ENS(c, c->z_comp, comp_bytes + 1024); ENS(c, c->z_off, 2 * (n + 1) * 8 + 64); ENS(c, c->z_text, out_off[n] + 64); ENS(c, c->z_err, n * 4 + 64);
    hipStream_t us = c->kn.copy_streams && c->up_stream ? c->up_stream : c->stream;
    hipStream_t ds = c->kn.copy_streams && c->down_stream ? c->down_stream : c->stream;
    HIPCHK(c, hipMemcpyAsync(c->z_comp.p, comp, comp_bytes, hipMemcpyHostToDevice, us));
    HIPCHK(c, hipMemcpyAsync(c->z_off.p, blk_off, (n + 1) * 8, hipMemcpyHostToDevice, us));
    HIPCHK(c, hipMemcpyAsync(c->z_off.as<u64>() + (n + 1), out_off, (n + 1) * 8, hipMemcpyHostToDevice, us));
    HIPCHK(c, hipStreamSynchronize(us));
    const double t1 = wall();
    hipLaunchKernelGGL(k_bgzf_inflate, dim3((unsigned)n), dim3(64), 0, c->stream, c->z_comp.as<u8>(), c->z_off.as<u64>(), c->z_off.as<u64>() + (n + 1), (long)n,
                       c->z_text.as<char>(), c->z_err.as<u32>());
    const u64 n_cnt = nl_per_64k ? (window_shift + out_off[n] + 65535) >> 16 : 0;
    if (n_cnt) {
        ENS(c, c->z_nl, n_cnt * 4 + 64);
        hipLaunchKernelGGL(k_nl_count64k, dim3((unsigned)n_cnt), dim3(256), 0, c->stream, c->z_text.as<char>(), out_off[n], window_shift, c->z_nl.as<u32>());
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double t2 = wall();
    std::vector<u32> err(n);
    HIPCHK(c, hipMemcpyAsync(err.data(), c->z_err.p, n * 4, hipMemcpyDeviceToHost, ds));
    if (n_cnt) HIPCHK(c, hipMemcpyAsync(nl_per_64k, c->z_nl.p, n_cnt * 4, hipMemcpyDeviceToHost, ds));
    int rc = d2h_chunked(c, text, c->z_text.as<char>(), out_off[n], ds);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(ds));
    if (trace) fprintf(stderr, "[inflate] %lu blocks, %.1f MB -> %.1f MB: alloc+upload %.2f  kernel %.2f  download %.2f ms\n", (unsigned long)n, (double)comp_bytes / 1e6, (double)out_off[n] / 1e6,
                       (t1 - t0) * 1e3, (t2 - t1) * 1e3, (wall() - t2) * 1e3);
    for (u64 i = 0; i < n; i++)
        if (err[i]) { c->err = "corrupt BGZF block in the .gz input (block " + std::to_string(i) + " of this window, code " + std::to_string(err[i]) + ")"; return BMBS_EINVAL; }
    return BMBS_OK;
}

extern "C" int bmbs_inflate_bgzf(bmbs_ctx* X, const void* comp, uint64_t comp_bytes, const uint64_t* blk_off, const uint64_t* out_off, int64_t n_blocks,
                                 char* text, uint64_t text_bytes, uint32_t* nl_per_64k, uint64_t window_shift)
{
    Lane* c = lane0(X);
    if (!c) return BMBS_EINVAL;
    return fin(X, c, lane_inflate_bgzf(c, comp, comp_bytes, blk_off, out_off, n_blocks, text, text_bytes, nl_per_64k, window_shift));
}

#ifdef INF_PROFILE
// profiling build only (tools/inflate_prof.sh): the phase cycle sums of k_bgzf_inflate since the last call
extern "C" int bmbs_debug_inflate_prof(uint64_t* out16)
{
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_inf_prof), sizeof z) != hipSuccess) return BMBS_ENODEV;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_inf_prof), z, sizeof z) == hipSuccess ? BMBS_OK : BMBS_ENODEV;
}
#endif

#ifdef BGZF_PROFILE
// profiling build only (tools/bgzf_prof.sh): the phase cycle sums of k_bgzf_block since the last call
extern "C" int bmbs_debug_bgzf_prof(uint64_t* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bgzf_prof), sizeof z) != hipSuccess) return BMBS_ENODEV;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bgzf_prof), z, sizeof z) == hipSuccess ? BMBS_OK : BMBS_ENODEV;
}
#endif

// diagnostic: the Huffman code lengths the device's BGZF deflater gives a table of symbol frequencies (n <= 320, maxbits <= 15)
extern "C" int bmbs_debug_huff_lengths(bmbs_ctx* X, const uint32_t* freq, int32_t n, int32_t maxbits, uint8_t* len_out)
{
    Lane* c = lane0(X);
    if (!c) return BMBS_EINVAL;
    if (!freq || !len_out || n < 2 || n > 320 || maxbits < 2 || maxbits > 15) { c->err = "huff lengths: bad argument"; return fin(X, c, BMBS_EINVAL); }
    if (hipSetDevice(c->dev) != hipSuccess) return BMBS_ENODEV;
    u32* df = nullptr; u8* dl = nullptr;
    if (hipMalloc((void**)&df, 321 * 4) != hipSuccess || hipMalloc((void**)&dl, 320) != hipSuccess) { if (df) (void)hipFree(df); return BMBS_ENOMEM; }
    int rc = BMBS_OK;
    u32 differ = 0;
    if (hipMemcpy(df, freq, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemset(df + 320, 0, 4) != hipSuccess) rc = BMBS_ENODEV;
    if (!rc) { hipLaunchKernelGGL(k_debug_huff, dim3(1), dim3(256), 0, c->stream, df, n, maxbits, dl, df + 320); if (hipStreamSynchronize(c->stream) != hipSuccess) rc = BMBS_ENODEV; }
    if (!rc && (hipMemcpy(len_out, dl, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&differ, df + 320, 4, hipMemcpyDeviceToHost) != hipSuccess)) rc = BMBS_ENODEV;
    (void)hipFree(df); (void)hipFree(dl);
    if (!rc && differ) { c->err = "huff lengths: the workgroup forms (huff_lengths_block / huff_codes_block) differ from the serial ones"; return fin(X, c, BMBS_ESTATE); }
    return rc;
}

// diagnostic: seconds the context's text calls held the link (uploads, downloads: copy + wait for its end, the wait for the link's lock
// excluded), seconds inside those calls, and their number
extern "C" int bmbs_text_times(bmbs_ctx* X, double out[4])
{
    if (!X || !out) return BMBS_EINVAL;
    out[0] = out[1] = out[2] = out[3] = 0;
    for (Lane* c : X->lanes) { out[0] += c->link_up_s; out[1] += c->link_down_s; out[2] += c->text_call_s; out[3] += (double)c->text_calls; }
    return BMBS_OK;
}


extern "C" int bmbs_sam_refs(bmbs_ctx* X, const char* const* names, int32_t n_names) { ON_LANE0(lane_sam_refs(c, names, n_names)); }
extern "C" int bmbs_map_se_text(bmbs_ctx* X, const char* text, uint64_t text_bytes, int64_t n_records, int32_t flags, char* sam, uint64_t sam_cap,
                                uint64_t* sam_bytes, int64_t* n_lines)
{ ON_LANE0(lane_map_text(c, false, text, text_bytes, nullptr, 0, n_records, flags, sam, sam_cap, sam_bytes, n_lines)); }
extern "C" int bmbs_text_open_bgzf(bmbs_ctx* X, const bmbs_ztext* mate1, const bmbs_ztext* mate2, int64_t max_records, int32_t last1, int32_t last2,
                                   int64_t* n_records, char* tail1, uint64_t tail_cap, uint64_t* tail1_bytes, char* tail2, uint64_t* tail2_bytes)
{ ON_LANE0(lane_text_open_bgzf(c, mate1, mate2, max_records, last1, last2, n_records, tail1, tail_cap, tail1_bytes, tail2, tail2_bytes)); }
extern "C" int bmbs_text_map_open(bmbs_ctx* X, int32_t flags, char* sam, uint64_t sam_cap, uint64_t* sam_bytes, int64_t* n_lines)
{ ON_LANE0(lane_text_map_open(c, flags, sam, sam_cap, sam_bytes, n_lines)); }
extern "C" int bmbs_map_pe_text(bmbs_ctx* X, const char* text1, uint64_t bytes1, const char* text2, uint64_t bytes2, int64_t n_pairs, int32_t flags,
                                char* sam, uint64_t sam_cap, uint64_t* sam_bytes, int64_t* n_lines)
{ ON_LANE0(lane_map_text(c, true, text1, bytes1, text2, bytes2, n_pairs, flags, sam, sam_cap, sam_bytes, n_lines)); }
extern "C" int bmbs_map_se_fastq(bmbs_ctx* X, const bmbs_fastq_view* reads, int64_t n_reads, int32_t L_max, int32_t uniform, int32_t pbat,
                                 bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used)
{ ON_LANE0(lane_map_se_fastq(c, reads, n_reads, L_max, uniform, pbat, results, cigar_pool, cigar_cap, n_cigar_used)); }
extern "C" int bmbs_map_pe_fastq(bmbs_ctx* X, const bmbs_fastq_view* mate1, const bmbs_fastq_view* mate2, int64_t n_pairs, int32_t L_max,
                                 int32_t uniform, bmbs_result* results, uint32_t* cigar_pool, int64_t cigar_cap, int64_t* n_cigar_used)
{ ON_LANE0(lane_map_pe_fastq(c, mate1, mate2, n_pairs, L_max, uniform, results, cigar_pool, cigar_cap, n_cigar_used)); }

