// bitmapperbs_amd/csrc/bmbs_inflate.hip -- bgzip'ed FASTQ inflated on the device (round 4).
//
// The reference reads .fastq.gz through zlib's gzread on its one reader thread (Process_Reads.cpp:1455-1514).  A BGZF file is a
// series of independent gzip members of at most 64 KiB of text with their compressed size in the header: nothing in one block
// depends on another, so a window of them is inflated here by one WAVE per block -- the host of an MI355X box (16 cores' worth of
// CPU for the whole driver) inflates ~7 GB/s of text at best, a third of what the mapping path takes.
//   every lane decodes the token that would start at its own bit offset (tables in LDS: 10-bit root for literals / lengths with two
//   literals per entry where both codes fit, 9-bit root for distances; longer codes -- rare by construction -- by a canonical search
//   in the lane that holds them), a walk from offset 0 picks the tokens that are really there (~12 per 64-bit window of FASTQ text), literals are stored
//   by their lanes, matches copied by the whole wave; the block's CRC-32 and ISIZE are checked against its trailer (segments
//   combined with x^(8n) mod P as in bmbs_bam.hip).
// Everything zlib's inflate refuses is refused (err[block] != 0): over-subscribed / incomplete codes, missing end-of-block code,
// reserved block type, distance too far back, stored-length check, output longer than the trailer says, CRC mismatch.
#ifndef BMBS_INFLATE_HIP
#define BMBS_INFLATE_HIP
#include "bmbs_bytes.h"

#define INF_LIT_ROOT 10
#define INF_DIST_ROOT 9

// -DINF_PROFILE (tools/inflate_prof.sh builds a second library with it): cycles per phase of k_bgzf_inflate, summed over the blocks
#ifdef INF_PROFILE
__device__ unsigned long long g_inf_prof[16];
#define INF_T(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = clock64(); prof[k] += t_ - t_last; t_last = t_; } while (0)
#define INF_N(k, v) prof[k] += (v)
#else
#define INF_T(k)
#define INF_N(k, v)
#endif

// entry = value << 16 | extra bits (or literal count) << 8 | kind << 5 | code bits
#define IK_BAD 0
#define IK_LIT 1
#define IK_BASE 2
#define IK_EOB 3
#define IK_LONG 4            // a code longer than the root: canonical search

__constant__ u16 c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ u8 c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ u16 c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ u8 c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct InfCode {             // canonical description of one Huffman code, in LDS
    u16 count[16];           // codes per length
    u16 first[16];           // first (MSB-first) code of each length
    u16 offs[16];            // index of that code's symbol in `sorted`
    u16* sorted;             // symbols by (length, symbol)
};

DEVI u32 inf_mk(u32 value, u32 extra, u32 kind, u32 bits) { return value << 16 | extra << 8 | kind << 5 | bits; }

// lens[0, n) -> count / first / offs / sorted (all lanes; returns false on a set zlib refuses).  n <= 288.
DEVI bool inf_canonical(const u8* lens, int n, InfCode& c, int lane)
{
    if (lane < 16) c.count[lane] = 0;
    __syncthreads();
    for (int s = lane; s < n; s += 64) if (lens[s]) atomicAdd(reinterpret_cast<u32*>(&c.count[lens[s] & ~1]), lens[s] & 1 ? 0x10000u : 1u);
    __syncthreads();
    // Kraft sum, first codes, offsets: 15 steps, every lane the same (LDS broadcast reads)
    int left = 1, maxl = 0; u32 code = 0, off = 0;
    bool over = false;
    u32 firsts[16], offsv[16];
    firsts[0] = 0; offsv[0] = 0;
    for (int l = 1; l <= 15; l++) {
        const int cn = c.count[l];
        left = left * 2 - cn; if (left < 0) over = true;
        if (cn) maxl = l;
        code = (code + (l > 1 ? (u32)c.count[l - 1] : 0u)) << 1;
        firsts[l] = code; offsv[l] = off; off += (u32)cn;
    }
    if (over) return false;
    if (left > 0 && maxl > 1) return false;                      // incomplete: only "one code of one bit" / "no code" pass
    if (lane < 16) { c.first[lane] = (u16)firsts[lane]; c.offs[lane] = (u16)offsv[lane]; }
    // sorted: position of symbol s = offs[len] + symbols t < s of the same length
    for (int s = lane; s < n; s += 64) {
        const int l = lens[s];
        if (!l) continue;
        int r = 0;
        for (int t = 0; t < s; t++) r += lens[t] == l;
        c.sorted[offsv[l] + (u32)r] = (u16)s;
    }
    __syncthreads();
    return true;
}
// the symbol whose code (of up to `maxbits` bits) starts the LSB-first bit string v: -> (symbol, length), length 0 = none
DEVI int inf_search(const InfCode& c, u32 v, int from, int maxbits, int& len_out)
{
    u32 code = 0;
    for (int l = 1; l <= maxbits; l++) {
        code = (code << 1) | ((v >> (l - 1)) & 1u);
        if (l < from) continue;
        const u32 d = code - (u32)c.first[l];
        if (code >= (u32)c.first[l] && d < (u32)c.count[l]) { len_out = l; return (int)c.sorted[(u32)c.offs[l] + d]; }
    }
    len_out = 0;
    return -1;
}
// root table of a code: every lane fills its share of the indexes by a canonical search over the index's own bits
DEVI void inf_root(const InfCode& c, bool dist, u32* tab, int root, int lane)
{
    const int n = 1 << root;
    for (int i = lane; i < n; i += 64) {
        int l; const int s = inf_search(c, (u32)i, 1, root, l);
        u32 e;
        if (s < 0) e = inf_mk(0, 0, IK_LONG, 0);                                  // (also covers "no code at all": the search fails again)
        else if (dist) e = s < 30 ? inf_mk(c_dist_base[s], c_dist_extra[s], IK_BASE, (u32)l) : inf_mk(0, 0, IK_BAD, (u32)l);
        else if (s < 256) e = inf_mk((u32)s, 1, IK_LIT, (u32)l);
        else if (s == 256) e = inf_mk(0, 0, IK_EOB, (u32)l);
        else e = s < 286 ? inf_mk(c_len_base[s - 257], c_len_extra[s - 257], IK_BASE, (u32)l) : inf_mk(0, 0, IK_BAD, (u32)l);
        tab[i] = e;
    }
    __syncthreads();
    if (!dist) {
        // two literals per entry where both codes fit into the root index (read from the finished single-literal table, written to a
        // register first: an entry's partner may be any other entry)
        u32 mine[(1 << INF_LIT_ROOT) / 64];
        for (int q = 0; q < n / 64; q++) {
            const int i = lane + 64 * q;
            u32 e1 = tab[i];
            if (((e1 >> 5) & 7u) == IK_LIT && (int)(e1 & 31u) < root) {
                const u32 e2 = tab[(u32)i >> (e1 & 31u)];
                if (((e2 >> 5) & 7u) == IK_LIT && ((e2 >> 8) & 31u) == 1 && (e1 & 31u) + (e2 & 31u) <= (u32)root)
                    e1 = inf_mk((e1 >> 16) | ((e2 >> 16) << 8), 2, IK_LIT, (e1 & 31u) + (e2 & 31u));
            }
            mine[q] = e1;
        }
        __syncthreads();
        for (int q = 0; q < n / 64; q++) tab[lane + 64 * q] = mine[q];
        __syncthreads();
    }
}

struct InfBits {             // lane 0's bit reader over global memory: aligned 4-byte loads, 33 bits or more held after refill()
    const u8* p; const u8* end; u64 buf; int cnt;
    DEVI void init(const u8* b, const u8* e)
    {
        p = b; end = e; buf = 0; cnt = 0;
        while (((size_t)p & 3) && cnt < 24) { buf |= (u64)*p++ << cnt; cnt += 8; }      // (bytes behind `end` belong to the buffer too: comp has slack)
    }
    DEVI void refill()
    {
        if (cnt <= 32) { buf |= (u64)*reinterpret_cast<const u32*>(p) << cnt; p += 4; cnt += 32; }
    }
    DEVI u32 peek(int n) const { return (u32)(buf & ((1ull << n) - 1)); }
    DEVI void drop(int n) { buf >>= n; cnt -= n; }
    DEVI u32 take(int n) { const u32 v = peek(n); drop(n); return v; }
    DEVI const u8* at() const { return p - (cnt >> 3); }                       // the first byte that has not been consumed (cnt % 8 == 0)
    DEVI bool over() const { return p - ((cnt + 7) >> 3) > end; }
};

// the token that starts an LSB-first bit string v (64 bits of it): table entry, match length / distance, and
// packed = bits of the token | flag << 6 | bytes of text << 9
// (flag: 1 end of block, 2 a code longer than the root, 3 invalid, 4 a match that would not fit into a window -- copied by the whole wave).
// No branches: both look-ups are made whatever the first one says.
struct InfTok { u32 e, packed, mlen, mdist; };
DEVI InfTok inf_token(u64 v, const u32* s_lit, const u32* s_dist)
{
    InfTok k;
    k.e = s_lit[(u32)v & ((1u << INF_LIT_ROOT) - 1)];
    const u32 kind = (k.e >> 5) & 7u, tb0 = k.e & 31u, ex = (k.e >> 8) & 31u;
    const bool base = kind == IK_BASE;
    const u32 tb1 = tb0 + (base ? ex : 0u);
    const u32 d = s_dist[(u32)(v >> tb1) & ((1u << INF_DIST_ROOT) - 1)];
    const u32 dk = (d >> 5) & 7u, dx = (d >> 8) & 31u, tb2 = tb1 + (d & 31u);
    k.mlen = (k.e >> 16) + ((u32)(v >> tb0) & ((1u << ex) - 1u));
    k.mdist = (d >> 16) + ((u32)(v >> tb2) & ((1u << dx) - 1u));
    const bool okm = base && dk == IK_BASE;
    const u32 tb = okm ? tb2 + dx : tb0;
    u32 flag = kind == IK_LIT ? 0u : kind == IK_EOB ? 1u : kind == IK_LONG ? 2u : base ? (dk == IK_BASE ? 0u : dk == IK_LONG ? 2u : 3u) : 3u;
    if (okm && k.mlen > 250) flag = 4;
    const u32 ol = kind == IK_LIT ? ex : okm ? k.mlen : 0u;
    k.packed = tb | flag << 6 | ol << 9;
    return k;
}

// ---- one wave inflates one deflate stream (or a stretch of one) -------------------------------------------------------------------
// A whole gzip member's deflate data from start_bit (a BGZF block), bytes out.  (Round 4 also ran this loop over spans of ONE long
// deflate stream with 16-bit symbols -- the device form of csrc/pgz.h; exact, and slower than the host's: removed in round 5.
// Round 5 also tried keeping the last 2 KiB of the text in an LDS ring, so that what a match copies from before the window being
// assembled needs neither the round trip to memory nor the fence that three windows in four took: the same 4.7 ms per window
// (tools/inflate_kbench.hip) -- the round trip was already hidden behind the pointer jumping; a wave's own chain of dependent
// instructions bounds the kernel, and two windows side by side take 3.5 ms each.)
DEVI void inf_wave(const u8* z, const u8* zend, u32 start_bit, u8* out, u32 isize, int lane, u32& status_r, u32& n_out_r)
{
    typedef u8 S;
    __shared__ u32 s_lit[1 << INF_LIT_ROOT];
    __shared__ u32 s_dist[1 << INF_DIST_ROOT];
    __shared__ u8 s_lens[320];
    __shared__ u16 s_sorted_l[288]; __shared__ u16 s_sorted_d[32]; __shared__ u16 s_sorted_c[20];
    __shared__ InfCode s_cl, s_cd, s_cc;
    __shared__ u32 s_ref32[64];                           // the window being put together: where each of its 256 bytes comes from ...
    __shared__ u32 s_val32[64 + 4];                       // ... and the symbols that are known (literals, text from before the window)
    S* s_val = reinterpret_cast<S*>(s_val32);
    __shared__ u32 s_tid32[64];                           // per window byte: the id of the token that starts there (0: none)
    __shared__ u32 s_rec[128];                            // per token id - 1: its record (see the window assembly)
    u8* s_tid = reinterpret_cast<u8*>(s_tid32);
    u8* s_ref = reinterpret_cast<u8*>(s_ref32);
    u32 status = 0;                                       // 0 ok so far
#ifdef INF_PROFILE
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = clock64();
    const unsigned long long t_begin = t_last;
#endif
    s_cl.sorted = s_sorted_l; s_cd.sorted = s_sorted_d; s_cc.sorted = s_sorted_c;
    __syncthreads();
    InfBits in; in.init(z + (start_bit >> 3), zend); in.refill(); in.drop((int)(start_bit & 7u));
    // the compressed bytes, 256 at a time: lane l holds dword l of chunk kc (`cur`) and of the chunk behind it (`nxt`, on its way
    // while `cur` is decoded); the five dwords under a window are picked out of them with v_readlane
    const u32* zw = reinterpret_cast<const u32*>((size_t)z & ~(size_t)3);
    const u32 zsh = ((u32)(size_t)z & 3u) * 8;
    u32 kc = 0xfffffff0u, cur = 0, nxt = 0;
    u32 n_out = 0;                                        // bytes of text written so far (wave-uniform)
    u32 fenced = 0;                                       // every byte of the text below this offset is visible to every lane
    S last_byte = (S)0;                                   // the symbol at n_out - 1 (wave-uniform)
    bool last = false;
    while (!status && !last) {
        // ---- block header (lane 0 reads, the wave follows)
        u32 btype = 0;
        if (lane == 0) { in.refill(); last = in.take(1) != 0; btype = in.take(2); }
        last = __shfl((int)last, 0) != 0; btype = (u32)__shfl((int)btype, 0);
        if (btype == 3) { status = 2; break; }
        if (btype == 0) {
            // stored: LEN, NLEN, bytes
            u32 len = 0, bad = 0; u64 at = 0;
            if (lane == 0) {
                in.drop(in.cnt & 7); in.refill();
                len = in.take(16); in.refill(); const u32 nlen = in.take(16);
                bad = (len ^ 0xffffu) != nlen;
                at = (u64)(in.at() - z);                              // byte position of the data
            }
            len = (u32)__shfl((int)len, 0); bad = (u32)__shfl((int)bad, 0);
            at = ((u64)(u32)__shfl((int)(at >> 32), 0) << 32) | (u32)__shfl((int)at, 0);
            if (bad || at + len > (u64)(zend - z) || n_out + len > isize) { status = 3; break; }
            for (u32 i = lane; i < len; i += 64) out[n_out + i] = (S)z[at + i];
            n_out += len;
            if (len) last_byte = (S)z[at + len - 1];
            if (lane == 0) in.init(z + at + len, zend);
            continue;
        }
        // ---- the two codes
        bool ok = true;
        if (btype == 1) {
            for (int s = lane; s < 288; s += 64) s_lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) s_lens[288 + lane] = 5;
            __syncthreads();
            ok = inf_canonical(s_lens, 288, s_cl, lane) && inf_canonical(s_lens + 288, 32, s_cd, lane);
        } else {
            // dynamic: HLIT, HDIST, HCLEN, the code-length code, then the lengths (lane 0 walks them)
            int hlit = 0, hdist = 0, hclen = 0;
            if (lane < 19) s_lens[lane] = 0;
            __syncthreads();
            if (lane == 0) {
                in.refill();
                hlit = (int)in.take(5) + 257; hdist = (int)in.take(5) + 1; hclen = (int)in.take(4) + 4;
                const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                for (int i = 0; i < hclen; i++) { in.refill(); s_lens[order[i]] = (u8)in.take(3); }
            }
            hlit = __shfl(hlit, 0); hdist = __shfl(hdist, 0);
            __syncthreads();
            if (hlit > 286 || hdist > 30) { status = 4; break; }
            ok = inf_canonical(s_lens, 19, s_cc, lane);
            {
                // zlib: the code-length code has to be complete
                int left = 1; for (int l = 1; l <= 7; l++) left = left * 2 - (int)s_cc.count[l];
                if (left != 0) ok = false;
            }
            if (!ok) { status = 4; break; }
            u32 bad = 0;
            if (lane == 0) {
                // (the 19 lengths were consumed: s_lens is rewritten with the literal/length + distance lengths behind a copy of nothing
                // -- the code-length code now lives in s_cc alone)
                int have = 0; const int total = hlit + hdist;
                u8 prev = 0;
                while (have < total && !bad) {
                    in.refill();
                    int l; const int s = inf_search(s_cc, in.peek(7), 1, 7, l);
                    if (s < 0) { bad = 1; break; }
                    in.drop(l);
                    if (s < 16) { prev = (u8)s; s_lens[have < hlit ? have : 288 + (have - hlit)] = prev; have++; continue; }
                    int rep; u8 v = 0;
                    if (s == 16) { if (!have) { bad = 1; break; } v = prev; rep = 3 + (int)in.take(2); }
                    else if (s == 17) rep = 3 + (int)in.take(3);
                    else rep = 11 + (int)in.take(7);
                    if (have + rep > total) { bad = 1; break; }
                    if (s != 16) prev = 0;
                    while (rep--) { s_lens[have < hlit ? have : 288 + (have - hlit)] = v; have++; }
                }
                for (int i = hlit; i < 288; i++) s_lens[i] = 0;
                for (int i = hdist; i < 32; i++) s_lens[288 + i] = 0;
                if (!bad && !s_lens[256]) bad = 1;                       // zlib: missing end-of-block code
                if (in.over()) bad = 1;
            }
            bad = (u32)__shfl((int)bad, 0);
            __syncthreads();
            if (bad) { status = 4; break; }
            ok = inf_canonical(s_lens, 288, s_cl, lane) && inf_canonical(s_lens + 288, 32, s_cd, lane);
        }
        if (!ok) { status = 4; break; }
        inf_root(s_cl, false, s_lit, INF_LIT_ROOT, lane);
        inf_root(s_cd, true, s_dist, INF_DIST_ROOT, lane);
        // ---- symbols.  Every lane decodes the tokens that WOULD start at its own two bit offsets (bp + lane, bp + 64 + lane): one or
        // two literals, a match with its length / distance codes and extra bits, or the end-of-block code -- two dependent LDS
        // look-ups for all 128 offsets at once.  The tokens that really are in the stream are the ones reached from offset 0 by
        // following the token lengths: a walk in scalar registers (one v_readlane per token) that also hands every token its place
        // in the text.  A 128-bit window holds ~12 tokens of FASTQ text; one lane decoding alone took ~0.5 us per token.
        INF_T(0);
        u32 bp = 0;
        if (lane == 0) bp = (u32)((in.p - z) * 8) - (u32)in.cnt;
        bp = (u32)__shfl((int)bp, 0);
        const u32 end_bits = (u32)(zend - z) * 8;
        bool block_done = false;
        while (!block_done && !status) {
            // (bp, the chain position and the output offsets are the same in every lane: kept in scalar registers, so that a token's
            // fields are read with v_readlane instead of a shuffle through LDS)
            bp = (u32)__builtin_amdgcn_readfirstlane((int)bp);
            n_out = (u32)__builtin_amdgcn_readfirstlane((int)n_out);
            if (bp > end_bits) { status = 7; break; }
            // 224 bits from the aligned dword that holds bit bp (the same seven words for every lane)
            const u32 gb = zsh + bp, wq = gb >> 5, kq = (u32)__builtin_amdgcn_readfirstlane((int)(wq >> 6));
            if (kq != kc) {
                cur = kq == kc + 1 ? nxt : zw[(size_t)kq * 64 + lane];
                nxt = zw[((size_t)kq + 1) * 64 + lane];
                kc = kq;
            }
            const u32 wl = wq & 63u;
            u32 dw[7];
            if (wl + 7 <= 64) {
                // (the usual case touches `cur` alone: no wait for the chunk that is still on its way)
#pragma unroll
                for (u32 i = 0; i < 7; i++) dw[i] = (u32)__builtin_amdgcn_readlane((int)cur, (int)(wl + i));
            } else {
#pragma unroll
                for (u32 i = 0; i < 7; i++) {
                    const u32 a = (u32)__builtin_amdgcn_readlane((int)cur, (int)((wl + i) & 63u)), c2 = (u32)__builtin_amdgcn_readlane((int)nxt, (int)((wl + i) & 63u));
                    dw[i] = wl + i < 64 ? a : c2;
                }
            }
            const u32 o = (gb & 31u) + (u32)lane;                                          // 0 .. 94
            INF_T(1); INF_N(9, 1);
            // the tokens at this lane's two offsets: A at bp + lane, B 64 bits further
            const u32 j = o >> 5, sh = o & 31u;
            const u32 x0 = j == 0 ? dw[0] : j == 1 ? dw[1] : dw[2], x1 = j == 0 ? dw[1] : j == 1 ? dw[2] : dw[3], x2 = j == 0 ? dw[2] : j == 1 ? dw[3] : dw[4];
            const u32 x3 = j == 0 ? dw[3] : j == 1 ? dw[4] : dw[5], x4 = j == 0 ? dw[4] : j == 1 ? dw[5] : dw[6];
            u64 vA = (((u64)x1 << 32) | x0) >> sh, vB = (((u64)x3 << 32) | x2) >> sh;
            if (sh) { vA |= (u64)x2 << (64 - sh); vB |= (u64)x4 << (64 - sh); }
            const InfTok A = inf_token(vA, s_lit, s_dist), B = inf_token(vB, s_lit, s_dist);
            INF_T(2);
            // the chain from offset 0: the tokens it visits (mA, mB: one bit per lane) and their places in the text.  It ends at the
            // first token that is not a plain one, or when the window (256 bytes of text from `lo` on, see below) is full.
            const u32 lo = (u32)((size_t)(out + n_out) / sizeof(S)) & 3u;                  // the window starts at the 4-symbol boundary at or below n_out
            u32 t = 0, run = n_out, stop = 0;
            unsigned long long mA = 0, mB = 0;
            u32 outA = 0, outB = 0;
            while (t < 64) {
                const u32 p = (u32)__builtin_amdgcn_readlane((int)A.packed, (int)t);
                const u32 f = (p >> 6) & 7u, ol = p >> 9;
                if (f) { stop = f; break; }
                if (run - n_out + lo + ol > 256) { stop = 5; break; }
                mA |= 1ull << t;
                outA = (u32)lane == t ? run : outA;
                run += ol; t = (u32)__builtin_amdgcn_readfirstlane((int)(t + (p & 63u)));
            }
            if (!stop) while (t < 128) {
                const u32 p = (u32)__builtin_amdgcn_readlane((int)B.packed, (int)(t - 64));
                const u32 f = (p >> 6) & 7u, ol = p >> 9;
                if (f) { stop = f; break; }
                if (run - n_out + lo + ol > 256) { stop = 5; break; }
                mB |= 1ull << (t - 64);
                outB = (u32)lane == t - 64 ? run : outB;
                run += ol; t = (u32)__builtin_amdgcn_readfirstlane((int)(t + (p & 63u)));
            }
            INF_T(3);
            if (run > isize) { status = 5; break; }
            INF_N(10, __popcll(mA) + __popcll(mB));
            if (run > n_out) {
                // ---- the bytes of these tokens, put together in LDS.  Byte i of the window (text offset n_out - lo + i) is a literal,
                // or a copy of text from before the window (read from memory: one round trip for all tokens at once), or a copy of an
                // earlier byte of the window itself: s_ref[i] = that byte's index.  Following the references to their ends (pointer
                // jumping, a few LDS rounds while the reads from memory are on their way) gives every byte, however the matches lean on
                // each other -- bisulfite reads are three-letter text, zlib's matches there reach 20..40 bytes back -- and the
                // window is stored with one aligned dword per lane.
                // ---- (round 5) every lane puts together the four bytes of ITS dword of the window.  A token leaves a 32-bit record under its
                // id (= its bit offset in the step + 1: ids rise with the tokens' places in the text) and its id at the window byte it starts
                // at; a prefix maximum over the window's bytes tells every byte which token it belongs to.  A literal's byte is in the
                // record; a match's byte is a reference to an earlier byte of the window (s_ref), or text from before the window, read from
                // memory a byte at a time (the few lanes that need it; the others do not wait).  The round-4 form walked every token's
                // bytes in a loop of its own lane (up to 16 rounds per window, two token sets) and copied long matches with the whole wave,
                // one after the other: ~800 of a step's 1 140 instructions.
                const bool mineA = (mA >> lane) & 1ull, mineB = (mB >> lane) & 1ull;
                const u32 kindA = (A.e >> 5) & 7u, kindB = (B.e >> 5) & 7u;
                const bool matA = mineA && kindA == IK_BASE, matB = mineB && kindB == IK_BASE;
                if (__ballot((matA && A.mdist > outA) || (matB && B.mdist > outB))) { status = 6; break; }
                const u32 idxA = outA - n_out + lo, idxB = outB - n_out + lo;               // window index of the token's first byte
                const u32 hi = run - n_out + lo;                                            // window symbols [lo, hi) are text
                s_tid32[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                // record: bit 31 match; bits 0..7 window index of the first byte; bits 8..23 the distance, or the (one or two) literal bytes
                if (mineA) { s_rec[lane] = (matA ? 0x80000000u | (A.mdist << 8) : ((A.e >> 16) & 0xffffu) << 8) | idxA; s_tid[idxA] = (u8)(lane + 1); }
                if (mineB) { s_rec[64 + lane] = (matB ? 0x80000000u | (B.mdist << 8) : ((B.e >> 16) & 0xffffu) << 8) | idxB; s_tid[idxB] = (u8)(65 + lane); }
                __builtin_amdgcn_wave_barrier();
                u32 idk[4];
                {
                    const u32 t4 = s_tid32[lane];
                    idk[0] = t4 & 0xffu; idk[1] = max(idk[0], (t4 >> 8) & 0xffu); idk[2] = max(idk[1], (t4 >> 16) & 0xffu); idk[3] = max(idk[2], t4 >> 24);
                    u32 inc = idk[3];
                    for (int o = 1; o < 64; o <<= 1) { const u32 v = (u32)__shfl_up((int)inc, o, 64); if (lane >= o) inc = max(inc, v); }
                    u32 pre = (u32)__shfl_up((int)inc, 1, 64);
                    if (lane == 0) pre = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) idk[k] = max(idk[k], pre);
                }
                // (selects, not branches: every per-lane `if` costs the wave three scalar instructions of exec-mask bookkeeping, and the
                // scalar unit is what this kernel is bound by once two windows share the chip)
                u32 refs = 0, vals = 0, extm = 0, extq[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u32 wi = 4u * (u32)lane + (u32)k;
                    const bool in = wi >= lo && wi < hi;
                    const u32 rec = s_rec[(idk[k] - 1u) & 127u];
                    const u32 o = wi - (rec & 0xffu);
                    const bool mat = (rec >> 31) != 0;
                    const u32 d = (rec >> 8) & 0xffffu;
                    const bool inwin = wi >= lo + d;                                         // the source is an earlier byte of this window
                    const u32 ref = in && mat && inwin ? wi - d : wi;
                    const u32 val = in && !mat ? (rec >> (8 + 8 * (o & 1u))) & 0xffu : 0u;
                    extm |= (in && mat && !inwin ? 1u : 0u) << k;                            // ... or text from before it
                    extq[k] = n_out - lo + wi - d;
                    refs |= ref << (8 * k); vals |= val << (8 * k);
                }
                // bytes stored since the last fence are not visible to the other lanes yet: a fence when a source reaches into them
                {
                    u32 top = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) top = max(top, (extm >> k) & 1u ? extq[k] + 1 : 0u);
                    if (__ballot(top > fenced)) { __threadfence_block(); fenced = n_out; INF_N(13, 1); }
                }
                u32 ev[4] = {0, 0, 0, 0};
                if (extm) {
                    // (a byte that needs nothing reads the lane's first needed address again: one branch for the four loads)
                    const u32 q0 = extq[__builtin_ctz(extm)];
#pragma unroll
                    for (int k = 0; k < 4; k++) ev[k] = out[(extm >> k) & 1u ? extq[k] : q0];
                }
                s_ref32[lane] = refs;
                const bool inside = __ballot(refs != (u32)lane * 0x04040404u + 0x03020100u) != 0;
                __builtin_amdgcn_wave_barrier();
                if (inside) {
                    for (int it = 0; it < 9; it++) {
                        const u32 r4 = s_ref32[lane];
                        const u32 n4 = (u32)s_ref[r4 & 255u] | (u32)s_ref[(r4 >> 8) & 255u] << 8 | (u32)s_ref[(r4 >> 16) & 255u] << 16 | (u32)s_ref[r4 >> 24] << 24;
                        __builtin_amdgcn_wave_barrier();
                        s_ref32[lane] = n4;
                        __builtin_amdgcn_wave_barrier();
                        if (!__ballot(n4 != r4)) break;
                    }
                }
                INF_T(4);
                // the literals, and the text from before the window as it arrives
#pragma unroll
                for (int k = 0; k < 4; k++) vals |= ((extm >> k) & 1u ? ev[k] & 0xffu : 0u) << (8 * k);
                s_val32[lane] = vals;
                __builtin_amdgcn_wave_barrier();
                const u32 r4 = s_ref32[lane];
                const u32 w = (u32)s_val[r4 & 255u] | (u32)s_val[(r4 >> 8) & 255u] << 8 | (u32)s_val[(r4 >> 16) & 255u] << 16 | (u32)s_val[r4 >> 24] << 24;
                S* wp = out + n_out - lo + 4 * (u32)lane;
                const u32 r0 = 4 * (u32)lane;
                if (r0 >= lo && r0 + 4 <= hi) { *reinterpret_cast<u32*>(wp) = w; }
                else {
#pragma unroll
                    for (u32 k = 0; k < 4; k++) if (r0 + k >= lo && r0 + k < hi) wp[k] = (S)(w >> (8 * k));
                }
                last_byte = (S)((u32)__builtin_amdgcn_readlane((int)w, (int)((hi - 1) >> 2)) >> (8 * ((hi - 1) & 3u)));
                __builtin_amdgcn_wave_barrier();
                INF_T(5);
            }
            n_out = run; bp += t;
            const u32 tl = t & 63u;
            // the token the chain stopped at
            const u32 sp_ = t < 64 ? (u32)__builtin_amdgcn_readlane((int)A.packed, (int)tl) : (u32)__builtin_amdgcn_readlane((int)B.packed, (int)tl);
            if (stop == 1) { bp += sp_ & 63u; block_done = true; }
            else if (stop == 3) status = 7;
            else if (stop == 2 || stop == 4) {
                INF_N(14, 1);
                u32 ev = 0, lit = 0, sl = 0, sd = 0, used = 0;
                if (stop == 4) {
                    // a long match (a quality string): by the whole wave
                    ev = 2; used = sp_ & 63u;
                    sl = t < 64 ? (u32)__builtin_amdgcn_readlane((int)A.mlen, (int)tl) : (u32)__builtin_amdgcn_readlane((int)B.mlen, (int)tl);
                    sd = t < 64 ? (u32)__builtin_amdgcn_readlane((int)A.mdist, (int)tl) : (u32)__builtin_amdgcn_readlane((int)B.mdist, (int)tl);
                } else {
                // a code longer than the root tables (rare by construction): the lane the walk stopped at holds 64 bits from that
                // token's first bit -- more than any token has (15 + 5 + 15 + 13) -- and searches the canonical codes in LDS by itself
                if ((u32)lane == tl) {
                    u64 q = t < 64 ? vA : vB;
                    const u32 e = t < 64 ? A.e : B.e, kind = (e >> 5) & 7u;
                    int l = 0; int sy = -1;                                               // -1 invalid, -2 a length code from the root table
                    if (kind == IK_LONG) { sy = inf_search(s_cl, (u32)q & 0x7fffu, INF_LIT_ROOT + 1, 15, l); if (sy >= 0) { q >>= l; used += (u32)l; } }
                    else if (kind == IK_BASE) {
                        const u32 ex = (e >> 8) & 31u;
                        sy = -2; q >>= e & 31u; used += e & 31u;
                        sl = (e >> 16) + ((u32)q & ((1u << ex) - 1u)); q >>= ex; used += ex;
                    }
                    if (sy >= 257 && sy < 286) { const u32 ex = c_len_extra[sy - 257]; sl = c_len_base[sy - 257] + ((u32)q & ((1u << ex) - 1u)); q >>= ex; used += ex; }
                    if (sy == -1 || sy >= 286) ev = 9;
                    else if (sy >= 0 && sy < 256) { ev = 1; lit = (u32)sy; }
                    else if (sy == 256) ev = 3;
                    else {
                        const u32 d = s_dist[(u32)q & ((1u << INF_DIST_ROOT) - 1)];
                        const u32 dk = (d >> 5) & 7u;
                        if (dk == IK_LONG) {
                            const int ds = inf_search(s_cd, (u32)q & 0x7fffu, INF_DIST_ROOT + 1, 15, l);
                            if (ds >= 0 && ds < 30) { q >>= l; used += (u32)l; const u32 dx = c_dist_extra[ds]; sd = c_dist_base[ds] + ((u32)q & ((1u << dx) - 1u)); used += dx; ev = 2; }
                            else ev = 9;
                        } else if (dk == IK_BASE) {
                            const u32 dx = (d >> 8) & 31u;
                            q >>= d & 31u; used += d & 31u;
                            sd = (d >> 16) + ((u32)q & ((1u << dx) - 1u)); used += dx; ev = 2;
                        } else ev = 9;
                    }
                }
                ev = (u32)__builtin_amdgcn_readlane((int)ev, (int)tl); lit = (u32)__builtin_amdgcn_readlane((int)lit, (int)tl);
                sl = (u32)__builtin_amdgcn_readlane((int)sl, (int)tl); sd = (u32)__builtin_amdgcn_readlane((int)sd, (int)tl);
                used = (u32)__builtin_amdgcn_readlane((int)used, (int)tl);
                }
                bp += used;
                if (ev == 1) {
                    if (n_out + 1 > isize) { status = 5; break; }
                    if (lane == 0) out[n_out] = (S)lit;
                    n_out++; last_byte = (S)lit;
                } else if (ev == 2) {
                    if (sd > n_out || n_out + sl > isize) { status = 6; break; }
                    S myv = last_byte;
                    if (sd == 1) {
                        // a run of the byte in front of it (quality strings): no read at all
                        for (u32 i = lane; i < sl; i += 64) out[n_out + i] = myv;
                    } else {
                        // bytes stored since the last fence are not visible to the other lanes yet: a fence when the source reaches into them
                        if ((int)(sd >= sl ? n_out + sl : n_out + sd) - (int)sd > (int)fenced) { __threadfence_block(); fenced = n_out; INF_N(13, 1); }
                        for (u32 i = lane; i < sl; i += 64) { const u32 q = n_out - sd + (sd >= sl ? i : i % sd); myv = out[q]; out[n_out + i] = myv; }
                    }
                    last_byte = (S)__builtin_amdgcn_readlane((int)myv, (int)((sl - 1) & 63u));
                    n_out += sl;
                } else if (ev == 3) block_done = true;
                else status = 7;
            }
            INF_T(7);
        }
        // the next block header is read by lane 0 from bp
        if (lane == 0) { in.init(z + (bp >> 3), zend); in.refill(); in.drop((int)(bp & 7u)); }
    }
    status_r = status; n_out_r = n_out;
#ifdef INF_PROFILE
    prof[15] = clock64() - t_begin;
    if (lane == 0) for (int k = 0; k < 16; k++) atomicAdd(&g_inf_prof[k], prof[k]);
#endif
}

// CRC-32 of out[0, isize) by one wave (the table in LDS): every lane one segment, in four quarters side by side (a table look-up waits
// ~100 cycles for LDS: four chains in flight instead of one); the quarters are joined with x^(8 * quarter) -- the same factor in every
// lane but the last --, the lanes' segments with x^(8 * bytes behind them)
DEVI void inf_crc_table(u32* s_crc_tab, int lane)
{
    for (int t = 0; t < 4; t++) { u32 v = (u32)(lane + 64 * t); for (int k = 0; k < 8; k++) v = (v & 1) ? (v >> 1) ^ 0xedb88320u : v >> 1; s_crc_tab[lane + 64 * t] = v; }
}
DEVI u32 inf_crc32_wave(const u8* out, u32 isize, const u32* s_crc_tab, int lane)
{
    // every lane one segment, in four quarters side by side (a table look-up waits ~100 cycles for LDS: four chains in flight
    // instead of one); the quarters are joined with x^(8 * quarter) -- the same factor in every lane but the last
    const u32 seg = (((isize + 63) / 64) + 3) & ~3u, q = seg / 4;
    const u32 xq = crc_x8n(q);
    u32 cc[4], ea[4], ee[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 a = (u32)lane * seg + (u32)k * q;
        ea[k] = a < isize ? a : isize; ee[k] = a + q < isize ? a + q : isize;
        cc[k] = 0xffffffffu;
    }
    // four bytes of every quarter per step (one unaligned load each, issued before the sixteen look-ups that use them)
    for (u32 i = 0; i + 4 <= q; i += 4) {
        u32 w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { w[k] = 0; if (ea[k] + i + 4 <= ee[k]) __builtin_memcpy(&w[k], out + ea[k] + i, 4); }
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (ea[k] + i + 4 <= ee[k]) cc[k] = s_crc_tab[(cc[k] ^ (w[k] >> (8 * bb))) & 0xffu] ^ (cc[k] >> 8);
        }
    }
    // (what is left of a quarter: up to three bytes, or the whole of a quarter that the text ends in)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 len = ee[k] - ea[k];
        for (u32 i = len & ~3u; i < len; i++) cc[k] = s_crc_tab[(cc[k] ^ out[ea[k] + i]) & 0xffu] ^ (cc[k] >> 8);
    }
    u32 c = ~cc[0];
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const u32 lk = ee[k] - ea[k];
        if (lk) c = crc_multmodp(lk == q ? xq : crc_x8n(lk), c) ^ ~cc[k];
    }
    u32 x = ea[0] < ee[3] ? crc_multmodp(crc_x8n(isize - ee[3]), c) : 0u;
    for (int o = 32; o > 0; o >>= 1) x ^= __shfl_down(x, o, 64);
    x = (u32)__shfl((int)x, 0);
    return x;
}

// comp: the compressed bytes of n BGZF blocks, block b at comp + blk_off[b] (blk_off[n] = end); its text goes to text + out_off[b]
// (out_off[b + 1] - out_off[b] = the ISIZE its trailer states).  err[b] = 0 when the block was well-formed and its CRC matched.
__global__ void __launch_bounds__(64, 4)
k_bgzf_inflate(const u8* __restrict__ comp, const u64* __restrict__ blk_off, const u64* __restrict__ out_off, long n, char* __restrict__ text, u32* __restrict__ err)
{
    __shared__ u32 s_crc_tab[256];
    const long b = blockIdx.x;
    if (b >= n) return;
    const int lane = threadIdx.x;
    const u8* z = comp + blk_off[b];
    const u64 zlen = blk_off[b + 1] - blk_off[b];
    char* out = text + out_off[b];
    const u32 isize = (u32)(out_off[b + 1] - out_off[b]);
    u32 status = 0;                                       // 0 ok so far
    // ---- gzip member header (BGZF: FEXTRA with the BC subfield; any other well-formed header is read too)
    u32 body = 0;
    if (zlen < 18 + 8 || z[0] != 0x1f || z[1] != 0x8b || z[2] != 8 || (z[3] & 0xe0)) status = 1;
    else {
        const int flg = z[3];
        u64 q = 10;
        if (flg & 4) { const u64 xl = (u64)z[q] | ((u64)z[q + 1] << 8); q += 2 + xl; }
        if ((flg & 8) && q < zlen) { while (q < zlen && z[q]) q++; q++; }
        if ((flg & 16) && q < zlen) { while (q < zlen && z[q]) q++; q++; }
        if (flg & 2) q += 2;
        if (q + 8 > zlen) status = 1;
        body = (u32)q;
    }
    inf_crc_table(s_crc_tab, lane);
    __syncthreads();
    u32 n_out = 0;
    if (!status) {
        inf_wave(z, z + zlen - 8, body * 8, reinterpret_cast<u8*>(out), isize, lane, status, n_out);
    }
    // ---- trailer: ISIZE and CRC-32 (every lane one segment, the segments combined by x^(8 * bytes behind them))
    if (!status) {
        const u8* t = z + zlen - 8;
        const u32 want_crc = (u32)t[0] | (u32)t[1] << 8 | (u32)t[2] << 16 | (u32)t[3] << 24;
        const u32 want_len = (u32)t[4] | (u32)t[5] << 8 | (u32)t[6] << 16 | (u32)t[7] << 24;
        if (n_out != isize || want_len != isize) status = 8;
        else {
            __threadfence_block();
            const u32 x = inf_crc32_wave(reinterpret_cast<const u8*>(out), isize, s_crc_tab, lane);
            if (x != want_crc) status = 9;
        }
    }
    if (lane == 0) err[b] = status;
}

// a window that ends the file: an unterminated last line gets its newline (the text buffer has the room); *added = 1 when it did
__global__ void k_close_last_line(char* __restrict__ text, u64 bytes, u64* __restrict__ added)
{
    if (text[bytes - 1] != '\n') { text[bytes] = '\n'; *added = 1; } else *added = 0;
}

// newlines of the inflated text per 64 KiB of the caller's window: text byte i sits at window offset shift + i; counts[j] = newlines
// among the bytes whose window offset lies in [j * 65536, (j + 1) * 65536) (the driver's reader needs nothing else from the text)
__global__ void __launch_bounds__(256)
k_nl_count64k(const char* __restrict__ text, u64 total, u64 shift, u32* __restrict__ counts)
{
    __shared__ u32 sh[4];
    const u64 j = blockIdx.x;
    const u64 w0 = j << 16, w1 = w0 + 65536;
    const u64 a = w0 > shift ? w0 - shift : 0, e = w1 > shift ? (w1 - shift < total ? w1 - shift : total) : 0;
    u32 c = 0;
    // 16 bytes per thread and step where the range allows (the text buffer is 16-byte aligned, `a` is not in general)
    u64 i = a + threadIdx.x;
    const u64 a16 = (a + 15) & ~15ull, e16 = e & ~15ull;
    if (a16 < e16) {
        for (u64 q = a + threadIdx.x; q < a16; q += 256) c += text[q] == '\n';
        const uint4* p = reinterpret_cast<const uint4*>(text + a16);
        const u64 n16 = (e16 - a16) >> 4;
        for (u64 q = threadIdx.x; q < n16; q += 256) {
            const uint4 v = p[q];
            c += (u32)__popc(nl_mask4(v.x)) + (u32)__popc(nl_mask4(v.y)) + (u32)__popc(nl_mask4(v.z)) + (u32)__popc(nl_mask4(v.w));
        }
        for (u64 q = e16 + threadIdx.x; q < e; q += 256) c += text[q] == '\n';
    } else for (; i < e; i += 256) c += text[i] == '\n';
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[j] = sh[0] + sh[1] + sh[2] + sh[3];
}

#endif
