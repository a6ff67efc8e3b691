// bitmapperbs_amd/csrc/k_finalize.hip -- finalize: MAPQ, placement, stats
// (one stage of the mapping path; included by bmbs_kernels.hip, in the order the stages run: no translation unit of its own)
// ================================================================================================
// finalize: MAPQ, placement, stats
// ================================================================================================
DEVI u16 sat16(u32 v) { return v > 0xffffu ? (u16)0xffffu : (u16)v; }

// mapq_lut[(ed) * (range+1) + sd]: MAP_Calculation (Schema.cpp:168-405) tabulated on the host in
// IEEE double for this k: ed = min(second_best_diff, k+1), sd = clamp(score + range, 0, range).
__global__ void __launch_bounds__(256)
k_finalize(DevIndex ix, ScoreParams sp, const int* __restrict__ pen_lut, const u8* __restrict__ mapq_lut,
           const u32* __restrict__ mapq_off, int unit, const char* __restrict__ seq, const char* __restrict__ qual, ReadGeom gm,
           int stride, long n, ReadState st, const int* __restrict__ a_start, const int* __restrict__ a_end,
           const u32* __restrict__ a_nm, const int* __restrict__ a_score, const int* __restrict__ a_nops,
           int max_ops, u32 cigar_base, int ambiguous_out, const u64* __restrict__ sp0, const u32* __restrict__ hits0,
           bmbs_result_dev* __restrict__ res, unsigned long long* __restrict__ stats)
{
    __shared__ unsigned long long sh[5];
    __shared__ u64 s_cs[BMBS_CS_LDS];
    if (threadIdx.x < 5) sh[threadIdx.x] = 0;
    const u64* cs = chrom_table(ix, s_cs);
    __syncthreads();
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;          // this lane's contribution to the five counters
    if (r < n) {
        bmbs_result_dev o;
        o.pos = 0; o.cigar_off = 0; o.chrom = -1; o.status = 0; o.mapq = 0; o.flag = 0; o.nm = 0; o.score = 0;
        o.n_cigar = 0; o.path = 0; o.n_cand = sat16(st.n_cand[r]); o.tlen = 0;
        const int verdict = st.verdict[r];
        const int L = gm.rl(r), k = gm.rk(L);
        const int range = unit * k;
        mapq_lut += mapq_off[k];                  // MAP_Calculation table of this read's own threshold
        bool have = false, amb = false;
        u64 site = 0; long long start_site = 0, end_site = 0;
        u32 nm = 0; int score = 0; u32 sbd = 0; int mapq = 0;
        if (verdict == 1) { have = true; site = st.exit_site[r]; start_site = 0; end_site = L - 1; nm = 0; score = 0; mapq = 42; o.path = 1; }
        else if (verdict == 2) {
            have = true; site = st.exit_site[r]; start_site = 0; end_site = L - 1; nm = 1; o.path = 2;
            const int mv = st.mm_site[r], ms = mv & 0x7fff;          // bit 15: the read has 'N' there (k_seed_decide)
            score = (mv & 0x8000) ? -sp.np : -pen_lut[(unsigned char)qual[(size_t)r * stride + ms]];
            sbd = 0xffffffffu;
        } else if (verdict == 4) {
            o.status = 2; o.path = 4;
            if (ambiguous_out) {
                // output_ambiguous_exact_map (Schema.cpp:24072-24115): first row in SA order whose placement stays inside
                // its chromosome, MAPQ 1; none -> the read counts as unmapped
                const u64 sp_ = sp0[r];
                u32 nh = hits0[r]; if (nh > 1000u) nh = 1000u;
                o.status = 3;
                for (u32 i = 0; i < nh; i++) {
                    const u64 s_ = ix.total - sa_at(ix, sp_ + i) - (u64)L;
                    u64 loc = s_; int flag;
                    if (loc >= ix.G) { loc = ix.G * 2 - (loc + (u64)(L - 1)) - 1; flag = 16; } else flag = 0;
                    int c = 0;
                    c = chrom_of(cs, ix.n_chrom, loc);
                    if (c >= ix.n_chrom) continue;
                    const u64 pos = loc + 1 - cs[c];
                    if (pos + (u64)(L - 1) > cs[c + 1] - cs[c]) continue;
                    o.pos = pos; o.chrom = c; o.flag = (u16)flag; o.mapq = 1; o.status = 2;
                    break;
                }
            }
        }
        else if (verdict == 3) {
            o.path = 3;
            const int rs = st.red_status[r];
            if (rs == 1 || (rs == 2 && ambiguous_out)) {
                amb = rs == 2;
                have = true; site = st.best_site[r]; sbd = st.sbd[r];
                if (st.job_flag[r]) {
                    const u64 jb = st.job_off[r];
                    start_site = a_start[jb]; end_site = a_end[jb]; nm = a_nm[jb]; score = a_score[jb];
                    const int no = a_nops[jb];
                    o.cigar_off = cigar_base + (u32)(jb * (u64)max_ops);
                    o.n_cigar = no < 0 ? 255 : (u8)no;
                } else { end_site = st.best_end[r]; start_site = end_site - L + 1; nm = 0; score = 0; }
            } else if (rs == 2) o.status = 2;
        }
        if (have) {
            if (verdict != 1) {
                int sd = score + range; if (sd < 0) sd = 0; if (sd > range) sd = range;
                const u32 ed = sbd > (u32)k ? (u32)k + 1 : sbd;
                mapq = mapq_lut[(size_t)ed * (range + 1) + sd];
            }
            // output_sam_end_to_end placement (Schema.cpp:11941-11986)
            u64 loc = site; int flag;
            if (loc >= ix.G) { loc = loc + (u64)end_site; loc = ix.G * 2 - loc - 1; flag = 16; }
            else { loc = loc + (u64)start_site; flag = 0; }
            int c = 0;
            c = chrom_of(cs, ix.n_chrom, loc);
            bool ok = c < ix.n_chrom;
            u64 pos = 0;
            if (ok) {
                pos = loc + 1 - cs[c];
                const u64 clen = cs[c + 1] - cs[c];
                if (pos + (u64)end_site - (u64)start_site > clen) ok = false;
            }
            o.pos = pos; o.chrom = c < ix.n_chrom ? c : -1; o.flag = (u16)flag; o.mapq = (u8)mapq;
            o.nm = (u16)nm; o.score = (int16_t)score;
            o.status = ok ? (amb ? 2 : 1) : 3;
        }
        res[r] = o;
        s0 = 1;
        if (o.status == 1) { s1 = 1; s3 = (u32)L; s4 = (u32)nm; }
        else if (o.status == 2) s2 = 1;
    }
    wave_stats_add(sh, s0, s1, s2, s3, s4);
    __syncthreads();
    if (threadIdx.x < 5 && sh[threadIdx.x]) atomicAdd(&SHARD(stats)[threadIdx.x], sh[threadIdx.x]);
}
