// bitmapperbs_amd/csrc/bmbs_host.h -- host-side state of the library (lanes, buffers, the public handle) and the functions its translation
// units share: bmbs_api.hip (mapping kernels, the stage and mapping entry points) and bmbs_textpath.hip (the two ends of the file path:
// FASTQ text / BGZF in, SAM text / BAM out).  Round 5 split them: one 3 000-line file had rebuilt ~85 kernels for a one-line change.
#ifndef BMBS_HOST_H
#define BMBS_HOST_H
#include "../../include/bmbs.h"
#include "bmbs_dev.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <sys/syscall.h>
#include <unistd.h>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>


struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    bool arena = false;                 // carved out of the lane's arena: never freed on its own
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Work buffers of a lane come out of a few large slabs: hipMalloc / hipFree wait for the whole device, and a lane's first call
// used to make ~70 of them -- with three contexts on one GPU every such call stalled the other two (bmbs_search, first batches:
// 50-140 ms instead of 2).  Buffers only ever grow; a grown buffer leaves its old region behind until the lane is destroyed.
struct Arena {
    struct Slab { char* p; size_t cap, used; };
    std::vector<Slab> slabs;
    size_t total = 0;
    void* alloc(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        for (size_t i = slabs.size(); i-- > 0;)
            if (slabs[i].used + bytes <= slabs[i].cap) { void* r = slabs[i].p + slabs[i].used; slabs[i].used += bytes; return r; }
        size_t cap = std::max<size_t>(bytes, std::min<size_t>((size_t)2 << 30, std::max<size_t>((size_t)512 << 20, total / 2)));
        void* p = nullptr;
        if (hipMalloc(&p, cap) != hipSuccess) {
            cap = bytes;
            if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
        }
        slabs.push_back({(char*)p, cap, bytes});
        total += cap;
        return p;
    }
    void free_all() { for (auto& s : slabs) (void)hipFree(s.p); slabs.clear(); total = 0; }
};

struct Prof { const char* name; hipEvent_t a, b; bool used; };

// Switches (environment), read ONCE when a context is created -- never on the launch path.  Every form gives identical results and the
// GPU suite runs each one in single-end and paired-end mode (test_ab_switches_give_identical_records); DESIGN.md section 3 lists them.
//   forms:      BMBS_LEGACY=1 (the round-1 ASCII-row seeding engine and byte-wise mate preparation, as a whole), BMBS_SW=reg2|reg|wave,
//               BMBS_KGRAM=0|1|2, BMBS_T20=0, BMBS_TDEPTH=20|21, BMBS_WIDE=1 (+ BMBS_SUPER_SHIFT), BMBS_LANES=n, BMBS_EXACT=1, BMBS_SEED_WAVES=n
//   test aids:  BMBS_CAP_SCALE, BMBS_SPLIT_MIN, BMBS_CHUNK, BMBS_PEF_LONG=2 (make small inputs reach the paths large ones take)
struct Knobs {
    int sw_form = 0;            // BMBS_SW: 0 default (reg2 from k = 5), 1 reg, 2 reg2, 3 wave
    bool rows_ascii = false;    // BMBS_LEGACY=1: seeding on the ASCII rows (no packed copy), mate 2 prepared byte-wise with its full ASCII text
    int seed_waves = 65536;     // BMBS_SEED_WAVES
    bool exact = false;         // BMBS_EXACT=1: every call waits for its stage counts (the round-2 launch sequence)
    int lanes = 3;              // BMBS_LANES: chunks of a device call in flight per context (each on a lane = a stream of its own)
    int host_lanes = 4;         // BMBS_HOST_LANES: lanes the host-buffer entry points deal their chunks to (>= lanes): upload, kernels and download of
                                // a chunk follow each other on a lane, so what overlaps is what different lanes do -- 2 M pairs through
                                // bmbs_map_pe_packed: 164 M reads/s on two lanes, 180 on four (tools/hostbuf_probe.py)
    long chunk = 0;             // BMBS_CHUNK: units per chunk of a split call (0: n / lanes, at least BMBS_SPLIT_MIN)
    long split_min = 250000;    // BMBS_SPLIT_MIN: calls with fewer units than twice this run on one lane
    int up_turns = 1;           // BMBS_UP_TURNS: 0 the lanes of a host-buffer call upload side by side (the round-5 form)
    int scan_chain = 0;         // BMBS_SCAN_CHAIN: 1 every scan is ONE launch (k_scan_chain, a chained scan with look-back) instead of two: eight launches fewer per
                                // paired-end chunk, 0.5-1 % fewer reads/s (tools/scan_kbench.hip: 52 against 45 us at 10 M entries) -- the lanes are bound by
                                // GPU time, not by launches, so the two-launch form stays the default
    int prefilter = 1;          // BMBS_PREFILTER: 0 the paired-end long-list kernels sort every located site (the round-5 form; A/B runs, tests)
    int pef_long = 1;           // BMBS_PEF_LONG: 1 long lists of k_pe_filter_pairs get a wave when the input is repeat-rich, 2 always (tests)
    int kgram = 1;              // BMBS_KGRAM: 0 no trigram table, 1 (default) its kernels are used once a context has seen reads that walk the index in long chains, 2 always
    double cap_scale = 1.0;     // BMBS_CAP_SCALE: scales the learned capacities (tests: a small value forces the repeat-with-exact-sizes path)
    static const bool copy_streams = true, copy_lock = true, arena = true;     // (measured in round 3; the alternatives are gone)
    void read()
    {
        auto is = [](const char* e, const char* v) { return e && !strcmp(e, v); };
        const char* e = getenv("BMBS_SW");
        sw_form = is(e, "reg") ? 1 : is(e, "reg2") ? 2 : is(e, "wave") ? 3 : 0;
        rows_ascii = is(getenv("BMBS_LEGACY"), "1");
        if ((e = getenv("BMBS_SEED_WAVES"))) seed_waves = atoi(e);
        exact = is(getenv("BMBS_EXACT"), "1");
        if ((e = getenv("BMBS_LANES"))) lanes = atoi(e);
        if (lanes < 1) lanes = 1;
        if (lanes > 8) lanes = 8;
        if ((e = getenv("BMBS_HOST_LANES"))) host_lanes = atoi(e);
        if (host_lanes < lanes) host_lanes = lanes;
        if (host_lanes > 8) host_lanes = 8;
        if ((e = getenv("BMBS_KGRAM"))) kgram = atoi(e);
        if ((e = getenv("BMBS_CHUNK"))) chunk = atol(e);
        if ((e = getenv("BMBS_SPLIT_MIN"))) split_min = atol(e);
        if (split_min < 1) split_min = 1;
        if ((e = getenv("BMBS_CAP_SCALE"))) cap_scale = atof(e);
        if ((e = getenv("BMBS_PEF_LONG"))) pef_long = atoi(e);
        if ((e = getenv("BMBS_PREFILTER"))) prefilter = atoi(e);
        if ((e = getenv("BMBS_UP_TURNS"))) up_turns = atoi(e);
        if ((e = getenv("BMBS_SCAN_CHAIN"))) scan_chain = atoi(e);
    }
};

// one call that has been enqueued on a lane and not been waited for yet (bmbs_sync / the next call on the lane settles it)
struct Pending {
    bool pe = false, exact = false;
    int slot = 0;
    uint64_t a[4] = {0, 0, 0, 0};            // d_seq(1), d_qual(1), d_seq2, d_qual2
    const u16* d_len = nullptr;
    int32_t L = 0, stride = 0;
    int64_t n = 0;
    uint64_t d_results = 0, d_cigar_pool = 0;
    int64_t cigar_cap = 0;
    u32 cigar_base = 0;
    bool prepared = false;                   // map_pe_dev: the lane's pe_seq already holds the rows (FASTQ-text entry point)
    int packed_hw = 0;                       // > 0: a[0] (a[2]: mate 2) are the caller's PACKED rows, packed_hw words apart (bmbs_map_*_packed)
    bool staged = false;                     // the call reads lane-owned staging buffers, which the lane's next call overwrites
};

// The trigram rank table (DevIndex::occ3, 27.9 GB at GRCh38 size) is built when a context that uses the index has seen reads that walk
// it in long chains (lr_chain >= 3 extensions per lookup: a repeat-rich genome) -- not at attach, where round 4 built it for every
// index: on a repeat-poor genome it was never used and cost a third context its work buffers.  One object per attached index, shared
// by the owner's lanes and by every context that shares the index; the table's memory belongs to it.
struct Occ3Shared {
    std::mutex mu;
    bool tried = false;
    int skipped = 0;                    // builds put off because the device had no room for the table beside its margin
    int dev = 0;
    DevIndex base;                      // the index without the table (what the builder reads)
    u64 rows = 0;
    void* occ3 = nullptr; void* c3 = nullptr; u64 nb3 = 0;
    ~Occ3Shared() { if (occ3 || c3) { (void)hipSetDevice(dev); if (occ3) (void)hipFree(occ3); if (c3) (void)hipFree(c3); } }
};

// One lane = one stream with its own work buffers, counters and HIP-event profile: what a whole context was in round 2.  A context
// owns BMBS_LANES (default three, device_lanes) of them on one attached index and deals the chunks of a call to them, so that the issue-bound kernels (DP, Myers,
// row preparation) of one chunk run beside the memory-bound seeding kernels of another without a second context or host thread.
struct Lane {
    int dev = 0;
    hipStream_t stream = nullptr;
    // copies to and from host buffers go on streams that never carry a kernel: a hipMemcpyAsync on a stream that also runs kernels
    // moved 26-30 GB/s on the MI355X boxes (ROCm 7.2), on a stream of its own 56 (tools/e2e_trace.sh)
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    hipEvent_t ev_up = nullptr, ev_k = nullptr;
    bmbs_params prm;
    ScoreParams sp;
    Knobs kn;
    Arena arena;
    std::string err;
    bool attached = false;
    DevIndex ix;
    u64 rows = 0;
    // index buffers
    DevBuf occ, hash, sa, gen2, gen2p, chrom_start, t20;
    std::shared_ptr<Occ3Shared> o3;            // the index's trigram table, built on demand (occ3_want)
    // LUTs
    DevBuf pen_lut, mapq_lut;
    bool luts_ready = false;
    int mapq_unit = 0;                         // max(gap_open + gap_ext, mp_max): score range per unit of threshold
    DevBuf mapq_off, klut;                     // per-threshold table offsets; threshold per read length
    // per-read workspace
    DevBuf verdict, n_seeds, multi, mm_site, exit_site, seeds, n_cand, cand_off, n_votes, best_site,
        best_end, best_err, sbd, red_status, job_flag, job_off, scan_tmp, totals;
    // chained scan (k_scan_chain): scan_tmp holds the tiles' status words, scan_ticket the tile ticket; the host counts the tickets
    // the launches so far took and the scans so far (the epoch the status words carry)
    DevBuf scan_ticket;
    u32 scan_ticket_base = 0, scan_epoch = 0;
    // per-candidate / per-job workspace
    DevBuf vote_list;
    DevBuf cand, votes, slot_read, vote_off, votes_dense, dense_read, ferr, fend, job_read, job_site, job_end, job_err, need_sw, sw_off, sw_job, trace,
        a_start, a_end, a_nm, a_score, a_nops;
    // host-variant staging
    DevBuf in_seq, in_qual, out_res, cig_pool, in_a, in_b, in_c, in_d, in_len;
    DevBuf pe_mid_flag, pe_mid_list;                    // k_vote_pe_mid work list
    DevBuf big_list;                                    // reads whose lists exceed the wave form of k_vote_[pe_]long
    DevBuf wavelog_buf, wavelog_count; std::string wavelog_path;    // BMBS_WAVELOG diagnostic
    DevBuf prow, prow_dirty;                            // packed copy of the read rows (k_pack_rows), one dirty byte per read
    double link_up_s = 0, link_down_s = 0, text_call_s = 0; u64 text_calls = 0;       // bmbs_text_times: wall seconds the text calls' copies held the link
    DevBuf pk_in1, pk_in2, pk_ascii;                    // bmbs_map_*_packed: the caller's packed rows as uploaded; single end: the sparse ASCII rows
    DevBuf fq_text1, fq_text2, fq_idx;                  // bmbs_map_*_fastq: FASTQ text windows and the per-record line index
    // bmbs_map_*_text: newline index built on the device, SAM text written on the device
    DevBuf tx_tilecnt, tx_tileoff, tx_nl[2], tx_rec[2], tx_info, sam_len, sam_off, sam_out, chrom_chars, chrom_off;
    DevBuf bam_raw, bam_slots, bam_slot_len, bam_off, stats_snap;      // --bam: record stream, deflate scratch, BGZF slots
    DevBuf z_comp, z_off, z_text, z_err, z_nl, z_comp2, z_off2, z_err2;          // bmbs_inflate_bgzf; (…2: mate 2 of bmbs_text_open_bgzf)
    struct OpenText { bool valid = false, pe = false; u64 bytes1 = 0, bytes2 = 0; int64_t n = 0; } open_text;      // between bmbs_text_open_bgzf and bmbs_text_map_open
    u32* h_info = nullptr;                              // page-locked: 8 info words + 4 totals of the text path
    int n_refs = 0, max_ref_len = 0;
    // paired-end workspace
    DevBuf sd_sp0, sd_hits0, sd_ml0, sd_tm, sd_seed_id, sd_clen, sd_first_ml, sd_flag_c, sd_flag_d, sd_off_c, sd_off_d, sd_list_c, sd_list_d;
    DevBuf pe_seq, pe_B, pe_occ, pe_len, pe_cur, pe_vround, pe_dead, pe_both, pe_npair, pe_sbd, in_seq2, in_qual2;
    DevBuf pe_first, pe_full, pe_R, pe_roff, pe_rflag, pe_rscan, pe_rlist, pe_rcnt, pe_ritem_off, pe_rcand;     // --sensitive
    u64 last_reseeded = 0, last_reseed_cand = 0;
    DevBuf stats, call_stats, flags, counters, long_flag, long_off, long_list;     // long_*: reads whose candidate lists go to k_vote_long
    // HIP-event profile: one set of event pairs per call in flight (slot); lane_settle reads a settled call's set into `last`
    // (bmbs_profile_last) and adds it to `acc` (bmbs_profile_total: sums since the last reset, read without forcing a wait per call)
    std::vector<Prof> profset[8];
    int prof_used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cur_slot = 0;
    struct ProfSum { const char* name; double ms; };
    std::vector<ProfSum> last, acc;
    u64 acc_calls = 0;
    u64 last_total_cand = 0, last_n_jobs = 0;
    int last_max_ops = 0;
    double lr_chain = 0;                       // extensions per 16-mer lookup of the last settled call (three-letter steps from 3 on)
    u64 h_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // ---- launches without host round trips: what the stages of earlier calls needed, per read (0: nothing known yet -> the
    // first call of a lane waits for its counts); the pinned words the device leaves its counts and guard flags in; the call in flight
    double lr_cand = 0, lr_sw = 0, lr_rcand = 0, lr_long = 0;
    u64* h_tot = nullptr;                      // page-locked: per call in flight, totals[16] followed by the flag words (call_end)
    std::deque<Pending> inflight;
    int next_slot = 0;
    u64 n_retries = 0;
};


// the public handle: parameters + the lanes (lane 0 owns the index unless the context shares another one's)
struct bmbs_ctx {
    int dev = 0;
    bmbs_params prm;
    Knobs kn;
    std::string err;
    std::vector<Lane*> lanes;
    int used_lanes = 1;                        // lanes the last mapping call ran on (profile / counters aggregate over them)
    int next_lane = 0;
};

#define HIPCHK(c, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                 \
            return BMBS_ENODEV;                                                                           \
        }                                                                                                 \
    } while (0)
#define ENS(c, buf, bytes) do { int rc_ = ensure((c), (buf), (bytes)); if (rc_) return rc_; } while (0)

// ---- shared by the translation units (defined in bmbs_api.hip unless noted) -----------------------------------------------------------
int ensure(Lane* c, DevBuf& b, size_t bytes, bool zero = false);
void release(DevBuf& b);
inline unsigned nblk(u64 n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }
void prof_begin(Lane* c, const char* name);
void prof_end(Lane* c);
int scan_u32(Lane* c, const u32* in, u64 n, u64* out, int slot, u32* list = nullptr, int nz = 0, const u64* n_dev = nullptr);
int threshold_k(const bmbs_params& P, int L);
int cigar_ops_bound(const bmbs_params& P, int L, int k);
int lane_settle(Lane* c);
int lane_enqueue(Lane* c, Pending P, bool staged);
int settle_all(bmbs_ctx* X);
inline Lane* lane0(bmbs_ctx* X) { return X && !X->lanes.empty() ? X->lanes[0] : nullptr; }
inline int fin(bmbs_ctx* X, Lane* c, int rc) { if (rc && X && c) X->err = c->err; return rc; }
// the stage entry points and the text calls run on lane 0, behind everything the context's lanes have in flight
#define ON_LANE0(call) Lane* c = lane0(X); if (!c) return BMBS_EINVAL; { const int rs_ = settle_all(X); if (rs_) return rs_; } X->used_lanes = 1; return fin(X, c, call)
extern std::mutex g_h2d_mu[16], g_d2h_mu[16];     // one copy per direction and device at a time (bmbs_textpath.hip)
void textpath_device_init(Lane* c);               // bmbs_textpath.hip: the constants its kernels read (CRC-32 powers)
#endif
