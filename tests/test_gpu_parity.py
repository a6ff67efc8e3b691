"""-m gpu: parity of the HIP path (through the C-ABI) with the oracle on identical seeded inputs, against
the committed golden SAM of the reference, and -- at BASELINE.json's full batch size -- through
size-independent properties.  Bit-exact everywhere: integer / byte / index work."""
import gzip
import os

import numpy as np
import pytest

import orc
from common import GOLD, ROOT, e_of, golden_args, gunzip_to, plant_repeats, read_fastq

pytestmark = pytest.mark.gpu


def compare_records(res, pool, recs, L, amb=False):
    """amb: --ambiguous_out, ambiguous records (status 2) carry a full alignment too"""
    from bitmapperbs_amd import mapper
    bad = []
    assert res.size == recs.size
    st_ok = res["status"].astype(np.int64) == recs["status"].astype(np.int64)
    for i in np.nonzero(~st_ok)[0][:5]:
        bad.append((int(i), "status", int(res[i]["status"]), int(recs[i]["status"])))
    sel = (recs["status"] == 1) | ((recs["status"] == 3) & (recs["path"] != 4))
    if amb:
        sel |= recs["status"] == 2
    mapped = np.nonzero(st_ok & sel)[0]
    for f in ("chrom", "pos", "flag", "mapq", "nm", "score", "path"):
        neq = mapped[res[f][mapped].astype(np.int64) != recs[f][mapped].astype(np.int64)]
        for i in neq[:5]:
            bad.append((int(i), f, int(res[i][f]), int(recs[i][f])))
    Ls = np.broadcast_to(np.asarray(L), (res.size,))           # scalar or per-read lengths
    for i in mapped:
        Li = int(Ls[i])
        if int(res[i]["n_cigar"]) or recs[i]["cigar"] != b"%dM" % Li:
            if mapper.cigar_text(res[i], pool, Li) != recs[i]["cigar"].decode():
                bad.append((int(i), "cigar", mapper.cigar_text(res[i], pool, Li), recs[i]["cigar"].decode()))
    return bad


@pytest.fixture(scope="module")
def env(tmp_path_factory):
    """a repeat-rich 3-chromosome genome + product-built index + both handles"""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from bitmapperbs_amd import synth, mapper
    wd = tmp_path_factory.mktemp("gpu")
    names, chroms = synth.make_genome(1_500_000, 3, seed=77)
    plant_repeats(chroms, seed=78)
    fa = str(wd / "g.fa")
    synth.write_fasta(fa, names, chroms)
    mapper.Index.build(fa, fa, threads=8)
    return dict(fa=fa, chroms=chroms, ix=mapper.Index(fa), oix=orc.OrcIndex(fa), wd=str(wd))


CASES = [
    dict(n=30000, L=100, seed=1, sub=0.005, indel=0.0002, qual="const", e=0.08),
    dict(n=30000, L=150, seed=2, sub=0.02, indel=0.002, qual="random", n_rate=0.002, e=0.04),
    dict(n=20000, L=150, seed=3, sub=0.04, indel=0.003, qual="random", conv=0.8, e=0.08),
    dict(n=8000, L=250, seed=4, sub=0.03, indel=0.001, qual="random", e=0.08),
    dict(n=20000, L=75, seed=5, sub=0.04, indel=0.004, qual="random", n_rate=0.01, e=0.08),
    dict(n=10000, L=36, seed=6, sub=0.02, indel=0.0, qual="random", e=0.1),
    dict(n=2000, L=400, seed=7, sub=0.03, indel=0.001, qual="random", e=0.08),        # k = 31 cap... (32 -> 31)
    # 401 .. 998 bases (the reference itself breaks at 999 and 1000, tests/golden/make_golden.py): k stays 31, 25 seeds, rows beyond the LDS slots of the short-read forms
    dict(n=1500, L=600, seed=21, sub=0.02, indel=0.002, qual="random", e=0.08),
    dict(n=1000, L=998, seed=22, sub=0.015, indel=0.0015, qual="random", n_rate=0.001, e=0.08),
    dict(n=1000, L=998, seed=23, sub=0.004, indel=0.0005, qual="const", e=0.02),     # k = 19 on 998 bases
    dict(n=1000, L=777, seed=24, sub=0.03, indel=0.003, qual="random", e=0.08),
    dict(n=5000, L=150, seed=8, sub=0.0, indel=0.0, qual="const", conv=0.0, e=0.0),   # k = 0
    # --ambiguous_out: one hit of each ambiguous read is aligned and returned (exact-ambiguous and tie-in-the-filter forms)
    dict(n=30000, L=100, seed=9, sub=0.01, indel=0.001, qual="random", e=0.08, amb=1),
    dict(n=20000, L=60, seed=10, sub=0.0, indel=0.0, qual="const", conv=0.0, e=0.08, amb=1),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "L%d_e%.2f_s%d%s" % (c["L"], c["e"], c["seed"], "_amb" if c.get("amb") else ""))
def test_map_se_records_and_stats_match_oracle(case, env):
    from bitmapperbs_amd import synth, mapper
    c = dict(case); e = c.pop("e"); amb = c.pop("amb", 0)
    r = synth.make_reads_se(env["chroms"], **c)
    L = c["L"]
    m = mapper.Mapper(env["ix"], 0, e_f=e, ambiguous_out=amb)
    res, pool = m.map_se(r["seq"], r["qual"], L)
    recs, ost, cnt = env["oix"].map_se(orc.params(e_f=e, ambiguous_out=amb), r["seq"], r["qual"], L)
    if amb:
        assert (recs["status"] == 2).sum() > 20          # the case really has ambiguous reads
    bad = compare_records(res, pool, recs, L, amb=bool(amb))
    assert not bad, bad[:10]
    assert (m.stats() == ost).all()
    g = m.counters()
    # event counts that do not depend on how the device walks the index (n_ext does: single-row intervals are
    # finished against the genome instead of by LF steps)
    assert (g["n_hash"], g["n_filter"], g["n_jobs"], g["n_ungapped"]) == \
           (cnt["n_hash"], cnt["n_cand"], cnt["n_sw"], cnt["n_ungapped"])
    assert g["n_ext"] <= cnt["n_ext"]
    m.close()


@pytest.mark.parametrize("name", sorted(golden_args()))
def test_gpu_sam_equals_reference_golden(name, tmp_path):
    """FASTQ -> GPU -> SAM text == the SAM the real reference wrote for the same input"""
    from bitmapperbs_amd import mapper
    fa = str(tmp_path / "genome.fa"); fq = str(tmp_path / "r.fq")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    gunzip_to(os.path.join(GOLD, "se_%s.fq.gz" % name), fq)
    mapper.Index.build(fa, fa, threads=4)
    ix = mapper.Index(fa)
    names, seq, qual = read_fastq(fq)
    L = seq.shape[1]
    m = mapper.Mapper(ix, 0, e_f=e_of(golden_args()[name]))
    res, pool = m.map_se(seq, qual, L)
    text = mapper.sam_header(ix, "") + "".join(mapper.sam_lines_se(ix, names, seq, qual, L, res, pool))
    mine = "".join(l + "\n" for l in text.split("\n")[:-1] if not l.startswith("@PG"))
    assert mine == gzip.open(os.path.join(GOLD, "se_%s.ref.sam.gz" % name), "rt").read()
    from bitmapperbs_amd import distributed
    assert distributed.mapstats_text(m.stats()) == open(os.path.join(GOLD, "se_%s.ref.stats" % name)).read()
    m.close()


@pytest.mark.parametrize("packed", [False, True], ids=["ascii_rows", "packed_rows"])
@pytest.mark.parametrize("L,e", [(120, 0.08), (150, 0.08), (180, 0.08), (250, 0.08), (40, 0.15), (333, 0.05)])
def test_filter_stage_matches_oracle_incl_invalid_sites(env, L, e, packed):
    """K7+K8 through bmbs_filter_batch / bmbs_filter_batch_packed on arbitrary (read, site) pairs, incl. strand ends and wild sites.
    The lengths take the kernel's three forms of the window: held in registers (L + 2k <= 192), streamed with a 32-bit band (180
    bases, k = 14) and streamed with a 64-bit band (250 bases, k = 20).  packed_rows = the form the mapping calls run (rows packed
    on the device by k_pack_rows, bpm_planes<W, true>): clean rows, and dirty ones -- a third of the reads carry an N, three of them
    nothing but N, some an N in their last character (the ragged end of the packed words)"""
    from bitmapperbs_amd import synth, mapper
    r = synth.make_reads_se(env["chroms"], n=3000, L=L, seed=21, sub=0.03, indel=0.004, qual="const", n_rate=0.003)
    r["seq"][5, :] = ord("N"); r["seq"][6, :] = ord("N"); r["seq"][77, :] = ord("N")
    r["seq"][100:160, L - 1] = ord("N"); r["seq"][200:230, 0] = ord("N")
    m = mapper.Mapper(env["ix"], 0, e_f=e)
    k = m.threshold(L)
    G = env["ix"].ref_len
    c, p, minus = r["truth"]
    offs = np.concatenate([[0], np.cumsum([ch.size for ch in env["chroms"]])])[:-1]
    true_site = np.where(minus, 2 * G - (offs[c] + p + L + 8) + 8, offs[c] + p).astype(np.int64) - k
    rng = np.random.default_rng(5)
    read_of, site = [], []
    for i in range(3000):
        for d in (0, int(rng.integers(-k, k + 1)), int(rng.integers(-40, 40))):
            read_of.append(i); site.append(np.uint64(max(0, int(true_site[i]) + d)))
    edge = [0, 1, G - L - 2 * k, G - L - 2 * k + 1, G - 1, G, G + 1, 2 * G - L - 2 * k, 2 * G - L - 2 * k + 1, 2 * G - 1, 2 * G,
            2 * G + 5, (1 << 64) - 3, (1 << 63), G - 5, 2 * G - 40]
    for j, s in enumerate(edge):
        read_of.append(j); site.append(np.uint64(s))
    read_of = np.array(read_of, dtype=np.uint32); site = np.array(site, dtype=np.uint64)
    err, end = m.filter(r["seq"], L, read_of, site, packed=packed)
    for j in range(read_of.size):
        w = env["oix"].window(int(site[j]), L + 2 * k)
        oe, oend = orc.bpm(w, r["seq"][read_of[j]], k)
        assert (int(err[j]), int(end[j])) == (oe, oend), (j, int(site[j]))
    assert (err != 0xFFFFFFFF).sum() > 2000          # the test really exercised accepted candidates
    m.close()


@pytest.mark.parametrize("L,e", [(150, 0.08), (100, 0.08), (165, 0.08), (40, 0.15), (180, 0.08)])
def test_filter_stage_runs_of_one_read_fill_whole_waves(env, L, e):
    """The rows of bpm_planes<u32, true> when every lane of a wave verifies a candidate of the SAME read (the dense list of a
    repeat-rich input: hundreds of candidates per read): the character is a scalar there and the row's match vector one funnel shift of
    a per-letter match plane.  Runs of 1 ... 300 candidates per read laid end to end, so that waves lie wholly inside a run, across
    two runs, and at the ragged end; reads with N (those waves take the checked rows), sites near, at and far from the truth,
    180 bases: the streamed window, which has no such rows."""
    from bitmapperbs_amd import synth, mapper
    r = synth.make_reads_se(env["chroms"], n=400, L=L, seed=77, sub=0.03, indel=0.004, qual="const", n_rate=0.0)
    r["seq"][3, L // 2] = ord("N"); r["seq"][4, :] = ord("N"); r["seq"][9, L - 1] = ord("N")
    m = mapper.Mapper(env["ix"], 0, e_f=e)
    k = m.threshold(L)
    G = env["ix"].ref_len
    c, p, minus = r["truth"]
    offs = np.concatenate([[0], np.cumsum([ch.size for ch in env["chroms"]])])[:-1]
    true_site = np.where(minus, 2 * G - (offs[c] + p + L + 8) + 8, offs[c] + p).astype(np.int64) - k
    rng = np.random.default_rng(11)
    read_of, site = [], []
    for i in range(60):
        run = [64, 128, 300, 63, 65, 1, 200, 130, 64, 257][i % 10]
        other = true_site[rng.integers(0, 400, run)]                  # somebody else's site: a repeat copy that does not match
        d = rng.integers(-k, k + 1, run)
        near = rng.random(run) < 0.5
        st = np.where(near, true_site[i] + d, other + d)
        st[rng.random(run) < 0.02] = 2 * G + 7                        # a wild site inside the run
        read_of += [i] * run; site += [np.uint64(max(0, int(x))) for x in st]
    read_of = np.array(read_of, dtype=np.uint32); site = np.array(site, dtype=np.uint64)
    err, end = m.filter(r["seq"], L, read_of, site, packed=True)
    err_a, end_a = m.filter(r["seq"], L, read_of, site, packed=False)
    assert (err == err_a).all() and (end == end_a).all()
    for j in range(read_of.size):
        w = env["oix"].window(int(site[j]), L + 2 * k)
        oe, oend = orc.bpm(w, r["seq"][read_of[j]], k)
        assert (int(err[j]), int(end[j])) == (oe, oend), (j, int(read_of[j]), int(site[j]))
    assert (err != 0xFFFFFFFF).sum() > 1500
    m.close()


@pytest.mark.parametrize("form,L,e", [("reg2", 150, 0.08), ("reg", 150, 0.08), ("wave", 150, 0.08),          # k = 12: 32 lanes per alignment
                                      ("reg2", 150, 0.04), ("wave", 150, 0.04), ("reg", 150, 0.04),          # k = 6: 16 lanes
                                      ("reg2", 250, 0.08), ("wave", 250, 0.08), ("reg", 250, 0.08),          # k = 20: a whole wave
                                      ("reg2", 100, 0.31), ("wave", 100, 0.31), ("reg2", 61, 0.05), ("wave", 61, 0.05),   # k = 31 (band 63), k = 3
                                      ("reg2", 400, 0.08), ("reg2", 36, 0.1),
                                      ("reg2", 600, 0.08), ("reg", 600, 0.08), ("wave", 600, 0.08), ("reg2", 998, 0.08), ("reg", 998, 0.08), ("wave", 998, 0.02)])
def test_align_stage_matches_oracle(env, monkeypatch, form, L, e):
    """K11-K13 through bmbs_align_batch: jobs = every accepted candidate of the filter stage with err > 0.
    The three forms of the DP kernel: `reg2` = two alignments per lane in packed 16-bit arithmetic (k_align_sw2, the default),
    `reg` = one alignment per lane in 32 bits (k_align_sw: mixed-length batches, extreme penalties), `wave` = one alignment per
    16 / 32 / 64 lanes with the trace in LDS (k_align_sw_wave)"""
    from bitmapperbs_amd import synth, mapper
    monkeypatch.setenv("BMBS_SW", form)
    r = synth.make_reads_se(env["chroms"], n=4000, L=L, seed=31, sub=0.02, indel=0.006, qual="random", n_rate=0.002)
    m = mapper.Mapper(env["ix"], 0, e_f=e)
    k = m.threshold(L)
    G = env["ix"].ref_len
    c, p, minus = r["truth"]
    offs = np.concatenate([[0], np.cumsum([ch.size for ch in env["chroms"]])])[:-1]
    true_site = np.maximum(0, np.where(minus, 2 * G - (offs[c] + p + L + 8) + 8, offs[c] + p).astype(np.int64) - k).astype(np.uint64)
    read_of = np.arange(4000, dtype=np.uint32)
    err, end = m.filter(r["seq"], L, read_of, true_site)
    keep = np.nonzero((err != 0xFFFFFFFF))[0]
    out = m.align(r["seq"], r["qual"], L, read_of[keep], true_site[keep], end[keep], err[keep])
    prm = orc.params(e_f=e)
    n_sw = 0
    for jj, i in enumerate(keep):
        if err[i] == 0:
            continue
        w = env["oix"].window(int(true_site[i]), L + 2 * k)
        o = orc.align(prm, w, r["seq"][i], r["qual"][i], k, int(end[i]), int(err[i]), int(true_site[i] < G))
        n_ops = int(out["n_ops"][jj])
        cg = "%dM" % L if n_ops == 0 else "".join("%d%s" % (int(x) >> 4, "MDISH"[int(x) & 15]) for x in out["ops"][jj][:n_ops])
        assert (int(out["start"][jj]), int(out["end"][jj]), int(out["nm"][jj]), int(out["score"][jj]), cg) == \
               (o["start"], o["end"], o["nm"], o["score"], o["cigar"]), int(i)
        n_sw += n_ops != 0
    assert n_sw > 200
    m.close()


@pytest.mark.parametrize("L,e,sub", [(100, 0.08, 0.01), (150, 0.08, 0.04), (250, 0.08, 0.03), (600, 0.08, 0.02), (998, 0.08, 0.015)])
def test_seed_stage_verdicts_and_vote_sites(env, L, e, sub):
    """K1-K6 + a8-a10 through bmbs_seed_batch: verdicts, and for every general-path read the vote list itself -- sites AND
    counts, in the reference's visiting order (std::sort by vote, unstable) -- against the oracle's"""
    from bitmapperbs_amd import synth, mapper
    r = synth.make_reads_se(env["chroms"], n=20000 if L <= 250 else 4000, L=L, seed=41, sub=sub, indel=0.001, qual="const", n_rate=0.002)
    m = mapper.Mapper(env["ix"], 0, e_f=e)
    s = m.seed(r["seq"], L, vote_cap=4096 * 1024)
    recs, ovs, ovc, ovo = env["oix"].map_se_votes(orc.params(e_f=e), r["seq"], r["qual"], L)
    # oracle path: 1 exit A, 2 exit C, 3 general (incl. "no unique"), 4 exact ambiguous, 0 nothing
    v = s["verdict"].astype(np.int64)
    op = recs["path"].astype(np.int64)
    gen = (op == 3) | ((op == 0) & (recs["n_cand"] > 0))
    assert (v[op == 1] == 1).all() and (v[op == 2] == 2).all() and (v[op == 4] == 4).all()
    assert (v[gen] == 3).all()
    assert (s["n_votes"][gen].astype(np.int64) == recs["n_votes"][gen]).all()
    # the vote lists themselves: same sites, same counts, same (unstable-sort) order
    n_lists = n_long = 0
    for i in np.nonzero(gen)[0]:
        a = int(s["seg_off"][i]); nv = int(s["n_votes"][i])
        oa, ob = int(ovo[i]), int(ovo[i + 1])
        assert ob - oa == nv, i
        assert (s["vote_site"][a:a + nv] == ovs[oa:ob]).all(), i
        assert (s["vote_cnt"][a:a + nv] == ovc[oa:ob]).all(), i
        n_lists += 1; n_long += nv > 16
    assert n_lists > (500 if L <= 250 else 100) and (L < 150 or n_long > 0)
    m.close()


def test_edge_cases_empty_single_allN_short(env):
    from bitmapperbs_amd import mapper
    m = mapper.Mapper(env["ix"], 0)
    res, pool = m.map_se(np.zeros((0, 112), np.uint8), np.zeros((0, 112), np.uint8), 100)
    assert res.size == 0 and pool.size == 0
    for L in (16, 17, 18, 30):
        seq = np.full((3, 32), ord("N"), dtype=np.uint8)
        seq[1, :L] = np.frombuffer(b"ACGT" * 8, dtype=np.uint8)[:L]
        seq[2, :L] = env["chroms"][0][1000:1000 + L]
        q = np.full((3, 32), ord("I"), dtype=np.uint8)
        res, pool = m.map_se(seq, q, L)
        recs, ost, _ = env["oix"].map_se(orc.params(), seq, q, L)
        assert not compare_records(res, pool, recs, L)
    m.close()


def test_full_size_properties_idempotent_and_split_invariant(env):
    """at BASELINE configs[1]'s full batch (10 M x 150 bp, -e 0.04, on the test genome): run-to-run identical bytes, and the
    batch result equals the concatenation of two half-batches (reads are independent units)"""
    import torch
    from bitmapperbs_amd import gpusynth, mapper, capi
    L, stride, n = 150, 160, 10_000_000
    gd, ld = gpusynth.upload_genome(env["chroms"])
    seq, qual = gpusynth.make_reads_se(gd, ld, n, L, stride, seed=99, sub=0.01, indel=0.0005)
    m = mapper.Mapper(env["ix"], 0, e_f=0.04)
    k = m.threshold(L); cap = n * (2 * k + 8)

    def run(a, b):
        res = torch.zeros((b - a, 32), dtype=torch.uint8, device="cuda")
        cig = torch.zeros((cap,), dtype=torch.int32, device="cuda")
        m.reset_stats()
        m.map_se_device(seq[a:b].data_ptr(), qual[a:b].data_ptr(), L, stride, b - a, res.data_ptr(), cig.data_ptr(), cap)
        m.sync()
        return res.cpu().numpy().view(capi.RESULT_DTYPE).reshape(-1), cig.cpu().numpy().view(np.uint32), m.stats()

    r1, c1, s1 = run(0, n)
    r2, c2, s2 = run(0, n)
    assert r1.tobytes() == r2.tobytes() and (s1 == s2).all()
    ra, ca, sa = run(0, n // 2)
    rb, cb, sb = run(n // 2, n)
    assert ((sa + sb) == s1).all() and s1[0] == n
    f = [x for x in capi.RESULT_DTYPE.names if x not in ("cigar_off",)]
    whole = np.concatenate([ra, rb])
    for x in f:
        assert (whole[x] == r1[x]).all(), x
    # cigars of a sample of gapped reads agree too
    idx = np.nonzero(r1["n_cigar"] > 0)[0][:5000]
    for i in idx:
        part, pc = (ra, ca) if i < n // 2 else (rb, cb)
        j = i if i < n // 2 else i - n // 2
        assert mapper.cigar_text(r1[i], c1, L) == mapper.cigar_text(part[j], pc, L)
    # oracle spot-check on a slice of the big batch
    sl = slice(1000, 21000)
    recs, _, _ = env["oix"].map_se(orc.params(e_f=0.04), seq[sl].cpu().numpy(), qual[sl].cpu().numpy(), L)
    assert not compare_records(r1[sl], c1, recs, L)
    m.close()


@pytest.mark.parametrize("mode,L,n,e", [("fast", 150, 5_000_000, 0.08), ("sensitive", 150, 5_000_000, 0.08), ("fast", 250, 1_000_000, 0.08),
                                       ("sensitive", 250, 1_000_000, 0.08), ("fast", 150, 10_000_000, 0.08)])
def test_full_size_properties_paired_end(env, mode, L, n, e):
    """BASELINE configs[2] / [3] launch sizes (5 M pairs of 150 bp, fast and --sensitive; and the 10 M pairs of ONE bench.py launch,
    which the context cuts into a chunk per lane) and configs[4]'s shape (250 bp pairs, -e 0.08: k = 20, 64-bit Myers words, band of
    41): run-to-run identical bytes, split invariance (pairs are independent units), stats add up, and an oracle slice"""
    import torch
    from bitmapperbs_amd import gpusynth, mapper, capi
    stride = (L + 15) // 16 * 16
    gd, ld = gpusynth.upload_genome(env["chroms"])
    s1, q1, s2, q2 = gpusynth.make_reads_pe(gd, ld, n, L, stride, seed=123, sub=0.01, indel=0.0005, qual="random",
                                             ins_hi=400 if L == 150 else 700)
    prm = dict(e_f=e, sensitive=1 if mode == "sensitive" else 0)
    if L == 250:
        prm["max_ins"] = 800
    m = mapper.Mapper(env["ix"], 0, **prm)
    k = m.threshold(L); cap = 2 * n * (2 * k + 8)

    def run(a, b):
        res = torch.zeros((2 * (b - a), 32), dtype=torch.uint8, device="cuda")
        cig = torch.zeros((cap,), dtype=torch.int32, device="cuda")
        m.reset_stats()
        m.map_pe_device(s1[a:b].data_ptr(), q1[a:b].data_ptr(), s2[a:b].data_ptr(), q2[a:b].data_ptr(), L, stride, b - a, res.data_ptr(),
                        cig.data_ptr(), cap)
        m.sync()
        return res.cpu().numpy().view(capi.RESULT_DTYPE).reshape(-1), cig.cpu().numpy().view(np.uint32), m.stats()

    r1, c1, st1 = run(0, n)
    r2, c2, st2 = run(0, n)
    assert r1.tobytes() == r2.tobytes() and (st1 == st2).all()
    ra, ca, sa = run(0, n // 2)
    rb, cb, sb = run(n // 2, n)
    assert ((sa + sb) == st1).all() and st1[0] == n
    whole = np.concatenate([ra, rb])
    for x in [x for x in capi.RESULT_DTYPE.names if x not in ("cigar_off",)]:
        assert (whole[x] == r1[x]).all(), x
    idx = np.nonzero(r1["n_cigar"] > 0)[0][:3000]
    for i in idx:
        part, pc = (ra, ca) if i < n else (rb, cb)        # record 2*pair + mate: the first half holds pairs < n/2
        j = i if i < n else i - n
        assert mapper.cigar_text(r1[i], c1, L) == mapper.cigar_text(part[j], pc, L)
    # oracle spot-check on a slice of the big batch
    a, b = 2000, 12000
    kw = dict(e_f=e, sensitive=1 if mode == "sensitive" else 0)
    if L == 250:
        kw["max_ins"] = 800
    recs, _, _ = env["oix"].map_pe(orc.params(**kw), s1[a:b].cpu().numpy(), q1[a:b].cpu().numpy(), s2[a:b].cpu().numpy(), q2[a:b].cpu().numpy(), L)
    assert not compare_pe(r1[2 * a:2 * b], c1, recs, L)
    m.close()


# ---- paired end (fast mode) ------------------------------------------------------------------------
def compare_pe(res, pool, recs, L, L2=None):
    from bitmapperbs_amd import mapper
    bad = []
    n = recs.size
    La = np.broadcast_to(np.asarray(L), (n,)); Lb = La if L2 is None else np.broadcast_to(np.asarray(L2), (n,))
    for i in range(n):
        a1, a2, b = res[2 * i], res[2 * i + 1], recs[i]
        if int(a1["status"]) != int(b["status"]) or int(a2["status"]) != int(b["status"]):
            bad.append((i, "status", int(a1["status"]), int(b["status"]))); continue
        if int(b["status"]) != 1 and not (int(b["status"]) == 2 and b["cigar1"] != b""):
            continue
        got = (int(a1["flag"]), int(a2["flag"]), int(a1["chrom"]), int(a2["chrom"]), int(a1["pos"]), int(a2["pos"]), int(a1["mapq"]),
               int(a2["mapq"]), int(a1["nm"]), int(a2["nm"]), int(a1["score"]), int(a2["score"]), int(a1["tlen"]),
               mapper.cigar_text(a1, pool, int(La[i])), mapper.cigar_text(a2, pool, int(Lb[i])))
        exp = (int(b["flag1"]), int(b["flag2"]), int(b["chrom1"]), int(b["chrom2"]), int(b["pos1"]), int(b["pos2"]), int(b["mapq"]),
               int(b["mapq"]), int(b["nm1"]), int(b["nm2"]), int(b["score1"]), int(b["score2"]), int(b["tlen"]),
               b["cigar1"].decode(), b["cigar2"].decode())
        if got != exp:
            bad.append((i, got, exp))
    return bad


PE_CASES = [
    dict(n=20000, L=150, seed=1, sub=0.01, indel=0.001, qual="random", prm={}),
    dict(n=20000, L=100, seed=2, sub=0.02, indel=0.002, qual="random", ins_hi=560, prm=dict(e_f=0.04, max_ins=520)),
    dict(n=6000, L=250, seed=3, sub=0.03, indel=0.001, qual="random", ins_hi=700, prm=dict(max_ins=800)),
    dict(n=15000, L=75, seed=4, sub=0.005, indel=0.0, qual="const", ins_lo=60, ins_hi=300, prm=dict(min_ins=100, max_ins=250)),
    dict(n=10000, L=150, seed=5, sub=0.0, indel=0.0, qual="const", conv=0.0, prm={}),     # exact reads, no conversion: exit A/B heavy
    # --sensitive (Map_Pair_Seq_end_to_end): ordered verification, mate-filtered votes, re-seeding rescue
    dict(n=20000, L=100, seed=6, sub=0.06, indel=0.003, qual="random", prm=dict(sensitive=1)),
    dict(n=12000, L=150, seed=7, sub=0.07, indel=0.004, qual="random", ins_hi=450, prm=dict(sensitive=1, e_f=0.1, max_ins=450)),
    dict(n=20000, L=100, seed=8, sub=0.02, indel=0.002, qual="random", ins_hi=560, prm=dict(sensitive=1, e_f=0.04, max_ins=520)),
    dict(n=10000, L=150, seed=9, sub=0.0, indel=0.0, qual="const", conv=0.0, prm=dict(sensitive=1)),
    dict(n=8000, L=75, seed=10, sub=0.08, indel=0.0, qual="const", ins_lo=60, ins_hi=300, prm=dict(sensitive=1, min_ins=100, max_ins=250)),
    # --ambiguous_out for pairs (Schema.cpp:19345 / 21251)
    dict(n=20000, L=75, seed=11, sub=0.01, indel=0.001, qual="random", ins_lo=60, ins_hi=300, prm=dict(ambiguous_out=1, min_ins=100, max_ins=250)),
    dict(n=15000, L=100, seed=12, sub=0.05, indel=0.002, qual="random", prm=dict(sensitive=1, ambiguous_out=1)),
    # mates of 600 / 998 bases
    dict(n=1200, L=600, seed=13, sub=0.02, indel=0.002, qual="random", ins_lo=620, ins_hi=1100, prm=dict(max_ins=1200)),
    dict(n=800, L=998, seed=14, sub=0.015, indel=0.001, qual="random", ins_lo=1020, ins_hi=1500, prm=dict(max_ins=1600)),
    dict(n=1000, L=600, seed=15, sub=0.05, indel=0.002, qual="random", ins_lo=620, ins_hi=1100, prm=dict(sensitive=1, max_ins=1200)),
    dict(n=600, L=998, seed=16, sub=0.04, indel=0.002, qual="random", ins_lo=1020, ins_hi=1500, prm=dict(sensitive=1, max_ins=1600)),
]


@pytest.mark.parametrize("case", PE_CASES, ids=lambda c: "L%d_s%d%s" % (c["L"], c["seed"], "_sens" if c["prm"].get("sensitive") else ""))
def test_map_pe_records_and_stats_match_oracle(case, env):
    from bitmapperbs_amd import synth, mapper
    c = dict(case); prm = c.pop("prm")
    m1, m2 = synth.make_reads_pe(env["chroms"], **c)
    L = c["L"]
    m = mapper.Mapper(env["ix"], 0, **prm)
    res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
    recs, ost, cnt = env["oix"].map_pe(orc.params(**prm), m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
    bad = compare_pe(res, pool, recs, L)
    assert not bad, bad[:5]
    assert (m.stats() == ost).all(), (m.stats(), ost)
    m.close()


def pe_golden_args():
    import json
    return json.load(open(os.path.join(GOLD, "pe_args.json")))


@pytest.mark.parametrize("name", sorted(pe_golden_args()))
def test_gpu_pe_sam_equals_reference_golden(name, tmp_path):
    from bitmapperbs_amd import mapper, distributed
    from test_oracle import pe_params
    fa = str(tmp_path / "genome.fa"); f1 = str(tmp_path / "1.fq"); f2 = str(tmp_path / "2.fq")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    gunzip_to(os.path.join(GOLD, "pe_%s_1.fq.gz" % name), f1)
    gunzip_to(os.path.join(GOLD, "pe_%s_2.fq.gz" % name), f2)
    mapper.Index.build(fa, fa, threads=4)
    ix = mapper.Index(fa)
    n1, s1, q1 = read_fastq(f1); n2, s2, q2 = read_fastq(f2)
    L = s1.shape[1]
    m = mapper.Mapper(ix, 0, **pe_params(pe_golden_args()[name]))
    res, pool = m.map_pe(s1, q1, s2, q2, L)
    text = mapper.sam_header(ix, "") + "".join(mapper.sam_lines_pe(ix, n1, n2, s1, q1, s2, q2, L, res, pool))
    mine = "".join(l + "\n" for l in text.split("\n")[:-1] if not l.startswith("@PG"))
    assert mine == gzip.open(os.path.join(GOLD, "pe_%s.ref.sam.gz" % name), "rt").read()
    assert distributed.mapstats_text(m.stats()) == open(os.path.join(GOLD, "pe_%s.ref.stats" % name)).read()
    m.close()


# ---- the C++ driver: FASTQ -> GPU -> SAM file, whole-program drop-in for `bitmapperBS --search` --------
def _driver():
    from common import ROOT
    p = os.path.join(ROOT, "bitmapperbs_amd", "bmbs_search")
    assert os.path.exists(p), "bmbs_search not built (make -C bitmapperbs_amd/csrc)"
    return p


@pytest.mark.parametrize("name", ["b150", "e75"])
def test_cpp_driver_se_sam_file_equals_reference_golden(name, tmp_path):
    import shutil
    import subprocess
    from bitmapperbs_amd import mapper
    fa = str(tmp_path / "genome.fa"); fq = str(tmp_path / "r.fq"); out = str(tmp_path / "o.sam"); ms = str(tmp_path / "ms.txt")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    # the driver reads gzipped FASTQ directly (Process_Reads.cpp:1455-1514)
    shutil.copy(os.path.join(GOLD, "se_%s.fq.gz" % name), fq + ".gz")
    mapper.Index.build(fa, fa, threads=4)
    p = subprocess.run([_driver(), "--search", fa, "--seq", fq + ".gz", "-o", out, "--mapstats", ms, "--batch", "700"] + golden_args()[name],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    mine = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert mine == gzip.open(os.path.join(GOLD, "se_%s.ref.sam.gz" % name), "rt").read()
    assert open(ms).read() == open(os.path.join(GOLD, "se_%s.ref.stats" % name)).read()


def test_cpp_driver_lower_case_bases(tmp_path):
    """the reference's reader upper-cases the bases (Process_Reads.cpp:860); that happens on the device now (k_fastq_rows) and again
    when SEQ is printed from the FASTQ text: lower-casing a third of the bases must not change a byte of the SAM.  (A quality line
    shorter than its sequence leaves stale buffer bytes in the reference's record, :834-874 -- not a behaviour to pin.)"""
    import subprocess
    from bitmapperbs_amd import mapper
    fa = str(tmp_path / "genome.fa"); fq = str(tmp_path / "r.fq"); out = str(tmp_path / "o.sam")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    mapper.Index.build(fa, fa, threads=4)
    lines = gzip.open(os.path.join(GOLD, "se_b150.fq.gz"), "rt").read().split("\n")
    rng = np.random.default_rng(3)
    for i in range(1, len(lines) - 1, 4):
        s_ = np.frombuffer(lines[i].encode(), dtype=np.uint8).copy()
        m_ = rng.random(s_.size) < 0.33
        s_[m_] |= 0x20
        lines[i] = s_.tobytes().decode()
    open(fq, "w").write("\n".join(lines))
    p = subprocess.run([_driver(), "--search", fa, "--seq", fq, "-o", out, "--batch", "333"] + golden_args()["b150"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    mine = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert mine == gzip.open(os.path.join(GOLD, "se_b150.ref.sam.gz"), "rt").read()


@pytest.mark.parametrize("name", ["p100", "s100"])
def test_cpp_driver_pe_sam_file_equals_reference_golden(name, tmp_path):
    import subprocess
    from bitmapperbs_amd import mapper
    fa = str(tmp_path / "genome.fa"); f1 = str(tmp_path / "1.fq"); f2 = str(tmp_path / "2.fq"); out = str(tmp_path / "o.sam")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    gunzip_to(os.path.join(GOLD, "pe_%s_1.fq.gz" % name), f1)
    gunzip_to(os.path.join(GOLD, "pe_%s_2.fq.gz" % name), f2)
    mapper.Index.build(fa, fa, threads=4)
    p = subprocess.run([_driver(), "--search", fa, "--seq1", f1, "--seq2", f2, "-o", out] + pe_golden_args()[name], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    mine = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert mine == gzip.open(os.path.join(GOLD, "pe_%s.ref.sam.gz" % name), "rt").read()
    stats = "".join(l + "\n" for l in p.stderr.splitlines() if l.startswith("No. of") or l.startswith("Mismatch"))
    assert stats == open(os.path.join(GOLD, "pe_%s.ref.stats" % name)).read()


@pytest.mark.parametrize("kind,name,extra", [
    ("se", "b150", ["--devices", "0,0", "--contexts", "2", "--batch", "97"]),       # two index copies x two contexts: four workers
    ("pe", "p100", ["--devices", "0,0", "--contexts", "1", "--batch", "61"]),
    ("pe", "s100", ["--device", "0", "--contexts", "3", "--batch", "130"]),
])
def test_cpp_driver_several_devices_and_contexts_keep_order_and_stats(kind, name, extra, tmp_path):
    """--devices a,b (one context per listed device, here the same GPU twice) and --contexts S (contexts sharing a device's index):
    batches are dealt to whichever worker is free, the SAM keeps the input order and the five counters are summed over the
    contexts (the reference's N mapping threads + get_mapping_informations, Schema.cpp:26336-26633, 451-476)"""
    import subprocess
    from bitmapperbs_amd import mapper
    fa = str(tmp_path / "genome.fa"); out = str(tmp_path / "o.sam")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    mapper.Index.build(fa, fa, threads=4)
    if kind == "se":
        fq = str(tmp_path / "r.fq")
        gunzip_to(os.path.join(GOLD, "se_%s.fq.gz" % name), fq)
        inp = ["--seq", fq]; args = golden_args()[name]
    else:
        f1 = str(tmp_path / "1.fq"); f2 = str(tmp_path / "2.fq")
        gunzip_to(os.path.join(GOLD, "pe_%s_1.fq.gz" % name), f1)
        gunzip_to(os.path.join(GOLD, "pe_%s_2.fq.gz" % name), f2)
        inp = ["--seq1", f1, "--seq2", f2]; args = pe_golden_args()[name]
    p = subprocess.run([_driver(), "--search", fa] + inp + ["-o", out] + extra + args, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    mine = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert mine == gzip.open(os.path.join(GOLD, "%s_%s.ref.sam.gz" % (kind, name)), "rt").read()
    stats = "".join(l + "\n" for l in p.stderr.splitlines() if l.startswith("No. of") or l.startswith("Mismatch"))
    assert stats == open(os.path.join(GOLD, "%s_%s.ref.stats" % (kind, name))).read()


@pytest.mark.parametrize("name", sorted(__import__("test_oracle").variants()))
def test_cpp_driver_output_variants_equal_reference_golden(name, tmp_path):
    """--unmapped_out / --ambiguous_out / --pbat through the C++ driver (device: bmbs_params.ambiguous_out)"""
    import subprocess
    from bitmapperbs_amd import mapper
    from test_oracle import variants, variant_inputs
    v = variants()[name]
    fa = str(tmp_path / "genome.fa"); out = str(tmp_path / "o.sam")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    mapper.Index.build(fa, fa, threads=4)
    p = subprocess.run([_driver(), "--search", fa] + variant_inputs(v, tmp_path) + ["-o", out, "--batch", "500"] + v["args"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    if v.get("bam"):
        # --bam: same reference dictionary and byte-identical record stream as the reference's (htslib-written) BAM
        from common import bam_payload
        assert bam_payload(out) == bam_payload(os.path.join(GOLD, "var_%s.ref.bam" % name))
        assert open(out, "rb").read()[-28:] == open(os.path.join(GOLD, "var_%s.ref.bam" % name), "rb").read()[-28:]     # BGZF EOF marker
    else:
        mine = "".join(l for l in open(out) if not l.startswith("@PG"))
        assert mine == gzip.open(os.path.join(GOLD, "var_%s.ref.sam.gz" % name), "rt").read()
    stats = "".join(l + "\n" for l in p.stderr.splitlines() if l.startswith("No. of") or l.startswith("Mismatch"))
    assert stats == open(os.path.join(GOLD, "var_%s.ref.stats" % name)).read()


# ---- reads of different lengths in one batch (bmbs_map_se_var / bmbs_map_pe_var) ----------------------------------------
def _trim(rows, lens):
    out = rows.copy()
    cols = np.arange(rows.shape[1])[None, :]
    out[cols >= lens[:, None]] = 0
    return out


@pytest.mark.parametrize("e", [0.08, 0.04, 0.12])
def test_map_se_mixed_lengths_match_oracle(e, env):
    """a trimmed library: every read its own length, threshold, seed count and MAPQ table"""
    from bitmapperbs_amd import synth, mapper
    r = synth.make_reads_se(env["chroms"], n=30000, L=150, seed=31, sub=0.02, indel=0.002, qual="random", n_rate=0.002)
    rng = np.random.default_rng(32)
    lens = rng.integers(20, 151, 30000).astype(np.uint16)
    lens[:2000] = 150; lens[2000:2100] = 17; lens[2100:2200] = 36
    seq, qual = _trim(r["seq"], lens), _trim(r["qual"], lens)
    m = mapper.Mapper(env["ix"], 0, e_f=e)
    res, pool = m.map_se_var(seq, qual, lens)
    recs, ost, cnt = env["oix"].map_se_var(orc.params(e_f=e), seq, qual, lens)
    assert (recs["status"] == 1).sum() > 15000
    bad = compare_records(res, pool, recs, lens)
    assert not bad, bad[:10]
    assert (m.stats() == ost).all(), (m.stats(), ost)
    m.close()


@pytest.mark.parametrize("prm", [dict(), dict(sensitive=1), dict(e_f=0.04, max_ins=520, ambiguous_out=1)], ids=["fast", "sensitive", "e004_amb"])
def test_map_pe_mixed_lengths_match_oracle(prm, env):
    """mates trimmed independently: pairs whose mates have different lengths"""
    from bitmapperbs_amd import synth, mapper
    m1, m2 = synth.make_reads_pe(env["chroms"], n=20000, L=125, seed=33, sub=0.02, indel=0.002, qual="random", ins_hi=480)
    rng = np.random.default_rng(34)
    l1 = rng.integers(30, 126, 20000).astype(np.uint16); l2 = rng.integers(30, 126, 20000).astype(np.uint16)
    l1[:3000] = 125; l2[:3000] = 125
    # mate 2 is trimmed at its 3' end too: the FASTQ record keeps its first l2 characters
    s1, q1, s2, q2 = _trim(m1["seq"], l1), _trim(m1["qual"], l1), _trim(m2["seq"], l2), _trim(m2["qual"], l2)
    m = mapper.Mapper(env["ix"], 0, **prm)
    res, pool = m.map_pe_var(s1, q1, s2, q2, l1, l2)
    recs, ost, cnt = env["oix"].map_pe_var(orc.params(**prm), s1, q1, s2, q2, l1, l2)
    assert (recs["status"] == 1).sum() > 8000
    bad = compare_pe(res, pool, recs, l1, l2)
    assert not bad, bad[:5]
    assert (m.stats() == ost).all(), (m.stats(), ost)
    m.close()


# ---- packed reads: 2 bits per base + an 'N' plane over the link (bmbs_map_*_packed) -----------------------------------------------------
def _sprinkle_n(rng, seq, lens, rate=0.003):
    out = seq.copy()
    hit = rng.random(out.shape) < rate
    hit &= np.arange(out.shape[1])[None, :] < (lens[:, None] if lens is not None else out.shape[1])
    hit[::7] = False                                   # most reads stay clean: both kinds of piece in one batch
    out[hit] = ord("N")
    return out


@pytest.mark.parametrize("case", ["se150", "se_mixed", "se998", "pe100_fast", "pe150_sensitive", "pe_mixed", "pe_chunks"])
def test_packed_reads_match_oracle(case, env, monkeypatch):
    """bmbs_map_se_packed / bmbs_map_pe_packed: the caller's 2-bit rows (bmbs_pack_rows) -- mate 2 in FASTQ orientation, reverse-
    complemented on the device on the packed words; 'N' in a plane of its own, its text rebuilt only where a kernel asks for it --
    give the oracle's records, CIGARs and mapstats, like the ASCII calls: uniform and mixed lengths, 998 bases (32 words per row),
    fast and --sensitive pairs, and a call cut into chunks over three lanes"""
    from bitmapperbs_amd import synth, mapper
    M = mapper.Mapper
    rng = np.random.default_rng(5)
    if case.startswith("se"):
        L = 998 if case == "se998" else 150
        n = 1500 if case == "se998" else 20000
        r = synth.make_reads_se(env["chroms"], n=n, L=L, seed=51, sub=0.02, indel=0.002, qual="random", n_rate=0.002)
        lens = None
        seq, qual = r["seq"], r["qual"]
        if case == "se_mixed":
            lens = rng.integers(20, L + 1, n).astype(np.uint16); lens[:500] = L
            seq, qual = _trim(seq, lens), _trim(qual, lens)
        m = M(env["ix"], 0, e_f=0.08)
        rows = M.pack_rows(seq, L, lens)
        assert rows.shape[1] == (L + 31) // 32 + (L + 63) // 64
        res, pool = m.map_se_packed(rows, qual, L, lens)
        if lens is None:
            recs, ost, _ = env["oix"].map_se(orc.params(e_f=0.08), seq, qual, L)
        else:
            recs, ost, _ = env["oix"].map_se_var(orc.params(e_f=0.08), seq, qual, lens)
        assert (recs["status"] == 1).sum() > n // 2
        bad = compare_records(res, pool, recs, L if lens is None else lens)
        assert not bad, bad[:5]
        assert (m.stats() == ost).all(), (m.stats(), ost)
        m.close()
        return
    prm = dict(sensitive=1) if case == "pe150_sensitive" else dict()
    L = 150 if case in ("pe150_sensitive", "pe_chunks") else 100 if case == "pe100_fast" else 125
    n = 12000
    if case == "pe_chunks":
        monkeypatch.setenv("BMBS_LANES", "3"); monkeypatch.setenv("BMBS_SPLIT_MIN", "1000")
    m1, m2 = synth.make_reads_pe(env["chroms"], n=n, L=L, seed=52, sub=0.02, indel=0.002, qual="random", ins_hi=480)
    l1 = l2 = None
    s1, q1, s2, q2 = m1["seq"], m1["qual"], m2["seq"], m2["qual"]
    if case == "pe_mixed":
        l1 = rng.integers(30, L + 1, n).astype(np.uint16); l2 = rng.integers(30, L + 1, n).astype(np.uint16)
        l1[:2000] = L; l2[:2000] = L
        s1, q1, s2, q2 = _trim(s1, l1), _trim(q1, l1), _trim(s2, l2), _trim(q2, l2)
    s1 = _sprinkle_n(rng, s1, l1); s2 = _sprinkle_n(rng, s2, l2)
    m = M(env["ix"], 0, **prm)
    r1 = M.pack_rows(s1, L, l1, pwords=12 if L <= 150 else None); r2 = M.pack_rows(s2, L, l2, pwords=12 if L <= 150 else None)      # (rows may be further apart than they need)
    res, pool = m.map_pe_packed(r1, r2, q1, q2, L, l1, l2)
    if l1 is None:
        recs, ost, _ = env["oix"].map_pe(orc.params(**prm), s1, q1, s2, q2, L)
        bad = compare_pe(res, pool, recs, L, L)
    else:
        recs, ost, _ = env["oix"].map_pe_var(orc.params(**prm), s1, q1, s2, q2, l1, l2)
        bad = compare_pe(res, pool, recs, l1, l2)
    assert (recs["status"] == 1).sum() > n // 2
    assert not bad, bad[:5]
    assert (m.stats() == ost).all(), (m.stats(), ost)
    # ... and exactly what the ASCII call returns for the same rows
    m.reset_stats()
    res2, pool2 = (m.map_pe(s1, q1, s2, q2, L) if l1 is None else m.map_pe_var(s1, q1, s2, q2, l1, l2))
    for f in ("status", "chrom", "pos", "flag", "mapq", "nm", "score", "n_cigar", "tlen"):
        assert (res[f] == res2[f]).all(), f
    m.close()


@pytest.mark.parametrize("super_shift", [None, "20"])
def test_wide_index_forms_match_oracle(env, monkeypatch, super_shift):
    """texts of 2^32 symbols and more (GRCh38): 64-bit suffix array + Occ counts relative to super-blocks whose sums travel in
    the index descriptor, forced on the test genome with BMBS_WIDE=1 (read by bmbs_index_attach).  The default super-block of 2^31
    symbols leaves this 3 M-symbol text with one; BMBS_SUPER_SHIFT=20 gives it three, as GRCh38 has at 2^31."""
    from bitmapperbs_amd import synth, mapper
    monkeypatch.setenv("BMBS_WIDE", "1")
    if super_shift:
        monkeypatch.setenv("BMBS_SUPER_SHIFT", super_shift)
    r = synth.make_reads_se(env["chroms"], n=20000, L=120, seed=41, sub=0.02, indel=0.002, qual="random", n_rate=0.002)
    m = mapper.Mapper(env["ix"], 0, e_f=0.08)
    res, pool = m.map_se(r["seq"], r["qual"], 120)
    recs, ost, cnt = env["oix"].map_se(orc.params(e_f=0.08), r["seq"], r["qual"], 120)
    assert not compare_records(res, pool, recs, 120)
    assert (m.stats() == ost).all()
    m.close()
    m1, m2 = synth.make_reads_pe(env["chroms"], n=8000, L=100, seed=42, sub=0.03, indel=0.002, qual="random")
    m = mapper.Mapper(env["ix"], 0, sensitive=1)
    res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
    recs, ost, cnt = env["oix"].map_pe(orc.params(sensitive=1), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
    assert not compare_pe(res, pool, recs, 100)
    assert (m.stats() == ost).all()
    m.close()


@pytest.mark.parametrize("wide", ["0", "1"])
def test_locate_and_window_stages_match_oracle(wide, env, monkeypatch):
    """K5 (SA[row]) and K7 (genome window) through their stage entry points, against the oracle's reading of the index files;
    also in the wide (>= 2^32 symbols) device forms"""
    from bitmapperbs_amd import mapper
    monkeypatch.setenv("BMBS_WIDE", wide)
    m = mapper.Mapper(env["ix"], 0)
    G = env["ix"].ref_len
    rng = np.random.default_rng(11)
    rows = np.concatenate([rng.integers(0, 2 * G + 1, 20000), [0, 1, 2 * G - 1, 2 * G]]).astype(np.uint64)
    got = m.locate(rows)
    oix = env["oix"]
    exp = np.array([oix.L.orc_sa_at(oix.h, int(r)) for r in rows], dtype=np.uint64)
    assert (got == exp).all()
    sites = np.concatenate([rng.integers(0, 2 * G, 5000), [0, G - 130, G - 129, G - 1, G, 2 * G - 130, 2 * G - 129, 2 * G + 7]]).astype(np.uint64)
    W = m.windows(sites, 130)
    for j in range(sites.size):
        w = oix.window(int(sites[j]), 130)
        assert bytes(W[j]) == bytes(w[:130]), int(sites[j])
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", [0, 1], ids=["wave", "block"])
def test_vote_order_kernels_give_std_sort_permutation(form, env, tmp_path):
    """a9: the parallel vote sort of k_vote_long (block-wide partition passes + per-lane short ranges + stable final pass)
    visits every list in exactly the order libstdc++'s std::sort does -- random, structured and adversarial lists"""
    import ctypes as C
    import subprocess
    from bitmapperbs_amd import mapper
    so = str(tmp_path / "std_order.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "csrc", "std_order.cpp")], check=True)
    ref = C.CDLL(so)
    cap = 4096 if form else 256
    rng = np.random.default_rng(900 + form)
    lists = []
    for c in range(1500 if form else 6000):
        if c % 7 == 0:
            n = int(rng.integers(cap // 2, cap + 1))
        elif c % 3 == 0:
            n = int(rng.integers(17, min(cap, 600) + 1))
        else:
            n = int(rng.integers(1, 48))
        maxv = (2, 6, 25, 255)[c % 4]
        kind = c % 8
        i = np.arange(n)
        if kind <= 2:
            v = 1 + rng.integers(0, maxv, n)
        elif kind == 3:
            v = 1 + (i * maxv) // n
        elif kind == 4:
            v = maxv - (i * maxv) // n
        elif kind == 5:
            v = 1 + np.minimum(i, n - i) % maxv
        elif kind == 6:
            v = np.ones(n, dtype=np.int64)
        else:
            v = 1 + np.where(rng.random(n) < 0.9, 0, rng.integers(0, maxv, n))
        lists.append(v.astype(np.uint8))
    for n in (17, 33, 64, 120, 129, 200, 255):          # depth-limit / heapsort branch (serial fallback or per-lane heapsort)
        kv = np.zeros(n, dtype=np.uint8)
        ref.killer_votes(n, kv.ctypes.data_as(C.c_void_p))
        lists.append(kv)
        if form:                                        # an adversarial stretch inside a long list of larger votes... and of equal ones
            lists.append(np.concatenate([kv, np.full(3000, 1, np.uint8)]))
            lists.append(np.concatenate([np.full(500, 255, np.uint8), kv, rng.integers(1, 3, 2000).astype(np.uint8)]))
    seg = np.zeros(len(lists) + 1, dtype=np.int64)
    seg[1:] = np.cumsum([x.size for x in lists])
    vote = np.concatenate(lists)
    want = np.zeros(vote.size, dtype=np.uint32)
    ref.std_order(vote.ctypes.data_as(C.c_void_p), seg.ctypes.data_as(C.c_void_p), C.c_int64(len(lists)), want.ctypes.data_as(C.c_void_p))
    m = mapper.Mapper(env["ix"], 0)
    got = m.vote_order(vote, seg, form)
    m.close()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first difference in list %d (length %d)" % (np.searchsorted(seg, bad[0], "right") - 1,
                                                                        lists[np.searchsorted(seg, bad[0], "right") - 1].size)


@pytest.mark.gpu
@pytest.mark.parametrize("legacy", [0, 1])
def test_seed_extra_without_lds_rows_matches_oracle(env, monkeypatch, legacy):
    """k_seed_extra without the rows in LDS: the form for reads too long to stage 64 rows there (packed rows: beyond 16 KB per
    wave, i.e. reads of more than ~650 bases; ASCII rows under BMBS_LEGACY=1: beyond 48 KB)"""
    from bitmapperbs_amd import synth, mapper
    if legacy:
        monkeypatch.setenv("BMBS_LEGACY", "1")
    r = synth.make_reads_se(env["chroms"], n=3000, L=800, seed=77, sub=0.04, indel=0.003, qual="random", n_rate=0.003)
    m = mapper.Mapper(env["ix"], 0, e_f=0.08)
    res, pool = m.map_se(r["seq"], r["qual"], 800)
    recs, ost, cnt = env["oix"].map_se(orc.params(e_f=0.08), r["seq"], r["qual"], 800)
    assert not compare_records(res, pool, recs, 800)
    assert (m.stats() == ost).all()
    m.close()


@pytest.mark.gpu
def test_without_the_20mer_table_matches_oracle(env, monkeypatch):
    """BMBS_T20=0 (what bmbs_index_attach also falls back to when the device lacks 76 GB of free memory): the 16-mer lookup +
    backward extensions do all the seeding; single-end and --sensitive paired-end"""
    from bitmapperbs_amd import synth, mapper
    monkeypatch.setenv("BMBS_T20", "0")
    r = synth.make_reads_se(env["chroms"], n=20000, L=150, seed=91, sub=0.03, indel=0.002, qual="random", n_rate=0.003)
    m = mapper.Mapper(env["ix"], 0, e_f=0.08)
    res, pool = m.map_se(r["seq"], r["qual"], 150)
    recs, ost, cnt = env["oix"].map_se(orc.params(e_f=0.08), r["seq"], r["qual"], 150)
    assert not compare_records(res, pool, recs, 150)
    assert (m.stats() == ost).all()
    m.close()
    m1, m2 = synth.make_reads_pe(env["chroms"], n=8000, L=100, seed=92, sub=0.03, indel=0.002, qual="random")
    m = mapper.Mapper(env["ix"], 0, sensitive=1)
    res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
    recs, ost, cnt = env["oix"].map_pe(orc.params(sensitive=1), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
    assert not compare_pe(res, pool, recs, 100)
    assert (m.stats() == ost).all()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["se", "pe", "pe_sensitive"])
@pytest.mark.parametrize("knob", ["BMBS_LEGACY=1", "BMBS_SW=wave", "BMBS_SW=reg", "BMBS_KGRAM=2", "BMBS_KGRAM=0", "BMBS_T20=0", "BMBS_TDEPTH=21", "BMBS_WIDE=1",
                                  "BMBS_LANES=1", "BMBS_EXACT=1", "BMBS_SEED_WAVES=4096", "BMBS_SCAN_CHAIN=1"])
def test_ab_switches_give_identical_records(knob, mode, env, monkeypatch):
    """every documented switch (bmbs_api.hip: struct Knobs) maps exactly like the default forms: single-end,
    paired-end and --sensitive, reads with letters outside ACGT, lengths that are not a multiple of 16"""
    from bitmapperbs_amd import synth, mapper
    name, val = knob.split("=")
    monkeypatch.setenv(name, val)
    if name == "BMBS_LANES":
        monkeypatch.setenv("BMBS_SPLIT_MIN", "2000")
    L = 251 if name == "BMBS_SW" else 150
    sens = 1 if mode == "pe_sensitive" else 0
    m = mapper.Mapper(env["ix"], 0, e_f=0.08, sensitive=sens)
    if mode == "se":
        r = synth.make_reads_se(env["chroms"], n=12000, L=L, seed=300 + len(knob), sub=0.03, indel=0.002, qual="random", n_rate=0.003)
        res, pool = m.map_se(r["seq"], r["qual"], L)
        recs, ost, cnt = env["oix"].map_se(orc.params(e_f=0.08), r["seq"], r["qual"], L)
        assert not compare_records(res, pool, recs, L)
    else:
        m1, m2 = synth.make_reads_pe(env["chroms"], n=8000, L=L, seed=400 + len(knob), sub=0.05 if sens else 0.02, indel=0.002, qual="random", ins_hi=max(500, 2 * L + 50))
        rng = np.random.default_rng(len(knob))
        for mm in (m1, m2):
            pos = rng.random(mm["seq"].shape) < 0.003
            mm["seq"][pos] = np.frombuffer(b"NNNRY", dtype=np.uint8)[rng.integers(0, 5, int(pos.sum()))]
        prm = dict(max_ins=max(500, 2 * L + 50))
        m.close()
        m = mapper.Mapper(env["ix"], 0, e_f=0.08, sensitive=sens, **prm)
        res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
        recs, ost, _ = env["oix"].map_pe(orc.params(e_f=0.08, sensitive=sens, **prm), m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
        assert not compare_pe(res, pool, recs, L)
    assert (m.stats() == ost).all()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["se", "pe", "pe_sensitive"])
def test_three_letter_index_steps_change_nothing_but_the_gathers(tmp_path, monkeypatch, mode):
    """the trigram rank table (DevIndex::occ3: three backward extensions per gather pair while an interval shrinks slowly) on a
    genome whose reads walk the index for most of their length -- families of 40 .. 400 near-identical copies: records and
    statistics equal the oracle's, the event counters (16-mer lookups, extensions -- the reference's events) equal those of
    BMBS_KGRAM=0, and the steps were really taken"""
    from bitmapperbs_amd import synth, mapper
    names, chroms = synth.make_genome(1_200_000, 2, seed=91)
    rng = np.random.default_rng(92)
    for (elen, copies, div) in [(900, 400, 0.01), (2500, 60, 0.004), (400, 150, 0.03), (3000, 40, 0.0)]:
        el = synth._ACGT[rng.integers(0, 4, elen)]
        for _ in range(copies):
            ch = chroms[rng.integers(0, len(chroms))]
            p = int(rng.integers(0, ch.size - elen))
            e = el.copy(); mm = rng.random(elen) < div
            e[mm] = synth._ACGT[rng.integers(0, 4, int(mm.sum()))]
            if rng.random() < 0.5: e = synth.revcomp(e)
            ch[p:p + elen] = e
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, chroms)
    mapper.Index.build(fa, fa, threads=8)
    ix = mapper.Index(fa); oix = orc.OrcIndex(fa)
    out = {}
    for kg in ("1", "0"):
        monkeypatch.setenv("BMBS_KGRAM", "2" if kg == "1" else "0")       # 2: the trigram kernels from the first call on (1 = once long chains were seen)
        sens = 1 if mode == "pe_sensitive" else 0
        m = mapper.Mapper(ix, 0, sensitive=sens)
        if mode == "se":
            r = synth.make_reads_se(chroms, n=20000, L=150, seed=5, sub=0.01, indel=0.001, qual="random", n_rate=0.001)
            res, pool = m.map_se(r["seq"], r["qual"], 150)
            if kg == "1":
                recs, ost, _ = oix.map_se(orc.params(), r["seq"], r["qual"], 150)
                assert not compare_records(res, pool, recs, 150)
        else:
            m1, m2 = synth.make_reads_pe(chroms, n=12000, L=150, seed=6, sub=0.04 if sens else 0.01, indel=0.001, qual="random")
            res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
            if kg == "1":
                recs, ost, _ = oix.map_pe(orc.params(sensitive=sens), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
                assert not compare_pe(res, pool, recs, 150)
        if kg == "1":
            assert (m.stats() == ost).all()
        out[kg] = (res.tobytes(), pool.tobytes(), m.stats().tolist(), m.counters())
        m.close()
    a, b = out["1"], out["0"]
    assert a[:3] == b[:3]
    assert (a[3]["n_hash"], a[3]["n_ext"], a[3]["n_sa"]) == (b[3]["n_hash"], b[3]["n_ext"], b[3]["n_sa"])
    assert b[3]["n_jump"] == 0 and a[3]["n_jump"] > 1000, a[3]
    assert 3 * a[3]["n_jump"] < a[3]["n_ext"]


@pytest.mark.gpu
@pytest.mark.parametrize("sensitive,rows", [(0, "packed"), (1, "packed"), (0, "ascii"), (1, "ascii")])     # ascii = BMBS_LEGACY=1
def test_depth21_outcome_table_paired_end(env, monkeypatch, sensitive, rows):
    """the 21-mer form of the outcome table (3^21 entries, 84 GB: what a GRCh38-size index gets) forced on the test genome: the
    seeding engine reads five letters ahead instead of four; pairs in both modes against the oracle, with the seeding kernels on
    the packed rows k_pe_prepare writes (the paired-end default) and on the ASCII rows"""
    from bitmapperbs_amd import synth, mapper
    monkeypatch.setenv("BMBS_TDEPTH", "21")
    if rows == "ascii":
        monkeypatch.setenv("BMBS_LEGACY", "1")
    m1, m2 = synth.make_reads_pe(env["chroms"], n=12000, L=150, seed=77 + sensitive, sub=0.03, indel=0.002, qual="random")
    # a few characters outside ACGT, 'N' and others, so that the not-ACGT plane of the packed rows is exercised by pairs too
    rng = np.random.default_rng(5)
    for mm in (m1, m2):
        pos = rng.random(mm["seq"].shape) < 0.003
        mm["seq"][pos] = np.frombuffer(b"NNNR", dtype=np.uint8)[rng.integers(0, 4, int(pos.sum()))]
    m = mapper.Mapper(env["ix"], 0, sensitive=sensitive)
    res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
    recs, ost, _ = env["oix"].map_pe(orc.params(sensitive=sensitive), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
    assert not compare_pe(res, pool, recs, 150)
    assert (m.stats() == ost).all()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sensitive", [0, 1])
@pytest.mark.parametrize("form", ["sparse", "full"])
@pytest.mark.parametrize("L", [40, 101, 150, 251])
def test_pe_prepare_forms_on_odd_lengths(env, monkeypatch, L, form, sensitive):
    """k_pe_prepare_p (the default: mate 2 reverse-complemented on the packed words, ASCII text kept only for the 16-byte pieces
    that hold a character outside ACGT) against the byte-wise form (BMBS_LEGACY=1): fixed lengths that are and are not
    multiples of 16 / 32 / 64, then mates trimmed independently, with 'N' and other letters in both mates; fast mode and
    --sensitive (whose re-seeding, k_pes_reseed, searches the packed rows too)"""
    from bitmapperbs_amd import synth, mapper
    if form == "full":
        monkeypatch.setenv("BMBS_LEGACY", "1")
    m1, m2 = synth.make_reads_pe(env["chroms"], n=6000, L=L, seed=900 + L, sub=0.05 if sensitive else 0.02, indel=0.002, qual="random",
                                 ins_hi=max(500, 2 * L + 50))
    rng = np.random.default_rng(L)
    for mm in (m1, m2):
        pos = rng.random(mm["seq"].shape) < 0.004
        mm["seq"][pos] = np.frombuffer(b"NNNRY", dtype=np.uint8)[rng.integers(0, 5, int(pos.sum()))]
    m = mapper.Mapper(env["ix"], 0, sensitive=sensitive)
    res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
    recs, ost, _ = env["oix"].map_pe(orc.params(sensitive=sensitive), m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
    assert (recs["status"] == 1).sum() > 2000
    assert not compare_pe(res, pool, recs, L)
    assert (m.stats() == ost).all()
    l1 = rng.integers(max(20, L // 3), L + 1, 6000).astype(np.uint16); l2 = rng.integers(max(20, L // 3), L + 1, 6000).astype(np.uint16)
    l1[:500] = L; l2[:500] = L
    s1, q1, s2, q2 = _trim(m1["seq"], l1), _trim(m1["qual"], l1), _trim(m2["seq"], l2), _trim(m2["qual"], l2)
    m.reset_stats() if hasattr(m, "reset_stats") else None
    res, pool = m.map_pe_var(s1, q1, s2, q2, l1, l2)
    recs, _, _ = env["oix"].map_pe_var(orc.params(sensitive=sensitive), s1, q1, s2, q2, l1, l2)
    bad = compare_pe(res, pool, recs, l1, l2)
    assert not bad, bad[:5]
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(10))
def test_fuzzed_parameters_match_oracle(env, k):
    """ten fixed draws of tools/fuzz_parity.py's random trials (read length 20-300, threshold, scoring parameters, insert bounds,
    error rates, letters outside ACGT, SE / PE fast / PE --sensitive, fixed and mixed lengths, --ambiguous_out): records and stats
    equal the oracle's.  The tool itself ran 60 + 400 such trials on the MI355X box in round 2 without a difference
    (profiles/r02_fuzz_parity.txt)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    rng = np.random.default_rng(4242)
    t = None
    for _ in range(k + 1):
        t = fz.draw(rng)
    bad, mapped = fz.run_trial(t, env)
    assert not bad, (t, bad[:3])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["se", "pe", "pes"])
def test_cheap_gaps_many_cigar_operations(env, mode):
    """--gap_open 1 --gap_extension 3 with mismatches at 8-9: gaps are cheaper than mismatches and the DP returns alignments with
    more than 2k + 8 CIGAR operations (found by tools/fuzz_parity.py, seed 2026 trial 136: the pool had 2k + 8 slots per read
    whatever the penalties, and the overflowing records printed garbage).  bmbs_max_cigar_ops sizes the slots from the penalties."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    prm = dict(e_f=0.08, mp_max=9, mp_min=8, np=1, gap_open=1, gap_ext=3, ambiguous_out=1)
    t = dict(mode=mode, L=116, prm=prm, n=6702, seed=160399697, sub=0.05, indel=0.0005, qual="const", conv=0.0, n_rate=0.003, mixed=mode != "se")
    if mode != "se":
        prm.update(min_ins=120, max_ins=800, sensitive=1 if mode == "pes" else 0)
        t["ins_hi"] = 780
    from bitmapperbs_amd import mapper
    m = mapper.Mapper(env["ix"], 0, **prm)
    assert m.max_cigar_ops(116) > 2 * m.threshold(116) + 8
    m.close()
    bad, mapped = fz.run_trial(t, env)
    assert mapped > 1000 and not bad, bad[:3]


@pytest.mark.gpu
def test_fuzzed_files_and_options_equal_the_reference_binary(tmp_path):
    """eight fixed draws of tools/fuzz_e2e.py: random FASTQ files and --search options through bmbs_search (GPU) and through the
    real reference binary (oracle/_ref/bitmapperBS, -t 1) -- same SAM body, same --mapstats.  Independent of the oracle.  The
    tool ran 60 + 400 such trials on the MI355X box in round 2 (profiles/r02_fuzz_e2e.txt)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_e2e", os.path.join(ROOT, "tools", "fuzz_e2e.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    if not os.path.exists(fz.REF):
        pytest.skip("oracle/_ref/bitmapperBS has not travelled to this box")
    env = fz.make_env(str(tmp_path))
    rng = np.random.default_rng(99)
    for i in range(8):
        t = fz.draw(rng)
        bad, lines = fz.run_trial(t, env, str(tmp_path))
        assert not bad, (t, bad[:2])


@pytest.mark.gpu
def test_shared_index_contexts_map_concurrently(env):
    """bmbs_index_share: three contexts on one attached index, driven by three host threads at once, give the oracle's records"""
    from concurrent.futures import ThreadPoolExecutor
    from bitmapperbs_amd import synth, mapper
    owner = mapper.Mapper(env["ix"], 0, e_f=0.08)
    ms = [owner] + [mapper.Mapper(env["ix"], 0, share=owner, e_f=0.08) for _ in range(2)]
    batches = [synth.make_reads_se(env["chroms"], n=15000, L=150, seed=500 + i, sub=0.03, indel=0.002, qual="random", n_rate=0.002) for i in range(3)]

    def run(i):
        out = None
        for _ in range(3):                      # several calls per context while the others are busy too
            out = ms[i].map_se(batches[i]["seq"], batches[i]["qual"], 150)
        return out
    with ThreadPoolExecutor(3) as ex:
        outs = list(ex.map(run, range(3)))
    for i in range(3):
        recs, ost, cnt = env["oix"].map_se(orc.params(e_f=0.08), batches[i]["seq"], batches[i]["qual"], 150)
        assert not compare_records(outs[i][0], outs[i][1], recs, 150)
    for mm in ms[1:]:
        mm.close()
    owner.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_contigs", [500, 1500, 40000])
def test_many_contigs_match_oracle(tmp_path, n_contigs):
    """an assembly of 500 / 1500 / 40 000 short sequences (a scaffold-level assembly: bmbs_result.chrom is 32 bits since round 3):
    placement by binary search over the contig starts (SE and PE); the finalize kernels keep the table in LDS up to 1024 sequences
    and read it from global memory beyond"""
    from bitmapperbs_amd import synth, mapper
    names, chroms = synth.make_genome(1_200_000 * n_contigs // 500, n_contigs, seed=611)
    fa = str(tmp_path / "contigs.fa")
    synth.write_fasta(fa, names, chroms)
    if n_contigs > 5000:
        mapper.Index.build(fa, fa, threads=8, device=0)
    else:
        mapper.Index.build(fa, fa, threads=8)
    ix, oix = mapper.Index(fa), orc.OrcIndex(fa)
    r = synth.make_reads_se(chroms, n=30000, L=100, seed=612, sub=0.02, indel=0.002, qual="random", n_rate=0.002)
    m = mapper.Mapper(ix, 0, e_f=0.08)
    res, pool = m.map_se(r["seq"], r["qual"], 100)
    recs, ost, cnt = oix.map_se(orc.params(e_f=0.08), r["seq"], r["qual"], 100)
    assert not compare_records(res, pool, recs, 100)
    assert (m.stats() == ost).all()
    assert len(set(int(x) for x in res["chrom"][res["status"] == 1])) > 0.4 * min(n_contigs, 30000)          # the reads really land on many contigs
    if n_contigs > 32767:
        assert int(res["chrom"][res["status"] == 1].max()) > 32767
    m.close()
    m1, m2 = synth.make_reads_pe(chroms, n=10000, L=100, seed=613, sub=0.02, indel=0.002, qual="random")
    m = mapper.Mapper(ix, 0)
    res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
    recs, ost, cnt = oix.map_pe(orc.params(), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
    assert not compare_pe(res, pool, recs, 100)
    assert (m.stats() == ost).all()
    m.close()


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """the N > 1 path of bench.py end to end on a 1-GPU box: `--gpus 2` starts two ranks itself, rank 0 builds the index while the
    other waits at the barrier, each rank attaches its own copy and maps its own reads, the mapstats are all-reduced (gloo here:
    RCCL refuses two ranks on one device) and rank 0 prints ONE line whose totals are those of both ranks"""
    import json
    import subprocess
    import sys
    from common import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "1", "--units", "500000", "--steps", "2", "--warmup", "1",
                        "--min-seconds", "0", "--no-cpu", "--no-secondary", "--dist-backend", "gloo", "--same-device", "--workdir", "/tmp/bmbs_bench_t2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["mapstats"]["reads_or_pairs"] == 2 * 2 * 500000          # two ranks x two steps x 500 k reads
    assert d["value"] > 0 and d["config"]["reads_per_gpu_per_step"] == 500000


# ---- round 3: lanes, launches that do not wait for their stage counts, the repeat with exact sizes ---------------------------------
def _se_reads(env, seed, sub, n=24000, L=150):
    from bitmapperbs_amd import synth
    return synth.make_reads_se(env["chroms"], n=n, L=L, seed=seed, sub=sub, indel=0.002, qual="random", n_rate=0.002)


@pytest.mark.parametrize("lanes,chunk", [(1, 0), (2, 0), (3, 5000)])
def test_split_calls_and_async_launch_sequence_se(env, monkeypatch, lanes, chunk):
    """a context's later calls do not wait for their stage counts (capacities learned from the first call, guards on the device) and
    a call is cut into chunks dealt to the lanes: same records, same statistics as the oracle, whatever the split"""
    from bitmapperbs_amd import mapper
    monkeypatch.setenv("BMBS_LANES", str(lanes))
    monkeypatch.setenv("BMBS_SPLIT_MIN", "2000")
    if chunk:
        monkeypatch.setenv("BMBS_CHUNK", str(chunk))
    m = mapper.Mapper(env["ix"], 0, e_f=0.08)
    tot = np.zeros(5, dtype=np.int64)
    for rep, (seed, sub) in enumerate([(501, 0.02), (502, 0.02), (503, 0.03)]):
        r = _se_reads(env, seed, sub)
        res, pool = m.map_se(r["seq"], r["qual"], 150)
        recs, ost, _ = env["oix"].map_se(orc.params(e_f=0.08), r["seq"], r["qual"], 150)
        bad = compare_records(res, pool, recs, 150)
        assert not bad, (rep, bad[:10])
        tot += ost
        assert (m.stats() == tot).all(), rep
    m.close()


def test_capacity_overflow_repeats_the_call_with_exact_sizes(env, monkeypatch):
    """BMBS_CAP_SCALE shrinks the learned capacities so that the candidate and DP-job counts of the second and third call do not fit:
    the guards take the work away, nothing is committed, and the call is issued again with exact sizes -- records and statistics
    as if nothing had happened (single end, pairs, --sensitive with its re-seeding stage)"""
    from bitmapperbs_amd import synth, mapper
    monkeypatch.setenv("BMBS_CAP_SCALE", "0.02")
    monkeypatch.setenv("BMBS_LANES", "2")
    monkeypatch.setenv("BMBS_SPLIT_MIN", "3000")
    m = mapper.Mapper(env["ix"], 0, e_f=0.08)
    tot = np.zeros(5, dtype=np.int64)
    for seed, sub in [(601, 0.005), (602, 0.05), (603, 0.05)]:
        r = _se_reads(env, seed, sub)
        res, pool = m.map_se(r["seq"], r["qual"], 150)
        recs, ost, _ = env["oix"].map_se(orc.params(e_f=0.08), r["seq"], r["qual"], 150)
        assert not compare_records(res, pool, recs, 150)
        tot += ost
        assert (m.stats() == tot).all()
    assert m.retries() > 0
    m.close()
    for sensitive in (0, 1):
        m = mapper.Mapper(env["ix"], 0, sensitive=sensitive)
        tot = np.zeros(5, dtype=np.int64)
        for seed, sub in [(611, 0.005), (612, 0.05), (613, 0.06)]:
            m1, m2 = synth.make_reads_pe(env["chroms"], n=9000, L=100, seed=seed + sensitive, sub=sub, indel=0.002, qual="random")
            res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
            recs, ost, _ = env["oix"].map_pe(orc.params(sensitive=sensitive), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
            assert not compare_pe(res, pool, recs, 100)
            tot += ost
            assert (m.stats() == tot).all()
        assert m.retries() > 0
        m.close()


@pytest.mark.parametrize("sensitive", [0, 1])
def test_split_calls_and_async_launch_sequence_pe(env, monkeypatch, sensitive):
    from bitmapperbs_amd import synth, mapper
    monkeypatch.setenv("BMBS_LANES", "3")
    monkeypatch.setenv("BMBS_SPLIT_MIN", "1500")
    m = mapper.Mapper(env["ix"], 0, sensitive=sensitive)
    tot = np.zeros(5, dtype=np.int64)
    for seed in (701, 702, 703):
        m1, m2 = synth.make_reads_pe(env["chroms"], n=10000, L=150, seed=seed, sub=0.03, indel=0.002, qual="random")
        res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
        recs, ost, _ = env["oix"].map_pe(orc.params(sensitive=sensitive), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
        assert not compare_pe(res, pool, recs, 150)
        tot += ost
        assert (m.stats() == tot).all()
    # trimmed mates, chunks with their own slices of the length arrays
    m1, m2 = synth.make_reads_pe(env["chroms"], n=8000, L=120, seed=704, sub=0.02, indel=0.002, qual="random")
    rng = np.random.default_rng(9)
    l1 = rng.integers(40, 121, 8000).astype(np.uint16); l2 = rng.integers(40, 121, 8000).astype(np.uint16)
    s1, q1, s2, q2 = _trim(m1["seq"], l1), _trim(m1["qual"], l1), _trim(m2["seq"], l2), _trim(m2["qual"], l2)
    res, pool = m.map_pe_var(s1, q1, s2, q2, l1, l2)
    recs, ost, _ = env["oix"].map_pe_var(orc.params(sensitive=sensitive), s1, q1, s2, q2, l1, l2)
    assert not compare_pe(res, pool, recs, l1, l2)
    m.close()


def test_device_entry_points_split_over_lanes(env, monkeypatch):
    """bmbs_map_pe_device / _var_device with the batch resident in HBM (what bench.py times): chunks on three lanes, several calls
    in flight before bmbs_sync, records equal to the single-lane exact run"""
    import torch
    from bitmapperbs_amd import synth, mapper, capi
    m1, m2 = synth.make_reads_pe(env["chroms"], n=12000, L=150, seed=801, sub=0.03, indel=0.002, qual="random")
    n, L, stride = 12000, 150, 160

    def dev(a):
        t = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
        t[:, :a.shape[1]] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return t
    d = [dev(x) for x in (m1["seq"], m1["qual"], m2["seq"], m2["qual"])]
    rng = np.random.default_rng(3)
    lens = np.concatenate([rng.integers(50, 151, n), rng.integers(50, 151, n)]).astype(np.uint16)
    d_len = torch.from_numpy(lens.view(np.int16)).cuda()
    out = {}
    for tag, envs in (("one", {"BMBS_LANES": "1", "BMBS_EXACT": "1"}), ("split", {"BMBS_LANES": "3", "BMBS_SPLIT_MIN": "1000"})):
        for k_, v_ in envs.items():
            monkeypatch.setenv(k_, v_)
        m = mapper.Mapper(env["ix"], 0)
        ops = m.max_cigar_ops(L)
        res = torch.zeros((2 * n, 32), dtype=torch.uint8, device="cuda")
        cig = torch.zeros((2 * n * ops,), dtype=torch.int32, device="cuda")
        got = []
        for rep in range(3):                         # the second and third call do not wait for their counts
            m.map_pe_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), L, stride, n, res.data_ptr(), cig.data_ptr(), 2 * n * ops)
        m.sync()
        got.append((res.cpu().numpy().view(capi.RESULT_DTYPE).reshape(-1).copy(), cig.cpu().numpy().view(np.uint32).copy()))
        lib = capi.lib()
        rc = lib.bmbs_map_pe_var_device(m._ctx, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_len.data_ptr(), L, stride, n,
                                        res.data_ptr(), cig.data_ptr(), 2 * n * ops)
        assert rc == 0
        m.sync()
        got.append((res.cpu().numpy().view(capi.RESULT_DTYPE).reshape(-1).copy(), cig.cpu().numpy().view(np.uint32).copy()))
        out[tag] = (got, m.stats().copy())
        m.close()
        for k_ in envs:
            monkeypatch.delenv(k_)
    assert (out["one"][1] == out["split"][1]).all()
    for (ra, ca), (rb, cb), Ls in zip(out["one"][0], out["split"][0], (L, None)):
        for f in ("status", "chrom", "pos", "flag", "mapq", "nm", "score", "path", "n_cigar", "tlen"):
            assert (ra[f] == rb[f]).all(), f
        for i in np.nonzero(ra["n_cigar"] > 0)[0]:
            a = ca[int(ra[i]["cigar_off"]):int(ra[i]["cigar_off"]) + int(ra[i]["n_cigar"])]
            b = cb[int(rb[i]["cigar_off"]):int(rb[i]["cigar_off"]) + int(rb[i]["n_cigar"])]
            assert (a == b).all(), i


# ---- FASTQ text in, SAM text out: line index and SAM formatting on the device (bmbs_map_*_text) ---------------------------------------
def _sam_body(text):
    return "".join(l for l in text.splitlines(keepends=True) if not l.startswith("@"))


@pytest.mark.parametrize("name", sorted(golden_args()))
def test_device_sam_text_equals_reference_golden_se(name, tmp_path):
    """the FASTQ file's bytes go to the device as they are; the SAM lines that come back are the reference's, byte for byte"""
    from bitmapperbs_amd import mapper, distributed
    fa = str(tmp_path / "genome.fa")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    mapper.Index.build(fa, fa, threads=4)
    ix = mapper.Index(fa)
    text = gzip.open(os.path.join(GOLD, "se_%s.fq.gz" % name), "rb").read()
    n = text.count(b"\n") // 4
    m = mapper.Mapper(ix, 0, e_f=e_of(golden_args()[name]))
    # the whole file in one call, then the same records in three windows (a window may hold more text than the records asked for)
    mine = m.map_text(text, n).decode()
    assert mine == _sam_body(gzip.open(os.path.join(GOLD, "se_%s.ref.sam.gz" % name), "rt").read())
    assert distributed.mapstats_text(m.stats()) == open(os.path.join(GOLD, "se_%s.ref.stats" % name)).read()
    lines = text.split(b"\n")
    cut = [0, n // 3, n // 3 + 1, n]
    parts = []
    for a, b in zip(cut[:-1], cut[1:]):
        w = b"\n".join(lines[4 * a:]) if b == n else b"\n".join(lines[4 * a:4 * b + 2]) + b"\n"
        parts.append(m.map_text(w, b - a).decode())
    assert "".join(parts) == mine
    m.close()


@pytest.mark.parametrize("name", sorted(pe_golden_args()))
def test_device_sam_text_equals_reference_golden_pe(name, tmp_path):
    from bitmapperbs_amd import mapper, distributed
    from test_oracle import pe_params
    fa = str(tmp_path / "genome.fa")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    mapper.Index.build(fa, fa, threads=4)
    ix = mapper.Index(fa)
    t1 = gzip.open(os.path.join(GOLD, "pe_%s_1.fq.gz" % name), "rb").read()
    t2 = gzip.open(os.path.join(GOLD, "pe_%s_2.fq.gz" % name), "rb").read()
    n = t1.count(b"\n") // 4
    m = mapper.Mapper(ix, 0, **pe_params(pe_golden_args()[name]))
    mine = m.map_text(t1, n, t2).decode()
    assert mine == _sam_body(gzip.open(os.path.join(GOLD, "pe_%s.ref.sam.gz" % name), "rt").read())
    assert distributed.mapstats_text(m.stats()) == open(os.path.join(GOLD, "pe_%s.ref.stats" % name)).read()
    m.close()


@pytest.mark.gpu
def test_reads_of_999_and_1000_bases_are_refused(env):
    """the reference corrupts its own 1000-byte buffers at 999 and 1000 bases (tests/golden/make_golden.py): nothing to be identical
    to, so every entry point says BMBS_EINVAL (include/bmbs.h, BMBS_MAX_READ = 998) instead of mapping them unpinned; 998 maps"""
    import ctypes as C
    from bitmapperbs_amd import mapper, capi, synth
    lib = capi.lib()
    m = mapper.Mapper(env["ix"], 0)
    m._set_refs()
    for L in (998, 999, 1000):
        r = synth.make_reads_se(env["chroms"], n=8, L=L, seed=3, sub=0.01, indel=0.0, qual="const")
        stride = 1008
        seq = np.zeros((8, stride), dtype=np.uint8); qual = np.zeros((8, stride), dtype=np.uint8)
        seq[:, :L] = r["seq"]; qual[:, :L] = r["qual"]
        res = np.zeros(8 * 32, dtype=np.uint8); pool = np.zeros(8 * 300, dtype=np.uint32); used = C.c_int64(0)
        rc = lib.bmbs_map_se(m._ctx, capi.ptr(seq), capi.ptr(qual), L, stride, 8, capi.ptr(res), capi.ptr(pool), pool.size, C.byref(used))
        assert rc == (0 if L == 998 else -22), (L, rc)
        assert (lib.bmbs_max_cigar_ops(C.byref(m.params), L) > 0) == (L == 998)
        text = b"".join(b"@r%d\n" % i + r["seq"][i].tobytes() + b"\n+\n" + r["qual"][i].tobytes() + b"\n" for i in range(8))
        if L == 998:
            assert m.map_text(text, 8, flags=mapper.Mapper.TEXT_UNMAPPED).count(b"\n") == 8
        else:
            with pytest.raises(RuntimeError, match="998"):
                m.map_text(text, 8)
    m.close()


def test_device_sam_text_odd_input(env):
    """names with blanks and slashes, lower-case bases, a quality line shorter than its sequence, no newline at the end of the
    window, mixed lengths, unmapped_out: the device text against the host-side formatter over the same records"""
    from bitmapperbs_amd import synth, mapper
    r = synth.make_reads_se(env["chroms"], n=3000, L=120, seed=41, sub=0.03, indel=0.002, qual="random", n_rate=0.002)
    rng = np.random.default_rng(6)
    recs = []
    for i in range(3000):
        Li = int(rng.integers(30, 121))
        s = r["seq"][i, :Li].tobytes(); q = r["qual"][i, :Li].tobytes()
        if i % 7 == 0: s = s.lower()
        if i % 11 == 0: q = q[:max(1, Li - 5)]
        nm = b"@read%d" % i + (b" extra/1" if i % 3 == 0 else b"/1" if i % 3 == 1 else b"")
        recs.append(nm + b"\n" + s + b"\n+\n" + q)
    text = b"\n".join(recs)                                  # no '\n' behind the last quality line
    m = mapper.Mapper(env["ix"], 0)
    got = m.map_text(text + b"\n", 3000, flags=mapper.Mapper.TEXT_UNMAPPED).decode().splitlines()
    m.close()
    # expected: the records through the row interface + the host formatter
    m = mapper.Mapper(env["ix"], 0)
    L = 120
    seq = np.zeros((3000, L), dtype=np.uint8); qual = np.zeros((3000, L), dtype=np.uint8); lens = np.zeros(3000, dtype=np.uint16)
    names = []
    for i, rec in enumerate(recs):
        nm, s, _, q = rec.split(b"\n")
        s = s.upper(); q = q + b" " * (len(s) - len(q))
        seq[i, :len(s)] = np.frombuffer(s, dtype=np.uint8); qual[i, :len(s)] = np.frombuffer(q, dtype=np.uint8); lens[i] = len(s)
        names.append(nm[1:])
    res, pool = m.map_se_var(seq, qual, lens)
    m.close()
    exp = []
    for i in range(3000):
        x = res[i]; Li = int(lens[i])
        nm = names[i].decode().split(" ")[0].split("/")[0]
        s = seq[i, :Li]; q = qual[i, :Li]
        if int(x["status"]) == 1:
            if int(x["flag"]) & 16:
                s = mapper._COMP[s][::-1]; q = q[::-1]
            exp.append("%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d" % (nm, int(x["flag"]), env["ix"].chrom_names[int(x["chrom"])], int(x["pos"]),
                       int(x["mapq"]), mapper.cigar_text(x, pool, Li), s.tobytes().decode(), q.tobytes().decode(), int(x["nm"])))
        elif int(x["status"]) != 2:
            exp.append("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s" % (nm, s.tobytes().decode(), q.tobytes().decode()))
    assert got == exp


def _odd_fastq(env, n=6000, L=120, seed=41):
    from bitmapperbs_amd import synth
    r = synth.make_reads_se(env["chroms"], n=n, L=L, seed=seed, sub=0.03, indel=0.002, qual="random", n_rate=0.002)
    rng = np.random.default_rng(6)
    recs = []
    for i in range(n):
        Li = int(rng.integers(30, L + 1))
        s = r["seq"][i, :Li].tobytes(); q = r["qual"][i, :Li].tobytes()
        if i % 7 == 0: s = s.lower()
        if i % 11 == 0: q = q[:max(1, Li - 5)]
        nm = b"@read%d" % i + (b" extra/1" if i % 3 == 0 else b"/1" if i % 3 == 1 else b"")
        recs.append(nm + b"\n" + s + b"\n+\n" + q)
    return b"\n".join(recs) + b"\n"


@pytest.mark.parametrize("case", ["se_odd_unmapped", "se_pbat", "pe_p100", "pe_s150_unmapped", "se_constant_quality"])
def test_device_bam_blocks_inflate_to_the_records_of_the_sam_text(env, case):
    """--bam on the device (bmbs_bam.hip): the BGZF blocks a text call returns with BMBS_TEXT_BAM are well-formed (BC field, CRC-32,
    ISIZE, <= 0xff00 bytes each) and inflate to exactly the BAM records htslib makes of the SAM lines the same call prints without the
    flag (those lines are pinned to the reference's by the golden tests): names, flags, bins, CIGAR words, 4-bit bases, qualities - 33
    (a padded quality is 0xff), NM tag types, mates' positions and signed TLEN; several blocks per call, records straddling blocks"""
    from bitmapperbs_amd import mapper
    from common import sam_to_bam_records, bgzf_blocks
    M = mapper.Mapper
    if case.startswith("se"):
        if case == "se_constant_quality":
            from bitmapperbs_amd import synth
            r = synth.make_reads_se(env["chroms"], n=5000, L=150, seed=77, sub=0.01, indel=0.001, qual="const")
            text = b"".join(b"@q%d\n" % i + r["seq"][i].tobytes() + b"\n+\n" + r["qual"][i].tobytes() + b"\n" for i in range(5000))
            n, flags = 5000, 0
        else:
            text = _odd_fastq(env); n = 6000
            flags = M.TEXT_UNMAPPED if case == "se_odd_unmapped" else (M.TEXT_PBAT | M.TEXT_UNMAPPED)     # (few of these reads map as --pbat reads)
        m = M(env["ix"], 0)
        sam = m.map_text(text, n, flags=flags)
        z = m.map_text(text, n, flags=flags | M.TEXT_BAM)
        names = env["ix"].chrom_names
    else:
        from bitmapperbs_amd import synth
        sens = case.startswith("pe_s")
        L = 150 if sens else 100
        m1, m2 = synth.make_reads_pe(env["chroms"], n=4000, L=L, seed=5, sub=0.03, indel=0.003, qual="random")

        def fq(mm):
            return b"".join(b"@" + mm["names"][i] + b"\n" + mm["seq"][i].tobytes() + b"\n+\n" + mm["qual"][i].tobytes() + b"\n" for i in range(4000))
        t1, t2 = fq(m1), fq(m2)
        n = 4000
        m = M(env["ix"], 0, sensitive=1 if sens else 0)
        flags = M.TEXT_UNMAPPED if case.endswith("unmapped") else 0
        sam = m.map_text(t1, n, t2, flags=flags)
        z = m.map_text(t1, n, t2, flags=flags | M.TEXT_BAM)
        names = env["ix"].chrom_names
    m.close()
    blocks = bgzf_blocks(z)
    assert len(blocks) >= 2
    assert all(len(raw) == 0xff00 for _, raw in blocks[:-1])
    got = b"".join(raw for _, raw in blocks)
    want = sam_to_bam_records(sam, names)
    assert len(got) == len(want)
    assert got == want
    if case == "se_constant_quality":
        assert len(z) < len(got) // 2             # runs + Huffman: constant qualities and 4-bit bases compress


def test_device_huffman_lengths_form_a_complete_code(env):
    """the length-limited Huffman codes of the device's BGZF deflater (bmbs_bam.hip: huff_lengths): whatever the symbol frequencies --
    Fibonacci counts whose tree is 8 levels deeper than the 15-bit limit, the same under a bush of other symbols (the shape of the BAM
    block a fuzz trial of round 4 found over-subscribed by one code: zlib refused it), the 19-symbol code-length code at 7 bits, random
    tables -- every used symbol gets a length within the limit, rarer symbols never get shorter codes, and the Kraft sum is exactly one"""
    import ctypes as C
    from fractions import Fraction
    from bitmapperbs_amd import mapper, capi
    m = mapper.Mapper(env["ix"], 0)
    L = capi.lib()
    rng = np.random.default_rng(11)
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    tables = [(fib, 15), (fib[:21] + [int(x) for x in rng.integers(50, 3000, 60)], 15), (fib[:19], 7), ([1] * 19, 7), ([5, 0, 0, 9], 15)]
    for _ in range(150):
        n = int(rng.integers(2, 321))
        f = (rng.integers(0, 4, n) == 0) * rng.integers(1, 1 << int(rng.integers(1, 20)), n)
        f = f * (rng.random(n) < rng.random())
        tables.append(([int(x) for x in f], int(rng.choice([7, 9, 15])) if n <= 19 else 15))
    for freq, maxbits in tables:
        if sum(1 for x in freq if x) < 2:
            continue
        a = np.array(freq, dtype=np.uint32); out = np.zeros(len(freq), dtype=np.uint8)
        assert L.bmbs_debug_huff_lengths(m._ctx, capi.ptr(a), len(freq), maxbits, capi.ptr(out)) == 0
        used = [(int(x), int(l)) for x, l in zip(freq, out) if x]
        assert all(1 <= l <= maxbits for _, l in used), (freq, out)
        assert all(int(l) == 0 for x, l in zip(freq, out) if not x)
        assert sum(Fraction(1, 1 << l) for _, l in used) == 1, (freq, list(out))
        assert all(l1 >= l2 for (x1, l1) in used for (x2, l2) in used if x1 < x2), (freq, list(out))
    m.close()


def test_device_bgzf_inflate_equals_zlib(env):
    """bmbs_inflate_bgzf (one wave per BGZF block, bmbs_inflate.hip) against zlib: FASTQ text at every compression level and
    strategy (dynamic, fixed and stored blocks, codes longer than the root tables, runs that overlap themselves, several deflate
    blocks inside one BGZF block, empty blocks), random bytes, and blocks that zlib refuses -- a flipped bit, a wrong CRC, a wrong
    ISIZE -- which have to be refused here as well"""
    import struct
    import zlib
    from bitmapperbs_amd import mapper
    rng = np.random.default_rng(3)
    fq = _odd_fastq(env, n=4000)

    def member(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        if flush_every:
            z = b"".join(co.compress(chunk[i:i + flush_every]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(chunk), flush_every)) + co.flush()
        else:
            z = co.compress(chunk) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(z) + 25) + z +
                struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))

    def bgzf(data, size, **kw):
        return b"".join(member(data[i:i + size], **kw) for i in range(0, len(data), size))

    skew = bytes(rng.choice(256, size=60000, p=(lambda w: w / w.sum())(1.0 / np.arange(1, 257) ** 3)).astype(np.uint8))     # 15-bit codes
    runs = b"".join(bytes([int(rng.integers(65, 70))]) * int(rng.integers(1, 600)) for _ in range(400))
    cases = {
        "l1": bgzf(fq, 65280, level=1), "l6": bgzf(fq, 65280, level=6), "l9_small_blocks": bgzf(fq, 3000, level=9),
        "stored": bgzf(fq[:200000], 60000, level=0), "fixed": bgzf(fq[:300000], 50000, strategy=zlib.Z_FIXED),
        "huffman_only": bgzf(fq[:300000], 65280, strategy=zlib.Z_HUFFMAN_ONLY), "rle": bgzf(fq[:300000], 65280, strategy=zlib.Z_RLE),
        "several_deflate_blocks": bgzf(fq[:400000], 65280, flush_every=7000), "random_bytes": bgzf(bytes(rng.integers(0, 256, 200000, dtype=np.uint8)), 65280),
        "long_codes": bgzf(skew, 65280, level=9), "runs": bgzf(runs, 65280, level=6),
        "with_empty_blocks": member(b"") + bgzf(fq[:100000], 40000) + member(b"") + member(b"x"),
    }
    m = mapper.Mapper(env["ix"], 0)
    for name, z in cases.items():
        want = zlib.decompressobj(31)
        ref = b""
        d = z
        while d:
            o = zlib.decompressobj(31); ref += o.decompress(d); d = o.unused_data
        got = m.inflate_bgzf(z)
        assert got == ref, (name, len(got), len(ref))
        # ... and the newline counts per 64 KiB of a window in which the text starts `shift` bytes in
        for shift in (0, 1, 40000, 65536, 70001):
            got2, cnt = m.inflate_bgzf(z, shift=shift)
            arr = np.frombuffer(b"\0" * shift + ref, dtype=np.uint8) == 10
            want = [int(arr[i:i + 65536].sum()) for i in range(0, max(1, arr.size), 65536)]
            assert got2 == ref and cnt.tolist()[:len(want)] == want, (name, shift)
    good = cases["l6"]
    first = struct.unpack("<H", good[16:18])[0] + 1
    for name, pos, xor in (("flipped_bit", 30, 0x10), ("crc", first - 8, 0x01), ("isize_vs_data", 40, 0x80)):
        b = bytearray(good); b[pos] ^= xor
        with pytest.raises(RuntimeError):
            m.inflate_bgzf(bytes(b))
    m.close()


@pytest.mark.parametrize("mode", ["se", "pe", "pe_bam"])
def test_bgzf_windows_opened_on_the_device_map_like_the_text(env, mode):
    """bmbs_text_open_bgzf + bmbs_text_map_open (compressed input that never leaves the device): a FASTQ file cut into windows of BGZF
    blocks at block boundaries -- records straddle windows, so every window carries the previous one's tail as its prefix, the
    mates' windows hold different numbers of records, the last line has no newline -- gives, window by window, exactly the lines
    bmbs_map_*_text prints for the whole text"""
    from bitmapperbs_amd import synth, mapper
    from common import write_bgzf
    M = mapper.Mapper
    pe = mode != "se"
    n = 5000
    if pe:
        m1, m2 = synth.make_reads_pe(env["chroms"], n=n, L=120, seed=61, sub=0.02, indel=0.002, qual="random")
        rng = np.random.default_rng(7)
        l1 = rng.integers(40, 121, n); l2 = rng.integers(40, 121, n)
        texts = [b"".join(b"@" + mm["names"][i] + b"\n" + mm["seq"][i, :ll[i]].tobytes() + b"\n+\n" + mm["qual"][i, :ll[i]].tobytes() + b"\n" for i in range(n))[:-1]
                 for mm, ll in ((m1, l1), (m2, l2))]
    else:
        texts = [_odd_fastq(env, n=n)[:-1]]
    flags = M.TEXT_UNMAPPED | (M.TEXT_BAM if mode == "pe_bam" else 0)
    m = M(env["ix"], 0)
    want = m.map_text(texts[0] + b"\n", n, (texts[1] + b"\n") if pe else None, flags=flags)
    # windows: different block sizes for the two files, a few blocks per window
    import io, tempfile, os as _os
    files = []
    for k, t in enumerate(texts):
        f = tempfile.mktemp(suffix=".gz")
        write_bgzf(f, t, block=[9000, 13000][k], level=1)
        z = open(f, "rb").read(); _os.unlink(f)
        files.append(z[:-28])                                   # without the EOF marker block
    def blocks(z):
        import struct
        out = []; at = 0
        while at < len(z):
            bs = struct.unpack("<H", z[at + 16:at + 18])[0] + 1
            out.append(z[at:at + bs]); at += bs
        return out
    bl = [blocks(z) for z in files]
    pos = [0, 0]; carry = [b"", b""]
    got = []; total = 0
    step = [7, 5]
    while True:
        w = []
        for k in range(len(texts)):
            take = bl[k][pos[k]:pos[k] + step[k]]; pos[k] += len(take)
            w.append((carry[k], b"".join(take)))
        last = tuple(pos[k] >= len(bl[k]) for k in range(len(texts))) + ((False,) if not pe else ())
        nrec, t1, t2 = m.text_open_bgzf(w[0], w[1] if pe else None, max_records=700, last=(last[0], last[1] if pe else False))
        carry = [t1, t2]
        if nrec:
            got.append(m.text_map_open(flags=flags)); total += nrec
        if all(last[:len(texts)]) and (nrec == 0 or not t1 or (pe and not t2)):
            break
    assert total == n
    if mode == "pe_bam":
        from common import bgzf_blocks
        assert b"".join(raw for _, raw in bgzf_blocks(b"".join(got))) == b"".join(raw for _, raw in bgzf_blocks(want))
    else:
        assert b"".join(got) == want
    m.close()


def test_text_call_with_a_small_buffer_can_be_repeated_and_counts_once(env):
    """BMBS_ENOMEM of bmbs_map_*_text (output buffer too small) is retriable (*sam_bytes = the size needed): the batch it mapped is
    NOT added to the context's mapstats, so that the repeat counts it once (SAM text and BAM)"""
    import ctypes as C
    from bitmapperbs_amd import mapper, capi
    text = _odd_fastq(env, n=3000)
    a1 = np.frombuffer(text, dtype=np.uint8)
    for flags in (mapper.Mapper.TEXT_UNMAPPED, mapper.Mapper.TEXT_UNMAPPED | mapper.Mapper.TEXT_BAM):
        m = mapper.Mapper(env["ix"], 0)
        m._set_refs()
        ref = m.map_text(text, 3000, flags=flags)
        st1 = m.stats().copy()
        small = np.empty(1000, dtype=np.uint8)
        used = C.c_uint64(0); lines = C.c_int64(0)
        rc = m._lib.bmbs_map_se_text(m._ctx, capi.ptr(a1), a1.size, 3000, flags, capi.ptr(small), small.size, C.byref(used), C.byref(lines))
        assert rc == -12 and used.value == len(ref)
        assert (m.stats() == st1).all()
        again = m.map_text(text, 3000, flags=flags, cap=int(used.value))
        assert again == ref
        assert (m.stats() == 2 * st1).all()
        m.close()


@pytest.mark.parametrize("kind,name,nparts,mode", [
    ("se", "b150", 3, "plain"), ("pe", "p100", 4, "plain"), ("pe", "s100", 2, "names_differ"), ("pe", "p150", 3, "no_names"),
    ("se", "e75", 2, "gz"), ("pe", "p75", 3, "bam"), ("pe", "p100", 2, "bgzf"), ("se", "b150", 3, "bgzf"), ("pe", "p150", 2, "bgzf_tails_grow"),
    ("se", "b150", 2, "gz"), ("pe", "p100", 2, "gz"),
])
def test_cpp_driver_out_parts_concatenate_to_the_one_file_output(kind, name, nparts, mode, tmp_path):
    """--out-parts N: the input is cut into N record ranges (pairs: at the same record in both files, found by the read names, or by
    counting lines when the names do not identify a record), every range is read, mapped and written by a pipeline of its own,
    and `cat` of the parts in order is byte for byte what the one-file run writes (= the reference's output)"""
    import shutil
    import subprocess
    from bitmapperbs_amd import mapper
    fa = str(tmp_path / "genome.fa"); out = str(tmp_path / "o.sam")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    mapper.Index.build(fa, fa, threads=4)
    if kind == "se":
        fq = str(tmp_path / "r.fq")
        gunzip_to(os.path.join(GOLD, "se_%s.fq.gz" % name), fq)
        if mode == "gz":
            shutil.copy(os.path.join(GOLD, "se_%s.fq.gz" % name), fq + ".gz"); fq += ".gz"
        if mode == "bgzf":              # bgzip-style input: independent blocks, inflated by several threads
            from common import write_bgzf
            write_bgzf(fq + ".gz", open(fq, "rb").read(), block=7000); fq += ".gz"
        inp = ["--seq", fq]; args = golden_args()[name]
    else:
        f1 = str(tmp_path / "1.fq"); f2 = str(tmp_path / "2.fq")
        gunzip_to(os.path.join(GOLD, "pe_%s_1.fq.gz" % name), f1)
        gunzip_to(os.path.join(GOLD, "pe_%s_2.fq.gz" % name), f2)
        if mode == "names_differ":          # mate files whose names differ behind a blank, and records of different sizes in the two files
            t = open(f2, "rb").read().split(b"\n")
            for i in range(0, len(t) - 1, 4):
                t[i] = t[i] + b" 2:N:0:" + b"X" * ((i // 4) % 17)
            open(f2, "wb").write(b"\n".join(t))
        if mode == "no_names":              # every record has the same name: the cut has to be found by counting lines
            for f in (f1, f2):
                t = open(f, "rb").read().split(b"\n")
                for i in range(0, len(t) - 1, 4):
                    t[i] = b"@same"
                open(f, "wb").write(b"\n".join(t))
        if mode == "gz":                    # ordinary one-member .gz files: the driver's block-parallel host inflater (pgz.h)
            shutil.copy(os.path.join(GOLD, "pe_%s_1.fq.gz" % name), f1 + ".gz"); shutil.copy(os.path.join(GOLD, "pe_%s_2.fq.gz" % name), f2 + ".gz")
            f1 += ".gz"; f2 += ".gz"
        if mode in ("bgzf", "bgzf_tails_grow"):
            from common import write_bgzf
            write_bgzf(f1 + ".gz", open(f1, "rb").read(), block=9000); write_bgzf(f2 + ".gz", open(f2, "rb").read(), block=65000)
            f1 += ".gz"; f2 += ".gz"
        inp = ["--seq1", f1, "--seq2", f2]; args = pe_golden_args()[name]
    if mode == "bam":
        args = args + ["--bam"]
    # (bgzf_tails_grow: the device-side reader starts with 8-byte buffers for what a window leaves behind its last whole record and has to
    # come back with room, bmbs_text_open_bgzf's BMBS_ENOMEM protocol)
    env = dict(os.environ, BMBS_Z_TAIL="8") if mode == "bgzf_tails_grow" else None
    one = subprocess.run([_driver(), "--search", fa] + inp + ["-o", out, "--batch", "211"] + args, capture_output=True, text=True, env=env)
    assert one.returncode == 0, one.stderr
    par = subprocess.run([_driver(), "--search", fa] + inp + ["-o", out + ".p", "--batch", "211", "--out-parts", str(nparts)] + args,
                         capture_output=True, text=True)
    assert par.returncode == 0, par.stderr
    parts = [open(out + ".p.part%03d" % i, "rb").read() for i in range(nparts)]
    if mode not in ("gz", "bgzf", "bgzf_tails_grow"):
        assert all(len(x) > 0 for x in parts[1:])         # every part got its share
    if mode == "bam":
        from common import bam_payload
        cat = str(tmp_path / "cat.bam")
        open(cat, "wb").write(b"".join(parts))
        assert bam_payload(cat) == bam_payload(out)
    else:
        strip = lambda b: b"".join(l for l in b.splitlines(keepends=True) if not l.startswith(b"@PG"))
        assert strip(b"".join(parts)) == strip(open(out, "rb").read())
        if mode in ("plain", "gz", "bgzf", "bgzf_tails_grow"):
            ref = gzip.open(os.path.join(GOLD, "%s_%s.ref.sam.gz" % (kind, name)), "rb").read()
            assert strip(b"".join(parts)) == ref
    st = lambda p: "".join(l + "\n" for l in p.stderr.splitlines() if l.startswith("No. of") or l.startswith("Mismatch"))
    assert st(one) == st(par)


# ---- a text of more than 2^32 symbols, for real (the wide device forms without any override) ---------------------------------------------
@pytest.fixture(scope="module")
def wide_env(tmp_path_factory):
    """2.2 Gb uniform-random genome (4.4 G doubled symbols > 2^32: 64-bit suffix array, three Occ super-blocks at 2^31, the 3^21 outcome
    table when the device has the room), index built on the device; the oracle loads the very same files"""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from bitmapperbs_amd import synth, mapper
    wd = tmp_path_factory.mktemp("wide")
    names, chroms = synth.make_genome(2_200_000_000, 12, seed=4242)
    fa = str(wd / "w.fa")
    synth.write_fasta(fa, names, chroms)
    mapper.Index.build(fa, fa, threads=min(64, os.cpu_count() or 8), device=0)
    ix = mapper.Index(fa)
    assert 2 * ix.ref_len + 1 > (1 << 32)
    env = dict(fa=fa, chroms=chroms, ix=ix, oix=orc.OrcIndex(fa))
    yield env
    env["oix"].close(); ix.close()
    for f in os.listdir(str(wd)):
        os.unlink(os.path.join(str(wd), f))


def test_true_size_wide_index_matches_oracle(wide_env, monkeypatch):
    """reads over the whole 2.2 Gb genome -- half of the suffix-array rows they touch lie beyond 2^32 -- single end, pairs in fast
    mode and --sensitive, every record and the statistics against the oracle; SA[row] for rows on both sides of 2^32 and of the
    super-block borders.  Nothing is forced: this is the code path a GRCh38 index takes (bwt.h:1059-1070, bwt.cpp:2563-2643)"""
    from bitmapperbs_amd import synth, mapper
    for v in ("BMBS_WIDE", "BMBS_SUPER_SHIFT", "BMBS_TDEPTH", "BMBS_T20"):
        monkeypatch.delenv(v, raising=False)
    env = wide_env
    G = env["ix"].ref_len
    oix = env["oix"]
    m = mapper.Mapper(env["ix"], 0, e_f=0.08)
    rng = np.random.default_rng(12)
    edges = [0, 1, (1 << 31) - 1, 1 << 31, (1 << 31) + 1, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, 2 * G - 1, 2 * G]
    rows = np.concatenate([rng.integers(0, 2 * G + 1, 6000), rng.integers(1 << 32, 2 * G + 1, 6000), edges]).astype(np.uint64)
    got = m.locate(rows)
    exp = np.array([oix.L.orc_sa_at(oix.h, int(r)) for r in rows], dtype=np.uint64)
    assert (got == exp).all()
    assert (got >= (1 << 32)).sum() > 100                       # text positions beyond 32 bits really occur (2.4 % of a 4.4 G text)
    r = synth.make_reads_se(env["chroms"], n=20000, L=150, seed=51, sub=0.02, indel=0.002, qual="random", n_rate=0.001)
    res, pool = m.map_se(r["seq"], r["qual"], 150)
    recs, ost, _ = oix.map_se(orc.params(e_f=0.08), r["seq"], r["qual"], 150)
    assert (recs["status"] == 1).sum() > 18000
    assert not compare_records(res, pool, recs, 150)
    assert (m.stats() == ost).all()
    m.close()
    for sensitive in (0, 1):
        m1, m2 = synth.make_reads_pe(env["chroms"], n=10000, L=150, seed=52 + sensitive, sub=0.04 if sensitive else 0.02, indel=0.002, qual="random")
        m = mapper.Mapper(env["ix"], 0, sensitive=sensitive)
        res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
        recs, ost, _ = oix.map_pe(orc.params(sensitive=sensitive), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
        assert (recs["status"] == 1).sum() > 8000
        assert not compare_pe(res, pool, recs, 150)
        assert (m.stats() == ost).all()
        m.close()


@pytest.mark.parametrize("sensitive", [0, 1])
def test_long_list_kernels_forced_on_pairs(env, monkeypatch, sensitive):
    """k_pe_filter_pairs_long (a wave per pair whose candidate lists are long) forced for every pair above the threshold
    (BMBS_PEF_LONG=2), with a minimum insert that makes the hit rule order-dependent (lane 0 keeps the reference's loop) and
    without; on the second call the mid-list vote kernel is on as well (the first call told the context that the input is repeat-rich)"""
    from bitmapperbs_amd import synth, mapper
    monkeypatch.setenv("BMBS_PEF_LONG", "2")
    for prm in (dict(), dict(min_ins=260, max_ins=700)):
        m1, m2 = synth.make_reads_pe(env["chroms"], n=12000, L=100, seed=991 + sensitive, sub=0.03, indel=0.002, qual="random", ins_hi=680)
        m = mapper.Mapper(env["ix"], 0, sensitive=sensitive, **prm)
        for rep in range(2):
            res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
            recs, ost, _ = env["oix"].map_pe(orc.params(sensitive=sensitive, **prm), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
            assert not compare_pe(res, pool, recs, 100), (prm, rep)
        m.close()


@pytest.fixture(scope="module")
def both_mates_repeat_env(tmp_path_factory):
    """1 Mb genome, 40 % of it copies of a 600-base element at 1.5 % divergence and identical copies of a 450-base one: a pair that
    falls inside a copy has hundreds of candidates on BOTH mates and as many admissible pairs"""
    from bitmapperbs_amd import synth, mapper
    d = tmp_path_factory.mktemp("bothrep")
    names, chroms = synth.make_genome(1_000_000, 2, seed=4242)
    rng = np.random.default_rng(2424)
    for elen, copies, div in ((600, 700, 0.015), (450, 300, 0.0)):
        el = synth._ACGT[rng.integers(0, 4, elen)]
        for _ in range(copies):
            ch = chroms[int(rng.integers(0, 2))]
            p_ = int(rng.integers(0, ch.size - elen))
            e = el.copy()
            m_ = rng.random(elen) < div
            e[m_] = synth._ACGT[rng.integers(0, 4, int(m_.sum()))]
            ch[p_:p_ + elen] = synth.revcomp(e) if rng.random() < 0.5 else e
    fa = str(d / "both.fa")
    synth.write_fasta(fa, names, chroms)
    mapper.Index.build(fa, fa, threads=8)
    ix, oix = mapper.Index(fa), orc.OrcIndex(fa)
    m1, m2 = synth.make_reads_pe(chroms, n=5000, L=100, seed=515, sub=0.01, indel=0.001, qual="random")
    se = synth.make_reads_se(chroms, n=8000, L=120, seed=516, sub=0.01, indel=0.001, qual="random")
    yield dict(ix=ix, oix=oix, m1=m1, m2=m2, se=se, chroms=chroms)
    ix.close(); oix.close()


@pytest.mark.parametrize("form", ["block"])
def test_sensitive_reseeded_mates_with_long_lists_match_oracle(both_mates_repeat_env, monkeypatch, form):
    """--sensitive on diverged pairs inside the repeat families: a mate left without a hit is re-seeded and its candidates (hundreds)
    are located, sorted, made distinct and tested against the verified mate's hits -- by a block per mate (k_pes_vote_long: binary
    searches instead of the reference's running lower bound; shorter lists by one lane running the reference's loop);
    with the default insert range and with a minimum insert (the lower bound of the distance above zero)"""
    from bitmapperbs_amd import synth, mapper
    e = both_mates_repeat_env
    m1, m2 = synth.make_reads_pe(e["chroms"], n=5000, L=100, seed=616, sub=0.05, indel=0.002, qual="random")
    for prm in (dict(), dict(min_ins=150, max_ins=600)):
        recs, ost, _ = e["oix"].map_pe(orc.params(sensitive=1, **prm), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
        m = mapper.Mapper(e["ix"], 0, sensitive=1, **prm)
        for rep in range(2):
            res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
            assert not compare_pe(res, pool, recs, 100), (form, prm, rep)
        m.close()


@pytest.mark.parametrize("class3", ["128"])
def test_long_lists_single_end_size_classes_match_oracle(both_mates_repeat_env, monkeypatch, class3):
    """single-end reads on the same genome: lists of hundreds of candidates through the three forms of k_vote_long (a wave; a block
    of 128 or 256 threads over up to 1024 keys; 256 threads over up to 4096) -- sites, votes and std::sort's visiting order must
    be the same whichever form a list takes, with and without --ambiguous_out"""
    from bitmapperbs_amd import mapper
    e = both_mates_repeat_env
    r = e["se"]
    for amb in (0, 1):
        recs, ost, cnt = e["oix"].map_se(orc.params(ambiguous_out=amb), r["seq"], r["qual"], 120)
        m = mapper.Mapper(e["ix"], 0, ambiguous_out=amb)
        for rep in range(2):
            res, pool = m.map_se(r["seq"], r["qual"], 120)
            assert int((res["n_cand"] > 256).sum()) > 100           # lists beyond the wave form
            assert not compare_records(res, pool, recs, 120, amb=bool(amb)), (class3, amb, rep)
        m.close()


@pytest.mark.parametrize("variant", ["default", "min_insert", "ambiguous_out", "sensitive"])
def test_long_lists_on_both_mates_match_oracle(both_mates_repeat_env, monkeypatch, variant):
    """the wave-cooperative parts of k_pe_compact / k_pe_prune / k_pe_pair (lists of more than 64 entries on both mates: compaction by
    ballot, prune by binary search, pairing by ordered summaries) against the oracle's loops -- ties of the best error sum, the
    early exit at a second pair without errors, `second_best_diff`; with a minimum insert the reference's loop is order-dependent
    and one lane runs it"""
    from bitmapperbs_amd import mapper
    e = both_mates_repeat_env
    prm, sensitive = {}, 0
    if variant == "min_insert":
        prm = dict(min_ins=150, max_ins=600)
    elif variant == "ambiguous_out":
        prm = dict(ambiguous_out=1)
    elif variant == "sensitive":
        sensitive = 1
    m1, m2 = e["m1"], e["m2"]
    recs, ost, _ = e["oix"].map_pe(orc.params(sensitive=sensitive, **prm), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
    m = mapper.Mapper(e["ix"], 0, sensitive=sensitive, **prm)
    for rep in range(2):
        res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 100)
        nc = res["n_cand"].astype(np.int64).reshape(-1, 2)
        assert int((nc.min(axis=1) > 64).sum()) > 100         # pairs with long lists on both mates
        assert not compare_pe(res, pool, recs, 100), (variant, rep)
    m.close()


@pytest.fixture(scope="module")
def huge_lists_env(tmp_path_factory):
    """1.4 Mb genome with 990 identical, non-overlapping copies of a 500-base element: a 250-base read inside a copy places up to
    eight seeds of 990 rows each -- lists of 4 000 to 7 000 candidates, beyond the LDS capacity of the long-list kernels"""
    from bitmapperbs_amd import synth, mapper
    d = tmp_path_factory.mktemp("huge")
    names, chroms = synth.make_genome(1_400_000, 2, seed=777)
    rng = np.random.default_rng(888)
    el = synth._ACGT[rng.integers(0, 4, 500)]
    k_ = 0
    for ch in chroms:
        for p_ in range(100, ch.size - 600, 1400):
            if k_ >= 990:
                break
            ch[p_:p_ + 500] = synth.revcomp(el) if rng.random() < 0.5 else el
            k_ += 1
    # eight 31-base elements, one copy of each in the free part of every slot, and ONE place where they stand side by side: a read
    # from there (with a substitution at each junction) has seeds of 991 rows in six or more of them that agree on nothing --
    # about 6 000 candidates and as many distinct sites
    mosaic = [synth._ACGT[rng.integers(0, 4, 31)] for _ in range(8)]
    k_ = 0
    for ch in chroms:
        for p_ in range(100, ch.size - 600, 1400):
            if k_ >= 990:
                break
            for i, el8 in enumerate(mosaic):
                ch[p_ + 560 + 43 * i:p_ + 560 + 43 * i + 31] = el8
            k_ += 1
    chroms[0][1150:1150 + 248] = np.concatenate(mosaic)
    fa = str(d / "huge.fa")
    synth.write_fasta(fa, names, chroms)
    mapper.Index.build(fa, fa, threads=8)
    ix, oix = mapper.Index(fa), orc.OrcIndex(fa)
    yield dict(ix=ix, oix=oix, chroms=chroms, mosaic=np.concatenate(mosaic))
    ix.close(); oix.close()


@pytest.mark.parametrize("amb", [0, 1])
def test_more_distinct_sites_than_the_lds_holds_match_oracle(huge_lists_env, amb):
    """single-end reads whose seeds hit 991 rows each in six different families and agree on nothing: more than 4096 DISTINCT sites
    -- the votes of such a list cannot be ordered in LDS; the sites are still sorted by the block (vl_sort_huge), std::sort's
    loop on the votes runs on one lane"""
    from bitmapperbs_amd import mapper
    e = huge_lists_env
    rng = np.random.default_rng(99)
    n, L = 200, 248
    seq = np.tile(e["mosaic"], (n, 1)).copy()
    other = {65: 67, 67: 71, 71: 84, 84: 65}
    for i in range(n):
        for el8 in range(1, 8):                            # one substitution at the end of each of the first seven elements
            seq[i, el8 * 31 - 1] = other[int(seq[i, el8 * 31 - 1])]
        if i % 3 == 0:                                     # ... and one more somewhere
            j = int(rng.integers(5, L - 5))
            seq[i, j] = other[int(seq[i, j])]
    qual = rng.integers(35, 74, (n, L)).astype(np.uint8)
    recs, ost, cnt = e["oix"].map_se(orc.params(ambiguous_out=amb), seq, qual, L)
    assert int((recs["n_cand"] > 4096).sum()) > n // 2
    m = mapper.Mapper(e["ix"], 0, ambiguous_out=amb)
    for rep in range(2):
        res, pool = m.map_se(seq, qual, L)
        assert not compare_records(res, pool, recs, L, amb=bool(amb)), (amb, rep)
    m.close()


@pytest.mark.parametrize("mode", ["se", "se_ambiguous_out", "pe", "pe_sensitive"])
def test_lists_beyond_the_lds_capacity_match_oracle(huge_lists_env, mode):
    """candidate lists of more than 4096 sites: sorted in tiles by a block and merged by rank (vl_sort_huge) in k_vote_long (votes and
    std::sort's visiting order after it), k_vote_pe_long and k_pes_vote_long -- one lane sorting such a list in global memory
    used to hold its whole launch for tens of milliseconds"""
    from bitmapperbs_amd import synth, mapper
    e = huge_lists_env
    L = 250
    if mode.startswith("se"):
        amb = 1 if mode == "se_ambiguous_out" else 0
        r = synth.make_reads_se(e["chroms"], n=4000, L=L, seed=79, sub=0.04, indel=0.001, qual="random")
        recs, ost, cnt = e["oix"].map_se(orc.params(ambiguous_out=amb), r["seq"], r["qual"], L)
        assert int((recs["n_cand"] > 4096).sum()) > 50
        m = mapper.Mapper(e["ix"], 0, ambiguous_out=amb)
        for rep in range(2):
            res, pool = m.map_se(r["seq"], r["qual"], L)
            assert not compare_records(res, pool, recs, L, amb=bool(amb)), (mode, rep)
        m.close()
    else:
        sens = 1 if mode == "pe_sensitive" else 0
        m1, m2 = synth.make_reads_pe(e["chroms"], n=3000, L=L, seed=80, sub=0.05, indel=0.001, qual="random", ins_hi=480)
        recs, ost, _ = e["oix"].map_pe(orc.params(sensitive=sens), m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
        m = mapper.Mapper(e["ix"], 0, sensitive=sens)
        for rep in range(2):
            res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
            assert int((res["n_cand"].astype(np.int64) > 4096).sum()) > 20
            assert not compare_pe(res, pool, recs, L), (mode, rep)
        m.close()


def test_very_long_candidate_lists_match_oracle(tmp_path):
    """reads inside a family of 900 identical copies (and one of 600 copies at 2 % divergence): seeds that hit hundreds of rows each
    give candidate lists of up to 2 400 -- the block form of the long-list kernels with its radix sort (lists beyond 512), the
    wave-per-pair filter; SE and pairs, first call and second (which has the mid-list kernels on)"""
    from bitmapperbs_amd import synth, mapper
    names, chroms = synth.make_genome(2_000_000, 2, seed=1234)
    rng = np.random.default_rng(4321)
    for elen, copies, div in ((120, 900, 0.0), (200, 600, 0.02)):
        el = synth._ACGT[rng.integers(0, 4, elen)]
        for _ in range(copies):
            ch = chroms[int(rng.integers(0, 2))]
            p = int(rng.integers(0, ch.size - elen))
            e = el.copy()
            m_ = rng.random(elen) < div
            e[m_] = synth._ACGT[rng.integers(0, 4, int(m_.sum()))]
            ch[p:p + elen] = synth.revcomp(e) if rng.random() < 0.5 else e
    fa = str(tmp_path / "rep.fa")
    synth.write_fasta(fa, names, chroms)
    mapper.Index.build(fa, fa, threads=8)
    ix, oix = mapper.Index(fa), orc.OrcIndex(fa)
    r = synth.make_reads_se(chroms, n=12000, L=150, seed=77, sub=0.01, indel=0.001, qual="random")
    m = mapper.Mapper(ix, 0, e_f=0.08, ambiguous_out=1)
    recs, ost, cnt = oix.map_se(orc.params(e_f=0.08, ambiguous_out=1), r["seq"], r["qual"], 150)
    assert int(recs["n_cand"].max()) > 1000                     # the lists really get long
    for rep in range(2):
        res, pool = m.map_se(r["seq"], r["qual"], 150)
        assert not compare_records(res, pool, recs, 150, amb=True), rep
    m.close()
    m1, m2 = synth.make_reads_pe(chroms, n=8000, L=150, seed=78, sub=0.01, indel=0.001, qual="random")
    for sensitive in (0, 1):
        m = mapper.Mapper(ix, 0, sensitive=sensitive)
        recs, ost, _ = oix.map_pe(orc.params(sensitive=sensitive), m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
        for rep in range(2):
            res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], 150)
            assert not compare_pe(res, pool, recs, 150), (sensitive, rep)
        m.close()
