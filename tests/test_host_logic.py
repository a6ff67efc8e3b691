"""Host-side pieces that need no GPU: SAM emit from result records, mapstats text, shard ranges, and the
world_size-2 (gloo) run of the multi-process path (shard -> map -> all-reduce stats -> ordered concat).
The per-shard mapper in the gloo test is the oracle standing in for the GPU (no GPU here): what is under
test is the host logic in bitmapperbs_amd.distributed / mapper.sam_lines_se."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

import orc
from common import GOLD, ROOT, e_of, golden_args, gunzip_to, read_fastq


def recs_to_results(recs, L):
    """oracle records -> bmbs_result records (+ cigar pool), the way the device writes them"""
    from bitmapperbs_amd import capi
    n = recs.size
    res = np.zeros(n, dtype=capi.RESULT_DTYPE)
    pool = []
    for i in range(n):
        r = recs[i]
        st = int(r["status"])
        res[i]["status"] = st
        if st in (1, 3):
            res[i]["chrom"] = r["chrom"]; res[i]["pos"] = r["pos"]; res[i]["flag"] = r["flag"]
            res[i]["mapq"] = r["mapq"]; res[i]["nm"] = r["nm"]; res[i]["score"] = r["score"]
            cg = r["cigar"].decode()
            if cg != "%dM" % L:
                ops, num = [], ""
                for ch in cg:
                    if ch.isdigit():
                        num += ch
                    else:
                        ops.append((int(num) << 4) | "MDISH".index(ch)); num = ""
                res[i]["cigar_off"] = len(pool); res[i]["n_cigar"] = len(ops)
                pool.extend(ops)
    return res, np.array(pool, dtype=np.uint32)


@pytest.fixture(scope="module")
def gold(tmp_path_factory, oracle):
    wd = tmp_path_factory.mktemp("host")
    fa = str(wd / "genome.fa")
    gunzip_to(os.path.join(GOLD, "genome.fa.gz"), fa)
    assert oracle.orc_index_build(fa.encode(), fa.encode()) == 0
    return fa, str(wd)


@pytest.mark.parametrize("name", ["b150", "e75"])
def test_host_sam_emit_matches_reference_text(name, gold):
    from bitmapperbs_amd import mapper
    fa, wd = gold
    fq = os.path.join(wd, name + ".fq")
    gunzip_to(os.path.join(GOLD, "se_%s.fq.gz" % name), fq)
    names, seq, qual = read_fastq(fq)
    L = seq.shape[1]
    recs, st, _ = orc.OrcIndex(fa).map_se(orc.params(e_f=e_of(golden_args()[name])), seq, qual, L)
    res, pool = recs_to_results(recs, L)
    ix = mapper.Index(fa)
    text = mapper.sam_header(ix, "") + "".join(mapper.sam_lines_se(ix, names, seq, qual, L, res, pool))
    gold_text = gzip.open(os.path.join(GOLD, "se_%s.ref.sam.gz" % name), "rt").read()
    mine = "".join(l + "\n" for l in text.split("\n")[:-1] if not l.startswith("@PG"))
    assert mine == gold_text


def test_mapstats_text_format(gold):
    from bitmapperbs_amd import distributed
    ref = open(os.path.join(GOLD, "se_b150.ref.stats")).read()
    rows = ref.split("\n")
    reads = int(rows[0].split()[-1]); uniq = int(rows[1].split()[5]); amb = int(rows[2].split()[5])
    # bases/errors are not printed directly; reproduce the rate line through the oracle's counters
    fa, wd = gold
    fq = os.path.join(wd, "b150.fq")
    gunzip_to(os.path.join(GOLD, "se_b150.fq.gz"), fq)
    names, seq, qual = read_fastq(fq)
    recs, st, _ = orc.OrcIndex(fa).map_se(orc.params(e_f=0.04), seq, qual, seq.shape[1])
    assert (st[0], st[1], st[2]) == (reads, uniq, amb)
    assert distributed.mapstats_text(st) == ref


def test_shard_ranges_cover_exactly():
    from bitmapperbs_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
from bitmapperbs_amd import distributed, mapper
import orc
from common import read_fastq
from test_host_logic import recs_to_results
fa, fq, out, e = sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5])
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
names, seq, qual = read_fastq(fq)
L = seq.shape[1]
lo, hi = distributed.shard_range(seq.shape[0], rank, world)
recs, st, _ = orc.OrcIndex(fa).map_se(orc.params(e_f=e), seq[lo:hi], qual[lo:hi], L)   # stand-in for the GPU mapper
res, pool = recs_to_results(recs, L)
ix = mapper.Index(fa)
with open("%s.part%d" % (out, rank), "w") as f:
    f.writelines(mapper.sam_lines_se(ix, names[lo:hi], seq[lo:hi], qual[lo:hi], L, res, pool))
tot = distributed.allreduce_stats(st)
dist.barrier()
distributed.concat_parts(out, mapper.sam_header(ix, ""), world, rank)
if rank == 0:
    open(out + ".stats", "w").write(distributed.mapstats_text(tot))
dist.barrier()
dist.destroy_process_group()
'''


def test_world_size_2_gloo_shard_reduce_concat(gold, tmp_path):
    fa, wd = gold
    fq = os.path.join(wd, "c150.fq")
    gunzip_to(os.path.join(GOLD, "se_c150.fq.gz"), fq)
    worker = str(tmp_path / "worker.py")
    open(worker, "w").write(WORKER)
    out = str(tmp_path / "out.sam")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", worker, ROOT, fa, fq, out,
                        str(e_of(golden_args()["c150"]))], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    mine = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert mine == gzip.open(os.path.join(GOLD, "se_c150.ref.sam.gz"), "rt").read()
    assert open(out + ".stats").read() == open(os.path.join(GOLD, "se_c150.ref.stats")).read()


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must start two ranks itself (as a child torch.distributed.run,
    before anything touches a GPU) and report the world size the process group has; --dry-run = gloo, no mapping"""
    import json
    import subprocess
    import sys
    from common import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2
    assert d["mapstats_sum"][:2] == [3, 2]          # ranks contributed (rank + 1, 1): one all-reduce over both
    assert d["per_rank_s"] == [1.0, 2.0]             # every rank's own timed seconds, gathered in rank order


def test_bench_configs_match_baseline_json():
    """the default bench workload is BASELINE.json's metric configuration (configs[2]: 150 bp PE, GRCh38-size, fast mode)"""
    import importlib.util
    import json
    from common import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    a = b.parse([])
    assert a.config == 2 and a.cfg["pe"] and not a.cfg["sensitive"] and a.cfg["read_len"] == 150
    assert a.cfg["genome"] >= 3_000_000_000 and a.cfg["units"] * a.cfg["launches"] == 50_000_000
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "150bp PE" in base["metric"] and "50 M synthetic 150 bp PE reads vs GRCh38" in base["configs"][2]
    a1 = b.parse(["--config", "1"])
    assert not a1.cfg["pe"] and a1.cfg["units"] == 10_000_000 and abs(a1.cfg["e"] - 0.04) < 1e-12


def _fastq(path, names, lens, rng, at_quals=False):
    with open(path, "wb") as f:
        for nm, L in zip(names, lens):
            seq = bytes(rng.choice(list(b"ACGT"), L).tolist())
            q = bytearray(rng.integers(35, 74, L).astype("u1").tobytes())
            if at_quals and L > 1:
                q[0] = ord("@") if rng.random() < 0.5 else ord("+")       # quality lines that begin like a header / separator line
            f.write(b"@" + nm + b"\n" + seq + b"\n+\n" + bytes(q) + b"\n")


@pytest.mark.parametrize("mode", ["names", "repeated_names", "se"])
def test_out_parts_cut_points_are_the_same_record_in_both_files(tmp_path, mode):
    """bmbs_search --out-parts: the input is cut at record starts, pairs at the SAME record in both files -- found by the read names,
    by counting lines when names repeat; quality lines that begin with '@' or '+' must not be mistaken for headers (host logic: the
    driver prints the cut points and exits before it would need a GPU)"""
    import subprocess
    drv = os.path.join(ROOT, "bitmapperbs_amd", "bmbs_search")
    if not os.path.exists(drv):
        pytest.skip("bmbs_search not built")
    rng = np.random.default_rng(17)
    n = 60000
    l1 = rng.integers(30, 151, n); l2 = rng.integers(30, 151, n)
    if mode == "repeated_names":
        names = [b"same" for _ in range(n)]
    else:
        names = [b"r%d:%d" % (i, int(rng.integers(0, 1 << 30))) for i in range(n)]
    f1 = str(tmp_path / "a_1.fq"); f2 = str(tmp_path / "a_2.fq")
    _fastq(f1, [x + b" 1:N:0" for x in names], l1, rng, at_quals=True)
    _fastq(f2, [x + b" 2:N:0:" + b"X" * (i % 13) for i, x in enumerate(names)], l2, rng, at_quals=True)
    parts = 7
    inp = ["--seq", f1] if mode == "se" else ["--seq1", f1, "--seq2", f2]
    p = subprocess.run([drv, "--search", "unused", "--print-parts", "--out-parts", str(parts), "-t", "4"] + inp, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    rows = [tuple(int(x) for x in l.split()) for l in p.stdout.splitlines()]
    assert len(rows) == parts + 1 and rows[0][1:] == (0, 0)
    t1 = open(f1, "rb").read(); t2 = open(f2, "rb").read()
    assert rows[-1][1] == len(t1)
    starts1 = np.concatenate([[0], np.cumsum(1 + np.array([len(x) for x in names]) + len(b" 1:N:0") + 1 + l1 + 3 + l1 + 1)])
    prev = -1
    for k, c1, c2 in rows[1:-1]:
        rec = int(np.searchsorted(starts1, c1))
        assert starts1[rec] == c1, "cut %d of file 1 is not a record start" % k
        assert rec > prev and 0.5 * n * k / parts < rec < 1.5 * n * k / parts + 10          # roughly even, strictly increasing
        prev = rec
        if mode != "se":
            # the same record index in file 2
            assert t2[:c2].count(b"\n") == 4 * rec, "cut %d: file 2 is not at the record file 1 is at" % k


@pytest.mark.parametrize("kind", ["plain", "gzip", "bgzf_small_blocks", "bgzf_big_blocks"])
def test_driver_reader_hands_on_every_record_once(tmp_path, kind):
    """the FASTQ reader of bmbs_search on its own (bmbs_reader_test: no GPU): whatever the window size, the input format (plain
    text, one-member gzip inflated by one thread, bgzip-style blocks inflated by several threads in parallel) and the number of
    threads, the windows' whole records concatenate to the input text; an unterminated last line gets its newline"""
    import gzip
    import subprocess
    from common import write_bgzf
    exe = os.path.join(ROOT, "bitmapperbs_amd", "bmbs_reader_test")
    if not os.path.exists(exe):
        pytest.skip("bmbs_reader_test not built")
    rng = np.random.default_rng(5)
    n = 20000
    f = str(tmp_path / "r.fq")
    _fastq(f, [b"r%d extra" % i for i in range(n)], rng.integers(20, 200, n), rng, at_quals=True)
    raw = open(f, "rb").read()
    for trunc in (False, True):
        text = raw[:-1] if trunc else raw                       # without the final newline
        src = str(tmp_path / ("t%d" % trunc))
        if kind == "plain":
            open(src, "wb").write(text)
        elif kind == "gzip":
            open(src, "wb").write(gzip.compress(text, 1))
        else:
            write_bgzf(src, text, block=3000 if kind == "bgzf_small_blocks" else 65000)
        for window, threads in ((5000, 1), (70000, 3), (1 << 20, 8), (1 << 26, 16)):
            p = subprocess.run([exe, src, str(window), str(threads)], capture_output=True, timeout=120)
            assert p.returncode == 0, p.stderr
            assert p.stdout == raw, (kind, trunc, window, threads, len(p.stdout), len(raw))


@pytest.mark.parametrize("span", [4096, 100_000, 1 << 20])
def test_parallel_gzip_reader_equals_zlib(tmp_path, span):
    """ordinary gzip input is inflated block-parallel (csrc/pgz.h: speculative block starts, 16-bit window markers, in-order
    resolution): whatever the compression level / strategy, the number of members, flush points, stored and fixed blocks,
    trailing garbage, the span between cuts and the number of threads, the reader hands on exactly the text zlib inflates;
    truncated or corrupted files end with an error, never with different text"""
    import gzip
    import subprocess
    import zlib
    from common import write_bgzf
    exe = os.path.join(ROOT, "bitmapperbs_amd", "bmbs_reader_test")
    if not os.path.exists(exe):
        pytest.skip("bmbs_reader_test not built")
    rng = np.random.default_rng(11)
    n = 12000
    f = str(tmp_path / "r.fq")
    _fastq(f, [b"read%d/1" % i for i in range(n)], rng.integers(100, 151, n), rng, at_quals=True)
    raw = open(f, "rb").read()
    half = raw.index(b"\n@read%d/1" % (n // 2)) + 1

    def strat(st, level=6):
        c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, st)
        return c.compress(raw) + c.flush()

    def flushes():
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        out = []
        for k, i in enumerate(range(0, len(raw), 300_000)):
            out.append(c.compress(raw[i:i + 300_000]))
            out.append(c.flush(zlib.Z_FULL_FLUSH if k % 2 else zlib.Z_SYNC_FLUSH))
        return b"".join(out) + c.flush()

    bg = str(tmp_path / "b.gz")
    write_bgzf(bg, raw[:half], block=40000)
    files = {
        "l1": gzip.compress(raw, 1), "l6": gzip.compress(raw, 6), "l9": gzip.compress(raw, 9), "l0_stored": gzip.compress(raw, 0),
        "members": b"".join(gzip.compress(raw[i:i + 700_001], 4) for i in range(0, len(raw), 700_001)),
        "fixed": strat(zlib.Z_FIXED), "huffman_only": strat(zlib.Z_HUFFMAN_ONLY), "rle": strat(zlib.Z_RLE), "flushes": flushes(),
        "garbage_behind": gzip.compress(raw, 6) + b"\0" * 50 + b"junk",
        "bgzf_then_plain": open(bg, "rb").read() + gzip.compress(raw[half:], 5),
        "named": gzip.compress(raw, 6)[:3] + b"\x08" + gzip.compress(raw, 6)[4:10] + b"reads.fq\0" + gzip.compress(raw, 6)[10:],
    }
    env = dict(os.environ, BMBS_GZ_SPAN=str(span))
    for name, z in files.items():
        assert zlib.decompressobj(31).decompress(z)[:64] == raw[:64], name
        src = str(tmp_path / (name + ".gz"))
        open(src, "wb").write(z)
        for threads in (1, 5):
            p = subprocess.run([exe, src, str(1 << 22), str(threads)], capture_output=True, timeout=120, env=env)
            assert p.returncode == 0, (name, threads, p.stderr)
            assert p.stdout == raw, (name, span, threads, len(p.stdout), len(raw))
    z = files["l6"]
    bad = {"truncated": z[:len(z) // 2], "no_trailer": z[:-8]}
    for k in range(3):
        b = bytearray(z)
        b[len(z) // 4 * (k + 1) + 7 * k] ^= 0x20 << k
        bad["flipped%d" % k] = bytes(b)
    for name, zb in bad.items():
        src = str(tmp_path / (name + ".gz"))
        open(src, "wb").write(zb)
        p = subprocess.run([exe, src, str(1 << 22), "4"], capture_output=True, timeout=120, env=env)
        ok = True
        try:
            want = zlib.decompressobj(31).decompress(zb)
            ok = zb is z
        except zlib.error:
            ok = False
        if not ok:
            assert p.returncode != 0 and p.stderr, name
        else:
            assert p.stdout == want


def test_driver_plan_for_an_eight_gpu_node(tmp_path):
    """bmbs_search --devices 0,..,7 --out-parts 8 --print-plan (no GPU touched): the input is cut into eight record ranges at record
    starts (pairs: at the same record in both files), every device gets an index copy and `--contexts` contexts, one worker per
    context takes batches of any range, and the statistics are the sum over all of them -- SURVEY section 8e's partition"""
    import subprocess
    from common import ROOT
    drv = os.path.join(ROOT, "bitmapperbs_amd", "bmbs_search")
    if not os.path.exists(drv):
        pytest.skip("bmbs_search not built")
    rng = np.random.default_rng(8)
    n = 16000
    names = [b"p%d" % i for i in range(n)]
    f1 = str(tmp_path / "a_1.fq"); f2 = str(tmp_path / "a_2.fq")
    l1 = rng.integers(50, 151, n); l2 = rng.integers(50, 151, n)
    _fastq(f1, [x + b"/1" for x in names], l1, rng); _fastq(f2, [x + b"/2" for x in names], l2, rng)
    p = subprocess.run([drv, "--search", "unused", "--print-plan", "--devices", "0,1,2,3,4,5,6,7", "--contexts", "3", "--out-parts", "8", "-t", "8",
                        "--seq1", f1, "--seq2", f2], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    rows = [l.split("\t") for l in p.stdout.splitlines()]
    cuts = [tuple(int(x) for x in r) for r in rows if r[0].isdigit()]
    assert len(cuts) == 9 and cuts[0][1:] == (0, 0) and cuts[-1][1:] == (os.path.getsize(f1), os.path.getsize(f2))
    t1 = open(f1, "rb").read(); t2 = open(f2, "rb").read()
    recs = [t1[:c1].count(b"\n") // 4 for _, c1, _ in cuts]
    assert recs == sorted(recs) and all(t2[:c2].count(b"\n") == 4 * r for (_, _, c2), r in zip(cuts, recs))
    assert all(abs((b - a) - n / 8) < n / 16 for a, b in zip(recs, recs[1:]))                   # even ranges
    plan = [r for r in rows if r[0] == "plan"][0]
    kv = dict(zip(plan[1::2], plan[2::2]))
    assert (kv["devices"], kv["contexts_per_device"], kv["workers"], kv["parts"]) == ("8", "3", "24", "8")
    assert [r[1] for r in rows if r[0] == "device"] == [str(i) for i in range(8)]


def test_pack_rows_refuses_what_the_format_cannot_hold():
    """bmbs_pack_rows (host code: runs without a device): letters other than A C G T N have no place in two bits and a mark"""
    from bitmapperbs_amd import mapper
    seq = np.frombuffer(b"ACGTNACGTNACGTNA" * 4, dtype=np.uint8).reshape(2, 32).copy()
    rows = mapper.Mapper.pack_rows(seq, 30)
    assert rows.shape == (2, 2)
    want = 0
    for j, ch in enumerate(b"ACGTNACGTNACGTNAACGTNACGTNACGTN"[:30]):
        want |= {65: 0, 67: 1, 71: 2, 84: 3, 78: 0}[ch] << (2 * j)
    assert int(rows[0, 0]) == want
    assert int(rows[0, 1]) == sum(1 << j for j, ch in enumerate(b"ACGTNACGTNACGTNAACGTNACGTNACGTN"[:30]) if ch == 78)
    seq[1, 7] = ord("R")
    with pytest.raises(ValueError, match="row 1"):
        mapper.Mapper.pack_rows(seq, 30)
