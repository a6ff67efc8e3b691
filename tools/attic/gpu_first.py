"""First end-to-end GPU check (development aid; the formal versions live in tests/ -m gpu)."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bitmapperbs_amd import synth, mapper, capi
import orc

def main():
    wd = tempfile.mkdtemp(prefix="bmbs_")
    size = int(os.environ.get("GSIZE", 2_000_000))
    names, chroms = synth.make_genome(size, 4, seed=11)
    # plant a few repeats so the multi-hit paths run
    rng = np.random.default_rng(5)
    for (elen, copies, div) in [(400, 60, 0.03), (150, 100, 0.0)]:
        el = synth._ACGT[rng.integers(0, 4, elen)]
        for c in range(copies):
            ch = chroms[rng.integers(0, len(chroms))]
            p = int(rng.integers(0, ch.size - elen))
            e = el.copy(); m = rng.random(elen) < rng.random() * div
            e[m] = synth._ACGT[rng.integers(0, 4, int(m.sum()))]
            ch[p:p + elen] = e
    fa = os.path.join(wd, "g.fa")
    synth.write_fasta(fa, names, chroms)
    t = time.time(); mapper.Index.build(fa, fa, 8); print("index build %.1fs" % (time.time() - t))
    ix = mapper.Index(fa)
    oix = orc.OrcIndex(fa)
    sets = [dict(n=20000, L=100, seed=1, sub=0.005, indel=0.0002, qual="const", e=0.08),
            dict(n=20000, L=150, seed=2, sub=0.02, indel=0.002, qual="random", n_rate=0.002, e=0.04),
            dict(n=20000, L=150, seed=3, sub=0.03, indel=0.002, qual="random", e=0.08),
            dict(n=8000, L=250, seed=4, sub=0.03, indel=0.001, qual="random", e=0.08)]
    ok_all = True
    for s in sets:
        e = s.pop("e")
        r = synth.make_reads_se(chroms, **s)
        L = s["L"]
        m = mapper.Mapper(ix, 0, e_f=e)
        t = time.time(); res, pool = m.map_se(r["seq"], r["qual"], L); dt = time.time() - t
        st = m.stats()
        prm = orc.params(e_f=e)
        t = time.time(); recs, ost, cnt = oix.map_se(prm, r["seq"], r["qual"], L); odt = time.time() - t
        bad = 0
        for i in range(s["n"]):
            a, b = res[i], recs[i]
            same = int(a["status"]) == int(b["status"])
            if same and int(a["status"]) in (1, 3):
                same = (int(a["chrom"]) == int(b["chrom"]) and int(a["pos"]) == int(b["pos"]) and int(a["flag"]) == int(b["flag"])
                        and int(a["mapq"]) == int(b["mapq"]) and int(a["nm"]) == int(b["nm"]) and int(a["score"]) == int(b["score"])
                        and mapper.cigar_text(a, pool, L) == b["cigar"].decode() and int(a["path"]) == int(b["path"]))
            if not same:
                bad += 1
                if bad <= 5:
                    print("  MISMATCH read", i, dict(zip(a.dtype.names, a.tolist())), mapper.cigar_text(a, pool, L) if a["status"] == 1 else "", "| oracle",
                          {k: (b[k].decode() if k == "cigar" else int(b[k])) for k in ("status", "chrom", "pos", "flag", "mapq", "nm", "score", "path", "n_cand", "n_votes", "cigar")})
        print("set L=%d e=%.2f n=%d: mismatches=%d  gpu %.3fs  oracle %.3fs  stats gpu=%s oracle=%s %s" % (
            L, e, s["n"], bad, dt, odt, st.tolist(), ost.tolist(), "OK" if (bad == 0 and (st == ost).all()) else "FAIL"))
        print("   profile:", ", ".join("%s=%.3fms" % p for p in m.profile()))
        print("   counters gpu:", m.counters(), " oracle:", cnt)
        ok_all &= bad == 0 and bool((st == ost).all())
        m.close()
    print("ALL OK" if ok_all else "FAILED")
    return 0 if ok_all else 1

if __name__ == "__main__":
    sys.exit(main())
