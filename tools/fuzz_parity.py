#!/usr/bin/env python3
"""tools/fuzz_parity.py [--trials N] [--seed S] -- randomised parity soak on the MI355X box: random read length, threshold,
scoring parameters, insert bounds, error rates, qualities, letters outside ACGT, SE / PE fast / PE --sensitive, fixed and mixed
lengths, --ambiguous_out; every trial maps the same reads with the HIP path (through the C ABI) and with the oracle and compares
every record and the stats.  Prints one line per trial and a summary; exit code 1 if any trial differs.  (The committed GPU tests
hold a fixed handful of these trials: tests/test_gpu_parity.py::test_fuzzed_parameters_match_oracle.)"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def draw(rng):
    """one random trial description (plain dict, JSON-able)"""
    mode = ["se", "pe", "pes"][int(rng.integers(0, 3))]
    L = int(rng.choice([int(rng.integers(20, 60)), int(rng.integers(60, 160)), int(rng.integers(160, 301))]))
    if rng.integers(0, 10) == 0:
        L = int(rng.integers(301, 999))            # one trial in ten: up to 998 bases, the longest read the reference itself handles
    mp_max = int(rng.integers(2, 11))
    prm = dict(e_f=float(rng.choice([0.0, 0.02, 0.04, 0.06, 0.08, 0.1, 0.12, 0.15])), mp_max=mp_max, mp_min=int(rng.integers(0, mp_max + 1)),
               np=int(rng.integers(0, 4)), gap_open=int(rng.integers(1, 9)), gap_ext=int(rng.integers(1, 6)), ambiguous_out=int(rng.integers(0, 2)))
    t = dict(mode=mode, L=L, prm=prm, n=int(rng.integers(2000, 9000)) if L <= 300 else int(rng.integers(300, 1200)), seed=int(rng.integers(1, 1 << 30)), sub=float(rng.choice([0.0, 0.005, 0.02, 0.05, 0.08])),
             indel=float(rng.choice([0.0, 0.0005, 0.003])), qual=str(rng.choice(["const", "random"])), conv=float(rng.choice([0.0, 0.5, 0.99])),
             n_rate=float(rng.choice([0.0, 0.0, 0.003, 0.02])), mixed=bool(rng.integers(0, 3) == 0))
    if mode != "se":
        prm["min_ins"] = int(rng.choice([0, 0, 50, 120])); prm["max_ins"] = int(rng.choice([300, 500, 500, 800]))
        prm["sensitive"] = 1 if mode == "pes" else 0
        if L > 300:
            prm["max_ins"] = L + int(rng.choice([200, 500]))
        t["ins_hi"] = int(max(L + 40, prm["max_ins"] + int(rng.integers(-60, 80))))
    return t


def run_trial(t, env):
    """-> list of differences (empty = parity)"""
    from bitmapperbs_amd import synth, mapper
    import orc
    import test_gpu_parity as T
    L, prm = t["L"], t["prm"]
    rng = np.random.default_rng(t["seed"])
    m = mapper.Mapper(env["ix"], 0, **prm)
    bad = []
    if m.max_cigar_ops(L) > 254:
        # penalties under which an alignment may have more CIGAR operations than a record holds: the library refuses the call
        m.close()
        return [], -1
    try:
        if t["mode"] == "se":
            r = synth.make_reads_se(env["chroms"], n=t["n"], L=L, seed=t["seed"], sub=t["sub"], indel=t["indel"], qual=t["qual"], conv=t["conv"], n_rate=t["n_rate"])
            if t["mixed"]:
                lens = rng.integers(max(17, L // 3), L + 1, t["n"]).astype(np.uint16); lens[: t["n"] // 8] = L
                seq, qual = T._trim(r["seq"], lens), T._trim(r["qual"], lens)
                res, pool = m.map_se_var(seq, qual, lens)
                recs, ost, _ = env["oix"].map_se_var(orc.params(**prm), seq, qual, lens)
                bad = T.compare_records(res, pool, recs, lens, amb=bool(prm["ambiguous_out"]))
            else:
                res, pool = m.map_se(r["seq"], r["qual"], L)
                recs, ost, _ = env["oix"].map_se(orc.params(**prm), r["seq"], r["qual"], L)
                bad = T.compare_records(res, pool, recs, L, amb=bool(prm["ambiguous_out"]))
        else:
            m1, m2 = synth.make_reads_pe(env["chroms"], n=t["n"], L=L, seed=t["seed"], sub=t["sub"], indel=t["indel"], qual=t["qual"], conv=t["conv"],
                                         ins_hi=max(t["ins_hi"], L + 30))
            if t["n_rate"]:
                for mm in (m1, m2):
                    pos = rng.random(mm["seq"].shape) < t["n_rate"]
                    mm["seq"][pos] = np.frombuffer(b"NNNRY", dtype=np.uint8)[rng.integers(0, 5, int(pos.sum()))]
            if t["mixed"]:
                l1 = rng.integers(max(17, L // 3), L + 1, t["n"]).astype(np.uint16); l2 = rng.integers(max(17, L // 3), L + 1, t["n"]).astype(np.uint16)
                l1[: t["n"] // 8] = L; l2[: t["n"] // 8] = L
                s1, q1, s2, q2 = T._trim(m1["seq"], l1), T._trim(m1["qual"], l1), T._trim(m2["seq"], l2), T._trim(m2["qual"], l2)
                res, pool = m.map_pe_var(s1, q1, s2, q2, l1, l2)
                recs, ost, _ = env["oix"].map_pe_var(orc.params(**prm), s1, q1, s2, q2, l1, l2)
                bad = T.compare_pe(res, pool, recs, l1, l2)
            else:
                res, pool = m.map_pe(m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
                recs, ost, _ = env["oix"].map_pe(orc.params(**prm), m1["seq"], m1["qual"], m2["seq"], m2["qual"], L)
                bad = T.compare_pe(res, pool, recs, L)
        if not bad and not (m.stats() == ost).all():
            bad = [("stats", m.stats().tolist(), np.asarray(ost).tolist())]
        mapped = int((recs["status"] == 1).sum())
        if not bad:
            # the same batch through the packed entry points (2 bits per base + an N plane), whenever its letters can be packed
            try:
                if t["mode"] == "se":
                    a = (seq, qual, lens) if t["mixed"] else (r["seq"], r["qual"], None)
                    res2, pool2 = m.map_se_packed(mapper.Mapper.pack_rows(a[0], L, a[2]), a[1], L, a[2])
                else:
                    a = (s1, q1, s2, q2, l1, l2) if t["mixed"] else (m1["seq"], m1["qual"], m2["seq"], m2["qual"], None, None)
                    res2, pool2 = m.map_pe_packed(mapper.Mapper.pack_rows(a[0], L, a[4]), mapper.Mapper.pack_rows(a[2], L, a[5]), a[1], a[3], L, a[4], a[5])
                for f in ("status", "chrom", "pos", "flag", "mapq", "nm", "score", "n_cigar", "tlen", "path"):
                    if not (res[f] == res2[f]).all():
                        bad = [("packed entry differs in", f, int(np.nonzero(res[f] != res2[f])[0][0]))]
                        break
                if not bad:
                    for i in np.nonzero(res["n_cigar"] > 0)[0][:20000]:
                        x, y = res[i], res2[i]
                        if not (pool[int(x["cigar_off"]):int(x["cigar_off"]) + int(x["n_cigar"])] == pool2[int(y["cigar_off"]):int(y["cigar_off"]) + int(y["n_cigar"])]).all():
                            bad = [("packed entry differs in the CIGAR of", int(i))]
                            break
            except ValueError:
                pass                                   # letters other than A C G T N: the packed format cannot hold them
    finally:
        m.close()
    return bad, mapped


def make_env(wd):
    from bitmapperbs_amd import synth, mapper
    import orc
    from common import plant_repeats
    names, chroms = synth.make_genome(1_500_000, 3, seed=77)
    plant_repeats(chroms, seed=78)
    fa = os.path.join(wd, "g.fa")
    synth.write_fasta(fa, names, chroms)
    mapper.Index.build(fa, fa, threads=8)
    return dict(fa=fa, chroms=chroms, ix=mapper.Index(fa), oix=orc.OrcIndex(fa))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_parity.json"))
    a = ap.parse_args()
    import orc
    orc.build()
    rng = np.random.default_rng(a.seed)
    with tempfile.TemporaryDirectory() as wd:
        env = make_env(wd)
        fails = []
        for i in range(a.trials):
            t = draw(rng)
            bad, mapped = run_trial(t, env)
            print("trial %3d %-3s L=%3d e=%.2f n=%5d mixed=%d mapped=%5d %s" % (i, t["mode"], t["L"], t["prm"]["e_f"], t["n"], t["mixed"], mapped, ("REFUSED (more than 254 CIGAR operations possible)" if mapped < 0 else "OK") if not bad else "DIFF %s" % (bad[:2],)), flush=True)
            if bad:
                fails.append(dict(trial=t, first=str(bad[:3])))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(trials=a.trials, seed=a.seed, failures=fails), open(a.out, "w"), indent=1)
    print("%d trials, %d with differences" % (a.trials, len(fails)))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
