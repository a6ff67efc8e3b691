"""bmbs_index_build_device (suffix sort + BWT/Occ/SA_flag/16-mer derivation on the GPU) writes the same six files, byte for
byte, as the host builder -- which tests/test_index_build.py pins to the reference's own `--index` output."""
import json
import os

import numpy as np
import pytest

from common import GOLD, gunzip_to, plant_repeats, sha_file

pytestmark = pytest.mark.gpu

SUFFIXES = ("index", "index.bs.pac", "index.bs.index", "index.bs.index.occ", "index.bs.index.bwt", "index.bs.index.sa")


def _both(tmp_path, write):
    from bitmapperbs_amd import mapper
    fa_h = str(tmp_path / "h" / "genome.fa"); fa_d = str(tmp_path / "d" / "genome.fa")
    os.makedirs(os.path.dirname(fa_h)); os.makedirs(os.path.dirname(fa_d))
    write(fa_h); write(fa_d)
    mapper.Index.build(fa_h, fa_h, threads=8)
    mapper.Index.build(fa_d, fa_d, threads=8, device=0)
    for s in SUFFIXES:
        a = open(fa_h + "." + s, "rb").read(); b = open(fa_d + "." + s, "rb").read()
        assert len(a) == len(b), s
        assert a == b, s
    return fa_d


def test_device_builder_matches_reference_index_hashes(tmp_path):
    fa = _both(tmp_path, lambda p: gunzip_to(os.path.join(GOLD, "genome.fa.gz"), p))
    ref = json.load(open(os.path.join(GOLD, "index_ref_sha256.json")))
    for s in SUFFIXES[:-1]:
        assert sha_file(fa + "." + s) == ref[s], s
    assert sha_file(fa + ".index.bs.index.sa", 8) == ref["index.bs.index.sa[:-8]"]


@pytest.mark.parametrize("case", ["repeats", "tandem", "odd_sizes", "lowercase_N"])
def test_device_builder_equals_host_builder(tmp_path, case):
    """repeat-rich texts drive the prefix doubling through many rounds (ties far beyond the 32-symbol radix key); sizes that
    are not multiples of 64 / 128 / 256 / 65536 exercise the tails of the bit-plane, SA_flag and super-block layouts"""
    from bitmapperbs_amd import synth
    rng = np.random.default_rng(11)
    if case == "repeats":
        names, chroms = synth.make_genome(1_500_000, 3, seed=5)
        plant_repeats(chroms, seed=9, spec=((5000, 12, 0.0), (700, 60, 0.01), (150, 300, 0.0), (40, 500, 0.0)))
    elif case == "tandem":
        names, chroms = synth.make_genome(400_000, 2, seed=6)
        unit = synth._ACGT[rng.integers(0, 4, 7)]
        chroms[0][1000:1000 + 7 * 3000] = np.tile(unit, 3000)              # 21 kb tandem repeat: LCPs of thousands
        chroms[1][5000:25000] = ord("A")                                    # 20 kb homopolymer
        chroms[1][50000:60000] = ord("C")
    elif case == "odd_sizes":
        names = ["a", "b", "c"]
        chroms = [synth._ACGT[rng.integers(0, 4, n)] for n in (65536 + 77, 131072 - 3, 4099)]
    else:
        names, chroms = synth.make_genome(300_000, 2, seed=8)
    def write(p):
        synth.write_fasta(p, names, chroms)
        if case == "lowercase_N":
            b = bytearray(open(p, "rb").read())
            for i in range(200, len(b), 977):
                if b[i] in b"ACGT":
                    b[i] = ord("N") if i % 3 else b[i] + 32
            open(p, "wb").write(bytes(b))
    _both(tmp_path, write)
