#!/usr/bin/env python3
"""tools/inflate_bench.py [MB=160 | file.fq] [lib] -- on the GPU box: k_bgzf_inflate alone on FASTQ-like text (bgzip level 1 blocks), kernel
time from the library's own trace (BMBS_TEXT_TRACE).  With a library built with -DINF_PROFILE (tools/inflate_prof.sh) as the second
argument it also prints the cycles per phase."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BMBS_TEXT_TRACE"] = "1"
from bitmapperbs_amd import capi  # noqa: E402

if len(sys.argv) > 2:
    capi.LIB_PATH = os.path.abspath(sys.argv[2])
import struct  # noqa: E402


def fastq_text(mb, L=150, seed=5):
    rng = np.random.default_rng(seed)
    n = mb * 1000000 // (2 * L + 20)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    # qualities in runs, as a binned Illumina run has them
    q = np.frombuffer(b"F:,#", dtype=np.uint8)[np.minimum(3, rng.geometric(0.8, size=(n, L // 5 + 1)) - 1)].repeat(5, axis=1)[:, :L]
    out = bytearray()
    for i in range(n):
        out += b"@sim.%d/1\n" % i; out += seq[i].tobytes(); out += b"\n+\n"; out += q[i].tobytes(); out += b"\n"
    return bytes(out)


def bgzf(data, level=1):
    out = bytearray()
    for a in range(0, len(data), 65280):
        blk = data[a:a + 65280]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        d = c.compress(blk) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + (len(d) + 25).to_bytes(2, "little") + d
        out += zlib.crc32(blk).to_bytes(4, "little") + len(blk).to_bytes(4, "little")
    return bytes(out)


def main():
    arg = sys.argv[1] if len(sys.argv) > 1 else "160"
    if arg.isdigit():
        text = fastq_text(int(arg))
    else:                                        # the first 160 MB of a FASTQ file
        with open(arg, "rb") as f:
            text = f.read(160_000_000)
    comp = bgzf(text)
    print("text %.1f MB -> %.1f MB of BGZF" % (len(text) / 1e6, len(comp) / 1e6), flush=True)
    L = capi.lib()
    params = capi.default_params()
    ctx = L.bmbs_create(0, C.byref(params))
    assert ctx, "no HIP device"
    blk = [0]; out_off = [0]
    at = 0
    while at < len(comp):
        bs = struct.unpack("<H", comp[at + 16:at + 18])[0] + 1
        at += bs; blk.append(at); out_off.append(out_off[-1] + struct.unpack("<I", comp[at - 4:at])[0])
    a = np.frombuffer(comp, dtype=np.uint8); b = np.array(blk, dtype=np.uint64); o = np.array(out_off, dtype=np.uint64)
    got = np.empty(out_off[-1], dtype=np.uint8)

    def inflate():
        rc = L.bmbs_inflate_bgzf(ctx, capi.ptr(a), a.size, capi.ptr(b), capi.ptr(o), len(blk) - 1, capi.ptr(got), out_off[-1], None, 0)
        assert rc == 0, L.bmbs_last_error(ctx)

    for _ in range(4):
        inflate()
    assert got.tobytes() == text
    if len(sys.argv) > 2 and hasattr(L, "bmbs_debug_inflate_prof"):
        out = (C.c_uint64 * 24)()
        L.bmbs_debug_inflate_prof(out)          # (the four calls above)
        inflate()
        L.bmbs_debug_inflate_prof(out)
        v = list(out)
        names = ["header+tables", "window words", "lookups", "chain walk", "fence+refs+jumps", "text+gather+store", "-", "stop tokens", "crc",
                 "windows", "tokens", "serial matches#", "lane matches#", "fences#", "long codes#", "total"]
        tot = v[15]
        for k in (0, 1, 2, 3, 4, 5, 7, 8):
            print("  %-16s %6.1f%%" % (names[k], 100.0 * v[k] / tot))
        print("  windows %d  tokens in them %d  fences %d  tokens by the wave (long codes, long matches) %d  cycles/window %.0f" %
              (v[9], v[10], v[13], v[14], tot / max(1, v[9])))


main()
