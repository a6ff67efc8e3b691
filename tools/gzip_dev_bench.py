#!/usr/bin/env python3
"""tools/gzip_dev_bench.py [file.fq | MB=160] [window MB=64] [lib] -- on the GPU box: an ordinary one-member gzip stream (zlib level 1 of FASTQ
text) through bmbs_inflate_gzip, window by window as a reader would call it; phase times from the library's trace (BMBS_TEXT_TRACE)."""
import ctypes as C
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BMBS_TEXT_TRACE"] = "1"
from bitmapperbs_amd import capi  # noqa: E402
if len(sys.argv) > 3:                       # another build of the library (tools/inflate_prof.sh: cycle counters)
    capi.LIB_PATH = os.path.abspath(sys.argv[3])
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    arg = sys.argv[1] if len(sys.argv) > 1 else "160"
    win = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) << 20
    if arg.isdigit():
        import inflate_bench
        text = inflate_bench.fastq_text(int(arg))
    else:
        with open(arg, "rb") as f:
            text = f.read(160_000_000)
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    data = co.compress(text) + co.flush()
    print("text %.1f MB -> %.1f MB raw deflate (level 1)" % (len(text) / 1e6, len(data) / 1e6), flush=True)
    L = capi.lib()
    params = capi.default_params()
    ctx = L.bmbs_create(0, C.byref(params))
    assert ctx
    a = np.frombuffer(data, dtype=np.uint8)
    for rep in range(2):
        bit = 0; wbytes = b""; out = []; calls = 0
        t0 = time.time()
        while True:
            b0 = bit >> 3
            nbytes = min(len(data) - b0, win)
            eof = b0 + nbytes >= len(data)
            limit = nbytes if eof else nbytes - (1 << 18)
            cap = nbytes * 12
            text_o = np.empty(cap, dtype=np.uint8); wout = np.empty(32768, dtype=np.uint8)
            tb = C.c_uint64(0); eb = C.c_uint64(0); fin = C.c_int32(0); wl = C.c_uint32(0)
            w = np.frombuffer(wbytes, dtype=np.uint8) if wbytes else np.zeros(1, dtype=np.uint8)
            chunk = np.ascontiguousarray(a[b0:b0 + nbytes])
            rc = L.bmbs_inflate_gzip(ctx, capi.ptr(chunk), nbytes, bit & 7, limit, capi.ptr(w), len(wbytes), capi.ptr(text_o), cap, C.byref(tb), C.byref(eb), C.byref(fin), capi.ptr(wout), C.byref(wl))
            assert rc == 0, L.bmbs_last_error(ctx)
            calls += 1
            assert tb.value or fin.value, "no progress at bit %d" % bit
            out.append(text_o[:tb.value].tobytes()); wbytes = wout[:wl.value].tobytes(); bit = b0 * 8 + eb.value
            if fin.value:
                break
        dt = time.time() - t0
        got = b"".join(out)
        print("pass %d: %d calls, %.1f MB in %.3f s (host loop included), identical: %s" % (rep, calls, len(got) / 1e6, dt, got == text), flush=True)
        if hasattr(L, "bmbs_debug_inflate_prof"):
            out = (C.c_uint64 * 24)()
            L.bmbs_debug_inflate_prof(out)
            w = list(out)[:16]
            names = ["header+tables", "window words", "lookups", "chain walk", "fence+refs+jumps", "text+gather+store", "-", "stop tokens"]
            if w[15]:
                print("  span decode: " + "  ".join("%s %.1f%%" % (names[k], 100.0 * w[k] / w[15]) for k in (0, 1, 2, 3, 4, 5, 7)) +
                      "  | windows %d  tokens %d  cycles/window %.0f" % (w[9], w[10], w[15] / max(1, w[9])))
            v = list(out)[16:]
            if v[5]:
                print("  starts: %d span-waves, %.0f steps each, lanes with a candidate per step %.2f, full header checks per step %.2f, cycles per wave %.0f (%.0f in the checks)" %
                      (v[5], v[0] / v[5], v[1] / max(1, v[0]), v[2] / max(1, v[0]), v[3] / v[5], v[4] / v[5]))


main()
