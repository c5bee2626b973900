#!/usr/bin/env python3
"""tools/sanitize/host_driver.py — every entry point of include/dmxq.h called WITHOUT a GPU, through ctypes, on a library whose host code
is instrumented (tools/sanitize/build_host_asan.py; run with the ASan runtime preloaded and DMXQ_LIB_PATH pointing at it).  No tensor is
ever touched: the "device pointers" are aligned fake addresses that only the (never launched) kernels would dereference.  What runs is the
host side of each call -- argument validation, shape decomposition, plan selection, descriptor packing into the kernel-argument
structs, the gate registry -- over a sweep of shapes, dtypes, block sizes, alignments and tensor counts chosen to reach every size class
and every descriptor-array boundary (kMax tensors per launch, sets that split over several launches, empty and one-element sets).
A call may return OK only when it is a no-op; with work to do it must end in a launch, which fails here (no device) and is reported as a
status -- anything else (a crash, an ASan / UBSan report) is the finding.
"""
import ctypes
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DMXQ_BINDING", "ctypes")
from dmx_compressor_amd import _lib  # noqa: E402

vp = ctypes.c_void_p
L = _lib.lib()
assert "asan" in _lib.LIB_PATH or os.environ.get("DMXQ_DRIVER_ANY_LIB"), f"driver expects the instrumented library, got {_lib.LIB_PATH}"
BF16, F16, F32 = _lib.BF16, _lib.F16, _lib.F32
calls = statuses = 0
seen = {}


def call(name, *a):
    global calls
    rc = getattr(L, name)(*a)
    calls += 1
    seen.setdefault(name, set()).add(rc)
    return rc


def fake(i, align=16):
    """a distinct fake device address (never dereferenced on the host)"""
    return vp(0x7F0000000000 + i * 0x1000000 + (0 if align == 16 else align))


def main():
    s = vp(None)
    shapes = [(1, 16, 1), (4, 4096, 1), (300, 1500, 1), (37, 400, 1), (64, 512, 196), (64, 512, 784), (8, 1500, 64), (2, 3, 50176), (512, 512, 9),
              (1, 4096, 4096), (4096, 64, 49), (1 << 20, 4096, 1), ((1 << 31) + 5, 16, 1), (7, 84, 1), (1, 1, 1), (0, 16, 1)]
    for (outer, Ld, inner), din, dout, B, wl, rnd, sym, al in itertools.product(
            shapes, (BF16, F16, F32), (BF16, F16, F32), (1, 8, 16, 24, 64, 128, 256, 1024), (2, 4, 8, 16, 20, 22, 23), (0, 1, 2, 3), (0, 1), (16, 2)):
        if (din, dout) in ((BF16, F16), (F16, BF16)) or (wl, rnd) in ((2, 0), (23, 3)) and al == 2:
            continue
        if (hash((outer, Ld, inner, din, dout, B, wl, rnd, sym, al)) & 7) != 0:   # one eighth of the product: ~5,000 calls
            continue
        call("dmxq_bfp_qdq", fake(1, al), fake(2, al), din, dout, outer, Ld, inner, B, wl, rnd, sym, 7, s)
        buf = ctypes.create_string_buffer(256)
        call("dmxq_bfp_qdq_describe", din, dout, outer, Ld, inner, B, wl, rnd, sym, 1 if al == 16 else 0, buf, 256)
    # multi-tensor launchers: descriptor arrays of 0 .. 80 tensors of mixed sizes (their kMax boundaries: 20 / 32 per launch)
    for n in (0, 1, 2, 11, 12, 19, 20, 21, 31, 32, 33, 64, 73, 80):
        td = (_lib.TensorDesc * max(n, 1))()
        ad = (_lib.AffineDesc * max(n, 1))()
        hd = (_lib.HypernetDesc * max(n, 1))()
        for i in range(n):
            rows, cols = (768, 768) if i % 3 else (3072, 768 + 8 * (i % 2))
            td[i].in_, td[i].out, td[i].outer, td[i].L, td[i].inner = fake(10 + i).value, fake(200 + i).value, rows, cols, 1
            ad[i].in_, ad[i].out, ad[i].scale, ad[i].zero_point = fake(10 + i).value, fake(200 + i).value, fake(400 + i).value, fake(600 + i).value
            ad[i].outer, ad[i].C, ad[i].inner = 1, rows, cols
            hd[i].w, hd[i].score, hd[i].sq_scale, hd[i].out = fake(10 + i).value, fake(800 + i).value, (fake(900 + i).value if i % 2 else None), fake(200 + i).value
            hd[i].rows, hd[i].L = rows, cols - cols % 64
        for dt in (BF16, F16, F32):
            call("dmxq_bfp_qdq_multi", td, n, dt, dt, 64, 8, 2, 1, 0, s)
            call("dmxq_bfp_qdq_multi", td, n, dt, F32, 16, 8, 3, 1, 5, s)
            call("dmxq_float_qdq_multi", td, n, dt, dt, 10, 5, 15, 1, 0, 2, 0, s)
            for gs in (1, 128, 100):
                call("dmxq_fixed_qdq_multi", ad, n, dt, dt, 8, 0, 1, 1, 2, gs, 0, s)
            call("dmxq_fixed_qdq_multi", ad, n, dt, dt, 8, 2, 0, 0, 3, 128, 9, s)
            for nf in (0, 1, 6, 12, 13):
                call("dmxq_fixed_float_qdq_multi", ad, n, 8, 0, 1, 1, 2, 128, td, min(nf, max(n, 1)), 22, 8, 127, 0, 0, 2, dt, 0, s)
            for M, K in ((0, 0), (4, 2), (8, 4), (3, 1)):
                call("dmxq_weight_hypernet_multi", hd, n, dt, F32 if dt == F32 else dt, K, M, dt, 64, 8, 1, s)
    # the single-tensor entry points, one sweep each over sizes that cross their plan boundaries
    sizes = [0, 1, 7, 8, 4096, (1 << 17) * 8, (3 << 18) * 8 + 8, (5 << 18) * 8, (1 << 21) * 8, (1 << 21) * 8 + 16, (1 << 31) + 24]
    f16 = _lib.FloatFmt(10, 5, 15, 1)
    pf = ctypes.cast(ctypes.pointer(f16), vp)
    for n, dt in itertools.product(sizes, (BF16, F16, F32)):
        call("dmxq_float_qdq", fake(1), fake(2), dt, dt, n, 10, 5, 15, 1, 0, 2, 0, s)
        call("dmxq_float_qdq", fake(1), fake(2), dt, F32, n, 3, 4, 7, 0, 0, 3, 11, s)
        call("dmxq_fixed_qdq", fake(1), fake(2), dt, dt, 1, 1, n, 8, 0, 1, 1, 2, None, None, 1, 0, s)
        call("dmxq_fixed_qdq", fake(1), fake(2), dt, dt, 1, max(n // 4096, 1), 4096, 8, 0, 1, 1, 2, fake(3), fake(4), 128, 0, s)
        call("dmxq_fixed_qdq", fake(1), fake(2), dt, dt, max(n // 4096, 1), 4096, 1, 8, 0, 1, 0, 2, fake(3), fake(4), 1, 0, s)
        call("dmxq_scale_channels", fake(1), fake(2), dt, dt, max(n // 4096, 1), 4096, 1, fake(3), 1, s)
        call("dmxq_scale_channels", fake(1), fake(2), dt, F32, 1, max(n // 768, 1), 768, fake(3), 0, s)
        call("dmxq_gelu", fake(1), fake(2), dt, dt, n, 0, s)
        call("dmxq_unary", fake(1), fake(2), dt, dt, n, 2, ctypes.c_float(0.0), s)
        call("dmxq_unary_cast", fake(1), fake(2), dt, n, 0, ctypes.c_float(0.0), pf, pf, s)
        call("dmxq_relu_cast", fake(1), fake(2), dt, n, pf, pf, s)
        call("dmxq_binary_cast", fake(1), fake(2), fake(3), dt, n, 0, pf, pf, pf, s)
        call("dmxq_nm_mask", fake(1), F32, fake(2), dt, None, 0, fake(3), dt, max(n // 4096, 1), 4096, 1, 2, 4, s)
        call("dmxq_nm_mask", fake(1), dt, None, 0, fake(3), dt, None, 0, 1, 4096, max(n // 4096, 1), 4, 8, s)
        call("dmxq_group_minmax", fake(1), dt, 1, max(n // 4096, 1), 4096, 128, fake(2), fake(3), s)
        call("dmxq_group_minmax_accumulate", fake(1), dt, 1, 1, n, 1, fake(2), fake(3), s)
        call("dmxq_channel_maxabs", fake(1), dt, max(n // 4096, 1), 4096, 1, fake(2), s)
        call("dmxq_histc", fake(1), dt, n, 2048, -4.0, 4.0, fake(2), s)
        call("dmxq_sbfp_qdq", fake(1), fake(2), dt, dt, max(n // 4096, 1), 4096, 1, 16, 4, 1, 1, 4, 4, 7, 1, s)
        call("dmxq_mxfp_qdq", fake(1), fake(2), dt, dt, max(n // 4096, 1), 4096, 1, 32, 3, 4, s)
        call("dmxq_bfp_pack", fake(1), dt, fake(2), fake(3), max(n // 4096, 1), 4096, 16, 8, 1, s)
        call("dmxq_bfp_unpack", fake(1), fake(2), fake(3), dt, max(n // 4096, 1), 4096, 16, 8, s)
        call("dmxq_weight_hypernet", fake(1), dt, fake(2), F32, 2, 4, fake(3), fake(4), dt, max(n // 4096, 1), 4096, 64, 8, 1, s)
        call("dmxq_input_hypernet", fake(1), dt, fake(2), fake(3), F32, max(n // 4096, 1), 4096, 64, 8, 1, s)
        for cols in (8, 64, 197, 768, 1500, 1536, 4096, 16384, 40000):
            rows = max(n // cols, 1) if n else 0
            call("dmxq_softmax", fake(1), fake(2), dt, dt, rows, cols, ctypes.c_float(-100.0), s)
            call("dmxq_softmax_cast", fake(1), fake(2), dt, rows, cols, ctypes.c_float(float("-inf")), pf, pf, s)
            call("dmxq_layernorm", fake(1), fake(2), dt, dt, rows, cols, fake(3), fake(4), dt, ctypes.c_float(1e-5), s)
            call("dmxq_layernorm_cast", fake(1), fake(2), dt, rows, cols, fake(3), fake(4), ctypes.c_float(1e-5), pf, pf, s)
            call("dmxq_rmsnorm", fake(1), fake(2), dt, dt, rows, cols, fake(3), dt, ctypes.c_float(1e-6), s)
            call("dmxq_rmsnorm_cast", fake(1), fake(2), dt, rows, cols, fake(3), ctypes.c_float(1e-6), pf, pf, s)
    n_ok = sum(1 for v in seen.values() if v == {0})
    print(f"host driver: {calls} calls over {len(seen)} entry points of include/dmxq.h on {_lib.LIB_PATH}; status codes seen per entry point:")
    for k in sorted(seen):
        print(f"  {k:34s} {sorted(seen[k])}")
    missing = sorted(set(_lib.SIGNATURES) - set(seen))
    print(f"entry points not driven here ({len(missing)}): {', '.join(missing)}")
    print(f"({n_ok} entry points only ever returned OK)")


if __name__ == "__main__":
    main()
