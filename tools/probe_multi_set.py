#!/usr/bin/env python3
"""tools/probe_multi_set.py — where the multi-tensor INT8 launch of ONE opt-125m decoder layer (6 weights + 6 biases, 57 MB of traffic in
float32) stands: the one launch (dmxq_fixed_float_qdq_multi), the same weights without the biases, six launches, ONE tensor of the same
element count through dmxq_fixed_qdq / dmxq_float_qdq, and torch's copy of that tensor -- the ceiling of this SIZE CLASS.
Output: profiles/r06_probe_multi_set.txt."""
import ctypes, sys, math, statistics
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch
from dmx_compressor_amd import _lib
import importlib.util
spec = importlib.util.spec_from_file_location("t2", "tools/bench_tier2.py"); t2 = importlib.util.module_from_spec(spec); spec.loader.exec_module(t2)
dev = torch.device("cuda:0"); L = _lib.lib(); vp = ctypes.c_void_p; F32 = _lib.F32; BF16 = _lib.BF16
T = t2.Timer(dev, 100); sp = T.sp
OPT = t2.OPT_LAYER
def report(name, us, nbytes): print(f"{name:70s} {us:8.2f} us {100*nbytes/(us*1e-6)/8e12:6.1f} %", flush=True)
for dt, code, esz in ((torch.float32, F32, 4), (torch.bfloat16, BF16, 2)):
    k = t2.sets_for(2 * esz * sum(r * c for _, r, c in OPT))
    layers = []
    for s in range(k):
        ws = [(t2.heavy((r, c), 100 * s + i, dev, torch.float32, spread=0.0) * 0.05).to(dt) for i, (_, r, c) in enumerate(OPT)]
        bs = [(t2.heavy((r,), 100 * s + 50 + i, dev, torch.float32, spread=0.0) * 0.02).to(dt) for i, (_, r, c) in enumerate(OPT)]
        scs = [(w.float().reshape(-1, 128, w.shape[1]).abs().amax(dim=(1, 2)) / 127.0).contiguous() for w in ws]
        zps = [torch.zeros(sc.numel(), dtype=torch.int64, device=dev) for sc in scs]
        wo, bo = [torch.empty_like(w) for w in ws], [torch.empty_like(b) for b in bs]
        ad = (_lib.AffineDesc * len(ws))(); fd = (_lib.TensorDesc * len(bs))()
        for d, w, o, sc, zp in zip(ad, ws, wo, scs, zps):
            d.in_, d.out, d.scale, d.zero_point, d.outer, d.C, d.inner = w.data_ptr(), o.data_ptr(), sc.data_ptr(), zp.data_ptr(), 1, w.shape[0], w.shape[1]
        for d, b, o in zip(fd, bs, bo):
            d.in_, d.out, d.outer, d.L, d.inner = b.data_ptr(), o.data_ptr(), 1, b.numel(), 1
        layers.append((ws, bs, scs, zps, wo, bo, ad, fd))
    nel = sum(r * c for _, r, c in OPT)
    nb = 2 * esz * (nel + sum(r for _, r, c in OPT))
    tag = str(dt)[6:]
    us, _ = T.time(lambda i: L.dmxq_fixed_float_qdq_multi(layers[i][6], 6, 8, 0, 1, 1, 2, 128, layers[i][7], 6, 22, 8, 127, 0, 0, 2, code, 0, sp), k)
    report(f"{tag}: fixed_float_qdq_multi 6 weights + 6 biases, one launch", us, nb)
    us, _ = T.time(lambda i: L.dmxq_fixed_qdq_multi(layers[i][6], 6, code, code, 8, 0, 1, 1, 2, 128, 0, sp), k)
    report(f"{tag}: fixed_qdq_multi 6 weights (no biases)", us, 2 * esz * nel)
    def each(i):
        ws, bs, scs, zps, wo, bo, ad, fd = layers[i]
        for w, o, sc, zp in zip(ws, wo, scs, zps):
            L.dmxq_fixed_qdq(vp(w.data_ptr()), vp(o.data_ptr()), code, code, 1, w.shape[0], w.shape[1], 8, 0, 1, 1, 2, vp(sc.data_ptr()), vp(zp.data_ptr()), 128, 0, sp)
    us, _ = T.time(each, k)
    report(f"{tag}: six dmxq_fixed_qdq launches", us, 2 * esz * nel)
    # ONE tensor with the same number of elements: [9216, 768], group 128
    big = [(t2.heavy((9216, 768), 900 + i, dev, torch.float32, spread=0.0) * 0.05).to(dt) for i in range(k)]
    bo_ = [torch.empty_like(b) for b in big]
    sc1 = (big[0].float().reshape(-1, 128, 768).abs().amax(dim=(1, 2)) / 127.0).contiguous(); zp1 = torch.zeros(72, dtype=torch.int64, device=dev)
    us, _ = T.time(lambda i: L.dmxq_fixed_qdq(vp(big[i].data_ptr()), vp(bo_[i].data_ptr()), code, code, 1, 9216, 768, 8, 0, 1, 1, 2, vp(sc1.data_ptr()), vp(zp1.data_ptr()), 128, 0, sp), k)
    report(f"{tag}: ONE dmxq_fixed_qdq on [9216, 768] (the same element count)", us, 2 * esz * 9216 * 768)
    us, _ = T.time(lambda i: L.dmxq_float_qdq(vp(big[i].data_ptr()), vp(bo_[i].data_ptr()), code, code, 9216 * 768, 10, 5, 15, 1, 0, 2, 0, sp), k)
    report(f"{tag}: ONE dmxq_float_qdq FLOAT16 on the same tensor", us, 2 * esz * 9216 * 768)
    us, _ = T.time(lambda i: bo_[i].copy_(big[i]), k)
    report(f"{tag}: torch copy_ of the same tensor", us, 2 * esz * 9216 * 768)
    del layers, big, bo_
