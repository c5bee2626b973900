"""time series of K=20 eager regions under different end-of-region waits"""
import ctypes, os, sys, time, json, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dmx_compressor_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
N = 20
ins = [torch.randn(4096, 4096, device=dev).to(torch.bfloat16) for _ in range(N)]
outs = [torch.empty_like(t) for t in ins]
stream = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(stream.cuda_stream)
calls = [(ctypes.c_void_p(ins[i].data_ptr()), ctypes.c_void_p(outs[i].data_ptr()), 2, 2, 4096, 4096, 1, 16, 8, 2, 1, 0) for i in range(N)]
f = L.dmxq_bfp_qdq
def run_k(K=20):
    for i in range(K):
        f(*calls[i % N], sp)
hip = ctypes.CDLL("libamdhip64.so")
def region(mode, K=20):
    torch.cuda.synchronize(dev); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_k(K)
    if mode == "poll":
        while not stream.query():
            pass
    elif mode == "hipq":
        while hip.hipStreamQuery(sp) != 0:
            pass
    elif mode == "ssync":
        stream.synchronize()
    elif mode == "hipss":
        hip.hipStreamSynchronize(sp)
    elif mode == "evsync":
        ev.record(stream)
        ev.synchronize()
    elif mode == "evpoll":
        ev.record(stream)
        while not ev.query():
            pass
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e6
res = {}
ev = torch.cuda.Event()
with torch.cuda.stream(stream):
    run_k(5)
    for label in ("default", "spinflag"):
        if label == "spinflag":
            rc = hip.hipSetDeviceFlags(1)   # hipDeviceScheduleSpin
            print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", rc)
        for mode in ("block", "ssync", "hipss", "evsync", "evpoll", "poll", "block", "ssync"):
            ts = [region(mode) for _ in range(60)]
            res[f"{label}/{mode}/{len(res)}"] = [round(t, 1) for t in ts]
    # host-side enqueue time of 20 launches alone (no wait)
    torch.cuda.synchronize(dev)
    hs = []
    for _ in range(30):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); run_k(20); hs.append((time.perf_counter() - t0) * 1e6)
    res["host_enqueue_20"] = [round(t, 1) for t in hs]
    # K sweep: latency = intercept
    for K in (1, 2, 5, 10, 20, 40):
        ts = [region("poll", K) for _ in range(40)]
        res[f"K{K}/poll"] = round(statistics.median(ts), 1)
for k, v in res.items():
    if isinstance(v, list):
        print(k, "med", statistics.median(v), "min", min(v), "first10", v[:10], "last5", v[-5:])
    else:
        print(k, v)
