#!/usr/bin/env python3
"""tools/profile_eager_layer.py — WHERE the host time of an eager configured layer goes (cProfile over N forwards, GPU box).
    python tools/profile_eager_layer.py --model opt125m [--n 200] [--top 45]
"""
import argparse
import cProfile
import importlib.util
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="opt125m")
ap.add_argument("--n", type=int, default=200)
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--sort", default="tottime")
a = ap.parse_args()
spec = importlib.util.spec_from_file_location("bench_layer", os.path.join(ROOT, "tools", "bench_layer.py"))
bl = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bl)
dev = torch.device("cuda", 0)
m, x, extra = bl.build_layer(a.model, dev)
with torch.no_grad():
    for _ in range(20):
        m(x, *extra)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.n):
        m(x, *extra)
    torch.cuda.synchronize()
    print(f"{a.model}: {1e6 * (time.perf_counter() - t0) / a.n:.1f} us per eager forward (no profiler)")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.n):
        m(x, *extra)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).strip_dirs().sort_stats(a.sort).print_stats(a.top)
print(s.getvalue().replace("\n\n", "\n"))
