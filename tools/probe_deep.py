import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmx_compressor_amd import _lib
L = _lib.lib(); vp = ctypes.c_void_p
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(); sp = vp(stream.cuda_stream)
def run(R, C, nbuf, kind, B=16):
    torch.cuda.empty_cache()
    if kind == "heavy":
        xs = [(torch.randn(R, C, device=dev) * torch.exp(2 * torch.randn(R, C, device=dev))).to(torch.bfloat16) for _ in range(nbuf)]
    else:
        xs = [torch.randn(R, C, device=dev).to(torch.bfloat16) for _ in range(nbuf)]
    ys = [torch.empty_like(x) for x in xs]
    best = 1e9
    with torch.cuda.stream(stream):
        for rep in range(3):
            for i in range(100):
                L.dmxq_bfp_qdq(vp(xs[i % nbuf].data_ptr()), vp(ys[i % nbuf].data_ptr()), _lib.BF16, _lib.BF16, R, C, 1, B, 8, 2, 1, 0, sp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(200):
                L.dmxq_bfp_qdq(vp(xs[i % nbuf].data_ptr()), vp(ys[i % nbuf].data_ptr()), _lib.BF16, _lib.BF16, R, C, 1, B, 8, 2, 1, 0, sp)
            e1.record(stream); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
    print(f"{R}x{C} nbuf {nbuf:2d} {kind:6s} B{B}: {best:6.2f} us  {R*C*4/best/1e3/8000*100:5.1f}%  ptr%2MiB={xs[0].data_ptr() % (2<<20)} {xs[1].data_ptr()-xs[0].data_ptr()}", flush=True)
rows = [int(a) for a in sys.argv[1:]] or [4100, 4352, 4608]
for R in rows:
    for nbuf in ((9, 10, 16) if len(sys.argv) == 1 else (10,)):
        for kind in ("heavy", "normal"):
            run(R, 4096, nbuf, kind)
