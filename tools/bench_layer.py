#!/usr/bin/env python3
"""tools/bench_layer.py — one CONFIGURED decoder / encoder layer of BASELINE.json configs 3 / 4 / 5, end to end.

    python bench.py --workload layer --model {opt125m,llama,whisper} [--steps K --warmup W]      (prints one JSON line)
    python tools/bench_layer.py --model llama                                                     (the same, stand-alone)
    rocprofv3 --kernel-trace --stats ... -- python3 bench.py --workload layer --model llama --layer-modes live
    python tools/bench_layer.py --summarise <dir with *_kernel_stats.csv>                          (dmxq / GEMM / other shares)

The layer is built from this library's DmxModules (dmx_compressor_amd.nn: the mirror of modeling/nn/torch_modules.py) with
synthetic weights of the model's true shapes (tests/_model_shapes.py holds the same shapes for the parity tests) and configured
like the BASELINE config: BASIC rules (src/dmx/compressor/__init__.py:306-469) plus
  opt125m  fp32, 2 x 128 tokens: every Linear weight INT8 group-quantised (MinMax, group_size 128 rows), calibrated once;
  llama    bf16, 1 x 128 tokens: BTOPK{2:4,-1} weight sparsity on the 7 Linears, RoPE, grouped-query attention, SiLU gate;
  whisper  fp32, 1 x 1500 positions: SmoothQuant (migration strength 0.5, calibrated on one batch) on the 6 Linears, GELU.
What is timed: one forward of the layer, (a) eager -- Python, dispatcher and launches included, (b) as ONE hipGraph replay
(the GPU time of the same kernels with the host out of the way), each for the weights as configured ("live": the weight chain
re-runs on every forward, as the reference does until fold_weights_and_biases) and folded, and for this library's module fusions
switched off ("unfused": every cast its own launch, the reference's structure).  The GEMMs are torch's (rocBLAS / hipBLASLt).
"""
import argparse
import csv
import glob
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODELS = {
    "opt125m": dict(dtype=torch.float32, B=2, S=128, H=768, F=3072, NH=12, KV=12, act="relu", norm="ln", rope=False, gated=False,
                    desc="facebook/opt-125m decoder layer, fp32, 2 x 128 tokens, BASIC + INT8 group-128 Linear weights (BASELINE.json configs[2])"),
    "llama": dict(dtype=torch.bfloat16, B=1, S=128, H=4096, F=14336, NH=32, KV=8, act="silu", norm="rms", rope=True, gated=True,
                  desc="Llama-3-8B decoder block, bf16, 1 x 128 tokens, BASIC + BTOPK{2:4,-1} weights (BASELINE.json configs[3])"),
    "whisper": dict(dtype=torch.float32, B=1, S=1500, H=768, F=3072, NH=12, KV=12, act="gelu", norm="ln", rope=False, gated=False,
                    desc="openai/whisper-small encoder layer, fp32, 1 x 1500 positions, BASIC + SmoothQuant 0.5 (BASELINE.json configs[4])"),
}


def build_layer(name, dev):
    import dmx_compressor_amd as d
    nn = d.nn
    c = MODELS[name]
    dt, H, F, NH, KV = c["dtype"], c["H"], c["F"], c["NH"], c["KV"]
    D = H // NH

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            norm = (lambda: nn.RMSNorm(H, eps=1e-5)) if c["norm"] == "rms" else (lambda: nn.LayerNorm(H))
            bias = name != "llama"
            self.norm1, self.norm2 = norm(), norm()
            self.q_proj, self.k_proj, self.v_proj = nn.Linear(H, H, bias=bias), nn.Linear(H, KV * D, bias=bias and name != "whisper"), nn.Linear(H, KV * D, bias=bias)
            self.o_proj = nn.Linear(H, H, bias=bias)
            self.qk, self.pv, self.softmax = nn.ActActMatMul(), nn.ActActMatMul(), nn.Softmax(dim=-1)
            self.res1, self.res2 = nn.ResAdd(), nn.ResAdd()
            self.rope = nn.ApplyRotaryPosEmb() if c["rope"] else None
            if c["gated"]:
                self.gate_proj, self.up_proj, self.down_proj = nn.Linear(H, F, bias=False), nn.Linear(H, F, bias=False), nn.Linear(F, H, bias=False)
                self.act, self.mul = nn.SiLU(), nn.Mul()
            else:
                self.fc1, self.fc2 = nn.Linear(H, F), nn.Linear(F, H)
                self.act = nn.GELU() if c["act"] == "gelu" else nn.ReLU()

        def forward(self, x, cos=None, sin=None):
            B, S, _ = x.shape
            h = self.norm1(x)
            q = self.q_proj(h).view(B, S, NH, D).transpose(1, 2)
            k = self.k_proj(h).view(B, S, KV, D).transpose(1, 2)
            v = self.v_proj(h).view(B, S, KV, D).transpose(1, 2)
            if self.rope is not None:
                q, k = self.rope(q.contiguous(), k.contiguous(), cos, sin)
            if KV != NH:
                k = k.repeat_interleave(NH // KV, dim=1)
                v = v.repeat_interleave(NH // KV, dim=1)
            s = self.qk(q * (1.0 / math.sqrt(D)), k.transpose(-1, -2))
            p = self.softmax(s)
            a = self.pv(p, v).transpose(1, 2).reshape(B, S, H)
            x = self.res1(self.o_proj(a), x)
            h = self.norm2(x)
            if c["gated"]:
                f = self.down_proj(self.mul(self.act(self.gate_proj(h)), self.up_proj(h)))
            else:
                f = self.fc2(self.act(self.fc1(h)))
            return self.res2(f, x)

    torch.manual_seed(0)
    m = Layer()
    for p in m.parameters():
        with torch.no_grad():
            p.copy_(torch.randn_like(p) * (0.03 if p.dim() > 1 else 0.02))
    for nm, p in m.named_parameters():
        if nm.endswith("weight") and p.dim() == 1:   # norm gains around 1
            with torch.no_grad():
                p.copy_(torch.randn_like(p) * 0.1 + 1.0)
    m = m.to(dev).to(dt).eval()
    d.configure_model(m, *d.config_rules.BASIC)
    # the probabilities have ONE consumer (no dropout in inference, attention weights not returned): the softmax launch may apply the
    # `p @ v` matmul's input cast too (nn.link_consumer; switched with the other fusions by set_fusions)
    nn.link_consumer(m.softmax, m.pv)
    # ... and the pre-norms feed nothing but the projections (the residual branches off BEFORE the norm)
    nn.link_consumer(m.norm1, m.q_proj, m.k_proj, m.v_proj)
    nn.link_consumer(m.norm2, *((m.gate_proj, m.up_proj) if c["gated"] else (m.fc1,)))
    # GEMM-bearing modules whose result has one consumer that takes it first: their FLOAT16 output cast is the consumer's input cast
    nn.link_consumer(m.qk, m.softmax)
    nn.link_consumer(m.o_proj, m.res1)
    if c["gated"]:
        nn.link_consumer(m.gate_proj, m.act)
        nn.link_consumer(m.up_proj, (m.mul, 1))
        nn.link_consumer(m.down_proj, m.res2)
        nn.link_consumer(m.mul, m.down_proj)   # ... and the gated product / the activation feed one projection: its BFP cast rides along
    else:
        nn.link_consumer(m.fc1, m.act)
        nn.link_consumer(m.fc2, m.res2)
        nn.link_consumer(m.act, m.fc2)
    lin = [mod for mod in m.modules() if isinstance(mod, nn.Linear)]
    x = (torch.randn(c["B"], c["S"], H, device=dev) * 1.5).to(dt)
    extra = ()
    if c["rope"]:
        pos = torch.arange(c["S"], device=dev, dtype=torch.float32)
        inv = 1.0 / (500000.0 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
        ang = torch.cat([pos[:, None] * inv[None, :]] * 2, dim=-1)[None]
        extra = (ang.cos().to(dt), ang.sin().to(dt))
    with torch.no_grad():
        if name == "opt125m":
            hp = nn.DmxModuleQuantizerCalibrationHyperparams(weight=nn.DmxQuantizerCalibrationHyperparams(
                observer_cls=d.MinMaxObserver, qscheme_to_overload=torch.per_tensor_symmetric, group_size=128, ch_axis=0))
            for mod in lin:
                mod.configure(dict(weight_format="XP[8,0](CSN)"))
                with mod.calibrating_quantizers(hp):
                    mod._weight
        elif name == "llama":
            for mod in lin:
                mod.configure(dict(weight_sparseness="BTOPK{2:4,-1}(U)"))
            m(x, *extra)   # materialises the lazy scores
        else:
            hp = nn.DmxModuleSmoothQuantHyperparams(migration_strength=0.5, fuse_to_weight=False)
            for mod in lin:
                mod.enable_smoothquant_calib(True, hp)
            m(x, *extra)
            for mod in lin:
                mod.enable_smoothquant_calib(False, hp)
    return m, x, extra


FUSE_FLAGS = ("fuse_weight_hypernet", "fuse_input_hypernet", "fuse_binary", "fuse_relu", "fuse_rope", "fuse_activation", "fuse_next_cast")


def set_fusions(m, on):
    import dmx_compressor_amd as d
    for mod in m.modules():
        if isinstance(mod, d.nn.DmxModule):
            for f in FUSE_FLAGS:
                if hasattr(mod, f):
                    setattr(mod, f, on)


def time_forward(m, x, extra, steps, warmup, dev):
    """(eager us per forward: wall clock over `steps` back-to-back forwards, synchronised at both ends; graph us per forward: one hipGraph
    of ONE forward replayed `steps` times; launches per forward counted from the captured graph is not available -> None)"""
    with torch.no_grad():
        for _ in range(warmup):
            m(x, *extra)
        torch.cuda.synchronize(dev)
        reps = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(steps):
                m(x, *extra)
            torch.cuda.synchronize(dev)
            reps.append((time.perf_counter() - t0) / steps * 1e6)
        eager = statistics.median(reps)
        graph_us = None
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                m(x, *extra)
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    y = m(x, *extra)
                g.replay()
                torch.cuda.synchronize(dev)
                reps = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    for _ in range(steps):
                        g.replay()
                    e1.record(st)
                    torch.cuda.synchronize(dev)
                    reps.append(e0.elapsed_time(e1) * 1e3 / steps)
                graph_us = statistics.median(reps)
            del g, y
        except Exception as e:  # a capture failure is a finding, not a crash of the bench
            graph_us = f"capture failed: {type(e).__name__}: {str(e)[:120]}"
    return eager, graph_us


def run(name, steps=20, warmup=5, dev=None, modes=("live", "folded", "unfused")):
    import dmx_compressor_amd as d
    dev = dev or torch.device("cuda", 0)
    c = MODELS[name]
    res = {}
    m, x, extra = build_layer(name, dev)
    n_mod = sum(1 for mod in m.modules() if isinstance(mod, d.nn.DmxModule))
    if "live" in modes:
        res["live"] = time_forward(m, x, extra, steps, warmup, dev)
        # the same un-folded weights, their chains batched into one multi-tensor launch per group of sibling weights per forward
        # (nn.LiveWeightBatch, round 5: what GraphedForward installs by default)
        batch = d.nn.LiveWeightBatch(m, replan=False)   # (the layer's configuration is frozen here: the plan of the first forward is kept)
        res["live_batched"] = time_forward(m, x, extra, steps, warmup, dev)
        batch.remove()
    if "unfused" in modes:
        set_fusions(m, False)
        res["unfused"] = time_forward(m, x, extra, steps, warmup, dev)
        set_fusions(m, True)
    if "folded" in modes:
        with torch.no_grad():
            d.nn.fold_weights_and_biases(m)
        res["folded"] = time_forward(m, x, extra, steps, warmup, dev)
        if "unfused" in modes:
            set_fusions(m, False)
            res["folded_unfused"] = time_forward(m, x, extra, steps, warmup, dev)
    head = res.get("live") or next(iter(res.values()))
    line = {
        "metric": f"us per forward of one configured {name} layer (eager, weights re-quantised every forward)",
        "value": round(head[0], 1), "unit": "us", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(head[0] / 1e3, 5),
        "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": str(c["dtype"]).replace("torch.", ""), "data": "synthetic",
        "config": {"workload": c["desc"], "dmx_modules": n_mod, "tokens": c["B"] * c["S"],
                   "timing": "eager: wall clock of `steps` back-to-back forwards / steps (median of 5 regions); graph: HIP events around `steps` "
                             "replays of ONE captured forward / steps (median of 5)"},
        "layer_us": {k: {"eager": round(v[0], 1), "graph": (round(v[1], 1) if isinstance(v[1], float) else v[1])} for k, v in res.items()},
    }
    return line


def summarise(path):
    """dmxq / GEMM / other shares of GPU time from a rocprofv3 --kernel-trace --stats run (the *_kernel_stats.csv under `path`)"""
    files = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        print("no *_kernel_stats.csv under", path)
        return
    tot = {"dmxq": [0.0, 0], "gemm": [0.0, 0], "other": [0.0, 0]}
    rows = []
    for r in csv.DictReader(open(files[0])):
        nm, ns, calls = r["Name"], float(r["TotalDurationNs"]), int(r["Calls"])
        ours = "dmxq::" in nm or any(t in nm for t in ("fused_cast_generic_kernel", "binary_range_bf16_kernel", "float_range_bf16_kernel"))
        k = "dmxq" if ours else ("gemm" if any(t in nm for t in ("Cijk_", "gemm", "Gemm", "GEMM")) else "other")
        tot[k][0] += ns
        tot[k][1] += calls
        rows.append((ns, calls, k, nm))
    s = sum(v[0] for v in tot.values())
    print(f"GPU kernel time by class ({os.path.basename(files[0])}):")
    for k, (ns, calls) in tot.items():
        print(f"  {k:6s} {ns / 1e3:12.1f} us  {100 * ns / s:5.1f} %   {calls} launches")
    print("top kernels:")
    for ns, calls, k, nm in sorted(rows, reverse=True)[:14]:
        print(f"  {ns / 1e3:10.1f} us {calls:6d} x  [{k}] {nm[:110]}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=sorted(MODELS), default="llama")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--modes", default="live,folded,unfused")
    ap.add_argument("--summarise", default=None)
    a = ap.parse_args()
    if a.summarise:
        summarise(a.summarise)
    else:
        print(json.dumps(run(a.model, a.steps, a.warmup, modes=tuple(a.modes.split(",")))), flush=True)
