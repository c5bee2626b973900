#!/usr/bin/env python3
"""tools/accuracy_table.py — measured accuracy of the floating-point ops of the approximator slot and the SmoothQuant
scale, per output dtype: maximum distance in ulps OF THE OUTPUT FORMAT from the ground truth
(the function evaluated in float64 on the same inputs and rounded ONCE to the output format), for
  * the HIP kernels of this repo, and
  * torch's own CPU result in the tensor's dtype -- what the reference computes (functional/approximate.py:300-304) --
so that "within 1 ulp of the stated format" is checked against the truth rather than against another rounded
implementation.  Writes a table to stdout (committed as profiles/r03_accuracy_table.txt; round 3 adds the fused modules)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import dmx_compressor_amd as d  # noqa: E402
from _data import err_in_ulps, make, round_once  # noqa: E402

dev = torch.device("cuda:0")


def row(name, dt, got, torch_cpu, truth64, floor=None):
    """errors in ulps of the output format against the float64 truth rounded once (tests/_data.py err_in_ulps; `floor` = the
    magnitude of the cancelling terms for gelu / layer_norm, see there)"""
    a, b = err_in_ulps(got, truth64, dt, floor), err_in_ulps(torch_cpu, truth64, dt, floor)
    print(f"{name:46s} {str(dt).replace('torch.', ''):9s} {a:10.2f} {b:14.2f}{'   (vs cancelling terms)' if floor is not None else ''}", flush=True)


def main():
    print(f"{'op (inputs)':46s} {'dtype':9s} {'HIP max ulp':>10s} {'torch CPU max ulp':>14s}   (vs float64 truth rounded once)")
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        x = (make("normal", (1 << 20,), seed=1) * 4.0).to(dt)
        xd = x.double()
        row("gelu erf, N(0,16)", dt, d.ops.gelu(x.to(dev)), F.gelu(x), F.gelu(xd), floor=xd.abs() / 2)
        row("gelu tanh, N(0,16)", dt, d.ops.gelu(x.to(dev), "tanh"), F.gelu(x, approximate="tanh"), F.gelu(xd, approximate="tanh"), floor=xd.abs() / 2)
        row("silu, N(0,16)", dt, d.ops.silu(x.to(dev)), F.silu(x), F.silu(xd))
        row("exp, N(0,16) clamped to +-10", dt, d.ops.exp(x.clamp(-10, 10).to(dev)), torch.exp(x.clamp(-10, 10)), torch.exp(x.clamp(-10, 10).double()))
        # quick_gelu is DEFINED in the input dtype (three roundings): truth = torch's own chain, evaluated with float64 sigmoid
        t1 = (float(torch.tensor(1.702, dtype=torch.float32)) * xd).to(dt)   # torch multiplies by float32(1.702)
        qg_truth = xd * torch.sigmoid(t1.double()).to(dt).double()
        row("quick_gelu (dtype chain), N(0,16)", dt, d.ops.quick_gelu(x.to(dev)), x * torch.sigmoid(1.702 * x), qg_truth)
        for cols, scale in ((1500, 3.0), (1500, 1.0), (768, 1.0), (4096, 1.0)):
            r = (make("normal", (512, cols), seed=cols) * scale).to(dt)
            row(f"softmax rows of {cols}, N(0,{scale * scale:g})", dt, d.ops.softmax(r.to(dev)), torch.softmax(r, -1), torch.softmax(r.double(), -1))
        for cols in (768, 4096):
            r = (make("normal", (512, cols), seed=cols + 7) * 2.0 + 0.3).to(dt)
            w = (1.0 + 0.1 * make("normal", (cols,), seed=3)).to(dt)
            b = (0.1 * make("normal", (cols,), seed=4)).to(dt)
            rd = r.double()
            mu, rstd = rd.mean(-1, keepdim=True), (rd.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
            row(f"layer_norm rows of {cols}, affine", dt, d.ops.layernorm(r.to(dev), cols, w.to(dev), b.to(dev), 1e-5), F.layer_norm(r, (cols,), w, b, 1e-5),
                F.layer_norm(rd, (cols,), w.double(), b.double(), 1e-5), floor=(rd.abs().amax(-1, keepdim=True) + mu.abs()) * rstd * w.double().abs() + b.double().abs())
            row(f"rms_norm rows of {cols}, weight", dt, d.ops.rmsnorm(r.to(dev), cols, w.to(dev), 1e-6), F.rms_norm(r, (cols,), w, 1e-6),
                F.rms_norm(r.double(), (cols,), w.double(), 1e-6))
    # SmoothQuant scale = clamp(a^alpha / clamp(b, min)^(1 - alpha), min): fp32 vectors (smoothquant.py:301-321)
    a = make("normal", (1 << 16,), seed=5).abs() * 10 + 1e-3
    b = make("normal", (1 << 16,), seed=6).abs() + 1e-3
    for alpha in (0.5, 0.25, 0.8):
        truth = (a.double().pow(alpha) / b.double().clamp(min=1e-5).pow(1 - alpha)).clamp(min=1e-5)
        cpu = (a.pow(alpha) / b.clamp(min=1e-5).pow(1 - alpha)).clamp(min=1e-5)
        row(f"smoothquant scale, alpha = {alpha}", torch.float32, d.ops.smoothquant_scale(a.to(dev), b.to(dev), alpha), cpu, truth)


def fused_modules():
    """Round 3: the activation / normalisation MODULES as one launch (cast_in -> f -> cast_out, BASIC rules: FLOAT16 casts) against
    the float64 truth pushed through the same casts (this library's bit-exact cast kernels), in ulps of the TENSOR dtype (float32 tensors: of the output cast's 10-bit format), next to the
    unfused module (this library's casts around torch's GPU function); and how many elements the two paths disagree on."""
    nn = d.nn
    f16 = d.format.FLOAT16
    cast = lambda t: f16.cast(t, out_dtype=t.dtype)
    print()
    print(f"{'fused module, FLOAT16 casts (inputs)':46s} {'dtype':9s} {'fused max ulp':>13s} {'unfused max ulp':>15s} {'fused != unfused':>17s} {'torch-CPU module max ulp':>25s} {'fused != torch-CPU module':>26s}   (ulps vs cast(round(f64 truth(cast(x)))); 'torch-CPU module' = the bit-exact casts around torch's CPU function: what the reference returns with vsimd absent)")
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        x = (make("normal", (1 << 20,), seed=11) * 4.0).to(dt).to(dev)
        r15 = (make("normal", (512, 1500), seed=12) * 3.0).to(dt).to(dev)
        r768 = (make("normal", (1024, 768), seed=13) * 2.0 + 0.3).to(dt).to(dev)
        r4096 = (make("normal", (256, 4096), seed=14) * 2.0 + 0.3).to(dt).to(dev)
        w768, b768 = (1.0 + 0.1 * make("normal", (768,), seed=3)).to(dt).to(dev), (0.1 * make("normal", (768,), seed=4)).to(dt).to(dev)
        w4096 = (1.0 + 0.1 * make("normal", (4096,), seed=5)).to(dt).to(dev)

        def ln_floor(c, w, b):
            cd = c.double()
            mu, rstd = cd.mean(-1, keepdim=True), (cd.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
            return (cd.abs().amax(-1, keepdim=True) + mu.abs()) * rstd * w.double().abs() + b.double().abs()

        cases = [
            ("GELU, N(0,16)", nn.GELU(), x, lambda c: F.gelu(c.double()), lambda c: c.double().abs() / 2, F.gelu),
            ("SiLU, N(0,16)", nn.SiLU(), x, lambda c: F.silu(c.double()), None, F.silu),
            ("Exp, N(0,16) clamped to +-10", nn.Exp(), x.clamp(-10, 10), lambda c: torch.exp(c.double()), None, torch.exp),
            ("Softmax rows of 1500, N(0,9)", nn.Softmax(dim=-1), r15, lambda c: torch.softmax(c.double(), -1), None, lambda c: torch.softmax(c, -1)),
            ("LayerNorm rows of 768, affine", nn.LayerNorm(768), r768, lambda c: F.layer_norm(c.double(), (768,), w768.double(), b768.double(), 1e-5), lambda c: ln_floor(c, w768, b768),
             lambda c: F.layer_norm(c, (768,), w768.cpu(), b768.cpu(), 1e-5)),
            ("RMSNorm rows of 4096, weight", nn.RMSNorm(4096, eps=1e-5), r4096, lambda c: F.rms_norm(c.double(), (4096,), w4096.double(), 1e-5), None,
             lambda c: F.rms_norm(c, (4096,), w4096.cpu(), 1e-5)),
        ]
        for name, m, inp, f64, floor_fn, cpu_fn in cases:
            m = m.to(dev).to(dt)
            d.configure_model(m, *d.config_rules.BASIC)
            with torch.no_grad():
                if isinstance(m, nn.LayerNorm):
                    m.weight.copy_(w768); m.bias.copy_(b768)
                if isinstance(m, nn.RMSNorm):
                    m.weight.copy_(w4096)
                m.lut_activation = False     # the DIRECT fused kernel first; the table (the default for 16-bit tensors) on its own line below
                fused = m(inp)
                m.fuse_activation = False
                unfused = m(inp)
                c = cast(inp)
                truth = cast(round_once(f64(c).cpu(), dt).to(dev))   # cast_out(round_D(truth)), ONE rounding: the value both paths approximate
            fl = None if floor_fn is None else floor_fn(c)
            unit = dt if dt != torch.float32 else torch.float16   # float32 tensors: ulps of the OUTPUT CAST's format (10 mantissa bits)
            a, b = err_in_ulps(fused, truth.double(), unit, fl), err_in_ulps(unfused, truth.double(), unit, fl)
            diff = float((fused.float() != unfused.float()).float().mean()) * 100
            with torch.no_grad():
                refm = cast(cpu_fn(c.cpu()).to(dt).to(dev))    # the reference's module: bit-exact casts around torch's CPU evaluation
            r_ulp = err_in_ulps(refm, truth.double(), unit, fl)
            rdiff = float((fused.float() != refm.float()).float().mean()) * 100
            print(f"{name:46s} {str(dt).replace('torch.', ''):9s} {a:13.2f} {b:15.2f} {diff:16.3f}% {r_ulp:25.2f} {rdiff:25.3f}%", flush=True)
            if dt != torch.float32 and isinstance(m, (nn.GELU, nn.SiLU, nn.Exp)):
                # the same module through its 65,536-entry table (lut_activation = True: the default)
                m.fuse_activation, m.lut_activation = True, True
                with torch.no_grad():
                    tab = m(inp)
                a2 = err_in_ulps(tab, truth.double(), unit, fl)
                d2 = float((tab.float() != refm.float()).float().mean()) * 100
                print(f"{'  ... as a table (lut_activation = True: default)':46s} {str(dt).replace('torch.', ''):9s} {a2:13.2f} {'':>15s} {'':>17s} {'':>25s} {d2:25.3f}%", flush=True)


if __name__ == "__main__":
    main()
    fused_modules()
