#!/usr/bin/env python3
"""oracle/gen_golden_r2.py — BUILD-CONTAINER ONLY (round 2 additions; oracle/gen_golden.py is unchanged so that its
fixtures stay byte-identical).  Runs the real reference (/root/reference through oracle/ref_shim.py, and its compiled
CPU extension oracle/_ref/quant_cpu.so) and emits

  tests/golden/approx.npz        the reference's one in-repo approximation, experimental.silu
                                 (functional/functions.py:7-21), driven through its own SiLU DmxModule with
                                 approximation_function = "SILU[experimental]{}(scale=...)"; bit patterns.
  tests/golden/native_asym.npz   block_quantize_nearest / _down / _up (a, wl, dim, symmetric=False): the NATIVE asymmetric
                                 branch of quant_cpu.cpp:247-253 from the reference's own C++, on inputs that hit it.
  tests/golden/model_shapes.json SHA-256 digests of per-stage outputs of the reference's DmxModules at the TRUE shapes
                                 of BASELINE.json configs 3 / 4 / 5 (opt-125m decoder layer, Llama-3-8B block, Whisper-small
                                 encoder layer) on counter-generated inputs (tests/_data.py) -- digests, not tensors --
                                 plus tests/golden/model_scales.npz (the SmoothQuant scale vectors, needed as stage inputs
                                 because a `pow` differs by an ulp between libms).

and asserts along the way that this repo's oracle reproduces every reference output bit for bit.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_r2.py [approx] [native] [models]
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "_ref"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
import ref_shim  # noqa: E402
from _data import make  # noqa: E402
from _model_shapes import STAGES, digest, stage_input  # noqa: E402  (shared with the GPU test: same inputs, same hashing)

GOLD = os.path.join(ROOT, "tests", "golden")
ref = ref_shim.load_reference()
from dmx.compressor.modeling import nn as rnn  # noqa: E402

NP_BITS = {torch.float32: np.uint32, torch.bfloat16: np.uint16, torch.float16: np.uint16}
T_BITS = {torch.float32: torch.int32, torch.bfloat16: torch.int16, torch.float16: torch.int16}
DT_NAME = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}
checked = 0


def bits(t):
    t = t.detach().contiguous()
    return t.view(T_BITS[t.dtype]).numpy().view(NP_BITS[t.dtype]).copy()


def check_same(a, b, what):
    global checked
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    af, bf = a.float(), b.float()
    both_nan = (torch.isnan(af) & torch.isnan(bf)).numpy()
    assert bool(((bits(a) == bits(b)) | both_nan).all()), f"ORACLE != REFERENCE: {what}"
    checked += 1


# ------------------------------------------------------------------------------------------------ experimental.silu
def approx_cases():
    store = {}
    special = torch.tensor([0.0, -0.0, 1.0, -1.0, float("inf"), float("-inf"), float("nan"), 65504.0, 70000.0, -70000.0,
                            6e-8, -6e-8, 3e-8, 1e-10, 0.333251953125, 2049.0, 2051.0])
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        x = torch.cat([special, make("heavy", (4096 - special.numel(),), seed=17).clamp(-6e4, 6e4)]).to(dt)
        store[f"x_{DT_NAME[dt]}"] = bits(x)
        for i, scale in enumerate((0.5, 1.0, 0.7310585786300049, 3.0)):
            m = rnn.SiLU()
            m.configure(dict(approximation_function=f"SILU[experimental]{{}}(scale={scale})"))
            with torch.no_grad():
                raw = m._forward(x)            # approx_forward: exact F.silu overwritten by the approximation (fp16)
                y = m(x)                       # through the (SAME) casts and back to the input dtype
            assert raw.dtype == torch.float16, raw.dtype
            store[f"raw_{DT_NAME[dt]}_{i}"], store[f"y_{DT_NAME[dt]}_{i}"] = bits(raw), bits(y)
            store[f"scale_{i}"] = np.array(scale)
    np.savez_compressed(os.path.join(GOLD, "approx.npz"), **store)


# ------------------------------------------------------------------------------------------------ native symmetric = false
def native_cases():
    import quant_cpu  # the reference's own C++ (oracle/Makefile `ref`)

    store, n_hit = {}, 0
    cases = []
    for seed in range(12):
        for shape, dim in (((16, 64), 0), ((8, 32), -1), ((4, 16, 8), 1), ((4, 16, 8), 2)):
            x = make("mixed_nd" if seed % 2 else "normal", shape, seed=200 + seed)
            flat = x.reshape(-1)
            # plant -max elements whose maximum has its top 7 mantissa bits set (1.1111111b * 2^k), some with lower bits too
            big = float(x.abs().max()) * 4.0
            e = int(np.floor(np.log2(big)))
            mx = torch.tensor((2.0 - 2.0 ** -7 + (2.0 ** -12 if seed % 3 == 1 else 0.0)) * 2.0 ** e)
            if dim == 0:
                rows = x.reshape(shape[0], -1)
                for r in range(0, shape[0], 2):
                    rows[r, (seed + r) % rows.shape[1]] = -mx
                    rows[r, (seed + r + 5) % rows.shape[1]] = mx if r % 4 == 0 else rows[r, (seed + r + 5) % rows.shape[1]]
            else:
                flat[seed % flat.numel()] = -mx
            cases.append((x.contiguous(), dim))
    for i, (x, dim) in enumerate(cases):
        store[f"x{i}"], store[f"dim{i}"] = bits(x), np.array(dim)
        for wl in (4, 8, 12):
            for rnd in ("nearest", "down", "up"):
                fn = getattr(quant_cpu, f"block_quantize_{rnd}")
                y_sym, y_nat = fn(x, wl, dim, True), fn(x, wl, dim, False)
                n_hit += int((bits(y_sym) != bits(y_nat)).any())
                check_same(y_nat, O.block_quantize_native(x, wl, dim, False, rnd), f"native asym case {i} wl {wl} {rnd}")
                check_same(y_sym, O.block_quantize_native(x, wl, dim, True, rnd), f"native sym case {i} wl {wl} {rnd}")
                store[f"y{i}_{wl}_{rnd}"] = bits(y_nat)
    store["n"] = np.array(len(cases))
    assert n_hit > 50, f"the native asymmetric branch was exercised by only {n_hit} (case, wl, rounding) combinations"
    print(f"native symmetric=False differs from symmetric=True in {n_hit} of {len(cases) * 9} combinations")
    np.savez_compressed(os.path.join(GOLD, "native_asym.npz"), **store)


# ------------------------------------------------------------------------------------------------ model-shape digests
def model_cases():
    """Per-stage outputs of the reference's DmxModules at true shapes; see tests/_model_shapes.py for the stage list."""
    out, scales = {}, {}
    for cfg_name, build in STAGES.items():
        stages = build(rnn, ref, torch.device("cpu"), scales_out=scales)
        for name, tensor in stages:
            out[f"{cfg_name}/{name}"] = {"sha256": digest(tensor), "shape": list(tensor.shape), "dtype": str(tensor.dtype).replace("torch.", "")}
            print(f"{cfg_name}/{name}: {tuple(tensor.shape)} {tensor.dtype} {out[f'{cfg_name}/{name}']['sha256'][:16]}", flush=True)
    with open(os.path.join(GOLD, "model_shapes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(GOLD, "model_scales.npz"), **{k: bits(v) for k, v in scales.items()})


if __name__ == "__main__":
    what = sys.argv[1:] or ["approx", "native", "models"]
    if "approx" in what:
        approx_cases()
    if "native" in what:
        native_cases()
    if "models" in what:
        model_cases()
    print(f"oracle == reference on {checked} comparisons")
