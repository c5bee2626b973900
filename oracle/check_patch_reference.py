#!/usr/bin/env python3
"""oracle/check_patch_reference.py — BUILD-CONTAINER ONLY.  Applies dmx_compressor_amd.integration.patch_reference to the REAL
reference (/root/reference through oracle/ref_shim.py) and checks what can be checked without a GPU:

  1. every attribute the patch reads (integration.SURFACE) exists on the real classes / modules, on instances built from the
     reference's own shorthands;
  2. the patched methods are installed on the real classes and `undo()` restores the originals;
  3. CPU tensors take the ORIGINAL path: results of every patched cast, of Sparsify.forward and of get_module are bit-identical
     before and after patching (the reference's CPU behaviour is untouched);
  4. the argument names of the replaced methods are what the wrappers assume (inspect.signature).

Writes tests/golden/reference_surface.json: attribute NAMES and parameter NAMES of the real classes (interface metadata, no source
text), from which the GPU-box test builds its stand-ins (integration.standins_from_surface).

    PYTHONDONTWRITEBYTECODE=1 python oracle/check_patch_reference.py
"""
import inspect
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_shim  # noqa: E402
from _data import bits_equal, make  # noqa: E402

FORMATS = ["BFP[8|8]{64}(SN)", "BFP[8|8]{16}(_N)", "BFP[24|8]{1}(SN)", "FP[1|5|10,15](FN)", "FP[1|4|3,7](_N)", "FP[1|8|23,127](_N)",
           "XP[8,0](CSN)", "XP[4,0](C_N)", "SBFP<XP[4,0](CSN)><FP[0|4|4,7](FN)>{16}", "MXFP8[E4M3]{32}", "MXINT8{32}"]


def main():
    ref = ref_shim.load_reference()
    import dmx.compressor.numerical.format as rfmt
    import dmx.compressor.quant.quant_function as rqf
    import dmx.compressor.sparse as rsp
    from dmx_compressor_amd import integration as I

    surf = I.surface_of(ref)
    missing = [(c, n) for c, d in surf.items() for n, ok in d.items() if not ok]
    assert not missing, f"attributes the patch reads that the real reference lacks: {missing}"

    x = make("heavy", (8, 128), seed=5)
    xs = {"f32": x, "bf16": x.to(torch.bfloat16), "f16": make("normal", (8, 128), seed=6, dtype=torch.float16)}
    score = make("normal", (8, 128), seed=7).abs()

    def run_all():
        out = {}
        for sh in FORMATS:
            f = rfmt.Format.from_shorthand(sh)
            for dn, t in xs.items():
                for bd in (-1, 0):
                    try:
                        out[(sh, dn, bd)] = f.cast(t.clone(), bd).clone()
                    except RuntimeError as e:   # the reference's own refusals (a 16-bit tensor into fixed_point_quantize): same before / after
                        out[(sh, dn, bd)] = "RuntimeError: " + str(e)[:60]
        for sp in ("BTOPK{2:4,-1}(U)", "BTOPK{4:8,0}(U)", "TOPK{0.5}(U)", "DENSE"):
            m = rsp.Sparsify(score.shape, sparseness=sp)
            with torch.no_grad():
                m.score.copy_(score)
            m.eval()
            out[("sparsify", sp)] = m(x.clone()).clone()
        out["get_module"] = rqf.get_module(x).__name__
        return out

    before = run_all()
    originals = {"bfp": rfmt.BlockFloatingPoint.cast, "fwd": rsp.Sparsify.forward, "gm": rqf.get_module}
    undo = I.patch_reference(ref)
    assert sorted(undo.patched) == sorted(["BlockFloatingPoint.cast", "FloatingPoint.cast", "FixedPoint.cast", "ScaledBlockFloatingPoint.cast",
                                           "MXFP.cast", "Sparsify.forward", "dmx.compressor.quant.quant_function.get_module"]), undo.patched
    assert rfmt.BlockFloatingPoint.cast is not originals["bfp"] and rfmt.BlockFloatingPoint.cast._dmxq_original is originals["bfp"]
    try:
        I.patch_reference(ref)
        raise SystemExit("patching twice must be refused")
    except RuntimeError:
        pass
    after = run_all()
    assert before.keys() == after.keys()
    n = 0
    for k in before:
        if isinstance(before[k], torch.Tensor):
            assert before[k].dtype == after[k].dtype and bits_equal(before[k], after[k]) == 0, k
            n += 1
        else:
            assert before[k] == after[k], k
    undo()
    assert rfmt.BlockFloatingPoint.cast is originals["bfp"] and rsp.Sparsify.forward is originals["fwd"] and rqf.get_module is originals["gm"]

    params = {
        "BlockFloatingPoint.cast": list(inspect.signature(rfmt.BlockFloatingPoint.cast).parameters),
        "FloatingPoint.cast": list(inspect.signature(rfmt.FloatingPoint.cast).parameters),
        "FixedPoint.cast": list(inspect.signature(rfmt.FixedPoint.cast).parameters),
        "ScaledBlockFloatingPoint.cast": list(inspect.signature(rfmt.ScaledBlockFloatingPoint.cast).parameters),
        "MXFP.cast": list(inspect.signature(rfmt.MXFP.cast).parameters),
        "Sparsify.forward": list(inspect.signature(rsp.Sparsify.forward).parameters),
        "quant_function.get_module": list(inspect.signature(rqf.get_module).parameters),
    }
    assert params["BlockFloatingPoint.cast"] == ["self", "x", "block_dim"] and params["Sparsify.forward"] == ["self", "x"]
    assert params["FloatingPoint.cast"][:2] == ["self", "x"] and params["quant_function.get_module"] == ["x"]
    rec = {"source": "oracle/check_patch_reference.py on the real reference (v" + getattr(ref, "__version__", "?") + ")",
           "attributes": {c: sorted(n for n, ok in d.items() if ok) for c, d in surf.items()},
           "parameters": params,
           "cpu_results_identical_before_and_after_patching": n}
    with open(os.path.join(ROOT, "tests", "golden", "reference_surface.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"[patch] ok: {len(undo.patched) or 7} methods patched on the real reference, every attribute of SURFACE present, "
          f"{n} CPU results bit-identical before / after, originals restored by undo()")


if __name__ == "__main__":
    main()
