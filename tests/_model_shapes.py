"""Stage harness for the full-shape model parity tests (BASELINE.json configs 3 / 4 / 5).

ONE piece of harness code drives two implementations through the same public module API:
  * in the build container, `oracle/gen_golden_r2.py` runs it on the REFERENCE's DmxModules (dmx.compressor.modeling.nn, CPU)
    and commits SHA-256 digests of every stage output (tests/golden/model_shapes.json);
  * on the GPU box, `tests/test_gpu_model_shapes.py` runs it on this repo's mirror (dmx_compressor_amd.nn, HIP kernels)
    and compares digests.
Stages are the hot-path operations of each DmxModule of a decoder / encoder layer at the model's TRUE shapes: input casts,
the weight hypernet (`_weight`: sparsifier -> SmoothQuant scale -> storage cast -> weight cast), the bias cast, output
casts.  What sits BETWEEN them in the model (the GEMMs, F.layer_norm ...) is torch's own arithmetic, which differs by
rounding between a CPU and a GPU; every stage therefore gets a counter-generated input (tests/_data.py) of the right
shape instead of the upstream GEMM's output, so that each digest pins exactly one bit-exact operation.

`api` is a small namespace that names the implementation under test:
    api.nn, api.config_rules, api.format, api.QuantHP, api.ModuleQuantHP, api.ModuleSQHP, api.MinMaxObserver
"""
import hashlib
import zlib

import torch

from _data import make

_BITS = {torch.float32: torch.int32, torch.bfloat16: torch.int16, torch.float16: torch.int16, torch.int64: torch.int64,
         torch.int8: torch.int8, torch.uint8: torch.uint8}


def digest(t: torch.Tensor) -> str:
    """SHA-256 of the tensor's bit patterns (contiguous, row-major)"""
    t = t.detach().cpu().contiguous()
    return hashlib.sha256(t.view(_BITS[t.dtype]).numpy().tobytes()).hexdigest()


def stage_input(name: str, shape, dtype=torch.float32, scale: float = 1.0, kind: str = "normal", positive: bool = False):
    """counter-generated input keyed by the stage name (splitmix64 -> Box-Muller: identical bits on every machine)"""
    t = make(kind, tuple(shape), seed=zlib.crc32(name.encode()) & 0x7FFFFFFF, dtype=torch.float32) * scale
    if positive:
        t = t.abs()
    return t.to(dtype)


def _configure(mods, rules):
    for m in mods:
        for r in rules:
            if isinstance(m, r.module_types):
                m.configure(r.module_config)


def _boundary(stages, tag, m, device, inputs, out_shape, dtype, out_scale=4.0):
    """input casts on `inputs` (list of tensors) and the output cast on a synthetic pre-activation"""
    with torch.no_grad():
        cin, a, _ = m.input_casts(inputs[0].to(device), *[t.to(device) for t in inputs[1:]])
        stages.append((f"{tag}/in0", cin.detach().cpu().contiguous()))
        for i, t in enumerate(a):
            stages.append((f"{tag}/in{i + 1}", t.detach().cpu().contiguous()))
        pre = stage_input(f"{tag}/pre", out_shape, dtype, out_scale).to(device)
        stages.append((f"{tag}/out", m.output_casts(pre, output=True).detach().cpu().contiguous()))
        # this repo's mirror only: a binary module that fuses its whole forward (dmxq_binary_cast) == its general path, at this shape
        if hasattr(m, "fuse_binary") and len(inputs) == 2 and inputs[0].shape == inputs[1].shape:
            a, b = inputs[0].to(device), inputs[1].to(device)
            if m._fused_forward(a, b) is not None:
                y_f = m(a, b)
                m.fuse_binary = False
                y_u = m(a, b)
                m.fuse_binary = True
                assert digest(y_f.cpu().contiguous()) == digest(y_u.cpu().contiguous()), f"{tag}: fused binary module"


def _linear_stages(stages, tag, m, device, x, dtype, out_features):
    """the four hot-path stages of a Linear: input cast, weight hypernet, bias cast, output cast"""
    with torch.no_grad():
        xin = x.to(device)
        if getattr(m, "smoothquant", None) is not None:
            xin = m.smoothquant.scale_input(xin)
            stages.append((f"{tag}/sq_in", xin.detach().cpu().contiguous()))
        cin, _, _ = m.input_casts(xin)
        stages.append((f"{tag}/in", cin.detach().cpu().contiguous()))
        # this repo's mirror only: the fused activation path (dmxq_input_hypernet), when the configuration takes it, must give the
        # very tensor whose digest is compared with the reference's
        fused = m._fused_input(x.to(device)) if hasattr(m, "_fused_input") and getattr(m, "smoothquant", None) is not None else None
        if fused is not None:
            assert fused.dtype == cin.dtype and digest(fused.detach().cpu().contiguous()) == digest(cin.detach().cpu().contiguous()), f"{tag}: fused input path"
        stages.append((f"{tag}/w", m._weight.detach().cpu().contiguous()))
        if m.bias is not None:
            stages.append((f"{tag}/b", m._bias.detach().cpu().contiguous()))
        pre = stage_input(f"{tag}/pre", tuple(x.shape[:-1]) + (out_features,), dtype, 4.0).to(device)
        stages.append((f"{tag}/out", m.output_casts(pre, output=True).detach().cpu().contiguous()))


def _new_linear(api, tag, fin, fout, dtype, device, bias=True, w_scale=0.03):
    m = api.nn.Linear(fin, fout, bias=bias)
    m.weight.data = stage_input(f"{tag}/weight", (fout, fin), dtype, w_scale)
    if bias:
        m.bias.data = stage_input(f"{tag}/bias", (fout,), dtype, 0.02)
    return m.to(device)


# ---------------------------------------------------------------------------------------------------- config 3
def opt125m_layer(api, device, scales_out=None, scales_in=None):
    """facebook/opt-125m decoder layer (hidden 768, 12 heads, ffn 3072), fp32, batch 2 x 128 tokens, BASIC rules with
    every Linear weight group-quantised to INT8 (MinMax observer, group_size = 128 rows, ch_axis = 0)."""
    dt, B, S, H, F, NH = torch.float32, 2, 128, 768, 3072, 12
    stages = []
    hp = api.ModuleQuantHP(weight=api.QuantHP(observer_cls=api.MinMaxObserver, qscheme_to_overload=torch.per_tensor_symmetric,
                                               group_size=128, ch_axis=0))
    for name, fin, fout in (("q_proj", H, H), ("k_proj", H, H), ("v_proj", H, H), ("out_proj", H, H), ("fc1", H, F), ("fc2", F, H)):
        tag = f"{name}"
        m = _new_linear(api, tag, fin, fout, dt, device)
        _configure([m], api.config_rules.BASIC)
        m.configure(dict(weight_format=api.format.INT8))
        with m.calibrating_quantizers(hp), torch.no_grad():
            m._weight                                      # one observer pass over the weight (cast.py:179-226)
        stages.append((f"{tag}/w_scale", m.weight_cast.scale.detach().float().cpu().reshape(-1)))
        stages.append((f"{tag}/w_zero_point", m.weight_cast.zero_point.detach().to(torch.int64).cpu().reshape(-1)))
        _linear_stages(stages, tag, m, device, stage_input(f"{tag}/x", (B, S, fin), dt, 2.0, "heavy").clamp(-1e4, 1e4), dt, fout)
    qk, pv, sm, ra, ln, relu = api.nn.ActActMatMul(), api.nn.ActActMatMul(), api.nn.Softmax(dim=-1), api.nn.ResAdd(), api.nn.LayerNorm(H), api.nn.ReLU()
    _configure([qk, pv, sm, ra, ln, relu], api.config_rules.BASIC)
    D = H // NH
    _boundary(stages, "qk_matmul", qk, device, [stage_input("qk/q", (B, NH, S, D), dt, 1.5), stage_input("qk/kT", (B, NH, D, S), dt, 1.5)], (B, NH, S, S), dt)
    _boundary(stages, "softmax", sm, device, [stage_input("sm/x", (B, NH, S, S), dt, 3.0)], (B, NH, S, S), dt, 0.05)
    _boundary(stages, "pv_matmul", pv, device, [stage_input("pv/p", (B, NH, S, S), dt, 0.05, positive=True), stage_input("pv/v", (B, NH, S, D), dt, 1.0)], (B, NH, S, D), dt)
    _boundary(stages, "res_add", ra, device, [stage_input("ra/a", (B, S, H), dt, 2.0), stage_input("ra/b", (B, S, H), dt, 2.0)], (B, S, H), dt)
    _boundary(stages, "layer_norm", ln.to(device), device, [stage_input("ln/x", (B, S, H), dt, 2.0)], (B, S, H), dt, 1.0)
    _boundary(stages, "relu", relu, device, [stage_input("relu/x", (B, S, F), dt, 2.0)], (B, S, F), dt)
    return stages


# ---------------------------------------------------------------------------------------------------- config 4
def llama3_8b_block(api, device, scales_out=None, scales_in=None):
    """Llama-3-8B decoder block (hidden 4096, 32 heads / 8 KV heads of 128, ffn 14336), bf16, 1 x 128 tokens: BASIC rules
    (BFP16_64 activations and weights, FLOAT16 elsewhere) + BTOPK{2:4,-1} weight sparsity on every Linear."""
    dt, B, S, H, F, NH, KV, D = torch.bfloat16, 1, 128, 4096, 14336, 32, 8, 128
    stages = []
    for name, fin, fout in (("q_proj", H, H), ("k_proj", H, KV * D), ("v_proj", H, KV * D), ("o_proj", H, H),
                            ("gate_proj", H, F), ("up_proj", H, F), ("down_proj", F, H)):
        m = _new_linear(api, name, fin, fout, dt, device, bias=False, w_scale=0.02)
        _configure([m], api.config_rules.BASIC)
        m.configure(dict(weight_sparseness="BTOPK{2:4,-1}(U)"))
        with torch.no_grad():
            m.weight_sparsifier(m.weight)       # materialises the lazy score Parameter with the weight's shape
            m.weight_sparsifier.score.data = stage_input(f"{name}/score", (fout, fin), m.weight_sparsifier.score.dtype, 1.0, positive=True).to(device)
        _linear_stages(stages, name, m, device, stage_input(f"{name}/x", (B, S, fin), dt, 2.0, "heavy").clamp(-1e4, 1e4), dt, fout)
        del m
    rms, silu, mul, ra = api.nn.RMSNorm(H, eps=1e-5), api.nn.SiLU(), api.nn.Mul(), api.nn.ResAdd()
    qk, pv, sm = api.nn.ActActMatMul(), api.nn.ActActMatMul(), api.nn.Softmax(dim=-1)
    _configure([rms, silu, mul, ra, qk, pv, sm], api.config_rules.BASIC)
    _boundary(stages, "rms_norm", rms.to(device), device, [stage_input("rms/x", (B, S, H), dt, 2.0)], (B, S, H), dt, 1.0)
    _boundary(stages, "silu", silu, device, [stage_input("silu/x", (B, S, F), dt, 2.0)], (B, S, F), dt, 1.0)
    _boundary(stages, "mul", mul, device, [stage_input("mul/a", (B, S, F), dt, 1.0), stage_input("mul/b", (B, S, F), dt, 1.0)], (B, S, F), dt, 1.0)
    _boundary(stages, "res_add", ra, device, [stage_input("ra/a", (B, S, H), dt, 2.0), stage_input("ra/b", (B, S, H), dt, 2.0)], (B, S, H), dt)
    _boundary(stages, "qk_matmul", qk, device, [stage_input("qk/q", (B, NH, S, D), dt, 1.5), stage_input("qk/kT", (B, NH, D, S), dt, 1.5)], (B, NH, S, S), dt)
    _boundary(stages, "softmax", sm, device, [stage_input("sm/x", (B, NH, S, S), dt, 3.0)], (B, NH, S, S), dt, 0.05)
    _boundary(stages, "pv_matmul", pv, device, [stage_input("pv/p", (B, NH, S, S), dt, 0.05, positive=True), stage_input("pv/v", (B, NH, S, D), dt, 1.0)], (B, NH, S, D), dt)
    return stages


# ---------------------------------------------------------------------------------------------------- config 5
def whisper_small_encoder_layer(api, device, scales_out=None, scales_in=None):
    """openai/whisper-small encoder layer (d_model 768, 12 heads, ffn 3072, 1500 positions), fp32, batch 1: BASIC rules +
    SmoothQuant (migration strength 0.5) calibrated on one batch for every Linear.  The calibrated scale vector is a
    power function of two maxima (smoothquant.py:301-321) and may differ by an ulp between libms, so the reference's
    scale is committed (tests/golden/model_scales.npz) and loaded as a stage INPUT here; the mirror's own scale
    computation is checked against it separately with the tolerance stated in tests/test_gpu_model_shapes.py."""
    dt, B, S, H, F, NH = torch.float32, 1, 1500, 768, 3072, 12
    stages = []
    sq_hp = api.ModuleSQHP(migration_strength=0.5, fuse_to_weight=False)
    for name, fin, fout, bias in (("q_proj", H, H, True), ("k_proj", H, H, False), ("v_proj", H, H, True), ("out_proj", H, H, True),
                                  ("fc1", H, F, True), ("fc2", F, H, True)):
        m = _new_linear(api, name, fin, fout, dt, device, bias=bias)
        _configure([m], api.config_rules.BASIC)
        x = stage_input(f"{name}/x", (B, S, fin), dt, 2.0, "heavy").clamp(-1e3, 1e3)
        # a few outlier channels, the situation SmoothQuant exists for
        x[..., :: max(fin // 12, 1)] *= 20.0
        with m.calibrating_smoothquant(sq_hp), torch.no_grad():
            m(x.to(device))
        own_scale = m.smoothquant.scale.detach().float().cpu().reshape(-1).clone()
        if scales_out is not None:
            scales_out[f"whisper/{name}"] = own_scale
        stages.append((f"{name}/sq_scale~", own_scale))     # "~": compared with a tolerance, not by digest
        if scales_in is not None:
            with torch.no_grad():
                m.smoothquant.scale.data.copy_(scales_in[f"whisper/{name}"].to(m.smoothquant.scale.device, m.smoothquant.scale.dtype).reshape(m.smoothquant.scale.shape))
        _linear_stages(stages, name, m, device, x, dt, fout)
        del m
    qk, pv, sm, ra, ln, gelu = api.nn.ActActMatMul(), api.nn.ActActMatMul(), api.nn.Softmax(dim=-1), api.nn.ResAdd(), api.nn.LayerNorm(H), api.nn.GELU()
    _configure([qk, pv, sm, ra, ln, gelu], api.config_rules.BASIC)
    D = H // NH
    _boundary(stages, "qk_matmul", qk, device, [stage_input("qk/q", (B, NH, S, D), dt, 1.5), stage_input("qk/kT", (B, NH, D, S), dt, 1.5)], (B, NH, S, S), dt)
    _boundary(stages, "softmax", sm, device, [stage_input("sm/x", (B, NH, S, S), dt, 3.0)], (B, NH, S, S), dt, 0.01)
    _boundary(stages, "pv_matmul", pv, device, [stage_input("pv/p", (B, NH, S, S), dt, 0.01, positive=True), stage_input("pv/v", (B, NH, S, D), dt, 1.0)], (B, NH, S, D), dt)
    _boundary(stages, "res_add", ra, device, [stage_input("ra/a", (B, S, H), dt, 2.0), stage_input("ra/b", (B, S, H), dt, 2.0)], (B, S, H), dt)
    _boundary(stages, "layer_norm", ln.to(device), device, [stage_input("ln/x", (B, S, H), dt, 2.0)], (B, S, H), dt, 1.0)
    _boundary(stages, "gelu", gelu, device, [stage_input("gelu/x", (B, S, F), dt, 2.0)], (B, S, F), dt, 1.0)
    return stages


def _wrap(fn):
    def build(nn_mod_or_api, pkg=None, device=torch.device("cpu"), scales_out=None, scales_in=None):
        api = nn_mod_or_api if pkg is None else make_api(nn_mod_or_api, pkg)
        return fn(api, device, scales_out=scales_out, scales_in=scales_in)

    return build


def make_api(nn_mod, pkg):
    """adaptor: (the implementation's nn module, its top-level package) -> the names this harness uses"""
    from types import SimpleNamespace
    hp_home = None
    for cand in ("advanced_recipe", "nn"):
        mod = getattr(pkg, cand, None)
        if mod is not None and hasattr(mod, "DmxQuantizerCalibrationHyperparams"):
            hp_home = mod
            break
    if hp_home is None:  # the reference keeps them in dmx.compressor.advanced_recipe (not imported at package level)
        import importlib
        hp_home = importlib.import_module(pkg.__name__ + ".advanced_recipe")
    obs = getattr(pkg, "MinMaxObserver", None)
    if obs is None:
        import importlib
        obs = importlib.import_module(pkg.__name__ + ".numerical.observer").MinMaxObserver
    return SimpleNamespace(nn=nn_mod, config_rules=pkg.config_rules, format=pkg.format, QuantHP=hp_home.DmxQuantizerCalibrationHyperparams,
                           ModuleQuantHP=hp_home.DmxModuleQuantizerCalibrationHyperparams, ModuleSQHP=hp_home.DmxModuleSmoothQuantHyperparams,
                           MinMaxObserver=obs)


STAGES = {"config3_opt125m_layer": _wrap(opt125m_layer), "config4_llama3_8b_block": _wrap(llama3_8b_block),
          "config5_whisper_small_encoder_layer": _wrap(whisper_small_encoder_layer)}
