"""-m gpu: an activation-function / normalisation DmxModule as ONE launch (include/dmxq.h dmxq_unary_cast,
dmxq_softmax_cast, dmxq_layernorm_cast, dmxq_rmsnorm_cast):  out = cast_out(f(cast_in(x))).

Reference path replaced: modeling/nn/core.py:228-264 (input CastTo -> `_forward` -> output CastTo) for torch_modules.py GELU,
SiLU (:1559-1576), Exp (:236-242), custom_modules.py:112-117 QuickGELU, Softmax (:989-998), LayerNorm (:1062-1082), RMSNorm
(:1144-1170), whose `_forward` is `approx_forward` = the exact torch function with vsimd absent (functional/approximate.py:300-304).

Oracle: the two casts come from oracle/oracle.py (bit-exact restatement of FloatingPoint.cast + CastTo's `.to(dtype)`); the
function is floating point, so the truth is float64 on the cast input and the contract is
    got == cast_out(v)   for a v within N ulps (of the tensor dtype) of the truth,
N = 1 for 16-bit tensors (QuickGELU 2: it is DEFINED with three roundings in the tensor dtype, and a last-place difference in its
sigmoid factor moves the product by up to 2), for float32 tensors the measured fp32 tolerances of the unfused functions
(tests/test_gpu_sparse_calib_approx.py TOL: gelu 2, softmax 8, layer_norm 3, rms_norm 4, silu / exp / quick_gelu 3), or 64 fp32
ulps (2^-17 relative) where the output cast keeps <= 16 mantissa bits and the kernels use the v_exp / v_rcp forms: tests/_data.py
`outside_cast_bracket` states it exactly.  With SAME output casts this is "within N ulp of the truth"; with FLOAT16-style
output casts it is "within 1 ulp of the output cast's format" (VERDICT r2 next-3)."""
import math

import pytest
import torch

from _data import make, outside_cast_bracket

pytestmark = pytest.mark.gpu
F = torch.nn.functional
FMT = {"SAME": None, "FLOAT16": "FP[1|5|10,15](FN)", "BFLOAT16": "FP[1|8|7,127](FN)", "E4M3": "FP[1|4|3,7](_N)", "FP24": "FP[1|8|15,127](FN)",
       "E5M10_noflush": "FP[1|5|10,15](_N)"}


def _fmt(dmx, name):
    return None if FMT[name] is None else dmx.Format.from_shorthand(FMT[name])


def _cpu_cast(oracle, f):
    """CastTo.forward on the CPU through the oracle: dtype -> same dtype"""
    if f is None:
        return lambda x: x.clone()
    return lambda x: oracle.floating_point_cast(x, f.mantissa, f.exponent, f.bias, f.flush_subnormal).to(x.dtype)


def _inputs(shape, dtype, seed, scale=3.0, specials=True):
    x = make("normal", shape, seed=seed) * scale
    if specials:  # saturating, flushed, signed-zero, non-finite values, the FLOAT16 thresholds
        flat = x.reshape(-1)
        sp = [0.0, -0.0, 65504.0, 65520.0, -65536.0, 131008.0, 1e30, -3e38, 6.1e-5, 6.0e-5, -6.2e-5, 1e-30, -1e-40, 88.5, -88.5, 11.0, -11.0,
              float("inf"), float("-inf"), float("nan")]
        k = min(len(sp), flat.numel() // 4)
        flat[torch.arange(k) * 3 + 1] = torch.tensor(sp[:k])
    return x.to(dtype)


def _quick_gelu64(c, dtype):
    """transformers' QuickGELUActivation evaluated in the tensor dtype (three roundings), in float64 up to the LAST one"""
    t1 = (c.double() * float(torch.tensor(1.702, dtype=torch.float32))).to(dtype).double()
    return c.double() * torch.sigmoid(t1).to(dtype).double()


UNARY = {
    "gelu": (lambda c, dt: F.gelu(c.double()), lambda c: c.double().abs() / 2, {torch.float32: 2}),
    "gelu_tanh": (lambda c, dt: F.gelu(c.double(), approximate="tanh"), lambda c: c.double().abs() / 2, {torch.float32: 2}),
    "silu": (lambda c, dt: F.silu(c.double()), None, {torch.float32: 3}),
    "exp": (lambda c, dt: torch.exp(c.double()), None, {torch.float32: 3}),
    "quick_gelu": (_quick_gelu64, None, {torch.float32: 3, torch.bfloat16: 2, torch.float16: 2}),
}


def _n_ulp(table, dtype, fo):
    n = table.get(dtype, 1)
    if dtype == torch.float32 and fo is not None and fo.mantissa <= 16:
        n = 64  # v_exp / v_rcp forms behind an output cast of <= 16 mantissa bits (csrc/act_cast.hip)
    return n


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("func", sorted(UNARY))
def test_unary_cast_contract(dmx, cuda, oracle, dtype, func):
    f64, floor_fn, tol = UNARY[func]
    combos = [("FLOAT16", "FLOAT16"), ("SAME", "SAME"), ("FLOAT16", "SAME"), ("SAME", "FLOAT16"), ("BFLOAT16", "BFLOAT16")]
    if dtype == torch.float32:
        combos += [("E4M3", "E4M3"), ("FLOAT16", "FP24"), ("E5M10_noflush", "E5M10_noflush")]
    for n, (ci, co) in enumerate(combos):
        fi, fo = _fmt(dmx, ci), _fmt(dmx, co)
        x = _inputs((64, 1024), dtype, seed=500 + n)
        got = dmx.ops.unary_cast(x.to(cuda), func, fi, fo)
        if got is None:  # only legal for 16-bit tensors with a cast that is not range-only for the dtype
            assert dtype != torch.float32 and (dtype == torch.float16 and "BFLOAT16" in (ci, co)), (func, dtype, ci, co)
            continue
        assert got.dtype == dtype and got.shape == x.shape
        cin = _cpu_cast(oracle, fi)(x)
        bad = outside_cast_bracket(got, f64(cin, dtype), _cpu_cast(oracle, fo), dtype, _n_ulp(tol, dtype, fo), None if floor_fn is None else floor_fn(cin))
        assert bad == 0, (func, dtype, ci, co, bad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_unary_cast_every_16_bit_pattern(dmx, cuda, oracle, dtype):
    """all 65536 bit patterns of the tensor dtype through FLOAT16 -> f -> FLOAT16 (the BASIC rules), every function"""
    x = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dtype)
    fi = fo = _fmt(dmx, "FLOAT16")
    cin = _cpu_cast(oracle, fi)(x)
    for func, (f64, floor_fn, tol) in sorted(UNARY.items()):
        got = dmx.ops.unary_cast(x.to(cuda), func, fi, fo)
        bad = outside_cast_bracket(got, f64(cin, dtype), _cpu_cast(oracle, fo), dtype, _n_ulp(tol, dtype, fo), None if floor_fn is None else floor_fn(cin))
        assert bad == 0, (func, dtype, bad)


def _ln_truth(c, cols, w, b, eps):
    xd = c.double()
    truth = F.layer_norm(xd, (cols,), None if w is None else w.double(), None if b is None else b.double(), eps)
    mu, rstd = xd.mean(-1, keepdim=True), (xd.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
    floor = (xd.abs().amax(-1, keepdim=True) + mu.abs()) * rstd * (1.0 if w is None else w.double().abs()) + (0.0 if b is None else b.double().abs())
    return truth, floor


def _rms_truth(c, cols, w, eps):
    xd = c.double()
    y = xd * (xd.pow(2).mean(-1, keepdim=True) + eps).rsqrt()
    return y if w is None else y * w.double()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_row_function_cast_contract(dmx, cuda, oracle, dtype):
    """softmax / layer_norm / rms_norm between two casts over row lengths that select every register-resident kernel shape
    (64 / 32 lanes per row, 8-byte vectors, ragged rows of 1500 / 197, workgroup-per-row norms), ragged row counts, masked rows."""
    combos = [("FLOAT16", "FLOAT16"), ("SAME", "SAME"), ("BFLOAT16", "FLOAT16")] + ([("E4M3", "FLOAT16")] if dtype == torch.float32 else [])
    for n, cols in enumerate((8, 64, 256, 768, 1024, 1500, 1536, 2048, 4096, 197, 12 * 256, 16384)):
        rows = (3, 7, 33, 130)[n % 4]
        for ci, co in combos:
            fi, fo = _fmt(dmx, ci), _fmt(dmx, co)
            if dtype == torch.float16 and "BFLOAT16" in (ci, co):
                continue
            cast_o = _cpu_cast(oracle, fo)
            x = _inputs((rows, cols), dtype, seed=700 + n, scale=2.0, specials=False)
            x[0, : cols // 2] = float("-inf")          # an attention-mask row: FLOAT16 turns -inf into -131008 (no Inf codes)
            if cols >= 64:
                x[1, 5] = 70000.0 if dtype != torch.float16 else 65504.0   # saturates in FLOAT16
                x[1, 9] = 1e-6                                          # flushed by FLOAT16
            cin = _cpu_cast(oracle, fi)(x)
            got = dmx.ops.softmax_cast(x.to(cuda), -1, fi, fo)
            if got is not None:
                bad = outside_cast_bracket(got, F.softmax(cin.double(), -1), cast_o, dtype, _n_ulp({torch.float32: 8}, dtype, fo))
                assert bad == 0, ("softmax", dtype, cols, rows, ci, co, bad)
            else:  # rows longer than 1024 lane-vectors are not register resident
                assert cols > 1024 * (4 if dtype == torch.float32 else 8), ("softmax not fused", dtype, cols)
            # norms: finite rows
            xn = (_inputs((rows, cols), dtype, seed=800 + n, scale=2.0, specials=False).float() + 0.5).to(dtype)
            if cols >= 64:
                xn[1, 5] = 70000.0 if dtype != torch.float16 else 65504.0
            cn = _cpu_cast(oracle, fi)(xn)
            w = (make("normal", (cols,), seed=12) * 0.1 + 1).to(dtype)
            b = (make("normal", (cols,), seed=13) * 0.1).to(dtype)
            for ww, bb in ((w, b), (w, None), (None, None)):
                got = dmx.ops.layernorm_cast(xn.to(cuda), (cols,), None if ww is None else ww.to(cuda), None if bb is None else bb.to(cuda), 1e-5, fi, fo)
                if got is None:
                    assert cols % 4 != 0 or cols > 8 * 256 * (4 if dtype == torch.float32 else 8), ("layernorm not fused", dtype, cols)
                    continue
                truth, floor = _ln_truth(cn, cols, ww, bb, 1e-5)
                bad = outside_cast_bracket(got, truth, cast_o, dtype, _n_ulp({torch.float32: 3}, dtype, None), floor)
                assert bad == 0, ("layernorm", dtype, cols, rows, ci, co, ww is not None, bb is not None, bad)
            got = dmx.ops.rmsnorm_cast(xn.to(cuda), (cols,), w.to(cuda), 1e-6, fi, fo)
            if got is not None:
                bad = outside_cast_bracket(got, _rms_truth(cn, cols, w, 1e-6), cast_o, dtype, _n_ulp({torch.float32: 4}, dtype, None))
                assert bad == 0, ("rmsnorm", dtype, cols, rows, ci, co, bad)


def test_fused_casts_equal_the_cast_kernels_where_the_function_is_exact(dmx, cuda, oracle):
    """exp(0) = 1, silu / gelu(0) = 0, softmax of a one-hot-by-mask row: points where f is exact pin the CASTS of the fused kernels
    bit for bit against the oracle (saturation of Inf / NaN, flush, signed zeros) on all three dtypes."""
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        for ci, co in (("FLOAT16", "FLOAT16"), ("SAME", "FLOAT16"), ("FLOAT16", "SAME")):
            fi, fo = _fmt(dmx, ci), _fmt(dmx, co)
            # rows with a single unmasked entry: softmax is exactly 1 there and exactly 0 elsewhere
            x = torch.full((16, 256), float("-inf"), dtype=dtype)
            x[torch.arange(16), torch.arange(16) * 7] = torch.linspace(-5, 5, 16).to(dtype)
            # (FLOAT16 turns -inf into -131008: exp(-131008 - m) underflows to exactly 0 all the same)
            got = dmx.ops.softmax_cast(x.to(cuda), -1, fi, fo).cpu()
            want = torch.zeros((16, 256), dtype=dtype)
            want[torch.arange(16), torch.arange(16) * 7] = 1.0
            assert torch.equal(got.float(), want.float()), (dtype, ci, co)


# ------------------------------------------------------------------------------------------------------------ modules
def _basic(dmx, m):
    for r in dmx.config_rules.BASIC:
        if isinstance(m, r.module_types):
            m.configure(r.module_config)
    return m


MODS = {
    "GELU": (lambda nn: nn.GELU(), "gelu", (64, 3072)),
    "GELU_tanh": (lambda nn: nn.GELU(approximate="tanh"), "gelu_tanh", (64, 3072)),
    "SiLU": (lambda nn: nn.SiLU(), "silu", (64, 14336)),
    "QuickGELU": (lambda nn: nn.QuickGELU(), "quick_gelu", (64, 3072)),
    "Exp": (lambda nn: nn.Exp(), "exp", (64, 1024)),
}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("name", sorted(MODS))
def test_activation_module_runs_the_fused_kernel(dmx, cuda, oracle, dtype, name):
    """BASIC rules (FLOAT16 in, FLOAT16 out): the module's forward IS the one-launch kernel (bit-identical to ops.unary_cast), the
    general three-launch path (fuse_activation = False: this library's casts around torch's GPU function) satisfies the same contract,
    and nothing is fused when autograd needs the graph."""
    ctor, func, shape = MODS[name]
    m = _basic(dmx, ctor(dmx.nn)).to(cuda)
    x = _inputs(shape, dtype, seed=900).to(cuda)
    fi, fo = m.input_casts.input_cast.format, m.output_casts.output_cast.format
    y = m(x)
    assert m._fused_forward(x) is not None, "fused path not taken"
    # 16-bit tensors from lut_min_elems elements up take the module's TABLE (csrc/lut16.hip: float64 evaluation rounded once), smaller
    # ones and float32 the direct kernel
    it = torch.int16 if dtype != torch.float32 else torch.int32
    # default policy: 16-bit tensors through the module's TABLE (csrc/lut16.hip: float64 evaluation rounded once), float32 tensors and
    # lut_activation = False / "auto" below its sizes through the direct kernel
    assert m.lut_activation is True and m._lut_wanted(x, func) == (dtype != torch.float32)
    want = dmx.ops.lut16_apply(x, dmx.ops.unary_cast_table(x, func, fi, fo)) if dtype != torch.float32 else dmx.ops.unary_cast(x, func, fi, fo)
    assert torch.equal(y.view(it), want.view(it))
    for mode in (False, "auto"):       # 64 Ki elements: below "auto"'s sizes
        m.lut_activation = mode
        y_d = m(x)
        assert torch.equal(y_d.view(it), dmx.ops.unary_cast(x, func, fi, fo).view(it))
        f64_, floor_fn_, tol_ = UNARY[func]
        cin_ = _cpu_cast(oracle, fi)(x.cpu())
        assert outside_cast_bracket(y_d, f64_(cin_, dtype), _cpu_cast(oracle, fo), dtype, _n_ulp(tol_, dtype, fo), None if floor_fn_ is None else floor_fn_(cin_)) == 0
    m.lut_activation = True
    assert y.dtype == dtype and y.data_ptr() != x.data_ptr()
    f64, floor_fn, tol = UNARY[func]
    cin = _cpu_cast(oracle, fi)(x.cpu())
    floor = None if floor_fn is None else floor_fn(cin)
    assert outside_cast_bracket(y, f64(cin, dtype), _cpu_cast(oracle, fo), dtype, _n_ulp(tol, dtype, fo), floor) == 0
    m.fuse_activation = False
    y_u = m(x)
    # torch's GPU evaluation of the same function: 1 ulp for 16-bit results; fp32: torch's own accuracy (gelu's erf tail) is its own
    n = max(_n_ulp(tol, dtype, fo), 2 if dtype != torch.float32 else 64)
    assert outside_cast_bracket(y_u, f64(cin, dtype), _cpu_cast(oracle, fo), dtype, n, floor) == 0
    m.fuse_activation = True
    xg = x.clone().requires_grad_(True)
    assert m._fused_forward(xg) is None
    yg = m(xg)
    yg.float().nan_to_num(0.0, 0.0, 0.0).sum().backward()
    assert xg.grad is not None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_softmax_and_norm_modules_run_the_fused_kernels(dmx, cuda, oracle, dtype):
    nn = dmx.nn
    x = _inputs((4, 12, 64, 1500), dtype, seed=910, scale=3.0, specials=False).to(cuda)
    sm = _basic(dmx, nn.Softmax(dim=-1)).to(cuda)
    fi, fo = sm.input_casts.input_cast.format, sm.output_casts.output_cast.format
    y = sm(x)
    assert sm._fused_forward(x) is not None
    it = torch.int16 if dtype != torch.float32 else torch.int32
    assert torch.equal(y.view(it), dmx.ops.softmax_cast(x, -1, fi, fo).view(it))
    cin = _cpu_cast(oracle, fi)(x.cpu())
    assert outside_cast_bracket(y, F.softmax(cin.double(), -1), _cpu_cast(oracle, fo), dtype, _n_ulp({torch.float32: 8}, dtype, fo)) == 0
    sm2 = _basic(dmx, nn.Softmax(dim=1)).to(cuda)        # not the last dim: the general path
    assert sm2._fused_forward(x) is None and sm2(x).shape == x.shape
    for cols, ctor, kind in ((768, lambda: nn.LayerNorm(768), "ln"), (4096, lambda: nn.RMSNorm(4096, eps=1e-5), "rms")):
        m = _basic(dmx, ctor()).to(cuda).to(dtype)
        with torch.no_grad():
            m.weight.copy_((make("normal", (cols,), seed=12) * 0.1 + 1).to(dtype))
            if kind == "ln":
                m.bias.copy_((make("normal", (cols,), seed=13) * 0.1).to(dtype))
        xn = (_inputs((2, 128, cols), dtype, seed=920, scale=2.0, specials=False).float() + 0.25).to(dtype).to(cuda)
        assert m._fused_forward(xn) is None, "grad enabled and the weight requires grad: torch's function builds the graph"
        with torch.no_grad():
            y = m(xn)
            assert m._fused_forward(xn) is not None, kind
        cn = _cpu_cast(oracle, fi)(xn.cpu())
        if kind == "ln":
            truth, floor = _ln_truth(cn, cols, m.weight.detach().cpu(), m.bias.detach().cpu(), m.eps)
            n = _n_ulp({torch.float32: 3}, dtype, None)
        else:
            truth, floor, n = _rms_truth(cn, cols, m.weight.detach().cpu(), m.eps), None, _n_ulp({torch.float32: 4}, dtype, None)
        assert outside_cast_bracket(y, truth, _cpu_cast(oracle, fo), dtype, n, floor) == 0, kind
        m.configure(dict(weight_format="BFP[8|8]{64}(SN)"))      # a real weight cast: raw parameters no longer apply -> general path
        with torch.no_grad():
            assert m._fused_forward(xn) is None and m(xn).shape == xn.shape


def test_dmxq_approximator_is_not_evaluated_twice(dmx, cuda):
    """`FUNC[dmxq]` IS the exact function on this library's kernels: in inference torch's own evaluation is skipped (VERDICT r2
    missing-2), fused or not; with autograd it still runs (it builds the graph)."""
    x = (make("normal", (32, 768), seed=3) * 2).to(torch.bfloat16).to(cuda)
    for name, ctor in (("SOFTMAX", lambda: dmx.nn.Softmax(dim=-1)), ("GELU", lambda: dmx.nn.GELU()), ("SILU", lambda: dmx.nn.SiLU()),
                       ("LAYER_NORM", lambda: dmx.nn.LayerNorm(768)), ("RMS_NORM", lambda: dmx.nn.RMSNorm(768))):
        m = _basic(dmx, ctor()).to(cuda).to(torch.bfloat16)
        m.configure(dict(approximation_function=f"{name}[dmxq]{{}}()"))
        calls = []
        orig = m.functional_forward
        m.functional_forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        with torch.no_grad():
            y_f = m(x)                       # fused: one launch
            assert m._fused_forward(x) is not None, name
            m.fuse_activation = False
            y_u = m(x)                       # three launches, the middle one this library's kernel
        assert not calls, name
        assert (y_f.float() - y_u.float()).abs().max() <= 2.0 ** -6 * y_u.float().abs().max()
        xg = x.clone().requires_grad_(True)
        m(xg)
        assert calls, name
    sm = _basic(dmx, dmx.nn.Softmax(dim=-1)).to(cuda)
    sm.configure(dict(approximation_function="SOFTMAX[dmxq]{input_clamp=-1.0}()"))
    with torch.no_grad():
        y = sm(x)
        assert sm._fused_forward(x) is not None
    c = x.float().clamp(min=-1.0)
    assert (y.float() - torch.softmax(c, -1)).abs().max() < 2e-3     # the wrapper's clamp reaches the fused kernel


# ------------------------------------------------------------------------------------------------------------ model shapes
SHAPES = [
    ("whisper-small GELU [1,1500,3072] fp32", torch.float32, (1, 1500, 3072), "gelu"),
    ("whisper-small softmax [1,12,1500,1500] fp32", torch.float32, (1, 12, 1500, 1500), "softmax"),
    ("whisper-small LayerNorm 768 [1,1500,768] fp32", torch.float32, (1, 1500, 768), "layernorm"),
    ("llama-3-8b SiLU [1,128,14336] bf16", torch.bfloat16, (1, 128, 14336), "silu"),
    ("llama-3-8b RMSNorm 4096 [1,128,4096] bf16", torch.bfloat16, (1, 128, 4096), "rmsnorm"),
    ("llama-3-8b softmax [1,32,128,128] bf16", torch.bfloat16, (1, 32, 128, 128), "softmax"),
    ("opt-125m softmax [2,12,128,128] fp32", torch.float32, (2, 12, 128, 128), "softmax"),
    ("opt-125m LayerNorm 768 [2,128,768] fp32", torch.float32, (2, 128, 768), "layernorm"),
]


@pytest.mark.parametrize("tag,dtype,shape,kind", SHAPES, ids=[s[0] for s in SHAPES])
def test_model_shape_activation_stages(dmx, cuda, oracle, tag, dtype, shape, kind):
    """BASELINE.json configs 3 / 4 / 5: the activation / normalisation modules at the models' TRUE shapes under the BASIC rules,
    through the module API (one fused launch each), against the float64 truth with the stated tolerance (no digest: the function
    is floating point)."""
    nn = dmx.nn
    cols = shape[-1]
    m = {"gelu": lambda: nn.GELU(), "silu": lambda: nn.SiLU(), "softmax": lambda: nn.Softmax(dim=-1), "layernorm": lambda: nn.LayerNorm(cols),
         "rmsnorm": lambda: nn.RMSNorm(cols, eps=1e-5)}[kind]()
    m = _basic(dmx, m).to(cuda).to(dtype)
    if kind in ("layernorm", "rmsnorm"):
        with torch.no_grad():
            m.weight.copy_((make("normal", (cols,), seed=12) * 0.1 + 1).to(dtype))
            if kind == "layernorm":
                m.bias.copy_((make("normal", (cols,), seed=13) * 0.1).to(dtype))
    x = _inputs(shape, dtype, seed=950, scale=3.0 if kind == "softmax" else 2.0, specials=False).to(cuda)
    with torch.no_grad():
        y = m(x)
        assert m._fused_forward(x) is not None, "fused path not taken"
    fi, fo = m.input_casts.input_cast.format, m.output_casts.output_cast.format
    cin = _cpu_cast(oracle, fi)(x.cpu())
    floor, tol = None, {}
    if kind in UNARY:
        f64, floor_fn, tol = UNARY[kind]
        truth, floor = f64(cin, dtype), None if floor_fn is None else floor_fn(cin)
        n = _n_ulp(tol, dtype, fo)
    elif kind == "softmax":
        truth, n = F.softmax(cin.double(), -1), _n_ulp({torch.float32: 8}, dtype, fo)
    elif kind == "layernorm":
        truth, floor = _ln_truth(cin, cols, m.weight.detach().cpu(), m.bias.detach().cpu(), m.eps)
        n = _n_ulp({torch.float32: 3}, dtype, None)
    else:
        truth, n = _rms_truth(cin, cols, m.weight.detach().cpu(), m.eps), _n_ulp({torch.float32: 4}, dtype, None)
    assert outside_cast_bracket(y, truth, _cpu_cast(oracle, fo), dtype, n, floor) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
def test_softmax_cast_then_bfp_equals_the_two_launches(dmx, cuda, dtype):
    """dmxq_softmax_cast_bfp: the consumer's BFP input cast applied in the softmax module's launch == dmxq_softmax_cast followed by
    dmxq_bfp_qdq, bit for bit (both pinned elsewhere: the module contract above, the BFP cast against the reference) -- attention
    shapes with whole and ragged last blocks, 16- and 8-byte lane-vectors, rows shorter than a block, several rows per wave, both
    bindings' front ends; shapes the fused form does not take return None."""
    ops = dmx.ops
    f16 = _fmt(dmx, "FLOAT16")
    shapes = [(12, 100, 1500), (3, 7, 128), (2, 5, 64), (4, 33, 2048), (2, 9, 200), (1, 4, 4096), (5, 32), (2, 3, 24)]
    for shape in shapes:
        x = (make("normal", shape, seed=sum(shape), dtype=torch.float32) * 3).to(dtype).to(cuda)
        for cin, cout in ((f16, f16), (None, f16), (None, None)):
            for B, wl in ((64, 8), (16, 8), (32, 4), (128, 8)):
                fused = ops.softmax_cast(x, -1, cin, cout, then_bfp=(wl, B))
                two = ops.softmax_cast(x, -1, cin, cout)
                if two is None:
                    assert fused is None
                    continue
                want = ops.bfp_qdq(two, wl, B)
                if fused is None:
                    # not fusable: the row is not whole lane-vectors, or B is not a power-of-two number of them
                    epl = 4 if dtype == torch.float32 else 8
                    assert shape[-1] % 4 != 0 or (B // epl) * epl != B or shape[-1] % epl != 0 and (dtype == torch.float32 or B % 4 != 0), (shape, B)
                    continue
                bad = int((fused.view(torch.int32 if dtype == torch.float32 else torch.int16) != want.view(torch.int32 if dtype == torch.float32 else torch.int16)).sum())
                assert bad == 0, f"{shape} {dtype} B={B} wl={wl} casts=({cin}, {cout}): {bad} elements differ"


def test_linked_softmax_feeds_the_matmul_without_a_second_pass(dmx, cuda):
    """nn.link_consumer / the compound ScaledDotProductAttention: the Softmax launch applies the `probs @ value` ActActMatMul's input
    cast, the matmul skips it, results identical to the unlinked modules; reconfiguring the consumer's format is picked up; an active
    dropout or a format the fused kernel does not take fall back to the reference's call sequence."""
    nn = dmx.nn
    torch.manual_seed(0)
    for dtype in (torch.bfloat16, torch.float32):
        sm, pv = nn.Softmax(dim=-1).to(cuda).eval(), nn.ActActMatMul().to(cuda).eval()
        dmx.configure_model(torch.nn.ModuleList([sm, pv]), *dmx.config_rules.BASIC)
        s = (torch.randn(2, 4, 96, 128, device=cuda) * 2).to(dtype)
        v = torch.randn(2, 4, 128, 64, device=cuda).to(dtype)
        with torch.no_grad():
            want = pv(sm(s), v)
            nn.link_consumer(sm, pv)
            p = sm(s)
            assert getattr(p, "_dmx_precast", None) == (pv._first_input_cast(),)
            got = pv(p, v)
            assert torch.equal(got, want)
            pv.configure(dict(input_formats=["BFP[4|8]{32}(SN)", "BFP[8|8]{64}(SN)"]))       # picked up at the next forward
            want2 = pv(_unlinked(sm, s), v)
            assert torch.equal(pv(sm(s), v), want2)
            pv.configure(dict(input_formats=["BFP[8|8]{64}(_N)", "BFP[8|8]{64}(SN)"]))       # asymmetric: not fused, still right
            p = sm(s)
            assert getattr(p, "_dmx_precast", None) is None
            nn.link_consumer(sm, None)
        # the compound module
        att = nn.ScaledDotProductAttention().to(cuda).eval()
        dmx.configure_model(att, *dmx.config_rules.BASIC)
        q, k = (torch.randn(2, 4, 96, 64, device=cuda)).to(dtype), (torch.randn(2, 4, 128, 64, device=cuda)).to(dtype)
        with torch.no_grad():
            got = att(q, k, v)
            att.softmax.fuse_next_cast = False
            want = att(q, k, v)
            att.softmax.fuse_next_cast = True
        assert torch.equal(got, want)


def _unlinked(sm, s):
    sm.fuse_next_cast = False
    try:
        return sm(s)
    finally:
        sm.fuse_next_cast = True


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
def test_norm_cast_then_bfp_equals_the_two_launches(dmx, cuda, dtype):
    """dmxq_layernorm_cast_bfp / dmxq_rmsnorm_cast_bfp == the module's launch followed by dmxq_bfp_qdq, bit for bit: rows on the wave kernel
    (32 / 64 lanes per row, several rows per wave), on the workgroup kernel (4096, 8192), with a ragged last block, with and without
    weight / bias."""
    ops = dmx.ops
    f16 = _fmt(dmx, "FLOAT16")
    int_t = torch.int32 if dtype == torch.float32 else torch.int16
    for rows, cols in ((50, 768), (7, 256), (33, 1024), (5, 4096), (3, 8192), (9, 1504), (4, 64), (2, 3072)):
        x = (make("normal", (rows, cols), seed=rows + cols, dtype=torch.float32) * 2 + 0.3).to(dtype).to(cuda)
        w = (torch.rand(cols, generator=torch.Generator().manual_seed(1)) + 0.5).to(dtype).to(cuda)
        b = (torch.rand(cols, generator=torch.Generator().manual_seed(2)) - 0.5).to(dtype).to(cuda)
        for B, wl in ((64, 8), (16, 8), (128, 6)):
            for name, fused_fn, plain_fn in (
                    ("layernorm", lambda t: ops.layernorm_cast(t, cols, w, b, 1e-5, f16, f16, then_bfp=(wl, B)), lambda t: ops.layernorm_cast(t, cols, w, b, 1e-5, f16, f16)),
                    ("layernorm no affine", lambda t: ops.layernorm_cast(t, cols, None, None, 1e-5, None, f16, then_bfp=(wl, B)), lambda t: ops.layernorm_cast(t, cols, None, None, 1e-5, None, f16)),
                    ("rmsnorm", lambda t: ops.rmsnorm_cast(t, cols, w, 1e-6, f16, f16, then_bfp=(wl, B)), lambda t: ops.rmsnorm_cast(t, cols, w, 1e-6, f16, f16))):
                two = plain_fn(x)
                fused = fused_fn(x)
                if two is None or fused is None:
                    assert fused is None
                    continue
                want = ops.bfp_qdq(two, wl, B)
                bad = int((fused.view(int_t) != want.view(int_t)).sum())
                assert bad == 0, f"{name} {dtype} [{rows}, {cols}] B={B} wl={wl}: {bad} elements differ"


def test_linked_norm_feeds_several_linears_with_one_cast(dmx, cuda):
    """nn.link_consumer(norm, q, k, v): the norm launch applies the three Linears' (identical) BFP input cast, each Linear skips its own;
    outputs identical to the unlinked modules.  A consumer with SmoothQuant scaling switched on, or consumers whose formats differ, are
    not fused (and still right)."""
    nn = dmx.nn
    torch.manual_seed(1)
    for dtype, Norm in ((torch.bfloat16, lambda: nn.RMSNorm(512, eps=1e-5)), (torch.float32, lambda: nn.LayerNorm(512))):
        norm = Norm().to(cuda).to(dtype).eval()
        lins = [nn.Linear(512, 256).to(cuda).to(dtype).eval() for _ in range(3)]
        dmx.configure_model(torch.nn.ModuleList([norm] + lins), *dmx.config_rules.BASIC)
        x = torch.randn(4, 40, 512, device=cuda).to(dtype)
        with torch.no_grad():
            want = [l(norm(x)) for l in lins]
            nn.link_consumer(norm, *lins)
            h = norm(x)
            assert getattr(h, "_dmx_precast", None) == tuple(l._first_input_cast() for l in lins)
            got = [l(h) for l in lins]
            assert all(torch.equal(a, b) for a, b in zip(got, want))
            lins[1].configure(dict(input_formats=["BFP[8|8]{32}(SN)"]))          # formats differ now: no fused cast
            h = norm(x)
            assert getattr(h, "_dmx_precast", None) is None
            lins[1].configure(dict(input_formats=["BFP[8|8]{64}(SN)"]))
            lins[2].smoothquant.scale = (torch.rand(512, device=cuda) + 0.5)
            lins[2].smoothquant.enable()                                          # x / s in front of that cast: no fused cast
            h = norm(x)
            assert getattr(h, "_dmx_precast", None) is None
            lins[2].smoothquant.disable()
            assert getattr(norm(x), "_dmx_precast", None) is not None
            nn.link_consumer(norm)
            assert getattr(norm(x), "_dmx_precast", None) is None


def test_links_derived_from_an_fx_graph(dmx, cuda):
    """nn.link_consumers_from_fx on a traced pre-norm attention block: norm -> q / k / v (one cast), softmax -> dropout (identity) ->
    `p @ v`; a softmax whose probabilities are ALSO returned is left alone; results equal the unlinked model."""
    import torch.fx as fx
    nn = dmx.nn

    class Block(torch.nn.Module):
        def __init__(self, return_probs):
            super().__init__()
            self.norm = nn.LayerNorm(256)
            self.q, self.k, self.v, self.o = (nn.Linear(256, 256) for _ in range(4))
            self.qk, self.pv, self.softmax, self.drop, self.res = nn.ActActMatMul(), nn.ActActMatMul(), nn.Softmax(dim=-1), nn.Dropout(0.1), nn.ResAdd()
            self.return_probs = return_probs

        def forward(self, x):
            h = self.norm(x)
            q, k, v = self.q(h), self.k(h), self.v(h)
            p = self.drop(self.softmax(self.qk(q, k.transpose(-1, -2))))
            y = self.res(self.o(self.pv(p, v)), x)
            return (y, p) if self.return_probs else y

    torch.manual_seed(3)
    x = torch.randn(2, 64, 256, device=cuda)
    for return_probs in (False, True):
        m = Block(return_probs).to(cuda).eval()
        dmx.configure_model(m, *dmx.config_rules.BASIC)
        with torch.no_grad():
            want = m(x)
        gm = fx.GraphModule(m, nn.DmxTracer().trace(m))
        nn.link_consumers_from_fx(gm)
        assert m.norm.__dict__["_next_consumers"] == ((m.q, 0), (m.k, 0), (m.v, 0))
        assert ("_next_consumers" in m.softmax.__dict__) == (not return_probs)
        assert m.qk.__dict__["_next_consumers"] == ((m.softmax, 0),) and m.o.__dict__["_next_consumers"] == ((m.res, 0),)
        assert m.v.__dict__["_next_consumers"] == ((m.pv, 1),) and "_next_consumers" not in m.k.__dict__   # second operand; through a transpose
        assert m.qk._output_cast_absorbed(torch.empty(1, device=cuda)) and not m.q._output_cast_absorbed(torch.empty(1, device=cuda))
        with torch.no_grad():
            got = gm(x)
        if return_probs:
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        else:
            assert torch.equal(got, want)


def test_output_cast_absorbed_by_the_consumers_identical_input_cast(dmx, cuda):
    """A Linear / ActActMatMul whose FLOAT16 output cast is followed by the linked consumer's FLOAT16 input cast skips its own launch:
    same results (the cast is a projection), also for bf16 tensors (where `.to(bfloat16)` follows each cast) and saturating values;
    not while the output cast observes, not when the formats differ."""
    nn = dmx.nn
    torch.manual_seed(5)
    for dtype in (torch.float32, torch.bfloat16):
        lin, act, mm, sm = nn.Linear(128, 256).to(cuda).to(dtype).eval(), nn.GELU().to(cuda).eval(), nn.ActActMatMul().to(cuda).eval(), nn.Softmax(dim=-1).to(cuda).eval()
        dmx.configure_model(torch.nn.ModuleList([lin, act, mm, sm]), *dmx.config_rules.BASIC)
        x = (torch.randn(3, 50, 128, device=cuda) * 40).to(dtype)
        x[0, 0, :4] = torch.tensor([3e5, -3e5, 1e-6, -1e-7], device=cuda).to(dtype)      # beyond FLOAT16's range / below its flush threshold
        lin.weight.data[:4, :4] = torch.eye(4, device=cuda).to(dtype) * 8
        q, k = torch.randn(2, 4, 32, 64, device=cuda).to(dtype) * 30, torch.randn(2, 4, 64, 48, device=cuda).to(dtype) * 30
        with torch.no_grad():
            want_a, want_s = act(lin(x)), sm(mm(q, k))
            nn.link_consumer(lin, act)
            nn.link_consumer(mm, sm)
            assert lin._output_cast_absorbed(x) and mm._output_cast_absorbed(x)
            assert torch.equal(act(lin(x)), want_a) and torch.equal(sm(mm(q, k)), want_s)
            act.configure(dict(input_formats=["FP[1|5|10,15](_N)"]))                       # no flush: a different cast
            assert not lin._output_cast_absorbed(x) and torch.equal(act(lin(x)), act(_unlinked(lin, x)))
            act.configure(dict(input_formats=["FP[1|5|10,15](FN)"]))
            lin.output_casts[next(iter(lin.output_casts.keys()))]._set_flag("observer_enabled", True)
            assert not lin._output_cast_absorbed(x)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
def test_binary_and_relu_cast_then_bfp_equal_the_two_launches(dmx, cuda, dtype):
    """dmxq_binary_cast_bfp / dmxq_relu_cast_bfp: the consumer's BFP input cast in the Mul / ResAdd / ReLU module's launch == the module
    launch followed by dmxq_bfp_qdq, bit for bit -- MLP shapes (whole tiles, a partial last tile, rows shorter than a workgroup's reach),
    casts that round and casts that do not, blocks of 16..128, zero / denormal / huge blocks (the non-fast block path); rows that are
    not whole blocks return None."""
    ops = dmx.ops
    f16, e4m3 = _fmt(dmx, "FLOAT16"), dmx.Format.from_shorthand("FP[1|4|3,7](FN)")
    view = torch.int32 if dtype == torch.float32 else torch.int16
    for shape in [(128, 14336), (256, 3072), (3, 50, 1024), (7, 192), (1, 64), (5, 3, 320), (2, 128)]:
        a = (make("normal", shape, seed=sum(shape), dtype=torch.float32) * 3).to(dtype).to(cuda)
        b = (make("normal", shape, seed=sum(shape) + 1, dtype=torch.float32) * 2).to(dtype).to(cuda)
        a.view(-1)[:64] = 0                                           # an all-zero block
        a.view(-1)[64:128] *= 1e-30 if dtype != torch.float16 else 1e-7   # a denormal-range block
        if shape[-1] >= 256:
            a.view(-1)[128:192] *= 1e30 if dtype != torch.float16 else 6e3
        for casts in ((f16, f16, f16), (None, None, None), (e4m3, f16, None)):
            for B, wl in ((64, 8), (16, 8), (32, 4), (128, 8)):
                takes = shape[-1] % B == 0
                for op in ("add", "mul"):
                    fused = ops.binary_cast(a, b, op, *casts, then_bfp=(wl, B))
                    two = ops.binary_cast(a, b, op, *casts)
                    if two is None or not takes:
                        assert fused is None, (shape, B, op)
                        continue
                    assert fused is not None, (shape, dtype, B, op, casts)
                    want = ops.bfp_qdq(two, wl, B)
                    assert int((fused.view(view) != want.view(view)).sum()) == 0, (shape, dtype, B, wl, op, casts)
                fused, two = ops.relu_cast(a, casts[0], casts[2], then_bfp=(wl, B)), ops.relu_cast(a, casts[0], casts[2])
                if two is None or not takes:
                    assert fused is None
                    continue
                want = ops.bfp_qdq(two, wl, B)
                assert fused is not None and int((fused.view(view) != want.view(view)).sum()) == 0, (shape, dtype, B, wl, casts)


def test_linked_mul_and_relu_feed_a_linear_without_a_second_pass(dmx, cuda):
    """nn.link_consumer(mul, down_proj) / (relu, fc2): the producer's launch applies the Linear's BFP input cast, the Linear skips its own;
    outputs identical to the unlinked modules, fused only while the link is live and the formats allow."""
    nn = dmx.nn
    torch.manual_seed(7)
    for dtype in (torch.bfloat16, torch.float32):
        mul, relu, add = nn.Mul().to(cuda).eval(), nn.ReLU().to(cuda).eval(), nn.ResAdd().to(cuda).eval()
        lin = nn.Linear(1024, 256).to(cuda).to(dtype).eval()
        dmx.configure_model(torch.nn.ModuleList([mul, relu, add, lin]), *dmx.config_rules.BASIC)
        a, b = (torch.randn(2, 60, 1024, device=cuda) * 3).to(dtype), (torch.randn(2, 60, 1024, device=cuda) * 2).to(dtype)
        with torch.no_grad():
            for prod, args in ((mul, (a, b)), (relu, (a,)), (add, (a, b))):
                want = lin(prod(*args))
                nn.link_consumer(prod, lin)
                h = prod(*args)
                assert getattr(h, "_dmx_precast", None) == (lin._first_input_cast(),), type(prod).__name__
                assert torch.equal(lin(h), want)
                h = prod(*args)
                h.mul_(1.7)                                                  # changed in place after the producer returned it: the mark is void
                assert torch.equal(lin(h), lin(h.clone()))
                lin.configure(dict(input_formats=["BFP[8|8]{64}(SS)"]))     # stochastic rounding: not the fused cast
                assert getattr(prod(*args), "_dmx_precast", None) is None
                lin.configure(dict(input_formats=["BFP[8|8]{64}(SN)"]))
                nn.link_consumer(prod)
                assert getattr(prod(*args), "_dmx_precast", None) is None
                assert torch.equal(lin(prod(*args)), want)
