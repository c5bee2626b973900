"""CPU-side tests (-m "not gpu"): the C-ABI library loads and exports exactly what include/dmxq.h declares,
argument validation never needs a GPU, and the host-side vocabulary (shorthands, aliases) round-trips."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "dmxq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(dmxq_\w+)\s*\(", src))


def test_library_exports_every_declared_symbol(dmx):
    L = dmx._lib.lib()
    declared = _header_functions()
    assert declared, "no declarations parsed from include/dmxq.h"
    for name in declared:
        assert hasattr(L, name), f"libdmxq.so does not export {name}"
    # the ctypes table covers the whole header (minus the two helpers bound separately)
    assert declared - {"dmxq_status_string", "dmxq_abi_version"} == set(dmx._lib.SIGNATURES)
    assert L.dmxq_abi_version() == 4
    assert L.dmxq_status_string(0) == b"ok" and b"bad" in L.dmxq_status_string(1)


def test_argument_validation_without_gpu(dmx):
    """Bad arguments are rejected before anything is launched (no GPU needed); n == 0 is a successful no-op."""
    L = dmx._lib.lib()
    lib = dmx._lib
    null = ctypes.c_void_p(None)
    one = ctypes.c_void_p(16)
    assert L.dmxq_bfp_qdq(null, null, lib.BF16, lib.BF16, 0, 16, 1, 16, 8, lib.ROUND_NEAREST, 1, 0, null) == lib.OK
    assert L.dmxq_bfp_qdq(null, null, lib.BF16, lib.BF16, 4, 16, 1, 16, 8, lib.ROUND_NEAREST, 1, 0, null) == lib.ERR_BAD_ARG
    assert L.dmxq_bfp_qdq(one, one, 7, lib.BF16, 4, 16, 1, 16, 8, lib.ROUND_NEAREST, 1, 0, null) == lib.ERR_BAD_ARG
    assert L.dmxq_bfp_qdq(one, one, lib.BF16, lib.BF16, 4, 16, 1, 0, 8, lib.ROUND_NEAREST, 1, 0, null) == lib.ERR_BAD_ARG
    assert L.dmxq_bfp_qdq(one, one, lib.BF16, lib.BF16, 4, 16, 1, 16, 8, 9, 1, 0, null) == lib.ERR_BAD_ARG
    assert L.dmxq_bfp_qdq(one, one, lib.BF16, lib.BF16, 4, 16, 1, 16, 23, lib.ROUND_NEAREST, 1, 0, null) == lib.ERR_UNSUPPORTED
    assert L.dmxq_float_qdq(one, one, lib.F32, lib.F32, 8, 23, 8, 127, 0, 0, lib.ROUND_NEAREST, 0, null) == lib.ERR_UNSUPPORTED
    assert L.dmxq_float_qdq(one, one, lib.F32, lib.F32, 8, 3, 9, 7, 0, 0, lib.ROUND_NEAREST, 0, null) == lib.ERR_BAD_ARG
    assert L.dmxq_fixed_qdq(one, one, lib.F32, lib.F32, 1, 1, 8, 8, 0, 1, 1, lib.ROUND_NEAREST, one, null, 1, 0, null) == lib.ERR_BAD_ARG
    assert L.dmxq_nm_mask(one, lib.F32, null, 0, one, lib.F32, null, 0, 1, 6, 1, 2, 4, null) == lib.ERR_BAD_ARG  # L % M
    assert L.dmxq_nm_mask(one, lib.F32, null, 0, one, lib.F32, null, 0, 1, 8, 1, 5, 4, null) == lib.ERR_BAD_ARG  # K > M
    assert L.dmxq_nm_mask(one, lib.F32, null, 0, null, 0, null, 0, 1, 8, 1, 2, 4, null) == lib.ERR_BAD_ARG      # no output
    # round 3: a module AND its consumer's BFP input cast in one launch -- the same rules, checked before any launch
    assert L.dmxq_binary_cast_bfp(null, null, null, lib.BF16, 0, 0, null, null, null, 64, 64, 8, null) == lib.OK            # n == 0
    assert L.dmxq_binary_cast_bfp(null, null, null, lib.BF16, 64, 0, null, null, null, 64, 64, 8, null) == lib.ERR_BAD_ARG   # null tensors
    assert L.dmxq_binary_cast_bfp(one, one, one, lib.BF16, 64, 2, null, null, null, 64, 64, 8, null) == lib.ERR_BAD_ARG      # op
    assert L.dmxq_binary_cast_bfp(one, one, one, 9, 64, 0, null, null, null, 64, 64, 8, null) == lib.ERR_BAD_ARG            # dtype
    assert L.dmxq_binary_cast_bfp(one, one, one, lib.BF16, 64, 0, null, null, null, 64, 0, 8, null) == lib.ERR_BAD_ARG       # block size
    assert L.dmxq_binary_cast_bfp(one, one, one, lib.BF16, 128, 0, null, null, null, 96, 64, 8, null) == lib.ERR_UNSUPPORTED # rows of 1.5 blocks
    assert L.dmxq_binary_cast_bfp(one, one, one, lib.BF16, 128, 0, null, null, null, 64, 24, 8, null) == lib.ERR_UNSUPPORTED # not 2^k lane-vectors
    assert L.dmxq_binary_cast_bfp(one, one, one, lib.BF16, 128, 0, null, null, null, 64, 64, 40, null) == lib.ERR_UNSUPPORTED # precision
    assert L.dmxq_relu_cast_bfp(null, null, lib.F32, 0, null, null, 64, 64, 8, null) == lib.OK
    assert L.dmxq_relu_cast_bfp(null, one, lib.F32, 64, null, null, 64, 64, 8, null) == lib.ERR_BAD_ARG
    assert L.dmxq_relu_cast_bfp(one, one, lib.F32, 128, null, null, 96, 64, 8, null) == lib.ERR_UNSUPPORTED
    assert L.dmxq_relu_cast_bfp(one, one, lib.F32, 1024, null, null, 512, 512, 8, null) == lib.ERR_UNSUPPORTED               # a block wider than a wave


def test_no_cpu_fallback(dmx):
    x = torch.randn(4, 16)
    for call in (lambda: dmx.ops.bfp_qdq(x, 8, 16), lambda: dmx.ops.float_qdq(x, 10, 5, 15, True),
                 lambda: dmx.ops.fixed_qdq(x, 8, 0), lambda: dmx.ops.nm_mask(x, 2, 4),
                 lambda: dmx.CastTo(format="BFP[8|8]{16}(SN)")(x), lambda: dmx.ops.gelu(x),
                 lambda: dmx.ops.input_hypernet(x, torch.ones(16), 8, 16), lambda: dmx.ops.binary_cast(x, x, "add"),
                 lambda: dmx.ops.rope_cast(x.view(1, 1, 4, 16), torch.ones(1, 4, 16), torch.ones(1, 4, 16)),
                 lambda: dmx.ops.weight_hypernet(x, 8, 16)):
        with pytest.raises(dmx.DmxqError):
            call()
    # the fused module paths never catch that: a CPU tensor goes to the general path and fails there, loudly
    m = dmx.nn.ResAdd()
    m.configure(dict(input_formats=["FP[1|5|10,15](FN)"] * 2, output_formats=["FP[1|5|10,15](FN)"]))
    with pytest.raises(dmx.DmxqError):
        m(x.to(torch.bfloat16), x.to(torch.bfloat16))


SHORTHANDS = ["SAME", "XP[8,0](CSN)", "XP[4,0](CSN)", "XP[8,+4](C_N)", "XP[16,-2](_SS)", "FP[1|5|10,15](FN)",
              "FP[1|8|7,127](FN)", "FP[1|4|3,7](_N)", "FP[0|4|4,7](FN)", "FP[1|5|2,15](_S)", "BFP[8|8]{16}(SN)",
              "BFP[8|8]{64}(_N)", "BFP[24|8]{1}(SN)", "BFP[4|8]{128}(SU)", "SBFP<XP[4,0](CSN)><FP[0|4|4,7](FN)>{16}",
              "MXFP8[E4M3]{32}", "MXFP4[E2M1]{128}", "MXINT8{32}", "MXINT4{64}"]


@pytest.mark.parametrize("sh", SHORTHANDS)
def test_format_shorthand_roundtrip(dmx, sh):
    f = dmx.Format.from_shorthand(sh)
    assert repr(f) == sh
    assert repr(dmx.Format.from_shorthand(repr(f))) == sh
    str(f)


def test_format_validation_and_properties(dmx):
    with pytest.raises(ValueError):
        dmx.Format.from_shorthand("BFP[8|8]{16,1}(SN)")          # legacy form is not the current grammar
    fmt, dim = dmx.BlockFloatingPoint.parse_legacy("BFP[8|8]{64,1}(SN)")
    assert repr(fmt) == "BFP[8|8]{64}(SN)" and dim == 1
    with pytest.raises(ValueError):
        dmx.Format.from_shorthand("INT8")
    with pytest.raises(AssertionError):
        dmx.FixedPoint(25, 0)
    with pytest.raises(AssertionError):
        dmx.FloatingPoint(mantissa=3, exponent=4, bias=-200)
    with pytest.raises(AssertionError):
        dmx.BlockFloatingPoint(precision=1)
    assert dmx.format.BFP16_16.bytes_per_elem == (8 + 8 / 16) / 8
    assert dmx.format.BFP16_16.bit_precision == 8.5
    assert dmx.format.INT4.bytes_per_elem == 0.5
    assert dmx.format.FLOAT16.bit_precision == 16.0
    assert dmx.format.AFLOAT8.largest_representable_power_of_two == 256
    assert dmx.format.SBFP12_16.man_scaling == 7
    assert repr(dmx.format.BFP16A_16) == "BFP[6|8]{16}(_N)"     # reference alias quirk kept
    assert repr(dmx.format.MXINT8_K32) == "MXINT8{32}" and isinstance(dmx.format.MXINT8_K32, dmx.BlockFloatingPoint)
    assert len(vars(dmx.format)) == 76


def test_sparseness_shorthands(dmx):
    for sh in ["DENSE", "TOPK{0.5}(U)", "BTOPK{2:4,-1}(U)", "BTOPK{4:8,1}(M)", "BERN"]:
        assert repr(dmx.Sparseness.from_shorthand(sh)) == sh
    assert dmx.sparseness.BTK8_2_LD.density == 0.25 and dmx.sparseness.BTK8_4_FD.block_dim == 1
    with pytest.raises(AssertionError):
        dmx.BlockTopK(K=5, block_size=4)
    with pytest.raises(ValueError):
        dmx.Sparseness.from_shorthand("NM{2:4}")


def test_quant_seam_signatures(dmx):
    q = dmx.quant.quant_hip
    for fam in ("block_quantize", "float_quantize", "fixed_point_quantize"):
        for r in ("nearest", "stochastic", "down", "up"):
            assert callable(getattr(q, f"{fam}_{r}"))
    with pytest.raises(AssertionError):
        dmx.quant.float_quantize(torch.zeros(2), 5, 10, rounding="up")   # quant_function.py:138-140
    with pytest.raises(AssertionError):
        dmx.quant.block_quantize(torch.zeros(2), 8, rounding="bogus")
    with pytest.raises(dmx.DmxqError):
        dmx.quant.block_quantize(torch.zeros(2, 16), 8, 0, True, "nearest")


def test_castto_configuration_surface(dmx):
    c = dmx.CastTo()
    assert repr(c.format) == "SAME" and c.block_dim == -1 and c.group_size is None
    c.set_format("BFP[8|8]{64}(SN)")
    assert isinstance(c.format, dmx.BlockFloatingPoint) and c.get_precision() == 8.125
    c.set_pre_transform({"format": "FP[1|5|10,15](FN)", "shaping": [("view", (-1, 4))]})
    assert isinstance(c.pre_transform["format"], dmx.FloatingPoint)
    x = torch.arange(24.0).reshape(2, 3, 4)
    y, inv = dmx.CastTo.apply_shaping_seq(x, [("permute", (2, 0, 1)), ("flatten", (0, 1)), ("view", (4, 6))])
    z, _ = dmx.CastTo.apply_shaping_seq(y, inv)
    assert torch.equal(z, x)
    with pytest.raises(AssertionError):
        dmx.CastTo(format="XP[8,0](CSN)", group_size=4, qscheme=torch.per_channel_symmetric)
    d = dmx.CastToDict({"input_cast": dmx.CastTo(), "other_cast": dmx.CastTo()})
    d.set_format(["BFP[8|8]{16}(SN)", None])
    assert repr(d["input_cast"].format) == "BFP[8|8]{16}(SN)" and repr(d["other_cast"].format) == "SAME"
    with pytest.raises(RuntimeError):
        d.set_format({"nope": "SAME", "x": "SAME"})
    # SAME format needs no GPU: a clone by default, like the reference's Same.cast (numerical/format.py:89-90); the
    # zero-copy form is what DmxModule sets on the casts it owns
    t = torch.randn(3)
    c = dmx.CastTo()
    out = c(t)
    assert torch.equal(out, t) and out is not t and out.data_ptr() != t.data_ptr()
    c.copy_on_same = False
    assert c(t) is t
    # the two switches are mirrored on the host (forward never reads a device buffer) and survive a state_dict round trip
    c.enable_calibration(True, dmx.DummyObserver)
    assert c._flag("fake_quant_enabled") == 0 and c._flag("observer_enabled") == 1 and int(c.observer_enabled[0]) == 1
    c2 = dmx.CastTo()
    c2.load_state_dict(c.state_dict())
    assert c2._flag("fake_quant_enabled") == 0 and c2._flag("observer_enabled") == 1
    c.enable_calibration(False)
    assert c._flag("fake_quant_enabled") == 1 and c._flag("observer_enabled") == 0


LEGACY_YAML = """
conv1:
  accum_format: SAME
  approximation_function: NONE
  bias_format: SAME
  input_format: BFP[8|8]{64,1}(SN)
  instance: Conv2d
  output_format: FP[1|5|10,15](FN)
  weight_format: BFP[8|8]{64,1}(SN)
  weight_sparseness: DENSE
fc1:
  bias_format: SAME
  input_format: BFP[8|8]{64,-1}(SN)
  instance: Linear
  output_format: FP[1|5|10,15](FN)
  weight_format: BFP[8|8]{64,-1}(SN)
mp1:
  input_format: SAME
  instance: MaxPool2d
  output_format: FP[1|5|10,15](FN)
"""


def _tiny_net(dmx):
    nn = dmx.nn
    return torch.nn.Sequential(__import__("collections").OrderedDict(
        conv1=nn.Conv2d(1, 6, 5), relu=nn.ReLU(), mp1=nn.MaxPool2d(2), flat=torch.nn.Flatten(1), fc1=nn.Linear(6 * 14 * 14, 10)))


def test_config1_baseline_rules_on_cpu_equal_the_raw_model(dmx):
    """BASELINE.json config 1 (LeNet-5 + legacy yaml + BASELINE rules, CPU torch): under the reference's current code
    the yaml is a silent no-op and BASELINE = all-SAME casts, so the output is the raw torch model's, bit for bit.
    SAME casts are clones and need no GPU."""
    torch.manual_seed(0)
    net = _tiny_net(dmx)
    dmx.configure_model(net, *dmx.config_rules.BASELINE)
    for m in net.modules():
        if isinstance(m, dmx.DmxModule):
            assert all(repr(f) == "SAME" for f in m.input_formats + m.output_formats)
    raw = torch.nn.Sequential(torch.nn.Conv2d(1, 6, 5), torch.nn.ReLU(), torch.nn.MaxPool2d(2), torch.nn.Flatten(1), torch.nn.Linear(6 * 14 * 14, 10))
    raw[0].load_state_dict({k: v for k, v in net.conv1.state_dict().items() if k in ("weight", "bias")})
    raw[4].load_state_dict({k: v for k, v in net.fc1.state_dict().items() if k in ("weight", "bias")})
    x = torch.randn(3, 1, 32, 32)
    assert torch.equal(net(x), raw(x))
    net.fc1.fold_weight_and_bias()
    assert torch.equal(net(x), raw(x))


def test_legacy_yaml_intent_loader(dmx):
    cfg = dmx.load_legacy_config(LEGACY_YAML)
    assert set(cfg) == {"conv1", "fc1", "mp1"}
    assert repr(cfg["conv1"]["config"]["input_formats"][0]) == "BFP[8|8]{64}(SN)" and cfg["conv1"]["block_dims"] == {"input_cast": 1, "weight_cast": 1}
    assert cfg["fc1"]["block_dims"] == {"input_cast": -1, "weight_cast": -1}
    net = _tiny_net(dmx)
    assert dmx.apply_legacy_config(net, LEGACY_YAML) == 3
    assert repr(net.conv1.weight_format) == "BFP[8|8]{64}(SN)" and net.conv1.weight_cast.block_dim == 1
    assert repr(net.conv1.output_formats[0]) == "FP[1|5|10,15](FN)" and repr(net.conv1.bias_format) == "SAME"
    assert repr(net.mp1.output_formats[0]) == "FP[1|5|10,15](FN)" and repr(net.mp1.input_formats[0]) == "SAME"
    with pytest.raises(TypeError):
        dmx.apply_legacy_config(net, LEGACY_YAML.replace("instance: Linear", "instance: Conv2d"))


def test_rule_sets_and_module_surface(dmx):
    nn, rules = dmx.nn, dmx.config_rules
    lin, conv, mm, add = nn.Linear(16, 8), nn.Conv2d(3, 4, 3), nn.ActActMatMul(), nn.ResAdd()
    net = torch.nn.ModuleDict(dict(a=lin, b=conv, c=mm, d=add, e=nn.Softmax(), f=nn.GELU(), g=nn.LayerNorm(8)))
    dmx.configure_model(net, *rules.BASIC)
    assert [repr(f) for f in lin.input_formats] == ["BFP[8|8]{64}(SN)"] and repr(lin.weight_format) == "BFP[8|8]{64}(SN)"
    assert repr(lin.bias_format) == "BFP[24|8]{1}(SN)" and repr(lin.output_formats[0]) == "FP[1|5|10,15](FN)"
    assert conv.input_casts.input_cast.block_dim == 1 and conv.weight_cast.block_dim == 1 and lin.weight_cast.block_dim == -1
    assert [repr(f) for f in mm.input_formats] == ["BFP[8|8]{64}(SN)"] * 2 and mm.input_casts.multiplier_cast.block_dim == -2
    assert [repr(f) for f in add.input_formats] == ["FP[1|5|10,15](FN)"] * 2
    assert repr(net["e"].approximator.function) == "NONE" and repr(net["f"].input_formats[0]) == "FP[1|5|10,15](FN)"
    dmx.configure_model(net, *rules.FP8)
    assert repr(lin.weight_format) == "FP[1|4|3,7](_N)" and repr(lin.bias_format) == "FP[1|8|23,127](_N)"
    dmx.configure_model(net, *rules.SBFP_WEIGHT_STORAGE)
    assert repr(lin.weight_storage_cast.format) == "SBFP<XP[4,0](CSN)><FP[0|4|4,7](FN)>{16}"
    r = dmx.DmxConfigRule(module_types=(nn.Linear,), name_re="a", module_config=dict(weight_sparseness="BTOPK{2:4,-1}(U)"))
    assert r.names_in(net) == ["a"]
    r.apply_to(net)
    assert repr(lin.weight_sparseness) == "BTOPK{2:4,-1}(U)" and lin.ch_axis == -1 and lin.wout_ch_axis == 0 and conv.win_ch_axis == 1
    assert lin.smoothquant is not None and mm.smoothquant is None and mm.weight_cast is None
    f = dmx.ApproximationFunction.from_shorthand("SOFTMAX[dmxq]{input_clamp=-100}(max_adjust=0.1141)")
    assert f.wrapper_params == {"input_clamp": -100} and f.extra_params == {"max_adjust": 0.1141} and repr(f).startswith("SOFTMAX[dmxq]")
    with pytest.raises(ValueError):
        dmx.ApproximationFunction.from_shorthand("FOO[x]{}()")


def test_module_results_never_alias_inputs_or_parameters(dmx):
    """Reference semantics (every SAME cast is `x.clone()`, numerical/format.py:89-90): whatever a DmxModule returns may
    be mutated in place by the caller without touching the module's input or parameters.  The casts a DmxModule owns
    do not copy; the guarantee is re-established at the module boundary.  All-SAME (BASELINE) modules run on the CPU."""
    nn = dmx.nn
    lin = nn.Linear(8, 4)
    assert all(not c.copy_on_same for c in lin.modules() if isinstance(c, dmx.CastTo))
    x = torch.randn(3, 8)
    x0 = x.clone()
    y = lin(x)
    y.add_(100.0)
    assert torch.equal(x, x0)
    # pass-through shapes: the result of the internal chain IS the input -> one clone at the boundary
    class Identity(nn.DmxModule):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self._dmx_init()

        def _forward(self, _input):
            return _input

    ident = Identity()
    out = ident(x)
    assert out is not x and out.untyped_storage().data_ptr() != x.untyped_storage().data_ptr() and torch.equal(out, x)
    out.zero_()
    assert torch.equal(x, x0)
    sliced = Identity()
    sliced._forward = lambda _input: _input[1:]        # a view of the input must not escape either
    out = sliced(x)
    out.zero_()
    assert torch.equal(x, x0)
    # the weight / bias views handed to a caller are copies; the read-only internal ones may be the Parameters
    w0 = lin.weight.detach().clone()
    assert lin._weight_ro.data_ptr() == lin.weight.data_ptr()
    lin._weight.data.zero_()
    lin._bias.data.zero_()
    assert torch.equal(lin.weight, w0)
    lin.fold_weight_and_bias()
    lin._weight.data.zero_()
    assert torch.equal(lin.weight, w0)
    assert all(not c.copy_on_same for c in lin.modules() if isinstance(c, dmx.CastTo))


# ------------------------------------------------------------------------------------------------ ADVICE r3
def test_fx_links_skip_producers_called_at_several_sites_and_compound_consumers(dmx):
    """link_consumers_from_fx: the link is stored on the MODULE, so a producer instance called twice has no single consumer set and
    must stay unlinked; a consumer that does not run DmxModule.forward (the compound SDPA never applies its own input casts)
    must not absorb a producer's cast."""
    import torch.fx as fx
    nn = dmx.nn

    class Shared(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc0, self.fc1, self.fc2 = nn.Linear(64, 64), nn.Linear(64, 64), nn.Linear(64, 64)
            self.act = nn.ReLU()
            self.norm = nn.LayerNorm(64)

        def forward(self, x):
            a = self.act(self.fc0(x))            # first call site: read by torch.sin AND fc1
            b = self.act(self.fc1(a) + torch.sin(a))   # second call site: read by fc2 only
            return self.fc2(self.norm(b))

    m = Shared()
    for mod in m.modules():
        if isinstance(mod, nn.DmxModule):
            mod.configure({"input_formats": ["BFP[8|8]{64}(SN)"]} if isinstance(mod, nn.Linear) else {})
    gm = fx.GraphModule(m, nn.DmxTracer().trace(m))
    nn.link_consumers_from_fx(gm)
    assert "_next_consumers" not in m.act.__dict__, "a module called at two sites must not be linked by its last call"
    assert m.norm.__dict__.get("_next_consumers") == ((m.fc2, 0),)      # single call site, single DmxModule reader: linked

    class IntoCompound(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q = nn.Linear(64, 64)
            self.sdpa = nn.ScaledDotProductAttention()

        def forward(self, x):
            q = self.q(x)
            return self.sdpa(q, x, x)

    m2 = IntoCompound()
    gm2 = fx.GraphModule(m2, nn.DmxTracer().trace(m2))
    nn.link_consumers_from_fx(gm2)
    assert "_next_consumers" not in m2.q.__dict__


def test_dropout_and_quick_gelu_survive_deepcopy_and_pickle(dmx):
    """functional_forward must not be a closure over the instance: a deepcopy would keep reading the ORIGINAL's p / training, and a
    local lambda cannot be pickled (the reference's modules have neither problem)"""
    import copy
    import io
    import pickle

    nn = dmx.nn
    d = nn.Dropout(p=0.5)
    d.train()
    c = copy.deepcopy(d)
    c.eval()
    x = torch.ones(1000)
    assert torch.equal(c(x), x), "a copied Dropout in eval() must be the identity"
    d.eval()
    c.train()
    assert int((c(x) == 0).sum()) > 300 and torch.equal(d(x), x)
    for mod in (nn.Dropout(p=0.25), nn.QuickGELU()):
        back = pickle.loads(pickle.dumps(mod))
        assert type(back) is type(mod)
        buf = io.BytesIO()
        torch.save(mod, buf)
    q = nn.QuickGELU()
    xs = torch.linspace(-3, 3, 50)
    assert torch.equal(copy.deepcopy(q)(xs), xs * torch.sigmoid(1.702 * xs))


def test_host_flag_mirrors_fall_back_to_the_buffers(dmx):
    """_flags.HostFlags: a module whose __dict__ lacks the mirrors (pickled by an earlier version) reads its buffers once; the
    scalar mirror equals the BUFFER's value (0.3 -> float32(0.3)) whichever path wrote it"""
    sq = dmx.smoothquant.ActivationWeightSmoothQuant(ch_axis=-1, win_ch_axis=-1) if hasattr(dmx, "smoothquant") else None
    if sq is None:
        pytest.skip("no smoothquant module")
    sq.set_migration_strength(0.3)
    via_set = sq._scalar("migration_strength")
    for k in [k for k in sq.__dict__ if k.startswith("_h_")]:
        del sq.__dict__[k]
    assert sq._scalar("migration_strength") == via_set == float(torch.tensor(0.3, dtype=torch.float32))
    assert sq._flag("enabled") in (0, 1)
    assert "migration_strength" in sq.extra_repr()


def test_patch_reference_surface_and_cpu_routing(dmx):
    """integration.patch_reference against the RECORDED surface of the real classes (oracle/check_patch_reference.py ran it on the
    reference itself): every attribute the wrappers read is there; on stand-ins, CPU tensors reach the original methods and undo()
    restores them."""
    import json

    from dmx_compressor_amd import integration as I

    surf = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_surface.json")))
    for cls, names in I.SURFACE.items():
        assert set(names) <= set(surf["attributes"][cls]), cls
    assert surf["cpu_results_identical_before_and_after_patching"] >= 60
    fm, sm, qm, StandinCalled = I.standins_from_surface(surf["attributes"])
    before = fm.BlockFloatingPoint.cast
    undo = I.patch_reference(format_module=fm, sparse_module=sm, quant_function_module=qm)
    assert len(undo.patched) == 7 and fm.BlockFloatingPoint.cast is not before
    f = fm.BlockFloatingPoint()
    f.precision, f.block_size, f.symmetric, f.rounding = 8, 16, True, "nearest"
    with pytest.raises(StandinCalled):
        f.cast(torch.zeros(4, 16), -1)
    assert qm.get_module(torch.zeros(2)) == "reference-native-module"
    undo()
    assert fm.BlockFloatingPoint.cast is before


def test_one_front_end_two_bindings_same_raw_schema(dmx):
    """`_front.py` is written once over a raw namespace; both bindings must offer every raw op it calls, under the SAME name and
    parameter list as the dispatcher schema of csrc/torch_binding.cpp (no second front end to keep in step)."""
    import inspect

    from dmx_compressor_amd import _backend_ctypes as B

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    front = open(os.path.join(root, "dmx-compressor_amd", "_front.py")).read()
    used = set(re.findall(r"_ops\.([a-z_0-9]+)", front))
    cpp = open(os.path.join(root, "dmx-compressor_amd", "csrc", "torch_binding.cpp")).read()
    schemas = {n: re.sub(r"\(\w!\)", "", ps) for n, ps in re.findall(r'm\.def\("([a-z_0-9]+)\((.*?)\) ->', cpp)}   # (drop alias annotations)
    assert used <= set(schemas), used - set(schemas)
    for name, params in schemas.items():
        assert hasattr(B, name), f"the ctypes binding lacks the raw op {name}"
        want = [p.strip().split("=")[0].split()[-1] for p in params.split(",") if p.strip()]
        got = list(inspect.signature(getattr(B, name)).parameters)
        assert got == want, (name, got, want)
    assert not os.path.exists(os.path.join(root, "dmx-compressor_amd", "_ops_ctypes.py"))


def test_rows_plan_size_classes_through_describe():
    """csrc/common.hpp rows_plan, without a GPU (dmxq_bfp_qdq_describe launches nothing): from 20 to 40 MiB the symmetric 16-bit ->
    same-16-bit build runs ONE round of <= 256 workgroups whose depth is ceil(n_vec / 2^17) (round 4); the float32 build up to depth
    16; asymmetric / widening builds keep 512 x 16 and the multi-round 512 x 2."""
    import ctypes
    import re
    from dmx_compressor_amd import _lib
    L = _lib.lib()
    buf = ctypes.create_string_buffer(256)

    def plan(dt_in, dt_out, rows, cols, sym=1):
        assert L.dmxq_bfp_qdq_describe(dt_in, dt_out, rows, cols, 1, 16, 8, _lib.ROUND_NEAREST, sym, 1, buf, 256) == _lib.OK
        m = re.search(r"tile (\d+)x(\d+) vectors, grid (\d+)", buf.value.decode())
        return int(m.group(1)), int(m.group(2)), int(m.group(3))

    for rows in range(2561, 5121, 37):
        t, u, grid = plan(_lib.BF16, _lib.BF16, rows, 4096)
        assert (t, u) == (512, -(-rows // 256)) and grid <= 256 and grid * 512 * u >= rows * 512, (rows, t, u, grid)
    assert plan(_lib.BF16, _lib.BF16, 2560, 4096)[:2] == (128, 8)
    assert plan(_lib.BF16, _lib.BF16, 5121, 4096)[:2] == (512, 2)
    assert plan(_lib.F16, _lib.F16, 4300, 4096)[:2] == (512, 17)
    assert plan(_lib.F32, _lib.F32, 3072, 2048)[:2] == (512, 12) and plan(_lib.F32, _lib.F32, 4200, 2048)[:2] == (512, 2)
    assert plan(_lib.BF16, _lib.BF16, 3072, 4096, sym=0)[:2] == (512, 16) and plan(_lib.BF16, _lib.BF16, 4300, 4096, sym=0)[:2] == (512, 2)
    assert plan(_lib.BF16, _lib.F32, 1400, 4096)[:2] == (512, 16)   # widening: lane-vectors of 4 elements, no exact-depth build
