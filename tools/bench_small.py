#!/usr/bin/env python3
"""tools/bench_small.py — GPU time per launch of this library's ops at the tensor sizes of ONE configured layer (BASELINE.json configs
2 / 3 / 4: opt-125m 2 x 128 tokens fp32, Llama-3-8B 1 x 128 tokens bf16, Whisper-small 1 x 1500 positions fp32), where a launch is
0.5-5 MB and the time is launch ramp + one memory round trip, not bandwidth.  Each op is captured 50 times in a hipGraph over 8 rotating
inputs and replayed (no host cost in the number); the yardstick is an empty-ish launch: a 16-byte-per-lane copy of the same tensor by
torch (`x.clone()`).  Output: profiles/r03_small_tensor_ops.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dmx_compressor_amd as dmx  # noqa: E402

ops = dmx.ops
dev = torch.device("cuda:0")
F16 = dmx.Format.from_shorthand("FP[1|5|10,15](FN)")
BF16, F32 = torch.bfloat16, torch.float32
N_IN, N_CALL = 8, 50


def graph_us(fn, xs):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3):
            fn(xs[i % N_IN])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(N_CALL):
                y = fn(xs[i % N_IN])
                assert y is not None
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (10 * N_CALL))
    return best


def row(name, shape, dtype, fn, make=None):
    xs = [(make() if make else torch.randn(shape, device=dev).to(dtype)) for _ in range(N_IN)]
    us = graph_us(fn, xs)
    base = graph_us(lambda t: t.clone(), xs)
    mb = xs[0].numel() * xs[0].element_size() / 1e6
    print(f"{name:86s} {mb:6.2f} MB  {us:6.2f} us   (torch clone of the input: {base:5.2f} us)", flush=True)


def main():
    # ---- opt-125m decoder layer, fp32, 2 x 128 tokens, hidden 768, 12 heads
    t, h = 256, 768
    w = torch.randn(h, device=dev)
    print("# opt-125m layer (fp32, 256 tokens x 768)")
    row("FLOAT16 cast of an activation [256, 768]", (t, h), F32, lambda x: ops.float_qdq(x, 10, 5, 15, True))
    row("BFP16_64 input cast [256, 768]", (t, h), F32, lambda x: ops.bfp_qdq(x, 8, 64))
    row("BFP16_64 input cast [256, 3072]", (t, 4 * h), F32, lambda x: ops.bfp_qdq(x, 8, 64))
    row("LayerNorm module [256, 768] (FLOAT16 casts)", (t, h), F32, lambda x: ops.layernorm_cast(x, h, w, w, 1e-5, F16, F16))
    row("ReLU module [256, 3072] (FLOAT16 casts)", (t, 4 * h), F32, lambda x: ops.relu_cast(x, F16, F16))
    row("ReLU module + fc2's BFP16_64 input cast [256, 3072], one launch", (t, 4 * h), F32, lambda x: ops.relu_cast(x, F16, F16, then_bfp=(8, 64)))
    row("ResAdd module [256, 768] (FLOAT16 casts)", (t, h), F32, lambda x: ops.binary_cast(x, x, "add", F16, F16, F16))
    row("softmax module [24, 128, 128] (FLOAT16 casts)", (24, 128, 128), F32, lambda x: ops.softmax_cast(x, -1, F16, F16))
    row("BFP16_64 of the V operand [24, 128, 64] along dim -2", (24, 128, 64), F32, lambda x: ops.bfp_qdq(x, 8, 64, block_dim=-2))
    sc = torch.rand(6, device=dev) * 0.01 + 0.001
    zp = torch.zeros(6, dtype=torch.int64, device=dev)
    row("INT8 group-128 weight cast [768, 768]", (h, h), F32, lambda x: ops.fixed_qdq(x, 8, 0, scale=sc, zero_point=zp, ch_axis=0, group_size=128))
    # ---- Llama-3-8B block, bf16, 128 tokens, hidden 4096
    t, h = 128, 4096
    w = torch.randn(h, device=dev).to(BF16)
    print("# Llama-3-8B block (bf16, 128 tokens x 4096)")
    row("FLOAT16 cast of an activation [128, 4096]", (t, h), BF16, lambda x: ops.float_qdq(x, 10, 5, 15, True))
    row("BFP16_64 input cast [128, 4096]", (t, h), BF16, lambda x: ops.bfp_qdq(x, 8, 64))
    row("BFP16_64 input cast [128, 14336]", (t, 14336), BF16, lambda x: ops.bfp_qdq(x, 8, 64))
    row("RMSNorm module [128, 4096] (FLOAT16 casts)", (t, h), BF16, lambda x: ops.rmsnorm_cast(x, h, w, 1e-5, F16, F16))
    row("SiLU module [128, 14336] (FLOAT16 casts)", (t, 14336), BF16, lambda x: ops.unary_cast(x, "silu", F16, F16))
    # the same module as a table lookup (csrc/lut16.hip; the table is built once): where the 128 KiB table copy per workgroup pays
    _tab = ops.unary_cast_table(torch.empty(8, device=dev, dtype=BF16), "silu", F16, F16)
    row("  ... as a table lookup (dmxq_lut16_apply)", (t, 14336), BF16, lambda x: ops.lut16_apply(x, _tab))
    for rows_ in (16, 64, 512, 2048):
        row(f"SiLU module [{rows_}, 4096] bf16: direct kernel", (rows_, 4096), BF16, lambda x: ops.unary_cast(x, "silu", F16, F16))
        row(f"SiLU module [{rows_}, 4096] bf16: table lookup", (rows_, 4096), BF16, lambda x: ops.lut16_apply(x, _tab))
    row("Mul module [128, 14336] (FLOAT16 casts)", (t, 14336), BF16, lambda x: ops.binary_cast(x, x, "mul", F16, F16, F16))
    row("Mul module + down_proj's BFP16_64 input cast [128, 14336], one launch", (t, 14336), BF16, lambda x: ops.binary_cast(x, x, "mul", F16, F16, F16, then_bfp=(8, 64)))
    row("softmax module [32, 128, 128] (FLOAT16 casts)", (32, 128, 128), BF16, lambda x: ops.softmax_cast(x, -1, F16, F16))
    row("BFP16_64 of the V operand [32, 128, 128] along dim -2", (32, 128, 128), BF16, lambda x: ops.bfp_qdq(x, 8, 64, block_dim=-2))
    # ---- Whisper-small encoder layer, fp32, 1500 positions, hidden 768
    t, h = 1500, 768
    w = torch.randn(h, device=dev)
    s = torch.rand(h, device=dev) + 0.5
    print("# Whisper-small encoder layer (fp32, 1500 positions x 768)")
    row("FLOAT16 cast of an activation [1500, 768]", (t, h), F32, lambda x: ops.float_qdq(x, 10, 5, 15, True))
    row("SmoothQuant x / s + BFP16_64 input cast [1500, 768]", (t, h), F32, lambda x: ops.input_hypernet(x, s, 8, 64))
    row("dense SmoothQuant w * s + BFP16_64 weight path [768, 768]", (h, h), F32, lambda x: ops.weight_hypernet(x, 8, 64, True, sq_scale=s))
    row("dense SmoothQuant w * s + BFP16_64 weight path [3072, 768]", (4 * h, h), F32, lambda x: ops.weight_hypernet(x, 8, 64, True, sq_scale=s))
    row("LayerNorm module [1500, 768] (FLOAT16 casts)", (t, h), F32, lambda x: ops.layernorm_cast(x, h, w, w, 1e-5, F16, F16))
    row("GELU module [1500, 3072] (FLOAT16 casts)", (t, 4 * h), F32, lambda x: ops.unary_cast(x, "gelu", F16, F16))
    row("ResAdd module [1500, 768] (FLOAT16 casts)", (t, h), F32, lambda x: ops.binary_cast(x, x, "add", F16, F16, F16))
    row("softmax module [12, 1500, 1500] (FLOAT16 casts)", (12, 1500, 1500), F32, lambda x: ops.softmax_cast(x, -1, F16, F16))
    row("BFP16_64 of the attention probabilities [12, 1500, 1500]", (12, 1500, 1500), F32, lambda x: ops.bfp_qdq(x, 8, 64))
    row("BFP16_64 of the V operand [12, 1500, 64] along dim -2", (12, 1500, 64), F32, lambda x: ops.bfp_qdq(x, 8, 64, block_dim=-2))
    row("channel_maxabs of an activation [1500, 768] (SmoothQuant calibration)", (t, h), F32, lambda x: ops.channel_maxabs(x, 1))


if __name__ == "__main__":
    main()
