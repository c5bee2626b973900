"""Weight sparsity — mirror of the reference's `sparse.py` (Sparseness shorthands, Sparsify module).

Hot path: `BlockTopK` (N:M structured sparsity, sparse.py:140-198) and `Sparsify.forward` (sparse.py:287-301).
The reference builds the mask with argsort (int64 indices, 8 B/elem) + ones + scatter and then multiplies;
here mask and `x * mask` come out of ONE kernel in which each lane ranks its M-group in registers.
`TopK` (global unstructured: a radix select + one masking pass instead of a whole-tensor argsort, csrc/topk.hip) and
`Bernoulli` (counter-based draws) are device launches too.
"""
import re
from typing import Optional

import torch
from torch.autograd import Function

from . import ops

__all__ = ["Sparseness", "Dense", "TopK", "BlockTopK", "Bernoulli", "Sparsify"]


class Sparseness:
    blocked: bool = False
    density: Optional[float] = None

    def __init__(self, mask_gradient=False):
        self.mask_gradient = bool(mask_gradient)

    def get_mask(self, score):
        raise NotImplementedError

    @classmethod
    def from_shorthand(cls, sh: str):
        if isinstance(sh, Sparseness):
            return sh
        for prefix, klass in (("DENSE", Dense), ("TOPK", TopK), ("BTOPK", BlockTopK), ("BERN", Bernoulli)):
            if sh.startswith(prefix):
                return klass.from_shorthand(sh)
        raise ValueError(f"unrecognized sparseness shorthand: {sh}")


class Dense(Sparseness):
    def __init__(self, mask_gradient=False):
        super().__init__(mask_gradient)
        self.density = 1.0

    def get_mask(self, score):
        return None

    @classmethod
    def from_shorthand(cls, sh: str):
        return cls()

    def __str__(self):
        return "Dummy sparseness: no pruning"

    def __repr__(self):
        return "DENSE"


class _TopKMask(Function):
    """mask = TopK(score); identity gradient to the score (sparse.py:125-127)."""

    @staticmethod
    def forward(ctx, score, density):
        return ops.topk_mask(score, density)

    @staticmethod
    def backward(ctx, g):
        return g, None


class TopK(Sparseness):
    """Global top-K (sparse.py:95-137): the int(n * (1 - density)) lowest scores of the whole tensor are zeroed."""

    def __init__(self, density=0.5, mask_gradient=False):
        super().__init__(mask_gradient)
        assert 0 <= density <= 1.0, "density has to be between 0 and 1"
        self.density = density

    def get_mask(self, score):
        return _TopKMask.apply(score, self.density)

    @classmethod
    def from_shorthand(cls, sh: str):
        m = re.fullmatch(r"TOPK\{([-+]?\d*\.?\d+(?:[eE][-+]?\d+)?)\}\(([MU])\)", sh)
        if m is None:
            raise ValueError(f"unrecognized sparseness shorthand: {sh}")
        return cls(density=float(m[1]), mask_gradient=m[2] == "M")

    def __str__(self):
        return f"Global TopK sparseness: density = {self.density}"

    def __repr__(self):
        return f"TOPK{{{self.density}}}({'M' if self.mask_gradient else 'U'})"


class _NMMask(Function):
    """mask = N:M(score); identity gradient to the score (sparse.py:182-184)."""

    @staticmethod
    def forward(ctx, score, K, M, block_dim):
        return ops.nm_mask(score, K, M, block_dim)

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None


class _NMSparsify(Function):
    """y = x * N:M(score), fused.  Backward of the product: dx = g * mask (STE passes weight gradients through the
    multiply exactly as autograd would for `x * mask` in the reference), dscore = g * x when requested."""

    @staticmethod
    def forward(ctx, x, score, K, M, block_dim, need_mask_grad):
        y, mask = ops.nm_sparsify(x, score, K, M, block_dim, return_mask=True)
        ctx.save_for_backward(mask, x if need_mask_grad else None)
        ctx.x_dtype, ctx.s_dtype = x.dtype, score.dtype
        return y, mask

    @staticmethod
    def backward(ctx, gy, gmask):
        mask, x = ctx.saved_tensors
        gx = (gy * mask).to(ctx.x_dtype) if gy is not None and ctx.needs_input_grad[0] else None
        gs = None
        if ctx.needs_input_grad[1]:
            gs = (gy * x).to(ctx.s_dtype) if (gy is not None and x is not None) else None
            if gmask is not None:
                gs = gmask if gs is None else gs + gmask
        return gx, gs, None, None, None, None


class BlockTopK(Sparseness):
    """K non-zeros out of every `block_size` consecutive elements along `block_dim` (sparse.py:140-198)."""

    blocked = True

    def __init__(self, K=4, block_size=8, block_dim=-1, mask_gradient=False):
        super().__init__(mask_gradient)
        assert 0 < K <= block_size, "N and M must be positive and N no greater than M"
        self.K, self.block_size, self.block_dim = K, block_size, block_dim
        self.density = K / block_size

    def get_mask(self, score):
        return _NMMask.apply(score, self.K, self.block_size, self.block_dim)

    @classmethod
    def from_shorthand(cls, sh: str):
        m = re.fullmatch(r"BTOPK\{(\d+):(\d+),(-?\d+)\}\(([MU])\)", sh)
        if m is None:
            raise ValueError(f"unrecognized sparseness shorthand: {sh}")
        return cls(K=int(m[1]), block_size=int(m[2]), block_dim=int(m[3]), mask_gradient=m[4] == "M")

    def __str__(self):
        return f"Block TopK sparseness: pattern = {self.K}:{self.block_size}, block dimension = {self.block_dim}"

    def __repr__(self):
        return f"BTOPK{{{self.K}:{self.block_size},{self.block_dim}}}({'M' if self.mask_gradient else 'U'})"


class _BernoulliMask(Function):
    @staticmethod
    def forward(ctx, score):
        return ops.bernoulli_mask(score)

    @staticmethod
    def backward(ctx, g):
        return g


class Bernoulli(Sparseness):
    """Bernoulli supermask sampler (sparse.py:201-242).  Draws come from a counter-based stream, not torch's global
    generator: statistical parity with the reference."""

    def get_mask(self, score):
        # sparse.py:211-213: the scores need to be within [0, 1] (a device->host read, as in the reference)
        mn, mx = ops.group_minmax(score.detach().reshape(1, -1), 0, 1)
        assert float(mx) <= 1 and float(mn) >= 0
        return _BernoulliMask.apply(score)

    @classmethod
    def from_shorthand(cls, sh: str):
        return cls()

    def __str__(self):
        return "Bernoulli sparseness"

    def __repr__(self):
        return "BERN"


class Sparsify(torch.nn.Module):
    """Sparsification module (sparse.py:245-320): holds a `score` Parameter (uniform random init), recomputes the
    mask on every forward and multiplies.  Reference quirk kept: a `score_func` result
    is used for exactly one forward and never written back to `self.score`."""

    # which of (x, score) receive a gradient through x * mask, per backward mode (sparse.py:266-275)
    _GRADIENT_ROUTES = {"ste": (True, False), "supermask": (False, True), "joint": (True, True)}

    def __init__(self, tensor_shape, sparseness="DENSE", backward_mode="STE", score_func=None):
        super().__init__()
        self._mask, self._mask_pending, self.plastic = None, True, False
        self.score = torch.nn.Parameter(torch.rand(tensor_shape), requires_grad=True)   # uniform-random until trained / assigned
        self._set_sparseness(sparseness)
        self._set_backward_mode(backward_mode)
        if score_func is not None:
            self.score_func = score_func   # (the constructor does NOT arm the one-shot rewiring: only configure() does)

    # configure(): every argument optional, each with a setter of its own
    def _set_sparseness(self, spec):
        new = Sparseness.from_shorthand(spec)
        current = getattr(self, "sparseness", None)
        if current is None or repr(current) != repr(new):   # the shorthand is the identity of a sparseness
            self.sparseness = new

    def _set_backward_mode(self, mode):
        self.backward_mode = mode
        self.enable_weight_gradient, self.enable_mask_gradient = self._GRADIENT_ROUTES.get(mode.lower(), (False, False))

    def configure(self, sparseness=None, backward_mode=None, score_func=None):
        if sparseness is not None:
            self._set_sparseness(sparseness)
        if backward_mode is not None:
            self._set_backward_mode(backward_mode)
        if score_func is not None:
            self.score_func, self.plastic = score_func, True   # the next forward scores through it, once (quirk above)

    def update_mask(self, score):
        self.mask = self.sparseness.get_mask(score)

    @property
    def mask(self):
        """The reference's constructor ends with `update_mask(self.score)` (sparse.py:260-262): a freshly built Sparsify HAS the mask of
        its initial score before any forward.  Here that first mask is computed when it is first READ, from the score as it is then: the
        mask kernels run on the GPU only and the constructor's score lives on the host (a constructor-time launch would copy every
        score of a model to the device and back for a mask most callers never look at; Llama-3-8B: 7 G elements).  A score that sits on
        the host is masked on the device and the mask returned to the score's device."""
        if self._mask is None and self._mask_pending:
            sc = self.score.detach()
            on = sc if sc.is_cuda else sc.to("cuda")
            m = self.sparseness.get_mask(on)
            self._mask, self._mask_pending = (None if m is None else m.to(sc.device)), False
        return self._mask

    @mask.setter
    def mask(self, m):
        self._mask, self._mask_pending = m, False

    @property
    def density(self):
        """the sparseness' nominal density, or -- where it has none -- the measured one of the mask the stored score gives
        (which also becomes `self.mask`, as in the reference: sparse.py:303-308)"""
        nominal = self.sparseness.density
        if nominal is not None:
            return nominal
        self.update_mask(self.score)
        kept = self.mask.data
        return kept.sum() / kept.numel()

    def extra_repr(self):
        return f"sparseness = {self.sparseness!r}, backward_mode = {self.backward_mode}"

    def forward(self, x):
        if isinstance(self.sparseness, Dense):
            return x
        if self.plastic:
            score = self.score_func(self.score, x)
            self.plastic = False
        else:
            score = self.score
        if not isinstance(self.sparseness, BlockTopK):
            if isinstance(self.sparseness, TopK) and not (torch.is_grad_enabled() and (x.requires_grad or score.requires_grad)):
                y, self.mask = ops.topk_sparsify(x, score.to(x.device), self.sparseness.density, return_mask=True)
                return y    # mask and x * mask out of the same final pass
            self.update_mask(score)
            if self.training:  # sparse.py:296-300
                x = x if self.enable_weight_gradient else x.detach()
                self.mask = self.mask if self.enable_mask_gradient else self.mask.detach()
            return x * self.mask
        sp = self.sparseness
        if score.device != x.device:
            score = score.to(x.device)
        if self.training:
            xin = x if self.enable_weight_gradient else x.detach()
            sin = score if self.enable_mask_gradient else score.detach()
        else:
            xin, sin = x, score
        y, mask = _NMSparsify.apply(xin, sin, sp.K, sp.block_size, sp.block_dim,
                                    bool(self.training and self.enable_mask_gradient))
        self.mask = mask
        return y

