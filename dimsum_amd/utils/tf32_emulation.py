"""Measurement infrastructure (tests/, bench.py's deviation columns): the reference's matmul arithmetic restated for comparisons -- no
product path imports this.

The reference runs its Linears under torch.backends.cuda.matmul.allow_tf32 = True (dimsum/train.py:20-21, sample_ddp.py:56): on its
hardware every matmul operand is rounded to TF32 (8-bit exponent, 10-bit mantissa), products are accumulated in fp32. `emulated_tf32()`
reproduces that on any device: inside the context every torch matmul-class call (linear, matmul, mm, addmm, bmm, baddbmm) gets its
floating-point operands rounded to 10 mantissa bits (round to nearest even -- the kinder reading of the hardware, which may truncate)
and is then evaluated exactly in fp32 (allow_tf32 off). Kernels of libdimsum_hip.so are not torch ops and stay as they are -- except the
attention core: the reference evaluates it with matmuls under the same flag (dimsum/attention_fusion.py:44-57), so inside the context
`attention_fusion._XattnCoreFn` is replaced by the same math written out in torch with rounded matmul operands, forward and backward."""
import contextlib

import torch
from torch.overrides import TorchFunctionMode


def round_tf32(x):
    if not (torch.is_tensor(x) and x.dtype == torch.float32):
        return x
    b = x.contiguous().view(torch.int32)
    b = (b + 0xFFF + ((b >> 13) & 1)) & ~0x1FFF
    return b.view(torch.float32).view(x.shape)


_MATMULS = {torch.nn.functional.linear, torch.matmul, torch.mm, torch.bmm, torch.Tensor.matmul, torch.Tensor.mm, torch.Tensor.bmm,
            torch.Tensor.__matmul__, torch.Tensor.__rmatmul__}
_ADDMMS = {torch.addmm, torch.baddbmm, torch.Tensor.addmm, torch.Tensor.baddbmm}


class _EmuLinearFn(torch.autograd.Function):
    """F.linear under emulated TF32 in BOTH directions: torch's own autograd of F.linear runs its two backward matmuls inside the C++ engine,
    where a TorchFunctionMode does not see them -- the reference's backward GEMMs run under allow_tf32 too (dimsum/train.py:20-21)"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        y = torch.mm(round_tf32(x.reshape(-1, x.shape[-1])), round_tf32(weight).t()).view(*x.shape[:-1], weight.shape[0])
        return y if bias is None else y + bias

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy2 = round_tf32(dy.reshape(-1, dy.shape[-1]).contiguous())
        dx = torch.mm(dy2, round_tf32(weight)).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = torch.mm(dy2.t(), round_tf32(x.reshape(-1, x.shape[-1]))) if ctx.needs_input_grad[1] else None
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


class _EmuXattnFn(torch.autograd.Function):
    """the fusion core (attention_fusion.py:44-79: q * scale, q k^T, softmax, attn v; both directions, or self-attention with qkv2 = None) with
    every matmul operand rounded to TF32, and its adjoint written out the same way -- same signature as attention_fusion._XattnCoreFn"""

    @staticmethod
    def _split(qkv, bias, heads):
        B, L, W = qkv.shape
        t = qkv if bias is None else qkv + bias
        return t.reshape(B, L, 3, heads, W // (3 * heads)).permute(2, 0, 3, 1, 4).unbind(0)

    @staticmethod
    def _dirs(qkv1, qkv2, bias1, bias2, heads):
        qa, ka, va = _EmuXattnFn._split(qkv1, bias1, heads)
        if qkv2 is None:
            return [(qa, ka, va)]
        qb, kb, vb = _EmuXattnFn._split(qkv2, bias2, heads)
        return [(qa, kb, vb), (qb, ka, va)]

    @staticmethod
    def forward(ctx, qkv1, qkv2, bias1, bias2, heads):
        B, L, W = qkv1.shape
        hd = W // (3 * heads)
        outs, probs = [], []
        for q, k, v in _EmuXattnFn._dirs(qkv1, qkv2, bias1, bias2, heads):
            p = torch.softmax(torch.matmul(round_tf32(q * hd ** -0.5), round_tf32(k).transpose(-1, -2)), dim=-1)
            outs.append(torch.matmul(round_tf32(p), round_tf32(v)))
            probs.append(p)
        ctx.heads = heads
        ctx.save_for_backward(qkv1, qkv2, bias1, bias2, *probs, *outs)
        return torch.cat([o.transpose(1, 2).reshape(B, L, heads * hd) for o in outs], dim=-1)

    @staticmethod
    def backward(ctx, dout):
        qkv1, qkv2, bias1, bias2, *rest = ctx.saved_tensors
        heads = ctx.heads
        B, L, W = qkv1.shape
        hd, C = W // (3 * heads), W // 3
        dirs = _EmuXattnFn._dirs(qkv1, qkv2, bias1, bias2, heads)
        probs, outs = rest[:len(dirs)], rest[len(dirs):]
        rows = lambda t: t.permute(0, 2, 1, 3).reshape(B, L, C)
        grads = []
        for d, ((q, k, v), p, o) in enumerate(zip(dirs, probs, outs)):
            do = dout[..., d * C:(d + 1) * C].reshape(B, L, heads, hd).permute(0, 2, 1, 3)
            dp = torch.matmul(round_tf32(do), round_tf32(v).transpose(-1, -2))
            ds = round_tf32(p * (dp - (do * o).sum(-1, keepdim=True)))
            grads.append((torch.matmul(ds, round_tf32(k)) * hd ** -0.5, torch.matmul(ds.transpose(-1, -2), round_tf32(q * hd ** -0.5)),
                          torch.matmul(round_tf32(p).transpose(-1, -2), round_tf32(do))))
        if qkv2 is None:
            d1, d2 = torch.cat([rows(g) for g in grads[0]], dim=-1), None
        else:
            (dqa, dkb, dvb), (dqb, dka, dva) = grads
            d1, d2 = torch.cat((rows(dqa), rows(dka), rows(dva)), dim=-1), torch.cat((rows(dqb), rows(dkb), rows(dvb)), dim=-1)
        db1 = d1.reshape(-1, W).sum(0) if bias1 is not None else None
        db2 = d2.reshape(-1, W).sum(0) if (bias2 is not None and d2 is not None) else None
        return d1, d2, db1, db2, None


class _Tf32Mode(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.nn.functional.linear and torch.is_grad_enabled():
            x, w = args[0], (args[1] if len(args) > 1 else kwargs["weight"])
            b = args[2] if len(args) > 2 else kwargs.get("bias")
            if x.dtype == torch.float32 and w.dtype == torch.float32 and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
                return _EmuLinearFn.apply(x, w, b)
        if func in _MATMULS:
            args = tuple(round_tf32(a) for a in args[:2]) + tuple(args[2:])
            if func is torch.nn.functional.linear and "weight" in kwargs:
                kwargs = dict(kwargs, weight=round_tf32(kwargs["weight"]))
        elif func in _ADDMMS:
            args = (args[0],) + tuple(round_tf32(a) for a in args[1:3]) + tuple(args[3:])
        return func(*args, **kwargs)


@contextlib.contextmanager
def emulated_tf32():
    from .. import attention_fusion
    old, core = torch.backends.cuda.matmul.allow_tf32, attention_fusion._XattnCoreFn
    torch.backends.cuda.matmul.allow_tf32 = False
    attention_fusion._XattnCoreFn = _EmuXattnFn
    try:
        with _Tf32Mode():
            yield
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old
        attention_fusion._XattnCoreFn = core
