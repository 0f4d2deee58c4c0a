"""Measurement infrastructure (tests/, bench.py's deviation columns): the reference's matmul arithmetic restated for comparisons -- no
product path imports this.

The reference runs its Linears under torch.backends.cuda.matmul.allow_tf32 = True (dimsum/train.py:20-21, sample_ddp.py:56): on its
hardware every matmul operand is rounded to TF32 (8-bit exponent, 10-bit mantissa), products are accumulated in fp32. `emulated_tf32()`
reproduces that on any device: inside the context every torch matmul-class call (linear, matmul, mm, addmm, bmm, baddbmm) gets its
floating-point operands rounded to 10 mantissa bits (round to nearest even -- the kinder reading of the hardware, which may truncate)
and is then evaluated exactly in fp32 (allow_tf32 off). Kernels of libdimsum_hip.so are not torch ops and stay as they are."""
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
    old = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        with _Tf32Mode():
            yield
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old
