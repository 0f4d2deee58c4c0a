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


class _Tf32Mode(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
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
