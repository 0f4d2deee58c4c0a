"""Fused residual-add + LayerNorm / RMSNorm -- same API as mamba/mamba_ssm/ops/triton/layernorm.py:367-486
(LayerNormFn, layer_norm_fn, rms_norm_fn, RMSNorm), running the HIP kernels of csrc/norm.hip instead of Triton."""
import torch
from torch.amp import custom_bwd, custom_fwd  # noqa: F401  (kept for API parity with the reference module)

from .. import native


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False, is_rms_norm=False):
        x_shape_og = x.shape
        x = x.reshape(-1, x.shape[-1])
        if x.stride(-1) != 1:
            x = x.contiguous()
        if residual is not None:
            assert residual.shape == x_shape_og
            residual = residual.reshape(-1, residual.shape[-1])
            if residual.stride(-1) != 1:
                residual = residual.contiguous()
        weight = weight.contiguous()
        bias = bias.contiguous() if bias is not None else None
        residual_dtype = residual.dtype if residual is not None else (torch.float32 if residual_in_fp32 else None)
        y, mean, rstd, residual_out = native.layer_norm_fwd(x, weight, bias, eps, residual, residual_dtype=residual_dtype,
                                                            is_rms_norm=is_rms_norm)
        ctx.save_for_backward(residual_out, weight, bias, mean, rstd)
        ctx.x_shape_og, ctx.eps, ctx.is_rms_norm = x_shape_og, eps, is_rms_norm
        ctx.has_residual, ctx.prenorm, ctx.x_dtype = residual is not None, prenorm, x.dtype
        y = y.reshape(x_shape_og)
        return y if not prenorm else (y, residual_out.reshape(x_shape_og))

    @staticmethod
    def backward(ctx, dy, *args):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        dy = dy.reshape(-1, dy.shape[-1])
        if dy.stride(-1) != 1:
            dy = dy.contiguous()
        dresidual = None
        if ctx.prenorm:
            dresidual = args[0].reshape(-1, x.shape[-1])
            if dresidual.stride(-1) != 1:
                dresidual = dresidual.contiguous()
        dx, dw, db, dresidual_in = native.layer_norm_bwd(dy, x, weight, bias, ctx.eps, mean, rstd, dresidual,
                                                         ctx.has_residual, ctx.is_rms_norm, x_dtype=ctx.x_dtype)
        return (dx.reshape(ctx.x_shape_og), dw, db, dresidual_in.reshape(ctx.x_shape_og) if ctx.has_residual else None,
                None, None, None, None)


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False, is_rms_norm=False):
    return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)


def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
    return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, True)


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.empty(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        return rms_norm_fn(x, self.weight, self.bias, residual=residual, eps=self.eps, prenorm=prenorm,
                           residual_in_fp32=residual_in_fp32)
