"""Fused residual-add + LayerNorm / RMSNorm on csrc/norm.hip.

Public names and call signatures are the drop-in contract of mamba/mamba_ssm/ops/triton/layernorm.py:367-486
(`LayerNormFn`, `layer_norm_fn`, `rms_norm_fn`, `RMSNorm`): what `models_dim.py:11,20-23` and `mamba_simple.py:36-39` import.
Semantics kept from there: any leading shape, rows = product of the leading dims; `prenorm=True` also returns the updated
residual stream (`x + residual`, in fp32 when `residual_in_fp32`); gradients for x, weight, bias and the residual."""
import torch

from .. import native


def _as_rows(t, width):
    """(..., width) -> (rows, width) with unit innermost stride (a view whenever the layout allows it)"""
    if t is None:
        return None
    t = t.reshape(-1, width)
    return t if t.stride(-1) == 1 else t.contiguous()


class LayerNormFn(torch.autograd.Function):
    """apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm) -> y | (y, residual_out)"""

    @staticmethod
    def forward(ctx, x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False, is_rms_norm=False):
        shape, width = x.shape, x.shape[-1]
        if residual is not None and residual.shape != shape:
            raise RuntimeError(f"layer norm: residual {tuple(residual.shape)} does not match x {tuple(shape)}")
        rows, res_rows = _as_rows(x, width), _as_rows(residual, width)
        res_dtype = res_rows.dtype if res_rows is not None else (torch.float32 if residual_in_fp32 else None)
        y, mean, rstd, stream = native.layer_norm_fwd(rows, weight.contiguous(), None if bias is None else bias.contiguous(), eps,
                                                      res_rows, residual_dtype=res_dtype, is_rms_norm=is_rms_norm)
        ctx.save_for_backward(stream, weight, bias, mean, rstd)          # `stream` = the normalised tensor (x + residual)
        ctx.meta = (shape, eps, is_rms_norm, residual is not None, prenorm, rows.dtype)
        return (y.view(shape), stream.reshape(shape)) if prenorm else y.view(shape)

    @staticmethod
    def backward(ctx, dy, *d_stream):
        stream, weight, bias, mean, rstd = ctx.saved_tensors
        shape, eps, is_rms_norm, has_residual, prenorm, x_dtype = ctx.meta
        width = shape[-1]
        dx, dw, db, dres = native.layer_norm_bwd(_as_rows(dy, width), stream, weight, bias, eps, mean, rstd,
                                                 _as_rows(d_stream[0], width) if prenorm else None, has_residual, is_rms_norm,
                                                 x_dtype=x_dtype)
        return dx.reshape(shape), dw, db, (dres.reshape(shape) if has_residual else None), None, None, None, None


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False, is_rms_norm=False):
    return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)


def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
    return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, True)


class RMSNorm(torch.nn.Module):
    """weight-only RMS norm module; state_dict key `weight` (the reference registers `bias` as None: no key)"""

    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        return rms_norm_fn(x, self.weight, self.bias, residual=residual, prenorm=prenorm, residual_in_fp32=residual_in_fp32, eps=self.eps)
