"""causal_conv1d_fn -- same API as causal-conv1d/causal_conv1d/causal_conv1d_interface.py:8-45, on the HIP kernels."""
import torch

from .. import native


class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias=None, activation=None):
        if activation not in (None, "silu", "swish"):
            raise NotImplementedError("activation must be None, silu, or swish")
        if x.stride(2) != 1:
            x = x.contiguous()          # the reference also accepts channel-last; here it is made seqlen-contiguous
        bias = bias.contiguous() if bias is not None else None
        ctx.save_for_backward(x, weight, bias)
        ctx.activation = activation in ("silu", "swish")
        return native.causal_conv1d_fwd(x, weight, bias, ctx.activation)

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias = ctx.saved_tensors
        if dout.stride(2) != 1:
            dout = dout.contiguous()
        dx, dweight, dbias = native.causal_conv1d_bwd(x, weight, bias, dout, None, ctx.activation)
        return dx, dweight, dbias if bias is not None else None, None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """x: (batch, dim, seqlen), weight: (dim, width), bias: (dim,), activation: None | "silu" | "swish"."""
    return CausalConv1dFn.apply(x, weight, bias, activation)


def causal_conv1d_update(*args, **kwargs):
    raise NotImplementedError("causal_conv1d_update (single-token decode) is outside the denoiser hot path")
