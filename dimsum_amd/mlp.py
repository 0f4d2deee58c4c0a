"""GatedMLP -- dimsum/mlp.py:49-70: w3( act(x W12a) * (x W12b) ). With the tanh-GELU the DiM blocks use, the
activation-and-gate epilogue runs as one fused HIP pass over the w12 output (csrc/token_transform.hip, gated GeLU)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native


class _GatedGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x12):
        x12 = x12.contiguous()
        ctx.save_for_backward(x12)
        return native.gated_gelu_fwd(x12)

    @staticmethod
    def backward(ctx, dh):
        (x12,) = ctx.saved_tensors
        return native.gated_gelu_bwd(x12, dh)


def gated_gelu(x12):
    return _GatedGeluFn.apply(x12)


class GatedMLP(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=F.gelu, drop=0.0, bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)
        self.act_layer = act_layer()
        self._fused = isinstance(self.act_layer, nn.GELU) and self.act_layer.approximate == "tanh"

    def forward(self, x):
        x12 = self.w12(x)
        if self._fused and x12.dtype == torch.float32:
            return self.w3(gated_gelu(x12))
        x1, x2 = x12.chunk(2, dim=-1)
        return self.w3(self.act_layer(x1) * x2)
