"""GatedMLP -- dimsum/mlp.py:49-70: w3( act(x W12a + b) * (x W12b + b) ) + b3. With the tanh-GELU the DiM blocks use,
bias + activation + gate run as ONE fused HIP pass over the bias-free w12 GEMM output (csrc/token_transform.hip,
gated GeLU); the w3 bias can be handed to the caller's fused residual pass (`forward_deferred`). Keeping the biases out
of the GEMMs matters on gfx950: hipBLASLt serves bias-free fp32 matmuls under the reference's TF32 policy
(train.py:20-21) with its split-bf16 MFMA path (2.5x the fp32 rate at 4e-6 relative error), but not its bias-epilogue
kernels."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gemm, native


class _GatedGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x12, bias):
        x12 = x12.contiguous()
        ctx.save_for_backward(x12, bias)
        return native.gated_gelu_fwd(x12, bias)

    @staticmethod
    def backward(ctx, dh):
        x12, bias = ctx.saved_tensors
        dx12, dbias = native.gated_gelu_bwd(x12, bias, dh, need_dbias=bias is not None and ctx.needs_input_grad[1])
        return dx12, dbias


def gated_gelu(x12, bias=None):
    """gelu_tanh(x12[..., :H] + bias[:H]) * (x12[..., H:] + bias[H:])"""
    return _GatedGeluFn.apply(x12, bias)


class GatedMLP(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=F.gelu, drop=0.0, bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)
        self.act_layer = act_layer()
        self._fused = isinstance(self.act_layer, nn.GELU) and self.act_layer.approximate == "tanh"

    def forward_deferred(self, x, x3=None):
        """-> (y, b): the module's output is y + b; b (w3's bias or None) is left to the caller's fused residual pass.
        x3: the caller's producer kernel already wrote x as a split-bf16 operand image (gemm.py, split3): both GEMMs then
        run as plain bf16 GEMMs over the hi / lo images, the gated GeLU writes the w3 GEMM's image directly."""
        if x3 is not None:
            b12 = self.w12.bias
            h3 = native.gated_gelu_fwd(gemm.linear_split3(x3, self.w12.weight), None if b12 is None else b12.float(), split3=True)
            return gemm.linear_split3(h3, self.w3.weight).view(*x.shape[:-1], self.w3.weight.shape[0]), self.w3.bias
        if self._fused and x.dtype == torch.float32:
            b12 = self.w12.bias
            h = gated_gelu(gemm.linear(x, self.w12.weight), None if b12 is None else b12.float())
        else:
            x1, x2 = self.w12(x).chunk(2, dim=-1)
            h = self.act_layer(x1) * x2
        return gemm.linear(h, self.w3.weight), self.w3.bias

    def forward(self, x):
        y, b = self.forward_deferred(x)
        return y if b is None else y + b
