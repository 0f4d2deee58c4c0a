"""CrossAttentionFusion -- dimsum/attention_fusion.py:9-84: two qkv projections, swapped-KV attention
(x12 = softmax(q1 k2^T / sqrt(hd)) v2, x21 = softmax(q2 k1^T / sqrt(hd)) v1), concat, proj.
The attention core (both directions, straight from the bias-free qkv GEMM outputs to the concatenated proj input) is one
HIP kernel with MFMA QK^T / PV (csrc/xattn_fusion.hip); its backward is two more (csrc/xattn_fusion_bwd.hip). Variants
the published configs never enable (qk_norm, swap_k, attention dropout) run on torch SDPA on the GPU -- announced once and
counted (dimsum_amd.utils.note_torch_path); the plain variant on a shape without a HIP kernel is an error unless
DIMSUM_ALLOW_TORCH_SDPA=1."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gemm, native
from .utils import note_torch_path


class _XattnCoreFn(torch.autograd.Function):
    """fused (B, N, 2C) = cat(x12, x21) from bias-free qkv GEMM outputs + the qkv biases; with qkv2 = None plain
    self-attention (B, N, C) of qkv1 (the shared DiTBlock)"""

    @staticmethod
    def forward(ctx, qkv1, qkv2, bias1, bias2, heads):
        b1 = None if bias1 is None else bias1.float().contiguous()
        b2 = None if bias2 is None else bias2.float().contiguous()
        if not any(ctx.needs_input_grad[:4]):        # inference: no log-sum-exp to save
            return native.xattn_fusion_fwd(qkv1, qkv2, heads, bias1=b1, bias2=b2)
        out, lse = native.xattn_fusion_fwd(qkv1, qkv2, heads, need_lse=True, bias1=b1, bias2=b2)
        ctx.heads = heads
        ctx.f16 = gemm.attn_bwd_f16_enabled(qkv1)      # policy "f16s" under training: the backward pair on ONE fp16 product per element
        ctx.save_for_backward(qkv1, qkv2, b1, b2, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv1, qkv2, b1, b2, out, lse = ctx.saved_tensors
        dqkv1, dqkv2 = native.xattn_fusion_bwd(qkv1, qkv2, out, lse, dout, ctx.heads, bias1=b1, bias2=b2, f16=ctx.f16)
        W = dqkv1.shape[-1]
        db1 = dqkv1.reshape(-1, W).sum(0) if (b1 is not None and ctx.needs_input_grad[2]) else None
        db2 = dqkv2.reshape(-1, W).sum(0) if (b2 is not None and dqkv2 is not None and ctx.needs_input_grad[3]) else None
        return dqkv1, dqkv2, db1, db2, None


class CrossAttentionFusion(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, attn_drop=0.0, proj_drop=0.0,
                 norm_layer=nn.LayerNorm, swap_k=False):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        self.head_dim = dim // 2 // num_heads
        self.scale = self.head_dim ** -0.5
        self.swap_k = swap_k
        self.qkv1 = nn.Linear(dim // 2, dim // 2 * 3, bias=qkv_bias)
        self.q_norm1 = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm1 = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.qkv2 = nn.Linear(dim // 2, dim // 2 * 3, bias=qkv_bias)
        self.q_norm2 = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm2 = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self._plain = not qk_norm and not swap_k

    def _split(self, qkv, B, N):
        return qkv.reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)

    def forward(self, x1, x2):
        y, b = self.forward_deferred(x1, x2)
        return y if b is None else y + b

    def takes_images(self, ref):
        """whether forward_deferred(images=True) is served: the plain variant on the MFMA kernels, no dropout (ref: an fp32 activation)"""
        return self._plain and not (self.training and (self.attn_drop.p > 0.0 or self.proj_drop.p > 0.0)) and native.xattn_supported(ref, self.head_dim)

    def forward_deferred(self, x1, x2, images=False, residual=None):
        """-> (y, b): the module's output is y + b; b (proj's bias) is left to the caller's fused residual pass.
        images (inference under allow_tf32, see takes_images): x1 / x2 are split-bf16 operand images (B, N, 3 C) written by the
        branches' last pass; qkv and proj run as plain bf16 GEMMs over them and the attention kernel writes proj's image.
        residual (B, N, dim), with images: "hidden + proj(...) + b" in the proj GEMM's epilogue -- the call returns (that sum, None)."""
        if images:
            B, N, C3 = x1.shape
            b1, b2 = self.qkv1.bias, self.qkv2.bias
            if (b1 is None) != (b2 is None):
                raise RuntimeError("CrossAttentionFusion: qkv1 / qkv2 must both have a bias or none")
            # the qkv biases ride in the GEMMs' epilogues: the attention kernel then stages K / V without the adds
            bk1 = {} if b1 is None else {"bias": b1.float().contiguous()}
            bk2 = {} if b2 is None else {"bias": b2.float().contiguous()}
            f16 = isinstance(x1, native.F16Image) and (self.num_heads * self.head_dim) % 8 == 0
            qkv1 = qkv2 = None
            if f16:
                # scaled-fp16 policy: q | k | v leave the qkv GEMMs as scaled fp16 (the F16_QKV epilogue: no fp32 qkv tensors) where the shape allows
                kvb = gemm.attn_kv_bound(self.qkv1.weight, b1, self.qkv2.weight, b2)
                qkv1 = gemm.qkv_f16s(x1.reshape(B * N, C3), self.qkv1.weight, b1, N, kvb[0:2])
                qkv2 = gemm.qkv_f16s(x2.reshape(B * N, C3), self.qkv2.weight, b2, N, kvb[2:4]) if qkv1 is not None else None
            if qkv2 is None:
                qkv1 = gemm.linear_split3(x1.reshape(B * N, C3), self.qkv1.weight, **bk1)
                qkv2 = gemm.linear_split3(x2.reshape(B * N, C3), self.qkv2.weight, **bk2)
            qkv1, qkv2 = qkv1.view(B, N, -1), qkv2.view(B, N, -1)
            if f16:
                # ONE fp16 product per element in QK^T / PV too, proj's operand image written with the same kind of scale
                f3 = native.xattn_fusion_fwd(qkv1, qkv2, self.num_heads, split3="f16s", f16s=(x1.inv.reshape(B, N), x2.inv.reshape(B, N), kvb))
            else:
                f3 = native.xattn_fusion_fwd(qkv1, qkv2, self.num_heads, split_bf16=True, split3="pair" if isinstance(x1, native.PairImage) else True)
            if residual is not None:
                pb = None if self.proj.bias is None else self.proj.bias.float()
                y = gemm.linear_split3(f3.reshape(B * N, -1), self.proj.weight, bias=pb, residual=residual.reshape(B * N, -1))
                return y.view(residual.shape), None
            return gemm.linear_split3(f3.reshape(B * N, -1), self.proj.weight).view(B, N, -1), self.proj.bias
        B, N, C = x1.shape
        drop = self.attn_drop.p if self.training else 0.0
        if self._plain and drop == 0.0 and native.xattn_supported(x1, self.head_dim):
            # bias-free qkv GEMMs (fast hipBLASLt path); the biases are added inside the attention kernels
            b1, b2 = self.qkv1.bias, self.qkv2.bias
            if (b1 is None) != (b2 is None):
                raise RuntimeError("CrossAttentionFusion: qkv1 / qkv2 must both have a bias or none")
            fused = _XattnCoreFn.apply(gemm.linear(x1, self.qkv1.weight), gemm.linear(x2, self.qkv2.weight), b1, b2, self.num_heads)
        else:
            if self._plain and drop == 0.0:
                note_torch_path(f"CrossAttentionFusion core for head_dim {self.head_dim} / {x1.dtype}", required_opt_in=True)
            else:
                note_torch_path("CrossAttentionFusion variant (qk_norm / swap_k / attention dropout: unused by the published configs)")
            qkv1, qkv2 = self.qkv1(x1), self.qkv2(x2)
            q1, k1, v1 = self._split(qkv1, B, N)
            q2, k2, v2 = self._split(qkv2, B, N)
            q1, k1, q2, k2 = self.q_norm1(q1), self.k_norm1(k1), self.q_norm2(q2), self.k_norm2(k2)
            if not self.swap_k:
                x12 = F.scaled_dot_product_attention(q1, k2, v2, dropout_p=drop)
                x21 = F.scaled_dot_product_attention(q2, k1, v1, dropout_p=drop)
            else:
                x12 = F.scaled_dot_product_attention(q2, k1, v2, dropout_p=drop)
                x21 = F.scaled_dot_product_attention(q1, k2, v1, dropout_p=drop)
            fused = torch.cat((x12.transpose(1, 2).reshape(B, N, C), x21.transpose(1, 2).reshape(B, N, C)), dim=-1)
        if isinstance(self.proj_drop, nn.Dropout) and self.proj_drop.p > 0.0 and self.training:
            return self.proj_drop(self.proj(fused)), None
        return gemm.linear(fused, self.proj.weight), self.proj.bias
