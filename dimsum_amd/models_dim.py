"""DiM denoiser -- same module tree, constructor flags and state_dict keys as dimsum/models_dim.py
(DiM :1557-1930, DiMBlockCombined :974-1117, DiMBlockCombinedFourier :1120-1264, DiMBlockRaw :1402-1529,
WaveDiMBlock :505-710, DCTBlock :778-933, DiTBlock :1532-1554, embedders/FinalLayer :129-220, create_block :2001-2160,
zoo :2163-2236), composed from the HIP operators of dimsum_amd.ops.

What is structured differently (results equal to fp32 roundoff; pinned by tests/golden/*):
  * every chain of token reorders (transpose, continuity flip, sequence flip, 4x4-window scan, zigzag) is composed at
    construction into one gather table; the pre-mixer path  modulate(P(T(x)))  and the post-mixer path
    x + T^-1(P^-1(gate * mixer(...)))  are each ONE fused pass (ops/token_ops.py), T = Haar / DCT / identity;
  * the GatedMLP activation is a fused epilogue; RMSNorm is the HIP fused add+norm, not Triton.
Out of scope (constructor raises): block types linear/window/combined_einfft, MoE, rope/cpe positional encodings and
`enable_fourier_layers` -- unused by every published config.
"""
import math
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import scanning_orders as so
from . import gemm
from .attention_fusion import CrossAttentionFusion
from .dct_layer import init_dct_kernel, init_idct_kernel
from .mlp import GatedMLP
from .modules.mamba_simple import CondMamba, Mamba
from .ops import token_ops
from .ops.layernorm import RMSNorm, layer_norm_fn, rms_norm_fn
from .wavelet_layer import DWT_2D, IDWT_2D


def modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


# ---- fixed 2-D sin-cos positional embedding (models_dim.py:44-91, from MAE) ---------------------------------------------
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0))
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size):
    gh = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gh, gh), axis=0).reshape(2, 1, grid_size, grid_size)      # w first
    return np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                           get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)


def interpolate_pos_embed(model, checkpoint_model):
    """bicubic resize of a checkpoint's pos_embed to this model's grid (models_dim.py:99-121)."""
    if "pos_embed" not in checkpoint_model:
        return
    pe = checkpoint_model["pos_embed"]
    n_new = model.x_embedder.num_patches
    extra = model.pos_embed.shape[-2] - n_new
    old, new = int((pe.shape[-2] - extra) ** 0.5), int(n_new ** 0.5)
    if old != new:
        tok = pe[:, extra:].reshape(-1, old, old, pe.shape[-1]).permute(0, 3, 1, 2)
        tok = torch.nn.functional.interpolate(tok, size=(new, new), mode="bicubic", align_corners=False)
        checkpoint_model["pos_embed"] = torch.cat((pe[:, :extra], tok.permute(0, 2, 3, 1).flatten(1, 2)), dim=1)


# ---- embedders ----------------------------------------------------------------------------------------------------------
class PatchEmbed(nn.Module):
    """timm 0.9.12 PatchEmbed as used at models_dim.py:1620: Conv2d(k = s = patch) -> (B, T, D)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = nn.Identity()

    def forward(self, x):
        # Conv2d with kernel = stride = patch is a Linear over the flattened (c, i, j) patch: run it as ONE GEMM (MIOpen
        # serves this conv with a per-image im2col + GEMM pair, 2 x batch launches). `proj` stays a Conv2d for the
        # reference's state_dict layout (x_embedder.proj.weight (D, C, p, p)).
        B, C, H, W = x.shape
        ph, pw = self.patch_size
        patches = x.reshape(B, C, H // ph, ph, W // pw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // ph) * (W // pw), C * ph * pw)
        return F.linear(patches, self.proj.weight.reshape(self.proj.weight.shape[0], -1), self.proj.bias)


class TimestepEmbedder(nn.Module):
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        half = dim // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)   # on the device: HIP-graph capturable
        args = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb

    def forward(self, t):
        return self.mlp(self.timestep_embedding(t, self.frequency_embedding_size))


class LabelEmbedder(nn.Module):
    """class embedding with label dropout for classifier-free guidance (models_dim.py:170-202)."""

    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        self.in_channels = num_classes + int(dropout_prob > 0)
        self.embedding_table = nn.Embedding(self.in_channels, hidden_size)
        self.num_classes, self.dropout_prob = num_classes, dropout_prob

    def token_drop(self, labels, force_drop_ids=None):
        drop = torch.rand(labels.shape[0], device=labels.device) < self.dropout_prob if force_drop_ids is None else force_drop_ids == 1
        return torch.where(drop, self.num_classes, labels)

    def forward(self, labels, train, force_drop_ids=None):
        if (train and self.dropout_prob > 0) or force_drop_ids is not None:
            labels = self.token_drop(labels, force_drop_ids)
        return self.embedding_table(labels)

    def get_in_channels(self):
        return self.in_channels


# Two-stream branches (inference): read ONCE from DIMSUM_BRANCH_STREAMS when the module is imported; `branch_streams(False)` overrides it for
# the calls made inside the scope of the calling thread (bench.py's single-stream roofline pass) -- no per-forward environment lookups,
# no process-wide mutation from a measurement.
import contextvars as _contextvars

_BRANCH_STREAMS_DEFAULT = os.environ.get("DIMSUM_BRANCH_STREAMS", "1") != "0"
_branch_streams = _contextvars.ContextVar("dimsum_branch_streams", default=None)


def branch_streams_enabled():
    v = _branch_streams.get()
    return _BRANCH_STREAMS_DEFAULT if v is None else v


class branch_streams:
    """with branch_streams(False): the two branches of every combined block run on ONE stream inside the scope"""

    def __init__(self, enabled):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.token = _branch_streams.set(self.enabled)
        return self

    def __exit__(self, *exc):
        _branch_streams.reset(self.token)
        return False


def _ln_modulate(norm, x, shift, scale, split3=False):
    """modulate(LayerNorm(x), shift, scale) for the affine-free LayerNorms of DiTBlock / FinalLayer (models_dim.py:1536-1553,
    214-219). Inference on the GPU: ONE pass of the fused norm kernel (csrc/norm.hip with the modulation folded in) instead of
    torch's LayerNorm followed by a modulate pass; under autograd the two-step form, whose pieces have backward kernels.
    split3: the result as the split-bf16 operand image (B * L, 3H) of the Linear that consumes it (gemm.py)."""
    if torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32 or norm.weight is not None:
        return None
    from . import native
    B, L, H = x.shape
    ones = norm.__dict__.get("_dimsum_ones")
    if ones is None or ones.device != x.device:
        ones = norm.__dict__["_dimsum_ones"] = torch.ones(H, device=x.device, dtype=torch.float32)
    y = native.layer_norm_fwd(x.reshape(B * L, H), ones, None, norm.eps, is_rms_norm=False, mod_scale=scale, mod_shift=shift, rows_per_batch=L,
                              **({"split3": split3} if split3 else {}))[0]
    return y if split3 else y.view(B, L, H)


class FinalLayer(nn.Module):
    def __init__(self, hidden_size, patch_size, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))

    def forward(self, x, c):
        shift, scale = _modulation(self.adaLN_modulation, c).chunk(2, dim=1)
        h = _ln_modulate(self.norm_final, x, shift, scale)
        return self.linear(modulate(self.norm_final(x), shift, scale) if h is None else h)


class Attention(nn.Module):
    """timm 0.9.12 vision_transformer.Attention (qkv -> SDPA -> proj), as used by DiTBlock (models_dim.py:1540)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, **_):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward_deferred(self, x, x3=None, residual=None, gate=None):
        """-> (y, b): the module's output is y + b (proj's bias is left to the caller's fused residual pass).
        x3: x as a split-bf16 operand image written by the caller's norm kernel (gemm.py, split3); x then only carries the shape.
        residual (B, N, C) [, gate (B, C)] (with x3 on the MFMA attention kernels): the block's residual tail rides in the proj GEMM's
        epilogue -- the call returns (residual + gate * (attn(x) + b), None)."""
        from . import native
        from .attention_fusion import _XattnCoreFn
        B, N, C = x.shape
        if x3 is not None:
            # (the qkv bias rides in the GEMM's epilogue where the MFMA attention kernels follow: they then stage K / V without the adds)
            qb = None if self.qkv.bias is None else self.qkv.bias.float().contiguous()
            in_gemm = qb is not None and native.xattn_supported(x, self.head_dim)
            f16 = isinstance(x3, native.F16Image) and C % 8 == 0 and native.xattn_supported(x, self.head_dim)
            qkv = None
            if f16:     # scaled-fp16 policy: q | k | v as scaled fp16 out of the GEMM's epilogue (gemm.qkv_f16s) where the shape allows
                kvb = gemm.attn_kv_bound(self.qkv.weight, self.qkv.bias)
                qkv = gemm.qkv_f16s(x3.reshape(B * N, -1), self.qkv.weight, self.qkv.bias, N, kvb[0:2])
                if qkv is not None:
                    qkv, in_gemm = qkv.view(B, N, 3 * C), True
            if qkv is None:
                qkv = gemm.linear_split3(x3, self.qkv.weight, **({"bias": qb} if in_gemm else {})).view(B, N, 3 * C)
            if native.xattn_supported(x, self.head_dim):
                # the attention kernel writes the operand image of proj directly
                if in_gemm:
                    qb = None
                if f16:                                                     # the single-product attention kernel
                    o3 = native.xattn_fusion_fwd(qkv, None, self.num_heads, bias1=qb, split3="f16s", f16s=(x3.inv.reshape(B, N), None, kvb))
                else:
                    o3 = native.xattn_fusion_fwd(qkv, None, self.num_heads, bias1=qb, split_bf16=True, split3="pair" if isinstance(x3, native.PairImage) else True)
                if residual is not None:
                    pb = None if self.proj.bias is None else self.proj.bias.float()
                    y = gemm.linear_split3(o3.reshape(B * N, -1), self.proj.weight, bias=pb, residual=residual.reshape(B * N, C), gate=gate, rows_per_batch=N)
                    return y.view(B, N, C), None
                return gemm.linear_split3(o3.reshape(B * N, -1), self.proj.weight).view(B, N, C), self.proj.bias
        else:
            qkv = gemm.linear(x, self.qkv.weight)                     # bias-free GEMM (fast hipBLASLt path)
        if native.xattn_supported(qkv, self.head_dim):
            # MFMA self-attention core (csrc/xattn_fusion*.hip, n_dirs = 1); the qkv bias is added inside the kernels
            o = _XattnCoreFn.apply(qkv, None, self.qkv.bias, None, self.num_heads)
        else:
            from .utils import note_torch_path
            note_torch_path(f"Attention (shared DiTBlock) core for head_dim {self.head_dim} / {qkv.dtype}", required_opt_in=True)
            if self.qkv.bias is not None:
                qkv = qkv + self.qkv.bias
            q, k, v = qkv.reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
            o = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C)
        return gemm.linear(o, self.proj.weight), self.proj.bias

    def forward(self, x):
        y, b = self.forward_deferred(x)
        return y if b is None else y + b


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def drop_path(x, drop_prob=0.0, training=False, scale_by_keep=True):
    if drop_prob == 0.0 or not training:
        return x
    keep = 1 - drop_prob
    mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
    if keep > 0.0 and scale_by_keep:
        mask.div_(keep)
    return x * mask


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training, self.scale_by_keep)


_approx_gelu = lambda: nn.GELU(approximate="tanh")  # noqa: E731


def _make_mlp(dim, use_gated_mlp=True):
    cls = GatedMLP if use_gated_mlp else Mlp
    return cls(in_features=dim, hidden_features=int(dim * 4), act_layer=_approx_gelu, drop=0)


def _modulation(seq, c):
    """seq(c) for an adaLN head `nn.Sequential(nn.SiLU(), nn.Linear)` (models_dim.py:1455, 1544): every block of a DiM applies the SAME SiLU to the
    same conditioning vector -- inside one DiM forward it is computed once (54 launches per DiM-L/2 forward otherwise), bit-identical"""
    hit = getattr(gemm._tls, "cond", None)
    if hit is not None and hit[0] is c and len(seq) == 2 and isinstance(seq[0], nn.SiLU):
        return seq[1](hit[1])
    return seq(c)


def _mlp_tail(mlp, x, normed, shift, scale, gate):
    """x + gate * mlp(modulate(normed, shift, scale))  (models_dim.py:1111-1115, 1551-1553)"""
    if getattr(mlp, "_fused", False) and gemm.split3_train_enabled(normed, mlp.w12.weight):
        from .mlp import mod_gated_mlp_images
        m, mb = mod_gated_mlp_images(mlp, normed, shift, scale)              # training: every MLP GEMM on operand images
        return token_ops.gate_residual(x, m, gate, mb)
    h = token_ops.pre_mixer(normed, "none", None, shift, scale)               # modulate, one pass
    if hasattr(mlp, "forward_deferred"):
        m, mb = mlp.forward_deferred(h)
        return token_ops.gate_residual(x, m, gate, mb)
    return token_ops.gate_residual(x, mlp(h), gate, None)


def _mix_through_images(mixer, hidden_states, kind, table, shift, scale, c, fork=False):
    """mixer(pre_mixer(hidden_states)); at inference under allow_tf32 the pre-mixer pass writes the in_proj operand as a
    split-bf16 image (gemm.py, split3) instead of fp32.
    fork: -> (mixer output, hidden_states for the residual tail). Under autograd the second is an alias handed out by the pre-mixer
    pass itself, whose backward kernel then adds the tail's gradient to its own (token_ops.pre_mixer_fork): hidden_states has ONE consumer."""
    if getattr(mixer, "takes_image", lambda: False)() and gemm.split3_enabled(hidden_states, mixer.in_proj.weight, left=False):
        m = mixer(None, c, x3=token_ops.pre_mixer(hidden_states, kind, table, shift, scale,
                                                  split3=gemm.split3_enabled(hidden_states, mixer.in_proj.weight, left=False)))
        return (m, hidden_states) if fork else m
    if fork and hidden_states.is_cuda and torch.is_grad_enabled() and hidden_states.requires_grad:
        t, hidden_states = token_ops.pre_mixer_fork(hidden_states, kind, table, shift, scale)
        return mixer(t, c), hidden_states
    m = mixer(token_ops.pre_mixer(hidden_states, kind, table, shift, scale), c)
    return (m, hidden_states) if fork else m


# ---- shared block plumbing ----------------------------------------------------------------------------------------------
class _BlockBase(nn.Module):
    """Add -> Norm prologue shared by all blocks (e.g. models_dim.py:1460-1494) + cached gather tables."""

    def _prenorm(self, hidden_states, residual):
        if not self.fused_add_norm:
            residual = hidden_states if residual is None else residual + self.drop_path(hidden_states)
            hidden_states = (self.norm(residual) if isinstance(self.norm, nn.Identity)
                             else self.norm(residual.to(dtype=self.norm.weight.dtype)))
            if self.residual_in_fp32:
                residual = residual.to(torch.float32)
            return hidden_states, residual
        fn = rms_norm_fn if isinstance(self.norm, RMSNorm) else layer_norm_fn
        x = hidden_states if residual is None else self.drop_path(hidden_states)
        return fn(x, self.norm.weight, self.norm.bias, residual=residual, prenorm=True,
                  residual_in_fp32=self.residual_in_fp32, eps=self.norm.eps)

    def _table(self, L, device, build):
        """gather tables for sequence length L, built once per (L, device): fwd (int64), inv (int64), inv32 (int32)"""
        cache = self.__dict__.setdefault("_tables", {})
        key = (L, str(device))
        if key not in cache:
            fwd = build(math.isqrt(L))
            # a zigzag mixer (scan_type zigma_N / sweep_N / jpeg_N) gathers its tokens once more by its layer's path
            # (mamba_simple.py:627-657): composed into this block's table, the mixer then skips its two gathers
            mixer = getattr(self, "mixer", None)
            if mixer is not None and getattr(mixer, "_is_zigzag", lambda: False)() and getattr(mixer, "zigzag_paths", None) is not None \
                    and os.environ.get("DIMSUM_FOLD_ZIGZAG", "1") != "0":
                perm = mixer.zigzag_paths[mixer.layer_idx].detach().cpu().numpy().astype(np.int64)
                assert perm.shape[0] == L, "zigzag path length != sequence length"
                fwd = perm if fwd is None else so.compose(fwd, perm)
                mixer._zigzag_folded = True
            if fwd is None:
                cache[key] = None
            else:
                inv = so.reverse_permut_np(fwd)
                cache[key] = {"fwd": so.as_index(fwd, device), "inv": so.as_index(inv, device),
                              "inv32": torch.as_tensor(inv.astype(np.int32), device=device)}
        return cache[key]

    def allocate_inference_cache(self, *a, **k):
        raise NotImplementedError("autoregressive decode caches are outside the denoiser hot path")


class DiMBlockRaw(_BlockBase):
    """spatial Mamba branch: reorder -> h + gate * mixer(modulate(h)) -> undo (models_dim.py:1402-1529)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False, drop_path=0.0,
                 reverse=False, transpose=False, scanning_continuity=False, c_dim=None):
        super().__init__()
        self.residual_in_fp32, self.fused_add_norm = residual_in_fp32, fused_add_norm
        self.reverse, self.transpose, self.scanning_continuity = reverse, transpose, scanning_continuity
        c_dim = dim if c_dim is None else c_dim
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(c_dim, 3 * dim, bias=True))

    def _order(self, H):
        if not (self.reverse or self.transpose or self.scanning_continuity):
            return None
        return so.block_order_table(H, self.reverse, self.transpose, self.scanning_continuity)

    def forward(self, hidden_states, residual=None, c=None, inference_params=None, out_split3=False):
        """out_split3 (inference, set by an enclosing combined block): the result as the split-bf16 operand image of the qkv Linear"""
        hidden_states, residual = self._prenorm(hidden_states, residual)
        table = self._table(hidden_states.shape[1], hidden_states.device, self._order)
        shift, scale, gate = _modulation(self.adaLN_modulation, c).chunk(3, dim=1)
        m, hidden_states = _mix_through_images(self.mixer, hidden_states, "none", table, shift, scale, c, fork=True)
        return token_ops.post_mixer(hidden_states, m, gate, "none", table, **({"split3": out_split3} if out_split3 else {})), residual


class _FreqBlock(_BlockBase):
    """frequency branch shared by WaveDiMBlock (Haar) and DCTBlock (DCT): T -> reorder -> mixer -> undo -> T^-1."""
    kind = "none"

    def _init_common(self, dim, mixer_cls, norm_cls, fused_add_norm, residual_in_fp32, drop_path, reverse, transpose,
                     scanning_continuity, no_ffn, c_dim):
        self.residual_in_fp32, self.fused_add_norm = residual_in_fp32, fused_add_norm
        self.reverse, self.transpose, self.scanning_continuity, self.no_ffn = reverse, transpose, scanning_continuity, no_ffn
        self._c_dim = dim if c_dim is None else c_dim

    def _finish_common(self, dim, norm_cls, drop_path):
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(self._c_dim, 6 * dim if not self.no_ffn else 3 * dim, bias=True))
        if not self.no_ffn:
            self.norm_2 = norm_cls(dim)
            self.mlp = _make_mlp(dim)

    def forward(self, hidden_states, residual=None, c=None, inference_params=None, out_split3=False):
        hidden_states, residual = self._prenorm(hidden_states, residual)
        table = self._table(hidden_states.shape[1], hidden_states.device, self._order)
        mods = _modulation(self.adaLN_modulation, c).chunk(3 if self.no_ffn else 6, dim=1)
        shift, scale, gate = mods[:3]
        if self.no_ffn:
            m, hidden_states = _mix_through_images(self.mixer, hidden_states, self.kind, table, shift, scale, c, fork=True)
            return token_ops.post_mixer(hidden_states, m, gate, self.kind, table, **({"split3": out_split3} if out_split3 else {})), residual
        assert not out_split3
        # with an FFN the reference keeps working in the transformed / reordered token space (models_dim.py:678-684)
        t = token_ops.pre_mixer(hidden_states, self.kind, table, torch.zeros_like(shift), torch.zeros_like(scale))
        t = t + gate.unsqueeze(1) * self.mixer(modulate(t, shift, scale), c)
        t = t + mods[5].unsqueeze(1) * self.mlp(modulate(self.norm_2(t), mods[3], mods[4]))
        zero = torch.zeros_like(hidden_states)
        return token_ops.post_mixer(zero, t, torch.ones_like(gate), self.kind, table), residual


class WaveDiMBlock(_FreqBlock):
    """2-level Haar branch (models_dim.py:505-710)."""
    kind = "haar"

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False, drop_path=0.0,
                 reverse=False, transpose=False, scanning_continuity=False, skip=False, no_ffn=False, c_dim=None,
                 window_scan=True, num_wavelet_lv=2):
        super().__init__()
        assert num_wavelet_lv == 2, "only the two-level transform of the published configs is implemented"
        self._init_common(dim, mixer_cls, norm_cls, fused_add_norm, residual_in_fp32, drop_path, reverse, transpose,
                          scanning_continuity, no_ffn, c_dim)
        self.window_scan, self.num_wavelet_lv = window_scan, num_wavelet_lv
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)
        self.dwt, self.idwt = DWT_2D(wave="haar"), IDWT_2D(wave="haar")
        self._finish_common(dim, norm_cls, drop_path)

    def _order(self, H):
        if self.window_scan:      # local_scan(w = W // 4, column_first = transpose)  (models_dim.py:659-664)
            tab = so.local_scan_table(H, H // 4, column_first=bool(self.transpose))
        else:
            tab = so.block_order_table(H, False, self.transpose, False)
        return so.compose(tab, so.block_order_table(H, self.reverse, False, self.scanning_continuity))


class DCTBlock(_FreqBlock):
    """4x4 block-DCT branch (models_dim.py:778-933)."""
    kind = "dct"

    def __init__(self, dim, mixer_cls, fused_add_norm=False, residual_in_fp32=False, drop_path=0.0, norm_cls=nn.LayerNorm,
                 dct_size=2, reverse=False, transpose=False, scanning_continuity=False, no_ffn=False, c_dim=None):
        super().__init__()
        assert dct_size == 4, "only the 4x4 DCT of DiMBlockCombinedFourier is implemented"
        self._init_common(dim, mixer_cls, norm_cls, fused_add_norm, residual_in_fp32, drop_path, reverse, transpose,
                          scanning_continuity, no_ffn, c_dim)
        self.dim, self.dct_size, self.reserve_kernel = dim, dct_size, dct_size
        self.norm = norm_cls(dim)
        self.mixer = mixer_cls(dim)
        self._finish_common(dim, norm_cls, drop_path)
        self.dct_conv = init_dct_kernel(dim, dct_size, dct_size)
        self.idct_conv = nn.Sequential(init_idct_kernel(dim, dct_size, dct_size), nn.PixelShuffle(dct_size))

    def _order(self, H):
        if not (self.reverse or self.transpose or self.scanning_continuity):
            return None
        return so.block_order_table(H, self.reverse, self.transpose, self.scanning_continuity)


class _ForkHalves(torch.autograd.Function):
    """hidden -> (first channel half, second half, hidden): the branch inputs and the stream of the fusion's residual tail. The backward joins
    the three gradients in two strided adds -- 3 x (B, L, dim) of traffic instead of the 5 x of the engine's cat + add."""

    @staticmethod
    def forward(ctx, h):
        C = h.shape[-1] // 2
        return h[..., :C], h[..., C:], h.view_as(h)

    @staticmethod
    def backward(ctx, d1, d2, dres):
        if dres is None:
            return torch.cat((d1, d2), dim=-1)
        C = dres.shape[-1] // 2
        out = torch.empty(dres.shape, device=dres.device, dtype=dres.dtype)
        for d, sl in ((d1, slice(0, C)), (d2, slice(C, 2 * C))):
            if d is None:
                out[..., sl] = dres[..., sl]
            else:
                torch.add(d, dres[..., sl], out=out[..., sl])
        return out


class _CombinedBase(_BlockBase):
    def _init_tail(self, dim, norm_cls, drop_path, use_gated_mlp, swap_k_kw):
        self.proj = CrossAttentionFusion(dim, num_heads=8, qkv_bias=True, **swap_k_kw)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm_2 = norm_cls(dim)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(dim, 3 * dim, bias=True))
        self.mlp = _make_mlp(dim, use_gated_mlp)

    def forward(self, hidden_states, residual=None, c=None, inference_params=None):
        hidden_states, residual = self._prenorm(hidden_states, residual)
        if hidden_states.is_cuda and torch.is_grad_enabled() and hidden_states.requires_grad:
            x1, x2, hidden_states = _ForkHalves.apply(hidden_states)        # training: ONE consumer of hidden_states in the graph (see there)
        else:
            x1, x2 = hidden_states.chunk(2, dim=2)
        # inference under allow_tf32: the branches hand their results over as split-bf16 operand images of the qkv Linears
        img = self.proj.takes_images(hidden_states) and gemm.split3_enabled(x1, self.proj.qkv1.weight)      # False / True / "f16s"
        kw = {"out_split3": img} if img else {}
        if ((not torch.is_grad_enabled()) and hidden_states.is_cuda and branch_streams_enabled()
                and not torch.cuda.is_current_stream_capturing()):
            # inference: the two branches are independent until the fusion -- the frequency branch on a second HIP stream lets the
            # memory-bound passes of one branch (conv1d, scan, token passes) overlap the GEMMs of the other: -1.3 .. -2.4 % per
            # forward, bit-identical (tests/test_model_gpu.py::test_two_stream_branches_are_bit_identical). DIMSUM_BRANCH_STREAMS=0
            # keeps one stream: kernels that share the chip cannot be timed individually (the scan's launch-to-launch time goes
            # 0.33 -> 0.52 ms), so bench.py measures its per-kernel roofline in a short single-stream pass and says so on the line.
            cur = torch.cuda.current_stream(hidden_states.device)
            side = self.__dict__.get("_side_stream")
            if side is None or side.device != hidden_states.device:
                side = self.__dict__["_side_stream"] = torch.cuda.Stream(device=hidden_states.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                x2, _ = self.freq_mamba(x2, None, c, inference_params, **kw)
            x1, _ = self.spatial_mamba(x1, None, c, inference_params, **kw)
            cur.wait_stream(side)
            x2.record_stream(cur)       # allocated on the side stream, consumed (and released) on the current one
        else:
            x1, _ = self.spatial_mamba(x1, None, c, inference_params, **kw)
            x2, _ = self.freq_mamba(x2, None, c, inference_params, **kw)
        # residual tails as single fused passes; the Linear biases ride along (mlp.py / attention_fusion.py docstrings)
        fast_tail = (not torch.is_grad_enabled() and isinstance(self.norm_2, RMSNorm) and hasattr(self.mlp, "forward_deferred")
                     and hidden_states.dtype == torch.float32)
        # (inference on operand images: "h + proj(..) + b" already in the proj GEMM's epilogue, the norm pass then reads ONE tensor)
        in_epilogue = bool(img) and fast_tail and hidden_states.is_contiguous()
        fused, pb = self.proj.forward_deferred(x1, x2, **({"images": True} if img else {}), **({"residual": hidden_states} if in_epilogue else {}))
        shift, scale, gate = _modulation(self.adaLN_modulation, c).chunk(3, dim=1)
        if fast_tail:
            # inference: h' = h + proj(..) + b, RMSNorm(h'), modulate -- ONE pass (csrc/norm.hip with x_bias + modulation)
            from . import native
            B, L, H = hidden_states.shape
            # ... written directly as the split-bf16 operand image of the w12 GEMM when the library would split it anyway (gemm.py)
            s3 = getattr(self.mlp, "_fused", False) and gemm.split3_enabled(hidden_states, self.mlp.w12.weight, producer="norm")   # False / True / "f16s"
            y, _, _, hnew = native.layer_norm_fwd(fused.reshape(B * L, H), self.norm_2.weight, self.norm_2.bias, self.norm_2.eps,
                                                  residual=None if in_epilogue else hidden_states.reshape(B * L, H), is_rms_norm=True, x_bias=pb,
                                                  mod_scale=scale, mod_shift=shift, rows_per_batch=L, **({"split3": s3} if s3 else {}))
            if s3 and hnew.is_contiguous():       # ... and the residual tail "h + gate * (mlp + b)" in the epilogue of the w3 GEMM
                return self.mlp.forward_deferred(hidden_states, x3=y, residual=hnew.view(B, L, H), gate=gate)[0], residual
            m, mb = self.mlp.forward_deferred(hidden_states, x3=y) if s3 else self.mlp.forward_deferred(y.view(B, L, H))
            return token_ops.gate_residual(hnew.view(B, L, H), m, gate, mb), residual
        hidden_states = token_ops.gate_residual(hidden_states, fused, None, pb)
        if isinstance(self.norm_2, RMSNorm) and hidden_states.is_cuda and torch.is_grad_enabled():
            # training: the norm hands the stream on (prenorm), so that "d norm + d tail" is formed inside the norm's backward kernel
            # (its dresidual input) instead of by an add of two (B, L, dim) gradients in the autograd engine
            normed, hidden_states = self.norm_2(hidden_states, prenorm=True)
            return _mlp_tail(self.mlp, hidden_states, normed, shift, scale, gate), residual
        return _mlp_tail(self.mlp, hidden_states, self.norm_2(hidden_states), shift, scale, gate), residual


class DiMBlockCombined(_CombinedBase):
    """spatial Mamba || Haar-frequency Mamba -> cross-attention fusion -> gated MLP (models_dim.py:974-1117)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False, drop_path=0.0,
                 reverse=False, transpose=False, scanning_continuity=False, use_gated_mlp=True):
        super().__init__()
        self.residual_in_fp32, self.fused_add_norm = residual_in_fp32, fused_add_norm
        self.reverse, self.transpose, self.scanning_continuity = reverse, transpose, scanning_continuity
        self.norm = norm_cls(dim)
        kw = dict(norm_cls=nn.Identity, drop_path=0.0, fused_add_norm=False, residual_in_fp32=residual_in_fp32,
                  scanning_continuity=scanning_continuity, c_dim=dim)
        self.spatial_mamba = DiMBlockRaw(dim // 2, mixer_cls, reverse=reverse, transpose=transpose, **kw)
        self.freq_mamba = WaveDiMBlock(dim // 2, mixer_cls, reverse=False, transpose=reverse, no_ffn=True, num_wavelet_lv=2, **kw)
        self._init_tail(dim, norm_cls, drop_path, use_gated_mlp, dict(swap_k=False))


class DiMBlockCombinedFourier(_CombinedBase):
    """spatial Mamba || DCT-frequency Mamba (jpeg_2 zigzag mixer) (models_dim.py:1120-1264)."""

    def __init__(self, dim, mixer_cls, mixer_cls_2, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False,
                 drop_path=0.0, reverse=False, transpose=False, scanning_continuity=False, use_gated_mlp=True):
        super().__init__()
        self.residual_in_fp32, self.fused_add_norm = residual_in_fp32, fused_add_norm
        self.reverse, self.transpose, self.scanning_continuity = reverse, transpose, scanning_continuity
        self.norm = norm_cls(dim)
        kw = dict(norm_cls=nn.Identity, drop_path=0.0, fused_add_norm=False, residual_in_fp32=residual_in_fp32,
                  scanning_continuity=scanning_continuity, c_dim=dim)
        self.spatial_mamba = DiMBlockRaw(dim // 2, mixer_cls, reverse=reverse, transpose=transpose, **kw)
        self.freq_mamba = DCTBlock(dim // 2, mixer_cls_2, reverse=False, transpose=False, no_ffn=True, dct_size=4, **kw)
        self._init_tail(dim, norm_cls, drop_path, use_gated_mlp, {})


class DiTBlock(nn.Module):
    """adaLN-Zero transformer block shared every k layers (models_dim.py:1532-1554)."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, use_gated_mlp=True, **block_kwargs):
        super().__init__()
        self.norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.attn = Attention(hidden_size, num_heads=num_heads, qkv_bias=True, **block_kwargs)
        self.norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        cls = GatedMLP if use_gated_mlp else Mlp
        self.mlp = cls(in_features=hidden_size, hidden_features=int(hidden_size * mlp_ratio), act_layer=_approx_gelu, drop=0)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))

    def forward(self, x, c=None, **kwargs):
        sa, ca, ga, sm, cm, gm = _modulation(self.adaLN_modulation, c).chunk(6, dim=1)
        s3 = gemm.split3_enabled(x, self.attn.qkv.weight, producer="norm")      # inference under allow_tf32: the norm passes write operand images
        h = _ln_modulate(self.norm1, x, sa, ca, split3=s3)
        from . import native
        if h is not None and s3 and x.is_contiguous() and native.xattn_supported(x, self.attn.head_dim):
            x = self.attn.forward_deferred(x, x3=h, residual=x, gate=ga)[0]      # "x + gate_msa * attn(..)" in the proj GEMM's epilogue
        else:
            if h is not None and s3:
                a, ab = self.attn.forward_deferred(x, x3=h)
            else:
                a, ab = self.attn.forward_deferred(token_ops.pre_mixer(self.norm1(x), "none", None, sa, ca) if h is None else h)
            x = token_ops.gate_residual(x, a, ga, ab)
        s3 = getattr(self.mlp, "_fused", False) and s3        # (keeps the mode: False / True / "f16s")
        h = _ln_modulate(self.norm2, x, sm, cm, split3=s3)
        if h is None:
            return _mlp_tail(self.mlp, x, self.norm2(x), sm, cm, gm)
        if hasattr(self.mlp, "forward_deferred"):
            if s3 and x.is_contiguous():
                return self.mlp.forward_deferred(x, x3=h, residual=x, gate=gm)[0]
            m, mb = self.mlp.forward_deferred(x, x3=h) if s3 else self.mlp.forward_deferred(h)
            return token_ops.gate_residual(x, m, gm, mb)
        return token_ops.gate_residual(x, self.mlp(h), gm, None)


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True, n_residuals_per_layer=1):
    """GPT-2 style init (models_dim.py:1969-1998): zero Linear biases (except dt_proj), scaled out_proj/fc2."""
    if isinstance(module, nn.Linear):
        if module.bias is not None and not getattr(module.bias, "_no_reinit", False):
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        for name, p in module.named_parameters():
            if name in ("out_proj.weight", "fc2.weight"):
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                with torch.no_grad():
                    p /= math.sqrt(n_residuals_per_layer * n_layer)


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, drop_path=0.0, rms_norm=False, residual_in_fp32=True,
                 fused_add_norm=False, layer_idx=None, device=None, dtype=None, scan_type="none", add_bias_linear=False,
                 gated_linear_unit=True, routing_mode="sinkhorn", num_moe_experts=8, mamba_moe_layers=None, is_moe=False,
                 block_type="linear", reverse=False, transpose=False, cond_mamba=False, scanning_continuity=False,
                 skip=False, use_gated_mlp=True, block_kwargs={}, block_kwargs2={}):
    if is_moe:
        raise NotImplementedError("MoE blocks are outside the denoiser hot path (never enabled by a published config)")
    ssm_cfg = ssm_cfg or {}
    fk = {"device": device, "dtype": dtype}
    norm_cls = partial(nn.LayerNorm if not rms_norm else RMSNorm, eps=norm_epsilon, **fk)
    if cond_mamba:
        # the reference passes scan_type twice here when block_kwargs carries one (SURVEY finding 2); block_kwargs wins
        kw = dict(layer_idx=layer_idx, scan_type=scan_type, d_cond=d_model, **ssm_cfg, **fk)
        kw.update(block_kwargs)
        mixer_cls = partial(CondMamba, **kw)
    else:
        mixer_cls = partial(Mamba, layer_idx=layer_idx, scan_type=scan_type, **ssm_cfg, **fk)
    common = dict(norm_cls=norm_cls, drop_path=drop_path, fused_add_norm=fused_add_norm, residual_in_fp32=residual_in_fp32,
                  scanning_continuity=scanning_continuity)
    if block_type == "raw":
        block = DiMBlockRaw(d_model, mixer_cls, reverse=reverse, transpose=transpose, **common)
    elif block_type == "wave":
        block = WaveDiMBlock(d_model, mixer_cls, reverse=reverse, transpose=transpose, skip=skip, window_scan=False, **common)
    elif block_type == "combined":
        block = DiMBlockCombined(d_model, mixer_cls, reverse=reverse, transpose=transpose, use_gated_mlp=use_gated_mlp, **common)
    elif block_type == "combined_fourier":
        mixer_cls_2 = partial(CondMamba, layer_idx=layer_idx, d_cond=d_model, **ssm_cfg, **block_kwargs2, **fk)
        block = DiMBlockCombinedFourier(d_model, mixer_cls, mixer_cls_2, reverse=reverse, transpose=transpose,
                                        use_gated_mlp=use_gated_mlp, **common)
    else:
        raise NotImplementedError(f"block_type={block_type!r} is outside the denoiser hot path "
                                  "(published configs use 'combined'; also available: raw, wave, combined_fourier)")
    block.layer_idx = layer_idx
    return block


class DiM(nn.Module):
    def __init__(self, img_resolution=32, patch_size=2, in_channels=4, hidden_size=1024, depth=16, label_dropout=0.1,
                 num_classes=1000, learn_sigma=False, ssm_cfg=None, rms_norm=False, residual_in_fp32=True,
                 fused_add_norm=False, scan_type="none", initializer_cfg=None, num_moe_experts=8, mamba_moe_layers=None,
                 add_bias_linear=False, gated_linear_unit=True, routing_mode="top1", is_moe=False, pe_type="ape",
                 block_type="linear", cond_mamba=False, scanning_continuity=False, enable_fourier_layers=False,
                 learnable_pe=False, skip=False, drop_path=0.0, use_final_norm=False, use_attn_every_k_layers=-1,
                 use_gated_mlp=True, use_independent_attn=False):
        super().__init__()
        if pe_type != "ape":
            raise NotImplementedError("only the absolute positional embedding of the published configs is implemented")
        if enable_fourier_layers:
            raise NotImplementedError("enable_fourier_layers is broken in the reference (models_dim.py:1702) and unused")
        self.depth = int(depth * 3) if block_type == "raw" else depth
        self.learn_sigma, self.in_channels = learn_sigma, in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.patch_size, self.num_classes, self.initializer_cfg = patch_size, num_classes, initializer_cfg
        self.enable_fourier_layers, self.fused_add_norm, self.residual_in_fp32 = False, fused_add_norm, residual_in_fp32
        self.use_attn_every_k_layers, self.use_independent_attn = use_attn_every_k_layers, use_independent_attn
        self.pe_type, self.block_type = pe_type, block_type
        if use_independent_attn:
            n_tb = self.depth // use_attn_every_k_layers - 1
            self.depth = self.depth - self.depth // use_attn_every_k_layers

        self.x_embedder = PatchEmbed(img_resolution, patch_size, in_channels, hidden_size)
        self.t_embedder = TimestepEmbedder(hidden_size)
        self.y_embedder = LabelEmbedder(num_classes, hidden_size, label_dropout)
        num_patches = self.x_embedder.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, hidden_size), requires_grad=learnable_pe)
        dpr = [x.item() for x in torch.linspace(0, drop_path, self.depth, device="cpu")]
        inter_dpr = [0.0] + dpr
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        grid = int(math.sqrt(num_patches))

        def gen_paths(N, st):           # models_dim.py:1640-1658
            kind, n = st.split("_")[0], int(st.split("_")[1])
            zz = so.SCAN_ZOO[kind](N)[:n]
            rev = [so.reverse_permut_np(p) for p in zz]
            return dict(zigzag_paths=torch.cat([torch.from_numpy(p)[None] for p in zz] * self.depth, dim=0),
                        zigzag_paths_reverse=torch.cat([torch.from_numpy(p)[None] for p in rev] * self.depth, dim=0),
                        scan_type=st)

        block_kwargs = gen_paths(grid, scan_type) if scan_type.startswith(("zigma", "sweep", "jpeg")) else {}
        block_kwargs2 = gen_paths(grid, "jpeg_2")       # fixed (models_dim.py:1664)
        self.blocks = nn.ModuleList([
            create_block(hidden_size, ssm_cfg=ssm_cfg, norm_epsilon=1e-5, rms_norm=rms_norm, residual_in_fp32=residual_in_fp32,
                         fused_add_norm=fused_add_norm, layer_idx=i, scan_type=scan_type, drop_path=inter_dpr[i],
                         num_moe_experts=num_moe_experts, mamba_moe_layers=mamba_moe_layers, add_bias_linear=add_bias_linear,
                         gated_linear_unit=gated_linear_unit, routing_mode=routing_mode, is_moe=is_moe, block_type=block_type,
                         reverse=(scan_type == "none") and (i % 2 > 0), transpose=(scan_type == "none") and (i % 4 >= 2),
                         cond_mamba=cond_mamba, scanning_continuity=scanning_continuity, use_gated_mlp=use_gated_mlp,
                         block_kwargs=block_kwargs, block_kwargs2=block_kwargs2)
            for i in range(self.depth)])
        if use_attn_every_k_layers > 0:
            if use_independent_attn:
                self.attn_block = nn.ModuleList([DiTBlock(hidden_size, 16, use_gated_mlp=use_gated_mlp) for _ in range(n_tb)])
            else:
                self.attn_block = DiTBlock(hidden_size, 16, use_gated_mlp=use_gated_mlp)
        self.norm_f = (nn.LayerNorm if not rms_norm else RMSNorm)(hidden_size, eps=1e-5) if use_final_norm else None
        self.final_layer = FinalLayer(hidden_size, patch_size, self.out_channels)
        self.initialize_weights()

    def initialize_weights(self):
        """models_dim.py:1744-1779 (adaLN-zero: a freshly initialised model outputs exactly 0)."""
        pe = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.x_embedder.num_patches ** 0.5))
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.constant_(self.x_embedder.proj.bias, 0)
        nn.init.normal_(self.y_embedder.embedding_table.weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        for block in self.blocks:
            nn.init.constant_(block.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(block.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].weight, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)
        self.apply(partial(_init_weights, n_layer=self.depth, **(self.initializer_cfg or {})))

    def unpatchify(self, x):
        c, p = self.out_channels, self.x_embedder.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(x.shape[0], h, w, p, p, c)
        return torch.einsum("nhwpqc->nchpwq", x).reshape(x.shape[0], c, h * p, h * p)

    def forward(self, x, t, y=None, inference_params=None, **kwargs):
        """x: (N, C, H, W) latents, t: (N,) times, y: (N,) labels -> (N, out_channels, H, W)."""
        if t is None:
            t = torch.randint(0, 1000, (x.shape[0],), device=x.device)
        if y is None:
            y = torch.ones(x.size(0), dtype=torch.long, device=x.device) * (self.y_embedder.get_in_channels() - 1)
        with gemm.forward_scope(self, x.shape[0] * self.x_embedder.num_patches):     # (inference under the scaled-fp16 policy: one weight-image launch)
            return self._forward(x, t, y, inference_params)

    def _forward(self, x, t, y, inference_params):
        c = self.t_embedder(t) + self.y_embedder(y, self.training)
        gemm._tls.cond = (c, F.silu(c)) if os.environ.get("DIMSUM_FORWARD_MEMO", "1") != "0" else None      # (the adaLN heads' shared SiLU(c): _modulation)
        try:
            return self._forward_blocks(x, c, inference_params)
        finally:
            gemm._tls.cond = None

    def _forward_blocks(self, x, c, inference_params):
        x = self.x_embedder(x) + self.pos_embed
        residual = None
        for idx, block in enumerate(self.blocks):
            x, residual = block(x, residual, c, inference_params=inference_params)
            if self.use_attn_every_k_layers > 0 and (idx + 1) % self.use_attn_every_k_layers == 0:
                if self.use_independent_attn:
                    x = self.attn_block[(idx + 1) // self.use_attn_every_k_layers - 1](x, c)
                else:
                    x = self.attn_block(x, c)
        if self.norm_f is not None:
            if not self.fused_add_norm:
                residual = x if residual is None else residual + self.drop_path(x)
                x = self.norm_f(residual.to(dtype=self.norm_f.weight.dtype))
            else:
                fn = rms_norm_fn if isinstance(self.norm_f, RMSNorm) else layer_norm_fn
                x = fn(self.drop_path(x), self.norm_f.weight, self.norm_f.bias, eps=self.norm_f.eps, residual=residual,
                       prenorm=False, residual_in_fp32=self.residual_in_fp32)
        return self.unpatchify(self.final_layer(x, c))

    def forward_with_cfg(self, x, t, y=None, inference_params=None, cfg_scale=1.0, **kwargs):
        """classifier-free guidance on a [cond | uncond] batch (models_dim.py:1886-1902)."""
        half = x[: len(x) // 2]
        out = self.forward(torch.cat([half, half], dim=0), t, y, inference_params)
        eps, rest = out[:, : self.in_channels], out[:, self.in_channels:]
        cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
        g = uncond + cfg_scale * (cond - uncond)
        return torch.cat([torch.cat([g, g], dim=0), rest], dim=1)

    def forward_with_adacfg(self, x, t, y=None, inference_params=None, cfg_scale=3.8, scale_pow=4.0, **kwargs):
        """time-dependent (power-cosine) guidance scale (models_dim.py:1904-1924)."""
        if cfg_scale is None:
            return self.forward(x, t, y, inference_params)
        half = x[: len(x) // 2]
        out = self.forward(torch.cat([half, half], dim=0), t, y, inference_params)
        eps, rest = out[:, : self.in_channels], out[:, self.in_channels:]
        cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
        step = (1 - torch.cos(((1 - t) ** scale_pow) * math.pi)) / 2
        s = ((cfg_scale - 1) * step + 1)[: len(x) // 2].view(-1, 1, 1, 1)
        g = uncond + s * (cond - uncond)
        return torch.cat([torch.cat([g, g], dim=0), rest], dim=1)


def _zoo(depth, hidden_size, patch_size):
    def make(**kwargs):
        return DiM(depth=depth, hidden_size=hidden_size, patch_size=patch_size, initializer_cfg=None, ssm_cfg=None, **kwargs)
    return make


DiM_XL_2, DiM_L_2, DiM_L_2_v1 = _zoo(24, 1152, 2), _zoo(16, 1024, 2), _zoo(20, 1024, 2)
DiM_B_2, DiM_L_4, DiM_L_4_v1 = _zoo(12, 768, 2), _zoo(16, 1024, 4), _zoo(20, 1024, 4)
DiM_S_2 = _zoo(12, 384, 2)      # not in the reference zoo: DiT-S analogy used by BASELINE config 1 (SURVEY finding 6)

DiM_models = {"DiM-XL/2": DiM_XL_2, "DiM-L/2": DiM_L_2, "DiM-L/2-v1": DiM_L_2_v1, "DiM-B/2": DiM_B_2,
              "DiM-L/4": DiM_L_4, "DiM-L/4-v1": DiM_L_4_v1, "DiM-S/2": DiM_S_2}
