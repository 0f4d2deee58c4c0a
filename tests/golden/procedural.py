"""Deterministic, name-keyed weight fill shared by tools/gen_golden.py (applied to the REFERENCE model in the
build container) and by the tests (applied to the dimsum_amd model, here and on the GPU box).

Why: golden outputs for DiM-S/2 / DiM-L/2 need identical weights on both sides, but 50-460 M parameters cannot be
committed. Every floating tensor of a state_dict is regenerated from (crc32(key) ^ seed) with numpy's legacy
RandomState (bit-stable across numpy versions), in ranges that mimic the reference initialisers
(mamba_simple.py:494-526, models_dim.py:1744-1779) but with NO all-zero tensors (SURVEY finding 5: the
reference-initialised model outputs exactly 0 because of adaLN-zero).
"""
import math
import zlib

import numpy as np
import torch

# buffers / frozen analytic tensors that must keep their constructor values
_KEEP_SUFFIXES = (
    "pos_embed", "dwt.w_ll", "dwt.w_lh", "dwt.w_hl", "dwt.w_hh", "idwt.filters",
    "zigzag_paths", "zigzag_paths_reverse", "dct_conv.weight", "idct_conv.0.weight",
)


def _values(key, shape, seed):
    rs = np.random.RandomState((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = key.split(".")[-1]
    if leaf == "A_log" or leaf == "A_b_log":
        d, s = shape
        base = np.log(np.arange(1, s + 1, dtype=np.float64))[None, :].repeat(d, 0)
        return base + 0.1 * rs.standard_normal((d, s))
    if leaf in ("D", "D_b"):
        return 1.0 + 0.1 * rs.standard_normal(shape)
    if key.endswith("dt_proj.bias") or key.endswith("dt_proj_b.bias"):
        dt = np.exp(rs.uniform(size=shape) * (math.log(0.1) - math.log(0.001)) + math.log(0.001))
        dt = np.maximum(dt, 1e-4)
        return dt + np.log(-np.expm1(-dt))
    if key.endswith("dt_proj.weight") or key.endswith("dt_proj_b.weight"):
        std = shape[1] ** -0.5
        return rs.uniform(-std, std, size=shape)
    if "norm" in key.split(".")[-2:][0] and leaf == "weight" and len(shape) == 1:
        return 1.0 + 0.1 * rs.standard_normal(shape)
    if "conv1d" in key and leaf == "weight":
        return rs.uniform(-0.5, 0.5, size=shape)
    if "embedding_table" in key:
        return 0.1 * rs.standard_normal(shape)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return rs.standard_normal(shape) / math.sqrt(fan_in)
    return 0.05 * rs.standard_normal(shape)


@torch.no_grad()
def procedural_fill(model, seed=0):
    """In-place fill of every floating parameter/buffer of `model` (except analytic buffers)."""
    sd = model.state_dict()
    for key in sorted(sd.keys()):
        t = sd[key]
        if not torch.is_floating_point(t) or key.endswith(_KEEP_SUFFIXES):
            continue
        v = _values(key, tuple(t.shape), seed)
        t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(t.shape))
    return model


def seeded(shape, seed, scale=1.0, kind="normal"):
    """Seeded numpy input tensor (float32), used for inputs whose values are not stored in the fixture."""
    rs = np.random.RandomState(seed)
    v = rs.standard_normal(shape) if kind == "normal" else rs.uniform(size=shape)
    return (scale * v).astype(np.float32)


def toy_denoiser(x, t, y=None):
    """Closed-form stand-in for the denoiser in the transport fixtures (tools/gen_golden.py:gen_transport and
    tests/test_transport_golden.py call the same function): smooth, bounded, depends on x, t and the label."""
    tt = t.view(t.shape[0], *([1] * (x.dim() - 1))).to(x)
    out = torch.tanh(0.7 * x + 0.3 * tt) - 0.25 * x * tt
    if y is not None:
        out = out + 0.01 * y.view(tt.shape).to(x)
    return out
