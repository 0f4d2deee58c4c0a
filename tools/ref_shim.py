"""Import-time shims that let the *reference's own pure-PyTorch paths* run in this
container (CPU only, no CUDA extensions, no timm/pywt/torchdiffeq).

THIS FILE NEVER TRAVELS INTO THE PRODUCT PATH: it is used only by
tools/gen_golden.py (to capture golden vectors from /root/reference) and by
the optional reference-vs-oracle cross-check in tests (skipped when
/root/reference is absent, as on the GPU box).

Recipe = SURVEY.md §8(c):
 1. stub the two native modules the reference hard-imports
    (mamba/mamba_ssm/ops/selective_scan_interface.py:3-4)
 2. register a bare `mamba_ssm` package (its __init__ pulls in transformers symbols)
 3. sys.path += causal-conv1d, dimsum
 4. stub timm PatchEmbed/Attention/Mlp/use_fused_attn (timm==0.9.12 semantics)
 5. stub pywt.Wavelet("haar") (PyWavelets==1.6.0 filter constants)
 6. route every CUDA/Triton entry point to the reference's own *_ref functions
"""
import math
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("DIMSUM_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "dimsum"))


def _stub_native():
    for name in ("causal_conv1d_cuda", "selective_scan_cuda"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)


def _stub_mamba_pkg():
    if "mamba_ssm" in sys.modules:
        return
    pkg = types.ModuleType("mamba_ssm")
    pkg.__path__ = [os.path.join(REF, "mamba", "mamba_ssm")]
    sys.modules["mamba_ssm"] = pkg


def _stub_timm():
    if "timm" in sys.modules:
        return

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
            super().__init__()
            self.img_size = (img_size, img_size)
            self.patch_size = (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
            self.norm = nn.Identity()

        def forward(self, x):
            return self.norm(self.proj(x).flatten(2).transpose(1, 2))

    class Attention(nn.Module):
        def __init__(self, dim, num_heads=8, qkv_bias=False, **kw):
            super().__init__()
            self.num_heads = num_heads
            self.head_dim = dim // num_heads
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.proj = nn.Linear(dim, dim)

        def forward(self, x):
            B, N, C = x.shape
            qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
            q, k, v = qkv.unbind(0)
            x = F.scaled_dot_product_attention(q, k, v)
            return self.proj(x.transpose(1, 2).reshape(B, N, C))

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features or in_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    layers = types.ModuleType("timm.layers")
    vt.PatchEmbed, vt.Attention, vt.Mlp = PatchEmbed, Attention, Mlp
    layers.use_fused_attn = lambda *a, **k: True
    timm.models, timm.layers, models.vision_transformer = models, layers, vt
    sys.modules.update({"timm": timm, "timm.models": models,
                        "timm.models.vision_transformer": vt, "timm.layers": layers})


def _stub_pywt():
    if "pywt" in sys.modules:
        return
    s = 1.0 / math.sqrt(2.0)

    class Wavelet:
        def __init__(self, name):
            assert name == "haar"
            self.dec_lo, self.dec_hi = [s, s], [-s, s]
            self.rec_lo, self.rec_hi = [s, s], [s, -s]

    m = types.ModuleType("pywt")
    m.Wavelet = Wavelet
    sys.modules["pywt"] = m


_LOADED = {}


def load():
    """Returns a namespace with the reference modules routed to their own CPU refs."""
    if "ns" in _LOADED:
        return _LOADED["ns"]
    assert available(), "reference tree not present"
    _stub_native(); _stub_mamba_pkg(); _stub_timm(); _stub_pywt()
    for p in (os.path.join(REF, "causal-conv1d"), os.path.join(REF, "dimsum")):
        if p not in sys.path:
            sys.path.insert(0, p)

    import causal_conv1d.causal_conv1d_interface as cci
    import mamba_ssm.ops.selective_scan_interface as ssi
    import mamba_ssm.ops.triton.layernorm as ln
    import mamba_ssm.modules.mamba_simple as ms

    def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
        return ln.rms_norm_ref(x, weight, bias, residual=residual, eps=eps, prenorm=prenorm, upcast=True)

    def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                      is_rms_norm=False):
        fn = ln.rms_norm_ref if is_rms_norm else ln.layer_norm_ref
        return fn(x, weight, bias, residual=residual, eps=eps, prenorm=prenorm, upcast=True)

    ln.rms_norm_fn, ln.layer_norm_fn = rms_norm_fn, layer_norm_fn
    ln.RMSNorm.forward = lambda self, x, residual=None, prenorm=False, residual_in_fp32=False: rms_norm_fn(
        x, self.weight, self.bias, residual=residual, eps=self.eps, prenorm=prenorm)
    ssi.causal_conv1d_fn = cci.causal_conv1d_ref
    ssi.selective_scan_fn = ssi.selective_scan_ref
    ms.selective_scan_fn = ssi.selective_scan_ref
    ms.causal_conv1d_fn = cci.causal_conv1d_ref
    ms.rms_norm_fn, ms.layer_norm_fn = rms_norm_fn, layer_norm_fn

    import scanning_orders
    import wavelet_layer
    import dct_layer
    import attention_fusion
    import mlp
    import models_dim

    models_dim.rms_norm_fn, models_dim.layer_norm_fn = rms_norm_fn, layer_norm_fn

    ns = types.SimpleNamespace(cci=cci, ssi=ssi, ln=ln, ms=ms, scanning_orders=scanning_orders,
                               wavelet_layer=wavelet_layer, dct_layer=dct_layer,
                               attention_fusion=attention_fusion, mlp=mlp, models_dim=models_dim)
    _LOADED["ns"] = ns
    return ns


def load_transport():
    """The reference's flow-matching harness (dimsum/transport/: Transport.training_losses :127-164, Sampler.sample_sde
    :286-341, Sampler.sample_ode :343-386, path.py, integrators.py) imported as the package `transport`. Its one missing
    import is `torchdiffeq.odeint` (requirements.txt:172, absent from this image): the stand-in below does explicit Euler
    on the given grid -- `x += (t[i+1] - t[i]) f(t[i], x)`, torchdiffeq's published fixed-grid "euler" -- and refuses
    every other method, so fixtures made through it pin the reference's OWN code around the solver (the drift of each
    model type, `t` as ones(B) * t, check_interval's end points, reverse time), not torchdiffeq."""
    if "transport" in _LOADED:
        return _LOADED["transport"]
    assert available(), "reference tree not present"
    if "torchdiffeq" not in sys.modules:
        m = types.ModuleType("torchdiffeq")

        def odeint(fn, x, t, method=None, atol=None, rtol=None):
            assert method == "euler", "stand-in for torchdiffeq: fixed-grid euler only"
            if isinstance(x, tuple):              # tuple states (the likelihood ODE): component-wise, one stacked tensor each
                out = [x]
                for a, b in zip(t[:-1], t[1:]):
                    x = tuple(c + (b - a) * d for c, d in zip(x, fn(a, x)))
                    out.append(x)
                return tuple(torch.stack(c) for c in zip(*out))
            out = [x]
            for a, b in zip(t[:-1], t[1:]):
                x = x + (b - a) * fn(a, x)
                out.append(x)
            return torch.stack(out)

        m.odeint = odeint
        sys.modules["torchdiffeq"] = m
    p = os.path.join(REF, "dimsum")
    if p not in sys.path:
        sys.path.insert(0, p)
    import transport
    assert os.path.realpath(transport.__file__).startswith(os.path.realpath(REF)), transport.__file__
    _LOADED["transport"] = transport
    return transport


def slow_path(model):
    """Force every mixer onto the pure-PyTorch path (mamba_simple.py:658-700). The slow path drops the zigzag gather
    that the fast path applies around mamba_inner_fn (mamba_simple.py:627-657, SURVEY finding 3), so mixers with a
    zigzag scan_type get their forward wrapped: gather tokens by perm -> slow path -> inverse gather. Gathering the
    columns of xz equals gathering the tokens before the per-token in_proj."""
    import types as _t
    for m in model.modules():
        if hasattr(m, "use_fast_path"):
            m.use_fast_path = False
            st = getattr(m, "scan_type", "none")
            if st.startswith(("zigma", "sweep", "jpeg")) and not getattr(m, "_zigzag_wrapped", False):
                inner = m.forward

                def fwd(self, hidden_states, *a, _inner=inner, **k):
                    perm = self.zigzag_paths[self.layer_idx]
                    rev = self.zigzag_paths_reverse[self.layer_idx]
                    return _inner(hidden_states[:, perm], *a, **k)[:, rev]

                m.forward = _t.MethodType(fwd, m)
                m._zigzag_wrapped = True
    return model


def allow_zigzag_through_dim(ns):
    """SURVEY finding 2: with cond_mamba=True the reference's create_block passes `scan_type` twice to
    partial(CondMamba, ...) (models_dim.py:2035-2043: once by name, once inside **block_kwargs built at :1657) and raises
    TypeError, so `DiM(scan_type="zigma_8")` cannot be constructed in this snapshot. Both values are the same string; this
    wrapper drops the duplicate before delegating to the reference's own create_block, which is the intended semantics
    (CondMamba.forward :627-657 with zigzag_paths / zigzag_paths_reverse of :1640-1658)."""
    md = ns.models_dim
    if getattr(md.create_block, "_dedup", False):
        return
    orig = md.create_block

    def create_block(*a, scan_type="none", block_kwargs={}, **k):
        bk = dict(block_kwargs)
        st = bk.pop("scan_type", scan_type)
        return orig(*a, scan_type=st, block_kwargs=bk, **k)

    create_block._dedup = True
    md.create_block = create_block


def route_no_out_proj_to_refs(ns):
    """scan_type="v2" (mamba_simple.py:593-625) only exists on the fast path, whose mamba_inner_fn_no_out_proj_cond is
    CUDA-only (selective_scan_interface.py:375-576). Route it to the reference's own mamba_inner_ref (:1455-1500, i.e.
    causal_conv1d_ref + selective_scan_ref) with an identity out-projection -- exact in fp32 (products by 1, sums with 0) --
    so that the reference's v2 forward code runs unchanged on CPU."""
    ssi, ms = ns.ssi, ns.ms

    def no_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, B=None, C=None, D=None, delta_bias=None,
                    B_proj_bias=None, C_proj_bias=None, delta_softplus=True, init_states=None):
        eye = torch.eye(A.shape[0], dtype=xz.dtype)
        y = ssi.mamba_inner_ref(xz, conv_w, conv_b, x_proj_w, dt_proj_w, eye, None, A, B, C, D, delta_bias=delta_bias,
                                B_proj_bias=B_proj_bias, C_proj_bias=C_proj_bias, delta_softplus=delta_softplus)
        return y.transpose(1, 2)        # "b l d -> b d l"

    ms.mamba_inner_fn_no_out_proj_cond = no_out_proj
    ms.mamba_inner_fn_no_out_proj = no_out_proj


def rerandomize_zeros(model, std=0.02, seed=1234):
    """SURVEY finding 5: adaLN-zero / zero final layer make the reference-initialised model output exactly 0.
    Every all-zero parameter <- N(0, std^2) so goldens are not vacuous."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
            if p.numel() > 0 and torch.count_nonzero(p) == 0:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return model
