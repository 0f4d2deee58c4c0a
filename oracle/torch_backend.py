"""CPU stand-in for dimsum_amd.native built on oracle/ -- TEST INFRASTRUCTURE (tests/, smoke(), bench.py cpu_baseline).

`with cpu_oracle_backend():` monkeypatches the tensor-level native entry points (same signatures) with implementations
that run the CPU oracle, and lifts the GPU-only guard of ops/token_ops.py. This lets
the `-m "not gpu"` suite exercise the host logic of dimsum_amd (autograd Functions, layouts, module composition,
state_dict keys) against the reference goldens. The product never enables this itself."""
import contextlib

import numpy as np
import torch

from . import c_ops, np_ops


def _np(t):
    return None if t is None else t.detach().float().cpu().numpy()


def _like(a, ref, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype or ref.dtype)


def selective_scan_fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus, need_out=True, need_x=True, need_ckpt=False):
    y, oz, x = c_ops.selective_scan_fwd(_np(u), _np(delta), _np(A), _np(B), _np(C), _np(D), _np(z), _np(delta_bias), delta_softplus)
    out = torch.empty_like(delta)
    out.copy_(_like(y, u))
    res = [out, torch.from_numpy(x)]
    if z is not None:
        out_z = torch.empty_like(z)
        out_z.copy_(_like(oz, u))
        res.append(out_z)
    if need_ckpt:
        res.append(None)        # the CPU oracle's backward re-derives everything from its inputs
    return res


def selective_scan_fwd_torch_loop(u, delta, A, B, C, D, z, delta_bias, delta_softplus, need_out=True, need_x=True, need_ckpt=False):
    """The reference's own CPU evaluation of the scan, restated: `selective_scan_ref` (mamba/mamba_ssm/ops/selective_scan_interface.py:
    104-171) materialises exp(delta A) and delta B u as two (B, D, L, N) fp32 tensors (:140-148) and then walks the sequence in a Python
    loop of elementwise torch ops (:154-166). Same arithmetic as oracle/ssm_oracle.c; kept only so that bench.py can time the SHAPE of
    the reference's pure-PyTorch path (cpu_baseline.reference_shaped) -- checked against the C oracle in tests/test_oracle_golden.py."""
    dt = delta.float()
    if delta_bias is not None:
        dt = dt + delta_bias.float()[:, None]
    if delta_softplus:
        dt = torch.nn.functional.softplus(dt)
    uf = u.float()
    Bsz, Dm, L = uf.shape
    G = B.shape[1]
    Bx = B.float().repeat_interleave(Dm // G, dim=1).permute(0, 1, 3, 2)           # (B, D, L, N)
    Cx = C.float().repeat_interleave(Dm // G, dim=1).permute(0, 1, 3, 2)
    decay = torch.exp(dt[..., None] * A.float()[None, :, None, :])                 # (B, D, L, N)
    drive = (dt * uf)[..., None] * Bx
    h = torch.zeros(Bsz, Dm, A.shape[1])
    ys = []
    for t in range(L):
        h = decay[:, :, t] * h + drive[:, :, t]
        ys.append((h * Cx[:, :, t]).sum(-1))
    y = torch.stack(ys, dim=2)
    if D is not None:
        y = y + uf * D.float()[:, None]
    out = torch.empty_like(delta)
    out.copy_(y.to(u.dtype))
    res = [out, torch.stack([torch.zeros_like(h), h], -1).reshape(Bsz, Dm, 1, -1)]  # (B, D, 1, 2N): last_state = x[:, :, -1, 1::2]
    if z is not None:
        zz = z.float()
        out_z = torch.empty_like(z)
        out_z.copy_((y * zz * torch.sigmoid(zz)).to(u.dtype))
        res.append(out_z)
    if need_ckpt:
        res.append(None)
    return res


def selective_scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, dz, delta_softplus, recompute_out_z, ckpt=None):
    r = c_ops.selective_scan_bwd(_np(u), _np(delta), _np(A), _np(B), _np(C), _np(D), _np(z), _np(delta_bias), delta_softplus, _np(dout))
    du = torch.empty_like(u).copy_(_like(r["du"], u))
    ddelta = torch.empty_like(delta).copy_(_like(r["ddelta"], u))
    res = [du, ddelta, _like(r["dA"], A), _like(r["dB"], B), _like(r["dC"], C),
           _like(r["dD"], D) if D is not None else None, _like(r["ddelta_bias"], delta_bias) if delta_bias is not None else None]
    if z is not None:
        if dz is None:
            dz = torch.empty_like(z)
        dz.copy_(_like(r["dz"], z))
        res.append(dz)
        if recompute_out_z:
            zz = z.float()
            res.append((out.float() * zz * torch.sigmoid(zz)).to(out.dtype))
    return res


def causal_conv1d_fwd(x, weight, bias, silu_activation, out=None):
    o = c_ops.causal_conv1d_fwd(_np(x), _np(weight), _np(bias), silu_activation)
    if out is None:
        out = torch.empty_like(x)
    out.copy_(_like(o, x))
    return out


def causal_conv1d_fwd_cond(x, weight, bias, silu_activation, init_x):
    return causal_conv1d_fwd(x, weight, bias, silu_activation, out=init_x)


def causal_conv1d_bwd(x, weight, bias, dout, dx, silu_activation):
    rdx, rdw, rdb = c_ops.causal_conv1d_bwd(_np(x), _np(weight), _np(bias), _np(dout), silu_activation)
    if dx is None:
        dx = torch.empty_like(x)
    dx.copy_(_like(rdx, x))
    return [dx, _like(rdw, weight), _like(rdb, bias) if bias is not None else None]


def layer_norm_fwd(x, weight, bias, eps, residual=None, out_dtype=None, residual_dtype=None, is_rms_norm=False,
                   x_bias=None, mod_scale=None, mod_shift=None, rows_per_batch=0):
    xin = _np(x) if x_bias is None else _np(x) + _np(x_bias)
    y, ro, mean, rstd = c_ops.norm_fwd(np.ascontiguousarray(xin, dtype=np.float32), _np(weight), _np(bias), _np(residual), eps, is_rms_norm)
    if mod_scale is not None:
        y = (y.reshape(-1, rows_per_batch, y.shape[-1]) * (1 + _np(mod_scale)[:, None]) + _np(mod_shift)[:, None]).reshape(y.shape).astype(np.float32)
    if residual is not None:
        residual_dtype = residual.dtype
    need = residual is not None or x_bias is not None or (residual_dtype is not None and residual_dtype != x.dtype)
    if residual_dtype is None:
        residual_dtype = x.dtype
    res_out = torch.from_numpy(ro).to(residual_dtype) if need else x
    return (torch.from_numpy(y).to(out_dtype or x.dtype), None if is_rms_norm else torch.from_numpy(mean), torch.from_numpy(rstd), res_out)


def layer_norm_bwd(dy, x, weight, bias, eps, mean, rstd, dresidual=None, has_residual=False, is_rms_norm=False, x_dtype=None):
    dr, dw, db = c_ops.norm_bwd(_np(x), _np(weight), _np(dy), _np(dresidual), eps, is_rms_norm)
    dx = torch.from_numpy(dr).to(x_dtype or x.dtype)
    return dx, _like(dw, weight), _like(db, bias) if bias is not None else None, (torch.from_numpy(dr).to(x.dtype) if has_residual else None)


def gated_gelu_fwd(x12, bias=None):
    x = _np(x12) if bias is None else _np(x12) + _np(bias)
    return torch.from_numpy(np_ops.gated_gelu(np.ascontiguousarray(x, dtype=np.float32)))


def gated_gelu_bwd(x12, bias, dh, need_dbias=True):
    xr = (x12.detach() if bias is None else x12.detach() + bias.detach()).clone().requires_grad_()
    H = xr.shape[-1] // 2
    with torch.enable_grad():
        (torch.nn.functional.gelu(xr[..., :H], approximate="tanh") * xr[..., H:]).backward(dh)
    dbias = xr.grad.reshape(-1, 2 * H).sum(0) if (bias is not None and need_dbias) else None
    return xr.grad, dbias


def token_transform(x, kind, forward, in_index=None, out_index=None, gate=None, scale=None, shift=None, residual=None,
                    w=None, want_y=True, want_wsum=False, want_tsum=False):
    v = _np(x)
    if gate is not None:
        v = v * _np(gate)[:, None]
    if in_index is not None:
        v = v[:, in_index.cpu().numpy()]
    T = {("haar", True): np_ops.haar_dwt_tokens, ("haar", False): np_ops.haar_idwt_tokens, ("dct", True): np_ops.dct_tokens,
         ("dct", False): np_ops.idct_tokens}.get((kind, bool(forward)), lambda a: a)
    t = T(np.ascontiguousarray(v))
    wdot = wsum = None
    tsum = torch.from_numpy(t.astype(np.float64).sum(1).astype(np.float32)) if want_tsum else None
    if w is not None:       # reductions against w indexed like y: w[b, out_index[s], c] pairs with t[b, s, c]
        wn = _np(w).astype(np.float64)
        ws = wn if out_index is None else wn[:, out_index.cpu().numpy()]
        wdot = torch.from_numpy((t.astype(np.float64) * ws).sum(1).astype(np.float32))
        wsum = torch.from_numpy(ws.sum(1).astype(np.float32)) if want_wsum else None
    if scale is not None:
        t = t * (1 + _np(scale)[:, None])
    if shift is not None:
        t = t + _np(shift)[:, None]
    y = np.empty_like(t)
    if out_index is not None:
        y[:, out_index.cpu().numpy()] = t
    else:
        y = t
    if residual is not None:
        y = y + _np(residual)
    y = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)) if want_y else None
    if want_tsum:
        return (y, wdot, wsum, tsum)
    return y if w is None else (y, wdot, wsum)


def xattn_supported(qkv, head_dim):
    return False


@contextlib.contextmanager
def cpu_oracle_backend(scan="c"):
    """scan="torch_loop": the forward scan as the reference's pure-PyTorch loop (selective_scan_fwd_torch_loop) instead of the C oracle"""
    from dimsum_amd import attention_fusion, native, utils
    from dimsum_amd.ops import token_ops
    names = ["selective_scan_fwd", "selective_scan_bwd", "causal_conv1d_fwd", "causal_conv1d_fwd_cond", "causal_conv1d_bwd",
             "layer_norm_fwd", "layer_norm_bwd", "gated_gelu_fwd", "gated_gelu_bwd", "token_transform", "xattn_supported"]
    saved = {n: getattr(native, n, None) for n in names}
    guard, note = token_ops._require_gpu, utils.note_torch_path
    try:
        for n in names:
            setattr(native, n, globals()[n])
        if scan == "torch_loop":
            native.selective_scan_fwd = selective_scan_fwd_torch_loop
        token_ops._require_gpu = lambda x: None
        # with xattn_supported() == False the host code takes its torch-SDPA branch, which the product counts and (for the
        # plain published form) refuses without an opt-in: here that branch IS the checker (SDPA math on the CPU), so the
        # bookkeeping is lifted for the duration -- the product code itself knows nothing about this backend
        utils.note_torch_path = attention_fusion.note_torch_path = lambda *a, **k: None
        yield
    finally:
        for n, f in saved.items():
            if f is None:
                delattr(native, n)
            else:
                setattr(native, n, f)
        token_ops._require_gpu = guard
        utils.note_torch_path = attention_fusion.note_torch_path = note
