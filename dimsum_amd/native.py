"""Tensor-level entry points with the SAME signatures as the reference's native modules
(`selective_scan_cuda`, `causal_conv1d_cuda`, and the Triton `_layer_norm_fwd/_bwd`), implemented on top of the C ABI
of libdimsum_hip.so.  This is the seam the reference's Python would bind: INTEGRATION.md shows the two-line shim.

Conventions kept from the reference host wrappers:
  * argument checks raise RuntimeError where the reference has TORCH_CHECK (selective_scan.cpp:235-305,
    causal_conv1d.cpp:226-262),
  * outputs are allocated here with torch's caching allocator (`out = empty_like(delta)`, selective_scan.cpp:311),
  * launches go to torch's CURRENT stream of the input's device, no synchronisation.
There is no CPU path: a non-GPU tensor or a missing library is an error.
"""
import contextlib

import os

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


class _ZeroArena:
    """Zero-filled fp32 scratch for the small accumulators the backward kernels ADD into (conv / norm / scan weight-gradient sums, the token
    passes' (3, B, C) reductions, bias-gradient sums): one fill launch per 16-MB chunk instead of one per accumulator -- ~300 fills of a few KB
    .. 800 KB per DiM-L/2 training step. A slice is handed out once and never again; a chunk lives as long as any of its slices (a parameter
    gradient that is such a slice keeps its chunk until the next zero_grad). Per (device, stream); bypassed under stream capture."""
    CHUNK = 1 << 22          # floats

    def __init__(self):
        import threading
        self._lock, self._cur = threading.Lock(), {}

    def take(self, n, device):
        n = int(n)
        if n <= 0 or n > self.CHUNK // 4 or torch.cuda.is_current_stream_capturing() or os.environ.get("DIMSUM_ZERO_ARENA", "1") == "0":
            return torch.zeros(max(n, 0), device=device, dtype=torch.float32)
        pad = (n + 63) // 64 * 64                          # 256-byte slices: vectorised consumers (fused optimizers) stay on their fast path
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        with self._lock:
            buf, used = self._cur.get(key, (None, 0))
            if buf is None or used + pad > self.CHUNK:
                buf, used = torch.zeros(self.CHUNK, device=device, dtype=torch.float32), 0
            self._cur[key] = (buf, used + pad)
        return buf[used:used + n]


_zero_arena = _ZeroArena()


def _zeros(n, device):
    """n zero floats (a fresh slice of the zero arena)"""
    return _zero_arena.take(n, device)


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _gpu(*ts):
    for t in ts:
        if t is not None:
            _check(t.is_cuda, "dimsum_amd.native: expected a GPU tensor (there is no CPU fallback; the CPU oracle "
                              "lives under oracle/ and is test infrastructure only)")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------------------------------
# selective_scan_cuda.fwd / .bwd   (mamba/csrc/selective_scan/selective_scan.cpp:226-492)
# ---------------------------------------------------------------------------------------------------------------------
# the producers' image modes (dimsum_*_params_t.y_split3 / out_split3): True = three pieces [hi | hi | lo], "f16s" = scaled fp16, "pair" = [hi | lo]
_SPLIT_MODE = {"f16s": 2, "pair": 3}
_scan_fwd_variant = 0     # dimsum_ssm_params_t.kernel_variant of the calls made from here: 0 = the library's own choice


@contextlib.contextmanager
def scan_fwd_variant(v):
    """tests / tuning: ask for one forward scan kernel (lanes per channel: 1, 2, 4, 16; 0 = automatic) for the calls made inside.
    A per-call field of the C ABI -- the library keeps no state; this host-side setting is not thread-safe."""
    global _scan_fwd_variant
    old, _scan_fwd_variant = _scan_fwd_variant, int(v)
    try:
        yield
    finally:
        _scan_fwd_variant = old


def scan_fwd_kernel_for(batch, dim, seqlen, dstate, n_groups=1):
    """which forward scan kernel the library runs for this launch shape, as lanes per channel (dimsum_ssm_scan_fwd_variant): 1 = the
    64-channel kernel (a full chip), 2 / 4 / 16 = the state-split kernels of underfilled launches"""
    P = _lib.SsmParams()
    P.batch, P.dim, P.seqlen, P.dstate, P.n_groups, P.n_chunks = batch, dim, seqlen, dstate, n_groups, (seqlen + 2047) // 2048
    if _scan_fwd_variant:
        _lib.attach_ext(P, _lib.SsmExt).kernel_variant = _scan_fwd_variant
    return int(_lib.load().dimsum_ssm_scan_fwd_variant(P))


def _fill_ssm(P, u, delta, A, B, C, D, z, delta_bias, delta_softplus, out, x, out_z, ckpt=None):
    """fills the reference-shaped struct P (dimsum_ssm_params_t) and links a zeroed dimsum_ssm_ext_t to it -> the extension (the caller sets
    what it uses beyond the reference interface: saved states, the inference fusions, timing events)"""
    E = _lib.attach_ext(P, _lib.SsmExt)
    E.kernel_variant, E.ckpt_ptr = _scan_fwd_variant, _ptr(ckpt)
    batch, dim, seqlen = u.shape
    P.batch, P.dim, P.seqlen, P.dstate = batch, dim, seqlen, A.shape[1]
    P.n_groups, P.n_chunks = B.shape[1], (seqlen + 2047) // 2048
    P.delta_softplus, P.dtype = int(bool(delta_softplus)), _DT[u.dtype]
    P.A_d_stride, P.A_dstate_stride = A.stride(0), A.stride(1)
    P.B_batch_stride, P.B_group_stride, P.B_dstate_stride = B.stride(0), B.stride(1), B.stride(2)
    P.C_batch_stride, P.C_group_stride, P.C_dstate_stride = C.stride(0), C.stride(1), C.stride(2)
    P.u_batch_stride, P.u_d_stride = u.stride(0), u.stride(1)
    P.delta_batch_stride, P.delta_d_stride = delta.stride(0), delta.stride(1)
    if z is not None:
        P.z_batch_stride, P.z_d_stride = z.stride(0), z.stride(1)
    if out is not None:
        P.out_batch_stride, P.out_d_stride = out.stride(0), out.stride(1)
    if out_z is not None:
        P.out_z_batch_stride, P.out_z_d_stride = out_z.stride(0), out_z.stride(1)
    P.A_ptr, P.B_ptr, P.C_ptr, P.D_ptr = _ptr(A), _ptr(B), _ptr(C), _ptr(D)
    P.u_ptr, P.delta_ptr, P.delta_bias_ptr, P.z_ptr = _ptr(u), _ptr(delta), _ptr(delta_bias), _ptr(z)
    P.out_ptr, P.x_ptr, P.out_z_ptr = _ptr(out), _ptr(x), _ptr(out_z)
    return E


def _check_ssm(u, delta, A, B, C, D, z, delta_bias):
    _gpu(u, delta, A, B, C, D, z, delta_bias)
    _check(u.dtype in _DT, "selective_scan: input must be float32, float16 or bfloat16")
    _check(A.dtype == torch.float32, "selective_scan: complex A is out of scope of this build (DiMSUM uses real A, "
                                     "mamba_simple.py:586); A must be float32")
    _check(B.dim() == 4 and C.dim() == 4, "selective_scan: only input-dependent B and C of shape (batch, groups, "
                                          "dstate, seqlen) are supported (constant B/C is unused by DiMSUM)")
    _check(delta.dtype == u.dtype and B.dtype == u.dtype and C.dtype == u.dtype, "selective_scan: dtype mismatch")
    batch, dim, seqlen = u.shape
    dstate, groups = A.shape[1], B.shape[1]
    _check(dstate <= 256, "selective_scan only supports state dimension <= 256")
    _check(tuple(delta.shape) == (batch, dim, seqlen), "selective_scan: delta must have shape (batch, dim, seqlen)")
    _check(tuple(A.shape) == (dim, dstate), "selective_scan: A must have shape (dim, dstate)")
    _check(tuple(B.shape) == (batch, groups, dstate, seqlen) and tuple(C.shape) == (batch, groups, dstate, seqlen),
           "selective_scan: B and C must have shape (batch, groups, dstate, seqlen)")
    for t, name in ((u, "u"), (delta, "delta"), (B, "B"), (C, "C"), (z, "z")):
        if t is not None:
            _check(t.stride(-1) == 1 or t.shape[-1] == 1, f"selective_scan: {name}.stride(-1) must be 1")
    if D is not None:
        _check(D.dtype == torch.float32 and tuple(D.shape) == (dim,) and D.stride(-1) == 1, "selective_scan: bad D")
    if delta_bias is not None:
        _check(delta_bias.dtype == torch.float32 and tuple(delta_bias.shape) == (dim,) and delta_bias.stride(-1) == 1,
               "selective_scan: bad delta_bias")
    if z is not None:
        _check(z.dtype == u.dtype and tuple(z.shape) == (batch, dim, seqlen), "selective_scan: bad z")


def scan_ckpt_shape(batch, dim, seqlen, dstate):
    """states kept for the backward: h before every 8th step, (batch, ceil(L/8), dstate, dim) fp32"""
    return (batch, (seqlen + 7) // 8, dstate, dim)


def scan_dt_proj_supported(u, z, A, dt_w, dt_xt, n_groups=1):
    """whether selective_scan_fwd(dt_proj=(dt_w, dt_xt)) is served (csrc/ssm_scan_fwd_kernel.hpp, kDt): the 64-channels-per-wave kernel on
    its full fp32 vector path -- float32, dstate 16, dim / n_groups % 64 == 0, seqlen % 4 == 0, 16-byte aligned rows, dt_rank % 4 == 0 and
    <= 32, dt_xt (dt_rank, batch seqlen) r-major -- and only where that kernel is the one a launch of this shape gets"""
    batch, dim, seqlen = u.shape
    R = dt_w.shape[1]
    return (u.is_cuda and u.dtype == torch.float32 and z is not None and A.shape[1] == 16 and (dim // n_groups) % 64 == 0 and seqlen % 4 == 0
            and dt_w.dtype == torch.float32 and dt_xt.dtype == torch.float32 and dt_w.dim() == 2 and dt_w.shape[0] == dim and R % 4 == 0 and 0 < R <= 32
            and dt_w.stride(1) == 1 and dt_w.stride(0) % 4 == 0 and dt_w.data_ptr() % 16 == 0
            and dt_xt.dim() == 2 and dt_xt.shape == (R, batch * seqlen) and dt_xt.stride(1) == 1 and 32 * dt_xt.stride(0) < 2 ** 31
            and u.stride(2) == 1 and z.stride(2) == 1 and all(st % 4 == 0 for st in (*u.stride()[:2], *z.stride()[:2])) and u.data_ptr() % 16 == 0
            and z.data_ptr() % 16 == 0 and scan_fwd_kernel_for(batch, dim, seqlen, A.shape[1], n_groups) == 1)


def scan_out_z_f16_supported(u, z, A, n_groups):
    """whether selective_scan_fwd(out_z_f16=True) is served: the 64-channel kernel's full fp32 inference path on whole 32-step tiles"""
    return (z is not None and u.dtype == torch.float32 and z.dtype == torch.float32 and A.shape[1] == 16 and not A.is_complex() and n_groups == 1
            and u.shape[1] % 64 == 0 and u.shape[2] % 32 == 0 and u.numel() > 0)


def selective_scan_fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus, need_out=True, need_x=True, need_ckpt=False, out_z_planes=False, dt_proj=None, out_z_f16=False):
    """-> [out, x, (out_z)]   exactly like selective_scan_cuda.fwd.
    `need_out=False` / `need_x=False` are inference extras: the corresponding store is skipped and None returned.
    `need_ckpt=True` (training extra) appends the tile-boundary states the backward kernel consumes.
    `out_z_planes=True` (inference extra, float32): out_z comes back as its split-bf16 pair of d-major planes, a (2 dim, batch seqlen)
    bfloat16 matrix [hi; lo] -- the operand image of out_proj's GEMM (gemm_tn(..., alias_rows=dim)), the same 4 bytes per element.
    `out_z_f16=True` (inference extra, see scan_out_z_f16_supported): out_z comes back as (image, inv): a (dim, batch seqlen) float16 matrix of
    block-scaled values and the (batch seqlen / 32, dim / 64) float32 table of the blocks' inverse scales (include/dimsum_hip.h, out_z_f16)."""
    if dt_proj is not None:
        # fused dt_proj (inference extra): delta = dt_w @ dt_xt is formed inside the scan (scan_dt_proj_supported); `delta` only lends its layout
        dt_w, dt_xt = dt_proj
        _gpu(dt_w, dt_xt)
        _check(delta is None and not need_ckpt and not need_out and scan_dt_proj_supported(u, z, A, dt_w, dt_xt, B.shape[1]),
               "selective_scan_fwd: dt_proj = (dt_w, dt_xt) needs delta = None, no saved states, no `out` and a shape scan_dt_proj_supported takes")
        delta = u                  # (argument checks and the out_z layout below; the kernel never reads it)
    _check_ssm(u, delta, A, B, C, D, z, delta_bias)
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    n_chunks = (seqlen + 2047) // 2048
    out = torch.empty_like(delta) if need_out else None          # HBL layout like delta (selective_scan.cpp:310-311)
    x = torch.empty((batch, dim, n_chunks, dstate * 2), device=u.device, dtype=torch.float32) if need_x else None
    planes = z16 = None
    if out_z_f16:
        _check(not out_z_planes and not need_ckpt and not need_out and scan_out_z_f16_supported(u, z, A, B.shape[1]),
               "selective_scan_fwd: out_z_f16 needs no saved states, no `out`, no planes and a shape scan_out_z_f16_supported takes")
        z16 = (torch.empty((dim, batch * seqlen), device=u.device, dtype=torch.float16), torch.empty((batch * seqlen // 32, dim // 64), device=u.device, dtype=torch.float32))
        out_z = z16[0].view(dim, batch, seqlen).permute(1, 0, 2)
    elif out_z_planes:
        _check(z is not None and u.dtype == torch.float32 and seqlen % 8 == 0, "selective_scan_fwd: out_z_planes needs z, float32 I/O and seqlen % 8 == 0")
        planes = torch.empty((2 * dim, batch * seqlen), device=u.device, dtype=torch.bfloat16)
        out_z = planes[:dim].view(dim, batch, seqlen).permute(1, 0, 2)           # the hi plane as a (batch, dim, seqlen) view
    else:
        out_z = torch.empty_like(z) if z is not None else None
    ckpt = torch.empty(scan_ckpt_shape(batch, dim, seqlen, dstate), device=u.device, dtype=torch.float32) if need_ckpt else None
    if u.numel() > 0:
        P = _lib.SsmParams()
        E = _fill_ssm(P, u, delta, A, B, C, D, z, delta_bias, delta_softplus, out, x, out_z, ckpt)
        if planes is not None:
            E.out_z_lo_offset = dim * batch * seqlen
        if z16 is not None:
            E.out_z_f16, E.out_z_scale_ptr, E.out_z_scale_ld = 1, _ptr(z16[1]), z16[1].stride(0)
        if dt_proj is not None:
            P.delta_ptr = None
            E.dt_w_ptr, E.dt_x_ptr, E.dt_w_row_stride, E.dt_x_row_stride, E.dt_rank = _ptr(dt_w), _ptr(dt_xt), dt_w.stride(0), dt_xt.stride(0), dt_w.shape[1]
        with torch.cuda.device(u.device):
            _lib.check(_lib.load().dimsum_ssm_scan_fwd(P, _stream(u)), "selective_scan_fwd")
    res = [out, x]
    if z is not None:
        res.append(z16 if z16 is not None else out_z if planes is None else planes)
    if need_ckpt:
        res.append(ckpt)
    return res


# ---------------------------------------------------------------------------------------------------------------------
# causal_conv1d_cuda.*   (causal-conv1d/csrc/causal_conv1d.cpp:221-577)
# ---------------------------------------------------------------------------------------------------------------------
def _check_conv(x, weight, bias):
    _gpu(x, weight, bias)
    _check(x.dim() == 3, "causal_conv1d: x must be (batch, dim, seqlen)")
    _check(x.dtype in _DT, "causal_conv1d: input must be float32, float16 or bfloat16")
    _check(x.stride(2) == 1 or x.shape[2] == 1, "causal_conv1d: only seqlen-contiguous (batch, dim, seqlen) inputs are "
                                                "supported here (DiMSUM never feeds channel-last)")
    dim, width = weight.shape
    _check(dim == x.shape[1], "causal_conv1d: weight must be (dim, width)")
    _check(2 <= width <= 4, "causal_conv1d only supports width between 2 and 4")
    if bias is not None:
        _check(tuple(bias.shape) == (dim,) and bias.stride(-1) == 1, "causal_conv1d: bias must be (dim,)")


def _fill_conv(P, x, weight, bias, silu, out):
    P.batch, P.dim, P.seqlen = x.shape
    P.width, P.silu_activation, P.dtype = weight.shape[1], int(bool(silu)), _DT[x.dtype]
    P.x_batch_stride, P.x_c_stride = x.stride(0), x.stride(1)
    P.weight_c_stride, P.weight_width_stride = weight.stride(0), weight.stride(1)
    P.x_ptr, P.weight_ptr, P.bias_ptr = _ptr(x), _ptr(weight), _ptr(bias)
    if out is not None:
        P.out_batch_stride, P.out_c_stride, P.out_ptr = out.stride(0), out.stride(1), _ptr(out)


def causal_conv1d_fwd(x, weight, bias, silu_activation, out=None):
    """-> out (batch, dim, seqlen), like causal_conv1d_cuda.causal_conv1d_fwd. Weights are used in fp32."""
    _check_conv(x, weight, bias)
    weight = weight.float()
    bias = bias.float().contiguous() if bias is not None else None
    if out is None:
        out = torch.empty_like(x)       # inherits x's layout like the reference (causal_conv1d.cpp:262): d-major in Mamba
    if x.numel() > 0:
        P = _lib.ConvParams()
        _fill_conv(P, x, weight, bias, silu_activation, out)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dimsum_causal_conv1d_fwd(P, _stream(x)), "causal_conv1d_fwd")
    return out


def causal_conv1d_fwd_cond(x, weight, bias, silu_activation, init_x):
    """causal_conv1d_cuda.causal_conv1d_fwd_cond: the result is written INTO init_x and returned; the values of init_x
    never enter the computation (causal_conv1d.cpp:326-329, SURVEY finding 1)."""
    _check(init_x.shape == x.shape and init_x.dtype == x.dtype and init_x.stride(-1) == 1, "causal_conv1d_fwd_cond: bad init_x")
    return causal_conv1d_fwd(x, weight, bias, silu_activation, out=init_x)


def causal_conv1d_bwd(x, weight, bias, dout, dx, silu_activation):
    """-> [dx, dweight, dbias] like causal_conv1d_cuda.causal_conv1d_bwd (dx may be a caller-provided view)."""
    _check_conv(x, weight, bias)
    _gpu(dout, dx)
    _check(dout.shape == x.shape and dout.dtype == x.dtype and (dout.stride(2) == 1 or dout.shape[2] == 1), "causal_conv1d_bwd: bad dout")
    if dx is None:
        dx = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    else:
        _check(dx.shape == x.shape and dx.dtype == x.dtype and dx.stride(2) == 1, "causal_conv1d_bwd: bad dx")
    w32 = weight.float()
    b32 = bias.float().contiguous() if bias is not None else None
    # fp32 accumulate, then cast (cpp:405-425); ONE zero-filled buffer for both accumulators (one fill launch instead of two)
    acc = _zeros(weight.numel() + (weight.shape[0] if bias is not None else 0), x.device)
    dweight = acc[:weight.numel()].view(weight.shape)
    dbias = acc[weight.numel():] if bias is not None else None
    if x.numel() > 0:
        Q = _lib.ConvBwdParams()
        _fill_conv(Q.fwd, x, w32, b32, silu_activation, None)
        Q.dout_batch_stride, Q.dout_c_stride = dout.stride(0), dout.stride(1)
        Q.dx_batch_stride, Q.dx_c_stride = dx.stride(0), dx.stride(1)
        Q.dweight_c_stride, Q.dweight_width_stride = dweight.stride(0), dweight.stride(1)
        Q.dout_ptr, Q.dx_ptr, Q.dweight_ptr, Q.dbias_ptr = _ptr(dout), _ptr(dx), _ptr(dweight), _ptr(dbias)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dimsum_causal_conv1d_bwd(Q, _stream(x)), "causal_conv1d_bwd")
    return [dx, dweight.to(weight.dtype), dbias.to(bias.dtype) if bias is not None else None]


# ---------------------------------------------------------------------------------------------------------------------
# fused add + RMSNorm / LayerNorm   (Triton _layer_norm_fwd / _layer_norm_bwd, ops/triton/layernorm.py:120-364)
# ---------------------------------------------------------------------------------------------------------------------
def layer_norm_fwd(x, weight, bias, eps, residual=None, out_dtype=None, residual_dtype=None, is_rms_norm=False,
                   x_bias=None, mod_scale=None, mod_shift=None, rows_per_batch=0, split3=False):
    """x: (M, N) -> (y, mean, rstd, residual_out), like _layer_norm_fwd (layernorm.py:120-187).
    residual_out is x itself when no residual is added and no dtype change is requested.
    Extras: `x_bias` (N) is added to x first (the bias of the Linear that produced x); `mod_scale/mod_shift`
    ((M / rows_per_batch, N), sharing a row stride) apply y * (1 + scale) + shift per batch element after the norm;
    `split3`: y is returned as the split-bf16 left operand image (M, 3N) bfloat16 of the Linear that consumes it (split3_rows);
    `split3="f16s"`: as a scaled-fp16 image (F16Image, rows_f16s)."""
    _gpu(x, weight, bias, residual)
    _check(x.dim() == 2 and x.stride(-1) == 1, "layer_norm_fwd: x must be (M, N) with contiguous rows")
    M, N = x.shape
    _check(tuple(weight.shape) == (N,), "layer_norm_fwd: weight must be (N,)")
    if residual is not None:
        _check(residual.shape == x.shape and residual.stride(-1) == 1, "layer_norm_fwd: bad residual")
        residual_dtype = residual.dtype
    y_inv = None
    if split3 == "f16s":
        _check(N % 4 == 0, "layer_norm_fwd: the f16s image needs N % 4 == 0")
        y = torch.empty((M, N), device=x.device, dtype=torch.float16)
        y_inv = torch.empty((M,), device=x.device, dtype=torch.float32)
    elif split3:
        _check(N % 4 == 0 and out_dtype in (None, torch.bfloat16), "layer_norm_fwd: split3 needs N % 4 == 0 (bfloat16 output)")
        y = torch.empty((M, (2 if split3 == "pair" else 3) * N), device=x.device, dtype=torch.bfloat16)
    else:
        y = torch.empty((M, N), device=x.device, dtype=x.dtype if out_dtype is None else out_dtype)
    need_res_out = residual is not None or x_bias is not None or (residual_dtype is not None and residual_dtype != x.dtype)
    if residual_dtype is None:
        residual_dtype = x.dtype
    residual_out = torch.empty((M, N), device=x.device, dtype=residual_dtype) if need_res_out else None
    mean = torch.empty((M,), device=x.device, dtype=torch.float32) if not is_rms_norm else None
    rstd = torch.empty((M,), device=x.device, dtype=torch.float32)
    if M > 0:
        P = _lib.NormParams()
        P.rows, P.cols, P.is_rms_norm, P.eps = M, N, int(is_rms_norm), float(eps)
        P.x_dtype, P.out_dtype, P.y_split3 = _DT[x.dtype], _DT[y.dtype], _SPLIT_MODE.get(split3, int(bool(split3)))
        P.y_inv_scale_ptr = _ptr(y_inv)
        P.residual_dtype = _DT[residual_out.dtype] if residual_out is not None else _DT[x.dtype]
        P.x_row_stride, P.y_row_stride = x.stride(0), y.stride(0)
        if residual is not None:
            P.residual_row_stride = residual.stride(0)
        if residual_out is not None:
            P.residual_out_row_stride = residual_out.stride(0)
        w32 = weight.float().contiguous()
        b32 = bias.float().contiguous() if bias is not None else None
        P.x_ptr, P.residual_ptr, P.weight_ptr, P.bias_ptr = _ptr(x), _ptr(residual), _ptr(w32), _ptr(b32)
        P.y_ptr, P.residual_out_ptr, P.mean_ptr, P.rstd_ptr = _ptr(y), _ptr(residual_out), _ptr(mean), _ptr(rstd)
        if x_bias is not None:
            xb32 = x_bias.float().contiguous()
            _check(tuple(xb32.shape) == (N,), "layer_norm_fwd: x_bias must be (N,)")
            P.xbias_ptr = _ptr(xb32)
        if mod_scale is not None:
            _gpu(mod_scale, mod_shift)
            _check(mod_shift is not None and rows_per_batch > 0 and M % rows_per_batch == 0, "layer_norm_fwd: bad modulation arguments")
            nb = M // rows_per_batch
            _check(tuple(mod_scale.shape) == (nb, N) and tuple(mod_shift.shape) == (nb, N) and mod_scale.dtype == torch.float32
                   and mod_shift.dtype == torch.float32 and mod_scale.stride(1) == 1 and mod_shift.stride(1) == 1
                   and mod_scale.stride(0) == mod_shift.stride(0), "layer_norm_fwd: mod_scale / mod_shift must be (M / rows_per_batch, N) float32 sharing a row stride")
            P.mod_scale_ptr, P.mod_shift_ptr, P.mod_row_stride, P.rows_per_batch = _ptr(mod_scale), _ptr(mod_shift), mod_scale.stride(0), rows_per_batch
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dimsum_norm_fwd(P, _stream(x)), "layer_norm_fwd")
    if y_inv is not None:
        y = F16Image(y, y_inv)
    elif split3 == "pair":
        y = PairImage(y)
    return y, mean, rstd, residual_out if residual_out is not None else x


def layer_norm_bwd(dy, x, weight, bias, eps, mean, rstd, dresidual=None, has_residual=False, is_rms_norm=False, x_dtype=None):
    """-> (dx, dw, db, dresidual_in), like _layer_norm_bwd (layernorm.py:288-364). `x` is the saved residual_out."""
    _gpu(dy, x, weight, rstd, dresidual)
    M, N = x.shape
    xf = x if x.dtype == torch.float32 else x.float()
    dyf = dy if dy.dtype == torch.float32 else dy.float()
    drf = None if dresidual is None else (dresidual if dresidual.dtype == torch.float32 else dresidual.float())
    dx32 = torch.empty((M, N), device=x.device, dtype=torch.float32)
    acc = _zeros(2 * N if bias is not None else N, x.device)        # (both accumulators out of the zero arena)
    dw = acc[:N]
    db = acc[N:] if bias is not None else None
    if M > 0:
        P = _lib.NormBwdParams()
        P.rows, P.cols, P.is_rms_norm, P.eps = M, N, int(is_rms_norm), float(eps)
        P.r_row_stride, P.dy_row_stride, P.dx_row_stride = xf.stride(0), dyf.stride(0), dx32.stride(0)
        if drf is not None:
            P.dres_row_stride = drf.stride(0)
        w32 = weight.float().contiguous()
        P.r_ptr, P.weight_ptr, P.mean_ptr, P.rstd_ptr = _ptr(xf), _ptr(w32), _ptr(mean), _ptr(rstd)
        P.dy_ptr, P.dres_ptr, P.dx_ptr, P.dweight_ptr, P.dbias_ptr = _ptr(dyf), _ptr(drf), _ptr(dx32), _ptr(dw), _ptr(db)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dimsum_norm_bwd(P, _stream(x)), "layer_norm_bwd")
    x_dtype = x_dtype or x.dtype
    dx = dx32 if x_dtype == torch.float32 else dx32.to(x_dtype)
    dresidual_in = None
    if has_residual:
        dresidual_in = dx if dx.dtype == x.dtype else dx32.to(x.dtype)
    return dx, dw.to(weight.dtype), db.to(bias.dtype) if bias is not None else None, dresidual_in


def selective_scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, dz, delta_softplus, recompute_out_z, ckpt=None, dB=None, dC=None):
    """-> [du, ddelta, dA, dB, dC, dD, ddelta_bias, (dz), (out_z)] exactly like selective_scan_cuda.bwd
    (selective_scan.cpp:338-492). dz may be a caller-provided view (fused chunk backward, :433-441).
    `ckpt` (extra): the tile-boundary states of selective_scan_fwd(need_ckpt=True); without it they are rebuilt by one
    state-only sweep into a scratch buffer."""
    _check_ssm(u, delta, A, B, C, D, z, delta_bias)
    _gpu(dout, x, out, dz)
    batch, dim, seqlen = u.shape
    _check(tuple(dout.shape) == (batch, dim, seqlen) and dout.dtype == u.dtype and (dout.stride(-1) == 1 or seqlen == 1),
           "selective_scan_bwd: bad dout")
    has_z = z is not None
    if has_z:
        _check(out is not None and out.shape == u.shape and out.stride(-1) == 1, "selective_scan_bwd: `out` of the forward is required with z")
        if dz is None:
            dz = torch.empty_like(z)
        else:
            _check(dz.shape == z.shape and dz.dtype == z.dtype and dz.stride(-1) == 1, "selective_scan_bwd: bad dz")
    out_z = torch.empty_like(out) if (has_z and recompute_out_z) else None
    du = torch.empty_like(u)
    ddelta = torch.empty_like(delta)
    # the three zero-filled fp32 accumulators (selective_scan.cpp:458-466) out of ONE buffer: one fill launch instead of three
    nA, nD = A.numel(), (dim if D is not None else 0)
    acc = _zeros(nA + nD + (dim if delta_bias is not None else 0), u.device)
    dA = acc[:nA].view(A.shape)
    # dB / dC: fp32 then cast (cpp:461-462,488); a caller may hand in fp32 views of B's / C's shape (unit stride along l) to have them written in place
    for t, like in ((dB, B), (dC, C)):
        _check(t is None or (t.shape == like.shape and t.dtype == torch.float32 and t.is_cuda and t.stride(-1) == 1), "selective_scan_bwd: bad dB / dC buffer")
    dB = torch.empty(B.shape, device=u.device, dtype=torch.float32) if dB is None else dB
    dC = torch.empty(C.shape, device=u.device, dtype=torch.float32) if dC is None else dC
    dD = acc[nA:nA + nD] if D is not None else None
    ddelta_bias = acc[nA + nD:] if delta_bias is not None else None
    if u.numel() > 0:
        Q = _lib.SsmBwdParams()
        if ckpt is not None:
            _gpu(ckpt)
            _check(tuple(ckpt.shape) == scan_ckpt_shape(batch, dim, seqlen, A.shape[1]) and ckpt.dtype == torch.float32
                   and ckpt.is_contiguous(), "selective_scan_bwd: bad ckpt")
        _fill_ssm(Q.fwd, u, delta, A, B, C, D, z, delta_bias, delta_softplus, out, x, out_z, ckpt)
        Q.dout_batch_stride, Q.dout_d_stride = dout.stride(0), dout.stride(1)
        Q.dA_d_stride, Q.dA_dstate_stride = dA.stride(0), dA.stride(1)
        Q.dB_batch_stride, Q.dB_group_stride, Q.dB_dstate_stride = dB.stride(0), dB.stride(1), dB.stride(2)
        Q.dC_batch_stride, Q.dC_group_stride, Q.dC_dstate_stride = dC.stride(0), dC.stride(1), dC.stride(2)
        Q.du_batch_stride, Q.du_d_stride = du.stride(0), du.stride(1)
        Q.ddelta_batch_stride, Q.ddelta_d_stride = ddelta.stride(0), ddelta.stride(1)
        if dz is not None:
            Q.dz_batch_stride, Q.dz_d_stride = dz.stride(0), dz.stride(1)
        Q.dout_ptr, Q.dA_ptr, Q.dB_ptr, Q.dC_ptr, Q.dD_ptr = _ptr(dout), _ptr(dA), _ptr(dB), _ptr(dC), _ptr(dD)
        Q.du_ptr, Q.dz_ptr, Q.ddelta_ptr, Q.ddelta_bias_ptr = _ptr(du), _ptr(dz), _ptr(ddelta), _ptr(ddelta_bias)
        lib = _lib.load()
        nbytes = lib.dimsum_ssm_scan_bwd_workspace_bytes(batch, dim, seqlen, A.shape[1], B.shape[1])
        if ckpt is not None:
            nbytes -= ckpt.numel() * 4          # the states part is only needed when they must be rebuilt
        ws = torch.empty((nbytes + 3) // 4, device=u.device, dtype=torch.float32)       # per-wave partial dB / dC (+ states)
        Q.workspace_ptr, Q.workspace_bytes = _ptr(ws), nbytes
        with torch.cuda.device(u.device):
            _lib.check(lib.dimsum_ssm_scan_bwd(Q, _stream(u)), "selective_scan_bwd")
    res = [du, ddelta, dA, dB.to(B.dtype), dC.to(C.dtype), dD, ddelta_bias]
    if has_z:
        res.append(dz)
    if out_z is not None:
        res.append(out_z)
    return res


# ---------------------------------------------------------------------------------------------------------------------
# fused token-space transform (csrc/token_transform.hip) and GatedMLP epilogue
# ---------------------------------------------------------------------------------------------------------------------
_TT_KIND = {("none", True): 0, ("none", False): 0, ("haar", True): 1, ("haar", False): 2, ("dct", True): 3, ("dct", False): 4}


def token_transform(x, kind, forward, in_index=None, out_index=None, gate=None, scale=None, shift=None, residual=None,
                    w=None, want_y=True, want_wsum=False, want_tsum=False, split3=False):
    """y[b, out_index[s], c] = T(x[b, in_index[s], c] * gate[b, c])[s] * (1 + scale[b, c]) + shift[b, c] + residual[b, out_index[s], c]
    x: (B, L, C) fp32 with unit channel stride (may be a channel slice of a wider tensor); index tables int32 (L,).
    With a weight tensor `w` (indexed like y) the same pass also reduces, per (batch, channel),
        wdot = sum_s T(.)[s, c] * w[b, out_index[s], c]     and (want_wsum)  wsum = sum_s w[b, out_index[s], c]
    and returns (y or None, wdot, wsum or None) -- the adaLN-modulation gradients of the block backward.
    `want_tsum` appends tsum = sum_s T(.)[s, c] (the plain token sum) to the returned tuple.
    `split3`: y is returned as the split-bf16 left operand image (B, L, 3C) bfloat16 of the Linear that consumes it (split3_rows);
    `split3="f16s"`: as a scaled-fp16 image (F16Image: data (B, L, C) float16, inv (B, L); C <= 1024)."""
    _gpu(x, in_index, out_index, gate, scale, shift, residual, w)
    _check(x.dim() == 3 and x.dtype == torch.float32 and x.stride(2) == 1, "token_transform: x must be (B, L, C) float32, channel-contiguous")
    B, L, C = x.shape
    grid = int(round(L ** 0.5))
    if kind != "none":
        _check(grid * grid == L and grid % 4 == 0, "token_transform: the token grid must be square with side % 4 == 0")
    _check(want_y or w is not None or want_tsum, "token_transform: nothing to compute")
    y_inv = None
    if split3 == "f16s":
        _check(want_y and C % 4 == 0 and C <= (2048 if kind == "none" else 1024) and w is None and not want_tsum,
               "token_transform: the f16s image needs channels % 4 == 0, <= 1024 (2048 without a transform), no reductions")
        y = torch.empty((B, L, C), device=x.device, dtype=torch.float16)
        y_inv = torch.empty((B, L), device=x.device, dtype=torch.float32)
    elif split3:
        _check(want_y and C % 4 == 0, "token_transform: split3 needs an output and channels % 4 == 0")
        y = torch.empty((B, L, (2 if split3 == "pair" else 3) * C), device=x.device, dtype=torch.bfloat16)
    else:
        y = torch.empty((B, L, C), device=x.device, dtype=torch.float32) if want_y else None
    mods = [m for m in (gate, scale, shift) if m is not None]
    mstride = 0
    for m in mods:
        _check(m.dtype == torch.float32 and tuple(m.shape) == (B, C) and m.stride(1) == 1, "token_transform: gate/scale/shift must be (B, C) float32")
        _check(mstride in (0, m.stride(0)), "token_transform: gate/scale/shift must share one row stride")
        mstride = m.stride(0)
    for t in (in_index, out_index):
        if t is not None:
            _check(t.dtype == torch.int32 and t.numel() == L and t.is_contiguous(), "token_transform: index tables must be int32 (L,)")
    if residual is not None:
        _check(residual.shape == x.shape and residual.dtype == torch.float32 and residual.stride(2) == 1, "token_transform: bad residual")
    wdot = wsum = tsum = None
    if w is not None:
        _check(w.shape == x.shape and w.dtype == torch.float32 and w.stride(2) == 1, "token_transform: bad w")
    if w is not None or want_tsum:
        red = _zeros(3 * B * C, x.device).view(3, B, C)
        wdot, wsum = (red[0] if w is not None else None), (red[1] if (w is not None and want_wsum) else None)
        tsum = red[2] if want_tsum else None
    if B > 0:
        P = _lib.TtParams()
        P.batch, P.tokens, P.channels, P.grid, P.kind = B, L, C, grid, _TT_KIND[(kind, bool(forward))]
        P.y_split3 = _SPLIT_MODE.get(split3, int(bool(split3)))
        P.y_inv_scale_ptr = _ptr(y_inv)
        P.x_batch_stride, P.x_token_stride = x.stride(0), x.stride(1)
        if y is not None:
            P.y_batch_stride, P.y_token_stride = y.stride(0), y.stride(1)
        if residual is not None:
            P.res_batch_stride, P.res_token_stride = residual.stride(0), residual.stride(1)
        P.red_batch_stride = C
        if w is not None:
            P.w_batch_stride, P.w_token_stride = w.stride(0), w.stride(1)
        P.mod_batch_stride = mstride
        P.x_ptr, P.in_index_ptr, P.out_index_ptr = _ptr(x), _ptr(in_index), _ptr(out_index)
        P.gate_ptr, P.scale_ptr, P.shift_ptr, P.residual_ptr, P.y_ptr = _ptr(gate), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y)
        P.w_ptr, P.wdot_ptr, P.wsum_ptr, P.tsum_ptr = _ptr(w), _ptr(wdot), _ptr(wsum), _ptr(tsum)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dimsum_token_transform(P, _stream(x)), "token_transform")
    if y_inv is not None:
        return F16Image(y, y_inv)
    if split3 == "pair":
        y = PairImage(y)
    if want_tsum:
        return (y, wdot, wsum, tsum)
    return y if w is None else (y, wdot, wsum)


def gated_gelu_fwd(x12, bias=None, split3=False):
    """x12: (..., 2H) fp32 contiguous (w12 GEMM output WITHOUT bias), bias (2H) or None
    -> gelu_tanh(x12[..., :H] + bias[:H]) * (x12[..., H:] + bias[H:])   (mlp.py:66-70)
    split3: the result as the split-bf16 left operand image (..., 3H) bfloat16 of the w3 GEMM (split3_rows)"""
    _gpu(x12, bias)
    _check(x12.dtype == torch.float32 and x12.is_contiguous() and x12.shape[-1] % 8 == 0, "gated_gelu: x12 must be contiguous float32 with 2H % 8 == 0")
    H = x12.shape[-1] // 2
    if bias is not None:
        _check(bias.dtype == torch.float32 and tuple(bias.shape) == (2 * H,) and bias.is_contiguous(), "gated_gelu: bias must be (2H,) float32")
    h = torch.empty(x12.shape[:-1] + ((3 * H,) if split3 else (H,)), device=x12.device, dtype=torch.bfloat16 if split3 else torch.float32)
    rows = x12.numel() // (2 * H)
    fn = _lib.load().dimsum_gated_gelu_fwd_split3 if split3 else _lib.load().dimsum_gated_gelu_fwd
    with torch.cuda.device(x12.device):
        _lib.check(fn(_ptr(x12), _ptr(bias), _ptr(h), rows, H, _stream(x12)), "gated_gelu_fwd")
    return h


def split3_rows(x, left):
    """(R, K) float32 rows (stride(1) == 1) -> (R, 3K) bfloat16 split operand image: x = hi + lo, hi = bf16(x), lo = bf16(x - hi);
    left: [hi | hi | lo] (activations), else [hi | lo | hi] (weights), so that  left_image @ weight_image.T  accumulates
    hi.hi + hi.lo + lo.hi -- the three products of hipBLASLt's fp32-under-allow_tf32 path -- in ONE bf16 GEMM (csrc/operand_split.hip)."""
    _gpu(x)
    _check(x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[1] % 4 == 0 and x.stride(0) % 4 == 0,
           "split3_rows: x must be (R, K) float32 with contiguous rows, K % 4 == 0 and a row stride % 4 == 0")
    R, K = x.shape
    pair = isinstance(left, str) and left == "pair"          # the pair [hi | lo] (PairImage): the consumer chooses the reading order
    out = torch.empty((R, (2 if pair else 3) * K), device=x.device, dtype=torch.bfloat16)
    if R > 0:
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dimsum_split3(_ptr(x), R, K, x.stride(0), _ptr(out), 2 if pair else int(bool(left)), _stream(x)), "split3_rows")
    return PairImage(out) if pair else out


def split3_rows_t(w):
    """weight (N, K) float32 (rows contiguous) -> (3 K, N) bfloat16: the row stack [hi; lo; hi] of w^T (csrc/operand_split.hip), the right
    operand of gemm_tn(planes, ., alias_rows=K)"""
    _gpu(w)
    _check(w.dim() == 2 and w.dtype == torch.float32 and w.stride(1) == 1 and w.shape[0] % 2 == 0, "split3_rows_t: w must be (N, K) float32 rows, N even")
    N, K = w.shape
    out = torch.empty((3 * K, N), device=w.device, dtype=torch.bfloat16)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().dimsum_split3_t(_ptr(w), N, K, w.stride(0), _ptr(out), _stream(w)), "split3_rows_t")
    return out


class PairImage:
    """split-bf16 left operand image stored as the PAIR [hi | lo]: data (M, 2K) bfloat16. The GEMM kernel reads it as [hi | hi | lo]
    (a_alias_rows = K: K tiles past the first piece re-read the tiles one piece earlier) -- a third less image traffic than the three-piece
    form the library GEMM needs; .image3() expands it for a consumer without the kernel."""
    __slots__ = ("data",)

    def __init__(self, data):
        self.data = data

    @property
    def k(self):
        return self.data.shape[-1] // 2

    def image3(self):
        K = self.k
        return torch.cat([self.data[..., :K], self.data[..., :K], self.data[..., K:]], dim=-1)

    @property
    def shape(self):
        return self.data.shape

    def reshape(self, *shape):
        return PairImage(self.data.reshape(*shape))

    view = reshape

    def record_stream(self, stream):
        self.data.record_stream(stream)


class F16Image:
    """scaled-fp16 operand image (csrc/common.hpp, f16s): data (..., K) float16 = fp16(x * 2^s), one s per row, inv (...) float32 = 2^-s.
    Quacks like the tensor it replaces where the host layer only reshapes it and hands it to the next GEMM."""
    __slots__ = ("data", "inv")

    def __init__(self, data, inv):
        self.data, self.inv = data, inv

    @property
    def shape(self):
        return self.data.shape

    def reshape(self, *shape):
        d = self.data.reshape(*shape)
        return F16Image(d, self.inv.reshape(d.shape[:-1]))

    view = reshape

    def record_stream(self, stream):
        self.data.record_stream(stream)
        self.inv.record_stream(stream)

    def float(self):
        return self.data.float() * self.inv.unsqueeze(-1)


def rows_f16s(x, want_l1=False):
    """(R, K) float32 rows (stride(1) == 1) -> F16Image [, l1max]: row r = fp16(x_r * 2^s_r) with 2^s_r max|x_r| in [2^14, 2^15) and
    inv[r] = 2^-s_r (csrc/operand_split.hip). want_l1: also max_r sum_k |x_rk| as a 1-element float32 tensor (weights: the bound of
    the gated GEMM epilogue)."""
    _gpu(x)
    _check(x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[1] % 4 == 0 and x.stride(0) % 4 == 0,
           "rows_f16s: x must be (R, K) float32 with contiguous rows, K % 4 == 0 and a row stride % 4 == 0")
    R, K = x.shape
    out = torch.empty((R, K), device=x.device, dtype=torch.float16)
    inv = torch.empty((R,), device=x.device, dtype=torch.float32)
    l1 = torch.zeros((1,), device=x.device, dtype=torch.float32) if want_l1 else None
    if R > 0:
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dimsum_rows_f16s(_ptr(x), R, K, x.stride(0), _ptr(out), K, _ptr(inv), _ptr(l1), _stream(x)), "rows_f16s")
    img = F16Image(out, inv)
    return (img, l1) if want_l1 else img


def rows_block_f16s(x):
    """(D, M) float32 rows (a d-major activation: channels x tokens; D % 64 == 0, M % 256 == 0) -> (image (D, M) float16, table (M / 32, D / 64) float32):
    every 64 x 32 block as fp16(x 2^s) with 2^-s in the table -- the layout selective_scan_fwd(out_z_f16=True) returns, for launches the state-split scan
    kernels serve (csrc/operand_split.hip, rows_block_f16s_kernel)"""
    _gpu(x)
    _check(x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[0] % 64 == 0 and x.shape[1] % 256 == 0 and x.stride(0) % 4 == 0
           and x.data_ptr() % 16 == 0, "rows_block_f16s: x must be (D % 64, M % 256) float32 rows, 16-byte aligned")
    D, M = x.shape
    img = torch.empty((D, M), device=x.device, dtype=torch.float16)
    table = torch.empty((M // 32, D // 64), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dimsum_rows_block_f16s(_ptr(x), D, M, x.stride(0), _ptr(img), M, _ptr(table), table.stride(0), _stream(x)), "rows_block_f16s")
    return img, table


def rows_f16s_multi(jobs, n_slots=None):
    """ONE launch per 24 jobs (csrc/operand_split.hip, dimsum_rows_f16s_multi). jobs: list of (x, image, l1_slot, absmax_slot, l1_factor):
    x (R, K) float32 rows (or a (K,) vector: one row); image: build the F16Image; l1_slot / absmax_slot: index into the returned float32
    scalar buffer that receives l1_factor * max_r sum_k |x_rk| / max |x| (None: not wanted; several jobs may NOT share a slot's meaning but
    may share a slot: the maximum over them lands there); an optional 6th element (data (R, K) float16, inv (R,) float32) makes the job
    write its image there (e.g. slices of one buffer). -> ([F16Image or None per job], scalars)"""
    if not jobs:
        return [], None
    dev = jobs[0][0].device
    used = 1 + max([-1] + [s_ for j in jobs for s_ in j[2:4] if s_ is not None])
    n_slots = used if n_slots is None else max(int(n_slots), used)        # (a caller that slices the scalars by its own slot layout passes its total)
    scal = torch.zeros(max(n_slots, 1), device=dev, dtype=torch.float32)
    arr = (_lib.F16sJob * len(jobs))()
    images, keep = [], []
    for q, job in zip(arr, jobs):
        x, image, l1_slot, abs_slot, factor = job[:5]
        _gpu(x)
        x2 = x if x.dim() == 2 else x.reshape(1, -1)
        _check(x2.dtype == torch.float32 and x2.dim() == 2 and x2.stride(1) == 1 and x2.shape[1] % 4 == 0 and (x2.shape[0] == 1 or x2.stride(0) % 4 == 0),
               "rows_f16s_multi: operands must be float32 rows with K % 4 == 0 and a row stride % 4 == 0")
        R, K = x2.shape
        q.src, q.rows, q.cols, q.src_row_stride = _ptr(x2), R, K, (x2.stride(0) if R > 1 else K)
        if image:
            if len(job) > 5:
                out, inv = job[5]
                _check(out.shape == (R, K) and out.dtype == torch.float16 and out.is_contiguous() and inv.shape == (R,) and inv.dtype == torch.float32 and inv.is_contiguous(),
                       "rows_f16s_multi: a job's own destination must be a contiguous (R, K) float16 / (R,) float32 pair")
            else:
                out, inv = torch.empty((R, K), device=dev, dtype=torch.float16), torch.empty((R,), device=dev, dtype=torch.float32)
            q.dst, q.inv_scale_ptr, q.dst_row_stride = _ptr(out), _ptr(inv), K
            images.append(F16Image(out, inv))
        else:
            images.append(None)
        if l1_slot is not None:
            q.l1max_ptr = scal.data_ptr() + 4 * l1_slot
        if abs_slot is not None:
            q.absmax_ptr = scal.data_ptr() + 4 * abs_slot
        q.l1_factor = float(factor)
        keep.append(x2)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().dimsum_rows_f16s_multi(arr, len(jobs), _stream(jobs[0][0])), "rows_f16s_multi")
    return images, scal


def gemm_nt_supported(a, b, gated=False, pair=False, pair_b=False):
    """shapes the hand-written NT GEMM takes (csrc/gemm_nt_kernel.hpp): 256-row panels of 16-bit rows, 64-deep K tiles.
    pair / pair_b (or a PairImage operand): that operand is the [hi | lo] pair (rows of 2C) of a split-bf16 image over K = 3C (C % 64 == 0)"""
    if isinstance(a, PairImage):
        a, pair = a.data, True
    if isinstance(b, PairImage):
        b, pair_b = b.data, True
    if not (a.is_cuda and a.dim() == 2 and b.dim() == 2 and a.dtype == b.dtype and a.dtype in (torch.bfloat16, torch.float16)):
        return False
    M, K = a.shape
    N, Kb = b.shape
    if pair:
        if K % 128 != 0 or a.dtype != torch.bfloat16:
            return False
        K = K // 2 * 3
    if pair_b:
        if pair or gated or Kb % 128 != 0 or b.dtype != torch.bfloat16:
            return False
        Kb = Kb // 2 * 3
    return (Kb == K and M > 0 and M % 256 == 0 and K % 64 == 0 and K >= 128 and N % (16 if gated else 4) == 0
            and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and 512 * max(a.stride(0), b.stride(0)) < 2 ** 31)


def gemm_tn_supported(a, b):
    """shapes the TN variant takes (csrc/gemm_nt_kernel.hpp, kVarTN): a (R, P), b (R, Q) 16-bit rows over the reduction index R, whole
    256 x 256 output tiles, 64-row reduction tiles"""
    if not (a.is_cuda and a.dim() == 2 and b.dim() == 2 and a.dtype == b.dtype and a.dtype in (torch.bfloat16, torch.float16)):
        return False
    R, P = a.shape
    Q = b.shape[1]
    # (Q % 256 != 0: b must be a column slice of rows ZERO-PADDED to whole 256-column tiles -- its row stride says so; gemm.weight_f16s_t pads)
    q_ok = Q % 256 == 0 or (Q % 4 == 0 and b.stride(0) >= (Q + 255) // 256 * 256)
    return (b.shape[0] == R and R % 64 == 0 and R >= 128 and P % 256 == 0 and q_ok and P > 0 and Q > 0
            and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and 128 * max(a.stride(0), b.stride(0)) + 512 < 2 ** 31)


def gemm_tn_splits(R, P, Q, second_round=True):
    """reduction ranges of a TN launch: enough workgroups to fill the 256 CUs when the output has few tiles (a weight gradient (8192, 1024)
    is 128 tiles, (1536, 512) is 12), ranges of whole 64-row tiles, at least 2048 rows each. second_round=False (the single-product fp16
    launches: a third of the three-product kernel's time per range, so the partial results' write + sum weigh three times as much): one
    round of workgroups -- tools/scratch/tn_splits.py: d w12 at 16384 rows 335 -> 297 us, d w3 180 -> 144"""
    tiles = (P // 256) * (Q // 256)
    s = 1
    ok = lambda s2: R % (s2 * 64) == 0 and R // s2 >= 2048
    while tiles * s < 192 and ok(2 * s):         # three quarters of the 256 CUs at least ...
        s *= 2
    if second_round and tiles >= 64 and tiles * s < 512 and ok(2 * s):       # ... and a second round where the partial results are few (tools/scratch/tn_perf.py)
        s *= 2
    return s


def gemm_tn_pairs(a, b, splits=None):
    """a, b PairImages (M, 2P) / (M, 2Q): dW = a^T b with a read in weight order [hi | lo | hi] and b in left order [hi | hi | lo] (the three
    products of the split-bf16 policy) -> (P, Q) float32. The reduction runs over 3 pieces x `splits` row ranges of M."""
    ad, bd = a.data, b.data
    _gpu(ad, bd)
    M, P2 = ad.shape
    P, Q = P2 // 2, bd.shape[1] // 2
    _check(bd.shape[0] == M and gemm_tn_supported(ad[:, :P], bd[:, :Q]), "gemm_tn_pairs: unsupported operands")
    if splits is None:
        tiles = (P // 256) * (Q // 256)
        # whole rounds of the 256 CUs: 3 x splits x tiles workgroups of equal length (128 tiles x 3 = 384 would be one and a half)
        splits = 1
        while (tiles * 3 * splits) % 256 != 0 and tiles * 3 * splits < 1024 and M % (2 * splits * 64) == 0 and M // (2 * splits) >= 2048:
            splits *= 2
    _check(M % (splits * 64) == 0 and M // splits >= 128, "gemm_tn_pairs: splits must cut M into ranges of whole 64-row tiles")
    out = torch.empty((3 * splits, P, Q), device=ad.device, dtype=torch.float32)
    G = _lib.GemmParams()
    G.m, G.n, G.k = P, Q, M
    G.operand_dtype, G.epilogue, G.out_scale = _DT[ad.dtype], _lib.GEMM_EPI_F32, 1.0
    G.lda, G.ldb, G.ldc = ad.stride(0), bd.stride(0), Q
    G.a_ptr, G.b_ptr, G.c_ptr = _ptr(ad), _ptr(bd), _ptr(out)
    X = _lib.attach_ext(G, _lib.GemmExt)
    X.tn_pair_a_cols, X.tn_pair_b_cols = P, Q
    with torch.cuda.device(ad.device):
        _lib.check(_lib.load().dimsum_gemm_tn(G, 3 * splits, P * Q, _stream(ad)), "gemm_tn_pairs")
    return out.sum(0)


def row_factors(a_inv, b_inv=None):
    """(R,) float32 inverse row scales of two scaled-fp16 images whose ROWS are the reduction index of a product a^T b -> (k_fac (R,) float16,
    c_scale (1,) float32): term r carries a_inv[r] b_inv[r]; k_fac = that / its maximum (powers of two <= 1; below 2^-24: 0), c_scale = the
    maximum (include/dimsum_hip.h, dimsum_gemm_ext_t.k_scale_ptr). b_inv None = 1. One small launch (csrc/operand_split.hip)."""
    _gpu(a_inv, b_inv)
    a_inv = a_inv.reshape(-1)
    _check(a_inv.dtype == torch.float32 and a_inv.is_contiguous() and (b_inv is None or (b_inv.dtype == torch.float32 and b_inv.numel() == a_inv.numel())),
           "row_factors: (R,) float32 inverse scales")
    if b_inv is not None:
        b_inv = b_inv.reshape(-1).contiguous()
    fac = torch.empty(a_inv.numel(), device=a_inv.device, dtype=torch.float16)
    top = torch.empty(1, device=a_inv.device, dtype=torch.float32)
    with torch.cuda.device(a_inv.device):
        _lib.check(_lib.load().dimsum_row_factors(_ptr(a_inv), _ptr(b_inv), a_inv.numel(), _ptr(fac), _ptr(top), _stream(a_inv)), "row_factors")
    return fac, top


def gemm_tn(a, b, splits=None, events=None, alias_rows=0, scales=None, row_scales=None, row_invs=None):
    """a (R, P)^T @ b (R, Q) -> (P, Q) float32 on the hand-written MFMA kernel's TN variant: the weight-gradient product of a Linear
    (reduction over the rows). The reduction is cut into `splits` ranges whose partial results are added in a fixed order.
    alias_rows = D: a is a (2 D, P) pair of planes [hi; lo] read as the row stack [hi; hi; lo] (b: (3 D, Q)); one range.
    scales = (a_inv, b_inv (Q,)): float16 operands carrying power-of-two scales (one range): a_inv (P,) -> the result is multiplied by
    a_inv[p] b_inv[q]; a_inv (P / 32, R / 64) -> block-scaled a (the scan's fp16 out_z with its table, selective_scan_fwd(out_z_f16=True)):
    the values of tokens [32 g, 32 g + 32) x reduction rows [64 t, 64 t + 64) stand for value * a_inv[g][t]; R <= 4096 (out_proj of a
    Mamba mixer as ONE fp16 product per element).
    row_scales = (k_fac (R,) float16, c_scale (1,) float32) from row_factors(a_inv, b_inv): both operands are scaled-fp16 images with one scale
    per ROW and the rows are the reduction index (the weight gradient dW = dy^T x of a Linear under the scaled-fp16 policy): a's row r is
    multiplied by k_fac[r] as it is read, the result by c_scale; ranges of at most 16384 rows.
    row_invs = (a_inv (R,), b_inv (R,) or None) float32: the same product with the factors formed INSIDE the kernel from the two images' row scales
    (every workgroup normalises by the maximum of its own range): no row_factors launch in front of the GEMM."""
    _gpu(a, b)
    blocks = None
    if row_invs is not None:
        ia, ib = row_invs
        _gpu(ia, ib)
        _check(row_scales is None and scales is None and not alias_rows and a.dtype == torch.float16 and ia.dtype == torch.float32 and ia.numel() == a.shape[0]
               and ia.is_contiguous() and (ib is None or (ib.dtype == torch.float32 and ib.numel() == a.shape[0] and ib.is_contiguous())),
               "gemm_tn: row_invs = (a_inv (R,), b_inv (R,) or None) float32 with float16 operands")
        if os.environ.get("DIMSUM_ROW_FACTORS_KERNEL", "0") == "1":        # (A / B: the factor table from its own launch, as before)
            row_scales, row_invs = row_factors(ia, ib), None
    if row_scales is not None:
        kf, cs = row_scales
        _gpu(kf, cs)
        _check(scales is None and not alias_rows and a.dtype == torch.float16 and kf.dtype == torch.float16 and kf.numel() == a.shape[0] and kf.is_contiguous()
               and cs.dtype == torch.float32 and cs.numel() == 1, "gemm_tn: row_scales = (k_fac (R,) float16, c_scale (1,) float32) with float16 operands")
    if scales is not None:
        _gpu(*scales)
        sa, sb = scales
        _check(a.dtype == torch.float16 and not alias_rows and sa.dtype == torch.float32 and sb.dtype == torch.float32 and sb.numel() == b.shape[1] and sb.is_contiguous(),
               "gemm_tn: scales = (a_inv, b_inv (Q,)) float32 with float16 operands")
        if sa.dim() == 2:
            _check(sa.shape[0] == a.shape[1] // 32 and sa.shape[1] >= a.shape[0] // 64 and sa.stride(1) == 1 and a.shape[0] <= 4096 and a.shape[1] % 32 == 0,
                   "gemm_tn: a block table must be (P / 32, R / 64) float32, R <= 4096")
            blocks = sa
        else:
            _check(sa.numel() == a.shape[1] and sa.is_contiguous(), "gemm_tn: a_inv must be (P,)")
        splits = 1
    if alias_rows:
        _check(a.dim() == 2 and a.shape[0] == 2 * alias_rows and b.shape[0] == 3 * alias_rows and alias_rows % 64 == 0 and gemm_tn_supported(a[:alias_rows], b[:alias_rows]),
               "gemm_tn: alias_rows = D takes a (2 D, P) plane pair and a (3 D, Q) row stack, D % 64 == 0")
        R, P = 3 * alias_rows, a.shape[1]
        splits = 1
    else:
        _check(gemm_tn_supported(a, b), "gemm_tn: unsupported operands (R % 64, P % 256, Q % 256, 16-bit rows, 16-byte aligned)")
        R, P = a.shape
    Q = b.shape[1]
    if splits is None:
        rowfac = row_scales is not None or row_invs is not None
        splits = gemm_tn_splits(R, P, Q, second_round=not rowfac)
        while rowfac and R // splits > 16384 and R % (2 * splits * 64) == 0:          # (the factors of one range live in 32 KB of LDS)
            splits *= 2
    if row_scales is not None or row_invs is not None:
        _check(R // splits <= 16384, "gemm_tn: row factors need ranges of at most 16384 reduction rows")
    _check(splits >= 1 and R % (splits * 64) == 0 and R // splits >= 128, "gemm_tn: splits must cut R into ranges of whole 64-row tiles (>= 2)")
    out = torch.empty((splits, P, Q), device=a.device, dtype=torch.float32)
    G = _lib.GemmParams()
    G.m, G.n, G.k = P, Q, R
    G.operand_dtype = _DT[a.dtype]
    G.epilogue = _lib.GEMM_EPI_F32
    G.out_scale = 1.0
    G.lda, G.ldb, G.ldc = a.stride(0), b.stride(0), Q
    G.a_ptr, G.b_ptr, G.c_ptr = _ptr(a), _ptr(b), _ptr(out)
    X = _lib.attach_ext(G, _lib.GemmExt)
    X.a_alias_rows = alias_rows
    if scales is not None:
        G.b_inv_scale_ptr = _ptr(scales[1])
        if blocks is not None:
            X.a_block_inv_ptr, X.a_block_inv_ld = _ptr(blocks), blocks.stride(0)
        else:
            G.a_inv_scale_ptr = _ptr(scales[0])
    if row_scales is not None:
        X.k_scale_ptr, X.c_scale_ptr = _ptr(row_scales[0]), _ptr(row_scales[1])
    if row_invs is not None:
        X.k_inv_a_ptr, X.k_inv_b_ptr = _ptr(row_invs[0]), _ptr(row_invs[1])
    if events is not None:
        X.timing_start_event, X.timing_stop_event = events
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().dimsum_gemm_tn(G, splits, P * Q, _stream(a)), "gemm_tn")
    return out[0] if splits == 1 else out.sum(0)


_gemm_kernel_log = None      # a list while gemm_kernel_log() is open: (epilogue, dimsum_gemm_nt_kernel_for) of every gemm_nt call


@contextlib.contextmanager
def gemm_kernel_log():
    """tests / measurement: records, for every gemm_nt call made inside, which kernel family the library launches (_lib.GEMM_NT_KERNELS:
    0 = 256-row tiles, 1 = 128-row tiles, 2 = the persistent K-tile stream). Host-side, not thread-safe."""
    global _gemm_kernel_log
    old, _gemm_kernel_log = _gemm_kernel_log, []
    try:
        yield _gemm_kernel_log
    finally:
        _gemm_kernel_log = old


def gemm_nn_supported(a, b):
    """shapes dimsum_gemm_nn takes: a (P, R) float16 rows contiguous along the reduction, b (R, Q) float16 rows over it; whole 256 x 256 output
    tiles, whole 64-row reduction tiles"""
    if not (a.is_cuda and a.dim() == 2 and b.dim() == 2 and a.dtype == torch.float16 and b.dtype == torch.float16):
        return False
    P, R = a.shape
    Q = b.shape[1]
    return (b.shape[0] == R and R % 64 == 0 and R >= 128 and P % 256 == 0 and Q % 256 == 0 and a.stride(1) == 1 and b.stride(1) == 1
            and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
            and 256 * a.stride(0) * 2 < 2 ** 31 and 64 * b.stride(0) * 2 + 512 < 2 ** 31)


def gemm_nn(a, a_inv, b, b_inv, splits=None):
    """a (P, R) @ b (R, Q) -> (P, Q) float32 from two scaled-fp16 images: a = fp16 rows with one scale per row (a_inv (P,): its rows are OUTPUT rows --
    a d-major activation, channels x tokens), b = fp16 rows with one scale per row (b_inv (R,): its rows are the REDUCTION index -- a token-major
    activation): the weight gradients of the Mamba projections (d in_proj.weight = dxz x, d out_proj.weight^T = out_z dout) as ONE fp16 product
    per element. b's scales travel as per-reduction-row factors (row_factors with a constant partner)."""
    _gpu(a, a_inv, b, b_inv)
    _check(gemm_nn_supported(a, b) and a_inv.dtype == torch.float32 and a_inv.numel() == a.shape[0] and a_inv.is_contiguous()
           and b_inv.dtype == torch.float32 and b_inv.numel() == b.shape[0], "gemm_nn: unsupported operands")
    P, R = a.shape
    Q = b.shape[1]
    if splits is None:
        splits = gemm_tn_splits(R, P, Q, second_round=False)
        while R // splits > 16384 and R % (2 * splits * 64) == 0:
            splits *= 2
    _check(splits >= 1 and R % (splits * 64) == 0 and 128 <= R // splits <= 16384, "gemm_nn: splits must cut R into ranges of 2 .. 256 whole 64-row tiles")
    in_kernel = os.environ.get("DIMSUM_ROW_FACTORS_KERNEL", "0") != "1" and b_inv.is_contiguous()       # factors formed inside the GEMM (no launch in front)
    fac, top = (None, None) if in_kernel else row_factors(b_inv)
    out = torch.empty((splits, P, Q), device=a.device, dtype=torch.float32)
    G = _lib.GemmParams()
    G.m, G.n, G.k = P, Q, R
    G.operand_dtype, G.epilogue, G.out_scale = _DT[a.dtype], _lib.GEMM_EPI_F32, 1.0
    G.lda, G.ldb, G.ldc = a.stride(0), b.stride(0), Q
    G.a_ptr, G.b_ptr, G.c_ptr, G.a_inv_scale_ptr = _ptr(a), _ptr(b), _ptr(out), _ptr(a_inv)
    X = _lib.attach_ext(G, _lib.GemmExt)
    if in_kernel:
        X.k_inv_a_ptr = _ptr(b_inv)
    else:
        X.k_scale_ptr, X.c_scale_ptr = _ptr(fac), _ptr(top)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().dimsum_gemm_nn(G, splits, P * Q, _stream(a)), "gemm_nn")
    return out[0] if splits == 1 else out.sum(0)


def gemm_nt(a, b, bias=None, epilogue="f32", out=None, out_scale=1.0, events=None, tune=None, scales=None, gate_bound=None, residual=None,
            gate=None, rows_per_batch=None, keep_x12=False, pair_out=False, weight_order=False, q_cols=None, conv=None):
    """a (M, K) @ b (N, K)^T on the hand-written MFMA kernel, 16-bit operands (bfloat16: split-bf16 images over 3 K; float16: scaled rows),
    fp32 accumulation.
      epilogue "f32"          -> (M, N) float32 (+ bias[N])
               "gated_split3" -> b = the (2F, K) w12 weight image, bias (2F) or None: the LEFT split-bf16 image (M, 3F) bfloat16 of
                                 gelu_tanh(x1 + b1) * (x2 + b2)   (mlp.py:66-70; the fp32 (M, 2F) x12 never exists)
                                 keep_x12 (training forward): the bias-free float32 (M, 2F) [x1 | x2] is stored too -> (image, x12)
               "gated_f16"    -> same, (M, F) float16 of h * out_scale
    residual (M, N) float32 [, gate (M / rows_per_batch, N) float32]: epilogue "f32" becomes residual + gate[row // rows_per_batch] * (a b^T + bias)
    -- the residual tail of a block in the epilogue of its last Linear (rows_per_batch % 256 == 0).
    scales: (a_inv (M,), b_inv (N,)) float32 inverse scales of scaled-fp16 operand images (F16Image): C[m, n] *= a_inv[m] * b_inv[n].
    gate_bound ("gated_f16" over scaled operands): 2-element float32 device tensor {max_n sum_k |w_nk|, max |bias|}: the h image gets a
    per-row power-of-two scale derived from it and the call returns (h16, h_inv).
    events: optional (start, stop) raw hipEvent_t handles recorded at the kernel's dispatch boundaries (bench.py)."""
    pair_in = isinstance(a, PairImage)         # a: the pair [hi | lo] of a left image, read as [hi | hi | lo]
    pair_b = isinstance(b, PairImage)          # b: the same on the right (in_proj: weight image x activation pair -> d-major output)
    if pair_in:
        a = a.data
    if pair_b:
        b = b.data
    _gpu(a, b, bias)
    gated = epilogue in ("gated_split3", "gated_f16")
    _check(not (pair_in and pair_b) and not (pair_b and (gated or residual is not None or bias is not None)), "gemm_nt: a right-hand pair goes with the plain fp32 epilogue")
    _check(gemm_nt_supported(a, b, gated, pair=pair_in, pair_b=pair_b), "gemm_nt: unsupported operands (M % 256, K % 64, K >= 128, 16-bit K-contiguous rows, 16-byte aligned)")
    M, K = a.shape[0], (a.shape[1] if pair_b else b.shape[1])
    N = b.shape[0]
    P = _lib.GemmParams()
    X = _lib.attach_ext(P, _lib.GemmExt)            # fused-epilogue operands, image read modes, timing, tuning (dimsum_gemm_ext_t)
    P.m, P.n, P.k = M, N, K
    if pair_in:
        X.a_alias_rows = K // 3
        X.a_alias_weight_order = int(bool(weight_order))       # the pair read as [hi | lo | hi] (a gradient image x a left-order weight image)
    if pair_b:
        X.b_alias_rows = K // 3
    P.operand_dtype = _DT[a.dtype]
    P.out_scale = float(out_scale)
    P.lda, P.ldb = a.stride(0), b.stride(0)
    if bias is not None:
        _check(bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N, "gemm_nt: bias must be (N,) float32")
    if epilogue == "f32":
        P.epilogue = _lib.GEMM_EPI_F32 if bias is None else _lib.GEMM_EPI_F32_BIAS
        if out is None:
            out = torch.empty((M, N), device=a.device, dtype=torch.float32)
        _check(out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1, "gemm_nt: out must be (M, N) float32 rows")
        if conv is not None:
            # conv = (weight (rows, width) f32, bias (rows) f32 or None, seq): rows [0, rows) of the d-major product get the causal conv1d + SiLU
            # along their columns (sequences of `seq` columns) in the epilogue -- in_proj + causal_conv1d_fn of a Mamba mixer in one kernel
            cw, cb, seq = conv
            _gpu(cw, cb)
            _check(bias is None and residual is None and cw.dtype == torch.float32 and cw.dim() == 2 and cw.stride(1) == 1 and cw.shape[0] % 256 == 0
                   and cw.shape[0] <= M and 2 <= cw.shape[1] <= 4 and 256 % seq == 0 and seq % 4 == 0 and N % seq == 0
                   and (cb is None or (cb.dtype == torch.float32 and cb.is_contiguous() and cb.numel() == cw.shape[0])),
                   "gemm_nt: conv = (weight (rows % 256 == 0, width 2..4) f32, bias or None, seq with 256 % seq == 0)")
            P.epilogue = _lib.GEMM_EPI_F32_CONV
            X.conv_weight_ptr, X.conv_bias_ptr = _ptr(cw), _ptr(cb)
            X.conv_rows, X.conv_width, X.conv_seq, X.conv_weight_ld = cw.shape[0], cw.shape[1], seq, cw.stride(0)
        if residual is not None:
            _gpu(residual, gate)
            _check(residual.dtype == torch.float32 and residual.shape == (M, N) and residual.stride(1) == 1, "gemm_nt: residual must be (M, N) float32 rows")
            P.epilogue = _lib.GEMM_EPI_F32_GATE_RESIDUAL
            X.residual_ptr, X.residual_ld = _ptr(residual), residual.stride(0)
            if gate is not None:
                _check(rows_per_batch and rows_per_batch % 256 == 0 and M % rows_per_batch == 0 and gate.dtype == torch.float32
                       and gate.shape == (M // rows_per_batch, N) and gate.stride(1) == 1, "gemm_nt: gate must be (M / rows_per_batch, N) float32, rows_per_batch % 256 == 0")
                X.gate_ptr, X.gate_ld, X.rows_per_batch = _ptr(gate), gate.stride(0), rows_per_batch
    elif epilogue == "gated_split3":
        P.epilogue = _lib.GEMM_EPI_GATED_GELU_SPLIT3
        pieces = 2 if pair_out else 3
        X.c_image_pieces = pieces
        if out is None:
            out = torch.empty((M, pieces * (N // 2)), device=a.device, dtype=torch.bfloat16)
        _check(out.dtype == torch.bfloat16 and out.shape == (M, pieces * (N // 2)) and out.stride(1) == 1, "gemm_nt: out must be (M, 3F) / (M, 2F) bfloat16 rows")
    elif epilogue == "f16_qkv":
        # the qkv Linear of the attention fusion as scaled fp16 (include/dimsum_hip.h, DIMSUM_GEMM_EPI_F16_QKV): q_cols = the width of q
        P.epilogue = _lib.GEMM_EPI_F16_QKV
        _check(scales is not None and gate_bound is not None and rows_per_batch and rows_per_batch % 256 == 0 and M % rows_per_batch == 0
               and q_cols and q_cols % 16 == 0 and N % 8 == 0, "gemm_nt: f16_qkv needs scales, gate_bound = {wl1, bmax}, rows_per_batch % 256 == 0, q_cols % 16 == 0")
        X.rows_per_batch, X.qkv_q_cols = rows_per_batch, q_cols
        if out is None:
            out = torch.empty((M, N), device=a.device, dtype=torch.float16)
        _check(out.dtype == torch.float16 and out.shape == (M, N) and out.stride(1) == 1, "gemm_nt: out must be (M, N) float16 rows")
    elif epilogue == "gated_f16":
        P.epilogue = _lib.GEMM_EPI_GATED_GELU_F16
        if out is None:
            out = torch.empty((M, N // 2), device=a.device, dtype=torch.float16)
        _check(out.dtype == torch.float16 and out.shape == (M, N // 2) and out.stride(1) == 1, "gemm_nt: out must be (M, F) float16 rows")
    else:
        raise ValueError(f"gemm_nt: unknown epilogue {epilogue!r}")
    P.ldc = out.stride(0)
    h_inv = None
    if scales is not None:
        sa, sb = scales
        _gpu(sa, sb)
        _check(sa.dtype == torch.float32 and sb.dtype == torch.float32 and sa.is_contiguous() and sb.is_contiguous() and sa.numel() == M
               and sb.numel() == N, "gemm_nt: scales must be contiguous float32 (M,) and (N,)")
        P.a_inv_scale_ptr, P.b_inv_scale_ptr = _ptr(sa), _ptr(sb)
        if gate_bound is not None:
            _gpu(gate_bound)
            _check(epilogue in ("gated_f16", "f16_qkv") and gate_bound.dtype == torch.float32 and gate_bound.numel() == 2 and gate_bound.is_contiguous(),
                   "gemm_nt: gate_bound is a 2-element float32 tensor for the gated_f16 / f16_qkv epilogues")
            X.gate_bound_ptr = _ptr(gate_bound)
            if epilogue == "gated_f16":
                h_inv = torch.empty((M,), device=a.device, dtype=torch.float32)
                X.h_inv_scale_ptr = _ptr(h_inv)
    P.a_ptr, P.b_ptr, P.bias_ptr, P.c_ptr = _ptr(a), _ptr(b), _ptr(bias), _ptr(out)
    x12 = None
    if keep_x12:
        _check((epilogue == "gated_split3" and a.dtype == torch.bfloat16 and scales is None) or (epilogue == "gated_f16" and a.dtype == torch.float16 and h_inv is not None),
               "gemm_nt: keep_x12 goes with the gated_split3 epilogue over bf16 images or the gated_f16 epilogue over scaled-fp16 operands with gate_bound")
        x12 = torch.empty((M, N), device=a.device, dtype=torch.float32)
        X.x12_ptr, X.x12_ld = _ptr(x12), N
    if events is not None:
        X.timing_start_event, X.timing_stop_event = events
    if tune is not None:
        X.tune_variant, X.tune_group_m = tune[:2]
        X.tune_reserved = tune[2] if len(tune) > 2 else 0
    elif epilogue == "gated_f16" and os.environ.get("DIMSUM_GEMM_PERSIST", "1") == "0":
        X.tune_variant = 513          # A / B switch: the gated GEMM one workgroup per tile instead of the persistent stream (csrc/gemm_nt_kernel.hpp, kVarPersist)
    with torch.cuda.device(a.device):
        if _gemm_kernel_log is not None:            # tests / bench: which kernel family the library picks for this call
            _gemm_kernel_log.append((epilogue if conv is None else "f32_conv", int(_lib.load().dimsum_gemm_nt_kernel_for(P))))
        _lib.check(_lib.load().dimsum_gemm_nt(P, _stream(a)), "gemm_nt")
    if x12 is not None:
        if h_inv is not None:
            return F16Image(out, h_inv), x12
        return (PairImage(out) if pair_out else out), x12
    if pair_out:
        return PairImage(out)
    return out if h_inv is None else F16Image(out, h_inv)


def gated_gelu_bwd(x12, bias, dh, need_dbias=True, split3=False):
    """-> (dx12, dbias or None).  split3: dx12 as a split-bf16 operand image in weight order, (..., 3 * 2H) bfloat16 [hi | lo | hi];
    split3="f16s": as a scaled-fp16 image (F16Image: (..., 2H) float16 + one inverse scale per row, the exact row maximum; H <= 5120)"""
    _gpu(x12, bias, dh)
    dh = dh.contiguous()
    H = x12.shape[-1] // 2
    if split3 == "f16s":
        _check(x12.is_contiguous() and x12.dtype == torch.float32 and dh.dtype == torch.float32 and H % 4 == 0 and H <= 5120, "gated_gelu_bwd: the f16s image needs contiguous float32 rows, H % 4 == 0, H <= 5120")
        rows = x12.numel() // (2 * H)
        img = torch.empty(x12.shape, device=x12.device, dtype=torch.float16)
        inv = torch.empty(x12.shape[:-1], device=x12.device, dtype=torch.float32)
        dbias = _zeros(2 * H, x12.device) if (bias is not None and need_dbias) else None
        with torch.cuda.device(x12.device):
            _lib.check(_lib.load().dimsum_gated_gelu_bwd_f16s(_ptr(x12), _ptr(bias), _ptr(dh), _ptr(img), _ptr(inv), _ptr(dbias), rows, H, _stream(x12)), "gated_gelu_bwd")
        return F16Image(img, inv), dbias
    pair = split3 == "pair"        # (..., 2 * 2H) bfloat16 [hi | lo] (PairImage)
    dx12 = torch.empty(x12.shape[:-1] + ((4 if pair else 6) * H,), device=x12.device, dtype=torch.bfloat16) if split3 else torch.empty_like(x12)
    dbias = _zeros(2 * H, x12.device) if (bias is not None and need_dbias) else None
    rows = x12.numel() // (2 * H)
    with torch.cuda.device(x12.device):
        lib = _lib.load()
        fn = lib.dimsum_gated_gelu_bwd_pair if pair else (lib.dimsum_gated_gelu_bwd_split3 if split3 else lib.dimsum_gated_gelu_bwd)
        _lib.check(fn(_ptr(x12), _ptr(bias), _ptr(dh), _ptr(dx12), _ptr(dbias), rows, H, _stream(x12)), "gated_gelu_bwd")
    return (PairImage(dx12) if pair else dx12), dbias


# ---------------------------------------------------------------------------------------------------------------------
# cross-attention fusion core (csrc/xattn_fusion.hip)
# ---------------------------------------------------------------------------------------------------------------------
def xattn_supported(qkv, head_dim):
    """the MFMA kernel is instantiated for the head sizes of the DiM zoo: 24 (S/2), 48 (B/2), 64 (L/*), 72 (XL/2), and 32"""
    return qkv.is_cuda and qkv.dtype == torch.float32 and qkv.stride(-1) == 1 and head_dim in (24, 32, 48, 64, 72)


def xattn_fusion_fwd(qkv1, qkv2, heads, need_lse=False, bias1=None, bias2=None, split_bf16=None, split3=False, f16s=None):
    """qkv*: (B, L, 3*heads*hd) -> (B, L, 2*heads*hd) = cat(softmax(q1 k2^T/sqrt(hd)) v2, softmax(q2 k1^T/sqrt(hd)) v1).
    qkv2 = None: plain self-attention softmax(q1 k1^T/sqrt(hd)) v1 -> (B, L, heads*hd).
    bias1/bias2 (3*heads*hd): the qkv Linear biases when qkv* are bias-free GEMM outputs (added inside the kernel).
    split_bf16: None follows torch.backends.cuda.matmul.allow_tf32 (the reference's GEMM policy, train.py:20-21: the
    QK^T / PV contractions then run as split-bf16 MFMA like the library GEMMs do); False = exact fp32 MFMA.
    split3 (split-bf16 kernel only): the result as the split-bf16 left operand image (B, L, 3 * width) bfloat16 of the proj Linear.
    f16s = (x1_inv, x2_inv or None, kv_bound): the single-product fp16 kernel (csrc/xattn_fusion_f16.hip, the "f16s" GEMM policy):
    x*_inv (B, L) are the inverse row scales of the scaled-fp16 images the qkv GEMMs consumed, kv_bound a 4-element float32 tensor
    {max_n sum_c |W1_nc|, max|bias1|, the same for qkv2}; with split3="f16s" the result is the scaled-fp16 image (F16Image) of proj."""
    self_attn = qkv2 is None
    if split_bf16 is None:
        split_bf16 = bool(torch.backends.cuda.matmul.allow_tf32)
    _gpu(qkv1, qkv2, bias1, bias2)
    _check(self_attn or (bias1 is None) == (bias2 is None), "xattn_fusion: pass both biases or none")
    for bb in (bias1, bias2):
        if bb is not None:
            _check(bb.dtype == torch.float32 and bb.numel() == qkv1.shape[2] and bb.is_contiguous(), "xattn_fusion: bad bias")
    B, L, W = qkv1.shape
    hd = W // (3 * heads)
    qkv_f16 = qkv1.dtype == torch.float16          # the F16_QKV epilogue's output (gemm_nt(epilogue="f16_qkv")): the fp16 kernel's own scaling
    _check((qkv1.dtype == torch.float32 or (qkv_f16 and f16s is not None and bias1 is None and bias2 is None)) and qkv1.stride(2) == 1, "xattn_fusion: bad qkv")
    if not self_attn:
        _check(qkv1.shape == qkv2.shape and qkv2.dtype == qkv1.dtype and qkv1.stride() == qkv2.stride(), "xattn_fusion: qkv1/qkv2 must share shape, dtype and strides")
    nd = 1 if self_attn else 2
    out_inv = None
    if f16s is not None:
        x1_inv, x2_inv, kv_bound = f16s
        _gpu(x1_inv, x2_inv, kv_bound)
        _check(x1_inv.dtype == torch.float32 and x1_inv.numel() == B * L and x1_inv.is_contiguous()
               and (self_attn or (x2_inv is not None and x2_inv.dtype == torch.float32 and x2_inv.numel() == B * L and x2_inv.is_contiguous()))
               and kv_bound.dtype == torch.float32 and kv_bound.numel() == 4 and kv_bound.is_contiguous(), "xattn_fusion: bad f16s arguments")
        _check(split3 in (False, "f16s"), "xattn_fusion: the fp16 kernel writes fp32 or the scaled-fp16 image")
    if split3 == "f16s":
        _check(f16s is not None and (heads * hd) % 8 == 0, "xattn_fusion: the scaled-fp16 image needs the fp16 kernel")
        out = torch.empty((B, L, nd * heads * hd), device=qkv1.device, dtype=torch.float16)
        out_inv = torch.empty((B, L), device=qkv1.device, dtype=torch.float32)
    elif split3:
        _check(split_bf16, "xattn_fusion: the operand image output exists for the split-bf16 kernel only")
        out = torch.empty((B, L, (2 if split3 == "pair" else 3) * nd * heads * hd), device=qkv1.device, dtype=torch.bfloat16)
    else:
        out = torch.empty((B, L, nd * heads * hd), device=qkv1.device, dtype=torch.float32)
    lse = torch.empty((B, nd, heads, L), device=qkv1.device, dtype=torch.float32) if need_lse else None
    if B > 0:
        P = _lib.XattnParams()
        P.batch, P.seqlen, P.heads, P.head_dim, P.scale, P.n_dirs = B, L, heads, hd, hd ** -0.5, nd
        P.qkv_batch_stride, P.qkv_token_stride = qkv1.stride(0), qkv1.stride(1)
        P.out_batch_stride, P.out_token_stride = out.stride(0), out.stride(1)
        P.qkv1_ptr, P.qkv2_ptr, P.out_ptr, P.lse_ptr = _ptr(qkv1), _ptr(qkv2), _ptr(out), _ptr(lse)
        P.bias1_ptr, P.bias2_ptr = _ptr(bias1), _ptr(bias2)
        P.precision = 1 if split_bf16 else 0
        P.out_split3 = _SPLIT_MODE.get(split3, int(bool(split3)))
        if f16s is not None:
            P.precision = 2
            P.qkv_f16 = int(qkv_f16)
            P.x1_inv_ptr, P.x2_inv_ptr, P.kv_bound_ptr, P.out_inv_ptr = _ptr(x1_inv), _ptr(x2_inv), _ptr(kv_bound), _ptr(out_inv)
        with torch.cuda.device(qkv1.device):
            _lib.check(_lib.load().dimsum_xattn_fusion_fwd(P, _stream(qkv1)), "xattn_fusion_fwd")
    if out_inv is not None:
        out = F16Image(out, out_inv)
    elif split3 == "pair":
        out = PairImage(out)
    return (out, lse) if need_lse else out


def xattn_fusion_bwd(qkv1, qkv2, out, lse, dout, heads, bias1=None, bias2=None, split_bf16=None, f16=False):
    """backward of xattn_fusion_fwd -> (dqkv1, dqkv2), each (B, L, 3*heads*hd); `out`, `lse` are the forward's results.
    qkv2 = None (self-attention): -> (dqkv1, None).
    f16: ONE fp16 MFMA product per element (precision 2: the TF32-equivalent carrier of the "f16s" policy; dout rows scaled by exact powers of
    two inside the kernels, csrc/xattn_fusion_bwd.hip) instead of the three split-bf16 products."""
    self_attn = qkv2 is None
    _gpu(qkv1, qkv2, out, lse, dout, bias1, bias2)
    B, L, W = qkv1.shape
    hd = W // (3 * heads)
    nd = 1 if self_attn else 2
    _check(qkv1.dtype == torch.float32 and qkv1.stride(2) == 1, "xattn_fusion_bwd: bad qkv")
    if not self_attn:
        _check(qkv1.shape == qkv2.shape and qkv1.stride() == qkv2.stride(), "xattn_fusion_bwd: bad qkv2")
    _check(tuple(out.shape) == (B, L, nd * heads * hd) and out.stride(2) == 1 and tuple(lse.shape) == (B, nd, heads, L) and lse.is_contiguous(), "xattn_fusion_bwd: bad out / lse")
    if dout.stride() != out.stride():
        dout, out = dout.contiguous(), out.contiguous()
    _check(self_attn or (bias1 is None) == (bias2 is None), "xattn_fusion_bwd: pass both biases or none")
    dqkv1 = torch.empty((B, L, W), device=qkv1.device, dtype=torch.float32)
    dqkv2 = None if self_attn else torch.empty((B, L, W), device=qkv1.device, dtype=torch.float32)
    delta = torch.empty((2 if f16 else 1, B, nd, heads, L), device=qkv1.device, dtype=torch.float32)      # D rows (+ the dout row scales, f16)
    if B > 0:
        Q = _lib.XattnBwdParams()
        P = Q.fwd
        P.batch, P.seqlen, P.heads, P.head_dim, P.scale, P.n_dirs = B, L, heads, hd, hd ** -0.5, nd
        P.qkv_batch_stride, P.qkv_token_stride = qkv1.stride(0), qkv1.stride(1)
        P.out_batch_stride, P.out_token_stride = out.stride(0), out.stride(1)
        P.qkv1_ptr, P.qkv2_ptr, P.out_ptr, P.lse_ptr = _ptr(qkv1), _ptr(qkv2), _ptr(out), _ptr(lse)
        P.bias1_ptr, P.bias2_ptr = _ptr(bias1), _ptr(bias2)
        if split_bf16 is None:                    # same policy switch as the forward and the library GEMMs
            split_bf16 = bool(torch.backends.cuda.matmul.allow_tf32)
        P.precision = 2 if f16 else (1 if split_bf16 else 0)
        Q.dqkv_batch_stride, Q.dqkv_token_stride = dqkv1.stride(0), dqkv1.stride(1)
        Q.dout_ptr, Q.dqkv1_ptr, Q.dqkv2_ptr, Q.delta_ptr = _ptr(dout), _ptr(dqkv1), _ptr(dqkv2), _ptr(delta)
        with torch.cuda.device(qkv1.device):
            _lib.check(_lib.load().dimsum_xattn_fusion_bwd(Q, _stream(qkv1)), "xattn_fusion_bwd")
    return dqkv1, dqkv2
