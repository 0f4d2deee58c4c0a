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
import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


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
def _fill_ssm(P, u, delta, A, B, C, D, z, delta_bias, delta_softplus, out, x, out_z):
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


def selective_scan_fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus, need_out=True, need_x=True):
    """-> [out, x, (out_z)]   exactly like selective_scan_cuda.fwd.
    `need_out=False` / `need_x=False` are inference extras: the corresponding store is skipped and None returned."""
    _check_ssm(u, delta, A, B, C, D, z, delta_bias)
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    n_chunks = (seqlen + 2047) // 2048
    out = torch.empty_like(delta) if need_out else None          # HBL layout like delta (selective_scan.cpp:310-311)
    x = torch.empty((batch, dim, n_chunks, dstate * 2), device=u.device, dtype=torch.float32) if need_x else None
    out_z = torch.empty_like(z) if z is not None else None
    if u.numel() > 0:
        P = _lib.SsmParams()
        _fill_ssm(P, u, delta, A, B, C, D, z, delta_bias, delta_softplus, out, x, out_z)
        with torch.cuda.device(u.device):
            _lib.check(_lib.load().dimsum_ssm_scan_fwd(P, _stream(u)), "selective_scan_fwd")
    res = [out, x]
    if z is not None:
        res.append(out_z)
    return res
