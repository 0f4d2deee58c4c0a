"""ctypes binding of oracle/ssm_oracle.c on numpy float32 arrays. TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import ctypes as C

import numpy as np

from . import build as _build

_lib = None
_f = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_f)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def selective_scan_fwd(u, delta, A, B, Cm, D=None, z=None, delta_bias=None, delta_softplus=False):
    """Returns (y, out_z or None, x). B, C: (batch, G, N, L) or (batch, N, L).
    Math: selective_scan_ref, mamba/mamba_ssm/ops/selective_scan_interface.py:104-171."""
    u, delta, A, D, z, delta_bias = _c(u), _c(delta), _c(A), _c(D), _c(z), _c(delta_bias)
    B = _c(B if B.ndim == 4 else B[:, None])
    Cm = _c(Cm if Cm.ndim == 4 else Cm[:, None])
    Bsz, Dm, L = u.shape
    N, G = A.shape[1], B.shape[1]
    y = np.empty_like(u)
    oz = np.empty_like(u) if z is not None else None
    x = np.empty((Bsz, Dm, (L + 2047) // 2048, 2 * N), np.float32)
    rc = lib().oracle_selective_scan_fwd(_p(u), _p(delta), _p(A), _p(B), _p(Cm), _p(D), _p(z), _p(delta_bias),
                                         int(delta_softplus), _p(y), _p(oz), _p(x), Bsz, Dm, L, N, G)
    assert rc == 0, rc
    return y, oz, x


def selective_scan_bwd(u, delta, A, B, Cm, D, z, delta_bias, delta_softplus, dout):
    """Returns dict(du, ddelta, dA, dB, dC, dD, dz, ddelta_bias). dout = grad of the final output."""
    u, delta, A, D, z, delta_bias, dout = _c(u), _c(delta), _c(A), _c(D), _c(z), _c(delta_bias), _c(dout)
    squeeze = B.ndim == 3
    B = _c(B if B.ndim == 4 else B[:, None])
    Cm = _c(Cm if Cm.ndim == 4 else Cm[:, None])
    Bsz, Dm, L = u.shape
    N, G = A.shape[1], B.shape[1]
    r = dict(du=np.empty_like(u), ddelta=np.empty_like(u), dA=np.empty_like(A), dB=np.empty_like(B),
             dC=np.empty_like(Cm), dD=np.empty(Dm, np.float32), dz=np.empty_like(u) if z is not None else None,
             ddelta_bias=np.empty(Dm, np.float32))
    rc = lib().oracle_selective_scan_bwd(_p(u), _p(delta), _p(A), _p(B), _p(Cm), _p(D), _p(z), _p(delta_bias),
                                         int(delta_softplus), _p(dout), _p(r["du"]), _p(r["ddelta"]), _p(r["dA"]),
                                         _p(r["dB"]), _p(r["dC"]), _p(r["dD"]), _p(r["dz"]), _p(r["ddelta_bias"]),
                                         Bsz, Dm, L, N, G)
    assert rc == 0, rc
    if squeeze:
        r["dB"], r["dC"] = r["dB"][:, 0], r["dC"][:, 0]
    return r


def causal_conv1d_fwd(x, weight, bias=None, silu=False):
    """causal_conv1d_ref, causal-conv1d/causal_conv1d/causal_conv1d_interface.py:48-64. x may be a strided view."""
    x = np.asarray(x, np.float32)
    if x.strides[-1] != 4:
        x = np.ascontiguousarray(x)
    weight, bias = _c(weight), _c(bias)
    Bsz, Dm, L = x.shape
    out = np.empty((Bsz, Dm, L), np.float32)
    rc = lib().oracle_causal_conv1d_fwd(_p(x), C.c_long(x.strides[0] // 4), C.c_long(x.strides[1] // 4), _p(weight),
                                        _p(bias), int(silu), _p(out), Bsz, Dm, L, weight.shape[1])
    assert rc == 0, rc
    return out


def causal_conv1d_bwd(x, weight, bias, dout, silu=False):
    x, weight, bias, dout = _c(x), _c(weight), _c(bias), _c(dout)
    Bsz, Dm, L = x.shape
    dx, dw, db = np.empty_like(x), np.empty_like(weight), np.empty(Dm, np.float32)
    rc = lib().oracle_causal_conv1d_bwd(_p(x), C.c_long(Dm * L), C.c_long(L), _p(weight), _p(bias), int(silu),
                                        _p(dout), _p(dx), _p(dw), _p(db), Bsz, Dm, L, weight.shape[1])
    assert rc == 0, rc
    return dx, dw, db


def norm_fwd(x, weight, bias=None, residual=None, eps=1e-5, is_rms=True):
    """rms_norm_ref / layer_norm_ref (upcast), mamba/mamba_ssm/ops/triton/layernorm.py:19-45.
    Returns (y, res_out, mean, rstd) on 2-D (M, N)."""
    x, weight, bias, residual = _c(x), _c(weight), _c(bias), _c(residual)
    M, N = x.shape
    y, ro = np.empty_like(x), np.empty_like(x)
    rstd, mean = np.empty(M, np.float32), np.zeros(M, np.float32)
    rc = lib().oracle_norm_fwd(_p(x), _p(residual), _p(weight), _p(bias), C.c_double(eps), int(is_rms), _p(y), _p(ro),
                               _p(rstd), _p(mean), M, N)
    assert rc == 0, rc
    return y, ro, mean, rstd


def norm_bwd(r, weight, dy, dres_out=None, eps=1e-5, is_rms=True):
    """Returns (dr, dweight, dbias); dr is the gradient wrt x and wrt residual alike."""
    r, weight, dy, dres_out = _c(r), _c(weight), _c(dy), _c(dres_out)
    M, N = r.shape
    dr, dw, db = np.empty_like(r), np.empty(N, np.float32), np.empty(N, np.float32)
    rc = lib().oracle_norm_bwd(_p(r), _p(weight), C.c_double(eps), int(is_rms), _p(dy), _p(dres_out), _p(dr), _p(dw),
                               _p(db), M, N)
    assert rc == 0, rc
    return dr, dw, db
