"""numpy restatements of the integer / small-matrix parts of the hot path. TEST INFRASTRUCTURE (oracle/__init__.py).

Deliberately written as plain loops that follow the reference step by step (the product code in dimsum_amd/ uses
closed forms and fused kernels); both are pinned to the same golden tables (tests/golden/perm_tables.npz, ...).
"""
import math

import numpy as np


# ----------------------------------------------------------------------------------------------------------------
# scan-order tables  (dimsum/scanning_orders.py:7-253)
# ----------------------------------------------------------------------------------------------------------------
_CORNERS = lambda N: [(0, 0, 1, 1), (0, N - 1, 1, -1), (N - 1, 0, -1, 1), (N - 1, N - 1, -1, -1)]  # :29-34


def _emit(N, sr, sc, dr, dc, v, h):
    return (sr + dr * v) * N + sc + dc * h


def sweep_paths(N):
    """scanning_orders.py:7-40: per corner, row-major then column-major raster."""
    out = []
    for sr, sc, dr, dc in _CORNERS(N):
        out.append(np.array([_emit(N, sr, sc, dr, dc, i, j) for i in range(N) for j in range(N)]))
        out.append(np.array([_emit(N, sr, sc, dr, dc, i, j) for j in range(N) for i in range(N)]))
    return out


def zigma_paths(N):
    """scanning_orders.py:43-78: boustrophedon (odd rows / columns walked backwards)."""
    out = []
    for sr, sc, dr, dc in _CORNERS(N):
        out.append(np.array([_emit(N, sr, sc, dr, dc, i, j if i % 2 == 0 else N - 1 - j)
                             for i in range(N) for j in range(N)]))
        out.append(np.array([_emit(N, sr, sc, dr, dc, i if j % 2 == 0 else N - 1 - i, j)
                             for j in range(N) for i in range(N)]))
    return out


def _jpeg_walk(N, first_right):
    """JPEG zig-zag over an NxN grid as (v, h) pairs: anti-diagonals s = v+h, alternating direction.
    `first_right`: the 2nd visited cell is (0,1) (scanning_orders.py:81-160, "lr") else (1,0) (:162-231, "tb")."""
    cells = []
    for s in range(2 * N - 1):
        lo, hi = max(0, s - (N - 1)), min(s, N - 1)
        diag = [(v, s - v) for v in range(lo, hi + 1)]          # v ascending = moving down-left
        up = (s % 2 == 0)                                       # lr: even diagonals are walked upwards (v descending)
        if first_right:
            cells += diag[::-1] if up else diag
        else:
            cells += diag if up else diag[::-1]
    return cells


def jpeg_paths(N):
    out = []
    for sr, sc, dr, dc in _CORNERS(N):
        for first_right in (True, False):
            out.append(np.array([_emit(N, sr, sc, dr, dc, v, h) for v, h in _jpeg_walk(N, first_right)]))
    return out


SCAN_ZOO = {"sweep": sweep_paths, "zigma": zigma_paths, "jpeg": jpeg_paths}


def inverse_permutation(p):
    """reverse_permut_np, scanning_orders.py:248-253."""
    r = np.zeros(len(p), dtype=np.int64)
    for i, v in enumerate(p):
        r[v] = i
    return r


def local_scan_index(H, w, column_first):
    """Gather table of local_scan (scanning_orders.py:347-367) for H == W divisible by w: out[j] = in[idx[j]]."""
    Hg = H // w
    idx = []
    if column_first:      # view(B,Hg,w,Wg,w,C).permute(0,3,1,4,2,5): order (wg, hg, wi, hi)
        for wg in range(Hg):
            for hg in range(Hg):
                for wi in range(w):
                    for hi in range(w):
                        idx.append((hg * w + hi) * H + wg * w + wi)
    else:                 # permute(0,1,3,2,4,5): order (hg, wg, hi, wi)
        for hg in range(Hg):
            for wg in range(Hg):
                for hi in range(w):
                    for wi in range(w):
                        idx.append((hg * w + hi) * H + wg * w + wi)
    return np.array(idx)


def block_order_index(H, reverse, transpose, continuity):
    """Token order the mixer sees in DiMBlockRaw.forward (models_dim.py:1496-1507): out[j] = in[idx[j]]."""
    ids = np.arange(H * H).reshape(H, H)
    if transpose:                       # "n (h w) c -> n (w h) c"
        ids = ids.T
    ids = ids.copy()
    if continuity:                      # rows 1::2 of the (w h) grid flipped
        ids[1::2] = ids[1::2, ::-1]
    ids = ids.reshape(-1)
    if reverse:
        ids = ids[::-1]
    return ids.copy()


# ----------------------------------------------------------------------------------------------------------------
# 2-level Haar on the token grid  (models_dim.py:572-604 over wavelet_layer.py:7-115)
# ----------------------------------------------------------------------------------------------------------------
def _dwt_level(x):
    """x: (B, C, H, W) -> (B, 4C, H/2, W/2), bands concatenated [ll | lh | hl | hh] (wavelet_layer.py:8-22).
    Filters: dec_lo[::-1] = [s, s], dec_hi[::-1] = [s, -s]; w_lh[i][j] = hi[i]*lo[j] (rows get the high-pass)."""
    a, b = x[:, :, 0::2, 0::2], x[:, :, 0::2, 1::2]
    c, d = x[:, :, 1::2, 0::2], x[:, :, 1::2, 1::2]
    ll, lh = (a + b + c + d) * 0.5, (a + b - c - d) * 0.5
    hl, hh = (a - b + c - d) * 0.5, (a - b - c + d) * 0.5
    return np.concatenate([ll, lh, hl, hh], axis=1)


def _idwt_level(x):
    """x: (B, 4C, H, W) bands [ll|lh|hl|hh] -> (B, C, 2H, 2W)  (wavelet_layer.py:41-54; rec_hi = [s, -s])."""
    B, C4, H, W = x.shape
    C = C4 // 4
    ll, lh, hl, hh = x[:, :C], x[:, C:2 * C], x[:, 2 * C:3 * C], x[:, 3 * C:]
    out = np.zeros((B, C, 2 * H, 2 * W), x.dtype)
    out[:, :, 0::2, 0::2] = (ll + lh + hl + hh) * 0.5
    out[:, :, 0::2, 1::2] = (ll + lh - hl - hh) * 0.5
    out[:, :, 1::2, 0::2] = (ll - lh + hl - hh) * 0.5
    out[:, :, 1::2, 1::2] = (ll - lh - hl + hh) * 0.5
    return out


_SHUFFLE = [i % 4 * 4 + i // 4 for i in range(16)]  # models_dim.py:580-583


def haar_dwt_tokens(x):
    """_dwt_fast (models_dim.py:572-586), num_wavelet_lv = 2.  x: (B, L, C) -> (B, L, C)."""
    B, L, C = x.shape
    H = int(math.isqrt(L))
    img = x.transpose(0, 2, 1).reshape(B, C, H, H)
    sub = _dwt_level(_dwt_level(img)) / 4.0                      # (B, 16C, H/4, H/4)
    chunks = np.split(sub, 16, axis=1)
    out = np.concatenate([chunks[i] for i in _SHUFFLE], axis=1)  # (B, 16C, h, w)
    h = H // 4
    # "b (c p1 p2) h w -> b (h p1 w p2) c"
    out = out.reshape(B, C, 4, 4, h, h).transpose(0, 4, 2, 5, 3, 1).reshape(B, L, C)
    return np.ascontiguousarray(out)


def haar_idwt_tokens(x):
    """_idwt_fast (models_dim.py:588-604)."""
    B, L, C = x.shape
    H = int(math.isqrt(L))
    h = H // 4
    # "b (h p1 w p2) c -> b (c p1 p2) h w"
    sub = (x * 4.0).reshape(B, h, 4, h, 4, C).transpose(0, 5, 2, 4, 1, 3).reshape(B, 16 * C, h, h)
    chunks = np.split(sub, 16, axis=1)
    sub = np.concatenate([chunks[i] for i in _SHUFFLE], axis=1)
    img = _idwt_level(_idwt_level(sub))                          # (B, C, H, H)
    return np.ascontiguousarray(img.reshape(B, C, L).transpose(0, 2, 1))


# ----------------------------------------------------------------------------------------------------------------
# 4x4 block DCT-II on the token grid  (dct_layer.py:6-84, models_dim.py:876-882,919-928)
# ----------------------------------------------------------------------------------------------------------------
def dct_basis(k=4):
    """basis[v*k+u, y, x] = (2 C_v C_u / k) cos((2y+1) v pi / 2k) cos((2x+1) u pi / 2k)   (dct_layer.py:21-29)."""
    Cn = np.ones(k)
    Cn[0] = 1 / np.sqrt(2)
    bas = np.zeros((k * k, k, k))
    for v in range(k):
        for u in range(k):
            for y in range(k):
                for x in range(k):
                    bas[v * k + u, y, x] = (2 * Cn[v] * Cn[u] / k) * np.cos((2 * y + 1) * v * np.pi / (2 * k)) \
                        * np.cos((2 * x + 1) * u * np.pi / (2 * k))
    return bas.astype(np.float32)


def dct_tokens(x):
    """x: (B, L, C) -> coefficients laid out on the token grid: token (4h+v, 4w+u) = coef (v,u) of block (h,w)."""
    B, L, C = x.shape
    H = int(math.isqrt(L))
    h = H // 4
    bas = dct_basis().reshape(16, 16).astype(np.float64)
    blk = x.reshape(B, h, 4, h, 4, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, h, h, C, 16)   # (.., y*4+x)
    co = np.einsum("bhwcp,kp->bhwck", blk.astype(np.float64), bas)                          # k = v*4+u
    out = co.reshape(B, h, h, C, 4, 4).transpose(0, 1, 4, 2, 5, 3).reshape(B, L, C)
    return out.astype(np.float32)


def idct_tokens(x):
    B, L, C = x.shape
    H = int(math.isqrt(L))
    h = H // 4
    bas = dct_basis().reshape(16, 16).astype(np.float64)
    co = x.reshape(B, h, 4, h, 4, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, h, h, C, 16)     # k = v*4+u
    px = np.einsum("bhwck,kp->bhwcp", co.astype(np.float64), bas)                           # p = y*4+x (dct_layer.py:62-71)
    out = px.reshape(B, h, h, C, 4, 4).transpose(0, 1, 4, 2, 5, 3).reshape(B, L, C)
    return out.astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------
# attention fusion core, gated GeLU, embeddings
# ----------------------------------------------------------------------------------------------------------------
def xattn_fusion_core(qkv1, qkv2, heads):
    """attention_fusion.py:64-79 (swap_k=False) after the qkv Linears and before proj.
    qkv*: (B, L, 3*heads*hd) -> (B, L, 2*heads*hd) = cat(x12, x21)."""
    B, L, W = qkv1.shape
    hd = W // (3 * heads)

    def split(qkv):
        t = qkv.astype(np.float64).reshape(B, L, 3, heads, hd).transpose(2, 0, 3, 1, 4)
        return t[0], t[1], t[2]

    def attn(q, k, v):
        s = np.einsum("bhid,bhjd->bhij", q, k) * hd ** -0.5
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(-1, keepdims=True)
        return np.einsum("bhij,bhjd->bhid", p, v).transpose(0, 2, 1, 3).reshape(B, L, heads * hd)

    q1, k1, v1 = split(qkv1)
    q2, k2, v2 = split(qkv2)
    return np.concatenate([attn(q1, k2, v2), attn(q2, k1, v1)], axis=-1).astype(np.float32)


def gated_gelu(x12):
    """mlp.py:66-70 with nn.GELU(approximate='tanh')."""
    x = x12.astype(np.float64)
    H = x.shape[-1] // 2
    a, b = x[..., :H], x[..., H:]
    g = 0.5 * a * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (a + 0.044715 * a ** 3)))
    return (g * b).astype(np.float32)
