"""Token scan orders of the DiM blocks as integer gather tables (`out[j] = in[table[j]]`), built once on the host
with vectorised numpy closed forms. Same public names as dimsum/scanning_orders.py (sweep_path, zigma_path,
jpeg_zigzag, reverse_permut_np, SCAN_ZOO, local_scan, local_reverse); bit-exact against the reference tables
(tests/golden/perm_tables.npz, sha256 pins of SURVEY.md section 8 a1).

The reference materialises every reorder as a chain of einops/flip/view copies (models_dim.py:1496-1524, 656-705);
here each block composes its whole chain into ONE table at construction and the runtime cost is a single gather
(fused into the token-transform kernel, csrc/token_transform.hip).
"""
import numpy as np
import torch

_CORNERS = ((0, 0), (0, 1), (1, 0), (1, 1))      # (bottom?, right?) -> start corner, scanning_orders.py:29-34


def _oriented(v, h, N, bottom, right):
    """index of the cell reached after v vertical / h horizontal moves from the chosen start corner"""
    r = (N - 1 - v) if bottom else v
    c = (N - 1 - h) if right else h
    return r * N + c


def _paths(N, walk_rowmajor, walk_colmajor):
    out = []
    for bottom, right in _CORNERS:
        for vh in (walk_rowmajor, walk_colmajor):
            v, h = vh
            out.append(_oriented(v, h, N, bottom, right).astype(np.int64))
    return out


def sweep_path(N):
    """raster scans (scanning_orders.py:7-40): row-major then column-major, from each corner."""
    a, b = np.divmod(np.arange(N * N), N)
    return _paths(N, (a, b), (b, a))


def zigma_path(N):
    """boustrophedon scans (scanning_orders.py:43-78)."""
    a, b = np.divmod(np.arange(N * N), N)
    snake = np.where(a % 2 == 0, b, N - 1 - b)
    return _paths(N, (a, snake), (snake, a))


def _jpeg_vh(N, first_right):
    s = np.concatenate([np.full(min(d, 2 * N - 2 - d) + 1, d) for d in range(2 * N - 1)])       # anti-diagonal of each step
    start = np.concatenate([[0], np.cumsum([min(d, 2 * N - 2 - d) + 1 for d in range(2 * N - 1)])[:-1]])
    k = np.arange(N * N) - start[s]                                                             # position inside the diagonal
    lo = np.maximum(0, s - (N - 1))
    ln = np.minimum(s, 2 * N - 2 - s)
    down = (s % 2 == 1) if first_right else (s % 2 == 0)       # direction in which v grows along this diagonal
    v = np.where(down, lo + k, lo + ln - k)
    return v, s - v


def jpeg_zigzag(N):
    """JPEG zig-zag (scanning_orders.py:81-245): "lr" starts to the right, "tb" starts downwards."""
    return _paths(N, _jpeg_vh(N, True), _jpeg_vh(N, False))


def reverse_permut_np(permutation):
    permutation = np.asarray(permutation)
    rev = np.empty_like(permutation)
    rev[permutation] = np.arange(len(permutation))
    return rev


SCAN_ZOO = {"sweep": sweep_path, "zigma": zigma_path, "jpeg": jpeg_zigzag}


# ---- block-level reorders as tables ----------------------------------------------------------------------------------
def block_order_table(H, reverse=False, transpose=False, continuity=False):
    """order seen by the mixer in DiMBlockRaw / DCTBlock (models_dim.py:1496-1507, 884-893)."""
    ids = np.arange(H * H).reshape(H, H)
    if transpose:
        ids = ids.T
    ids = ids.copy()
    if continuity:
        ids[1::2] = ids[1::2, ::-1]
    ids = ids.reshape(-1)
    return (ids[::-1] if reverse else ids).copy()


def local_scan_table(H, w, column_first=False):
    """gather table of local_scan(x, w, H, H, column_first) for H % w == 0 (scanning_orders.py:347-367)."""
    g = H // w
    ids = np.arange(H * H).reshape(g, w, g, w)              # (hg, hi, wg, wi)
    ids = ids.transpose(2, 0, 3, 1) if column_first else ids.transpose(0, 2, 1, 3)
    return ids.reshape(-1).copy()


def compose(first, then):
    """table of `gather(gather(x, first), then)`"""
    return np.asarray(first)[np.asarray(then)]


def as_index(table, device=None):
    return torch.as_tensor(np.ascontiguousarray(table), dtype=torch.int64, device=device)


# ---- tensor versions with the reference's signatures (kept for API parity; square grids divisible by w) -------------
def local_scan(x, w=7, H=14, W=14, flip=False, column_first=False):
    assert H == W and H % w == 0, "only square grids divisible by the window are used by DiMSUM"
    out = x.index_select(1, as_index(local_scan_table(H, w, column_first), x.device))
    return out.flip([1]) if flip else out


def local_reverse(x, w=7, H=14, W=14, flip=False, column_first=False):
    assert H == W and H % w == 0
    if flip:
        x = x.flip([1])
    return x.index_select(1, as_index(reverse_permut_np(local_scan_table(H, w, column_first)), x.device))
