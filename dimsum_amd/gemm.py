"""GEMM precision policy and dispatch of the host layer (the large bias-free Linears of the denoiser: in_proj, out_proj, qkv,
proj, w12, w3 -- everything else is small). Since round 4 the operand-image products below run on this package's own MFMA kernel
(csrc/gemm_nt_kernel.hpp, native.gemm_nt: `_nt`, `_nt_f16s`, `gated_mlp_hidden_split3`; DIMSUM_GEMM_NT=0 or a shape outside its tiling
hands them to the library), with the gated GeLU and the block's residual tail as epilogues.

  "default"  whatever torch is set to: torch.backends.cuda.matmul.allow_tf32 = True (the reference's own setting,
             dimsum/train.py:20-21) makes hipBLASLt run its split-bf16 path on gfx950 (3 bf16 products per fp32 product,
             ~4e-6 rms relative error, ~380-420 TFLOP/s-equivalent); False = exact fp32 MFMA (~144 TFLOP/s).
  "fp16"     OPT-IN, inference only. TF32 keeps 10 mantissa bits of each operand and accumulates in fp32; gfx950 has no
             TF32 MFMA, but an fp16-input / fp32-accumulate GEMM has exactly that mantissa (10 bits) and ONE product per
             element instead of three. What it does NOT have is TF32's 8-bit exponent: operands above 65504 overflow and
             operands below 6e-5 lose bits (below 6e-8 they vanish). The DiM activations that feed these GEMMs are
             RMS-normalised / gated O(1..1e2) values and the weights O(1e-2), so the model's outputs stay within the
             TF32-class error (~5e-4 relative; tests/test_model_gpu.py::test_fp16_product_gemm_policy) -- but this is a
             weaker guarantee than the default and is therefore never the default nor bench.py's headline.
  "f16s"     the TF32-equivalent single product (inference): the large Linears take SCALED fp16 operand images
             (native.F16Image: fp16(row * 2^s) + the exact 2^-s per row, written by the producer kernels with exact row maxima or
             bound-derived scales), ONE fp16 MFMA product per element, fp32 accumulation, the scales undone in the GEMM epilogue;
             the attention fusion runs its single-product kernel. 10-bit mantissas like TF32, no range loss by construction:
             deviation from exact fp32 below an emulated-TF32 run's (tests/test_f16s_gpu.py). Not the default; bench.py reports it
             as `tf32_single_product_f16s` next to the headline.
  split3     (not a policy: a faster carrier of the "default" policy, inference only) when allow_tf32 is set, a kernel of this
             package that produces the left operand of a Linear can write it as a split-bf16 image (M, 3K) bfloat16
             [hi | hi | lo] (csrc/operand_split.hip, native.split3_rows); `linear_split3` multiplies it with the weight image
             [hi | lo | hi] in ONE plain bf16 GEMM with fp32 accumulation and output: the same three products the library
             forms inside its fp32 kernels, on its faster bf16 kernels (w12: 2.99 -> 2.67 ms, w3: 1.45 -> 1.27 ms at 65536 rows).
             `DIMSUM_SPLIT3=0` keeps the fp32 operands; launches with fewer than `DIMSUM_SPLIT3_MIN_ROWS` (8192) rows keep them too.
             Under autograd (training) the gated MLP (mlp.py) and every bias-free Linear that goes through `linear` run their
             forward, input-gradient and weight-gradient products on such images too (`DIMSUM_SPLIT3_TRAIN=0` switches that off).
  sliced     weight gradients reduce over the batch's rows into a small output; `mm_tn` / `mm_nn_rows` run such a product as one
             batched GEMM over row slices plus a sum when the output has fewer than 256 tiles of 256 x 256 (the library does
             not split the reduction itself: in_proj's weight gradient 1.59 -> 0.46 ms).
The operands are converted on every call -- weights too (a cached copy could not see in-place updates made through `.data`:
EMA updates, load_state_dict do not bump a tensor's version counter) -- except inside `frozen_weights()`, the scope a sampler
opens around its NFE loop, where each weight image is built once. Outputs stay fp32."""
import contextlib

import threading

import os

import torch
import torch.nn.functional as F

_policy = "default"


class _Local(threading.local):
    frozen = None       # weight-image cache of the innermost frozen_weights() scope of THIS thread, else None


_tls = _Local()


@contextlib.contextmanager
def frozen_weights():
    """Scope in which the caller guarantees that no weight changes (the NFE loop of a sampler: 250 forwards over constant
    weights): the split-bf16 weight images [hi | lo | hi] (and the fp16 copies of the opt-in policy) are built once per weight
    instead of once per call -- 128 ~14 us launches per DiM-L/2 forward. Outside such a scope every call converts afresh, because a
    cached image cannot see in-place updates made through `.data` (EMA, load_state_dict). The cache dies with the scope."""
    outer = _tls.frozen
    if outer is None:
        _tls.frozen = {}
    try:
        yield
    finally:
        if outer is None:
            _tls.frozen = None


def _cached(kind, weight, make):
    # Under hipGraph capture the cache is bypassed: a captured GEMM holds the raw pointer of the image it read, and the graph
    # (hip_graph.GraphedForward, kept by the caller across batches) outlives this scope -- a cached image would be freed under it
    # and would also freeze the weights of the first capture into every replay. Built inside the capture, the conversion kernel
    # is part of the graph (it re-reads the parameter on every replay) and its output lives in the graph's private pool.
    frozen = _tls.frozen
    if frozen is None or (weight.is_cuda and torch.cuda.is_current_stream_capturing()):
        return make()
    key = (kind, weight.data_ptr(), tuple(weight.shape), tuple(weight.stride()), weight.dtype)
    hit = frozen.get(key)
    if hit is None:
        # the entry keeps the weight alive: its storage cannot be freed and handed to another tensor while the image is cached
        hit = frozen[key] = (make(), weight)
    return hit[0]


def weight_image(weight):
    """weight (N, K) float32 -> its split-bf16 image (N, 3K) bfloat16 in weight order [hi | lo | hi]"""
    from . import native
    return _cached("w3", weight, lambda: native.split3_rows(weight.detach(), left=False))


def out_proj_planes_enabled(xz, weight, rows, scan_kernel=1):
    """whether MambaInnerFn's out_proj runs as gemm_tn over the scan's split-bf16 out_z planes (inference, fp32 under allow_tf32: the products
    the library's fp32 GEMM spends there too, without its conversion of the d-major operand): `rows` tokens of a (d_model, d_inner) weight.
    DIMSUM_OUT_PROJ_PLANES = auto (default): where the scan runs one of its state-split kernels (scan_kernel != 1, underfilled launches:
    DiM-XL/2 at 512 px 320.5 -> 315.7 ms per forward, the scan +1.5 %); the 64-channel kernel of a full chip is VALU-bound and pays +5 %
    for the conversion in its epilogue (0.338 -> 0.356 ms) against -0.06 ms in out_proj: -0.7 % per DiM-L/2 forward, but the headline
    kernel's roofline fraction 0.51 -> 0.49 -- left to `1` (always); `0` = never. tools/scratch/ab_planes.sh."""
    import os
    mode = os.environ.get("DIMSUM_OUT_PROJ_PLANES", "auto")
    if mode == "0" or (mode != "1" and scan_kernel == 1):
        return False
    # ... and only where BOTH kernels take the operands (else the fp32 out_z + library GEMM, never an error): the TN GEMM wants whole
    # 64-row reduction tiles with at least two of them per piece (d_inner % 64, d_inner >= 128), 256-column panels on both sides and
    # 32-bit in-tile offsets of the (2 d_inner, rows) plane pair (native.gemm_tn_supported); the scan's plane epilogue is part of its
    # 16-byte vector path (ssm_scan_fwd.hip: every xz row base 4-element aligned; seqlen % 8 is checked by the caller)
    D = weight.shape[1]
    return (_policy in ("default", "f16s") and torch.backends.cuda.matmul.allow_tf32 and os.environ.get("DIMSUM_SPLIT3", "1") != "0"
            and own_gemm_enabled() and xz.is_cuda and xz.dtype == torch.float32 and weight.dtype == torch.float32
            and weight.stride(1) == 1 and rows % 256 == 0 and weight.shape[0] % 256 == 0 and D % 64 == 0 and D >= 128
            and 128 * rows + 512 < 2 ** 31 and xz.data_ptr() % 16 == 0 and all(st % 4 == 0 for st in xz.stride()[:-1]) and xz.stride(-1) == 1
            and rows >= int(os.environ.get("DIMSUM_SPLIT3_MIN_ROWS", "8192")))


def out_proj_planes(planes, weight):
    """planes (2 D, M) bfloat16 [hi; lo] of the d-major out_z, weight (N, D) -> out_z^T weight^T (M, N) float32: hi.hi + hi.lo + lo.hi on the
    kernel's transposing-read variant; the plane pair is read as the row stack [hi; hi; lo]"""
    from . import native
    D = weight.shape[1]
    wt = _cached("w3t", weight, lambda: native.split3_rows_t(weight.detach()))
    return native.gemm_tn(planes, wt, alias_rows=D)


def out_proj_f16_enabled(xz, weight, rows, seqlen, scan_kernel=1):
    """whether MambaInnerFn's out_proj runs as ONE fp16 product per element over the scan's block-scaled fp16 out_z (inference under the
    scaled-fp16 policy; native.selective_scan_fwd(out_z_f16=True) + native.gemm_tn(scales = (table, ..))): the 64-channel scan kernel only
    (scan_kernel == 1), whole 32-step tiles, the TN GEMM's shapes. DIMSUM_OUT_PROJ_F16=0 hands out_proj back to the library's fp32 GEMM."""
    import os
    D = weight.shape[1]
    return (_policy == "f16s" and scan_kernel == 1 and os.environ.get("DIMSUM_OUT_PROJ_F16", "1") != "0" and torch.backends.cuda.matmul.allow_tf32
            and os.environ.get("DIMSUM_SPLIT3", "1") != "0" and own_gemm_enabled() and xz.is_cuda and xz.dtype == torch.float32
            and weight.dtype == torch.float32 and weight.stride(1) == 1 and rows % 256 == 0 and seqlen % 32 == 0 and weight.shape[0] % 256 == 0
            and D % 64 == 0 and D >= 128 and 128 * rows + 512 < 2 ** 31 and xz.data_ptr() % 16 == 0
            and all(st % 4 == 0 for st in xz.stride()[:-1]) and xz.stride(-1) == 1 and rows >= int(os.environ.get("DIMSUM_SPLIT3_MIN_ROWS", "8192")))


def _pad_cols_256(t):
    """(.., K, N) -> the same values as a column slice of a ZERO-padded (.., K, ceil(N / 256) 256) buffer (the TN kernel reads whole 256-column tiles of its
    right operand: native.gemm_tn_supported); a no-op copy when N % 256 == 0"""
    N = t.shape[-1]
    Np = (N + 255) // 256 * 256
    if Np == N:
        return t.contiguous()
    buf = torch.zeros(t.shape[:-1] + (Np,), device=t.device, dtype=t.dtype)
    buf[..., :N] = t
    return buf[..., :N]


def out_proj_f16_convert_enabled(xz, weight, rows, seqlen, scan_kernel):
    """whether MambaInnerFn's out_proj runs as ONE fp16 product per element where a STATE-SPLIT scan kernel serves the launch (scan_kernel != 1: fewer
    than 2048 waves, e.g. DiM-XL/2 at 512 px, batch 64): those kernels write fp32 out_z (16 channels per wave: no wave sees a 64-channel block), a
    conversion pass (native.rows_block_f16s, 6 bytes per element) builds the block-scaled image the 64-channel kernel would have written and the same
    TN product follows (out_proj_f16) -- instead of the library's fp32 GEMM on the d-major operand (three products + a transposing read: 361 -> ~200 us
    per mixer at XL/2-512). DIMSUM_OUT_PROJ_F16=0 switches it off."""
    import os
    D = weight.shape[1]
    return (_policy == "f16s" and scan_kernel != 1 and os.environ.get("DIMSUM_OUT_PROJ_F16", "1") != "0" and torch.backends.cuda.matmul.allow_tf32
            and os.environ.get("DIMSUM_SPLIT3", "1") != "0" and own_gemm_enabled() and xz.is_cuda and xz.dtype == torch.float32
            and weight.dtype == torch.float32 and weight.stride(1) == 1 and rows % 256 == 0 and seqlen % 32 == 0 and weight.shape[0] % 4 == 0
            and D % 64 == 0 and 128 <= D <= 4096 and 128 * rows + 512 < 2 ** 31 and xz.data_ptr() % 16 == 0
            and all(st % 4 == 0 for st in xz.stride()[:-1]) and xz.stride(-1) == 1 and rows >= int(os.environ.get("DIMSUM_SPLIT3_MIN_ROWS", "8192")))


def weight_f16s_t(weight):
    """weight (N, K) float32 -> (image^T (K, N) float16, inv (N,)): the scaled-fp16 rows of the weight as the right operand of a TN product
    (columns carry the scales); N % 256 != 0 (DiM-XL/2: 576): rows zero-padded to whole 256-column tiles behind the returned slice"""
    def make():
        img = weight_f16s(weight)
        return _pad_cols_256(img.data.t()), img.inv
    return _cached("w16t", weight, make)


def out_proj_f16(image, inv, weight):
    """image (D, M) float16, inv (M / 32, D / 64) float32: the scan's block-scaled out_z (include/dimsum_hip.h, out_z_f16); weight (N, D)
    -> out_z^T weight^T (M, N) float32, one fp16 product per element. A 32-token group goes onto ONE scale (its largest block's) inside
    the GEMM: the blocks are multiplied by exact powers of two <= 1 as they are read (a value more than 2^29 below its group's maximum
    loses bits there: 2^-40 of the maximum)."""
    from . import native
    wt, w_inv = weight_f16s_t(weight)
    return native.gemm_tn(image, wt, scales=(inv, w_inv))


def set_policy(policy):
    global _policy
    if policy not in ("default", "fp16", "f16s"):
        raise ValueError(f"unknown GEMM policy {policy!r}")
    _policy = policy


def get_policy():
    return _policy


def _use_fp16(x, weight):
    return (_policy == "fp16" and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)))


def _w16(weight):
    return _cached("w16", weight, lambda: weight.detach().to(torch.float16))


def _slices(rows, n_out, k_out):
    """how many slices of a `rows`-long reduction to run as one batched GEMM: a weight gradient (n_out, k_out) of 256 x 256 output
    tiles fills 256 CUs only if it has >= 256 tiles -- in_proj's (2048, 512) has 16 and took 1.59 ms as one GEMM, 0.46 ms as 16
    slices + a sum (tools/scratch/splitk_probe.py). Slices stay >= 2048 rows long."""
    tiles = ((n_out + 255) // 256) * ((k_out + 255) // 256)
    s = 1
    while s < 32 and tiles * s * 2 <= 256 and rows % (s * 2) == 0 and rows // (s * 2) >= 2048:
        s *= 2
    return s


def mm_tn(a, b, out_dtype=None):
    """a (R, N), b (R, K) -> a^T b (N, K): the weight-gradient shape (reduction over the R rows), split over row slices when the
    output is small"""
    from . import native as _nat
    if isinstance(a, _nat.PairImage):        # both operands as [hi | lo] pairs of (M, .) matrices (training images, mlp.py)
        return _nat.gemm_tn_pairs(a, b)
    R, N = a.shape
    K = b.shape[1]
    if a.is_cuda and a.dtype in (torch.bfloat16, torch.float16) and out_dtype in (None, torch.float32) and own_gemm_enabled():
        from . import native
        if native.gemm_tn_supported(a, b):      # operand images: the transposing-read variant of the hand-written kernel (dW12 2.99 -> 2.42 ms)
            y = native.gemm_tn(a, b)
            return y if out_dtype is not None else y.to(a.dtype)
    kw = {} if out_dtype is None else {"out_dtype": out_dtype}
    s = _slices(R, N, K) if a.is_cuda else 1
    if s == 1:
        return torch.mm(a.t(), b, **kw)
    return torch.bmm(a.view(s, R // s, N).transpose(1, 2), b.view(s, R // s, K), **kw).sum(0)


def mm_nn_rows(a_rows, b, out_dtype=None):
    """a_rows (N, R) (a d-major matrix: rows contiguous along the reduction), b (R, K) -> (N, K), split like mm_tn"""
    N, R = a_rows.shape
    K = b.shape[1]
    kw = {} if out_dtype is None else {"out_dtype": out_dtype}
    s = _slices(R, N, K) if a_rows.is_cuda else 1
    if s == 1:
        return torch.mm(a_rows, b, **kw)
    return torch.bmm(a_rows.view(N, s, R // s).permute(1, 0, 2), b.view(s, R // s, K), **kw).sum(0)


def mm_nt_rows(a_rows, b_rows, out_dtype=None):
    """a_rows (N, R), b_rows (K, R), both contiguous along the (long) reduction -> a_rows @ b_rows^T (N, K), split like mm_tn"""
    N, R = a_rows.shape
    K = b_rows.shape[0]
    kw = {} if out_dtype is None else {"out_dtype": out_dtype}
    s = _slices(R, N, K) if a_rows.is_cuda else 1
    if s == 1:
        return torch.mm(a_rows, b_rows.t(), **kw)
    return torch.bmm(a_rows.view(N, s, R // s).permute(1, 0, 2), b_rows.view(K, s, R // s).permute(1, 2, 0), **kw).sum(0)


class _MatmulWxFn(torch.autograd.Function):
    """weight (N, K) @ x^T (K, M) -> (N, M) (the in_proj site) with the weight gradient as a sliced reduction (mm_nn_rows)"""

    @staticmethod
    def forward(ctx, weight, x2):
        ctx.save_for_backward(weight, x2)
        return weight @ x2.t()

    @staticmethod
    def backward(ctx, dy):
        weight, x2 = ctx.saved_tensors
        dw = mm_nn_rows(dy.contiguous(), x2) if ctx.needs_input_grad[0] else None
        # dx row-major (M, K): its consumer (the pre-mixer's adjoint pass) reads channel-contiguous rows -- (W^T dy)^T as a view cost a
        # transposing copy of (M, K) per branch there (0.19 ms at 65536 x 512)
        dx = torch.mm(dy.t(), weight) if ctx.needs_input_grad[1] else None
        return dw, dx


class _LinearImagesFn(torch.autograd.Function):
    """y = x W^T under autograd with all three GEMMs (y, dx, dW) on split-bf16 operand images that this function builds itself
    (one conversion pass over x, one over dy): for a Linear whose producer kernel does not write the image. The weight-gradient
    product alone pays for the two passes (65536 rows, 1024 -> 1024: dW 1.74 -> 0.85 ms, dx 0.51 -> 0.37, y 0.43 -> 0.35;
    a pass costs ~0.14 ms; tools/scratch/ksplit_probe3.py)."""

    @staticmethod
    def forward(ctx, x, weight):
        from . import native
        K = x.shape[-1]
        x3 = native.split3_rows(x.reshape(-1, K), left=True)
        ctx.save_for_backward(x3, weight)
        ctx.x_shape = x.shape
        return linear_split3(x3, weight).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        from . import native
        x3, weight = ctx.saved_tensors
        N, K = weight.shape
        M = x3.shape[0]
        dy_w = native.split3_rows(dy.reshape(M, N).contiguous(), left=False)          # weight order [hi | lo | hi]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt3 = native.split3_rows(weight.detach().t().contiguous(), left=True)     # (K, 3N)
            dx = _nt(dy_w, wt3).view(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            dw = mm_tn(dy_w.view(3 * M, N), x3.view(3 * M, K), out_dtype=torch.float32)
        return dx, dw


def _rows_ok(x):
    """x (..., K) flattens to (M, K) rows the converter accepts without a copy"""
    return x.is_contiguous() and x.shape[-1] % 4 == 0


def linear(x, weight):
    """x (..., K) @ weight (N, K)^T -> (..., N), no bias (the bias rides in the consumer kernel)"""
    if (x.requires_grad or weight.requires_grad) and weight.shape[0] % 4 == 0 and _rows_ok(x):
        mode = split3_train_enabled(x, weight)
        if mode == "f16s":
            return _LinearF16sFn.apply(x, weight)
        if mode:
            return _LinearImagesFn.apply(x, weight)
    if not _use_fp16(x, weight):
        return F.linear(x, weight)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    # (a row-major cast of out_proj's d-major input was tried: torch's transposing copy costs more than the slower GEMM kernel)
    y = torch.mm(x2.to(torch.float16), _w16(weight).t(), out_dtype=torch.float32)
    return y.view(*x.shape[:-1], weight.shape[0])


class _MatmulWxF16sFn(torch.autograd.Function):
    """the in_proj site under autograd on the single-product carrier (policy "f16s"): xz (N, M) = W (N, K) x^T as an NT product of scaled-fp16
    images (d-major output, no copy); backward from ONE image of the d-major gradient dxz (N, M) -- fp16 rows with one scale per channel
    (native.rows_f16s, long rows):
        dW (N, K) = dxz x       the mixed-layout product (native.gemm_nn): dxz rows run along the reduction (tokens), x rows over it;
                                x's per-token scales travel as per-reduction-row factors
        dx (M, K) = dxz^T W     the TN product (native.gemm_tn(row_scales=..)): both operands' rows are the reduction index (channels)
    (mamba_simple.py in_proj under train.py:20-21's TF32: torch's autograd runs the same two products on TF32-rounded operands.)"""

    @staticmethod
    def forward(ctx, weight, x2):
        from . import native
        x16 = native.rows_f16s(x2)
        w16 = weight_f16s_train(weight)
        ctx.save_for_backward(w16.data, w16.inv, x16.data, x16.inv)
        return native.gemm_nt(w16.data, x16.data, scales=(w16.inv, x16.inv))

    @staticmethod
    def backward(ctx, dy):
        from . import native
        wd, wi, xd, xi = ctx.saved_tensors
        dy16 = native.rows_f16s(dy.contiguous())                 # (N, M): one scale per channel
        dw = native.gemm_nn(dy16.data, dy16.inv, xd, xi) if ctx.needs_input_grad[0] else None
        dx = None
        if ctx.needs_input_grad[1]:
            dx = native.gemm_tn(dy16.data, wd, row_invs=(dy16.inv.reshape(-1).contiguous(), wi.contiguous()))
        return dw, dx


def mamba_f16s_train_ok(M, d_model, d_inner):
    """shapes the single-product training GEMMs of a Mamba mixer take (in_proj / out_proj forward, input- and weight-gradients): whole tiles of the
    NT / TN / NN kernels on both sides, 32-bit in-tile offsets of the (channels, M) d-major operands"""
    return (own_gemm_enabled() and M % 256 == 0 and d_model % 256 == 0 and d_inner % 256 == 0 and d_model >= 128 and 256 * M * 2 < 2 ** 31)


def matmul_wx(weight, xt):
    """weight (N, K) @ xt (K, M) -> (N, M): the in_proj site, whose output is consumed d-major without a copy"""
    if not _use_fp16(xt, weight):
        x2 = xt.t()
        if (torch.is_grad_enabled() and weight.requires_grad and xt.is_cuda and x2.is_contiguous()
                and not torch.is_autocast_enabled("cuda")):          # (under autocast the plain product differentiates with mixed dtypes)
            if (split3_train_enabled(x2, weight) == "f16s" and mamba_f16s_train_ok(x2.shape[0], weight.shape[1], weight.shape[0] // 2)
                    and weight.shape[1] % 64 == 0 and x2.data_ptr() % 16 == 0):
                return _MatmulWxF16sFn.apply(weight, x2)
            return _MatmulWxFn.apply(weight, x2)
        return weight @ xt
    return torch.mm(_w16(weight), xt.to(torch.float16), out_dtype=torch.float32)


def split3_enabled(x, weight, producer="token", left=True):
    """the operand-image carriers serve exactly the launches the library would run as split-bf16 fp32 GEMMs: inference, fp32, allow_tf32.
    -> False, True (split-bf16 images: three bf16 products per fp32 product), "pair" (the same image stored as [hi | lo], `left` consumers
    on the hand-written GEMM) or "f16s" (policy "f16s": scaled-fp16 images, ONE product):
    the value is what the producer kernels take as their `split3` argument. producer: which kernel family writes the image ("token":
    csrc/token_transform.hip, "norm": csrc/norm.hip) -- their scaled-fp16 variants hold rows of different width."""
    import os
    mode = "f16s" if _policy == "f16s" else True
    if mode == "f16s" and x.shape[-1] > (1024 if producer == "token" else 2048):
        mode = True        # the token passes hold one channel group per thread (C <= 1024), the norm passes a row in registers (<= 2048):
                           # wider rows keep the split-bf16 images
    if not (_policy in ("default", "f16s") and torch.backends.cuda.matmul.allow_tf32 and os.environ.get("DIMSUM_SPLIT3", "1") != "0"
            and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.shape[-1] % 4 == 0
            and not torch.is_grad_enabled()):
        return False
    # the weight image is rebuilt per call (one ~14 us launch): it pays from a few thousand rows on, and in the launch-bound
    # small-batch regime every extra launch costs wall time -- fp32 operands below DIMSUM_SPLIT3_MIN_ROWS (default 8192) rows
    if x.numel() // x.shape[-1] < int(os.environ.get("DIMSUM_SPLIT3_MIN_ROWS", "8192")):
        return False
    # "pair": the image as [hi | lo] where its consumer is the hand-written GEMM, which reads it as [hi | hi | lo] (a_alias_rows; b_alias_rows
    # for in_proj, whose activation image is the RIGHT operand: left=False): the producer writes a third less.
    if (mode is True and own_gemm_enabled() and x.shape[-1] % 64 == 0 and (x.numel() // x.shape[-1] if left else weight.shape[0]) % 256 == 0
            and os.environ.get("DIMSUM_PAIR_IMAGES", "1") != "0"):
        return "pair"
    return mode


_K10 = 1.0 + 2.0 ** -10          # covers the fp16 rounding of the operands in the bound-derived scales


def weight_f16s(weight, want_l1=False):
    """weight (N, K) float32 -> F16Image (N, K) [, l1]: its scaled-fp16 image (one exact power-of-two scale per row); want_l1: and
    max_n sum_k |w_nk| as a 1-element device tensor (the bound-derived scales need it). ONE cache entry per weight (image, l1): built
    once per frozen_weights() scope -- or, for a whole model in one launch, by forward_scope()."""
    from . import native
    img, l1 = _cached("w16s", weight, lambda: native.rows_f16s(weight.detach(), want_l1=True))
    if not want_l1:
        return img
    return img, (l1() if callable(l1) else l1)          # (forward_scope keeps the bound-scaled value and divides on demand)


def _absmax(b, like):
    return b.detach().float().abs().max().reshape(1) if b is not None else torch.zeros(1, device=like.device)


def gated_bound(w12, b12):
    """the 2-element bound tensor of the gated GEMM epilogue's h image: {max_n sum_k |w_nk| (1 + 2^-10), max |b|}"""
    return _cached("w16s_bound", w12, lambda: torch.cat([weight_f16s(w12, want_l1=True)[1] * _K10, _absmax(b12, w12)]).contiguous())


def attn_kv_bound(w1, b1, w2=None, b2=None):
    """the 4-element bound tensor of the fp16 attention kernel (native.xattn_fusion_fwd(f16s=...)): {max_n sum_c |W1_nc| (1 + 2^-10), max|b1|,
    the same for the second qkv Linear (self-attention: the first again)}; the weight sums come with the weights' scaled-fp16 images"""
    def one(w, b):
        return [weight_f16s(w, want_l1=True)[1] * _K10, _absmax(b, w)]
    return _cached("kvbound", w1, lambda: torch.cat(one(w1, b1) + (one(w2, b2) if w2 is not None else one(w1, b1))).contiguous())


def qkv_f16s(x, weight, bias, rows_per_batch, bound2):
    """the qkv Linear of an attention under the scaled-fp16 policy with q | k | v written as scaled fp16 by the GEMM's epilogue (F16_QKV: q per
    row, k / v per batch element, from bound2 = this Linear's half {wl1 (1 + 2^-10), max|b|} of attn_kv_bound): x F16Image (M, K) ->
    (M, 3 C) float16, or None where the kernel does not take the shape (the caller then runs the fp32-output GEMM). The attention kernel
    reads half the bytes (it was HBM-bound on the fp32 qkv tensors) and the GEMM writes half. DIMSUM_QKV_F16=0 switches it off."""
    from . import native
    w16 = weight_f16s(weight)
    N = weight.shape[0]
    if not (os.environ.get("DIMSUM_QKV_F16", "1") != "0" and own_gemm_enabled() and native.gemm_nt_supported(x.data, w16.data) and rows_per_batch % 256 == 0
            and x.data.shape[0] % rows_per_batch == 0 and N % 3 == 0 and (N // 3) % 16 == 0):
        return None
    b = None if bias is None else bias.detach().float().contiguous()
    return native.gemm_nt(x.data, w16.data, bias=b, epilogue="f16_qkv", scales=(x.inv, w16.inv), gate_bound=bound2, rows_per_batch=rows_per_batch,
                          q_cols=N // 3)


def f16s_plan(model):
    """what a forward of `model` under the scaled-fp16 policy converts: [(weight, kind, bias, partner)] over its large bias-free-GEMM
    Linears -- the mixers' in_proj, the fusion's qkv1 / qkv2 / proj, the shared attention's qkv / proj, the gated MLP's w12 / w3
    (dimsum/models_dim.py:974-1117, mlp.py:49-70, attention_fusion.py:44-79). kind: "plain" (image), "gated" (image + the gate epilogue's
    bound with `bias`), "kv" (image + its half of the attention kernel's bound; `partner` = the other qkv Linear's weight or None),
    "plain_t" (the mixers' out_proj: the image transposed, the right operand of out_proj_f16's TN product)."""
    from .attention_fusion import CrossAttentionFusion
    from .mlp import GatedMLP
    plan = []
    for m in model.modules():
        if hasattr(m, "in_proj") and hasattr(m, "x_proj") and hasattr(m, "dt_proj"):
            plan.append((m.in_proj.weight, "plain", None, None))
            if getattr(m, "out_proj", None) is not None and getattr(m.out_proj, "bias", None) is None:
                plan.append((m.out_proj.weight, "plain_t", None, None))
        elif isinstance(m, CrossAttentionFusion):
            plan += [(m.qkv1.weight, "kv", m.qkv1.bias, m.qkv2), (m.proj.weight, "plain", None, None)]
        elif isinstance(m, GatedMLP):
            plan += [(m.w12.weight, "gated", m.w12.bias, None), (m.w3.weight, "plain", None, None)]
        elif type(m).__name__ == "Attention" and hasattr(m, "qkv") and hasattr(m, "proj"):
            plan += [(m.qkv.weight, "kv", m.qkv.bias, None), (m.proj.weight, "plain", None, None)]
    return plan


@contextlib.contextmanager
def forward_scope(model, rows):
    """ONE inference forward of `model` over `rows` tokens under the scaled-fp16 policy: every weight image, weight L1 bound and bias maximum
    the forward will ask for is built up front by ONE multi-job launch per 24 jobs (native.rows_f16s_multi) instead of ~150 small launches
    and ~250 small torch reductions between the big kernels (3 ms of a 90-ms DiM-L/2 forward at batch 256), and lives exactly as long as
    the forward -- weights cannot change inside one. Everything is still rebuilt on EVERY forward (`.data` updates between forwards are
    seen); inside a frozen_weights() scope, under hipGraph capture (the conversion kernels must be part of the graph and their outputs
    live in its pool: _cached bypasses the cache there) or another policy this is a no-op. Under autograd (training under the policy) the same
    launch serves the training forward's weight images (weight_f16s_train, gated_bound_train)."""
    import os
    from . import native
    first = next(model.parameters(), None)
    # under autograd (training on the single-product carrier, section 3.7 of DESIGN.md) the scope serves the forward's weight images too:
    # gemm.weight_f16s_train finds them here (~150 conversion launches per step); the transposed out_proj images are inference-only
    training = torch.is_grad_enabled()
    if (_policy != "f16s" or _tls.frozen is not None or not torch.backends.cuda.matmul.allow_tf32 or first is None
            or (training and (os.environ.get("DIMSUM_F16S_TRAIN", "1") == "0" or os.environ.get("DIMSUM_SPLIT3_TRAIN", "1") == "0"
                              or os.environ.get("DIMSUM_FORWARD_SCOPE_TRAIN", "1") == "0" or first.dtype != torch.float32 or torch.is_autocast_enabled("cuda")))
            or not first.is_cuda or torch.cuda.is_current_stream_capturing() or not own_gemm_enabled() or os.environ.get("DIMSUM_SPLIT3", "1") == "0"
            or rows < int(os.environ.get("DIMSUM_SPLIT3_MIN_ROWS", "8192")) or os.environ.get("DIMSUM_FORWARD_SCOPE", "1") == "0"):
        yield
        return
    # the plan is cached on the model, keyed by the identity of its parameters: a Linear swapped (or re-created) after the first forward rebuilds it
    ident = tuple(id(p_) for p_ in model.parameters())
    cached = model.__dict__.get("_f16s_plan")
    if cached is None or cached[0] != ident:
        cached = model.__dict__["_f16s_plan"] = (ident, f16s_plan(model))
    plan = cached[1]

    def src_ok(t):          # what dimsum_rows_f16s_multi takes as a job's source: fp32 rows, 16-byte aligned, lengths / strides % 4
        if t is None:
            return True
        t2 = t if t.dim() == 2 else t.reshape(1, -1)
        return (t.is_cuda and t.dtype == torch.float32 and t2.stride(1) == 1 and t2.shape[1] % 4 == 0 and (t2.shape[0] == 1 or t2.stride(0) % 4 == 0)
                and t.data_ptr() % 16 == 0)
    jobs, slot = [], 0
    tbufs = {}                      # "plain_t": weight shape -> job indices
    layout = []                     # per plan entry: (first job index, slot of its l1 [, ...])
    for w, kind, bias, partner in plan:
        # every source of the entry's jobs is validated up front (the weight, its bias, the partner qkv Linear's weight and bias): an entry
        # with anything the multi-job kernel does not take drops to the lazy per-weight path instead of failing the whole forward
        ok = src_ok(w) and src_ok(bias) and (partner is None or (src_ok(partner.weight) and src_ok(partner.bias)))
        if not ok:
            layout.append(None)
            continue
        layout.append((len(jobs), slot))
        if kind == "plain":
            jobs.append((w.detach(), True, slot, None, 1.0))            # (the L1 norm rides along: one cache entry kind per weight)
            slot += 1
        elif kind == "plain_t" and (training or os.environ.get("DIMSUM_OUT_PROJ_F16", "1") == "0"):
            layout[-1] = None
        elif kind == "plain_t":
            # all transposed images of one shape share ONE buffer: one transposing copy per shape after the launch instead of one per weight
            buf = tbufs.setdefault(tuple(w.shape), [])
            buf.append(len(jobs))
            jobs.append((w.detach(), True, slot, None, 1.0, None))
            slot += 1
        elif kind == "gated":                                            # slots: l1 (1 + 2^-10), max |b| -- the bound tensor itself; then the plain l1
            jobs.append((w.detach(), True, slot, None, _K10))
            if bias is not None:
                jobs.append((bias.detach().float(), False, None, slot + 1, 1.0))
            slot += 2
        else:                                                            # kv: slots l1_1 K10, max|b1|, l1_2 K10, max|b2| (self-attention: the pair twice)
            ws = [(w, bias)] + ([(partner.weight, partner.bias)] if partner is not None else [])
            for i, (wi, bi) in enumerate(ws):
                jobs.append((wi.detach(), True, slot + 2 * i, None, _K10))
                if bi is not None:
                    jobs.append((bi.detach().float(), False, None, slot + 2 * i + 1, 1.0))
            slot += 4
    tdata = {}
    for shape, idx in tbufs.items():
        data = torch.empty((len(idx),) + shape, device=first.device, dtype=torch.float16)
        inv = torch.empty((len(idx), shape[0]), device=first.device, dtype=torch.float32)
        for n, j in enumerate(idx):
            jobs[j] = jobs[j][:5] + ((data[n], inv[n]),)
            tdata[j] = (shape, n)
        tbufs[shape] = (data, inv)
    images, scal = native.rows_f16s_multi(jobs, n_slots=slot)      # (every entry's slots exist, also a trailing bias-free one's)
    tbufs = {shape: (_pad_cols_256(data.transpose(1, 2)), inv) for shape, (data, inv) in tbufs.items()}
    cache = {}
    key = lambda kind, w: (kind, w.data_ptr(), tuple(w.shape), tuple(w.stride()), w.dtype)
    for (w, kind, bias, partner), lay in zip(plan, layout):
        if lay is None:
            continue
        j, sl = lay
        if kind == "plain":
            cache[key("w16s", w)] = ((images[j], scal[sl:sl + 1]), w)
        elif kind == "plain_t":
            shape, n = tdata[j]
            cache[key("w16t", w)] = ((tbufs[shape][0][n], tbufs[shape][1][n]), w)
        elif kind == "gated":
            cache[key("w16s", w)] = ((images[j], lambda i=sl: scal[i:i + 1] / _K10), w)          # (only evaluated if someone asks for the raw l1)
            cache[key("w16s_bound", w)] = (scal[sl:sl + 2], w)
        else:
            cache[key("w16s", w)] = ((images[j], lambda i=sl: scal[i:i + 1] / _K10), w)
            if partner is not None:
                j2 = j + (2 if bias is not None else 1)
                cache[key("w16s", partner.weight)] = ((images[j2], lambda i=sl + 2: scal[i:i + 1] / _K10), partner.weight)
                cache[key("kvbound", w)] = (scal[sl:sl + 4], w)
            else:
                cache[key("kvbound", w)] = (torch.cat([scal[sl:sl + 2], scal[sl:sl + 2]]), w)
    if not training and os.environ.get("DIMSUM_FORWARD_MEMO", "1") != "0":
        # A = -exp(A_log) of every mixer (mamba_simple.py:120: two tiny launches per mixer call) as two multi-tensor launches per forward
        alogs = [a for m in model.modules() for a in (getattr(m, "A_log", None), getattr(m, "A_b_log", None)) if isinstance(a, torch.nn.Parameter) and a.is_cuda]
        if alogs:
            vals = torch._foreach_neg(torch._foreach_exp([a.detach().float() for a in alogs]))
            for a, v in zip(alogs, vals):
                cache[key("negexpA", a)] = (v, a)
    _tls.frozen = cache
    try:
        yield
    finally:
        _tls.frozen = None


def neg_exp(a_log):
    """-exp(A_log.float()) (mamba_simple.py:120); inside an inference forward_scope: the value built for all mixers at once"""
    frozen = _tls.frozen
    if frozen is not None and not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing():
        hit = frozen.get(("negexpA", a_log.data_ptr(), tuple(a_log.shape), tuple(a_log.stride()), a_log.dtype))
        if hit is not None:
            return hit[0]
    return -torch.exp(a_log.float())


def own_gemm_enabled():
    """the image GEMMs run on this package's MFMA kernel (csrc/gemm_nt_kernel.hpp); DIMSUM_GEMM_NT=0 hands them back to the library"""
    import os
    return os.environ.get("DIMSUM_GEMM_NT", "1") != "0"


def _nt(a, b, **kw):
    """a (M, K) @ b (N, K)^T -> float32 (M, N) (or the fused epilogue's image), 16-bit K-contiguous operands: the hand-written kernel
    when the shape fits its tiling (256-row panels, 64-deep K tiles), else the library"""
    from . import native
    if own_gemm_enabled() and native.gemm_nt_supported(a, b):
        return native.gemm_nt(a, b, **kw)
    y = torch.mm(a, b.t(), out_dtype=torch.float32)
    return y if kw.get("bias") is None else y + kw["bias"]


def _tail(y, bias, residual, gate, rows_per_batch):
    """the unfused form of the gate + residual epilogue: residual + gate[row // rows_per_batch] * (y + bias)"""
    if bias is not None:
        y = y + bias
    if residual is None:
        return y
    if gate is not None:
        y = (y.view(-1, rows_per_batch, y.shape[-1]) * gate.unsqueeze(1)).view(y.shape)
    return residual + y


def _nt_f16s(a, b, bias=None, residual=None, gate=None, rows_per_batch=None):
    """F16Image a (M, K) x F16Image b (N, K)^T -> (M, N) float32: ONE fp16 product per element, the row scales undone in the epilogue"""
    from . import native
    if own_gemm_enabled() and native.gemm_nt_supported(a.data, b.data) and (gate is None or rows_per_batch % 256 == 0):
        return native.gemm_nt(a.data, b.data, bias=bias, scales=(a.inv, b.inv), residual=residual, gate=gate, rows_per_batch=rows_per_batch)
    return _tail(torch.mm(a.data, b.data.t(), out_dtype=torch.float32) * a.inv[:, None] * b.inv[None, :], bias, residual, gate, rows_per_batch)


def linear_split3(x3, weight, bias=None, residual=None, gate=None, rows_per_batch=None):
    """x3 (M, 3K) bfloat16 left image [hi | hi | lo] (or a scaled-fp16 F16Image (M, K)) @ weight (N, K)^T -> (M, N) float32.
    bias (N) / residual (M, N) / gate (M / rows_per_batch, N): residual + gate[row // rows_per_batch] * (x W^T + bias) -- the residual
    tail of a block ("x = x + gate * mlp(...)", models_dim.py:1107-1113) in the epilogue of the GEMM instead of a pass of its own."""
    from . import native
    if isinstance(x3, native.F16Image):
        return _nt_f16s(x3, weight_f16s(weight), bias, residual, gate, rows_per_batch)
    w3i = weight_image(weight)
    if isinstance(x3, native.PairImage) and not (own_gemm_enabled() and native.gemm_nt_supported(x3, w3i) and (gate is None or rows_per_batch % 256 == 0)):
        x3 = x3.image3()            # a consumer without the kernel: the three-piece image the library GEMM needs
    if residual is None and bias is None:
        return _nt(x3, w3i)
    if own_gemm_enabled() and native.gemm_nt_supported(x3, w3i) and (gate is None or rows_per_batch % 256 == 0):
        return native.gemm_nt(x3, w3i, bias=bias, residual=residual, gate=gate, rows_per_batch=rows_per_batch)
    return _tail(torch.mm(x3, w3i.t(), out_dtype=torch.float32), bias, residual, gate, rows_per_batch)


def matmul_wx_split3(weight, x3, conv=None):
    """weight (N, K) @ x^T -> (N, M) float32 with x given as its left image x3 (M, 3K) / F16Image: the in_proj site (d-major output).
    conv = (conv_weight (D, width), conv_bias (D) or None, seq): ask for the mixer's causal conv1d + SiLU over the first D output rows in the
    GEMM's epilogue (native.gemm_nt(conv=...): sequences of `seq` tokens, 256 % seq == 0) -> (product, True) when the kernel took it, else
    (plain product, False): the caller then runs the conv kernel as before. DIMSUM_INPROJ_CONV=0 switches the fusion off."""
    from . import native
    if conv is not None:
        cw, cb, seq = conv
        a = weight_f16s(weight) if isinstance(x3, native.F16Image) else weight_image(weight)
        ad, bd = (a.data, x3.data) if isinstance(x3, native.F16Image) else (a, x3)
        ok = (os.environ.get("DIMSUM_INPROJ_CONV", "1") != "0" and own_gemm_enabled() and native.gemm_nt_supported(ad, bd)
              and cw.dtype == torch.float32 and cw.dim() == 2 and cw.stride(1) == 1 and cw.shape[0] % 256 == 0 and 2 <= cw.shape[1] <= 4
              and seq % 4 == 0 and 256 % seq == 0 and x3.shape[0] % seq == 0)
        if ok:
            cbf = None if cb is None else cb.detach().float().contiguous()
            if isinstance(x3, native.F16Image):
                return native.gemm_nt(ad, bd, scales=(a.inv, x3.inv), conv=(cw.detach(), cbf, seq)), True
            return native.gemm_nt(a, x3, conv=(cw.detach(), cbf, seq)), True
        return matmul_wx_split3(weight, x3), False
    if isinstance(x3, native.F16Image):
        return _nt_f16s(weight_f16s(weight), x3)
    w3i = weight_image(weight)
    if isinstance(x3, native.PairImage) and not (own_gemm_enabled() and native.gemm_nt_supported(w3i, x3)):
        x3 = x3.image3()
    return _nt(w3i, x3)


def gated_mlp_hidden_split3(x3, w12, b12):
    """x3 (M, 3K) left image, w12 (2F, K), b12 (2F) or None -> the left image (M, 3F) of gelu_tanh(x W12a^T + b) * (x W12b^T + b)
    (dimsum/mlp.py:66-70): ONE kernel, the gate in the GEMM's epilogue -- the fp32 (M, 2F) tensor never exists; the library GEMM + the
    gated-GeLU pass (csrc/token_transform.hip) when the shape does not fit the kernel's tiling."""
    from . import native
    if isinstance(x3, native.F16Image):
        w16 = weight_f16s(w12)
        if own_gemm_enabled() and native.gemm_nt_supported(x3.data, w16.data, gated=True):
            # the h image's per-row scale comes from the bound |x1|, |x2| <= max|x_r| * max_n sum_k |w_nk| + max|b| -- no row reduction
            # (gated_bound: the 2^-10 on the weight bound covers the fp16 rounding of the operands)
            return native.gemm_nt(x3.data, w16.data, bias=b12, epilogue="gated_f16", scales=(x3.inv, w16.inv), gate_bound=gated_bound(w12, b12))
        x12 = _nt_f16s(x3, w16)
        return native.rows_f16s(native.gated_gelu_fwd(x12, b12))
    w3i = weight_image(w12)
    if own_gemm_enabled() and native.gemm_nt_supported(x3, w3i, gated=True):
        # the h image as the pair [hi | lo] where its consumer (the w3 GEMM on the same kernel) can read it as [hi | hi | lo]: the epilogue
        # stores a third less (1.6 -> 1.07 GB per MLP at 65536 x 4096)
        pair = (w12.shape[0] // 2) % 64 == 0 and os.environ.get("DIMSUM_PAIR_IMAGES", "1") != "0"
        return native.gemm_nt(x3, w3i, bias=b12, epilogue="gated_split3", pair_out=pair)
    if isinstance(x3, native.PairImage):
        x3 = x3.image3()
    return native.gated_gelu_fwd(torch.mm(x3, w3i.t(), out_dtype=torch.float32), b12, split3=True)


def train_pairs_enabled(M, *widths):
    """the training images of the gated MLP as [hi | lo] pairs (every consumer is the hand-written kernel: NT with K-tile aliasing, TN over
    piece ranges): a third less image traffic (gated-GeLU adjoint 3.2 -> 2.1 GB, h image 1.6 -> 1.07 GB at 65536 x 4096)"""
    return (own_gemm_enabled() and os.environ.get("DIMSUM_PAIR_IMAGES", "1") != "0" and M % 256 == 0 and M >= 2048
            and all(w % 256 == 0 for w in widths))


def gated_mlp_hidden_split3_train(x3, w12, b12):
    """the training forward of the same product: -> (h image (M, 3F), x12 (M, 2F) float32 without the bias): the backward's gated-GeLU
    adjoint reads x12, so the GEMM's gate epilogue stores its accumulators next to the image (one kernel) instead of a plain GEMM
    followed by a gate pass that reads them back"""
    from . import native
    w3i = weight_image(w12)
    if own_gemm_enabled() and native.gemm_nt_supported(x3, w3i, gated=True):
        return native.gemm_nt(x3, w3i, bias=b12, epilogue="gated_split3", keep_x12=True, pair_out=isinstance(x3, native.PairImage))
    if isinstance(x3, native.PairImage):
        x3 = x3.image3()
    x12 = _nt(x3, w3i)
    return native.gated_gelu_fwd(x12, b12, split3=True), x12


def split3_train_enabled(x, weight):
    """the operand-image carrier under autograd (mlp.py: forward AND backward GEMMs of the gated MLP; `linear`: qkv / proj; the Mamba GEMMs):
    fp32 CUDA training under allow_tf32, outside autocast; DIMSUM_SPLIT3_TRAIN=0 / DIMSUM_SPLIT3=0 switch it off; same row threshold as
    inference. -> False, True (split-bf16 images: three bf16 products per fp32 product) or "f16s" (policy "f16s": scaled-fp16 images, ONE fp16
    product per element in the forward, the input-gradient AND the weight-gradient GEMMs -- the reference trains under TF32 too,
    dimsum/train.py:20-21; DIMSUM_F16S_TRAIN=0 keeps the three-product images under that policy)"""
    import os
    if not (_policy in ("default", "f16s") and torch.backends.cuda.matmul.allow_tf32 and os.environ.get("DIMSUM_SPLIT3", "1") != "0"
            and os.environ.get("DIMSUM_SPLIT3_TRAIN", "1") != "0" and x.is_cuda and x.dtype == torch.float32
            and weight.dtype == torch.float32 and x.shape[-1] % 4 == 0 and torch.is_grad_enabled()
            and not torch.is_autocast_enabled("cuda")):
        return False
    if x.numel() // x.shape[-1] < int(os.environ.get("DIMSUM_SPLIT3_MIN_ROWS", "8192")):
        return False
    if _policy == "f16s" and os.environ.get("DIMSUM_F16S_TRAIN", "1") != "0" and own_gemm_enabled():
        return "f16s"
    return True


def attn_bwd_f16_enabled(qkv):
    """the attention backward pair (csrc/xattn_fusion_bwd.hip) on the single-product fp16 carrier: the "f16s" policy's training arithmetic
    (split3_train_enabled's switches, no row threshold: the kernels are the same at every size)"""
    import os
    return bool(_policy == "f16s" and torch.backends.cuda.matmul.allow_tf32 and os.environ.get("DIMSUM_F16S_TRAIN", "1") != "0"
                and os.environ.get("DIMSUM_SPLIT3_TRAIN", "1") != "0" and os.environ.get("DIMSUM_SPLIT3", "1") != "0"
                and qkv.is_cuda and qkv.dtype == torch.float32 and not torch.is_autocast_enabled("cuda"))


def weight_f16s_train(weight, want_l1=False):
    """training: the scaled-fp16 image of a weight, rebuilt on every forward (the optimizer changes it between steps) -> F16Image [, l1]: from the
    forward's one multi-job launch when DiM.forward opened a forward_scope, else by a launch of its own"""
    from . import native
    if _tls.frozen is not None and not torch.cuda.is_current_stream_capturing():
        return weight_f16s(weight, want_l1=want_l1)
    return native.rows_f16s(weight.detach(), want_l1=want_l1)


def train_scope_active():
    """whether a forward_scope holds this forward's weight images (training under the f16s policy through DiM.forward)"""
    return _tls.frozen is not None and not torch.cuda.is_current_stream_capturing()


def weight_t_f16s_train(weight):
    """training: the scaled-fp16 image of weight^T ((K, N) rows: one scale per INPUT feature) -- the right operand of the input-gradient
    product dx = dy W as an NT GEMM over the output features"""
    from . import native
    return native.rows_f16s(weight.detach().t().contiguous())


def nt_f16s_any(a, b):
    """F16Image a (M, K) x F16Image b (N, K)^T -> (M, N) float32, on the hand-written kernel where the shape fits, else the library on the decoded rows"""
    from . import native
    if own_gemm_enabled() and native.gemm_nt_supported(a.data, b.data):
        return native.gemm_nt(a.data, b.data, scales=(a.inv.reshape(-1), b.inv))
    return torch.mm(a.data, b.data.t(), out_dtype=torch.float32) * a.inv.reshape(-1)[:, None] * b.inv[None, :]


def dx_f16s(dy16, w16):
    """dx (M, K) = dy (M, N) W (N, K) from the image of dy (one scale per token) and THE FORWARD'S image of W (one scale per output feature n --
    the reduction index here): the mixed-layout kernel (native.gemm_nn: dy's rows run along the reduction, W's over it, W's scales travel as
    per-reduction-row factors) -- no transposed weight image, no second conversion of the weight"""
    from . import native
    a = dy16.data.reshape(-1, dy16.data.shape[-1])
    if own_gemm_enabled() and native.gemm_nn_supported(a, w16.data) and w16.data.shape[0] <= 16384:
        return native.gemm_nn(a, dy16.inv.reshape(-1).contiguous(), w16.data, w16.inv)
    return torch.mm(a.float() * dy16.inv.reshape(-1)[:, None], w16.data.float() * w16.inv[:, None])


def dw_f16s(dy16, x16):
    """dW = dy^T x (N, K) float32 from two scaled-fp16 images whose rows (tokens) are the reduction index: the TN kernel with per-reduction-row
    factors (native.gemm_tn(row_scales=...)) -- ONE fp16 product per element -- else the library on the decoded rows"""
    from . import native
    a, b = dy16.data.reshape(-1, dy16.data.shape[-1]), x16.data.reshape(-1, x16.data.shape[-1])
    ai, bi = dy16.inv.reshape(-1), x16.inv.reshape(-1)
    if own_gemm_enabled() and native.gemm_tn_supported(a, b):
        return native.gemm_tn(a, b, row_invs=(ai.contiguous(), bi.contiguous()))
    return mm_tn(a.float() * ai[:, None], b.float() * bi[:, None])


class _LinearF16sFn(torch.autograd.Function):
    """y = x W^T under autograd with all three GEMMs (y, dx, dW) as ONE fp16 product per element over scaled-fp16 images (policy "f16s"; the
    reference's TF32 training arithmetic, dimsum/train.py:20-21: 10-bit operand mantissas, fp32 accumulation): x, W and dy are converted once each
    (row scales = exact row maxima); y = x16 W16^T is an NT product whose epilogue undoes the row scales; dx = dy16 W16 reduces over W's rows
    (gemm_nn: W's row scales as per-reduction-row factors), dW = dy16^T x16 over the tokens (gemm_tn: both operands' row scales as factors)."""

    @staticmethod
    def forward(ctx, x, weight):
        from . import native
        K = x.shape[-1]
        x16 = native.rows_f16s(x.reshape(-1, K))
        w16 = weight_f16s_train(weight)
        ctx.save_for_backward(x16.data, x16.inv, w16.data, w16.inv)          # (the weight's image serves the backward too: half the weight's bytes)
        ctx.x_shape = x.shape
        return nt_f16s_any(x16, w16).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        from . import native
        xd, xi, wd, wi = ctx.saved_tensors
        N = wd.shape[0]
        M = xd.shape[0]
        dy16 = native.rows_f16s(dy.reshape(M, N).contiguous())
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = dx_f16s(dy16, native.F16Image(wd, wi)).view(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            dw = dw_f16s(dy16, native.F16Image(xd, xi))
        return dx, dw
