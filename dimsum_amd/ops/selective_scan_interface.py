"""Autograd front-ends of the selective scan and of the fused Mamba inner block -- same names, argument order and
return conventions as mamba/mamba_ssm/ops/selective_scan_interface.py, implemented on dimsum_amd.native (HIP, gfx950).

  selective_scan_fn                      <- :94-101   (SelectiveScanFn :12-91)
  mamba_inner_fn[_cond]                  <- :1277-1348 (MambaInnerFn :579-790, MambaInnerFnCond :793-1007)
  mamba_inner_fn_no_out_proj[_cond]      <- :1389-1452 (MambaInnerFnNoOutProj[Cond] :174-576)

The four fused variants of the reference are four near-identical 200-line classes; here they are ONE Function with two
switches (out-projection, conditional conv entry). Semantics kept: checkpoint_lvl=1 recomputation of conv1d_out and
delta in the backward, dx/dz written side by side into one `dxz`, `dcond = None` (SURVEY finding 1), fp32 dB/dC
accumulation cast back to the input dtype, autocast-aware weight casts.
One deliberate difference in what is SAVED (not in what is computed): the reference drops `out_z` after the forward and has
the backward kernel recompute and re-write it (selective_scan_interface.py:952, `recompute_out_z=True`) to save one
activation tensor on 40-80 GB GPUs. With 288 GB of HBM the forward's `out_z` is simply kept for `d out_proj.weight`, and the
backward kernel skips that store (0.27 GB less traffic per call at DiM-L/2, batch 256). DIMSUM_RECOMPUTE_OUT_Z=1 restores
the reference's trade.

Not implemented (raise): complex A, constant (non input-dependent) B/C -- unused by DiMSUM (mamba_simple.py:586,602-603).
"""
import os

import torch
import torch.nn.functional as F
from torch.amp import custom_bwd, custom_fwd

from .. import gemm, native


def _last_contig(t):
    return t if t is None or t.stride(-1) == 1 else t.contiguous()


class SelectiveScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, return_last_state=False,
                need_ckpt=False):
        u, delta, B, C, z = (_last_contig(t) for t in (u, delta, B, C, z))
        D = D.contiguous() if D is not None else None
        if B.dim() < 3 or C.dim() < 3:
            raise NotImplementedError("selective_scan_fn: constant B/C (shape (dim, dstate)) is outside this build's scope")
        ctx.squeeze_B, ctx.squeeze_C = B.dim() == 3, C.dim() == 3
        if ctx.squeeze_B:
            B = B.unsqueeze(1)
        if ctx.squeeze_C:
            C = C.unsqueeze(1)
        # training extra: saved states for the backward kernel (`need_ckpt` is decided by the caller: inside forward()
        # grad mode is off and ctx.needs_input_grad ignores torch.no_grad())
        out, x, *rest = native.selective_scan_fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus, need_ckpt=need_ckpt)
        ckpt = rest.pop() if need_ckpt else None
        ctx.delta_softplus = delta_softplus
        ctx.has_z = z is not None
        ctx.has_D, ctx.has_bias = D is not None, delta_bias is not None
        last_state = x[:, :, -1, 1::2]                     # (batch, dim, dstate)   (:39)
        ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias, x, out if ctx.has_z else None, ckpt)
        res = rest[0] if ctx.has_z else out
        return res if not return_last_state else (res, last_state)

    @staticmethod
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, z, delta_bias, x, out, ckpt = ctx.saved_tensors
        dout = _last_contig(dout)
        du, ddelta, dA, dB, dC, dD, ddelta_bias, *rest = native.selective_scan_bwd(
            u, delta, A, B, C, D, z, delta_bias, dout, x, out, None, ctx.delta_softplus, False, ckpt=ckpt)
        dz = rest[0] if ctx.has_z else None
        dB = dB.squeeze(1) if ctx.squeeze_B else dB
        dC = dC.squeeze(1) if ctx.squeeze_C else dC
        return (du, ddelta, dA, dB, dC, dD if ctx.has_D else None, dz, ddelta_bias if ctx.has_bias else None, None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, return_last_state=False):
    """out (or (out, last_state)); the gradient of last_state is not propagated (as in the reference)."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state,
                                 _will_backprop(u, delta, A, B, C, D, z, delta_bias))


def _f16s_train(xz, out_proj_weight, will_backprop):
    """whether this training call's out_proj GEMMs run on the single-product carrier (gemm.split3_train_enabled == "f16s" and shapes the kernels take)"""
    if not will_backprop or out_proj_weight is None or not xz.is_cuda or xz.dtype != torch.float32:
        return False
    bsz, d2, L = xz.shape
    return (gemm.split3_train_enabled(_Rows(bsz * L, d2, xz), out_proj_weight) == "f16s"
            and gemm.mamba_f16s_train_ok(bsz * L, out_proj_weight.shape[0], d2 // 2))


class _Rows:
    """a shape-only stand-in for the (tokens, features) view of a d-major tensor: what gemm.split3_train_enabled looks at"""

    def __init__(self, rows, cols, like):
        self.shape, self.is_cuda, self.dtype, self._n = (rows, cols), like.is_cuda, like.dtype, rows * cols

    def numel(self):
        return self._n


def _will_backprop(*tensors):
    """True when autograd will record this call: only then is it worth storing the scan's saved states"""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _checkpoint_lvl():
    """The reference hard-codes checkpoint_lvl = 1 (selective_scan_interface.py:588: conv_out and delta are dropped after the forward and recomputed in
    the backward -- a conv1d launch and a dt_proj GEMM per mixer, to save 2 b d l 4 bytes of activations on 40 / 80-GB parts). With 288 GB per GPU the
    default here is 0: both tensors stay (4.4 GB at DiM-L/2, 64 latents) and the backward starts from them; the gradients are bit-identical
    (the recomputation is deterministic). DIMSUM_MAMBA_CHECKPOINT_LVL=1 restores the reference's trade."""
    return 1 if os.environ.get("DIMSUM_MAMBA_CHECKPOINT_LVL", "0") == "1" else 0


def _rows(t):   # "b d l -> d (b l)" as a view-friendly reshape
    b, d, l = t.shape
    return t.permute(1, 0, 2).reshape(d, b * l)


class _MambaInner(torch.autograd.Function):
    """conv1d(silu) -> x_proj -> dt_proj -> selective scan (+ silu(z) gate) [-> out_proj]."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, init_states, has_out_proj, checkpoint_lvl,
                need_ckpt=False, conv_done=False, f16s_train=False):
        # conv_done (inference extra, not in the reference's signature): xz[:, :d_inner] already holds the causal conv1d + SiLU of the in_proj
        # output (the GEMM's epilogue formed it, modules/mamba_simple.py) -- the conv kernel is skipped
        # f16s_train (training extra, decided by the caller like need_ckpt: grad mode is off in here): out_proj's forward, input-gradient and
        # weight-gradient GEMMs as ONE fp16 product per element over scaled-fp16 images (gemm.py, policy "f16s")
        assert checkpoint_lvl in (0, 1)
        assert not (conv_done and need_ckpt), "conv_done is an inference extra"
        if A.is_complex():
            raise NotImplementedError("mamba_inner_fn: complex A is outside this build's scope")
        if B is not None or C is not None:
            raise NotImplementedError("mamba_inner_fn: constant B/C is outside this build's scope")
        L = xz.shape[-1]
        R = delta_proj_weight.shape[1]
        N = A.shape[-1]
        if torch.is_autocast_enabled("cuda"):
            adt = torch.get_autocast_dtype("cuda")
            x_proj_weight, delta_proj_weight = x_proj_weight.to(adt), delta_proj_weight.to(adt)
            if has_out_proj:
                out_proj_weight = out_proj_weight.to(adt)
                out_proj_bias = out_proj_bias.to(adt) if out_proj_bias is not None else None
        xz = _last_contig(xz)
        conv_w = conv1d_weight.reshape(conv1d_weight.shape[0], conv1d_weight.shape[-1])     # "d 1 w -> d w"
        conv_b = conv1d_bias.contiguous() if conv1d_bias is not None else None
        x, z = xz.chunk(2, dim=1)
        # `init_states` is numerically dead in the reference (its buffer is overwritten with the plain conv result,
        # causal_conv1d.cpp:326-329). A full-size buffer is honoured as the output buffer; anything else (e.g. the
        # (batch, d_inner) cond_proj output CondMamba passes) only keeps the autograd edge.
        if conv_done:
            conv_out = x
        elif init_states is not None and init_states.shape == x.shape and init_states.dtype == x.dtype and init_states.stride(-1) == 1:
            conv_out = native.causal_conv1d_fwd_cond(x, conv_w, conv_b, True, init_states)
        else:
            conv_out = native.causal_conv1d_fwd(x, conv_w, conv_b, True)
        bsz, d_inner = conv_out.shape[0], conv_out.shape[1]
        keep_stores = need_ckpt or os.environ.get("DIMSUM_SCAN_INFER_STORES", "0") == "1"      # (see `keep` below)
        # layouts chosen like the reference (:622-626): the GEMM writes delta d-major so that it needs no transpose
        conv_rows = _rows(conv_out)
        if not need_ckpt and B_proj_bias is None and C_proj_bias is None and conv_rows.stride(1) == 1:
            # inference: x_proj written transposed, (R + 2N, b l): its rows are the d-major delta input, B and C as the scan reads
            # them -- (b, 1, N, l) views with strides (l, ., b l, 1) -- without the two transposing copies per mixer
            x_dbl_t = x_proj_weight @ conv_rows                                                          # (R + 2N, b l)
            x_dbl = None
            # dt_proj inside the scan (csrc/ssm_scan_fwd_kernel.hpp, kDt): where the 64-channel kernel serves the launch, delta = W_dt x_dbl[:R] is
            # formed tile by tile on the matrix cores and the (b, d, l) tensor never exists: one GEMM launch and 2 b d l 4 bytes less per mixer.
            # The launch keeps no `out` / `x` stores then (DIMSUM_SCAN_INFER_STORES=1, the reference interface's launch, takes the GEMM).
            # Only under allow_tf32 (the reference's own setting, train.py:20-21): the in-scan product is three bf16 products per fp32 product
            # (2e-5 relative, what the library's TF32-policy GEMM spends) -- with the flag off the reference multiplies in exact fp32 and so
            # does this path (the library's fp32 GEMM), which also keeps bench.py's exact-fp32 reference leg exact.
            dt_fused = (not keep_stores and not torch.is_autocast_enabled("cuda") and os.environ.get("DIMSUM_SCAN_DT_PROJ", "1") != "0"
                        and torch.backends.cuda.matmul.allow_tf32
                        and native.scan_dt_proj_supported(conv_out, z, A, delta_proj_weight, x_dbl_t[:R]))
            delta = None if dt_fused else (delta_proj_weight @ x_dbl_t[:R]).view(d_inner, bsz, L).permute(1, 0, 2)
            Bm = x_dbl_t[R:R + N].view(N, bsz, L).permute(1, 0, 2).unsqueeze(1)
            Cm = x_dbl_t[R + N:].view(N, bsz, L).permute(1, 0, 2).unsqueeze(1)
        elif need_ckpt and B_proj_bias is None and C_proj_bias is None and conv_rows.stride(1) == 1 and xz.is_cuda \
                and not torch.is_autocast_enabled("cuda") and os.environ.get("DIMSUM_XDBL_T_TRAIN", "1") != "0":
            # training, the same transposed x_proj: B and C are read in place by both scan kernels, and the backward writes dB / dC straight into
            # the rows of d x_dbl^T -- 7 transposing / slicing copies per mixer less than the (b l, R + 2N) layout of the reference (:840-860)
            x_dbl = x_dbl_t = x_proj_weight @ conv_rows                                                  # (R + 2N, b l), saved as x_dbl
            dt_fused = False
            delta = (delta_proj_weight @ x_dbl_t[:R]).view(d_inner, bsz, L).permute(1, 0, 2)
            Bm = x_dbl_t[R:R + N].view(N, bsz, L).permute(1, 0, 2).unsqueeze(1)
            Cm = x_dbl_t[R + N:].view(N, bsz, L).permute(1, 0, 2).unsqueeze(1)
        else:
            x_dbl_t = None
            x_dbl = F.linear(conv_out.transpose(1, 2).reshape(bsz * L, d_inner), x_proj_weight)        # (b l, R + 2N)
            delta = (delta_proj_weight @ x_dbl[:, :R].t()).view(d_inner, bsz, L).permute(1, 0, 2)       # (b, d, l), strides (L, bL, 1)
            Bm = x_dbl[:, R:R + N]
            Cm = x_dbl[:, -N:]
            if B_proj_bias is not None:
                Bm = Bm + B_proj_bias.to(Bm.dtype)
            if C_proj_bias is not None:
                Cm = Cm + C_proj_bias.to(Cm.dtype)
            Bm = Bm.reshape(bsz, L, N).permute(0, 2, 1).unsqueeze(1).contiguous()                       # (b, 1, N, l)
            Cm = Cm.reshape(bsz, L, N).permute(0, 2, 1).unsqueeze(1).contiguous()
        D = D.contiguous() if D is not None else None
        # the tile-boundary states ride along to the backward (one activation tensor): it then needs no sweep of
        # its own to rebuild them. They depend on (conv_out, delta, A, B) only, which the backward recomputes identically.
        need = need_ckpt        # decided by the caller (grad mode is off in here)
        # `out` (ungated) and the chunk states `x` only serve the backward. The reference's kernel always stores them
        # (selective_scan_fwd_kernel.cuh:239-254); here an inference call (nothing to differentiate) passes NULL out_ptr / x_ptr and the
        # kernel skips both stores: 1.082 instead of 1.384 GB per launch at DiM-L/2, batch 256 (SURVEY 8(d)'s "inference-only lower bound"),
        # same arithmetic, bit-identical out_z. DIMSUM_SCAN_INFER_STORES=1 restores the reference interface's stores (bench.py prices
        # that launch too, as `roofline_full_interface`).
        keep = need or keep_stores
        # inference under allow_tf32: out_z leaves the scan as its split-bf16 pair of planes (the same 4 bytes per element) and out_proj runs
        # on the hand-written kernel's transposing-read variant straight from them (0.26 -> 0.17 ms per mixer at 65536 tokens)
        scan_k = native.scan_fwd_kernel_for(bsz, d_inner, L, N, Bm.shape[1]) if xz.is_cuda else 1
        # ... under the scaled-fp16 policy where a state-split kernel serves the launch (fp32 out_z: 16 channels per wave, no wave sees a 64-channel
        # block): ONE fp16 product behind a conversion pass that builds the block-scaled image (gemm.out_proj_f16_convert_enabled); takes precedence
        # over the planes (three products): 22.97 -> 22.77 ms per DiM-L/2 forward at batch 32, 176.6 -> 170.6 ms at DiM-XL/2 512 px (whose d_model
        # 576 the planes' TN product does not take)
        conv16 = (has_out_proj and not keep and out_proj_bias is None and xz.is_cuda and not torch.is_autocast_enabled("cuda")
                  and gemm.out_proj_f16_convert_enabled(xz, out_proj_weight, bsz * L, L, scan_k))
        planes = (has_out_proj and not need and not conv16 and out_proj_bias is None and L % 8 == 0
                  and d_inner % 64 == 0 and xz.is_cuda
                  and gemm.out_proj_planes_enabled(xz, out_proj_weight, bsz * L, scan_k))
        # inference under the scaled-fp16 policy on the 64-channel kernel: out_z leaves the scan as block-scaled fp16 (half the bytes) and
        # out_proj is ONE fp16 product per element on the hand-written TN GEMM (gemm.out_proj_f16) instead of the library's fp32 GEMM
        z16 = (has_out_proj and not keep and not planes and not conv16 and out_proj_bias is None and xz.is_cuda and not torch.is_autocast_enabled("cuda")
               and native.scan_out_z_f16_supported(conv_out, z, A, Bm.shape[1])
               and gemm.out_proj_f16_enabled(xz, out_proj_weight, bsz * L, L, scan_k))
        out, scan_x, out_z, *rest = native.selective_scan_fwd(conv_out, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus,
                                                              need_out=keep, need_x=keep, need_ckpt=need, **({"out_z_planes": True} if planes else {}),
                                                              **({"out_z_f16": True} if z16 else {}),
                                                              **({"dt_proj": (delta_proj_weight, x_dbl_t[:R])} if delta is None else {}))
        if z16:
            return gemm.out_proj_f16(out_z[0], out_z[1], out_proj_weight).view(bsz, L, out_proj_weight.shape[0])
        if conv16:
            oz_rows = _rows(out_z)
            if oz_rows.stride(1) == 1 and oz_rows.stride(0) % 4 == 0:
                img, tab = native.rows_block_f16s(oz_rows)
                return gemm.out_proj_f16(img, tab, out_proj_weight).view(bsz, L, out_proj_weight.shape[0])
        if planes:
            return gemm.out_proj_planes(out_z, out_proj_weight).view(bsz, L, out_proj_weight.shape[0])
        ckpt = rest[0] if need else None
        ctx.delta_softplus, ctx.has_out_proj, ctx.checkpoint_lvl = delta_softplus, has_out_proj, checkpoint_lvl
        ctx.xdbl_t = x_dbl is not None and x_dbl is x_dbl_t       # (training: x_dbl is saved as its transpose)
        ctx.flags = (conv1d_bias is not None, D is not None, delta_bias is not None, B_proj_bias is not None,
                     C_proj_bias is not None, has_out_proj and out_proj_bias is not None)
        if checkpoint_lvl >= 1:
            conv_out, delta = None, None            # recomputed in the backward (:663-664)
        keep_out_z = has_out_proj and need and os.environ.get("DIMSUM_RECOMPUTE_OUT_Z", "0") != "1"
        f16s = bool(f16s_train) and has_out_proj and need and out_proj_bias is None and keep_out_z
        ctx.f16s = f16s
        if f16s:
            # out = out_z^T W_out^T as the TN product of two images whose rows are the reduction index (channels): out_z (d, b l) with one scale per
            # channel, W_out^T (d, e) with one per channel -> per-reduction-row factors. The image replaces the fp32 out_z among the saved tensors
            # (d out_proj.weight reads it again): half the bytes kept per mixer.
            oz16 = native.rows_f16s(_rows(out_z))
            wt16 = gemm.weight_t_f16s_train(out_proj_weight)
            y = native.gemm_tn(oz16.data, wt16.data, row_invs=(oz16.inv, wt16.inv))
            ctx.save_for_backward(xz, conv_w, conv_b, x_dbl, x_proj_weight, delta_proj_weight, out_proj_weight, conv_out, delta, A, Bm, Cm, D, delta_bias,
                                  scan_x, out, ckpt, None, oz16.data, oz16.inv, wt16.data, wt16.inv)
            return y.view(bsz, L, out_proj_weight.shape[0])
        ctx.save_for_backward(xz, conv_w, conv_b, x_dbl, x_proj_weight, delta_proj_weight,
                              out_proj_weight if has_out_proj else None, conv_out, delta, A, Bm, Cm, D, delta_bias, scan_x, out, ckpt,
                              out_z if keep_out_z else None, None, None, None, None)
        if not has_out_proj:
            return out_z                                                                                # (b, d, l)
        if out_proj_bias is None:
            return gemm.linear(out_z.transpose(1, 2), out_proj_weight)                                   # (b, l, d_model)
        return F.linear(out_z.transpose(1, 2), out_proj_weight, out_proj_bias)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        (xz, conv_w, conv_b, x_dbl, x_proj_weight, delta_proj_weight, out_proj_weight, conv_out, delta, A, Bm, Cm, D,
         delta_bias, scan_x, out, ckpt, kept_out_z, oz16_data, oz16_inv, wt16_data, wt16_inv) = ctx.saved_tensors
        has_conv_b, has_D, has_dbias, has_Bb, has_Cb, has_ob = ctx.flags
        L = xz.shape[-1]
        R = delta_proj_weight.shape[1]
        N = A.shape[-1]
        x, z = xz.chunk(2, dim=1)
        bsz, d_inner = x.shape[0], x.shape[1]
        dout = _last_contig(dout)
        if ctx.checkpoint_lvl == 1:
            conv_out = native.causal_conv1d_fwd(x, conv_w, conv_b, True)         # the NON-cond entry, as in :929
            delta = (delta_proj_weight @ (x_dbl[:R] if ctx.xdbl_t else x_dbl[:, :R].t())).view(d_inner, bsz, L).permute(1, 0, 2)
        dxz = torch.empty_like(xz)
        dx, dz = dxz.chunk(2, dim=1)
        dout16 = None
        if ctx.f16s:
            # dout_y (d, b l) = W_out^T dout^T as an NT product of the images of W_out^T (one scale per channel = output row) and dout (one per token)
            dout16 = native.rows_f16s(dout.reshape(bsz * L, -1))
            dout_y = native.gemm_nt(wt16_data, dout16.data, scales=(wt16_inv, dout16.inv)).view(d_inner, bsz, L).permute(1, 0, 2)
        elif ctx.has_out_proj:
            dout2 = dout.reshape(bsz * L, -1).t()                                                       # "b l e -> e (b l)"
            dout_y = (out_proj_weight.t() @ dout2).view(d_inner, bsz, L).permute(1, 0, 2)               # d-major like delta
        else:
            dout_y = dout
        recompute = ctx.has_out_proj and kept_out_z is None and not ctx.f16s     # only d out_proj.weight needs out_z
        dx_dbl = torch.empty_like(x_dbl)
        into = {}
        if ctx.xdbl_t:      # dB / dC land in rows R .. R + 2N of d x_dbl^T (R + 2N, b l)
            into = {"dB": dx_dbl[R:R + N].view(N, bsz, L).permute(1, 0, 2).unsqueeze(1), "dC": dx_dbl[R + N:].view(N, bsz, L).permute(1, 0, 2).unsqueeze(1)}
        dconv_out, ddelta, dA, dB, dC, dD, ddelta_bias, dz, *rest = native.selective_scan_bwd(
            conv_out, delta, A, Bm, Cm, D, z, delta_bias, dout_y, scan_x, out, dz, ctx.delta_softplus, recompute, ckpt=ckpt, **into)
        out_z = rest[0] if recompute else kept_out_z
        dout_proj_weight = dout_proj_bias = None
        if ctx.f16s:
            # "eB,dB->ed" as the mixed-layout product out_z (d, B) dout (B, e): the saved image's rows run along the reduction, dout's over it
            dout_proj_weight = native.gemm_nn(oz16_data, oz16_inv, dout16.data, dout16.inv).t()
        elif ctx.has_out_proj:
            # "eB,dB->ed": a (d_model, d_inner) output over a b*l-long reduction -- sliced, it would fill 8 of 256 CUs otherwise
            dout_proj_weight = gemm.mm_nn_rows(_rows(out_z), dout.reshape(bsz * L, -1)).t()
        if ctx.has_out_proj:
            dout_proj_bias = dout.sum(dim=(0, 1)) if has_ob else None
        if ctx.xdbl_t:
            ddelta2, conv_rows = _rows(ddelta), _rows(conv_out)                                         # (d, b l) views
            conv_rows = conv_rows if conv_rows.stride(1) == 1 else conv_rows.contiguous()
            ddelta_proj_weight = gemm.mm_nt_rows(ddelta2, x_dbl[:R])                                    # "dB,rB->dr", sliced reduction
            torch.mm(delta_proj_weight.t(), ddelta2, out=dx_dbl[:R])                                    # "dr,dB->rB"
            dx_proj_weight = gemm.mm_nt_rows(dx_dbl, conv_rows)                                         # "rB,dB->rd", sliced reduction
            dconv2 = _rows(dconv_out)                  # (d, b l) view of the scan's own du: the product is added in place (no copy of the addend)
            dconv2 = dconv2.addmm_(x_proj_weight.t(), dx_dbl) if dconv2.stride(1) == 1 else torch.addmm(dconv2, x_proj_weight.t(), dx_dbl)
            dconv_out = dconv2.view(d_inner, bsz, L).permute(1, 0, 2)
            _, dconv_w, dconv_b = native.causal_conv1d_bwd(x, conv_w, conv_b, dconv_out, dx, True)
            return (dxz, dconv_w.unsqueeze(1), dconv_b if has_conv_b else None, dx_proj_weight, ddelta_proj_weight,
                    dout_proj_weight, dout_proj_bias, dA, None, None, dD if has_D else None,
                    ddelta_bias if has_dbias else None, None, None, None, None, None, None, None, None, None)
        dBf = dB.squeeze(1).permute(0, 2, 1).reshape(bsz * L, N)                                        # "b 1 n l -> (b l) n"
        dCf = dC.squeeze(1).permute(0, 2, 1).reshape(bsz * L, N)
        dB_proj_bias = dBf.sum(0) if has_Bb else None
        dC_proj_bias = dCf.sum(0) if has_Cb else None
        dx_dbl[:, R:R + N] = dBf
        dx_dbl[:, -N:] = dCf
        ddelta2 = _rows(ddelta)                                                                         # (d, b l)
        ddelta_proj_weight = gemm.mm_nn_rows(ddelta2, x_dbl[:, :R])                                     # "dB,Br->dr", sliced reduction
        dx_dbl[:, :R] = ddelta2.t() @ delta_proj_weight                                                 # "dB,dr->Br"
        dconv2 = _rows(dconv_out)                                                                       # (d, b l)
        dx_proj_weight = gemm.mm_nn_rows(_rows(conv_out), dx_dbl).t()                                   # "Br,Bd->rd", sliced reduction
        dconv2 = torch.addmm(dconv2, x_proj_weight.t(), dx_dbl.t())
        dconv_out = dconv2.view(d_inner, bsz, L).permute(1, 0, 2)
        _, dconv_w, dconv_b = native.causal_conv1d_bwd(x, conv_w, conv_b, dconv_out, dx, True)
        return (dxz, dconv_w.unsqueeze(1), dconv_b if has_conv_b else None, dx_proj_weight, ddelta_proj_weight,
                dout_proj_weight, dout_proj_bias, dA, None, None, dD if has_D else None,
                ddelta_bias if has_dbias else None, dB_proj_bias, dC_proj_bias, None, None, None, None, None, None, None)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A,
                   B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    wb = _will_backprop(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A, B, C, D, delta_bias,
                        B_proj_bias, C_proj_bias, None)
    return _MambaInner.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                             out_proj_bias, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, None, True, _checkpoint_lvl(),
                             wb, False, _f16s_train(xz, out_proj_weight, wb))


def mamba_inner_fn_cond(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                        A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                        delta_softplus=True, init_states=None, conv_done=False):
    wb = _will_backprop(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A, B, C, D, delta_bias,
                        B_proj_bias, C_proj_bias, init_states)
    return _MambaInner.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                             out_proj_bias, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, init_states,
                             True, _checkpoint_lvl(), wb, conv_done, _f16s_train(xz, out_proj_weight, wb))


def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None,
                               D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    return _MambaInner.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, None, None, A, B, C, D,
                             delta_bias, B_proj_bias, C_proj_bias, delta_softplus, None, False, _checkpoint_lvl(),
                             _will_backprop(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, None))


def mamba_inner_fn_no_out_proj_cond(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None,
                                    D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True,
                                    init_states=None):
    return _MambaInner.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, None, None, A, B, C, D,
                             delta_bias, B_proj_bias, C_proj_bias, delta_softplus, init_states, False, _checkpoint_lvl(),
                             _will_backprop(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, init_states))


def bimamba_inner_fn(*args, **kwargs):
    raise NotImplementedError("bimamba_inner_fn is defined by the reference but never called by its modules "
                              "(mamba_simple.py uses two mamba_inner_fn_no_out_proj calls for scan_type='v2')")
