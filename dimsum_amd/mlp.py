"""GatedMLP -- dimsum/mlp.py:49-70: w3( act(x W12a + b) * (x W12b + b) ) + b3. With the tanh-GELU the DiM blocks use,
bias + activation + gate run as ONE fused HIP pass over the bias-free w12 GEMM output (csrc/token_transform.hip,
gated GeLU); the w3 bias can be handed to the caller's fused residual pass (`forward_deferred`). Keeping the biases out
of the GEMMs matters on gfx950: hipBLASLt serves bias-free fp32 matmuls under the reference's TF32 policy
(train.py:20-21) with its split-bf16 MFMA path (2.5x the fp32 rate at 4e-6 relative error), but not its bias-epilogue
kernels."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gemm, native


class _GatedGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x12, bias):
        x12 = x12.contiguous()
        ctx.save_for_backward(x12, bias)
        return native.gated_gelu_fwd(x12, bias)

    @staticmethod
    def backward(ctx, dh):
        x12, bias = ctx.saved_tensors
        dx12, dbias = native.gated_gelu_bwd(x12, bias, dh, need_dbias=bias is not None and ctx.needs_input_grad[1])
        return dx12, dbias


def gated_gelu(x12, bias=None):
    """gelu_tanh(x12[..., :H] + bias[:H]) * (x12[..., H:] + bias[H:])"""
    return _GatedGeluFn.apply(x12, bias)


class _ModGatedMlpImagesFn(torch.autograd.Function):
    """m = w3(gated_gelu(w12(modulate(normed, shift, scale)) + b12))  WITHOUT w3's bias, training, with every GEMM of the forward
    and the backward on split-bf16 operand images (gemm.py, split3):
        forward : h3 = image(modulate(normed))   [token_transform y_split3]      x12 = h3 . W12img^T
                  g3 = image(gelu(x12a + b) (x12g + b))  [the GEMM's epilogue]    m = g3 . W3img^T
        backward: dm_w = image_w(dm)                                              dg = dm_w . (W3^T)img^T      dW3 = dm_w^T . g3
                  dx12_w = image_w(gated GeLU backward)  [one pass]               dh = dx12_w . (W12^T)img^T   dW12 = dx12_w^T . h3
                  d normed / d shift / d scale: the pre-mixer adjoints (ops/token_ops.py)
    image = left order [hi | hi | lo], image_w = weight order [hi | lo | hi]; a (M, 3K) image viewed as (3M, K) is the row-stacked
    image of the same matrix, which is what the weight-gradient products (reduction over the M rows) consume. The same three
    products per fp32 product as hipBLASLt's fp32-under-allow_tf32 kernels, on its plain bf16 kernels: measured per GEMM at 65536
    rows (tools/scratch/ksplit_probe3.py): dW12 4.07 -> 3.18 ms, dh 3.47 -> 2.59, dW3 2.80 -> 1.79, dg 1.79 -> 1.34."""

    @staticmethod
    def forward(ctx, normed, shift, scale, w12, b12, w3):
        B, L, H = normed.shape
        M = B * L
        normed = normed if normed.stride(-1) == 1 else normed.contiguous()
        # every image as the pair [hi | lo] where all its consumers are the hand-written kernel (gemm.train_pairs_enabled)
        pairs = gemm.train_pairs_enabled(M, H, w12.shape[0] // 2, w3.shape[0])
        h3 = native.token_transform(normed, "none", True, scale=scale, shift=shift, split3="pair" if pairs else True)   # (B, L, 3H) / pair
        b12f = None if b12 is None else b12.float()
        g3, x12 = gemm.gated_mlp_hidden_split3_train(h3.reshape(M, -1), w12, b12f)                        # (M, 3F) image / pair, (M, 2F) fp32
        m = gemm.linear_split3(g3, w3)                                                                     # (M, H)
        ctx.pairs = pairs and isinstance(g3, native.PairImage)
        ctx.save_for_backward(normed, scale, h3.data if pairs else h3, x12, g3.data if ctx.pairs else g3, w12, b12f, w3)
        if pairs and not ctx.pairs:
            raise RuntimeError("mlp: the pair images need the hand-written GEMM for every product of the MLP")
        return m.view(B, L, w3.shape[0])

    @staticmethod
    def backward(ctx, dm):
        normed, scale, h3, x12, g3, w12, b12f, w3 = ctx.saved_tensors
        B, L, H = normed.shape
        M, F2 = x12.shape
        Fh = F2 // 2
        Ho = w3.shape[0]
        if ctx.pairs:
            P = native.PairImage
            dm_w = native.split3_rows(dm.reshape(M, Ho).contiguous(), left="pair")                         # (M, 2Ho) pair, read in weight order below
            dg = native.gemm_nt(dm_w, native.split3_rows(w3.detach().t().contiguous(), left=True), weight_order=True)            # (M, F)
            dw3 = gemm.mm_tn(dm_w, P(g3)) if ctx.needs_input_grad[5] else None                              # (Ho, F)
            dx12_w, db12 = native.gated_gelu_bwd(x12, b12f, dg, need_dbias=b12f is not None and ctx.needs_input_grad[4], split3="pair")
            dh = native.gemm_nt(dx12_w, native.split3_rows(w12.detach().t().contiguous(), left=True), weight_order=True)          # (M, H)
            dw12 = gemm.mm_tn(dx12_w, P(h3.view(M, 2 * H))) if ctx.needs_input_grad[3] else None           # (2F, H)
        else:
            dm_w = native.split3_rows(dm.reshape(M, Ho).contiguous(), left=False)                              # (M, 3Ho), weight order
            dg = gemm._nt(dm_w, native.split3_rows(w3.detach().t().contiguous(), left=True))                    # (M, F)
            dw3 = gemm.mm_tn(dm_w.view(3 * M, Ho), g3.view(3 * M, Fh), out_dtype=torch.float32) if ctx.needs_input_grad[5] else None   # (Ho, F)
            dx12_w, db12 = native.gated_gelu_bwd(x12, b12f, dg, need_dbias=b12f is not None and ctx.needs_input_grad[4], split3=True)
            dh = gemm._nt(dx12_w, native.split3_rows(w12.detach().t().contiguous(), left=True))                 # (M, H)
            dw12 = gemm.mm_tn(dx12_w.view(3 * M, F2), h3.view(3 * M, H), out_dtype=torch.float32) if ctx.needs_input_grad[3] else None   # (2F, H)
        dh = dh.view(B, L, H)
        dnormed = dshift = dscale = None
        if ctx.needs_input_grad[0] and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            # one pass over (dh, normed): d normed = dh (1 + scale), d scale = sum_t dh normed, d shift = sum_t dh (the reductions see the
            # un-modulated dh, like the gate / bias sums of token_ops._GateResidual)
            dnormed, dscale, _, dshift = native.token_transform(dh, "none", True, scale=scale, w=normed, want_y=True, want_tsum=True)
        elif ctx.needs_input_grad[0]:
            dnormed = native.token_transform(dh, "none", False, gate=1.0 + scale)
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            _, dscale, dshift = native.token_transform(normed, "none", True, w=dh, want_y=False, want_wsum=True)
        if db12 is not None and b12f is not None:
            db12 = db12.to(b12f.dtype)
        return dnormed, dshift, dscale, dw12, db12, dw3


class _ModGatedMlpF16sFn(torch.autograd.Function):
    """the same module function with every GEMM of the forward and the backward as ONE fp16 product per element over scaled-fp16 images (policy
    "f16s", gemm.py; the reference trains under TF32 as well, dimsum/train.py:20-21):
        forward : h16 = image(modulate(normed))  [token_transform y_split3 = 2]     g16, x12 = gate epilogue of h16 . W12_16^T (row scale of g from the
                  epilogue's bound, x12 kept in fp32 for the adjoint)               m = g16 . W3_16^T
        backward: dm16 = image(dm)                dg = dm16 . W3_16                 dW3 = dm16^T g16   [TN, per-reduction-row factors]
                  dx12_16 = image(gated GeLU adjoint)  [one pass, exact row maxima]  dh = dx12_16 . W12_16          dW12 = dx12_16^T h16
                  (dg / dh: gemm.dx_f16s -- the forward's weight images again, reduced over their rows: no transposed weight image)
    Six products, each a third of the three-product carrier's MFMA work; the images are 2 bytes per element instead of 4 (pairs) / 6."""

    @staticmethod
    def forward(ctx, normed, shift, scale, w12, b12, w3):
        B, L, H = normed.shape
        M = B * L
        normed = normed if normed.stride(-1) == 1 else normed.contiguous()
        h16 = native.token_transform(normed, "none", True, scale=scale, shift=shift, split3="f16s")      # data (B, L, H) float16, inv (B, L)
        b12f = None if b12 is None else b12.float()
        if gemm.train_scope_active():          # (DiM.forward's forward_scope built every weight image and the gate's bound in one launch)
            w12_16, bound = gemm.weight_f16s_train(w12), gemm.gated_bound(w12, b12)
        else:
            w12_16, l1 = gemm.weight_f16s_train(w12, want_l1=True)
            bound = torch.cat([l1 * gemm._K10, gemm._absmax(b12f, w12)]).contiguous()
        g16, x12 = native.gemm_nt(h16.data.view(M, H), w12_16.data, bias=b12f, epilogue="gated_f16", scales=(h16.inv.view(M), w12_16.inv), gate_bound=bound, keep_x12=True)
        w3_16 = gemm.weight_f16s_train(w3)
        m = gemm.nt_f16s_any(g16, w3_16)
        # (the weights' forward images serve the backward's input-gradient products too: gemm.dx_f16s reduces over their rows)
        ctx.save_for_backward(normed, scale, h16.data, h16.inv, x12, g16.data, g16.inv, w12_16.data, w12_16.inv, b12f, w3_16.data, w3_16.inv)
        return m.view(B, L, w3.shape[0])

    @staticmethod
    def backward(ctx, dm):
        normed, scale, hd, hi, x12, gd, gi, w12d, w12i, b12f, w3d, w3i = ctx.saved_tensors
        B, L, H = normed.shape
        M = B * L
        Ho = w3d.shape[0]
        F16 = native.F16Image
        dm16 = native.rows_f16s(dm.reshape(M, Ho).contiguous())
        dg = gemm.dx_f16s(dm16, F16(w3d, w3i))                                                                # (M, F)
        dw3 = gemm.dw_f16s(dm16, F16(gd, gi)) if ctx.needs_input_grad[5] else None                            # (Ho, F)
        dx12_16, db12 = native.gated_gelu_bwd(x12, b12f, dg, need_dbias=b12f is not None and ctx.needs_input_grad[4], split3="f16s")
        dh = gemm.dx_f16s(dx12_16, F16(w12d, w12i)).view(B, L, H)                                              # (M, H)
        dw12 = gemm.dw_f16s(dx12_16, F16(hd, hi)) if ctx.needs_input_grad[3] else None                        # (2F, H)
        dnormed = dshift = dscale = None
        if ctx.needs_input_grad[0] and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            dnormed, dscale, _, dshift = native.token_transform(dh, "none", True, scale=scale, w=normed, want_y=True, want_tsum=True)
        elif ctx.needs_input_grad[0]:
            dnormed = native.token_transform(dh, "none", False, gate=1.0 + scale)
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            _, dscale, dshift = native.token_transform(normed, "none", True, w=dh, want_y=False, want_wsum=True)
        if db12 is not None and b12f is not None:
            db12 = db12.to(b12f.dtype)
        return dnormed, dshift, dscale, dw12, db12, dw3


def f16s_mlp_train_ok(mlp, normed):
    """shapes the single-product training MLP takes: the gate epilogue's tiling (M % 256, H % 64, F % 128), the token pass's image (H <= 2048), the
    gated-GeLU adjoint's row kernel (F <= 5120)"""
    H, F2 = mlp.w12.weight.shape[1], mlp.w12.weight.shape[0]
    M = normed.numel() // normed.shape[-1]
    return M % 256 == 0 and H % 64 == 0 and H >= 128 and H <= 2048 and (F2 // 2) % 128 == 0 and F2 // 2 <= 5120 and mlp.w3.weight.shape[0] % 4 == 0


def mod_gated_mlp_images(mlp, normed, shift, scale):
    """-> (m, b): mlp(modulate(normed, shift, scale)) = m + b for a fused GatedMLP, all GEMMs on operand images (training): scaled-fp16 images
    (one product) under the "f16s" policy, split-bf16 images (three products) otherwise"""
    if gemm.split3_train_enabled(normed, mlp.w12.weight) == "f16s" and f16s_mlp_train_ok(mlp, normed):
        return _ModGatedMlpF16sFn.apply(normed, shift, scale, mlp.w12.weight, mlp.w12.bias, mlp.w3.weight), mlp.w3.bias
    return _ModGatedMlpImagesFn.apply(normed, shift, scale, mlp.w12.weight, mlp.w12.bias, mlp.w3.weight), mlp.w3.bias


class GatedMLP(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=F.gelu, drop=0.0, bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)
        self.act_layer = act_layer()
        self._fused = isinstance(self.act_layer, nn.GELU) and self.act_layer.approximate == "tanh"

    def forward_deferred(self, x, x3=None, residual=None, gate=None):
        """-> (y, b): the module's output is y + b; b (w3's bias or None) is left to the caller's fused residual pass.
        x3: the caller's producer kernel already wrote x as a split-bf16 operand image (gemm.py, split3): both GEMMs then
        run as plain bf16 GEMMs over the hi / lo images, the gated GeLU writes the w3 GEMM's image directly.
        residual (B, L, H) [, gate (B, H)] (with x3): the block's residual tail rides in the w3 GEMM's epilogue -- the call returns
        (residual + gate * (mlp(x) + b3), None)."""
        if x3 is not None:
            b12 = self.w12.bias
            h3 = gemm.gated_mlp_hidden_split3(x3, self.w12.weight, None if b12 is None else b12.float())
            if residual is not None:
                H = self.w3.weight.shape[0]
                b3 = None if self.w3.bias is None else self.w3.bias.float()
                y = gemm.linear_split3(h3, self.w3.weight, bias=b3, residual=residual.reshape(-1, H), gate=gate, rows_per_batch=residual.shape[-2])
                return y.view(residual.shape), None
            return gemm.linear_split3(h3, self.w3.weight).view(*x.shape[:-1], self.w3.weight.shape[0]), self.w3.bias
        if self._fused and x.dtype == torch.float32:
            b12 = self.w12.bias
            h = gated_gelu(gemm.linear(x, self.w12.weight), None if b12 is None else b12.float())
        else:
            x1, x2 = self.w12(x).chunk(2, dim=-1)
            h = self.act_layer(x1) * x2
        return gemm.linear(h, self.w3.weight), self.w3.bias

    def forward(self, x):
        y, b = self.forward_deferred(x)
        return y if b is None else y + b
