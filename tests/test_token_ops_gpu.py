"""GPU parity: fused token-transform kernel and gated GeLU (through the C ABI) vs the numpy oracle and the reference
goldens (haar.npz / dct.npz / block_orders). Permutations are pure moves: bit-exact. Transforms: rtol 1e-5 + 1e-6 * max."""
import numpy as np
import pytest
import torch

from conftest import assert_close, golden

pytestmark = pytest.mark.gpu


def _g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("H", [16, 32, 4])
def test_haar_vs_golden(H):
    from dimsum_amd import native
    g = golden("haar")
    y = native.token_transform(_g(g[f"H{H}_x"]), "haar", True)
    assert_close(y.cpu().numpy(), g[f"H{H}_dwt"], 1e-5, 0, "dwt", scale_atol=1e-6)
    xi = native.token_transform(_g(g[f"H{H}_y2"]), "haar", False)
    assert_close(xi.cpu().numpy(), g[f"H{H}_idwt"], 1e-5, 0, "idwt", scale_atol=1e-6)
    rt = native.token_transform(native.token_transform(_g(g[f"H{H}_x"]), "haar", True), "haar", False)
    assert_close(rt.cpu().numpy(), g[f"H{H}_x"], 1e-5, 0, "roundtrip", scale_atol=1e-6)


@pytest.mark.parametrize("H", [16, 32])
def test_dct_vs_golden(H):
    from dimsum_amd import native
    g = golden("dct")
    y = native.token_transform(_g(g[f"H{H}_x"]), "dct", True)
    assert_close(y.cpu().numpy(), g[f"H{H}_dct"], 1e-5, 0, "dct", scale_atol=2e-6)
    back = native.token_transform(y, "dct", False)
    assert_close(back.cpu().numpy(), g[f"H{H}_roundtrip"], 1e-5, 0, "idct", scale_atol=2e-6)


@pytest.mark.parametrize("C,H,kind", [(512, 16, "haar"), (576, 32, "haar"), (192, 16, "haar"), (512, 16, "dct"), (512, 16, "none"), (20, 8, "haar")])
def test_fused_pre_post_vs_oracle(C, H, kind):
    """pre: y = modulate(P(T(x))) on a channel-slice view; post: y = x + T^-1(P^-1(gate*m))."""
    from dimsum_amd import native, scanning_orders as so
    from oracle import np_ops
    rs = np.random.RandomState(C + H)
    B, L = 3, H * H
    full = rs.standard_normal((B, L, 2 * C)).astype(np.float32)
    x = full[:, :, C:]                                     # like x2 = hidden.chunk(2, dim=2)[1]
    shift, scale, gate = (rs.standard_normal((B, 3 * C)).astype(np.float32)[:, i * C:(i + 1) * C] for i in range(3))
    table = so.compose(so.local_scan_table(H, H // 4, True), so.block_order_table(H, True, False, True))
    inv = so.reverse_permut_np(table)
    T = {"haar": np_ops.haar_dwt_tokens, "dct": np_ops.dct_tokens, "none": lambda a: a}[kind]
    Ti = {"haar": np_ops.haar_idwt_tokens, "dct": np_ops.idct_tokens, "none": lambda a: a}[kind]
    full_g = _g(full)
    mods = _g(np.concatenate([shift, scale, gate], 1))     # one (B, 3C) adaLN output, sliced like .chunk(3, dim=1)
    sh_g, sc_g, ga_g = mods[:, :C], mods[:, C:2 * C], mods[:, 2 * C:]
    inv32 = torch.from_numpy(inv.astype(np.int32)).cuda()
    y = native.token_transform(full_g[:, :, C:], kind, True, out_index=inv32, scale=sc_g, shift=sh_g)
    ref = T(np.ascontiguousarray(x))[:, table] * (1 + scale[:, None]) + shift[:, None]
    assert_close(y.cpu().numpy(), ref, 1e-5, 0, "pre", scale_atol=1e-6)
    m = rs.standard_normal((B, L, C)).astype(np.float32)
    out = native.token_transform(_g(m), kind, False, in_index=inv32, gate=ga_g, residual=full_g[:, :, C:])
    ref2 = x + Ti(np.ascontiguousarray((gate[:, None] * m)[:, inv]))
    assert_close(out.cpu().numpy(), ref2, 1e-5, 0, "post", scale_atol=1e-6)


def test_pure_permutation_is_bit_exact():
    from dimsum_amd import native, scanning_orders as so
    x = torch.randn(2, 256, 64, device="cuda")
    for r in (0, 1):
        for t in (0, 1):
            for c in (0, 1):
                tab = so.block_order_table(16, r, t, c)
                inv32 = torch.from_numpy(so.reverse_permut_np(tab).astype(np.int32)).cuda()
                y = native.token_transform(x, "none", True, out_index=inv32)
                assert torch.equal(y, x[:, torch.from_numpy(tab).cuda()])
                back = native.token_transform(y, "none", False, in_index=inv32)
                assert torch.equal(back, x)


@pytest.mark.parametrize("rows,H,with_bias", [(37, 96, False), (37, 96, True), (300, 4096, True), (65, 1032, True)])
def test_gated_gelu(rows, H, with_bias):
    """fused bias + tanh-GELU + gate epilogue of the w12 GEMM, forward and backward (dx12 and the in-kernel d bias column
    sums) vs torch autograd. fp32: rtol 1e-5 + 1e-6 max (fwd), 1e-4 + 1e-5 max (bwd), d bias 2e-4 + 2e-5 max."""
    from dimsum_amd import native
    g = torch.Generator().manual_seed(rows + H)
    x = (torch.randn(rows, 2 * H, generator=g) * 2).cuda()
    bias = torch.randn(2 * H, generator=g).cuda() if with_bias else None
    xr, br = x.detach().clone().requires_grad_(), (bias.detach().clone().requires_grad_() if with_bias else None)
    xb = xr + br if with_bias else xr
    ref = torch.nn.functional.gelu(xb[:, :H], approximate="tanh") * xb[:, H:]
    h = native.gated_gelu_fwd(x, bias)
    assert_close(h.cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, 0, "fwd", scale_atol=1e-6)
    dh = torch.randn(ref.shape, generator=g).cuda()
    ref.backward(dh)
    dx, db = native.gated_gelu_bwd(x, bias, dh)
    assert_close(dx.cpu().numpy(), xr.grad.cpu().numpy(), 1e-4, 0, "bwd", scale_atol=1e-5)
    if with_bias:
        assert_close(db.cpu().numpy(), br.grad.cpu().numpy(), 2e-4, 0, "dbias", scale_atol=2e-5)
    else:
        assert db is None


@pytest.mark.parametrize("with_gate,with_bias", [(True, True), (True, False), (False, True), (False, False)])
def test_gate_residual_autograd(with_gate, with_bias):
    """y = res + gate * (m + bias) as one fused pass, forward and backward, vs the torch expression."""
    from dimsum_amd.ops import token_ops as to
    B, L, C = 3, 64, 512
    torch.manual_seed(5)
    dy = torch.randn(B, L, C, device="cuda")
    outs = []
    for fused in (True, False):
        torch.manual_seed(6)
        res = torch.randn(B, L, C, device="cuda", requires_grad=True)
        m = torch.randn(B, L, C, device="cuda", requires_grad=True)
        mods = torch.randn(B, 3 * C, device="cuda", requires_grad=True)
        bias = torch.randn(C, device="cuda", requires_grad=True)
        gate = mods.chunk(3, dim=1)[2] if with_gate else None
        bb = bias if with_bias else None
        if fused:
            y = to.gate_residual(res, m, gate, bb)
        else:
            t = m if bb is None else m + bb
            y = res + (t if gate is None else gate.unsqueeze(1) * t)
        y.backward(dy)
        outs.append((y.detach(), res.grad, m.grad, mods.grad if with_gate else torch.zeros(1), bias.grad if with_bias else torch.zeros(1)))
    for name, a, b, (rt, sa) in zip(("y", "dres", "dm", "dmods", "dbias"), outs[0], outs[1],
                                    ((2e-6, 2e-6), (0, 0), (2e-6, 2e-6), (1e-4, 1e-5), (1e-4, 1e-5))):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), rt, 0, name, scale_atol=sa)


@pytest.mark.parametrize("C,H,kind", [(512, 16, "haar"), (576, 32, "haar"), (64, 16, "dct"), (512, 16, "none"), (20, 8, "haar")])
def test_pre_post_autograd_vs_torch_expression(C, H, kind):
    """backward of the fused pre/post passes (adjoint = the other fusion with a rescaled gate + in-kernel adaLN
    reductions) vs torch autograd through the plain expression of the same math (token_ops.haar_dwt_tokens, ...).
    fp32 tolerance: rtol 2e-5 + 2e-6 max|ref| on tensors, 1e-4 + 1e-5 max|ref| on the L-token reductions."""
    from dimsum_amd import scanning_orders as so
    from dimsum_amd.ops import token_ops as to
    torch.manual_seed(C + H)
    B, L = 3, H * H
    table = so.compose(so.local_scan_table(H, H // 4, True), so.block_order_table(H, True, False, True))
    inv = so.reverse_permut_np(table)
    tab = {"fwd": torch.from_numpy(table).cuda(), "inv": torch.from_numpy(inv).cuda(), "inv32": torch.from_numpy(inv.astype(np.int32)).cuda()}
    Tf = {"haar": to.haar_dwt_tokens, "dct": to.dct_tokens, "none": lambda a: a}[kind]
    Ti = {"haar": to.haar_idwt_tokens, "dct": to.idct_tokens, "none": lambda a: a}[kind]

    def leafs():
        torch.manual_seed(C + H)
        full = torch.randn(B, L, 2 * C, device="cuda", requires_grad=True)
        mods = torch.randn(B, 3 * C, device="cuda", requires_grad=True)
        m = torch.randn(B, L, C, device="cuda", requires_grad=True)
        return full, mods, m

    dy1, dy2 = torch.randn(B, L, C, device="cuda"), torch.randn(B, L, C, device="cuda")
    grads = []
    for fused in (True, False):
        full, mods, m = leafs()
        x = full[:, :, C:]
        shift, scale, gate = mods.chunk(3, dim=1)
        if fused:
            y1 = to.pre_mixer(x, kind, tab, shift, scale)
            y2 = to.post_mixer(x, m, gate, kind, tab)
        else:
            y1 = to.modulate(Tf(x).index_select(1, tab["fwd"]), shift, scale)
            y2 = x + Ti((gate.unsqueeze(1) * m).index_select(1, tab["inv"]))
        torch.autograd.backward((y1, y2), (dy1, dy2))
        grads.append((y1.detach(), y2.detach(), full.grad, mods.grad, m.grad))
    for name, a, b, (rt, sa) in zip(("pre", "post", "dx", "dmods", "dm"), grads[0], grads[1],
                                    ((2e-5, 2e-6), (2e-5, 2e-6), (2e-5, 2e-6), (1e-4, 1e-5), (2e-5, 2e-6))):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), rt, 0, name, scale_atol=sa)
