"""CPU tests of the host-side (Python) logic of dimsum_amd: scan-order tables (bit-exact vs reference goldens), C-ABI
library exports, loud failure without a GPU."""
import ctypes
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import golden


@pytest.mark.parametrize("kind", ["sweep", "zigma", "jpeg"])
@pytest.mark.parametrize("N", [4, 8, 16, 32])
def test_scan_tables_bit_exact(kind, N):
    from dimsum_amd import scanning_orders as so
    g = golden("perm_tables")
    paths = np.stack(so.SCAN_ZOO[kind](N))
    assert paths.dtype == np.int64
    assert np.array_equal(paths, g[f"{kind}{N}"].astype(np.int64))
    assert hashlib.sha256(paths.tobytes()).hexdigest()[:16] == str(g[f"sha_{kind}{N}"])
    inv = np.stack([so.reverse_permut_np(p) for p in paths])
    assert np.array_equal(inv, g[f"{kind}{N}_inv"].astype(np.int64))


def test_block_and_local_tables():
    from dimsum_amd import scanning_orders as so
    g = golden("block_orders")
    for H in (4, 16, 32):
        for r in (0, 1):
            for t in (0, 1):
                for c in (0, 1):
                    assert np.array_equal(so.block_order_table(H, r, t, c), g[f"H{H}_r{r}_t{t}_c{c}"])
    g = golden("perm_tables")
    for (H, w) in ((4, 2), (16, 4), (32, 8), (8, 2)):
        for cf in (False, True):
            tab = so.local_scan_table(H, w, cf)
            assert np.array_equal(tab, g[f"local_H{H}_w{w}_{'col' if cf else 'row'}"])
            x = torch.arange(H * H, dtype=torch.float32).view(1, -1, 1)
            assert torch.equal(so.local_reverse(so.local_scan(x, w, H, H, column_first=cf), w, H, H, column_first=cf), x)
    # compose(first, then): gather twice == gather once
    a, b = so.block_order_table(16, 1, 1, 1), so.local_scan_table(16, 4, True)
    x = np.arange(256) * 3
    assert np.array_equal(x[a][b], x[so.compose(a, b)])


def test_library_exports_every_declared_symbol():
    """The C-ABI library must load without a GPU and export every symbol of include/dimsum_hip.h."""
    from dimsum_amd import _lib
    lib = _lib.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "dimsum_hip.h")).read()
    import re
    declared = set(re.findall(r"\b(dimsum_[a-z0-9_]+)\s*\(", header)) - {"dimsum_hip"}
    assert set(_lib.EXPORTS) == declared, (set(_lib.EXPORTS) ^ declared)
    for name in declared:
        assert hasattr(lib, name), f"libdimsum_hip.so does not export {name}"
    assert lib.dimsum_target_arch() == b"gfx950"
    assert lib.dimsum_status_string(3).decode().startswith("unsupported shape")


_STRUCTS = [("dimsum_ssm_params_t", "SsmParams"), ("dimsum_ssm_ext_t", "SsmExt"), ("dimsum_ssm_bwd_params_t", "SsmBwdParams"),
            ("dimsum_conv_params_t", "ConvParams"), ("dimsum_conv_bwd_params_t", "ConvBwdParams"), ("dimsum_norm_params_t", "NormParams"),
            ("dimsum_norm_bwd_params_t", "NormBwdParams"), ("dimsum_tt_params_t", "TtParams"), ("dimsum_xattn_params_t", "XattnParams"),
            ("dimsum_xattn_bwd_params_t", "XattnBwdParams"), ("dimsum_gemm_params_t", "GemmParams"), ("dimsum_gemm_ext_t", "GemmExt"),
            ("dimsum_f16s_job_t", "F16sJob")]


def _header_layout(pairs):
    """{c struct: (sizeof, {field: offset})} from a C program compiled against include/dimsum_hip.h; pairs: [(c struct, ctypes mirror)]"""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = ""
    for cname, mirror in pairs:
        body += f'printf("{cname} %zu", sizeof({cname}));'
        body += "".join(f'printf(" {f}=%zu", offsetof({cname}, {f}));' for f, _ in mirror._fields_)
        body += 'printf("\\n");'
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "dimsum_hip.h"\nint main(){' + body + 'return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(root, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        out = subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout
    layout = {}
    for line in out.strip().splitlines():
        name, size, *fields = line.split()
        layout[name] = (int(size), {f.split("=")[0]: int(f.split("=")[1]) for f in fields})
    return layout


def test_struct_layouts_match_header():
    """ctypes mirrors must have the C layout: sizeof AND the offset of every field of every struct (a swap of two equally sized fields
    would pass a sizeof test), from a C program compiled against the header; every struct starts with struct_size"""
    from dimsum_amd import _lib
    pairs = [(c, getattr(_lib, m)) for c, m in _STRUCTS]
    layout = _header_layout(pairs)
    for cname, mirror in pairs:
        size, offs = layout[cname]
        assert size == ctypes.sizeof(mirror), cname
        assert offs == {f: getattr(mirror, f).offset for f, _ in mirror._fields_}, cname
        if cname != "dimsum_f16s_job_t":
            assert mirror._fields_[0][0] == "struct_size" and getattr(mirror, "struct_size").offset == 0, cname
            assert mirror().struct_size == size, cname                       # filled in on construction


def test_documented_stub_matches_the_header():
    """INTEGRATION.md section 2 shows the ctypes stub a maintainer of the reference would write. The block between its begin / end markers
    is executed as it stands and its two structs are compared with the header: sizeof and every field offset. (Round 5 shipped a stub that
    stopped 8 fields short of the struct it described.)"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"# --- documented stub: begin ---\n(.*?)# --- documented stub: end ---", text, re.S)
    assert m, "INTEGRATION.md lost its documented stub"
    ns = {}
    exec(m.group(1), ns)
    pairs = [("dimsum_ssm_params_t", ns["SsmParams"]), ("dimsum_ssm_ext_t", ns["SsmExt"])]
    layout = _header_layout(pairs)
    for cname, mirror in pairs:
        size, offs = layout[cname]
        assert size == ctypes.sizeof(mirror), cname
        assert offs == {f: getattr(mirror, f).offset for f, _ in mirror._fields_}, cname
    # every field of the header's two structs is in the stub (the offsets above only cover what the stub names)
    header = open(os.path.join(root, "include", "dimsum_hip.h")).read()
    for cname, mirror in pairs:
        body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} " + cname + ";", header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                names += [re.sub(r"[\s\*]", "", part).split(" ")[-1] for part in re.sub(r"^(const\s+)?[A-Za-z_0-9]+\s+", "", decl).split(",")]
        assert names == [f for f, _ in mirror._fields_], (cname, names)
    # and the stub binds: the library loads without a GPU and a stale size is refused before anything is read (no device call is made)
    lib = ns["bind"](os.path.join(root, "dimsum_amd", "lib", "libdimsum_hip.so"))
    p = ns["SsmParams"]()
    p.struct_size = ctypes.sizeof(ns["SsmParams"]) - 8                       # a caller built against an older, shorter header
    assert lib.dimsum_ssm_scan_fwd(p, None) == 7
    assert "struct_size" in lib.dimsum_status_string(7).decode()


def test_stale_or_foreign_structs_are_refused_not_read():
    """every entry point checks struct_size first (DIMSUM_ERR_ABI = 7): a zeroed struct of the wrong size never reaches a pointer check;
    an extension larger than the library's is refused, a shorter one reads its missing tail as zeros. CPU-only: nothing is launched."""
    from dimsum_amd import _lib
    lib = _lib.load()
    for fn, mirror in ((lib.dimsum_ssm_scan_fwd, _lib.SsmParams), (lib.dimsum_ssm_scan_bwd, _lib.SsmBwdParams),
                       (lib.dimsum_causal_conv1d_fwd, _lib.ConvParams), (lib.dimsum_causal_conv1d_bwd, _lib.ConvBwdParams),
                       (lib.dimsum_norm_fwd, _lib.NormParams), (lib.dimsum_norm_bwd, _lib.NormBwdParams),
                       (lib.dimsum_token_transform, _lib.TtParams), (lib.dimsum_xattn_fusion_fwd, _lib.XattnParams),
                       (lib.dimsum_xattn_fusion_bwd, _lib.XattnBwdParams), (lib.dimsum_gemm_nt, _lib.GemmParams)):
        P = mirror()
        assert fn(P, None) != 7, mirror                                      # the right size gets past the check (and fails on its NULL pointers)
        for bad in (0, ctypes.sizeof(mirror) - 8, ctypes.sizeof(mirror) + 8):
            P.struct_size = bad
            assert fn(P, None) == 7, (mirror, bad)
    P = _lib.GemmParams()
    P.struct_size = 0
    assert lib.dimsum_gemm_tn(P, 1, 0, None) == 7 and lib.dimsum_gemm_nt_kernel_for(P) == -7
    # extensions: the dispatch query is a pure host function of (*p, *p->ext)
    P = _lib.SsmParams()
    P.batch, P.dim, P.seqlen, P.dstate, P.n_groups, P.n_chunks = 256, 1024, 256, 16, 1, 1
    assert lib.dimsum_ssm_scan_fwd_variant(P) == 1                           # ext = NULL: the reference interface, automatic dispatch
    E = _lib.attach_ext(P, _lib.SsmExt)
    E.kernel_variant = 4
    assert lib.dimsum_ssm_scan_fwd_variant(P) == 4
    E.struct_size = 8                                                        # a caller whose dimsum_ssm_ext_t ends after kernel_variant
    assert lib.dimsum_ssm_scan_fwd_variant(P) == 4
    E.struct_size = 4                                                        # ... and one that only knows struct_size: the rest reads as 0
    assert lib.dimsum_ssm_scan_fwd_variant(P) == 1
    E.struct_size = ctypes.sizeof(_lib.SsmExt) + 8                           # a caller that knows MORE than this library
    assert lib.dimsum_ssm_scan_fwd_variant(P) == -1 and lib.dimsum_ssm_scan_fwd(P, None) == 7


def test_ops_fail_loudly_without_gpu():
    from dimsum_amd import native
    from dimsum_amd.ops import causal_conv1d_fn, rms_norm_fn, selective_scan_fn
    u = torch.randn(1, 4, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        selective_scan_fn(u, u, -torch.rand(4, 8), torch.randn(1, 8, 8), torch.randn(1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        causal_conv1d_fn(u, torch.randn(4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rms_norm_fn(torch.randn(3, 16), torch.ones(16), None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.selective_scan_fwd(u, u, -torch.rand(4, 8), torch.randn(1, 1, 8, 8), torch.randn(1, 1, 8, 8), None, None, None, True)


def test_split3_carrier_is_never_enabled_off_gpu_or_under_autograd():
    """gemm.split3_enabled: the operand-image carrier serves inference launches of fp32 CUDA tensors under allow_tf32 only"""
    import torch
    from dimsum_amd import gemm
    x, w = torch.zeros(4, 8), torch.zeros(8, 8)
    old = torch.backends.cuda.matmul.allow_tf32
    try:
        torch.backends.cuda.matmul.allow_tf32 = True
        with torch.no_grad():
            assert not gemm.split3_enabled(x, w)                    # CPU tensors
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old


def test_gemm_tail_fallback_semantics():
    """gemm._tail = the unfused form of the GEMM's gate + residual epilogue (what runs when a shape does not fit the kernel's tiling):
    residual + gate[row // rows_per_batch] * (y + bias), every optional part omitted independently"""
    import torch
    from dimsum_amd import gemm
    g = torch.Generator().manual_seed(0)
    B, L, N = 3, 5, 8
    y, res = torch.randn(B * L, N, generator=g), torch.randn(B * L, N, generator=g)
    bias, gate = torch.randn(N, generator=g), torch.randn(B, N, generator=g)
    want = res + gate.repeat_interleave(L, 0) * (y + bias)
    assert torch.allclose(gemm._tail(y, bias, res, gate, L), want)
    assert torch.allclose(gemm._tail(y, None, res, None, None), res + y)
    assert torch.allclose(gemm._tail(y, bias, None, None, None), y + bias)
    assert gemm._tail(y, None, None, None, None) is y


def test_f16image_quacks_like_its_tensor():
    import torch
    from dimsum_amd.native import F16Image
    img = F16Image(torch.zeros(2, 6, 8, dtype=torch.float16), torch.ones(2, 6))
    flat = img.reshape(12, -1)
    assert tuple(flat.shape) == (12, 8) and tuple(flat.inv.shape) == (12,) and tuple(img.shape) == (2, 6, 8)
    assert tuple(flat.view(2, 6, 8).inv.shape) == (2, 6) and img.float().shape == (2, 6, 8)


# ---- host-side pieces of the round-6 training path that run without a GPU ---------------------------------------------------------------
def test_sliced_reduction_products_match_plain_matmul():
    """gemm.mm_nn_rows / mm_nt_rows (the Mamba projections' weight gradients as sliced reductions over b * l): the CPU route is the plain product"""
    import torch
    from dimsum_amd import gemm
    g = torch.Generator().manual_seed(0)
    a, b, c = torch.randn(24, 4096, generator=g), torch.randn(4096, 16, generator=g), torch.randn(16, 4096, generator=g)
    assert torch.allclose(gemm.mm_nn_rows(a, b), a @ b, rtol=1e-5, atol=1e-4)
    assert torch.allclose(gemm.mm_nt_rows(a, c), a @ c.t(), rtol=1e-5, atol=1e-4)


def test_fork_halves_joins_three_gradients():
    """models_dim._ForkHalves: (x1, x2, h) = halves + the stream itself; its backward = cat(d1, d2) + dres, also with a missing branch"""
    import torch
    from dimsum_amd.models_dim import _ForkHalves
    g = torch.Generator().manual_seed(1)
    h = torch.randn(2, 6, 8, generator=g, dtype=torch.float64, requires_grad=True)
    w1, w2, w3 = (torch.randn(s, generator=g, dtype=torch.float64) for s in ((2, 6, 4), (2, 6, 4), (2, 6, 8)))
    x1, x2, hs = _ForkHalves.apply(h)
    ((x1 * w1).sum() + (x2 * w2).sum() + (hs * w3).sum()).backward()
    assert torch.equal(h.grad, torch.cat((w1, w2), dim=-1) + w3)
    h2 = h.detach().clone().requires_grad_()
    x1, x2, hs = _ForkHalves.apply(h2)
    ((x1 * w1).sum() + (x2 * w2).sum()).backward()
    assert torch.equal(h2.grad, torch.cat((w1, w2), dim=-1))


def test_emulated_tf32_attention_core_is_the_reference_math():
    """utils/tf32_emulation._EmuXattnFn (what emulated_tf32() puts in place of the attention kernels) in float64, where rounding is the identity:
    output and all four gradients = torch autograd through scaled_dot_product_attention (attention_fusion.py:44-79), cross and self attention"""
    import torch
    from dimsum_amd.utils.tf32_emulation import _EmuXattnFn
    F = torch.nn.functional.scaled_dot_product_attention
    B, L, heads, hd = 2, 24, 2, 8
    W = 3 * heads * hd
    g = torch.Generator().manual_seed(2)
    q1, q2 = torch.randn(B, L, W, generator=g, dtype=torch.float64), torch.randn(B, L, W, generator=g, dtype=torch.float64)
    b1, b2 = torch.randn(W, generator=g, dtype=torch.float64), torch.randn(W, generator=g, dtype=torch.float64)
    dout = torch.randn(B, L, 2 * heads * hd, generator=g, dtype=torch.float64)
    split = lambda t, bb: (t + bb).reshape(B, L, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
    a = [t.clone().requires_grad_() for t in (q1, q2, b1, b2)]
    o = _EmuXattnFn.apply(*a, heads)
    o.backward(dout)
    r = [t.clone().requires_grad_() for t in (q1, q2, b1, b2)]
    (qa, ka, va), (qb, kb, vb) = split(r[0], r[2]), split(r[1], r[3])
    ref = torch.cat((F(qa, kb, vb).transpose(1, 2).reshape(B, L, -1), F(qb, ka, va).transpose(1, 2).reshape(B, L, -1)), dim=-1)
    ref.backward(dout)
    assert torch.allclose(o, ref, rtol=0, atol=1e-12)
    for x, y in zip(a, r):
        assert torch.allclose(x.grad, y.grad, rtol=0, atol=1e-11)
    a = [q1.clone().requires_grad_(), b1.clone().requires_grad_()]
    o = _EmuXattnFn.apply(a[0], None, a[1], None, heads)
    o.backward(dout[..., :heads * hd])
    r = [q1.clone().requires_grad_(), b1.clone().requires_grad_()]
    qa, ka, va = split(r[0], r[1])
    ref = F(qa, ka, va).transpose(1, 2).reshape(B, L, -1)
    ref.backward(dout[..., :heads * hd])
    assert torch.allclose(o, ref, rtol=0, atol=1e-12) and all(torch.allclose(x.grad, y.grad, rtol=0, atol=1e-11) for x, y in zip(a, r))
