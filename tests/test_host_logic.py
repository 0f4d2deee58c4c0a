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


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout: compile a tiny C program against the header and compare sizeof."""
    import subprocess
    import tempfile
    from dimsum_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = '#include <stdio.h>\n#include "dimsum_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(dimsum_ssm_params_t),sizeof(dimsum_ssm_bwd_params_t),sizeof(dimsum_conv_params_t),' \
          'sizeof(dimsum_conv_bwd_params_t),sizeof(dimsum_norm_params_t),sizeof(dimsum_norm_bwd_params_t),' \
          'sizeof(dimsum_tt_params_t),sizeof(dimsum_xattn_params_t),sizeof(dimsum_xattn_bwd_params_t),sizeof(dimsum_gemm_params_t));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(root, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        sizes = [int(v) for v in subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout.split()]
    mirrors = [_lib.SsmParams, _lib.SsmBwdParams, _lib.ConvParams, _lib.ConvBwdParams, _lib.NormParams, _lib.NormBwdParams,
               _lib.TtParams, _lib.XattnParams, _lib.XattnBwdParams, _lib.GemmParams]
    assert sizes == [ctypes.sizeof(m) for m in mirrors]


def test_field_offsets_of_the_round5_fields_match_header():
    """the fields added in round 5 sit where the header puts them (a swap of two equally sized fields would pass the sizeof test)"""
    import subprocess
    import tempfile
    from dimsum_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [("dimsum_ssm_params_t", _lib.SsmParams, ["dt_w_ptr", "dt_x_ptr", "dt_w_row_stride", "dt_x_row_stride", "dt_rank", "out_z_f16", "out_z_scale_ptr", "out_z_scale_ld"]),
              ("dimsum_xattn_params_t", _lib.XattnParams, ["qkv_f16"]),
              ("dimsum_gemm_params_t", _lib.GemmParams, ["qkv_q_cols", "conv_weight_ptr", "conv_bias_ptr", "conv_rows", "conv_width", "conv_seq", "conv_weight_ld",
                                                         "a_block_inv_ptr", "a_block_inv_ld", "tn_pair_a_cols", "tune_variant", "tune_reserved"])]
    body = "".join(f'printf("%zu ", offsetof({st}, {f}));' for st, _, fs in fields for f in fs)
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "dimsum_hip.h"\nint main(){' + body + 'return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "o.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(root, "include"), os.path.join(d, "o.c"), "-o", os.path.join(d, "o")], check=True)
        offs = [int(v) for v in subprocess.run([os.path.join(d, "o")], capture_output=True, text=True, check=True).stdout.split()]
    assert offs == [getattr(m, f).offset for _, m, fs in fields for f in fs]
    # the job table of dimsum_rows_f16s_multi
    src = '#include <stdio.h>\n#include "dimsum_hip.h"\nint main(){printf("%zu", sizeof(dimsum_f16s_job_t));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "j.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(root, "include"), os.path.join(d, "j.c"), "-o", os.path.join(d, "j")], check=True)
        assert int(subprocess.run([os.path.join(d, "j")], capture_output=True, text=True, check=True).stdout) == ctypes.sizeof(_lib.F16sJob)


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
