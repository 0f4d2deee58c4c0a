"""times (library, fp32 under allow_tf32) of every matmul of one Mamba mixer's training forward + backward at DiM-L/2 batch 256 shapes"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dimsum_amd import gemm
torch.backends.cuda.matmul.allow_tf32 = True
M, dm, D, R, N = 65536, 512, 1024, 32, 16
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x, Win, Wx, Wdt, Wout = r(M, dm), r(2 * D, dm), r(R + 2 * N, D), r(D, R), r(dm, D)
conv_out, out_z, dxz, ddelta, dconv = r(D, M), r(D, M), r(2 * D, M), r(D, M), r(D, M)
x_dbl, dx_dbl, dout = r(M, R + 2 * N), r(M, R + 2 * N), r(M, dm)
ops = {
 "F1 in_proj   W_in @ x^T -> (2D, M)": lambda: Win @ x.t(),
 "F2 x_proj    linear(conv_out^T, W_x) -> (M, 64)": lambda: torch.nn.functional.linear(conv_out.t(), Wx),
 "F3 dt_proj   W_dt @ x_dbl[:, :R]^T -> (D, M)": lambda: Wdt @ x_dbl[:, :R].t(),
 "F4 out_proj  linear(out_z^T, W_out) -> (M, 512)": lambda: torch.nn.functional.linear(out_z.t(), Wout),
 "B1 dout_y    W_out^T @ dout^T -> (D, M)": lambda: Wout.t() @ dout.t(),
 "B2 dW_out    mm_nn_rows(out_z, dout) -> (D, 512)": lambda: gemm.mm_nn_rows(out_z, dout),
 "B4 ddt_w     mm_nn_rows(ddelta, x_dbl[:, :R]) -> (D, 32)": lambda: gemm.mm_nn_rows(ddelta, x_dbl[:, :R]),
 "B5 dx_dbl_r  ddelta^T @ W_dt -> (M, 32)": lambda: ddelta.t() @ Wdt,
 "B6 dW_x      mm_nn_rows(conv_out, dx_dbl) -> (D, 64)": lambda: gemm.mm_nn_rows(conv_out, dx_dbl),
 "B7 dconv     addmm(dconv, W_x^T, dx_dbl^T) -> (D, M)": lambda: torch.addmm(dconv, Wx.t(), dx_dbl.t()),
 "B9 dW_in     mm_nn_rows(dxz, x) -> (2D, 512)": lambda: gemm.mm_nn_rows(dxz, x),
 "B10 dx_in    mm(dxz^T, W_in) -> (M, 512)": lambda: torch.mm(dxz.t(), Win),
}
tot = 0
for name, f in ops.items():
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10; tot += ms
    print(f"{name:60s} {ms * 1e3:8.1f} us")
print("sum per mixer", tot, "ms")
