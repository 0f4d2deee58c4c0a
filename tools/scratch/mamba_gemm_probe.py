#!/usr/bin/env python3
"""the GEMMs of one Mamba mixer's backward at the DiM-L/2 block shape, exactly as MambaInnerFn / in_proj issue them (fp32 operands
under allow_tf32), and image alternatives for the two weight gradients whose output is small and whose reduction runs over the 65536 rows"""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = True
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def planes(x, left):            # (R, C) fp32 -> (3R, C) bf16, plane-major
    hi = x.bfloat16(); lo = (x - hi.float()).bfloat16()
    return torch.cat([hi, hi, lo] if left else [hi, lo, hi], 0).contiguous()
def cols(x, left):              # (R, C) -> (R, 3C)
    hi = x.bfloat16(); lo = (x - hi.float()).bfloat16()
    return torch.cat([hi, hi, lo] if left else [hi, lo, hi], 1).contiguous()
M, D, dm, R, N = 65536, 1024, 512, 32, 16
dev = "cuda"
dout = torch.randn(M, dm, device=dev); out_z = torch.randn(D, M, device=dev); Wout = torch.randn(dm, D, device=dev)
dxz = torch.randn(2 * D, M, device=dev); x = torch.randn(M, dm, device=dev); Win = torch.randn(2 * D, dm, device=dev)
ddelta = torch.randn(D, M, device=dev); x_dbl = torch.randn(M, R + 2 * N, device=dev); Wdt = torch.randn(D, R, device=dev)
dx_dbl = torch.randn(M, R + 2 * N, device=dev); conv = torch.randn(D, M, device=dev); Wx = torch.randn(R + 2 * N, D, device=dev)
dconv = torch.randn(D, M, device=dev)
dout2 = dout.t()
print("a  dW_out  = dout^T @ out_z^T        %.3f ms" % timeit(lambda: dout2 @ out_z.t()))
print("f  dout_y  = Wout^T @ dout^T         %.3f ms" % timeit(lambda: Wout.t() @ dout2))
print("b  dW_in   = dxz @ x                 %.3f ms" % timeit(lambda: dxz @ x))
print("g  dx^T    = Win^T @ dxz             %.3f ms" % timeit(lambda: Win.t() @ dxz))
print("c  dW_dt   = ddelta @ x_dbl[:, :R]   %.3f ms" % timeit(lambda: ddelta @ x_dbl[:, :R]))
print("c2 dx_dbl  = ddelta^T @ Wdt          %.3f ms" % timeit(lambda: ddelta.t() @ Wdt))
print("d  dW_x    = dx_dbl^T @ conv^T       %.3f ms" % timeit(lambda: dx_dbl.t() @ conv.t()))
print("e  dconv  += Wx^T @ dx_dbl^T         %.3f ms" % timeit(lambda: torch.addmm(dconv, Wx.t(), dx_dbl.t())))
print("fw in_proj = Win @ x^T               %.3f ms" % timeit(lambda: Win @ x.t()))
print("fw out_proj= out_z^T @ Wout^T        %.3f ms" % timeit(lambda: torch.nn.functional.linear(out_z.t(), Wout)))
print("fw x_proj  = conv^T @ Wx^T           %.3f ms" % timeit(lambda: torch.nn.functional.linear(conv.t(), Wx)))
print("fw dt_proj = Wdt @ x_dbl[:, :R]^T    %.3f ms" % timeit(lambda: Wdt @ x_dbl[:, :R].t()))
# images
oz3 = cols(out_z, True); do3 = planes(dout, False)
print("a' dW_out^T = img(out_z) @ stack(dout)   %.3f ms (+ conversions %.3f + %.3f)" % (timeit(lambda: torch.mm(oz3, do3, out_dtype=torch.float32)),
      timeit(lambda: cols(out_z, True)), timeit(lambda: planes(dout, False))))
dz3 = cols(dxz, True); x3 = planes(x, False)
print("b' dW_in    = img(dxz) @ stack(x)        %.3f ms" % timeit(lambda: torch.mm(dz3, x3, out_dtype=torch.float32)))
ref = (dxz.double() @ x.double()); got = torch.mm(dz3, x3, out_dtype=torch.float32)
print("   err", ((got - ref).abs().max() / ref.abs().max()).item())
