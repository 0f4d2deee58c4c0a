#!/usr/bin/env python3
"""weight gradients with a small output and a 65536-long reduction: one GEMM (16 workgroups on 256 CUs) vs a batched GEMM over S
slices of the reduction + a sum (split-K by hand), fp32 operands under allow_tf32"""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = True
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
M, D, dm = 65536, 1024, 512
dev = "cuda"
dout = torch.randn(M, dm, device=dev); out_z = torch.randn(D, M, device=dev)
dxz = torch.randn(2 * D, M, device=dev); x = torch.randn(M, dm, device=dev)
ref_in = dxz.double() @ x.double(); ref_out = dout.t().double() @ out_z.t().double()
print("dW_in  one GEMM %.3f ms" % timeit(lambda: dxz @ x))
print("dW_out one GEMM %.3f ms" % timeit(lambda: dout.t() @ out_z.t()))
for S in (4, 8, 16, 32):
    f_in = lambda: torch.bmm(dxz.view(2 * D, S, M // S).permute(1, 0, 2), x.view(S, M // S, dm)).sum(0)
    f_out = lambda: torch.bmm(dout.view(S, M // S, dm).transpose(1, 2), out_z.view(D, S, M // S).permute(1, 2, 0)).sum(0)
    e_in = ((f_in() - ref_in).abs().max() / ref_in.abs().max()).item(); e_out = ((f_out() - ref_out).abs().max() / ref_out.abs().max()).item()
    print("S=%2d  dW_in %.3f ms (err %.1e)   dW_out %.3f ms (err %.1e)" % (S, timeit(f_in), e_in, timeit(f_out), e_out))
