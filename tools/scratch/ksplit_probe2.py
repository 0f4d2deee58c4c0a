#!/usr/bin/env python3
"""out_proj with a d-major (K-major) left operand: fp32 library split path vs one bf16 GEMM over K-stacked hi/lo images. GPU box."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = True
dev = "cuda"
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for (B, L, K, N) in [(256, 256, 1024, 512), (64, 1024, 1152, 576)]:
    M = B * L
    xt = torch.randn(K, M, device=dev)             # d-major activations: (D, B*L)
    w = torch.randn(N, K, device=dev)
    x_view = xt.view(K, B, L).permute(1, 2, 0)     # (B, L, K) with strides (L, 1, B*L): what out_proj sees
    t0 = timeit(lambda: torch.nn.functional.linear(x_view, w))
    hi = xt.bfloat16(); lo = (xt - hi.float()).bfloat16()
    x3 = torch.cat([hi, hi, lo], 0).contiguous()   # (3K, M)
    wh = w.bfloat16(); wl = (w - wh.float()).bfloat16()
    w3 = torch.cat([wh, wl, wh], 1).contiguous()   # (N, 3K)
    t1 = timeit(lambda: torch.mm(x3.t(), w3.t(), out_dtype=torch.float32))
    t2 = timeit(lambda: torch.mm(w3, x3, out_dtype=torch.float32))   # (N, M): transposed result
    ref = torch.nn.functional.linear(x_view, w).reshape(M, N)
    y1 = torch.mm(x3.t(), w3.t(), out_dtype=torch.float32)
    print((B, L, K, N), "fp32-split %.4f ms | bf16 3K (M,N) %.4f ms | bf16 3K (N,M) %.4f ms | maxdiff %.2e" % (t0, t1, t2, (y1 - ref).abs().max().item() / ref.abs().max().item()), flush=True)
