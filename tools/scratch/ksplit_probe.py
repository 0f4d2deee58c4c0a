#!/usr/bin/env python3
"""Is ONE plain bf16 library GEMM over K-stacked hi / lo operands ([Xhi Xhi Xlo] . [Whi; Wlo; Whi], fp32 accumulate and output)
faster than hipBLASLt's own split-bf16 path (fp32 operands under allow_tf32)?  GPU box."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = True
dev = "cuda"

def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

import sys
SHAPES = [(65536, 1024, 8192), (65536, 4096, 1024), (65536, 512, 2048), (65536, 1024, 512)] if len(sys.argv) < 2 else [(65536, 512, 1536), (65536, 1024, 1024), (65536, 1024, 3072), (65536, 576, 1728), (65536, 1152, 1152)]
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    fl = 2 * M * K * N
    t0 = timeit(lambda: torch.nn.functional.linear(x, w))
    xh = x.bfloat16(); xl = (x - xh.float()).bfloat16(); wh = w.bfloat16(); wl = (w - wh.float()).bfloat16()
    xs = torch.cat([xh, xh, xl], 1).contiguous(); ws = torch.cat([wh, wl, wh], 1).contiguous()
    ref = torch.nn.functional.linear(x.double(), w.double())
    y0 = torch.nn.functional.linear(x, w)
    res = {"split_lib_ms": t0, "TFeq": fl / t0 / 1e9}
    try:
        f1 = lambda: torch.mm(xs, ws.t(), out_dtype=torch.float32)
        t1 = timeit(f1); y1 = f1()
        res.update(kstack_f32out_ms=t1, kstack_TFeq=fl / t1 / 1e9, err_kstack=((y1 - ref).abs().max() / ref.abs().max()).item())
    except Exception as e:
        res["kstack_f32out"] = repr(e)[:120]
    t2 = timeit(lambda: torch.mm(xs, ws.t()))
    res.update(kstack_bf16out_ms=t2, kstack_bf16out_TF=3 * fl / t2 / 1e9, err_lib=((y0 - ref).abs().max() / ref.abs().max()).item())
    t3 = timeit(lambda: torch.cat([xh, xh, xl], 1))
    res["build_xs_ms"] = t3
    print((M, K, N), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}, flush=True)
