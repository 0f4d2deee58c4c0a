import sys, os, torch
sys.path.insert(0, "/root/repo")
from dimsum_amd import native
def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)
def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, M, N, K, dt in (("in_proj f16 (d-major out)", 2048, 65536, 512, torch.float16), ("in_proj split3 (d-major)", 2048, 65536, 1536, torch.bfloat16),
                          ("qkv f16", 65536, 1536, 512, torch.float16), ("qkv split3", 65536, 1536, 1536, torch.bfloat16),
                          ("proj f16", 65536, 1024, 1024, torch.float16), ("proj split3", 65536, 1024, 3072, torch.bfloat16),
                          ("w3 f16", 65536, 1024, 4096, torch.float16)):
    a, b = rnd((M, K), dt, 1), rnd((N, K), dt, 2, K ** -0.5)
    c = torch.empty((M, N), device="cuda")
    res = {}
    for gm in (1, 2, 4, 8, 16):
        f = lambda gm=gm: native.gemm_nt(a, b, out=c, tune=(0, gm, 0))
        f(); torch.cuda.synchronize()
        res[gm] = min(timed(f) for _ in range(3))
    lib = lambda: torch.mm(a, b.t(), out_dtype=torch.float32)
    lib(); tl = min(timed(lib) for _ in range(3))
    print(f"{name:28s} M={M} N={N} K={K}: " + "  ".join(f"gm{g} {t*1e3:6.1f}us" for g, t in res.items()) + f"  library {tl*1e3:6.1f}us", flush=True)
