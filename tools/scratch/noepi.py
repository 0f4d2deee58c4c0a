"""tools/scratch/noepi.py (GPU box, DIMSUM_HIP_LIB = a -DDIMSUM_GEMM_TUNE build): the K loop alone (tune 2: accumulators discarded) against the
kernel with its fp32 stores (tune 12) on the forward's launch shapes, bf16 operands (the same MFMA rate as fp16): what a tile's epilogue costs"""
import json, sys, torch
sys.path.insert(0, ".")
from dimsum_amd import native
def t(f, n=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for name, M, N, K in [("in_proj", 2048, 65536, 512), ("qkv", 65536, 1536, 512), ("proj", 65536, 1024, 1024), ("w12 (fp32 out)", 65536, 8192, 1024), ("w3", 65536, 1024, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda")
    r = {}
    for tag, tv in (("k_loop_only", 2), ("with_stores", 12)):
        ms = sorted(t(lambda: native.gemm_nt(a, b, out=out, tune=(tv, 0, 0))) for _ in range(5))[2]
        r[tag] = round(ms, 4)
    r["TF_k_loop"] = round(2.0 * M * N * K / r["k_loop_only"] / 1e9, 1)
    r["out_GB"] = round(M * N * 4 / 1e9, 3)
    r["store_GBps_if_serial"] = round(M * N * 4 / ((r["with_stores"] - r["k_loop_only"]) * 1e-3) / 1e9, 0)
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, **r}), flush=True)
