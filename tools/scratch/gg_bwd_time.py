"""time of the gated-GeLU adjoint (scaled-fp16 image) at the training shapes; A / B over DIMSUM_GG_RPW_DIV (rows per workgroup = rows / div)"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dimsum_amd import native
for rows in (16384, 65536):
    x12, b, dh = torch.randn(rows, 8192, device="cuda"), torch.randn(8192, device="cuda"), torch.randn(rows, 4096, device="cuda")
    for _ in range(3): native.gated_gelu_bwd(x12, b, dh, split3="f16s")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): native.gated_gelu_bwd(x12, b, dh, split3="f16s")
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("DIMSUM_GG_RPW_DIV", "default"), rows, round(e0.elapsed_time(e1) / 10 * 1e3, 1), "us")
