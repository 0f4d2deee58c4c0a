"""tools/bench_token.py (GPU box): the token passes around the mixers (csrc/token_transform.hip) at the DiM-L/2 launch shapes:
microseconds and GB/s of algorithmic traffic per variant (fp32 / split-bf16 image / scaled-fp16 image output)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dimsum_amd import native  # noqa: E402
from dimsum_amd.ops import token_ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B, L, C = 256, 256, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    hs = torch.randn(B, L, 2 * C, device="cuda", generator=g)
    x = hs[:, :, :C]
    m = torch.randn(B, L, C, device="cuda", generator=g)
    shift, scale, gate = (0.1 * torch.randn(B, C, device="cuda", generator=g) for _ in range(3))
    perm = torch.randperm(L, device="cuda", generator=g).to(torch.int32)
    table = {"inv32": perm}
    n = B * L * C
    for kind in ("none", "haar", "dct"):
        for mode, out_b in ((False, 4), (True, 6), ("f16s", 2)):
            t = timed(lambda: token_ops.pre_mixer(x, kind, table, shift, scale, split3=mode))
            print(f"pre_mixer  {kind:5s} out={str(mode):5s} {t:7.1f} us  {n * (4 + out_b) / t / 1e3:7.1f} GB/s", flush=True)
            t = timed(lambda: token_ops.post_mixer(x, m, gate, kind, table, split3=mode))
            print(f"post_mixer {kind:5s} out={str(mode):5s} {t:7.1f} us  {n * (8 + out_b) / t / 1e3:7.1f} GB/s", flush=True)
    full = torch.randn(B, L, 2 * C, device="cuda", generator=g)
    y = torch.randn(B, L, 2 * C, device="cuda", generator=g)
    g2 = 0.1 * torch.randn(B, 2 * C, device="cuda", generator=g)
    bias = torch.randn(2 * C, device="cuda", generator=g)
    t = timed(lambda: token_ops.gate_residual(full, y, g2, bias))
    print(f"gate_residual (B, L, 1024)       {t:7.1f} us  {B * L * 2 * C * 12 / t / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
