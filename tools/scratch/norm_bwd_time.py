"""norm backward (RMS, with dresidual) at the training legs' shapes: time per launch and bytes / time"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dimsum_amd import native
for rows, N in ((16384, 1024), (65536, 1024), (65536, 1152)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x, dy, dres = (torch.randn(rows, N, device="cuda", generator=g) for _ in range(3))
    w = torch.randn(N, device="cuda", generator=g)
    y, mean, rstd, stream = native.layer_norm_fwd(x, w, None, 1e-5, None, is_rms_norm=True)
    f = lambda: native.layer_norm_bwd(dy, x, w, None, 1e-5, mean, rstd, dres, False, True)
    for _ in range(3): f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for i in range(20):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
    print(rows, N, "median %.1f us  min %.1f us  %.2f TB/s" % (ms[10] * 1e3, ms[0] * 1e3, 4 * rows * N * 4 / ms[10] / 1e9))
