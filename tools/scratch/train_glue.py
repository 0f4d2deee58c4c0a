#!/usr/bin/env python3
"""Where the non-dimsum kernels of a training step come from (GPU box): one DiM-L/2 step at batch 64 under torch.profiler with Python stacks;
device time of every torch op grouped by (op, innermost dimsum_amd frame). -> gpurun_out/train_glue.txt"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from dimsum_amd.train import build_training, train_step  # noqa: E402
from dimsum_amd.transport import create_transport  # noqa: E402


def sites(step):
    """call sites of the glue ops: a TorchDispatchMode (propagated to the autograd threads) that records op, innermost dimsum_amd frame, bytes touched"""
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    watch = ("sum", "copy_", "add", "add_", "fill_", "mul", "cat", "neg", "sub", "zero_", "clone", "mm", "addmm", "bmm", "silu", "silu_backward", "exp",
             "zeros", "_to_copy", "zeros_like", "new_zeros", "full", "ones_like", "ones", "div", "abs", "max", "amax", "expand_copy", "index_select", "gather")
    if os.environ.get("GLUE_ALL"):
        watch = None
    names = collections.Counter()
    agg = collections.defaultdict(lambda: [0, 0])

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = func.__name__.split(".")[0]
            names[name] += 1
            if watch is None or name in watch:
                site = "<engine>"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "dimsum_amd/" in fr.filename and "train.py" not in fr.filename:
                        site = f"{fr.filename.split('dimsum_amd/')[-1]}:{fr.lineno} {fr.line[:70]}"
                        break
                nb = 0
                for t in list(args) + [out]:
                    for u in (t if isinstance(t, (list, tuple)) else [t]):
                        if torch.is_tensor(u):
                            nb += u.numel() * u.element_size()
                if site == "<engine>" and name in ("add", "cat", "copy_", "mul", "sum", "mm"):
                    shp = [tuple(u.shape) for t in args for u in (t if isinstance(t, (list, tuple)) else [t]) if torch.is_tensor(u)]
                    site = "<engine> " + str(shp)[:120]
                k = (name, site)
                agg[k][0] += 1
                agg[k][1] += nb
            return out

    with Mode():
        step()
    torch.cuda.synchronize()
    lines = ["ops by count: " + ", ".join(f"{k} {v}" for k, v in names.most_common(60)), "glue op call sites (count, MB touched):"]
    for (name, site), (n, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:120]:
        lines.append(f"{n:5d} {nb / 1e6:10.1f} MB  {name:14s} {site}")
    open("gpurun_out/train_glue_sites.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:100]))


def main():
    os.environ.setdefault("DIMSUM_BRANCH_STREAMS", "0")
    dev = torch.device("cuda:0")
    os.makedirs("gpurun_out", exist_ok=True)
    torch.backends.cuda.matmul.allow_tf32 = True
    torch.backends.cudnn.allow_tf32 = True
    from dimsum_amd import gemm
    gemm.set_policy("f16s")
    batch = int(os.environ.get("GLUE_BATCH", "64"))
    model = bench.build_model("DiM-L/2", dev, 256)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(batch, 4, 32, 32, device=dev, generator=g)
    y = torch.randint(0, 1000, (batch,), device=dev, generator=g)
    ddp, ema, opt = build_training(model.train(), dev, 1e-4, 1, [0])
    tr = create_transport("GVP", "velocity")
    for _ in range(3):
        train_step(ddp, ema, opt, tr, x, y)
    torch.cuda.synchronize()
    sites(lambda: train_step(ddp, ema, opt, tr, x, y))
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        train_step(ddp, ema, opt, tr, x, y)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    total = 0.0
    for ev in prof.events():
        t = getattr(ev, "self_device_time_total", 0) or 0
        if t <= 0 or ev.device_type != torch.autograd.DeviceType.CPU:
            continue
        kern = ",".join(sorted({k.name[:60] for k in ev.kernels})) if ev.kernels else ""
        if "dimsum::" in kern and "at::" not in kern:
            continue
        site = "<engine>"
        for fr in ev.stack or []:
            if "dimsum_amd" in fr or "bench.py" in fr:
                site = fr.split("dimsum_amd/")[-1][:110]
                break
        key = (ev.name, site)
        agg[key][0] += t
        agg[key][1] += 1
        total += t
    out = [f"torch-op device time in one step: {total / 1e3:.2f} ms (batch {batch})"]
    for (name, site), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:90]:
        out.append(f"{t / 1e3:8.3f} ms {n:5d}  {name:34s} {site}")
    open("gpurun_out/train_glue.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:70]))


if __name__ == "__main__":
    main()
