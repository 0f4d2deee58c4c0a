#!/usr/bin/env python3
"""bench.py -- DiM-L/2 (256 px: 4x32x32 latents, L = 256 tokens) denoiser-forward throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one forward of the denoiser hot path (every Mamba / frequency / fusion op through libdimsum_hip.so, GEMMs in
hipBLASLt fp32) over one batch of 256 synthetic latents per GPU -- BASELINE.json configs[1]. The batch is resident in
HBM before the timed region. Weak scaling: every rank owns a replica and its own batch, no data-path collective
(the path shards by independent latents); timing is barrier + synchronize on both sides, max over ranks.

Extra objects on the JSON line:
  roofline      selective-scan forward kernel (the path's dominant hand-written kernel): algorithmic bytes per launch
                (SURVEY.md 8d formula) / its average launch duration measured live with HIP events on the launch
                stream during the timed steps; peak = 8 TB/s HBM3E (MI355X_MICROARCH.md)
  cpu_baseline  the same denoiser forward on the host cores through the CPU oracle ("port"), bounded sample
`--mode sample` times 250-NFE fixed-step Euler flow-matching sampling instead (samples/s; one RCCL all-gather of the
final latents per batch).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0


def usable_cores():
    """host cores this process may really use: min(os.cpu_count, affinity mask, cgroup-v2 cpu.max quota)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def scan_bytes(B, D, L, N, G=1, s=4):
    """5 B D L s + 2 B G N L s + B D ceil(L/2048) 2N 4 + (D N + 2 D) 4"""
    return 5 * B * D * L * s + 2 * B * G * N * L * s + B * D * ((L + 2047) // 2048) * 2 * N * 4 + (D * N + 2 * D) * 4


class ScanTimer:
    """HIP-event pairs around every selective-scan launch (recorded on torch's current stream = the launch stream)."""

    def __init__(self):
        self.events, self.bytes, self.enabled = [], [], False

    def install(self):
        from dimsum_amd import native
        inner = native.selective_scan_fwd
        timer = self

        def timed(u, delta, A, B, C, D, z, delta_bias, delta_softplus, **kw):
            if not timer.enabled:
                return inner(u, delta, A, B, C, D, z, delta_bias, delta_softplus, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = inner(u, delta, A, B, C, D, z, delta_bias, delta_softplus, **kw)
            e1.record()
            timer.events.append((e0, e1))
            timer.bytes.append(scan_bytes(u.shape[0], u.shape[1], u.shape[2], A.shape[1], B.shape[1], u.element_size()))
            return r

        native.selective_scan_fwd = timed

    def summary(self):
        if not self.events:
            return None
        ms = [a.elapsed_time(b) for a, b in self.events]
        avg_ms = sum(ms) / len(ms)
        avg_bytes = sum(self.bytes) / len(self.bytes)
        return avg_ms, avg_bytes, len(ms)


def build_model(name, device, image_size=256):
    from dimsum_amd.create_model import create_model, published_config
    from dimsum_amd.utils import rerandomize_zeros
    torch.manual_seed(0)
    model = create_model(published_config(model=name, image_size=image_size))
    rerandomize_zeros(model, std=0.02, seed=0)       # reference init is adaLN-zero (SURVEY finding 5)
    return model.to(device).eval()


def cpu_baseline(name, latents=2, image_size=256):
    """Bounded CPU sample of the same workload: one DiM forward of `latents` latents through the CPU oracle ("port" of
    the reference's pure-PyTorch path: GEMMs in torch-CPU, scan/conv/norm in oracle/ssm_oracle.c with OpenMP)."""
    from oracle import c_ops
    from oracle.torch_backend import cpu_oracle_backend
    cores = usable_cores()
    torch.set_num_threads(cores)
    c_ops.set_num_threads(cores)
    model = build_model(name, "cpu", image_size)
    r = image_size // 8
    x, t, y = torch.randn(latents, 4, r, r), torch.rand(latents), torch.randint(0, 1000, (latents,))
    with torch.no_grad(), cpu_oracle_backend():
        model(x[:1], t[:1], y[:1])                   # warm-up (allocator, oracle build/load)
        t0 = time.perf_counter()
        model(x, t, y)
        dt = time.perf_counter() - t0
    return {"value": latents / dt, "unit": "latents/s", "cores": cores, "kind": "port",
            "sample": f"1 forward of {name} on {latents} latents (fp32, torch-CPU GEMMs + OpenMP C oracle), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="DiM-L/2")
    ap.add_argument("--batch", type=int, default=256, help="latents per GPU per step")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--mode", choices=["fwd", "sample"], default="fwd")
    ap.add_argument("--nfe", type=int, default=250)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl")          # RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.set_num_threads(max(1, usable_cores() // max(1, world)))

    from dimsum_amd import _lib
    _lib.load()                                   # fail loudly if the HIP library is missing
    model = build_model(args.model, dev, args.image_size)
    r = args.image_size // 8
    gen = torch.Generator(device=dev).manual_seed(0 * world + rank)      # sample_ddp.py:64 seeding rule
    x = torch.randn(args.batch, 4, r, r, device=dev, generator=gen)
    t = torch.rand(args.batch, device=dev, generator=gen)
    y = torch.randint(0, 1000, (args.batch,), device=dev, generator=gen)

    timer = ScanTimer()
    timer.install()

    if args.mode == "fwd":
        def step():
            with torch.no_grad():
                return model(x, t, y)
        units_per_step = args.batch
    else:
        from dimsum_amd.sample_ddp import sample_batch
        def step():
            return sample_batch(model, x, y, num_steps=args.nfe, world_size=world)
        units_per_step = args.batch

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    if rank == 0:
        value = units_per_step * world * args.steps / elapsed
        fwd_mode = args.mode == "fwd"
        line = {
            "metric": "denoiser-fwd latents/sec" if fwd_mode else f"{args.nfe}-NFE samples/sec",
            "value": value, "unit": "latents/s" if fwd_mode else "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} denoiser {'forward' if fwd_mode else f'{args.nfe}-NFE Euler sampling'}, "
                                   f"{args.image_size}px (4x{r}x{r} latents, {(r // 2) ** 2} tokens), {args.batch} latents per GPU, "
                                   "random-init weights (reference init, zero tensors re-drawn N(0, 0.02^2))",
                       "global_batch": args.batch * world, "parallelism": f"dp{world} (replicas, independent latents)"},
        }
        if fwd_mode:
            line["samples_per_sec_at_250_nfe"] = value / 250.0
        s = timer.summary()
        if s is not None:
            avg_ms, avg_bytes, n = s
            achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
            traffic = None
            prof = os.path.join(ROOT, "profiles", "scan_fwd_pmc.json")    # HBM bytes per launch from rocprofv3 --pmc
            if os.path.exists(prof):
                traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
            line["roofline"] = {"kernel": "ssm_scan_fwd_kernel<float,16>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                                "algorithmic_bytes_per_launch": avg_bytes, "avg_launch_ms": avg_ms, "launches_timed": n}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.model, 8, args.image_size)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
