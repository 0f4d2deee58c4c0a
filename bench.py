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


def scan_bwd_bytes(B, D, L, N, G=1, s=4):
    """SURVEY.md 8(d), with the out_z recompute MambaInnerFn always requests: reads u, delta, z, out, dout, B, C, x;
    writes du, ddelta, dz, out_z, dB, dC (fp32): 9 B D L s + 2 B G N L (s + 4) + x"""
    return 9 * B * D * L * s + 2 * B * G * N * L * (s + 4) + B * D * ((L + 2047) // 2048) * 2 * N * 4 + (D * N + 2 * D) * 4


class ScanTimer:
    """HIP-event pairs around every selective-scan launch (recorded on torch's current stream = the launch stream).
    The events bracket exactly one kernel launch: native.selective_scan_fwd/_bwd allocate (caching allocator, no
    device work) and launch; the bwd's accumulator memsets are issued before the first event."""

    def __init__(self):
        self.events, self.bytes, self.enabled = {"fwd": [], "bwd": []}, {"fwd": [], "bwd": []}, False

    def install(self):
        from dimsum_amd import _lib
        lib = _lib.load()
        timer = self

        class Timed:
            """proxy of the ctypes library: times the two scan entry points, forwards everything else"""

            def __getattr__(self, name):
                return getattr(lib, name)

            def _timed(self, which, fn, P, stream):
                if not timer.enabled:
                    return fn(P, stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(P, stream)
                e1.record()
                p = P.fwd if which == "bwd" else P
                f = scan_bwd_bytes if which == "bwd" else scan_bytes
                s = {_lib.F32: 4}.get(p.dtype, 2)
                timer.events[which].append((e0, e1))
                timer.bytes[which].append(f(p.batch, p.dim, p.seqlen, p.dstate, p.n_groups, s))
                return rc

            def dimsum_ssm_scan_fwd(self, P, stream):
                return self._timed("fwd", lib.dimsum_ssm_scan_fwd, P, stream)

            def dimsum_ssm_scan_bwd(self, P, stream):
                return self._timed("bwd", lib.dimsum_ssm_scan_bwd, P, stream)

        proxy = Timed()
        _lib.load = lambda: proxy

    def summary(self, which="fwd"):
        if not self.events[which]:
            return None
        ms = [a.elapsed_time(b) for a, b in self.events[which]]
        avg_ms = sum(ms) / len(ms)
        avg_bytes = sum(self.bytes[which]) / len(self.bytes[which])
        return avg_ms, avg_bytes, len(ms)


def build_model(name, device, image_size=256, scan_type="none"):
    from dimsum_amd.create_model import create_model, published_config
    from dimsum_amd.utils import rerandomize_zeros
    torch.manual_seed(0)
    model = create_model(published_config(model=name, image_size=image_size, bimamba_type=scan_type))
    rerandomize_zeros(model, std=0.02, seed=0)       # reference init is adaLN-zero (SURVEY finding 5)
    return model.to(device).eval()


def build_block(name, device):
    """BASELINE configs[2]: one DiMBlockCombined of `name`'s width with the published flags (layer 1: reverse sweep,
    transposed Haar window scan), weights as in build_model."""
    from dimsum_amd.models_dim import create_block
    from dimsum_amd.utils import rerandomize_zeros
    hidden = {"DiM-S/2": 384, "DiM-B/2": 768, "DiM-L/2": 1024, "DiM-L/4": 1024, "DiM-XL/2": 1152}[name]
    torch.manual_seed(0)
    blk = create_block(hidden, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=True, transpose=False, cond_mamba=True,
                       scanning_continuity=False, use_gated_mlp=True)
    rerandomize_zeros(blk, std=0.02, seed=0)
    return blk.to(device).train(), hidden


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
    ap.add_argument("--scan-type", default="none", help="none (published configs) | zigma_8 | sweep_8 | jpeg_8: zigzag token "
                                                         "orders inside the mixers (BASELINE configs[4])")
    ap.add_argument("--mode", choices=["fwd", "sample", "block", "train"], default="fwd",
                    help="fwd: denoiser forward (headline, BASELINE configs[1]); sample: --nfe Euler steps + all-gather "
                         "(configs[3]); block: ONE DiMBlockCombined forward+backward (configs[2]); train: one whole "
                         "flow-matching training step (loss, backward, DDP all-reduce over RCCL, clip, AdamW, EMA)")
    ap.add_argument("--nfe", type=int, default=250)
    ap.add_argument("--hip-graph", action="store_true", help="fwd / sample: replay the denoiser forward from a captured hipGraph "
                                                             "(dimsum_amd/hip_graph.py): for per-GPU batches below ~32, where eager is launch-bound")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the extra exact-fp32 timing (for profiling runs)")
    ap.add_argument("--matmul", choices=["tf32", "fp32", "fp16"], default="tf32",
                    help="library-GEMM policy. tf32 = the reference's own setting (torch.backends.cuda.matmul.allow_tf32 = "
                         "True, dimsum/train.py:20-21, sample_ddp.py:56); on gfx950 hipBLASLt serves it with a split-bf16 "
                         "MFMA path measured at 4e-6 rms relative error (real TF32: ~5e-4). fp32 = exact fp32 MFMA. "
                         "fp16 = opt-in, inference only: fp16 operands (TF32's 10 mantissa bits, NOT its exponent range) with "
                         "fp32 accumulation for the large Linears (dimsum_amd/gemm.py); never the headline.")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)            # before the process group: every rank binds its own GPU
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)          # RCCL on ROCm
    torch.set_num_threads(max(1, usable_cores() // max(1, world)))

    def set_matmul(policy):
        from dimsum_amd import gemm
        torch.backends.cuda.matmul.allow_tf32 = policy in ("tf32", "fp16")
        torch.backends.cudnn.allow_tf32 = policy in ("tf32", "fp16")
        gemm.set_policy("fp16" if policy == "fp16" else "default")
    set_matmul(args.matmul)

    from dimsum_amd import _lib
    _lib.load()                                   # fail loudly if the HIP library is missing
    r = args.image_size // 8
    if args.mode == "block":
        model, hidden = build_block(args.model, dev)
    else:
        model = build_model(args.model, dev, args.image_size, args.scan_type)
    gen = torch.Generator(device=dev).manual_seed(0 * world + rank)      # sample_ddp.py:64 seeding rule
    x = torch.randn(args.batch, 4, r, r, device=dev, generator=gen)
    t = torch.rand(args.batch, device=dev, generator=gen)
    y = torch.randint(0, 1000, (args.batch,), device=dev, generator=gen)

    timer = ScanTimer()
    timer.install()

    if args.mode == "train":
        from dimsum_amd.train import build_training, train_step
        from dimsum_amd.transport import create_transport
        ddp, ema, opt = build_training(model.train(), dev, 1e-4, world, [local_rank])
        transport = create_transport("GVP", "velocity")

        def step():
            return train_step(ddp, ema, opt, transport, x, y)
        units_per_step = args.batch
    elif args.mode == "block":
        ntok = (r // 2) ** 2
        hs = torch.randn(args.batch, ntok, hidden, device=dev, generator=gen).requires_grad_()
        res = torch.randn(args.batch, ntok, hidden, device=dev, generator=gen).requires_grad_()
        cond = torch.randn(args.batch, hidden, device=dev, generator=gen).requires_grad_()
        dy = torch.randn(args.batch, ntok, hidden, device=dev, generator=gen)

        def step():
            for p_ in model.parameters():
                p_.grad = None
            hs.grad = res.grad = cond.grad = None
            out, res_out = model(hs, res, cond)
            torch.autograd.backward((out, res_out), (dy, dy))
        units_per_step = args.batch
    elif args.mode == "fwd":
        if args.hip_graph:
            from dimsum_amd.hip_graph import GraphedForward
            graphed = GraphedForward(model)

            def step():
                return graphed(x, t, y)
        else:
            def step():
                with torch.no_grad():
                    return model(x, t, y)
        units_per_step = args.batch
    else:
        from dimsum_amd.sample_ddp import sample_batch
        graphs = {} if args.hip_graph else None

        def step():
            return sample_batch(model, x, y, num_steps=args.nfe, world_size=world, hip_graph=graphs)
        units_per_step = args.batch

    def fence():
        if world > 1:
            dist.barrier(device_ids=[local_rank])
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
        what = {"fwd": "denoiser forward", "sample": f"denoiser {args.nfe}-NFE Euler sampling",
                "block": "ONE DiMBlockCombined (scan + Haar + attention fusion + gated MLP) forward+backward",
                "train": "flow-matching training step (GVP velocity loss, backward, grad all-reduce, clip, AdamW, EMA)"}[args.mode]
        line = {
            "metric": {"fwd": "denoiser-fwd latents/sec", "sample": f"{args.nfe}-NFE samples/sec",
                       "block": "block fwd+bwd latents/sec", "train": "training latents/sec"}[args.mode],
            "value": value, "unit": "samples/s" if args.mode == "sample" else "latents/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} {what}, "
                                   f"{args.image_size}px (4x{r}x{r} latents, {(r // 2) ** 2} tokens), {args.batch} latents per GPU, "
                                   "random-init weights (reference init, zero tensors re-drawn N(0, 0.02^2))"
                                   + (f", scan_type={args.scan_type}" if args.scan_type != "none" else ""),
                       "global_batch": args.batch * world, "parallelism": f"dp{world} (replicas, independent latents)",
                       "launch": "hipGraph replay" if args.hip_graph else "eager",
                       "matmul_policy": {"tf32": "allow_tf32=True like the reference (train.py:20-21); gfx950 split-bf16 path, 4e-6 rms rel err",
                                         "fp32": "exact fp32",
                                         "fp16": "OPT-IN: fp16 operands (TF32 mantissa, fp16 exponent range) + fp32 accumulation for the large "
                                                 "Linears, split-bf16 elsewhere"}[args.matmul]},
        }
        if fwd_mode:
            line["samples_per_sec_at_250_nfe"] = value / 250.0

        def roof(which, kernel, pmc):
            s = timer.summary(which)
            if s is None:
                return None
            avg_ms, avg_bytes, n = s
            achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
            traffic = None
            prof = os.path.join(ROOT, "profiles", pmc)    # HBM bytes per launch from rocprofv3 --pmc
            if os.path.exists(prof):
                traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
            return {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes_per_launch": avg_bytes,
                    "avg_launch_ms": avg_ms, "launches_timed": n}

        rf = roof("fwd", "ssm_scan_fwd_kernel<float,16>", "scan_fwd_pmc.json")
        if rf is not None:
            line["roofline"] = rf
        rb = roof("bwd", "ssm_scan_bwd_kernel<float,16>", "scan_bwd_pmc.json")
        if rb is not None:
            line["roofline_bwd"] = rb
        if world == 1 and args.matmul == "tf32" and args.mode in ("fwd", "block") and not args.no_fp32_leg:
            # the same step with exact-fp32 library GEMMs, for reference (2 untimed + 2 timed steps)
            set_matmul("fp32")
            timer.enabled = False
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 2
            line["fp32_exact_matmul"] = {"value_per_gpu": units_per_step / dt, "ms_per_step": 1e3 * dt}
            if args.mode == "fwd":
                # opt-in fp16-operand policy of dimsum_amd/gemm.py, for reference only (never the headline)
                ref = step()
                set_matmul("fp16")
                for _ in range(2):
                    got = step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 3
                line["fp16_operand_matmul_optin"] = {"value_per_gpu": units_per_step / dt, "ms_per_step": 1e3 * dt,
                                                     "max_abs_dev_over_max_abs_vs_exact_fp32": ((got - ref).abs().max() / ref.abs().max()).item()}
            set_matmul("tf32")
        if world == 1 and not args.no_cpu_baseline and args.mode in ("fwd", "sample"):
            line["cpu_baseline"] = cpu_baseline(args.model, 8, args.image_size)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier(device_ids=[local_rank])     # nobody tears the communicator down while rank 0 is still reporting
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
