#!/usr/bin/env python3
"""bench.py -- the DiMSUM denoiser hot path on MI355X: every BASELINE.json config on one JSON line.

  python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU. Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
RANK / WORLD_SIZE in the environment) or, run plainly, this file starts them itself as child processes (launch_ranks).

Headline (`value`, configs[1]): DiM-L/2 (256 px: 4x32x32 latents, 256 tokens) denoiser-forward throughput. A step = one
forward of the denoiser (every Mamba / frequency / fusion op through libdimsum_hip.so, GEMMs in hipBLASLt) over one batch
of 256 synthetic latents per GPU, resident in HBM before the timed region. Weak scaling: every rank owns a replica and
its own batch, no data-path collective (the path shards by independent latents); timing is barrier +
synchronize on both sides, max over ranks.

Further legs on the same line (default `--mode all`; each is also a `--mode` of its own for profiling runs):
  sample_250nfe  configs[3]: REAL 250-NFE fixed-step Euler flow-matching sampling of 128 latents per GPU, one RCCL
                 all-gather of the final latents -- at every N (so the scaling runs exercise the collective)
  block_fwdbwd   configs[2]: one DiMBlockCombined(1024) forward+backward at batch 256, with `roofline_bwd`   (N = 1)
  xl512_zigzag   configs[4]: DiM-XL/2 at 512 px (1024 tokens), batch 64, 8-way zigzag scanning orders       (N = 1)
  train_step     SURVEY 8 f2: one whole flow-matching training step of DiM-L/2 (loss, backward, clip, fused AdamW, EMA), batch 64 (N = 1)
  cpu_baseline   the same forward on the host cores through the CPU oracle ("port"): 1 warm-up + 3 runs, median (N = 1)
Every `roofline*` object: algorithmic bytes per launch (SURVEY.md 8d formula) / average launch duration of the scan
kernel measured live with HIP events on the launch stream -- during the timed steps of the training legs (one stream), and
for the inference legs (whose blocks run their two branches on two HIP streams: overlapped kernels have no duration of
their own) in a short single-stream pass right after the timed region (`roofline.timed_in`); the kernel name comes from
the library's own dispatch (dimsum_ssm_scan_fwd_variant); `traffic` (HBM bytes from rocprofv3 --pmc, profiles/scan_pmc.json) is
attached only when the timed launches have exactly the profiled shape and kernel, otherwise null.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0
WEIGHTS = "random-init weights (reference init, zero tensors re-drawn N(0, 0.02^2))"


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


def scan_bytes(B, D, L, N, G=1, s=4, has_out=True, has_x=True, dt_rank=0, out_z_f16=False):
    """5 B D L s + 2 B G N L s + B D ceil(L/2048) 2N 4 + (D N + 2 D) 4 for the full interface (reads u, delta, z,
    B, C, A, D, delta_bias; writes out, out_z, x). A launch that skips the `out` / `x` stores (inference: out_ptr / x_ptr NULL)
    is priced with what it moves: one B D L s term / the chunk-state term less -- 8(d)'s "inference-only lower bound" 1.082 GB; a launch
    with the fused dt_proj (dt_rank > 0: delta = W_dt x_dbl[:R] formed inside the kernel) reads x_dbl[:R] and W_dt instead of delta; one that
    writes out_z as block-scaled fp16 (out_z_f16) writes 2 bytes per element + one scale per 64 x 32 block."""
    terms = (5 if has_out else 4) - (1 if dt_rank else 0)      # fused dt_proj: `delta` is not read; the launch reads x_dbl[:R] and W_dt instead
    return (terms * B * D * L * s + 2 * B * G * N * L * s + (B * D * ((L + 2047) // 2048) * 2 * N * 4 if has_x else 0)
            + (D * N + 2 * D) * 4 + (B * L * dt_rank + D * dt_rank) * 4 - ((B * D * L * (s - 2) - B * D * L // 2048 * 4) if out_z_f16 else 0))


def scan_bwd_bytes(B, D, L, N, G=1, s=4, recompute_out_z=True):
    """reads u, delta, z, out, dout, B, C, x; writes du, ddelta, dz, dB, dC (fp32) (+ out_z when the caller
    asks for its recompute, like the reference's MambaInnerFn): (8 [+ 1]) B D L s + 2 B G N L (s + 4) + x"""
    return (9 if recompute_out_z else 8) * B * D * L * s + 2 * B * G * N * L * (s + 4) + B * D * ((L + 2047) // 2048) * 2 * N * 4 + (D * N + 2 * D) * 4


class ScanTimer:
    """HIP-event pairs around every selective-scan launch (recorded on torch's current stream = the launch stream).
    The events bracket exactly one C-ABI call: native.selective_scan_fwd/_bwd allocate (caching allocator, no device
    work) and launch; the bwd's accumulator memsets are issued before the first event. Every record keeps the launch
    shape and the kernel the library's dispatch picks for it."""

    def __init__(self):
        self.records, self.enabled = {"fwd": [], "bwd": []}, False

    def reset(self):
        self.records = {"fwd": [], "bwd": []}
        self._free = list(getattr(self, "_all", []))

    def pool_event(self, lib):
        """raw hipEvent_t handles, created once and reused after every reset()"""
        if not getattr(self, "_free", None):
            self._all = getattr(self, "_all", [])
            self._all.append(lib.dimsum_event_create())
            return self._all[-1]
        return self._free.pop()

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
                p = P.fwd if which == "bwd" else P
                x = p.ext.contents if p.ext else _lib.attach_ext(p, _lib.SsmExt)       # dimsum_ssm_ext_t: what the call uses beyond the reference interface
                if which == "fwd":
                    kernel = (_lib.SCAN_FWD_KERNELS[lib.dimsum_ssm_scan_fwd_variant(p)] + (" (+ saved states)" if x.ckpt_ptr else "")
                              + ("" if (p.out_ptr and p.x_ptr) else " (inference: no out / x stores)")
                              + (" (+ fused dt_proj: delta formed in the kernel, not read)" if x.dt_w_ptr else "")
                              + (" (out_z as block-scaled fp16)" if x.out_z_f16 else ""))
                else:
                    kernel = "ssm_scan_bwd_kernel (+ ssm_scan_bwd_reduce_kernel)" + ("" if p.out_z_ptr else ", no out_z recompute")
                # HIP events recorded at the begin of the call's first kernel and the end of its last one (the per-call
                # timing_start_event / timing_stop_event fields of the parameters: hipExtLaunchKernel on the launch stream) --
                # the kernels' own time, like rocprofv3's; under a hipGraph capture (events are not capturable) plain event
                # records around the call
                if torch.cuda.is_current_stream_capturing():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = fn(P, stream)
                    e1.record()
                else:
                    e0, e1 = timer.pool_event(lib), timer.pool_event(lib)
                    x.timing_start_event, x.timing_stop_event = e0, e1
                    rc = fn(P, stream)
                    x.timing_start_event = x.timing_stop_event = None
                s = {_lib.F32: 4}.get(p.dtype, 2)
                shape = (p.batch, p.dim, p.seqlen, p.dstate)
                # algorithmic bytes = SURVEY 8(d)'s formula for the scan this launch performs (full interface, or its inference-only lower bound
                # when the `out` / `x` stores are skipped); `moved`: what the launch itself reads and writes -- less than that when dt_proj runs
                # inside it (delta is formed on the matrix cores instead of being read)
                nbytes = (scan_bwd_bytes(*shape, p.n_groups, s, recompute_out_z=bool(p.out_z_ptr)) if which == "bwd"
                          else scan_bytes(*shape, p.n_groups, s, has_out=bool(p.out_ptr), has_x=bool(p.x_ptr)))
                moved = nbytes if (which == "bwd" or not (x.dt_w_ptr or x.out_z_f16)) else scan_bytes(*shape, p.n_groups, s, has_out=bool(p.out_ptr), has_x=bool(p.x_ptr),
                                                                                                      dt_rank=x.dt_rank if x.dt_w_ptr else 0, out_z_f16=bool(x.out_z_f16))
                timer.records[which].append((e0, e1, nbytes, shape, kernel, moved))
                return rc

            def dimsum_ssm_scan_fwd(self, P, stream):
                return self._timed("fwd", lib.dimsum_ssm_scan_fwd, P, stream)

            def dimsum_ssm_scan_bwd(self, P, stream):
                return self._timed("bwd", lib.dimsum_ssm_scan_bwd, P, stream)

        proxy = Timed()
        _lib.load = lambda: proxy

    def roofline(self, which):
        """roofline object of the dominant (shape, kernel) class among the timed launches of `which`, or None"""
        recs = self.records[which]
        if not recs:
            return None
        classes = {}
        for r in recs:
            classes.setdefault((r[3], r[4]), []).append(r)
        (shape, kernel), rs = max(classes.items(), key=lambda kv: len(kv[1]))
        from dimsum_amd import _lib
        lib = _lib.load()
        ms = lambda a, b: a.elapsed_time(b) if isinstance(a, torch.cuda.Event) else float(lib.dimsum_event_elapsed_ms(a, b))
        avg_ms = sum(ms(r[0], r[1]) for r in rs) / len(rs)
        nbytes = rs[0][2]
        traffic = valu = None
        prof = os.path.join(ROOT, "profiles", "scan_pmc.json")         # HBM bytes per launch + VALU counters from rocprofv3 --pmc
        if os.path.exists(prof):
            for e in json.load(open(prof)).get("entries", []):
                if tuple(e.get("shape_BDLN", ())) == shape and e.get("bench_kernel") == kernel:
                    traffic = e.get("hbm_bytes_per_launch")
                    valu = e.get("valu")
        # `achieved` / `frac` price the bytes THIS launch moves across HBM. For the reference-shaped launches that is SURVEY 8(d)'s formula;
        # a launch with dt_proj fused in (delta formed on the matrix cores, not read) and / or fp16 out_z moves less than the scan it
        # performs would by 8(d): its 8(d) figure is kept under explicitly named keys, never as `frac`.
        moved = rs[0][5]
        achieved = moved / (avg_ms * 1e-3) / 1e9
        rf = {"kernel": kernel, "shape_BDLN": list(shape), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
              "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes_per_launch": moved, "avg_launch_ms": avg_ms,
              "launches_timed": len(rs)}
        if moved != nbytes:
            rf["bytes_8d_of_the_scan_performed"] = nbytes
            rf["frac_if_priced_by_8d"] = nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        if valu:
            # counters of the same kernel at the same shape (profiles/scan_pmc.json <- rocprofv3 --pmc): the share of cycles a SIMD spends issuing
            # VALU work and the VALU instructions per (step, state) pair of a lane, next to the kernel's stated floor. The launch is VALU-bound
            # when that share exceeds the share of the ACHIEVABLE HBM rate (6.3 TB/s, MI355X_MICROARCH.md) its traffic takes.
            rf["valu"] = valu
            if valu.get("busy") is not None and valu["busy"] > achieved / 6300.0:
                rf["bound"] = "valu"
        if _BOX:
            rf["frac_of_box_copy"] = achieved / _BOX["copy_GBps"]
        return rf


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


def cpu_baseline(name, latents=4, image_size=256, runs=3, scan="c", warm=True):
    """Bounded CPU sample of the headline workload: DiM forwards of `latents` latents through the CPU oracle
    ("port" of the reference's pure-PyTorch path: GEMMs in torch-CPU, scan / conv / norm in oracle/ssm_oracle.c with OpenMP),
    1 warm-up + `runs` timed runs, median."""
    from oracle import c_ops
    from oracle.torch_backend import cpu_oracle_backend
    cores = usable_cores()
    torch.set_num_threads(cores)
    c_ops.set_num_threads(cores)
    model = build_model(name, "cpu", image_size)
    r = image_size // 8
    x, t, y = torch.randn(latents, 4, r, r), torch.rand(latents), torch.randint(0, 1000, (latents,))
    times = []
    with torch.no_grad(), cpu_oracle_backend(scan=scan):
        if warm:
            model(x, t, y)                           # warm-up (allocator, oracle build/load, page-in of the weights)
        for _ in range(runs):
            t0 = time.perf_counter()
            model(x, t, y)
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    how = ("torch-CPU GEMMs + OpenMP C oracle" if scan == "c" else
           "torch-CPU GEMMs, the scan as the reference's pure-PyTorch selective_scan_ref restated (two (B, D, L, N) temporaries + a Python loop "
           "over L), conv / norm in the C oracle")
    return {"value": latents / med, "unit": "latents/s", "cores": cores, "kind": "port",
            "sample": f"{name} forward on {latents} latents (fp32, {how}): {1 if warm else 0} warm-up + {runs} runs, "
                      f"median {med:.2f} s (runs: {', '.join(f'{v:.2f}' for v in times)})"}


def rank_summary(elapsed_s, issue_s, steps, extra=None):
    """At N > 1 the line must explain itself (a sub-linear curve could be a straggler, the collective or the host): every rank
    contributes its own timed-region wall time, the time its host needed to ISSUE the steps (before the closing fence: close to the
    wall time = launch-bound, far below = GPU-bound) and its thread count; rank-agnostic (gloo / nccl)."""
    mine = {"rank": dist.get_rank(), "ms_per_step": 1e3 * elapsed_s / steps, "host_issue_fraction": issue_s / max(elapsed_s, 1e-12),
            "torch_threads": torch.get_num_threads()}
    if extra:
        mine.update(extra)
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, mine)
    ms = [e["ms_per_step"] for e in every]
    out = {"per_rank_ms_per_step": {"min": min(ms), "max": max(ms), "rank_of_max": ms.index(max(ms)), "all": [round(v, 3) for v in ms]},
           "host_issue_fraction_max": max(e["host_issue_fraction"] for e in every),
           "torch_threads_per_rank": sorted({e["torch_threads"] for e in every})}
    for k in (extra or {}):
        vals = [e[k] for e in every]
        out[k] = {"min": min(vals), "max": max(vals)} if all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in vals) else all(vals)
    return out


def launch_ranks(gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: this process is only the launcher. It starts N FRESH
    rank processes (one per GPU) as children through torch.distributed.run -- the same command line the driver would use,
    like the reference's `torchrun --nnodes=1 --nproc_per_node=N sample_ddp.py` (scripts/eval.sh:73) -- lets them inherit
    stdout (rank 0 prints the JSON line) and returns their exit code. It never touches the GPU itself: no HIP call has
    happened in this process (import torch does not initialise the device), and nothing is exec'ed."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # --standalone: torchrun hosts the rendezvous itself on a port it binds (no pick-then-rebind race); 127.0.0.1 because the
    # container hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={gpus}", os.path.abspath(__file__), *argv]
    return subprocess.run(cmd, env=env).returncode


class Bench:
    def __init__(self, args):
        self.args = args
        self.ranks = None
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        assert self.world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={self.world} in the environment"
        torch.cuda.set_device(self.local_rank)            # before the process group: every rank binds its own GPU
        self.dev = torch.device("cuda", self.local_rank)
        if self.world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=self.dev)          # RCCL on ROCm
        torch.set_num_threads(max(1, usable_cores() // max(1, self.world)))
        self.set_matmul(args.matmul)
        from dimsum_amd import _lib
        _lib.load()                                   # fail loudly if the HIP library is missing
        self.timer = ScanTimer()
        self.timer.install()
        if not getattr(args, "no_box_probe", False):
            box_probe(self.dev)

    @staticmethod
    def set_matmul(policy):
        from dimsum_amd import gemm
        torch.backends.cuda.matmul.allow_tf32 = policy in ("tf32", "fp16", "f16s")
        torch.backends.cudnn.allow_tf32 = policy in ("tf32", "fp16", "f16s")
        gemm.set_policy(policy if policy in ("fp16", "f16s") else "default")

    def fence(self):
        if self.world > 1:
            dist.barrier(device_ids=[self.local_rank])
        torch.cuda.synchronize()

    def timed(self, step, steps, warmup, time_scans=True):
        """`warmup` untimed steps, then exactly `steps` steps between two fences; max over ranks. -> seconds.
        time_scans=False: no per-launch HIP events (inference legs: the two branches of a block run on two streams there, and a
        kernel that shares the chip has no duration of its own -- see scan_roofline_pass)."""
        for _ in range(warmup):
            step()
        self.fence()
        self.timer.reset()
        self.timer.enabled = time_scans
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        issued = time.perf_counter() - t0
        if self.world > 1:
            torch.cuda.synchronize()
            own = time.perf_counter() - t0           # this rank's own wall time, before it waits for the others
        self.fence()
        elapsed = time.perf_counter() - t0
        self.timer.enabled = False
        self.ranks = None
        if self.world > 1:
            self.ranks = rank_summary(own, issued, steps, getattr(self, "rank_extra", None))
            tmax = torch.tensor([elapsed], device=self.dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = tmax.item()
        return elapsed

    def scan_roofline_pass(self, step, n=2, full_interface=False):
        """Inference runs the two branches of every block on two HIP streams (dimsum_amd/models_dim.py, the default): the scan of
        one branch overlaps the GEMMs of the other and its launch-to-launch time is no longer its own. The roofline of the scan
        kernel is therefore measured in a short pass of `n` more steps of the SAME step function on ONE stream
        (DIMSUM_BRANCH_STREAMS=0), right after the timed region, with one HIP-event pair per launch; the line says so.
        full_interface: the same pass with DIMSUM_SCAN_INFER_STORES=1, i.e. the launch with the reference interface's `out` / `x`
        stores that an inference call skips by default (SURVEY 8(d)'s full-interface pricing: the figure the 0.5 target is quoted on)."""
        from dimsum_amd.models_dim import branch_streams
        old = os.environ.get("DIMSUM_SCAN_INFER_STORES")
        if full_interface:
            os.environ["DIMSUM_SCAN_INFER_STORES"] = "1"
        try:
            with branch_streams(False):
                step()
                torch.cuda.synchronize()
                self.timer.reset()
                self.timer.enabled = True
                for _ in range(n):
                    step()
                torch.cuda.synchronize()
        finally:
            self.timer.enabled = False
            if full_interface:
                if old is None:
                    del os.environ["DIMSUM_SCAN_INFER_STORES"]
                else:
                    os.environ["DIMSUM_SCAN_INFER_STORES"] = old
        rf = self.timer.roofline("fwd")
        if rf is not None:
            rf["timed_in"] = (f"single-stream pass of {n} steps right after the timed region ({rf['launches_timed']} launches, HIP events at the "
                              "kernels' own begin / end); the timed region itself runs the two branches of every block on two HIP streams"
                              + ("; THIS pass with DIMSUM_SCAN_INFER_STORES=1: the launch keeps the reference interface's `out` / `x` stores, "
                                 "which the timed region's inference launches skip" if full_interface else ""))
        return rf

    def scan_rooflines(self, step, out):
        """`roofline`: the scan launch as the timed region runs it (inference: no `out` / `x` stores, priced with the bytes it moves =
        SURVEY 8(d)'s inference-only lower bound); `roofline_full_interface`: the same kernel with the reference interface's stores on
        (8(d)'s full formula), from a second single-stream pass -- the figure comparable across rounds and with the 0.5 target."""
        rf = self.scan_roofline_pass(step)
        if rf is not None:
            out["roofline"] = rf
            if "no out / x stores" in rf["kernel"]:
                rf["pricing"] = ("SURVEY 8(d) inference-only lower bound: the launch skips the `out` / `x` stores nothing reads, and is priced without them"
                                 + ("; dt_proj ALSO runs inside this launch on the matrix cores (delta = W_dt x_dbl[:R] per tile: one GEMM launch less per "
                                    "mixer), so the launch does not read delta: `achieved` / `frac` price the bytes that cross HBM; what SURVEY 8(d) would "
                                    "charge the scan it performs is `bytes_8d_of_the_scan_performed` / `frac_if_priced_by_8d` (not a bandwidth)" if "fused dt_proj" in rf["kernel"] else "")
                                 + ("; out_z leaves as block-scaled fp16 (2 bytes per element + one scale per 64 x 32 block: the operand of out_proj's single "
                                    "fp16 product)" if "block-scaled fp16" in rf["kernel"] else ""))
                full = self.scan_roofline_pass(step, full_interface=True)
                if full is not None:
                    full["pricing"] = "SURVEY 8(d) full interface (reads u, delta, z, B, C, A, D, delta_bias; writes out, out_z, x)"
                    out["roofline_full_interface"] = full

    @staticmethod
    def streams_note():
        from dimsum_amd.models_dim import branch_streams_enabled
        return ("one stream (DIMSUM_BRANCH_STREAMS=0)" if not branch_streams_enabled()
                else "two HIP streams per block (spatial || frequency branch), bit-identical to one stream")

    def inputs(self, batch, r):
        gen = torch.Generator(device=self.dev).manual_seed(0 * self.world + self.rank)      # sample_ddp.py:64 seeding rule
        x = torch.randn(batch, 4, r, r, device=self.dev, generator=gen)
        t = torch.rand(batch, device=self.dev, generator=gen)
        y = torch.randint(0, 1000, (batch,), device=self.dev, generator=gen)
        return x, t, y, gen

    def free(self):
        import gc
        gc.collect()
        torch.cuda.empty_cache()

    # ---- legs -----------------------------------------------------------------------------------------------------------
    def leg_fwd(self, model_name, image_size, batch, scan_type, steps, warmup, extra_precisions):
        a, r = self.args, image_size // 8
        model = build_model(model_name, self.dev, image_size, scan_type)
        x, t, y, _ = self.inputs(batch, r)
        if a.hip_graph:
            from dimsum_amd.hip_graph import GraphedForward
            graphed = GraphedForward(model)

            def step():
                return graphed(x, t, y)
        else:
            def step():
                with torch.no_grad():
                    return model(x, t, y)
        elapsed = self.timed(step, steps, warmup, time_scans=False)
        out = {"workload": f"{model_name} denoiser forward, {image_size}px (4x{r}x{r} latents, {(r // 2) ** 2} tokens), {batch} latents per GPU, "
                           + WEIGHTS + (f", scan_type={scan_type}" if scan_type != "none" else ""),
               "value": batch * self.world * steps / elapsed, "unit": "latents/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
               "warmup": warmup, "batch_per_gpu": batch, "launch": "hipGraph replay" if a.hip_graph else "eager",
               "branch_streams": "one stream (captured graph)" if a.hip_graph else self.streams_note()}
        if self.ranks:
            out["ranks"] = self.ranks
        if not a.hip_graph:                                                   # (a replayed graph makes no library calls to time)
            self.scan_rooflines(step, out)
        if extra_precisions and self.world == 1 and a.matmul in ("tf32", "f16s") and not a.hip_graph:
            # the same step under the other carrier of the reference's allow_tf32 policy, with exact-fp32 library GEMMs, under the emulated
            # TF32 arithmetic of the reference's own hardware path, and with the opt-in fp16-operand policy (never the headline).
            # (skipped under --hip-graph: a captured graph has the policy of its capture baked in)
            def dev(got, ref):
                d = (got.double() - ref.double()).abs()
                return {"max_over_max_abs": (d.max() / ref.abs().max()).item(), "rms_over_max_abs": (d.pow(2).mean().sqrt() / ref.abs().max()).item()}
            outs = {a.matmul: step()}
            self.set_matmul("fp32")
            dt = self.timed(step, 2, 2, time_scans=False) / 2
            out["fp32_exact_matmul"] = {"value_per_gpu": batch / dt, "ms_per_step": 1e3 * dt}
            ref = step()
            # the reference's own arithmetic, emulated: every torch matmul with operands rounded to TF32 (10 mantissa bits), fp32 sums
            # (the attention core and the patch-embedding conv stay exact fp32 in this run: it UNDERSTATES a real TF32 run's deviation)
            from dimsum_amd.utils.tf32_emulation import emulated_tf32
            with emulated_tf32():
                tf = step()
            other = "tf32" if a.matmul == "f16s" else "f16s"
            self.set_matmul(other)
            outs[other] = step()
            n1 = max(3, steps // 2)
            dt = self.timed(step, n1, 2, time_scans=False) / n1
            devs = {"f16s_single_product": dev(outs["f16s"], ref), "three_product_split_bf16": dev(outs["tf32"], ref),
                    "emulated_tf32_matmul_operands": dev(tf, ref)}
            what = {"f16s": "the large Linears (in_proj, qkv, proj, w12 + gate, w3) on scaled-fp16 operand images: ONE v_mfma_f32_16x16x32_f16 product per "
                            "element, fp32 accumulation, exact power-of-two row scales undone in the GEMM epilogue (dimsum_amd.gemm policy 'f16s'), the "
                            "attention fusion on its single-product fp16 kernel",
                    "tf32": "the same Linears on split-bf16 operand images: THREE bf16 MFMA products per element (hi.hi + hi.lo + lo.hi), fp32-class "
                            "(4e-6 rms): what `--matmul tf32` makes the headline (rounds 1-4)"}
            key = {"f16s": "tf32_single_product_f16s", "tf32": "tf32_three_product_split_bf16"}[other]
            out[key] = {"what": what[other] + "; NOT the headline value of this line", "value_per_gpu": batch / dt, "ms_per_step": 1e3 * dt, "steps": n1}
            out["deviation_vs_exact_fp32"] = dict(devs, headline=("f16s_single_product" if a.matmul == "f16s" else "three_product_split_bf16"),
                                                  what="max / rms |out - out_fp32| over max|out_fp32| of the denoiser output on the bench batch; exact fp32 = "
                                                       "allow_tf32 off (fp32 MFMA GEMMs, fp32 attention); emulated TF32 = the reference's hardware arithmetic "
                                                       "for every torch matmul (10-bit operand mantissas, fp32 sums), attention / conv left exact")
            self.set_matmul("fp16")
            got = step()
            dt = self.timed(step, 3, 1, time_scans=False) / 3
            out["fp16_operand_matmul_optin"] = {"value_per_gpu": batch / dt, "ms_per_step": 1e3 * dt,
                                                "max_abs_dev_over_max_abs_vs_exact_fp32": ((got - ref).abs().max() / ref.abs().max()).item()}
            self.set_matmul(a.matmul)
        elif self.world == 1 and a.matmul in ("tf32", "f16s") and not a.hip_graph and not a.no_fp32_leg:
            # (legs without the full extras: only the other carrier's timing, 3 steps)
            other = "tf32" if a.matmul == "f16s" else "f16s"
            self.set_matmul(other)
            dt = self.timed(step, 3, 2, time_scans=False) / 3
            out[{"f16s": "tf32_single_product_f16s", "tf32": "tf32_three_product_split_bf16"}[other]] = {
                "value_per_gpu": batch / dt, "ms_per_step": 1e3 * dt, "steps": 3,
                "what": f"dimsum_amd.gemm policy of `--matmul {other}` (see the headline leg's key of the same name); NOT this leg's value"}
            self.set_matmul(a.matmul)
        del model
        self.free()
        return out

    TRAIN_MATMUL = {"f16s": "allow_tf32=True (dimsum/train.py:20-21 turns TF32 on for training too) served as ONE fp16 MFMA product per element over scaled-fp16 operand "
                            "images in the forward, input-gradient and weight-gradient GEMMs (dW: row scales become per-reduction-row factors inside the TN kernel) and in the attention "
                            "backward pair (dout rows scaled by exact powers of two in-kernel; the attention forward under autograd keeps three split-bf16 products)",
                    "tf32": "allow_tf32=True served by the three-product split-bf16 images (fp32-class)", "fp32": "exact fp32", "fp16": "as tf32"}

    def leg_block(self, model_name, image_size, batch, steps, warmup):
        r = image_size // 8
        model, hidden = build_block(model_name, self.dev)
        _, _, _, gen = self.inputs(1, r)
        ntok = (r // 2) ** 2
        hs = torch.randn(batch, ntok, hidden, device=self.dev, generator=gen).requires_grad_()
        res = torch.randn(batch, ntok, hidden, device=self.dev, generator=gen).requires_grad_()
        cond = torch.randn(batch, hidden, device=self.dev, generator=gen).requires_grad_()
        dy = torch.randn(batch, ntok, hidden, device=self.dev, generator=gen)

        def step():
            for p_ in model.parameters():
                p_.grad = None
            hs.grad = res.grad = cond.grad = None
            out, res_out = model(hs, res, cond)
            torch.autograd.backward((out, res_out), (dy, dy))

        def grads():
            g = {"dx": hs.grad, "dc": cond.grad, "d w12": model.mlp.w12.weight.grad, "d w3": model.mlp.w3.weight.grad, "d qkv1": model.proj.qkv1.weight.grad,
                 "d in_proj (spatial)": model.spatial_mamba.mixer.in_proj.weight.grad, "d out_proj (freq)": model.freq_mamba.mixer.out_proj.weight.grad,
                 "d x_proj (spatial)": model.spatial_mamba.mixer.x_proj.weight.grad}
            return {k: v.detach().clone() for k, v in g.items()}
        elapsed = self.timed(step, steps, warmup)
        out = {"workload": f"ONE DiMBlockCombined({hidden}) of {model_name} (scan + Haar + attention fusion + gated MLP) forward+backward, "
                           f"{ntok} tokens, batch {batch}, " + WEIGHTS,
               "value": batch * self.world * steps / elapsed, "unit": "latents/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup, "batch_per_gpu": batch}
        for key, which in (("roofline", "fwd"), ("roofline_bwd", "bwd")):
            rf = self.timer.roofline(which)
            if rf is not None:
                out[key] = rf
        out["matmul"] = self.TRAIN_MATMUL[self.args.matmul]
        if self.args.matmul == "f16s" and not self.args.no_fp32_leg:
            # the other two arithmetic legs of the same step and the gradient deviations: max |g - g_exact| / max |g_exact| per checked gradient
            mine = grads()
            self.set_matmul("tf32")
            e3 = self.timed(step, 3, 1, time_scans=False)
            three = grads()
            self.set_matmul("fp32")
            step()
            torch.cuda.synchronize()
            exact = grads()
            self.set_matmul("f16s")
            dev = lambda g: {k: float((g[k] - exact[k]).abs().max() / exact[k].abs().max()) for k in exact}
            out["three_product_split_bf16"] = {"ms_per_step": 1e3 * e3 / 3, "what": self.TRAIN_MATMUL["tf32"]}
            out["gradient_deviation_vs_exact_fp32"] = {"single_product_f16s": dev(mine), "three_product_split_bf16": dev(three),
                                                       "what": "max |g - g_exact| / max |g_exact|, g_exact = the same step with exact-fp32 matmuls (allow_tf32 off)"}
            del mine, three, exact
        del model, hs, res, cond, dy
        self.free()
        return out

    def leg_sample(self, model_name, image_size, batch, nfe):
        """configs[3]: `nfe` Euler evaluations of dx/dt = model(x, t, y) on `batch` latents per GPU + ONE all-gather"""
        from dimsum_amd.sample_ddp import sample_batch
        a, r = self.args, image_size // 8
        model = build_model(model_name, self.dev, image_size)
        x, _, y, _ = self.inputs(batch, r)
        graphs = {} if a.hip_graph else None
        sample_batch(model, x[:8], y[:8], num_steps=2, world_size=1, gather=False)       # warm-up: allocator, GEMM heuristics
        res = {}

        stats = {}
        self.rank_extra = stats                     # (filled by the step, read by timed() after it: the all-gather's own time per rank)

        def step():
            res["out"] = sample_batch(model, x, y, num_steps=nfe, world_size=self.world, hip_graph=graphs, stats=stats if self.world > 1 else None)
        elapsed = self.timed(step, 1, 0, time_scans=False)
        self.rank_extra = None
        out = {"workload": f"{model_name} {nfe}-NFE fixed-step Euler flow-matching sampling, {image_size}px, {batch} latents per GPU "
                           f"(global batch {batch * self.world}), one all_gather_into_tensor of the final latents, " + WEIGHTS,
               "value": batch * self.world / elapsed, "unit": "samples/s", "nfe": nfe, "s_per_batch": elapsed,
               "ms_per_nfe": 1e3 * elapsed / nfe, "batch_per_gpu": batch,
               "seed_rule": f"global_seed * world + rank = 0 * {self.world} + {self.rank} (sample_ddp.py:64)", "gathered_shape": list(res["out"].shape), "finite": bool(torch.isfinite(res["out"]).all().item()),
               "launch": "hipGraph replay" if a.hip_graph else "eager",
               "branch_streams": "one stream (captured graph)" if a.hip_graph else self.streams_note()}
        if self.ranks:
            out["ranks"] = self.ranks
            assert self.ranks["gathered_block_equals_own_output"], "a rank's block of the gathered latents differs from its own output"
        if not a.hip_graph:
            tt = torch.full((batch,), 0.5, device=self.dev)

            def one_nfe():
                with torch.no_grad():
                    return model(x, tt, y)
            self.scan_rooflines(one_nfe, out)
        del model
        self.free()
        return out

    def leg_train(self, model_name, image_size, batch, steps, warmup):
        from dimsum_amd.train import build_training, train_step
        from dimsum_amd.transport import create_transport
        r = image_size // 8
        model = build_model(model_name, self.dev, image_size)
        x, _, y, _ = self.inputs(batch, r)
        ddp, ema, opt = build_training(model.train(), self.dev, 1e-4, self.world, [self.local_rank])
        transport = create_transport("GVP", "velocity")
        elapsed = self.timed(lambda: train_step(ddp, ema, opt, transport, x, y), steps, warmup)
        out = {"workload": f"{model_name} flow-matching training step (GVP velocity loss, backward, grad all-reduce, clip, AdamW, EMA), "
                           f"{batch} latents per GPU, " + WEIGHTS,
               "value": batch * self.world * steps / elapsed, "unit": "latents/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup, "batch_per_gpu": batch}
        rb = self.timer.roofline("bwd")
        if rb is not None:
            out["roofline_bwd"] = rb
        out["matmul"] = self.TRAIN_MATMUL[self.args.matmul]
        if self.args.matmul == "f16s" and not self.args.no_fp32_leg:
            self.set_matmul("tf32")
            e3 = self.timed(lambda: train_step(ddp, ema, opt, transport, x, y), 2, 1, time_scans=False)
            self.set_matmul("f16s")
            out["three_product_split_bf16"] = {"value": batch * self.world * 2 / e3, "ms_per_step": 1e3 * e3 / 2, "what": self.TRAIN_MATMUL["tf32"]}
        del ddp, ema, opt, model
        self.free()
        return out


_BOX = {}


def box_probe(dev):
    """This box's own memory system, measured in THIS process (boxes of the pool differ by 7-10 % on memory-bound kernels): a 1-GiB
    device copy (read + write) and a 3-stream add, GB/s by HIP events. Every roofline object carries `frac_of_box_copy` next to
    `frac`, so a fraction quoted on one box can be compared with one quoted on another."""
    if not _BOX:
        n = 1 << 28
        a, b2 = torch.empty(n, device=dev), torch.empty(n, device=dev)
        c = torch.empty(n, device=dev)
        a.normal_(); b2.normal_()
        def ev(fn, reps=10):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        _BOX.update(copy_GBps=2 * 4 * n / (ev(lambda: c.copy_(a)) * 1e-3) / 1e9, add_GBps=3 * 4 * n / (ev(lambda: torch.add(a, b2, out=c)) * 1e-3) / 1e9,
                    what="torch copy_ / add(out=) over 1 GiB fp32 operands, HIP events, this process, before the legs")
        del a, b2, c
        torch.cuda.empty_cache()
    return _BOX


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="DiM-L/2")
    ap.add_argument("--batch", type=int, default=256, help="latents per GPU per step (headline / single-mode runs)")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--scan-type", default="none", help="none (published configs) | zigma_8 | sweep_8 | jpeg_8: zigzag token "
                                                         "orders inside the mixers (BASELINE configs[4])")
    ap.add_argument("--mode", choices=["all", "fwd", "sample", "block", "train", "xl512"], default="all",
                    help="all (default): headline forward (configs[1]) + every other BASELINE config as an extra leg; fwd / "
                         "sample / block / xl512 / train: that leg alone on the headline keys (profiling runs)")
    ap.add_argument("--nfe", type=int, default=250)
    ap.add_argument("--sample-batch", type=int, default=128, help="latents per GPU of the sampling leg (configs[3]: 1024 over 8 GPUs)")
    ap.add_argument("--hip-graph", action="store_true", help="fwd / sample: replay the denoiser forward from a captured hipGraph "
                                                             "(dimsum_amd/hip_graph.py): for per-GPU batches below ~32, where eager is launch-bound")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-box-probe", action="store_true", help="skip the 1-GiB copy / add bandwidth probe at process start (kernel-trace runs: keeps its rows out of the stats)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the extra exact-fp32 / fp16-operand timings (for profiling runs)")
    ap.add_argument("--matmul", choices=["tf32", "fp32", "fp16", "f16s"], default="f16s",
                    help="how the reference's matmul policy (torch.backends.cuda.matmul.allow_tf32 = True, dimsum/train.py:20-21, "
                         "sample_ddp.py:56) is served. f16s (default, the headline since round 5) = TF32's own arithmetic -- 10-bit operand "
                         "mantissas, fp32 accumulation -- as ONE fp16 MFMA product per element over scaled-fp16 operand images (exact "
                         "power-of-two row scales restore TF32's exponent range; deviation from exact fp32 below an emulated-TF32 run's); "
                         "training legs run it too since round 6 (forward, input- and weight-gradient GEMMs). tf32 = three bf16 products per element (split-bf16, "
                         "4e-6 rms: fp32-class, the headline of rounds 1-4). fp32 = exact fp32 MFMA. fp16 = opt-in: plain fp16 operands "
                         "(TF32's mantissa, NOT its exponent range); never the headline.")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))           # before anything touches the GPU
    b = Bench(args)
    world, rank = b.world, b.rank
    policy = {"tf32": "allow_tf32=True like the reference (train.py:20-21); gfx950 split-bf16 path (3 bf16 products per fp32 product), 4e-6 rms rel err"
                      + ("; the inference GEMMs in_proj / qkv / proj / w12 / w3 take their left operands as split images written by the producer kernels, the training GEMMs of qkv / proj / w12 / w3 run forward and backward on such images (dimsum_amd/gemm.py, split3); on this package's MFMA kernel (dimsum_gemm_nt / dimsum_gemm_tn) the images travel as [hi | lo] pairs read as [hi | hi | lo] by K-tile aliasing: the same three products"
                         if os.environ.get("DIMSUM_SPLIT3", "1") != "0" else "; DIMSUM_SPLIT3=0: fp32 operands everywhere"),
              "fp32": "exact fp32",
              "f16s": "allow_tf32=True like the reference (train.py:20-21), served as the TF32-equivalent SINGLE product: the large Linears (in_proj, qkv, proj, "
                      "w12 + gate, w3) and the attention fusion's QK^T / PV on scaled-fp16 operand images (10-bit mantissas like TF32, exact power-of-two "
                      "row scales: no range loss; one v_mfma_f32_16x16x32_f16 per element, fp32 accumulation); split-bf16 (3 products, fp32-class) for "
                      "x_proj / dt_proj (small library GEMMs); training legs on the same carrier (gradient deviations under `block_fwdbwd`); deviation from exact fp32 on this line under `deviation_vs_exact_fp32`, "
                      "next to the emulated-TF32 run's and the three-product carrier's (`--matmul tf32`, the headline of rounds 1-4, timed as "
                      "`tf32_three_product_split_bf16`)",
              "fp16": "OPT-IN: fp16 operands (TF32 mantissa, fp16 exponent range) + fp32 accumulation for the large Linears, split-bf16 elsewhere"}[args.matmul]

    extras = {}
    if args.mode in ("all", "fwd"):
        head = b.leg_fwd(args.model, args.image_size, args.batch, args.scan_type, args.steps, args.warmup, extra_precisions=not args.no_fp32_leg)
        metric = "denoiser-fwd latents/sec"
    elif args.mode == "xl512":
        head = b.leg_fwd("DiM-XL/2", 512, 64 if args.batch == 256 else args.batch, "zigma_8" if args.scan_type == "none" else args.scan_type,
                         args.steps, args.warmup, extra_precisions=False)
        metric = "denoiser-fwd latents/sec (DiM-XL/2 512px, 8-way zigzag)"
    elif args.mode == "block":
        head = b.leg_block(args.model, args.image_size, args.batch, args.steps, args.warmup)
        metric = "block fwd+bwd latents/sec"
    elif args.mode == "train":
        head = b.leg_train(args.model, args.image_size, args.batch, args.steps, args.warmup)
        metric = "training latents/sec"
    else:
        head = b.leg_sample(args.model, args.image_size, args.batch if args.batch != 256 else args.sample_batch, args.nfe)
        head.update(steps=1, warmup=0, ms_per_step=1e3 * head["s_per_batch"])
        metric = f"{args.nfe}-NFE samples/sec"
    if args.mode == "all":
        # configs[3] at every N: the scaling runs then exercise the sampler and its one RCCL collective
        extras["sample_250nfe"] = b.leg_sample(args.model, args.image_size, args.sample_batch, args.nfe)
        if world == 1:
            extras["block_fwdbwd"] = b.leg_block(args.model, args.image_size, 256, 5, 2)                      # configs[2]
            extras["xl512_zigzag"] = b.leg_fwd("DiM-XL/2", 512, 64, "zigma_8", 5, 2, extra_precisions=False)   # configs[4]
            extras["train_step"] = b.leg_train(args.model, args.image_size, 64, 3, 2)

    if rank == 0:
        # key order: the contract's keys, then the graded objects (roofline, roofline_full_interface, cpu_baseline) with their long
        # explanations moved to `notes` at the end -- a reader (or a log tail cut at 2 KB) sees the numbers first
        notes = {"matmul_policy": policy}
        line = {"metric": metric, "value": head["value"], "unit": head["unit"], "n_gpus": world, "steps": head["steps"], "warmup": head["warmup"],
                "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": head["workload"], "global_batch": head["batch_per_gpu"] * world,
                           "parallelism": f"dp{world} (replicas, independent latents)", "launch": head.get("launch", "eager"),
                           "matmul_policy": args.matmul + " (see notes.matmul_policy)"}}
        for k in ("roofline", "roofline_full_interface", "roofline_bwd"):
            if k in head:
                rf = dict(head[k])
                for long_key in ("pricing", "timed_in"):
                    if long_key in rf:
                        notes[f"{k}.{long_key}"] = rf.pop(long_key)
                if isinstance(rf.get("valu"), dict) and "what" in rf["valu"]:
                    rf["valu"] = dict(rf["valu"])
                    notes["roofline.valu.what"] = rf["valu"].pop("what")
                line[k] = rf
        if world == 1 and not args.no_cpu_baseline and args.mode in ("all", "fwd", "sample"):
            cb = cpu_baseline(args.model, 4, args.image_size)
            # the SHAPE of the reference's own CPU path: its pure-PyTorch selective_scan_ref, at batch 16
            shaped = cpu_baseline(args.model, 16, args.image_size, runs=1, scan="torch_loop", warm=False)   # (33 s per run: the port's runs above warmed the process)
            line["cpu_baseline"] = cb
            extra_cpu = {"reference_shaped": shaped}
            if args.mode == "all":      # BASELINE configs[0], SURVEY 8(d): DiM-S/2, batch 4 on the same host cores
                extra_cpu["config0_S2_batch4"] = cpu_baseline("DiM-S/2", 4, args.image_size)
        else:
            extra_cpu = None
        line["box"] = dict(_BOX)
        line["dist"] = {"world_size": dist.get_world_size() if world > 1 else 1,
                        "backend": (dist.get_backend() + " (RCCL)") if world > 1 else "none (single process)"}
        for k in ("deviation_vs_exact_fp32", "fp32_exact_matmul", "tf32_single_product_f16s",
                  "tf32_three_product_split_bf16", "fp16_operand_matmul_optin", "nfe", "s_per_batch", "gathered_shape", "finite"):
            if k in head:
                line[k] = head[k]
        if "sample_250nfe" in extras:
            line["samples_per_sec_250nfe_measured"] = extras["sample_250nfe"]["value"]
        line.update(extras)
        if extra_cpu:
            line["cpu_baseline_more"] = extra_cpu
        line["notes"] = notes
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier(device_ids=[b.local_rank])     # nobody tears the communicator down while rank 0 is still reporting
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
