"""Data-parallel flow-matching sampling -- the counterpart of dimsum/sample_ddp.py:52-237 for the denoiser hot path.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI). Latents are independent, so the global batch
is sharded over ranks with NO collective during the NFE loop; rank r seeds `global_seed * world + r` and draws its own
z, y (sample_ddp.py:61-66, 161-165). Where the reference writes per-rank PNG files and barriers (:184-191), this build
collects the final latents with ONE `all_gather_into_tensor` per batch (2 MiB per rank at 128x4x32x32 fp32).
VAE decoding / PNG / FID are outside the hot path."""
import argparse
import os

import torch
import torch.distributed as dist

from . import gemm
from .transport import Sampler, create_transport


def _dist_ready():
    return dist.is_available() and dist.is_initialized()


@torch.no_grad()
def sample_batch(model, z, y, num_steps=250, sampling_method="euler", cfg_scale=None, path_type="GVP", world_size=None, gather=True,
                 hip_graph=None, stats=None):
    """Integrates dx/dt = model(x, t, y) from t = 0 (noise z) to t = 1 on linspace(0, 1, num_steps + 1): exactly
    `num_steps` function evaluations with Euler. With cfg_scale the batch is doubled like sample_ddp.py:168-173.
    Returns this rank's samples, or the all-gathered (world * B, C, H, W) tensor when a process group is up.
    stats (a dict, measurement only): receives the collective's own duration ("all_gather_ms": device events on a GPU, host clock
    after the call on a CPU process group) and "gathered_block_equals_own_output": this rank's block of the gathered tensor is its output."""
    fwd = model.forward if hasattr(model, "forward") else model
    fwd_cfg = getattr(model, "forward_with_cfg", None)
    if hip_graph is not None:                     # a dict owned by the caller: the captured graphs live across batches
        from .hip_graph import GraphedForward
        fwd = hip_graph.setdefault("fwd", GraphedForward(fwd))
        if fwd_cfg is not None:
            fwd_cfg = hip_graph.setdefault("cfg", GraphedForward(fwd_cfg))
    sampler = Sampler(create_transport(path_type, "velocity"))
    fn = sampler.sample_ode(sampling_method=sampling_method, num_steps=num_steps + 1)
    with gemm.frozen_weights():       # the weights are constant over the NFE loop: every weight image is built once per batch
        if cfg_scale is not None and cfg_scale > 1.0:
            n = z.shape[0]
            zz = torch.cat([z, z], 0)
            y_null = torch.full_like(y, getattr(model, "num_classes", 1000))
            out = fn(zz, fwd_cfg, return_trajectory=False, y=torch.cat([y, y_null], 0), cfg_scale=cfg_scale)[:n]
        else:
            out = fn(z, fwd, return_trajectory=False, y=y)
    ws = world_size if world_size is not None else (dist.get_world_size() if _dist_ready() else 1)
    if gather and _dist_ready() and (ws > 1 or gather == "force"):      # "force": also with one rank (collective bring-up tests)
        out = out.contiguous()
        full = torch.empty((ws * out.shape[0],) + tuple(out.shape[1:]), device=out.device, dtype=out.dtype)
        if stats is None:
            dist.all_gather_into_tensor(full, out)
            return full
        import time
        if out.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_gather_into_tensor(full, out)
            e1.record()
            e1.synchronize()
            stats["all_gather_ms"] = e0.elapsed_time(e1)
        else:
            t0 = time.perf_counter()
            dist.all_gather_into_tensor(full, out)
            stats["all_gather_ms"] = 1e3 * (time.perf_counter() - t0)
        r, n = dist.get_rank(), out.shape[0]
        stats["gathered_block_equals_own_output"] = bool(torch.equal(full[r * n:(r + 1) * n], out))
        return full
    return out


def load_denoiser_weights(model, path):
    """Loads a checkpoint the way the reference's samplers do (dimsum/download.py:17-37 `find_model`, sample_ddp.py:96-103):
    either a bare state_dict or the training container of train.py:355-373 -- {"model", "ema", "opt", "args", "epoch",
    "train_steps"} with "args" an argparse.Namespace -- from which the EMA weights are preferred. Strict key match: the
    module tree has the reference's state_dict keys, incl. the numerically dead cond_proj tensors (SURVEY finding 1).
    The container is a pickle (Namespace): not loadable under torch >= 2.6's weights_only default; checkpoints are trusted
    input here, as in the reference."""
    sd = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(sd, dict) and ("ema" in sd or "model" in sd):
        sd = sd.get("ema", sd.get("model"))
    model.load_state_dict(sd, strict=True)
    return model


def shard_range(total, rank, world):
    """indices [lo, hi) of the global sample list owned by `rank` (contiguous, remainder to the first ranks)"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def main(argv=None):
    from .create_model import create_model, published_config
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="DiM-L/2")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--num-classes", type=int, default=1000)
    ap.add_argument("--per-proc-batch-size", type=int, default=128)
    ap.add_argument("--num-fid-samples", type=int, default=1024)
    ap.add_argument("--num-sampling-steps", type=int, default=250)
    ap.add_argument("--sampling-method", default="euler")
    ap.add_argument("--cfg-scale", type=float, default=1.0)
    ap.add_argument("--path-type", default="GVP")
    ap.add_argument("--global-seed", type=int, default=0)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--out", default=None, help="rank 0 writes the gathered latents here (.pt)")
    ap.add_argument("--hip-graph", action="store_true", help="replay the denoiser forward from a captured hipGraph (small per-GPU batches)")
    ap.add_argument("--tf32", action=argparse.BooleanOptionalAction, default=True,
                    help="library-GEMM policy of the reference (sample_ddp.py:56,267-272); exact fp32 with --no-tf32")
    args, _ = ap.parse_known_args(argv)          # unknown flags are ignored like sample_ddp.py:369
    torch.backends.cuda.matmul.allow_tf32 = args.tf32
    torch.backends.cudnn.allow_tf32 = args.tf32

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)            # before the process group: every rank binds its own GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank, world = dist.get_rank(), dist.get_world_size()
    device = local_rank
    torch.manual_seed(args.global_seed * world + rank)
    model = create_model(published_config(args.model, args.image_size, args.num_classes)).to(device).eval()
    if args.ckpt:
        load_denoiser_weights(model, args.ckpt)                  # EMA preferred (download.py:26-27)
    r = args.image_size // 8
    n_iter = -(-args.num_fid_samples // (args.per_proc_batch_size * world))
    chunks = []
    graphs = {} if args.hip_graph else None
    for _ in range(n_iter):
        z = torch.randn(args.per_proc_batch_size, model.in_channels, r, r, device=device)
        y = torch.randint(0, args.num_classes, (args.per_proc_batch_size,), device=device)
        # the reference integrates on t = linspace(0, 1, --num-sampling-steps) (transport.py:379-386, integrators.py:98-111):
        # N grid points = N - 1 Euler evaluations; sample_batch counts evaluations
        chunks.append(sample_batch(model, z, y, max(1, args.num_sampling_steps - 1), args.sampling_method,
                                   args.cfg_scale if args.cfg_scale > 1.0 else None, args.path_type, hip_graph=graphs))
    dist.barrier()
    if rank == 0 and args.out:
        torch.save(torch.cat(chunks)[: args.num_fid_samples].cpu(), args.out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
