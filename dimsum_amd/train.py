"""Data-parallel flow-matching training step around the denoiser -- the counterpart of dimsum/train.py:299-321 (the step),
:180-203 (DDP, AdamW(weight_decay=0), EMA initialised from the synced weights), :351-376 (checkpoint container) and
:238-252 (resume). One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI): DDP's bucketed gradient
all-reduce (DiM-L/2: 460 M fp32 parameters = 1.84 GB per step) overlaps with the backward kernels.

Scope: the step itself on latents that are already on the GPU (synthetic by default). Datasets, the VAE encoder, FID
evaluation, plotting and logging-to-file of the reference's driver are outside the denoiser hot path (SURVEY.md 2.1
rows 15, 19, 23)."""
import argparse
import copy
import os
import time
from collections import OrderedDict

import torch
import torch.distributed as dist

from .transport import create_transport


@torch.no_grad()
def update_ema(ema_model, model, decay=0.9999):
    """ema <- decay * ema + (1 - decay) * model over all named parameters (train.py:55-64), as fused multi-tensor ops"""
    ema_params = OrderedDict(ema_model.named_parameters())
    ps, es = [], []
    for name, p in model.named_parameters():
        es.append(ema_params[name])
        ps.append(p.detach())
    torch._foreach_mul_(es, decay)
    torch._foreach_add_(es, ps, alpha=1 - decay)


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def train_step(model, ema, opt, transport, x, y, max_grad_norm=2.0, ema_decay=0.9999):
    """one optimisation step (train.py:311-321): velocity-matching loss -> backward (DDP all-reduce) -> clip -> AdamW -> EMA.
    `model` may be a DistributedDataParallel wrapper. Returns the (detached) mean loss."""
    loss = transport.training_losses(model, x, dict(y=y))["loss"].mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    opt.step()
    update_ema(ema, model.module if hasattr(model, "module") else model, ema_decay)
    return loss.detach()


def checkpoint_content(model, ema, opt, args, epoch, train_steps):
    """the container of train.py:355-373: {"epoch", "train_steps", "args", "model", "opt", "ema"}"""
    net = model.module if hasattr(model, "module") else model
    return {"epoch": epoch + 1, "train_steps": train_steps, "args": args, "model": net.state_dict(), "opt": opt.state_dict(),
            "ema": ema.state_dict()}


def load_checkpoint(path, model, ema=None, opt=None, map_location="cpu", lr=None):
    """resume (train.py:238-252): -> (init_epoch, train_steps). Inference loaders prefer the "ema" weights (download.py:26-27).
    `lr`: the run's --lr, written over the restored param groups like train.py:248-249 (a resumed run follows its own flag,
    not the checkpoint's). The container holds an argparse.Namespace ("args"), hence weights_only=False: trusted files only."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    net = model.module if hasattr(model, "module") else model
    net.load_state_dict(ck["model"], strict=True)
    if ema is not None:
        ema.load_state_dict(ck["ema"], strict=True)
    if opt is not None and "opt" in ck:
        opt.load_state_dict(ck["opt"])
        if lr is not None:
            for g in opt.param_groups:
                g["lr"] = lr
    return ck.get("epoch", 0), ck.get("train_steps", 0)


def build_training(model, device, lr=1e-4, world_size=1, device_ids=None):
    """-> (ddp_or_model, ema, opt): EMA is a frozen deep copy initialised from the weights every rank agrees on."""
    ema = copy.deepcopy(model).to(device)
    requires_grad(ema, False)
    if world_size > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, find_unused_parameters=False)
    fused = os.environ.get("DIMSUM_FUSED_ADAMW", "1") != "0" and torch.device(device).type == "cuda"
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=0, fused=fused)      # one multi-tensor kernel for the whole step
    update_ema(ema, model.module if hasattr(model, "module") else model, decay=0)
    ema.eval()
    return model, ema, opt


def main(argv=None):
    from .create_model import create_model, published_config
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="DiM-L/2")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--num-classes", type=int, default=1000)
    ap.add_argument("--global-batch-size", type=int, default=704)       # scripts/train.sh:86-112
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--max-grad-norm", type=float, default=2.0)
    ap.add_argument("--path-type", default="GVP")
    ap.add_argument("--prediction", default="velocity")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--global-seed", type=int, default=0)
    ap.add_argument("--log-every", type=int, default=5)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--save", default=None, help="rank 0 writes the checkpoint container here at the end")
    ap.add_argument("--tf32", action=argparse.BooleanOptionalAction, default=True)
    args, _ = ap.parse_known_args(argv)
    torch.backends.cuda.matmul.allow_tf32 = args.tf32          # train.py:20-21
    torch.backends.cudnn.allow_tf32 = args.tf32

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)            # before the process group: every rank binds its own GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank, world = dist.get_rank(), dist.get_world_size()
    assert args.global_batch_size % world == 0, "Batch size must be divisible by world size."
    device = local_rank
    torch.manual_seed(args.global_seed * world + rank)
    model = create_model(published_config(args.model, args.image_size, args.num_classes)).to(device)
    model, ema, opt = build_training(model, device, args.lr, world, [device])
    transport = create_transport(args.path_type, args.prediction)
    init_epoch, train_steps = (load_checkpoint(args.resume, model, ema, opt, lr=args.lr) if args.resume else (0, 0))
    model.train()
    r, b = args.image_size // 8, args.global_batch_size // world
    running, t0 = 0.0, time.time()
    for step in range(args.steps):
        x = torch.randn(b, 4, r, r, device=device)              # synthetic latents (already scaled like vae.encode * 0.18215)
        y = torch.randint(0, args.num_classes, (b,), device=device)
        running += train_step(model, ema, opt, transport, x, y, args.max_grad_norm).item()
        train_steps += 1
        if train_steps % args.log_every == 0:
            torch.cuda.synchronize()
            avg = torch.tensor(running / args.log_every, device=device)
            dist.all_reduce(avg, op=dist.ReduceOp.SUM)                # the reference's only data-path scalar all-reduce (:333-335)
            if rank == 0:
                print(f"(step={train_steps:07d}) Train Loss: {avg.item() / world:.4f}, "
                      f"Train Steps/Sec: {args.log_every / (time.time() - t0):.2f}", flush=True)
            running, t0 = 0.0, time.time()
    if rank == 0 and args.save:
        os.makedirs(os.path.dirname(os.path.abspath(args.save)), exist_ok=True)
        torch.save(checkpoint_content(model, ema, opt, vars(args), init_epoch, train_steps), args.save)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
