"""CPU tests of the training step host logic (dimsum_amd/train.py): EMA update, checkpoint container / resume round trip,
and a world_size-2 gloo DistributedDataParallel step (gradients averaged over ranks, replicas stay identical) with the CPU
oracle standing in for the HIP library."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dimsum_amd.models_dim import DiM
from dimsum_amd.train import build_training, checkpoint_content, load_checkpoint, train_step, update_ema
from dimsum_amd.transport import create_transport
from oracle.torch_backend import cpu_oracle_backend

KW = dict(img_resolution=8, in_channels=4, label_dropout=0.1, num_classes=10, scan_type="none", pe_type="ape", block_type="combined",
          cond_mamba=True, rms_norm=True, fused_add_norm=True, learnable_pe=True, use_attn_every_k_layers=2)


def _tiny(seed=0):
    torch.manual_seed(seed)
    m = DiM(depth=2, hidden_size=32, patch_size=2, **KW)
    with torch.no_grad():       # adaLN-zero init would make every gradient but the last layer's vanish
        for p in m.parameters():
            if p.numel() > 0 and torch.count_nonzero(p) == 0:
                p.normal_(0, 0.02)
    return m


def test_update_ema():
    m, e = _tiny(0), _tiny(1)
    before = {k: v.clone() for k, v in e.named_parameters()}
    update_ema(e, m, decay=0.9)
    for (k, pe), (_, pm) in zip(e.named_parameters(), m.named_parameters()):
        assert torch.allclose(pe, 0.9 * before[k] + 0.1 * pm, atol=1e-7)
    update_ema(e, m, decay=0)
    assert all(torch.equal(a, b) for a, b in zip(e.parameters(), m.parameters()))


def test_step_and_checkpoint_roundtrip(tmp_path):
    with cpu_oracle_backend():
        model, ema, opt = build_training(_tiny(), "cpu", lr=1e-3)
        tr = create_transport("GVP", "velocity")
        torch.manual_seed(3)
        x, y = torch.randn(2, 4, 8, 8), torch.randint(0, 10, (2,))
        w0 = model.final_layer.linear.weight.detach().clone()
        loss = train_step(model.train(), ema, opt, tr, x, y, max_grad_norm=2.0, ema_decay=0.5)
        assert torch.isfinite(loss) and not torch.equal(model.final_layer.linear.weight, w0)
        assert torch.allclose(ema.final_layer.linear.weight, 0.5 * w0 + 0.5 * model.final_layer.linear.weight, atol=1e-7)
        # the numerically dead cond_proj (SURVEY finding 1) gets no gradient
        cp = [p for n, p in model.named_parameters() if "cond_proj" in n]
        assert cp and all(p.grad is None or not p.grad.any() for p in cp)
        content = checkpoint_content(model, ema, opt, {"model": "tiny"}, epoch=4, train_steps=17)
        assert set(content) == {"epoch", "train_steps", "args", "model", "opt", "ema"}        # train.py:355-373
        path = tmp_path / "content.pth"
        torch.save(content, path)
        m2, e2, o2 = build_training(_tiny(5), "cpu", lr=1e-3)
        assert load_checkpoint(str(path), m2, e2, o2) == (5, 17)
        assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), model.state_dict().values()))
        assert all(torch.equal(a, b) for a, b in zip(e2.state_dict().values(), ema.state_dict().values()))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    with cpu_oracle_backend():
        model, ema, opt = build_training(_tiny(0), "cpu", lr=1e-3, world_size=world)
        tr = create_transport("GVP", "velocity")
        torch.manual_seed(100 + rank)                                   # rank-dependent data (train.py:150-152)
        x, y = torch.randn(2, 4, 8, 8), torch.randint(0, 10, (2,))
        loss = train_step(model.train(), ema, opt, tr, x, y)
        w = model.module.final_layer.linear.weight.detach().clone()
        g = model.module.final_layer.linear.weight.grad.detach().clone()
    q.put((rank, float(loss), w.numpy().copy(), g.numpy().copy()))      # by value: the worker may exit before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ddp_step():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, w0, g0), (_, l1, w1, g1) = res
    assert l0 != l1                                   # different data per rank
    assert (g0 == g1).all() and (w0 == w1).all() and g0.any()      # all-reduced gradients -> identical replicas


def test_sampler_loads_reference_style_container(tmp_path):
    """the reference's training container (train.py:355-373) stores `args` as an argparse.Namespace and both "model" and
    "ema"; its samplers take the EMA weights (download.py:26-27). A bare state_dict loads too; a wrong key set is refused."""
    import argparse
    from dimsum_amd.sample_ddp import load_denoiser_weights
    m, e = _tiny(0), _tiny(1)
    path = tmp_path / "content.pth"
    torch.save({"epoch": 3, "train_steps": 10, "args": argparse.Namespace(model="tiny", lr=1e-4), "model": m.state_dict(),
                "opt": {}, "ema": e.state_dict()}, path)
    got = load_denoiser_weights(_tiny(2), str(path))
    assert all(torch.equal(a, b) for a, b in zip(got.state_dict().values(), e.state_dict().values()))      # EMA, not "model"
    torch.save(m.state_dict(), path)
    got = load_denoiser_weights(_tiny(2), str(path))
    assert all(torch.equal(a, b) for a, b in zip(got.state_dict().values(), m.state_dict().values()))
    bad = dict(m.state_dict())
    bad.pop(next(iter(bad)))
    torch.save(bad, path)
    with pytest.raises(RuntimeError):
        load_denoiser_weights(_tiny(2), str(path))
