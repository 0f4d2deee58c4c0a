"""GPU tests of the harness either side of the denoiser (e):
  * one flow-matching training step (dimsum/train.py:299-321: loss -> backward -> clip -> AdamW -> EMA) on the HIP path vs
    the same step on the CPU through the oracle backend, same weights / data / noise;
  * checkpoint container round trip on the GPU (train.py:351-376, 238-252);
  * RCCL bring-up in a fresh child process: init_process_group("nccl", world_size=1), sample_batch through the
    all_gather_into_tensor branch, tear-down."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import assert_close
from procedural import procedural_fill, seeded

pytestmark = pytest.mark.gpu
T = torch.from_numpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KW = dict(img_resolution=32, in_channels=4, label_dropout=0.0, num_classes=1000, learn_sigma=False, scan_type="none", pe_type="ape",
          block_type="combined", cond_mamba=True, scanning_continuity=False, drop_path=0.0, rms_norm=True, fused_add_norm=True,
          learnable_pe=True, use_final_norm=False, use_attn_every_k_layers=4, use_gated_mlp=True)


def _model():
    from dimsum_amd.models_dim import DiM
    m = DiM(depth=4, hidden_size=64, patch_size=2, **KW)
    procedural_fill(m, seed=3)
    return m


def _fixed_transport(t, x0):
    """create_transport("GVP", "velocity") whose (t, x0) draw is fixed: the reference draws t on the CPU and x0 on the
    data's device (transport.py:109-125), which no two devices reproduce"""
    from dimsum_amd.transport import create_transport
    tr = create_transport("GVP", "velocity")
    tr.sample = lambda x1: (t.to(x1), x0.to(x1), x1)
    return tr


def _one_step(dev, lr, decay):
    from dimsum_amd.train import build_training, train_step
    model, ema, opt = build_training(_model().to(dev), dev, lr=lr)
    x, y = T(seeded((4, 4, 32, 32), 81)).to(dev), torch.tensor([1, 22, 333, 999], device=dev)
    tr = _fixed_transport(T(seeded((4,), 82, kind="uniform")), T(seeded((4, 4, 32, 32), 83)))
    loss = train_step(model.train(), ema, opt, tr, x, y, max_grad_norm=2.0, ema_decay=decay)
    grads = {k: (None if p.grad is None else p.grad.detach().cpu().numpy()) for k, p in model.named_parameters()}
    return loss.item(), grads, {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}, \
        {k: v.detach().cpu().numpy() for k, v in ema.named_parameters()}, (model, ema, opt)


@pytest.mark.usefixtures("allow_torch_sdpa")      # hidden 64: head_dim 4 (conftest)
def test_train_step_hip_vs_cpu_oracle():
    """loss rtol 1e-4; every clipped gradient rtol 1e-3 + 2e-4 * max|ref| (the tolerance of the block-level gradient goldens);
    updated parameters and EMA: AdamW's first update is -lr * g / (|g| + 1e-8) ~ -lr * sign(g), so the two runs agree to
    1e-7 everywhere except where a gradient's sign is below its own rounding noise (|g| < 1e-6 max|g|); those elements are
    counted (< 0.5 %) and may differ by at most 2 lr."""
    from oracle.torch_backend import cpu_oracle_backend
    torch.backends.cuda.matmul.allow_tf32 = False
    lr, decay = 1e-4, 0.5
    loss_g, grads_g, params_g, ema_g, _ = _one_step("cuda", lr, decay)
    with cpu_oracle_backend():
        loss_c, grads_c, params_c, ema_c, _ = _one_step("cpu", lr, decay)
    assert abs(loss_g - loss_c) <= 1e-4 * abs(loss_c), (loss_g, loss_c)
    init = {k: v.detach().numpy() for k, v in _model().named_parameters()}
    n_tot = n_off = 0
    for k in params_c:
        if grads_c[k] is None:
            assert grads_g[k] is None or not grads_g[k].any(), k          # cond_proj: dead in the reference (SURVEY finding 1)
            assert np.array_equal(params_g[k], init[k])
            continue
        assert_close(grads_g[k], grads_c[k], 1e-3, 0, "grad " + k, scale_atol=2e-4)
        for got, ref in ((params_g[k], params_c[k]), (ema_g[k], ema_c[k])):
            d = np.abs(got.astype(np.float64) - ref)
            assert d.max() <= 2 * lr * (1 + 1e-3), (k, d.max())
            n_tot += d.size
            n_off += int((d > 1e-7 + 1e-6 * np.abs(ref)).sum())
        assert_close(ema_g[k], decay * init[k].astype(np.float64) + (1 - decay) * params_g[k], 1e-6, 1e-7, "ema relation " + k)
        assert not np.array_equal(params_g[k], init[k]), k               # every live parameter moved
    assert n_off <= 0.005 * n_tot, (n_off, n_tot)


@pytest.mark.usefixtures("allow_torch_sdpa")
def test_checkpoint_roundtrip_on_gpu(tmp_path):
    from dimsum_amd.train import build_training, checkpoint_content, load_checkpoint
    *_, (model, ema, opt) = _one_step("cuda", 1e-3, 0.9)
    path = str(tmp_path / "content.pth")
    torch.save(checkpoint_content(model, ema, opt, {"model": "tiny"}, epoch=2, train_steps=1), path)
    from dimsum_amd.models_dim import DiM
    m2, e2, o2 = build_training(DiM(depth=4, hidden_size=64, patch_size=2, **KW).cuda(), "cuda", lr=5e-4)
    assert load_checkpoint(path, m2, e2, o2, map_location="cuda", lr=5e-4) == (3, 1)
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), model.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(e2.state_dict().values(), ema.state_dict().values()))
    assert all(g["lr"] == 5e-4 for g in o2.param_groups)                  # train.py:248-249: the run's own --lr wins
    s1, s2 = opt.state_dict()["state"], o2.state_dict()["state"]
    assert s1.keys() == s2.keys() and all(torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) for k in s1)


_CHILD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests", "golden"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["DIMSUM_ALLOW_TORCH_SDPA"] = "1"      # hidden 64: head_dim 4 has no MFMA attention kernel
from dimsum_amd.models_dim import DiM
from dimsum_amd.sample_ddp import sample_batch
from procedural import procedural_fill
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
kw = dict(img_resolution=32, in_channels=4, label_dropout=0.15, num_classes=1000, scan_type="none", pe_type="ape", block_type="combined",
          cond_mamba=True, rms_norm=True, fused_add_norm=True, learnable_pe=True, use_attn_every_k_layers=4)
m = DiM(depth=4, hidden_size=64, patch_size=2, **kw).eval(); procedural_fill(m, seed=3); m = m.cuda()
g = torch.Generator(device="cuda").manual_seed(0)
z = torch.randn(8, 4, 32, 32, device="cuda", generator=g); y = torch.randint(0, 1000, (8,), device="cuda", generator=g)
local = sample_batch(m, z, y, num_steps=5, gather=False)
full = sample_batch(m, z, y, num_steps=5, gather="force")          # through dist.all_gather_into_tensor on RCCL
assert full.shape == (8, 4, 32, 32) and torch.equal(full, local) and torch.isfinite(full).all()
t = torch.ones(4, device="cuda"); dist.all_reduce(t); assert torch.equal(t, torch.ones(4, device="cuda"))
dist.barrier(device_ids=[0]); dist.destroy_process_group()
print("NCCL_WORLD1_OK")
"""


def test_rccl_world1_sample_batch_all_gather_in_child_process():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-c", _CHILD, ROOT, str(port)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
