"""small host utilities"""
import torch


@torch.no_grad()
def rerandomize_zeros(model, std=0.02, seed=0):
    """The reference initialisation is adaLN-zero + zero final layer, so a freshly initialised DiM outputs exactly 0
    (dimsum/models_dim.py:1762-1770; SURVEY finding 5). Benchmarks with random-init weights re-draw every all-zero
    parameter from N(0, std^2) so that no block degenerates to the identity."""
    g = torch.Generator().manual_seed(seed)
    for _, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
        if p.numel() > 0 and torch.count_nonzero(p) == 0:
            p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device, p.dtype))
    return model
