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


# ---- explicit accounting of torch-library paths ---------------------------------------------------------------------------
# The hot path runs on libdimsum_hip.so. A few module VARIANTS that no published config enables (qk_norm / swap_k /
# attention dropout in CrossAttentionFusion) have no HIP kernel and run on torch's SDPA; shapes the MFMA attention
# kernels are not instantiated for (head_dim outside {24, 32, 48, 64, 72}, non-fp32 activations) are an ERROR unless
# DIMSUM_ALLOW_TORCH_SDPA=1. Every such call is counted here and announced once, so that "no silent fallback" is checkable:
# tests assert torch_path_counts() stays empty on the published configs.
_torch_paths = {}


def note_torch_path(what, required_opt_in=False):
    import os
    import warnings
    if required_opt_in and os.environ.get("DIMSUM_ALLOW_TORCH_SDPA", "0") != "1":
        raise RuntimeError(f"dimsum_amd: {what} has no HIP kernel in this build (the MFMA attention kernels cover fp32 activations with "
                           "head_dim in {24, 32, 48, 64, 72}). Set DIMSUM_ALLOW_TORCH_SDPA=1 to run it on torch's "
                           "scaled_dot_product_attention instead (counted in dimsum_amd.utils.torch_path_counts()).")
    if what not in _torch_paths:
        warnings.warn(f"dimsum_amd: {what} runs on torch's scaled_dot_product_attention, not on libdimsum_hip.so", stacklevel=3)
    _torch_paths[what] = _torch_paths.get(what, 0) + 1


def torch_path_counts():
    """{description: calls} of everything that ran on a torch-library path instead of a HIP kernel of this build"""
    return dict(_torch_paths)
