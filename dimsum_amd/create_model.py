"""create_model(args) -- argparse namespace -> DiM constructor, like dimsum/create_model.py:5-38 (DiT baselines are not
part of this build)."""
from .models_dim import DiM_models


def create_model(config):
    if "DiM" not in config.model:
        raise NotImplementedError(f"{config.model}: only the DiM family is part of the MI355X hot-path build")
    return DiM_models[config.model](
        img_resolution=config.image_size // 8, in_channels=config.num_in_channels, label_dropout=config.label_dropout,
        num_classes=config.num_classes, gated_linear_unit=getattr(config, "gated_linear_unit", True),
        routing_mode=getattr(config, "routing_mode", "top1"), num_moe_experts=getattr(config, "num_moe_experts", 8),
        is_moe=getattr(config, "is_moe", False), learn_sigma=config.learn_sigma, scan_type=config.bimamba_type,
        pe_type=config.pe_type, block_type=config.block_type, cond_mamba=config.cond_mamba,
        scanning_continuity=config.scanning_continuity, enable_fourier_layers=config.enable_fourier_layers,
        drop_path=config.drop_path, rms_norm=config.rms_norm, fused_add_norm=config.fused_add_norm,
        learnable_pe=config.learnable_pe, use_final_norm=config.use_final_norm,
        use_attn_every_k_layers=config.use_attn_every_k_layers, use_gated_mlp=not config.not_use_gated_mlp)


def published_config(model="DiM-L/2", image_size=256, num_classes=1000, **over):
    """flags of scripts/train.sh / scripts/eval.sh as an argparse-like namespace"""
    from types import SimpleNamespace
    cfg = dict(model=model, image_size=image_size, num_in_channels=4, label_dropout=0.15, num_classes=num_classes,
               learn_sigma=False, bimamba_type="none", pe_type="ape", block_type="combined", cond_mamba=True,
               scanning_continuity=False, enable_fourier_layers=False, drop_path=0.0, rms_norm=True, fused_add_norm=True,
               learnable_pe=True, use_final_norm=False, use_attn_every_k_layers=4, not_use_gated_mlp=False)
    cfg.update(over)
    return SimpleNamespace(**cfg)
