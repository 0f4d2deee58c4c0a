"""DWT_2D / IDWT_2D modules of dimsum/wavelet_layer.py:68-115 -- kept for their checkpoint buffers
(`dwt.w_ll/w_lh/w_hl/w_hh`, `idwt.filters`; Haar constants of PyWavelets 1.6.0). The transform itself is computed by
dimsum_amd.ops.token_ops (fused HIP kernel / differentiable torch butterflies), not by grouped convolutions."""
import math

import torch
import torch.nn as nn

_S = 1.0 / math.sqrt(2.0)
_HAAR = {"dec_lo": [_S, _S], "dec_hi": [-_S, _S], "rec_lo": [_S, _S], "rec_hi": [_S, -_S]}


def _outer(rows, cols):
    return (torch.tensor(cols).unsqueeze(0) * torch.tensor(rows).unsqueeze(1)).float()


class DWT_2D(nn.Module):
    def __init__(self, wave="haar"):
        super().__init__()
        assert wave == "haar", "DiMSUM only uses the Haar wavelet"
        lo, hi = _HAAR["dec_lo"][::-1], _HAAR["dec_hi"][::-1]
        for name, (r, c) in {"w_ll": (lo, lo), "w_lh": (hi, lo), "w_hl": (lo, hi), "w_hh": (hi, hi)}.items():
            self.register_buffer(name, _outer(r, c)[None, None])          # w_xy[i][j] = rows[i] * cols[j]


class IDWT_2D(nn.Module):
    def __init__(self, wave="haar"):
        super().__init__()
        assert wave == "haar"
        lo, hi = _HAAR["rec_lo"], _HAAR["rec_hi"]
        self.register_buffer("filters", torch.stack([_outer(lo, lo), _outer(hi, lo), _outer(lo, hi), _outer(hi, hi)])[:, None])
