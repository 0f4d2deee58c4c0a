"""init_dct_kernel / init_idct_kernel of dimsum/dct_layer.py:6-84: frozen conv modules holding the 4x4 (k x k) DCT-II
basis. They exist for checkpoint compatibility (`dct_conv.weight`, `idct_conv.0.weight`); the transform is executed by
dimsum_amd.ops.token_ops (fused HIP kernel), which uses the same basis analytically."""
import math

import torch
import torch.nn as nn


def dct_basis(ksize, rsize):
    """basis[u + v*rsize, y, x] = (2 C_v C_u / ksize) cos((2y+1) v pi / 2k) cos((2x+1) u pi / 2k)"""
    k = torch.arange(ksize, dtype=torch.float64)
    cn = torch.ones(ksize, dtype=torch.float64)
    cn[0] = 1 / math.sqrt(2)
    cosm = torch.cos((2 * k[None, :] + 1) * k[:, None] * math.pi / (2 * ksize))          # [freq, pos]
    full = (2 * cn[:, None, None, None] * cn[None, :, None, None] / ksize) * cosm[:, None, :, None] * cosm[None, :, None, :]
    return full[:rsize, :rsize].reshape(rsize * rsize, ksize, ksize)                       # [(v u), y, x]


def init_dct_kernel(in_ch, ksize=8, rsize=2):
    conv = nn.Conv2d(in_ch, rsize ** 2 * in_ch, kernel_size=ksize, stride=ksize, padding=0, groups=in_ch, bias=False)
    w = dct_basis(ksize, rsize).float()[:, None].repeat(in_ch, 1, 1, 1)
    conv.weight = nn.Parameter(w, requires_grad=False)
    return conv


def init_idct_kernel(out_ch, ksize=8, rsize=2):
    conv = nn.Conv2d(rsize ** 2 * out_ch, ksize ** 2 * out_ch, kernel_size=1, stride=1, padding=0, groups=out_ch, bias=False)
    w = dct_basis(ksize, rsize).float().reshape(rsize * rsize, ksize * ksize).t()[:, :, None, None].repeat(out_ch, 1, 1, 1)
    conv.weight = nn.Parameter(w.contiguous(), requires_grad=False)
    return conv
