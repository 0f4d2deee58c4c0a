"""Mamba / CondMamba mixers -- same constructor arguments, parameter names (state_dict keys) and forward contract as
mamba/mamba_ssm/modules/mamba_simple.py (Mamba :42-297, CondMamba :438-701), running on the HIP operators of
dimsum_amd.ops. Only the diffusion path exists here: the autoregressive `step`/inference-cache API is decode-only and
out of scope.

Differences that do not change results:
  * `cond_proj(c)` is numerically dead in the reference (its output only donates a buffer to the conv kernel,
    causal_conv1d.cpp:326-329). The parameter is kept (checkpoint compatibility); under grad mode its (batch, d_inner)
    output is threaded through the autograd Function so the graph has the same edges (and the same `None` gradient),
    but the (batch, d_inner, seqlen) expand+copy of mamba_simple.py:589 -- 268 MB per call at DiM-L/2, batch 256 --
    is never materialised.
  * the zigzag gather / inverse gather (mamba_simple.py:627-657) are index_selects on the token axis.
"""
import math

import torch
import torch.nn as nn

from .. import gemm
from ..ops.selective_scan_interface import (mamba_inner_fn_cond, mamba_inner_fn_no_out_proj_cond)

_ZIGZAG = ("zigma", "sweep", "jpeg")


class _MambaBase(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True,
                 layer_idx=None, device=None, dtype=None, scan_type="none", d_cond=None, **kwargs):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path, self.layer_idx, self.scan_type, self.d_cond = use_fast_path, layer_idx, scan_type, d_cond

        self.in_proj = nn.Linear(d_model, self.d_inner * 2, bias=bias, **fk)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv, groups=self.d_inner,
                                padding=d_conv - 1, **fk)
        self.activation = "silu"
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + d_state * 2, bias=False, **fk)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        if d_cond is not None:
            self.cond_proj = nn.Linear(d_cond, self.d_inner, bias=True, **fk)
        self._init_dt(self.dt_proj, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, fk)
        self.A_log = self._s4d_real(device)
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        if scan_type == "v2":       # bidirectional twin (mamba_simple.py:529-553)
            self.A_b_log = self._s4d_real(device)
            self.conv1d_b = nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv, groups=self.d_inner,
                                      padding=d_conv - 1, **fk)
            self.x_proj_b = nn.Linear(self.d_inner, self.dt_rank + d_state * 2, bias=False, **fk)
            self.dt_proj_b = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
            self._init_dt(self.dt_proj_b, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, fk)
            self.D_b = nn.Parameter(torch.ones(self.d_inner, device=device))
            self.D_b._no_weight_decay = True
        else:
            self.A_b_log = self.conv1d_b = self.x_proj_b = self.dt_proj_b = self.D_b = None
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)
        self.register_buffer("zigzag_paths", kwargs.get("zigzag_paths", None))
        self.register_buffer("zigzag_paths_reverse", kwargs.get("zigzag_paths_reverse", None))

    def _s4d_real(self, device):
        A = torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device).repeat(self.d_inner, 1).contiguous()
        p = nn.Parameter(torch.log(A))      # kept in fp32
        p._no_weight_decay = True
        return p

    def _init_dt(self, proj, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, fk):
        """dt_proj initialised so that softplus(bias) is log-uniform in [dt_min, dt_max] (mamba_simple.py:494-512)."""
        std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(proj.weight, -std, std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(self.d_inner, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = dt.clamp(min=dt_init_floor)
        with torch.no_grad():
            proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))      # inverse softplus
        proj.bias._no_reinit = True

    def _is_zigzag(self):
        return self.scan_type.startswith(_ZIGZAG)

    def takes_image(self):
        """whether `x3` (the input as a split-bf16 operand image, gemm.py) can replace hidden_states: not when this mixer
        gathers its tokens itself"""
        return not (self._is_zigzag() and not getattr(self, "_zigzag_folded", False))

    def _mix(self, hidden_states, cond, x3=None):
        bsz, L, _ = (hidden_states if x3 is None else x3).shape
        own_gather = self._is_zigzag() and not getattr(self, "_zigzag_folded", False)   # folded: the enclosing block's token
        if own_gather:                                                                  # tables already include the path
            assert x3 is None
            # xz[..., j] = xz[..., perm[j]]: permuting the columns of xz == permuting the tokens before in_proj
            hidden_states = hidden_states.index_select(1, self.zigzag_paths[self.layer_idx])
        # in_proj with the transpose fused: (2D, d_model) @ (d_model, B*L) viewed as (B, 2D, L) -- d-major, no copy
        conv_done = False
        if x3 is not None:
            # inference on operand images: where a 256-token tile of the GEMM holds whole sequences, the mixer's causal conv1d + SiLU runs in
            # in_proj's epilogue (csrc/gemm_nt_kernel.hpp, kEpiF32Conv): xz[:, :d_inner] then already IS the conv output and
            # mamba_inner_fn skips the conv kernel (one launch and a read + write of (b, d_inner, l) fp32 less per mixer)
            cw = self.conv1d.weight
            if (self.scan_type != "v2" and self.in_proj.bias is None and not torch.is_grad_enabled() and cond is None and cw.dtype == torch.float32
                    and self.d_inner % 256 == 0 and 256 % L == 0 and L % 4 == 0):
                xz, conv_done = gemm.matmul_wx_split3(self.in_proj.weight, x3.reshape(bsz * L, -1),
                                                      conv=(cw.reshape(cw.shape[0], cw.shape[-1]), self.conv1d.bias, L))
            else:
                xz = gemm.matmul_wx_split3(self.in_proj.weight, x3.reshape(bsz * L, -1))
            xz = xz.view(2 * self.d_inner, bsz, L).permute(1, 0, 2)
        else:
            xz = gemm.matmul_wx(self.in_proj.weight, hidden_states.reshape(bsz * L, -1).t()).view(2 * self.d_inner, bsz, L).permute(1, 0, 2)
        if self.in_proj.bias is not None:
            xz = xz + self.in_proj.bias.to(xz.dtype).view(1, -1, 1)
        # recomputed on every call (two tiny launches): a cached copy could not see in-place parameter updates made through
        # `.data` (EMA, load_state_dict), which do not bump the version counter, and would be frozen into a captured hipGraph
        A = gemm.neg_exp(self.A_log)
        if self.scan_type == "v2":
            A_b = gemm.neg_exp(self.A_b_log)
            out = mamba_inner_fn_no_out_proj_cond(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                                  self.dt_proj.weight, A, None, None, self.D.float(),
                                                  delta_bias=self.dt_proj.bias.float(), delta_softplus=True, init_states=cond)
            out_b = mamba_inner_fn_no_out_proj_cond(xz.flip([-1]), self.conv1d_b.weight, self.conv1d_b.bias,
                                                    self.x_proj_b.weight, self.dt_proj_b.weight, A_b, None, None,
                                                    self.D_b.float(), delta_bias=self.dt_proj_b.bias.float(),
                                                    delta_softplus=True, init_states=cond)
            y = (out + out_b.flip([-1])).transpose(1, 2)
            return nn.functional.linear(y, self.out_proj.weight, self.out_proj.bias)
        out = mamba_inner_fn_cond(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight,
                                  self.out_proj.weight, self.out_proj.bias, A, None, None, self.D.float(),
                                  delta_bias=self.dt_proj.bias.float(), delta_softplus=True, init_states=cond, conv_done=conv_done)
        if own_gather:
            out = out.index_select(1, self.zigzag_paths_reverse[self.layer_idx])
        return out

    def allocate_inference_cache(self, *a, **k):
        raise NotImplementedError("autoregressive decode caches are outside the denoiser hot path")


class Mamba(_MambaBase):
    def forward(self, hidden_states, inference_params=None, x3=None):
        """hidden_states: (B, L, D) -> (B, L, D).  x3: the input as a split-bf16 operand image instead (inference, gemm.py)."""
        assert inference_params is None, "autoregressive decode is outside the denoiser hot path"
        return self._mix(hidden_states, None, x3=x3)


class CondMamba(_MambaBase):
    def forward(self, hidden_states, cond_emb=None, inference_params=None, x3=None):
        """hidden_states: (B, L, D), cond_emb: (B, d_cond) -> (B, L, D). See the module docstring about cond_proj.
        x3: the input as a split-bf16 operand image instead (inference, gemm.py)."""
        assert inference_params is None, "autoregressive decode is outside the denoiser hot path"
        cond = None
        if cond_emb is not None and torch.is_grad_enabled() and self.d_cond is not None:
            cond = self.cond_proj(cond_emb)       # (B, d_inner): graph edge only, never read by a kernel
        return self._mix(hidden_states, cond, x3=x3)
