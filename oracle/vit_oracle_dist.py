"""ORACLE (two-stream "stochastic" model) — test infrastructure, NOT product code.

CPU fp32 restatement of DistVisionTransformerForCyclicalTraining (modeling_cyclical_dist.py:14-165),
its two-stream Block / Wasserstein Attention (modeling_finetune_dist.py:15-59, 61-179),
wasserstein_distance_matmul (uncertainty_evaluations.py:276-294), WassersteinLoss
(distloss.py:7-30, 73-79) and the `stochastic=True` branches of train_one_epoch
(engine_for_cyclical.py:69-86, 125-126, 152-161).  Pinned to reference-generated goldens
(tests/golden/dist_*.npz via tools/gen_golden.py); checked by tests/test_oracle_golden.py.

Reference quirks that are part of the contract and reproduced here:
  * the covariance stream's QKV uses the SAME `qkv.weight` as the mean stream; `cov_qkv.weight`
    exists in the state dict but never receives a gradient (EMA still averages it);
  * norm1 / norm2 / mlp / gamma_1 / gamma_2 and the final norm are shared by both streams;
  * q is scaled, cov_q is not; attention probabilities are squared for the covariance output;
  * four independent drop-path draws per block; no position embedding.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import vit_oracle as vo

Tensor = torch.Tensor


def param_shapes(cfg: vo.VitConfig) -> Dict[str, tuple]:
    """State-dict names/shapes in the reference's registration order (115,778,640 params for ViT-B)."""
    C, Hd, P = cfg.embed_dim, cfg.hidden, cfg.patch_size
    s: Dict[str, tuple] = {"cls_token": (1, 1, C), "cov_cls_token": (1, 1, C), "mask_token": (1, 1, C),
                           "cov_mask_token": (1, 1, C),
                           "patch_embed.proj.weight": (C, cfg.in_chans, P, P), "patch_embed.proj.bias": (C,),
                           "cov_patch_embed.proj.weight": (C, cfg.in_chans, P, P), "cov_patch_embed.proj.bias": (C,)}
    if cfg.use_shared_rel_pos_bias:
        s["rel_pos_bias.relative_position_bias_table"] = ((2 * cfg.grid - 1) ** 2 + 3, cfg.num_heads)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s[b + "gamma_1"] = (C,)
        s[b + "gamma_2"] = (C,)
        s[b + "norm1.weight"] = (C,)
        s[b + "norm1.bias"] = (C,)
        for n in ("q_bias", "v_bias", "cov_q_bias", "cov_v_bias"):
            s[b + "attn." + n] = (C,)
        s[b + "attn.qkv.weight"] = (3 * C, C)
        s[b + "attn.cov_qkv.weight"] = (3 * C, C)
        s[b + "attn.proj.weight"] = (C, C)
        s[b + "attn.proj.bias"] = (C,)
        s[b + "attn.cov_proj.weight"] = (C, C)
        s[b + "attn.cov_proj.bias"] = (C,)
        s[b + "norm2.weight"] = (C,)
        s[b + "norm2.bias"] = (C,)
        s[b + "mlp.fc1.weight"] = (Hd, C)
        s[b + "mlp.fc1.bias"] = (Hd,)
        s[b + "mlp.fc2.weight"] = (C, Hd)
        s[b + "mlp.fc2.bias"] = (C,)
    s["norm.weight"] = (C,)
    s["norm.bias"] = (C,)
    for n in ("lm_head", "cov_lm_head"):
        s[n + ".weight"] = (C, C)
        s[n + ".bias"] = (C,)
    return s


def wasserstein_distance_matmul(mean1: Tensor, cov1: Tensor, mean2: Tensor, cov2: Tensor) -> Tensor:
    """uncertainty_evaluations.py:276-294: pairwise squared 2-Wasserstein distance between diagonal
    Gaussians after a sigmoid on every input (sqrt of the covariances clamped at 1e-24)."""
    m1, m2, c1, c2 = torch.sigmoid(mean1), torch.sigmoid(mean2), torch.sigmoid(cov1), torch.sigmoid(cov2)
    ret = (m1 ** 2).sum(-1, keepdim=True) + (m2 ** 2).sum(-1, keepdim=True).transpose(-1, -2) - 2 * m1 @ m2.transpose(-1, -2)
    s1, s2 = torch.sqrt(c1.clamp(min=1e-24)), torch.sqrt(c2.clamp(min=1e-24))
    cov = c1.sum(-1, keepdim=True) + c2.sum(-1, keepdim=True).transpose(-1, -2) - 2 * s1 @ s2.transpose(-1, -2)
    return ret + cov


def attention(p, pre, cfg, x, cov_x, bias, keep=None):
    """modeling_finetune_dist.py:111-179."""
    B, N, C = x.shape
    H, d = cfg.num_heads, cfg.head_dim
    zeros = torch.zeros_like(p[pre + "v_bias"])
    qkv = x @ p[pre + "qkv.weight"].t() + torch.cat((p[pre + "q_bias"], zeros, p[pre + "v_bias"]))
    q, k, v = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
    cqkv = F.elu(cov_x @ p[pre + "qkv.weight"].t() + torch.cat((p[pre + "cov_q_bias"], zeros, p[pre + "cov_v_bias"]))) + 1
    cq, ck, cv = cqkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
    q = q * (d ** -0.5)
    a = torch.sigmoid(-wasserstein_distance_matmul(q, cq, k, ck) + 1e-24)
    a = (a + bias).softmax(dim=-1)
    if keep is not None:
        a = a * keep
    mean = (a @ v).transpose(1, 2).reshape(B, N, C)
    cov = ((a ** 2) @ cv).transpose(1, 2).reshape(B, N, C)
    return (mean @ p[pre + "proj.weight"].t() + p[pre + "proj.bias"],
            cov @ p[pre + "cov_proj.weight"].t() + p[pre + "cov_proj.bias"])


def block(p, i, cfg, xm, xc, bias, drop: Optional["DistDropState"]):
    """modeling_finetune_dist.py:41-59 (gamma branch): shared norms / mlp / gammas, four drop-path draws."""
    b = f"blocks.{i}."
    ln = lambda t, n: vo.layer_norm(t, p[b + n + ".weight"], p[b + n + ".bias"], cfg.ln_eps)  # noqa: E731
    keep = drop.attn[i] if drop and drop.attn else None
    m, c = attention(p, b + "attn.", cfg, ln(xm, "norm1"), ln(xc, "norm1"), bias, keep)
    dp = (lambda k: drop.path[k][i].reshape(-1, 1, 1) if drop and drop.path and drop.path[k][i] is not None else 1.0)
    xm = xm + dp(0) * (p[b + "gamma_1"] * m)
    fm = dp(1) * (p[b + "gamma_2"] * vo.mlp(p, b + "mlp.", ln(xm, "norm2")))
    xc = xc + dp(2) * (p[b + "gamma_1"] * c)
    fc = dp(3) * (p[b + "gamma_2"] * vo.mlp(p, b + "mlp.", ln(xc, "norm2")))
    return xm + fm, xc + fc


class DistDropState:
    """Injected randomness: path[k][layer] (B,) multipliers for the 4 draws (mean attn, mean mlp, cov attn,
    cov mlp); attn[layer] (B,H,N,N) dropout multipliers."""

    def __init__(self, path=None, attn=None):
        self.path, self.attn = path, attn


def forward_features(p, cfg, x, mask, layer_results, drop=None):
    """modeling_cyclical_dist.py:106-144."""
    B = x.shape[0]
    pe = {"patch_embed.proj.weight": p["patch_embed.proj.weight"], "patch_embed.proj.bias": p["patch_embed.proj.bias"]}
    xm = vo.patch_embed(pe, cfg, x)
    pc = {"patch_embed.proj.weight": p["cov_patch_embed.proj.weight"], "patch_embed.proj.bias": p["cov_patch_embed.proj.bias"]}
    xc = vo.patch_embed(pc, cfg, x)
    if mask is not None:
        w = mask.reshape(B, -1, 1).to(xm.dtype)
        xm = xm * (1 - w) + p["mask_token"] * w
        xc = xc * (1 - w) + p["cov_mask_token"] * w
    xm = torch.cat((p["cls_token"].expand(B, -1, -1), xm), dim=1)
    xc = torch.cat((p["cov_cls_token"].expand(B, -1, -1), xc), dim=1)
    bias = vo.rel_pos_bias(p, cfg)
    zm, zc = [], []
    for i in range(cfg.depth):
        xm, xc = block(p, i, cfg, xm, xc, bias, drop)
        if layer_results == "end":
            zm.append(xm)
            zc.append(xc)
    if layer_results:
        return zm, zc
    n = lambda t: vo.layer_norm(t, p["norm.weight"], p["norm.bias"], cfg.ln_eps)  # noqa: E731
    return n(xm), n(xc)


def forward(p, cfg, x, mask, return_all_tokens=False, layer_results=None, drop=None):
    """modeling_cyclical_dist.py:146-165."""
    m, c = forward_features(p, cfg, x, mask, layer_results, drop)
    if layer_results:
        return [z[:, 1:] for z in m], [z[:, 1:] for z in c]
    m, c = m[:, 1:], c[:, 1:]
    if not return_all_tokens:
        sel = mask.flatten().bool()
        m, c = m.reshape(-1, m.shape[-1])[sel], c.reshape(-1, c.shape[-1])[sel]
    return m @ p["lm_head.weight"].t() + p["lm_head.bias"], c @ p["cov_lm_head.weight"].t() + p["cov_lm_head.bias"]


def wasserstein_loss(mean_out, cov_out, mean_t, cov_t, lam):
    """distloss.py:13-30 + wasserstein_distance :73-79 (global-max normalisations make it non-local)."""
    mo, co, mt, ct = torch.sigmoid(mean_out), torch.sigmoid(cov_out), torch.sigmoid(mean_t), torch.sigmoid(cov_t)
    pos = ((mo - mt) ** 2).sum(-1)
    s1, s2 = torch.sqrt(co.clamp(min=1e-24)), torch.sqrt(ct.clamp(min=1e-24))
    pos = pos + ((s1 - s2) ** 2).sum(-1)
    pos = pos / pos.abs().max()
    loss = -torch.log(torch.sigmoid(-pos + 1e-24))
    loss = loss / loss.abs().max()
    return loss.sum() * lam


def train_step(params, ema, m, v, cfg, hp: vo.StepHParams, samples, mask, step, lam=1e-5, drop=None):
    """engine_for_cyclical.py:45-186 with stochastic=True."""
    with torch.no_grad():
        tm, tc = forward(ema, cfg, samples, None, True, hp.layer_results)
        # the batch- / instance-norm variants act on the MEAN targets only (engine_for_cyclical.py:93-118); the covariance targets
        # (:73-86) know `target_layer_norm_last` and `post_target_layer_norm` and nothing else
        import dataclasses
        hp_cov = dataclasses.replace(hp, target_batch_norm=False, target_instance_norm=False, post_target_instance_norm=False)
        targets, cov_targets = vo.build_targets(tm, mask, hp), vo.build_targets(tc, mask, hp_cov)
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in params.items()}
    out, cov_out = forward(leaves, cfg, samples, mask, False, None, drop)
    # loss = loss_cyc + std_loss0 * var_w0 + loss_stochastic (engine_for_cyclical.py:130-139, 161): the variance term acts on the MEAN outputs
    loss_cyc = vo.regression_loss(out, targets, vo.StepHParams(l1_beta=hp.l1_beta, l2_loss=hp.l2_loss, var_w0=hp.var_w0, var_margin0=hp.var_margin0))
    loss_w = wasserstein_loss(out.float(), cov_out.float(), targets, cov_targets, lam)
    loss = loss_cyc + loss_w
    if hp.loss_scale != -1:
        loss = loss * hp.loss_scale
    loss.backward()
    grads = {k: t.grad.detach() for k, t in leaves.items() if t.grad is not None}
    raw = {k: g.clone() for k, g in grads.items()}
    gnorm = vo.clip_grad_norm(grads, hp.clip_grad) if hp.clip_grad is not None else \
        torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    with torch.no_grad():
        vo.adamw_step(params, grads, m, v, step, hp)
        vo.ema_update(ema, params, hp.ema_decay)
    return vo.StepResult(float(loss.detach()), float(gnorm), out.detach(), targets, raw), float(loss_w.detach()), cov_out.detach(), cov_targets


def drop_path_scales(seed: int, step: int, cfg: vo.VitConfig, B: int):
    """path[k][layer] (B,) multipliers for the four draws of each block (k = 0 mean attn, 1 mean mlp, 2 cov attn,
    3 cov mlp); mirrors `droppath_kernel` with nbr = 4 in csrc/elementwise.hip."""
    import numpy as np
    rates = cfg.drop_path_rates()
    path = [[None] * cfg.depth for _ in range(4)]
    with np.errstate(over="ignore"):
        for i, r in enumerate(rates):
            if r <= 0:
                continue
            for k in range(4):
                key = vo._mix32(np.uint32(seed) ^ (np.uint32(step * 4 * cfg.depth + 4 * i + k + 1) * np.uint32(0x9E3779B9)))
                rnd = vo._mix32(np.arange(B, dtype=np.uint32) ^ key)
                thr = np.uint32(min(int(r * 4294967296.0), 0xFFFFFFFF))
                path[k][i] = torch.from_numpy((rnd >= thr).astype(np.float32) / np.float32(1.0 - r))
    return path
