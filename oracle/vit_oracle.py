"""ORACLE — test infrastructure, NOT product code.

A CPU fp32 restatement (plain PyTorch on the host, functional style over a state-dict of
tensors) of the reference's data2vec-style student / EMA-teacher training step.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`uncertainty-vit_amd/`) never does and fails loudly when its HIP
library is missing.

Parity status: PINNED against outputs of the reference itself, run in the build container
through `tools/ref_harness.py` (reference files imported unmodified) by `tools/gen_golden.py`;
the vectors live in `tests/golden/*.npz` and `tests/test_oracle_golden.py` checks this file
against them.  The reference has no tests/golden vectors of its own.
The four timm symbols the reference calls (drop_path, trunc_normal_, ModelEmaV2, registry)
are not under /root/reference; their semantics are restated here and are "parity unpinned"
at that boundary -- goldens use explicit weights, injected dropout masks and the EMA lambda of
engine_for_cyclical.py:183, so nothing pinned depends on them.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class VitConfig:
    """Constructor arguments of VisionTransformerForCyclicalTraining (modeling_cyclical.py:34-60)."""
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    init_values: float = 1e-4            # --layer_scale_init_value
    drop_path_rate: float = 0.0
    attn_drop_rate: float = 0.0
    use_abs_pos_emb: bool = False
    use_shared_rel_pos_bias: bool = True
    ln_eps: float = 1e-6                 # partial(nn.LayerNorm, eps=1e-6), modeling_cyclical.py:294

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def num_tokens(self) -> int:
        return self.num_patches + 1

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def hidden(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)

    def drop_path_rates(self) -> List[float]:
        # modeling_cyclical.py:94-96 : torch.linspace(0, rate, depth)
        return [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]


@dataclass
class StepHParams:
    """The per-step knobs train_one_epoch consumes (engine_for_cyclical.py:24-32)."""
    target_layers: Sequence[int] = (6, 7, 8, 9, 10, 11)
    layer_results: str = "end"
    l1_beta: float = 2.0
    l2_loss: bool = False
    loss_scale: float = -1
    target_layer_norm_last: bool = True
    post_target_layer_norm: bool = True
    clip_grad: Optional[float] = 3.0
    lr: float = 2e-3
    weight_decay: float = 0.05
    betas: Sequence[float] = (0.9, 0.999)
    eps: float = 1e-8
    ema_decay: float = 0.9998
    var_w0: float = 0.0          # weight of the variance term (engine_for_cyclical.py:130-139, 161)
    var_margin0: float = 0.5
    target_batch_norm: bool = False           # engine_for_cyclical.py:94-104
    target_instance_norm: bool = False
    post_target_instance_norm: bool = False   # :112-115


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def relative_position_index(ws: int) -> np.ndarray:
    """int64 (ws*ws+1, ws*ws+1) index buffer; modeling_finetune.py:339-353.

    Token->token entries enumerate the (2ws-1)^2 relative offsets; three extra slots serve
    cls->token, token->cls and cls->cls.
    """
    n_rel = (2 * ws - 1) * (2 * ws - 1) + 3
    ys, xs = np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + (ws - 1)
    dx = xs[:, None] - xs[None, :] + (ws - 1)
    idx = np.zeros((ws * ws + 1, ws * ws + 1), dtype=np.int64)
    idx[1:, 1:] = dy * (2 * ws - 1) + dx
    idx[0, :] = n_rel - 3
    idx[:, 0] = n_rel - 2
    idx[0, 0] = n_rel - 1
    return idx


def param_shapes(cfg: VitConfig) -> Dict[str, tuple]:
    """State-dict key names and shapes of the base model (SURVEY.md section 8b registry row)."""
    C, Hd, P = cfg.embed_dim, cfg.hidden, cfg.patch_size
    s: Dict[str, tuple] = {
        "cls_token": (1, 1, C),
        "mask_token": (1, 1, C),
    }
    if cfg.use_abs_pos_emb:
        s["pos_embed"] = (1, cfg.num_tokens, C)
    s["patch_embed.proj.weight"] = (C, cfg.in_chans, P, P)
    s["patch_embed.proj.bias"] = (C,)
    if cfg.use_shared_rel_pos_bias:
        s["rel_pos_bias.relative_position_bias_table"] = ((2 * cfg.grid - 1) ** 2 + 3, cfg.num_heads)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s[b + "gamma_1"] = (C,)
        s[b + "gamma_2"] = (C,)
        s[b + "norm1.weight"] = (C,)
        s[b + "norm1.bias"] = (C,)
        s[b + "attn.q_bias"] = (C,)
        s[b + "attn.v_bias"] = (C,)
        s[b + "attn.qkv.weight"] = (3 * C, C)
        s[b + "attn.proj.weight"] = (C, C)
        s[b + "attn.proj.bias"] = (C,)
        s[b + "norm2.weight"] = (C,)
        s[b + "norm2.bias"] = (C,)
        s[b + "mlp.fc1.weight"] = (Hd, C)
        s[b + "mlp.fc1.bias"] = (Hd,)
        s[b + "mlp.fc2.weight"] = (C, Hd)
        s[b + "mlp.fc2.bias"] = (C,)
    s["norm.weight"] = (C,)
    s["norm.bias"] = (C,)
    s["lm_head.weight"] = (C, C)
    s["lm_head.bias"] = (C,)
    return s


def trunc_normal_(t: Tensor, std: float, gen: torch.Generator) -> Tensor:
    """Truncated normal on [-std, std] (modeling_cyclical.py:23-24 passes a=-std, b=std)."""
    lo = (1.0 + math.erf(-1.0 / math.sqrt(2.0))) / 2.0
    hi = (1.0 + math.erf(1.0 / math.sqrt(2.0))) / 2.0
    u = torch.empty_like(t).uniform_(2 * lo - 1, 2 * hi - 1, generator=gen)
    return t.copy_(u.erfinv_().mul_(std * math.sqrt(2.0)).clamp_(-std, std))


def init_params(cfg: VitConfig, seed: int = 0, std: float = 0.02,
                rel_table_std: float = 0.0) -> Dict[str, Tensor]:
    """Init rule of modeling_cyclical.py:135-161: trunc-normal(std) for Linear/Conv weights and
    the cls/mask tokens, zeros for biases, LayerNorm (1, 0), gamma = init_values, then
    proj.weight / fc2.weight divided by sqrt(2*(layer+1)).  The rel-pos table stays zero in the
    reference (modeling_finetune.py:357 is commented out); `rel_table_std` > 0 lets tests use a
    non-zero table so bias bugs cannot hide."""
    gen = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        t = torch.zeros(shape, dtype=torch.float32)
        if name.endswith("gamma_1") or name.endswith("gamma_2"):
            t.fill_(cfg.init_values)
        elif ("norm" in name) and name.endswith("weight"):
            t.fill_(1.0)
        elif name.endswith("relative_position_bias_table"):
            if rel_table_std > 0:
                t.normal_(0.0, rel_table_std, generator=gen)
        elif name.endswith(".weight") or name in ("cls_token", "mask_token", "pos_embed"):
            trunc_normal_(t, std, gen)
        p[name] = t
    for i in range(cfg.depth):
        p[f"blocks.{i}.attn.proj.weight"].div_(math.sqrt(2.0 * (i + 1)))
        p[f"blocks.{i}.mlp.fc2.weight"].div_(math.sqrt(2.0 * (i + 1)))
    return p


def no_decay_names(params: Dict[str, Tensor]) -> set:
    """optim_factory.py:58-97 with skip list {'pos_embed','cls_token'} (modeling_cyclical.py:163-165):
    1-D tensors, '*.bias' and the skip list get weight_decay 0."""
    out = set()
    for name, t in params.items():
        if t.ndim == 1 or name.endswith(".bias") or name in ("pos_embed", "cls_token"):
            out.add(name)
    return out


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
def patch_embed(p: Dict[str, Tensor], cfg: VitConfig, x: Tensor) -> Tensor:
    """Conv2d(k=stride=patch) + flatten(2).transpose(1,2): modeling_finetune.py:319-325."""
    B = x.shape[0]
    P, g = cfg.patch_size, cfg.grid
    # non-overlapping conv == GEMM over re-indexed pixels
    cols = x.reshape(B, cfg.in_chans, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, -1)
    w = p["patch_embed.proj.weight"].reshape(cfg.embed_dim, -1)
    return cols @ w.t() + p["patch_embed.proj.bias"]


def rel_pos_bias(p: Dict[str, Tensor], cfg: VitConfig) -> Optional[Tensor]:
    """Gather table -> (H, N, N): modeling_finetune.py:359-364."""
    if not cfg.use_shared_rel_pos_bias:
        return None
    idx = torch.from_numpy(relative_position_index(cfg.grid)).reshape(-1)
    N = cfg.num_tokens
    return p["rel_pos_bias.relative_position_bias_table"][idx].reshape(N, N, -1).permute(2, 0, 1)


def attention(p: Dict[str, Tensor], pre: str, cfg: VitConfig, x: Tensor, bias: Optional[Tensor],
              keep: Optional[Tensor] = None) -> Tensor:
    """modeling_finetune.py:145-188.  `keep` is an optional (B,H,N,N) dropout multiplier
    (0 or 1/(1-p)) replacing nn.Dropout's RNG so GPU and CPU share one mask."""
    B, N, C = x.shape
    H, d = cfg.num_heads, cfg.head_dim
    qkv_bias = torch.cat((p[pre + "q_bias"], torch.zeros_like(p[pre + "v_bias"]), p[pre + "v_bias"]))
    qkv = x @ p[pre + "qkv.weight"].t() + qkv_bias
    qkv = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (d ** -0.5), qkv[1], qkv[2]
    s = q @ k.transpose(-2, -1)
    if bias is not None:
        s = s + bias
    a = s.softmax(dim=-1)
    if keep is not None:
        a = a * keep
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return o @ p[pre + "proj.weight"].t() + p[pre + "proj.bias"]


def mlp(p: Dict[str, Tensor], pre: str, x: Tensor) -> Tensor:
    """fc1 -> exact GELU -> fc2: modeling_finetune.py:75-82."""
    h = F.gelu(x @ p[pre + "fc1.weight"].t() + p[pre + "fc1.bias"])
    return h @ p[pre + "fc2.weight"].t() + p[pre + "fc2.bias"]


def layer_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps)
    if w is not None:
        y = y * w + b
    return y


@dataclass
class DropState:
    """Injected randomness for one student forward: per-layer per-sample drop-path multipliers
    (0 or 1/keep; timm drop_path semantics, modeling_finetune.py:51-62) and per-layer attention
    dropout multipliers.  None entries mean "no dropout"."""
    path1: List[Optional[Tensor]] = field(default_factory=list)   # (B,) per layer, attn branch
    path2: List[Optional[Tensor]] = field(default_factory=list)   # (B,) per layer, mlp branch
    attn: List[Optional[Tensor]] = field(default_factory=list)    # (B,H,N,N) per layer


def block(p: Dict[str, Tensor], i: int, cfg: VitConfig, x: Tensor, bias: Optional[Tensor],
          drop: Optional[DropState]):
    """modeling_finetune.py:290-299 (gamma branch)."""
    b = f"blocks.{i}."
    keep = drop.attn[i] if drop and drop.attn else None
    a = attention(p, b + "attn.", cfg, layer_norm(x, p[b + "norm1.weight"], p[b + "norm1.bias"], cfg.ln_eps),
                  bias, keep)
    a = p[b + "gamma_1"] * a
    if drop and drop.path1 and drop.path1[i] is not None:
        a = a * drop.path1[i].reshape(-1, 1, 1)
    x = x + a
    f = p[b + "gamma_2"] * mlp(p, b + "mlp.", layer_norm(x, p[b + "norm2.weight"], p[b + "norm2.bias"], cfg.ln_eps))
    if drop and drop.path2 and drop.path2[i] is not None:
        f = f * drop.path2[i].reshape(-1, 1, 1)
    return x + f, f


def forward_features(p: Dict[str, Tensor], cfg: VitConfig, x: Tensor, mask: Optional[Tensor],
                     layer_results: Optional[str], drop: Optional[DropState] = None):
    """modeling_cyclical.py:170-207."""
    x = patch_embed(p, cfg, x)
    B = x.shape[0]
    if mask is not None:
        w = mask.reshape(B, -1, 1).to(x.dtype)
        x = x * (1 - w) + p["mask_token"] * w
    x = torch.cat((p["cls_token"].expand(B, -1, -1), x), dim=1)
    if cfg.use_abs_pos_emb:
        x = x + p["pos_embed"]
    bias = rel_pos_bias(p, cfg)
    z = []
    for i in range(cfg.depth):
        x, fc = block(p, i, cfg, x, bias, drop)
        if layer_results == "end":
            z.append(x)
        elif layer_results == "fc":
            z.append(fc)
    if layer_results:
        return z
    return layer_norm(x, p["norm.weight"], p["norm.bias"], cfg.ln_eps)


def forward(p: Dict[str, Tensor], cfg: VitConfig, x: Tensor, mask: Optional[Tensor],
            return_all_tokens: bool = False, layer_results: Optional[str] = None,
            drop: Optional[DropState] = None):
    """The three return modes of modeling_cyclical.py:209-225."""
    out = forward_features(p, cfg, x, mask, layer_results, drop)
    if layer_results:
        return [z[:, 1:] for z in out]
    out = out[:, 1:]
    if not return_all_tokens:
        out = out.reshape(-1, out.shape[-1])[mask.flatten().bool()]
    return out @ p["lm_head.weight"].t() + p["lm_head.bias"]


def build_targets(layer_outs: List[Tensor], mask: Tensor, hp: StepHParams) -> Tensor:
    """engine_for_cyclical.py:90-122: per layer [batch norm over (B, T) per channel] [instance norm over T per
    (sample, channel)] (both affine-free, biased variance, eps 1e-5) [affine-free LayerNorm (eps 1e-5)]; mean over the
    layers; [post instance norm] [post LayerNorm]; gather the masked rows."""
    C = layer_outs[0].shape[-1]
    vals = [layer_outs[i].float() for i in hp.target_layers]
    if hp.target_batch_norm or hp.target_instance_norm:
        vals = [v.permute(0, 2, 1) for v in vals]                      # btc -> bct
        if hp.target_batch_norm:
            vals = [F.batch_norm(v, running_mean=None, running_var=None, training=True) for v in vals]
        if hp.target_instance_norm:
            vals = [F.instance_norm(v) for v in vals]
        vals = [v.permute(0, 2, 1) for v in vals]
    if hp.target_layer_norm_last:
        vals = [F.layer_norm(v, (C,)) for v in vals]
    t = sum(vals) / len(hp.target_layers)
    if hp.post_target_instance_norm:
        t = F.instance_norm(t.permute(0, 2, 1).float()).permute(0, 2, 1)
    if hp.post_target_layer_norm:
        t = F.layer_norm(t.float(), (C,))
    return t.reshape(-1, C)[mask.flatten().bool()]


def variance_term(outputs: Tensor, hp: StepHParams) -> Tensor:
    """engine_for_cyclical.py:130-139: z0 = sqrt(var over the masked rows (unbiased) + 1e-6) per channel;
    std_loss0 = sum(relu(var_margin0 - z0)) / channels when var_w0 > 0, else 0."""
    z0 = torch.sqrt(outputs.float().reshape(-1, outputs.shape[-1]).var(dim=0) + 1e-6)
    if hp.var_w0 > 0:
        return torch.sum(F.relu(hp.var_margin0 - z0)) / z0.shape[0]
    return torch.zeros((), dtype=torch.float32)


def regression_loss(outputs: Tensor, targets: Tensor, hp: StepHParams) -> Tensor:
    """engine_for_cyclical.py:130-163: smooth-L1 / MSE + var_w0 * std_loss0, then loss_scale."""
    outputs = outputs.float()
    assert outputs.shape == targets.shape
    if hp.l2_loss:
        loss = F.mse_loss(outputs, targets)
    else:
        loss = F.smooth_l1_loss(outputs, targets, beta=hp.l1_beta)
    if hp.var_w0 > 0:
        loss = loss + variance_term(outputs, hp) * hp.var_w0
    if hp.loss_scale != -1:
        loss = loss * hp.loss_scale
    return loss


# --------------------------------------------------------------------------------------
# optimiser / EMA
# --------------------------------------------------------------------------------------
def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_ semantics (utils.py:375-376): global L2 norm, scale by
    min(1, max_norm / (norm + 1e-6)); returns the unclipped norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def adamw_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], m: Dict[str, Tensor],
               v: Dict[str, Tensor], step: int, hp: StepHParams, lr: Optional[float] = None,
               wd: Optional[float] = None) -> None:
    """torch.optim.AdamW (optim_factory.py:133-134) with the decay / no-decay split of
    optim_factory.py:58-97; `step` is 1-based."""
    lr = hp.lr if lr is None else lr
    wd = hp.weight_decay if wd is None else wd
    b1, b2 = hp.betas
    nd = no_decay_names(params)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for name, p in params.items():
        if name not in grads:
            continue
        g = grads[name]
        if name not in nd:
            p.mul_(1.0 - lr * wd)
        m[name].mul_(b1).add_(g, alpha=1.0 - b1)
        v[name].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v[name].sqrt() / math.sqrt(bc2)).add_(hp.eps)
        p.addcdiv_(m[name], denom, value=-lr / bc1)


def ema_update(ema: Dict[str, Tensor], params: Dict[str, Tensor], decay: float) -> None:
    """e <- d*e + (1-d)*m for every float state-dict value (engine_for_cyclical.py:183).
    The int64 index buffer is copied through."""
    for name in ema:
        ema[name].copy_(decay * ema[name] + (1.0 - decay) * params[name])


@dataclass
class StepResult:
    loss: float
    grad_norm: float
    outputs: Tensor
    targets: Tensor
    grads: Dict[str, Tensor]
    loss_var0: float = 0.0


def train_step(params: Dict[str, Tensor], ema: Dict[str, Tensor], m: Dict[str, Tensor],
               v: Dict[str, Tensor], cfg: VitConfig, hp: StepHParams, samples: Tensor,
               mask: Tensor, step: int, drop: Optional[DropState] = None,
               lr: Optional[float] = None, wd: Optional[float] = None,
               decay: Optional[float] = None) -> StepResult:
    """One iteration of engine_for_cyclical.py:45-186 (non-stochastic branch): teacher forward
    under no_grad -> targets; student forward; loss; backward; clip; AdamW; EMA."""
    with torch.no_grad():
        t_layers = forward(ema, cfg, samples, None, True, hp.layer_results)
        targets = build_targets(t_layers, mask, hp)
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in params.items()}
    outputs = forward(leaves, cfg, samples, mask, False, None, drop)
    loss = regression_loss(outputs, targets, hp)
    loss.backward()
    grads = {k: t.grad.detach() for k, t in leaves.items() if t.grad is not None}
    raw = {k: g.clone() for k, g in grads.items()}
    if hp.clip_grad is not None:
        gnorm = clip_grad_norm(grads, hp.clip_grad)
    else:
        gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    with torch.no_grad():
        adamw_step(params, grads, m, v, step, hp, lr, wd)
        ema_update(ema, params, hp.ema_decay if decay is None else decay)
    res = StepResult(float(loss.detach()), float(gnorm), outputs.detach(), targets, raw)
    res.loss_var0 = float(variance_term(outputs.detach(), hp))     # the `loss_var0` meter (engine_for_cyclical.py:198)
    return res


# --------------------------------------------------------------------------------------
# counter-based dropout mask shared with the HIP kernels
# --------------------------------------------------------------------------------------
def _mix32(x: np.ndarray) -> np.ndarray:
    """32-bit finaliser (same constants as uvit_hash32 in csrc/common.h)."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def attn_keep_mask(seed: int, layer: int, B: int, H: int, N: int, p_drop: float, NP: int = 208) -> Tensor:
    """(B,H,N,N) multiplier, 0 or 1/(1-p).  Mirrors `keep4` / `keep1` in csrc/attention.hip: one 32-bit hash per
    (query row, key pair), pair index = ((b*H+h)*N + q) * (NP/2) + (key >> 1); key takes the low (even key) or
    high (odd key) 16 bits and is kept when that half >= int(p * 65536)."""
    if p_drop <= 0:
        return torch.ones(B, H, N, N)
    with np.errstate(over="ignore"):
        key32 = _mix32(np.uint32(seed) ^ (np.uint32(layer + 1) * np.uint32(0x9E3779B9)))
        rows = (np.arange(B * H * N, dtype=np.uint64) * np.uint64(NP // 2)).astype(np.uint32)
        pairs = (rows[:, None] + np.arange((N + 1) // 2, dtype=np.uint32)[None, :]).astype(np.uint32)
        x = ((pairs ^ key32) * np.uint32(0x9E3779B1)).astype(np.uint32)
        x ^= x >> np.uint32(15)
        x = (x * np.uint32(0x85EBCA77)).astype(np.uint32)
        x ^= x >> np.uint32(13)
    halves = np.stack([x & np.uint32(0xFFFF), x >> np.uint32(16)], axis=-1).reshape(B * H * N, -1)[:, :N]
    thr = min(int(p_drop * 65536.0), 65535)
    keep = (halves >= thr).astype(np.float32) / np.float32(1.0 - p_drop)
    return torch.from_numpy(keep.reshape(B, H, N, N))


def drop_path_scales(seed: int, step: int, cfg: VitConfig, B: int):
    """Per-layer (B,) multipliers for both branches; mirrors `uvit_droppath_scales`."""
    rates = cfg.drop_path_rates()
    p1, p2 = [], []
    with np.errstate(over="ignore"):
        for i, r in enumerate(rates):
            for br, dst in ((0, p1), (1, p2)):
                if r <= 0:
                    dst.append(None)
                    continue
                key = _mix32(np.uint32(seed) ^ (np.uint32(step * 2 * cfg.depth + 2 * i + br + 1) * np.uint32(0x9E3779B9)))
                rnd = _mix32(np.arange(B, dtype=np.uint32) ^ key)
                thr = np.uint32(min(int(r * 4294967296.0), 0xFFFFFFFF))
                dst.append(torch.from_numpy((rnd >= thr).astype(np.float32) / np.float32(1.0 - r)))
    return p1, p2
