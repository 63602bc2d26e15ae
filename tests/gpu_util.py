"""Shared helpers for the GPU parity tests and smoke(): build the native model with closed-form
weights, run native training steps, run the oracle on the same inputs."""
from functools import partial

import torch

from oracle import vit_oracle as vo
from oracle import vit_oracle_dist as vd
from oracle.closed_form import closed_form_state


def native_model(cfg: vo.VitConfig, gamma=None, device="cuda", two_stream=False):
    """VisionTransformerForCyclicalTraining (or the two-stream Dist... model) with closed-form weights."""
    from uncertainty_vit_amd.modeling_cyclical import (DistVisionTransformerForCyclicalTraining,
                                                       VisionTransformerForCyclicalTraining)
    cls = DistVisionTransformerForCyclicalTraining if two_stream else VisionTransformerForCyclicalTraining
    m = cls(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
            num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
            norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps), init_values=cfg.init_values,
            use_shared_rel_pos_bias=cfg.use_shared_rel_pos_bias, use_abs_pos_emb=cfg.use_abs_pos_emb and not two_stream,
            drop_path_rate=cfg.drop_path_rate, attn_drop_rate=cfg.attn_drop_rate)
    shapes = vd.param_shapes(cfg) if two_stream else vo.param_shapes(cfg)
    sd = closed_form_state(shapes, gamma=cfg.init_values if gamma is None else gamma)
    m.load_state_dict(sd, strict=False)
    return m.to(device), sd


class Args:
    opt, lr, weight_decay, opt_eps, opt_betas, momentum = "adamw", 2e-3, 0.05, 1e-8, (0.9, 0.999), 0.9


def native_trainer(model, lr=2e-3, wd=0.05, decay=0.9998):
    from uncertainty_vit_amd import optim_factory, utils
    a = Args()
    a.lr, a.weight_decay = lr, wd
    ema = utils.ModelEmaV2(model, decay=decay)
    opt = optim_factory.create_optimizer(a, model)
    return ema, opt


def native_steps(model, ema, opt, batches, target_layers, start=0, clip=3.0, l1_beta=2.0, decay=0.9998, l2_loss=False,
                 loss_scale=-1, post_target_layer_norm=True, stochastic=False, lam=1e-5, layer_results="end", var_w0=0.0,
                 var_margin0=0.5, **target_flags):
    """Each batch through the product's train_one_epoch (one-iteration loader); returns per-step stats."""
    from uncertainty_vit_amd import engine_for_cyclical as eng, utils
    out = []
    for s, (x, m) in enumerate(batches):
        loader = [((x, m), torch.zeros(1))]
        st = eng.train_one_epoch(model, ema, 0, decay, decay, target_layers, loader, opt, torch.device("cuda"), 0,
                                 utils.NativeScalerWithGradNormCount(), max_norm=clip, l1_beta=l1_beta, start_steps=start + s,
                                 layer_results=layer_results, var_w0=var_w0, var_margin0=var_margin0, loss_scale=loss_scale,
                                 **{"target_layer_norm_last": True, "post_target_layer_norm": post_target_layer_norm, **target_flags},
                                 l2_loss=l2_loss, stochastic=stochastic, lambda_pretraining=lam)
        out.append(st)
    return out


def oracle_state(sd):
    """(params, ema, adam m, adam v) dictionaries for the oracle, cloned from a closed-form state."""
    p = {k: v.clone() for k, v in sd.items()}
    e = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    return p, e, m, v


def grad_errors(native_grads, ref_grads, names=None):
    """Per tensor: (max-norm error / max|ref|, relative L2 error ||g - r|| / ||r||)."""
    out = {}
    for n in (names if names is not None else ref_grads.keys()):
        g, r = native_grads[n].detach().float().cpu().double(), ref_grads[n].double()
        rmax, rl2 = r.abs().max().item(), r.norm().item()
        out[n] = ((g - r).abs().max().item() / (rmax + 1e-30), (g - r).norm().item() / (rl2 + 1e-30))
    return out


def assert_grads_close(native_grads, ref_grads, names=None, max_tol=5e-2, l2_tol=2e-2, what=""):
    """Two bounds per tensor: the max-norm one (|g - r| <= max_tol * max|r|) catches a wrong large entry, the relative-L2
    one (||g - r|| <= l2_tol * ||r||) catches errors confined to many small-magnitude entries (a mis-indexed bias slice, a
    pad row leaking into a column sum) that the max-norm bound lets through."""
    errs = grad_errors(native_grads, ref_grads, names)
    worst = sorted(errs.items(), key=lambda kv: -kv[1][1])[:5]
    print(f"{what}worst relative-L2 gradient errors:", [(n, f"{e[1]:.2e}", f"max {e[0]:.2e}") for n, e in worst])
    bad = {n: e for n, e in errs.items() if e[0] > max_tol or e[1] > l2_tol}
    assert not bad, f"{what}gradient mismatch (max-norm ratio, relative L2): {bad}"


def full_size_step_properties(cfg_drop, cfg_nodrop, B, img, n_patches, n_mask, target_layers, two_stream=False, lam=1e-5,
                              lr=2e-3, wd=0.05, decay=0.9998, tag="full"):
    """Size-independent properties of ONE full-size step (the oracle cannot run these sizes in seconds):
    reported grad-norm = norm of the gradient arena, the AdamW bound |dw| <= lr (+ decay) with dw = -lr sign(g) where the
    clipped gradient is not tiny, the EMA identity, replay determinism of the counter-based dropout (same seed -> same
    loss, other seed -> other loss), frozen tensors untouched, and linearity of the gradient in the batch (dropout off:
    grad(B) = mean of the two half batches, every image having the same number of masked rows)."""
    import numpy as np
    import pytest
    from oracle.closed_form import closed_form_images, exact_masks
    x = closed_form_images(tag, B, img).cuda()
    mask = exact_masks(B, n_patches, n_mask, 77).cuda()

    def one_step(c, xs, ms, seed=99, lam=lam):
        model, _ = native_model(c, two_stream=two_stream)
        ema, opt = native_trainer(model, lr=lr, wd=wd, decay=decay)
        p0 = {n: t.detach().clone() for n, t in model.state_dict().items()}
        torch.manual_seed(seed)
        st = native_steps(model, ema, opt, [(xs, ms)], target_layers, start=3, clip=3.0, decay=decay, stochastic=two_stream,
                          lam=lam)[0]
        return model, ema, p0, st

    model, ema, p0, st = one_step(cfg_drop, x, mask)
    assert np.isfinite(st["loss"]) and 0.0 < st["loss"] < 10.0
    g = model._grad_arena
    gn = float(torch.sqrt((g.double() ** 2).sum()))
    assert st["grad_norm"] == pytest.approx(gn, rel=1e-4)
    coef = min(1.0, 3.0 / (gn + 1e-6))
    sd, esd = model.state_dict(), ema.module.state_dict()
    frozen = {n for n, _, _, _, dk in model._layout if dk == 2}
    decay_names = {n for n, p in model.named_parameters() if p.ndim > 1 and n not in ("cls_token", "pos_embed")}
    for n, p in model.named_parameters():
        if n in frozen:        # dead cov_qkv.weight: no gradient, no AdamW, no weight decay; EMA of an unchanged value
            assert torch.equal(sd[n], p0[n]) and float(p.grad.abs().sum()) == 0.0, n
            continue
        d = sd[n] - p0[n] * ((1 - lr * wd) if n in decay_names else 1.0)
        assert float(d.abs().max()) <= lr * (1 + 1e-3), n                    # first AdamW step: |m / sqrt(v)| <= 1
        big = (p.grad.abs() * coef) > 1e-5                                   # there the first step is exactly -lr * sign(g)
        if big.any():
            torch.testing.assert_close(d[big], -lr * torch.sign(p.grad[big]), rtol=0, atol=lr * 2e-3)
        torch.testing.assert_close(esd[n], decay * p0[n] + (1 - decay) * sd[n], rtol=0, atol=1e-7 + 2e-7 * float(p0[n].abs().max()))
    del model, ema
    # replay: same seed and iteration -> the same dropout masks -> the same loss (split-K atomics reorder fp32 sums)
    m2, _, _, st2 = one_step(cfg_drop, x, mask)
    del m2
    assert st2["loss"] == pytest.approx(st["loss"], rel=1e-5)
    m3, _, _, st3 = one_step(cfg_drop, x, mask, seed=100)
    del m3
    assert abs(st3["loss"] - st["loss"]) > 1e-7                              # another seed, another mask set
    # WassersteinLoss normalises by a batch-global max, so its gradient is not linear in the batch: lambda = 0 there
    ll = 0.0 if two_stream else lam
    mfull, _, _, _ = one_step(cfg_nodrop, x, mask, lam=ll)
    gfull = {n: p.grad.clone() for n, p in mfull.named_parameters()}
    del mfull
    h = B // 2
    ma, _, _, _ = one_step(cfg_nodrop, x[:h], mask[:h], lam=ll)
    ga = {n: p.grad.clone() for n, p in ma.named_parameters()}
    del ma
    mb, _, _, _ = one_step(cfg_nodrop, x[h:], mask[h:], lam=ll)
    for n, p in mb.named_parameters():
        if n in frozen:
            continue
        ref = 0.5 * (ga[n] + p.grad)
        err = float((gfull[n] - ref).abs().max())
        assert err <= 2e-2 * float(ref.abs().max()) + 1e-9, (n, err, float(ref.abs().max()))
        rl2 = float((gfull[n] - ref).norm() / (ref.norm() + 1e-30))
        assert rl2 <= 2e-2, (n, "relative L2", rl2)
    return st
