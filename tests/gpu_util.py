"""Shared helpers for the GPU parity tests and smoke(): build the native model with closed-form
weights, run native training steps, run the oracle on the same inputs."""
from functools import partial

import torch

from oracle import vit_oracle as vo
from oracle.closed_form import closed_form_state


def native_model(cfg: vo.VitConfig, gamma=None, device="cuda"):
    from uncertainty_vit_amd.modeling_cyclical import VisionTransformerForCyclicalTraining
    m = VisionTransformerForCyclicalTraining(
        img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
        num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
        norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps), init_values=cfg.init_values,
        use_shared_rel_pos_bias=cfg.use_shared_rel_pos_bias, use_abs_pos_emb=False,
        drop_path_rate=cfg.drop_path_rate, attn_drop_rate=cfg.attn_drop_rate)
    sd = closed_form_state(vo.param_shapes(cfg), gamma=cfg.init_values if gamma is None else gamma)
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
                 loss_scale=-1, post_target_layer_norm=True):
    """Each batch through the product's train_one_epoch (one-iteration loader); returns per-step stats."""
    from uncertainty_vit_amd import engine_for_cyclical as eng, utils
    out = []
    for s, (x, m) in enumerate(batches):
        loader = [((x, m), torch.zeros(1))]
        st = eng.train_one_epoch(model, ema, 0, decay, decay, target_layers, loader, opt, torch.device("cuda"), 0,
                                 utils.NativeScalerWithGradNormCount(), max_norm=clip, l1_beta=l1_beta, start_steps=start + s,
                                 layer_results="end", loss_scale=loss_scale, target_layer_norm_last=True,
                                 post_target_layer_norm=post_target_layer_norm, l2_loss=l2_loss)
        out.append(st)
    return out
