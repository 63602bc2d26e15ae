"""Generate tests/golden/*.npz by running the UNMODIFIED reference (imported from
/root/reference through tools/ref_harness.py).  Build-container only; never runs on the GPU box.

    python tools/gen_golden.py            # rewrites every fixture

Fixtures are data only.  Weights and images come from oracle/closed_form.py (evaluated
identically by the tests), so the files hold just masks and expected outputs: small tensors
in full, large ones as (sum, abs-sum) + 64 evenly spaced samples.
"""
import argparse
import os
import sys
from functools import partial
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ref_harness  # noqa: E402
from oracle.closed_form import (checksum, closed_form_images, closed_form_state,  # noqa: E402
                                exact_masks)

OUT = os.path.join(ROOT, "tests", "golden")
FULL_LIMIT = 4096   # tensors up to this many elements are stored in full


def put(out, key, t):
    t = t.detach()
    if t.numel() <= FULL_LIMIT:
        out[key + "/full"] = t.float().numpy()
    else:
        s, v = checksum(t)
        out[key + "/sum"], out[key + "/samples"] = s, v


def build(mc, img, dim, depth, heads, init_values, abs_pos=False):
    model = mc.VisionTransformerForCyclicalTraining(
        img_size=img, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4,
        qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), init_values=init_values,
        use_shared_rel_pos_bias=True, use_abs_pos_emb=abs_pos, drop_path_rate=0.0, attn_drop_rate=0.0)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if v.dtype == torch.float32}
    model.load_state_dict(closed_form_state(shapes, gamma=init_values), strict=False)
    return model


def run_reference_steps(eng, model, batches, target_layers, lr=2e-3, wd=0.05, clip=3.0,
                        l1_beta=2.0, decay=0.9998, layer_results="end", var_w0=0.0, var_margin0=0.5, tflags=None):
    import optim_factory
    import timm.utils as U
    ema = U.ModelEmaV2(model, decay=decay)
    args = SimpleNamespace(opt="adamw", lr=lr, weight_decay=wd, opt_eps=1e-8, opt_betas=(0.9, 0.999),
                           momentum=0.9)
    opt = optim_factory.create_optimizer(args, model)
    names = {id(p): n for n, p in model.named_parameters()}
    groups = {("decay" if g["weight_decay"] > 0 else "no_decay"): [names[id(p)] for p in g["params"]]
              for g in opt.param_groups}
    scaler = ref_harness.HarnessScaler()
    rec = {"loss": [], "grad_norm": []}
    first_grads = None
    for s, (bx, bm) in enumerate(batches):
        loader = [((bx, bm), torch.zeros(1))]
        stats = eng.train_one_epoch(
            model, ema, 0, decay, decay, target_layers, loader, opt, torch.device("cpu"), 0, scaler,
            max_norm=clip, l1_beta=l1_beta, log_writer=None, lr_scheduler=None, start_steps=s,
            lr_schedule_values=None, wd_schedule_values=None, l2_loss=False, layer_results=layer_results,
            var_w0=var_w0, var_w1=0.0, var_margin0=var_margin0, start_lr_decay_at_step=-1, loss_scale=-1, mask_dropout_prob=-1.0,
            stochastic=False, **(tflags or dict(target_layer_norm_last=True, target_batch_norm=False, target_instance_norm=False,
                                                post_target_instance_norm=False, post_target_layer_norm=True)))
        rec["loss"].append(stats["loss"])
        rec["grad_norm"].append(float(stats["grad_norm"]))
        rec.setdefault("loss_var0", []).append(float(stats["loss_var0"]))
        if s == 0:
            pn = [n for n, _ in model.named_parameters()]
            first_grads = {n: g for n, g in zip(pn, scaler.grads) if g is not None}
    return rec, first_grads, ema, groups


def gen_model_case(mc, eng, tag, img, dim, depth, heads, init_values, B, n_mask, seed, steps=3):
    model = build(mc, img, dim, depth, heads, init_values)
    n_patches = (img // 16) ** 2
    batches = [(closed_form_images(f"{tag}/{s}", B, img), exact_masks(B, n_patches, n_mask, seed + s))
               for s in range(steps)]
    out = {"cfg": np.array([img, dim, depth, heads, B, n_mask, steps], dtype=np.int64),
           "init_values": np.float64(init_values), "mask_seed": np.int64(seed)}
    x, mask = batches[0]
    model.eval()
    with torch.no_grad():
        ends = model(x, None, True, layer_results="end")
        fcs = model(x, None, True, layer_results="fc")
        stu = model(x, mask, return_all_tokens=False)
        stu_all = model(x, mask, return_all_tokens=True)
    for i, (e, f) in enumerate(zip(ends, fcs)):
        put(out, f"fwd/end{i}", e)
        put(out, f"fwd/fc{i}", f)
    put(out, "fwd/student_masked", stu)
    put(out, "fwd/student_all", stu_all)
    for s, (_, bm) in enumerate(batches):
        out[f"mask{s}"] = bm.numpy()
    model.train()
    tl = list(range(depth // 2, depth))
    rec, grads, ema, groups = run_reference_steps(eng, model, batches, tl)
    out["target_layers"] = np.array(tl, dtype=np.int64)
    out["step/loss"] = np.array(rec["loss"], dtype=np.float64)
    out["step/grad_norm"] = np.array(rec["grad_norm"], dtype=np.float64)
    for k, v in grads.items():
        put(out, "grad0/" + k, v)
    for k, v in model.state_dict().items():
        if v.dtype == torch.float32:
            put(out, "post/" + k, v)
    for k, v in ema.module.state_dict().items():
        if v.dtype == torch.float32:
            put(out, "ema/" + k, v)
    out["groups/decay"] = np.array(groups["decay"])
    out["groups/no_decay"] = np.array(groups["no_decay"])
    np.savez_compressed(os.path.join(OUT, f"model_{tag}.npz"), **out)
    print("wrote", tag, "loss", rec["loss"], "gnorm", rec["grad_norm"])


def gen_flag_case(mc, eng):
    """The flag-gated arithmetic of the step that no BASELINE config switches on (VERDICT r1 "Missing 7"), from the reference:
    `--layer_results fc` targets (teacher MLP-branch outputs, modeling_cyclical.py:199-205) and the variance term
    `--var_w0 1 --var_margin0 1` (engine_for_cyclical.py:130-139); tiny model, 2 steps."""
    img, dim, depth, heads, B, n_mask = 48, 128, 3, 2, 5, 4
    model = build(mc, img, dim, depth, heads, 0.1)
    batches = [(closed_form_images(f"flags/{s}", B, img), exact_masks(B, 9, n_mask, 700 + s)) for s in range(2)]
    model.train()
    rec, grads, ema, _ = run_reference_steps(eng, model, batches, [1, 2], layer_results="fc", var_w0=1.0, var_margin0=1.0)
    out = {"cfg": np.array([img, dim, depth, heads, B, n_mask, 2], dtype=np.int64), "loss": np.array(rec["loss"]),
           "grad_norm": np.array(rec["grad_norm"]), "loss_var0": np.array(rec["loss_var0"])}
    for s, (_, bm) in enumerate(batches):
        out[f"mask{s}"] = bm.numpy()
    for k, v in grads.items():
        put(out, "grad0/" + k, v)
    for k, v in model.state_dict().items():
        if v.dtype == torch.float32:
            put(out, "post/" + k, v)
    np.savez_compressed(os.path.join(OUT, "model_flags.npz"), **out)
    print("wrote flag case: loss", rec["loss"], "loss_var0", rec["loss_var0"], "gnorm", rec["grad_norm"])


TARGET_NORM_CASES = {   # name -> the five target-builder flags of engine_for_cyclical.py:94-118
    "bn":        dict(target_batch_norm=True, target_instance_norm=False, target_layer_norm_last=True, post_target_instance_norm=False, post_target_layer_norm=True),
    "in":        dict(target_batch_norm=False, target_instance_norm=True, target_layer_norm_last=True, post_target_instance_norm=False, post_target_layer_norm=False),
    "bn_in_pin": dict(target_batch_norm=True, target_instance_norm=True, target_layer_norm_last=False, post_target_instance_norm=True, post_target_layer_norm=True),
    "raw_pin":   dict(target_batch_norm=False, target_instance_norm=False, target_layer_norm_last=False, post_target_instance_norm=True, post_target_layer_norm=False),
}


def gen_target_norm_cases(mc, eng):
    """Batch- / instance-norm target variants (engine_for_cyclical.py:94-118; flags off in every BASELINE config): one
    reference step per flag combination on a tiny model (10 tokens), loss + grad-norm + a few gradients."""
    img, dim, depth, heads, B, n_mask = 48, 128, 3, 2, 6, 4
    out = {"cfg": np.array([img, dim, depth, heads, B, n_mask, 1], dtype=np.int64), "cases": np.array(list(TARGET_NORM_CASES))}
    x, mask = closed_form_images("tnorm", B, img), exact_masks(B, 9, n_mask, 900)
    out["mask"] = mask.numpy()
    for name, fl in TARGET_NORM_CASES.items():
        model = build(mc, img, dim, depth, heads, 0.1)
        model.train()
        rec, grads, _, _ = run_reference_steps(eng, model, [(x, mask)], [1, 2], tflags=fl)
        out[f"{name}/loss"], out[f"{name}/grad_norm"] = np.float64(rec["loss"][0]), np.float64(rec["grad_norm"][0])
        for k in ("lm_head.weight", "blocks.2.mlp.fc2.weight", "blocks.0.attn.qkv.weight", "norm.weight"):
            put(out, f"{name}/grad/{k}", grads[k])
        print("target-norm case", name, "loss", rec["loss"][0], "gnorm", rec["grad_norm"][0])
    np.savez_compressed(os.path.join(OUT, "target_norms.npz"), **out)


def gen_abs_pos_case(mc, eng):
    """`--abs_pos_emb` (modeling_cyclical.py:80-84,193-194; off in every BASELINE config): forward + 2 steps, tiny model."""
    img, dim, depth, heads, B, n_mask = 48, 128, 2, 2, 4, 4
    model = build(mc, img, dim, depth, heads, 0.1, abs_pos=True)
    names = list(model.state_dict().keys())
    batches = [(closed_form_images(f"abspos/{s}", B, img), exact_masks(B, 9, n_mask, 800 + s)) for s in range(2)]
    out = {"cfg": np.array([img, dim, depth, heads, B, n_mask, 2], dtype=np.int64), "names": np.array(names)}
    model.eval()
    with torch.no_grad():
        ends = model(batches[0][0], None, True, layer_results="end")
        stu = model(batches[0][0], batches[0][1], return_all_tokens=False)
    for i, e in enumerate(ends):
        put(out, f"fwd/end{i}", e)
    put(out, "fwd/student_masked", stu)
    model.train()
    rec, grads, ema, groups = run_reference_steps(eng, model, batches, [1])
    out["loss"], out["grad_norm"] = np.array(rec["loss"]), np.array(rec["grad_norm"])
    out["groups/no_decay"] = np.array(groups["no_decay"])
    for s, (_, bm) in enumerate(batches):
        out[f"mask{s}"] = bm.numpy()
    for k, v in grads.items():
        put(out, "grad0/" + k, v)
    for k, v in model.state_dict().items():
        if v.dtype == torch.float32:
            put(out, "post/" + k, v)
    np.savez_compressed(os.path.join(OUT, "model_abspos.npz"), **out)
    print("wrote abs-pos case: loss", rec["loss"], "gnorm", rec["grad_norm"])


CURVE_LR = 1e-4   # constant; the README recipe warms up from 1e-6, 2e-3 from step 0 is chaotic on a tiny model


def gen_dist_case(mc, eng, tag, img, dim, depth, heads, init_values, B, n_mask, seed, steps=2):
    """Two-stream model (DistVisionTransformerForCyclicalTraining) + train_one_epoch(stochastic=True)."""
    import modeling_cyclical_dist as mcd
    from oracle.closed_form import closed_form
    model = mcd.DistVisionTransformerForCyclicalTraining(
        img_size=img, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4, qkv_bias=True,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), init_values=init_values, use_shared_rel_pos_bias=True,
        use_abs_pos_emb=False, drop_path_rate=0.0, attn_drop_rate=0.0)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if v.dtype == torch.float32}
    model.load_state_dict(closed_form_state(shapes, gamma=init_values), strict=False)
    n_patches = (img // 16) ** 2
    batches = [(closed_form_images(f"{tag}/{s}", B, img), exact_masks(B, n_patches, n_mask, seed + s)) for s in range(steps)]
    out = {"cfg": np.array([img, dim, depth, heads, B, n_mask, steps], dtype=np.int64),
           "init_values": np.float64(init_values), "names": np.array(list(shapes.keys()))}
    x, mask = batches[0]
    model.eval()
    with torch.no_grad():
        em, ec = model(x, None, True, layer_results="end")
        sm, sc = model(x, mask, return_all_tokens=False)
    for i in range(depth):
        put(out, f"fwd/mean_end{i}", em[i])
        put(out, f"fwd/cov_end{i}", ec[i])
    put(out, "fwd/student_mean", sm)
    put(out, "fwd/student_cov", sc)
    for s, (_, bm) in enumerate(batches):
        out[f"mask{s}"] = bm.numpy()
    model.train()
    import optim_factory
    import timm.utils as U
    ema = U.ModelEmaV2(model, decay=0.9998)
    args = SimpleNamespace(opt="adamw", lr=2e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=(0.9, 0.999), momentum=0.9)
    opt = optim_factory.create_optimizer(args, model)
    scaler = ref_harness.HarnessScaler()
    tl = list(range(depth // 2, depth))
    losses, gnorms = [], []
    for s, (bx, bm) in enumerate(batches):
        st = eng.train_one_epoch(model, ema, 0, 0.9998, 0.9998, tl, [((bx, bm), torch.zeros(1))], opt, torch.device("cpu"), 0, scaler,
                                 max_norm=3.0, l1_beta=2.0, start_steps=s, layer_results="end", loss_scale=-1,
                                 target_layer_norm_last=True, post_target_layer_norm=True, stochastic=True, lambda_pretraining=1e-2)
        losses.append(st["loss"]); gnorms.append(float(st["grad_norm"]))
        if s == 0:
            pn = [n for n, _ in model.named_parameters()]
            out["grad0_none"] = np.array([n for n, g in zip(pn, scaler.grads) if g is None])
            for n, g in zip(pn, scaler.grads):
                if g is not None:
                    put(out, "grad0/" + n, g)
    out["target_layers"] = np.array(tl, dtype=np.int64)
    out["step/loss"], out["step/grad_norm"] = np.array(losses), np.array(gnorms)
    for k, v in model.state_dict().items():
        if v.dtype == torch.float32:
            put(out, "post/" + k, v)
    for k, v in ema.module.state_dict().items():
        if v.dtype == torch.float32:
            put(out, "ema/" + k, v)
    np.savez_compressed(os.path.join(OUT, f"dist_{tag}.npz"), **out)
    print("wrote dist", tag, "loss", losses, "gnorm", gnorms)


DIST_TARGET_NORM_CASES = {   # two-stream step with the mean-target variants of engine_for_cyclical.py:93-118 (covariance targets :73-86)
    "bn":        dict(target_batch_norm=True, target_instance_norm=False, target_layer_norm_last=True, post_target_instance_norm=False, post_target_layer_norm=True),
    "bn_in_pin": dict(target_batch_norm=True, target_instance_norm=True, target_layer_norm_last=False, post_target_instance_norm=True, post_target_layer_norm=True),
}


def gen_dist_target_norm_cases(mc, eng):
    """`train_one_epoch(stochastic=True)` with the batch- / instance-norm target variants: the reference applies them to the MEAN
    targets (engine_for_cyclical.py:93-118) while the covariance targets (:73-86) only follow the two layer-norm flags.  One reference
    step per flag combination on the tiny two-stream model: loss, grad-norm and a few gradients."""
    import modeling_cyclical_dist as mcd
    import optim_factory
    import timm.utils as U
    img, dim, depth, heads, B, n_mask = 48, 128, 3, 2, 5, 4
    out = {"cfg": np.array([img, dim, depth, heads, B, n_mask, 1], dtype=np.int64), "cases": np.array(list(DIST_TARGET_NORM_CASES)),
           "init_values": np.float64(0.1)}
    x, mask = closed_form_images("dtnorm", B, img), exact_masks(B, 9, n_mask, 950)
    out["mask"] = mask.numpy()
    for name, fl in DIST_TARGET_NORM_CASES.items():
        model = mcd.DistVisionTransformerForCyclicalTraining(
            img_size=img, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4, qkv_bias=True,
            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), init_values=0.1, use_shared_rel_pos_bias=True,
            use_abs_pos_emb=False, drop_path_rate=0.0, attn_drop_rate=0.0)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if v.dtype == torch.float32}
        model.load_state_dict(closed_form_state(shapes, gamma=0.1), strict=False)
        model.train()
        ema = U.ModelEmaV2(model, decay=0.9998)
        args = SimpleNamespace(opt="adamw", lr=2e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=(0.9, 0.999), momentum=0.9)
        opt = optim_factory.create_optimizer(args, model)
        scaler = ref_harness.HarnessScaler()
        st = eng.train_one_epoch(model, ema, 0, 0.9998, 0.9998, [1, 2], [((x, mask), torch.zeros(1))], opt, torch.device("cpu"), 0, scaler,
                                 max_norm=3.0, l1_beta=2.0, start_steps=0, layer_results="end", loss_scale=-1, stochastic=True,
                                 lambda_pretraining=1e-2, **fl)
        out[f"{name}/loss"], out[f"{name}/grad_norm"] = np.float64(st["loss"]), np.float64(float(st["grad_norm"]))
        pn = [n for n, _ in model.named_parameters()]
        grads = {n: g for n, g in zip(pn, scaler.grads) if g is not None}
        for k in ("lm_head.weight", "cov_lm_head.weight", "blocks.2.mlp.fc2.weight", "blocks.0.attn.qkv.weight", "norm.weight"):
            put(out, f"{name}/grad/{k}", grads[k])
        print("dist target-norm case", name, "loss", st["loss"], "gnorm", float(st["grad_norm"]))
    np.savez_compressed(os.path.join(OUT, "dist_target_norms.npz"), **out)


def gen_loss_curve(mc, eng):
    img, dim, depth, heads, B, n_mask = 48, 128, 2, 2, 4, 5
    model = build(mc, img, dim, depth, heads, 0.1)
    fixed = [(closed_form_images(f"curve/{s}", B, img), exact_masks(B, 9, n_mask, 300 + s)) for s in range(4)]
    batches = [fixed[s % 4] for s in range(100)]
    model.train()
    rec, _, _, _ = run_reference_steps(eng, model, batches, [1], lr=CURVE_LR)
    out = {"cfg": np.array([img, dim, depth, heads, B, n_mask, 100], dtype=np.int64),
           "loss": np.array(rec["loss"]), "grad_norm": np.array(rec["grad_norm"]), "lr": np.float64(CURVE_LR)}
    for s, (_, bm) in enumerate(fixed):
        out[f"mask{s}"] = bm.numpy()
    np.savez_compressed(os.path.join(OUT, "loss_curve.npz"), **out)
    print("wrote loss curve", rec["loss"][:3], "...", rec["loss"][-3:])


def gen_loss_curve_vitb(mc, eng):
    """north_star's loss-curve criterion at the ViT-B/16 shape it is stated for: the reference's train_one_epoch,
    base model, B=2, dropout 0, four fixed batches cycled for 100 steps, constant lr (SURVEY 8c G5 "base config")."""
    B, n_mask = 2, 120
    model = build(mc, 224, 768, 12, 12, 0.1)
    fixed = [(closed_form_images(f"curveB/{s}", B, 224), exact_masks(B, 196, n_mask, 500 + s)) for s in range(4)]
    batches = [fixed[s % 4] for s in range(100)]
    model.train()
    rec, _, _, _ = run_reference_steps(eng, model, batches, list(range(6, 12)), lr=CURVE_LR)
    out = {"cfg": np.array([224, 768, 12, 12, B, n_mask, 100], dtype=np.int64),
           "loss": np.array(rec["loss"]), "grad_norm": np.array(rec["grad_norm"]), "lr": np.float64(CURVE_LR)}
    for s, (_, bm) in enumerate(fixed):
        out[f"mask{s}"] = bm.numpy()
    np.savez_compressed(os.path.join(OUT, "loss_curve_vitb.npz"), **out)
    print("wrote ViT-B loss curve", rec["loss"][:3], "...", rec["loss"][-3:])


def gen_dist_vitb_spot(mc, eng):
    """Two-stream model at the ViT-B shape (BASELINE config 3's architecture), B=2, closed-form weights: per-layer
    checksums + samples of both streams, the student heads, and ONE reference train_one_epoch(stochastic=True) step
    (loss, grad-norm, gradients of blocks 0 / 11 and the non-block parameters, names of the gradient-less tensors)."""
    import modeling_cyclical_dist as mcd
    import optim_factory
    import timm.utils as U
    model = mcd.DistVisionTransformerForCyclicalTraining(
        img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), init_values=0.1, use_shared_rel_pos_bias=True,
        use_abs_pos_emb=False, drop_path_rate=0.0, attn_drop_rate=0.0)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if v.dtype == torch.float32}
    model.load_state_dict(closed_form_state(shapes, gamma=0.1), strict=False)
    B = 2
    x = closed_form_images("dvitb", B, 224)
    mask = exact_masks(B, 196, 120, 13)
    out = {"mask": mask.numpy(), "n_params": np.int64(sum(p.numel() for p in model.parameters()))}
    model.eval()
    with torch.no_grad():
        em, ec = model(x, None, True, layer_results="end")
        sm, sc = model(x, mask, return_all_tokens=False)
    for i in range(12):
        put(out, f"mean_end{i}", em[i])
        put(out, f"cov_end{i}", ec[i])
    put(out, "student_mean", sm)
    put(out, "student_cov", sc)
    model.train()
    ema = U.ModelEmaV2(model, decay=0.9998)
    args = SimpleNamespace(opt="adamw", lr=2e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=(0.9, 0.999), momentum=0.9)
    opt = optim_factory.create_optimizer(args, model)
    scaler = ref_harness.HarnessScaler()
    st = eng.train_one_epoch(model, ema, 0, 0.9998, 0.9998, list(range(6, 12)), [((x, mask), torch.zeros(1))], opt,
                             torch.device("cpu"), 0, scaler, max_norm=3.0, l1_beta=2.0, start_steps=0, layer_results="end",
                             loss_scale=-1, target_layer_norm_last=True, post_target_layer_norm=True, stochastic=True,
                             lambda_pretraining=1e-2)
    out["step/loss"], out["step/grad_norm"] = np.float64(st["loss"]), np.float64(float(st["grad_norm"]))
    pn = [n for n, _ in model.named_parameters()]
    out["grad0_none"] = np.array([n for n, g in zip(pn, scaler.grads) if g is None])
    for n, g in zip(pn, scaler.grads):
        if g is None or (n.startswith("blocks.") and not (n.startswith("blocks.0.") or n.startswith("blocks.11."))):
            continue
        put(out, "grad0/" + n, g)
        out["grad0/" + n + "/l2"] = np.float64(g.double().norm().item())
    np.savez_compressed(os.path.join(OUT, "dist_vitb_spot.npz"), **out)
    print("wrote dist vitb spot; loss", st["loss"], "gnorm", float(st["grad_norm"]))


def gen_vitb_spot(mc):
    """ViT-B shape, B=2, closed-form weights: per-layer checksums + samples, and gradients of
    smooth_l1(student, 0) for blocks 0 and 11 and the non-block parameters."""
    model = build(mc, 224, 768, 12, 12, 0.1)
    B = 2
    x = closed_form_images("vitb", B, 224)
    mask = exact_masks(B, 196, 120, 11)
    out = {"mask": mask.numpy()}
    model.eval()
    with torch.no_grad():
        ends = model(x, None, True, layer_results="end")
    for li, e in enumerate(ends):
        put(out, f"end{li}", e)
    stu = model(x, mask, return_all_tokens=False)
    put(out, "student", stu)
    loss = torch.nn.functional.smooth_l1_loss(stu, torch.zeros_like(stu), beta=2.0)
    loss.backward()
    out["loss_vs_zero"] = np.float64(loss.item())
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        if n.startswith("blocks.") and not (n.startswith("blocks.0.") or n.startswith("blocks.11.")):
            continue
        put(out, f"grad/{n}", p.grad)
    out["rel_index_14"] = model.rel_pos_bias.relative_position_index.numpy()
    np.savez_compressed(os.path.join(OUT, "vitb_spot.npz"), **out)
    print("wrote vitb spot; loss_vs_zero", loss.item())


def gen_schedules():
    import utils
    out = {}
    out["cos_a"] = utils.cosine_scheduler(2e-3, 1e-5, 4, 10, warmup_epochs=1, start_warmup_value=1e-6)
    out["cos_b"] = utils.cosine_scheduler(0.05, 0.05, 3, 7)
    out["cos_c"] = utils.cosine_scheduler(5e-4, 1e-6, 5, 9, warmup_epochs=2, start_warmup_value=1e-6, warmup_steps=4)
    out["tri_a"] = utils.tri_phase_scheduler(2e-3, 1e-5, 5, 20, warmup_perc=0.05, decay_perc=0.15, start_warmup_value=1e-6)
    out["tri_b"] = utils.tri_phase_scheduler(1e-3, 0.0, 2, 10, warmup_perc=0.0, decay_perc=0.5)
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **out)
    print("wrote schedules")


def gen_masks():
    """Reference MaskingGenerator (masking_generator.py:29-92) under seeded global `random`; np.int no longer
    exists in NumPy >= 1.24, so the harness restores the alias for the call."""
    import random
    import masking_generator as mg
    if not hasattr(np, "int"):
        np.int = int
    out = {}
    for name, (size, n, mn, mx) in {"a": (14, 120, 16, None), "b": (14, 75, 16, None), "c": (14, 120, 4, 40), "d": (7, 20, 4, None)}.items():
        for seed in (0, 1, 2):
            random.seed(1000 * seed + 7)
            g = mg.MaskingGenerator(size, n, min_num_patches=mn, max_num_patches=mx)
            out[f"{name}{seed}"] = np.stack([np.asarray(g(), dtype=np.int64) for _ in range(4)])
    np.savez_compressed(os.path.join(OUT, "masks.npz"), **out)
    print("wrote masks", {k: int(v.sum()) for k, v in list(out.items())[:4]})


def gen_cli_defaults():
    """Flag names / defaults of the reference CLI: exec the get_args() definition of run_cyclical.py:36-284
    (the module itself cannot be imported: datasets.py needs the missing cifar_semi)."""
    import json
    src = open(os.path.join(ref_harness.REFERENCE, "run_cyclical.py")).read()
    body = src[src.index("def get_args():"):src.index("def get_model(args):")]
    ns = {"argparse": argparse}
    exec(compile(body, "run_cyclical.get_args", "exec"), ns)
    old = sys.argv
    sys.argv = ["run_cyclical.py"]
    try:
        a = ns["get_args"]()
    finally:
        sys.argv = old
    with open(os.path.join(OUT, "cli_defaults.json"), "w") as f:
        json.dump(vars(a), f, indent=1, sort_keys=True)
    print("wrote cli defaults:", len(vars(a)), "flags")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    mc, eng = ref_harness.import_reference()
    if a.only in (None, "model"):
        gen_model_case(mc, eng, "t48", img=48, dim=128, depth=2, heads=2, init_values=0.1, B=3, n_mask=4, seed=1)
        gen_model_case(mc, eng, "t32", img=32, dim=192, depth=3, heads=3, init_values=1e-4, B=2, n_mask=2, seed=2)
    if a.only in (None, "dist"):
        gen_dist_case(mc, eng, "d48", img=48, dim=128, depth=2, heads=2, init_values=0.1, B=3, n_mask=4, seed=41)
    if a.only in (None, "curve"):
        gen_loss_curve(mc, eng)
    if a.only in (None, "vitb"):
        gen_vitb_spot(mc)
    if a.only in (None, "tnorm"):
        gen_target_norm_cases(mc, eng)
    if a.only in (None, "dtnorm"):
        gen_dist_target_norm_cases(mc, eng)
    if a.only in (None, "abspos"):
        gen_abs_pos_case(mc, eng)
    if a.only in (None, "flags"):
        gen_flag_case(mc, eng)
    if a.only in (None, "dvitb"):
        gen_dist_vitb_spot(mc, eng)
    if a.only in (None, "curveB"):
        gen_loss_curve_vitb(mc, eng)
    if a.only in (None, "sched"):
        gen_schedules()
    if a.only in (None, "masks"):
        gen_masks()
    if a.only in (None, "cli"):
        gen_cli_defaults()


if __name__ == "__main__":
    main()
