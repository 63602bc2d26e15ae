"""GPU parity of the whole hot path (host mirror -> C ABI -> HIP kernels) against (a) the
committed golden vectors produced by the reference and (b) the oracle run on the same inputs.

Tolerances: GEMM operands and saved activations are bf16 (8-bit mantissa, eps = 3.9e-3) with fp32
accumulation, the residual stream / LayerNorm statistics / loss / optimizer are fp32.  Stated
per check below; north_star's loss-curve bound (1e-3 over 100 steps) is tested as such."""
import os

import numpy as np
import pytest
import torch

from golden_util import check_entry, entries
from oracle import vit_oracle as vo
from oracle.closed_form import closed_form_images, closed_form_state, exact_masks
from gpu_util import (assert_grads_close, full_size_step_properties, native_model, native_steps, native_trainer,
                      oracle_state)

pytestmark = pytest.mark.gpu

ACT_RT, ACT_AT = 2e-2, 2e-2      # activations O(1): bf16 round-off through depth layers


def load_case(golden_dir, tag):
    fx = np.load(os.path.join(golden_dir, f"model_{tag}.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=float(fx["init_values"]))
    return fx, cfg, B, n_mask, steps


@pytest.mark.parametrize("tag", ["t48", "t32"])
def test_forward_modes_vs_golden(golden_dir, tag):
    fx, cfg, B, n_mask, _ = load_case(golden_dir, tag)
    model, _ = native_model(cfg)
    model.eval()
    x = closed_form_images(f"{tag}/0", B, cfg.img_size).cuda()
    mask = torch.from_numpy(fx["mask0"]).cuda()
    ends = model(x, None, True, layer_results="end")
    for i in range(cfg.depth):
        check_entry(fx, f"fwd/end{i}", ends[i], ACT_RT, ACT_AT)
    fcs = model(x, None, True, layer_results="fc")
    for i in range(cfg.depth):
        # fc = gamma * mlp(...): magnitude ~ init_values
        check_entry(fx, f"fwd/fc{i}", fcs[i], 5e-2, 3e-2 * cfg.init_values + 1e-6)
    check_entry(fx, "fwd/student_masked", model(x, mask, return_all_tokens=False), ACT_RT, ACT_AT)
    check_entry(fx, "fwd/student_all", model(x, mask, return_all_tokens=True), ACT_RT, ACT_AT)


@pytest.mark.parametrize("tag", ["t48", "t32"])
def test_train_steps_vs_golden(golden_dir, tag):
    fx, cfg, B, n_mask, steps = load_case(golden_dir, tag)
    model, sd0 = native_model(cfg)
    ema, opt = native_trainer(model)
    tl = [int(v) for v in fx["target_layers"]]
    batches = [(closed_form_images(f"{tag}/{s}", B, cfg.img_size).cuda(), torch.from_numpy(fx[f"mask{s}"]).cuda())
               for s in range(steps)]
    st = native_steps(model, ema, opt, batches[:1], tl)
    assert st[0]["loss"] == pytest.approx(float(fx["step/loss"][0]), rel=5e-3)
    assert st[0]["grad_norm"] == pytest.approx(float(fx["step/grad_norm"][0]), rel=3e-2)
    # gradients of the first step (raw, before clipping) live in the flat gradient arena
    grads = {n: p.grad for n, p in model.named_parameters()}
    gmax = max(float(np.abs(fx[k]).max()) for k in fx.files if k.startswith("grad0/") and not k.endswith("/sum"))
    for n in entries(fx, "grad0"):
        assert grads[n] is not None, n
        check_entry(fx, "grad0/" + n, grads[n], 5e-2, 2e-2 * gmax, what=f"[{tag}] ")
    # every tensor again, whole, against the oracle (itself pinned to this fixture by tests/test_oracle_golden.py):
    # max-norm AND relative-L2 bound per tensor
    p, e, m1, v1 = oracle_state(sd0)
    ref = vo.train_step(p, e, m1, v1, cfg, vo.StepHParams(target_layers=tuple(tl)), batches[0][0].cpu(), batches[0][1].cpu(), 1)
    assert_grads_close({n: g.clone() for n, g in grads.items()}, ref.grads, what=f"[{tag}] ")
    st += native_steps(model, ema, opt, batches[1:], tl, start=1)
    for s in range(1, steps):
        assert st[s]["loss"] == pytest.approx(float(fx["step/loss"][s]), rel=2e-2)
    sd, esd = model.state_dict(), ema.module.state_dict()
    for n in entries(fx, "post"):
        check_entry(fx, "post/" + n, sd[n], 0, 3 * 2e-3 + 1e-4, what="post ")     # <= 3 AdamW steps of lr each
    for n in entries(fx, "ema"):
        check_entry(fx, "ema/" + n, esd[n], 0, 3 * 2e-3 * 2e-4 + 1e-5, what="ema ")


def test_loss_curve_100_steps(golden_dir):
    """north_star: loss curve within 1e-3 of the reference CPU path over 100 synthetic steps."""
    fx = np.load(os.path.join(golden_dir, "loss_curve.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1)
    model, _ = native_model(cfg)
    ema, opt = native_trainer(model, lr=float(fx["lr"]))
    fixed = [(closed_form_images(f"curve/{s}", B, img).cuda(), torch.from_numpy(fx[f"mask{s}"]).cuda()) for s in range(4)]
    st = native_steps(model, ema, opt, [fixed[s % 4] for s in range(steps)], [1])
    losses = np.array([s["loss"] for s in st])
    err = np.abs(losses - fx["loss"])
    print("max |loss - reference| over 100 steps:", err.max())
    assert err.max() < 1e-3, (err.max(), int(err.argmax()))


def test_vitb_forward_vs_golden(golden_dir):
    fx = np.load(os.path.join(golden_dir, "vitb_spot.npz"))
    cfg = vo.VitConfig(init_values=0.1)
    model, _ = native_model(cfg)
    model.eval()
    x = closed_form_images("vitb", 2, 224).cuda()
    mask = torch.from_numpy(fx["mask"]).cuda()
    ends = model(x, None, True, layer_results="end")
    for i in range(12):
        check_entry(fx, f"end{i}", ends[i], ACT_RT, ACT_AT)
    check_entry(fx, "student", model(x, mask, return_all_tokens=False), ACT_RT, ACT_AT)


def test_vitb_step_vs_oracle():
    """ViT-B/16 224, B=2, one full step: HIP path vs the oracle on the host cores."""
    cfg = vo.VitConfig(init_values=0.1)
    model, sd = native_model(cfg)
    ema, opt = native_trainer(model)
    x = closed_form_images("vitb-step", 2, 224)
    mask = exact_masks(2, 196, 120, 21)
    st = native_steps(model, ema, opt, [(x.cuda(), mask.cuda())], list(range(6, 12)))[0]
    p = {k: v.clone() for k, v in sd.items()}
    e = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    ref = vo.train_step(p, e, m, v, cfg, vo.StepHParams(), x, mask, 1)
    assert st["loss"] == pytest.approx(ref.loss, rel=5e-3)
    assert st["grad_norm"] == pytest.approx(ref.grad_norm, rel=3e-2)
    grads = {n: q.grad for n, q in model.named_parameters()}
    assert_grads_close(grads, ref.grads, what="[ViT-B] ")      # EVERY tensor: max-norm and relative-L2 bound
    # The regression targets: the product's teacher runs its GEMMs on bf16 operands (deviation 1), the reference's
    # teacher is fp32.  Measured error of the (M, 768) targets -- post-LayerNorm values, rms 1 -- against the fp32 oracle:
    M = int(mask.sum())
    tgt = model._engine.ws_tensor("targets", 0, (M, cfg.embed_dim)).cpu()
    d = (tgt - ref.targets).double()
    rel_l2, max_abs = float(d.norm() / ref.targets.double().norm()), float(d.abs().max())
    print(f"[ViT-B] teacher targets (bf16 GEMM operands) vs fp32 oracle: relative L2 {rel_l2:.3e}, max |diff| {max_abs:.3e} "
          f"(target rms {float(ref.targets.double().pow(2).mean().sqrt()):.3f}, max {float(ref.targets.abs().max()):.2f})")
    assert rel_l2 < 1e-2 and max_abs < 0.1
    esd = ema.module.state_dict()
    for n in ["blocks.4.mlp.fc1.weight", "norm.weight"]:
        # first AdamW step moves every weight by +-lr; a sign flip on a ~0 gradient is 2*lr apart
        torch.testing.assert_close(esd[n].cpu(), e[n], rtol=0, atol=2 * 2e-3 * 2e-4 + 1e-7)


@pytest.mark.parametrize("dropout", [False, True])
def test_last_block_on_masked_rows_equals_all_rows(dropout):
    """uvit_step_params.n_rows_hint (include/uvit.h): with a host-side bound on the masked patches -- the loader's mask is a CPU tensor --
    the last block's MLP runs on those rows only.  Same model, same batch, same seeds: the step must give the loss and the gradients of the
    all-rows step (the fc1 / fc2 weight gradients sum fewer zero rows in another order, everything else is the same arithmetic), with and
    without dropout / drop-path (the per-sample drop-path scale and the LayerScale gradient go through the row list), and with ragged masks
    (another count per sample).  ViT-B/16, B = 8: 8 x 197 = 1,576 token rows, 529 masked -> 576 compact rows."""
    cfg = vo.VitConfig(init_values=0.1, attn_drop_rate=0.05 if dropout else 0.0, drop_path_rate=0.25 if dropout else 0.0)
    x = closed_form_images("rows", 8, 224)
    mask = exact_masks(8, 196, 75, 5)
    mask[3, :] = False; mask[3, :4] = True          # ragged: 4 masked patches in one sample, 75 in the others
    res = {}
    for mode in ("all", "rows"):
        model, _ = native_model(cfg)
        ema, opt = native_trainer(model)
        if dropout:
            model.train()
        batch = (x.cuda(), mask.cuda()) if mode == "all" else (x, mask)       # a CPU mask is counted on the host: the row bound
        st = native_steps(model, ema, opt, [batch], list(range(6, 12)))[0]
        res[mode] = (st, {n: q.grad.detach().float().cpu().clone() for n, q in model.named_parameters()})
        assert (model._engine.compact_rows() > 0) == (mode == "rows")
    (sa, ga), (sr, gr) = res["all"], res["rows"]
    assert sr["loss"] == pytest.approx(sa["loss"], rel=1e-6) and sr["grad_norm"] == pytest.approx(sa["grad_norm"], rel=1e-5)
    assert_grads_close(gr, ga, max_tol=1e-4, l2_tol=1e-4, what="[masked rows vs all rows] ")


@pytest.mark.parametrize("variant", ["l2_loss", "beta_small", "loss_scale", "no_post_ln", "no_clip", "ema_off", "ragged_masks"])
def test_step_hyperparameter_variants_vs_oracle(variant):
    """The knobs of train_one_epoch that BASELINE's recipes toggle (engine_for_cyclical.py:24-32, 147-163, 182-185;
    utils.py:375-381): two steps of the HIP path against the oracle on a small model."""
    cfg = vo.VitConfig(img_size=48, embed_dim=128, depth=3, num_heads=2, init_values=0.1)
    kw = dict(l2_loss=False, l1_beta=2.0, loss_scale=-1, post_target_layer_norm=True, clip=3.0, decay=0.9998)
    hp = dict(l2_loss=False, l1_beta=2.0, loss_scale=-1, post_target_layer_norm=True, clip_grad=3.0, ema_decay=0.9998)
    if variant == "l2_loss":
        kw["l2_loss"] = hp["l2_loss"] = True
    elif variant == "beta_small":
        kw["l1_beta"] = hp["l1_beta"] = 0.12
    elif variant == "loss_scale":
        kw["loss_scale"] = hp["loss_scale"] = 8.0
    elif variant == "no_post_ln":
        kw["post_target_layer_norm"] = hp["post_target_layer_norm"] = False
    elif variant == "no_clip":
        kw["clip"], hp["clip_grad"] = None, None
    elif variant == "ema_off":
        kw["decay"] = hp["ema_decay"] = 1.0
    model, sd = native_model(cfg)
    ema, opt = native_trainer(model, decay=kw["decay"])
    B = 5
    batches = [(closed_form_images(f"hv{variant}{s}", B, 48), exact_masks(B, 9, 4, 40 + s)) for s in range(2)]
    if variant == "ragged_masks":      # the block-wise generator returns AT MOST n ones: 0..9 masked patches per image
        for s, counts in enumerate(([0, 9, 1, 4, 7], [3, 0, 0, 9, 2])):
            mk = torch.zeros(B, 9, dtype=torch.int64)
            for i, n in enumerate(counts):
                mk[i, torch.arange(9).roll(i + s)[:n]] = 1
            batches[s] = (batches[s][0], mk.view(B, 3, 3))
    st = native_steps(model, ema, opt, [(x.cuda(), m.cuda()) for x, m in batches], [1, 2], **kw)
    p = {k: v.clone() for k, v in sd.items()}
    e = {k: v.clone() for k, v in sd.items()}
    m1 = {k: torch.zeros_like(v) for k, v in p.items()}
    v1 = {k: torch.zeros_like(t) for k, t in p.items()}
    ohp = vo.StepHParams(target_layers=(1, 2), **hp)
    for s, (x, mk) in enumerate(batches):
        ref = vo.train_step(p, e, m1, v1, cfg, ohp, x, mk, s + 1)
        assert st[s]["loss"] == pytest.approx(ref.loss, rel=2e-2 if s else 5e-3), (variant, s)
        assert st[s]["grad_norm"] == pytest.approx(ref.grad_norm, rel=5e-2), (variant, s)
    esd = ema.module.state_dict()
    for n in ["blocks.1.mlp.fc1.weight", "norm.weight", "blocks.2.attn.qkv.weight"]:
        tol = 0.0 if variant == "ema_off" else 6 * 2e-3 * 2e-4 + 1e-7     # sign flips: weights differ <= 2 lr after step 1, 4 lr after step 2
        torch.testing.assert_close(esd[n].cpu(), e[n], rtol=0, atol=tol)
    sdn = model.state_dict()
    for n in ["blocks.0.mlp.fc2.weight", "lm_head.weight"]:
        torch.testing.assert_close(sdn[n].cpu(), p[n], rtol=0, atol=2 * 2 * 2e-3 + 1e-6)


def test_dropout_step_matches_oracle_with_replayed_masks():
    """attn_drop 0.1 + drop_path 0.3: the oracle replays the kernel's counter-based masks."""
    cfg = vo.VitConfig(img_size=48, embed_dim=128, depth=3, num_heads=2, init_values=0.1, drop_path_rate=0.3, attn_drop_rate=0.1)
    model, sd = native_model(cfg)
    ema, opt = native_trainer(model)
    B = 6
    x = closed_form_images("drop", B, 48)
    mask = exact_masks(B, 9, 4, 31)
    torch.manual_seed(1234)
    st = native_steps(model, ema, opt, [(x.cuda(), mask.cuda())], [1, 2], start=7)[0]
    seed, it = torch.initial_seed() & 0xFFFFFFFF, 7
    aseed = int(vo._mix32(np.uint32(seed) ^ np.uint32((it * 0x85EBCA6B + 0x1234567) & 0xFFFFFFFF)))
    p1, p2 = vo.drop_path_scales(seed, it, cfg, B)
    drop = vo.DropState(path1=p1, path2=p2, attn=[vo.attn_keep_mask(aseed, l, B, 2, 10, 0.1) for l in range(3)])
    p = {k: v.clone() for k, v in sd.items()}
    e = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    ref = vo.train_step(p, e, m, v, cfg, vo.StepHParams(target_layers=(1, 2)), x, mask, 1, drop=drop)
    assert st["loss"] == pytest.approx(ref.loss, rel=5e-3)
    assert st["grad_norm"] == pytest.approx(ref.grad_norm, rel=3e-2)
    assert any(t is not None and (t == 0).any() for t in p1 + p2), "test should exercise a dropped path"
    # the step ran with the drop-path sample lists (compact branches); every gradient against the oracle, which multiplies dropped branches by 0
    assert model._engine.drop_path_rows
    grads = {n: q.grad for n, q in model.named_parameters()}
    assert_grads_close(grads, ref.grads, what="[drop-path lists vs oracle] ")


def test_vit_large_step_vs_oracle():
    """beit_large_patch16_224 (embed 1024, depth 24, heads 16; BASELINE config 5's architecture), B=1: one full step."""
    cfg = vo.VitConfig(embed_dim=1024, depth=24, num_heads=16, init_values=0.1)
    model, sd = native_model(cfg)
    ema, opt = native_trainer(model)
    x = closed_form_images("vitl-step", 1, 224)
    mask = exact_masks(1, 196, 120, 23)
    st = native_steps(model, ema, opt, [(x.cuda(), mask.cuda())], list(range(12, 24)))[0]
    p = {k: v.clone() for k, v in sd.items()}
    e = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    ref = vo.train_step(p, e, m, v, cfg, vo.StepHParams(target_layers=tuple(range(12, 24))), x, mask, 1)
    assert st["loss"] == pytest.approx(ref.loss, rel=1e-2)
    assert st["grad_norm"] == pytest.approx(ref.grad_norm, rel=5e-2)
    grads = {n: q.grad for n, q in model.named_parameters()}
    assert_grads_close(grads, ref.grads, max_tol=8e-2, l2_tol=3e-2, what="[ViT-L] ")      # 24 blocks of bf16 round-off


def test_full_size_step_properties():
    """BASELINE config 2, the headline configuration (ViT-B/16, bs=128, 120 masked patches, attn-drop 0.05, drop-path
    0.25, clip 3): size-independent properties of one full step (tests/gpu_util.py:full_size_step_properties)."""
    full_size_step_properties(vo.VitConfig(init_values=1e-4, drop_path_rate=0.25, attn_drop_rate=0.05),
                              vo.VitConfig(init_values=1e-4), B=128, img=224, n_patches=196, n_mask=120,
                              target_layers=list(range(6, 12)))


def test_loss_curve_vitb_100_steps(golden_dir):
    """north_star, at the shape it is stated for: the ViT-B/16 loss curve stays within 1e-3 of the reference's CPU
    train_one_epoch over 100 synthetic steps (B=2, four fixed batches, dropout 0, constant lr; fixture written by
    tools/gen_golden.py:gen_loss_curve_vitb from the unmodified reference).  bf16 GEMM operands through 12 blocks and
    100 AdamW updates of every weight are exactly what this bounds."""
    fx = np.load(os.path.join(golden_dir, "loss_curve_vitb.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1)
    model, _ = native_model(cfg)
    ema, opt = native_trainer(model, lr=float(fx["lr"]))
    fixed = [(closed_form_images(f"curveB/{s}", B, img).cuda(), torch.from_numpy(fx[f"mask{s}"]).cuda()) for s in range(4)]
    st = native_steps(model, ema, opt, [fixed[s % 4] for s in range(steps)], list(range(6, 12)))
    losses = np.array([s["loss"] for s in st])
    err = np.abs(losses - fx["loss"])
    print("ViT-B/16: max |loss - reference| over 100 steps:", err.max(), "at step", int(err.argmax()),
          "| loss", fx["loss"][0], "->", fx["loss"][-1])
    assert err.max() < 1e-3, (err.max(), int(err.argmax()))
    gn = np.array([s["grad_norm"] for s in st])
    assert np.all(np.abs(gn - fx["grad_norm"]) <= 5e-2 * fx["grad_norm"] + 1e-3), np.abs(gn - fx["grad_norm"]).max()


def test_flag_gated_fc_targets_and_variance_term(golden_dir):
    """`--layer_results fc` targets and the variance term (`--var_w0 1 --var_margin0 1`; engine_for_cyclical.py:88-139): two
    steps of the HIP path against the reference's own numbers (tests/golden/model_flags.npz) and, per tensor, the oracle."""
    fx = np.load(os.path.join(golden_dir, "model_flags.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1)
    model, sd0 = native_model(cfg)
    ema, opt = native_trainer(model)
    batches = [(closed_form_images(f"flags/{s}", B, img), torch.from_numpy(fx[f"mask{s}"])) for s in range(steps)]
    kw = dict(layer_results="fc", var_w0=1.0, var_margin0=1.0)
    st = native_steps(model, ema, opt, [(batches[0][0].cuda(), batches[0][1].cuda())], [1, 2], **kw)
    assert st[0]["loss"] == pytest.approx(float(fx["loss"][0]), rel=5e-3)
    assert st[0]["loss_var0"] == pytest.approx(float(fx["loss_var0"][0]), rel=5e-3) and st[0]["loss_var0"] > 0.5
    assert st[0]["grad_norm"] == pytest.approx(float(fx["grad_norm"][0]), rel=3e-2)
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    p, e, m1, v1 = oracle_state(sd0)
    hp = vo.StepHParams(target_layers=(1, 2), layer_results="fc", var_w0=1.0, var_margin0=1.0)
    ref = vo.train_step(p, e, m1, v1, cfg, hp, batches[0][0], batches[0][1], 1)
    assert_grads_close(grads, ref.grads, what="[flags] ")
    st += native_steps(model, ema, opt, [(batches[1][0].cuda(), batches[1][1].cuda())], [1, 2], start=1, **kw)
    assert st[1]["loss"] == pytest.approx(float(fx["loss"][1]), rel=2e-2)
    assert st[1]["loss_var0"] == pytest.approx(float(fx["loss_var0"][1]), rel=2e-2)


def test_abs_pos_emb_forward_and_steps(golden_dir):
    """`--abs_pos_emb` (pos_embed added after the cls concat, gradient = batch sum, no-decay group): forward and two steps
    against the reference's numbers (tests/golden/model_abspos.npz) and the oracle's gradients."""
    fx = np.load(os.path.join(golden_dir, "model_abspos.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1, use_abs_pos_emb=True)
    model, sd0 = native_model(cfg)
    assert [n for n in model.state_dict() if not n.endswith("relative_position_index")] == \
        [n for n in fx["names"].tolist() if not n.endswith("relative_position_index")]
    model.eval()
    x0, m0 = closed_form_images("abspos/0", B, img), torch.from_numpy(fx["mask0"])
    ends = model(x0.cuda(), None, True, layer_results="end")
    for i in range(depth):
        check_entry(fx, f"fwd/end{i}", ends[i], ACT_RT, ACT_AT)
    check_entry(fx, "fwd/student_masked", model(x0.cuda(), m0.cuda(), return_all_tokens=False), ACT_RT, ACT_AT)
    ema, opt = native_trainer(model)
    assert "pos_embed" in opt.group_names["no_decay"]
    st = native_steps(model, ema, opt, [(x0.cuda(), m0.cuda())], [1])
    assert st[0]["loss"] == pytest.approx(float(fx["loss"][0]), rel=5e-3)
    assert st[0]["grad_norm"] == pytest.approx(float(fx["grad_norm"][0]), rel=3e-2)
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    p, e, m1, v1 = oracle_state(sd0)
    ref = vo.train_step(p, e, m1, v1, cfg, vo.StepHParams(target_layers=(1,)), x0, m0, 1)
    assert_grads_close(grads, ref.grads, what="[abs-pos] ")
    st += native_steps(model, ema, opt, [(closed_form_images("abspos/1", B, img).cuda(), torch.from_numpy(fx["mask1"]).cuda())], [1], start=1)
    assert st[1]["loss"] == pytest.approx(float(fx["loss"][1]), rel=2e-2)


TNORM_CASES = {   # the flag combinations of tests/golden/target_norms.npz (tools/gen_golden.py TARGET_NORM_CASES)
    "bn":        dict(target_batch_norm=True, target_instance_norm=False, target_layer_norm_last=True, post_target_instance_norm=False, post_target_layer_norm=True),
    "in":        dict(target_batch_norm=False, target_instance_norm=True, target_layer_norm_last=True, post_target_instance_norm=False, post_target_layer_norm=False),
    "bn_in_pin": dict(target_batch_norm=True, target_instance_norm=True, target_layer_norm_last=False, post_target_instance_norm=True, post_target_layer_norm=True),
    "raw_pin":   dict(target_batch_norm=False, target_instance_norm=False, target_layer_norm_last=False, post_target_instance_norm=True, post_target_layer_norm=False),
}


@pytest.mark.parametrize("case", list(TNORM_CASES))
def test_batch_and_instance_norm_target_variants(golden_dir, case):
    """`--target_batch_norm`, `--target_instance_norm`, `--no_target_layer_norm_last`, `--post_target_instance_norm`
    (engine_for_cyclical.py:94-118): one step of the HIP path (dense target builder) per flag combination against the
    reference's numbers (tests/golden/target_norms.npz) and, per tensor, the oracle."""
    fx = np.load(os.path.join(golden_dir, "target_norms.npz"))
    img, dim, depth, heads, B, n_mask, _ = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1)
    model, sd0 = native_model(cfg)
    ema, opt = native_trainer(model)
    x, mask = closed_form_images("tnorm", B, img), torch.from_numpy(fx["mask"])
    fl = TNORM_CASES[case]
    st = native_steps(model, ema, opt, [(x.cuda(), mask.cuda())], [1, 2], **fl)
    assert st[0]["loss"] == pytest.approx(float(fx[f"{case}/loss"]), rel=5e-3)
    assert st[0]["grad_norm"] == pytest.approx(float(fx[f"{case}/grad_norm"]), rel=3e-2)
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    p, e, m1, v1 = oracle_state(sd0)
    ref = vo.train_step(p, e, m1, v1, cfg, vo.StepHParams(target_layers=(1, 2), **fl), x, mask, 1)
    assert_grads_close(grads, ref.grads, what=f"[{case}] ")
    for n in entries(fx, f"{case}/grad"):
        key = f"{case}/grad/{n}/full"
        if key in fx:
            g, r = grads[n].float().cpu().double(), torch.from_numpy(np.asarray(fx[key])).double()
            assert (g - r).norm() <= 2e-2 * r.norm(), (n, float((g - r).norm() / r.norm()))


@pytest.mark.parametrize("shape", ["tiny", "vitb8", "vitb32_hint", "wide1024"])
def test_drop_path_sample_lists_equal_all_samples(shape):
    """uvit_engine_set_drop_path_rows (include/uvit.h): a Block branch whose DropPath dropped a sample is multiplied by 0 for it in the
    forward and receives no gradient (modeling_finetune.py:51-62, 295-298), so the step runs each branch on the kept samples only, in compact
    rows.  Same model, batch and seeds with the lists on and off: same loss, same gradients (the weight gradients sum the same non-zero rows in
    another order), same targets.  `tiny` has ungrouped wgrads and 10-token samples (compact row counts that are no multiple of 64),
    `vitb8` is ViT-B/16 with drop_path 0.5 (a third of the samples dropped per branch on average), `vitb32_hint` adds the masked-row last
    block (n_rows_hint), whose MLP keeps its own row list while the layers below run the sample lists; `wide1024` has ViT-L's row length
    (C = 1024, 16 heads: the four-float4-per-lane instantiations of the LayerNorm kernels) on three blocks."""
    if shape == "wide1024":
        cfg = vo.VitConfig(embed_dim=1024, depth=3, num_heads=16, init_values=0.1, drop_path_rate=0.5, attn_drop_rate=0.05)
        B, img, P, nm, tl = 8, 224, 196, 75, [1, 2]
    elif shape == "tiny":
        cfg = vo.VitConfig(img_size=48, embed_dim=128, depth=4, num_heads=2, init_values=0.1, drop_path_rate=0.5, attn_drop_rate=0.1)
        B, img, P, nm, tl = 12, 48, 9, 4, [2, 3]
    else:
        cfg = vo.VitConfig(init_values=0.1, drop_path_rate=0.5, attn_drop_rate=0.05)
        B, img, P, nm, tl = (8 if shape == "vitb8" else 32), 224, 196, 75, list(range(6, 12))
    x = closed_form_images("dplists/" + shape, B, img)
    mask = exact_masks(B, P, nm, 11)
    res = {}
    for mode in ("all", "lists"):
        model, _ = native_model(cfg)
        model.drop_path_rows = mode == "lists"
        ema, opt = native_trainer(model)
        model.train()
        torch.manual_seed(4321)
        batch = (x, mask) if shape == "vitb32_hint" else (x.cuda(), mask.cuda())
        sts = native_steps(model, ema, opt, [batch], tl, start=3)
        grads = {n: q.grad.detach().float().cpu().clone() for n, q in model.named_parameters()}
        sts += native_steps(model, ema, opt, [batch], tl, start=4)          # other draws, on the updated weights
        res[mode] = (sts, grads, {n: q.detach().float().cpu().clone() for n, q in model.named_parameters()})
        assert model._engine.drop_path_rows == (mode == "lists")
    (sa, ga, pa), (sl, gl, pl) = res["all"], res["lists"]
    assert sl[0]["loss"] == pytest.approx(sa[0]["loss"], rel=2e-5) and sl[0]["grad_norm"] == pytest.approx(sa[0]["grad_norm"], rel=2e-4)
    assert_grads_close(gl, ga, max_tol=2e-3, l2_tol=2e-3, what="[drop-path lists vs all samples] ")
    # second step: AdamW's first update is lr * sign(g), so the summation-order noise on near-zero gradients moves a few weights by 2 lr
    assert sl[1]["loss"] == pytest.approx(sa[1]["loss"], rel=1e-3) and sl[1]["grad_norm"] == pytest.approx(sa[1]["grad_norm"], rel=2e-2)
    for n in ("blocks.1.mlp.fc1.weight", "blocks.2.attn.proj.weight", "blocks.0.norm1.weight"):
        torch.testing.assert_close(pl[n], pa[n], rtol=0, atol=2 * 2e-3 * 2 + 1e-6)     # two AdamW steps: a sign flip on a ~0 gradient is 2 lr apart
