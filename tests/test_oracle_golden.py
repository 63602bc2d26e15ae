"""CPU: the oracle restatement (oracle/vit_oracle.py) against vectors produced by the
reference itself (tests/golden/*.npz, written by tools/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from golden_util import check_entry, entries
from oracle import vit_oracle as vo
from oracle.closed_form import closed_form_images, closed_form_state, exact_masks

RT, AT = 2e-4, 1e-5   # fp32 round-off between two eager CPU formulations


def load_case(golden_dir, tag):
    fx = np.load(os.path.join(golden_dir, f"model_{tag}.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads,
                       init_values=float(fx["init_values"]))
    params = closed_form_state(vo.param_shapes(cfg), gamma=cfg.init_values)
    return fx, cfg, params, B, n_mask, steps


@pytest.mark.parametrize("tag", ["t48", "t32"])
def test_forward_modes(golden_dir, tag):
    fx, cfg, p, B, n_mask, _ = load_case(golden_dir, tag)
    x = closed_form_images(f"{tag}/0", B, cfg.img_size)
    mask = torch.from_numpy(fx["mask0"])
    assert torch.equal(mask, exact_masks(B, cfg.num_patches, n_mask, int(fx["mask_seed"])))
    ends = vo.forward(p, cfg, x, None, True, "end")
    fcs = vo.forward(p, cfg, x, None, True, "fc")
    for i in range(cfg.depth):
        check_entry(fx, f"fwd/end{i}", ends[i], RT, AT)
        check_entry(fx, f"fwd/fc{i}", fcs[i], RT, AT)
    check_entry(fx, "fwd/student_masked", vo.forward(p, cfg, x, mask, False), RT, AT)
    check_entry(fx, "fwd/student_all", vo.forward(p, cfg, x, mask, True), RT, AT)


@pytest.mark.parametrize("tag", ["t48", "t32"])
def test_train_steps(golden_dir, tag):
    fx, cfg, p, B, n_mask, steps = load_case(golden_dir, tag)
    hp = vo.StepHParams(target_layers=tuple(int(v) for v in fx["target_layers"]))
    ema = {k: v.clone() for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    for s in range(steps):
        x = closed_form_images(f"{tag}/{s}", B, cfg.img_size)
        mask = torch.from_numpy(fx[f"mask{s}"])
        res = vo.train_step(p, ema, m, v, cfg, hp, x, mask, s + 1)
        assert res.loss == pytest.approx(float(fx["step/loss"][s]), rel=2e-4)
        assert res.grad_norm == pytest.approx(float(fx["step/grad_norm"][s]), rel=2e-3)
        if s == 0:
            names = entries(fx, "grad0")
            assert set(names) == set(res.grads.keys())
            for n in names:
                check_entry(fx, "grad0/" + n, res.grads[n], 2e-3, 2e-7)
    for n in entries(fx, "post"):
        check_entry(fx, "post/" + n, p[n], 1e-3, 2e-5)
    for n in entries(fx, "ema"):
        check_entry(fx, "ema/" + n, ema[n], 1e-5, 1e-7)


def test_param_groups(golden_dir):
    fx, cfg, p, *_ = load_case(golden_dir, "t48")
    nd = vo.no_decay_names(p)
    assert nd == set(fx["groups/no_decay"].tolist())
    assert set(p) - nd == set(fx["groups/decay"].tolist())


def test_rel_pos_index_known_answer(golden_dir):
    fx = np.load(os.path.join(golden_dir, "vitb_spot.npz"))
    assert np.array_equal(vo.relative_position_index(14), fx["rel_index_14"])


def test_loss_curve_100_steps(golden_dir):
    """north_star: loss curve within 1e-3 of the reference CPU path over 100 steps."""
    fx = np.load(os.path.join(golden_dir, "loss_curve.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1)
    p = closed_form_state(vo.param_shapes(cfg), gamma=0.1)
    hp = vo.StepHParams(target_layers=(1,), lr=float(fx["lr"]))
    ema = {k: t.clone() for k, t in p.items()}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    fixed = [(closed_form_images(f"curve/{s}", B, img), torch.from_numpy(fx[f"mask{s}"])) for s in range(4)]
    losses = []
    for s in range(steps):
        x, mask = fixed[s % 4]
        losses.append(vo.train_step(p, ema, m, v, cfg, hp, x, mask, s + 1).loss)
    np.testing.assert_allclose(np.array(losses), fx["loss"], atol=1e-3, rtol=0)


def test_vitb_shape_spot(golden_dir):
    fx = np.load(os.path.join(golden_dir, "vitb_spot.npz"))
    cfg = vo.VitConfig(init_values=0.1)
    p = closed_form_state(vo.param_shapes(cfg), gamma=0.1)
    x = closed_form_images("vitb", 2, 224)
    mask = torch.from_numpy(fx["mask"])
    ends = vo.forward(p, cfg, x, None, True, "end")
    for i in range(12):
        check_entry(fx, f"end{i}", ends[i], 5e-4, 5e-6)
    leaves = {k: t.clone().requires_grad_(True) for k, t in p.items()}
    stu = vo.forward(leaves, cfg, x, mask, False)
    check_entry(fx, "student", stu, 5e-4, 5e-6)
    loss = torch.nn.functional.smooth_l1_loss(stu, torch.zeros_like(stu), beta=2.0)
    assert float(loss) == pytest.approx(float(fx["loss_vs_zero"]), rel=1e-4)
    loss.backward()
    for n in entries(fx, "grad"):
        check_entry(fx, "grad/" + n, leaves[n].grad, 2e-3, 1e-8)


def test_adamw_matches_torch_optim():
    torch.manual_seed(0)
    p = {"a.weight": torch.randn(5, 7), "a.bias": torch.randn(5)}
    ref = {k: t.clone().requires_grad_(True) for k, t in p.items()}
    opt = torch.optim.AdamW([{"params": [ref["a.weight"]], "weight_decay": 0.05},
                             {"params": [ref["a.bias"]], "weight_decay": 0.0}], lr=2e-3, eps=1e-8)
    hp = vo.StepHParams()
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    for s in range(4):
        g = {k: torch.randn_like(t) for k, t in p.items()}
        for k in ref:
            ref[k].grad = g[k].clone()
        opt.step()
        vo.adamw_step(p, g, m, v, s + 1, hp)
    for k in p:
        torch.testing.assert_close(p[k], ref[k].detach(), rtol=1e-5, atol=1e-7)


def test_two_stream_model_and_step(golden_dir):
    """Two-stream ("--stochastic") model: forward, Wasserstein loss, gradients, dead cov_qkv.weight, EMA."""
    from oracle import vit_oracle_dist as vd
    fx = np.load(os.path.join(golden_dir, "dist_d48.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=float(fx["init_values"]))
    assert list(vd.param_shapes(cfg)) == fx["names"].tolist()           # reference state-dict order
    p = closed_form_state(vd.param_shapes(cfg), gamma=cfg.init_values)
    x = closed_form_images("d48/0", B, img)
    mask = torch.from_numpy(fx["mask0"])
    em, ec = vd.forward(p, cfg, x, None, True, "end")
    for i in range(depth):
        check_entry(fx, f"fwd/mean_end{i}", em[i], RT, AT)
        check_entry(fx, f"fwd/cov_end{i}", ec[i], RT, AT)
    sm, sc = vd.forward(p, cfg, x, mask, False)
    check_entry(fx, "fwd/student_mean", sm, RT, AT)
    check_entry(fx, "fwd/student_cov", sc, RT, AT)
    hp = vo.StepHParams(target_layers=tuple(int(v) for v in fx["target_layers"]))
    ema = {k: t.clone() for k, t in p.items()}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    for s in range(steps):
        xs = closed_form_images(f"d48/{s}", B, img)
        res, lw, _, _ = vd.train_step(p, ema, m, v, cfg, hp, xs, torch.from_numpy(fx[f"mask{s}"]), s + 1, lam=1e-2)
        assert res.loss == pytest.approx(float(fx["step/loss"][s]), rel=2e-4)
        assert res.grad_norm == pytest.approx(float(fx["step/grad_norm"][s]), rel=2e-3)
        assert lw > 0
        if s == 0:
            assert set(fx["grad0_none"].tolist()) == {f"blocks.{i}.attn.cov_qkv.weight" for i in range(depth)}
            assert set(entries(fx, "grad0")) == set(res.grads)
            for n in entries(fx, "grad0"):
                check_entry(fx, "grad0/" + n, res.grads[n], 2e-3, 2e-7)
    for n in entries(fx, "post"):
        check_entry(fx, "post/" + n, p[n], 1e-3, 2e-5)
    for n in entries(fx, "ema"):
        check_entry(fx, "ema/" + n, ema[n], 1e-5, 1e-7)


def test_two_stream_vitb_shape_spot(golden_dir):
    """The two-stream oracle at the ViT-B/16 shape (12 heads, 197 tokens; BASELINE config 3's architecture): forward of
    both streams, and one stochastic=True step (loss, grad-norm, gradients) against the reference's own outputs."""
    from oracle import vit_oracle_dist as vd
    fx = np.load(os.path.join(golden_dir, "dist_vitb_spot.npz"))
    cfg = vo.VitConfig(init_values=0.1)
    p = closed_form_state(vd.param_shapes(cfg), gamma=0.1)
    assert sum(t.numel() for t in p.values()) == int(fx["n_params"])
    x = closed_form_images("dvitb", 2, 224)
    mask = torch.from_numpy(fx["mask"])
    em, ec = vd.forward(p, cfg, x, None, True, "end")
    for i in range(12):
        check_entry(fx, f"mean_end{i}", em[i], 5e-4, 5e-6)
        check_entry(fx, f"cov_end{i}", ec[i], 5e-4, 5e-6)
    ema = {k: t.clone() for k, t in p.items()}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    res, lw, _, _ = vd.train_step(p, ema, m, v, cfg, vo.StepHParams(target_layers=tuple(range(6, 12))), x, mask, 1, lam=1e-2)
    assert res.loss == pytest.approx(float(fx["step/loss"]), rel=2e-4)
    assert res.grad_norm == pytest.approx(float(fx["step/grad_norm"]), rel=2e-3)
    assert set(fx["grad0_none"].tolist()) == {f"blocks.{i}.attn.cov_qkv.weight" for i in range(12)}
    gmax = max(float(np.abs(fx[k]).max()) for k in fx.files if k.startswith("grad0/") and k.endswith(("/samples", "/full")))
    for n in entries(fx, "grad0"):
        check_entry(fx, "grad0/" + n, res.grads[n], 2e-3, 1e-5 * gmax)
        assert float(res.grads[n].double().norm()) == pytest.approx(float(fx["grad0/" + n + "/l2"]), rel=2e-3, abs=1e-6), n


def test_loss_curve_vitb_first_steps(golden_dir):
    """The ViT-B/16 loss-curve fixture (reference train_one_epoch, 100 steps): the oracle follows its first 12 steps --
    the fast-moving part of the curve -- to fp32 round-off (the whole curve is the GPU test's job; 100 ViT-B steps on
    the host cores would not fit the CPU suite's budget)."""
    fx = np.load(os.path.join(golden_dir, "loss_curve_vitb.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    assert (img, dim, depth, heads, steps) == (224, 768, 12, 12, 100)
    cfg = vo.VitConfig(init_values=0.1)
    p = closed_form_state(vo.param_shapes(cfg), gamma=0.1)
    hp = vo.StepHParams(target_layers=tuple(range(6, 12)), lr=float(fx["lr"]))
    ema = {k: t.clone() for k, t in p.items()}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    fixed = [(closed_form_images(f"curveB/{s}", B, img), torch.from_numpy(fx[f"mask{s}"])) for s in range(4)]
    for s in range(4):
        assert torch.equal(fixed[s][1], exact_masks(B, 196, n_mask, 500 + s))
    losses = [vo.train_step(p, ema, m, v, cfg, hp, *fixed[s % 4], s + 1).loss for s in range(12)]
    np.testing.assert_allclose(np.array(losses), fx["loss"][:12], atol=2e-4, rtol=0)
    assert fx["loss"][0] - fx["loss"][-1] > 0.1          # the curve really moves: 0.40 -> 0.24


def test_flag_gated_targets_and_variance_term(golden_dir):
    """`--layer_results fc` targets + the variance term `--var_w0 1 --var_margin0 1` (engine_for_cyclical.py:88-139),
    which no BASELINE config switches on: the oracle against the reference's own two steps."""
    fx = np.load(os.path.join(golden_dir, "model_flags.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1)
    p = closed_form_state(vo.param_shapes(cfg), gamma=0.1)
    hp = vo.StepHParams(target_layers=(1, 2), layer_results="fc", var_w0=1.0, var_margin0=1.0)
    ema = {k: t.clone() for k, t in p.items()}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    for s in range(steps):
        x = closed_form_images(f"flags/{s}", B, img)
        res = vo.train_step(p, ema, m, v, cfg, hp, x, torch.from_numpy(fx[f"mask{s}"]), s + 1)
        assert res.loss == pytest.approx(float(fx["loss"][s]), rel=2e-4)
        assert res.loss_var0 == pytest.approx(float(fx["loss_var0"][s]), rel=2e-4) and res.loss_var0 > 0.5
        assert res.grad_norm == pytest.approx(float(fx["grad_norm"][s]), rel=2e-3)
        if s == 0:
            for n in entries(fx, "grad0"):
                check_entry(fx, "grad0/" + n, res.grads[n], 2e-3, 2e-7)
    for n in entries(fx, "post"):
        check_entry(fx, "post/" + n, p[n], 1e-3, 2e-5)


def test_abs_pos_emb_case(golden_dir):
    """`--abs_pos_emb`: pos_embed (1, N, C) added after the cls concat, in the no-decay group (modeling_cyclical.py:80-84,
    163-165, 193-194): forward and two steps of the oracle against the reference."""
    fx = np.load(os.path.join(golden_dir, "model_abspos.npz"))
    img, dim, depth, heads, B, n_mask, steps = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1, use_abs_pos_emb=True)
    shapes = vo.param_shapes(cfg)
    assert [n for n in fx["names"].tolist() if not n.endswith("relative_position_index")] == list(shapes)     # state-dict order
    assert "pos_embed" in fx["groups/no_decay"].tolist()
    p = closed_form_state(shapes, gamma=0.1)
    x0, m0 = closed_form_images("abspos/0", B, img), torch.from_numpy(fx["mask0"])
    ends = vo.forward(p, cfg, x0, None, True, "end")
    for i in range(depth):
        check_entry(fx, f"fwd/end{i}", ends[i], RT, AT)
    check_entry(fx, "fwd/student_masked", vo.forward(p, cfg, x0, m0, False), RT, AT)
    hp = vo.StepHParams(target_layers=(1,))
    ema = {k: t.clone() for k, t in p.items()}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    for s in range(steps):
        res = vo.train_step(p, ema, m, v, cfg, hp, closed_form_images(f"abspos/{s}", B, img), torch.from_numpy(fx[f"mask{s}"]), s + 1)
        assert res.loss == pytest.approx(float(fx["loss"][s]), rel=2e-4)
        assert res.grad_norm == pytest.approx(float(fx["grad_norm"][s]), rel=2e-3)
        if s == 0:
            assert set(entries(fx, "grad0")) == set(res.grads) and "pos_embed" in res.grads
            for n in entries(fx, "grad0"):
                check_entry(fx, "grad0/" + n, res.grads[n], 2e-3, 2e-7)
    for n in entries(fx, "post"):
        check_entry(fx, "post/" + n, p[n], 1e-3, 2e-5)


def _tnorm_hp(case):
    fl = {"bn": dict(target_batch_norm=True, target_layer_norm_last=True, post_target_layer_norm=True),
          "in": dict(target_instance_norm=True, target_layer_norm_last=True, post_target_layer_norm=False),
          "bn_in_pin": dict(target_batch_norm=True, target_instance_norm=True, target_layer_norm_last=False,
                            post_target_instance_norm=True, post_target_layer_norm=True),
          "raw_pin": dict(target_layer_norm_last=False, post_target_instance_norm=True, post_target_layer_norm=False)}[case]
    return vo.StepHParams(target_layers=(1, 2), **fl)


@pytest.mark.parametrize("case", ["bn", "in", "bn_in_pin", "raw_pin"])
def test_batch_and_instance_norm_target_variants(golden_dir, case):
    """engine_for_cyclical.py:94-118 (batch norm over (B, T), instance norm over T, with / without the per-layer LayerNorm,
    post instance norm): one step of the oracle per flag combination against the reference."""
    fx = np.load(os.path.join(golden_dir, "target_norms.npz"))
    img, dim, depth, heads, B, n_mask, _ = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1)
    p = closed_form_state(vo.param_shapes(cfg), gamma=0.1)
    ema = {k: t.clone() for k, t in p.items()}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    res = vo.train_step(p, ema, m, v, cfg, _tnorm_hp(case), closed_form_images("tnorm", B, img), torch.from_numpy(fx["mask"]), 1)
    assert res.loss == pytest.approx(float(fx[f"{case}/loss"]), rel=2e-4)
    assert res.grad_norm == pytest.approx(float(fx[f"{case}/grad_norm"]), rel=2e-3)
    for n in entries(fx, f"{case}/grad"):
        check_entry(fx, f"{case}/grad/{n}", res.grads[n], 2e-3, 2e-7)


DIST_TNORM = {"bn": dict(target_batch_norm=True, target_layer_norm_last=True, post_target_layer_norm=True),
              "bn_in_pin": dict(target_batch_norm=True, target_instance_norm=True, target_layer_norm_last=False,
                                post_target_instance_norm=True, post_target_layer_norm=True)}


@pytest.mark.parametrize("case", list(DIST_TNORM))
def test_two_stream_step_with_target_norm_variants(golden_dir, case):
    """engine_for_cyclical.py:93-118 in a `stochastic=True` step: the batch- / instance-norm variants act on the MEAN targets, the
    covariance targets (:73-86) follow the two layer-norm flags only.  One oracle step per flag combination against the reference."""
    from oracle import vit_oracle_dist as vd
    fx = np.load(os.path.join(golden_dir, "dist_target_norms.npz"))
    img, dim, depth, heads, B, n_mask, _ = [int(v) for v in fx["cfg"]]
    cfg = vo.VitConfig(img_size=img, embed_dim=dim, depth=depth, num_heads=heads, init_values=0.1)
    p = closed_form_state(vd.param_shapes(cfg), gamma=0.1)
    ema = {k: t.clone() for k, t in p.items()}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    hp = vo.StepHParams(target_layers=(1, 2), **DIST_TNORM[case])
    res, _, _, _ = vd.train_step(p, ema, m, v, cfg, hp, closed_form_images("dtnorm", B, img), torch.from_numpy(fx["mask"]), 1, lam=1e-2)
    assert res.loss == pytest.approx(float(fx[f"{case}/loss"]), rel=2e-4)
    assert res.grad_norm == pytest.approx(float(fx[f"{case}/grad_norm"]), rel=2e-3)
    for n in entries(fx, f"{case}/grad"):
        check_entry(fx, f"{case}/grad/{n}", res.grads[n], 2e-3, 2e-7)
