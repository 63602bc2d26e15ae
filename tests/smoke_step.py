"""smoke(): one tiny invocation of the hot path on cuda:0, checked against the oracle."""
import torch

from oracle import vit_oracle as vo
from oracle.closed_form import closed_form_images, exact_masks
from tests.gpu_util import native_model, native_steps, native_trainer


def run_smoke():
    cfg = vo.VitConfig(img_size=48, embed_dim=128, depth=2, num_heads=2, init_values=0.1)
    B = 3
    x = closed_form_images("smoke", B, 48)
    mask = exact_masks(B, 9, 4, 5)
    model, sd = native_model(cfg)
    ema, opt = native_trainer(model)
    st = native_steps(model, ema, opt, [(x.cuda(), mask.cuda())], [1])[0]
    p = {k: v.clone() for k, v in sd.items()}
    e = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    ref = vo.train_step(p, e, m, v, cfg, vo.StepHParams(target_layers=(1,)), x, mask, 1)
    assert abs(st["loss"] - ref.loss) < 5e-3 * abs(ref.loss) + 1e-4, (st["loss"], ref.loss)
    assert abs(st["grad_norm"] - ref.grad_norm) < 3e-2 * ref.grad_norm, (st["grad_norm"], ref.grad_norm)
    w = model.state_dict()["blocks.1.mlp.fc1.weight"].cpu()
    # the first AdamW step moves every weight by +-lr (2e-3); bf16 noise may flip the sign of a ~0 gradient
    diff = (w - p["blocks.1.mlp.fc1.weight"]).abs()
    assert diff.max() <= 2 * 2e-3 + 1e-4 and (diff > 5e-4).float().mean() < 0.02, (diff.max(), (diff > 5e-4).float().mean())
    print(f"smoke ok: loss {st['loss']:.5f} (oracle {ref.loss:.5f}), grad_norm {st['grad_norm']:.4f} (oracle {ref.grad_norm:.4f})")
