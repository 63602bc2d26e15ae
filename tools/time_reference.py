"""Time the UNMODIFIED reference engine (engine_for_cyclical.train_one_epoch, imported from /root/reference through
tools/ref_harness.py) on this container's host cores: the "reference's own CPU path timed beside" the GPU number
(BASELINE.json north_star).  Build-container only -- the reference never travels to the GPU box.

    python tools/time_reference.py [--threads 8] [--batch 4] [--steps 3] [--model base|dist]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ref_harness  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--model", default="base", choices=["base", "dist"])
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    mc, eng = ref_harness.import_reference()
    import optim_factory
    import timm.models as M
    import timm.utils as U
    name = "beit_base_patch16_224" if a.model == "base" else "dist_beit_base_patch16_224"
    torch.manual_seed(0)
    model = M.create_model(name, pretrained=False, drop_path_rate=0.25, drop_rate=0.0, use_shared_rel_pos_bias=True,
                           use_abs_pos_emb=False, init_values=1e-4, attn_drop_rate=0.05)
    ema = U.ModelEmaV2(model, decay=0.9998)
    opt = optim_factory.create_optimizer(SimpleNamespace(opt="adamw", lr=2e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=(0.9, 0.999),
                                                         momentum=0.9), model)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(a.batch, 3, 224, 224, generator=g)
    m = torch.zeros(a.batch, 196, dtype=torch.int64)
    for b in range(a.batch):
        m[b, torch.randperm(196, generator=g)[:120]] = 1
    loader = [((x, m.view(a.batch, 14, 14)), torch.zeros(1))]
    scaler = ref_harness.HarnessScaler()

    def step(s):
        eng.train_one_epoch(model, ema, 0, 0.9998, 0.9998, list(range(6, 12)), loader, opt, torch.device("cpu"), 0, scaler,
                            max_norm=3.0, l1_beta=2.0, start_steps=s, layer_results="end", loss_scale=-1,
                            target_layer_norm_last=True, post_target_layer_norm=True, stochastic=a.model == "dist",
                            lambda_pretraining=1e-5)
    step(0)                                   # warm-up
    t0 = time.time()
    for s in range(a.steps):
        step(1 + s)
    dt = (time.time() - t0) / a.steps
    print(json.dumps({"what": "reference engine_for_cyclical.train_one_epoch on host cores (build container)", "model": name,
                      "batch": a.batch, "threads": a.threads, "timed_steps": a.steps, "s_per_step": round(dt, 3),
                      "img_per_s": round(a.batch / dt, 3)}))


if __name__ == "__main__":
    main()
