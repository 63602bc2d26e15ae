"""Worker for tests/test_gpu_ddp.py: one data-parallel rank of the native step.
UVIT_BACKEND = gloo (default: ranks share GPU 0) | nccl (RCCL: rank r on GPU r).
UVIT_DROPOUT = 1: attn-drop 0.1 / drop-path 0.3 with EVERY rank on the SAME images, seeded seed + rank as
run_cyclical.py:315 does (UVIT_SAME_SEED = 1: the same seed on every rank)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = os.environ["UVIT_OUT"]
    backend = os.environ.get("UVIT_BACKEND", "gloo")
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)
    dropout = os.environ.get("UVIT_DROPOUT") == "1"
    torch.manual_seed(123 + (0 if os.environ.get("UVIT_SAME_SEED") == "1" else rank))        # run_cyclical.py:315
    from oracle import vit_oracle as vo
    from oracle.closed_form import closed_form_images, exact_masks
    from gpu_util import native_model, native_steps, native_trainer
    cfg = vo.VitConfig(img_size=48, embed_dim=128, depth=3, num_heads=2, init_values=0.1,
                       drop_path_rate=0.3 if dropout else 0.0, attn_drop_rate=0.1 if dropout else 0.0)
    B = 8
    x = closed_form_images("ddp", B, 48)
    mask = exact_masks(B, 9, 4, 77)
    per = B // world
    if dropout:
        lo, per = 0, 4                      # every rank (and the 1-rank run) sees the same four images
    else:
        lo = rank * per
    xs, ms = x[lo:lo + per].cuda(), mask[lo:lo + per].cuda()
    model, _ = native_model(cfg)
    ema, opt = native_trainer(model)
    stats = native_steps(model, ema, opt, [(xs, ms)] * 2, [1, 2])
    torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()},
                "ema": {k: v.cpu() for k, v in ema.module.state_dict().items()},
                "loss": [s["loss"] for s in stats], "gnorm": [s["grad_norm"] for s in stats]}, f"{out}.rank{rank}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
