#!/usr/bin/env python
"""Command line of the data2vec ViT pre-training path -- same flag names, types and defaults as the
reference's run_cyclical.py:36-284, same orchestration as run_cyclical.py:307-650:

    init_distributed -> seeds -> create_model -> ModelEmaV2 -> optimizer -> lr/wd tables
    -> auto-resume -> epochs of train_one_epoch -> checkpoints + log.txt

The step itself runs as HIP kernels (uncertainty-vit_amd/).  One process per GPU, launched with
torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK); gradients are all-reduced by the native
engine's reducer over RCCL, so the model is NOT wrapped in torch DDP.

New (not in the reference): `--data_set SYNTHETIC` (seeded N(0,1) images + exactly
`--num_mask_patches` masked patches per image; there is no dataset / network in this image) and
`--synthetic_len`.  Real-data pipelines (datasets.py, transforms.py) are out of scope.
"""
import argparse
import datetime
import json
import os
import time
from ast import literal_eval
from pathlib import Path

import numpy as np
import torch

from uncertainty_vit_amd import utils
from uncertainty_vit_amd.engine_for_cyclical import train_one_epoch
from uncertainty_vit_amd.modeling_cyclical import create_model
from uncertainty_vit_amd.optim_factory import create_optimizer
from uncertainty_vit_amd.utils import ModelEmaV2, NativeScalerWithGradNormCount as NativeScaler


def get_args(argv=None):
    p = argparse.ArgumentParser("BEiT pre-training script", add_help=False)
    a = p.add_argument
    a("--batch_size", default=64, type=int)
    a("--epochs", default=15, type=int)
    a("--save_ckpt_freq", default=10, type=int)
    # model
    a("--model", default="deit_base_patch16_224", type=str, metavar="MODEL")
    a("--rel_pos_bias", action="store_true")
    a("--disable_rel_pos_bias", action="store_false", dest="rel_pos_bias")
    p.set_defaults(rel_pos_bias=True)
    a("--abs_pos_emb", action="store_true")
    p.set_defaults(abs_pos_emb=False)
    a("--layer_scale_init_value", default=0.1, type=float)
    a("--num_mask_patches", default=75, type=int)
    a("--max_mask_patches_per_block", type=int, default=None)
    a("--min_mask_patches_per_block", type=int, default=16)
    a("--input_size", default=224, type=int)
    a("--drop_path", type=float, default=0.1, metavar="PCT")
    a("--drop", type=float, default=0.0, metavar="PCT")
    # optimizer
    a("--opt", default="adamw", type=str, metavar="OPTIMIZER")
    a("--opt_eps", default=1e-8, type=float, metavar="EPSILON")
    a("--opt_betas", default=None, type=float, nargs="+", metavar="BETA")
    a("--clip_grad", type=float, default=None, metavar="NORM")
    a("--momentum", type=float, default=0.9, metavar="M")
    a("--weight_decay", type=float, default=0.05)
    a("--weight_decay_end", type=float, default=None)
    a("--lr", type=float, default=5e-4, metavar="LR")
    a("--warmup_lr", type=float, default=1e-6, metavar="LR")
    a("--min_lr", type=float, default=1e-5, metavar="LR")
    a("--tri_phase_schedule", type=str, default=None)
    a("--warmup_epochs", type=int, default=5, metavar="N")
    a("--warmup_steps", type=int, default=-1, metavar="N")
    # augmentation (accepted for command-line compatibility; the data pipeline is out of scope)
    a("--color_jitter", type=float, default=0.4, metavar="PCT")
    a("--train_interpolation", type=str, default="bicubic")
    a("--aug_level", default=-1, type=int)
    a("--target_layers", type=str, default="[]")
    # dataset
    a("--data_path", default="/datasets01/imagenet_full_size/061417/", type=str)
    a("--data_set", default="IMNET", choices=["CIFAR100", "CIFAR10", "IMNET", "image_folder", "tiny_IMNET", "SYNTHETIC"], type=str)
    a("--synthetic_len", default=1024, type=int, help="images per epoch of the SYNTHETIC data set (new flag)")
    a("--synthetic_masks", default="uniform", choices=["uniform", "block"],
      help="SYNTHETIC masks: exactly num_mask_patches uniformly (default) or the block-wise generator (<= num_mask_patches) (new flag)")
    a("--imagenet_default_mean_and_std", default=False, action="store_true")
    a("--output_dir", default="")
    a("--log_dir", default=None)
    a("--device", default="cuda")
    a("--seed", default=0, type=int)
    a("--resume", default="")
    a("--auto_resume", action="store_true")
    a("--no_auto_resume", action="store_false", dest="auto_resume")
    p.set_defaults(auto_resume=True)
    a("--ema_decay_init", default=0.999, type=float)
    a("--ema_decay", default=0.9998, type=float)
    a("--ema_start_at", default=25000, type=int)
    a("--start_epoch", default=0, type=int, metavar="N")
    a("--num_workers", default=10, type=int)
    a("--pin_mem", action="store_true")
    a("--no_pin_mem", action="store_false", dest="pin_mem")
    p.set_defaults(pin_mem=True)
    # distributed
    a("--world_size", default=1, type=int)
    a("--local_rank", default=-1, type=int)
    a("--dist_on_itp", action="store_true")
    a("--dist_url", default="env://")
    a("--seed_model", default=None, type=str)
    a("--model_key", default="model|module", type=str)
    a("--model_prefix", default="", type=str)
    a("--l2_loss", default=False, action="store_true")
    a("--l1_beta", default=0.12, type=float)
    a("--layer_results", default="end", type=str)
    a("--var_w0", default=0.0, type=float)
    a("--var_w1", default=0.0, type=float)
    a("--var_margin0", default=0.5, type=float)
    a("--var_margin1", default=0.5, type=float)
    a("--skip_ema_during_lr_decay_for_tri", action="store_true")
    a("--loss_scale", default=-1, type=float)
    a("--ema_annealing_till_end", default=False, action="store_true")
    a("--attn_drop_rate", default=0.0, type=float)
    a("--mask_dropout_prob", default=-1.0, type=float)
    a("--no_target_layer_norm_last", default=False, action="store_true")
    a("--target_batch_norm", default=False, action="store_true")
    a("--target_instance_norm", default=False, action="store_true")
    a("--post_target_instance_norm", default=False, action="store_true")
    a("--post_target_layer_norm", default=False, action="store_true")
    a("--gp_layer", default=False, action="store_true")
    a("--gumbel_softmax", default=False, action="store_true")
    a("--sinkformer", action="store_true")
    a("--h_sto_trans", default=False, action="store_true")
    a("--stochastic", default=False, action="store_true")
    a("--lambda_pretraining", type=float, default=1e-5)
    return p.parse_args(argv)


class SyntheticPretrainSet(torch.utils.data.Dataset):
    """((image (3,S,S) f32 ~ N(0,1), mask (g,g) int64 with exactly n ones), 0) -- the loader contract of
    datasets.py:110-118 / engine_for_cyclical.py:45,58."""

    def __init__(self, n, size, window, num_mask, seed, block=None):
        self.n, self.size, self.window, self.num_mask, self.seed = n, size, window, num_mask, seed
        self.block = block          # (min_patches_per_block, max_patches_per_block) -> block-wise masks

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        img = torch.randn(3, self.size, self.size, generator=g)
        if self.block is not None:
            from uncertainty_vit_amd.masking_generator import MaskingGenerator
            gen = MaskingGenerator(tuple(self.window), self.num_mask, min_num_patches=self.block[0], max_num_patches=self.block[1],
                                   seed=self.seed * 1_000_003 + i)
            return (img, torch.from_numpy(gen())), 0
        P = self.window[0] * self.window[1]
        m = torch.zeros(P, dtype=torch.int64)
        m[torch.randperm(P, generator=g)[: self.num_mask]] = 1
        return (img, m.view(*self.window)), 0


def get_model(args):
    name = args.model
    if args.stochastic and name == "beit_base_patch16_224":
        # README.md:42-61 documents `--model beit_base_patch16_224 --stochastic`, which cannot run in the reference
        # (the base model returns no (mean, cov) pairs).  The only self-consistent reading is the
        # two-stream architecture registered as dist_beit_base_patch16_224 (modeling_cyclical.py:304-323).
        name = "dist_beit_base_patch16_224"
    if args.stochastic and name == "beit_large_patch16_224":
        name = "dist_beit_large_patch16_224"
    print(f"Creating model: {name}")
    return create_model(name, pretrained=False, drop_path_rate=args.drop_path, drop_rate=args.drop,
                        use_shared_rel_pos_bias=args.rel_pos_bias, use_abs_pos_emb=args.abs_pos_emb,
                        init_values=args.layer_scale_init_value, attn_drop_rate=args.attn_drop_rate,
                        gp_layer=args.gp_layer, gumbel_softmax=args.gumbel_softmax, sinkformer=args.sinkformer,
                        h_sto_trans=args.h_sto_trans)


def main(args):
    utils.init_distributed_mode(args)
    print(args)
    device = torch.device(args.device)
    seed = args.seed + utils.get_rank()
    torch.manual_seed(args.seed)      # identical initial weights on every rank (the reference relies on DDP's broadcast)
    np.random.seed(seed)

    model = get_model(args)
    patch_size = model.patch_embed.patch_size
    print("Patch size = %s" % str(patch_size))
    args.window_size = (args.input_size // patch_size[0], args.input_size // patch_size[1])
    args.patch_size = patch_size
    if args.seed_model:
        raise NotImplementedError("--seed_model (checkpoint surgery with rel-pos interpolation) is out of scope")
    if args.data_set != "SYNTHETIC":
        raise NotImplementedError("only --data_set SYNTHETIC is available here: the image pipelines of datasets.py are out of scope")

    block = (args.min_mask_patches_per_block, args.max_mask_patches_per_block) if args.synthetic_masks == "block" else None
    dataset_train = SyntheticPretrainSet(args.synthetic_len, args.input_size, args.window_size, args.num_mask_patches, args.seed, block)
    num_tasks, global_rank = utils.get_world_size(), utils.get_rank()
    num_training_steps_per_epoch = len(dataset_train) // args.batch_size // num_tasks
    sampler_train = torch.utils.data.DistributedSampler(dataset_train, num_replicas=num_tasks, rank=global_rank, shuffle=True)
    log_writer = None
    if global_rank == 0 and args.log_dir is not None:
        os.makedirs(args.log_dir, exist_ok=True)
        log_writer = utils.TensorboardLogger(log_dir=args.log_dir)
    data_loader_train = torch.utils.data.DataLoader(dataset_train, sampler=sampler_train, batch_size=args.batch_size,
                                                    num_workers=args.num_workers, pin_memory=args.pin_mem, drop_last=True)

    model.to(device)
    model_without_ddp = model
    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("number of params:", n_parameters)
    torch.manual_seed(seed)
    model_ema = ModelEmaV2(model, decay=args.ema_decay)
    print("Using EMA with decay = %.8f" % args.ema_decay)
    total_batch_size = args.batch_size * utils.get_world_size()
    print("LR = %.8f" % args.lr)
    print("Batch size = %d" % total_batch_size)
    print("Number of training steps = %d" % num_training_steps_per_epoch)
    print("Number of training examples per epoch = %d" % (total_batch_size * num_training_steps_per_epoch))

    optimizer = create_optimizer(args, model_without_ddp)
    loss_scaler = NativeScaler()
    start_lr_decay_at_step = -1
    if args.tri_phase_schedule is not None:
        warmup_phase, decay_phase = literal_eval(args.tri_phase_schedule)
        print("Use tri phase lr schedule!", warmup_phase, decay_phase)
        lr_schedule_values = utils.tri_phase_scheduler(args.lr, args.min_lr, args.epochs, num_training_steps_per_epoch,
                                                       warmup_perc=warmup_phase, decay_perc=decay_phase)
        if args.skip_ema_during_lr_decay_for_tri:
            start_lr_decay_at_step = (1 - decay_phase) * args.epochs * num_training_steps_per_epoch
            print("ema will be skipped after " + str(start_lr_decay_at_step) + " updates")
    else:
        print("Use step level LR & WD scheduler!")
        lr_schedule_values = utils.cosine_scheduler(args.lr, args.min_lr, args.epochs, num_training_steps_per_epoch,
                                                    warmup_epochs=args.warmup_epochs, warmup_steps=args.warmup_steps)
    if args.weight_decay_end is None:
        args.weight_decay_end = args.weight_decay
    wd_schedule_values = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs, num_training_steps_per_epoch)
    print("Max WD = %.7f, Min WD = %.7f" % (max(wd_schedule_values), min(wd_schedule_values)))

    utils.auto_load_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer,
                          loss_scaler=loss_scaler, model_ema=model_ema)
    # DDP's constructor broadcast (run_cyclical.py:516): every rank continues from rank 0's weights, EMA and optimizer moments
    utils.broadcast_model_state(model_without_ddp, model_ema, optimizer, src=0)
    target_layers = literal_eval(args.target_layers)
    assert len(target_layers) > 0
    print(f"target layers: {target_layers}")
    print(f"Start training for {args.epochs} epochs")
    if args.ema_annealing_till_end:
        args.ema_start_at = args.epochs * num_training_steps_per_epoch
        print("EMA annealing till the end activated")

    start_time = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        data_loader_train.sampler.set_epoch(epoch)
        if log_writer is not None:
            log_writer.set_step(epoch * num_training_steps_per_epoch)
        train_stats = train_one_epoch(
            model, model_ema, args.ema_start_at, args.ema_decay_init, args.ema_decay, target_layers, data_loader_train,
            optimizer, device, epoch, loss_scaler, args.clip_grad, l1_beta=args.l1_beta, log_writer=log_writer,
            start_steps=epoch * num_training_steps_per_epoch, lr_schedule_values=lr_schedule_values,
            wd_schedule_values=wd_schedule_values, l2_loss=args.l2_loss, layer_results=args.layer_results,
            var_w0=args.var_w0, var_w1=args.var_w1, var_margin0=args.var_margin0, var_margin1=args.var_margin1,
            start_lr_decay_at_step=start_lr_decay_at_step, loss_scale=args.loss_scale,
            mask_dropout_prob=args.mask_dropout_prob, target_layer_norm_last=not args.no_target_layer_norm_last,
            target_batch_norm=args.target_batch_norm, target_instance_norm=args.target_instance_norm,
            post_target_instance_norm=args.post_target_instance_norm, post_target_layer_norm=args.post_target_layer_norm,
            stochastic=args.stochastic, lambda_pretraining=args.lambda_pretraining)
        if args.output_dir and ((epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs):
            utils.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer,
                             loss_scaler=loss_scaler, epoch=epoch, model_ema=model_ema)
        log_stats = {**{f"train_{k}": v for k, v in train_stats.items()}, "epoch": epoch, "n_parameters": n_parameters}
        if args.output_dir and utils.is_main_process():
            if log_writer is not None:
                log_writer.flush()
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(log_stats) + "\n")
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))


if __name__ == "__main__":
    opts = get_args()
    if opts.output_dir:
        Path(opts.output_dir).mkdir(parents=True, exist_ok=True)
    main(opts)
