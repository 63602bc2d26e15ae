"""train_one_epoch with the reference's signature and meter names (engine_for_cyclical.py:24-227);
each iteration runs as hand-written HIP kernels through libuvit:

    teacher forward (EMA weights) -> targets -> student forward -> SmoothL1 -> backward
    -> [RCCL all-reduce of the gradient arena, bucketed per block, overlapped with backward]
    -> global-norm clip + AdamW -> EMA

Data parallelism: one process per GPU; gradients are averaged with torch.distributed
(`nccl` = RCCL over xGMI) on a side stream while the compute stream keeps running backward.
"""
import ctypes as C
import math
import sys
from typing import Iterable

import torch
import torch.distributed as dist

from . import utils
from .native import MAX_DEPTH, StepParams, check, cur_stream, lib, ptr


def _unwrap(model):
    return model.module if hasattr(model, "module") and not hasattr(model, "_arena") else model


class GradReducer:
    """Bucketed all-reduce(SUM) of the flat gradient arena, launched as soon as a block's
    gradients are final (the reference gets this from DDP hooks, run_cyclical.py:515-519)."""

    def __init__(self, model, enabled, comm_dtype=torch.float32):
        """comm_dtype=torch.bfloat16: each bucket is rounded to bf16 for the wire (164.5 MB instead of 329 MB per step
        for ViT-B), summed by RCCL in bf16 and widened back into the fp32 gradient arena; the clip norm, AdamW
        and its moments stay fp32.  Default fp32 = DDP's arithmetic (run_cyclical.py:515-519)."""
        self.enabled = enabled and utils.get_world_size() > 1
        self.comm_dtype = comm_dtype
        if not self.enabled:
            return
        self.world = utils.get_world_size()
        self.on_gpu = model._arena.is_cuda
        self.comm = torch.cuda.Stream() if self.on_gpu else None
        lay = {n: (o, k) for n, o, k, _, _ in model._layout}
        d = model.depth
        self.layer_ranges = []
        for i in range(d):
            lo = lay[f"blocks.{i}.attn.qkv.weight"][0]
            hi = sum(lay[f"blocks.{i}.mlp.fc2.weight"])
            self.layer_ranges.append((lo, hi))
        self.head_range = (lay["lm_head.weight"][0], model._n_decay)      # lm_head (+ cov_lm_head) weights end the decay region
        self.embed_range = (0, lay["blocks.0.attn.qkv.weight"][0])
        # every live no-decay tensor, one message (frozen tensors -- decay flag 2, laid out last -- have no gradient)
        frozen = [o for _, o, _, _, dk in model._layout if dk == 2]
        self.small_range = (model._n_decay, min(frozen) if frozen else model._arena.numel())
        self.pending = []

    def reduce(self, grads, rng, engine=None, layer=None):
        if not self.enabled or rng[1] <= rng[0]:
            return
        view = grads[rng[0]:rng[1]]
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm.wait_event(ev)
            if layer is not None:      # the block's wgrads run on the engine's second stream
                check(lib().uvit_step_wait_layer_grads(engine.h, layer, C.c_void_p(self.comm.cuda_stream)), "wait_layer_grads")
            with torch.cuda.stream(self.comm):
                self._all_reduce(view)
        else:
            self._all_reduce(view)

    def _all_reduce(self, view):
        if self.comm_dtype == torch.float32:
            self.pending.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True), None, None))
        else:
            wire = view.to(self.comm_dtype)
            self.pending.append((dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True), wire, view))

    def finish(self):
        if not self.enabled:
            return
        # Over RCCL, Work.wait() does not block the host: it makes the stream that is CURRENT at the call wait for the
        # collective.  The widening copy of a bf16 bucket is enqueued on the communication stream, so that is the stream that
        # has to wait (with the compute stream current, the copy could read `wire` before RCCL had written it; gloo's wait()
        # blocks the host and hides the difference).  The compute stream then joins the communication stream once, below.
        # `wire` was allocated on the communication stream and stays referenced until its copy has been enqueued there.
        for w, wire, view in self.pending:
            if self.on_gpu:
                with torch.cuda.stream(self.comm):
                    w.wait()
                    if wire is not None:
                        view.copy_(wire)
            else:
                w.wait()
                if wire is not None:
                    view.copy_(wire)
        self.pending = []
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm)


class DevicePrefetcher:
    """Keeps one batch ahead of the step in HBM: the host->device copy of batch i+1 runs on a side HIP stream (from
    pinned memory when the loader pins) while step i computes.  The reference copies inside the step loop on the
    compute stream (engine_for_cyclical.py:58-59); the tensors handed out are the same.  CPU devices pass through."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.on_gpu else None
        self.mask_rows = 0     # masked patches of the batch being handed out, counted on the host before its upload (0 = unknown)

    def __len__(self):
        return len(self.loader)

    def _upload(self, item):
        (samples, mask), label = item
        rows = int(mask.sum()) if (torch.is_tensor(mask) and not mask.is_cuda) else 0      # free on the host; a device mask would need a sync
        if not self.on_gpu:
            return ((samples.to(self.device).float().contiguous(), mask.to(self.device)), label), rows
        with torch.cuda.stream(self.stream):
            samples = samples.to(self.device, non_blocking=True).float().contiguous()
            mask = mask.to(self.device, non_blocking=True)
        return ((samples, mask), label), rows

    def __iter__(self):
        it = iter(self.loader)
        try:
            ahead = self._upload(next(it))
        except StopIteration:
            return
        while ahead is not None:
            cur, self.mask_rows = ahead
            if self.on_gpu:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                for t in cur[0]:
                    t.record_stream(torch.cuda.current_stream(self.device))
            try:
                ahead = self._upload(next(it))
            except StopIteration:
                ahead = None
            yield cur


def native_step(engine, reducer, samples, mask, hp):
    """One training iteration on the current stream; returns nothing (stats stay on the device)."""
    L, h, s = lib(), engine.h, cur_stream()
    hp_ref = C.byref(hp)
    if reducer is None or not reducer.enabled:
        check(L.uvit_train_step(h, ptr(samples), ptr(mask), hp_ref, s), "uvit_train_step")
        return
    g = engine.grads
    check(L.uvit_step_begin(h, ptr(samples), ptr(mask), hp_ref, s), "uvit_step_begin")
    reducer.reduce(g, reducer.head_range)
    for l in range(engine.model.depth - 1, -1, -1):
        check(L.uvit_step_backward_layer(h, l, hp_ref, s), "uvit_step_backward_layer")
        reducer.reduce(g, reducer.layer_ranges[l], engine, l)
    check(L.uvit_step_backward_embed(h, s), "uvit_step_backward_embed")
    reducer.reduce(g, reducer.embed_range)
    reducer.reduce(g, reducer.small_range)
    reducer.finish()
    check(L.uvit_step_update(h, hp_ref, s), "uvit_step_update")


def make_step_params(target_layers, optimizer, max_norm, l1_beta, l2_loss, loss_scale, target_layer_norm_last,
                     post_target_layer_norm, cur_decay, do_ema, world, seed, it, train_dropout=True, lambda_pretraining=1e-5,
                     depth=None, target_batch_norm=False, target_instance_norm=False, post_target_instance_norm=False,
                     n_rows_hint=0):
    """n_rows_hint: an upper bound on the batch's masked patches known on the HOST (0 = unknown): the base model's last block then runs
    its MLP on those rows only (include/uvit.h, uvit_step_params.n_rows_hint); the results are those of the all-rows step."""
    hp = StepParams()
    if len(target_layers) > MAX_DEPTH:
        raise ValueError(f"at most {MAX_DEPTH} target layers")
    for i, t in enumerate(target_layers):
        # `[targets[i] for i in target_layers]` (engine_for_cyclical.py:92) is Python list indexing: negative indices count
        # from the last block, anything outside [-depth, depth) raises IndexError, a repeated index is averaged twice
        t = int(t)
        if depth is not None:
            if not -depth <= t < depth:
                raise IndexError("list index out of range")
            t = t % depth
        hp.target_layers[i] = t
    hp.n_target_layers = len(target_layers)
    hp.target_layer_norm_last = int(bool(target_layer_norm_last))
    hp.post_target_layer_norm = int(bool(post_target_layer_norm))
    hp.target_batch_norm, hp.target_instance_norm = int(bool(target_batch_norm)), int(bool(target_instance_norm))
    hp.post_target_instance_norm = int(bool(post_target_instance_norm))
    hp.l2_loss = int(bool(l2_loss))
    hp.l1_beta, hp.loss_scale = float(l1_beta), float(loss_scale)
    hp.clip_grad = float(max_norm) if max_norm else 0.0
    g0 = optimizer.param_groups[0]
    hp.lr, hp.weight_decay = float(g0["lr"]), float(g0["weight_decay"])
    hp.beta1, hp.beta2, hp.eps = float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"])
    hp.opt_step = optimizer.step_count + 1
    hp.ema_decay, hp.do_ema = float(cur_decay), int(bool(do_ema))
    hp.grad_scale = 1.0 / world
    hp.seed, hp.it = int(seed) & 0xFFFFFFFF, int(it) & 0xFFFFFFFF
    hp.train_dropout = int(bool(train_dropout))
    hp.lambda_pretraining = float(lambda_pretraining)
    hp.n_rows_hint = max(int(n_rows_hint or 0), 0)
    return hp


def epoch_scalars(optimizer, start_steps, n_iters, lr_schedule_values, wd_schedule_values, ema_start_at, decay_init, decay,
                  start_lr_decay_at_step):
    """[(lr, weight_decay, ema_decay or -1 for "skip EMA")] for the n_iters iterations of one epoch, by the sequential rules of
    engine_for_cyclical.py:41,47-56,182-185 (cur_decay is STATE: it keeps its last annealed value once `it` passes
    ema_start_at, and stays 0 after the first skipped update).  Values are rounded to float32 as the C ABI carries them."""
    import ctypes
    f32 = lambda v: ctypes.c_float(float(v)).value  # noqa: E731
    g0 = optimizer.param_groups[0]
    lr, wd = float(g0["lr"]), float(g0["weight_decay"])
    cur_decay, out = decay, []
    for it in range(start_steps, start_steps + n_iters):
        if lr_schedule_values is not None:
            lr = lr_schedule_values[it] * g0.get("lr_scale", 1.0)
        if wd_schedule_values is not None and g0["weight_decay"] > 0:
            wd = wd_schedule_values[it]
        if it < ema_start_at:
            cur_decay = decay_init + it * (decay - decay_init) / ema_start_at
        do_ema = cur_decay != 1 and (start_lr_decay_at_step == -1 or it <= start_lr_decay_at_step)
        if not do_ema:
            cur_decay = 0
        out.append((f32(lr), f32(wd), f32(cur_decay) if do_ema else -1.0))
    return out


def train_one_epoch(model: torch.nn.Module, model_ema: torch.nn.Module, ema_start_at, decay_init, decay, target_layers,
                    data_loader: Iterable, optimizer, device: torch.device, epoch: int, loss_scaler,
                    max_norm: float = 0, l1_beta: float = 0.12, log_writer=None, lr_scheduler=None, start_steps=None,
                    lr_schedule_values=None, wd_schedule_values=None, l2_loss=False, layer_results='end',
                    var_w0=0, var_w1=0, var_margin0=0.5, var_margin1=0.5, start_lr_decay_at_step=-1, loss_scale=-1,
                    mask_dropout_prob=-1.0, target_layer_norm_last=True, target_batch_norm=False,
                    target_instance_norm=False, post_target_instance_norm=False, post_target_layer_norm=False,
                    stochastic=False, lambda_pretraining=1e-5):
    print(' <<<<<<<< layer_results >>>>>>>>', layer_results)
    print(' <<<<<<<< var_w0, var_w1 >>>>>>>>', var_w0, var_w1)
    # flags whose arithmetic is not on the configured hot path are refused, never approximated
    if layer_results not in ('end', 'fc'):
        raise NotImplementedError(f"--layer_results {layer_results}: the blocks return 'end' and 'fc' results (modeling_finetune.py:185-203)")
    dense_targets = target_batch_norm or target_instance_norm or post_target_instance_norm or not target_layer_norm_last
    model.train()
    net = _unwrap(model)
    teacher = model_ema.module
    if stochastic and layer_results != 'end':
        # the variance term (engine_for_cyclical.py:130-139,161) and, since round 4, the batch- / instance-norm target variants on the
        # MEAN targets (:93-118; the covariance targets :73-86 only know the two layer-norm flags) are native with the two-stream step
        raise NotImplementedError("the two-stream step is native for layer_results='end' (modeling_cyclical_dist.py:136-139 collects "
                                  "only 'end' results)")
    if bool(stochastic) != bool(getattr(net, "_two_stream", False)):
        # the reference unpacks (mean, cov) pairs when stochastic (engine_for_cyclical.py:70,126): only the two-stream
        # model (dist_beit_base_patch16_224) returns them -- SURVEY.md F8
        raise ValueError("stochastic=True needs the two-stream model (dist_beit_base_patch16_224) and vice versa")
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    metric_logger.add_meter('min_lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    metric_logger.add_meter('loss_var0', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    print_freq = 10
    world = utils.get_world_size()
    reducer = None
    engine = None
    ring = torch.zeros(4, 8, dtype=torch.float32)      # per slot: loss, grad_norm, -, -, std_loss0 (loss_var0 meter), ...
    events = None
    if torch.cuda.is_available():
        ring = ring.pin_memory()
        events = [torch.cuda.Event() for _ in range(ring.shape[0])]
    pending = []
    seed = torch.initial_seed()

    def publish(slot, entry):
        """Meters / TensorBoard of one finished iteration (engine_for_cyclical.py:188-219)."""
        if events is not None:
            events[slot].synchronize()
        loss_value, grad_norm = float(ring[slot, 0]), float(ring[slot, 1])
        if not (math.isfinite(loss_value) and math.isfinite(grad_norm)):
            # The reference stops before backward / optimizer.step / EMA (engine_for_cyclical.py:166-168); here the step was
            # already enqueued, so AdamW and EMA test the device-side loss and (all-reduced) gradient norm themselves and
            # leave the weights untouched (every later step then sees the same poisoned state and skips too).  The
            # all-reduced norm is NaN on EVERY rank when any rank's loss was, so all ranks take this exit together instead of
            # one leaving its peers blocked in the next collective.
            print("Loss is {}, stopping training".format(loss_value), force=True) if world > 1 else \
                print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        metric_logger.update(loss=loss_value)
        metric_logger.update(loss_scale=entry["loss_scale"])
        metric_logger.update(lr=entry["lr"])
        metric_logger.update(min_lr=entry["min_lr"])
        metric_logger.update(loss_var0=float(ring[slot, 4]) if var_w0 > 0 else 0)       # std_loss0 (engine_for_cyclical.py:198)
        metric_logger.update(weight_decay=entry["weight_decay"])
        metric_logger.update(grad_norm=grad_norm)
        metric_logger.update(cur_decay=entry["cur_decay"])
        if log_writer is not None:
            log_writer.update(loss=loss_value, head="loss")
            log_writer.update(loss_scale=entry["loss_scale"], head="opt")
            log_writer.update(lr=entry["lr"], head="opt")
            log_writer.update(min_lr=entry["min_lr"], head="opt")
            log_writer.update(weight_decay=entry["weight_decay"], head="opt")
            log_writer.update(grad_norm=grad_norm, head="opt")
            log_writer.update(cur_decay=entry["cur_decay"], head="cur_decay")
            log_writer.set_step()

    # Per-iteration scalars of the whole epoch, computed ONCE with the reference's own sequential rules and uploaded as a
    # device table {lr, weight_decay, ema_decay}[it]: the optimizer kernels index it on the device, so a step's launch
    # arguments no longer carry iteration-dependent values.  The same list feeds the meters.
    n_iters = len(data_loader) if hasattr(data_loader, "__len__") else None
    scalars = epoch_scalars(optimizer, start_steps, n_iters, lr_schedule_values, wd_schedule_values, ema_start_at, decay_init,
                            decay, start_lr_decay_at_step) if n_iters else None
    sched_dev = None

    cur_decay = decay
    prefetcher = DevicePrefetcher(data_loader, device)
    for step, (batch, _) in enumerate(metric_logger.log_every(prefetcher, print_freq, header)):
        it = start_steps + step  # global training iteration
        # per-step lr / weight-decay (engine_for_cyclical.py:47-53): the param groups keep showing the current values
        if lr_schedule_values is not None or wd_schedule_values is not None:
            for param_group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    param_group["lr"] = lr_schedule_values[it] * param_group["lr_scale"]
                if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                    param_group["weight_decay"] = wd_schedule_values[it]
        if it < ema_start_at:
            cur_decay = decay_init + it * (decay - decay_init) / ema_start_at

        samples, bool_masked_pos = batch
        if mask_dropout_prob > 0:
            keep = torch.bernoulli(torch.full_like(bool_masked_pos, 1 - mask_dropout_prob, dtype=samples.dtype))
            bool_masked_pos = torch.logical_and(keep, bool_masked_pos)
        mask = bool_masked_pos.reshape(samples.shape[0], -1).to(torch.int64).contiguous()

        if engine is None:
            optimizer._ensure_state()
            engine = net.engine(samples.shape[0], teacher=teacher, adam_m=optimizer.exp_avg, adam_v=optimizer.exp_avg_sq)
            reducer = GradReducer(net, world > 1)
        if samples.shape[0] != engine.batch:
            raise ValueError("the native step runs at the batch size the workspace was planned for (drop_last=True loader)")

        do_ema = cur_decay != 1 and (start_lr_decay_at_step == -1 or it <= start_lr_decay_at_step)
        if not do_ema:
            cur_decay = 0
        hp = make_step_params(target_layers, optimizer, max_norm, l1_beta, l2_loss, loss_scale, target_layer_norm_last,
                              post_target_layer_norm, cur_decay, do_ema, world, seed, it, lambda_pretraining=lambda_pretraining,
                              depth=net.depth, target_batch_norm=target_batch_norm, target_instance_norm=target_instance_norm,
                              post_target_instance_norm=post_target_instance_norm,
                              n_rows_hint=prefetcher.mask_rows)      # (mask dropout only removes rows: still an upper bound)
        if scalars is not None and step < len(scalars):
            assert scalars[step] == (hp.lr, hp.weight_decay, hp.ema_decay if do_ema else -1.0), (scalars[step], hp.lr, hp.weight_decay, hp.ema_decay)
            if sched_dev is None:
                sched_dev = torch.tensor([[sc[k] for sc in scalars] for k in range(3)], dtype=torch.float32, device=samples.device)
            hp.sched_dev, hp.sched_len, hp.sched_index = sched_dev.data_ptr(), len(scalars), step
        hp.layer_results_fc = int(layer_results == 'fc')
        hp.var_w0, hp.var_margin0 = float(var_w0), float(var_margin0)
        native_step(engine, reducer, samples, mask, hp)
        optimizer.step_count += 1

        # Metrics leave the device asynchronously: {loss, grad_norm} of this step are copied into a pinned ring slot behind the
        # step's kernels and read ONE STEP LATE, after the next step has been enqueued -- the host never drains the stream
        # inside the loop (the reference syncs twice per step: loss.item() and torch.cuda.synchronize(),
        # engine_for_cyclical.py:164,186).  The host-side scalars of the step travel with the slot so every meter still
        # receives the values of ONE iteration together.
        lrs = [g["lr"] for g in optimizer.param_groups]
        wds = [g["weight_decay"] for g in optimizer.param_groups if g["weight_decay"] > 0]
        entry = dict(lr=max(lrs), min_lr=min(lrs), weight_decay=wds[-1] if wds else None, cur_decay=cur_decay,
                     loss_scale=loss_scaler.state_dict()["scale"] if loss_scaler is not None else 1.0)
        slot = step % ring.shape[0]
        check(lib().uvit_engine_read_stats_async(engine.h, C.c_void_p(ring[slot].data_ptr()), cur_stream()), "read_stats_async")
        if events is not None:
            events[slot].record()
        pending.append((slot, entry))
        while len(pending) > 1:
            publish(*pending.pop(0))
        if lr_scheduler is not None:
            lr_scheduler.step_update(start_steps + step)

    while pending:
        publish(*pending.pop(0))
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
