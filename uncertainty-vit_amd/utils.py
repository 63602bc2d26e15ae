"""Host-side runtime services of the pre-training path, mirroring the names the reference's
`utils.py` exposes to run_cyclical.py / engine_for_cyclical.py: meters, schedules, distributed
init, checkpoints, EMA wrapper.  Host logic only (no kernels here)."""
import datetime
import glob
import math
import os
import time
from collections import OrderedDict, deque

import numpy as np
import torch
import torch.distributed as dist


# ---- meters: utils.py:34-177 ----
class SmoothedValue:
    """Windowed + global statistics of a scalar series (utils.py:34-93)."""

    def __init__(self, window_size=20, fmt=None):
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """One 2-element all-reduce per meter (utils.py:52-63); the window is not synchronised."""
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    median = property(lambda self: float(np.median(np.asarray(self.deque, dtype=np.float64))) if self.deque else 0.0)
    avg = property(lambda self: float(np.mean(np.asarray(self.deque, dtype=np.float32))) if self.deque else 0.0)
    global_avg = property(lambda self: self.total / max(self.count, 1))
    # empty until the first (one-step-late) metrics arrive: the meter line of iteration 0 then shows zeros
    max = property(lambda self: max(self.deque) if self.deque else 0.0)
    value = property(lambda self: self.deque[-1] if self.deque else 0.0)

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = OrderedDict()
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters.setdefault(k, SmoothedValue()).update(v)

    def __getattr__(self, attr):
        meters = self.__dict__.get("meters", {})
        if attr in meters:
            return meters[attr]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        n = len(iterable)
        it_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        start = end = time.time()
        width = len(str(n))
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            it_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = datetime.timedelta(seconds=int(it_time.global_avg * (n - i)))
                parts = [header, f"[{i:{width}d}/{n}]", f"eta: {eta}", str(self), f"time: {it_time}", f"data: {data_time}"]
                if torch.cuda.is_available():
                    parts.append(f"max mem: {torch.cuda.max_memory_allocated() / 2 ** 20:.0f}")
                print(self.delimiter.join(parts))
            end = time.time()
        total = time.time() - start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({total / max(n, 1):.4f} s / it)")


class TensorboardLogger:
    """utils.py:180-201; tensorboardX is optional and absent in this image."""

    def __init__(self, log_dir):
        from tensorboardX import SummaryWriter
        self.writer, self.step = SummaryWriter(logdir=log_dir), 0

    def set_step(self, step=None):
        self.step = step if step is not None else self.step + 1

    def update(self, head="scalar", step=None, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            v = v.item() if isinstance(v, torch.Tensor) else v
            self.writer.add_scalar(head + "/" + k, v, self.step if step is None else step)

    def flush(self):
        self.writer.flush()


# ---- distributed: utils.py:218-312 ----
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def setup_for_distributed(is_master):
    """Silence print on non-master ranks unless force=True (utils.py:218-230)."""
    import builtins
    builtin_print = builtins.print

    def print_(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)
    builtins.print = print_


RCCL_MAX_CHANNELS = "12"


def cap_rccl_channels():
    """Every RCCL channel is a workgroup that holds a CU for the length of a collective.  Several backward kernels are sized to
    fill the 256 CUs in ONE round (320-row dgrads: 237 workgroups, attention backward: 240, grouped wgrad: 216 items), and with
    CUs taken away they need a second, nearly empty round: a run with 240 of 256 CUs (ROC_GLOBAL_CU_MASK) is 18 % slower.
    The gradient buckets need 329 MB per ~10 ms of backward (31 GB/s algorithmic), far below what 12 channels move over xGMI (and 256 - 12 = 244 CUs still hold every one-round kernel),
    so the channel count is capped unless the user has set it.  Must run before the process group is created."""
    os.environ.setdefault("NCCL_MAX_NCHANNELS", RCCL_MAX_CHANNELS)
    return os.environ["NCCL_MAX_NCHANNELS"]


def init_distributed_mode(args):
    """One process per GPU from RANK / WORLD_SIZE / LOCAL_RANK (utils.py:262-312). On ROCm the
    'nccl' backend is RCCL over xGMI."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    else:
        print("Not using distributed mode")
        args.distributed = False
        return
    args.distributed = True
    use_gpu = torch.cuda.is_available() and str(getattr(args, "device", "cuda")).startswith("cuda")
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if use_gpu else "gloo"
    if use_gpu:
        cap_rccl_channels()
    print(f"| distributed init (rank {args.rank}): {args.dist_url}, gpu {args.gpu}", flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size, rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


# ---- schedules: utils.py:408-459 ----
def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    warmup_iters = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    print("Set warmup steps = %d" % warmup_iters)
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    n = epochs * niter_per_ep - warmup_iters
    i = np.arange(n)
    body = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * k / n)) for k in i])
    sched = np.concatenate((warm, body))
    assert len(sched) == epochs * niter_per_ep
    return sched


def tri_phase_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_perc=0.05, decay_perc=0.05, start_warmup_value=0):
    assert warmup_perc + decay_perc <= 1
    total = int(epochs * niter_per_ep)
    n_warm, n_decay = int(warmup_perc * total), int(decay_perc * total)
    n_hold = total - n_warm - n_decay
    print("Set warmup steps = %d" % n_warm)
    parts = [np.linspace(start_warmup_value, base_value, n_warm) if n_warm > 0 else np.array([]),
             np.full(n_hold, base_value) if n_hold > 0 else np.array([]),
             np.linspace(base_value, final_value, n_decay) if n_decay > 0 else np.array([])]
    sched = np.concatenate(parts)
    assert len(sched) == epochs * niter_per_ep, f"e: {epochs}, it: {niter_per_ep}, w: {n_warm}, h: {n_hold}, d: {n_decay}"
    return sched


# ---- EMA wrapper: timm.utils.ModelEmaV2 as used at run_cyclical.py:503 and engine_for_cyclical.py:183 ----
class ModelEmaV2(torch.nn.Module):
    def __init__(self, model, decay=0.9999, device=None):
        super().__init__()
        import copy
        self.module = copy.deepcopy(model)
        self.module.eval()
        self.decay = decay

    def _update(self, model, update_fn):
        """Generic (slow, per-tensor) path kept for API parity; the training engine uses the fused
        HIP kernel over the flat arena instead."""
        with torch.no_grad():
            for e, m in zip(self.module.state_dict().values(), model.state_dict().values()):
                if e.dtype.is_floating_point:
                    e.copy_(update_fn(e, m))
        if hasattr(self.module, "mark_weights_changed"):
            self.module.mark_weights_changed()

    def update(self, model):
        self._update(model, update_fn=lambda e, m: self.decay * e + (1.0 - self.decay) * m)

    def set(self, model):
        self._update(model, update_fn=lambda e, m: m)


def broadcast_model_state(model, model_ema=None, optimizer=None, src=0):
    """What DDP's constructor does at `run_cyclical.py:516` (broadcast of rank 0's parameters and buffers), for the flat arenas:
    one `dist.broadcast` of the parameter arena, one of the EMA teacher's, one each of the optimizer's moment arenas (a resumed
    run).  Called after the checkpoint load, so that every rank starts from rank 0's state whatever the init or resume path did.
    No-op without an initialised process group or with one rank.  Returns the number of broadcasts issued."""
    if not is_dist_avail_and_initialized() or get_world_size() < 2:
        return 0
    n = 0
    arenas = [getattr(model, "_arena", None)]
    if model_ema is not None:
        arenas.append(getattr(model_ema.module, "_arena", None))
    if optimizer is not None:
        arenas += [getattr(optimizer, "exp_avg", None), getattr(optimizer, "exp_avg_sq", None)]
    with torch.no_grad():
        for a in arenas:
            if a is not None:
                dist.broadcast(a, src=src)
                n += 1
    for m in (model, model_ema.module if model_ema is not None else None):
        if m is not None and hasattr(m, "mark_weights_changed"):
            m.mark_weights_changed()          # the bf16 shadows are rebuilt from the arena before the next step
    return n


def get_state_dict(model, unwrap_fn=None):
    return (model.module if hasattr(model, "module") and not isinstance(model, ModelEmaV2) else model).state_dict()


# ---- scaler: utils.py:364-390 ----
class NativeScalerWithGradNormCount:
    """bf16 GEMM inputs with fp32 accumulation and fp32 master weights need no loss scaling; the
    object exists because train_one_epoch receives it and reads state_dict()['scale']."""
    state_dict_key = "amp_scaler"

    def state_dict(self):
        return {"scale": 1.0}

    def load_state_dict(self, state_dict):
        pass


# ---- checkpoints: utils.py:462-545 (file format and key names kept) ----
def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    from pathlib import Path
    path = Path(args.output_dir) / f"checkpoint-{epoch}.pth"
    to_save = {"model": model_without_ddp.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch,
               "scaler": loss_scaler.state_dict() if loss_scaler is not None else {}, "args": args}
    if model_ema is not None:
        to_save["model_ema"] = get_state_dict(model_ema.module if hasattr(model_ema, "module") else model_ema)
    save_on_master(to_save, path)


def auto_load_model(args, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """Resume from the newest checkpoint-*.pth (utils.py:491-521).  As in the reference the saved
    teacher ('model_ema') is only restored when args.model_ema is set -- run_cyclical.py defines no
    such flag, so a resumed run restarts the teacher from the student's initial copy."""
    if getattr(args, "auto_resume", False) and len(getattr(args, "resume", "")) == 0 and args.output_dir:
        ckpts = glob.glob(os.path.join(args.output_dir, "checkpoint-*.pth"))
        latest = max((int(os.path.basename(c).split("-")[-1].split(".")[0]) for c in ckpts
                      if os.path.basename(c).split("-")[-1].split(".")[0].isdigit()), default=-1)
        if latest >= 0:
            args.resume = os.path.join(args.output_dir, f"checkpoint-{latest}.pth")
        print("Auto resume checkpoint: %s" % args.resume)
    if getattr(args, "resume", ""):
        if args.resume.startswith("https"):
            raise RuntimeError("no network: remote checkpoints are not supported")
        ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
        model_without_ddp.load_state_dict(ckpt["model"])
        print("Resume checkpoint %s" % args.resume)
        if "optimizer" in ckpt and "epoch" in ckpt:
            optimizer.load_state_dict(ckpt["optimizer"])
            args.start_epoch = ckpt["epoch"] + 1
            if getattr(args, "model_ema", False) and model_ema is not None and "model_ema" in ckpt:
                model_ema.module.load_state_dict(ckpt["model_ema"])
            if "scaler" in ckpt and loss_scaler is not None:
                loss_scaler.load_state_dict(ckpt["scaler"])
            print("With optim & sched!")
