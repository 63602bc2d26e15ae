#!/usr/bin/env python
"""bench.py -- pre-train images/sec of the data2vec ViT-B/16 step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N=1; N>1 launches its own ranks (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no RANK / WORLD_SIZE in the environment, bench.py starts N ranks itself (one process per GPU,
`torch.distributed.run` on 127.0.0.1) BEFORE anything touches the GPU and exits with their code; under an external
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.

A "step" is one full iteration of engine_for_cyclical.train_one_epoch on one synthetic batch that
is already resident in HBM: teacher forward -> targets -> student forward (attn-drop 0.05,
drop-path 0.25) -> SmoothL1 -> backward -> [gradient all-reduce] -> clip + AdamW -> EMA.
Rank 0 prints ONE JSON line.  `roofline` is for the largest kernel by time (the bf16 MFMA GEMM
gemm_nt256_kernel in its residual-epilogue instantiation: attention proj + MLP fc2; the fc1 launches ride along per
instantiation), timed with HIP events on the launch stream;
`cpu_baseline` times the oracle (the CPU restatement, kind "port") on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE = 140.698        # student fwd + bwd + teacher fwd, algorithmic 2MNK
GFLOP_BY_MODEL = {"beit_base_patch16_224": 140.698, "dist_beit_base_patch16_224": 281.396,
                  "beit_large_patch16_224": 492.876, "dist_beit_large_patch16_224": 985.752}
PEAK_BF16 = 2.5e15               # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM = 8.0e12                # B/s (MI355X_MICROARCH.md)
# committed rocprofv3 summaries of THIS round that the live numbers must agree with (tools/pmc_run.sh, tools/prof_step.sh)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "round4_pmc_hbm.txt")
KERNEL_STATS = os.path.join(ROOT, "profiles", "round4_step_kernel_stats_singlestream.txt")
# the four bracketed forward Linears of a block (uvit_engine_profile_read_kind): kind id, kernel instantiation, PMC tag
PROF_KINDS = {"fc1_teacher": (0, "gemm_nt256_kernel<EPI_GELU, MT 4, persistent> (fc1, bias + GELU)", "gemm_nt256_kernel<2, 4", None),
              "fc1_student": (1, "gemm_nt256_kernel<EPI_GELU_DG, MT 4, persistent> (fc1, bias + GELU, also stores GELU')", "gemm_nt256_kernel<8, 4", None),
              "proj": (2, "gemm_nt256_kernel<EPI_RESID, MT 5> (attention proj: bias, LayerScale, DropPath, fp32 residual)", "gemm_nt256_kernel<3, 5", "[proj]"),
              "fc2": (3, "gemm_nt256_kernel<EPI_RESID, MT 5> (MLP fc2: bias, LayerScale, DropPath, fp32 residual)", "gemm_nt256_kernel<3, 5", "[fc2]")}


def pmc_traffic(path=PMC_SUMMARY):
    """HBM-side bytes per launch of each bracketed Linear from the committed rocprofv3 PMC summary (separate FETCH_SIZE / WRITE_SIZE
    passes; counters in KiB; FETCH_SIZE reads half the bytes of a wide coalesced stream on gfx950 and is doubled: MI355X_MICROARCH.md,
    HBM).  -> {name: {"fetch": B, "write": B, "dram_read": B or None}}; empty when the summary is not there."""
    out = {}
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return out
    for name, (_, _, tag, sub) in PROF_KINDS.items():
        d = {}
        for line in lines:
            if tag not in line or "mean" not in line or (sub and sub not in line) or (not sub and "[" in line.split("mean")[0].split(tag)[1]):
                continue
            try:
                val = float(line.split("mean")[1].split(",")[0])
            except (ValueError, IndexError):
                continue
            head = line.split(":")[0].strip().lower()
            if head.startswith("fetch"):
                d["fetch"] = int(2 * val * 1024)
            elif head.startswith("write"):
                d["write"] = int(val * 1024)
            elif "rdreq_dram" in head:
                d["dram_read"] = int(val * 64)          # 64-B requests that reached HBM itself (not served by the Infinity Cache)
        if "fetch" in d and "write" in d:
            out[name] = d
    return out


def rocprof_avg_us(path=KERNEL_STATS):
    """avg us per launch by kernel-name prefix from the committed single-stream kernel-stats summary (tools/prof_summary.py)."""
    out = {}
    try:
        for line in open(path):
            parts = line.split()
            if len(parts) >= 5 and not line.startswith("#") and parts[-1].replace(".", "", 1).isdigit():
                out[line[:72].strip()] = float(parts[-2])
                out["calls:" + line[:72].strip()] = float(parts[-4])
    except (OSError, ValueError):
        pass
    return out


# the reference's OWN engine_for_cyclical.train_one_epoch timed in the build container (tools/time_reference.py; the
# reference cannot travel to the GPU box): bs=4, 8 threads, 3 timed steps -- quoted beside the port's figure
REFERENCE_ENGINE_BUILD_CONTAINER = {"img_per_s": 2.292, "s_per_step": 1.745, "threads": 8, "batch": 4, "timed_steps": 3,
                                    "source": "profiles/round2_reference_engine_cpu.json"}


def synthetic_batch(B, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, 224, 224, generator=g)
    m = torch.zeros(B, 196, dtype=torch.int64)
    for b in range(B):
        m[b, torch.randperm(196, generator=g)[:120]] = 1          # exactly 120 masked patches
    return x.to(device), m.view(B, 14, 14).to(device)


def host_threads():
    """Threads for the CPU baseline: the cores this process may use, capped at the GPU box's CPU share (16 per GPU) --
    torch's default there is all 128 SMT threads of the host, which oversubscribes the share and runs slower."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline(sample_bs=4, steps=3):
    """Oracle (oracle/vit_oracle.py) timed on the host cores: same step, fp32, bounded sample."""
    from oracle import vit_oracle as vo
    threads = host_threads()
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        cfg = vo.VitConfig(init_values=1e-4)
        p = vo.init_params(cfg, seed=0)
        e = {k: v.clone() for k, v in p.items()}
        m = {k: torch.zeros_like(v) for k, v in p.items()}
        v = {k: torch.zeros_like(t) for k, t in p.items()}
        x, mask = synthetic_batch(sample_bs, 0, "cpu")
        hp = vo.StepHParams()
        vo.train_step(p, e, m, v, cfg, hp, x, mask, 1)               # warm-up
        t0 = time.time()
        for s in range(steps):
            vo.train_step(p, e, m, v, cfg, hp, x, mask, s + 2)
        dt = (time.time() - t0) / steps
    finally:
        torch.set_num_threads(prev)
    return {"value": round(sample_bs / dt, 3), "unit": "img/s", "cores": threads, "kind": "port",
            "sample": f"{steps} timed steps (+1 warm-up) of the same ViT-B/16 step at bs={sample_bs}, fp32 eager PyTorch, "
                      f"torch.set_num_threads({threads}), dropout off",
            "reference_engine_build_container": REFERENCE_ENGINE_BUILD_CONTAINER}


def input_staging_rates(a, L, feed, step, fence, dev, x, mask, steps=8):
    """SURVEY 8f-1, measured beside (never instead of) `value`: the same step fed (a) from pinned host memory through the
    double-buffered prefetcher of engine_for_cyclical.DevicePrefetcher -- batch i+1 crosses PCIe on a side stream while step i
    computes -- and (b) by the on-device batch synthesis kernels (uvit_op_synth_batch), a fresh batch every step."""
    from uncertainty_vit_amd.engine_for_cyclical import DevicePrefetcher
    from uncertainty_vit_amd.native import check
    out = {}
    host = [((x.cpu().pin_memory(), mask.cpu().pin_memory()), 0), ((x.flip(0).cpu().pin_memory(), mask.flip(0).cpu().pin_memory()), 0)]
    batches = [host[i % 2] for i in range(steps + 2)]
    i0 = 10_000
    fence()
    t0 = None
    for i, ((xs, ms), _) in enumerate(DevicePrefetcher(batches, dev)):
        if i == 2:
            fence(); t0 = time.perf_counter()
        feed["x"], feed["mask"] = xs, ms.reshape(a.batch, -1).contiguous()
        step(i0 + i)
    fence()
    dt = time.perf_counter() - t0
    out["value_with_h2d"] = round(a.batch * steps / dt, 2)
    out["h2d"] = f"{steps} steps fed from pinned host memory, H2D of batch i+1 ({x.numel() * 4 / 1e6:.0f} MB) on a side stream beside step i"
    xs, ms = torch.empty_like(x), torch.empty_like(mask)
    feed["x"], feed["mask"] = xs, ms
    n_mask = int(mask[0].sum().item())
    fence(); t0 = time.perf_counter()
    for i in range(steps):
        check(L.uvit_op_synth_batch(C.c_void_p(xs.data_ptr()), C.c_void_p(ms.data_ptr()), a.batch, 3, x.shape[-1], ms.shape[1], n_mask,
                                    4321, i, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "synth_batch")
        step(i0 + 100 + i)
    fence()
    dt = time.perf_counter() - t0
    out["value_with_device_synth"] = round(a.batch * steps / dt, 2)
    out["device_synth"] = f"{steps} steps, a fresh N(0,1) batch + exactly-{n_mask}-ones masks generated on the GPU inside every step"
    feed["x"], feed["mask"] = x, mask
    return out


def self_launch(a):
    """--gpus N > 1 without a launcher: start N ranks (one process per GPU) before any GPU call and mirror their exit code."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()          # does not initialise the GPU on this image
    if n_dev < a.gpus:
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} visible GPUs, this machine has {n_dev}. "
              "Run on a node with enough GPUs, or use --gpus 1.", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = launch_command(a.gpus, port, sys.argv[1:])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def rccl_participation(dev):
    """What RCCL itself reports: the size of the process group and the sum of one `1` per rank through an all-reduce on the
    GPU -- proof that N ranks took part in a collective, not an echo of WORLD_SIZE."""
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "allreduce_ones": int(ones.item())}


def launch_command(n, port, argv):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (BASELINE config: 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-input-staging", action="store_true", help="skip the two extra short regions that time H2D-fed and device-synthesised batches")
    ap.add_argument("--single-stream", action="store_true", help="no second HIP stream (per-kernel profiling runs)")
    ap.add_argument("--all-rows", action="store_true", help="every row and every sample through every branch: no masked-row bound for the last block's MLP (uvit_step_params.n_rows_hint) "
                                                            "and no drop-path sample lists (uvit_engine_set_drop_path_rows) -- the full-size launches")
    ap.add_argument("--no-alone", action="store_true", help="skip the 3 extra single-stream steps that time the dominant kernel alone (profiling runs of the two-stream schedule)")
    ap.add_argument("--grad-comm-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="dtype of the gradient all-reduce buckets (bf16 halves the xGMI bytes; AdamW accumulates in fp32 either way)")
    ap.add_argument("--model", default="beit_base_patch16_224",
                    help="beit_base_patch16_224 (headline) | dist_beit_base_patch16_224 (--stochastic two-stream, BASELINE config 3) | "
                         "beit_large_patch16_224 | dist_beit_large_patch16_224 (config 5 model)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))               # nothing has touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}; start it as "
              f"`python bench.py --gpus {a.gpus}` or with --nproc-per-node {a.gpus}.", file=sys.stderr)
        sys.exit(2)
    if local >= torch.cuda.device_count():
        print(f"bench.py: LOCAL_RANK {local} has no GPU ({torch.cuda.device_count()} visible).", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_channels = None
    if world > 1:
        from uncertainty_vit_amd.utils import cap_rccl_channels
        rccl_channels = cap_rccl_channels()      # RCCL channel workgroups hold CUs the one-round backward kernels count on
        dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)     # nccl = RCCL on ROCm
    rccl = rccl_participation(dev) if world > 1 else None

    from uncertainty_vit_amd import optim_factory, utils
    from uncertainty_vit_amd.engine_for_cyclical import GradReducer, make_step_params, native_step
    from uncertainty_vit_amd.modeling_cyclical import create_model
    from uncertainty_vit_amd.native import check, cur_stream, lib

    torch.manual_seed(0)           # identical initial weights on every rank (DDP broadcasts rank 0's)
    stochastic = a.model.startswith("dist_")
    model = create_model(a.model, pretrained=False, drop_path_rate=0.25, drop_rate=0.0,
                         use_shared_rel_pos_bias=True, use_abs_pos_emb=False, init_values=1e-4, attn_drop_rate=0.05,
                         gp_layer=False, gumbel_softmax=False, sinkformer=False, h_sto_trans=False).to(dev)
    model.train()
    ema = utils.ModelEmaV2(model, decay=0.9998)
    if world > 1:
        utils.broadcast_model_state(model, ema, src=0)     # DDP's constructor broadcast (run_cyclical.py:516)

    class A:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 2e-3, 0.05, 1e-8, (0.9, 0.999)
    import builtins
    _p = builtins.print
    builtins.print = lambda *x, **k: None           # keep stdout to the single JSON line
    opt = optim_factory.create_optimizer(A(), model)
    builtins.print = _p
    opt._ensure_state()
    engine = model.engine(a.batch, teacher=ema.module, adam_m=opt.exp_avg, adam_v=opt.exp_avg_sq)
    if a.all_rows:
        engine.set_drop_path_rows(False)
    reducer = GradReducer(model, world > 1, comm_dtype=torch.bfloat16 if a.grad_comm_dtype == "bf16" else torch.float32)
    x, mask = synthetic_batch(a.batch, 1000 + rank, dev)
    mask = mask.reshape(a.batch, -1).contiguous()
    L = lib()
    if os.environ.get("UVIT_TN_TARGET") or os.environ.get("UVIT_GEMM_VARIANT") or os.environ.get("UVIT_WG_CHUNKS"):     # tuning experiments only
        from uncertainty_vit_amd.native import Tuning
        tu = Tuning.default()
        tu.tn_split_target = int(os.environ.get("UVIT_TN_TARGET", tu.tn_split_target))
        tu.nt_variant = int(os.environ.get("UVIT_GEMM_VARIANT", tu.nt_variant))
        tu.wgrad_group_chunks = int(os.environ.get("UVIT_WG_CHUNKS", tu.wgrad_group_chunks))
        check(L.uvit_engine_set_tuning(engine.h, C.byref(tu)), "set_tuning")
    seed = 1000 + rank             # run_cyclical.py:315 seeds with seed + rank: every rank draws its own dropout / drop-path masks

    # masked patches of the batch, counted once on the host (every batch the staging regions feed has the same count): the step's row bound
    feed = {"x": x, "mask": mask, "mask_rows": 0 if a.all_rows else int(mask.sum().item())}      # what a step consumes (swapped by the input-staging regions below)

    def step(i):
        depth = model.depth
        hp = make_step_params(list(range(depth // 2, depth)), opt, 3.0, 2.0, False, -1, True, True, 0.9998, True, world, seed, i,
                              lambda_pretraining=1e-5, depth=depth, n_rows_hint=feed["mask_rows"])
        hp.lr = 2e-5       # warm-up-sized lr: random-init weights and fixed synthetic data, 2e-3 is the post-warm-up peak
        native_step(engine, reducer, feed["x"], feed["mask"], hp)
        opt.step_count += 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.single_stream:
        check(L.uvit_engine_set_streams(engine.h, 0), "set_streams")
    for i in range(a.warmup):
        step(i)
    fence()
    # timed region: K steps.  The dominant kernel's launches are bracketed with HIP events on their own stream;
    # in the default two-stream schedule other kernels share the GPU with it, so that duration is "overlapped".
    check(L.uvit_engine_profile(engine.h, 1, 4096), "profile on")
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    check(L.uvit_engine_profile(engine.h, 0, 0), "profile off")
    def read_kinds():
        """{name: (avg ms per launch, launches, flops per launch, algorithmic bytes per launch)} of the bracketed Linears"""
        res = {}
        for name, (kind, _, _, _) in PROF_KINDS.items():
            tot, n, fl, by = C.c_double(), C.c_int(), C.c_double(), C.c_double()
            check(L.uvit_engine_profile_read_kind(engine.h, kind, C.byref(tot), C.byref(n), C.byref(fl), C.byref(by)), "profile read")
            if n.value:
                res[name] = (tot.value / n.value, n.value, fl.value, by.value)
        return res
    sched = read_kinds()           # inside the timed region (two-stream schedule unless --single-stream)
    alone = None
    if not a.single_stream and not a.no_alone:
        # the same kernels with the GPU to themselves: a few extra single-stream steps right after the timed region
        check(L.uvit_engine_set_streams(engine.h, 0), "set_streams")
        check(L.uvit_engine_profile(engine.h, 1, 4096), "profile on")
        for i in range(3):
            step(a.warmup + a.steps + i)
        fence()
        check(L.uvit_engine_profile(engine.h, 0, 0), "profile off")
        alone = read_kinds()
        check(L.uvit_engine_set_streams(engine.h, engine.stream_mode), "set_streams")
    # the step without the masked-row bound (every row through the last block's MLP): img/s beside `value`, same run
    all_rows_value = None
    if world == 1 and feed["mask_rows"] and not a.no_alone and not a.single_stream:      # (two-stream model: the row bound is ignored, the sample lists are not)      # (profiling runs pass --single-stream / --no-alone: timed steps only)
        keep = feed["mask_rows"]
        feed["mask_rows"] = 0
        keep_dp = engine.drop_path_rows
        engine.set_drop_path_rows(False)                 # ... and every Block branch on every sample, dropped ones multiplied by 0
        for i in range(2):
            step(a.warmup + a.steps + 10 + i)
        fence(); t1 = time.perf_counter()
        for i in range(a.steps):
            step(a.warmup + a.steps + 12 + i)
        fence()
        all_rows_value = round(a.batch * a.steps / (time.perf_counter() - t1), 2)
        feed["mask_rows"] = keep
        engine.set_drop_path_rows(keep_dp)
    stats = torch.zeros(2).pin_memory()
    check(L.uvit_engine_read_stats(engine.h, C.c_void_p(stats.data_ptr()), cur_stream()), "stats")
    staging = None
    if world == 1 and not a.no_input_staging:
        staging = input_staging_rates(a, L, feed, step, fence, dev, x, mask)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = a.batch * world * a.steps / dt
        gflop = GFLOP_BY_MODEL.get(a.model, GFLOP_PER_IMAGE)
        # FLOPs the step really executes: with the masked-row bound the last block's MLP (teacher forward; student forward, two
        # dgrads, two wgrads = 8 GEMMs of 2 rows C Hd) skips the rows that feed neither the loss nor the targets
        M_rows, Cd, Hdd = a.batch * 197, model.embed_dim, 4 * model.embed_dim
        skipped = 16.0 * max(M_rows - feed["mask_rows"], 0) * Cd * Hdd / a.batch / 1e9 if (feed["mask_rows"] and not stochastic) else 0.0
        # drop-path sample lists (base model): a student branch skips the samples its DropPath dropped -- in expectation rate_l of them in
        # layer l (rates linspace(0, 0.25, depth)); per token the attention branch is 8 C^2 (QKV, proj) + 4 N C (core), the MLP 16 C^2, and
        # the student runs them forward + backward (x3; the attention core's backward is 2.5 forwards).  The masked-row last block keeps
        # the dense attention branch and its own row list.
        dp_lists = bool(engine.drop_path_rows)
        skipped_dp = 0.0
        if dp_lists:
            depth_ = model.depth
            for l_ in range(depth_):
                r_ = 0.25 * l_ / (depth_ - 1)
                last_masked = bool(feed["mask_rows"]) and not stochastic and l_ == depth_ - 1      # that block's MLP keeps its masked-row list instead
                if stochastic:                                              # two-stream: the MLP branches only, per stream
                    skipped_dp += 2 * r_ * 197 * (3 * 16 * Cd * Cd) / 1e9
                else:
                    skipped_dp += r_ * 197 * (3 * 8 * Cd * Cd + 3.5 * 4 * 197 * Cd + (0 if last_masked else 3 * 16 * Cd * Cd)) / 1e9
        skipped += skipped_dp
        traffic = pmc_traffic()
        rp = rocprof_avg_us()

        # algorithmic bytes of the FULL-SIZE launch of each kind (all rows, all samples): the shape the PMC passes measured (tools/pmc_run.sh runs
        # --all-rows with UVIT_DP_ROWS=0); the event brackets cover every launch, compact ones included, at their mean row count
        Mf = float(M_rows * (2 if stochastic else 1))
        full_bytes = {"fc1_teacher": 2.0 * (Mf * Cd + Hdd * Cd + Mf * Hdd), "fc1_student": 2.0 * (Mf * Cd + Hdd * Cd + 2 * Mf * Hdd),
                      "proj": 2.0 * (M_rows * Cd + Cd * Cd) + 8.0 * M_rows * Cd, "fc2": 2.0 * (M_rows * Hdd + Cd * Hdd) + 8.0 * M_rows * Cd}

        def entry(name, src):
            ms_, n_, fl_, by_ = src[name]
            d = {"avg_launch_ms": round(ms_, 4), "launches": n_, "achieved_tflops": round(fl_ / (ms_ * 1e-3) / 1e12, 1),
                 "frac_mfma": round(fl_ / (ms_ * 1e-3) / PEAK_BF16, 4), "flops_per_launch": fl_, "algorithmic_bytes": int(by_),
                 "frac_hbm_algorithmic": round(by_ / (ms_ * 1e-3) / PEAK_HBM, 4)}
            t = traffic.get(name)
            if t and name in full_bytes:
                d["traffic"] = t["fetch"] + t["write"]
                d["traffic_shape"] = "full-size launch (all rows, all samples)"
                d["full_size_algorithmic_bytes"] = int(full_bytes[name])
                d["traffic_over_algorithmic"] = round((t["fetch"] + t["write"]) / full_bytes[name], 3)
                d["traffic_detail"] = t
            return d
        best = alone if alone else sched              # single-stream numbers are the ones rocprofv3's summary can reproduce
        fam = [k for k in ("proj", "fc2") if k in best]            # largest kernel by time: the residual-epilogue instantiation
        fam_ms = sum(best[k][0] * best[k][1] for k in fam)
        fam_fl = sum(best[k][2] * best[k][1] for k in fam)
        fam_by = sum(best[k][3] * best[k][1] for k in fam)
        fam_n = sum(best[k][1] for k in fam)
        achieved = fam_fl / (fam_ms * 1e-3) / 1e12 if fam_ms else 0.0
        sch_ms = sum(sched[k][0] * sched[k][1] for k in fam if k in sched)
        sch_fl = sum(sched[k][2] * sched[k][1] for k in fam if k in sched)
        fam_traffic = (sum((traffic[k]["fetch"] + traffic[k]["write"]) * best[k][1] for k in fam) // max(fam_n, 1)) if all(k in traffic for k in fam) and fam else None
        # the residual-epilogue Linears run as the 320-row tile (<3, 5, false>) and, for some compact row counts, the 256-row one (<3, 4, false>)
        rp_keys = [k for k in rp if k.startswith("void gemm_nt256_kernel<3, 5, false>") or k.startswith("void gemm_nt256_kernel<3, 4, false>")]
        rp_calls = sum(rp["calls:" + k] for k in rp_keys)
        rp_avg = round(sum(rp[k] * rp["calls:" + k] for k in rp_keys) / rp_calls, 1) if rp_calls else None
        out = {
            "metric": "pretrain images/sec (ViT-B/16 224, bs=128/GPU)", "value": round(value, 2), "unit": "img/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{a.model} data2vec pretrain step{' (--stochastic: two-stream + Wasserstein loss)' if stochastic else ''}, "
                                   f"bs={a.batch}/GPU, 224x224 synthetic, target_layers=upper half, attn_drop 0.05, drop_path 0.25, clip 3.0, "
                                   "AdamW, EMA 0.9998",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}",
                       "model": a.model, "step_mfma_frac": round(value / world * gflop * 1e9 / PEAK_BF16, 4),
                       "executed_flops_frac": round(value / world * (gflop - skipped) * 1e9 / PEAK_BF16, 4),
                       "executed_gflop_per_image": round(gflop - skipped, 3),
                       "value_all_rows": all_rows_value,
                       "final_loss": round(float(stats[0]), 5),
                       "last_block_mlp_rows": (f"{feed['mask_rows']} masked rows of {a.batch * 197} (host-side bound; same results as all rows)"
                                               if feed["mask_rows"] and not stochastic else "all"),
                       "drop_path_rows": ("each Block branch of the student runs on the samples its DropPath kept (compact rows sized on the host from the "
                                          f"device's counter-based hash; same results as all samples); expected skip {skipped_dp:.2f} GFLOP/image"
                                          if dp_lists else "all samples"),
                       "step_mfma_frac_note": "step_mfma_frac = algorithmic FLOPs of the reference's step (SURVEY 8d) / time / peak; executed_flops_frac "
                                              "counts only the FLOPs this step runs (the last block's MLP skips unmasked rows, every student branch the samples its DropPath "
                                              "dropped); value_all_rows = img/s with every row and every sample through every branch, same process"},
            "roofline": {"bound": "mfma",
                         "kernel": "gemm_nt256_kernel<EPI_RESID, 320x256 tile> -- the largest kernel by time of the single-stream rocprofv3 summary: "
                                   "the Linears with the LayerScale x DropPath x fp32-residual epilogue (attention proj, K=768, and MLP fc2, K=3072; "
                                   f"M<={M_rows} N={Cd}), {fam_n} timed launches; per-shape figures and the fc1 launches under by_instantiation",
                         "achieved": round(achieved, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / PEAK_BF16, 4),
                         "avg_launch_ms": round(fam_ms / max(fam_n, 1), 4), "launches_timed": fam_n,
                         "algorithmic_bytes": int(fam_by / max(fam_n, 1)), "traffic": fam_traffic,
                         "traffic_shape": "full-size launches (all rows, all samples); achieved / avg_launch_ms / algorithmic_bytes are over every launch "
                                          "of the family, the compact ones (drop-path sample lists, masked-row last block) included",
                         "traffic_source": "profiles/round4_pmc_hbm.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a single-stream run, tools/pmc_run.sh; "
                                           "2 x FETCH_SIZE + WRITE_SIZE per launch, per shape; FETCH_SIZE counts Infinity-Cache hits, dram_read "
                                           "(TCC_EA0_RDREQ_DRAM, where collected) is the part that reached HBM)",
                         "measured": ("HIP events on the launch stream over the timed region, single stream (the kernel has the GPU to itself)" if a.single_stream or not alone else
                                      "HIP events on the launch stream over 3 single-stream steps right after the timed region (the kernel has the GPU to "
                                      "itself: what rocprofv3's per-kernel average of a single-stream run reproduces); in_schedule = the same launches "
                                      "inside the timed region, where the second stream's kernels share the CUs"),
                         "rocprof_check": {"file": "profiles/round4_step_kernel_stats_singlestream.txt", "avg_launch_us": rp_avg,
                                           "kernels": [k[:44] for k in rp_keys]},
                         "in_schedule": None if not sch_ms else {"avg_launch_ms": round(sch_ms / sum(sched[k][1] for k in fam if k in sched), 4),
                                                                 "achieved": round(sch_fl / (sch_ms * 1e-3) / 1e12, 2),
                                                                 "frac": round(sch_fl / (sch_ms * 1e-3) / PEAK_BF16, 4)},
                         "by_instantiation": {k: entry(k, best) for k in PROF_KINDS if k in best}},
            "rccl_ranks": rccl["allreduce_ones"] if rccl else 1,      # sum of one `1` per rank through an RCCL all-reduce
            "rccl": rccl,
            "rccl_max_channels": rccl_channels,
            "input_staging": staging,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
