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
Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the bf16 MFMA GEMM
gemm_nt256_kernel, timed on its fc1 bias+GELU launches with HIP events on the launch stream);
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
# fc1 GEMM (M=25216, N=3072, K=768): algorithmic HBM bytes per launch = A + W read, h + gelu(h) written (bf16)
ALGO_BYTES = 2 * (25216 * 768 + 3072 * 768 + 2 * 25216 * 3072)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "round3_pmc_hbm.txt")      # written by tools/pmc_run.sh (two separate --pmc passes)


def pmc_traffic_bytes(path=PMC_SUMMARY):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary: mean over its two instantiations (teacher
    launch: GELU; student launch: GELU + GELU') of 2 x FETCH_SIZE + WRITE_SIZE, counters in KiB (FETCH_SIZE reads half the bytes of a wide
    coalesced stream on gfx950: MI355X_MICROARCH.md, HBM).  None when the summary is not there."""
    try:
        vals = {}
        for line in open(path):
            for tag in ("gemm_nt256_kernel<2, 4", "gemm_nt256_kernel<8, 4"):
                if tag in line and "mean" in line:
                    kind = "fetch" if line.startswith("fetch") else "write"
                    vals[(tag, kind)] = float(line.split("mean")[1].split(",")[0])
        per = [2 * vals[(t, "fetch")] + vals[(t, "write")] for t in ("gemm_nt256_kernel<2, 4", "gemm_nt256_kernel<8, 4")]
        return int(sum(per) / len(per) * 1024)
    except (OSError, KeyError, ValueError, IndexError):
        return None
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
    ap.add_argument("--all-rows", action="store_true", help="no masked-row bound for the step: the last block's MLP runs on every token row (A/B of uvit_step_params.n_rows_hint)")
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
    reducer = GradReducer(model, world > 1, comm_dtype=torch.bfloat16 if a.grad_comm_dtype == "bf16" else torch.float32)
    x, mask = synthetic_batch(a.batch, 1000 + rank, dev)
    mask = mask.reshape(a.batch, -1).contiguous()
    L = lib()
    if os.environ.get("UVIT_TN_TARGET") or os.environ.get("UVIT_GEMM_VARIANT"):     # tuning experiments only
        from uncertainty_vit_amd.native import Tuning
        tu = Tuning.default()
        tu.tn_split_target = int(os.environ.get("UVIT_TN_TARGET", tu.tn_split_target))
        tu.nt_variant = int(os.environ.get("UVIT_GEMM_VARIANT", tu.nt_variant))
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
    tot, n, fl = C.c_double(), C.c_int(), C.c_double()
    check(L.uvit_engine_profile_read(engine.h, C.byref(tot), C.byref(n), C.byref(fl)), "profile read")
    in_ms, in_n, flops = tot.value / max(n.value, 1), n.value, fl.value
    alone_ms = None
    if not a.single_stream and not a.no_alone:
        # the same kernel alone on the GPU: a few extra single-stream steps right after the timed region
        check(L.uvit_engine_set_streams(engine.h, 0), "set_streams")
        check(L.uvit_engine_profile(engine.h, 1, 4096), "profile on")
        for i in range(3):
            step(a.warmup + a.steps + i)
        fence()
        check(L.uvit_engine_profile(engine.h, 0, 0), "profile off")
        check(L.uvit_engine_profile_read(engine.h, C.byref(tot), C.byref(n), C.byref(fl)), "profile read")
        alone_ms = tot.value / max(n.value, 1)
        check(L.uvit_engine_set_streams(engine.h, 1), "set_streams")
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
        # headline: the dominant kernel's launches INSIDE the timed region (HIP events on the stream it is launched on).  In
        # the default two-stream schedule other kernels share the CUs with it there; "alone" is the same kernel by itself.
        achieved = flops / (in_ms * 1e-3) / 1e12 if in_n else 0.0
        out = {
            "metric": "pretrain images/sec (ViT-B/16 224, bs=128/GPU)", "value": round(value, 2), "unit": "img/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{a.model} data2vec pretrain step{' (--stochastic: two-stream + Wasserstein loss)' if stochastic else ''}, "
                                   f"bs={a.batch}/GPU, 224x224 synthetic, target_layers=upper half, attn_drop 0.05, drop_path 0.25, clip 3.0, "
                                   "AdamW, EMA 0.9998",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}",
                       "model": a.model, "step_mfma_frac": round(value / world * GFLOP_BY_MODEL.get(a.model, GFLOP_PER_IMAGE) * 1e9 / PEAK_BF16, 4),
                       "final_loss": round(float(stats[0]), 5),
                       "last_block_mlp_rows": (f"{feed['mask_rows']} masked rows of {a.batch * 197} (host-side bound; same results as all rows)"
                                               if feed["mask_rows"] and not stochastic else "all"),
                       "step_mfma_frac_note": "algorithmic FLOPs of the reference's step (SURVEY 8d) / time / peak"},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt256_kernel<EPI_GELU | EPI_GELU_DG> (fc1: M=25216 N=3072 K=768, bf16 MFMA, fused bias+GELU; the student launch also stores gelu'(h))",
                         "achieved": round(achieved, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / PEAK_BF16, 4), "traffic": pmc_traffic_bytes(),
                         "traffic_source": "profiles/round3_pmc_hbm.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_run.sh): 2 x FETCH_SIZE + WRITE_SIZE per launch",
                         "algorithmic_bytes": ALGO_BYTES, "launches_timed": in_n, "avg_launch_ms": round(in_ms, 4),
                         "flops_per_launch": flops,
                         "measured": "HIP events on the launch stream over the timed region (" +
                                     ("single stream: the kernel has the GPU to itself)" if a.single_stream else
                                      "two-stream schedule: other kernels share the CUs with it)"),
                         "alone": None if alone_ms is None else {
                             "avg_launch_ms": round(alone_ms, 4), "achieved": round(flops / (alone_ms * 1e-3) / 1e12, 2),
                             "frac": round(flops / (alone_ms * 1e-3) / PEAK_BF16, 4),
                             "measured": "3 single-stream steps right after the timed region (kernel alone on the GPU)"}},
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
