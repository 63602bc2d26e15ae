"""CPU (no GPU): host logic of the product path, the C-ABI library's symbols and layout, the CLI
surface, checkpoints, and the data-parallel reducer under gloo with world_size 2."""
import copy
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import vit_oracle as vo  # noqa: E402


@pytest.fixture(scope="module")
def native():
    from uncertainty_vit_amd import native as n
    n.build()                      # hipcc cross-compiles gfx950 without a GPU
    return n


def tiny_model(**kw):
    from functools import partial
    from uncertainty_vit_amd.modeling_cyclical import VisionTransformerForCyclicalTraining
    args = dict(img_size=48, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
                norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), init_values=0.1, use_shared_rel_pos_bias=True,
                use_abs_pos_emb=False)
    args.update(kw)
    return VisionTransformerForCyclicalTraining(**args)


def test_library_exports_every_declared_symbol(native):
    hdr = open(os.path.join(ROOT, "include", "uvit.h")).read()
    declared = set(re.findall(r"\b(uvit_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(native.SYMBOLS), declared ^ set(native.SYMBOLS)
    lib = C.CDLL(native.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s
    assert native.lib().uvit_version() == 100


def test_ctypes_struct_mirrors_follow_the_header(native):
    """The structs that cross the C ABI by value / by pointer are mirrored field by field in native.py: a field added to
    include/uvit.h and forgotten there shifts everything behind it.  uvit_tuning: names from the header's one-line typedef, and
    the library's defaults read back through the mirror (host-only call)."""
    hdr = open(os.path.join(ROOT, "include", "uvit.h")).read()
    m = re.search(r"typedef struct uvit_tuning \{ int32_t ([a-z_, ]+); \} uvit_tuning;", hdr)
    fields = [f.strip() for f in m.group(1).split(",")]
    assert [n for n, _ in native.Tuning._fields_] == fields
    assert C.sizeof(native.Tuning) == 4 * len(fields)
    t = native.Tuning.default()
    assert (t.nt_variant, t.tn_variant, t.tn_split_target, t.nt_persist) == (3, 3, 512, 1)
    body = hdr[hdr.index("typedef struct uvit_step_params"):hdr.index("} uvit_step_params;")]
    names = re.findall(r"\b([a-z_0-9]+)(?:\[[A-Z_0-9]+\])?\s*[,;]", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    mirror = [n for n, _ in native.StepParams._fields_]
    assert [n for n in names if n in mirror] == mirror, (names, mirror)


def test_arena_layout_matches_reference_state_dict(native):
    L = native.lib()
    cfg = native.Config(224, 16, 3, 768, 12, 12, 3072, 1, 0, 128, 1e-6, 0.05, 0.25, 0)
    nd = C.c_int64()
    total = L.uvit_arena_numel(C.byref(cfg), C.byref(nd))
    ref = vo.param_shapes(vo.VitConfig())
    seen, end = {}, 0
    spans = []
    for i in range(L.uvit_layout_count(C.byref(cfg))):
        e = native.LayoutEntry()
        assert L.uvit_layout_get(C.byref(cfg), i, C.byref(e)) == 0
        name, shape = e.name.decode(), tuple(e.shape[: e.ndim])
        seen[name] = shape
        assert e.offset % 64 == 0 and e.numel == int(np.prod(shape))
        assert (e.offset < nd.value) == (e.decay == 1)
        spans.append((e.offset, e.offset + e.numel))
        end = max(end, e.offset + e.numel)
    assert seen == ref                                        # 189 float tensors, reference names and shapes
    assert sum(int(np.prod(s)) for s in seen.values()) == 86_256_720
    spans.sort()
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and end <= total
    decay = {n for n in seen if n not in vo.no_decay_names({k: torch.empty(v) for k, v in seen.items()})}
    assert decay == {e for e in seen if [1 for i in range(1)] and seen[e] and (len(seen[e]) > 1 and not e.endswith(".bias") and e != "cls_token")}
    # unsupported shapes are refused, not approximated
    bad = native.Config(224, 16, 3, 760, 12, 12, 3072, 1, 0, 128, 1e-6, 0.0, 0.0, 0)
    assert L.uvit_arena_numel(C.byref(bad), None) < 0


def test_model_surface_matches_reference(native, golden_dir):
    m = tiny_model()
    cfg = vo.VitConfig(img_size=48, embed_dim=128, depth=2, num_heads=2, init_values=0.1)
    assert [(k, tuple(v.shape)) for k, v in m.named_parameters()] == list(vo.param_shapes(cfg).items())
    sd = m.state_dict()
    assert "rel_pos_bias.relative_position_index" in sd and sd["rel_pos_bias.relative_position_index"].dtype == torch.int64
    assert torch.equal(sd["rel_pos_bias.relative_position_index"], torch.from_numpy(vo.relative_position_index(3)))
    assert m.patch_embed.patch_size == (16, 16) and m.patch_embed.patch_shape == (3, 3) and m.patch_embed.num_patches == 9
    assert m.get_num_layers() == 2 and m.no_weight_decay() == {"pos_embed", "cls_token"} and "mean" in m.default_cfg
    # init rule (modeling_cyclical.py:135-161)
    assert torch.all(sd["blocks.0.gamma_1"] == 0.1) and torch.all(sd["norm.weight"] == 1) and torch.all(sd["lm_head.bias"] == 0)
    assert sd["rel_pos_bias.relative_position_bias_table"].abs().sum() == 0
    assert sd["cls_token"].abs().max() <= 0.02 + 1e-7 and sd["blocks.1.mlp.fc1.weight"].std() > 0.005
    # parameters are views of ONE arena; deepcopy (ModelEmaV2) gets its own
    m2 = copy.deepcopy(m)
    assert m2._arena.data_ptr() != m._arena.data_ptr()
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    with torch.no_grad():
        m.cls_token.add_(1.0)
    assert m._arena[m._layout[[n for n, *_ in m._layout].index("cls_token")][1]] == m.cls_token.view(-1)[0]
    assert not torch.equal(m.cls_token, m2.cls_token)
    # state-dict round trip (checkpoints load both ways)
    m2.load_state_dict(m.state_dict())
    assert torch.equal(m2._arena, m._arena)
    # the groups of optim_factory.py:58-97
    fx = np.load(os.path.join(golden_dir, "model_t48.npz"))
    from uncertainty_vit_amd.optim_factory import ArenaAdamW
    opt = ArenaAdamW(m, lr=1e-3, weight_decay=0.05)
    assert opt.group_names["decay"] == fx["groups/decay"].tolist()
    assert opt.group_names["no_decay"] == fx["groups/no_decay"].tolist()


def test_no_cpu_fallback_and_rejected_options(native):
    m = tiny_model()
    with pytest.raises(native.UvitError):
        m(torch.zeros(1, 3, 48, 48), None, True, layer_results="end")
    for kw in (dict(gp_layer=True), dict(sinkformer=True), dict(init_values=None), dict(embed_dim=96, num_heads=2)):
        with pytest.raises((NotImplementedError, TypeError)):
            tiny_model(**kw)
    # --abs_pos_emb is native since round 2: pos_embed (1, N, C) sits right after the tokens in the state dict
    # (modeling_cyclical.py:80-84) and in the no-decay group (no_weight_decay(), :163-165)
    mp = tiny_model(use_abs_pos_emb=True)
    keys = list(mp.state_dict())
    assert keys[:3] == ["cls_token", "mask_token", "pos_embed"] and tuple(mp.pos_embed.shape) == (1, mp.patch_embed.num_patches + 1, mp.embed_dim)
    assert float(mp.pos_embed.abs().max()) > 0 and float(mp.pos_embed.abs().max()) <= mp.init_std + 1e-6       # trunc-normal on [-std, std]
    from uncertainty_vit_amd.modeling_cyclical import create_model
    with pytest.raises(RuntimeError):
        create_model("not_a_model")


def test_schedules_match_reference_tables(native, golden_dir):
    from uncertainty_vit_amd import utils
    fx = np.load(os.path.join(golden_dir, "schedules.npz"))
    np.testing.assert_allclose(utils.cosine_scheduler(2e-3, 1e-5, 4, 10, warmup_epochs=1, start_warmup_value=1e-6), fx["cos_a"], rtol=1e-12)
    np.testing.assert_allclose(utils.cosine_scheduler(0.05, 0.05, 3, 7), fx["cos_b"], rtol=1e-12)
    np.testing.assert_allclose(utils.cosine_scheduler(5e-4, 1e-6, 5, 9, warmup_epochs=2, start_warmup_value=1e-6, warmup_steps=4), fx["cos_c"], rtol=1e-12)
    np.testing.assert_allclose(utils.tri_phase_scheduler(2e-3, 1e-5, 5, 20, warmup_perc=0.05, decay_perc=0.15, start_warmup_value=1e-6), fx["tri_a"], rtol=1e-12)
    np.testing.assert_allclose(utils.tri_phase_scheduler(1e-3, 0.0, 2, 10, warmup_perc=0.0, decay_perc=0.5), fx["tri_b"], rtol=1e-12)


def test_cli_flags_and_defaults_match_reference(native, golden_dir):
    sys.path.insert(0, ROOT)
    import run_cyclical
    ref = json.load(open(os.path.join(golden_dir, "cli_defaults.json")))
    mine = vars(run_cyclical.get_args([]))
    assert mine.pop("synthetic_len") == 1024 and mine.pop("synthetic_masks") == "uniform"     # the only new flags
    assert mine == ref
    a = run_cyclical.get_args("--model beit_base_patch16_224 --stochastic --target_layers [6,7,8,9,10,11] --data_set SYNTHETIC".split())
    assert a.stochastic and a.model == "beit_base_patch16_224" and a.data_set == "SYNTHETIC"
    ds = run_cyclical.SyntheticPretrainSet(8, 224, (14, 14), 120, seed=0)
    (img, mask), _ = ds[3]
    assert img.shape == (3, 224, 224) and mask.shape == (14, 14) and mask.dtype == torch.int64 and int(mask.sum()) == 120
    assert torch.equal(ds[3][0][1], mask)


def test_meters_and_checkpoint_roundtrip(native, tmp_path):
    from types import SimpleNamespace
    from uncertainty_vit_amd import utils
    from uncertainty_vit_amd.optim_factory import ArenaAdamW
    ml = utils.MetricLogger(delimiter="  ")
    ml.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    for v in (1.0, 2.0, 6.0):
        ml.update(loss=v, lr=0.5, skipped=None)
    assert ml.loss.global_avg == 3.0 and ml.loss.median == 2.0 and str(ml.lr) == "0.500000" and "skipped" not in ml.meters
    assert list(ml.log_every([1, 2, 3], 10, "h")) == [1, 2, 3]
    m = tiny_model()
    ema = utils.ModelEmaV2(m, decay=0.5)
    with torch.no_grad():
        m._arena.add_(1.0)
    before = ema.module._arena.clone()
    ema._update(m, update_fn=lambda e, mm: 0.5 * e + 0.5 * mm)             # engine_for_cyclical.py:183 lambda
    torch.testing.assert_close(ema.module._arena[:128], 0.5 * before[:128] + 0.5 * m._arena[:128])
    opt = ArenaAdamW(m, lr=1e-3, weight_decay=0.05)
    opt._ensure_state()
    opt.exp_avg.fill_(0.25)
    opt.step_count = 7
    args = SimpleNamespace(output_dir=str(tmp_path), auto_resume=True, resume="", start_epoch=0)
    utils.save_model(args, 4, m, m, opt, utils.NativeScalerWithGradNormCount(), model_ema=ema)
    ck = torch.load(os.path.join(tmp_path, "checkpoint-4.pth"), weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args", "model_ema"} and set(ck["model"]) == set(m.state_dict())
    m2 = tiny_model()
    opt2 = ArenaAdamW(m2, lr=1e-3, weight_decay=0.05)
    utils.auto_load_model(args, m2, m2, opt2, utils.NativeScalerWithGradNormCount(), model_ema=None)
    assert args.start_epoch == 5 and opt2.step_count == 7
    # the optimizer entry is per parameter (the reference's torch.optim.AdamW structure): every parameter's slice comes back
    assert all(torch.all(opt2.exp_avg[o:o + k] == 0.25) for _, o, k, _, _ in m2._layout)
    assert set(ck["optimizer"]) == {"state", "param_groups"} and set(ck["optimizer"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), m.state_dict().values()))


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["UVIT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["UVIT_ROOT"], "tests"))
from test_host_cpu import tiny_model
from uncertainty_vit_amd import utils
from uncertainty_vit_amd.engine_for_cyclical import GradReducer, make_step_params
from uncertainty_vit_amd.optim_factory import ArenaAdamW
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", init_method="env://", world_size=2, rank=rank)
m = tiny_model()
red = GradReducer(m, True)
assert red.enabled and red.world == 2
g = torch.full_like(m._arena, float(rank + 1))
# the buckets tile the arena exactly once: head, blocks (reverse order, as backward produces them), embed, small
for l in range(m.depth - 1, -1, -1):
    red.reduce(g, red.layer_ranges[l])
red.reduce(g, red.head_range); red.reduce(g, red.embed_range); red.reduce(g, red.small_range)
red.finish()
assert torch.all(g == 3.0), "every gradient element must be summed over the 2 ranks exactly once"
# bf16 buckets (SURVEY 8e: 164.5 MB on the wire instead of 329 MB): same tiling, values rounded to bf16 once per rank
red16 = GradReducer(m, True, comm_dtype=torch.bfloat16)
g16 = torch.full_like(m._arena, 1.0 + 2.0 ** -10) * float(rank + 1)      # 1 + 2^-10 is not a bf16 number
for l in range(m.depth - 1, -1, -1):
    red16.reduce(g16, red16.layer_ranges[l])
red16.reduce(g16, red16.head_range); red16.reduce(g16, red16.embed_range); red16.reduce(g16, red16.small_range)
red16.finish()
assert g16.dtype == torch.float32 and torch.all(g16 == 3.0), "bf16 wire: round(1+2^-10) + round(2+2^-9) = 3, widened back to fp32"
hp = make_step_params([1], ArenaAdamW(m, 1e-3, 0.05), 3.0, 2.0, False, -1, True, True, 0.9998, True, utils.get_world_size(), 0, 0)
assert abs(hp.grad_scale - 0.5) < 1e-9          # SUM all-reduce then 1/world == DDP's mean
sv = utils.SmoothedValue(); sv.update(float(rank + 1)); sv.synchronize_between_processes()
assert sv.count == 2 and sv.total == 3.0
# DDP's constructor broadcast (run_cyclical.py:516): ranks that start from different weights / EMA / moments continue from rank 0's
m2 = tiny_model(); ema2 = utils.ModelEmaV2(m2, decay=0.5); opt2 = ArenaAdamW(m2, 1e-3, 0.05); opt2._ensure_state()
with torch.no_grad():
    m2._arena.fill_(float(rank + 1)); ema2.module._arena.fill_(10.0 * (rank + 1)); opt2.exp_avg.fill_(100.0 * (rank + 1)); opt2.exp_avg_sq.fill_(7.0 + rank)
m2._shadows_stale = False
assert utils.broadcast_model_state(m2, ema2, opt2, src=0) == 4
assert torch.all(m2._arena == 1.0) and torch.all(ema2.module._arena == 10.0) and torch.all(opt2.exp_avg == 100.0) and torch.all(opt2.exp_avg_sq == 7.0)
assert m2._shadows_stale, "the bf16 shadows must be rebuilt after the broadcast"
assert all(torch.all(p == 1.0) for p in m2.parameters()), "the nn.Parameters are views of the arena"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_data_parallel_reducer_gloo_world2(native, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, UVIT_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
        assert "ok" in o


def test_blockwise_mask_generator_matches_reference(golden_dir):
    """Bit-exact against masks drawn from the reference generator under the same random stream."""
    from uncertainty_vit_amd.masking_generator import MaskingGenerator
    fx = np.load(os.path.join(golden_dir, "masks.npz"))
    cfgs = {"a": (14, 120, 16, None), "b": (14, 75, 16, None), "c": (14, 120, 4, 40), "d": (7, 20, 4, None)}
    for name, (size, n, mn, mx) in cfgs.items():
        for seed in (0, 1, 2):
            g = MaskingGenerator(size, n, min_num_patches=mn, max_num_patches=mx, seed=1000 * seed + 7)
            got = np.stack([g() for _ in range(4)])
            assert got.dtype == np.int64 and np.array_equal(got, fx[f"{name}{seed}"]), (name, seed)
            assert (got.sum(axis=(1, 2)) <= n).all() and got.max() <= 1        # at most n ones
            # the batched call (one per step in the loader) consumes the same stream: identical masks
            gb = MaskingGenerator(size, n, min_num_patches=mn, max_num_patches=mx, seed=1000 * seed + 7)
            assert np.array_equal(gb.batch(4), fx[f"{name}{seed}"]), (name, seed, "batch")


def test_device_prefetcher_passes_batches_through_in_order():
    from uncertainty_vit_amd.engine_for_cyclical import DevicePrefetcher
    batches = [((torch.full((2, 3, 4, 4), float(i), dtype=torch.float64), torch.ones(2, 2, 2, dtype=torch.bool)), i) for i in range(5)]
    pf = DevicePrefetcher(batches, "cpu")
    assert len(pf) == 5
    got = list(pf)
    assert [lbl for _, lbl in got] == list(range(5))
    assert all(s.dtype == torch.float32 and float(s[0, 0, 0, 0]) == i for i, ((s, _), _) in enumerate(got))
    assert list(DevicePrefetcher([], "cpu")) == []


def test_prefetcher_counts_the_masked_rows_of_the_batch_it_hands_out():
    """The host-side row bound of the step (uvit_step_params.n_rows_hint) is the number of masked patches of the batch being consumed,
    counted on the CPU mask before the upload -- not of the batch already prefetched behind it."""
    from uncertainty_vit_amd.engine_for_cyclical import DevicePrefetcher, make_step_params
    masks = [torch.zeros(2, 4, 4, dtype=torch.bool) for _ in range(3)]
    for i, m in enumerate(masks):
        m.view(-1)[: 3 * i + 1] = True
    pf = DevicePrefetcher([((torch.zeros(2, 3, 4, 4), m), 0) for m in masks], "cpu")
    assert pf.mask_rows == 0
    seen = []
    for (_, m), _ in pf:
        seen.append((pf.mask_rows, int(m.sum())))
    assert seen == [(1, 1), (4, 4), (7, 7)]

    class Opt:
        param_groups = [dict(lr=1e-3, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8)]
        step_count = 0
    hp = make_step_params([1], Opt(), 3.0, 2.0, False, -1, True, True, 0.9998, True, 1, 0, 0, n_rows_hint=7)
    assert hp.n_rows_hint == 7 and make_step_params([1], Opt(), 3.0, 2.0, False, -1, True, True, 0.9998, True, 1, 0, 0).n_rows_hint == 0


def test_bench_self_launch_command_and_clean_failure_without_gpus():
    """`python bench.py --gpus N` must work when invoked directly: it builds a torch.distributed.run
    command for N ranks on 127.0.0.1 before touching the GPU, and on a machine with fewer GPUs it exits 2 with a message
    instead of tripping an assert."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launch_command(4, 29777, ["--gpus", "4", "--steps", "3"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29777" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], os.path.join(ROOT, "bench.py"))
    if torch.cuda.device_count() < 2:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 2 and "needs 2 visible GPUs" in r.stderr and "AssertionError" not in r.stderr, r.stderr[-800:]
        # a launcher whose world size disagrees with --gpus is refused the same way
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                           capture_output=True, text=True, timeout=240)
        assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_reducer_waits_for_the_collective_on_the_stream_that_consumes_it(monkeypatch):
    """Over RCCL Work.wait() only makes the CURRENT stream wait for the collective.  GradReducer.finish() enqueues the widening
    copy of a bf16 bucket on its communication stream, so the wait must be issued with THAT stream current (with the
    compute stream current the copy could read the wire buffer before RCCL had written it; gloo blocks the host and cannot see
    it).  Fake stream / work objects record which stream each wait and copy was issued on; the old ordering fails here."""
    from uncertainty_vit_amd.engine_for_cyclical import GradReducer
    state = {"current": "compute"}
    log = []

    class FakeStream:
        def __init__(self, name): self.name = name
        def wait_stream(self, other): log.append(("join", self.name, other.name))

    class FakeCtx:
        def __init__(self, st): self.st = st
        def __enter__(self): self.prev = state["current"]; state["current"] = self.st.name
        def __exit__(self, *a): state["current"] = self.prev

    class FakeWork:
        def __init__(self, i): self.i, self.waited_on = i, set()
        def wait(self): self.waited_on.add(state["current"]); log.append(("wait", self.i, state["current"]))

    class FakeView:
        def __init__(self, work): self.work = work
        def copy_(self, wire):
            # the stream the copy is enqueued on must already be waiting for the collective that fills `wire`
            assert state["current"] in self.work.waited_on, f"copy on '{state['current']}' but the collective was only awaited on {self.work.waited_on}"
            log.append(("copy", self.work.i, state["current"]))

    monkeypatch.setattr(torch.cuda, "stream", lambda st: FakeCtx(st))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: FakeStream(state["current"]))
    red = GradReducer.__new__(GradReducer)
    red.enabled, red.on_gpu, red.comm, red.comm_dtype = True, True, FakeStream("comm"), torch.bfloat16
    works = [FakeWork(i) for i in range(3)]
    red.pending = [(works[0], object(), FakeView(works[0])), (works[1], None, None), (works[2], object(), FakeView(works[2]))]
    red.finish()
    assert red.pending == []
    assert [e for e in log if e[0] == "wait"] == [("wait", 0, "comm"), ("wait", 1, "comm"), ("wait", 2, "comm")]
    assert [e for e in log if e[0] == "copy"] == [("copy", 0, "comm"), ("copy", 2, "comm")]
    assert log[-1] == ("join", "compute", "comm") and state["current"] == "compute"      # AdamW (compute stream) runs behind every bucket


def test_rccl_channel_cap_is_a_default_not_an_override(monkeypatch):
    """utils.cap_rccl_channels: NCCL_MAX_NCHANNELS defaults to 12 (each channel workgroup holds a CU the one-round backward
    kernels count on) but a value the user exported wins."""
    from uncertainty_vit_amd import utils
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    assert utils.cap_rccl_channels() == "12" and os.environ["NCCL_MAX_NCHANNELS"] == "12"
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "32")
    assert utils.cap_rccl_channels() == "32"


def test_target_layers_follow_python_list_indexing(native):
    """`[targets[i] for i in target_layers]` (engine_for_cyclical.py:92): negative indices count from the last block,
    out-of-range raises IndexError, a repeated index is kept (it is averaged twice, the divisor is len(target_layers))."""
    from uncertainty_vit_amd.engine_for_cyclical import make_step_params
    from uncertainty_vit_amd.optim_factory import ArenaAdamW
    opt = ArenaAdamW(tiny_model(), 1e-3, 0.05)
    mk = lambda tl: make_step_params(tl, opt, 3.0, 2.0, False, -1, True, True, 0.9998, True, 1, 0, 0, depth=12)  # noqa: E731
    hp = mk([-1, 10, 10, -12])
    assert hp.n_target_layers == 4 and list(hp.target_layers[:4]) == [11, 10, 10, 0]
    for bad in ([12], [-13], [6, 7, 99]):
        with pytest.raises(IndexError):
            mk(bad)


def test_epoch_scalars_follow_the_reference_loop_state(native):
    """Device schedule table = the per-iteration scalars of engine_for_cyclical.py:41,47-56,182-185, with
    cur_decay as loop STATE: annealed while it < ema_start_at, then frozen at its last annealed value for the rest of the
    epoch; 0 / "skip" once it passes start_lr_decay_at_step."""
    import ctypes
    from uncertainty_vit_amd.engine_for_cyclical import epoch_scalars
    from uncertainty_vit_amd.optim_factory import ArenaAdamW
    f32 = lambda v: ctypes.c_float(float(v)).value  # noqa: E731
    opt = ArenaAdamW(tiny_model(), 1e-3, 0.05)
    lr_tab = [1e-4 * (i + 1) for i in range(20)]
    wd_tab = [0.05 + 0.01 * i for i in range(20)]
    sc = epoch_scalars(opt, 4, 8, lr_tab, wd_tab, ema_start_at=7, decay_init=0.99, decay=0.9998, start_lr_decay_at_step=9)
    assert [s[0] for s in sc] == [f32(lr_tab[i]) for i in range(4, 12)] and [s[1] for s in sc] == [f32(wd_tab[i]) for i in range(4, 12)]
    anneal = lambda it: 0.99 + it * (0.9998 - 0.99) / 7  # noqa: E731
    assert sc[0][2] == f32(anneal(4)) and sc[2][2] == f32(anneal(6))
    assert sc[3][2] == f32(anneal(6)) and sc[5][2] == f32(anneal(6))        # it = 7..9: frozen at the last annealed value
    assert sc[6][2] == -1.0 and sc[7][2] == -1.0                             # it = 10, 11 > start_lr_decay_at_step: EMA skipped
    flat = epoch_scalars(opt, 0, 3, None, None, 0, 0.99, 0.9998, -1)
    assert flat == [(f32(1e-3), f32(0.05), f32(0.9998))] * 3
    assert epoch_scalars(opt, 0, 2, None, None, 0, 1.0, 1.0, -1)[0][2] == -1.0       # decay 1: `cur_decay != 1` is false


def test_optimizer_state_dict_interchanges_with_torch_adamw(native):
    """Checkpoint interchange (VERDICT r1 "Missing 8", utils.py:462-479): ArenaAdamW.state_dict() has the structure of the
    torch.optim.AdamW the reference builds over the same two groups -- torch loads it, and a torch state dict loads into
    the arenas at the right offsets -- so `checkpoint-N.pth['optimizer']` moves both ways."""
    from uncertainty_vit_amd.optim_factory import ArenaAdamW, get_parameter_groups
    m = tiny_model()
    opt = ArenaAdamW(m, lr=1e-3, weight_decay=0.05)
    groups, _ = get_parameter_groups(m, 0.05, m.no_weight_decay())
    ref = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    g = torch.Generator().manual_seed(0)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    before = m._arena.clone()
    ref.step(); ref.step()
    m._arena.copy_(before)                    # torch stepped the shared parameters: restore, only the state matters here
    tsd = ref.state_dict()
    opt.load_state_dict(tsd)                  # torch -> arena
    assert opt.step_count == 2
    lay = {n: (o, k, shape) for n, o, k, shape, _ in m._layout}
    order = [n for grp in ("decay", "no_decay") for n in opt.group_names[grp]]
    for i, n in enumerate(order):
        off, numel, shape = lay[n]
        assert torch.equal(opt.exp_avg[off:off + numel].view(shape), tsd["state"][i]["exp_avg"]), n
        assert torch.equal(opt.exp_avg_sq[off:off + numel].view(shape), tsd["state"][i]["exp_avg_sq"]), n
    asd = opt.state_dict()                    # arena -> torch
    assert [g_["params"] for g_ in asd["param_groups"]] == [g_["params"] for g_ in tsd["param_groups"]]
    ref2 = torch.optim.AdamW(get_parameter_groups(m, 0.05, m.no_weight_decay())[0], lr=1e-3)
    ref2.load_state_dict(asd)
    for i in tsd["state"]:
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(ref2.state_dict()["state"][i][k], tsd["state"][i][k])
        assert float(ref2.state_dict()["state"][i]["step"]) == 2.0
    assert ArenaAdamW(tiny_model(), 1e-3, 0.05).state_dict()["state"] == {}      # nothing stepped yet: empty, as in torch


def test_mfma_hazard_guard_flags_the_broken_stream_and_passes_the_tree():
    """tools/check_mfma_hazard.py (run by build.sh): the hand-written broken stream and the hipcc-compiled micro kernel with a
    wave-uniform branch between an MFMA and the read of its result are flagged; the attention kernels of the tree are clean."""
    tool = os.path.join(ROOT, "tools", "check_mfma_hazard.py")
    assert subprocess.run([sys.executable, tool, "--self-test"], capture_output=True, text=True).returncode == 0
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    r = subprocess.run([sys.executable, tool, "--compile", os.path.join(ROOT, "tools", "micro", "mfma_branch_hazard.hip")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "taken branch" in r.stdout, r.stdout + r.stderr
    csrc = os.path.join(ROOT, "uncertainty-vit_amd", "csrc")
    r = subprocess.run([sys.executable, tool, "--compile", os.path.join(csrc, "attention.hip"), os.path.join(csrc, "attention2.hip")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("draws,depth,B,rate", [(2, 12, 128, 0.25), (2, 24, 64, 0.4), (4, 12, 128, 0.25), (2, 3, 6, 0.3), (4, 4, 12, 0.5)])
def test_host_side_drop_path_counts_match_the_oracle_draws(native, draws, depth, B, rate):
    """uvit_drop_path_kept_counts (include/uvit.h): the engine sizes the compact launches of a step (drop-path sample lists) from ITS evaluation of
    the counter-based hash the device draws the DropPath multipliers with.  Pure host arithmetic, so it is pinned here, without a GPU, against
    the oracle's replay of the same draws (oracle/vit_oracle.py::drop_path_scales, oracle/vit_oracle_dist.py::drop_path_scales: the masks the
    GPU parity steps replay) over several seeds and iterations -- kept samples per (layer, draw)."""
    from oracle import vit_oracle_dist as vd
    L = native.lib()
    cfg = vo.VitConfig(depth=depth, drop_path_rate=rate)
    for seed, it in [(0, 0), (1234, 7), (0xFFFFFFFF, 3), (99, 100000)]:
        out = (C.c_int32 * (draws * depth))()
        assert L.uvit_drop_path_kept_counts(depth, C.c_float(rate), draws, B, C.c_uint32(seed), C.c_uint32(it), out) == 0
        got = np.array(out[:]).reshape(depth, draws)
        if draws == 2:
            p1, p2 = vo.drop_path_scales(seed, it, cfg, B)
            cols = [p1, p2]
        else:
            cols = vd.drop_path_scales(seed, it, cfg, B)
        want = np.array([[B if cols[k][l] is None else int((cols[k][l] != 0).sum()) for k in range(draws)] for l in range(depth)])
        assert (got == want).all(), (seed, it, got.tolist(), want.tolist())
        assert (got[0] == B).all() and (rate == 0 or got.min() < B)          # rate 0 in the first block; somebody is dropped further up
    assert L.uvit_drop_path_kept_counts(depth, C.c_float(rate), 3, B, 0, 0, out) != 0      # draws per block: 2 or 4
