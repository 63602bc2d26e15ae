"""Host-side mirror of the reference's model surface for the data2vec pre-training path.

Same names and call contract as /root/reference/modeling_cyclical.py:
  * `VisionTransformerForCyclicalTraining(img_size, patch_size, ..., init_values, use_shared_rel_pos_bias, ...)`
    (modeling_cyclical.py:33-60) with `forward(x, bool_masked_pos, return_all_tokens=False, layer_results=None)`
    and its three return modes (modeling_cyclical.py:209-225);
  * registry entry points `beit_base_patch16_224` / `beit_large_patch16_224` reached through
    `create_model` (modeling_cyclical.py:282-301, run_cyclical.py:289-302);
  * identical state-dict key names / shapes (checkpoints load both ways).
Nothing here computes: parameters are views into ONE flat fp32 arena whose layout comes from the
native library (uvit_layout_get), and forward / backward run as hand-written HIP kernels through
the C ABI (include/uvit.h).  There is no eager/CPU fallback: calling forward off-GPU raises.
"""
import ctypes as C
import math
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import native
from .native import Buffers, Config, LayoutEntry, check, cur_stream, lib, ptr

__all__ = ["beit_base_patch16_224", "dist_beit_base_patch16_224", "beit_large_patch16_224", "dist_beit_large_patch16_224", "create_model", "register_model",
           "VisionTransformerForCyclicalTraining", "DistVisionTransformerForCyclicalTraining"]


def _cfg(url="", **kwargs):
    # modeling_finetune.py:27-35
    return {"url": url, "num_classes": 1000, "input_size": (3, 224, 224), "pool_size": None, "crop_pct": 0.9,
            "interpolation": "bicubic", "mean": (0.5, 0.5, 0.5), "std": (0.5, 0.5, 0.5), **kwargs}


def trunc_normal_(tensor, mean=0.0, std=1.0):
    """modeling_cyclical.py:23-24: truncated normal on [-std, std]."""
    with torch.no_grad():
        lo = (1.0 + math.erf(-1.0 / math.sqrt(2.0))) / 2.0
        hi = (1.0 + math.erf(1.0 / math.sqrt(2.0))) / 2.0
        tensor.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.0)).add_(mean)
        tensor.clamp_(min=mean - std, max=mean + std)
    return tensor


def relative_position_index(ws):
    """modeling_finetune.py:339-353."""
    n_rel = (2 * ws - 1) * (2 * ws - 1) + 3
    ys, xs = np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    idx = np.zeros((ws * ws + 1,) * 2, dtype=np.int64)
    idx[1:, 1:] = (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)
    idx[0, :] = n_rel - 3
    idx[:, 0] = n_rel - 2
    idx[0, 0] = n_rel - 1
    return torch.from_numpy(idx)


class _Holder(nn.Module):
    """Container that only exists to give parameters their reference state-dict names."""


class _PatchEmbedInfo(_Holder):
    def __init__(self, img_size, patch_size):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.patch_shape = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.proj = _Holder()


class NativeEngine:
    """Owns the uvit_engine handle plus the torch tensors backing its buffers."""

    def __init__(self, model, teacher, batch, adam_m=None, adam_v=None):
        L = lib()
        dev = model._arena.device
        if dev.type != "cuda":
            raise native.UvitError("the HIP path needs the model on a GPU (no CPU fallback)")
        n = model._arena.numel()
        self.model, self.teacher, self.batch = model, teacher, batch
        self.cfg = model._native_config(batch)
        z = lambda dt=torch.float32, k=n: torch.zeros(k, dtype=dt, device=dev)  # noqa: E731
        self.grads = model._grad_arena if model._grad_arena is not None else z()
        model._grad_arena = self.grads
        self.adam_m = adam_m if adam_m is not None else z()
        self.adam_v = adam_v if adam_v is not None else z()
        self.ema = teacher._arena if teacher is not None else model._arena
        self.params_bf16, self.params_bf16_t, self.ema_bf16 = z(torch.bfloat16), z(torch.bfloat16), z(torch.bfloat16)
        self.rel_index = None
        if model.rel_pos_bias is not None:
            self.rel_index = model.rel_pos_bias.relative_position_index.to(dev, torch.int32).contiguous().view(-1)
        ws = L.uvit_workspace_bytes(C.byref(self.cfg))
        if ws < 0:
            check(int(ws), "uvit_workspace_bytes")
        self.workspace = torch.empty(ws, dtype=torch.uint8, device=dev)
        b = Buffers(model._arena.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                    self.ema.data_ptr(), self.params_bf16.data_ptr(), self.params_bf16_t.data_ptr(),
                    self.ema_bf16.data_ptr(), 0 if self.rel_index is None else self.rel_index.data_ptr(),
                    self.workspace.data_ptr(), ws)
        err = C.c_int(0)
        with torch.cuda.device(dev):
            self.h = L.uvit_engine_create(C.byref(self.cfg), C.byref(b), cur_stream(), C.byref(err))
        if not self.h:
            check(err.value or -1, "uvit_engine_create")
        self.h = C.c_void_p(self.h)
        # second-stream mode (include/uvit.h, uvit_engine_set_streams): below the caller's stream when this process has the GPU to itself;
        # equal priority in data-parallel runs, where RCCL's channel workgroups hold CUs during backward
        import os
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.stream_mode = 0 if os.environ.get("UVIT_SINGLE_STREAM") == "1" else (2 if world == 1 else 1)
        if os.environ.get("UVIT_STREAM_MODE"):          # A/B runs
            self.stream_mode = int(os.environ["UVIT_STREAM_MODE"])
        check(L.uvit_engine_set_streams(self.h, self.stream_mode), "set_streams")
        # drop-path sample lists (include/uvit.h, uvit_engine_set_drop_path_rows): `model.drop_path_rows = False` before the first step, or
        # set_drop_path_rows(False) later, runs every branch on every sample as the reference does (same results; bench.py's all-rows line)
        self.drop_path_rows = bool(getattr(model, "drop_path_rows", os.environ.get("UVIT_DP_ROWS", "1") != "0"))
        self.set_drop_path_rows(self.drop_path_rows)
        self.sync_shadows(3)

    def set_drop_path_rows(self, on):
        self.drop_path_rows = bool(on)
        check(lib().uvit_engine_set_drop_path_rows(self.h, 1 if on else 0), "set_drop_path_rows")

    def sync_shadows(self, which=3):
        check(lib().uvit_engine_sync_shadows(self.h, which, cur_stream()), "sync_shadows")

    def compact_rows(self):
        """Rows the last training step ran its last block's MLP on (0 = every token row)."""
        return int(lib().uvit_engine_compact_rows(self.h))

    def ws_tensor(self, name, layer, shape, dtype=torch.float32):
        p = lib().uvit_engine_ws_ptr(self.h, name.encode(), layer)
        if not p:
            raise native.UvitError(f"no workspace buffer {name}[{layer}]")
        off = p - self.workspace.data_ptr()
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return self.workspace[off:off + n].view(dtype).view(*shape)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().uvit_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass


class VisionTransformerForCyclicalTraining(nn.Module):
    _two_stream = False      # True in DistVisionTransformerForCyclicalTraining

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=None,
                 init_values=None, attn_head_dim=None, use_abs_pos_emb=True, use_rel_pos_bias=False,
                 use_shared_rel_pos_bias=False, init_std=0.02, gp_layer=False, gumbel_softmax=False, sinkformer=False,
                 h_sto_trans=False, stosa=False):
        super().__init__()
        self._ctor = dict(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth,
                          num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, attn_drop_rate=attn_drop_rate,
                          drop_path_rate=drop_path_rate, norm_layer=norm_layer, init_values=init_values,
                          use_abs_pos_emb=use_abs_pos_emb, use_shared_rel_pos_bias=use_shared_rel_pos_bias, init_std=init_std)
        # options outside the data2vec pre-training hot path are rejected, not silently ignored
        unsupported = dict(qk_scale=qk_scale, attn_head_dim=attn_head_dim, gp_layer=gp_layer, gumbel_softmax=gumbel_softmax,
                           sinkformer=sinkformer, h_sto_trans=h_sto_trans, stosa=stosa, use_rel_pos_bias=use_rel_pos_bias,
                           use_abs_pos_emb=use_abs_pos_emb and self._two_stream)   # the two-stream model has no pos_embed
        bad = [k for k, v in unsupported.items() if v]
        if bad or not qkv_bias or drop_rate:
            raise NotImplementedError(f"not on the MI355X hot path (SURVEY.md section 8): {bad or 'qkv_bias/drop_rate'}")
        if init_values is None or not init_values > 0:
            # modeling_finetune.py:284 `if init_values > 0` -- None raises TypeError there too
            raise TypeError("init_values must be > 0 (LayerScale is part of the configured path)")
        if embed_dim != num_heads * 64:
            raise NotImplementedError("attention kernels are specialised for head_dim 64 (ViT-B/L/H)")
        self.num_features = self.embed_dim = embed_dim
        self.depth, self.num_heads, self.mlp_hidden = depth, num_heads, int(embed_dim * mlp_ratio)
        self.img_size, self.in_chans = img_size, in_chans
        self.attn_drop_rate, self.drop_path_rate, self.init_values = attn_drop_rate, drop_path_rate, init_values
        self.ln_eps = 1e-6
        if norm_layer is not None:
            probe = norm_layer(4)
            self.ln_eps = float(getattr(probe, "eps", 1e-6))
        self.init_std = init_std
        self.patch_embed = _PatchEmbedInfo(img_size, patch_size)
        self.stosa = False
        self.use_abs_pos_emb = bool(use_abs_pos_emb)
        if not self.use_abs_pos_emb:
            self.pos_embed = None            # otherwise an arena view registered below (modeling_cyclical.py:80-84)
        ws = self.patch_embed.patch_shape[0]

        # ---- one flat arena; parameters are views (layout owned by the native library) ----
        cfg = self._native_config(1)
        L = lib()
        nd = C.c_int64()
        n = L.uvit_arena_numel(C.byref(cfg), C.byref(nd))
        if n < 0:
            check(int(n), "uvit_arena_numel (unsupported model shape)")
        self._n_decay = nd.value
        self._arena = torch.zeros(n, dtype=torch.float32)
        self._grad_arena = None
        self._layout = []
        for i in range(L.uvit_layout_count(C.byref(cfg))):
            e = LayoutEntry()
            check(L.uvit_layout_get(C.byref(cfg), i, C.byref(e)), "uvit_layout_get")
            self._layout.append((e.name.decode(), int(e.offset), int(e.numel), tuple(e.shape[:e.ndim]), int(e.decay)))
        self.blocks = nn.ModuleList()
        for _ in range(depth):
            blk = _Holder()
            blk.norm1, blk.norm2, blk.attn, blk.mlp = _Holder(), _Holder(), _Holder(), _Holder()
            blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2 = _Holder(), _Holder(), _Holder(), _Holder()
            if self._two_stream:
                blk.attn.cov_qkv, blk.attn.cov_proj = _Holder(), _Holder()
            self.blocks.append(blk)
        self.norm, self.lm_head = _Holder(), _Holder()
        if self._two_stream:
            self.cov_patch_embed = _PatchEmbedInfo(img_size, patch_size)
            self.cov_lm_head = _Holder()
        if use_shared_rel_pos_bias:
            self.rel_pos_bias = _Holder()
            self.rel_pos_bias.window_size = self.patch_embed.patch_shape
            self.rel_pos_bias.num_relative_distance = (2 * ws - 1) ** 2 + 3
        else:
            self.rel_pos_bias = None
        self._register_views()
        if use_shared_rel_pos_bias:
            self.rel_pos_bias.register_buffer("relative_position_index", relative_position_index(ws))
        self._order_like_reference()
        if self._two_stream:
            self._init_weights_dist()
        else:
            self._init_weights()
        self._engine = None
        self.default_cfg = _cfg()

    # ---- arena plumbing ----
    def _native_config(self, batch):
        return Config(self.img_size, self.patch_embed.patch_size[0], self.in_chans, self.embed_dim, self.depth,
                      self.num_heads, self.mlp_hidden, 1 if getattr(self, "rel_pos_bias", True) is not None else 0,
                      1 if getattr(self, "use_abs_pos_emb", False) else 0,
                      batch, self.ln_eps, self.attn_drop_rate, self.drop_path_rate, 0, 1 if self._two_stream else 0)

    def _owner(self, name):
        mod = self
        parts = name.split(".")
        for p in parts[:-1]:
            mod = mod[int(p)] if p.isdigit() else getattr(mod, p)
        return mod, parts[-1]

    def _register_views(self):
        for name, off, numel, shape, _ in self._layout:
            mod, leaf = self._owner(name)
            mod.register_parameter(leaf, nn.Parameter(self._arena[off:off + numel].view(shape)))

    def _order_like_reference(self):
        """state_dict()/named_parameters() iterate in the reference's registration order
        (ModelEmaV2 zips state-dict VALUES by position, engine_for_cyclical.py:183)."""
        def reorder(mod, order):
            mod._parameters = {k: mod._parameters[k] for k in order if k in mod._parameters}
        reorder(self, ["cls_token", "cov_cls_token", "mask_token", "cov_mask_token", "pos_embed"])
        self._modules = {k: self._modules[k] for k in ("patch_embed", "cov_patch_embed", "rel_pos_bias", "blocks", "norm",
                                                       "lm_head", "cov_lm_head") if k in self._modules}
        for blk in self.blocks:
            reorder(blk, ["gamma_1", "gamma_2"])
            blk._modules = {k: blk._modules[k] for k in ("norm1", "attn", "norm2", "mlp")}
            reorder(blk.attn, ["q_bias", "v_bias", "cov_q_bias", "cov_v_bias"])
            blk.attn._modules = {k: blk.attn._modules[k] for k in ("qkv", "cov_qkv", "proj", "cov_proj") if k in blk.attn._modules}

    def _rebind(self):
        for name, off, numel, shape, _ in self._layout:
            mod, leaf = self._owner(name)
            p = mod._parameters[leaf]
            p.data = self._arena[off:off + numel].view(shape)
            if self._grad_arena is not None:
                p.grad = self._grad_arena[off:off + numel].view(shape)

    def _apply(self, fn, recurse=True):
        new = fn(self._arena)
        if new.dtype != torch.float32:
            raise NotImplementedError("master weights stay fp32; bf16 shadows are managed natively")
        moved = new.device != self._arena.device
        self._arena = new.contiguous()
        if moved:
            self._grad_arena = None
            self._engine = None
        self._rebind()
        for m in self.modules():
            for k, b in m._buffers.items():
                if b is not None:
                    m._buffers[k] = fn(b)
        return self

    def __deepcopy__(self, memo):
        """ModelEmaV2 deep-copies the student (run_cyclical.py:503): rebuild from the constructor
        arguments and copy the arena, so the copy owns an independent flat arena."""
        new = type(self)(**self._ctor)
        dev = self._arena.device
        new._apply(lambda t: t.to(dev))
        new._arena.copy_(self._arena)
        new.train(self.training)
        for p, q in zip(new.parameters(), self.parameters()):
            p.requires_grad_(q.requires_grad)
        new._shadows_stale = True
        memo[id(self)] = new
        return new

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._shadows_stale = True

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.mark_weights_changed()
        return out

    def mark_weights_changed(self):
        """Call after editing parameters outside the native optimizer (bf16 shadows are refreshed lazily)."""
        self._shadows_stale = True

    # ---- init: modeling_cyclical.py:135-161 ----
    def _init_weights(self):
        sd = dict(self.named_parameters())
        with torch.no_grad():
            for name, p in sd.items():
                if name.endswith("gamma_1") or name.endswith("gamma_2"):
                    p.fill_(self.init_values)
                elif "norm" in name and name.endswith("weight"):
                    p.fill_(1.0)
                elif name.endswith("relative_position_bias_table") or name.endswith("bias"):
                    p.zero_()
                elif name.endswith("weight") or name in ("cls_token", "mask_token", "pos_embed"):
                    trunc_normal_(p, std=self.init_std)
            for i in range(self.depth):
                sd[f"blocks.{i}.attn.proj.weight"].div_(math.sqrt(2.0 * (i + 1)))
                sd[f"blocks.{i}.mlp.fc2.weight"].div_(math.sqrt(2.0 * (i + 1)))
        self._shadows_stale = True

    def _init_weights_dist(self):
        """modeling_cyclical_dist.py:62-90: cls tokens trunc-normal (timm default bounds +-2), Linear weights
        trunc-normal(.02), biases 0, LayerNorm (1, 0); mask tokens stay 0 and the patch-embedding convs keep
        nn.Conv2d's default init (neither is touched by the reference's _init_weights); proj / cov_proj / fc2
        rescaled by 1/sqrt(2*(layer+1))."""
        sd = dict(self.named_parameters())
        with torch.no_grad():
            for name, p in sd.items():
                if name.endswith("gamma_1") or name.endswith("gamma_2"):
                    p.fill_(self.init_values)
                elif "norm" in name and name.endswith("weight"):
                    p.fill_(1.0)
                elif "patch_embed" in name:
                    bound = 1.0 / math.sqrt(self.in_chans * self.patch_embed.patch_size[0] ** 2)
                    p.uniform_(-bound, bound)
                elif name.endswith("relative_position_bias_table") or name.endswith("bias") or "mask_token" in name:
                    p.zero_()
                elif name.endswith("weight") or name.endswith("cls_token"):
                    p.normal_(0.0, 0.02).clamp_(-2.0, 2.0)
            for i in range(self.depth):
                for n in ("attn.proj.weight", "attn.cov_proj.weight", "mlp.fc2.weight"):
                    sd[f"blocks.{i}.{n}"].div_(math.sqrt(2.0 * (i + 1)))
        self._shadows_stale = True

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def get_num_layers(self):
        return len(self.blocks)

    # ---- native engine ----
    def engine(self, batch, teacher=None, adam_m=None, adam_v=None):
        e = self._engine
        if e is None or e.batch < batch or (teacher is not None and e.teacher is not teacher) or \
                (adam_m is not None and e.adam_m is not adam_m):
            keep_t = teacher if teacher is not None else (e.teacher if e is not None else None)
            self._engine = e = NativeEngine(self, keep_t, batch, adam_m if adam_m is not None else (e.adam_m if e else None),
                                            adam_v if adam_v is not None else (e.adam_v if e else None))
            self._rebind()
            self._shadows_stale = False
        if getattr(self, "_shadows_stale", False):
            e.sync_shadows(3)
            self._shadows_stale = False
        return e

    def forward_features(self, x, bool_masked_pos, layer_results):
        if not x.is_cuda:
            raise native.UvitError("forward needs a GPU tensor: the HIP path has no CPU fallback")
        B, Cin, H, W = x.shape
        assert H == self.img_size and W == self.img_size, \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size}*{self.img_size})."
        e = self.engine(B)
        x = x.contiguous().float()
        m = None if bool_masked_pos is None else bool_masked_pos.reshape(B, -1).to(torch.int64).contiguous()
        check(lib().uvit_engine_forward_features(e.h, 0, ptr(x), ptr(m), B, 1 if self.training else 0,
                                                 C.c_uint32(int(torch.initial_seed()) & 0xFFFFFFFF),
                                                 C.c_uint32(getattr(self, "_fwd_count", 0)), cur_stream()), "forward_features")
        self._fwd_count = getattr(self, "_fwd_count", 0) + 1
        N = self.patch_embed.num_patches + 1
        if self._two_stream:
            if layer_results == "end":
                return ([e.ws_tensor("x", i + 1, (B, N, self.embed_dim)).clone() for i in range(self.depth)],
                        [e.ws_tensor("x_cov", i + 1, (B, N, self.embed_dim)).clone() for i in range(self.depth)])
            if layer_results:
                raise NotImplementedError("the two-stream model only collects layer_results='end' (modeling_cyclical_dist.py:136-139)")
            return e
        if layer_results == "end":
            return [e.ws_tensor("x", i + 1, (B, N, self.embed_dim)).clone() for i in range(self.depth)]
        if layer_results == "fc":
            return [e.ws_tensor("x", i + 1, (B, N, self.embed_dim)) - e.ws_tensor("xm", i, (B, N, self.embed_dim))
                    for i in range(self.depth)]
        return e

    def forward(self, x, bool_masked_pos, return_all_tokens=False, layer_results=None):
        """Inference-style forward (no autograd graph): training goes through
        engine_for_cyclical.train_one_epoch, whose backward is native as well."""
        out = self.forward_features(x, bool_masked_pos=bool_masked_pos, layer_results=layer_results)
        if layer_results:
            if self._two_stream:
                return [z[:, 1:] for z in out[0]], [z[:, 1:] for z in out[1]]
            return [z[:, 1:] for z in out]
        e, B, P = out, x.shape[0], self.patch_embed.num_patches
        res = []
        for st in range(2 if self._two_stream else 1):       # stream 1: cov_lm_head on the covariance stream
            buf = torch.empty(B * P, self.embed_dim, device=x.device)
            cnt = torch.zeros(1, dtype=torch.int32, device=x.device)
            check(lib().uvit_engine_head(e.h, 2 * st, 1 if return_all_tokens else 0, ptr(buf), ptr(cnt), cur_stream()), "head")
            res.append(buf.view(B, P, self.embed_dim) if return_all_tokens else buf[: int(cnt.item())])
        return tuple(res) if self._two_stream else res[0]


# ---- registry: timm.models.registry / create_model as used at run_cyclical.py:289-302 ----
_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn


def create_model(model_name, pretrained=False, **kwargs):
    if model_name not in _REGISTRY:
        raise RuntimeError(f"Unknown model ({model_name})")
    return _REGISTRY[model_name](pretrained=pretrained, pretrained_cfg=None, pretrained_cfg_overlay=None, **kwargs)


class DistVisionTransformerForCyclicalTraining(VisionTransformerForCyclicalTraining):
    """Two-stream (mean, covariance) model of modeling_cyclical_dist.py:14-165 with Wasserstein attention
    (modeling_finetune_dist.py:61-179).  forward() returns 2-tuples (modeling_cyclical_dist.py:146-165).
    The reference quirks are kept: the covariance QKV uses `attn.qkv.weight`, `attn.cov_qkv.weight` is a dead
    parameter (no gradient, still EMA-averaged), norms / mlp / gammas are shared, no position embedding."""
    _two_stream = True


def _build(pretrained, kwargs, cls=None, **arch):
    kwargs.pop("pretrained_cfg", None)
    kwargs.pop("pretrained_cfg_overlay", None)
    kwargs.pop("num_classes", None)
    init_ckpt = kwargs.pop("init_ckpt", None)
    model = (cls or VisionTransformerForCyclicalTraining)(patch_size=16, mlp_ratio=4, qkv_bias=True,
                                                          norm_layer=partial(nn.LayerNorm, eps=1e-6), **arch, **kwargs)
    if pretrained:
        model.load_state_dict(torch.load(init_ckpt, map_location="cpu")["model"])
    return model


@register_model
def beit_base_patch16_224(pretrained=False, **kwargs):
    """modeling_cyclical.py:282-301."""
    return _build(pretrained, kwargs, embed_dim=768, depth=12, num_heads=12)


@register_model
def dist_beit_base_patch16_224(pretrained=False, **kwargs):
    """modeling_cyclical.py:304-323: the model that returns (mean, cov) pairs, i.e. what `--stochastic` needs."""
    return _build(pretrained, kwargs, cls=DistVisionTransformerForCyclicalTraining, embed_dim=768, depth=12, num_heads=12)


@register_model
def dist_beit_large_patch16_224(pretrained=False, **kwargs):
    """No reference entry point exists; defined by analogy for BASELINE config 5
    (`beit_large_patch16_224 --stochastic`): embed 1024, depth 24, heads 16, two streams."""
    return _build(pretrained, kwargs, cls=DistVisionTransformerForCyclicalTraining, embed_dim=1024, depth=24, num_heads=16)


@register_model
def beit_large_patch16_224(pretrained=False, **kwargs):
    """modeling_cyclical.py:326-343 (the reference entry point is broken under create_model, SURVEY F9;
    this one accepts the same kwargs as the base entry point)."""
    return _build(pretrained, kwargs, embed_dim=1024, depth=24, num_heads=16)
