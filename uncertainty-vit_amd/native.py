"""ctypes binding of libuvit.so (the C ABI in include/uvit.h).

The product path has NO CPU fallback: `lib()` raises if the HIP library is missing, and every
wrapper raises `UvitError` on a non-zero return code.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuvit.so")
MAX_DEPTH = 64


class UvitError(RuntimeError):
    pass


_ERR = {-1: "bad argument", -2: "unsupported shape", -3: "HIP launch/runtime error", -4: "workspace too small"}


def check(rc, what=""):
    if rc != 0:
        raise UvitError(f"libuvit: {what} failed: {_ERR.get(rc, rc)} ({rc})")


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("img_size", "patch_size", "in_chans", "embed_dim", "depth", "num_heads",
                                          "mlp_hidden", "use_shared_rel_pos_bias", "use_abs_pos_emb", "batch")] + \
               [("ln_eps", C.c_float), ("attn_drop_rate", C.c_float), ("drop_path_rate", C.c_float),
                ("bias_chunk", C.c_int32), ("two_stream", C.c_int32)]


class LayoutEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("offset", C.c_int64), ("numel", C.c_int64), ("ndim", C.c_int32),
                ("decay", C.c_int32), ("shape", C.c_int64 * 4)]


class Buffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("params", "grads", "adam_m", "adam_v", "ema", "params_bf16",
                                           "params_bf16_t", "ema_bf16", "rel_index", "workspace")] + \
               [("workspace_bytes", C.c_int64)]


class StepParams(C.Structure):
    _fields_ = [("target_layers", C.c_int32 * MAX_DEPTH), ("n_target_layers", C.c_int32),
                ("target_layer_norm_last", C.c_int32), ("post_target_layer_norm", C.c_int32), ("l2_loss", C.c_int32),
                ("l1_beta", C.c_float), ("loss_scale", C.c_float), ("clip_grad", C.c_float), ("lr", C.c_float),
                ("weight_decay", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("opt_step", C.c_int32), ("ema_decay", C.c_float), ("do_ema", C.c_int32), ("grad_scale", C.c_float),
                ("seed", C.c_uint32), ("it", C.c_uint32), ("train_dropout", C.c_int32), ("lambda_pretraining", C.c_float),
                ("sched_dev", C.c_void_p), ("sched_len", C.c_int32), ("sched_index", C.c_int32),
                ("layer_results_fc", C.c_int32), ("var_w0", C.c_float), ("var_margin0", C.c_float),
                ("target_batch_norm", C.c_int32), ("target_instance_norm", C.c_int32), ("post_target_instance_norm", C.c_int32),
                ("n_rows_hint", C.c_int32)]


class WgradProblem(C.Structure):
    """uvit_wgrad_problem (include/uvit.h)."""
    _fields_ = [("Y", C.c_void_p), ("X", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("bias2", C.c_void_p),
                ("bias_end", C.c_int32), ("bias2_begin", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("ldy", C.c_int32), ("ldx", C.c_int32), ("ldc", C.c_int32)]


class GemmEpilogue(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("out", "out2", "bias", "bias2", "gamma", "resid", "rowscale", "aux", "mask",
                                           "mask_token")] + [("ldo", C.c_int32), ("tokens", C.c_int32), ("patches", C.c_int32),
                                                             ("row0", C.c_int32)]


class Tuning(C.Structure):
    """uvit_tuning (include/uvit.h): launch tuning passed per call / held by the engine."""
    _fields_ = [("nt_variant", C.c_int32), ("tn_variant", C.c_int32), ("tn_split_target", C.c_int32),
                ("wgrad_group_chunks", C.c_int32), ("nt_group", C.c_int32), ("nt_persist", C.c_int32)]

    @classmethod
    def default(cls, **over):
        t = cls()
        lib().uvit_tuning_default(C.byref(t))
        for k, v in over.items():
            setattr(t, k, v)
        return t


_vp, _i, _i64, _f, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
# prototypes of every symbol include/uvit.h declares (argtypes matter: int64 / float arguments)
_PROTOTYPES = {
    "uvit_version": (_i, []),
    "uvit_source_hash": (C.c_char_p, []),
    "uvit_layout_count": (_i, [_vp]),
    "uvit_layout_get": (_i, [_vp, _i, _vp]),
    "uvit_arena_numel": (_i64, [_vp, _vp]),
    "uvit_workspace_bytes": (_i64, [_vp]),
    "uvit_engine_create": (_vp, [_vp, _vp, _vp, _vp]),
    "uvit_engine_destroy": (None, [_vp]),
    "uvit_engine_sync_shadows": (_i, [_vp, _i, _vp]),
    "uvit_engine_forward_features": (_i, [_vp, _i, _vp, _vp, _i, _i, _u32, _u32, _vp]),
    "uvit_engine_head": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "uvit_engine_ws_ptr": (_vp, [_vp, C.c_char_p, _i]),
    "uvit_engine_compact_rows": (_i, [_vp]),
    "uvit_step_begin": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "uvit_step_backward_layer": (_i, [_vp, _i, _vp, _vp]),
    "uvit_step_backward_embed": (_i, [_vp, _vp]),
    "uvit_step_wait_layer_grads": (_i, [_vp, _i, _vp]),
    "uvit_step_update": (_i, [_vp, _vp, _vp]),
    "uvit_train_step": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "uvit_engine_read_stats": (_i, [_vp, _vp, _vp]),
    "uvit_engine_read_stats_async": (_i, [_vp, _vp, _vp]),
    "uvit_tuning_default": (None, [_vp]),
    "uvit_engine_set_tuning": (_i, [_vp, _vp]),
    "uvit_op_wgrad_group": (_i, [_vp, _i, _vp, _vp]),
    "uvit_engine_set_streams": (_i, [_vp, _i]),
    "uvit_engine_set_drop_path_rows": (_i, [_vp, _i]),
    "uvit_drop_path_kept_counts": (_i, [_i, _f, _i, _i, _u32, _u32, _vp]),
    "uvit_engine_profile": (_i, [_vp, _i, _i]),
    "uvit_engine_profile_read": (_i, [_vp, _vp, _vp, _vp]),
    "uvit_engine_profile_read_kind": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "uvit_op_gemm_nt": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "uvit_op_gemm_nt_tuned": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "uvit_op_gemm_nt_sched": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "uvit_op_gemm_tn": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "uvit_op_wasserstein_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "uvit_op_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _u32, _u32, _vp]),
    "uvit_op_attn_bwd_ws_bytes": (_i64, [_i, _i, _i]),
    "uvit_op_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _f, _u32, _u32, _vp]),
    "uvit_op_attn2_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _u32, _u32, _vp]),
    "uvit_op_attn2_bwd_ws_bytes": (_i64, [_i, _i, _i]),
    "uvit_op_attn2_bwd": (_i, [_vp] * 12 + [_i, _vp, _i, _i, _i, _i, _f, _f, _u32, _u32, _vp]),
    "uvit_op_relpos_gather": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "uvit_op_relpos_scatter": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "uvit_op_ln_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "uvit_op_ln_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "uvit_op_ema": (_i, [_vp, _vp, _vp, _i64, _f, _vp]),
    "uvit_op_sumsq": (_i, [_vp, _i64, _vp, _vp]),
    "uvit_op_adamw": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _vp, _vp]),
    "uvit_op_smooth_l1": (_i, [_vp, _vp, _vp, _f, _i, _f, _vp, _vp, _i, _i, _vp]),
    "uvit_op_target_accum": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "uvit_op_variance_loss": (_i, [_vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "uvit_op_target_finalize": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "uvit_op_mask_compact": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "uvit_op_im2col": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "uvit_op_synth_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _u32, _u32, _vp]),
    "uvit_op_droppath": (_i, [_vp, _vp, _i, _i, _u32, _u32, _vp]),
    "uvit_op_cast_bf16": (_i, [_vp, _vp, _i64, _vp]),
}
SYMBOLS = list(_PROTOTYPES)

_lib = None


def source_hash():
    """Fingerprint of csrc/{*.hip,*.h} + include/uvit.h as csrc/build.sh computes it."""
    import hashlib
    src = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for f in sorted(n for n in os.listdir(src) if n.endswith((".hip", ".h"))):
        with open(os.path.join(src, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(_HERE, "..", "include", "uvit.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False):
    """Compile csrc/*.hip for gfx950 into libuvit.so (hipcc cross-compiles without a GPU)."""
    if os.path.exists(LIB_PATH) and not force:
        # build.sh leaves the fingerprint beside the library (dlopen-ing the old file here would pin it in this process)
        try:
            with open(LIB_PATH + ".hash") as fh:
                if fh.read().strip() == source_hash():
                    return LIB_PATH
        except OSError:
            pass
    subprocess.run(["bash", os.path.join(_HERE, "csrc", "build.sh"), LIB_PATH], check=True)
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UvitError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # A/B runs of two builds in one gpurun call (tools/ab.sh "UVIT_LIB_AB=uncertainty-vit_amd/libuvit_prev.so" ...): load another build of
    # the same ABI instead; the staleness check below does not apply to it
    alt = os.environ.get("UVIT_LIB_AB")
    L = C.CDLL(os.path.abspath(alt) if alt else LIB_PATH)
    for name, (res, args) in _PROTOTYPES.items():
        f = getattr(L, name)          # AttributeError here = a symbol of include/uvit.h is missing
        f.restype, f.argtypes = res, args
    if L.uvit_version() != 100:
        raise UvitError("libuvit version mismatch")
    built, want = L.uvit_source_hash().decode(), source_hash()
    if built != want and not alt:
        # never auto-build here: lib() runs inside profiled / multi-rank processes
        raise UvitError(f"{LIB_PATH} is stale: built from sources {built}, on disk {want}. "
                        "Run `python -c 'import __graft_entry__ as g; g.build()'`.")
    _lib = L
    return L


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def cur_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32(x):
    return C.c_float(float(x))
