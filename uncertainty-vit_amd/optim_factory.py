"""Optimizer wiring of the pre-training path (reference optim_factory.py:58-97,100-134).

`create_optimizer` returns an `ArenaAdamW`: it keeps the reference's two param groups
(decay / no_decay, each with `lr`, `weight_decay`, `lr_scale`) so train_one_epoch can rewrite
them every step, but the update itself is ONE fused HIP kernel over the flat parameter arena
(whose [decay | no-decay] layout mirrors these groups)."""
import json

import torch


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None):
    if get_num_layer is not None or get_layer_scale is not None:
        raise NotImplementedError("layer-wise lr decay belongs to fine-tuning (out of scope)")
    groups = {"decay": {"weight_decay": weight_decay, "params": [], "lr_scale": 1.0},
              "no_decay": {"weight_decay": 0.0, "params": [], "lr_scale": 1.0}}
    names = {"decay": [], "no_decay": []}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g = "no_decay" if (p.ndim == 1 or name.endswith(".bias") or name in skip_list) else "decay"
        groups[g]["params"].append(p)
        names[g].append(name)
    print("Param groups = %s" % json.dumps({k: {"weight_decay": groups[k]["weight_decay"], "params": v,
                                                "lr_scale": 1.0} for k, v in names.items()}, indent=2))
    return [groups["decay"], groups["no_decay"]], names


class ArenaAdamW:
    """torch.optim.AdamW semantics (lr, betas, eps, decoupled weight decay) on the flat arena."""

    def __init__(self, model, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
        skip = model.no_weight_decay() if hasattr(model, "no_weight_decay") else ()
        groups, self.group_names = get_parameter_groups(model, weight_decay, skip)
        # the arena layout must agree with the reference's grouping rule
        # decay flag 2 = parameters whose .grad is always None in the reference (two-stream attn.cov_qkv.weight):
        # nominally in the decay group, never stepped by torch's AdamW; the arena keeps them constant
        arena_decay = {n for n, _, _, _, d in model._layout if d}
        assert arena_decay == set(self.group_names["decay"]), "arena decay region != optim_factory decay group"
        self.model = model
        self.param_groups = groups
        for g in self.param_groups:
            g.update(lr=lr, betas=tuple(betas), eps=eps)
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None

    def _ensure_state(self):
        if self.exp_avg is None or self.exp_avg.device != self.model._arena.device:
            self.exp_avg = torch.zeros_like(self.model._arena)
            self.exp_avg_sq = torch.zeros_like(self.model._arena)

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @property
    def weight_decay(self):
        return self.param_groups[0]["weight_decay"]

    def zero_grad(self, set_to_none=True):
        pass   # the native step overwrites / re-zeros the gradient arena itself

    def state_dict(self):
        self._ensure_state()
        return {"state": {"step": self.step_count, "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu()},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self._ensure_state()
        self.step_count = int(sd["state"]["step"])
        self.exp_avg.copy_(sd["state"]["exp_avg"])
        self.exp_avg_sq.copy_(sd["state"]["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    opt = args.opt.lower().split("_")[-1]
    if opt != "adamw":
        raise NotImplementedError(f"--opt {args.opt}: only adamw is on the pre-training path (README.md:11-25)")
    betas = tuple(args.opt_betas) if getattr(args, "opt_betas", None) else (0.9, 0.999)
    eps = args.opt_eps if getattr(args, "opt_eps", None) is not None else 1e-8
    return ArenaAdamW(model, lr=args.lr, weight_decay=args.weight_decay, betas=betas, eps=eps)
