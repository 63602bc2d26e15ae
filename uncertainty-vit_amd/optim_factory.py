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

    # ---- checkpoint interchange with the reference (utils.py:462-479 saves optimizer.state_dict() of a torch.optim.AdamW
    # built over these two groups): same structure both ways --
    #   'state': {param index: {'step', 'exp_avg', 'exp_avg_sq'}}, indices running over group 0 (decay) then group 1
    #   (no_decay) in named_parameters() order; 'param_groups': hyper-parameters + 'params': [indices].
    # Parameters that never receive a gradient (the two-stream model's dead attn.cov_qkv.weight) have no 'state' entry,
    # exactly as in torch (state is created on the first step with a gradient).
    def _index(self):
        lay = {n: (o, k, shape, d) for n, o, k, shape, d in self.model._layout}
        order = [n for grp in ("decay", "no_decay") for n in self.group_names[grp]]
        return lay, order

    def state_dict(self):
        self._ensure_state()
        lay, order = self._index()
        state, i0 = {}, 0
        groups = []
        for g, grp in zip(self.param_groups, ("decay", "no_decay")):
            names = self.group_names[grp]
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "amsgrad": False, "maximize": False, "foreach": None,
                           "capturable": False, "differentiable": False, "fused": None,
                           "params": list(range(i0, i0 + len(names)))})
            i0 += len(names)
        if self.step_count > 0:
            for i, n in enumerate(order):
                off, numel, shape, frozen = lay[n]
                if frozen == 2:
                    continue
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + numel].view(shape).cpu().clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + numel].view(shape).cpu().clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self._ensure_state()
        st = sd["state"]
        if "exp_avg" in st:                    # round-1 files of this code base: flat arenas
            self.step_count = int(st["step"])
            self.exp_avg.copy_(st["exp_avg"])
            self.exp_avg_sq.copy_(st["exp_avg_sq"])
        else:                                  # torch.optim.AdamW layout (the reference's checkpoints)
            lay, order = self._index()
            n_expected = sum(len(g["params"]) for g in sd["param_groups"])
            if n_expected != len(order):
                raise ValueError(f"optimizer state has {n_expected} parameters, the model has {len(order)}")
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            steps = set()
            for i, n in enumerate(order):
                e = st.get(i, st.get(str(i)))
                if e is None:
                    continue
                off, numel, shape, _ = lay[n]
                self.exp_avg[off:off + numel].view(shape).copy_(e["exp_avg"])
                self.exp_avg_sq[off:off + numel].view(shape).copy_(e["exp_avg_sq"])
                steps.add(int(float(e["step"])))
            if len(steps) > 1:
                raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused step keeps one count")
            self.step_count = steps.pop() if steps else 0
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s_.items() if k in ("lr", "betas", "eps", "weight_decay", "lr_scale")})
            g["betas"] = tuple(g["betas"])


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    opt = args.opt.lower().split("_")[-1]
    if opt != "adamw":
        raise NotImplementedError(f"--opt {args.opt}: only adamw is on the pre-training path (README.md:11-25)")
    betas = tuple(args.opt_betas) if getattr(args, "opt_betas", None) else (0.9, 0.999)
    eps = args.opt_eps if getattr(args, "opt_eps", None) is not None else 1e-8
    return ArenaAdamW(model, lr=args.lr, weight_decay=args.weight_decay, betas=betas, eps=eps)
