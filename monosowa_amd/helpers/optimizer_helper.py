"""Optimizer construction and the reference's AdamW variant, evaluated with multi-tensor kernels.

Reference: lib/helpers/optimizer_helper.py -- ``build_optimizer`` :7-27 (biases get weight_decay 0),
``AdamW.step`` :69-129.  The update is NOT torch.optim.AdamW: eps is added to sqrt(v) before the bias
correction and the decay term is scaled by the bias-corrected step size:

    m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  denom = sqrt(v) + eps
    step_size = lr * sqrt(1-b2^t) / (1-b1^t)
    p = p - step_size * (wd*p + m/denom)

The reference loops over ~580 parameters in Python (about six tiny launches each); here the same
element-wise operations, in the same order, are issued through ``torch._foreach_*`` so a step is a
handful of launches.  State keys ('step','exp_avg','exp_avg_sq') match for checkpoint exchange.
"""
import math

import torch
import torch.optim as optim
from torch.optim.optimizer import Optimizer


def build_optimizer(cfg_optimizer, model):
    weights, biases = [], []
    for name, param in model.named_parameters():
        (biases if "bias" in name else weights).append(param)
    parameters = [{"params": biases, "weight_decay": 0},
                  {"params": weights, "weight_decay": cfg_optimizer["weight_decay"]}]
    kind = cfg_optimizer["type"]
    if kind == "sgd":
        return optim.SGD(parameters, lr=cfg_optimizer["lr"], momentum=0.9)
    if kind == "adam":
        return optim.Adam(parameters, lr=cfg_optimizer["lr"])
    if kind == "adamw":
        return AdamW(parameters, lr=cfg_optimizer["lr"])
    raise NotImplementedError("%s optimizer is not supported" % kind)


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameters: {}".format(betas))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("amsgrad", False)

    def _fused_step(self, group, step, params, grads, exp_avgs, exp_avg_sqs):
        """All parameters of the group in one HIP launch (monosowa_amd/csrc/pointwise.hip adamw_kernel) when they are
        dense contiguous float32 GPU tensors; same operations in the same order as the foreach formulation below."""
        # element-wise over storage order: any dense layout works (channels_last convolution weights included) as long
        # as the parameter, its gradient and both moments share it
        def dense(t):
            return t.is_cuda and t.dtype == torch.float32 and (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)))
        if group["amsgrad"] or not params or not all(
                dense(p) and p.stride() == g.stride() == m.stride() == v.stride() and g.dtype == torch.float32 and g.is_cuda
                for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs)):
            return False
        from ..pointwise import FusedAdamWPlan
        plans = self.__dict__.setdefault("_fused_plans", {})
        key = id(group["params"])
        plan = plans.get(key)
        if plan is None or len(plan.keys[0]) != len(params) or not plan.matches(params, exp_avgs, exp_avg_sqs):
            plan = plans[key] = FusedAdamWPlan(params, exp_avgs, exp_avg_sqs, group["weight_decay"])
        beta1, beta2 = group["betas"]
        step_size = group["lr"] * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
        plan.step(grads, beta1, beta2, group["eps"], step_size)
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            buckets = {}      # step count -> lists (all equal in practice: one bucket)
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)
                    if group["amsgrad"]:
                        state["max_exp_avg_sq"] = torch.zeros_like(p)
                state["step"] += 1
                b = buckets.setdefault(int(state["step"]), ([], [], [], [], []))
                b[0].append(p)
                b[1].append(p.grad)
                b[2].append(state["exp_avg"])
                b[3].append(state["exp_avg_sq"])
                if group["amsgrad"]:
                    b[4].append(state["max_exp_avg_sq"])
            for step, (params, grads, exp_avgs, exp_avg_sqs, max_sqs) in buckets.items():
                if self._fused_step(group, step, params, grads, exp_avgs, exp_avg_sqs):
                    continue
                torch._foreach_mul_(exp_avgs, beta1)
                torch._foreach_add_(exp_avgs, grads, alpha=1 - beta1)
                torch._foreach_mul_(exp_avg_sqs, beta2)
                torch._foreach_addcmul_(exp_avg_sqs, grads, grads, value=1 - beta2)
                if group["amsgrad"]:
                    torch._foreach_maximum_(max_sqs, exp_avg_sqs)
                    denom = torch._foreach_sqrt(max_sqs)
                else:
                    denom = torch._foreach_sqrt(exp_avg_sqs)
                torch._foreach_add_(denom, group["eps"])
                step_size = group["lr"] * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
                update = torch._foreach_mul(params, group["weight_decay"])
                torch._foreach_addcdiv_(update, exp_avgs, denom, value=1)
                torch._foreach_add_(params, update, alpha=-step_size)
        return loss
