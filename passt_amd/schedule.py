"""Learning-rate multipliers and weight averaging of the reference's training scripts.

* ``exp_warmup_linear_down(warmup, rampdown_length, start_rampdown, last_value)`` — the per-epoch LR factor of
  ``ex_audioset.py:86-101`` (``get_scheduler_lambda``: 5, 50, 50, 0.01), restating ``helpers/ramp.py:37-47`` (exp
  ramp-up, arXiv 1610.02242), ``:61-70`` (linear ramp-down) and ``:93-98`` (their product).
* ``SWA`` — ``helpers/swa_callback.py:246-268`` (``update_parameters`` + ``avg_fn``) on the flat parameter
  buffer of ``TrainStep``: one fused kernel instead of a Python loop over 159 tensors, and
  ``copy_to(net)`` for evaluation with the averaged weights (``transfer_weights`` :217-219).
"""
import math

import torch

from . import ops


def exp_rampup(length):
    def f(epoch):
        if epoch >= length:
            return 1.0
        e = min(max(epoch, 0.5), length)          # the reference clips to [0.5, length]: epoch 0 is not exp(-5)
        return float(math.exp(-5.0 * (1.0 - e / length) ** 2))
    return f


def linear_rampdown(length, start=0, last_value=0.0):
    def f(epoch):
        if epoch <= start:
            return 1.0
        if epoch - start < length:
            return last_value + (1.0 - last_value) * (length - epoch + start) / length
        return last_value
    return f


def exp_warmup_linear_down(warmup, rampdown_length, start_rampdown, last_value):
    up, down = exp_rampup(warmup), linear_rampdown(rampdown_length, start_rampdown, last_value)
    return lambda epoch: up(epoch) * down(epoch)


class SWA:
    """Running equal-weight average of a TrainStep's parameters (f32, same flat layout)."""

    def __init__(self, train_step):
        self.ts = train_step
        self.avg = torch.empty_like(train_step.flat_p)
        self.n_averaged = 0

    def update(self):
        ops.swa_update(self.avg, self.ts.flat_p, self.n_averaged)
        self.n_averaged += 1

    def copy_to(self, net):
        """Load the averaged weights into ``net`` (a deepcopy of the trained module, or the module itself)."""
        off = 0
        with torch.no_grad():
            for name, p in self.ts.named:
                k = p.numel()
                dict(net.named_parameters())[name].copy_(self.avg[off:off + k].view(p.shape))
                off += k
        if hasattr(net, "mark_params_updated"):
            net.mark_params_updated()
        return net
