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
    """Running equal-weight average of a TrainStep's parameters (f32, same flat layout).

    ``update()`` is ``update_parameters`` + ``avg_fn`` (helpers/swa_callback.py:246-268).  ``on_train_epoch_start(epoch,
    max_epochs)`` is the callback's schedule (:126-132, :161-197) for callers that drive epochs themselves: averaging
    starts at the 0-based epoch ``max(swa_epoch_start - 1, 0)`` (a float ``swa_epoch_start`` is a fraction of ``max_epochs``,
    :153-154), repeats every ``swa_freq`` epochs up to the last one, and the count restarts from zero whenever the start epoch
    is entered; defaults = ex_audioset.py:323-324 (50, 5).  Returns True when the epoch's snapshot was averaged in -- the epochs
    on which the reference also validates the averaged network (``do_swa``, :208-214)."""

    def __init__(self, train_step, swa_epoch_start=50, swa_freq=5):
        if isinstance(swa_epoch_start, int) and swa_epoch_start < 1 or \
                isinstance(swa_epoch_start, float) and not (0 <= swa_epoch_start <= 1):
            raise ValueError("swa_epoch_start should be a >0 integer or a float between 0 and 1.")     # :98-103
        self.ts = train_step
        self.avg = torch.empty_like(train_step.flat_p)
        self.n_averaged = 0
        self.swa_epoch_start, self.swa_freq = swa_epoch_start, swa_freq

    def update(self):
        ops.swa_update(self.avg, self.ts.flat_p, self.n_averaged)
        self.n_averaged += 1

    def swa_start(self, max_epochs):
        s = self.swa_epoch_start
        if isinstance(s, float):
            s = int(max_epochs * s)
        return max(s - 1, 0)

    def on_train_epoch_start(self, epoch, max_epochs):
        start = self.swa_start(max_epochs)
        if epoch == start:
            self.n_averaged = 0
        if start <= epoch <= max_epochs - 1 and (epoch - start) % self.swa_freq == 0:
            self.update()
            return True
        return False

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
