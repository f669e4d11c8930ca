"""``my_mixup`` for the drop-in (autograd) path: the reference's function with its results already on the device.

The reference (helpers/mixup.py:5-12) returns a CPU permutation and a CPU ``lam``; its callers then do three synchronising
pageable host->device copies per training step (ex_audioset.py:171-183, ex_esc50.py:152-161)::

    rn_indices, lam = my_mixup(batch_size, self.mixup_alpha)
    lam = lam.to(x.device)                                  # pageable H2D copy, host waits
    x = x * lam.reshape(...) + x[rn_indices] * (1. - ...)   # index tensor uploaded, host waits
    ...
    y_mix = y * ... + y[rn_indices] * ...                   # index tensor uploaded AGAIN, after the forward was enqueued

Each of them leaves the GPU idle while the host restarts (0.7-1.7 ms per step measured on MI355X,
profiles/r04_autograd_vs_trainstep.md).  The replacement is a one-word import change::

    -from helpers.mixup import my_mixup
    +from passt_amd.mixup import my_mixup

Same signature for the existing call sites, the same RNG calls in the same order (``torch.randperm(size)`` on torch's CPU
generator, then ``np.random.beta(alpha, alpha, size)``), the same values bit for bit (tests/test_caller_flow_cpu.py pins them to
the live reference function) -- but both results are device tensors uploaded through the library's page-locked staging ring
(asynchronous copies): ``lam.to(x.device)`` is then a no-op and ``x[rn_indices]`` / ``y[rn_indices]`` index with a device
tensor, so the step has no host synchronisation left.

CONSTRAINT (the one way this differs from the reference function): with a HIP device visible the results are DEVICE tensors,
on ``torch.cuda.current_device()`` unless ``device=`` says otherwise.  A caller that indexes CPU tensors with them
(data-loader-side mixup, ``x_cpu[rn_indices]``) or that never selected its device (``torch.cuda.set_device``; Lightning does)
must pass ``device="cpu"`` / ``device=x.device``, or set ``PASST_AMD_MIXUP_DEVICE=cpu`` in the environment -- then this IS the
reference function, CPU results and all.
"""
import os

import numpy as np
import torch

from . import ops


def my_mixup(size, alpha, device=None):
    """helpers/mixup.py:5-12.  ``device``: where the results go; None = the current HIP device when one is visible (one process
    per GPU: the device Lightning / the caller selected), else the CPU -- then this IS the reference function."""
    # the reference's two draws, in its order: torch's CPU generator for the permutation, numpy's global generator for the Beta
    # samples (helpers/mixup.py:6-7); lam = max(l, 1 - l) in float32 (:8: concatenate([l, 1 - l]).max(1) is that element-wise max)
    rn_indices = torch.randperm(size)
    beta = np.random.beta(alpha, alpha, size).astype(np.float32)
    lam = torch.from_numpy(np.maximum(beta, np.float32(1.0) - beta))
    if device is None and os.environ.get("PASST_AMD_MIXUP_DEVICE"):
        device = os.environ["PASST_AMD_MIXUP_DEVICE"]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    if device.type == "cpu":
        return rn_indices, lam
    return ops.upload_small(rn_indices, device), ops.upload_small(lam, device)
