"""Host-side cost of one training step: time to ENQUEUE a step (no sync) vs the steady-state step time, for the headline
configuration (c2: B = 64, 10 s clips) and the ESC-50 one (c5: B = 12, 5 s clips, launch bound).
Run 71 (r01): 4.6 ms of host time per 25.5 ms step -- the launch path (ctypes, ~450 launches) is not the bottleneck at B = 64."""
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import passt_amd  # noqa: E402
from passt_amd.train import TrainStep  # noqa: E402

dev = "cuda"
for name, kw, B, L, ncls, loss in (("c2", dict(s_patchout_t=40, s_patchout_f=4), 64, 320000, 527, "bce"),
                                   ("c5", dict(s_patchout_t=10, s_patchout_f=3), 12, 160000, 50, "ce")):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = passt_amd.get_model(arch="passt_s_swa_p16_128_ap476", pretrained=False, n_classes=ncls, **kw).to(dev).train()
        mel = passt_amd.AugmentMelSTFT(fmin_aug_range=10, fmax_aug_range=2000).to(dev).train()
    net.precision = "bf16"
    ts = TrainStep(net, mel, lr=2e-5, weight_decay=1e-4, loss=loss)
    x = (torch.rand(B, 1, L, device=dev) * 2 - 1) * 0.1
    y = (torch.rand(B, ncls, device=dev) < 0.005).float() if loss == "bce" else torch.randint(0, ncls, (B,), device=dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(5):
            ts.step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ts.step(x, y)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{name}: enqueue {1e3 * (t1 - t0) / 10:.2f} ms/step, total {1e3 * (t2 - t0) / 10:.2f} ms/step", flush=True)
    del ts, net, mel
