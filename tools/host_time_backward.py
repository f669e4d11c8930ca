"""Host time of the SAME passt_backward call in its two callers: TrainStep.step (trainer thread) and _PasstFunction.backward
(autograd's device thread), c5 shapes (B = 12: the host matters).  Wraps passt_amd.passt.passt_backward with a timer.
    python tools/host_time_backward.py"""
import os
import sys
import threading
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import passt_amd  # noqa: E402
from passt_amd import passt as P  # noqa: E402
from passt_amd import train as T  # noqa: E402

acc = {}
orig = P.passt_backward


def timed(*a, **k):
    t0 = time.perf_counter()
    r = orig(*a, **k)
    dt = time.perf_counter() - t0
    key = threading.current_thread().name
    n, s = acc.get(key, (0, 0.0))
    acc[key] = (n + 1, s + dt)
    return r


P.passt_backward = timed
T.passt_backward = timed
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c5"
cfg = bench.CONFIGS[cfgname]
dev = torch.device("cuda", 0)
for path in ("trainstep", "autograd"):
    torch.manual_seed(0)
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = passt_amd.get_model(arch=cfg["arch"], pretrained=False, n_classes=cfg["n_classes"], **cfg["net_kw"]).to(dev).train()
        mel = passt_amd.AugmentMelSTFT(n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, fmin=0.0, fmax=None,
                                       fmin_aug_range=10, fmax_aug_range=2000, **cfg["mel_kw"]).to(dev).train()
        if path == "autograd":
            ts = bench.AutogradStep(net, mel, lr=2e-5, weight_decay=1e-4, loss=cfg["loss"], mixup_alpha=0.3, precision="bf16",
                                    comm_dtype="fp32", transport="torch")
        else:
            net.precision = "bf16"
            ts = T.TrainStep(net, mel, lr=2e-5, weight_decay=1e-4, loss=cfg["loss"])
        B = cfg["batch"]
        x = (torch.rand(B, 1, cfg["clip"], device=dev) * 2 - 1) * 0.1
        y = (torch.rand(B, cfg["n_classes"], device=dev) < 2.7 / 527).float() if cfg["loss"] == "bce" else torch.randint(0, cfg["n_classes"], (B,), device=dev)
        for _ in range(5):
            ts.step(x, y)
        torch.cuda.synchronize()
        acc.clear()
        t0 = time.perf_counter()
        for _ in range(20):
            ts.step(x, y)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{cfgname} {path}: enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, total {1e3 * (t2 - t0) / 20:.2f} ms/step; passt_backward host time by thread: "
          + ", ".join(f"{k}: {1e3 * s / n:.2f} ms x {n}" for k, (n, s) in acc.items()), flush=True)
    del ts, net, mel
