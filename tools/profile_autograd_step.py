"""Where the drop-in (autograd) path spends host and device time beyond TrainStep: torch.profiler over a few steps of
bench.py's AutogradStep (c2 or c5), operator table sorted by host time and by launch count.
    python tools/profile_autograd_step.py [c2|c5]"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import passt_amd  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
np.random.seed(0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    net = passt_amd.get_model(arch=cfg["arch"], pretrained=False, n_classes=cfg["n_classes"], **cfg["net_kw"]).to(dev).train()
    mel = passt_amd.AugmentMelSTFT(n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, fmin=0.0, fmax=None,
                                   fmin_aug_range=10, fmax_aug_range=2000, **cfg["mel_kw"]).to(dev).train()
    ts = bench.AutogradStep(net, mel, lr=2e-5, weight_decay=1e-4, loss=cfg["loss"], mixup_alpha=0.3, precision="bf16",
                            comm_dtype="fp32", transport="torch")
    B = cfg["batch"]
    x = (torch.rand(B, 1, cfg["clip"], device=dev) * 2 - 1) * 0.1
    y = (torch.rand(B, cfg["n_classes"], device=dev) < 2.7 / 527).float() if cfg["loss"] == "bce" else torch.randint(0, cfg["n_classes"], (B,), device=dev)
    for _ in range(4):
        ts.step(x, y)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile, record_function
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            with record_function("STEP"):
                ts.step(x, y)
        torch.cuda.synchronize()
ka = prof.key_averages()
print(ka.table(sort_by="self_cpu_time_total", row_limit=35, max_name_column_width=60))
print(ka.table(sort_by="count", row_limit=30, max_name_column_width=60))
