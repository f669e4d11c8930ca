#!/usr/bin/env python
"""Inference-path throughput: eval-mode forward, no Patchout (N = 1190 tokens), bf16, mel front end
included.  Not the headline metric (bench.py is); prints one JSON line for DESIGN.md.

    python tools/bench_eval.py [--batch 64] [--iters 10]
"""
import argparse, json, os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import passt_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = passt_amd.get_model(arch="passt_s_swa_p16_128_ap476", pretrained=False, n_classes=527, s_patchout_t=0,
                                  s_patchout_f=0).to(dev).eval()
        mel = passt_amd.AugmentMelSTFT(fmin_aug_range=10, fmax_aug_range=2000).to(dev).eval()
    net.precision = "bf16"
    wave = (torch.rand(a.batch, 320000, device=dev) * 2 - 1) * 0.1

    def step():
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return net(mel(wave).unsqueeze(1))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(json.dumps({"metric": "clips/s (10s@32k) eval forward passt_s, N=1190, mel included", "value": round(a.batch / ms * 1e3, 1),
                      "ms_per_batch": round(ms, 3), "batch": a.batch, "dtype": "bf16",
                      "algorithmic_gflop_per_clip": 254.81}))


if __name__ == "__main__":
    main()
