"""Host-side cost of one training step: time to ENQUEUE a step (no sync) vs the steady-state step time.
Run 71: 4.6 ms of host time per 25.5 ms step -- the launch path (ctypes, ~450 launches) is not the bottleneck."""
import sys, time, warnings, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import passt_amd
from passt_amd.train import TrainStep
dev='cuda'
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    net = passt_amd.get_model(arch="passt_s_swa_p16_128_ap476", pretrained=False, s_patchout_t=40, s_patchout_f=4).to(dev).train()
    mel = passt_amd.AugmentMelSTFT(fmin_aug_range=10, fmax_aug_range=2000).to(dev).train()
net.precision='bf16'
ts = TrainStep(net, mel, lr=2e-5, weight_decay=1e-4)
x=(torch.rand(64,1,320000,device=dev)*2-1)*0.1; y=(torch.rand(64,527,device=dev)<0.005).float()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(5): ts.step(x,y)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(10): ts.step(x,y)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
