"""Worker of tests/test_gpu_ddp.py: one process = one data-parallel rank of the REAL training step (passt_amd.train.TrainStep
on the HIP kernels, passt_backward's per-block bucket callbacks, per-bucket optimizer launches) -- every rank on the single
GPU of the test box, transport "gloo" on device tensors (RCCL refuses two ranks on one device; the reducer code path is
the same, only the backend string differs).

    python tests/ddp_worker.py --out ref.pt                       single process, batch = both halves concatenated
    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/ddp_worker.py --out dp.pt [--comm-dtype bf16]
"""
import argparse
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import passt_amd  # noqa: E402
from oracle import detgen  # noqa: E402
from passt_amd.train import TrainStep  # noqa: E402
from tests.golden import make_golden as G  # noqa: E402


class _AutogradRunner:
    """The drop-in path as ex_audioset.py drives it (:179-186, 104-109): net(x) -> BCE mean -> loss.backward() -> torch
    optimizer; data parallel through passt_amd.ddp.attach (the node reduces its own buckets from inside the backward) or
    through torch's own DistributedDataParallel.  `flat_p` mirrors TrainStep's flat parameter buffer (same order) so the
    test can compare the two paths entry by entry."""

    def __init__(self, net, args, lr, world, dev):
        from passt_amd import ddp
        self.net, self.fwd = net, net
        if args.path == "attach":
            self.red = ddp.attach(net, comm_dtype=args.comm_dtype, transport=args.transport)
        else:
            self.red = None
            if world > 1:
                self.fwd = torch.nn.parallel.DistributedDataParallel(net, device_ids=[dev.index], find_unused_parameters=True)
        self.params = [p for n, p in net.named_parameters() if not n.startswith("head_dist.")]
        if args.optimizer == "adamw":
            self.opt = torch.optim.AdamW(self.params, lr=lr, weight_decay=1e-2)
        else:
            self.opt = torch.optim.SGD(self.params, lr=lr)

    @property
    def flat_p(self):
        return torch.cat([p.detach().reshape(-1) for p in self.params])

    def step(self, x, y):
        logits, _ = self.fwd(x)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, y)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach()

    def close(self):
        from passt_amd import ddp
        ddp.detach(self.net)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--comm-dtype", default="fp32")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--overlap-wgrad", action="store_true")
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--backend", default="gloo", help="gloo: every rank on cuda:0; nccl (= RCCL): rank r on cuda:r")
    ap.add_argument("--transport", default="torch", help="torch | rccl_abi (pa_comm_* entry points; needs --backend nccl devices)")
    ap.add_argument("--path", default="trainstep", choices=["trainstep", "attach", "torch_ddp"],
                    help="trainstep: passt_amd.train.TrainStep.  attach / torch_ddp: the AUTOGRAD path (net(x); loss.backward(); "
                         "torch.optim.SGD) with passt_amd.ddp.attach(net) resp. torch's DistributedDataParallel wrapper "
                         "(find_unused_parameters=True, as the reference needs for head_dist)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", rank if (args.backend == "nccl" and world > 1) else 0)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    case = dict(G.CASES["model_small_train"], B=8, seed=333)
    cfg = case["cfg"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = passt_amd.PaSST(u_patchout=cfg["u_patchout"], s_patchout_t=cfg["s_patchout_t"], s_patchout_f=cfg["s_patchout_f"],
                              img_size=cfg["img_size"], patch_size=cfg["patch"], stride=cfg["stride"],
                              num_classes=cfg["num_classes"], embed_dim=cfg["embed_dim"], depth=cfg["depth"],
                              num_heads=cfg["num_heads"], distilled=True)
    sd = detgen.passt_state_dict(cfg, case["seed"])
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    net = net.to(dev).train()
    net.precision = "fp32"
    net.overlap_wgrad = args.overlap_wgrad
    if rank > 0:                        # a replica that was initialised differently: TrainStep must overwrite it with rank 0's
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.25)
    x, y = G.model_inputs(case)         # the global batch of 8 clips; rank r takes rows [4r, 4r+4)
    xg, yg = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    if world > 1:
        per = x.shape[0] // world
        xg, yg = xg[rank * per:(rank + 1) * per].contiguous(), yg[rank * per:(rank + 1) * per].contiguous()
    lr = 1e-3 if args.optimizer == "adamw" else 0.05
    if args.path == "trainstep":
        ts = TrainStep(net, None, lr=lr, weight_decay=1e-2, use_mixup=False,
                       comm_dtype=args.comm_dtype, optimizer=args.optimizer, transport=args.transport)
    else:
        ts = _AutogradRunner(net, args, lr, world, dev)
    init = ts.flat_p.clone()            # after the constructor's broadcast: rank 0's weights on every rank
    losses = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(args.steps):
            torch.manual_seed(900 + step)   # the SAME patchout draws on every rank and in the single-process run
            np.random.seed(900 + step)
            losses.append(float(ts.step(xg, yg).item()))
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        # all ranks must hold the same parameters after the steps
        mine = ts.flat_p.clone()
        ref0 = ts.flat_p.clone()
        dist.broadcast(ref0, 0)
        same = bool(torch.equal(mine, ref0))
        flags = [None] * world
        dist.all_gather_object(flags, (same, losses))
    else:
        flags = [(True, losses)]
    red = getattr(ts, "reducer", None) or getattr(ts, "red", None)
    comm = red.comm_info() if (red is not None and world > 1) else None       # what the communicator itself reports (pa_comm_info)
    if rank == 0:
        torch.save({"params": ts.flat_p.cpu(), "init": init.cpu(), "flags": flags, "world": world, "comm_info": comm}, args.out)
    ts.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
