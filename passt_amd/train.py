"""The training step of the reference's caller, on the kernel path.

Mirrors ``M.training_step`` (ex_audioset.py:155-198) + the optimizer step Lightning performs:
waveform -> AugmentMelSTFT (:158) -> spectrogram mixup (:171-177, helpers/mixup.py) -> PaSST (:179)
-> BCE-with-logits mean on the mixed targets (:181-186) -> backward -> gradient all-reduce (DDP) ->
AdamW (:104-109) or SGD (model_speed_test, :392).  No autograd graph is built: forward/backward are
explicit kernel sequences, parameters and gradients live in flat f32 buffers (one fused optimizer
launch, per-block all-reduce buckets).
"""
import os

import numpy as np
import torch

from . import ops
from .ddp import GradReducer
from .passt import _precision, passt_backward, passt_forward, patchout_draws


class TrainStep:
    def __init__(self, net, mel=None, lr=2e-5, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, optimizer="adamw",
                 mixup_alpha=0.3, use_mixup=True, process_group=None, loss="bce", comm_dtype="fp32", transport="torch", graph=False):
        """graph=True (single GPU): after ``graph_warmup`` eager steps the network's forward, the loss, the backward and the
        per-bucket optimizer launches (~290 of the step's ~300 launches) are captured once as a hipGraph (torch.cuda.CUDAGraph)
        and replayed; only the front end, mixup and the uploads of the step's host-drawn arrays stay eager.  Same kernels, same
        arguments, bit-identical parameters (tests/test_gpu_model.py::test_train_step_graph_equals_eager); what changes is the
        host: one replay instead of ~290 ctypes launches -- it matters where the step is short (ESC-50 at batch 12)."""
        self.net, self.mel = net, mel
        self.lr, self.wd, self.betas, self.eps, self.optimizer = lr, weight_decay, betas, eps, optimizer
        self.mixup_alpha, self.use_mixup = mixup_alpha, use_mixup
        assert loss in ("bce", "ce")      # bce: ex_audioset.py:181-186 ; ce: ex_esc50.py:159-165 (class-index targets)
        self.loss = loss
        dev = next(net.parameters()).device
        frozen = [n for n, p in net.named_parameters() if not p.requires_grad and not n.startswith("head_dist.")]
        if frozen:
            raise NotImplementedError("TrainStep updates every PaSST parameter through one flat buffer (the reference trains all of "
                                      f"them, ex_audioset.py:104-109); frozen parameters are not supported here: {frozen[:3]} ... "
                                      "use the autograd path (net(x); loss.backward()) with your own optimizer instead")
        names = net._grad_names
        self.named = [(n, p) for n, p in net.named_parameters() if n in names]
        total = sum(p.numel() for _, p in self.named)
        # flat parameter buffer: every nn.Parameter becomes a view (state_dict / load_state_dict keep working)
        self.flat_p = torch.empty(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grads, off = {}, 0
        for n, p in self.named:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            self.grads[n] = self.flat_g[off:off + k].view(p.shape)
            p.grad = self.grads[n]
            off += k
        self.m = torch.zeros_like(self.flat_p) if optimizer == "adamw" else None
        self.v = torch.zeros_like(self.flat_p) if optimizer == "adamw" else None
        self.reducer = GradReducer(self.flat_g, [(n, p.numel()) for n, p in self.named], len(net.blocks), process_group,
                                   comm_dtype=comm_dtype, transport=transport)
        # identical replicas: rank 0's parameters everywhere (what Lightning's DDP wrapper does at construction,
        # ex_audioset.py:488-489); a caller that seeded per rank or loaded different state must not train diverging copies
        self.reducer.broadcast_(self.flat_p)
        if self.reducer.world > 1 and os.environ.get("PASST_AMD_DDP_PERSISTENT") != "1":
            # the all-reduce kernels of the communication stream take CUs while the backward runs: GEMMs go out one work
            # item per workgroup (the hardware hands them to whatever CUs are free: -1.6 % alone on one GPU) instead of as 256
            # resident workgroups, which would wait for the occupied CUs and double the launch (PA_GEMM_NO_PERSIST).
            # Per model (net._gemm_flags rides on every pa_gemm_nt call of this net's forward / backward); close() undoes it.
            object.__setattr__(net, "_gemm_flags", getattr(net, "_gemm_flags", 0) | ops._lib.GEMM_NO_PERSIST)
            self._set_no_persist = True
        self.graph = bool(graph) and self.reducer.world == 1
        if self.graph and optimizer != "adamw":
            raise NotImplementedError("TrainStep(graph=True) captures the per-bucket AdamW launches (pa_adamw_dev reads its scalars "
                                      "from device memory); optimizer='sgd' has no such form: use graph=False")
        self.graph_warmup, self._g = 3, None
        self.t = self._t_now = 0
        self.block_optimizer = os.environ.get("PASST_AMD_BLOCK_OPT", "1") != "0"      # one GPU: per-bucket updates from the backward
        self.base_lr = lr
        # AdamW and the GEMM-ready weight copies in one launch per bucket (pa_adamw_stage, round 6): the copies the next forward
        # needs are written from the registers that hold the updated parameters; PASST_AMD_NO_FUSED_STAGE=1: A/B, the optimizer
        # leaves them stale and the next forward's first GEMM refreshes all of them with one pa_stage_weights launch
        self.fused_stage = optimizer == "adamw" and os.environ.get("PASST_AMD_NO_FUSED_STAGE") != "1"
        self._offsets = []
        off = 0
        for _, p in self.named:
            self._offsets.append(off)
            off += p.numel()
        self._fresh = []
        net.mark_params_updated()

    def _optimizer(self, s, e):
        if self._g is not None and self._g.get("capturing"):
            # inside the capture: the step's scalars come from device memory (launch arguments must not change between replays)
            if self.optimizer != "adamw":
                raise NotImplementedError("graph mode: AdamW only")
            return ops.adamw_dev(self.flat_p[s:e], self.flat_g[s:e], self.m[s:e], self.v[s:e], self._g["hyper"])
        if self.optimizer == "adamw" and self.fused_stage:
            # no kernel enqueued after this point reads this bucket's copies before the next forward: the bucket's own input-
            # gradient GEMMs (the readers of W^T) are already on the stream, earlier blocks read their own weights
            st, dt = self.net._staged, _precision(self.net)
            lo, hi = self._span_params(s, e)
            tab, n, items, keys = st.adamw_table((s, e), [(self.named[i][1], self._offsets[i] - s) for i in range(lo, hi)], dt)
            ops.adamw_stage(self.flat_p[s:e], self.flat_g[s:e], self.m[s:e], self.v[s:e], tab, n, items, dt, self.lr, self.betas[0],
                            self.betas[1], self.eps, self.wd, self._t_now)
            self._fresh += keys
        elif self.optimizer == "adamw":
            ops.adamw(self.flat_p[s:e], self.flat_g[s:e], self.m[s:e], self.v[s:e], self.lr, self.betas[0], self.betas[1],
                      self.eps, self.wd, self._t_now)
        else:
            ops.sgd(self.flat_p[s:e], self.flat_g[s:e], self.lr)

    def _span_params(self, s, e):
        """indices [lo, hi) into self.named of the parameters that make up flat span [s, e) (spans are unions of whole parameters)"""
        import bisect
        lo, hi = bisect.bisect_left(self._offsets, s), bisect.bisect_left(self._offsets, e)
        assert self._offsets[lo] == s and (hi == len(self._offsets) or self._offsets[hi] == e)
        return lo, hi

    def _params_updated(self):
        """End of a step: every parameter changed (epoch bump: all staged copies stale) -- except that the copies the fused
        optimizer launches of this step rewrote are current."""
        self.net.mark_params_updated()
        if self._fresh:
            self.net._staged.mark_fresh(self._fresh)
            self._fresh = []

    def _block_done(self, i):
        """Called by the backward on its finishing stream once bucket i's gradients are complete (head, blocks depth-1 .. 0,
        patch embedding).  One GPU: update that bucket's parameters right away -- no later kernel of this backward reads
        them (input gradients use the bf16 copies staged before the step), so the bandwidth-bound optimizer pass runs next
        to the remaining blocks' GEMMs instead of after them.  Several GPUs: start the bucket's all-reduce."""
        if self.reducer.world > 1:
            return self.reducer.on_block_done(i)
        if self.block_optimizer:
            s, e = self.reducer.spans[i]
            if e > s:
                self._optimizer(s, e)

    def close(self):
        """Release the reducer's communicator (C-ABI RCCL transport); torch.distributed groups belong to the caller."""
        self.reducer.close()
        if getattr(self, "_set_no_persist", False):
            object.__setattr__(self.net, "_gemm_flags", getattr(self.net, "_gemm_flags", 0) & ~ops._lib.GEMM_NO_PERSIST)
            self._set_no_persist = False

    def set_lr_factor(self, factor):
        """Per-epoch LR schedule (ex_audioset.py:86-101): lr = base_lr * factor, e.g. from
        schedule.exp_warmup_linear_down(5, 50, 50, 0.01)(epoch)."""
        self.lr = self.base_lr * float(factor)

    # ---- captured-graph mode ------------------------------------------------------------------------------------------
    def _graph_fill(self, g, d, x, y, y2, lam_d):
        g["x"].copy_(x)
        g["y"].copy_(y)
        if y2 is not None:
            g["y2"].copy_(y2)
        if lam_d is not None:
            g["lam"].copy_(lam_d)
        g["pf"].copy_(ops.upload_small(d["pf"], x.device))
        g["pt"].copy_(ops.upload_small(d["pt"], x.device))
        g["pt_pos"].copy_(ops.upload_small(d["pt"] + np.int32(d["toff"]), x.device))
        # (through the pinned ring like the index arrays: a fixed host buffer could be rewritten for step k + 1 before the
        # asynchronous copy of step k has read it)
        ops.adamw_hyper(self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self._t_now, g["hyper_host"])
        g["hyper"].copy_(ops.upload_small(g["hyper_host"], x.device))

    def _graph_body(self, g):
        net = self.net
        logits, feat, ctx = passt_forward(net, g["x"], save=True, draws=g)
        if self.loss == "bce":
            loss, dlogits = ops.bce_fwd_bwd(logits, g["y"], grad_scale=1.0)
        elif g["y2"] is not None:
            loss, dlogits = ops.ce_mixup_fwd_bwd(logits, g["y"], g["y2"], g["lam"], grad_scale=1.0)
        else:
            loss, dlogits = ops.ce_mixup_fwd_bwd(logits, g["y"], grad_scale=1.0)
        passt_backward(net, ctx, dlogits, None, self.grads, on_block_done=self._block_done)
        if not self.block_optimizer:
            self._optimizer(0, self.flat_p.numel())
        return loss

    def step(self, wave_or_spec, target):
        """wave (B,1,L) / (B,L) when a mel module was given, else a spectrogram (B,1,F,T).
        Returns the loss as a 1-element device tensor (no host sync)."""
        net = self.net
        x = wave_or_spec
        if self.mel is not None:
            if x.dim() == 3:
                x = x.reshape(-1, x.shape[2])                                   # mel_forward, :142-145
            x = self.mel(x.contiguous().float()).unsqueeze(1)
        else:
            x = x.contiguous().float()
        # the kernels reinterpret raw memory: hand them dense f32 targets (BCE) / integer class ids (CE)
        y = target.contiguous() if self.loss == "ce" else target.to(torch.float32).contiguous()
        if self.use_mixup:
            B = x.shape[0]
            perm = torch.randperm(B)                                            # helpers/mixup.py:6
            lam = np.random.beta(self.mixup_alpha, self.mixup_alpha, B).astype(np.float32)
            lam = np.maximum(lam, 1.0 - lam)
            perm_d = ops.upload_small(perm.to(torch.int32), x.device)           # page-locked staging: asynchronous H2D
            lam_d = ops.upload_small(lam, x.device)
            x = ops.mixup(x, perm_d, lam_d)
            if self.loss == "bce":
                y = ops.mixup(y, perm_d, lam_d)
        if self.graph and self.t >= self.graph_warmup and self._graph_fits(x):
            return self._graph_step(x, y, perm_d if self.use_mixup else None, lam_d if self.use_mixup else None)
        logits, feat, ctx = passt_forward(net, x, save=True)
        gs = 1.0 / self.reducer.world
        if self.loss == "bce":
            loss, dlogits = ops.bce_fwd_bwd(logits, y, grad_scale=gs)
        else:
            y32 = y.to(torch.int32)
            if self.use_mixup:
                loss, dlogits = ops.ce_mixup_fwd_bwd(logits, y32, y32[perm_d.long()].contiguous(), lam_d, grad_scale=gs)
            else:
                loss, dlogits = ops.ce_mixup_fwd_bwd(logits, y32, grad_scale=gs)
        # The step count advances only when the whole step was enqueued.  With per-bucket updates (the default) a step that
        # raises in the middle of the backward (out of memory, a PA_E* code) has ALREADY updated the buckets that completed
        # before it (head, later blocks) with step number t + 1 -- parameters, m and v of those buckets -- and not the rest:
        # a failed step is not atomic; recover from a checkpoint (or run with PASST_AMD_BLOCK_OPT=0, where nothing is
        # touched before the backward has finished).
        self._t_now = self.t + 1
        passt_backward(net, ctx, dlogits, None, self.grads, on_block_done=self._block_done)
        if self.reducer.world == 1 and not self.block_optimizer:
            self._optimizer(0, self.flat_p.numel())
        if self.reducer.world > 1:
            # one optimizer launch per all-reduce bucket, in the order the buckets were launched: the update of bucket k runs
            # while buckets k+1.. are still on the wire, so the LAST bucket (block 0 + patch embedding, released only when
            # the backward ends) is covered by ~0.4 ms of optimizer work instead of being exposed in front of one big launch
            for s_, e_ in self.reducer.drain():
                self._optimizer(s_, e_)
        self.t = self._t_now
        self._params_updated()
        return loss

    def _graph_fits(self, x):
        """The captured graph replays ONE batch shape.  The reference's loaders never set drop_last, so the last batch of an epoch
        is smaller (and variable-length clips change the frame count): such a step runs the eager kernel sequence instead --
        same kernels, same results, decided here before any of the step's random draws is consumed (the number of kept patches
        follows from the shape alone).  The graph stays valid for the next full batch."""
        g = self._g
        return g is None or "graph" not in g or tuple(x.shape) == tuple(g["x"].shape)

    def _graph_step(self, x, y, perm_d, lam_d):
        """x: mixed spectrogram; y: targets (BCE: already mixed f32; CE: class ids).  Host draws in the eager order, fixed-address
        buffers refilled, ONE graph replay."""
        net = self.net
        y2 = None
        if self.loss == "ce":
            y = y.to(torch.int32)
            if perm_d is not None:
                y2 = y[perm_d.long()].contiguous()
            else:
                lam_d = None
        else:
            lam_d = None
        self._t_now = self.t + 1
        d = patchout_draws(net, x.shape)                     # the Patchout draws of this step, where passt_forward would draw them
        g = self._g
        if g is None or "graph" not in g:
            if g is None:
                g = dict(capturing=False, x=torch.empty_like(x), Np=d["Np"],
                         pf=torch.empty(d["Np"], device=x.device, dtype=torch.int32), pt=torch.empty(d["Np"], device=x.device, dtype=torch.int32),
                         pt_pos=torch.empty(d["Np"], device=x.device, dtype=torch.int32), y=torch.empty_like(y),
                         y2=None if y2 is None else torch.empty_like(y2), lam=None if lam_d is None else torch.empty_like(lam_d),
                         hyper=torch.empty(7, device=x.device, dtype=torch.float32), hyper_host=torch.empty(7, dtype=torch.float32))
                self._g = g
            self._graph_fill(g, d, x, y, y2, lam_d)
            net.mark_params_updated()                        # the staged weight copies are stale: their refresh launch is captured too
            net._staged.stage_table(_precision(net))         # (its descriptor table is uploaded here: no copies inside a capture)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            g["capturing"] = True
            try:
                with torch.cuda.graph(graph):
                    g["loss"] = self._graph_body(g)
            finally:
                g["capturing"] = False
            g["graph"] = graph
        else:
            # _graph_fits() routed every other shape to the eager path before the draws; the kept-patch count follows from the shape
            assert d["Np"] == g["Np"] and x.shape == g["x"].shape
            self._graph_fill(g, d, x, y, y2, lam_d)
        g["graph"].replay()
        self.t = self._t_now
        net.mark_params_updated()
        # a fresh tensor per step, like the eager path: the next replay overwrites the captured one, and callers collect losses
        # lazily (append now, .item() later)
        return g["loss"].clone()
