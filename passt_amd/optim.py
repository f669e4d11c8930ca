"""Optimizers for the drop-in (autograd) path: ``torch.optim`` classes whose step is the library's fused kernel.

The reference builds ``torch.optim.AdamW(self.parameters(), lr=lr, weight_decay=weight_decay)`` (ex_audioset.py:104-109,
293-302).  torch's default implementation on a HIP device is the multi-tensor ("foreach") one: eight passes over the 86 M
parameters and their moments (measured on MI355X: 1.7 ms per step for passt_s against 0.4 ms for one fused pass,
profiles/r04_autograd_vs_trainstep.md).  ``passt_amd.optim.AdamW`` is the one-word change::

    -        return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)
    +        return passt_amd.optim.AdamW(params, lr=lr, weight_decay=weight_decay)

Same arithmetic as torch's (decoupled weight decay, bias corrections from a PER-PARAMETER step count, eps outside the square
root), same constructor
arguments, ``param_groups`` (LR schedulers keep working: ``group["lr"]`` is read every step), ``state_dict`` layout
({"step", "exp_avg", "exp_avg_sq"} per parameter).  Parameters and both moments of a group live in flat f32 buffers
(``p.data`` becomes a view, like TrainStep's); every step issues ONE ``pa_adamw`` launch per run of parameters whose
gradients are contiguous in memory in parameter order -- the autograd node of passt_amd.PaSST returns its gradients as views
of one flat buffer, so a whole PaSST is one launch -- and one launch per parameter otherwise.  Parameters without a
gradient (``head_dist.*``) are skipped, as torch does.

Round 6: when a group holds ALL the parameters of a live ``passt_amd.PaSST`` (the reference's ``self.parameters()`` does), the
optimizer BINDS the model (``PaSST.bind_flat_grads``): it owns one flat gradient buffer next to the flat parameters and moments,
the model's backward writes into it in place and ``p.grad`` are standing views of it.  The step is then ONE ``pa_adamw_stage``
launch (update + the bf16 weight copies of the next forward) and O(1) Python: no 159 AccumulateGrad nodes, no per-parameter
``state["step"]`` / version bumps (the parameters of a bound model share one step tensor), ``zero_grad()`` marks the buffer fresh
instead of dropping 159 ``.grad`` tensors.  ``PASST_AMD_NO_FLAT_GRADS=1`` keeps the unbound behaviour (A/B).
"""
import os

import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False,
                 foreach=None, capturable=False, differentiable=False, fused=None):
        if amsgrad or maximize or capturable or differentiable:
            raise NotImplementedError("passt_amd.optim.AdamW: amsgrad / maximize / capturable / differentiable are not supported "
                                      "(the reference uses none of them, ex_audioset.py:108)")
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = {}            # group index -> dict(ids, flat_p, m, v, offs)
        self._bound = {}           # group index -> dict(net (weakref), i0, i1, s, e, fl (the model's binding), step)

    # ---- flat storage -----------------------------------------------------------------------------------------------
    def _ensure_flat(self, gi, group):
        fl = self._flat.get(gi)
        b = self._bound.get(gi)
        if fl is not None and b is not None and fl.get("n_group") == len(group["params"]):
            # a bound model: its forward validates the parameter tree, its binding object is dropped by anything that moves the
            # storages (PaSST._apply) and a loaded optimizer state replaces the shared step tensor -- no per-parameter checks here
            net = b["net"]()
            if (net is not None and net._flat is b["fl"] and b["step"] is self.state[fl["ps"][b["i0"]]].get("step")
                    and all(p.requires_grad for p in fl["ps"])):
                return fl, fl["ps"]
        ps = [p for p in group["params"] if p.requires_grad]
        ok = fl is not None and fl["ids"] == [id(p) for p in ps]
        if ok:
            base = fl["flat_p"].data_ptr()
            ok = all(p.data_ptr() == base + 4 * off and p.is_contiguous() for p, off in zip(ps, fl["offs"]))
            # a loaded state_dict replaces the moment views by the loaded tensors: fold them back in
            if ok:
                for p, off in zip(ps, fl["offs"]):
                    st = self.state.get(p)
                    if st and st["exp_avg"].data_ptr() != fl["m"].data_ptr() + 4 * off:
                        ok = False
                        break
        if ok:
            if gi not in self._bound and fl.get("bind_retries", 0) > 0:     # e.g. only part of the gradients existed at the first step
                fl["bind_retries"] -= 1
                self._bind_model(gi, fl, ps)
            return fl, ps
        for p in ps:
            if p.dtype != torch.float32 or p.is_sparse:
                raise NotImplementedError("passt_amd.optim.AdamW updates dense float32 parameters")
        dev = ps[0].device
        total = sum(p.numel() for p in ps)
        flat_p = torch.empty(total, device=dev, dtype=torch.float32)
        m = torch.zeros(total, device=dev, dtype=torch.float32)
        v = torch.zeros(total, device=dev, dtype=torch.float32)
        offs, off, steps = [], 0, []
        for p in ps:
            n = p.numel()
            flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat_p[off:off + n].view(p.shape)
            st = self.state.get(p)
            if st:                 # moments that already exist (re-layout, load_state_dict) move into the flat buffers
                m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.append(int(st["step"]) if st else 0)     # PER PARAMETER, as torch counts them (state[p]["step"])
            offs.append(off)
            off += n
        fl = self._flat[gi] = dict(ids=[id(p) for p in ps], flat_p=flat_p, m=m, v=v, offs=offs, steps=steps, ps=ps, n_group=len(group["params"]), bind_retries=3)
        for p, o, t in zip(ps, offs, steps):
            if self.state.get(p):                       # existing per-parameter state now views the flat moments
                self._bind_state(fl, p, o, t)
        self._bind_model(gi, fl, ps)
        return fl, ps

    def _bind_model(self, gi, fl, ps):
        """If the group contains every gradient-carrying parameter of a live passt_amd.PaSST as one consecutive run (the
        reference's ``self.parameters()``: the whole model, head_dist.* behind it), give that model one flat gradient buffer."""
        import weakref

        from . import passt
        old = self._bound.pop(gi, None)
        if old is not None and old["net"]() is not None and old["net"]()._flat is old["fl"]:
            old["net"]().unbind_flat_grads(keep_grads=True)     # this step still consumes them, per parameter
        if os.environ.get("PASST_AMD_NO_FLAT_GRADS") == "1":
            return
        index = {id(p): i for i, p in enumerate(ps)}
        for net in list(passt._LIVE):
            if net._flat is not None:
                continue                                 # bound by another optimizer
            named = net._graph_params()[0]
            i0 = index.get(id(named[0][1])) if named else None
            if i0 is None or i0 + len(named) > len(ps) or any(ps[i0 + k] is not p for k, (_, p) in enumerate(named)):
                continue
            steps = fl["steps"][i0:i0 + len(named)]
            if len(set(steps)) != 1:
                continue                                 # per-parameter step counts differ (partially trained state): stay per parameter
            s, e = fl["offs"][i0], fl["offs"][i0 + len(named) - 1] + named[-1][1].numel()
            flat_g = torch.zeros(e - s, device=fl["flat_p"].device, dtype=torch.float32)
            b = net.bind_flat_grads(flat_g)
            if b is None:
                continue
            # one step counter for the run: every parameter's state["step"] is the SAME tensor (state_dict() still lists it per
            # parameter; a loaded state_dict brings separate tensors back and the next step re-shares them)
            step_t = torch.tensor(float(steps[0]))
            for k, (_, p) in enumerate(named):
                st = self._bind_state(fl, p, fl["offs"][i0 + k], steps[0])
                st["step"] = step_t
            self._bound[gi] = dict(net=weakref.ref(net), i0=i0, i1=i0 + len(named), s=s, e=e, fl=b, flat_g=flat_g, step=step_t)
            return

    def zero_grad(self, set_to_none=True):
        """Gradients of a bound model are overwritten by its next backward (the buffer is marked fresh; ``p.grad`` stay views of
        it); everything else as torch.optim.Optimizer.zero_grad."""
        if not self._bound:
            return super().zero_grad(set_to_none)
        for gi, group in enumerate(self.param_groups):
            b = self._bound.get(gi)
            ps = group["params"]
            if b is None or b["net"]() is None or b["net"]()._flat is not b["fl"]:
                rest = ps
            else:
                b["fl"]["fresh"] = True
                fl = self._flat[gi]
                ids = set(fl["ids"][b["i0"]:b["i1"]])
                rest = [p for p in ps if id(p) not in ids] if len(ps) != b["i1"] - b["i0"] else []
            for p in rest:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_().zero_()

    def _bind_state(self, fl, p, off, step):
        """per-parameter state as torch lays it out ({"step", "exp_avg", "exp_avg_sq"}), the moments being views of the flat
        buffers; created on a parameter's first gradient, like torch.optim.AdamW does"""
        st = self.state[p]
        st["exp_avg"] = fl["m"][off:off + p.numel()].view(p.shape)
        st["exp_avg_sq"] = fl["v"][off:off + p.numel()].view(p.shape)
        if not torch.is_tensor(st.get("step")):
            st["step"] = torch.tensor(float(step))
        return st

    def _step_bound(self, b, fl, ps, lr, b1, b2, eps, wd):
        """The bound model's parameters: ONE pa_adamw_stage launch on (flat parameters, the model's flat gradient buffer, flat
        moments) that also rewrites the GEMM-ready weight copies of the dtype the model last ran in.  False: the binding is gone
        (model moved / re-built / its .grad replaced by the caller) -- the per-parameter path takes over and re-binds later."""
        net = b["net"]()
        i0, i1 = b["i0"], b["i1"]
        if net is None or net._flat is not b["fl"] or b["step"] is not self.state[ps[i0]].get("step"):
            self._bound = {k: v for k, v in self._bound.items() if v is not b}
            if net is not None and net._flat is b["fl"]:
                net.unbind_flat_grads(keep_grads=True)
            return False
        g0, g1 = ps[i0].grad, ps[i1 - 1].grad
        base = b["flat_g"].data_ptr()
        if g0 is None or g1 is None or g0.data_ptr() != base or g1.data_ptr() != base + 4 * (fl["offs"][i1 - 1] - b["s"]):
            return False                                 # the caller replaced .grad (set_to_none by foreign code, clipping copies): this step per parameter
        if b["fl"]["fresh"]:
            return True                                  # no backward since zero_grad(): nothing to apply (torch skips parameters without grad)
        from ._lib import PA_BF16
        t = fl["steps"][i0] + 1
        s, e = b["s"], b["e"]
        st = net._staged
        dt = net._last_dt if net._last_dt is not None else PA_BF16
        tab, n, items, keys = st.adamw_table(("optim", id(self), s, e), [(ps[k], fl["offs"][k] - s) for k in range(i0, i1)], dt)
        ops.adamw_stage(fl["flat_p"][s:e], b["flat_g"], fl["m"][s:e], fl["v"][s:e], tab, n, items, dt, lr, b1, b2, eps, wd, t)
        b["step"].fill_(float(t))
        for k in range(i0, i1):
            fl["steps"][k] = t
        net.mark_params_updated()
        st.mark_fresh(keys)
        return True

    def _launch(self, p, g, m, v, lr, b1, b2, eps, wd, step):
        ops.adamw(p, g, m, v, lr, b1, b2, eps, wd, step)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if not any(p.requires_grad for p in group["params"]):
                continue
            fl, ps = self._ensure_flat(gi, group)
            (b1, b2), lr, eps, wd = group["betas"], float(group["lr"]), group["eps"], group["weight_decay"]
            offs, steps = fl["offs"], fl["steps"]
            # runs of consecutive parameters whose gradients are dense f32 and adjacent in memory: one launch each
            i, n_p = 0, len(ps)
            bound = self._bound.get(gi)
            if bound is not None and not self._step_bound(bound, fl, ps, lr, b1, b2, eps, wd):
                bound = None
            while i < n_p:
                if bound is not None and i == bound["i0"]:
                    i = bound["i1"]                      # the model's run went out as one launch above
                    continue
                g = ps[i].grad
                if g is None:
                    i += 1
                    continue
                if g.dtype != torch.float32 or g.is_sparse:
                    raise NotImplementedError("passt_amd.optim.AdamW needs dense float32 gradients")
                if not g.is_contiguous():
                    g = ps[i].grad = g.contiguous()
                start, gptr = offs[i], g.data_ptr()
                end = start + ps[i].numel()
                j = i + 1
                while j < n_p:
                    gj = ps[j].grad
                    # adjacent AND one allocation AND the same step count: the bias corrections are per parameter (a parameter
                    # whose first gradient arrives later -- unfrozen layer -- starts at step 1 like torch's, not at the group's)
                    if (gj is None or gj.dtype != torch.float32 or not gj.is_contiguous() or gj.data_ptr() != gptr + 4 * (end - start)
                            or gj.untyped_storage().data_ptr() != g.untyped_storage().data_ptr() or steps[j] != steps[i]):
                        break
                    end += ps[j].numel()
                    j += 1
                n = end - start
                t = steps[i] + 1
                # one flat view over the run's gradients (adjacent views of one allocation: as_strided from the first)
                gflat = g.as_strided((n,), (1,)) if j > i + 1 else g.reshape(-1)
                self._launch(fl["flat_p"][start:end], gflat, fl["m"][start:end], fl["v"][start:end], lr, b1, b2, eps, wd, t)
                for k in range(i, j):
                    st = self.state[ps[k]]
                    if "exp_avg" not in st:
                        st = self._bind_state(fl, ps[k], offs[k], t)
                    st["step"].fill_(float(t))
                    steps[k] = t
                    torch.autograd.graph.increment_version(ps[k])      # raw-pointer update: consumers key on _version
                i = j
        return loss
