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
"""
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

    # ---- flat storage -----------------------------------------------------------------------------------------------
    def _ensure_flat(self, gi, group):
        ps = [p for p in group["params"] if p.requires_grad]
        fl = self._flat.get(gi)
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
        fl = self._flat[gi] = dict(ids=[id(p) for p in ps], flat_p=flat_p, m=m, v=v, offs=offs, steps=steps)
        for p, o, t in zip(ps, offs, steps):
            if self.state.get(p):                       # existing per-parameter state now views the flat moments
                self._bind_state(fl, p, o, t)
        return fl, ps

    def _bind_state(self, fl, p, off, step):
        """per-parameter state as torch lays it out ({"step", "exp_avg", "exp_avg_sq"}), the moments being views of the flat
        buffers; created on a parameter's first gradient, like torch.optim.AdamW does"""
        st = self.state[p]
        st["exp_avg"] = fl["m"][off:off + p.numel()].view(p.shape)
        st["exp_avg_sq"] = fl["v"][off:off + p.numel()].view(p.shape)
        if not torch.is_tensor(st.get("step")):
            st["step"] = torch.tensor(float(step))
        return st

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
            while i < n_p:
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
