"""Host logic of passt_amd.optim.AdamW without a GPU: flat parameter / moment storage, runs of adjacent gradients -> one
launch each, parameters without a gradient skipped, LR schedulers, state_dict round trip.  The kernel call is replaced by
its arithmetic in torch (csrc/optim.hip adamw_one, restated); the HIP kernel itself is checked on the GPU
(tests/test_gpu_model.py::test_optim_adamw_matches_torch)."""
import copy
import math

import torch

from passt_amd.optim import AdamW


class _RefLaunch(AdamW):
    launches = 0

    def _launch(self, p, g, m, v, lr, b1, b2, eps, wd, step):
        type(self).launches += 1
        bc1, bc2s = 1.0 - b1 ** step, math.sqrt(1.0 - b2 ** step)
        p.mul_(1.0 - lr * wd)
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        p.sub_((lr / bc1) * (m / (v.sqrt() / bc2s + eps)))


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1, 1, 8), (8, 4), (8,), (12, 8), (12,), (5, 8), (5,), (3, 3)]      # the last one never gets a gradient
    return [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]


def _set_grads(ps, step, flat_layout):
    g = torch.Generator().manual_seed(100 + step)
    live = ps[:-1]
    if flat_layout:                                     # what _PasstFunction.backward returns: views of one buffer
        flat = torch.randn(sum(p.numel() for p in live), generator=g)
        off = 0
        for p in live:
            p.grad = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
    else:
        for p in live:
            p.grad = torch.randn(p.shape, generator=g)


def _run(opt_cls, flat_layout, steps=4, sched=True, reload_at=None):
    ps = _params(3)
    opt = opt_cls(ps, lr=1e-2, weight_decay=0.05)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 1.0 / (1 + e)) if sched else None
    for s in range(steps):
        if reload_at == s:                              # checkpoint round trip in the middle of training
            sd = copy.deepcopy(opt.state_dict())
            opt = opt_cls(ps, lr=1e-2, weight_decay=0.05)
            opt.load_state_dict(sd)
            if sched:
                sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 1.0 / (1 + e), last_epoch=s - 1)
        _set_grads(ps, s, flat_layout)
        opt.step()
        if sch:
            sch.step()
        opt.zero_grad()
    return ps, opt


def test_matches_torch_adamw_flat_and_scattered_gradients():
    ref, _ = _run(torch.optim.AdamW, False)
    for flat_layout, want_launches in ((True, 4 * 1), (False, 4 * 7)):
        _RefLaunch.launches = 0
        got, opt = _run(_RefLaunch, flat_layout)
        # flat gradient layout: ONE launch per step for the seven live parameters; scattered: one each
        assert _RefLaunch.launches == want_launches if flat_layout else _RefLaunch.launches >= 4
        for a, b in zip(got, ref if not flat_layout else _run(torch.optim.AdamW, True)[0]):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)
        assert got[-1].grad is None and not opt.state.get(got[-1])     # never had a gradient: no state, as in torch
        # parameters are views of one flat buffer, in order
        fl = opt._flat[0]
        assert all(p.data_ptr() == fl["flat_p"].data_ptr() + 4 * o for p, o in zip([p for p in got if p.requires_grad], fl["offs"]))


def test_versions_bump_and_state_dict_round_trip():
    ps, opt = _run(_RefLaunch, True, steps=1, sched=False)
    v0 = [p._version for p in ps]
    _set_grads(ps, 7, True)
    opt.step()
    assert all(p._version > a for p, a in zip(ps[:-1], v0[:-1])) and ps[-1]._version == v0[-1]
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 2.0
    ref, _ = _run(torch.optim.AdamW, True, steps=5)
    got, _ = _run(_RefLaunch, True, steps=5, reload_at=3)
    for a, b in zip(got, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)


def test_step_count_is_per_parameter_like_torch():
    """A parameter whose first gradient arrives later (unfrozen layer) gets the bias correction of ITS first step, not of the
    group's step count: torch.optim.AdamW keeps state[p]['step'] per parameter.  Also across a state_dict round
    trip with heterogeneous steps."""
    def run(opt_cls, reload_at=None):
        ps = _params(5)
        opt = opt_cls(ps, lr=1e-2, weight_decay=0.05)
        for s in range(8):
            if reload_at == s:
                sd = copy.deepcopy(opt.state_dict())
                opt = opt_cls(ps, lr=1e-2, weight_decay=0.05)
                opt.load_state_dict(sd)
            _set_grads(ps, s, True)
            if s < 5:                                   # parameters 2 and 5 are "frozen" for the first five steps
                ps[2].grad = None
                ps[5].grad = None
            opt.step()
            opt.zero_grad()
        return ps, opt
    ref, ropt = run(torch.optim.AdamW)
    for reload_at in (None, 6):
        got, opt = run(_RefLaunch, reload_at)
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), i
        for p, q in zip(got[:-1], ref[:-1]):
            assert float(opt.state[p]["step"]) == float(ropt.state[q]["step"])
        assert float(opt.state[got[2]]["step"]) == 3.0 and float(opt.state[got[0]]["step"]) == 8.0
