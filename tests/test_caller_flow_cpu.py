"""The reference's own caller flow around the two modules, on CPU (behaviours ex_audioset.py relies on).

* ``passt_amd.mixup.my_mixup`` == the live ``helpers/mixup.py`` function, bit for bit, on the same RNG state;
* ``count_non_zero_params(net)`` (ex_audioset.py:121), imported live from the reference, walks a passt_amd.PaSST;
* ``torch.compile(net)`` (ex_audioset.py:135, model_speed_test :391): the compiled module runs the forward as ONE opaque
  eager call -- no graph is captured, nothing is recompiled on later calls (the GPU half of this is
  tests/test_gpu_model.py::test_model_speed_test_flow).
"""
import os
import warnings

import numpy as np
import pytest
import torch

import passt_amd
from oracle import ref_import
from passt_amd.mixup import my_mixup

needs_ref = pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")


@needs_ref
@pytest.mark.parametrize("size,alpha", [(64, 0.3), (12, 0.3), (1, 0.3), (7, 2.0)])
def test_my_mixup_is_the_reference_function(size, alpha):
    ref = ref_import.import_reference_file("helpers/mixup.py").my_mixup
    for seed in (0, 1, 1234):
        torch.manual_seed(seed)
        np.random.seed(seed)
        r_idx, r_lam = ref(size, alpha)
        after_ref = (torch.randint(1 << 30, (1,)).item(), np.random.randint(1 << 30))
        torch.manual_seed(seed)
        np.random.seed(seed)
        idx, lam = my_mixup(size, alpha, device="cpu")
        after = (torch.randint(1 << 30, (1,)).item(), np.random.randint(1 << 30))
        assert idx.dtype == r_idx.dtype and lam.dtype == r_lam.dtype and lam.shape == r_lam.shape
        assert torch.equal(idx, r_idx) and torch.equal(lam, r_lam)
        assert after == after_ref                      # both generators were advanced exactly as the reference advances them
        assert float(lam.min()) >= 0.5                 # max(lam, 1 - lam)


def test_my_mixup_default_device_without_a_gpu_is_the_cpu():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible: the default is that device (GPU suite)")
    idx, lam = my_mixup(8, 0.3)
    assert idx.device.type == "cpu" and lam.device.type == "cpu" and sorted(idx.tolist()) == list(range(8))


def _passt_s():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return passt_amd.get_model(arch="passt_s_swa_p16_128_ap476", pretrained=False, s_patchout_t=40, s_patchout_f=4)


@needs_ref
def test_count_non_zero_params_walks_the_module():
    count = ref_import.import_reference_file("helpers/models_size.py").count_non_zero_params
    net = _passt_s()
    desc, total, nonzero = count(net)
    assert total == sum(p.numel() for p in net.parameters()) == 86_153_758          # passt_s / 527 classes
    assert 0 < nonzero <= total and "type Linear, weight" in desc and "type PaSST, cls_token" in desc


def test_eager_import_does_not_load_dynamo():
    """The eager path must not import torch._dynamo (its ~900 modules make every full cyclic-GC pass of the process far more
    expensive: measured +6 ms per training step at ESC-50's batch 12 with bench.py's event bookkeeping): the compiler-opaque
    wrapper of PaSST.forward / AugmentMelSTFT.forward (_lib.compile_opaque) is in place from class definition on and needs
    nothing of it; and no process-wide nn.Module registration hook is installed by the import."""
    import subprocess
    import sys
    code = ("import sys, warnings; warnings.simplefilter('ignore'); import torch; "
            "from torch.nn.modules import module as M; "
            "h0 = (len(M._global_parameter_registration_hooks), len(M._global_module_registration_hooks)); "
            "import passt_amd; "
            "net = passt_amd.PaSST(img_size=(128, 250), stride=10, num_classes=7, embed_dim=128, depth=1, num_heads=2, distilled=True); "
            "mel = passt_amd.AugmentMelSTFT(); "
            "assert 'torch._dynamo' not in sys.modules; "
            "assert h0 == (len(M._global_parameter_registration_hooks), len(M._global_module_registration_hooks)); "
            "assert passt_amd.PaSST.forward._torchdynamo_disable and type(mel).forward._torchdynamo_disable; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]


def _small_net():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return passt_amd.PaSST(img_size=(128, 250), stride=10, num_classes=37, embed_dim=128, depth=2, num_heads=2,
                               distilled=True, s_patchout_t=6, s_patchout_f=3).train()


def test_torch_compile_leaves_the_forward_opaque():
    from torch._dynamo.utils import counters
    net = _small_net()
    compiled = torch.compile(net)
    assert list(compiled.state_dict()) == ["_orig_mod." + k for k in net.state_dict()]
    assert [id(p) for p in compiled.parameters()] == [id(p) for p in net.parameters()]     # SGD(net.parameters()) after compile
    counters.clear()
    x = torch.ones(2, 1, 128, 250)
    for _ in range(3):
        # the eager forward is reached through the compiled wrapper (and refuses a CPU tensor: no CPU path)
        with pytest.raises(passt_amd._lib.PasstAmdError, match="HIP device only"):
            compiled(x)
    assert counters["stats"].get("unique_graphs", 0) == 0            # nothing captured
    assert counters["frames"].get("total", 0) <= 1                   # the wrapper frame, once: no recompilation per call


@pytest.mark.parametrize("flow", ["function", "bound_forward", "module_compile", "mel_function"])
def test_every_compile_flow_meets_the_opaque_forward(flow, monkeypatch):
    """torch.compile(train_step_fn) and torch.compile(net.forward) register nothing on a module, so a wrapper
    installed lazily from a registration hook never saw them and dynamo walked into the ctypes launches.  The wrapper is now
    part of the class: whichever way the compile is set up, the tracer meets a function marked compiler-disabled, breaks the
    graph around it and the forward body runs eagerly (observed through a counting stand-in for the kernel sequence; the
    tensor math around the call IS captured)."""
    import torch._dynamo
    from torch._dynamo.utils import counters
    net = _small_net()
    ran = []

    def fake_forward(model, x, save, draws=None):            # what the eager body calls; must never be traced
        assert not torch.compiler.is_compiling()
        ran.append(tuple(x.shape))
        return torch.zeros(x.shape[0], 37), torch.zeros(x.shape[0], 128), None

    monkeypatch.setattr(passt_amd.passt, "passt_forward", fake_forward)
    monkeypatch.setattr(passt_amd.ops, "mel_frontend", lambda x, *a: (ran.append(tuple(x.shape)), x.new_zeros(x.shape[0], 128, 4))[1])
    torch._dynamo.reset()
    counters.clear()
    x = torch.ones(2, 1, 128, 250)
    with torch.no_grad():
        if flow == "function":
            fn = torch.compile(lambda t: net(t + 1.0)[0] * 3.0, backend="eager")
        elif flow == "bound_forward":
            fn = torch.compile(net.forward, backend="eager")
        elif flow == "module_compile":
            net.compile(backend="eager")
            fn = net
        else:
            mel = passt_amd.AugmentMelSTFT().eval()
            monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))      # reach the launch site on this CPU box
            fn, x = torch.compile(lambda w: mel(w * 0.5) + 1.0, backend="eager"), torch.ones(2, 3200)
        for _ in range(3):
            out = fn(x)
    assert len(ran) == 3 and all(r == tuple(x.shape) for r in ran)
    assert (out[0] if isinstance(out, tuple) else out).shape[0] == 2
    if flow in ("function", "mel_function"):
        assert counters["stats"].get("unique_graphs", 0) == 2         # the math before and after the opaque call, nothing of it
        assert any("disable" in k for k in counters["graph_break"])
    else:
        assert counters["stats"].get("unique_graphs", 0) == 0
    assert counters["frames"].get("total", 0) <= 2                    # no recompilation per call


def test_parameter_list_cache_follows_surgery_anywhere_in_the_tree():
    """the autograd node's cached (name, parameter) list must not survive net.head[1] = nn.Linear(...) or the
    replacement of a block's sub-module (the kernels read the live modules)."""
    import copy
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = passt_amd.PaSST(img_size=(128, 250), stride=10, num_classes=37, embed_dim=128, depth=2, num_heads=2, distilled=True)
    a = net._graph_params()
    assert net._graph_params()[0] is a[0]                            # cached while nothing is registered anywhere
    net.head[1] = torch.nn.Linear(128, 50)
    b = net._graph_params()
    assert b[0] is not a[0] and dict(b[0])["head.1.weight"] is net.head[1].weight and b[1] == a[1] + 13 * 129
    net.blocks[0].mlp.fc1 = torch.nn.Linear(128, 512)
    assert dict(net._graph_params()[0])["blocks.0.mlp.fc1.weight"] is net.blocks[0].mlp.fc1.weight
    twin = copy.deepcopy(net)
    assert [n for n, _ in twin._graph_params()[0]] == [n for n, _ in net._graph_params()[0]]
    assert all(p is q for (_, p), q in zip(twin._graph_params()[0], [p for n, p in twin.named_parameters() if not n.startswith("head_dist.")]))
