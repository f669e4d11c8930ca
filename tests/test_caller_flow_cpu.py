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
    wrapper of PaSST.forward is installed only when a compile is being set up."""
    import subprocess
    import sys
    code = ("import sys, warnings; warnings.simplefilter('ignore'); import passt_amd; "
            "net = passt_amd.PaSST(img_size=(128, 250), stride=10, num_classes=7, embed_dim=128, depth=1, num_heads=2, distilled=True); "
            "mel = passt_amd.AugmentMelSTFT(); "
            "assert 'torch._dynamo' not in sys.modules and not passt_amd.passt._OPAQUE['done']; "
            "import torch; f0 = passt_amd.PaSST.forward; c = torch.compile(net); "
            "assert passt_amd.passt._OPAQUE['done'] and passt_amd.PaSST.forward is not f0 and 'torch._dynamo' in sys.modules; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]


def test_torch_compile_leaves_the_forward_opaque():
    from torch._dynamo.utils import counters
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = passt_amd.PaSST(img_size=(128, 250), stride=10, num_classes=37, embed_dim=128, depth=2, num_heads=2,
                              distilled=True, s_patchout_t=6, s_patchout_f=3).train()
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
