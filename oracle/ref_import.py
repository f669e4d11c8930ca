"""Import the *real* reference (kkoutini/PaSST at /root/reference) on CPU (TEST INFRASTRUCTURE).

Only usable in the build container: /root/reference does not exist on the GPU box, so this
module is imported solely by ``tests/golden/make_golden.py`` (fixture generation) and by the
``-m "not gpu"`` pinning tests, which skip when the tree is absent.

The reference needs third-party packages that are not installed here (timm, ba3l/sacred,
torchaudio).  We stub exactly the symbols it touches:

* ``timm.models._hub.download_cached_file``  (models/helpers/vit_helpers.py:13-16)
* ``ba3l.ingredients.ingredient.Ingredient`` (models/passt.py:915-922, models/preprocess.py:8-18)
* ``torchaudio.compliance.kaldi.get_mel_banks`` and ``torchaudio.transforms.{Frequency,Time}Masking``
  (models/preprocess.py:50,54,71-72) -- torchaudio 0.13.1 is an un-vendored dependency
  (environment.yml:99); its published algorithm is restated in ``oracle/passt_oracle.py``
  and injected here, so the reference's own ``AugmentMelSTFT.forward`` runs end to end.
"""
import contextlib
import io
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "passt.py"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = {}


def load_reference():
    """Returns (ref_passt_module, ref_preprocess_module)."""
    if _loaded:
        return _loaded["passt"], _loaded["pre"]
    if not reference_available():
        raise RuntimeError("reference tree not present at " + REFERENCE_ROOT)
    import torch
    from . import passt_oracle as O

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models._hub", download_cached_file=lambda *a, **k: None)

    class Ingredient:
        def __init__(self, path):
            self.path = path

        def add_config(self, **kw):
            pass

        def command(self, f=None, **kw):
            return f

    _mod("ba3l")
    _mod("ba3l.ingredients")
    _mod("ba3l.ingredients.ingredient", Ingredient=Ingredient)

    class _AxisMasking(torch.nn.Module):
        def __init__(self, mask_param, axis, iid_masks):
            super().__init__()
            self.mask_param, self.axis, self.iid_masks = mask_param, axis, iid_masks

        def forward(self, specgram, mask_value=0.0):
            # torchaudio 0.13.1 _AxisMasking.forward: iid variant only for 4-D input.
            assert not (self.iid_masks and specgram.dim() == 4)
            return O.mask_along_axis(specgram, self.mask_param, mask_value, self.axis)

    class FrequencyMasking(_AxisMasking):
        def __init__(self, freq_mask_param, iid_masks=False):
            super().__init__(freq_mask_param, 1, iid_masks)

    class TimeMasking(_AxisMasking):
        def __init__(self, time_mask_param, iid_masks=False, p=1.0):
            super().__init__(time_mask_param, 2, iid_masks)

    ta = _mod("torchaudio")
    ta.compliance = _mod("torchaudio.compliance")
    ta.compliance.kaldi = _mod("torchaudio.compliance.kaldi", get_mel_banks=O.kaldi_get_mel_banks)
    ta.transforms = _mod("torchaudio.transforms", FrequencyMasking=FrequencyMasking,
                         TimeMasking=TimeMasking)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    saved = sys.modules.pop("models", None), sys.modules.pop("models.passt", None), \
        sys.modules.pop("models.preprocess", None)
    with contextlib.redirect_stdout(io.StringIO()):
        import importlib
        ref_passt = importlib.import_module("models.passt")
        ref_pre = importlib.import_module("models.preprocess")
    # keep the reference modules under private names and restore whatever 'models' was
    for k in ("models", "models.passt", "models.preprocess", "models.helpers",
              "models.helpers.vit_helpers"):
        m = sys.modules.pop(k, None)
        if m is not None:
            sys.modules["_ref_" + k] = m
    for k, m in zip(("models", "models.passt", "models.preprocess"), saved):
        if m is not None:
            sys.modules[k] = m
    sys.path.remove(REFERENCE_ROOT)
    # the stubs stay referenced by the reference modules; drop them from sys.modules so that other
    # libraries probing e.g. importlib.util.find_spec("torchaudio") do not trip over them
    for k in [k for k in sys.modules if k.split(".")[0] in ("timm", "ba3l", "torchaudio")]:
        if getattr(sys.modules[k], "__spec__", None) is None:
            del sys.modules[k]
    _loaded["passt"], _loaded["pre"] = ref_passt, ref_pre
    return ref_passt, ref_pre


def build_reference_passt(cfg: dict, state_dict: dict):
    """Instantiate the reference ``PaSST`` class (models/passt.py:383) with ``cfg`` and load
    ``state_dict`` (numpy arrays) into it.  stdout chatter (first_RUN prints) is silenced."""
    import torch
    ref_passt, _ = load_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref_passt.PaSST(
            u_patchout=cfg.get("u_patchout", 0), s_patchout_t=cfg.get("s_patchout_t", 0),
            s_patchout_f=cfg.get("s_patchout_f", 0), img_size=tuple(cfg["img_size"]),
            patch_size=cfg.get("patch", 16), stride=tuple(cfg["stride"]), in_chans=1,
            num_classes=cfg["num_classes"], embed_dim=cfg["embed_dim"], depth=cfg["depth"],
            num_heads=cfg["num_heads"], distilled=True)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state_dict.items()}, strict=True)
    return m


def run_silently(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def import_reference_file(relpath: str):
    """Load ONE reference source file (e.g. ``helpers/ramp.py``, ``helpers/mixup.py``) as an anonymous module, with
    the ``ba3l`` stub in place while it executes.  Used to pin small restatements (LR ramps, SWA formula)."""
    import importlib.util
    if not reference_available():
        raise RuntimeError("reference tree not present at " + REFERENCE_ROOT)

    class Ingredient:
        def __init__(self, path):
            self.path = path

        def add_config(self, **kw):
            pass

        def command(self, f=None, **kw):
            return f if f is not None else (lambda g: g)

    stubs = {"ba3l": _mod("ba3l"), "ba3l.ingredients": _mod("ba3l.ingredients"),
             "ba3l.ingredients.ingredient": _mod("ba3l.ingredients.ingredient", Ingredient=Ingredient)}
    if "pytorch_lightning" not in sys.modules:
        # helpers/swa_callback.py:25-27 subclasses Lightning's Callback and raises its MisconfigurationException; Lightning is
        # not installed here.  The callback's arithmetic (update_parameters / avg_fn, :246-268) and its epoch schedule
        # (on_train_epoch_start, :161-197) touch nothing else of Lightning, so a bare base class is all the file needs to run.
        class Callback:
            pass

        class MisconfigurationException(Exception):
            pass

        pl = _mod("pytorch_lightning", LightningModule=object, Trainer=object)
        pl.callbacks = _mod("pytorch_lightning.callbacks", Callback=Callback)
        pl.utilities = _mod("pytorch_lightning.utilities")
        pl.utilities.exceptions = _mod("pytorch_lightning.utilities.exceptions", MisconfigurationException=MisconfigurationException)
        stubs.update({k: sys.modules[k] for k in ("pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.utilities",
                                                  "pytorch_lightning.utilities.exceptions")})
    try:
        spec = importlib.util.spec_from_file_location("_ref_file_" + relpath.replace("/", "_").replace(".", "_"),
                                                      os.path.join(REFERENCE_ROOT, relpath))
        mod = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(mod)
    finally:
        for k in stubs:
            if getattr(sys.modules.get(k), "__spec__", None) is None:
                sys.modules.pop(k, None)
    return mod
