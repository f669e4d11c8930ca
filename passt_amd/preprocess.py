"""AugmentMelSTFT on MI355X: drop-in for the reference's ``models/preprocess.py``.

Same constructor kwargs, buffers (non-persistent => empty ``state_dict``), RNG consumption order
and ``forward(x: (B, L)) -> (B, n_mels, 1 + (L-1)//hop)`` as the reference class
(models/preprocess.py:19-86).  The whole chain -- pre-emphasis, STFT, power, kaldi mel filterbank,
log, SpecAugment masks, normalisation -- is ONE fused HIP kernel (pa_mel_frontend_fwd); the host only
draws the random numbers (same torch CPU RNG calls, same order) and passes five scalars.
"""
import math

import torch
import torch.nn as nn

from . import ops
from ._lib import MelParams, PasstAmdError, compile_opaque


def _draw_mask(mask_param, size):
    """The band [start, end) torchaudio.functional.mask_along_axis draws (0.13.1 with the transforms' default p = 1.0 --
    ``_get_mask_param`` leaves mask_param unclamped there -- and 0.11.0, which has no clamp at all; non-iid path, because the
    reference hands the transforms a 3-D tensor: SURVEY.md App. A.4).  Two CPU ``torch.rand(1)`` draws; for an axis shorter
    than mask_param the start can be negative and the band can cover the whole axis, as in torchaudio."""
    if mask_param < 1:
        return 0, 0
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (size - value)
    start = int(min_value.long())
    return start, start + int(value.long())


class _AxisMasking(nn.Module):
    """What ``self.freqm`` / ``self.timem`` are in the reference (models/preprocess.py:47-54: torchaudio.transforms.
    FrequencyMasking / TimeMasking(param, iid_masks=True), nn.Identity for 0): parameter-free modules carrying
    ``mask_param`` / ``axis`` / ``iid_masks``, so ``print(mel)``, ``mel.freqm.mask_param`` and ``isinstance(mel.freqm,
    nn.Identity)`` read as they do there.  The masking itself is a predicate inside the fused front-end kernel; the module
    only holds the draw (``draw(size)``), called by AugmentMelSTFT.forward in the reference's order."""

    def __init__(self, mask_param, axis, iid_masks):
        super().__init__()
        self.mask_param, self.axis, self.iid_masks, self.p = int(mask_param), axis, iid_masks, 1.0

    def draw(self, size):
        return _draw_mask(self.mask_param, size)

    def forward(self, specgram, mask_value=0.0):
        raise PasstAmdError("the SpecAugment masks are applied inside pa_mel_frontend_fwd (AugmentMelSTFT.forward); this module "
                            "only carries mask_param")


class FrequencyMasking(_AxisMasking):
    def __init__(self, freq_mask_param, iid_masks=False):
        super().__init__(freq_mask_param, 1, iid_masks)


class TimeMasking(_AxisMasking):
    def __init__(self, time_mask_param, iid_masks=False):
        super().__init__(time_mask_param, 2, iid_masks)


class AugmentMelSTFT(nn.Module):
    def __init__(self, n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, freqm=48, timem=192,
                 htk=False, fmin=0.0, fmax=None, norm=1, fmin_aug_range=1, fmax_aug_range=1000):
        torch.nn.Module.__init__(self)
        self.win_length, self.n_mels, self.n_fft, self.sr, self.htk, self.fmin = win_length, n_mels, n_fft, sr, htk, fmin
        if fmax is None:
            fmax = sr // 2 - fmax_aug_range // 2
            print(f"Warning: FMAX is None setting to {fmax} ")
        self.fmax, self.norm, self.hopsize = fmax, norm, hopsize
        self.register_buffer('window', torch.hann_window(win_length, periodic=False), persistent=False)
        assert fmin_aug_range >= 1, f"fmin_aug_range={fmin_aug_range} should be >=1; 1 means no augmentation"
        assert fmax_aug_range >= 1, f"fmax_aug_range={fmax_aug_range} should be >=1; 1 means no augmentation"
        self.fmin_aug_range, self.fmax_aug_range = fmin_aug_range, fmax_aug_range
        self.register_buffer("preemphasis_coefficient", torch.as_tensor([[[-.97, 1]]]), persistent=False)
        self.freqm = nn.Identity() if freqm == 0 else FrequencyMasking(freqm, iid_masks=True)      # :47-50
        self.timem = nn.Identity() if timem == 0 else TimeMasking(timem, iid_masks=True)           # :51-54
        # constant tables of the fused kernel (f64 -> f32), non-persistent like the reference buffers
        left = (n_fft - win_length) // 2
        wpad = torch.zeros(n_fft)
        wpad[left:left + win_length] = self.window                      # torch.stft centres the window
        k = torch.arange(n_fft // 2, dtype=torch.float64)
        bin_mel = 1127.0 * torch.log1p(k * (sr / n_fft) / 700.0)         # kaldi mel of FFT bin k
        ang = 2.0 * math.pi * k / n_fft
        tw = torch.stack([torch.cos(ang), -torch.sin(ang)], dim=1)
        self.register_buffer("_window_padded", wpad, persistent=False)
        self.register_buffer("_bin_mel", bin_mel.float(), persistent=False)
        self.register_buffer("_twiddle", tw.float().contiguous(), persistent=False)

    @compile_opaque                 # ONE opaque eager call under torch.compile, like PaSST.forward
    def forward(self, x):
        if not x.is_cuda:
            raise PasstAmdError("passt_amd.AugmentMelSTFT runs on a HIP device only (no CPU fallback)")
        if x.dim() != 2:
            raise ValueError("expected (batch, samples)")
        x = x.contiguous().float()
        B, L = x.shape
        # RNG order of the reference: both randint calls always execute (:63-64)
        fmin = self.fmin + torch.randint(self.fmin_aug_range, (1,)).item()
        fmax = self.fmax + self.fmax_aug_range // 2 - torch.randint(self.fmax_aug_range, (1,)).item()
        if not self.training:
            fmin, fmax = self.fmin, self.fmax
        p = MelParams()
        p.n_fft, p.hop, p.n_mels = self.n_fft, self.hopsize, self.n_mels
        p.n_frames = 1 + (L - 1) // self.hopsize
        p.preemph = 0.97
        mel_low = 1127.0 * math.log(1.0 + fmin / 700.0)
        mel_high = 1127.0 * math.log(1.0 + fmax / 700.0)
        p.mel_low = mel_low
        p.inv_mel_delta = (self.n_mels + 1) / (mel_high - mel_low)
        p.log_eps, p.out_add, p.out_scale = 0.00001, 4.5, 1.0 / 5.0
        p.fmask_start = p.fmask_end = p.tmask_start = p.tmask_end = 0
        if self.training:
            if isinstance(self.freqm, _AxisMasking):
                p.fmask_start, p.fmask_end = self.freqm.draw(self.n_mels)                 # :81
            if isinstance(self.timem, _AxisMasking):
                p.tmask_start, p.tmask_end = self.timem.draw(p.n_frames)                  # :82
        return ops.mel_frontend(x, self._window_padded, self._bin_mel, self._twiddle, p)

    def extra_repr(self):
        return 'winsize={}, hopsize={}'.format(self.win_length, self.hopsize)


try:  # reference: ``model_ing = Ingredient("spectrograms")`` with AugmentMelSTFT as a command (:10,:18)
    from ba3l.ingredients.ingredient import Ingredient  # type: ignore

    model_ing = Ingredient("spectrograms")
    AugmentMelSTFT = model_ing.command(AugmentMelSTFT)
except Exception:
    model_ing = None
