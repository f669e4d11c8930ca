"""Waveform-side augmentation on the GPU, ahead of the fused front end.

The reference applies these per clip inside DataLoader workers (audioset/dataset.py): gain (``pydub_augment``
:102-112), ``pad_or_truncate`` (:73-78), ``roll_func`` (:315-329) and ``MixupDataset`` (:115-140).  Here the
loader only hands over raw clips; ``WaveAugment`` draws the same random variables on the host (same
distributions, per clip, in the reference's per-item order: gain, roll, mix decision, partner, beta) and one
kernel pair (``pa_wave_augment``) produces the ``(B, 1, L)`` batch and the mixed targets, so the batch contract
``(waveform (B,1,L) f32, target (B,C) f32)`` of ``ex_audioset.py:155-160`` is unchanged.  One deliberate
difference: the mixing partner is another clip of the SAME batch (the reference indexes the whole dataset).
"""
import numpy as np
import torch

from . import ops


class WaveAugment:
    def __init__(self, clip_samples=320000, gain_augment=7, roll_shift_range=50, wavmix_rate=0.5, wavmix_beta=2.0):
        self.L, self.gain_augment = int(clip_samples), int(gain_augment)
        self.roll_shift_range, self.wavmix_rate, self.wavmix_beta = int(roll_shift_range), float(wavmix_rate), wavmix_beta

    def draw(self, B):
        """Host-side random parameters of one batch: (gain_db, shift, partner, lam) as numpy arrays."""
        gain_db = np.zeros(B, np.int32)
        shift = np.zeros(B, np.int32)
        partner = np.full(B, -1, np.int32)
        lam = np.ones(B, np.float32)
        for b in range(B):
            if self.gain_augment:
                gain_db[b] = int(torch.randint(self.gain_augment * 2, (1,)).item()) - self.gain_augment   # :108
            if self.roll_shift_range:
                shift[b] = int(np.random.randint(-self.roll_shift_range, self.roll_shift_range + 1))       # :323 (inclusive)
        for b in range(B):
            if self.wavmix_rate and float(torch.rand(1)) < self.wavmix_rate:                               # :124
                partner[b] = int(torch.randint(B, (1,)).item())                                            # :126
                lam[b] = np.random.beta(self.wavmix_beta, self.wavmix_beta)                                # :128
        return gain_db, shift, partner, lam

    def __call__(self, raw, target=None, lengths=None, params=None):
        """raw (B, ldx) f32 on the GPU (zero padded rows; ``lengths`` int32 valid samples per row).  Returns
        (wave (B,1,L), mixed target or None)."""
        B = raw.shape[0]
        gain_db, shift, partner, lam = params if params is not None else self.draw(B)
        dev = raw.device
        amp = torch.from_numpy((10.0 ** (np.asarray(gain_db, np.float64) / 20.0)).astype(np.float32)).to(dev)
        sh = torch.from_numpy(np.asarray(shift, np.int32)).to(dev)
        pt = torch.from_numpy(np.asarray(partner, np.int32)).to(dev)
        lm = torch.from_numpy(np.asarray(lam, np.float32)).to(dev)
        if lengths is not None:
            lengths = lengths.to(device=dev, dtype=torch.int32)
        wave = ops.wave_augment(raw.contiguous().float(), self.L, lengths, amp, sh, pt, lm)
        if target is not None:                                                 # y = w y1 + (1 - w) y2, :137
            w = np.where(np.asarray(partner) >= 0, np.maximum(lam, 1.0 - np.asarray(lam)), 1.0).astype(np.float32)
            perm = np.where(np.asarray(partner) >= 0, partner, np.arange(B)).astype(np.int32)
            target = ops.mixup(target.contiguous().float(), torch.from_numpy(perm).to(dev), torch.from_numpy(w).to(dev))
        return wave.unsqueeze(1), target
