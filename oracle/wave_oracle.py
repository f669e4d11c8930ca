"""TEST INFRASTRUCTURE -- CPU restatement of the reference's waveform-side augmentations.

Follows audioset/dataset.py of the reference, in the order its data pipeline applies them to one clip:
  pydub_augment (:102-112)   gain: x * 10**(gain_db / 20), gain_db = randint(2*gain_augment) - gain_augment
  pad_or_truncate (:73-78)   zero pad / cut to clip_length samples
  roll_func (:315-329)       x.roll(shift, axis=1)
  MixupDataset.__getitem__ (:123-137)   l = max(l, 1-l); x1 -= mean; x2 -= mean; x = l x1 + (1-l) x2; x -= mean;
                                        y = l y1 + (1-l) y2
The module itself cannot be imported here (av / h5py / librosa are not installed), so tests/golden/make_golden.py
executes the REAL function bodies extracted from the reference file with `ast` and commits their outputs as
tests/golden/wave_augment.npz; tests/test_oracle_pinned.py pins this restatement to that fixture.
Parity status: pinned (to outputs of the reference's own code run in this container).
"""
import numpy as np


def pad_or_truncate(x, n):
    x = np.asarray(x, np.float32)
    if len(x) <= n:
        return np.concatenate([x, np.zeros(n - len(x), np.float32)])
    return x[:n]


def gain(x, gain_db):
    return (np.asarray(x, np.float32) * (10 ** (gain_db / 20))).astype(np.float32)


def item(raw, gain_db, shift, L):
    """One clip as the (augmenting) dataset hands it over: (1, L) f32."""
    x = pad_or_truncate(gain(raw, gain_db) if gain_db is not None else raw, L)
    return np.roll(x, shift)[None, :]


def wav_mixup(x1, x2, lam):
    l = max(lam, 1.0 - lam)
    x1 = x1 - x1.mean(dtype=np.float32)
    x2 = x2 - x2.mean(dtype=np.float32)
    x = (x1 * np.float32(l) + x2 * np.float32(1.0 - l)).astype(np.float32)
    return x - x.mean(dtype=np.float32), l


def augment_batch(raws, gain_db, shift, partner, lam, L):
    """raws: list of 1-D f32 arrays (any length).  partner[b] < 0: clip b is not mixed.  Returns (B, L) f32 and the
    effective mixing weights (1.0 where not mixed)."""
    items = [item(r, g, s, L)[0] for r, g, s in zip(raws, gain_db, shift)]
    out = np.empty((len(raws), L), np.float32)
    w = np.ones(len(raws), np.float32)
    for b, x in enumerate(items):
        if partner[b] < 0:
            out[b] = x
        else:
            out[b], w[b] = wav_mixup(x, items[partner[b]], float(lam[b]))
    return out, w
