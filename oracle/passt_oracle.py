"""CPU restatement of the kkoutini/PaSST training hot path (TEST INFRASTRUCTURE -- never
imported by the product package ``passt_amd``).

Plain functional torch-CPU code (fp32 by default, fp64 on request), one function per
reference symbol, each citing the reference file:line it follows.  Backward comes from
autograd over this restatement.

Pinning status
--------------
* Model path (patch-embed .. head, fwd+bwd): PINNED -- checked against the real reference
  classes imported from /root/reference (tests/test_oracle_pinned.py) and against the
  committed fixtures ``tests/golden/*.npz`` produced by ``tests/golden/make_golden.py``.
* Front end: the STFT/power/log/affine chain is PINNED the same way (the reference's own
  ``AugmentMelSTFT.forward`` runs with ``torch.stft``).  Two pieces live in the un-vendored
  dependency torchaudio 0.13.1 (environment.yml:99): ``compliance.kaldi.get_mel_banks`` and
  ``functional.mask_along_axis``.  They are restated here from the published algorithm;
  ``get_mel_banks`` is cross-checked against the independent
  ``transformers.audio_utils.mel_filter_bank(mel_scale="kaldi")``; ``mask_along_axis`` has no
  offline cross-check => **parity unpinned for SpecAugment masks** (train-mode freqm/timem).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Front end  (reference: models/preprocess.py)
# --------------------------------------------------------------------------------------


def kaldi_get_mel_banks(num_bins, window_length_padded, sample_freq, low_freq, high_freq,
                        vtln_low=100.0, vtln_high=-500.0, vtln_warp_factor=1.0):
    """Restatement of torchaudio.compliance.kaldi.get_mel_banks (torchaudio 0.13.1; call site
    models/preprocess.py:71-72, vtln_warp_factor == 1.0 => no warping).  fp32 tensor math,
    python-double scalars, exactly like the original.  Returns (bins[num_bins, n_fft/2], center_freqs)."""
    assert num_bins > 3 and window_length_padded % 2 == 0 and vtln_warp_factor == 1.0
    num_fft_bins = window_length_padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    assert 0.0 <= low_freq < nyquist and 0.0 < high_freq <= nyquist and low_freq < high_freq
    fft_bin_width = sample_freq / window_length_padded
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    mel_delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left_mel = mel_low + b * mel_delta
    center_mel = mel_low + (b + 1.0) * mel_delta
    right_mel = mel_low + (b + 2.0) * mel_delta
    center_freqs = 700.0 * ((center_mel / 1127.0).exp() - 1.0)
    mel = (1127.0 * (1.0 + (fft_bin_width * torch.arange(num_fft_bins)) / 700.0).log()).unsqueeze(0)
    up = (mel - left_mel) / (center_mel - left_mel)
    down = (right_mel - mel) / (right_mel - center_mel)
    bins = torch.max(torch.zeros(1), torch.min(up, down))
    return bins, center_freqs


def mask_along_axis(specgram, mask_param, mask_value, axis):
    """Restatement of torchaudio.functional.mask_along_axis (0.13.1 with p = 1.0; 0.11.0 has no p at all), the path taken
    by FrequencyMasking/TimeMasking(iid_masks=True) on the reference's 3-D (B, mel, T) input
    (models/preprocess.py:50,54,81-82).  Consumes two CPU ``torch.rand(1)``.
    mask_param is NOT clamped to the axis length: 0.13.1's ``_get_mask_param`` returns it unchanged for p == 1.0 (the
    transforms' default, which the reference uses) and 0.11.0 never clamps; for an axis shorter than mask_param the band
    start ``rand * (size - value)`` can therefore be negative (truncated toward zero by ``.long()``) and the band can
    cover the whole axis.  torchaudio's source is not on this machine: this function remains a restatement (parity
    UNPINNED)."""
    assert axis in (1, 2)
    if mask_param < 1:
        return specgram
    shape = specgram.size()
    specgram = specgram.reshape([-1] + list(shape[-2:]))
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (specgram.size(axis) - value)
    mask_start = (min_value.long()).squeeze()
    mask_end = (min_value.long() + value.long()).squeeze()
    mask = torch.arange(0, specgram.shape[axis], device=specgram.device, dtype=specgram.dtype)
    mask = (mask >= mask_start) & (mask < mask_end)
    if axis == 1:
        mask = mask.unsqueeze(-1)
    specgram = specgram.masked_fill(mask, mask_value)
    return specgram.reshape(shape[:-2] + specgram.shape[-2:])


def draw_mask_params(mask_param, size):
    """The (start, end) a ``mask_along_axis`` call draws -- same two ``torch.rand(1)`` calls (no clamp, see above)."""
    if mask_param < 1:
        return 0, 0
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (size - value)
    s = int(min_value.long())
    return s, s + int(value.long())


MEL_DEFAULTS = dict(n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, freqm=48,
                    timem=192, fmin=0.0, fmax=None, fmin_aug_range=1, fmax_aug_range=1000)


def resolve_fmax(sr, fmax, fmax_aug_range):
    """models/preprocess.py:32-35."""
    return sr // 2 - fmax_aug_range // 2 if fmax is None else fmax


def stft_power(x, n_fft=1024, hop=320, win_length=800, dtype=torch.float32):
    """models/preprocess.py:59-62 restated without torch.stft: pre-emphasis (valid conv with
    [-0.97, 1]), reflect pad n_fft/2, frames of n_fft every hop, hann(win_length,
    periodic=False) zero-padded *centred* to n_fft, one-sided DFT, power = re^2 + im^2.
    x: (B, L) -> (B, n_fft/2+1, 1 + (L-1)//hop)."""
    x = x.to(dtype)
    y = x[:, 1:] - 0.97 * x[:, :-1]
    pad = n_fft // 2
    yp = F.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    frames = yp.unfold(1, n_fft, hop)                           # (B, T, n_fft)
    win = torch.hann_window(win_length, periodic=False, dtype=dtype)
    left = (n_fft - win_length) // 2
    w = F.pad(win, (left, n_fft - win_length - left))
    spec = torch.fft.rfft(frames * w, dim=-1)                   # (B, T, n_fft/2+1)
    p = spec.real ** 2 + spec.imag ** 2
    return p.transpose(1, 2).contiguous()


def mel_frontend(x, training=False, n_mels=128, sr=32000, win_length=800, hopsize=320,
                 n_fft=1024, freqm=48, timem=192, fmin=0.0, fmax=None, fmin_aug_range=1,
                 fmax_aug_range=1000, dtype=torch.float32, return_aux=False):
    """AugmentMelSTFT.forward (models/preprocess.py:57-86).  RNG consumption order is the
    reference's: randint(fmin_aug_range), randint(fmax_aug_range) (always, :63-64), then in
    training 2x rand(1) for the frequency mask and 2x rand(1) for the time mask (:80-82)."""
    fmax = resolve_fmax(sr, fmax, fmax_aug_range)
    p = stft_power(x, n_fft, hopsize, win_length, dtype)
    fmin_d = fmin + torch.randint(fmin_aug_range, (1,)).item()
    fmax_d = fmax + fmax_aug_range // 2 - torch.randint(fmax_aug_range, (1,)).item()
    if not training:
        fmin_d, fmax_d = fmin, fmax
    basis, _ = kaldi_get_mel_banks(n_mels, n_fft, sr, fmin_d, fmax_d, 100.0, -500.0, 1.0)
    basis = F.pad(basis, (0, 1), value=0.0).to(dtype)           # (n_mels, n_fft/2+1)  :73
    mel = torch.matmul(basis, p)                                # :76
    mel = (mel + 0.00001).log()                                 # :78
    aux = dict(fmin=fmin_d, fmax=fmax_d)
    if training:
        if freqm:
            mel = mask_along_axis(mel, freqm, 0.0, 1)
        if timem:
            mel = mask_along_axis(mel, timem, 0.0, 2)
    mel = (mel + 4.5) / 5.0                                     # :84
    return (mel, aux) if return_aux else mel


# --------------------------------------------------------------------------------------
# Model  (reference: models/passt.py)
# --------------------------------------------------------------------------------------

def make_cfg(embed_dim=768, depth=12, num_heads=12, num_classes=527, img_size=(128, 998),
             stride=(10, 10), patch=16, s_patchout_t=0, s_patchout_f=0, u_patchout=0):
    """Shape bookkeeping of PaSST.__init__/PatchEmbed.__init__ (models/passt.py:304-317, 429-442).
    ``grid`` is the positional-embedding grid, img_size // stride (:311)."""
    return dict(embed_dim=embed_dim, depth=depth, num_heads=num_heads, num_classes=num_classes,
                img_size=tuple(img_size), stride=tuple(stride), patch=patch,
                grid=(img_size[0] // stride[0], img_size[1] // stride[1]),
                s_patchout_t=s_patchout_t, s_patchout_f=s_patchout_f, u_patchout=u_patchout)


def draw_patchout(cfg, F_dim, T_dim, training):
    """The index draws of forward_features in the reference's order (models/passt.py:513-553;
    SURVEY.md App. C): optional randint for the time-pos-embed offset, randperm(T), randperm(F),
    optional randperm(S).  Returns dict(toff, idx_t, idx_f, idx_u) with None where not drawn.
    T_dim is the *pre-cut* grid width (:509), exactly like the reference."""
    Tpe = cfg["grid"][1]
    out = dict(toff=0, idx_t=None, idx_f=None, idx_u=None, T_eff=T_dim)
    if T_dim < Tpe:
        if training:
            out["toff"] = torch.randint(1 + Tpe - T_dim, (1,)).item()
    else:
        out["T_eff"] = Tpe                                      # x is cut to the embedding (:526)
    T_cur, F_cur = out["T_eff"], F_dim
    if training and cfg["s_patchout_t"]:
        out["idx_t"] = torch.randperm(T_dim)[:T_dim - cfg["s_patchout_t"]].sort().values
        T_cur = out["idx_t"].numel()
    if training and cfg["s_patchout_f"]:
        out["idx_f"] = torch.randperm(F_dim)[:F_dim - cfg["s_patchout_f"]].sort().values
        F_cur = out["idx_f"].numel()
    if training and cfg["u_patchout"]:
        S = F_cur * T_cur
        out["idx_u"] = torch.randperm(S)[:S - cfg["u_patchout"]].sort().values
    return out


def attention(x, qkv_w, qkv_b, proj_w, proj_b, num_heads):
    """Attention.forward (models/passt.py:343-361); scale applied after QK^T (:348)."""
    B, N, C = x.shape
    dh = C // num_heads
    qkv = F.linear(x, qkv_w, qkv_b).reshape(B, N, 3, num_heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, proj_w, proj_b)


def block(x, sd, p, num_heads):
    """Block.forward (models/passt.py:377-380), LayerNorm eps 1e-6 (:426), exact-erf GELU (:286)."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    x = x + attention(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"],
                      sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], num_heads)
    h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.gelu(h)
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h


def passt_forward(sd, x, cfg, training=False, draws=None):
    """PaSST.forward (models/passt.py:576-595) incl. forward_features (:506-574), in the
    reference's own order: embed ALL patches, add positional terms, then discard.
    sd: dict name -> tensor (state_dict schema).  x: (B,1,F,T).  Returns (logits, features).
    ``draws`` (from draw_patchout) may be passed to replay indices; otherwise drawn here."""
    D = cfg["embed_dim"]
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"],
                 stride=cfg["stride"])                                               # :323
    B, _, F_dim, T_dim = x.shape
    if draws is None:
        draws = draw_patchout(cfg, F_dim, T_dim, training)
    tpe = sd["time_new_pos_embed"]
    if T_dim < tpe.shape[-1]:
        tpe = tpe[:, :, :, draws["toff"]:draws["toff"] + T_dim]                      # :514-521
    else:
        x = x[:, :, :, :tpe.shape[-1]]                                               # :526
    x = x + tpe                                                                      # :527
    x = x + sd["freq_new_pos_embed"]                                                 # :529
    if draws["idx_t"] is not None:
        x = x[:, :, :, draws["idx_t"]]                                               # :536
    if draws["idx_f"] is not None:
        x = x[:, :, draws["idx_f"], :]                                               # :542
    x = x.flatten(2).transpose(1, 2)                                                 # :546
    if draws["idx_u"] is not None:
        x = x[:, draws["idx_u"], :]                                                  # :552
    cls = sd["cls_token"].expand(B, -1, -1) + sd["new_pos_embed"][:, :1, :]          # :557
    dist = sd["dist_token"].expand(B, -1, -1) + sd["new_pos_embed"][:, 1:, :]        # :562
    x = torch.cat((cls, dist, x), dim=1)                                             # :564
    for i in range(cfg["depth"]):
        x = block(x, sd, f"blocks.{i}.", cfg["num_heads"])
    x = F.layer_norm(x, (D,), sd["norm.weight"], sd["norm.bias"], 1e-6)              # :570
    feat = (x[:, 0] + x[:, 1]) / 2                                                   # :583
    h = F.layer_norm(feat, (D,), sd["head.0.weight"], sd["head.0.bias"], 1e-5)       # :463
    logits = F.linear(h, sd["head.1.weight"], sd["head.1.bias"])                     # :464
    return logits, feat


# --------------------------------------------------------------------------------------
# Training-step glue in the caller  (reference: ex_audioset.py:155-198, helpers/mixup.py)
# --------------------------------------------------------------------------------------

def my_mixup(size, alpha):
    """helpers/mixup.py:5-12 -- torch.randperm then numpy beta; lam = max(l, 1-l)."""
    rn_indices = torch.randperm(size)
    lambd = np.random.beta(alpha, alpha, size).astype(np.float32)
    lambd = np.concatenate([lambd[:, None], 1 - lambd[:, None]], 1).max(1)
    return rn_indices, torch.FloatTensor(lambd)


def mixup_apply(x, y, rn_indices, lam):
    """ex_audioset.py:173-183."""
    B = x.shape[0]
    xm = x * lam.reshape(B, 1, 1, 1) + x[rn_indices] * (1.0 - lam.reshape(B, 1, 1, 1))
    ym = y * lam.reshape(B, 1) + y[rn_indices] * (1.0 - lam.reshape(B, 1))
    return xm, ym


def bce_loss(logits, target):
    """ex_audioset.py:184-186: BCE-with-logits, reduction none, then mean over (B, C)."""
    return F.binary_cross_entropy_with_logits(logits, target, reduction="none").mean()


def ce_mixup_loss(logits, target, rn_indices=None, lam=None):
    """ex_esc50.py:159-169: class-index targets; with mixup the two cross-entropies (own label, partner's label) are mixed
    per sample with lam / (1 - lam), without mixup plain cross-entropy; mean over the batch."""
    if rn_indices is None:
        return F.cross_entropy(logits, target, reduction="none").mean()
    B = logits.shape[0]
    sl = (F.cross_entropy(logits, target, reduction="none") * lam.reshape(B)
          + F.cross_entropy(logits, target[rn_indices], reduction="none") * (1.0 - lam.reshape(B)))
    return sl.mean()


def to_torch(sd_np, dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
        out[k] = t.requires_grad_(True) if requires_grad else t
    return out
