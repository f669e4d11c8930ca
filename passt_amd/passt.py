"""PaSST on MI355X: drop-in for the reference's ``models/passt.py``.

Same public surface as the reference (kkoutini/PaSST, file:line below are the reference's):
``PaSST`` (models/passt.py:383) with identical constructor kwargs, ``state_dict`` keys/shapes
(SURVEY.md App. D), parameter order and ``forward(x) -> (logits, features)`` (:576-595);
``get_model`` (:957-1018) / ``get_model_passt`` (hear21passt alias), ``get_ensemble_model``
(:1039-1045), ``EnsembelerModel`` (:1021-1037), ``fix_embedding_layer``/``lighten_model``
(:924-954) and a ``model_ing`` ingredient when ``ba3l`` is importable.

Underneath, nothing of torch's operator library is used on the hot path: ``forward`` /
``backward`` sequence the hand-written gfx950 kernels of libpasst_amd.so (include/passt_amd.h)
through one ``torch.autograd.Function``; the sub-modules below (``nn.Linear`` ...) are
*parameter containers only*, kept so that ``state_dict()``, ``.parameters()`` order,
``deepcopy`` (SWA), ``print(model)`` and checkpoint loading behave exactly like the reference.

Precision: plain call = exact-f32 MFMA path (matches the reference's fp32 forward/backward
<= 1e-3 rel); inside ``torch.autocast("cuda")`` or with ``model.precision = "bf16"`` = bf16 MFMA
with f32 accumulation, f32 residual stream / LayerNorm / softmax (the reference's AMP regime).
Patchout indices are drawn with the reference's own torch CPU RNG calls in the reference's order
(:513-553), hence bit-exact.
"""
import math
import os
import warnings
from collections import OrderedDict
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import EPI_DGELU, EPI_RESID, EPI_STORE, PA_BF16, PA_F32, PasstAmdError, compile_opaque


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


# --------------------------------------------------------------------------------------------
# parameter containers (never executed on the product path)
# --------------------------------------------------------------------------------------------
class PatchEmbed(nn.Module):
    """Parameter container mirroring models/passt.py:298-328 (Conv2d 1->D, k=patch, stride)."""

    def __init__(self, img_size=224, patch_size=16, stride=16, in_chans=3, embed_dim=768, norm_layer=None,
                 flatten=True):
        super().__init__()
        self.img_size = to_2tuple(img_size)
        self.patch_size = to_2tuple(patch_size)
        self.stride = to_2tuple(stride)
        self.grid_size = (self.img_size[0] // self.stride[0], self.img_size[1] // self.stride[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.embed_dim = embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.stride)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)


def _init_vit_weights(module):
    """models/passt.py:598-629 as reached through ``self.apply`` (no name => generic branch)."""
    if isinstance(module, nn.Linear):
        nn.init.trunc_normal_(module.weight, std=.02)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d)):
        nn.init.zeros_(module.bias)
        nn.init.ones_(module.weight)


# --------------------------------------------------------------------------------------------
# host-side index logic (bit-exact with the reference: same torch CPU RNG calls, same order)
# --------------------------------------------------------------------------------------------
def draw_patchout(model, F_dim, T_dim):
    """models/passt.py:513-553.  Returns (toff, T_eff, idx_t, idx_f, idx_u) (numpy / None)."""
    Tpe = model.time_new_pos_embed.shape[-1]
    toff, T_eff = 0, T_dim
    if T_dim < Tpe:
        if model.training:
            toff = torch.randint(1 + Tpe - T_dim, (1,)).item()                         # :516
    else:
        warnings.warn(f"the patches shape:{(F_dim, T_dim)} are larger than the expected time encodings "
                      f"{tuple(model.time_new_pos_embed.shape)}, x will be cut")         # :524-526
        T_eff = Tpe
    idx_t = idx_f = idx_u = None
    T_cur, F_cur = T_eff, F_dim
    if model.training and model.s_patchout_t:
        idx_t = torch.randperm(T_dim)[:T_dim - model.s_patchout_t].sort().values.numpy()   # :535
        if idx_t.size and idx_t.max() >= T_eff:
            raise IndexError("time patchout index out of range (reference would fail at models/passt.py:536)")
        T_cur = idx_t.size
    if model.training and model.s_patchout_f:
        idx_f = torch.randperm(F_dim)[:F_dim - model.s_patchout_f].sort().values.numpy()   # :541
        F_cur = idx_f.size
    if model.training and model.u_patchout:
        S = F_cur * T_cur
        idx_u = torch.randperm(S)[:S - model.u_patchout].sort().values.numpy()             # :551
    return toff, T_eff, idx_t, idx_f, idx_u


def kept_patches(F_dim, T_eff, idx_t, idx_f, idx_u):
    """Grid coordinates of the surviving patches in sequence order (frequency-major flatten, :546)."""
    ts = np.arange(T_eff) if idx_t is None else idx_t
    fs = np.arange(F_dim) if idx_f is None else idx_f
    pf = np.repeat(fs, ts.size)
    pt = np.tile(ts, fs.size)
    if idx_u is not None:
        pf, pt = pf[idx_u], pt[idx_u]
    return pf.astype(np.int32), pt.astype(np.int32)


# --------------------------------------------------------------------------------------------
# the kernel sequence
# --------------------------------------------------------------------------------------------
class _Staged:
    """Per-dtype GEMM-ready copies of the weights (bf16 cast and/or transposes), re-made only when a
    parameter changed (``_version`` bump by the optimizer / load_state_dict).  The first time a copy is
    asked for it is made on its own; from then on a stale copy triggers ONE batched launch
    (pa_stage_weights) that refreshes every known copy of that dtype in place -- after an optimizer step
    all of them are stale together."""

    def __init__(self):
        self.cache = {}     # (id(p), dtype, transposed) -> [version, tensor, p]
        self.tables = {}    # dtype -> (signature, device table, n, tiles, keys)
        self.epoch = 0      # bumped by optimizers that update parameters through raw pointers
        self.gen = 0        # bumped whenever a copy tensor is created: fused optimizer tables (adamw_table) are keyed on it
        self.fused = {}     # (span key, dtype) -> (gen, device table, n, items, cache keys covered)

    def _version(self, p):
        return (p._version, p.data_ptr(), self.epoch)

    def _refresh_all(self, dtype):
        if os.environ.get("PASST_AMD_NO_BATCH_STAGE"):          # A/B knob: one launch per copy, as on first use
            return False
        tab = self.stage_table(dtype)
        if tab is None:
            return False
        ops.stage_weights(tab[1], tab[2], tab[3], dtype)
        for key, ent in self.cache.items():
            if key[1] == dtype:
                ent[0] = self._version(ent[2])
        return True

    def stage_table(self, dtype):
        """(signature, device table, n, tiles) of the batched refresh of every known copy of ``dtype``; (re)built -- one small
        host-to-device copy -- when the set of copies changed.  TrainStep's graph mode calls this BEFORE it starts capturing: with
        the optimizer rewriting the copies itself (pa_adamw_stage) the eager warm-up steps never need the refresh launch, and a
        copy is not allowed inside a stream capture."""
        groups = {}
        for key, (ver, out, p) in self.cache.items():
            if key[1] == dtype:
                groups.setdefault(key[0], [p, None, None])[2 if key[2] else 1] = out
        if not groups:
            return None
        sig = tuple((pid, g[0].data_ptr(), None if g[1] is None else g[1].data_ptr(),
                     None if g[2] is None else g[2].data_ptr()) for pid, g in groups.items())
        tab = self.tables.get(dtype)
        if tab is None or tab[0] != sig:
            entries = []
            for p, dst, dst_t in groups.values():
                w2 = p.detach().reshape(p.shape[0], -1)
                if not w2.is_contiguous():
                    return None
                entries.append((w2, dst, dst_t))
            tab = self.tables[dtype] = (sig,) + ops.make_stage_table(entries, next(iter(groups.values()))[0].device)
        return tab

    def get(self, p, dtype, transposed):
        if dtype == PA_F32 and not transposed:                  # used in place
            w2 = p.detach().reshape(p.shape[0], -1)
            return w2 if w2.is_contiguous() else w2.contiguous()
        key = (id(p), dtype, transposed)
        ver = self._version(p)
        hit = self.cache.get(key)
        if hit is not None:
            if hit[0] == ver:
                return hit[1]
            if hit[0][1] == ver[1] and self._refresh_all(dtype):     # same storage, new values: batched refresh
                return hit[1]
        w = p.detach()
        w2 = w.reshape(w.shape[0], -1)
        if not w2.is_contiguous():
            w2 = w2.contiguous()
        if transposed:
            out = ops.transpose(w2, dtype)
        else:
            out = ops.convert(w2, dtype)
        self.cache[key] = [ver, out, p]
        self.tables.pop(dtype, None)
        self.gen += 1
        return out

    def adamw_table(self, span_key, params, dtype):
        """The pa_adamw_stage descriptor table of one optimizer launch: ``params`` = [(parameter, element offset in the launch's
        flat buffers)], every parameter of the launch in buffer order.  A parameter with GEMM-ready copies of ``dtype`` in the
        cache gets them rewritten by the optimizer launch itself (no separate pa_stage_weights pass reads the updated values
        back); returns (device table, n, work items, covered cache keys) -- mark_fresh(keys) once the parameters' epoch was bumped."""
        hit = self.fused.get((span_key, dtype))
        if hit is not None and hit[0] == self.gen:
            return hit[1:]
        entries, keys = [], []
        for p, off in params:
            kd, kt = (id(p), dtype, False), (id(p), dtype, True)
            d, t = self.cache.get(kd), self.cache.get(kt)
            ok = (d is not None or t is not None) and p.dim() >= 2 and p.is_contiguous()
            if ok and d is not None and d[2] is not p or ok and t is not None and t[2] is not p:
                ok = False                                      # an id re-used by another tensor: leave it to get()
            if not ok:
                entries.append((off, 1, p.numel(), None, None))
                continue
            rows = p.shape[0]
            entries.append((off, rows, p.numel() // rows, None if d is None else d[1], None if t is None else t[1]))
            keys += [k for k, e in ((kd, d), (kt, t)) if e is not None]
        tab = ops.make_adamw_stage_table(entries, params[0][0].device)
        self.fused[(span_key, dtype)] = (self.gen,) + tab + (keys,)
        return tab + (keys,)

    def mark_fresh(self, keys):
        """The copies under ``keys`` were just rewritten from the current parameter values (fused optimizer launch)."""
        for k in keys:
            ent = self.cache.get(k)
            if ent is not None:
                ent[0] = self._version(ent[2])


def _prefix_rows(model, B, Ntok, device):
    """int32 row indices of the cls / dist tokens in the [B*Ntok] token matrix (cached per shape)."""
    key = ("pidx", B, Ntok, str(device))
    t = model._scratch.get(key)
    if t is None:
        idx = (np.arange(B, dtype=np.int32)[:, None] * Ntok + np.arange(2, dtype=np.int32)[None, :]).reshape(-1)
        t = model._scratch[key] = torch.from_numpy(idx).to(device)
    return t


def _precision(model):
    if model.precision is not None:
        return {"fp32": PA_F32, "f32": PA_F32, "bf16": PA_BF16}[model.precision]
    return PA_BF16 if torch.is_autocast_enabled() else PA_F32


def patchout_draws(model, x_shape):
    """The host part of a forward: geometry checks and the reference's Patchout draws (models/passt.py:513-553, same torch CPU
    RNG calls in the same order), as numpy arrays: the grid coordinates of the kept patches (pf, pt) and the time-positional
    offset.  passt_forward() calls this itself; TrainStep's captured-graph mode calls it ahead of the replay and hands the
    result over in device buffers of fixed address."""
    B, Cin, F, T = x_shape
    P, (fs, ts) = model.patch_embed.patch_size[0], model.patch_embed.stride
    if not (F == model.patch_embed.img_size[0] and T == model.patch_embed.img_size[1]):
        warnings.warn(f"Input image size ({F}*{T}) doesn't match model "
                      f"({model.patch_embed.img_size[0]}*{model.patch_embed.img_size[1]}).")   # :320-321
    F_dim, T_dim = (F - P) // fs + 1, (T - P) // ts + 1
    Fpe = model.freq_new_pos_embed.shape[-2]
    if F_dim != Fpe:
        raise RuntimeError(f"patch grid has {F_dim} frequency rows but freq_new_pos_embed has {Fpe}")
    toff, T_eff, idx_t, idx_f, idx_u = draw_patchout(model, F_dim, T_dim)
    pf_np, pt_np = kept_patches(F_dim, T_eff, idx_t, idx_f, idx_u)
    return dict(pf=pf_np, pt=pt_np, toff=toff, Np=pf_np.size)


def passt_forward(model, x, save, draws=None):
    """Kernel sequence of PaSST.forward (:576-595).  Returns (logits, features, ctx).  ``draws``: device-resident Patchout
    draws {pf, pt, pt_pos (= pt + time offset), Np} prepared by the caller (captured-graph mode); None = draw here."""
    with ops.gemm_flags(getattr(model, "_gemm_flags", 0)):
        return _passt_forward(model, x, save, draws)


def _passt_forward(model, x, save, draws=None):
    if not x.is_cuda:
        raise PasstAmdError("passt_amd.PaSST runs on a HIP device only (no CPU fallback); got a CPU tensor")
    dt = _precision(model)
    object.__setattr__(model, "_last_dt", dt)       # the copies a bound optimizer refreshes (it steps outside torch.autocast)
    st = model._staged
    x = x.contiguous().float()
    if x.dim() != 4 or x.shape[1] != 1:
        raise ValueError(f"PaSST expects a (B, 1, n_mels, frames) spectrogram, got {tuple(x.shape)} "
                         "(in_chans = 1 in every reference arch, models/passt.py:961)")
    B, Cin, F, T = x.shape
    P, (fs, ts) = model.patch_embed.patch_size[0], model.patch_embed.stride
    if draws is None:
        d = patchout_draws(model, x.shape)
        pf = ops.upload_small(d["pf"], x.device)         # page-locked staging: asynchronous H2D
        pt = ops.upload_small(d["pt"], x.device)
        pt_pos, toff, Np = pt, d["toff"], d["Np"]        # the positional kernels add the offset themselves
    else:
        pf, pt, pt_pos, toff, Np = draws["pf"], draws["pt"], draws["pt_pos"], 0, draws["Np"]
    D, H, depth = model.embed_dim, model.num_heads, len(model.blocks)
    Ntok, M = Np + 2, B * (Np + 2)
    scale = (D // H) ** -0.5

    # patch embedding: gather-first im2col GEMM, epilogue adds bias + time/freq positional rows and
    # scatters to token rows 2.. ; rows 0,1 = cls/dist + new_pos_embed                 (:323,:527-564)
    cols = ops.patch_gather(x, pf, pt, P, fs, ts, dt)
    tok = torch.empty((B, Ntok, D), device=x.device, dtype=torch.float32)
    table = ops.patch_pos_table(model.patch_embed.proj.bias, model.time_new_pos_embed, model.freq_new_pos_embed,
                                pf, pt_pos, toff, model.cls_token, model.dist_token, model.new_pos_embed, tok)
    ops.gemm_nt(cols, st.get(model.patch_embed.proj.weight, dt, False), dt, EPI_RESID, resid=table, out_f32=tok,
                row_mod=Np, out_batch_rows=Ntok, out_row_off=2)

    xs = tok.view(M, D)
    saved = []
    nblk = len(model.blocks)
    for bi, blk in enumerate(model.blocks):
        last = bi == nblk - 1
        ln1, mean1, rstd1 = ops.layernorm_fwd(xs, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, dt, save)
        # the q third leaves the GEMM as q * scale * log2(e) (one rounding): attention takes "score - row reference"
        # straight from the matrix pipe (ATTN_Q_PRESCALED); every gradient stays the gradient of the unscaled Linear
        qkv = ops.linear(ln1, st.get(blk.attn.qkv.weight, dt, False), blk.attn.qkv.bias, dt,
                         colscale_n=D, colscale=scale * ops.LOG2E)
        aflags = ops.ATTN_Q_PRESCALED
        if not last:
            att, lse = ops.attention_fwd(qkv, B, H, Ntok, scale, flags=aflags)
            x_res = xs
        else:
            # PREFIX-ONLY TAIL.  The network output reads the last block at the cls/dist rows only
            # (models/passt.py:570-574, 583), so from here on just those 2 rows per clip are computed: attention for
            # 2 queries (keys/values still span every token), then proj / LN2 / MLP on [2B, D].  Exact, not an
            # approximation: the reference computes the other N-2 rows and discards them.
            pidx = _prefix_rows(model, B, Ntok, x.device)
            att, lse = ops.attention_fwd(qkv, B, H, Ntok, scale, nq=2, flags=aflags)
            x_res = ops.gather_rows(xs, pidx)
        x_mid = ops.linear_resid(att, st.get(blk.attn.proj.weight, dt, False), blk.attn.proj.bias, x_res, dt)
        ln2, mean2, rstd2 = ops.layernorm_fwd(x_mid, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, dt, save)
        h_pre, h_act = ops.linear_gelu(ln2, st.get(blk.mlp.fc1.weight, dt, False), blk.mlp.fc1.bias, dt)
        x_out = ops.linear_resid(h_act, st.get(blk.mlp.fc2.weight, dt, False), blk.mlp.fc2.bias, x_mid, dt)
        if save:
            saved.append((xs, ln1, mean1, rstd1, qkv, att, lse, x_mid, ln2, mean2, rstd2, h_pre, h_act))
        xs = x_out
    xl = xs.view(B, 2, D)                  # compact: the two prefix tokens of every clip
    feat, hn, stats = ops.head_pre_fwd(xl, model.norm.weight, model.norm.bias, model.norm.eps, model.head[0].weight,
                                       model.head[0].bias, model.head[0].eps)
    logits = ops.linear_f32_fwd(hn, model.head[1].weight, model.head[1].bias)
    ctx = None
    if save:
        ctx = dict(dt=dt, B=B, Ntok=Ntok, Np=Np, pf=pf, pt=pt_pos, toff=toff, cols=cols, saved=saved, xl=xl, feat=feat,
                   hn=hn, stats=stats, scale=scale)
    return logits, feat, ctx


def _wgrad_pair(dY, X, dW, db, dt, scratch, accumulate):
    """dW[N][K] = dY^T X, db[N] = colsum(dY) from row-major dY[M][N], X[M][K], both read in place
    (pa_gemm_tn: transpose-read MFMA operands, deterministic split-K over tokens)."""
    fused = db is not None and ops.wgrad_tn_fuses_bias(dt) and not os.environ.get("PASST_AMD_NO_FUSED_BIAS")
    scratch["part"] = ops.wgrad_tn(dY, X, dW.view(dY.shape[1], -1), dt, accumulate, scratch.get("part"), db=db if fused else None)
    if db is not None and not fused:
        ops.colsum(dY, db, accumulate=accumulate)


class _SideStream:
    """Weight-gradient GEMMs and bias-gradient reductions are off the input-gradient critical path: they run
    on a second HIP stream so their workgroups interleave with the dgrad / LayerNorm / attention kernels
    (fills tile-quantisation tails, overlaps epilogue write bursts with compute).  ``fork(*tensors)`` orders the
    side stream after everything enqueued so far on the main stream; ``join()`` makes the main stream wait."""

    def __init__(self, device, enabled=True):
        self.enabled = enabled and torch.cuda.is_available()
        self.stream = torch.cuda.Stream(device=device) if self.enabled else None

    def fork(self, *tensors):
        if not self.enabled:
            return torch.cuda.stream(None)
        self.stream.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            if t is not None:
                t.record_stream(self.stream)          # caching-allocator safety across streams
        return torch.cuda.stream(self.stream)

    def join(self):
        if self.enabled:
            torch.cuda.current_stream().wait_stream(self.stream)


def passt_backward(model, ctx, dlogits, dfeat, grads, on_block_done=None):
    """Backward of passt_forward.  ``grads``: dict param-name -> f32 tensor to OVERWRITE.
    ``on_block_done(i)`` is called after block i's parameter gradients are enqueued (i = depth for the
    head, then depth-1 .. 0, then -1 for the patch embedding) -- the hook the data-parallel reducer
    uses to start all-reducing finished buckets while the rest of the backward runs."""
    with ops.gemm_flags(getattr(model, "_gemm_flags", 0)):
        return _passt_backward(model, ctx, dlogits, dfeat, grads, on_block_done)


def _passt_backward(model, ctx, dlogits, dfeat, grads, on_block_done=None):
    dt, B, Ntok, Np = ctx["dt"], ctx["B"], ctx["Ntok"], ctx["Np"]
    st = model._staged
    D, H = model.embed_dim, model.num_heads
    M = B * Ntok
    scratch = model._scratch
    g = grads
    side = scratch.get("side")
    if side is not None and side.enabled != bool(getattr(model, "overlap_wgrad", False)):
        side = None
    if side is None:
        side = scratch["side"] = _SideStream(dlogits.device, enabled=getattr(model, "overlap_wgrad", False))

    def wgrad_async(dY, X, dW, db, done=None):
        """dW, db on the side stream; optionally report block completion from there (DDP bucket launch)."""
        with side.fork(dY, X):
            _wgrad_pair(dY, X, dW, db, dt, scratch, False)
            if done is not None and on_block_done:
                on_block_done(done)

    # bf16, single stream: the four weight gradients of a block wait until the block's last operand exists and go out as
    # ONE batched launch (+ one batched split-K reduction): no drain / prologue between them, equal-sized work items
    pending = [] if (dt == PA_BF16 and not side.enabled and not os.environ.get("PASST_AMD_NO_BATCH_WGRAD")) else None
    # ... and so do the block's small finishing reductions: the two LayerNorms' dgamma | dbeta (+ the bias gradient each carries)
    # and the GELU' epilogue's fc1.bias rows stay as partial rows and are reduced by the SAME finishing launch as the split-K
    # slabs (three launches per block less; PASST_AMD_NO_DEFER_ROWS=1: A/B, every reduction right behind its producer)
    rowjobs = [] if (pending is not None and not os.environ.get("PASST_AMD_NO_DEFER_ROWS")) else None
    # PASST_AMD_BIAS_FROM_WGRAD=1 (A/B; required by PA_EPILOGUE_V3=1): fc1.bias out of the weight-gradient launch instead of the
    # GELU' epilogue's lane-local column sums
    bias_from_wgrad = (dt == PA_BF16 and ops.wgrad_tn_fuses_bias(dt) and not os.environ.get("PASST_AMD_NO_FUSED_BIAS")
                       and (os.environ.get("PASST_AMD_BIAS_FROM_WGRAD") == "1" or os.environ.get("PA_EPILOGUE_V3") == "1"))

    def wgrad(dY, X, dW, db, done=None, fuse=False):
        if pending is None:
            return wgrad_async(dY, X, dW, db, done)
        # the block's last problem (qkv) -- and fc1, `fuse` -- get their bias gradient out of the batched launch itself (a
        # ninth MFMA per phase in the tiles of the first X-column block); the compact fc2 of the last block uses the
        # column-sum kernel right away
        fused_bias = (done is not None or fuse) and db is not None and not os.environ.get("PASST_AMD_NO_FUSED_BIAS")
        pending.append((dY, X, dW.view(dY.shape[1], -1), False, db if fused_bias else None))
        if db is not None and not fused_bias:
            ops.colsum(dY, db)
        if done is not None:
            scratch["part"] = ops.wgrad_tn_batched(pending, dt, scratch.get("part"), row_jobs=rowjobs)
            pending.clear()
            if rowjobs is not None:
                rowjobs.clear()
            if on_block_done:
                on_block_done(done)

    # head: logits = hn W^T + b ; hn = LN_1e-5(feat) ; feat = mean of the two normed prefix tokens
    dhn = ops.linear_f32_bwd(dlogits.contiguous(), ctx["hn"], model.head[1].weight, g["head.1.weight"],
                             g["head.1.bias"])
    dxl, part = ops.head_pre_bwd(dhn, dfeat, ctx["xl"], ctx["feat"], model.norm.weight, model.head[0].weight,
                                 ctx["stats"])
    part4 = part.view(B, 4, D)
    for j, name in enumerate(("head.0.weight", "head.0.bias", "norm.weight", "norm.bias")):
        ops.colsum_f32(part4[:, j, :], g[name])
    if on_block_done:
        on_block_done(len(model.blocks))
    nblk = len(model.blocks)
    dx = dxl.view(2 * B, D)                 # gradient w.r.t. the compact (prefix-rows) output of the last block
    dx_lp = ops.convert(dx, dt)
    for i in range(nblk - 1, -1, -1):
        blk = model.blocks[i]
        last = i == nblk - 1
        pfx = f"blocks.{i}."
        xs, ln1, mean1, rstd1, qkv, att, lse, x_mid, ln2, mean2, rstd2, h_pre, h_act = ctx["saved"][i]
        # ---- MLP:  x_out = x_mid + fc2(gelu(fc1(LN2(x_mid))))      (on [2B, D] rows for the last block)
        # fc2.bias gradient = column sums of dx: already produced by the LayerNorm backward that made dx (the next
        # block's norm1) -- except for the last block, whose dx comes from the head
        wgrad(dx_lp, h_act, g[pfx + "mlp.fc2.weight"], g[pfx + "mlp.fc2.bias"] if last else None)
        if bias_from_wgrad:
            # the fc1.bias gradient (column sums of d_pre) rides in the weight-gradient launch, like qkv.bias (the LDS-free
            # GELU' epilogue -- transposed accumulators, lane = token -- has no lane-local column sums to offer)
            d_pre = ops.dgelu_gemm(dx_lp, st.get(blk.mlp.fc2.weight, dt, True), h_pre, dt)
            wgrad(d_pre, ln2, g[pfx + "mlp.fc1.weight"], g[pfx + "mlp.fc1.bias"], fuse=True)
        else:
            # the fc1.bias gradient (column sums of d_pre) comes out of the GELU' epilogue
            cws = scratch["colsum_ws"] = ops.gemm_colsum_ws(h_pre.shape[0], h_pre.shape[1], dx_lp.device, scratch.get("colsum_ws"))
            d_pre = ops.dgelu_gemm(dx_lp, st.get(blk.mlp.fc2.weight, dt, True), h_pre, dt,
                                   colsum_out=g[pfx + "mlp.fc1.bias"], colsum_ws=cws, defer=rowjobs)
            wgrad(d_pre, ln2, g[pfx + "mlp.fc1.weight"], None)
        d_ln2 = torch.empty_like(ln2)
        ops.gemm_nt(d_pre, st.get(blk.mlp.fc1.weight, dt, True), dt, EPI_STORE, out_lp=d_ln2)
        del d_pre
        dx, dx_lp = ops.layernorm_bwd(d_ln2, x_mid, blk.norm2.weight, mean2, rstd2, dx, g[pfx + "norm2.weight"],
                                      g[pfx + "norm2.bias"], True, dcolsum=g[pfx + "attn.proj.bias"], defer=rowjobs)
        # ---- attention:  x_mid = x_in + proj(attn(qkv(LN1(x_in))))      (proj.bias gradient came out of LN2' above)
        wgrad(dx_lp, att, g[pfx + "attn.proj.weight"], None)
        d_att = torch.empty_like(att)
        ops.gemm_nt(dx_lp, st.get(blk.attn.proj.weight, dt, True), dt, EPI_STORE, out_lp=d_att)
        if not last:
            d_qkv = ops.attention_bwd(qkv, att, d_att, lse, B, H, Ntok, ctx["scale"], flags=ops.ATTN_Q_PRESCALED)
            dres = dx
        else:
            # only 2 queries per sequence carry a gradient; the residual gradient lives on the prefix rows only
            d_qkv = ops.attention_bwd(qkv, att, d_att, lse, B, H, Ntok, ctx["scale"], nq=2, flags=ops.ATTN_Q_PRESCALED)
            dres = ops.scatter_rows_into_zeros(dx, _prefix_rows(model, B, Ntok, dx.device), M)
        d_ln1 = torch.empty_like(ln1)
        ops.gemm_nt(d_qkv, st.get(blk.attn.qkv.weight, dt, True), dt, EPI_STORE, out_lp=d_ln1)
        dx, dx_lp = ops.layernorm_bwd(d_ln1, xs, blk.norm1.weight, mean1, rstd1, dres, g[pfx + "norm1.weight"],
                                      g[pfx + "norm1.bias"], i > 0,
                                      dcolsum=g[f"blocks.{i - 1}.mlp.fc2.bias"] if i > 0 else None, defer=rowjobs)
        # last weight gradient of the block; the side stream (ordered after the LayerNorm gradients above)
        # then reports the block complete, so its all-reduce bucket starts without stalling the main stream
        wgrad(d_qkv, ln1, g[pfx + "attn.qkv.weight"], g[pfx + "attn.qkv.bias"], done=i)
    # ---- patch embedding / positional parameters / prefix tokens
    Tpe, Fpe = model.time_new_pos_embed.shape[-1], model.freq_new_pos_embed.shape[-2]
    dpatch = ops.patch_bwd(dx.view(B, Ntok, D), ctx["pf"], ctx["pt"], ctx["toff"], Tpe, Fpe, g["cls_token"],
                           g["dist_token"], g["new_pos_embed"], g["patch_embed.proj.bias"],
                           g["time_new_pos_embed"], g["freq_new_pos_embed"], dt)
    wgrad_async(dpatch, ctx["cols"], g["patch_embed.proj.weight"], None, done=-1)
    side.join()


class _PasstFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward/backward are kernel sequences, torch only
    sees (x, *parameters) -> (logits, features)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        if x.requires_grad:
            raise NotImplementedError("passt_amd.PaSST does not produce a gradient w.r.t. its input spectrogram (the reference's "
                                      "training never asks for one); detach() the input")
        logits, feat, c = passt_forward(model, x, save=True)
        ctx.model, ctx.c = model, c
        ctx.named, ctx.total = model._graph_params(validate=False)     # the list forward() just handed to apply()
        # bound to a passt_amd.optim.AdamW (PaSST.bind_flat_grads): the only input is a token; the backward writes the gradients
        # straight into the optimizer's persistent flat buffer -- p.grad are views of it -- and hands autograd nothing
        ctx.flat = model._flat if (len(params) == 1 and model._flat is not None and params[0] is model._flat["token"]) else None
        ctx.set_materialize_grads(False)        # an unused `features` output arrives as None, not as a zero tensor
        return logits, feat

    @staticmethod
    def backward(ctx, dlogits, dfeat):
        model, c = ctx.model, ctx.c
        if c is None:
            raise RuntimeError("passt_amd.PaSST: the saved activations of this forward were already consumed by a backward "
                               "pass (retain_graph / double backward are not supported: run the forward again)")
        # gradient buffers for EVERY parameter the backward writes (all but head_dist.*): the kernel sequence produces
        # them all; parameters with requires_grad=False are simply not handed back to autograd (frozen backbone, ...)
        named, total = ctx.named, ctx.total
        fl = ctx.flat
        if fl is not None and fl["fresh"]:
            flat, grads = fl["flat_g"], fl["grads"]     # overwritten in place: the caller zeroed (optimizer.zero_grad()) since the last backward
        else:
            flat = torch.empty(total, device=dlogits.device, dtype=torch.float32)
            # one C++ call makes the 159 views (a Python loop of slice + view costs 1.3 ms of host time in front of the first
            # backward kernel: exposed whenever the caller synchronised in this step, and the reference's mixup does)
            views = torch._C._nn.unflatten_dense_tensors(flat, [p for _, p in named])
            grads = {n: v for (n, _), v in zip(named, views)}
        # a fresh flat buffer per backward: autograd may keep (not copy) the views as .grad
        if dlogits is None:                     # only `features` fed the loss
            dlogits = torch.zeros((c["B"], model.num_classes), device=flat.device, dtype=torch.float32)
        dlogits = dlogits.contiguous()
        dfeat = None if dfeat is None else dfeat.contiguous()
        red = getattr(model, "_ddp", None)
        if red is not None and red.world > 1:
            # passt_amd.ddp.attach(net): this node reduces its own gradients.  `flat` is laid out like the reducer's buckets
            # (named_parameters() order without head_dist.*); every bucket's all-reduce starts from on_block_done while the
            # rest of the backward runs, and the node returns once the current stream is ordered behind the last bucket.
            # Mean over ranks (DDP's semantics) = sum of gradients of loss / world: the backward is linear in (dlogits, dfeat).
            if flat.numel() != red.total:
                raise RuntimeError("passt_amd.ddp.attach: the parameter set changed since attach(); call attach(net) again")
            inv = 1.0 / red.world
            dlogits = dlogits * inv
            dfeat = None if dfeat is None else dfeat * inv
            red.flat = flat
            try:
                passt_backward(model, c, dlogits, dfeat, grads, on_block_done=red.on_block_done)
            finally:
                red.wait()              # also after an exception: no collective may stay in flight on a buffer we drop
        else:
            passt_backward(model, c, dlogits, dfeat, grads)
        ctx.c = None
        if fl is not None:
            if not fl["fresh"]:                 # a second backward without zero_grad (gradient accumulation): add, as AccumulateGrad would
                fl["flat_g"].add_(flat)
            fl["fresh"] = False
            return None, None, None
        out = [None, None]
        for n, p in named:                      # the same list, in the same order, as PaSST.forward handed to apply()
            out.append(grads[n] if p.requires_grad else None)
        return tuple(out)


def _tree_signature(root):
    """[(module, ids of its parameters, ids of its sub-modules)] over the module tree: what PaSST._graph_params() validates its
    cached parameter list against on every forward (~110 modules, two small tuples each: ~30 us).  Local to this model: surgery
    anywhere in ITS tree (net.head[1] = nn.Linear(768, 50), a block's sub-module replaced, a parameter re-registered) changes
    a tuple; nothing is hooked process-wide."""
    return [(m, tuple(map(id, m._parameters.values())), tuple(map(id, m._modules.values()))) for m in root.modules()]


import weakref

_LIVE = weakref.WeakSet()            # live PaSST instances: how passt_amd.optim.AdamW finds the model its parameters belong to
_RUNTIME_ATTRS = ("_staged", "_scratch", "_ddp", "_gemm_flags", "_flat", "_last_dt")


class PaSST(nn.Module):
    """Same constructor as the reference (models/passt.py:391-395)."""

    def __init__(self, u_patchout=0, s_patchout_t=0, s_patchout_f=0, img_size=(128, 998), patch_size=16, stride=16,
                 in_chans=1, num_classes=527, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=True,
                 representation_size=None, distilled=False, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 embed_layer=PatchEmbed, norm_layer=None, act_layer=None, weight_init=''):
        super().__init__()
        if not distilled:
            raise NotImplementedError("passt_amd covers the distilled (cls+dist token) PaSST archs -- every "
                                      "arch get_model() can return (models/passt.py:963-1008)")
        if in_chans != 1 or drop_rate or attn_drop_rate or drop_path_rate or mlp_ratio != 4. or not qkv_bias \
                or representation_size:
            raise NotImplementedError("passt_amd hot path: in_chans=1, no dropout/drop-path, mlp_ratio=4, qkv_bias")
        if embed_dim % num_heads or embed_dim // num_heads != 64:
            raise NotImplementedError("attention kernels are built for head_dim 64 (768/12, 1024/16, 384/6 ...)")
        self.num_classes = num_classes
        self.u_patchout, self.s_patchout_t, self.s_patchout_f = u_patchout, s_patchout_t, s_patchout_f
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.num_tokens = 2
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, stride=stride, in_chans=in_chans,
                                       embed_dim=embed_dim, flatten=False)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.new_pos_embed = nn.Parameter(torch.zeros(1, self.num_tokens, embed_dim))
        self.freq_new_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, self.patch_embed.grid_size[0], 1))
        self.time_new_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, 1, self.patch_embed.grid_size[1]))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, norm_layer=norm_layer,
                  act_layer=act_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Sequential(nn.LayerNorm(self.num_features),
                                  nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity())
        self.head_dist = nn.Linear(self.embed_dim, self.num_classes) if num_classes > 0 else nn.Identity()
        self.precision = None          # None: follow torch.autocast; or "fp32" / "bf16"
        # opt-in: run weight/bias-gradient kernels on a second stream (+2.4 % step throughput measured on MI355X);
        # off by default so that every kernel runs alone and per-kernel timings (bench.py roofline, rocprof) are exact
        self.overlap_wgrad = False
        self.init_weights(weight_init)
        self._reset_runtime()

    # ---- runtime state that must not be deep-copied / pickled (SWA deepcopies the net) ----------
    def _reset_runtime(self):
        object.__setattr__(self, "_staged", _Staged())
        object.__setattr__(self, "_scratch", {})
        object.__setattr__(self, "_ddp", None)          # passt_amd.ddp.attach(): gradient reducer of the autograd path
        object.__setattr__(self, "_gemm_flags", 0)      # pa_gemm_args.reserved bits of this model's GEMM launches
        object.__setattr__(self, "_flat", None)         # bind_flat_grads(): flat gradient buffer owned by a passt_amd.optim.AdamW
        object.__setattr__(self, "_last_dt", None)
        _LIVE.add(self)

    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        import copy
        for k, v in self.__dict__.items():
            if k in _RUNTIME_ATTRS:
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._reset_runtime()
        return new

    def __getstate__(self):
        d = self.__dict__.copy()
        for k in _RUNTIME_ATTRS:
            d.pop(k, None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._reset_runtime()

    def _graph_params(self, validate=True):
        """([(name, parameter)] without head_dist.*, total numel): what the autograd node takes and returns gradients for, in
        named_parameters() order.  Cached (named_parameters() over the tree costs 0.4 ms per call); the cache is validated per
        forward against the identity of every module's parameters and children (``_tree_signature``: the cached list keeps the
        old objects alive, so an id cannot be reused), and dropped when the module is moved / cast (``_apply``).  The backward
        of a forward passes validate=False: it must hand back gradients for exactly the list that forward gave to apply()."""
        hit = self._scratch.get("graph_params")
        if hit is not None and validate:
            for m, pids, mids in hit[2]:
                if tuple(map(id, m._parameters.values())) != pids or tuple(map(id, m._modules.values())) != mids:
                    hit = None
                    break
        if hit is None:
            named = [(n, p) for n, p in self.named_parameters() if not n.startswith("head_dist.")]
            hit = self._scratch["graph_params"] = (named, sum(p.numel() for _, p in named), _tree_signature(self))
        return hit[:2]

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)) and "_scratch" in self.__dict__:
            self._scratch.pop("graph_params", None)
        super().__setattr__(name, value)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if "_scratch" in self.__dict__:
            self._scratch.pop("graph_params", None)
            self.unbind_flat_grads()            # .to() / .cuda() / .float() replaced the storages the flat views pointed into
        return out

    # ---- one flat gradient for the drop-in path (FSDP use_orig_params-style) -------------------------------------
    def bind_flat_grads(self, flat_g):
        """Called by passt_amd.optim.AdamW once it owns this model's parameters as ONE flat buffer: ``flat_g`` (f32, numel = all
        parameters the network produces gradients for, named_parameters() order without head_dist.*) becomes the model's
        persistent gradient buffer.  From then on forward() hands autograd a single token instead of 159 parameters, the backward
        writes every gradient straight into ``flat_g`` and ``p.grad`` of every parameter is a standing view of it: no
        AccumulateGrad nodes, no per-parameter host work in backward / zero_grad / step -- what made the drop-in path host-bound at
        ESC-50's batch of 12 (ex_esc50.py:40).  ``state_dict`` keys, ``parameters()`` order and the parameters themselves are
        untouched.  Gradient semantics: after ``optimizer.zero_grad()`` (the bound optimizer marks the buffer fresh) a backward
        overwrites; a further backward without zero_grad adds, like autograd.  Not for a torch DistributedDataParallel wrapper (its
        reducer hooks AccumulateGrad; use passt_amd.ddp.attach, which reduces this buffer per block)."""
        named, total = self._graph_params()
        if any(not p.requires_grad for _, p in named):
            return None                                  # frozen parameters: autograd decides which gradients exist; stay unbound
        assert flat_g.numel() == total and flat_g.dtype == torch.float32
        views = torch._C._nn.unflatten_dense_tensors(flat_g, [p for _, p in named])
        # gradients that already exist (the optimizer binds inside its first step(): the first backward has run) move in
        have = [p.grad is not None for _, p in named]
        if all(have):
            torch._foreach_copy_(list(views), [p.grad for _, p in named])
        elif any(have):
            return None                                  # a partial set: leave this step to the per-parameter path
        for (n, p), v in zip(named, views):
            p.grad = v
        fl = dict(token=torch.zeros((), device=flat_g.device, requires_grad=True), flat_g=flat_g, grads={n: v for (n, _), v in zip(named, views)},
                  named=named, fresh=not all(have))
        object.__setattr__(self, "_flat", fl)
        return fl

    def unbind_flat_grads(self, keep_grads=False):
        """keep_grads: the gradients of the last backward stay in ``p.grad`` (still views of the old buffer) for the step that is
        about to consume them; parameters that were frozen meanwhile lose theirs."""
        fl = self.__dict__.get("_flat")
        if fl is not None:
            object.__setattr__(self, "_flat", None)
            for _, p in fl["named"]:
                if not (keep_grads and p.requires_grad and not fl["fresh"]):
                    p.grad = None

    @property
    def _grad_names(self):
        return {n for n, p in self.named_parameters() if p.requires_grad and not n.startswith("head_dist.")}

    @property
    def _n_grad_elems(self):
        names = self._grad_names
        return sum(p.numel() for n, p in self.named_parameters() if n in names)

    def init_weights(self, mode=''):
        """models/passt.py:471-484."""
        assert mode in ('jax', 'jax_nlhb', 'nlhb', '')
        nn.init.trunc_normal_(self.new_pos_embed, std=.02)
        nn.init.trunc_normal_(self.freq_new_pos_embed, std=.02)
        nn.init.trunc_normal_(self.time_new_pos_embed, std=.02)
        nn.init.trunc_normal_(self.dist_token, std=.02)
        if mode.startswith('jax'):
            raise RuntimeError("Not supported yet")
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.apply(_init_vit_weights)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'new_pos_embed', 'freq_new_pos_embed', 'time_new_pos_embed', 'cls_token', 'dist_token'}

    def get_classifier(self):
        return self.head, self.head_dist

    def reset_classifier(self, num_classes, global_pool=''):
        raise NotImplementedError("reset_classifier drops the head LayerNorm in the reference "
                                  "(models/passt.py:500-504); build a new model with n_classes instead")

    def forward_features(self, x):
        raise NotImplementedError("only forward() is on the accelerated path (returns (logits, features))")

    def mark_params_updated(self):
        """Call after updating parameters through raw device pointers (passt_amd.optim does)."""
        self._staged.epoch += 1

    @compile_opaque
    def forward(self, x):
        """x: (B,1,F,T) -> (logits (B,C), features (B,D)); always a tuple (models/passt.py:588,595).

        ``torch.compile(net)`` (ex_audioset.py:135, model_speed_test :391): the whole forward is ONE opaque call to the
        compiler (``_lib.compile_opaque``: torch.compiler.disable's mechanism without the torch._dynamo import, installed at class
        definition so that every compile flow meets it) -- there is nothing for Inductor to fuse, every kernel of the network is already
        a hand-written launch behind the C ABI, and dynamo cannot trace ctypes calls; the compiled module therefore runs this
        function eagerly, captures no graph and never recompiles (tests/test_abi_cpu.py, tests/test_gpu_model.py speed-test
        flow).  Under ``torch.autocast`` of either 16-bit type (Lightning precision=16 / torch.cuda.amp.autocast() are fp16) the
        kernels run the bf16 MFMA path with f32 accumulation and return f32 logits / features: bf16 has f32's exponent range,
        so a GradScaler's loss scale flows through the backward without overflow and its inf checks never fire."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # head_dist.* is not part of the graph -- as in the reference, whose forward never touches it
            # (models/passt.py:583-595; hence find_unused_parameters=True under torch DDP there and here)
            named = self._graph_params()[0]
            fl = self._flat
            if fl is not None:
                if fl["named"] is named:                # same validated parameter list as at bind time
                    return _PasstFunction.apply(self, x, fl["token"])
                self.unbind_flat_grads()                # surgery since: the optimizer re-binds at its next step
            return _PasstFunction.apply(self, x, *[p for _, p in named])
        logits, feat, _ = passt_forward(self, x, save=False)
        return logits, feat


# --------------------------------------------------------------------------------------------
# architecture constructors + get_model  (models/passt.py:734-1018)
# --------------------------------------------------------------------------------------------
_ARCHS = {
    # arch name -> (depth, expected stride, note)
    "passt_deit_bd_p16_384": (12, None),
    "passt_s_kd_p16_128_ap486": (12, (10, 10)),
    "passt_l_kd_p16_128_ap47": (7, (10, 10)),
    "passt_s_swa_p16_128_ap476": (12, (10, 10)),
    "passt_s_swa_p16_128_ap4761": (12, (10, 10)),
    "passt_s_p16_128_ap472": (12, (10, 10)),
    "passt_s_p16_s16_128_ap468": (12, (16, 16)),
    "passt_s_swa_p16_s16_128_ap473": (12, (16, 16)),
    "passt_s_swa_p16_s14_128_ap471": (12, (14, 14)),
    "passt_s_p16_s14_128_ap469": (12, (14, 14)),
    "passt_s_swa_p16_s12_128_ap473": (12, (12, 12)),
    "passt_s_p16_s12_128_ap470": (12, (12, 12)),
    "passt_s_f128_20sec_p16_s10_ap474": (12, (10, 10)),
    "passt_s_f128_30sec_p16_s10_ap473": (12, (10, 10)),
}


def fix_embedding_layer(model, embed="default"):
    if embed != "default":
        raise NotImplementedError("only embed='default' is reachable in the reference (models/passt.py:924-931)")
    return model


def lighten_model(model, cut_depth=0):
    """models/passt.py:934-954."""
    if cut_depth == 0:
        return model
    old = list(model.blocks.children())
    if cut_depth < 0:
        old = [old[0]] + old[1:-1:-cut_depth] + [old[-1]]
    else:
        if len(model.blocks) < cut_depth + 2:
            raise ValueError(f"Cut depth a VIT with {len(model.blocks)} layers should be between 1 and "
                             f"{len(model.blocks) - 2}")
        old = [old[0]] + old[cut_depth + 1:]
    model.blocks = nn.Sequential(*old)
    return model


def get_model(arch="passt_s_kd_p16_128_ap486", pretrained=True, n_classes=527, in_channels=1, fstride=10, tstride=10,
              input_fdim=128, input_tdim=998, u_patchout=0, s_patchout_t=0, s_patchout_f=0):
    """Same signature/defaults as models/passt.py:958-961.  ``pretrained=True`` needs a checkpoint
    download (vit_helpers.py:85-91) which this offline build cannot do: pass ``pretrained=False`` and
    ``load_state_dict`` a reference checkpoint (identical keys)."""
    if arch not in _ARCHS:
        raise RuntimeError(f"Unknown model {arch}")
    ckpt = None
    if pretrained:
        # the reference downloads <arch>.pt from its GitHub release (models/helpers/vit_helpers.py:85-91); this build is
        # offline, so the checkpoint must already be on disk: PASST_AMD_CHECKPOINT_DIR/<arch>.pt (same file, same keys)
        ckpt_dir = os.environ.get("PASST_AMD_CHECKPOINT_DIR")
        ckpt = os.path.join(ckpt_dir, arch + ".pt") if ckpt_dir else None
        if not ckpt or not os.path.isfile(ckpt):
            raise RuntimeError(f"pretrained=True: no local checkpoint for {arch} (no network here to download the reference's "
                               f"release file).  Put {arch}.pt into a directory and set PASST_AMD_CHECKPOINT_DIR to it, or use "
                               "pretrained=False and model.load_state_dict(torch.load(<reference .pt>))")
    depth, want = _ARCHS[arch]
    stride = (fstride, tstride)
    if want is not None and stride != want:
        warnings.warn(f"This model was pre-trained with strides {want}, but now you set (fstride,tstride) to {stride}.")
    model = PaSST(u_patchout=u_patchout, s_patchout_t=s_patchout_t, s_patchout_f=s_patchout_f,
                  img_size=(input_fdim, input_tdim), patch_size=16, stride=stride, in_chans=in_channels,
                  num_classes=n_classes, embed_dim=768, depth=depth, num_heads=12, distilled=True)
    if ckpt is not None:
        sd = torch.load(ckpt, map_location="cpu")
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
        if n_classes != 527:       # the reference drops the classifier of the pre-trained net when the label set differs
            sd = {k: v for k, v in sd.items() if not k.startswith(("head.1.", "head_dist."))}
            missing = model.load_state_dict(sd, strict=False)
            assert all(k.startswith(("head.1.", "head_dist.")) for k in missing.missing_keys), missing
        else:
            model.load_state_dict(sd, strict=True)
    model = fix_embedding_layer(model)
    model = lighten_model(model)
    return model


get_model_passt = get_model     # hear21passt's name for the same function (README.md:49,70,77)


class EnsembelerModel(nn.Module):
    """models/passt.py:1021-1037."""

    def __init__(self, models):
        super().__init__()
        self.models = nn.ModuleList(models)

    def forward(self, x):
        all_out = None
        for m in self.models:
            out, _ = m(x)
            all_out = out if all_out is None else out + all_out
        all_out = all_out / len(self.models)
        return all_out, all_out


def get_ensemble_model(arch_list=[], *, pretrained=True):
    """models/passt.py:1039-1045: every member through get_model() with ITS default pretrained=True, i.e. here through the
    local checkpoint directory (PASST_AMD_CHECKPOINT_DIR/<arch>.pt; raises without one: no network).  `pretrained` is a
    keyword-only extension for random-init members (tests, benchmarks)."""
    models_list = [get_model(arch=a, fstride=f, tstride=t, pretrained=pretrained) for a, f, t in arch_list]
    model = EnsembelerModel(models_list)
    print(model)
    return model


try:  # the reference exposes these through a sacred/ba3l ingredient (models/passt.py:915-922)
    from ba3l.ingredients.ingredient import Ingredient  # type: ignore

    model_ing = Ingredient("passt")
    model_ing.add_config(instance_cmd="get_model")
    fix_embedding_layer = model_ing.command(fix_embedding_layer)
    lighten_model = model_ing.command(lighten_model)
    get_model = model_ing.command(get_model)
    get_ensemble_model = model_ing.command(get_ensemble_model)
except Exception:  # ba3l not installed: plain functions
    model_ing = None
