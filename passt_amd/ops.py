"""Tensor-level wrappers over the C ABI (include/passt_amd.h).

Every function takes CUDA(HIP) torch tensors, passes raw device pointers + the current stream to
libpasst_amd.so and returns the output tensors it allocated (torch is the allocator only).
No fallback: a non-CUDA tensor or a missing library raises.
"""
import ctypes as C
import functools
import threading
import os

import torch

from . import _lib
from ._lib import (GEMM_BLOCKED_PRE, EPI_DGELU, EPI_GELU, EPI_PARTIAL, EPI_RESID, EPI_STORE, PA_BF16, PA_F32,
                   GemmArgs, MelParams, check)

TORCH_DTYPE = {PA_F32: torch.float32, PA_BF16: torch.bfloat16}
PA_DTYPE = {torch.float32: PA_F32, torch.bfloat16: PA_BF16}


class _CallState(threading.local):
    """Per-thread (the extension is entered from the trainer thread AND the autograd thread): the device of the
    tensors of the C-ABI call being assembled -- set by _p(), consumed by _stream()."""
    dev = None
    gemm_flags = 0      # ambient pa_gemm_args.reserved bits of the model whose kernel sequence this thread is issuing


_call = _CallState()


class gemm_flags:
    """``with ops.gemm_flags(bits):`` -- every pa_gemm_nt call issued by THIS thread inside the block carries ``bits`` in
    pa_gemm_args.reserved (PA_GEMM_NO_PERSIST for a model whose backward shares the CUs with an all-reduce).  Per model and
    per thread, so one data-parallel TrainStep does not change how other models of the process launch their GEMMs."""

    def __init__(self, bits):
        self.bits = int(bits or 0)

    def __enter__(self):
        self.prev, _call.gemm_flags = _call.gemm_flags, self.bits

    def __exit__(self, *exc):
        _call.gemm_flags = self.prev


def _stream():
    """HIP stream the call is ordered on: the current stream OF THE DEVICE THE CALL'S TENSORS LIVE ON (not of
    torch.cuda.current_device(): a model on cuda:1 without set_device must not launch on cuda:0's stream).  Every
    wrapper passes its tensors through _p() first and _stream() last."""
    dev, _call.dev = _call.dev, None
    return torch.cuda.current_stream(dev).cuda_stream


def _p(t, dtype=None, strided=False):
    """Raw device pointer of ``t`` for the C ABI, after the checks the kernels rely on: HIP tensor, the expected
    element type (``dtype``: a torch dtype, PA_F32 / PA_BF16, or None = any), dense rows (contiguous; ``strided`` =
    only the last dimension must be dense, the leading dimension is passed separately) and the same device as the
    other tensors of the call.  The kernels reinterpret memory: a wrong dtype or a strided view would read garbage or
    out of bounds, so this raises instead."""
    if t is None:
        return None
    if not t.is_cuda:
        _call.dev = None
        raise _lib.PasstAmdError("passt_amd kernels need CUDA/HIP tensors (no CPU fallback)")
    if dtype is not None:
        want = TORCH_DTYPE.get(dtype, dtype)
        if t.dtype != want:
            _call.dev = None
            raise _lib.PasstAmdError(f"expected a {want} tensor, got {t.dtype}")
    if not (t.is_contiguous() or (strided and t.dim() >= 1 and (t.shape[-1] <= 1 or t.stride(-1) == 1))):
        _call.dev = None
        raise _lib.PasstAmdError(f"expected a {'row-dense' if strided else 'contiguous'} tensor, got strides {tuple(t.stride())} "
                                 f"for shape {tuple(t.shape)}")
    if _call.dev is None:
        _call.dev = t.device
    elif t.device != _call.dev:
        d0, _call.dev = _call.dev, None
        raise _lib.PasstAmdError(f"tensors of one kernel call live on different devices: {d0} and {t.device}")
    return t.data_ptr()


class _PinnedRing:
    """Small per-step host arrays (Patchout indices, mixup permutation / lambda) reach the device through page-locked
    staging buffers: a copy from pageable memory is a synchronous staging copy that serialises against the communication
    stream in a data-parallel run.  Four slots per (device, dtype, length) rotate; a slot is reused only after the event
    behind its last copy has fired."""

    MAX_RINGS = 32          # distinct (device, dtype, length) rings kept: variable-length uploads must not pin memory without bound

    def __init__(self, slots=4):
        self.slots, self.rings = slots, {}

    def upload(self, host, device):
        host = torch.as_tensor(host)
        key = (str(device), host.dtype, host.numel())
        ring = self.rings.pop(key, None)            # re-inserted below: the dict is kept in least-recently-used order
        if ring is None:
            while len(self.rings) >= self.MAX_RINGS:
                old = self.rings.pop(next(iter(self.rings)))
                for ev in old["ev"]:                # its copies must have left the page-locked buffers before they are freed
                    if ev is not None:
                        ev.synchronize()
            ring = {"i": 0, "buf": [torch.empty(host.numel(), dtype=host.dtype).pin_memory() for _ in range(self.slots)],
                    "ev": [None] * self.slots}
        self.rings[key] = ring
        i = ring["i"]
        ring["i"] = (i + 1) % self.slots
        if ring["ev"][i] is not None:
            ring["ev"][i].synchronize()
        ring["buf"][i].copy_(host.reshape(-1))
        out = ring["buf"][i].to(device, non_blocking=True).view(host.shape)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ring["ev"][i] = ev
        return out


_pinned = threading.local()


def upload_small(host, device):
    """host (numpy array / CPU tensor) -> device tensor through a pinned staging ring (asynchronous H2D)."""
    ring = getattr(_pinned, "ring", None)
    if ring is None:
        ring = _pinned.ring = _PinnedRing()
    return ring.upload(host, torch.device(device))


# bench.py sets this to a dict to time every GEMM launch with HIP events on the launch stream:
# {epilogue: [(start_event, end_event, algorithmic_flops)]}
GEMM_PROFILE = None
PROFILE_BY_SHAPE = bool(os.environ.get("PASST_AMD_PROFILE_BY_SHAPE"))     # bench.py per_epilogue keyed by (epilogue, M, N, K)
GEMM_TUNE = 0          # pa_gemm_args.tune for every pa_gemm_nt call (0 = library default)
TN_BATCH_ORDER = int(os.environ.get("PASST_AMD_TN_ORDER", "0"))   # 2: problem-major item order (A/B only, see gemm.hip)
# pa_gemm_args.reserved of every pa_gemm_nt call (probe builds: tools/probe_epilogue.py; A/B: PASST_AMD_GEMM_FLAGS=0x1000 =
# _lib.GEMM_EPILOGUE_V3, the LDS-free epilogues)
GEMM_RESERVED = int(os.environ.get("PASST_AMD_GEMM_FLAGS", "0"), 0)
_EPI_NAME = {EPI_STORE: "store", EPI_GELU: "gelu", EPI_RESID: "resid", EPI_DGELU: "dgelu", EPI_PARTIAL: "wgrad_partial"}


def _timed(kind, work, fn):
    """Run fn(); when bench.py profiles this step, bracket it with HIP events on the launch stream and file
    (start, end, work) under `kind` (work = algorithmic FLOPs or bytes of the call)."""
    if GEMM_PROFILE is None:
        return fn()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    r = fn()
    ev1.record()
    GEMM_PROFILE.setdefault(kind, []).append((ev0, ev1, work))
    return r


def kpad(dtype):
    """GEMM K granularity in elements (128 bytes)."""
    return 64 if dtype == PA_BF16 else 32


def round_up(a, b):
    return (a + b - 1) // b * b


# ---- front end -----------------------------------------------------------------------------
def mel_frontend(wave, window, bin_mel, twiddle, params: MelParams):
    B, L = wave.shape
    out = torch.empty((B, params.n_mels, params.n_frames), device=wave.device, dtype=torch.float32)
    _timed("mel", 4.0 * (B * L + out.numel()),
           lambda: check(_lib.load().pa_mel_frontend_fwd(_p(wave, torch.float32), B, L, _p(window, torch.float32), _p(bin_mel, torch.float32),
                                                         _p(twiddle, torch.float32), _p(out), C.byref(params), _stream()),
                         "pa_mel_frontend_fwd"))
    return out


# ---- staging ---------------------------------------------------------------------------------
def convert(x_f32, dtype):
    if dtype == PA_F32:
        return x_f32
    out = torch.empty(x_f32.shape, device=x_f32.device, dtype=TORCH_DTYPE[dtype])
    check(_lib.load().pa_convert_f32(_p(x_f32, torch.float32), _p(out), x_f32.numel(), dtype, _stream()), "pa_convert_f32")
    return out


def convert_f32(x_f32, out_lp):
    """out_lp[i] = (bf16 / f32) x_f32[i] into an existing tensor"""
    check(_lib.load().pa_convert_f32(_p(x_f32, torch.float32), _p(out_lp), x_f32.numel(), PA_DTYPE[out_lp.dtype], _stream()),
          "pa_convert_f32")


def convert_to_f32(x_lp, out_f32):
    """out_f32[i] = (float) x_lp[i]"""
    check(_lib.load().pa_convert_to_f32(_p(x_lp), PA_DTYPE[x_lp.dtype], _p(out_f32, torch.float32), x_lp.numel(), _stream()),
          "pa_convert_to_f32")


def transpose(x, out_dtype, ldo=None, out=None):
    """x [R][C] (contiguous rows, any supported dtype) -> [C][ldo] (ldo >= R, zero padded)."""
    R, Cc = x.shape
    ldo = R if ldo is None else ldo
    if out is None:
        out = torch.empty((Cc, ldo), device=x.device, dtype=TORCH_DTYPE[out_dtype])
    check(_lib.load().pa_transpose(_p(x, None, True), PA_DTYPE[x.dtype], R, Cc, x.stride(0), _p(out), out_dtype, ldo,
                                   _stream()), "pa_transpose")
    return out


def make_stage_table(entries, device):
    """entries: [(src f32 [R][C] contiguous, dst or None, dst_t or None)] -> (device table, n, total_tiles) for
    stage_weights; the tensors must stay alive (and in place) for as long as the table is used."""
    import numpy as np
    from ._lib import StageDesc
    arr = (StageDesc * len(entries))()
    tiles = 0
    for d, (src, dst, dst_t) in zip(arr, entries):
        R, Cc = src.shape
        assert src.is_contiguous() and src.dtype == torch.float32
        d.src, d.dst, d.dst_t = _p(src), _p(dst), _p(dst_t)
        d.rows, d.cols, d.tile_begin = R, Cc, tiles
        tiles += ((R + 63) // 64) * ((Cc + 63) // 64)
    host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
    return host.to(device), len(entries), tiles


def make_adamw_stage_table(entries, device):
    """entries: [(offset, rows, cols, dst or None, dst_t or None)] in flat-buffer order -> (device table, n, total work items) for
    adamw_stage (pa_adamw_stage_desc; a parameter without copies is one run of 4096 elements per work item)."""
    import numpy as np
    from ._lib import AdamwStageDesc
    arr = (AdamwStageDesc * len(entries))()
    items = 0
    for d, (off, rows, cols, dst, dst_t) in zip(arr, entries):
        d.offset, d.rows, d.cols, d.dst, d.dst_t, d.tile_begin = off, rows, cols, _p(dst), _p(dst_t), items
        items += ((rows + 63) // 64) * ((cols + 63) // 64) if (dst is not None or dst_t is not None) else (rows * cols + 4095) // 4096
    host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
    return host.to(device), len(entries), items


def adamw_stage(p, g, m, v, table, n, items, dtype, lr, beta1, beta2, eps, weight_decay, step, hyper_dev=None):
    """AdamW over the parameters the table lists (offsets relative to p / g / m / v) + their GEMM-ready copies, one launch."""
    check(_lib.load().pa_adamw_stage(_p(p, torch.float32), _p(g, torch.float32), _p(m, torch.float32), _p(v, torch.float32), _p(table), n, items,
                                     dtype, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                     _p(hyper_dev), _stream()), "pa_adamw_stage")


def stage_weights(table, n, tiles, dtype):
    check(_lib.load().pa_stage_weights(_p(table), n, tiles, dtype, _stream()), "pa_stage_weights")


# ---- layer norm ------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, dtype, save_stats=True):
    M, D = x.shape
    y = torch.empty((M, D), device=x.device, dtype=TORCH_DTYPE[dtype])
    mean = torch.empty(M, device=x.device, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(M, device=x.device, dtype=torch.float32) if save_stats else None
    check(_lib.load().pa_layernorm_fwd(_p(x, torch.float32), _p(gamma, torch.float32), _p(beta, torch.float32), _p(y), dtype, _p(mean), _p(rstd), M, D, eps,
                                       _stream()), "pa_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, want_lp, accumulate=False, dcolsum=None, defer=None):
    """Returns (dx f32, dx_lp or None).  dgamma/dbeta (and dcolsum = column sums of dx, if given) are written in place.
    defer: a list -- the parameter gradients stay as partial rows and (partial, rows, pitch, n, out) jobs are appended to it for
    the block's ONE finishing launch (wgrad_tn_batched(..., row_jobs=defer)) instead of a reduction launch per LayerNorm."""
    M, D = x.shape
    dtype = PA_DTYPE[dy.dtype]
    lib = _lib.load()
    dx = torch.empty((M, D), device=x.device, dtype=torch.float32)
    dx_lp = torch.empty((M, D), device=x.device, dtype=dy.dtype) if (want_lp and dtype != PA_F32) else None
    ws = torch.empty(lib.pa_layernorm_bwd_ws_floats(M, D), device=x.device, dtype=torch.float32)
    if defer is not None and not accumulate:
        check(lib.pa_layernorm_bwd_partial(_p(dy), dtype, _p(x, torch.float32), _p(gamma, torch.float32), _p(mean, torch.float32),
                                           _p(rstd, torch.float32), _p(dres, torch.float32), _p(dx), _p(dx_lp), _p(ws), M, D, _stream()),
              "pa_layernorm_bwd_partial")
        rows = lib.pa_layernorm_bwd_rows(M)
        for j, out in enumerate((dgamma, dbeta, dcolsum)):
            if out is not None:
                defer.append((ws[j * D:], rows, 3 * D, D, out))
        return dx, (dx if dtype == PA_F32 else dx_lp)
    check(lib.pa_layernorm_bwd(_p(dy), dtype, _p(x, torch.float32), _p(gamma, torch.float32), _p(mean, torch.float32), _p(rstd, torch.float32), _p(dres, torch.float32), _p(dx), _p(dx_lp),
                               _p(dgamma), _p(dbeta), _p(dcolsum), int(accumulate), _p(ws), M, D, _stream()),
          "pa_layernorm_bwd")
    return dx, (dx if dtype == PA_F32 else dx_lp)


# ---- GEMM ------------------------------------------------------------------------------------
# A/B knobs: the tile variant of every plain-store / residual NT GEMM of the step (PASST_AMD_TUNE_STORE / PASST_AMD_TUNE_RESID =
# a pa_gemm_args.tune value, e.g. 13 = the 192 x 128 two-workgroups-per-CU tile with epilogue v2)
_TUNE_BY_EPI = {EPI_STORE: int(os.environ.get("PASST_AMD_TUNE_STORE", "0")), EPI_RESID: int(os.environ.get("PASST_AMD_TUNE_RESID", "0"))}


def gemm_nt(A, B, dtype, epilogue=EPI_STORE, bias=None, resid=None, aux=None, out_f32=None, out_lp=None,
            out_lp2=None, row_mod=0, out_batch_rows=0, out_row_off=0, split_k=1, M=None, N=None, K=None,
            colsum_out=None, colsum_ws=None, colsum_accumulate=False, flags=0, colscale_n=0, colscale=1.0, tune=None):
    """C[M][N] = A[M][K] B[N][K]^T with the fused epilogues of include/passt_amd.h.  EPI_DGELU can also return the
    column sums of its output (colsum_out [N] f32; colsum_ws from gemm_colsum_ws)."""
    a = GemmArgs()
    a.dtype, a.epilogue = dtype, epilogue
    a.M = A.shape[0] if M is None else M
    a.N = B.shape[0] if N is None else N
    a.K = A.shape[1] if K is None else K
    a.lda, a.ldb = A.stride(0), B.stride(0)
    a.A, a.B = _p(A, dtype, True), _p(B, dtype, True)
    a.bias = _p(bias, torch.float32)
    a.resid = _p(resid, torch.float32, True)
    a.ldr = resid.stride(0) if resid is not None else 0
    a.row_mod, a.out_batch_rows, a.out_row_off = row_mod, out_batch_rows, out_row_off
    a.aux = _p(aux, dtype, True)
    a.ldaux = aux.stride(0) if aux is not None else 0
    a.out_f32 = _p(out_f32, torch.float32, True)
    a.ldo32 = (out_f32.stride(-2) if out_f32 is not None else 0)
    a.out_lp = _p(out_lp, dtype, True)
    a.ldolp = out_lp.stride(0) if out_lp is not None else 0
    a.out_lp2 = _p(out_lp2, dtype, True)
    a.ldolp2 = out_lp2.stride(0) if out_lp2 is not None else 0
    a.split_k = split_k
    if tune is None:
        tune = _TUNE_BY_EPI.get(epilogue) or GEMM_TUNE
    a.tune = tune
    a.reserved = GEMM_RESERVED | _call.gemm_flags | flags
    a.colscale_n, a.colscale = colscale_n, colscale
    a.colsum_out, a.colsum_ws, a.colsum_accumulate = _p(colsum_out, torch.float32), _p(colsum_ws, torch.float32), int(colsum_accumulate)
    # problems that leave most CUs idle (few [M][768] tiles, long K) go through the split-K entry with a workspace
    ws_n = 0
    if epilogue in (EPI_STORE, EPI_RESID) and dtype == PA_BF16 and row_mod == 0 and split_k == 1 and not tune:
        ws_n = _splitk_ws_floats(a.M, a.N, a.K, epilogue, dtype)

    def launch():
        if ws_n:
            ws = _splitk_ws(A.device, ws_n)
            check(_lib.load().pa_gemm_nt_splitk(C.byref(a), ws.data_ptr(), ws_n, _stream()), "pa_gemm_nt_splitk")
        else:
            check(_lib.load().pa_gemm_nt(C.byref(a), _stream()), "pa_gemm_nt")

    if GEMM_PROFILE is None:
        launch()
        return
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    launch()
    ev1.record()
    kind = _EPI_NAME[epilogue] if not PROFILE_BY_SHAPE else f"{_EPI_NAME[epilogue]}_M{a.M}_N{a.N}_K{a.K}"
    GEMM_PROFILE.setdefault(kind, []).append((ev0, ev1, 2.0 * a.M * a.N * a.K))


@functools.lru_cache(maxsize=None)
def _splitk_ws_floats(M, N, K, epilogue, dtype):
    return int(_lib.load().pa_gemm_nt_splitk_ws_floats(M, N, K, epilogue, dtype))


_SPLITK_WS = {}


def _splitk_ws(device, n):
    """f32 workspace of pa_gemm_nt_splitk, one per (device, stream, host thread): the partial tiles live only between the two
    launches of one call and calls of one thread on one stream are ordered; two host threads on the same stream (trainer +
    autograd thread) could interleave their launch pairs, so they do not share a workspace."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _SPLITK_WS[key] = torch.empty(n, device=device, dtype=torch.float32)
    return ws


def gemm_colsum_ws(M, N, device, ws=None):
    """f32 workspace for gemm_nt(..., colsum_out=...); reuses `ws` when it is large enough."""
    n = _lib.load().pa_gemm_colsum_ws_floats(M, N)
    if ws is None or ws.numel() < n:
        ws = torch.empty(n, device=device, dtype=torch.float32)
    return ws


def linear(x_lp, W_lp, bias, dtype, colscale_n=0, colscale=1.0):
    """out_lp[M][N] = x W^T + b; the first colscale_n output columns times colscale (the qkv Linear hands attention
    q * scale * log2(e): ATTN_Q_PRESCALED)"""
    out = torch.empty((x_lp.shape[0], W_lp.shape[0]), device=x_lp.device, dtype=TORCH_DTYPE[dtype])
    gemm_nt(x_lp, W_lp, dtype, EPI_STORE, bias=bias, out_lp=out, colscale_n=colscale_n, colscale=colscale)
    return out


class BlockedPre:
    """The pre-activation of an MLP in the library's blocked layout (PA_GEMM_BLOCKED_PRE): opaque storage written by the
    fc1 + GELU epilogue and read by the GELU' epilogue of the matching input-gradient GEMM, nothing else."""
    __slots__ = ("buf", "shape")

    def __init__(self, buf, shape):
        self.buf, self.shape = buf, shape


@functools.lru_cache(maxsize=None)
def _lib_blocked_pre_ok(M, N, K):
    return bool(_lib.load().pa_gemm_blocked_pre_ok(M, N, K))


def blocked_pre_ok(M, N, K):
    # only the library query is cached: GEMM_TUNE and the environment switch are read on every call (a
    # result cached under a forced variant disabled -- or wrongly enabled -- the blocked path for the rest of the process)
    return not GEMM_TUNE and not os.environ.get("PASST_AMD_NO_BLOCKED_PRE") and _lib_blocked_pre_ok(M, N, K)


# A/B knobs (tools/bench_epi13.py, profiles/r06_gemm_variants_epi13.txt): the tile variant of the two MLP-epilogue GEMMs in the
# step, blocked pre-activation kept (every variant listed in csrc/gemm.hip launch_gemm's blocked switch has it)
TUNE_GELU = int(os.environ.get("PASST_AMD_TUNE_GELU", "0")) or None
TUNE_DGELU = int(os.environ.get("PASST_AMD_TUNE_DGELU", "0")) or None


def linear_gelu(x_lp, W_lp, bias, dtype):
    """(pre, act): act = gelu(x W^T + b) row-major; pre = x W^T + b, row-major or -- where the library has the blocked
    form for this shape -- a BlockedPre that only dgelu_gemm() can consume."""
    M, N = x_lp.shape[0], W_lp.shape[0]
    act = torch.empty((M, N), device=x_lp.device, dtype=TORCH_DTYPE[dtype])
    if dtype == PA_BF16 and blocked_pre_ok(M, N, x_lp.shape[1]):
        buf = torch.empty(_lib.load().pa_gemm_blocked_pre_elems(M, N), device=x_lp.device, dtype=TORCH_DTYPE[dtype])
        # blocks of rows no wave tile of this GEMM writes: the GELU' epilogue of the last row tile adds and subtracts them
        # for the fused bias sums, so they must be finite whatever tile height the two GEMMs resolve to
        buf[(M + 31) // 32 * 32 * N:].zero_()
        gemm_nt(x_lp, W_lp, dtype, EPI_GELU, bias=bias, out_lp=buf.view(-1, N), out_lp2=act, flags=GEMM_BLOCKED_PRE, tune=TUNE_GELU)
        return BlockedPre(buf, (M, N)), act
    pre = torch.empty((M, N), device=x_lp.device, dtype=TORCH_DTYPE[dtype])
    gemm_nt(x_lp, W_lp, dtype, EPI_GELU, bias=bias, out_lp=pre, out_lp2=act)
    return pre, act


def dgelu_gemm(dy_lp, Wt_lp, pre, dtype, colsum_out=None, colsum_ws=None, defer=None):
    """d_pre[M][N] = (dy Wt^T) * gelu'(pre): the input gradient of fc2 times the GELU derivative; pre as returned by
    linear_gelu (row-major tensor or BlockedPre).  defer (a list, with colsum_out): the column sums stay as one partial row per
    wave tile in colsum_ws and a (partial, rows, pitch, n, out) job is appended for the block's finishing launch."""
    blocked = isinstance(pre, BlockedPre)
    M, N = pre.shape
    d_pre = torch.empty((M, N), device=dy_lp.device, dtype=TORCH_DTYPE[dtype])
    deferred = defer is not None and colsum_out is not None
    gemm_nt(dy_lp, Wt_lp, dtype, EPI_DGELU, aux=pre.buf.view(-1, N) if blocked else pre, out_lp=d_pre,
            colsum_out=colsum_out, colsum_ws=colsum_ws, flags=(GEMM_BLOCKED_PRE if blocked else 0) | (_lib.GEMM_COLSUM_DEFER if deferred else 0),
            tune=TUNE_DGELU if blocked else None)
    if deferred:
        defer.append((colsum_ws, _lib.load().pa_gemm_last_colsum_rows(), N, N, colsum_out))
    return d_pre


def linear_resid(x_lp, W_lp, bias, resid_f32, dtype, out=None):
    """out_f32[M][N] = resid + x W^T + b"""
    if out is None:
        out = torch.empty_like(resid_f32)
    gemm_nt(x_lp, W_lp, dtype, EPI_RESID, bias=bias, resid=resid_f32, out_f32=out)
    return out


def pick_split_k(n_out_tiles, ksteps):
    """Enough workgroups to fill 256 CUs x 2, at least 8 K-steps per slice."""
    s = max(1, min(64, (768 + n_out_tiles - 1) // n_out_tiles))
    return max(1, min(s, ksteps // 8 if ksteps >= 8 else 1))


def wgrad(dY_t, X_t, out_f32, dtype, accumulate=False, partial_ws=None):
    """out[N][K] (+)= dY_t[N][Mp] . X_t[K][Mp]^T  -- the weight gradient dY^T X from transposed,
    zero-padded operands; deterministic split-K (f32 partial slabs + ordered reduction)."""
    N, Mp = dY_t.shape
    K = X_t.shape[0]
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    ksteps = Mp // kpad(dtype)
    S = pick_split_k(tiles, ksteps)
    need = S * N * K
    if partial_ws is None or partial_ws.numel() < need:
        partial_ws = torch.empty(need, device=dY_t.device, dtype=torch.float32)
    part = partial_ws[:need].view(S, N, K)
    gemm_nt(dY_t, X_t, dtype, EPI_PARTIAL, out_f32=part, split_k=S)
    check(_lib.load().pa_reduce_partials(_p(part), S, N * K, _p(out_f32), int(accumulate), _stream()),
          "pa_reduce_partials")
    return partial_ws


def tn_step_rows():
    """PA_TN_STEP_ROWS of the loaded library: tokens per pipeline stage of the bf16 role-split TN kernel."""
    return _lib.load().pa_gemm_tn_step_rows()



def pick_split_k_slots(tiles, steps, slots=256):
    """split count for the one-workgroup-per-CU kernels: fill whole rounds of `slots` workgroups
    (tile quantisation), at least 8 reduction steps per slice, prefer fewer slices on ties."""
    best, best_score = 1, -1.0
    for S in range(1, max(1, min(64, steps // 8)) + 1):
        blocks = tiles * S
        eff = blocks / (((blocks + slots - 1) // slots) * slots)
        score = eff - 0.003 * S
        if score > best_score:
            best, best_score = S, score
    return best


_SPLITS_CACHE = {}
_FORCE_SLICES = int(os.environ.get("PASST_AMD_WGRAD_SLICES", "0"))


def pick_batched_splits(probs, slots=256):
    """Memoised front of _pick_batched_splits (the search walks ~470 slice lengths in Python: ~1 ms per call, and the step
    asks the same question once per transformer block)."""
    key = (tuple(probs), slots)
    hit = _SPLITS_CACHE.get(key)
    if hit is None:
        hit = _SPLITS_CACHE[key] = _pick_batched_splits(list(probs), slots)
        if _FORCE_SLICES:       # A/B knob (profiles/r06_wgrad_slices.txt): every problem cut into this many token slices
            hit = _SPLITS_CACHE[key] = [max(1, min(_FORCE_SLICES, st)) for _, st in probs]
    return hit


def _pick_batched_splits(probs, slots=256):
    """probs: [(tiles, k_steps)] of the problems sharing one launch.  Every work item is one 256x256 tile x one K slice of
    L steps of 48 tokens (S_p = ceil(steps_p / L)).  Cost model (us, MI355X measurements): rounds x (L + 4) x 1.2 for the K loops and
    per-item epilogues, with rounds = ceil(#items / slots), plus 0.05 per item for the f32 partial slab it writes and the
    reduction reads back (256 KiB each way; about half of it hides under the K loops -- calibrated on the passt_s block:
    7 slices beat 2 and 4, run 82)."""
    best, best_cost = None, None
    for L in range(max(p[1] for p in probs), 3, -1):
        S = [max(1, min(64, -(-st // L))) for _, st in probs]
        items = sum(t * s_ for (t, _), s_ in zip(probs, S))
        per = max(-(-st // s_) for (_, st), s_ in zip(probs, S))
        cost = -(-items // slots) * (per + 4) * 1.2 + items * 0.05        # 48-token steps: ~1.2 us each, ~4.8 us fixed per item
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = S, cost
    return best if best is not None else [1] * len(probs)


def wgrad_tn(dY, X, out_f32, dtype, accumulate=False, partial_ws=None, db=None):
    """out[N][K] (+)= dY[M][N]^T X[M][K], operands read in place (pa_gemm_tn), deterministic split-K.  db [N] f32 (bf16
    role-split kernel only, see wgrad_tn_fuses_bias): (+)= column sums of dY out of the same launch."""
    Mtok, N = dY.shape
    K = X.shape[1]
    if dtype == PA_BF16 and GEMM_TUNE != 1:      # role-split 256x256 kernel, one workgroup per CU
        tiles = ((N + 255) // 256) * ((K + 255) // 256)
        S = pick_split_k_slots(tiles, (Mtok + tn_step_rows() - 1) // tn_step_rows())
    else:
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        mrows = 64 if dtype == PA_BF16 else 32
        S = pick_split_k(tiles, (Mtok + mrows - 1) // mrows)
    assert db is None or wgrad_tn_fuses_bias(dtype)
    need = S * N * K + (S * N if db is not None else 0)
    if partial_ws is None or partial_ws.numel() < need:
        partial_ws = torch.empty(need, device=dY.device, dtype=torch.float32)
    part = partial_ws[:S * N * K].view(S, N, K)
    a = GemmArgs()
    a.dtype, a.epilogue = dtype, EPI_PARTIAL
    a.M, a.N, a.K = N, K, Mtok
    a.lda, a.ldb = dY.stride(0), X.stride(0)
    a.A, a.B = _p(dY, dtype, True), _p(X, dtype, True)
    a.out_f32, a.ldo32 = _p(part), K
    a.split_k = S
    a.tune = GEMM_TUNE
    bpart = partial_ws[S * N * K:need] if db is not None else None
    a.colsum_ws = _p(bpart)
    lib = _lib.load()
    # the bracket covers the split-K reductions too: they are part of what a weight gradient costs
    ev0 = ev1 = None
    if GEMM_PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib.pa_gemm_tn(C.byref(a), _stream()), "pa_gemm_tn")
    check(lib.pa_reduce_partials(_p(part), S, N * K, _p(out_f32), int(accumulate), _stream()), "pa_reduce_partials")
    if db is not None:
        check(lib.pa_reduce_partials(_p(bpart), S, N, _p(db, torch.float32), int(accumulate), _stream()), "pa_reduce_partials")
    if ev0 is not None:
        ev1.record()
        GEMM_PROFILE.setdefault("wgrad_tn", []).append((ev0, ev1, 2.0 * N * K * Mtok))
    return partial_ws


def wgrad_tn_fuses_bias(dtype):
    """True when wgrad_tn / wgrad_tn_batched can return colsum(dY) from the weight-gradient launch itself."""
    return dtype == PA_BF16 and GEMM_TUNE != 1


def wgrad_tn_batched(problems, dtype, partial_ws=None, row_jobs=None):
    """problems: up to 4 of (dY [M][N], X [M][K], out [N][K] f32, accumulate[, db [N] f32 or None]): all weight gradients of a
    block in ONE pa_gemm_tn_batched launch + ONE batched finishing reduction.  A problem with db also gets its bias gradient
    (column sums of dY) out of the same launch.  row_jobs: (partial, rows, pitch, n, out) reductions of many short partial rows
    (deferred LayerNorm / GELU' parameter gradients of the block, see layernorm_bwd / dgelu_gemm) that ride in the same finishing
    launch.  Returns the (possibly grown) partial workspace."""
    from ._lib import ReduceDesc
    assert dtype == PA_BF16 and 1 <= len(problems) <= 4
    problems = [tuple(p) + (None,) * (5 - len(p)) for p in problems]
    metas = []
    for dY, X, out, acc, db in problems:
        Mtok, N = dY.shape
        metas.append((Mtok, N, X.shape[1]))
    splits = pick_batched_splits([(((N + 255) // 256) * ((K + 255) // 256), (Mtok + tn_step_rows() - 1) // tn_step_rows())
                                  for Mtok, N, K in metas])
    need = sum(S * m[1] * (m[2] + (1 if p[4] is not None else 0)) for S, m, p in zip(splits, metas, problems))
    if partial_ws is None or partial_ws.numel() < need:
        partial_ws = torch.empty(need, device=problems[0][0].device, dtype=torch.float32)
    row_jobs = row_jobs or []
    nred = len(problems) + sum(1 for p in problems if p[4] is not None) + len(row_jobs)
    args = (GemmArgs * len(problems))()
    red = (ReduceDesc * nred)()
    off, flops, k = 0, 0.0, len(problems)
    for a, r, (dY, X, out, acc, db), (Mtok, N, K), S in zip(args, red, problems, metas, splits):
        part = partial_ws[off:off + S * N * K]
        off += S * N * K
        a.dtype, a.epilogue = dtype, EPI_PARTIAL
        a.M, a.N, a.K = N, K, Mtok
        a.lda, a.ldb = dY.stride(0), X.stride(0)
        a.A, a.B = _p(dY, dtype, True), _p(X, dtype, True)
        a.out_f32, a.ldo32 = _p(part), K
        a.split_k, a.tune = S, TN_BATCH_ORDER
        r.partial, r.out, r.n, r.splits, r.accumulate = _p(part), _p(out, torch.float32), N * K, S, int(acc)
        if db is not None:              # bias partials [S][N] behind the weight partials; one more slab set to reduce
            bpart = partial_ws[off:off + S * N]
            off += S * N
            a.colsum_ws = _p(bpart)
            rb = red[k]
            k += 1
            rb.partial, rb.out, rb.n, rb.splits, rb.accumulate = _p(bpart), _p(db, torch.float32), N, S, int(acc)
        flops += 2.0 * N * K * Mtok
    for part_rows, rows, pitch, n, out in row_jobs:
        rb = red[k]
        k += 1
        rb.partial, rb.out, rb.n, rb.splits, rb.accumulate = _p(part_rows, torch.float32, True), _p(out, torch.float32), n, rows, 0
        rb.pitch, rb.mode = pitch, _lib.REDUCE_ROWS
    lib = _lib.load()
    ev0 = ev1 = None
    if GEMM_PROFILE is not None:        # the bracket covers the batched split-K reduction too
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib.pa_gemm_tn_batched(args, len(problems), _stream()), "pa_gemm_tn_batched")
    # a normal block fills the launch exactly (4 weight gradients + qkv.bias + 2 x 3 LayerNorm rows + fc1.bias = 12 =
    # PA_REDUCE_BATCH_MAX); anything beyond goes out as a second launch instead of failing in the middle of a backward
    for lo in range(0, nred, REDUCE_BATCH_MAX):
        n = min(REDUCE_BATCH_MAX, nred - lo)
        check(lib.pa_reduce_partials_batched(C.cast(C.byref(red, lo * C.sizeof(ReduceDesc)), C.POINTER(ReduceDesc)), n, _stream()),
              "pa_reduce_partials_batched")
    if ev0 is not None:
        ev1.record()
        GEMM_PROFILE.setdefault("wgrad_tn", []).append((ev0, ev1, flops))
    return partial_ws


REDUCE_BATCH_MAX = 12         # include/passt_amd.h PA_REDUCE_BATCH_MAX


def colsum(x, out_f32, accumulate=False):
    """out[C] (+)= column sums of the tall matrix x[R][C] (bf16 or f32)."""
    R, Cc = x.shape
    lib = _lib.load()
    ws = torch.empty(lib.pa_colsum_ws_floats(R, Cc), device=x.device, dtype=torch.float32)
    check(lib.pa_colsum(_p(x, None, True), PA_DTYPE[x.dtype], R, Cc, x.stride(0), _p(out_f32, torch.float32), int(accumulate), _p(ws),
                        _stream()), "pa_colsum")


def rowsum(x, out_f32, ncols=None, accumulate=False):
    R, Cc = x.shape
    check(_lib.load().pa_rowsum(_p(x, None, True), PA_DTYPE[x.dtype], R, Cc if ncols is None else ncols, x.stride(0),
                                _p(out_f32), int(accumulate), _stream()), "pa_rowsum")


def colsum_f32(x, out_f32, accumulate=False):
    R, Cc = x.shape
    check(_lib.load().pa_colsum_f32(_p(x, torch.float32, True), R, Cc, x.stride(0), _p(out_f32, torch.float32), int(accumulate), _stream()),
          "pa_colsum_f32")


# ---- attention -------------------------------------------------------------------------------
ATTN_Q_PRESCALED = 1          # include/passt_amd.h PA_ATTN_Q_PRESCALED
ATTN_BWD_TWO_PASS = 2         # PA_ATTN_BWD_TWO_PASS: force the dQ + dK/dV kernel pair (A/B, tests); PASST_AMD_ATTN_BWD=two_pass sets it everywhere
ATTN_BWD_SINGLE_PASS = 4      # PA_ATTN_BWD_SINGLE_PASS: force the single-pass kernel wherever it applies; PASST_AMD_ATTN_BWD=single_pass
ATTN_BWD_SINGLE_PASS_W16 = 8  # PA_ATTN_BWD_SINGLE_PASS_W16 (ABI 6): the single pass as sixteen waves of 32 keys; PASST_AMD_ATTN_BWD=single_pass_w16
_ATTN_BWD_FORCE = {"two_pass": ATTN_BWD_TWO_PASS, "single_pass": ATTN_BWD_SINGLE_PASS,
                   "single_pass_w16": ATTN_BWD_SINGLE_PASS_W16}.get(os.environ.get("PASST_AMD_ATTN_BWD", ""), 0)
LOG2E = 1.4426950408889634


def attention_fwd(qkv, B, H, N, scale, nq=None, flags=0):
    """o[(b*nq+q)][H*64], lse[(b*H+h)*nq+q] for the first nq queries of every sequence (nq=None: all N).
    flags=ATTN_Q_PRESCALED: the q third of qkv holds q * scale * log2(e)."""
    dtype = PA_DTYPE[qkv.dtype]
    nq = N if nq is None else nq
    D = H * 64
    o = torch.empty((B * nq, D), device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty((B * H * nq,), device=qkv.device, dtype=torch.float32)
    _timed("attn_fwd", 4.0 * nq * N * 64 * B * H,
           lambda: check(_lib.load().pa_attention_fwd(_p(qkv, None, True), qkv.stride(0), _p(o), o.stride(0), _p(lse), B, H, N, nq,
                                                      scale, dtype, flags, _stream()), "pa_attention_fwd"))
    return o, lse


def attention_bwd(qkv, o, d_o, lse, B, H, N, scale, nq=None, flags=0):
    """dqkv [B*N][3D]; with nq < N (o, d_o, lse compact) the Q third is zero outside the first nq rows."""
    dtype = PA_DTYPE[qkv.dtype]
    nq = N if nq is None else nq
    dqkv = torch.empty_like(qkv)
    lib = _lib.load()
    delta = torch.empty(lib.pa_attention_bwd_ws_floats(B, H, nq), device=lse.device, dtype=torch.float32)
    if nq < N:
        es = qkv.element_size()
        check(lib.pa_zero2d(_p(dqkv), dqkv.stride(0) * es, H * 64 * es, B * N, _stream()), "pa_zero2d")
    _timed("attn_bwd", 10.0 * nq * N * 64 * B * H,      # five N x N x 64 products: S, dP, dV, dK, dQ
           lambda: check(lib.pa_attention_bwd(_p(qkv, None, True), qkv.stride(0), _p(o, qkv.dtype, True), _p(d_o, qkv.dtype, True),
                                              o.stride(0), _p(lse, torch.float32), _p(delta), _p(dqkv), dqkv.stride(0), B, H, N, nq,
                                              scale, dtype, flags | _ATTN_BWD_FORCE, _stream()), "pa_attention_bwd"))
    return dqkv


def gather_rows(x, idx_i32):
    """out[i] = x[idx[i]] for a 2-D (or 1-D) contiguous tensor."""
    n = idx_i32.numel()
    out = torch.empty((n,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    row_bytes = x.stride(0) * x.element_size()
    check(_lib.load().pa_gather_rows(_p(x), _p(idx_i32, torch.int32), n, row_bytes, _p(out), _stream()), "pa_gather_rows")
    return out


def scatter_rows_into_zeros(x_rows, idx_i32, n_rows):
    """zeros(n_rows, D) with out[idx[i]] = x_rows[i]"""
    out = torch.empty((n_rows,) + tuple(x_rows.shape[1:]), device=x_rows.device, dtype=x_rows.dtype)
    row_bytes = x_rows.stride(0) * x_rows.element_size()
    lib = _lib.load()
    check(lib.pa_zero2d(_p(out), row_bytes, row_bytes, n_rows, _stream()), "pa_zero2d")
    check(lib.pa_scatter_rows(_p(x_rows), _p(idx_i32, torch.int32), idx_i32.numel(), row_bytes, _p(out), _stream()), "pa_scatter_rows")
    return out


# ---- patch embedding -------------------------------------------------------------------------
def patch_gather(x, patch_f, patch_t, P, fstride, tstride, dtype):
    B, _, F, T = x.shape
    Np = patch_f.numel()
    cols = torch.empty((B * Np, P * P), device=x.device, dtype=TORCH_DTYPE[dtype])
    check(_lib.load().pa_patch_gather(_p(x, torch.float32), B, F, T, _p(patch_f, torch.int32), _p(patch_t, torch.int32), Np, P, fstride, tstride, _p(cols),
                                      dtype, _stream()), "pa_patch_gather")
    return cols


def patch_pos_table(bias, time_pos, freq_pos, patch_f, patch_t, toff, cls, dist, npe, tok):
    B, Ntok, D = tok.shape
    Np = patch_f.numel()
    Tpe, Fpe = time_pos.shape[-1], freq_pos.shape[-2]
    table = torch.empty((Np, D), device=tok.device, dtype=torch.float32)
    check(_lib.load().pa_patch_pos_table(_p(bias), _p(time_pos), Tpe, _p(freq_pos), Fpe, _p(patch_f), _p(patch_t),
                                         Np, toff, D, _p(table), _p(cls), _p(dist), _p(npe), _p(tok), B, Ntok,
                                         _stream()), "pa_patch_pos_table")
    return table


def patch_bwd(dtok, patch_f, patch_t, toff, Tpe, Fpe, d_cls, d_dist, d_npe, d_bias, d_tpos, d_fpos, dtype,
              accumulate=False):
    B, Ntok, D = dtok.shape
    Np = patch_f.numel()
    gsum = torch.empty((Ntok, D), device=dtok.device, dtype=torch.float32)
    dpatch = torch.empty((B * Np, D), device=dtok.device, dtype=TORCH_DTYPE[dtype])
    check(_lib.load().pa_patch_bwd(_p(dtok), B, Ntok, D, _p(patch_f), _p(patch_t), Np, toff, Tpe, Fpe, _p(gsum),
                                   _p(d_cls), _p(d_dist), _p(d_npe), _p(d_bias), _p(d_tpos), _p(d_fpos),
                                   int(accumulate), _p(dpatch), dtype, _stream()), "pa_patch_bwd")
    return dpatch


# ---- head / loss -----------------------------------------------------------------------------
def head_pre_fwd(x, norm_g, norm_b, eps_norm, hg, hb, eps_head):
    B, Ntok, D = x.shape
    feat = torch.empty((B, D), device=x.device, dtype=torch.float32)
    hn = torch.empty((B, D), device=x.device, dtype=torch.float32)
    stats = torch.empty((B, 6), device=x.device, dtype=torch.float32)
    check(_lib.load().pa_head_pre_fwd(_p(x), B, Ntok, D, _p(norm_g), _p(norm_b), eps_norm, _p(hg), _p(hb), eps_head,
                                      _p(feat), _p(hn), _p(stats), _stream()), "pa_head_pre_fwd")
    return feat, hn, stats


def head_pre_bwd(dhn, dfeat, x, feat, norm_g, hg, stats):
    B, Ntok, D = x.shape
    dx = torch.empty((B, Ntok, D), device=x.device, dtype=torch.float32)
    part = torch.empty((B, 4 * D), device=x.device, dtype=torch.float32)
    check(_lib.load().pa_head_pre_bwd(_p(dhn), _p(dfeat), _p(x), _p(feat), B, Ntok, D, _p(norm_g), _p(hg), _p(stats),
                                      _p(dx), _p(part), _stream()), "pa_head_pre_bwd")
    return dx, part


def linear_f32_fwd(x, W, b):
    B, D = x.shape
    Cc = W.shape[0]
    y = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
    check(_lib.load().pa_linear_f32_fwd(_p(x, torch.float32), _p(W, torch.float32), _p(b, torch.float32), _p(y), B, Cc, D, _stream()), "pa_linear_f32_fwd")
    return y


def linear_f32_bwd(dy, x, W, dW, db, accumulate=False):
    B, Cc = dy.shape
    D = x.shape[1]
    dx = torch.empty((B, D), device=x.device, dtype=torch.float32)
    check(_lib.load().pa_linear_f32_bwd(_p(dy, torch.float32), _p(x, torch.float32), _p(W, torch.float32), _p(dx), _p(dW, torch.float32), _p(db, torch.float32), int(accumulate), B, Cc, D,
                                        _stream()), "pa_linear_f32_bwd")
    return dx


def bce_fwd_bwd(logits, target, grad_scale=1.0):
    B, Cc = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    dlogits = torch.empty_like(logits)
    ws = torch.empty(1 + (B * Cc + 255) // 256, device=logits.device, dtype=torch.float32)
    check(_lib.load().pa_bce_fwd_bwd(_p(logits, torch.float32), _p(target, torch.float32), B, Cc, grad_scale, _p(loss), _p(dlogits), _p(ws),
                                     _stream()), "pa_bce_fwd_bwd")
    return loss, dlogits


def ce_mixup_fwd_bwd(logits, target_i32, target2_i32=None, lam=None, grad_scale=1.0):
    """ESC-50 loss (ex_esc50.py:159-165); targets are int32 class indices."""
    B, Cc = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    dlogits = torch.empty_like(logits)
    ws = torch.empty(max(B, 1), device=logits.device, dtype=torch.float32)
    check(_lib.load().pa_ce_mixup_fwd_bwd(_p(logits, torch.float32), _p(target_i32, torch.int32), _p(target2_i32, torch.int32), _p(lam, torch.float32), B, Cc, grad_scale,
                                          _p(loss), _p(dlogits), _p(ws), _stream()), "pa_ce_mixup_fwd_bwd")
    return loss, dlogits


# ---- caller glue -----------------------------------------------------------------------------
def mixup(x, perm_i32, lam):
    B = x.shape[0]
    out = torch.empty_like(x)
    check(_lib.load().pa_mixup(_p(x, torch.float32), _p(perm_i32, torch.int32), _p(lam, torch.float32), _p(out), B, x[0].numel(), _stream()), "pa_mixup")
    return out


def wave_augment(x, L, lengths=None, amp=None, shift=None, partner=None, lam=None):
    """x [B][ldx] f32 raw clips -> [B][L]: gain, pad/truncate, roll, waveform mixup (pa_wave_augment)."""
    B, ldx = x.shape
    out = torch.empty((B, L), device=x.device, dtype=torch.float32)
    ws = torch.empty(B, device=x.device, dtype=torch.float32) if partner is not None else None
    check(_lib.load().pa_wave_augment(_p(x, torch.float32, True), B, x.stride(0), _p(lengths, torch.int32), _p(amp, torch.float32), _p(shift, torch.int32), _p(partner, torch.int32), _p(lam, torch.float32), _p(ws),
                                      _p(out), L, _stream()), "pa_wave_augment")
    return out


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step):
    check(_lib.load().pa_adamw(_p(p, torch.float32), _p(g, torch.float32), _p(m, torch.float32), _p(v, torch.float32), p.numel(), lr, beta1, beta2, eps, wd, step, _stream()),
          "pa_adamw")


def adamw_dev(p, g, m, v, hyper_dev):
    """the same update with the step's scalars in device memory (7 floats, see adamw_hyper): capturable in a hipGraph"""
    check(_lib.load().pa_adamw_dev(_p(p, torch.float32), _p(g, torch.float32), _p(m, torch.float32), _p(v, torch.float32), p.numel(),
                                   _p(hyper_dev, torch.float32), _stream()), "pa_adamw_dev")


def adamw_hyper(lr, beta1, beta2, eps, wd, step, out_host):
    """fill the 7-float HOST tensor `out_host` with [lr, beta1, beta2, eps, wd, 1 - beta1^step, sqrt(1 - beta2^step)]"""
    assert out_host.dtype == torch.float32 and out_host.numel() == 7 and not out_host.is_cuda
    _lib.load().pa_adamw_hyper(lr, beta1, beta2, eps, wd, int(step), C.cast(out_host.data_ptr(), C.POINTER(C.c_float)))


def swa_update(avg, p, num_averaged):
    """avg = p if num_averaged == 0 else avg + (p - avg) / (num_averaged + 1), flat f32 buffers."""
    check(_lib.load().pa_swa_update(_p(avg, torch.float32), _p(p, torch.float32), p.numel(), int(num_averaged), _stream()), "pa_swa_update")


def sgd(p, g, lr):
    check(_lib.load().pa_sgd(_p(p, torch.float32), _p(g, torch.float32), p.numel(), lr, _stream()), "pa_sgd")
