"""ctypes binding of libpasst_amd.so (C ABI: include/passt_amd.h).

The product path has NO fallback: if the shared library is missing or a call fails, we raise.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C passt_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpasst_amd.so")
# A/B benchmarking of two builds of the library: PASST_AMD_LIB=/path/to/other.so (same ABI)
LIB_PATH = os.environ.get("PASST_AMD_LIB", LIB_PATH)

PA_F32, PA_BF16 = 0, 1
EPI_STORE, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_PARTIAL = 0, 1, 2, 3, 4

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class MelParams(C.Structure):
    _fields_ = [("n_fft", i32), ("hop", i32), ("n_mels", i32), ("n_frames", i32),
                ("preemph", f32), ("mel_low", f32), ("inv_mel_delta", f32), ("log_eps", f32),
                ("out_add", f32), ("out_scale", f32),
                ("fmask_start", i32), ("fmask_end", i32), ("tmask_start", i32), ("tmask_end", i32)]


class GemmArgs(C.Structure):
    _fields_ = [("dtype", i32), ("epilogue", i32), ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("ldb", i32), ("A", vp), ("B", vp), ("bias", vp), ("resid", vp),
                ("ldr", i32), ("row_mod", i32), ("out_batch_rows", i32), ("out_row_off", i32),
                ("aux", vp), ("ldaux", i32), ("out_f32", vp), ("ldo32", i32), ("out_lp", vp),
                ("ldolp", i32), ("out_lp2", vp), ("ldolp2", i32), ("split_k", i32), ("tune", i32),
                ("colsum_out", vp), ("colsum_ws", vp), ("colsum_accumulate", i32), ("reserved", i32),
                ("colscale_n", i32), ("colscale", f32)]


class ReduceDesc(C.Structure):         # == pa_reduce_desc
    _fields_ = [("partial", vp), ("out", vp), ("n", i64), ("splits", i32), ("accumulate", i32), ("pitch", i64), ("mode", i32),
                ("reserved", i32)]


class StageDesc(C.Structure):          # == pa_stage_desc
    _fields_ = [("src", vp), ("dst", vp), ("dst_t", vp), ("rows", i32), ("cols", i32), ("tile_begin", i32), ("reserved", i32)]


class AdamwStageDesc(C.Structure):     # == pa_adamw_stage_desc
    _fields_ = [("offset", i64), ("dst", vp), ("dst_t", vp), ("rows", i32), ("cols", i32), ("tile_begin", i32), ("reserved", i32)]


# name -> (restype, argtypes); must list every symbol include/passt_amd.h declares
SIGNATURES = {
    "pa_abi_version": (i32, []),
    "pa_error_string": (C.c_char_p, [i32]),
    "pa_last_hip_error": (C.c_char_p, []),
    "pa_mel_num_frames": (i32, [i32, i32]),
    "pa_mel_frontend_fwd": (i32, [vp, i32, i32, vp, vp, vp, vp, C.POINTER(MelParams), vp]),
    "pa_convert_f32": (i32, [vp, vp, i64, i32, vp]),
    "pa_convert_to_f32": (i32, [vp, i32, vp, i64, vp]),
    "pa_transpose": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, vp]),
    "pa_stage_weights": (i32, [vp, i32, i32, i32, vp]),
    "pa_layernorm_fwd": (i32, [vp, vp, vp, vp, i32, vp, vp, i32, i32, f32, vp]),
    "pa_layernorm_bwd_ws_floats": (i64, [i32, i32]),
    "pa_layernorm_bwd": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, vp]),
    "pa_layernorm_bwd_rows": (i32, [i32]),
    "pa_layernorm_bwd_partial": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "pa_gemm_last_colsum_rows": (i32, []),
    "pa_gemm_colsum_ws_floats": (i64, [i32, i32]),
    "pa_gemm_blocked_pre_ok": (i32, [i32, i32, i32]),
    "pa_gemm_blocked_pre_elems": (i64, [i32, i32]),
    "pa_gemm_nt": (i32, [C.POINTER(GemmArgs), vp]),
    "pa_gemm_nt_splitk_plan": (i32, [i32, i32, i32, i32, i32]),
    "pa_gemm_nt_splitk_ws_floats": (i64, [i32, i32, i32, i32, i32]),
    "pa_gemm_nt_splitk": (i32, [C.POINTER(GemmArgs), vp, i64, vp]),
    "pa_gemm_tn": (i32, [C.POINTER(GemmArgs), vp]),
    "pa_gemm_tn_batched": (i32, [C.POINTER(GemmArgs), i32, vp]),
    "pa_gemm_tn_step_rows": (i32, []),
    "pa_reduce_partials_batched": (i32, [C.POINTER(ReduceDesc), i32, vp]),
    "pa_colsum_ws_floats": (i64, [i32, i32]),
    "pa_colsum": (i32, [vp, i32, i32, i32, i32, vp, i32, vp, vp]),
    "pa_reduce_partials": (i32, [vp, i32, i64, vp, i32, vp]),
    "pa_rowsum": (i32, [vp, i32, i32, i32, i32, vp, i32, vp]),
    "pa_colsum_f32": (i32, [vp, i32, i32, i32, vp, i32, vp]),
    "pa_attention_fwd": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, f32, i32, i32, vp]),
    "pa_attention_bwd_ws_floats": (i64, [i32, i32, i32]),
    "pa_attention_bwd": (i32, [vp, i32, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, i32, vp]),
    "pa_gather_rows": (i32, [vp, vp, i32, i64, vp, vp]),
    "pa_scatter_rows": (i32, [vp, vp, i32, i64, vp, vp]),
    "pa_zero2d": (i32, [vp, i64, i64, i64, vp]),
    "pa_patch_gather": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    "pa_patch_pos_table": (i32, [vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp]),
    "pa_patch_bwd": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp]),
    "pa_head_pre_fwd": (i32, [vp, i32, i32, i32, vp, vp, f32, vp, vp, f32, vp, vp, vp, vp]),
    "pa_linear_f32_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "pa_linear_f32_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "pa_head_pre_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "pa_bce_fwd_bwd": (i32, [vp, vp, i32, i32, f32, vp, vp, vp, vp]),
    "pa_ce_mixup_fwd_bwd": (i32, [vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]),
    "pa_mixup": (i32, [vp, vp, vp, vp, i32, i64, vp]),
    "pa_adamw": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]),
    "pa_adamw_dev": (i32, [vp, vp, vp, vp, i64, vp, vp]),
    "pa_adamw_stage": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, f32, f32, f32, i32, vp, vp]),
    "pa_adamw_hyper": (None, [f32, f32, f32, f32, f32, i32, C.POINTER(f32)]),
    "pa_sgd": (i32, [vp, vp, i64, f32, vp]),
    "pa_swa_update": (i32, [vp, vp, i64, i32, vp]),
    "pa_wave_augment": (i32, [vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "pa_comm_unique_id": (i32, [vp]),
    "pa_comm_init": (i32, [vp, i32, i32, C.POINTER(vp)]),
    "pa_allreduce_bucket": (i32, [vp, vp, i64, i32, vp]),
    "pa_comm_destroy": (i32, [vp]),
    "pa_comm_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "pa_comm_last_error": (C.c_char_p, []),
}
GEMM_BLOCKED_PRE = 0x100      # pa_gemm_args.reserved flags (include/passt_amd.h)
GEMM_NO_PERSIST = 0x400
GEMM_COLSUM_DEFER = 0x2000    # PA_EPI_DGELU column sums stay as partial rows for the block's one finishing launch
REDUCE_SLABS, REDUCE_ROWS = 0, 1    # pa_reduce_desc.mode
GEMM_EPILOGUE_V3 = 0x1000     # run a pa_gemm_nt call with the LDS-free epilogue (A/B, equality test; slower: opt-in)
COMM_ID_BYTES = 128

_lib = None


class PasstAmdError(RuntimeError):
    pass


def load():
    """Load (once) and type the shared library.  Raises if it is not built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise PasstAmdError(
            f"{LIB_PATH} not found: the HIP extension is not built.  Run "
            "`make -C passt_amd/csrc` (needs hipcc, --offload-arch=gfx950).  passt_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    if lib.pa_abi_version() != 6:      # include/passt_amd.h PA_ABI_VERSION
        raise PasstAmdError("libpasst_amd.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def compile_opaque(fn):
    """``torch.compiler.disable`` for a forward that sequences ctypes launches -- WITHOUT importing ``torch._dynamo``.

    torch.compile must see PaSST.forward / AugmentMelSTFT.forward as ONE opaque eager call whichever way a compile is set up:
    ``torch.compile(net)`` (ex_audioset.py:135), ``net.compile()``, ``torch.compile(net.forward)``, ``torch.compile(step_fn)``
    with ``net(x)`` somewhere inside.  The wrapper is installed at class definition, so no flow can reach the undecorated
    function.  ``torch.compiler.disable`` itself would import torch._dynamo (~900 modules, millions of GC-tracked objects)
    into every process that only wants the eager path: measured +6 ms per training step at ESC-50's batch 12 with bench.py's
    event bookkeeping (profiles/r05_dynamo_import_gc.txt).  What that decorator does is small and lives in torch._C, which is
    always loaded: (i) the marker attributes dynamo's tracer looks for (``_torchdynamo_disable``: the call becomes a graph
    break, not inlined), (ii) ``set_eval_frame(None)`` around the call (nothing below is handed to the frame evaluator), (iii)
    the wrapper's own code object marked "skip" for the frame evaluator (torch's own wrapper gets that from living in a
    skip-listed file).  No ``_torchdynamo_orig_callable``: ``torch.compile(net.forward)`` would unwrap it to the unbound
    function and lose ``self``.  On a torch build without these private entry points: fall back to torch.compiler.disable."""
    import functools

    import torch
    try:
        ef = torch._C._dynamo.eval_frame
        set_eval_frame = ef.set_eval_frame
        skip = ef._FrameExecStrategy(ef._FrameAction.SKIP, ef._FrameAction.DEFAULT)
    except AttributeError:
        return torch.compiler.disable(fn)

    @functools.wraps(fn)
    def opaque_forward(*args, **kwargs):
        prior = set_eval_frame(None)
        try:
            return fn(*args, **kwargs)
        finally:
            set_eval_frame(prior)

    ef.set_code_exec_strategy(opaque_forward.__code__, skip)
    opaque_forward._torchdynamo_disable = True
    opaque_forward._torchdynamo_disable_msg = "passt_amd: hand-written HIP kernels behind a C ABI (ctypes); nothing to trace"
    opaque_forward._torchdynamo_disable_recursive = True
    return opaque_forward


def check(rc, what=""):
    if rc != 0:
        lib = load()
        msg = lib.pa_error_string(rc).decode()
        if rc == -3:
            msg += ": " + lib.pa_last_hip_error().decode()
        if rc == -4:
            msg += ": " + lib.pa_comm_last_error().decode()
        raise PasstAmdError(f"{what} failed: {msg} (code {rc})")
