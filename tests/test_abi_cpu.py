"""CPU-only checks of the drop-in boundary: the shared library loads without a GPU, exports every
symbol include/passt_amd.h declares, the ctypes structs match the C layout, and the product path
refuses to run without a HIP device (no silent fallback)."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "passt_amd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    from passt_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert n in _lib.SIGNATURES, f"{n} declared in the header but not bound in passt_amd/_lib.py"
        assert hasattr(lib, n), f"{n} not exported by libpasst_amd.so"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.pa_abi_version() == 6
    assert lib.pa_error_string(-2) == b"unsupported shape or dtype"
    assert lib.pa_mel_num_frames(320000, 320) == 1000          # 10 s -> 1000 frames
    assert lib.pa_layernorm_bwd_ws_floats(30336, 768) == 1024 * 3 * 768 and lib.pa_layernorm_bwd_rows(30336) == 1024
    assert lib.pa_layernorm_bwd_rows(4236) == 530 and lib.pa_layernorm_bwd_ws_floats(4236, 768) == 530 * 3 * 768


def test_ctypes_structs_match_the_c_layout():
    from passt_amd import _lib
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "passt_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(pa_gemm_args), offsetof(pa_gemm_args, A), offsetof(pa_gemm_args, resid),
         offsetof(pa_gemm_args, aux), offsetof(pa_gemm_args, out_f32), offsetof(pa_gemm_args, out_lp2),
         offsetof(pa_gemm_args, tune), sizeof(pa_mel_params), offsetof(pa_gemm_args, colsum_out),
         offsetof(pa_gemm_args, colsum_accumulate), sizeof(pa_stage_desc), offsetof(pa_gemm_args, colscale_n),
         offsetof(pa_gemm_args, colscale), sizeof(pa_reduce_desc), offsetof(pa_reduce_desc, pitch), offsetof(pa_reduce_desc, mode),
         sizeof(pa_adamw_stage_desc), offsetof(pa_adamw_stage_desc, dst_t), offsetof(pa_adamw_stage_desc, tile_begin));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    G = _lib.GemmArgs
    got = [ctypes.sizeof(G), G.A.offset, G.resid.offset, G.aux.offset, G.out_f32.offset, G.out_lp2.offset,
           G.tune.offset, ctypes.sizeof(_lib.MelParams), G.colsum_out.offset, G.colsum_accumulate.offset,
           ctypes.sizeof(_lib.StageDesc), G.colscale_n.offset, G.colscale.offset, ctypes.sizeof(_lib.ReduceDesc),
           _lib.ReduceDesc.pitch.offset, _lib.ReduceDesc.mode.offset, ctypes.sizeof(_lib.AdamwStageDesc),
           _lib.AdamwStageDesc.dst_t.offset, _lib.AdamwStageDesc.tile_begin.offset]
    assert got == [int(v) for v in out]


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    from passt_amd import _lib
    lib = _lib.load()
    a = _lib.GemmArgs()
    assert lib.pa_gemm_nt(ctypes.byref(a), None) == -1                        # null operands
    assert lib.pa_layernorm_fwd(None, None, None, None, 0, None, None, 4, 4, 1e-6, None) == -1
    p = _lib.MelParams()
    assert lib.pa_mel_frontend_fwd(None, 1, 100, None, None, None, None, ctypes.byref(p), None) == -1
    # ABI 5: the attention-backward form flags are validated before anything is launched; unknown bits are an error
    buf = (ctypes.c_float * 64)()
    q = ctypes.cast(buf, ctypes.c_void_p)
    for flags, want in ((16, -1), (1 | 32, -1)):          # (8 = PA_ATTN_BWD_SINGLE_PASS_W16 since ABI 6)
        assert lib.pa_attention_bwd(q, 192, q, q, 64, q, q, q, 192, 1, 1, 4, 4, ctypes.c_float(0.125), 1, flags, None) == want
    assert lib.pa_attention_fwd(q, 192, q, 64, q, 1, 1, 4, 4, ctypes.c_float(0.125), 1, 2, None) == -1      # backward-only flag
    # the batched finishing reduction: mode / pitch / count checks
    d = (_lib.ReduceDesc * 13)()
    for e in d:
        e.partial, e.out, e.n, e.splits, e.mode, e.pitch = q, q, 16, 2, _lib.REDUCE_ROWS, 16
    assert lib.pa_reduce_partials_batched(d, 13, None) == -1                   # > PA_REDUCE_BATCH_MAX
    d[0].mode = 7
    assert lib.pa_reduce_partials_batched(d, 1, None) == -1
    d[0].mode, d[0].pitch = _lib.REDUCE_ROWS, 8                               # rows shorter than n
    assert lib.pa_reduce_partials_batched(d, 1, None) == -1
    assert lib.pa_layernorm_bwd_partial(None, 1, None, None, None, None, None, None, None, None, 4, 4, None) == -1
    # ABI 6: the fused optimizer + staging entry: null table, empty table, unknown dtype, step 0 without device hyper-parameters
    sd = (_lib.AdamwStageDesc * 1)()
    f = ctypes.c_float
    for descs, n, items, dt, step in ((None, 1, 1, 1, 1), (sd, 0, 1, 1, 1), (sd, 1, 0, 1, 1), (sd, 1, 1, 7, 1), (sd, 1, 1, 1, 0)):
        assert lib.pa_adamw_stage(q, q, q, q, descs, n, items, dt, f(1e-3), f(0.9), f(0.999), f(1e-8), f(0.0), step, None, None) == -1
    assert lib.pa_layernorm_bwd_rows(0) == 0 and lib.pa_layernorm_bwd_rows(7) == 1


def test_product_path_has_no_cpu_fallback():
    import passt_amd
    from passt_amd._lib import PasstAmdError
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = passt_amd.PaSST(img_size=(128, 100), stride=10, embed_dim=128, depth=1, num_heads=2, num_classes=5,
                            distilled=True)
    with pytest.raises(PasstAmdError):
        m(torch.zeros(1, 1, 128, 100))
    mel = passt_amd.AugmentMelSTFT(fmax=15000)
    with pytest.raises(PasstAmdError):
        mel(torch.zeros(1, 32000))
    # and nothing in the product imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "passt_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(root, f)).read().replace("# oracle", ""), f


def test_state_dict_schema_and_parameter_order():
    import passt_amd
    from oracle import detgen
    from oracle import passt_oracle as O
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = passt_amd.get_model(arch="passt_s_swa_p16_128_ap476", pretrained=False, n_classes=527)
    sd = m.state_dict()
    want = detgen.passt_state_dict(O.make_cfg(), 0)
    assert list(sd.keys()) == list(want.keys()) or sorted(sd.keys()) == sorted(want.keys())
    for k, v in want.items():
        assert tuple(sd[k].shape) == v.shape, k
    assert sum(p.numel() for p in m.parameters()) == 86153758
    names = [n for n, _ in m.named_parameters()]
    assert names[:5] == ["cls_token", "dist_token", "new_pos_embed", "freq_new_pos_embed", "time_new_pos_embed"]
    assert names[-2:] == ["head_dist.weight", "head_dist.bias"]
    from oracle import ref_import
    if ref_import.reference_available():
        ref = ref_import.build_reference_passt(O.make_cfg(embed_dim=128, depth=2, num_heads=2, num_classes=9,
                                                          img_size=(128, 100)),
                                               detgen.passt_state_dict(O.make_cfg(embed_dim=128, depth=2, num_heads=2,
                                                                                  num_classes=9, img_size=(128, 100)), 1))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mine = passt_amd.PaSST(img_size=(128, 100), stride=10, embed_dim=128, depth=2, num_heads=2, num_classes=9,
                                   distilled=True)
        assert [n for n, _ in ref.named_parameters()] == [n for n, _ in mine.named_parameters()]
        mine.load_state_dict(ref.state_dict(), strict=True)


def test_host_index_logic_matches_reference_order():
    """passt_amd's kept-patch enumeration == the reference's embed-then-index order (oracle semantics)."""
    import passt_amd
    from passt_amd.passt import draw_patchout, kept_patches
    from oracle import passt_oracle as O
    for (st, sf, u, T) in ((6, 3, 5, 250), (0, 2, 0, 250), (4, 0, 9, 180), (0, 0, 0, 250)):
        cfg = O.make_cfg(embed_dim=128, depth=1, num_heads=2, num_classes=3, img_size=(128, 250), stride=(10, 10),
                         s_patchout_t=st, s_patchout_f=sf, u_patchout=u)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = passt_amd.PaSST(u_patchout=u, s_patchout_t=st, s_patchout_f=sf, img_size=(128, 250), stride=10,
                                embed_dim=128, depth=1, num_heads=2, num_classes=3, distilled=True).train()
        Fd, Td = 12, (T - 16) // 10 + 1
        torch.manual_seed(42)
        d = O.draw_patchout(cfg, Fd, Td, True)
        torch.manual_seed(42)
        toff, T_eff, it, if_, iu = draw_patchout(m, Fd, Td)
        assert toff == d["toff"] and T_eff == d["T_eff"]
        # reference order on a grid of (f, t) labels
        grid_f = torch.arange(Fd).view(Fd, 1).expand(Fd, T_eff)
        grid_t = torch.arange(T_eff).view(1, T_eff).expand(Fd, T_eff)
        gf, gt = grid_f, grid_t
        if d["idx_t"] is not None:
            gf, gt = gf[:, d["idx_t"]], gt[:, d["idx_t"]]
        if d["idx_f"] is not None:
            gf, gt = gf[d["idx_f"], :], gt[d["idx_f"], :]
        gf, gt = gf.reshape(-1), gt.reshape(-1)
        if d["idx_u"] is not None:
            gf, gt = gf[d["idx_u"]], gt[d["idx_u"]]
        pf, pt = kept_patches(Fd, T_eff, it, if_, iu)
        assert np.array_equal(pf, gf.numpy()) and np.array_equal(pt, gt.numpy())


def test_reference_module_paths_resolve_to_this_implementation():
    """`models.passt` / `models.preprocess` (the paths ex_audioset.py:61-70 configures) are served by passt_amd."""
    import importlib
    import sys
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    mp = importlib.import_module("models.passt")
    mpre = importlib.import_module("models.preprocess")
    import passt_amd
    assert mp.get_model is passt_amd.passt.get_model and mp.PaSST is passt_amd.PaSST
    assert mpre.AugmentMelSTFT is passt_amd.AugmentMelSTFT
    assert mp.get_model_passt is mp.get_model


def test_front_end_module_surface_matches_the_reference():
    """models/preprocess.py:47-54: freqm / timem are torchaudio FrequencyMasking / TimeMasking modules (nn.Identity for 0)
    -- print(mel), mel.freqm.mask_param and the Identity check read the same here; no parameters, empty state_dict."""
    import contextlib
    import io

    import torch

    import passt_amd
    with contextlib.redirect_stdout(io.StringIO()):
        mel = passt_amd.AugmentMelSTFT(freqm=48, timem=192)
        off = passt_amd.AugmentMelSTFT(freqm=0, timem=0)
    text = repr(mel)
    assert "(freqm): FrequencyMasking()" in text and "(timem): TimeMasking()" in text and "winsize=800, hopsize=320" in text
    assert mel.freqm.mask_param == 48 and mel.timem.mask_param == 192 and mel.freqm.iid_masks and mel.timem.axis == 2
    assert isinstance(off.freqm, torch.nn.Identity) and isinstance(off.timem, torch.nn.Identity)
    assert len(mel.state_dict()) == 0 and not list(mel.parameters())


def test_comm_entry_points_without_a_gpu():
    """pa_comm_*: argument checks and the out-of-band id need no device (RCCL itself is loaded lazily by the library)."""
    import ctypes as C
    from passt_amd import _lib
    lib = _lib.load()
    assert lib.pa_comm_init(None, 0, 1, None) == -1 and lib.pa_allreduce_bucket(None, None, 0, 0, None) == -1
    assert lib.pa_comm_destroy(None) == -1 and lib.pa_comm_unique_id(None) == -1
    buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
    rc = lib.pa_comm_unique_id(buf)
    if rc == 0:
        assert any(buf.raw)                                   # an ncclUniqueId was produced
    else:                                                     # a box without librccl: a clean error, not a crash
        assert rc == -4 and b"rccl" in lib.pa_comm_last_error().lower()


def test_no_kernel_of_the_library_uses_scratch():
    """A register spill inside a hot loop once cost 75 us per launch without failing any parity test: every kernel of the
    default bf16 step must fit its registers (.private_segment_fixed_size == 0 in the code objects' metadata)."""
    so = os.path.join(ROOT, "passt_amd", "libpasst_amd.so")
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(so) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or llvm tools not available")
    objcopy, bundler, readelf = tools
    names, sizes = [], []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", so, os.path.join(d, "copy.so")], check=True, capture_output=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        assert starts, "no offload bundles in .hip_fatbin"
        for k, st in enumerate(starts):                       # one bundle per translation unit
            part, co = os.path.join(d, f"b{k}.bin"), os.path.join(d, f"b{k}.co")
            open(part, "wb").write(blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            r = subprocess.run([bundler, "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}",
                                f"--output={co}"], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split(".agpr_count")[1:]:       # one metadata map per kernel
                n = re.search(r"\.name:\s+(\S+)", blk)
                z = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
                if n and z:
                    names.append(n.group(1))
                    sizes.append(int(z.group(1)))
    assert len(names) > 40, len(names)
    # A/B-only tile variants and wide LayerNorm instantiations may spill; the kernels of the default bf16 step must not
    spilled = [n for n, z in zip(names, sizes) if z > 0]
    # (attn_bwd_fused_kernel<true> = the sixteen-wave A/B form of round 6: 72 B of loop-invariant addresses, profiles/r06_attention_w16.txt)
    hot = [n for n in spilled if "gemm_tn_stagger" in n or ("DF16b" in n and ("gemm_nt_stagger_kernel" in n or "attn_" in n))
           and "attn_bwd_fused_kernelILb1E" not in n]
    assert not hot, hot


def test_host_side_queries_answer_without_a_gpu():
    """Pure host entry points of the C ABI (no device touched): the blocked pre-activation query / size and the token
    step of the weight-gradient kernel that callers size their split-K slices with."""
    from passt_amd import _lib
    lib = _lib.load()
    assert lib.pa_gemm_tn_step_rows() == 48
    # headline MLP shape: role-split kernel -> blocked form available; rows padded to whole 256-row tiles
    assert lib.pa_gemm_blocked_pre_ok(64 * 474, 3072, 768) == 1
    assert lib.pa_gemm_blocked_pre_elems(64 * 474, 3072) == 30720 * 3072      # rows rounded up to 768 = lcm of the tile heights
    assert lib.pa_gemm_blocked_pre_elems(256, 64) == 768 * 64
    # compact rows of the last block (2 per clip) run on the generic kernel; N must be a multiple of 64
    assert lib.pa_gemm_blocked_pre_ok(128, 3072, 768) == 0
    assert lib.pa_gemm_blocked_pre_ok(64 * 474, 3080, 768) == 0
    assert lib.pa_gemm_blocked_pre_ok(0, 3072, 768) == 0
    # split-K plan of the NT GEMM: only problems that leave most CUs idle and have a long K; never at the headline shapes
    S, R, G, BF, F32 = _lib.EPI_STORE, _lib.EPI_RESID, _lib.EPI_GELU, _lib.PA_BF16, _lib.PA_F32
    assert lib.pa_gemm_nt_splitk_plan(12 * 353, 768, 3072, R, BF) == 2        # ESC-50 batch 12: 102 tiles of 128 x 256
    assert lib.pa_gemm_nt_splitk_plan(12 * 353, 768, 2304, S, BF) == 2
    assert lib.pa_gemm_nt_splitk_plan(24, 768, 3072, S, BF) == 8              # prefix-only tail: 3 tiles, 48 K-tiles
    assert lib.pa_gemm_nt_splitk_plan(12 * 353, 768, 768, S, BF) == 1         # short K
    assert lib.pa_gemm_nt_splitk_plan(64 * 474, 768, 3072, R, BF) == 1        # headline: 474 tiles
    assert lib.pa_gemm_nt_splitk_plan(12 * 353, 2304, 768, S, BF) == 1
    assert lib.pa_gemm_nt_splitk_plan(24, 768, 3072, S, F32) == 1 and lib.pa_gemm_nt_splitk_plan(24, 3072, 3072, G, BF) == 1
    assert lib.pa_gemm_nt_splitk_ws_floats(24, 768, 3072, S, BF) == 8 * 24 * 768
    assert lib.pa_gemm_nt_splitk_ws_floats(64 * 474, 768, 3072, R, BF) == 0


def test_attention_isa_never_touches_in_flight_lds_fragments():
    """tools/check_lds_asm.py on the compiled attention kernels: the asm-issued transposed LDS reads are settled by counted
    waits the compiler does not know about, so no instruction may read or overwrite their destination registers before the
    wait (round 3: a register copy at a control-flow merge did, and output rows were sporadically garbage at B = 64)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "passt_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "attention.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                        "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                        os.path.join(src, "attention.hip"), "-o", out], check=True, capture_output=True)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_asm.py"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


def test_shipped_library_carries_the_cache_policy():
    """Round 6 (profiles/r06_cache_policy.txt): the step's GEMMs mark what they touch once non-temporal -- outputs, epilogue
    operand rows, the A operand of the residual GEMMs -- and nothing else (weights and the other A operands stay on the default policy;
    attention and LayerNorm kernels carry no hint: both measured slower with one).  Checked on the ISA of the BUILT library
    (tools/so_isa.py disassembles the code objects embedded in libpasst_amd.so): a build with -DPA_NO_CACHE_POLICY, or a lost `aux`
    operand, would go unnoticed by every numerical test."""
    import importlib.util
    from passt_amd import _lib
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("no llvm-objdump / library not built")
    spec = importlib.util.spec_from_file_location("so_isa", os.path.join(ROOT, "tools", "so_isa.py"))
    so_isa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(so_isa)
    cos = so_isa.code_objects(_lib.LIB_PATH)
    by_kernel = {}
    for _, text in cos:
        by_kernel.update(so_isa.kernel_bodies(text))

    def counts(fragment):
        hits = [k for k in by_kernel if fragment in k]
        assert len(hits) == 1, (fragment, hits)
        body = by_kernel[hits[0]]
        c = so_isa.policy_counts(body)
        c["lds_dma"] = body.count("global_load_lds_dwordx4")
        return c
    # <bf16, EPI, TM, A3, BLK, TR>: the kernels config #2 runs (pick_nt_variant): store / residual TM 3 with A two tiles ahead, MLP TM 4 blocked
    store, gelu = counts("gemm_nt_stagger_kernelIDF16bLi0ELi3ELb1ELb0ELb0E"), counts("gemm_nt_stagger_kernelIDF16bLi1ELi4ELb0ELb1ELb0E")
    resid, dgelu = counts("gemm_nt_stagger_kernelIDF16bLi2ELi3ELb1ELb0ELb0E"), counts("gemm_nt_stagger_kernelIDF16bLi3ELi4ELb0ELb1ELb0E")
    assert store["buffer_store nt"] >= 8 and store["lds_dma nt"] == 0
    assert gelu["buffer_store nt"] >= 16 and gelu["lds_dma nt"] == 0                  # activation + blocked pre-activation
    assert resid["buffer_load nt"] >= 8 and resid["buffer_store nt"] == 0             # residual rows in; the f32 stream out stays cached
    assert 0 < resid["lds_dma nt"] < resid["lds_dma"]                                 # A non-temporal, the weights not
    assert dgelu["buffer_store nt"] >= 8 and dgelu["buffer_load nt"] >= 8 and dgelu["lds_dma nt"] == 0
    assert counts("adamw_stage_kernelIDF16b")["global nt"] > 0
    for frag in ("attn_fwd_kernelIDF16bLb1E", "attn_bwd_fused_kernelILb0E", "ln_fwd_kernelIDF16bLi3E", "ln_bwd_kernelIDF16bLi3E", "gemm_tn_stagger_batched_kernel"):
        c = counts(frag)
        assert c["buffer_store nt"] == c["buffer_load nt"] == c["lds_dma nt"] == c["global nt"] == 0, (frag, c)


def test_single_pass_attention_backward_index_math_on_the_cpu():
    """tools/emulate_attn_bwd_fused.py: the transposition buffer T of attn_bwd_fused_kernel (phase-1 write addresses, XOR swizzle)
    and the phase-2 operand fetches (ds_read_b64_tr_b16 piece addresses for K^T and dS^T, 16x16x32 MFMA k-slot order) reproduce
    dQ^T = K^T dS^T exactly on integer data, and every access pattern is bank-conflict free under the MI355X lane-group rules.
    The formulas in the tool are the kernel's; change both together."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emulate_attn_bwd_fused.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "phase-2 values OK" in r.stdout, r.stdout[-800:] + r.stderr[-800:]


def test_front_end_band_stage_emulated_on_the_cpu():
    """mel.hip's band stage (lane-local segment recurrence + one DPP segmented scan across the wave) statement by statement in
    numpy against the definition of the sparse filterbank product, over random geometries"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("emulate_mel_bands", os.path.join(ROOT, "tools", "emulate_mel_bands.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.self_check(trials=40, seed=3) < 5e-6
