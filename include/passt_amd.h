/* passt_amd.h -- C ABI of libpasst_amd.so, the MI355X (gfx950) implementation of the
 * kkoutini/PaSST training hot path.
 *
 * The reference has no FFI of its own: its hot path is a chain of ATen / torchaudio calls inside
 * two nn.Modules (models/preprocess.py:19 AugmentMelSTFT, models/passt.py:383 PaSST).  This
 * header is the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md); each entry
 * point names the reference call sites (file:line under the reference root) it replaces.
 *
 * Conventions
 *   - plain C types only: raw DEVICE pointers, ints, floats; no torch types.
 *   - the CALLER allocates every output and workspace; the library never allocates or frees
 *     device memory and keeps no pointer after return.
 *   - every call is asynchronous and ordered on `stream` (a hipStream_t passed as void*);
 *     nothing synchronises the device.  Re-entrant: no mutable global state.
 *   - return value: 0 on success, a negative PA_E* code otherwise (never throws);
 *     pa_error_string(code) describes it.
 *   - `dtype` selects the storage/compute type of the "low precision" activations and GEMM
 *     operands: PA_F32 (exact-f32 MFMA, the <=1e-3 parity mode) or PA_BF16 (bf16 MFMA with f32
 *     accumulation, the throughput mode).  The residual stream, LayerNorm statistics, softmax,
 *     master weights and all weight gradients are always f32.
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef PASST_AMD_H
#define PASST_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_ABI_VERSION 6   /* 2 (round 3): pa_attention_* take flags, pa_attention_bwd workspace query, pa_gemm_args.colscale*;
                            * 3: pa_gemm_nt_splitk*;  4 (round 4): pa_comm_info, pa_adamw_dev / pa_adamw_hyper, PA_GEMM_EPILOGUE_V3;
                            * 5 (round 5): PA_ATTN_BWD_TWO_PASS (pa_attention_bwd defaults to the single-pass kernel where it applies);
                            * 6 (round 6): pa_adamw_stage (optimizer update + GEMM-ready weight copies in one pass) */

enum { PA_F32 = 0, PA_BF16 = 1 };

enum {
    PA_OK = 0,
    PA_EINVAL = -1,   /* bad argument (null pointer, negative size, ...) */
    PA_EUNSUPPORTED = -2, /* shape/dtype outside what the kernels cover */
    PA_ELAUNCH = -3,  /* hipLaunch failed; see pa_last_hip_error() */
    PA_ECOMM = -4     /* RCCL missing or a collective call failed; see pa_comm_last_error() */
};

int pa_abi_version(void);
const char* pa_error_string(int code);
/* hipGetLastError() of the calling thread as text (valid until the next call). */
const char* pa_last_hip_error(void);

/* ------------------------------------------------------------------------------------------
 * Front end: AugmentMelSTFT.forward, models/preprocess.py:57-86 (K1..K8 in SURVEY.md 2.3)
 * pre-emphasis (:59) -> STFT 1024/hop/hann(win) reflect-centred (:60-61) -> power (:62) ->
 * kaldi mel filterbank as a sparse band product (:71-76) -> log(x+eps) (:78) -> SpecAugment
 * frequency/time mask (:80-82) -> (x+4.5)/5 (:84), fused in one kernel.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_fft;        /* must be 1024 */
    int32_t hop;          /* 320 */
    int32_t n_mels;       /* <= 128 */
    int32_t n_frames;     /* T = 1 + (L-1-1+... ) see pa_mel_num_frames */
    float preemph;        /* 0.97: y[n] = x[n+1] - preemph*x[n]  (:46,:59) */
    float mel_low;        /* mel(fmin) = 1127 ln(1+fmin/700) */
    float inv_mel_delta;  /* (n_mels+1) / (mel(fmax) - mel(fmin)) */
    float log_eps;        /* 1e-5 (:78) */
    float out_add;        /* 4.5  (:84) */
    float out_scale;      /* 0.2  (:84) */
    int32_t fmask_start, fmask_end;   /* [start,end) mel rows forced to (0+out_add)*out_scale; empty if start>=end */
    int32_t tmask_start, tmask_end;   /* same along time */
} pa_mel_params;

/* frames produced for a waveform of L samples: 1 + (L-1)/hop  (pre-emphasis shortens by one) */
int pa_mel_num_frames(int L, int hop);
/* wave[B][L] f32 -> out[B][n_mels][n_frames] f32.
 * window[n_fft]  : analysis window already zero-padded/centred to n_fft (:38-40 + torch.stft rule)
 * bin_mel[n_fft/2]: mel value of FFT bin k, 1127 ln(1 + k*sr/n_fft/700) (kaldi.get_mel_banks)
 * twiddle[n_fft/2][2]: (cos, -sin)(2 pi k / n_fft), k = 0..n_fft/2-1 */
int pa_mel_frontend_fwd(const float* wave, int B, int L, const float* window, const float* bin_mel,
                        const float* twiddle, float* out, const pa_mel_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Parameter staging (no reference counterpart: AMP autocast casts weights per op)
 * ------------------------------------------------------------------------------------------ */
/* out[i] = (dtype) in[i] */
int pa_convert_f32(const float* in, void* out, int64_t n, int dtype, void* stream);
/* out[i] = (float) in[i], in of `dtype` (the bf16 gradient wire of the data-parallel reducer, passt_amd/ddp.py) */
int pa_convert_to_f32(const void* in, int dtype, float* out, int64_t n, void* stream);
/* Batched form of the two calls below, for refreshing every GEMM-ready weight copy after an optimizer
 * step in ONE launch: entry e reads src[rows][cols] (f32, contiguous) and writes, where non-NULL,
 * dst[rows][cols] (straight cast) and dst_t[cols][rows] (transposed), both of `dtype` and contiguous.
 * `descs` is an array of n_desc entries in DEVICE memory; tile_begin is the running count of 64x64 tiles
 * (ceil(rows/64)*ceil(cols/64) per entry, in order); total_tiles is their sum. */
typedef struct pa_stage_desc {
    const float* src;
    void* dst;
    void* dst_t;
    int32_t rows, cols;
    int32_t tile_begin;
    int32_t reserved;
} pa_stage_desc;
int pa_stage_weights(const pa_stage_desc* descs, int n_desc, int total_tiles, int dtype, void* stream);
/* in[R][C] (ld = ldi) of dtype in_dtype -> out[C][ldo] of dtype out_dtype, out[c][r] = in[r][c];
 * columns r in [R, ldo) of out are zero-filled (K-padding for the weight-gradient GEMM). */
int pa_transpose(const void* in, int in_dtype, int R, int C, int ldi, void* out, int out_dtype,
                 int ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm: nn.LayerNorm in Block (models/passt.py:369,373,378-379; eps 1e-6 :426), final norm
 * (:450,:570) and head.0 (:463, eps 1e-5).  K15 in SURVEY.md.
 * ------------------------------------------------------------------------------------------ */
/* x[M][D] f32 -> y[M][D] (dtype), mean[M], rstd[M] f32 (saved for backward; may be NULL). */
int pa_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int dtype,
                     float* mean, float* rstd, int M, int D, float eps, void* stream);
/* dx[M][D] = (dres ? dres : 0) + LN'(dy); optional dx_lp (dtype) copy of dx for the next GEMM.
 * dgamma/dbeta[D] are OVERWRITTEN (or accumulated when accumulate != 0).
 * dcolsum[D] (optional): column sums of the OUTPUT dx -- the bias gradient of the Linear whose output fed
 * this LayerNorm through the residual stream (proj.bias for norm2, the previous block's fc2.bias for norm1),
 * fused here because the kernel already reduces over rows.
 * ws: f32 workspace of pa_layernorm_bwd_ws_floats(M, D) elements. */
int64_t pa_layernorm_bwd_ws_floats(int M, int D);
int pa_layernorm_bwd(const void* dy, int dtype, const float* x, const float* gamma,
                     const float* mean, const float* rstd, const float* dres, float* dx,
                     void* dx_lp, float* dgamma, float* dbeta, float* dcolsum, int accumulate, float* ws,
                     int M, int D, void* stream);
/* The same without the finishing launch (ABI 5): dx / dx_lp are complete, the parameter gradients stay as
 * pa_layernorm_bwd_rows(M) partial rows ws[row][3][D] = {dgamma | dbeta | column sums of dx}; the caller reduces them with
 * pa_reduce_partials_batched (PA_REDUCE_ROWS, pitch 3 D) -- together with the block's other finishing reductions. */
int pa_layernorm_bwd_rows(int M);
int pa_layernorm_bwd_partial(const void* dy, int dtype, const float* x, const float* gamma,
                             const float* mean, const float* rstd, const float* dres, float* dx,
                             void* dx_lp, float* ws, int M, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * GEMM  C[M][N] = A[M][K] * B[N][K]^T  (both operands K-contiguous), MFMA, f32 accumulation.
 * Replaces every nn.Linear of the path: qkv (models/passt.py:345), proj (:359), fc1+GELU
 * (:285-286), fc2 (:288), the patch-embed conv as an im2col GEMM (:323) and, with transposed
 * operands, all their input/weight gradients (autograd mm / addmm backward; K26).
 * ------------------------------------------------------------------------------------------ */
enum {
    PA_EPI_STORE = 0,      /* out_lp = acc + bias                                   */
    PA_EPI_GELU = 1,       /* out_lp = acc + bias ; out_lp2 = gelu_erf(out_lp)      */
    PA_EPI_RESID = 2,      /* out_f32[orow] = acc + bias + resid[rrow]  (f32 residual stream) */
    PA_EPI_DGELU = 3,      /* out_lp = acc * gelu_erf'(aux)                         */
    PA_EPI_PARTIAL = 4     /* split-K: out_f32[z][M][N] = partial sums (no bias)    */
};
typedef struct {
    int32_t dtype;         /* PA_F32 / PA_BF16: type of A, B, aux, out_lp, out_lp2 */
    int32_t epilogue;
    int32_t M, N, K;       /* K*sizeof(dtype) must be a multiple of 128 bytes */
    int32_t lda, ldb;
    const void* A;
    const void* B;
    const float* bias;     /* [N] or NULL */
    const float* resid;    /* PA_EPI_RESID: f32 [*][ldr] */
    int32_t ldr;
    /* PA_EPI_RESID row remap (patch-embed writes token rows and adds a per-patch table):
     * if row_mod > 0: rrow = m % row_mod, orow = (m / row_mod) * out_batch_rows + out_row_off + m % row_mod
     * else rrow = orow = m. */
    int32_t row_mod, out_batch_rows, out_row_off;
    const void* aux;       /* PA_EPI_DGELU: pre-activation [M][ldaux] (dtype) */
    int32_t ldaux;
    float* out_f32;        /* PA_EPI_RESID / PA_EPI_PARTIAL */
    int32_t ldo32;
    void* out_lp;
    int32_t ldolp;
    void* out_lp2;
    int32_t ldolp2;
    int32_t split_k;       /* PA_EPI_PARTIAL: number of K slices (grid.z); else must be 1 */
    int32_t tune;          /* 0 = library heuristic (tile quantisation x measured rates).  Forcing a variant
                            * (benchmarking / autotuning), pa_gemm_nt bf16: 1 = 128x128 tile, 4 waves, 2 workgroups/CU;
                            * 2 = 256x256 lockstep; 6 / 7 / 8 = role-split 256x256 / 192x256 / 128x256 (8 waves,
                            * staggered wave groups); 9 = 128x256, 4 waves, 64-byte stages, 2 workgroups/CU (slower: DESIGN.md 4.1).
                            * 17 / 18 = 7 / 8 with three A slots (A requested two K-tiles ahead; what 0 picks for them).
                            * 3 = 192x128, 4 waves, 80 KiB: 2 workgroups/CU (first-generation epilogue; with PA_GEMM_BLOCKED_PRE: epilogue
                            * v2 + blocked pre-activation; PA_NT_MLP_2WG=1 makes 0 pick it for the large MLP-epilogue GEMMs);
                            * 13 / 19 = 3 / 9 with epilogue v2 and row-major outputs (round 6, A/B: profiles/r06_gemm_variants_epi13.txt).
                            * pa_gemm_tn bf16: 1 = 128x128, otherwise role-split 256x256.  pa_gemm_tn_batched: tune of the
                            * FIRST problem = 2 orders the work items problem-major instead of slice-major (A/B only). */
    /* PA_EPI_DGELU only, optional: colsum_out[n] = (colsum_accumulate ? colsum_out[n] : 0) + sum_m out[m][n] (of the
     * f32 values, before rounding) -- the bias gradient of the Linear whose pre-activation is `aux` (fc1.bias),
     * reduced inside the epilogue instead of by a separate pass over out_lp.  colsum_ws: f32 workspace of
     * pa_gemm_colsum_ws_floats(M, N) elements.  NULL colsum_out: not computed. */
    float* colsum_out;
    float* colsum_ws;
    int32_t colsum_accumulate;
    int32_t reserved;      /* flags: 0, or PA_GEMM_BLOCKED_PRE (bits 0..7 are ignored by the product library) */
    /* PA_EPI_STORE only: out_lp[m][n] = (acc + bias[n]) * colscale for the columns n < colscale_n (a multiple of 64; 0 = off),
     * one rounding.  The qkv Linear writes q * head_dim^-0.5 * log2(e) this way for pa_attention_*(PA_ATTN_Q_PRESCALED). */
    int32_t colscale_n;
    float colscale;
} pa_gemm_args;
/* PA_EPI_GELU: out_lp (the pre-activation) is written, PA_EPI_DGELU: aux (the same tensor) is read, in the library's
 * blocked layout instead of row-major -- 4 KiB blocks of 32 rows x 64 columns in MFMA accumulator order, which both
 * epilogues move with contiguous 16-byte-per-lane accesses and no LDS transposition.  The buffer is opaque to the caller:
 * pa_gemm_blocked_pre_elems(M, N) elements (rows padded to whole tiles), ldolp / ldaux ignored.  Only for shapes where
 * pa_gemm_blocked_pre_ok(M, N, K) returns 1 (bf16, tune = 0); otherwise pa_gemm_nt returns PA_EUNSUPPORTED. */
#define PA_GEMM_BLOCKED_PRE 0x100
/* pa_gemm_nt, role-split kernels: one work item per workgroup (hardware dispatch) instead of the persistent form, for callers
 * that run another kernel next to the GEMMs -- the gradient all-reduce of data-parallel training (ex_audioset.py:488-489): a
 * persistent launch whose workgroups cannot all be resident at once takes twice as long.  Same results. */
#define PA_GEMM_NO_PERSIST 0x400
/* pa_gemm_nt, bf16 role-split kernels: run this call with the LDS-free epilogue (round 4: accumulators computed transposed,
 * rows loaded / stored straight from the registers) instead of the default one that transposes the output tile through LDS.
 * Bit-identical results; measured slower in the training step (partial-line stores), kept for A/B measurements and the
 * equality test.  PA_EPILOGUE_V3=1 in the environment selects it process-wide. */
#define PA_GEMM_EPILOGUE_V3 0x1000
/* PA_EPI_DGELU with colsum_out (ABI 5): leave the column sums as partial rows in colsum_ws (one per wave tile) and skip the
 * finishing launch; pa_gemm_last_colsum_rows() (the calling thread's last such call) says how many rows were written, the caller
 * reduces them with pa_reduce_partials_batched (PA_REDUCE_ROWS, pitch N).  colsum_out must still be non-NULL (it is not written). */
#define PA_GEMM_COLSUM_DEFER 0x2000
int pa_gemm_blocked_pre_ok(int M, int N, int K);
int64_t pa_gemm_blocked_pre_elems(int M, int N);
int64_t pa_gemm_colsum_ws_floats(int M, int N);
/* Memory policy (round 6, csrc/gemm.hip PA_AUX_*): the bf16 outputs (out_lp, out_lp2), the blocked pre-activation and the epilogue
 * operand rows (resid, aux) of the bf16 role-split kernels are written / read NON-TEMPORALLY, as is the A operand of PA_EPI_RESID
 * calls: they are complete and visible at the end of the launch like any other result; a consumer simply finds them in HBM rather
 * than in the Infinity Cache.  A library built with -DPA_NO_CACHE_POLICY uses the default policy everywhere (same results). */
int pa_gemm_nt(const pa_gemm_args* a, void* stream);
/* The same product for problems too small to fill the chip (few output tiles, long K: the [M][768] outputs of fc2 / the
 * fc1 and qkv input gradients at ESC-50 batch sizes, ex_esc50.py:40; the prefix-only tail of the last block): K is cut into
 * pa_gemm_nt_splitk_plan(...) slices whose f32 partial tiles go to the caller's workspace `ws` (at least
 * pa_gemm_nt_splitk_ws_floats(...) floats) and one elementwise pass applies the epilogue.  PA_EPI_STORE / PA_EPI_RESID
 * (row_mod == 0), bf16, tune == 0; whenever the plan is 1 slice -- or ws is NULL / too small -- this IS pa_gemm_nt(a).
 * Results differ from pa_gemm_nt only in the order of the f32 partial sums. */
int pa_gemm_nt_splitk_plan(int M, int N, int K, int epilogue, int dtype);
int64_t pa_gemm_nt_splitk_ws_floats(int M, int N, int K, int epilogue, int dtype);
int pa_gemm_nt_splitk(const pa_gemm_args* a, float* ws, int64_t ws_floats, void* stream);
/* Weight gradient  C[a->M][a->N] = sum_{m < a->K} A[m][a->M]^T B[m][a->N]  (A = dY, B = X, both row-major
 * [tokens][features] read IN PLACE, no transposed copies).  epilogue must be PA_EPI_PARTIAL: split-K over
 * the token axis, out_f32[split_k][M][N] partial slabs, finished by pa_reduce_partials (deterministic).
 * Optional (bf16 role-split kernel only, i.e. tune != 1): colsum_ws != NULL asks for the column sums of A as well --
 * colsum_ws[split_k][a->M] partial sums over each slice's tokens (the bias gradient of the Linear whose output
 * gradient is A: nn.Linear, models/passt.py:345), computed by one extra MFMA per phase in the tiles of the first
 * B-column block instead of a separate pass over A; finish with pa_reduce_partials(colsum_ws, split_k, a->M, ...).
 * colsum_out / colsum_accumulate are ignored here. */
int pa_gemm_tn(const pa_gemm_args* a, void* stream);
/* Up to PA_TN_BATCH_MAX bf16 weight-gradient problems in ONE launch (e.g. the four Linears of a transformer block,
 * once all their operands exist): the work items of all problems share one grid, so there is no drain / prologue
 * between problems and items of different length pack onto the CUs.  Same arguments and results as n calls of
 * pa_gemm_tn; `a` is a HOST array.  When all problems use the same split_k the items are dealt slice-major with
 * every XCD owning a contiguous run, so the tiles sharing a dY / X panel of a token slice share an L2. */
#define PA_TN_BATCH_MAX 4
/* tokens the bf16 role-split weight-gradient kernel consumes per pipeline stage: a K slice of pa_gemm_tn /
 * pa_gemm_tn_batched is ceil(ceil(K / PA_TN_STEP_ROWS) / split_k) such steps (callers sizing split_k use it) */
#ifndef PA_TN_STEP_ROWS
#define PA_TN_STEP_ROWS 48
#endif
int pa_gemm_tn_step_rows(void);    /* the value the library was built with */
int pa_gemm_tn_batched(const pa_gemm_args* a, int n, void* stream);
/* Row gather / scatter and strided zero fill (prefix-token path of the last block):
 * gather: out[i] = in[idx[i]]; scatter: out[idx[i]] = in[i]; rows of row_bytes bytes (multiple of 4). */
int pa_gather_rows(const void* in, const int32_t* idx, int n_idx, int64_t row_bytes, void* out, void* stream);
int pa_scatter_rows(const void* in, const int32_t* idx, int n_idx, int64_t row_bytes, void* out, void* stream);
/* zero `rows` rows of width_bytes at pitch_bytes */
int pa_zero2d(void* ptr, int64_t pitch_bytes, int64_t width_bytes, int64_t rows, void* stream);
/* out[i] = (accumulate ? out[i] : 0) + sum_z partial[z][i], i < n */
int pa_reduce_partials(const float* partial, int splits, int64_t n, float* out, int accumulate,
                       void* stream);
/* the same for up to PA_REDUCE_BATCH_MAX reductions in one launch (`d` is a HOST array).  Two forms (ABI 5):
 *   mode PA_REDUCE_SLABS (0): out[i] (+)= sum_{z < splits} partial[z * n + i], i < n -- few slices of a long vector (the split-K
 *                             slabs of the weight gradients); 16-byte accesses, one thread per four outputs;
 *   mode PA_REDUCE_ROWS  (1): out[c] (+)= sum_{r < splits} partial[r * pitch + c], c < n -- MANY short rows (the per-workgroup
 *                             partial rows of pa_layernorm_bwd_partial, the per-wave-tile rows of a deferred PA_EPI_DGELU column
 *                             sum): 16 columns x 16 row groups per workgroup, four loads in flight per thread.
 * One launch per transformer block finishes all of its parameter gradients: four weight gradients + qkv.bias (slabs), two
 * LayerNorms' dgamma | dbeta and the bias gradient each of them carries, fc1.bias (rows). */
#define PA_REDUCE_BATCH_MAX 12
#define PA_REDUCE_SLABS 0
#define PA_REDUCE_ROWS 1
typedef struct pa_reduce_desc {
    const float* partial;
    float* out;
    int64_t n;
    int32_t splits;
    int32_t accumulate;
    int64_t pitch;         /* PA_REDUCE_ROWS: floats between two rows of `partial` (SLABS: ignored, the slices are dense) */
    int32_t mode;
    int32_t reserved;
} pa_reduce_desc;
int pa_reduce_partials_batched(const pa_reduce_desc* d, int n, void* stream);
int pa_gemm_last_colsum_rows(void);
/* out[r] = (accumulate ? out[r] : 0) + sum_c in[r][c], c < C  (bias gradients from dY^T) */
int pa_rowsum(const void* in, int dtype, int R, int C, int ld, float* out, int accumulate,
              void* stream);
/* out[c] = (accumulate ? out[c] : 0) + sum_r in[r][c] for a tall [R][C] matrix of `dtype` (bias gradients
 * straight from dY); ws: f32 workspace of pa_colsum_ws_floats(R, C) elements. */
int64_t pa_colsum_ws_floats(int R, int C);
int pa_colsum(const void* in, int dtype, int R, int C, int ld, float* out, int accumulate, float* ws,
              void* stream);
/* out[c] = (accumulate ? out[c] : 0) + sum_r in[r][c]  -- f32 in, small R (head/LN partials) */
int pa_colsum_f32(const float* in, int R, int C, int ld, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attention: Attention.forward, models/passt.py:343-361 minus the two Linears (K17-K19):
 * softmax((Q K^T) * scale) V per (batch, head), flash-style (scores never reach HBM), head_dim 64.
 * q/k/v are read in place from the qkv GEMM output [B*N][3*H*64] ([q|k|v] x head x 64, the
 * reshape of :345); o is written token-major [B*N][H*64] (the transpose+reshape of :358).
 *
 * nq (1..N): only the first nq queries of every sequence are produced -- N for a normal block; 2 for the LAST
 * block, whose output is only ever read at the cls/dist rows (models/passt.py:570-574).  o, d_o, lse and delta are
 * then COMPACT: o[(b*nq + q)][H*64], lse[(b*H + h)*nq + q].  K and V always span all N tokens.
 *
 * flags: 0, or PA_ATTN_Q_PRESCALED: the q third of qkv holds q * scale * log2(e) (the qkv GEMM wrote it so:
 * pa_gemm_args.colscale_n / colscale, one rounding).  The kernels then take "score - row reference" straight from the
 * matrix pipe (no multiply-add per score); o, lse and every gradient are the same quantities as without the flag
 * (dqkv's q third is the gradient with respect to the UNSCALED q).  The same flag must be given to fwd and bwd.
 * ------------------------------------------------------------------------------------------ */
#define PA_ATTN_Q_PRESCALED 1
/* pa_attention_bwd only (ABI 5).  bf16 + PA_ATTN_Q_PRESCALED + nq == N <= 512 can run as ONE kernel (one workgroup per
 * (sequence, head): S / dP / exp formed once, dQ contracted over all keys through an LDS transposition buffer; `delta` is then not
 * touched) instead of the dQ kernel followed by the dK/dV kernel: same quantities, both deterministic.  By default the library
 * picks the single pass when B * H >= 512 (two rounds of 256 CUs).  PA_ATTN_BWD_TWO_PASS forces the kernel pair,
 * PA_ATTN_BWD_SINGLE_PASS the single kernel wherever it applies (elsewhere it is ignored).
 * PA_ATTN_BWD_SINGLE_PASS_W16 (ABI 6): the single kernel in its sixteen-wave form (1024 threads, four waves per SIMD, one
 * 32-key block per wave) wherever the single pass applies -- same quantities; measured against the eight-wave form in
 * profiles/r06_attention_w16.txt. */
#define PA_ATTN_BWD_TWO_PASS 2
#define PA_ATTN_BWD_SINGLE_PASS 4
#define PA_ATTN_BWD_SINGLE_PASS_W16 8
int pa_attention_fwd(const void* qkv, int ldqkv, void* o, int ldo, float* lse, int B, int H, int N, int nq,
                     float scale, int dtype, int flags, void* stream);
/* number of floats of pa_attention_bwd's `delta` workspace */
int64_t pa_attention_bwd_ws_floats(int B, int H, int nq);
/* dqkv[B*N][3*H*64] from d_o[B*nq][H*64]; lse from the forward; delta: f32 workspace of pa_attention_bwd_ws_floats()
 * (per-query scalars the dQ kernel hands to the dK/dV kernel: -rowsum(dO*O), then -lse*log2 e).  The K and V thirds
 * of dqkv are written for all N tokens, the Q third only for rows q < nq (the caller zeroes the rest: pa_zero2d). */
int pa_attention_bwd(const void* qkv, int ldqkv, const void* o, const void* d_o, int ldo,
                     const float* lse, float* delta, void* dqkv, int lddqkv, int B, int H, int N, int nq,
                     float scale, int dtype, int flags, void* stream);

/* ------------------------------------------------------------------------------------------
 * Patch embedding + positional terms + Patchout: PatchEmbed.forward (models/passt.py:318-328) and
 * PaSST.forward_features :508-564 (K10-K14).  Gather-first: only the patches that survive
 * structured/unstructured Patchout are embedded.
 * ------------------------------------------------------------------------------------------ */
/* im2col of the kept patches: x[B][1][F][T] f32 -> cols[B*Np][P*P] (dtype).
 * patch_f[Np], patch_t[Np]: grid coordinates (frequency row, time column) of kept patch p. */
int pa_patch_gather(const float* x, int B, int F, int T, const int32_t* patch_f,
                    const int32_t* patch_t, int Np, int P, int fstride, int tstride, void* cols,
                    int dtype, void* stream);
/* table[p][D] = bias + time_pos[:, toff + patch_t[p]] + freq_pos[:, patch_f[p]]   (f32)
 * time_pos is [D][Tpe], freq_pos is [D][Fpe] (the reference's (1,D,1,Tpe)/(1,D,Fpe,1) params).
 * Also writes the two prefix tokens: tok[b][0] = cls + npe[0], tok[b][1] = dist + npe[1]. */
int pa_patch_pos_table(const float* bias, const float* time_pos, int Tpe, const float* freq_pos,
                       int Fpe, const int32_t* patch_f, const int32_t* patch_t, int Np, int toff,
                       int D, float* table, const float* cls, const float* dist, const float* npe,
                       float* tok, int B, int Ntok, void* stream);
/* backward of the above given dtok[B][Ntok][D] f32: gsum[Ntok][D] = sum_b dtok (ws), then
 * d_cls, d_dist, d_npe[2][D], d_bias[D], d_time_pos[D][Tpe], d_freq_pos[D][Fpe] (all overwritten
 * or accumulated), and dcols_src: the patch rows of dtok compacted to [B*Np][D] (dtype) for the
 * weight-gradient GEMM. */
int pa_patch_bwd(const float* dtok, int B, int Ntok, int D, const int32_t* patch_f,
                 const int32_t* patch_t, int Np, int toff, int Tpe, int Fpe, float* gsum,
                 float* d_cls, float* d_dist, float* d_npe, float* d_bias, float* d_time_pos,
                 float* d_freq_pos, int accumulate, void* dpatch, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Head: final norm on the two prefix tokens, their mean, head LayerNorm + Linear
 * (models/passt.py:570-574, 583-585, 463-464; K23-K24) and the BCE loss of the caller
 * (ex_audioset.py:184-186; K25).
 * ------------------------------------------------------------------------------------------ */
/* x[B][Ntok][D] f32 (last block output). feat[B][D], hn[B][D] (LN_1e-5(feat)), and saved
 * statistics stats[B][6] = (mean0, rstd0, mean1, rstd1, meanh, rstdh). */
int pa_head_pre_fwd(const float* x, int B, int Ntok, int D, const float* norm_g,
                    const float* norm_b, float eps_norm, const float* hg, const float* hb,
                    float eps_head, float* feat, float* hn, float* stats, void* stream);
/* logits[B][C] = hn[B][D] W[C][D]^T + b  (small f32 GEMM, any C) */
int pa_linear_f32_fwd(const float* x, const float* W, const float* b, float* y, int B, int C, int D,
                      void* stream);
/* dx[B][D] = dy[B][C] W[C][D]; dW[C][D] (+)= dy^T x; db[C] (+)= colsum(dy) */
int pa_linear_f32_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW,
                      float* db, int accumulate, int B, int C, int D, void* stream);
/* backward of pa_head_pre_fwd: dhn[B][D] (+ optional dfeat[B][D]) -> dx[B][Ntok][D] (rows 0,1
 * written, all other rows ZEROED), and per-batch partials part[B][4][D] =
 * (d_hg, d_hb, d_norm_g, d_norm_b) to be column-summed by the caller with pa_colsum_f32. */
int pa_head_pre_bwd(const float* dhn, const float* dfeat, const float* x, const float* feat, int B,
                    int Ntok, int D, const float* norm_g, const float* hg, const float* stats, float* dx,
                    float* part, void* stream);
/* loss[0] = mean BCE-with-logits; dlogits = grad_scale * d loss / d logits.  ws: >= 1 + ceil(B*C/256) floats */
int pa_bce_fwd_bwd(const float* logits, const float* target, int B, int C, float grad_scale,
                   float* loss, float* dlogits, float* ws, void* stream);

/* ESC-50 caller (ex_esc50.py:159-165): loss[0] = mean_b [ lam_b CE(z_b, target_b) + (1-lam_b) CE(z_b, target2_b) ];
 * target2 / lam may be NULL (plain cross entropy).  Targets are class indices.  ws: >= B floats. */
int pa_ce_mixup_fwd_bwd(const float* logits, const int32_t* target, const int32_t* target2, const float* lam,
                        int B, int C, float grad_scale, float* loss, float* dlogits, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-step glue of the caller ("next" rows)
 * ------------------------------------------------------------------------------------------ */
/* ex_audioset.py:175-177 / helpers/mixup.py: out[b] = x[b]*lam[b] + x[perm[b]]*(1-lam[b]) */
int pa_mixup(const float* x, const int32_t* perm, const float* lam, float* out, int B,
             int64_t per_sample, void* stream);
/* Waveform-side augmentation of one batch, ahead of the front end.  What the reference's data
 * pipeline does per clip on CPU workers, in its order: gain (audioset/dataset.py:102-112, amp = 10^(dB/20)),
 * pad_or_truncate to L samples (:73-78), roll (:315-329: out[t] = in[(t - shift) mod L]) and waveform mixup
 * (MixupDataset.__getitem__ :123-137: both clips mean-centred, w = max(lam, 1 - lam), out = w a + (1 - w) b).
 * The random parameters are drawn by the caller (host RNG, reference order).
 *   x: [B][ldx] f32 raw clips; len[b] valid samples (NULL: ldx); amp[b] (NULL: 1); shift[b] (NULL: 0);
 *   partner[b] >= 0: mix clip b with clip partner[b] using lam[b]; < 0: not mixed (NULL: no mixing at all).
 *   out: [B][L] f32 (the (B, 1, L) batch the training step takes); ws: f32 workspace of B floats.
 * The reference's last `x - x.mean()` of a mix is omitted: the mean of two centred signals is f32 rounding
 * noise (< 1e-7 of the amplitude). */
int pa_wave_augment(const float* x, int B, int64_t ldx, const int32_t* len, const float* amp, const int32_t* shift,
                    const int32_t* partner, const float* lam, float* ws, float* out, int64_t L, void* stream);
/* torch.optim.AdamW (ex_audioset.py:104-109) on one flat f32 parameter buffer */
int pa_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
             float beta2, float eps, float weight_decay, int step, void* stream);
/* The same update with the step's scalars in DEVICE memory: hyper = 7 floats [lr, beta1, beta2, eps, weight_decay,
 * 1 - beta1^step, sqrt(1 - beta2^step)] (pa_adamw_hyper fills a HOST array of 7 with exactly what pa_adamw computes from its
 * by-value arguments; the caller copies it to the device).  The launch's arguments are then the same every step, so it can be
 * part of a captured hipGraph (passt_amd.train.TrainStep(graph=True)). */
int pa_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, void* stream);
void pa_adamw_hyper(float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* hyper7_host);
/* AdamW of one bucket of the flat buffers AND the GEMM-ready copies of its weight matrices in ONE launch (ABI 6): what
 * pa_adamw followed by pa_stage_weights produce -- parameters, moments and copies bit-identical -- without reading the updated
 * parameters back (torch.optim.AdamW, ex_audioset.py:104-109; the copies have no reference counterpart: AMP autocast casts the
 * weights per op).  `descs` (DEVICE memory, n_desc entries, in work-item order) lists every parameter of the bucket:
 *   offset      element offset of the parameter in p / g / m / v
 *   rows, cols  its shape as a row-major matrix (a parameter without copies: rows = 1, cols = numel)
 *   dst, dst_t  where non-NULL: the straight cast [rows][cols] and the transposed cast [cols][rows] of `dtype`, contiguous
 *   tile_begin  running count of work items: ceil(rows/64)*ceil(cols/64) tiles of 64x64 for a parameter with a copy,
 *               ceil(numel/4096) runs for one without;  total_items = their sum (= the grid).
 * hyper_dev == NULL: the step's scalars by value, as pa_adamw;  else the 7 device floats of pa_adamw_dev (by-value ones ignored). */
typedef struct pa_adamw_stage_desc {
    int64_t offset;
    void* dst;
    void* dst_t;
    int32_t rows, cols;
    int32_t tile_begin;
    int32_t reserved;
} pa_adamw_stage_desc;
int pa_adamw_stage(float* p, const float* g, float* m, float* v, const pa_adamw_stage_desc* descs, int n_desc, int total_items,
                   int dtype, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   const float* hyper_dev, void* stream);
/* torch.optim.SGD lr only (ex_audioset.py:392 model_speed_test) */
int pa_sgd(float* p, const float* g, int64_t n, float lr, void* stream);
/* Stochastic weight averaging step on flat buffers (helpers/swa_callback.py:246-268, update_parameters + avg_fn):
 * num_averaged == 0: avg = p;  else avg += (p - avg) / (num_averaged + 1).  The caller increments the count. */
int pa_swa_update(float* avg, const float* p, int64_t n, int num_averaged, void* stream);

/* ------------------------------------------------------------------------------------------
 * Data-parallel exchange step: gradient all-reduce over RCCL / xGMI, one communicator per process (= per GPU).
 * Replaces Lightning's DDP plugin (ex_audioset.py:488-489 -> torch DDP: NCCL all-reduce of gradient buckets from
 * autograd hooks; mean of the per-rank gradients).  Here the caller owns the buckets: contiguous slices of its flat
 * gradient buffer, reduced IN PLACE (sum; the 1/world factor is folded into the loss gradient), launched as soon as a
 * block's gradients are complete (passt_amd.ddp.GradReducer drives it from the backward's callbacks).
 * RCCL is loaded at run time (dlopen); a process that never calls these functions does not need it.
 * ------------------------------------------------------------------------------------------ */
#define PA_COMM_ID_BYTES 128
/* rank 0: 128 opaque bytes (an ncclUniqueId) to hand to every rank out of band (file, socket, torch store ...) */
int pa_comm_unique_id(void* id_out);
/* every rank, after selecting its device (hipSetDevice / torch.cuda.set_device): blocks until all `world` ranks joined */
int pa_comm_init(const void* id, int rank, int world, void** comm_out);
/* buf[count] (dtype PA_F32 or PA_BF16) <- sum over ranks, in place, asynchronous and ordered on `stream` */
int pa_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, void* stream);
int pa_comm_destroy(void* comm);
/* What the communicator itself reports (ncclGetVersion / ncclCommCount / ncclCommUserRank / ncclCommCuDevice): the evidence
 * bench.py prints for an N > 1 run.  Any out pointer may be NULL; comm == NULL with only rccl_version asked returns the
 * version of the RCCL the library resolved (e.g. 22703 = 2.27.3). */
int pa_comm_info(void* comm, int* rccl_version, int* nranks, int* rank, int* device);
/* text of the last PA_ECOMM on the calling thread */
const char* pa_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PASST_AMD_H */
