/* sdumc_hip.h — C ABI of libsdumc_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the ONE hot path of WarmCongee/SDUMC: the
 * WengnetMOSEIMultViewsTextMissing forward/backward, its losses and the Adam
 * step.  The reference has no native code; what its Python calls
 * on this path are torch eager ops.  Each entry point below names the reference
 * code it replaces (file:line relative to the reference repo).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise;
 *     tensors are dense row-major with the strides given;
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*)
 *     and returns immediately: no allocation, no synchronisation, no host
 *     copies -> every call is hipGraph-capturable;
 *   - `stream` must belong to the CURRENT device (hipSetDevice) of the calling thread, like any HIP launch; the
 *     network-level calls keep their internal streams per device (or per caller context, sdumc_net_io.ctx), so calls
 *     on different devices or with different contexts share no state and may run from different host threads;
 *   - return value: 0 = ok, <0 = SDUMC_E* (never throws across the ABI);
 *   - scratch memory is caller-owned; *_workspace_bytes() says how much.
 */
#ifndef SDUMC_HIP_H
#define SDUMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDUMC_OK 0
#define SDUMC_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define SDUMC_ELAUNCH (-2)  /* hipLaunch / HIP runtime error */
#define SDUMC_ENOMEM (-3)   /* workspace too small */

#define SDUMC_MAX_GROUPS 8
#define SDUMC_D 256         /* general_dim            (model :191) */
#define SDUMC_H 128         /* layers_list[-1]        (model :187) */
#define SDUMC_NQ 7          /* multi-view queries     (model :332) */
#define SDUMC_RNC_DIM 64    /* orgin_linear_change out (model :246-250) */
#define SDUMC_N_SITES 35    /* nn.Dropout call sites of one forward */

/* ------------------------------------------------------------------------
 * Dropout descriptor.  Replaces nn.Dropout / aten::bernoulli_ (model :54,:77,
 * :270).  Masks are a pure function of (seed, call, site, sample, row, col):
 * Philox4x32-10, ctr = (row*(width/4)+col/4, sample0+sample, site, call+stream),
 * key = seed; element dropped when word[col%4] < threshold; kept elements are
 * multiplied by `scale`.  The row space of the tensor the descriptor is attached
 * to is [streams][samples][rows][width].
 * ---------------------------------------------------------------------- */
typedef struct sdumc_dropout {
  uint32_t enabled;    /* 0 = identity (eval mode) */
  uint32_t site;       /* 0..SDUMC_N_SITES-1 */
  uint32_t threshold;  /* floor(p * 2^32) */
  float scale;         /* 1/(1-p) */
  uint32_t rows;       /* rows per sample (T, NQ or 1) */
  uint32_t width;      /* row width; a multiple of 4 wherever the mask is fused into a GEMM */
  uint32_t samples;    /* samples per stream (local batch B) */
  uint32_t sample0;    /* global index of local sample 0 (data-parallel shard offset) */
  uint32_t call0;      /* Philox call index of stream 0 (ignored when dev_state != NULL) */
  uint32_t stream0;    /* added to the stream index (a tensor that holds only stream 1 passes 1) */
  uint32_t seed_lo, seed_hi;
  const uint32_t* dev_state; /* optional device {seed_lo, seed_hi, call0}: lets a captured
                                hipGraph draw fresh masks on every replay */
  const uint8_t* bits;       /* optional keep-bits precomputed by sdumc_dropout_bits for exactly this row space:
                                byte [row * width/4 + q], bit e = keep column 4q+e.  Kernels that stage the same
                                tile several times (4 n-tiles of a GEMM, forward + backward) then read one byte
                                instead of re-running Philox (~90 VALU ops per 4 elements) */
} sdumc_dropout;

/* ------------------------------------------------------------------------
 * fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32), grouped, with the fusions this path
 * needs.  Replaces F.linear / torch.bmm / their autograd (aten::addmm, mm).
 *   layout SDUMC_NT: C[M,N] = A[M,K] . B[N,K]^T   (forward Linear: x.W^T)
 *   layout SDUMC_NN: C[M,N] = A[M,K] . B[K,N]     (dX = dY.W)
 *   layout SDUMC_TN: C[M,N] = A[K,M]^T . B[K,N]   (dW = dY^T.X)
 * ---------------------------------------------------------------------- */
enum { SDUMC_NT = 0, SDUMC_NN = 1, SDUMC_TN = 2 };
enum { SDUMC_ACT_NONE = 0, SDUMC_ACT_RELU = 1, SDUMC_ACT_TANH = 2 };

typedef struct sdumc_gemm {
  int32_t layout, M, N, K;
  int32_t groups;                        /* 1..SDUMC_MAX_GROUPS independent problems of one shape */
  const float* A[SDUMC_MAX_GROUPS];
  const float* B[SDUMC_MAX_GROUPS];
  float* C[SDUMC_MAX_GROUPS];
  const float* bias[SDUMC_MAX_GROUPS];   /* [N] or NULL */
  int32_t lda, ldb, ldc;
  int32_t a_row_mod;   /* >0: A's source row = row % a_row_mod (NT/NN rows = m; TN rows = k).
                          Lets the two streams share one x_audio / x_video in HBM. */
  int32_t b_row_mod;   /* TN only: B's source row = k % b_row_mod */
  sdumc_dropout a_drop; /* NT/NN: fused dropout on A as it is staged (rows = m, cols = k) */
  sdumc_dropout b_drop; /* TN: fused dropout on B as it is staged (rows = k, cols = n) */
  int32_t act;          /* epilogue activation */
  sdumc_dropout c_drop; /* epilogue dropout on C (rows = m, cols = n) */
  int32_t c_drop_group_stride; /* group g uses site c_drop.site + g * stride (grouped MLPs) */
  int32_t accumulate;   /* C += result (requires act none, no c_drop, no bias) */
  int32_t splitk;       /* 0: auto (tile + split chosen to fill 256 CUs); 1: no split; >1: K split over workgroups.
                           Split results go through fp32 slabs in `workspace` and an ordered, deterministic reduce */
  int32_t tile;         /* 0 auto, 1 = 128x128, 2 = 64x64, 4 = 128x64, 3 = small-problem kernel (32x32 tile, the four waves split K;
                           for the launch-bound utterance-level layers; no operand-side fusions) */
  int32_t ab_drop_group_stride; /* group g draws a_drop / b_drop from site + g * stride ... */
  const uint8_t* ab_drop_bits[SDUMC_MAX_GROUPS]; /* ... or from these per-group keep-bits (NULL: a_drop.bits / b_drop.bits) */
  const float* c_mask_y[SDUMC_MAX_GROUPS]; /* optional epilogue (selects the small-problem kernel): C = C * [Y > 0] * c_mask_scale,
                           Y laid out like C, applied after `accumulate`.  Folds the backward of the NEXT
                           Linear->ReLU->Dropout (dz = dy * [y > 0] / (1-p), model :264-273) into the dX GEMM */
  float c_mask_scale;
  float* colsum_a[SDUMC_MAX_GROUPS]; /* TN only, optional: out[m] (+)= sum_k A[k,m] fused into the staging of A
                           (the bias gradient when A = dz); `accumulate` applies to it too */
  int32_t bf16;         /* 1: operands rounded to bf16 while staged (row-contiguous ones transposed by their LDS stores),
                           v_mfma_f32_32x32x16_bf16, fp32 accumulate and fp32 epilogue / column sums (the "bf16 compute" mode,
                           BASELINE configs[2]; 64x64 and 128x128 tiles, channel extents multiples of 4); 0: exact fp32 (default) */
  float* workspace;
  size_t workspace_bytes;
  int32_t batch;        /* >1: strided-batched mode -- every group is `batch` products of one shape, entry z reads
                           A + z*stride_a, B + z*stride_b and writes C + z*stride_c (the per-(sample, head) QK^T / PV
                           products of sdumc_mha_*, multihead_attention.py:103,123: with [T, B, H*d_h] activations the
                           head slice of (b, h) starts at (b*H + h)*d_h, so one stride walks samples and heads).
                           Plain products only: no operand/epilogue dropout, bias, column sums or split-K */
  int64_t stride_a, stride_b, stride_c;   /* floats */
} sdumc_gemm;

size_t sdumc_gemm_workspace_bytes(const sdumc_gemm* g);
int sdumc_gemm_f32(const sdumc_gemm* g, void* stream);

/* ------------------------------------------------------------------------
 * GEMM on bf16 STORAGE (BASELINE configs[2], configs[4]): A and B are bf16 in HBM, products accumulate in fp32 on
 * v_mfma_f32_32x32x16_bf16, C is fp32 or bf16.  The frame-level products of the step in bf16 mode
 * (sdumc_net_dims.bf16 = 2): frame_dim_reshape (model :282-284), input_proj of FRA2UTT_new / Cross_Attention
 * (model :60, :82) and their autograd (main :149).
 *   SDUMC_NT: C[M,N] = A[M,K] . B[N,K]^T (+ bias, tanh / ReLU; accumulate: C += ...);  K a multiple of 64
 *   SDUMC_TN: C[M,N] = A[K,M]^T . B[K,N]  (fp32 C; split-K through fp32 slabs, ordered reduce; colsum_a = column sums of A)
 * Leading dimensions and the contiguous extents are multiples of 8 elements (16-byte rows), pointers 16-byte aligned.
 * No dropout descriptor: in bf16 mode the masked frames xd = drop(x) are materialised once (sdumc_mask_apply_bf16).
 * ---------------------------------------------------------------------- */
typedef struct sdumc_gemm_bf16 {
  int32_t layout, M, N, K, groups;
  const void* A[SDUMC_MAX_GROUPS];      /* bf16 */
  const void* B[SDUMC_MAX_GROUPS];      /* bf16 */
  void* C[SDUMC_MAX_GROUPS];            /* fp32, or bf16 when c_bf16 */
  const float* bias[SDUMC_MAX_GROUPS];  /* [N] fp32 or NULL */
  float* colsum_a[SDUMC_MAX_GROUPS];    /* TN only, optional: out[m] (+)= sum_k A[k, m] */
  int32_t lda, ldb, ldc;
  int32_t a_row_mod;                    /* NT: A's source row = m % a_row_mod (both streams share x in eval mode) */
  int32_t b_row_mod;                    /* TN: B's source row = k % b_row_mod */
  int32_t act, accumulate, c_bf16;
  int32_t splitk;                       /* TN: 0 auto, >= 1 explicit */
  float* workspace;
  size_t workspace_bytes;
} sdumc_gemm_bf16;
size_t sdumc_gemm_bf16_workspace_bytes(const sdumc_gemm_bf16* g);
int sdumc_gemm_bf16_run(const sdumc_gemm_bf16* g, void* stream);

/* ------------------------------------------------------------------------
 * Grouped weight-gradient GEMM: every dW = dz^T . x of a backward phase in ONE persistent launch (+ one ordered reduce).
 * Replaces the autograd products of the Linear layers (aten::mm of grad_output^T with the saved input, plus the bias
 * gradient's sum over rows) of model :282-284 (frame_dim_reshape), :60 / :82 (input_proj of FRA2UTT_new / Cross_Attention)
 * and :293-368 (utterance-level MLPs) when main :149 calls loss.backward().
 *   problem i:  C_i[M,N] (+)= sum over K-segments s of  A_is[K_s, M]^T . mask_is(B_is[K_s, N]),   colsum_i[m] (+)= sum_k A[k, m]
 * Both operands are row-contiguous in the contraction index (dz [rows, out], x [rows, in]).  The launch is a stream-K
 * decomposition: the k-tiles of all output tiles of all problems form one line that the resident workgroups (one per CU)
 * cut into equal contiguous ranges, so every CU multiplies for the same time whatever the shapes; a tile that ends up in
 * several ranges goes through fp32 partial slabs in `workspace` and the reduce launch sums them in ascending k order
 * (bit-identical from run to run).  Constraints: M, N, lda, ldb multiples of 4; pointers 16-byte aligned; every operand
 * below 4 GiB; M <= 256 * 255.
 * ---------------------------------------------------------------------- */
#define SDUMC_GG_MAX_PROBLEMS 40   /* per launch pair; longer lists are cut into several */
typedef struct sdumc_gg_problem {
  const float* A[2];        /* [K_s, M] with lda: the gradient w.r.t. the layer's pre-activation (segment 1 unused when K[1] == 0) */
  const float* B[2];        /* [K_s, N] with ldb: the layer's saved input */
  const uint8_t* b_bits[2]; /* optional keep-bits of a dropout fused on B (byte [row * bits_qw + col/4], bit col%4; rows = k) */
  int32_t K[2];             /* rows per segment: two segments = the two streams' text frames, which live in separate tensors */
  int32_t b_row_mod[2];     /* >0: B's source row = k % b_row_mod (the streams share x_audio / x_video) */
  float* C;                 /* [M, N] with ldc */
  float* colsum_a;          /* [M] or NULL: the bias gradient */
  int32_t M, N, lda, ldb, ldc;
  int32_t bits_qw;          /* bytes per row of b_bits */
  float b_scale;            /* 1 / (1 - p) of the fused dropout */
  int32_t accumulate;       /* C += ..., colsum_a += ... */
  /* Optional row maps (fp32 operands with the products on the bf16 matrix pipe, i.e. the default arithmetic; else SDUMC_EINVAL):
   * B[s] is then the packed tensor of a RESIDENT feature store, of any size, and k-row r of segment s is ITS row b_map[s][r] (int32,
   * device) -- the frame projections' weight gradient reads the batch in place.  No b_row_mod with it.  The bf16 form
   * (sdumc_gemm_group_tn_bf16) takes maps too; it fetches entries four at a time: b_map[s] must be readable up to 4 * ceil(K[s] / 4)
   * entries. */
  const int32_t* b_map[2];
} sdumc_gg_problem;
size_t sdumc_gemm_group_workspace_bytes(const sdumc_gg_problem* probs, int32_t n);
int sdumc_gemm_group_tn(const sdumc_gg_problem* probs, int32_t n, void* workspace, size_t workspace_bytes, void* stream);
/* The same on bf16 STORAGE (sdumc_net_dims.bf16 = 2; BASELINE configs[2] / [4]): A and B point to bf16 tensors (lda / ldb in
 * elements; M, N, lda, ldb multiples of 8; b_row_mod 0 or >= 64; no fused dropout -- b_bits must be NULL, the engine materialises
 * the masked frames), products accumulate in fp32 on v_mfma_f32_32x32x16_bf16, C / colsum_a / the slabs are fp32. */
size_t sdumc_gemm_group_bf16_workspace_bytes(const sdumc_gg_problem* probs, int32_t n);
int sdumc_gemm_group_tn_bf16(const sdumc_gg_problem* probs, int32_t n, void* workspace, size_t workspace_bytes, void* stream);
/* fp32 products on the bf16 matrix pipe.  v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate; an fp32 value is the exact
 * sum of three bf16 values, so the fp32 GEMM kernels below split both operands in registers (round to nearest, exact residuals)
 * and accumulate the six largest of the nine bf16 x bf16 products -- each exact in fp32 -- with six v_mfma_f32_32x32x16_bf16 per
 * 32 x 32 x 16 block instead of eight fp32 MFMAs.  The three dropped products are below 2^-23 |a b| together, the size of ONE fp32
 * rounding: measured against fp64 the results are as close as the fp32-MFMA kernels' (tools/gg_split_check.py,
 * tools/wide_split_check.py; tests/test_gpu_split.py).  Inputs, outputs and accumulators stay fp32.
 * Measurement / test hook, process-wide: bit 0 the grouped weight-gradient launch (sdumc_gemm_group_tn), bit 1 the wide-tile NT
 * launches of the GEMM entry point (frame / key projections), bit 2 the key projection inside sdumc_umca_fwd, bit 3
 * sdumc_gemm_rows256.
 * Default all (environment SDUMC_SPLIT); 0 = every product on the fp32 MFMAs.
 *
 * CONTRACT AT THE EDGES of the split kernels (tests/test_split_arithmetic.py restates it on the CPU, tests/test_gpu_split.py holds
 * each kernel family to it; x = any operand element, fp32 MFMA = what sdumc_set_split_(0) returns):
 *   - finite x with |x| < 0x1.FEp127 (3.3895e38): as documented above.  -0 behaves as +0 in a product sum (as in fp32).
 *   - NaN operand -> NaN in every output element its row / column feeds (same as the fp32 MFMAs).
 *   - +-Inf operand -> NaN there (the fp32 MFMAs keep +-Inf unless 0 * Inf or Inf - Inf occurs): the first part is Inf, the residual
 *     Inf - Inf is NaN.  A step that meets an infinity is lost either way (NaN loss; Adam then spreads it).
 *   - finite |x| >= 0x1.FEp127 (within half a bf16 ulp of FLT_MAX: 3.3895e38 .. 3.4028e38) rounds to Inf in its first part and is
 *     treated like Inf: NaN.  The fp32 MFMAs overflow for such values as soon as |x * b| >= 2^128.
 *   - tiny values: bf16 has fp32's exponent range but its subnormals stop at 2^-133, and the matrix pipe may flush subnormal
 *     inputs.  |x| >= 2^-102: every non-zero part is a normal bf16 -- exact as above.  Below: a part under 2^-126 may be lost, the
 *     absolute error of such an x is < 2^-125 (of a product: < 2^-125 |b|) -- nothing on this path (features ~N(0,1), weights
 *     ~1e-2, gradients >= 1e-20) comes near it.
 * No clamping is done: the kernels do not spend VALU work on values a training step cannot survive anyway.
 *
 * The switch is an option of the execution context -- sdumc_ctx_set_option(ctx, SDUMC_OPT_SPLIT, mask): the network-level calls made
 * with that context then take it (two contexts can hold different arithmetic, nothing process-wide changes).
 * sdumc_set_split_ sets the PROCESS DEFAULT used by contexts without the option and by the stand-alone kernel entry points (deprecated
 * as a control for network-level calls; kept for kernel-level tests and benches); sdumc_get_split_ returns the mask in force for the
 * calling thread (the default, or the context's inside a network-level call). */
void sdumc_set_split_(int mask);
int sdumc_get_split_(void);

/* ------------------------------------------------------------------------
 * fp32 GEMM on operands split ONCE PER TENSOR into three bf16 planes (csrc/gemm_p3.hip).  Replaces F.linear of
 * frame_dim_reshape_{0,1,2} (model :193-195, :282-284) and of input_proj in FRA2UTT_new / Cross_Attention (model :60, :82) in
 * the fp32 step: same arithmetic as the split kernels above (six bf16 x bf16 part products per fp32 product, fp32 accumulation,
 * same edge contract), but the split is paid where a tensor is PRODUCED -- sdumc_p3_split for the features (once per batch; they
 * do not change across epochs) and the weights (once per step), the epilogue of this GEMM for the projected frames -- instead of
 * per workgroup per k-tile.
 * P3 layout of a row-major X[rows][K], K % 8 == 0: row r = 6 K bytes at r * ld_bytes; chunk c (48 bytes) holds k = 8 c .. 8 c + 7
 * as [plane 0: 8 bf16][plane 1][plane 2]; p0 = bf16(x) (round to nearest even), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1);
 * x == (p0 + p1) + p2 exactly for finite |x| < 0x1.FEp127 (edges: see sdumc_set_split_).  1.5x the bytes of the fp32 tensor.
 * The weight operand of the NT product is stored FRAGMENT-MAJOR (sdumc_p3_split_frag): [N / 32][K / 16][3 planes][64 lanes][16 bytes],
 * lane (li = lane & 31, lh = lane >> 5) of block (rb, kt) holds W[32 rb + li][16 kt + 8 lh .. + 7] -- a wavefront's MFMA operand of a
 * k-tile is three contiguous KiB, loaded straight into registers (the weight never passes through LDS: only the wave that owns the
 * 32 output columns multiplies them).
 *   NT:  C[M, N] = act((A . keep)[M, K] . B[N, K]^T * a_scale + bias),  N % 256 == 0, K % 64 == 0
 * ---------------------------------------------------------------------- */
typedef struct sdumc_gemm_p3 {
  int32_t M, N, K;
  const void* A;            /* P3 [M or a_row_mod rows][K], row stride lda BYTES (>= 6 K, multiple of 16; operand below 4 GiB) */
  const void* B;            /* fragment-major P3 of the weight [N][K] (sdumc_p3_split_frag) */
  int64_t lda, ldb;         /* ldb: bytes between two 32-row blocks of B (>= K / 16 * 3072) */
  int32_t a_row_mod;        /* >0: A's source row = m % a_row_mod (the two streams share x_audio / x_video) */
  const void* A2;           /* optional second A tensor: rows [a2_row0, M) are its rows 0 .. (the two streams' text features live in */
  int32_t a2_row0;          /* separate tensors, their projections in adjacent rows of one C); a multiple of 64; no a_row_mod */
  const uint8_t* a_bits;    /* optional keep-bits of a dropout fused on A: byte [m * bits_qw + k/4], bit k%4 (rows = VIRTUAL rows m) */
  int32_t bits_qw;          /* bytes per row of a_bits (>= K/4, multiple of 4) */
  float a_scale;            /* 1 / (1 - p); read only with a_bits */
  const float* bias;        /* [N] or NULL */
  int32_t act;              /* SDUMC_ACT_* */
  float* C;                 /* fp32 [M][N], row stride ldc floats; may be NULL when C_p3 is given */
  int32_t ldc;
  void* C_p3;               /* optional: the same values as a P3 tensor, row stride ldc_p3 bytes (>= 6 N) */
  int64_t ldc_p3;
  int32_t splitk;           /* 0 auto, 1 none, > 1: K split over workgroups (fp32 slabs in workspace + an ordered reduce) */
  int32_t tile_m;           /* 0 = 64-row tiles on 256-thread workgroups, two per CU (the form the step uses: it shares a CU with other lanes'
                               kernels); 64 / 96 / 128 = that many rows per 512-thread workgroup (one per CU: fastest alone on shapes that fill
                               the chip in one round) */
  float* workspace;         /* >= sdumc_gemm_p3_workspace_bytes */
  size_t workspace_bytes;
  /* Optional row maps: A (and A2) are then the packed P3 tensors of a RESIDENT feature store, of any size, and virtual row m reads
   * A's row a_map[m] (m < a2_row0 or no A2) / A2's row a2_map[m - a2_row0]: the batch is read in place (a padded frame's entry names
   * an all-zero row the store keeps).  Rows are fetched by 64-bit address (global_load_lds), so the 4 GiB operand limit does not
   * apply.  Without a_bits / a_row_mod / tile_m; a2_map if and only if A2.  a_map_rows (optional): rows of the packed tensor(s) (the
   * larger of the two): when a_map_rows * lda stays below 4 GiB the kernel keeps its descriptor addressing (faster than 64-bit
   * addresses); 0 = unknown. */
  const int32_t* a_map;
  const int32_t* a2_map;
  int64_t a_map_rows;
} sdumc_gemm_p3;
size_t sdumc_gemm_p3_workspace_bytes(const sdumc_gemm_p3* g);
int sdumc_gemm_p3_nt(const sdumc_gemm_p3* g, void* stream);
/* fp32 [rows][cols] (row stride ld floats, cols % 8 == 0, 16-byte aligned) <-> P3 (row stride ld_bytes); join(split(x)) == x bit for
 * bit for every finite non-zero x below 0x1.FEp127 in magnitude; -0 comes back as +0 */
int sdumc_p3_split(const float* src, int64_t ld, void* dst, int64_t ld_bytes, int64_t rows, int32_t cols, void* stream);
/* fp32 weight [rows][cols] (rows % 32 == 0, cols % 16 == 0) -> fragment-major P3, 6 rows cols bytes, blocks of 32 rows contiguous */
int sdumc_p3_split_frag(const float* src, int64_t ld, void* dst, int32_t rows, int32_t cols, void* stream);
int sdumc_p3_join(const void* src, int64_t ld_bytes, float* dst, int64_t ld, int64_t rows, int32_t cols, void* stream);

/* ------------------------------------------------------------------------
 * The same NT product for the bf16-STORAGE step (sdumc_net_dims.bf16 = 2; csrc/gemm_b1.hip): A and the weight are bf16, fp32
 * accumulation, C bf16 or fp32.  The weight is stored fragment-major -- [N / 32][K / 16][64 lanes][16 bytes], lane (li, lh) of block
 * (rb, kt) holds W[32 rb + li][16 kt + 8 lh .. + 7] (sdumc_b1_frag_multi writes it from the fp32 parameters, round to nearest even)
 * -- and goes straight into registers; only A passes through LDS.
 *   C[M, N] = act(A[M, K] . B[N, K]^T + bias),  N % 256 == 0, K % 128 == 0
 * ---------------------------------------------------------------------- */
typedef struct sdumc_gemm_b1 {
  int32_t M, N, K;
  const void* A;            /* bf16 [M or a_row_mod rows][K], row stride lda ELEMENTS (>= K, multiple of 8; operand below 4 GiB) */
  const void* B;            /* fragment-major bf16 of the weight [N][K] */
  int64_t lda, ldb;         /* ldb: BYTES between two 32-row blocks of B (>= K / 16 * 1024) */
  int32_t a_row_mod;        /* >0: A's source row = m % a_row_mod */
  const void* A2;           /* optional second A tensor holding rows [a2_row0, M) (a2_row0 % 64 == 0; no a_row_mod) */
  int32_t a2_row0;
  const float* bias;        /* [N] fp32 or NULL */
  int32_t act;              /* SDUMC_ACT_* */
  void* C;                  /* [M][N], row stride ldc elements (multiple of 8) */
  int32_t ldc;
  int32_t c_bf16;           /* 1: C is bf16 (round to nearest even), 0: fp32 */
  int32_t splitk;           /* 0 auto, 1 none, > 1: K split over workgroups (fp32 slabs in workspace + an ordered reduce) */
  float* workspace;         /* >= sdumc_gemm_b1_workspace_bytes */
  size_t workspace_bytes;
  /* Optional row maps, as in sdumc_gemm_p3: A (and A2) are the packed bf16 tensors of a resident feature store and virtual row m reads
   * row a_map[m] (a2_map[m - a2_row0]); a_map_rows = rows of the packed tensor(s), 0 = unknown (64-bit addressing). */
  const int32_t* a_map;
  const int32_t* a2_map;
  int64_t a_map_rows;
} sdumc_gemm_b1;
size_t sdumc_gemm_b1_workspace_bytes(const sdumc_gemm_b1* g);
int sdumc_gemm_b1_nt(const sdumc_gemm_b1* g, void* stream);
/* n <= 12 fp32 weights [rows_i][cols_i] at P + src_off[i] floats (contiguous rows; rows % 32 == 0, cols % 16 == 0) -> fragment-major
 * bf16 at dst + dst_off[i] bytes (2 rows cols bytes each), one launch */
int sdumc_b1_frag_multi(const float* P, void* dst, const int64_t* src_off, const int64_t* dst_off, const int32_t* rows,
                        const int32_t* cols, int n, void* stream);

/* ------------------------------------------------------------------------
 * The tall 256 x 256 products of the frame-level part in one persistent launch (gemm_rows.hip):
 *   C[M x 256] = act((A . keep) B * a_scale + bias) (+ C),   K = N = 256
 * - the key projections  keys = tanh(drop(x) W^T + b)  of FRA2UTT_new / Cross_Attention (model :60, :82; main :144 in train
 *   mode): A = the projected frames, a_bits = the keep-bits of the fused input dropout, B = W^T (a [in][out] copy);
 * - their input gradients dxd += dz W (autograd, main :149): A = dz, B = the weight as stored ([out][in]), accumulate = 1.
 * A is [M][256] row-major with row stride lda (floats, 16-byte aligned rows); row r reads source row r % a_row_mod when
 * a_row_mod > 0 (the two streams share the audio / video frames).  B is [256 k][256 n] with row stride ldb.  a_bits (NULL = no
 * dropout) holds one byte per 4 columns and row of A's VIRTUAL rows ([M][64], low nibble, as sdumc_dropout_bits writes it).
 * Every problem of one call takes the same kernel variant: all masked, all accumulating, or neither (else SDUMC_EINVAL).
 * ------------------------------------------------------------------------ */
#define SDUMC_ROWS_MAX_PROBLEMS 8
typedef struct sdumc_rows_problem {
  const float* A;
  const float* B;
  const uint8_t* a_bits;
  const float* bias;        /* [256] or NULL */
  float* C;                 /* [M][256], row stride ldc */
  int32_t M, lda, ldb, ldc;
  int32_t a_row_mod;
  float a_scale;            /* 1 / (1 - p); read only with a_bits */
  int32_t accumulate;
  int32_t act;              /* SDUMC_ACT_NONE or SDUMC_ACT_TANH */
  /* optional rank-nq term of the attention pooling's own input gradient, folded into the product as one more k-tile:
   *   C[r] = A[r] B + sum_i pool_w[r][i] * pool_g[r / pool_T][i][:]
   * pool_w = the attention weights [M][pool_nq] of the site, pool_g = dout * out-dropout mask [M / pool_T][pool_nq][256]
   * (sdumc_attnpool_bwd.dout_masked).  dxd = dz W + (pooling part) then leaves this kernel in ONE pass: the pooling backward
   * does not write dxd and this launch does not read it back.  Needs pool_nq <= 8 (<= 7 with fold), pool_T >= 63 or pool_T == 32 (a
   * 64-row tile spans at most two samples), no a_bits, no bias / act / a_row_mod, M % pool_T == 0; accumulate only together with fold. */
  const float* pool_w;
  const float* pool_g;
  int32_t pool_nq, pool_T;
  /* optional, with the pooling term (fp32: split arithmetic only; sdumc_gemm_rows256_bf16 takes it too -- pool_w / pool_g stay fp32,
   * every problem of a launch the same fold): the mask-sum of the frame-level input dropouts folded in as well --
   *   C[r] (+)= sum_{s < fold} keep_s[r] . (A[s R + r] B + pooling term of row s R + r) * c_scale,   R = M / fold rows of C
   * i.e. dx of a modality = the sum over the streams that share its frames (fold = 2; 1 for separate frames) of the masked dxd of one
   * attention site, written ONCE: dxd never exists in memory.  c_bits = the keep-bits of the site's input dropout over the M virtual
   * rows ([M][64], as a_bits; NULL = keep everything, c_scale unused); accumulate = 1 adds onto C (the second site of the modality).
   * fold = 0: off.  R % pool_T == 0. */
  const uint8_t* c_bits;
  float c_scale;
  int32_t fold;
} sdumc_rows_problem;
int sdumc_gemm_rows256(const sdumc_rows_problem* probs, int32_t n, void* stream);
/* The same on bf16 STORAGE (sdumc_net_dims.bf16 = 2): A ([M][256]), B and C are bf16 tensors (lda / ldb / ldc in elements, lda
 * and ldb multiples of 8, A and B 16-byte aligned); B is [256 n][256 k] -- the rows of B are the OUTPUT columns, C = A B^T: the
 * bf16 weight copy as stored for the key projections, its transposed copy for dxd += dz W; bias fp32; a_bits must be NULL (the
 * engine materialises the masked frames); products accumulate in fp32 on v_mfma_f32_32x32x16_bf16. */
int sdumc_gemm_rows256_bf16(const sdumc_rows_problem* probs, int32_t n, void* stream);

/* ------------------------------------------------------------------------
 * Attention pooling over the time axis = the body shared by
 *   FRA2UTT_new.forward     (model :56-68; nq = 1, query = attention_context_vector, q_stride 0)
 *   Cross_Attention.forward (model :79-95; nq = 7, query = query_proj(multi_query))
 * after the key projection K = tanh(drop(x).W^T + b) has been produced by
 * sdumc_gemm_f32 (NT, a_drop, act tanh).
 *   S[v,t,i] = K[v,t,:].Q[v,i,:] ; A = softmax_t(0.3 S) ; O[v,i,:] = sum_t A[v,t,i] drop(x)[v,t,:]
 *   out = drop_out(O)
 * v = virtual sample = stream*B + b; x row of v is (v % x_samples).
 * ---------------------------------------------------------------------- */
typedef struct sdumc_attnpool {
  int32_t V;           /* virtual samples (streams * B) */
  int32_t T;           /* frames */
  int32_t nq;          /* 1..8 */
  int32_t x_samples;   /* x holds this many samples; x row of v = v % x_samples */
  const float* x;      /* [x_samples, T, 256] */
  const float* keys;   /* [V, T, 256] tanh keys */
  const float* q;      /* [V or 1, nq, 256] projected queries */
  int64_t q_stride;    /* floats between the queries of consecutive v (0 = shared) */
  float scale;         /* 0.3 */
  sdumc_dropout x_drop;   /* the input dropout (same descriptor the key GEMM used) */
  sdumc_dropout out_drop; /* the output dropout, rows = nq */
  float* attn;         /* [V, T, nq]  softmax weights (returned by the reference, saved for backward) */
  float* pooled;       /* [V, nq, 256] O before the output dropout (saved for backward) */
  float* out;          /* [V, nq, 256] */
  float* workspace;    /* >= sdumc_attnpool_fwd_workspace_bytes(V, T, nq): per-chunk softmax partials */
  size_t workspace_bytes;
  int32_t dim;         /* channels per row: 0 or 256 (the model's general_dim, model :191), or 512 / 768 / 1024 for the
                          blocks used on their own (FRA2UTT_new / Cross_Attention default to input_dim = 1024,
                          model :47,:71).  Every "256" in the shapes above and below then reads `dim`; the dropout
                          descriptors carry width = dim; workspaces are sized by the *_dim queries */
  const int32_t* lengths; /* EXTENSION (default NULL = the reference's behaviour, where zero-padded frames take
                          part in the softmax, read_data.py:139-151 / model :63,:90): device int32 [V], valid frames per
                          virtual sample; frames t >= max(1, lengths[v]) get weight exactly 0 (key-padding mask) and
                          therefore zero gradient in sdumc_attnpool_bwd, which needs no mask of its own */
  int32_t bf16;        /* 1: x and keys (and, in sdumc_attnpool_bwd, dz and dxd) are bf16 tensors of the same shapes (the engine's
                          bf16-storage mode, BASELINE configs[2] / [4]); scores, softmax, pooling, attn, pooled, out, dq stay
                          fp32.  dim must be 256; a fused x_drop needs precomputed keep-bits.  0 (default): fp32 */
  uint32_t* tickets;   /* NULL, or 2 * V device counters that are ZERO when the call starts (and again when it ends): the
                          per-sample second passes then run inside the first kernels -- the last 64-frame chunk of a sample to
                          finish combines that sample's softmax partials (forward, counters [0, V)) / sums its dq slabs
                          (backward, counters [V, 2V)) -- instead of in a second launch.  Same arithmetic in the same order.
                          (The engine leaves this NULL: at the MOSEI shapes the second launch measured cheaper, see engine.hip.) */
  int32_t partial_only; /* forward: 1 = stop after the per-chunk pass: `workspace` then holds the flash-style partials -- unnormalised
                          pooled rows [V][nchunk][nq][dim], then {chunk max, chunk sum} [V][nchunk][2][8] -- and `attn` the unnormalised
                          weights; out / pooled are NOT written.  The caller finishes the softmax itself (the engine's clustered
                          utterance-level stages do: csrc/chain_cluster.hip).  Requires tickets == NULL.
                          sdumc_attnpool_bwd_multi with f.partial_only = 1: the per-chunk dq slabs [V][nchunk][nq][256] stay in
                          `workspace` and dq is NOT written (the caller sums the chunks). */
} sdumc_attnpool;

size_t sdumc_attnpool_fwd_workspace_bytes(int32_t V, int32_t T, int32_t nq);
size_t sdumc_attnpool_fwd_workspace_bytes_dim(int32_t V, int32_t T, int32_t nq, int32_t dim);
size_t sdumc_attnpool_bwd_workspace_bytes_dim(int32_t V, int32_t T, int32_t nq, int32_t dim);
int sdumc_attnpool_fwd(const sdumc_attnpool* p, void* stream);

/* backward of the above. Produces
 *   dz    [V,T,256]  gradient w.r.t. the pre-tanh key projection (feeds the dW / dX GEMMs)
 *   dxd   [V,T,256]  gradient w.r.t. drop(x) through the pooling ("V") path; the key-projection
 *                    path is ADDED to it by the caller's NN GEMM (accumulate=1)
 *   dq    [V,nq,256] gradient w.r.t. the projected queries (nq=1 shared query: per-v partials,
 *                    reduce over v with sdumc_colsum)
 */
typedef struct sdumc_attnpool_bwd {
  sdumc_attnpool f;      /* the forward descriptor (x, keys, q, attn, pooled, dropouts) */
  const float* dout;     /* [V, nq, 256] gradient w.r.t. out */
  float* dz;
  float* dxd;
  float* dq;
  float* workspace;      /* >= sdumc_attnpool_bwd_workspace_bytes(V, T, nq): per-chunk dq slabs */
  size_t workspace_bytes;
  float* dq_sum;         /* optional [nq, dim]: SHARED query (f.q_stride == 0, FRA2UTT_new's context vector, model :50-52): the
                            sum over v of the per-sample gradients, written by ONE launch that reduces the per-chunk slabs of
                            every sample in a fixed order; dq is then NOT written.  NULL = per-sample dq as above.  Single-site
                            sdumc_attnpool_bwd only (the _multi form requires NULL). */
  float* dout_masked;    /* optional [V, nq, 256]: dout * out-dropout mask as the kernel uses it -- for a consumer that adds the pooling
                            part of dxd itself (sdumc_rows_problem.pool_g); dxd may then be NULL and is not written */
} sdumc_attnpool_bwd_t;

size_t sdumc_attnpool_bwd_workspace_bytes(int32_t V, int32_t T, int32_t nq);
int sdumc_attnpool_bwd(const sdumc_attnpool_bwd_t* p, void* stream);
/* K3, the fused UMCA forward (Cross_Attention model :79-95, FRA2UTT_new :56-68 = the nq = 1 case): key
 * projection + scores + softmax partials + pooling of one 64-frame chunk in ONE kernel -- K = tanh(drop(x) W_in^T + b) is
 * computed on the matrix cores into an LDS tile and consumed there; it goes to HBM only if a.keys != NULL (training keeps the
 * keys for sdumc_attnpool_bwd; inference passes NULL and the [V, T, 256] tensor never exists).  Takes the place of
 * the NT key-projection GEMM (fused input dropout, bias, tanh) followed by sdumc_attnpool_fwd on the same descriptor; same results.
 * fp32, 256 channels, input mask as keep-bits (x_drop.bits) or none, a.tickets NULL. */
typedef struct sdumc_umca {
  sdumc_attnpool a;      /* as for sdumc_attnpool_fwd; a.keys = OUTPUT [V, T, 256] or NULL */
  const float* w_in;     /* input_proj.weight [256, 256] as stored (model :60 / :82) */
  const float* b_in;     /* input_proj.bias [256] */
  /* optional, both or neither: the projection's operands split once per tensor (sdumc_gemm_p3): the frames a.x as a P3 tensor
   * ([x_samples * T][256], rows of 1536 bytes) and w_in fragment-major (sdumc_p3_split_frag).  a.x (fp32) is still read by the
   * pooling part.  Needs the split arithmetic on (sdumc_set_split_ bit 2), else SDUMC_EINVAL. */
  const void* x_p3;
  const void* w_in_p3f;
} sdumc_umca;
int sdumc_umca_fwd(const sdumc_umca* p, void* stream);

/* Up to four pooling sites in ONE launch pair (forward: partial + combine; backward: rows + dq reduce) -- what the step uses for
 * its three Cross_Attention blocks, which are independent and too short to be worth a stream fork.  Every site: 256-channel
 * rows, keep-bits attached or no input mask, the same `bf16` flag, its OWN workspace.  Results are bit-identical to n single
 * calls.  Replaces the three Cross_Attention.forward calls of models/wengnet_mosei_multviews_text_missing.py:334-336. */
int sdumc_attnpool_fwd_multi(const sdumc_attnpool* sites, int32_t n, void* stream);
int sdumc_attnpool_bwd_multi(const sdumc_attnpool_bwd_t* sites, int32_t n, void* stream);
/* Measurement / test hook: 0 routes the 256-channel, keep-bits, two-pass-combine cases back to the round-3 pooling kernels
 * (the round-4 "v2" kernels -- every global load issued up front, XCD-aware stream pairing -- are the default; results are
 * bit-identical: tests/test_gpu_ops.py).  Process-wide. */
int sdumc_attnpool_set_v2_(int on);

/* ------------------------------------------------------------------------
 * Small fused element-wise / reduction kernels of the utterance-level network.
 * ---------------------------------------------------------------------- */

/* dz = dy * [y > 0] * scale  (backward of Linear->ReLU->Dropout given the saved
 * post-dropout output y, model :264-273);  in place allowed.  n elements. */
int sdumc_relu_drop_bwd(const float* dy, const float* y, float scale, float* dz, int64_t n, void* stream);

/* out[j] (+)= sum_r a[r, j], a is [rows, cols] with leading dimension lda (bias gradients).
 * Deterministic two-stage reduction; workspace >= sdumc_colsum_workspace_bytes. */
size_t sdumc_colsum_workspace_bytes(int64_t rows, int32_t cols);
int sdumc_colsum(const float* a, int64_t rows, int32_t cols, int32_t lda, float* out, int32_t accumulate,
                 float* workspace, void* stream);

/* y[i] = sum_k x_k[i] (k <= 8 inputs), or out = a + b*c style helpers */
int sdumc_add_n(const float* const* xs, int32_t k, float* y, int64_t n, void* stream);

/* dx[b,t,:] = sum over `k` (site,stream) terms of g_k[v,t,:] * mask_k   (model backward of the
 * four dropout applications that read one projected feature tensor x) */
typedef struct sdumc_dropsum {
  int32_t terms;                 /* <= 8 */
  const float* g[8];             /* each [samples, T, 256] (already offset to the stream) */
  sdumc_dropout drop[8];         /* per-term dropout; stream offset given through call0/dev_state + stream_idx */
  int32_t stream_idx[8];
  int32_t samples, T;
  float* dx;                     /* [samples, T, 256] */
  int32_t bf16;                  /* 1: g[k] and dx are bf16 tensors (the engine's bf16-storage mode); 0: fp32 */
} sdumc_dropsum;
int sdumc_dropsum_bwd(const sdumc_dropsum* p, void* stream);

/* bf16-storage mode: out[row, :] = bf16(x[row % x_rows, :] * keep * scale) -- the masked frames drop(x) of one attention site
 * (nn.Dropout inside FRA2UTT_new / Cross_Attention, model :59, :81) materialised once, from keep-bits produced by
 * sdumc_dropout_bits (byte [row * width/4 + q], bit e = keep column 4q + e).  x, out: bf16 [.., width], width % 8 == 0. */
int sdumc_mask_apply_bf16(const void* x, const uint8_t* bits, void* out, int64_t rows, int64_t x_rows, int32_t width, float scale,
                          void* stream);

/* Modality fusion (model :301-332 algebra).  u [V,3,256], alpha [V,3]
 *   qin [7][V,256] = (f, f_at, f_tv, f_av, u_a, u_t, u_v)                         */
int sdumc_fusion_fwd(const float* u, const float* alpha, float* qin, int32_t V, void* stream);
/* dqin [7][V,256] -> du [V,3,256] (overwritten), dalpha [V,3] (ACCUMULATED: the second-level
 * fusion's contribution is already there, it runs earlier in the backward order) */
int sdumc_fusion_bwd(const float* u, const float* alpha, const float* dqin, float* du, float* dalpha,
                     int32_t V, void* stream);

/* Second-level fusion (model :346-358).  c [3][V,7,128] (modality-major), alpha [V,3] -> h [V,7,128] */
int sdumc_hweight_fwd(const float* c, const float* alpha, float* h, int32_t V, void* stream);
/* dh [V,7,128] -> dc [3][V,7,128] = alpha_m dh (+ dct [V,7,128] on m = 1, the external gradient of
 * cross_hiddens[:,1]; may be NULL), dalpha [V,3] (overwritten) */
int sdumc_hweight_bwd(const float* c, const float* alpha, const float* dh, const float* dct, float* dc,
                      float* dalpha, int32_t V, float relu_scale, void* stream);
/* relu_scale > 0: dc is additionally multiplied by [c > 0] * relu_scale (c is the post-ReLU/dropout output of
 * cross_*_mlp, so this is the gradient w.r.t. its pre-activation); 0: plain */
/* z[v,:] = sum_i beta[v,i] h[v,i,:]  (model :356-358) */
int sdumc_zpool_fwd(const float* h, const float* beta, float* z, int32_t V, void* stream);
/* dz [V,128] -> dh [V,7,128] (overwritten), dbeta [V,7] (overwritten) */
int sdumc_zpool_bwd(const float* h, const float* beta, const float* dz, float* dh, float* dbeta,
                    int32_t V, void* stream);

/* ------------------------------------------------------------------------
 * Losses (toolkit/utils/loss.py) — value and gradient in one call.
 * ---------------------------------------------------------------------- */

/* MSELoss (loss.py:19-33): loss = sum((pred-target)^2)/denom; dpred = w*2*(pred-target)/denom.
 * denom = len(pred) of the GLOBAL batch (= rows on one GPU).  loss_out (device scalar) is WRITTEN
 * with the unweighted local sum / denom. */
int sdumc_mse_fwd_bwd(const float* pred, const float* target, int32_t rows, float denom, float weight,
                      float* loss_out, float* dpred, void* stream);

/* Sum of squared differences, for RMSELoss (loss.py:37-51) = sqrt(ssd / numel).
 * Splitting ssd from the sqrt lets data-parallel ranks all-reduce ssd first. */
int sdumc_ssd(const float* a, const float* b, int64_t n, float* ssd_out, float* workspace, void* stream);
size_t sdumc_ssd_workspace_bytes(int64_t n);
/* da (+)= w*(a-b)/(rmse*numel_global), db (+)= -that ; rmse = sqrt(*ssd_global/numel_global)
 * loss_out written with rmse.  da / db may be NULL (detached side, main :148). */
int sdumc_rmse_bwd(const float* a, const float* b, int64_t n_local, const float* ssd_global, double numel_global,
                   float weight, float* loss_out, float* da, int32_t da_accumulate, float* db,
                   int32_t db_accumulate, void* stream);

/* RnCLoss (loss.py:271-315). feats [n,dim] = cat(r_stream0, r_stream1), labels [n] (already
 * repeated).  Computes the loss over ALL n rows and the gradient for rows
 * [row0, row0+rows_local) only (data-parallel: every rank holds the gathered feats).
 * loss_out written with the unweighted loss; dfeats [rows_local, dim] overwritten with
 * weight * dLoss/dfeats. */
size_t sdumc_rnc_workspace_bytes(int32_t n);
int sdumc_rnc_fwd_bwd(const float* feats, const float* labels, int32_t n, int32_t dim, float temperature,
                      float weight, int32_t row0, int32_t rows_local, float* loss_out, float* dfeats,
                      float* workspace, void* stream);
/* the same with the labels given once ([n/2]) and read as labels[j % (n/2)] (loss.py:283 labels.repeat(2, 1)) */
int sdumc_rnc_fwd_bwd_rep(const float* feats, const float* labels_half, int32_t n, int32_t dim, float temperature,
                          float weight, int32_t row0, int32_t rows_local, float* loss_out, float* dfeats,
                          float* workspace, void* stream);
/* The five batch-local terms of main :137-148 in two launches: MSELoss(vals_s, labels) for both streams and the
 * three RMSELoss pairs (text_hidden, cross_text, fused; stream 1 vs stream 0, teacher side detached for the
 * first two).  All tensors are the [2B, ...] stream-major network outputs; weights5 = (full_mse, missing_mse,
 * text_feat, text_query_feat, features); denom = B_global; ssd_global = all-reduced sums or NULL (computed
 * here).  Writes losses[1..5] (unweighted) and the four gradient buffers (every element, incl. the zeros). */
size_t sdumc_distill_workspace_bytes(int32_t B);
int sdumc_distill_fwd_bwd(int32_t B, float denom, const float* vals, const float* labels, const float* th,
                          const float* ct, const float* z, const float* weights5, const float* ssd_global,
                          float* d_vals, float* d_th, float* d_ct, float* d_z, float* losses, float* workspace,
                          void* stream);
/* gradient rows [row0, row0+rows) from the workspace a previous sdumc_rnc_fwd_bwd call filled
 * (a data-parallel rank owns two row ranges of the gathered matrix: its stream-0 and stream-1 rows) */
int sdumc_rnc_dfeat_rows(const float* feats, int32_t n, int32_t dim, float temperature, float weight, int32_t row0,
                         int32_t rows, float* dfeats, float* workspace, void* stream);
/* the boolean neg_mask of loss.py:303 for anchor i, exported for the bit-exactness test:
 * mask [n, n-1, n-1] uint8, diagonal removed as loss.py:294-296 does */
int sdumc_rnc_mask(const float* labels, int32_t n, uint8_t* mask, void* stream);

/* ------------------------------------------------------------------------
 * Optimiser: torch.optim.Adam(lr, betas, eps, weight_decay) with coupled L2
 * (main :317) over one flat parameter bucket.  hyper (device, 4 floats) =
 * {lr (host-written), step count t (incremented by the call), lr/(1-beta1^t), sqrt(1-beta2^t)}:
 * kept on the device so that a captured graph replays with the right step count.
 * grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
 * ---------------------------------------------------------------------- */
int sdumc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    float* hyper, float beta1, float beta2, float eps, float weight_decay,
                    float grad_scale, void* stream);

/* Batch assembly from a device-resident packed feature store = the collater's padding
 * (toolkit/utils/read_data.py:139-151, :223-248; toolkit/data/feat_data.py:232-253) on the GPU:
 * out[b, t, :] = packed[start[b] + t, :] for t < len[b], zero beyond, out is [B, Tmax, d].
 * packed [sum T, d] fp32, start int64 [B], len int32 [B] (device), d % 4 == 0. */
int sdumc_gather_pad(const float* packed, const int64_t* start, const int32_t* len, int32_t B, int32_t Tmax, int32_t d,
                     float* out, void* stream);
/* The same from store-wide tables: start_all int64 [N], len_all int32 [N] (device, one entry per stored utterance) and a
 * device index vector idx int64 [B] naming the batch's utterances -- an epoch then assembles its batches without copying
 * per-batch tables to the device.  len_out (optional, int32 [B]): min(len, Tmax) of the batch = the `lengths` the
 * key-padding extension takes (maxT - pad_len of feat_data.py:244-253). */
int sdumc_gather_pad_idx(const float* packed, const int64_t* start_all, const int32_t* len_all, const int64_t* idx, int32_t B,
                         int32_t Tmax, int32_t d, float* out, int32_t* len_out, void* stream);

/* A WHOLE batch in one launch (the collater above for every modality at once, toolkit/data/feat_data.py:232-253): up to
 * SDUMC_GATHER_MAX_SEGS packed tensors -- the four modalities' features and, when the store holds them, their P3 planes
 * (sdumc_p3_split of the packed tensor, made once per dataset: the planes of a zero row are zero bytes, so gathering plane rows
 * equals splitting the gathered batch bit for bit) -- into the step's input buffers, the labels (labels_out[b] =
 * labels_all[idx[b]], both or neither) and the valid frame counts (seg.len_out, optional).  A segment's rows are `d4` 16-byte
 * units wide (fp32 rows: d / 4; bf16 rows: d / 8; P3 rows: 3 d / 8), whatever they hold.  A segment with map_out writes the batch's
 * ROW MAP instead of (or beside) the padded copy: 4 bytes per frame instead of its 10 d.  max_workgroups > 0 caps the grid (the
 * kernel strides): a capped launch on a side stream fills the NEXT step's buffers beside the running step without queueing in
 * front of its kernels (sdumc_amd/engine.py, FusedTrainer); 0 = one pass.  seg.unit0 and total are filled by the call. */
#define SDUMC_GATHER_MAX_SEGS 8
typedef struct sdumc_gather_seg {
  const void* packed;        /* [sum T, 16 d4 bytes] */
  const int64_t* start_all;  /* [N] first row of utterance e */
  const int32_t* len_all;    /* [N] its frames */
  void* out;                 /* [B, Tmax, 16 d4 bytes]; NULL (with d4 = 0 and map_out given): no padded copy, the map only */
  int32_t* len_out;          /* optional [B] */
  int32_t* map_out;          /* optional int32 [B * Tmax]: store row of batch row (b, t) = start_all[e] + t, or zero_row for padding
                                (sdumc_net_io.row_map: the step then reads the batch in place) */
  int32_t zero_row;          /* index of an all-zero row of the packed tensor (the store appends one) */
  int32_t Tmax, d4;
  int64_t unit0;            /* (filled by the call: first output row of the segment in the launch's row space) */
} sdumc_gather_seg;
typedef struct sdumc_gather_desc {
  sdumc_gather_seg seg[SDUMC_GATHER_MAX_SEGS];
  int32_t nseg, B;
  const int64_t* idx;        /* device [B] */
  const float* labels_all;   /* [N] or NULL */
  float* labels_out;         /* [B] or NULL */
  int64_t total;             /* (filled by the call: output rows of all segments) */
} sdumc_gather_desc;
int sdumc_gather_batch(const sdumc_gather_desc* g, int32_t max_workgroups, void* stream);

/* ------------------------------------------------------------------------
 * Generic fairseq-style multi-head attention and the pieces of the pre-LN Transformer encoder
 * (toolkit/models/modules/transformers_encoder/, all three files).
 * Activations are Time x Batch x Channel ([T, B, E] row-major) as in the reference (multihead_attention.py:51).
 * Dropout descriptors on [T, B, E] tensors use samples = T, rows = B, width = E; on the attention
 * probabilities [B*H, Tq, Tk] samples = B*H, rows = Tq, width = Tk (a width that is not a multiple of 4
 * uses ceil(width/4) Philox calls per row).
 * ---------------------------------------------------------------------- */

/* nn.LayerNorm(E) (transformer.py:201-203): y = (x - mean) * rstd * gamma + beta over the last axis,
 * biased variance, rstd = 1/sqrt(var + eps).  mean / rstd [rows] are saved for the backward. */
int sdumc_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int64_t rows, int32_t width, float eps, void* stream);
/* dx = LayerNorm backward (+ dx_add when not NULL: the residual branch's gradient meeting at x; dx_add may alias dx);
 * dgamma, dbeta [width] overwritten (deterministic two-stage column reduction through `workspace`). */
size_t sdumc_layernorm_bwd_workspace_bytes(int64_t rows, int32_t width);
int sdumc_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                        float* dx, float* dgamma, float* dbeta, const float* dx_add, int64_t rows, int32_t width,
                        float* workspace, size_t workspace_bytes, void* stream);

/* Softmax over the key axis of the attention scores, with the additive mask, the attention dropout and the
 * head-averaged weights the reference returns (multihead_attention.py:104-117, :128-130). */
typedef struct sdumc_softmax {
  int32_t batch, heads, tq, tk;
  float scale;            /* d_h^-0.5: the reference scales q (:84); here it multiplies S = q.k^T */
  const float* mask;      /* optional additive [tq, tk] (attn_mask.unsqueeze(0), :108); may hold -inf */
  float* scores;          /* in: S [batch*heads, tq, tk]; out: P = softmax(scale*S + mask) (saved for backward) */
  float* probs_drop;      /* drop.enabled: P * keep/(1-p), what multiplies V (:117); else unused (may be NULL) */
  float* weights;         /* optional [batch, tq, tk]: mean over heads of what multiplies V (:128-130) */
  sdumc_dropout drop;     /* F.dropout(attn_weights, p=attn_dropout) (:117) */
} sdumc_softmax;
int sdumc_softmax_fwd(const sdumc_softmax* s, void* stream);
/* dscores [batch*heads, tq, tk]: in dL/d(what multiplies V), out (in place) dL/dS.  Reads s->scores (= P). */
int sdumc_softmax_bwd(const sdumc_softmax* s, float* dscores, void* stream);

/* y = drop(alpha * x + pos) + residual over a [samples, rows, width] tensor:
 *   the "dropout -> add residual" of TransformerEncoderLayer.forward (transformer.py:161-162, :168-172; alpha = 1),
 *   the embedding stage of TransformerEncoder.forward (:68-71; alpha = sqrt(E), pos_table = sinusoidal table,
 *   position of token (t, b) = t + 1 if x[t, b, 0] != 0 else 0 -- position_embedding.py:8-26 with padding_idx 0),
 *   and the backward of both (x = dy, residual NULL). */
typedef struct sdumc_dropadd {
  const float* x;
  float alpha;
  const float* pos_table;  /* optional [samples + 1, width], row 0 = padding row */
  const float* pos_src;    /* tensor whose first channel decides padding (the un-scaled input); NULL with pos_table NULL */
  const float* residual;   /* optional */
  float* y;
  int32_t samples, rows, width;
  sdumc_dropout drop;
} sdumc_dropadd;
int sdumc_drop_add(const sdumc_dropadd* d, void* stream);

/* MultiheadAttention.forward (multihead_attention.py:48-131) and its backward.
 * query [tq, B, E], key / value [tk, B, E].  Pointer equality of query / key / value selects the fused
 * projection paths the reference selects with data_ptr() tests (:61-62); the results do not depend on it.
 * add_bias_kv / add_zero_attn (:28-38, default off, never enabled by transformer.py) are not implemented. */
typedef struct sdumc_mha {
  int32_t tq, tk, batch, embed, heads;
  const float* query;
  const float* key;
  const float* value;
  const float* in_proj_weight;   /* [3E, E] rows = (q | k | v) (:24, :133-154) */
  const float* in_proj_bias;     /* [3E] or NULL */
  const float* out_proj_weight;  /* [E, E] (:30) */
  const float* out_proj_bias;    /* [E] or NULL */
  const float* attn_mask;        /* optional additive [tq, tk] */
  sdumc_dropout attn_drop;       /* enabled = training and attn_dropout > 0 */
  float* out;                    /* [tq, B, E] */
  float* weights;                /* [B, tq, tk] head-averaged attention weights, or NULL */
  /* caller-owned intermediates, saved for the backward */
  float* q;                      /* [tq, B, E] projected queries (unscaled) */
  float* k;                      /* [tk, B, E] */
  float* v;                      /* [tk, B, E] */
  float* probs;                  /* [B*H, tq, tk] softmax output */
  float* probs_drop;             /* [B*H, tq, tk] when attn_drop.enabled, else NULL */
  float* ctx;                    /* [tq, B, E] attention output before out_proj */
  float* workspace;              /* >= sdumc_mha_workspace_bytes */
  size_t workspace_bytes;
  int32_t bf16;                  /* 1: every product (projections, q.k^T, P.v and their backward) rounds its operands to bf16,
                                    fp32 accumulate (needs embed and head_dim multiples of 4; otherwise the call stays fp32);
                                    softmax, dropout, bias and the stored tensors are fp32 either way.  0: exact fp32 */
  /* add_bias_kv / add_zero_attn (multihead_attention.py:28-38, :86-104).  With ts = tk + (bias_k != NULL) + add_zero_attn the
   * source length becomes ts: k, v are [ts, B, E], probs / probs_drop [B*H, tq, ts], weights [B, tq, ts], attn_drop.width = ts;
   * attn_mask stays [tq, tk] (the extra columns are unmasked, as the reference's zero-padded mask).  Defaults: NULL, NULL, 0. */
  const float* bias_k;           /* [E] or NULL; both or neither */
  const float* bias_v;
  int32_t add_zero_attn;
} sdumc_mha;

typedef struct sdumc_mha_grads {
  const float* dout;             /* [tq, B, E] */
  float* dquery;                 /* [tq, B, E] overwritten.  Where the forward inputs aliased, pass the same */
  float* dkey;                   /* [tk, B, E] aliasing here: the contributions are then summed into one buffer */
  float* dvalue;                 /* [tk, B, E] */
  float* d_in_proj_weight;       /* [3E, E] overwritten */
  float* d_in_proj_bias;         /* [3E] or NULL */
  float* d_out_proj_weight;      /* [E, E] */
  float* d_out_proj_bias;        /* [E] or NULL */
  float* d_bias_k;               /* [E], required when bias_k is given */
  float* d_bias_v;
} sdumc_mha_grads;

size_t sdumc_mha_workspace_bytes(const sdumc_mha* m, int32_t backward);
int sdumc_mha_forward(const sdumc_mha* m, void* stream);
int sdumc_mha_backward(const sdumc_mha* m, const sdumc_mha_grads* g, void* stream);

/* misc */
/* n (<= 8) strided copies in one launch */
typedef struct sdumc_copy_seg {
  const float* src; float* dst;
  int32_t ld_src, ld_dst, rows, cols;
} sdumc_copy_seg;
int sdumc_copy2d_multi(const sdumc_copy_seg* segs, int32_t n, void* stream);
/* dst[r, 0:cols] = src[r, 0:cols] for r < rows, with leading dimensions */
/* dst[r, 0:cols] += src[r, 0:cols] */
int sdumc_axpy2d(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int32_t rows, int32_t cols, void* stream);
int sdumc_copy2d(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int32_t rows, int32_t cols, void* stream);
int sdumc_fill(float* p, float v, int64_t n, void* stream);
/* device rng/step state helpers: state = {seed_lo, seed_hi, call0}; adds `inc` to call0 */
int sdumc_rng_advance(uint32_t* dev_state, uint32_t inc, void* stream);
/* keep-bits of a [streams*samples, rows, width] row space, one byte per 4 columns (see sdumc_dropout.bits) */
int sdumc_dropout_bits(const sdumc_dropout* d, int32_t streams, uint8_t* bits, void* stream);
/* the same for nsite (<= 4) sites d->site + s * site_stride that share the row space, in one launch
 * (width must be a multiple of 16: four quads are packed per 32-bit store; bits[s] 4-byte aligned) */
int sdumc_dropout_bits_multi(const sdumc_dropout* d, int32_t streams, int32_t nsite, int32_t site_stride,
                             uint8_t* const* bits, void* stream);
/* bf16-storage mode: the keep-bits of the TWO sites d->site and d->site + site_stride (they read the same frames: fra2utt_m and
 * cross_att_fra2utt_m, model :59, :81) and the masked frames xd[s][row, :] = bf16(x[row % x_rows, :] * keep_s * d->scale) of
 * both, in one pass over x -- what sdumc_dropout_bits_multi followed by two sdumc_mask_apply_bf16 produce, bit for bit.
 * x, xd[s]: bf16 [.., width], 16-byte aligned; width % 16 == 0; d->enabled must be set. */
int sdumc_dropout_bits_apply_bf16(const sdumc_dropout* d, int32_t streams, int32_t site_stride, uint8_t* const* bits,
                                  const void* x, int64_t x_rows, void* const* xd, void* stream);
/* writes the dropout mask values (0 or scale) of a [streams*samples, rows, width] tensor: test hook */
int sdumc_dropout_mask(const sdumc_dropout* d, int32_t streams, float* mask, void* stream);
const char* sdumc_version(void);

/* Optional per-launch timing of the GEMM kernels with HIP events recorded on the launch stream
 * (bench.py's roofline leg; not for use under graph capture).  enable(1) clears and starts,
 * report() synchronises the recorded events and aggregates per kernel variant. */
typedef struct sdumc_prof_entry {
  const char* name;      /* e.g. "gemm_nt_128x128" */
  int64_t launches;
  double total_ms;
  double total_flops;    /* algorithmic: 2*M*N*K*groups per launch */
} sdumc_prof_entry;
int sdumc_profile_enable(int on);
int sdumc_profile_report(sdumc_prof_entry* out, int max_entries);

/* ========================================================================
 * Network level: WengnetMOSEIMultViewsTextMissing (model :186-370) as one call.
 *
 * Parameters live in ONE flat fp32 buffer: the live parameters (those that receive a
 * gradient) first, the dead ones after; every tensor starts on a
 * 16-byte boundary.  sdumc_param_table() prints "name offset rows cols live\n" lines with the
 * reference's state_dict names, so the Python nn.Module exposes nn.Parameters that are views of
 * the flat buffer and the published checkpoint loads by name.
 * ==================================================================== */
int64_t sdumc_param_count(int32_t da, int32_t dt, int32_t dv);      /* floats in the flat buffer */
int64_t sdumc_param_live_count(int32_t da, int32_t dt, int32_t dv); /* floats in the live prefix (gradient bucket) */
int32_t sdumc_param_table(int32_t da, int32_t dt, int32_t dv, char* buf, size_t buflen); /* bytes written or <0 */

typedef struct sdumc_net_dims {
  int32_t B;        /* local batch */
  int32_t streams;  /* 1 = one reference forward call; 2 = both streams of main :119,:131 batched */
  int32_t Ta, Tv;   /* frames of audio / video (padded to the batch max by the collater) */
  int32_t Tt[2];    /* frames of the text-slot input of stream 0 (text) and stream 1 (feat4) */
  int32_t da, dt, dv;   /* feature widths = args.input_dims[0:3] (model :193-195) */
  int32_t train;    /* 1: dropout on (model.train()), 0: identity (model.eval()) */
  int32_t sample0;  /* global index of local sample 0 (data-parallel shard offset) */
  double p_frame;   /* 0.5: nn.Dropout inside FRA2UTT_new / Cross_Attention (model :54,:77) */
  double p_mlp;     /* 0.3: constructor default dropout (model :187) */
  int32_t bf16;     /* 0 (default): exact fp32 everywhere.
                       1: fp32 storage; the frame-level projections (frame_dim_reshape_*, model :282-284, and the input_proj
                       keys of FRA2UTT_new / Cross_Attention, model :60,:82), forward and backward, round their operands to bf16
                       on the way to v_mfma_f32_32x32x16_bf16 (fp32 accumulation).
                       2: bf16 STORAGE (BASELINE configs[2], configs[4]): the input features (sdumc_net_io.audio / video /
                       text are then bf16 tensors), the projected frames, the masked frames, the tanh keys and the frame-level
                       gradients live in HBM as bf16; products accumulate in fp32, bias / tanh / softmax / pooling / the
                       utterance-level network / losses / Adam (fp32 master weights) stay fp32.  Needs da, dt, dv % 64 == 0. */
} sdumc_net_dims;

typedef struct sdumc_net_io {
  const float* audio;    /* [B, Ta, da] */
  const float* video;    /* [B, Tv, dv] */
  const float* text[2];  /* [B, Tt[s], dt]: stream 0 = text, stream 1 = feat4 (streams==1: text[0]) */
  float* params;         /* flat parameter buffer */
  const uint32_t* rng_state; /* device {seed_lo, seed_hi, call0}; stream s draws call0 + s */
  void* workspace;       /* >= sdumc_net_workspace_bytes(dims); holds the activations saved for backward,
                            so it must stay untouched between forward and backward */
  size_t workspace_bytes;
  /* outputs, V = streams*B rows, stream-major (model :370) */
  float* vals;           /* [V, 1]      vals_out */
  float* fused;          /* [V, 128]    cross_fused_feat */
  float* rnc;            /* [V, 64]     feat4rnc */
  float* text_hidden;    /* [V, 256]    text_hidden (after cross_text_query_mlp, model :329) */
  float* cross_text;     /* [V, 7, 128] cross_hiddens[:,1] */
  /* EXTENSION, default all NULL = reference behaviour: device int32 [B] valid frame counts of audio, text, video, feat4
   * (maxT - pad_len of toolkit/data/feat_data.py:232-253's `pads`); when given, the six attention poolings mask the
   * padded frames (sdumc_attnpool.lengths).  All four (three when streams == 1) or none. */
  const int32_t* lengths[4];
  /* Optional, fp32 storage only: the same features as P3 tensors (sdumc_p3_split: three bf16 planes, 6 bytes per element, rows of
   * 6 d bytes, 16-byte aligned) -- all of them (audio_p3, video_p3, text_p3[0 .. streams - 1]) or none.  When given, the frame
   * projections (model :282-284) and the Cross_Attention key projections (model :82) read planes that were split ONCE -- the
   * features when the batch was installed (they do not change across epochs), the projected frames by the projection's own epilogue --
   * instead of splitting fp32 operands per workgroup per k-tile (csrc/gemm_p3.hip).  The fp32 tensors above are still read by the
   * weight gradients.  Same arithmetic, same results to the last fp32 rounding or two (another summation order). */
  const void* audio_p3;
  const void* video_p3;
  const void* text_p3[2];
  /* Optional, fp32 storage with planes: the batch is read IN PLACE from a resident feature store.  audio / video / text[s] (and the
   * *_p3 pointers) are then the store's PACKED tensors [rows + 1][d] -- every utterance's frames one after the other, of any total size
   * (tens of GB: 64-bit addressing), the last row all zero -- and row_map[i] (i = audio, text, video, feat4 as in `lengths`; device
   * int32 [B * T_i]) names, for batch row b * T_i + t, the store row it is: start of utterance b + t for a valid frame, the zero row for
   * padding (sdumc_gather_batch writes such maps: sdumc_gather_seg.map_out).  No padded copy of the batch exists anywhere: the frame
   * projections fetch their A rows through the map (sdumc_gemm_p3.a_map), the frame projections' weight gradients their B rows
   * (sdumc_gg_problem.b_map); nothing else reads the features.  All four (three when streams == 1) or none; fp32 storage: needs the
   * *_p3 planes and the default split arithmetic; bf16 storage (audio / video / text[s] = the store's packed bf16 tensors): feature
   * widths multiples of 128; else SDUMC_EINVAL.  The bf16 weight-gradient kernel reads map entries four at a time: each map must be
   * readable up to a multiple of four entries.  store_rows[i] (optional): rows of packed tensor i -- tensors (and planes) below
   * 4 GiB keep the kernels' descriptor addressing, which is faster than 64-bit addresses; 0 = unknown. */
  const int32_t* row_map[4];
  int64_t store_rows[4];
  /* Optional, train mode in fp32 storage: sdumc_net_bits_next_bytes(dims) bytes the caller keeps ACROSS calls of the same dims (its
   * own allocation, not part of `workspace`; zero it once) -- TWO sets of keep-bits for the frame-level input dropouts, each with a
   * {seed, call} tag.  A call reads set (bits_phase & 1); in its latency-bound middle, where the chip idles, it fills the OTHER set
   * for the next call (Philox call index + 2: what sdumc_train_step advances by) and tags it.  The caller flips bits_phase from call
   * to call (forward and backward of one step take the same value); the next call's head launch then finds its set tagged with its
   * own {seed, call} and has nothing to generate -- at the head the Philox launches compete with the frame projections.
   * Bit-identical masks either way: a set whose tag names another seed, call or batch shape (first call, a reset counter, a different
   * advance, a phase that was not flipped, a ragged epoch's next shape that was not announced) is regenerated at the head of the call
   * that reads it, and re-tagged for that call.
   * Shapes that change from call to call (toolkit/utils/read_data.py:223-248 pads every batch to its own maximum): the buffer is
   * sized for the LARGEST dims of the run (bits_next_bytes = sdumc_net_bits_next_bytes of those; 0 = exactly this call's, the set
   * stride then follows the call's own dims), and bits_next_dims names the dims of the NEXT call (same widths, streams, train and
   * bf16 fields; B, sample0 and the frame counts may differ; NULL = the same as this call): the middle of this call lays the other
   * set out for them.
   * Ordering: sdumc_net_forward returns with the fill of the other set ordered before `stream` (a caller may then advance the
   * counter, or free / re-zero the buffer, behind `stream`); inside sdumc_train_step that join is left to the backward. */
  void* bits_next;
  int32_t bits_phase;
  size_t bits_next_bytes;
  const struct sdumc_net_dims* bits_next_dims;
  /* Optional: the assembly of the NEXT batch (sdumc_gather_batch's descriptor: its row maps, or its padded copy; the output buffers
   * must not be ones this call reads), issued by the call itself on its background lane at the start of its utterance-level middle --
   * from there to the backward's frame-level tail the chip's HBM is mostly idle (the step is latency- and matrix-bound there), so the
   * batch is put together beside the step instead of in front of the next one.  sdumc_train_step (and a sdumc_net_forward on its own)
   * returns with the gather ordered before `stream`.  prefetch_workgroups caps its grid (0 = 512: two small workgroups per CU leave
   * every CU room for the step's own kernels).  The descriptor is read during the call only. */
  const sdumc_gather_desc* prefetch;
  int32_t prefetch_workgroups;
  /* Optional caller-owned execution context (sdumc_ctx_create): the internal side streams and the event ring the call forks
   * its branches on.  NULL = the default context of the current device (one per device, created on first use).  Two host
   * threads that run steps concurrently -- on distinct streams of one device, or on two devices -- give each its own
   * context; a context is used by one thread at a time and only on the device it was created on (else SDUMC_EINVAL). */
  void* ctx;
} sdumc_net_io;

/* Execution contexts (see sdumc_net_io.ctx).  Create / destroy outside stream capture; destroy only when no work issued
 * through the context is pending.  Stream lifetime: the clustered utterance-level launches of a device order themselves behind
 * the previous such launch with an event recorded on THAT launch's stream -- a caller that destroys the stream it ran its last
 * step on must have synchronised it first (torch's pooled streams are never destroyed; the library's own lanes are handled by
 * sdumc_ctx_destroy). */
size_t sdumc_net_bits_next_bytes(const sdumc_net_dims* d);   /* 0 = the dims do not use it (eval, bf16 storage) */
int sdumc_ctx_create(void** ctx);
int sdumc_ctx_destroy(void* ctx);
/* Schedule options of ONE execution context (ctx NULL = the current device's default context): nothing process-wide changes, so
 * two callers -- or two tests -- with their own contexts cannot disturb each other.  value < 0 returns the option to the
 * process-wide default (the sdumc_set_* calls below, kept for callers without a context).
 *   SDUMC_OPT_CONCURRENCY      0 = every launch on the caller's stream (per-kernel profiling), 1 = the internal lanes
 *   SDUMC_OPT_BACKGROUND_LANE  0 / 2 / 3 as sdumc_set_background_lane
 *   SDUMC_OPT_CHAIN_CLUSTER    0 = csrc/chain.hip's kernels, 1 = the clustered ones wherever the shape fits
 *   SDUMC_OPT_SPLIT            0..15: which fp32 GEMM kernel families multiply on the bf16 matrix pipe (bits as sdumc_set_split_) */
enum { SDUMC_OPT_CONCURRENCY = 0, SDUMC_OPT_BACKGROUND_LANE = 1, SDUMC_OPT_CHAIN_CLUSTER = 2, SDUMC_OPT_SPLIT = 3 };
int sdumc_ctx_set_option(void* ctx, int32_t option, int32_t value);

/* The network-level calls issue independent branches (the three per-modality chains; the dW GEMMs) on up to
 * two internal side streams forked from / joined to `stream` with events: the call stays stream-ordered
 * with respect to `stream`.  sdumc_set_concurrency(0) keeps everything on `stream` (per-kernel profiling:
 * overlapping kernels share the GPU, so their individual durations stop being meaningful). */
int sdumc_set_concurrency(int on);
/* The Cross_Attention-site key projections (forward, dW, dX: ~0.4 ms of MFMA-bound GEMMs per step at C2) are not on
 * the critical path of the launch-bound utterance-level chain (model :293-332 forward, its mirror backward).  1 = issue
 * them on a fourth internal stream beside that chain instead of grouped with the FRA2UTT-site ones, 2 = the forward
 * ones only (+1.0 % per step over mode 0 on MI355X: the HBM-bound FRA2UTT pooling then overlaps the MFMA-bound
 * Cross_Attention key GEMM), 3 (the default) = 2 plus the audio modality's backward ones (+0.6 % over 2: the longest
 * frame-level chain gets shorter), 4 = audio and video (+0.4 %); mode 1 loses 1.6 % against mode 2. */
int sdumc_set_background_lane(int on);
/* Debug timeline: with marks on, sdumc_train_step records an event on the caller's stream at fixed points (0 start, 1/2 around
 * the first utterance-level launch, 3/4 the second, 5 after the losses, 6 after the first backward utterance-level launch, 7/8
 * around the second, 9 before Adam, 10 end); _read returns their times in ms since mark 0 (-1 = not recorded) once the caller
 * has synchronised.  Process-wide and single-threaded: a measurement aid (tools/step_marks.py), not part of the data path. */
/* The utterance-level stages run as clustered kernels (four workgroups split every layer's columns and exchange their slices
 * through HBM + a flag, csrc/chain_cluster.hip) whenever all their workgroups fit the device at once (2 x streams x B <= 128
 * on MI355X) and the stream is not being captured; 0 keeps the one-workgroup-per-sample-pair kernels of csrc/chain.hip.
 * Failure contract: the members of a cluster wait for each other with a bounded spin.  A spin that runs into its cap (another
 * process holding the GPU's CUs, a hung member) sets a sticky per-device error word and lets the kernel finish with wrong data;
 * from then on every Adam launch of this library on that device (sdumc_train_step, sdumc_adam_step) applies NOTHING -- parameters
 * and moments keep their values -- and the fused step's total loss reads NaN.  sdumc_chain_cluster_error_() synchronises the
 * device and returns the word (0 = fine); sdumc_chain_cluster_reset_error() clears it (and the clusters' counters) once the
 * caller has dealt with the failed step, e.g. by repeating it with sdumc_set_chain_cluster(0).
 * sdumc_chain_cluster_test_hold_(1) is a test hook: workgroup 0 of every following clustered launch withholds its arrivals, which
 * drives exactly that failure path on purpose (tests/test_gpu_net.py). */
int sdumc_set_chain_cluster(int on);
int sdumc_chain_cluster_error_(void);
int sdumc_chain_cluster_reset_error(void);
int sdumc_chain_cluster_test_hold_(int on);
/* Data-parallel runs (one process per GPU): the error word is per device, so the ranks exchange it with the gradients.
 * _flag writes 1.0f (word set) or 0.0f into `flag` -- a float the caller appends to its gradient bucket -- stream-ordered and
 * without a host synchronisation; after the SUM all-reduce of the bucket, _merge sets the local word when the reduced flag is
 * non-zero, so that every rank's Adam applies nothing and every rank raises (sdumc_amd/trainer.py). */
/* Diagnosis (SDUMC_CL_MODE bit 4): records of cached weight loads that differed from agent-scope loads of the same address in the
 * stage-A forward kernel: out[0] = count, out[8 + 12 i ..] = {workgroup, thread, ring slot, address low word, 4 cached words, 4
 * coherent words}; synchronises the device and clears the records.  Not part of the data path. */
int sdumc_chain_cluster_debug_read_(uint32_t* out, int n);
int sdumc_chain_cluster_error_flag(float* flag, void* stream);
int sdumc_chain_cluster_error_merge(const float* flag, void* stream);
int sdumc_debug_marks(int on);
int sdumc_debug_marks_read(float* ms, int n);
/* Debug: the workspace plan of sdumc_net_forward as text lines "name offset length" (floats, relative to
 * sdumc_net_io.workspace); returns the text length, or minus the buffer size needed.  For probes that compare intermediate
 * tensors of two runs (tools/fwd_determinism_probe.py); not part of the data path. */
int32_t sdumc_debug_plan_table(const sdumc_net_dims* d, char* buf, size_t buflen);
size_t sdumc_net_workspace_bytes(const sdumc_net_dims* d);
int sdumc_net_forward(const sdumc_net_dims* d, const sdumc_net_io* io, void* stream);

typedef struct sdumc_net_grads {
  const float* d_vals;        /* gradients w.r.t. the five outputs, same shapes; NULL = zero */
  const float* d_fused;
  const float* d_rnc;
  const float* d_text_hidden;
  const float* d_cross_text;
  float* grads;               /* [live count] flat gradient bucket; every live tensor is OVERWRITTEN
                                 (alignment padding is left untouched: allocate it zeroed) */
} sdumc_net_grads;

/* loss.backward() through the network (main :149).  Needs the workspace of the matching forward. */
int sdumc_net_backward(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_net_grads* g, void* stream);
/* The same in two calls, for a data-parallel step that overlaps its gradient all-reduce with the backward:
 * phase 0 = the utterance-level layers; when it returns (stream-ordered) grads[0, sdumc_param_early_count) are final.
 * phase 1 = the frame-level layers (input_proj of both attention sites, frame_dim_reshape: 0.9 of the 1.4 ms of
 * backward GEMM time at C2), which finish grads[early, live). */
int sdumc_net_backward_phase(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_net_grads* g, int32_t phase,
                             void* stream);
int64_t sdumc_param_early_count(int32_t da, int32_t dt, int32_t dv);

/* Two-stream self-distillation step (main :119-150). */
typedef struct sdumc_step_cfg {
  float weights[6];      /* full_mse, missing_mse, text_feat, text_query_feat, features, rnc (main :234-239) */
  float temperature;     /* RnCLoss temperature = 2 (loss.py:272) */
  float beta1, beta2, eps, weight_decay;   /* Adam (main :317) */
  const float* labels;   /* [B] sentiment values of the local samples */
  float* adam_m;         /* [live count] */
  float* adam_v;         /* [live count] */
  float* hyper;          /* device float[4], see sdumc_adam_step */
  float* losses;         /* device float[8]: total, mse_full, mse_missing, rmse_text, rmse_query, rmse_fused, rnc, 0.
                            MSE entries are the LOCAL sum / B_global (sum over ranks = global value). */
  /* data-parallel exactness; all zero/NULL on one GPU */
  int32_t B_global;      /* 0 -> B */
  const float* ssd_global;        /* device float[3]: all-reduced sums of squared differences (text, query, fused) */
  const float* rnc_feats_global;  /* [2*B_global, 64] = cat(r_stream0 of all ranks, r_stream1 of all ranks) */
  const float* rnc_labels_global; /* [2*B_global] */
  int32_t rnc_row0[2];   /* rows of this rank's stream-0 / stream-1 features in the gathered matrix */
} sdumc_step_cfg;

size_t sdumc_loss_workspace_bytes(const sdumc_net_dims* d, int32_t B_global);
/* local sums of squared differences of the three RMSE pairs -> ssd_out float[3] (for the all-reduce) */
int sdumc_loss_ssd(const sdumc_net_dims* d, const sdumc_net_io* io, float* ssd_out, void* scratch,
                   size_t scratch_bytes, void* stream);
/* The data-parallel exactness exchange as ONE record per rank,
 *   record = [ rnc features of this rank, stream-major (2*B x rd) | labels (B) | sums of squares (3) ],  n = 2*B*rd + B + 3 floats,
 * so that one all-gather of W records replaces the three exchanges.  sdumc_dp_unpack turns the gathered [W, n] matrix into
 * what sdumc_step_cfg takes: feats [2*W*B, rd] = (stream-0 rows of rank 0..W-1, then stream-1 rows of rank 0..W-1),
 * labels2 [2*W*B] = the W*B labels twice, ssd[3] = the per-rank sums added in rank order (same bits on every rank).
 * The reference has no data-parallel path: the truth these reproduce is the single-process step on the
 * whole batch (main_frame_val_text_missing.py:119-150). */
int sdumc_dp_pack(const float* rnc, const float* labels, const float* ssd, int32_t B, int32_t rd, float* record,
                  void* stream);
/* sdumc_loss_ssd + sdumc_dp_pack in one launch: the three sums of squares (text_hidden, cross_text, fused: rows [B, 2B)
 * minus rows [0, B) of each [2B, .] output) are reduced in a fixed order by the last block to finish.  `workspace`
 * (>= sdumc_dp_record_workspace_bytes(B), 4-byte aligned) must be ZERO before the first call; every call leaves it
 * ready for the next. */
size_t sdumc_dp_record_workspace_bytes(int32_t B);
int sdumc_dp_record(int32_t B, int32_t rd, const float* text_hidden, const float* cross_text, const float* fused,
                    const float* rnc, const float* labels, float* record, void* workspace, void* stream);
int sdumc_dp_unpack(const float* records, int32_t W, int32_t B, int32_t rd, float* feats, float* labels2, float* ssd,
                    void* stream);
/* loss values + gradients w.r.t. the five network outputs (written to the d_* buffers of `g`,
 * which here are OUTPUTS and must all be non-NULL) */
int sdumc_loss_backward(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_step_cfg* cfg,
                        const sdumc_net_grads* g, void* scratch, size_t scratch_bytes, void* stream);

/* forward (both streams) + losses + backward + Adam on one GPU.  workspace >= sdumc_step_workspace_bytes. */
size_t sdumc_step_workspace_bytes(const sdumc_net_dims* d);
int sdumc_train_step(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_step_cfg* cfg, void* stream);
/* byte offset of the flat gradient bucket inside the sdumc_train_step workspace (tests, DP all-reduce) */
size_t sdumc_step_grads_offset(const sdumc_net_dims* d);

#ifdef __cplusplus
}
#endif
#endif /* SDUMC_HIP_H */
