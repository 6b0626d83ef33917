// common.h — shared device/host helpers for libsdumc_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sdumc_hip.h"

#define SDUMC_CHECK_LAUNCH()                                 \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return SDUMC_ELAUNCH; \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// The dynamic-LDS limit of a kernel is a PER-DEVICE function attribute (a process that drives several GPUs sets it on each of
// them).  sdumc_once_per_device runs `setup` (returns true on success) the first time the CURRENT device asks: under a lock, and the
// device's bit is set only AFTER the setup has succeeded -- a second host thread on the same device either finds the attribute applied
// or waits for it, and a failed setup is retried by the next call instead of being remembered as done.
#ifdef __cplusplus
#include <atomic>
#include <mutex>
struct sdumc_dev_once {
  std::mutex mu;
  std::atomic<uint64_t> done{0};
};
template <class F>
static inline int sdumc_once_per_device(sdumc_dev_once& o, F&& setup) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return setup() ? SDUMC_OK : SDUMC_ELAUNCH;   // (idempotent: set again)
  const uint64_t bit = 1ull << dev;
  if (o.done.load(std::memory_order_acquire) & bit) return SDUMC_OK;
  std::lock_guard<std::mutex> lk(o.mu);
  if (o.done.load(std::memory_order_relaxed) & bit) return SDUMC_OK;
  if (!setup()) return SDUMC_ELAUNCH;
  o.done.fetch_or(bit, std::memory_order_release);
  return SDUMC_OK;
}
template <class K>
static inline bool sdumc_set_dyn_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}
#endif

// ---------------------------------------------------------------------------
// Internal entry points (not part of include/sdumc_hip.h): fused variants the engine's train step uses to take
// ~6.6 us dependent launches off its critical chain (measured: five such launches removed = 2.152 -> 2.119 ms).
// ---------------------------------------------------------------------------
// the fused step's weighted total loss (losses[0]) is written by the Adam launch instead of by a one-thread kernel of its own
struct sdumc_total_loss {
  float* losses;   // [8] or nullptr
  float w[6];
  const int32_t* chain_err;   // chain_cluster.hip's error word (a cluster spin ran into its cap) or nullptr: poisons the total with NaN
};
extern "C" {
// sdumc_zpool_bwd with dz := dz + dz_add (dz_add may be NULL): folds the external gradient of cross_fused_feat in
int sdumc_zpool_bwd_add_(const float* h, const float* beta, const float* dz, const float* dz_add, float* dh, float* dbeta,
                         int32_t V, void* stream);
// sdumc_relu_drop_bwd over rows of `row_len` with add[row, 0:width] added to dy[row, col0:col0+width] first (add may be NULL)
int sdumc_relu_drop_bwd_add_(const float* dy, const float* y, float scale, float* dz, int64_t n, const float* add,
                             int32_t row_len, int32_t col0, int32_t width, void* stream);
// the Adam bias-correction update alone (what sdumc_adam_step launches first) ...
int sdumc_adam_hyper_(float* hyper, float beta1, float beta2, void* stream);
// ... and the parameter update alone; rng_state != NULL: its call counter advances by rng_inc in the same launch
int sdumc_adam_apply_(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper,
                      float beta1, float beta2, float eps, float weight_decay, float grad_scale, uint32_t* rng_state,
                      uint32_t rng_inc, const struct sdumc_total_loss* total, void* stream);
}

// ---------------------------------------------------------------------------
// chain.hip: the utterance-level network in four launches.  Arguments = device pointers into the flat parameter buffer
// (backward), its transposed mirror (forward) and the engine's workspace.
// ---------------------------------------------------------------------------
struct sdumc_chain_fold {
  const float* part[3];       // [V][nchunk][nq][256] unnormalised pooled rows
  const float* stats[3];      // [V][nchunk][2][8]: chunk max, chunk sum per query
  float* attn[3];             // [V][T][nq] softmax weights, normalised in place
  float* pooled[3];           // [V][nq][256] pooled rows before the output dropout (the pooling backward reads them)
  int32_t nchunk[3], T[3];
  int32_t site[3];            // Philox site of the sites' output dropout
  uint32_t threshold;         // ... and its keep threshold / scale (p_frame)
  float scale;
};
struct sdumc_chain_args {
  int32_t V, B;                 // virtual samples (streams * B), samples per stream
  sdumc_dropout drop;           // template: enabled, threshold / scale of p_mlp, samples, sample0, dev_state (site, rows, width per layer)
  float relu_scale;             // 1 / (1 - p_mlp) in train mode, 1 in eval mode (backward masks)
  int32_t w_bf16;               // 1: the streamed weight matrices (everything below except fc_att / cfa / fcv / rnc*) are bf16 copies
  int32_t no_packed_fp32;       // 1: take the entry points compiled without packed FP32 VALU ops (bf16 MFMA kernels run beside: chain_common.h)
  // transposed weights (forward) / weights as stored (backward), biases
  const float *umlp0_w[3], *umlp0_b[3], *umlp3_w[3], *umlp3_b[3];
  const float *att0_w, *att0_b, *att3_w, *att3_b, *fc_att_w, *fc_att_b;       // fc_att_w always as stored [3][256]
  const float *query_w[7], *query_b[7], *caq_w[3], *caq_b[3];
  const float *cmlp0_w[3], *cmlp0_b[3], *cmlp3_w[3], *cmlp3_b[3];
  const float *catt0_w, *catt0_b, *catt3_w, *catt3_b, *cfa_w, *cfa_b;        // cfa_w as stored [7][128]
  const float *fcv_w, *fcv_b, *rnc0_w, *rnc0_b, *rnc2_w, *rnc2_b;             // fcv_w as stored [1][128]
  // activations (engine Plan offsets)
  float *hpre, *u1, *u, *att1, *att2, *alpha, *qin, *q, *qp, *ca_out, *c1, *c, *h, *e1, *e2, *beta, *z, *vals, *r1, *r;
  // outputs of the network (may be null)
  float *o_vals, *o_fused, *o_rnc, *o_text_hidden, *o_cross_text;
  // backward: external gradients (may be null) and gradient buffers
  const float *g_vals, *g_fused, *g_rnc, *g_text_hidden, *g_cross_text;
  float *d_r1, *d_z, *d_beta, *d_e2, *d_e1, *d_h, *d_c, *d_c1, *d_ca_out, *d_alpha;
  float *d_qp, *d_q, *d_qin, *d_u, *d_att2, *d_att1, *d_u1, *d_hpre;
  // chain_cluster.hip, forward stages: the flash-style softmax partials of the three pooling sites in front of the stage
  // (sdumc_attnpool.partial_only; layouts: attn_pool.hip fwd_ws, 256 channels), combined by the stage's own prologue instead of by
  // a combine launch in front of it.  fra: the FRA2UTT sites (nq = 1) -> stage A's input rows hpre; ca: the Cross_Attention sites
  // (nq = 7) -> stage B's input rows ca_out.  part[0] == nullptr: the stage reads its input rows from HBM as before.
  sdumc_chain_fold fra, ca;
  // stage A backward of chain_cluster.hip: the per-chunk dq slabs of the three Cross_Attention sites' pooling backward
  // (sdumc_attnpool_bwd_multi with partial_only), summed over the chunks by the stage's prologue (ascending chunk order, as
  // dq_reduce_multi did) and written to d_qp.  dq_part[0] == nullptr: the stage reads d_qp as before.
  const float* dq_part[3];      // [V][nchunk][7][256]
  int32_t dq_nchunk[3];
  // chain_cluster.hip only (filled by sdumc_chain_cluster_launch_): per-cluster arrival / departure counters, error word
  uint32_t* cl_flags;
  int32_t* cl_err;
  unsigned long long* cl_trace;   // null, or 32 timestamps of workgroup 0 (debug)
  int32_t cl_test_hold;           // test hook: workgroup 0 withholds its arrivals (sdumc_chain_cluster_test_hold_)
  int32_t cl_mode;                // exchange hand-shake: bit 0 = release / acquire fences at agent scope around the arrival counter
                                  // (SDUMC_CL_MODE; bits 1.. = diagnosis switches of the stage-A forward kernel, chain_cluster.hip)
  uint32_t* cl_dbg;               // diagnosis records (sdumc_chain_cluster_debug_read_)
};

extern "C" {
// loss.hip: the six loss launches of a single-GPU step in two (1 = shape not taken)
// hyper != nullptr: the second pass also makes the Adam bias-correction update of this step (as total_loss_kernel did)
int sdumc_losses_fused_(int32_t B, const float* vals, const float* labels, const float* th, const float* ct, const float* z,
                        const float* rnc_feats, int32_t rd, float temperature, const float* weights6, float* d_vals, float* d_th,
                        float* d_ct, float* d_z, float* d_rnc, float* losses, float* distill_ws, float* rnc_workspace,
                        float* hyper, double beta1, double beta2, void* stream);
// which: 0 = stage A forward, 1 = stage B forward, 2 = stage B backward, 3 = stage A backward
int sdumc_chain_launch_(const sdumc_chain_args* a, int which, void* stream);
// the same four stages with every layer's output columns split over clusters of 4 workgroups (chain_cluster.hip);
// returns 1 when the shape does not qualify (the caller then takes sdumc_chain_launch_)
int sdumc_chain_cluster_launch_(const sdumc_chain_args* a, int which, void* stream);
int sdumc_chain_cluster_ok_(int V);       // the process-wide switch is on AND the shape fits
int sdumc_chain_cluster_fits_(int V);     // capability only: every workgroup of the clustered kernels resident at once on this device
int sdumc_chain_cluster_forget_stream_(void* stream);    // before a stream is destroyed: it may be the one the next clustered launch orders itself behind
const int32_t* sdumc_chain_cluster_err_ptr_(void);    // device address of the error word (nullptr before the first cluster launch)
// dst[off ..] = transpose of the n listed [out][in] matrices of src (same offsets in both buffers)
// fp32 parameters -> bf16 copies as stored (dst) and, where want_t[i], transposed (dst_t); same element offsets as in src
int sdumc_gemm_small_tn_multi_(const sdumc_gemm* gs, int n, void* stream);
// the shape words of a keep-bits tag: {B, sample0, Ta, Tv, Tt[0], Tt[1]} of the call the set is laid out for
struct sdumc_bits_shape {
  uint32_t w[6];
};
int sdumc_dropout_bits_multi_ex_(const sdumc_dropout* d, int32_t streams, int32_t nsite, int32_t site_stride, uint8_t* const* bits,
                                 const uint32_t* tag, int32_t call_add, const struct sdumc_bits_shape* shape, void* stream);
int sdumc_bits_tag_(const sdumc_dropout* d, uint32_t* tag, int32_t call_add, const struct sdumc_bits_shape* shape, void* stream);
int sdumc_weights_to_bf16_(const float* src, void* dst, void* dst_t, const int64_t* offs, const int32_t* outs, const int32_t* ins,
                           const int32_t* want_t, int n, void* stream);
size_t sdumc_gg_slab_bytes_(int tiles);   // gemm_group.hip: workspace bound for sdumc_gemm_group_tn by output-tile count
int sdumc_gemm_rows_prepare_(void);       // gemm_rows.hip: the kernels' per-device attributes, set outside any stream capture
// fp32 GEMM kernels whose products run on the bf16 matrix pipe from exactly split operands (gemm_group.hip has the arithmetic):
// sdumc_split_on_(bit) is the process-wide switch the launchers ask (sdumc_hip.h: sdumc_set_split_)
#define SDUMC_SPLIT_GROUP 1   /* gemm_group.hip: the grouped weight-gradient launch */
#define SDUMC_SPLIT_WIDE 2    /* gemm_wide.hip: NT launches (frame projections, key projections) */
#define SDUMC_SPLIT_UMCA 4    /* attn_pool.hip: the fused key projection of sdumc_umca_fwd */
#define SDUMC_SPLIT_ROWS 8    /* gemm_rows.hip */
#define SDUMC_SPLIT_ALL 15
int sdumc_split_on_(int bit);
int sdumc_split_scope_(int mask);   // this host thread's override for the duration of a network-level call (-1 = none); returns the previous value
int sdumc_gemm_rows256_capped_(const sdumc_rows_problem* probs, int32_t n, int32_t max_wg, void* stream);   // sdumc_gemm_rows256 on <= max_wg workgroups
int sdumc_gemm_rows256_bf16_capped_(const sdumc_rows_problem* probs, int32_t n, int32_t max_wg, void* stream);
// gemm_p3.hip: n <= 12 weights of the flat parameter buffer -> fragment-major bf16 planes (the B operand of sdumc_gemm_p3_nt)
int sdumc_p3_split_frag_multi_(const float* P, void* dst, const int64_t* src_off, const int64_t* dst_off, const int32_t* rows,
                               const int32_t* cols, int n, void* stream);
int sdumc_chain_transpose_(const float* src, float* dst, const int64_t* offs, const int32_t* outs, const int32_t* ins, int n,
                           void* stream);
}

// ---------------------------------------------------------------------------
// XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs (linear workgroup id % 8) and every XCD has its
// own 4 MB L2, so tiles that read the same operand panel should sit behind ONE L2 at about the same time: XCD x walks the
// contiguous tile range [x T/8, (x+1) T/8) in dispatch order.  L = linear workgroup id inside one z-plane of the grid (a
// constant offset of the real id only renames the XCDs), total = tiles in the plane; bijective for any total.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int xcd_tile(int L, int total) {
  const int q = total >> 3, r = total & 7;
  const int x = L & 7, s = L >> 3;
  return x * q + (x < r ? x : r) + s;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. SC'11).  Bit-identical to oracle/philox.py.
// ---------------------------------------------------------------------------
struct Philox4 {
  uint32_t w[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // (one 32 x 32 -> 64 multiply per word pair: v_mad_u64_u32 gives high and low half in ONE quarter-rate instruction where
    //  __umulhi + '*' took two -- the multiplies are what bounds the keep-bits kernels)
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
  return o;
}

// Resolved (device-side) view of an sdumc_dropout.
struct DropRT {
  uint32_t enabled, site, threshold, rows, qwidth, samples, sample0, call0, stream0, k0, k1;
  float scale;
  const uint8_t* bits;
};

__device__ __forceinline__ DropRT drop_resolve(const sdumc_dropout& d) {
  DropRT r;
  r.enabled = d.enabled;
  r.site = d.site;
  r.threshold = d.threshold;
  r.rows = d.rows ? d.rows : 1u;
  r.qwidth = (d.width + 3u) >> 2;   // ceil: a ragged last quad still costs one Philox call
  r.samples = d.samples ? d.samples : 1u;
  r.sample0 = d.sample0;
  r.stream0 = d.stream0;
  r.scale = d.scale;
  r.bits = d.bits;
  if (d.dev_state) {
    r.k0 = d.dev_state[0];
    r.k1 = d.dev_state[1];
    r.call0 = d.dev_state[2];
  } else {
    r.k0 = d.seed_lo;
    r.k1 = d.seed_hi;
    r.call0 = d.call0;
  }
  return r;
}

// The 4 multiplicative mask values of columns [4*cq, 4*cq+4) of virtual row `vrow`
// (row space [streams][samples][rows]).
__device__ __forceinline__ f32x4 drop_mask4(const DropRT& d, uint32_t vrow, uint32_t cq) {
  if (d.bits) {   // precomputed keep-bits (sdumc_dropout_bits)
    const uint32_t b = d.bits[(size_t)vrow * d.qwidth + cq];
    f32x4 m;
    m[0] = (b & 1u) ? d.scale : 0.f;
    m[1] = (b & 2u) ? d.scale : 0.f;
    m[2] = (b & 4u) ? d.scale : 0.f;
    m[3] = (b & 8u) ? d.scale : 0.f;
    return m;
  }
  const uint32_t v = vrow / d.rows;
  const uint32_t r = vrow - v * d.rows;
  const uint32_t s = v / d.samples;
  const uint32_t b = v - s * d.samples;
  const Philox4 p = philox4x32_10(r * d.qwidth + cq, d.sample0 + b, d.site, d.call0 + d.stream0 + s, d.k0, d.k1);
  f32x4 m;
  m[0] = p.w[0] >= d.threshold ? d.scale : 0.f;
  m[1] = p.w[1] >= d.threshold ? d.scale : 0.f;
  m[2] = p.w[2] >= d.threshold ? d.scale : 0.f;
  m[3] = p.w[3] >= d.threshold ? d.scale : 0.f;
  return m;
}

__device__ __forceinline__ float drop_mask1(const DropRT& d, uint32_t vrow, uint32_t col) {
  const f32x4 m = drop_mask4(d, vrow, col >> 2);
  return m[col & 3];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
