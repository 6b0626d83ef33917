// engine.hip — WengnetMOSEIMultViewsTextMissing forward / backward and the two-stream
// self-distillation step as a sequence of HIP kernel launches on one stream (host side of the C ABI).
//
//   sdumc_net_forward   model :275-370 for 1 stream, or for both streams of main :119,:131 at once
//   sdumc_net_backward  loss.backward() (main :149) through that network, hand-derived
//   sdumc_loss_backward main :137-148 (MSE x2, RMSE x3, RnC) value + gradient w.r.t. network outputs
//   sdumc_train_step    main :119-150 on one GPU: forward, losses, backward, Adam
//
// "Virtual batch": the two streams are stacked, V = streams*B rows, stream-major.  Every
// utterance-level layer then runs ONCE on V rows (the reference calls the same weights twice);
// the audio/video frame projections are computed once and shared by both streams (a_row_mod), only
// the dropout masks differ per stream (Philox call index = call0 + stream).
// No allocation, no synchronisation, no host copies: everything here is hipGraph-capturable.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "common.h"

extern "C" int sdumc_axpy2d(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int32_t rows, int32_t cols,
                            void* stream);

namespace {

constexpr int D = SDUMC_D, H = SDUMC_H, NQ = SDUMC_NQ, RD = SDUMC_RNC_DIM;

// dropout call sites in the reference's call order (oracle/sdumc_oracle.py SITE_*)
constexpr int SITE_IN[2][3] = {{0, 2, 4}, {21, 23, 25}};    // [fra|cross][modality] input dropout
constexpr int SITE_OUT[2][3] = {{1, 3, 5}, {22, 24, 26}};   // output dropout
constexpr int SITE_UMLP0 = 6, SITE_UMLP1 = 7;               // + 2*m
constexpr int SITE_ATT0 = 12, SITE_ATT1 = 13, SITE_QUERY = 14;
constexpr int SITE_CMLP0 = 27, SITE_CMLP1 = 28;             // + 2*m
constexpr int SITE_CATT0 = 33, SITE_CATT1 = 34;

#define RET(x)                 \
  do {                         \
    int _rc = (x);             \
    if (_rc != SDUMC_OK) return _rc; \
  } while (0)

// ------------------------------------------------------------------------------------------
// parameter layout
// ------------------------------------------------------------------------------------------
struct Lin {
  int64_t w = 0, b = 0;
  int out = 0, in = 0;
};
struct PEntry {
  std::string name;
  int64_t off;
  int rows, cols;  // cols = 0 for 1-D tensors
  bool live;
};
struct ParamMap {
  Lin frame[3], fra_proj[3], umlp0[3], umlp3[3], att0, att3, fc_att, query[7], ca_q[3], ca_in[3], cmlp0[3], cmlp3[3],
      catt0, catt3, cross_fc_att, fc_out_v, rnc0, rnc2;
  int64_t fra_ctx[3] = {0, 0, 0};
  int64_t early = 0, live = 0, total = 0;   // [0, early): utterance-level, [early, live): frame-level, [live, total): never trained
  std::vector<PEntry> table;
};

inline int64_t align4(int64_t n) { return (n + 3) & ~(int64_t)3; }

struct PBuilder {
  ParamMap& pm;
  int64_t cur = 0;
  bool live = true;
  int64_t add(const std::string& name, int rows, int cols) {
    const int64_t off = cur;
    pm.table.push_back({name, off, rows, cols, live});
    cur += align4((int64_t)rows * (cols ? cols : 1));
    return off;
  }
  Lin lin(const std::string& name, int out, int in) {
    Lin l;
    l.out = out;
    l.in = in;
    l.w = add(name + ".weight", out, in);
    l.b = add(name + ".bias", out, 0);
    return l;
  }
};

ParamMap build_params_uncached(int da, int dt, int dv);
// the layout depends on the three feature widths only: build it once per (da, dt, dv)
const ParamMap& build_params(int da, int dt, int dv) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int>, ParamMap> cache;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(da, dt, dv);
  auto it = cache.find(key);
  if (it == cache.end()) it = cache.emplace(key, build_params_uncached(da, dt, dv)).first;
  return it->second;
}

ParamMap build_params_uncached(int da, int dt, int dv) {
  ParamMap pm;
  PBuilder b{pm};
  static const char* MOD[3] = {"audio", "text", "video"};
  static const char* QN[7] = {"fused", "at", "tv", "av", "audio", "text", "video"};
  const int din[3] = {da, dt, dv};
  // Order inside the live block = order in which the backward FINISHES the gradients: the utterance-level layers
  // first ("early": final once the utterance-level backward is done), then the frame-level ones ("late": input_proj of
  // both attention sites, the FRA2UTT context vectors, frame_dim_reshape).  A data-parallel step can then all-reduce
  // the early slice while the frame-level backward (0.9 ms of GEMMs at C2) is still running.
  for (int m = 0; m < 3; ++m) {
    pm.umlp0[m] = b.lin(std::string(MOD[m]) + "_mlp.0", D, D);
    pm.umlp3[m] = b.lin(std::string(MOD[m]) + "_mlp.3", D, D);
  }
  pm.att0 = b.lin("attention_mlp.0", D, 3 * D);
  pm.att3 = b.lin("attention_mlp.3", D, D);
  pm.fc_att = b.lin("fc_att", 3, D);
  for (int i = 0; i < 7; ++i) pm.query[i] = b.lin(std::string("cross_") + QN[i] + "_query_mlp.0", D, D);
  for (int m = 0; m < 3; ++m) pm.ca_q[m] = b.lin("cross_att_fra2utt_" + std::to_string(m) + ".query_proj", D, D);
  for (int m = 0; m < 3; ++m) {
    pm.cmlp0[m] = b.lin(std::string("cross_") + MOD[m] + "_mlp.0", D, D);
    pm.cmlp3[m] = b.lin(std::string("cross_") + MOD[m] + "_mlp.3", H, D);
  }
  pm.catt0 = b.lin("cross_attention_mlp.0", D, NQ * H);
  pm.catt3 = b.lin("cross_attention_mlp.3", H, D);
  pm.cross_fc_att = b.lin("cross_fc_att", NQ, H);
  pm.fc_out_v = b.lin("fc_out_v", 1, H);
  pm.rnc0 = b.lin("orgin_linear_change.0", RD, H);
  pm.rnc2 = b.lin("orgin_linear_change.2", RD, RD);
  pm.early = b.cur;
  for (int m = 0; m < 3; ++m) pm.ca_in[m] = b.lin("cross_att_fra2utt_" + std::to_string(m) + ".input_proj", D, D);
  for (int m = 0; m < 3; ++m) {
    const std::string p = "fra2utt_" + std::to_string(m);
    pm.fra_ctx[m] = b.add(p + ".attention_context_vector", 1, D);
    pm.fra_proj[m] = b.lin(p + ".input_proj", D, D);
  }
  for (int m = 0; m < 3; ++m) pm.frame[m] = b.lin("frame_dim_reshape_" + std::to_string(m), D, din[m]);
  pm.live = b.cur;
  // parameters that exist for state_dict compatibility but never receive a gradient
  // (model :202-203, :242, :244, :257, :260)
  b.live = false;
  struct AE {
    const char* name;
    int dim, lat;
  } aes[2] = {{"missing_text_imagination_mlp", D, 128}, {"missing_cross_text_query_imagination_mlp", 128, 64}};
  for (const AE& a : aes) {
    const std::string p = a.name;
    b.lin(p + ".transition.0", a.dim, a.dim * 3);
    b.lin(p + ".transition.2", a.dim, a.dim);
    b.lin(p + ".encoder_0.0", a.lat, a.dim);
    b.lin(p + ".decoder_0.0", a.dim, a.lat);
  }
  b.lin("fc_out_e", 1, H);
  b.lin("fc_out_ev", 1, 1);
  b.add("prelu.weight", 6, 0);
  b.add("layer_normali.weight", D, 0);
  b.add("layer_normali.bias", D, 0);
  pm.total = b.cur;
  return pm;
}

// ------------------------------------------------------------------------------------------
// workspace plan (offsets in floats; every buffer 64-float = 256-byte aligned)
// ------------------------------------------------------------------------------------------
struct Seg {       // a run of virtual samples of one modality that share T and sit contiguously
  int s0;          // first stream in the run
  int V;           // virtual samples in the run
  int T;
  int x_samples;   // samples held by the x buffer (B when both streams share x, else V)
  int64_t x_off;   // offset of the x buffer
  int64_t row0;    // first virtual row (v*T + t) of the run inside the modality's row space
};

struct Plan {
  int B = 0, S = 0, V = 0;
  int64_t cur = 0;
  int T[3][2];
  int64_t x[3][2];          // projected features; audio/video: [m][0] only
  int64_t rows[3];          // virtual rows of modality m = sum over streams of B*T
  std::vector<Seg> segs[3];
  int64_t keys[2][3], attn[2][3], pooled[2][3];
  int64_t bits[2][3];       // input-dropout keep-bits, one byte per (virtual row, 4 channels)
  int64_t hpre, u1, u, att1, att2, alpha, qin, q, qp, ca_out, c1, c, h, e1, e2, beta, z, vals, r1, r;
  // backward
  int64_t dz[2][3], dxd[2][3], dx[3][2];
  int64_t d_hpre, d_u1, d_u, d_att1, d_att2, d_alpha, d_qin, d_q, d_qp, d_ca_out, d_c1, d_c, d_h, d_e1, d_e2, d_beta,
      d_z, d_r1, dq_fra;
  int64_t lens = 0;          // int32 [3][V]: valid frames per (modality, virtual sample) when sdumc_net_io.lengths is given
  int64_t wt = 0;            // transposed mirror of the utterance-level weights [0, early): chain.hip's forward layout
  // bf16-storage mode (sdumc_net_dims.bf16 == 2): x, keys, dz, dxd, dx are bf16 (their offsets above stay in FLOATS, the
  // buffers are half as long); plus the masked frames xd of each attention site and bf16 copies of the frame-level weights
  bool hf = false;
  int64_t xd[2][3] = {{0, 0, 0}, {0, 0, 0}};
  int64_t wh = 0, wht = 0;   // [live] bf16 each: weights as stored / transposed (input_proj only), parameter offsets
  int64_t tickets = 0;   // per-sample arrival counters of the attention kernels
  int64_t fold_ws[2][3] = {{0, 0, 0}, {0, 0, 0}};   // softmax partials of site (k, m), kept until the clustered stage behind them has combined them (fra_fold)
  // fp32 storage, features given as bf16 planes too (sdumc_net_io.*_p3; gemm_p3.hip): fragment-major planes of the frame-level weights
  // (refreshed at the head of each forward) and P3 copies of the projected frames (written by the frame projection's epilogue)
  int64_t dom[2][3] = {{0, 0, 0}, {0, 0, 0}};      // dout * out-dropout mask of attention site (k, m) ([V][nq][D]), written by its pooling backward (dxfold)
  bool dxfold[3] = {false, false, false};   // per modality: dxd = dz W + the pooling's own input gradient in ONE pass of the rows launch (no dxd write by the pooling backward)
  int64_t wb1 = 0;                     // bf16 storage: float offset of the fragment-major bf16 weights (wp3_frame / wp3_key: byte offsets inside it)
  int64_t wp3 = 0;                     // float offset of the weight-plane region
  int64_t wp3_frame[3] = {0, 0, 0};    // byte offsets inside it
  int64_t wp3_key[2][3] = {{0, 0, 0}, {0, 0, 0}};
  int64_t xp3[3][2] = {{0, 0}, {0, 0}, {0, 0}};      // float offsets: [rows][256] planes = 1536 bytes per row
  int64_t dq_ws = 0;        // per-chunk dq slabs of the grouped Cross_Attention pooling backward, kept until the clustered stage 7'-3' has summed them
                            // (their own allocation: with one lane every "lane" shares scratch[0], and early_keys() may use it in between)
  int64_t scratch[4] = {0, 0, 0, 0}, scratch_floats = 0;   // one scratch per lane (stream)
  int64_t gg_slab[2] = {0, 0}, gg_slab_floats[2] = {0, 0}; // partial-tile slabs of the grouped dW launches: [0] lane 3, [1] the frame dW
  int64_t alloc(int64_t n) {
    const int64_t o = cur;
    cur += (n + 63) & ~(int64_t)63;
    return o;
  }
};

int pick_splitk(int M, int N, int K, int groups) {
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128) * groups;
  if (tiles >= 256) return 1;
  long s = (384 + tiles - 1) / tiles;
  const long kmax = K / 256;
  if (s > kmax) s = kmax;
  return s < 1 ? 1 : (int)s;
}

size_t attnpool_bwd_ws_floats(int V, int T, int nq) { return (size_t)V * ((T + 63) / 64) * (nq * D + 16); }  // fwd needs +16/chunk

bool make_plan(const sdumc_net_dims& d, Plan& p) {
  if (d.B <= 0 || (d.streams != 1 && d.streams != 2) || d.Ta <= 0 || d.Tv <= 0 || d.Tt[0] <= 0) return false;
  if (d.streams == 2 && d.Tt[1] <= 0) return false;
  if (d.da <= 0 || d.dt <= 0 || d.dv <= 0) return false;
  if (d.bf16 == 2 && ((d.da | d.dt | d.dv) & 63)) return false;     // bf16 storage: feature widths in whole 64-element k-tiles
  p.B = d.B;
  p.S = d.streams;
  p.V = d.B * d.streams;
  p.hf = d.bf16 == 2;
  const int B = p.B, S = p.S, V = p.V;
  const int HS = p.hf ? 2 : 1;          // bf16 frame tensors take half the floats
  for (int s = 0; s < 2; ++s) {
    p.T[0][s] = d.Ta;
    p.T[1][s] = d.Tt[s < S ? s : 0];
    p.T[2][s] = d.Tv;
  }
  // x buffers; the text buffers of the two streams are adjacent so that equal T merges them
  p.x[0][0] = p.alloc((int64_t)B * d.Ta * D / HS);
  p.x[0][1] = p.x[0][0];
  p.x[2][0] = p.alloc((int64_t)B * d.Tv * D / HS);
  p.x[2][1] = p.x[2][0];
  {
    const int64_t n0 = (int64_t)B * p.T[1][0] * D / HS;
    const int64_t n1 = S == 2 ? (int64_t)B * p.T[1][1] * D / HS : 0;
    p.x[1][0] = p.alloc(n0 + n1);
    p.x[1][1] = p.x[1][0] + n0;
  }
  for (int m = 0; m < 3; ++m) {
    p.segs[m].clear();
    if (m != 1 || S == 1 || p.T[1][0] == p.T[1][1]) {
      Seg sg;
      sg.s0 = 0;
      sg.V = V;
      sg.T = p.T[m][0];
      sg.x_samples = (m == 1) ? V : B;
      sg.x_off = p.x[m][0];
      sg.row0 = 0;
      p.segs[m].push_back(sg);
      p.rows[m] = (int64_t)V * sg.T;
    } else {
      int64_t row0 = 0;
      for (int s = 0; s < 2; ++s) {
        Seg sg;
        sg.s0 = s;
        sg.V = B;
        sg.T = p.T[1][s];
        sg.x_samples = B;
        sg.x_off = p.x[1][s];
        sg.row0 = row0;
        p.segs[m].push_back(sg);
        row0 += (int64_t)B * sg.T;
      }
      p.rows[m] = row0;
    }
  }
  const int nq[2] = {1, NQ};
  for (int k = 0; k < 2; ++k)
    for (int m = 0; m < 3; ++m) {
      p.keys[k][m] = p.alloc(p.rows[m] * D / HS);
      if (p.hf && d.train) p.xd[k][m] = p.alloc(p.rows[m] * D / 2);
      p.bits[k][m] = p.alloc(p.rows[m] * (D / 4) / 4 + 1);
      p.attn[k][m] = p.alloc(p.rows[m] * nq[k]);
      p.pooled[k][m] = p.alloc((int64_t)V * nq[k] * D);
    }
  p.hpre = p.alloc(3LL * V * D);
  p.u1 = p.alloc(3LL * V * D);
  p.u = p.alloc(3LL * V * D);
  p.att1 = p.alloc((int64_t)V * D);
  p.att2 = p.alloc((int64_t)V * D);
  p.alpha = p.alloc((int64_t)V * 3);
  p.qin = p.alloc(7LL * V * D);
  p.q = p.alloc(7LL * V * D);
  p.qp = p.alloc(3LL * V * NQ * D);
  p.ca_out = p.alloc(3LL * V * NQ * D);
  p.c1 = p.alloc(3LL * V * NQ * D);
  p.c = p.alloc(3LL * V * NQ * H);
  p.h = p.alloc((int64_t)V * NQ * H);
  p.e1 = p.alloc((int64_t)V * D);
  p.e2 = p.alloc((int64_t)V * H);
  p.beta = p.alloc((int64_t)V * NQ);
  p.z = p.alloc((int64_t)V * H);
  p.vals = p.alloc(V);
  p.r1 = p.alloc((int64_t)V * RD);
  p.r = p.alloc((int64_t)V * RD);
  // backward
  for (int k = 0; k < 2; ++k)
    for (int m = 0; m < 3; ++m) {
      p.dz[k][m] = p.alloc(p.rows[m] * D / HS);
      p.dxd[k][m] = p.alloc(p.rows[m] * D / HS);
    }
  {
    // the rows launch adds the rank-nq pooling term as one more k-tile when a 64-row tile spans at most two samples (sdumc_rows_problem)
    // (decided per modality: a ragged epoch's text runs are often shorter than 63 frames, its audio / video runs hardly ever)
    for (int m = 0; m < 3; ++m) {
      bool ok = D == 256 && (d.bf16 == 0 || p.hf);
      for (const Seg& sg : p.segs[m]) ok = ok && (sg.T >= 63 || sg.T == 32) && (int64_t)sg.V * sg.T * D * 4 < 0x7FFF0000LL;
      p.dxfold[m] = ok;
      if (ok)
        for (int k = 0; k < 2; ++k) p.dom[k][m] = p.alloc((int64_t)V * (k == 0 ? 1 : NQ) * D);
    }
  }
  p.dx[0][0] = p.dx[0][1] = p.alloc((int64_t)B * d.Ta * D / HS);
  p.dx[2][0] = p.dx[2][1] = p.alloc((int64_t)B * d.Tv * D / HS);
  {      // (the text slot's two streams adjacent, like their x: one run of 2 B samples when the streams share T)
    const int64_t n0 = (int64_t)B * p.T[1][0] * D / HS;
    p.dx[1][0] = p.alloc(n0 + (S == 2 ? (int64_t)B * p.T[1][1] * D / HS : 0));
    p.dx[1][1] = S == 2 ? p.dx[1][0] + n0 : p.dx[1][0];
  }
  p.d_hpre = p.alloc(3LL * V * D);
  p.d_u1 = p.alloc(3LL * V * D);
  p.d_u = p.alloc(3LL * V * D);
  p.d_att1 = p.alloc((int64_t)V * D);
  p.d_att2 = p.alloc((int64_t)V * D);
  p.d_alpha = p.alloc((int64_t)V * 3);
  p.d_qin = p.alloc(7LL * V * D);
  p.d_q = p.alloc(7LL * V * D);
  p.d_qp = p.alloc(3LL * V * NQ * D);
  p.d_ca_out = p.alloc(3LL * V * NQ * D);
  p.d_c1 = p.alloc(3LL * V * NQ * D);
  p.d_c = p.alloc(3LL * V * NQ * H);
  p.d_h = p.alloc((int64_t)V * NQ * H);
  p.d_e1 = p.alloc((int64_t)V * D);
  p.d_e2 = p.alloc((int64_t)V * H);
  p.d_beta = p.alloc((int64_t)V * NQ);
  p.d_z = p.alloc((int64_t)V * H);
  p.d_r1 = p.alloc((int64_t)V * RD);
  p.dq_fra = p.alloc(3LL * V * D);
  p.lens = p.alloc(3LL * V);
  p.tickets = p.alloc(12LL * V + 16);    // attention sites [2][3] x 2V counters (sdumc_attnpool.tickets), zeroed by forward()
  for (int k = 0; k < 2; ++k)
    for (int m = 0; m < 3; ++m)
      p.fold_ws[k][m] = p.alloc((int64_t)(sdumc_attnpool_fwd_workspace_bytes(V, p.segs[m][0].T, nq[k]) / sizeof(float)));
  {
    int64_t sum = 0;
    for (int m = 0; m < 3; ++m)
      for (const Seg& sg : p.segs[m])
        sum += ((int64_t)(sdumc_attnpool_bwd_workspace_bytes(sg.V, sg.T, NQ) / sizeof(float)) + 63) / 64 * 64;
    p.dq_ws = p.alloc(sum);
  }
  if (d.bf16 == 2 && D == 256 && ((d.da | d.dt | d.dv) & 127) == 0) {      // fragment-major bf16 weights for sdumc_gemm_b1_nt
    const int din_[3] = {d.da, d.dt, d.dv};
    int64_t bytes = 0;
    for (int m = 0; m < 3; ++m) { p.wp3_frame[m] = bytes; bytes += (int64_t)D * din_[m] * 2; }
    for (int k = 0; k < 2; ++k)
      for (int m = 0; m < 3; ++m) { p.wp3_key[k][m] = bytes; bytes += (int64_t)D * D * 2; }
    p.wb1 = p.alloc(bytes / 4);
  }
  if (d.bf16 == 0 && D == 256 && ((d.da | d.dt | d.dv) & 63) == 0) {      // planes for sdumc_gemm_p3_nt (used when the caller passes feature planes)
    const int din_[3] = {d.da, d.dt, d.dv};
    int64_t bytes = 0;
    for (int m = 0; m < 3; ++m) { p.wp3_frame[m] = bytes; bytes += (int64_t)D * din_[m] * 6; }
    for (int k = 0; k < 2; ++k)
      for (int m = 0; m < 3; ++m) { p.wp3_key[k][m] = bytes; bytes += (int64_t)D * D * 6; }
    p.wp3 = p.alloc(bytes / 4);
    p.xp3[0][0] = p.xp3[0][1] = p.alloc((int64_t)B * d.Ta * D * 6 / 4);
    p.xp3[2][0] = p.xp3[2][1] = p.alloc((int64_t)B * d.Tv * D * 6 / 4);
    {
      const int64_t n0 = (int64_t)B * p.T[1][0] * D * 6 / 4;
      const int64_t n1 = S == 2 ? (int64_t)B * p.T[1][1] * D * 6 / 4 : 0;
      p.xp3[1][0] = p.alloc(n0 + n1);
      p.xp3[1][1] = p.xp3[1][0] + n0;
    }
  }
  p.wt = p.alloc(build_params(d.da, d.dt, d.dv).live);   // transposed mirror: utterance-level layers (chain) + the six input_proj
  if (p.hf) {
    const int64_t live = build_params(d.da, d.dt, d.dv).live;
    p.wh = p.alloc(live / 2 + 8);
    p.wht = p.alloc(live / 2 + 8);
  }
  // scratch shared by split-K slabs, column-sum partials and the attention-pool dq slabs
  int64_t sc = 1 << 16;
  const int din[3] = {d.da, d.dt, d.dv};
  for (int m = 0; m < 3; ++m) {
    for (int s = 0; s < S; ++s) {
      const int K = B * p.T[m][s];
      sc = std::max<int64_t>(sc, (int64_t)pick_splitk(D, din[m], K, 1) * D * din[m]);
    }
    sc = std::max<int64_t>(sc, (int64_t)pick_splitk(D, D, (int)p.rows[m], 1) * D * D);
    for (const Seg& sg : p.segs[m]) sc = std::max<int64_t>(sc, (int64_t)attnpool_bwd_ws_floats(sg.V, sg.T, NQ));
    sc = std::max<int64_t>(sc, (int64_t)((p.rows[m] + 511) / 512 + 1) * D);
  }
  {   // the grouped Cross_Attention launches carve ONE lane's scratch into per-site workspaces
    int64_t sum = 0;
    for (int m = 0; m < 3; ++m)
      for (const Seg& sg : p.segs[m])
        sum += ((int64_t)(sdumc_attnpool_fwd_workspace_bytes(sg.V, sg.T, NQ) / sizeof(float)) + 63) / 64 * 64;
    sc = std::max<int64_t>(sc, sum);
  }
  sc = std::max<int64_t>(sc, (int64_t)8 * 4 * D * NQ * H);  // grouped utterance-level split-K upper bound
  sc = std::max<int64_t>(sc, (int64_t)13 << 20);            // auto split-K: <= ~(320 + tiles) slabs of 64 KiB
  p.scratch_floats = sc;
  for (int l = 0; l < 4; ++l) p.scratch[l] = p.alloc(sc);
  {   // grouped weight-gradient launches (gemm_group.hip): at most one 256 x 128 tile per 128 output columns of every layer
    const int utt_tiles = 160;                                            // upper bound over the utterance-level layers
    const int key_tiles = 6 * 2 * 2;                                      // six input_proj layers, up to two runs each
    const int frame_tiles = (d.da + 127) / 128 + 2 * ((d.dt + 127) / 128) + (d.dv + 127) / 128;
    p.gg_slab_floats[0] = (int64_t)(sdumc_gg_slab_bytes_(utt_tiles + key_tiles) / sizeof(float));
    p.gg_slab_floats[1] = (int64_t)(sdumc_gg_slab_bytes_(frame_tiles) / sizeof(float));
    for (int i = 0; i < 2; ++i) p.gg_slab[i] = p.alloc(p.gg_slab_floats[i]);
  }
  return true;
}

// ------------------------------------------------------------------------------------------
// Lanes: lane 0 is the caller's stream; lanes 1, 2 are internal side streams forked from / joined to it with
// events.  Independent branches (the three per-modality chains; the dW GEMMs, which are off the dX critical
// path) are issued on different lanes: eagerly they overlap on the GPU, under hipGraph capture they become
// parallel branches of the graph.  That fills the partially empty last dispatch round of the big kernels and
// hides the launch-bound small ones.  Each lane has its own scratch, so concurrent kernels never share slabs.
// A lane set = the internal streams and the event ring one caller context issues its branches on.  There is one default set
// per DEVICE (created on first use, under a lock), and callers that drive several steps from several host threads create
// their own with sdumc_ctx_create and pass it in sdumc_net_io.ctx: nothing here is shared between two lane sets, and the
// event ring hands its entries out atomically, so two threads on distinct contexts (or distinct devices) never see each
// other's events.  (One context is still for one thread at a time: its lanes are ordered streams.)
struct LaneSet {
  hipStream_t s[2] = {nullptr, nullptr};
  hipStream_t bg = nullptr;   // lane 3: dW batches, keep-bits, the forward Cross_Attention key GEMMs
  static constexpr unsigned NEV = 256;
  hipEvent_t ev[NEV];
  std::atomic<unsigned> next{0};
  int device = -1;
  bool ok = false;
  // per-context schedule options (sdumc_ctx_set_option); -1 = the process-wide default (sdumc_set_concurrency / _background_lane /
  // _chain_cluster, kept as the defaults of contexts that set nothing)
  int opt_concurrency = -1, opt_background = -1, opt_chain_cluster = -1;
  int opt_split = -1;   // SDUMC_OPT_SPLIT: which fp32 GEMM families multiply on the bf16 matrix pipe inside this context's calls
};
// Debug timeline (tools/step_marks.py): sdumc_debug_marks(1) makes the step record an event on the caller's stream at a few
// fixed points; sdumc_debug_marks_read returns their times since mark 0.  Process-wide, single-threaded use only.
constexpr int kMarks = 48;
bool g_marks_on = false;
hipEvent_t g_marks[kMarks] = {};
bool g_mark_set[kMarks] = {};
inline void mark(hipStream_t st, int id) {
  if (!g_marks_on) return;
  if (!g_marks[id] && hipEventCreate(&g_marks[id]) != hipSuccess) return;
  g_mark_set[id] = hipEventRecord(g_marks[id], st) == hipSuccess;
}
bool g_concurrency = true;   // sdumc_set_concurrency(0): everything on the caller's stream (profiling)
int g_background = 3;      // 0 off; 2 forward only (+1.0 % per step over 0); 3 (default) = 2 + the AUDIO Cross_Attention key-projection
                           // backward early on lane 3 (it shortens the longest frame-level chain; more modalities early measured
                           // slower in rounds 1 and 2: profiles/README.md)

// The side lanes are HIGH-priority streams.  The HIP runtime multiplexes the streams of one priority class onto
// GPU_MAX_HW_QUEUES (default 4) hardware queues, handing a new stream the least-used queue: once other components
// hold normal-priority streams (an RCCL communicator created before these lanes; torch's stream pool) two lanes can
// land on ONE hardware queue and the three modality chains serialise -- measured on MI355X: 2.59-2.63 ms per step
// instead of 2.18 whenever the process group had been initialised first (tools/rccl_queue_probe.py), and 2.18 again
// with GPU_MAX_HW_QUEUES=8 or with the lanes in their own priority class (2.17-2.20 ms in both orders).
// (Measured and rejected for lane 3: a stream confined to 7/8 or 1/2 of the CUs with hipExtStreamCreateWithCUMask, 128x128
// tiles on it, the lowest priority, a fourth side stream of the same class: profiles/README.md.)
bool create_lanes(LaneSet& S) {
  if (hipGetDevice(&S.device) != hipSuccess) return false;
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return false;
  (void)least;
  const int prio = greatest;
  for (int i = 0; i < 2; ++i)
    if (hipStreamCreateWithPriority(&S.s[i], hipStreamNonBlocking, prio) != hipSuccess) return false;
  for (unsigned i = 0; i < LaneSet::NEV; ++i)
    if (hipEventCreateWithFlags(&S.ev[i], hipEventDisableTiming) != hipSuccess) return false;
  // (lane 3 at the lowest priority, so that the grouped weight-gradient launches would only fill what the critical lanes
  //  leave: measured 1.877 vs 1.784 ms per step -- the launches then start late and end up as the step's tail)
  if (hipStreamCreateWithPriority(&S.bg, hipStreamNonBlocking, prio) != hipSuccess) return false;
  S.ok = true;
  return true;
}
void destroy_lanes(LaneSet& S) {
  for (int i = 0; i < 2; ++i)
    if (S.s[i]) { (void)sdumc_chain_cluster_forget_stream_(S.s[i]); (void)hipStreamDestroy(S.s[i]); }
  if (S.bg) { (void)sdumc_chain_cluster_forget_stream_(S.bg); (void)hipStreamDestroy(S.bg); }
  if (S.ok)
    for (unsigned i = 0; i < LaneSet::NEV; ++i) (void)hipEventDestroy(S.ev[i]);
  S.ok = false;
}
// the default lane set of the CURRENT device; created outside any capture (the *_workspace_bytes queries every caller
// makes first call this)
LaneSet* default_lanes() {
  static std::mutex mu;
  static std::map<int, LaneSet*> per_device;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  auto it = per_device.find(dev);
  if (it != per_device.end()) return it->second;
  LaneSet* S = new LaneSet();
  if (!create_lanes(*S)) { destroy_lanes(*S); delete S; S = nullptr; }
  per_device[dev] = S;
  return S;
}
void ensure_side_streams() { (void)default_lanes(); }
}  // namespace
extern "C" int sdumc_preload_gemm_f32_(void);
extern "C" int sdumc_preload_gemm_wide_(void);
extern "C" int sdumc_preload_gemm_p3_(void);
extern "C" int sdumc_preload_gemm_b1_(void);
extern "C" int sdumc_preload_gemm_group_(void);
extern "C" int sdumc_preload_gemm_rows_(void);
extern "C" int sdumc_preload_gemm_bf16_(void);
extern "C" int sdumc_preload_attn_pool_(void);
extern "C" int sdumc_preload_elementwise_(void);
extern "C" int sdumc_preload_loss_(void);
extern "C" int sdumc_preload_adam_(void);
extern "C" int sdumc_preload_chain_(void);
extern "C" int sdumc_preload_chain_cluster_(void);
extern "C" int sdumc_preload_transformer_(void);
namespace {
__global__ void sdumc_preload_engine_kernel() {}
// every code object of the library loaded on the current device, once per device (see the note at the end of any kernel source);
// called from the *_workspace_bytes queries every caller makes first -- never under stream capture, never inside a step
void preload_code_objects() {
  static sdumc_dev_once once;
  (void)sdumc_once_per_device(once, [] {
    hipFuncAttributes a;
    bool ok = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&sdumc_preload_engine_kernel)) == hipSuccess;
    ok = sdumc_preload_gemm_f32_() == SDUMC_OK && ok;
    ok = sdumc_preload_gemm_wide_() == SDUMC_OK && ok;
    ok = sdumc_preload_gemm_p3_() == SDUMC_OK && ok;
    ok = sdumc_preload_gemm_b1_() == SDUMC_OK && ok;
    ok = sdumc_preload_gemm_group_() == SDUMC_OK && ok;
    ok = sdumc_preload_gemm_rows_() == SDUMC_OK && ok;
    ok = sdumc_preload_gemm_bf16_() == SDUMC_OK && ok;
    ok = sdumc_preload_attn_pool_() == SDUMC_OK && ok;
    ok = sdumc_preload_elementwise_() == SDUMC_OK && ok;
    ok = sdumc_preload_loss_() == SDUMC_OK && ok;
    ok = sdumc_preload_adam_() == SDUMC_OK && ok;
    ok = sdumc_preload_chain_() == SDUMC_OK && ok;
    ok = sdumc_preload_chain_cluster_() == SDUMC_OK && ok;
    ok = sdumc_preload_transformer_() == SDUMC_OK && ok;
    return ok;
  });
}

// the context's split option for the duration of one network-level call (thread-local: the launchers ask sdumc_split_on_)
struct SplitScope {
  int prev = -1;
  bool on = false;
  explicit SplitScope(const sdumc_net_io* io) {
    const LaneSet* S = !io ? nullptr : (io->ctx ? static_cast<const LaneSet*>(io->ctx) : default_lanes());
    if (S && S->opt_split >= 0) { prev = sdumc_split_scope_(S->opt_split); on = true; }
  }
  ~SplitScope() { if (on) sdumc_split_scope_(prev); }
  SplitScope(const SplitScope&) = delete;
  SplitScope& operator=(const SplitScope&) = delete;
};

struct Ctx {
  const sdumc_net_dims& d;
  const sdumc_net_io& io;
  mutable hipStream_t st;   // stream of the current lane
  const ParamMap& pm;
  Plan pl;
  float* W;     // workspace base
  float* P;     // parameters
  float* G;     // gradient bucket (backward only)
  hipStream_t sts[4] = {nullptr, nullptr, nullptr, nullptr};
  LaneSet* lanes = nullptr;   // io.ctx (caller-owned) or the device's default set
  bool bg = false;   // the Cross_Attention-site key projections are issued on lane 3 (background)
  bool capturing = false;   // the caller's stream is under hipGraph capture: only the plain three-lane fork/join pattern is used
  int chain_cluster_opt = -1;   // the context's SDUMC_OPT_CHAIN_CLUSTER (-1: the process-wide switch decides)
  mutable const float* dq_part[3] = {nullptr, nullptr, nullptr};   // backward: the Cross_Attention sites' dq slabs left for stage A (fra_fold)
  mutable int dq_nchunk[3] = {0, 0, 0};
  int bgb = 0;       // bit m: the Cross_Attention key-projection BACKWARD of modality m runs early, on lane 3, beside steps 7'-3'
  mutable float* scr = nullptr;   // scratch of the current lane
  bool multi = false;
  // weight-gradient GEMMs of the utterance-level layers, queued by lin_bwd* and issued in batches on lane 3 (flush_dw)
  mutable std::vector<sdumc_gemm> deferred;
  // the same products as problems of ONE persistent launch (gemm_group.hip): everything queued here is issued by the next flush_dw
  mutable std::vector<sdumc_gg_problem> gg;
  mutable std::vector<sdumc_gg_problem> ggh;     // ... on bf16 operands (bf16-storage mode: sdumc_gemm_group_tn_bf16)
  float* p(int64_t off) const { return W + off; }
  // bf16 buffers: `off` is the buffer's offset in floats, `elems` an element offset inside it
  unsigned short* ph(int64_t off, int64_t elems = 0) const { return reinterpret_cast<unsigned short*>(W + off) + elems; }
  bool h() const { return pl.hf; }
  void init_lanes() {
    lanes = io.ctx ? static_cast<LaneSet*>(io.ctx) : default_lanes();
    static LaneSet none;
    const LaneSet& S = lanes ? *lanes : none;
    sts[0] = st;
    const int background = S.opt_background >= 0 ? S.opt_background : g_background;
    multi = S.ok && (S.opt_concurrency >= 0 ? S.opt_concurrency != 0 : g_concurrency);
    bg = background != 0;                 // the launch decomposition is the same with and without real streams
    bgb = background == 3 ? 1 : 0;        // 3: the audio modality's
    chain_cluster_opt = S.opt_chain_cluster;
    {   // under hipGraph capture the extra lane-3 dependencies (three lanes -> lane 3 -> lane 0) make hipStreamEndCapture
        // segfault on this stack (ROCm 7.0 runtime inside torch 2.10): captured steps keep the grouped launches
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) { bg = false; bgb = 0; capturing = true; }
    }
    sts[1] = multi ? S.s[0] : st;
    sts[2] = multi ? S.s[1] : st;
    sts[3] = multi ? S.bg : st;   // lane 3: deferred dW batches (and, with the background option, the Cross_Attention keys)
    use(0);
  }
  void use(int lane) const {
    st = sts[lane];
    scr = W + pl.scratch[multi ? lane : 0];
  }
};

hipEvent_t next_event(const Ctx& c) {
  LaneSet& S = *c.lanes;       // only reached when lanes exist (every caller is behind a sts[a] != sts[b] test)
  return S.ev[S.next.fetch_add(1, std::memory_order_relaxed) % LaneSet::NEV];
}
// lane `from` -> lane `to`: everything issued on `to` after this waits for what `from` has issued so far
int link(const Ctx& c, int from, int to) {
  if (c.sts[from] == c.sts[to]) return SDUMC_OK;
  hipEvent_t e = next_event(c);
  if (hipEventRecord(e, c.sts[from]) != hipSuccess) return SDUMC_ELAUNCH;
  if (hipStreamWaitEvent(c.sts[to], e, 0) != hipSuccess) return SDUMC_ELAUNCH;
  return SDUMC_OK;
}
int fork_all(const Ctx& c) {
  RET(link(c, 0, 1));
  return link(c, 0, 2);
}
int join_all(const Ctx& c) {
  RET(link(c, 1, 0));
  return link(c, 2, 0);
}
struct LaneMap {
  // audio (the heaviest chain) -> side lane 2, text -> side lane 1, video -> the caller's stream.  The side lanes are
  // high-priority streams, so this gives the longest chain dispatch preference: 29.87 k samples/s against 29.72 k with audio on
  // the caller's stream ({0, 2, 1}), the better of the two in each of four alternations (six permutations tried)
  int v[3] = {2, 1, 0};
  int operator[](int m) const { return v[m]; }
};
const LaneMap LANE_OF;

sdumc_dropout mkdrop(const Ctx& c, int site, double prob, int rows, int width, int stream0 = 0) {
  sdumc_dropout r;
  memset(&r, 0, sizeof(r));
  r.enabled = c.d.train ? 1u : 0u;
  r.site = (uint32_t)site;
  r.threshold = (uint32_t)(uint64_t)(prob * 4294967296.0);
  r.scale = 1.0f / (1.0f - (float)prob);
  r.rows = (uint32_t)rows;
  r.width = (uint32_t)width;
  r.samples = (uint32_t)c.d.B;
  r.sample0 = (uint32_t)c.d.sample0;
  r.stream0 = (uint32_t)stream0;
  r.dev_state = c.io.rng_state;
  return r;
}

// sdumc_net_io.bits_next: TWO sets of [tag 64 B][site 0: audio, text, video rows x 64 B][site 1: ...]: the keep-bits this call reads
// (set bits_phase & 1) and the ones it fills for the next call (the other set).  (k = 2: the size of a set.)
int64_t bits_next_off(const Plan& pl, int k, int m) {
  int64_t off = 64;
  for (int kk = 0; kk < 2; ++kk)
    for (int mm = 0; mm < 3; ++mm) {
      if (kk == k && mm == m) return off;
      off += pl.rows[mm] * (D / 4);
    }
  return (off + 255) & ~(int64_t)255;
}
bool bits_pregen(const Ctx& c) { return c.io.bits_next != nullptr && c.d.train && !c.h() && c.d.p_frame > 0.0; }
// (a buffer sized for the run's largest dims keeps its two sets half the buffer apart whatever the call's own shape is)
int64_t bits_set_stride(const Ctx& c) { return c.io.bits_next_bytes ? (int64_t)((c.io.bits_next_bytes / 2) & ~(size_t)255) : bits_next_off(c.pl, 2, 0); }
uint8_t* bits_set(const Ctx& c, int other) {
  return static_cast<uint8_t*>(c.io.bits_next) + (((c.io.bits_phase & 1) ^ other) ? bits_set_stride(c) : 0);
}
// the shape words of a keep-bits tag (elementwise.hip): a set laid out for another batch shape or shard offset is a foreign set
sdumc_bits_shape bits_shape(const sdumc_net_dims& d) {
  sdumc_bits_shape sh;
  sh.w[0] = (uint32_t)d.B;
  sh.w[1] = (uint32_t)d.sample0;
  sh.w[2] = (uint32_t)d.Ta;
  sh.w[3] = (uint32_t)d.Tv;
  sh.w[4] = (uint32_t)d.Tt[0];
  sh.w[5] = (uint32_t)(d.streams == 2 ? d.Tt[1] : 0);
  return sh;
}
// where the keep-bits of attention site (k, m) live for this call
uint8_t* bits_ptr(const Ctx& c, int k, int m) {
  return bits_pregen(c) ? bits_set(c, 0) + bits_next_off(c.pl, k, m) : reinterpret_cast<uint8_t*>(c.p(c.pl.bits[k][m]));
}
// the input dropout of attention site (k, m) over run `sg`, with its precomputed keep-bits in train mode
struct Seg;
sdumc_dropout in_drop(const Ctx& c, int k, int m, int T, int s0, int64_t row0) {
  sdumc_dropout d = mkdrop(c, SITE_IN[k][m], c.d.p_frame, T, D, s0);
  if (d.enabled) d.bits = bits_ptr(c, k, m) + row0 * (D / 4);
  return d;
}

sdumc_gemm G_(int layout, int M, int N, int K, int groups = 1) {
  sdumc_gemm g;
  memset(&g, 0, sizeof(g));
  g.layout = layout;
  g.M = M;
  g.N = N;
  g.K = K;
  g.groups = groups;
  return g;
}

int run(const Ctx& c, sdumc_gemm& g) {
  g.workspace = c.scr;
  g.workspace_bytes = (size_t)c.pl.scratch_floats * sizeof(float);
  return sdumc_gemm_f32(&g, c.st);
}

// y = act(x W^T + b) (+ dropout), single group
int lin_fwd(const Ctx& c, const Lin& L, const float* x, int lda, int M, float* y, int ldc, int act,
            const sdumc_dropout* drop, bool bf16 = false, int tile = 0, int splitk = 0) {
  sdumc_gemm g = G_(SDUMC_NT, M, L.out, L.in);
  g.tile = tile;
  g.splitk = splitk;
  g.bf16 = bf16 && (lda % 4 == 0) && (L.in % 4 == 0) ? 1 : 0;
  g.A[0] = x;
  g.lda = lda;
  g.B[0] = c.P + L.w;
  g.ldb = L.in;
  g.bias[0] = c.P + L.b;
  g.C[0] = y;
  g.ldc = ldc;
  g.act = act;
  if (drop) g.c_drop = *drop;
  return run(c, g);
}

int colsum(const Ctx& c, const float* a, int64_t rows, int cols, int lda, float* out, int accumulate) {
  return sdumc_colsum(a, rows, cols, lda, out, accumulate, c.scr, c.st);
}

// Weight gradients are off every critical path (only Adam and a data-parallel all-reduce read them): lin_bwd* queue them and
// flush_dw issues what is queued on lane 3 -- as ONE persistent launch + one reduce (gemm_group.hip) -- ordered after everything
// lane 0 has issued so far.
// Why batches: every hipEventRecord on a stream costs its NEXT kernel ~12 us of bubble on this stack (kernel trace of the
// step: the dX chain ran at one 9 us kernel per 26-30 us while each layer forked its own dW to a side lane, against
// back-to-back kernels in the link-free forward chain), so the chain records an event at three points instead of at
// every layer.  Every operand of a queued GEMM (dz, the saved input) is final when it is queued and is not written
// again before the end of the backward pass, so running it later is safe.
// SDUMC_DW_GROUP=0 restores one split-K GEMM per layer (the round-2 schedule, kept for A/B measurements and as the path of
// shapes the grouped kernel does not take)
bool gg_on() {
  static const bool on = [] {
    const char* e = getenv("SDUMC_DW_GROUP");
    return !(e && e[0] == '0');
  }();
  return on;
}
// the queued weight-gradient GEMMs go through ONE persistent launch (gemm_group.hip)
// (bf16-storage mode too: its utterance-level operands are fp32; ONE launch behind the frame-level pooling backward -- 1.087 vs
//  1.107 ms per step against the fifteen per-layer launches, which end up as lane 3's tail)
bool gg_utt(const Ctx& c) { (void)c; return gg_on(); }
// ... and so do the frame-level ones (input_proj of both attention sites, frame_dim_reshape), on fp32 or on bf16 storage
// (not in the operand-rounding mode bf16 = 1, whose products round fp32 operands while staging them)
bool gg_frame(const Ctx& c) { return gg_on() && (c.h() || c.d.bf16 == 0); }

// The input gradients dxd += dz W of the frame-level part go through the persistent B-stationary launch (gemm_rows.hip), one
// launch per modality on the modality's lane (SDUMC_ROWS=0: the 64x64 NN kernel).  Measured at C2, fp32 (tools/rows_bench.py and
// bench.py, alternated on one box): the five sites 138 us at 113 TF against 219 us at 71 TF alone; inside the step 1.737-1.749
// against 1.748-1.753 ms -- the phase is a sequence of chip-filling launches either way.  Also measured and dropped: all
// modalities in ONE launch on the caller's stream behind the three pooling backwards (1.768-1.774: the grouped dW launch then
// queues behind it with its small leading GEMMs starved); the Cross_Attention key projections of the forward pass through the
// masked variant of the same kernel (72 vs 71 us for the audio site alone, 37 vs 51 video; step 1.780-1.791: an exclusive
// persistent launch on the background lane holds the CUs the three modality lanes' critical kernels are waiting for).
bool rows_on() {
  static const bool on = [] { const char* e = getenv("SDUMC_ROWS"); return !(e && e[0] == '0'); }();
  return on;
}
bool rows_ok(const Ctx& c) { return D == 256 && !c.h() && c.d.bf16 == 0; }
// dxd of a site leaves the rows launch in one pass: dz W + sum_i attn_i * (dout_i * mask) (the pooling backward then writes no dxd and
// this launch reads none back: -2 KB of HBM traffic per virtual row and site).  SDUMC_DXFOLD=0: A/B against the two-pass form.
// ... and the mask-sum of the input dropouts as well: every site's launch adds keep . dxd of its streams straight into dx (the
// Cross_Attention site writes, the FRA2UTT site adds onto it) -- dxd is never in memory and there is no mask-sum launch.
// SDUMC_DXFOLD=1: the pooling term only.
// bf16 storage: the same through gr_bf16_kernel's folding variants (no pooling-term-only form there).
int dxfold_level() {
  static const int on = [] { const char* e = getenv("SDUMC_DXFOLD"); return e ? atoi(e) : 2; }();
  return on;
}
bool dxfold(const Ctx& c, int m) {
  if (!c.pl.dxfold[m] || !rows_on()) return false;
  if (c.h()) return dxfold_level() >= 2;
  return dxfold_level() >= 1 && rows_ok(c) && sdumc_split_on_(SDUMC_SPLIT_ROWS);
}
bool dxsum(const Ctx& c, int m) { return dxfold_level() >= 2 && dxfold(c, m); }

// one group of a queued TN descriptor as a problem of the grouped launch; false = the grouped kernel does not take it
bool gg_from_gemm(const sdumc_gemm& g, int grp, sdumc_gg_problem& q) {
  if (g.layout != SDUMC_TN || g.bf16 || g.batch > 1 || g.a_drop.enabled || g.b_drop.enabled || g.c_drop.enabled || g.a_row_mod ||
      g.act != SDUMC_ACT_NONE || g.bias[grp] || g.c_mask_y[grp])
    return false;
  if ((g.M & 3) || (g.N & 3) || (g.lda & 3) || (g.ldb & 3) || g.M < 4 || g.N < 4) return false;
  if ((reinterpret_cast<uintptr_t>(g.A[grp]) | reinterpret_cast<uintptr_t>(g.B[grp])) & 15) return false;
  if (g.b_row_mod > 0 && g.b_row_mod < 16) return false;
  memset(&q, 0, sizeof(q));
  q.A[0] = g.A[grp];
  q.B[0] = g.B[grp];
  q.K[0] = g.K;
  q.b_row_mod[0] = g.b_row_mod;
  q.C = g.C[grp];
  q.colsum_a = g.colsum_a[grp];
  q.M = g.M;
  q.N = g.N;
  q.lda = g.lda;
  q.ldb = g.ldb;
  q.ldc = g.ldc;
  q.b_scale = 1.f;
  q.accumulate = g.accumulate;
  return true;
}

// one grouped launch; a problem list whose partial-tile slots do not fit the planned slab (the plan sizes it from upper bounds on the
// layers' output tiles) is issued in halves instead of failing the step
int gg_launch(const Ctx& c, const sdumc_gg_problem* p, int n, bool hf, int slab) {
  void* ws = c.p(c.pl.gg_slab[slab]);
  const size_t bytes = (size_t)c.pl.gg_slab_floats[slab] * sizeof(float);
  const int rc = hf ? sdumc_gemm_group_tn_bf16(p, n, ws, bytes, c.st) : sdumc_gemm_group_tn(p, n, ws, bytes, c.st);
  if (rc != SDUMC_ENOMEM || n < 2) return rc;
  RET(gg_launch(c, p, n / 2, hf, slab));
  return gg_launch(c, p + n / 2, n - n / 2, hf, slab);
}

// issues everything queued (c.deferred, c.gg) on `lane`, ordered after what lane `after` has issued so far
int flush_dw_on(const Ctx& c, int after, int lane, int slab) {
  if (c.deferred.empty() && c.gg.empty() && c.ggh.empty()) return SDUMC_OK;
  RET(link(c, after, lane));
  c.use(lane);
  const bool grouped = gg_utt(c);
  std::vector<sdumc_gemm> rest;      // what the grouped launch does not take
  for (sdumc_gemm& g : c.deferred) {
    bool taken = grouped;
    if (grouped) {
      const size_t n0 = c.gg.size();
      for (int grp = 0; grp < g.groups && taken; ++grp) {
        sdumc_gg_problem q;
        taken = gg_from_gemm(g, grp, q);
        if (taken) c.gg.push_back(q);
      }
      if (!taken) c.gg.resize(n0);
    }
    // (fc_att, cross_fc_att, fc_out_v: 3 / 7 / 1 output rows.  They stay IN FRONT of the grouped launch: behind it -- so that the
    //  persistent launch starts 11-41 us x 3 earlier -- measured 1.684-1.687 vs 1.673-1.680 ms fp32, 0.954-0.956 vs 0.952-0.957 bf16)
    if (!taken) rest.push_back(g);
  }
  // (round 5: the three of them -- different shapes, four to eight workgroups each -- in ONE launch when the small-problem kernel takes
  //  them, in bf16 storage only: the grouped launch behind them then starts ~20 us earlier, which pays where the dX launches beside it
  //  are short (bf16 0.8475-0.8488 against 0.8607-0.8654 ms) and costs where they are not (fp32 1.3144-1.3151 against 1.2882-1.2907:
  //  the persistent launch takes the CUs the dX chain is waiting for -- the same sign as moving the three behind it did in round 4))
  if (c.h() && rest.size() >= 2 && rest.size() <= 3 && sdumc_gemm_small_tn_multi_(rest.data(), (int)rest.size(), c.st) == SDUMC_OK) rest.clear();
  for (sdumc_gemm& g : rest) RET(run(c, g));
  c.deferred.clear();
  if (!c.gg.empty()) {
    const int rc = gg_launch(c, c.gg.data(), (int)c.gg.size(), false, slab);
    c.gg.clear();
    if (rc != SDUMC_OK) return rc;
  }
  if (!c.ggh.empty()) {      // (same slab: this launch is ordered behind the previous one's reduce)
    const int rc = gg_launch(c, c.ggh.data(), (int)c.ggh.size(), true, slab);
    c.ggh.clear();
    if (rc != SDUMC_OK) return rc;
  }
  c.use(after);
  return SDUMC_OK;
}
int flush_dw(const Ctx& c) { return flush_dw_on(c, 0, 3, 0); }
// (measured and dropped, round 5: the three weight gradients the grouped launch refuses -- fc_att, cross_fc_att, fc_out_v: 3 / 7 / 1
//  output rows, four-workgroup launches that sit in front of the first grouped launch on lane 3 and take 13-44 us each there --
//  issued as soon as they are queued, on a side lane that idles through the utterance-level backward: 1.402-1.409 against 1.374-1.388 ms
//  fp32, 0.961-0.969 against 0.948-0.961 bf16; the two extra event records on the caller's stream cost the latency-bound chain
//  more than the grouped launch's earlier start gives back)

// backward of y = act(x W^T + b) given dzv = gradient w.r.t. the pre-activation, [M, L.out] with ld lddz:
//   dW = dz^T x, db = colsum(dz), dx (=|+=) dz W
//   dx_y != nullptr: x itself is the saved output of a Linear->ReLU->Dropout layer; the dX GEMM's epilogue then
//   multiplies dx by [dx_y > 0] * dx_scale, i.e. hands back the gradient w.r.t. that layer's PRE-activation
int lin_bwd(const Ctx& c, const Lin& L, const float* dzv, int lddz, const float* x, int ldx, int M, float* dx, int lddx,
            int dx_accumulate, const float* dx_y = nullptr, float dx_scale = 1.f) {
  sdumc_gemm gw = G_(SDUMC_TN, L.out, L.in, M);
  gw.A[0] = dzv;
  gw.lda = lddz;
  gw.B[0] = x;
  gw.ldb = ldx;
  gw.C[0] = c.G + L.w;
  gw.ldc = L.in;
  gw.colsum_a[0] = c.G + L.b;   // db rides along with the staging of dz
  c.deferred.push_back(gw);     // dW is off the dX critical path: issued later, in a batch, on lane 3 (flush_dw)
  if (dx) {
    sdumc_gemm gx = G_(SDUMC_NN, M, L.in, L.out);
    gx.A[0] = dzv;
    gx.lda = lddz;
    gx.B[0] = c.P + L.w;
    gx.ldb = L.in;
    gx.C[0] = dx;
    gx.ldc = lddx;
    gx.accumulate = dx_accumulate;
    gx.c_mask_y[0] = dx_y;
    gx.c_mask_scale = dx_scale;
    RET(run(c, gx));
  }
  return SDUMC_OK;
}

// grouped version: identical shapes, per-group pointers given by base + g*stride (floats)
struct GroupPtrs {
  const float* dz;  int64_t dz_gs;  int lddz;
  const float* x;   int64_t x_gs;   int ldx;
  float* dx;        int64_t dx_gs;  int lddx;
  const float* dx_y = nullptr;   // see lin_bwd: same layout/strides as dx
  float dx_scale = 1.f;
};
int lin_bwd_grouped(const Ctx& c, const Lin* L, int ng, int M, const GroupPtrs& q) {
  sdumc_gemm gw = G_(SDUMC_TN, L[0].out, L[0].in, M, ng);
  sdumc_gemm gx = G_(SDUMC_NN, M, L[0].in, L[0].out, ng);
  for (int g = 0; g < ng; ++g) {
    gw.A[g] = q.dz + g * q.dz_gs;
    gw.B[g] = q.x + g * q.x_gs;
    gw.C[g] = c.G + L[g].w;
    gw.colsum_a[g] = c.G + L[g].b;
    gx.A[g] = q.dz + g * q.dz_gs;
    gx.B[g] = c.P + L[g].w;
    gx.C[g] = q.dx ? q.dx + g * q.dx_gs : nullptr;
    gx.c_mask_y[g] = q.dx_y ? q.dx_y + g * q.dx_gs : nullptr;
  }
  gw.lda = q.lddz;
  gw.ldb = q.ldx;
  gw.ldc = L[0].in;
  c.deferred.push_back(gw);
  if (q.dx) {
    gx.c_mask_scale = q.dx_scale;
    gx.lda = q.lddz;
    gx.ldb = L[0].in;
    gx.ldc = q.lddx;
    RET(run(c, gx));
  }
  return SDUMC_OK;
}

// bytes of ONE keep-bits set of sdumc_net_io.bits_next for dims `d` (= bits_next_off(plan, 2, 0) without building the plan)
size_t bits_set_bytes(const sdumc_net_dims& d) {
  const int64_t V = (int64_t)d.B * d.streams;
  const int64_t rows = V * d.Ta + V * d.Tv + (int64_t)d.B * (d.Tt[0] + (d.streams == 2 ? d.Tt[1] : 0));
  return (size_t)((64 + 2 * rows * (D / 4) + 255) & ~(int64_t)255);
}

int check_io(const sdumc_net_dims* d, const sdumc_net_io* io) {
  if (!d || !io) return SDUMC_EINVAL;
  if (!io->audio || !io->video || !io->text[0] || !io->params || !io->workspace) return SDUMC_EINVAL;
  if (d->streams == 2 && !io->text[1]) return SDUMC_EINVAL;
  if (d->train && !io->rng_state) return SDUMC_EINVAL;
  if (reinterpret_cast<uintptr_t>(io->workspace) & 255) return SDUMC_EINVAL;
  if (reinterpret_cast<uintptr_t>(io->params) & 15) return SDUMC_EINVAL;
  if (reinterpret_cast<uintptr_t>(io->bits_next) & 15) return SDUMC_EINVAL;
  if (io->bits_next && d->train && d->bf16 != 2) {
    const sdumc_net_dims* nd = io->bits_next_dims;
    if (nd && (nd->streams != d->streams || nd->da != d->da || nd->dt != d->dt || nd->dv != d->dv || nd->train != d->train || nd->bf16 != d->bf16 ||
               nd->p_frame != d->p_frame))
      return SDUMC_EINVAL;
    if (nd && (nd->B <= 0 || nd->Ta <= 0 || nd->Tv <= 0 || nd->Tt[0] <= 0 || (nd->streams == 2 && nd->Tt[1] <= 0))) return SDUMC_EINVAL;
    // a set must fit its half of a stated capacity; with none stated the buffer is exactly this call's size
    const size_t half = io->bits_next_bytes ? ((io->bits_next_bytes / 2) & ~(size_t)255) : bits_set_bytes(*d);
    if (bits_set_bytes(*d) > half || (nd && bits_set_bytes(*nd) > half)) return SDUMC_ENOMEM;
  }
  if (io->ctx) {   // a caller-owned context belongs to the device it was created on
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || static_cast<const LaneSet*>(io->ctx)->device != dev) return SDUMC_EINVAL;
  }
  {   // bf16-plane copies of the features: all of them or none; fp32 storage only
    const int need = 2 + d->streams;
    int have = (io->audio_p3 != nullptr) + (io->video_p3 != nullptr);
    for (int s = 0; s < d->streams; ++s) have += io->text_p3[s] != nullptr;
    if (have != 0 && (have != need || d->bf16 == 2)) return SDUMC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(io->audio_p3) | reinterpret_cast<uintptr_t>(io->video_p3) | reinterpret_cast<uintptr_t>(io->text_p3[0]) |
         reinterpret_cast<uintptr_t>(io->text_p3[1])) & 15)
      return SDUMC_EINVAL;
  }
  {   // row maps (the batch read in place from a resident store): all of (audio, text, video[, feat4]) or none; fp32 storage with planes
    const int need = d->streams == 2 ? 4 : 3;
    int have = 0;
    for (int i = 0; i < need; ++i) have += io->row_map[i] != nullptr;
    if (have != 0 && (have != need || d->bf16 == 1 || (d->bf16 == 0 && !io->audio_p3))) return SDUMC_EINVAL;
    for (int i = 0; i < 4; ++i)
      if ((reinterpret_cast<uintptr_t>(io->row_map[i]) & 15) || io->store_rows[i] < 0) return SDUMC_EINVAL;
  }
  {   // key-padding lengths: all of (audio, text, video[, feat4]) or none
    const int need = d->streams == 2 ? 4 : 3;
    int have = 0;
    for (int i = 0; i < need; ++i) have += io->lengths[i] != nullptr;
    if (have != 0 && have != need) return SDUMC_EINVAL;
  }
  return SDUMC_OK;
}

// lens[m][s * B + b] = lengths of modality m for stream s (audio / video: the same for both streams; text: stream 1 reads feat4)
__global__ void expand_lengths_kernel(const int32_t* la, const int32_t* lt, const int32_t* lv, const int32_t* l4, int B, int S,
                                      int32_t* lens) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, V = B * S;
  if (i >= 3 * V) return;
  const int m = i / V, v = i - m * V, s = v / B, b = v - s * B;
  lens[i] = m == 0 ? la[b] : (m == 2 ? lv[b] : (s == 0 ? lt[b] : l4[b]));
}

// the attention-pooling descriptor of (site kind k, modality m, run sg) — shared by forward and backward
sdumc_attnpool attn_desc(const Ctx& c, int k, int m, const Seg& sg) {
  const Plan& pl = c.pl;
  const int nq = k == 0 ? 1 : NQ;
  sdumc_attnpool a;
  memset(&a, 0, sizeof(a));
  a.V = sg.V;
  a.T = sg.T;
  a.nq = nq;
  a.x_samples = sg.x_samples;
  a.x = c.p(sg.x_off);
  a.keys = c.p(pl.keys[k][m]) + sg.row0 * D;
  if (c.h()) {     // bf16 frames; in train mode the masked frames xd of this site stand in for (x, input dropout)
    a.bf16 = 1;
    a.keys = reinterpret_cast<const float*>(c.ph(pl.keys[k][m], sg.row0 * D));
    if (c.d.train) {
      a.x = reinterpret_cast<const float*>(c.ph(pl.xd[k][m], sg.row0 * D));
      a.x_samples = sg.V;
    }
  }
  if (k == 0) {
    a.q = c.P + c.pm.fra_ctx[m];
    a.q_stride = 0;
  } else {
    a.q = c.p(pl.qp) + ((int64_t)m * pl.V + (int64_t)sg.s0 * pl.B) * NQ * D;
    a.q_stride = (int64_t)NQ * D;
  }
  a.scale = 0.3f;
  a.x_drop = in_drop(c, k, m, sg.T, sg.s0, sg.row0);
  if (c.h()) a.x_drop.enabled = 0;
  a.out_drop = mkdrop(c, SITE_OUT[k][m], c.d.p_frame, nq, D, sg.s0);
  a.attn = c.p(pl.attn[k][m]) + sg.row0 * nq;
  a.pooled = c.p(pl.pooled[k][m]) + (int64_t)sg.s0 * pl.B * nq * D;
  float* outbase = k == 0 ? c.p(pl.hpre) + (int64_t)m * pl.V * D : c.p(pl.ca_out) + (int64_t)m * pl.V * NQ * D;
  a.out = outbase + (int64_t)sg.s0 * pl.B * nq * D;
  if (c.io.lengths[0])
    a.lengths = reinterpret_cast<const int32_t*>(c.p(pl.lens)) + (int64_t)m * pl.V + (int64_t)sg.s0 * pl.B;
  return a;
}

// keys = tanh(drop(x) W^T + b) of the attention sites [k0, k1) of modality m (FRA2UTT_new = 0, Cross_Attention = 1:
// they read the same x with different masks and weights).  Both in one grouped launch fill the last dispatch round
// of the 64x64 tiles better; one at a time lets the Cross_Attention half run on the background lane.
int run_h(const Ctx& c, sdumc_gemm_bf16& g) {
  g.workspace = c.scr;
  g.workspace_bytes = (size_t)c.pl.scratch_floats * sizeof(float);
  return sdumc_gemm_bf16_run(&g, c.st);
}
sdumc_gemm_bf16 GH_(int layout, int M, int N, int K, int groups = 1) {
  sdumc_gemm_bf16 g;
  memset(&g, 0, sizeof(g));
  g.layout = layout;
  g.M = M;
  g.N = N;
  g.K = K;
  g.groups = groups;
  return g;
}

// bf16 storage: the NT products of the frame-level forward through gemm_b1.hip (the weight fragment-major, straight into registers).
// SDUMC_P3=0 (the switch of "operands prepared once per tensor", shared with the fp32 planes path): A/B against gemm_bf16.hip's LDS-staged tiles.
bool prepared_operands_on() {
  static const int on = [] { const char* e = getenv("SDUMC_P3"); return e ? atoi(e) : 1; }();
  return on != 0;
}
bool b1_mode(const Ctx& c) { return prepared_operands_on() && c.h() && c.pl.wb1 != 0; }
char* wb1_ptr(const Ctx& c, int64_t byte_off) { return reinterpret_cast<char*>(c.p(c.pl.wb1)) + byte_off; }
// modality m's three frame-level weights (frame_dim_reshape_m, the two input_proj) -> fragment-major bf16: one small launch at the head
// of the modality's lane
int b1_refresh_weights(const Ctx& c, int m) {
  const int din[3] = {c.d.da, c.d.dt, c.d.dv};
  int64_t so[3], dofs[3];
  int32_t rows[3], cols[3];
  so[0] = c.pm.frame[m].w; dofs[0] = c.pl.wp3_frame[m]; rows[0] = D; cols[0] = din[m];
  so[1] = c.pm.fra_proj[m].w; dofs[1] = c.pl.wp3_key[0][m]; rows[1] = D; cols[1] = D;
  so[2] = c.pm.ca_in[m].w; dofs[2] = c.pl.wp3_key[1][m]; rows[2] = D; cols[2] = D;
  return sdumc_b1_frag_multi(c.P, wb1_ptr(c, 0), so, dofs, rows, cols, 3, c.st);
}
// frame_dim_reshape_m on bf16 features: x (bf16) of stream s (and, with feat2, of the next stream in the adjacent rows)
const int32_t* feat_map(const Ctx& c, int m, int s);
int64_t feat_store_rows(const Ctx& c, int m, int s, bool both);
int b1_frame_proj(const Ctx& c, int m, int s, const void* feat, int rows, const void* feat2 = nullptr, int rows2 = 0) {
  const int din[3] = {c.d.da, c.d.dt, c.d.dv};
  sdumc_gemm_b1 g;
  memset(&g, 0, sizeof(g));
  g.M = rows + rows2; g.N = D; g.K = din[m];
  g.A = feat; g.lda = din[m];
  g.a_map = feat_map(c, m, s);      // (a resident store's packed bf16 rows, read in place)
  g.a_map_rows = feat_store_rows(c, m, s, feat2 != nullptr);
  if (feat2) { g.A2 = feat2; g.a2_row0 = rows; g.a2_map = feat_map(c, m, 1); }
  g.B = wb1_ptr(c, c.pl.wp3_frame[m]); g.ldb = (int64_t)(din[m] / 16) * 1024;
  g.bias = c.P + c.pm.frame[m].b;
  g.act = SDUMC_ACT_NONE;
  g.C = c.ph(c.pl.x[m][s]); g.ldc = D; g.c_bf16 = 1;
  // (the text slot, 4096 rows x 4096: K over 4 workgroups -- 0.8748-0.8841 ms against 0.8832-0.8871 with the 8 the plan picks for a launch
  //  alone, 0.8763-0.8837 with 2, 0.8778-0.8917 with none: in the step a half-filled chip is filled by the other lanes)
  if (g.M < 8192 && g.K >= 2048) g.splitk = 4;
  g.workspace = c.scr;
  g.workspace_bytes = (size_t)c.pl.scratch_floats * sizeof(float);
  return sdumc_gemm_b1_nt(&g, c.st);
}

// bf16-storage mode: keys = tanh(xd W^T + b) of the sites [k0, k1), xd = drop(x) materialised by forward() (train) or x (eval)
int keys_gemm_fwd_h(const Ctx& c, int m, int k0, int k1) {
  const Plan& pl = c.pl;
  if (b1_mode(c)) {
    for (int k = k0; k < k1; ++k)
      for (const Seg& sg : pl.segs[m]) {
        const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
        sdumc_gemm_b1 g;
        memset(&g, 0, sizeof(g));
        g.M = sg.V * sg.T; g.N = D; g.K = D;
        g.A = c.d.train ? c.ph(pl.xd[k][m], sg.row0 * D) : c.ph(sg.x_off);
        g.lda = D;
        g.a_row_mod = (!c.d.train && sg.x_samples < sg.V) ? sg.x_samples * sg.T : 0;
        g.B = wb1_ptr(c, pl.wp3_key[k][m]); g.ldb = (int64_t)(D / 16) * 1024;
        g.bias = c.P + L.b;
        g.act = SDUMC_ACT_TANH;
        g.C = c.ph(pl.keys[k][m], sg.row0 * D); g.ldc = D; g.c_bf16 = 1;
        g.splitk = 1;
        RET(sdumc_gemm_b1_nt(&g, c.st));
      }
    return SDUMC_OK;
  }
  for (const Seg& sg : pl.segs[m]) {
    const int64_t rows = (int64_t)sg.V * sg.T;
    sdumc_gemm_bf16 g = GH_(SDUMC_NT, (int)rows, D, D, k1 - k0);
    for (int k = k0; k < k1; ++k) {
      const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
      if (c.d.train) {     // the masked frames xd were written together with the keep-bits (forward())
        g.A[k - k0] = c.ph(pl.xd[k][m], sg.row0 * D);
      } else {
        g.A[k - k0] = c.ph(sg.x_off);
      }
      g.B[k - k0] = c.ph(pl.wh, L.w);
      g.bias[k - k0] = c.P + L.b;
      g.C[k - k0] = c.ph(pl.keys[k][m], sg.row0 * D);
    }
    g.lda = g.ldb = g.ldc = D;
    g.a_row_mod = (!c.d.train && sg.x_samples < sg.V) ? sg.x_samples * sg.T : 0;
    g.act = SDUMC_ACT_TANH;
    g.c_bf16 = 1;
    RET(run_h(c, g));
  }
  return SDUMC_OK;
}

// fp32 storage with the features also given as bf16 planes (sdumc_net_io.*_p3): the frame projections and the key projections that
// go through the wide NT kernel run on operands split once per tensor (gemm_p3.hip).  SDUMC_P3=0: A/B against the in-kernel split.
bool p3_mode(const Ctx& c) {
  return prepared_operands_on() && c.d.bf16 == 0 && c.pl.wp3 != 0 && c.io.audio_p3 != nullptr && sdumc_split_on_(SDUMC_SPLIT_WIDE);
}
char* wp3_ptr(const Ctx& c, int64_t byte_off) { return reinterpret_cast<char*>(c.p(c.pl.wp3)) + byte_off; }
// fragment-major planes of modality m's three frame-level weights (frame_dim_reshape_m, the two input_proj): one small launch at the
// head of the modality's lane (the weights change every step)
int p3_refresh_weights(const Ctx& c, int m) {
  const int din[3] = {c.d.da, c.d.dt, c.d.dv};
  int64_t so[3], dofs[3];
  int32_t rows[3], cols[3];
  so[0] = c.pm.frame[m].w; dofs[0] = c.pl.wp3_frame[m]; rows[0] = D; cols[0] = din[m];
  so[1] = c.pm.fra_proj[m].w; dofs[1] = c.pl.wp3_key[0][m]; rows[1] = D; cols[1] = D;
  so[2] = c.pm.ca_in[m].w; dofs[2] = c.pl.wp3_key[1][m]; rows[2] = D; cols[2] = D;
  return sdumc_p3_split_frag_multi_(c.P, wp3_ptr(c, 0), so, dofs, rows, cols, 3, c.st);
}
// frame_dim_reshape_m on feature planes: x (fp32, for the pooling kernels and K3) and its planes (for the key projections)
// the row map of modality m's features, stream s (sdumc_net_io.row_map is ordered audio, text, video, feat4) or nullptr
int feat_slot(int m, int s) { return m == 0 ? 0 : (m == 2 ? 2 : (s == 0 ? 1 : 3)); }
const int32_t* feat_map(const Ctx& c, int m, int s) { return c.io.row_map[feat_slot(m, s)]; }
// rows of the packed tensor(s) a mapped launch reads (the larger when it reads two; 0 = unknown: 64-bit addressing)
int64_t feat_store_rows(const Ctx& c, int m, int s, bool both) {
  const int64_t a = c.io.store_rows[feat_slot(m, s)], b = both ? c.io.store_rows[feat_slot(m, 1)] : a;
  return (a > 0 && b > 0) ? std::max(a, b) : 0;
}
int p3_frame_proj(const Ctx& c, int m, int s, const void* feat_p3, int rows, int splitk, const void* feat2_p3 = nullptr, int rows2 = 0) {
  const int din[3] = {c.d.da, c.d.dt, c.d.dv};
  sdumc_gemm_p3 g;
  memset(&g, 0, sizeof(g));
  g.M = rows + rows2; g.N = D; g.K = din[m];
  g.A = feat_p3; g.lda = (int64_t)din[m] * 6;
  g.a_map = feat_map(c, m, s);      // (a resident store's packed planes, read in place)
  g.a_map_rows = feat_store_rows(c, m, s, feat2_p3 != nullptr);
  if (feat2_p3) { g.A2 = feat2_p3; g.a2_row0 = rows; g.a2_map = feat_map(c, m, 1); }      // (the second stream's rows follow the first's in x)
  g.B = wp3_ptr(c, c.pl.wp3_frame[m]); g.ldb = (int64_t)(din[m] / 16) * 3072;
  g.bias = c.P + c.pm.frame[m].b;
  g.act = SDUMC_ACT_NONE;
  g.C = c.p(c.pl.x[m][s]); g.ldc = D;
  g.C_p3 = c.p(c.pl.xp3[m][s]); g.ldc_p3 = (int64_t)D * 6;
  g.splitk = splitk;
  g.workspace = c.scr;
  g.workspace_bytes = (size_t)c.pl.scratch_floats * sizeof(float);
  return sdumc_gemm_p3_nt(&g, c.st);
}
// input_proj of site (k, m) on the projected frames' planes: keys = tanh(drop(x) W^T + b)
int p3_keys_fwd(const Ctx& c, int m, int k) {
  for (const Seg& sg : c.pl.segs[m]) {
    const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
    const sdumc_dropout dr = in_drop(c, k, m, sg.T, sg.s0, sg.row0);
    sdumc_gemm_p3 g;
    memset(&g, 0, sizeof(g));
    g.M = sg.V * sg.T; g.N = D; g.K = D;
    g.A = c.p(c.pl.xp3[m][sg.s0]); g.lda = (int64_t)D * 6;
    g.a_row_mod = sg.x_samples < sg.V ? sg.x_samples * sg.T : 0;
    if (dr.enabled) { g.a_bits = dr.bits; g.bits_qw = D / 4; g.a_scale = dr.scale; }
    g.B = wp3_ptr(c, c.pl.wp3_key[k][m]); g.ldb = (int64_t)(D / 16) * 3072;
    g.bias = c.P + L.b;
    g.act = SDUMC_ACT_TANH;
    g.C = c.p(c.pl.keys[k][m]) + sg.row0 * D; g.ldc = D;
    g.splitk = 1;
    RET(sdumc_gemm_p3_nt(&g, c.st));
  }
  return SDUMC_OK;
}

int keys_gemm_fwd(const Ctx& c, int m, int k0, int k1) {
  if (c.h()) return keys_gemm_fwd_h(c, m, k0, k1);
  if (p3_mode(c)) {
    for (int k = k0; k < k1; ++k) RET(p3_keys_fwd(c, m, k));
    return SDUMC_OK;
  }
  for (const Seg& sg : c.pl.segs[m]) {
    sdumc_gemm g = G_(SDUMC_NT, sg.V * sg.T, D, D, k1 - k0);
    for (int k = k0; k < k1; ++k) {
      const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
      g.A[k - k0] = c.p(sg.x_off);
      g.B[k - k0] = c.P + L.w;
      g.bias[k - k0] = c.P + L.b;
      g.C[k - k0] = c.p(c.pl.keys[k][m]) + sg.row0 * D;
      g.ab_drop_bits[k - k0] = in_drop(c, k, m, sg.T, sg.s0, sg.row0).bits;
    }
    g.lda = g.ldb = g.ldc = D;
    g.a_row_mod = sg.x_samples < sg.V ? sg.x_samples * sg.T : 0;
    g.a_drop = in_drop(c, k0, m, sg.T, sg.s0, sg.row0);
    g.ab_drop_group_stride = SITE_IN[1][m] - SITE_IN[0][m];
    g.act = SDUMC_ACT_TANH;
    g.bf16 = c.d.bf16 ? 1 : 0;
    RET(run(c, g));
  }
  return SDUMC_OK;
}

// softmax-over-time pooling of site (k, m) given its keys (and, for k = 1, the projected queries)
// Sites (k, m = 0..2) in one launch on the current lane (sdumc_attnpool_fwd_multi / _bwd_multi); false = take the per-lane path
// (more than four runs).
bool attn_multi_ok(const Ctx& c) {
  size_t n = 0;
  for (int m = 0; m < 3; ++m) n += c.pl.segs[m].size();
  return n <= 4;
}

bool fra_fold(const Ctx& c);                                        // (defined with the chain launch helpers below)
void fold_partial_only(const Ctx& c, int k, int m, sdumc_attnpool& a);
int pool_fwd_multi(const Ctx& c, int k) {
  sdumc_attnpool a[4];
  int n = 0;
  float* ws = c.scr;
  const int order[3] = {0, 2, 1};            // heaviest first
  for (int oi = 0; oi < 3; ++oi)
    for (const Seg& sg : c.pl.segs[order[oi]]) {
      a[n] = attn_desc(c, k, order[oi], sg);
      const size_t bytes = sdumc_attnpool_fwd_workspace_bytes(sg.V, sg.T, a[n].nq);
      a[n].workspace = ws;
      a[n].workspace_bytes = bytes;
      ws += (bytes / sizeof(float) + 63) / 64 * 64;
      if (fra_fold(c)) fold_partial_only(c, k, order[oi], a[n]);      // (the clustered stage behind this launch combines)
      ++n;
    }
  return sdumc_attnpool_fwd_multi(a, n, c.st);
}

// K3 (sdumc_umca_fwd): key projection + pooling of site (k, m) in one kernel per run; the keys also go to HBM (`keep`: the
// backward reads them, and a backward may follow an eval-mode forward too).  fp32 storage only.
// Used for the FRA2UTT site (k = 0) of the modalities with more than one 64-frame chunk per sample.  Measured at C2
// (tools/umca_bench.py, tools/infer_bench.py, bench.py; SDUMC_K3=0 restores the two-kernel path):
//   * one site alone: audio 95 vs 97 us (GEMM + pooling), video 59 vs 69 us, C5's T = 512 68 vs 84 us; text's single 50-frame
//     chunk 39 vs 33 us (22 % of its 64-row tile is padding and 128 workgroups do not fill the chip) -> text keeps the pair;
//   * training step: 1.903 vs 1.914 ms (two alternations);
//   * the Cross_Attention site (k = 1) does NOT use it: its projected queries exist only after the first utterance-level
//     stage, so fusing would pull the key projection out of that stage's shadow -- an inference forward with all six sites
//     fused measured 570 vs 544 us (one stream), 716 vs 706 us (two).
bool k3_ok(const Ctx& c, int m) {
  static const int on = [] { const char* e = getenv("SDUMC_K3"); return e ? atoi(e) : 1; }();
  if (!on || c.h() || c.d.bf16) return false;
  for (const Seg& sg : c.pl.segs[m])
    if (sg.T < 96) return false;
  return true;
}
int umca_site(const Ctx& c, int k, int m, bool keep) {
  for (const Seg& sg : c.pl.segs[m]) {
    sdumc_umca u;
    memset(&u, 0, sizeof(u));
    u.a = attn_desc(c, k, m, sg);
    u.a.tickets = nullptr;
    if (!keep) u.a.keys = nullptr;
    u.a.workspace = c.scr;
    u.a.workspace_bytes = (size_t)c.pl.scratch_floats * sizeof(float);
    if (fra_fold(c)) fold_partial_only(c, k, m, u.a);
    const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
    u.w_in = c.P + L.w;
    u.b_in = c.P + L.b;
    if (p3_mode(c) && sdumc_split_on_(SDUMC_SPLIT_UMCA)) {      // the fused kernel's projection on planes (1.361-1.364 against 1.374-1.376 ms)
      u.x_p3 = c.p(c.pl.xp3[m][sg.s0]);
      u.w_in_p3f = wp3_ptr(c, c.pl.wp3_key[k][m]);
    }
    RET(sdumc_umca_fwd(&u, c.st));
  }
  return SDUMC_OK;
}

int pool_fwd(const Ctx& c, int k, int m) {
  for (const Seg& sg : c.pl.segs[m]) {
    sdumc_attnpool a = attn_desc(c, k, m, sg);
    a.workspace = c.scr;
    a.workspace_bytes = (size_t)c.pl.scratch_floats * sizeof(float);
    if (fra_fold(c)) fold_partial_only(c, k, m, a);
    RET(sdumc_attnpool_fwd(&a, c.st));
  }
  return SDUMC_OK;
}


// ------------------------------------------------------------------------------------------
// chain.hip: the utterance-level network (model :293-332, :338-368) in one launch per stage and direction instead of one
// launch per layer.  Taken when the virtual batch is small enough that those layers are launch-bound (V <= 512: up to
// B = 256 per GPU); larger batches keep the per-layer MFMA GEMMs, which are then compute-bound.  SDUMC_CHAIN=0 / 1 forces
// either path (A/B measurements, parity tests of both).
// ------------------------------------------------------------------------------------------
bool use_chain(const Ctx& c) {
  static const int forced = [] { const char* e = getenv("SDUMC_CHAIN"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced != 0;
  return c.pl.V <= 512;
}

// chain_cluster.hip instead of chain.hip: every workgroup of the clustered kernels must be resident at once (V <= 128 on
// MI355X) and their launches wait on a per-device event, which a stream capture cannot contain
bool use_cluster(const Ctx& c) {
  if (!use_chain(c) || c.capturing || c.chain_cluster_opt == 0) return false;
  return c.chain_cluster_opt == 1 ? sdumc_chain_cluster_fits_(c.pl.V) == 1 : sdumc_chain_cluster_ok_(c.pl.V) == 1;
}

// The clustered stage A combines the FRA2UTT sites' softmax partials in its own prologue (chain_cluster.hip::fra_combine): the
// pooling launches of the three modality lanes then stop after their per-chunk pass (sdumc_attnpool.partial_only) and the three
// combine launches -- 31-39 us each inside the step, on every lane's way to this stage -- are gone.  One run per modality (equal
// text / feat4 lengths), at most 32 chunks per sample (the separate combine launches remain the path of the plain kernels).
bool fra_fold(const Ctx& c) {
  if (!use_cluster(c)) return false;
  for (int m = 0; m < 3; ++m)
    if (c.pl.segs[m].size() != 1 || (c.pl.segs[m][0].T + 63) / 64 > 32) return false;
  return true;
}
// site (k, m)'s descriptor for the partial-only pass
void fold_partial_only(const Ctx& c, int k, int m, sdumc_attnpool& a) {
  a.partial_only = 1;
  a.tickets = nullptr;
  a.workspace = c.p(c.pl.fold_ws[k][m]);
  a.workspace_bytes = sdumc_attnpool_fwd_workspace_bytes(a.V, a.T, a.nq);
}

int chain_launch(const Ctx& c, const sdumc_chain_args& ca, int which) {
  if (use_cluster(c)) {
    const int rc = sdumc_chain_cluster_launch_(&ca, which, c.st);
    if (rc != 1) return rc;
  }
  if (ca.fra.part[0] || ca.ca.part[0] || ca.dq_part[0]) return SDUMC_ELAUNCH;      // (the partials are only combined by the clustered stages)
  return sdumc_chain_launch_(&ca, which, c.st);
}

// the Linear layers of both stages, in one list (transposed once per forward call)
std::vector<const Lin*> chain_lins(const ParamMap& pm) {
  std::vector<const Lin*> v;
  for (int m = 0; m < 3; ++m) { v.push_back(&pm.umlp0[m]); v.push_back(&pm.umlp3[m]); }
  v.push_back(&pm.att0); v.push_back(&pm.att3);
  for (int i = 0; i < 7; ++i) v.push_back(&pm.query[i]);
  for (int m = 0; m < 3; ++m) v.push_back(&pm.ca_q[m]);
  for (int m = 0; m < 3; ++m) { v.push_back(&pm.cmlp0[m]); v.push_back(&pm.cmlp3[m]); }
  v.push_back(&pm.catt0); v.push_back(&pm.catt3);
  v.push_back(&pm.rnc0); v.push_back(&pm.rnc2);
  return v;
}

// transposed fp32 mirror of the utterance-level weights, refreshed once per forward (the chain kernels stream W^T forward).
// (The key-projection dX as NT on such a mirror of the input_proj weights through the wide LDS-DMA kernel: faster alone, 96-98 vs
//  70 TF, slower inside the step in rounds 2 and 3 -- 1.884 vs 1.784 ms -- its 60 KB LDS rings co-reside badly with the other
//  lanes' kernels; the small-footprint NN kernel stays.)
int chain_transpose(const Ctx& c) {
  int64_t offs[40];
  int32_t outs[40], ins[40];
  int n = 0;
  if (use_chain(c)) {
    const std::vector<const Lin*> ls = chain_lins(c.pm);
    for (const Lin* L : ls) { offs[n] = L->w; outs[n] = L->out; ins[n] = L->in; ++n; }
  }
  if (n == 0) return SDUMC_OK;
  return sdumc_chain_transpose_(c.P, c.p(c.pl.wt), offs, outs, ins, n, c.st);
}

// fwd: weights from the transposed mirror; bwd: as stored
// stage_a: the launch is one of the stage-A kernels.  In bf16-storage mode only those stream bf16 weights: their layers are
// 2-row products bound by the weight stream (fwd 104 -> 82 us, bwd 100 -> 86 us at C2), while stage B's 14-row layers are bound
// by their FMAs and the bf16 unpacking made them slower (69 -> 85 us), so stage B keeps reading the fp32 weights.
sdumc_chain_args chain_args(const Ctx& c, bool fwd, const sdumc_net_grads* og, bool stage_a) {
  const Plan& pl = c.pl;
  const ParamMap& pm = c.pm;
  sdumc_chain_args a;
  memset(&a, 0, sizeof(a));
  a.V = pl.V;
  a.B = pl.B;
  a.drop = mkdrop(c, 0, c.d.p_mlp, 1, D);
  a.relu_scale = c.d.train ? 1.0f / (1.0f - (float)c.d.p_mlp) : 1.0f;
  const float* WB = fwd ? c.p(pl.wt) : c.P;
  // bf16-storage mode: the streamed matrices are read from the bf16 copies (transposed for the forward, as stored for the
  // backward: sdumc_weights_to_bf16_ in forward()); orgin_linear_change stays fp32 (64 columns: too narrow for 8-column lanes)
  const bool wb = c.h() && stage_a && !use_cluster(c);
  a.w_bf16 = wb ? 1 : 0;
  // bf16 MFMA kernels run beside the utterance-level stages (chain_common.h): in the bf16 modes always, in fp32 when the GEMM
  // kernels compute their products on the bf16 matrix pipe (sdumc_set_split_).
  a.no_packed_fp32 = 1;      // (the whole device build carries no packed fp32 operations since round 4: one entry point per stage)
  auto W = [&](const Lin& L) -> const float* {
    if (wb) return reinterpret_cast<const float*>(c.ph(fwd ? pl.wht : pl.wh, L.w));
    return WB + L.w;
  };
  auto Bs = [&](const Lin& L) { return c.P + L.b; };
  for (int m = 0; m < 3; ++m) {
    a.umlp0_w[m] = W(pm.umlp0[m]); a.umlp0_b[m] = Bs(pm.umlp0[m]);
    a.umlp3_w[m] = W(pm.umlp3[m]); a.umlp3_b[m] = Bs(pm.umlp3[m]);
    a.caq_w[m] = W(pm.ca_q[m]); a.caq_b[m] = Bs(pm.ca_q[m]);
    a.cmlp0_w[m] = W(pm.cmlp0[m]); a.cmlp0_b[m] = Bs(pm.cmlp0[m]);
    a.cmlp3_w[m] = W(pm.cmlp3[m]); a.cmlp3_b[m] = Bs(pm.cmlp3[m]);
  }
  for (int i = 0; i < 7; ++i) { a.query_w[i] = W(pm.query[i]); a.query_b[i] = Bs(pm.query[i]); }
  a.att0_w = W(pm.att0); a.att0_b = Bs(pm.att0);
  a.att3_w = W(pm.att3); a.att3_b = Bs(pm.att3);
  a.fc_att_w = c.P + pm.fc_att.w; a.fc_att_b = Bs(pm.fc_att);
  a.catt0_w = W(pm.catt0); a.catt0_b = Bs(pm.catt0);
  a.catt3_w = W(pm.catt3); a.catt3_b = Bs(pm.catt3);
  a.cfa_w = c.P + pm.cross_fc_att.w; a.cfa_b = Bs(pm.cross_fc_att);
  a.fcv_w = c.P + pm.fc_out_v.w; a.fcv_b = Bs(pm.fc_out_v);
  a.rnc0_w = WB + pm.rnc0.w; a.rnc0_b = Bs(pm.rnc0);
  a.rnc2_w = WB + pm.rnc2.w; a.rnc2_b = Bs(pm.rnc2);
  a.hpre = c.p(pl.hpre); a.u1 = c.p(pl.u1); a.u = c.p(pl.u); a.att1 = c.p(pl.att1); a.att2 = c.p(pl.att2);
  a.alpha = c.p(pl.alpha); a.qin = c.p(pl.qin); a.q = c.p(pl.q); a.qp = c.p(pl.qp); a.ca_out = c.p(pl.ca_out);
  a.c1 = c.p(pl.c1); a.c = c.p(pl.c); a.h = c.p(pl.h); a.e1 = c.p(pl.e1); a.e2 = c.p(pl.e2); a.beta = c.p(pl.beta);
  a.z = c.p(pl.z); a.vals = c.p(pl.vals); a.r1 = c.p(pl.r1); a.r = c.p(pl.r);
  if (fwd) {
    a.o_vals = c.io.vals; a.o_fused = c.io.fused; a.o_rnc = c.io.rnc; a.o_text_hidden = c.io.text_hidden;
    a.o_cross_text = c.io.cross_text;
  }
  if (!fwd && stage_a && c.dq_part[0]) {
    for (int m = 0; m < 3; ++m) { a.dq_part[m] = c.dq_part[m]; a.dq_nchunk[m] = c.dq_nchunk[m]; }
  }
  if (fwd && fra_fold(c)) {      // stage A combines the FRA2UTT sites' partials, stage B the Cross_Attention sites'
    const int k = stage_a ? 0 : 1, nq = stage_a ? 1 : NQ;
    sdumc_chain_fold& f = stage_a ? a.fra : a.ca;
    for (int m = 0; m < 3; ++m) {
      const Seg& sg = pl.segs[m][0];
      const int nchunk = (sg.T + 63) / 64;
      f.part[m] = c.p(pl.fold_ws[k][m]);
      f.stats[m] = c.p(pl.fold_ws[k][m]) + (int64_t)pl.V * nchunk * nq * D;
      f.attn[m] = c.p(pl.attn[k][m]);
      f.pooled[m] = c.p(pl.pooled[k][m]);
      f.nchunk[m] = nchunk;
      f.T[m] = sg.T;
      f.site[m] = SITE_OUT[k][m];
    }
    f.threshold = (uint32_t)(uint64_t)(c.d.p_frame * 4294967296.0);
    f.scale = 1.0f / (1.0f - (float)c.d.p_frame);
  }
  if (og) {
    a.g_vals = og->d_vals; a.g_fused = og->d_fused; a.g_rnc = og->d_rnc; a.g_text_hidden = og->d_text_hidden;
    a.g_cross_text = og->d_cross_text;
  }
  a.d_r1 = c.p(pl.d_r1); a.d_z = c.p(pl.d_z); a.d_beta = c.p(pl.d_beta); a.d_e2 = c.p(pl.d_e2); a.d_e1 = c.p(pl.d_e1);
  a.d_h = c.p(pl.d_h); a.d_c = c.p(pl.d_c); a.d_c1 = c.p(pl.d_c1); a.d_ca_out = c.p(pl.d_ca_out); a.d_alpha = c.p(pl.d_alpha);
  a.d_qp = c.p(pl.d_qp); a.d_q = c.p(pl.d_q); a.d_qin = c.p(pl.d_qin); a.d_u = c.p(pl.d_u); a.d_att2 = c.p(pl.d_att2);
  a.d_att1 = c.p(pl.d_att1); a.d_u1 = c.p(pl.d_u1); a.d_hpre = c.p(pl.d_hpre);
  return a;
}

// sdumc_net_io.prefetch: the next batch's assembly on lane 3, behind what the caller's stream has issued so far (the frame-level head
// of this call: its projections are the step's heaviest HBM readers) -- lane 3 idles from here to the backward's early launch, and the
// step's final join of that lane orders the gather before whatever follows the step.  (Measured, round 6: a stream of its own at
// the LOWEST priority starves -- the gather then ends behind the step, 2.14 against 1.55 ms per step with padded copies and 1.68
// against 1.33 with row maps; in place on the caller's stream 1.70 / 1.33; profiles/README.md.)
int issue_prefetch(const Ctx& c) {
  if (!c.io.prefetch) return SDUMC_OK;
  const int wgs = c.io.prefetch_workgroups > 0 ? c.io.prefetch_workgroups : 512;
  if (!c.multi || c.capturing) return sdumc_gather_batch(c.io.prefetch, wgs, c.sts[0]);
  RET(link(c, 0, 3));
  return sdumc_gather_batch(c.io.prefetch, wgs, c.sts[3]);
}

int forward(const Ctx& c) {
  const Plan& pl = c.pl;
  const ParamMap& pm = c.pm;
  const int B = pl.B, S = pl.S, V = pl.V;
  const int din[3] = {c.d.da, c.d.dt, c.d.dv};
  if (c.io.lengths[0]) {
    hipLaunchKernelGGL(expand_lengths_kernel, dim3((3 * V + 255) / 256), dim3(256), 0, c.st, c.io.lengths[0], c.io.lengths[1],
                       c.io.lengths[2], S == 2 ? c.io.lengths[3] : c.io.lengths[1], B, S,
                       reinterpret_cast<int32_t*>(c.p(pl.lens)));
    SDUMC_CHECK_LAUNCH();
  }
  // 1+2. three independent per-modality chains, one per lane:
  //      keep-bits of the two frame-level input dropouts -> frame_dim_reshape_m (model :282-284; audio/video once
  //      for both streams) -> keys of fra2utt_m AND cross_att_fra2utt_m -> FRA2UTT pooling (model :288-290)
  if (c.io.row_map[0] && !p3_mode(c) && !b1_mode(c)) return SDUMC_EINVAL;      // (only gemm_p3 / gemm_b1 fetch their rows through a map)
  RET(fork_all(c));
  const bool chain = use_chain(c);
  hipEvent_t wt_done = nullptr;
  if (c.h()) {   // bf16 copies of the frame-level weights (+ the transposed input_proj copies the dX products read)
    int64_t offs[16];
    int32_t outs[16], ins[16], wantt[16];
    int n = 0;
    for (int m = 0; m < 3; ++m) { offs[n] = pm.frame[m].w; outs[n] = D; ins[n] = din[m]; wantt[n] = 0; ++n; }
    for (int m = 0; m < 3; ++m) {
      offs[n] = pm.fra_proj[m].w; outs[n] = D; ins[n] = D; wantt[n] = 1; ++n;
      offs[n] = pm.ca_in[m].w; outs[n] = D; ins[n] = D; wantt[n] = 1; ++n;
    }
    // (with gemm_b1.hip the forward reads fragment-major copies made on each modality's own lane; these row-major / transposed
    //  copies are first read by the backward dX products: lane 3, off the head of the step)
    if (b1_mode(c)) { RET(link(c, 0, 3)); c.use(3); }
    RET(sdumc_weights_to_bf16_(c.P, c.ph(pl.wh), c.ph(pl.wht), offs, outs, ins, wantt, n, c.st));
    if (chain && !use_cluster(c)) {   // + the utterance-level matrices chain.hip streams (both layouts: forward and backward)
      const std::vector<const Lin*> ls = chain_lins(pm);
      n = 0;
      for (size_t i = 0; i <= ls.size(); ++i) {
        if (n == 16 || (i == ls.size() && n > 0)) {
          RET(sdumc_weights_to_bf16_(c.P, c.ph(pl.wh), c.ph(pl.wht), offs, outs, ins, wantt, n, c.st));
          n = 0;
        }
        if (i < ls.size()) { offs[n] = ls[i]->w; outs[n] = ls[i]->out; ins[n] = ls[i]->in; wantt[n] = 1; ++n; }
      }
    }
    if (b1_mode(c)) c.use(0);
    else RET(fork_all(c));      // (the lanes forked above did not see these launches)
  }
  if (chain) {   // transposed mirror (first needed after the frame-level part): lane 3
    RET(link(c, 0, 3));
    c.use(3);
    RET(chain_transpose(c));
    if (c.sts[3] != c.sts[0]) {
      wt_done = next_event(c);
      if (hipEventRecord(wt_done, c.st) != hipSuccess) return SDUMC_ELAUNCH;
    }
    c.use(0);
  }
  // The keep-bits (Philox once per element instead of ~10x in the kernels that stage these tiles; VALU-bound) are generated
  // on lane 3 in the shadow of the MFMA-bound frame projections, which read no mask; each modality's lane waits for its
  // own bits before the key projections.  Heaviest modality first.
  hipEvent_t bits_done[3] = {nullptr, nullptr, nullptr};
  // bf16 storage: the keep-bits are produced together with the masked frames xd of both sites, behind the modality's frame
  // projection on the modality's own lane (one pass over x instead of a bits launch on lane 3 plus one mask_apply per site)
  const bool bits_with_xd = c.h() && c.d.train;
  const bool pregen = bits_pregen(c);
  const sdumc_bits_shape shape_now = bits_shape(c.d);
  if (c.d.train && !bits_with_xd) {
    RET(link(c, 0, 3));
    c.use(3);
    const int order[3] = {0, 2, 1};
    for (int oi = 0; oi < 3; ++oi) {
      const int m = order[oi];
      for (const Seg& sg : pl.segs[m]) {   // both sites (fra2utt_m, cross_att_fra2utt_m) in one launch
        sdumc_dropout d = mkdrop(c, SITE_IN[0][m], c.d.p_frame, sg.T, D, sg.s0);
        uint8_t* outs[2] = {bits_ptr(c, 0, m) + sg.row0 * (D / 4), bits_ptr(c, 1, m) + sg.row0 * (D / 4)};
        // (bits_next: nothing to do when the previous call filled this set for this call and shape -- its tag says so)
        RET(sdumc_dropout_bits_multi_ex_(&d, sg.V / B, 2, SITE_IN[1][m] - SITE_IN[0][m], outs,
                                         pregen ? reinterpret_cast<const uint32_t*>(bits_set(c, 0)) : nullptr, 0, &shape_now, c.st));
      }
      if (c.sts[3] != c.sts[LANE_OF[m]]) {
        bits_done[m] = next_event(c);
        if (hipEventRecord(bits_done[m], c.st) != hipSuccess) return SDUMC_ELAUNCH;
      }
    }
    if (pregen) {
      // Whatever the set's tag said, the set now holds THIS call's masks: say so (behind the three launches that read the tag, on their
      // lane).  A set regenerated under a foreign tag would otherwise keep naming the call it was once filled for, and a counter that
      // comes back to that value (a rewind, a phase that was not flipped) would find stale masks under a matching tag.
      const sdumc_dropout d0 = mkdrop(c, SITE_IN[0][0], c.d.p_frame, pl.segs[0][0].T, D, 0);
      RET(sdumc_bits_tag_(&d0, reinterpret_cast<uint32_t*>(bits_set(c, 0)), 0, &shape_now, c.st));
    }
  }
  hipEvent_t fra_done[3] = {nullptr, nullptr, nullptr}, ca_done[3] = {nullptr, nullptr, nullptr};
  // frame_dim_reshape_m on feature planes (gemm_p3.hip), all streams of the modality, on the current lane
  auto p3_frames = [&](int m) -> int {
    // (the text slot: 2048 x 256 x 4096 per stream -- K split over workgroups; both streams in one launch, their x rows are adjacent)
    constexpr int p3_text_split = 4;      // (alone: split 8 34.6 us, 4 37.9, 2 63.8 per stream; both streams in one launch: 4)
    RET(p3_refresh_weights(c, m));
    for (int s = 0; s < (m == 1 ? S : 1); ++s) {
      const void* fp = m == 0 ? c.io.audio_p3 : (m == 2 ? c.io.video_p3 : c.io.text_p3[s]);
      const int rows_p = B * pl.T[m][s];
      const bool few = rows_p < 8192 && din[m] >= 2048;
      if (m == 1 && S == 2 && few && rows_p % 64 == 0) {
        RET(p3_frame_proj(c, m, 0, fp, rows_p, p3_text_split, c.io.text_p3[1], B * pl.T[1][1]));
        break;
      }
      RET(p3_frame_proj(c, m, s, fp, rows_p, few ? 2 * p3_text_split : 1));
    }
    return SDUMC_OK;
  };
  // (measured and dropped: the three modalities' projections one after the other on one lane, every modality's lane waiting for
  //  its own -- 1.50-1.53 against 1.37-1.39 ms: side by side the lanes' kernels do fill each other's bandwidth bursts)
  for (int m = 0; m < 3; ++m) {
    c.use(LANE_OF[m]);
    if (p3_mode(c)) RET(p3_frames(m));
    if (b1_mode(c)) RET(b1_refresh_weights(c, m));
    for (int s = 0; s < (m == 1 ? S : 1) && !p3_mode(c); ++s) {
      const float* in = m == 0 ? c.io.audio : (m == 2 ? c.io.video : c.io.text[s]);
      if (b1_mode(c)) {    // (the text slot's two streams in one launch: their x rows are adjacent)
        const int rows_p = B * pl.T[m][s];
        if (m == 1 && S == 2 && rows_p % 64 == 0) {
          RET(b1_frame_proj(c, m, 0, in, rows_p, c.io.text[1], B * pl.T[1][1]));
          break;
        }
        RET(b1_frame_proj(c, m, s, in, rows_p));
        continue;
      }
      if (c.h()) {    // features and projected frames in bf16
        sdumc_gemm_bf16 g = GH_(SDUMC_NT, B * pl.T[m][s], D, din[m]);
        g.A[0] = in;
        g.lda = din[m];
        g.B[0] = c.ph(pl.wh, pm.frame[m].w);
        g.ldb = din[m];
        g.bias[0] = c.P + pm.frame[m].b;
        g.C[0] = c.ph(pl.x[m][s]);
        g.ldc = D;
        g.c_bf16 = 1;
        RET(run_h(c, g));
        continue;
      }
      // few-row / long-K projections (the text slot: 2048 x 256 x 4096 per stream at C2) through the wide LDS-DMA kernel with K split
      // 8 ways over workgroups (512 of them, 32 k-tiles each) instead of the 64x64 register-staged kernel's automatic split:
      // fp32 C2 step 1.665-1.669 vs 1.671-1.676 ms (split 4: 1.674-1.678, split 2: 1.705-1.710).  With the products on the bf16
      // matrix pipe the k-loop is shorter and half as many slabs win: split 4 1.424-1.438 against split 8 1.444-1.450 and split
      // 2 1.440-1.450 (three alternations)
      constexpr int text_wide = 4;
      const int rows_ms = B * pl.T[m][s];
      const bool wide_split = text_wide > 0 && !c.d.bf16 && rows_ms < 8192 && din[m] >= 2048 && (din[m] % (16 * text_wide)) == 0 && (rows_ms % 64) == 0;
      RET(lin_fwd(c, pm.frame[m], in, din[m], rows_ms, c.p(pl.x[m][s]), D, SDUMC_ACT_NONE, nullptr, c.d.bf16 != 0,
                  wide_split ? 14 : 0, wide_split ? text_wide : 0));
    }
    mark(c.st, 28 + 4 * m);      // (debug marks 28..39: this modality's lane, frame-level forward: projection done)
    if (bits_done[m] && hipStreamWaitEvent(c.st, bits_done[m], 0) != hipSuccess) return SDUMC_ELAUNCH;
    if (bits_with_xd) {
      for (const Seg& sg : pl.segs[m]) {
        const sdumc_dropout d = mkdrop(c, SITE_IN[0][m], c.d.p_frame, sg.T, D, sg.s0);
        uint8_t* bo[2] = {reinterpret_cast<uint8_t*>(c.p(pl.bits[0][m])) + sg.row0 * (D / 4),
                          reinterpret_cast<uint8_t*>(c.p(pl.bits[1][m])) + sg.row0 * (D / 4)};
        void* xo[2] = {c.ph(pl.xd[0][m], sg.row0 * D), c.ph(pl.xd[1][m], sg.row0 * D)};
        RET(sdumc_dropout_bits_apply_bf16(&d, sg.V / B, SITE_IN[1][m] - SITE_IN[0][m], bo, c.ph(sg.x_off), (int64_t)sg.x_samples * sg.T,
                                          xo, c.st));
      }
    }
    mark(c.st, 29 + 4 * m);      // keep-bits awaited
    const bool k3_site0 = k3_ok(c, m);
    // Audio's Cross_Attention keys (the largest of the three key projections) stay on audio's own lane, BEHIND a partial
    // join: the caller's stream (stage A) waits only for the FRA2UTT site, the keys are awaited before step 8.  Lane 3 then
    // carries two key projections instead of three and the pooling of step 8 waits ~10 us less for it (fp32 C2 1.726-1.731
    // against 1.733-1.738 ms, bf16 1.046-1.047 against 1.053-1.057; video's instead: 1.756-1.772, both: 1.732-1.738).
    constexpr int ca_own = 1;
    if (c.bg && (ca_own & (1 << m)) && LANE_OF[m] != 0 && c.multi && !c.capturing) {
      if (k3_site0) RET(umca_site(c, 0, m, true));
      else { RET(keys_gemm_fwd(c, m, 0, 1)); RET(pool_fwd(c, 0, m)); }
      fra_done[m] = next_event(c);
      if (hipEventRecord(fra_done[m], c.st) != hipSuccess) return SDUMC_ELAUNCH;
      RET(keys_gemm_fwd(c, m, 1, 2));
      ca_done[m] = next_event(c);
      if (hipEventRecord(ca_done[m], c.st) != hipSuccess) return SDUMC_ELAUNCH;
    } else if (c.bg) {   // the Cross_Attention keys are not needed before step 8: background lane, beside steps 2-7
      if (k3_site0) RET(umca_site(c, 0, m, true));
      else RET(keys_gemm_fwd(c, m, 0, 1));
      RET(link(c, LANE_OF[m], 3));
      c.use(3);
      RET(keys_gemm_fwd(c, m, 1, 2));
      c.use(LANE_OF[m]);
      if (!k3_site0) RET(pool_fwd(c, 0, m));
    } else if (k3_site0) {
      RET(umca_site(c, 0, m, true));
      RET(keys_gemm_fwd(c, m, 1, 2));
    } else {
      RET(keys_gemm_fwd(c, m, 0, 2));
      RET(pool_fwd(c, 0, m));
    }
    mark(c.st, 30 + 4 * m);      // FRA2UTT site done
  }
  mark(c.sts[3], 40);            // lane 3: keep-bits + Cross_Attention key projections done
  c.use(0);
  for (int lane = 1; lane <= 2; ++lane) {
    int m_of = -1;
    for (int m = 0; m < 3; ++m) if (LANE_OF[m] == lane && fra_done[m]) m_of = m;
    if (m_of >= 0) {
      if (hipStreamWaitEvent(c.sts[0], fra_done[m_of], 0) != hipSuccess) return SDUMC_ELAUNCH;
    } else {
      RET(link(c, lane, 0));
    }
  }
  RET(issue_prefetch(c));      // the NEXT batch's assembly, beside everything from here on (sdumc_net_io.prefetch)
  if (chain) {   // steps 3-7 in one launch
    if (wt_done && hipStreamWaitEvent(c.st, wt_done, 0) != hipSuccess) return SDUMC_ELAUNCH;
    const sdumc_chain_args ca = chain_args(c, true, nullptr, true);
    mark(c.st, 1);
    RET(chain_launch(c, ca, 0));
    mark(c.st, 2);
  } else {
  // 3. audio/text/video_mlp (model :293-295), grouped over the modality
  {
    sdumc_gemm g = G_(SDUMC_NT, V, D, D, 3);
    for (int m = 0; m < 3; ++m) {
      g.A[m] = c.p(pl.hpre) + (int64_t)m * V * D;
      g.B[m] = c.P + pm.umlp0[m].w;
      g.bias[m] = c.P + pm.umlp0[m].b;
      g.C[m] = c.p(pl.u1) + (int64_t)m * V * D;
    }
    g.lda = g.ldb = g.ldc = D;
    g.act = SDUMC_ACT_RELU;
    g.c_drop = mkdrop(c, SITE_UMLP0, c.d.p_mlp, 1, D);
    g.c_drop_group_stride = 2;
    RET(run(c, g));
    for (int m = 0; m < 3; ++m) {
      g.A[m] = c.p(pl.u1) + (int64_t)m * V * D;
      g.B[m] = c.P + pm.umlp3[m].w;
      g.bias[m] = c.P + pm.umlp3[m].b;
      g.C[m] = c.p(pl.u) + m * D;  // u is [V, 3, D]: the torch.cat of model :301 for free
    }
    g.ldc = 3 * D;
    g.c_drop = mkdrop(c, SITE_UMLP1, c.d.p_mlp, 1, D);
    RET(run(c, g));
  }
  // 4. attention_mlp + fc_att (model :302-303)
  {
    sdumc_dropout d0 = mkdrop(c, SITE_ATT0, c.d.p_mlp, 1, D), d1 = mkdrop(c, SITE_ATT1, c.d.p_mlp, 1, D);
    RET(lin_fwd(c, pm.att0, c.p(pl.u), 3 * D, V, c.p(pl.att1), D, SDUMC_ACT_RELU, &d0));
    RET(lin_fwd(c, pm.att3, c.p(pl.att1), D, V, c.p(pl.att2), D, SDUMC_ACT_RELU, &d1));
    RET(lin_fwd(c, pm.fc_att, c.p(pl.att2), D, V, c.p(pl.alpha), 3, SDUMC_ACT_NONE, nullptr));
  }
  // 5. fused / at / tv / av features (model :305-320)
  RET(sdumc_fusion_fwd(c.p(pl.u), c.p(pl.alpha), c.p(pl.qin), V, c.st));
  // 6. the 7 query MLPs, written straight into multi_query [V,7,D] (model :324-332)
  {
    sdumc_gemm g = G_(SDUMC_NT, V, D, D, 7);
    for (int i = 0; i < 7; ++i) {
      g.A[i] = c.p(pl.qin) + (int64_t)i * V * D;
      g.B[i] = c.P + pm.query[i].w;
      g.bias[i] = c.P + pm.query[i].b;
      g.C[i] = c.p(pl.q) + i * D;
    }
    g.lda = g.ldb = D;
    g.ldc = NQ * D;
    g.act = SDUMC_ACT_RELU;
    g.c_drop = mkdrop(c, SITE_QUERY, c.d.p_mlp, 1, D);
    g.c_drop_group_stride = 1;
    RET(run(c, g));
  }
  // 7. query_proj of the three Cross_Attention blocks (model :85)
  {
    sdumc_gemm g = G_(SDUMC_NT, V * NQ, D, D, 3);
    for (int m = 0; m < 3; ++m) {
      g.A[m] = c.p(pl.q);
      g.B[m] = c.P + pm.ca_q[m].w;
      g.bias[m] = c.P + pm.ca_q[m].b;
      g.C[m] = c.p(pl.qp) + (int64_t)m * V * NQ * D;
    }
    g.lda = g.ldb = g.ldc = D;
    RET(run(c, g));
  }
  }   // !chain
  // 8. cross_att_fra2utt_{0,1,2} (model :334-336): one grouped launch on the caller's stream (a fork/join around three 17-30 us
  //    kernels cost 86-91 us between the two utterance-level launches, of which ~35 us were cross-queue event latency)
  // their keys: lane 3 collects the own-lane projections' events, so that the caller's stream -- the critical chain -- takes ONE
  // cross-stream wait between stage A and the pooling instead of one per lane
  for (int m = 0; m < 3; ++m)
    if (ca_done[m] && hipStreamWaitEvent(c.sts[3], ca_done[m], 0) != hipSuccess) return SDUMC_ELAUNCH;
  RET(link(c, 3, 0));
  if (pregen) {
    // The keep-bits of the NEXT call (call index + 2: sdumc_train_step's advance) into the OTHER set, on lane 3 -- idle from here to
    // the backward's early launch -- beside the latency-bound utterance-level stages: the Philox launches (VALU-bound, ~100 us of lane
    // time at C2 where they competed with the frame projections, 23 + 17 + 7 us here) leave the head of the step.  The other set's tag
    // is cleared before the set is refilled and written behind it, all on this lane; the next call -- the caller flips bits_phase --
    // reads that set, and its head launch finds the tag on the same lane, behind its fork.  Issued BEHIND the event the caller's
    // stream waits for in front of the Cross_Attention pooling: that wait must not include these launches.
    c.use(3);
    uint8_t* const nx = bits_set(c, 1);
    uint32_t* tag = reinterpret_cast<uint32_t*>(nx);
    // the NEXT call's dims (a ragged epoch announces them: sdumc_net_io.bits_next_dims) decide the layout of the set it will read
    const sdumc_net_dims& nd = c.io.bits_next_dims ? *c.io.bits_next_dims : c.d;
    Plan pn_own;
    if (c.io.bits_next_dims && !make_plan(nd, pn_own)) return SDUMC_EINVAL;
    const Plan& pn = c.io.bits_next_dims ? pn_own : pl;
    const sdumc_bits_shape shape_next = bits_shape(nd);
    auto mkdrop_next = [&](int site, int T, int s0) {
      sdumc_dropout d = mkdrop(c, site, c.d.p_frame, T, D, s0);
      d.samples = (uint32_t)nd.B;
      d.sample0 = (uint32_t)nd.sample0;
      return d;
    };
    const sdumc_dropout d0 = mkdrop_next(SITE_IN[0][0], pn.segs[0][0].T, 0);
    RET(sdumc_bits_tag_(&d0, tag, -1, &shape_next, c.st));
    const int order[3] = {0, 2, 1};
    for (int oi = 0; oi < 3; ++oi) {
      const int m = order[oi];
      for (const Seg& sg : pn.segs[m]) {
        sdumc_dropout d = mkdrop_next(SITE_IN[0][m], sg.T, sg.s0);
        uint8_t* outs[2] = {nx + bits_next_off(pn, 0, m) + sg.row0 * (D / 4), nx + bits_next_off(pn, 1, m) + sg.row0 * (D / 4)};
        RET(sdumc_dropout_bits_multi_ex_(&d, sg.V / pn.B, 2, SITE_IN[1][m] - SITE_IN[0][m], outs, nullptr, 2, nullptr, c.st));
      }
    }
    RET(sdumc_bits_tag_(&d0, tag, 2, &shape_next, c.st));
    c.use(0);
  }
  if (attn_multi_ok(c)) {
    RET(pool_fwd_multi(c, 1));
  } else {
    RET(fork_all(c));
    for (int m = 0; m < 3; ++m) {
      c.use(LANE_OF[m]);
      RET(pool_fwd(c, 1, m));
    }
    c.use(0);
    RET(join_all(c));
  }
  if (chain) {   // steps 9-12 and the outputs in one launch
    const sdumc_chain_args ca = chain_args(c, true, nullptr, false);
    mark(c.st, 3);
    RET(chain_launch(c, ca, 1));
    mark(c.st, 4);
    return SDUMC_OK;
  }
  // 9. cross_{audio,text,video}_mlp (model :338-340)
  {
    sdumc_gemm g = G_(SDUMC_NT, V * NQ, D, D, 3);
    for (int m = 0; m < 3; ++m) {
      g.A[m] = c.p(pl.ca_out) + (int64_t)m * V * NQ * D;
      g.B[m] = c.P + pm.cmlp0[m].w;
      g.bias[m] = c.P + pm.cmlp0[m].b;
      g.C[m] = c.p(pl.c1) + (int64_t)m * V * NQ * D;
    }
    g.lda = g.ldb = g.ldc = D;
    g.act = SDUMC_ACT_RELU;
    g.c_drop = mkdrop(c, SITE_CMLP0, c.d.p_mlp, NQ, D);
    g.c_drop_group_stride = 2;
    RET(run(c, g));
    g.N = H;
    for (int m = 0; m < 3; ++m) {
      g.A[m] = c.p(pl.c1) + (int64_t)m * V * NQ * D;
      g.B[m] = c.P + pm.cmlp3[m].w;
      g.bias[m] = c.P + pm.cmlp3[m].b;
      g.C[m] = c.p(pl.c) + (int64_t)m * V * NQ * H;
    }
    g.ldc = H;
    g.c_drop = mkdrop(c, SITE_CMLP1, c.d.p_mlp, NQ, H);
    RET(run(c, g));
  }
  // 10. modality-weighted sum (model :346-349)
  RET(sdumc_hweight_fwd(c.p(pl.c), c.p(pl.alpha), c.p(pl.h), V, c.st));
  // 11. cross_attention_mlp + cross_fc_att (model :352-354)
  {
    sdumc_dropout d0 = mkdrop(c, SITE_CATT0, c.d.p_mlp, 1, D), d1 = mkdrop(c, SITE_CATT1, c.d.p_mlp, 1, H);
    RET(lin_fwd(c, pm.catt0, c.p(pl.h), NQ * H, V, c.p(pl.e1), D, SDUMC_ACT_RELU, &d0));
    RET(lin_fwd(c, pm.catt3, c.p(pl.e1), D, V, c.p(pl.e2), H, SDUMC_ACT_RELU, &d1));
    RET(lin_fwd(c, pm.cross_fc_att, c.p(pl.e2), H, V, c.p(pl.beta), NQ, SDUMC_ACT_NONE, nullptr));
  }
  // 12. cross_fused_feat, regression head, RnC embedding (model :356-368)
  RET(sdumc_zpool_fwd(c.p(pl.h), c.p(pl.beta), c.p(pl.z), V, c.st));
  RET(lin_fwd(c, pm.fc_out_v, c.p(pl.z), H, V, c.p(pl.vals), 1, SDUMC_ACT_NONE, nullptr));
  RET(lin_fwd(c, pm.rnc0, c.p(pl.z), H, V, c.p(pl.r1), RD, SDUMC_ACT_RELU, nullptr));
  RET(lin_fwd(c, pm.rnc2, c.p(pl.r1), RD, V, c.p(pl.r), RD, SDUMC_ACT_NONE, nullptr));
  // outputs (model :370)
  {
    sdumc_copy_seg sg[5];
    int n = 0;
    if (c.io.vals) sg[n++] = {c.p(pl.vals), c.io.vals, 1, 1, V, 1};
    if (c.io.fused) sg[n++] = {c.p(pl.z), c.io.fused, H, H, V, H};
    if (c.io.rnc) sg[n++] = {c.p(pl.r), c.io.rnc, RD, RD, V, RD};
    if (c.io.text_hidden) sg[n++] = {c.p(pl.q) + 5 * D, c.io.text_hidden, NQ * D, D, V, D};
    if (c.io.cross_text) sg[n++] = {c.p(pl.c) + (int64_t)V * NQ * H, c.io.cross_text, NQ * H, NQ * H, V, NQ * H};
    if (n) RET(sdumc_copy2d_multi(sg, n, c.st));
  }
  return SDUMC_OK;
}

// backward of the pooling of site (k, m): dz (pre-tanh key gradient), dxd (pooling path), dq
int pool_bwd(const Ctx& c, int k, int m, const float* dout_base /* [V, nq, D] */, float* dq_base /* [V, nq, D] */,
             float* dq_sum = nullptr /* shared query: [nq, D] sum over the samples instead of dq */) {
  const Plan& pl = c.pl;
  const int nq = k == 0 ? 1 : NQ;
  for (const Seg& sg : pl.segs[m]) {
    sdumc_attnpool_bwd_t b;
    memset(&b, 0, sizeof(b));
    b.f = attn_desc(c, k, m, sg);
    const int64_t voff = (int64_t)sg.s0 * pl.B * nq * D;
    b.dout = dout_base + voff;
    b.dz = c.p(pl.dz[k][m]) + sg.row0 * D;
    b.dxd = c.p(pl.dxd[k][m]) + sg.row0 * D;
    if (c.h()) {
      b.dz = reinterpret_cast<float*>(c.ph(pl.dz[k][m], sg.row0 * D));
      b.dxd = reinterpret_cast<float*>(c.ph(pl.dxd[k][m], sg.row0 * D));
    }
    if (dxfold(c, m)) { b.dxd = nullptr; b.dout_masked = c.p(pl.dom[k][m]) + voff; }
    b.dq = dq_base + voff;
    b.dq_sum = dq_sum;
    b.workspace = c.scr;
    b.workspace_bytes = (size_t)pl.scratch_floats * sizeof(float);
    RET(sdumc_attnpool_bwd(&b, c.st));
  }
  return SDUMC_OK;
}

// input_proj backward of the sites [k0, k1) of modality m (grouped when both): dW = dz^T drop(x) (+ db), dxd += dz W
int keys_dx_sum(const Ctx& c, int m, int k, int rows_cap);
// parts: bit 0 = dW (off every critical path: feeds only the gradient bucket), bit 1 = dX
int keys_gemm_bwd_h(const Ctx& c, int m, int k0, int k1, int parts, int rows_cap = 0) {
  const Plan& pl = c.pl;
  if (parts & 1) {
    bool first = true;
    for (const Seg& sg : pl.segs[m]) {
      const int rows = sg.V * sg.T;
      sdumc_gemm_bf16 g = GH_(SDUMC_TN, D, D, rows, k1 - k0);
      for (int k = k0; k < k1; ++k) {
        const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
        g.A[k - k0] = c.ph(pl.dz[k][m], sg.row0 * D);
        g.B[k - k0] = c.d.train ? c.ph(pl.xd[k][m], sg.row0 * D) : c.ph(sg.x_off);
        g.C[k - k0] = c.G + L.w;
        g.colsum_a[k - k0] = c.G + L.b;
      }
      g.lda = g.ldb = g.ldc = D;
      g.b_row_mod = (!c.d.train && sg.x_samples < sg.V) ? sg.x_samples * sg.T : 0;
      g.accumulate = first ? 0 : 1;
      RET(run_h(c, g));
      first = false;
    }
  }
  if (!(parts & 2)) return SDUMC_OK;
  if (dxsum(c, m)) {      // (site 1 first: it writes dx)
    for (int k = k1 - 1; k >= k0; --k) RET(keys_dx_sum(c, m, k, rows_cap));
    return SDUMC_OK;
  }
  sdumc_gemm_bf16 g = GH_(SDUMC_NT, (int)pl.rows[m], D, D, k1 - k0);      // dxd += dz W: NT on the transposed weight copy
  for (int k = k0; k < k1; ++k) {
    const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
    g.A[k - k0] = c.ph(pl.dz[k][m]);
    g.B[k - k0] = c.ph(pl.wht, L.w);
    g.C[k - k0] = c.ph(pl.dxd[k][m]);
  }
  g.lda = g.ldb = g.ldc = D;
  g.accumulate = 1;
  g.c_bf16 = 1;
  if (rows_on() && parts == 2) {   // (bf16 C2 step 1.077-1.080 against 1.091-1.092 ms; the forward key projections the same way: 1.094-1.099, not used)
    sdumc_rows_problem q[2];
    for (int k = k0; k < k1; ++k) {
      sdumc_rows_problem& r = q[k - k0];
      memset(&r, 0, sizeof(r));
      r.A = reinterpret_cast<const float*>(g.A[k - k0]);
      r.B = reinterpret_cast<const float*>(g.B[k - k0]);
      r.C = reinterpret_cast<float*>(g.C[k - k0]);
      r.M = g.M;
      r.lda = r.ldb = r.ldc = D;
      r.accumulate = 1;
    }
    const int rc = sdumc_gemm_rows256_bf16_capped_(q, k1 - k0, rows_cap > 0 ? rows_cap : 0, c.st);
    if (rc != SDUMC_EINVAL) return rc;
  }
  return run_h(c, g);
}

// dW (+ db) of the input_proj layers of sites [k0, k1) of modality m, queued for the next grouped launch: one problem per layer,
// the runs of a modality (streams whose text lengths differ) as its K segments
bool keys_dw_groupable(const Ctx& c, int m) {
  if (c.pl.segs[m].size() > 2) return false;
  const int min_mod = c.h() ? 64 : 16;
  for (const Seg& sg : c.pl.segs[m]) {
    const bool shared = c.h() ? (!c.d.train && sg.x_samples < sg.V) : sg.x_samples < sg.V;
    if (shared && sg.x_samples * sg.T < min_mod) return false;
  }
  return true;
}
void keys_dw_queue(const Ctx& c, int m, int k0, int k1) {
  const Plan& pl = c.pl;
  for (int k = k0; k < k1; ++k) {
    const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
    sdumc_gg_problem q;
    memset(&q, 0, sizeof(q));
    int s = 0;
    for (const Seg& sg : pl.segs[m]) {
      q.K[s] = sg.V * sg.T;
      if (c.h()) {     // bf16 storage: dz and the materialised masked frames xd (train) / the shared x (eval)
        q.A[s] = reinterpret_cast<const float*>(c.ph(pl.dz[k][m], sg.row0 * D));
        q.B[s] = reinterpret_cast<const float*>(c.d.train ? c.ph(pl.xd[k][m], sg.row0 * D) : c.ph(sg.x_off));
        q.b_row_mod[s] = (!c.d.train && sg.x_samples < sg.V) ? sg.x_samples * sg.T : 0;
      } else {
        q.A[s] = c.p(pl.dz[k][m]) + sg.row0 * D;
        q.B[s] = c.p(sg.x_off);
        q.b_row_mod[s] = sg.x_samples < sg.V ? sg.x_samples * sg.T : 0;
        const sdumc_dropout dd = in_drop(c, k, m, sg.T, sg.s0, sg.row0);
        q.b_bits[s] = dd.enabled ? dd.bits : nullptr;
        if (dd.enabled) q.b_scale = dd.scale;
      }
      ++s;
    }
    if (q.b_scale == 0.f) q.b_scale = 1.f;
    q.C = c.G + L.w;
    q.colsum_a = c.G + L.b;
    q.M = q.N = q.lda = q.ldb = q.ldc = D;
    q.bits_qw = D / 4;
    (c.h() ? c.ggh : c.gg).push_back(q);
  }
}

// dxd += dz W of sites [k0, k1) of modality m as problems of a rows launch (dxfold: dxd = dz W + the pooling term, one problem per run)
int keys_dx_rows(const Ctx& c, int m, int k0, int k1, sdumc_rows_problem* q) {
  int n = 0;
  const bool fold = dxfold(c, m);
  for (int k = k0; k < k1; ++k) {
    const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
    if (fold) {
      const int nq = k == 0 ? 1 : NQ;
      for (const Seg& sg : c.pl.segs[m]) {
        sdumc_rows_problem& r = q[n++];
        memset(&r, 0, sizeof(r));
        r.A = c.p(c.pl.dz[k][m]) + sg.row0 * D;
        r.B = c.P + L.w;
        r.C = c.p(c.pl.dxd[k][m]) + sg.row0 * D;
        r.M = sg.V * sg.T;
        r.lda = r.ldb = r.ldc = D;
        r.pool_w = c.p(c.pl.attn[k][m]) + sg.row0 * nq;
        r.pool_g = c.p(c.pl.dom[k][m]) + (int64_t)sg.s0 * c.pl.B * nq * D;
        r.pool_nq = nq;
        r.pool_T = sg.T;
      }
      continue;
    }
    sdumc_rows_problem& r = q[n++];
    memset(&r, 0, sizeof(r));
    r.A = c.p(c.pl.dz[k][m]);
    r.B = c.P + L.w;
    r.C = c.p(c.pl.dxd[k][m]);
    r.M = (int)c.pl.rows[m];
    r.lda = r.ldb = r.ldc = D;
    r.accumulate = 1;
  }
  return n;
}

// dxsum: dx of modality m (+)= sum over the streams of keep . (dz W + pooling term) of site k, one rows launch (the Cross_Attention
// site of a modality always runs first and writes, the FRA2UTT site adds onto it)
int keys_dx_sum_problems(const Ctx& c, int m, int k, sdumc_rows_problem* q) {
  const Plan& pl = c.pl;
  const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
  const int nq = k == 0 ? 1 : NQ;
  int n = 0;
  for (const Seg& sg : pl.segs[m]) {
    sdumc_rows_problem& r = q[n++];
    memset(&r, 0, sizeof(r));
    const int fold = sg.V / sg.x_samples;              // streams that share the run's frames: 2 (audio, video), 1 (the text slot: dx rows = virtual rows)
    r.A = c.p(pl.dz[k][m]) + sg.row0 * D;
    r.B = c.P + L.w;
    r.C = c.p(pl.dx[m][sg.s0]);
    if (c.h()) {      // bf16 dz / dx; B = the transposed bf16 weight copy (its rows are the output columns)
      r.A = reinterpret_cast<const float*>(c.ph(pl.dz[k][m], sg.row0 * D));
      r.B = reinterpret_cast<const float*>(c.ph(pl.wht, L.w));
      r.C = reinterpret_cast<float*>(c.ph(pl.dx[m][sg.s0]));
    }
    r.M = sg.V * sg.T;
    r.lda = r.ldb = r.ldc = D;
    r.accumulate = k == 0 ? 1 : 0;
    r.pool_w = c.p(pl.attn[k][m]) + sg.row0 * nq;
    r.pool_g = c.p(pl.dom[k][m]) + (int64_t)sg.s0 * pl.B * nq * D;
    r.pool_nq = nq;
    r.pool_T = sg.T;
    r.fold = fold;
    const sdumc_dropout dr = in_drop(c, k, m, sg.T, sg.s0, sg.row0);
    if (dr.enabled) { r.c_bits = dr.bits; r.c_scale = dr.scale; }
  }
  return n;
}
int dx_sum_launch(const Ctx& c, const sdumc_rows_problem* q, int n, int rows_cap) {
  return c.h() ? sdumc_gemm_rows256_bf16_capped_(q, n, rows_cap > 0 ? rows_cap : 0, c.st) : sdumc_gemm_rows256_capped_(q, n, rows_cap > 0 ? rows_cap : 0, c.st);
}
int keys_dx_sum(const Ctx& c, int m, int k, int rows_cap) {
  sdumc_rows_problem q[4];
  const int n = keys_dx_sum_problems(c, m, k, q);
  return dx_sum_launch(c, q, n, rows_cap);
}

// rows_cap (fp32 dX through the persistent rows launch): > 0 = at most that many workgroups, < 0 = the tiled 64x64 kernel instead
int keys_gemm_bwd(const Ctx& c, int m, int k0, int k1, int parts = 3, int rows_cap = 0) {
  if (c.h()) return keys_gemm_bwd_h(c, m, k0, k1, parts, 0);      // (bf16 storage: the cap measured 0.957-0.966 vs 0.951-0.959 ms: every CU)
  const Plan& pl = c.pl;
  // dW: one grouped GEMM per run (runs differ in their x buffer), later runs accumulate
  bool first = true;
  if (parts & 1)
  for (const Seg& sg : pl.segs[m]) {
    const int rows = sg.V * sg.T;
    sdumc_gemm g = G_(SDUMC_TN, D, D, rows, k1 - k0);
    for (int k = k0; k < k1; ++k) {
      const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
      g.A[k - k0] = c.p(pl.dz[k][m]) + sg.row0 * D;
      g.B[k - k0] = c.p(sg.x_off);
      g.C[k - k0] = c.G + L.w;
      g.colsum_a[k - k0] = c.G + L.b;
      g.ab_drop_bits[k - k0] = in_drop(c, k, m, sg.T, sg.s0, sg.row0).bits;
    }
    g.lda = g.ldb = g.ldc = D;
    g.b_row_mod = sg.x_samples < sg.V ? sg.x_samples * sg.T : 0;
    g.b_drop = in_drop(c, k0, m, sg.T, sg.s0, sg.row0);
    g.ab_drop_group_stride = SITE_IN[1][m] - SITE_IN[0][m];
    g.accumulate = first ? 0 : 1;
    g.bf16 = c.d.bf16 ? 1 : 0;
    RET(run(c, g));
    first = false;
  }
  if (!(parts & 2)) return SDUMC_OK;
  // dxd += dz W (the key-projection path joins the pooling path)
  if (dxsum(c, m)) {      // (site 1 first: it writes dx)
    for (int k = k1 - 1; k >= k0; --k) RET(keys_dx_sum(c, m, k, rows_cap));
    return SDUMC_OK;
  }
  if (dxfold(c, m) || (parts == 2 && rows_cap >= 0 && rows_on() && rows_ok(c))) {   // (the early dW + dX pair keeps the small-footprint kernels: it runs beside
                                                         //  the co-resident utterance-level stage, which a persistent launch would stall)
    sdumc_rows_problem q[8];
    const int n = keys_dx_rows(c, m, k0, k1, q);
    const int rc = sdumc_gemm_rows256_capped_(q, n, rows_cap > 0 ? rows_cap : 0, c.st);
    if (rc != SDUMC_EINVAL || dxfold(c, m)) return rc;      // (a shape the rows launch refuses -- 2 GiB of rows -- takes the tiled kernel below;
                                                          //  dxfold: the plan checked the shapes, a refusal is an error -- nothing else wrote dxd)
  }
  sdumc_gemm g = G_(SDUMC_NN, (int)pl.rows[m], D, D, k1 - k0);
  for (int k = k0; k < k1; ++k) {
    const Lin& L = k == 0 ? c.pm.fra_proj[m] : c.pm.ca_in[m];
    g.A[k - k0] = c.p(pl.dz[k][m]);
    g.B[k - k0] = c.P + L.w;
    g.C[k - k0] = c.p(pl.dxd[k][m]);
  }
  g.lda = g.ldb = g.ldc = D;
  g.accumulate = 1;
  g.bf16 = c.d.bf16 ? 1 : 0;
  return run(c, g);
}

// phases: bit 0 = the utterance-level part (finishes every gradient in [0, pm.early)), bit 1 = the frame-level part
int backward(const Ctx& c, const sdumc_net_grads& og, int phases = 3) {
  const Plan& pl = c.pl;
  const ParamMap& pm = c.pm;
  const int B = pl.B, S = pl.S, V = pl.V;
  const int din[3] = {c.d.da, c.d.dt, c.d.dv};
  const float s_mlp = c.d.train ? 1.0f / (1.0f - (float)c.d.p_mlp) : 1.0f;
  // Early Cross_Attention key-projection backward (bit m of c.bgb): a modality that owns a side lane keeps going on that lane --
  // the caller's stream waits only for the pooling backward before it, so the GEMMs run beside steps 7'-3' and, in phased
  // calls (the data-parallel step), beside whatever the caller does between the phases; a modality on the caller's stream
  // goes through lane 3, which phased calls cannot use for frame-level work (phase 0 ends by waiting for lane 3).
  // With the grouped weight-gradient launches lane 3 carries persistent kernels that fill the chip for 0.1-0.2 ms each, so
  // nothing the critical path waits for may queue behind them: the early dX then always takes the modality's own lane.
  const bool ggf = gg_frame(c);
  // The early modality keeps its per-layer dW + dX pair (small-footprint kernels that run beside the utterance-level stage
  // 7'-3'); measured in the grouped mode, fp32 C2: 1.787 ms, against 1.816 with only its dX early and the dW grouped, 1.833 with
  // no early work at all, 1.831 with the pair issued behind that stage instead of beside it.
  int own_lane = 0;
  for (int m = 0; m < 3; ++m)
    if (phases != 3 && (c.bgb & (1 << m)) && c.multi && LANE_OF[m] != 0) own_lane |= 1 << m;
  // (in a single call the lane-3 route measured 0.15 % faster than the own-lane route; phased calls gain 0.6 % from the latter)
  // (dxsum: audio's AND video's Cross_Attention sites go early, in one launch -- each site is a launch of its own there, the first of
  //  a modality writes dx, and two of them back to back on video's lane made it the longest: 1.2917-1.2925 ms against 1.307-1.308
  //  with audio's alone, 1.2946-1.2986 with all three)
  const int bgb = phases == 3 ? ((c.bgb && dxsum(c, 0) && dxsum(c, 2)) ? 5 : c.bgb) : own_lane;
  // grouped mode: the dW of the Cross_Attention input_proj layers rides in the launch right behind the (grouped) pooling backward
  // of phase 0 (bit m of ca_dw_mask), the dW of the FRA2UTT ones in the launch behind the FRA2UTT pooling backward of phase 1
  const bool ca_dw_grouped = ggf && attn_multi_ok(c);
  int ca_dw_mask = 0, fra_dw_mask = 0;
  for (int m = 0; m < 3; ++m) {
    // Only the dX of the early modality's Cross_Attention site runs early (it feeds that modality's mask-sum); its dW rides in the
    // grouped launch like every other weight gradient.  Round 3 measured the per-layer dW + dX PAIR ahead (1.787 vs 1.816 ms);
    // with the round-4 pooling kernels and reduce launches the order flipped: fp32 C2 1.690-1.698 vs 1.701-1.706 ms with the pair,
    // bf16 storage 0.956-0.959 vs 1.019-1.021 (three alternations on one box).
    if (ca_dw_grouped && keys_dw_groupable(c, m)) ca_dw_mask |= 1 << m;
    if (ggf && keys_dw_groupable(c, m)) fra_dw_mask |= 1 << m;
  }
  // early_done[m]: lane 3 has finished modality m's early key-projection backward (dxd of its Cross_Attention site).  The
  // modality's own lane waits for exactly this event before its mask-sum -- not for whatever else lane 3 has been handed by
  // then (the utterance-level dW batches, the other key-projection dW GEMMs: ~0.25 ms of work that only feeds the bucket).
  hipEvent_t early_done[3] = {nullptr, nullptr, nullptr};
  auto record_early = [&](int m) -> int {
    if (c.sts[3] == c.sts[LANE_OF[m]]) return SDUMC_OK;
    early_done[m] = next_event(c);
    return hipEventRecord(early_done[m], c.sts[3]) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
  };
  // the early key-projection backward of the Cross_Attention sites (bit m of bgb), ordered after what lane 0 has issued so far
  auto early_keys = [&]() -> int {
    constexpr int ecap = 160;      // (workgroups of the merged early launch: 160 1.3266-1.3297 ms, 128 1.3304-1.3345, 96 1.3645-1.3712, 200 / 256 +1.5 %)
    bool all_sum = true;
    for (int m = 0; m < 3; ++m) all_sum = all_sum && (!(bgb & (1 << m)) || dxsum(c, m));
    if (all_sum && !own_lane && bgb && !(bgb & ~ca_dw_mask)) {      // the early Cross_Attention sites of every early modality: ONE rows launch on lane 3
      sdumc_rows_problem q[8];
      int n = 0;
      for (int m = 0; m < 3; ++m)
        if (bgb & (1 << m)) n += keys_dx_sum_problems(c, m, 1, q + n);
      RET(link(c, 0, 3));
      c.use(3);
      bool one_fold = true;      // (a bf16 launch takes one fold: runs of shared and of separate frames go in separate launches there)
      for (int i = 1; i < n; ++i) one_fold = one_fold && q[i].fold == q[0].fold;
      if (c.h() && !one_fold) {
        for (int i = 0; i < n; ++i) RET(dx_sum_launch(c, q + i, 1, ecap));
      } else
      RET(dx_sum_launch(c, q, n, ecap));
      for (int m = 0; m < 3; ++m)
        if (bgb & (1 << m)) RET(record_early(m));
      c.use(0);
      return SDUMC_OK;
    }
    for (int m = 0; m < 3; ++m) {
      if (!(bgb & (1 << m))) continue;
      const int lane = (own_lane & (1 << m)) ? LANE_OF[m] : 3;
      RET(link(c, 0, lane));
      c.use(lane);
      // The early dX runs as the persistent rows launch on HALF of the CUs: a full-chip launch holds every CU's register file, so
      // the FRA2UTT pooling backward of the three modality lanes -- the next link of the critical chain, HBM-bound, issued while
      // this launch is still running -- crawled beside it (86-104 us instead of ~60); the early dX itself has slack until its
      // modality's mask-sum.  Measured, fp32 C2, three alternations on one box: 128 workgroups 1.670-1.680 ms, 64: 1.675-1.679,
      // 256 (every CU): 1.686-1.689, the tiled 64x64 kernel: 1.689-1.695.
      // (round 4, rows launch on the bf16 matrix pipe: 160 workgroups 1.414-1.427 against 128 1.424-1.438 and 192 1.425-1.438)
      constexpr int early_dx = 160;
      RET(keys_gemm_bwd(c, m, 1, 2, (ca_dw_mask & (1 << m)) ? 2 : 3, early_dx));
      if (lane == 3) RET(record_early(m));
      c.use(0);
    }
    return SDUMC_OK;
  };
  if (phases & 1) {
  // every live gradient tensor is overwritten below when all five output gradients are given
  if (!og.d_vals || !og.d_fused || !og.d_rnc || !og.d_text_hidden || !og.d_cross_text) RET(sdumc_fill(c.G, 0.f, pm.live, c.st));

  const bool chain = use_chain(c);
  if (chain) {
    // 12'-9' in one launch (chain.hip); the weight gradients of these layers are queued for lane 3 as before
    const sdumc_chain_args ca = chain_args(c, false, &og, false);
    RET(chain_launch(c, ca, 2));
    mark(c.st, 6);
    const int M7 = V * NQ;
    if (og.d_rnc) {
      RET(lin_bwd(c, pm.rnc2, og.d_rnc, RD, c.p(pl.r1), RD, V, nullptr, 0, 0));
      RET(lin_bwd(c, pm.rnc0, c.p(pl.d_r1), RD, c.p(pl.z), H, V, nullptr, 0, 0));
    }
    if (og.d_vals) RET(lin_bwd(c, pm.fc_out_v, og.d_vals, 1, c.p(pl.z), H, V, nullptr, 0, 0));
    RET(lin_bwd(c, pm.cross_fc_att, c.p(pl.d_beta), NQ, c.p(pl.e2), H, V, nullptr, 0, 0));
    RET(lin_bwd(c, pm.catt3, c.p(pl.d_e2), H, c.p(pl.e1), D, V, nullptr, 0, 0));
    RET(lin_bwd(c, pm.catt0, c.p(pl.d_e1), D, c.p(pl.h), NQ * H, V, nullptr, 0, 0));
    GroupPtrs q3 = {c.p(pl.d_c), (int64_t)M7 * H, H, c.p(pl.c1), (int64_t)M7 * D, D, nullptr, 0, 0};
    RET(lin_bwd_grouped(c, pm.cmlp3, 3, M7, q3));
    GroupPtrs q0 = {c.p(pl.d_c1), (int64_t)M7 * D, D, c.p(pl.ca_out), (int64_t)M7 * D, D, nullptr, 0, 0};
    RET(lin_bwd_grouped(c, pm.cmlp0, 3, M7, q0));
  } else {
  // 12'. heads: r = L2(relu(L0(z))), vals = fc_out_v(z), plus the external gradient of cross_fused_feat
  float* d_z = c.p(pl.d_z);
  if (og.d_rnc) {
    RET(lin_bwd(c, pm.rnc2, og.d_rnc, RD, c.p(pl.r1), RD, V, c.p(pl.d_r1), RD, 0, c.p(pl.r1), 1.0f));
    RET(lin_bwd(c, pm.rnc0, c.p(pl.d_r1), RD, c.p(pl.z), H, V, d_z, H, 0));
  } else {
    RET(sdumc_fill(d_z, 0.f, (int64_t)V * H, c.st));
  }
  if (og.d_vals) RET(lin_bwd(c, pm.fc_out_v, og.d_vals, 1, c.p(pl.z), H, V, d_z, H, 1));
  //      (d_z + the external gradient of cross_fused_feat, added inside the kernel instead of by an axpy launch)
  RET(sdumc_zpool_bwd_add_(c.p(pl.h), c.p(pl.beta), d_z, og.d_fused, c.p(pl.d_h), c.p(pl.d_beta), V, c.st));
  // 11'. cross_fc_att, cross_attention_mlp
  RET(lin_bwd(c, pm.cross_fc_att, c.p(pl.d_beta), NQ, c.p(pl.e2), H, V, c.p(pl.d_e2), H, 0, c.p(pl.e2), s_mlp));
  RET(lin_bwd(c, pm.catt3, c.p(pl.d_e2), H, c.p(pl.e1), D, V, c.p(pl.d_e1), D, 0, c.p(pl.e1), s_mlp));
  RET(lin_bwd(c, pm.catt0, c.p(pl.d_e1), D, c.p(pl.h), NQ * H, V, c.p(pl.d_h), NQ * H, 1));
  // 10'. modality-weighted sum; the external gradient of cross_hiddens[:,1] joins here
  RET(sdumc_hweight_bwd(c.p(pl.c), c.p(pl.alpha), c.p(pl.d_h), og.d_cross_text, c.p(pl.d_c), c.p(pl.d_alpha), V, s_mlp,
                        c.st));   // d_c comes out already masked: gradient w.r.t. the pre-activation of cross_*_mlp.3
  // 9'. cross_{audio,text,video}_mlp
  {
    const int M = V * NQ;
    GroupPtrs q3 = {c.p(pl.d_c), (int64_t)M * H, H, c.p(pl.c1), (int64_t)M * D, D, c.p(pl.d_c1), (int64_t)M * D, D,
                    c.p(pl.c1), s_mlp};
    RET(lin_bwd_grouped(c, pm.cmlp3, 3, M, q3));
    GroupPtrs q0 = {c.p(pl.d_c1), (int64_t)M * D, D, c.p(pl.ca_out), (int64_t)M * D, D, c.p(pl.d_ca_out), (int64_t)M * D, D};
    RET(lin_bwd_grouped(c, pm.cmlp0, 3, M, q0));
  }
  }   // !chain
  // 8'. the three Cross_Attention blocks
  const bool grouped = attn_multi_ok(c);
  // batch 1 of the weight gradients (heads, cross_attention_mlp, cross_*_mlp) -- in the grouped mode together with the
  // Cross_Attention input_proj dW, right behind the pooling backward that produces their dz
  if (!ca_dw_grouped) RET(flush_dw(c));
  if (grouped) {   // one grouped launch on the caller's stream, then the early key-projection backwards leave for their lanes
    sdumc_attnpool_bwd_t bb[4];
    int n = 0;
    float* ws = fra_fold(c) ? c.p(pl.dq_ws) : c.scr;   // folded: the slabs outlive this launch (stage 7'-3' sums them)
    const int order[3] = {0, 2, 1};
    for (int oi = 0; oi < 3; ++oi) {
      const int m = order[oi];
      for (const Seg& sg : pl.segs[m]) {
        sdumc_attnpool_bwd_t& b = bb[n++];
        memset(&b, 0, sizeof(b));
        b.f = attn_desc(c, 1, m, sg);
        const int64_t voff = (int64_t)sg.s0 * pl.B * NQ * D;
        b.dout = c.p(pl.d_ca_out) + (int64_t)m * V * NQ * D + voff;
        b.dq = c.p(pl.d_qp) + (int64_t)m * V * NQ * D + voff;
        if (c.h()) {
          b.dz = reinterpret_cast<float*>(c.ph(pl.dz[1][m], sg.row0 * D));
          b.dxd = reinterpret_cast<float*>(c.ph(pl.dxd[1][m], sg.row0 * D));
        } else {
          b.dz = c.p(pl.dz[1][m]) + sg.row0 * D;
          b.dxd = c.p(pl.dxd[1][m]) + sg.row0 * D;
        }
        if (dxfold(c, m)) { b.dxd = nullptr; b.dout_masked = c.p(pl.dom[1][m]) + voff; }
        const size_t bytes = sdumc_attnpool_bwd_workspace_bytes(sg.V, sg.T, NQ);
        b.workspace = ws;
        b.workspace_bytes = bytes;
        if (fra_fold(c)) {      // the clustered stage 7'-3' behind this launch sums the chunks
          b.f.partial_only = 1;
          c.dq_part[m] = ws;
          c.dq_nchunk[m] = (sg.T + 63) / 64;
        }
        ws += (bytes / sizeof(float) + 63) / 64 * 64;
      }
    }
    RET(sdumc_attnpool_bwd_multi(bb, n, c.st));
    for (int m = 0; m < 3; ++m)
      if (ca_dw_mask & (1 << m)) keys_dw_queue(c, m, 1, 2);
    RET(early_keys());
    // (no flush here: a persistent launch now would hold every CU's registers while the latency-bound utterance-level stage
    //  7'-3' -- 256 co-resident workgroups -- is trying to start: measured +135 us on that stage.  The queued problems leave
    //  with the FRA2UTT ones, beside the dX products of the frame-level part.)
  }
  if (!grouped) RET(fork_all(c));
  hipEvent_t pooled[3] = {nullptr, nullptr, nullptr};   // per side lane: its pooling backward is done (partial join)
  for (int m = 0; !grouped && m < 3; ++m) {
    c.use(LANE_OF[m]);
    RET(pool_bwd(c, 1, m, c.p(pl.d_ca_out) + (int64_t)m * V * NQ * D, c.p(pl.d_qp) + (int64_t)m * V * NQ * D));
    if (own_lane & bgb & (1 << m)) {   // the input_proj backward has everything it needs: stay on this lane, beside 7'-3'
      pooled[LANE_OF[m]] = next_event(c);
      if (hipEventRecord(pooled[LANE_OF[m]], c.st) != hipSuccess) return SDUMC_ELAUNCH;
      RET(keys_gemm_bwd(c, m, 1, 2));
    } else if (bgb & (1 << m)) {       // same, through the background lane
      RET(link(c, LANE_OF[m], 3));
      c.use(3);
      RET(keys_gemm_bwd(c, m, 1, 2));
      RET(record_early(m));
    }
  }
  c.use(0);
  for (int lane = 1; !grouped && lane <= 2; ++lane) {
    if (pooled[lane]) {
      if (hipStreamWaitEvent(c.sts[0], pooled[lane], 0) != hipSuccess) return SDUMC_ELAUNCH;
    } else {
      RET(link(c, lane, 0));
    }
  }
  if (chain) {
    // 7'-3' in one launch; dW of these layers queued for lane 3
    const sdumc_chain_args ca = chain_args(c, false, &og, true);
    mark(c.st, 7);
    RET(chain_launch(c, ca, 3));
    mark(c.st, 8);
    const int M7 = V * NQ;
    {
      sdumc_gemm gw = G_(SDUMC_TN, D, D, M7, 3);
      for (int m = 0; m < 3; ++m) {
        gw.A[m] = c.p(pl.d_qp) + (int64_t)m * M7 * D;
        gw.B[m] = c.p(pl.q);
        gw.C[m] = c.G + pm.ca_q[m].w;
        gw.colsum_a[m] = c.G + pm.ca_q[m].b;
      }
      gw.lda = gw.ldb = gw.ldc = D;
      c.deferred.push_back(gw);
    }
    GroupPtrs qq = {c.p(pl.d_q), D, NQ * D, c.p(pl.qin), (int64_t)V * D, D, nullptr, 0, 0};
    RET(lin_bwd_grouped(c, pm.query, 7, V, qq));
    RET(lin_bwd(c, pm.fc_att, c.p(pl.d_alpha), 3, c.p(pl.att2), D, V, nullptr, 0, 0));
    RET(lin_bwd(c, pm.att3, c.p(pl.d_att2), D, c.p(pl.att1), D, V, nullptr, 0, 0));
    RET(lin_bwd(c, pm.att0, c.p(pl.d_att1), D, c.p(pl.u), 3 * D, V, nullptr, 0, 0));
    GroupPtrs q3 = {c.p(pl.d_u), D, 3 * D, c.p(pl.u1), (int64_t)V * D, D, nullptr, 0, 0};
    RET(lin_bwd_grouped(c, pm.umlp3, 3, V, q3));
    GroupPtrs q0 = {c.p(pl.d_u1), (int64_t)V * D, D, c.p(pl.hpre), (int64_t)V * D, D, nullptr, 0, 0};
    RET(lin_bwd_grouped(c, pm.umlp0, 3, V, q0));
  } else {
  // 7'. query_proj: dW/db per modality, d_q = sum_m d_qp[m] W_q[m]
  {
    const int M = V * NQ;
    sdumc_gemm gw = G_(SDUMC_TN, D, D, M, 3);
    for (int m = 0; m < 3; ++m) {
      gw.A[m] = c.p(pl.d_qp) + (int64_t)m * M * D;
      gw.B[m] = c.p(pl.q);
      gw.C[m] = c.G + pm.ca_q[m].w;
      gw.colsum_a[m] = c.G + pm.ca_q[m].b;
    }
    gw.lda = gw.ldb = gw.ldc = D;
    c.deferred.push_back(gw);
      for (int m = 0; m < 3; ++m) {
      sdumc_gemm gx = G_(SDUMC_NN, M, D, D);
      gx.A[0] = c.p(pl.d_qp) + (int64_t)m * M * D;
      gx.lda = D;
      gx.B[0] = c.P + pm.ca_q[m].w;
      gx.ldb = D;
      gx.C[0] = c.p(pl.d_q);
      gx.ldc = D;
      gx.accumulate = m > 0;
      RET(run(c, gx));
    }
  }
  // 6'. the 7 query MLPs
  {
    //   the external gradient of text_hidden (= q[:, 5]) joins d_q inside the same launch
    RET(sdumc_relu_drop_bwd_add_(c.p(pl.d_q), c.p(pl.q), s_mlp, c.p(pl.d_q), 7LL * V * D, og.d_text_hidden, NQ * D, 5 * D, D,
                                 c.st));
    GroupPtrs qq = {c.p(pl.d_q), D, NQ * D, c.p(pl.qin), (int64_t)V * D, D, c.p(pl.d_qin), (int64_t)V * D, D};
    RET(lin_bwd_grouped(c, pm.query, 7, V, qq));
  }
  if (!gg_utt(c)) RET(flush_dw(c));   // batch 2: query_proj, the query MLPs (grouped mode: they wait for batch 3)
  // 5'. fusion algebra (d_alpha already holds the second-level contribution)
  RET(sdumc_fusion_bwd(c.p(pl.u), c.p(pl.alpha), c.p(pl.d_qin), c.p(pl.d_u), c.p(pl.d_alpha), V, c.st));
  // 4'. fc_att, attention_mlp
  RET(lin_bwd(c, pm.fc_att, c.p(pl.d_alpha), 3, c.p(pl.att2), D, V, c.p(pl.d_att2), D, 0, c.p(pl.att2), s_mlp));
  RET(lin_bwd(c, pm.att3, c.p(pl.d_att2), D, c.p(pl.att1), D, V, c.p(pl.d_att1), D, 0, c.p(pl.att1), s_mlp));
  //   d_u = (fusion part + att0 part) masked by u: the accumulate runs first, then the mask
  RET(lin_bwd(c, pm.att0, c.p(pl.d_att1), D, c.p(pl.u), 3 * D, V, c.p(pl.d_u), 3 * D, 1, c.p(pl.u), s_mlp));
  // 3'. audio/text/video_mlp
  {
    GroupPtrs q3 = {c.p(pl.d_u), D, 3 * D, c.p(pl.u1), (int64_t)V * D, D, c.p(pl.d_u1), (int64_t)V * D, D,
                    c.p(pl.u1), s_mlp};
    RET(lin_bwd_grouped(c, pm.umlp3, 3, V, q3));
    GroupPtrs q0 = {c.p(pl.d_u1), (int64_t)V * D, D, c.p(pl.hpre), (int64_t)V * D, D, c.p(pl.d_hpre), (int64_t)V * D, D};
    RET(lin_bwd_grouped(c, pm.umlp0, 3, V, q0));
  }
  }   // !chain
  // batch 3: fc_att, attention_mlp, audio/text/video_mlp.  Grouped mode with the frame-level part following in this call:
  // they stay queued and ride with the FRA2UTT input_proj dW in one launch (below).
  if (!(ggf && (phases & 2))) RET(flush_dw(c));
  // the dW GEMMs of this part ran on lane 3: after this link [0, pm.early) is final on the caller's stream.  When the
  // frame-level part follows in the same call the link at its end does the same job.
  if (!(phases & 2)) RET(link(c, 3, 0));
  }   // phases & 1
  if (!(phases & 2)) return SDUMC_OK;
  // 2'+1'. three independent per-modality chains, one per lane:
  //   fra2utt_m pooling backward (the shared context vector's gradient = sum of the per-sample dq)
  //   -> input_proj backward of both sites (grouped) -> dx = sum of the (up to) four masked paths into the
  //   projected features -> frame_dim_reshape_m: dW = dx^T feat with db fused
  RET(fork_all(c));
  for (int m = 0; m < 3; ++m) {       // pass 1: the pooling backward of every modality (produces dz of the FRA2UTT site)
    c.use(LANE_OF[m]);
    float* dq = c.p(pl.dq_fra) + (int64_t)m * V * D;
    // (the context vector is one query shared by every sample: its gradient, the sum over the samples' dq, leaves the pooling
    //  backward's own reduce launch -- one run per modality; two runs (unequal text lengths) keep dq + a column sum)
    float* ctx_grad = pl.segs[m].size() == 1 ? c.G + pm.fra_ctx[m] : nullptr;
    RET(pool_bwd(c, 0, m, c.p(pl.d_hpre) + (int64_t)m * V * D, dq, ctx_grad));
    if (!ctx_grad) RET(colsum(c, dq, V, D, D, c.G + pm.fra_ctx[m], 0));
    mark(c.st, 12 + 5 * m);      // (debug marks 12..26: this modality's lane, frame-level backward)
    if (fra_dw_mask & (1 << m)) {
      keys_dw_queue(c, m, 0, 1);
      if (!(ca_dw_mask & (1 << m)) && !(bgb & (1 << m))) keys_dw_queue(c, m, 1, 2);   // (the Cross_Attention one, when it did not leave in phase 0)
      if (!c.capturing) RET(link(c, LANE_OF[m], 3));   // lane 3 (the grouped launch below) needs this lane's dz
    }
  }
  if (ggf) {   // every queued utterance-level dW and the input_proj dW: one launch on lane 3, beside the dX products below
    c.use(0);
    // (under hipGraph capture three lanes -> lane 3 -> lane 0 is the dependency pattern hipStreamEndCapture cannot take on this
    //  stack: the lanes meet on the caller's stream instead and fork again)
    if (c.capturing) RET(join_all(c));
    RET(flush_dw(c));
    if (c.capturing) RET(fork_all(c));
  }
  const bool sum_in_dx_m[3] = {dxsum(c, 0), dxsum(c, 1), dxsum(c, 2)};      // the rows launches add keep . dxd straight into dx: no mask-sum launch, and the FRA2UTT site's
                                        // launch (which adds onto dx) has to follow the modality's early Cross_Attention one
  auto await_early = [&](int m) -> int {
    if (!(bgb & ~own_lane & (1 << m))) return SDUMC_OK;   // dxd of this modality's Cross_Attention site (issued early on lane 3)
    if (early_done[m]) return hipStreamWaitEvent(c.st, early_done[m], 0) == hipSuccess ? SDUMC_OK : SDUMC_ELAUNCH;
    return link(c, 3, LANE_OF[m]);
  };
  for (int m = 0; m < 3; ++m) {       // pass 2: dX of the key projections, the mask-sum, the frame projection's dW
    c.use(LANE_OF[m]);
    const bool sum_in_dx = sum_in_dx_m[m];
    if (sum_in_dx) RET(await_early(m));
    {
      const int k1 = (bgb & (1 << m)) ? 1 : 2;
      // which of this modality's key-projection dW products are already on their way in a grouped launch
      const bool fra_q = (fra_dw_mask & (1 << m)) != 0;
      const bool ca_q = k1 == 2 && (((ca_dw_mask | fra_dw_mask) & (1 << m)) != 0);
      // bit m: modality m's key-projection dW (off the dz -> dX -> mask-sum -> frame dW chain) runs on lane 3 (ungrouped modes).
      // (re-measured with the clustered utterance-level kernels: fp32 1.923 ms with none on lane 3 vs 1.930 with audio's;
      //  bf16 storage 1.148 vs 1.128 -- so the default follows the mode)
      const int dw_off = c.h() ? 1 : 0;
      if (fra_q) {
        if (k1 == 2 && !ca_q) RET(keys_gemm_bwd(c, m, 1, 2, 1));     // (never in practice: both sites are groupable or neither)
        RET(keys_gemm_bwd(c, m, 0, k1, 2));
      } else if (c.multi && !c.capturing && (dw_off & (1 << m))) {
        RET(link(c, LANE_OF[m], 3));
        c.use(3);
        RET(keys_gemm_bwd(c, m, 0, k1, 1));
        c.use(LANE_OF[m]);
        RET(keys_gemm_bwd(c, m, 0, k1, 2));
      } else {
        RET(keys_gemm_bwd(c, m, 0, k1));
      }
    }
    mark(c.st, 13 + 5 * m);
    if (!sum_in_dx) RET(await_early(m));
    sdumc_gg_problem fq;          // grouped mode: this modality's frame_dim_reshape dW, the streams as K segments
    memset(&fq, 0, sizeof(fq));
    // grouped or per-layer is decided ONCE per modality: the grouped problem overwrites its output (accumulate = 0) after the
    // lanes have joined, so a modality whose streams took different paths would lose the per-layer stream's contribution
    bool frame_grouped = ggf && (c.h() || (din[m] & 3) == 0);
    for (int s = 0; s < (m == 1 ? S : 1); ++s) {
      const float* in_s = m == 0 ? c.io.audio : (m == 2 ? c.io.video : c.io.text[s]);
      if (reinterpret_cast<uintptr_t>(in_s) & 15) frame_grouped = false;
    }
    for (int s = 0; s < (m == 1 ? S : 1); ++s) {
      const int T = pl.T[m][s];
      sdumc_dropsum ds;
      memset(&ds, 0, sizeof(ds));
      ds.samples = B;
      ds.T = T;
      ds.dx = c.p(pl.dx[m][s]);
      int nt = 0;
      for (int k = 0; k < 2; ++k)
        for (int ss = 0; ss < S; ++ss) {
          if (m == 1 && ss != s) continue;
          int64_t roff = 0;  // virtual-row offset of stream ss inside modality m
          for (int q = 0; q < ss; ++q) roff += (int64_t)B * pl.T[m][q];
          ds.g[nt] = c.h() ? reinterpret_cast<const float*>(c.ph(pl.dxd[k][m], roff * D)) : c.p(pl.dxd[k][m]) + roff * D;
          ds.drop[nt] = in_drop(c, k, m, T, ss, roff);   // row space of this term = stream ss alone
          ds.stream_idx[nt] = 0;
          ++nt;
        }
      ds.terms = nt;
      if (c.h()) ds.bf16 = 1;          // (dx is a half-length buffer: its float offset is its start either way)
      if (s == 0) mark(c.st, 14 + 5 * m);     // (after the wait for the early key-projection backward)
      if (!sum_in_dx) RET(sdumc_dropsum_bwd(&ds, c.st));
      if (s == 0) mark(c.st, 15 + 5 * m);
      const float* in = m == 0 ? c.io.audio : (m == 2 ? c.io.video : c.io.text[s]);
      const int rows = B * T;
      if (c.h() && frame_grouped) {
        fq.A[s] = reinterpret_cast<const float*>(c.ph(pl.dx[m][s]));
        fq.B[s] = in;
        fq.K[s] = rows;
        fq.b_map[s] = feat_map(c, m, s);      // (a resident store's packed bf16 rows, read in place)
        continue;
      }
      if (c.h() && c.io.row_map[0]) return SDUMC_EINVAL;      // (only the grouped launch fetches its rows through a map)
      if (c.h()) {     // dW_frame = dx^T features on bf16 storage
        sdumc_gemm_bf16 gh = GH_(SDUMC_TN, D, din[m], rows);
        gh.A[0] = c.ph(pl.dx[m][s]);
        gh.lda = D;
        gh.B[0] = in;
        gh.ldb = din[m];
        gh.C[0] = c.G + pm.frame[m].w;
        gh.ldc = din[m];
        gh.colsum_a[0] = c.G + pm.frame[m].b;
        gh.accumulate = s > 0;
        RET(run_h(c, gh));
        continue;
      }
      if (frame_grouped) {
        fq.A[s] = c.p(pl.dx[m][s]);
        fq.B[s] = in;
        fq.K[s] = rows;
        fq.b_map[s] = feat_map(c, m, s);      // (a resident store's packed rows, read in place)
        continue;
      }
      if (c.io.row_map[0]) return SDUMC_EINVAL;      // (only the grouped launch fetches its rows through a map)
      sdumc_gemm g = G_(SDUMC_TN, D, din[m], rows);
      g.A[0] = c.p(pl.dx[m][s]);
      g.lda = D;
      g.B[0] = in;
      g.ldb = din[m];
      g.C[0] = c.G + pm.frame[m].w;
      g.ldc = din[m];
      g.colsum_a[0] = c.G + pm.frame[m].b;
      g.accumulate = s > 0;
      g.bf16 = c.d.bf16 && (din[m] % 4 == 0) ? 1 : 0;
      RET(run(c, g));
    }
    if (fq.K[0] > 0) {
      fq.C = c.G + pm.frame[m].w;
      fq.colsum_a = c.G + pm.frame[m].b;
      fq.M = D;
      fq.N = din[m];
      fq.lda = D;
      fq.ldb = din[m];
      fq.ldc = din[m];
      fq.b_scale = 1.f;
      // (measured and dropped, round 5: the text slot's frame dW -- its dx is final long before audio's -- as a launch of its own on
      //  text's lane, beside the other modalities' dX launches: 1.425-1.431 against 1.390-1.395 ms; a persistent launch in the middle
      //  of the backward holds the CUs the dX chain is waiting for)
      (c.h() ? c.ggh : c.gg).push_back(fq);
    }
  }
  c.use(0);
  for (int m = 0; m < 3; ++m) mark(c.sts[LANE_OF[m]], 16 + 5 * m);
  mark(c.sts[3], 27);
  RET(join_all(c));
  // grouped mode: the three frame_dim_reshape dW (29 of the step's 127 GFLOP at C2) in one launch on the caller's stream, once
  // every modality's dx is there
  if (!c.gg.empty() || !c.ggh.empty()) RET(flush_dw_on(c, 0, 0, 1));
  mark(c.st, 41);
  RET(link(c, 3, 0));   // the dW launches of lane 3
  return SDUMC_OK;
}

// hyper != nullptr: also the Adam bias-correction update of this step (adam.hip's adam_hyper_kernel, same double
// arithmetic) -- it depends on nothing the step computes, so it rides here instead of as a launch of its own
__global__ void total_loss_kernel(float* losses, float w0, float w1, float w2, float w3, float w4, float w5, float* hyper,
                                  double beta1, double beta2) {
  losses[0] = w0 * losses[1] + w1 * losses[2] + w2 * losses[3] + w3 * losses[4] + w4 * losses[5] + w5 * losses[6];
  losses[7] = 0.f;
  if (hyper) {
    const double t = (double)hyper[1] + 1.0;
    hyper[1] = (float)t;
    hyper[2] = (float)((double)hyper[0] / (1.0 - pow(beta1, t)));
    hyper[3] = (float)sqrt(1.0 - pow(beta2, t));
  }
}

struct LossScratch {
  float* ssd;      // [3] (+1 pad)
  float* ssd_ws;   // 3 partial regions
  int64_t ssd_ws_each;
  float* labels2;  // [2B]
  float* rnc_ws;
  size_t total_floats;
};
LossScratch loss_scratch(const sdumc_net_dims& d, int B_global, float* base) {
  LossScratch s;
  const int Bg = B_global > 0 ? B_global : d.B;
  int64_t cur = 0;
  auto al = [&](int64_t n) {
    const int64_t o = cur;
    cur += (n + 63) & ~(int64_t)63;
    return o;
  };
  const int64_t o_ssd = al(4);
  s.ssd_ws_each = std::max<int64_t>((int64_t)(sdumc_ssd_workspace_bytes((int64_t)d.B * NQ * H) / sizeof(float)) + 64,
                                   (int64_t)(sdumc_distill_workspace_bytes(d.B) / sizeof(float)));
  const int64_t o_ws = al(3 * s.ssd_ws_each);
  const int64_t o_l2 = al(2LL * d.B);
  const int64_t o_rnc = al((int64_t)(sdumc_rnc_workspace_bytes(2 * Bg) / sizeof(float)));
  s.total_floats = (size_t)cur;
  s.ssd = base ? base + o_ssd : nullptr;
  s.ssd_ws = base ? base + o_ws : nullptr;
  s.labels2 = base ? base + o_l2 : nullptr;
  s.rnc_ws = base ? base + o_rnc : nullptr;
  return s;
}

int loss_ssd(const sdumc_net_dims& d, const sdumc_net_io& io, float* ssd_out, const LossScratch& ls, hipStream_t st) {
  const int B = d.B;
  if (!io.text_hidden || !io.cross_text || !io.fused) return SDUMC_EINVAL;
  RET(sdumc_ssd(io.text_hidden + (int64_t)B * D, io.text_hidden, (int64_t)B * D, ssd_out + 0, ls.ssd_ws, st));
  RET(sdumc_ssd(io.cross_text + (int64_t)B * NQ * H, io.cross_text, (int64_t)B * NQ * H, ssd_out + 1,
                ls.ssd_ws + ls.ssd_ws_each, st));
  RET(sdumc_ssd(io.fused + (int64_t)B * H, io.fused, (int64_t)B * H, ssd_out + 2, ls.ssd_ws + 2 * ls.ssd_ws_each, st));
  return SDUMC_OK;
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" int sdumc_ctx_create(void** ctx) {
  if (!ctx) return SDUMC_EINVAL;
  LaneSet* S = new LaneSet();
  if (!create_lanes(*S)) {
    destroy_lanes(*S);
    delete S;
    return SDUMC_ELAUNCH;
  }
  *ctx = S;
  return SDUMC_OK;
}
extern "C" int sdumc_ctx_destroy(void* ctx) {
  if (!ctx) return SDUMC_EINVAL;
  LaneSet* S = static_cast<LaneSet*>(ctx);
  destroy_lanes(*S);
  delete S;
  return SDUMC_OK;
}

extern "C" int sdumc_ctx_set_option(void* ctx, int32_t option, int32_t value) {
  LaneSet* S = ctx ? static_cast<LaneSet*>(ctx) : default_lanes();
  if (!S) return SDUMC_ELAUNCH;
  switch (option) {
    case SDUMC_OPT_CONCURRENCY: S->opt_concurrency = value < 0 ? -1 : (value != 0); break;
    case SDUMC_OPT_BACKGROUND_LANE:
      if (value > 3) return SDUMC_EINVAL;
      S->opt_background = value < 0 ? -1 : value;
      break;
    case SDUMC_OPT_CHAIN_CLUSTER: S->opt_chain_cluster = value < 0 ? -1 : (value != 0); break;
    case SDUMC_OPT_SPLIT:
      if (value > SDUMC_SPLIT_ALL) return SDUMC_EINVAL;
      S->opt_split = value < 0 ? -1 : value;
      break;
    default: return SDUMC_EINVAL;
  }
  return SDUMC_OK;
}

extern "C" int sdumc_set_concurrency(int on) {
  g_concurrency = on != 0;
  return SDUMC_OK;
}
extern "C" int sdumc_set_background_lane(int on) {
  g_background = on;
  return SDUMC_OK;
}

extern "C" int64_t sdumc_param_count(int32_t da, int32_t dt, int32_t dv) { return build_params(da, dt, dv).total; }
extern "C" int64_t sdumc_param_live_count(int32_t da, int32_t dt, int32_t dv) { return build_params(da, dt, dv).live; }
extern "C" int32_t sdumc_param_table(int32_t da, int32_t dt, int32_t dv, char* buf, size_t buflen) {
  const ParamMap& pm = build_params(da, dt, dv);
  std::string s;
  for (const PEntry& e : pm.table) {
    char line[256];
    snprintf(line, sizeof(line), "%s %lld %d %d %d\n", e.name.c_str(), (long long)e.off, e.rows, e.cols, e.live ? 1 : 0);
    s += line;
  }
  if (!buf || s.size() + 1 > buflen) return -(int32_t)(s.size() + 1);
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int32_t)s.size();
}

extern "C" size_t sdumc_net_workspace_bytes(const sdumc_net_dims* d) {
  ensure_side_streams();   // never called under stream capture: the place to create the internal lanes
  preload_code_objects();
  (void)sdumc_gemm_rows_prepare_();
  Plan p;
  if (!d || !make_plan(*d, p)) return 0;
  return (size_t)p.cur * sizeof(float);
}

namespace {
// join_bits: order the middle's fill of the other keep-bits set (lane 3, issued behind the last lane-3 -> caller link of forward())
// before `stream` on return.  sdumc_train_step leaves that to its backward's final join; a forward on its own must not return with
// lane 3 still reading rng_state / writing bits_next behind the caller's back (and, under capture, with an unjoined branch).
int net_forward_impl(const sdumc_net_dims* d, const sdumc_net_io* io, void* stream, bool join_bits) {
  RET(check_io(d, io));
  const SplitScope split_scope(io);
  Ctx c{*d, *io, as_stream(stream), build_params(d->da, d->dt, d->dv), Plan(), nullptr, nullptr, nullptr};
  if (!make_plan(*d, c.pl)) return SDUMC_EINVAL;
  if (io->workspace_bytes < (size_t)c.pl.cur * sizeof(float)) return SDUMC_ENOMEM;
  c.W = static_cast<float*>(io->workspace);
  c.P = io->params;
  c.init_lanes();
  RET(forward(c));
  if (join_bits && (bits_pregen(c) || io->prefetch)) RET(link(c, 3, 0));
  return SDUMC_OK;
}
}  // namespace
extern "C" int sdumc_net_forward(const sdumc_net_dims* d, const sdumc_net_io* io, void* stream) {
  return net_forward_impl(d, io, stream, true);
}

extern "C" int sdumc_net_backward(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_net_grads* g,
                                  void* stream) {
  RET(check_io(d, io));
  const SplitScope split_scope(io);
  if (!g || !g->grads || (reinterpret_cast<uintptr_t>(g->grads) & 15)) return SDUMC_EINVAL;
  Ctx c{*d, *io, as_stream(stream), build_params(d->da, d->dt, d->dv), Plan(), nullptr, nullptr, nullptr};
  if (!make_plan(*d, c.pl)) return SDUMC_EINVAL;
  if (io->workspace_bytes < (size_t)c.pl.cur * sizeof(float)) return SDUMC_ENOMEM;
  c.W = static_cast<float*>(io->workspace);
  c.P = io->params;
  c.G = g->grads;
  c.init_lanes();
  return backward(c, *g);
}

extern "C" int sdumc_net_backward_phase(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_net_grads* g,
                                        int32_t phase, void* stream) {
  RET(check_io(d, io));
  const SplitScope split_scope(io);
  if (!g || !g->grads || (reinterpret_cast<uintptr_t>(g->grads) & 15) || phase < 0 || phase > 1) return SDUMC_EINVAL;
  Ctx c{*d, *io, as_stream(stream), build_params(d->da, d->dt, d->dv), Plan(), nullptr, nullptr, nullptr};
  if (!make_plan(*d, c.pl)) return SDUMC_EINVAL;
  if (io->workspace_bytes < (size_t)c.pl.cur * sizeof(float)) return SDUMC_ENOMEM;
  c.W = static_cast<float*>(io->workspace);
  c.P = io->params;
  c.G = g->grads;
  c.init_lanes();
  return backward(c, *g, 1 << phase);
}
extern "C" int64_t sdumc_param_early_count(int32_t da, int32_t dt, int32_t dv) { return build_params(da, dt, dv).early; }

extern "C" size_t sdumc_loss_workspace_bytes(const sdumc_net_dims* d, int32_t B_global) {
  if (!d || d->B <= 0) return 0;
  return loss_scratch(*d, B_global, nullptr).total_floats * sizeof(float);
}

extern "C" int sdumc_loss_ssd(const sdumc_net_dims* d, const sdumc_net_io* io, float* ssd_out, void* scratch,
                              size_t scratch_bytes, void* stream) {
  if (!d || !io || !ssd_out || !scratch || d->streams != 2) return SDUMC_EINVAL;
  const LossScratch ls = loss_scratch(*d, 0, static_cast<float*>(scratch));
  if (scratch_bytes < ls.total_floats * sizeof(float)) return SDUMC_ENOMEM;
  return loss_ssd(*d, *io, ssd_out, ls, as_stream(stream));
}

namespace {
int loss_backward_impl(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_step_cfg* cfg, const sdumc_net_grads* g,
                       void* scratch, size_t scratch_bytes, void* stream, float* hyper, bool* total_pending = nullptr);
}
extern "C" int sdumc_loss_backward(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_step_cfg* cfg,
                                   const sdumc_net_grads* g, void* scratch, size_t scratch_bytes, void* stream) {
  return loss_backward_impl(d, io, cfg, g, scratch, scratch_bytes, stream, nullptr);
}
namespace {
// total_pending (the fused step): when the merged loss passes ran, the Adam bias correction has been made by their second pass
// and losses[0] is left to the Adam launch (sdumc_total_loss) -- no one-thread total_loss launch on the critical path
int loss_backward_impl(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_step_cfg* cfg, const sdumc_net_grads* g,
                       void* scratch, size_t scratch_bytes, void* stream, float* hyper, bool* total_pending) {
  if (total_pending) *total_pending = false;
  if (!d || !io || !cfg || !g || !scratch || d->streams != 2) return SDUMC_EINVAL;
  if (!io->vals || !io->fused || !io->rnc || !io->text_hidden || !io->cross_text) return SDUMC_EINVAL;
  if (!g->d_vals || !g->d_fused || !g->d_rnc || !g->d_text_hidden || !g->d_cross_text) return SDUMC_EINVAL;
  if (!cfg->labels || !cfg->losses) return SDUMC_EINVAL;
  hipStream_t st = as_stream(stream);
  const int B = d->B, Bg = cfg->B_global > 0 ? cfg->B_global : B;
  const LossScratch ls = loss_scratch(*d, cfg->B_global, static_cast<float*>(scratch));
  if (scratch_bytes < ls.total_floats * sizeof(float)) return SDUMC_ENOMEM;
  float* dv = const_cast<float*>(g->d_vals);
  float* df = const_cast<float*>(g->d_fused);
  float* dr = const_cast<float*>(g->d_rnc);
  float* dth = const_cast<float*>(g->d_text_hidden);
  float* dct = const_cast<float*>(g->d_cross_text);
  float* L = cfg->losses;
  const float* w = cfg->weights;
  // single GPU (no global sums / features handed in): the distillation terms and RnC share their two passes (loss.hip)
  bool done = false;
  if (!cfg->rnc_feats_global && !cfg->ssd_global && Bg == B) {
    const int rc = sdumc_losses_fused_(B, io->vals, cfg->labels, io->text_hidden, io->cross_text, io->fused, io->rnc, RD,
                                       cfg->temperature, w, dv, dth, dct, df, dr, L, ls.ssd_ws, ls.rnc_ws,
                                       total_pending ? hyper : nullptr, (double)cfg->beta1, (double)cfg->beta2, stream);
    if (rc < 0) return rc;
    done = rc == 0;
    if (done && total_pending && hyper) {
      *total_pending = true;
      return SDUMC_OK;
    }
  }
  if (!done) {
  // MSELoss x2 (main :137-138) + RMSELoss x3 (main :148; teacher side detached for text_feat / text_query_feat,
  // not for features): value and gradients in two launches
  RET(sdumc_distill_fwd_bwd(B, (float)Bg, io->vals, cfg->labels, io->text_hidden, io->cross_text, io->fused, w,
                            cfg->ssd_global, dv, dth, dct, df, L, ls.ssd_ws, stream));
  // RnCLoss over cat(r_stream0, r_stream1) with labels repeated (main :134,:140; loss.py:282-283)
  if (cfg->rnc_feats_global) {
    if (!cfg->rnc_labels_global) return SDUMC_EINVAL;
    RET(sdumc_rnc_fwd_bwd(cfg->rnc_feats_global, cfg->rnc_labels_global, 2 * Bg, RD, cfg->temperature, w[5],
                          cfg->rnc_row0[0], B, L + 6, dr, ls.rnc_ws, st));
    RET(sdumc_rnc_dfeat_rows(cfg->rnc_feats_global, 2 * Bg, RD, cfg->temperature, w[5], cfg->rnc_row0[1], B,
                             dr + (int64_t)B * RD, ls.rnc_ws, st));
  } else {
    RET(sdumc_rnc_fwd_bwd_rep(io->rnc, cfg->labels, 2 * B, RD, cfg->temperature, w[5], 0, 2 * B, L + 6, dr, ls.rnc_ws, st));
  }
  }   // !done
  hipLaunchKernelGGL(total_loss_kernel, dim3(1), dim3(1), 0, st, L, w[0], w[1], w[2], w[3], w[4], w[5], hyper,
                     (double)cfg->beta1, (double)cfg->beta2);
  SDUMC_CHECK_LAUNCH();
  return SDUMC_OK;
}
}  // namespace

namespace {
struct StepLayout {
  size_t net, grads, dout, loss, total;  // byte offsets / sizes
  size_t live;
};
StepLayout step_layout(const sdumc_net_dims& d) {
  // [gradient bucket | output gradients | loss scratch | network workspace]: the bucket comes FIRST so that it sits at the
  // same address for every batch shape a run goes through inside one capacity-sized workspace (engine.FusedTrainer) --
  // its alignment padding is zeroed once and stays zero, and a data-parallel all-reduce always finds it in one place.
  StepLayout s;
  auto up = [](size_t n) { return (n + 255) & ~(size_t)255; };
  s.live = (size_t)build_params(d.da, d.dt, d.dv).live;
  const size_t V = (size_t)d.B * 2;
  size_t cur = 0;
  s.grads = cur;
  cur += up(s.live * sizeof(float));
  s.dout = cur;
  cur += up(V * (64 + H + RD + D + NQ * H) * sizeof(float));
  s.loss = cur;
  cur += up(sdumc_loss_workspace_bytes(&d, 0));
  s.net = cur;
  cur += up(sdumc_net_workspace_bytes(&d));
  s.total = cur;
  return s;
}
}  // namespace

extern "C" size_t sdumc_net_bits_next_bytes(const sdumc_net_dims* d) {
  Plan p;
  if (!d || !d->train || d->bf16 == 2 || !make_plan(*d, p) || p.hf) return 0;
  return 2 * bits_set_bytes(*d);      // (two sets; == 2 * bits_next_off(p, 2, 0))
}

extern "C" size_t sdumc_step_workspace_bytes(const sdumc_net_dims* d) {
  if (!d || d->streams != 2) return 0;
  Plan p;
  if (!make_plan(*d, p)) return 0;
  return step_layout(*d).total;
}

extern "C" int sdumc_train_step(const sdumc_net_dims* d, const sdumc_net_io* io, const sdumc_step_cfg* cfg,
                                void* stream) {
  RET(check_io(d, io));
  if (!cfg || d->streams != 2 || !cfg->adam_m || !cfg->adam_v || !cfg->hyper) return SDUMC_EINVAL;
  const StepLayout sl = step_layout(*d);
  if (io->workspace_bytes < sl.total) return SDUMC_ENOMEM;
  char* base = static_cast<char*>(io->workspace);
  sdumc_net_io nio = *io;
  nio.workspace = base + sl.net;
  nio.workspace_bytes = sl.total - sl.net;
  mark(static_cast<hipStream_t>(stream), 0);
  RET(net_forward_impl(d, &nio, stream, false));
  const size_t V = (size_t)d->B * 2;
  float* dout = reinterpret_cast<float*>(base + sl.dout);
  sdumc_net_grads g;
  g.d_vals = dout;
  g.d_fused = dout + 64 * V;
  g.d_rnc = g.d_fused + H * V;
  g.d_text_hidden = g.d_rnc + RD * V;
  g.d_cross_text = g.d_text_hidden + D * V;
  g.grads = reinterpret_cast<float*>(base + sl.grads);
  // the Adam bias-correction update rides in the loss's last launch, the dropout call counter in the Adam launch
  bool total_pending = false;
  RET(loss_backward_impl(d, &nio, cfg, &g, base + sl.loss, sl.net - sl.loss, stream, cfg->hyper, &total_pending));
  mark(static_cast<hipStream_t>(stream), 5);
  RET(sdumc_net_backward(d, &nio, &g, stream));
  mark(static_cast<hipStream_t>(stream), 9);
  sdumc_total_loss tl;
  tl.losses = cfg->losses;
  tl.chain_err = sdumc_chain_cluster_err_ptr_();
  for (int i = 0; i < 6; ++i) tl.w[i] = cfg->weights[i];
  RET(sdumc_adam_apply_(io->params, g.grads, cfg->adam_m, cfg->adam_v, (int64_t)sl.live, cfg->hyper, cfg->beta1, cfg->beta2,
                        cfg->eps, cfg->weight_decay, 1.0f, d->train ? const_cast<uint32_t*>(io->rng_state) : nullptr, 2u,
                        total_pending ? &tl : nullptr, stream));
  mark(static_cast<hipStream_t>(stream), 10);
  return SDUMC_OK;      // (the backward's last act joined lane 3: the next batch's assembly, too, is ordered before whatever follows)
}

extern "C" int sdumc_debug_marks(int on) {
  g_marks_on = on != 0;
  return SDUMC_OK;
}
// ms since mark 0 of every mark recorded by the last step (-1 where none); the caller synchronises first
extern "C" int sdumc_debug_marks_read(float* ms, int n) {
  for (int i = 0; i < n; ++i) {
    ms[i] = -1.f;
    if (i < kMarks && g_mark_set[i] && g_mark_set[0]) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, g_marks[0], g_marks[i]) == hipSuccess) ms[i] = t;
    }
  }
  return SDUMC_OK;
}

// debug / probes (tools/fwd_determinism_probe.py): the workspace plan of sdumc_net_forward as text lines
// "name offset_in_floats length_in_floats" (offsets relative to sdumc_net_io.workspace; bf16-storage tensors: their length in floats)
extern "C" int32_t sdumc_debug_plan_table(const sdumc_net_dims* d, char* buf, size_t buflen) {
  Plan p;
  if (!d || !make_plan(*d, p)) return 0;
  const int HS = p.hf ? 2 : 1;
  const int64_t V = p.V;
  std::string s;
  auto put = [&](const std::string& name, int64_t off, int64_t n) {
    char line[160];
    snprintf(line, sizeof(line), "%s %lld %lld\n", name.c_str(), (long long)off, (long long)n);
    s += line;
  };
  const char* mod[3] = {"a", "t", "v"};
  for (int m = 0; m < 3; ++m) {
    put(std::string("x_") + mod[m], p.x[m][0], p.rows[m] / (m == 1 ? 1 : p.S) * D / HS);
    for (int k = 0; k < 2; ++k) {
      const std::string sfx = std::to_string(k) + mod[m];
      put("keys" + sfx, p.keys[k][m], p.rows[m] * D / HS);
      put("attn" + sfx, p.attn[k][m], p.rows[m] * (k == 0 ? 1 : NQ));
      put("pooled" + sfx, p.pooled[k][m], V * (k == 0 ? 1 : NQ) * D);
    }
  }
  put("hpre", p.hpre, 3 * V * D); put("u1", p.u1, 3 * V * D); put("u", p.u, 3 * V * D); put("att1", p.att1, V * D);
  put("att2", p.att2, V * D); put("alpha", p.alpha, V * 3); put("qin", p.qin, 7 * V * D); put("q", p.q, 7 * V * D);
  put("qp", p.qp, 3 * V * NQ * D); put("ca_out", p.ca_out, 3 * V * NQ * D); put("c1", p.c1, 3 * V * NQ * D);
  put("c", p.c, 3 * V * NQ * H); put("h", p.h, V * NQ * H); put("e1", p.e1, V * D); put("e2", p.e2, V * H);
  put("beta", p.beta, V * NQ); put("z", p.z, V * H); put("vals", p.vals, V); put("r1", p.r1, V * RD); put("r", p.r, V * RD);
  put("wt", p.wt, build_params(d->da, d->dt, d->dv).early);
  for (int m = 0; m < 3; ++m) {      // backward (train-mode steps): per site dz and dout * mask, per modality dx
    for (int k = 0; k < 2; ++k) {
      const std::string sfx = std::to_string(k) + mod[m];
      put("dz" + sfx, p.dz[k][m], p.rows[m] * D / HS);
      if (p.dom[k][m]) put("dom" + sfx, p.dom[k][m], V * (k == 0 ? 1 : NQ) * D);
    }
    put(std::string("dx_") + mod[m], p.dx[m][0], p.rows[m] / (m == 1 ? 1 : p.S) * D / HS);
  }
  put("d_hpre", p.d_hpre, 3 * V * D); put("d_ca_out", p.d_ca_out, 3 * V * NQ * D);
  if (!buf || s.size() + 1 > buflen) return -(int32_t)(s.size() + 1);
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int32_t)s.size();
}

// gradient bucket of the most recent sdumc_train_step inside its workspace (for tests / DP all-reduce)
extern "C" size_t sdumc_step_grads_offset(const sdumc_net_dims* d) {
  if (!d || d->streams != 2) return 0;
  return step_layout(*d).grads;
}
