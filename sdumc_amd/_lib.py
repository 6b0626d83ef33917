"""ctypes binding of libsdumc_hip.so (include/sdumc_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent the
import of any product module raises.  PyTorch-ROCm tensors provide device
memory and streams; every pointer crossing this boundary is `tensor.data_ptr()`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDUMC_LIB") or os.path.join(_HERE, "csrc", "libsdumc_hip.so")   # override: A/B of builds

MAX_GROUPS = 8
D, H, NQ, RNC_DIM, N_SITES = 256, 128, 7, 64, 35
NT, NN, TN = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2

c_float_p = C.c_void_p  # device pointers travel as integers


class Dropout(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("site", C.c_uint32), ("threshold", C.c_uint32), ("scale", C.c_float),
                ("rows", C.c_uint32), ("width", C.c_uint32), ("samples", C.c_uint32), ("sample0", C.c_uint32),
                ("call0", C.c_uint32), ("stream0", C.c_uint32), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
                ("dev_state", C.c_void_p), ("bits", C.c_void_p)]


class Gemm(C.Structure):
    _fields_ = [("layout", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("groups", C.c_int32),
                ("A", C.c_void_p * MAX_GROUPS), ("B", C.c_void_p * MAX_GROUPS),
                ("C", C.c_void_p * MAX_GROUPS), ("bias", C.c_void_p * MAX_GROUPS),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("a_row_mod", C.c_int32), ("b_row_mod", C.c_int32),
                ("a_drop", Dropout), ("b_drop", Dropout),
                ("act", C.c_int32), ("c_drop", Dropout), ("c_drop_group_stride", C.c_int32),
                ("accumulate", C.c_int32), ("splitk", C.c_int32), ("tile", C.c_int32),
                ("ab_drop_group_stride", C.c_int32), ("ab_drop_bits", C.c_void_p * MAX_GROUPS),
                ("c_mask_y", C.c_void_p * MAX_GROUPS), ("c_mask_scale", C.c_float),
                ("colsum_a", C.c_void_p * MAX_GROUPS),
                ("bf16", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("batch", C.c_int32),
                ("stride_a", C.c_int64), ("stride_b", C.c_int64), ("stride_c", C.c_int64)]


class GemmBf16(C.Structure):
    _fields_ = [("layout", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("groups", C.c_int32),
                ("A", C.c_void_p * MAX_GROUPS), ("B", C.c_void_p * MAX_GROUPS), ("C", C.c_void_p * MAX_GROUPS),
                ("bias", C.c_void_p * MAX_GROUPS), ("colsum_a", C.c_void_p * MAX_GROUPS),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("a_row_mod", C.c_int32), ("b_row_mod", C.c_int32),
                ("act", C.c_int32), ("accumulate", C.c_int32), ("c_bf16", C.c_int32), ("splitk", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class GGProblem(C.Structure):
    _fields_ = [("A", C.c_void_p * 2), ("B", C.c_void_p * 2), ("b_bits", C.c_void_p * 2),
                ("K", C.c_int32 * 2), ("b_row_mod", C.c_int32 * 2),
                ("C", C.c_void_p), ("colsum_a", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("bits_qw", C.c_int32), ("b_scale", C.c_float), ("accumulate", C.c_int32), ("b_map", C.c_void_p * 2)]


class GemmP3(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("A", C.c_void_p), ("B", C.c_void_p), ("lda", C.c_int64), ("ldb", C.c_int64),
                ("a_row_mod", C.c_int32), ("A2", C.c_void_p), ("a2_row0", C.c_int32),
                ("a_bits", C.c_void_p), ("bits_qw", C.c_int32), ("a_scale", C.c_float),
                ("bias", C.c_void_p), ("act", C.c_int32),
                ("C", C.c_void_p), ("ldc", C.c_int32), ("C_p3", C.c_void_p), ("ldc_p3", C.c_int64),
                ("splitk", C.c_int32), ("tile_m", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("a_map", C.c_void_p), ("a2_map", C.c_void_p), ("a_map_rows", C.c_int64)]


class GemmB1(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("A", C.c_void_p), ("B", C.c_void_p), ("lda", C.c_int64), ("ldb", C.c_int64),
                ("a_row_mod", C.c_int32), ("A2", C.c_void_p), ("a2_row0", C.c_int32),
                ("bias", C.c_void_p), ("act", C.c_int32),
                ("C", C.c_void_p), ("ldc", C.c_int32), ("c_bf16", C.c_int32), ("splitk", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("a_map", C.c_void_p), ("a2_map", C.c_void_p), ("a_map_rows", C.c_int64)]


class RowsProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("a_bits", C.c_void_p), ("bias", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("a_row_mod", C.c_int32),
                ("a_scale", C.c_float), ("accumulate", C.c_int32), ("act", C.c_int32),
                ("pool_w", C.c_void_p), ("pool_g", C.c_void_p), ("pool_nq", C.c_int32), ("pool_T", C.c_int32),
                ("c_bits", C.c_void_p), ("c_scale", C.c_float), ("fold", C.c_int32)]


class AttnPool(C.Structure):
    _fields_ = [("V", C.c_int32), ("T", C.c_int32), ("nq", C.c_int32), ("x_samples", C.c_int32),
                ("x", C.c_void_p), ("keys", C.c_void_p), ("q", C.c_void_p), ("q_stride", C.c_int64),
                ("scale", C.c_float), ("x_drop", Dropout), ("out_drop", Dropout),
                ("attn", C.c_void_p), ("pooled", C.c_void_p), ("out", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("dim", C.c_int32),
                ("lengths", C.c_void_p), ("bf16", C.c_int32), ("tickets", C.c_void_p), ("partial_only", C.c_int32)]


class Umca(C.Structure):
    _fields_ = [("a", AttnPool), ("w_in", C.c_void_p), ("b_in", C.c_void_p), ("x_p3", C.c_void_p), ("w_in_p3f", C.c_void_p)]


class AttnPoolBwd(C.Structure):
    _fields_ = [("f", AttnPool), ("dout", C.c_void_p), ("dz", C.c_void_p), ("dxd", C.c_void_p),
                ("dq", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("dq_sum", C.c_void_p),
                ("dout_masked", C.c_void_p)]


class DropSum(C.Structure):
    _fields_ = [("terms", C.c_int32), ("g", C.c_void_p * 8), ("drop", Dropout * 8),
                ("stream_idx", C.c_int32 * 8), ("samples", C.c_int32), ("T", C.c_int32), ("dx", C.c_void_p),
                ("bf16", C.c_int32)]


class NetDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("streams", C.c_int32),
                ("Ta", C.c_int32), ("Tv", C.c_int32), ("Tt", C.c_int32 * 2),
                ("da", C.c_int32), ("dt", C.c_int32), ("dv", C.c_int32),
                ("train", C.c_int32), ("sample0", C.c_int32),
                ("p_frame", C.c_double), ("p_mlp", C.c_double), ("bf16", C.c_int32)]


class NetIO(C.Structure):
    _fields_ = [("audio", C.c_void_p), ("video", C.c_void_p), ("text", C.c_void_p * 2),
                ("params", C.c_void_p), ("rng_state", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                # outputs, each [streams*B, ...]
                ("vals", C.c_void_p), ("fused", C.c_void_p), ("rnc", C.c_void_p),
                ("text_hidden", C.c_void_p), ("cross_text", C.c_void_p),
                ("lengths", C.c_void_p * 4),
                ("audio_p3", C.c_void_p), ("video_p3", C.c_void_p), ("text_p3", C.c_void_p * 2),   # optional bf16-plane copies of the features
                ("row_map", C.c_void_p * 4),      # optional: the features are a resident store's packed tensors, read in place through these maps
                ("store_rows", C.c_int64 * 4),    # ... and those tensors' row counts (0 = unknown)
                ("bits_next", C.c_void_p), ("bits_phase", C.c_int32),   # optional: two sets of keep-bits, the next call's generated in this call's middle
                ("bits_next_bytes", C.c_size_t), ("bits_next_dims", C.c_void_p),   # its capacity (0 = this call's dims) / dims of the NEXT call (NULL = the same)
                ("prefetch", C.c_void_p), ("prefetch_workgroups", C.c_int32),      # optional: the NEXT batch's gather descriptor, issued in this call's middle
                ("ctx", C.c_void_p)]          # optional caller-owned execution context (sdumc_ctx_create); None = device default


class NetGrads(C.Structure):
    _fields_ = [("d_vals", C.c_void_p), ("d_fused", C.c_void_p), ("d_rnc", C.c_void_p),
                ("d_text_hidden", C.c_void_p), ("d_cross_text", C.c_void_p),
                ("grads", C.c_void_p)]


class StepCfg(C.Structure):
    _fields_ = [("weights", C.c_float * 6), ("temperature", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("labels", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("hyper", C.c_void_p),
                ("losses", C.c_void_p),
                ("B_global", C.c_int32), ("ssd_global", C.c_void_p), ("rnc_feats_global", C.c_void_p),
                ("rnc_labels_global", C.c_void_p), ("rnc_row0", C.c_int32 * 2)]


GATHER_MAX_SEGS = 8


class GatherSeg(C.Structure):
    _fields_ = [("packed", C.c_void_p), ("start_all", C.c_void_p), ("len_all", C.c_void_p), ("out", C.c_void_p),
                ("len_out", C.c_void_p), ("map_out", C.c_void_p), ("zero_row", C.c_int32), ("Tmax", C.c_int32), ("d4", C.c_int32),
                ("unit0", C.c_int64)]


class GatherBatch(C.Structure):
    _fields_ = [("seg", GatherSeg * GATHER_MAX_SEGS), ("nseg", C.c_int32), ("B", C.c_int32), ("idx", C.c_void_p),
                ("labels_all", C.c_void_p), ("labels_out", C.c_void_p), ("total", C.c_int64)]


class Softmax(C.Structure):
    _fields_ = [("batch", C.c_int32), ("heads", C.c_int32), ("tq", C.c_int32), ("tk", C.c_int32),
                ("scale", C.c_float), ("mask", C.c_void_p), ("scores", C.c_void_p), ("probs_drop", C.c_void_p),
                ("weights", C.c_void_p), ("drop", Dropout)]


class DropAdd(C.Structure):
    _fields_ = [("x", C.c_void_p), ("alpha", C.c_float), ("pos_table", C.c_void_p), ("pos_src", C.c_void_p),
                ("residual", C.c_void_p), ("y", C.c_void_p),
                ("samples", C.c_int32), ("rows", C.c_int32), ("width", C.c_int32), ("drop", Dropout)]


class Mha(C.Structure):
    _fields_ = [("tq", C.c_int32), ("tk", C.c_int32), ("batch", C.c_int32), ("embed", C.c_int32), ("heads", C.c_int32),
                ("query", C.c_void_p), ("key", C.c_void_p), ("value", C.c_void_p),
                ("in_proj_weight", C.c_void_p), ("in_proj_bias", C.c_void_p),
                ("out_proj_weight", C.c_void_p), ("out_proj_bias", C.c_void_p),
                ("attn_mask", C.c_void_p), ("attn_drop", Dropout),
                ("out", C.c_void_p), ("weights", C.c_void_p),
                ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
                ("probs", C.c_void_p), ("probs_drop", C.c_void_p), ("ctx", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("bf16", C.c_int32),
                ("bias_k", C.c_void_p), ("bias_v", C.c_void_p), ("add_zero_attn", C.c_int32)]


class MhaGrads(C.Structure):
    _fields_ = [("dout", C.c_void_p), ("dquery", C.c_void_p), ("dkey", C.c_void_p), ("dvalue", C.c_void_p),
                ("d_in_proj_weight", C.c_void_p), ("d_in_proj_bias", C.c_void_p),
                ("d_out_proj_weight", C.c_void_p), ("d_out_proj_bias", C.c_void_p),
                ("d_bias_k", C.c_void_p), ("d_bias_v", C.c_void_p)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int64), ("total_ms", C.c_double), ("total_flops", C.c_double)]


class CopySeg(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("ld_src", C.c_int32), ("ld_dst", C.c_int32),
                ("rows", C.c_int32), ("cols", C.c_int32)]


_SIGS = {
    "sdumc_copy2d_multi": (C.c_int, [C.POINTER(CopySeg), C.c_int32, C.c_void_p]),
    "sdumc_profile_enable": (C.c_int, [C.c_int]),
    "sdumc_profile_report": (C.c_int, [C.POINTER(ProfEntry), C.c_int]),
    "sdumc_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(Gemm)]),
    "sdumc_gemm_f32": (C.c_int, [C.POINTER(Gemm), C.c_void_p]),
    "sdumc_gemm_group_workspace_bytes": (C.c_size_t, [C.POINTER(GGProblem), C.c_int32]),
    "sdumc_gemm_group_tn": (C.c_int, [C.POINTER(GGProblem), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdumc_gemm_group_bf16_workspace_bytes": (C.c_size_t, [C.POINTER(GGProblem), C.c_int32]),
    "sdumc_gemm_group_tn_bf16": (C.c_int, [C.POINTER(GGProblem), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdumc_set_split_": (None, [C.c_int]),
    "sdumc_get_split_": (C.c_int, []),
    "sdumc_gemm_p3_workspace_bytes": (C.c_size_t, [C.POINTER(GemmP3)]),
    "sdumc_gemm_p3_nt": (C.c_int, [C.POINTER(GemmP3), C.c_void_p]),
    "sdumc_p3_split": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "sdumc_p3_split_frag": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sdumc_p3_join": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "sdumc_gemm_b1_workspace_bytes": (C.c_size_t, [C.POINTER(GemmB1)]),
    "sdumc_gemm_b1_nt": (C.c_int, [C.POINTER(GemmB1), C.c_void_p]),
    "sdumc_b1_frag_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "sdumc_gemm_rows256": (C.c_int, [C.POINTER(RowsProblem), C.c_int32, C.c_void_p]),
    "sdumc_gemm_rows256_bf16": (C.c_int, [C.POINTER(RowsProblem), C.c_int32, C.c_void_p]),
    "sdumc_attnpool_fwd_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "sdumc_attnpool_fwd": (C.c_int, [C.POINTER(AttnPool), C.c_void_p]),
    "sdumc_attnpool_fwd_workspace_bytes_dim": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sdumc_attnpool_bwd_workspace_bytes_dim": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sdumc_attnpool_bwd_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "sdumc_attnpool_bwd": (C.c_int, [C.POINTER(AttnPoolBwd), C.c_void_p]),
    "sdumc_attnpool_fwd_multi": (C.c_int, [C.POINTER(AttnPool), C.c_int32, C.c_void_p]),
    "sdumc_umca_fwd": (C.c_int, [C.POINTER(Umca), C.c_void_p]),
    "sdumc_attnpool_bwd_multi": (C.c_int, [C.POINTER(AttnPoolBwd), C.c_int32, C.c_void_p]),
    "sdumc_relu_drop_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "sdumc_colsum_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "sdumc_colsum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "sdumc_add_n": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "sdumc_dropsum_bwd": (C.c_int, [C.POINTER(DropSum), C.c_void_p]),
    "sdumc_fusion_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "sdumc_fusion_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "sdumc_hweight_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "sdumc_hweight_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p]),
    "sdumc_zpool_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "sdumc_zpool_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "sdumc_mse_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdumc_ssd_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "sdumc_ssd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdumc_rmse_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_float, C.c_void_p,
                                 C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "sdumc_rnc_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "sdumc_rnc_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdumc_rnc_fwd_bwd_rep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdumc_distill_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "sdumc_distill_fwd_bwd": (C.c_int, [C.c_int32, C.c_float] + [C.c_void_p] * 14),
    "sdumc_rnc_dfeat_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdumc_rnc_mask": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "sdumc_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "sdumc_copy2d": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sdumc_axpy2d": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sdumc_gather_pad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sdumc_mask_apply_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "sdumc_gemm_bf16_workspace_bytes": (C.c_size_t, [C.POINTER(GemmBf16)]),
    "sdumc_gemm_bf16_run": (C.c_int, [C.POINTER(GemmBf16), C.c_void_p]),
    "sdumc_ctx_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "sdumc_ctx_destroy": (C.c_int, [C.c_void_p]),
    "sdumc_ctx_set_option": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "sdumc_gather_pad_idx": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "sdumc_gather_batch": (C.c_int, [C.POINTER(GatherBatch), C.c_int32, C.c_void_p]),
    "sdumc_fill": (C.c_int, [C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    "sdumc_rng_advance": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "sdumc_dropout_bits": (C.c_int, [C.POINTER(Dropout), C.c_int32, C.c_void_p, C.c_void_p]),
    "sdumc_dropout_bits_multi": (C.c_int, [C.POINTER(Dropout), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_void_p]),
    "sdumc_dropout_bits_apply_bf16": (C.c_int, [C.POINTER(Dropout), C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_void_p, C.c_int64,
                                               C.POINTER(C.c_void_p), C.c_void_p]),
    "sdumc_dropout_mask": (C.c_int, [C.POINTER(Dropout), C.c_int32, C.c_void_p, C.c_void_p]),
    "sdumc_version": (C.c_char_p, []),
    # generic MHA / Transformer-encoder pieces (transformer.hip)
    "sdumc_layernorm_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "sdumc_layernorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "sdumc_layernorm_bwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int64, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdumc_softmax_fwd": (C.c_int, [C.POINTER(Softmax), C.c_void_p]),
    "sdumc_softmax_bwd": (C.c_int, [C.POINTER(Softmax), C.c_void_p, C.c_void_p]),
    "sdumc_drop_add": (C.c_int, [C.POINTER(DropAdd), C.c_void_p]),
    "sdumc_mha_workspace_bytes": (C.c_size_t, [C.POINTER(Mha), C.c_int32]),
    "sdumc_mha_forward": (C.c_int, [C.POINTER(Mha), C.c_void_p]),
    "sdumc_mha_backward": (C.c_int, [C.POINTER(Mha), C.POINTER(MhaGrads), C.c_void_p]),
    # network level (engine.hip)
    "sdumc_param_count": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "sdumc_param_live_count": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "sdumc_param_table": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]),
    "sdumc_set_concurrency": (C.c_int, [C.c_int]),
    "sdumc_set_background_lane": (C.c_int, [C.c_int]),
    "sdumc_set_chain_cluster": (C.c_int, [C.c_int]),
    "sdumc_chain_cluster_error_": (C.c_int, []),
    "sdumc_chain_cluster_reset_error": (C.c_int, []),
    "sdumc_chain_cluster_test_hold_": (C.c_int, [C.c_int]),
    "sdumc_chain_cluster_debug_read_": (C.c_int, [C.c_void_p, C.c_int]),
    "sdumc_chain_cluster_error_flag": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sdumc_chain_cluster_error_merge": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sdumc_debug_marks": (C.c_int, [C.c_int]),
    "sdumc_debug_marks_read": (C.c_int, [C.POINTER(C.c_float), C.c_int]),
    "sdumc_attnpool_set_v2_": (C.c_int, [C.c_int]),
    "sdumc_debug_plan_table": (C.c_int32, [C.POINTER(NetDims), C.c_char_p, C.c_size_t]),
    "sdumc_net_workspace_bytes": (C.c_size_t, [C.POINTER(NetDims)]),
    "sdumc_net_bits_next_bytes": (C.c_size_t, [C.POINTER(NetDims)]),
    "sdumc_net_forward": (C.c_int, [C.POINTER(NetDims), C.POINTER(NetIO), C.c_void_p]),
    "sdumc_net_backward": (C.c_int, [C.POINTER(NetDims), C.POINTER(NetIO), C.POINTER(NetGrads), C.c_void_p]),
    "sdumc_net_backward_phase": (C.c_int, [C.POINTER(NetDims), C.POINTER(NetIO), C.POINTER(NetGrads), C.c_int32, C.c_void_p]),
    "sdumc_param_early_count": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "sdumc_step_workspace_bytes": (C.c_size_t, [C.POINTER(NetDims)]),
    "sdumc_train_step": (C.c_int, [C.POINTER(NetDims), C.POINTER(NetIO), C.POINTER(StepCfg), C.c_void_p]),
    "sdumc_loss_workspace_bytes": (C.c_size_t, [C.POINTER(NetDims), C.c_int32]),
    "sdumc_loss_ssd": (C.c_int, [C.POINTER(NetDims), C.POINTER(NetIO), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdumc_loss_backward": (C.c_int, [C.POINTER(NetDims), C.POINTER(NetIO), C.POINTER(StepCfg), C.POINTER(NetGrads),
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdumc_dp_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sdumc_dp_record_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "sdumc_dp_record": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdumc_dp_unpack": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "sdumc_step_grads_offset": (C.c_size_t, [C.POINTER(NetDims)]),
}

EXPORTS = tuple(_SIGS)


class SdumcError(RuntimeError):
    pass


_ERR = {-1: "SDUMC_EINVAL (bad argument)", -2: "SDUMC_ELAUNCH (HIP launch/runtime error)",
        -3: "SDUMC_ENOMEM (workspace too small)"}


def _load():
    if not os.path.exists(LIB_PATH):
        raise SdumcError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sdumc_amd/csrc`. sdumc_amd has no CPU or PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        if os.environ.get("SDUMC_LIB") and not hasattr(lib, name):
            continue             # A/B against an older build: it may lack newer entry points
        fn = getattr(lib, name)  # AttributeError (loud) if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, what=""):
    if rc != 0:
        raise SdumcError(f"{what} failed: {_ERR.get(rc, rc)}")


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def drop_threshold(p):
    return int(float(p) * 4294967296.0)


def make_dropout(enabled, site, p, rows, width, samples, sample0=0, call0=0, seed=0, dev_state=None):
    import numpy as np
    d = Dropout()
    d.enabled = 1 if enabled else 0
    d.site = site
    d.threshold = drop_threshold(p)
    d.scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    d.rows, d.width, d.samples, d.sample0, d.call0 = rows, width, samples, sample0, call0
    d.seed_lo, d.seed_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    d.dev_state = ptr(dev_state)
    return d
