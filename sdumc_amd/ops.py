"""Operator-level Python wrappers over the C ABI (one function per entry point of
include/sdumc_hip.h).  Used by the drop-in modules, the data-parallel trainer and
the parity tests.  Device tensors in, device tensors out; no fallback path."""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check, ptr, make_dropout

NT, NN, TN = _lib.NT, _lib.NN, _lib.TN
ACT_NONE, ACT_RELU, ACT_TANH = _lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_TANH


def _st():
    return _lib.current_stream()


def gemm(layout, A, B, M, N, K, bias=None, C_out=None, lda=None, ldb=None, ldc=None, act=ACT_NONE, a_row_mod=0,
         b_row_mod=0, a_drop=None, b_drop=None, c_drop=None, c_drop_group_stride=0, accumulate=False, splitk=1,
         tile=0, colsum_a=None, ab_drop_group_stride=0, ab_drop_bits=None, c_mask_y=None, c_mask_scale=1.0, bf16=False,
         batch=0, stride_a=0, stride_b=0, stride_c=0):
    """Grouped when A/B/(bias)/C_out are lists; strided-batched when batch > 1 (C_out required)."""
    As = A if isinstance(A, (list, tuple)) else [A]
    Bs = B if isinstance(B, (list, tuple)) else [B]
    groups = len(As)
    dev = As[0].device
    if C_out is None:
        Cs = [torch.empty(M, N, device=dev) for _ in range(groups)]
    else:
        Cs = C_out if isinstance(C_out, (list, tuple)) else [C_out]
    biases = bias if isinstance(bias, (list, tuple)) else [bias] * groups
    g = _lib.Gemm()
    g.layout, g.M, g.N, g.K, g.groups = layout, M, N, K, groups
    for i in range(groups):
        g.A[i], g.B[i], g.C[i], g.bias[i] = ptr(As[i]), ptr(Bs[i]), ptr(Cs[i]), ptr(biases[i])
    g.lda = lda if lda is not None else (K if layout != TN else M)
    g.ldb = ldb if ldb is not None else (K if layout == NT else N)
    g.ldc = ldc if ldc is not None else N
    g.a_row_mod, g.b_row_mod = a_row_mod, b_row_mod
    if a_drop is not None:
        g.a_drop = a_drop
    if b_drop is not None:
        g.b_drop = b_drop
    if c_drop is not None:
        g.c_drop = c_drop
    g.c_drop_group_stride = c_drop_group_stride
    g.act, g.accumulate, g.splitk, g.tile = act, 1 if accumulate else 0, splitk, tile
    g.ab_drop_group_stride = ab_drop_group_stride
    g.bf16 = 1 if bf16 else 0
    g.batch, g.stride_a, g.stride_b, g.stride_c = batch, stride_a, stride_b, stride_c
    if c_mask_y is not None:
        ys = c_mask_y if isinstance(c_mask_y, (list, tuple)) else [c_mask_y]
        for i in range(groups):
            g.c_mask_y[i] = ptr(ys[i])
        g.c_mask_scale = c_mask_scale
    if ab_drop_bits is not None:
        for i in range(groups):
            g.ab_drop_bits[i] = ptr(ab_drop_bits[i])
    if colsum_a is not None:
        cs = colsum_a if isinstance(colsum_a, (list, tuple)) else [colsum_a]
        for i in range(groups):
            g.colsum_a[i] = ptr(cs[i])
    need = lib.sdumc_gemm_workspace_bytes(C.byref(g))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    g.workspace, g.workspace_bytes = ptr(ws), need
    check(lib.sdumc_gemm_f32(C.byref(g), _st()), "sdumc_gemm_f32")
    return Cs if isinstance(A, (list, tuple)) else Cs[0]


def gemm_group_tn(problems, workspace=None):
    """All weight gradients of a phase in one persistent launch + one ordered reduce (sdumc_gemm_group_tn).

    problems: list of dicts {A, B, C, [A1, B1], [bits, bits1], [colsum], [b_row_mod, b_row_mod1], [scale], [accumulate],
    [M, N, K, K1, lda, ldb, ldc]}: C[M, N] (+)= A[K, M]^T . mask(B[K, N]) (+ the same over the second K segment A1 / B1);
    A, B are 2-D device tensors (row-contiguous), bits the uint8 keep-bits [K, N / 4] of a dropout fused on B."""
    n = len(problems)
    arr = (_lib.GGProblem * n)()
    keep = []
    for i, q in enumerate(problems):
        g = arr[i]
        A, B = q["A"], q["B"]
        g.A[0], g.B[0] = ptr(A), ptr(B)
        g.M = q.get("M", A.shape[1])
        g.N = q.get("N", B.shape[1])
        g.K[0] = q.get("K", A.shape[0])
        g.lda = q.get("lda", A.stride(0))
        g.ldb = q.get("ldb", B.stride(0))
        if q.get("A1") is not None:
            g.A[1], g.B[1] = ptr(q["A1"]), ptr(q["B1"])
            g.K[1] = q.get("K1", q["A1"].shape[0])
        g.b_row_mod[0], g.b_row_mod[1] = q.get("b_row_mod", 0), q.get("b_row_mod1", 0)
        g.b_map[0], g.b_map[1] = ptr(q.get("b_map")), ptr(q.get("b_map1"))      # int32 row maps: B / B1 are a store's packed rows
        if q.get("bits") is not None:
            g.b_bits[0] = ptr(q["bits"])
            g.bits_qw = q["bits"].stride(0)
            if q.get("bits1") is not None:
                g.b_bits[1] = ptr(q["bits1"])
        g.b_scale = q.get("scale", 1.0)
        Cm = q.get("C")
        if Cm is None:
            Cm = torch.empty(g.M, g.N, device=A.device)
            q["C"] = Cm
        g.C, g.ldc = ptr(Cm), q.get("ldc", Cm.stride(0))
        g.colsum_a = ptr(q.get("colsum"))
        g.accumulate = 1 if q.get("accumulate") else 0
        keep.append(Cm)
    hf = problems[0]["A"].dtype == torch.bfloat16      # bf16 storage: sdumc_gemm_group_tn_bf16 (C stays fp32)
    need = (lib.sdumc_gemm_group_bf16_workspace_bytes if hf else lib.sdumc_gemm_group_workspace_bytes)(arr, n)
    if workspace is None:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=problems[0]["A"].device)
    fn = lib.sdumc_gemm_group_tn_bf16 if hf else lib.sdumc_gemm_group_tn
    check(fn(arr, n, ptr(workspace), workspace.numel() * workspace.element_size(), _st()), "sdumc_gemm_group_tn")
    return [q["C"] for q in problems]


def p3_split(x, out=None):
    """fp32 [rows, cols] (cols % 8 == 0) -> its P3 tensor: three bf16 planes, chunk-interleaved ([rows][cols / 8][3][8] bf16 =
    6 bytes per element, include/sdumc_hip.h: sdumc_gemm_p3); returned as a uint8 tensor [rows, 6 * cols]."""
    rows, cols = x.shape
    if out is None:
        out = torch.empty(rows, 6 * cols, dtype=torch.uint8, device=x.device)
    check(lib.sdumc_p3_split(ptr(x), x.stride(0), ptr(out), out.stride(0), rows, cols, _st()), "sdumc_p3_split")
    return out


def p3_split_frag(w, out=None):
    """fp32 weight [rows, cols] (rows % 32 == 0, cols % 16 == 0) -> its fragment-major P3 tensor (the B operand of gemm_p3_nt):
    uint8 [rows / 32, cols / 16 * 3072]."""
    rows, cols = w.shape
    if out is None:
        out = torch.empty(rows // 32, cols // 16 * 3072, dtype=torch.uint8, device=w.device)
    check(lib.sdumc_p3_split_frag(ptr(w), w.stride(0), ptr(out), rows, cols, _st()), "sdumc_p3_split_frag")
    return out


def p3_join(p3, cols):
    """P3 tensor -> fp32 [rows, cols], bit-exact inverse of p3_split"""
    rows = p3.shape[0]
    out = torch.empty(rows, cols, device=p3.device)
    check(lib.sdumc_p3_join(ptr(p3), p3.stride(0), ptr(out), out.stride(0), rows, cols, _st()), "sdumc_p3_join")
    return out


def gemm_p3_nt_call(A3, B3, M, N, K, bias=None, act=ACT_NONE, a_row_mod=0, bits=None, scale=1.0, C_out=None, want_f32=True,
                    want_p3=False, splitk=0, tile_m=0, A3_second=None, second_row0=0, a_map=None, a2_map=None, a_map_rows=0):
    """The prepared call of gemm_p3_nt: returns (launch, results) -- launch() enqueues sdumc_gemm_p3_nt on the current stream
    (a few microseconds of host time: benches), results = the output tensor(s)."""
    dev = A3.device
    g = _lib.GemmP3()
    g.M, g.N, g.K = M, N, K
    g.A, g.B, g.lda, g.ldb = ptr(A3), ptr(B3), A3.stride(0), B3.stride(0)
    g.a_row_mod = a_row_mod
    if A3_second is not None:
        g.A2, g.a2_row0 = ptr(A3_second), second_row0
    g.a_map, g.a2_map, g.a_map_rows = ptr(a_map), ptr(a2_map), int(a_map_rows)      # int32 row maps: A / A2 are a store's packed planes, read in place
    if bits is not None:
        g.a_bits, g.bits_qw, g.a_scale = ptr(bits), bits.stride(0), scale
    g.bias, g.act = ptr(bias), act
    Cf = C_out if C_out is not None else (torch.empty(M, N, device=dev) if want_f32 else None)
    Cp = torch.empty(M, 6 * N, dtype=torch.uint8, device=dev) if want_p3 else None
    if Cf is not None:
        g.C, g.ldc = ptr(Cf), Cf.stride(0)
    if Cp is not None:
        g.C_p3, g.ldc_p3 = ptr(Cp), Cp.stride(0)
    g.splitk, g.tile_m = splitk, tile_m
    need = lib.sdumc_gemm_p3_workspace_bytes(C.byref(g))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    g.workspace, g.workspace_bytes = ptr(ws), need
    keep = (A3, B3, bias, bits, ws, Cf, Cp, A3_second, a_map, a2_map)

    def launch(_keep=keep):
        check(lib.sdumc_gemm_p3_nt(C.byref(g), _st()), "sdumc_gemm_p3_nt")
    res = (Cf, Cp) if (want_p3 and Cf is not None) else (Cp if Cf is None else Cf)
    return launch, res


def gemm_p3_nt(*args, **kw):
    """C[M, N] = act((A . keep) B^T * scale + bias) on P3 operands (sdumc_gemm_p3_nt).  Returns C (fp32), or (C, C_p3) with
    want_p3, or C_p3 alone with want_f32=False."""
    launch, res = gemm_p3_nt_call(*args, **kw)
    launch()
    return res


def b1_frag(W):
    """fp32 weight [rows, cols] -> the fragment-major bf16 operand of gemm_b1_nt (sdumc_b1_frag_multi): uint8 [rows / 32, 64 cols]."""
    rows, cols = W.shape
    W = W.contiguous()
    out = torch.empty(rows // 32, 64 * cols, dtype=torch.uint8, device=W.device)
    so, do = (C.c_int64 * 1)(0), (C.c_int64 * 1)(0)
    r, c = (C.c_int32 * 1)(rows), (C.c_int32 * 1)(cols)
    check(lib.sdumc_b1_frag_multi(ptr(W), ptr(out), so, do, r, c, 1, _st()), "sdumc_b1_frag_multi")
    return out


def gemm_b1_nt_call(A, Bf, M, N, K, bias=None, act=ACT_NONE, a_row_mod=0, out_dtype=torch.bfloat16, splitk=0, A_second=None,
                    second_row0=0, C_out=None, a_map=None, a2_map=None, a_map_rows=0):
    """The prepared call of sdumc_gemm_b1_nt (bf16 A [rows, K], fragment-major bf16 weight): returns (launch, C)."""
    dev = A.device
    g = _lib.GemmB1()
    g.M, g.N, g.K = M, N, K
    g.A, g.B, g.lda, g.ldb = ptr(A), ptr(Bf), A.stride(0), Bf.stride(0)
    g.a_row_mod = a_row_mod
    if A_second is not None:
        g.A2, g.a2_row0 = ptr(A_second), second_row0
    g.bias, g.act = ptr(bias), act
    Cm = C_out if C_out is not None else torch.empty(M, N, dtype=out_dtype, device=dev)
    g.C, g.ldc, g.c_bf16 = ptr(Cm), Cm.stride(0), int(Cm.dtype == torch.bfloat16)
    g.a_map, g.a2_map, g.a_map_rows = ptr(a_map), ptr(a2_map), int(a_map_rows)      # int32 row maps: A / A2 are a store's packed rows
    g.splitk = splitk
    need = lib.sdumc_gemm_b1_workspace_bytes(C.byref(g))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    g.workspace, g.workspace_bytes = ptr(ws), need
    keep = (A, Bf, bias, ws, Cm, A_second, a_map, a2_map)

    def launch(_keep=keep):
        check(lib.sdumc_gemm_b1_nt(C.byref(g), _st()), "sdumc_gemm_b1_nt")
    return launch, Cm


def gemm_b1_nt(*args, **kw):
    """C[M, N] = act(A B^T + bias), A bf16, B the fragment-major bf16 weight (b1_frag), C bf16 or fp32 (sdumc_gemm_b1_nt)."""
    launch, res = gemm_b1_nt_call(*args, **kw)
    launch()
    return res


def gemm_rows256(problems):
    """The tall 256 x 256 products of the frame-level part in one persistent launch (sdumc_gemm_rows256).

    problems: list of dicts {A, B, [C], [bits], [bias], [scale], [a_row_mod], [accumulate], [act], [M]}:
    C[M, 256] = act((A . keep) B * scale + bias) (+ C); A [rows, 256] and B [256 k, 256 n] 2-D fp32 device tensors, bits the
    uint8 keep-bits [M, 64] of a dropout fused on A's virtual rows.  All masked, all accumulating, or neither.
    torch.bfloat16 tensors go to sdumc_gemm_rows256_bf16: B is then [256 n, 256 k] (C = A B^T), C bf16, no bits."""
    n = len(problems)
    arr = (_lib.RowsProblem * n)()
    for i, q in enumerate(problems):
        g = arr[i]
        A, B = q["A"], q["B"]
        g.A, g.B = ptr(A), ptr(B)
        g.M = q.get("M", A.shape[0])
        g.lda, g.ldb = A.stride(0), B.stride(0)
        g.a_row_mod = q.get("a_row_mod", 0)
        g.a_bits = ptr(q.get("bits"))
        g.bias = ptr(q.get("bias"))
        g.a_scale = q.get("scale", 1.0)
        Cm = q.get("C")
        if Cm is None:
            Cm = torch.empty(g.M, 256, device=A.device, dtype=A.dtype)
            q["C"] = Cm
        g.C, g.ldc = ptr(Cm), Cm.stride(0)
        g.accumulate = 1 if q.get("accumulate") else 0
        g.act = q.get("act", ACT_NONE)
        if q.get("pool_w") is not None:      # + sum_i pool_w[r, i] * pool_g[r // pool_T, i, :]  (sdumc_rows_problem.pool_*)
            pw, pg = q["pool_w"], q["pool_g"]
            g.pool_w, g.pool_g, g.pool_nq, g.pool_T = ptr(pw), ptr(pg), pw.shape[1], q["pool_T"]
            if q.get("fold"):                # the mask-sum over `fold` row blocks folded in (sdumc_rows_problem.fold)
                g.fold, g.c_bits, g.c_scale = q["fold"], ptr(q.get("c_bits")), q.get("c_scale", 1.0)
    hf = problems[0]["A"].dtype == torch.bfloat16
    check((lib.sdumc_gemm_rows256_bf16 if hf else lib.sdumc_gemm_rows256)(arr, n, _st()), "sdumc_gemm_rows256")
    return [q["C"] for q in problems]


def gemm_bf16(layout, A, B, M, N, K, bias=None, C_out=None, lda=None, ldb=None, ldc=None, act=ACT_NONE, a_row_mod=0,
              b_row_mod=0, accumulate=False, c_bf16=False, splitk=0, colsum_a=None):
    """GEMM on bf16 storage (sdumc_gemm_bf16_run): A, B torch.bfloat16 device tensors (lists = grouped); C fp32 or bf16."""
    As = A if isinstance(A, (list, tuple)) else [A]
    Bs = B if isinstance(B, (list, tuple)) else [B]
    groups = len(As)
    dev = As[0].device
    if C_out is None:
        Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16 if c_bf16 else torch.float32) for _ in range(groups)]
    else:
        Cs = C_out if isinstance(C_out, (list, tuple)) else [C_out]
    biases = bias if isinstance(bias, (list, tuple)) else [bias] * groups
    g = _lib.GemmBf16()
    g.layout, g.M, g.N, g.K, g.groups = layout, M, N, K, groups
    for i in range(groups):
        g.A[i], g.B[i], g.C[i], g.bias[i] = ptr(As[i]), ptr(Bs[i]), ptr(Cs[i]), ptr(biases[i])
    g.lda = lda if lda is not None else (K if layout == NT else M)
    g.ldb = ldb if ldb is not None else (K if layout == NT else N)
    g.ldc = ldc if ldc is not None else N
    g.a_row_mod, g.b_row_mod = a_row_mod, b_row_mod
    g.act, g.accumulate, g.c_bf16, g.splitk = act, 1 if accumulate else 0, 1 if c_bf16 else 0, splitk
    if colsum_a is not None:
        cs = colsum_a if isinstance(colsum_a, (list, tuple)) else [colsum_a]
        for i in range(groups):
            g.colsum_a[i] = ptr(cs[i])
    need = lib.sdumc_gemm_bf16_workspace_bytes(C.byref(g))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    g.workspace, g.workspace_bytes = ptr(ws), need
    check(lib.sdumc_gemm_bf16_run(C.byref(g), _st()), "sdumc_gemm_bf16_run")
    return Cs if isinstance(A, (list, tuple)) else Cs[0]


def attnpool_desc(x, keys, q, V, T, nq, x_samples, q_stride, x_drop, out_drop, attn, pooled, out, scale=0.3, dim=0,
                  lengths=None, tickets=True):
    a = _lib.AttnPool()
    if tickets:      # zeroed per-sample counters: the combine / dq reduction run inside the first kernels
        a._keep_tickets = torch.zeros(2 * V, dtype=torch.int32, device=attn.device)
        a.tickets = ptr(a._keep_tickets)
    a.dim = dim
    a.lengths = ptr(lengths)   # int32 [V] key-padding extension, or None = the reference's behaviour
    a.V, a.T, a.nq, a.x_samples = V, T, nq, x_samples
    a.x, a.keys, a.q, a.q_stride, a.scale = ptr(x), ptr(keys), ptr(q), q_stride, scale
    if x_drop is not None:
        a.x_drop = x_drop
    if out_drop is not None:
        a.out_drop = out_drop
    a.attn, a.pooled, a.out = ptr(attn), ptr(pooled), ptr(out)
    return a


def attnpool_fwd(x, keys, q, nq, x_samples=None, q_shared=False, x_drop=None, out_drop=None, lengths=None, tickets=True):
    V, T, Dm = keys.shape
    dev = keys.device
    attn = torch.empty(V, T, nq, device=dev)
    pooled = torch.empty(V, nq, Dm, device=dev)
    out = torch.empty(V, nq, Dm, device=dev)
    if lengths is not None and (lengths.dtype != torch.int32 or not lengths.is_cuda or lengths.numel() != V):
        raise _lib.SdumcError("lengths must be a cuda int32 tensor with one entry per virtual sample")
    a = attnpool_desc(x, keys, q, V, T, nq, x_samples or x.shape[0], 0 if q_shared else nq * Dm, x_drop, out_drop,
                      attn, pooled, out, lengths=lengths, tickets=tickets)
    a._keep_lengths = lengths
    need = lib.sdumc_attnpool_fwd_workspace_bytes(V, T, nq)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    a.workspace, a.workspace_bytes = ptr(ws), need
    check(lib.sdumc_attnpool_fwd(C.byref(a), _st()), "sdumc_attnpool_fwd")
    return out, attn, pooled, a


def attnpool_bwd(desc, dout, keep, shared_q_sum=False):
    """desc: the AttnPool returned by attnpool_fwd; keep: tensors that must stay alive.  shared_q_sum (descriptors with a shared
    query, q_stride 0): dq comes back as the [1, nq, dim] SUM over the samples, from the backward's own single reduce launch."""
    V, T, nq = desc.V, desc.T, desc.nq
    dev = dout.device
    Dm = desc.dim or _lib.D
    fdt = torch.bfloat16 if desc.bf16 else torch.float32
    dz = torch.empty(V, T, Dm, device=dev, dtype=fdt)
    dxd = torch.empty(V, T, Dm, device=dev, dtype=fdt)
    dq = torch.empty(1 if shared_q_sum else V, nq, Dm, device=dev)
    need = lib.sdumc_attnpool_bwd_workspace_bytes_dim(V, T, nq, Dm)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    b = _lib.AttnPoolBwd()
    b.f = desc
    b.dout, b.dz, b.dxd = ptr(dout), ptr(dz), ptr(dxd)
    if shared_q_sum:
        b.dq_sum = ptr(dq)
    else:
        b.dq = ptr(dq)
    b.workspace, b.workspace_bytes = ptr(ws), need
    check(lib.sdumc_attnpool_bwd(C.byref(b), _st()), "sdumc_attnpool_bwd")
    return dz, dxd, dq


def umca_fwd(x, W, b, q, nq, x_samples=None, q_shared=False, x_drop=None, out_drop=None, lengths=None, want_keys=True, V=None,
             planes=False):
    """K3 (sdumc_umca_fwd): key projection + scores + softmax partials + pooling in one kernel.  x [x_samples, T, 256] fp32,
    W [256, 256], b [256]; x_drop must carry keep-bits (dropout_bits).  -> (out, attn, pooled, keys or None, desc); the desc
    (with keys attached) is what attnpool_bwd takes."""
    xs, T, Dm = x.shape
    V = V or (q.shape[0] if not q_shared else xs)
    dev = x.device
    attn = torch.empty(V, T, nq, device=dev)
    pooled = torch.empty(V, nq, Dm, device=dev)
    out = torch.empty(V, nq, Dm, device=dev)
    keys = torch.empty(V, T, Dm, device=dev) if want_keys else None
    a = attnpool_desc(x, keys, q, V, T, nq, x_samples or xs, 0 if q_shared else nq * Dm, x_drop, out_drop, attn, pooled, out,
                      lengths=lengths, tickets=False)
    need = lib.sdumc_attnpool_fwd_workspace_bytes(V, T, nq)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    a.workspace, a.workspace_bytes = ptr(ws), need
    u = _lib.Umca()
    u.a = a
    u.w_in, u.b_in = ptr(W), ptr(b)
    x3 = w3 = None
    if planes:      # the projection's operands split once per tensor (sdumc_umca.x_p3 / w_in_p3f)
        x3, w3 = p3_split(x.reshape(xs * T, Dm)), p3_split_frag(W)
        u.x_p3, u.w_in_p3f = ptr(x3), ptr(w3)
    check(lib.sdumc_umca_fwd(C.byref(u), _st()), "sdumc_umca_fwd")
    torch.cuda.current_stream().synchronize()
    a._keep = (ws, keys, x, q, W, b, lengths, x3, w3)
    return out, attn, pooled, keys, a


def attnpool_fwd_multi(sites):
    """sites: list of dicts with the keyword arguments of attnpool_fwd (x, keys, q, nq, ...); ONE launch pair for all of them
    (sdumc_attnpool_fwd_multi).  -> list of (out, attn, pooled, desc)."""
    arr = (_lib.AttnPool * len(sites))()
    res, keep = [], []
    for i, kw in enumerate(sites):
        x, keys, q, nq = kw["x"], kw["keys"], kw["q"], kw["nq"]
        V, T, Dm = keys.shape
        dev = keys.device
        attn, pooled, out = torch.empty(V, T, nq, device=dev), torch.empty(V, nq, Dm, device=dev), torch.empty(V, nq, Dm, device=dev)
        a = attnpool_desc(x, keys, q, V, T, nq, kw.get("x_samples") or x.shape[0], 0 if kw.get("q_shared") else nq * Dm,
                          kw.get("x_drop"), kw.get("out_drop"), attn, pooled, out, lengths=kw.get("lengths"),
                          tickets=kw.get("tickets", True))
        need = lib.sdumc_attnpool_fwd_workspace_bytes(V, T, nq)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        a.workspace, a.workspace_bytes = ptr(ws), need
        arr[i] = a
        keep.append(ws)
        res.append((out, attn, pooled, a))
    check(lib.sdumc_attnpool_fwd_multi(arr, len(sites), _st()), "sdumc_attnpool_fwd_multi")
    torch.cuda.current_stream().synchronize()      # (the workspaces die with this frame)
    return res


def attnpool_bwd_multi(descs, douts):
    """One launch pair for the backward of several sites (sdumc_attnpool_bwd_multi) -> list of (dz, dxd, dq)."""
    arr = (_lib.AttnPoolBwd * len(descs))()
    res, keep = [], []
    for i, (desc, dout) in enumerate(zip(descs, douts)):
        V, T, nq, Dm = desc.V, desc.T, desc.nq, _lib.D
        dev = dout.device
        dz, dxd, dq = torch.empty(V, T, Dm, device=dev), torch.empty(V, T, Dm, device=dev), torch.empty(V, nq, Dm, device=dev)
        need = lib.sdumc_attnpool_bwd_workspace_bytes(V, T, nq)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        b = _lib.AttnPoolBwd()
        b.f = desc
        b.dout, b.dz, b.dxd, b.dq = ptr(dout), ptr(dz), ptr(dxd), ptr(dq)
        b.workspace, b.workspace_bytes = ptr(ws), need
        arr[i] = b
        keep.append(ws)
        res.append((dz, dxd, dq))
    check(lib.sdumc_attnpool_bwd_multi(arr, len(descs), _st()), "sdumc_attnpool_bwd_multi")
    torch.cuda.current_stream().synchronize()
    return res


def dropout_mask(d, streams):
    n = streams * d.samples * max(d.rows, 1) * d.width
    dev = torch.device("cuda")
    m = torch.empty(n, device=dev)
    check(lib.sdumc_dropout_mask(C.byref(d), streams, ptr(m), _st()), "sdumc_dropout_mask")
    return m.view(streams * d.samples, max(d.rows, 1), d.width)


def dropout_bits(d, streams):
    """uint8 keep-bits [streams*samples*rows, width/4] for the row space of `d`; also attaches them to `d`."""
    n = streams * d.samples * max(d.rows, 1) * (d.width // 4)
    bits = torch.empty(n, dtype=torch.uint8, device="cuda")
    saved, d.bits = d.bits, None
    check(lib.sdumc_dropout_bits(C.byref(d), streams, ptr(bits), _st()), "sdumc_dropout_bits")
    d.bits = ptr(bits)
    return bits


def dropout_bits_apply_bf16(d, streams, site_stride, x):
    """bf16 storage: keep-bits of sites d.site and d.site + site_stride and the masked frames of both, in one pass over x
    (sdumc_dropout_bits_apply_bf16).  x: bf16 [x_rows, width]; returns (bits0, bits1, xd0, xd1)."""
    rows = streams * d.samples * max(d.rows, 1)
    bits = [torch.empty(rows * (d.width // 4), dtype=torch.uint8, device=x.device) for _ in range(2)]
    xd = [torch.empty(rows, d.width, dtype=torch.bfloat16, device=x.device) for _ in range(2)]
    barr = (C.c_void_p * 2)(ptr(bits[0]), ptr(bits[1]))
    xarr = (C.c_void_p * 2)(ptr(xd[0]), ptr(xd[1]))
    saved, d.bits = d.bits, None
    check(lib.sdumc_dropout_bits_apply_bf16(C.byref(d), streams, site_stride, barr, ptr(x), x.shape[0], xarr, _st()),
          "sdumc_dropout_bits_apply_bf16")
    d.bits = saved
    return bits[0], bits[1], xd[0], xd[1]


def colsum(a, accumulate_into=None):
    rows, cols = a.shape
    out = accumulate_into if accumulate_into is not None else torch.empty(cols, device=a.device)
    ws = torch.empty(max(lib.sdumc_colsum_workspace_bytes(rows, cols), 16), dtype=torch.uint8, device=a.device)
    check(lib.sdumc_colsum(ptr(a), rows, cols, a.stride(0), ptr(out), 1 if accumulate_into is not None else 0,
                           ptr(ws), _st()), "sdumc_colsum")
    return out


def mse_fwd_bwd(pred, target, weight=1.0, denom=None):
    rows = pred.numel()
    loss = torch.empty(1, device=pred.device)
    dpred = torch.empty_like(pred)
    check(lib.sdumc_mse_fwd_bwd(ptr(pred), ptr(target), rows, float(denom or rows), weight, ptr(loss), ptr(dpred),
                                _st()), "sdumc_mse_fwd_bwd")
    return loss, dpred


def ssd(a, b):
    n = a.numel()
    out = torch.empty(1, device=a.device)
    ws = torch.empty(max(lib.sdumc_ssd_workspace_bytes(n), 16), dtype=torch.uint8, device=a.device)
    check(lib.sdumc_ssd(ptr(a), ptr(b), n, ptr(out), ptr(ws), _st()), "sdumc_ssd")
    return out


def rmse_fwd_bwd(a, b, weight=1.0, ssd_global=None, numel_global=None, need_db=True):
    s = ssd_global if ssd_global is not None else ssd(a, b)
    loss = torch.empty(1, device=a.device)
    da = torch.empty_like(a)
    db = torch.empty_like(b) if need_db else None
    check(lib.sdumc_rmse_bwd(ptr(a), ptr(b), a.numel(), ptr(s), float(numel_global or a.numel()), weight, ptr(loss),
                             ptr(da), 0, ptr(db), 0, _st()), "sdumc_rmse_bwd")
    return loss, da, db


def rnc_fwd_bwd(feats, labels, temperature=2.0, weight=1.0, row0=0, rows_local=None):
    n, dim = feats.shape
    rows_local = n if rows_local is None else rows_local
    loss = torch.empty(1, device=feats.device)
    df = torch.empty(rows_local, dim, device=feats.device)
    ws = torch.empty(lib.sdumc_rnc_workspace_bytes(n), dtype=torch.uint8, device=feats.device)
    check(lib.sdumc_rnc_fwd_bwd(ptr(feats), ptr(labels), n, dim, temperature, weight, row0, rows_local, ptr(loss),
                                ptr(df), ptr(ws), _st()), "sdumc_rnc_fwd_bwd")
    return loss, df, ws


def rnc_dfeat_rows(feats, ws, row0, rows, temperature=2.0, weight=1.0):
    n, dim = feats.shape
    df = torch.empty(rows, dim, device=feats.device)
    check(lib.sdumc_rnc_dfeat_rows(ptr(feats), n, dim, temperature, weight, row0, rows, ptr(df), ptr(ws), _st()),
          "sdumc_rnc_dfeat_rows")
    return df


def rnc_mask(labels):
    n = labels.numel()
    m = torch.empty(n, n - 1, n - 1, dtype=torch.uint8, device=labels.device)
    check(lib.sdumc_rnc_mask(ptr(labels), n, ptr(m), _st()), "sdumc_rnc_mask")
    return m


def adam_step(param, grad, m, v, hyper, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-5, grad_scale=1.0):
    check(lib.sdumc_adam_step(ptr(param), ptr(grad), ptr(m), ptr(v), param.numel(), ptr(hyper), beta1, beta2, eps,
                              weight_decay, grad_scale, _st()), "sdumc_adam_step")


# ---- generic MHA / Transformer-encoder pieces (transformer.hip) ------------------------------------
def layernorm_fwd(x, gamma, beta, eps=1e-5):
    width = x.shape[-1]
    rows = x.numel() // width
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device)
    rstd = torch.empty(rows, device=x.device)
    check(lib.sdumc_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, width, eps,
                                  _st()), "sdumc_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dx_add=None, need_params=True):
    """dx = LayerNorm backward (+ dx_add: the residual branch's gradient at x, summed in the same kernel)."""
    width = x.shape[-1]
    rows = x.numel() // width
    dx = torch.empty_like(x)
    dg = torch.empty(width, device=x.device) if need_params else None
    db = torch.empty(width, device=x.device) if need_params else None
    need = lib.sdumc_layernorm_bwd_workspace_bytes(rows, width)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    check(lib.sdumc_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dg), ptr(db),
                                  ptr(dx_add), rows, width, ptr(ws), need, _st()), "sdumc_layernorm_bwd")
    return dx, dg, db


def softmax_desc(scores, batch, heads, scale, mask=None, drop=None, probs_drop=None, weights=None):
    bh, tq, tk = scores.shape
    assert bh == batch * heads
    s = _lib.Softmax()
    s.batch, s.heads, s.tq, s.tk, s.scale = batch, heads, tq, tk, scale
    s.mask, s.scores, s.probs_drop, s.weights = ptr(mask), ptr(scores), ptr(probs_drop), ptr(weights)
    if drop is not None:
        s.drop = drop
    return s


def softmax_fwd(scores, batch, heads, scale, mask=None, drop=None, need_weights=True):
    """In place on `scores` ([batch*heads, tq, tk]).  Returns (P, P_dropped or None, head-mean weights or None, desc)."""
    bh, tq, tk = scores.shape
    pd = torch.empty_like(scores) if drop is not None and drop.enabled else None
    w = torch.empty(batch, tq, tk, device=scores.device) if need_weights else None
    s = softmax_desc(scores, batch, heads, scale, mask, drop, pd, w)
    check(lib.sdumc_softmax_fwd(C.byref(s), _st()), "sdumc_softmax_fwd")
    return scores, pd, w, s


def softmax_bwd(desc, dscores):
    check(lib.sdumc_softmax_bwd(C.byref(desc), ptr(dscores), _st()), "sdumc_softmax_bwd")
    return dscores


def drop_add(x, residual=None, drop=None, alpha=1.0, pos_table=None, pos_src=None, out=None):
    """y = drop(alpha * x + pos) + residual over a [samples, rows, width] tensor."""
    samples, rows, width = x.shape
    y = out if out is not None else torch.empty_like(x)
    d = _lib.DropAdd()
    d.x, d.alpha, d.pos_table, d.pos_src, d.residual, d.y = ptr(x), alpha, ptr(pos_table), ptr(pos_src), ptr(residual), ptr(y)
    d.samples, d.rows, d.width = samples, rows, width
    if drop is not None:
        d.drop = drop
    check(lib.sdumc_drop_add(C.byref(d), _st()), "sdumc_drop_add")
    return y


def mha_forward(query, key, value, w_in, b_in, w_out, b_out, heads, attn_mask=None, attn_drop=None, need_weights=True,
                bf16=False, bias_k=None, bias_v=None, add_zero_attn=False):
    """MultiheadAttention.forward on [T, B, E] tensors.  Returns (out, weights, saved) where `saved` is what
    mha_backward needs.  bias_k / bias_v ([E]) and add_zero_attn lengthen the source by one row each
    (multihead_attention.py:86-104): the weights are then [B, T_q, T_k + extras]."""
    tq, B, E = query.shape
    tk = key.shape[0]
    ts = tk + (1 if bias_k is not None else 0) + (1 if add_zero_attn else 0)
    dev = query.device
    m = _lib.Mha()
    m.tq, m.tk, m.batch, m.embed, m.heads = tq, tk, B, E, heads
    m.query, m.key, m.value = ptr(query), ptr(key), ptr(value)
    m.in_proj_weight, m.in_proj_bias, m.out_proj_weight, m.out_proj_bias = ptr(w_in), ptr(b_in), ptr(w_out), ptr(b_out)
    m.attn_mask = ptr(attn_mask)
    m.bias_k, m.bias_v, m.add_zero_attn = ptr(bias_k), ptr(bias_v), 1 if add_zero_attn else 0
    m.bf16 = 1 if bf16 else 0
    drop_on = attn_drop is not None and attn_drop.enabled
    if drop_on:
        m.attn_drop = attn_drop
    t = {"out": torch.empty(tq, B, E, device=dev),
         "weights": torch.empty(B, tq, ts, device=dev) if need_weights else None,
         "q": torch.empty(tq, B, E, device=dev), "k": torch.empty(ts, B, E, device=dev),
         "v": torch.empty(ts, B, E, device=dev), "probs": torch.empty(B * heads, tq, ts, device=dev),
         "probs_drop": torch.empty(B * heads, tq, ts, device=dev) if drop_on else None,
         "ctx": torch.empty(tq, B, E, device=dev)}
    for name, ten in t.items():
        setattr(m, name, ptr(ten))
    need = lib.sdumc_mha_workspace_bytes(C.byref(m), 0)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    m.workspace, m.workspace_bytes = ptr(ws), need
    check(lib.sdumc_mha_forward(C.byref(m), _st()), "sdumc_mha_forward")
    t["inputs"] = (query, key, value, w_in, b_in, w_out, b_out, attn_mask)   # keep-alive
    t["bias_kv"] = (bias_k, bias_v)
    return t["out"], t["weights"], (m, t)


def mha_backward(saved, dout):
    """Returns (dquery, dkey, dvalue, dw_in, db_in, dw_out, db_out[, d_bias_k, d_bias_v when add_bias_kv]).  Inputs that
    aliased in the forward share ONE summed gradient buffer; it is returned once (for the first of them) and the others come
    back as None."""
    m, t = saved
    query, key, value, w_in, b_in, w_out, b_out, _ = t["inputs"]
    dev = dout.device
    dq = torch.empty_like(query)
    dk = dq if key.data_ptr() == query.data_ptr() else torch.empty_like(key)
    dv = dq if value.data_ptr() == query.data_ptr() else (dk if value.data_ptr() == key.data_ptr() else torch.empty_like(value))
    g = _lib.MhaGrads()
    dw_in, dw_out = torch.empty_like(w_in), torch.empty_like(w_out)
    db_in = torch.empty_like(b_in) if b_in is not None else None
    db_out = torch.empty_like(b_out) if b_out is not None else None
    g.dout, g.dquery, g.dkey, g.dvalue = ptr(dout), ptr(dq), ptr(dk), ptr(dv)
    g.d_in_proj_weight, g.d_in_proj_bias, g.d_out_proj_weight, g.d_out_proj_bias = ptr(dw_in), ptr(db_in), ptr(dw_out), ptr(db_out)
    bias_k, bias_v = t.get("bias_kv", (None, None))
    dbk = torch.empty_like(bias_k) if bias_k is not None else None
    dbv = torch.empty_like(bias_v) if bias_v is not None else None
    g.d_bias_k, g.d_bias_v = ptr(dbk), ptr(dbv)
    need = lib.sdumc_mha_workspace_bytes(C.byref(m), 1)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    m.workspace, m.workspace_bytes = ptr(ws), need
    check(lib.sdumc_mha_backward(C.byref(m), C.byref(g), _st()), "sdumc_mha_backward")
    res = (dq, (None if dk is dq else dk), (None if dv is dq or dv is dk else dv), dw_in, db_in, dw_out, db_out)
    return res + (dbk, dbv) if bias_k is not None else res
