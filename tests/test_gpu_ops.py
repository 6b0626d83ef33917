"""GPU parity tests, operator level: each C-ABI kernel against the CPU oracle
(torch fp32/fp64 on the host).  Tolerances: fp32 1e-3 is the north-star bar;
these tests hold the kernels to 1e-4 or tighter.  Dropout masks, RnC masks and
index math are bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import ops as o
    return o


def dev(t):
    return t.cuda().contiguous()


def close(got, want, tol=1e-4, msg=""):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy() if isinstance(want, torch.Tensor) else np.asarray(want, dtype=np.float64)
    scale = max(1.0, np.abs(want).max())
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol * scale, err_msg=msg)


def test_dropout_mask_bit_exact(ops):
    from oracle import philox
    from sdumc_amd._lib import make_dropout
    for p, rows, width, samples, streams, sample0, site, call, seed in [
            (0.5, 11, 256, 3, 2, 0, 4, 7, 12345), (0.3, 7, 128, 5, 1, 40, 28, 2, (1 << 40) + 17),
            (0.3, 1, 256, 4, 2, 3, 12, 0, 99)]:
        d = make_dropout(True, site, p, rows, width, samples, sample0=sample0, call0=call, seed=seed)
        got = ops.dropout_mask(d, streams).cpu().numpy()
        want = np.concatenate([philox.dropout_mask(samples, rows, width, p, seed, call + s, site, sample0)
                               for s in range(streams)])
        np.testing.assert_array_equal(got, want)
        # precomputed keep-bits decode to the same mask, and kernels fed with them give identical results
        bits = ops.dropout_bits(d, streams)
        unpacked = ((bits.cpu().numpy()[:, None] >> np.arange(4)) & 1).reshape(want.shape).astype(np.float32)
        np.testing.assert_array_equal(unpacked * np.float32(d.scale), want)
        np.testing.assert_array_equal(ops.dropout_mask(d, streams).cpu().numpy(), want)   # d.bits now attached
        d.bits = None
    # device-resident state overrides seed/call
    st = torch.tensor([5, 0, 9], dtype=torch.int32).cuda()
    d = make_dropout(True, 3, 0.5, 2, 8, 2, dev_state=st)
    np.testing.assert_array_equal(ops.dropout_mask(d, 1).cpu().numpy(), philox.dropout_mask(2, 2, 8, 0.5, 5, 9, 3))


def test_dropout_bits_and_masked_frames_in_one_pass_bf16(ops):
    """sdumc_dropout_bits_apply_bf16 == sdumc_dropout_bits (per site) followed by sdumc_mask_apply_bf16, bit for bit; the two
    streams read the same frames (row modulo)."""
    import ctypes as C
    from sdumc_amd._lib import lib, make_dropout, ptr
    for samples, T, streams, stride in [(3, 37, 2, 24), (2, 64, 1, 5)]:
        g = torch.Generator().manual_seed(samples * T)
        x = torch.randn(samples * T, 256, generator=g).to(torch.bfloat16).cuda()
        d = make_dropout(True, 4, 0.5, T, 256, samples, call0=3, seed=777)
        b0, b1, xd0, xd1 = ops.dropout_bits_apply_bf16(d, streams, stride, x)
        rows = streams * samples * T
        for site, bits, xd in ((4, b0, xd0), (4 + stride, b1, xd1)):
            ds = make_dropout(True, site, 0.5, T, 256, samples, call0=3, seed=777)
            want_bits = ops.dropout_bits(ds, streams)
            assert torch.equal(bits, want_bits)
            want = torch.empty(rows, 256, dtype=torch.bfloat16, device="cuda")
            assert lib.sdumc_mask_apply_bf16(ptr(x), ptr(want_bits), ptr(want), rows, samples * T, 256, C.c_float(ds.scale),
                                             None) == 0
            torch.cuda.synchronize()
            assert torch.equal(xd.view(torch.int16), want.view(torch.int16))
            ds.bits = None


@pytest.mark.parametrize("M,N,K,tile", [(128, 256, 256, 0), (300, 256, 1024, 1), (77, 3, 256, 2), (64, 7, 128, 0),
                                        (513, 130, 96, 1), (2048, 256, 4096, 0), (5, 1, 128, 0), (300, 256, 1024, 4),
                                        (129, 65, 40, 4), (16384, 256, 1024, 0)])
def test_gemm_nt_bias_act(ops, M, N, K, tile):
    g = torch.Generator().manual_seed(M * 7 + N)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    for act, f in ((ops.ACT_NONE, lambda x: x), (ops.ACT_RELU, torch.relu), (ops.ACT_TANH, torch.tanh)):
        got = ops.gemm(ops.NT, dev(A), dev(W), M, N, K, bias=dev(b), act=act, tile=tile)
        close(got, f(ref), 2e-5, f"act={act}")


def test_gemm_nn_tn_accumulate_splitk(ops):
    g = torch.Generator().manual_seed(3)
    M, N, K = 200, 256, 384
    A, Bm = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) / K ** 0.5
    close(ops.gemm(ops.NN, dev(A), dev(Bm), M, N, K), A.double() @ Bm.double(), 2e-5)
    C0 = torch.randn(M, N, generator=g)
    Cd = dev(C0.clone())
    ops.gemm(ops.NN, dev(A), dev(Bm), M, N, K, C_out=Cd, accumulate=True)
    close(Cd, C0.double() + A.double() @ Bm.double(), 2e-5)
    # TN: dW = dY^T X with a long reduction dimension, with and without split-K (deterministic)
    R, Mo, Ni = 5000, 256, 320
    dY, X = torch.randn(R, Mo, generator=g), torch.randn(R, Ni, generator=g)
    ref = dY.double().T @ X.double()
    for sk in (1, 7, 16):
        got = ops.gemm(ops.TN, dev(dY), dev(X), Mo, Ni, R, splitk=sk)
        close(got, ref, 2e-5, f"splitk={sk}")
        again = ops.gemm(ops.TN, dev(dY), dev(X), Mo, Ni, R, splitk=sk)
        assert torch.equal(got, again), "split-K reduction must be bitwise reproducible"
    acc = dev(torch.ones(Mo, Ni))
    ops.gemm(ops.TN, dev(dY), dev(X), Mo, Ni, R, C_out=acc, splitk=5, accumulate=True)
    close(acc, ref + 1, 2e-5)
    # fused column sums of A (the bias gradient), every split mode incl. auto (0), with accumulate
    for sk in (0, 1, 6):
        for tile in (1, 2, 4):
            cs = dev(torch.full((Mo,), 2.0))
            acc = dev(torch.ones(Mo, Ni))
            ops.gemm(ops.TN, dev(dY), dev(X), Mo, Ni, R, C_out=acc, splitk=sk, tile=tile, accumulate=True, colsum_a=cs)
            close(acc, ref + 1, 2e-5, f"splitk={sk} tile={tile}")
            close(cs, dY.double().sum(0) + 2, 2e-5, f"colsum splitk={sk} tile={tile}")
    cs = torch.empty(3).cuda()
    got = ops.gemm(ops.TN, dev(dY[:130, :3].contiguous()), dev(X[:130, :256].contiguous()), 3, 256, 130, splitk=0, colsum_a=cs)
    close(cs, dY[:130, :3].double().sum(0), 2e-5)
    # auto plan on forward shapes (split-K + bias/activation in the reduce)
    A2, W2, b2 = torch.randn(2048, 4096, generator=g), torch.randn(256, 4096, generator=g) / 64, torch.randn(256, generator=g)
    close(ops.gemm(ops.NT, dev(A2), dev(W2), 2048, 256, 4096, bias=dev(b2), act=ops.ACT_RELU, splitk=0),
          torch.relu(A2.double() @ W2.double().T + b2.double()), 2e-5)
    # unaligned leading dimensions (fc_att backward: [rows,3])
    dY3, X3 = torch.randn(130, 3, generator=g), torch.randn(130, 256, generator=g)
    close(ops.gemm(ops.TN, dev(dY3), dev(X3), 3, 256, 130), dY3.double().T @ X3.double(), 2e-5)
    W3 = torch.randn(3, 256, generator=g)
    close(ops.gemm(ops.NN, dev(dY3), dev(W3), 130, 256, 3), dY3.double() @ W3.double(), 2e-5)


@pytest.mark.parametrize("M,N,K", [(128, 256, 256), (896, 128, 256), (128, 3, 256), (128, 896, 256), (130, 70, 100),
                                   (128, 256, 896), (128, 1, 128), (5, 64, 64), (128, 256, 7)])
def test_gemm_small_kernel_all_layouts(ops, M, N, K):
    """tile=3: 32x32 tile, the four waves split K; NT / NN / TN, bias + activation, accumulate, fused colsum."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    ref = A.double() @ W.double().T
    close(ops.gemm(ops.NT, dev(A), dev(W), M, N, K, bias=dev(b), act=ops.ACT_RELU, tile=3), torch.relu(ref + b.double()), 2e-5)
    auto = ops.gemm(ops.NT, dev(A), dev(W), M, N, K, bias=dev(b), splitk=0)          # planner picks it for these sizes
    close(auto, ref + b.double(), 2e-5)
    Wt = W.T.contiguous()
    C0 = torch.randn(M, N, generator=g)
    Cd = dev(C0.clone())
    ops.gemm(ops.NN, dev(A), dev(Wt), M, N, K, C_out=Cd, accumulate=True, tile=3)
    close(Cd, C0.double() + ref, 2e-5)
    At = A.T.contiguous()                                                          # TN: [K, M]^T [K, N]
    X = torch.randn(M, N, generator=g)
    cs = dev(torch.full((K,), 1.5))
    got = ops.gemm(ops.TN, dev(A), dev(X), K, N, M, tile=3, colsum_a=cs, accumulate=False)
    close(got, A.double().T @ X.double(), 2e-5)
    close(cs, A.double().sum(0), 2e-5)
    again = ops.gemm(ops.TN, dev(A), dev(X), K, N, M, tile=3)
    assert torch.equal(got, again)


@pytest.mark.parametrize("layout", ["NN", "TN"])
@pytest.mark.parametrize("M,N,K,tile", [(300, 256, 1024, 2), (2048, 256, 256, 0), (256, 1024, 5000, 0), (512, 128, 96, 1)])
def test_gemm_bf16_row_contiguous_layouts(ops, layout, M, N, K, tile):
    """bf16-operand mode of the NN (dX) and TN (dW, with the fused column sums) products: the row-contiguous operands are
    transposed by their LDS stores.  Reference: the product of the bf16-rounded operands in float64 (tight), and the fp32
    product (bf16 tolerance)."""
    g = torch.Generator().manual_seed(M + N + K)
    if layout == "NN":
        A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) / K ** 0.5
        ref32 = A.double() @ B.double()
        refbf = A.bfloat16().double() @ B.bfloat16().double()
        got = ops.gemm(ops.NN, dev(A), dev(B), M, N, K, tile=tile, splitk=0, bf16=True)
    else:
        A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g) / K ** 0.5
        ref32 = A.double().T @ B.double()
        refbf = A.bfloat16().double().T @ B.bfloat16().double()
        cs = dev(torch.zeros(M))
        got = ops.gemm(ops.TN, dev(A), dev(B), M, N, K, tile=tile, splitk=0, bf16=True, colsum_a=cs)
        close(cs, A.double().sum(0), 2e-5)          # the column sums ride on the fp32 staging registers: exact fp32
    close(got, refbf, 2e-5)
    close(got, ref32, 2e-2)


@pytest.mark.parametrize("M,N,K,tile", [(300, 256, 1024, 2), (2048, 256, 4096, 0), (513, 130, 96, 1), (48000, 256, 256, 0)])
def test_gemm_bf16_operand_mode(ops, M, N, K, tile):
    """bf16=True: operands rounded to bf16 (RNE) while staged, fp32 accumulate: must equal the fp64 product of the
    bf16-rounded operands to fp32-accumulation accuracy, and stay within bf16 distance of the exact product."""
    g = torch.Generator().manual_seed(M + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    got = ops.gemm(ops.NT, dev(A), dev(W), M, N, K, bias=dev(b), act=ops.ACT_TANH, tile=tile, splitk=0, bf16=True)
    Ab, Wb = A.bfloat16().double(), W.bfloat16().double()
    close(got, torch.tanh(Ab @ Wb.T + b.double()), 2e-5)
    close(got, torch.tanh(A.double() @ W.double().T + b.double()), 2e-2)


def test_gemm_grouped_strided_and_dropout(ops):
    from oracle import philox
    from sdumc_amd._lib import make_dropout
    g = torch.Generator().manual_seed(5)
    V, Dm = 24, 256
    xs = [torch.randn(V, Dm, generator=g) for _ in range(3)]
    Ws = [torch.randn(Dm, Dm, generator=g) / 16 for _ in range(3)]
    bs = [torch.randn(Dm, generator=g) for _ in range(3)]
    out = torch.zeros(V, 3 * Dm).cuda()
    seed, call = 77, 4
    cd = make_dropout(True, 7, 0.3, 1, Dm, 12, sample0=2, call0=call, seed=seed)  # 2 streams x 12 samples
    ops.gemm(ops.NT, [dev(x) for x in xs], [dev(w) for w in Ws], V, Dm, Dm, bias=[dev(b) for b in bs],
             C_out=[out[:, m * Dm:] for m in range(3)], ldc=3 * Dm, act=ops.ACT_RELU, c_drop=cd, c_drop_group_stride=2)
    for m in range(3):
        mask = np.concatenate([philox.dropout_mask(12, 1, Dm, 0.3, seed, call + s, 7 + 2 * m, 2) for s in range(2)])
        ref = torch.relu(xs[m].double() @ Ws[m].double().T + bs[m].double()) * torch.from_numpy(mask.reshape(V, Dm)).double()
        close(out[:, m * Dm:(m + 1) * Dm], ref, 2e-5, f"group {m}")


def test_gemm_fused_input_dropout_and_row_mod(ops):
    """K = tanh(drop(x) W^T + b) with both streams sharing x (a_row_mod), and dW = dz^T drop(x) (TN b_drop)."""
    from oracle import philox
    from sdumc_amd._lib import make_dropout
    g = torch.Generator().manual_seed(8)
    B, T, Dm, S = 3, 37, 256, 2
    x = torch.randn(B, T, Dm, generator=g)
    W, b = torch.randn(Dm, Dm, generator=g) / 16, torch.randn(Dm, generator=g)
    seed, call, site = 4242, 10, 21
    ad = make_dropout(True, site, 0.5, T, Dm, B, call0=call, seed=seed)
    got = ops.gemm(ops.NT, dev(x), dev(W), S * B * T, Dm, Dm, bias=dev(b), act=ops.ACT_TANH, a_row_mod=B * T, a_drop=ad)
    masks = np.concatenate([philox.dropout_mask(B, T, Dm, 0.5, seed, call + s, site) for s in range(S)])
    xd = torch.cat([x, x]).double() * torch.from_numpy(masks).double()
    close(got, torch.tanh(xd.reshape(-1, Dm) @ W.double().T + b.double()), 2e-5)
    dz = torch.randn(S * B * T, Dm, generator=g)
    gw = ops.gemm(ops.TN, dev(dz), dev(x), Dm, Dm, S * B * T, b_row_mod=B * T, b_drop=ad, splitk=3)
    close(gw, dz.double().T @ xd.reshape(-1, Dm), 2e-5)


def _attn_ref(x, W, b, q, xmask, omask, dtype=torch.float64):
    xd = (x * xmask).to(dtype)
    keys = torch.tanh(xd @ W.to(dtype).T + b.to(dtype))
    s = 0.3 * keys @ q.to(dtype).transpose(1, 2)
    a = torch.softmax(s, dim=1)
    pooled = a.transpose(1, 2) @ xd
    return pooled * omask.to(dtype), a, pooled, keys


@pytest.mark.parametrize("nq,T,shared_q", [(7, 37, False), (1, 5, True), (7, 130, False), (1, 375, True)])
def test_attnpool_fwd_bwd(ops, nq, T, shared_q):
    from oracle import philox
    from sdumc_amd._lib import make_dropout
    g = torch.Generator().manual_seed(nq * 100 + T)
    B, S, Dm = 3, 2, 256
    V = B * S
    x = torch.randn(B, T, Dm, generator=g)
    W, b = torch.randn(Dm, Dm, generator=g) / 16, torch.randn(Dm, generator=g) * 0.1
    q = torch.randn(1 if shared_q else V, nq, Dm, generator=g) / 4
    seed, call = 31, 6
    xdrop = make_dropout(True, 23, 0.5, T, Dm, B, call0=call, seed=seed)
    odrop = make_dropout(True, 24, 0.5, nq, Dm, B, call0=call, seed=seed)
    xm = torch.from_numpy(np.concatenate([philox.dropout_mask(B, T, Dm, 0.5, seed, call + s, 23) for s in range(S)]))
    om = torch.from_numpy(np.concatenate([philox.dropout_mask(B, nq, Dm, 0.5, seed, call + s, 24) for s in range(S)]))
    xx = torch.cat([x] * S).double().requires_grad_()
    Wd, bd, qd = W.double().requires_grad_(), b.double().requires_grad_(), q.double().requires_grad_()
    out_ref, a_ref, pooled_ref, keys_ref = _attn_ref(xx, Wd, bd, qd.expand(V, nq, Dm), xm, om)
    # HIP: keys from the fused GEMM, then the pooling kernels
    xg = dev(x)
    keys = ops.gemm(ops.NT, xg, dev(W), V * T, Dm, Dm, bias=dev(b), act=ops.ACT_TANH, a_row_mod=B * T,
                    a_drop=xdrop).view(V, T, Dm)
    qg = dev(q)
    out, attn, pooled, desc = ops.attnpool_fwd(xg, keys, qg, nq, x_samples=B, q_shared=shared_q, x_drop=xdrop,
                                               out_drop=odrop)
    close(keys, keys_ref, 2e-5)
    close(attn, a_ref, 2e-5)
    close(pooled, pooled_ref, 2e-5)
    close(out, out_ref, 2e-5)
    np.testing.assert_allclose(attn.sum(1).cpu().numpy(), 1.0, rtol=1e-5)
    # backward
    dout = torch.randn(V, nq, Dm, generator=g)
    out_ref.backward(dout.double())
    dz, dxd, dq = ops.attnpool_bwd(desc, dev(dout), (xg, keys, qg))
    # dW, db through the key projection; dxd total = pooling path + dz W
    gw = ops.gemm(ops.TN, dz.view(-1, Dm), xg, Dm, Dm, V * T, b_row_mod=B * T, b_drop=xdrop, splitk=2)
    close(gw, Wd.grad, 1e-4)
    close(ops.colsum(dz.view(-1, Dm)), bd.grad, 1e-4)
    ops.gemm(ops.NN, dz.view(-1, Dm), dev(W), V * T, Dm, Dm, C_out=dxd.view(-1, Dm), accumulate=True)
    close(dxd * xm.cuda(), xx.grad, 1e-4)
    dq_ref = qd.grad
    close(dq.sum(0, keepdim=True) if shared_q else dq, dq_ref, 1e-4)


@pytest.mark.parametrize("nq,T,shared_q,mask", [(7, 130, False, True), (1, 375, True, True), (7, 37, False, False), (7, 64, False, True)])
def test_umca_fused_forward_k3(ops, nq, T, shared_q, mask):
    """K3 (sdumc_umca_fwd: projection + scores + softmax + pooling in one kernel) against the fp64 restatement of
    Cross_Attention / FRA2UTT_new (model :79-95, :56-68) and against the two-kernel composition it replaces; with and
    without keeping the keys; its keys then drive the ordinary backward."""
    from oracle import philox
    from sdumc_amd._lib import make_dropout
    g = torch.Generator().manual_seed(nq * 100 + T)
    B, S, Dm = 3, 2, 256
    V = B * S
    x = torch.randn(B, T, Dm, generator=g)
    W, b = torch.randn(Dm, Dm, generator=g) / 16, torch.randn(Dm, generator=g) * 0.1
    q = torch.randn(1 if shared_q else V, nq, Dm, generator=g) / 4
    seed, call = 31, 6
    xdrop = make_dropout(mask, 23, 0.5, T, Dm, B, call0=call, seed=seed)
    odrop = make_dropout(True, 24, 0.5, nq, Dm, B, call0=call, seed=seed)
    ones = torch.ones(V, T, Dm, dtype=torch.float64)
    xm = torch.from_numpy(np.concatenate([philox.dropout_mask(B, T, Dm, 0.5, seed, call + s, 23) for s in range(S)])) if mask else ones
    om = torch.from_numpy(np.concatenate([philox.dropout_mask(B, nq, Dm, 0.5, seed, call + s, 24) for s in range(S)]))
    xx = torch.cat([x] * S).double()
    out_ref, a_ref, pooled_ref, keys_ref = _attn_ref(xx, W.double(), b.double(), q.double().expand(V, nq, Dm), xm, om)
    xg, Wg, bg, qg = dev(x), dev(W), dev(b), dev(q)
    bits = ops.dropout_bits(xdrop, S) if mask else None
    out, attn, pooled, keys, desc = ops.umca_fwd(xg, Wg, bg, qg, nq, x_samples=B, q_shared=shared_q, x_drop=xdrop if mask else None,
                                                 out_drop=odrop, V=V)
    close(keys, keys_ref, 2e-5)
    close(attn, a_ref, 2e-5)
    close(pooled, pooled_ref, 2e-5)
    close(out, out_ref, 2e-5)
    # the composition K3 replaces: NT GEMM + the pooling pair.  With every product on the fp32 MFMAs (sdumc_set_split_(0)) the
    # two sides run the same tile arithmetic and agree to 1e-6; K3's default -- the products of the key projection on the bf16
    # matrix pipe from exactly split operands -- is a different summation of the same exact products: within 5e-6 of it (and,
    # above, as close to fp64)
    from sdumc_amd import _lib
    try:
        _lib.lib.sdumc_set_split_(0)
        keys2 = ops.gemm(ops.NT, xg, Wg, V * T, Dm, Dm, bias=bg, act=ops.ACT_TANH, a_row_mod=B * T,
                         a_drop=xdrop if mask else None).view(V, T, Dm)
        out2, attn2, pooled2, _ = ops.attnpool_fwd(xg, keys2, qg, nq, x_samples=B, q_shared=shared_q, x_drop=xdrop if mask else None,
                                                   out_drop=odrop, tickets=False)
        out_f, attn_f, pooled_f, keys_f, _ = ops.umca_fwd(xg, Wg, bg, qg, nq, x_samples=B, q_shared=shared_q,
                                                          x_drop=xdrop if mask else None, out_drop=odrop, V=V)
    finally:
        _lib.lib.sdumc_set_split_(int(os.environ.get("SDUMC_SPLIT", "15")))      # (the session's default, as tests/conftest.py restores it)
    close(keys_f, keys2, 1e-6)
    close(out_f, out2, 2e-6)
    close(attn_f, attn2, 2e-6)
    close(keys, keys_f, 5e-6)
    close(out, out_f, 1e-5)
    close(attn, attn_f, 1e-5)
    # round 5: the projection on operands split once per tensor (x as a P3 tensor, W fragment-major; csrc/p3_loop.h) -- the form the
    # step runs: as close to fp64 as the other two, within 5e-6 / 1e-5 of the in-kernel split form
    out_p, attn_p, pooled_p, keys_p, _ = ops.umca_fwd(xg, Wg, bg, qg, nq, x_samples=B, q_shared=shared_q, x_drop=xdrop if mask else None,
                                                      out_drop=odrop, V=V, planes=True)
    close(keys_p, keys_ref, 2e-5)
    close(attn_p, a_ref, 2e-5)
    close(pooled_p, pooled_ref, 2e-5)
    close(out_p, out_ref, 2e-5)
    close(keys_p, keys, 5e-6)
    close(out_p, out, 1e-5)
    # inference form: no keys tensor at all, same outputs
    out3, attn3, pooled3, none, _ = ops.umca_fwd(xg, Wg, bg, qg, nq, x_samples=B, q_shared=shared_q, x_drop=xdrop if mask else None,
                                                 out_drop=odrop, want_keys=False, V=V)
    assert none is None and torch.equal(out3, out) and torch.equal(attn3, attn) and torch.equal(pooled3, pooled)
    # K3's keys feed the ordinary backward
    dout = dev(torch.randn(V, nq, Dm, generator=g))
    dz, dxd, dq = ops.attnpool_bwd(desc, dout, ())
    assert bool(torch.isfinite(dz).all()) and bool(torch.isfinite(dxd).all()) and bool(torch.isfinite(dq).all())
    del bits


def test_umca_k3_with_key_padding_lengths(ops):
    """K3 with the key-padding extension (per-sample valid frame counts): identical to the two-kernel path with the same
    lengths, and frames beyond a sample's length get weight exactly 0."""
    from sdumc_amd._lib import make_dropout
    g = torch.Generator().manual_seed(11)
    B, S, Dm, nq, T = 3, 2, 256, 7, 150
    V = B * S
    x = dev(torch.randn(B, T, Dm, generator=g))
    W, b = dev(torch.randn(Dm, Dm, generator=g) / 16), dev(torch.randn(Dm, generator=g) * 0.1)
    q = dev(torch.randn(V, nq, Dm, generator=g) / 4)
    lengths = dev(torch.tensor([T, 1, 64, 65, 128, 17], dtype=torch.int32))
    xdrop = make_dropout(True, 23, 0.5, T, Dm, B, call0=2, seed=5)
    bits = ops.dropout_bits(xdrop, S)
    odrop = make_dropout(True, 24, 0.5, nq, Dm, B, call0=2, seed=5)
    out, attn, pooled, keys, _ = ops.umca_fwd(x, W, b, q, nq, x_samples=B, x_drop=xdrop, out_drop=odrop, lengths=lengths, V=V)
    keys2 = ops.gemm(ops.NT, x, W, V * T, Dm, Dm, bias=b, act=ops.ACT_TANH, a_row_mod=B * T, a_drop=xdrop).view(V, T, Dm)
    out2, attn2, pooled2, _ = ops.attnpool_fwd(x, keys2, q, nq, x_samples=B, x_drop=xdrop, out_drop=odrop, lengths=lengths, tickets=False)
    close(out, out2, 2e-6)
    close(attn, attn2, 2e-6)
    close(pooled, pooled2, 2e-6)
    for v, n in enumerate(lengths.tolist()):
        assert float(attn[v, n:].abs().max()) == 0.0 if n < T else True
        np.testing.assert_allclose(attn[v, :n].sum(0).cpu().numpy(), 1.0, rtol=1e-5)
    del bits


def test_attnpool_multi_equals_single_calls(ops):
    """sdumc_attnpool_fwd_multi / _bwd_multi (the step's three Cross_Attention blocks in one launch pair) against one call per
    site: bit-identical, ragged T, keep-bits masks, key-padding lengths on one site."""
    from sdumc_amd._lib import make_dropout
    g = torch.Generator().manual_seed(5)
    B, S, Dm, nq = 3, 2, 256, 7
    V = B * S
    sites, singles, douts = [], [], []
    for i, T in enumerate((130, 64, 37)):
        x = dev(torch.randn(B, T, Dm, generator=g))
        keys = dev(torch.tanh(torch.randn(V, T, Dm, generator=g)))
        q = dev(torch.randn(V, nq, Dm, generator=g) / 4)
        xdrop = make_dropout(True, 23 + 2 * i, 0.5, T, Dm, B, call0=4, seed=9)
        bits = ops.dropout_bits(xdrop, S)
        odrop = make_dropout(True, 24 + 2 * i, 0.5, nq, Dm, B, call0=4, seed=9)
        lengths = dev(torch.tensor([T, T - 3, 5, 1, T, 7], dtype=torch.int32)) if i == 1 else None
        kw = dict(x=x, keys=keys, q=q, nq=nq, x_samples=B, x_drop=xdrop, out_drop=odrop, lengths=lengths, _bits=bits)
        sites.append(kw)
        # the single calls take the two-launch path (tickets=False), the grouped call the fused second passes (tickets)
        singles.append(ops.attnpool_fwd(x, keys, q, nq, x_samples=B, x_drop=xdrop, out_drop=odrop, lengths=lengths, tickets=False))
        douts.append(dev(torch.randn(V, nq, Dm, generator=g)))
    multi = ops.attnpool_fwd_multi(sites)
    for (o1, a1, p1, _), (o2, a2, p2, _) in zip(singles, multi):
        assert torch.equal(o1, o2) and torch.equal(a1, a2) and torch.equal(p1, p2)
    back1 = [ops.attnpool_bwd(s[3], d, ()) for s, d in zip(singles, douts)]
    back2 = ops.attnpool_bwd_multi([m[3] for m in multi], douts)
    for r1, r2 in zip(back1, back2):
        for t1, t2 in zip(r1, r2):
            assert torch.equal(t1, t2)
    for m in multi:                                   # the counters re-armed themselves
        assert int(m[3]._keep_tickets.abs().sum()) == 0


@pytest.mark.parametrize("nq,T,shared_x,bf16", [(7, 130, True, False), (1, 375, True, False), (7, 32, False, False), (1, 64, True, False),
                                                (7, 225, True, True), (1, 37, False, True)])
def test_attnpool_v2_kernels_equal_the_round3_kernels_and_fp64(ops, nq, T, shared_x, bf16):
    """The round-4 pooling kernels (every global load issued up front, softmax factors through v_readlane, XCD-aware pairing of
    the two streams' workgroups: what the engine launches -- keep-bits masks, two-pass combine) against the round-3 kernels on
    the same descriptors: bit-identical forward (out, weights, pooled rows) and backward (dz, dxd, dq), fp32 and bf16 frames,
    nq = 1 (FRA2UTT_new, model :56-68) and 7 (Cross_Attention, model :79-95), ragged last chunks, x shared by the two streams
    or not, key-padding lengths; the fp32 forward also against the fp64 restatement; the shared-query gradient from the
    backward's single reduce launch (dq_sum) against the per-sample dq summed."""
    from sdumc_amd import _lib
    from sdumc_amd._lib import make_dropout
    g = torch.Generator().manual_seed(nq * 1000 + T)
    B, S, Dm = 5, 2, 256
    V = B * S
    xs = B if shared_x else V
    x = torch.randn(xs, T, Dm, generator=g)
    keys = torch.tanh(torch.randn(V, T, Dm, generator=g))
    q = torch.randn(V, nq, Dm, generator=g) / 4
    lengths = dev(torch.randint(1, T + 1, (V,), generator=g).to(torch.int32))
    xdrop = make_dropout(True, 23, 0.5, T, Dm, B, call0=4, seed=9)
    odrop = make_dropout(True, 24, 0.5, nq, Dm, B, call0=4, seed=9)
    xg, kg, qg = dev(x), dev(keys), dev(q)
    if bf16:      # bf16 storage: the masked frames exist per virtual sample, the pooling kernels read no mask
        m = ops.dropout_mask(xdrop, S).view(V, T, Dm)
        xg = ((xg if not shared_x else xg.repeat(S, 1, 1)) * m).to(torch.bfloat16).contiguous()
        kg = kg.to(torch.bfloat16)
        xs, xdrop = V, None
    else:
        bits = ops.dropout_bits(xdrop, S)
    dout = dev(torch.randn(V, nq, Dm, generator=g))
    res = []
    try:
        for v2 in (0, 1):
            _lib.lib.sdumc_attnpool_set_v2_(v2)
            for lens in (None, lengths):
                attn, pooled, out = [torch.empty(s_, device="cuda") for s_ in ((V, T, nq), (V, nq, Dm), (V, nq, Dm))]
                a = ops.attnpool_desc(xg, kg, qg, V, T, nq, xs, nq * Dm, xdrop, odrop, attn, pooled, out, lengths=lens, tickets=False)
                a.bf16 = 1 if bf16 else 0
                need = _lib.lib.sdumc_attnpool_fwd_workspace_bytes(V, T, nq)
                ws = torch.empty(need, dtype=torch.uint8, device="cuda")
                a.workspace, a.workspace_bytes = _lib.ptr(ws), need
                import ctypes as C
                _lib.check(_lib.lib.sdumc_attnpool_fwd(C.byref(a), _lib.current_stream()), "sdumc_attnpool_fwd")
                dz, dxd, dq = ops.attnpool_bwd(a, dout, ())
                torch.cuda.synchronize()
                res.append((out.clone(), attn.clone(), pooled.clone(), dz, dxd, dq))
    finally:
        _lib.lib.sdumc_attnpool_set_v2_(1)
    for old, new in zip(res[:2], res[2:]):
        for t_old, t_new in zip(old, new):
            assert torch.equal(t_old, t_new)
    assert not torch.equal(res[2][0], res[3][0])          # (the key-padding lengths do change the result)
    if not bf16:
        from oracle import philox
        xm = torch.from_numpy(np.concatenate([philox.dropout_mask(B, T, Dm, 0.5, 9, 4 + s_, 23) for s_ in range(S)])).double()
        xd = (x if not shared_x else x.repeat(S, 1, 1)).double() * xm
        sc = torch.einsum("vtd,vqd->vtq", keys.double(), q.double()) * 0.3
        att = torch.softmax(sc, dim=1)
        close(res[2][1], att, 2e-5)
        close(res[2][2], torch.einsum("vtq,vtd->vqd", att, xd), 2e-5)
    if nq == 1:       # a shared query's gradient from the one-launch reduce == the per-sample gradients summed
        attn, pooled, out = [torch.empty(s_, device="cuda") for s_ in ((V, T, nq), (V, nq, Dm), (V, nq, Dm))]
        q1 = qg[:1].contiguous()
        a = ops.attnpool_desc(xg, kg, q1, V, T, nq, xs, 0, xdrop, odrop, attn, pooled, out, tickets=False)
        a.bf16 = 1 if bf16 else 0
        need = _lib.lib.sdumc_attnpool_fwd_workspace_bytes(V, T, nq)
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        a.workspace, a.workspace_bytes = _lib.ptr(ws), need
        import ctypes as C
        _lib.check(_lib.lib.sdumc_attnpool_fwd(C.byref(a), _lib.current_stream()), "sdumc_attnpool_fwd")
        _, _, dq_each = ops.attnpool_bwd(a, dout, ())
        dz2, dxd2, dq_sum = ops.attnpool_bwd(a, dout, (), shared_q_sum=True)
        close(dq_sum, dq_each.double().sum(0, keepdim=True), 1e-5)
        again = ops.attnpool_bwd(a, dout, (), shared_q_sum=True)[2]
        assert torch.equal(dq_sum, again)


def test_losses_against_reference_goldens(ops, golden):
    g = golden("losses")
    T = lambda k: dev(torch.from_numpy(g[k]))
    loss, dp = ops.mse_fwd_bwd(T("mse_pred"), T("mse_tgt"))
    close(loss, g["mse"].reshape(1), 1e-6)
    close(dp, g["mse_dpred"], 1e-6)
    for tag in ("2d", "3d"):
        loss, da, db = ops.rmse_fwd_bwd(T(f"rmse{tag}_a"), T(f"rmse{tag}_b"))
        close(loss, g[f"rmse{tag}"].reshape(1), 1e-6)
        close(da, g[f"rmse{tag}_da"], 1e-5)
        close(db, g[f"rmse{tag}_db"], 1e-5)
    for tag in ("rnc", "rnctie"):
        f = torch.from_numpy(g[f"{tag}_f"])
        feats = dev(torch.cat([f[:, 0], f[:, 1]], dim=0))
        y2 = dev(torch.from_numpy(g[f"{tag}_y"]).repeat(2, 1).reshape(-1))
        loss, df, ws = ops.rnc_fwd_bwd(feats, y2)
        close(loss, g[tag].reshape(1), 1e-5)
        B = f.shape[0]
        want = np.concatenate([g[f"{tag}_df"][:, 0], g[f"{tag}_df"][:, 1]], axis=0)
        close(df, want, 1e-4)
        np.testing.assert_array_equal(ops.rnc_mask(y2).cpu().numpy(), g[f"{tag}_mask"])   # bit-exact membership
        # a data-parallel rank owns two row ranges of the gathered matrix
        part = ops.rnc_dfeat_rows(feats, ws, B, B)
        close(part, want[B:], 1e-4)


def test_rnc_large_matches_oracle(ops):
    from oracle import sdumc_oracle as O
    g = torch.Generator().manual_seed(2)
    B = 48
    f = torch.randn(B, 2, 64, generator=g).requires_grad_()
    y = (torch.rand(B, 1, generator=g) * 6 - 3).round(decimals=1)   # many ties
    l = O.rnc_loss(f, y)
    l.backward()
    feats = dev(torch.cat([f[:, 0], f[:, 1]], dim=0).detach())
    loss, df, _ = ops.rnc_fwd_bwd(feats, dev(y.repeat(2, 1).reshape(-1)), weight=0.8)
    close(loss, l.detach().reshape(1), 1e-5)
    close(df, 0.8 * torch.cat([f.grad[:, 0], f.grad[:, 1]]), 1e-4)
    np.testing.assert_array_equal(ops.rnc_mask(dev(y.repeat(2, 1).reshape(-1))).cpu().numpy(),
                                  O.rnc_masks(y.repeat(2, 1)).numpy())


@pytest.mark.parametrize("B,decimals", [(150, 1), (384, 0), (512, 3)])
def test_rnc_sorted_formulation_matches_oracle(ops, B, decimals):
    """n = 2B > 256 rows takes the O(n^2 log n) sorted formulation (what every rank of a data-parallel job evaluates:
    B_global = 512 at BASELINE configs[3]); loss and gradient against the oracle's literal restatement of the
    reference, with many tied labels (decimals 0: only 7 distinct values) and with almost none."""
    from oracle import sdumc_oracle as O
    g = torch.Generator().manual_seed(B + decimals)
    f = torch.randn(B, 2, 64, generator=g).double().requires_grad_()
    y = (torch.rand(B, 1, generator=g) * 6 - 3).round(decimals=decimals)
    l = O.rnc_loss(f, y.double())
    l.backward()
    feats = dev(torch.cat([f[:, 0], f[:, 1]], dim=0).detach().float())
    loss, df, _ = ops.rnc_fwd_bwd(feats, dev(y.repeat(2, 1).reshape(-1)), weight=0.8)
    close(loss, l.detach().reshape(1), 2e-5)
    ref = 0.8 * torch.cat([f.grad[:, 0], f.grad[:, 1]])
    scale = float(ref.abs().max())
    assert float((df.cpu().double() - ref).abs().max()) <= 1e-3 * scale


def test_adam_matches_torch(ops):
    g = torch.Generator().manual_seed(4)
    n = 10007
    p0 = torch.randn(n, generator=g)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=1e-4, weight_decay=1e-5)
    pd, m, v = dev(p0.clone()), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    hyper = torch.tensor([1e-4, 0, 0, 0]).cuda()
    for step in range(4):
        grad = torch.randn(n, generator=g)
        p.grad = grad.clone()
        opt.step()
        ops.adam_step(pd, dev(grad), m, v, hyper)
        close((pd.cpu() - p0) * 1e4, (p.detach() - p0) * 1e4, 1e-4, f"step {step}")
    assert float(hyper[1]) == 4.0
