"""Batches read IN PLACE from a resident feature store (include/sdumc_hip.h: sdumc_net_io.row_map, sdumc_gemm_p3.a_map,
sdumc_gemm_b1.a_map, sdumc_gg_problem.b_map; sdumc_gather_batch's map_out).  The two products that read features -- the frame
projections frame_dim_reshape_m (model :282-284) and their weight gradients (autograd of the same Linear) -- must give, through a row
map into the packed store, BIT FOR BIT what they give on the padded copy of the batch that the reference's collater builds
(toolkit/utils/read_data.py:223-248, toolkit/data/feat_data.py:232-253); the whole-step / whole-epoch form of the same statement is
tests/test_gpu_configs.py::test_run_epoch_*."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import _lib, ops
    return _lib, ops


def _store(n_utt, T, d, seed):
    """packed [sum T + 1, d] with a trailing zero row, per-utterance starts / lengths (ragged)"""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(max(1, T // 4), T + 1, (n_utt,), generator=g)
    rows = int(lens.sum())
    X = torch.zeros(rows + 1, d)
    X[:rows] = torch.randn(rows, d, generator=g)
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)[:-1]])
    return X.cuda(), starts, lens, rows


def _batch_map(starts, lens, zero_row, B, seed):
    g = torch.Generator().manual_seed(seed)
    idx = torch.randperm(len(lens), generator=g)[:B]
    Tp = int(lens[idx].max())
    m = torch.full((B, Tp), zero_row, dtype=torch.int32)
    for b, e in enumerate(idx.tolist()):
        n = int(lens[e])
        m[b, :n] = torch.arange(int(starts[e]), int(starts[e]) + n, dtype=torch.int32)
    return m.reshape(-1), idx, Tp


def _padded(x):      # int32 map -> device, with the 64 spare entries the bf16 weight-gradient kernel may read past a batch's last row
    return torch.cat([x, torch.zeros(64, dtype=torch.int32)]).cuda()


def test_gather_batch_writes_the_row_maps_the_host_would(env):
    """sdumc_gather_batch(map_out): entry (b, t) = start of utterance idx[b] + t for a valid frame, the store's zero row for padding;
    labels and valid frame counts ride along; a padded copy through the map equals the gathered padded copy."""
    _lib, ops = env
    from sdumc_amd.data import DeviceFeatureStore
    dims, Tcap = (64, 128, 64, 128), (23, 7, 15, 6)
    store = DeviceFeatureStore.synthetic(30, Tcap, dims, seed=3, planes=True)
    idx = torch.tensor([4, 29, 0, 17, 11], dtype=torch.int64)
    B, T = store.batch_shape(idx)
    maps = [torch.full((B * T[i] + 64,), -7, dtype=torch.int32, device="cuda") for i in range(4)]
    labels = torch.empty(B, device="cuda")
    lens = [torch.empty(B, dtype=torch.int32, device="cuda") for _ in range(4)]
    idx_d = idx.cuda()
    g = store.gather_desc(idx_d.data_ptr(), B, T, None, labels, lens, maps_out=maps)
    _lib.check(_lib.lib.sdumc_gather_batch(C.byref(g), 0, _lib.current_stream()), "sdumc_gather_batch")
    bd, pads, emos, vals, names = store.batch(idx)
    assert torch.equal(labels, vals)
    for i, (m, key) in enumerate(zip(store.MODS, ("audios", "texts", "videos", "feat4s"))):
        zr = store.packed[m].shape[0] - 1
        want = torch.full((B, T[i]), zr, dtype=torch.int32)
        for b, e in enumerate(idx.tolist()):
            n = int(store.length[m][e])
            want[b, :n] = torch.arange(int(store.start[m][e]), int(store.start[m][e]) + n, dtype=torch.int32)
        assert torch.equal(maps[i][:B * T[i]].cpu(), want.reshape(-1))
        assert torch.equal(maps[i][B * T[i]:].cpu(), torch.full((64,), -7, dtype=torch.int32)), "wrote past the batch's rows"
        assert torch.equal(lens[i].cpu(), store.length[m][idx].clamp(max=T[i]))
        assert float(store.packed[m][zr].abs().max()) == 0.0 and int(store.packed_p3[m][zr].max()) == 0      # the zero rows
        assert torch.equal(store.packed[m][maps[i][:B * T[i]].long()].view(B, T[i], -1), bd[key])          # map == the collater's padding


@pytest.mark.parametrize("shape", [(33, 128, 1), (70, 256, 1), (9, 1024, 4)])
def test_frame_projection_through_a_row_map_equals_the_padded_copy(env, shape):
    """sdumc_gemm_p3_nt / sdumc_gemm_b1_nt with a_map (descriptor form via a_map_rows, and the 64-bit form), one tensor and two (the
    text slot's two streams: A2 / a2_map), split-K included: bit for bit the product on the padded copy."""
    _lib, ops = env
    T, d, sk = shape
    B = 6
    X, starts, lens, rows = _store(20, T, d, 5)
    X2, starts2, lens2, rows2 = _store(20, T, d, 6)
    g = torch.Generator(device="cuda").manual_seed(9)
    W = torch.randn(256, d, device="cuda", generator=g) / d ** 0.5
    bias = torch.randn(256, device="cuda", generator=g)
    amap, _, Tp = _batch_map(starts, lens, rows, B, 1)
    M = B * Tp
    Xb = X[amap.cuda().long()].contiguous()
    X3, Xb3, W3 = ops.p3_split(X), ops.p3_split(Xb), ops.p3_split_frag(W)
    amap_d = _padded(amap)
    want = ops.gemm_p3_nt(Xb3, W3, M, 256, d, bias=bias, splitk=sk)
    for rows_hint in (X.shape[0], 0):
        got = ops.gemm_p3_nt(X3, W3, M, 256, d, bias=bias, splitk=sk, a_map=amap_d, a_map_rows=rows_hint)
        assert torch.equal(got, want), f"gemm_p3 mapped (a_map_rows={rows_hint})"
    # two tensors in one launch: rows [0, M2) from X through its map, rows [M2, 2 M2) from X2 through its own (M2 a multiple of 64)
    B2 = 64
    m1 = torch.randint(0, rows + 1, (B2 * 2,), dtype=torch.int32)
    m2 = torch.randint(0, rows2 + 1, (B2 * 2,), dtype=torch.int32)
    Xc = torch.cat([X[m1.cuda().long()], X2[m2.cuda().long()]]).contiguous()
    want2 = ops.gemm_p3_nt(ops.p3_split(Xc), W3, 4 * B2, 256, d, bias=bias, splitk=sk)
    got2 = ops.gemm_p3_nt(X3, W3, 4 * B2, 256, d, bias=bias, splitk=sk, A3_second=ops.p3_split(X2), second_row0=2 * B2,
                          a_map=_padded(m1), a2_map=_padded(m2), a_map_rows=max(X.shape[0], X2.shape[0]))
    assert torch.equal(got2, want2), "gemm_p3 mapped, two tensors"
    # bf16 storage
    if d % 128 == 0:
        Xh, Xbh, Wb = X.bfloat16(), Xb.bfloat16(), ops.b1_frag(W)
        wanth = ops.gemm_b1_nt(Xbh, Wb, M, 256, d, bias=bias, splitk=sk)
        for rows_hint in (X.shape[0], 0):
            goth = ops.gemm_b1_nt(Xh, Wb, M, 256, d, bias=bias, splitk=sk, a_map=amap_d, a_map_rows=rows_hint)
            assert torch.equal(goth, wanth), f"gemm_b1 mapped (a_map_rows={rows_hint})"


@pytest.mark.parametrize("shape", [(33, 128), (70, 256), (21, 1024)])
def test_frame_weight_gradient_through_a_row_map_equals_the_padded_copy(env, shape):
    """sdumc_gemm_group_tn / _bf16 with b_map: C = dx^T . features with the features' k-rows fetched through the map (K not a multiple
    of the k-tile, two K segments = the two text streams from two packed tensors): weight AND bias gradient bit for bit."""
    _lib, ops = env
    T, d = shape
    B = 5
    X, starts, lens, rows = _store(20, T, d, 7)
    X2, starts2, lens2, rows2 = _store(20, T, d, 8)
    g = torch.Generator(device="cuda").manual_seed(3)
    amap, _, Tp = _batch_map(starts, lens, rows, B, 2)
    amap2, _, Tp2 = _batch_map(starts2, lens2, rows2, B, 3)
    K, K2 = B * Tp, B * Tp2
    Xb, Xb2 = X[amap.cuda().long()].contiguous(), X2[amap2.cuda().long()].contiguous()
    dx, dx2 = torch.randn(K, 256, device="cuda", generator=g), torch.randn(K2, 256, device="cuda", generator=g)
    for hf in (False, True):
        if hf and d % 8:
            continue
        cv = (lambda t: t.bfloat16()) if hf else (lambda t: t)
        p0 = [dict(A=cv(dx), B=cv(Xb), colsum=torch.empty(256, device="cuda")),
              dict(A=cv(dx), B=cv(Xb), A1=cv(dx2), B1=cv(Xb2), colsum=torch.empty(256, device="cuda"))]
        p1 = [dict(A=cv(dx), B=cv(X), b_map=_padded(amap), K=K, colsum=torch.empty(256, device="cuda")),
              dict(A=cv(dx), B=cv(X), b_map=_padded(amap), K=K, A1=cv(dx2), B1=cv(X2), b_map1=_padded(amap2), K1=K2,
                   colsum=torch.empty(256, device="cuda"))]
        ops.gemm_group_tn(p0)
        ops.gemm_group_tn(p1)
        torch.cuda.synchronize()
        for a, b in zip(p0, p1):
            assert torch.equal(a["C"], b["C"]) and torch.equal(a["colsum"], b["colsum"]), f"grouped dW mapped (bf16={hf})"


def test_row_map_argument_contract(env):
    """SDUMC_EINVAL where a map cannot be honoured: with a_row_mod / keep-bits / an explicit tile height on the plane GEMM, a2_map
    without A2, b_row_mod beside b_map, and on the fp32-MFMA form of the grouped kernel (sdumc_set_split_(0))."""
    _lib, ops = env
    lib = _lib.lib
    X = torch.randn(129, 128, device="cuda")
    X[-1] = 0
    W = torch.randn(256, 128, device="cuda")
    X3, W3 = ops.p3_split(X), ops.p3_split_frag(W)
    amap = _padded(torch.randint(0, 129, (64,), dtype=torch.int32))
    bits = torch.randint(0, 16, (64, 32), device="cuda", dtype=torch.uint8)
    for kw in (dict(a_row_mod=32), dict(bits=bits, scale=2.0), dict(tile_m=64), dict(a2_map=amap)):
        with pytest.raises(_lib.SdumcError):
            ops.gemm_p3_nt(X3, W3, 64, 256, 128, a_map=amap, **kw)
    dx = torch.randn(64, 256, device="cuda")
    with pytest.raises(_lib.SdumcError):
        ops.gemm_group_tn([dict(A=dx, B=X, b_map=amap, K=64, b_row_mod=32)])
    prev = lib.sdumc_get_split_()
    try:
        lib.sdumc_set_split_(0)
        with pytest.raises(_lib.SdumcError):
            ops.gemm_group_tn([dict(A=dx, B=X, b_map=amap, K=64)])
    finally:
        lib.sdumc_set_split_(prev)
