"""GPU parity tests, network level: forward / backward / full train step through the
C ABI (sdumc_net_forward, sdumc_net_backward, sdumc_train_step) against
  (a) golden vectors generated from the REAL reference (tests/golden/*.npz), and
  (b) the CPU oracle on the same seeded inputs (incl. ragged/zero-padded batches,
      unequal text/feat4 lengths, MOSEI-sized shapes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ("vals", "fused", "rnc", "text_hidden", "cross_text")


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import engine
    return engine


def close(got, want, tol=1e-4, msg=""):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy() if isinstance(want, torch.Tensor) else np.asarray(want, dtype=np.float64)
    scale = max(1.0, np.abs(want).max())
    np.testing.assert_allclose(got, want.reshape(got.shape), rtol=tol, atol=tol * scale, err_msg=msg)


def flat_from(E, P, dims):
    lay = E.ParamLayout.get(*dims[:3])
    flat = torch.zeros(lay.total)
    for k, v in lay.views(flat).items():
        v.copy_(P[k])
    return flat.cuda(), lay


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_param_layout_matches_reference_state_dict(E):
    from oracle import sdumc_oracle as O
    dims = (1024, 4096, 1024)
    lay = E.ParamLayout.get(*dims)
    shapes = O.param_shapes(dims + (4096,))
    assert set(lay.entries) == set(shapes)
    for k, shp in shapes.items():
        off, s, live = lay.entries[k]
        assert tuple(s) == tuple(shp) and off % 4 == 0 and live == (not O.is_dead(k)), k
    assert sum(int(np.prod(s)) for s in shapes.values()) == 4268884
    assert lay.live >= 3857291 and all(lay.entries[k][0] < lay.live for k in lay.live_names())


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("streams", [1, 2])
def test_forward_vs_reference_golden(E, golden, mode, streams):
    from oracle import sdumc_oracle as O
    g = golden("forward")
    dims = tuple(int(v) for v in g["dims"])
    flat, _ = flat_from(E, O.init_params(dims, seed=int(g["pseed"])), dims)
    audio, video = T(g["audio"]).cuda(), T(g["video"]).cuda()
    texts = [T(g["text"]).cuda(), T(g["feat4"]).cuda()]
    seed, step = int(g["seed"]), int(g["step"])
    B = audio.shape[0]
    if streams == 2:   # text T=5, feat4 T=6: exercises the unequal-length (two-run) path
        rng = E.RngState(seed, audio.device, call=2 * step)
        outs = E.NetCall(flat, audio, texts, video, mode == "train", rng).forward()
        for s in range(2):
            for n, t in zip(NAMES, outs):
                close(t[s * B:(s + 1) * B], g[f"{mode}{s}_{n}"], 2e-5, f"{n} stream {s}")
    else:
        for s in range(2):
            rng = E.RngState(seed, audio.device, call=2 * step + s)
            outs = E.NetCall(flat, audio, [texts[s]], video, mode == "train", rng).forward()
            for n, t in zip(NAMES, outs):
                close(t, g[f"{mode}{s}_{n}"], 2e-5, f"{n} stream {s}")


def _oracle_grads(P, audio, texts, video, mode, seed, call0, douts):
    from oracle import sdumc_oracle as O
    leaves = {k: v.double().requires_grad_(not O.is_dead(k)) for k, v in P.items()}
    total = 0
    for s, tx in enumerate(texts):
        d = O.DropCtx("eval" if mode == "eval" else "philox", seed, call0 + s)
        y, (z, r, th, ct) = O.forward(leaves, audio.double(), tx.double(), video.double(), d)
        for o, do in zip((y, z, r, th, ct), douts):
            B = y.shape[0]
            total = total + (o * do[s * B:(s + 1) * B].double().reshape(o.shape)).sum()
    total.backward()
    return {k: v.grad for k, v in leaves.items() if v.grad is not None}


@pytest.mark.parametrize("equal_T", [True, False])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_backward_vs_oracle(E, mode, equal_T):
    from oracle import sdumc_oracle as O
    dims = (48, 32, 40, 32)
    B, Tn = 5, (70, 9, 33, 9 if equal_T else 4)      # T_a > 64: more than one row chunk in the pooling kernels
    P = O.init_params(dims, seed=11)
    flat, lay = flat_from(E, P, dims)
    audio, text, video, feat4, _ = O.synthetic_batch(B, Tn, dims, seed=5)
    audio[1, 40:] = 0      # zero-padded (ragged) samples take part in the softmax like in the reference
    video[3, 10:] = 0
    seed, call0 = 99, 4
    g = torch.Generator().manual_seed(1)
    V = 2 * B
    douts = [torch.randn(V, 1, generator=g), torch.randn(V, 128, generator=g), torch.randn(V, 64, generator=g),
             torch.randn(V, 256, generator=g), torch.randn(V, 7, 128, generator=g)]
    want = _oracle_grads(P, audio, [text, feat4], video, mode, seed, call0, douts)
    rng = E.RngState(seed, "cuda", call=call0)
    call = E.NetCall(flat, audio.cuda(), [text.cuda(), feat4.cuda()], video.cuda(), mode == "train", rng)
    call.forward()
    grads = call.backward(*[d.cuda().contiguous() for d in douts])
    gv = lay.views(torch.cat([grads.cpu(), torch.zeros(lay.total - lay.live)]))
    assert set(want) == set(lay.live_names())
    for k in lay.live_names():
        close(gv[k], want[k], 2e-4, k)
    # bitwise reproducible (no float atomics anywhere)
    call.forward()
    again = call.backward(*[d.cuda().contiguous() for d in douts])
    assert torch.equal(grads, again)


@pytest.mark.parametrize("B,Tn,dims", [
    (1, (1, 1, 1, 1), (4, 8, 12, 8)),          # one utterance, one frame per modality, minimal widths
    (2, (3, 1, 2, 5), (7, 9, 5, 9)),            # widths that are not multiples of 4 (no 16-byte rows anywhere)
    (3, (65, 1, 64, 2), (20, 36, 28, 36)),      # T straddling the 64-row pooling chunk; one text frame
    (1, (512, 300, 257, 1), (16, 16, 16, 16)),  # long ragged sequences for a single utterance
])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_edge_shapes_forward_backward_vs_oracle(E, mode, B, Tn, dims):
    """Smallest and most awkward shapes the boundary admits: B = 1 (RnC over n = 2 rows), single frames (softmax over one
    element), odd feature widths, sequences that straddle the kernels' chunk sizes, an all-zero (fully padded) utterance."""
    from oracle import sdumc_oracle as O
    P = O.init_params(dims, seed=21)
    flat, lay = flat_from(E, P, dims)
    audio, text, video, feat4, _ = O.synthetic_batch(B, Tn, dims, seed=6)
    if B > 1:
        video[B - 1] = 0                      # a fully padded utterance: uniform attention over zero frames
    seed, call0 = 7, 2
    g = torch.Generator().manual_seed(2)
    V = 2 * B
    douts = [torch.randn(V, 1, generator=g), torch.randn(V, 128, generator=g), torch.randn(V, 64, generator=g),
             torch.randn(V, 256, generator=g), torch.randn(V, 7, 128, generator=g)]
    rng = E.RngState(seed, "cuda", call=call0)
    call = E.NetCall(flat, audio.cuda(), [text.cuda(), feat4.cuda()], video.cuda(), mode == "train", rng)
    outs = [o.clone() for o in call.forward()]
    for s, tx in enumerate((text, feat4)):
        d = O.DropCtx("eval" if mode == "eval" else "philox", seed, call0 + s)
        y, (z, r, th, ct) = O.forward({k: v.double() for k, v in P.items()}, audio.double(), tx.double(), video.double(), d)
        for name, got, want in zip(NAMES, outs, (y, z, r, th, ct)):
            close(got[s * B:(s + 1) * B], want, 1e-4, f"{name} stream {s}")
    want = _oracle_grads(P, audio, [text, feat4], video, mode, seed, call0, douts)
    grads = call.backward(*[d.cuda().contiguous() for d in douts])
    gv = lay.views(torch.cat([grads.cpu(), torch.zeros(lay.total - lay.live)]))
    for k in lay.live_names():
        close(gv[k], want[k], 2e-4, k)


def test_train_step_vs_reference_golden(E, golden):
    from oracle import sdumc_oracle as O
    from tests.golden.make_goldens import digest
    g = golden("step")
    dims = tuple(int(v) for v in g["dims"])
    P = O.init_params(dims, seed=int(g["pseed"]))
    flat, lay = flat_from(E, P, dims)
    before = flat.clone()
    Tn = tuple(int(v) for v in g["T"])
    B = g["audio"].shape[0]
    ts = E.TrainStep(flat, B, Tn, dims, weights=tuple(float(w) for w in g["weights"]), seed=int(g["seed"]))
    ts.rng.set_call(2 * int(g["step"]))
    ts.set_batch(T(g["audio"]).cuda(), T(g["text"]).cuda(), T(g["video"]).cuda(), T(g["feat4"]).cuda(),
                 T(g["vals"]).cuda())
    losses = ts.run().cpu().numpy()
    np.testing.assert_allclose(losses[0], float(g["loss"]), rtol=2e-5)
    np.testing.assert_allclose(losses[1:7], g["terms"], rtol=2e-5, atol=1e-6)
    close(ts.vals[:B], g["y0"], 2e-5)
    close(ts.vals[B:], g["y1"], 2e-5)
    names = [str(n) for n in g["names"]]
    dead = {str(n) for n in g["dead"]}
    gv = lay.views(torch.cat([ts.grads.cpu(), torch.zeros(lay.total - lay.live)]))
    pv_new, pv_old = lay.views(flat.cpu()), lay.views(before.cpu())
    for i, k in enumerate(names):
        if k in dead:
            assert torch.equal(pv_new[k], pv_old[k]), f"dead parameter {k} moved"
            continue
        got = digest(gv[k], k)
        scale = max(1e-6, abs(g["grad_digest"][i][1]))
        # + absolute floor: orgin_linear_change.*.bias gradients are analytically 0 (RnC is translation invariant)
        np.testing.assert_allclose(got, g["grad_digest"][i], rtol=5e-4, atol=5e-5 * scale + 1e-6, err_msg=k)
        if "grad__" + k in g.files:
            close(gv[k], g["grad__" + k], 2e-4, k)
        if "delta__" + k in g.files:
            # the first Adam step is lr*g/(|g|+eps): ill-conditioned where g is rounding noise (|g| ~ eps=1e-8)
            ok = np.abs(g["grad__" + k].reshape(pv_new[k].shape)) > 1e-5
            np.testing.assert_allclose(((pv_new[k] - pv_old[k]) * 1e4).numpy()[ok],
                                       g["delta__" + k].reshape(pv_new[k].shape)[ok], rtol=5e-3, atol=5e-3, err_msg=k)
    assert ts.rng.call == 2 * int(g["step"]) + 2        # the step consumed two Philox call indices


def test_graph_replay_equals_eager_and_advances_state(E):
    from oracle import sdumc_oracle as O
    dims = (64, 32, 64, 32)
    B, Tn = 8, (40, 6, 20, 6)
    P = O.init_params(dims, seed=2)
    batch = O.synthetic_batch(B, Tn, dims, seed=3)
    res = []
    from sdumc_amd import _lib
    try:
        _lib.lib.sdumc_set_chain_cluster(0)      # a capture takes chain.hip's kernels: compare like with like
        _lib.lib.sdumc_set_background_lane(0)    # ... and no early key-projection backward (which decides what rides in which
                                                 # grouped weight-gradient launch, i.e. where their K ranges are cut)
        for use_graph in (False, True):
            flat, lay = flat_from(E, P, dims)
            ts = E.TrainStep(flat, B, Tn, dims, seed=5)
            ts.set_batch(*[t.cuda() for t in batch])
            if use_graph:
                ts.capture()
            ls = []
            for _ in range(3):
                ls.append(ts.run().cpu().clone())
            torch.cuda.synchronize()
            res.append((flat.cpu().clone(), ls, ts.rng.call, float(ts.hyper[1])))
    finally:                                     # process-wide switches (tests/conftest.py restores them too)
        _lib.lib.sdumc_set_chain_cluster(1)
        _lib.lib.sdumc_set_background_lane(3)
    assert torch.equal(res[0][0], res[1][0]), "graph replay must equal eager launches bit for bit"
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    assert res[0][2] == res[1][2] == 6 and res[0][3] == res[1][3] == 3.0
    assert not torch.equal(res[0][1][0], res[0][1][1])     # fresh masks + updated weights each step


def test_mosei_shape_forward_and_step_vs_oracle(E):
    """BASELINE config C2 shapes at B = 8 (the plain, un-clustered schedule of a small batch; the stated batch of 64 runs against
    the oracle in tests/test_gpu_fullsize.py): full T and feature widths, train mode with Philox masks, one complete
    optimisation step."""
    from oracle import sdumc_oracle as O
    dims = (1024, 4096, 1024, 4096)
    B, Tn = 8, (375, 32, 225, 32)
    P = O.init_params(dims, seed=0)
    flat, lay = flat_from(E, P, dims)
    audio, text, video, feat4, vals = O.synthetic_batch(B, Tn, dims, seed=1234)
    ts = E.TrainStep(flat, B, Tn, dims, seed=777)
    ts.set_batch(audio.cuda(), text.cuda(), video.cuda(), feat4.cuda(), vals.cuda())
    losses = ts.run().cpu().numpy()
    torch.set_num_threads(max(1, torch.get_num_threads()))
    Pd = {k: v.clone() for k, v in P.items()}
    loss, terms, grads, outs = O.train_step(Pd, {}, audio, text, video, feat4, vals, mode="philox", seed=777, step=0)
    np.testing.assert_allclose(losses[0], float(loss), rtol=1e-3)          # north-star tolerance 1e-3
    np.testing.assert_allclose(losses[1:7], [float(t) for t in terms], rtol=1e-3, atol=1e-5)
    close(ts.vals[:B], outs[0][0], 1e-3)
    close(ts.cross_text[B:], outs[1][4], 1e-3)
    gv = lay.views(torch.cat([ts.grads.cpu(), torch.zeros(lay.total - lay.live)]))
    for k in lay.live_names():
        close(gv[k], grads[k], 1e-3, k)
    pv = lay.views(flat.cpu())
    for k in ("frame_dim_reshape_1.weight", "cross_att_fra2utt_0.input_proj.weight", "fc_out_v.weight"):
        close((pv[k] - P[k]) * 1e4, (Pd[k] - P[k]) * 1e4, 2e-2, k)


@pytest.mark.parametrize("B", [1, 2, 3])
def test_train_step_smallest_batches_vs_oracle(E, B):
    """The fused step at the smallest batches: B = 1 puts n = 2 rows into the RnC loss (one positive, no negative per anchor),
    B = 2 / 3 the first non-trivial masks; label ties included (B = 3: two equal labels)."""
    from oracle import sdumc_oracle as O
    dims, Tn = (24, 16, 20, 16), (9, 2, 5, 3)
    P = O.init_params(dims, seed=4)
    flat, lay = flat_from(E, P, dims)
    audio, text, video, feat4, vals = O.synthetic_batch(B, Tn, dims, seed=77)
    if B == 3:
        vals[2] = vals[0]
    ts = E.TrainStep(flat, B, Tn, dims, seed=5)
    ts.set_batch(audio.cuda(), text.cuda(), video.cuda(), feat4.cuda(), vals.cuda())
    losses = ts.run().cpu().numpy()
    Pd = {k: v.clone() for k, v in P.items()}
    loss, terms, grads, outs = O.train_step(Pd, {}, audio, text, video, feat4, vals, mode="philox", seed=5, step=0)
    want = np.array([float(t) for t in terms])
    assert np.array_equal(np.isfinite(losses[1:7]), np.isfinite(want))       # same NaN/inf pattern as the reference math
    fin = np.isfinite(want)
    np.testing.assert_allclose(losses[1:7][fin], want[fin], rtol=1e-4, atol=1e-6)
    if np.isfinite(float(loss)):
        np.testing.assert_allclose(losses[0], float(loss), rtol=1e-4)
        gv = lay.views(torch.cat([ts.grads.cpu(), torch.zeros(lay.total - lay.live)]))
        for k in lay.live_names():
            close(gv[k], grads[k], 5e-4, k)


def test_full_c2_batch_properties(E):
    """B=64 MOSEI shapes (BASELINE configs[1]): size-independent properties at full size."""
    from oracle import sdumc_oracle as O
    dims = (1024, 4096, 1024, 4096)
    B, Tn = 64, (375, 32, 225, 32)
    P = O.init_params(dims, seed=0)
    flat, lay = flat_from(E, P, dims)
    audio, text, video, feat4, vals = [t.cuda() for t in O.synthetic_batch(B, Tn, dims, seed=1234)]
    # eval forward: batch of 64 == two half batches (samples are independent), both streams
    full = E.NetCall(flat, audio, [text, feat4], video, False, None).forward()
    h = B // 2
    for lo in (0, h):
        part = E.NetCall(flat, audio[lo:lo + h].contiguous(), [text[lo:lo + h].contiguous(), feat4[lo:lo + h].contiguous()],
                         video[lo:lo + h].contiguous(), False, None).forward()
        for n, f, p in zip(NAMES, full, part):
            for s in range(2):
                close(p[s * h:(s + 1) * h], f[s * B + lo:s * B + lo + h], 1e-5, n)
    # train mode: a shard with sample0 = 32 draws the same masks as rows 32.. of the full batch
    rng = E.RngState(9, audio.device, call=0)
    full = [t.clone() for t in E.NetCall(flat, audio, [text, feat4], video, True, rng).forward()]
    part = E.NetCall(flat, audio[h:].contiguous(), [text[h:].contiguous(), feat4[h:].contiguous()],
                     video[h:].contiguous(), True, rng, sample0=h).forward()
    for n, f, p in zip(NAMES, full, part):
        for s in range(2):
            close(p[s * h:(s + 1) * h], f[s * B + h:s * B + B], 1e-5, n)
    # one full train step: finite loss, every live tensor got a finite non-zero gradient, dead params untouched
    before = flat.clone()
    ts = E.TrainStep(flat, B, Tn, dims, seed=1)
    ts.set_batch(audio, text, video, feat4, vals)
    losses = ts.run().cpu()
    assert torch.isfinite(losses).all() and losses[0] > 0
    gv = lay.views(torch.cat([ts.grads.cpu(), torch.zeros(lay.total - lay.live)]))
    for k in lay.live_names():
        assert torch.isfinite(gv[k]).all() and gv[k].abs().sum() > 0, k
    assert torch.equal(flat[lay.live:], before[lay.live:])
    assert not torch.equal(flat[:lay.live], before[:lay.live])


def test_c5_global_batch_on_one_gpu_properties(E):
    """BASELINE configs[4] at its GLOBAL batch (256 x T 512 x d 1024 = 2 x 131072 rows per modality, the largest
    configuration in the list) on one GPU: size-independent properties -- the batch equals its two halves (eval), a shard
    with sample0 reproduces the full batch's rows (train), the fused step is finite, deterministic bit for bit, and the RnC
    loss runs its sorted formulation at n = 512."""
    dims = (1024, 1024, 1024, 1024)
    B, Tn = 256, (512, 512, 512, 512)
    from oracle import sdumc_oracle as O
    P = O.init_params(dims, seed=2)
    flat, lay = flat_from(E, P, dims)
    g = torch.Generator(device="cuda").manual_seed(5)
    audio, text, video, feat4 = [torch.randn(B, Tn[i], dims[i], device="cuda", generator=g) for i in range(4)]
    vals = torch.rand(B, device="cuda", generator=g) * 6 - 3
    full = [t.clone() for t in E.NetCall(flat, audio, [text, feat4], video, False, None).forward()]
    h = B // 2
    for lo in (0, h):
        part = E.NetCall(flat, audio[lo:lo + h].contiguous(), [text[lo:lo + h].contiguous(), feat4[lo:lo + h].contiguous()],
                         video[lo:lo + h].contiguous(), False, None).forward()
        for n, f, p in zip(NAMES, full, part):
            for s in range(2):
                close(p[s * h:(s + 1) * h], f[s * B + lo:s * B + lo + h], 1e-5, n)
    rng = E.RngState(9, audio.device, call=0)
    full = [t.clone() for t in E.NetCall(flat, audio, [text, feat4], video, True, rng).forward()]
    part = E.NetCall(flat, audio[h:].contiguous(), [text[h:].contiguous(), feat4[h:].contiguous()],
                     video[h:].contiguous(), True, rng, sample0=h).forward()
    for n, f, p in zip(NAMES, full, part):
        for s in range(2):
            close(p[s * h:(s + 1) * h], f[s * B + h:s * B + B], 1e-5, n)
    del full, part
    results = []
    for _ in range(2):
        fl = flat.clone()
        ts = E.TrainStep(fl, B, Tn, dims, seed=3)
        ts.set_batch(audio, text, video, feat4, vals)
        losses = ts.run().clone()
        results.append((losses, ts.grads.clone(), fl))
        del ts
    assert torch.isfinite(results[0][0]).all() and float(results[0][0][0]) > 0
    for a, b in zip(results[0], results[1]):
        assert torch.equal(a, b)
    gv = lay.views(torch.cat([results[0][1].cpu(), torch.zeros(lay.total - lay.live)]))
    for k in lay.live_names():
        assert torch.isfinite(gv[k]).all() and gv[k].abs().sum() > 0, k


def test_long_sequence_c5_shapes_vs_oracle(E):
    """BASELINE configs[4] shape family: T_t = T_a = T_v = 512 (8 row chunks per sample in the pooling kernels),
    d = 1024 for every modality; one train step against the oracle at B = 4 (the per-GPU slice B = 32: tests/test_gpu_fullsize.py)."""
    from oracle import sdumc_oracle as O
    dims = (1024, 1024, 1024, 1024)
    B, Tn = 4, (512, 512, 512, 512)
    P = O.init_params(dims, seed=4)
    flat, lay = flat_from(E, P, dims)
    batch = O.synthetic_batch(B, Tn, dims, seed=21)
    ts = E.TrainStep(flat, B, Tn, dims, seed=11)
    ts.set_batch(*[t.cuda() for t in batch])
    losses = ts.run().cpu().numpy()
    loss, terms, grads, outs = O.train_step({k: v.clone() for k, v in P.items()}, {}, *batch, mode="philox", seed=11, step=0)
    np.testing.assert_allclose(losses[0], float(loss), rtol=1e-3)
    np.testing.assert_allclose(losses[1:7], [float(t) for t in terms], rtol=1e-3, atol=1e-5)
    gv = lay.views(torch.cat([ts.grads.cpu(), torch.zeros(lay.total - lay.live)]))
    for k in lay.live_names():
        close(gv[k], grads[k], 1e-3, k)


def test_large_batch_rnc_1024_rows(E):
    """B = 512 (n = 2B = 1024 rows in RnCLoss: what every rank evaluates in the exact 8-GPU mode), short
    sequences; forward + loss terms against the oracle, plus shard invariance of the whole step's loss."""
    from oracle import sdumc_oracle as O
    dims = (64, 32, 48, 32)
    B, Tn = 512, (6, 3, 5, 3)
    P = O.init_params(dims, seed=6)
    flat, lay = flat_from(E, P, dims)
    batch = O.synthetic_batch(B, Tn, dims, seed=22)
    batch = batch[:4] + ((batch[4] * 4).round() / 4,)          # many tied labels
    ts = E.TrainStep(flat, B, Tn, dims, seed=12)
    ts.set_batch(*[t.cuda() for t in batch])
    losses = ts.run().cpu().numpy()
    loss, terms, _ = O.step_loss(P, *batch, mode="philox", seed=12, step=0)
    np.testing.assert_allclose(losses[1:7], [float(t) for t in terms], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(losses[0], float(loss), rtol=2e-4)
    assert torch.isfinite(ts.grads).all()


def test_bf16_operand_mode_c3(E):
    """BASELINE configs[2] (C3: MOSEI shapes, text-missing stream + self-distillation, bf16): the frame-level forward
    projections round their operands to bf16 (fp32 accumulate); SURVEY §8d bar for that mode: activations <= 2e-2 of
    the fp32 oracle, loss in fp32.  B = 8 here; the stated batch of 64 against both oracles: tests/test_gpu_fullsize.py."""
    from oracle import sdumc_oracle as O
    dims = (1024, 4096, 1024, 4096)
    B, Tn = 8, (375, 32, 225, 32)
    P = O.init_params(dims, seed=0)
    flat, lay = flat_from(E, P, dims)
    audio, text, video, feat4, vals = O.synthetic_batch(B, Tn, dims, seed=1234)
    dev = [t.cuda() for t in (audio, text, video, feat4)]
    f32 = [t.clone() for t in E.NetCall(flat, dev[0], [dev[1], dev[3]], dev[2], False, None).forward()]
    b16 = [t.clone() for t in E.NetCall(flat, dev[0], [dev[1], dev[3]], dev[2], False, None, bf16=True).forward()]
    diffs = []
    for n, a, b in zip(NAMES, f32, b16):
        close(b, a, 2e-2, "bf16 " + n)
        diffs.append(float((a - b).abs().max()))
    assert max(diffs) > 1e-6, "bf16 mode produced bit-identical outputs: the mode is not active"
    # a complete bf16-mode optimisation step against the fp32 oracle
    ts = E.TrainStep(flat, B, Tn, dims, seed=777, bf16=True)
    ts.set_batch(dev[0], dev[1], dev[2], dev[3], vals.cuda())
    losses = ts.run().cpu().numpy()
    Pd = {k: v.clone() for k, v in P.items()}
    loss, terms, grads, outs = O.train_step(Pd, {}, audio, text, video, feat4, vals, mode="philox", seed=777, step=0)
    np.testing.assert_allclose(losses[0], float(loss), rtol=2e-2)
    np.testing.assert_allclose(losses[1:7], [float(t) for t in terms], rtol=2e-2, atol=1e-4)
    assert np.isfinite(ts.grads.cpu().numpy()).all()
    # gradients with bf16 operands in the frame-level forward and backward GEMMs, norm-wise against the fp32 oracle:
    # median 2 %, worst tensor 10 % (bf16 carries 8 mantissa bits; the forward rounding alone produces these figures,
    # the bf16 backward adds nothing measurable).  orgin_linear_change.2.bias is analytically zero (RnC is translation
    # invariant) and is skipped.
    gv = lay.views(torch.cat([ts.grads.cpu(), torch.zeros(lay.total - lay.live)]))
    errs = []
    for k in lay.live_names():
        if k == "orgin_linear_change.2.bias":
            continue
        ref = grads[k].double()
        errs.append(float((gv[k].double() - ref).norm() / (ref.norm() + 1e-12)))
    assert float(np.median(errs)) < 4e-2 and max(errs) < 0.2, (float(np.median(errs)), max(errs))


def test_fused_trainer_over_changing_batch_shapes_vs_oracle(E):
    """The reference's batches change shape from one to the next (per-batch padding, short last batch): FusedTrainer runs
    each through a per-shape fused step while ONE optimiser state (Adam moments, step count, dropout call counter)
    continues -- four steps over three shapes (one evicted and rebuilt) against the oracle's train_step sequence."""
    from oracle import sdumc_oracle as O
    dims = (24, 16, 20, 16)
    shapes = [(4, (9, 2, 5, 3)), (3, (7, 4, 6, 1)), (4, (9, 2, 5, 3)), (2, (12, 3, 2, 2))]
    P = O.init_params(dims, seed=8)
    flat, lay = flat_from(E, P, dims)
    tr = E.FusedTrainer(flat, dims, max_cached=2, lr=1e-3, seed=17)
    Pd, state = {k: v.clone() for k, v in P.items()}, {}
    for step, (B, Tn) in enumerate(shapes):
        audio, text, video, feat4, vals = O.synthetic_batch(B, Tn, dims, seed=100 + step)
        losses = tr.step(audio.cuda(), text.cuda(), video.cuda(), feat4.cuda(), vals.cuda()).cpu().numpy()
        loss, terms, grads, outs = O.train_step(Pd, state, audio, text, video, feat4, vals, mode="philox", seed=17, step=step,
                                                lr=1e-3)
        np.testing.assert_allclose(losses[1:7], [float(t) for t in terms], rtol=2e-4, atol=1e-6, err_msg=f"step {step}")
    assert len(tr._steps) == 2
    pv = lay.views(flat.cpu())
    for k in lay.live_names():
        close((pv[k] - P[k]) * 1e2, (Pd[k] - P[k]) * 1e2, 2e-2, k)       # four Adam steps of lr 1e-3
    assert tr.state.rng.call == 2 * len(shapes) and int(tr.state.hyper[1].item()) == len(shapes)


def test_cluster_spin_cap_fails_loudly_and_applies_nothing(E):
    """The failure path of csrc/chain_cluster.hip, driven on purpose: workgroup 0 withholds its arrivals
    (sdumc_chain_cluster_test_hold_), its cluster's members spin into their cap, the error word is set.  Contract
    (include/sdumc_hip.h): the step's total loss reads NaN, Adam applies NOTHING (parameters and moments unchanged, in the fused
    step and in the stand-alone sdumc_adam_step the data-parallel trainer calls), the word stays set until
    sdumc_chain_cluster_reset_error, after which steps are normal again."""
    from oracle import sdumc_oracle as O
    from sdumc_amd import _lib, ops
    dims = (64, 32, 64, 32)
    B, Tn = 6, (40, 6, 20, 6)
    P = O.init_params(dims, seed=4)
    batch = O.synthetic_batch(B, Tn, dims, seed=6)
    flat, lay = flat_from(E, P, dims)
    ts = E.TrainStep(flat, B, Tn, dims, seed=9)
    ts.set_batch(*[t.cuda() for t in batch])
    good = ts.run().cpu().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(good).all() and _lib.lib.sdumc_chain_cluster_error_() == 0
    before = flat.clone()
    m_before, v_before = ts.adam_m.clone(), ts.adam_v.clone()
    try:
        _lib.lib.sdumc_chain_cluster_test_hold_(1)
        bad = ts.run().cpu().clone()
        torch.cuda.synchronize()
    finally:
        _lib.lib.sdumc_chain_cluster_test_hold_(0)
    assert _lib.lib.sdumc_chain_cluster_error_() != 0
    assert torch.isnan(bad[0]), "the total loss of a step whose cluster spin hit its cap must read NaN"
    assert torch.equal(flat, before) and torch.equal(ts.adam_m, m_before) and torch.equal(ts.adam_v, v_before)
    # the word is sticky: the stand-alone Adam entry point (data-parallel trainer) is guarded by it too
    p2, g2 = torch.ones(1024).cuda(), torch.ones(1024).cuda()
    m2, v2 = torch.zeros(1024).cuda(), torch.zeros(1024).cuda()
    hyper = torch.tensor([1e-2, 0.0, 0.0, 0.0]).cuda()
    ops.adam_step(p2, g2, m2, v2, hyper)
    torch.cuda.synchronize()
    assert torch.equal(p2.cpu(), torch.ones(1024)) and float(m2.abs().max()) == 0.0
    assert _lib.lib.sdumc_chain_cluster_reset_error() == 0 and _lib.lib.sdumc_chain_cluster_error_() == 0
    ops.adam_step(p2, g2, m2, v2, hyper)
    again = ts.run().cpu().clone()
    torch.cuda.synchronize()
    assert float(p2.max()) < 1.0 and torch.isfinite(again).all() and not torch.equal(flat, before)
    assert _lib.lib.sdumc_chain_cluster_error_() == 0


def test_clustered_utterance_level_kernels_equal_the_plain_ones(E):
    """csrc/chain_cluster.hip (columns of every utterance-level layer split over 4 workgroups that exchange slices through HBM)
    against csrc/chain.hip on the same inputs: three optimisation steps with fresh masks each, ragged V (B = 7: the last
    cluster holds rows beyond V).  Same arithmetic, different summation order -> 1e-5; a stale exchange would be O(1)."""
    from oracle import sdumc_oracle as O
    from sdumc_amd import _lib
    dims = (64, 32, 64, 32)
    B, Tn = 7, (40, 6, 20, 6)
    P = O.init_params(dims, seed=4)
    batch = O.synthetic_batch(B, Tn, dims, seed=6)
    res = []
    try:
        for cluster in (0, 1):
            _lib.lib.sdumc_set_chain_cluster(cluster)
            flat, lay = flat_from(E, P, dims)
            ts = E.TrainStep(flat, B, Tn, dims, seed=9)
            ts.set_batch(*[t.cuda() for t in batch])
            ls = [ts.run().cpu().clone() for _ in range(3)]
            torch.cuda.synchronize()
            res.append((flat.cpu().clone(), ls, ts.grads.cpu().clone()))
    finally:
        _lib.lib.sdumc_set_chain_cluster(1)
    assert _lib.lib.sdumc_chain_cluster_error_() == 0
    for a, b in zip(res[0][1], res[1][1]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-5, atol=1e-6)
    gscale = float(res[0][2].abs().max())
    assert float((res[0][2] - res[1][2]).abs().max()) < 2e-5 * gscale
    assert not torch.equal(res[0][2], res[1][2])            # (the clustered path really ran)
    assert float((res[0][0] - res[1][0]).abs().max()) < 1e-4
