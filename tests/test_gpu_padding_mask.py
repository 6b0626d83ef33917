"""GPU: the key-padding extension of SURVEY §8f F1 (default OFF: the reference lets zero-padded frames take part in the
softmax over time, model :63,:90).  With `lengths` given, frames at or beyond a sample's valid length get attention weight
exactly 0.  The defining property: a right-padded batch with lengths == every sample run on its own, unpadded."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def close(got, want, tol=1e-4, msg=""):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    scale = max(1.0, np.abs(want).max())
    np.testing.assert_allclose(got, want.reshape(got.shape), rtol=tol, atol=tol * scale, err_msg=msg)


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import engine
    return engine


@pytest.mark.parametrize("nq,T", [(1, 5), (7, 64), (7, 200), (3, 129)])
def test_attnpool_lengths_vs_masked_softmax(E, nq, T):
    from sdumc_amd import ops
    V, D = 6, 256
    g = torch.Generator().manual_seed(nq * 1000 + T)
    x = torch.randn(V, T, D, generator=g)
    keys = torch.tanh(torch.randn(V, T, D, generator=g))
    q = torch.randn(V, nq, D, generator=g) * 0.3
    lens = torch.tensor([T, 1, max(1, T // 2), max(1, T - 1), min(T, 65), 0], dtype=torch.int32)   # 0 is clamped to 1
    xg, kg, qg, lg = x.cuda(), keys.cuda(), q.cuda(), lens.cuda()     # the descriptor holds raw pointers: keep them alive
    out, attn, pooled, desc = ops.attnpool_fwd(xg, kg, qg, nq, lengths=lg)
    xd, kd, qd = x.double().requires_grad_(), keys.double().requires_grad_(), q.double().requires_grad_()
    sc = 0.3 * torch.bmm(kd, qd.transpose(1, 2))
    valid = torch.arange(T).unsqueeze(0) < lens.clamp(min=1).unsqueeze(1)
    w = torch.softmax(sc.masked_fill(~valid.unsqueeze(2), float("-inf")), dim=1)
    want = torch.einsum("vtq,vtd->vqd", w, xd)
    close(attn, w, 1e-5, "weights")
    assert bool((attn.cpu()[~valid] == 0).all())                    # exactly zero, not merely small
    close(out, want, 1e-4, "pooled")
    dout = torch.randn(V, nq, D, generator=g)
    want.backward(dout.double())
    dz, dxd, dq = ops.attnpool_bwd(desc, dout.cuda(), (xg, kg, qg, lg))
    close(dxd, xd.grad, 1e-4, "dxd")
    close(dq, qd.grad, 1e-4, "dq")
    close(dz, kd.grad * (1 - keys.double() ** 2), 1e-4, "dz")       # dz = gradient w.r.t. the pre-tanh projection
    assert bool((dz.cpu()[~valid] == 0).all()) and bool((dxd.cpu()[~valid] == 0).all())


def _params(E, dims, seed):
    from oracle import sdumc_oracle as O
    P = O.init_params(dims, seed=seed)
    lay = E.ParamLayout.get(*dims[:3])
    flat = torch.zeros(lay.total)
    for k, v in lay.views(flat).items():
        v.copy_(P[k])
    return P, flat.cuda(), lay


def test_padded_batch_with_lengths_equals_unpadded_samples(E):
    """eval mode, both streams: the batch result row b == the same sample alone with its true lengths; and without
    `lengths` the padded frames DO change the result (the reference's behaviour the default keeps)."""
    from oracle import sdumc_oracle as O
    dims, Tn, B = (24, 16, 20, 16), (70, 9, 33, 6), 4
    P, flat, lay = _params(E, dims, 13)
    audio, text, video, feat4, _ = O.synthetic_batch(B, Tn, dims, seed=3)
    lens = [torch.tensor(l, dtype=torch.int32) for l in ([70, 12, 65, 1], [9, 3, 1, 8], [33, 33, 2, 17], [6, 1, 4, 5])]
    for t, l in zip((audio, text, video, feat4), lens):
        for b in range(B):
            t[b, int(l[b]):] = 0
    call = E.NetCall(flat, audio.cuda(), [text.cuda(), feat4.cuda()], video.cuda(), False, None, lengths=lens)
    outs = [o.cpu().clone() for o in call.forward()]
    plain = [o.cpu().clone() for o in E.NetCall(flat, audio.cuda(), [text.cuda(), feat4.cuda()], video.cuda(), False, None).forward()]
    rel = max(float((o - q_).abs().max() / q_.abs().max()) for o, q_ in zip(outs[1:], plain[1:]))
    assert rel > 1e-3, rel          # without the mask the zero frames do take part in the softmax
    for b in range(B):
        a1 = audio[b:b + 1, :int(lens[0][b])].contiguous().cuda()
        t1 = text[b:b + 1, :int(lens[1][b])].contiguous().cuda()
        v1 = video[b:b + 1, :int(lens[2][b])].contiguous().cuda()
        f1 = feat4[b:b + 1, :int(lens[3][b])].contiguous().cuda()
        solo = [o.cpu() for o in E.NetCall(flat, a1, [t1, f1], v1, False, None).forward()]
        for o, s_ in zip(outs, solo):
            close(o[b], s_[0], 1e-4, f"sample {b} stream 0")
            close(o[B + b], s_[1], 1e-4, f"sample {b} stream 1")


def test_train_step_with_lengths_vs_oracle(E):
    """fused step, train mode (Philox masks), key-padding lengths on: losses and every gradient against the oracle."""
    from oracle import sdumc_oracle as O
    dims, Tn, B = (24, 16, 20, 16), (70, 9, 33, 6), 4
    P, flat, lay = _params(E, dims, 14)
    audio, text, video, feat4, vals = O.synthetic_batch(B, Tn, dims, seed=4)
    lens = [torch.tensor(l, dtype=torch.int32) for l in ([70, 12, 65, 1], [9, 3, 1, 8], [33, 33, 2, 17], [6, 1, 4, 5])]
    for t, l in zip((audio, text, video, feat4), lens):
        for b in range(B):
            t[b, int(l[b]):] = 0
    ts = E.TrainStep(flat, B, Tn, dims, seed=31)
    ts.set_batch(audio.cuda(), text.cuda(), video.cuda(), feat4.cuda(), vals.cuda())
    ts.set_lengths(lens)
    losses = ts.run().cpu().numpy()
    Pd = {k: v.clone() for k, v in P.items()}
    loss, terms, grads, outs = O.train_step(Pd, {}, audio, text, video, feat4, vals, mode="philox", seed=31, step=0, lengths=lens)
    np.testing.assert_allclose(losses[0], float(loss), rtol=1e-4)
    np.testing.assert_allclose(losses[1:7], [float(t) for t in terms], rtol=1e-4, atol=1e-6)
    gv = lay.views(torch.cat([ts.grads.cpu(), torch.zeros(lay.total - lay.live)]))
    for k in lay.live_names():
        close(gv[k], grads[k], 5e-4, k)
    # switching the extension off again restores the reference's behaviour
    ts2 = E.TrainStep(_params(E, dims, 14)[1], B, Tn, dims, seed=31)
    ts2.set_batch(audio.cuda(), text.cuda(), video.cuda(), feat4.cuda(), vals.cuda())
    ts2.set_lengths(lens)
    ts2.set_lengths(None)
    l_off = ts2.run().cpu().numpy()
    _, terms_off, _, _ = O.train_step({k: v.clone() for k, v in P.items()}, {}, audio, text, video, feat4, vals, mode="philox", seed=31, step=0)
    np.testing.assert_allclose(l_off[1:7], [float(t) for t in terms_off], rtol=1e-4, atol=1e-6)


def test_lengths_must_be_all_or_none(E):
    from oracle import sdumc_oracle as O
    from sdumc_amd import _lib
    dims, Tn, B = (24, 16, 20, 16), (5, 3, 4, 2), 2
    P, flat, lay = _params(E, dims, 1)
    audio, text, video, feat4, _ = O.synthetic_batch(B, Tn, dims, seed=1)
    with pytest.raises(_lib.SdumcError):
        E.NetCall(flat, audio.cuda(), [text.cuda(), feat4.cuda()], video.cuda(), False, None,
                  lengths=[torch.ones(B, dtype=torch.int32)] * 3)
