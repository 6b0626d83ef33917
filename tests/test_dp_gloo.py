"""World-size-2 data-parallel test on CPU (gloo): the collective algebra of sdumc_amd/trainer.py
(RMSE sum all-reduce, RnC feature/label all-gather with per-rank row ranges, ONE flat gradient
all-reduce, Philox masks keyed by the global sample index) must make DP(2 x B) identical to one
process on the whole batch.  The compute backend injected here is the CPU oracle (test
infrastructure); on the GPU the same trainer drives HipBackend (tests/test_gpu_dp.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

DIMS = (24, 16, 20, 16)
T_ = (9, 4, 7, 3)
B_LOCAL, WORLD, STEPS, SEED = 3, 2, 2, 4321
WEIGHTS = (0.5, 0.5, 0.1, 0.7, 0.1, 0.8)


class OracleBackend:
    def __init__(self, flat_params, B, T, dims, weights, lr, betas, eps, weight_decay, seed, sample0, B_global):
        from oracle import sdumc_oracle as O
        self.O = O
        self.P = flat_params            # here: dict name -> tensor (float64), updated in place
        self.names = [k for k in self.P if not O.is_dead(k)]
        self.B, self.Bg, self.w, self.seed, self.sample0 = B, B_global, weights, seed, sample0
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.state, self.step_idx = {}, 0
        self.losses = torch.zeros(8, dtype=torch.float64)

    def set_batch(self, audio, text, video, feat4, labels):
        self.batch = (audio, text, video, feat4)
        self.labels = labels.reshape(-1).clone()

    def forward(self):
        O = self.O
        self.leaves = {k: v.detach().clone().requires_grad_(k in self.names) for k, v in self.P.items()}
        a, t, v, f4 = self.batch
        outs = []
        for s, tx in enumerate((t, f4)):
            d = O.DropCtx("philox", self.seed, 2 * self.step_idx + s, self.sample0)
            y, (z, r, th, ct) = O.forward(self.leaves, a, tx, v, d)
            outs.append((y, z, r, th, ct))
        self.outs = outs
        return torch.cat([outs[0][2], outs[1][2]]).detach()

    def local_ssd(self):
        (y0, z0, r0, t0, c0), (y1, z1, r1, t1, c1) = self.outs
        return torch.stack([((t1 - t0) ** 2).sum(), ((c1 - c0) ** 2).sum(), ((z1 - z0) ** 2).sum()]).detach()

    def loss_backward(self, ssd_global=None, feats_global=None, labels_global=None, row0=(0, 0)):
        O, B, Bg, w = self.O, self.B, self.Bg, self.w
        (y0, z0, r0, t0, c0), (y1, z1, r1, t1, c1) = self.outs
        y = self.labels
        ssd = ssd_global if ssd_global is not None else self.local_ssd()
        numel = [Bg * 256, Bg * 7 * 128, Bg * 128]
        rm = [torch.sqrt(ssd[i] / numel[i]) for i in range(3)]
        mse0, mse1 = ((y0.reshape(-1) - y) ** 2).sum() / Bg, ((y1.reshape(-1) - y) ** 2).sum() / Bg
        sur = (w[0] * mse0 + w[1] * mse1
               + w[2] * ((t1 - t0.detach()) ** 2).sum() / (2 * rm[0] * numel[0])
               + w[3] * ((c1 - c0.detach()) ** 2).sum() / (2 * rm[1] * numel[1])
               + w[4] * ((z1 - z0) ** 2).sum() / (2 * rm[2] * numel[2]))
        if feats_global is None:
            F, lab = torch.cat([r0, r1]), torch.cat([y, y])
        else:
            F = feats_global.clone()
            F = torch.cat([F[:row0[0]], r0, F[row0[0] + B:row0[1]], r1, F[row0[1] + B:]])
            lab = labels_global
        rnc = O.rnc_loss(torch.stack((F[:Bg], F[Bg:]), dim=1), lab[:Bg].reshape(-1, 1))
        (sur + w[5] * rnc).backward()
        self.losses = torch.stack([torch.zeros(()), mse0, mse1, rm[0], rm[1], rm[2], rnc, torch.zeros(())]).detach().double()
        self.losses[0] = sum(wi * v for wi, v in zip(w, self.losses[1:7]))
        return self.losses

    def backward(self):
        self.grads = torch.cat([self.leaves[k].grad.reshape(-1) for k in self.names])
        return self.grads

    def adam(self, grad_scale=1.0):
        O, off = self.O, 0
        for k in self.names:
            n = self.P[k].numel()
            g = self.grads[off:off + n].view(self.P[k].shape) * grad_scale
            off += n
            m, v = self.state.get(k, (torch.zeros_like(g), torch.zeros_like(g)))
            p, m, v = O.adam_update(self.P[k], g, m, v, self.step_idx + 1, lr=self.lr, b1=self.betas[0],
                                    b2=self.betas[1], eps=self.eps, wd=self.wd)
            self.P[k].copy_(p)
            self.state[k] = (m, v)
        self.step_idx += 1


def _global_batch():
    from oracle import sdumc_oracle as O
    return [t.double() for t in O.synthetic_batch(B_LOCAL * WORLD, T_, DIMS, seed=77)]


def _worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    from oracle import sdumc_oracle as O
    from sdumc_amd.trainer import DataParallelStep
    P = {k: v.double() for k, v in O.init_params(DIMS, seed=5).items()}
    dp = DataParallelStep(P, B_LOCAL, T_, DIMS, weights=WEIGHTS, seed=SEED, exact=True, backend_factory=OracleBackend)
    assert dp.world == WORLD and dp.B_global == B_LOCAL * WORLD and dp.be.sample0 == rank * B_LOCAL
    gb = _global_batch()
    lo = rank * B_LOCAL
    dp.set_batch(*[t[lo:lo + B_LOCAL] for t in gb])
    hist = []
    for _ in range(STEPS):
        hist.append(dp.global_losses(dp.step()).clone())
    if rank == 0:
        torch.save({"P": P, "losses": torch.stack(hist)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
    got = torch.load(out)
    from oracle import sdumc_oracle as O
    P = {k: v.double() for k, v in O.init_params(DIMS, seed=5).items()}
    state, gb = {}, _global_batch()
    ref_losses = []
    for step in range(STEPS):
        loss, terms, _, _ = O.train_step(P, state, *gb, weights=WEIGHTS, mode="philox", seed=SEED, step=step)
        ref_losses.append([float(loss)] + [float(t) for t in terms])
    np.testing.assert_allclose(got["losses"][:, :7].numpy(), np.array(ref_losses), rtol=1e-9, atol=1e-12)
    for k in P:
        np.testing.assert_allclose(got["P"][k].numpy(), P[k].numpy(), rtol=1e-8, atol=1e-12, err_msg=k)


def test_single_process_trainer_path_equals_train_step():
    """world_size 1: DataParallelStep degenerates to the plain step (no collectives)."""
    from oracle import sdumc_oracle as O
    from sdumc_amd.trainer import DataParallelStep
    P = {k: v.double() for k, v in O.init_params(DIMS, seed=5).items()}
    dp = DataParallelStep(P, B_LOCAL * WORLD, T_, DIMS, weights=WEIGHTS, seed=SEED, backend_factory=OracleBackend)
    gb = _global_batch()
    dp.set_batch(*gb)
    l = dp.step()
    Q = {k: v.double() for k, v in O.init_params(DIMS, seed=5).items()}
    loss, terms, _, _ = O.train_step(Q, {}, *gb, weights=WEIGHTS, mode="philox", seed=SEED, step=0)
    np.testing.assert_allclose(float(l[0]), float(loss), rtol=1e-10)
    for k in Q:
        np.testing.assert_allclose(P[k].numpy(), Q[k].numpy(), rtol=1e-9, atol=1e-13, err_msg=k)
