"""GPU: the HIP backend of the data-parallel trainer, two ranks simulated in one process (the
collectives are replaced by explicit sums/concats): shard gradients summed == full-batch gradient,
post-Adam parameters identical to the single-GPU fused step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_simulated_ranks_equal_full_batch():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import sdumc_oracle as O
    from sdumc_amd import engine
    from sdumc_amd.trainer import HipBackend
    dims, Tn, B, W, seed = (64, 32, 48, 32), (70, 6, 30, 5), 4, 2, 99
    weights = engine.DEFAULT_WEIGHTS
    P = O.init_params(dims, seed=3)
    lay = engine.ParamLayout.get(*dims[:3])

    def flat():
        f = torch.zeros(lay.total)
        for k, v in lay.views(f).items():
            v.copy_(P[k])
        return f.cuda()

    gb = [t.cuda() for t in O.synthetic_batch(B * W, Tn, dims, seed=8)]
    full_p = flat()
    ts = engine.TrainStep(full_p, B * W, Tn, dims, weights=weights, seed=seed)
    ts.set_batch(*gb)
    ref_losses = ts.run().cpu().clone()
    ref_grads = ts.grads.clone()

    bes = []
    for r in range(W):
        be = HipBackend(flat(), B, Tn, dims, weights, 1e-4, (0.9, 0.999), 1e-8, 1e-5, seed, r * B, B * W)
        be.set_batch(*[t[r * B:(r + 1) * B].contiguous() for t in gb])
        bes.append(be)
    rncs = [be.forward().clone() for be in bes]
    ssd = sum(be.local_ssd().clone() for be in bes)                                   # all-reduce
    feats = torch.cat([p[:B] for p in rncs] + [p[B:] for p in rncs]).contiguous()     # all-gather + reorder
    lab = torch.cat([be.labels for be in bes])
    labels2 = torch.cat([lab, lab]).contiguous()
    ls = [be.loss_backward(ssd, feats, labels2, (r * B, W * B + r * B)).cpu().clone() for r, be in enumerate(bes)]
    gsum = sum(be.backward().clone() for be in bes)                                   # gradient all-reduce
    np.testing.assert_allclose(gsum.cpu().numpy(), ref_grads.cpu().numpy(), rtol=2e-3, atol=2e-6)
    # global loss terms: MSE entries are local sums / B_global; RMSE and RnC are global already
    np.testing.assert_allclose((ls[0][1:3] + ls[1][1:3]).numpy(), ref_losses[1:3].numpy(), rtol=1e-5)
    np.testing.assert_allclose(ls[0][3:7].numpy(), ref_losses[3:7].numpy(), rtol=1e-5)
    np.testing.assert_allclose(ls[1][3:7].numpy(), ref_losses[3:7].numpy(), rtol=1e-5)
    for be in bes:
        be.grads.copy_(gsum)
        be.adam(1.0)
    torch.cuda.synchronize()
    assert torch.equal(bes[0].params, bes[1].params)
    np.testing.assert_allclose(((bes[0].params - flat()) * 1e4).cpu().numpy(), ((full_p - flat()) * 1e4).cpu().numpy(),
                               rtol=2e-2, atol=2e-2)
    assert bes[0].rng.call == 2


def test_backward_phases_equal_one_call_and_early_slice_is_final():
    """sdumc_net_backward_phase(0) + (1) == sdumc_net_backward (to rounding: the grouped weight-gradient launches cut K at
    points that depend on what else rides in the launch, so the two differ in the last bits), and after phase 0 alone the slice
    [0, layout.early) -- what the data-parallel step all-reduces while phase 1 runs -- already holds its final
    values.  The early/late split is by layer: utterance-level first, frame-level (input_proj, context vectors,
    frame_dim_reshape) last."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import sdumc_oracle as O
    from sdumc_amd import engine
    from sdumc_amd.trainer import HipBackend
    dims, Tn, B, seed = (64, 32, 48, 32), (70, 6, 30, 5), 4, 7
    P = O.init_params(dims, seed=3)
    lay = engine.ParamLayout.get(*dims[:3])
    late = {n for n in lay.live_names() if lay.entries[n][0] >= lay.early}
    assert late == {n for n in lay.live_names()
                    if n.startswith("frame_dim_reshape_") or n.startswith("fra2utt_") or ".input_proj." in n}
    assert 0 < lay.early < lay.live and lay.early % 4 == 0

    def flat():
        f = torch.zeros(lay.total)
        for k, v in lay.views(f).items():
            v.copy_(P[k])
        return f.cuda()

    gb = [t.cuda() for t in O.synthetic_batch(B, Tn, dims, seed=8)]
    used = torch.zeros(lay.live, dtype=torch.bool)          # alignment padding between tensors is never written
    for n in lay.live_names():
        off, shape, _ = lay.entries[n]
        used[off:off + int(np.prod(shape))] = True
    res = []
    for phased in (False, True):
        be = HipBackend(flat(), B, Tn, dims, engine.DEFAULT_WEIGHTS, 1e-4, (0.9, 0.999), 1e-8, 1e-5, seed, 0, B)
        be.set_batch(*gb)
        be.forward()
        be.loss_backward()
        if phased:
            be.grads.fill_(float("nan"))
            early = be.backward_phase(0)
            torch.cuda.synchronize()
            assert early.numel() == lay.early and torch.isfinite(early.cpu()[used[:lay.early]]).all()
            # phase 1 has not run: the late slice is untouched, except the Cross_Attention input_proj gradients, whose
            # products need nothing from phase 1 and leave with phase 0's weight-gradient launch
            pending = used.clone()
            for n in lay.live_names():
                if n.startswith("cross_att_fra2utt_") and ".input_proj." in n:
                    off, shape, _ = lay.entries[n]
                    pending[off:off + int(np.prod(shape))] = False
            assert not torch.isfinite(be.grads.cpu()[lay.early:][pending[lay.early:]]).any()
            snap = early.clone()
            be.backward_phase(1)
            torch.cuda.synchronize()
            assert torch.equal(snap.cpu()[used[:lay.early]], be.grads.cpu()[:lay.early][used[:lay.early]])   # untouched by phase 1
        else:
            be.backward()
        torch.cuda.synchronize()
        res.append(be.grads.clone())
    a, b = res[0].cpu()[used].double(), res[1].cpu()[used].double()
    assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())


_RCCL_ONE_RANK = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SDUMC_REPO"])
from oracle import sdumc_oracle as O
from sdumc_amd import engine
from sdumc_amd.trainer import DataParallelStep
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["SDUMC_PORT"], rank=0, world_size=1, device_id=dev)
dims, Tn, B, seed = (64, 32, 48, 32), (70, 6, 30, 5), 6, 99
P = O.init_params(dims, seed=3)
lay = engine.ParamLayout.get(*dims[:3])
def flat():
    f = torch.zeros(lay.total)
    for k, v in lay.views(f).items():
        v.copy_(P[k])
    return f.cuda()
gb = [t.cuda() for t in O.synthetic_batch(B, Tn, dims, seed=8)]
p_ref = flat()
ts = engine.TrainStep(p_ref, B, Tn, dims, seed=seed); ts.set_batch(*gb)
p_dp = flat()
dp = DataParallelStep(p_dp, B, Tn, dims, seed=seed, exact=True, force_collectives=True); dp.set_batch(*gb)
assert dp.collect and dp.overlap == (os.environ.get('SDUMC_DP_OVERLAP') == '1')
ok = True
for it in range(3):
    l_ref = ts.run().cpu().clone()
    l_dp = dp.global_losses(dp.step()).cpu().clone()
    ok = ok and torch.allclose(l_ref[:7], l_dp[:7], rtol=1e-5, atol=1e-6)
torch.cuda.synchronize()
err = float((p_ref - p_dp).abs().max())
print(json.dumps({"losses_ok": bool(ok), "max_param_diff": err}))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_rccl_one_rank_communicator_runs_every_collective_and_matches_fused_step(tmp_path, overlap):
    """The N > 1 code path with a real RCCL communicator (world size 1, `force_collectives`): the merged exactness
    exchange and the gradient all-reduce -- one flat bucket (overlap 0, the default), or the asynchronous early-slice
    all-reduce beside the frame-level backward followed by the late slice (overlap 1).
    Three steps must reproduce the fused single-GPU TrainStep."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_ONE_RANK)
    env = dict(os.environ, SDUMC_REPO=repo, SDUMC_PORT=str(29547 + int(overlap)), HSA_ENABLE_IPC_MODE_LEGACY="0",
               SDUMC_DP_OVERLAP=overlap)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])   # RCCL prints a banner on stdout
    assert out["losses_ok"], out
    assert out["max_param_diff"] < 5e-6, out       # Adam steps of 1e-4: identical up to the reduction order of dW


@pytest.mark.parametrize("W,B", [(1, 5), (3, 4), (8, 64)])
def test_dp_exchange_record_pack_and_unpack_layout(W, B):
    """sdumc_dp_pack / sdumc_dp_unpack against the layout the loss takes: feats = stream-0 rows of every
    rank then stream-1 rows of every rank, labels twice, sums of squares added in rank order.  Bit-exact (pure moves +
    one fixed-order sum)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ctypes as C
    from sdumc_amd import _lib
    rd = 64
    g = torch.Generator().manual_seed(W * 100 + B)
    rncs = [torch.randn(2 * B, rd, generator=g).cuda() for _ in range(W)]
    labs = [torch.randn(B, generator=g).cuda() for _ in range(W)]
    ssds = [torch.rand(4, generator=g).cuda() for _ in range(W)]
    n = 2 * B * rd + B + 3
    recs = torch.empty(W, n, device="cuda")
    for r in range(W):
        _lib.check(_lib.lib.sdumc_dp_pack(_lib.ptr(rncs[r]), _lib.ptr(labs[r]), _lib.ptr(ssds[r]), B, rd,
                                          _lib.ptr(recs[r]), _lib.current_stream()), "sdumc_dp_pack")
        assert torch.equal(recs[r], torch.cat([rncs[r].reshape(-1), labs[r], ssds[r][:3]]))
    feats = torch.empty(2 * W * B, rd, device="cuda")
    labels2 = torch.empty(2 * W * B, device="cuda")
    ssd = torch.empty(4, device="cuda")
    _lib.check(_lib.lib.sdumc_dp_unpack(_lib.ptr(recs), W, B, rd, _lib.ptr(feats), _lib.ptr(labels2), _lib.ptr(ssd),
                                        _lib.current_stream()), "sdumc_dp_unpack")
    want_f = torch.cat([x[:B] for x in rncs] + [x[B:] for x in rncs])
    lab = torch.cat(labs)
    want_s = ssds[0][:3].clone()
    for r in range(1, W):
        want_s = want_s + ssds[r][:3]
    assert torch.equal(feats, want_f)
    assert torch.equal(labels2, torch.cat([lab, lab]))
    assert torch.equal(ssd[:3], want_s)


@pytest.mark.parametrize("B", [3, 64, 257])
def test_dp_record_one_launch_equals_ssd_plus_pack_and_is_deterministic(B):
    """sdumc_dp_record (sums of squares + pack, last-block ordered reduction) == float64 sums to 1e-5 relative, the copied
    part bit-exact, identical bits across repeated launches, workspace left re-armed."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import _lib, engine
    lib = _lib.lib
    rd, D, H, NQ = engine.RNC_DIM, engine.D, engine.H, engine.NQ
    g = torch.Generator().manual_seed(B)
    th = torch.randn(2 * B, D, generator=g).cuda()
    ct = torch.randn(2 * B, NQ, H, generator=g).cuda()
    z = torch.randn(2 * B, H, generator=g).cuda()
    rnc = torch.randn(2 * B, rd, generator=g).cuda()
    labels = torch.randn(B, generator=g).cuda()
    ws = torch.zeros(lib.sdumc_dp_record_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    nf = 2 * B * rd
    outs = []
    for _ in range(3):
        rec = torch.full((nf + B + 3,), float("nan"), device="cuda")
        _lib.check(lib.sdumc_dp_record(B, rd, _lib.ptr(th), _lib.ptr(ct), _lib.ptr(z), _lib.ptr(rnc), _lib.ptr(labels),
                                       _lib.ptr(rec), _lib.ptr(ws), _lib.current_stream()), "sdumc_dp_record")
        outs.append(rec.clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    rec = outs[0]
    assert torch.equal(rec[:nf], rnc.reshape(-1))
    assert torch.equal(rec[nf:nf + B], labels)
    want = [float(((x[B:].double() - x[:B].double()) ** 2).sum()) for x in (th, ct, z)]
    np.testing.assert_allclose(rec[nf + B:].cpu().numpy(), np.array(want), rtol=1e-5)
    assert int(ws[:4].view(torch.int32).item()) == 0


def _run_bench(extra_env, timeout):
    import json
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDUMC_DIST_BACKEND="gloo", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, timeout=timeout, cwd=root)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    return r.returncode, ([json.loads(ln) for ln in lines]), time.time() - t0, r.stderr.decode(errors="replace")


def test_bench_spawns_its_own_ranks_and_reports_one_line():
    """`bench.py --gpus 2` without a launcher (a fresh process that starts its ranks before any GPU call of its own), two ranks
    sharing the one GPU over gloo: ONE JSON line with n_gpus 2 and the data_parallel block (the driver's contract)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    rc, lines, dt, err = _run_bench({}, 600)
    assert rc == 0, err[-2000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and "data_parallel" in d
    assert d["value"] > 0 and np.isfinite(d["config"]["final_loss"])


def test_bench_parent_stops_everything_when_a_rank_dies():
    """rank 1 exits in its set-up (SDUMC_TEST_FAIL_RANK): rank 0 would sit in the rendezvous until torch's timeout; the parent's
    watchdog terminates it and returns rank 1's exit code within a minute."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    rc, lines, dt, err = _run_bench({"SDUMC_TEST_FAIL_RANK": "1"}, 120)
    assert rc == 3 and not lines and dt < 60, (rc, dt, err[-1000:])
    assert "rank 1 exited with code 3" in err


_TWO_RANKS_ONE_FAILS = r"""
import json, os, sys
sys.path.insert(0, os.environ["SDUMC_REPO"])
import torch
import torch.distributed as dist
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2, init_method="tcp://127.0.0.1:" + os.environ["SDUMC_PORT"])
torch.cuda.set_device(0)
from oracle import sdumc_oracle as O       # (parameter initialisation and the synthetic batch only)
from sdumc_amd import _lib, engine as E
from sdumc_amd.trainer import DataParallelStep
dims, B, Tn = (64, 32, 64, 32), 4, (40, 6, 20, 6)
P = O.init_params(dims, seed=4)
lay = E.ParamLayout.get(*dims[:3])
flat = torch.zeros(lay.total)
for k, v in lay.views(flat).items():
    v.copy_(P[k])
flat = flat.cuda()
batch = [t.cuda() for t in O.synthetic_batch(2 * B, Tn, dims, seed=6)]
dp = DataParallelStep(flat, B, Tn, dims, seed=9)
dp.set_batch(*[t[rank * B:(rank + 1) * B].contiguous() for t in batch])
good = dp.global_losses(dp.step())
torch.cuda.synchronize()
after_good = flat.clone()
if rank == 1:
    _lib.lib.sdumc_chain_cluster_test_hold_(1)      # THIS rank's clustered kernels run into their spin cap
losses = dp.step()
torch.cuda.synchronize()
_lib.lib.sdumc_chain_cluster_test_hold_(0)
raised = False
try:
    dp.global_losses(losses)
except _lib.SdumcError:
    raised = True
out = {"rank": rank, "good_finite": bool(torch.isfinite(good).all()), "unchanged_in_failed_step": bool(torch.equal(flat, after_good)), "raised": raised,
       "error_word": int(_lib.lib.sdumc_chain_cluster_error_())}
print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_two_ranks_one_cluster_failure_stops_every_rank(tmp_path):
    """Data-parallel failure contract: the clustered kernels' error word is per device, so a rank whose spin hit its cap
    (driven on purpose on rank 1 with sdumc_chain_cluster_test_hold_) must not leave the OTHER rank applying the all-reduced
    garbage.  The word rides behind the gradient bucket through the same all-reduce (trainer.HipBackend.err_flag / err_merge):
    after the failed step BOTH ranks' parameters are unchanged, BOTH error words are set and BOTH raise in global_losses.
    Two processes sharing the one GPU over gloo."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two_ranks_one_fails.py"
    script.write_text(_TWO_RANKS_ONE_FAILS)
    procs = []
    for rank in range(2):
        env = dict(os.environ, SDUMC_REPO=repo, SDUMC_PORT="29561", RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se[-3000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("{")][-1]))
    for o in outs:
        assert o["good_finite"] and o["unchanged_in_failed_step"] and o["raised"] and o["error_word"] != 0, o
