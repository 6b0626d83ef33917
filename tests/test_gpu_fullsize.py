"""GPU parity at the batch sizes BASELINE.json STATES, against the CPU oracle (it takes 5-10 s per step at these sizes):
  configs[1]  C2  MOSEI shapes, B = 64, fp32        one full train step vs oracle.train_step (Philox masks)
  configs[2]  C3  the same in bf16 storage           vs the fp32 oracle (activation bar 2e-2) AND vs an fp64 evaluation of the
                                                     same graph with the engine's bf16 rounding points (oracle/bf16_storage.py)
  configs[4]  C5  T = 512 x 3, d = 1024, per-GPU slice B = 32, fp32 and bf16 storage
B = 64 is where the production schedule differs from what the small-batch tests run: V = 128 -> the clustered utterance-level
kernels on 256 workgroups, the stream-K cuts of the grouped weight-gradient launches across 37 problems, the rows GEMM at
M = 48 000.  Reference: main_frame_val_text_missing.py:119-150 (the step), toolkit/models/wengnet_mosei_mult_views_text_missing.py
:275-370 (the network), toolkit/utils/loss.py:19-51, :271-315 (the losses)."""
import numpy as np
import pytest
import torch

from .test_gpu_configs import NAMES, close, flat_from, grad_errors

pytestmark = pytest.mark.gpu

C2 = ((1024, 4096, 1024, 4096), 64, (375, 32, 225, 32))
C5 = ((1024, 1024, 1024, 1024), 32, (512, 512, 512, 512))
# bf16 storage against the fp64 evaluation that rounds where the engine rounds: what is left is fp32 accumulation order, the
# hardware tanh / exp, and bf16 roundings that flip where the two sides differ in the last fp32 bits (a few thousand of the 24 M
# stored frame elements; they show in bias gradients that are sums with heavy cancellation).  Measured on MI355X: median over the
# gradient tensors 1.6e-5 (C5) / ~1e-5 (C3), worst tensor 2.0e-3 (C3: cross_fused_query_mlp.0.bias) and 4.8e-3 (C5:
# video_mlp.0.bias).  A kernel off by 5 % moves every tensor downstream of it to 5e-2.
BF16_EMU_GRAD_TOL = 1e-2          # every tensor
BF16_EMU_GRAD_P90 = 2e-3          # 90th percentile over the tensors
BF16_EMU_GRAD_MEDIAN = 5e-4
BF16_EMU_OUT_TOL = 2e-3


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import engine
    return engine


# fp32 at full size: every live gradient tensor by its OWN norm (77 of C2's 83 gradient tensors have max |g| < 1e-3 -- frame weights
# ~1e-5, the smallest 3.5e-10 -- so an absolute tolerance tied to max(1, |g|) passes an all-zero gradient; that was this test until
# round 6).  Measured on MI355X (profiles/r6_fullsize_grad_errors.txt): C2 median 1.0e-6, 90th percentile 1.3e-6, worst 2.5e-5
# (planes) / 5.7e-5 (in-kernel split), both on the RnC head (orgin_linear_change.0: small gradients with heavy cancellation); C5
# median 1.2e-6, worst 4e-6.  (The RnC head's analytically zero biases are handled by grad_errors: they must be as small on the
# device.)  A kernel that drops a term moves its tensors to >= 1e-2.
FP32_GRAD_TOL = 2e-4


_ORACLE_CACHE = {}


def _oracle_step(cfg, pseed, bseed, seed):
    """(P, batch, Pd after Adam, loss, terms, grads, outs): 5-10 s of CPU per call at these sizes, shared by the two product paths"""
    from oracle import sdumc_oracle as O
    key = (cfg, pseed, bseed, seed)
    if key not in _ORACLE_CACHE:
        dims, B, Tn = cfg
        P = O.init_params(dims, seed=pseed)
        batch = O.synthetic_batch(B, Tn, dims, seed=bseed)
        Pd = {k: v.clone() for k, v in P.items()}
        _ORACLE_CACHE[key] = (P, batch, Pd) + tuple(O.train_step(Pd, {}, *batch, mode="philox", seed=seed, step=0))
    return _ORACLE_CACHE[key]


def _step_fp32_vs_oracle(E, cfg, pseed, bseed, seed, delta_names, planes):
    """planes=False: the step a set_batch-per-step loop runs (in-kernel operand split, csrc/gemm_wide.hip); planes=True: the step over
    RESIDENT batches -- bench.py's headline and FusedTrainer.run_epoch -- whose frame / key projections read bf16 planes (csrc/gemm_p3.hip)."""
    dims, B, Tn = cfg
    P, batch, Pd, loss, terms, grads, outs = _oracle_step(cfg, pseed, bseed, seed)
    flat, lay = flat_from(E, P, dims)
    ts = E.TrainStep(flat, B, Tn, dims, seed=seed, planes=planes)
    assert (ts._planes is not None) == planes
    ts.set_batch(*[t.cuda() for t in batch])
    losses = ts.run().cpu().numpy()
    np.testing.assert_allclose(losses[0], float(loss), rtol=1e-3)              # north-star tolerance: 1e-3
    np.testing.assert_allclose(losses[1:7], [float(t) for t in terms], rtol=1e-3, atol=1e-5)
    for s in range(2):
        for n, got, want in zip(NAMES, (ts.vals, ts.fused, ts.rnc, ts.text_hidden, ts.cross_text), outs[s]):
            close(got[s * B:(s + 1) * B], want, 1e-3, f"{n} stream {s}")
    gv = lay.views(torch.cat([ts.grads.cpu(), torch.zeros(lay.total - lay.live)]))
    assert set(grads) == set(lay.live_names())
    for k in lay.live_names():
        close(gv[k], grads[k], 1e-3, k)           # (the north-star's absolute bar; says little about tensors far below 1e-3 ...)
    errs = grad_errors(lay, gv, grads)            # (... so: every tensor relative to its own norm)
    worst = max(errs, key=errs.get)
    vals = sorted(errs.values())
    print("fp32 %s, planes=%s: gradient errors relative to each tensor's norm: median %.3g, p90 %.3g, worst %s = %.3g" %
          ("C2" if cfg is C2 else "C5", planes, float(np.median(vals)), vals[int(0.9 * len(vals))], worst, errs[worst]))
    for k, e in errs.items():
        assert e < FP32_GRAD_TOL, f"{k}: relative gradient error {e:.3e} (|g| = {float(grads[k].norm()):.3e})"
    pv = lay.views(flat.cpu())
    for k in delta_names:          # post-Adam deltas (first step: lr * g / (|g| + eps))
        close((pv[k] - P[k]) * 1e4, (Pd[k] - P[k]) * 1e4, 2e-2, k)


@pytest.mark.parametrize("planes", [True, False])
def test_c2_fp32_step_at_batch_64_vs_oracle(E, planes):
    """BASELINE configs[1] at its stated batch: loss + six terms, five outputs of both streams, EVERY live gradient tensor (relative to
    its own norm) and the post-Adam deltas of three tensors against oracle.train_step(mode="philox"), on both product paths."""
    _step_fp32_vs_oracle(E, C2, 0, 1234, 777,
                         ("frame_dim_reshape_1.weight", "cross_att_fra2utt_0.input_proj.weight", "fc_out_v.weight"), planes)


@pytest.mark.parametrize("planes", [True, False])
def test_c5_fp32_step_at_per_gpu_batch_32_vs_oracle(E, planes):
    """BASELINE configs[4]'s per-GPU slice (B = 32, T = 512 for every modality, d = 1024) in fp32 against the oracle."""
    _step_fp32_vs_oracle(E, C5, 4, 21, 11,
                         ("frame_dim_reshape_0.weight", "cross_att_fra2utt_2.input_proj.weight", "cross_attention_mlp.0.weight"), planes)


def _step_bf16_vs_oracles(E, cfg, pseed, bseed, seed):
    from oracle import sdumc_oracle as O
    from oracle.bf16_storage import Bf16Storage
    dims, B, Tn = cfg
    P = O.init_params(dims, seed=pseed)
    flat, lay = flat_from(E, P, dims)
    batch = O.synthetic_batch(B, Tn, dims, seed=bseed)
    ts = E.TrainStep(flat, B, Tn, dims, seed=seed, bf16=True)
    ts.set_batch(*[t.cuda() for t in batch])
    losses = ts.run().cpu().numpy()
    got_outs = [t.cpu().clone() for t in (ts.vals, ts.fused, ts.rnc, ts.text_hidden, ts.cross_text)]
    gv = lay.views(torch.cat([ts.grads.cpu(), torch.zeros(lay.total - lay.live)]))
    # (a) the fp32 oracle: the activation bar of the bf16 configs (2e-2)
    loss, terms, grads32, outs = O.train_step({k: v.clone() for k, v in P.items()}, {}, *batch, mode="philox", seed=seed, step=0)
    np.testing.assert_allclose(losses[0], float(loss), rtol=2e-2)
    np.testing.assert_allclose(losses[1:7], [float(t) for t in terms], rtol=2e-2, atol=1e-4)
    for s in range(2):
        for n, got, want in zip(NAMES, got_outs, outs[s]):
            close(got[s * B:(s + 1) * B], want, 2e-2, f"bf16 vs fp32 oracle: {n} stream {s}")
    # (b) fp64 evaluation of the same graph on bf16-rounded stored tensors: rounding is reproduced, errors are not
    P64 = {k: v.double() for k, v in P.items()}
    loss64, terms64, grads64, outs64 = O.train_step(P64, {}, *[t.double() for t in batch], mode="philox", seed=seed, step=0,
                                                    st=Bf16Storage())
    np.testing.assert_allclose(losses[0], float(loss64), rtol=BF16_EMU_OUT_TOL)
    np.testing.assert_allclose(losses[1:7], [float(t) for t in terms64], rtol=BF16_EMU_OUT_TOL, atol=1e-5)
    for s in range(2):
        for n, got, want in zip(NAMES, got_outs, outs64[s]):
            close(got[s * B:(s + 1) * B], want, BF16_EMU_OUT_TOL, f"bf16 vs rounding emulation: {n} stream {s}")
    errs = grad_errors(lay, gv, grads64)
    worst = max(errs, key=errs.get)
    vals = sorted(errs.values())
    print("bf16 storage vs fp64 rounding emulation, gradient errors: median %.3g, worst %s = %.3g; vs fp32 oracle: worst %.3g" %
          (float(np.median(vals)), worst, errs[worst], max(grad_errors(lay, gv, grads32).values())))
    for k, e in errs.items():
        assert e < BF16_EMU_GRAD_TOL, (k, e)
    assert vals[int(0.9 * len(vals))] < BF16_EMU_GRAD_P90 and float(np.median(vals)) < BF16_EMU_GRAD_MEDIAN, \
        (float(np.median(vals)), vals[int(0.9 * len(vals))])


def test_c3_bf16_step_at_batch_64_vs_oracles(E):
    """BASELINE configs[2] at its stated batch and arithmetic (MOSEI shapes, B = 64, text-missing stream + self-distillation, bf16
    storage): against the fp32 oracle at the bf16 activation bar, and EVERY gradient tensor against the fp64 rounding emulation."""
    _step_bf16_vs_oracles(E, C2, 0, 1234, 777)


def test_c5_bf16_step_at_per_gpu_batch_32_vs_oracles(E):
    """BASELINE configs[4]'s per-GPU slice in its stated bf16 arithmetic."""
    _step_bf16_vs_oracles(E, C5, 4, 21, 11)
