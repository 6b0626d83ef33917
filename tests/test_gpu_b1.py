"""csrc/gemm_b1.hip: the NT products of the bf16-storage step with the weight fragment-major in registers (include/sdumc_hip.h:
sdumc_gemm_b1).  Replaces, with sdumc_net_dims.bf16 = 2, F.linear of frame_dim_reshape_{0,1,2} (model :193-195, :282-284) and of the
input_proj key projections (model :60, :82).  bf16 operands, fp32 accumulation: against an fp64 product of the SAME bf16 operands the
fp32 output is accurate to accumulation order (1e-5 of the largest value), the bf16 output to half a bf16 ulp on top (4e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import ops
    return ops


def err(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


def _case(ops, M, K, act, mod=0, sk=0, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    X = torch.randn(mod or M, K, device="cuda", generator=g).to(BF)
    W = torch.randn(256, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(256, device="cuda", generator=g) * 0.1
    Xd = X.double().repeat(2, 1)[:M] if mod else X.double()
    ref = Xd @ W.to(BF).double().t() + b.double()
    ref = torch.tanh(ref) if act == ops.ACT_TANH else (ref.clamp_min(0) if act == ops.ACT_RELU else ref)
    return X, ops.b1_frag(W), b, ref


@pytest.mark.parametrize("M,K,act,mod,sk", [(9003, 1024, 0, 0, 0), (9003, 1024, 2, 0, 4), (2048, 4096, 0, 0, 0), (2048, 4096, 0, 0, 3),
                                            (700, 128, 1, 0, 0), (36896, 256, 2, 18448, 0), (64, 256, 0, 0, 0), (1, 128, 0, 0, 0)])
def test_nt_against_fp64_of_the_same_operands(ops, M, K, act, mod, sk):
    X, Wf, b, ref = _case(ops, M, K, act, mod, sk)
    for od, bound in ((torch.float32, 1e-5), (BF, 4e-3)):
        c = ops.gemm_b1_nt(X, Wf, M, 256, K, bias=b, act=act, a_row_mod=mod, out_dtype=od, splitk=sk)
        assert c.dtype == od and err(c, ref) < bound
        assert torch.equal(c, ops.gemm_b1_nt(X, Wf, M, 256, K, bias=b, act=act, a_row_mod=mod, out_dtype=od, splitk=sk)), "run to run"


def test_fragment_major_weight_is_the_rounded_weight(ops):
    g = torch.Generator(device="cuda").manual_seed(3)
    W = torch.randn(256, 128, device="cuda", generator=g)
    f = ops.b1_frag(W).view(BF).view(8, 8, 2, 32, 8)          # [row block][k-tile][k half][row in block][8 k]
    back = f.permute(0, 3, 1, 2, 4).reshape(256, 128)
    assert torch.equal(back, W.to(BF))


def test_two_a_tensors_share_one_launch(ops):
    g = torch.Generator(device="cuda").manual_seed(5)
    Xa, Xb = (torch.randn(1024, 4096, device="cuda", generator=g).to(BF) for _ in range(2))
    W, b = torch.randn(256, 4096, device="cuda", generator=g) / 64, torch.randn(256, device="cuda", generator=g)
    Wf = ops.b1_frag(W)
    both = ops.gemm_b1_nt(Xa, Wf, 2048, 256, 4096, bias=b, out_dtype=torch.float32, A_second=Xb, second_row0=1024)
    one = ops.gemm_b1_nt(torch.cat([Xa, Xb]), Wf, 2048, 256, 4096, bias=b, out_dtype=torch.float32)
    assert torch.equal(both, one)


def test_rejects_what_it_cannot_run(ops):
    from sdumc_amd._lib import SdumcError
    X = torch.zeros(64, 192, device="cuda", dtype=BF)
    Wf = torch.zeros(8, 64 * 192, device="cuda", dtype=torch.uint8)
    with pytest.raises(SdumcError):
        ops.gemm_b1_nt(X, Wf, 64, 256, 192)            # K % 128
    with pytest.raises(SdumcError):
        ops.gemm_b1_nt(X[:, :128], Wf, 64, 128, 128)   # N % 256


def test_step_matches_the_lds_staged_kernels(ops, monkeypatch):
    """The bf16-storage train step through gemm_b1 lands where the step through gemm_bf16.hip's tiles does: same operands, same fp32
    accumulation, different summation order inside a product -> losses agree to 1e-3 relative after 3 steps (subprocess: the switch is
    read once per process)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json,torch,sys; sys.path.insert(0, %r); from sdumc_amd import engine as E; from sdumc_amd.engine import ParamLayout;"
            "lay = ParamLayout.get(1024, 4096, 1024); g = torch.Generator().manual_seed(0);"
            "flat = (torch.randn(lay.total, generator=g) * 0.02).cuda(); B = 16; T = (50, 32, 30, 32);"
            "ts = E.TrainStep(flat, B, T, (1024, 4096, 1024), seed=7, bf16=True);"
            "r = lambda *s: torch.randn(*s, generator=g).cuda();"
            "ts.set_batch(r(B, 50, 1024), r(B, 32, 4096), r(B, 30, 1024), r(B, 32, 4096), r(B));"
            "print(json.dumps([float(ts.run()[0]) for _ in range(3)]))") % root
    out = []
    for v in ("0", "1"):
        env = dict(os.environ, SDUMC_P3=v)
        out.append(json.loads(subprocess.check_output([sys.executable, "-c", code], env=env, cwd=root).decode().strip().splitlines()[-1]))
    for a, b in zip(*out):
        assert abs(a - b) <= 1e-3 * abs(a), out
