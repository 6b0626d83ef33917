"""K3 (sdumc_umca_fwd, one kernel: key projection + scores + softmax partials + pooling) against the composition it replaces
(wide NT GEMM with fused mask/bias/tanh -> keys in HBM -> sdumc_attnpool_fwd) at the C2 shapes of the three modalities, both
attention sites (nq = 1: FRA2UTT_new, nq = 7: Cross_Attention).  Reports us per call, with and without keeping the keys."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops, _lib

dev = "cuda"


def timeit(fn, reps=30, rounds=3):
    best = 1e9
    for _ in range(rounds):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def case(name, B, T, nq):
    import ctypes as C
    V, Dm = 2 * B, 256
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, T, Dm, device=dev, generator=g)
    W = torch.randn(Dm, Dm, device=dev, generator=g) / 16
    b = torch.randn(Dm, device=dev, generator=g) * 0.1
    q = torch.randn(V, nq, Dm, device=dev, generator=g) / 4
    xdrop = _lib.make_dropout(True, 23, 0.5, T, Dm, B, call0=4, seed=9)
    bits = ops.dropout_bits(xdrop, 2)
    odrop = _lib.make_dropout(True, 24, 0.5, nq, Dm, B, call0=4, seed=9)
    # pre-built descriptors (the timing excludes allocation)
    attn, pooled, out = torch.empty(V, T, nq, device=dev), torch.empty(V, nq, Dm, device=dev), torch.empty(V, nq, Dm, device=dev)
    keys = torch.empty(V, T, Dm, device=dev)
    need = _lib.lib.sdumc_attnpool_fwd_workspace_bytes(V, T, nq)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)

    def desc(k):
        a = ops.attnpool_desc(x, k, q, V, T, nq, B, nq * Dm, xdrop, odrop, attn, pooled, out, tickets=False)
        a.workspace, a.workspace_bytes = _lib.ptr(ws), need
        return a
    a_keys, a_nokeys = desc(keys), desc(None)
    u1, u0 = _lib.Umca(), _lib.Umca()
    u1.a, u0.a = a_keys, a_nokeys
    for u in (u1, u0):
        u.w_in, u.b_in = _lib.ptr(W), _lib.ptr(b)
    st = _lib.current_stream

    def k3(u):
        _lib.check(_lib.lib.sdumc_umca_fwd(C.byref(u), st()), "umca")

    def comp():
        ops.gemm(ops.NT, x, W, V * T, Dm, Dm, bias=b, act=ops.ACT_TANH, a_row_mod=B * T, a_drop=xdrop, C_out=keys.view(V * T, Dm))
        _lib.check(_lib.lib.sdumc_attnpool_fwd(C.byref(a_keys), st()), "attn")
    t_comp, t_k3, t_k3n = timeit(comp), timeit(lambda: k3(u1)), timeit(lambda: k3(u0))
    gf = 2.0 * V * T * Dm * Dm / 1e9
    print(f"{name:22s} V={V} T={T:4d} nq={nq}: GEMM+pool {t_comp:7.1f} us | K3 keeping keys {t_k3:7.1f} us | K3 no keys {t_k3n:7.1f} us"
          f"  ({gf:.2f} GFLOP of projection: {gf / t_k3n * 1e3:.0f} TF through K3)", flush=True)
    del bits


if __name__ == "__main__":
    for nq in (1, 7):
        case("audio", 64, 375, nq)
        case("video", 64, 225, nq)
        case("text", 64, 50, nq)
    case("C5 audio T=512", 32, 512, 7)
