"""The weight-gradient problems of one C2 backward (shared by tools/gg_bench.py and tools/gg_split_check.py)."""
import torch

B, Ta, Tt, Tv, D = 64, 375, 32, 225, 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)


def rn(*s):
    return torch.randn(*s, device=dev, generator=g)


def bits(K, N):
    return torch.randint(0, 16, (K, N // 4), device=dev, generator=g, dtype=torch.uint8)


def timeit(fn, reps):
    import time
    t0 = time.time()
    while time.time() - t0 < 0.4:      # clocks ramp up under load: warm up for 0.4 s of back-to-back launches
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def flops(ps):
    f = 0
    for q in ps:
        f += 2.0 * q["A"].shape[1] * q["B"].shape[1] * (q["A"].shape[0] + (q["A1"].shape[0] if q.get("A1") is not None else 0))
    return f


def frame_problems():
    ps = []
    ps.append({"A": rn(B * Ta, D), "B": rn(B * Ta, 1024), "colsum": torch.zeros(D, device=dev)})
    ps.append({"A": rn(B * Tv, D), "B": rn(B * Tv, 1024), "colsum": torch.zeros(D, device=dev)})
    ps.append({"A": rn(B * Tt, D), "B": rn(B * Tt, 4096), "A1": rn(B * Tt, D), "B1": rn(B * Tt, 4096),
               "colsum": torch.zeros(D, device=dev)})
    return ps


def key_problems(sites=(0, 1)):
    ps = []
    for T in (Ta, Tv, Tt):
        x = rn(B * T, D)
        for _ in sites:
            ps.append({"A": rn(2 * B * T, D), "B": x, "b_row_mod": B * T, "bits": bits(2 * B * T, D), "scale": 2.0,
                       "colsum": torch.zeros(D, device=dev)})
    return ps


def utt_problems():
    V, V7 = 2 * B, 14 * B
    ps = []
    for _ in range(6 + 1 + 7):
        ps.append({"A": rn(V, D), "B": rn(V, D), "colsum": torch.zeros(D, device=dev)})
    ps.append({"A": rn(V, D), "B": rn(V, 3 * D), "colsum": torch.zeros(D, device=dev)})
    for _ in range(6):
        ps.append({"A": rn(V7, D), "B": rn(V7, D), "colsum": torch.zeros(D, device=dev)})
    for _ in range(3):
        ps.append({"A": rn(V7, 128), "B": rn(V7, D), "colsum": torch.zeros(128, device=dev)})
    ps.append({"A": rn(V, D), "B": rn(V, 896), "colsum": torch.zeros(D, device=dev)})
    ps.append({"A": rn(V, 128), "B": rn(V, D), "colsum": torch.zeros(128, device=dev)})
    ps.append({"A": rn(V, 64), "B": rn(V, 128), "colsum": torch.zeros(64, device=dev)})
    ps.append({"A": rn(V, 64), "B": rn(V, 64), "colsum": torch.zeros(64, device=dev)})
    return ps


