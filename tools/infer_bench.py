"""Inference forward (eval mode, one or two streams) at the C2 shapes: samples/s of sdumc_net_forward with the six attention
sites as K3 (fused key projection + pooling, no keys tensor) against the two-kernel path (SDUMC_K3=0), and a parity check of
the two.  usage: python tools/infer_bench.py [B=64]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    import bench
    from sdumc_amd import engine
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
    dev = torch.device("cuda:0")
    flat, lay = bench.init_flat_params(engine, dev)
    audio, text, video, feat4, labels = [t.to(dev) for t in bench.synthetic_shard(B, 0)]
    out = {}
    for streams, texts in ((1, [text]), (2, [text, feat4])):
        nc = engine.NetCall(flat, audio, texts, video, train=False, rng=None)
        for _ in range(5):
            nc.forward()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(50):
                nc.forward()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
        out[streams] = (best, [t.clone() for t in nc.forward()])
        torch.cuda.synchronize()
        print(f"K3={os.environ.get('SDUMC_K3', '1')} streams={streams} B={B}: {best * 1e3:8.1f} us per forward, {B / best * 1e3:9.0f} samples/s",
              flush=True)
    torch.save({k: [t.cpu() for t in v[1]] for k, v in out.items()}, f"/tmp/infer_k3_{os.environ.get('SDUMC_K3', '1')}.pt")


if __name__ == "__main__":
    if os.environ.get("SDUMC_INFER_CHILD"):
        run()
    else:
        for k3 in ("1", "0"):
            env = dict(os.environ, SDUMC_K3=k3, SDUMC_INFER_CHILD="1")
            subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, check=True)
        import torch
        a, b = torch.load("/tmp/infer_k3_1.pt"), torch.load("/tmp/infer_k3_0.pt")
        worst = max(float((x - y).abs().max() / (y.abs().max() + 1e-30)) for s in a for x, y in zip(a[s], b[s]))
        print(f"max relative difference K3 vs two-kernel path over all outputs: {worst:.2e}")
