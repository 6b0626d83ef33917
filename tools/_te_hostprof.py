import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import transformers_encoder as te, ops
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
B, H, T, E = 32, 8, 512, 1024
m = te.MultiheadAttention(E, H, attn_dropout=0.1).cuda().train()
x = torch.randn(T, B, E, device="cuda")
mask = te.buffered_future_mask(x)
with torch.no_grad():
    for rep in range(2):
        print("train nomask", t(lambda: m(x, x, x)))
        print("train mask", t(lambda: m(x, x, x, attn_mask=mask)))
        m.eval()
        print("eval mask", t(lambda: m(x, x, x, attn_mask=mask)))
        m.train()
    m.attn_dropout = 0.0
    print("train p=0 mask", t(lambda: m(x, x, x, attn_mask=mask)))
    m.attn_dropout = 0.1
    te.manual_seed(5, 0)
    print("train mask after reseed", t(lambda: m(x, x, x, attn_mask=mask)))
    print("site", te.dropout_stream.site)
