"""DEBUG: compare the step-0 intermediates dumped by the revision-2 forward kernel with a host recomputation."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import ttt_oracle as O
import test_kernels_gpu as T

e = T.ext()
d = T.round_acts(O.make_inputs("mlp", 1, 1, 2, 64, 64, seed=55), torch.bfloat16)
# make biases / LN params non-trivial
g = torch.Generator().manual_seed(3)
d["b1"] = 0.05 * torch.randn(d["b1"].shape, generator=g)
d["b2"] = 0.05 * torch.randn(d["b2"].shape, generator=g)
d["ln_w"] = 1 + 0.1 * torch.randn(d["ln_w"].shape, generator=g)
d["ln_b"] = 0.1 * torch.randn(d["ln_b"].shape, generator=g)
buf = torch.zeros(120000, device="cuda")
e.debug_dump(buf)
out, cks, _ = T.run_mlp(e, d, 1, torch.bfloat16, impl="mfma", bwd_impl="generic")
e.debug_dump(None)
D = buf.cpu().double()

f = lambda k: d[k].double()
K, Q, V = f("XK")[0, 0, 0], f("XQ")[0, 0, 0], f("XV")[0, 0, 0]
eta = f("eta")[0, 0, 0, -1]                       # [CS]
W1, b1, W2, b2 = f("W1")[0], f("b1")[0], f("W2")[0], f("b2")[0]
gam, bet = f("ln_w")[0], f("ln_b")[0]
A, C = O.GELU_A, O.GELU_C
def gelu(x):
    return 0.5 * x * (1 + torch.tanh(A * x * (1 + C * x * x)))
def dgelu(x):
    t = torch.tanh(A * x * (1 + C * x * x))
    return 0.5 * x * ((1 - t * t) * (A + 3 * A * C * x * x)) + 0.5 * (1 + t)
Z1 = K @ W1 + b1
X2 = gelu(Z1)
Z2 = X2 @ W2 + b2
mu = Z2.mean(-1, keepdim=True); var = Z2.var(-1, keepdim=True, unbiased=False); std = torch.sqrt(var + 1e-8)
xh = (Z2 - mu) / std
go = gam * xh + bet - (V - K)
gx = go * gam
gZ2 = (64 * gx - gx.sum(-1, keepdim=True) - xh * (gx * xh).sum(-1, keepdim=True)) / (64 * std)
Gs = -eta[:, None] * gZ2
gZ1s = (Gs @ W2.T) * dgelu(Z1)
W1n = W1 + K.T @ gZ1s
b1n = b1 + gZ1s.sum(0, keepdim=True)
W2n = W2 + X2.T @ Gs
b2n = b2 + Gs.sum(0, keepdim=True)
Z1b = Q @ W1n + b1n
X2b = gelu(Z1b)
Z2b = X2b @ W2n + b2n
rel = lambda a, b: float((a - b).norm() / b.norm())
print("X2    ", rel(D[0:16384].view(256, 64), X2.T))
print("Z2    ", rel(D[16384:20480].view(64, 64), Z2))
for ww in range(4):
    ref = X2[:, 64 * ww:64 * ww + 64] @ W2[64 * ww:64 * ww + 64]
    got = D[45376 + ww * 4096:45376 + (ww + 1) * 4096].view(64, 64)
    err = (got - ref).abs()
    blk = err.view(2, 32, 2, 32).amax(dim=(1, 3))
    print(f"partial {ww}: rel {rel(got, ref):.4f}; max-abs err per (t-half, f-half) block {blk.tolist()}; ref absmax {float(ref.abs().max()):.3f}")
    if ww == 0:
        bad = (err > 0.05 * ref.abs().max()).nonzero()
        print("   first bad (t,f):", bad[:12].tolist(), " count", len(bad))
        e8 = err.view(8, 8, 8, 8).amax(dim=(1, 3))
        print("   8x8 block max err:\n", (e8 * 1000).round().int())
print("mu    ", rel(D[60000:60064], mu[:, 0]), " rstd", rel(D[60064:60128], 1 / std[:, 0]))
print("Gs    ", rel(D[20480:24576].view(64, 64), Gs))
print("b1'   ", rel(D[24576:24832], b1n[0]), " delta", rel(D[24576:24832] - b1[0], b1n[0] - b1[0]))
print("b2'   ", rel(D[24832:24896], b2n[0]), " delta", rel(D[24832:24896] - b2[0], b2n[0] - b2[0]))
print("W1'   ", rel(D[65536:81920].view(64, 256), W1n), " delta", rel(D[65536:81920].view(64, 256) - W1, W1n - W1))
print("W2'   ", rel(D[81920:98304].view(256, 64), W2n), " delta", rel(D[81920:98304].view(256, 64) - W2, W2n - W2))
print("W2T'  ", rel(D[98304:114688].view(256, 64), W2n), " delta", rel(D[98304:114688].view(256, 64) - W2, W2n - W2))
print("X2b   ", rel(D[28992:45376].view(256, 64), X2b.T))
print("Z2b   ", rel(D[24896:28992].view(64, 64), Z2b))
ro, _, _ = T.oracle_on(d, 1, "mlp")
print("out step0", rel(out[0, 0, 0].cpu().double(), ro[0, 0, 0]), " step1", rel(out[0, 0, 1].cpu().double(), ro[0, 0, 1]))
