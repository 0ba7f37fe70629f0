"""Which of the training step's kernels slows the copies out?  (tools/pcie_probe2.py: 55 - 57 GB/s beside GEMM / elementwise loops; inside the
forward of the step 13 - 25 GB/s.)  D2H of 48 x 111 MB beside: the TTT-MLP forward scan (pair of workgroups per head, flag polling), the
attention forward, the scan on a side stream beside GEMMs, and with every copy made to wait for an event of the compute stream."""
import json, sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ttt-video-dit_amd"))
import test_time_training as ext
from ttt_amd.models.ssm.mlp_tk import TkMLP
ext.load_library()
dev = torch.device("cuda:0")
n, chunk = 48, 110911488
pool = torch.empty(1 << 33, dtype=torch.uint8, pin_memory=True)
host = [pool[i * (chunk + 4096): i * (chunk + 4096) + chunk] for i in range(n)]
gpu = [torch.empty(chunk, dtype=torch.uint8, device=dev) for _ in range(8)]
so, side = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()
gen = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=gen)
B, NH, NC, CS, F, G = 1, 48, 804, 64, 64, 16
XQ = torch.nn.functional.normalize(rn(B, NH, NC, CS, F), dim=-1).bfloat16(); XK = torch.nn.functional.normalize(rn(B, NH, NC, CS, F), dim=-1).bfloat16()
XV = rn(B, NH, NC, CS, F).bfloat16(); eta = (0.1 * torch.sigmoid(rn(B, NH, NC, 1, CS)) / (F * CS)).bfloat16()
ln_w, ln_b = torch.ones(NH, F, device=dev), torch.zeros(NH, F, device=dev)
W1, b1, W2, b2 = 0.02 * rn(NH, F, 256), torch.zeros(NH, 1, 256, device=dev), 0.02 * rn(NH, 256, F), torch.zeros(NH, 1, F, device=dev)
ex = lambda p: p.unsqueeze(0).expand(B, *p.shape)
S = 18052
mk = lambda: torch.randn(1, S, 48, 64, device=dev, generator=gen).bfloat16().transpose(1, 2)
q, k, v = mk(), mk(), mk()
ao = torch.empty(1, S, 48, 64, device=dev, dtype=torch.bfloat16).transpose(1, 2); lse = torch.empty(1, 48, S, device=dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)

def scan():
    with torch.no_grad():
        TkMLP.apply(ln_w, ln_b, ex(W1), ex(b1), ex(W2), ex(b2), XQ, XV, XK, eta, G)

def load(kind):
    if kind == "scan":
        scan()
    elif kind == "attn":
        for _ in range(2):
            ext.attn_forward(q, k, v, ao, lse, 0.125)
    elif kind == "scan_side+gemm":
        side.wait_stream(main)
        with torch.cuda.stream(side):
            scan()
        for _ in range(4):
            a @ b
        main.wait_stream(side)
    elif kind == "gemm":
        for _ in range(5):
            a @ b

res = {}
for kind, dep in (("gemm", False), ("scan", False), ("attn", False), ("scan_side+gemm", False), ("gemm", True), ("scan_side+gemm", True)):
    load(kind); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0.record()
    load(kind)
    with torch.cuda.stream(so):
        e0.record()
    for i in range(n):
        if dep:
            so.wait_event(main.record_event())
        with torch.cuda.stream(so):
            host[i].copy_(gpu[i % 8], non_blocking=True)
        if i % 2 == 1:
            load(kind)
    with torch.cuda.stream(so):
        e1.record()
    for _ in range(4):
        load(kind)
    m1.record()
    torch.cuda.synchronize()
    res[f"d2h_beside_{kind}{'_with_event_deps' if dep else ''}"] = {"gbps": round(n * chunk / 1e9 / (e0.elapsed_time(e1) * 1e-3), 1), "copy_ms": round(e0.elapsed_time(e1), 1), "load_ms": round(m0.elapsed_time(m1), 1)}
print(json.dumps(res))
