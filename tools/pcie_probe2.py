"""D2H / H2D rate beside different kinds of compute on the default stream (tools/pcie_probe.py found 57 GB/s alone, 48 duplex; beside the
training step's forward the copies out reach 13 - 25 GB/s): a GEMM loop (matrix-pipe-bound), an elementwise loop (HBM-bound), both."""
import json, time, torch
dev = torch.device("cuda:0")
n, chunk = 48, 316145664
pool = torch.empty(1 << 34, dtype=torch.uint8, pin_memory=True)
host = [pool[i * (chunk + 4096): i * (chunk + 4096) + chunk] for i in range(n)]
gpu = [torch.empty(chunk, dtype=torch.uint8, device=dev) for _ in range(8)]
so, si = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
x = torch.randn(1 << 29, device=dev, dtype=torch.bfloat16)       # 1 GiB

def copies(direction):
    st = so if direction == "d2h" else si
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record()
        for i in range(n):
            if direction == "d2h":
                host[i].copy_(gpu[i % 8], non_blocking=True)
            else:
                gpu[i % 8].copy_(host[i], non_blocking=True)
        e1.record()
    return e0, e1

def load(kind, iters):
    for _ in range(iters):
        if kind in ("gemm", "both"):
            a @ b
        if kind in ("elementwise", "both"):
            x.add_(1.0)

res = {}
for kind, iters in (("none", 0), ("gemm", 700), ("elementwise", 2500), ("both", 600)):
    for direction in ("d2h", "h2d"):
        torch.cuda.synchronize()
        m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m0.record()
        load(kind, iters // 10)                  # the copies start inside the load
        e0, e1 = copies(direction)
        load(kind, iters)
        m1.record()
        torch.cuda.synchronize()
        res[f"{direction}_beside_{kind}"] = {"gbps": round(n * chunk / 1e9 / (e0.elapsed_time(e1) * 1e-3), 1), "copy_ms": round(e0.elapsed_time(e1), 1), "load_ms": round(m0.elapsed_time(m1), 1)}
print(json.dumps(res))
