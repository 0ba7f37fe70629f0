"""Host-link probe for the activation-offload question: how much pinned host memory the GPU box gives, and what
D2H / H2D copies on side streams reach alone, together, and beside a saturating GEMM loop on the default stream.

    python tools/pcie_probe.py [--gib 24] [--chunk-mib 256]

Prints one JSON object.  Nothing here touches the product path."""
import argparse
import json
import resource
import time

import torch


def meminfo():
    out = {}
    for ln in open("/proc/meminfo"):
        k, v = ln.split(":")
        if k in ("MemTotal", "MemFree", "MemAvailable", "Hugepagesize", "HugePages_Total"):
            out[k] = v.strip()
    return out


def timed(fn, sync=True):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    if sync:
        torch.cuda.synchronize()
    return time.perf_counter() - t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=24.0)
    ap.add_argument("--chunk-mib", type=int, default=256)
    args = ap.parse_args()
    res = {"meminfo": meminfo(), "memlock": resource.getrlimit(resource.RLIMIT_MEMLOCK)}
    dev = torch.device("cuda:0")
    n_chunks = int(args.gib * 1024 / args.chunk_mib)
    chunk = args.chunk_mib << 20
    t = time.perf_counter()
    host = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(n_chunks)]
    res["pin_s_per_gib"] = (time.perf_counter() - t) / args.gib
    res["pinned_gib"] = args.gib
    gpu = [torch.empty(chunk, dtype=torch.uint8, device=dev) for _ in range(min(n_chunks, 16))]
    s_out, s_in = torch.cuda.Stream(), torch.cuda.Stream()

    def d2h(n):
        with torch.cuda.stream(s_out):
            for i in range(n):
                host[i % n_chunks].copy_(gpu[i % len(gpu)], non_blocking=True)

    def h2d(n):
        with torch.cuda.stream(s_in):
            for i in range(n):
                gpu[(i + 8) % len(gpu)].copy_(host[(i + n_chunks // 2) % n_chunks], non_blocking=True)

    n = n_chunks
    gb = n * chunk / 1e9
    d2h(4), h2d(4)
    res["d2h_alone_gbps"] = gb / timed(lambda: d2h(n))
    res["h2d_alone_gbps"] = gb / timed(lambda: h2d(n))
    dt = timed(lambda: (d2h(n), h2d(n)))
    res["duplex_each_gbps"] = gb / dt

    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        a @ b
    n_mm = 400
    t_mm = timed(lambda: [a @ b for _ in range(n_mm)])
    res["gemm_alone_ms"] = 1e3 * t_mm / n_mm

    def both():
        d2h(n), h2d(n)
        for _ in range(n_mm):
            a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ec0, ec1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s_out):
        ec0.record()
    e0.record()
    both()
    e1.record()
    with torch.cuda.stream(s_out):
        ec1.record()
    torch.cuda.synchronize()
    res["gemm_beside_copies_ms"] = e0.elapsed_time(e1) / n_mm
    res["d2h_beside_gemm_and_h2d_gbps"] = gb / (ec0.elapsed_time(ec1) / 1e3)
    # small pieces: 32 MiB copies (what a per-tensor offload would issue)
    small = 32 << 20
    def d2h_small(k):
        with torch.cuda.stream(s_out):
            for i in range(k):
                host[i % n_chunks][:small].copy_(gpu[i % len(gpu)][:small], non_blocking=True)
    k = 256
    res["d2h_32mib_gbps"] = k * small / 1e9 / timed(lambda: d2h_small(k))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
