"""Are copies to / from SLICES of one pinned chunk asynchronous?  (host time of the copy_ calls against the transfer time)"""
import json, time, torch
dev = torch.device("cuda:0")
chunk = torch.empty(1 << 32, dtype=torch.uint8, pin_memory=True)
res = {"chunk_is_pinned": chunk.is_pinned()}
n = 316145664
sl = [chunk[o:o + n] for o in (0, 4096 + n, 2 * (4096 + n) + 8192)]
res["slices_pinned"] = [s.is_pinned() for s in sl]
g = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in sl]
st = torch.cuda.Stream()
for name, pairs in (("d2h", list(zip(sl, g))), ("h2d", list(zip(g, sl)))):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            for dst, src in pairs:
                dst.copy_(src, non_blocking=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res[f"{name}_{rep}"] = {"host_ms_enqueue": round(1e3 * (t1 - t0), 2), "total_ms": round(1e3 * (t2 - t0), 2)}
print(json.dumps(res))
