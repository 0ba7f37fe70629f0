"""Saved activations of remat-free layers parked in pinned HOST memory during the forward pass and brought back ahead of the backward.

Why: at the 9 s geometry 28 of 42 layers still re-materialise (14 ms each) because a layer that keeps everything holds 8.7 GiB of
HBM.  The host link of an MI355X box moves ~50 GB/s in EACH direction on the SDMA engines, beside the compute units and for 0.7 % of the HBM
bandwidth: what the first layers save (needed LAST in the backward) can leave the device while the later layers run, and come back while the
later layers' backward runs.  The arithmetic is untouched - the same tensors, the same bits, a round trip through host memory.  The reference
has no counterpart (it checkpoints every layer, ``configs/train/ttt-mlp/9s.toml:30-36``); this is a memory policy beside ``remat_free_layers``.

Mechanics (``torch.autograd.graph.saved_tensors_hooks`` around a free layer's forward):

* pack: a saved CUDA tensor of at least ``min_bytes`` that is not a parameter, until the layer's byte budget is spent, is offloaded WHOLE
  STORAGE-wise (views of one buffer - the q / k / v column blocks, an input shared by two nodes - travel once): the compute stream records an
  event, the D2H stream waits for it and copies the storage into a pinned slot (slot k of a step is re-used by every later step: the pinned
  pool is built once, in the first step).  The device copy is let go as soon as the copy's event has completed (polled at every pack; the
  host thread waits for the oldest copies when more than ``max_backlog_bytes`` are queued or the allocator's footprint exceeds
  ``soft_limit_bytes``, and for all of them at ``end_forward()``), so a backlog costs memory only while there is memory to spare.
* prefetch: the backward reaches layer i (a tensor hook on the layer's output gradient) -> the storages of layers i-1 .. i-``lookahead`` are
  allocated on the compute stream and filled on the H2D stream (which waits for the compute stream's position at that moment - the block may
  be in use until then - and for the slot's D2H copy).
* unpack: the compute stream waits for the storage's H2D event; the saved view is rebuilt over the restored storage.  A storage whose device
  copy is still held (the copy out has not finished, or nothing pressed for memory) is simply handed back.

On a CPU tensor (the unit tests) the same bookkeeping runs with synchronous copies and no streams."""
from __future__ import annotations

import collections
import contextlib
import time

import torch


def host_room_gib():
    """GiB of host memory this process may still take: MemAvailable, and the cgroup's limit minus its usage where there is one (pinned memory
    cannot be paged out and is charged to the box; None if nothing can be read)."""
    room = []
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                room.append(int(ln.split()[1]) / 2 ** 20)
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v.isdigit() and int(v) < 1 << 60:
                room.append((int(v) - int(open(cur).read().strip())) / 2 ** 30)
        except (OSError, ValueError):
            pass
    return min(room) if room else None


class _Stor:
    """One offloaded storage of one step."""
    __slots__ = ("layer", "nbytes", "slot", "dev", "d2h_done", "restored", "h2d_done", "device", "queued")

    def __init__(self, layer, nbytes, slot, dev, device):
        self.layer, self.nbytes, self.slot, self.dev, self.device = layer, nbytes, slot, dev, device
        self.d2h_done = self.restored = self.h2d_done = None
        self.queued = False                           # batch mode: the copy out has not been issued yet


class _View:
    """What the pack hook returns for an offloaded tensor: the storage + the geometry of the saved view."""
    __slots__ = ("stor", "dtype", "size", "stride", "offset")

    def __init__(self, stor, t):
        self.stor, self.dtype, self.size, self.stride, self.offset = stor, t.dtype, tuple(t.shape), tuple(t.stride()), t.storage_offset()


def _bytes_of(storage, device):
    """the storage as a flat uint8 tensor (an alias: it keeps the storage alive)"""
    return torch.empty(0, dtype=torch.uint8, device=device).set_(storage, 0, (storage.nbytes(),), (1,))


class HostOffload:
    def __init__(self, bytes_per_layer: int, layers: int | None = None, min_bytes: int = 96 << 20, lookahead: int = 2,
                 soft_limit_bytes: int | None = None, max_storage_ratio: float = 4.0, pin: bool = True, park_kept: bool = False,
                 max_backlog_bytes: int | None = None, max_pinned_bytes: int = 160 << 30):
        self.bytes_per_layer, self.layers, self.min_bytes, self.lookahead = int(bytes_per_layer), layers, int(min_bytes), int(lookahead)
        self.soft_limit_bytes, self.max_storage_ratio, self.pin = soft_limit_bytes, max_storage_ratio, pin
        self.max_backlog_bytes = max_backlog_bytes
        self.max_pinned_bytes = int(max_pinned_bytes)  # cap of the pinned pool; what does not fit stays on the device
        room = host_room_gib() if pin else None
        if room is not None:                           # ... and never more than 60 % of what the host has to spare when the pool is created
            self.max_pinned_bytes = min(self.max_pinned_bytes, int(0.6 * room * 2 ** 30))
        # batch: the copies out of a layer are issued together behind the layer's forward, behind ONE event of the compute stream (the runtime
        # picks the SDMA engine of a copy when it is queued; a marker between two copies makes it choose again, and with the first engine busy it
        # takes the next free one - copies that trickle in one by one end up spread over engines that are slower on the host link)
        self.batch = True
        # one_stream: copies out and in share ONE side stream (measured: with a stream each, the fourth and fifth stream of the process alias
        # with the scan's / the backward's side streams on the 4 hardware queues and the step loses 12 %; with 8 queues the baseline loses 4.6 %)
        self.one_stream = True
        # end_forward() waits for every copy out (False, the default: it lets go of what is complete and the rest follows as the backward
        # proceeds - the host thread of the training step has no slack: every wait of it is idle time of the device)
        self.blocking_end = False
        self._batch: list = []
        self._scope_depth = 0
        self.park_kept = park_kept                    # the kernel outputs re-materialised layers keep (remat_cache) wait in host memory as well
        self._slots: list[torch.Tensor] = []          # pinned host buffers, slot k = the k-th storage a step offloads
        self.chunk_bytes, self._chunk, self._chunk_used, self._pinned_total = 1 << 32, None, 0, 0
        self._streams = None
        self.stats = collections.Counter()
        self.trace = None                             # DEBUG: a list -> (kind, bytes, start event, end event) of every copy and compute-stream wait
        self._new_step()

    # ---- per-step state ------------------------------------------------------------------------------------------------------
    def _new_step(self):
        self._by_layer: dict[int, list[_Stor]] = collections.defaultdict(list)
        self._by_ptr: dict[int, _Stor] = {}           # device data_ptr of a storage that is offloaded in this step (while it is alive)
        self._pending: collections.deque[_Stor] = collections.deque()     # copies out whose device copy is still held
        self._next_slot = 0
        self._spent = 0
        self._cur_layer = None

    def begin_step(self):
        """Call once in front of a forward pass (``DiffusionTransformer.forward`` does): forgets the previous step's handles."""
        self._new_step()

    def applies(self, layer: int) -> bool:
        return self.bytes_per_layer > 0 and (self.layers is None or layer < self.layers)

    # ---- streams (CUDA only) -----------------------------------------------------------------------------------------------------
    def _st(self, device):
        """(copy-out stream, copy-in stream).  ``one_stream``: the same stream for both directions - the copies out belong to the forward, the
        copies in to the backward, and a process has few hardware queues (4 by default: the compute stream, the scan's side stream, the
        backward's side stream ... a stream more aliases with one of them, and its copies then sit in front of that stream's kernels)."""
        if self._streams is None:
            a = torch.cuda.Stream(device=device)
            self._streams = (a, a if self.one_stream else torch.cuda.Stream(device=device))
        return self._streams

    def _ev(self, stream, rec=None):
        """an event on ``stream``; DEBUG tracing: timed events, and ``rec = (kind, bytes, start event)`` is logged with this one as its end"""
        if self.trace is None:
            return stream.record_event() if rec is not None else None
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        if rec is not None:
            self.trace.append((*rec, e))
        return e

    def trace_summary(self):
        """DEBUG (after a device synchronisation): per kind the device time between start and end events, bytes and GB/s"""
        out = {}
        for kind, nbytes, e0, e1 in self.trace or ():
            d = out.setdefault(kind, {"n": 0, "ms": 0.0, "gib": 0.0})
            d["n"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["gib"] += nbytes / 2 ** 30
        for d in out.values():
            d["gbps"] = round(d["gib"] * 2 ** 30 / 1e9 / max(d["ms"] * 1e-3, 1e-9), 1)
            d["ms"], d["gib"] = round(d["ms"], 1), round(d["gib"], 2)
        return out

    def _slot(self, k, nbytes):
        """slot k of the pinned pool, at least ``nbytes`` long.  The pool is a list of pinned chunks of a power-of-two size (torch's pinned
        allocator rounds every request up to one: a 316-MB slot of its own would pin 512 MB) that slots are carved out of; a slot that turns out
        too small in a later step (another geometry) is carved again and its old bytes stay unused."""
        if k < len(self._slots) and self._slots[k].numel() >= nbytes:
            return self._slots[k]
        need = (nbytes + 4095) & ~4095
        if self._chunk is None or self._chunk_used + need > self._chunk.numel():
            size = self.chunk_bytes
            while size < need:
                size *= 2
            if self._pinned_total + size > self.max_pinned_bytes:
                # The pool is full: the caller keeps the tensor on the device.  (Pinned memory cannot be paged out: a pool beyond what the box
                # gives the process takes the box down with it - round 6, call HO14, a setting that would have pinned 560 GiB.)
                self.stats["pool_full_refusals"] += 1
                return None
            self._chunk, self._chunk_used = torch.empty(size, dtype=torch.uint8, pin_memory=self.pin and torch.cuda.is_available()), 0
            self.stats["pinned_bytes"] += size
            self._pinned_total += size
        buf = self._chunk[self._chunk_used:self._chunk_used + need]
        self._chunk_used += need
        if k < len(self._slots):
            self._slots[k] = buf
        else:
            assert k == len(self._slots)
            self._slots.append(buf)
        return buf

    # ---- forward -------------------------------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def layer(self, idx: int):
        """Context for the forward of free layer ``idx``: what it saves is offloaded within the layer's byte budget."""
        if not self.applies(idx):
            yield
            return
        prev, self._cur_layer, self._spent = self._cur_layer, idx, 0
        try:
            with self.scope(), torch.autograd.graph.saved_tensors_hooks(self._pack, self._unpack):
                yield
        finally:
            self._cur_layer = prev

    @contextlib.contextmanager
    def scope(self):
        """Batch mode: the copies out queued inside are issued together at the exit (a layer's forward: free layers through ``layer()``, the
        forward of a re-materialised layer whose kept outputs are parked through this)."""
        self._scope_depth += 1
        try:
            yield
        finally:
            self._scope_depth -= 1
            if self._scope_depth == 0:
                self._flush()

    def _pack(self, t):
        if (not isinstance(t, torch.Tensor) or isinstance(t, torch.nn.Parameter) or (t.requires_grad and t.is_leaf) or t.is_sparse
                or t.numel() == 0 or t.device.type not in ("cuda", "cpu")):
            return t                                                  # (parameters and other trainable leaves live on the device anyway)
        nbytes_view = t.numel() * t.element_size()
        if nbytes_view < self.min_bytes:
            return t
        storage = t.untyped_storage()
        ptr, nbytes = storage.data_ptr(), storage.nbytes()
        hit = self._by_ptr.get(ptr)
        if hit is not None and hit.nbytes == nbytes:                  # another view of a storage that already travels
            self.stats["views_shared"] += 1
            return _View(hit, t)
        if nbytes > self.max_storage_ratio * nbytes_view or self._spent + nbytes > self.bytes_per_layer:
            return t                                                  # a slice of a big buffer / the layer's budget is spent
        stor = self._send(self._cur_layer, t)
        if stor is None:
            return t
        self._spent += nbytes
        if stor.dev is not None:                                      # (a throttled copy may be complete, and its device copy let go, already)
            self._by_ptr[ptr] = stor
        return _View(stor, t)

    def _send(self, layer, t) -> _Stor | None:
        """queues the copy out of ``t``'s whole storage into the step's next pinned slot (None: the pinned pool is at its cap)"""
        self._release_done()
        storage = t.untyped_storage()
        nbytes = storage.nbytes()
        host = self._slot(self._next_slot, nbytes)
        if host is None:
            return None
        host = host[:nbytes]
        stor = _Stor(layer, nbytes, self._next_slot, _bytes_of(storage, t.device), t.device)
        if t.is_cuda and self.batch and self._scope_depth > 0:
            stor.queued = True
            self._batch.append((stor, host))
            self._pending.append(stor)
        elif t.is_cuda:
            self._issue([(stor, host)])
            self._pending.append(stor)
            self._throttle()
        else:
            host.copy_(stor.dev)                                      # (CPU tensors - the unit tests: the same bookkeeping, copies at once)
            self._pending.append(stor)
        self._next_slot += 1
        self._by_layer[layer].append(stor)
        self.stats["offloaded_bytes"] += nbytes
        self.stats["offloaded_storages"] += 1
        return stor

    def _issue(self, items):
        """the copies out of ``items`` = [(storage record, pinned slot)], behind ONE event of the compute stream"""
        dev = items[0][0].device
        main = torch.cuda.current_stream(dev)
        out, _ = self._st(dev)
        t0 = time.perf_counter()
        out.wait_event(main.record_event())
        with torch.cuda.stream(out):
            for stor, host in items:
                e0 = self._ev(out)
                host.copy_(stor.dev, non_blocking=True)
                stor.d2h_done = self._ev(out, ("d2h", stor.nbytes, e0))
                stor.queued = False
        self.stats["host_s_enqueue_out"] += time.perf_counter() - t0

    def _flush(self):
        if self._batch:
            items, self._batch = self._batch, []
            self._issue(items)
            self._throttle()

    # ---- kernel outputs a re-materialised layer keeps (ttt_amd/infra/remat_cache.py) ------------------------------------------------
    def park(self, layer: int, tensors):
        """The kept kernel outputs of a re-materialised layer (attention outputs, scan outputs + checkpoints, the MLP output) wait in host
        memory too: returns one handle per tensor (the tensor itself where it is small or does not own its storage) for ``unpark``."""
        out = []
        for t in tensors:
            nb = t.numel() * t.element_size()
            if nb < self.min_bytes or t.storage_offset() != 0 or t.untyped_storage().nbytes() != nb:
                out.append(t)                                         # (small, or not the sole - possibly permuted - owner of its storage)
            else:
                stor = self._send(layer, t)
                out.append(t if stor is None else _View(stor, t))
        return tuple(out)

    def unpark(self, handles):
        return tuple(self._unpack(h) for h in handles)

    def _release_done(self):
        """lets go of the device copies whose copy out has completed (oldest first: the D2H stream is in order)"""
        while self._pending and not self._pending[0].queued and (self._pending[0].dev is None or self._pending[0].d2h_done is None
                                                                 or self._pending[0].d2h_done.query()):
            self._drop_dev(self._pending.popleft())

    def _drop_dev(self, s):
        if s.dev is not None and s.restored is None:
            self._by_ptr.pop(s.dev.untyped_storage().data_ptr(), None)   # (the address may be handed out again)
            s.dev = None

    def _backlog(self):
        return sum(s.nbytes for s in self._pending if s.dev is not None)

    def _throttle(self):
        """The host thread runs ahead of the device, and the copies out start when the DEVICE reaches them: the device copies of everything
        the host has queued are 'allocated' as far as torch's allocator knows.  Two bounds keep that from running into the memory cap: the
        bytes queued (``max_backlog_bytes``: the host waits for the oldest copy - it stays that far ahead of the D2H stream, which is a few
        layers of compute: the device does not run dry) and, above ``soft_limit_bytes`` of allocated memory, no backlog at all."""
        if self.soft_limit_bytes is None and self.max_backlog_bytes is None:
            return
        self._flush()
        t0 = time.perf_counter()
        while self._pending and ((self.max_backlog_bytes is not None and self._backlog() > self.max_backlog_bytes)
                                 or (self.soft_limit_bytes is not None and torch.cuda.memory_allocated() > self.soft_limit_bytes)):
            s = self._pending.popleft()
            if s.dev is not None:
                if s.d2h_done is not None:
                    s.d2h_done.synchronize()
                self.stats["throttle_waits"] += 1
                self._drop_dev(s)
        self.stats["host_s_throttle"] += time.perf_counter() - t0

    def end_forward(self):
        """Every copy out finished, every device copy let go (the host thread waits; the compute stream does not): call behind the last
        layer's forward, where the step's memory peak is."""
        self._flush()
        if not self.blocking_end:
            self._release_done()
            return
        t0 = time.perf_counter()
        while self._pending:
            s = self._pending.popleft()
            if s.dev is not None:
                if s.d2h_done is not None:
                    s.d2h_done.synchronize()
                self._drop_dev(s)
        self.stats["host_s_end_forward"] += time.perf_counter() - t0

    # ---- backward ------------------------------------------------------------------------------------------------------------------
    def backward_reaches(self, idx: int):
        """The backward is about to run layer ``idx`` (hook on the gradient of its output): fetch the layers below it, let go of those above."""
        self._release_done()
        for j in [k for k in self._by_layer if k > idx + 1]:
            for s in self._by_layer.pop(j):
                s.restored = s.dev = None
        for j in range(idx, idx - self.lookahead - 1, -1):
            for s in self._by_layer.get(j, ()):
                self._fetch(s)

    def _fetch(self, s):
        if s.restored is not None:
            return
        if s.dev is not None:                                        # the device copy never left: hand it back
            s.restored, s.h2d_done = s.dev, None
            self.stats["kept_on_device"] += 1
            return
        host = self._slots[s.slot][:s.nbytes]
        if s.device.type == "cuda":
            t0 = time.perf_counter()
            main = torch.cuda.current_stream(s.device)
            _, inn = self._st(s.device)
            buf = torch.empty(s.nbytes, dtype=torch.uint8, device=s.device)
            inn.wait_event(main.record_event())                       # the block may be in use on the compute stream up to here
            if s.d2h_done is not None:
                inn.wait_event(s.d2h_done)
            with torch.cuda.stream(inn):
                e0 = self._ev(inn)
                buf.copy_(host, non_blocking=True)
                s.h2d_done = self._ev(inn, ("h2d", s.nbytes, e0))
            buf.record_stream(inn)
            self.stats["host_s_enqueue_in"] += time.perf_counter() - t0
        else:
            buf = host.clone()
        s.restored = buf
        self.stats["fetched_bytes"] += s.nbytes

    def _unpack(self, x):
        if not isinstance(x, _View):
            return x
        s = x.stor
        if s.restored is None:
            self.stats["late_fetches"] += 1
            self._fetch(s)
        if s.h2d_done is not None:
            main = torch.cuda.current_stream(s.device)
            e0 = self._ev(main)
            main.wait_event(s.h2d_done)
            if self.trace is not None:
                self._ev(main, ("wait", s.nbytes, e0))
        return torch.empty(0, dtype=x.dtype, device=s.device).set_(s.restored.untyped_storage(), x.offset, x.size, x.stride)
