"""bench.py's N > 1 control flow on two gloo ranks with a FAKE device (no GPU): the sizing of ``remat_free_layers``, the warm-up
and the timed region are one function (``bench.size_warm_and_time``) that every rank must walk identically - a rank that takes
another branch leaves the others in a collective.  Checked here: ranks with DIFFERENT memory head-room agree on the minimum;
an out-of-memory error of ONE rank in the warm-up makes BOTH back off (and the region is warmed again); an out-of-memory error
inside the timed region of a multi-rank run is fatal on that rank (it cannot be recovered while the others wait); the
single-rank path backs off and times the region again.  Reference: the 8-rank launch scripts/train_singlenode.sh:25-38."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FakeOOM(RuntimeError):
    pass


class FakeDevice:
    """Memory model: a step with n un-checkpointed layers peaks at base + n * per_layer bytes; above total_memory it raises."""
    oom = FakeOOM
    num_layers = 42
    probe_layers = 1

    def __init__(self, world, total, base, per_layer, warm_oom_at=None, timed_oom_at=None, thrash_above=None):
        self.world, self.total_memory, self.base, self.per_layer = world, total, base, per_layer
        self.thrash_above, self.retries = thrash_above, 0       # timed steps with more free layers make the allocator retry
        self.warm_oom_at, self.timed_oom_at = warm_oom_at, timed_oom_at      # n_free values at which the allocator "fragments"
        self.n_free, self.peak, self.in_timed, self.log = 0, 0, False, []
        self.timed_regions = 0

    def step(self):
        need = self.base + self.n_free * self.per_layer
        if need > self.total_memory:
            raise FakeOOM("probe / warm-up does not fit")
        if not self.in_timed and self.warm_oom_at is not None and self.n_free == self.warm_oom_at:
            raise FakeOOM("fragmentation in the warm-up")        # (persistent: this setting never fits on this rank)
        if self.in_timed and self.timed_oom_at is not None and self.n_free == self.timed_oom_at:
            self.timed_oom_at = None
            raise FakeOOM("fragmentation in the timed region")
        self.peak = max(self.peak, need)
        if self.in_timed and self.thrash_above is not None and self.n_free > self.thrash_above:
            self.retries += 1
        return torch.tensor(1.0)

    def alloc_retries(self): return self.retries
    def reduce_keep(self): return False
    def set_free_layers(self, n): self.n_free = n
    def reset_peak(self): self.peak = 0
    def max_allocated(self): return self.peak
    def synchronize(self): pass
    def release(self): self.log.append("release")

    def all_reduce_min(self, v):
        t = torch.tensor([v])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t)

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def before_timed(self, n_free):
        self.in_timed = True
        self.timed_regions += 1

    def after_timed(self): self.in_timed = False


def _worker(rank, world, port, case, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    GB = 1 << 30
    try:
        if case == "min_over_ranks":          # rank 1 has less head-room (another process on its GPU): both take ITS answer
            hk = FakeDevice(world, 288 * GB, (60 if rank == 0 else 100) * GB, 5 * GB)
            n, dt, loss = bench.size_warm_and_time(hk.step, hk, "auto", 1, 2, world)
            out[rank] = ("ok", n, hk.timed_regions)
        elif case == "warmup_oom_on_one_rank":
            hk = FakeDevice(world, 288 * GB, 60 * GB, 5 * GB, warm_oom_at=(34 if rank == 1 else None))
            n, dt, loss = bench.size_warm_and_time(hk.step, hk, "auto", 1, 2, world)
            out[rank] = ("ok", n, hk.timed_regions, hk.log.count("release"))
        elif case == "timed_oom_is_fatal":
            hk = FakeDevice(world, 288 * GB, 60 * GB, 5 * GB, timed_oom_at=(34 if rank == 1 else None))
            if rank == 1:
                with pytest.raises(FakeOOM):
                    bench.size_warm_and_time(hk.step, hk, "auto", 1, 2, world)
                out[rank] = ("raised",)
            else:
                out[rank] = ("skipped",)      # (rank 0 would wait in the closing barrier: a real run dies with rank 1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["min_over_ranks", "warmup_oom_on_one_rank"])
def test_two_ranks_walk_the_same_control_flow(case):
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29650 + (hash(case) % 200)
    mp.spawn(_worker, args=(2, port, case, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0[0] == r1[0] == "ok"
    assert r0[1] == r1[1], (r0, r1)                    # the same remat_free_layers on both ranks
    GB = 1 << 30
    if case == "min_over_ranks":
        # 80 % of 288 GB; rank 1: (230.4 - 100) / 5 = 26 layers, rank 0 alone would have taken 34
        assert r0[1] == 26 and r0[2] == r1[2] == 1
    else:
        # both computed 34; rank 1's warm-up runs out of memory there -> BOTH back off by max(1, 34 // 10) = 3 and warm up again;
        # the refinement (measured footprint at 31 layers says 34 would fit) stays below the setting that failed: 33, which fits
        assert r0[1] == 33 and r0[2] == r1[2] == 1 and r0[3] == r1[3] == 1


def test_single_rank_backs_off_and_times_again():
    import bench
    GB = 1 << 30
    hk = FakeDevice(1, 288 * GB, 60 * GB, 5 * GB, timed_oom_at=38)      # 88 % of 288 GB: (253.4 - 60) / 5 = 38 layers
    n, dt, loss = bench.size_warm_and_time(hk.step, hk, "auto", 1, 3, 1)
    # first region abandoned, 38 - 3 = 35 layers warmed; the refinement may climb again but stays below the setting that failed
    assert n == 37 and hk.timed_regions == 2
    # an explicit setting is never second-guessed: out of memory is fatal
    hk = FakeDevice(1, 288 * GB, 60 * GB, 5 * GB, timed_oom_at=10)
    with pytest.raises(FakeOOM):
        bench.size_warm_and_time(hk.step, hk, "10", 1, 3, 1)


def test_allocator_retries_inside_the_timed_region_cost_layers_not_the_measurement():
    import bench
    GB = 1 << 30
    hk = FakeDevice(1, 288 * GB, 60 * GB, 5 * GB, thrash_above=33)      # fits 38 by the arithmetic, thrashes above 33
    n, dt, loss = bench.size_warm_and_time(hk.step, hk, "auto", 1, 2, 1)
    assert n <= 33 and hk.timed_regions >= 2
    hk = FakeDevice(1, 288 * GB, 60 * GB, 5 * GB, thrash_above=33)      # an explicit setting is measured as it is
    n, dt, loss = bench.size_warm_and_time(hk.step, hk, "36", 1, 2, 1)
    assert n == 36 and hk.timed_regions == 1


def test_keeping_policy_is_reduced_when_even_full_rematerialisation_does_not_fit():
    """30 s geometry: with every layer re-materialised, the kept kernel outputs alone exceed the device; the policy drops them
    kind by kind before giving up (one GPU)."""
    import bench
    GB = 1 << 30

    class Dev(FakeDevice):
        keep = ["attn", "scan"]

        def step(self):
            if not self.in_timed and self.keep == ["attn", "scan"]:
                raise FakeOOM("kept scan outputs do not fit")
            return super().step()

        def reduce_keep(self):
            if not self.keep:
                return False
            self.keep = self.keep[:-1]
            return True

    hk = Dev(1, 288 * GB, 200 * GB, 20 * GB)
    n, dt, loss = bench.size_warm_and_time(hk.step, hk, "auto", 1, 2, 1)
    assert hk.keep == ["attn"] and n == 2 and hk.timed_regions == 1        # (253.4 - 200) / 20 = 2 remat-free layers


def test_timed_region_oom_on_a_multi_rank_run_is_fatal_on_that_rank():
    """(one process playing rank 1 of 2: the point is the branch it takes, no peer is needed before the failure)"""
    import bench
    GB = 1 << 30

    class Solo(FakeDevice):
        def all_reduce_min(self, v): return v
        def barrier(self): pass

    hk = Solo(2, 288 * GB, 60 * GB, 5 * GB, timed_oom_at=34)
    with pytest.raises(FakeOOM):
        bench.size_warm_and_time(hk.step, hk, "auto", 1, 2, 2)
