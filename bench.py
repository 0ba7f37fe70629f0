#!/usr/bin/env python
"""Benchmark: CogVideoX-5B + TTT-MLP training step (forward + backward + AdamW), BASELINE.json metric
"DiT+TTT fwd/bwd video-tokens/sec".

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (no launcher: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): the configuration BASELINE.json's metric is quoted on - the ("5B", "9sec") preset (42 layers,
D=3072, 48 heads, TTT-MLP with CS=64, 37 latent frames of 30x45 tokens in 3 interleaved scenes + 3 x 502 text tokens =
51 456 tokens per sample; it fits one 288-GB MI355X), adapter "qkvo" as in the reference's configs/train/ttt-mlp/9s.toml
(attention / TTT projections, TTT inner parameters and gates train), bf16 compute with fp32 master weights - via FSDP2 on
several GPUs (same wrapping as the reference's apply_fsdp), via the same arithmetic without FSDP2's per-parameter copies on
one (--fsdp) -, local batch 1 per GPU (weak scaling), synthetic latents / text embeddings, random-init weights.  One step =
zero_grad, loss = CogVideoX(vid, text).mean(), backward, clip_grad_norm, fused AdamW step.  ``--video-length 3sec`` is
BASELINE configs[1] (adapter "sft", configs/train/ttt-mlp/3s.toml), 30sec / 63sec the later stages.

MI355X-first memory policy (same arithmetic as the reference's settings, which remain available as flags):
  --remat-free-layers auto   leading layers that keep their activations instead of being re-materialised in backward,
                             sized to the GPU's 288 GB with untimed probe steps (0 = the reference's 80-GB setting)
  (default)                  FSDP keeps the gathered bf16 parameters resident (--reshard-after-forward = reference)
  (default)                  committed hipBLASLt / rocBLAS solution selections for this model's GEMM shapes (--no-tuned-gemms)

Rank 0 prints ONE JSON line.  Besides the driver contract it carries
  roofline     - the dominant hand-written kernel (TTT-MLP backward scan), algorithmic FLOPs / measured launch time (HIP
                 events around every launch on the launch stream); `traffic` = HBM bytes per launch from the committed PMC
                 passes of the same kernel build at the same scan length (profiles/*pmc_traffic*.json, `traffic_source`), null
                 when none matches; the other hand-written kernels are listed under roofline.other
  cpu_baseline - N=1 only: the CPU port of the same layer (fp32, eager; the TTT scan = the oracle's restatement of the
                 reference ops path, dual form with the reference's checkpoint groups) timed on the host cores on a bounded
                 sample
  fsdp1        - N=1 only: the same step through FSDP2 over a one-rank mesh (the code path of N > 1), a few timed steps, so
                 that a 1 -> 8 curve can be read like for like
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBPS = 8000.0              # HBM3E, same table
TOKENS_PER_FRAME = 30 * 45


def log(msg):
    """progress to stderr (the single JSON line owns stdout)"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--video-length", default="9sec", choices=["3sec", "9sec", "18sec", "30sec", "63sec"],
                    help="9sec (default) = the configuration BASELINE.json's metric is quoted on; 3sec = BASELINE configs[1]")
    ap.add_argument("--ssm-layer", default="ttt_mlp", choices=["ttt_mlp", "ttt_linear"])
    ap.add_argument("--impl", default="auto", choices=["auto", "generic", "mfma"])
    ap.add_argument("--adapter", default="auto", choices=["auto", "sft", "qkvo"],
                    help="which parameters train: sft = all (the reference's 3 s stage, configs/train/ttt-mlp/3s.toml), qkvo = the attention / "
                         "TTT projections, TTT inner parameters and gates only (its longer stages, 9s.toml ...); auto = the reference's "
                         "setting for the chosen --video-length")
    ap.add_argument("--no-fsdp1-compare", action="store_true",
                    help="N=1: skip the extra short run through FSDP2 over a one-rank mesh (reported as `fsdp1`)")
    ap.add_argument("--fsdp1-steps", type=int, default=3)
    ap.add_argument("--cpu-baseline-budget", type=float, default=25.0, help="seconds of CPU work the cpu_baseline sample is sized for")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    # debugging knobs: anything that shrinks the workload marks the result invalid
    ap.add_argument("--tp", type=int, default=0,
                    help="tensor parallelism in groups of T neighbouring ranks (--gpus = dp x T): ONE sample per group and step, head shards for attention and the "
                         "TTT layer, token shards for the token-wise work (apply_tp layout 'full', reference parallelisms.py:106-152); "
                         "parameters, gradients and optimizer state sharded by FSDP2 over all ranks (apply_parallelisms; --fsdp off: replicas + "
                         "a gradient all-reduce, world == T).  --tp 1 on one GPU runs the same code path over a "
                         "one-rank group (what the layout's unfused glue costs); 0 = off")
    ap.add_argument("--remat-keep", default="attn,scan,fc2",
                    help="outputs a re-materialised layer keeps instead of recomputing them (comma list of attn, scan, fc2; 'none' = "
                         "the reference's behaviour: the whole layer is recomputed) - ttt_amd/infra/remat_cache.py")
    ap.add_argument("--remat-keep-layers", type=int, default=None,
                    help="only the first N re-materialised layers keep their kernel outputs (default: all of them) - the 63 s step on one GPU has "
                         "room for the attention outputs of about ten layers")
    ap.add_argument("--remat-free-layers", default="auto",
                    help="transformer layers that keep their activations instead of being re-materialised in backward: "
                         "'auto' (sized to the free HBM of this GPU), or an integer; 0 = the reference's 80-GB-GPU setting")
    ap.add_argument("--offload-gib-per-layer", type=float, default=0.0,
                    help="GiB of saved activations EVERY remat-free layer parks in pinned host memory between its forward and its backward "
                         "(ttt_amd/infra/host_offload.py: D2H / H2D copies on side streams beside the compute, same bits); the freed HBM goes to more "
                         "remat-free layers (--remat-free-layers auto sizes them with the offload in place); 0 = off")
    ap.add_argument("--offload-park-kept", action="store_true",
                    help="the kernel outputs re-materialised layers keep (--remat-keep) wait in pinned host memory too, fetched back two layers ahead of the backward")
    ap.add_argument("--offload-layers", type=int, default=None, help="only the first N remat-free layers offload (default: all of them)")
    ap.add_argument("--offload-soft-frac", type=float, default=2.0,
                    help="share of the device memory above which the host thread waits for every copy out before it saves more (default: never)")
    ap.add_argument("--offload-one-stream", action="store_true", help="(the default since call HO13; kept for the call scripts)")
    ap.add_argument("--offload-nonblocking-end", action="store_true", help="(the default since call HO13; kept for the call scripts)")
    ap.add_argument("--offload-two-streams", action="store_true", help="A/B: a side stream per copy direction (aliases with the scan / backward side streams on 4 hardware queues: -12 %%)")
    ap.add_argument("--offload-blocking-end", action="store_true", help="A/B: the host thread waits for every copy out at the end of the forward")
    ap.add_argument("--offload-no-batch", action="store_true", help="A/B: every copy out is issued at once behind its own event instead of together with its layer's (4 - 5 x slower copies)")
    ap.add_argument("--offload-lookahead", type=int, default=2, help="layers ahead of the backward whose parked tensors are fetched")
    ap.add_argument("--offload-trace", action="store_true", help="DEBUG: timed events around every copy and compute-stream wait of the LAST timed step (config.host_offload.trace)")
    ap.add_argument("--offload-backlog-gib", type=float, default=64.0,
                    help="GiB of copies out the host thread may have queued before it waits for the oldest")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="use hipBLASLt's default heuristic instead of the committed "
                    "solution selections (ttt_amd/infra/gemm_tuning_gfx950.csv)")
    ap.add_argument("--reshard-after-forward", action="store_true",
                    help="FSDP: free the gathered bf16 parameters after each layer's forward and all-gather them again in "
                         "backward (the reference's 80-GB setting); default: keep them resident (14.5 GB of 288)")
    ap.add_argument("--local-batch", type=int, default=1,
                    help="samples per GPU (default 1 = the reference's training setting).  The TTT scans of a second sample run "
                         "beside the first at no extra wall time (one workgroup per head, 48 of 256 CUs at batch 1)")
    ap.add_argument("--fsdp", default="auto", choices=["auto", "flat", "on", "off"],
                    help="how parameters / gradients / optimizer state are kept.  flat = ttt_amd.infra.flat_fsdp.FlatFSDP: the reference's "
                         "FSDP partitioning (one unit per TransformerLayer + root, fp32 masters and AdamW state sharded over the ranks, bf16 "
                         "compute parameters, fp32 gradient reduction) on flat buffers - one in-place all-gather and one reduce-scatter per unit "
                         "and step on a side stream, frozen parameters replicated (round 4: FSDP2's per-parameter copies cost 7 %% of the step on "
                         "one rank, profiles/r4i_*).  on = the reference's own FSDP2 wrapping (apply_fsdp).  off (one GPU only): "
                         "ReplicaMixedPrecision, the same arithmetic without any sharding machinery.  auto = off on one GPU, flat on several")
    ap.add_argument("--no-fsdp", action="store_true", help="same as --fsdp off")
    ap.add_argument("--role", default=None, choices=["orchestrate", "worker"],
                    help="worker = a measuring process (one per GPU; what a launcher starts).  orchestrate (the default WITHOUT a launcher "
                         "environment) = a thin parent that holds no GPU context: it starts the measurement as child process(es) - under "
                         "torch.distributed.run for --gpus N > 1 -, retries ONCE with conservative memory settings if that fails, then (N = 1, "
                         "default workload) runs the metric's other contexts as legs `ctx3s` / `ctx63s` and the CPU baseline, and prints the ONE JSON line")
    ap.add_argument("--no-legs", action="store_true", help="N=1: skip the legs (`ctx3s` = BASELINE configs[1], `ctx63s` = the metric's 63 s context, `ctx30s` = configs[3], `sample63s` = configs[4])")
    ap.add_argument("--leg-steps", type=int, default=2, help="timed steps of each leg (after one warm-up step)")
    ap.add_argument("--legs", default="all", help="N=1: which legs run after the main measurement: 'all' or a comma list of ctx3s, ctx63s, ctx30s, sample63s")
    ap.add_argument("--time-budget", type=float, default=1500.0,
                    help="seconds the whole default run may take: a leg whose estimated duration does not fit is skipped with a stated reason")
    ap.add_argument("--retry-reason", default=None, help=argparse.SUPPRESS)          # set by the orchestrator on its second attempt
    ap.add_argument("--pipeline-parts", type=int, default=None,
                    help="TTT-MLP layer forward as a pipeline over this many parts of the sequence (the scan of one part on a side stream beside "
                         "the projections of the next, ttt_amd/models/ssm/pipeline.py); 0 = one piece; default: the library's (TTT_PIPELINE_PARTS)")
    ap.add_argument("--debug-option", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B: a library debug option (ttt_hip_debug_option: overlap_tail, deriver_split, fast_records, groups_per_chunk) for this run; "
                         "recorded in config.debug_options")
    ap.add_argument("--layers", type=int, default=None, help="DEBUG: fewer layers (result flagged invalid)")
    ap.add_argument("--torch-profile", default=None, metavar="FILE",
                    help="DEBUG: after the timed region run ONE more step under torch.profiler and write per-operator tables "
                         "(device time by operator, and by operator + input shapes + call site) to FILE")
    return ap.parse_args()


# [optimizer] / [training] of the reference's configs/train/ttt-mlp/{3s,9s,18s,30s,63s}.toml
_OPT = {"lr": 1e-5, "lr_ssm": 1e-5, "lr_end": 1e-5, "gradient_clipping_norm": 0.1}
OPTIMIZER = {"3sec": dict(_OPT, lr_ssm=1e-4, warmup_steps=100, steps=5000), "9sec": dict(_OPT, warmup_steps=100, steps=5000),
             "18sec": dict(_OPT, warmup_steps=50, steps=1000), "30sec": dict(_OPT, warmup_steps=50, steps=1000),
             "63sec": dict(_OPT, warmup_steps=25, steps=250)}
TEXT_LEN = {"3sec": 498, "9sec": 502, "18sec": 471, "30sec": 497, "63sec": 458}   # configs/eval/ttt-mlp/*.toml:16 (L % 64 == 0)


class KernelTimer:
    """HIP-event timing of every TTT scan launch, recorded on the stream the kernel is enqueued on
    (torch's current stream - the extension launches there)."""

    def __init__(self, ext):
        self.ext, self.active, self.events = ext, False, {"fwd": [], "bwd": [], "attn_fwd": [], "attn_bwd": [], "fwd_part": []}
        self.part_calls = 0          # pipelined forwards (ttt_forward_chunk launches that start at step 0)
        self._orig = {}

    def install(self):
        if getattr(self.ext, "_bench_timer", None) is not None:      # a previous run (replica -> FSDP fallback) wrapped it already
            self.ext._bench_timer.uninstall()
        self.ext._bench_timer = self
        for name, key in (("ttt_forward", "fwd"), ("ttt_backward", "bwd"), ("ttt_linear_forward", "fwd"), ("ttt_linear_backward", "bwd"),
                          ("attn_forward", "attn_fwd"), ("attn_backward", "attn_bwd"), ("ttt_forward_chunk", "fwd_part")):
            orig = getattr(self.ext, name)
            self._orig[name] = orig

            def wrapped(*a, _o=orig, _k=key):
                if not self.active:
                    return _o(*a)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                _o(*a)
                e.record()
                self.events[_k].append((s, e, tuple(a[0].shape)))
                if _k == "fwd_part" and int(a[16]) == 0:
                    self.part_calls += 1
            setattr(self.ext, name, wrapped)

    def reset(self):
        for ev in self.events.values():
            ev.clear()
        self.part_calls = 0

    def uninstall(self):
        for name, orig in self._orig.items():
            setattr(self.ext, name, orig)
        self._orig = {}
        self.ext._bench_timer = None

    def summary(self):
        out = {}
        for k, ev in self.events.items():
            if ev:
                ms = [s.elapsed_time(e) for s, e, _ in ev]
                out[k] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms), "shape": ev[0][2]}
        if "fwd_part" in out and self.part_calls:
            # a pipelined forward scan = several launches (parts of the sequence) on the side stream: reported per whole scan
            p = out.pop("fwd_part")
            merged = {"launches": self.part_calls, "avg_ms": p["total_ms"] / self.part_calls, "total_ms": p["total_ms"], "shape": p["shape"],
                      "parts_per_scan": p["launches"] / self.part_calls}
            if "fwd" in out:         # (re-materialised layers that keep no scan result run the one-call scan too)
                f = out["fwd"]
                merged["total_ms"] += f["total_ms"]
                merged["launches"] += f["launches"]
                merged["avg_ms"] = merged["total_ms"] / merged["launches"]
            out["fwd"] = merged
        return out


class ClockSampler:
    """Shader clock (MHz) and socket power (W) of one GPU sampled by a background thread (amdsmi) during the timed region, so that
    the 2.4 GHz behind the MFMA peak (MI355X_MICROARCH.md) can be checked against what the step actually ran at.  Best effort:
    `summary()` is {"error": ...} when amdsmi is missing or refuses (the measurement itself never depends on it)."""

    def __init__(self, index=0, period_s=0.25):
        self.index, self.period_s, self.samples, self.error = index, period_s, [], None
        self._stop, self._thread, self._h, self._smi = None, None, None, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            # HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumber torch's devices; amdsmi sees all of them
            vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
            if vis:
                try:
                    index = int(vis.split(",")[index])
                except (ValueError, IndexError):
                    pass
            self._h, self._smi = hs[index if index < len(hs) else 0], amdsmi
        except Exception as ex:            # noqa: BLE001 - any failure of the SMI library only loses the clock fields
            self.error = repr(ex)[:200]

    def read(self):
        """(MHz, W) now; a field is None when the library does not report it"""
        smi, mhz, w = self._smi, None, None
        try:
            c = smi.amdsmi_get_clock_info(self._h, smi.AmdSmiClkType.GFX)
            mhz = c.get("clk") if isinstance(c.get("clk"), (int, float)) else None
        except Exception as ex:            # noqa: BLE001
            self.error = repr(ex)[:200]
        try:
            p = smi.amdsmi_get_power_info(self._h)
            for k in ("current_socket_power", "socket_power", "average_socket_power"):
                if isinstance(p.get(k), (int, float)) and 0 < p[k] < 0xFFFF:
                    w = p[k]
                    break
        except Exception as ex:            # noqa: BLE001
            self.error = repr(ex)[:200]
        return mhz, w

    def start(self):
        import threading
        if self._h is None or self._thread is not None:
            return
        self.samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append(self.read())
                self._stop.wait(self.period_s)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=5)
            self._thread = None

    def summary(self):
        mhz = [m for m, _ in self.samples if m]
        w = [p for _, p in self.samples if p]
        if not mhz and not w:
            return {"error": self.error or "no samples"}
        avg = lambda v: round(sum(v) / len(v), 1) if v else None
        return {"clock_mhz_avg": avg(mhz), "clock_mhz_min": min(mhz) if mhz else None, "clock_mhz_max": max(mhz) if mhz else None,
                "power_w_avg": avg(w), "power_w_max": max(w) if w else None, "samples": len(self.samples), "period_s": self.period_s}


def pmc_traffic(kernel, B, NH, NC):
    """(bytes, source file) - HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/*pmc_traffic*.json, written by tools/pmc_traffic.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of
    tools/op_bench.py, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Only a summary measured at THIS
    launch geometry (B, NH, NC) counts; the newest one wins.  PMC passes cannot run inside the timed region, hence a
    committed measurement of the same kernel build instead of a live one; (None, None) when nothing matches."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
            geo = d.get("geometry", {"B": 1, "NH": 48, "NC": 282})
            k = d["kernels"].get(kernel)
        except Exception:
            continue
        if k and (geo.get("B"), geo.get("NH"), geo.get("NC")) == (B, NH, NC):
            return (k.get("traffic_bytes_per_backward") or (k.get("fetch_bytes", 0) + k.get("write_bytes", 0))), os.path.relpath(f, ROOT)
    return None, None


def cpu_baseline(ssm_layer, budget_s=25.0):
    """CPU port of the same workload on the host cores, fp32, eager: ONE 5B-geometry TransformerLayer (torch CPU; AdaLN, local
    attention via SDPA, projections, MLP) whose TTT scans run as the ORACLE's restatement of the reference ops path
    (oracle/ttt_oracle.py:scan_dual = ops/ttt_mlp.py:9-99 statements, dual form on the full eta tile, the reference's
    checkpoint groups of 16, torch.autograd for the backward) - i.e. what `use_kernel=False` executes in the reference.
    Forward + backward of the layer at the real per-step shapes (48 heads, mini-batches of 64 tokens, head dim 64) on a
    BOUNDED sample: one scene of `f` latent frames + its text tokens, f chosen from a 1-frame probe so that the timed pass
    stays near `budget_s` seconds.  Scaled to whole-model video-tokens/s by the 42 layers (per-token cost of attention
    grows with the segment length, so a short sample flatters the CPU: the sample is stated)."""
    from oracle import ttt_oracle as O
    import ttt_amd.models.ssm.ttt_layer as TL
    from ttt_amd.models.cogvideo.dit import TransformerLayer
    from ttt_amd.models.cogvideo.utils import SequenceMetadata
    from ttt_amd.models.configs import ModelConfig

    def oracle_scan(kind):
        def run(XK, XQ, XV, eta, ln_w, ln_b, *states_and_g):
            *st, G = states_and_g
            out, _ = O.scan_dual(kind, XQ, XK, XV, eta, ln_w, ln_b, *st, checkpoint_group_size=int(G))
            return out.permute(0, 2, 3, 1, 4)            # the reference's [B, NC, CS, NH, F] (ops/ttt_mlp.py:99)
        return run

    saved = (TL.ttt_mlp, TL.ttt_linear)
    TL.ttt_mlp, TL.ttt_linear = oracle_scan("mlp"), oracle_scan("linear")
    try:
        torch.manual_seed(0)
        threads = torch.get_num_threads()

        def timed(frames, n_text):
            cfg = ModelConfig.get_preset("5B", "3sec", ssm_layer=ssm_layer, adapter_method="sft", compressed_num_frames=frames)
            layer = TransformerLayer(cfg)
            for m in layer.modules():
                if hasattr(m, "use_kernel"):
                    m.use_kernel = False
            n_vid = frames * TOKENS_PER_FRAME
            meta = SequenceMetadata(text_length=n_text, seq_text_length=n_text, num_frames=frames, num_chunks=1,
                                    tokens_per_frame=TOKENS_PER_FRAME, latent_height=60, latent_width=90,
                                    t_emb=torch.randn(1, cfg.time_embed_dim))
            vid = torch.randn(1, n_vid, cfg.model_dim, requires_grad=True)
            txt = torch.randn(1, n_text, cfg.model_dim, requires_grad=True)
            t0 = time.perf_counter()
            v, t = layer(vid, txt, meta)
            (v.square().mean() + t.square().mean()).backward()
            return time.perf_counter() - t0, n_vid, n_vid + n_text

        # text lengths keep L a multiple of the mini-batch size (64): 1 frame + 58, 2 + 52, 4 + 40, 7 + 22, 13 + 498 (= the 3 s segment)
        text_for = {1: 58, 2: 52, 4: 40, 7: 22, 13: 498}
        timed(1, 58)                                      # warm-up (thread pools, oneDNN primitives)
        t1, _, _ = timed(1, 58)
        frames = 1
        # at most 4 frames (L = 5440): SDPA's math fallback on the CPU would need 4 S^2 bytes per head and direction beyond that
        for f in (4, 2):
            if t1 * f * (1.0 + 0.25 * f) <= budget_s:      # linear part + the quadratic attention share, measured at 1 frame
                frames = f
                break
        dt, n_vid, L = (t1, TOKENS_PER_FRAME, TOKENS_PER_FRAME + 58) if frames == 1 else timed(frames, text_for[frames])
    finally:
        TL.ttt_mlp, TL.ttt_linear = saved
    tok_s = n_vid / (dt * 42)
    # How much the short sample flatters the CPU: the REFERENCE's own TransformerLayer (ttt/models/cogvideo/dit.py, ops path) timed
    # in the build container on these very samples and on the whole 13-frame attention segment (tools/ref_cpu_layer_baseline.py,
    # BASELINE.md section 2, 8 vCPU): seconds per video token and layer relative to the full segment (L = 18 048).
    REF_FULL_SEGMENT_RATIO = {1: 2.78, 4: 1.56, 13: 1.0}
    ratio = REF_FULL_SEGMENT_RATIO.get(frames)
    return {"value": tok_s, "unit": "video-tokens/s", "cores": threads, "kind": "port", "dtype": "f32",
            "full_segment_ratio": ratio, "value_at_full_segment": (tok_s / ratio) if ratio else None,
            "reference_measured": {"where": "build container, 8 vCPU Xeon 2.1 GHz, BASELINE.md section 2", "video_tok_s_42_layers": {"L=1408": 7.76, "L=5440": 4.36, "L=18048": 2.79}},
            "sample": f"1 of 42 TransformerLayers (5B geometry, {ssm_layer}) fwd+bwd, fp32 eager, scan = oracle dual form (reference ops "
                      f"path restated, checkpoint groups of 16), one scene of {frames} latent frame(s) + {L - n_vid} text tokens "
                      f"(L={L}; the 3 s segment is 13 frames, L=18048): {dt:.2f} s on {threads} threads; scaled by 42 layers.  `value` is "
                      f"that sample; per video token the reference's own layer costs {ratio} x more at the full segment than on this "
                      f"sample (measured with the reference's code, BASELINE.md section 2), so the bench workload's CPU rate is "
                      f"value / {ratio} = `value_at_full_segment`"}


def size_warm_and_time(step, hk, remat_free_layers, warmup, steps, world):
    """Activation re-materialisation sized for this GPU, warm-up and the timed region - the control flow every rank of an
    N-GPU run must walk IDENTICALLY (a rank that takes another branch leaves the others in a collective).  `hk` supplies the
    device: memory statistics, synchronisation, the MIN all-reduce, barriers (tests/test_bench_policy_gloo.py drives this
    function on two gloo ranks with a fake device).  Returns (remat_free_layers used, seconds of the timed region, last loss).

    The reference checkpoints every transformer layer (configs/train/ttt-mlp/3s.toml:31, tuned for 80 GB GPUs).  With 288 GB
    per MI355X most layers can keep their activations: probe the per-layer activation footprint with two untimed steps and
    keep as many layers un-checkpointed as fit under `cap` of the device memory.  Same arithmetic, same results."""
    # share of the device memory the activations may fill: multi-rank runs stay further from the edge, because an
    # out-of-memory error on ONE rank cannot be recovered from while the others sit in a collective
    cap = 0.88 if world == 1 else 0.80
    auto = remat_free_layers == "auto"
    peak0 = 0
    if auto:
        # layers in the second probe step: 4 at the 3 s geometry (2.7 GB of activations per layer and sample), 1 for the long
        # videos, whose layers are 3 - 20 x larger; an out-of-memory error in the probe means "none fit"
        probe = hk.probe_layers
        # first probe step: every layer re-materialised.  If even that does not fit - the kernel outputs the re-materialised
        # layers KEEP (remat_keep) are 5 GB per layer at 30 s - the keeping policy is reduced kind by kind (one GPU; on several
        # ranks an out-of-memory error inside a step's collectives is not recoverable: state --remat-keep explicitly there)
        while True:
            hk.reset_peak()
            hk.set_free_layers(0)
            try:
                step()
                hk.synchronize()
                break
            except hk.oom:
                if world > 1 or not hk.reduce_keep():
                    raise
                hk.release()
        peak0 = hk.max_allocated()
        n_free = 0
        if probe:
            hk.reset_peak()
            hk.set_free_layers(probe)
            try:
                step()
                hk.synchronize()
                per_layer = max((hk.max_allocated() - peak0) / probe, 1.0)
                n_free = int(max(0, min(hk.num_layers, (cap * hk.total_memory - peak0) // per_layer)))
                log(f"sizing: peak with every layer re-materialised {peak0 / 2**30:.1f} GiB, + {per_layer / 2**30:.2f} GiB per free layer "
                    f"({probe} probed), cap {cap * hk.total_memory / 2**30:.0f} GiB -> {n_free} layers")
            except hk.oom:
                if world > 1:
                    raise                  # the other ranks sit in a collective: not recoverable
                hk.release()
                n_free = 0
        if world > 1:       # every rank must take the same decision
            n_free = hk.all_reduce_min(n_free)
    else:
        n_free = int(remat_free_layers)
    # warm-up with the chosen setting; if the caching allocator's fragmentation pushes it over the edge, back off and retry
    # (still untimed).  With an explicit --remat-free-layers N an out-of-memory error is fatal, as it should be.
    refinements, fail_at, thrash_rounds = 0, None, 0      # settings at or above `fail_at` ran out of memory in a warm-up: never tried again

    def back_off():
        hk.release()
        return max(0, n_free - max(1, n_free // 10))

    while True:
        hk.set_free_layers(n_free)
        hk.reset_peak()
        try:
            for _ in range(max(warmup, 1 if auto else 0)):
                step()
            hk.synchronize()
            ok = 1
        except hk.oom:
            if not auto:
                raise
            ok = 0
        if world > 1:
            ok = hk.all_reduce_min(ok)
        if ok and auto and refinements < 3 and 0 < n_free < hk.num_layers:
            # up to two free refinements with the footprint measured at the chosen setting (the walk at 9 s was 3 -> 8 -> 13; a third
            # refinement finds room for a 14th layer and buys nothing measurable: 8 139 / 8 114 against 8 155 / 8 104 video-tok/s on
            # one box at 243 instead of 237 GiB, profiles/r5h_*; the probe over-estimates a layer's
            # footprint - by a third when re-materialised layers keep their kernel outputs, which a layer that keeps everything
            # no longer needs)
            refinements += 1
            per_layer = max((hk.max_allocated() - peak0) / n_free, 1.0)
            better = int(max(0, min(hk.num_layers, (cap * hk.total_memory - peak0) // per_layer)))
            log(f"sizing: peak at {n_free} free layers {hk.max_allocated() / 2**30:.1f} GiB ({per_layer / 2**30:.2f} GiB per layer) -> {better} layers")
            if refinements == 3:
                # (round 6: the third record buffer of the TTT-MLP backward's schedule 2 moved the second refinement from 13 to 12 layers
                # although 13 fit - 237.9 of 253 GiB, profiles/r6f_*; the third refinement may add ONE layer, not walk on to the edge)
                better = min(better, n_free + 1)
            if fail_at is not None:
                better = min(better, fail_at - 1)
            if world > 1:
                better = hk.all_reduce_min(better)
            if better > n_free:
                n_free = better
                continue
        if not ok:
            fail_at = n_free if fail_at is None else min(fail_at, n_free)
            if world > 1:
                fail_at = hk.all_reduce_min(fail_at)
            n_free = back_off()
            continue
        # ---- timed region.  (One GPU, automatic setting: an out-of-memory error here - allocator fragmentation that the
        # warm-up step did not show - costs one layer and the whole region is warmed and timed again; nothing of a
        # failed attempt enters the result.)
        hk.barrier()
        hk.synchronize()
        hk.before_timed(n_free)
        retries0 = hk.alloc_retries()
        try:
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = step()
            hk.synchronize()
        except hk.oom:
            hk.after_timed()
            if not auto or world > 1:
                raise
            log(f"out of memory inside the timed region at remat_free_layers={n_free}: backing off, timing again")
            fail_at = n_free if fail_at is None else min(fail_at, n_free)
            n_free = back_off()
            continue
        hk.barrier()
        dt = time.perf_counter() - t0
        hk.after_timed()
        # The caching allocator ran dry inside the timed region and recovered by freeing its cache and allocating again
        # (device-synchronising hipFree / hipMalloc between kernels: call Q measured a 3 s step at 6.6 s that way, with the
        # reserved memory at the allocator's cap): that region measures the allocator, not the step.  Automatic setting: every
        # rank agrees (MIN all-reduce, walked by all ranks whether or not they saw retries), backs off and times again.
        thrash = 1 if hk.alloc_retries() > retries0 else 0
        if world > 1:
            thrash = 1 - hk.all_reduce_min(1 - thrash)
        if thrash and auto and n_free > 0 and thrash_rounds < 3:
            thrash_rounds += 1
            log(f"allocator retries inside the timed region at remat_free_layers={n_free}: backing off, timing again")
            fail_at = n_free if fail_at is None else min(fail_at, n_free)
            n_free = back_off()
            continue
        return n_free, dt, loss


def launcher_argv(gpus, argv, port=None):
    """The command line of an N-GPU measurement: one rank per GPU of this node under torch.distributed.run, rendezvous on 127.0.0.1
    (the container's hostname may not resolve) on a free port - what the reference's scripts/train_singlenode.sh:25-38 does with
    torchrun.  The ranks inherit stdout: rank 0 prints the one JSON line."""
    if port is None:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def worker_command(gpus, argv, port=None):
    """command line of the measuring process(es) for `argv` (bench.py's own arguments)"""
    argv = [a for a in argv] + ["--role", "worker"]
    return launcher_argv(gpus, argv, port) if gpus > 1 else [sys.executable, os.path.abspath(__file__)] + argv


# what the orchestrator's second attempt adds: every layer re-materialised, kernel outputs kept - the smallest activation footprint
# that still avoids recomputing the sequence kernels (an out-of-memory error of ONE rank of an N-GPU run cannot be recovered inside the run)
SAFE_MEMORY_ARGS = ["--remat-free-layers", "0", "--remat-keep", "attn,scan,fc2"]


def kill_process_group(proc, grace_s=10.0):
    """SIGTERM, then SIGKILL, to the process GROUP of `proc` (started with start_new_session=True), and wait until the group's
    leader is gone.  Falls back to the child alone where the group cannot be signalled."""
    import signal
    import subprocess
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError, OSError):
            try:
                proc.send_signal(sig)
            except (ProcessLookupError, OSError):
                pass
        try:
            proc.wait(timeout=grace_s)
            break
        except subprocess.TimeoutExpired:
            continue
    t_end = time.time() + grace_s                       # the ranks of a launcher die a moment after it: let the group drain
    while time.time() < t_end:
        try:
            os.killpg(proc.pid, 0)
        except (ProcessLookupError, PermissionError, OSError):
            break
        time.sleep(0.2)


def run_child(cmd, timeout, env=None):
    """Runs one child to its end.  stdout is captured (the JSON line), stderr is passed through line by line and its tail kept.
    Returns (return code, the LAST JSON object line of stdout or None, tail of stderr).  (tests replace this function)"""
    import subprocess
    import threading
    # own session = own process group: for --gpus N > 1 the child is the torch.distributed.run launcher, and its rank processes must
    # not outlive a time-out (they would hold the GPUs under the retry and the legs)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    tail, out = [], []

    def pump_err():
        for ln in proc.stderr:
            sys.stderr.write(ln)
            sys.stderr.flush()
            tail.append(ln.rstrip())
            del tail[:-30]

    def pump_out():
        for ln in proc.stdout:
            out.append(ln)

    th = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for t in th:
        t.start()
    try:
        rc = proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        kill_process_group(proc)
        rc = -9
        tail.append(f"bench.py: child exceeded {timeout:.0f} s and was killed (with its process group)")
    for t in th:
        t.join(timeout=10)
    line = None
    for ln in reversed(out):
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                line = json.loads(ln)
                break
            except ValueError:
                continue
    return rc, line, tail


def rccl_summary(log_dir):
    """What RCCL decided for this job, from its own INIT / GRAPH log lines (NCCL_DEBUG_FILE of the ranks): channel count, ring / tree
    lines of rank 0 - a first look at the topology an 8-GPU run actually got.  {} when nothing was logged."""
    import glob
    import re
    res = {}
    files = sorted(glob.glob(os.path.join(log_dir, "rccl.*.log")))
    if not files:
        return res
    rings, trees, chans, nranks = [], [], set(), None
    for ln in open(files[0], errors="replace"):
        m = re.search(r"Channel (\d+)/(\d+)\s*:\s*(.*)", ln)
        if m and "Ring" not in ln and "Tree" not in ln:
            chans.add(int(m.group(1)))
            if len(rings) < 4:
                rings.append(f"channel {m.group(1)}: {m.group(3).strip()[:80]}")
        if "Trees" in ln and len(trees) < 2:
            trees.append(ln.split("Trees", 1)[1].strip()[:120])
        m = re.search(r"nranks (\d+)", ln)
        if m:
            nranks = int(m.group(1))
    res = {"rank_logs": len(files), "nranks": nranks, "channels": len(chans), "rings_head": rings, "trees_head": trees}
    return res


# the 63 s step on ONE GPU: every layer re-materialised; of the kernel outputs a re-materialised layer could keep only the attention
# outputs (2.3 GB per layer at 63 s, 37 ms saved per GB) of the first ten layers fit beside 223 GiB (round 5, one box,
# profiles/r5h_*: 6 929 against 6 804 video-tok/s, 49.3 against 50.2 s per step, 245.5 GiB allocated / 256.2 reserved)
# The 63 s leg: every layer re-materialised; the attention outputs of ALL 42 layers are kept - parked in pinned host memory (91 GiB per step,
# ttt_amd/infra/host_offload.py; round 6, call HO13: 42.9 s per step against 45.9 with the attention outputs of ten layers kept on the device,
# the setting of rounds 5 / 6 and the leg's fallback if the parked attempt fails).
CTX63S_KEEP = ["--remat-keep", "attn", "--offload-park-kept", "--offload-lookahead", "2"]
CTX63S_KEEP_FALLBACK = ["--remat-keep", "attn", "--remat-keep-layers", "10"]
# generous estimates (model build + sizing probes + 1 warm-up + 2 timed steps; 24 s per step at 30 s, 49 s at 63 s; sampling: two
# network evaluations of 22 s on the guidance pair + the build)
LEG_ESTIMATE_S = {"ctx3s": 150.0, "ctx63s": 480.0, "ctx30s": 380.0, "sample63s": 240.0}
LEG_ORDER = ("ctx3s", "ctx63s", "ctx30s", "sample63s")     # the metric's two contexts first, then BASELINE configs[3] and configs[4]


def host_room_gib():
    """GiB of host memory this process may still take (ttt_amd/infra/host_offload.py: MemAvailable and the cgroup's limit; tests replace this)"""
    from ttt_amd.infra.host_offload import host_room_gib as f
    return f()


CTX63S_PARKED_HOST_GIB = 160.0      # the parked 63 s leg pins 92 GiB; it is only tried with this much host memory to spare


def leg_command(name, args, fallback=False):
    """`ctx3s` = BASELINE configs[1] (configs/train/ttt-mlp/3s.toml: one segment, adapter sft), `ctx63s` = the metric's second context
    (63s.toml: 21 scenes, L = 351 168; every layer re-materialised, the first ten keep their attention outputs - what fits ONE 288-GB GPU,
    DESIGN.md section 6; the reference shards this stage over 4 x 4 GPUs), `ctx30s` = BASELINE configs[3] (30s.toml: 10 scenes,
    L = 168 320, the long-sequence chunked scan) as a training step on one GPU, `sample63s` = BASELINE configs[4] (configs/eval/ttt-mlp/63s.toml:
    mini-batches of 16, no scan checkpoints): ONE denoising step of the mirrored DPM-Solver++ sampler on the batched guidance pair
    (tools/sample_bench.py).  A leg is its own process: memory state and failures stay its own."""
    if name == "sample63s":
        return [sys.executable, os.path.join(ROOT, "tools", "sample_bench.py"), "--video-length", "63sec", "--steps", "1", "--impl", args.impl]
    length = {"ctx3s": "3sec", "ctx63s": "63sec", "ctx30s": "30sec"}[name]
    argv = ["--gpus", "1", "--video-length", length, "--steps", str(max(1, args.leg_steps)), "--warmup", "1", "--no-fsdp1-compare",
            "--ssm-layer", args.ssm_layer, "--impl", args.impl]
    if name == "ctx63s":
        argv += ["--remat-free-layers", "0"] + (CTX63S_KEEP_FALLBACK if fallback else CTX63S_KEEP)
    if args.no_tuned_gemms:
        argv.append("--no-tuned-gemms")
    if args.pipeline_parts is not None:
        argv += ["--pipeline-parts", str(args.pipeline_parts)]
    return worker_command(1, argv)


def leg_summary(line):
    if line.get("metric") == "sampling_denoising_step_seconds":          # tools/sample_bench.py
        c = line.get("config", {})
        return {"value": line["value"], "unit": line["unit"], "latent_frames_per_s": line.get("latent_frames_per_s"),
                "projected_50_step_video_s": line.get("projected_50_step_video_s"), "workload": c.get("workload"), "tokens": c.get("tokens"),
                "mini_batches": c.get("mini_batches"), "scan_impl": c.get("scan_impl"), "steps": c.get("timed_steps"),
                "peak_mem_gib": line.get("peak_mem_GiB"), "valid": c.get("layers") == 42}
    r = line.get("roofline") or {}
    other = r.get("other") or {}
    dom_bwd = "bwd" in (r.get("kernel") or "")
    pick = lambda k: round(other[k]["avg_ms"], 3) if k in other else None
    return {"value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "steps": line["steps"], "warmup": line["warmup"],
            "workload": line["config"]["workload"], "remat_free_layers": line["config"]["remat_free_layers"],
            "remat_keep": line["config"].get("remat_keep"), "remat_keep_layers": line["config"].get("remat_keep_layers"),
            "host_offload_gib_per_step": (line["config"].get("host_offload") or {}).get("gib_per_step"), "peak_mem_gib": line.get("peak_mem_gib"), "peak_reserved_gib": line.get("peak_reserved_gib"),
            "ttt_mlp_bwd_ms": round(r["avg_launch_ms"], 3) if dom_bwd else pick("bwd"), "scan_fwd_ms": pick("fwd") if dom_bwd else round(r.get("avg_launch_ms", 0.0), 3),
            "attn_fwd_ms": pick("attn_fwd"), "attn_bwd_ms": pick("attn_bwd"), "roofline_frac": r.get("frac"), "valid": line["config"].get("valid")}


def attach_leg(line, name, summary):
    """A leg's result goes where the driver's record keeps it: `config.legs[name]` (the whole summary) and flat scalars
    `config.leg_<name>_*` (round 5: the driver kept `config` / `roofline` scalars and only the NAMES of other top-level keys)."""
    cfg = line.setdefault("config", {})
    cfg.setdefault("legs", {})[name] = summary
    line[name] = summary                                        # (top level too: where rounds 4 / 5 had it)
    for k in ("value", "ms_per_step", "peak_mem_gib", "latent_frames_per_s", "projected_50_step_video_s", "ttt_mlp_bwd_ms", "valid"):
        if isinstance(summary.get(k), (int, float, bool)):
            cfg[f"leg_{name}_{k}"] = summary[k]
    for k in ("skipped", "error"):
        if k in summary:
            cfg[f"leg_{name}_{k}"] = str(summary[k])[:200]


def orchestrate(args, argv):
    """The default entry (no launcher environment): see --role.  Prints ONE JSON line; exits non-zero when no measurement succeeded."""
    import tempfile
    t_start = time.time()
    argv = [a for a in argv]
    env = dict(os.environ)
    rccl_dir = None
    if args.gpus > 1:
        # RCCL: warnings on stderr as usual; unless the caller chose otherwise, its INIT / GRAPH decisions go to per-rank files that
        # the summary below reads (the first multi-GPU run of this code may be the only one: leave evidence)
        if "NCCL_DEBUG" not in env:
            rccl_dir = tempfile.mkdtemp(prefix="bench_rccl_")
            env.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,ENV", NCCL_DEBUG_FILE=os.path.join(rccl_dir, "rccl.%h.%p.log"))
    line, failures = None, []
    for attempt, extra in enumerate(([], SAFE_MEMORY_ARGS)):
        if attempt:
            extra = extra + ["--retry-reason", failures[-1][:300]]
            log(f"measurement failed ({failures[-1][:200]}): retrying ONCE with {' '.join(SAFE_MEMORY_ARGS)}")
        rc, line, tail = run_child(worker_command(args.gpus, argv + extra), timeout=max(600.0, args.time_budget), env=env)
        if rc == 0 and line is not None:
            break
        why = next((t for t in reversed(tail) if "OutOfMemory" in t or "out of memory" in t), None) or \
            next((t for t in reversed(tail) if "Error" in t or "error" in t), tail[-1] if tail else "no output")
        failures.append(f"rc {rc}: {why.strip()}")
        line = None
    if line is None:
        print(f"bench.py: no measurement ({'; '.join(failures)})", file=sys.stderr, flush=True)
        sys.exit(1)
    if rccl_dir:
        line["rccl"] = rccl_summary(rccl_dir)
        log(f"RCCL: {line['rccl']}")
    default_workload = args.video_length == "9sec" and args.layers is None and not args.tp and args.local_batch == 1
    if "fsdp1" in line:                                         # (measured inside the main child)
        line["config"]["fsdp1"] = line["fsdp1"]
        for k in ("value", "ms_per_step"):
            if isinstance(line["fsdp1"].get(k), (int, float)):
                line["config"][f"fsdp1_{k}"] = line["fsdp1"][k]
    if args.gpus == 1 and not args.no_legs and default_workload:
        wanted = [n for n in LEG_ORDER if args.legs == "all" or n in args.legs.split(",")]
        for name in wanted:
            left = args.time_budget - (time.time() - t_start)
            if left < LEG_ESTIMATE_S[name]:
                res = {"skipped": f"{left:.0f} s of the {args.time_budget:.0f} s budget left, the leg needs ~{LEG_ESTIMATE_S[name]:.0f} s"}
                log(f"{name}: skipped ({res['skipped']})")
                attach_leg(line, name, res)
                continue
            log(f"{name} leg (child process)")
            t0 = time.time()
            room = host_room_gib() if name == "ctx63s" else None
            short = room is not None and room < CTX63S_PARKED_HOST_GIB
            if short:
                log(f"{name}: {room:.0f} GiB of host memory to spare (< {CTX63S_PARKED_HOST_GIB:.0f}): attention outputs of ten layers on the device instead of parked")
            rc, leg, tail = run_child(leg_command(name, args, fallback=short), timeout=max(300.0, left))
            fell_back = f"host memory: {room:.0f} GiB to spare" if short else None
            if name == "ctx63s" and not short and not (rc == 0 and leg is not None) and args.time_budget - (time.time() - t_start) >= LEG_ESTIMATE_S[name]:
                # the parked attempt failed: once more with the attention outputs of ten layers kept on the device (rounds 5 / 6)
                fell_back = f"rc {rc}: {(tail[-1] if tail else 'no output')[:200]}"
                log(f"{name}: parked attempt failed ({fell_back}); once more with {' '.join(CTX63S_KEEP_FALLBACK)}")
                rc, leg, tail = run_child(leg_command(name, args, fallback=True), timeout=max(300.0, args.time_budget - (time.time() - t_start)))
            if rc == 0 and leg is not None:
                res = leg_summary(leg)
                res["leg_wall_s"] = round(time.time() - t0, 1)
                if fell_back:
                    res["fallback_after"] = fell_back
            else:
                res = {"error": f"rc {rc}: {(tail[-1] if tail else 'no output')[:300]}"}
            attach_leg(line, name, res)
    if args.gpus == 1 and not args.no_cpu_baseline:
        # the CPU leg runs in a child process with a wall-clock limit: whatever happens to it (host out-of-memory kill, a slow
        # box) the GPU measurement above is still printed
        log("cpu_baseline leg (child process)")
        rc, cb, tail = run_child([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--ssm-layer", args.ssm_layer,
                                  "--cpu-baseline-budget", str(args.cpu_baseline_budget)], timeout=max(240.0, 12 * args.cpu_baseline_budget))
        line["cpu_baseline"] = cb["cpu_baseline"] if (rc == 0 and cb and "cpu_baseline" in cb) else {"error": f"rc {rc}: {(tail[-1] if tail else 'no output')[:300]}"}
    line["bench_wall_s"] = round(time.time() - t_start, 1)
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps({"cpu_baseline": cpu_baseline(args.ssm_layer, args.cpu_baseline_budget)}))
        return
    under_launcher = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if args.role == "orchestrate" or (args.role is None and not under_launcher):
        # started without a launcher (the driver calls `python bench.py --gpus N` for every N): a thin parent starts the measuring
        # process(es) as children - under torch.distributed.run for N > 1 - and survives their failure (one retry)
        return orchestrate(args, sys.argv[1:])
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus"
    os.environ.setdefault("NCCL_DEBUG", "WARN")             # RCCL's warnings on stderr (the orchestrator sets INFO + per-rank files for N > 1)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.adapter == "auto":
        args.adapter = "sft" if args.video_length == "3sec" else "qkvo"      # configs/train/ttt-mlp/{3s,9s,...}.toml
    mode = "off" if args.no_fsdp else args.fsdp
    if args.tp >= 1:
        assert world % args.tp == 0, "--tp T: the ranks form world / T groups of T neighbours, each group works on ONE sample"
        line = _run(args, world, rank, local_rank, dev, no_fsdp=(mode == "off"), tp=args.tp)
        if rank == 0 and line is not None:
            print(json.dumps(line), flush=True)
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()
        return
    assert not (mode == "off" and world > 1), "--fsdp off is the one-GPU replica path"
    multi = {"auto": "flat", "flat": "flat", "on": "fsdp2"}.get(mode, "flat")           # the sharded implementation of this run
    import gc
    line = None
    if world == 1 and mode in ("auto", "off"):
        # one GPU: ReplicaMixedPrecision - the sharded path's arithmetic with nothing to shard.  (Round 4 measured the flat FSDP path
        # with its collectives skipped in this place on one box: 6 638 against 6 505 ms per step at equal re-materialisation - the
        # per-unit gradient casts it queues beside the backward cost more than the one pass at the end; profiles/r4l_*.)
        try:
            log("replica path (one GPU)")
            line = _run(args, world, rank, local_rank, dev, no_fsdp=True)
        except Exception as ex:      # an untested corner of the replica path must not cost the measurement
            if mode == "off":
                raise
            print(f"bench.py: replica path failed ({ex!r}); falling back to the flat FSDP path over a one-rank group", file=sys.stderr, flush=True)
        gc.collect()
        torch.cuda.empty_cache()
        if line is None:
            line = _run(args, world, rank, local_rank, dev, no_fsdp=False, sharded=multi)
        elif not args.no_fsdp1_compare and mode == "auto":
            # the same step through the sharded path over a one-rank group WITH its collectives (RCCL all-gather / reduce-scatter
            # of one rank) = the code path of N > 1 (like-for-like point of a 1 -> N curve)
            try:
                log(f"main line done: {line['value']:.1f} video-tok/s, {line['ms_per_step']:.0f} ms/step; fsdp1 comparison run")
                import copy
                a2 = copy.copy(args)
                a2.steps, a2.warmup = max(1, args.fsdp1_steps), 1
                a2.remat_free_layers = str(max(0, line["config"]["remat_free_layers"] - 1))
                f = _run(a2, world, rank, local_rank, dev, no_fsdp=False, quiet=True, sharded=multi)
                line["fsdp1"] = {"impl": multi, "value": f["value"], "ms_per_step": f["ms_per_step"], "steps": a2.steps,
                                 "remat_free_layers": f["config"]["remat_free_layers"], "peak_mem_gib": f["peak_mem_gib"],
                                 "ttt_mlp_bwd_ms": round(f["roofline"]["avg_launch_ms"], 3), "attn_bwd_ms": round(f["roofline"]["other"]["attn_bwd"]["avg_ms"], 3)}
            except Exception as ex:
                line["fsdp1"] = {"error": repr(ex)[:300]}
            gc.collect()
            torch.cuda.empty_cache()
    else:
        line = _run(args, world, rank, local_rank, dev, no_fsdp=False, sharded=multi)
    if rank == 0 and line is not None:
        if args.retry_reason:
            line["config"]["retry_reason"] = args.retry_reason
        print(json.dumps(line), flush=True)
    dist.barrier(device_ids=[local_rank])
    dist.destroy_process_group()


def _run(args, world, rank, local_rank, dev, no_fsdp, quiet=False, tp=False, sharded="fsdp2", communicate=True):
    import test_time_training as ext
    from ttt_amd.infra.parallelisms import (ReplicaMixedPrecision, apply_fsdp, enable_tuned_gemms, get_dp_mesh, init_distributed,
                                            init_model_parameters)
    from ttt_amd.infra.train_step import checked_optimizer_step
    from ttt_amd.models.cogvideo.model import CogVideoX
    from ttt_amd.models.configs import ModelConfig

    ext.load_library()
    ext.set_impl(args.impl)
    for kv in args.debug_option:
        ext.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    init_distributed("nccl")
    # RCCL builds its communicator (and allocates ~0.5 GiB of device buffers) at the FIRST collective: do that now, while the
    # device is empty - at 30 s the first collective used to come when the model had filled HBM and RCCL's allocation failed
    warm = torch.zeros(1, device=dev)
    dist.all_reduce(warm)
    torch.cuda.synchronize()
    # ... and torch's caching allocator leaves 4 % of HBM (11 GB) to everybody else - hipBLAS handles and workspaces of the
    # autograd threads, RCCL, the HIP library's own buffers: when the 30 s sizing probe ran torch to the brim, the NEXT
    # allocation to fail was hipblasCreate(), which is not an out-of-memory error the sizing can recover from
    torch.cuda.set_per_process_memory_fraction(0.96, dev)
    tuned = (not args.no_tuned_gemms) and enable_tuned_gemms()

    over = {}
    if args.layers is not None:
        over["num_layers"] = args.layers
    if args.ssm_layer == "ttt_linear":      # the reference trains TTT-Linear with these (configs/train/ttt-linear/3s.toml:9,32)
        over.update(mini_batch_size=16, scan_checkpoint_group_size=4)
    keep_items = [k for k in args.remat_keep.split(",") if k and k != "none"]          # "scan:20" = scan outputs in the first 20 re-materialised layers only
    over["remat_keep"] = tuple(k.split(":")[0] for k in keep_items)
    over["remat_keep_limits"] = {k.split(":")[0]: int(k.split(":")[1]) for k in keep_items if ":" in k}
    over["remat_keep_layers"] = args.remat_keep_layers
    cfg = ModelConfig.get_preset("5B", args.video_length, ssm_layer=args.ssm_layer, adapter_method=args.adapter, **over)
    frames, text_len = cfg.compressed_num_frames, TEXT_LEN[args.video_length]
    scenes = max((frames - 1) // 12, 1)
    L = frames * TOKENS_PER_FRAME + scenes * text_len
    assert L % cfg.mini_batch_size == 0

    tp = int(tp)                                          # 0: off; T: TP groups of T neighbouring ranks, world = dp x T
    dp, dp_rank = (world // tp, rank // tp) if tp else (world, rank)       # (the ranks of a TP group are ONE data-parallel rank)
    with torch.device("meta"):
        model = CogVideoX(cfg, effective_rank=dp_rank, effective_world_size=dp)
    if tp and no_fsdp:                                    # replicas of the parameters, partial gradients summed by an all-reduce
        from ttt_amd.infra.parallelisms import apply_tp, tp_sync_gradients
        assert tp == world, "--tp T --fsdp off: one TP group of replicas (world == T)"
        apply_tp(model, dist.group.WORLD, layout="full")
    elif tp:                                              # FSDP2 over all dp x T ranks; its reduce-scatter sums the TP partials
        from ttt_amd.infra.parallelisms import apply_parallelisms
        apply_parallelisms(model, tp_sharding=tp, reshard_after_forward=args.reshard_after_forward, tp_layout_on_one_rank=True)   # reference parallelisms.py:92-104
    elif not no_fsdp and sharded == "fsdp2":
        apply_fsdp(model, get_dp_mesh(), reshard_after_forward=args.reshard_after_forward)   # reference parallelisms.py:155-175
    model.to_empty(device=dev)
    torch.manual_seed(1234)                                # same init on every rank, then sharded
    with torch.no_grad():
        init_model_parameters(model)
        model.init_ssm_weights()
    model.setup_generator(seed=dp_rank, device=dev)        # (a TP group works on ONE sample: same draws on its ranks)
    parts_used = None
    for mod in model.modules():
        if hasattr(mod, "pipeline_parts"):
            if args.pipeline_parts is not None:
                mod.pipeline_parts, mod.pipeline_parts_auto = args.pipeline_parts, False
            parts_used = f"{mod.pipeline_parts}{'+' if getattr(mod, 'pipeline_parts_auto', False) else ''}"      # "4+": up to 8 for long scans
    replica = ReplicaMixedPrecision(model.dit) if no_fsdp else None
    flat = None
    if not no_fsdp and not tp and sharded == "flat":      # the same partitioning on flat buffers (ttt_amd/infra/flat_fsdp.py)
        from ttt_amd.infra.flat_fsdp import FlatFSDP
        flat = FlatFSDP(model.dit, always_communicate=communicate)
    # the reference's optimizer (ttt/infra/optimizers.py:200-264 with the values of configs/train/ttt-mlp/*.toml): four AdamW groups by
    # parameter NAME - TTT / SSM parameters at lr_ssm, the others at lr; bias / norm / b1 / b2 without weight decay -, betas
    # (0.9, 0.95), one LambdaLR schedule per group, gradient clipping at 0.1.  Works on the masters of either holder.
    from ttt_amd.infra.optimizers import ScheduleType, create_grouped_lr_scheduler, create_specialized_optimizer
    oc = OPTIMIZER[args.video_length]
    opt, sched_cfgs = create_specialized_optimizer(model, oc["lr"], oc["lr_ssm"], oc["lr_end"], oc["warmup_steps"], oc["steps"],
                                                   ScheduleType.LINEAR, ScheduleType.COSINE, args.adapter)
    lr_sched = create_grouped_lr_scheduler(opt, sched_cfgs)
    train_params = [p for g_ in opt.param_groups for p in g_["params"]]
    clip = oc["gradient_clipping_norm"]
    if flat:
        flat.attach_optimizer(opt)              # step pre-hook: finish_backward + the sweep-error gate; post-hook: publish (all-gathers)

    g = torch.Generator(device=dev).manual_seed(100 + dp_rank)
    LB = args.local_batch
    vid = torch.randn(LB, frames, 16, 60, 90, device=dev, generator=g)
    text = torch.randn(LB, scenes, text_len, cfg.text_dim, device=dev, generator=g)

    timer = KernelTimer(ext)
    timer.install()
    clocks = ClockSampler(local_rank) if rank == 0 else None     # sclk / socket power during the timed region (amdsmi, best effort)

    def step():
        # the reference's loop (train.py:131-166): zero_grad, loss, backward, clip, optimizer.step, lr_scheduler.step
        if args.offload_trace and dit.host_offload is not None:
            dit.host_offload.trace = []                # (DEBUG: the events of the most recent step)
        opt.zero_grad(set_to_none=True)
        loss = model(vid, text).mean()
        loss.backward()
        if flat:
            # (the units' reduce-scatters were queued by their last gradients' hooks; optimizer.step()'s hooks finish the backward,
            # look at the hand-over error word of the TTT-MLP backward - ONE device synchronisation, which a production loop needs
            # before AdamW, so the benchmark pays for it too - and publish the new parameters)
            flat.finish_backward()
            flat.clip_grad_norm_(clip)
            opt.step()
            if flat.last_step_skipped:
                raise RuntimeError("a TTT-MLP backward hand-over timed out (or the gradient norm is not finite): step skipped")
        else:
            if tp and no_fsdp:
                tp_sync_gradients(model)            # partial parameter gradients (a rank's tokens / heads) summed over the group
            if replica:
                replica.collect_grads()             # bf16 gradients -> fp32 gradients of the masters (what FSDP's reduce does)
            if checked_optimizer_step(opt, train_params, clip, on_skip=replica.zero_grad if replica else None) is None:
                raise RuntimeError("a TTT-MLP backward hand-over timed out (or the gradient norm is not finite): step skipped")
            if replica:
                replica.publish()                   # fp32 masters -> bf16 compute copies (what FSDP's all-gather does)
        lr_sched.step()
        return loss

    # ---- activation re-materialisation sized for this GPU (untimed), warm-up, timed region: size_warm_and_time() ------------
    dit = model.dit if hasattr(model, "dit") else model
    fast0 = [0]
    offload = None
    if args.offload_gib_per_layer > 0 or args.offload_park_kept:
        from ttt_amd.infra.host_offload import HostOffload
        offload = dit.host_offload = HostOffload(int(args.offload_gib_per_layer * 2 ** 30), layers=args.offload_layers, park_kept=args.offload_park_kept,
                                                 max_backlog_bytes=int(args.offload_backlog_gib * 2 ** 30), lookahead=args.offload_lookahead,
                                                 soft_limit_bytes=int(args.offload_soft_frac * torch.cuda.get_device_properties(dev).total_memory))

        offload.batch = not args.offload_no_batch
        offload.one_stream, offload.blocking_end = not args.offload_two_streams, args.offload_blocking_end

    class Hooks:
        oom = torch.cuda.OutOfMemoryError
        total_memory = torch.cuda.get_device_properties(dev).total_memory
        num_layers = cfg.num_layers
        probe_layers = (4 if frames <= 13 and LB == 1 else 1) if cfg.num_layers >= 8 else 0

        @staticmethod
        def set_free_layers(n):
            dit.remat_free_layers = n

        @staticmethod
        def reset_peak():
            torch.cuda.reset_peak_memory_stats()

        @staticmethod
        def max_allocated():
            return torch.cuda.max_memory_allocated()

        @staticmethod
        def synchronize():
            torch.cuda.synchronize()

        @staticmethod
        def release():
            opt.zero_grad(set_to_none=True)
            if replica:
                replica.zero_grad()
            if flat:
                flat.zero_grad()
            torch.cuda.empty_cache()

        @staticmethod
        def reduce_keep():
            """drop the most expensive kind of kept outputs ("scan" 1.3 GB per layer at 9 s, then "fc2" 0.32 GB, then "attn" 0.34 GB)"""
            if not dit.remat_keep:
                return False
            drop = ([k for k in ("scan", "fc2", "attn") if k in dit.remat_keep] or list(dit.remat_keep))[0]      # least ms per GB first
            dit.remat_keep = tuple(k for k in dit.remat_keep if k != drop)
            log(f"out of memory with every layer re-materialised: keeping {list(dit.remat_keep) or 'nothing'} instead")
            return True

        @staticmethod
        def alloc_retries():
            return int(torch.cuda.memory_stats(dev).get("num_alloc_retries", 0))

        @staticmethod
        def all_reduce_min(v):
            t = torch.tensor([v], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t)

        @staticmethod
        def barrier():
            dist.barrier(device_ids=[local_rank])

        @staticmethod
        def before_timed(n_free):
            fast0[0] = ext.sweep_fast_count()           # (synchronises: outside the timed region)
            if rank == 0:
                log(f"timed region: {args.steps} steps, remat_free_layers={n_free}")
            timer.reset()
            timer.active = True
            if offload:
                offload.stats.clear()
            if clocks:
                clocks.start()

        @staticmethod
        def after_timed():
            timer.active = False
            if clocks:
                clocks.stop()

    n_free, dt, loss = size_warm_and_time(step, Hooks, args.remat_free_layers, args.warmup, args.steps, world)
    fast0 = fast0[0]
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss_val = float(loss.detach())
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30
    peaks = [torch.zeros(2, device=dev, dtype=torch.float64) for _ in range(world)]      # (allocated, reserved) GiB of every rank
    dist.all_gather(peaks, torch.tensor([peak_mem, torch.cuda.max_memory_reserved() / 2 ** 30], device=dev, dtype=torch.float64))
    fast_wgs = ext.sweep_fast_count() - fast0
    # a cluster hand-over of the TTT-MLP backward that gave up inside the timed region poisons that step's gradients (NaN) and
    # makes the next extension call raise; a line measured with one is not a measurement (synchronises: region is over)
    sweep_err = torch.tensor([ext.sweep_error()], device=dev, dtype=torch.int64)
    dist.all_reduce(sweep_err, op=dist.ReduceOp.MAX)
    sweep_err = int(sweep_err)
    if args.torch_profile and world == 1:        # (one process only: a lone extra step would hang the others' collectives)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
            step()
            torch.cuda.synchronize()
        with open(args.torch_profile, "w") as fh:
            fh.write(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=70))
            fh.write("\n\n")
            fh.write(prof.key_averages(group_by_input_shape=True, group_by_stack_n=8).table(
                sort_by="self_cuda_time_total", row_limit=80, max_name_column_width=60, max_src_column_width=110, max_shapes_column_width=70))

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        video_tokens = dp * LB * frames * TOKENS_PER_FRAME
        value = video_tokens / (dt / args.steps)
        ks = timer.summary()
        # ---- roofline of the dominant hand-written kernel ---------------------------------
        B, NH, NC, CS, F = LB, cfg.num_heads, L // cfg.mini_batch_size, cfg.mini_batch_size, cfg.head_dim
        if args.ssm_layer == "ttt_mlp":
            gemm = 2.0 * CS * F * 4 * F
            flops = {"fwd": 7 * gemm, "bwd": 14 * gemm}         # algorithmic; the in-kernel recompute (7g) is excluded
        else:
            gemm = 2.0 * CS * F * F
            flops = {"fwd": 3 * gemm, "bwd": 6 * gemm}
        # local attention (hand-written MFMA kernels too): algorithmic 4 S^2 D NH forward, 2.5x that backward
        seg_S = 13 * TOKENS_PER_FRAME + text_len
        a_flops = 4.0 * LB * seg_S * seg_S * cfg.head_dim * cfg.num_heads
        flops["attn_fwd"], flops["attn_bwd"] = a_flops / (B * NH * NC), 2.5 * a_flops / (B * NH * NC)     # per (b,h,step) units like the scan
        scan_keys = [k for k in ks if k in ("fwd", "bwd")]
        Kg = -(-NC // max(1, min(cfg.scan_checkpoint_group_size, NC)))                  # checkpoint groups
        gpc = max(1, min(256 // (B * NH) if B * NH < 256 else 1, Kg))                   # groups per backward chunk (csrc/ttt_mfma_bwd2.hip)
        sweep_chunks = -(-Kg // gpc)
        dom = max(scan_keys, key=lambda k: ks[k]["total_ms"]) if scan_keys else None
        roof = None
        if dom:
            per_launch = B * NH * NC * flops[dom]
            ach = per_launch / (ks[dom]["avg_ms"] * 1e-3) / 1e12
            impl = ext.resolved_impl(B, NH, NC, CS, F, min(cfg.scan_checkpoint_group_size, NC), torch.bfloat16,
                                     mlp=args.ssm_layer == "ttt_mlp", backward=dom == "bwd")
            kname = f"{args.ssm_layer}_{dom}_scan[{impl}]"
            traffic, traffic_src = pmc_traffic(kname, B, NH, NC)
            # algorithmic bytes per launch: forward Q, K, V, out tiles + eta + one checkpoint per G steps;
            # backward Q, K, V, dOut in, dQ, dK, dV out + eta, d(eta) + one checkpoint read per G steps
            tile_b, ck_b = CS * F * 2, ((2 * F * 4 * F + 4 * F + F) * 4 if args.ssm_layer == "ttt_mlp" else (F * F + F) * 4)
            per_step = ((4 * tile_b + CS * 2) if dom == "fwd" else (7 * tile_b + 2 * CS * 2)) + ck_b / max(1, min(cfg.scan_checkpoint_group_size, NC))
            alg_bytes = B * NH * NC * per_step
            roof = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                    "flops_per_launch": per_launch, "avg_launch_ms": ks[dom]["avg_ms"], "launches_timed": ks[dom]["launches"],
                    # CUs the timed kernel actually occupies: one workgroup per (b,h) for the forward scans, a cluster of FOUR
                    # workgroups (one per CU) per (b,h) for the TTT-MLP backward sweep at mini-batches of 64
                    "occupied_cus": (occ_cus := min((4 if (dom == "bwd" and args.ssm_layer == "ttt_mlp" and CS == 64) else 1) * B * NH, 256)),
                    "occupied_cu_frac": ach / (MFMA_BF16_PEAK_TFLOPS * occ_cus / 256.0),
                    # which roof the kernel leans on: counter bytes / launch time against the HBM peak
                    "hbm_gbps": (traffic / (ks[dom]["avg_ms"] * 1e-3) / 1e9) if traffic else None,
                    "hbm_frac": (traffic / (ks[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "other": {k: {"avg_ms": v["avg_ms"], "launches": v["launches"],
                                  "achieved_tflops": B * NH * NC * flops[k] / (v["avg_ms"] * 1e-3) / 1e12} for k, v in ks.items() if k != dom},
                    "scan_share_of_step": sum(ks[k]["total_ms"] for k in scan_keys) / (1e3 * dt),
                    # TTT-MLP backward sweep: cluster workgroups of the timed region that PROVED same-XCD placement and used
                    # plain (L2-resident) hand-over records, out of all launched (4 per (b,h) and chunk)
                    "sweep_same_xcd_frac": (fast_wgs / (ks["bwd"]["launches"] * sweep_chunks * 4 * B * NH)
                                            if args.ssm_layer == "ttt_mlp" and "bwd" in ks and CS == 64 else None),
                    "attention_share_of_step": sum(v["total_ms"] for k, v in ks.items() if k.startswith("attn")) / (1e3 * dt)}
            # the other hand-written kernels as FLAT scalars too (the driver's record keeps roofline's scalars, not `other`)
            for k, v in roof["other"].items():
                name = {"fwd": "scan_fwd", "bwd": "scan_bwd"}.get(k, k)
                roof[f"{name}_ms"] = round(v["avg_ms"], 4)
                roof[f"{name}_tflops"] = round(v["achieved_tflops"], 2)
                roof[f"{name}_frac"] = round(v["achieved_tflops"] / MFMA_BF16_PEAK_TFLOPS, 5)
        line = {"metric": "DiT+TTT fwd/bwd video-tokens/sec", "value": value, "unit": "video-tokens/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
                "scaling": "strong" if (tp and dp == 1 and world > 1) else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"CogVideoX-5B+{args.ssm_layer} {args.video_length} training step (fwd+bwd+AdamW), "
                                       f"{cfg.num_layers} layers, L={L} tokens/sample, adapter={args.adapter}",
                           "global_batch": dp * LB, "seq_len": L, "parallelism": (f"tp{tp}" if no_fsdp else f"fsdp{world}(dp{dp}xtp{tp})") if tp else ("replica1" if no_fsdp else (f"flat_fsdp{world}" if (communicate or world > 1) else "flat1") if flat else f"fsdp2_{world}"), "ttt_impl": args.impl,
                           "remat_free_layers": n_free, "remat_keep": list(dit.remat_keep), "remat_keep_layers": args.remat_keep_layers, "remat_keep_limits": dict(dit.remat_keep_limits) or None, "ttt_pipeline_parts": parts_used, "fsdp_reshard_after_forward": bool(args.reshard_after_forward), "tuned_gemm_selections": bool(tuned),
                           "debug_options": ",".join(args.debug_option) or None,
                           "host_offload": ({"gib_per_layer": args.offload_gib_per_layer, "layers": args.offload_layers, "park_kept": args.offload_park_kept,
                                             "gib_per_step": round(offload.stats["offloaded_bytes"] / args.steps / 2 ** 30, 2),
                                             "storages_per_step": offload.stats["offloaded_storages"] // args.steps,
                                             "kept_on_device": offload.stats["kept_on_device"], "late_fetches": offload.stats["late_fetches"],
                                             "throttle_waits": offload.stats["throttle_waits"], "lookahead": args.offload_lookahead, "backlog_gib": args.offload_backlog_gib,
                                             "trace": offload.trace_summary() if offload.trace is not None else None,
                                             "host_s_per_step": {k[7:]: round(v / args.steps, 3) for k, v in offload.stats.items() if k.startswith("host_s_")},
                                             "pinned_gib": round(sum(t.numel() for t in offload._slots) / 2 ** 30, 1)} if offload else None),
                           "sweep_error": sweep_err, "valid": args.layers is None and sweep_err == 0,
                           **({k: v for k, v in clocks.summary().items() if k in ("clock_mhz_avg", "clock_mhz_min", "clock_mhz_max", "power_w_avg", "power_w_max")} if clocks else {}),
                           "clocks": clocks.summary() if clocks else None},
                "roofline": roof, "loss": loss_val, "peak_mem_gib": peak_mem, "peak_reserved_gib": torch.cuda.max_memory_reserved() / 2 ** 30, "alloc_retries_total": Hooks.alloc_retries(), "total_tokens_per_s": dp * L / (dt / args.steps),
                "peak_mem_gib_per_rank": [[round(float(x), 1) for x in pk] for pk in peaks],
                "optimizer": {"groups": [c.group_name for c in sched_cfgs], "lr": oc["lr"], "lr_ssm": oc["lr_ssm"], "clip": clip,
                              "weight_decay": [g_["weight_decay"] for g_ in opt.param_groups]}}
        if flat:
            flat.remove()
        return line
    if flat:
        flat.remove()
    return None


if __name__ == "__main__":
    main()
