// MFMA TTT-MLP backward for gfx950 (revision 4, round 3): host-side orchestration.
//
// A backward call walks the sequence in CHUNKS of `gpc` checkpoint groups, last chunk first; per chunk three launches:
//   A  group recompute (ttt_mfma_rc4.hip): one workgroup per (b, h, group) re-runs the group's forward from its checkpoint in the
//      8-wave register-resident form and stores the SLIM step record (ttt_bwd4_dev.h: Z1, Z1b, gZ2, the LayerNorm rows);
//   B  reverse sweep (ttt_mfma_bwd4.hip): sequential over the chunk's steps, carrying dW1 / dW2 / db1 / db2 / dgamma / dbeta; FOUR
//      workgroups per (b, h) with role-specialised waves (compute, owners, derivers), at most n_cu / 4 (b, h) per launch;
//   C  tail (ttt_mfma_bwd4.hip): dK and dQ need the carried dW1 and the step's dZ1 but nothing downstream needs them, so the
//      sweep stores those and this fully parallel kernel (one workgroup per step) finishes
//      dK = -eta (gZ1 dW1'^T) + dZ1 W1^T - dt   and   dQ = dOut + dZ1b W1'^T.
// Schedules (debug option "overlap_tail"; measured on one MI355X at NC = 804, profiles/r3g_*): 0 = one stream, 15.8 ms;
// 1 (default) = the tail of chunk c on a side stream beside the sweep of chunk c-1 (two record buffers), 14.3 ms.  (The recompute
// of chunk c-2 beside that sweep too measured 14.6 ms - it hides 0.17 ms per chunk and slows the sweep, which is bound by what its
// CUs' memory pipelines move, by as much - and is gone.)
// History: revisions 1 - 3 (4-wave sweep; 8-wave single-workgroup sweep; cluster sweep over 570-KiB register-image records)
// were removed in rounds 2 and 3, each after losing its A/B on hardware; revision 2's sweep was also found to be inaccurate on
// model-like inputs.
// Math: SURVEY.md Appendix A backward; oracle/ttt_oracle.py:_mlp_step_bwd is the executable spec.
#include "ttt_mfma.h"
#include "ttt_mfma_dev.h"
#include "ttt_mfma_int.h"
#include "ttt_mfma_bwd_dev.h"
#include "ttt_bwd4_dev.h"
#include <mutex>

namespace ttt {
namespace mfma {
using namespace ttt::mf;


// ---------------------------------------------------------------------------------------------------------------------------
bool bwd_available() { return true; }

static int g_forced_gpc = 0;
void set_debug_groups_per_chunk(int g) { g_forced_gpc = g; }
static int g_overlap = 2;             // 1: tail of chunk c on a side stream beside the sweep of chunk c-1; 0 = one stream;
                                      // 2: the recompute of chunk c-1 too beside the sweep of chunk c (THREE record buffers; round 6 A/B)
void set_debug_overlap_tail(int v) { g_overlap = v; }
static int record_buffers() { return g_overlap >= 2 ? 3 : 2; }
static int g_fast_records = 1;        // cluster hand-over: plain (L2-resident) records once same-XCD placement is proven; 0 = always write-through
void set_debug_fast_records(int v) { g_fast_records = v; }
int get_debug_fast_records() { return g_fast_records; }
// Decided by A/Bs on hardware and no longer options (rounds 3 - 5): non-temporal stores of the step records (14.28 against 14.68 ms per
// backward), bf16 inner-LayerNorm owner rows (11.35 against 11.63), L2 prefetch touches two steps ahead (14.68 against 16.03), and
// the hand-over flags of the NEXT sweep cleared right behind the current one instead of in front of the next: a rocprofv3 trace of
// the training step (profiles/r4y_ttt_bwd_launches.csv.gz, tools/sweep_launches.py) showed the sweep starting 13 - 15 us after its
// recompute - the memset - and the previous chunk's tail, released by an event behind the same recompute, after the same latency:
// a coin toss who is dispatched first, and "tail first" is the slow outcome (1.00 - 1.12 against 0.92 ms).  With the memset out of
// the way the sweep follows its recompute kernel-to-kernel; round 4's driver line confirmed it inside the step (the sharded
// `fsdp1` point within 0.1 % of the replica line, where it had been 2.4 % behind).  The gate kernels tried beside it are gone.
static int g_deriver_split = 1;       // sweep: barrier Bc inside the derivers' reverse step (round 6); 0 = behind it (rounds 3 - 5)
void set_debug_deriver_split(int v) { g_deriver_split = v; }
static int g_sweep_fault = 0;         // DEBUG fault injection (tests of the hand-over failure path)
void set_debug_sweep_fault(int v) { g_sweep_fault = v; }

// Compute units of the current device (cached per device): the cluster sweep needs its four workgroups co-resident, one per
// CU (157 KiB of LDS each), so a launch carries at most n_cu / 4 clusters.
static std::mutex g_dev_mutex;         // guards the per-device caches below (two autograd threads may enter with one device each - or the same)
static int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus[dev] = n;
    }
    return cus[dev];
}
int sweep_clusters_per_launch() { return device_cus() / 4; }
int device_cu_count() { return device_cus(); }

int groups_per_chunk(const ttt_dims* d) {
    const int nbh = d->B * d->NH;
    const int K = (d->NC + d->G - 1) / d->G;
    // recompute workgroups (one per (b,h,group), one per CU: 135 KiB of LDS) should fill the 256 CUs in ONE wave: with
    // ceil(256/nbh) groups (288 workgroups at nbh = 48) the last 32 run alone and the launch takes twice as long
    int g = nbh < 256 ? 256 / nbh : 1;
    if (g_forced_gpc > 0) g = g_forced_gpc;   // DEBUG knob (tests exercise the chunk hand-over at small sizes)
    // bound the slot area to ~4 GiB
    const size_t per_group = (size_t)nbh * d->G * s4::SLOT4_BYTES;
    const size_t cap = (size_t)4 << 30;
    while (g > 1 && per_group * g > cap) --g;
    if (g > K) g = K;
    return g < 1 ? 1 : g;
}

static size_t align128(size_t v) { return (v + 127) & ~(size_t)127; }

size_t workspace_bytes(const ttt_dims* d, bool mlp, bool backward) {
    if (mlp && !backward && d->CS == 64) return scan_pair_workspace_bytes(d->B * d->NH);     // the pair scan's ring + flag lines
    if (!mlp || !backward) return 0;
    const size_t nbh = (size_t)d->B * d->NH;
    const size_t slots = (size_t)groups_per_chunk(d) * d->G + 1;
    // two record buffers + carried state gradient + exchange records and flag lines of the cluster sweep + the state after the
    // last step + the derivers' parking areas
    return nbh * ((size_t)record_buffers() * slots * s4::SLOT4_BYTES + b2::CARRY_FLOATS2 * sizeof(float)) + align128(nbh * 64) +
           nbh * (b2::XCH_BH_BYTES + 4 * b2::FLAG_STRIDE * sizeof(unsigned)) + nbh * (s4::FINAL_FLOATS * sizeof(float) + 8 * s4::PARK4_BYTES) +
           nbh * (size_t)((d->NC + d->G - 1) / d->G) * 64 * 256 * sizeof(float);                      // dW1 anchors of the group-sequential tail
}

// Side stream and the events of the two-buffer hand-over, one set per device, created on first use.
struct OverlapRes {
    hipStream_t side = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr}, tail_done[2] = {nullptr, nullptr}, entry = nullptr;
    hipEvent_t rc_done[3] = {nullptr, nullptr, nullptr}, sweep_done[3] = {nullptr, nullptr, nullptr};     // schedule 2
    int state = 0;                      // 0 = not tried, 1 = usable, -1 = creation failed (one stream from then on)
};
static OverlapRes* overlap_resources() {
    static OverlapRes res[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    OverlapRes& r = res[dev];
    if (r.state == 0) {
        bool ok = hipStreamCreateWithFlags(&r.side, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&r.entry, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < 2; ++i) {
            ok = ok && hipEventCreateWithFlags(&r.ready[i], hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&r.tail_done[i], hipEventDisableTiming) == hipSuccess;
        }
        for (int i = 0; ok && i < 3; ++i) {
            ok = ok && hipEventCreateWithFlags(&r.rc_done[i], hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&r.sweep_done[i], hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) {                      // usable only when EVERY object exists: release what was created
            for (int i = 0; i < 3; ++i) {
                if (r.rc_done[i]) (void)hipEventDestroy(r.rc_done[i]);
                if (r.sweep_done[i]) (void)hipEventDestroy(r.sweep_done[i]);
                r.rc_done[i] = r.sweep_done[i] = nullptr;
            }
            for (int i = 0; i < 2; ++i) {
                if (r.ready[i]) (void)hipEventDestroy(r.ready[i]);
                if (r.tail_done[i]) (void)hipEventDestroy(r.tail_done[i]);
                r.ready[i] = r.tail_done[i] = nullptr;
            }
            if (r.entry) (void)hipEventDestroy(r.entry);
            r.entry = nullptr;
            if (r.side) (void)hipStreamDestroy(r.side);
            r.side = nullptr;
            (void)hipGetLastError();
        }
        r.state = ok ? 1 : -1;
    }
    return r.state == 1 ? &r : nullptr;
}

// Revision 4 (round 3): the same chunk walk and the same two-stream schedule over the slim step record:
//   A  s4::launch_recompute4   (ttt_mfma_rc4.hip: 8-wave recompute, 120.5 KiB per step)
//   B  s4::launch_sweep_cluster4 (ttt_mfma_bwd4.hip: cluster sweep with deriver waves)
//   C  s4::launch_tail4        (dK / dQ; beside the next chunk's sweep)
static int mlp_backward4(const ttt_dims* d, const ttt_mlp_bwd_args* a, void* ws, hipStream_t s, int per_launch, unsigned* err_word) {
    const int nbh = d->B * d->NH, G = d->G, NC = d->NC;
    const int K = (NC + G - 1) / G;
    const int gpc = groups_per_chunk(d);
    const size_t slot_stride = ((size_t)gpc * G + 1) * s4::SLOT4_BYTES;
    char* slots = (char*)ws;                                   // two (schedule 2: three) buffers of nbh * slot_stride bytes
    const size_t slot_buf = (size_t)nbh * slot_stride;
    const int NBUF = record_buffers();                         // chunk c lives in buffer c % NBUF
    float* carry = (float*)(slots + (size_t)NBUF * slot_buf);
    char* xch = (char*)(carry + (size_t)nbh * b2::CARRY_FLOATS2) + align128((size_t)nbh * 64);
    unsigned* flags = (unsigned*)(xch + (size_t)nbh * b2::XCH_BH_BYTES);
    const size_t flag_bytes = (size_t)nbh * 4 * b2::FLAG_STRIDE * sizeof(unsigned);
    float* wfinal = (float*)((char*)flags + flag_bytes);
    char* park = (char*)(wfinal + (size_t)nbh * s4::FINAL_FLOATS);
    float* danchor = (float*)(park + (size_t)nbh * 8 * s4::PARK4_BYTES);

    s4::RecomputeParams rp = {};
    rp.XQ = (const __bf16*)a->XQ; rp.XK = (const __bf16*)a->XK; rp.XV = (const __bf16*)a->XV; rp.eta = (const __bf16*)a->last_eta;
    rp.ln_w = a->ttt_norm_weight; rp.ln_b = a->ttt_norm_bias;
    rp.W1c = a->W1_checkpoints; rp.b1c = a->b1_checkpoints; rp.W2c = a->W2_checkpoints; rp.b2c = a->b2_checkpoints;
    rp.slot_stride_bh = slot_stride; rp.wfinal = wfinal;
    rp.NH = d->NH; rp.NC = NC; rp.G = G; rp.K = K; rp.eps = d->eps; rp.nt = 1; rp.own16 = 1;

    s4::SweepParams4 bp = {};
    bp.XQ = (const __bf16*)a->XQ; bp.XK = (const __bf16*)a->XK; bp.dOut = (const __bf16*)a->grad_L_XQW; bp.eta = (const __bf16*)a->last_eta;
    bp.ln_w = a->ttt_norm_weight;
    bp.uW1 = a->grad_L_W1_last; bp.ub1 = a->grad_L_b1_last; bp.uW2 = a->grad_L_W2_last; bp.ub2 = a->grad_L_b2_last;
    bp.slot_stride_bh = slot_stride; bp.carry = carry;
    bp.dXV = (__bf16*)a->grad_L_XV; bp.deta = (__bf16*)a->grad_L_last_eta;
    bp.dW1 = a->grad_L_W1_init; bp.db1 = a->grad_L_b1_init; bp.dW2 = a->grad_L_W2_init; bp.db2 = a->grad_L_b2_init;
    bp.dlnw = a->grad_L_ttt_norm_weight; bp.dlnb = a->grad_L_ttt_norm_bias;
    bp.NH = d->NH; bp.NC = NC;
    bp.xch = xch; bp.flags = flags; bp.fast_records = g_fast_records;
    bp.err = err_word; bp.fault = g_sweep_fault;
    bp.W1c = a->W1_checkpoints; bp.W2c = a->W2_checkpoints; bp.wfinal = wfinal; bp.park = park; bp.G = G; bp.K = K; bp.danchor = danchor; bp.prefetch = 1; bp.own16 = 1; bp.split = g_deriver_split ? 1 : 0;

    int rc = 0;                               // a failed event / stream call leaves the two streams unordered: the call fails (-12)
    auto chk = [&](hipError_t e) { if (e != hipSuccess) rc = -12; };
    const int nchunks = (K + gpc - 1) / gpc;
    // Chunk ch covers checkpoint groups [chunk_g0(ch), + chunk_ng(ch)).  The RAGGED chunk is chunk 0 - the one processed LAST - since round 6: with
    // the ragged chunk at the end of the sequence (processed first: 1 group of 51 at the 9 s length) the first sweep lasted 58 us and the second
    // chunk's recompute - on the side stream, in launches of 64 workgroups - ran 0.5 ms with nothing beside it (profiles/r6tl_bwd_timeline_nc804.txt).
    const int r0 = K - (nchunks - 1) * gpc;
    auto chunk_g0 = [&](int ch) { return ch == 0 ? 0 : r0 + (ch - 1) * gpc; };
    auto chunk_ng = [&](int ch) { return ch == 0 ? r0 : gpc; };
    OverlapRes* ov = (g_overlap && nchunks > 1) ? overlap_resources() : nullptr;
    const int free_cus = device_cus() - 4 * (nbh < per_launch ? nbh : per_launch);
    if (ov && (free_cus < 32 || nbh > per_launch)) ov = nullptr;      // nothing free beside the sweep, or several sweep launches per chunk
    auto recompute = [&](int ch, int max_wg, hipStream_t st) {
        const int g0 = chunk_g0(ch), ng = chunk_ng(ch);
        rp.chunk_group0 = g0; rp.chunk_groups = ng; rp.chunk_lo = g0 * G;
        rp.slots = slots + (size_t)(ch % NBUF) * slot_buf;
        s4::launch_recompute4(rp, nbh, max_wg, st);
    };
    auto tail = [&](int ch, hipStream_t st) {
        const int g0 = chunk_g0(ch), ng = chunk_ng(ch);
        const int lo = g0 * G;
        s4::Tail5Args ta = {(const __bf16*)a->XQ, (const __bf16*)a->XK, (const __bf16*)a->grad_L_XQW, (const __bf16*)a->last_eta,
                            (const __bf16*)a->grad_L_XV, slots + (size_t)(ch % NBUF) * slot_buf, slot_stride, a->W1_checkpoints, wfinal, danchor,
                            (__bf16*)a->grad_L_XQ, (__bf16*)a->grad_L_XK, NC, G, K, lo, g0, ng};
        s4::launch_tail5(ta, nbh, st);
    };
    auto sweep = [&](int ch) {
        const int g0 = chunk_g0(ch), ng = chunk_ng(ch);
        bp.slots = slots + (size_t)(ch % NBUF) * slot_buf;
        bp.chunk_lo = g0 * G;
        bp.chunk_hi = ((g0 + ng) * G < NC) ? (g0 + ng) * G : NC;
        bp.first = (ch == nchunks - 1);
        bp.last = (ch == 0);
        bp.dbg = get_debug_timing();
        if (ch == nchunks - 1) chk(hipMemsetAsync(flags, 0, flag_bytes, s));         // hand-over flags restart at 0 for every launch
        for (int bh0 = 0; bh0 < nbh; bh0 += per_launch) {
            bp.bh0 = bh0;
            bp.nbh = nbh - bh0 < per_launch ? nbh - bh0 : per_launch;
            s4::launch_sweep_cluster4(bp, bp.nbh, s);
        }
        if (ch > 0) chk(hipMemsetAsync(flags, 0, flag_bytes, s));   // for sweep(ch - 1): behind this sweep, in front of its recompute
    };
    if (ov) {           // the side stream starts after everything queued on `s` before this call (the inputs), not after A(n-1)
        chk(hipEventRecord(ov->entry, s));
        chk(hipStreamWaitEvent(ov->side, ov->entry, 0));
    }
    recompute(nchunks - 1, 0, s);
    // (Round 6 also tried schedule 3 - the recompute beside the sweep as in schedule 2, the tail BEHIND its sweep on the whole chip, where it takes
    // 0.06 ms against 0.26 ms beside a sweep - and it LOST: 9.75 - 9.80 against 9.54 - 9.57 ms per backward alone, 9.53 - 9.55 against 9.25 - 9.30
    // inside the step; profiles/r6s3_*.  Gone.)
    if (ov && NBUF == 3) {
        // Schedule 2 (round 6 A/B): stream s carries the sweeps only, A(n-1) B(n-1) B(n-2) ... ; beside B(c) the side stream runs
        // A(c-1) - in launches of at most the CUs the sweep leaves free, so that a recompute workgroup never holds a CU a cluster
        // member of the NEXT sweep is waiting for - and then C(c+1).  Chunk c lives in buffer c % 3: A(c-1) writes, B(c) reads,
        // C(c+1) reads three different buffers; A(c-1) follows C(c+2) (same buffer) in stream order.
        // (Round 6 also tried the tails on a second, low-priority side stream - on one side stream the recompute of chunk c - 2 queues behind
        // the tail of chunk c + 1, and recompute + tail, 0.66 + 0.26 ms on the 64 free CUs, are longer than a sweep - and LOST: 10.88 against
        // 10.38 ms per backward alone, and the whole training step 10 % slower, attention kernels included; profiles/r6p_*.  Gone.)
        chk(hipEventRecord(ov->sweep_done[nchunks % 3], s));                       // "B(n) is done" = A(n-1) is complete
        for (int ch = nchunks - 1; ch >= 0; --ch) {
            if (ch + 1 < nchunks) chk(hipStreamWaitEvent(s, ov->rc_done[ch % 3], 0));
            sweep(ch);
            chk(hipEventRecord(ov->sweep_done[ch % 3], s));
            chk(hipStreamWaitEvent(ov->side, ov->sweep_done[(ch + 1) % 3], 0));    // beside B(ch): released when B(ch + 1) is done
            if (ch > 0) {
                recompute(ch - 1, free_cus, ov->side);
                chk(hipEventRecord(ov->rc_done[(ch - 1) % 3], ov->side));
            }
            if (ch + 1 < nchunks) tail(ch + 1, ov->side);
        }
        chk(hipStreamWaitEvent(ov->side, ov->sweep_done[0], 0));
        tail(0, ov->side);
        chk(hipEventRecord(ov->tail_done[0], ov->side));
        chk(hipStreamWaitEvent(s, ov->tail_done[0], 0));
        return rc;
    }
    if (!ov) {          // one stream: A(c) B(c) C(c) per chunk (chunk c in slot buffer c % NBUF)
        for (int ch = nchunks - 1; ch >= 0; --ch) {
            sweep(ch);
            if (ch > 0) recompute(ch - 1, 0, s);
            tail(ch, s);
        }
        return rc;
    }
    {
        // Tail beside the next sweep only (round 2's schedule): stream s: A(n-1) B(n-1) A(n-2) B(n-2) ... ; side: C(c) beside B(c-1).
        // C(c) is released when A(c-1) is complete - the moment B(c-1) starts -, and A(c-2), which overwrites C(c)'s buffer, waits.
        for (int ch = nchunks - 1; ch >= 0; --ch) {
            const int buf = ch & 1;
            sweep(ch);
            if (ch > 0) {
                if (ch + 1 < nchunks) chk(hipStreamWaitEvent(s, ov->tail_done[buf ^ 1], 0));
                recompute(ch - 1, 0, s);
            }
            chk(hipEventRecord(ov->ready[buf], s));
            chk(hipStreamWaitEvent(ov->side, ov->ready[buf], 0));
            tail(ch, ov->side);
            chk(hipEventRecord(ov->tail_done[buf], ov->side));
        }
        chk(hipStreamWaitEvent(s, ov->tail_done[0], 0));
        return rc;
    }
}

int mlp_backward(const ttt_dims* d, const ttt_mlp_bwd_args* a, void* ws, hipStream_t s) {
    const int nbh = d->B * d->NH, G = d->G, NC = d->NC;
    const int per_launch = sweep_clusters_per_launch();
    if (per_launch < 1) return -10;      // fewer than 4 compute units visible: the cluster sweep cannot be co-resident
    unsigned* err_word = sweep_error_word();
    if (!err_word) return -11;
    return mlp_backward4(d, a, ws, s, per_launch, err_word);
}

}  // namespace mfma
}  // namespace ttt
