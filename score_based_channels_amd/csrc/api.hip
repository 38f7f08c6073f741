// C ABI of libsbc_hip.so (include/sbc_hip.h): error reporting, single-op launch, plans (eager or hipGraph
// replay), per-tag kernel timing, host-side weight packing.
#include <cmath>
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include "common.h"

namespace sbc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_dyn_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> high;      // (device, kernel) -> limit already set
    int dev = 0;
    SBC_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = high[std::make_pair(dev, kernel)];
    if (bytes > cur) {
        SBC_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        cur = bytes;
    }
    return SBC_OK;
}

// One 4-byte flag word per device for the f16x2 kernels (include/sbc_hip.h: sbc_range_flag).  Allocated outside any stream
// capture: sbc_plan_create resolves every convolution once (dry run) before a plan can be captured.
int range_flag_ptr(unsigned** out) {
    static std::mutex mu;
    static std::map<int, unsigned*> words;
    int dev = 0;
    SBC_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    unsigned*& w = words[dev];
    if (!w) {
        SBC_CHECK_HIP(hipMalloc((void**)&w, 16));
        SBC_CHECK_HIP(hipMemset(w, 0, 16));
    }
    *out = w;
    return SBC_OK;
}

// kind-specific extension structs, copied at plan creation (sbc_op.ext is host memory the caller may free)
union OpExt {
    sbc_langevin lang;     // LANGEVIN / MEASURE
    sbc_endconv endc;      // END_CONV / END_CONV_BWD
    sbc_dsm dsm;           // DSM_PERTURB / DSM_LOSS
    sbc_adam adam;         // ADAM_EMA
    sbc_chain chain;       // CHAIN
};

struct PlanOp {
    sbc_op op;
    OpExt ext;
};

// size of the extension struct an op kind carries (0 = none)
static size_t ext_size(int kind) {
    switch (kind) {
        case SBC_OP_LANGEVIN: case SBC_OP_MEASURE: return sizeof(sbc_langevin);
        case SBC_OP_END_CONV: case SBC_OP_END_CONV_BWD: return sizeof(sbc_endconv);
        case SBC_OP_DSM_PERTURB: case SBC_OP_DSM_LOSS: return sizeof(sbc_dsm);
        case SBC_OP_ADAM_EMA: return sizeof(sbc_adam);
        case SBC_OP_CHAIN: return sizeof(sbc_chain);
        default: return 0;
    }
}

static int dispatch(const sbc_op& op, const void* ext, hipStream_t s) {
    const sbc_langevin* lang = (const sbc_langevin*)ext;
    const sbc_endconv* endc = (const sbc_endconv*)ext;
    SBC_REQUIRE(ext || ext_size(op.kind) == 0, "op kind %d: ext must be set", op.kind);
    switch (op.kind) {
        case SBC_OP_DSM_PERTURB: return launch_dsm_perturb(op, *(const sbc_dsm*)ext, s);
        case SBC_OP_DSM_LOSS: return launch_dsm_loss(op, *(const sbc_dsm*)ext, s);
        case SBC_OP_GRAD_ADD: return launch_grad_add(op, s);
        case SBC_OP_INORM_BWD: return launch_inorm_bwd(op, s);
        case SBC_OP_MAXPOOL5_BWD: return launch_maxpool5_bwd(op, s);
        case SBC_OP_UPSAMPLE_BWD: return launch_upsample_bwd(op, s);
        case SBC_OP_POOL_BWD: return launch_pool_bwd(op, s);
        case SBC_OP_CONV_WGRAD: return launch_conv_wgrad(op, s);
        case SBC_OP_PACK_WEIGHT: return launch_pack_weight(op, s);
        case SBC_OP_END_CONV_BWD: return launch_end_conv_bwd(op, *endc, s);
        case SBC_OP_BEGIN_CONV_BWD: return launch_begin_conv_bwd(op, s);
        case SBC_OP_ADAM_EMA: return launch_adam_ema(op, *(const sbc_adam*)ext, s);
        case SBC_OP_BEGIN_CONV: return launch_begin_conv(op, s);
        case SBC_OP_INORM_STATS: return launch_inorm_stats(op, s);
        case SBC_OP_CONV: return launch_conv(op, s);
        case SBC_OP_CONV_PAIR: return launch_conv_pair(op, s);
        case SBC_OP_CONV_POOL: return launch_conv_pool(op, s);
        case SBC_OP_RES_BLOCK: return launch_res_block(op, s);
        case SBC_OP_CHAIN: return launch_chain(op, *(const sbc_chain*)ext, s);
        case SBC_OP_CONV_DOWN: return launch_conv_down(op, s);
        case SBC_OP_MAXPOOL5: return launch_maxpool5(op, s);
        case SBC_OP_END_CONV:
            SBC_REQUIRE(endc, "end_conv: ext (sbc_endconv) must be set");
            return launch_end_conv(op, *endc, s);
        case SBC_OP_LANGEVIN:
            SBC_REQUIRE(lang, "langevin: ext (sbc_langevin) must be set");
            return launch_langevin(op, *lang, s);
        case SBC_OP_MEASURE:
            SBC_REQUIRE(lang, "measure: ext (sbc_langevin) must be set");
            return launch_measure(op, *lang, s);
        case SBC_OP_STEP_INC: return launch_step_inc(op, s);
        default: set_error("unknown op kind %d", op.kind); return SBC_ERR_INVALID;
    }
}

static std::atomic<int> g_persistent_cus{0};           // sbc_set_persistent_cus: the process default
// sbc_plan_set_persistent_cus: while a plan with its own setting runs (or is captured) on this host thread, that setting wins --
// two plans driven from two threads, or two handles on two devices, do not see each other's width (ABI 13)
static thread_local int tls_persistent_cus = 0;
int persistent_cus(int cus) {
    static const int env = getenv("SBC_PERSIST_CUS") ? atoi(getenv("SBC_PERSIST_CUS")) : 0;      // A/B aid: overrides the setting
    const int n = env > 0 ? env : tls_persistent_cus > 0 ? tls_persistent_cus : g_persistent_cus.load(std::memory_order_relaxed);
    return n > 0 && n < cus ? n : cus;
}
// Grid of a persistent kernel whose workgroups take WHOLE samples, one workgroup per CU (conv_res, end_conv_self): the rounds a
// workgroup makes are an integer, so no more workgroups than that round count needs -- 213 samples on a width of 128 are two
// rounds either way, 107 workgroups do them and leave 21 CUs to the other stream (425 trajectories per GPU: 1.55 -> 1.47 ms per step).
// (Taking up to a quarter MORE workgroups than the plan's width where that saves a round -- 850 samples: 142 workgroups, 6 rounds
// instead of 7 -- was measured too: it costs the other stream what it gains, 4.26 -> 4.29 ms at 1700.)  (A/B aid: SBC_NO_BALANCED_GRID)
int balanced_sample_grid(int samples, int cus) {
    static const bool off = getenv("SBC_NO_BALANCED_GRID") != nullptr;
    const int w = persistent_cus(cus);
    if (off || samples <= 0) return samples < w ? (samples > 0 ? samples : 1) : w;
    const int rounds = (samples + w - 1) / w;
    return (samples + rounds - 1) / rounds;
}
// The lane streams (sbc_op.lane, ABI 14) that go with a run stream: one per (device, run stream, lane), created on first use, never
// destroyed.  A lane must not share a HARDWARE queue with its run stream: the runtime multiplexes its streams onto a few hardware queues
// (least-referenced first), and in a process that has created many streams a new one lands on the run stream's queue as often as not --
// the lane then runs strictly behind it (measured: the lanes' gain gone in the last segments of bench.py, present in a fresh process).
// Streams of another PRIORITY come from another pool of queues, but measured far worse in a process with many live streams (3.0 against
// 1.0 ms per step, high and low priority alike: profiles/r06_skip_lanes.txt).  So the library PROBES: a 200 us spin kernel on the run
// stream with an event behind it, then a trivial launch on the candidate; if the candidate's launch completes while the run stream's
// event is still pending, the two are on different hardware queues.  Up to eight fresh candidates (consecutive creations rotate over the
// queues); rejected ones are destroyed.  Costs ~0.3 ms per candidate, once per run stream.
__global__ void lane_probe_spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (ticks < 0) *sink = 1;
}
__global__ void lane_probe_touch(int* sink) { if (sink == nullptr) __builtin_trap(); }

int lane_stream_for(hipStream_t run, int lane, hipStream_t* out) {
    static std::mutex mu;
    static std::map<std::pair<std::pair<int, hipStream_t>, int>, hipStream_t> streams;
    int dev = 0;
    SBC_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    hipStream_t& found = streams[std::make_pair(std::make_pair(dev, run), lane)];
    if (!found) {
        unsigned* w = nullptr;
        { const int rc = range_flag_ptr(&w); if (rc) return rc; }
        hipEvent_t e_run = nullptr, e_cand = nullptr;
        SBC_CHECK_HIP(hipEventCreateWithFlags(&e_run, hipEventDisableTiming));
        SBC_CHECK_HIP(hipEventCreateWithFlags(&e_cand, hipEventDisableTiming));
        hipStream_t rejected[8];
        int n_rej = 0;
        for (int k = 0; k < 8 && !found; ++k) {
            hipStream_t c = nullptr;
            SBC_CHECK_HIP(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
            hipLaunchKernelGGL(lane_probe_spin, dim3(1), dim3(64), 0, run, (long long)20000, (int*)(w + 2));    // 100 MHz ticks: 200 us
            SBC_CHECK_HIP(hipEventRecord(e_run, run));
            hipLaunchKernelGGL(lane_probe_touch, dim3(1), dim3(64), 0, c, (int*)(w + 2));
            SBC_CHECK_HIP(hipEventRecord(e_cand, c));
            SBC_CHECK_HIP(hipEventSynchronize(e_cand));
            const bool concurrent = hipEventQuery(e_run) == hipErrorNotReady;
            (void)hipGetLastError();                                           // (hipErrorNotReady is not an error here)
            SBC_CHECK_HIP(hipEventSynchronize(e_run));
            if (concurrent || k == 7) found = c;                               // (the last candidate is taken whatever the probe says)
            else rejected[n_rej++] = c;
        }
        for (int k = 0; k < n_rej; ++k) (void)hipStreamDestroy(rejected[k]);
        (void)hipEventDestroy(e_run);
        (void)hipEventDestroy(e_cand);
    }
    *out = found;
    return SBC_OK;
}

struct PersistentCusScope {                              // RAII: a plan's width for the duration of its launches on this thread
    int saved;
    explicit PersistentCusScope(int n) : saved(tls_persistent_cus) { if (n > 0) tls_persistent_cus = n; }
    ~PersistentCusScope() { tls_persistent_cus = saved; }
};

}  // namespace sbc

struct sbc_plan {
    std::vector<sbc::PlanOp> ops;
    // hipGraph replay
    hipGraphExec_t exec = nullptr;
    hipStream_t graph_stream = nullptr;
    bool graph_flat = true;              // the captured graph has the lane records on the run stream (use_graph = 1)
    // side stream for ops flagged SBC_OP_SIDE (created on first use; forked from / joined into the run stream by events)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // launch lanes (sbc_op.lane / signal / wait, ABI 14): the lane streams that go with the run stream (lane_stream_for) and the plan's own events
    hipStream_t lanes[SBC_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_evt[SBC_MAX_EVENTS + 1] = {};
    hipEvent_t ev_begin = nullptr, ev_lane_end[SBC_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    bool uses_lanes = false;
    hipStream_t lane_run = nullptr;      // the run stream `lanes` were resolved for
    bool lanes_resolved = false;
    // per-tag timing
    int prof_tag = -1;
    std::vector<hipEvent_t> ev_pool;     // pairs
    size_t ev_used = 0;
    double prof_ms = 0.0;
    int64_t prof_n = 0;
    int persistent_cus = 0;              // sbc_plan_set_persistent_cus (0: the process default)
};

using namespace sbc;

static int drain_events(sbc_plan* plan) {
    for (size_t i = 0; i + 1 < plan->ev_used; i += 2) {
        SBC_CHECK_HIP(hipEventSynchronize(plan->ev_pool[i + 1]));
        float ms = 0.f;
        SBC_CHECK_HIP(hipEventElapsedTime(&ms, plan->ev_pool[i], plan->ev_pool[i + 1]));
        plan->prof_ms += ms;
        plan->prof_n += 1;
    }
    plan->ev_used = 0;
    return SBC_OK;
}

static int join_side(sbc_plan* plan, hipStream_t s) {
    SBC_CHECK_HIP(hipEventRecord(plan->ev_join, plan->side));
    SBC_CHECK_HIP(hipStreamWaitEvent(s, plan->ev_join, 0));
    return SBC_OK;
}

// the lane streams that go with run stream `s` (lane_stream_for: probed once per run stream); not inside a stream capture
static int resolve_lanes(sbc_plan* plan, hipStream_t s) {
    if (!plan->uses_lanes || (plan->lanes_resolved && plan->lane_run == s)) return SBC_OK;
    for (int l = 1; l < SBC_MAX_LANES; ++l)
        if (plan->ev_lane_end[l]) {
            const int rc = sbc::lane_stream_for(s, l, &plan->lanes[l]);
            if (rc) return rc;
        }
    plan->lane_run = s;
    plan->lanes_resolved = true;
    return SBC_OK;
}

// Lanes: every lane of the plan starts behind what `s` holds at the start of an sbc_plan_run call ...
static int fork_lanes(sbc_plan* plan, hipStream_t s) {
    if (!plan->uses_lanes) return SBC_OK;
    SBC_CHECK_HIP(hipEventRecord(plan->ev_begin, s));
    for (int l = 1; l < SBC_MAX_LANES; ++l)
        if (plan->lanes[l]) SBC_CHECK_HIP(hipStreamWaitEvent(plan->lanes[l], plan->ev_begin, 0));
    return SBC_OK;
}
// ... and `s` continues behind every lane at its end
static int join_lanes(sbc_plan* plan, hipStream_t s) {
    if (!plan->uses_lanes) return SBC_OK;
    for (int l = 1; l < SBC_MAX_LANES; ++l)
        if (plan->lanes[l]) {
            SBC_CHECK_HIP(hipEventRecord(plan->ev_lane_end[l], plan->lanes[l]));
            SBC_CHECK_HIP(hipStreamWaitEvent(s, plan->ev_lane_end[l], 0));
        }
    return SBC_OK;
}

static int run_eager(sbc_plan* plan, hipStream_t s, bool flat = false) {
    // flat: every record on `s` in list order, lanes and events ignored (a valid order: what a lane record waits for is an earlier record)
    bool side_busy = false;
    bool main_moved = true;          // the run stream has had work queued since the side stream last forked from it
    for (auto& po : plan->ops) {
        hipStream_t os = s;                              // the stream this op runs on
        if (po.op.lane > 0 && !flat) os = plan->lanes[po.op.lane];
        for (int w = 0; w < 2 && !flat; ++w)
            if (po.op.wait[w] > 0) SBC_CHECK_HIP(hipStreamWaitEvent(os, plan->lane_evt[po.op.wait[w]], 0));
        if (po.op.flags & SBC_OP_SIDE) {
            if (!plan->side) {
                SBC_CHECK_HIP(hipStreamCreateWithFlags(&plan->side, hipStreamNonBlocking));
                SBC_CHECK_HIP(hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming));
                SBC_CHECK_HIP(hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming));
            }
            if (main_moved) {
                // everything issued so far happens before the side op.  (Side ops keep their order among themselves, so a side op
                // that directly follows another one -- a weight gradient and its reduction -- needs no second fork: two runtime
                // calls less per such launch, on a path that is bound by how fast the host can issue it.)
                SBC_CHECK_HIP(hipEventRecord(plan->ev_fork, s));
                SBC_CHECK_HIP(hipStreamWaitEvent(plan->side, plan->ev_fork, 0));
                main_moved = false;
            }
            os = plan->side;
            side_busy = true;
        } else {
            main_moved = true;
            if ((po.op.flags & SBC_OP_JOIN) && side_busy) {
                const int rc = join_side(plan, s);
                if (rc) return rc;
                side_busy = false;
            }
        }
        const bool timed = plan->prof_tag >= 0 && po.op.tag == plan->prof_tag;
        if (timed) {
            if (plan->ev_used + 2 > plan->ev_pool.size()) {
                if (plan->ev_pool.size() >= 8192) {       // bounded pool: fold what is recorded so far
                    const int rc = drain_events(plan);
                    if (rc) return rc;
                } else {
                    hipEvent_t a, b;
                    SBC_CHECK_HIP(hipEventCreate(&a));
                    SBC_CHECK_HIP(hipEventCreate(&b));
                    plan->ev_pool.push_back(a);
                    plan->ev_pool.push_back(b);
                }
            }
            SBC_CHECK_HIP(hipEventRecord(plan->ev_pool[plan->ev_used], os));
        }
        const int rc = dispatch(po.op, ext_size(po.op.kind) ? &po.ext : nullptr, os);
        if (rc) return rc;
        if (timed) {
            SBC_CHECK_HIP(hipEventRecord(plan->ev_pool[plan->ev_used + 1], os));
            plan->ev_used += 2;
        }
        if (po.op.signal > 0 && !flat) SBC_CHECK_HIP(hipEventRecord(plan->lane_evt[po.op.signal], os));
    }
    if (side_busy) return join_side(plan, s);            // never return with side work the run stream does not wait for
    return SBC_OK;
}

extern "C" {

int sbc_abi_version(void) { return SBC_ABI_VERSION; }

int sbc_set_persistent_cus(int32_t n) {
    SBC_REQUIRE(n >= 0, "sbc_set_persistent_cus: negative CU count %d", n);
    sbc::g_persistent_cus.store(n, std::memory_order_relaxed);
    return SBC_OK;
}

const char* sbc_last_error(void) { return g_err; }

int sbc_device_count(void) {
    int n = 0;
    SBC_CHECK_HIP(hipGetDeviceCount(&n));
    return n;
}

int sbc_range_flag(int32_t* flag, int32_t reset) {
    SBC_REQUIRE(flag, "sbc_range_flag: flag is NULL");
    unsigned* w = nullptr;
    const int rc = range_flag_ptr(&w);
    if (rc) return rc;
    unsigned v = 0;
    // every stream of the device, non-blocking ones included (a blocking copy on the legacy stream alone does not wait for them:
    // the flag of a run still in flight on a torch side stream would surface one call late)
    SBC_CHECK_HIP(hipDeviceSynchronize());
    SBC_CHECK_HIP(hipMemcpy(&v, w, sizeof(v), hipMemcpyDeviceToHost));
    if (reset && v) SBC_CHECK_HIP(hipMemset(w, 0, sizeof(v)));
    *flag = (int32_t)v;
    return SBC_OK;
}

// ---- f16x2 activation scales (include/sbc_hip.h: sbc_f16x2_calibrate) ----------------------------------------------------
// The fixed calibration input: CN(0, 1)-like values (re, im each of variance 1/2) from a counter-based hash, the same on every
// host and for every batch, so that the scales are a function of the checkpoint and the array size only.
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
int sbc_f16x2_calibration_input(float* x, int64_t n) {
    SBC_REQUIRE(x && n >= 0, "sbc_f16x2_calibration_input: bad arguments");
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0;                                   // Irwin-Hall: four uniforms, variance 1/3
        for (uint32_t k = 1; k <= 4; ++k) sum += (double)(fmix32((uint32_t)i * 4u + k * 0x9E3779B9u) >> 8) * (1.0 / 16777216.0);
        x[i] = (float)((sum - 2.0) * 1.224744871391589);    // sqrt(3) * sqrt(1/2)
    }
    return SBC_OK;
}

int sbc_f16x2_calibrate(const sbc_op* ops, int32_t n_ops, void* stream) {
    SBC_REQUIRE(ops && n_ops > 0, "sbc_f16x2_calibrate: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    std::vector<sbc_op> run(ops, ops + n_ops);
    std::vector<int> slot_of(n_ops, -1);
    int n_slots = 0;
    for (int i = 0; i < n_ops; ++i) {
        run[i].B = 1;                                       // the first sample of every buffer
        run[i].flags &= ~(SBC_OP_SIDE | SBC_OP_JOIN);
        run[i].lane = run[i].signal = run[i].wait[0] = run[i].wait[1] = 0;        // the pass runs every record in list order on `stream`
        if ((run[i].flags & SBC_CONV_F16X2) && (run[i].kind == SBC_OP_CONV || run[i].kind == SBC_OP_CONV_PAIR || run[i].kind == SBC_OP_CONV_POOL ||
                                                  run[i].kind == SBC_OP_RES_BLOCK || run[i].kind == SBC_OP_CONV_DOWN)) {
            slot_of[i] = n_slots;
            n_slots += 2;
        } else if ((run[i].flags & SBC_CONV_F16X2) && run[i].kind == SBC_OP_CHAIN && run[i].ext) {
            slot_of[i] = n_slots;                           // three per block: the inputs of conv 1, conv 2 and the shortcut conv
            n_slots += 3 * SBC_CHAIN_MAX_BLOCKS;
        }
    }
    if (n_slots == 0) return SBC_OK;                        // nothing to calibrate in this record list
    float* dslots = nullptr;
    SBC_CHECK_HIP(hipMalloc((void**)&dslots, n_slots * sizeof(float)));
    auto fail = [&](int rc) { (void)hipFree(dslots); return rc; };
    if (hipMemsetAsync(dslots, 0, n_slots * sizeof(float), s) != hipSuccess) { set_error("sbc_f16x2_calibrate: hipMemsetAsync failed"); return fail(SBC_ERR_HIP); }
    for (int i = 0; i < n_ops; ++i) {
        if (slot_of[i] >= 0) run[i].calib = dslots + slot_of[i];
        const int rc = dispatch(run[i], run[i].ext, s);
        if (rc) return fail(rc);
    }
    if (hipStreamSynchronize(s) != hipSuccess) { set_error("sbc_f16x2_calibrate: the calibration pass failed on the device"); return fail(SBC_ERR_HIP); }
    std::vector<float> amax(n_slots);
    if (hipMemcpy(amax.data(), dslots, n_slots * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { set_error("sbc_f16x2_calibrate: copy failed"); return fail(SBC_ERR_HIP); }
    (void)hipFree(dslots);
    // act_scale = the power of two that puts the layer's calibration maximum into [2^8, 2^9): 31x head room below the overflow
    // guard, full two-term precision for everything within 2^-11 of that maximum
    auto scale_for = [](float m) {
        if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
        int e = 0;
        (void)frexpf(m, &e);                                // m = f 2^e, f in [0.5, 1)  ->  m 2^(9 - e) in [2^8, 2^9)
        int k = 9 - e;
        k = k < -24 ? -24 : k > 40 ? 40 : k;
        return ldexpf(1.f, k);
    };
    // (fourth word: the layer's inputs are small -- maximum below 2^-4: exp(x) - 1 then carries 6e-8 / 0.01 ~ 5e-6 of a typical
    // value -- so its ELU prologue must keep RELATIVE accuracy for small negative values: SBC_PRO_ELU_ACC, common.h)
    auto set_trailer = [&](const void* w, int taps, int cin, int cout, float sc, float amax_in) -> int {
        if (!w) return SBC_OK;
        float* tr = (float*)((char*)const_cast<void*>(w) + (size_t)taps * cin * cout * 2 * sizeof(uint16_t));
        float old[4];
        SBC_CHECK_HIP(hipMemcpy(old, tr, sizeof(old), hipMemcpyDeviceToHost));
        const float wd = old[2] != 0.f ? old[2] : old[1] * old[0];          // the weights' own descale 2^-s
        const float upd[4] = {sc, wd / sc, wd, (amax_in > 0.f && amax_in < 0.0625f) ? 1.f : 0.f};
        SBC_CHECK_HIP(hipMemcpy(tr, upd, sizeof(upd), hipMemcpyHostToDevice));
        return SBC_OK;
    };
    for (int i = 0; i < n_ops; ++i) {
        if (slot_of[i] < 0) continue;
        const sbc_op& o = ops[i];
        const float s1 = scale_for(amax[slot_of[i]]), s2 = scale_for(amax[slot_of[i] + 1]);
        int rc = SBC_OK;
        // (every form of a layer the record carries gets the scale, read by this record's kernel or not: a weight buffer shared
        // between array sizes must not hold two scales for one layer -- fused direct form here, unfused Winograd form there)
        if (o.kind == SBC_OP_CHAIN) {
            const sbc_chain& c = *(const sbc_chain*)o.ext;
            for (int b = 0; b < c.n_blocks && !rc; ++b) {
                const float a1 = amax[slot_of[i] + 3 * b], a2 = amax[slot_of[i] + 3 * b + 1], a3 = amax[slot_of[i] + 3 * b + 2];
                rc = set_trailer(c.w1[b], 9, o.cin, o.cout, scale_for(a1), a1);
                if (!rc) rc = set_trailer(c.w1_wino[b], 16, o.cin, o.cout, scale_for(a1), a1);
                if (!rc) rc = set_trailer(c.w2[b], 9, o.cin, o.cout, scale_for(a2), a2);
                if (!rc) rc = set_trailer(c.w2_wino[b], 16, o.cin, o.cout, scale_for(a2), a2);
                if (!rc && c.type[b] == SBC_CHAIN_RES) rc = set_trailer(c.w3[b], 9, o.cin, o.cout, scale_for(a3), a3);
            }
        } else if (o.kind == SBC_OP_CONV_DOWN) {
            rc = set_trailer(o.weight_split, 16, o.cin, o.cout, s1, amax[slot_of[i]]);              // the pooled 3x3 filter (4x4 taps)
            if (!rc) rc = set_trailer(o.weight2_split, 4, o.cin, o.cout, s2, amax[slot_of[i] + 1]);   // the pooled 1x1 shortcut (2x2 taps)
            // ... and the layers' UNPOOLED forms, which the unfused launches of another array size read (include/sbc_hip.h: SBC_OP_CONV_DOWN)
            if (!rc) rc = set_trailer(o.weight, 9, o.cin, o.cout, s1, amax[slot_of[i]]);
            if (!rc) rc = set_trailer(o.weight_wino_split, 16, o.cin, o.cout, s1, amax[slot_of[i]]);
            if (!rc) rc = set_trailer(o.weight_wino, 1, o.cin, o.cout, s2, amax[slot_of[i] + 1]);
        } else if (o.kind == SBC_OP_CONV_PAIR || o.kind == SBC_OP_RES_BLOCK) {
            rc = set_trailer(o.weight_split, 9, o.cin, o.cout, s1, amax[slot_of[i]]);
            if (!rc) rc = set_trailer(o.weight_wino_split, 16, o.cin, o.cout, s1, amax[slot_of[i]]);
            if (!rc) rc = set_trailer(o.weight2_split, 9, o.cout, o.cout, s2, amax[slot_of[i] + 1]);
            if (!rc) rc = set_trailer(o.weight2_wino_split, 16, o.cout, o.cout, s2, amax[slot_of[i] + 1]);
        } else if (o.kind == SBC_OP_CONV_POOL) {
            rc = set_trailer(o.weight_split, 9, o.cin, o.cout, s1, amax[slot_of[i]]);
            if (!rc) rc = set_trailer(o.weight_wino_split, 16, o.cin, o.cout, s1, amax[slot_of[i]]);
        } else {
            rc = set_trailer(o.weight_split, o.ksize * o.ksize, o.cin, o.cout, s1, amax[slot_of[i]]);
            if (!rc && o.ksize == 3 && o.dil == 1) rc = set_trailer(o.weight_wino_split, 16, o.cin, o.cout, s1, amax[slot_of[i]]);
        }
        if (rc) return rc;
    }
    // whatever the pass itself flagged (it ran with the scales it is about to replace) is not a verdict on a real run
    unsigned* w = nullptr;
    { const int rc = range_flag_ptr(&w); if (rc) return rc; }
    SBC_CHECK_HIP(hipMemset(w, 0, sizeof(unsigned)));
    return SBC_OK;
}

int sbc_op_launch(const sbc_op* op, void* stream) {
    SBC_REQUIRE(op, "sbc_op_launch: op is NULL");
    return dispatch(*op, op->ext, (hipStream_t)stream);
}

int sbc_plan_create(const sbc_op* ops, int32_t n_ops, sbc_plan** out_plan) {
    SBC_REQUIRE(ops && n_ops > 0 && out_plan, "sbc_plan_create: bad arguments");
    sbc_plan* plan = new sbc_plan();
    plan->ops.resize(n_ops);
    for (int i = 0; i < n_ops; ++i) {
        PlanOp& po = plan->ops[i];
        po.op = ops[i];
        memset(&po.ext, 0, sizeof(po.ext));
        if (const size_t es = ext_size(ops[i].kind)) {
            if (!ops[i].ext) { delete plan; set_error("op %d (kind %d): ext is NULL", i, ops[i].kind); return SBC_ERR_INVALID; }
            memcpy(&po.ext, ops[i].ext, es);
        }
        po.op.ext = nullptr;
        // resolve kernel variants / set function attributes now, so a later hipGraph capture sees launches only
        int rc = SBC_OK;
        if (po.op.kind == SBC_OP_CONV) rc = launch_conv(po.op, nullptr, true);
        else if (po.op.kind == SBC_OP_CONV_PAIR) rc = launch_conv_pair(po.op, nullptr, true);
        else if (po.op.kind == SBC_OP_CONV_POOL) rc = launch_conv_pool(po.op, nullptr, true);
        else if (po.op.kind == SBC_OP_RES_BLOCK) rc = launch_res_block(po.op, nullptr, true);
        else if (po.op.kind == SBC_OP_CHAIN) rc = launch_chain(po.op, po.ext.chain, nullptr, true);
        else if (po.op.kind == SBC_OP_CONV_DOWN) rc = launch_conv_down(po.op, nullptr, true);
        else if (po.op.kind == SBC_OP_END_CONV) rc = launch_end_conv(po.op, po.ext.endc, nullptr, true);
        else if (po.op.kind == SBC_OP_LANGEVIN) rc = launch_langevin(po.op, po.ext.lang, nullptr, true);
        if (rc) { delete plan; return rc; }
    }
    // launch lanes (ABI 14): validate the records' lane / event fields, create the streams and events they name
    {
        bool signalled[SBC_MAX_EVENTS + 1] = {};
        auto bad = [&](const char* what, int i, int v) { set_error("sbc_plan_create: op %d: %s %d", i, what, v); sbc_plan_destroy(plan); return SBC_ERR_INVALID; };
        for (int i = 0; i < n_ops; ++i) {
            const sbc_op& o = plan->ops[i].op;
            if (o.lane < 0 || o.lane >= SBC_MAX_LANES) return bad("lane out of range:", i, o.lane);
            if (o.signal < 0 || o.signal > SBC_MAX_EVENTS) return bad("signal id out of range:", i, o.signal);
            if ((o.flags & (SBC_OP_SIDE | SBC_OP_JOIN)) && (o.lane || o.signal || o.wait[0] || o.wait[1])) return bad("SBC_OP_SIDE / SBC_OP_JOIN and lanes do not mix, lane", i, o.lane);
            for (int w = 0; w < 2; ++w) {
                if (o.wait[w] < 0 || o.wait[w] > SBC_MAX_EVENTS) return bad("wait id out of range:", i, o.wait[w]);
                if (o.wait[w] > 0 && !signalled[o.wait[w]]) return bad("waits for an event no earlier record signals:", i, o.wait[w]);
            }
            if (o.signal > 0) signalled[o.signal] = true;
            if (o.lane > 0 || o.signal > 0) plan->uses_lanes = true;
        }
        if (plan->uses_lanes) {
            auto hip_fail = [&](hipError_t e) { set_error("sbc_plan_create: lanes: %s", hipGetErrorString(e)); sbc_plan_destroy(plan); return SBC_ERR_HIP; };
            // (no timing, no system-scope fence: the events order launches of ONE device; its kernels' own agent-scope release / acquire
            // make their results visible to each other.  Measured: the same step time with plain hipEventDisableTiming events)
            const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
            hipError_t e = hipEventCreateWithFlags(&plan->ev_begin, evf);
            if (e != hipSuccess) return hip_fail(e);
            for (int i = 0; i < n_ops; ++i) {
                const sbc_op& o = plan->ops[i].op;
                if (o.lane > 0 && !plan->ev_lane_end[o.lane]) {          // (the lane's stream itself: resolve_lanes, at the first run)
                    if ((e = hipEventCreateWithFlags(&plan->ev_lane_end[o.lane], evf)) != hipSuccess) return hip_fail(e);
                }
                if (o.signal > 0 && !plan->lane_evt[o.signal] && (e = hipEventCreateWithFlags(&plan->lane_evt[o.signal], evf)) != hipSuccess) return hip_fail(e);
            }
        }
    }
    *out_plan = plan;
    return SBC_OK;
}

int sbc_plan_run(sbc_plan* plan, void* stream, int32_t n_iters, int32_t use_graph) {
    SBC_REQUIRE(plan && n_iters >= 0, "sbc_plan_run: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    PersistentCusScope width(plan->persistent_cus);
    if (n_iters > 0) { const int rc = resolve_lanes(plan, s); if (rc) return rc; }
    if (!use_graph || plan->prof_tag >= 0) {
        if (n_iters == 0) return SBC_OK;
        int rc = fork_lanes(plan, s);
        for (int it = 0; it < n_iters && !rc; ++it) rc = run_eager(plan, s);
        const int rj = join_lanes(plan, s);              // also after a failed launch: never return with lane work `s` does not wait for
        return rc ? rc : rj;
    }
    SBC_REQUIRE(s != nullptr, "sbc_plan_run: graph replay needs a non-default stream");
    // use_graph = 1: the captured graph is FLAT -- lane records on the run stream in list order.  A graph with parallel branches replays
    // correctly and side by side (ald.AldPair, use_graph = 2), but hipGraphLaunch of such a graph crashed inside the runtime
    // (hip::Graph::UpdateStreams, ROCm 7.0.2) in a process that had created many streams -- reproducibly, depending on what ran before
    // (tests/test_gpu_parity.py in one particular selection order) -- and graph replay is slower than eager launches anyway.
    const bool flat = use_graph != 2;
    if (!plan->exec || plan->graph_stream != s || plan->graph_flat != flat) {
        if (plan->exec) { (void)hipGraphExecDestroy(plan->exec); plan->exec = nullptr; }
        hipGraph_t graph = nullptr;
        SBC_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        int rc = flat ? SBC_OK : fork_lanes(plan, s);    // (lanes join the capture through the events they wait for)
        if (!rc) rc = run_eager(plan, s, flat);
        if (!flat) { const int rj = join_lanes(plan, s); if (!rc) rc = rj; }
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        SBC_CHECK_HIP(e);
        SBC_CHECK_HIP(hipGraphInstantiate(&plan->exec, graph, nullptr, nullptr, 0));
        SBC_CHECK_HIP(hipGraphDestroy(graph));
        plan->graph_stream = s;
        plan->graph_flat = flat;
    }
    for (int it = 0; it < n_iters; ++it) SBC_CHECK_HIP(hipGraphLaunch(plan->exec, s));
    return SBC_OK;
}

int sbc_plan_set_persistent_cus(sbc_plan* plan, int32_t n) {
    SBC_REQUIRE(plan && n >= 0, "sbc_plan_set_persistent_cus: bad arguments");
    if (n != plan->persistent_cus && plan->exec) {       // a captured graph has the old grids baked in
        (void)hipGraphExecDestroy(plan->exec);
        plan->exec = nullptr;
    }
    plan->persistent_cus = n;
    return SBC_OK;
}

void sbc_plan_destroy(sbc_plan* plan) {
    if (!plan) return;
    if (plan->exec) (void)hipGraphExecDestroy(plan->exec);
    for (hipEvent_t e : plan->ev_pool) (void)hipEventDestroy(e);
    if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
    if (plan->ev_join) (void)hipEventDestroy(plan->ev_join);
    if (plan->side) (void)hipStreamDestroy(plan->side);
    if (plan->ev_begin) (void)hipEventDestroy(plan->ev_begin);
    for (int l = 1; l < SBC_MAX_LANES; ++l)                   // (the lane streams belong to the run stream, not to the plan: lane_stream_for)
        if (plan->ev_lane_end[l]) (void)hipEventDestroy(plan->ev_lane_end[l]);
    for (hipEvent_t e : plan->lane_evt) if (e) (void)hipEventDestroy(e);
    delete plan;
}

int sbc_plan_profile(sbc_plan* plan, int32_t tag) {
    SBC_REQUIRE(plan, "sbc_plan_profile: plan is NULL");
    plan->prof_tag = tag;
    plan->ev_used = 0;
    plan->prof_ms = 0.0;
    plan->prof_n = 0;
    return SBC_OK;
}

int sbc_plan_profile_read(sbc_plan* plan, double* total_ms, int64_t* n_launches) {
    SBC_REQUIRE(plan && total_ms && n_launches, "sbc_plan_profile_read: bad arguments");
    const int rc = drain_events(plan);
    if (rc) return rc;
    *total_ms = plan->prof_ms;
    *n_launches = plan->prof_n;
    plan->prof_ms = 0.0;
    plan->prof_n = 0;
    return SBC_OK;
}

int sbc_pack_conv_weight(const float* src, int32_t cout, int32_t cin, int32_t ksize, float* dst) {
    SBC_REQUIRE(src && dst, "sbc_pack_conv_weight: NULL pointer");
    SBC_REQUIRE(cin % 8 == 0 && cout % 32 == 0 && (ksize == 1 || ksize == 3),
                "sbc_pack_conv_weight: cin %% 8, cout %% 32, ksize in {1,3} required (got %d, %d, %d)", cin, cout, ksize);
    const int taps = ksize * ksize, KG = cin / 8, NB = cout / 32;
    for (int tap = 0; tap < taps; ++tap)
        for (int g = 0; g < KG; ++g)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int co = nb * 32 + (lane & 31), ci = g * 8 + 4 * (lane >> 5) + j;
                        dst[((((size_t)tap * KG + g) * NB + nb) * 64 + lane) * 4 + j] =
                            src[((size_t)co * cin + ci) * taps + tap];
                    }
    return SBC_OK;
}

int sbc_pack_conv_weight_winograd(const float* src, int32_t cout, int32_t cin, float* dst) {
    SBC_REQUIRE(src && dst, "sbc_pack_conv_weight_winograd: NULL pointer");
    SBC_REQUIRE(cin % 8 == 0 && cout % 32 == 0, "sbc_pack_conv_weight_winograd: cin %% 8, cout %% 32 required (got %d, %d)",
                cin, cout);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> u((size_t)cout * cin * 16);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = src + ((size_t)co * cin + ci) * 9;
            for (int i = 0; i < 4; ++i)
                for (int l = 0; l < 4; ++l) {
                    double s = 0;
                    for (int j = 0; j < 3; ++j)
                        for (int k = 0; k < 3; ++k) s += G[i][j] * (double)g[j * 3 + k] * G[l][k];
                    u[((size_t)co * cin + ci) * 16 + i * 4 + l] = (float)s;
                }
        }
    const int KG = cin / 8, NB = cout / 32;
    for (int tap = 0; tap < 16; ++tap)
        for (int g = 0; g < KG; ++g)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int co = nb * 32 + (lane & 31), ci = g * 8 + 4 * (lane >> 5) + j;
                        dst[((((size_t)tap * KG + g) * NB + nb) * 64 + lane) * 4 + j] = u[((size_t)co * cin + ci) * 16 + tap];
                    }
    return SBC_OK;
}

// round-to-nearest-even fp32 -> bf16 (bit pattern), as v_cvt_pk_bf16_f32 does for finite values
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int sbc_pack_conv_weight_split(const float* src, int32_t cout, int32_t cin, int32_t ksize, uint16_t* dst) {
    SBC_REQUIRE(src && dst, "sbc_pack_conv_weight_split: NULL pointer");
    SBC_REQUIRE(cin % 16 == 0 && cout % 32 == 0 && (ksize == 1 || ksize == 3),
                "sbc_pack_conv_weight_split: cin %% 16, cout %% 32, ksize in {1,3} required (got %d, %d, %d)", cin, cout,
                ksize);
    const int taps = ksize * ksize, KG = cin / 16, NB = cout / 32;
    for (int tap = 0; tap < taps; ++tap)
        for (int g = 0; g < KG; ++g)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = nb * 32 + (lane & 31), ci = g * 16 + 8 * (lane >> 5) + j;
                        const float w = src[((size_t)co * cin + ci) * taps + tap];
                        const uint16_t h = bf16_rne(w);
                        const float r1 = w - bf16_to_f32(h);
                        const uint16_t m = bf16_rne(r1);
                        const uint16_t l = bf16_rne(r1 - bf16_to_f32(m));
                        const size_t base = (((size_t)tap * KG + g) * NB + nb) * 3;
                        dst[((base + 0) * 64 + lane) * 8 + j] = h;
                        dst[((base + 1) * 64 + lane) * 8 + j] = m;
                        dst[((base + 2) * 64 + lane) * 8 + j] = l;
                    }
    return SBC_OK;
}

int sbc_pack_conv_weight_winograd_split(const float* src, int32_t cout, int32_t cin, uint16_t* dst) {
    SBC_REQUIRE(src && dst, "sbc_pack_conv_weight_winograd_split: NULL pointer");
    SBC_REQUIRE(cin % 16 == 0 && cout % 32 == 0, "sbc_pack_conv_weight_winograd_split: cin %% 16, cout %% 32 required (got %d, %d)",
                cin, cout);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> u((size_t)cout * cin * 16);             // torch-like [cout][cin][4][4]
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = src + ((size_t)co * cin + ci) * 9;
            for (int i = 0; i < 4; ++i)
                for (int l = 0; l < 4; ++l) {
                    double s = 0;
                    for (int j = 0; j < 3; ++j)
                        for (int k = 0; k < 3; ++k) s += G[i][j] * (double)g[j * 3 + k] * G[l][k];
                    u[((size_t)co * cin + ci) * 16 + i * 4 + l] = (float)s;
                }
        }
    const int KG = cin / 16, NB = cout / 32;
    for (int tap = 0; tap < 16; ++tap)
        for (int g = 0; g < KG; ++g)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = nb * 32 + (lane & 31), ci = g * 16 + 8 * (lane >> 5) + j;
                        const float w = u[((size_t)co * cin + ci) * 16 + tap];
                        const uint16_t h = bf16_rne(w);
                        const float r1 = w - bf16_to_f32(h);
                        const uint16_t m = bf16_rne(r1);
                        const uint16_t l = bf16_rne(r1 - bf16_to_f32(m));
                        const size_t base = (((size_t)tap * KG + g) * NB + nb) * 3;
                        dst[((base + 0) * 64 + lane) * 8 + j] = h;
                        dst[((base + 1) * 64 + lane) * 8 + j] = m;
                        dst[((base + 2) * 64 + lane) * 8 + j] = l;
                    }
    return SBC_OK;
}

// round-to-nearest-even fp32 -> fp16 bit pattern (the host compiler's _Float16 conversion is IEEE RNE, like v_cvt_f16_f32)
static inline uint16_t f16_rne(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

int sbc_pack_conv_weight_f16(const float* src, int32_t cout, int32_t cin, int32_t ksize, uint16_t* dst) {
    SBC_REQUIRE(src && dst, "sbc_pack_conv_weight_f16: NULL pointer");
    SBC_REQUIRE(cin % 16 == 0 && cout % 32 == 0 && (ksize == 1 || ksize == 3),
                "sbc_pack_conv_weight_f16: cin %% 16, cout %% 32, ksize in {1,3} required (got %d, %d, %d)", cin, cout, ksize);
    const int taps = ksize * ksize, KG = cin / 16, NB = cout / 32;
    for (int tap = 0; tap < taps; ++tap)
        for (int g = 0; g < KG; ++g)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = nb * 32 + (lane & 31), ci = g * 16 + 8 * (lane >> 5) + j;
                        dst[((((size_t)tap * KG + g) * NB + nb) * 64 + lane) * 8 + j] =
                            f16_rne(src[((size_t)co * cin + ci) * taps + tap]);
                    }
    return SBC_OK;
}

int sbc_pack_conv_weight_winograd_f16(const float* src, int32_t cout, int32_t cin, uint16_t* dst) {
    SBC_REQUIRE(src && dst, "sbc_pack_conv_weight_winograd_f16: NULL pointer");
    SBC_REQUIRE(cin % 16 == 0 && cout % 32 == 0, "sbc_pack_conv_weight_winograd_f16: cin %% 16, cout %% 32 required (got %d, %d)",
                cin, cout);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int KG = cin / 16, NB = cout / 32;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = src + ((size_t)co * cin + ci) * 9;
            const int nb = co / 32, kg = ci / 16, lane = (co & 31) + 32 * ((ci & 15) >> 3), j = ci & 7;
            for (int i = 0; i < 4; ++i)
                for (int l = 0; l < 4; ++l) {
                    double u = 0;
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) u += G[i][a] * (double)g[a * 3 + b] * G[l][b];
                    dst[((((size_t)(i * 4 + l) * KG + kg) * NB + nb) * 64 + lane) * 8 + j] = f16_rne((float)u);
                }
        }
    return SBC_OK;
}

// ---- f16x2 forms (SBC_CONV_F16X2): two fp16 terms of w * 2^s + the scale trailer --------------------------------------
// s: the power of two that puts the largest |w| of the layer into [2^13, 2^14) (fp16 keeps 11 significant bits down to
// 2^-14, so every weight above 2^-27 of the largest one keeps its 22 bits); activations are scaled by 2^SBC_F16X2_ACT_SHIFT.
static int f16x2_shift(const float* w, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) { const float a = fabsf(w[i]); if (a > m) m = a; }
    if (!(m > 0.f) || !std::isfinite(m)) return 0;
    int e = 0;
    (void)frexpf(m, &e);                       // m = f * 2^e, f in [0.5, 1)
    int s = 14 - e;
    return s < -100 ? -100 : s > 100 ? 100 : s;
}
static inline void f16x2_terms(float w, int s, uint16_t* h, uint16_t* l) {
    const float a = ldexpf(w, s);
    const _Float16 hh = (_Float16)a;
    const _Float16 ll = (_Float16)(a - (float)hh);
    memcpy(h, &hh, 2);
    memcpy(l, &ll, 2);
}
static void f16x2_trailer_write(uint16_t* dst, size_t n16, int s) {
    // (act_scale, descale = 1 / (act_scale weight_scale), the weights' own descale 2^-s, 0): sbc_f16x2_calibrate rewrites the first two
    const float tr[4] = {ldexpf(1.f, SBC_F16X2_ACT_SHIFT), ldexpf(1.f, -(s + SBC_F16X2_ACT_SHIFT)), ldexpf(1.f, -s), 0.f};
    memcpy(dst + n16, tr, sizeof(tr));
}

// [cout][cin][taps] float32 -> the two-term fp16 fragment order (+ trailer); shared by the plain and the pooled packer
static void pack_f16x2_taps(const float* src, int cout, int cin, int taps, uint16_t* dst) {
    const int KG = cin / 16, NB = cout / 32;
    const int s = f16x2_shift(src, (size_t)cout * cin * taps);
    for (int tap = 0; tap < taps; ++tap)
        for (int g = 0; g < KG; ++g)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = nb * 32 + (lane & 31), ci = g * 16 + 8 * (lane >> 5) + j;
                        const size_t base = (((size_t)tap * KG + g) * NB + nb) * 2;
                        f16x2_terms(src[((size_t)co * cin + ci) * taps + tap], s, &dst[((base + 0) * 64 + lane) * 8 + j],
                                    &dst[((base + 1) * 64 + lane) * 8 + j]);
                    }
    f16x2_trailer_write(dst, (size_t)taps * KG * NB * 2 * 512, s);
}

int sbc_pack_conv_weight_f16x2(const float* src, int32_t cout, int32_t cin, int32_t ksize, uint16_t* dst) {
    SBC_REQUIRE(src && dst, "sbc_pack_conv_weight_f16x2: NULL pointer");
    SBC_REQUIRE(cin % 16 == 0 && cout % 32 == 0 && (ksize == 1 || ksize == 3),
                "sbc_pack_conv_weight_f16x2: cin %% 16, cout %% 32, ksize in {1,3} required (got %d, %d, %d)", cin, cout, ksize);
    pack_f16x2_taps(src, cout, cin, ksize * ksize, dst);
    return SBC_OK;
}

int sbc_pack_conv_weight_pooled_f16x2(const float* src, int32_t cout, int32_t cin, int32_t ksize, uint16_t* dst) {
    SBC_REQUIRE(src && dst, "sbc_pack_conv_weight_pooled_f16x2: NULL pointer");
    SBC_REQUIRE(cin % 16 == 0 && cout % 32 == 0 && (ksize == 1 || ksize == 3),
                "sbc_pack_conv_weight_pooled_f16x2: cin %% 16, cout %% 32, ksize in {1,3} required (got %d, %d, %d)", cin, cout, ksize);
    // meanpool2(conv_k(x)) = conv_{k+1, stride 2}(x) with W'[p][q] = 1/4 sum_{a,b in {0,1}} W[p - a][q - b]   (layers.py:309-313)
    const int k = ksize, k1 = ksize + 1;
    std::vector<float> wp((size_t)cout * cin * k1 * k1);
    for (size_t oc = 0; oc < (size_t)cout * cin; ++oc)
        for (int p = 0; p < k1; ++p)
            for (int q = 0; q < k1; ++q) {
                double v = 0;
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b)
                        if (p - a >= 0 && p - a < k && q - b >= 0 && q - b < k) v += (double)src[oc * k * k + (p - a) * k + (q - b)];
                wp[oc * k1 * k1 + p * k1 + q] = (float)(0.25 * v);
            }
    pack_f16x2_taps(wp.data(), cout, cin, k1 * k1, dst);
    return SBC_OK;
}

int sbc_pack_conv_weight_winograd_f16x2(const float* src, int32_t cout, int32_t cin, uint16_t* dst) {
    SBC_REQUIRE(src && dst, "sbc_pack_conv_weight_winograd_f16x2: NULL pointer");
    SBC_REQUIRE(cin % 16 == 0 && cout % 32 == 0, "sbc_pack_conv_weight_winograd_f16x2: cin %% 16, cout %% 32 required (got %d, %d)",
                cin, cout);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> u((size_t)cout * cin * 16);             // torch-like [cout][cin][4][4]
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = src + ((size_t)co * cin + ci) * 9;
            for (int i = 0; i < 4; ++i)
                for (int l = 0; l < 4; ++l) {
                    double v = 0;
                    for (int j = 0; j < 3; ++j)
                        for (int k = 0; k < 3; ++k) v += G[i][j] * (double)g[j * 3 + k] * G[l][k];
                    u[((size_t)co * cin + ci) * 16 + i * 4 + l] = (float)v;
                }
        }
    const int KG = cin / 16, NB = cout / 32;
    const int s = f16x2_shift(u.data(), u.size());
    for (int tap = 0; tap < 16; ++tap)
        for (int g = 0; g < KG; ++g)
            for (int nb = 0; nb < NB; ++nb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = nb * 32 + (lane & 31), ci = g * 16 + 8 * (lane >> 5) + j;
                        const size_t base = (((size_t)tap * KG + g) * NB + nb) * 2;
                        f16x2_terms(u[((size_t)co * cin + ci) * 16 + tap], s, &dst[((base + 0) * 64 + lane) * 8 + j],
                                    &dst[((base + 1) * 64 + lane) * 8 + j]);
                    }
    f16x2_trailer_write(dst, (size_t)16 * KG * NB * 2 * 512, s);
    return SBC_OK;
}

}  // extern "C"
