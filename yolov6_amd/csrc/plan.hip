// plan.hip — native executor: an ordered list of hot-path ops with fixed device pointers,
// replayed once per forward (eagerly or from a captured hipGraph), with per-conv kernel
// autotuning and a per-op hipEvent profile.  This is what sits behind Model.forward
// (reference yolov6/models/yolo.py:33-41) instead of ~200 aten dispatches.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include "plan_internal.hpp"

static thread_local char g_err[512] = "";

void y6_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* y6_last_error(void) { return g_err; }
extern "C" int y6_abi_version(void) { return Y6_ABI_VERSION; }

// sizeof of a public struct of include/yolov6_hip.h by name (0: unknown).  A binding in another language mirrors these structs
// by hand (yolov6_amd/_lib.py does, with ctypes): it checks its mirrors against this table when it loads the library, so a field
// added on one side only is a load-time error instead of a kernel reading past a descriptor.
extern "C" size_t y6_abi_sizeof(const char* name) {
    static const struct { const char* n; size_t s; } kTab[] = {
        {"y6_tensor", sizeof(y6_tensor)},
        {"y6_conv_desc", sizeof(y6_conv_desc)},
        {"y6_conv_geometry", sizeof(y6_conv_geometry)},
        {"y6_conv_i8_desc", sizeof(y6_conv_i8_desc)},
        {"y6_convt_desc", sizeof(y6_convt_desc)},
        {"y6_stem_desc", sizeof(y6_stem_desc)},
        {"y6_pw_s2_desc", sizeof(y6_pw_s2_desc)},
        {"y6_stem_s2_desc", sizeof(y6_stem_s2_desc)},
        {"y6_letterbox_desc", sizeof(y6_letterbox_desc)},
        {"y6_decode_desc", sizeof(y6_decode_desc)},
        {"y6_pred_decode_desc", sizeof(y6_pred_decode_desc)},
        {"y6_nms_sink", sizeof(y6_nms_sink)},
        {"y6_nms_desc", sizeof(y6_nms_desc)},
        {"y6_tal_desc", sizeof(y6_tal_desc)},
        {"y6_atss_desc", sizeof(y6_atss_desc)},
        {"y6_loss_desc", sizeof(y6_loss_desc)},
        {"y6_distill_desc", sizeof(y6_distill_desc)},
        {"y6_bn_train_desc", sizeof(y6_bn_train_desc)},
        {"y6_bn_train_multi_desc", sizeof(y6_bn_train_multi_desc)},
        {"y6_bnact_desc", sizeof(y6_bnact_desc)},
        {"y6_bnact_bwd_desc", sizeof(y6_bnact_bwd_desc)},
        {"y6_wgrad_t_desc", sizeof(y6_wgrad_t_desc)},
        {"y6_wgrad_desc", sizeof(y6_wgrad_desc)},
        {"y6_wgrad_nhwc_desc", sizeof(y6_wgrad_nhwc_desc)},
        {"y6_wgrad_stem_desc", sizeof(y6_wgrad_stem_desc)},
        {"y6_wgrad_flat_geom", sizeof(y6_wgrad_flat_geom)},
        {"y6_pack_job", sizeof(y6_pack_job)},
        {"y6_pack_batch_desc", sizeof(y6_pack_batch_desc)},
        {"y6_sppf_bwd_desc", sizeof(y6_sppf_bwd_desc)},
        {"y6_dgrad_s2_desc", sizeof(y6_dgrad_s2_desc)},
        {"y6_sppf_q_desc", sizeof(y6_sppf_q_desc)},
        {"y6_head_pack_desc", sizeof(y6_head_pack_desc)},
        {"y6_head_ab_desc", sizeof(y6_head_ab_desc)},
        {"y6_loss_grad_desc", sizeof(y6_loss_grad_desc)}};
    if (!name) return 0;
    for (const auto& e : kTab)
        if (strcmp(e.n, name) == 0) return e.s;
    return 0;
}

extern "C" int y6_device_info(int* n_cu, char* arch, size_t arch_len) {
    int dev = 0;
    Y6_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    Y6_HIP(hipGetDeviceProperties(&prop, dev));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (arch && arch_len) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return Y6_OK;
}

namespace {
struct Op {
    int kind;
    y6_conv_desc conv;
    y6_convt_desc convt;
    y6_stem_desc stem;
    y6_decode_desc dec;
    y6_tensor t[4];
    const void* src;
    void* dst;
    int dtype;
    // generic op (training path, train.hip / wgrad.hip / loss.hip): launcher + descriptor blob
    y6_generic_fn gfn;
    int gtag;
    double gflops, gbytes;
    int gout_off;   // >= 0: byte offset inside `blob` of the pointer to a caller-visible output tensor (rebindable)
    int gin_off;    // >= 0: ... of the pointer to the caller's boundary input (the NCHW image of a fused stem op)
    alignas(16) unsigned char blob[Y6_GENERIC_BLOB];
};
}  // namespace

struct y6_plan {
    std::vector<Op> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipEvent_t> events;  // timing slots: (ops+1) events per slot
    int slots = 0;
    int slots_used = 0;
    // side stream (y6_plan_mark_side): ops nobody on the main stream waits for before the end of a run
    std::vector<char> side;          // per op (shorter than ops: the rest are main-stream ops)
    int side_pending = 0;            // side-stream ops enqueued since the side stream last joined the caller's stream
    hipStream_t side_stream = nullptr;
    std::vector<hipEvent_t> sync_ev; // ring of fork / join events
    size_t sync_pos = 0;
    // two-stream schedule of whole-plan eager runs (y6_plan_set_schedule): enqueue order, stream per op, event waits
    std::vector<int32_t> sched_order, sched_stream;
    std::vector<std::vector<int32_t>> sched_waits;   // per op: ops of the OTHER stream it waits for
    std::vector<char> sched_records;                 // per op: somebody waits for it -> an event is recorded behind it
    std::vector<hipEvent_t> sched_ev;                // per op: that event, in the run being enqueued
};

static int run_op(const Op& op, hipStream_t s) {
    switch (op.kind) {
        case Y6_OP_CONV: return y6_conv2d(&op.conv, s);
        case Y6_OP_CONVT: return y6_convt2x2(&op.convt, s);
        case Y6_OP_STEM: return y6_stem_conv(&op.stem, s);
        case Y6_OP_SPPF: return y6_sppf_pool(&op.t[0], &op.t[1], &op.t[2], &op.t[3], s);
        case Y6_OP_DECODE: return y6_head_decode(&op.dec, s);
        case Y6_OP_NCHW2NHWC: return y6_nchw_to_nhwc(op.src, op.dtype, &op.t[0], s);
        case Y6_OP_NHWC2NCHW: return y6_nhwc_to_nchw(&op.t[0], op.dst, op.dtype, s);
        case Y6_OP_GENERIC: return op.gfn(op.blob, s);
    }
    y6_set_error("plan: unknown op kind %d", op.kind);
    return Y6_EINVAL;
}

// algorithmic FLOPs / HBM bytes of one op (input once + output once + weights once)
static void op_cost(const Op& op, double* pf, double* pby) {
    double f = 0.0, by = 0.0;
        if (op.kind == Y6_OP_CONV) {
        f = y6_conv_flops(&op.conv);
        by = y6_conv_bytes(&op.conv);
    } else if (op.kind == Y6_OP_CONVT) {
        const y6_tensor &a = op.convt.in, &o = op.convt.out;
        f = 2.0 * o.B * o.H * o.W * (double)o.C * a.C;
        by = 2.0 * ((double)a.B * a.H * a.W * a.C + (double)o.B * o.H * o.W * o.C + 4.0 * a.C * o.C);
    } else if (op.kind == Y6_OP_STEM) {
        const y6_tensor& o = op.stem.out;
        f = 2.0 * o.B * o.H * o.W * (double)o.C * op.stem.Cin * 9;
        by = (op.stem.in_dtype == Y6_F16 ? 2.0 : 4.0) * op.stem.B * op.stem.Cin * (double)op.stem.H * op.stem.W +
             2.0 * o.B * o.H * o.W * o.C;
    } else if (op.kind == Y6_OP_SPPF) {
        by = 2.0 * 4.0 * op.t[0].B * op.t[0].H * op.t[0].W * op.t[0].C;
    } else if (op.kind == Y6_OP_GENERIC) {
        f = op.gflops;
        by = op.gbytes;
    } else if (op.kind == Y6_OP_DECODE) {
        double A = 0;
        for (int l = 0; l < op.dec.n_levels; ++l) A += (double)op.dec.cls[l].H * op.dec.cls[l].W;
        const double B = op.dec.cls[0].B;
        by = B * A * ((op.dec.nc + 5) * 4.0 + (op.dec.nc + op.dec.reg[0].C) * 2.0);
    }
    *pf = f;
    *pby = by;
}

static void drop_graph(y6_plan* p) {
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    p->exec = nullptr;
    p->graph = nullptr;
}

extern "C" y6_plan* y6_plan_create(void) { return new y6_plan(); }
static void drop_events(y6_plan* p) {
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    p->events.clear();
    p->slots = p->slots_used = 0;
}
extern "C" void y6_plan_destroy(y6_plan* p) {
    if (!p) return;
    drop_graph(p);
    drop_events(p);
    for (hipEvent_t e : p->sync_ev) (void)hipEventDestroy(e);
    if (p->side_stream) (void)hipStreamDestroy(p->side_stream);
    delete p;
}

// The op added last runs on the plan's SIDE stream in eager runs (y6_plan_run / y6_plan_run_range; timed runs and captured
// graphs stay on one stream).  Contract, kept by the caller (train_engine.py marks weight-gradient work: operand transposes,
// weight-gradient GEMMs, bias sums): a side op may read anything ops BEFORE it in plan order wrote; nothing it reads is
// written by a later op; nothing it writes is read or written by a main-stream op of the same run.  The executor forks the
// side stream off the main stream in front of a side op whenever main-stream ops were enqueued since the last fork, and
// joins it back at the end of the run / range - so for every caller the main stream still orders everything.
// Why: the backward pass of a conv net is one dependent chain (data gradients, BatchNorm backward) plus weight-gradient work
// that hangs off it and feeds only the optimizer; on one stream the two alternate and the small weight-gradient launches
// (1x1 convs, narrow maps: 45-130 TFLOP/s, 20 us operand transposes) run alone on a 256-CU chip.
extern "C" int y6_plan_mark_side(y6_plan* p) {
    Y6_REQUIRE(p && !p->ops.empty(), "plan_mark_side: no op to mark");
    p->side.resize(p->ops.size(), 0);
    p->side.back() = 1;
    return Y6_OK;
}

static bool side_stream_enabled() {
    static const bool v = [] {
        const char* e = getenv("Y6_SIDE_STREAM");
        return e ? atoi(e) != 0 : true;   // on since r03t: 45.5 -> 41.8 ms per training step, same box (profiles/r03/bench_train_r03t_{one1,side1}.json)
    }();
    return v;
}

static int next_sync_event(y6_plan* p, hipEvent_t* out) {
    constexpr size_t kRing = 512;    // an event is re-recorded only after 511 younger ones: its waits are long enqueued
    if (p->sync_ev.size() < kRing) {
        hipEvent_t e;
        Y6_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->sync_ev.push_back(e);
        *out = e;
        return Y6_OK;
    }
    *out = p->sync_ev[p->sync_pos];
    p->sync_pos = (p->sync_pos + 1) % kRing;
    return Y6_OK;
}

static int run_ops(y6_plan* p, hipStream_t s, size_t first, size_t last) {
    bool any_side = false;
    if (side_stream_enabled())
        for (size_t i = first; i < last && i < p->side.size(); ++i) any_side = any_side || p->side[i];
    if (!any_side) {
        for (size_t i = first; i < last; ++i) {
            int rc = run_op(p->ops[i], s);
            if (rc) return rc;
        }
        return Y6_OK;
    }
    if (!p->side_stream) {
        // A/B switch Y6_SIDE_PRIO: "high" / "low" = the side stream's queue at the device's highest / lowest priority (its
        // kernels are dispatched before / after the main stream's whenever both have workgroups waiting)
        static const char* prio = getenv("Y6_SIDE_PRIO");
        if (prio && (prio[0] == 'h' || prio[0] == 'l')) {
            int lo = 0, hi = 0;
            Y6_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = numerically greatest = least priority
            Y6_HIP(hipStreamCreateWithPriority(&p->side_stream, hipStreamNonBlocking, prio[0] == 'h' ? hi : lo));
        } else {
            Y6_HIP(hipStreamCreateWithFlags(&p->side_stream, hipStreamNonBlocking));
        }
    }
    bool main_ahead = true, side_used = false;   // main_ahead: the main stream holds work the side stream has not been ordered behind
    for (size_t i = first; i < last; ++i) {
        const bool on_side = i < p->side.size() && p->side[i];
        if (on_side) {
            if (main_ahead) {
                hipEvent_t e;
                int rc = next_sync_event(p, &e);
                if (rc) return rc;
                Y6_HIP(hipEventRecord(e, s));
                Y6_HIP(hipStreamWaitEvent(p->side_stream, e, 0));
                main_ahead = false;
            }
            int rc = run_op(p->ops[i], p->side_stream);
            if (rc) return rc;
            side_used = true;
            ++p->side_pending;
        } else {
            int rc = run_op(p->ops[i], s);
            if (rc) return rc;
            main_ahead = true;
        }
    }
    if (side_used) {
        hipEvent_t e;
        int rc = next_sync_event(p, &e);
        if (rc) return rc;
        Y6_HIP(hipEventRecord(e, p->side_stream));
        Y6_HIP(hipStreamWaitEvent(s, e, 0));
        p->side_pending = 0;
    }
    return Y6_OK;
}

// Side-stream ops enqueued by eager runs that the caller's stream has not been ordered behind yet: 0 after every y6_plan_run /
// y6_plan_run_range returns (they join before returning).  parallel.GradReducer asserts it before it records the event an
// all-reduce of the finished gradient chunk waits for.
extern "C" int y6_plan_side_pending(const y6_plan* p) { return p ? p->side_pending : 0; }

// ---- two-stream schedule (inference plans; yolov6_amd/schedule.py decides it from the ops' tensor views) ----------------
// order[n]: a permutation of the ops = the order they are enqueued in; stream[n] (by op index): 0 = the caller's stream,
// 1 = the plan's side stream; edges[2 * nedges]: (src, dst) pairs on DIFFERENT streams with src enqueued before dst - dst waits
// for an event recorded right behind src.  Every run forks the side stream off the caller's stream first and joins it back
// last, so for the caller (and for the previous / next run of the same plan) the main stream still orders everything.
// n = 0 drops the schedule.  Only y6_plan_run uses it; ranges, timed runs and captured graphs keep plan order on one stream.
extern "C" int y6_plan_set_schedule(y6_plan* p, const int32_t* order, const int32_t* stream, int n, const int32_t* edges, int nedges) {
    Y6_REQUIRE(p, "plan_set_schedule: null plan");
    p->sched_order.clear();
    p->sched_stream.clear();
    p->sched_waits.clear();
    p->sched_records.clear();
    if (n == 0) return Y6_OK;
    Y6_REQUIRE(order && stream && n == (int)p->ops.size() && nedges >= 0 && (edges || nedges == 0), "plan_set_schedule: bad arguments");
    std::vector<int> pos((size_t)n, -1);
    for (int i = 0; i < n; ++i) {
        Y6_REQUIRE(order[i] >= 0 && order[i] < n && pos[order[i]] < 0, "plan_set_schedule: order is not a permutation of the ops");
        Y6_REQUIRE(stream[i] == 0 || stream[i] == 1, "plan_set_schedule: stream must be 0 or 1");
        pos[order[i]] = i;
    }
    std::vector<std::vector<int32_t>> waits((size_t)n);
    std::vector<char> records((size_t)n, 0);
    for (int e = 0; e < nedges; ++e) {
        const int a = edges[2 * e], b = edges[2 * e + 1];
        Y6_REQUIRE(a >= 0 && a < n && b >= 0 && b < n && pos[a] < pos[b] && stream[a] != stream[b],
                   "plan_set_schedule: edge %d (%d -> %d) must point forward in the enqueue order, across streams", e, a, b);
        waits[b].push_back(a);
        records[a] = 1;
    }
    {   // the events of one run come out of a 512-entry ring (next_sync_event): fork + join + one per recording op must fit,
        // or an event would be re-recorded while a later op of the same run still has to wait for its first recording
        int nrec = 0;
        for (char r : records) nrec += r;
        Y6_REQUIRE(nrec + 2 <= 510, "plan_set_schedule: %d cross-stream events in one run exceed the event ring", nrec);
    }
    p->sched_order.assign(order, order + n);
    p->sched_stream.assign(stream, stream + n);
    p->sched_waits.swap(waits);
    p->sched_records.swap(records);
    return Y6_OK;
}

static int run_scheduled(y6_plan* p, hipStream_t s) {
    const size_t n = p->ops.size();
    if (!p->side_stream) Y6_HIP(hipStreamCreateWithFlags(&p->side_stream, hipStreamNonBlocking));
    hipEvent_t e;
    int rc = next_sync_event(p, &e);   // fork: the side stream runs behind everything the caller has enqueued so far
    if (rc) return rc;
    Y6_HIP(hipEventRecord(e, s));
    Y6_HIP(hipStreamWaitEvent(p->side_stream, e, 0));
    p->sched_ev.assign(n, nullptr);
    for (size_t q = 0; q < n; ++q) {
        const int i = p->sched_order[q];
        hipStream_t st = p->sched_stream[i] ? p->side_stream : s;
        for (int w : p->sched_waits[i]) {
            Y6_REQUIRE(p->sched_ev[w] != nullptr, "plan_run: scheduled op %d waits for op %d, which has not been enqueued", i, w);
            Y6_HIP(hipStreamWaitEvent(st, p->sched_ev[w], 0));
        }
        rc = run_op(p->ops[i], st);
        if (rc) return rc;
        if (p->sched_records[i]) {
            rc = next_sync_event(p, &e);
            if (rc) return rc;
            Y6_HIP(hipEventRecord(e, st));
            p->sched_ev[i] = e;
        }
    }
    rc = next_sync_event(p, &e);       // join
    if (rc) return rc;
    Y6_HIP(hipEventRecord(e, p->side_stream));
    Y6_HIP(hipStreamWaitEvent(s, e, 0));
    return Y6_OK;
}

extern "C" int y6_plan_timing_begin(y6_plan* p, int slots) {
    Y6_REQUIRE(p && slots > 0, "plan_timing_begin: bad arguments");
    drop_events(p);
    const size_t n = (size_t)slots * (p->ops.size() + 1);
    p->events.resize(n);
    for (size_t i = 0; i < n; ++i) Y6_HIP(hipEventCreate(&p->events[i]));
    p->slots = slots;
    return Y6_OK;
}

// Timing granularity of y6_plan_run_timed: an event after EVERY op ("op", default) or only where the kernel
// class (kind, ksize, stride) changes ("class": the ~75 event packets per pass shrink to ~45; a class's time is
// exact, an op's share inside a run of equal-class ops is apportioned by FLOPs).  Env Y6_TIMED_EVENTS.
static int op_class(const Op& op) {
    return op.kind == Y6_OP_CONV ? 100 + op.conv.ksize * 10 + op.conv.stride : op.kind;
}
static bool timed_by_class() {
    static const int v = [] {
        const char* e = getenv("Y6_TIMED_EVENTS");
        return (e && strcmp(e, "class") == 0) ? 1 : 0;
    }();
    return v != 0;
}
static bool event_after(const y6_plan* p, size_t i) {   // is there an event between op i and op i+1 ?
    if (!timed_by_class() || i + 1 >= p->ops.size()) return true;
    return op_class(p->ops[i]) != op_class(p->ops[i + 1]);
}

extern "C" int y6_plan_run_timed(y6_plan* p, void* stream) {
    // Eager run with hipEvents between ops (on `stream`, the stream the kernels are launched on).
    // No synchronisation here; y6_plan_timing_read sums the slots afterwards.
    Y6_REQUIRE(p && p->slots_used < p->slots, "plan_run_timed: no timing slot left (call y6_plan_timing_begin)");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = p->ops.size();
    hipEvent_t* ev = &p->events[(size_t)p->slots_used * (n + 1)];
    Y6_HIP(hipEventRecord(ev[0], s));
    for (size_t i = 0; i < n; ++i) {
        int rc = run_op(p->ops[i], s);
        if (rc) return rc;
        if (event_after(p, i)) Y6_HIP(hipEventRecord(ev[i + 1], s));
    }
    ++p->slots_used;
    return Y6_OK;
}

extern "C" int y6_plan_timing_read(y6_plan* p, float* ms_sum, int cap) {
    // ms_sum[i] = total milliseconds op i took over all used slots. Call after synchronising.
    Y6_REQUIRE(p && ms_sum, "plan_timing_read: null argument");
    const size_t n = p->ops.size();
    for (size_t i = 0; i < n && (int)i < cap; ++i) ms_sum[i] = 0.f;
    for (int sl = 0; sl < p->slots_used; ++sl) {
        hipEvent_t* ev = &p->events[(size_t)sl * (n + 1)];
        size_t g0 = 0;
        for (size_t i = 0; i < n; ++i) {
            if (!event_after(p, i)) continue;
            float ms = 0.f;   // ops g0..i ran between ev[g0] and ev[i+1]
            Y6_HIP(hipEventElapsedTime(&ms, ev[g0], ev[i + 1]));
            double tot = 0.0;
            for (size_t j = g0; j <= i; ++j) {
                double f = 0.0, by = 0.0;
                op_cost(p->ops[j], &f, &by);
                tot += f > 0.0 ? f : by;
            }
            for (size_t j = g0; j <= i && (int)j < cap; ++j) {
                double f = 0.0, by = 0.0;
                op_cost(p->ops[j], &f, &by);
                const double share = tot > 0.0 ? (f > 0.0 ? f : by) / tot : 1.0 / (double)(i - g0 + 1);
                ms_sum[j] += (float)(ms * share);
            }
            g0 = i + 1;
        }
    }
    return p->slots_used;
}

extern "C" int y6_plan_op_info(const y6_plan* p, int i, int32_t* kind, int32_t* variant, int32_t* ksize, int32_t* stride,
                               double* flops, double* bytes) {
    Y6_REQUIRE(p && i >= 0 && i < (int)p->ops.size(), "plan_op_info: bad index");
    const Op& op = p->ops[i];
    if (kind) *kind = op.kind;
    if (variant) *variant = op.kind == Y6_OP_CONV ? op.conv.variant : -1;
    if (ksize) *ksize = op.kind == Y6_OP_CONV ? op.conv.ksize : (op.kind == Y6_OP_GENERIC ? op.gtag : 0);
    if (stride) *stride = op.kind == Y6_OP_CONV ? op.conv.stride : 0;
    double f = 0.0, by = 0.0;
    op_cost(op, &f, &by);
    if (flops) *flops = f;
    if (bytes) *bytes = by;
    return Y6_OK;
}
extern "C" int y6_plan_num_ops(const y6_plan* p) { return p ? (int)p->ops.size() : 0; }

#define PLAN_ADD_PROLOGUE()                                   \
    Y6_REQUIRE(p && d, "plan_add: null argument");            \
    drop_graph(p);                                            \
    Op op;                                                    \
    memset((void*)&op, 0, sizeof(op))

extern "C" int y6_plan_add_conv(y6_plan* p, const y6_conv_desc* d) {
    PLAN_ADD_PROLOGUE();
    op.kind = Y6_OP_CONV;
    op.conv = *d;
    p->ops.push_back(op);
    return Y6_OK;
}
extern "C" int y6_plan_add_convt(y6_plan* p, const y6_convt_desc* d) {
    PLAN_ADD_PROLOGUE();
    op.kind = Y6_OP_CONVT;
    op.convt = *d;
    p->ops.push_back(op);
    return Y6_OK;
}
extern "C" int y6_plan_add_stem(y6_plan* p, const y6_stem_desc* d) {
    PLAN_ADD_PROLOGUE();
    op.kind = Y6_OP_STEM;
    op.stem = *d;
    p->ops.push_back(op);
    return Y6_OK;
}
extern "C" int y6_plan_add_sppf(y6_plan* p, const y6_tensor* x, const y6_tensor* y1, const y6_tensor* y2,
                                const y6_tensor* y3) {
    const y6_tensor* d = x;
    PLAN_ADD_PROLOGUE();
    Y6_REQUIRE(y1 && y2 && y3, "plan_add_sppf: null tensor");
    op.kind = Y6_OP_SPPF;
    op.t[0] = *x;
    op.t[1] = *y1;
    op.t[2] = *y2;
    op.t[3] = *y3;
    p->ops.push_back(op);
    return Y6_OK;
}
extern "C" int y6_plan_add_decode(y6_plan* p, const y6_decode_desc* d) {
    PLAN_ADD_PROLOGUE();
    op.kind = Y6_OP_DECODE;
    op.dec = *d;
    p->ops.push_back(op);
    return Y6_OK;
}
extern "C" int y6_plan_add_nchw2nhwc(y6_plan* p, const void* src, int src_dtype, const y6_tensor* d) {
    PLAN_ADD_PROLOGUE();
    Y6_REQUIRE(src, "plan_add_nchw2nhwc: null src");
    op.kind = Y6_OP_NCHW2NHWC;
    op.src = src;
    op.dtype = src_dtype;
    op.t[0] = *d;
    p->ops.push_back(op);
    return Y6_OK;
}
extern "C" int y6_plan_add_nhwc2nchw(y6_plan* p, const y6_tensor* d, void* dst, int dst_dtype) {
    PLAN_ADD_PROLOGUE();
    Y6_REQUIRE(dst, "plan_add_nhwc2nchw: null dst");
    op.kind = Y6_OP_NHWC2NCHW;
    op.dst = dst;
    op.dtype = dst_dtype;
    op.t[0] = *d;
    p->ops.push_back(op);
    return Y6_OK;
}

int y6_plan_push_generic(y6_plan* p, y6_generic_fn fn, const void* desc, size_t size, int tag, double flops, double bytes) {
    Y6_REQUIRE(p && fn && desc && size <= Y6_GENERIC_BLOB, "plan_push_generic: bad arguments");
    drop_graph(p);
    p->ops.emplace_back();
    Op& op = p->ops.back();
    memset((void*)&op, 0, sizeof(op));
    op.kind = Y6_OP_GENERIC;
    op.gfn = fn;
    op.gtag = tag;
    op.gflops = flops;
    op.gbytes = bytes;
    op.gout_off = -1;
    op.gin_off = -1;
    memcpy(op.blob, desc, size);
    return Y6_OK;
}

int y6_plan_mark_output(y6_plan* p, size_t offset) {
    Y6_REQUIRE(p && !p->ops.empty() && p->ops.back().kind == Y6_OP_GENERIC && offset + sizeof(void*) <= Y6_GENERIC_BLOB,
               "plan_mark_output: no generic op to mark");
    p->ops.back().gout_off = (int)offset;
    return Y6_OK;
}

extern "C" int y6_plan_set_nms_sink(y6_plan* p, const y6_nms_sink* sink) {
    Y6_REQUIRE(p && sink, "plan_set_nms_sink: null argument");
    int changed = 0;
    for (Op& op : p->ops) {
        if (op.kind != Y6_OP_GENERIC || op.gtag != Y6_TOP_PRED_DECODE) continue;
        y6_pred_decode_desc d;
        memcpy(&d, op.blob, sizeof(d));
        d.cand = *sink;
        if (sink->workspace != nullptr && !y6_head_pred_decode_supported(&d)) continue;   // this op keeps running without a sink
        memcpy(op.blob, &d, sizeof(d));
        ++changed;
    }
    if (changed) drop_graph(p);
    return changed;
}

int y6_plan_mark_input(y6_plan* p, size_t offset) {
    Y6_REQUIRE(p && !p->ops.empty() && p->ops.back().kind == Y6_OP_GENERIC && offset + sizeof(void*) <= Y6_GENERIC_BLOB,
               "plan_mark_input: no generic op to mark");
    p->ops.back().gin_off = (int)offset;
    return Y6_OK;
}

// the boundary-input pointer of an op (stem, NCHW->NHWC adapter, a fused op marked with y6_plan_mark_input), or null
static const void** boundary_input(Op& op) {
    if (op.kind == Y6_OP_STEM) return &op.stem.in_nchw;
    if (op.kind == Y6_OP_NCHW2NHWC) return &op.src;
    if (op.kind == Y6_OP_GENERIC && op.gin_off >= 0) return reinterpret_cast<const void**>(op.blob + op.gin_off);
    return nullptr;
}

extern "C" int y6_plan_rebind(y6_plan* p, const void* old_ptr, const void* new_ptr) {
    // Point every op that reads the caller's boundary tensor `old_ptr` at `new_ptr` (same shape
    // and dtype).  Drops a captured graph: its kernel nodes baked the old address.
    Y6_REQUIRE(p && old_ptr && new_ptr, "plan_rebind: null argument");
    int n = 0;
    for (Op& op : p->ops) {
        const void** slot = boundary_input(op);
        if (slot && *slot == old_ptr) {
            *slot = new_ptr;
            ++n;
        }
    }
    if (n) drop_graph(p);
    return n;
}

extern "C" int y6_plan_rebind_input(y6_plan* p, int index, const void* new_ptr) {
    Y6_REQUIRE(p && index >= 0 && new_ptr, "plan_rebind_input: bad argument");
    int k = 0;
    for (Op& op : p->ops) {
        const void** slot = boundary_input(op);
        if (!slot) continue;
        if (k++ != index) continue;
        if (*slot == new_ptr) return 0;
        *slot = new_ptr;
        drop_graph(p);
        return 1;
    }
    y6_set_error("plan_rebind_input: the plan has only %d boundary inputs (index %d)", k, index);
    return Y6_EINVAL;
}

extern "C" int y6_plan_rebind_output(y6_plan* p, const void* old_ptr, void* new_ptr) {
    // Point every op that WRITES the caller-visible boundary tensor `old_ptr` (the decode kernel's [B,A,5+nc] output, an
    // NHWC->NCHW adapter's destination) at `new_ptr` (same shape and dtype): lets the host hand out results without a
    // copy by alternating between output buffers.  Drops a captured graph: its kernel nodes baked the old address.
    Y6_REQUIRE(p && old_ptr && new_ptr, "plan_rebind_output: null argument");
    int n = 0;
    for (Op& op : p->ops) {
        if (op.kind == Y6_OP_DECODE && op.dec.out == old_ptr) {
            op.dec.out = (float*)new_ptr;
            ++n;
        }
        if (op.kind == Y6_OP_NHWC2NCHW && op.dst == old_ptr) {
            op.dst = new_ptr;
            ++n;
        }
        if (op.kind == Y6_OP_GENERIC && op.gout_off >= 0) {   // fused ops that write the boundary tensor (head_pred_decode)
            void* cur = nullptr;
            memcpy(&cur, op.blob + op.gout_off, sizeof(cur));
            if (cur == old_ptr) {
                memcpy(op.blob + op.gout_off, &new_ptr, sizeof(new_ptr));
                ++n;
            }
        }
    }
    if (n) drop_graph(p);
    return n;
}

extern "C" int y6_plan_run(y6_plan* p, void* stream) {
    Y6_REQUIRE(p, "plan_run: null plan");
    hipStream_t s = (hipStream_t)stream;
    if (p->exec) {
        Y6_HIP(hipGraphLaunch(p->exec, s));
        return Y6_OK;
    }
    if (p->sched_order.size() == p->ops.size() && !p->ops.empty()) return run_scheduled(p, s);
    return run_ops(p, s, 0, p->ops.size());
}

extern "C" int y6_plan_run_range(y6_plan* p, void* stream, int first, int last) {
    Y6_REQUIRE(p && first >= 0 && first <= last && last <= (int)p->ops.size(), "plan_run_range: bad range");
    return run_ops(p, (hipStream_t)stream, (size_t)first, (size_t)last);
}

// ---- data dependences between ops (for the multi-stream capture) -------------------------------------
namespace {
struct Span {            // channel slice [c0, c1) of the buffer at `base` (whole buffer: [0, INT_MAX))
    const void* base;
    int c0, c1;
};
inline bool overlaps(const Span& a, const Span& b) { return a.base && a.base == b.base && a.c0 < b.c1 && b.c0 < a.c1; }
inline Span span_of(const y6_tensor& t) { return Span{t.data, t.coff, t.coff + t.C}; }
inline Span whole(const void* p) { return Span{p, 0, 0x7fffffff}; }

void op_access(const Op& op, std::vector<Span>* rd, std::vector<Span>* wr) {
    switch (op.kind) {
        case Y6_OP_CONV:
            rd->push_back(span_of(op.conv.in));
            if (op.conv.res.data) rd->push_back(span_of(op.conv.res));
            wr->push_back(span_of(op.conv.out));
            break;
        case Y6_OP_CONVT:
            rd->push_back(span_of(op.convt.in));
            wr->push_back(span_of(op.convt.out));
            break;
        case Y6_OP_STEM:
            rd->push_back(whole(op.stem.in_nchw));
            wr->push_back(span_of(op.stem.out));
            break;
        case Y6_OP_SPPF:
            rd->push_back(span_of(op.t[0]));
            for (int i = 1; i < 4; ++i) wr->push_back(span_of(op.t[i]));
            break;
        case Y6_OP_DECODE:
            for (int l = 0; l < op.dec.n_levels; ++l) {
                rd->push_back(span_of(op.dec.cls[l]));
                rd->push_back(span_of(op.dec.reg[l]));
            }
            wr->push_back(whole(op.dec.out));
            break;
        case Y6_OP_NCHW2NHWC:
            rd->push_back(whole(op.src));
            wr->push_back(span_of(op.t[0]));
            break;
        case Y6_OP_NHWC2NCHW:
            rd->push_back(span_of(op.t[0]));
            wr->push_back(whole(op.dst));
            break;
    }
}
}  // namespace

// Capture the plan into a hipGraph.  EXPERIMENTAL, off by default: with Y6_GRAPH_STREAMS=2 independent ops - the three head levels
// and their cls / reg branches, the neck's lateral convs - are captured on two streams joined by events, so the
// graph has parallel branches and a replay can overlap their latency-bound small kernels.  Dependences are
// derived from the tensor views (RAW, WAR, WAW on overlapping channel slices of the same buffer).  Measured on
// YOLOv6-S b32 (tools/graph_ab.py, r14): 2.831 ms with one stream, 2.768 ms with two - the persistent conv
// kernels already fill the chip, so overlapping them buys 2 %; not used by bench.py.
extern "C" int y6_plan_capture(y6_plan* p, void* stream) {
    Y6_REQUIRE(p, "plan_capture: null plan");
    hipStream_t s = (hipStream_t)stream;
    drop_graph(p);
    int nstreams = 1;
    if (const char* e = getenv("Y6_GRAPH_STREAMS")) nstreams = atoi(e);
    const bool dbg = getenv("Y6_PLAN_DEBUG") != nullptr;
#define Y6_DBG(...) do { if (dbg) { fprintf(stderr, "[plan_capture] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
    Y6_DBG("begin: %zu ops, %d streams", p->ops.size(), nstreams);
    if (nstreams < 1) nstreams = 1;
    if (nstreams > 2) nstreams = 2;
    for (const Op& op : p->ops)
        if (op.kind == Y6_OP_GENERIC) nstreams = 1;   // no tensor-view dependence info for generic ops   // hipStreamEndCapture of a 4-stream capture of the P6 models crashes inside the runtime (ROCm 7.2): capped
    const size_t n = p->ops.size();

    std::vector<hipStream_t> st(nstreams, s);
    std::vector<hipEvent_t> ev(n + 1, nullptr);   // ev[i]: recorded after op i when another stream needs it; ev[n]: start
    std::vector<std::vector<int>> deps(n);
    if (nstreams > 1) {
        std::vector<std::vector<Span>> rd(n), wr(n);
        for (size_t i = 0; i < n; ++i) op_access(p->ops[i], &rd[i], &wr[i]);
        for (size_t j = 0; j < n; ++j)
            for (size_t i = 0; i < j; ++i) {
                bool d = false;
                for (const Span& w : wr[i]) {
                    for (const Span& r : rd[j]) d = d || overlaps(w, r);
                    for (const Span& w2 : wr[j]) d = d || overlaps(w, w2);
                }
                for (const Span& r : rd[i])
                    for (const Span& w2 : wr[j]) d = d || overlaps(r, w2);
                if (d) deps[j].push_back((int)i);
            }
        for (int k = 1; k < nstreams; ++k) Y6_HIP(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    }
    auto cleanup = [&]() {
        for (int k = 1; k < nstreams; ++k)
            if (st[k] != s) (void)hipStreamDestroy(st[k]);
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    };

    Y6_DBG("deps computed");
    Y6_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = Y6_OK;
    hipError_t herr = hipSuccess;
#define Y6_CAP(call)                                    \
    do {                                                \
        if (rc == Y6_OK && herr == hipSuccess) herr = (call); \
    } while (0)
    std::vector<int> on(n, 0);                 // stream of op i
    std::vector<int> tail(nstreams, -1);       // last op queued on stream k
    std::vector<char> joined(nstreams, 0);     // stream k is part of the capture
    joined[0] = 1;
    if (nstreams > 1) {
        Y6_CAP(hipEventCreateWithFlags(&ev[n], hipEventDisableTiming));
        Y6_CAP(hipEventRecord(ev[n], s));
    }
    for (size_t j = 0; j < n && rc == Y6_OK && herr == hipSuccess; ++j) {
        int k = 0;
        if (nstreams > 1) {
            // stay on the stream of the latest producer if nothing was queued behind it; else take the stream
            // that has been idle longest
            k = -1;
            int latest = -1;
            for (int d : deps[j])
                if (d > latest) latest = d;
            if (latest >= 0 && tail[on[latest]] == latest) k = on[latest];
            if (k < 0) {
                k = 0;
                for (int q = 1; q < nstreams; ++q)
                    if (tail[q] < tail[k]) k = q;
            }
            if (!joined[k]) {
                Y6_CAP(hipStreamWaitEvent(st[k], ev[n], 0));
                joined[k] = 1;
            }
            for (int d : deps[j]) {
                if (on[d] == k) continue;   // same stream: in order
                Y6_CAP(hipStreamWaitEvent(st[k], ev[d], 0));
            }
        }
        Y6_DBG("op %zu kind %d -> stream %d (%zu deps) herr %d", j, p->ops[j].kind, k, deps[j].size(), (int)herr);
        if (rc == Y6_OK && herr == hipSuccess) rc = run_op(p->ops[j], st[k]);
        if (nstreams > 1) {   // every op gets its event right behind it (a late record hangs off unrelated later ops)
            Y6_CAP(hipEventCreateWithFlags(&ev[j], hipEventDisableTiming));
            Y6_CAP(hipEventRecord(ev[j], st[k]));
        }
        on[j] = k;
        tail[k] = (int)j;
    }
    // join every side stream back into the origin stream
    for (int k = 1; k < nstreams && rc == Y6_OK && herr == hipSuccess; ++k) {
        if (!joined[k]) continue;
        hipEvent_t je = nullptr;
        Y6_CAP(hipEventCreateWithFlags(&je, hipEventDisableTiming));
        ev.push_back(je);
        Y6_CAP(hipEventRecord(je, st[k]));
        Y6_CAP(hipStreamWaitEvent(s, je, 0));
    }
#undef Y6_CAP
    Y6_DBG("joined; rc %d herr %d", rc, (int)herr);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    Y6_DBG("end capture: %d graph %p", (int)e, (void*)g);
    cleanup();
    Y6_DBG("cleanup done");
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (herr != hipSuccess || e != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        y6_set_error("plan_capture failed: %s", hipGetErrorString(herr != hipSuccess ? herr : e));
        return Y6_EHIP;
    }
    p->graph = g;
    Y6_HIP(hipGraphInstantiate(&p->exec, g, nullptr, nullptr, 0));
    Y6_DBG("instantiated");
#undef Y6_DBG
    return Y6_OK;
}

static int time_op(const Op& op, hipStream_t s, int iters, float* ms_out) {
    hipEvent_t e0, e1;
    Y6_HIP(hipEventCreate(&e0));
    Y6_HIP(hipEventCreate(&e1));
    int rc = run_op(op, s);  // warm
    if (rc == Y6_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && rc == Y6_OK; ++i) rc = run_op(op, s);
        (void)hipEventRecord(e1, s);
        hipError_t e = hipEventSynchronize(e1);
        if (rc == Y6_OK && e != hipSuccess) {
            y6_set_error("op failed during timing: %s", hipGetErrorString(e));
            rc = Y6_EHIP;
        }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms / (iters > 0 ? iters : 1);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// In-context timing for the autotuner: `op` is timed right behind the op that precedes it in the plan (with that
// op's current kernel choice).  A burst of the same kernel over the same tensors flatters it twice: its input
// stays resident from the previous repetition, and it never pays the first-launch-after-a-kernel-switch cost
// (+3...35 us depending on the box) that it pays in the real sequence.  Minimum over `iters` pairs.
static int time_op_after(const Op& prev, const Op& op, hipStream_t s, int iters, float* ms_out) {
    hipEvent_t e0, e1;
    Y6_HIP(hipEventCreate(&e0));
    Y6_HIP(hipEventCreate(&e1));
    int rc = run_op(op, s);  // warm
    float best = 1e30f;
    for (int i = 0; i < iters && rc == Y6_OK; ++i) {
        rc = run_op(prev, s);
        if (rc) break;
        (void)hipEventRecord(e0, s);
        rc = run_op(op, s);
        (void)hipEventRecord(e1, s);
        hipError_t e = hipEventSynchronize(e1);
        if (rc == Y6_OK && e != hipSuccess) {
            y6_set_error("op failed during timing: %s", hipGetErrorString(e));
            rc = Y6_EHIP;
        }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    *ms_out = best;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

extern "C" int y6_plan_autotune(y6_plan* p, void* stream, int iters) {
    Y6_REQUIRE(p, "plan_autotune: null plan");
    hipStream_t s = (hipStream_t)stream;
    drop_graph(p);
    if (iters < 1) iters = 3;
    const int nv = y6_conv_variants();
    const char* logpath = getenv("Y6_AUTOTUNE_LOG");   // optional: append every (op, variant, ms) measurement
    FILE* logf = logpath ? fopen(logpath, "a") : nullptr;
    // optional persistent choices (env Y6_AUTOTUNE_CACHE=<file>): "signature variant-name" per line, looked up
    // before measuring and appended after - a deployment tunes once per machine, and a profiler run of the
    // same command sees only the chosen kernels.
    const char* cachepath = getenv("Y6_AUTOTUNE_CACHE");
    std::vector<std::pair<std::string, std::string>> cache;
    if (cachepath) {
        if (FILE* cf = fopen(cachepath, "r")) {
            char key[256], name[64];
            while (fscanf(cf, "%255s %63s", key, name) == 2) cache.emplace_back(key, name);
            fclose(cf);
        }
    }
    auto signature = [](const y6_conv_desc& c) {
        char b[256];
        snprintf(b, sizeof(b), "k%ds%d_ci%d_co%d_b%d_h%d_w%d_ics%d_ocs%d_act%d_res%d_ps%d", c.ksize, c.stride, c.in.C,
                 c.out.C, c.in.B, c.in.H, c.in.W, c.in.cstride, c.out.cstride, c.act, c.res.data != nullptr,
                 c.post_scale != nullptr);
        return std::string(b);
    };
    // Candidate set.  Timing a layer alone in a burst flatters kernels that lose in sequence: with every
    // variant allowed the plan hops between seven kernels and the whole step is 2.4 % slower than with the
    // per-tap kernels + pipe_c2p2/pipe_c2p1 only (same box, tools/gpu_ab_variants.sh, r13).  The others stay
    // built, tested and selectable: Y6_AUTOTUNE_EXCLUDE="" allows all, "15,16" excludes just those.
    std::vector<char> excluded(nv, 0);
    const char* ex = getenv("Y6_AUTOTUNE_EXCLUDE");
    // Default candidates: every built form except the per-tap 2-fragment tiles on 3x3 stride-1 layers' behalf (they never win
    // there but hop the plan between kernels).  Y6_AUTOTUNE_EXCLUDE="" allows all; a comma list of variant NAMES (or indices)
    // excludes just those.  History of what was measured and retired: DESIGN 6 / 6b / 6c.
    if (!ex) ex = "";
    {
        for (const char* c = ex; *c;) {
            const char* e = c;
            while (*e && *e != ',') ++e;
            const std::string tok(c, e);
            if (!tok.empty()) {
                char* end = nullptr;
                const long v = strtol(tok.c_str(), &end, 10);
                if (end && *end == 0) {
                    if (v >= 0 && v < nv) excluded[v] = 1;
                } else {
                    for (int i = 0; i < nv; ++i)
                        if (tok == y6_conv_variant_name(i)) excluded[i] = 1;
                }
            }
            c = *e ? e + 1 : e;
        }
    }
    static const bool in_context = !(getenv("Y6_AUTOTUNE_BURST") != nullptr);   // A/B switch: old burst timing
    std::vector<std::vector<float>> times(p->ops.size());
    std::vector<char> measured(p->ops.size(), 0);
    for (size_t i = 0; i < p->ops.size(); ++i) {
        Op& op = p->ops[i];
        if (op.kind != Y6_OP_CONV) continue;
        int best = -1;
        float best_ms = 1e30f;
        const std::string sig = signature(op.conv);
        if (cachepath) {
            for (const auto& kv : cache) {
                if (kv.first != sig) continue;
                for (int v = 1; v < nv; ++v)
                    if (kv.second == y6_conv_variant_name(v) && y6_conv_variant_supports(&op.conv, v)) best = v;
            }
            if (best >= 0) {
                op.conv.variant = best;
                continue;
            }
        }
        std::vector<float>& row = times[i];
        row.assign(nv, 1e30f);
        bool reused = false;
        for (size_t j = 0; j < i && !reused; ++j) {   // an identical layer measured earlier in this call: same numbers
            if (!measured[j] || signature(p->ops[j].conv) != sig) continue;
            row = times[j];
            for (int v = 1; v < nv; ++v)
                if (row[v] < best_ms && y6_conv_variant_supports(&op.conv, v)) {
                    best_ms = row[v];
                    best = v;
                }
            reused = best > 0;
        }
        for (int v = 1; v < nv && !reused; ++v) {  // variant 0 (naive) is a cross-check, never a candidate
            if (!y6_conv_variant_supports(&op.conv, v) || excluded[v]) continue;
            Op trial = op;
            trial.conv.variant = v;
            float ms = 1e30f;
            int rc = Y6_OK;
            if (i > 0 && in_context) {
                rc = time_op_after(p->ops[i - 1], trial, s, 3 * iters, &ms);
            } else {
                for (int rep = 0; rep < 3 && rc == Y6_OK; ++rep) {   // min of three short bursts: robust to clock ramps
                    float t = 0.f;
                    rc = time_op(trial, s, iters, &t);
                    if (t < ms) ms = t;
                }
            }
            if (rc) {
                if (logf) fclose(logf);
                return rc;
            }
            if (logf)
                fprintf(logf, "op %zu k%d s%d cin %d cout %d in %dx%dx%d variant %s ms %.5f gflops %.1f\n", i,
                        op.conv.ksize, op.conv.stride, op.conv.in.C, op.conv.out.C, op.conv.in.B, op.conv.in.H,
                        op.conv.in.W, y6_conv_variant_name(v), ms, y6_conv_flops(&op.conv) / (ms * 1e6));
            row[v] = ms;
            if (ms < best_ms) {
                best_ms = ms;
                best = v;
            }
        }
        if (best < 0) best = y6_conv_variant_supports(&op.conv, 0) ? 0 : -1;
        Y6_REQUIRE(best >= 0, "plan_autotune: op %zu has no runnable conv variant", i);
        op.conv.variant = best;
        measured[i] = best > 0;
        if (logf) fflush(logf);
    }
    // Consolidate: burst timings of the top variants are often within 1-3 % of each other, and a plan that
    // hops between many different kernels pays for it at every switch (cold instruction cache, LDS / scratch
    // reconfiguration: +15-35 us on the first launch after a switch, r12).  Variants are ranked by how many
    // layers of this (ksize, stride) class they win, and a layer takes the highest ranked variant that is
    // within 3 % of its own best time.
    {
        std::vector<std::vector<int>> wins(4, std::vector<int>(nv, 0));
        auto klass = [](const y6_conv_desc& c) { return (c.ksize == 3 ? 0 : 2) + (c.stride == 2 ? 1 : 0); };
        for (size_t i = 0; i < p->ops.size(); ++i)
            if (measured[i]) wins[klass(p->ops[i].conv)][p->ops[i].conv.variant]++;
        for (size_t i = 0; i < p->ops.size(); ++i) {
            if (!measured[i]) continue;
            Op& op = p->ops[i];
            const std::vector<float>& row = times[i];
            const std::vector<int>& w = wins[klass(op.conv)];
            const float best_ms = row[op.conv.variant];
            int pick = op.conv.variant;
            for (int v = 1; v < nv; ++v)
                if (row[v] <= best_ms * 1.03f && (w[v] > w[pick] || (w[v] == w[pick] && row[v] < row[pick]))) pick = v;
            if (logf && pick != op.conv.variant)
                fprintf(logf, "op %zu consolidated %s (%.5f ms) -> %s (%.5f ms)\n", i, y6_conv_variant_name(op.conv.variant),
                        best_ms, y6_conv_variant_name(pick), row[pick]);
            op.conv.variant = pick;
        }
    }
    // ---- whole-step refinement (round 5).  A layer timed by itself - even right behind its predecessor - is not the layer in the
    // step: the FIRST launch of a kernel function that has not run for a few hundred microseconds costs 20-35 us more than the
    // following ones (profiles/r05/first_of_run_*.json: 256 -> 256 @40x40 on wreg_p7 takes 80 us as the first wreg_p7 launch of
    // the step and 58 us when the 80x80 layers in front of it ran on wreg_p7 too), so a table of per-layer winners that hops
    // between kernel functions loses to a table with fewer functions - by 4 % of the step on one box, while on another box the
    // per-layer table wins.  So the STEP is what gets timed: start from the better of {the per-layer table above, the shape-derived
    // table (y6_conv_default_variant: what a plan that is not autotuned runs)}, then walk the layer groups (equal signature =
    // equal variant) by descending time and try, for every group, the variants whose per-layer time is within 30 % of the group's
    // best; a change is kept if the whole step gets faster by more than the noise.  Y6_AUTOTUNE_MODE=layer keeps the per-layer table.
    {
        const char* mode = getenv("Y6_AUTOTUNE_MODE");
        bool any = false;
        for (size_t i = 0; i < p->ops.size(); ++i) any = any || measured[i];
        // The pass replays EVERY op of the plan a few hundred times: only for plans made of pure ops.  A training graph's
        // BatchNorm statistics op moves running statistics, its weight-gradient ops accumulate: such plans keep
        // the per-layer table unless the caller asks (Y6_AUTOTUNE_MODE=step).
        bool stateful = false;
        for (const Op& o : p->ops)
            if (o.kind == Y6_OP_GENERIC) {
                const int t = o.gtag;
                if (!(t == Y6_TOP_CONV_I8 || t == Y6_TOP_ABSMAX || t == Y6_TOP_QUANT || t == Y6_TOP_PRED_DECODE || t == Y6_TOP_PW_S2 ||
                      t == Y6_TOP_STEM_S2))
                    stateful = true;
            }
        if (stateful && !(mode && strcmp(mode, "step") == 0)) any = false;
        if (any && !(mode && strcmp(mode, "layer") == 0)) {
            const size_t n = p->ops.size();
            hipEvent_t e0, e1;
            Y6_HIP(hipEventCreate(&e0));
            Y6_HIP(hipEventCreate(&e1));
            int rc_step = Y6_OK;
            auto step_ms = [&](int reps) -> float {   // median whole-step time (plan order, this stream)
                std::vector<float> t;
                for (int r = 0; r < reps && rc_step == Y6_OK; ++r) {
                    (void)hipEventRecord(e0, s);
                    for (size_t i = 0; i < n && rc_step == Y6_OK; ++i) rc_step = run_op(p->ops[i], s);
                    (void)hipEventRecord(e1, s);
                    if (hipEventSynchronize(e1) != hipSuccess) rc_step = Y6_EHIP;
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    t.push_back(ms);
                }
                if (t.empty()) return 1e30f;
                std::sort(t.begin(), t.end());
                return t[t.size() / 2];
            };
            std::vector<int> layer_tab(n, -1), shape_tab(n, -1);
            for (size_t i = 0; i < n; ++i) {
                if (!measured[i]) continue;
                layer_tab[i] = p->ops[i].conv.variant;
                const int dv = y6_conv_default_variant(&p->ops[i].conv);
                shape_tab[i] = (dv > 0 && !excluded[dv]) ? dv : layer_tab[i];
            }
            auto apply = [&](const std::vector<int>& tab) {
                for (size_t i = 0; i < n; ++i)
                    if (measured[i]) p->ops[i].conv.variant = tab[i];
            };
            (void)step_ms(2);   // warm
            apply(shape_tab);
            const float t_shape = step_ms(5);
            apply(layer_tab);
            const float t_layer = step_ms(5);
            std::vector<int> cur = t_shape < t_layer ? shape_tab : layer_tab;
            float best_t = t_shape < t_layer ? t_shape : t_layer;
            apply(cur);
            if (logf) fprintf(logf, "whole step: shape-derived table %.4f ms, per-layer table %.4f ms\n", t_shape, t_layer);
            // groups of equal signature
            std::vector<std::string> sigs(n);
            std::vector<std::vector<size_t>> groups;
            for (size_t i = 0; i < n; ++i) {
                if (!measured[i]) continue;
                sigs[i] = signature(p->ops[i].conv);
                bool found = false;
                for (auto& g : groups)
                    if (sigs[g[0]] == sigs[i]) {
                        g.push_back(i);
                        found = true;
                        break;
                    }
                if (!found) groups.push_back({i});
            }
            auto group_time = [&](const std::vector<size_t>& g) { return (double)times[g[0]][cur[g[0]]] * (double)g.size(); };
            std::sort(groups.begin(), groups.end(), [&](const auto& a, const auto& b) { return group_time(a) > group_time(b); });
            int trials = 0, kept = 0;
            for (const auto& g : groups) {
                const std::vector<float>& row = times[g[0]];
                float iso_best = 1e30f;
                for (int v = 1; v < nv; ++v)
                    if (row[v] < iso_best) iso_best = row[v];
                for (int v = 1; v < nv && rc_step == Y6_OK; ++v) {
                    if (v == cur[g[0]] || row[v] > 1e29f || excluded[v]) continue;
                    if (row[v] > iso_best * 1.3f && v != shape_tab[g[0]] && v != layer_tab[g[0]]) continue;
                    // the signature leaves out the channel offset, the pointer alignment and the packed-weight flag that
                    // y6_conv_variant_supports tests: every member of the group must be able to run the trial variant
                    bool all_ok = true;
                    for (size_t i : g) all_ok = all_ok && y6_conv_variant_supports(&p->ops[i].conv, v);
                    if (!all_ok) continue;
                    const int old = cur[g[0]];
                    for (size_t i : g) p->ops[i].conv.variant = v;
                    const float t = step_ms(5);
                    ++trials;
                    if (t < best_t * 0.996f) {
                        if (logf) fprintf(logf, "whole step: %s x%zu %s -> %s: %.4f -> %.4f ms\n", sigs[g[0]].c_str(), g.size(), y6_conv_variant_name(old), y6_conv_variant_name(v), best_t, t);
                        best_t = t;
                        for (size_t i : g) cur[i] = v;
                        ++kept;
                    } else {
                        for (size_t i : g) p->ops[i].conv.variant = old;
                    }
                }
            }
            if (logf) fprintf(logf, "whole step: %d trials, %d kept, %.4f ms\n", trials, kept, best_t);
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            if (rc_step) {
                apply(cur);        // never leave a trial variant behind
                if (logf) fclose(logf);
                return rc_step;
            }
        }
    }
    if (cachepath) {
        for (size_t i = 0; i < p->ops.size(); ++i) {
            if (!measured[i]) continue;
            const std::string sig = signature(p->ops[i].conv);
            bool have = false;
            for (const auto& kv : cache) have = have || kv.first == sig;
            if (have) continue;
            cache.emplace_back(sig, y6_conv_variant_name(p->ops[i].conv.variant));
            if (FILE* cf = fopen(cachepath, "a")) {
                fprintf(cf, "%s %s\n", sig.c_str(), y6_conv_variant_name(p->ops[i].conv.variant));
                fclose(cf);
            }
        }
    }
    if (logf) fclose(logf);
    Y6_HIP(hipStreamSynchronize(s));
    return Y6_OK;
}

// The conv kernel choices of `src` for `dst` - two plans lowered from the same module for the same shapes (HipModule.new_plan: the
// in-flight slots of pipeline.InflightRunner / bench.py) run the same kernels without tuning twice.
extern "C" int y6_plan_copy_variants(y6_plan* dst, const y6_plan* src) {
    Y6_REQUIRE(dst && src && dst->ops.size() == src->ops.size(), "plan_copy_variants: the plans differ in length");
    for (size_t i = 0; i < dst->ops.size(); ++i) {
        const Op& a = src->ops[i];
        Op& b = dst->ops[i];
        Y6_REQUIRE(a.kind == b.kind, "plan_copy_variants: op %zu differs in kind", i);
        if (a.kind != Y6_OP_CONV) continue;
        Y6_REQUIRE(a.conv.ksize == b.conv.ksize && a.conv.stride == b.conv.stride && a.conv.in.C == b.conv.in.C && a.conv.out.C == b.conv.out.C &&
                       a.conv.in.B == b.conv.in.B && a.conv.in.H == b.conv.in.H && a.conv.in.W == b.conv.in.W,
                   "plan_copy_variants: op %zu differs in shape", i);
        if (a.conv.variant >= 0) Y6_REQUIRE(y6_conv_variant_supports(&b.conv, a.conv.variant), "plan_copy_variants: op %zu cannot run variant %d", i, a.conv.variant);
        b.conv.variant = a.conv.variant;
    }
    drop_graph(dst);
    return Y6_OK;
}

extern "C" int y6_plan_profile(y6_plan* p, void* stream, int iters, float* ms, int32_t* kind, int32_t* variant,
                               double* flops, double* bytes, int cap) {
    if (!p) return Y6_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int n = (int)p->ops.size();
    for (int i = 0; i < n && i < cap; ++i) {
        const Op& op = p->ops[i];
        float t = 0.f;
        int rc = time_op(op, s, iters, &t);
        if (rc) return rc;
        if (ms) ms[i] = t;
        if (kind) kind[i] = op.kind;
        if (variant) variant[i] = op.kind == Y6_OP_CONV ? op.conv.variant : -1;
        double f = 0.0, by = 0.0;
        op_cost(op, &f, &by);
        if (flops) flops[i] = f;
        if (bytes) bytes[i] = by;
    }
    return n;
}
