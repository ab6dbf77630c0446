// plan.hip — native executor: an ordered list of hot-path ops with fixed device pointers,
// replayed once per forward (eagerly or from a captured hipGraph), with per-conv kernel
// autotuning and a per-op hipEvent profile.  This is what sits behind Model.forward
// (reference yolov6/models/yolo.py:33-41) instead of ~200 aten dispatches.
#include <string>
#include <vector>

#include "common.hpp"

static thread_local char g_err[512] = "";

void y6_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* y6_last_error(void) { return g_err; }
extern "C" int y6_abi_version(void) { return Y6_ABI_VERSION; }

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
};
}  // namespace

struct y6_plan {
    std::vector<Op> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
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
    }
    y6_set_error("plan: unknown op kind %d", op.kind);
    return Y6_EINVAL;
}

static void drop_graph(y6_plan* p) {
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    p->exec = nullptr;
    p->graph = nullptr;
}

extern "C" y6_plan* y6_plan_create(void) { return new y6_plan(); }
extern "C" void y6_plan_destroy(y6_plan* p) {
    if (!p) return;
    drop_graph(p);
    delete p;
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

extern "C" int y6_plan_rebind(y6_plan* p, const void* old_ptr, const void* new_ptr) {
    // Point every op that reads the caller's boundary tensor `old_ptr` at `new_ptr` (same shape
    // and dtype).  Drops a captured graph: its kernel nodes baked the old address.
    Y6_REQUIRE(p && old_ptr && new_ptr, "plan_rebind: null argument");
    int n = 0;
    for (Op& op : p->ops) {
        if (op.kind == Y6_OP_STEM && op.stem.in_nchw == old_ptr) {
            op.stem.in_nchw = new_ptr;
            ++n;
        }
        if (op.kind == Y6_OP_NCHW2NHWC && op.src == old_ptr) {
            op.src = new_ptr;
            ++n;
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
    for (size_t i = 0; i < p->ops.size(); ++i) {
        int rc = run_op(p->ops[i], s);
        if (rc) return rc;
    }
    return Y6_OK;
}

extern "C" int y6_plan_capture(y6_plan* p, void* stream) {
    Y6_REQUIRE(p, "plan_capture: null plan");
    hipStream_t s = (hipStream_t)stream;
    drop_graph(p);
    Y6_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = Y6_OK;
    for (size_t i = 0; i < p->ops.size() && rc == Y6_OK; ++i) rc = run_op(p->ops[i], s);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess) {
        y6_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return Y6_EHIP;
    }
    p->graph = g;
    Y6_HIP(hipGraphInstantiate(&p->exec, g, nullptr, nullptr, 0));
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

extern "C" int y6_plan_autotune(y6_plan* p, void* stream, int iters) {
    Y6_REQUIRE(p, "plan_autotune: null plan");
    hipStream_t s = (hipStream_t)stream;
    drop_graph(p);
    if (iters < 1) iters = 3;
    const int nv = y6_conv_variants();
    for (size_t i = 0; i < p->ops.size(); ++i) {
        Op& op = p->ops[i];
        if (op.kind != Y6_OP_CONV) continue;
        int best = -1;
        float best_ms = 1e30f;
        for (int v = 1; v < nv; ++v) {  // variant 0 (naive) is a cross-check, never a candidate
            if (!y6_conv_variant_supports(&op.conv, v)) continue;
            Op trial = op;
            trial.conv.variant = v;
            float ms = 0.f;
            int rc = time_op(trial, s, iters, &ms);
            if (rc) return rc;
            if (ms < best_ms) {
                best_ms = ms;
                best = v;
            }
        }
        if (best < 0) best = y6_conv_variant_supports(&op.conv, 0) ? 0 : -1;
        Y6_REQUIRE(best >= 0, "plan_autotune: op %zu has no runnable conv variant", i);
        op.conv.variant = best;
    }
    Y6_HIP(hipStreamSynchronize(s));
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
        } else if (op.kind == Y6_OP_DECODE) {
            double A = 0;
            for (int l = 0; l < op.dec.n_levels; ++l) A += (double)op.dec.cls[l].H * op.dec.cls[l].W;
            const double B = op.dec.cls[0].B;
            by = B * A * ((op.dec.nc + 5) * 4.0 + (op.dec.nc + op.dec.reg[0].C) * 2.0);
        }
        if (flops) flops[i] = f;
        if (bytes) bytes[i] = by;
    }
    return n;
}
