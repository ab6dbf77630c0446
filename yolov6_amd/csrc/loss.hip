// loss.hip — forward value of the training loss.
// Restates (reference yolov6/models/losses/loss.py)
//   bbox_decode      :194-198  softmax over the reg_max+1 DFL bins . linspace(0, reg_max), dist2bbox xyxy (general.py:32-43)
//   VarifocalLoss    :201-211  BCE(p, q) * (0.75 p^2 (1-y) + q y), fp32, logs clamped at -100 (torch BCE)
//   BboxLoss         :214-278  IOUloss (utils/figure_iou.py:7-100, xyxy, eps 1e-10) * sum_c target_scores, DFL two-bin
//                              cross entropy (:267-278); both / target_scores_sum when that is > 1 (:168-169, :238-261)
//   weights          :171-181  loss = w_class*cls + w_iou*iou + w_dfl*dfl; items = (w_iou*iou, w_dfl*dfl, w_class*cls)
// The label assignment in between is y6_tal_assign / y6_atss_assign (tal.hip).  y6_loss_forward_backward adds the gradients
// wrt pred_scores / pred_distri (dual-number IoU, closed forms for the rest).
// Compile with -ffp-contract=off: every elementwise term follows the reference's unfused fp32 arithmetic; only the
// final sums differ (block partials in double, one double atomic per block - order independent to ~1e-12).
#include "common.hpp"

namespace {

constexpr float kPi = 3.14159265358979323846f;

__global__ __launch_bounds__(256) void bbox_decode_kernel(const float* __restrict__ dist, const float* __restrict__ pts,
                                                          int B, int A, int use_dfl, int reg_max, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * A) return;
    const int a = (int)(i % A);
    float d[4];
    if (use_dfl) {
        const int nb = reg_max + 1;
        const float* p = dist + i * 4 * nb;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* q = p + s * nb;
            float m = -INFINITY;
            for (int k = 0; k < nb; ++k) m = fmaxf(m, q[k]);
            float den = 0.f;
            for (int k = 0; k < nb; ++k) den += expf(q[k] - m);
            float v = 0.f;
            for (int k = 0; k < nb; ++k) v += (expf(q[k] - m) / den) * (float)k;   // softmax, then . proj
            d[s] = v;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) d[s] = dist[i * 4 + s];
    }
    const float ax = pts[a * 2], ay = pts[a * 2 + 1];
    float4 o;
    o.x = ax - d[0];
    o.y = ay - d[1];
    o.z = ax + d[2];
    o.w = ay + d[3];
    reinterpret_cast<float4*>(out)[i] = o;
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    __syncthreads();
    return t;   // valid in thread 0
}

// acc[0] = sum VFL terms, acc[1] = sum target_scores
__global__ __launch_bounds__(256) void loss_cls_kernel(const float* __restrict__ pred, const float* __restrict__ tscore,
                                                       const int64_t* __restrict__ tlabel, const uint8_t* __restrict__ fg,
                                                       size_t n_ba, int C, double* __restrict__ acc) {
    __shared__ double sh[4];
    double s_cls = 0.0, s_ts = 0.0;
    const size_t total = n_ba * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t ba = i / C;
        const int c = (int)(i - ba * C);
        const float p = pred[i], q = tscore[i];
        const float y = (fg[ba] && tlabel[ba] == (int64_t)c) ? 1.f : 0.f;
        const float weight = 0.75f * (p * p) * (1.f - y) + q * y;
        const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
        const float bce = -(q * lp + (1.f - q) * l1p);
        s_cls += (double)(bce * weight);
        s_ts += (double)q;
    }
    const double a = block_sum(s_cls, sh);
    const double b = block_sum(s_ts, sh);
    if (threadIdx.x == 0) {
        atomicAdd(&acc[0], a);
        atomicAdd(&acc[1], b);
    }
}

// Four classes per thread (C % 4 == 0, fewer than 2^32 scores, 16-byte aligned tensors): one 16-byte load per tensor, one 32-bit
// division per four elements (the per-element form divides a 64-bit index by C 43 M times per b64 step), and log(p) only where
// the target score is not zero - `q * lp` is an exact (signed) zero for q == 0 because lp is clamped to [-100, 0], so every term
// has the value the per-element kernel gives it.
__global__ __launch_bounds__(256) void loss_cls4_kernel(const float4* __restrict__ pred, const float4* __restrict__ tscore,
                                                        const int64_t* __restrict__ tlabel, const uint8_t* __restrict__ fg,
                                                        unsigned total4, unsigned C, double* __restrict__ acc) {
    __shared__ double sh[4];
    double s_cls = 0.0, s_ts = 0.0;
    for (unsigned v = blockIdx.x * blockDim.x + threadIdx.x; v < total4; v += gridDim.x * blockDim.x) {
        const float4 p4 = pred[v], q4 = tscore[v];
        const unsigned i0 = v * 4u, ba = i0 / C, c0 = i0 - ba * C;
        const int lab = fg[ba] ? (int)tlabel[ba] : -1;
        const float pp[4] = {p4.x, p4.y, p4.z, p4.w}, qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float p = pp[j], q = qq[j];
            const float y = (lab == (int)(c0 + j)) ? 1.f : 0.f;
            const float weight = 0.75f * (p * p) * (1.f - y) + q * y;
            const float lp = q != 0.f ? fmaxf(logf(p), -100.f) : -1.f, l1p = fmaxf(logf(1.f - p), -100.f);
            const float bce = -(q * lp + (1.f - q) * l1p);
            s_cls += (double)(bce * weight);
            s_ts += (double)q;
        }
    }
    const double a = block_sum(s_cls, sh);
    const double b = block_sum(s_ts, sh);
    if (threadIdx.x == 0) {
        atomicAdd(&acc[0], a);
        atomicAdd(&acc[1], b);
    }
}

__device__ __forceinline__ float iou_loss_xyxy(const float4 b1, const float4 b2, int type) {
    const float e = 1e-10f;
    const float iw = fmaxf(fminf(b1.z, b2.z) - fmaxf(b1.x, b2.x), 0.f);
    const float ih = fmaxf(fminf(b1.w, b2.w) - fmaxf(b1.y, b2.y), 0.f);
    const float inter = iw * ih;
    const float w1 = b1.z - b1.x, h1 = b1.w - b1.y + e;
    const float w2 = b2.z - b2.x, h2 = b2.w - b2.y + e;
    const float uni = w1 * h1 + w2 * h2 - inter + e;
    float iou = inter / uni;
    const float cw = fmaxf(b1.z, b2.z) - fminf(b1.x, b2.x);
    const float ch = fmaxf(b1.w, b2.w) - fminf(b1.y, b2.y);
    if (type == Y6_IOU_GIOU) {
        const float c_area = cw * ch + e;
        iou = iou - (c_area - uni) / c_area;
    } else if (type == Y6_IOU_DIOU || type == Y6_IOU_CIOU) {
        const float c2 = cw * cw + ch * ch + e;
        const float dx = b2.x + b2.z - b1.x - b1.z, dy = b2.y + b2.w - b1.y - b1.w;
        const float rho2 = (dx * dx + dy * dy) / 4.f;
        if (type == Y6_IOU_DIOU) {
            iou = iou - rho2 / c2;
        } else {
            const float t = atanf(w2 / h2) - atanf(w1 / h1);
            const float v = (4.f / (kPi * kPi)) * (t * t);
            const float alpha = v / (v - iou + (1.f + e));
            iou = iou - (rho2 / c2 + v * alpha);
        }
    } else if (type == Y6_IOU_SIOU) {
        const float s_cw = (b2.x + b2.z - b1.x - b1.z) * 0.5f + e;
        const float s_ch = (b2.y + b2.w - b1.y - b1.w) * 0.5f + e;
        const float sigma = sqrtf(s_cw * s_cw + s_ch * s_ch);
        const float sa1 = fabsf(s_cw) / sigma, sa2 = fabsf(s_ch) / sigma;
        const float thr = 0.70710678118654752f;
        const float sa = sa1 > thr ? sa2 : sa1;
        const float angle_cost = cosf(asinf(sa) * 2.f - kPi / 2.f);
        const float rx = (s_cw / cw) * (s_cw / cw), ry = (s_ch / ch) * (s_ch / ch);
        const float gamma = angle_cost - 2.f;
        const float distance_cost = 2.f - expf(gamma * rx) - expf(gamma * ry);
        const float ow = fabsf(w1 - w2) / fmaxf(w1, w2), oh = fabsf(h1 - h2) / fmaxf(h1, h2);
        const float tw = 1.f - expf(-ow), th = 1.f - expf(-oh);
        const float shape_cost = (tw * tw) * (tw * tw) + (th * th) * (th * th);
        iou = iou - 0.5f * (distance_cost + shape_cost);
    }
    return 1.f - iou;
}

// one thread per (b, a); foreground anchors contribute.  acc[2] = sum iou_loss * w, acc[3] = sum dfl * w, acc[4] = num_pos
__global__ __launch_bounds__(256) void loss_box_kernel(const float* __restrict__ pred_distri,
                                                       const float* __restrict__ pred_bboxes,
                                                       const float* __restrict__ pts, const float* __restrict__ stride,
                                                       const float* __restrict__ tboxes, const float* __restrict__ tscore,
                                                       const uint8_t* __restrict__ fg, int B, int A, int C, int use_dfl,
                                                       int reg_max, int iou_type, double* __restrict__ acc) {
    __shared__ double sh[4];
    double s_iou = 0.0, s_dfl = 0.0, s_n = 0.0;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)B * A && fg[i]) {
        const int a = (int)(i % A);
        float w = 0.f;
        for (int c = 0; c < C; ++c) w += tscore[i * C + c];     // target_scores.sum(-1)  (:232-233)
        const float st = stride[a];
        const float4 tb4 = reinterpret_cast<const float4*>(tboxes)[i];
        const float4 tb = make_float4(tb4.x / st, tb4.y / st, tb4.z / st, tb4.w / st);   // target_bboxes /= stride (:154)
        const float4 pb = reinterpret_cast<const float4*>(pred_bboxes)[i];
        s_iou = (double)(iou_loss_xyxy(pb, tb, iou_type) * w);
        s_n = 1.0;
        if (use_dfl) {
            const int nb = reg_max + 1;
            const float ax = pts[a * 2], ay = pts[a * 2 + 1];
            const float hi = (float)reg_max - 0.01f;
            float t[4] = {ax - tb.x, ay - tb.y, tb.z - ax, tb.w - ay};   // bbox2dist (general.py:45-49)
            float l4 = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float tv = fminf(fmaxf(t[s], 0.f), hi);
                const int tl = (int)tv, tr = tl + 1;
                const float wl = (float)tr - tv, wr = 1.f - wl;
                const float* q = pred_distri + (i * 4 + s) * nb;
                float m = -INFINITY;
                for (int k = 0; k < nb; ++k) m = fmaxf(m, q[k]);
                float den = 0.f;
                for (int k = 0; k < nb; ++k) den += expf(q[k] - m);
                const float lse = logf(den) + m;
                l4 += (lse - q[tl]) * wl + (lse - q[tr]) * wr;
            }
            s_dfl = (double)((l4 / 4.f) * w);    // .mean(-1) over the four sides (:278)
        }
    }
    const double a0 = block_sum(s_iou, sh);
    const double a1 = block_sum(s_dfl, sh);
    const double a2 = block_sum(s_n, sh);
    if (threadIdx.x == 0 && a2 > 0.0) {
        atomicAdd(&acc[2], a0);
        atomicAdd(&acc[3], a1);
        atomicAdd(&acc[4], a2);
    }
}

// Normalisation by target_scores_sum: loss.py divides when the sum is > 1 (:168-169, :238-261); the self-distillation losses
// divide the class term when it is > 0 (loss_distill.py:178-183) and the box terms unless it is exactly 0 (:283-330).
__device__ __forceinline__ bool norm_cls(double ts, int mode) { return mode ? ts > 0.0 : ts > 1.0; }
__device__ __forceinline__ bool norm_box(double ts, int mode) { return mode ? ts != 0.0 : ts > 1.0; }

__global__ void loss_finalize_kernel(const double* __restrict__ acc, float w_class, float w_iou, float w_dfl, int use_dfl,
                                     double* __restrict__ out, int norm_mode) {
    double cls = acc[0], iou = 0.0, dfl = 0.0;
    const double ts = acc[1], npos = acc[4];
    if (norm_cls(ts, norm_mode)) cls /= ts;
    if (npos > 0.0) {                             // :227
        iou = acc[2];
        dfl = use_dfl ? acc[3] : 0.0;
        if (norm_box(ts, norm_mode)) {
            iou /= ts;
            dfl /= ts;
        }
    }
    out[0] = (double)w_class * cls + (double)w_iou * iou + (double)w_dfl * dfl;
    out[1] = (double)w_iou * iou;
    out[2] = (double)w_dfl * dfl;
    out[3] = (double)w_class * cls;
    out[4] = ts;
    out[5] = npos;
}

// ------------------------------------------------------------------ gradient (training step)
// Forward-mode dual numbers over the four coordinates of the predicted box: ONE statement of the IoU family gives the
// exact derivative autograd would produce for every iou_type.  Conventions at non-smooth points follow torch:
// maximum/minimum split the gradient at ties, clamp passes it at the bound, abs'(0) = 0.
struct D4 {
    float v;
    float d[4];
};
__device__ __forceinline__ D4 dconst(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 dvar(float c, int i) {
    D4 r = dconst(c);
    r.d[i] = 1.f;
    return r;
}
#define D4_EACH(expr)              \
    D4 r;                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) r.d[i] = (expr);
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b) { D4_EACH(a.d[i] + b.d[i]) r.v = a.v + b.v; return r; }
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b) { D4_EACH(a.d[i] - b.d[i]) r.v = a.v - b.v; return r; }
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) { D4_EACH(a.d[i] * b.v + a.v * b.d[i]) r.v = a.v * b.v; return r; }
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) {
    const float q = a.v / b.v;
    D4_EACH((a.d[i] - q * b.d[i]) / b.v) r.v = q; return r;
}
__device__ __forceinline__ D4 operator+(const D4& a, float c) { D4 r = a; r.v += c; return r; }
__device__ __forceinline__ D4 operator-(const D4& a, float c) { D4 r = a; r.v -= c; return r; }
__device__ __forceinline__ D4 operator-(float c, const D4& a) { D4_EACH(-a.d[i]) r.v = c - a.v; return r; }
__device__ __forceinline__ D4 operator*(const D4& a, float c) { D4_EACH(a.d[i] * c) r.v = a.v * c; return r; }
__device__ __forceinline__ D4 dmax(const D4& a, const D4& b) {
    if (a.v > b.v) return a;
    if (b.v > a.v) return b;
    D4_EACH(0.5f * (a.d[i] + b.d[i])) r.v = a.v; return r;
}
__device__ __forceinline__ D4 dmin(const D4& a, const D4& b) {
    if (a.v < b.v) return a;
    if (b.v < a.v) return b;
    D4_EACH(0.5f * (a.d[i] + b.d[i])) r.v = a.v; return r;
}
__device__ __forceinline__ D4 dclamp0(const D4& a) { return a.v >= 0.f ? a : dconst(0.f); }
__device__ __forceinline__ D4 dabs(const D4& a) {
    const float s = a.v > 0.f ? 1.f : (a.v < 0.f ? -1.f : 0.f);
    D4_EACH(s * a.d[i]) r.v = fabsf(a.v); return r;
}
__device__ __forceinline__ D4 dfun(const D4& a, float value, float slope) { D4_EACH(slope * a.d[i]) r.v = value; return r; }
__device__ __forceinline__ D4 dsqrt(const D4& a) { const float s = sqrtf(a.v); return dfun(a, s, 0.5f / s); }
__device__ __forceinline__ D4 dexp(const D4& a) { const float e = expf(a.v); return dfun(a, e, e); }
__device__ __forceinline__ D4 datan(const D4& a) { return dfun(a, atanf(a.v), 1.f / (1.f + a.v * a.v)); }
__device__ __forceinline__ D4 dasin(const D4& a) { return dfun(a, asinf(a.v), 1.f / sqrtf(1.f - a.v * a.v)); }
__device__ __forceinline__ D4 dcos(const D4& a) { return dfun(a, cosf(a.v), -sinf(a.v)); }
#undef D4_EACH

// utils/figure_iou.py:53-95 on dual numbers; b1 = prediction (variables), b2 = target (constants).  Returns 1 - iou.
__device__ D4 iou_loss_dual(const float4 p, const float4 t, int type) {
    const float e = 1e-10f;
    const D4 x1 = dvar(p.x, 0), y1 = dvar(p.y, 1), x2 = dvar(p.z, 2), y2 = dvar(p.w, 3);
    const D4 tx1 = dconst(t.x), ty1 = dconst(t.y), tx2 = dconst(t.z), ty2 = dconst(t.w);
    const D4 inter = dclamp0(dmin(x2, tx2) - dmax(x1, tx1)) * dclamp0(dmin(y2, ty2) - dmax(y1, ty1));
    const D4 w1 = x2 - x1, h1 = y2 - y1 + e;
    const float w2 = t.z - t.x, h2 = t.w - t.y + e;
    const D4 uni = w1 * h1 + (w2 * h2) - inter + e;
    D4 iou = inter / uni;
    const D4 cw = dmax(x2, tx2) - dmin(x1, tx1), ch = dmax(y2, ty2) - dmin(y1, ty1);
    if (type == Y6_IOU_GIOU) {
        const D4 c_area = cw * ch + e;
        iou = iou - (c_area - uni) / c_area;
    } else if (type == Y6_IOU_DIOU || type == Y6_IOU_CIOU) {
        const D4 c2 = cw * cw + ch * ch + e;
        const D4 dx = (t.x + t.z) - x1 - x2, dy = (t.y + t.w) - y1 - y2;
        const D4 rho2 = (dx * dx + dy * dy) * 0.25f;
        if (type == Y6_IOU_DIOU) {
            iou = iou - rho2 / c2;
        } else {
            const D4 dt = atanf(w2 / h2) - datan(w1 / h1);
            const D4 v = (dt * dt) * (4.f / (kPi * kPi));
            const float alpha = v.v / (v.v - iou.v + (1.f + e));      // torch.no_grad() in the reference (:76-77)
            iou = iou - (rho2 / c2 + v * alpha);
        }
    } else if (type == Y6_IOU_SIOU) {
        const D4 s_cw = ((t.x + t.z) - x1 - x2) * 0.5f + e;
        const D4 s_ch = ((t.y + t.w) - y1 - y2) * 0.5f + e;
        const D4 sigma = dsqrt(s_cw * s_cw + s_ch * s_ch);
        const D4 sa1 = dabs(s_cw) / sigma, sa2 = dabs(s_ch) / sigma;
        const D4 sa = sa1.v > 0.70710678118654752f ? sa2 : sa1;
        const D4 angle_cost = dcos(dasin(sa) * 2.f - (kPi / 2.f));
        const D4 rx = (s_cw / cw) * (s_cw / cw), ry = (s_ch / ch) * (s_ch / ch);
        const D4 gamma = angle_cost - 2.f;
        const D4 distance_cost = 2.f - dexp(gamma * rx) - dexp(gamma * ry);
        const D4 ow = dabs(w1 - w2) / dmax(w1, dconst(w2)), oh = dabs(h1 - h2) / dmax(h1, dconst(h2));
        const D4 tw = 1.f - dexp(dconst(0.f) - ow), th = 1.f - dexp(dconst(0.f) - oh);
        const D4 shape_cost = (tw * tw) * (tw * tw) + (th * th) * (th * th);
        iou = iou - (distance_cost + shape_cost) * 0.5f;
    }
    return 1.f - iou;
}

// d loss / d pred_scores: VarifocalLoss with its prediction-dependent weight (loss.py:203-209); BCE's gradient as torch
// writes it, (p - q) / max((1-p) p, 1e-12).  fin = y6_loss_forward's out[] (fin[4] = target_scores_sum).
__global__ __launch_bounds__(256) void loss_cls_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tscore,
                                                           const int64_t* __restrict__ tlabel, const uint8_t* __restrict__ fg,
                                                           size_t n_ba, int C, const double* __restrict__ fin, float w_class,
                                                           const float* __restrict__ grad_scale, float* __restrict__ dpred, int norm_mode) {
    const double ts = fin[4];
    const float coef = w_class * (grad_scale ? *grad_scale : 1.f) * (norm_cls(ts, norm_mode) ? (float)(1.0 / ts) : 1.f);
    const size_t total = n_ba * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t ba = i / C;
        const int c = (int)(i - ba * C);
        const float p = pred[i], q = tscore[i];
        const float y = (fg[ba] && tlabel[ba] == (int64_t)c) ? 1.f : 0.f;
        const float weight = 0.75f * (p * p) * (1.f - y) + q * y;
        const float dweight = 0.75f * 2.f * p * (1.f - y);
        const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
        const float bce = -(q * lp + (1.f - q) * l1p);
        const float dbce = (p - q) / fmaxf((1.f - p) * p, 1e-12f);
        dpred[i] = coef * (bce * dweight + weight * dbce);
    }
}

// (four classes per thread: see loss_cls4_kernel)
__global__ __launch_bounds__(256) void loss_cls_bwd4_kernel(const float4* __restrict__ pred, const float4* __restrict__ tscore,
                                                            const int64_t* __restrict__ tlabel, const uint8_t* __restrict__ fg,
                                                            unsigned total4, unsigned C, const double* __restrict__ fin, float w_class,
                                                            const float* __restrict__ grad_scale, float4* __restrict__ dpred, int norm_mode) {
    const double ts = fin[4];
    const float coef = w_class * (grad_scale ? *grad_scale : 1.f) * (norm_cls(ts, norm_mode) ? (float)(1.0 / ts) : 1.f);
    for (unsigned v = blockIdx.x * blockDim.x + threadIdx.x; v < total4; v += gridDim.x * blockDim.x) {
        const float4 p4 = pred[v], q4 = tscore[v];
        const unsigned i0 = v * 4u, ba = i0 / C, c0 = i0 - ba * C;
        const int lab = fg[ba] ? (int)tlabel[ba] : -1;
        const float pp[4] = {p4.x, p4.y, p4.z, p4.w}, qq[4] = {q4.x, q4.y, q4.z, q4.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float p = pp[j], q = qq[j];
            const float y = (lab == (int)(c0 + j)) ? 1.f : 0.f;
            const float weight = 0.75f * (p * p) * (1.f - y) + q * y;
            const float dweight = 0.75f * 2.f * p * (1.f - y);
            const float lp = q != 0.f ? fmaxf(logf(p), -100.f) : -1.f, l1p = fmaxf(logf(1.f - p), -100.f);
            const float bce = -(q * lp + (1.f - q) * l1p);
            const float dbce = (p - q) / fmaxf((1.f - p) * p, 1e-12f);
            o[j] = coef * (bce * dweight + weight * dbce);
        }
        dpred[v] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// d loss / d pred_distri through IoU loss (dual numbers) -> dist2bbox -> DFL projection, plus the DFL cross entropies
__global__ __launch_bounds__(256) void loss_box_bwd_kernel(const float* __restrict__ pred_distri, const float* __restrict__ pred_bboxes,
                                                           const float* __restrict__ pts, const float* __restrict__ stride,
                                                           const float* __restrict__ tboxes, const float* __restrict__ tscore,
                                                           const uint8_t* __restrict__ fg, int B, int A, int C, int use_dfl, int reg_max,
                                                           int iou_type, const double* __restrict__ fin, float w_iou, float w_dfl,
                                                           const float* __restrict__ grad_scale, float* __restrict__ ddistri, int box_mode,
                                                           int norm_mode) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * A) return;
    const int nb = use_dfl ? reg_max + 1 : 1;
    float* out = ddistri + i * 4 * nb;
    if (!fg[i]) {
        for (int k = 0; k < 4 * nb; ++k) out[k] = 0.f;
        return;
    }
    const double ts = fin[4];
    const float norm = (grad_scale ? *grad_scale : 1.f) * (norm_box(ts, norm_mode) ? (float)(1.0 / ts) : 1.f);
    const int a = (int)(i % A);
    float w = 0.f;
    for (int c = 0; c < C; ++c) w += tscore[i * C + c];
    const float st = stride[a];
    const float4 tb4 = reinterpret_cast<const float4*>(tboxes)[i];
    const float4 tb = make_float4(tb4.x / st, tb4.y / st, tb4.z / st, tb4.w / st);
    const float4 pb = reinterpret_cast<const float4*>(pred_bboxes)[i];
    const D4 l = iou_loss_dual(pb, tb, iou_type);
    const float ci = w * w_iou * norm;
    // dist2bbox (general.py:32-43): x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3
    if (box_mode == 1) {   // (cx, cy, w, h): x1 = cx - w/2, x2 = x1 + w (general.py:52-58)
        out[0] = ci * (l.d[0] + l.d[2]);
        out[1] = ci * (l.d[1] + l.d[3]);
        out[2] = ci * 0.5f * (l.d[2] - l.d[0]);
        out[3] = ci * 0.5f * (l.d[3] - l.d[1]);
        return;
    }
    const float dd[4] = {-ci * l.d[0], -ci * l.d[1], ci * l.d[2], ci * l.d[3]};
    if (!use_dfl) {
#pragma unroll
        for (int s = 0; s < 4; ++s) out[s] = dd[s];
        return;
    }
    const float ax = pts[a * 2], ay = pts[a * 2 + 1];
    const float hi = (float)reg_max - 0.01f;
    const float t[4] = {ax - tb.x, ay - tb.y, tb.z - ax, tb.w - ay};
    const float cd = w * w_dfl * norm * 0.25f;          // .mean(-1) over the four sides (:278)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float* q = pred_distri + (i * 4 + s) * nb;
        float m = -INFINITY;
        for (int k = 0; k < nb; ++k) m = fmaxf(m, q[k]);
        float den = 0.f;
        for (int k = 0; k < nb; ++k) den += expf(q[k] - m);
        float ex = 0.f;
        for (int k = 0; k < nb; ++k) ex += (expf(q[k] - m) / den) * (float)k;
        const float tv = fminf(fmaxf(t[s], 0.f), hi);
        const int tl = (int)tv, tr = tl + 1;
        const float wl = (float)tr - tv, wr = 1.f - wl;
        for (int k = 0; k < nb; ++k) {
            const float pk = expf(q[k] - m) / den;
            float g = dd[s] * pk * ((float)k - ex);                               // softmax . proj
            g += cd * (pk - (k == tl ? wl : 0.f) - (k == tr ? wr : 0.f));         // two-bin cross entropy
            out[s * nb + k] = g;
        }
    }
}


// ------------------------------------------------------------------ self-distillation terms
// Restates (reference yolov6/models/losses/loss_distill.py; the N / S variant loss_distill_ns.py has the same two functions)
//   distill_loss_cls  :210-221  KL(softmax(teacher / T) || softmax(student / T)) over the class axis, summed over ALL anchors
//                               (the "logits" are the post-sigmoid class scores of the two heads, as the reference passes them)
//   distill_loss_dfl  :349-359  on the positive anchors: the same KL over the reg_max + 1 bins of each of the 4 sides
// as sums (acc[0], acc[1]); the caller applies T^2, the mean over (positives x 4), the positives' weights, target_scores_sum and
// the loss weights (all scalars: yolov6_amd/models/losses/loss_distill.py).  acc[2] = sum of the positives' weights
// (target_scores.sum(-1)), acc[3] = number of positives.  One thread per anchor; fp32 terms, double sums.
struct DistillArgs {
    const float* ps;
    const float* pt;
    const float* ds;
    const float* dt;
    const uint8_t* fg;
    const float* tscore;
    size_t n_ba;
    int C, nb;
    float invT;
    double* acc;
    const float* coef;
    float* dscores;
    float* ddistri;
};

// KL(softmax(t * invT) || softmax(s * invT)) over n elements (stride 1); optionally d KL / d s_j * coef added to g[j]
__device__ __forceinline__ float kl_softened(const float* __restrict__ s, const float* __restrict__ t, int n, float invT, float coef,
                                             float* __restrict__ g) {
    float ms = -INFINITY, mt = -INFINITY;
    for (int k = 0; k < n; ++k) {
        ms = fmaxf(ms, s[k] * invT);
        mt = fmaxf(mt, t[k] * invT);
    }
    float zs = 0.f, zt = 0.f;
    for (int k = 0; k < n; ++k) {
        zs += expf(s[k] * invT - ms);
        zt += expf(t[k] * invT - mt);
    }
    float kl = 0.f;
    for (int k = 0; k < n; ++k) {
        const float p_s = expf(s[k] * invT - ms) / zs, p_t = expf(t[k] * invT - mt) / zt;
        if (p_t > 0.f) kl += p_t * (logf(p_t) - logf(p_s));          // F.kl_div(log p_s, p_t): xlogy(t, t) - t log s
        if (g) g[k] += coef * (p_s - p_t) * invT;                    // d/ds_j of -sum p_t log softmax(s / T)_j
    }
    return kl;
}

template <bool BWD>
__global__ __launch_bounds__(256) void distill_kernel(const DistillArgs a) {
    __shared__ double sh[4];
    double kc = 0.0, kd = 0.0, wsum = 0.0, npos = 0.0;
    const float c0 = BWD ? a.coef[0] : 0.f, c1 = BWD ? a.coef[1] : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_ba; i += (size_t)gridDim.x * blockDim.x) {
        const float k0 = kl_softened(a.ps + i * a.C, a.pt + i * a.C, a.C, a.invT, c0, BWD ? a.dscores + i * a.C : nullptr);
        kc += (double)k0;
        if (a.ds != nullptr && a.fg[i]) {
            float w = 0.f;
            for (int c = 0; c < a.C; ++c) w += a.tscore[i * a.C + c];
            wsum += (double)w;
            npos += 1.0;
            for (int side = 0; side < 4; ++side) {
                const size_t o = (i * 4 + side) * a.nb;
                kd += (double)kl_softened(a.ds + o, a.dt + o, a.nb, a.invT, c1, BWD ? a.ddistri + o : nullptr);
            }
        }
    }
    if (BWD) return;
    const double t0 = block_sum(kc, sh), t1 = block_sum(kd, sh), t2 = block_sum(wsum, sh), t3 = block_sum(npos, sh);
    if (threadIdx.x == 0) {
        atomicAdd(&a.acc[0], t0);
        if (t3 > 0.0) {
            atomicAdd(&a.acc[1], t1);
            atomicAdd(&a.acc[2], t2);
            atomicAdd(&a.acc[3], t3);
        }
    }
}

// Channel-wise feature distillation (loss_distill.py:222-246): per (image, channel) row of H*W values,
//   KL(softmax_hw(t / T) || softmax_hw(s / T))   summed into acc[0];   backward: ds[j] += coef * (softmax(s/T)_j - softmax(t/T)_j) / T.
// One block per row; rows are dense ([N, C, H*W] fp32).
template <bool BWD>
__global__ __launch_bounds__(256) void distill_cw_kernel(const float* __restrict__ sf, const float* __restrict__ tf, int hw, float invT,
                                                         double* __restrict__ acc, const float* __restrict__ coef, float* __restrict__ dsf) {
    __shared__ double sh[4];
    __shared__ float red[8];
    const size_t row = blockIdx.x;
    const float* s = sf + row * hw;
    const float* t = tf + row * hw;
    auto bmax = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        const float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        return r;
    };
    auto bsum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        const float r = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        return r;
    };
    float ms = -INFINITY, mt = -INFINITY;
    for (int k = threadIdx.x; k < hw; k += 256) {
        ms = fmaxf(ms, s[k] * invT);
        mt = fmaxf(mt, t[k] * invT);
    }
    ms = bmax(ms);
    mt = bmax(mt);
    float zs = 0.f, zt = 0.f;
    for (int k = threadIdx.x; k < hw; k += 256) {
        zs += expf(s[k] * invT - ms);
        zt += expf(t[k] * invT - mt);
    }
    zs = bsum(zs);
    zt = bsum(zt);
    const float lzs = logf(zs), lzt = logf(zt);
    const float c = BWD ? coef[0] : 0.f;
    double kl = 0.0;
    for (int k = threadIdx.x; k < hw; k += 256) {
        const float ls = s[k] * invT - ms - lzs, lt = t[k] * invT - mt - lzt;      // log_softmax of both (log_target = True)
        if (BWD)
            dsf[row * hw + k] = c * (expf(ls) - expf(lt)) * invT;
        else
            kl += (double)(expf(lt) * (lt - ls));
    }
    if (BWD) return;
    const double tot = block_sum(kl, sh);
    if (threadIdx.x == 0) atomicAdd(acc, tot);
}

}  // namespace

extern "C" int y6_bbox_decode(const float* pred_distri, const float* anchor_points_s, int B, int A, int use_dfl, int reg_max,
                              float* pred_bboxes, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(pred_distri && anchor_points_s && pred_bboxes && B > 0 && A > 0, "bbox_decode: bad arguments");
    Y6_REQUIRE(!use_dfl || (reg_max >= 1 && reg_max <= 63), "bbox_decode: reg_max %d out of range", reg_max);
    Y6_REQUIRE(((uintptr_t)pred_bboxes & 15) == 0, "bbox_decode: output must be 16-byte aligned");
    const size_t n = (size_t)B * A;
    hipLaunchKernelGGL(bbox_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred_distri,
                       anchor_points_s, B, A, use_dfl, reg_max, pred_bboxes);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

extern "C" size_t y6_loss_workspace_bytes(void) { return 8 * sizeof(double); }

extern "C" int y6_loss_forward(const y6_loss_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d && d->pred_scores && d->pred_distri && d->pred_bboxes && d->anchor_points_s && d->stride &&
                   d->target_labels && d->target_bboxes && d->target_scores && d->fg_mask && d->out && d->workspace,
               "loss_forward: null argument");
    Y6_REQUIRE(d->B > 0 && d->A > 0 && d->C > 0, "loss_forward: bad sizes");
    Y6_REQUIRE(d->iou_type >= Y6_IOU_GIOU && d->iou_type <= Y6_IOU_SIOU, "loss_forward: unknown iou_type %d", d->iou_type);
    Y6_REQUIRE(!d->use_dfl || (d->reg_max >= 1 && d->reg_max <= 63), "loss_forward: reg_max %d out of range", d->reg_max);
    Y6_REQUIRE(d->workspace_bytes >= y6_loss_workspace_bytes(), "loss_forward: workspace too small");
    Y6_REQUIRE((((uintptr_t)d->pred_bboxes | (uintptr_t)d->target_bboxes) & 15) == 0, "loss_forward: boxes must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    double* acc = (double*)d->workspace;
    Y6_HIP(hipMemsetAsync(acc, 0, 8 * sizeof(double), s));
    const size_t n_ba = (size_t)d->B * d->A;
    size_t g = (n_ba * d->C + 255) / 256;
    if (g > 4096) g = 4096;
    static const bool vec4 = !(getenv("Y6_LOSS_VEC4") && atoi(getenv("Y6_LOSS_VEC4")) == 0);     // A/B switch
    const bool cls4 = vec4 && d->C % 4 == 0 && n_ba * d->C < (1ull << 32) &&
                      (((uintptr_t)d->pred_scores | (uintptr_t)d->target_scores) & 15) == 0;
    if (cls4) {
        size_t g4 = (n_ba * d->C / 4 + 255) / 256;
        if (g4 > 1024) g4 = 1024;     // (two double atomics per block on the SAME two addresses: 8 192 of them in a row were most of the launch)
        hipLaunchKernelGGL(loss_cls4_kernel, dim3((unsigned)g4), dim3(256), 0, s, (const float4*)d->pred_scores, (const float4*)d->target_scores,
                           d->target_labels, d->fg_mask, (unsigned)(n_ba * d->C / 4), (unsigned)d->C, acc);
    } else
    hipLaunchKernelGGL(loss_cls_kernel, dim3((unsigned)g), dim3(256), 0, s, d->pred_scores, d->target_scores, d->target_labels,
                       d->fg_mask, n_ba, d->C, acc);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_box_kernel, dim3((unsigned)((n_ba + 255) / 256)), dim3(256), 0, s, d->pred_distri, d->pred_bboxes,
                       d->anchor_points_s, d->stride, d->target_bboxes, d->target_scores, d->fg_mask, d->B, d->A, d->C,
                       d->use_dfl, d->reg_max, d->iou_type, acc);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, s, acc, d->w_class, d->w_iou, d->w_dfl, d->use_dfl, d->out, d->norm_mode);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

extern "C" int y6_loss_forward_backward(const y6_loss_grad_desc* g, void* stream) {
    Y6_REQUIRE(g, "loss_forward_backward: null argument");
    int rc = y6_loss_forward(&g->fwd, stream);
    if (rc) return rc;
    return y6_loss_backward(g, stream);
}

extern "C" int y6_loss_backward(const y6_loss_grad_desc* g, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(g && g->dpred_scores && g->dpred_distri && g->fwd.out, "loss_backward: null argument");
    const y6_loss_desc* d = &g->fwd;
    hipStream_t s = (hipStream_t)stream;
    const size_t n_ba = (size_t)d->B * d->A;
    size_t gr = (n_ba * d->C + 255) / 256;
    if (gr > 8192) gr = 8192;
    static const bool vec4 = !(getenv("Y6_LOSS_VEC4") && atoi(getenv("Y6_LOSS_VEC4")) == 0);
    const bool cls4 = vec4 && d->C % 4 == 0 && n_ba * d->C < (1ull << 32) &&
                      (((uintptr_t)d->pred_scores | (uintptr_t)d->target_scores | (uintptr_t)g->dpred_scores) & 15) == 0;
    if (cls4) {
        size_t g4 = (n_ba * d->C / 4 + 255) / 256;
        if (g4 > 8192) g4 = 8192;
        hipLaunchKernelGGL(loss_cls_bwd4_kernel, dim3((unsigned)g4), dim3(256), 0, s, (const float4*)d->pred_scores,
                           (const float4*)d->target_scores, d->target_labels, d->fg_mask, (unsigned)(n_ba * d->C / 4), (unsigned)d->C,
                           d->out, d->w_class, g->grad_scale, (float4*)g->dpred_scores, d->norm_mode);
    } else
    hipLaunchKernelGGL(loss_cls_bwd_kernel, dim3((unsigned)gr), dim3(256), 0, s, d->pred_scores, d->target_scores, d->target_labels,
                       d->fg_mask, n_ba, d->C, d->out, d->w_class, g->grad_scale, g->dpred_scores, d->norm_mode);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_box_bwd_kernel, dim3((unsigned)((n_ba + 255) / 256)), dim3(256), 0, s, d->pred_distri, d->pred_bboxes,
                       d->anchor_points_s, d->stride, d->target_bboxes, d->target_scores, d->fg_mask, d->B, d->A, d->C, d->use_dfl,
                       d->reg_max, d->iou_type, d->out, d->w_iou, d->w_dfl, g->grad_scale, g->dpred_distri, d->box_mode, d->norm_mode);
    Y6_REQUIRE(d->box_mode == 0 || !d->use_dfl, "loss_backward: the anchor-based box form has no DFL");
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ---- self-distillation terms (descriptor: include/yolov6_hip.h y6_distill_desc)
static int distill_fill(const y6_distill_desc* d, DistillArgs* a, bool bwd) {
    Y6_REQUIRE(d && d->scores_s && d->scores_t && d->BA > 0 && d->C > 0 && d->temperature > 0.f, "distill: bad arguments");
    Y6_REQUIRE((d->distri_s == nullptr) == (d->distri_t == nullptr), "distill: student AND teacher DFL logits, or neither");
    Y6_REQUIRE(!d->distri_s || (d->fg_mask && d->target_scores && d->reg_max >= 1 && d->reg_max <= 63), "distill: the DFL term needs fg_mask, target_scores, reg_max");
    Y6_REQUIRE(bwd ? (d->coef && d->dscores && (!d->distri_s || d->ddistri)) : (d->acc != nullptr), "distill: null output");
    a->ps = d->scores_s;
    a->pt = d->scores_t;
    a->ds = d->distri_s;
    a->dt = d->distri_t;
    a->fg = d->fg_mask;
    a->tscore = d->target_scores;
    a->n_ba = (size_t)d->BA;
    a->C = d->C;
    a->nb = d->reg_max + 1;
    a->invT = 1.f / d->temperature;
    a->acc = d->acc;
    a->coef = d->coef;
    a->dscores = d->dscores;
    a->ddistri = d->ddistri;
    return Y6_OK;
}
extern "C" int y6_distill_forward(const y6_distill_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    DistillArgs a;
    int rc = distill_fill(d, &a, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Y6_HIP(hipMemsetAsync(a.acc, 0, 4 * sizeof(double), s));
    size_t g = (a.n_ba + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(distill_kernel<false>, dim3((unsigned)g), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
extern "C" int y6_distill_backward(const y6_distill_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    DistillArgs a;
    int rc = distill_fill(d, &a, true);
    if (rc) return rc;
    size_t g = (a.n_ba + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(distill_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
extern "C" int y6_distill_cw(const float* s_feat, const float* t_feat, int rows, int hw, float temperature, double* acc, const float* coef,
                             float* d_s_feat, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(s_feat && t_feat && rows > 0 && hw > 0 && temperature > 0.f && (acc != nullptr) != (coef != nullptr && d_s_feat != nullptr),
               "distill_cw: forward takes acc, backward takes coef and d_s_feat");
    if (acc)
        hipLaunchKernelGGL(distill_cw_kernel<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s_feat, t_feat, hw, 1.f / temperature,
                           acc, nullptr, nullptr);
    else
        hipLaunchKernelGGL(distill_cw_kernel<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s_feat, t_feat, hw, 1.f / temperature,
                           nullptr, coef, d_s_feat);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
