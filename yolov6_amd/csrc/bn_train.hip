// bn_train.hip — BatchNorm2d in TRAINING mode (batch statistics) on NHWC fp16 views: the two primitives the
// training-form forward needs around the conv kernels.
//   y6_bn_stats   per-channel mean and biased variance over B*H*W of a conv output - what
//                 torch.nn.functional.batch_norm(training=True) normalises with in every ConvModule
//                 (reference yolov6/layers/common.py:26-54) and every RepVGG branch (:250-255, :341-347)
//   y6_bn_apply   y = act( sum_b ( x_b * scale_b[c] + shift_b[c] ) ) for up to three branches: the RepVGG sum
//                 ReLU(bn(conv3x3) + bn(conv1x1) + bn_id(x)) (:250-255) in one pass, or a single ConvBN+act
// scale_b = gamma / sqrt(var + eps), shift_b = beta - mean * scale_b are tiny per-channel vectors the host derives
// from the statistics (and from which it updates running_mean / running_var with the unbiased variance, momentum
// 0.03, torch_utils.py:38-47).  Sums accumulate in double: E[x^2] - E[x]^2 is safe there.
// Forward only: there is no backward in the library yet.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void bn_stats_kernel(const __half* __restrict__ x, int cs, int co, long npix, int G,
                                                       long pix_per_block, double* __restrict__ ws, int C) {
    extern __shared__ double s_acc[];   // [2*C]
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * C; i += 256) s_acc[i] = 0.0;
    __syncthreads();
    const int R = 256 / G;              // pixel rows handled per step; threads >= R*G idle
    if (tid < R * G) {
        const int g = tid % G, prow = tid / G;
        const long p0 = (long)blockIdx.x * pix_per_block;
        const long p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
        double s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.0;
        for (long p = p0 + prow; p < p1; p += R) {
            const uint4 raw = *reinterpret_cast<const uint4*>(x + p * cs + co + g * 8);
            const __half* h = reinterpret_cast<const __half*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double v = (double)__half2float(h[j]);
                s[j] += v;
                q[j] += v * v;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            atomicAdd(&s_acc[g * 8 + j], s[j]);
            atomicAdd(&s_acc[C + g * 8 + j], q[j]);
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) atomicAdd(&ws[i], s_acc[i]);
}

__global__ void bn_stats_finalize_kernel(const double* __restrict__ ws, int C, double n, float* __restrict__ mean,
                                         float* __restrict__ var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = ws[c] / n;
    double v = ws[C + c] / n - m * m;
    mean[c] = (float)m;
    var[c] = (float)(v > 0.0 ? v : 0.0);
}

struct ApplyArgs {
    int n;
    const __half* x[3];
    int cs[3], co[3];
    const float* scale[3];
    const float* shift[3];
    __half* out;
    int ocs, oco, act, C;
    long npix;
};

__global__ __launch_bounds__(256) void bn_apply_kernel(const ApplyArgs a) {
    const int G = a.C >> 3;
    const long total = a.npix * G;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / G;
        const int g = (int)(i - p * G);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int b = 0; b < a.n; ++b) {
            const uint4 raw = *reinterpret_cast<const uint4*>(a.x[b] + p * a.cs[b] + a.co[b] + g * 8);
            const __half* h = reinterpret_cast<const __half*>(&raw);
            const float4 s0 = *reinterpret_cast<const float4*>(a.scale[b] + g * 8);
            const float4 s1 = *reinterpret_cast<const float4*>(a.scale[b] + g * 8 + 4);
            const float4 t0 = *reinterpret_cast<const float4*>(a.shift[b] + g * 8);
            const float4 t1 = *reinterpret_cast<const float4*>(a.shift[b] + g * 8 + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += __half2float(h[j]) * sc[j] + sh[j];
        }
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)y6_act(acc[j], a.act);
        *reinterpret_cast<h8_t*>(a.out + p * a.ocs + a.oco + g * 8) = o;
    }
}

bool view_ok(const y6_tensor& t) {
    return t.data && t.C % 8 == 0 && t.cstride % 8 == 0 && t.coff % 8 == 0 && (((uintptr_t)t.data) & 15) == 0;
}

}  // namespace

// [2C] final sums (the atomic path of y6_bn_stats) followed by [kBnPartBlocks][2C] block partials (y6_bn_train_stats, train.hip)
extern "C" size_t y6_bn_stats_workspace_bytes(int C) { return (size_t)2 * C * sizeof(double) * (size_t)(1 + kBnPartBlocks); }

extern "C" int y6_bn_stats(const y6_tensor* x, float* mean, float* var, void* workspace, size_t workspace_bytes, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(x && mean && var && workspace, "bn_stats: null argument");
    Y6_REQUIRE(view_ok(*x), "bn_stats: the view must be fp16 NHWC with 8-channel alignment");
    Y6_REQUIRE(x->C <= 2048, "bn_stats: at most 2048 channels");
    Y6_REQUIRE(workspace_bytes >= y6_bn_stats_workspace_bytes(x->C), "bn_stats: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int C = x->C, G = C / 8;
    const long npix = (long)x->B * x->H * x->W;
    Y6_REQUIRE(npix > 0, "bn_stats: empty tensor");
    double* ws = (double*)workspace;
    Y6_HIP(hipMemsetAsync(ws, 0, (size_t)2 * C * sizeof(double), s));
    const int R = 256 / G;
    long ppb = (long)R * 64;                      // 64 pixels per thread
    long blocks = (npix + ppb - 1) / ppb;
    if (blocks > 2048) {
        blocks = 2048;
        ppb = (npix + blocks - 1) / blocks;
    }
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)blocks), dim3(256), (size_t)2 * C * sizeof(double), s,
                       (const __half*)x->data, x->cstride, x->coff, npix, G, ppb, ws, C);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, ws, C, (double)npix,
                       mean, var);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

extern "C" int y6_bn_apply(const y6_bn_apply_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d && d->n >= 1 && d->n <= 3, "bn_apply: 1..3 branches");
    Y6_REQUIRE(view_ok(d->out), "bn_apply: the output view must be fp16 NHWC with 8-channel alignment");
    ApplyArgs a;
    memset(&a, 0, sizeof(a));
    a.n = d->n;
    for (int b = 0; b < d->n; ++b) {
        const y6_tensor& t = d->x[b];
        Y6_REQUIRE(view_ok(t) && d->scale[b] && d->shift[b], "bn_apply: branch %d is not a valid fp16 NHWC view", b);
        Y6_REQUIRE(t.B == d->out.B && t.H == d->out.H && t.W == d->out.W && t.C == d->out.C, "bn_apply: branch %d shape mismatch", b);
        Y6_REQUIRE((((uintptr_t)d->scale[b] | (uintptr_t)d->shift[b]) & 15) == 0, "bn_apply: scale/shift must be 16-byte aligned");
        a.x[b] = (const __half*)t.data;
        a.cs[b] = t.cstride;
        a.co[b] = t.coff;
        a.scale[b] = d->scale[b];
        a.shift[b] = d->shift[b];
    }
    a.out = (__half*)d->out.data;
    a.ocs = d->out.cstride;
    a.oco = d->out.coff;
    a.act = d->act;
    a.C = d->out.C;
    a.npix = (long)d->out.B * d->out.H * d->out.W;
    long g = (a.npix * (a.C / 8) + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
