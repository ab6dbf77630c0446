// train.hip — the elementwise / reduction half of the TRAINING step on NHWC fp16 activations (gfx950):
//   y6_bn_train_stats         batch statistics -> scale/shift/mean/invstd + running-stat update, all on device
//   y6_bnact_forward          act( sum_b x_b*scale_b + shift_b ) [+ alpha*res]      (RepVGG train-form sum, ConvModule BN+act)
//   y6_bnact_backward         its backward incl. BatchNorm's (two passes: per-channel sums, then the gradients)
//   y6_wgrad_transpose        channel-major sampling of an activation for the weight-gradient GEMM (wgrad.hip)
//   y6_pack_weights_batched   per-step fp32 master weights -> packed fp16 MFMA images (forward, data-gradient, convT)
//   y6_sppf_pool_backward     backward of the three chained 5x5 max-pools
//   y6_head_pack / _unpack    Detect training branch: sigmoid + [B,A,C] packing, and its backward
//   y6_space_to_depth2, y6_channel_sum, y6_tensor_add, fused SGD / overflow check / loss-scale update
// Reference: yolov6/layers/common.py:45-49, :250-255 (forward the autograd graph is built from), torch's
// batch_norm_backward / max_pool2d_with_indices_backward semantics, yolov6/models/effidehead.py:72-92,
// yolov6/solver/build.py:10-30, yolov6/core/engine.py:169-176, :258-266.
// All of these are HBM-bound: 16-byte accesses, one pass per tensor, per-channel sums in double.
#include "common.hpp"
#include "plan_internal.hpp"

namespace {

bool view_ok(const y6_tensor& t) {
    return t.data && t.C % 8 == 0 && t.cstride % 8 == 0 && t.coff % 8 == 0 && (((uintptr_t)t.data) & 15) == 0;
}
bool same_shape(const y6_tensor& a, const y6_tensor& b) { return a.B == b.B && a.H == b.H && a.W == b.W && a.C == b.C; }

inline unsigned grid_for(size_t total, int block, size_t cap = 256 * 32) {
    size_t g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ void load8(const __half* p, float (&v)[8]) {
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    const __half* h = reinterpret_cast<const __half*>(&raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __half2float(h[j]);
}
__device__ __forceinline__ void store8(__half* p, const float (&v)[8]) {
    h8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (_Float16)v[j];
    *reinterpret_cast<h8_t*>(p) = o;
}
__device__ __forceinline__ void loadf8(const float* p, float (&v)[8], float dflt) {
    if (!p) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = dflt;
        return;
    }
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
}

// ------------------------------------------------------------------ batch statistics (forward)
// The constants of one channel from its sums (torch.nn.BatchNorm2d training semantics).
__device__ __forceinline__ void bn_finalize_channel(int c, double sum, double sumsq, double n, const y6_bn_train_desc& d) {
    const double m = sum / n;
    double v = sumsq / n - m * m;
    v = v > 0.0 ? v : 0.0;
    const float mean = (float)m, var = (float)v;
    const float invstd = 1.f / sqrtf(var + d.eps);
    const float g = d.gamma ? d.gamma[c] : 1.f, b = d.beta ? d.beta[c] : 0.f;
    const float scale = g * invstd;
    d.scale[c] = scale;
    d.shift[c] = b - mean * scale;
    d.mean[c] = mean;
    d.invstd[c] = invstd;
    if (d.running_mean) {
        const float mo = d.momentum;
        const float unb = (float)(v * (n / (n > 1.0 ? n - 1.0 : 1.0)));
        d.running_mean[c] = (1.f - mo) * d.running_mean[c] + mo * mean;
        d.running_var[c] = (1.f - mo) * d.running_var[c] + mo * unb;
    }
}

__global__ __launch_bounds__(256) void bn_sum_kernel(const __half* __restrict__ x, int cs, int co, long npix, int G,
                                                     long pix_per_block, double* __restrict__ ws, int C) {
    extern __shared__ double s_acc[];   // [2*C]
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * C; i += 256) s_acc[i] = 0.0;
    __syncthreads();
    const int R = 256 / G;
    if (tid < R * G) {
        const int g = tid % G, prow = tid / G;
        const long p0 = (long)blockIdx.x * pix_per_block;
        const long p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
        double s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.0;
        // four independent 16-byte loads in flight per thread (a one-load-per-iteration loop ran at 1.3 TB/s); partial sums of
        // the four pixels are formed in fp32 (exact enough: fp16 inputs, 4 terms) before they enter the double accumulators
        long p = p0 + prow;
        for (; p + 3 * (long)R < p1; p += 4 * (long)R) {
            float v0[8], v1[8], v2[8], v3[8];
            load8(x + p * cs + co + g * 8, v0);
            load8(x + (p + R) * cs + co + g * 8, v1);
            load8(x + (p + 2 * (long)R) * cs + co + g * 8, v2);
            load8(x + (p + 3 * (long)R) * cs + co + g * 8, v3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += (double)((v0[j] + v1[j]) + (v2[j] + v3[j]));
                q[j] += (double)v0[j] * (double)v0[j] + (double)v1[j] * (double)v1[j] + (double)v2[j] * (double)v2[j] +
                        (double)v3[j] * (double)v3[j];
            }
        }
        for (; p < p1; p += R) {
            float v[8];
            load8(x + p * cs + co + g * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += (double)v[j];
                q[j] += (double)v[j] * (double)v[j];
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

// The same sums without atomics: every block stores its 2C partial sums, the finalize kernel adds the blocks' partials in a
// fixed order (16 interleaved sub-sums per value, then those in order) - deterministic, and no workspace has to start zeroed.
// Why: with the one-kernel-plus-atomics form a BatchNorm statistics launch cost 19 us + bytes / 4.5 TB/s
// (profiles/r03/bench_train_ops_r03q.json: 13 MB tensors 19.6 us, 52 MB 24.9 us) - the floor is the serial loop of a thread over
// its 64 pixel rows (16 rounds of four loads), which could not be shortened while every extra block meant 2C more same-address
// double atomics.  Partials make blocks cheap: a thread walks 16 rows (4 rounds).
__global__ __launch_bounds__(256) void bn_sum_part_kernel(const __half* __restrict__ x, int cs, int co, long npix, int G,
                                                          long pix_per_block, double* __restrict__ part, int C) {
    extern __shared__ double s_rows[];   // [R][2][8][G]: the sums of pixel-row thread r, value k, channel g*8 + j at ((r*2 + k)*8 + j)*G + g
    const int tid = threadIdx.x;
    const int R = 256 / G;
    if (tid < R * G) {
        const int g = tid % G, prow = tid / G;
        const long p0 = (long)blockIdx.x * pix_per_block;
        const long p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
        double s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.0;
        long p = p0 + prow;
        // eight independent 16-byte loads in flight per thread; the sums and the sums of squares of the eight pixels are formed in
        // fp32 (a product of two fp16 values is exact in fp32; eight terms) before they enter the double accumulators - one
        // conversion and one fp64 add per value and round instead of one fp64 multiply-add per element (round 6: 2.2 -> see
        // profiles/r06 TB/s; the fp64 work, not the loads, paced the loop)
        for (; p + 7 * (long)R < p1; p += 8 * (long)R) {
            uint4 raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const uint4*>(x + (p + u * (long)R) * cs + co + g * 8);
            float fs[8], fq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) fs[j] = fq[j] = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const __half* h = reinterpret_cast<const __half*>(&raw[u]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = __half2float(h[j]);
                    fs[j] += v;
                    fq[j] = __builtin_fmaf(v, v, fq[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += (double)fs[j];
                q[j] += (double)fq[j];
            }
        }
        for (; p < p1; p += R) {
            float v[8];
            load8(x + p * cs + co + g * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += (double)v[j];
                q[j] += (double)v[j] * (double)v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s_rows[((prow * 2 + 0) * 8 + j) * G + g] = s[j];
            s_rows[((prow * 2 + 1) * 8 + j) * G + g] = q[j];
        }
    }
    __syncthreads();
    // the R pixel-row threads of a channel are added in row order (no atomics anywhere: the statistics are reproducible)
    double* dst = part + (size_t)blockIdx.x * (size_t)(2 * C);
    for (int i = tid; i < 2 * C; i += 256) {   // i = (k*8 + j)*G + g
        const int g = i % G, kj = i / G;
        double tot = 0.0;
        for (int r = 0; r < R; ++r) tot += s_rows[(size_t)r * 2 * C + i];
        dst[(kj >> 3) * C + g * 8 + (kj & 7)] = tot;
    }
}

// sum over the blocks' partials of value v (stride `nvals` doubles between blocks): 16 interleaved sub-sums (one per 32-lane
// row of the block, four independent loads in flight), combined in row order.  512 threads; red: [16][33] doubles.
__device__ __forceinline__ double part_subsum(const double* __restrict__ p, size_t stride, int nblocks, int prt) {
    double s = 0.0;
    int b = prt;
    // (round 6, last session) four of the rounds below requested at once - 64 loads in flight, added in the order of the four rounds
    // (same bits): with 800-1024 partials the kernel was four dependent round trips to memory the partials were just written to
    for (; b + 3 * 256 + 240 < nblocks; b += 1024) {
        double v[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) v[u] = p[(size_t)(b + 16 * u) * stride];
#pragma unroll
        for (int u = 0; u < 64; u += 4) s += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
    }
    for (; b + 256 + 240 < nblocks; b += 512) {   // two rounds at once (400-1000 partials)
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = p[(size_t)(b + 16 * u) * stride];
#pragma unroll
        for (int u = 0; u < 32; u += 4) s += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
    }
    for (; b + 240 < nblocks; b += 256) {   // sixteen independent loads in flight (round 6: the loop is a chain of global round trips -
        double v[16];                       // 17.7 us per backward BatchNorm with four in flight, 63 of them per training step)
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(b + 16 * u) * stride];
#pragma unroll
        for (int u = 0; u < 16; u += 4) s += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
    }
    for (; b + 48 < nblocks; b += 64) {
        const double a0 = p[(size_t)b * stride], a1 = p[(size_t)(b + 16) * stride], a2 = p[(size_t)(b + 32) * stride],
                     a3 = p[(size_t)(b + 48) * stride];
        s += (a0 + a1) + (a2 + a3);
    }
    for (; b < nblocks; b += 16) s += p[(size_t)b * stride];
    return s;
}

// block = 16 channels x {sum, sum of squares}
__global__ __launch_bounds__(512) void bn_train_finalize_part_kernel(const double* __restrict__ part, int nblocks, int C, double n,
                                                                    const y6_bn_train_desc d) {
    __shared__ double red[16][33];
    const int t = threadIdx.x, vi = t & 31, prt = t >> 5;
    const int k = vi >> 4, c = blockIdx.x * 16 + (vi & 15);
    red[prt][vi] = c < C ? part_subsum(part + (size_t)k * C + c, (size_t)2 * C, nblocks, prt) : 0.0;
    __syncthreads();
    if (t < 32) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += red[q][t];
        red[0][t] = tot;
    }
    __syncthreads();
    if (blockIdx.x == 0 && t == 0 && d.num_batches_tracked) *d.num_batches_tracked += 1;
    if (t < 16 && blockIdx.x * 16 + t < C) bn_finalize_channel(blockIdx.x * 16 + t, red[0][t], red[0][16 + t], n, d);
}

// ---- (round 6) the statistics of up to three tensors of one shape in ONE pair of launches (a RepVGG block's 3x3 / 1x1 / identity
// branches, yolov6/layers/common.py:250-255): blockIdx.y picks the tensor.  Each tensor's block walks the same pixels and adds in the
// same order as in a launch of its own, so the statistics have the same bits; what goes is four of the six launches (a statistics
// pass over a 13-52 MB map cost 12-22 us, most of it the two launches' ramps).
struct BnMultiArgs {
    const __half* x[3];
    int cs[3], co[3];
    double* part[3];
    y6_bn_train_desc d[3];
};

__global__ __launch_bounds__(256) void bn_sum_part_multi_kernel(const BnMultiArgs m, long npix, int G, long pix_per_block, int C) {
    extern __shared__ double s_rows[];   // as in bn_sum_part_kernel
    const int t = blockIdx.y;
    const __half* __restrict__ x = m.x[t];
    const int cs = m.cs[t], co = m.co[t];
    const int tid = threadIdx.x;
    const int R = 256 / G;
    if (tid < R * G) {
        const int g = tid % G, prow = tid / G;
        const long p0 = (long)blockIdx.x * pix_per_block;
        const long p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
        double s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.0;
        long p = p0 + prow;
        for (; p + 7 * (long)R < p1; p += 8 * (long)R) {
            uint4 raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const uint4*>(x + (p + u * (long)R) * cs + co + g * 8);
            float fs[8], fq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) fs[j] = fq[j] = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const __half* h = reinterpret_cast<const __half*>(&raw[u]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = __half2float(h[j]);
                    fs[j] += v;
                    fq[j] = __builtin_fmaf(v, v, fq[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += (double)fs[j];
                q[j] += (double)fq[j];
            }
        }
        for (; p < p1; p += R) {
            float v[8];
            load8(x + p * cs + co + g * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += (double)v[j];
                q[j] += (double)v[j] * (double)v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s_rows[((prow * 2 + 0) * 8 + j) * G + g] = s[j];
            s_rows[((prow * 2 + 1) * 8 + j) * G + g] = q[j];
        }
    }
    __syncthreads();
    double* dst = m.part[t] + (size_t)blockIdx.x * (size_t)(2 * C);
    for (int i = tid; i < 2 * C; i += 256) {
        const int g = i % G, kj = i / G;
        double tot = 0.0;
        for (int r = 0; r < R; ++r) tot += s_rows[(size_t)r * 2 * C + i];
        dst[(kj >> 3) * C + g * 8 + (kj & 7)] = tot;
    }
}

__global__ __launch_bounds__(512) void bn_train_finalize_part_multi_kernel(const BnMultiArgs m, int nblocks, int C, double n) {
    __shared__ double red[16][33];
    const int tn = blockIdx.y;
    const double* __restrict__ part = m.part[tn];
    const y6_bn_train_desc& d = m.d[tn];
    const int t = threadIdx.x, vi = t & 31, prt = t >> 5;
    const int k = vi >> 4, c = blockIdx.x * 16 + (vi & 15);
    red[prt][vi] = c < C ? part_subsum(part + (size_t)k * C + c, (size_t)2 * C, nblocks, prt) : 0.0;
    __syncthreads();
    if (t < 32) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += red[q][t];
        red[0][t] = tot;
    }
    __syncthreads();
    if (blockIdx.x == 0 && t == 0 && d.num_batches_tracked) *d.num_batches_tracked += 1;
    if (t < 16 && blockIdx.x * 16 + t < C) bn_finalize_channel(blockIdx.x * 16 + t, red[0][t], red[0][16 + t], n, d);
}

// The atomic form (A/B: Y6_BN_ATOMICS=1): turns the sums into the BatchNorm constants and leaves the workspace zeroed again: a caller whose workspace starts zeroed
// (`workspace_clean`) never needs a memset launch (profiles/r03/rocprofv3_kernel_stats_train_r03c.csv: 279 of them per training
// step, ~4.6 us each).  Folding this kernel into the last block of the sums (device-scope ticket + __threadfence) was measured
// and lost: the agent-scope release / acquire is an L2 write-back + invalidate per block on gfx950 (bn_stats 4.07 -> 5.88 ms,
// bnact_bwd 8.94 -> 10.33 ms per step, profiles/r03/bench_train_r03l_{old,new}.json).
__global__ void bn_train_finalize_kernel(double* __restrict__ ws, int C, double n, const y6_bn_train_desc d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && d.num_batches_tracked) *d.num_batches_tracked += 1;
    if (c >= C) return;
    bn_finalize_channel(c, ws[c], ws[C + c], n, d);
    ws[c] = 0.0;
    ws[C + c] = 0.0;
}

static bool bn_use_atomics() {
    static const bool v = getenv("Y6_BN_ATOMICS") != nullptr;
    return v;
}

int bn_train_stats_launch(const y6_bn_train_desc* d, hipStream_t s) {
    Y6_REQUIRE(d && d->scale && d->shift && d->mean && d->invstd && d->workspace, "bn_train_stats: null argument");
    Y6_REQUIRE(view_ok(d->x), "bn_train_stats: the view must be fp16 NHWC with 8-channel alignment");
    const int C = d->x.C, G = C / 8;
    Y6_REQUIRE(C <= 2048, "bn_train_stats: at most 2048 channels");
    const long npix = (long)d->x.B * d->x.H * d->x.W;
    Y6_REQUIRE(npix > 0, "bn_train_stats: empty tensor");
    Y6_REQUIRE(d->workspace_bytes >= y6_bn_stats_workspace_bytes_for(C, npix), "bn_train_stats: workspace too small");
    double* ws = (double*)d->workspace;
    const int R = 256 / G;
    if (!bn_use_atomics()) {
        long ppb = (long)R * 16;
        long blocks = (npix + ppb - 1) / ppb;
        if (blocks > kBnPartBlocks) {
            blocks = kBnPartBlocks;
            ppb = (npix + blocks - 1) / blocks;
            blocks = (npix + ppb - 1) / ppb;
        }
        double* part = ws + (size_t)2 * C;
        hipLaunchKernelGGL(bn_sum_part_kernel, dim3((unsigned)blocks), dim3(256), (size_t)R * 2 * C * sizeof(double), s,
                           (const __half*)d->x.data, d->x.cstride, d->x.coff, npix, G, ppb, part, C);
        Y6_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_train_finalize_part_kernel, dim3((unsigned)((C + 15) / 16)), dim3(512), 0, s, (const double*)part, (int)blocks,
                           C, (double)npix, *d);
        Y6_LAUNCH_CHECK();
        return Y6_OK;
    }
    // the workspace must be all-zero when the sums start; the finalize kernel leaves it so.  A caller that allocated it zeroed
    // and uses it for nothing else says so (workspace_clean) and the memset launch is dropped.
    if (!d->workspace_clean) Y6_HIP(hipMemsetAsync(ws, 0, (size_t)2 * C * sizeof(double), s));
    long ppb = (long)R * 64;
    long blocks = (npix + ppb - 1) / ppb;
    if (blocks > 2048) {
        blocks = 2048;
        ppb = (npix + blocks - 1) / blocks;
    }
    hipLaunchKernelGGL(bn_sum_kernel, dim3((unsigned)blocks), dim3(256), (size_t)2 * C * sizeof(double), s,
                       (const __half*)d->x.data, d->x.cstride, d->x.coff, npix, G, ppb, ws, C);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, ws, C, (double)npix, *d);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// up to three statistics ops of one shape; anything else (different shapes, the atomic A/B form, one tensor) runs them one by one
int bn_train_stats_multi_launch(const y6_bn_train_multi_desc* md, hipStream_t s) {
    Y6_REQUIRE(md && md->n >= 1 && md->n <= 3, "bn_train_stats_multi: 1..3 tensors");
    const int n = md->n;
    bool same = n > 1 && !bn_use_atomics();
    for (int t = 0; t < n; ++t) {
        const y6_bn_train_desc* d = &md->d[t];
        Y6_REQUIRE(d->scale && d->shift && d->mean && d->invstd && d->workspace, "bn_train_stats_multi: null argument");
        Y6_REQUIRE(view_ok(d->x), "bn_train_stats_multi: the views must be fp16 NHWC with 8-channel alignment");
        same = same && same_shape(d->x, md->d[0].x);
        for (int u = 0; u < t; ++u)    // two ops of one launch must not share a BatchNorm (the running statistics would race)
            Y6_REQUIRE(d->scale != md->d[u].scale && d->workspace != md->d[u].workspace &&
                           (d->running_mean == nullptr || d->running_mean != md->d[u].running_mean),
                       "bn_train_stats_multi: two entries share outputs");
    }
    static const bool off = getenv("Y6_BN_MULTI") != nullptr && atoi(getenv("Y6_BN_MULTI")) == 0;   // A/B switch
    if (!same || off) {
        for (int t = 0; t < n; ++t) {
            int rc = bn_train_stats_launch(&md->d[t], s);
            if (rc) return rc;
        }
        return Y6_OK;
    }
    const y6_tensor& x0 = md->d[0].x;
    const int C = x0.C, G = C / 8;
    Y6_REQUIRE(C <= 2048, "bn_train_stats: at most 2048 channels");
    const long npix = (long)x0.B * x0.H * x0.W;
    Y6_REQUIRE(npix > 0, "bn_train_stats: empty tensor");
    const int R = 256 / G;
    long ppb = (long)R * 16;                         // the geometry of bn_train_stats_launch: same blocks, same additions
    long blocks = (npix + ppb - 1) / ppb;
    if (blocks > kBnPartBlocks) {
        blocks = kBnPartBlocks;
        ppb = (npix + blocks - 1) / blocks;
        blocks = (npix + ppb - 1) / ppb;
    }
    BnMultiArgs m;
    memset(&m, 0, sizeof(m));
    for (int t = 0; t < n; ++t) {
        const y6_bn_train_desc* d = &md->d[t];
        Y6_REQUIRE(d->workspace_bytes >= y6_bn_stats_workspace_bytes_for(C, npix), "bn_train_stats_multi: workspace too small");
        m.x[t] = (const __half*)d->x.data;
        m.cs[t] = d->x.cstride;
        m.co[t] = d->x.coff;
        m.part[t] = (double*)d->workspace + (size_t)2 * C;
        m.d[t] = *d;
    }
    hipLaunchKernelGGL(bn_sum_part_multi_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), (size_t)R * 2 * C * sizeof(double), s, m,
                       npix, G, ppb, C);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_train_finalize_part_multi_kernel, dim3((unsigned)((C + 15) / 16), (unsigned)n), dim3(512), 0, s, m, (int)blocks, C,
                       (double)npix);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ------------------------------------------------------------------ bn + branch sum + activation (+ shortcut)
struct BnActArgs {
    int n;
    const __half* x[3];
    int cs[3], co[3];
    const float* scale[3];
    const float* shift[3];
    const __half* res;
    int rcs, rco;
    const float* alpha;
    __half* out;
    int ocs, oco, act, C;
    long npix;
};

__device__ __forceinline__ void preact8(const BnActArgs& a, long p, int g, float (&z)[8], float (&xb)[3][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = 0.f;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (b >= a.n) break;
        float sc[8], sh[8];
        load8(a.x[b] + p * a.cs[b] + a.co[b] + g * 8, xb[b]);
        loadf8(a.scale[b] ? a.scale[b] + g * 8 : nullptr, sc, 1.f);
        loadf8(a.shift[b] ? a.shift[b] + g * 8 : nullptr, sh, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] += xb[b][j] * sc[j] + sh[j];
    }
}

__global__ __launch_bounds__(256) void bnact_fwd_kernel(const BnActArgs a) {
    const int G = a.C >> 3;
    const long total = a.npix * G;
    const float alpha = a.res ? (a.alpha ? *a.alpha : 1.f) : 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / G;
        const int g = (int)(i - p * G);
        float z[8], xb[3][8];
        preact8(a, p, g, z, xb);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = y6_act(z[j], a.act);
        if (a.res) {
            float r[8];
            load8(a.res + p * a.rcs + a.rco + g * 8, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += alpha * r[j];
        }
        store8(a.out + p * a.ocs + a.oco + g * 8, o);
    }
}

int fill_bnact_args(const y6_bnact_desc* d, BnActArgs* a, bool need_out) {
    Y6_REQUIRE(d && d->n >= 1 && d->n <= 3, "bnact: 1..3 branches");
    memset(a, 0, sizeof(*a));
    a->n = d->n;
    const y6_tensor& ref = d->x[0];
    for (int b = 0; b < d->n; ++b) {
        const y6_tensor& t = d->x[b];
        Y6_REQUIRE(view_ok(t), "bnact: branch %d is not a valid fp16 NHWC view", b);
        Y6_REQUIRE(same_shape(t, ref), "bnact: branch %d shape mismatch", b);
        Y6_REQUIRE((((uintptr_t)d->scale[b] | (uintptr_t)d->shift[b]) & 15) == 0, "bnact: scale/shift must be 16-byte aligned");
        a->x[b] = (const __half*)t.data;
        a->cs[b] = t.cstride;
        a->co[b] = t.coff;
        a->scale[b] = d->scale[b];
        a->shift[b] = d->shift[b];
    }
    if (d->res.data) {
        Y6_REQUIRE(view_ok(d->res) && same_shape(d->res, ref), "bnact: bad shortcut view");
        a->res = (const __half*)d->res.data;
        a->rcs = d->res.cstride;
        a->rco = d->res.coff;
        a->alpha = d->res_alpha;
    }
    if (need_out) {
        Y6_REQUIRE(view_ok(d->out) && same_shape(d->out, ref), "bnact: bad output view");
        a->out = (__half*)d->out.data;
        a->ocs = d->out.cstride;
        a->oco = d->out.coff;
    }
    Y6_REQUIRE(d->act == Y6_ACT_NONE || d->act == Y6_ACT_RELU || d->act == Y6_ACT_SILU || d->act == Y6_ACT_HARDSWISH, "bnact: bad activation");
    a->act = d->act;
    a->C = ref.C;
    a->npix = (long)ref.B * ref.H * ref.W;
    return Y6_OK;
}

int bnact_forward_launch(const y6_bnact_desc* d, hipStream_t s) {
    BnActArgs a;
    int rc = fill_bnact_args(d, &a, true);
    if (rc) return rc;
    hipLaunchKernelGGL(bnact_fwd_kernel, dim3(grid_for((size_t)a.npix * (a.C / 8), 256, 256 * 16)), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ------------------------------------------------------------------ backward of the above
__device__ __forceinline__ float act_grad(float z, int act) {
    switch (act) {
        case Y6_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case Y6_ACT_SILU: {
            const float s = 1.f / (1.f + __expf(-z));
            return s * (1.f + z * (1.f - s));
        }
        case Y6_ACT_HARDSWISH: return z < -3.f ? 0.f : (z > 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
        default: return 1.f;
    }
}

struct BnActBwdArgs {
    BnActArgs f;
    const float* mean[3];
    const float* invstd[3];
    const float* gamma[3];
    const __half* dout;
    int dcs, dco;
    __half* dx[3];
    int xcs[3], xco[3], xdil[3], xacc[3], xH[3], xW[3];
    float* dgamma[3];
    float* dbeta[3];
    __half* dres;
    int rcs, rco, racc;
    float* dalpha;
    double* ws;       // [ (1 + n) * C + 1 ] : sum dz | sum dz*xhat_b ... | sum dout*res
    int H, W;
    double* part;     // atomic-free form: [blocks][(1 + n) * C + 1] block partials of the same sums
    int nparts;
};

__device__ __forceinline__ void load4(const __half* p, float (&v)[4]) {
    const uint2 raw = *reinterpret_cast<const uint2*>(p);
    const __half* h = reinterpret_cast<const __half*>(&raw);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __half2float(h[j]);
}
__device__ __forceinline__ void store4(__half* p, const float (&v)[4]) {
    h4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (_Float16)v[j];
    *reinterpret_cast<h4_t*>(p) = o;
}
// pass 1: per-channel sums of dz and dz*x_b (RAW branch inputs; double), optional sum dout*res.
// sum dz*xhat_b = (sum dz*x_b - mean_b * sum dz) * invstd_b is formed afterwards: the loop needs no statistics, which keeps
// it at ~100 registers (4 waves per SIMD in flight instead of 2 - this pass is pure HBM streaming).
__global__ __launch_bounds__(256) void bnact_bwd_reduce_kernel(const BnActBwdArgs a, long pix_per_block) {
    extern __shared__ double s_acc[];   // [(1+n)*C]
    const int C = a.f.C, G = C >> 2, n = a.f.n;      // 4 channels per thread (8-byte loads)
    const int tid = threadIdx.x;
    const int nacc = (1 + n) * C;
    for (int i = tid; i < nacc; i += 256) s_acc[i] = 0.0;
    __syncthreads();
    const int R = 256 / G;
    double s_alpha = 0.0;
    if (tid < R * G) {
        const int g = tid % G, prow = tid / G, c0 = g * 4;
        const long p0 = (long)blockIdx.x * pix_per_block;
        const long p1 = p0 + pix_per_block < a.f.npix ? p0 + pix_per_block : a.f.npix;
        double sdz[4], sxy[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sdz[j] = sxy[0][j] = sxy[1][j] = sxy[2][j] = 0.0;
        float sc[3][4], sht[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sht[j] = 0.f;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc[b][j] = a.f.scale[b] ? a.f.scale[b][c0 + j] : 1.f;
                sht[j] += a.f.shift[b] ? a.f.shift[b][c0 + j] : 0.f;
            }
        }
        for (long p = p0 + prow; p < p1; p += R) {
            float z[4], xb[3][4], go[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = sht[j];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (b >= n) break;
                load4(a.f.x[b] + p * a.f.cs[b] + a.f.co[b] + c0, xb[b]);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] += xb[b][j] * sc[b][j];
            }
            load4(a.dout + p * a.dcs + a.dco + c0, go);
            if (a.dalpha) {
                float r[4];
                load4(a.f.res + p * a.f.rcs + a.f.rco + c0, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) s_alpha += (double)(go[j] * r[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dz = go[j] * act_grad(z[j], a.f.act);
                sdz[j] += (double)dz;
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    if (b < n) sxy[b][j] += (double)(dz * xb[b][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&s_acc[c0 + j], sdz[j]);
            for (int b = 0; b < n; ++b)
                if (a.mean[b]) atomicAdd(&s_acc[(1 + b) * C + c0 + j], sxy[b][j]);
        }
    }
    __syncthreads();
    for (int i = tid; i < nacc; i += 256) atomicAdd(&a.ws[i], s_acc[i]);
    if (a.dalpha) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s_alpha += __shfl_xor(s_alpha, o, 64);
        if ((tid & 63) == 0) atomicAdd(&a.ws[nacc], s_alpha);
    }
}

__device__ __forceinline__ long bwd_dx_index(const BnActBwdArgs& a, int b, long p) {
    if (a.xdil[b] != 2) return p;
    // (b, y, x) of the logical grid -> (b, 2y, 2x) of the dilated buffer
    const int hw = a.H * a.W;
    const int bi = (int)(p / hw), rem = (int)(p - (long)bi * hw);
    const int y = rem / a.W, x = rem - y * a.W;
    return ((long)bi * a.xH[b] + 2 * y) * a.xW[b] + 2 * x;
}

// pass 2: gradients wrt every branch input (and the shortcut), 4 channels per thread (8-byte accesses: the per-channel
// constants of 4 channels x 3 branches stay in ~50 registers).  With dz the activation gradient,
//   dx_b = k1_b*(dz - mean(dz) - xhat_b*mean(dz*xhat_b)) = k1_b*dz + A_b*x_b + B_b,
//   k1 = gamma*invstd,  m2 = (S_b - mu*S0)*invstd/N,  A = -k1*m2*invstd,  B = -k1*S0/N - A*mu     (S0 = sum dz, S_b = sum dz*x_b)
struct BwdConsts4 {
    float sc[3][4], k1[3][4], A[3][4], B[3][4], sht[4];
};
__device__ __forceinline__ void load_bwd_consts4(const BnActBwdArgs& a, int c0, BwdConsts4& c) {
    const int C = a.f.C;
    const double invN = 1.0 / (double)a.f.npix;
#pragma unroll
    for (int j = 0; j < 4; ++j) c.sht[j] = 0.f;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (b >= a.f.n) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = c0 + j;
            c.sc[b][j] = a.f.scale[b] ? a.f.scale[b][ch] : 1.f;
            c.sht[j] += a.f.shift[b] ? a.f.shift[b][ch] : 0.f;
            if (a.mean[b]) {
                const double s0 = a.ws[ch], sb = a.ws[(1 + b) * C + ch];
                const float mu = a.mean[b][ch], is = a.invstd[b][ch];
                const float k1 = (a.gamma[b] ? a.gamma[b][ch] : 1.f) * is;
                const float m2 = (float)((sb - (double)mu * s0) * (double)is * invN);
                c.k1[b][j] = k1;
                c.A[b][j] = -k1 * m2 * is;
                c.B[b][j] = -k1 * (float)(s0 * invN) - c.A[b][j] * mu;
            } else {
                c.k1[b][j] = c.sc[b][j];
                c.A[b][j] = c.B[b][j] = 0.f;
            }
        }
    }
}

__device__ __forceinline__ void bwd_apply_one4(const BnActBwdArgs& a, long p, int c0, const BwdConsts4& c, float alpha) {
    const int n = a.f.n;
    float z[4], xb[3][4], go[4], dz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = c.sht[j];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (b >= n) break;
        load4(a.f.x[b] + p * a.f.cs[b] + a.f.co[b] + c0, xb[b]);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] += xb[b][j] * c.sc[b][j];
    }
    load4(a.dout + p * a.dcs + a.dco + c0, go);
#pragma unroll
    for (int j = 0; j < 4; ++j) dz[j] = go[j] * act_grad(z[j], a.f.act);
    if (a.dres) {
        __half* q = a.dres + p * a.rcs + a.rco + c0;
        float r[4];
        if (a.racc) load4(q, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (a.racc ? r[j] : 0.f) + alpha * go[j];
        store4(q, r);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (b >= n) break;
        if (!a.dx[b]) continue;
        float d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = c.k1[b][j] * dz[j] + c.A[b][j] * xb[b][j] + c.B[b][j];
        long q = p;
        if (a.xdil[b] == 2) {       // (b, y, x) of the logical grid -> (b, 2y, 2x) of the dilated buffer
            const int hw = a.H * a.W;
            const int bi = (int)(p / hw), rem = (int)(p - (long)bi * hw);
            const int y = rem / a.W, x = rem - y * a.W;
            q = ((long)bi * a.xH[b] + 2 * y) * a.xW[b] + 2 * x;
        }
        __half* dst = a.dx[b] + q * a.xcs[b] + a.xco[b] + c0;
        if (a.xacc[b]) {
            float old[4];
            load4(dst, old);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] += old[j];
        }
        store4(dst, d);
    }
}

__global__ __launch_bounds__(256) void bnact_bwd_apply_kernel(const BnActBwdArgs a) {
    const int G4 = a.f.C >> 2;
    const float alpha = a.f.res ? (a.f.alpha ? *a.f.alpha : 1.f) : 0.f;
    BwdConsts4 c;
    if (256 % G4 == 0) {            // a thread keeps ONE 4-channel group: its constants stay in registers
        const int g = threadIdx.x % G4, rpb = 256 / G4;
        load_bwd_consts4(a, g * 4, c);
        for (long p = (long)blockIdx.x * rpb + threadIdx.x / G4; p < a.f.npix; p += (long)gridDim.x * rpb) bwd_apply_one4(a, p, g * 4, c, alpha);
        return;
    }
    const long total = a.f.npix * G4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / G4;
        const int g = (int)(i - p * G4);
        load_bwd_consts4(a, g * 4, c);
        bwd_apply_one4(a, p, g * 4, c, alpha);
    }
}

// ---- v2 of both passes: 8 channels per thread (16-byte accesses), the loop-invariant per-channel constants of the apply pass
// in LDS instead of registers.  The 4-channel kernels above keep 4 waves per SIMD with 4-7 loads of 8 bytes in flight per
// thread: 30-55 KB per CU, 2.1 TB/s measured (profiles/r03/bench_train_r03c.json: bnact_bwd 14.0 of the 57 ms step, the
// largest single item).  Here a thread has 5-8 loads of 16 bytes in flight (reduce: two pixels) at 4-5 waves per SIMD.
// Taken when 256 % (C / 8) == 0 and every view is 16-byte aligned (every BatchNorm of the N / S / L graphs).
// dgamma / dbeta of branch b, channel c, from the finished sums
__device__ __forceinline__ void bwd_param_grads(const BnActBwdArgs& a, int b, int c, double s0, double sb) {
    if (a.dgamma[b]) a.dgamma[b][c] += (float)((sb - (double)a.mean[b][c] * s0) * (double)a.invstd[b][c]);
    if (a.dbeta[b]) a.dbeta[b][c] += (float)s0;
}

__device__ __forceinline__ void unpack8(const uint4& raw, float (&v)[8]) {
    const __half* h = reinterpret_cast<const __half*>(&raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __half2float(h[j]);
}
__device__ __forceinline__ uint4 ld16(const __half* p) { return *reinterpret_cast<const uint4*>(p); }

__global__ __launch_bounds__(256, 4) void bnact_bwd_reduce8_kernel(const BnActBwdArgs a, long pix_per_block) {
    extern __shared__ double s_acc[];   // [(1+n)*C]
    const int C = a.f.C, G = C >> 3, n = a.f.n;
    const int tid = threadIdx.x;
    const int nacc = (1 + n) * C;
    for (int i = tid; i < nacc; i += 256) s_acc[i] = 0.0;
    __syncthreads();
    const int R = 256 / G;                       // 256 % G == 0 (host)
    const int g = tid % G, prow = tid / G, c0 = g * 8;
    const long p0 = (long)blockIdx.x * pix_per_block;
    const long p1 = p0 + pix_per_block < a.f.npix ? p0 + pix_per_block : a.f.npix;
    // fp32 partial sums of this thread's <= ~100 pixels, folded into the block's double accumulators afterwards
    float sdz[8], sxy[3][8], salpha = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sdz[j] = sxy[0][j] = sxy[1][j] = sxy[2][j] = 0.f;
    // scale_b and the summed shift per channel: LDS (behind the double accumulators), read where they are used
    float* s_k = reinterpret_cast<float*>(s_acc + nacc);      // [4][C]
    for (int ch = tid; ch < C; ch += 256) {
        float sh = 0.f;
        for (int b = 0; b < 3; ++b) {
            s_k[b * C + ch] = (b < n && a.f.scale[b]) ? a.f.scale[b][ch] : 1.f;
            if (b < n && a.f.shift[b]) sh += a.f.shift[b][ch];
        }
        s_k[3 * C + ch] = sh;
    }
    __syncthreads();
    auto cst = [&](int k, float (&v)[8]) {
        const float4 lo = *reinterpret_cast<const float4*>(s_k + k * C + c0), hi = *reinterpret_cast<const float4*>(s_k + k * C + c0 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
        v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    };
    const bool want_alpha = a.dalpha != nullptr;
    for (long p = p0 + prow; p < p1; p += R) {
        uint4 xr[3], gr, rr;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
            xr[b] = ld16(a.f.x[b] + p * a.f.cs[b] + a.f.co[b] + c0);
        }
        gr = ld16(a.dout + p * a.dcs + a.dco + c0);
        if (want_alpha) rr = ld16(a.f.res + p * a.f.rcs + a.f.rco + c0);
        float z[8], go[8], dz[8];
        cst(3, z);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
            float scv[8], xb[8];
            cst(b, scv);
            unpack8(xr[b], xb);
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] += xb[j] * scv[j];
        }
        unpack8(gr, go);
        if (want_alpha) {
            float r[8];
            unpack8(rr, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) salpha += go[j] * r[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dz[j] = go[j] * act_grad(z[j], a.f.act);
            sdz[j] += dz[j];
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
            float xb[8];
            unpack8(xr[b], xb);
#pragma unroll
            for (int j = 0; j < 8; ++j) sxy[b][j] += dz[j] * xb[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        atomicAdd(&s_acc[c0 + j], (double)sdz[j]);
        for (int b = 0; b < n; ++b)
            if (a.mean[b]) atomicAdd(&s_acc[(1 + b) * C + c0 + j], (double)sxy[b][j]);
    }
    __syncthreads();
    for (int i = tid; i < nacc; i += 256) atomicAdd(&a.ws[i], s_acc[i]);
    if (want_alpha) {
        double sa = (double)salpha;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sa += __shfl_xor(sa, o, 64);
        if ((tid & 63) == 0) atomicAdd(&a.ws[nacc], sa);
    }
}

__global__ __launch_bounds__(256, 4) void bnact_bwd_reduce8_part_kernel(const BnActBwdArgs a, long pix_per_block) {
    // atomic-free form of the kernel above: the R pixel-row threads of a channel leave their fp32 sums in LDS and are added in row
    // order (double), the block's sums go to its slot of a.part; bnact_bwd_sums_kernel adds the blocks
    extern __shared__ float s_rowf[];   // [R][1+n][8][G] | scale_b, summed shift [4][C]
    const int C = a.f.C, G = C >> 3, n = a.f.n;
    const int tid = threadIdx.x;
    const int nacc = (1 + n) * C;
    const int R = 256 / G;                       // 256 % G == 0 (host)
    const int g = tid % G, prow = tid / G, c0 = g * 8;
    const long p0 = (long)blockIdx.x * pix_per_block;
    const long p1 = p0 + pix_per_block < a.f.npix ? p0 + pix_per_block : a.f.npix;
    // fp32 partial sums of this thread's <= ~100 pixels, folded into the block's double accumulators afterwards
    float sdz[8], sxy[3][8], salpha = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sdz[j] = sxy[0][j] = sxy[1][j] = sxy[2][j] = 0.f;
    // scale_b and the summed shift per channel: LDS (behind the double accumulators), read where they are used
    float* s_k = s_rowf + (size_t)R * nacc;                   // [4][C]
    for (int ch = tid; ch < C; ch += 256) {
        float sh = 0.f;
        for (int b = 0; b < 3; ++b) {
            s_k[b * C + ch] = (b < n && a.f.scale[b]) ? a.f.scale[b][ch] : 1.f;
            if (b < n && a.f.shift[b]) sh += a.f.shift[b][ch];
        }
        s_k[3 * C + ch] = sh;
    }
    __syncthreads();
    auto cst = [&](int k, float (&v)[8]) {
        const float4 lo = *reinterpret_cast<const float4*>(s_k + k * C + c0), hi = *reinterpret_cast<const float4*>(s_k + k * C + c0 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
        v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    };
    const bool want_alpha = a.dalpha != nullptr;
    for (long p = p0 + prow; p < p1; p += R) {
        uint4 xr[3], gr, rr;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
            xr[b] = ld16(a.f.x[b] + p * a.f.cs[b] + a.f.co[b] + c0);
        }
        gr = ld16(a.dout + p * a.dcs + a.dco + c0);
        if (want_alpha) rr = ld16(a.f.res + p * a.f.rcs + a.f.rco + c0);
        float z[8], go[8], dz[8];
        cst(3, z);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
            float scv[8], xb[8];
            cst(b, scv);
            unpack8(xr[b], xb);
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] += xb[j] * scv[j];
        }
        unpack8(gr, go);
        if (want_alpha) {
            float r[8];
            unpack8(rr, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) salpha += go[j] * r[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dz[j] = go[j] * act_grad(z[j], a.f.act);
            sdz[j] += dz[j];
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
            float xb[8];
            unpack8(xr[b], xb);
#pragma unroll
            for (int j = 0; j < 8; ++j) sxy[b][j] += dz[j] * xb[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s_rowf[((prow * (1 + n) + 0) * 8 + j) * G + g] = sdz[j];
        for (int b = 0; b < n; ++b) s_rowf[((prow * (1 + n) + 1 + b) * 8 + j) * G + g] = a.mean[b] ? sxy[b][j] : 0.f;
    }
    __shared__ double s_al[4];
    {
        double sa = want_alpha ? (double)salpha : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sa += __shfl_xor(sa, o, 64);
        if ((tid & 63) == 0) s_al[tid >> 6] = sa;
    }
    __syncthreads();
    double* dst = a.part + (size_t)blockIdx.x * (size_t)(nacc + 1);
    for (int i = tid; i < nacc; i += 256) {      // i = (k*8 + j)*G + g
        const int gg = i % G, kj = i / G;
        double tot = 0.0;
        for (int r = 0; r < R; ++r) tot += (double)s_rowf[(size_t)r * nacc + i];
        dst[(kj >> 3) * C + gg * 8 + (kj & 7)] = tot;
    }
    if (tid == 0) dst[nacc] = (s_al[0] + s_al[1]) + (s_al[2] + s_al[3]);
}

// Adds the blocks' partials (fixed order), leaves the totals in a.ws for the apply pass and forms dgamma / dbeta / dalpha.
// block = 8 channels x (1 + n) sums; 512 threads = 16 interleaved sub-sums per value
__global__ __launch_bounds__(512) void bnact_bwd_sums_kernel(const BnActBwdArgs a) {
    __shared__ double red[16][33];
    __shared__ double al[512];
    const int C = a.f.C, n = a.f.n, nacc = (1 + n) * C;
    const int t = threadIdx.x, vi = t & 31, prt = t >> 5;
    const int k = vi >> 3, c = blockIdx.x * 8 + (vi & 7);
    const bool live = k <= n && c < C;
    red[prt][vi] = live ? part_subsum(a.part + (size_t)k * C + c, (size_t)(nacc + 1), a.nparts, prt) : 0.0;
    __syncthreads();
    if (t < 32) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += red[q][t];
        red[0][t] = tot;
        if (live) a.ws[k * C + c] = tot;
    }
    __syncthreads();
    if (t < 8 && blockIdx.x * 8 + t < C) {
        const int ch = blockIdx.x * 8 + t;
        for (int b = 0; b < n; ++b)
            if (a.mean[b]) bwd_param_grads(a, b, ch, red[0][t], red[0][(1 + b) * 8 + t]);
    }
    if (blockIdx.x == 0 && a.dalpha) {           // sum dout*res: one value per block partial
        double sa = 0.0;
        for (int b = t; b < a.nparts; b += 512) sa += a.part[(size_t)b * (size_t)(nacc + 1) + nacc];
        al[t] = sa;
        __syncthreads();
        for (int w = 256; w > 0; w >>= 1) {
            if (t < w) al[t] += al[t + w];
            __syncthreads();
        }
        if (t == 0) *a.dalpha += (float)al[0];
    }
}

__global__ __launch_bounds__(256, 4) void bnact_bwd_apply8_kernel(const BnActBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_c[];   // [13][C]: sc_b | k1_b | A_b | B_b (b = 0..2) | sum of shifts
    const int C = a.f.C, G = C >> 3, n = a.f.n;
    const int tid = threadIdx.x;
    {   // the per-channel constants of load_bwd_consts4, once per block
        const double invN = 1.0 / (double)a.f.npix;
        for (int ch = tid; ch < C; ch += 256) {
            float sh = 0.f;
            for (int b = 0; b < 3; ++b) {
                float scv = 1.f, k1 = 1.f, A = 0.f, Bc = 0.f;
                if (b < n) {
                    scv = a.f.scale[b] ? a.f.scale[b][ch] : 1.f;
                    sh += a.f.shift[b] ? a.f.shift[b][ch] : 0.f;
                    if (a.mean[b]) {
                        const double s0 = a.ws[ch], sb = a.ws[(1 + b) * C + ch];
                        const float mu = a.mean[b][ch], is = a.invstd[b][ch];
                        k1 = (a.gamma[b] ? a.gamma[b][ch] : 1.f) * is;
                        const float m2 = (float)((sb - (double)mu * s0) * (double)is * invN);
                        A = -k1 * m2 * is;
                        Bc = -k1 * (float)(s0 * invN) - A * mu;
                    } else {
                        k1 = scv;
                    }
                }
                s_c[(0 + b) * C + ch] = scv;
                s_c[(3 + b) * C + ch] = k1;
                s_c[(6 + b) * C + ch] = A;
                s_c[(9 + b) * C + ch] = Bc;
            }
            s_c[12 * C + ch] = sh;
        }
    }
    __syncthreads();
    const float alpha = a.f.res ? (a.f.alpha ? *a.f.alpha : 1.f) : 0.f;
    const int g = tid % G, rpb = 256 / G, c0 = g * 8;
    auto cst = [&](int k, float (&v)[8]) {
        const float4 lo = *reinterpret_cast<const float4*>(s_c + k * C + c0), hi = *reinterpret_cast<const float4*>(s_c + k * C + c0 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
        v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    };
    for (long p = (long)blockIdx.x * rpb + tid / G; p < a.f.npix; p += (long)gridDim.x * rpb) {
        // every load of this pixel group first
        uint4 xr[3], oldr[3], gr, rold;
        long qd[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
            xr[b] = ld16(a.f.x[b] + p * a.f.cs[b] + a.f.co[b] + c0);
            if (a.dx[b]) {
                qd[b] = bwd_dx_index(a, b, p);
                if (a.xacc[b]) oldr[b] = ld16(a.dx[b] + qd[b] * a.xcs[b] + a.xco[b] + c0);
            }
        }
        gr = ld16(a.dout + p * a.dcs + a.dco + c0);
        if (a.dres && a.racc) rold = ld16(a.dres + p * a.rcs + a.rco + c0);
        float z[8], go[8], dz[8], xb[3][8];
        cst(12, z);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
            float scv[8];
            cst(b, scv);
            unpack8(xr[b], xb[b]);
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] += xb[b][j] * scv[j];
        }
        unpack8(gr, go);
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] = go[j] * act_grad(z[j], a.f.act);
        if (a.dres) {
            float r[8];
            if (a.racc) unpack8(rold, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (a.racc ? r[j] : 0.f) + alpha * go[j];
            store8(a.dres + p * a.rcs + a.rco + c0, r);
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= n) break;
            if (!a.dx[b]) continue;
            float k1[8], A[8], Bc[8], d[8];
            cst(3 + b, k1);
            cst(6 + b, A);
            cst(9 + b, Bc);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = k1[j] * dz[j] + A[j] * xb[b][j] + Bc[j];
            if (a.xacc[b]) {
                float old[8];
                unpack8(oldr[b], old);
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] += old[j];
            }
            store8(a.dx[b] + qd[b] * a.xcs[b] + a.xco[b] + c0, d);
        }
    }
}

// runs after the apply pass: parameter gradients, then the workspace is zeroed for the next launch on it (see
// bn_train_finalize_kernel)
__global__ void bnact_bwd_params_kernel(const BnActBwdArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int C = a.f.C;
    if (c == 0) {
        if (a.dalpha) *a.dalpha += (float)a.ws[(1 + a.f.n) * C];
        a.ws[(1 + a.f.n) * C] = 0.0;
    }
    if (c >= C) return;
    const double s0 = a.ws[c];
    for (int b = 0; b < a.f.n; ++b) {
        if (a.mean[b]) bwd_param_grads(a, b, c, s0, a.ws[(1 + b) * C + c]);
        a.ws[(1 + b) * C + c] = 0.0;
    }
    a.ws[c] = 0.0;
}

int bnact_backward_launch(const y6_bnact_bwd_desc* d, hipStream_t s) {
    Y6_REQUIRE(d && d->workspace, "bnact_backward: null argument");
    BnActBwdArgs a;
    memset(&a, 0, sizeof(a));
    int rc = fill_bnact_args(&d->fwd, &a.f, false);
    if (rc) return rc;
    const y6_tensor& ref = d->fwd.x[0];
    const int C = ref.C, n = d->fwd.n;
    Y6_REQUIRE(C <= 1024, "bnact_backward: at most 1024 channels");
    Y6_REQUIRE(d->workspace_bytes >= y6_bnact_bwd_workspace_bytes_for(C, (long)ref.B * ref.H * ref.W), "bnact_backward: workspace too small");
    Y6_REQUIRE(view_ok(d->dout) && same_shape(d->dout, ref), "bnact_backward: bad dout view");
    a.dout = (const __half*)d->dout.data;
    a.dcs = d->dout.cstride;
    a.dco = d->dout.coff;
    a.H = ref.H;
    a.W = ref.W;
    for (int b = 0; b < n; ++b) {
        a.mean[b] = d->mean[b];
        a.invstd[b] = d->invstd[b];
        a.gamma[b] = d->gamma[b];
        Y6_REQUIRE((d->mean[b] == nullptr) == (d->invstd[b] == nullptr), "bnact_backward: branch %d needs mean AND invstd", b);
        Y6_REQUIRE((((uintptr_t)d->mean[b] | (uintptr_t)d->invstd[b] | (uintptr_t)d->gamma[b]) & 15) == 0,
                   "bnact_backward: statistics must be 16-byte aligned");
        a.dgamma[b] = d->dgamma[b];
        a.dbeta[b] = d->dbeta[b];
        const y6_tensor& t = d->dx[b];
        if (!t.data) continue;
        const int dil = d->dx_dil[b] == 2 ? 2 : 1;
        Y6_REQUIRE(view_ok(t) && t.C == C && t.B == ref.B, "bnact_backward: bad dx view of branch %d", b);
        if (dil == 1)
            Y6_REQUIRE(t.H == ref.H && t.W == ref.W, "bnact_backward: dx shape mismatch of branch %d", b);
        else
            Y6_REQUIRE(t.H >= 2 * ref.H - 1 && t.W >= 2 * ref.W - 1, "bnact_backward: dilated dx of branch %d is too small", b);
        a.dx[b] = (__half*)t.data;
        a.xcs[b] = t.cstride;
        a.xco[b] = t.coff;
        a.xdil[b] = dil;
        a.xacc[b] = d->dx_acc[b];
        a.xH[b] = t.H;
        a.xW[b] = t.W;
    }
    if (d->dres.data) {
        Y6_REQUIRE(a.f.res && view_ok(d->dres) && same_shape(d->dres, ref), "bnact_backward: bad dres view");
        a.dres = (__half*)d->dres.data;
        a.rcs = d->dres.cstride;
        a.rco = d->dres.coff;
        a.racc = d->dres_acc;
    }
    a.dalpha = (a.f.res && d->dalpha) ? d->dalpha : nullptr;
    a.ws = (double*)d->workspace;
    const size_t nacc = (size_t)(1 + n) * C + 1;
    if (!d->workspace_clean) Y6_HIP(hipMemsetAsync(a.ws, 0, nacc * sizeof(double), s));
    const int G = C / 8;
    // v2 (8 channels per thread, 16-byte accesses): every view 16-byte aligned, a thread keeps one channel group
    static const bool no_v2 = getenv("Y6_BNACT_BWD_V1") != nullptr;      // A/B switch
    bool v2 = !no_v2 && C % 8 == 0 && 256 % G == 0 && (size_t)13 * C * sizeof(float) <= 64 * 1024;
    auto al16 = [](const void* ptr, int cs, int co) { return ptr == nullptr || ((((uintptr_t)ptr) & 15) == 0 && cs % 8 == 0 && co % 8 == 0); };
    for (int b = 0; b < n; ++b) v2 = v2 && al16(a.f.x[b], a.f.cs[b], a.f.co[b]) && al16(a.dx[b], a.xcs[b], a.xco[b]);
    v2 = v2 && al16(a.dout, a.dcs, a.dco) && al16(a.f.res, a.f.rcs, a.f.rco) && al16(a.dres, a.rcs, a.rco);
    const int R = v2 ? 256 / G : 256 / (C / 4);
    long ppb = (long)R * 32;
    long blocks = (a.f.npix + ppb - 1) / ppb;
    if (blocks > 2048) {
        blocks = 2048;
        ppb = (a.f.npix + blocks - 1) / blocks;
    }
    if (v2 && !bn_use_atomics()) {
        // atomic-free form: short per-thread loops (8 pixel rows), block partials, ordered sums (see bn_sum_part_kernel)
        ppb = (long)R * 8;
        blocks = (a.f.npix + ppb - 1) / ppb;
        if (blocks > kBnPartBlocks) {
            blocks = kBnPartBlocks;
            ppb = (a.f.npix + blocks - 1) / blocks;
            blocks = (a.f.npix + ppb - 1) / ppb;
        }
        a.part = a.ws + nacc;
        a.nparts = (int)blocks;
        hipLaunchKernelGGL(bnact_bwd_reduce8_part_kernel, dim3((unsigned)blocks), dim3(256),
                           (size_t)R * (1 + n) * C * sizeof(float) + (size_t)4 * C * sizeof(float), s, a, ppb);
        Y6_LAUNCH_CHECK();
        hipLaunchKernelGGL(bnact_bwd_sums_kernel, dim3((unsigned)((C + 7) / 8)), dim3(512), 0, s, a);
        Y6_LAUNCH_CHECK();
        hipLaunchKernelGGL(bnact_bwd_apply8_kernel, dim3(grid_for((size_t)a.f.npix * G, 256, 256 * 16)), dim3(256), (size_t)13 * C * sizeof(float), s, a);
        Y6_LAUNCH_CHECK();
        return Y6_OK;
    }
    if (v2) {
        hipLaunchKernelGGL(bnact_bwd_reduce8_kernel, dim3((unsigned)blocks), dim3(256), (size_t)(1 + n) * C * sizeof(double) + (size_t)4 * C * sizeof(float), s, a, ppb);
        Y6_LAUNCH_CHECK();
        hipLaunchKernelGGL(bnact_bwd_apply8_kernel, dim3(grid_for((size_t)a.f.npix * G, 256, 256 * 16)), dim3(256), (size_t)13 * C * sizeof(float), s, a);
        Y6_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(bnact_bwd_reduce_kernel, dim3((unsigned)blocks), dim3(256), (size_t)(1 + n) * C * sizeof(double), s, a, ppb);
        Y6_LAUNCH_CHECK();
        hipLaunchKernelGGL(bnact_bwd_apply_kernel, dim3(grid_for((size_t)a.f.npix * G * 2, 256, 256 * 16)), dim3(256), 0, s, a);
        Y6_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bnact_bwd_params_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ------------------------------------------------------------------ pixel-run-major sampling for the weight gradient
// dst[b][r][q/8][c][q%8]: 8 consecutive columns of one channel form a 16-byte run (an MFMA operand of one lane), and the runs
// of consecutive channels are contiguous, so a wave's fragment load is 512 contiguous bytes per half (4 full cache lines).
// (A [c][b][r][q] layout made every lane of a load touch its own 128-byte line, 16 bytes at a time: the L1 thrashed and
// the big layers ran at 70 TFLOP/s.)
// block = (b, r, 64-column tile, 64-channel tile): NHWC rows -> LDS -> runs
__global__ __launch_bounds__(256) void wt_nhwc_kernel(const __half* __restrict__ src, int cs, int co, int B, int H, int W, int C,
                                                      int sy, int sx, int oy, int ox, int R, int Q, __half* __restrict__ dst,
                                                      int qtiles, int ctiles) {
    __shared__ __half tile[64][72];
    int bid = blockIdx.x;
    const int qt = bid % qtiles;
    bid /= qtiles;
    const int ct = bid % ctiles;
    bid /= ctiles;
    const int r = bid % R;
    const int b = bid / R;
    const int tid = threadIdx.x;
    const int y = r * sy + oy;
    const int q0 = qt * 64, c0 = ct * 64;
    const bool row_ok = y >= 0 && y < H;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int ql = (tid >> 3) + it * 32, cg = tid & 7;
        const int x = (q0 + ql) * sx + ox;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row_ok && x >= 0 && x < W && q0 + ql < Q && c0 + cg * 8 < C)
            v = *reinterpret_cast<const uint4*>(src + ((size_t)(b * H + y) * W + x) * cs + co + c0 + cg * 8);
        *reinterpret_cast<uint4*>(&tile[ql][cg * 8]) = v;
    }
    __syncthreads();
    // destination layout [b][r][run = q/8][c][8]: the 32 channels of an MFMA fragment are 512 contiguous bytes
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * 256;          // 8 runs x 64 channels, channel fastest (1 KiB contiguous per 64 threads)
        const int cl = item & 63, run = item >> 6;
        if (c0 + cl >= C || q0 + run * 8 >= Q) continue;
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = *reinterpret_cast<const _Float16*>(&tile[run * 8 + j][cl]);
        *reinterpret_cast<h8_t*>(dst + ((((size_t)b * R + r) * (Q >> 3) + (q0 >> 3) + run) * C + c0 + cl) * 8) = o;
    }
}

template <typename T>
__global__ void wt_nchw_kernel(const T* __restrict__ src, int B, int H, int W, int C, int sy, int sx, int oy, int ox, int R, int Q,
                               __half* __restrict__ dst) {
    const size_t total = (size_t)C * B * R * (Q / 8);        // destination [b][r][run][c][8]
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t t = i / C;
        const int run = (int)(t % (Q / 8));
        t /= (Q / 8);
        const int r = (int)(t % R);
        const int b = (int)(t / R);
        const int y = r * sy + oy;
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = (run * 8 + j) * sx + ox;
            float v = 0.f;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                if constexpr (sizeof(T) == 1) v = (float)src[(((size_t)b * C + c) * H + y) * W + x] / 255.f;   // uint8 pixels -> u/255
                else v = (float)src[(((size_t)b * C + c) * H + y) * W + x];
            }
            o[j] = (_Float16)v;
        }
        *reinterpret_cast<h8_t*>(dst + i * 8) = o;
    }
}

int wgrad_transpose_launch(const y6_wgrad_t_desc* d, hipStream_t s) {
    Y6_REQUIRE(d && d->src.data && d->dst, "wgrad_transpose: null argument");
    Y6_REQUIRE(d->R > 0 && d->Q > 0 && d->Q % 16 == 0 && d->sy >= 1 && d->sx >= 1, "wgrad_transpose: bad geometry");
    Y6_REQUIRE(((uintptr_t)d->dst & 15) == 0, "wgrad_transpose: dst must be 16-byte aligned");
    const y6_tensor& t = d->src;
    if (d->nchw) {
        const size_t total = (size_t)t.C * t.B * d->R * (d->Q / 8);
        if (d->src_dtype == Y6_F16)
            hipLaunchKernelGGL(wt_nchw_kernel<__half>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const __half*)t.data, t.B, t.H,
                               t.W, t.C, d->sy, d->sx, d->oy, d->ox, d->R, d->Q, (__half*)d->dst);
        else if (d->src_dtype == Y6_F32)
            hipLaunchKernelGGL(wt_nchw_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float*)t.data, t.B, t.H,
                               t.W, t.C, d->sy, d->sx, d->oy, d->ox, d->R, d->Q, (__half*)d->dst);
        else if (d->src_dtype == Y6_U8)
            hipLaunchKernelGGL(wt_nchw_kernel<uint8_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const uint8_t*)t.data, t.B, t.H,
                               t.W, t.C, d->sy, d->sx, d->oy, d->ox, d->R, d->Q, (__half*)d->dst);
        else
            Y6_REQUIRE(false, "wgrad_transpose: bad dtype %d", d->src_dtype);
        Y6_LAUNCH_CHECK();
        return Y6_OK;
    }
    Y6_REQUIRE(view_ok(t), "wgrad_transpose: the source must be an fp16 NHWC view with 8-channel alignment");
    const int qtiles = (d->Q + 63) / 64, ctiles = (t.C + 63) / 64;
    const size_t blocks = (size_t)t.B * d->R * qtiles * ctiles;
    Y6_REQUIRE(blocks < (1ull << 31), "wgrad_transpose: grid too large");
    hipLaunchKernelGGL(wt_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const __half*)t.data, t.cstride, t.coff, t.B, t.H,
                       t.W, t.C, d->sy, d->sx, d->oy, d->ox, d->R, d->Q, (__half*)d->dst, qtiles, ctiles);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ------------------------------------------------------------------ batched weight packing
// packed layout as y6_pack_conv_weight: dst[cfr][chunk][tap][ks][lane][j] = W'[o = cfr*32 + (lane&31)][i = chunk*32 + ks*16 + (lane>>5)*8 + j][tap]
__device__ __forceinline__ void pack_one_element(const y6_pack_job* __restrict__ jobs, int njobs, uint64_t i) {
    {
        int lo = 0, hi = njobs - 1;                 // last job with first <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first <= i) lo = mid; else hi = mid - 1;
        }
        const y6_pack_job jb = jobs[lo];
        const uint64_t e = i - jb.first;
        if (jb.kind == 4) {       // fp32 [Cout][Cin][3][3] with the 1x1 kernel at the centre tap (the stem's 1x1 stride-2 branch)
            const uint64_t oi = e / 9;
            reinterpret_cast<float*>(jb.dst)[e] = (e - oi * 9 == 4) ? jb.src[oi] : 0.f;
            return;
        }
        const int K = jb.K, NT = K * K;
        if (jb.kind == 2 && (jb.Cout % 32) != 0) {    // un-fused ConvTranspose2d image: four separately padded 1x1 weights
            const int nchunk2 = (jb.Cin + 31) / 32;
            const uint64_t per = (uint64_t)((((jb.Cout + 31) / 32 + 3) / 4) * 4) * nchunk2 * 1024;
            const int sub = (int)(e / per);
            const uint64_t ii = e - (uint64_t)sub * per;
            const int j2 = (int)(ii & 7), lane2 = (int)((ii >> 3) & 63), ks2 = (int)((ii >> 9) & 1);
            const uint64_t r2 = ii >> 10;
            const int chunk2 = (int)(r2 % nchunk2), cfr2 = (int)(r2 / nchunk2);
            const int co = cfr2 * 32 + (lane2 & 31), ci = chunk2 * 32 + ks2 * 16 + (lane2 >> 5) * 8 + j2;
            float v2 = 0.f;
            if (co < jb.Cout && ci < jb.Cin) v2 = jb.src[((size_t)ci * jb.Cout + co) * 4 + sub];
            reinterpret_cast<__half*>(jb.dst)[e] = __float2half(v2);
            return;
        }
        // logical output-channel / input-channel counts of the packed matrix
        int O = jb.Cout, I = jb.Cin;
        if (jb.kind == 1) { O = jb.Cin; I = jb.Cout; }
        if (jb.kind == 2) { O = 4 * jb.Cout; I = jb.Cin; }          // rows = sub*Cout + co
        if (jb.kind == 3) { O = jb.Cin; I = 4 * jb.Cout; }          // cols = sub*Cout + co
        const int nchunk = (I + 31) / 32;
        const int nt = (jb.kind >= 2) ? 1 : NT;
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), ks = (int)((e >> 9) & 1);
        uint64_t r = e >> 10;
        const int tap = (int)(r % nt);
        r /= nt;
        const int chunk = (int)(r % nchunk);
        const int cfr = (int)(r / nchunk);
        const int o = cfr * 32 + (lane & 31);
        const int ic = chunk * 32 + ks * 16 + (lane >> 5) * 8 + j;
        float v = 0.f;
        if (o < O && ic < I) {
            if (jb.kind == 0) {
                v = jb.src[((size_t)o * jb.Cin + ic) * NT + tap];
            } else if (jb.kind == 1) {                 // W'[ci=o][co=ic][tap] = W[co][ci][NT-1-tap]
                v = jb.src[((size_t)ic * jb.Cin + o) * NT + (NT - 1 - tap)];
            } else if (jb.kind == 2) {                 // IOHW [Cin][Cout][2][2]: row = sub*Cout + co
                const int sub = o / jb.Cout, co = o - sub * jb.Cout;
                v = jb.src[((size_t)ic * jb.Cout + co) * 4 + sub];
            } else {                                   // W'[ci=o][sub*Cout+co = ic]
                const int sub = ic / jb.Cout, co = ic - sub * jb.Cout;
                v = jb.src[((size_t)o * jb.Cout + co) * 4 + sub];
            }
        }
        reinterpret_cast<__half*>(jb.dst)[e] = __float2half(v);
    }
}


// (round 6, last session) one thread per EIGHT packed elements: the eight consecutive input channels of one (cout, tap) that form a
// 16-byte run of the packed image - one job search, one index decode and one 16-byte store instead of eight (the per-element form
// was 37 M threads-iterations with a binary search each: 0.32 ms per training step at 0.8 TB/s).  Groups that are not a whole run of
// a conv / convT job (the fp32 centre-tap image of kind 4, the un-fused ConvTranspose2d image, a job boundary) take the
// per-element code.
__global__ __launch_bounds__(256) void pack_batch_kernel(const y6_pack_job* __restrict__ jobs, int njobs, uint64_t total, int pack_tap_fastest) {
    const uint64_t groups = (total + 7) >> 3;
    for (uint64_t g8 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g8 < groups; g8 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = g8 << 3;
        int lo = 0, hi = njobs - 1;                 // last job with first <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first <= i) lo = mid; else hi = mid - 1;
        }
        const y6_pack_job jb = jobs[lo];
        const uint64_t e = i - jb.first;
        const uint64_t end = lo + 1 < njobs ? jobs[lo + 1].first : total;
        const bool fused_t = jb.kind != 2 || (jb.Cout % 32) == 0;
        if (jb.kind == 4 || !fused_t || (e & 7) != 0 || i + 8 > end) {
            for (uint64_t k = i; k < i + 8 && k < total; ++k) pack_one_element(jobs, njobs, k);
            continue;
        }
        const int K = jb.K, NT = K * K;
        int O = jb.Cout, I = jb.Cin;
        if (jb.kind == 1) { O = jb.Cin; I = jb.Cout; }
        if (jb.kind == 2) { O = 4 * jb.Cout; I = jb.Cin; }
        if (jb.kind == 3) { O = jb.Cin; I = 4 * jb.Cout; }
        const int nchunk = (I + 31) / 32;
        const int nt = (jb.kind >= 2) ? 1 : NT;
        // Which 16-byte run of the packed image this thread makes.  For the nine-tap jobs the tap runs FASTEST over the threads
        // (a permutation of the job's runs): the nine taps of a (cout, cin) pair are 36 consecutive bytes of the OIHW source, so
        // neighbouring threads now read the same cache lines - in packed order (tap above lane and k-step) every line of the
        // source went through nine different waves (0.155 ms per training step for 150 MB of algorithmic traffic).
        int lane, ks, tap;
        uint64_t r, edst = e;
        if (nt == 9 && jb.kind <= 1 && pack_tap_fastest) {
            uint64_t q = e >> 3;
            tap = (int)(q % 9);
            q /= 9;
            lane = (int)(q & 63);
            ks = (int)((q >> 6) & 1);
            r = q >> 7;
            edst = (((r * 9 + (uint64_t)tap) * 2 + (uint64_t)ks) * 64 + (uint64_t)lane) * 8;
        } else {
            lane = (int)((e >> 3) & 63);
            ks = (int)((e >> 9) & 1);
            r = e >> 10;
            tap = (int)(r % nt);
            r /= nt;
        }
        const int chunk = (int)(r % nchunk);
        const int cfr = (int)(r / nchunk);
        const int o = cfr * 32 + (lane & 31);
        const int ic0 = chunk * 32 + ks * 16 + (lane >> 5) * 8;
        h8_t out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ic = ic0 + j;
            float v = 0.f;
            if (o < O && ic < I) {
                if (jb.kind == 0) {
                    v = jb.src[((size_t)o * jb.Cin + ic) * NT + tap];
                } else if (jb.kind == 1) {
                    v = jb.src[((size_t)ic * jb.Cin + o) * NT + (NT - 1 - tap)];
                } else if (jb.kind == 2) {
                    const int sub = o / jb.Cout, co = o - sub * jb.Cout;
                    v = jb.src[((size_t)ic * jb.Cout + co) * 4 + sub];
                } else {
                    const int sub = ic / jb.Cout, co = ic - sub * jb.Cout;
                    v = jb.src[((size_t)o * jb.Cout + co) * 4 + sub];
                }
            }
            out[j] = (_Float16)v;
        }
        *reinterpret_cast<h8_t*>(reinterpret_cast<__half*>(jb.dst) + edst) = out;
    }
}

int pack_batch_launch(const y6_pack_batch_desc* d, hipStream_t s) {
    Y6_REQUIRE(d && d->jobs && d->njobs > 0 && d->total > 0, "pack_weights_batched: bad arguments");
    static const bool tap_fastest = !(getenv("Y6_PACK_TAPFAST") && atoi(getenv("Y6_PACK_TAPFAST")) == 0);     // A/B switch
    hipLaunchKernelGGL(pack_batch_kernel, dim3(grid_for(((size_t)d->total + 7) / 8, 256, 256 * 64)), dim3(256), 0, s, d->jobs, d->njobs, d->total, (int)tap_fastest);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ------------------------------------------------------------------ SPPF pools backward
// block = (image, 8-channel group); planes in LDS; three scatter passes.
// Deterministic: two outputs can send their gradient to the same input only if their 5x5 windows overlap, i.e. if they are less
// than 5 apart in y and in x.  The outputs are visited in 25 rounds by (y mod 5, x mod 5): inside a round no two outputs share a
// window, so the adds need no atomics, and an input receives its contributions round by round - in the same order on every run.
// (Until r04 this was a scatter with LDS float atomics: the order of the adds, and with it the last bits of d x, depended on
// how the waves of the block happened to be scheduled.)
__device__ __forceinline__ void pool_scatter(const float* __restrict__ in, const float* __restrict__ gout, float* __restrict__ gin,
                                             int H, int W, int tid, int nthr) {
    // in/gout/gin: [H*W][8] in LDS; gin += scatter(gout) to the first maximum of each 5x5 window of `in`
    const int ny = (H + 4) / 5, nx = (W + 4) / 5;
    for (int round = 0; round < 25; ++round) {
        const int cy = round / 5, cx = round - cy * 5;
        for (int i = tid; i < ny * nx * 8; i += nthr) {
            const int c = i & 7, q = i >> 3;
            const int y = (q / nx) * 5 + cy, x = (q % nx) * 5 + cx;
            if (y >= H || x >= W) continue;
            const int p = y * W + x;
            const float g = gout[p * 8 + c];
            if (g == 0.f) continue;
            float best = -INFINITY;
            int arg = p;
            bool found = false;
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -2; dx <= 2; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    const float v = in[(yy * W + xx) * 8 + c];
                    if (!found || v > best) {
                        best = v;
                        arg = yy * W + xx;
                        found = true;
                    }
                }
            }
            gin[arg * 8 + c] += g;
        }
        __syncthreads();
    }
}

struct SppfBwdArgs {
    const __half *x, *y1, *y2, *dy1, *dy2, *dy3;
    __half* dx;
    int cs[7], co[7];    // x, y1, y2, dy1, dy2, dy3, dx
    int H, W, C, acc;
};

__global__ __launch_bounds__(256) void sppf_bwd_kernel(const SppfBwdArgs a) {
    extern __shared__ float sm[];
    const int HW = a.H * a.W, G = a.C >> 3;
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    float* vin = sm;               // [HW][8] forward plane of the current stage
    float* gA = sm + HW * 8;       // gradient arriving at the stage's output
    float* gB = sm + 2 * HW * 8;   // gradient wrt the stage's input (accumulated)
    const int tid = threadIdx.x;
    auto load_plane = [&](const __half* base, int cs, int co, float* dst, bool add) {
        for (int p = tid; p < HW; p += 256) {
            float v[8];
            load8(base + ((size_t)b * HW + p) * cs + co + g * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[p * 8 + j] = add ? dst[p * 8 + j] + v[j] : v[j];
        }
    };
    // stage 3: y2 -> y3
    load_plane(a.y2, a.cs[2], a.co[2], vin, false);
    load_plane(a.dy3, a.cs[5], a.co[5], gA, false);
    load_plane(a.dy2, a.cs[4], a.co[4], gB, false);       // gradient that reached y2 directly
    __syncthreads();
    pool_scatter(vin, gA, gB, a.H, a.W, tid, 256);
    __syncthreads();
    // stage 2: y1 -> y2 ; gout = gB
    load_plane(a.y1, a.cs[1], a.co[1], vin, false);
    load_plane(a.dy1, a.cs[3], a.co[3], gA, false);
    __syncthreads();
    pool_scatter(vin, gB, gA, a.H, a.W, tid, 256);        // gA now holds d y1 total
    __syncthreads();
    // stage 1: x -> y1 ; gout = gA, result into gB
    load_plane(a.x, a.cs[0], a.co[0], vin, false);
    for (int i = tid; i < HW * 8; i += 256) gB[i] = 0.f;
    __syncthreads();
    if (a.acc) load_plane(a.dx, a.cs[6], a.co[6], gB, true);
    __syncthreads();
    pool_scatter(vin, gA, gB, a.H, a.W, tid, 256);
    __syncthreads();
    for (int p = tid; p < HW; p += 256) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gB[p * 8 + j];
        store8(a.dx + ((size_t)b * HW + p) * a.cs[6] + a.co[6] + g * 8, v);
    }
}

// Second form (round 6): the argmax of every window is found ONCE, separably - the leftmost maximum of each 5-wide row piece, then
// the first row that holds the window's maximum: torch's first maximum in row-major order - and kept as a 16-bit pixel index; then
// every INPUT pixel gathers from the <= 25 outputs whose window holds it, in row-major order of the outputs.  No rounds (the
// scatter form above walks 25 rounds x 3 stages with half of the block idle and 25 dependent LDS reads per output: 0.41 ms for
// the 13 MB tensors of a YOLOv6-S step), no atomics, the same additions in the same order on every run.
// block = (image, 8-channel group), 256 threads, a thread item = (pixel, channel pair): consecutive lanes touch consecutive words.
struct SppfLds {
    _Float16* vin;            // [HW][8] forward plane of the stage's input
    _Float16* rmv;            // [HW][8] maximum of the row piece x-2 .. x+2
    unsigned short* rmi;      // [HW][8] ... and the x of its leftmost occurrence
    unsigned short* arg;      // [HW][8] pixel index of the window's first maximum
    float *gA, *gB;           // [HW][8] gradients
};

__device__ __forceinline__ void pool_gather(const SppfLds& L, const float* __restrict__ gout, float* __restrict__ gin, int H, int W,
                                            int tid) {
    // branch-free on purpose: every LDS read of an item is unconditional (clamped index, the result masked), so the 5 / 5 / 50 reads
    // of the three passes go out back to back instead of one round trip per neighbour
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int n = H * W * 4;
    for (int i = tid; i < n; i += 256) {                     // row maxima (strict >: the leftmost one; values are finite)
        const int cp = i & 3, p = i >> 2;
        const int y = p / W, x = p - y * W;
        float b0 = -INFINITY, b1 = -INFINITY;
        int i0 = x, i1 = x;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx;
            const bool ok = (unsigned)xx < (unsigned)W;
            const int xc = ok ? xx : x;
            const h2 v = *reinterpret_cast<const h2*>(L.vin + (y * W + xc) * 8 + cp * 2);
            const float v0 = (float)v[0], v1 = (float)v[1];
            const bool t0 = ok && v0 > b0, t1 = ok && v1 > b1;
            b0 = t0 ? v0 : b0;
            i0 = t0 ? xx : i0;
            b1 = t1 ? v1 : b1;
            i1 = t1 ? xx : i1;
        }
        h2 o;
        o[0] = (_Float16)b0;
        o[1] = (_Float16)b1;
        *reinterpret_cast<h2*>(L.rmv + p * 8 + cp * 2) = o;
        *reinterpret_cast<unsigned*>(L.rmi + p * 8 + cp * 2) = (unsigned)i0 | ((unsigned)i1 << 16);
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {                     // first row holding the window maximum -> argmax pixel
        const int cp = i & 3, p = i >> 2;
        const int y = p / W, x = p - y * W;
        float b0 = -INFINITY, b1 = -INFINITY;
        unsigned a0 = p, a1 = p;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = y + dy;
            const bool ok = (unsigned)yy < (unsigned)H;
            const int yc = ok ? yy : y;
            const h2 v = *reinterpret_cast<const h2*>(L.rmv + (yc * W + x) * 8 + cp * 2);
            const unsigned xi = *reinterpret_cast<const unsigned*>(L.rmi + (yc * W + x) * 8 + cp * 2);
            const float v0 = (float)v[0], v1 = (float)v[1];
            const bool t0 = ok && v0 > b0, t1 = ok && v1 > b1;
            b0 = t0 ? v0 : b0;
            a0 = t0 ? (unsigned)yc * W + (xi & 0xffffu) : a0;
            b1 = t1 ? v1 : b1;
            a1 = t1 ? (unsigned)yc * W + (xi >> 16) : a1;
        }
        *reinterpret_cast<unsigned*>(L.arg + p * 8 + cp * 2) = a0 | (a1 << 16);
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {                     // gin[q] += the gradients of the outputs whose argmax is q, row-major
        const int cp = i & 3, q = i >> 2;
        const int y = q / W, x = q - y * W;
        float a0 = gin[q * 8 + cp * 2], a1 = gin[q * 8 + cp * 2 + 1];
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = y + dy;
            const bool oky = (unsigned)yy < (unsigned)H;
            const int yc = oky ? yy : y;
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                const bool ok = oky && (unsigned)xx < (unsigned)W;
                const int pc = yc * W + (((unsigned)xx < (unsigned)W) ? xx : x);
                const unsigned ar = *reinterpret_cast<const unsigned*>(L.arg + pc * 8 + cp * 2);
                const float2 gv = *reinterpret_cast<const float2*>(gout + pc * 8 + cp * 2);
                a0 += (ok && (int)(ar & 0xffffu) == q) ? gv.x : 0.f;
                a1 += (ok && (int)(ar >> 16) == q) ? gv.y : 0.f;
            }
        }
        *reinterpret_cast<float2*>(gin + q * 8 + cp * 2) = make_float2(a0, a1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void sppf_bwd2_kernel(const SppfBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char sm2[];
    const int HW = a.H * a.W, G = a.C >> 3;
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    SppfLds L;
    L.gA = reinterpret_cast<float*>(sm2);
    L.gB = L.gA + HW * 8;
    L.vin = reinterpret_cast<_Float16*>(L.gB + HW * 8);
    L.rmv = L.vin + HW * 8;
    L.rmi = reinterpret_cast<unsigned short*>(L.rmv + HW * 8);
    L.arg = L.rmi + HW * 8;
    const int tid = threadIdx.x;
    auto load_f = [&](const __half* base, int cs, int co, float* dst, bool add) {
        for (int p = tid; p < HW; p += 256) {
            float v[8];
            load8(base + ((size_t)b * HW + p) * cs + co + g * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[p * 8 + j] = add ? dst[p * 8 + j] + v[j] : v[j];
        }
    };
    auto load_h = [&](const __half* base, int cs, int co) {
        for (int p = tid; p < HW; p += 256)
            *reinterpret_cast<uint4*>(L.vin + p * 8) = *reinterpret_cast<const uint4*>(base + ((size_t)b * HW + p) * cs + co + g * 8);
    };
    // stage 3: y2 -> y3
    load_h(a.y2, a.cs[2], a.co[2]);
    load_f(a.dy3, a.cs[5], a.co[5], L.gA, false);
    load_f(a.dy2, a.cs[4], a.co[4], L.gB, false);        // gradient that reached y2 directly
    __syncthreads();
    pool_gather(L, L.gA, L.gB, a.H, a.W, tid);
    // stage 2: y1 -> y2 ; gB holds d y2
    load_h(a.y1, a.cs[1], a.co[1]);
    load_f(a.dy1, a.cs[3], a.co[3], L.gA, false);
    __syncthreads();
    pool_gather(L, L.gB, L.gA, a.H, a.W, tid);           // gA now holds d y1
    // stage 1: x -> y1 ; result into gB
    load_h(a.x, a.cs[0], a.co[0]);
    if (a.acc) load_f(a.dx, a.cs[6], a.co[6], L.gB, false);
    else
        for (int i = tid; i < HW * 8; i += 256) L.gB[i] = 0.f;
    __syncthreads();
    pool_gather(L, L.gA, L.gB, a.H, a.W, tid);
    for (int p = tid; p < HW; p += 256) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = L.gB[p * 8 + j];
        store8(a.dx + ((size_t)b * HW + p) * a.cs[6] + a.co[6] + g * 8, v);
    }
}

int sppf_backward_launch(const y6_sppf_bwd_desc* d, hipStream_t s) {
    Y6_REQUIRE(d, "sppf_pool_backward: null argument");
    const y6_tensor* t[7] = {&d->x, &d->y1, &d->y2, &d->dy1, &d->dy2, &d->dy3, &d->dx};
    SppfBwdArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 7; ++i) {
        Y6_REQUIRE(view_ok(*t[i]) && same_shape(*t[i], d->x), "sppf_pool_backward: tensor %d is not a matching fp16 NHWC view", i);
        a.cs[i] = t[i]->cstride;
        a.co[i] = t[i]->coff;
    }
    a.x = (const __half*)d->x.data;
    a.y1 = (const __half*)d->y1.data;
    a.y2 = (const __half*)d->y2.data;
    a.dy1 = (const __half*)d->dy1.data;
    a.dy2 = (const __half*)d->dy2.data;
    a.dy3 = (const __half*)d->dy3.data;
    a.dx = (__half*)d->dx.data;
    a.H = d->x.H;
    a.W = d->x.W;
    a.C = d->x.C;
    a.acc = d->dx_acc;
    // the gather form (round 6) where its six planes fit the LDS and the 16-bit pixel indices reach (Y6_SPPF_BWD=0: A/B)
    static const bool gather_form = !(getenv("Y6_SPPF_BWD") && atoi(getenv("Y6_SPPF_BWD")) == 0);
    const size_t lds2 = (size_t)a.H * a.W * 8 * (2 * sizeof(float) + 4 * sizeof(unsigned short));
    if (gather_form && lds2 <= 160 * 1024 - 1024 && a.H * a.W < 65536) {
        static bool big2 = false;
        if (lds2 > 64 * 1024 && !big2) {
            Y6_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_bwd2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            big2 = true;
        }
        hipLaunchKernelGGL(sppf_bwd2_kernel, dim3((unsigned)(d->x.B * (a.C / 8))), dim3(256), lds2, s, a);
        Y6_LAUNCH_CHECK();
        return Y6_OK;
    }
    const size_t lds = (size_t)3 * a.H * a.W * 8 * sizeof(float);
    Y6_REQUIRE(lds <= 160 * 1024 - 1024, "sppf_pool_backward: %dx%d plane does not fit the LDS", a.H, a.W);
    if (lds > 64 * 1024)
        Y6_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(sppf_bwd_kernel, dim3((unsigned)(d->x.B * (a.C / 8))), dim3(256), lds, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ------------------------------------------------------------------ Detect training branch: pack / unpack
struct HeadArgs {
    int n_levels, nc, nreg, B, A;
    const __half* cls[4];
    const __half* reg[4];
    __half* dcls[4];
    __half* dreg[4];
    int ccs[4], cco[4], rcs[4], rco[4], hw[4], a0[5];
    float* scores;
    float* distri;
    const float* dscores;
    const float* ddistri;
};

__global__ __launch_bounds__(256) void head_pack_kernel(const HeadArgs a) {
    const int per = a.nc + a.nreg;
    const size_t total = (size_t)a.B * a.A * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % per);
        const size_t ba = i / per;
        const int an = (int)(ba % a.A), b = (int)(ba / a.A);
        int l = 0;
        while (l + 1 < a.n_levels && an >= a.a0[l + 1]) ++l;
        const size_t pix = (size_t)b * a.hw[l] + (an - a.a0[l]);
        if (c < a.nc) {
            const float z = __half2float(a.cls[l][pix * a.ccs[l] + a.cco[l] + c]);
            a.scores[ba * a.nc + c] = 1.f / (1.f + expf(-z));
        } else {
            const int r = c - a.nc;
            a.distri[ba * a.nreg + r] = __half2float(a.reg[l][pix * a.rcs[l] + a.rco[l] + r]);
        }
    }
}

__global__ __launch_bounds__(256) void head_unpack_kernel(const HeadArgs a) {
    const int per = a.nc + a.nreg;
    const size_t total = (size_t)a.B * a.A * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % per);
        const size_t ba = i / per;
        const int an = (int)(ba % a.A), b = (int)(ba / a.A);
        int l = 0;
        while (l + 1 < a.n_levels && an >= a.a0[l + 1]) ++l;
        const size_t pix = (size_t)b * a.hw[l] + (an - a.a0[l]);
        if (c < a.nc) {
            const float p = a.scores[ba * a.nc + c];
            a.dcls[l][pix * a.ccs[l] + a.cco[l] + c] = __float2half(a.dscores[ba * a.nc + c] * p * (1.f - p));
        } else {
            const int r = c - a.nc;
            a.dreg[l][pix * a.rcs[l] + a.rco[l] + r] = __float2half(a.ddistri[ba * a.nreg + r]);
        }
    }
}

// (round 6, last session) four channels per thread - an 8-byte fp16 access and a 16-byte fp32 access, one index decode per four
// elements (the per-element kernels ran at 1.1 / 2.7 TB/s: 64-bit divisions and 2-byte accesses per element); the same arithmetic
// per element, so the same bits.  Taken when every channel count / pitch / offset is a multiple of four.
__global__ __launch_bounds__(256) void head_pack4_kernel(const HeadArgs a) {
    const unsigned gc = (unsigned)a.nc >> 2, per = gc + ((unsigned)a.nreg >> 2);
    const unsigned total = (unsigned)a.B * (unsigned)a.A * per;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned ba = i / per, g = i - ba * per;
        const unsigned b = ba / (unsigned)a.A, an = ba - b * (unsigned)a.A;
        int l = 0;
        while (l + 1 < a.n_levels && (int)an >= a.a0[l + 1]) ++l;
        const size_t pix = (size_t)b * a.hw[l] + (an - a.a0[l]);
        if (g < gc) {
            float z[4];
            load4(a.cls[l] + pix * a.ccs[l] + a.cco[l] + 4 * g, z);
            float4 o;
            o.x = 1.f / (1.f + expf(-z[0]));
            o.y = 1.f / (1.f + expf(-z[1]));
            o.z = 1.f / (1.f + expf(-z[2]));
            o.w = 1.f / (1.f + expf(-z[3]));
            *reinterpret_cast<float4*>(a.scores + (size_t)ba * a.nc + 4 * g) = o;
        } else {
            const unsigned r = 4 * (g - gc);
            float v[4];
            load4(a.reg[l] + pix * a.rcs[l] + a.rco[l] + r, v);
            *reinterpret_cast<float4*>(a.distri + (size_t)ba * a.nreg + r) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

__global__ __launch_bounds__(256) void head_unpack4_kernel(const HeadArgs a) {
    const unsigned gc = (unsigned)a.nc >> 2, per = gc + ((unsigned)a.nreg >> 2);
    const unsigned total = (unsigned)a.B * (unsigned)a.A * per;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned ba = i / per, g = i - ba * per;
        const unsigned b = ba / (unsigned)a.A, an = ba - b * (unsigned)a.A;
        int l = 0;
        while (l + 1 < a.n_levels && (int)an >= a.a0[l + 1]) ++l;
        const size_t pix = (size_t)b * a.hw[l] + (an - a.a0[l]);
        float o[4];
        if (g < gc) {
            const float4 p = *reinterpret_cast<const float4*>(a.scores + (size_t)ba * a.nc + 4 * g);
            const float4 d = *reinterpret_cast<const float4*>(a.dscores + (size_t)ba * a.nc + 4 * g);
            o[0] = d.x * p.x * (1.f - p.x);
            o[1] = d.y * p.y * (1.f - p.y);
            o[2] = d.z * p.z * (1.f - p.z);
            o[3] = d.w * p.w * (1.f - p.w);
            store4(a.dcls[l] + pix * a.ccs[l] + a.cco[l] + 4 * g, o);
        } else {
            const unsigned r = 4 * (g - gc);
            const float4 d = *reinterpret_cast<const float4*>(a.ddistri + (size_t)ba * a.nreg + r);
            o[0] = d.x, o[1] = d.y, o[2] = d.z, o[3] = d.w;
            store4(a.dreg[l] + pix * a.rcs[l] + a.rco[l] + r, o);
        }
    }
}

// every access of the four-channel kernels is aligned
bool head_vec4_ok(const HeadArgs& a, bool backward) {
    static const bool off = getenv("Y6_HEAD_VEC4") != nullptr && atoi(getenv("Y6_HEAD_VEC4")) == 0;   // A/B switch
    if (off || a.nc % 4 || a.nreg % 4 || a.nreg == 0) return false;
    if ((double)a.B * a.A * ((a.nc + a.nreg) / 4) >= 4.0e9) return false;
    auto al = [](const void* p, size_t n) { return (((uintptr_t)p) & (n - 1)) == 0; };
    for (int l = 0; l < a.n_levels; ++l) {
        if (a.rcs[l] % 4 || a.rco[l] % 4 || !al(a.reg[l], 8)) return false;
        if (a.nc && (a.ccs[l] % 4 || a.cco[l] % 4 || !al(a.cls[l], 8))) return false;
    }
    if (a.nc && (!al(a.scores, 16) || (backward && !al(a.dscores, 16)))) return false;
    return backward ? al(a.ddistri, 16) : al(a.distri, 16);
}

int fill_head_args(const y6_head_pack_desc* d, HeadArgs* a, bool backward) {
    // nc == 0: a regression-only pack (the plain-distance output of the distillation head, effidehead_distill_ns.py:95-101)
    Y6_REQUIRE(d && d->n_levels >= 1 && d->n_levels <= 4 && (d->scores || d->nc == 0), "head_pack: bad descriptor");
    Y6_REQUIRE(backward ? ((d->dscores || d->nc == 0) && d->ddistri) : (d->distri != nullptr), "head_pack: null buffer");
    memset(a, 0, sizeof(*a));
    a->n_levels = d->n_levels;
    a->nc = d->nc;
    a->nreg = d->nreg;
    int A = 0;
    for (int l = 0; l < d->n_levels; ++l) {
        const y6_tensor &c = d->cls[l], &r = d->reg[l];
        Y6_REQUIRE(r.data && r.C == d->nreg && (d->nc == 0 || (c.data && c.C == d->nc)), "head_pack: level %d channels", l);
        Y6_REQUIRE(r.B == d->reg[0].B && (d->nc == 0 || (c.B == r.B && c.H == r.H && c.W == r.W)), "head_pack: level %d shape mismatch", l);
        a->cls[l] = (const __half*)c.data;
        a->reg[l] = (const __half*)r.data;
        a->dcls[l] = (__half*)c.data;
        a->dreg[l] = (__half*)r.data;
        a->ccs[l] = c.cstride;
        a->cco[l] = c.coff;
        a->rcs[l] = r.cstride;
        a->rco[l] = r.coff;
        a->hw[l] = r.H * r.W;
        a->a0[l] = A;
        A += r.H * r.W;
    }
    a->a0[d->n_levels] = A;
    a->B = d->reg[0].B;
    a->A = A;
    a->scores = d->scores;
    a->distri = d->distri;
    a->dscores = d->dscores;
    a->ddistri = d->ddistri;
    return Y6_OK;
}

int head_pack_launch(const y6_head_pack_desc* d, hipStream_t s) {
    HeadArgs a;
    int rc = fill_head_args(d, &a, false);
    if (rc) return rc;
    if (head_vec4_ok(a, false))
        hipLaunchKernelGGL(head_pack4_kernel, dim3(grid_for((size_t)a.B * a.A * ((a.nc + a.nreg) / 4), 256, 256 * 32)), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(head_pack_kernel, dim3(grid_for((size_t)a.B * a.A * (a.nc + a.nreg), 256, 256 * 32)), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
int head_unpack_launch(const y6_head_pack_desc* d, hipStream_t s) {
    HeadArgs a;
    int rc = fill_head_args(d, &a, true);
    if (rc) return rc;
    if (head_vec4_ok(a, true))
        hipLaunchKernelGGL(head_unpack4_kernel, dim3(grid_for((size_t)a.B * a.A * ((a.nc + a.nreg) / 4), 256, 256 * 32)), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(head_unpack_kernel, dim3(grid_for((size_t)a.B * a.A * (a.nc + a.nreg), 256, 256 * 32)), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ------------------------------------------------------------------ fuse_ab head: anchor-based auxiliary branch
// Detect (effidehead_fuseab.py:110-124): per level cls_ab [B,H,W,na*nc] -> sigmoid -> scores [B, na*HW (anchor-major), nc];
// reg_ab [B,H,W,na*4] -> (dx, dy, (2 sigmoid(w))^2 * aw, (2 sigmoid(h))^2 * ah) -> distri [B, na*HW, 4]
struct HeadAbArgs {
    int n_levels, nc, na, B, A;          // A = sum_l na * H_l * W_l
    __half* cls[4];
    __half* reg[4];
    int ccs[4], cco[4], rcs[4], rco[4], hw[4], a0[5];
    float anchors[4][3][2];
    float* scores;
    float* distri;
    const float* dscores;
    const float* ddistri;
};

template <bool BWD>
__global__ __launch_bounds__(256) void head_ab_kernel(const HeadAbArgs a) {
    const int per = a.nc + 4;
    const size_t total = (size_t)a.B * a.A * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % per);
        const size_t ba = i / per;
        const int an = (int)(ba % a.A), b = (int)(ba / a.A);
        int l = 0;
        while (l + 1 < a.n_levels && an >= a.a0[l + 1]) ++l;
        const int local = an - a.a0[l];
        const int anc = local / a.hw[l], pixl = local - anc * a.hw[l];
        const size_t pix = (size_t)b * a.hw[l] + pixl;
        if (c < a.nc) {
            __half* q = a.cls[l] + pix * a.ccs[l] + a.cco[l] + anc * a.nc + c;
            if (!BWD) {
                a.scores[ba * a.nc + c] = 1.f / (1.f + expf(-__half2float(*q)));
            } else {
                const float p = a.scores[ba * a.nc + c];
                *q = __float2half(a.dscores[ba * a.nc + c] * p * (1.f - p));
            }
        } else {
            const int j = c - a.nc;
            __half* q = a.reg[l] + pix * a.rcs[l] + a.rco[l] + anc * 4 + j;
            if (!BWD) {
                const float v = __half2float(*q);
                float o = v;
                if (j >= 2) {
                    const float sg = 1.f / (1.f + expf(-v));
                    o = (2.f * sg) * (2.f * sg) * a.anchors[l][anc][j - 2];
                }
                a.distri[ba * 4 + j] = o;
            }
        }
    }
}

// backward of the box transform reads the forward reg maps (the gradient maps are separate buffers)
struct HeadAbBwdArgs {
    HeadAbArgs f;
    const __half* reg_fwd[4];
    int fcs[4], fco[4];
};
__global__ __launch_bounds__(256) void head_ab_regbwd_kernel(const HeadAbBwdArgs g) {
    const HeadAbArgs& a = g.f;
    const size_t total = (size_t)a.B * a.A * 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3);
        const size_t ba = i >> 2;
        const int an = (int)(ba % a.A), b = (int)(ba / a.A);
        int l = 0;
        while (l + 1 < a.n_levels && an >= a.a0[l + 1]) ++l;
        const int local = an - a.a0[l];
        const int anc = local / a.hw[l], pixl = local - anc * a.hw[l];
        const size_t pix = (size_t)b * a.hw[l] + pixl;
        float d = a.ddistri[ba * 4 + j];
        if (j >= 2) {
            const float v = __half2float(g.reg_fwd[l][pix * g.fcs[l] + g.fco[l] + anc * 4 + j]);
            const float sg = 1.f / (1.f + expf(-v));
            d *= 8.f * sg * sg * (1.f - sg) * a.anchors[l][anc][j - 2];
        }
        a.reg[l][pix * a.rcs[l] + a.rco[l] + anc * 4 + j] = __float2half(d);
    }
}

int fill_head_ab(const y6_head_ab_desc* d, HeadAbArgs* a, bool bwd) {
    Y6_REQUIRE(d && d->n_levels >= 1 && d->n_levels <= 4 && d->na >= 1 && d->na <= 3 && d->scores, "head_ab: bad descriptor");
    Y6_REQUIRE(bwd ? (d->dscores && d->ddistri) : (d->distri != nullptr), "head_ab: null buffer");
    memset(a, 0, sizeof(*a));
    a->n_levels = d->n_levels;
    a->nc = d->nc;
    a->na = d->na;
    int A = 0;
    for (int l = 0; l < d->n_levels; ++l) {
        const y6_tensor &c = d->cls[l], &r = d->reg[l];
        Y6_REQUIRE(c.data && r.data && c.C == d->na * d->nc && r.C == d->na * 4, "head_ab: level %d channels", l);
        Y6_REQUIRE(c.B == r.B && c.H == r.H && c.W == r.W && c.B == d->cls[0].B, "head_ab: level %d shape mismatch", l);
        a->cls[l] = (__half*)c.data;
        a->reg[l] = (__half*)r.data;
        a->ccs[l] = c.cstride;
        a->cco[l] = c.coff;
        a->rcs[l] = r.cstride;
        a->rco[l] = r.coff;
        a->hw[l] = c.H * c.W;
        a->a0[l] = A;
        A += d->na * c.H * c.W;
        for (int k = 0; k < d->na; ++k) {
            a->anchors[l][k][0] = d->anchors[(l * 3 + k) * 2];
            a->anchors[l][k][1] = d->anchors[(l * 3 + k) * 2 + 1];
        }
    }
    a->a0[d->n_levels] = A;
    a->B = d->cls[0].B;
    a->A = A;
    a->scores = d->scores;
    a->distri = d->distri;
    a->dscores = d->dscores;
    a->ddistri = d->ddistri;
    return Y6_OK;
}
int head_ab_pack_launch(const y6_head_ab_desc* d, hipStream_t s) {
    HeadAbArgs a;
    int rc = fill_head_ab(d, &a, false);
    if (rc) return rc;
    hipLaunchKernelGGL(head_ab_kernel<false>, dim3(grid_for((size_t)a.B * a.A * (a.nc + 4), 256, 256 * 32)), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
int head_ab_unpack_launch(const y6_head_ab_desc* d, hipStream_t s) {
    HeadAbBwdArgs g;
    memset(&g, 0, sizeof(g));
    int rc = fill_head_ab(d, &g.f, true);
    if (rc) return rc;
    for (int l = 0; l < d->n_levels; ++l) {
        Y6_REQUIRE(d->reg_fwd[l].data, "head_ab: backward needs the forward reg maps");
        g.reg_fwd[l] = (const __half*)d->reg_fwd[l].data;
        g.fcs[l] = d->reg_fwd[l].cstride;
        g.fco[l] = d->reg_fwd[l].coff;
    }
    hipLaunchKernelGGL(head_ab_kernel<true>, dim3(grid_for((size_t)g.f.B * g.f.A * (g.f.nc + 4), 256, 256 * 32)), dim3(256), 0, s, g.f);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_ab_regbwd_kernel, dim3(grid_for((size_t)g.f.B * g.f.A * 4, 256, 256 * 8)), dim3(256), 0, s, g);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ------------------------------------------------------------------ small data movers
struct TwoT {
    y6_tensor a, b;
    int flag;
};

__global__ __launch_bounds__(256) void s2d_kernel(const __half* __restrict__ src, int scs, int sco, int H, int W, int C,
                                                  __half* __restrict__ dst, int dcs, int dco, int B) {
    const int G = C >> 3, Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * 4 * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        size_t t = i / G;
        const int sub = (int)(t & 3);
        t >>= 2;
        const int x = (int)(t % Wo);
        t /= Wo;
        const int y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const uint4 v = *reinterpret_cast<const uint4*>(src + ((size_t)(b * H + 2 * y + (sub >> 1)) * W + 2 * x + (sub & 1)) * scs + sco + g * 8);
        *reinterpret_cast<uint4*>(dst + ((size_t)(b * Ho + y) * Wo + x) * dcs + dco + sub * C + g * 8) = v;
    }
}
int s2d_launch(const TwoT* d, hipStream_t s) {
    const y6_tensor &a = d->a, &o = d->b;
    Y6_REQUIRE(view_ok(a) && view_ok(o), "space_to_depth2: bad views");
    Y6_REQUIRE(a.H % 2 == 0 && a.W % 2 == 0 && o.H == a.H / 2 && o.W == a.W / 2 && o.C == 4 * a.C && o.B == a.B, "space_to_depth2: shape mismatch");
    hipLaunchKernelGGL(s2d_kernel, dim3(grid_for((size_t)a.B * a.H * a.W * (a.C / 8), 256)), dim3(256), 0, s, (const __half*)a.data,
                       a.cstride, a.coff, a.H, a.W, a.C, (__half*)o.data, o.cstride, o.coff, a.B);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// dst[b,y,x,:] = src[b,2y,2x,:]  (the input sampling of a 1x1 stride-2 conv: the MFMA 1x1 kernels are stride-1 GEMMs)
__global__ __launch_bounds__(256) void subsample2_kernel(const __half* __restrict__ src, int scs, int sco, int H, int W, int C,
                                                         __half* __restrict__ dst, int dcs, int dco, int B, int Ho, int Wo) {
    const int G = C >> 3;
    const size_t total = (size_t)B * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        size_t t = i / G;
        const int x = (int)(t % Wo);
        t /= Wo;
        const int y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        *reinterpret_cast<uint4*>(dst + ((size_t)(b * Ho + y) * Wo + x) * dcs + dco + g * 8) =
            *reinterpret_cast<const uint4*>(src + ((size_t)(b * H + 2 * y) * W + 2 * x) * scs + sco + g * 8);
    }
}
int subsample2_launch(const TwoT* d, hipStream_t s) {
    const y6_tensor &a = d->a, &o = d->b;
    Y6_REQUIRE(view_ok(a) && view_ok(o), "subsample2: bad views");
    Y6_REQUIRE(o.H == (a.H + 1) / 2 && o.W == (a.W + 1) / 2 && o.C == a.C && o.B == a.B, "subsample2: shape mismatch");
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for((size_t)o.B * o.H * o.W * (o.C / 8), 256)), dim3(256), 0, s, (const __half*)a.data,
                       a.cstride, a.coff, a.H, a.W, a.C, (__half*)o.data, o.cstride, o.coff, a.B, o.H, o.W);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

__global__ __launch_bounds__(256) void tensor_add_kernel(const __half* __restrict__ a, int acs, int aco, __half* __restrict__ dst, int dcs,
                                                         int dco, long npix, int C, int acc) {
    const int G = C >> 3;
    const long total = npix * G;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / G;
        const int g = (int)(i - p * G);
        float v[8];
        load8(a + p * acs + aco + g * 8, v);
        __half* q = dst + p * dcs + dco + g * 8;
        if (acc) {
            float o[8];
            load8(q, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += o[j];
        }
        store8(q, v);
    }
}
int tensor_add_launch(const TwoT* d, hipStream_t s) {
    Y6_REQUIRE(view_ok(d->a) && view_ok(d->b) && same_shape(d->a, d->b), "tensor_add: bad views");
    const long npix = (long)d->a.B * d->a.H * d->a.W;
    hipLaunchKernelGGL(tensor_add_kernel, dim3(grid_for((size_t)npix * (d->a.C / 8), 256)), dim3(256), 0, s, (const __half*)d->a.data,
                       d->a.cstride, d->a.coff, (__half*)d->b.data, d->b.cstride, d->b.coff, npix, d->a.C, d->flag);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// dst (=|+=) [src +] AvgPool2d(3, stride 1, pad 1, count_include_pad)(src): the `rbr_avg` branch of QARepVGGBlockV2 (reference
// yolov6/layers/common.py:404, :416-419) merged with the block's raw identity branch (flag bit 1).  The pooling is symmetric,
// so its backward is the same op on the gradient (flag bit 0: accumulate into dst).  fp32 sum of the nine fp16 taps / 9.
__global__ __launch_bounds__(256) void avgpool3_kernel(const __half* __restrict__ src, int scs, int sco, __half* __restrict__ dst, int dcs,
                                                       int dco, int B, int H, int W, int C, int acc, int with_id) {
    const int G = C >> 3;
    const size_t total = (size_t)B * H * W * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        size_t t = i / G;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        float sum[8], ctr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] = ctr[j] = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
                float v[8];
                load8(src + ((size_t)(b * H + yy) * W + xx) * scs + sco + g * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) sum[j] += v[j];
                if (dy == 0 && dx == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ctr[j] = v[j];
                }
            }
        __half* q = dst + ((size_t)(b * H + y) * W + x) * dcs + dco + g * 8;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = sum[j] / 9.f + (with_id ? ctr[j] : 0.f);
        if (acc) {
            float old[8];
            load8(q, old);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += old[j];
        }
        store8(q, o);
    }
}
int avgpool3_launch(const TwoT* d, hipStream_t s) {
    const y6_tensor &a = d->a, &o = d->b;
    Y6_REQUIRE(view_ok(a) && view_ok(o) && same_shape(a, o), "avgpool3: bad views");
    Y6_REQUIRE(a.data != o.data, "avgpool3: in-place pooling is not defined");
    hipLaunchKernelGGL(avgpool3_kernel, dim3(grid_for((size_t)a.B * a.H * a.W * (a.C / 8), 256, 256 * 16)), dim3(256), 0, s, (const __half*)a.data,
                       a.cstride, a.coff, (__half*)o.data, o.cstride, o.coff, a.B, a.H, a.W, a.C, d->flag & 1, (d->flag >> 1) & 1);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

struct ChanSum {
    y6_tensor x;
    float* out;
    void* ws;
    size_t ws_bytes;
};
__global__ void chan_sum_finalize_kernel(const double* __restrict__ ws, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) out[c] += (float)ws[c];
}
// generic (any C): one thread per (pixel-row-slice, channel); used for the prediction convs (C = 80, 68, 4)
__global__ __launch_bounds__(256) void chan_sum_kernel(const __half* __restrict__ x, int cs, int co, long npix, int C, double* __restrict__ ws) {
    const int rows = C <= 256 ? 256 / C : 1;          // rows of C threads; wider tensors: one row, channels strided by 256
    const int row = threadIdx.x / (C <= 256 ? C : 256);
    if (row >= rows) return;
    for (int c = threadIdx.x - row * (C <= 256 ? C : 256); c < C; c += 256) {
        double s = 0.0;
        for (long p = (long)blockIdx.x * rows + row; p < npix; p += (long)gridDim.x * rows) s += (double)__half2float(x[p * cs + co + c]);
        atomicAdd(&ws[c], s);
    }
}
int chan_sum_launch(const ChanSum* d, hipStream_t s) {
    const y6_tensor& t = d->x;
    Y6_REQUIRE(t.data && d->out && d->ws && t.C >= 1 && t.C <= 4096, "channel_sum: bad arguments (1..4096 channels)");
    Y6_REQUIRE(d->ws_bytes >= (size_t)t.C * sizeof(double), "channel_sum: workspace too small");
    const long npix = (long)t.B * t.H * t.W;
    // 16-byte path: the view padded up to a multiple of 8 channels lies inside the buffer (the prediction-conv gradients are
    // stored 8-padded with zero pad channels) and is aligned -> the BatchNorm statistics kernel, sums only
    const int C8 = (t.C + 7) / 8 * 8;
    if (t.coff % 8 == 0 && t.cstride % 8 == 0 && t.coff + C8 <= t.cstride && (((uintptr_t)t.data) & 15) == 0 && C8 <= 2048 &&
        d->ws_bytes >= (size_t)2 * C8 * sizeof(double)) {
        double* ws = (double*)d->ws;
        Y6_HIP(hipMemsetAsync(ws, 0, (size_t)2 * C8 * sizeof(double), s));
        const int G = C8 / 8, R = 256 / G;
        long ppb = (long)R * 64;
        long blocks = (npix + ppb - 1) / ppb;
        if (blocks > 2048) {
            blocks = 2048;
            ppb = (npix + blocks - 1) / blocks;
        }
        hipLaunchKernelGGL(bn_sum_kernel, dim3((unsigned)blocks), dim3(256), (size_t)2 * C8 * sizeof(double), s, (const __half*)t.data,
                           t.cstride, t.coff, npix, G, ppb, ws, C8);
        Y6_LAUNCH_CHECK();
        hipLaunchKernelGGL(chan_sum_finalize_kernel, dim3((unsigned)((t.C + 255) / 256)), dim3(256), 0, s, (const double*)ws, t.C, d->out);
        Y6_LAUNCH_CHECK();
        return Y6_OK;
    }
    Y6_HIP(hipMemsetAsync(d->ws, 0, (size_t)t.C * sizeof(double), s));
    long g = (npix + 63) / 64;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(chan_sum_kernel, dim3((unsigned)g), dim3(256), 0, s, (const __half*)t.data, t.cstride, t.coff, npix, t.C, (double*)d->ws);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(chan_sum_finalize_kernel, dim3((unsigned)((t.C + 255) / 256)), dim3(256), 0, s, (const double*)d->ws, t.C, d->out);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

struct FillZero {
    void* p;
    size_t n;
};
int fill_zero_launch(const FillZero* d, hipStream_t s) {
    Y6_HIP(hipMemsetAsync(d->p, 0, d->n, s));
    return Y6_OK;
}

// ------------------------------------------------------------------ optimizer
__global__ __launch_bounds__(256) void finite_check_kernel(const float* __restrict__ g, size_t n, int32_t* __restrict__ flag) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        bad = bad || !(fabsf(v) <= 3.402823466e38f);    // inf or nan
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, size_t n, float lr,
                                                  float mom, float wd, int nesterov, int first, const float* __restrict__ scale,
                                                  const int32_t* __restrict__ found_inf) {
    if (found_inf && *found_inf) return;
    const float inv = scale ? 1.f / *scale : 1.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float w = p[i];
        float d = g[i] * inv;
        if (wd != 0.f) d += wd * w;
        float b = d;
        if (mom != 0.f) {
            b = first ? d : mom * m[i] + d;
            m[i] = b;
            if (nesterov) b = d + mom * b;
        }
        p[i] = w - lr * b;
    }
}

struct SgdGroups {
    float lr[4];
    float wd[4];
};
// per-element parameter group (0: BatchNorm weights, 1: conv / convT weights, 2: biases, >= 3: not optimised), the three
// groups of yolov6/solver/build.py:12-29; grad_mul folds the DDP average (1 / world size) into the same pass
__global__ __launch_bounds__(256) void sgd_grouped_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          const uint8_t* __restrict__ group, size_t n, const SgdGroups gr, float mom, int nesterov,
                                                          int first, float grad_mul, const float* __restrict__ scale,
                                                          const int32_t* __restrict__ found_inf) {
    if (found_inf && *found_inf) return;
    const float inv = (scale ? 1.f / *scale : 1.f) * grad_mul;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = group[i];
        if (k > 2) continue;
        const float w = p[i];
        float d = g[i] * inv;
        if (gr.wd[k] != 0.f) d += gr.wd[k] * w;
        float b = d;
        if (mom != 0.f) {
            b = first ? d : mom * m[i] + d;
            m[i] = b;
            if (nesterov) b = d + mom * b;
        }
        p[i] = w - gr.lr[k] * b;
    }
}

__global__ void scaler_update_kernel(float* scale, int32_t* found_inf, int32_t* tracker, float growth, float backoff, int interval) {
    if (*found_inf) {
        *scale *= backoff;
        *tracker = 0;
    } else if (++(*tracker) >= interval) {
        *scale *= growth;
        *tracker = 0;
    }
    *found_inf = 0;
}

}  // namespace

// ====================================================================== C ABI
// [2C] totals (the atomic form, Y6_BN_ATOMICS=1) followed by [kBnPartBlocks][2C] block partials
extern "C" size_t y6_bn_stats_workspace_bytes(int C) { return (size_t)2 * C * sizeof(double) * (size_t)(1 + kBnPartBlocks); }
// ... sized for the tensor: the launcher uses min(kBnPartBlocks, ceil(npix / (16 pixel rows of 256 / (C / 8) pixels))) partial blocks.
// (The worst case above is 8.4 MB per BatchNorm at C = 512 - GBs over the hundreds of BatchNorms of an M / L training graph.)
static long bn_part_blocks(int C, long npix, int rows_per_thread) {
    const int G = C / 8 > 0 ? C / 8 : 1;
    const long ppb = (long)(256 / G > 0 ? 256 / G : 1) * rows_per_thread;
    const long blocks = (npix + ppb - 1) / ppb;
    return blocks > kBnPartBlocks ? kBnPartBlocks : (blocks < 1 ? 1 : blocks);
}
extern "C" size_t y6_bn_stats_workspace_bytes_for(int C, long npix) {
    return (size_t)2 * C * sizeof(double) * (size_t)(1 + bn_part_blocks(C, npix, 16));
}

// [4C + 1] totals followed by [kBnPartBlocks][4C + 1] block partials
extern "C" size_t y6_bnact_bwd_workspace_bytes(int C) { return ((size_t)4 * C + 1) * sizeof(double) * (size_t)(1 + kBnPartBlocks); }
extern "C" size_t y6_bnact_bwd_workspace_bytes_for(int C, long npix) {
    return ((size_t)4 * C + 1) * sizeof(double) * (size_t)(1 + bn_part_blocks(C, npix, 8));
}

extern "C" int y6_bn_train_stats(const y6_bn_train_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return bn_train_stats_launch(d, (hipStream_t)stream);
}
extern "C" int y6_bn_train_stats_multi(const y6_bn_train_multi_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return bn_train_stats_multi_launch(d, (hipStream_t)stream);
}
extern "C" int y6_bnact_forward(const y6_bnact_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return bnact_forward_launch(d, (hipStream_t)stream);
}
extern "C" int y6_bnact_backward(const y6_bnact_bwd_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return bnact_backward_launch(d, (hipStream_t)stream);
}
extern "C" int y6_wgrad_transpose(const y6_wgrad_t_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return wgrad_transpose_launch(d, (hipStream_t)stream);
}
extern "C" size_t y6_pack_job_elems(int kind, int Cout, int Cin, int K) {
    if (kind == 4) return (size_t)Cout * Cin * 9;
    if (kind == 2 && (Cout % 32) != 0) return (size_t)4 * (y6_cdiv(y6_cdiv(Cout, 32), 4) * 4) * y6_cdiv(Cin, 32) * 1024;
    int O = Cout, I = Cin, nt = K * K;
    if (kind == 1) { O = Cin; I = Cout; }
    if (kind == 2) { O = 4 * Cout; I = Cin; nt = 1; }
    if (kind == 3) { O = Cin; I = 4 * Cout; nt = 1; }
    const size_t cfr_pad = (size_t)y6_cdiv(y6_cdiv(O, 32), 4) * 4;
    return cfr_pad * y6_cdiv(I, 32) * nt * 1024;
}
extern "C" int y6_pack_weights_batched(const y6_pack_batch_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return pack_batch_launch(d, (hipStream_t)stream);
}
extern "C" int y6_sppf_pool_backward(const y6_sppf_bwd_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return sppf_backward_launch(d, (hipStream_t)stream);
}
extern "C" int y6_head_pack(const y6_head_pack_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return head_pack_launch(d, (hipStream_t)stream);
}
extern "C" int y6_head_unpack_backward(const y6_head_pack_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return head_unpack_launch(d, (hipStream_t)stream);
}
extern "C" int y6_head_ab_pack(const y6_head_ab_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return head_ab_pack_launch(d, (hipStream_t)stream);
}
extern "C" int y6_head_ab_unpack_backward(const y6_head_ab_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return head_ab_unpack_launch(d, (hipStream_t)stream);
}
extern "C" int y6_plan_add_head_ab_pack(y6_plan* p, const y6_head_ab_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    return y6_plan_push(p, head_ab_pack_launch, d, Y6_TOP_HEAD_PACK, 0.0, 0.0);
}
extern "C" int y6_plan_add_head_ab_unpack_backward(y6_plan* p, const y6_head_ab_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    return y6_plan_push(p, head_ab_unpack_launch, d, Y6_TOP_HEAD_UNPACK, 0.0, 0.0);
}
extern "C" int y6_space_to_depth2(const y6_tensor* src, const y6_tensor* dst, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(src && dst, "space_to_depth2: null argument");
    TwoT t{*src, *dst, 0};
    return s2d_launch(&t, (hipStream_t)stream);
}
extern "C" int y6_subsample2(const y6_tensor* src, const y6_tensor* dst, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(src && dst, "subsample2: null argument");
    TwoT t{*src, *dst, 0};
    return subsample2_launch(&t, (hipStream_t)stream);
}
extern "C" int y6_channel_sum(const y6_tensor* x, float* out_accum, void* workspace, size_t workspace_bytes, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(x, "channel_sum: null argument");
    ChanSum c{*x, out_accum, workspace, workspace_bytes};
    return chan_sum_launch(&c, (hipStream_t)stream);
}
extern "C" int y6_tensor_add(const y6_tensor* a, const y6_tensor* dst, int accumulate, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(a && dst, "tensor_add: null argument");
    TwoT t{*a, *dst, accumulate};
    return tensor_add_launch(&t, (hipStream_t)stream);
}
extern "C" int y6_grad_finite_check(const float* grad, size_t n, int32_t* found_inf, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(grad && found_inf, "grad_finite_check: null argument");
    if (n == 0) return Y6_OK;
    hipLaunchKernelGGL(finite_check_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, grad, n, found_inf);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
extern "C" int y6_sgd_step(float* param, const float* grad, float* momentum_buf, size_t n, float lr, float momentum, float weight_decay,
                           int nesterov, int first_step, const float* scale, const int32_t* found_inf, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(param && grad && (momentum == 0.f || momentum_buf), "sgd_step: null argument");
    if (n == 0) return Y6_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, n, lr, momentum,
                       weight_decay, nesterov, first_step, scale, found_inf);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
extern "C" int y6_sgd_step_grouped(float* param, const float* grad, float* momentum_buf, const uint8_t* group, size_t n, const float* lr3,
                                   const float* wd3, float momentum, int nesterov, int first_step, float grad_mul, const float* scale,
                                   const int32_t* found_inf, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(param && grad && group && lr3 && wd3 && (momentum == 0.f || momentum_buf), "sgd_step_grouped: null argument");
    if (n == 0) return Y6_OK;
    SgdGroups gr;
    for (int i = 0; i < 3; ++i) {
        gr.lr[i] = lr3[i];
        gr.wd[i] = wd3[i];
    }
    gr.lr[3] = gr.wd[3] = 0.f;
    hipLaunchKernelGGL(sgd_grouped_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, group, n,
                       gr, momentum, nesterov, first_step, grad_mul, scale, found_inf);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
extern "C" int y6_scaler_update(float* scale, int32_t* found_inf, int32_t* growth_tracker, float growth, float backoff, int interval, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(scale && found_inf && growth_tracker, "scaler_update: null argument");
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scale, found_inf, growth_tracker, growth, backoff, interval);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ---------------------------------------------------------------------- plan builders
static double nhwc_bytes(const y6_tensor& t) { return 2.0 * t.B * t.H * t.W * t.C; }

extern "C" int y6_plan_add_bn_train_stats(y6_plan* p, const y6_bn_train_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    return y6_plan_push(p, bn_train_stats_launch, d, Y6_TOP_BN_STATS, 0.0, nhwc_bytes(d->x));
}
extern "C" int y6_plan_add_bn_train_stats_multi(y6_plan* p, const y6_bn_train_multi_desc* d) {
    Y6_REQUIRE(p && d && d->n >= 1 && d->n <= 3, "plan_add: null argument");
    double bytes = 0.0;
    for (int t = 0; t < d->n; ++t) bytes += nhwc_bytes(d->d[t].x);
    return y6_plan_push(p, bn_train_stats_multi_launch, d, Y6_TOP_BN_STATS, 0.0, bytes);
}
extern "C" int y6_plan_add_bnact_forward(y6_plan* p, const y6_bnact_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    return y6_plan_push(p, bnact_forward_launch, d, Y6_TOP_BNACT_FWD, 0.0, nhwc_bytes(d->x[0]) * (d->n + 1 + (d->res.data ? 1 : 0)));
}
extern "C" int y6_plan_add_bnact_backward(y6_plan* p, const y6_bnact_bwd_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    int nout = 0;
    for (int b = 0; b < d->fwd.n; ++b) nout += d->dx[b].data ? 1 : 0;
    return y6_plan_push(p, bnact_backward_launch, d, Y6_TOP_BNACT_BWD, 0.0, nhwc_bytes(d->dout) * (2.0 * (d->fwd.n + 1) + nout));
}
extern "C" int y6_plan_add_wgrad_transpose(y6_plan* p, const y6_wgrad_t_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    return y6_plan_push(p, wgrad_transpose_launch, d, Y6_TOP_WGRAD_T, 0.0, 2.0 * 2.0 * d->src.C * d->src.B * d->R * d->Q);
}
extern "C" int y6_plan_add_pack_batch(y6_plan* p, const y6_pack_batch_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    return y6_plan_push(p, pack_batch_launch, d, Y6_TOP_PACK, 0.0, 6.0 * (double)d->total);
}
extern "C" int y6_plan_add_sppf_backward(y6_plan* p, const y6_sppf_bwd_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    return y6_plan_push(p, sppf_backward_launch, d, Y6_TOP_POOL_BWD, 0.0, nhwc_bytes(d->x) * 7);
}
extern "C" int y6_plan_add_head_pack(y6_plan* p, const y6_head_pack_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    double n = 0;
    for (int l = 0; l < d->n_levels; ++l) n += (double)d->cls[l].B * d->cls[l].H * d->cls[l].W;
    return y6_plan_push(p, head_pack_launch, d, Y6_TOP_HEAD_PACK, 0.0, n * (d->nc + d->nreg) * 6.0);
}
extern "C" int y6_plan_add_head_unpack_backward(y6_plan* p, const y6_head_pack_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    double n = 0;
    for (int l = 0; l < d->n_levels; ++l) n += (double)d->cls[l].B * d->cls[l].H * d->cls[l].W;
    return y6_plan_push(p, head_unpack_launch, d, Y6_TOP_HEAD_UNPACK, 0.0, n * (d->nc + d->nreg) * 10.0);
}
extern "C" int y6_plan_add_space_to_depth2(y6_plan* p, const y6_tensor* src, const y6_tensor* dst) {
    Y6_REQUIRE(p && src && dst, "plan_add: null argument");
    TwoT t{*src, *dst, 0};
    return y6_plan_push(p, s2d_launch, &t, Y6_TOP_S2D, 0.0, 2.0 * nhwc_bytes(*src));
}
extern "C" int y6_plan_add_subsample2(y6_plan* p, const y6_tensor* src, const y6_tensor* dst) {
    Y6_REQUIRE(p && src && dst, "plan_add: null argument");
    TwoT t{*src, *dst, 0};
    return y6_plan_push(p, subsample2_launch, &t, Y6_TOP_S2D, 0.0, 2.0 * nhwc_bytes(*dst));
}
extern "C" int y6_plan_add_channel_sum(y6_plan* p, const y6_tensor* x, float* out_accum, void* workspace, size_t workspace_bytes) {
    Y6_REQUIRE(p && x, "plan_add: null argument");
    ChanSum c{*x, out_accum, workspace, workspace_bytes};
    return y6_plan_push(p, chan_sum_launch, &c, Y6_TOP_BIAS_GRAD, 0.0, nhwc_bytes(*x));
}
extern "C" int y6_plan_add_tensor_add(y6_plan* p, const y6_tensor* a, const y6_tensor* dst, int accumulate) {
    Y6_REQUIRE(p && a && dst, "plan_add: null argument");
    TwoT t{*a, *dst, accumulate};
    return y6_plan_push(p, tensor_add_launch, &t, Y6_TOP_ADD, 0.0, nhwc_bytes(*a) * (accumulate ? 3 : 2));
}
extern "C" int y6_avgpool3(const y6_tensor* src, const y6_tensor* dst, int with_identity, int accumulate, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(src && dst, "avgpool3: null argument");
    TwoT t{*src, *dst, (accumulate ? 1 : 0) | (with_identity ? 2 : 0)};
    return avgpool3_launch(&t, (hipStream_t)stream);
}
extern "C" int y6_plan_add_avgpool3(y6_plan* p, const y6_tensor* src, const y6_tensor* dst, int with_identity, int accumulate) {
    Y6_REQUIRE(p && src && dst, "plan_add_avgpool3: null argument");
    TwoT t{*src, *dst, (accumulate ? 1 : 0) | (with_identity ? 2 : 0)};
    return y6_plan_push(p, avgpool3_launch, &t, Y6_TOP_AVGPOOL3, 0.0, nhwc_bytes(*src) * (accumulate ? 3 : 2));
}
extern "C" int y6_plan_add_fill_zero(y6_plan* p, void* ptr, size_t bytes) {
    Y6_REQUIRE(p && ptr, "plan_add: null argument");
    FillZero f{ptr, bytes};
    return y6_plan_push(p, fill_zero_launch, &f, Y6_TOP_FILL, 0.0, (double)bytes);
}
