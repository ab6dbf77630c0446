// quant.hip — the host side of the int8 path (include/yolov6_hip.h: y6_conv_i8_desc): weight packing, calibration
// reduction, stand-alone activation quantiser, C ABI + plan ops.  The int8 conv kernels themselves live next to the
// fp16 ones in conv_mfma.hip (they share tile geometry and the epilogue).
#include "common.hpp"
#include "plan_internal.hpp"

namespace {

// dst[cfr][chunk][tap][ks][lane][j] = Wq[o = cfr*32 + (lane&31)][i = chunk*64 + ks*32 + (lane>>5)*16 + j][tap]
__global__ void pack_i8_kernel(const signed char* __restrict__ src, int Cout, int Cin, int K, int cfr_pad, int nchunk,
                               signed char* __restrict__ dst) {
    const int NT = K * K;
    const size_t total = (size_t)cfr_pad * nchunk * NT * 2048;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        const int lane = (int)((i >> 4) & 63);
        const int ks = (int)((i >> 10) & 1);
        size_t r = i >> 11;
        const int tap = (int)(r % NT);
        r /= NT;
        const int chunk = (int)(r % nchunk);
        const int cfr = (int)(r / nchunk);
        const int o = cfr * 32 + (lane & 31);
        const int c = chunk * 64 + ks * 32 + (lane >> 5) * 16 + j;
        signed char v = 0;
        if (o < Cout && c < Cin) v = src[((size_t)o * Cin + c) * NT + tap];
        dst[i] = v;
    }
}

// max |x| of an fp16 NHWC view; non-negative floats order like their bit patterns, so the block result is merged with
// an unsigned atomicMax
__global__ __launch_bounds__(256) void absmax_kernel(const __half* __restrict__ x, size_t npix, int C, int cs, int co,
                                                     unsigned* __restrict__ out) {
    const int pieces = C >> 3;   // 16-byte pieces per pixel
    const size_t total = npix * pieces;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / pieces;
        const int q = (int)(i - p * pieces);
        const uint4 v = *reinterpret_cast<const uint4*>(x + p * cs + co + q * 8);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = fabsf(__half2float(__ushort_as_half((unsigned short)(w[k] & 0xffffu))));
            const float b = fabsf(__half2float(__ushort_as_half((unsigned short)(w[k] >> 16))));
            m = fmaxf(m, fmaxf(a, b));   // fmaxf drops NaN operands
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        atomicMax(out, __float_as_uint(m));
    }
}

__device__ __forceinline__ unsigned q8_pair(unsigned x2, unsigned inv2, unsigned lo2, unsigned hi2) {
    unsigned r;
    asm("v_pk_max_f16 %0, %1, %2\n\tv_pk_min_f16 %0, %0, %3\n\tv_pk_fma_f16 %0, %0, %4, %5"
        : "=&v"(r)
        : "v"(x2), "v"(lo2), "v"(hi2), "v"(inv2), "v"(0x66006600u));
    return r;
}
__global__ __launch_bounds__(256) void quantize_kernel(const __half* __restrict__ x, size_t npix, int C, int cs, int co,
                                                       signed char* __restrict__ q, int qcs, int qco, unsigned inv2,
                                                       unsigned lo2, unsigned hi2) {
    const int pieces = C >> 3;
    const size_t total = npix * pieces;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / pieces;
        const int k = (int)(i - p * pieces);
        const uint4 v = *reinterpret_cast<const uint4*>(x + p * cs + co + k * 8);
        uint2 o;
        o.x = __builtin_amdgcn_perm(q8_pair(v.y, inv2, lo2, hi2), q8_pair(v.x, inv2, lo2, hi2), 0x06040200u);
        o.y = __builtin_amdgcn_perm(q8_pair(v.w, inv2, lo2, hi2), q8_pair(v.z, inv2, lo2, hi2), 0x06040200u);
        *reinterpret_cast<uint2*>(q + p * qcs + qco + k * 8) = o;
    }
}

inline int grid_for(size_t total, int block, int cap = 256 * 8) {
    size_t g = (total + block - 1) / block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

unsigned half2_bits(float v) {
    const _Float16 h = (_Float16)v;
    unsigned short b;
    memcpy(&b, &h, 2);
    return (unsigned)b | ((unsigned)b << 16);
}

struct AbsmaxOp {
    y6_tensor x;
    float* out;
};
int absmax_launch(const AbsmaxOp* d, hipStream_t s) {
    const y6_tensor& x = d->x;
    Y6_REQUIRE(x.data && d->out, "absmax: null argument");
    Y6_REQUIRE(x.C % 8 == 0 && x.cstride % 8 == 0 && x.coff % 8 == 0 && ((uintptr_t)x.data & 15) == 0,
               "absmax: the view needs 8-channel alignment");
    const size_t npix = (size_t)x.B * x.H * x.W;
    if (npix == 0) return Y6_OK;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(npix * (x.C / 8), 256)), dim3(256), 0, s, (const __half*)x.data, npix, x.C,
                       x.cstride, x.coff, reinterpret_cast<unsigned*>(d->out));
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

struct QuantOp {
    y6_tensor x, q;
    float amax;
};
int quant_launch(const QuantOp* d, hipStream_t s) {
    const y6_tensor &x = d->x, &q = d->q;
    Y6_REQUIRE(x.data && q.data && d->amax > 0.f, "quantize_i8: null argument / non-positive amax");
    Y6_REQUIRE(x.B == q.B && x.H == q.H && x.W == q.W && x.C == q.C, "quantize_i8: shapes differ");
    Y6_REQUIRE(x.C % 8 == 0 && x.cstride % 8 == 0 && x.coff % 8 == 0 && ((uintptr_t)x.data & 15) == 0 && q.cstride % 8 == 0 &&
                   q.coff % 8 == 0 && ((uintptr_t)q.data & 7) == 0,
               "quantize_i8: the views need 8-channel alignment");
    const size_t npix = (size_t)x.B * x.H * x.W;
    if (npix == 0) return Y6_OK;
    const float ah = (float)(_Float16)d->amax;
    hipLaunchKernelGGL(quantize_kernel, dim3(grid_for(npix * (x.C / 8), 256)), dim3(256), 0, s, (const __half*)x.data, npix, x.C,
                       x.cstride, x.coff, (signed char*)q.data, q.cstride, q.coff, half2_bits(127.0f / ah), half2_bits(-ah),
                       half2_bits(ah));
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

double i8_bytes(const y6_conv_i8_desc* q) {
    const y6_conv_desc& d = q->conv;
    const y6_tensor& in = q->q_in.data ? q->q_in : d.in;
    const y6_tensor& out = d.out.data ? d.out : q->q_out;
    double by = (double)in.B * in.H * in.W * in.C * (q->q_in.data ? 1.0 : 2.0);
    if (d.out.data) by += 2.0 * out.B * out.H * out.W * out.C;
    if (q->q_out.data) by += 1.0 * out.B * out.H * out.W * out.C;
    by += (double)out.C * in.C * d.ksize * d.ksize;
    return by;
}
double i8_flops(const y6_conv_i8_desc* q) {
    const y6_conv_desc& d = q->conv;
    const y6_tensor& in = q->q_in.data ? q->q_in : d.in;
    const y6_tensor& out = d.out.data ? d.out : q->q_out;
    return 2.0 * out.B * out.H * out.W * (double)out.C * in.C * d.ksize * d.ksize;
}

}  // namespace

extern "C" size_t y6_packed_weight_i8_bytes(int Cout, int Cin, int K) {
    const size_t cfr_pad = (size_t)y6_cdiv(y6_cdiv(Cout, 32), 4) * 4;
    return cfr_pad * y6_cdiv(Cin, 64) * K * K * 2048;
}

extern "C" int y6_pack_conv_weight_i8(const void* src, int Cout, int Cin, int K, void* dst, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(src && dst && Cout > 0 && Cin > 0 && (K == 1 || K == 3), "pack_conv_weight_i8: bad arguments");
    const int cfr_pad = y6_cdiv(y6_cdiv(Cout, 32), 4) * 4, nchunk = y6_cdiv(Cin, 64);
    const size_t total = (size_t)cfr_pad * nchunk * K * K * 2048;
    hipLaunchKernelGGL(pack_i8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const signed char*)src, Cout,
                       Cin, K, cfr_pad, nchunk, (signed char*)dst);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

extern "C" int y6_absmax(const y6_tensor* x, float* out, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(x, "absmax: null argument");
    AbsmaxOp d{*x, out};
    return absmax_launch(&d, (hipStream_t)stream);
}

extern "C" int y6_quantize_i8(const y6_tensor* x, float amax, const y6_tensor* q, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(x && q, "quantize_i8: null argument");
    QuantOp d{*x, *q, amax};
    return quant_launch(&d, (hipStream_t)stream);
}

extern "C" int y6_conv2d_i8(const y6_conv_i8_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d, "conv2d_i8: null descriptor");
    return y6_conv_i8_launch(d, (hipStream_t)stream);
}

extern "C" int y6_conv2d_i8_variant(const y6_conv_i8_desc* d) {
    if (!d) return 0;
    return y6_conv_i8_variant(d);
}

extern "C" int y6_plan_add_conv_i8(y6_plan* p, const y6_conv_i8_desc* d) {
    Y6_REQUIRE(p && d, "plan_add_conv_i8: null argument");
    return y6_plan_push(p, y6_conv_i8_launch, d, Y6_TOP_CONV_I8, i8_flops(d), i8_bytes(d));
}

extern "C" int y6_plan_add_absmax(y6_plan* p, const y6_tensor* x, float* out) {
    Y6_REQUIRE(p && x && out, "plan_add_absmax: null argument");
    AbsmaxOp d{*x, out};
    return y6_plan_push(p, absmax_launch, &d, Y6_TOP_ABSMAX, 0.0, 2.0 * x->B * x->H * x->W * (double)x->C);
}

extern "C" int y6_plan_add_quantize_i8(y6_plan* p, const y6_tensor* x, float amax, const y6_tensor* q) {
    Y6_REQUIRE(p && x && q, "plan_add_quantize_i8: null argument");
    QuantOp d{*x, *q, amax};
    return y6_plan_push(p, quant_launch, &d, Y6_TOP_QUANT, 0.0, 3.0 * x->B * x->H * x->W * (double)x->C);
}
