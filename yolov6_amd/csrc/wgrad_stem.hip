// wgrad_stem.hip - the weight gradients of the stem block's convs in ONE pass over their operands (round 6, last session).
//
//   dW3[m][c][ky][kx] += sum_{b,y,x} dy3(b,y,x,m) * X(b, c, 2y+ky-1, 2x+kx-1)        3x3 stride 2 pad 1
//   dW1[m][c]         += sum_{b,y,x} dy1(b,y,x,m) * X(b, c, 2y, 2x)                  1x1 stride 2 (the RepVGG stem's second branch)
//
// Replaces the weight half of autograd's conv backward for the convs that read the caller's NCHW image (reference: the stem of
// EfficientRep / CSPBepBackbone, yolov6/models/efficientrep.py:28-41 - a RepVGGBlock(3 -> C, k3 s2) in train form, its forward
// yolov6/layers/common.py:250-255; core/engine.py:173 `.backward()`).
//
// Why its own kernel.  The generic route fed the plane-based GEMM (wgrad.hip) with channel-major copies of BOTH gradients (2 x 420 MB
// read + written for YOLOv6-S b64: 310 us) and five sampled planes of the image, then ran two GEMMs whose 27 / 3 reduction columns
// fill a fraction of a tile: 0.87 ms of a 33 ms step for 12.6 GFLOP.  Here
//   * the GEMM is [32 couts] x [32 columns n = c*9 + ky*3 + kx (27 live)] per wave, reduction over output pixels; the 1x1 conv's three
//     columns ARE the 3x3 conv's centre-tap columns (same image element), so ONE image operand serves two MFMAs (dy3 and dy1);
//   * dy arrives as it lies: an NHWC row of a 32-channel gradient is the [position][32 channels] image ds_read_b64_tr_b16 turns into
//     MFMA operands (8 consecutive positions of one channel per lane) - a global load and a ds_write_b128 per 16 bytes, the next
//     item's pieces in flight in registers while this one is multiplied;
//   * the image operand comes straight from the NCHW tensor: lane (n, half) needs 8 consecutive output columns of tap (c, ky, kx) =
//     every second element of 16 consecutive fp16 of image row 2y+ky-1 - two 16-byte loads (4 bytes early for kx = 0) and four
//     shift/mask packs; column -1 and row -1 are zeroed by hand (row H and column W never occur: H, W even);
//   * an item is (image, output row, 320-column segment): 20 k-steps over four waves; a block walks items block, block + grid, ...
//     so neighbouring blocks share image rows in L2; accumulators stay in registers over all items;
//   * per block one [2][32][32] fp32 partial, added by a second small kernel in a fixed order (deterministic, no float atomics).
// HBM-bound: the two gradients + the image once = 1.0 GB for S b64 (about 200 us at 5 TB/s).
#include "common.hpp"
#include "plan_internal.hpp"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

constexpr int SEG = 320;            // output columns per item
constexpr int KSW = SEG / 16 / 4;   // k-steps per wave and item
constexpr int NPC = SEG * 4 / 256;  // 16-byte pieces of one gradient image per thread
constexpr unsigned kOobS = 0xf0000000u;

struct StemWgArgs {
    const __half* x;
    unsigned x_bytes;
    const __half* dy3;
    const __half* dy1;       // nullptr: a single 3x3 conv (ConvBNSiLU stems)
    unsigned dy3_bytes, dy1_bytes;
    int cs3, co3, cs1, co1;  // pixel pitch / channel offset of the gradient views (halves)
    int B, Cin, H, W, Ho, Wo, Cout;
    int nseg, items;
    float* part;             // [cout tiles][gridDim.x][2][32][32]
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 tr_run8(const char* p) {   // 8 consecutive positions of one channel (wgrad_flat.hip tr_run)
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 256));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 r;
    r[0] = l2[0], r[1] = l2[1], r[2] = h2[0], r[3] = h2[1];
    return r;
}

__global__ __launch_bounds__(256, 2) void wgrad_stem_kernel(const StemWgArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * SEG * 64];
    char* const img3 = smem;
    char* const img1 = smem + SEG * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int ct = blockIdx.y;
    const int vch = a.Cout - 32 * ct < 32 ? a.Cout - 32 * ct : 32;
    const bool has1 = a.dy1 != nullptr;
    const __amdgpu_buffer_rsrc_t rs3 = rsrc_of(a.dy3, a.dy3_bytes), rs1 = rsrc_of(a.dy1, has1 ? a.dy1_bytes : 0u), rsx = rsrc_of(a.x, a.x_bytes);

    // this lane's column of the image operand
    const int n = lane & 31;
    const bool nlive = n < a.Cin * 9;
    const int ci = n / 9, ky = (n - ci * 9) / 3, kx = n - ci * 9 - ky * 3;
    const unsigned sh = kx == 1 ? 0u : 16u;      // even / odd elements of the 16 loaded
    const int early = kx == 0 ? 4 : 0;           // kx = 0 starts one element (of the odd ones: 4 bytes) earlier

    f32x16_t acc3, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc3[q] = acc1[q] = 0.f;

    const int lane_off = ((lane & 15) >> 2) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2 + half * 8 * 64;

    u32x4 p3[NPC], p1[NPC];
    auto fetch = [&](int item) {   // the item's gradient pieces -> registers (zeros outside the row / the tile's channels)
        const int seg = item % a.nseg, row = item / a.nseg;          // row = b * Ho + y
        const int wv = a.Wo - seg * SEG;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int q = tid + 256 * i;
            const int pos = q >> 2, pc = q & 3;
            const bool v = pos < wv && 8 * pc < vch;
            const unsigned pix = (unsigned)row * (unsigned)a.Wo + (unsigned)(seg * SEG + pos);
            p3[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs3, (int)(v ? (pix * (unsigned)a.cs3 + (unsigned)(a.co3 + 32 * ct + 8 * pc)) * 2u : kOobS), 0, 0));
            if (has1)
                p1[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)(v ? (pix * (unsigned)a.cs1 + (unsigned)(a.co1 + 32 * ct + 8 * pc)) * 2u : kOobS), 0, 0));
        }
    };

    int item = blockIdx.x;
    if (item < a.items) fetch(item);
    for (; item < a.items; item += gridDim.x) {
        const int seg = item % a.nseg, row = item / a.nseg;
        const int b = row / a.Ho, y = row - b * a.Ho;
        __syncthreads();                                   // the previous item's operand reads are finished
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int q = tid + 256 * i;
            *reinterpret_cast<u32x4*>(img3 + q * 16) = p3[i];
            if (has1) *reinterpret_cast<u32x4*>(img1 + q * 16) = p1[i];
        }
        // this wave's image operands: k-step ks covers output columns seg*SEG + 16 ks + 8 half .. + 7 of row (b, y)
        u32x4 xa[KSW], xb[KSW];
        const int r_in = 2 * y + ky - 1;
        const bool rowok = nlive && r_in >= 0;
        const unsigned rowbyte = (unsigned)((((size_t)b * a.Cin + ci) * a.H + (r_in < 0 ? 0 : r_in)) * a.W) * 2u;
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int ks = wave + 4 * j;
            const int x0 = seg * SEG + 16 * ks + 8 * half;
            const bool v = rowok && x0 < a.Wo;
            // (the row's first group of a kx = 0 lane cannot start 4 bytes early: the offset would be negative for the tensor's first
            // row, and a wrapped offset is out of range for the WHOLE request - it loads in place and shifts by one element below)
            const unsigned off = v ? rowbyte + (unsigned)(4 * x0) - (unsigned)(x0 == 0 ? 0 : early) : kOobS;
            xa[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)off, 0, 0));
            xb[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)(v ? off + 16u : kOobS), 0, 0));
        }
        __syncthreads();                                   // the gradient images are complete
        if (item + (int)gridDim.x < a.items) fetch(item + gridDim.x);   // the next item's pieces travel while this one is multiplied
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int ks = wave + 4 * j;
            if (seg * SEG + 16 * ks >= a.Wo) break;        // (wave-uniform) nothing but zeros behind the row's end
            const u32x4 A3 = tr_run8(img3 + lane_off + ks * 16 * 64);
            u32x4 A1;
            if (has1) A1 = tr_run8(img1 + lane_off + ks * 16 * 64);
            u32x4 bq;
            bq[0] = ((xa[j][0] >> sh) & 0xffffu) | ((xa[j][1] >> sh) << 16);
            bq[1] = ((xa[j][2] >> sh) & 0xffffu) | ((xa[j][3] >> sh) << 16);
            bq[2] = ((xb[j][0] >> sh) & 0xffffu) | ((xb[j][1] >> sh) << 16);
            bq[3] = ((xb[j][2] >> sh) & 0xffffu) | ((xb[j][3] >> sh) << 16);
            if (kx == 0 && seg == 0 && ks == 0 && half == 0) {   // (h1 h3)(h5 h7).. -> (0 h1)(h3 h5)..: column -1 is the zero pad
                bq[3] = (bq[3] << 16) | (bq[2] >> 16);
                bq[2] = (bq[2] << 16) | (bq[1] >> 16);
                bq[1] = (bq[1] << 16) | (bq[0] >> 16);
                bq[0] = bq[0] << 16;
            }
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, A3), __builtin_bit_cast(h8_t, bq), acc3, 0, 0, 0);
            if (has1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, A1), __builtin_bit_cast(h8_t, bq), acc1, 0, 0, 0);
        }
    }
    // ---- the block's partial: the four waves' tiles added in wave order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);           // [wave][2][32][32]
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int m = 8 * (q >> 2) + 4 * half + (q & 3);
        red[(wave * 2 + 0) * 1024 + m * 32 + n] = acc3[q];
        red[(wave * 2 + 1) * 1024 + m * 32 + n] = acc1[q];
    }
    __syncthreads();
    float* dst = a.part + ((size_t)ct * gridDim.x + blockIdx.x) * 2048;
    for (int o = tid; o < 2048; o += 256) dst[o] = ((red[o] + red[2048 + o]) + red[4096 + o]) + red[6144 + o];
}

// block = 32 outputs x 8 interleaved sub-sums over the blocks' partials (sixteen loads in flight), added in sub-sum order
__global__ __launch_bounds__(256) void wgrad_stem_sum_kernel(const float* __restrict__ part, int nblocks, int Cout, int Cin, float* __restrict__ out3,
                                                            float* __restrict__ out1) {
    __shared__ float red[8][33];
    const int t = threadIdx.x, oi = t & 31, sub = t >> 5;
    const int o = blockIdx.x * 32 + oi;                    // (cout tile, conv, m, n)
    const int ct = o >> 11, r = o & 2047;
    const float* p = part + (size_t)ct * nblocks * 2048 + r;
    float s = 0.f;
    int b = sub;
    for (; b + 120 < nblocks; b += 128) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(b + 8 * u) * 2048];
#pragma unroll
        for (int u = 0; u < 16; u += 4) s += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
    }
    for (; b < nblocks; b += 8) s += p[(size_t)b * 2048];
    red[sub][oi] = s;
    __syncthreads();
    if (t < 32) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot += red[q][t];
        const int conv = r >> 10, m = (r >> 5) & 31, n = r & 31;
        const int co = 32 * ct + m;
        if (co < Cout && n < Cin * 9) {
            if (conv == 0)
                out3[(size_t)co * Cin * 9 + n] += tot;
            else if (out1 != nullptr && n % 9 == 4)
                out1[(size_t)co * Cin + n / 9] += tot;
        }
    }
}

bool stem_view_ok(const y6_tensor& t) {
    return t.data && t.C % 8 == 0 && t.cstride % 8 == 0 && t.coff % 8 == 0 && (((uintptr_t)t.data) & 15) == 0;
}

int wgrad_stem_grid(const y6_wgrad_stem_desc* d) {
    const int Ho = d->dy3.H, Wo = d->dy3.W;
    const int nseg = (Wo + SEG - 1) / SEG;
    const long items = (long)d->B * Ho * nseg;
    return (int)(items < 512 ? items : 512);
}

int wgrad_stem_check(const y6_wgrad_stem_desc* d, bool set_error) {
#define STEM_REQ(cond, ...)                        \
    do {                                           \
        if (!(cond)) {                             \
            if (set_error) y6_set_error(__VA_ARGS__); \
            return 0;                              \
        }                                          \
    } while (0)
    STEM_REQ(d && d->x && d->out3, "wgrad_stem: null argument");
    STEM_REQ(d->in_dtype == Y6_F16, "wgrad_stem: fp16 images only");
    STEM_REQ(d->Cin >= 1 && d->Cin <= 3, "wgrad_stem: at most 3 input channels");
    STEM_REQ(d->B > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, "wgrad_stem: even image sizes");
    STEM_REQ((((uintptr_t)d->x) & 3) == 0 && (size_t)d->B * d->Cin * d->H * d->W * 2 < 0xe0000000ull, "wgrad_stem: image alignment / size");
    STEM_REQ(stem_view_ok(d->dy3) && d->dy3.B == d->B && d->dy3.H == d->H / 2 && d->dy3.W == d->W / 2, "wgrad_stem: bad dy3 view");
    STEM_REQ(d->Cout >= 8 && d->Cout % 8 == 0 && d->Cout <= d->dy3.C && d->Cout <= 256, "wgrad_stem: 8..256 couts, multiple of 8");
    STEM_REQ(y6_tensor_elems(d->dy3) * 2 < 0xe0000000ull, "wgrad_stem: gradient too large");
    if (d->dy1.data) {
        STEM_REQ(d->out1 != nullptr, "wgrad_stem: dy1 without out1");
        STEM_REQ(stem_view_ok(d->dy1) && d->dy1.B == d->B && d->dy1.H == d->dy3.H && d->dy1.W == d->dy3.W && d->Cout <= d->dy1.C, "wgrad_stem: bad dy1 view");
        STEM_REQ(y6_tensor_elems(d->dy1) * 2 < 0xe0000000ull, "wgrad_stem: gradient too large");
    }
    const int ntile = (d->Cout + 31) / 32;
    STEM_REQ(d->workspace && d->workspace_bytes >= (size_t)ntile * wgrad_stem_grid(d) * 2048 * sizeof(float), "wgrad_stem: workspace too small");
#undef STEM_REQ
    return 1;
}

int wgrad_stem_launch(const y6_wgrad_stem_desc* d, hipStream_t s) {
    if (!wgrad_stem_check(d, true)) return Y6_EINVAL;
    StemWgArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const __half*)d->x;
    a.x_bytes = (unsigned)((size_t)d->B * d->Cin * d->H * d->W * 2);
    a.dy3 = (const __half*)d->dy3.data;
    a.dy3_bytes = (unsigned)(y6_tensor_elems(d->dy3) * 2);
    a.cs3 = d->dy3.cstride;
    a.co3 = d->dy3.coff;
    if (d->dy1.data) {
        a.dy1 = (const __half*)d->dy1.data;
        a.dy1_bytes = (unsigned)(y6_tensor_elems(d->dy1) * 2);
        a.cs1 = d->dy1.cstride;
        a.co1 = d->dy1.coff;
    }
    a.B = d->B, a.Cin = d->Cin, a.H = d->H, a.W = d->W, a.Ho = d->dy3.H, a.Wo = d->dy3.W, a.Cout = d->Cout;
    a.nseg = (a.Wo + SEG - 1) / SEG;
    a.items = a.B * a.Ho * a.nseg;
    a.part = (float*)d->workspace;
    const int grid = wgrad_stem_grid(d), ntile = (d->Cout + 31) / 32;
    hipLaunchKernelGGL(wgrad_stem_kernel, dim3((unsigned)grid, (unsigned)ntile), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_stem_sum_kernel, dim3((unsigned)(ntile * 64)), dim3(256), 0, s, (const float*)a.part, grid, d->Cout, d->Cin, d->out3,
                       d->dy1.data ? d->out1 : (float*)nullptr);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

}  // namespace

extern "C" int y6_wgrad_stem_supported(const y6_wgrad_stem_desc* d) { return wgrad_stem_check(d, false); }
extern "C" size_t y6_wgrad_stem_workspace_bytes(int Cout) { return (size_t)((Cout + 31) / 32) * 512 * 2048 * sizeof(float); }
extern "C" int y6_wgrad_stem(const y6_wgrad_stem_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return wgrad_stem_launch(d, (hipStream_t)stream);
}
extern "C" int y6_plan_add_wgrad_stem(y6_plan* p, const y6_wgrad_stem_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    if (!wgrad_stem_check(d, true)) return Y6_EINVAL;
    const double pix = (double)d->B * d->dy3.H * d->dy3.W;
    const double flops = 2.0 * pix * d->Cout * d->Cin * (9.0 + (d->dy1.data ? 1.0 : 0.0));
    const double bytes = 2.0 * pix * d->Cout * (d->dy1.data ? 2.0 : 1.0) + 2.0 * d->B * d->Cin * d->H * d->W;
    return y6_plan_push(p, wgrad_stem_launch, d, Y6_TOP_WGRAD, flops, bytes);
}
