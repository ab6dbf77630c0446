// conv_misc.hip — the small kernels around the MFMA conv: weight packing, the naive
// cross-check conv, the NCHW stem conv, ConvTranspose2d(k2,s2), SPPF pooling and the
// NCHW<->NHWC boundary adapters.  File:line citations are into the reference tree.
#include "common.hpp"
#include <cstring>
#include "conv_common.hpp"
#include "plan_internal.hpp"
#include "stem_piece.hpp"

namespace {

// ------------------------------------------------------------------ weight packing
// dst[cfr][chunk][tap][ks][lane][j] = W[cout = cfr*32 + (lane&31)][cin = chunk*32 + ks*16 + (lane>>5)*8 + j][tap]
template <typename T>
__global__ void pack_conv_weight_kernel(const T* __restrict__ src, int Cout, int Cin, int K, int cfr_pad, int nchunk,
                                        __half* __restrict__ dst) {
    const int NT = K * K;
    const size_t total = (size_t)cfr_pad * nchunk * NT * 1024;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const int ks = (int)((i >> 9) & 1);
        size_t r = i >> 10;
        const int tap = (int)(r % NT);
        r /= NT;
        const int chunk = (int)(r % nchunk);
        const int cfr = (int)(r / nchunk);
        const int cout = cfr * 32 + (lane & 31);
        const int cin = chunk * 32 + ks * 16 + (lane >> 5) * 8 + j;
        float v = 0.f;
        if (cout < Cout && cin < Cin) v = (float)src[((size_t)cout * Cin + cin) * NT + tap];
        dst[i] = __float2half(v);
    }
}

// ConvTranspose2d weight is IOHW [Cin][Cout][2][2]; sub-kernel (dy,dx) is a 1x1 conv.
// fused (Cout % 32 == 0): ONE packed matrix of 4*Cout rows, row = sub*Cout + c, sub = dy*2+dx;
// otherwise four separately padded packed 1x1 weights.
template <typename T>
__global__ void pack_convt_weight_kernel(const T* __restrict__ src, int Cin, int Cout, int cfr_pad, int nchunk,
                                         int fused, __half* __restrict__ dst) {
    const size_t per = (size_t)cfr_pad * nchunk * 1024;
    const size_t total = fused ? per : per * 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int sub = fused ? 0 : (int)(i / per);
        const size_t ii = fused ? i : i - (size_t)sub * per;
        const int j = (int)(ii & 7);
        const int lane = (int)((ii >> 3) & 63);
        const int ks = (int)((ii >> 9) & 1);
        size_t r = ii >> 10;
        const int chunk = (int)(r % nchunk);
        const int cfr = (int)(r / nchunk);
        int cout = cfr * 32 + (lane & 31);
        bool ok = cout < Cout;
        if (fused) {
            sub = cout / Cout;
            ok = cout < 4 * Cout;
            cout -= sub * Cout;
        }
        const int cin = chunk * 32 + ks * 16 + (lane >> 5) * 8 + j;
        float v = 0.f;
        if (ok && cin < Cin) v = (float)src[((size_t)cin * Cout + cout) * 4 + sub];
        dst[i] = __float2half(v);
    }
}

// ------------------------------------------------------------------ naive conv (cross-check / any shape)
__global__ void conv_naive_kernel(const __half* __restrict__ in, __half* __restrict__ out,
                                  const __half* __restrict__ w /*OIHW*/, const float* __restrict__ bias,
                                  const float* __restrict__ pscale, const float* __restrict__ pshift,
                                  const __half* __restrict__ res, const float* __restrict__ res_alpha, int B, int H,
                                  int W, int Ho, int Wo, int Cin, int Cout, int in_cs, int in_co, int out_cs,
                                  int out_co, int res_cs, int res_co, int K, int S, int act) {
    const size_t total = (size_t)B * Ho * Wo * Cout;
    const int pad = K / 2;
    const float ra = (res && res_alpha) ? *res_alpha : 1.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        size_t p = i / Cout;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float acc = 0.f;
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * S - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * S - pad + kx;
                if (ix < 0 || ix >= W) continue;
                const __half* ip = in + ((size_t)(b * H + iy) * W + ix) * in_cs + in_co;
                const __half* wp = w + (size_t)co * Cin * K * K + ky * K + kx;
                for (int ci = 0; ci < Cin; ++ci) acc += __half2float(ip[ci]) * __half2float(wp[(size_t)ci * K * K]);
            }
        }
        if (bias) acc += bias[co];
        if (pscale) acc = y6_round_f16(acc) * pscale[co] + pshift[co];
        acc = y6_act(acc, act);
        const size_t op = ((size_t)(b * Ho + oy) * Wo + ox);
        if (res) acc = y6_round_f16(acc) + y6_round_f16(ra * __half2float(res[op * res_cs + res_co + co]));
        out[op * out_cs + out_co + co] = __float2half(acc);
    }
}

// ------------------------------------------------------------------ stem conv (NCHW in, NHWC out)
// One thread = one output pixel x CO couts.  Weights are read with wave-uniform indices
// (scalar loads); the 27 input taps are per-lane loads of neighbouring NCHW elements.
// Input element of the caller's image as the fp16 value the reference's model sees: fp16 / fp32 tensors as they are,
// uint8 pixels as `imgs.half() / 255` (core/evaler.py:121-123) = the fp16 rounding of u / 255 - computed in the kernel, so
// the uint8 batch is read once (1 byte per element) and no converted copy is ever written.
template <typename TI>
__device__ __forceinline__ float stem_in(TI v) { return (float)v; }
template <>
__device__ __forceinline__ float stem_in<uint8_t>(uint8_t v) { return (float)(_Float16)((float)v / 255.f); }

template <typename TI, int CO>
__global__ __launch_bounds__(256) void stem_conv_kernel(const TI* __restrict__ in, __half* __restrict__ out,
                                                        const float* __restrict__ w /*[CO][Cin][3][3]*/,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ pscale,
                                                        const float* __restrict__ pshift, int B, int Cin, int H, int W,
                                                        int Ho, int Wo, int out_cs, int out_co, int act) {
    const size_t total = (size_t)B * Ho * Wo;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % Wo);
    const size_t p = i / Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = bias ? bias[c] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                float v = 0.f;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = stem_in<TI>(in[(((size_t)b * Cin + ci) * H + iy) * W + ix]);
                const float* wp = w + (ci * 3 + ky) * 3 + kx;
#pragma unroll
                for (int c = 0; c < CO; ++c) acc[c] = fmaf(v, wp[(size_t)c * Cin * 9], acc[c]);
            }
        }
    }
    __half* op = out + i * out_cs + out_co;
#pragma unroll
    for (int c0 = 0; c0 < CO; c0 += 8) {
        h8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = acc[c0 + j];
            if (pscale) x = y6_round_f16(x) * pscale[c0 + j] + pshift[c0 + j];
            o[j] = (_Float16)y6_act(x, act);
        }
        *reinterpret_cast<h8_t*>(op + c0) = o;
    }
}

// ------------------------------------------------------------------ stem conv on the matrix cores
// The stem is a [pixels x 27] x [27 x Cout] GEMM (Cin = 3): K is padded to 32 = two
// v_mfma_f32_32x32x16_f16 k-steps.  A operand = weights (row = cout), B operand = the 3x3x3 patch of
// one output pixel (k = ci*9 + ky*3 + kx, exactly the OIHW flattening).
// One block = a 4 x 64 output tile (one wave per output row, 64 pixels each).  The 9 x 129 input
// window of each channel is read from the caller's NCHW image with coalesced loads (consecutive
// lanes = consecutive pixels of a row), converted to fp16 and staged in LDS; patches are then
// gathered from LDS (bank-conflict free: stride-2 halves across lanes).  v1 gathered the patches
// straight from HBM with 2-byte loads and was latency bound at 0.36 ms (roofline 0.07 ms).
constexpr int STEM_TOH = 4, STEM_TOW = 64;
constexpr int STEM_IH = 2 * STEM_TOH + 1, STEM_IW = 2 * STEM_TOW + 1, STEM_PITCH = STEM_IW + 1;  // 9 x 129 (+1 pad)

template <typename TI, int CF>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const TI* __restrict__ in, __half* __restrict__ out,
                                                        const float* __restrict__ w /*[Cout][Cin*9]*/,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ pscale,
                                                        const float* __restrict__ pshift, int B, int Cin, int H, int W,
                                                        int Ho, int Wo, int Cout, int out_cs, int out_co, int act,
                                                        int tiles_x, int tiles_y) {
    constexpr int RS = CF * 64 + 16;  // epilogue row pitch (bytes)
    __shared__ __attribute__((aligned(16))) char s_mem[(3 * STEM_IH * STEM_PITCH * 2 > 4 * 64 * RS)
                                                           ? 3 * STEM_IH * STEM_PITCH * 2
                                                           : 4 * 64 * RS];
    _Float16* s_in = reinterpret_cast<_Float16*>(s_mem);   // [ci][STEM_IH][STEM_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5;
    const int K = Cin * 9;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * STEM_TOH, ox0 = tx * STEM_TOW;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const size_t HW = (size_t)H * W;

    // stage the input window (zero outside the image = conv padding)
    const int nwin = Cin * STEM_IH * STEM_IW;
    for (int i = tid; i < nwin; i += 256) {
        const int ci = i / (STEM_IH * STEM_IW);
        const int r = i - ci * (STEM_IH * STEM_IW);
        const int yy = r / STEM_IW, xx = r - yy * STEM_IW;
        const int iy = iy0 + yy, ix = ix0 + xx;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = stem_in<TI>(in[((size_t)b * Cin + ci) * HW + (size_t)iy * W + ix]);
        s_in[(ci * STEM_IH + yy) * STEM_PITCH + xx] = (_Float16)v;
    }

    // weights: A fragments (cout = cf*32 + lane&31, k = ks*16 + kh*8 + j)
    h8_t af[CF][2];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int co = cf * 32 + (lane & 31), k = ks * 16 + kh * 8 + j;
                af[cf][ks][j] = (co < Cout && k < K) ? (_Float16)w[(size_t)co * K + k] : (_Float16)0.f;
            }
    __syncthreads();

    f32x16_t acc[CF][2];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cf][pf][r] = 0.f;

#pragma unroll
    for (int pf = 0; pf < 2; ++pf) {
        const int px = pf * 32 + (lane & 31);             // output column inside the tile; row = wave
        const _Float16* base = s_in + (2 * wave) * STEM_PITCH + 2 * px;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8_t bf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = ks * 16 + kh * 8 + j;
                const int ci = k / 9, r = k - ci * 9, ky = r / 3, kx = r - ky * 3;
                bf[j] = (k < K) ? base[(ci * STEM_IH + ky) * STEM_PITCH + kx] : (_Float16)0.f;
            }
#pragma unroll
            for (int cf = 0; cf < CF; ++cf)
                acc[cf][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cf][ks], bf, acc[cf][pf], 0, 0, 0);
        }
    }
    __syncthreads();   // every wave is done with the input window: the LDS becomes the output staging area

    // epilogue: C/D col = pixel (lane&31), row = cout = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // The 64 pixels of a wave are consecutive NHWC rows of one image row: stage the [64][CF*32] fp16
    // tile in the wave's LDS region and write it out as whole rows, 16 bytes per lane.
    char* tile = s_mem + wave * 64 * RS;
    const int oy = oy0 + wave;
    const bool rows_ok = (Cout % 8 == 0) && (out_cs % 8 == 0) && (out_co % 8 == 0);
    const size_t rowbase = ((size_t)b * Ho + oy) * Wo;     // first pixel of this image row
#pragma unroll
    for (int pf = 0; pf < 2; ++pf) {
        const int px = pf * 32 + (lane & 31);
        const bool pvalid = oy < Ho && (ox0 + px) < Wo;
        __half* orow = out + (rowbase + ox0 + px) * out_cs + out_co;
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c0 = cf * 32 + 8 * r4 + 4 * kh;
                h4_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + j;
                    float x = acc[cf][pf][r4 * 4 + j];
                    if (c < Cout) {
                        if (bias) x += bias[c];
                        if (pscale) x = y6_round_f16(x) * pscale[c] + pshift[c];
                        x = y6_act(x, act);
                    }
                    o[j] = (_Float16)x;
                }
                if (rows_ok) {
                    *reinterpret_cast<h4_t*>(tile + px * RS + c0 * 2) = o;
                } else if (pvalid && c0 < Cout) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j < Cout) orow[c0 + j] = (__half)o[j];
                }
            }
    }
    if (rows_ok) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int PPR = CF * 4;   // 16-byte pieces per pixel row
#pragma unroll
        for (int i = 0; i < PPR; ++i) {
            const int q = lane + 64 * i;
            const int px = q / PPR, pc = q - px * PPR;
            if (oy < Ho && (ox0 + px) < Wo && pc * 8 + 8 <= Cout)
                *reinterpret_cast<uint4*>(out + (rowbase + ox0 + px) * out_cs + out_co + pc * 8) =
                    *reinterpret_cast<const uint4*>(tile + px * RS + pc * 16);
        }
    }
}

// ---- stem v4: same tile (4 x 64 outputs) and MFMA mapping, but
//   * the input window is fetched as aligned 16-byte pieces (8 pixels of one NCHW row): 9 rows x 3 channels x
//     17 pieces per tile instead of 3483 two-byte loads;
//   * blocks are persistent: weights / bias live in registers for the whole kernel, and the NEXT tile's window
//     is requested before the current tile is multiplied and written, so the HBM latency is off the path.
// Needs W % 8 == 0 and a 16-byte aligned image (anything else takes the kernel above).
constexpr int STEM4_WPC = 17;                       // 16-byte pieces per window row: cols [2*ox0-8, 2*ox0+128)
constexpr int STEM4_PITCH = STEM4_WPC * 8 + 8;      // halves
constexpr int STEM4_NPIECE = 3 * STEM_IH * STEM4_WPC;

template <int ACT>
__device__ __forceinline__ float stem_act(float v) {
    return y6_act(v, ACT);
}

template <typename TI, int CF>
__global__ __launch_bounds__(256) void stem_mfma_v4_kernel(const TI* __restrict__ in, __half* __restrict__ out,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ pscale,
                                                           const float* __restrict__ pshift, int B, int H, int W, int Ho,
                                                           int Wo, int Cout, int out_cs, int out_co, int act, int tiles_x,
                                                           int tiles_y, signed char* __restrict__ qout, int q_cs, int q_co, unsigned q_inv2,
                                                           unsigned q_lo2, unsigned q_hi2) {
    constexpr int RS = CF * 64 + 16;   // epilogue row pitch (bytes)
    __shared__ __attribute__((aligned(16))) _Float16 s_in[3 * STEM_IH * STEM4_PITCH];
    __shared__ __attribute__((aligned(16))) char s_out[4 * 64 * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5;
    constexpr int K = 27;
    const int ntiles = tiles_x * tiles_y * B;
    const size_t HW = (size_t)H * W;

    // this thread's two window pieces: (channel, window row, piece column) - tile independent
    int pc_ci[2], pc_yy[2], pc_px[2];
    bool pc_on[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = tid + i * 256;
        pc_on[i] = p < STEM4_NPIECE;
        const int pp = pc_on[i] ? p : 0;
        pc_ci[i] = pp / (STEM_IH * STEM4_WPC);
        const int r = pp - pc_ci[i] * (STEM_IH * STEM4_WPC);
        pc_yy[i] = r / STEM4_WPC;
        pc_px[i] = r - pc_yy[i] * STEM4_WPC;
    }
    StemPiece<TI> pre[2];
    auto request = [&](int tile) {
        const int tx = tile % tiles_x;
        const int t2 = tile / tiles_x;
        const int ty = t2 % tiles_y;
        const int b = t2 / tiles_y;
        const int iy0 = 2 * ty * STEM_TOH - 1, cx0 = 2 * tx * STEM_TOW - 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int iy = iy0 + pc_yy[i], ix = cx0 + pc_px[i] * 8;
            if (pc_on[i] && iy >= 0 && iy < H && ix >= 0 && ix < W)   // W % 8 == 0: a piece is wholly in or out
                pre[i].load(in + ((size_t)b * 3 + pc_ci[i]) * HW + (size_t)iy * W + ix);
            else
                pre[i].zero();
        }
    };

    // weights: A fragments (cout = cf*32 + lane&31, k = ks*16 + kh*8 + j), bias / post-affine of this lane's couts
    h8_t af[CF][2];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int co = cf * 32 + (lane & 31), k = ks * 16 + kh * 8 + j;
                af[cf][ks][j] = (co < Cout && k < K) ? (_Float16)w[(size_t)co * K + k] : (_Float16)0.f;
            }
    float bz[CF][16], psc[CF][16], psh[CF][16];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = cf * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
            const bool ok = c < Cout;
            bz[cf][r] = (ok && bias) ? bias[c] : 0.f;
            psc[cf][r] = (ok && pscale) ? pscale[c] : 1.f;
            psh[cf][r] = (ok && pscale) ? pshift[c] : 0.f;
        }
    const bool affine = pscale != nullptr;
    const bool rows_ok = (Cout % 8 == 0) && (out_cs % 8 == 0) && (out_co % 8 == 0);

    int tile = blockIdx.x;
    if (tile < ntiles) request(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (pc_on[i])
                *reinterpret_cast<uint4*>(s_in + (pc_ci[i] * STEM_IH + pc_yy[i]) * STEM4_PITCH + pc_px[i] * 8) =
                    pre[i].as_half8();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);   // in flight during the MFMAs and the stores below

        const int tx = tile % tiles_x;
        const int t2 = tile / tiles_x;
        const int ty = t2 % tiles_y;
        const int b = t2 / tiles_y;
        const int oy0 = ty * STEM_TOH, ox0 = tx * STEM_TOW;

        f32x16_t acc[CF][2];
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cf][pf][r] = 0.f;
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int px = pf * 32 + (lane & 31);   // output column inside the tile; output row = wave
            const _Float16* base = s_in + (2 * wave) * STEM4_PITCH + 2 * px + 7;   // window col of (kx = 0)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8_t bf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = ks * 16 + kh * 8 + j;   // kh is per-lane: both alternatives are compile-time offsets
                    const int k0 = ks * 16 + j, k1 = ks * 16 + 8 + j;
                    const int o0 = ((k0 / 9) * STEM_IH + (k0 % 9) / 3) * STEM4_PITCH + (k0 % 3);
                    const int o1 = ((k1 / 9) * STEM_IH + (k1 % 9) / 3) * STEM4_PITCH + (k1 % 3);
                    const _Float16 v0 = (k0 < K) ? base[o0] : (_Float16)0.f;
                    const _Float16 v1 = (k1 < K) ? base[o1] : (_Float16)0.f;
                    bf[j] = kh ? v1 : v0;
                    (void)k;
                }
#pragma unroll
                for (int cf = 0; cf < CF; ++cf)
                    acc[cf][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cf][ks], bf, acc[cf][pf], 0, 0, 0);
            }
        }

        // epilogue: C/D col = pixel (lane&31), row = cout = (r&3) + 8*(r>>2) + 4*kh -> wave-private LDS tile -> rows
        char* tl = s_out + wave * 64 * RS;
        const int oy = oy0 + wave;
        const size_t rowbase = ((size_t)b * Ho + oy) * Wo;
        auto finish = [&](auto actfn) {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const int px = pf * 32 + (lane & 31);
                const bool pvalid = oy < Ho && (ox0 + px) < Wo;
                __half* orow = out + (rowbase + ox0 + px) * out_cs + out_co;
#pragma unroll
                for (int cf = 0; cf < CF; ++cf)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int c0 = cf * 32 + 8 * r4 + 4 * kh;
                        h4_t o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float x = acc[cf][pf][r4 * 4 + j] + bz[cf][r4 * 4 + j];
                            if (affine) x = y6_round_f16(x) * psc[cf][r4 * 4 + j] + psh[cf][r4 * 4 + j];
                            o[j] = (_Float16)actfn(x);
                        }
                        if (rows_ok) {
                            *reinterpret_cast<h4_t*>(tl + px * RS + c0 * 2) = o;
                        } else if (pvalid) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (c0 + j < Cout) orow[c0 + j] = (__half)o[j];
                        }
                    }
            }
        };
        switch (act) {   // one uniform branch per tile instead of one per element
            case Y6_ACT_RELU: finish([](float v) { return stem_act<Y6_ACT_RELU>(v); }); break;
            case Y6_ACT_SILU: finish([](float v) { return stem_act<Y6_ACT_SILU>(v); }); break;
            case Y6_ACT_HARDSWISH: finish([](float v) { return stem_act<Y6_ACT_HARDSWISH>(v); }); break;
            default: finish([](float v) { return v; }); break;
        }
        if (rows_ok) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int PPR = CF * 4;   // 16-byte pieces per pixel row
            if (out != nullptr) {
#pragma unroll
                for (int i = 0; i < PPR; ++i) {
                    const int q = lane + 64 * i;
                    const int px = q / PPR, pcq = q - px * PPR;
                    if (oy < Ho && (ox0 + px) < Wo && pcq * 8 + 8 <= Cout)
                        *reinterpret_cast<uint4*>(out + (rowbase + ox0 + px) * out_cs + out_co + pcq * 8) =
                            *reinterpret_cast<const uint4*>(tl + px * RS + pcq * 16);
                }
            }
            if (qout != nullptr) {   // the int8 twin: the SAME fp16 values, 16 channels per 16-byte piece (round 6)
                constexpr int QPR = CF * 2;
#pragma unroll
                for (int i = 0; i < QPR; ++i) {
                    const int q = lane + 64 * i;
                    const int px = q / QPR, pcq = q - px * QPR;
                    if (oy < Ho && (ox0 + px) < Wo && pcq * 16 + 16 <= Cout) {
                        const uint4 lo = *reinterpret_cast<const uint4*>(tl + px * RS + pcq * 32);
                        const uint4 hi = *reinterpret_cast<const uint4*>(tl + px * RS + pcq * 32 + 16);
                        *reinterpret_cast<uint4*>(qout + (rowbase + ox0 + px) * q_cs + q_co + pcq * 16) = q8_piece(lo, hi, q_inv2, q_lo2, q_hi2);
                    }
                }
            }
        }
        __syncthreads();   // window reads and tile reads are over: the next iteration overwrites both
    }
}

// ------------------------------------------------------------------ SPPF: 3 chained 5x5 s1 p2 max pools
// One block = one image x 8 channels; the HxW plane lives in LDS; each pool is a
// separable row-max / column-max pass with -inf padding (nn.MaxPool2d semantics).
// (Round 4, r04a: a packed-maximum form - four v_pk_max_f16 per eight channels - measured 32 -> 31 us: the kernel is bound by its
// 16-byte-per-pixel global accesses, not by the compares; removed.)
__device__ __forceinline__ h8_t hmax8(const h8_t& v, const h8_t& m) {
    h8_t r = m;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j] > m[j] ? v[j] : m[j];
    return r;
}

// int8 twins of the three pooled slices (y6_sppf_pool_q): q[i] == nullptr: none
struct SppfTwins {
    signed char* q[3];
    int cs[3], co[3];
    unsigned inv2, lo2, hi2;   // half2 constants of the quantiser (conv_common.hpp q8_quad)
};

// CG: 16-byte pieces (8 channels) of a pixel a block owns.  Round 6: CG = 4 - a block reads and writes 64 contiguous bytes per pixel
// instead of 16 (one piece per block touched every 128-byte line of the 20x20x256 map from eight different blocks: 32 us for 26 MB), and
// the blocks of one image share an XCD (id % 8), so the pieces of a line meet in one L2.
template <int CG>
__global__ __launch_bounds__(1024) void sppf_pool_kernel(const __half* __restrict__ x, int x_cs, int x_co,
                                                        __half* __restrict__ y1, int y1_cs, int y1_co,
                                                        __half* __restrict__ y2, int y2_cs, int y2_co,
                                                        __half* __restrict__ y3, int y3_cs, int y3_co, int H, int W,
                                                        int ncg, int B, SppfTwins tw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h8_t* cur = reinterpret_cast<h8_t*>(smem);
    h8_t* tmp = cur + H * W * CG;
    // id -> (image, channel group): groups of one image on one XCD
    const int id = blockIdx.x, lo = id & 7, r = id >> 3;
    const int cg = r % ncg, b = (r / ncg) * 8 + lo;
    if (b >= B) return;
    const int HW = H * W, N = HW * CG;
    const size_t pbase = (size_t)b * HW;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int p = i / CG, j = i - p * CG;
        cur[i] = *reinterpret_cast<const h8_t*>(x + (pbase + p) * x_cs + x_co + (cg * CG + j) * 8);
    }
    __syncthreads();
    __half* outs[3] = {y1, y2, y3};
    const int ocs[3] = {y1_cs, y2_cs, y3_cs};
    const int oco[3] = {y1_co, y2_co, y3_co};
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) {  // row max
            const int p = i / CG, j = i - p * CG;
            const int yy = p / W, xx = p - yy * W;
            h8_t m = cur[i];
            for (int d = -2; d <= 2; ++d) {
                const int x2 = xx + d;
                if (d == 0 || x2 < 0 || x2 >= W) continue;
                m = hmax8(cur[(yy * W + x2) * CG + j], m);
            }
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) {  // column max
            const int p = i / CG, j = i - p * CG;
            const int yy = p / W, xx = p - yy * W;
            h8_t m = tmp[i];
            for (int d = -2; d <= 2; ++d) {
                const int y2i = yy + d;
                if (d == 0 || y2i < 0 || y2i >= H) continue;
                m = hmax8(tmp[(y2i * W + xx) * CG + j], m);
            }
            *reinterpret_cast<h8_t*>(outs[pass] + (pbase + p) * ocs[pass] + oco[pass] + (cg * CG + j) * 8) = m;
            if (tw.q[pass] != nullptr) {   // the same eight values as int8 codes
                const uint4 mb = __builtin_bit_cast(uint4, m);
                *reinterpret_cast<uint2*>(tw.q[pass] + (pbase + p) * tw.cs[pass] + tw.co[pass] + (cg * CG + j) * 8) =
                    make_uint2(q8_quad(mb.x, mb.y, tw.inv2, tw.lo2, tw.hi2), q8_quad(mb.z, mb.w, tw.inv2, tw.lo2, tw.hi2));
            }
            // safe to overwrite cur[i]: the column pass reads tmp only
            cur[i] = m;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ layout adapters
template <typename T>
__global__ void nchw_to_nhwc_kernel(const T* __restrict__ src, __half* __restrict__ dst, int B, int C, int H, int W,
                                    int cs, int co) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;  // (b*H + y)*W + x
        const size_t hw = (size_t)H * W;
        const size_t b = p / hw, yx = p % hw;
        dst[p * cs + co + c] = __float2half((float)src[(b * C + c) * hw + yx]);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const __half* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W,
                                    int cs, int co) {
    const size_t total = (size_t)B * H * W * C;
    const size_t hw = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t yx = i % hw;
        const size_t bc = i / hw;
        const size_t c = bc % C, b = bc / C;
        dst[i] = (T)__half2float(src[(b * hw + yx) * cs + co + c]);
    }
}

inline int grid_for(size_t total, int block, int cap = 256 * 16) {
    size_t g = (total + block - 1) / block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" size_t y6_packed_weight_elems(int Cout, int Cin, int K) {
    const size_t cfr_pad = (size_t)y6_cdiv(y6_cdiv(Cout, 32), 4) * 4;
    return cfr_pad * y6_cdiv(Cin, 32) * K * K * 1024;
}

extern "C" int y6_pack_conv_weight(const void* src, int src_dtype, int Cout, int Cin, int K, void* dst, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(src && dst && Cout > 0 && Cin > 0 && (K == 1 || K == 3), "pack_conv_weight: bad arguments");
    const int cfr_pad = y6_cdiv(y6_cdiv(Cout, 32), 4) * 4, nchunk = y6_cdiv(Cin, 32);
    const size_t total = (size_t)cfr_pad * nchunk * K * K * 1024;
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == Y6_F16)
        hipLaunchKernelGGL(pack_conv_weight_kernel<__half>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                           (const __half*)src, Cout, Cin, K, cfr_pad, nchunk, (__half*)dst);
    else if (src_dtype == Y6_F32)
        hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                           (const float*)src, Cout, Cin, K, cfr_pad, nchunk, (__half*)dst);
    else
        Y6_REQUIRE(false, "pack_conv_weight: bad dtype %d", src_dtype);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

extern "C" int y6_pack_convt2x2_weight(const void* src, int src_dtype, int Cin, int Cout, void* dst, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(src && dst && Cout > 0 && Cin > 0, "pack_convt2x2_weight: bad arguments");
    const int fused = (Cout % 32) == 0;
    const int rows = fused ? 4 * Cout : Cout;
    const int cfr_pad = y6_cdiv(y6_cdiv(rows, 32), 4) * 4, nchunk = y6_cdiv(Cin, 32);
    const size_t total = (size_t)cfr_pad * nchunk * 1024 * (fused ? 1 : 4);
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == Y6_F16)
        hipLaunchKernelGGL(pack_convt_weight_kernel<__half>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                           (const __half*)src, Cin, Cout, cfr_pad, nchunk, fused, (__half*)dst);
    else if (src_dtype == Y6_F32)
        hipLaunchKernelGGL(pack_convt_weight_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                           (const float*)src, Cin, Cout, cfr_pad, nchunk, fused, (__half*)dst);
    else
        Y6_REQUIRE(false, "pack_convt2x2_weight: bad dtype %d", src_dtype);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

int y6_conv_naive_launch(const y6_conv_desc* d, hipStream_t s) {
    Y6_REQUIRE(d->w_oihw != nullptr, "conv naive: w_oihw not provided");
    const size_t total = (size_t)d->out.B * d->out.H * d->out.W * d->out.C;
    hipLaunchKernelGGL(conv_naive_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s,
                       (const __half*)d->in.data, (__half*)d->out.data, (const __half*)d->w_oihw, d->bias,
                       d->post_scale, d->post_shift, (const __half*)d->res.data, d->res_alpha, d->in.B, d->in.H,
                       d->in.W, d->out.H, d->out.W, d->in.C, d->out.C, d->in.cstride, d->in.coff, d->out.cstride,
                       d->out.coff, d->res.cstride, d->res.coff, d->ksize, d->stride, d->act);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

static int check_conv_desc(const y6_conv_desc* d) {
    Y6_REQUIRE(d && d->in.data && d->out.data, "conv2d: null tensor");
    Y6_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv2d: ksize %d unsupported (1 or 3)", d->ksize);
    Y6_REQUIRE(d->stride == 1 || d->stride == 2, "conv2d: stride %d unsupported", d->stride);
    const int pad = d->ksize / 2;
    const int Ho = (d->in.H + 2 * pad - d->ksize) / d->stride + 1, Wo = (d->in.W + 2 * pad - d->ksize) / d->stride + 1;
    Y6_REQUIRE(d->out.B == d->in.B && d->out.H == Ho && d->out.W == Wo, "conv2d: output shape [%d,%d,%d] != expected [%d,%d,%d]",
               d->out.B, d->out.H, d->out.W, d->in.B, Ho, Wo);
    Y6_REQUIRE(d->in.coff + d->in.C <= d->in.cstride && d->out.coff + d->out.C <= d->out.cstride,
               "conv2d: channel slice out of range");
    if (d->res.data)
        Y6_REQUIRE(d->res.B == d->out.B && d->res.H == d->out.H && d->res.W == d->out.W && d->res.C == d->out.C,
                   "conv2d: residual shape mismatch");
    Y6_REQUIRE((d->post_scale == nullptr) == (d->post_shift == nullptr), "conv2d: post_scale/post_shift must come together");
    return Y6_OK;
}

// Variant of a plan that was not autotuned: a function of the layer's shape only, so that two processes (two ranks of a
// training job, two runs of a regression test) run the same kernels and produce the same bits - an autotuned plan picks by
// timing, and timing is per process (the run-to-run differences of the training loss were this).
static int variant_by_name(const char* name) {
    static int n = y6_conv_variants();
    for (int v = 1; v < n; ++v)
        if (strcmp(y6_conv_variant_name(v), name) == 0) return v;
    return -1;
}
int y6_conv_default_variant(const y6_conv_desc* d) {
    const long px = (long)d->out.B * d->out.H * d->out.W;
    if (d->ksize == 3) {
        // round 4 (profiles/r04): the register-fed kernels win wherever they apply (Cin % 32 == 0, Cout % 128 == 0); stride 2: 3
        // fragments.  Fragments per wave from the item count of the 224-pixel form (round 6, the timed tables of the b32 inference and
        // the b64 training plans agree - profiles/r06/autotune_train_r06zj.log): 5 fragments (160-pixel items) when even the large
        // items make several rounds over the 512 block slots, 7 (224 pixels: the most MFMAs per weight fragment) while they still
        // reach most CUs, 4 (128 pixels) for the small maps.  E.g. 256->256 @40x40: b64 916 items -> p5 (98 us against 106 p7),
        // b32 458 -> p7; 128->128 @40x40 b32 229 -> p7; 256->256 @20x20 b32 116 -> p4.
        static const int p7 = variant_by_name("wreg_p7"), p5 = variant_by_name("wreg_p5"), p4 = variant_by_name("wreg_p4"),
                         s2p3 = variant_by_name("wregs2_p3");
        const long items224 = (px + 223) / 224 * ((d->out.C + 127) / 128);
        const int first = d->stride == 1 ? (items224 >= 900 ? p5 : (items224 >= 200 ? p7 : p4)) : s2p3;
        if (first > 0 && y6_conv_variant_supports(d, first)) return first;
        if (d->stride == 1 && first == p5 && p7 > 0 && y6_conv_variant_supports(d, p7)) return p7;
        if (d->stride == 1 && p4 > 0 && y6_conv_variant_supports(d, p4)) return p4;
    }
    if (d->ksize == 1 && d->stride == 1) {
        // round 6: the whole-reduction 1x1 kernel (conv_pw.hip) wherever it applies - 128-cout blocks, else 64-cout blocks.
        // Y6_CONV_PW=0: A/B switch (the per-tap kernel of round 1)
        static const bool pw_on = getenv("Y6_CONV_PW") ? atoi(getenv("Y6_CONV_PW")) != 0 : true;
        static const int pw4 = variant_by_name("pw_c4p2"), pw2 = variant_by_name("pw_c2p2");
        if (pw_on && pw4 > 0 && y6_conv_variant_supports(d, pw4)) return pw4;
        if (pw_on && pw2 > 0 && y6_conv_variant_supports(d, pw2)) return pw2;
    }
    // then (profiles/r02 autotune logs) the LDS-DMA kernels on 3x3 stride 1 - 256-pixel blocks when there are at least two of
    // them per CU slot, else 128-pixel blocks - then round 1's pipelined kernel; elsewhere high-occupancy small tiles win
    const long items256 = (px + 255) / 256 * ((d->out.C + 63) / 64);
    static const int dma22 = variant_by_name("dma_c2p2"), dma21 = variant_by_name("dma_c2p1"), pipe22 = variant_by_name("pipe_c2p2"),
                     pipe21 = variant_by_name("pipe_c2p1"), pipe12 = variant_by_name("pipe_c1p2"), c2p1 = variant_by_name("mfma_c2p1"), c1p1 = variant_by_name("mfma_c1p1"),
                     c2p2 = variant_by_name("mfma_c2p2"), c1p2 = variant_by_name("mfma_c1p2"), c4p1 = variant_by_name("mfma_c4p1"),
                     c4p2 = variant_by_name("mfma_c4p2");
    // (round 6: 256-pixel blocks only for the very large maps - 64->64 @160x160 b64: 149 against 156 us; @80x80 b64 and @160x160 b32
    // the 128-pixel blocks win by 3-5 us)
    const int dma_a = items256 >= 4096 ? dma22 : dma21, dma_b = items256 >= 4096 ? dma21 : dma22;
    const int prefs_s1[] = {dma_a, dma_b, pipe22, pipe21, pipe12, c2p1, c1p1, c2p2, c1p2, c4p1, c4p2, 0};
    static const int dma8s2 = variant_by_name("dma8s2_c2p1");     // round 6: the LDS-DMA stride-2 form first (32->64 @320x320 b64: 182 against 213 us)
    const int prefs_s2[] = {dma8s2, c2p1, c1p1, c4p1, 0};
    const int* prefs = d->stride == 1 ? prefs_s1 : prefs_s2;
    const int n = d->stride == 1 ? 12 : 5;
    for (int i = 0; i < n; ++i)
        if (prefs[i] >= 0 && y6_conv_variant_supports(d, prefs[i])) return prefs[i];
    return -1;
}

extern "C" int y6_conv_variant_supports(const y6_conv_desc* d, int i) {
    if (!d) return 0;
    if (i == 0) return d->w_oihw != nullptr;
    return y6_conv_mfma_supports(d, i);
}

extern "C" int y6_conv2d(const y6_conv_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    int rc = check_conv_desc(d);
    if (rc) return rc;
    int v = d->variant;
    if (v < 0) v = y6_conv_default_variant(d);
    Y6_REQUIRE(v >= 0, "conv2d: no kernel variant supports this conv (k%d s%d Cin %d Cout %d)", d->ksize, d->stride,
               d->in.C, d->out.C);
    if (y6_sync_trace()) {
        fprintf(stderr, "[y6-sync-trace] conv2d k%d s%d cin %d cout %d in %dx%dx%d (cs %d co %d) out cs %d co %d res %d act %d variant %d (%s) in %p out %p w %p\n",
                d->ksize, d->stride, d->in.C, d->out.C, d->in.B, d->in.H, d->in.W, d->in.cstride, d->in.coff, d->out.cstride, d->out.coff,
                d->res.data != nullptr, d->act, v, y6_conv_variant_name(v), d->in.data, d->out.data, d->w_packed);
        fflush(stderr);
    }
    if (v == 0) return y6_conv_naive_launch(d, (hipStream_t)stream);
    return y6_conv_mfma_launch(d, v, (hipStream_t)stream, 0, 0, 0);
}

double y6_conv_flops(const y6_conv_desc* d) {
    return 2.0 * d->out.B * d->out.H * d->out.W * (double)d->out.C * d->in.C * d->ksize * d->ksize;
}
double y6_conv_bytes(const y6_conv_desc* d) {
    // algorithmic: read input once, write output once, weights once (fp16) + bias
    return 2.0 * ((double)d->in.B * d->in.H * d->in.W * d->in.C + (double)d->out.B * d->out.H * d->out.W * d->out.C +
                  (double)d->out.C * d->in.C * d->ksize * d->ksize) +
           4.0 * d->out.C + (d->res.data ? 2.0 * d->out.B * d->out.H * d->out.W * d->out.C : 0.0);
}

// variant for a convT (sub-)GEMM: small pixel tiles win (autotune table, profiles/r01); the cout block
// must not straddle two (dy,dx) sub-kernels in the fused form
static int convt_variant(const y6_conv_desc* c, int upC, int fused) {
    if (fused) {   // round 6: the whole-reduction 1x1 kernel scatters whole cout blocks (conv_pw.hip); Y6_CONV_PW=0: A/B switch
        static const bool pw_on = getenv("Y6_CONV_PW") ? atoi(getenv("Y6_CONV_PW")) != 0 : true;
        static const int pw4 = variant_by_name("pw_c4p2"), pw2 = variant_by_name("pw_c2p2");
        if (pw_on && pw4 > 0 && upC % 128 == 0 && y6_conv_mfma_supports(c, pw4) && y6_tensor_elems(c->out) * 8 < 0xe0000000ull) return pw4;
        if (pw_on && pw2 > 0 && upC % 64 == 0 && y6_conv_mfma_supports(c, pw2) && y6_tensor_elems(c->out) * 8 < 0xe0000000ull) return pw2;
    }
    const int prefs[] = {2, 1, 5, 4, 3, 6};
    const int cfs[] = {0, 1, 2, 4, 1, 2, 4};
    for (int i = 0; i < 6; ++i) {
        const int v = prefs[i];
        if (fused && (upC % (cfs[v] * 32)) != 0) continue;
        if (y6_conv_mfma_supports(c, v)) return v;
    }
    return -1;
}

extern "C" int y6_convt2x2(const y6_convt_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d && d->in.data && d->out.data && d->w_packed, "convt2x2: null argument");
    Y6_REQUIRE(d->out.B == d->in.B && d->out.H == 2 * d->in.H && d->out.W == 2 * d->in.W, "convt2x2: output must be 2x input");
    const int Cout = d->out.C;
    y6_conv_desc c;
    memset(&c, 0, sizeof(c));
    c.in = d->in;
    c.out = d->out;
    // the 1x1 (sub-)conv runs over the INPUT grid; the kernel scatters to (2y+dy, 2x+dx)
    c.out.H = d->in.H;
    c.out.W = d->in.W;
    c.bias = d->bias;
    c.ksize = 1;
    c.stride = 1;
    c.act = Y6_ACT_NONE;
    if (Cout % 32 == 0) {  // one launch: the four sub-kernels are cout blocks of one [4*Cout x Cin] GEMM
        c.out.C = 4 * Cout;
        c.w_packed = d->w_packed;
        const int v = convt_variant(&c, Cout, 1);
        Y6_REQUIRE(v > 0, "convt2x2: unsupported shape Cin %d Cout %d", d->in.C, Cout);
        return y6_conv_mfma_launch(&c, v, (hipStream_t)stream, 2, 0, 0);
    }
    const size_t per = y6_packed_weight_elems(Cout, d->in.C, 1);
    for (int sub = 0; sub < 4; ++sub) {
        c.w_packed = (const __half*)d->w_packed + sub * per;
        const int v = convt_variant(&c, Cout, 0);
        Y6_REQUIRE(v > 0, "convt2x2: unsupported shape Cin %d Cout %d", d->in.C, Cout);
        int rc = y6_conv_mfma_launch(&c, v, (hipStream_t)stream, 1, sub >> 1, sub & 1);
        if (rc) return rc;
    }
    return Y6_OK;
}

static unsigned stem_half2_bits(float v) {
    const _Float16 h = (_Float16)v;
    unsigned short b;
    memcpy(&b, &h, 2);
    return (unsigned)b | ((unsigned)b << 16);
}

// the tiled kernel (stem_mfma_v4_kernel) takes the call: the only one that writes the int8 twin
static bool stem_v4_ok(const y6_stem_desc* d) {
    const size_t esz = d->in_dtype == Y6_F16 ? 2 : (d->in_dtype == Y6_U8 ? 1 : 4);
    const size_t palign = 8 * esz;
    static const bool no_v4 = getenv("Y6_STEM_NO_V4") != nullptr;   // A/B switch for profiling
    return d->Cin * 9 <= 32 && d->out.C <= 64 && d->out.cstride % 4 == 0 && d->out.coff % 4 == 0 && !no_v4 && d->Cin == 3 && d->W % 8 == 0 &&
           ((uintptr_t)d->in_nchw % palign) == 0 && ((size_t)d->H * d->W * esz) % palign == 0 &&
           (d->in_dtype == Y6_F16 || d->in_dtype == Y6_F32 || d->in_dtype == Y6_U8);
}

extern "C" int y6_stem_twin_supported(const y6_stem_desc* d) {
    if (!d || !d->in_nchw) return 0;
    const y6_tensor& q = d->q_out;
    return stem_v4_ok(d) && d->out.C % 16 == 0 && d->out.cstride % 8 == 0 && d->out.coff % 8 == 0 && q.C == d->out.C && q.cstride % 16 == 0 &&
                   q.coff % 16 == 0 && q.B == d->out.B && q.H == d->out.H && q.W == d->out.W
               ? 1
               : 0;
}

extern "C" int y6_stem_conv(const y6_stem_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d && d->in_nchw && (d->out.data || d->q_out.data) && d->w_oihw_f32, "stem_conv: null argument");
    Y6_REQUIRE(d->q_out.data == nullptr || (y6_stem_twin_supported(d) && (((uintptr_t)d->q_out.data) & 15) == 0 && d->q_out_amax > 0.f),
               "stem_conv: the int8 twin needs the tiled kernel's shapes (3 input channels, W %% 8 == 0, Cout %% 16 == 0, aligned views)");
    const int Ho = (d->H + 2 - 3) / 2 + 1, Wo = (d->W + 2 - 3) / 2 + 1;
    Y6_REQUIRE(d->out.B == d->B && d->out.H == Ho && d->out.W == Wo, "stem_conv: bad output shape");
    Y6_REQUIRE(d->out.cstride % 8 == 0 && d->out.coff % 8 == 0, "stem_conv: output slice must be 8-channel aligned");
    const int CO = d->out.C;
    const size_t total = (size_t)d->B * Ho * Wo;
    hipStream_t s = (hipStream_t)stream;
    if (d->Cin * 9 <= 32 && CO <= 64 && d->out.cstride % 4 == 0 && d->out.coff % 4 == 0) {
        const int tiles_x = y6_cdiv(Wo, STEM_TOW), tiles_y = y6_cdiv(Ho, STEM_TOH);
        dim3 g((unsigned)(tiles_x * tiles_y * d->B)), blk(256);
#define Y6_STEM_MFMA(TI, CF_)                                                                                     \
    hipLaunchKernelGGL((stem_mfma_kernel<TI, CF_>), g, blk, 0, s, (const TI*)d->in_nchw, (__half*)d->out.data,      \
                       d->w_oihw_f32, d->bias, d->post_scale, d->post_shift, d->B, d->Cin, d->H, d->W, Ho, Wo, CO, \
                       d->out.cstride, d->out.coff, d->act, tiles_x, tiles_y)
        if (stem_v4_ok(d)) {
            static int n_cu = 0;
            if (n_cu == 0) {
                int dev = 0;
                Y6_HIP(hipGetDevice(&dev));
                Y6_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
            }
            const int ntiles = tiles_x * tiles_y * d->B;
            int gridp = n_cu * 4;
            if (gridp > ntiles) gridp = ntiles;
#define Y6_STEM_V4(TI, CF_)                                                                                        \
    hipLaunchKernelGGL((stem_mfma_v4_kernel<TI, CF_>), dim3(gridp), blk, 0, s, (const TI*)d->in_nchw,                \
                       (__half*)d->out.data, d->w_oihw_f32, d->bias, d->post_scale, d->post_shift, d->B, d->H, d->W, \
                       Ho, Wo, CO, d->out.cstride, d->out.coff, d->act, tiles_x, tiles_y, (signed char*)d->q_out.data,           \
                       d->q_out.cstride, d->q_out.coff, q_inv2, q_lo2, q_hi2)
            unsigned q_inv2 = 0, q_lo2 = 0, q_hi2 = 0;
            if (d->q_out.data) {   // fp16(amax) and fp16(127 / fp16(amax)): the quantiser constants of y6_conv_i8_desc
                const float ah = (float)(_Float16)d->q_out_amax;
                q_inv2 = stem_half2_bits(127.0f / ah), q_lo2 = stem_half2_bits(-ah), q_hi2 = stem_half2_bits(ah);
            }
            if (d->in_dtype == Y6_F16) {
                if (CO <= 32) Y6_STEM_V4(__half, 1); else Y6_STEM_V4(__half, 2);
            } else if (d->in_dtype == Y6_U8) {
                if (CO <= 32) Y6_STEM_V4(uint8_t, 1); else Y6_STEM_V4(uint8_t, 2);
            } else {
                if (CO <= 32) Y6_STEM_V4(float, 1); else Y6_STEM_V4(float, 2);
            }
#undef Y6_STEM_V4
            Y6_LAUNCH_CHECK();
            return Y6_OK;
        }
        Y6_REQUIRE(d->out.data && !d->q_out.data, "stem_conv: only the tiled kernel writes the int8 twin");
        if (d->in_dtype == Y6_F16) {
            if (CO <= 32) Y6_STEM_MFMA(__half, 1); else Y6_STEM_MFMA(__half, 2);
        } else if (d->in_dtype == Y6_F32) {
            if (CO <= 32) Y6_STEM_MFMA(float, 1); else Y6_STEM_MFMA(float, 2);
        } else if (d->in_dtype == Y6_U8) {
            if (CO <= 32) Y6_STEM_MFMA(uint8_t, 1); else Y6_STEM_MFMA(uint8_t, 2);
        } else {
            Y6_REQUIRE(false, "stem_conv: bad input dtype");
        }
#undef Y6_STEM_MFMA
        Y6_LAUNCH_CHECK();
        return Y6_OK;
    }
    Y6_REQUIRE(d->out.data && !d->q_out.data, "stem_conv: only the tiled kernel writes the int8 twin");
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define Y6_STEM_CASE(TI, CO_)                                                                                      \
    hipLaunchKernelGGL((stem_conv_kernel<TI, CO_>), grid, block, 0, s, (const TI*)d->in_nchw, (__half*)d->out.data, \
                       d->w_oihw_f32, d->bias, d->post_scale, d->post_shift, d->B, d->Cin, d->H, d->W, Ho, Wo,      \
                       d->out.cstride, d->out.coff, d->act)
    if (d->in_dtype == Y6_F16) {
        switch (CO) {
            case 8: Y6_STEM_CASE(__half, 8); break;
            case 16: Y6_STEM_CASE(__half, 16); break;
            case 32: Y6_STEM_CASE(__half, 32); break;
            case 48: Y6_STEM_CASE(__half, 48); break;
            case 64: Y6_STEM_CASE(__half, 64); break;
            default: Y6_REQUIRE(false, "stem_conv: Cout %d unsupported (8,16,32,48,64)", CO);
        }
    } else if (d->in_dtype == Y6_U8) {
        switch (CO) {
            case 8: Y6_STEM_CASE(uint8_t, 8); break;
            case 16: Y6_STEM_CASE(uint8_t, 16); break;
            case 32: Y6_STEM_CASE(uint8_t, 32); break;
            case 48: Y6_STEM_CASE(uint8_t, 48); break;
            case 64: Y6_STEM_CASE(uint8_t, 64); break;
            default: Y6_REQUIRE(false, "stem_conv: Cout %d unsupported (8,16,32,48,64)", CO);
        }
    } else if (d->in_dtype == Y6_F32) {
        switch (CO) {
            case 8: Y6_STEM_CASE(float, 8); break;
            case 16: Y6_STEM_CASE(float, 16); break;
            case 32: Y6_STEM_CASE(float, 32); break;
            case 48: Y6_STEM_CASE(float, 48); break;
            case 64: Y6_STEM_CASE(float, 64); break;
            default: Y6_REQUIRE(false, "stem_conv: Cout %d unsupported (8,16,32,48,64)", CO);
        }
    } else {
        Y6_REQUIRE(false, "stem_conv: bad input dtype");
    }
#undef Y6_STEM_CASE
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

static unsigned sppf_half2_bits(float v) {
    const _Float16 h = (_Float16)v;
    unsigned short b;
    memcpy(&b, &h, 2);
    return (unsigned)b | ((unsigned)b << 16);
}

static int sppf_launch(const y6_sppf_q_desc* d, hipStream_t stream) {
    Y6_REQUIRE(d, "sppf_pool: null descriptor");
    const y6_tensor *x = &d->x, *y1 = &d->y1, *y2 = &d->y2, *y3 = &d->y3;
    Y6_REQUIRE(x->data && y1->data && y2->data && y3->data, "sppf_pool: null tensor");
    Y6_REQUIRE(x->C % 8 == 0 && x->coff % 8 == 0 && x->cstride % 8 == 0, "sppf_pool: channels must be 8-aligned");
    const y6_tensor* ys[3] = {y1, y2, y3};
    const y6_tensor* qs[3] = {&d->q1, &d->q2, &d->q3};
    SppfTwins tw;
    memset(&tw, 0, sizeof(tw));
    for (int i = 0; i < 3; ++i) {
        Y6_REQUIRE(ys[i]->B == x->B && ys[i]->H == x->H && ys[i]->W == x->W && ys[i]->C == x->C &&
                       ys[i]->coff % 8 == 0 && ys[i]->cstride % 8 == 0,
                   "sppf_pool: output %d shape/alignment mismatch", i);
        if (qs[i]->data) {
            Y6_REQUIRE(qs[i]->B == x->B && qs[i]->H == x->H && qs[i]->W == x->W && qs[i]->C == x->C && qs[i]->coff % 8 == 0 &&
                           qs[i]->cstride % 8 == 0 && (((uintptr_t)qs[i]->data) & 7) == 0 && d->q_amax > 0.f,
                       "sppf_pool: int8 twin %d shape/alignment mismatch (or q_amax <= 0)", i);
            tw.q[i] = (signed char*)qs[i]->data;
            tw.cs[i] = qs[i]->cstride;
            tw.co[i] = qs[i]->coff;
        }
    }
    if (d->q_amax > 0.f) {   // fp16(amax) and fp16(127 / fp16(amax)): the quantiser constants of y6_conv_i8_desc
        const float ah = (float)(_Float16)d->q_amax;
        tw.inv2 = sppf_half2_bits(127.0f / ah);
        tw.lo2 = sppf_half2_bits(-ah);
        tw.hi2 = sppf_half2_bits(ah);
    }
    const int C8 = x->C / 8;
    // pieces per pixel and block: four (64 bytes) when the two planes fit 64 KiB of LDS and the channels divide, else two, else one
    int cgp = 4;
    while (cgp > 1 && (C8 % cgp || (size_t)x->H * x->W * 16 * 2 * cgp > 64 * 1024)) cgp >>= 1;
    const size_t lds = (size_t)x->H * x->W * 16 * 2 * cgp;
    Y6_REQUIRE(lds <= 160 * 1024, "sppf_pool: plane %dx%d too large for LDS", x->H, x->W);
    const int ncg = C8 / cgp;
    const int grid = y6_cdiv(x->B, 8) * 8 * ncg;
    auto launch = [&](auto kern) {
        if (lds > 64 * 1024) Y6_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // one thread per (pixel, piece) where the plane allows: the six passes are LDS round trips, not bytes
        const int threads = x->H * x->W * cgp >= 1024 ? 1024 : (x->H * x->W * cgp >= 512 ? 512 : 256);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, (const __half*)x->data,
                           x->cstride, x->coff, (__half*)y1->data, y1->cstride, y1->coff, (__half*)y2->data, y2->cstride,
                           y2->coff, (__half*)y3->data, y3->cstride, y3->coff, x->H, x->W, ncg, x->B, tw);
        return Y6_OK;
    };
    int rc = cgp == 4 ? launch(sppf_pool_kernel<4>) : (cgp == 2 ? launch(sppf_pool_kernel<2>) : launch(sppf_pool_kernel<1>));
    if (rc) return rc;
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

extern "C" int y6_sppf_pool_q(const y6_sppf_q_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return sppf_launch(d, (hipStream_t)stream);
}
extern "C" int y6_sppf_pool(const y6_tensor* x, const y6_tensor* y1, const y6_tensor* y2, const y6_tensor* y3,
                            void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(x && y1 && y2 && y3, "sppf_pool: null tensor");
    y6_sppf_q_desc d;
    memset(&d, 0, sizeof(d));
    d.x = *x;
    d.y1 = *y1;
    d.y2 = *y2;
    d.y3 = *y3;
    return sppf_launch(&d, (hipStream_t)stream);
}
extern "C" int y6_plan_add_sppf_q(y6_plan* p, const y6_sppf_q_desc* d) {
    Y6_REQUIRE(p && d, "plan_add_sppf_q: null argument");
    const double px = (double)d->x.B * d->x.H * d->x.W * d->x.C;
    const int nq = (d->q1.data != nullptr) + (d->q2.data != nullptr) + (d->q3.data != nullptr);
    return y6_plan_push(p, sppf_launch, d, Y6_TOP_SPPF_Q, 0.0, px * (2.0 * 4 + nq));
}

extern "C" int y6_nchw_to_nhwc(const void* src, int src_dtype, const y6_tensor* dst, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(src && dst && dst->data, "nchw_to_nhwc: null argument");
    const size_t total = (size_t)dst->B * dst->H * dst->W * dst->C;
    dim3 grid(grid_for(total, 256)), block(256);
    if (src_dtype == Y6_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<__half>, grid, block, 0, (hipStream_t)stream, (const __half*)src,
                           (__half*)dst->data, dst->B, dst->C, dst->H, dst->W, dst->cstride, dst->coff);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)src,
                           (__half*)dst->data, dst->B, dst->C, dst->H, dst->W, dst->cstride, dst->coff);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

extern "C" int y6_nhwc_to_nchw(const y6_tensor* src, void* dst, int dst_dtype, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(src && dst && src->data, "nhwc_to_nchw: null argument");
    const size_t total = (size_t)src->B * src->H * src->W * src->C;
    dim3 grid(grid_for(total, 256)), block(256);
    if (dst_dtype == Y6_F16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<__half>, grid, block, 0, (hipStream_t)stream, (const __half*)src->data,
                           (__half*)dst, src->B, src->C, src->H, src->W, src->cstride, src->coff);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, block, 0, (hipStream_t)stream, (const __half*)src->data,
                           (float*)dst, src->B, src->C, src->H, src->W, src->cstride, src->coff);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
