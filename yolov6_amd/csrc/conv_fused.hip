// conv_fused.hip — "producer -> 3x3 stride-2 conv" pairs as ONE kernel: the producer's output tile never leaves the CU.
//
// Replaces two aten compositions of the deploy graph whose intermediate tensor is the largest HBM round trip of its stage:
//   * BiFusion.forward `downsample(cv2(x[2]))` (reference yolov6/layers/common.py:711-716): ConvBNReLU 1x1 at the
//     backbone's resolution followed by ConvBNReLU 3x3 stride 2 (YOLOv6-S b32: 64ch @ 160x160, 105 MB written and re-read;
//     128ch @ 80x80, 52 MB);
//   * EfficientRep `ERBlock_2[0](stem(x))` (yolov6/models/efficientrep.py:28-45): the image conv (RepVGG deploy form, 3x3
//     stride 2 on 3 channels) followed by the first stride-2 RepVGG block (210 MB at b32).
// Both pairs are HBM-bound as two launches (1x1: 2.0 TB/s of algorithmic traffic; stem 3.3 TB/s) and MFMA-trivial, so the
// design rule is "touch HBM once": a block owns a TH x TW tile of the stride-2 conv's OUTPUT, produces the (2 TH + 1) x
// (2 TW + 1) producer pixels that tile needs into LDS (a halo of one row / column is recomputed by the neighbouring tile:
// 8-16 % extra producer MFMAs, no extra HBM bytes beyond L2-resident re-reads), then runs the nine taps from LDS.
//
// Mid tile in LDS: planar, one plane per 8 channels, [plane][mid row][slot] x 16 B.  Rows are stored EVEN COLUMNS FIRST
// (slot = mx / 2 for even mx, RPE + mx / 2 for odd mx; RPE = TW + 1 even columns) so that the 16 consecutive output
// pixels a ds_read_b128 group serves (conv_dma.hip: frag_pixel) read 16 CONSECUTIVE slots for every tap - input column
// 2 tx + kx is even slot tx (kx 0), odd slot tx (kx 1), even slot tx + 1 (kx 2): conflict-free, as the stride-2 form of
// the LDS-DMA kernel.  The producer's C/D fragment (pixel = lane & 31) goes to the planes with two ds_write_b128 per
// 32 couts after v_permlane32_swap (conv_common.hpp), one wave store = 512 contiguous bytes per plane.
//
// Producers:
//   PW   1x1 conv: the MFMA pixel operand of a lane IS a 16-byte piece of the NHWC row, loaded straight from global memory
//        (conv1x1_stream_kernel's scheme), the next fragment's pieces in flight during this fragment's MFMAs; a wave keeps
//        the weight fragments of ONE cout fragment in registers for the whole kernel.
//   STEM 3x3 stride-2 conv on the caller's NCHW image (K = 27 -> 32, two k-steps): the image window of the mid tile is
//        staged as aligned 16-byte pieces (stem_mfma_v4_kernel's scheme, any of fp16 / fp32 / uint8).
// Out-of-image producer pixels are ZERO in LDS (they are the stride-2 conv's zero padding, not relu(bias)).
// Consumer: a wave owns one 32-pixel fragment x CFW cout fragments; weight fragments stream from the packed image (L2) through
// a register ring, the pixel operand is one ds_read_b128 per (tap, k-step).  Epilogue: bias + activation (finish16_any), the
// output tile staged through the (then dead) mid tile's LDS and stored as whole NHWC rows.  Same k order (tap-major inside 32-channel chunks ... see `consume`) is NOT the
// per-tap kernels' order, so results agree with the unfused ops to fp32 summation order (<= 1 fp16 ulp), not bit for bit;
// the rounding points (producer output fp16, consumer output fp16) are the unfused graph's.
#include <cstddef>

#include "common.hpp"
#include "conv_common.hpp"
#include "plan_internal.hpp"
#include "stem_piece.hpp"

namespace {

// lane (0..31) -> pixel of the fragment it holds: puts the two 16-lane groups ds_read_b128 serves per cycle
// ({0-3,12-15,20-27} / {4-11,16-19,28-31}) on pixels 0-15 / 16-31 (conv_dma.hip)
__device__ __forceinline__ int fz_frag_pixel(int l) {
    return l < 4 ? l : (l < 12 ? l + 12 : (l < 16 ? l - 8 : (l < 20 ? l + 8 : (l < 28 ? l - 12 : l))));
}

struct FusedArgs {
    ConvKArgs c;              // the consumer (3x3 stride 2): out view, bias, act, Cout, wpk, store flags; H / W = mid dims, Ho / Wo
    int B;
    // 1x1 producer
    const __half* in;
    int in_cs, in_co, Cin;
    const __half* w1;
    const float* b1;
    int act1;
    // image-conv producer
    const void* img;
    int IH, IW;
    const float* wst;         // OIHW fp32 [Cm][3][3][3] (fp16-rounded values)
    // tiling
    int TH, tiles_x, tiles_y, ntiles;
    int dbg;                  // timing probes (env Y6_FUSED_PROBE, WRONG RESULTS): 1 no producer arithmetic, 2 no consumer MFMA loop, 4 no output stores, 8 no input loads
    int MH, RPS, RPE, nslots, PLS;   // mid rows, slots per row (= 2 TW + 1), even columns per row (= TW + 1), MH * RPS, slots per plane (padded to 32)
};

// finished producer fragment (16 accumulators of this lane: pixel lane & 31, couts 8 g + 4 kh + j of cout fragment cfm) -> LDS planes
__device__ __forceinline__ void mid_store(char* mid, int PLS, int cfm, int slot, int lane, const float (&v)[16]) {
    const int kh = lane >> 5;
    unsigned pk[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
            h2_t t;
            t[0] = (_Float16)v[g * 4 + h * 2];
            t[1] = (_Float16)v[g * 4 + h * 2 + 1];
            pk[g][h] = __builtin_bit_cast(unsigned, t);
        }
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
        // lanes 0-31: couts 16 gp .. 16 gp + 7, lanes 32-63: couts 16 gp + 8 .. 16 gp + 15 of their pixel
        const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        *reinterpret_cast<uint4*>(mid + ((size_t)(cfm * 4 + 2 * gp + kh) * PLS + slot) * 16) = o;
    }
}

// producer epilogue arithmetic of one fragment: bias + activation (rounded where the reference's fp16 graph rounds: act_const),
// zero for out-of-image pixels.  ONE wave-uniform branch per fragment: a `switch (act)` per element costs thousands of cycles
// per fragment (conv_common.hpp finish16_any; the first version of these kernels spent 80 % of a tile there).
template <int ACT>
__device__ __forceinline__ void produce_act16(const f32x16_t& acc, const float (&bz)[16], bool valid, float (&v)[16]) {
    // (computed for every lane - the operands of out-of-image lanes are real pixels, finite - then selected: a per-lane
    //  `valid ? f(x) : 0` around SiLU / hardswish compiles to an exec-masked branch per element pair)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float t = act_const<ACT>(acc[r] + bz[r]);
        v[r] = valid ? t : 0.f;
    }
}
__device__ __forceinline__ void produce_act16_any(int act, const f32x16_t& acc, const float (&bz)[16], bool valid, float (&v)[16]) {
    switch (act) {
        case Y6_ACT_RELU: produce_act16<Y6_ACT_RELU>(acc, bz, valid, v); break;
        case Y6_ACT_SILU: produce_act16<Y6_ACT_SILU>(acc, bz, valid, v); break;
        case Y6_ACT_HARDSWISH: produce_act16<Y6_ACT_HARDSWISH>(acc, bz, valid, v); break;
        default: produce_act16<Y6_ACT_NONE>(acc, bz, valid, v); break;
    }
}

// slot (linear index into a mid plane) -> mid row / column, evens first
__device__ __forceinline__ void slot_to_mid(const FusedArgs& a, int slot, int& my, int& mx) {
    my = slot / a.RPS;
    const int ms = slot - my * a.RPS;
    mx = ms < a.RPE ? 2 * ms : 2 * (ms - a.RPE) + 1;
}

// ---- the consumer: 3x3 stride-2 conv of the LDS-resident mid tile; a wave owns PFW pixel fragments x CFW cout fragments.
// The (tap, k-step) sequence runs as a RUNTIME loop over groups of GK k-steps of one tap (fully unrolled, hipcc hoists the
// 36-144 loop-invariant weight addresses out of the tile loop: 256 VGPRs + scratch).  Weight fragments come straight from the
// packed image (L2) through a ring of three groups - two groups (an L2 round trip) ahead of the MFMAs - and every fragment
// feeds PFW MFMAs (a 32-pixel fragment per wave would need 1 KiB of weights per MFMA: twice the vector memory path's rate).
// RESIDENT: the wave's 9 * KSM weight fragments live in registers for the whole kernel (`wres`, loaded by the caller before the
// tile loop; affordable for KSM = 2, the image-conv pair: 72 VGPRs) - the consumer then touches no global memory at all.
template <int KSM, int PFW, int CFW, int TW, bool RESIDENT = false>
__device__ __forceinline__ void consume(const FusedArgs& a, const char* mid, int frag0, int cf0, int lane, int b, int oy0, int ox0,
                                        const h8_t* wres = nullptr) {
    constexpr int GK = KSM < 4 ? KSM : 4;       // k-steps per group
    constexpr int GPT = KSM / GK;               // groups per tap
    constexpr int NG = 9 * GPT;
    constexpr int NCH = (KSM + 1) / 2;          // 32-channel chunks of the packed weight image
    static_assert(KSM % GK == 0 && GK % 2 == 0 && NG % 3 == 0, "k-steps come in pairs (32-channel chunks); the ring has three slots");
    const int q = fz_frag_pixel(lane & 31);
    const int kh = lane >> 5;
    int ty[PFW], tx[PFW], mm[PFW];
    const char* base[PFW];
#pragma unroll
    for (int i = 0; i < PFW; ++i) {
        mm[i] = (frag0 + i) * 32 + q;
        ty[i] = mm[i] / TW;
        tx[i] = mm[i] - ty[i] * TW;
        base[i] = mid + ((size_t)kh * a.PLS + (2 * ty[i]) * a.RPS + tx[i]) * 16;
    }
    const int plane2 = a.PLS * 32;              // bytes between k-steps (two planes)
    const __half* wb = a.c.wpk + (size_t)cf0 * NCH * 9 * 1024 + lane * 8;   // [cfr][chunk][tap][ks][lane][8]
    f32x16_t acc[CFW][PFW];
#pragma unroll
    for (int cf = 0; cf < CFW; ++cf)
#pragma unroll
        for (int i = 0; i < PFW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cf][i][r] = 0.f;
    h8_t w0[GK][CFW], w1[GK][CFW], w2[GK][CFW];
    auto wload = [&](int g, h8_t (&w)[GK][CFW]) {
        const int t = g / GPT, kg = g - t * GPT;                       // tap, k-group (wave-uniform)
        const __half* p = wb + (size_t)(((kg * GK) >> 1) * 9 + t) * 1024;
#pragma unroll
        for (int j = 0; j < GK; ++j)
#pragma unroll
            for (int cf = 0; cf < CFW; ++cf)
                w[j][cf] = *reinterpret_cast<const h8_t*>(p + (size_t)((cf * NCH + (j >> 1)) * 9) * 1024 + (j & 1) * 512);
    };
    auto compute = [&](int g, const h8_t (&w)[GK][CFW]) {
        const int t = g / GPT, kg = g - t * GPT;
        const int ky = (t * 11) >> 5, kx = t - ky * 3;                  // t / 3 for t < 9
        const int off = (ky * a.RPS + (kx == 1 ? a.RPE : (kx >> 1))) * 16 + (kg * GK) * plane2;
#pragma unroll
        for (int j = 0; j < GK; ++j) {
            h8_t bf[PFW];
#pragma unroll
            for (int i = 0; i < PFW; ++i) bf[i] = *reinterpret_cast<const h8_t*>(base[i] + off + j * plane2);
#pragma unroll
            for (int cf = 0; cf < CFW; ++cf)
#pragma unroll
                for (int i = 0; i < PFW; ++i) acc[cf][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[j][cf], bf[i], acc[cf][i], 0, 0, 0);
        }
    };
    if (a.dbg & 2) {
    } else if constexpr (RESIDENT) {
        static_assert(CFW == 1, "resident weights: one cout fragment per wave");
#pragma unroll
        for (int u = 0; u < 9 * KSM; ++u) {
            const int t = u / KSM, ks = u - t * KSM;
            const int off = ((t / 3) * a.RPS + ((t % 3) == 1 ? a.RPE : ((t % 3) >> 1))) * 16 + ks * plane2;
#pragma unroll
            for (int i = 0; i < PFW; ++i) {
                const h8_t bf = *reinterpret_cast<const h8_t*>(base[i] + off);
                acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wres[u], bf, acc[0][i], 0, 0, 0);
            }
        }
    } else {
        wload(0, w0);
        wload(1, w1);
#pragma unroll 1
        for (int g = 0; g < NG; g += 3) {
            wload(g + 2, w2);
            compute(g, w0);
            if (g + 3 < NG) wload(g + 3, w0);
            compute(g + 1, w1);
            if (g + 4 < NG) wload(g + 4, w1);
            compute(g + 2, w2);
        }
    }
    // ---- epilogue, staged through LDS at BLOCK level.  A wave holds 32 of a pixel's couts: stored directly, every 128-byte
    // row of the output would be written in 16-byte pieces by 4-16 different store instructions of different waves (the
    // stem's direct stores ran at 550 GB/s).  The mid tile is dead once every wave has issued its last MFMA: its
    // LDS becomes the [tile pixel][Cout] fp16 image of the output tile, which leaves as whole rows, 16 bytes per lane.
    __syncthreads();
    unsigned zoff = 0;
    asm volatile("" : "+s"(zoff));     // (keeps the loop-invariant bias loads inside the tile loop: they would hold 16 registers across the producer phase)
    const float* bp = a.c.bias ? a.c.bias + zoff : nullptr;
    const int RS = a.c.Cout * 2 + 16;  // row pitch of the staged tile (bytes)
    char* stage = const_cast<char*>(mid);
    static_assert(CFW == 1, "one cout fragment per wave");
    float bz[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 t = bp ? *reinterpret_cast<const float4*>(bp + cf0 * 32 + 8 * g + 4 * kh) : make_float4(0.f, 0.f, 0.f, 0.f);   // Cout % 32 == 0 (host)
        bz[g * 4 + 0] = t.x;
        bz[g * 4 + 1] = t.y;
        bz[g * 4 + 2] = t.z;
        bz[g * 4 + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < PFW; ++i) {
        float v[16];
        finish16_any(a.c, acc[0][i], bz, cf0 * 32, kh, a.c.Cout, nullptr, 1.f, v);
        unsigned pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                h2_t t;
                t[0] = (_Float16)v[g * 4 + h * 2];
                t[1] = (_Float16)v[g * 4 + h * 2 + 1];
                pk[g][h] = __builtin_bit_cast(unsigned, t);
            }
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
            *reinterpret_cast<uint4*>(stage + (size_t)mm[i] * RS + (cf0 * 32 + 16 * gp + 8 * kh) * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    }
    __syncthreads();
    const int lpr = 31 - __builtin_clz(a.c.Cout >> 3);    // log2 of the 16-byte pieces per output row (Cout in {64, 128})
    const int npc = (a.TH * TW) << lpr;
    for (int qq = (int)threadIdx.x; qq < npc; qq += (int)blockDim.x) {
        const int row = qq >> lpr, pc = qq - (row << lpr);
        const int ry = row / TW, rx = row - ry * TW;
        const int oy = oy0 + ry, ox = ox0 + rx;
        if (oy < a.c.Ho && ox < a.c.Wo && !(a.dbg & 4))
            *reinterpret_cast<uint4*>(a.c.out + ((size_t)(b * a.c.Ho + oy) * a.c.Wo + ox) * a.c.out_cs + a.c.out_co + pc * 8) =
                *reinterpret_cast<const uint4*>(stage + (size_t)row * RS + pc * 16);
    }
}

// ---- 1x1 producer + consumer.  KSI: input k-steps (Cin / 16); CM: cout fragments of the 1x1 (Cm / 32), of which a wave computes
// PCM per pixel fragment (PCM = CM: every wave owns whole pixel fragments - each input piece is loaded once per block; PCM < CM:
// CM / PCM waves share a pixel fragment, each with its own cout fragments - what fits the registers at Cm = 128); the consumer has
// 2 CM k-steps and CFT cout fragments; NW waves; tile TH (runtime) x TW.
// The input pieces of a pixel fragment are requested TWO fragments ahead (a ring of three register sets): with one fragment
// ahead the producer ran at HBM latency per fragment - 38 of the 67 us of the 64-channel pair (tools/fused_bench.py probes).
template <int KSI, int CM, int CFT, int TW, int NW, int PCM>
__global__ __launch_bounds__(NW * 64, 2) void fused_pw_s2_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char mid[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5;
    const int Hm = a.c.H, Wm = a.c.W;
    static_assert(CM % PCM == 0 && NW % (CM / PCM) == 0, "waves per 1x1 cout-fragment group");
    constexpr int NGRP = CM / PCM;                     // wave groups along the 1x1's couts
    const int cfm0 = (wave % NGRP) * PCM;              // first cout fragment of this wave
    const int fstart = wave / NGRP;
    constexpr int FS = NW / NGRP;                      // pixel-fragment stride of a wave
    constexpr int nchunk1 = (KSI + 1) / 2;
    const int nmf = a.PLS >> 5;
    // consumer role: two pixel fragments x one cout fragment (host: NW = (TH * TW / 64) * CFT)
    const int npp = (a.TH * TW) >> 6;                  // pixel-fragment pairs of a tile
    const int cfrag_c = 2 * (wave % npp), cf_c = wave / npp;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int tx_i = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty_i = t2 % a.tiles_y;
        const int b = t2 / a.tiles_y;
        const int oy0 = ty_i * a.TH, ox0 = tx_i * TW;
        const int my0 = 2 * oy0 - 1, mx0 = 2 * ox0 - 1;
        // ---- produce: mid = act(conv1x1(in) + b1) for the tile's (2 TH + 1) x (2 TW + 1) pixels, zeros outside the image
        // (this wave's 1x1 weight fragments and bias are re-read per tile - L2 hits - so that they do not occupy registers
        //  during the consumer phase; the laundered offset keeps hipcc from hoisting the loads out of the tile loop)
        unsigned zoff = 0;
        asm volatile("" : "+s"(zoff));
        const __half* w1p = a.w1 + zoff;
        const float* b1p = a.b1 ? a.b1 + zoff : nullptr;
        h8_t w1[PCM][KSI];
        float bz1[PCM][16];
#pragma unroll
        for (int c = 0; c < PCM; ++c) {
#pragma unroll
            for (int ks = 0; ks < KSI; ++ks)
                w1[c][ks] = *reinterpret_cast<const h8_t*>(w1p + ((size_t)((cfm0 + c) * nchunk1 + (ks >> 1)) * 2 + (ks & 1)) * 512 + lane * 8);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = b1p ? *reinterpret_cast<const float4*>(b1p + (cfm0 + c) * 32 + 8 * g + 4 * kh) : make_float4(0.f, 0.f, 0.f, 0.f);
                bz1[c][g * 4 + 0] = t.x;
                bz1[c][g * 4 + 1] = t.y;
                bz1[c][g * 4 + 2] = t.z;
                bz1[c][g * 4 + 3] = t.w;
            }
        }
        auto issue = [&](int f, h8_t (&x)[KSI], bool& valid) {
            const int slot = f * 32 + (lane & 31);
            int my, mx;
            slot_to_mid(a, slot, my, mx);
            const int gy = my0 + my, gx = mx0 + mx;
            valid = slot < a.nslots && (unsigned)gy < (unsigned)Hm && (unsigned)gx < (unsigned)Wm;
            const size_t pix = valid ? ((size_t)b * Hm + gy) * Wm + gx : (size_t)b * Hm * Wm;   // invalid lanes read a real pixel and drop it
            const __half* p = a.in + pix * a.in_cs + a.in_co + kh * 8;
#pragma unroll
            for (int ks = 0; ks < KSI; ++ks) x[ks] = *reinterpret_cast<const h8_t*>(p + ks * 16);
        };
        auto work = [&](int f, const h8_t (&x)[KSI], bool valid) {
#pragma unroll
            for (int c = 0; c < PCM; ++c) {
                f32x16_t acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KSI; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[c][ks], x[ks], acc, 0, 0, 0);
                float v[16];
                produce_act16_any(a.act1, acc, bz1[c], valid, v);
                mid_store(mid, a.PLS, cfm0 + c, f * 32 + (lane & 31), lane, v);
            }
        };
        h8_t x0[KSI], x1[KSI], x2[KSI];
        bool v0 = false, v1 = false, v2 = false;
        int f = (a.dbg & 1) ? nmf : fstart;
        if (f < nmf) issue(f, x0, v0);
        if (f + FS < nmf) issue(f + FS, x1, v1);
        for (; f < nmf; f += 3 * FS) {
            if (f + 2 * FS < nmf) issue(f + 2 * FS, x2, v2);
            work(f, x0, v0);
            if (f + FS < nmf) {
                if (f + 3 * FS < nmf) issue(f + 3 * FS, x0, v0);
                work(f + FS, x1, v1);
            }
            if (f + 2 * FS < nmf) {
                if (f + 4 * FS < nmf) issue(f + 4 * FS, x1, v1);
                work(f + 2 * FS, x2, v2);
            }
        }
        __syncthreads();
        // ---- consume
        consume<2 * CM, 2, 1, TW>(a, mid, cfrag_c, cf_c, lane, b, oy0, ox0);
        __syncthreads();   // the next tile's producer overwrites the planes
    }
}

// ---- image-conv producer (Cm = 32) + consumer.  Tile TH = 4 x TW = 32 output pixels, 4 waves (wave = output row).
constexpr int FS_TH = 4, FS_TW = 32;
constexpr int FS_MH = 2 * FS_TH + 1, FS_MW = 2 * FS_TW + 1;    // 9 x 65 mid pixels
constexpr int FS_WR = 2 * FS_MH + 1;                           // 19 image rows
constexpr int FS_WPC = 17;                                     // 16-byte pieces per window row: image columns [4 ox0 - 8, 4 ox0 + 128)
constexpr int FS_PITCH = FS_WPC * 8;                           // halves
constexpr int FS_NPIECE = 3 * FS_WR * FS_WPC;                  // 969
constexpr int FS_PLS = ((FS_MH * FS_MW + 31) / 32) * 32;       // 608 slots per plane
constexpr int FS_MID_BYTES = 4 * FS_PLS * 16;                  // 38 912
constexpr int FS_WIN_BYTES = 3 * FS_WR * FS_PITCH * 2;         // 15 504

constexpr int FS_WLDS_BYTES = 2 * 64 * 16 + 32 * 4;           // the image conv's two A fragments per lane + 32 biases

template <typename TI>
__global__ __launch_bounds__(256, 2) void fused_stem_s2_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* mid = smem;
    _Float16* s_in = reinterpret_cast<_Float16*>(smem + FS_MID_BYTES);
    char* s_w = smem + FS_MID_BYTES + FS_WIN_BYTES;            // [ks][lane] x 16 B, then 32 floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5;
    const int Hm = a.c.H, Wm = a.c.W;
    constexpr int K = 27;
    const TI* img = reinterpret_cast<const TI*>(a.img);
    const size_t IHW = (size_t)a.IH * a.IW;
    // image-conv weights as A fragments (cout = lane & 31, k = ks * 16 + kh * 8 + j, k = ci * 9 + ky * 3 + kx; k >= 27: zero) and
    // the bias, ONCE per block into LDS: each tile reads them back with two + four ds_read_b128 (re-gathering 16 fp32 words per
    // lane from global memory per tile cost six dependent round trips per tile - 207 us for the pair on the first visit)
    if (tid < 128) {
        const int l = tid & 63, ks = tid >> 6;
        h8_t f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + (l >> 5) * 8 + j;
            f[j] = k < K ? (_Float16)a.wst[(size_t)(l & 31) * K + k] : (_Float16)0.f;
        }
        *reinterpret_cast<h8_t*>(s_w + (ks * 64 + l) * 16) = f;
    } else if (tid < 160) {
        reinterpret_cast<float*>(s_w + 2 * 64 * 16)[tid - 128] = a.b1 ? a.b1[tid - 128] : 0.f;
    }
    // this lane's 16 window offsets (halves): element k of the patch of mid pixel (my, mx) sits at
    // s_in[(ci * WR + 2 my + ky) * PITCH + 2 mx + 5 + kx]
    int woff[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + kh * 8 + j;
            const int kk = k < K ? k : 0;
            woff[ks][j] = ((kk / 9) * FS_WR + (kk % 9) / 3) * FS_PITCH + (kk % 3) + 5;
        }
    // this thread's window pieces (channel, window row, piece column): tile independent
    int pc_off[4], pc_yy[4], pc_px[4], pc_ci[4];
    bool pc_on[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + i * 256;
        pc_on[i] = p < FS_NPIECE;
        const int pp = pc_on[i] ? p : 0;
        pc_ci[i] = pp / (FS_WR * FS_WPC);
        const int r = pp - pc_ci[i] * (FS_WR * FS_WPC);
        pc_yy[i] = r / FS_WPC;
        pc_px[i] = r - pc_yy[i] * FS_WPC;
        pc_off[i] = (pc_ci[i] * FS_WR + pc_yy[i]) * FS_PITCH + pc_px[i] * 8;
    }
    constexpr int NMF = FS_PLS / 32;   // 19 mid fragments
    StemPiece<TI> pre[4];
    auto request = [&](int tile) {     // image window of `tile` -> registers (aligned 16-byte pieces; IW % 8 == 0: a piece is wholly in or out)
        const int tx_i = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty_i = t2 % a.tiles_y;
        const int b = t2 / a.tiles_y;
        const int iy0 = 4 * ty_i * FS_TH - 3, cx0 = 4 * tx_i * FS_TW - 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = iy0 + pc_yy[i], ix = cx0 + pc_px[i] * 8;
            if (pc_on[i] && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
                pre[i].load(img + ((size_t)b * 3 + pc_ci[i]) * IHW + (size_t)iy * a.IW + ix);
            else
                pre[i].zero();
        }
    };
    // the stride-2 conv's weight fragments of this wave's cout fragment (wave >> 1): 9 taps x 2 k-steps, resident
    h8_t wres[18];
    {
        const __half* wb = a.c.wpk + (size_t)(wave >> 1) * 9 * 1024 + lane * 8;       // [cfr][chunk = 0][tap][ks][lane][8]
#pragma unroll
        for (int u = 0; u < 18; ++u) wres[u] = *reinterpret_cast<const h8_t*>(wb + (size_t)(u >> 1) * 1024 + (u & 1) * 512);
    }
    int tile = blockIdx.x;
    if (tile < a.ntiles && !(a.dbg & 8)) request(tile);
    for (; tile < a.ntiles; tile += gridDim.x) {
        const int tx_i = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty_i = t2 % a.tiles_y;
        const int b = t2 / a.tiles_y;
        const int oy0 = ty_i * FS_TH, ox0 = tx_i * FS_TW;
        const int my0 = 2 * oy0 - 1, mx0 = 2 * ox0 - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (pc_on[i]) *reinterpret_cast<uint4*>(s_in + pc_off[i]) = pre[i].as_half8();
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles && !(a.dbg & 8)) request(tile + gridDim.x);   // in flight during the MFMAs and the stores below
        // ---- produce: mid = act(conv3x3s2(image) + b1), 32 channels = 4 planes
        h8_t af[2];
        af[0] = *reinterpret_cast<const h8_t*>(s_w + lane * 16);
        af[1] = *reinterpret_cast<const h8_t*>(s_w + (64 + lane) * 16);
        float bz1[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(s_w + 2 * 64 * 16 + (8 * g + 4 * kh) * 4);
            bz1[g * 4 + 0] = t.x;
            bz1[g * 4 + 1] = t.y;
            bz1[g * 4 + 2] = t.z;
            bz1[g * 4 + 3] = t.w;
        }
        for (int f = wave; f < NMF && !(a.dbg & 1); f += 4) {
            const int slot = f * 32 + (lane & 31);
            int my = slot / FS_MW;
            const int ms = slot - my * FS_MW;
            int mx = ms < FS_TW + 1 ? 2 * ms : 2 * (ms - (FS_TW + 1)) + 1;
            const bool inr = slot < FS_MH * FS_MW;
            if (!inr) {
                my = 0;
                mx = 0;
            }
            const int gy = my0 + my, gx = mx0 + mx;
            const bool valid = inr && (unsigned)gy < (unsigned)Hm && (unsigned)gx < (unsigned)Wm;
            const _Float16* base = s_in + (2 * my) * FS_PITCH + 2 * mx;
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8_t bf;
#pragma unroll
                for (int j = 0; j < 8; ++j) bf[j] = base[woff[ks][j]];      // (k >= 27 multiplies a zero weight)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], bf, acc, 0, 0, 0);
            }
            float v[16];
            produce_act16_any(a.act1, acc, bz1, valid, v);
            mid_store(mid, FS_PLS, 0, slot, lane, v);
        }
        __syncthreads();
        // ---- consume: output rows 2 (wave & 1), 2 (wave & 1) + 1 of the tile x cout fragment (wave >> 1)
        consume<2, 2, 1, FS_TW, true>(a, mid, 2 * (wave & 1), wave >> 1, lane, b, oy0, ox0, wres);
        __syncthreads();
    }
}

int n_cu_cached() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
    }
    return n_cu;
}

// consumer part of the argument block
int fill_consumer(const y6_conv_desc* s2, int B, int Hm, int Wm, FusedArgs* a) {
    ConvKArgs& k = a->c;
    memset(&k, 0, sizeof(k));
    static const int probe = getenv("Y6_FUSED_PROBE") ? atoi(getenv("Y6_FUSED_PROBE")) : 0;
    a->dbg = probe;
    k.out = (__half*)s2->out.data;
    k.wpk = (const __half*)s2->w_packed;
    k.bias = s2->bias;
    k.Cin = s2->in.C;
    k.Cout = s2->out.C;
    k.out_cs = s2->out.cstride;
    k.out_co = s2->out.coff;
    k.act = s2->act;
    k.vec_ok = (s2->out.cstride % 4 == 0) && (s2->out.coff % 4 == 0) && (((uintptr_t)s2->out.data & 7) == 0);
    k.vec16_ok = (s2->out.cstride % 8 == 0) && (s2->out.coff % 8 == 0) && (((uintptr_t)s2->out.data & 15) == 0);
    k.B = B;
    k.H = Hm;
    k.W = Wm;
    k.Ho = s2->out.H;
    k.Wo = s2->out.W;
    k.upC = k.Cout;
    a->B = B;
    return Y6_OK;
}

bool consumer_ok(const y6_conv_desc* s2, int B, int Hm, int Wm, int Cm) {
    return s2->ksize == 3 && s2->stride == 2 && s2->in.C == Cm && s2->in.B == B && s2->in.H == Hm && s2->in.W == Wm &&
           s2->out.B == B && s2->out.H == (Hm + 2 - 3) / 2 + 1 && s2->out.W == (Wm + 2 - 3) / 2 + 1 && s2->out.data && s2->w_packed &&
           !s2->post_scale && !s2->res.data && s2->out.C % 32 == 0 && s2->out.cstride % 8 == 0 && s2->out.coff % 8 == 0 &&
           (((uintptr_t)s2->out.data & 15) == 0) && (size_t)B * s2->out.H * s2->out.W < 0x7fffffffull;
}

struct KernState {   // per kernel instantiation: LDS opt-in done, resident blocks per CU for the footprint it was asked for
    bool big = false;
    size_t lds = 0;
    int bpc = 0;
};

template <typename KERN>
int launch_fused(KERN kern, const FusedArgs& a, int threads, size_t lds, hipStream_t s, KernState* st) {
    if (lds > 64 * 1024 && !st->big) {
        Y6_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        st->big = true;
    }
    Y6_REQUIRE(lds <= 160 * 1024, "conv_fused: tile needs %zu bytes of LDS", lds);
    if (st->lds != lds) {
        int bpc = 0;
        Y6_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, threads, lds));
        st->bpc = bpc < 1 ? 1 : bpc;
        st->lds = lds;
    }
    int grid = n_cu_cached() * st->bpc;
    if (grid > a.ntiles) grid = a.ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

struct PwCfg {
    int ksi, cm, cft, th;
};

int pw_s2_cfg(const y6_pw_s2_desc* d, PwCfg* cfg) {
    const y6_conv_desc &pw = d->pw, &s2 = d->s2;
    if (pw.ksize != 1 || pw.stride != 1 || pw.post_scale || pw.res.data || !pw.in.data || !pw.w_packed) return 0;
    if (pw.in.cstride % 8 || pw.in.coff % 8 || ((uintptr_t)pw.in.data & 15)) return 0;
    const int Cin = pw.in.C, Cm = pw.out.C;
    if (!consumer_ok(&s2, pw.in.B, pw.in.H, pw.in.W, Cm)) return 0;
    if (Cin == 64 && Cm == 64 && s2.out.C == 64) {
        *cfg = PwCfg{4, 2, 2, 8};
        return 1;
    }
    if (Cin == 128 && Cm == 128 && s2.out.C == 128) {
        *cfg = PwCfg{8, 4, 4, 4};
        return 1;
    }
    return 0;
}

int pw_s2_launch(const y6_pw_s2_desc* d, hipStream_t s) {
    PwCfg cfg;
    Y6_REQUIRE(pw_s2_cfg(d, &cfg), "fused_pw_s2: unsupported pair (1x1 64->64 / 128->128 into a 3x3 stride-2 conv of the same width)");
    constexpr int TW = 16;
    FusedArgs a;
    memset(&a, 0, sizeof(a));
    const y6_conv_desc &pw = d->pw, &s2 = d->s2;
    fill_consumer(&s2, pw.in.B, pw.in.H, pw.in.W, &a);
    a.in = (const __half*)pw.in.data;
    a.in_cs = pw.in.cstride;
    a.in_co = pw.in.coff;
    a.Cin = pw.in.C;
    a.w1 = (const __half*)pw.w_packed;
    a.b1 = pw.bias;
    a.act1 = pw.act;
    a.TH = cfg.th;
    a.tiles_x = y6_cdiv(a.c.Wo, TW);
    a.tiles_y = y6_cdiv(a.c.Ho, a.TH);
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    a.MH = 2 * a.TH + 1;
    a.RPS = 2 * TW + 1;
    a.RPE = TW + 1;
    a.nslots = a.MH * a.RPS;
    a.PLS = y6_cdiv(a.nslots, 32) * 32;
    const size_t lds = (size_t)(pw.out.C / 8) * a.PLS * 16;
    static KernState st64, st128;
    if (cfg.ksi == 4) return launch_fused(fused_pw_s2_kernel<4, 2, 2, 16, 4, 2>, a, 256, lds, s, &st64);
    return launch_fused(fused_pw_s2_kernel<8, 4, 4, 16, 4, 1>, a, 256, lds, s, &st128);
}

int stem_s2_ok(const y6_stem_s2_desc* d) {
    const y6_stem_desc& st = d->stem;
    if (!st.in_nchw || !st.w_oihw_f32 || st.Cin != 3 || st.post_scale || st.W % 8 != 0 || st.out.C != 32) return 0;
    const size_t esz = st.in_dtype == Y6_F16 ? 2 : (st.in_dtype == Y6_U8 ? 1 : (st.in_dtype == Y6_F32 ? 4 : 0));
    if (!esz || ((uintptr_t)st.in_nchw % (8 * esz)) != 0 || ((size_t)st.H * st.W * esz) % (8 * esz) != 0) return 0;
    const int Hm = (st.H + 2 - 3) / 2 + 1, Wm = (st.W + 2 - 3) / 2 + 1;
    if (!consumer_ok(&d->s2, st.B, Hm, Wm, 32)) return 0;
    return d->s2.out.C == 64;
}

int stem_s2_launch(const y6_stem_s2_desc* d, hipStream_t s) {
    Y6_REQUIRE(stem_s2_ok(d), "fused_stem_s2: unsupported pair (3-channel image conv to 32 channels into a 3x3 stride-2 conv to 64)");
    FusedArgs a;
    memset(&a, 0, sizeof(a));
    const y6_stem_desc& st = d->stem;
    const int Hm = (st.H + 2 - 3) / 2 + 1, Wm = (st.W + 2 - 3) / 2 + 1;
    fill_consumer(&d->s2, st.B, Hm, Wm, &a);
    a.img = st.in_nchw;
    a.IH = st.H;
    a.IW = st.W;
    a.wst = st.w_oihw_f32;
    a.b1 = st.bias;
    a.act1 = st.act;
    a.TH = FS_TH;
    a.tiles_x = y6_cdiv(a.c.Wo, FS_TW);
    a.tiles_y = y6_cdiv(a.c.Ho, FS_TH);
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    a.MH = FS_MH;
    a.RPS = FS_MW;
    a.RPE = FS_TW + 1;
    a.nslots = FS_MH * FS_MW;
    a.PLS = FS_PLS;
    const size_t lds = FS_MID_BYTES + FS_WIN_BYTES + FS_WLDS_BYTES;
    static KernState kst[3];
    if (st.in_dtype == Y6_F16) return launch_fused(fused_stem_s2_kernel<__half>, a, 256, lds, s, &kst[0]);
    if (st.in_dtype == Y6_U8) return launch_fused(fused_stem_s2_kernel<uint8_t>, a, 256, lds, s, &kst[1]);
    return launch_fused(fused_stem_s2_kernel<float>, a, 256, lds, s, &kst[2]);
}

double pair_flops(double px_mid, double k1, int Cm, const y6_conv_desc& s2) {
    return 2.0 * px_mid * k1 * Cm + 2.0 * s2.out.B * s2.out.H * s2.out.W * (double)s2.out.C * Cm * 9.0;
}

}  // namespace

extern "C" int y6_fused_pw_s2_supported(const y6_pw_s2_desc* d) {
    PwCfg cfg;
    return d ? pw_s2_cfg(d, &cfg) : 0;
}
extern "C" int y6_fused_pw_s2(const y6_pw_s2_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d, "fused_pw_s2: null descriptor");
    return pw_s2_launch(d, (hipStream_t)stream);
}
extern "C" int y6_plan_add_pw_s2(y6_plan* p, const y6_pw_s2_desc* d) {
    Y6_REQUIRE(p && d, "plan_add_pw_s2: null argument");
    Y6_REQUIRE(y6_fused_pw_s2_supported(d), "plan_add_pw_s2: unsupported pair");
    const y6_conv_desc &pw = d->pw, &s2 = d->s2;
    const double px = (double)pw.in.B * pw.in.H * pw.in.W;
    const double by = 2.0 * (px * pw.in.C + (double)s2.out.B * s2.out.H * s2.out.W * s2.out.C + (double)pw.in.C * pw.out.C + 9.0 * pw.out.C * s2.out.C);
    return y6_plan_push(p, pw_s2_launch, d, Y6_TOP_PW_S2, pair_flops(px, pw.in.C, pw.out.C, s2), by);
}

extern "C" int y6_fused_stem_s2_supported(const y6_stem_s2_desc* d) { return d ? stem_s2_ok(d) : 0; }
extern "C" int y6_fused_stem_s2(const y6_stem_s2_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d, "fused_stem_s2: null descriptor");
    return stem_s2_launch(d, (hipStream_t)stream);
}
extern "C" int y6_plan_add_stem_s2(y6_plan* p, const y6_stem_s2_desc* d) {
    Y6_REQUIRE(p && d, "plan_add_stem_s2: null argument");
    Y6_REQUIRE(y6_fused_stem_s2_supported(d), "plan_add_stem_s2: unsupported pair");
    const y6_stem_desc& st = d->stem;
    const y6_conv_desc& s2 = d->s2;
    const double pxm = (double)st.B * s2.in.H * s2.in.W;
    const double esz = st.in_dtype == Y6_F16 ? 2.0 : (st.in_dtype == Y6_U8 ? 1.0 : 4.0);
    const double by = esz * st.B * 3.0 * st.H * st.W + 2.0 * s2.out.B * s2.out.H * s2.out.W * (double)s2.out.C + 2.0 * 9.0 * 32 * s2.out.C;
    int rc = y6_plan_push(p, stem_s2_launch, d, Y6_TOP_STEM_S2, pair_flops(pxm, 27.0, 32, s2), by);
    if (rc) return rc;
    // the caller's image is a boundary input: y6_plan_rebind_input / y6_plan_rebind re-point it
    return y6_plan_mark_input(p, offsetof(y6_stem_s2_desc, stem) + offsetof(y6_stem_desc, in_nchw));
}
