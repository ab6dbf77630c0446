// conv_pw.hip - 1x1 convolution (stride 1) as a latency-lean GEMM: the WHOLE reduction of a block's pixel tile is requested at
// once (LDS-DMA in full 128-byte lines), the weights stream global -> VGPR, one block barrier in front of the matrix loop.
//
// Replaces the same aten composition as conv_mfma.hip's per-tap kernel for ksize 1 (reference yolov6/layers/common.py:51-54
// ConvModule.forward_fuse behind ConvBNReLU / ConvBNSiLU: the CSP-SPPF convs common.py:140-176, the neck's reduce / BiFusion convs
// reppan.py:215-237, the head stems effidehead.py:166-170) - 14 launches of a YOLOv6-S step.
//
// Why another kernel (round 6).  The per-tap kernel walks the reduction in 32-channel chunks: a chunk is four MFMAs per wave between
// two block barriers, its pixel piece prefetched ONE chunk ahead through registers - every chunk pays a full global-memory round
// trip, sixteen in a row for Cin = 512 (profiles/r06/ops_driver_r06ev.json: 512 -> 256 @20x20 b32, 3.4 GFLOP / 20 MB, 21.7 us; the
// bytes are 4 us of HBM time and the FLOPs 3 us of the matrix pipe).  The 1x1 launches are latency, not throughput.  Here
//   * a block owns TP = 64 / 128 pixels x TC = 128 / 64 couts and asks for its whole [TP][Cin] pixel image in the prologue - Cin / 64
//     stages of [pixel][128 B], each request 8 pixel rows x 128 B = eight whole cache lines (the per-tap kernel's pieces and the
//     streaming kernel's fragment-shaped loads touch 32-byte quarters of 32 lines) - so the memory round trip is paid ONCE;
//   * LDS image of a stage: [pixel][eight 16-byte slots], slot = piece ^ ((pixel >> 1) & 7), the swizzle applied on the SOURCE side
//     of the request (a lane may ask for any 16 bytes): the sixteen lanes ds_read_b128 serves per cycle (16 consecutive pixels, one
//     k-half) land on sixteen different bank groups; no pad bytes, so a request writes 1 KiB of contiguous LDS;
//   * a wave owns one 32-cout fragment x PF pixel fragments; its weight fragments (1 KiB per k-step, contiguous in the packed
//     weights) go global -> VGPR through a ring of up to sixteen loads (a k-step is only PF MFMAs: the ring has to cover the L2
//     latency by depth) and never touch LDS;
//   * ONE wait + barrier (everything this wave asked for has landed), then Cin / 16 k-steps without any block-wide
//     synchronisation; the reduction is fully unrolled (Cin / 64 is a template parameter: 1 ... 6, 8, 10, 12, 16);
//   * epilogue: bias + activation in conv_common.hpp's arithmetic, the fp16 tile staged through the (dead) pixel
//     image and written as whole NHWC rows - 16 lanes x 16 B per pixel row of 128 couts; the residual add (BottleRep, the
//     accumulating data-gradient convs of the training step) and the ConvTranspose2d(k2, s2) scatter happen on those rows.
// Vector-memory ordering follows conv_wreg.hip's rule: requests and VGPR loads are awaited TOGETHER (vmcnt(0)) where both are in
// flight; counted waits appear only where nothing but weight loads is outstanding.
#include "common.hpp"
#include "conv_common.hpp"
#include <type_traits>

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void pw_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pw_static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ void pw_load_frag(i32x4_t& dst, const i32x4_t& rsrc, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
template <int N>
__device__ __forceinline__ void pw_wait_frag(i32x4_t& frag) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(frag) : "n"(N) : "memory");
}
template <int IMM>
__device__ __forceinline__ void pw_lds_read16(i32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void pw_wait_lds(i32x4_t& frag) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N) : "memory");
}

// NKS: k-steps of the reduction - a k-step is 32 BYTES per pixel (16 fp16 / 32 int8 channels), a stage four of them (128 bytes per
// pixel); the last stage may be partial (int8: Cin = 64, 192).  WC x WP = 4 waves: WC along the couts (32 each), WP along the pixels
// (PF fragments of 32 each).  ConvKArgs as conv_mfma.hip's build_launch fills it for a 1x1 conv: W = flattened pixel count, TW = TP,
// ntiles, ncb, nids.
// I8: the int8 form (include/yolov6_hip.h y6_conv_i8_desc; BASELINE configs[4]) - the producer's int8 twin as input, v_mfma_i32_32x32x32_i8
// (the same 16 bytes per lane and operand: requests, LDS image, weight stream and waits are shared word for word), exact int32 sums
// times s_x * s_w[c] as a rounding of its own, then the fp16 epilogue; optionally the int8 twin of the output rows.
template <int NKS, int WC, int WP, int PF, bool I8>
__global__ __launch_bounds__(256, 2) void conv_pw_kernel(const ConvKArgs a) {
    constexpr int TP = WP * PF * 32, TC = WC * 32;
    constexpr int NK = (NKS + 3) / 4;                 // stages
    constexpr int ES = I8 ? 1 : 2;                    // bytes per input element
    typedef typename std::conditional<I8, i32x16_t, f32x16_t>::type acc_t;
    constexpr int R = NKS < 16 ? NKS : 16;            // weight fragments in flight per wave
    constexpr int LA = 2;                             // k-steps of pixel fragments requested ahead of the MFMAs
    constexpr int NREQ = NK * (TP / 8);               // 1 KiB requests of the block's pixel image
    constexpr int NREQW = NREQ / 4;                   // per wave
    constexpr int OP = TC * 2 + 16;                   // row pitch of the staged output tile
    static_assert(WC * WP == 4 && NREQ % 4 == 0 && (TP / 8) % 4 == 0, "four waves share the requests; a wave's pixel groups have one parity");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WC, wp = wave / WC;

    // ---- block -> (pixel tile, cout block); the cout blocks of one tile share an XCD (id % 8)
    int tile, cb;
    {
        const int id = blockIdx.x;
        if (a.ncb == 1) {
            tile = id;
            cb = 0;
        } else {
            const int lo = id & 7, r = id >> 3;
            cb = r % a.ncb;
            tile = (r / a.ncb) * 8 + lo;
        }
    }
    if (tile >= a.ntiles) return;
    const int npix = a.W;
    const int pix0 = tile * TP;
    const unsigned smem_base = lds_addr(smem);
    const int ics = I8 ? a.qin_cs : a.in_cs, ico = I8 ? a.qin_co : a.in_co;
    const unsigned rowB = (unsigned)ics * (unsigned)ES;

    const i32x4_t rsA = make_rsrc(I8 ? (const void*)a.qin : (const void*)a.in, (unsigned)((size_t)npix * ics * ES));
    const i32x4_t rsW = make_rsrc(a.wpk, 0xfffffe00u);

    // ---- the pixel image.  Request q = wave + 4 i covers stage s = q / (TP / 8), pixel group g = q % (TP / 8): lane l asks for
    //      piece (l & 7) ^ swz of pixel 8 g + (l >> 3), swz = ((8 g + (l >> 3)) >> 1) & 7 = ((l >> 4) + 4 g) & 7; g has the wave's parity.
    {
        const unsigned pl = (unsigned)(lane >> 3);
        const unsigned piece = ((unsigned)(lane & 7) ^ (((unsigned)(lane >> 4) + 4u * (unsigned)(wave & 1)) & 7u));
        const unsigned vlane = ((unsigned)pix0 + pl) * rowB + piece * 16u;   // (the tile's origin in the VECTOR offset: kOob + scalar offset must not wrap)
        const unsigned sbase = (unsigned)ico * (unsigned)ES;
#pragma unroll
        for (int i = 0; i < NREQW; ++i) {
            const int q = wave + 4 * i;
            const int s = q / (TP / 8), g = q % (TP / 8);
            const int vp = (NKS - 4 * s) * 2 < 8 ? (NKS - 4 * s) * 2 : 8;   // pieces of this stage that exist (a partial last stage: zeros behind them, never read)
            const bool v = pix0 + 8 * g + (int)pl < npix && (int)piece < vp;
            const unsigned voff = v ? vlane + (unsigned)(8 * g) * rowB : kOob;
            dma16(rsA, voff, sbase + (unsigned)s * 128u, smem_base + (unsigned)(s * TP + 8 * g) * 128u);
        }
    }
    // ---- the first R weight fragments of this wave's cout fragment
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned wbase = (unsigned)((cb * WC + wc) * NKS) * 1024u;
    i32x4_t wr[R];
#pragma unroll
    for (int u = 0; u < R; ++u) pw_load_frag(wr[u], rsW, lane16, wbase + (unsigned)(u * 1024));

    // the bias of this lane's sixteen couts (8 g + 4 (lane >> 5) .. + 3 of the wave's fragment), behind the same wait; a null bias is a
    // descriptor of zero records: the loads return zeros
    i32x4_t bzr[4];
    {
        // ConvTranspose2d(k2, s2) as one [4 Cout x Cin] GEMM (a.up == 2): the block's couts belong to ONE (dy, dx) sub-kernel
        // (host: upC % TC == 0) and the bias is indexed by the real channel
        const int cend = a.up == 2 ? a.upC : a.Cout;
        const int upc0 = a.up == 2 ? ((cb * TC) / a.upC) * a.upC : 0;
        const i32x4_t rsB = make_rsrc(a.bias, a.bias != nullptr ? (unsigned)cend * 4u : 0u);
        const unsigned bo = (unsigned)((cb * WC + wc) * 32 - upc0 + 4 * (lane >> 5)) * 4u;
#pragma unroll
        for (int g = 0; g < 4; ++g) pw_load_frag(bzr[g], rsB, bo + (unsigned)(g * 32), 0u);
    }
    i32x4_t qzr[I8 ? 4 : 1];   // int8: s_x * s_w[c] of the same couts
    if constexpr (I8) {
        const i32x4_t rsQ = make_rsrc(a.qscale, (unsigned)a.Cout * 4u);
        const unsigned qo = (unsigned)((cb * WC + wc) * 32 + 4 * (lane >> 5)) * 4u;
#pragma unroll
        for (int g = 0; g < 4; ++g) pw_load_frag(qzr[g], rsQ, qo + (unsigned)(g * 32), 0u);
    }

    // this lane's pixels: LDS address of k-step (kk & 3) of stage 0
    const int fq = frag_pixel(lane & 31);
    unsigned baddr[PF][4];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const unsigned p = (unsigned)(wp * (PF * 32) + pf * 32 + fq);
        const unsigned swz = (p >> 1) & 7u;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) baddr[pf][k4] = smem_base + p * 128u + ((((unsigned)(k4 * 2) + (unsigned)(lane >> 5)) ^ swz) * 16u);
    }

    acc_t acc[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pf][r] = 0;

    // everything this wave asked for (its share of the image, its first R weight fragments, the bias) has landed; then everybody's
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- the reduction: NKS k-steps x PF fragment products, pixel fragments LA k-steps ahead, weight fragments R k-steps ahead
    constexpr int NB = LA + 1;
    i32x4_t fb[NB][PF];
    auto frag_read = [&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        constexpr int s = kk >> 2, k4 = kk & 3;
        constexpr int off = s * TP * 128;
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) pw_lds_read16<(off & 0xffff)>(fb[kk % NB][pf], baddr[pf][k4] + (unsigned)(off & ~0xffff));
    };
    pw_static_for<0, (LA < NKS ? LA : NKS)>(frag_read);
    pw_static_for<0, NKS>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        if constexpr (kk + LA < NKS) frag_read(std::integral_constant<int, kk + LA>{});
        // weight fragment kk: the younger weight loads in flight are those of k-steps kk + 1 .. min(kk + R, NKS) - 1
        if constexpr (kk < R) {
            asm volatile("" : "+v"(wr[kk % R]));   // landed in front of the barrier
        } else {
            pw_wait_frag<(NKS - 1 - kk < R - 1 ? NKS - 1 - kk : R - 1)>(wr[kk % R]);
        }
        constexpr int ahead = (NKS - 1 - kk < LA ? NKS - 1 - kk : LA);   // k-steps of pixel fragments requested behind this one
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) {
            switch (PF - 1 - pf + PF * ahead) {   // (compile-time after unrolling)
                case 0: pw_wait_lds<0>(fb[kk % NB][pf]); break;
                case 1: pw_wait_lds<1>(fb[kk % NB][pf]); break;
                case 2: pw_wait_lds<2>(fb[kk % NB][pf]); break;
                case 3: pw_wait_lds<3>(fb[kk % NB][pf]); break;
                case 4: pw_wait_lds<4>(fb[kk % NB][pf]); break;
                case 5: pw_wait_lds<5>(fb[kk % NB][pf]); break;
                case 6: pw_wait_lds<6>(fb[kk % NB][pf]); break;
                case 7: pw_wait_lds<7>(fb[kk % NB][pf]); break;
                case 8: pw_wait_lds<8>(fb[kk % NB][pf]); break;
                default: pw_wait_lds<9>(fb[kk % NB][pf]); break;
            }
            if constexpr (I8)
                acc[pf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wr[kk % R], fb[kk % NB][pf], acc[pf], 0, 0, 0);
            else
                acc[pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, wr[kk % R]), __builtin_bit_cast(h8_t, fb[kk % NB][pf]), acc[pf], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kk + R < NKS) {   // the slot this k-step released
            pw_load_frag(wr[kk % R], rsW, lane16, wbase + (unsigned)((kk + R) * 1024));
            __builtin_amdgcn_sched_barrier(0);
        }
    });

    // ---- epilogue: the block's [TP][TC] fp16 tile through LDS (the pixel image is dead once everybody has left the loop)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
        const int kh = lane >> 5;
        float bias16[16];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bits = bzr[g][j];   // (a copy: __builtin_bit_cast of the vector-element lvalue itself reads element 0 whatever j is)
                bias16[g * 4 + j] = __builtin_bit_cast(float, bits);
            }
        float qs16[16];
        if constexpr (I8) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int bits = qzr[g][j];
                    qs16[g * 4 + j] = __builtin_bit_cast(float, bits);
                }
        }
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) {
            float x[16], v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (I8) {   // exact int32 -> fp32, * s_x * s_w[c] as a rounding of its own (no fma with the bias add): conv_i8_epilogue
                    float t = (float)acc[pf][r] * qs16[r];
                    asm volatile("" : "+v"(t));
                    x[r] = t + bias16[r];
                } else {
                    x[r] = acc[pf][r] + bias16[r];
                }
            }
            switch (a.act) {   // one wave-uniform branch per fragment; conv_common.hpp's arithmetic (finish16 without post-affine / residual)
                case Y6_ACT_RELU:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = act_const<Y6_ACT_RELU>(x[r]);
                    break;
                case Y6_ACT_SILU:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = act_const<Y6_ACT_SILU>(x[r]);
                    break;
                case Y6_ACT_HARDSWISH:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = act_const<Y6_ACT_HARDSWISH>(x[r]);
                    break;
                default:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = x[r];
                    break;
            }
            const int row = wp * (PF * 32) + pf * 32 + fq;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h4_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)v[g * 4 + j];
                *reinterpret_cast<h4_t*>(smem + row * OP + wc * 64 + g * 16 + kh * 8) = o;
            }
        }
    }
    __syncthreads();
    {
        constexpr int PPR = TC / 8;                  // 16-byte pieces per row
        constexpr int NPC = TP * PPR / 256;          // pieces per thread
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        if constexpr (I8) {   // (host: no residual, no scatter) the fp16 rows and / or their int8 twin for quantised consumers
            const bool has_out = a.out != nullptr, has_q = a.qout != nullptr;
            const __amdgpu_buffer_rsrc_t rsO =
                __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, has_out ? (int)(unsigned)((size_t)npix * a.out_cs * 2) : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsQo =
                __builtin_amdgcn_make_buffer_rsrc((void*)a.qout, 0, has_q ? (int)(unsigned)((size_t)npix * a.qout_cs) : 0, 0x00020000);
            const unsigned obase = ((unsigned)pix0 * (unsigned)a.out_cs + (unsigned)a.out_co + (unsigned)(cb * TC)) * 2u;
            const unsigned qbase = (unsigned)pix0 * (unsigned)a.qout_cs + (unsigned)a.qout_co + (unsigned)(cb * TC);
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int q = tid + 256 * i;
                const int row = q / PPR, pc = q % PPR;
                const u32x4_t o = *reinterpret_cast<const u32x4_t*>(smem + row * OP + pc * 16);
                const bool inr = pix0 + row < npix;
                if (has_out) __builtin_amdgcn_raw_buffer_store_b128(o, rsO, (int)(inr ? obase + (unsigned)row * (unsigned)a.out_cs * 2u + (unsigned)pc * 16u : kOob), 0, 0);
                if (has_q) {   // the SAME fp16 values, quantised with the consumers' scale (conv_common.hpp q8_quad)
                    const unsigned qb = inr ? qbase + (unsigned)row * (unsigned)a.qout_cs + (unsigned)pc * 8u : kOob;
                    __builtin_amdgcn_raw_buffer_store_b32(q8_quad(o[0], o[1], a.qo_inv2, a.qo_lo2, a.qo_hi2), rsQo, (int)qb, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(q8_quad(o[2], o[3], a.qo_inv2, a.qo_lo2, a.qo_hi2), rsQo, (int)(inr ? qb + 4u : kOob), 0, 0);
                }
            }
            return;
        }
        if (a.up == 2) {   // scatter: input pixel (b, y, x) -> output pixel (b, 2 y + dy, 2 x + dx), channels of the block's sub-kernel
            const int sub = (cb * TC) / a.upC;
            const int dy = sub >> 1, dx = sub & 1;
            const unsigned ccol = (unsigned)(a.out_co + cb * TC - sub * a.upC) * 2u;
            const __amdgpu_buffer_rsrc_t rsO =
                __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(unsigned)((size_t)npix * 4 * a.out_cs * 2), 0x00020000);
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int q = tid + 256 * i;
                const int row = q / PPR, pc = q % PPR;
                const u32x4_t o = *reinterpret_cast<const u32x4_t*>(smem + row * OP + pc * 16);
                const int ip = pix0 + row;
                const int x = ip % a.upW, t = ip / a.upW;
                const int y = t % a.upH, bb = t / a.upH;
                const unsigned op = (unsigned)((bb * 2 * a.upH + 2 * y + dy) * (2 * a.upW) + 2 * x + dx);
                const unsigned ob = ip < npix ? op * (unsigned)a.out_cs * 2u + ccol + (unsigned)pc * 16u : kOob;
                __builtin_amdgcn_raw_buffer_store_b128(o, rsO, (int)ob, 0, 0);
            }
            return;
        }
        const __amdgpu_buffer_rsrc_t rsO =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(unsigned)((size_t)npix * a.out_cs * 2), 0x00020000);
        const unsigned obase = ((unsigned)pix0 * (unsigned)a.out_cs + (unsigned)a.out_co + (unsigned)(cb * TC)) * 2u;
        if (a.res != nullptr) {   // out = fp16(x) + fp16(alpha * res) (conv_common.hpp finish16: BottleRep, the accumulating data-gradient convs)
            const float ralpha = a.res_alpha != nullptr ? *a.res_alpha : 1.f;
            const __amdgpu_buffer_rsrc_t rsR =
                __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, (int)(unsigned)((size_t)npix * a.res_cs * 2), 0x00020000);
            const unsigned rbase = ((unsigned)pix0 * (unsigned)a.res_cs + (unsigned)a.res_co + (unsigned)(cb * TC)) * 2u;
            u32x4_t rr[NPC];
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int q = tid + 256 * i;
                const int row = q / PPR, pc = q % PPR;
                const unsigned rb = pix0 + row < npix ? rbase + (unsigned)row * (unsigned)a.res_cs * 2u + (unsigned)pc * 16u : kOob;
                rr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)rb, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int q = tid + 256 * i;
                const int row = q / PPR, pc = q % PPR;
                const u32x4_t o = *reinterpret_cast<const u32x4_t*>(smem + row * OP + pc * 16);
                const h8_t xo = __builtin_bit_cast(h8_t, o), xr = __builtin_bit_cast(h8_t, rr[i]);
                h8_t y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (_Float16)((float)xo[e] + y6_round_f16(ralpha * (float)xr[e]));
                const unsigned ob = pix0 + row < npix ? obase + (unsigned)row * (unsigned)a.out_cs * 2u + (unsigned)pc * 16u : kOob;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, y), rsO, (int)ob, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int q = tid + 256 * i;
            const int row = q / PPR, pc = q % PPR;
            const u32x4_t o = *reinterpret_cast<const u32x4_t*>(smem + row * OP + pc * 16);
            const unsigned ob = pix0 + row < npix ? obase + (unsigned)row * (unsigned)a.out_cs * 2u + (unsigned)pc * 16u : kOob;   // overhang: dropped by the range check
            __builtin_amdgcn_raw_buffer_store_b128(o, rsO, (int)ob, 0, 0);
        }
    }
}

template <int NKS, int WC, int WP, int PF, bool I8>
int launch_pw_nk(const Launch& L, hipStream_t s) {
    constexpr int TP = WP * PF * 32, TC = WC * 32;
    auto kern = conv_pw_kernel<NKS, WC, WP, PF, I8>;
    size_t lds = (size_t)TP * ((NKS + 3) / 4) * 128;
    const size_t epi = (size_t)TP * (TC * 2 + 16);
    if (epi > lds) lds = epi;
    Y6_REQUIRE(lds <= 160 * 1024, "conv_pw: tile needs %zu bytes of LDS", lds);
    if (lds > 64 * 1024) {
        static std::mutex mu;
        static bool big[64] = {};
        int dev = 0;
        Y6_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev >= 0 && dev < 64 && !big[dev]) {
            Y6_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            big[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(L.grid), dim3(256), lds, s, L.k);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

template <int WC, int WP, int PF>
int launch_pw(const Launch& L, hipStream_t s) {
    switch (L.k.Cin / 64) {
        case 1: return launch_pw_nk<4, WC, WP, PF, false>(L, s);
        case 2: return launch_pw_nk<8, WC, WP, PF, false>(L, s);
        case 3: return launch_pw_nk<12, WC, WP, PF, false>(L, s);
        case 4: return launch_pw_nk<16, WC, WP, PF, false>(L, s);
        case 5: return launch_pw_nk<20, WC, WP, PF, false>(L, s);
        case 6: return launch_pw_nk<24, WC, WP, PF, false>(L, s);
        case 8: return launch_pw_nk<32, WC, WP, PF, false>(L, s);
        case 10: return launch_pw_nk<40, WC, WP, PF, false>(L, s);
        case 12: return launch_pw_nk<48, WC, WP, PF, false>(L, s);
        case 16: return launch_pw_nk<64, WC, WP, PF, false>(L, s);
    }
    y6_set_error("conv_pw: no instantiation for Cin %d", L.k.Cin);
    return Y6_EUNSUPPORTED;
}
template <int WC, int WP, int PF>
int launch_pw_i8(const Launch& L, hipStream_t s) {
    switch (L.k.Cin / 32) {
        case 2: return launch_pw_nk<2, WC, WP, PF, true>(L, s);
        case 4: return launch_pw_nk<4, WC, WP, PF, true>(L, s);
        case 6: return launch_pw_nk<6, WC, WP, PF, true>(L, s);
        case 8: return launch_pw_nk<8, WC, WP, PF, true>(L, s);
        case 12: return launch_pw_nk<12, WC, WP, PF, true>(L, s);
        case 16: return launch_pw_nk<16, WC, WP, PF, true>(L, s);
        case 32: return launch_pw_nk<32, WC, WP, PF, true>(L, s);
    }
    y6_set_error("conv_pw: no int8 instantiation for Cin %d", L.k.Cin);
    return Y6_EUNSUPPORTED;
}

}  // namespace

// Cin the kernel is built for (whole 64-channel stages; the reduction is unrolled per instantiation), and the block's whole pixel
// image [block_pixels][Cin] must fit the CU's LDS
int y6_conv_pw_cin_ok(int cin, int block_pixels) {
    if (cin % 64 || (size_t)block_pixels * cin * 2 > 160 * 1024) return 0;
    switch (cin / 64) {
        case 1: case 2: case 3: case 4: case 5: case 6: case 8: case 10: case 12: case 16: return 1;
    }
    return 0;
}

// the int8 form: Cin in {64, 128, 192, 256, 384, 512, 1024}
int y6_conv_pw_i8_cin_ok(int cin, int block_pixels) {
    if (cin % 32 || (size_t)block_pixels * ((cin + 127) / 128) * 128 > 160 * 1024) return 0;
    switch (cin / 32) {
        case 2: case 4: case 6: case 8: case 12: case 16: case 32: return 1;
    }
    return 0;
}

// L points at conv_mfma.hip's launch record (conv_common.hpp)
int y6_conv_pw_launch(const void* Lp, int wc, int pf, int i8, hipStream_t s) {
    const Launch& L = *static_cast<const Launch*>(Lp);
    if (i8) {
        if (wc == 4 && pf == 2) return launch_pw_i8<4, 1, 2>(L, s);
        if (wc == 2 && pf == 2) return launch_pw_i8<2, 2, 2>(L, s);
    } else {
        if (wc == 4 && pf == 2) return launch_pw<4, 1, 2>(L, s);
        if (wc == 2 && pf == 2) return launch_pw<2, 2, 2>(L, s);
    }
    y6_set_error("conv_pw: no instantiation %d cout waves, %d pixel fragments", wc, pf);
    return Y6_EUNSUPPORTED;
}
