// conv_wreg.hip - 3x3 convolution (stride 1 and 2; fp16, and int8 since r04ai), weights through REGISTERS, halo through LDS in
// 64-byte-per-pixel stages (32 fp16 / 64 int8 channels).
//
// Replaces the same aten compositions as conv_dma.hip (reference yolov6/layers/common.py:51-54 ConvModule, :247-248 RepVGGBlock
// deploy branch, :338-339 QARepVGGBlock, :605-608 BottleRep): the fused 3x3 conv + bias (+ post-affine) + activation of the
// deploy graphs, 84 % of the FLOPs of YOLOv6-S.
//
// Why another kernel (round 4).  conv_dma.hip stages BOTH operands through LDS in 16-channel chunks and meets a block-wide
// barrier per chunk: its timeline (tools/dma_trace.py, profiles/r04/trace_*) is 2 300 ideal MFMA cycles per chunk inside a
// 3 600-cycle period - 850 cycles per chunk at the barrier / vmcnt wait (all eight waves together), 62-75 % of the LDS-DMA bytes
// are tap images - and its fixed 128 / 256 / 512-pixel blocks quantise badly (3.1 -> 4 rounds of the persistent walk on the
// 60-GFLOP layers, 224 items for 256 CUs on the 40x40 maps).  Here
//   * a wave owns ONE cout fragment (32 couts) and ALL pixel fragments of the block's tile (PF <= 8): its weight fragments
//     (1 KiB per (tap, k-step), contiguous in the packed weights: [cout/32][cin/32][tap][k-step]) go global -> VGPR, kRing
//     loads in flight, and never touch LDS - no barrier is needed for them and the LDS traffic of a chunk drops to the halo;
//   * the halo image of a stage holds 32 input channels: a barrier every 18 (tap, k-step) units = 18 x PF MFMAs per wave
//     instead of 9 x CF x PF; its LDS-DMA requests are issued a whole stage (thousands of cycles) ahead;
//   * LDS image: pixel-major with an 80-byte pixel pitch - 4 data slots of 16 B + 1 pad slot, an ODD slot pitch - so the 16
//     lanes ds_read_b128 serves per cycle (16 consecutive pixels, same k-half) land on 16 different bank groups with NO
//     swizzle: a tap is a constant byte offset ((dy * RP + dx) * 80 + k-step * 32) from the lane's pixel address, nothing per
//     read but the ds_read itself.  A 1 KiB DMA request covers 12.8 pixels = 13 half cache lines (the planar image of
//     conv_dma.hip touches 64 lines per request); the pad lanes ask for an out-of-range piece;
//   * row pitch RP = TW + 16 (TW + 2 when TW is a multiple of 16): a read group that wraps from one tile row to the next
//     continues at p + 17 - consecutive modulo 16 - so ANY tile width is conflict-free.  That frees the tile shape: the host
//     picks TH x TW so that tiles divide the map and the item count fills whole rounds (10x20 / 5x40 tiles of 200 pixels =
//     7 fragments on the 80x80 / 40x40 / 20x20 maps of YOLOv6 at batch 32: 1 024 / 512 items for 512 resident blocks);
//   * vector-memory ordering: every VMEM instruction of the main loop is inline asm (weight loads, LDS-DMA).  LDS-DMA requests
//     and loads into VGPRs do NOT retire in order with respect to each other on gfx950: a first version counted the halo burst
//     into `s_waitcnt vmcnt(N)` (N = younger weight loads + the burst's requests) and passed every parity test on an idle chip,
//     but beside a bandwidth-hungry kernel on a second stream 299 of 300 runs came out wrong (tools/wreg_stress.py,
//     profiles/r04/wreg_stress_r04h.log).  So: a stage top waits for EVERYTHING this wave has in flight (vmcnt(0): its halo
//     requests of this stage, issued a whole stage ago, and the weight fragments of the first kRing - 1 units) before the
//     barrier; a counted wait (vmcnt(kRing - 1)) appears only where the awaited load AND all younger ones it counts are
//     weight loads into VGPRs - which do retire in order among themselves; requests in flight only make it stricter.
// The epilogue is conv_common.hpp's (bias, post-affine, activation, residual, ragged stores), run at the end of an item while
// the SIMD's other wave (the CU's second block) keeps the matrix pipe busy.
#include "common.hpp"
#include "conv_common.hpp"
#include <type_traits>

namespace {

constexpr int kPix = 80;     // LDS bytes per halo pixel: 32 channels (64 B) + one 16-byte pad slot
constexpr int kMaxP1 = 8;    // at most this many halo requests (1 KiB each) per wave and stage, stride 1
constexpr int kMaxP2 = 12;   // ... stride 2 (the halo of a tile is four times its pixels)
constexpr int kRing = 6;     // weight fragments in flight per wave (divides the 18 units of a stage)
constexpr int kUnits = 18;   // (tap, k-step) units per 32-channel stage

// one weight fragment: lane i gets 16 B from rsrc.base + soff + 16 * i (asynchronously: pair with wait_frag)
__device__ __forceinline__ void load_frag(i32x4_t& dst, const i32x4_t& rsrc, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
// the fragment has landed once at most N younger vector-memory instructions are in flight
template <int N>
__device__ __forceinline__ void wait_frag(i32x4_t& frag) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(frag) : "n"(N) : "memory");
}
// the same with a wave-uniform run-time count (kRing - 1 + the wave's halo requests per stage)
__device__ __forceinline__ void wait_frag_n(i32x4_t& frag, int n) {
    switch (n) {
        case 5: wait_frag<5>(frag); break;
        case 6: wait_frag<6>(frag); break;
        case 7: wait_frag<7>(frag); break;
        case 8: wait_frag<8>(frag); break;
        case 9: wait_frag<9>(frag); break;
        case 10: wait_frag<10>(frag); break;
        case 11: wait_frag<11>(frag); break;
        case 12: wait_frag<12>(frag); break;
        default: wait_frag<13>(frag); break;
    }
}
// one pixel fragment out of LDS, asynchronously (pair with wait_lds): lane i gets the 16 B at addr(lane) + IMM
template <int IMM>
__device__ __forceinline__ void lds_read16(i32x4_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory");
}
// the fragment has landed once at most N younger LDS reads are in flight (LDS returns in order; anything else counted by lgkmcnt
// only makes the wait stricter)
template <int N>
__device__ __forceinline__ void wait_lds(i32x4_t& frag) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// n / d for small non-negative n (below 2^16) with the host's 1.0f / d: exact - (n + 0.5) / d is at least 0.5 / d away from an integer
__device__ __forceinline__ int div_small(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

// Ceiling probes / trace (tools/build_probe_libs.py --wreg n; WRONG RESULTS for n >= 2, timing only), compile-time:
//   1 = s_memtime trace of block 0 / thread 0 into a.dbg (tools/dma_trace.py --wreg); 2 = no weight loads after the prologue (the
//   waits stay); 4 = no halo requests after the prologue; 5 = no epilogue;
//   6 = no MFMAs; 7 = no stage barrier; 9 (WRONG under load, see the header) = the first version's counted waits across the halo burst
#ifndef Y6_WREG_PROBE
#define Y6_WREG_PROBE 0
#endif
constexpr int kWregProbe = Y6_WREG_PROBE;

// I8: the int8 form (include/yolov6_hip.h: y6_conv_i8_desc).  v_mfma_i32_32x32x32_i8 takes the same 16 bytes per lane and operand
// as the fp16 instruction, so a stage is 64 int8 channels in the SAME 64-byte pixel slots, a weight fragment is the same 1 KiB
// (quant.hip's packing: [cout/32][cin/64][tap][k-step]) and the request / wait / barrier protocol below is shared word for word;
// what differs is the element size of the input view, int32 accumulators and the epilogue's dequantisation (+ the int8 twin).
// RELU_ONLY (round 5): the epilogue specialised at compile time for what 32 of the 38 register-fed launches of YOLOv6-S need - bias +
// ReLU into a 16-byte aligned fp16 view, no post-affine, no residual.  The general form's epilogue is ~20 KB of straight-line code
// per item (seven unrolled fast_unit copies, each carrying the post-affine / SiLU / hardswish arithmetic behind wave-uniform
// branches) plus the out-of-line general epilogue; code that runs once per item is executed at instruction-fetch latency whenever
// the function's lines have left the 64 KB instruction cache, i.e. after two or three other kernels (22 500 cycles
// instead of 7 200 for the first item's epilogue).  This form's epilogue is ~3 KB and the function ~40 KB instead of 110-180 KB.
// Same arithmetic in the same order: bit-identical outputs (tests/test_gpu_ops.py; A/B switch Y6_WREG_GENERAL_EPI=1).
// EPI: 0 = the general form, 1 = bias + ReLU only (RELU_ONLY above), 2 = bias + SiLU only (the head's cls / reg convs,
// effidehead.py:172-181: ConvBNSiLU - 3 launches of YOLOv6-S, one of them 108 us cold against 55 warm on the general form),
// 3 = bias only (the convs of the training-form graph: their BatchNorm is a kernel of its own), 4 (round 6) = bias only, ADDED to what the
// output view holds (res == out: the accumulating data-gradient convs of the training step - 24 launches per step that ran the
// general epilogue, a residual round trip per fragment: 132 us a launch against 82 for the same layers forward); the sixteen-byte
// pieces a lane will store are loaded for ALL its fragments before the first one is finished.
template <int PF, int WC, int WP, int ST, bool I8, int EPI = 0>
__global__ __launch_bounds__(WC * WP * 64, 2) void conv3x3_wreg_kernel(const ConvKArgs a) {
    constexpr bool RELU_ONLY = EPI == 1, SILU_ONLY = EPI == 2, SPECIAL = EPI != 0, ACCUM = EPI == 4;   // EPI == 3: bias only (the training step's convs: BatchNorm follows)
    constexpr int kMaxP = ST == 2 ? kMaxP2 : kMaxP1;   // halo requests per wave and stage
    constexpr int ES = I8 ? 1 : 2;                     // bytes per input element
    typedef typename std::conditional<I8, i32x16_t, f32x16_t>::type acc_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WC * WP;
    constexpr int R = kRing;
    static_assert(kUnits % R == 0, "ring slots are compile-time indices");
    static_assert(R - 1 + kMaxP1 == 13 && R - 1 == 5, "wait_frag_n's cases");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WC, wp = wave / WC;
    const int RP = a.dma_rp, NHP = a.dma_nhp;
    const unsigned stage_bytes = (unsigned)NHP * 1024u;
    const unsigned smem_base = lds_addr(smem);
    const int nsc = I8 ? (a.Cin + 63) >> 6 : a.Cin >> 5;   // stages of 64 bytes per pixel
    // int8 with fewer than 64 input channels (the 32 -> 64 stride-2 conv behind the stem): one stage whose upper pieces are not
    // requested (LDS-DMA writes zeros for an out-of-range piece, the packed weights hold zeros there as well)
    const int jmax = (I8 && a.Cin < 64) ? a.Cin >> 4 : 4;
    const int nids = a.nids;
    const int gstride = gridDim.x;
    const int ics = I8 ? a.qin_cs : a.in_cs, ico = I8 ? a.qin_co : a.in_co;

    const i32x4_t rsA = make_rsrc(I8 ? (const void*)a.qin : (const void*)a.in, (unsigned)((size_t)a.B * a.H * a.W * ics * ES));
    const i32x4_t rsW = make_rsrc(a.wpk, 0xfffffe00u);

    auto decode = [&](int id, int& tile, int& cb) {
        if (a.ncb == 1) {
            tile = id;
            cb = 0;
        } else {
            const int lo = id & 7, r = id >> 3;
            cb = r % a.ncb;
            tile = (r / a.ncb) * 8 + lo;
        }
    };
    auto next_valid = [&](int id) {
        for (id += gstride; id < nids; id += gstride) {
            int t, c;
            decode(id, t, c);
            if (t < a.ntiles) break;
        }
        return id;
    };
    int id = blockIdx.x;
    int cb = 0;
    {
        int t;
        decode(id, t, cb);
        if (t >= a.ntiles) id = next_valid(id);
    }
    if (id >= nids) return;
    if (a.prio_mode & 2) __builtin_amdgcn_s_setprio(3);
    int dbg_n = 0;
    const bool tracing = kWregProbe == 1 && a.dbg != nullptr && blockIdx.x == 0 && tid == 0;
#define DT(tag)                                                              \
    do {                                                                     \
        if (kWregProbe == 1 && tracing && dbg_n < 256) {                     \
            a.dbg[2 * dbg_n] = __builtin_amdgcn_s_memtime();                 \
            a.dbg[2 * dbg_n + 1] = (unsigned long long)(tag);                \
            ++dbg_n;                                                         \
        }                                                                    \
    } while (0)
    DT(1);
    if (kWregProbe == 1 && a.dbg != nullptr && tid == 0 && blockIdx.x < 1024)   // every block's start / end on the 100 MHz counter
        a.dbg[1024 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    if (kWregProbe == 1 && tracing) {   // the constant 100 MHz counter beside the shader clock: the kernel's effective frequency
        a.dbg[2 * dbg_n] = __builtin_amdgcn_s_memrealtime();
        a.dbg[2 * dbg_n + 1] = 90;
        ++dbg_n;
    }
    {
        int t;
        decode(id, t, cb);   // the grid stride is a multiple of 8 * ncb: every item of this block has this cout block
    }

    // ---- this wave's halo requests: slot s = 64 * P + lane of the stage image (P = wave + NW * i < NHP) is piece j of halo pixel
    //      p = s / 5 (j = 4: the pad slot).  Their byte offsets into the input view are the same for every stage of an item (the
    //      stage's channels are the scalar offset): computed once per item.
    const int npw = NHP > wave ? (NHP - wave + NW - 1) / NW : 0;
    unsigned hvoff[kMaxP];
    bool in_loop = false;
    auto setup_halo = [&](int iy0, int ix0, unsigned base) {
#pragma unroll
        for (int i = 0; i < kMaxP; ++i) {
            int l = lane;
            asm volatile("" : "+v"(l));   // opaque: otherwise hipcc hoists (row, column, piece) of every request out of the item loop and spills them
            const int s = (wave + NW * i) * 64 + l;
            const int p = s / 5, j = s - 5 * p;
            const int hy = div_small(p, a.inv_rp), hc = p - hy * RP;
            // stride 2: a halo row is stored as [even columns (TW + 1) | odd columns (TW)], so that the 16 lanes of a read group -
            // consecutive output pixels, input columns two apart - still read 16 consecutive pixel slots
            const int hx = ST == 1 ? hc : (hc <= a.TW ? 2 * hc : 2 * (hc - a.TW - 1) + 1);
            const bool v = (i < npw) && (j < jmax) && (hy < a.HH) && (ST == 1 || hc <= 2 * a.TW) && ((unsigned)(iy0 + hy) < (unsigned)a.H) && ((unsigned)(ix0 + hx) < (unsigned)a.W);
            hvoff[i] = v ? base + (unsigned)(hy * a.W + hx) * (unsigned)(ics * ES) + (unsigned)(j * 16) : kOob;
        }
    };
    // requests of one stage (always npw of them: the counted waits below rely on it; behind the last stage they ask for nothing)
    auto issue_halo = [&](bool real, unsigned soff, unsigned dst0) {
#pragma unroll
        for (int i = 0; i < kMaxP; ++i) {
            if (i < npw && (kWregProbe != 4 || !in_loop)) dma16(rsA, real ? hvoff[i] : kOob, soff, dst0 + (unsigned)(wave + NW * i) * 1024u);
        }
    };
    auto tile_origin = [&](int item, int& iy0, int& ix0, unsigned& base) {
        int tile, c;
        decode(item, tile, c);
        const int tx_i = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty_i = t2 % a.tiles_y;
        const int b = t2 / a.tiles_y;
        iy0 = ty_i * a.TH * ST - 1;
        ix0 = tx_i * a.TW * ST - 1;
        // modulo 2^32 (tensors up to 3.5 GiB): the origin may lie one row / column outside the image
        base = (((unsigned)(b * a.H + iy0) * (unsigned)a.W + (unsigned)ix0) * (unsigned)ics + (unsigned)ico) * (unsigned)ES;
    };

    const int fq = frag_pixel(lane & 31);   // fragment pixel this lane holds
    auto out_pix = [&](const ConvKArgs& ea, int item, int (&opix)[PF]) {
        int tile = item;
        if (ea.ncb != 1) tile = ((item >> 3) / ea.ncb) * 8 + (item & 7);
        const int tx_i = tile % ea.tiles_x;
        const int t2 = tile / ea.tiles_x;
        const int ty_i = t2 % ea.tiles_y;
        const int b = t2 / ea.tiles_y;
        const int oy0 = ty_i * ea.TH, ox0 = tx_i * ea.TW;
        const int base = (b * ea.Ho + oy0) * ea.Wo + ox0;
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) {   // (recomputed per item - opaque, or hipcc hoists and spills them: registers the main loop does not carry)
            int q = fq;
            asm volatile("" : "+v"(q));
            const int m = wp * (PF * 32) + pf * 32 + q;
            const int ty = div_small(m, ea.inv_tw), tx = m - ty * ea.TW;
            const bool v = m < ea.TH * ea.TW && (oy0 + ty < ea.Ho) && (ox0 + tx < ea.Wo);
            opix[pf] = v ? base + ty * ea.Wo + tx : -1;
        }
    };

    // ---- the weight stream of this wave: cout fragment g = cb * WC + wc, nsc stages of 18 KiB
    const unsigned wbase = (unsigned)((cb * WC + wc) * nsc) * (unsigned)(kUnits * 1024);
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned woff_cur = wbase;                                            // stage being multiplied
    unsigned woff_next = nsc > 1 ? wbase + kUnits * 1024 : wbase;         // the one after it (the next item restarts the stream)
    i32x4_t wr[R];

    acc_t acc[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pf][r] = 0;

    // ---- fast epilogue (conv + bias (+ QARepVGG post-affine) + activation, no residual, into a 16-byte aligned fp16 view - every
    //      3x3 of the deploy graphs but the BottleRep shortcut convs): per-channel vectors of the block's couts in LDS (the cout
    //      block is the same for all its items: written once, visible behind the first stage barrier), stores through a buffer
    //      descriptor with 32-bit byte offsets.  Anything else takes conv_common.hpp's general epilogue.
    //      The int8 form multiplies the exact int32 sums by s_x * s_w[c] first (a rounding of its own, as every int8 kernel of
    //      this library) and may write the int8 twin of its output for quantised consumers - with or without the fp16 view.
    float* ldsVec = reinterpret_cast<float*>(smem + 2u * stage_bytes);   // [bias | post scale | post shift | dequant][WC * 32]
    const bool has_post = !SPECIAL && a.pscale != nullptr;
    const bool has_out = a.out != nullptr, has_qout = I8 && a.qout != nullptr;
    const bool fast = (ACCUM || a.res == nullptr) && a.up == 0 && (has_out || has_qout) && (!has_out || a.vec16_ok) &&
                      (!has_qout || ((a.qout_cs | a.qout_co) & 3) == 0) && (!I8 || a.acc_out == nullptr) &&
                      (size_t)a.B * a.Ho * a.Wo * a.out_cs * 2 < 0xe0000000ull;
    const float fast_lo = (RELU_ONLY || a.act == Y6_ACT_RELU) ? 0.f : -__builtin_inff();
    const bool smooth_act = SILU_ONLY || (!SPECIAL && (a.act == Y6_ACT_SILU || a.act == Y6_ACT_HARDSWISH));
    const __amdgpu_buffer_rsrc_t rsO =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(unsigned)((size_t)a.B * a.Ho * a.Wo * a.out_cs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsQ =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.qout, 0, (int)(unsigned)((size_t)a.B * a.Ho * a.Wo * (I8 ? a.qout_cs : 0)), 0x00020000);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    auto fast_unit = [&](const acc_t& accv, unsigned obyte, unsigned qbyte, const float (&bias16)[16], const u32x4_t* prev) {
        const int kh = lane >> 5;
        const float* lb = ldsVec + wc * 32;
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float x[4] = {(float)accv[g * 4 + 0], (float)accv[g * 4 + 1], (float)accv[g * 4 + 2], (float)accv[g * 4 + 3]};
            if constexpr (I8) {   // exact int32 -> fp32, * s_x * s_w[c] as a rounding of its own (no fma with the bias add)
                const float4 qs = *reinterpret_cast<const float4*>(lb + 3 * WC * 32 + 8 * g + 4 * kh);
                x[0] *= qs.x;
                x[1] *= qs.y;
                x[2] *= qs.z;
                x[3] *= qs.w;
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(x[j]));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] += bias16[g * 4 + j];
            if (has_post) {   // QARepVGG: conv -> BatchNorm are two fp16 ops (finish16 in conv_common.hpp)
                const float4 ps = *reinterpret_cast<const float4*>(lb + WC * 32 + 8 * g + 4 * kh);
                const float4 pt = *reinterpret_cast<const float4*>(lb + 2 * WC * 32 + 8 * g + 4 * kh);
                x[0] = y6_round_f16(x[0]) * ps.x + pt.x;
                x[1] = y6_round_f16(x[1]) * ps.y + pt.y;
                x[2] = y6_round_f16(x[2]) * ps.z + pt.z;
                x[3] = y6_round_f16(x[3]) * ps.w + pt.w;
            }
            if (smooth_act) {   // one wave-uniform branch per group, the arithmetic of act_const<>
#pragma unroll
                for (int j = 0; j < 4; ++j) v[g * 4 + j] = (SILU_ONLY || a.act == Y6_ACT_SILU) ? act_const<Y6_ACT_SILU>(x[j]) : act_const<Y6_ACT_HARDSWISH>(x[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[g * 4 + j] = fmaxf(x[j], fast_lo);
            }
        }
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
        const unsigned ocol = (unsigned)((cb * WC + wc) * 32 + 8 * kh);
        if constexpr (I8) {
            if (has_qout) {   // the int8 twin: the SAME fp16 values, quantised with the consumers' scale; channels cfrag + 8 g + 4 kh ..
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned q = q8_quad(pk[g][0], pk[g][1], a.qo_inv2, a.qo_lo2, a.qo_hi2);
                    __builtin_amdgcn_raw_buffer_store_b32(q, rsQ, (int)(qbyte + ocol + (unsigned)(8 * g) - (unsigned)(4 * kh)), 0, 0);
                }
            }
            if (!has_out) return;
        }
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {   // pair groups across the two half-waves: one 16-byte store per lane and pair
            auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
            u32x4_t o = {s0[0], s1[0], s0[1], s1[1]};
            if constexpr (ACCUM) {   // fp16(x) + what the view held (conv_common.hpp finish16 with res == out, alpha 1: the same two roundings)
                const h8_t xo = __builtin_bit_cast(h8_t, o), xr = __builtin_bit_cast(h8_t, prev[gp]);
                h8_t y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (_Float16)((float)xo[e] + (float)xr[e]);
                o = __builtin_bit_cast(u32x4_t, y);
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, rsO, (int)(obyte + (ocol + (unsigned)(16 * gp)) * 2u), 0, 0);
        }
        // (r04n: the same stores as 64-byte runs - each fragment transposed through a per-wave LDS scratch so that four adjacent
        // lanes hold one pixel's 64 B - took 22.9 k cycles per item instead of 12.9 k: the end-of-launch store burst, 26 MB from
        // every block at once, is bound by the fabric's write bandwidth (7 us at 3.7 TB/s), not by the request count.  Removed.
        // r04w: cache-policy bits on these stores - sc1 / sc0 sc1 / nt sc1 (write-through) lose 5-7 % of the class, nt is level with the
        // default write-back policy (profiles/r04/bench_r04w_st*.json).  Default kept.)
    };

    // prologue: the first stage of the first item, the first R - 1 weight fragments
    int n_item = id;          // item whose stage is requested next
    int n_sc = 0;
    {
        int iy0, ix0;
        unsigned base;
        tile_origin(n_item, iy0, ix0, base);
        setup_halo(iy0, ix0, base);
        issue_halo(true, 0u, smem_base);
    }
#pragma unroll
    for (int u = 0; u < R - 1; ++u) load_frag(wr[u], rsW, lane16, woff_cur + (unsigned)(u * 1024));
    // the per-channel vectors of the block's couts (behind the first requests: their latency is the prologue's anyway)
    if (tid < WC * 32) {
        const int c = cb * WC * 32 + tid;
        ldsVec[tid] = a.bias != nullptr ? a.bias[c] : 0.f;
        ldsVec[WC * 32 + tid] = has_post ? a.pscale[c] : 1.f;
        ldsVec[2 * WC * 32 + tid] = has_post ? a.pshift[c] : 0.f;
        if constexpr (I8) ldsVec[3 * WC * 32 + tid] = a.qscale[c];
    }

    __builtin_amdgcn_sched_barrier(0);   // the requests above leave first; the arithmetic below fills their latency
    // this lane's pixels: LDS offset of the tap (0, 0) read (fixed for the whole kernel)
    unsigned pixaddr[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int m = wp * (PF * 32) + pf * 32 + fq;
        const int npx = a.TH * a.TW;
        const int mm = m < npx ? m : npx - 1;
        const int ty = div_small(mm, a.inv_tw), tx = mm - ty * a.TW;
        pixaddr[pf] = (unsigned)((ST * ty * RP + tx) * kPix + (lane >> 5) * 16);
    }

    const unsigned rp_bytes = (unsigned)(RP * kPix);
    const unsigned odd_bytes = (unsigned)((a.TW + 1) * kPix);   // stride 2: first odd-column slot of a halo row
    int sc = 0, stage = 0, gstage = 0;
    in_loop = true;
    DT(2);
    while (true) {
        DT(9);
        // ---- stage top.  First the bookkeeping of the NEXT stage's requests (the next item's first stage behind this item's
        // last; a new item: its tile origin and the byte offsets of this wave's requests) - VALU / SALU work that overlaps the
        // landing of the youngest weight fragment.  Then everything this wave has in flight - its halo requests of THIS stage
        // (issued a whole stage ago), the weight fragments of the first R - 1 units (the youngest one unit ago), an epilogue's
        // stores - and the barrier.  Then the requests, into the half everybody has finished reading.
        bool real = true;
        if (n_sc + 1 < nsc) {
            ++n_sc;
        } else {
            const int n = next_valid(n_item);
            if (n >= nids) {
                real = false;
            } else {
                n_item = n;
                n_sc = 0;
            }
        }
        if (real && n_sc == 0) {
            int iy0, ix0;
            unsigned base;
            tile_origin(n_item, iy0, ix0, base);
            setup_halo(iy0, ix0, base);
        }
        if (kWregProbe == 7) {
            if (sc == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else if (sc == 0 || kWregProbe != 9) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        DT(10);
        issue_halo(real, (unsigned)n_sc * 64u, smem_base + (stage ? 0u : stage_bytes));
        // The two blocks of a CU put one wave each on every SIMD, and the hardware arbitrates their issue by age: the older block's
        // waves win the matrix pipe whenever both want it, finish first and leave the younger block to run alone with nothing beside
        // its request / epilogue phases.  Wave priority as a counter-measure (a.prio_mode, A/B switch Y6_WREG_PRIO):
        //   bit 0: the matrix phase's priority falls from stage to stage (2, 1, 0, 2, ...): a block that lags by a stage has the
        //          higher priority two times out of three, so the lag stays within a stage or so and both finish together;
        //   bit 1: the non-matrix phases (stage top, requests, epilogue) run at priority 3: a wave between two matrix phases gets
        //          every issue slot it can use and is back at the matrix pipe sooner; the partner's MFMAs need one slot in 32.
        if (a.prio_mode & 1) {
            switch (gstage % 3) {
                case 0: __builtin_amdgcn_s_setprio(2); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
            }
        } else if (a.prio_mode & 2) {
            __builtin_amdgcn_s_setprio(0);
        }
        ++gstage;
        DT(11);
        // ---- the stage's 18 x PF fragment products.  The pixel fragments come out of LDS by inline asm with hand-counted lgkmcnt
        // (frag_ring below): fragment i + G is requested right behind the MFMA of fragment i, so G - 1 = PF - 1 reads are in flight behind
        // the one an MFMA waits for (a ring of two units, G = 2 PF, measured 1-2 % slower: r04m).  (hipcc's own counting forces lgkmcnt(0) in front of the first MFMA of every unit - the kernel
        // contains FLAT instructions in its general epilogue, and with a FLAT access "pending" forever (it never sees a vmcnt wait:
        // those are asm too) it may not assume in-order return - and it clusters the reads behind the MFMAs: every unit started
        // with a full LDS round trip.)
        const unsigned Aoff = smem_base + (stage ? stage_bytes : 0u);
        constexpr int G = PF, TOTAL = kUnits * PF;
        i32x4_t fb[G];
        auto frag_read = [&](auto ic) {   // request fragment i = (unit, pf) of this stage into its ring slot
            constexpr int i = decltype(ic)::value;
            constexpr int u = i / PF, pf = i % PF, tap = u >> 1;
            if constexpr (ST == 1) {
                const unsigned addr = Aoff + pixaddr[pf] + (unsigned)(tap / 3) * rp_bytes;
                lds_read16<(tap % 3) * kPix + (u & 1) * 32>(fb[i % G], addr);
            } else {   // input column 2 tx + dx: dx = 0 / 2 are even-plane slots tx / tx + 1, dx = 1 is odd-plane slot tx
                const unsigned addr = Aoff + pixaddr[pf] + (unsigned)(tap / 3) * rp_bytes + ((tap % 3) == 1 ? odd_bytes : 0u);
                lds_read16<((tap % 3) == 2 ? kPix : 0) + (u & 1) * 32>(fb[i % G], addr);
            }
        };
        static_for<0, G>(frag_read);
        static_for<0, TOTAL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int u = i / PF, pf = i % PF;
            if constexpr (pf == 0) {
                {   // the weight fragment R - 1 units ahead, into the slot unit u - 1 just released
                    constexpr int up = u + R - 1;
                    const unsigned so = up < kUnits ? woff_cur + (unsigned)(up * 1024) : woff_next + (unsigned)((up - kUnits) * 1024);
                    if (kWregProbe != 2) load_frag(wr[up % R], rsW, lane16, so);
                }
                // in flight behind this unit's fragment: the R - 1 younger fragments, and - for the fragments requested before this
                // stage's top - the halo burst
                if constexpr (u < R - 1) {   // landed at the stage top (tie the value to this point: no instruction)
                    if (kWregProbe == 9)
                        wait_frag_n(wr[u % R], R - 1 + npw);
                    else
                        asm volatile("" : "+v"(wr[u % R]));
                } else {                     // requested in this stage, behind the burst: only weight loads are younger
                    wait_frag<R - 1>(wr[u % R]);
                }
            }
            wait_lds<(TOTAL - 1 - i < G - 1 ? TOTAL - 1 - i : G - 1)>(fb[i % G]);
            if constexpr (I8) {
                if (kWregProbe != 6) acc[pf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wr[u % R], fb[i % G], acc[pf], 0, 0, 0);
            } else {
                if (kWregProbe != 6)
                    acc[pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, wr[u % R]), __builtin_bit_cast(h8_t, fb[i % G]), acc[pf], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i + G < TOTAL) {
                frag_read(std::integral_constant<int, i + G>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        DT(12);
        if (a.prio_mode & 2) __builtin_amdgcn_s_setprio(3);
        // ---- next stage
        stage ^= 1;
        ++sc;
        woff_cur = woff_next;
        if (sc == nsc) {
            // the item is complete: bias (+ post-affine) + activation (+ residual) -> fp16 NHWC
            if (fast) {
                int opix[PF];
                out_pix(a, id, opix);
                float bias16[16];   // the wave's 32 couts are the same for all its fragments: one read per item
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bz = *reinterpret_cast<const float4*>(ldsVec + wc * 32 + 8 * g + 4 * (lane >> 5));
                    bias16[g * 4 + 0] = bz.x;
                    bias16[g * 4 + 1] = bz.y;
                    bias16[g * 4 + 2] = bz.z;
                    bias16[g * 4 + 3] = bz.w;
                }
                DT(20);
                u32x4_t prev[ACCUM ? PF : 1][2];
                if constexpr (ACCUM) {   // every piece this lane is going to add to, requested before the first fragment is finished
                    const unsigned ocol2 = (unsigned)((cb * WC + wc) * 32 + 8 * (lane >> 5)) * 2u;
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) {
                        const unsigned ob = opix[pf] >= 0 ? ((unsigned)opix[pf] * (unsigned)a.out_cs + (unsigned)a.out_co) * 2u : kOob;
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) prev[pf][gp] = __builtin_amdgcn_raw_buffer_load_b128(rsO, (int)(ob + ocol2 + (unsigned)(32 * gp)), 0, 0);
                    }
                }
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) {
                    const unsigned ob = opix[pf] >= 0 ? ((unsigned)opix[pf] * (unsigned)a.out_cs + (unsigned)a.out_co) * 2u : kOob;   // overhang: dropped by the range check
                    const unsigned qb = (I8 && opix[pf] >= 0) ? (unsigned)opix[pf] * (unsigned)a.qout_cs + (unsigned)a.qout_co : kOob;
                    if (kWregProbe != 5) fast_unit(acc[pf], ob, qb, bias16, prev[ACCUM ? pf : 0]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[pf][r] = 0;
                }
            } else if constexpr (I8 || SPECIAL) {   // the host sends only fast-path layers here (y6_conv_i8_variant / wreg_special_epilogue)
                __builtin_trap();
            } else {
                const ConvKArgs ea = reload_args();
                int opix[PF];
                out_pix(ea, id, opix);
                BiasRegs<1> bz;
                load_bias<1>(ea, cb * WC + wc, 0, lane, bz);
                DT(20);
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) {
                    const int op1[1] = {opix[pf]};
                    if (kWregProbe != 5) conv_epilogue<1, 1>(ea, *reinterpret_cast<const f32x16_t(*)[1][1]>(&acc[pf]), op1, cb * WC + wc, 0, lane, bz);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[pf][r] = 0.f;
                }
            }
            DT(21);
            const int nid = next_valid(id);
            if (nid >= nids) break;
            id = nid;
            sc = 0;
        }
        woff_next = (sc + 1 < nsc) ? woff_cur + kUnits * 1024 : wbase;
    }
    // the fragments requested past the end and the dummy requests of the last stage land before the wave ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DT(22);
    if (kWregProbe == 1 && a.dbg != nullptr && tid == 0 && blockIdx.x < 1024) a.dbg[1024 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    if (kWregProbe == 1 && tracing && dbg_n < 256) {
        a.dbg[2 * dbg_n] = __builtin_amdgcn_s_memrealtime();
        a.dbg[2 * dbg_n + 1] = 91;
        ++dbg_n;
    }
#undef DT
}

// (Round 4, r04k: a ping-pong form - ONE 8-wave block whose two 4-wave groups strictly alternate matrix phases and request /
// epilogue phases between block barriers, the structure of the tuned attention kernels - was built, passed parity and the stress
// test, and ran 13 % SLOWER than the two free-running blocks above on the same tiles (profiles/r04/conv_bench_pingpong_r04k.json):
// a lone wave per SIMD issues MFMAs at 89 % of the pipe's rate, and every phase lasts as long as the longer of the two partners.
// Removed; git history has it.)

// which specialised epilogue a launch can take: the kernel's fast-epilogue condition (its `fast`) without a post-affine, with ReLU
// (1) or SiLU (2); 0: the general form
static int wreg_special_epilogue(const ConvKArgs& k) {
    static const bool off = getenv("Y6_WREG_GENERAL_EPI") != nullptr;   // A/B switch
    static const bool acc_on = getenv("Y6_WREG_ACC") ? atoi(getenv("Y6_WREG_ACC")) != 0 : true;   // A/B switch (4)
    if (off || k.pscale != nullptr || k.up != 0 || k.out == nullptr || !k.vec16_ok || (size_t)k.B * k.Ho * k.Wo * k.out_cs * 2 >= 0xe0000000ull)
        return 0;
    if (k.res != nullptr)   // accumulate into the output view itself, no scaling, no activation
        return (acc_on && k.res == k.out && k.res_cs == k.out_cs && k.res_co == k.out_co && k.res_alpha == nullptr && k.act == Y6_ACT_NONE) ? 4 : 0;
    return k.act == Y6_ACT_RELU ? 1 : (k.act == Y6_ACT_SILU ? 2 : (k.act == Y6_ACT_NONE ? 3 : 0));
}

template <int PF, int WC, int WP, int ST, bool I8 = false, int EPI = 0>
int launch_wreg(const Launch& L, hipStream_t s) {
    if constexpr (!I8 && EPI == 0) {
        switch (wreg_special_epilogue(L.k)) {
            case 1: return launch_wreg<PF, WC, WP, ST, false, 1>(L, s);
            case 2: return launch_wreg<PF, WC, WP, ST, false, 2>(L, s);
            case 3: return launch_wreg<PF, WC, WP, ST, false, 3>(L, s);
            case 4:
                if constexpr (ST == 1) return launch_wreg<PF, WC, WP, ST, false, 4>(L, s);   // (data-gradient convs are stride-1 convs)
                break;
        }
    }
    auto kern = conv3x3_wreg_kernel<PF, WC, WP, ST, I8, EPI>;
    Y6_REQUIRE(L.lds <= 160 * 1024, "conv_wreg: tile needs %zu bytes of LDS", L.lds);
    static OccupancyCache occ;
    int grid = 0;
    {
        int rc = resident_grid(occ, kern, WC * WP * 64, L.lds, 160 * 1024, &grid);
        if (rc) return rc;
    }
    const int gq = 8 * L.k.ncb;   // ids of one tile's cout blocks share id % 8 (XCD); (id >> 3) % ncb - the cout block - is the same for every item of a block
    grid -= grid % gq;
    if (grid < gq) grid = gq;
    if (grid > L.grid) grid = L.grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WC * WP * 64), L.lds, s, L.k);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

}  // namespace

int y6_conv_wreg_max_pieces(int nw, int stride) { return (stride == 2 ? kMaxP2 : kMaxP1) * nw; }

// L points at conv_mfma.hip's launch record (conv_common.hpp)
int y6_conv_wreg_launch(const void* Lp, int pf, int wc, int wpx, int stride, int i8, hipStream_t s) {
    const Launch& L = *static_cast<const Launch*>(Lp);
    if (i8) {   // conv_mfma.hip: y6_conv_i8_launch (variants 10 - 13)
        if (stride == 1 && wc == 4 && wpx == 1) {
            switch (pf) {
                case 7: return launch_wreg<7, 4, 1, 1, true>(L, s);
                case 4: return launch_wreg<4, 4, 1, 1, true>(L, s);
            }
        }
        if (stride == 2 && wc == 4 && wpx == 1 && pf == 3) return launch_wreg<3, 4, 1, 2, true>(L, s);
        // 64-cout blocks at stride 2: two cout waves x two pixel waves (r04ak: 64 -> 64 @160 -> 80: 36 us against 51 for the per-tap
        // kernel.  The stride-1 form of the same shape - 256 pixel slots, 16 x 16 tiles - lost 1-4 us per layer to the LDS-DMA kernel
        // on the 64 -> 64 layers of the 160 x 160 / 80 x 80 maps and was deleted.)
        if (stride == 2 && wc == 2 && wpx == 2 && pf == 2) return launch_wreg<2, 2, 2, 2, true>(L, s);
        y6_set_error("conv_wreg: no int8 instantiation pf %d, %d x %d waves, stride %d", pf, wc, wpx, stride);
        return Y6_EUNSUPPORTED;
    }
    if (stride == 2 && wc == 4 && wpx == 1) {
        switch (pf) {
            case 4: return launch_wreg<4, 4, 1, 2>(L, s);
            case 3: return launch_wreg<3, 4, 1, 2>(L, s);
        }
    }
    if (stride == 1 && wc == 4 && wpx == 1) {
        switch (pf) {
            case 7: return launch_wreg<7, 4, 1, 1>(L, s);
            case 6: return launch_wreg<6, 4, 1, 1>(L, s);
            case 5: return launch_wreg<5, 4, 1, 1>(L, s);
            case 4: return launch_wreg<4, 4, 1, 1>(L, s);
        }
    }
    y6_set_error("conv_wreg: no instantiation pf %d, %d x %d waves, stride %d", pf, wc, wpx, stride);
    return Y6_EUNSUPPORTED;
}
