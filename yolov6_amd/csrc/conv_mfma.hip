// conv_mfma.hip — NHWC fp16 direct convolution (k in {1,3}, stride in {1,2}) on the gfx950
// matrix cores, im2col-free: a spatial tile's input halo is staged ONCE per 32-channel
// chunk into LDS and re-used by all 9 taps; weights arrive by LDS-DMA (global_load_lds)
// in pre-packed MFMA fragment order through a 3-slot ring, two taps ahead.
//
// Replaces the aten compositions behind ConvModule.forward_fuse (reference
// yolov6/layers/common.py:51-54), RepVGGBlock deploy forward (:247-248), QARepVGGBlock
// deploy forward (:338-339), BottleRep's residual (:605-608) and the head's 1x1 convs
// (yolov6/models/effidehead.py:169-179).
//
// MFMA orientation (v_mfma_f32_32x32x16_f16, guide cdna_hip_programming.md §3):
//   A operand  = weights : row = cout (lane&31), k = cin (8 consecutive at (lane>>5)*8)
//   B operand  = pixels  : col = pixel (lane&31), same k
//   C/D        : col = pixel = lane&31, row = cout = (r&3) + 8*(r>>2) + 4*(lane>>5)
// so every lane ends up with 4 consecutive couts of ONE pixel per 4 accumulator registers
// -> 8-byte NHWC stores in the epilogue.
#include <cstdlib>

#include "common.hpp"
#include "conv_common.hpp"

namespace {

// ACT: the activation as a compile-time fact (Y6_ACT_RELU / Y6_ACT_SILU instantiations exist for the 1x1 stride-1 form: the 14
// 1x1 launches of a YOLOv6-S step), -1: decided per launch.  The same epilogue functions, the same arithmetic: bit-identical.
template <int CF, int PF, int KS, int ST, int ACT = -1>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = KS * KS;
    constexpr int WIMG = CF * 2 * 1024;  // bytes of one tap's weight image [cf][ks][lane][16B]
    constexpr int MAXHP = HaloCap<KS, ST, PF>::value;
    constexpr int NP = (MAXHP * 4 + 255) / 256;  // 16-byte halo pieces per thread
    constexpr int NWJ = (CF * 2 + 3) / 4;        // DMA pieces per wave per tap

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> (spatial tile, cout block); cout blocks of one tile share an XCD (id % 8)
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
    const int tx_i = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_i = t2 % a.tiles_y;
    const int b = t2 / a.tiles_y;
    const int oy0 = ty_i * a.TH, ox0 = tx_i * a.TW;
    const int iy0 = oy0 * ST - KS / 2, ix0 = ox0 * ST - KS / 2;

    char* ldsA = smem;
    char* ldsW = smem + a.ldsA_bytes;

    // ---- per-thread halo piece table (element offsets into a.in; <0: zero fill / skip)
    const int npieces = a.HH * a.HWd * 4;
    int goff[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = tid + i * 256;
        int g = -2;
        if (idx < npieces) {
            const int hp = idx >> 2, q = idx & 3;
            const int hy = hp / a.HWd, hx = hp - hy * a.HWd;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool v = (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
            g = v ? (((b * a.H + iy) * a.W + ix) * a.in_cs + a.in_co + q * 8) : -1;
        }
        goff[i] = g;
    }

    // ConvTranspose2d(k2,s2): which (dy,dx) this block scatters to, and its first real channel
    int updy = a.updy, updx = a.updx, upc0 = 0;
    if (a.up == 2) {
        const int sub = (cb * CF * 32) / a.upC;
        updy = sub >> 1;
        updx = sub & 1;
        upc0 = sub * a.upC;
    }

    // ---- per-lane pixel operand addressing
    int pixoff[PF];
    int opix[PF];  // output pixel index (elements / out_cs), -1 when masked
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int m = wave * (PF * 32) + pf * 32 + (lane & 31);
        const int npx = a.TH * a.TW;
        bool v = m < npx;
        const int mm = v ? m : npx - 1;
        const int ty = mm / a.TW, tx = mm - ty * a.TW;
        const int oy = oy0 + ty, ox = ox0 + tx;
        v = v && (oy < a.Ho) && (ox < a.Wo);
        pixoff[pf] = ((ty * ST) * a.HWd + tx * ST) * PIXB + (lane >> 5) * 16;
        int op;
        if (a.up) {  // ConvTranspose2d(k2,s2) scatter: ox is a flattened (b,y,x) index
            const int x = ox % a.upW;
            const int t = ox / a.upW;
            const int y = t % a.upH;
            const int bb = t / a.upH;
            op = (bb * 2 * a.upH + 2 * y + updy) * (2 * a.upW) + 2 * x + updx;
        } else {
            op = (b * a.Ho + oy) * a.Wo + ox;
        }
        opix[pf] = v ? op : -1;
    }

    BiasRegs<CF> bz;
    load_bias<CF>(a, cb, upc0, lane, bz);

    f32x16_t acc[CF][PF];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cf][pf][r] = 0.f;

    auto issue_w = [&](int step, int buf) {
        const int chunk = step / NT, tap = step - chunk * NT;
#pragma unroll
        for (int j = 0; j < NWJ; ++j) {
            const int p = wave + 4 * j;
            if (p < CF * 2) {
                const int cf = p >> 1, ks = p & 1;
                const size_t cfg = (size_t)cb * CF + cf;
                const __half* src = a.wpk + (((cfg * a.nchunk + chunk) * NT + tap) * 2 + ks) * 512 + lane * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ldsW + buf * WIMG + p * 1024),
                                                 16, 0, 0);
            }
        }
    };

    auto load_A = [&](int chunk, uint4 (&regs)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int idx = tid + i * 256;
            const int q = idx & 3;
            if (goff[i] >= 0 && (chunk * 32 + q * 8) < a.Cin)
                v = *reinterpret_cast<const uint4*>(a.in + goff[i] + chunk * 32);
            regs[i] = v;
        }
    };
    auto store_A = [&](const uint4 (&regs)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + i * 256;
            if (idx < npieces) *reinterpret_cast<uint4*>(ldsA + (idx >> 2) * PIXB + (idx & 3) * 16) = regs[i];
        }
    };

    // Weight ring: 3 tap images; the DMA for step s+2 is issued at the top of step s, so it
    // has a whole tap-step of MFMA work to land.  One raw barrier per step:
    //   vmcnt(NWJ)  -> this wave's DMA pieces for step s+1 have landed (only the NWJ pieces of
    //                  step s+2 may still fly; a wave's per-step piece count is constant),
    //   lgkmcnt(0)  -> this wave's LDS reads of step s are done, so the ring slot and (at a
    //                  chunk boundary) the halo buffer may be overwritten after the barrier.
    // The asm "memory" clobber keeps the compiler from moving LDS accesses across it.
    const int nsteps = a.nchunk * NT;
    uint4 areg[NP];
    load_A(0, areg);
    issue_w(0, 0);
    if (nsteps > 1) issue_w(1, 1);
    store_A(areg);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    int step = 0;
    int buf = 0;  // ring slot of `step`
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        const bool more = (chunk + 1) < a.nchunk;
        if (more) load_A(chunk + 1, areg);  // register prefetch of the next halo chunk
#pragma unroll
        for (int tap = 0; tap < NT; ++tap, ++step) {
            const int buf2 = (buf >= 1) ? buf - 1 : 2;  // (buf + 2) % 3
            const bool ahead = (step + 2) < nsteps;
            if (ahead) issue_w(step + 2, buf2);
            const int dy = tap / KS, dx = tap - dy * KS;
            const int tapoff = (dy * a.HWd + dx) * PIXB;
            const char* wb = ldsW + buf * WIMG + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8_t af[CF], bf[PF];
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) af[cf] = *reinterpret_cast<const h8_t*>(wb + (cf * 2 + ks) * 1024);
#pragma unroll
                for (int pf = 0; pf < PF; ++pf)
                    bf[pf] = *reinterpret_cast<const h8_t*>(ldsA + pixoff[pf] + tapoff + ks * 32);
#pragma unroll
                for (int cf = 0; cf < CF; ++cf)
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf)
                        acc[cf][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cf], bf[pf], acc[cf][pf], 0, 0, 0);
            }
            if (ahead) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NWJ) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            buf = (buf == 2) ? 0 : buf + 1;
        }
        if (more) {
            store_A(areg);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }

    // the last step's barrier has been passed by every wave: halo / weight LDS is dead, re-use it
    if (a.epi_lds)
        conv_epilogue_lds<CF, PF, ACT>(a, acc, opix, cb, upc0, lane, wave, bz, smem);
    else
        conv_epilogue<CF, PF, ACT>(a, acc, opix, cb, upc0, lane, bz);
}

// ------------------------------------------------------------------------------------------
// int8 conv (BASELINE configs[4]): int8 x int8 -> int32 on v_mfma_i32_32x32x32_i8, fp32 requantisation in
// the epilogue.  Same structure as conv_mfma_kernel (one block per (tile, cout block), halo chunk staged once and re-used
// by all taps, weights by LDS-DMA through a 3-slot ring) with a chunk of 64 INPUT CHANNELS = 64 B per halo pixel, i.e.
// the LDS images, pitches and fragment addresses are those of the fp16 kernel's 32-channel chunk and every MFMA does
// twice the work.
// The input is either the fp16 activation tensor, quantised on its way into LDS -
//     q = clamp(rne(x * inv), -127, 127),  inv = fp16(127 / fp16(amax))      (exact product, ONE rounding: v_pk_fma_f16
//     with the addend 1536 = 1.5 * 2^10, whose fp16 ulp is 1: the low byte of the result IS the two's-complement q)
// - or an int8 twin written by the producer's epilogue with the same formula (`qin`; halves the fill traffic).
// Operand K order: lane half h of k-step ks holds channels 64*chunk + 32*ks + 16*h + (0..15) for the pixel operand and
// for the weight operand alike (y6_pack_conv_weight_i8), so whatever K index the hardware assigns to (h, byte) is the
// same on both sides.

template <int CF, int PF, int KS, int ST>
__global__ __launch_bounds__(256, (CF * PF <= 4 ? 2 : 1)) void conv_i8_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = KS * KS;
    constexpr int WIMG = CF * 2 * 1024;  // bytes of one tap's weight image [cf][ks][lane][16 B]
    constexpr int MAXHP = HaloCap<KS, ST, PF>::value;
    constexpr int NP = (MAXHP * 4 + 255) / 256;  // 16-byte LDS pieces (16 channels) per thread
    constexpr int NWJ = (CF * 2 + 3) / 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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
    const int tx_i = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_i = t2 % a.tiles_y;
    const int b = t2 / a.tiles_y;
    const int oy0 = ty_i * a.TH, ox0 = tx_i * a.TW;
    const int iy0 = oy0 * ST - KS / 2, ix0 = ox0 * ST - KS / 2;

    char* ldsA = smem;
    char* ldsW = smem + a.ldsA_bytes;

    const bool from_q = a.qin != nullptr;
    const int ics = from_q ? a.qin_cs : a.in_cs, ico = from_q ? a.qin_co : a.in_co;
    const int npieces = a.HH * a.HWd * 4;
    int goff[NP];   // element offset of the piece's first channel in chunk 0 (<0: zero fill)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = tid + i * 256;
        int g = -2;
        if (idx < npieces) {
            const int hp = idx >> 2, q = idx & 3;
            const int hy = hp / a.HWd, hx = hp - hy * a.HWd;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool v = (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
            g = v ? (((b * a.H + iy) * a.W + ix) * ics + ico + q * 16) : -1;
        }
        goff[i] = g;
    }

    int pixoff[PF];
    int opix[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int m = wave * (PF * 32) + pf * 32 + (lane & 31);
        const int npx = a.TH * a.TW;
        bool v = m < npx;
        const int mm = v ? m : npx - 1;
        const int ty = mm / a.TW, tx = mm - ty * a.TW;
        const int oy = oy0 + ty, ox = ox0 + tx;
        v = v && (oy < a.Ho) && (ox < a.Wo);
        pixoff[pf] = ((ty * ST) * a.HWd + tx * ST) * PIXB + (lane >> 5) * 16;
        opix[pf] = v ? (b * a.Ho + oy) * a.Wo + ox : -1;
    }

    i32x16_t acc[CF][PF];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cf][pf][r] = 0;

    const char* wbase = reinterpret_cast<const char*>(a.wpk);
    auto issue_w = [&](int step, int buf) {
        const int chunk = step / NT, tap = step - chunk * NT;
#pragma unroll
        for (int j = 0; j < NWJ; ++j) {
            const int p = wave + 4 * j;
            if (p < CF * 2) {
                const int cf = p >> 1, ks = p & 1;
                const size_t cfg = (size_t)cb * CF + cf;
                const char* src = wbase + (((cfg * a.nchunk + chunk) * NT + tap) * 2 + ks) * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ldsW + buf * WIMG + p * 1024),
                                                 16, 0, 0);
            }
        }
    };

    // fp16 source: two 16-byte loads (16 channels) per LDS piece, quantised when they are written to LDS
    // int8 source: one 16-byte load per piece
    auto load_A = [&](int chunk, uint4 (&lo)[NP], uint4 (&hi)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            uint4 v0 = make_uint4(0u, 0u, 0u, 0u), v1 = v0;
            const int c0 = chunk * 64 + ((tid + i * 256) & 3) * 16;
            if (goff[i] >= 0 && c0 < a.Cin) {
                if (from_q) {
                    v0 = *reinterpret_cast<const uint4*>(a.qin + goff[i] + chunk * 64);   // Cin % 16 == 0 on this path
                } else {
                    v0 = *reinterpret_cast<const uint4*>(a.in + goff[i] + chunk * 64);
                    if (c0 + 8 < a.Cin) v1 = *reinterpret_cast<const uint4*>(a.in + goff[i] + chunk * 64 + 8);
                }
            }
            lo[i] = v0;
            hi[i] = v1;
        }
    };
    auto store_A = [&](const uint4 (&lo)[NP], const uint4 (&hi)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + i * 256;
            if (idx < npieces)
                *reinterpret_cast<uint4*>(ldsA + (idx >> 2) * PIXB + (idx & 3) * 16) =
                    from_q ? lo[i] : q8_piece(lo[i], hi[i], a.q_inv2, a.q_lo2, a.q_hi2);
        }
    };

    const int nsteps = a.nchunk * NT;
    uint4 alo[NP], ahi[NP];
    load_A(0, alo, ahi);
    issue_w(0, 0);
    if (nsteps > 1) issue_w(1, 1);
    store_A(alo, ahi);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    int step = 0;
    int buf = 0;
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        const bool more = (chunk + 1) < a.nchunk;
        if (more) load_A(chunk + 1, alo, ahi);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap, ++step) {
            const int buf2 = (buf >= 1) ? buf - 1 : 2;
            const bool ahead = (step + 2) < nsteps;
            if (ahead) issue_w(step + 2, buf2);
            const int dy = tap / KS, dx = tap - dy * KS;
            const int tapoff = (dy * a.HWd + dx) * PIXB;
            const char* wb = ldsW + buf * WIMG + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                i32x4_t af[CF], bf[PF];
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) af[cf] = *reinterpret_cast<const i32x4_t*>(wb + (cf * 2 + ks) * 1024);
#pragma unroll
                for (int pf = 0; pf < PF; ++pf)
                    bf[pf] = *reinterpret_cast<const i32x4_t*>(ldsA + pixoff[pf] + tapoff + ks * 32);
#pragma unroll
                for (int cf = 0; cf < CF; ++cf)
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf)
                        acc[cf][pf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[cf], bf[pf], acc[cf][pf], 0, 0, 0);
            }
            if (ahead) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NWJ) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            buf = (buf == 2) ? 0 : buf + 1;
        }
        if (more) {
            store_A(alo, ahi);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }

    // ---- epilogue, one cout fragment at a time (16 bias + 16 scale registers live next to the accumulators)
#pragma unroll
    for (int cf = 0; cf < CF; ++cf) {
        BiasRegs<1> bz, qs;
        load_bias<1>(a, cb * CF + cf, 0, lane, bz);
        {
            ConvKArgs t = a;
            t.bias = a.qscale;
            load_bias<1>(t, cb * CF + cf, 0, lane, qs);
        }
        conv_i8_epilogue<PF>(a, acc[cf], opix, cb * CF + cf, lane, bz.v[0], qs.v[0]);
    }
}

// ---------------------------------------------------------------------------------------------
// v3: pipelined persistent kernel (3x3 stride 1).  tools/mfma_ceiling.hip shows what LDS-fed MFMA loops
// sustain on this box (1.4-1.9 PFLOP/s); the s_memtime traces of the kernels above show where they lose
// it: the fill of the next chunk (global loads issued back-to-back saturate the 64 B/clk TA path and
// block the wave at issue; the publish sits between two barriers).  Here
//   * a chunk is 16 input channels (one MFMA k-step per tap) so that two LDS buffers of halo + weights
//     fit twice per CU (two blocks -> two waves per SIMD);
//   * the next chunk's global loads are issued one or two per tap during taps 0-3, its LDS publishes
//     during taps 5-8 into the OTHER buffer, and a single barrier per chunk closes it;
//   * work items (tile, cout block) are walked persistently, so the fill of the next item's first chunk
//     overlaps the last chunk of the current one, and the epilogue overlaps the co-resident block.
// Pixel pitch is 48 bytes (32 data + 16 pad): an odd number of 16-byte slots keeps a 32-pixel fragment
// read conflict-free, as with the 80-byte pitch above.
// ---------------------------------------------------------------------------------------------
constexpr int PIXP = 48;
// Ceiling probes of the pipe kernel (WRONG RESULTS, timing only): rebuild with -DY6_PIPE_PROBE=n.
//   1 = skip the pixel-fragment reads of taps kx=1,2; 2 = skip the weight-fragment reads of taps > 0;
//   3 = no fill (global loads + LDS publishes); 4 = no chunk barrier; 5 = no MFMAs; 6 = no epilogue stores.
// Compile-time on purpose: as RUNTIME flags the uniform branches around the staged loads made hipcc fall back
// to conservative waitcnts and cost the production kernel 35 % (r15).
#ifndef Y6_PIPE_PROBE
#define Y6_PIPE_PROBE 0
#endif
constexpr int kPipeProbe = Y6_PIPE_PROBE;

template <int BP>
struct PipeHaloCap {   // halo pixels (3x3 stride 1) of the largest tile shape offered for BP output pixels
    static constexpr int value = BP <= 128 ? 208 : (BP <= 256 ? 352 : (BP <= 512 ? 660 : 1190));
};

// s_memtime trace of block 0 / thread 0 (tools/conv_trace.py, env Y6_CONV_TRACE): the kernel declares `tracing` and `dbg_n`
#define Y6_TRACE(tag)                                                        \
    do {                                                                     \
        if (tracing && dbg_n < 256) {                                        \
            a.dbg[2 * dbg_n] = __builtin_amdgcn_s_memtime();                 \
            a.dbg[2 * dbg_n + 1] = (unsigned long long)(tag);                \
            ++dbg_n;                                                         \
        }                                                                    \
    } while (0)

// NW waves per block: 4 (two blocks per CU) or 8 (one block per CU, twice the pixels sharing one weight image)
// ST = 2 (stride-2 3x3): the halo is stored with its even and odd columns de-interleaved -
//   slot(hy, hx) = (2*hy + (hx & 1)) * HWp + (hx >> 1),  HWp = (HWd + 1) / 2
// so that the 32 output pixels of a fragment (input columns 2*tx + kx) read 32 CONSECUTIVE slots for every tap,
// conflict-free like stride 1 (read in place they would stride by 96 bytes = 6 slots: 2-way conflicts and more).
template <int CF, int PF, int WPS, int NW, int ST = 1, int DEPTH = 2>
__global__ __launch_bounds__(NW * 64, WPS) void conv_mfma_pipe_kernel(const ConvKArgs a) {
    constexpr int NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 9;
    constexpr int WQ = CF * NT * 64;             // 16-byte units of one 16-channel weight image (CF x 9 KiB)
    constexpr int NWR = (WQ + NTHR - 1) / NTHR;
    constexpr int MAXHP = ST == 1 ? PipeHaloCap<NW * PF * 32>::value : 1280;   // stride 2: 256-pixel tiles only (9x130 slots at most)
    constexpr int NP = (MAXHP * 2 + NTHR - 1) / NTHR;  // two 16-byte pieces per halo pixel
    constexpr int NL = NP + NWR;                 // staged 16-byte pieces per thread per chunk
    constexpr int LPT = (NL + 3) / 4;            // pieces published + re-requested per tap in taps 0-3

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ldsA = smem;                            // 2 x ldsA_bytes
    char* ldsW = smem + 2 * a.ldsA_bytes;         // 2 x WQ*16
    float* ldsBias = reinterpret_cast<float*>(ldsW + 2 * WQ * 16);
    char* ldsDump = reinterpret_cast<char*>(ldsBias + 2 * CF * 32);   // 16 bytes: where the staging threads without a piece publish
    const int nids = a.nids;
    const int gstride = gridDim.x;
    const int nch = (a.Cin + 15) >> 4;

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
    {
        int t, c;
        decode(id, t, c);
        if (t >= a.ntiles) id = next_valid(id);
    }
    if (id >= nids) return;

    const int npieces = a.HH * a.HWd * 2;
    int goff[NP];
    auto setup_goff = [&](int item) {
        int tile, cbx;
        decode(item, tile, cbx);
        const int tx_i = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty_i = t2 % a.tiles_y;
        const int b = t2 / a.tiles_y;
        const int iy0 = ty_i * a.TH * ST - 1, ix0 = tx_i * a.TW * ST - 1;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + i * NTHR;
            int g = -1;
            if (idx < npieces) {
                const int hp = idx >> 1, q = idx & 1;
                const int hy = hp / a.HWd, hx = hp - hy * a.HWd;
                const int iy = iy0 + hy, ix = ix0 + hx;
                const bool v = (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
                g = v ? (((b * a.H + iy) * a.W + ix) * a.in_cs + a.in_co + q * 8) : -1;
            }
            goff[i] = g;
        }
    };
    // per-thread constant part of the packed-weight address of staged piece j:
    // [cout/32][cin/32][tap][kstep=2][lane][8]  ->  cf, tap, lane of this thread's piece
    int woff[NWR];
#pragma unroll
    for (int j = 0; j < NWR; ++j) {
        const int q = tid + j * NTHR;
        const int cf = q / (NT * 64), r = q - cf * (NT * 64);
        woff[j] = (q < WQ) ? (cf * a.nchunk * NT * 2 * 64 + (r >> 6) * 2 * 64 + (r & 63)) * 8 : 0;   // pad threads re-read piece 0
    }
    // LDS offsets of this lane's pixels (needed all through the chunk loop) and, separately, their output
    // pixel indices (needed by the epilogue only: computed there so they are not live across the loop)
    int pixoff[PF], cb = 0;
    auto setup_pix = [&](int item) {
        int tile;
        decode(item, tile, cb);
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) {
            const int m = wave * (PF * 32) + pf * 32 + (lane & 31);
            const int npx = a.TH * a.TW;
            const int mm = m < npx ? m : npx - 1;
            const int ty = mm / a.TW, tx = mm - ty * a.TW;
            const int HWp = (a.HWd + 1) >> 1;
            pixoff[pf] = (ST == 1 ? (ty * a.HWd + tx) : (4 * ty * HWp + tx)) * PIXP + (lane >> 5) * 16;
        }
    };
    auto out_pix = [&](int item, int (&opix)[PF]) {
        int tile, cbx;
        decode(item, tile, cbx);
        const int tx_i = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty_i = t2 % a.tiles_y;
        const int b = t2 / a.tiles_y;
        const int oy0 = ty_i * a.TH, ox0 = tx_i * a.TW;
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) {
            const int m = wave * (PF * 32) + pf * 32 + (lane & 31);
            const int npx = a.TH * a.TW;
            bool v = m < npx;
            const int mm = v ? m : npx - 1;
            const int ty = mm / a.TW, tx = mm - ty * a.TW;
            const int oy = oy0 + ty, ox = ox0 + tx;
            v = v && (oy < a.Ho) && (ox < a.Wo);
            opix[pf] = v ? (b * a.Ho + oy) * a.Wo + ox : -1;
        }
    };

    // Staging cursor: the (item, chunk) whose data is being requested from global memory.  It runs DEPTH
    // chunks ahead of the chunk being multiplied.  DEPTH = 2: a piece is requested during chunk c, published
    // to the free LDS buffer during chunk c+1 (a whole chunk of MFMAs later) and consumed in chunk c+2.
    // DEPTH = 3: the HALO pieces (HBM / MALL latency) are requested during chunk c, held in registers through
    // chunk c+1, published during chunk c+2 and consumed in chunk c+3 - two register sets alternate and the
    // load has TWO chunk periods to arrive; the weight pieces (always L2 hits) stay at depth 2.
    // (ceiling probes, DESIGN.md §6: with one period the fill side alone runs at the full kernel's speed).
    uint4 stg0[NL];                        // halo pieces [0, NP) + weight pieces [NP, NL)
    uint4 stg1[DEPTH == 3 ? NP : 1];       // second halo set
    unsigned ok0 = 0, ok1 = 0;
    size_t w_wbase = 0;                    // DEPTH 3: weight image of the chunk ONE behind the cursor
    int s_item = id, s_chunk = 0, s_wcb = 0;
    int nx_cin0 = 0;        // first input channel of the staged chunk
    size_t nx_wbase = 0;    // element offset of its weight image for cf = 0
    auto stage_addr = [&]() {
        nx_cin0 = s_chunk * 16;
        nx_wbase = (((size_t)s_wcb * CF * a.nchunk + (s_chunk >> 1)) * NT * 2 + (s_chunk & 1)) * 64 * 8;
    };
    auto stage_advance = [&]() {
        w_wbase = nx_wbase;
        if (s_item >= nids) return;   // end of this block's stream: keep re-requesting the last (valid) chunk
        if (s_chunk + 1 < nch) {
            ++s_chunk;
        } else {
            const int n = next_valid(s_item);
            if (n >= nids) {
                s_item = n;
                return;
            }
            s_item = n;
            s_chunk = 0;
            int t;
            decode(s_item, t, s_wcb);
            setup_goff(s_item);   // the previous item's table is dead: all its chunks have been requested
        }
        stage_addr();
    };
    // Every staged load and publish is UNPREDICATED: out-of-image / out-of-channel halo pieces read the tensor
    // base and are zeroed at publish time through a per-thread flag word, pad threads re-read weight piece 0
    // into the buffers' pad area.  With exec-masked branches around them hipcc's waitcnt insertion falls
    // back to vmcnt(0) before every ds_write, i.e. the full load latency four times per chunk.
    auto load_piece = [&](int k, uint4* stg, unsigned& stg_ok, size_t wbase) {
        if (k < NP) {
            const int q = (tid + k * NTHR) & 1;
            const bool ok = goff[k] >= 0 && (nx_cin0 + q * 8) < a.Cin;
            stg[k] = *reinterpret_cast<const uint4*>(a.in + (ok ? goff[k] + nx_cin0 : 0));
            stg_ok = ok ? (stg_ok | (1u << k)) : (stg_ok & ~(1u << k));
        } else {
            stg[k] = *reinterpret_cast<const uint4*>(a.wpk + wbase + woff[k - NP]);
        }
    };
    auto store_piece = [&](int k, int buf, const uint4* stg, unsigned stg_ok) {
        if (k < NP) {
            const int idx = tid + k * NTHR;
            uint4 v = stg[k];
            if (!((stg_ok >> k) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
            int slot = idx >> 1;
            if (ST == 2) {
                const int hy = slot / a.HWd, hx = slot - hy * a.HWd;
                slot = (2 * hy + (hx & 1)) * ((a.HWd + 1) >> 1) + (hx >> 1);
            }
            char* dst = ldsA + buf * a.ldsA_bytes + slot * PIXP + (idx & 1) * 16;
            *reinterpret_cast<uint4*>(idx < npieces ? dst : ldsDump) = v;   // address select, not a branch
        } else {
            const int q = tid + (k - NP) * NTHR;
            char* dst = ldsW + (buf * WQ + q) * 16;
            *reinterpret_cast<uint4*>(q < WQ ? dst : ldsDump) = stg[k];
        }
    };

    int dbg_n = 0;
    const bool tracing = a.dbg != nullptr && blockIdx.x == 0 && tid == 0;
    Y6_TRACE(1);
    {
        int t;
        decode(id, t, s_wcb);
    }
    setup_goff(id);
    setup_pix(id);
    stage_addr();
#pragma unroll
    for (int k = 0; k < NL; ++k) load_piece(k, stg0, ok0, nx_wbase);
#pragma unroll
    for (int k = 0; k < NL; ++k) store_piece(k, 0, stg0, ok0);
    stage_advance();          // second chunk of the stream stays in registers until the first chunk's taps
    if (DEPTH == 3) {
#pragma unroll
        for (int k = 0; k < NP; ++k) load_piece(k, stg1, ok1, 0);
#pragma unroll
        for (int k = NP; k < NL; ++k) load_piece(k, stg0, ok0, nx_wbase);
        stage_advance();      // ... and the halo of the third one too
#pragma unroll
        for (int k = 0; k < NP; ++k) load_piece(k, stg0, ok0, 0);
    } else {
#pragma unroll
        for (int k = 0; k < NL; ++k) load_piece(k, stg0, ok0, nx_wbase);
    }
    __syncthreads();
    Y6_TRACE(2);

    int pb = 0, item_parity = 0;
    while (true) {
        f32x16_t acc[CF][PF];
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cf][pf][r] = 0.f;
        float* lbias = ldsBias + (item_parity ? CF * 32 : 0);
        if (tid < CF * 32) {
            const int c = cb * CF * 32 + tid;
            lbias[tid] = (a.bias != nullptr && c < a.Cout) ? a.bias[c] : 0.f;
        }
        const int nid = next_valid(id);
        bool synced = false;
        auto do_chunk = [&](int chunk, uint4* astg, unsigned& astg_ok) __attribute__((always_inline)) {
            const bool last = (chunk + 1) == nch;
            const bool have_next = !last || nid < nids;   // registers hold the chunk after this one
            if (have_next) stage_advance();               // ... and the one after that gets requested now
            Y6_TRACE(10);
            const char* Ab = ldsA + pb * a.ldsA_bytes;
            const char* Wb = ldsW + pb * (WQ * 16) + lane * 16;
            // A fragments (pixels) are requested one tap ahead.  The weight fragments too, unless the
            // accumulators already take 128 registers (CF*PF = 8): then they are requested at the top of
            // their own tap and the co-resident wave covers the LDS latency.
            constexpr int WST = (CF * PF >= 8) ? 1 : 2;
            h8_t fa[WST][CF], fb[2][PF];
            auto ldfragW = [&](int t, int buf) {
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) fa[buf][cf] = *reinterpret_cast<const h8_t*>(Wb + (cf * NT + t) * 1024);
            };
            auto ldfragA = [&](int t, int buf) {
                const int tapoff = (ST == 1 ? ((t / 3) * a.HWd + (t % 3))
                                            : ((2 * (t / 3) + ((t % 3) & 1)) * ((a.HWd + 1) >> 1) + ((t % 3) >> 1))) * PIXP;
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) fb[buf][pf] = *reinterpret_cast<const h8_t*>(Ab + pixoff[pf] + tapoff);
            };
            auto taps = [&](uint4* astg, unsigned& astg_ok) __attribute__((always_inline)) {   // astg: the halo set holding the chunk after this one
                ldfragA(0, 0);
                if (WST == 2) ldfragW(0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (WST == 1) ldfragW(t, 0);
                    if (t + 1 < NT) {
                        if (!(kPipeProbe == 1 && ((t + 1) % 3) != 0)) ldfragA(t + 1, (t + 1) & 1);
                        if (WST == 2 && kPipeProbe != 2) ldfragW(t + 1, (t + 1) & 1);
                    }
                    if (t < 4 && kPipeProbe != 3) {   // publish the oldest staged chunk, then reuse its registers (no
                                                           // branches: past the end of the stream this republishes /
                                                           // re-requests the last chunk)
#pragma unroll
                        for (int u = 0; u < LPT; ++u) {
                            const int kk = t * LPT + u;
                            if (kk < NL) store_piece(kk, pb ^ 1, kk < NP ? astg : stg0, kk < NP ? astg_ok : ok0);
                        }
#pragma unroll
                        for (int u = 0; u < LPT; ++u) {
                            const int kk = t * LPT + u;
                            if (kk < NP) load_piece(kk, astg, astg_ok, 0);
                            else if (kk < NL) load_piece(kk, stg0, ok0, DEPTH == 3 ? w_wbase : nx_wbase);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads of tap t+1 AHEAD of tap t's MFMAs
                    if (kPipeProbe != 5) {
#pragma unroll
                        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
                            for (int pf = 0; pf < PF; ++pf)
                                acc[cf][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[WST == 2 ? (t & 1) : 0][cf], fb[t & 1][pf], acc[cf][pf], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            taps(astg, astg_ok);
            Y6_TRACE(11);
            if (have_next) {
                if (kPipeProbe != 4) __syncthreads();   // next chunk published; everyone is done with this one
                synced = true;
                pb ^= 1;
            }
            Y6_TRACE(15);
        };
        if (DEPTH == 3) {
            // two chunks per trip so that the halo register sets alternate statically (nch is even for these
            // variants - y6_conv_mfma_supports): the chunk after an even chunk waits in set 1, after an odd one in set 0
            for (int chunk = 0; chunk < nch; chunk += 2) {
                do_chunk(chunk, stg1, ok1);
                do_chunk(chunk + 1, stg0, ok0);
            }
        } else {
            for (int chunk = 0; chunk < nch; ++chunk) do_chunk(chunk, stg0, ok0);
        }
        if (!synced) __syncthreads();   // no chunk barrier has published lbias yet
        int opix[PF];
        out_pix(id, opix);
        // one cout fragment at a time: 16 bias registers live instead of 16*CF next to 16*CF*PF accumulators
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) {
            BiasRegs<1> bz;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(lbias + cf * 32 + 8 * g + 4 * (lane >> 5));
                bz.v[0][g * 4 + 0] = t.x;
                bz.v[0][g * 4 + 1] = t.y;
                bz.v[0][g * 4 + 2] = t.z;
                bz.v[0][g * 4 + 3] = t.w;
            }
            if (kPipeProbe != 6)
                conv_epilogue<1, PF>(a, *reinterpret_cast<const f32x16_t(*)[1][PF]>(&acc[cf]), opix, cb * CF + cf, 0, lane, bz);
        }
        Y6_TRACE(20);
        if (nid >= nids) break;
        id = nid;
        item_parity ^= 1;
        setup_pix(id);
    }
}

// ---------------------------------------------------------------------------------------------
// 1x1 conv, streaming form.  A 1x1 conv is a GEMM [pixels x Cin] x [Cin x Cout] with no reuse of a pixel
// between waves, so nothing needs LDS: the lane layout of the MFMA pixel operand (pixel = lane & 31,
// channels (lane >> 5) * 8 .. +8 of a 16-channel k-step) IS a 16-byte piece of the NHWC row, loaded straight
// from global memory; the weights of the block's cout fragments (CF x Cin/16 fragments) are loaded ONCE
// into registers.  A wave then streams 32-pixel fragments: the loads of the next fragment are in flight
// while the current one is multiplied and stored.  No LDS, no barrier - the kernel moves bytes at the speed
// the memory system delivers them, which is what these layers (64-256 FLOP/B) need.
// Blocks are persistent over pixel fragments; blocks that share pixels (different cout blocks) sit on the
// same XCD.  Requires Cin = 16 * KS exactly, KS in {4, 8, 16}.
// ---------------------------------------------------------------------------------------------
template <int CF, int KS>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const ConvKArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.x;
    const int cb = (b >> 3) % a.ncb;                       // consecutive groups of 8 blocks: one per XCD, same cout block
    const int j = (b / (8 * a.ncb)) * 8 + (b & 7);         // pixel-stream index of this block among its cout block's
    const int nstream = (gridDim.x / a.ncb) * 4;           // waves sharing the pixel fragments of one cout block
    const int npix = a.W;                                  // flattened B*H*W (build_launch)
    const int nfrag = (npix + 31) >> 5;

    h8_t w[CF][KS];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            w[cf][ks] = *reinterpret_cast<const h8_t*>(
                a.wpk + ((((size_t)(cb * CF + cf) * a.nchunk + (ks >> 1)) * 2 + (ks & 1)) * 64 + lane) * 8);

    BiasRegs<CF> bzall;   // resident: a bias LOAD inside the loop would wait (in-order vmcnt) for the pixel prefetch too
    load_bias<CF>(a, cb, 0, lane, bzall);

    auto load_px = [&](int f, h8_t (&r)[KS]) {
        int px = f * 32 + (lane & 31);
        px = px < npix ? px : npix - 1;                    // clamped rows are computed and dropped by the epilogue
        const __half* p = a.in + (size_t)px * a.in_cs + a.in_co + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) r[ks] = *reinterpret_cast<const h8_t*>(p + ks * 16);
    };
    auto compute_store = [&](int f, const h8_t (&r)[KS]) {
        f32x16_t acc[CF][1];
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[cf][0][q] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int cf = 0; cf < CF; ++cf)
                acc[cf][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cf][ks], r[ks], acc[cf][0], 0, 0, 0);
        const int px = f * 32 + (lane & 31);
        int opix[1] = {px < npix ? px : -1};
        // one cout fragment at a time: keeps the epilogue's live set small
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) {
            BiasRegs<1> bz;
#pragma unroll
            for (int q = 0; q < 16; ++q) bz.v[0][q] = bzall.v[cf][q];
            conv_epilogue<1, 1>(a, *reinterpret_cast<const f32x16_t(*)[1][1]>(&acc[cf]), opix, cb * CF + cf, 0, lane, bz);
        }
    };

    h8_t r0[KS], r1[KS];
    int f = j * 4 + wave;
    if (f < nfrag) load_px(f, r0);
    while (f < nfrag) {
        const int f1 = f + nstream;
        if (f1 < nfrag) load_px(f1, r1);                   // in flight during this fragment's MFMAs and stores
        compute_store(f, r0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) r0[ks] = r1[ks];   // waits for the prefetch exactly where the next MFMAs would
        f = f1;
    }
}

// The kernel forms a conv op can run on (index = the `variant` of y6_conv_desc; NAMES are the stable handle: autotune caches,
// the default candidate set and the tools use them).  Round 4 retired the forms no committed autotune log picked since round 2
// (the chunk-granular persistent kernel pers_*, the 8-wave / 3-stage / 4-fragment pipe forms, dma8_c2p2, the 32-channel-chunk
// dmaw* forms, dmar8_c2p2, dma_c2p4, the 64-cout register-fed form): git history has them.
//   naive                      conv_misc.hip, cross-check only
//   mfma_c{1,2,4}p{1,2}        one block per (tile, cout block), barrier per tap: 1x1, 3x3 stride 2, convT scatter, odd shapes
//   pipe_c2p2 / c2p1 / c1p2    round 1's persistent 3x3 stride-1 kernel (residual / ragged-channel / narrow layers: the
//                              data-gradient convs of the first stages have 32 couts)
//   stream1x1_c1 / _c2         streaming 1x1 (Cin 64 / 128 / 256)
//   dma_c2p2 / c2p1 / c1p2, dma8_c4p1, dmarw8_c2p2 (resident weights, Cin <= 64), dmas2_c2p1 / dma8s2_c2p1 / dma8s2_c4p1
//                              LDS-DMA fed 3x3 kernels (conv_dma.hip); depth = LDS stages, hc = channels per chunk, cs = stride
//   wreg_p{4,5,6,7}, wregs2_p{3,4}   weights through registers, halo in 32-channel LDS stages (conv_wreg.hip), stride 1 / 2:
//                              cf = waves along the couts (32 each), pf = pixel fragments per wave, two 4-wave blocks per CU
//   pw_c4p2 / pw_c2p2          1x1 stride 1 with the whole reduction in LDS (conv_pw.hip)
const VariantCfg kVariants[] = {
    {0, 0, 0, "naive"},     {1, 1, 0, "mfma_c1p1"}, {2, 1, 0, "mfma_c2p1"}, {4, 1, 0, "mfma_c4p1"},
    {1, 2, 0, "mfma_c1p2"}, {2, 2, 0, "mfma_c2p2"}, {4, 2, 0, "mfma_c4p2"},
    {2, 2, 2, "pipe_c2p2"}, {2, 1, 2, "pipe_c2p1"}, {1, 2, 2, "pipe_c1p2"},
    {1, 1, 3, "stream1x1_c1"}, {2, 1, 3, "stream1x1_c2"},
    {2, 2, 4, "dma_c2p2", 4}, {2, 1, 4, "dma_c2p1", 4}, {1, 2, 4, "dma_c1p2", 4},
    {2, 1, 4, "dmas2_c2p1", 4, 1, 2, 16, 2}, {2, 1, 4, "dma8s2_c2p1", 8, 1, 2, 16, 2},
    {4, 1, 4, "dma8_c4p1", 8},
    {2, 2, 4, "dmarw8_c2p2", 8, 1, 2, 32, 1, 1},
    {4, 1, 4, "dma8s2_c4p1", 8, 1, 2, 16, 2},
    // tile geometries of the int8 LDS-DMA kernels (y6_conv_i8 variants 7 / 9; variant 8 shares dma_c2p2's): 512 pixels x 64 couts on
    // eight waves with 16- / 32-channel chunks.  Their fp16 forms lost everywhere and are no longer built.
    {2, 2, 4, "i8_dma8_c2p2", 8, 1, 2, 16, 1, 0, 1}, {2, 2, 4, "i8_dmaw8_c2p2", 8, 1, 2, 32, 1, 0, 1},
    {4, 6, 6, "wreg_p6", 4}, {4, 7, 6, "wreg_p7", 4}, {4, 4, 6, "wreg_p4", 4}, {4, 5, 6, "wreg_p5", 4},
    {4, 3, 6, "wregs2_p3", 4, 1, 2, 16, 2}, {4, 4, 6, "wregs2_p4", 4, 1, 2, 16, 2},
    // int8 only (y6_conv_i8 variant 13): 64-cout blocks of the register-fed stride-2 kernel - two cout waves x two pixel waves,
    // 128 pixel slots
    {2, 2, 6, "i8_wreg2s2_p2", 4, 1, 2, 16, 2, 0, 1},
    // conv_pw.hip (round 6): 1x1 stride 1, the whole reduction of a 64- / 128-pixel tile requested at once; cf = cout waves, pf = pixel
    // fragments per wave: 128 couts x 64 pixels / 64 couts x 128 pixels per block
    {4, 2, 7, "pw_c4p2", 4}, {2, 2, 7, "pw_c2p2", 4}};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static int variant_index(const char* name) {
    for (int i = 0; i < kNumVariants; ++i)
        if (strcmp(kVariants[i].name, name) == 0) return i;
    return -1;
}

int halo_cap(int ks, int st, int pf) {
    if (ks == 1) return pf * 128;
    if (st == 1) return pf == 2 ? 352 : 208;
    return 576;
}

// choose the spatial tile (TH x TW outputs) for a block of `bp` pixel slots
void choose_tile(int Ho, int Wo, int ks, int st, int bp, int cap, int* pTH, int* pTW, int tw_mult = 0) {
    double best = -1.0;
    int bTH = 1, bTW = 1;
    const int maxTW = Wo < bp ? Wo : bp;
    for (int TW = 1; TW <= maxTW; ++TW) {
        int TH = bp / TW;
        if (TH > Ho) TH = Ho;
        for (; TH >= 1; --TH) {
            const int HH = (TH - 1) * st + ks, HW = (TW - 1) * st + ks;
            if (HH * HW <= cap) break;
        }
        if (TH < 1) continue;
        const double tiles = (double)y6_cdiv(Ho, TH) * y6_cdiv(Wo, TW);
        const int HH = (TH - 1) * st + ks, HW = (TW - 1) * st + ks;
        // useful MFMA work / issued MFMA work, lightly penalised by halo staging volume
        const double eff = ((double)Ho * Wo) / (tiles * bp);
        const double halo = (double)(TH * st) * (TW * st) / ((double)HH * HW);
        // a 32-lane MFMA fragment that stays on one tile row reads 32 consecutive halo pixels: with the
        // 80-byte pixel pitch that is bank-conflict free; a fragment split over two rows is 2-way.  (Preferring
        // such tiles cut the conflict cycles from 0.41 to 0.15 per access and did not move the time - round 1,
        // profiles/r01 - so the per-tap kernels take the tile with the best fill.)
        // tw_mult (LDS-DMA kernels): widths that keep the 16-lane read groups on one halo row are conflict-free
        const double rowfit = tw_mult ? (TW % tw_mult == 0 ? 1.0 : 0.95) : 1.0;
        const double score = eff * (0.85 + 0.15 * halo) * rowfit;
        if (score > best + 1e-9) {
            best = score;
            bTH = TH;
            bTW = TW;
        }
    }
    *pTH = bTH;
    *pTW = bTW;
}


// conv_wreg.hip: halo row pitch for a tile width (a read group that wraps to the next tile row must continue at a pixel index
// that is consecutive modulo 16), 1 KiB requests per stage image, and the tile itself
// (The dense pitch TW + 2: a read group that wraps to the next tile row has two 2-way conflicts, one extra LDS cycle on some reads.
// The conflict-free alternative - TW + 16 for widths that are not multiples of 16 - was measured in round 4 and lost: the stage
// image is 25-40 % larger and the halo REQUESTS are what the stage top costs, 470 cycles each with ten per wave in flight,
// profiles/r04/v0_trace_wreg_*.txt.)
int wreg_row_pitch(int TW, int st) {
    if (st == 2) return 2 * TW + 2;   // TW + 1 even columns, TW odd ones, one pad slot
    return TW + 2;
}
int wreg_pieces(int TH, int TW, int st) { return y6_cdiv(5 * ((TH - 1) * st + 3) * wreg_row_pitch(TW, st), 64); }
// TH x TW <= bp pixel slots: fewest rounds of the persistent walk (items / resident blocks, rounded up) first - a block's time is
// its pixel SLOTS, so a tile that divides the map with a few idle slots beats a full tile that leaves a partial last round -
// then the fewest items (less padding inside the rounds), then the smallest stage image
void choose_tile_wreg(int B, int Ho, int Wo, int ncb, int bp, int max_pieces, int slots, int st, int* pTH, int* pTW) {
    long best_rounds = -1, best_items = 0;
    int best_halo = 0, bTH = 1, bTW = 1;
    for (int TW = 1; TW <= (Wo < bp ? Wo : bp); ++TW) {
        int TH = bp / TW;
        if (TH > Ho) TH = Ho;
        while (TH >= 1 && wreg_pieces(TH, TW, st) > max_pieces) --TH;
        if (TH < 1) continue;
        // (a smaller TH with the same row count per map never hurts the rounds: take the smallest that keeps the tile count)
        const int ty = y6_cdiv(Ho, TH);
        TH = y6_cdiv(Ho, ty);
        const long items = (long)B * ty * y6_cdiv(Wo, TW) * ncb;
        const int halo = wreg_pieces(TH, TW, st);
        // two blocks per CU need both stage images of both blocks in 160 KiB: a bigger image halves the resident blocks
        const int sl = (2 * halo * 1024 + 2048 > 80 * 1024) ? slots / 2 : slots;
        const long rounds = (items + sl - 1) / sl * (sl == slots ? 1 : 2);   // (a round of half the blocks counts double)
        if (best_rounds < 0 || rounds < best_rounds || (rounds == best_rounds && (items < best_items || (items == best_items && halo < best_halo)))) {
            best_rounds = rounds;
            best_items = items;
            best_halo = halo;
            bTH = TH;
            bTW = TW;
        }
    }
    *pTH = bTH;
    *pTW = bTW;
}

int device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

int build_launch(const y6_conv_desc* d, int variant, int up, int updy, int updx, Launch* L) {
    const VariantCfg& vc = kVariants[variant];
    const int ks = d->ksize, st = d->stride;
    ConvKArgs& k = L->k;
    memset(&k, 0, sizeof(k));
    k.in = (const __half*)d->in.data;
    k.out = (__half*)d->out.data;
    k.wpk = (const __half*)d->w_packed;
    k.bias = d->bias;
    k.pscale = d->post_scale;
    k.pshift = d->post_shift;
    k.res = (const __half*)d->res.data;
    k.res_alpha = d->res_alpha;
    k.Cin = d->in.C;
    k.Cout = d->out.C;
    k.in_cs = d->in.cstride;
    k.in_co = d->in.coff;
    k.out_cs = d->out.cstride;
    k.out_co = d->out.coff;
    k.res_cs = d->res.cstride;
    k.res_co = d->res.coff;
    k.act = d->act;
    k.vec_ok = (d->out.cstride % 4 == 0) && (d->out.coff % 4 == 0) && (((uintptr_t)d->out.data & 7) == 0);
    k.vec16_ok = (d->out.cstride % 8 == 0) && (d->out.coff % 8 == 0) && (((uintptr_t)d->out.data & 15) == 0);
    k.res_vec = d->res.data != nullptr && (d->res.cstride % 4 == 0) && (d->res.coff % 4 == 0) && (((uintptr_t)d->res.data & 7) == 0);
    {
        static const bool no_epi_lds = getenv("Y6_CONV_NO_EPI_LDS") != nullptr;   // A/B switch for profiling
        const int cend = up == 2 ? d->out.C / 4 : d->out.C;
        k.epi_lds = k.vec16_ok && (cend % 8 == 0) && !vc.persist && !no_epi_lds;
    }
    k.up = up;
    k.updy = updy;
    k.updx = updx;
    k.upH = d->in.H;
    k.upW = d->in.W;
    k.upC = up == 2 ? d->out.C / 4 : d->out.C;
    const int bp = (vc.persist == 6 || vc.persist == 7) ? 32 * vc.pf * (vc.nw / vc.cf) : 32 * vc.nw * vc.pf;
    if (ks == 1) {
        // a 1x1 conv is a GEMM over flattened pixels: one "image" of one row
        const long npix = (long)d->in.B * d->in.H * d->in.W;
        k.B = 1;
        k.H = 1;
        k.W = (int)npix;
        k.Ho = 1;
        k.Wo = (int)npix;
        k.TH = 1;
        k.TW = bp;
    } else {
        k.B = d->in.B;
        k.H = d->in.H;
        k.W = d->in.W;
        k.Ho = d->out.H;
        k.Wo = d->out.W;
        const int cap = vc.persist == 4 ? y6_conv_dma_halo_cap(bp, vc.cs) : vc.persist == 2 ? (st == 2 ? 1161 : (bp <= 128 ? 208 : (bp <= 256 ? 352 : (bp <= 512 ? 660 : 1190)))) : halo_cap(ks, st, vc.pf);
        if (vc.persist == 6)   // two 4-wave blocks per CU
            choose_tile_wreg(k.B, k.Ho, k.Wo, y6_cdiv(y6_cdiv(k.Cout, 32), vc.cf), bp, y6_conv_wreg_max_pieces(vc.nw, st), 2 * device_cus(), st, &k.TH, &k.TW);
        else
            choose_tile(k.Ho, k.Wo, ks, st, bp, cap, &k.TH, &k.TW, vc.persist == 4 ? 16 : 0);
    }
    k.tiles_x = y6_cdiv(k.Wo, k.TW);
    k.tiles_y = y6_cdiv(k.Ho, k.TH);
    k.ntiles = k.B * k.tiles_x * k.tiles_y;
    k.HH = (k.TH - 1) * st + ks;
    k.HWd = (k.TW - 1) * st + ks;
    k.nchunk = y6_cdiv(k.Cin, 32);
    k.ncb = y6_cdiv(y6_cdiv(k.Cout, 32), vc.cf);
    k.ldsA_bytes = vc.persist == 2 ? (st == 2 ? k.HH * 2 * ((k.HWd + 1) / 2) * PIXP : ((k.HH * k.HWd * PIXP + 15) & ~15))
                                   : k.HH * k.HWd * PIXB;
    k.nids = (k.ncb == 1) ? k.ntiles : y6_cdiv(k.ntiles, 8) * 8 * k.ncb;
    k.dbg = nullptr;
    if (const char* tr = getenv("Y6_CONV_TRACE")) {   // debug: device address of a 4 KiB trace buffer (decimal)
        k.dbg = (unsigned long long*)(uintptr_t)strtoull(tr, nullptr, 10);
    }
    L->grid = k.nids;
    if (vc.persist == 6) {
        k.dma_rp = wreg_row_pitch(k.TW, st);
        k.dma_pls = k.HH * k.dma_rp;
        k.dma_nhp = wreg_pieces(k.TH, k.TW, st);
        k.inv_rp = 1.0f / (float)k.dma_rp;
        k.inv_tw = 1.0f / (float)k.TW;
        {
            static const int prio = getenv("Y6_WREG_PRIO") ? atoi(getenv("Y6_WREG_PRIO")) : 0;   // A/B switch (unmeasured: off)
            k.prio_mode = prio;
        }
        L->lds = 2 * (size_t)k.dma_nhp * 1024 + 4 * (size_t)vc.cf * 32 * 4;   // two stage images + bias / post scale / post shift / dequant of the block's couts
    } else if (vc.persist == 4) {
        {
            static const int acc_on = getenv("Y6_DMA_ACC") ? atoi(getenv("Y6_DMA_ACC")) : 1;   // A/B switch
            k.accum_fast = acc_on;
        }
        k.dma_rp = k.HWd;
        k.dma_pls = k.HH * k.HWd;
        k.dma_nhp = y6_cdiv((vc.hc / 8) * k.dma_pls, 64);
        const size_t wp = 9 * vc.cf * (vc.hc / 16);   // tap images of a chunk, KiB
        L->lds = vc.wres ? vc.depth * (size_t)k.dma_nhp * 1024 + (size_t)(k.Cin / vc.hc) * wp * 1024 + 8 * vc.cf * 32 * 4   // halo stages, resident tap images of all chunks
                         : vc.depth * (size_t)(k.dma_nhp + wp) * 1024 + 8 * vc.cf * 32 * 4;   // stages of (halo + tap images), per-channel vectors [2 parities][bias | post scale | post shift | dequant]
    } else if (vc.persist == 2)
        L->lds = 2 * (size_t)k.ldsA_bytes + 2 * (size_t)9 * vc.cf * 1024 + 2 * vc.cf * 32 * 4 + 16;   // two buffers of halo + nine 16-channel tap images, bias x2, dump slot
    else
        L->lds = (size_t)k.ldsA_bytes + 3 * (size_t)vc.cf * 2 * 1024;      // 3-slot ring of tap images
    if (k.epi_lds) {
        const size_t e = 4 * ((size_t)vc.pf * 32 * (vc.cf * 64 + 16) + (size_t)vc.pf * 32 * 4);
        if (e > L->lds) L->lds = e;
    }
    return Y6_OK;
}

template <int CF, int PF, int KS, int ST, int ACT = -1>
int launch_one(const Launch& L, hipStream_t s) {
    if constexpr (KS == 1 && ST == 1 && ACT < 0) {   // the activation-specialised forms of the 1x1 kernel (A/B: Y6_CONV_GENERAL_EPI=1)
        static const bool general = getenv("Y6_CONV_GENERAL_EPI") != nullptr;
        if (!general && L.k.act == Y6_ACT_RELU) return launch_one<CF, PF, 1, 1, Y6_ACT_RELU>(L, s);
        if (!general && L.k.act == Y6_ACT_SILU) return launch_one<CF, PF, 1, 1, Y6_ACT_SILU>(L, s);
        if (!general && L.k.act == Y6_ACT_NONE) return launch_one<CF, PF, 1, 1, Y6_ACT_NONE>(L, s);   // (the training-form graph's 1x1 convs)
    }
    // dynamic LDS above 64 KiB (stride-2 halo + a 3-slot ring of 4-fragment weight images) must be
    // opted into once per kernel; gfx950 has 160 KiB per CU
    static bool big_lds_enabled = false;
    if (L.lds > 64 * 1024 && !big_lds_enabled) {
        Y6_HIP(hipFuncSetAttribute((const void*)conv_mfma_kernel<CF, PF, KS, ST, ACT>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        big_lds_enabled = true;
    }
    Y6_REQUIRE(L.lds <= 128 * 1024, "conv_mfma: tile needs %zu bytes of LDS", L.lds);
    hipLaunchKernelGGL((conv_mfma_kernel<CF, PF, KS, ST, ACT>), dim3(L.grid), dim3(256), L.lds, s, L.k);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

template <int CF, int PF, int WPS, int NW = 4, int ST = 1, int DEPTH = 2>
int launch_pipe(const Launch& L, hipStream_t s) {
    auto kern = conv_mfma_pipe_kernel<CF, PF, WPS, NW, ST, DEPTH>;
    Y6_REQUIRE(L.lds <= 160 * 1024, "conv_mfma: tile needs %zu bytes of LDS", L.lds);
    static OccupancyCache occ;
    int grid = 0;
    {
        int rc = resident_grid(occ, kern, NW * 64, L.lds, 160 * 1024, &grid);
        if (rc) return rc;
    }
    grid -= grid % 8;
    if (grid < 8) grid = 8;
    if (grid > L.grid) grid = L.grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), L.lds, s, L.k);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

template <int CF, int KS>
int launch_stream1x1(const Launch& L, hipStream_t s) {
    auto kern = conv1x1_stream_kernel<CF, KS>;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        Y6_HIP(hipGetDevice(&dev));
        Y6_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int ncb = L.k.ncb;
    const int nfrag = (L.k.W + 31) / 32;
    const int unit = 8 * ncb;                                   // blocks come in groups of 8 (one per XCD) per cout block
    int groups = (n_cu * 4) / unit;                             // ~4 resident blocks per CU
    const int need = y6_cdiv(y6_cdiv(nfrag, 4), 8);             // groups that still get at least one fragment per wave
    if (groups > need) groups = need;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(kern, dim3(groups * unit), dim3(256), 0, s, L.k);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

template <int CF>
int launch_stream1x1_cfg(const Launch& L, hipStream_t s) {
    switch (L.k.Cin / 16) {
        case 4: return launch_stream1x1<CF, 4>(L, s);
        case 8:
            if constexpr (CF == 1) return launch_stream1x1<1, 8>(L, s);
            break;
        case 16:
            if constexpr (CF == 1) return launch_stream1x1<1, 16>(L, s);
            break;
    }
    y6_set_error("conv1x1_stream: unsupported Cin %d", L.k.Cin);
    return Y6_EUNSUPPORTED;
}

template <int CF, int PF>
int launch_cfg(const Launch& L, int ks, int st, hipStream_t s) {
    if (ks == 1 && st == 1) return launch_one<CF, PF, 1, 1>(L, s);
    if (ks == 3 && st == 1) return launch_one<CF, PF, 3, 1>(L, s);
    if (ks == 3 && st == 2) {
        if constexpr (PF == 1) {
            return launch_one<CF, 1, 3, 2>(L, s);
        } else {
            y6_set_error("conv_mfma: stride-2 needs a pf=1 variant");
            return Y6_EUNSUPPORTED;
        }
    }
    y6_set_error("conv_mfma: unsupported ksize/stride %d/%d", ks, st);
    return Y6_EUNSUPPORTED;
}

}  // namespace

// ---- int8 launcher -----------------------------------------------------------------------
namespace {
unsigned half2_bits(float v) {
    const _Float16 h = (_Float16)v;
    unsigned short b;
    memcpy(&b, &h, 2);
    return (unsigned)b | ((unsigned)b << 16);
}
// fp16(amax) and fp16(127 / fp16(amax)): the quantiser constants of include/yolov6_hip.h (y6_conv_i8_desc)
void quantiser_consts(float amax, unsigned* inv2, unsigned* lo2, unsigned* hi2) {
    const float ah = (float)(_Float16)amax;
    *inv2 = half2_bits(127.0f / ah);
    *lo2 = half2_bits(-ah);
    *hi2 = half2_bits(ah);
}
template <int CF, int PF, int KS, int ST>
int launch_i8(const Launch& L, hipStream_t s) {
    static bool big_lds_enabled = false;
    if (L.lds > 64 * 1024 && !big_lds_enabled) {
        Y6_HIP(hipFuncSetAttribute((const void*)conv_i8_kernel<CF, PF, KS, ST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   128 * 1024));
        big_lds_enabled = true;
    }
    Y6_REQUIRE(L.lds <= 128 * 1024, "conv_i8: tile needs %zu bytes of LDS", L.lds);
    hipLaunchKernelGGL((conv_i8_kernel<CF, PF, KS, ST>), dim3(L.grid), dim3(256), L.lds, s, L.k);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
template <int CF, int PF>
int launch_i8_cfg(const Launch& L, int ks, int st, hipStream_t s) {
    if (ks == 1 && st == 1) return launch_i8<CF, PF, 1, 1>(L, s);
    if (ks == 3 && st == 1) return launch_i8<CF, PF, 3, 1>(L, s);
    if constexpr (PF == 1) {
        if (ks == 3 && st == 2) return launch_i8<CF, 1, 3, 2>(L, s);
    }
    y6_set_error("conv_i8: unsupported ksize/stride %d/%d for this tile shape", ks, st);
    return Y6_EUNSUPPORTED;
}
}  // namespace

constexpr bool kI8Wreg2Default = true;    // r04ak: 64 -> 64 stride 2 @160 -> 80: 50.8 -> 36 us; S-QA int8 15 961 -> 16 262 / 16 287 img/s with it and its (deleted) stride-1 sibling
constexpr bool kI8WregDefault = true;   // r04ai: S-QA int8 15 155 -> 15 875 img/s (two steps in flight), 12 825 -> 14 825 one at a time
// what conv_wreg.hip's int8 form handles (it has the fast epilogue only): 3x3 over the producer's int8 twin, whole 64-channel
// stages and 128-cout blocks, no residual, no raw accumulators, a 16-byte aligned fp16 view and / or a 4-byte aligned int8 twin
static bool i8_wreg_ok(const y6_conv_i8_desc* q, int stride, int cout_waves) {
    const y6_conv_desc* d = &q->conv;
    const y6_tensor& o = d->out.data ? d->out : q->q_out;
    if (d->ksize != 3 || d->stride != stride || !q->q_in.data || (q->q_in.C % 64 && q->q_in.C != 32) || o.C % (32 * cout_waves)) return false;
    if (d->res.data || q->acc_out) return false;
    if (d->out.data && (d->out.cstride % 8 || d->out.coff % 8 || ((uintptr_t)d->out.data & 15))) return false;
    if (q->q_out.data && ((q->q_out.cstride | q->q_out.coff) & 3)) return false;
    if (y6_tensor_elems(o) / o.C * o.cstride * 2 >= 0xe0000000ull) return false;
    return true;
}
// what conv_pw.hip's int8 form handles: 1x1 stride 1 over the producer's int8 twin, whole cout blocks, no residual / post-affine / raw
// accumulators, a 16-byte aligned fp16 view and / or a 4-byte aligned int8 twin
static bool i8_pw_ok(const y6_conv_i8_desc* q, int cout_waves) {
    const y6_conv_desc* d = &q->conv;
    const y6_tensor& o = d->out.data ? d->out : q->q_out;
    const int tp = 32 * 2 * (4 / cout_waves);
    if (d->ksize != 1 || d->stride != 1 || !q->q_in.data || !y6_conv_pw_i8_cin_ok(q->q_in.C, tp) || o.C % (32 * cout_waves)) return false;
    if (d->res.data || d->post_scale || q->acc_out) return false;
    if (d->out.data && (d->out.cstride % 8 || d->out.coff % 8 || ((uintptr_t)d->out.data & 15))) return false;
    if (q->q_out.data && ((q->q_out.cstride | q->q_out.coff) & 3)) return false;
    if (y6_tensor_elems(o) / o.C * o.cstride * 2 >= 0xe0000000ull) return false;
    return true;
}
int y6_conv_i8_variant(const y6_conv_i8_desc* q) {
    const y6_conv_desc* d = &q->conv;
    if (d->variant >= 1 && d->variant <= 15) return d->variant;
    const int co = d->out.data ? d->out.C : q->q_out.C;
    static const bool no_dma = getenv("Y6_I8_NO_DMA") != nullptr;   // A/B switch
    // 14 / 15 (round 6): the whole-reduction 1x1 kernel (conv_pw.hip, int8 form): 128- / 64-cout blocks.  Y6_I8_PW=0: A/B switch
    static const bool pw_on = getenv("Y6_I8_PW") ? atoi(getenv("Y6_I8_PW")) != 0 : true;
    if (pw_on && i8_pw_ok(q, 4)) return 14;
    if (pw_on && i8_pw_ok(q, 2)) return 15;
    // 10 / 11 / 12: the register-fed kernels (conv_wreg.hip, int8 form) - the producer's int8 twin, whole 64-channel stages, whole
    // 128-cout blocks; 7 pixel fragments per wave when the 200-pixel items fill the 512 resident blocks, else 4 (the fp16 rule,
    // conv_misc.hip: default_variant); stride 2: 3 fragments.  Y6_I8_WREG=0: A/B switch (the LDS-DMA / per-tap kernels of round 3)
    static const bool wreg = getenv("Y6_I8_WREG") ? atoi(getenv("Y6_I8_WREG")) != 0 : kI8WregDefault;
    if (wreg && !no_dma && i8_wreg_ok(q, d->stride, 4)) {
        const long px = (long)q->q_in.B * (q->q_in.H / d->stride) * (q->q_in.W / d->stride);
        const long items200 = (px + 199) / 200 * (co / 128);
        if (d->stride == 1) return items200 >= 512 ? 10 : 11;
        if (d->stride == 2 && q->q_in.H % 2 == 0 && q->q_in.W % 2 == 0) return 12;
    }
    // 13: the stride-2 kernel with 64-cout blocks (two cout waves x two pixel waves) for the layers 12 cannot take: Cout % 64 == 0,
    // Cin % 64 == 0 or Cin == 32.  Y6_I8_WREG2=0: A/B switch (the per-tap kernel).  (Its stride-1 sibling lost to the LDS-DMA
    // kernels on the 64 -> 64 layers, r04ak, and was deleted.)
    static const bool wreg2 = getenv("Y6_I8_WREG2") ? atoi(getenv("Y6_I8_WREG2")) != 0 : kI8Wreg2Default;
    if (wreg && wreg2 && !no_dma && d->stride == 2 && i8_wreg_ok(q, 2, 2) && q->q_in.H % 2 == 0 && q->q_in.W % 2 == 0) return 13;
    // 7 / 8 / 9: the LDS-DMA kernels (conv_dma.hip) - need the producer's int8 twin and whole 32- (9: 64-) channel chunks
    if (!no_dma && d->ksize == 3 && d->stride == 1 && q->q_in.data && q->q_in.C % 32 == 0 && co >= 64) {
        const long npix = (long)q->q_in.B * q->q_in.H * q->q_in.W;
        if (q->q_in.C % 64 == 0 && npix * y6_cdiv(co, 64) >= 512L * 128) return 9;   // 512-pixel blocks, 64-channel chunks
        return 8;                                                                  // 256-pixel blocks
    }
    if (d->ksize == 3 && d->stride == 1) return co >= 64 ? 5 : 4;            // c2p2 / c1p2
    return co >= 256 ? 3 : (co >= 64 ? 2 : 1);                                // c4p1 / c2p1 / c1p1
}

int y6_conv_i8_launch(const y6_conv_i8_desc* q, hipStream_t s) {
    y6_conv_desc d = q->conv;
    const bool has_out = d.out.data != nullptr, has_qout = q->q_out.data != nullptr, has_qin = q->q_in.data != nullptr;
    Y6_REQUIRE(has_out || has_qout, "conv_i8: neither an fp16 nor an int8 output");
    Y6_REQUIRE(d.w_packed && q->dequant, "conv_i8: packed int8 weights and the dequantisation vector are required");
    Y6_REQUIRE((d.ksize == 1 && d.stride == 1) || (d.ksize == 3 && (d.stride == 1 || d.stride == 2)), "conv_i8: k%d s%d", d.ksize,
               d.stride);
    if (!has_out) {   // geometry comes from the int8 output view
        d.out = q->q_out;
        d.out.data = nullptr;
    }
    if (has_qin) {
        const y6_tensor& t = q->q_in;
        Y6_REQUIRE(t.C % 16 == 0 && t.cstride % 16 == 0 && t.coff % 16 == 0 && ((uintptr_t)t.data & 15) == 0,
                   "conv_i8: the int8 input view needs 16-channel alignment");
        if (!d.in.data) d.in = t;
        Y6_REQUIRE(t.B == d.in.B && t.H == d.in.H && t.W == d.in.W && t.C == d.in.C, "conv_i8: int8 input view shape");
    } else {
        Y6_REQUIRE(d.in.data && d.in.C % 8 == 0 && d.in.cstride % 8 == 0 && d.in.coff % 8 == 0 && ((uintptr_t)d.in.data & 15) == 0,
                   "conv_i8: the fp16 input view needs 8-channel alignment");
        Y6_REQUIRE(q->in_amax > 0.f, "conv_i8: in_amax must be positive");
    }
    Y6_REQUIRE(((uintptr_t)d.w_packed & 15) == 0, "conv_i8: unaligned weights");
    Y6_REQUIRE(y6_tensor_elems(d.in) < ((size_t)1 << 31) && y6_tensor_elems(d.out) < ((size_t)1 << 31), "conv_i8: tensor too large");
    if (has_qout) {
        const y6_tensor& t = q->q_out;
        Y6_REQUIRE(t.B == d.out.B && t.H == d.out.H && t.W == d.out.W && t.C == d.out.C && t.C % 4 == 0 && t.cstride % 4 == 0 &&
                       t.coff % 4 == 0 && q->q_out_amax > 0.f,
                   "conv_i8: int8 output view");
    }
    const int variant = y6_conv_i8_variant(q);
    if (variant >= 14) {   // conv_pw.hip
        Y6_REQUIRE(variant <= 15 && i8_pw_ok(q, variant == 14 ? 4 : 2),
                   "conv_i8: the whole-reduction 1x1 variants need k1 s1, an int8 input view with Cin in {64 .. 1024}, whole cout blocks, no residual / post-affine, aligned outputs");
        static const int kv14 = variant_index("pw_c4p2"), kv15 = variant_index("pw_c2p2");
        Launch L;
        int rc = build_launch(&d, variant == 14 ? kv14 : kv15, 0, 0, 0, &L);
        if (rc) return rc;
        ConvKArgs& k = L.k;
        k.qscale = q->dequant;
        k.qin = (const signed char*)q->q_in.data;
        k.qin_cs = q->q_in.cstride;
        k.qin_co = q->q_in.coff;
        k.qout = (signed char*)q->q_out.data;
        k.qout_cs = q->q_out.cstride;
        k.qout_co = q->q_out.coff;
        if (has_qout) quantiser_consts(q->q_out_amax, &k.qo_inv2, &k.qo_lo2, &k.qo_hi2);
        if (!has_out) k.out = nullptr;
        return y6_conv_pw_launch(&L, variant == 14 ? 4 : 2, 2, 1, s);
    }
    const bool wreg = variant >= 10;
    const bool dma = variant >= 7 && !wreg;
    if (dma)
        Y6_REQUIRE(d.ksize == 3 && d.stride == 1 && has_qin && q->q_in.C % 32 == 0, "conv_i8: the LDS-DMA variants need k3 s1, an int8 input view and Cin %% 32 == 0");
    if (wreg)
        Y6_REQUIRE(i8_wreg_ok(q, variant >= 12 ? 2 : 1, variant == 13 ? 2 : 4),
                   "conv_i8: the register-fed variants need k3, an int8 input view, Cin %% 64 == 0 (or 32), Cout %% 128 == 0 (13: %% 64), no residual, aligned outputs");
    Y6_REQUIRE(dma || wreg || !(d.stride == 2 && kVariants[variant].pf != 1), "conv_i8: stride 2 needs a pf=1 variant");
    Launch L;
    static const int kv7 = variant_index("i8_dma8_c2p2"), kv8 = variant_index("dma_c2p2"), kv9 = variant_index("i8_dmaw8_c2p2"),
                     kv10 = variant_index("wreg_p7"), kv11 = variant_index("wreg_p4"), kv12 = variant_index("wregs2_p3"),
                     kv13 = variant_index("i8_wreg2s2_p2");
    const int kv = wreg ? (variant == 10 ? kv10 : variant == 11 ? kv11 : variant == 12 ? kv12 : kv13)
                        : dma ? (variant == 7 ? kv7 : (variant == 8 ? kv8 : kv9)) : variant;   // kVariants row that sizes the tile
    if (variant == 9) Y6_REQUIRE(q->q_in.C % 64 == 0, "conv_i8: variant 9 needs Cin %% 64 == 0");
    int rc = build_launch(&d, kv, 0, 0, 0, &L);
    if (rc) return rc;
    ConvKArgs& k = L.k;
    k.nchunk = y6_cdiv(k.Cin, 64);
    k.qscale = q->dequant;
    quantiser_consts(has_qin ? 1.f : q->in_amax, &k.q_inv2, &k.q_lo2, &k.q_hi2);
    k.qin = (const signed char*)q->q_in.data;
    k.qin_cs = q->q_in.cstride;
    k.qin_co = q->q_in.coff;
    k.qout = (signed char*)q->q_out.data;
    k.qout_cs = q->q_out.cstride;
    k.qout_co = q->q_out.coff;
    if (has_qout) quantiser_consts(q->q_out_amax, &k.qo_inv2, &k.qo_lo2, &k.qo_hi2);
    k.acc_out = (int*)q->acc_out;
    if (!has_out) {
        k.out = nullptr;
        k.epi_lds = 0;
    }
    if (wreg) return y6_conv_wreg_launch(&L, kVariants[kv].pf, kVariants[kv].cf, kVariants[kv].nw / kVariants[kv].cf, kVariants[kv].cs, 1, s);
    if (dma) return y6_conv_dma_launch(&L, kVariants[kv].cf, kVariants[kv].pf, kVariants[kv].nw, 2, 1, kVariants[kv].hc, 1, 1, 0, s);
    switch (variant) {
        case 1: return launch_i8_cfg<1, 1>(L, d.ksize, d.stride, s);
        case 2: return launch_i8_cfg<2, 1>(L, d.ksize, d.stride, s);
        case 3: return launch_i8_cfg<4, 1>(L, d.ksize, d.stride, s);
        case 4: return launch_i8_cfg<1, 2>(L, d.ksize, d.stride, s);
        case 5: return launch_i8_cfg<2, 2>(L, d.ksize, d.stride, s);
        case 6: return launch_i8_cfg<4, 2>(L, d.ksize, d.stride, s);
    }
    return Y6_EINVAL;
}

extern "C" int y6_conv_variants(void) { return kNumVariants; }
extern "C" const char* y6_conv_variant_name(int i) {
    return (i >= 0 && i < kNumVariants) ? kVariants[i].name : "?";
}

int y6_conv_mfma_supports(const y6_conv_desc* d, int variant) {
    if (variant < 1 || variant >= kNumVariants) return 0;
    const VariantCfg& vc = kVariants[variant];
    if (vc.i8only) return 0;
    const int ks = d->ksize, st = d->stride;
    if (!((ks == 1 && st == 1) || (ks == 3 && (st == 1 || st == 2)))) return 0;
    if (st == 2 && vc.pf != 1 && vc.persist != 6) return 0;
    if (vc.persist == 3) {   // streaming 1x1: Cin = 16 * {4, 8, 16} exactly, weights in registers
        if (ks != 1 || st != 1 || d->w_packed == nullptr) return 0;
        const int ksn = d->in.C / 16;
        if (d->in.C % 16 || !(ksn == 4 || ksn == 8 || ksn == 16)) return 0;
        if (vc.cf == 2 && ksn != 4) return 0;                // two cout fragments fit without spills for Cin 64 only (a spill is a
                                                             // scratch access, i.e. a vmcnt wait that also waits for the pixel prefetch)
        if (d->in.cstride % 8 || d->in.coff % 8 || ((uintptr_t)d->in.data & 15) || ((uintptr_t)d->w_packed & 15)) return 0;
        if (y6_tensor_elems(d->in) >= (size_t)1 << 31 || y6_tensor_elems(d->out) >= (size_t)1 << 31) return 0;
        return vc.cf <= y6_cdiv(d->out.C, 32);
    }
    if (vc.persist == 7) {   // conv_pw.hip: whole 64-channel stages, whole cout blocks, plain bias + activation into a 16-byte aligned view
        if (ks != 1 || st != 1 || d->w_packed == nullptr || d->post_scale != nullptr) return 0;
        if (d->res.data != nullptr && (d->res.cstride % 8 || d->res.coff % 8 || ((uintptr_t)d->res.data & 15) || y6_tensor_elems(d->res) * 2 >= 0xe0000000ull)) return 0;
        if (!y6_conv_pw_cin_ok(d->in.C, 32 * vc.pf * (vc.nw / vc.cf)) || d->out.C % (32 * vc.cf) || d->in.cstride % 8 || d->in.coff % 8) return 0;
        if (d->out.cstride % 8 || d->out.coff % 8 || ((uintptr_t)d->out.data & 15)) return 0;
        if (((uintptr_t)d->in.data & 15) || ((uintptr_t)d->w_packed & 15)) return 0;
        if (y6_tensor_elems(d->in) * 2 >= 0xe0000000ull || y6_tensor_elems(d->out) * 2 >= 0xe0000000ull) return 0;
        return 1;
    }
    if (vc.persist && ks != 3) return 0;
    if (vc.persist == 6) {   // weights through registers: whole 32-channel stages, whole cout blocks, 16-byte pieces straight from the tensor
        if (st != vc.cs || d->w_packed == nullptr) return 0;
        if (d->in.C % 32 || d->out.C % (32 * vc.cf) || d->in.cstride % 8 || d->in.coff % 8) return 0;
        if (((uintptr_t)d->in.data & 15) || ((uintptr_t)d->w_packed & 15)) return 0;
        if (y6_tensor_elems(d->in) * 2 >= 0xe0000000ull || y6_tensor_elems(d->out) * 2 >= 0xe0000000ull) return 0;
        return 1;
    }
    if (vc.persist == 4) {   // LDS-DMA kernels: whole 16-channel chunks, 16-byte pieces straight from the tensor
        if (st != vc.cs || d->w_packed == nullptr) return 0;   // (vc.st is the issue mode here; vc.cs the stride)
        if (d->in.C % vc.hc || d->in.cstride % 8 || d->in.coff % 8) return 0;
        if (vc.wres && d->in.C > 64) return 0;   // 9 x Cin x 64 couts of fp16 must fit beside two halo stages
        if (((uintptr_t)d->in.data & 15) || ((uintptr_t)d->w_packed & 15)) return 0;
        if (y6_tensor_elems(d->in) * 2 >= 0xe0000000ull || y6_tensor_elems(d->out) * 2 >= 0xe0000000ull) return 0;   // byte offsets + the range-check sentinel
        return vc.cf <= y6_cdiv(d->out.C, 32);
    }
    if (vc.persist == 2 && st != vc.st) return 0;
    if (vc.persist == 2 && vc.depth == 3 && (y6_cdiv(d->in.C, 16) & 1)) return 0;   // two chunks per loop trip
    if (d->w_packed == nullptr) return 0;
    // 16-byte halo pieces need 8-channel alignment of the input view
    if (d->in.C % 8 || d->in.cstride % 8 || d->in.coff % 8) return 0;
    if (((uintptr_t)d->in.data & 15) || ((uintptr_t)d->w_packed & 15)) return 0;
    // element offsets are 32-bit in the kernel
    if (y6_tensor_elems(d->in) >= (size_t)1 << 31 || y6_tensor_elems(d->out) >= (size_t)1 << 31) return 0;
    // don't spend a 4-frag (128 cout) block on a narrow layer
    const int cfr = y6_cdiv(d->out.C, 32);
    if (vc.cf > cfr) return 0;
    return 1;
}

extern "C" int y6_conv_launch_geometry(const y6_conv_desc* d, int variant, y6_conv_geometry* g) {
    Y6_REQUIRE(d && g, "conv_launch_geometry: null argument");
    if (variant < 1 || variant >= kNumVariants || !y6_conv_mfma_supports(d, variant)) {
        y6_set_error("conv_launch_geometry: variant %d does not take this conv", variant);
        return Y6_EUNSUPPORTED;
    }
    Launch L;
    int rc = build_launch(d, variant, 0, 0, 0, &L);
    if (rc) return rc;
    const VariantCfg& vc = kVariants[variant];
    const ConvKArgs& k = L.k;
    memset(g, 0, sizeof(*g));
    g->tile_h = k.TH;
    g->tile_w = k.TW;
    g->tiles_x = k.tiles_x;
    g->tiles_y = k.tiles_y;
    g->items = k.nids;
    g->cout_blocks = k.ncb;
    g->block_pixels = vc.persist == 6 ? 32 * vc.pf * (vc.nw / vc.cf) : 32 * vc.nw * vc.pf;
    g->halo_h = k.HH;
    g->halo_w = k.HWd;
    if (vc.persist == 6) {
        g->halo_pieces = k.dma_nhp;
        g->halo_pieces_max = y6_conv_wreg_max_pieces(vc.nw, d->stride);
        g->row_pitch = k.dma_rp;
    } else if (vc.persist == 4) {
        g->halo_pieces = k.dma_nhp;
    }
    g->lds_bytes = L.lds;
    g->lds_limit = (vc.persist == 0 ? 128 : 160) * 1024;
    return Y6_OK;
}

int y6_conv_mfma_launch(const y6_conv_desc* d, int variant, hipStream_t s, int up, int updy, int updx) {
    if (!y6_conv_mfma_supports(d, variant)) {
        y6_set_error("conv_mfma: variant %d does not support this conv (k%d s%d Cin %d Cout %d)", variant, d->ksize,
                     d->stride, d->in.C, d->out.C);
        return Y6_EUNSUPPORTED;
    }
    if (kVariants[variant].persist == 7 && up) {   // conv_pw.hip scatters whole cout blocks of the fused [4 Cout x Cin] form
        if (up != 2 || d->res.data != nullptr || (d->out.C / 4) % (32 * kVariants[variant].cf) != 0 || y6_tensor_elems(d->out) * 8 >= 0xe0000000ull) {
            y6_set_error("conv_pw: the convT scatter needs the fused form with whole cout blocks per sub-kernel");
            return Y6_EUNSUPPORTED;
        }
    } else if (kVariants[variant].persist && up) {
        y6_set_error("conv_mfma: persistent variants do not implement the convT scatter");
        return Y6_EUNSUPPORTED;
    }
    Launch L;
    int rc = build_launch(d, variant, up, updy, updx, &L);
    if (rc) return rc;
    const VariantCfg& vc = kVariants[variant];
    switch (vc.persist) {
        case 0:
            if (vc.cf == 1 && vc.pf == 1) return launch_cfg<1, 1>(L, d->ksize, d->stride, s);
            if (vc.cf == 2 && vc.pf == 1) return launch_cfg<2, 1>(L, d->ksize, d->stride, s);
            if (vc.cf == 4 && vc.pf == 1) return launch_cfg<4, 1>(L, d->ksize, d->stride, s);
            if (vc.cf == 1 && vc.pf == 2) return launch_cfg<1, 2>(L, d->ksize, d->stride, s);
            if (vc.cf == 2 && vc.pf == 2) return launch_cfg<2, 2>(L, d->ksize, d->stride, s);
            if (vc.cf == 4 && vc.pf == 2) return launch_cfg<4, 2>(L, d->ksize, d->stride, s);
            break;
        case 2:
            if (vc.cf == 2 && vc.pf == 2) return launch_pipe<2, 2, 2>(L, s);
            if (vc.cf == 2 && vc.pf == 1) return launch_pipe<2, 1, 2>(L, s);
            if (vc.cf == 1 && vc.pf == 2) return launch_pipe<1, 2, 2>(L, s);
            break;
        case 3:
            return vc.cf == 1 ? launch_stream1x1_cfg<1>(L, s) : launch_stream1x1_cfg<2>(L, s);
        case 4:
            return y6_conv_dma_launch(&L, vc.cf, vc.pf, vc.nw, vc.depth, vc.st, vc.hc, vc.cs, 0, vc.wres, s);
        case 6:
            return y6_conv_wreg_launch(&L, vc.pf, vc.cf, vc.nw / vc.cf, vc.cs, 0, s);
        case 7:
            return y6_conv_pw_launch(&L, vc.cf, vc.pf, 0, s);
    }
    return Y6_EINVAL;
}
