// conv_common.hpp - what the MFMA conv translation units share: kernel arguments, LDS-DMA helpers, the fused epilogues,
// the variant table entry and the launch record (conv_mfma.hip: per-tap / persistent / pipelined kernels and the
// dispatcher; conv_dma.hip: the LDS-DMA fed 3x3 kernels).  Everything lives in an anonymous namespace: each TU gets its
// own copy, nothing is exported.
#pragma once
#include <map>
#include <mutex>
#include <utility>
#include <cstdlib>

#include "common.hpp"

namespace {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

constexpr int PIXB = 80;  // LDS bytes per halo pixel: 32 ch * 2 B + 16 B pad (bank spread)

struct ConvKArgs {
    const __half* in;
    __half* out;
    const __half* wpk;
    const float* bias;
    const float* pscale;
    const float* pshift;
    const __half* res;
    const float* res_alpha;
    int B, H, W, Ho, Wo;
    int Cin, Cout;
    int in_cs, in_co, out_cs, out_co, res_cs, res_co;
    int TH, TW, tiles_x, tiles_y, ntiles;
    int HH, HWd;
    int nchunk, ncb;
    int ldsA_bytes;
    int act;
    int vec_ok;    // 8-byte stores legal (cstride/coff % 4 == 0)
    int vec16_ok;  // 16-byte stores legal (cstride/coff % 8 == 0, base 16-byte aligned)
    int epi_lds;   // stage the output tile through LDS and store whole NHWC rows (needs vec16_ok, Cout % 8 == 0)
    int res_vec;   // residual readable as 8-byte pieces (res cstride/coff % 4 == 0, base 8-byte aligned)
    int nids;  // padded (tile, cout-block) id space of the 1-D grid
    unsigned long long* dbg;  // optional s_memtime trace of block 0 / wave 0 (env Y6_CONV_TRACE), 2 x 256 words
    int up, updy, updx, upH, upW, upC;  // up: 0 none, 1 one (dy,dx) sub-conv, 2 all four fused (cout block -> sub)
    // int8 kernels (conv_i8_kernel) only
    const float* qscale;            // [Cout] s_x * s_w[c]
    unsigned q_inv2, q_lo2, q_hi2;  // half2 constants of the input quantiser: 127/amax, -amax, +amax
    const signed char* qin;         // optional int8 NHWC input view (then `in` is not read)
    int qin_cs, qin_co;
    signed char* qout;              // optional int8 NHWC copy of the output, quantised for its consumers
    int qout_cs, qout_co;
    unsigned qo_inv2, qo_lo2, qo_hi2;
    int* acc_out;                   // optional raw int32 accumulators [pixel][Cout] (parity tests)
    // LDS-DMA kernels (conv_dma.hip) only: halo row pitch (slots of 16 B), slots per plane, 1 KiB pieces of the two planes
    int dma_rp, dma_pls, dma_nhp;
    // conv_wreg.hip only: 1.0f / halo row pitch, 1.0f / tile width (div_small)
    float inv_rp, inv_tw;
    int prio_mode;   // conv_wreg.hip: bit 0 = matrix-phase priority falls from stage to stage, bit 1 = non-matrix phases at priority 3
    int accum_fast;  // conv_dma.hip: 1 = accumulating convs (res == out) take the end-of-item fast epilogue (A/B switch Y6_DMA_ACC=0)
};

// LDS-DMA of 16 B per lane (1 KiB per wave) issued from inline asm: hipcc does not see it, so it
// cannot (a) insert a conservative `s_waitcnt vmcnt(0)` before every later ds_read because the DMA
// "might alias", nor (b) count it - completion is awaited by the kernel's own vmcnt(0) at the chunk
// boundary.  M0 (the LDS destination base) is saved/restored inside the statement
// (cdna_hip_programming.md §5.7).  `lds_dst` must be wave-uniform.
__device__ __forceinline__ void lds_dma16(const void* gsrc_lane, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc_lane), "s"(lds_dst)
        : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// ---- LDS-DMA through a buffer descriptor (conv_dma.hip, conv_wreg.hip)
__device__ __forceinline__ i32x4_t make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)(uintptr_t)base;
    i32x4_t r;
    r[0] = (int)(unsigned)(p & 0xffffffffu);
    r[1] = (int)(unsigned)((p >> 32) & 0xffffu);   // stride 0: raw buffer, byte offsets, range check against num_records
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
}

// one LDS-DMA piece: lane i writes 16 B to lds_dst + 16*i from rsrc.base + voff(lane) + soff; out of range -> zeros.
// Issued from inline asm (hipcc neither counts it nor fences later ds_reads against it; completion = the kernel's own
// vmcnt(0) + barrier).  M0 is saved/restored inside the statement (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void dma16(const i32x4_t& rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
        : "memory");   // (both are wave-uniform by construction; under register pressure hipcc may still carry them in VGPRs)
}

// The argument block again, loaded where it is used.  The epilogue reads two dozen fields of ConvKArgs; taken from the
// by-value kernel parameter they are loaded at kernel entry and stay live across the chunk loop (200+ spilled SGPRs,
// v_readlane traffic inside the loop).  Re-reading the kernarg segment through a laundered pointer gives the epilogue
// its own short-lived copies (scalar loads, once per work item).
__device__ __forceinline__ ConvKArgs reload_args() {
    unsigned long long v = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(v));
    const __attribute__((address_space(4))) unsigned* src = reinterpret_cast<const __attribute__((address_space(4))) unsigned*>(v);
    static_assert(sizeof(ConvKArgs) % 4 == 0, "ConvKArgs is copied dword by dword");
    ConvKArgs r;
    unsigned* dst = reinterpret_cast<unsigned*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(ConvKArgs) / 4; ++i) dst[i] = src[i];   // scalar loads of the fields the caller uses
    return r;
}

constexpr unsigned kOob = 0xf0000000u;   // voffset of a piece that must read zeros / a store that must be dropped (tensors stay below 3.5 GiB)

// lane (0..31) -> pixel of the fragment it holds (see the header comment)
__device__ __forceinline__ int frag_pixel(int l) {
    return l < 4 ? l : (l < 12 ? l + 12 : (l < 16 ? l - 8 : (l < 20 ? l + 8 : (l < 28 ? l - 12 : l))));
}

template <int KS, int ST, int PF>
struct HaloCap {
    static constexpr int value = (KS == 1) ? PF * 128 : (ST == 1 ? (PF == 2 ? 352 : 208) : 576);
};

// Epilogue: bias (+affine) + activation (+residual) -> fp16 NHWC.
// C/D layout: col = pixel (lane&31), row = cout = (r&3) + 8*(r>>2) + 4*(lane>>5): a lane holds four
// consecutive couts of ONE pixel per group of 4 accumulator registers (group g = r>>2), and lane^32
// holds the next four couts of the same pixel.
//   * bias values are fetched up front by load_bias() (4 x 16-byte loads per fragment, issued before
//     the main loop) instead of 4 dependent loads in front of every store group;
//   * v_permlane32_swap pairs groups (g, g+1): afterwards lanes 0-31 own couts 8g..8g+7 and lanes 32-63
//     own 8(g+1)..8(g+1)+7 of their pixel -> ONE 16-byte store per lane per pair instead of two 8-byte
//     ones (the scattered 8-byte stores were store-issue bound: profiles/r01/conv_ablation_abl01.log).
// finish(): acc + bias -> (affine) -> activation -> (+alpha*residual), for the 16 values a lane holds of one
// fragment.  All mode decisions are wave-uniform and hoisted out of the element loop (the r07 trace
// showed ~4000 cycles per fragment when `switch(act)` / affine / residual were tested per element).
template <int ACT>
__device__ __forceinline__ float act_const(float v) {
    if constexpr (ACT == Y6_ACT_RELU) return v > 0.f ? v : 0.f;
    if constexpr (ACT == Y6_ACT_SILU) {
        v = y6_round_f16(v);            // the conv output is an fp16 tensor in the reference (common.hpp)
        return y6_div_tame(v, 1.f + __expf(fminf(-v, 80.f)))   /* exp stays finite: v / inf = -0 either way after the fp16 rounding */;
    }
    if constexpr (ACT == Y6_ACT_HARDSWISH) {
        v = y6_round_f16(v);
        float r = v + 3.f;
        r = r < 0.f ? 0.f : (r > 6.f ? 6.f : r);
        return v * r * (1.f / 6.f);
    }
    return v;
}

template <int ACT>
__device__ __forceinline__ void finish16(const ConvKArgs& a, const f32x16_t& acc, const float (&bias)[16], int cfrag,
                                         int kh, int cend, const __half* rrow, float ralpha, float (&v)[16]) {
    if (a.pscale == nullptr && rrow == nullptr) {   // the common case: conv + bias + act
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = act_const<ACT>(acc[r] + bias[r]);
        return;
    }
    // the residual of this lane's 16 channels (four groups of 4 consecutive couts): 8-byte loads where the view allows - the
    // accumulating data-gradient convs of the training step (res == out) spent their epilogue on 16 two-byte loads per lane
    float rv[16];
    if (rrow) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = cfrag + 8 * g + 4 * kh;
            if (a.res_vec && c0 >= 0 && c0 + 3 < cend) {
                const uint2 raw = *reinterpret_cast<const uint2*>(rrow + c0);
                const __half* h = reinterpret_cast<const __half*>(&raw);
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[g * 4 + j] = __half2float(h[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[g * 4 + j] = (c0 + j >= 0 && c0 + j < cend) ? __half2float(rrow[c0 + j]) : 0.f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = cfrag + 8 * (r >> 2) + 4 * kh + (r & 3);
        float x = acc[r] + bias[r];
        if (c < cend) {
            if (a.pscale) x = y6_round_f16(x) * a.pscale[c] + a.pshift[c];      // QARepVGG: conv -> BN are two fp16 ops
            x = act_const<ACT>(x);
            if (rrow) x = y6_round_f16(x) + y6_round_f16(ralpha * rv[r]);       // BottleRep: out + alpha*x
        } else {
            x = act_const<ACT>(x);
        }
        v[r] = x;
    }
}

// ACT >= 0: the activation is a compile-time fact of the kernel instantiation (round 5: the four-way switch below is instantiated
// per fragment in fully unrolled epilogues - three quarters of a 100-380 KB kernel that an instruction cache of 64 KB has to
// step over); ACT < 0: decided per launch.
template <int ACT = -1>
__device__ __forceinline__ void finish16_any(const ConvKArgs& a, const f32x16_t& acc, const float (&bias)[16], int cfrag,
                                             int kh, int cend, const __half* rrow, float ralpha, float (&v)[16]) {
    if constexpr (ACT >= 0) {
        finish16<ACT>(a, acc, bias, cfrag, kh, cend, rrow, ralpha, v);
        return;
    }
    switch (a.act) {   // one wave-uniform branch per fragment
        case Y6_ACT_RELU: finish16<Y6_ACT_RELU>(a, acc, bias, cfrag, kh, cend, rrow, ralpha, v); break;
        case Y6_ACT_SILU: finish16<Y6_ACT_SILU>(a, acc, bias, cfrag, kh, cend, rrow, ralpha, v); break;
        case Y6_ACT_HARDSWISH: finish16<Y6_ACT_HARDSWISH>(a, acc, bias, cfrag, kh, cend, rrow, ralpha, v); break;
        default: finish16<Y6_ACT_NONE>(a, acc, bias, cfrag, kh, cend, rrow, ralpha, v); break;
    }
}

template <int CF>
struct BiasRegs {
    float v[CF][16];
};

template <int CF>
__device__ __forceinline__ void load_bias(const ConvKArgs& a, int cb, int upc0, int lane, BiasRegs<CF>& bz) {
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = (cb * CF + cf) * 32 + 8 * g + 4 * (lane >> 5) - upc0;
            const int cend = a.up == 2 ? a.upC : a.Cout;
            if (a.bias != nullptr && c0 + 3 < cend && c0 >= 0) {
                const float4 t = *reinterpret_cast<const float4*>(a.bias + c0);
                bz.v[cf][g * 4 + 0] = t.x;
                bz.v[cf][g * 4 + 1] = t.y;
                bz.v[cf][g * 4 + 2] = t.z;
                bz.v[cf][g * 4 + 3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bz.v[cf][g * 4 + j] = (a.bias != nullptr && c0 + j < cend && c0 + j >= 0) ? a.bias[c0 + j] : 0.f;
            }
        }
}

template <int CF, int PF, int ACT = -1>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& a, const f32x16_t (&acc)[CF][PF], const int (&opix)[PF],
                                              int cb, int upc0, int lane, const BiasRegs<CF>& bz) {
    const float ralpha = (a.res != nullptr && a.res_alpha != nullptr) ? *a.res_alpha : 1.f;
    const int cend = a.up == 2 ? a.upC : a.Cout;
    const int kh = lane >> 5;
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const bool pvalid = opix[pf] >= 0;
        const size_t prow = pvalid ? (size_t)opix[pf] : 0;
        __half* orow = a.out + prow * a.out_cs + a.out_co;
        const __half* rrow = a.res ? a.res + prow * a.res_cs + a.res_co : nullptr;
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) {
            const int cfrag = (cb * CF + cf) * 32 - upc0;   // first output channel of this fragment
            // 1) finish the 16 values of this lane
            float v[16];
            finish16_any<ACT>(a, acc[cf][pf], bz.v[cf], cfrag, kh, cend, (pvalid ? rrow : nullptr), ralpha, v);
            // 2) pack to fp16 pairs: group g -> dwords pk[g][0..1]
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
            if (a.vec16_ok && (cfrag + 32) <= cend && cfrag >= 0) {
                // 3) pair groups across the two half-waves, 16-byte stores
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    unsigned lo0 = pk[2 * gp][0], lo1 = pk[2 * gp][1], hi0 = pk[2 * gp + 1][0], hi1 = pk[2 * gp + 1][1];
                    auto s0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
                    // lanes 0-31: {own g, partner's g} = couts 8g..8g+7 ; lanes 32-63: couts 8(g+1)..8(g+1)+7
                    if (pvalid) {
                        uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        *reinterpret_cast<uint4*>(orow + cfrag + 16 * gp + 8 * kh) = o;
                    }
                }
            } else if (pvalid) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = cfrag + 8 * g + 4 * kh;
                    if (c0 >= cend || c0 < 0) continue;
                    if (a.vec_ok && (c0 + 3) < cend) {
                        *reinterpret_cast<uint2*>(orow + c0) = make_uint2(pk[g][0], pk[g][1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (c0 + j < cend) orow[c0 + j] = __float2half(v[g * 4 + j]);
                    }
                }
            }
        }
    }
}

// LDS-staged epilogue: the direct path above makes every store instruction touch 32 different cache
// lines with 16-32 bytes each (store-issue / request-rate bound: the 210 MB stem output ran at
// ~550 GB/s).  Here each wave writes its [PF*32 pixels][CF*32 couts] fp16 tile into its own LDS
// region, then reads it back so that consecutive lanes hold consecutive 16-byte pieces of one pixel's
// channel row: a store instruction now covers 64*16 B of (at most 64/(CF*4)) complete NHWC rows.
// Wave-private region -> no block barrier (LDS ops of one wave complete in order; lgkmcnt(0) between
// the phases).  Callers must have passed a block barrier after the last main-loop LDS read.
template <int CF, int PF, int ACT = -1>
__device__ __forceinline__ void conv_epilogue_lds(const ConvKArgs& a, const f32x16_t (&acc)[CF][PF],
                                                  const int (&opix)[PF], int cb, int upc0, int lane, int wave,
                                                  const BiasRegs<CF>& bz, char* lds) {
    constexpr int RS = CF * 64 + 16;                 // row pitch: CF*32 couts * 2 B + 16 B (bank spread)
    constexpr int ROWS = PF * 32;
    constexpr int REGION = ROWS * RS + ROWS * 4;      // tile + one int (output pixel index) per row
    char* tile = lds + wave * REGION;
    int* rowpix = reinterpret_cast<int*>(tile + ROWS * RS);
    const float ralpha = (a.res != nullptr && a.res_alpha != nullptr) ? *a.res_alpha : 1.f;
    const int cend = a.up == 2 ? a.upC : a.Cout;
    const int kh = lane >> 5;
    const int cblock = cb * CF * 32 - upc0;           // first output channel of this block's tile
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const bool pvalid = opix[pf] >= 0;
        const size_t prow = pvalid ? (size_t)opix[pf] : 0;
        const __half* rrow = a.res ? a.res + prow * a.res_cs + a.res_co : nullptr;
        const int row = pf * 32 + (lane & 31);
        if (kh == 0) rowpix[row] = opix[pf];
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) {
            const int cfrag = cblock + cf * 32;
            float v[16];
            finish16_any<ACT>(a, acc[cf][pf], bz.v[cf], cfrag, kh, cend, (pvalid ? rrow : nullptr), ralpha, v);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef _Float16 h4v __attribute__((ext_vector_type(4)));
                h4v o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)v[g * 4 + j];
                *reinterpret_cast<h4v*>(tile + row * RS + cf * 64 + g * 16 + kh * 8) = o;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int PPR = CF * 4;                        // 16-byte pieces per row
    constexpr int NPC = ROWS * PPR;                    // pieces per wave tile
#pragma unroll
    for (int i = 0; i < NPC / 64; ++i) {
        const int q = lane + 64 * i;
        const int row = q / PPR, pc = q - row * PPR;
        const int op = rowpix[row];
        const int c0 = cblock + pc * 8;
        if (op >= 0 && c0 + 8 <= cend) {
            const uint4 v = *reinterpret_cast<const uint4*>(tile + row * RS + pc * 16);
            *reinterpret_cast<uint4*>(a.out + (size_t)op * a.out_cs + a.out_co + c0) = v;
        }
    }
}

template <int CF, int PF>
constexpr int epi_lds_bytes() {
    return 4 * (PF * 32 * (CF * 64 + 16) + PF * 32 * 4);
}

__device__ __forceinline__ unsigned q8_pair(unsigned x2, unsigned inv2, unsigned lo2, unsigned hi2) {
    unsigned r;
    asm("v_pk_max_f16 %0, %1, %2\n\tv_pk_min_f16 %0, %0, %3\n\tv_pk_fma_f16 %0, %0, %4, %5"
        : "=&v"(r)
        : "v"(x2), "v"(lo2), "v"(hi2), "v"(inv2), "v"(0x66006600u));
    return r;   // two fp16 values 1536 + q: low bytes are the int8 codes
}
__device__ __forceinline__ unsigned q8_quad(unsigned a, unsigned b, unsigned inv2, unsigned lo2, unsigned hi2) {
    return __builtin_amdgcn_perm(q8_pair(b, inv2, lo2, hi2), q8_pair(a, inv2, lo2, hi2), 0x06040200u);
}
__device__ __forceinline__ uint4 q8_piece(const uint4& lo, const uint4& hi, unsigned inv2, unsigned lo2, unsigned hi2) {
    return make_uint4(q8_quad(lo.x, lo.y, inv2, lo2, hi2), q8_quad(lo.z, lo.w, inv2, lo2, hi2),
                      q8_quad(hi.x, hi.y, inv2, lo2, hi2), q8_quad(hi.z, hi.w, inv2, lo2, hi2));
}


// int8 epilogue of one cout fragment (32 couts) x PF pixel fragments: exact int32 -> fp32 (one rounding), * s_x*s_w[c]
// (one rounding), then the arithmetic of the fp16 path's epilogue; optional raw accumulators (parity tests) and the int8
// twin for quantised consumers (the SAME fp16 values that go to `out`, quantised with the consumers' scale).
template <int PF>
__device__ __forceinline__ void conv_i8_epilogue(const ConvKArgs& a, const i32x16_t (&acc)[PF], const int (&opix)[PF], int cfi,
                                                 int lane, const float (&bias)[16], const float (&qs)[16]) {
    const int kh = lane >> 5;
    const float ralpha = (a.res != nullptr && a.res_alpha != nullptr) ? *a.res_alpha : 1.f;
    const int cfrag = cfi * 32;
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const bool pvalid = opix[pf] >= 0;
        const size_t prow = pvalid ? (size_t)opix[pf] : 0;
        if (a.acc_out != nullptr && pvalid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cfrag + 8 * (r >> 2) + 4 * kh + (r & 3);
                if (c < a.Cout) a.acc_out[prow * a.Cout + c] = acc[pf][r];
            }
        }
        f32x16_t accf;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = (float)acc[pf][r] * qs[r];
            asm volatile("" : "+v"(v));     // keep the product a separate rounding (no fma with the bias add)
            accf[r] = v;
        }
        const __half* rrow = (a.res && pvalid) ? a.res + prow * a.res_cs + a.res_co : nullptr;
        float v[16];
        finish16_any(a, accf, bias, cfrag, kh, a.Cout, rrow, ralpha, v);
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
        if (a.qout != nullptr) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned q = q8_quad(pk[g][0], pk[g][1], a.qo_inv2, a.qo_lo2, a.qo_hi2);
                const int c0 = cfrag + 8 * g + 4 * kh;
                if (pvalid && c0 + 3 < a.Cout) *reinterpret_cast<unsigned*>(a.qout + prow * a.qout_cs + a.qout_co + c0) = q;
            }
        }
        if (a.out == nullptr) continue;
        __half* orow = a.out + prow * a.out_cs + a.out_co;
        if (a.vec16_ok && (cfrag + 32) <= a.Cout) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
                if (pvalid) *reinterpret_cast<uint4*>(orow + cfrag + 16 * gp + 8 * kh) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        } else if (pvalid) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = cfrag + 8 * g + 4 * kh;
                if (c0 >= a.Cout) continue;
                if (a.vec_ok && (c0 + 3) < a.Cout) {
                    *reinterpret_cast<uint2*>(orow + c0) = make_uint2(pk[g][0], pk[g][1]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j < a.Cout) orow[c0 + j] = __float2half(v[g * 4 + j]);
                }
            }
        }
    }
}

struct VariantCfg {
    int cf, pf, persist;
    const char* name;
    int nw = 4;      // waves per block (pipe / dma kernels: 4 or 8)
    int st = 1;      // pipe kernels: the stride they are built for
    int depth = 2;   // pipe kernels: chunks the halo fill runs ahead; dma kernels: LDS stages
    int hc = 16;     // dma kernels: input channels per chunk
    int cs = 1;      // dma kernels: the conv stride they are built for
    int wres = 0;    // dma kernels: 1 = the block's tap images stay in LDS for all its work items (Cin <= 64)
    int i8only = 0;  // a tile geometry only the int8 path uses (y6_conv_i8): never offered to fp16 convs
};

struct Launch {
    ConvKArgs k;
    int grid;
    size_t lds;
};

// Resident grid of a persistent conv kernel: CUs of the current device x blocks of `threads` threads with `lds` bytes of dynamic LDS
// that fit a CU.  The occupancy query costs tens of microseconds of host time: its answers are kept per (device, LDS size) - one table
// per kernel instantiation (`cache` is a function-local static of the launcher template), behind a mutex (the single-entry
// cache it replaces was neither per-device nor thread-safe, and re-queried whenever consecutive layers differed in LDS size).
struct OccupancyCache {
    std::mutex mu;
    std::map<std::pair<int, size_t>, int> grid;   // (device, lds) -> n_cu * blocks per CU
    bool big_lds[64] = {};                        // per device: hipFuncAttributeMaxDynamicSharedMemorySize raised
};
template <class K>
int resident_grid(OccupancyCache& cache, K kern, int threads, size_t lds, size_t lds_limit, int* grid_out) {
    int dev = 0;
    Y6_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(cache.mu);
    if (lds > 64 * 1024 && dev >= 0 && dev < 64 && !cache.big_lds[dev]) {
        Y6_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        cache.big_lds[dev] = true;
    }
    auto it = cache.grid.find({dev, lds});
    if (it == cache.grid.end()) {
        int bpc = 0, n_cu = 0;
        Y6_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, threads, lds));
        Y6_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        it = cache.grid.emplace(std::make_pair(dev, lds), n_cu * (bpc < 1 ? 1 : bpc)).first;
    }
    *grid_out = it->second;
    return Y6_OK;
}

}  // namespace
