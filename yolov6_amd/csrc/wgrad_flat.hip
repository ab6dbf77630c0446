// wgrad_flat.hip - weight gradient of 3x3 stride-1 (pad 1) and 1x1 stride-1 convs as a block-tiled GEMM over a FLAT pixel index,
// read straight from the NHWC tensors (round 6; replaces the plane-fed kernel + its operand transposes on every map narrower than
// the row-ring kernel's 64 columns, and the row-ring kernel itself wherever the flat stages fit the LDS).
//
//   dW[m][n][ky][kx] += sum_{b,y,x} dy(b,y,x,m) * X(b, y+ky-1, x+kx-1, n)
//
// Replaces the weight half of autograd's conv backward (reference: the autograd graph of ConvModule / RepVGGBlock forward,
// yolov6/layers/common.py:45-49, :250-255; core/engine.py:173 `.backward()`).
//
// Why flat.  The GEMM's reduction runs over pixels.  The row-ring kernel walks it a map row at a time (one barrier per row): a 40- or
// 20-wide row is 3 or 2 k-steps, so the narrow maps - 2/3 of the weight-gradient FLOPs of YOLOv6-S - fell to the LDS-free plane-fed
// kernel (232 TFLOP/s, 0.093 of the MFMA peak).  Here the pixels of the WHOLE BATCH form one sequence with ONE shared
// zero column behind every row and ONE shared zero row behind every image:
//     Wp = W + 1,  Pp = (H + 1) * Wp
//     dy index  Q(b, y, x)  = b*Pp + y*Wp + x                      (x = W and y = H are the zero cells)
//     X  index  P(b, r, c)  = b*Pp + (r + 1)*Wp + (c + 1)          (column 0 of a row = column Wp of the row above: the shared pad)
// so that tap (ky, kx) of output position Q reads X at Q + ky*Wp + kx - a CONSTANT offset, also across row and image boundaries,
// where it meets a zero cell of one operand or the other.  The reduction is then cut into chunks of KC = 128 flat positions,
// wherever they fall: one barrier per 72 MFMAs per wave instead of one per row, no row-end masks, K efficiency H*W / Pp
// (0.907 at 20x20, 0.952 at 40x40).  1x1 convs use the plain pixel index (no padding).
//
// Block = TM x 64 (cout, cin) x all taps, TM = 128 (8 waves) or 64 (4 waves, two blocks per CU); a wave owns a 32 x 32 tile and all
// nine taps (144 accumulator registers).  A chunk's operands arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction,
// every request touching whole cache lines) as [32-channel chunk][position][32 channels] images - dy: KC positions, X: KC + 2*Wp + 24
// (the halo of the +-1 row taps and the neighbour runs of the +-1 column taps) - into one of two stages; the next chunk's requests
// are in flight while this one is multiplied.  Pad cells and everything outside the tensors are zero-filled by the buffer
// descriptor's range check.  Operands leave LDS through gfx950's transposing read (ds_read_b64_tr_b16: a lane receives 8 consecutive
// positions of one channel = one MFMA operand); lanes 0-31 walk the first half of the chunk and lanes 32-63 the second, so a lane's
// previous / next run - which the kx = 0 / 2 taps shift in by one element (v_alignbit) - is its own previous / next k-step.
// Slices of the flat range are summed by one small kernel in a fixed order (deterministic, no float atomics).
#include <cstdlib>

#include "common.hpp"
#include "plan_internal.hpp"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

constexpr int KC = 128;                    // flat positions per chunk (8 k-steps of 16)
constexpr int TN = 64;                     // cin per block
constexpr unsigned kOob = 0xf0000000u;     // voffset of a piece that must read zeros (tensors stay below 3.75 GiB)

struct WgFArgs {
    const __half* a;
    const __half* x;
    unsigned a_bytes, x_bytes;
    int a_cs, a_co, a_C;       // dy view: pixel pitch, channel offset (halves), readable channels (multiple of 8)
    int x_cs, x_co, x_C;
    int B, H, W;               // x (input) extent
    int Ho, Wo, stride;        // output grid; stride 1 or 2
    int dyd;                   // 2: dy is stored zero-inserted at (2y, 2x) of an [B, H, W] buffer (the data gradient's operand)
    int M, N;
    int Wp, Pp;                // padded row / plane of the flat index over the OUTPUT grid (1x1: Wo, Ho*Wo)
    float inv_Pp, inv_Wp;
    int total;                 // B * Pp
    int xpos;                  // positions of an X stage image (multiple of 16)
    int mt2, nt2, nsplit, chunks_per, nchunks;
    float* ws;
};

__device__ __forceinline__ h8_t as_h8(const u32x4 v) { return __builtin_bit_cast(h8_t, v); }

__device__ __forceinline__ i32x4_t make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)base;
    i32x4_t r;
    r[0] = (int)(unsigned)(p & 0xffffffffu);
    r[1] = (int)(unsigned)((p >> 32) & 0xffffu);   // stride 0: raw buffer, byte offsets, range check against num_records
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
}
// one LDS-DMA piece: lane i writes 16 B to lds_dst + 16*i from rsrc.base + voff(lane); out of range -> zeros
__device__ __forceinline__ void dma16(const i32x4_t& rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
// 8 consecutive positions of one channel: two transpose-reads 256 bytes (4 positions) apart
__device__ __forceinline__ u32x4 tr_run(const char* p) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 256));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 r;
    r[0] = l2[0], r[1] = l2[1], r[2] = h2[0], r[3] = h2[1];
    return r;
}
// [prev[7], cur[0..6]]  (element e of the result = element e-1 of the run sequence)
__device__ __forceinline__ u32x4 shift_m1(const u32x4 prev, const u32x4 cur) {
    u32x4 o;
    o[0] = __builtin_amdgcn_alignbit(cur[0], prev[3], 16);
    o[1] = __builtin_amdgcn_alignbit(cur[1], cur[0], 16);
    o[2] = __builtin_amdgcn_alignbit(cur[2], cur[1], 16);
    o[3] = __builtin_amdgcn_alignbit(cur[3], cur[2], 16);
    return o;
}
// [cur[1..7], next[0]]
__device__ __forceinline__ u32x4 shift_p1(const u32x4 cur, const u32x4 next) {
    u32x4 o;
    o[0] = __builtin_amdgcn_alignbit(cur[1], cur[0], 16);
    o[1] = __builtin_amdgcn_alignbit(cur[2], cur[1], 16);
    o[2] = __builtin_amdgcn_alignbit(cur[3], cur[2], 16);
    o[3] = __builtin_amdgcn_alignbit(next[0], cur[3], 16);
    return o;
}
// n / d for 0 <= n < 2^24 (exact: the float quotient is off by at most one)
__device__ __forceinline__ int div_small(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

template <int KS, int NWM>
__global__ __launch_bounds__(NWM * 128, 2) void wgrad_flat_kernel(const WgFArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = KS * KS;
    constexpr int NWV = NWM * 2;                       // waves per block: NWM along cout x 2 along cin
    constexpr int HALO = KS == 3 ? 8 : 0;              // positions in front of the chunk in an X image (the previous run)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware placement (block b runs on XCD b % 8): all tiles of one slice of the flat range read the same positions of dy and X,
    // so a slice's tiles sit on one XCD (one L2), consecutively; slices are dealt round-robin to the XCDs.
    const int tiles = a.mt2 * a.nt2;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int ks = (bi / tiles) * 8 + xcd, slot = bi % tiles;
    if (ks >= a.nsplit) return;
    const int n2 = slot % a.nt2, m2 = slot / a.nt2;
    const int mtl = wave % NWM, ntl = wave / NWM;
    const int c0 = ks * a.chunks_per;
    const int c1 = c0 + a.chunks_per < a.nchunks ? c0 + a.chunks_per : a.nchunks;

    const int a_img = KC * 64;                         // bytes of one 32-channel dy image of a stage
    const int x_img = a.xpos * 64;
    const int stage_bytes = NWM * a_img + 2 * x_img;
    const i32x4_t rsA = make_rsrc(a.a, a.a_bytes), rsX = make_rsrc(a.x, a.x_bytes);

    // ---- requests of one chunk: instruction i of the list [dy images | X images], 16 positions x 4 pieces of 8 channels each
    const int ia = KC / 16, ix = a.xpos / 16;
    const int ni = NWM * ia + 2 * ix;
    // (one request instruction; i = wave, wave + NWV, ... are this wave's)
    auto issue_piece = [&](int chunk, unsigned lds_stage, int i) {
        const int q0 = chunk * KC;
        {
            const bool isx = i >= NWM * ia;
            const int ii = isx ? i - NWM * ia : i;
            const int per = isx ? ix : ia;
            const int img = ii / per, j = ii - img * per;          // 32-channel image, instruction inside it (wave-uniform)
            const int p = j * 16 + (lane >> 2);                    // position inside the image
            unsigned voff = kOob;
            if (!isx) {
                const int ch = m2 * (NWM * 32) + img * 32 + (lane & 3) * 8;
                const int Q = q0 + p;
                if (KS == 1) {
                    if (Q < a.total && ch + 8 <= a.a_C) {
                        if (a.dyd == 1) {
                            voff = (unsigned)(((unsigned long long)Q * a.a_cs + a.a_co + ch) * 2ull);
                        } else {
                            const int b = div_small(Q, a.Pp, a.inv_Pp);
                            const int rem = Q - b * a.Pp;
                            const int y = div_small(rem, a.Wp, a.inv_Wp);
                            const int xx = rem - y * a.Wp;
                            voff = (unsigned)(((((unsigned long long)b * a.H + 2 * y) * a.W + 2 * xx) * a.a_cs + a.a_co + ch) * 2ull);
                        }
                    }
                } else if (Q < a.total && ch + 8 <= a.a_C) {
                    const int b = div_small(Q, a.Pp, a.inv_Pp);
                    const int rem = Q - b * a.Pp;
                    const int y = div_small(rem, a.Wp, a.inv_Wp);
                    const int xx = rem - y * a.Wp;
                    if (y < a.H && xx < a.W)
                        voff = (unsigned)(((((unsigned long long)b * a.H + y) * a.W + xx) * a.a_cs + a.a_co + ch) * 2ull);
                }
                dma16(rsA, voff, lds_stage + (unsigned)(img * a_img + j * 1024));
            } else {
                const int ch = n2 * TN + img * 32 + (lane & 3) * 8;
                const int P = q0 - HALO + p;
                if (KS == 1) {
                    if (P < a.total && ch + 8 <= a.x_C) {
                        if (a.stride == 1) {
                            voff = (unsigned)(((unsigned long long)P * a.x_cs + a.x_co + ch) * 2ull);
                        } else {                                   // 1x1 stride 2: output (y, x) reads input (2y, 2x)
                            const int b = div_small(P, a.Pp, a.inv_Pp);
                            const int rem = P - b * a.Pp;
                            const int y = div_small(rem, a.Wp, a.inv_Wp);
                            const int xx = rem - y * a.Wp;
                            voff = (unsigned)(((((unsigned long long)b * a.H + 2 * y) * a.W + 2 * xx) * a.x_cs + a.x_co + ch) * 2ull);
                        }
                    }
                } else if (P >= 0 && P < a.total && ch + 8 <= a.x_C) {
                    const int b = div_small(P, a.Pp, a.inv_Pp);
                    const int rem = P - b * a.Pp;
                    const int rr = div_small(rem, a.Wp, a.inv_Wp);
                    const int cc = rem - rr * a.Wp;
                    if (rr >= 1 && cc >= 1)                   // (rr <= H and cc <= W by construction)
                        voff = (unsigned)(((((unsigned long long)b * a.H + (rr - 1)) * a.W + (cc - 1)) * a.x_cs + a.x_co + ch) * 2ull);
                }
                dma16(rsX, voff, lds_stage + (unsigned)(NWM * a_img + img * x_img + j * 1024));
            }
        }
    };
    auto issue_chunk = [&](int chunk, unsigned lds_stage) {
        for (int i = wave; i < ni; i += NWV) issue_piece(chunk, lds_stage, i);
    };

    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    // lane part of a transpose-read address: position (lane&15)>>2 of the group's four, channels 4*(lane&3) + 16*((lane>>4)&1) of
    // the 32-channel image, and this half's first run
    const int lane_off = ((lane & 15) >> 2) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2 + half * (KC / 2) * 64;
    const unsigned lds0 = lds_addr(smem);

    if (c0 < c1) issue_chunk(c0, lds0);
    for (int c = c0; c < c1; ++c) {
        const int st = (c - c0) & 1;
        // chunk c has landed for this wave ... and for every wave, and everyone has left chunk c-1 (whose stage is refilled next)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // The next chunk's requests go out ONE PER K-STEP, between the MFMAs (their address arithmetic - two divisions per piece -
        // runs in the shadow of the matrix pipe; issued in a burst behind the barrier it kept every wave of the block off the
        // matrix pipe for ~500 cycles per chunk); what the k-steps do not cover follows the last one.
        const bool more = c + 1 < c1;
        const unsigned nst = lds0 + (unsigned)((st ^ 1) * stage_bytes);
        const char* const sb = smem + st * stage_bytes;
        const char* const ab = sb + mtl * a_img + lane_off;
        if constexpr (KS == 3) {
            const char* xb[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) xb[r] = sb + NWM * a_img + ntl * x_img + lane_off + (HALO + r * a.Wp + 1) * 64;
            // three run sets rotate through the roles (previous, current, next); fully unrolled, so the rotation is static
            u32x4 R[3][3], A[2];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                R[0][r] = tr_run(xb[r] - 8 * 64);
                R[1][r] = tr_run(xb[r]);
            }
            A[0] = tr_run(ab);
#pragma unroll
            for (int kk = 0; kk < KC / 16; ++kk) {
                const int sp = kk % 3, sc = (kk + 1) % 3, sn = (kk + 2) % 3;
                if (more && wave + kk * NWV < ni) issue_piece(c + 1, nst, wave + kk * NWV);
#pragma unroll
                for (int r = 0; r < 3; ++r) R[sn][r] = tr_run(xb[r] + (kk + 1) * 8 * 64);
                if (kk + 1 < KC / 16) A[(kk + 1) & 1] = tr_run(ab + (kk + 1) * 8 * 64);
                __builtin_amdgcn_sched_barrier(0);       // the reads of step kk+1 go out BEFORE the MFMAs of step kk
                const h8_t af = as_h8(A[kk & 1]);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc[r * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(shift_m1(R[sp][r], R[sc][r])), acc[r * 3 + 0], 0, 0, 0);
                    acc[r * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(R[sc][r]), acc[r * 3 + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);       // six MFMAs that do not need the new runs cover the LDS latency
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    acc[r * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(shift_p1(R[sc][r], R[sn][r])), acc[r * 3 + 2], 0, 0, 0);
            }
        } else {
            const char* const xb = sb + NWM * a_img + ntl * x_img + lane_off;
            u32x4 A[2], Bq[2];
            A[0] = tr_run(ab);
            Bq[0] = tr_run(xb);
#pragma unroll
            for (int kk = 0; kk < KC / 16; ++kk) {
                if (more && wave + kk * NWV < ni) issue_piece(c + 1, nst, wave + kk * NWV);
                if (kk + 1 < KC / 16) {
                    A[(kk + 1) & 1] = tr_run(ab + (kk + 1) * 8 * 64);
                    Bq[(kk + 1) & 1] = tr_run(xb + (kk + 1) * 8 * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[kk & 1]), as_h8(Bq[kk & 1]), acc[0], 0, 0, 0);
            }
        }
        if (more)
            for (int i = wave + (KC / 16) * NWV; i < ni; i += NWV) issue_piece(c + 1, nst, i);
    }
    // C/D layout: column n = lane & 31, row m = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5).  The slice's partial tile goes to the
    // workspace with plain stores (n contiguous across lanes); wgrad_flat_reduce_kernel sums the slices in a fixed order.
    const int n_out = n2 * TN + ntl * 32 + l31;
    if (n_out >= a.N) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m_out = m2 * (NWM * 32) + mtl * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
            if (m_out < a.M) a.ws[(((size_t)ks * NT + t) * a.M + m_out) * a.N + n_out] = acc[t][q];
        }
}

// ---- 3x3 stride 2 (pad 1): dW[m][n][ky][kx] += sum dy(b,y,x,m) * X(b, 2y+ky-1, 2x+kx-1, n) ------------------------------------
// The input splits into four row/column parity planes X_pq(r, c) = X(2r+p, 2c+q), each laid out over the OUTPUT grid's padded
// flat index exactly as the stride-1 X image (cell b*Pp + (r+1)*Wp + (c+1)).  Tap (ky, kx) then reads ONE plane at ONE constant
// offset from the output position Q:   ky: 0 -> odd rows, r = y-1 (+0) | 1 -> even rows, r = y (+Wp) | 2 -> odd rows, r = y (+Wp);
// kx likewise with +0 / +1 / +1.  No shifted runs: every tap's operand is a plain transpose-read at (position + offset).  A stage
// holds NWM dy images and the four plane images of 32 input channels (the LDS-DMA source address samples every other pixel and
// row); it is single-buffered and two blocks share a CU: one multiplies while the other's requests are in flight.
template <int NWM, int KCS>
__global__ __launch_bounds__(NWM * 64, 2) void wgrad_flat_s2_kernel(const WgFArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = lane >> 5, l31 = lane & 31;
    const int tiles = a.mt2 * a.nt2;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int ks = (bi / tiles) * 8 + xcd, slot = bi % tiles;
    if (ks >= a.nsplit) return;
    const int n2 = slot % a.nt2, m2 = slot / a.nt2;
    const int c0 = ks * a.chunks_per;
    const int c1 = c0 + a.chunks_per < a.nchunks ? c0 + a.chunks_per : a.nchunks;
    const int a_img = KCS * 64, x_img = a.xpos * 64;
    const i32x4_t rsA = make_rsrc(a.a, a.a_bytes), rsX = make_rsrc(a.x, a.x_bytes);
    const int ia = KCS / 16, ix = a.xpos / 16;
    const int ni = NWM * ia + 4 * ix;
    const unsigned lds0 = lds_addr(smem);
    auto issue_chunk = [&](int chunk) {
        const int q0 = chunk * KCS;
        for (int i = wave; i < ni; i += NWM) {
            const bool isx = i >= NWM * ia;
            const int ii = isx ? i - NWM * ia : i;
            const int per = isx ? ix : ia;
            const int img = ii / per, j = ii - img * per;
            const int P = q0 + j * 16 + (lane >> 2);
            unsigned voff = kOob;
            const int chl = (lane & 3) * 8;
            if (P < a.total) {
                const int b = div_small(P, a.Pp, a.inv_Pp);
                const int rem = P - b * a.Pp;
                const int rr = div_small(rem, a.Wp, a.inv_Wp);
                const int cc = rem - rr * a.Wp;
                if (!isx) {
                    const int ch = m2 * (NWM * 32) + img * 32 + chl;
                    if (rr < a.Ho && cc < a.Wo && ch + 8 <= a.a_C)
                        voff = a.dyd == 1 ? (unsigned)(((((unsigned long long)b * a.Ho + rr) * a.Wo + cc) * a.a_cs + a.a_co + ch) * 2ull)
                                          : (unsigned)(((((unsigned long long)b * a.H + 2 * rr) * a.W + 2 * cc) * a.a_cs + a.a_co + ch) * 2ull);
                } else {
                    const int ch = n2 * 32 + chl;
                    const int sr = 2 * (rr - 1) + (img >> 1), sc = 2 * (cc - 1) + (img & 1);
                    if (rr >= 1 && cc >= 1 && sr < a.H && sc < a.W && ch + 8 <= a.x_C)
                        voff = (unsigned)(((((unsigned long long)b * a.H + sr) * a.W + sc) * a.x_cs + a.x_co + ch) * 2ull);
                }
            }
            if (!isx) dma16(rsA, voff, lds0 + (unsigned)(img * a_img + j * 1024));
            else dma16(rsX, voff, lds0 + (unsigned)(NWM * a_img + img * x_img + j * 1024));
        }
    };
    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    const int lane_off = ((lane & 15) >> 2) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2 + half * (KCS / 2) * 64;
    const char* const ab = smem + wave * a_img + lane_off;
    const char* xt[9];                                   // per tap: plane image + constant offset
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t % 3;
        const int pl = (ky == 1 ? 0 : 2) + (kx == 1 ? 0 : 1);
        const int off = (ky == 0 ? 0 : a.Wp) + (kx == 0 ? 0 : 1);
        xt[t] = smem + NWM * a_img + pl * x_img + lane_off + off * 64;
    }
    for (int c = c0; c < c1; ++c) {
        issue_chunk(c);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");          // the chunk has landed for every wave
#pragma unroll
        for (int kk = 0; kk < KCS / 16; ++kk) {
            const h8_t af = as_h8(tr_run(ab + kk * 8 * 64));
            u32x4 Bq[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) Bq[t] = tr_run(xt[t] + kk * 8 * 64);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(Bq[t]), acc[t], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // everyone has read it: the stage may be refilled
    }
    const int n_out = n2 * 32 + l31;
    if (n_out >= a.N) return;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m_out = m2 * (NWM * 32) + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
            if (m_out < a.M) a.ws[(((size_t)ks * 9 + t) * a.M + m_out) * a.N + n_out] = acc[t][q];
        }
}

// out[m*sm + n*sn + t*st] += sum over the slices, in slice order (deterministic).  A thread owns four consecutive (m, n) cells of
// one tap (the [t][m][n] partials are contiguous over (m, n): a wave reads 1 KiB of every slice, sixteen 16-byte loads in flight per
// lane), and adds its four sums into the OIHW gradient.  What bounds this kernel is bytes in flight, not bytes: 128 slices of a
// 128 x 128 x 9 tile are 75 MB behind only 37 k threads (first forms, round 6: one element per thread in [t][m][n] order, then one
// cout row x 64 cin per block - 19-35 us per call, 2.1 ms of a training step).
__global__ __launch_bounds__(256) void wgrad_flat_reduce_kernel(const float* __restrict__ ws, int nsplit, int T, int M, int N, float* __restrict__ out,
                                                                int sm, int sn, int st) {
    const size_t quads = (size_t)T * M * N / 4;          // N % 8 == 0 (host): a quad never straddles two cout rows
    const size_t per4 = quads;
    const float4* p4 = reinterpret_cast<const float4*>(ws);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 16 <= nsplit; k += 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p4[(size_t)(k + u) * per4 + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) {               // slice order
                s.x += v[u].x;
                s.y += v[u].y;
                s.z += v[u].z;
                s.w += v[u].w;
            }
        }
        for (; k < nsplit; ++k) {
            const float4 v = p4[(size_t)k * per4 + i];
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        const size_t e = i * 4;
        const int n = (int)(e % N);
        const int m = (int)((e / N) % M);
        const int t = (int)(e / ((size_t)N * M));
        float* o = out + (size_t)m * sm + (size_t)n * sn + (size_t)t * st;
        o[0] += s.x;
        o[(size_t)sn] += s.y;
        o[(size_t)2 * sn] += s.z;
        o[(size_t)3 * sn] += s.w;
    }
}

// The same sum for the nine-tap layers whose gradient is OIHW-contiguous (sn == 9, st == 1): a block owns one cout row x 256 cins x
// all nine taps - wave t sums tap t, a lane four consecutive cins (the same coalesced 1 KiB per slice and wave as above) - and the
// 2 304 sums meet in LDS in the gradient's own [cin][tap] order, so the read-modify-write of the gradient is 576 contiguous 16-byte
// pieces instead of four 4-byte cells 36 bytes apart per thread (every 128-byte line of the gradient was touched by nine different
// waves).  The additions are the ones of the kernel above, in the same order: the same bits.
__global__ __launch_bounds__(576) void wgrad_flat_reduce9_kernel(const float* __restrict__ ws, int nsplit, int M, int N, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_t[256 * 9];
    const int t = threadIdx.x >> 6, j = threadIdx.x & 63;
    const int nb = (N + 255) >> 8;
    const int m = blockIdx.x / nb, n0 = (blockIdx.x - m * nb) * 256;
    const int n = n0 + 4 * j;
    const size_t per4 = (size_t)9 * M * N / 4;
    if (n < N) {
        const float4* p4 = reinterpret_cast<const float4*>(ws) + (((size_t)t * M + m) * N + n) / 4;
        float4 s = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 16 <= nsplit; k += 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p4[(size_t)(k + u) * per4];
#pragma unroll
            for (int u = 0; u < 16; ++u) {               // slice order
                s.x += v[u].x;
                s.y += v[u].y;
                s.z += v[u].z;
                s.w += v[u].w;
            }
        }
        for (; k < nsplit; ++k) {
            const float4 v = p4[(size_t)k * per4];
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        s_t[(4 * j + 0) * 9 + t] = s.x;
        s_t[(4 * j + 1) * 9 + t] = s.y;
        s_t[(4 * j + 2) * 9 + t] = s.z;
        s_t[(4 * j + 3) * 9 + t] = s.w;
    }
    __syncthreads();
    const int nn = N - n0 < 256 ? N - n0 : 256;            // cins of this block: nn * 9 floats, nn % 4 == 0 -> whole 16-byte pieces
    const int q = threadIdx.x;
    if (q * 4 < nn * 9) {
        float4* o = reinterpret_cast<float4*>(out + ((size_t)m * N + n0) * 9) + q;
        const float4 a = *o, b = *reinterpret_cast<const float4*>(s_t + 4 * q);
        *o = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

struct FlatGeom {
    int Wp, Pp, total, xpos, nwm, nchunks, stride, kc, tn;
    size_t lds;
};

bool flat_geom(const y6_wgrad_nhwc_desc* d, FlatGeom* g) {
    const int KS = d->ksize;
    const int stride = d->stride == 2 ? 2 : 1;
    const long Ho = stride == 2 ? (d->x.H - 1) / 2 + 1 : d->x.H, Wo = stride == 2 ? (d->x.W - 1) / 2 + 1 : d->x.W, B = d->x.B;
    g->stride = stride;
    g->Wp = KS == 3 ? (int)Wo + 1 : (int)Wo;
    g->Pp = KS == 3 ? (int)((Ho + 1) * (Wo + 1)) : (int)(Ho * Wo);
    const long total = B * (long)g->Pp;
    if (total >= (1l << 24)) return false;             // div_small's exact range
    g->total = (int)total;
    g->nwm = d->M > 64 ? 4 : 2;
    if (KS == 3 && stride == 2) {
        if (Wo > 48) return false;                     // (80-wide outputs: 216-483 us against the plane-fed kernel's 135-260, profiles/r06)
        // single stage, two blocks per CU: <= 80 KiB each; 128-position chunks where the four plane images fit, else 64
        g->tn = 32;
        // (a 64-position chunk fits the 80-wide outputs too, but it is 36 MFMAs per 56 request instructions and the zero-inserted dy
        // wastes half of every line: 696 us against the plane-fed kernel's 260 on 64 -> 128 @160 -> 80, profiles/r06 - 128 only)
        for (int kc : {128}) {
            g->kc = kc;
            g->xpos = (kc + g->Wp + 1 + 15) / 16 * 16;
            g->lds = (size_t)g->nwm * kc * 64 + 4 * (size_t)g->xpos * 64;
            if (g->lds <= 80u * 1024) break;
        }
        g->nchunks = (int)((total + g->kc - 1) / g->kc);
        return g->lds <= 80u * 1024;
    }
    g->tn = TN;
    g->kc = KC;
    const int xp = KS == 3 ? KC + 2 * g->Wp + 24 : KC;
    g->xpos = (xp + 15) / 16 * 16;
    g->lds = 2 * ((size_t)g->nwm * KC * 64 + 2 * (size_t)g->xpos * 64);
    g->nchunks = (int)((total + KC - 1) / KC);
    return g->lds <= 160u * 1024;                      // (4-wave blocks: two per CU while their stages stay below 80 KiB)
}

// slices of the flat range: one round of blocks over the chip (256 CUs x 1 block of 8 waves or 2 blocks of 4), a slice no shorter
// than 4 chunks, as many as the workspace holds partial tile sets for
void flat_split(const y6_wgrad_nhwc_desc* d, const FlatGeom& g, long tiles, int* nsplit_out, int* chunks_per_out) {
    const bool s2k3 = d->ksize == 3 && g.stride == 2;
    const size_t per = (size_t)d->ksize * d->ksize * d->M * d->N;
    // Y6_WGRAD_SLOTS_PCT=p (A/B; default 100): p % of a round of blocks - that share of the partial volume the slice sum moves, and the
    // rest of the chip left to the main stream's kernels while the GEMM runs longer on the side stream.  Measured on the YOLOv6-S
    // b64 step [GPU r06zc / r06zd, two visits, alternating runs]: 100 % 30.85 / 31.2 ms (weight gradients 7.9 ms by themselves),
    // 75 % 31.0 (8.6), 50 % 30.50 / 30.9 (9.95), 38 % 31.6 (11.7), 25 % 33.2 (14.8): 0.2-0.3 ms for 50 %, inside the run-to-run
    // spread of the second visit - the default stays a full round (and the GEMM's own rate the one the tables quote).
    static const int slots_pct = getenv("Y6_WGRAD_SLOTS_PCT") ? (atoi(getenv("Y6_WGRAD_SLOTS_PCT")) > 0 ? atoi(getenv("Y6_WGRAD_SLOTS_PCT")) : 100) : 100;
    long slots = (long)(s2k3 ? 512 : ((g.nwm == 4 || g.lds > 80u * 1024) ? 256 : 512)) * slots_pct / 100;
    if (slots < tiles) slots = tiles;
    long nsplit = slots / tiles;
    if (nsplit > g.nchunks / 4) nsplit = g.nchunks / 4;
    const long max_by_ws = (long)(d->workspace_bytes / (per * sizeof(float)));
    if (nsplit > max_by_ws) nsplit = max_by_ws;
    if (nsplit < 1) nsplit = 1;
    *chunks_per_out = (int)((g.nchunks + nsplit - 1) / nsplit);
    *nsplit_out = (int)((g.nchunks + *chunks_per_out - 1) / *chunks_per_out);
}

}  // namespace

bool wgrad_nhwc_view_ok(const y6_tensor& t);          // wgrad.hip

const char* wgrad_flat_unsupported(const y6_wgrad_nhwc_desc* d) {
    if (!d || !d->out) return "null argument";
    if (d->ksize != 1 && d->ksize != 3) return "ksize must be 1 or 3 (stride 1)";
    if (!wgrad_nhwc_view_ok(d->dy) || !wgrad_nhwc_view_ok(d->x)) return "views must be fp16 NHWC, 8-channel / 16-byte aligned, below 3.75 GiB";
    if (d->stride == 2) {
        const bool compact = d->dy.H == (d->x.H - 1) / 2 + 1 && d->dy.W == (d->x.W - 1) / 2 + 1;
        const bool dilated = d->dy.H == d->x.H && d->dy.W == d->x.W && d->x.H > 1 && d->x.W > 1;
        if (d->dy.B != d->x.B || !(compact || dilated) || d->x.B < 1 || d->x.H < 1 || d->x.W < 1)
            return "stride 2: dy must be [B, (H-1)/2+1, (W-1)/2+1, .] or the zero-inserted [B, H, W, .] for x [B, H, W, .]";
    } else if (d->dy.B != d->x.B || d->dy.H != d->x.H || d->dy.W != d->x.W || d->x.B < 1 || d->x.H < 1 || d->x.W < 1) {
        return "dy and x must have one spatial shape";
    }
    if (d->M < 1 || d->N < 1 || d->dy.C < d->M || d->x.C < d->N) return "views narrower than M / N";
    if (d->N % 4 != 0 || (((uintptr_t)d->workspace) & 15) != 0) return "N must be a multiple of 4 and the workspace 16-byte aligned (the slice sum reads float4)";
    if (!d->workspace || d->workspace_bytes < (size_t)d->ksize * d->ksize * d->M * d->N * sizeof(float)) return "workspace missing or smaller than one partial tile set";
    FlatGeom g;
    if (!flat_geom(d, &g)) return "map too wide (or batch too large) for the flat stages";
    return nullptr;
}

int wgrad_flat_launch(const y6_wgrad_nhwc_desc* d, hipStream_t s) {
    const char* why = wgrad_flat_unsupported(d);
    Y6_REQUIRE(why == nullptr, "wgrad_flat: %s", why ? why : "");
    FlatGeom g;
    flat_geom(d, &g);
    WgFArgs a;
    memset(&a, 0, sizeof(a));
    a.a = (const __half*)d->dy.data;
    a.x = (const __half*)d->x.data;
    a.a_bytes = (unsigned)((size_t)d->dy.B * d->dy.H * d->dy.W * d->dy.cstride * 2);
    a.x_bytes = (unsigned)((size_t)d->x.B * d->x.H * d->x.W * d->x.cstride * 2);
    a.a_cs = d->dy.cstride, a.a_co = d->dy.coff, a.a_C = d->dy.C;
    a.x_cs = d->x.cstride, a.x_co = d->x.coff, a.x_C = d->x.C;
    a.B = d->x.B, a.H = d->x.H, a.W = d->x.W;
    a.stride = g.stride;
    a.Ho = g.stride == 2 ? (d->x.H - 1) / 2 + 1 : d->x.H, a.Wo = g.stride == 2 ? (d->x.W - 1) / 2 + 1 : d->x.W;
    a.dyd = (g.stride == 2 && d->dy.H == d->x.H && d->dy.W == d->x.W && d->x.H > 1) ? 2 : 1;
    a.M = d->M, a.N = d->N;
    a.Wp = g.Wp, a.Pp = g.Pp, a.total = g.total, a.xpos = g.xpos, a.nchunks = g.nchunks;
    a.inv_Pp = 1.0f / (float)g.Pp, a.inv_Wp = 1.0f / (float)g.Wp;
    const int TM = g.nwm * 32;
    a.mt2 = y6_cdiv(d->M, TM);
    a.nt2 = y6_cdiv(d->N, g.tn);
    const int T = d->ksize * d->ksize;
    const long tiles = (long)a.mt2 * a.nt2;
    const size_t per = (size_t)T * d->M * d->N;
    Y6_REQUIRE(d->workspace && d->workspace_bytes >= per * sizeof(float), "wgrad_flat: workspace missing or too small");
    const bool s2k3 = d->ksize == 3 && g.stride == 2;
    flat_split(d, g, tiles, &a.nsplit, &a.chunks_per);
    a.ws = (float*)d->workspace;
    const unsigned grid = (unsigned)(8 * ((a.nsplit + 7) / 8) * tiles);
    static bool attr_set = false;
    if (!attr_set) {
        Y6_HIP(hipFuncSetAttribute((const void*)wgrad_flat_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        Y6_HIP(hipFuncSetAttribute((const void*)wgrad_flat_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        Y6_HIP(hipFuncSetAttribute((const void*)wgrad_flat_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        Y6_HIP(hipFuncSetAttribute((const void*)wgrad_flat_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        Y6_HIP(hipFuncSetAttribute((const void*)wgrad_flat_s2_kernel<4, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        Y6_HIP(hipFuncSetAttribute((const void*)wgrad_flat_s2_kernel<4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        Y6_HIP(hipFuncSetAttribute((const void*)wgrad_flat_s2_kernel<2, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        Y6_HIP(hipFuncSetAttribute((const void*)wgrad_flat_s2_kernel<2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr_set = true;
    }
    if (s2k3) {
        if (g.nwm == 4 && g.kc == 128) hipLaunchKernelGGL((wgrad_flat_s2_kernel<4, 128>), dim3(grid), dim3(256), g.lds, s, a);
        else if (g.nwm == 4) hipLaunchKernelGGL((wgrad_flat_s2_kernel<4, 64>), dim3(grid), dim3(256), g.lds, s, a);
        else if (g.kc == 128) hipLaunchKernelGGL((wgrad_flat_s2_kernel<2, 128>), dim3(grid), dim3(128), g.lds, s, a);
        else hipLaunchKernelGGL((wgrad_flat_s2_kernel<2, 64>), dim3(grid), dim3(128), g.lds, s, a);
    } else if (d->ksize == 3) {
        if (g.nwm == 4) hipLaunchKernelGGL((wgrad_flat_kernel<3, 4>), dim3(grid), dim3(512), g.lds, s, a);
        else hipLaunchKernelGGL((wgrad_flat_kernel<3, 2>), dim3(grid), dim3(256), g.lds, s, a);
    } else {
        if (g.nwm == 4) hipLaunchKernelGGL((wgrad_flat_kernel<1, 4>), dim3(grid), dim3(512), g.lds, s, a);
        else hipLaunchKernelGGL((wgrad_flat_kernel<1, 2>), dim3(grid), dim3(256), g.lds, s, a);
    }
    Y6_LAUNCH_CHECK();
    static const bool reduce9 = getenv("Y6_WGRAD_REDUCE9") ? atoi(getenv("Y6_WGRAD_REDUCE9")) != 0 : true;   // A/B switch
    if (reduce9 && T == 9 && d->sn == 9 && d->st == 1 && d->sm == (long)d->N * 9 && d->N % 4 == 0 && (((uintptr_t)d->out) & 15) == 0) {
        hipLaunchKernelGGL(wgrad_flat_reduce9_kernel, dim3((unsigned)(d->M * ((d->N + 255) / 256))), dim3(576), 0, s, a.ws, a.nsplit, d->M, d->N, d->out);
        Y6_LAUNCH_CHECK();
        return Y6_OK;
    }
    unsigned rg = (unsigned)((per / 4 + 255) / 256);
    if (rg > 8192) rg = 8192;
    hipLaunchKernelGGL(wgrad_flat_reduce_kernel, dim3(rg), dim3(256), 0, s, a.ws, a.nsplit, T, d->M, d->N, d->out, d->sm, d->sn, d->st);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// Host-only planning query (no launch, no pointer dereferenced): the flat kernel's geometry for a descriptor it takes.
extern "C" int y6_wgrad_flat_geometry(const y6_wgrad_nhwc_desc* d, y6_wgrad_flat_geom* out) {
    Y6_REQUIRE(d && out, "wgrad_flat_geometry: null argument");
    const char* why = wgrad_flat_unsupported(d);
    if (why) {
        y6_set_error("wgrad_flat_geometry: %s", why);
        return Y6_EUNSUPPORTED;
    }
    FlatGeom g;
    flat_geom(d, &g);
    memset(out, 0, sizeof(*out));
    out->row_pitch = g.Wp, out->plane = g.Pp, out->flat_positions = g.total;
    out->chunk = g.kc, out->chunks = g.nchunks;
    out->tile_m = g.nwm * 32, out->tile_n = g.tn;
    out->tiles = y6_cdiv(d->M, g.nwm * 32) * y6_cdiv(d->N, g.tn);
    out->x_positions = g.xpos;
    out->stages = (d->ksize == 3 && g.stride == 2) ? 1 : 2;
    out->lds_bytes = g.lds;
    flat_split(d, g, out->tiles, &out->slices, &out->chunks_per_slice);
    out->partial_bytes = (uint64_t)out->slices * d->ksize * d->ksize * d->M * d->N * sizeof(float);
    return Y6_OK;
}
