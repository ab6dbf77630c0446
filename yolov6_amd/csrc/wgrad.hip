// wgrad.hip — weight gradient of a convolution as a tap-table GEMM over pixels on the gfx950 matrix cores.
//
//   dW[m][n][t] += sum_{b, y, q}  A[m][b][y][q] * P_t[n][b][y + drow_t][q + shift_t]
//
// Replaces the weight half of autograd's conv backward for the training step (reference: the autograd graph of
// ConvModule / RepVGGBlock forward, yolov6/layers/common.py:45-49, :250-255; engine.py:173 `.backward()`).
//
// Design (MI355X-first, not a cuDNN wgrad port):
//  * The reduction runs over PIXELS, and NHWC keeps channels contiguous - the wrong way round for an MFMA operand
//    (a lane holds 8 consecutive k of one row).  y6_wgrad_transpose (train.hip) therefore writes channel-major copies
//    with the image row as the contiguous axis; both operands of v_mfma_f32_32x32x16_f16 are then plain 16-byte global
//    loads of one lane - no LDS, no barrier, no bank conflicts anywhere in this kernel.
//  * A 3x3 kernel's column taps (kx = 0 / 2) read the SAME 16-byte runs shifted by one element: the shifted fragments
//    are built in registers from the previous / current / next run with v_alignbit_b32 (4 VALU ops per fragment,
//    co-issued with the MFMAs); row taps are row offsets into a plane that carries one zero row above and below.
//    Stride-2 convs read four row/column parity planes instead (the transpose samples them), so every load stays a
//    contiguous aligned run.
//  * k order inside a row: lanes 0-31 walk the runs of the first half of the row, lanes 32-63 the second half, so a
//    lane's previous/next run is its own previous/next k-step (any k permutation is legal as long as A and B agree).
//  * One wave = one 32x32 (m, n) tile x all taps (9 x 16 accumulator registers) x one slice of the (image, row) range;
//    slices are summed with fp32 atomics straight into the OIHW gradient array (zeroed once per step by the caller).
#include "common.hpp"
#include "plan_internal.hpp"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WgArgs {
    const __half* a;
    int a_rows;                 // rows per image of A
    int M, N, B, Q, rows;       // rows: output rows per image walked (y range)
    const __half* plane[6];
    int plane_rows[6];
    int drow[6];
    float* out;
    int sm, sn, st;
    int mtiles, ntiles, nsplit, rows_per;   // work split
};

// per-mode stream table: for stream s, the tap fed by the run shifted by -1 / 0 / +1 (-1: unused)
template <int MODE> struct Mode;
template <> struct Mode<Y6_WG_3X3S1> {
    static constexpr int NS = 3, NT = 9;
    static constexpr int tap(int s, int sh) { return s * 3 + sh; }          // sh: 0 = shift -1 (kx 0), 1 = none, 2 = shift +1
};
template <> struct Mode<Y6_WG_1X1> {
    static constexpr int NS = 1, NT = 1;
    static constexpr int tap(int, int sh) { return sh == 1 ? 0 : -1; }
};
template <> struct Mode<Y6_WG_3X3S2> {
    // streams: s = ky*2 + colpar.  colpar 0 = even columns (kx 1, no shift); colpar 1 = odd columns (kx 0: shift -1, kx 2: none)
    static constexpr int NS = 6, NT = 9;
    static constexpr int tap(int s, int sh) {
        const int ky = s >> 1, cp = s & 1;
        if (cp == 0) return sh == 1 ? ky * 3 + 1 : -1;
        return sh == 0 ? ky * 3 + 0 : (sh == 1 ? ky * 3 + 2 : -1);
    }
};
template <> struct Mode<Y6_WG_CONVT> {
    static constexpr int NS = 4, NT = 4;
    static constexpr int tap(int s, int sh) { return sh == 1 ? s : -1; }
};

__device__ __forceinline__ h8_t as_h8(const u32x4 v) { return __builtin_bit_cast(h8_t, v); }

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

template <int MODE>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgArgs a) {
    using MD = Mode<MODE>;
    constexpr int NS = MD::NS, NT = MD::NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const long unit = (long)blockIdx.x * 4 + wave;
    const long units = (long)a.mtiles * a.ntiles * a.nsplit;
    if (unit >= units) return;
    const int nt = (int)(unit % a.ntiles);
    const int mt = (int)((unit / a.ntiles) % a.mtiles);
    const int ks = (int)(unit / ((long)a.ntiles * a.mtiles));
    const long total_rows = (long)a.B * a.rows;
    const long r0 = (long)ks * a.rows_per;
    long r1 = r0 + a.rows_per;
    if (r1 > total_rows) r1 = total_rows;

    int m = mt * 32 + l31, n = nt * 32 + l31;
    m = m < a.M ? m : a.M - 1;          // clamped rows/columns are computed and dropped at the end
    n = n < a.N ? n : a.N - 1;
    const int Qr = a.Q >> 3;            // 16-byte runs per row
    const int Qh = Qr >> 1;             // k-steps per row (Q % 16 == 0)
    const int j0 = half * Qh;           // first run of this half

    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (long r = r0; r < r1; ++r) {
        const int b = (int)(r / a.rows), y = (int)(r - (long)b * a.rows);
        const u32x4* ap = reinterpret_cast<const u32x4*>(a.a + (((size_t)m * a.B + b) * a.a_rows + y) * a.Q) + j0;
        const u32x4* bp[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
            bp[s] = reinterpret_cast<const u32x4*>(a.plane[s] + (((size_t)n * a.B + b) * a.plane_rows[s] + y + a.drow[s]) * a.Q) + j0;
        u32x4 prev[NS], cur[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cur[s] = bp[s][0];
            prev[s] = zero;
            if (MD::tap(s, 0) >= 0 && j0 > 0) prev[s] = bp[s][-1];
        }
        for (int k = 0; k < Qh; ++k) {
            const u32x4 av = ap[k];
            u32x4 next[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                next[s] = zero;
                if ((MD::tap(s, 2) >= 0 || k + 1 < Qh) && (j0 + k + 1 < Qr)) next[s] = bp[s][k + 1];
            }
            const h8_t af = as_h8(av);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (MD::tap(s, 0) >= 0)
                    acc[MD::tap(s, 0) >= 0 ? MD::tap(s, 0) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        af, as_h8(shift_m1(prev[s], cur[s])), acc[MD::tap(s, 0) >= 0 ? MD::tap(s, 0) : 0], 0, 0, 0);
                if (MD::tap(s, 1) >= 0)
                    acc[MD::tap(s, 1) >= 0 ? MD::tap(s, 1) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        af, as_h8(cur[s]), acc[MD::tap(s, 1) >= 0 ? MD::tap(s, 1) : 0], 0, 0, 0);
                if (MD::tap(s, 2) >= 0)
                    acc[MD::tap(s, 2) >= 0 ? MD::tap(s, 2) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        af, as_h8(shift_p1(cur[s], next[s])), acc[MD::tap(s, 2) >= 0 ? MD::tap(s, 2) : 0], 0, 0, 0);
                prev[s] = cur[s];
                cur[s] = next[s];
            }
        }
    }
    // C/D layout: column n = lane & 31, row m = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
    const int n_out = nt * 32 + l31;
    if (n_out >= a.N) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m_out = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
            if (m_out < a.M) atomicAdd(a.out + (size_t)m_out * a.sm + (size_t)n_out * a.sn + (size_t)t * a.st, acc[t][q]);
        }
}

int wgrad_launch(const y6_wgrad_desc* d, hipStream_t s) {
    Y6_REQUIRE(d && d->a && d->out, "wgrad: null argument");
    Y6_REQUIRE(d->mode >= Y6_WG_3X3S1 && d->mode <= Y6_WG_CONVT, "wgrad: unknown mode %d", d->mode);
    Y6_REQUIRE(d->M > 0 && d->N > 0 && d->B > 0 && d->rows > 0 && d->Q > 0 && d->Q % 16 == 0, "wgrad: bad sizes");
    Y6_REQUIRE(d->a_rows >= d->rows, "wgrad: A has fewer rows than the walked range");
    const int ns = d->mode == Y6_WG_3X3S1 ? 3 : d->mode == Y6_WG_1X1 ? 1 : d->mode == Y6_WG_3X3S2 ? 6 : 4;
    WgArgs a;
    memset(&a, 0, sizeof(a));
    a.a = (const __half*)d->a;
    a.a_rows = d->a_rows;
    a.M = d->M;
    a.N = d->N;
    a.B = d->B;
    a.Q = d->Q;
    a.rows = d->rows;
    Y6_REQUIRE(((uintptr_t)d->a & 15) == 0, "wgrad: A must be 16-byte aligned");
    for (int i = 0; i < ns; ++i) {
        Y6_REQUIRE(d->plane[i] && ((uintptr_t)d->plane[i] & 15) == 0, "wgrad: plane %d missing or unaligned", i);
        Y6_REQUIRE(d->drow[i] >= 0 && d->plane_rows[i] >= d->rows + d->drow[i], "wgrad: plane %d has too few rows", i);
        a.plane[i] = (const __half*)d->plane[i];
        a.plane_rows[i] = d->plane_rows[i];
        a.drow[i] = d->drow[i];
    }
    a.out = d->out;
    a.sm = d->sm;
    a.sn = d->sn;
    a.st = d->st;
    a.mtiles = y6_cdiv(d->M, 32);
    a.ntiles = y6_cdiv(d->N, 32);
    const long total_rows = (long)d->B * d->rows;
    const long tiles = (long)a.mtiles * a.ntiles;
    long nsplit = (4096 + tiles - 1) / tiles;        // ~4 waves per SIMD of work items over the chip
    if (nsplit > total_rows) nsplit = total_rows;
    if (nsplit < 1) nsplit = 1;
    a.rows_per = (int)((total_rows + nsplit - 1) / nsplit);
    a.nsplit = (int)((total_rows + a.rows_per - 1) / a.rows_per);
    const long units = tiles * a.nsplit;
    const unsigned grid = (unsigned)((units + 3) / 4);
    switch (d->mode) {
        case Y6_WG_3X3S1: hipLaunchKernelGGL(wgrad_kernel<Y6_WG_3X3S1>, dim3(grid), dim3(256), 0, s, a); break;
        case Y6_WG_1X1: hipLaunchKernelGGL(wgrad_kernel<Y6_WG_1X1>, dim3(grid), dim3(256), 0, s, a); break;
        case Y6_WG_3X3S2: hipLaunchKernelGGL(wgrad_kernel<Y6_WG_3X3S2>, dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(wgrad_kernel<Y6_WG_CONVT>, dim3(grid), dim3(256), 0, s, a); break;
    }
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

}  // namespace

extern "C" int y6_wgrad(const y6_wgrad_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return wgrad_launch(d, (hipStream_t)stream);
}

extern "C" int y6_plan_add_wgrad(y6_plan* p, const y6_wgrad_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    const double ns = d->mode == Y6_WG_3X3S1 ? 3 : d->mode == Y6_WG_1X1 ? 1 : d->mode == Y6_WG_3X3S2 ? 6 : 4;
    const double bytes = 2.0 * d->B * d->rows * d->Q * ((double)d->M + ns / 3.0 * d->N);
    return y6_plan_push(p, wgrad_launch, d, Y6_TOP_WGRAD, d->flops, bytes);
}
