// wgrad.hip — weight gradient of a convolution as a tap-table GEMM over pixels on the gfx950 matrix cores.
//
//   dW[m][n][t] += sum_{b, y, q}  A[m][b][y][q] * P_t[n][b][y + drow_t][q + shift_t]
//
// Replaces the weight half of autograd's conv backward for the training step (reference: the autograd graph of
// ConvModule / RepVGGBlock forward, yolov6/layers/common.py:45-49, :250-255; engine.py:173 `.backward()`).
//
// Design (MI355X-first, not a cuDNN wgrad port):
//  * The reduction runs over PIXELS, and NHWC keeps channels contiguous - the wrong way round for an MFMA operand
//    (a lane holds 8 consecutive k of one row).  y6_wgrad_transpose (train.hip) therefore writes copies in
//    [image][row][8-pixel run][channel][8] order: a lane's operand (8 consecutive pixels of one channel) is one 16-byte word
//    and the words of the 32 channels of a fragment are contiguous - both operands of v_mfma_f32_32x32x16_f16 are plain,
//    fully coalesced 16-byte global loads; no LDS, no barrier, no bank conflicts anywhere in this kernel.
//  * A 3x3 kernel's column taps (kx = 0 / 2) read the SAME 16-byte runs shifted by one element: the shifted fragments
//    are built in registers from the previous / current / next run with v_alignbit_b32 (4 VALU ops per fragment,
//    co-issued with the MFMAs); row taps are row offsets into a plane that carries one zero row above and below.
//    Stride-2 convs read four row/column parity planes instead (the transpose samples them), so every load stays a
//    contiguous aligned run.
//  * k order inside a row: lanes 0-31 walk the runs of the first half of the row, lanes 32-63 the second half, so a
//    lane's previous/next run is its own previous/next k-step (any k permutation is legal as long as A and B agree).
//  * One wave = one 32x32 (m, n) tile x one kernel row of taps (3 x 16 accumulator registers, 4 waves per SIMD) x one
//    slice of the (image, row) range;
//    each slice stores its partial tile to a workspace and a small second kernel sums the slices into the OIHW
//    gradient array (+=; zeroed once per step by the caller) - deterministic, no atomics.
#include <cstdlib>

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
    float* ws;                  // partial sums [nsplit][T][M][N]
    int T;
    int a_ch, b_ch;             // channels of the A plane / of the B planes (>= M, N: padded views)
};

// per-mode tables.  The taps of a mode are split into NG groups (one kernel row ky each for the 3x3 modes); a wave owns
// ONE group: 3 x 16 accumulator registers instead of 9 x 16, i.e. <= 128 registers and 4 waves per SIMD - the load latency
// of this LDS-free kernel is hidden by occupancy.  Group g reads streams [g*NSG, (g+1)*NSG); tap(s, sh) is the tap (within
// the group) fed by stream s of the group shifted by sh-1 columns, or -1.
template <int MODE> struct Mode;
template <> struct Mode<Y6_WG_3X3S1> {
    static constexpr int NG = 3, NSG = 1, NTG = 3;
    static constexpr int tap(int, int sh) { return sh; }                     // kx = sh (shift -1, 0, +1)
};
template <> struct Mode<Y6_WG_1X1> {
    static constexpr int NG = 1, NSG = 1, NTG = 1;
    static constexpr int tap(int, int sh) { return sh == 1 ? 0 : -1; }
};
template <> struct Mode<Y6_WG_3X3S2> {
    // group = ky; stream 0 = even columns (kx 1, no shift); stream 1 = odd columns (kx 0: shift -1, kx 2: none)
    static constexpr int NG = 3, NSG = 2, NTG = 3;
    static constexpr int tap(int s, int sh) {
        if (s == 0) return sh == 1 ? 1 : -1;
        return sh == 0 ? 0 : (sh == 1 ? 2 : -1);
    }
};
template <> struct Mode<Y6_WG_CONVT> {
    static constexpr int NG = 1, NSG = 4, NTG = 4;
    static constexpr int tap(int s, int sh) { return sh == 1 ? s : -1; }
};

__device__ __forceinline__ h8_t as_h8(const u32x4 v) { return __builtin_bit_cast(h8_t, v); }

// 16-byte global load the compiler cannot move or merge (see the operand-stream comment in the kernel); completion is
// awaited explicitly with wait_vmcnt<N>, which also ties the awaited register so that its uses stay behind the wait.
__device__ __forceinline__ void gload(u32x4& dst, const u32x4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt(u32x4& x) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N) : "memory");
}
// order a register of the same slot behind the wait that was tied to `w`
__device__ __forceinline__ void tie(u32x4& x, u32x4& w) { asm volatile("" : "+v"(x), "+v"(w)::"memory"); }

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

// waves per SIMD the register budget is cut for: the two-stream / four-stream modes would spill at 128 registers
template <int MODE> constexpr int kWavesPerSimd = (MODE == Y6_WG_3X3S1 || MODE == Y6_WG_1X1) ? 4 : 2;

template <int MODE>
__global__ __launch_bounds__(256, kWavesPerSimd<MODE>) void wgrad_kernel(const WgArgs a) {
    using MD = Mode<MODE>;
    constexpr int NG = MD::NG, NSG = MD::NSG, NTG = MD::NTG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware placement.  Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own
    // L2.  Every (m, n, tap-group) tile of one pixel slice `ks` reads the SAME operand rows, so all tiles of a slice are
    // put on one XCD, consecutively: the rows are fetched from HBM / Infinity Cache once instead of once per XCD
    // (measured: the 8x redundant fetch, not the MFMA rate, bounded the large layers at ~330 TFLOP/s).
    const int tg = a.mtiles * a.ntiles * NG;             // tiles x groups of one slice
    const int bps = (tg + 3) >> 2;                        // blocks per slice (4 waves each)
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int slot = bi % bps;
    const int ks = (bi / bps) * 8 + xcd;
    const int u = slot * 4 + wave;
    if (ks >= a.nsplit || u >= tg) return;
    // nt fastest, then the tap group: the waves of a block read the same A rows (L1 hits)
    const int nt = u % a.ntiles;
    const int grp = (u / a.ntiles) % NG;
    const int mt = u / (a.ntiles * NG);
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

    f32x16_t acc[NTG];
#pragma unroll
    for (int t = 0; t < NTG; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const __half* pl[NSG];
    int prow[NSG], drow[NSG];
#pragma unroll
    for (int s = 0; s < NSG; ++s) {
        pl[s] = a.plane[grp * NSG + s];
        prow[s] = a.plane_rows[grp * NSG + s];
        drow[s] = a.drow[grp * NSG + s];
    }

    // ---- operand stream ---------------------------------------------------------------------------------------------
    // A row is walked in Qh + 2 "slots": slot i fetches run i-1 of A and of every B stream (clamped into the row), so the
    // runs left and right of this half's range (the -1 / +1 column taps need them) arrive through the same stream.  The
    // fetches run ahead of the MFMAs ACROSS row boundaries (rows are only 2 ... 10 k-steps long: a per-row prologue would
    // expose one memory round trip per row).  Loads are issued from inline asm and awaited with an explicit s_waitcnt
    // vmcnt(N): left to itself hipcc sinks prefetch loads down to their uses (to meet the occupancy target) and every
    // k-step then waits a full round trip (measured 2200 cycles per k-step, 17 % MFMA duty).
    // Ring of R = 5 slots, addressed statically (the loop is unrolled by R): when slot t has arrived, the ring holds slots
    // t-2, t-1, t (= the -1 / 0 / +1 taps' runs, used in place - no copies) and t+1, t+2 in flight; entry (t-2) % R is
    // refilled with slot t+3 after the MFMAs of step t.  A register with a load in flight is never read or moved.
    constexpr int R = 5, LPS = NSG + 1;
    const int nslot = Qh + 2;
    const int lo = half ? -1 : 0;                      // first fetchable run relative to j0 (run -1 exists for the upper half)
    const int hi = Qr - 1 - j0;                        // last run inside the row
    const u32x4* const abase = reinterpret_cast<const u32x4*>(a.a);
    const int sa = a.a_ch, sb = a.b_ch;                // 16-byte words between consecutive runs
    long fr = r0;
    int fi = 0;
    int fb_ = (int)(r0 / a.rows), fy = (int)(r0 - (long)fb_ * a.rows);     // (image, row) of the fetch stream: no division in the loop
    const u32x4* fap = abase;
    const u32x4* fbp[NSG];
    auto set_row_ptrs = [&]() {
        fap = abase + (((size_t)fb_ * a.a_rows + fy) * Qr + j0) * sa + m;
#pragma unroll
        for (int s = 0; s < NSG; ++s)
            fbp[s] = reinterpret_cast<const u32x4*>(pl[s]) + (((size_t)fb_ * prow[s] + fy + drow[s]) * Qr + j0) * sb + n;
    };
    set_row_ptrs();
    auto fetch = [&](u32x4& fa, u32x4 (&fb)[NSG]) {    // issue the loads of slot (fr, fi), advance the stream
        int ra = fi - 1;
        ra = ra < 0 ? 0 : (ra > Qh - 1 ? Qh - 1 : ra);
        int rb = fi - 1;
        rb = rb < lo ? lo : (rb > hi ? hi : rb);
        gload(fa, fap + ra * sa);
#pragma unroll
        for (int s = 0; s < NSG; ++s) gload(fb[s], fbp[s] + rb * sb);
        if (++fi == nslot) {
            fi = 0;
            if (++fr < r1) {                           // past the end the stream re-reads the last row (never consumed)
                if (++fy == a.rows) {
                    fy = 0;
                    ++fb_;
                }
                set_row_ptrs();
            }
        }
    };
    u32x4 qa[R], qb[R][NSG];
#pragma unroll
    for (int d = 0; d < 3; ++d) fetch(qa[d], qb[d]);   // slots 0, 1, 2

    const unsigned mlo = half ? 0xffffffffu : 0u, mhi = half ? 0u : 0xffffffffu;   // validity of run -1 / run Qh for this lane
    const long nslots = (r1 - r0) * nslot;
    int ci = 0;                                        // slot index inside the current row (compute stream)
    for (long g = 0; g < nslots; g += R) {
#pragma unroll
        for (int d = 0; d < R; ++d) {
            if (g + d >= nslots) break;
            constexpr int dummy = 0;
            // entries: next = d, cur = d-1, prev = d-2 (mod R); refill target = d+3 (mod R) = prev's successor cycle
            const int e_next = d, e_cur = (d + R - 1) % R, e_prev = (d + R - 2) % R, e_fill = (d + 3) % R;
            wait_vmcnt<2 * LPS>(qa[e_next]);           // slots t+1, t+2 may be in flight; slot t has arrived
#pragma unroll
            for (int s = 0; s < NSG; ++s) tie(qb[e_next][s], qa[e_next]);
            const int i = ci;
            ci = ci + 1 == nslot ? 0 : ci + 1;
            if (i >= 2) {                              // k = i - 2: A run k (slot t-1), B runs k-1, k, k+1 (slots t-2, t-1, t)
                const h8_t af = as_h8(qa[e_cur]);
                const bool first = i == 2, last = i == nslot - 1;
#pragma unroll
                for (int s = 0; s < NSG; ++s) {
                    if (MD::tap(s, 0) >= 0) {
                        u32x4 pv = qb[e_prev][s];
                        if (first) pv[3] &= mlo;       // only element 7 of the previous run is used
                        acc[MD::tap(s, 0) >= 0 ? MD::tap(s, 0) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            af, as_h8(shift_m1(pv, qb[e_cur][s])), acc[MD::tap(s, 0) >= 0 ? MD::tap(s, 0) : 0], 0, 0, 0);
                    }
                    if (MD::tap(s, 1) >= 0)
                        acc[MD::tap(s, 1) >= 0 ? MD::tap(s, 1) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            af, as_h8(qb[e_cur][s]), acc[MD::tap(s, 1) >= 0 ? MD::tap(s, 1) : 0], 0, 0, 0);
                    if (MD::tap(s, 2) >= 0) {
                        u32x4 nx = qb[e_next][s];
                        if (last) nx[0] &= mhi;        // only element 0 of the next run is used
                        acc[MD::tap(s, 2) >= 0 ? MD::tap(s, 2) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            af, as_h8(shift_p1(qb[e_cur][s], nx)), acc[MD::tap(s, 2) >= 0 ? MD::tap(s, 2) : 0], 0, 0, 0);
                    }
                }
            }
            // entry e_fill held slot t-2 (consumed above as `prev`): refill it with slot t+3.  The MFMAs that read it were
            // issued; the in-order VMEM return writes it long after they have read their operands.
            fetch(qa[e_fill], qb[e_fill]);
            (void)dummy;
        }
    }
    wait_vmcnt<0>(qa[0]);                              // drain the look-ahead fetches before the registers die
#pragma unroll
    for (int d = 1; d < R; ++d) tie(qa[d], qa[0]);
    // C/D layout: column n = lane & 31, row m = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5).
    // The slice's partial tile goes to the workspace with plain stores (n contiguous across lanes); wgrad_reduce_kernel
    // sums the slices.  (Device-scope float atomics from eight XCDs resolve at the memory side: 25 M of them per launch
    // cost more than the MFMAs - measured 53 TFLOP/s with atomics.)
    const int n_out = nt * 32 + l31;
    if (n_out >= a.N) return;
#pragma unroll
    for (int t = 0; t < NTG; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m_out = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
            if (m_out < a.M) a.ws[(((size_t)ks * a.T + (grp * NTG + t)) * a.M + m_out) * a.N + n_out] = acc[t][q];
        }
}

// =====================================================================================================================
// NHWC-fed form (3x3 stride 1 and 1x1 stride 1 - 9.6 of the 13 ms the weight gradients of a YOLOv6-S step take): no transposed
// copies.  The run-major planes above exist because an MFMA operand wants 8 consecutive PIXELS of one channel per lane
// while NHWC keeps channels contiguous; gfx950's ds_read_b64_tr_b16 does that transpose on the way out of LDS (a 16-lane
// group reads a [4 pixels][16 channels] block and every lane receives one channel's four pixels).  So:
//  * a block (4 waves) owns one 64 x 64 (cout, cin) tile x one slice of the (image, row) range and walks its rows; the dy row
//    and the x rows y-1 / y / y+1 arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction) straight from the NHWC
//    tensors as [32-channel chunk][pixel][32 channels] images (64-byte pixel pitch: the 4 x 64 bytes a 32-lane half reads
//    per transpose-read are contiguous, no bank conflicts); x rows live in a ring of four (each is fetched ONCE per slice and
//    serves three output rows), dy rows in a ring of two; the next row's requests are in flight while the current row is
//    multiplied - one barrier per row; padding rows / columns beyond W / channels beyond the view are zero-filled by the
//    buffer descriptor's range check;
//  * two transpose-reads give a lane exactly the 16-byte run (8 consecutive pixels of one channel) the plane-fed kernel loads,
//    so the register side is unchanged: k order "lanes 0-31 first half of the row, 32-63 second half", +-1 column taps by
//    v_alignbit from the previous / next run, row-end masks;
//  * a wave owns one 32 x 32 tile and ALL taps (nine accumulator tiles): one dy operand and three x operands per k-step feed
//    nine MFMAs (the plane-fed kernel: two operands from L1 / L2 per three MFMAs).
struct WgLArgs {
    const __half* a;
    const __half* x;
    unsigned a_bytes, x_bytes;     // extents of the two buffers (descriptor range check)
    int a_cs, a_co, a_C;           // dy view: pixel pitch, channel offset (halves), readable channels (multiple of 8, >= M)
    int x_cs, x_co, x_C;
    int B, H, W, Q;
    int M, N;
    int mt2, nt2, nsplit, rows_per;
    float* ws;
};

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned kOobL = 0xf0000000u;   // voffset of a piece that must read zeros (tensors stay below 3.5 GiB)

__device__ __forceinline__ i32x4_t make_rsrc_l(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)base;
    i32x4_t r;
    r[0] = (int)(unsigned)(p & 0xffffffffu);
    r[1] = (int)(unsigned)((p >> 32) & 0xffffu);   // stride 0: raw buffer, byte offsets, range check against num_records
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
}
// one LDS-DMA piece (see conv_dma.hip): lane i writes 16 B to lds_dst + 16*i from rsrc.base + voff(lane); out of range -> zeros
__device__ __forceinline__ void dma16_l(const i32x4_t& rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr_l(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
// 8 consecutive pixels of one channel: two transpose-reads 256 bytes (4 pixels) apart
__device__ __forceinline__ u32x4 tr_run(const char* p) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 256));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 r;
    r[0] = l2[0], r[1] = l2[1], r[2] = h2[0], r[3] = h2[1];
    return r;
}

template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad_lds_kernel(const WgLArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = KS * KS;
    constexpr int XS = KS == 3 ? 4 : 2;                 // x ring slots
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware placement as above: the tiles of one slice share an XCD (its rows come from HBM once)
    const int tg = a.mt2 * a.nt2;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int slot = bi % tg, ks = (bi / tg) * 8 + xcd;
    if (ks >= a.nsplit) return;
    const int n2 = slot % a.nt2, m2 = slot / a.nt2;
    const int mtl = wave & 1, ntl = wave >> 1;
    const bool active = (m2 * 64 + mtl * 32 < a.M) && (n2 * 64 + ntl * 32 < a.N);
    const long total_rows = (long)a.B * a.H;
    const long r0 = (long)ks * a.rows_per;
    long r1 = r0 + a.rows_per;
    if (r1 > total_rows) r1 = total_rows;
    const int Qr = a.Q >> 3, Qh = Qr >> 1;
    const int rowb = a.Q * 64;                           // bytes of one 32-channel chunk row
    const int slotb = 2 * rowb;
    char* const ldsA = smem;                             // 2 slots
    char* const ldsX = smem + 2 * slotb;                 // XS slots
    const i32x4_t rsA = make_rsrc_l(a.a, a.a_bytes), rsX = make_rsrc_l(a.x, a.x_bytes);

    // one NHWC row (64 channels from ch0) -> the two chunk images of a slot; grow < 0: a zero row
    auto issue_row = [&](const i32x4_t& rs, int cs, int co, int Cv, int ch0, long grow, unsigned lds_base) {
        const int ni = a.Q >> 4;                         // 1 KiB instructions per chunk row
        for (int j = wave; j < 2 * ni; j += 4) {
            const int c = j >= ni ? 1 : 0, i = j - c * ni;
            if (ch0 + c * 32 >= Cv && grow >= 0) continue;       // a chunk nobody reads (its waves are inactive)
            const int pix = i * 16 + (lane >> 2);
            const int ch = ch0 + c * 32 + (lane & 3) * 8;
            const bool ok = grow >= 0 && pix < a.W && ch + 8 <= Cv;
            const unsigned voff = ok ? (unsigned)((((unsigned long long)grow * a.W + pix) * cs + co + ch) * 2ull) : kOobL;
            dma16_l(rs, voff, lds_base + (unsigned)(c * rowb + i * 1024));
        }
    };
    auto xrow_issue = [&](long p) {                      // p: padded row index (image stride H + 2) for 3x3, row index for 1x1
        long grow = p;
        if (KS == 3) {
            const long bp = p / (a.H + 2);
            const int yy = (int)(p - bp * (a.H + 2)) - 1;
            grow = (yy >= 0 && yy < a.H) ? bp * a.H + yy : -1;
        }
        issue_row(rsX, a.x_cs, a.x_co, a.x_C, n2 * 64, grow, lds_addr_l(ldsX) + (unsigned)(p & (XS - 1)) * (unsigned)slotb);
    };
    auto arow_issue = [&](long gr) { issue_row(rsA, a.a_cs, a.a_co, a.a_C, m2 * 64, gr, lds_addr_l(ldsA) + (unsigned)(gr & 1) * (unsigned)slotb); };
    auto sync_all = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    // lane part of a transpose-read address: pixel (lane&15)>>2 of the group's four, channels 4*(lane&3) + 16*((lane>>4)&1) of
    // the chunk, and this half's first run
    const int lane_off = ((lane & 15) >> 2) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2 + half * Qh * 512;
    const int run_lo = half ? -1 : 0;                    // first / last run this half may read, relative to its first run
    const int run_hi = half ? Qh - 1 : Qr - 1;
    const unsigned mlo = half ? 0xffffffffu : 0u, mhi = half ? 0u : 0xffffffffu;   // validity of run -1 / run Qh for this lane

    long xfront = -1, afront = r0 - 1;
    for (long gr = r0; gr < r1; ++gr) {
        const long bimg = gr / a.H;
        const long plo = KS == 3 ? bimg * (a.H + 2) + (gr - bimg * a.H) : gr;
        const long phi = plo + KS - 1;
        sync_all();                                      // the row requested one iteration ago has landed; everyone left row gr-1
        if (xfront < phi || afront < gr) {               // first row of the slice, first row of an image
            if (xfront < plo - 1) xfront = plo - 1;
            while (xfront < phi) xrow_issue(++xfront);
            if (afront < gr) {
                arow_issue(gr);
                afront = gr;
            }
            sync_all();
        }
        if (gr + 1 < r1) {                               // next row: its dy row and the one x row it adds
            arow_issue(gr + 1);
            afront = gr + 1;
            xrow_issue(phi + 1);
            xfront = phi + 1;
        }
        // (waves without a tile of their own - M or N of 32 - multiply whatever their chunk images hold and drop the result:
        // straight-line code below, no accumulator copies at control-flow joins)
        const char* ab = ldsA + (gr & 1) * slotb + mtl * rowb + lane_off;
        if constexpr (KS == 3) {
            const char* xb[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) xb[r] = ldsX + ((plo + r) & 3) * slotb + ntl * rowb + lane_off;
            u32x4 P[3], Cc[3], Nx[3], Ac, An;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                P[r] = tr_run(xb[r] + run_lo * 512);
                Cc[r] = tr_run(xb[r]);
                P[r][3] &= mlo;                          // k = 0: only element 7 of the previous run is used
            }
            Ac = tr_run(ab);
            for (int k = 0; k < Qh; ++k) {
                const int kn = k + 1 > run_hi ? run_hi : k + 1;
                const int ka = k + 1 < Qh ? k + 1 : k;
#pragma unroll
                for (int r = 0; r < 3; ++r) Nx[r] = tr_run(xb[r] + kn * 512);
                An = tr_run(ab + ka * 512);
                const h8_t af = as_h8(Ac);
                const unsigned mnext = k == Qh - 1 ? mhi : 0xffffffffu;   // k = Qh - 1: only element 0 of the next run is used
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc[r * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(shift_m1(P[r], Cc[r])), acc[r * 3 + 0], 0, 0, 0);
                    acc[r * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(Cc[r]), acc[r * 3 + 1], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    u32x4 nx = Nx[r];
                    nx[0] &= mnext;
                    acc[r * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(shift_p1(Cc[r], nx)), acc[r * 3 + 2], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    P[r] = Cc[r];
                    Cc[r] = Nx[r];
                }
                Ac = An;
            }
        } else {
            const char* xb = ldsX + (gr & 1) * slotb + ntl * rowb + lane_off;
            u32x4 Ac = tr_run(ab), Bc = tr_run(xb);
            for (int k = 0; k < Qh; ++k) {
                const int ka = k + 1 < Qh ? k + 1 : k;
                const u32x4 An = tr_run(ab + ka * 512), Bn = tr_run(xb + ka * 512);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(Ac), as_h8(Bc), acc[0], 0, 0, 0);
                Ac = An;
                Bc = Bn;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no request may outlive the block's LDS allocation
    if (!active) return;
    const int n_out = n2 * 64 + ntl * 32 + l31;
    if (n_out >= a.N) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m_out = m2 * 64 + mtl * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
            if (m_out < a.M) a.ws[(((size_t)ks * NT + t) * a.M + m_out) * a.N + n_out] = acc[t][q];
        }
}

// Two-level, deterministic sum of the slice partials (a single level - one thread walking all `nsplit` slices of an element -
// is a serial chain of strided loads: 0.77 ms for the 683 slices of the 64->64 @160x160 layers):
//   level 1: chunk c of kRedChunk slices -> ws2[c][t][m][n]              (grid: elements x chunks)
//   level 2: out[m*sm + n*sn + t*st] += sum_c ws2[c][t][m][n]
constexpr int kRedChunk = 16;
__global__ __launch_bounds__(256) void wgrad_reduce1_kernel(const float* __restrict__ ws, int nsplit, size_t per, float* __restrict__ ws2) {
    const int c = blockIdx.y;
    const int k0 = c * kRedChunk, k1 = (k0 + kRedChunk < nsplit) ? k0 + kRedChunk : nsplit;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
#pragma unroll 4
        for (int k = k0; k < k1; ++k) s += ws[(size_t)k * per + i];
        ws2[(size_t)c * per + i] = s;
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce2_kernel(const float* __restrict__ ws2, int nchunk, int T, int M, int N, float* __restrict__ out,
                                                            int sm, int sn, int st) {
    const size_t per = (size_t)T * M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < nchunk; ++c) s += ws2[(size_t)c * per + i];
        const int n = (int)(i % N);
        const int m = (int)((i / N) % M);
        const int t = (int)(i / ((size_t)N * M));
        out[(size_t)m * sm + (size_t)n * sn + (size_t)t * st] += s;
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
    Y6_REQUIRE(d->a_channels >= d->M && d->plane_channels >= d->N, "wgrad: plane channel counts smaller than M / N");
    a.a_ch = d->a_channels;
    a.b_ch = d->plane_channels;
    a.out = d->out;
    a.sm = d->sm;
    a.sn = d->sn;
    a.st = d->st;
    a.mtiles = y6_cdiv(d->M, 32);
    a.ntiles = y6_cdiv(d->N, 32);
    const int ng = (d->mode == Y6_WG_3X3S1 || d->mode == Y6_WG_3X3S2) ? 3 : 1;
    const int T = d->mode == Y6_WG_1X1 ? 1 : (d->mode == Y6_WG_CONVT ? 4 : 9);
    const long total_rows = (long)d->B * d->rows;
    const long tiles = (long)a.mtiles * a.ntiles * ng;
    // ~8 waves per SIMD of work items over the chip (4 resident); 4096 / 16384 items measured the same step time
    // (profiles/r03/bench_train_r03l_items{4k,16k}.json)
    long nsplit = (8192 + tiles - 1) / tiles;
    // ... but a slice should hold >= ~256 MFMAs (its fixed costs: pointer set-up, first-load latency per row, the partial tile)
    const long mfma_per_row = (long)(d->Q / 16) * (T / ng);
    const long max_by_work = (total_rows * mfma_per_row + 255) / 256;
    if (nsplit > max_by_work) nsplit = max_by_work;
    Y6_REQUIRE(d->workspace && d->workspace_bytes >= 2 * (size_t)T * d->M * d->N * sizeof(float), "wgrad: workspace missing or too small");
    // workspace: nsplit slice partials + ceil(nsplit / kRedChunk) chunk sums
    long max_by_ws = (long)(d->workspace_bytes / ((size_t)T * d->M * d->N * sizeof(float)));
    max_by_ws = max_by_ws * kRedChunk / (kRedChunk + 1) - 1;
    if (max_by_ws < 1) max_by_ws = 1;
    if (nsplit > max_by_ws) nsplit = max_by_ws;
    if (nsplit > total_rows) nsplit = total_rows;
    if (nsplit < 1) nsplit = 1;
    a.ws = (float*)d->workspace;
    a.T = T;
    a.rows_per = (int)((total_rows + nsplit - 1) / nsplit);
    a.nsplit = (int)((total_rows + a.rows_per - 1) / a.rows_per);
    const long bps = (tiles + 3) / 4;                               // blocks per slice; slices are dealt to the 8 XCDs
    const unsigned grid = (unsigned)(8 * ((a.nsplit + 7) / 8) * bps);
    switch (d->mode) {
        case Y6_WG_3X3S1: hipLaunchKernelGGL(wgrad_kernel<Y6_WG_3X3S1>, dim3(grid), dim3(256), 0, s, a); break;
        case Y6_WG_1X1: hipLaunchKernelGGL(wgrad_kernel<Y6_WG_1X1>, dim3(grid), dim3(256), 0, s, a); break;
        case Y6_WG_3X3S2: hipLaunchKernelGGL(wgrad_kernel<Y6_WG_3X3S2>, dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(wgrad_kernel<Y6_WG_CONVT>, dim3(grid), dim3(256), 0, s, a); break;
    }
    Y6_LAUNCH_CHECK();
    const size_t per = (size_t)T * d->M * d->N;
    unsigned rg = (unsigned)((per + 255) / 256);
    if (rg > 4096) rg = 4096;
    const int nchunk = (a.nsplit + kRedChunk - 1) / kRedChunk;
    float* ws2 = a.ws + (size_t)a.nsplit * per;
    hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3(rg, (unsigned)nchunk), dim3(256), 0, s, a.ws, a.nsplit, per, ws2);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3(rg), dim3(256), 0, s, ws2, nchunk, T, d->M, d->N, d->out, d->sm, d->sn, d->st);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

size_t wgrad_nhwc_lds_bytes(int ksize, int Q) { return (size_t)(2 + (ksize == 3 ? 4 : 2)) * 2 * Q * 64; }

bool wgrad_nhwc_view_ok(const y6_tensor& t) {
    return t.data && t.C % 8 == 0 && t.cstride % 8 == 0 && t.coff % 8 == 0 && (((uintptr_t)t.data) & 15) == 0 &&
           (size_t)t.B * t.H * t.W * t.cstride * 2 < 0xf0000000ull;
}

const char* wgrad_nhwc_unsupported(const y6_wgrad_nhwc_desc* d) {
    if (!d || !d->out) return "null argument";
    if (d->ksize != 1 && d->ksize != 3) return "ksize must be 1 or 3 (stride 1)";
    if (!wgrad_nhwc_view_ok(d->dy) || !wgrad_nhwc_view_ok(d->x)) return "views must be fp16 NHWC, 8-channel / 16-byte aligned, below 3.75 GiB";
    if (d->dy.B != d->x.B || d->dy.H != d->x.H || d->dy.W != d->x.W || d->x.B < 1 || d->x.H < 1 || d->x.W < 1) return "dy and x must have one spatial shape";
    if (d->M < 1 || d->N < 1 || d->dy.C < d->M || d->x.C < d->N) return "views narrower than M / N";
    const int Q = (d->x.W + 15) / 16 * 16;
    if (wgrad_nhwc_lds_bytes(d->ksize, Q) > 160 * 1024) return "row too wide for the LDS row ring";
    return nullptr;
}

int wgrad_nhwc_launch(const y6_wgrad_nhwc_desc* d, hipStream_t s) {
    const char* why = wgrad_nhwc_unsupported(d);
    Y6_REQUIRE(why == nullptr, "wgrad_nhwc: %s", why ? why : "");
    WgLArgs a;
    memset(&a, 0, sizeof(a));
    a.a = (const __half*)d->dy.data;
    a.x = (const __half*)d->x.data;
    a.a_bytes = (unsigned)((size_t)d->dy.B * d->dy.H * d->dy.W * d->dy.cstride * 2);
    a.x_bytes = (unsigned)((size_t)d->x.B * d->x.H * d->x.W * d->x.cstride * 2);
    a.a_cs = d->dy.cstride, a.a_co = d->dy.coff, a.a_C = d->dy.C;
    a.x_cs = d->x.cstride, a.x_co = d->x.coff, a.x_C = d->x.C;
    a.B = d->x.B, a.H = d->x.H, a.W = d->x.W;
    a.Q = (a.W + 15) / 16 * 16;
    a.M = d->M, a.N = d->N;
    a.mt2 = y6_cdiv(d->M, 64);
    a.nt2 = y6_cdiv(d->N, 64);
    const int T = d->ksize * d->ksize;
    const long total_rows = (long)a.B * a.H;
    const long tiles = (long)a.mt2 * a.nt2;
    // one block per CU (the row ring fills the LDS): three rounds of blocks over the chip, slices of >= 4 rows (a slice pays
    // for two extra x rows and one exposed round trip)
    long nsplit = (768 + tiles - 1) / tiles;
    if (nsplit > total_rows / 4) nsplit = total_rows / 4;
    if (nsplit < 1) nsplit = 1;
    const size_t per = (size_t)T * d->M * d->N;
    Y6_REQUIRE(d->workspace && d->workspace_bytes >= 2 * per * sizeof(float), "wgrad_nhwc: workspace missing or too small");
    long max_by_ws = (long)(d->workspace_bytes / (per * sizeof(float)));
    max_by_ws = max_by_ws * kRedChunk / (kRedChunk + 1) - 1;
    if (max_by_ws < 1) max_by_ws = 1;
    if (nsplit > max_by_ws) nsplit = max_by_ws;
    a.rows_per = (int)((total_rows + nsplit - 1) / nsplit);
    a.nsplit = (int)((total_rows + a.rows_per - 1) / a.rows_per);
    a.ws = (float*)d->workspace;
    const size_t lds = wgrad_nhwc_lds_bytes(d->ksize, a.Q);
    const unsigned grid = (unsigned)(8 * ((a.nsplit + 7) / 8) * tiles);
    static bool big3 = false, big1 = false;
    if (d->ksize == 3) {
        if (!big3) {
            Y6_HIP(hipFuncSetAttribute((const void*)wgrad_lds_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            big3 = true;
        }
        hipLaunchKernelGGL(wgrad_lds_kernel<3>, dim3(grid), dim3(256), lds, s, a);
    } else {
        if (!big1) {
            Y6_HIP(hipFuncSetAttribute((const void*)wgrad_lds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            big1 = true;
        }
        hipLaunchKernelGGL(wgrad_lds_kernel<1>, dim3(grid), dim3(256), lds, s, a);
    }
    Y6_LAUNCH_CHECK();
    unsigned rg = (unsigned)((per + 255) / 256);
    if (rg > 4096) rg = 4096;
    const int nchunk = (a.nsplit + kRedChunk - 1) / kRedChunk;
    float* ws2 = a.ws + (size_t)a.nsplit * per;
    hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3(rg, (unsigned)nchunk), dim3(256), 0, s, a.ws, a.nsplit, per, ws2);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3(rg), dim3(256), 0, s, ws2, nchunk, T, d->M, d->N, d->out, d->sm, d->sn, d->st);
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

extern "C" int y6_wgrad_nhwc_supported(const y6_wgrad_nhwc_desc* d) { return wgrad_nhwc_unsupported(d) == nullptr ? 1 : 0; }

extern "C" int y6_wgrad_nhwc(const y6_wgrad_nhwc_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return wgrad_nhwc_launch(d, (hipStream_t)stream);
}

extern "C" int y6_plan_add_wgrad_nhwc(y6_plan* p, const y6_wgrad_nhwc_desc* d) {
    Y6_REQUIRE(p && d, "plan_add: null argument");
    const char* why = wgrad_nhwc_unsupported(d);
    Y6_REQUIRE(why == nullptr, "wgrad_nhwc: %s", why ? why : "");
    const double bytes = 2.0 * d->x.B * d->x.H * d->x.W * ((double)d->M + d->N);
    return y6_plan_push(p, wgrad_nhwc_launch, d, Y6_TOP_WGRAD, d->flops, bytes);
}
