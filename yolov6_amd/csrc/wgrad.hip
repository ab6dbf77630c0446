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
    int da, dx;                    // rows the dy / x requests run ahead of the MFMAs (1 or 2, by LDS capacity)
    int dbg;                       // timing probes only
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

// wait until at most n of this wave's vector-memory operations are in flight (they complete in order: the older rows have landed)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int KS>
__global__ __launch_bounds__(512, 2) void wgrad_lds_kernel(const WgLArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = KS * KS;
    constexpr int NWV = 8;                               // waves per block: four tiles x two halves of a row's k-steps
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware placement: work items (slice, tile) in slice-major order, an eighth of the list per XCD (block b runs on XCD
    // b % 8) - the tiles of one slice land on one XCD wherever a slice has at least that many tiles' worth of neighbours, and
    // every XCD gets the same number of blocks also when there are fewer than eight slices (512 x 512 @ 20 x 20: 64 tiles x 4)
    const int tg = a.mt2 * a.nt2;
    const int nwork = a.nsplit * tg, per_xcd = (nwork + 7) >> 3;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int item = xcd * per_xcd + bi;
    if (bi >= per_xcd || item >= nwork) return;
    const int ks = item / tg, slot = item - ks * tg;
    const int n2 = slot % a.nt2, m2 = slot / a.nt2;
    const int mtl = wave & 1, ntl = (wave >> 1) & 1, kpart = wave >> 2;
    const bool active = (m2 * 64 + mtl * 32 < a.M) && (n2 * 64 + ntl * 32 < a.N);
    const int total_rows = a.B * a.H;
    const int r0 = ks * a.rows_per;
    const int r1 = r0 + a.rows_per < total_rows ? r0 + a.rows_per : total_rows;
    const int Qr = a.Q >> 3, Qh = Qr >> 1;
    const int kper = (Qh + 1) >> 1;                      // k-steps of a row this wave multiplies: [k0, k1)
    const int k0 = kpart * kper, k1 = k0 + kper < Qh ? k0 + kper : Qh;
    const int rowb = a.Q * 64;                           // bytes of one 32-channel chunk row
    const int slotb = 2 * rowb;
    // row rings: dy rows da rows ahead (da + 1 slots), x rows dx ahead (dx + 3 slots for the 3x3's three rows, dx + 1 for 1x1)
    const int AS = a.da + 1, XS = a.dx + (KS == 3 ? 3 : 1);
    char* const ldsA = smem;
    char* const ldsX = smem + AS * slotb;
    const i32x4_t rsA = make_rsrc_l(a.a, a.a_bytes), rsX = make_rsrc_l(a.x, a.x_bytes);
    const bool no_mfma = (a.dbg & 2) != 0;              // timing probe (tools/wgrad_bench.py): requests and barriers only

    // one NHWC row (64 channels from ch0) -> the two chunk images of a slot; grow < 0: a zero row.  Returns this wave's requests.
    auto issue_row = [&](const i32x4_t& rs, int cs, int co, int Cv, int ch0, int grow, unsigned lds_base) -> int {
        const int ni = a.Q >> 4;                         // 1 KiB instructions per chunk row
        int n = 0;
        for (int j = wave; j < 2 * ni; j += NWV) {
            const int c = j >= ni ? 1 : 0, i = j - c * ni;
            if (ch0 + c * 32 >= Cv) continue;            // a chunk nobody reads (its waves are inactive)
            const int pix = i * 16 + (lane >> 2);
            const int ch = ch0 + c * 32 + (lane & 3) * 8;
            const bool ok = grow >= 0 && pix < a.W && ch + 8 <= Cv;
            const unsigned voff = ok ? (unsigned)((((unsigned long long)grow * a.W + pix) * cs + co + ch) * 2ull) : kOobL;
            dma16_l(rs, voff, lds_base + (unsigned)(c * rowb + i * 1024));
            ++n;
        }
        return n;
    };
    // x rows are addressed by a PADDED row index (3x3: image stride H + 2 - one zero row above and below every image; 1x1: the
    // row itself); output row (b, y) multiplies padded rows b*(H+2) + y .. + KS - 1.  All ring slots, image / row counters advance
    // incrementally: a 32-bit division is ~150 cycles and ten of them per row cost more than the MFMAs of a 40 x 40 row.
    const int y_first = (int)((unsigned)r0 % (unsigned)a.H), b_first = (int)((unsigned)r0 / (unsigned)a.H);
    int y = y_first;                                     // row of gr inside its image
    int plo = KS == 3 ? b_first * (a.H + 2) + y_first : r0;
    auto pad_ahead = [&](int j) -> int {                 // padded top row of output row gr + j
        if (KS != 3) return plo + j;
        int yy = y + j, add = 0;
        while (yy >= a.H) {
            yy -= a.H;
            add += 2;
        }
        return plo + j + add;
    };
    // request frontier of x: padded row xfront = (image fb, padded row fyy of it), ring slot sxf
    int xfront = plo - 1, fb = b_first, fyy = y_first - 1, sxf, sx_lo;
    if (KS == 3) {
        if (fyy < 0) {
            fyy = a.H + 1;
            --fb;
        }
    }
    sx_lo = (int)((unsigned)plo % (unsigned)XS);
    sxf = sx_lo == 0 ? XS - 1 : sx_lo - 1;
    auto xrow_issue_next = [&]() -> int {                // requests padded row ++xfront
        ++xfront;
        if (++sxf == XS) sxf = 0;
        int grow = xfront;
        if (KS == 3) {
            if (++fyy == a.H + 2) {
                fyy = 0;
                ++fb;
            }
            grow = (fyy >= 1 && fyy <= a.H) ? fb * a.H + fyy - 1 : -1;
        }
        return issue_row(rsX, a.x_cs, a.x_co, a.x_C, n2 * 64, grow, lds_addr_l(ldsX) + (unsigned)sxf * (unsigned)slotb);
    };
    int afront = r0 - 1, sa = (int)((unsigned)r0 % (unsigned)AS), saf = sa == 0 ? AS - 1 : sa - 1;
    auto arow_issue_next = [&]() -> int {                // requests dy row ++afront
        ++afront;
        if (++saf == AS) saf = 0;
        return issue_row(rsA, a.a_cs, a.a_co, a.a_C, m2 * 64, afront, lds_addr_l(ldsA) + (unsigned)saf * (unsigned)slotb);
    };
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
    const int kprev = k0 - 1 < run_lo ? run_lo : k0 - 1;
    const unsigned mfirst = k0 == 0 ? mlo : 0xffffffffu;

    int allow = 0;                                       // requests of this wave that may still be in flight at the next row's start
    for (int gr = r0; gr < r1; ++gr) {
        const int phi = plo + KS - 1;
        wait_vm(allow);                                  // row gr's images have landed (younger rows may still be in flight) ...
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... for every wave, and everyone left row gr-1
        if (xfront < phi || afront < gr) {               // first row of the slice, first row of an image
            while (xfront < phi) xrow_issue_next();
            if (afront < gr) arow_issue_next();
            sync_all();
        }
        // requests for the rows ahead.  Slots: the ring holds dy rows gr .. gr+da and padded x rows plo .. plo+XS-1.
        {
            int issued = 0;
            bool younger = true;                         // none of them is needed by row gr+1
            const int alast = gr + a.da < r1 - 1 ? gr + a.da : r1 - 1;
            while (afront < alast) {
                issued += arow_issue_next();
                younger = younger && afront > gr + 1;
            }
            const int jlast = gr + a.dx < r1 - 1 ? a.dx : r1 - 1 - gr;
            int xlast = pad_ahead(jlast) + KS - 1;
            if (xlast > plo + XS - 1) xlast = plo + XS - 1;
            const int pnext = gr + 1 < r1 ? pad_ahead(1) + KS - 1 : -1;
            while (xfront < xlast) {
                issued += xrow_issue_next();
                younger = younger && xfront > pnext;
            }
            allow = younger ? issued : 0;
        }
        // this row's slots, then the counters move on to row gr + 1
        const int sa_now = sa, sx_now = sx_lo;
        {
            if (++sa == AS) sa = 0;
            int d = 1;
            if (++y == a.H) {
                y = 0;
                if (KS == 3) d = 3;
            }
            plo += d;
            sx_lo += d;
            while (sx_lo >= XS) sx_lo -= XS;
        }
        if (no_mfma) continue;
        // (waves without a tile of their own - M or N of 32 - multiply whatever their chunk images hold and drop the result:
        // straight-line code below, no accumulator copies at control-flow joins)
        const char* ab = ldsA + sa_now * slotb + mtl * rowb + lane_off;
        if constexpr (KS == 3) {
            const char* xb[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) xb[r] = ldsX + (sx_now + r < XS ? sx_now + r : sx_now + r - XS) * slotb + ntl * rowb + lane_off;
            // three run sets rotate through the roles (previous, current, next): the loop is unrolled by hand over the rotation
            // so that no set is ever copied; the dy operand alternates between two registers the same way
            u32x4 R0[3], R1[3], R2[3], A0, A1;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                R0[r] = tr_run(xb[r] + kprev * 512);
                R1[r] = tr_run(xb[r] + k0 * 512);
                R0[r][3] &= mfirst;                      // k = 0: only element 7 of the previous run is used
            }
            A0 = tr_run(ab + k0 * 512);
            auto step = [&](const u32x4 (&P)[3], const u32x4 (&Cc)[3], u32x4 (&Nx)[3], const u32x4& Ac, u32x4& An, int k) {
                const int kn = k + 1 > run_hi ? run_hi : k + 1;
                const int ka = k + 1 < k1 ? k + 1 : k;
#pragma unroll
                for (int r = 0; r < 3; ++r) Nx[r] = tr_run(xb[r] + kn * 512);
                An = tr_run(ab + ka * 512);
                __builtin_amdgcn_sched_barrier(0);       // the reads of step k+1 go out BEFORE the MFMAs of step k
                const h8_t af = as_h8(Ac);
                const unsigned mnext = k == Qh - 1 ? mhi : 0xffffffffu;   // k = Qh - 1: only element 0 of the next run is used
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc[r * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(shift_m1(P[r], Cc[r])), acc[r * 3 + 0], 0, 0, 0);
                    acc[r * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(Cc[r]), acc[r * 3 + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);       // six MFMAs that do not need the new runs cover the LDS latency
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    u32x4 nx = Nx[r];
                    nx[0] &= mnext;
                    acc[r * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, as_h8(shift_p1(Cc[r], nx)), acc[r * 3 + 2], 0, 0, 0);
                }
            };
            for (int k = k0; k < k1; k += 6) {           // 6 = lcm of the two rotations
                step(R0, R1, R2, A0, A1, k);
                if (k + 1 < k1) step(R1, R2, R0, A1, A0, k + 1);
                if (k + 2 < k1) step(R2, R0, R1, A0, A1, k + 2);
                if (k + 3 < k1) step(R0, R1, R2, A1, A0, k + 3);
                if (k + 4 < k1) step(R1, R2, R0, A0, A1, k + 4);
                if (k + 5 < k1) step(R2, R0, R1, A1, A0, k + 5);
            }
        } else {
            const char* xb = ldsX + sx_now * slotb + ntl * rowb + lane_off;
            u32x4 A0 = tr_run(ab + k0 * 512), B0 = tr_run(xb + k0 * 512), A1, B1;
            auto step = [&](const u32x4& Ac, const u32x4& Bc, u32x4& An, u32x4& Bn, int k) {
                const int ka = k + 1 < k1 ? k + 1 : k;
                An = tr_run(ab + ka * 512);
                Bn = tr_run(xb + ka * 512);
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(Ac), as_h8(Bc), acc[0], 0, 0, 0);
            };
            for (int k = k0; k < k1; k += 2) {
                step(A0, B0, A1, B1, k);
                if (k + 1 < k1) step(A1, B1, A0, B0, k + 1);
            }
        }
    }
    // the two halves of the k range meet in LDS (the row rings are dead): waves 4-7 park their tiles, waves 0-3 add them to
    // their own and write ONE partial tile set per slice (fixed order: deterministic)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // no request or read may outlive the rings
    float* park = reinterpret_cast<float*>(smem) + (size_t)(wave & 3) * NT * 16 * 64;
    if (kpart == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) park[(t * 16 + q) * 64 + lane] = acc[t][q];
    }
    __syncthreads();
    if (kpart == 1 || !active) return;
    const int n_out = n2 * 64 + ntl * 32 + l31;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float v = acc[t][q] + park[(t * 16 + q) * 64 + lane];
            const int m_out = m2 * 64 + mtl * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
            if (m_out < a.M && n_out < a.N) a.ws[(((size_t)ks * NT + t) * a.M + m_out) * a.N + n_out] = v;
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

size_t wgrad_nhwc_lds_bytes(int ksize, int Q, int da, int dx) { return (size_t)(da + 1 + dx + (ksize == 3 ? 3 : 1)) * 2 * Q * 64; }

}  // namespace

bool wgrad_nhwc_view_ok(const y6_tensor& t) {
    return t.data && t.C % 8 == 0 && t.cstride % 8 == 0 && t.coff % 8 == 0 && (((uintptr_t)t.data) & 15) == 0 &&
           (size_t)t.B * t.H * t.W * t.cstride * 2 < 0xf0000000ull;
}

// wgrad_flat.hip: the flat-index block-tiled kernel (round 6)
const char* wgrad_flat_unsupported(const y6_wgrad_nhwc_desc* d);
int wgrad_flat_launch(const y6_wgrad_nhwc_desc* d, hipStream_t s);

namespace {

// Which NHWC-fed kernel takes a conv: the flat-index kernel wherever its stages fit the LDS and the map is at most
// Y6_WGRAD_FLAT_MAXW columns wide (default: every fitting map), the row-ring kernel otherwise.  Y6_WGRAD_FLAT=0: row ring only (A/B).
bool use_flat(const y6_wgrad_nhwc_desc* d) {
    const char* e = getenv("Y6_WGRAD_FLAT");            // read per call: tests and tools/wgrad_bench.py flip it inside one process
    const char* w = getenv("Y6_WGRAD_FLAT_MAXW");
    if (!d || (e && atoi(e) == 0) || wgrad_flat_unsupported(d) != nullptr) return false;
    if (d->stride == 2) return true;                    // the row ring has no stride-2 form
    if (e && atoi(e) == 1) return true;                 // forced (tests, tools/wgrad_bench.py)
    // measured on the YOLOv6-S b64 shapes (profiles/r06/wgrad_bench_r06b.json): the flat kernel wins on every map up to 80 wide
    // with more than 64 couts (256 -> 256 @40: 169 against 265 us, 512 -> 512 @20: 172 against 431, 1x1s 23-58 against 46-245); the
    // row ring keeps the 160-wide maps (the flat stage's halo is 2*Wp + 24 positions per 128) and the 64-cout layers of the 80-wide
    // maps (a 64-cout flat block is four waves alone on a CU there: 97 against 68 us)
    const int maxw = w ? atoi(w) : 100;
    if (d->x.W > maxw) return false;
    if (d->M <= 64 && d->x.W >= 64) return false;
    return true;
}

const char* wgrad_nhwc_unsupported(const y6_wgrad_nhwc_desc* d) {
    if (use_flat(d)) return nullptr;
    if (d && d->stride == 2) return "stride 2 needs the flat-index kernel (map too wide for its stage, or Y6_WGRAD_FLAT=0)";
    if (!d || !d->out) return "null argument";
    if (d->ksize != 1 && d->ksize != 3) return "ksize must be 1 or 3 (stride 1)";
    if (!wgrad_nhwc_view_ok(d->dy) || !wgrad_nhwc_view_ok(d->x)) return "views must be fp16 NHWC, 8-channel / 16-byte aligned, below 3.75 GiB";
    if (d->dy.B != d->x.B || d->dy.H != d->x.H || d->dy.W != d->x.W || d->x.B < 1 || d->x.H < 1 || d->x.W < 1) return "dy and x must have one spatial shape";
    if (d->M < 1 || d->N < 1 || d->dy.C < d->M || d->x.C < d->N) return "views narrower than M / N";
    const int Q = (d->x.W + 15) / 16 * 16;
    if (wgrad_nhwc_lds_bytes(d->ksize, Q, 1, 1) > 160 * 1024) return "row too wide for the LDS row ring";
    return nullptr;
}

int wgrad_nhwc_launch(const y6_wgrad_nhwc_desc* d, hipStream_t s) {
    if (use_flat(d)) return wgrad_flat_launch(d, s);
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
    // one block per CU, one round: every slice costs two partial tiles (the k halves) that the reduction reads again
    // (three rounds: 216 MB of partials per 3x3 64 -> 64 launch, more time than its MFMAs - profiles/r03/r03n_wgrad_bench.json)
    long nsplit = 256 / tiles;
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
    static const bool probing = getenv("Y6_WGRAD_PROBE") != nullptr;
    a.dbg = probing && getenv("Y6_WGRAD_DBG") ? atoi(getenv("Y6_WGRAD_DBG")) : 0;
    // requests run two rows ahead where the LDS holds the deeper rings (one row of requests in flight per CU: 3.4 TB/s)
    a.da = a.dx = 1;
    if (wgrad_nhwc_lds_bytes(d->ksize, a.Q, 2, 2) < 160 * 1024) a.da = a.dx = 2;
    else if (wgrad_nhwc_lds_bytes(d->ksize, a.Q, 2, 1) < 160 * 1024) a.da = 2;
    size_t lds = wgrad_nhwc_lds_bytes(d->ksize, a.Q, a.da, a.dx);
    const size_t park = (size_t)4 * T * 16 * 64 * sizeof(float);      // where the two k halves of the four tiles meet
    if (lds < park) lds = park;
    const unsigned grid = (unsigned)(8 * (((long)a.nsplit * tiles + 7) / 8));
    static bool big3 = false, big1 = false;
    if (d->ksize == 3) {
        if (!big3) {
            Y6_HIP(hipFuncSetAttribute((const void*)wgrad_lds_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            big3 = true;
        }
        hipLaunchKernelGGL(wgrad_lds_kernel<3>, dim3(grid), dim3(512), lds, s, a);
    } else {
        if (!big1) {
            Y6_HIP(hipFuncSetAttribute((const void*)wgrad_lds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            big1 = true;
        }
        hipLaunchKernelGGL(wgrad_lds_kernel<1>, dim3(grid), dim3(512), lds, s, a);
    }
    Y6_LAUNCH_CHECK();
    unsigned rg = (unsigned)((per + 255) / 256);
    if (rg > 4096) rg = 4096;
    const int nparts = a.nsplit;
    const int nchunk = (nparts + kRedChunk - 1) / kRedChunk;
    float* ws2 = a.ws + (size_t)nparts * per;
    hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3(rg, (unsigned)nchunk), dim3(256), 0, s, a.ws, nparts, per, ws2);
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

// 0: no NHWC-fed kernel takes the descriptor (the caller keeps the plane-fed y6_wgrad), 1: the flat-index kernel, 2: the row ring
extern "C" int y6_wgrad_nhwc_route(const y6_wgrad_nhwc_desc* d) {
    if (wgrad_nhwc_unsupported(d) != nullptr) return 0;
    return use_flat(d) ? 1 : 2;
}

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
