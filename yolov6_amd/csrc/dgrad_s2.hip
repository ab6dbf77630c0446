// dgrad_s2.hip - data gradient of a 3x3 stride-2 conv (+ the 1x1 stride-2 conv of the same input) from the COMPACT output gradient.
//
// Replaces autograd's conv backward (input half) for the stride-2 blocks of the training step: RepVGGBlock(k3 s2) in train form
// (reference yolov6/layers/common.py:250-255: rbr_dense 3x3 s2 + rbr_1x1 1x1 s2, no identity), ConvBNReLU / ConvBNSiLU(k3 s2)
// (common.py:26-94), under scaler.scale(loss).backward() (yolov6/core/engine.py:173).
//
// Until round 6 the plan ran this as the ordinary stride-1 conv over a ZERO-INSERTED gradient (dy written at (2y, 2x) of a
// full-resolution buffer): four times the multiply-adds, four times the gradient bytes, and a second full read-modify-write of dx for
// the 1x1 branch - 1.75 ms of a 32 ms YOLOv6-S step for 0.24 TFLOP of useful work.  Here:
//
//   forward   y[i][j] = sum_{ky,kx} W[ky][kx] x[2i+ky-1][2j+kx-1]
//   backward  dx[2I+a][2J+b] = sum over the taps whose parity matches (a, b):
//        (a,b) = (0,0): W[1][1]^T dy[I][J]                                  (+ W1^T dy1[I][J], the 1x1 branch)
//                (0,1): W[1][2]^T dy[I][J]   + W[1][0]^T dy[I][J+1]
//                (1,0): W[2][1]^T dy[I][J]   + W[0][1]^T dy[I+1][J]
//                (1,1): W[2][2]^T dy[I][J]   + W[2][0]^T dy[I][J+1] + W[0][2]^T dy[I+1][J] + W[0][0]^T dy[I+1][J+1]
//   nine (ten) tap GEMMs over the compact gradient, every dx element written exactly once (or added to what dx holds).
//
// Kernel: one block = (TH x TW compact positions, CF x 32 input channels of the forward conv); the (TH+1) x (TW+1) halo of dy is
// staged per 32-channel chunk in LDS (80-byte pixel pitch, as conv_mfma.hip), the chunk's ten weight images arrive by LDS-DMA from the
// data-gradient pack the plan already keeps (y6_pack_job kind 1: W'[ci][co][2-ky][2-kx]); a wave holds four accumulator sets (one
// per output parity) of CF x PF fragments and issues 9 (10) x CF x PF x 2 MFMAs per chunk between two barriers; each of the four
// shifted pixel fragments of a chunk is read once and serves every tap of its shift.  Epilogue: the output rows of one parity staged
// in LDS and written (or read, added and written: the two roundings of the accumulating convs) as contiguous 16-byte pieces.
// HBM-bound for the wide early layers (the 32->64 block: 0.21 GB in, 0.42 GB out), MFMA-leaning for the 256->512 one.
#include "conv_common.hpp"
#include "plan_internal.hpp"

namespace {

struct DgS2Args {
    ConvKArgs k;                   // what the epilogue reads: out = dx, res = dx when accumulating, Cout = N, store-width flags
    const __half* dy3;
    const __half* dy1;             // nullptr: no 1x1 branch
    const __half* w3;
    const __half* w1;
    int d3_cs, d3_co, d1_cs, d1_co;
    int Ho, Wo;
    int TH, TW, tiles_x, tiles_y, ntiles, ncb, nchunk;
    int HH, HWd;
    int ldsA_bytes, ldsB_bytes;
};

template <int BP>                  // halo pixels of a block of BP pixel slots: (TH + 1) x (TW + 1) over the tiles choose_tile() may pick
struct S2HaloCap {
    static constexpr int value = BP == 128 ? 176 : 304;
};

// NW waves per block: 4 (two blocks per CU; the default) or 8 (one block: twice the pixels behind one set of weight images)
template <int CF, int PF, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void dgrad_s2_kernel(const DgS2Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WCF = 10 * 2048;                       // bytes of one cout fragment's ten tap images (2 k-steps x 1 KiB each)
    constexpr int NT = NW * 64;                          // threads
    constexpr int BP = NW * PF * 32;                     // pixel slots of the block
    constexpr int MAXHP = S2HaloCap<BP>::value;
    constexpr int NP = (MAXHP * 4 + NT - 1) / NT;        // 16-byte halo pieces per thread
    constexpr int NPB = BP * 4 / NT;                     // ... of the 1x1 branch's tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool has1 = g.dy1 != nullptr;

    int tile, cb;
    {
        const int id = blockIdx.x;
        if (g.ncb == 1) {
            tile = id;
            cb = 0;
        } else {                     // the cout blocks of one tile sit on one XCD (id % 8) and share its L2 for the halo
            const int lo = id & 7, r = id >> 3;
            cb = r % g.ncb;
            tile = (r / g.ncb) * 8 + lo;
        }
    }
    if (tile >= g.ntiles) return;
    const int tx_i = tile % g.tiles_x;
    const int t2 = tile / g.tiles_x;
    const int ty_i = t2 % g.tiles_y;
    const int b = t2 / g.tiles_y;
    const int oy0 = ty_i * g.TH, ox0 = tx_i * g.TW;

    char* ldsA = smem;
    char* ldsB = smem + g.ldsA_bytes;
    char* ldsW = ldsB + g.ldsB_bytes;

    // per-thread piece tables (element offsets; -1: zero fill, -2: no such piece)
    const int npieces = g.HH * g.HWd * 4;
    int goff[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = tid + i * NT;
        int o = -2;
        if (idx < npieces) {
            const int hp = idx >> 2, q = idx & 3;
            const int hy = hp / g.HWd, hx = hp - hy * g.HWd;
            const int iy = oy0 + hy, ix = ox0 + hx;
            o = (iy < g.Ho && ix < g.Wo) ? (((b * g.Ho + iy) * g.Wo + ix) * g.d3_cs + g.d3_co + q * 8) : -1;
        }
        goff[i] = o;
    }
    const int npx = g.TH * g.TW;
    int boff[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int idx = tid + i * NT;
        int o = -2;
        if (has1 && idx < npx * 4) {
            const int p = idx >> 2, q = idx & 3;
            const int ty = p / g.TW, tx = p - ty * g.TW;
            const int iy = oy0 + ty, ix = ox0 + tx;
            o = (iy < g.Ho && ix < g.Wo) ? (((b * g.Ho + iy) * g.Wo + ix) * g.d1_cs + g.d1_co + q * 8) : -1;
        }
        boff[i] = o;
    }

    // per-lane pixel operand addressing and the output position of class (0,0)
    int pixoff[PF], pixoffB[PF], obase[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int m = wave * (PF * 32) + pf * 32 + (lane & 31);
        bool v = m < npx;
        const int mm = v ? m : npx - 1;
        const int ty = mm / g.TW, tx = mm - ty * g.TW;
        const int oy = oy0 + ty, ox = ox0 + tx;
        v = v && (oy < g.Ho) && (ox < g.Wo);
        pixoff[pf] = (ty * g.HWd + tx) * PIXB + (lane >> 5) * 16;
        pixoffB[pf] = mm * PIXB + (lane >> 5) * 16;
        obase[pf] = v ? ((b * 2 * g.Ho + 2 * oy) * (2 * g.Wo) + 2 * ox) : -1;
    }

    f32x16_t acc[4][CF][PF];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][cf][pf][r] = 0.f;

    // the ten tap images of a chunk: nine contiguous KiB pairs per cout fragment of the 3x3 pack, one pair of the 1x1 pack
    auto issue_w = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < (CF * 18 + NW - 1) / NW; ++j) {
            const int p = wave + NW * j;
            if (p < CF * 18) {
                const int cf = p / 18, r = p - cf * 18;
                const __half* src = g.w3 + ((size_t)(cb * CF + cf) * g.nchunk + chunk) * (9 * 1024) + r * 512 + lane * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ldsW + cf * WCF + r * 1024), 16, 0, 0);
            }
        }
        if (has1) {
            const int p = wave;
            if (p < CF * 2) {
                const int cf = p >> 1, ks = p & 1;
                const __half* src = g.w1 + ((size_t)(cb * CF + cf) * g.nchunk + chunk) * 1024 + ks * 512 + lane * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ldsW + cf * WCF + (18 + ks) * 1024), 16, 0, 0);
            }
        }
    };
    auto load_A = [&](int chunk, uint4 (&ra)[NP], uint4 (&rb)[NPB]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (goff[i] >= 0) v = *reinterpret_cast<const uint4*>(g.dy3 + goff[i] + chunk * 32);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (boff[i] >= 0) v = *reinterpret_cast<const uint4*>(g.dy1 + boff[i] + chunk * 32);
            rb[i] = v;
        }
    };
    auto store_A = [&](const uint4 (&ra)[NP], const uint4 (&rb)[NPB]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + i * NT;
            if (goff[i] != -2) *reinterpret_cast<uint4*>(ldsA + (idx >> 2) * PIXB + (idx & 3) * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int idx = tid + i * NT;
            if (boff[i] != -2) *reinterpret_cast<uint4*>(ldsB + (idx >> 2) * PIXB + (idx & 3) * 16) = rb[i];
        }
    };

    uint4 areg[NP], breg[NPB];
    load_A(0, areg, breg);
    issue_w(0);
    store_A(areg, breg);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const char* wb = ldsW + lane * 16;
    for (int chunk = 0; chunk < g.nchunk; ++chunk) {
        const bool more = (chunk + 1) < g.nchunk;
        if (more) load_A(chunk + 1, areg, breg);      // register prefetch of the next chunk's pixels, in flight under the MFMAs
        // The four shifted pixel fragments of this chunk are each read once and used by every tap of their shift.
        // Weight fragments are requested one tap AHEAD of the MFMAs that use them (nxt), the pixel fragments of a shift when the
        // last tap of the previous shift goes out: the matrix pipe does not wait a full LDS round trip in front of every tap.
        h8_t bfr[PF][2], bnx[PF][2];
        h8_t cur[CF][2], nxt[CF][2];
#define Y6_S2_LOADW(DST, TAU)                                                                                               \
    _Pragma("unroll") for (int cf = 0; cf < CF; ++cf) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                      \
        DST[cf][ks] = *reinterpret_cast<const h8_t*>(wb + cf * WCF + ((TAU) * 2 + ks) * 1024);                              \
    __builtin_amdgcn_sched_barrier(0);     /* the scheduler would sink the reads back in front of their use */
#define Y6_S2_LOADB(DST, D)                                                                                                 \
    _Pragma("unroll") for (int pf = 0; pf < PF; ++pf) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                      \
        DST[pf][ks] = *reinterpret_cast<const h8_t*>(ldsA + pixoff[pf] + (((D) >> 1) * g.HWd + ((D) & 1)) * PIXB + ks * 32); \
    __builtin_amdgcn_sched_barrier(0);
#define Y6_S2_MFMA(CLS)                                                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int cf = 0; cf < CF; ++cf)                      \
        _Pragma("unroll") for (int pf = 0; pf < PF; ++pf)                                                                   \
            acc[CLS][cf][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[cf][ks], bfr[pf][ks], acc[CLS][cf][pf], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
#define Y6_S2_ROLL()                                                                                                        \
    _Pragma("unroll") for (int cf = 0; cf < CF; ++cf) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) cur[cf][ks] = nxt[cf][ks];
#define Y6_S2_ROLLB()                                                                                                       \
    _Pragma("unroll") for (int pf = 0; pf < PF; ++pf) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) bfr[pf][ks] = bnx[pf][ks];
        // (shift index = 2 * (row shift) + (column shift); image of the data-gradient pack = 8 - (ky*3 + kx))
        Y6_S2_LOADB(bfr, 0)                       // dy[I][J]
        Y6_S2_LOADW(cur, 4)
        Y6_S2_LOADW(nxt, 3)  Y6_S2_MFMA(0)  Y6_S2_ROLL()             //   (0,0): W[1][1]
        Y6_S2_LOADW(nxt, 1)  Y6_S2_MFMA(1)  Y6_S2_ROLL()             //   (0,1): W[1][2]
        Y6_S2_LOADW(nxt, 0)  Y6_S2_MFMA(2)  Y6_S2_ROLL()             //   (1,0): W[2][1]
        Y6_S2_LOADW(nxt, 5)  Y6_S2_LOADB(bnx, 1)  Y6_S2_MFMA(3)  Y6_S2_ROLL()  Y6_S2_ROLLB()   //   (1,1): W[2][2]; next: dy[I][J+1]
        Y6_S2_LOADW(nxt, 2)  Y6_S2_MFMA(1)  Y6_S2_ROLL()             //   (0,1): W[1][0]
        Y6_S2_LOADW(nxt, 7)  Y6_S2_LOADB(bnx, 2)  Y6_S2_MFMA(3)  Y6_S2_ROLL()  Y6_S2_ROLLB()   //   (1,1): W[2][0]; next: dy[I+1][J]
        Y6_S2_LOADW(nxt, 6)  Y6_S2_MFMA(2)  Y6_S2_ROLL()             //   (1,0): W[0][1]
        Y6_S2_LOADW(nxt, 8)  Y6_S2_LOADB(bnx, 3)  Y6_S2_MFMA(3)  Y6_S2_ROLL()  Y6_S2_ROLLB()   //   (1,1): W[0][2]; next: dy[I+1][J+1]
        Y6_S2_MFMA(3)                                                //   (1,1): W[0][0]
#undef Y6_S2_LOADW
#undef Y6_S2_LOADB
#undef Y6_S2_MFMA
#undef Y6_S2_ROLL
#undef Y6_S2_ROLLB
        if (has1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8_t b1[PF];
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) b1[pf] = *reinterpret_cast<const h8_t*>(ldsB + pixoffB[pf] + ks * 32);
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
                    const h8_t af = *reinterpret_cast<const h8_t*>(wb + cf * WCF + (18 + ks) * 1024);
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf)
                        acc[0][cf][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b1[pf], acc[0][cf][pf], 0, 0, 0);
                }
            }
        }
        // every wave's LDS reads of this chunk are done: the weight images and the pixel stages may be overwritten
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (more) {
            issue_w(chunk + 1);
            store_A(areg, breg);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }

    // Epilogue: the block's output is a dense (2 TH) x (2 TW) patch of dx, but a lane's accumulators are single pixels two apart -
    // stored as they lie they were 16-byte pieces scattered over as many cache lines (the first form of this kernel spent 2/3 of
    // its time here: 440 us with, 133 us without the epilogue on the 32-channel layer).  So the two classes of one output-row
    // parity are rounded to fp16 into an LDS image of their TH output rows ([row][2 TW pixels][CF*32 channels], the main loop's
    // LDS is dead behind its last barrier), and the block writes - or reads, adds and writes - those rows as contiguous 16-byte
    // pieces, consecutive lanes on consecutive addresses.  Same arithmetic as the accumulating convs: fp16(acc), + old, fp16.
    {
        constexpr int CFC = CF * 32;                   // channels of the block
        constexpr int RSB = CFC * 2 + 16;              // bytes per staged pixel (16 B of pad: bank spread)
        constexpr int PPX = CF * 4;                    // 16-byte pieces per pixel
        const int W2 = 2 * g.TW;
        const int npiece = g.TH * W2 * PPX;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (a) __syncthreads();                    // the rows of parity 0 have left the stage
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) {
                    const int m = wave * (PF * 32) + pf * 32 + (lane & 31);
                    if (m < npx) {
                        const int ty = m / g.TW, tx = m - ty * g.TW;
                        char* dst = smem + (ty * W2 + 2 * tx + bb) * RSB + (lane >> 5) * 8;
#pragma unroll
                        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
                                h4_t o;
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[j] = (_Float16)acc[a * 2 + bb][cf][pf][gq * 4 + j];
                                *reinterpret_cast<h4_t*>(dst + cf * 64 + gq * 16) = o;
                            }
                    }
                }
            __syncthreads();
            for (int q = tid; q < npiece; q += NT) {
                const int pix = q / PPX, pc = q - pix * PPX;
                const int ty = pix / W2, ox = pix - ty * W2;
                const int y = 2 * (oy0 + ty) + a, x = 2 * ox0 + ox;
                if (oy0 + ty >= g.Ho || x >= 2 * g.Wo) continue;
                h8_t v = *reinterpret_cast<const h8_t*>(smem + pix * RSB + pc * 16);
                __half* gp = g.k.out + ((size_t)(b * 2 * g.Ho + y) * (2 * g.Wo) + x) * g.k.out_cs + g.k.out_co + cb * CFC + pc * 8;
                if (g.k.res != nullptr) {
                    const h8_t old = *reinterpret_cast<const h8_t*>(gp);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (_Float16)((float)v[j] + (float)old[j]);
                }
                *reinterpret_cast<h8_t*>(gp) = v;
            }
        }
    }
}

// TH x TW compact positions for a block of `bp` pixel slots whose halo fits `cap` pixels: fewest pixel slots over the map first
// (a block's time is its slots), then the widest tile (longer contiguous runs in both tensors)
void choose_tile(int Ho, int Wo, int bp, int cap, int* th, int* tw) {
    long best = -1;
    *th = 1;
    *tw = 1;
    for (int TW = 1; TW <= Wo && TW <= bp; ++TW) {
        int TH = bp / TW;
        if (TH > Ho) TH = Ho;
        while (TH > 1 && (TH + 1) * (TW + 1) > cap) --TH;
        if ((TH + 1) * (TW + 1) > cap) continue;
        const long slots = (long)y6_cdiv(Ho, TH) * y6_cdiv(Wo, TW) * bp;
        if (best < 0 || slots < best || (slots == best && TW > *tw)) {
            best = slots;
            *th = TH;
            *tw = TW;
        }
    }
}

bool view16(const y6_tensor& t) {
    return t.data != nullptr && t.B > 0 && t.H > 0 && t.W > 0 && t.C > 0 && t.cstride % 8 == 0 && t.coff % 8 == 0 && t.coff >= 0 &&
           t.coff + t.C <= t.cstride && ((uintptr_t)t.data & 15) == 0;
}

int check(const y6_dgrad_s2_desc* d) {
    Y6_REQUIRE(d, "dgrad_s2: null argument");
    Y6_REQUIRE(view16(d->dy3) && view16(d->dx), "dgrad_s2: dy3 / dx must be 16-byte aligned fp16 NHWC views with 8-channel alignment");
    Y6_REQUIRE(d->dx.B == d->dy3.B && d->dx.H == 2 * d->dy3.H && d->dx.W == 2 * d->dy3.W,
               "dgrad_s2: dx must be [B, 2*Ho, 2*Wo, N] for dy3 [B, Ho, Wo, M] (even input sizes)");
    Y6_REQUIRE(d->dy3.C % 32 == 0 && d->dx.C % 32 == 0, "dgrad_s2: channel counts must be multiples of 32 (M = %d, N = %d)", d->dy3.C, d->dx.C);
    Y6_REQUIRE(d->w3_packed != nullptr, "dgrad_s2: no packed 3x3 weight");
    if (d->dy1.data != nullptr) {
        Y6_REQUIRE(view16(d->dy1) && d->dy1.B == d->dy3.B && d->dy1.H == d->dy3.H && d->dy1.W == d->dy3.W && d->dy1.C == d->dy3.C,
                   "dgrad_s2: dy1 must have the shape of dy3");
        Y6_REQUIRE(d->w1_packed != nullptr, "dgrad_s2: dy1 without a packed 1x1 weight");
    }
    Y6_REQUIRE((size_t)d->dy3.B * d->dy3.H * d->dy3.W * d->dy3.cstride < (1ull << 31) &&
                   (size_t)d->dx.B * d->dx.H * d->dx.W < (1ull << 31),
               "dgrad_s2: tensor too large for 32-bit element offsets");
    return Y6_OK;
}

template <int CF, int PF, int NW>
int launch_cfg(const y6_dgrad_s2_desc* d, hipStream_t s) {
    DgS2Args g;
    memset(&g, 0, sizeof(g));
    const int M = d->dy3.C, N = d->dx.C;
    g.k.out = (__half*)d->dx.data;
    g.k.out_cs = d->dx.cstride;
    g.k.out_co = d->dx.coff;
    if (d->accumulate) {
        g.k.res = (const __half*)d->dx.data;
        g.k.res_cs = d->dx.cstride;
        g.k.res_co = d->dx.coff;
        g.k.res_vec = 1;
    }
    g.k.Cout = N;
    g.k.Cin = M;
    g.k.act = Y6_ACT_NONE;
    g.k.vec_ok = 1;
    g.k.vec16_ok = 1;
    g.dy3 = (const __half*)d->dy3.data;
    g.d3_cs = d->dy3.cstride;
    g.d3_co = d->dy3.coff;
    g.w3 = (const __half*)d->w3_packed;
    if (d->dy1.data != nullptr) {
        g.dy1 = (const __half*)d->dy1.data;
        g.d1_cs = d->dy1.cstride;
        g.d1_co = d->dy1.coff;
        g.w1 = (const __half*)d->w1_packed;
    }
    g.Ho = d->dy3.H;
    g.Wo = d->dy3.W;
    choose_tile(g.Ho, g.Wo, NW * PF * 32, S2HaloCap<NW * PF * 32>::value, &g.TH, &g.TW);
    g.tiles_x = y6_cdiv(g.Wo, g.TW);
    g.tiles_y = y6_cdiv(g.Ho, g.TH);
    g.ntiles = g.tiles_x * g.tiles_y * d->dy3.B;
    g.ncb = N / (CF * 32);
    g.nchunk = M / 32;
    g.HH = g.TH + 1;
    g.HWd = g.TW + 1;
    g.ldsA_bytes = (g.HH * g.HWd * PIXB + 15) & ~15;
    g.ldsB_bytes = g.dy1 ? ((g.TH * g.TW * PIXB + 15) & ~15) : 0;
    size_t lds = (size_t)g.ldsA_bytes + g.ldsB_bytes + (size_t)CF * 10 * 2048;
    const size_t stage = (size_t)g.TH * 2 * g.TW * (CF * 64 + 16);     // the epilogue's image of one row parity
    if (lds < stage) lds = stage;
    static bool big = false;            // one attribute call per instantiation (benign if two threads race: same value)
    if (lds > 64 * 1024 && !big) {
        Y6_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_s2_kernel<CF, PF, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        big = true;
    }
    Y6_REQUIRE(lds <= 96 * 1024, "dgrad_s2: %zu bytes of LDS", lds);
    const int grid = g.ncb == 1 ? g.ntiles : y6_cdiv(g.ntiles, 8) * 8 * g.ncb;
    hipLaunchKernelGGL((dgrad_s2_kernel<CF, PF, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, s, g);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

int dgrad_s2_launch(const y6_dgrad_s2_desc* d, hipStream_t s) {
    if (int rc = check(d)) return rc;
    // Y6_DGRAD_S2_WAVES=8 (A/B): 256-pixel blocks of eight waves - half the weight traffic per pixel, measured no faster
    // (0.94 against 0.905 ms for the eight launches of a YOLOv6-S step [GPU r06zh]): the launch is its phases in a row, not L2-bound
    static const int waves = getenv("Y6_DGRAD_S2_WAVES") ? atoi(getenv("Y6_DGRAD_S2_WAVES")) : 4;
    if (d->dx.C % 64 == 0) return waves == 8 ? launch_cfg<2, 1, 8>(d, s) : launch_cfg<2, 1, 4>(d, s);
    return launch_cfg<1, 2, 4>(d, s);
}

}  // namespace

extern "C" int y6_dgrad_s2_supported(const y6_dgrad_s2_desc* d) { return check(d) == Y6_OK ? 1 : 0; }

extern "C" int y6_dgrad_s2(const y6_dgrad_s2_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    return dgrad_s2_launch(d, (hipStream_t)stream);
}

extern "C" int y6_plan_add_dgrad_s2(y6_plan* p, const y6_dgrad_s2_desc* d) {
    if (int rc = check(d)) return rc;
    const double px = (double)d->dy3.B * d->dy3.H * d->dy3.W;
    const int taps = 9 + (d->dy1.data != nullptr ? 1 : 0);
    const double flops = 2.0 * px * d->dy3.C * d->dx.C * taps;
    const double bytes = 2.0 * px * d->dy3.C * (d->dy1.data != nullptr ? 2 : 1) + 2.0 * 4.0 * px * d->dx.C * (d->accumulate ? 2 : 1);
    return y6_plan_push(p, dgrad_s2_launch, d, Y6_TOP_DGRAD_S2, flops, bytes);
}
