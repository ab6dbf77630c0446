// conv_wreg.hip - 3x3 stride-1 fp16 convolution, weights through REGISTERS, halo through LDS in 32-channel stages.
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
//   * vector-memory ordering is hand-counted: every VMEM instruction of the main loop is inline asm (weight loads, LDS-DMA),
//     issued in a fixed pattern (a burst of kMaxP halo requests per stage - dummies beyond the image - then one weight load
//     per unit), so `s_waitcnt vmcnt(N)` in front of a unit's MFMAs names exactly the loads that may stay in flight.
// The epilogue is conv_common.hpp's (bias, post-affine, activation, residual, ragged stores), run at the end of an item while
// the SIMD's other wave (the CU's second block) keeps the matrix pipe busy.
#include "common.hpp"
#include "conv_common.hpp"

namespace {

constexpr int kPix = 80;     // LDS bytes per halo pixel: 32 channels (64 B) + one 16-byte pad slot
constexpr int kMaxP = 10;    // halo requests (1 KiB each) per wave and stage - always issued, so that vmcnt arithmetic is static
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

template <int PF, int WC, int WP>
__global__ __launch_bounds__(WC * WP * 64, 2) void conv3x3_wreg_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WC * WP;
    constexpr int R = kRing;
    static_assert(kUnits % R == 0, "ring slots are compile-time indices");
    static_assert(R - 1 + kMaxP <= 63, "vmcnt is a 6-bit counter");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WC, wp = wave / WC;
    const int RP = a.dma_rp, NHP = a.dma_nhp;
    const unsigned stage_bytes = (unsigned)NHP * 1024u;
    const unsigned smem_base = lds_addr(smem);
    const unsigned dummy_dst = smem_base + 2u * stage_bytes;   // 1 KiB nobody reads
    const int nsc = a.Cin >> 5;
    const int nids = a.nids;
    const int gstride = gridDim.x;
    const int ics = a.in_cs;

    const i32x4_t rsA = make_rsrc(a.in, (unsigned)((size_t)a.B * a.H * a.W * ics * 2));
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
    {
        int t;
        decode(id, t, cb);   // the grid stride is a multiple of 8 * ncb: every item of this block has this cout block
    }

    // ---- this wave's halo requests: slot s = 64 * P + lane of the stage image -> (halo row, halo column, piece), fixed for the
    //      whole kernel (the divisions happen once); per item only the tile origin changes
    unsigned hinfo[kMaxP];
#pragma unroll
    for (int i = 0; i < kMaxP; ++i) {
        const int P = wave + NW * i;
        const int s = P * 64 + lane;
        const int p = s / 5, j = s - 5 * p;
        const int hy = p / RP, hx = p - hy * RP;
        const bool v = (P < NHP) && (j < 4) && (hx < a.HWd) && (hy < a.HH);
        hinfo[i] = v ? (unsigned)((hy << 16) | (hx << 3) | j) : 0xffffffffu;
    }
    // requests of one stage: the item's tile origin (iy0, ix0, byte offset of its pixel (0, 0)), the stage's channel offset
    auto issue_halo = [&](bool real, int iy0, int ix0, unsigned base, unsigned soff, unsigned dst0) {
#pragma unroll
        for (int i = 0; i < kMaxP; ++i) {
            const int P = wave + NW * i;
            const int hy = (int)(hinfo[i] >> 16), hx = (int)((hinfo[i] & 0xffffu) >> 3), j = (int)(hinfo[i] & 7u);
            const bool v = real && (hinfo[i] != 0xffffffffu) && ((unsigned)(iy0 + hy) < (unsigned)a.H) && ((unsigned)(ix0 + hx) < (unsigned)a.W);
            const unsigned voff = v ? base + (unsigned)(hy * a.W + hx) * (unsigned)(ics * 2) + (unsigned)(j * 16) : kOob;
            dma16(rsA, voff, soff, (real && P < NHP) ? dst0 + (unsigned)P * 1024u : dummy_dst);
        }
    };
    auto tile_origin = [&](int item, int& iy0, int& ix0, unsigned& base) {
        int tile, c;
        decode(item, tile, c);
        const int tx_i = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty_i = t2 % a.tiles_y;
        const int b = t2 / a.tiles_y;
        iy0 = ty_i * a.TH - 1;
        ix0 = tx_i * a.TW - 1;
        // modulo 2^32 (tensors up to 3.5 GiB): the origin may lie one row / column outside the image
        base = (((unsigned)(b * a.H + iy0) * (unsigned)a.W + (unsigned)ix0) * (unsigned)ics + (unsigned)a.in_co) * 2u;
    };

    // ---- this lane's pixels: position in the tile (fixed for the whole kernel), LDS offset of the tap (0, 0) read
    const int fq = frag_pixel(lane & 31);
    unsigned pixaddr[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int m = wp * (PF * 32) + pf * 32 + fq;
        const int npx = a.TH * a.TW;
        const int mm = m < npx ? m : npx - 1;
        const int ty = mm / a.TW, tx = mm - ty * a.TW;
        pixaddr[pf] = (unsigned)((ty * RP + tx) * kPix + (lane >> 5) * 16);
    }
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
        for (int pf = 0; pf < PF; ++pf) {   // (the divisions again, once per item: eight registers the main loop does not carry)
            const int m = wp * (PF * 32) + pf * 32 + fq;
            const int ty = m / ea.TW, tx = m - ty * ea.TW;
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

    f32x16_t acc[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pf][r] = 0.f;

    // prologue: the first stage of the first item, the first R - 1 weight fragments
    int n_item = id;          // item whose stage is requested next
    int n_sc = 0;
    {
        int iy0, ix0;
        unsigned base;
        tile_origin(n_item, iy0, ix0, base);
        issue_halo(true, iy0, ix0, base, 0u, smem_base);
    }
#pragma unroll
    for (int u = 0; u < R - 1; ++u) load_frag(wr[u], rsW, lane16, woff_cur + (unsigned)(u * 1024));

    const unsigned rp_bytes = (unsigned)(RP * kPix);
    int sc = 0, stage = 0;
    while (true) {
        // ---- stage top.  Inside an item the requests of this stage were issued a whole stage ago and are older than every
        // weight load still in flight: the counted waits of the previous stage's units already covered them.  At an item's
        // first stage the epilogue's stores may be outstanding (loads and stores retire out of order with respect to each
        // other): wait for everything.
        if (sc == 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // requests of the NEXT stage (the next item's first one behind this item's last) into the other half
        {
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
            int iy0 = 0, ix0 = 0;
            unsigned base = 0;
            if (real) tile_origin(n_item, iy0, ix0, base);
            issue_halo(real, iy0, ix0, base, (unsigned)n_sc * 64u, smem_base + (stage ? 0u : stage_bytes));
        }
        const char* Ab = smem + (stage ? stage_bytes : 0u);
        i32x4_t fb[PF];
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) fb[pf] = *reinterpret_cast<const i32x4_t*>(Ab + pixaddr[pf]);   // unit 0: tap (0, 0), k-step 0
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            {   // the fragment R - 1 units ahead, into the slot unit u - 1 just released
                const int up = u + R - 1;
                const unsigned so = up < kUnits ? woff_cur + (unsigned)(up * 1024) : woff_next + (unsigned)((up - kUnits) * 1024);
                load_frag(wr[up % R], rsW, lane16, so);
            }
            // in flight behind this unit's fragment: the R - 1 younger fragments, and - for the fragments requested before this
            // stage's top - the halo burst
            if (u < R - 1) {
                wait_frag<R - 1 + kMaxP>(wr[u % R]);
            } else {
                wait_frag<R - 1>(wr[u % R]);
            }
            const int un = u + 1;
            const unsigned offn = (unsigned)((un >> 1) / 3) * rp_bytes + (unsigned)((((un >> 1) % 3) * kPix) + (un & 1) * 32);
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                acc[pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, wr[u % R]), __builtin_bit_cast(h8_t, fb[pf]), acc[pf], 0, 0, 0);
                if (u + 1 < kUnits) fb[pf] = *reinterpret_cast<const i32x4_t*>(Ab + pixaddr[pf] + offn);
            }
        }
        // ---- next stage
        stage ^= 1;
        ++sc;
        woff_cur = woff_next;
        if (sc == nsc) {
            // the item is complete: bias (+ post-affine) + activation (+ residual) -> fp16 NHWC
            const ConvKArgs ea = reload_args();
            int opix[PF];
            out_pix(ea, id, opix);
            BiasRegs<1> bz;
            load_bias<1>(ea, cb * WC + wc, 0, lane, bz);
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                const int op1[1] = {opix[pf]};
                conv_epilogue<1, 1>(ea, *reinterpret_cast<const f32x16_t(*)[1][1]>(&acc[pf]), op1, cb * WC + wc, 0, lane, bz);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pf][r] = 0.f;
            }
            const int nid = next_valid(id);
            if (nid >= nids) break;
            id = nid;
            sc = 0;
        }
        woff_next = (sc + 1 < nsc) ? woff_cur + kUnits * 1024 : wbase;
    }
    // the fragments requested past the end and the dummy requests of the last stage land before the wave ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PF, int WC, int WP>
int launch_wreg(const Launch& L, hipStream_t s) {
    auto kern = conv3x3_wreg_kernel<PF, WC, WP>;
    static bool big_lds_enabled = false;
    if (L.lds > 64 * 1024 && !big_lds_enabled) {
        Y6_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        big_lds_enabled = true;
    }
    Y6_REQUIRE(L.lds <= 160 * 1024, "conv_wreg: tile needs %zu bytes of LDS", L.lds);
    static size_t cached_lds = 0;
    static int cached_bpc = 0, n_cu = 0;
    if (cached_lds != L.lds) {
        int bpc = 0;
        Y6_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, WC * WP * 64, L.lds));
        if (n_cu == 0) {
            int dev = 0;
            Y6_HIP(hipGetDevice(&dev));
            Y6_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        cached_bpc = bpc < 1 ? 1 : bpc;
        cached_lds = L.lds;
    }
    int grid = n_cu * cached_bpc;
    const int gq = 8 * L.k.ncb;   // ids of one tile's cout blocks share id % 8 (XCD); (id >> 3) % ncb - the cout block - is the same for every item of a block
    grid -= grid % gq;
    if (grid < gq) grid = gq;
    if (grid > L.grid) grid = L.grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WC * WP * 64), L.lds, s, L.k);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

}  // namespace

int y6_conv_wreg_max_pieces(int nw) { return kMaxP * nw; }

// L points at conv_mfma.hip's launch record (conv_common.hpp)
int y6_conv_wreg_launch(const void* Lp, int pf, int wc, int wpx, hipStream_t s) {
    const Launch& L = *static_cast<const Launch*>(Lp);
    if (wc == 4 && wpx == 1) {
        switch (pf) {
            case 8: return launch_wreg<8, 4, 1>(L, s);
            case 7: return launch_wreg<7, 4, 1>(L, s);
            case 4: return launch_wreg<4, 4, 1>(L, s);
        }
    }
    if (wc == 2 && wpx == 2) {
        switch (pf) {
            case 8: return launch_wreg<8, 2, 2>(L, s);
            case 7: return launch_wreg<7, 2, 2>(L, s);
        }
    }
    y6_set_error("conv_wreg: no instantiation pf %d, %d x %d waves", pf, wc, wpx);
    return Y6_EUNSUPPORTED;
}
