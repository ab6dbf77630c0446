// conv_dma.hip — 3x3 convolution (stride 1, fp16 and int8; a stride-2 form) fed entirely by LDS-DMA.
//
// Replaces the same aten compositions as conv_mfma.hip (reference yolov6/layers/common.py:51-54, :247-248, :338-339,
// :605-608): RepVGG / QARepVGG / ConvBN{ReLU,SiLU} / BottleRep 3x3 convs in deploy form, 84 % of the FLOPs of YOLOv6-S.
//
// Why another kernel.  Round 1's pipelined kernel (conv_mfma_pipe_kernel) stages every byte through registers: global load
// -> VGPR -> ds_write_b128.  Its probes put the fill at 43 % of its time, overlapping the MFMAs by a fifth:
// ds_write_b128 moves 79 B/clk/CU (MI355X_MICROARCH.md, LDS table) against 256 B/clk for the fragment reads, the staging
// registers cap it at 2 waves per SIMD, and every staged load drags VALU address selects with it.  Here
//   * BOTH operands arrive by `buffer_load_dwordx4 ... lds` (1 KiB per wave instruction, no VGPR, no ds_write); image
//     borders and tile overhang are zero-filled by the buffer descriptor's range check (voffset past num_records);
//   * halo image, 16-channel chunks: two planes (channels 0-7 / 8-15; int8: 0-15 / 16-31), 16 B per pixel per plane,
//     row-major with the halo width as pitch - lane-linear, so a DMA piece is 64 consecutive slots and the SOURCE address
//     does the gather; 32-channel chunks: pixel-major with a source-side XOR swizzle (template parameter HC below);
//   * fragment pixel q of a 32-pixel MFMA fragment sits in the lane (frag_pixel) that puts the two 16-lane groups
//     ds_read_b128 serves per cycle ({0-3,12-15,20-27} / {4-11,16-19,28-31}) on 16 CONSECUTIVE pixels each:
//     conflict-free for every tile width that is a multiple of 16, whatever the row pitch;
//   * one barrier per chunk: `s_waitcnt vmcnt(0)` (the wave's own pieces of THIS chunk, requested a whole chunk of MFMAs
//     ago) + s_barrier; the pieces of the next chunk are requested one per MFMA unit into the other stage while the nine
//     taps run.  The request stream crosses work items: the first chunk of the next (tile, cout block) lands early;
//   * the epilogue of item k is deferred into item k+1's first chunk (second accumulator set; see `fast_unit`).
// Production forms (autotuned per layer): 4 waves x (64 couts x 64 / 32 pixels per wave) = 256- / 128-pixel blocks, two /
// three blocks per CU, 16-channel chunks; the 8-wave 512-pixel block with 32-channel chunks is the fastest in steady state
// but pays a stage-start penalty on most boxes of this pool and is opt-in for fp16; int8 uses it.
// The int8 form (v_mfma_i32_32x32x32_i8, BASELINE configs[4]) is the same data movement byte for byte: a chunk is 32 (64)
// int8 channels = 32 (64) B per pixel; it reads the int8 twin its producer wrote (include/yolov6_hip.h y6_conv_i8_desc.q_in).
#include <type_traits>

#include "common.hpp"
#include "conv_common.hpp"

namespace {

// Ceiling probes / trace (tools/build_probe_libs.py --dma; WRONG RESULTS for n >= 2, timing only), compile-time:
//   1 = s_memtime trace of block 0 / thread 0 into a.dbg (tools/dma_trace.py); 2 = no epilogue; 3 = no MFMAs;
//   4 = no LDS-DMA requests after the prologue; 5 = fragment reads of tap 0 only (MFMAs on stale operands);
//   6 = no chunk barrier / vmcnt wait; 7 = as 2, and no weight requests (halo requests + barriers only); 8 = as 2, and no halo
//   requests (weight requests + barriers only).  (With the epilogue gone the compiler drops the MFMAs and fragment reads as
//   dead code: 2 / 7 / 8 time the requests and the chunk barriers alone.)
#ifndef Y6_DMA_PROBE
#define Y6_DMA_PROBE 0
#endif
constexpr int kDmaProbe = Y6_DMA_PROBE;
constexpr bool kProbeNoEpilogue = kDmaProbe == 2 || kDmaProbe == 7 || kDmaProbe == 8;

// Halo image of the fp16 stride-1 16-channel-chunk kernels: 0 (default since round 3) = pixel-major, [halo pixel][2 pieces of 16 B] with
// the piece index XOR bit 3 of the pixel index on the SOURCE side - two lanes of a request read one pixel, so a request touches
// 32 cache lines instead of 64, and the halo requests are what the fill costs; 1 = two planes (channels 0-7 /
// 8-15), one cache line per lane of a request, fragment addresses a constant offset per tap.  History: round 2 measured the
// pixel-major image 4-8 % SLOWER (profiles/r02/ab_layout_ablayout1.txt) - its fragment addresses were recomputed per read, four
// VALU instructions each; with the 9 x PF addresses of a lane kept in registers (Y6_DMA_PIX16_HOIST: they are kernel invariants;
// one v_add per fragment read, as the planar image needs) it is 1.3-1.4 % FASTER over the whole step, same box, alternating runs
// (profiles/r03/bench_infer_r03r_{planar,pixmajor}{1,2}.json; the tuner then prefers dma_c1p2 where it took dma_c2p1).
// The stride-2 form and the int8 kernels stay planar (the even/odd column split of stride 2 is a property of the planar rows;
// the int8 8-wave form has no registers to spare).  (The 32-channel-chunk kernels are pixel-major by construction.)
#ifndef Y6_DMA_PLANAR16
#define Y6_DMA_PLANAR16 0
#endif
#ifndef Y6_DMA_PIX16_HOIST
#define Y6_DMA_PIX16_HOIST 1
#endif

template <int BP, int ST = 1>
struct DmaHaloCap {   // halo pixels per plane for BP output pixels (stride 2: (2 TH + 1) x (2 TW + 1) input pixels)
    static constexpr int value = ST == 2 ? (BP <= 128 ? 576 : 1152) : (BP <= 128 ? 208 : (BP <= 256 ? 352 : (BP <= 512 ? 672 : 1216)));
};

template <bool I8>
struct AccT {
    typedef f32x16_t type;
};
template <>
struct AccT<true> {
    typedef i32x16_t type;
};

// uniform switch over the immediates s_waitcnt takes
__device__ __forceinline__ void wait_vm_barrier(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
}

// STG: LDS stages (2: the pieces of chunk c+1 are requested while chunk c is multiplied; 3: of chunk c+2, so a piece has
//      two chunk periods to land and the wait in front of the barrier is a counted vmcnt that leaves the younger chunk's
//      pieces in flight).  IL: 1 = the requests are issued one or two per tap BETWEEN the MFMAs of the running chunk (an
//      LDS-DMA instruction occupies the issuing wave for 60-180 cycles - MI355X_MICROARCH.md; behind four MFMAs that time
//      is covered by the matrix pipe), 0 = all of them right after the barrier (A/B).
// HC: input channels per chunk (int8: twice that).  16: a pixel's chunk is 32 B in two planes (see the header).  32: 64 B per
//     pixel, i.e. FOUR lanes of a DMA instruction read one pixel and an instruction touches 16 cache lines instead of 64 -
//     the halo pieces were the expensive ones (about 100 cycles each against about 16 for a weight piece: the vector memory
//     path serves one cache line per cycle or so; the fill alone ran at 18.7 B/clk/CU = 9.6 TB/s).  The LDS image is
//     pixel-major there, [halo pixel][4 pieces of 16 B], with the piece index XOR-swizzled by bits 2-3 of the pixel index
//     on the SOURCE side (cdna_hip_programming.md rule 21): a fragment read of 16 consecutive pixels still covers all 16
//     bank groups, for any alignment.
// ST: conv stride.  2 (planar 16-channel chunks only): the halo rows are stored with their even columns first, then the odd
//     ones (slot = hy * RP + (hx & 1 ? RPE : 0) + (hx >> 1), RPE = number of even columns) - the source address does the
//     de-interleave - so that the 16 consecutive output pixels of a read group (input columns 2 tx + kx) read 16 CONSECUTIVE
//     slots for every tap, as with stride 1: kx = 0 -> even slot tx, kx = 1 -> odd slot tx, kx = 2 -> even slot tx + 1.
// WRES: the block's weights stay in LDS for all its work items (layers of at most 64 input channels: 9 x Cin x CF*32 fp16 =
//     72 KB at CF 2) - a stage holds the halo only, and the host sizes the grid so that every item of a block has the same
//     cout block.  Without it 62-75 % of the bytes a block requests are tap images re-fetched from L2 for every tile
//     (tools/dma_model.py).
template <int CF, int PF, int NW, int WPS, int STG, int IL, int HC, bool I8, int ST = 1, bool WRES = false>
__global__ __launch_bounds__(NW * 64, WPS) void conv3x3_dma_kernel(const ConvKArgs a) {
    static_assert(ST == 1 || HC == 16, "stride 2 is built on the planar 16-channel-chunk image");
    static_assert(!WRES || (!I8 && ST == 1 && STG == 2), "resident weights: fp16, stride 1, two halo stages");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename AccT<I8>::type acc_t;
    constexpr int NT = 9;
    constexpr int KS = HC / 16;                                   // MFMA k-steps per chunk and tap
    constexpr int SPP = HC / 8;                                   // 16-byte pieces per halo pixel and chunk
    constexpr int JB = HC == 16 ? 1 : 2;                          // bits of the piece index
    constexpr bool PLANAR = HC == 16 && (Y6_DMA_PLANAR16 != 0 || ST == 2 || I8);    // else pixel-major [halo pixel][SPP pieces], piece index XOR bits (4-JB).. of the pixel index
    constexpr int WP = CF * NT * KS;                              // weight pieces (1 KiB) per chunk
    constexpr int MAXNHP = (DmaHaloCap<NW * PF * 32, ST>::value * SPP + 63) / 64;
    constexpr int ES = I8 ? 1 : 2;                                // bytes per input element

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RP = a.dma_rp, PLs = a.dma_pls, NHP = a.dma_nhp;
    const int stage_bytes = (NHP + (WRES ? 0 : WP)) * 1024;
    const unsigned smem_base = lds_addr(smem);
    float* ldsBias = reinterpret_cast<float*>(smem + STG * stage_bytes);   // [item parity][bias | post scale | post shift | dequant scale][CF*32]
    const int nch = a.Cin / (I8 ? 2 * HC : HC);
    if constexpr (WRES) ldsBias += nch * WP * 256;   // resident tap images [chunk][cf][tap][k-step] sit between the halo stages and these vectors
    const int nids = a.nids;
    const int gstride = gridDim.x;

    const char* inb = I8 ? reinterpret_cast<const char*>(a.qin) : reinterpret_cast<const char*>(a.in);
    const int ics = I8 ? a.qin_cs : a.in_cs, ico = I8 ? a.qin_co : a.in_co;
    const i32x4_t rsA = make_rsrc(inb, (unsigned)((size_t)a.B * a.H * a.W * ics * ES));
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
    {
        int t, c;
        decode(id, t, c);
        if (t >= a.ntiles) id = next_valid(id);
    }
    if (id >= nids) return;
    int dbg_n = 0;
    const bool tracing = kDmaProbe == 1 && a.dbg != nullptr && blockIdx.x == 0 && tid == 0;
#define DT(tag)                                                              \
    do {                                                                     \
        if (kDmaProbe == 1 && tracing && dbg_n < 256) {                      \
            a.dbg[2 * dbg_n] = __builtin_amdgcn_s_memtime();                 \
            a.dbg[2 * dbg_n + 1] = (unsigned long long)(tag);                \
            ++dbg_n;                                                         \
        }                                                                    \
    } while (0)
    DT(1);

    // ---- this wave's pieces of a chunk image  [halo (planes or pixel-major) | pad to 1 KiB][CF x 9 x KS weight fragments]:
    //      halo pieces P = wave + NW*i < NHP, weight fragments q = wave + NW*j < WP.
    // Halo: slot s = 64*P + lane -> (plane h, halo row hy, halo column hx) is fixed for the whole kernel (hinfo, computed
    // once with the divisions); per work item only the tile origin changes: a scalar base offset plus two range checks.
    constexpr int NPWH = (MAXNHP + NW - 1) / NW;
    constexpr int NPWW = (WP + NW - 1) / NW;
    unsigned hinfo[NPWH], hvoff[NPWH];
#pragma unroll
    for (int i = 0; i < NPWH; ++i) {
        const int P = wave + NW * i;
        const int s = P * 64 + lane;
        int r, j;   // halo pixel (linear) and piece of LDS slot s
        if (PLANAR) {
            j = s >= PLs ? 1 : 0;
            r = s - j * PLs;
        } else {
            r = s >> JB;
            j = (s & (SPP - 1)) ^ ((r >> (4 - JB)) & (SPP - 1));
        }
        const int hy = r / RP;
        int hx = r - hy * RP;
        if (ST == 2) {   // slot column -> image column: evens first
            const int rpe = (a.HWd + 1) >> 1;
            hx = hx < rpe ? 2 * hx : 2 * (hx - rpe) + 1;
        }
        const bool v = (P < NHP) && (s < SPP * PLs) && (hx < a.HWd);
        hinfo[i] = v ? (unsigned)((hy << 16) | (hx << JB) | j) : 0xffffffffu;
        hvoff[i] = kOob;
    }
    unsigned wsoff[NPWW];   // scalar part of the weight fragment addresses (cf, tap of this wave's j-th fragment)
#pragma unroll
    for (int j = 0; j < NPWW; ++j) {
        const int q = wave + NW * j;
        const int ct = q / KS, ks = q - ct * KS;   // LDS image [cf][tap][k-step]
        const int cf = ct / NT, tap = ct - cf * NT;
        wsoff[j] = (unsigned)(((cf * a.nchunk * NT + tap) * 2 + ks) * 1024);
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    auto setup_halo = [&](int item) {
        int tile, cbx;
        decode(item, tile, cbx);
        const int tx_i = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty_i = t2 % a.tiles_y;
        const int b = t2 / a.tiles_y;
        const int iy0 = ty_i * a.TH * ST - 1, ix0 = tx_i * a.TW * ST - 1;
        // byte offsets are computed modulo 2^32 (tensors up to 3.5 GiB): unsigned arithmetic, the tile origin may lie one row /
        // column outside the image and a valid piece still ends up at its true offset
        const unsigned base = (((unsigned)(b * a.H + iy0) * (unsigned)a.W + (unsigned)ix0) * (unsigned)ics + (unsigned)ico) * (unsigned)ES;
#pragma unroll
        for (int i = 0; i < NPWH; ++i) {
            const int hy = (int)(hinfo[i] >> 16), hx = (int)((hinfo[i] & 0xffffu) >> JB), h = (int)(hinfo[i] & ((1u << JB) - 1u));
            const bool v = (hinfo[i] != 0xffffffffu) && ((unsigned)(iy0 + hy) < (unsigned)a.H) && ((unsigned)(ix0 + hx) < (unsigned)a.W);
            hvoff[i] = v ? base + (unsigned)(hy * a.W + hx) * (unsigned)(ics * ES) + (unsigned)(h * 16) : kOob;
        }
    };
    // the request cursor: the chunk whose pieces are requested next, STG-1 chunks ahead of the one being multiplied
    int c_item = id, c_chunk = 0, c_cb = 0;
    bool c_valid = true;
    unsigned c_dst0 = 0, c_soffA = 0, c_soffW = 0;
    auto cursor_target = [&](int stage) {
        c_dst0 = smem_base + stage * stage_bytes;
        c_soffA = (unsigned)c_chunk * (unsigned)(SPP * 16);
        c_soffW = KS == 1 ? (unsigned)(((c_cb * CF * a.nchunk + (c_chunk >> 1)) * NT * 2 + (c_chunk & 1)) * 1024)
                          : (unsigned)(((c_cb * CF * a.nchunk + c_chunk) * NT * 2) * 1024);
    };
    constexpr int NPIECE = NPWH + (WRES ? 0 : NPWW);
    bool in_loop = false;
    auto issue_piece = [&](int k) {   // k: compile-time index, halo pieces first (longest latency)
        if (k < NPWH) {
            const int P = wave + NW * k;
            if (P < NHP && (kDmaProbe != 4 || !in_loop) && kDmaProbe != 8) dma16(rsA, hvoff[k < NPWH ? k : 0], c_soffA, c_dst0 + P * 1024);
        } else if (!WRES && k < NPIECE) {
            const int j = k - NPWH;
            const int q = wave + NW * j;
            if (q < WP && (kDmaProbe != 4 || !in_loop) && kDmaProbe != 7) dma16(rsW, lane16, c_soffW + wsoff[j < NPWW ? j : 0], c_dst0 + (NHP + q) * 1024);
        }
    };
    auto cursor_advance = [&]() {
        if (c_chunk + 1 < nch) {
            ++c_chunk;
            return;
        }
        const int n = next_valid(c_item);
        if (n >= nids) {
            c_valid = false;
            return;
        }
        c_item = n;
        c_chunk = 0;
        int t;
        decode(n, t, c_cb);
        setup_halo(n);   // the previous item's offsets are dead: all its chunks have been requested
    };
    // pieces this wave requests per chunk (the counted wait of the 3-stage form)
    int npw = 0;
#pragma unroll
    for (int k = 0; k < NPWH; ++k) npw += (wave + NW * k < NHP) ? 1 : 0;
#pragma unroll
    for (int j = 0; j < NPWW; ++j) npw += (!WRES && wave + NW * j < WP) ? 1 : 0;

    // ---- this lane's pixels: position in the tile (fixed for the whole kernel: computed once, with the divisions), LDS
    //      offsets of the fragment reads; per work item only the tile origin changes (scalar) - output offsets are one
    //      multiply-add and two range checks per pixel
    const int fq = frag_pixel(lane & 31);
    int pixoff[PF], ptytx[PF], cb = 0;   // ptytx: ty << 16 | tx, -1 for a lane beyond the tile
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int m = wave * (PF * 32) + pf * 32 + fq;
        const int npx = a.TH * a.TW;
        const int mm = m < npx ? m : npx - 1;
        const int ty = mm / a.TW, tx = mm - ty * a.TW;
        pixoff[pf] = PLANAR ? ((lane >> 5) * PLs + ty * ST * RP + tx) * 16 : (ty * RP + tx);   // pixel-major: the pixel's linear index
        ptytx[pf] = m < npx ? ((ty << 16) | tx) : -1;
    }
    constexpr bool HOIST = !PLANAR && HC == 16 && Y6_DMA_PIX16_HOIST != 0;
    int tapaddr[HOIST ? NT : 1][PF];
    if constexpr (HOIST) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                const int p = pixoff[pf] + (t / 3) * RP + (t % 3);
                tapaddr[t][pf] = p * (SPP * 16) + ((((lane >> 5) ^ (p >> (4 - JB))) & (SPP - 1)) << 4);
            }
    }
    auto setup_pix = [&](int item) {
        int tile;
        decode(item, tile, cb);
    };
    auto out_pix = [&](const ConvKArgs& a, int item, int (&opix)[PF]) {
        int tile = item;
        if (a.ncb != 1) tile = ((item >> 3) / a.ncb) * 8 + (item & 7);
        const int tx_i = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty_i = t2 % a.tiles_y;
        const int b = t2 / a.tiles_y;
        const int oy0 = ty_i * a.TH, ox0 = tx_i * a.TW;
        const int base = (b * a.Ho + oy0) * a.Wo + ox0;
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) {
            const int ty = ptytx[pf] >> 16, tx = ptytx[pf] & 0xffff;
            const bool v = ptytx[pf] >= 0 && (oy0 + ty < a.Ho) && (ox0 + tx < a.Wo);
            opix[pf] = v ? base + ty * a.Wo + tx : -1;
        }
    };

    setup_halo(id);
    setup_pix(id);
    c_cb = cb;
    if constexpr (WRES) {   // every tap image of this block's cout block, once; they land before the first chunk barrier
        constexpr int MAXCH = 64 / HC;
        constexpr int NPWR = (MAXCH * WP + NW - 1) / NW;
        const unsigned dstR = smem_base + STG * stage_bytes;
#pragma unroll
        for (int j = 0; j < NPWR; ++j) {
            const int q = wave + NW * j;
            if (q < nch * WP) {
                const int c = q / WP, f = q - c * WP;
                const int ct = f / KS, ks = f - ct * KS;
                const int cfi = ct / NT, tap = ct - cfi * NT;
                const unsigned chunk_off = KS == 1 ? (unsigned)(((cb * CF * a.nchunk + (c >> 1)) * NT * 2 + (c & 1)) * 1024)
                                                   : (unsigned)(((cb * CF * a.nchunk + c) * NT * 2) * 1024);
                dma16(rsW, lane16, chunk_off + (unsigned)(((cfi * a.nchunk * NT + tap) * 2 + ks) * 1024), dstR + q * 1024);
            }
        }
    }
#pragma unroll
    for (int st = 0; st < STG - 1; ++st) {
        if (c_valid) {
            cursor_target(st);
#pragma unroll
            for (int k = 0; k < NPIECE; ++k) issue_piece(k);
            cursor_advance();
        }
    }

    in_loop = true;
    DT(2);
    int pb = 0, item_parity = 0;
    bool prev_issued = c_valid || STG == 2;   // did the previous iteration put pieces of a YOUNGER chunk in flight

    // ---- epilogues.
    // General (any activation, QARepVGG post-affine, residual, ragged channel counts, int8): after the item's last chunk, one
    // (cout fragment, pixel fragment) unit after the other, on arguments re-read from the kernarg segment.
    // Fast + deferred (conv + bias (+ int8 dequantisation) (+ QARepVGG post-affine) + activation, no residual, into a 16-byte
    // aligned fp16 view and / or a 4-byte aligned int8 twin, whole cout blocks - every 3x3 of the deploy graphs but the
    // BottleRep shortcut convs): the finished accumulators of item k stay in their registers while item
    // k+1 accumulates into a second set, and the units of item k are issued one per tap BEHIND the MFMAs of item k+1's first
    // chunk - the matrix pipe keeps running while the wave does bias / max / pack / store (before: the epilogue took 20 % of
    // an item's time at Cin 128 and 33 % at Cin 64 with the matrix pipe waiting, tools/dma_trace.py).  Stores go through a
    // buffer descriptor: 32-bit byte offsets, overhang pixels dropped by the range check.  Needs 2 x CF*PF*16 accumulator
    // registers: variants with at most three waves per SIMD.
    constexpr bool DEFER = WPS <= 3 && CF * PF <= 4;
    constexpr int NUNIT = CF * PF;
    static_assert(!DEFER || NUNIT <= NT - 1, "one deferred epilogue unit per tap 1..8");
    auto epi_unit = [&](const ConvKArgs& ea, const acc_t (&accP)[CF][PF], const int (&opx)[PF], int cbq, const float* lb, int u) {
        const int cf = u / PF, pf = u - cf * PF;
        BiasRegs<1> bz;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(lb + cf * 32 + 8 * g + 4 * (lane >> 5));
            bz.v[0][g * 4 + 0] = t.x;
            bz.v[0][g * 4 + 1] = t.y;
            bz.v[0][g * 4 + 2] = t.z;
            bz.v[0][g * 4 + 3] = t.w;
        }
        if (kProbeNoEpilogue) return;
        const int op1[1] = {opx[pf]};
        if constexpr (I8) {
            BiasRegs<1> qs;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(lb + 3 * CF * 32 + cf * 32 + 8 * g + 4 * (lane >> 5));
                qs.v[0][g * 4 + 0] = t.x;
                qs.v[0][g * 4 + 1] = t.y;
                qs.v[0][g * 4 + 2] = t.z;
                qs.v[0][g * 4 + 3] = t.w;
            }
            conv_i8_epilogue<1>(ea, *reinterpret_cast<const i32x16_t(*)[1]>(&accP[cf][pf]), op1, cbq * CF + cf, lane, bz.v[0], qs.v[0]);
        } else {
            conv_epilogue<1, 1>(ea, *reinterpret_cast<const f32x16_t(*)[1][1]>(&accP[cf][pf]), op1, cbq * CF + cf, 0, lane, bz);
        }
    };
    // fast path: bias (+ int8 dequantisation) (+ QARepVGG post-affine) + activation, no residual, whole cout blocks,
    // 16-byte aligned fp16 view and / or 4-byte aligned int8 twin
    const bool has_out = a.out != nullptr, has_post = a.pscale != nullptr;
    const bool has_qout = I8 && a.qout != nullptr;
    const bool fast = DEFER && a.res == nullptr && (a.Cout % (CF * 32)) == 0 && a.up == 0 &&
                      (has_out || has_qout) && (!has_out || a.vec16_ok) && (!has_qout || ((a.qout_cs | a.qout_co) & 3) == 0) &&
                      (!I8 || a.acc_out == nullptr) && (size_t)a.B * a.Ho * a.Wo * a.out_cs * 2 < 0xe0000000ull;
    const float fast_lo = a.act == Y6_ACT_RELU ? 0.f : -__builtin_inff();
    const bool smooth_act = a.act == Y6_ACT_SILU || a.act == Y6_ACT_HARDSWISH;
    const __amdgpu_buffer_rsrc_t rsO =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(unsigned)((size_t)a.B * a.Ho * a.Wo * a.out_cs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsQ =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.qout, 0, (int)(unsigned)((size_t)a.B * a.Ho * a.Wo * (I8 ? a.qout_cs : 0)), 0x00020000);
    unsigned obyteP[PF];   // byte offset of the lane's output pixels of the PREVIOUS item in the fp16 view (kOob: none)
    unsigned qbyteP[PF];   // ... in the int8 twin
    unsigned ocolP = 0;    // + channel offset of its cout block and of this lane's piece (in channels)
    const float* lbP = ldsBias;
    bool haveP = false;
    // (round 6) the accumulating data-gradient convs of the training step: the same bias-only arithmetic ADDED to what the output view
    // holds (res == out, no scaling, no activation) - at the end of the item, not deferred (a load behind the next item's requests
    // could only be awaited by draining them), with the pieces of all units requested before the first one is finished.  The
    // general epilogue paid a residual round trip per unit: 270 us against 159 forward for 64 -> 64 @160 b64.
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const bool accum = !I8 && a.res != nullptr && a.res == a.out && a.res_cs == a.out_cs && a.res_co == a.out_co && a.res_alpha == nullptr &&
                       a.act == Y6_ACT_NONE && !has_post && (a.Cout % (CF * 32)) == 0 && a.up == 0 && has_out && a.vec16_ok && a.accum_fast &&
                       (size_t)a.B * a.Ho * a.Wo * a.out_cs * 2 < 0xe0000000ull;
    auto fast_unit = [&](const acc_t (&accP)[CF][PF], int u, const u32x4_t* prev = nullptr) {
        const int cf = u / PF, pf = u - cf * PF;
        const int kh = lane >> 5;
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bz = *reinterpret_cast<const float4*>(lbP + cf * 32 + 8 * g + 4 * kh);
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = (float)accP[cf][pf][g * 4 + j];
            if constexpr (I8) {   // exact int32 -> fp32, * s_x*s_w[c] as a rounding of its own (include/yolov6_hip.h)
                const float4 qs = *reinterpret_cast<const float4*>(lbP + 3 * CF * 32 + cf * 32 + 8 * g + 4 * kh);
                x[0] *= qs.x;
                x[1] *= qs.y;
                x[2] *= qs.z;
                x[3] *= qs.w;
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(x[j]));
            }
            x[0] += bz.x;
            x[1] += bz.y;
            x[2] += bz.z;
            x[3] += bz.w;
            if (has_post) {   // QARepVGG: conv -> BatchNorm are two fp16 ops (finish16 in conv_common.hpp)
                const float4 ps = *reinterpret_cast<const float4*>(lbP + CF * 32 + cf * 32 + 8 * g + 4 * kh);
                const float4 pt = *reinterpret_cast<const float4*>(lbP + 2 * CF * 32 + cf * 32 + 8 * g + 4 * kh);
                x[0] = y6_round_f16(x[0]) * ps.x + pt.x;
                x[1] = y6_round_f16(x[1]) * ps.y + pt.y;
                x[2] = y6_round_f16(x[2]) * ps.z + pt.z;
                x[3] = y6_round_f16(x[3]) * ps.w + pt.w;
            }
            if (smooth_act) {   // SiLU / hardswish: one wave-uniform branch per group, the arithmetic of act_const<>
#pragma unroll
                for (int j = 0; j < 4; ++j) v[g * 4 + j] = a.act == Y6_ACT_SILU ? act_const<Y6_ACT_SILU>(x[j]) : act_const<Y6_ACT_HARDSWISH>(x[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[g * 4 + j] = fmaxf(x[j], fast_lo);
            }
        }
        if (kProbeNoEpilogue) return;
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
        if constexpr (I8) {
            if (has_qout) {   // the int8 twin for quantised consumers: the SAME fp16 values, quantised with their scale
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned q = q8_quad(pk[g][0], pk[g][1], a.qo_inv2, a.qo_lo2, a.qo_hi2);
                    __builtin_amdgcn_raw_buffer_store_b32(q, rsQ, (int)(qbyteP[pf] + ocolP + (unsigned)(cf * 32 + 8 * g - 4 * kh)), 0, 0);
                }
            }
        }
        if (has_out) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
                u32x4_t o = {s0[0], s1[0], s0[1], s1[1]};
                if (prev != nullptr) {   // fp16(x) + what the view held: conv_common.hpp finish16 with res == out, alpha 1 (the same two roundings)
                    const h8_t xo = __builtin_bit_cast(h8_t, o), xr = __builtin_bit_cast(h8_t, prev[gp]);
                    h8_t y;
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = (_Float16)((float)xo[e] + (float)xr[e]);
                    o = __builtin_bit_cast(u32x4_t, y);
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, rsO, (int)(obyteP[pf] + (ocolP + (unsigned)(cf * 32 + 16 * gp)) * 2), 0, 0);
            }
        }
    };

    // one work item: accumulate into `acc`; (fast path) finish the previous item out of `accP` on the way.  Returns true
    // after the block's last item.  FAST is a compile-time tag so that the two forms are separate loops.
    auto do_item = [&](auto fast_tag, acc_t (&acc)[CF][PF], acc_t (&accP)[CF][PF]) __attribute__((always_inline)) -> bool {
        constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int pf = 0; pf < PF; ++pf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cf][pf][r] = 0;
        float* lbias = ldsBias + (item_parity ? 4 * CF * 32 : 0);
        if (tid < CF * 32) {
            const int c = cb * CF * 32 + tid;
            lbias[tid] = (a.bias != nullptr && c < a.Cout) ? a.bias[c] : 0.f;
            if (a.pscale != nullptr) {
                lbias[CF * 32 + tid] = c < a.Cout ? a.pscale[c] : 0.f;
                lbias[2 * CF * 32 + tid] = c < a.Cout ? a.pshift[c] : 0.f;
            }
            if (I8) lbias[3 * CF * 32 + tid] = (c < a.Cout) ? a.qscale[c] : 0.f;
        }
        const int nid = next_valid(id);
        for (int chunk = 0; chunk < nch; ++chunk) {
            // this wave's pieces of `chunk` have landed; after the barrier everybody's have, and nobody reads the other
            // stage any more (its last fragment reads fed MFMAs that were issued before the barrier)
            // (3 stages, inside an item: the younger chunk's pieces stay in flight; while epilogue stores may be outstanding -
            // an item's first chunk, the second one too with the deferred epilogue - wait for everything: loads and stores
            // retire out of order with respect to each other)
            DT(9);
            if (kDmaProbe == 6) {
            } else if (STG == 3 && chunk > (FAST ? 1 : 0) && prev_issued) {
                wait_vm_barrier(npw);
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            DT(10);
            const bool issuing = c_valid;
            if (issuing) {
                cursor_target(pb == 0 ? STG - 1 : pb - 1);   // the stage read in the previous iteration
                if (!IL) {
#pragma unroll
                    for (int k = 0; k < NPIECE; ++k) issue_piece(k);
                }
            }
            const bool epi_now = FAST && haveP && chunk == 0;
            const char* Ab = smem + pb * stage_bytes;
            const char* Wb = WRES ? smem + STG * stage_bytes + chunk * (WP * 1024) + lane * 16 : Ab + NHP * 1024 + lane * 16;
            constexpr int NU = NT * KS;                               // MFMA units (tap, k-step) per chunk
            constexpr int PPU0 = (NPIECE + NU - 2) / (NU - 1);        // DMA pieces per unit: all requested before the last one,
            constexpr int PPU = IL > PPU0 ? IL : PPU0;                // or IL per unit (front-loaded)
            i32x4_t fa[2][CF], fb[2][PF];                             // fragment reads run one unit ahead of the MFMAs
            int plc[PF];
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                plc[pf] = pixoff[pf];
                // opaque per chunk: otherwise hipcc hoists the 9 x PF swizzled tap addresses out of the chunk loop and spills
                if (!PLANAR && !HOIST) asm volatile("" : "+v"(plc[pf]));
            }
            auto ldfrag = [&](int u, int buf) {
                const int t = u / KS, ks = u - t * KS;
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) fa[buf][cf] = *reinterpret_cast<const i32x4_t*>(Wb + ((cf * NT + t) * KS + ks) * 1024);
                if constexpr (PLANAR) {
                    const int tapoff = ((t / 3) * RP + (ST == 1 ? (t % 3) : ((t % 3) == 1 ? ((a.HWd + 1) >> 1) : ((t % 3) >> 1)))) * 16;
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) fb[buf][pf] = *reinterpret_cast<const i32x4_t*>(Ab + pixoff[pf] + tapoff);
                } else if constexpr (HOIST) {
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) fb[buf][pf] = *reinterpret_cast<const i32x4_t*>(Ab + tapaddr[t][pf]);
                } else {
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) {
                        const int p = plc[pf] + (t / 3) * RP + (t % 3);
                        const int sw = (p >> (4 - JB)) & (SPP - 1);
                        fb[buf][pf] = *reinterpret_cast<const i32x4_t*>(Ab + p * (SPP * 16) + ((((ks << 1) | (lane >> 5)) ^ sw) << 4));
                    }
                }
            };
            ldfrag(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 1 < NU && kDmaProbe != 5) ldfrag(u + 1, (u + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads of unit u+1 AHEAD of unit u's MFMAs
#pragma unroll
                for (int cf = 0; cf < CF; ++cf)
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) {
                        if (kDmaProbe == 3) continue;
                        if constexpr (I8) {
                            acc[cf][pf] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u & 1][cf], fb[u & 1][pf], acc[cf][pf], 0, 0, 0);
                        } else {
                            acc[cf][pf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, fa[u & 1][cf]),
                                                                                 __builtin_bit_cast(h8_t, fb[u & 1][pf]), acc[cf][pf], 0, 0, 0);
                        }
                    }
                if (IL && issuing) {
#pragma unroll
                    for (int v = 0; v < PPU; ++v) issue_piece(u * PPU + v);
                }
                if constexpr (FAST) {
                    if (u % KS == KS - 1 && u / KS >= 1 && u / KS - 1 < NUNIT) {
                        if (epi_now) fast_unit(accP, u / KS - 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            DT(11);
            prev_issued = issuing;
            if (issuing) cursor_advance();
            pb = (pb + 1 == STG) ? 0 : pb + 1;
        }
        DT(19);
        if constexpr (FAST) {
            int opix[PF];
            out_pix(a, id, opix);
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                obyteP[pf] = opix[pf] >= 0 ? ((unsigned)opix[pf] * (unsigned)a.out_cs + (unsigned)a.out_co) * 2u : kOob;
                qbyteP[pf] = (I8 && opix[pf] >= 0) ? (unsigned)opix[pf] * (unsigned)a.qout_cs + (unsigned)a.qout_co : kOob;
            }
            ocolP = (unsigned)(cb * CF * 32 + 8 * (lane >> 5));
            lbP = lbias;
            haveP = true;
        } else if (accum) {
            int opix[PF];
            out_pix(a, id, opix);
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                obyteP[pf] = opix[pf] >= 0 ? ((unsigned)opix[pf] * (unsigned)a.out_cs + (unsigned)a.out_co) * 2u : kOob;
                qbyteP[pf] = kOob;
            }
            ocolP = (unsigned)(cb * CF * 32 + 8 * (lane >> 5));
            lbP = lbias;
            u32x4_t prev[NUNIT][2];
#pragma unroll
            for (int u = 0; u < NUNIT; ++u)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp)
                    prev[u][gp] = __builtin_amdgcn_raw_buffer_load_b128(rsO, (int)(obyteP[u % PF] + (ocolP + (unsigned)((u / PF) * 32 + 16 * gp)) * 2), 0, 0);
            DT(21);
#pragma unroll
            for (int u = 0; u < NUNIT; ++u) fast_unit(acc, u, prev[u]);
        } else {
            const ConvKArgs ea = reload_args();
            int opix[PF];
            out_pix(ea, id, opix);
            DT(21);
#pragma unroll
            for (int u = 0; u < NUNIT; ++u) epi_unit(ea, acc, opix, cb, lbias, u);
        }
        DT(20);
        if (nid >= nids) return true;
        id = nid;
        item_parity ^= 1;
        setup_pix(id);
        return false;
    };

    acc_t acc0[CF][PF];
    if (DEFER && fast) {
        if constexpr (DEFER) {
            acc_t acc1[CF][PF];
            typedef std::integral_constant<bool, true> yes_t;
            while (true) {
                if (do_item(yes_t{}, acc0, acc1)) {
#pragma unroll
                    for (int u = 0; u < NUNIT; ++u) fast_unit(acc0, u);
                    DT(23);
                    break;
                }
                if (do_item(yes_t{}, acc1, acc0)) {
#pragma unroll
                    for (int u = 0; u < NUNIT; ++u) fast_unit(acc1, u);
                    DT(23);
                    break;
                }
            }
        }
    } else {
        typedef std::integral_constant<bool, false> no_t;
        while (!do_item(no_t{}, acc0, acc0)) {
        }
    }
#undef DT
}

// Probe / self-test of the DMA addressing (tests/test_gpu_ops.py): one wave copies 1 KiB from `src` into LDS at byte
// offset `lds_off` (a block may own up to 160 KiB: the destination base in M0 must take offsets above 64 KiB), lanes with
// (oob_mask >> lane) & 1 ask for an out-of-range piece and must see zeros.
__global__ void dma_probe_kernel(const char* src, unsigned bytes, unsigned lds_off, unsigned long long oob_mask, char* dst) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    for (unsigned i = lane; i < 1024 / 4; i += 64) reinterpret_cast<unsigned*>(smem + lds_off)[i] = 0xdeadbeefu;
    __syncthreads();
    const i32x4_t rs = make_rsrc(src, bytes);
    const unsigned v = ((oob_mask >> lane) & 1ull) ? kOob : (unsigned)(lane * 16);
    dma16(rs, v, 0u, lds_addr(smem) + lds_off);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    *reinterpret_cast<uint4*>(dst + lane * 16) = *reinterpret_cast<const uint4*>(smem + lds_off + lane * 16);
}

template <int CF, int PF, int NW, int WPS, int STG, int IL, int HC, bool I8, int ST = 1, bool WRES = false>
int launch_dma(const Launch& L, hipStream_t s) {
    auto kern = conv3x3_dma_kernel<CF, PF, NW, WPS, STG, IL, HC, I8, ST, WRES>;
    Y6_REQUIRE(L.lds <= 160 * 1024, "conv_dma: tile needs %zu bytes of LDS", L.lds);
    static OccupancyCache occ;
    int grid = 0;
    {
        int rc = resident_grid(occ, kern, NW * 64, L.lds, 160 * 1024, &grid);
        if (rc) return rc;
    }
    // ids of one tile's cout blocks share id % 8 (XCD): keep the stride a multiple.  Resident weights: a multiple of 8 * ncb,
    // so that (id >> 3) % ncb - the cout block - is the same for every item of a block
    const int gq = (WRES && L.k.ncb > 1) ? 8 * L.k.ncb : 8;
    grid -= grid % gq;
    if (grid < gq) grid = gq;
    if (grid > L.grid) grid = L.grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), L.lds, s, L.k);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

template <bool I8>
int launch_dma_cfg(const Launch& L, int cf, int pf, int nw, int stg, int il, int hc, int stride, int wres, hipStream_t s) {
    if (wres) {
        if constexpr (!I8) {
            if (stride == 1 && stg == 2 && il == 1 && cf == 2 && pf == 2 && nw == 8 && hc == 32) return launch_dma<2, 2, 8, 2, 2, 1, 32, false, 1, true>(L, s);
        }
        y6_set_error("conv_dma: no resident-weight instantiation c%dp%d x %d waves, %d-channel chunks", cf, pf, nw, hc);
        return Y6_EUNSUPPORTED;
    }
    if (stride == 2) {
        if constexpr (!I8) {
            if (hc == 16 && stg == 2 && il == 1 && cf == 2 && pf == 1 && nw == 4) return launch_dma<2, 1, 4, 2, 2, 1, 16, false, 2>(L, s);
            if (hc == 16 && stg == 2 && il == 1 && cf == 2 && pf == 1 && nw == 8) return launch_dma<2, 1, 8, 2, 2, 1, 16, false, 2>(L, s);
            if (hc == 16 && stg == 2 && il == 1 && cf == 4 && pf == 1 && nw == 8) return launch_dma<4, 1, 8, 2, 2, 1, 16, false, 2>(L, s);
        }
        y6_set_error("conv_dma: no stride-2 instantiation c%dp%d x %d waves", cf, pf, nw);
        return Y6_EUNSUPPORTED;
    }
    if (hc == 16 && stg == 2 && il == 1 && cf == 2 && pf == 2 && nw == 4) return launch_dma<2, 2, 4, 2, 2, 1, 16, I8>(L, s);
    if constexpr (I8) {   // the 8-wave 512-pixel geometries: int8 only since round 4
        if (hc == 16 && stg == 2 && il == 1 && cf == 2 && pf == 2 && nw == 8) return launch_dma<2, 2, 8, 4, 2, 1, 16, true>(L, s);
        if (hc == 32 && stg == 2 && il == 1 && cf == 2 && pf == 2 && nw == 8) return launch_dma<2, 2, 8, 2, 2, 1, 32, true>(L, s);
    } else {
        if (hc == 16 && stg == 2 && il == 1) {
            if (cf == 2 && pf == 1 && nw == 4) return launch_dma<2, 1, 4, 3, 2, 1, 16, false>(L, s);
            if (cf == 1 && pf == 2 && nw == 4) return launch_dma<1, 2, 4, 2, 2, 1, 16, false>(L, s);
            if (cf == 4 && pf == 1 && nw == 8) return launch_dma<4, 1, 8, 2, 2, 1, 16, false>(L, s);
        }
    }
    y6_set_error("conv_dma: no instantiation c%dp%d x %d waves, %d stages, il %d, %d-channel chunks", cf, pf, nw, stg, il, hc);
    return Y6_EUNSUPPORTED;
}

}  // namespace

// L points at conv_mfma.hip's launch record (same struct: conv_common.hpp)
int y6_conv_dma_launch(const void* L, int cf, int pf, int nw, int stages, int interleave, int hc, int stride, int i8, int wres, hipStream_t s) {
    const Launch& l = *static_cast<const Launch*>(L);
    return i8 ? launch_dma_cfg<true>(l, cf, pf, nw, stages, interleave, hc, stride, wres, s)
              : launch_dma_cfg<false>(l, cf, pf, nw, stages, interleave, hc, stride, wres, s);
}

int y6_conv_dma_halo_cap(int bp, int stride) {
    if (stride == 2) return bp <= 128 ? 576 : 1152;
    return bp <= 128 ? 208 : (bp <= 256 ? 352 : (bp <= 512 ? 672 : 1216));
}

extern "C" int y6_dma_probe(const void* src, unsigned bytes, unsigned lds_off, unsigned long long oob_mask, void* dst, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(src && dst && (lds_off % 16) == 0 && lds_off + 1024 <= 160 * 1024, "dma_probe: bad arguments");
    static bool big = false;
    if (!big) {
        Y6_HIP(hipFuncSetAttribute((const void*)dma_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        big = true;
    }
    hipLaunchKernelGGL(dma_probe_kernel, dim3(1), dim3(64), lds_off + 1024, (hipStream_t)stream, (const char*)src, bytes, lds_off,
                       oob_mask, (char*)dst);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
