// head_decode.hip — EffiDeHead eval epilogue.
// Restates Detect.forward eval branch (reference yolov6/models/effidehead.py:93-139):
//   cls = sigmoid(cls_logits); reg = (use_dfl ? proj . softmax(bins) : reg)
//   anchor_points = (x+0.5, y+0.5), stride per level  (assigners/anchor_generator.py:13-33)
//   box = dist2bbox(reg, anchor_points, 'xywh') * stride   (utils/general.py:32-43)
//   out[b, a, :] = (cx, cy, w, h, 1.0, cls[0..nc))   fp32, levels concatenated along a.
// Four output elements per thread -> 16-byte coalesced fp32 stores; the fp16 NHWC logits of a
// pixel are contiguous so the class reads coalesce too.
#include <cstddef>

#include "common.hpp"
#include "nms_cand.hpp"
#include "plan_internal.hpp"

namespace {

struct DecodeArgs {
    int n_levels;
    const __half* cls[Y6_MAX_LEVELS];
    const __half* reg[Y6_MAX_LEVELS];
    int cls_cs[Y6_MAX_LEVELS], cls_co[Y6_MAX_LEVELS];
    int reg_cs[Y6_MAX_LEVELS], reg_co[Y6_MAX_LEVELS];
    int H[Y6_MAX_LEVELS], W[Y6_MAX_LEVELS];
    int astart[Y6_MAX_LEVELS + 1];  // first anchor of each level
    float stride[Y6_MAX_LEVELS];
    int use_dfl, reg_max;
    const float* proj;
    float cell_offset;
    float* out;
    int B, A, nc;
};

__device__ __forceinline__ float side_dist(const DecodeArgs& a, int l, size_t pix, int side) {
    const __half* r = a.reg[l] + pix * a.reg_cs[l] + a.reg_co[l];
    if (!a.use_dfl) return __half2float(r[side]);
    const int nb = a.reg_max + 1;
    const __half* bins = r + side * nb;
    float m = -INFINITY;
    for (int k = 0; k < nb; ++k) m = fmaxf(m, __half2float(bins[k]));
    // op boundaries of the reference's half-precision model (effidehead.py:107-109): F.softmax returns fp16
    // probabilities, proj_conv (fp32 accumulation) returns an fp16 distance - rounded at the same two places here
    float den = 0.f;
    for (int k = 0; k < nb; ++k) den += __expf(__half2float(bins[k]) - m);
    float num = 0.f;
    for (int k = 0; k < nb; ++k) num += y6_round_f16(__expf(__half2float(bins[k]) - m) / den) * a.proj[k];
    return y6_round_f16(num);
}

__device__ __forceinline__ float decode_element(const DecodeArgs& a, size_t i, int no) {
    const int j = (int)(i % no);
    const size_t ba = i / no;
    const int an = (int)(ba % a.A);
    const int b = (int)(ba / a.A);
    int l = 0;
#pragma unroll
    for (int t = 1; t < Y6_MAX_LEVELS; ++t)
        if (t < a.n_levels && an >= a.astart[t]) l = t;
    const int local = an - a.astart[l];
    const int y = local / a.W[l], x = local - y * a.W[l];
    const size_t pix = ((size_t)b * a.H[l] + y) * a.W[l] + x;
    if (j >= 5) {
        const float z = __half2float(a.cls[l][pix * a.cls_cs[l] + a.cls_co[l] + (j - 5)]);
        return y6_round_f16(1.f / (1.f + __expf(-z)));   // torch.sigmoid of an fp16 tensor is fp16
    }
    if (j == 4) return 1.f;
    // (l,t,r,b) distances -> xywh
    const int ax = j & 1;  // 0: x / w, 1: y / h
    const float lo = side_dist(a, l, pix, ax);       // left or top
    const float hi = side_dist(a, l, pix, 2 + ax);   // right or bottom
    const float ap = (ax == 0 ? (float)x : (float)y) + a.cell_offset;
    const float p1 = ap - lo, p2 = ap + hi;
    const float v = (j < 2) ? (p1 + p2) / 2.f : (p2 - p1);
    return v * a.stride[l];
}

// Flat variant: four consecutive output floats per thread -> one 16-byte store per lane.  Used when the
// class logits cannot be fetched as 16-byte pieces (nc % 8 != 0 or unaligned views).
__global__ __launch_bounds__(256) void head_decode_kernel(const DecodeArgs a) {
    const int no = a.nc + 5;
    const size_t total = (size_t)a.B * a.A * no;
    const size_t nvec = (total + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nvec; q += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = q * 4;
        if (i0 + 3 < total) {
            float4 v;
            v.x = decode_element(a, i0, no);
            v.y = decode_element(a, i0 + 1, no);
            v.z = decode_element(a, i0 + 2, no);
            v.w = decode_element(a, i0 + 3, no);
            *reinterpret_cast<float4*>(a.out + i0) = v;
        } else {
            for (size_t i = i0; i < total; ++i) a.out[i] = decode_element(a, i, no);
        }
    }
}

// Tiled variant: one block = DEC_TA consecutive anchors of one image.  The fp16 class logits of an anchor
// are read as 16-byte pieces (8 classes), sigmoid'ed into an LDS image of the block's [DEC_TA][5+nc] fp32
// output rows, the four box values are added by one thread per anchor, and the rows - contiguous in the
// output tensor - leave as 16-byte stores.  Reads and writes are both fully coalesced; the integer
// div/mod chain of the flat variant runs once per anchor instead of once per element.
constexpr int DEC_TA = 64;

__global__ __launch_bounds__(256) void head_decode_tiled_kernel(const DecodeArgs a, int blocks_per_image) {
    extern __shared__ __attribute__((aligned(16))) float s_rows[];   // [DEC_TA][no] then int pix[DEC_TA], lvl[DEC_TA]
    const int no = a.nc + 5, nc8 = a.nc >> 3;
    int* s_pix = reinterpret_cast<int*>(s_rows + DEC_TA * no);
    int* s_lvl = s_pix + DEC_TA;
    const int tid = threadIdx.x;
    const int b = blockIdx.x / blocks_per_image;
    const int a0 = (blockIdx.x - b * blocks_per_image) * DEC_TA;
    const int na = min(DEC_TA, a.A - a0);

    if (tid < DEC_TA && tid < na) {
        const int an = a0 + tid;
        int l = 0;
#pragma unroll
        for (int t = 1; t < Y6_MAX_LEVELS; ++t)
            if (t < a.n_levels && an >= a.astart[t]) l = t;
        const int local = an - a.astart[l];
        const int y = local / a.W[l], x = local - y * a.W[l];
        const int pix = (b * a.H[l] + y) * a.W[l] + x;
        s_pix[tid] = pix;
        s_lvl[tid] = l;
        // (l,t,r,b) distances -> xywh (utils/general.py:32-43), scaled by the level stride
        const float d0 = side_dist(a, l, (size_t)pix, 0), d1 = side_dist(a, l, (size_t)pix, 1);
        const float d2 = side_dist(a, l, (size_t)pix, 2), d3 = side_dist(a, l, (size_t)pix, 3);
        const float ax = (float)x + a.cell_offset, ay = (float)y + a.cell_offset;
        const float x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3;
        float* r = s_rows + tid * no;
        r[0] = (x1 + x2) / 2.f * a.stride[l];
        r[1] = (y1 + y2) / 2.f * a.stride[l];
        r[2] = (x2 - x1) * a.stride[l];
        r[3] = (y2 - y1) * a.stride[l];
        r[4] = 1.f;
    }
    __syncthreads();
    for (int idx = tid; idx < na * nc8; idx += 256) {
        const int t = idx / nc8, piece = idx - t * nc8;
        const int l = s_lvl[t];
        const uint4 raw = *reinterpret_cast<const uint4*>(a.cls[l] + (size_t)s_pix[t] * a.cls_cs[l] + a.cls_co[l] + piece * 8);
        const __half* h = reinterpret_cast<const __half*>(&raw);
        float* r = s_rows + t * no + 5 + piece * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = y6_round_f16(1.f / (1.f + __expf(-__half2float(h[j]))));
    }
    __syncthreads();
    const size_t obase = ((size_t)b * a.A + a0) * no;
    const int nfl = na * no;
    if ((obase & 3) == 0) {
        const int nv = nfl >> 2;
        for (int i = tid; i < nv; i += 256)
            reinterpret_cast<float4*>(a.out + obase)[i] = reinterpret_cast<const float4*>(s_rows)[i];
        for (int i = (nv << 2) + tid; i < nfl; i += 256) a.out[obase + i] = s_rows[i];
    } else {
        for (int i = tid; i < nfl; i += 256) a.out[obase + i] = s_rows[i];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Fused head tail: cls_pred + reg_pred (the two 1x1 convs of every level, effidehead.py:100-101) + the decode above in
// ONE launch.  Unfused, YOLOv6-S at b32 spends 7 launches (6 latency-bound 1x1 convs of 9-21 us + the decode) and writes /
// re-reads the [B,A,84] fp16 logits; here a block owns PD_TA consecutive anchors of one image and one level:
//   * the MFMA pixel operand of a lane IS a 16-byte piece of the anchor's NHWC feature row, loaded straight from global
//     memory (as conv1x1_stream_kernel); the weight fragments come from the packed 1x1 images (L2-resident, 10-50 KB);
//   * wave w multiplies pixel fragment (w & 1) by the cout fragments t = (w >> 1), (w >> 1) + 2, ... of [cls | reg];
//   * rounding points of the reference's fp16 graph are kept: conv output (+bias) -> fp16, sigmoid -> fp16, DFL softmax ->
//     fp16, projection -> fp16 (the arithmetic below is the decode kernel's, on the same fp16 values);
//   * logits never leave the CU: they go through the block's LDS image of the fp32 output rows.
// The k-step order of the accumulation is the unfused kernels' (one chain, ascending input channel), so the result is
// bit-identical to conv -> conv -> decode.
constexpr int PD_TA = 64;
static_assert(PD_TA <= y6cand::kCandRows, "one lane of a wave per row in y6cand::emit_candidates");

struct PredDecodeArgs {
    int n_levels;
    const __half* cf[Y6_MAX_LEVELS];       // cls_conv output view (input of cls_pred)
    const __half* rf[Y6_MAX_LEVELS];       // reg_conv output view (input of reg_pred)
    int cf_cs[Y6_MAX_LEVELS], cf_co[Y6_MAX_LEVELS], rf_cs[Y6_MAX_LEVELS], rf_co[Y6_MAX_LEVELS];
    const __half* wc[Y6_MAX_LEVELS];       // packed 1x1 weights (y6_pack_conv_weight)
    const __half* wr[Y6_MAX_LEVELS];
    const float* bc[Y6_MAX_LEVELS];
    const float* br[Y6_MAX_LEVELS];
    int C[Y6_MAX_LEVELS];                  // input channels of the level's pred convs (multiple of 16)
    int H[Y6_MAX_LEVELS], W[Y6_MAX_LEVELS];
    int astart[Y6_MAX_LEVELS + 1];         // first anchor of each level
    int bstart[Y6_MAX_LEVELS + 1];         // first block (inside one image) of each level
    float stride[Y6_MAX_LEVELS];
    int use_dfl, reg_max, nreg;
    const float* proj;
    float cell_offset;
    float* out;
    int B, A, nc;
    int ncf_c, ncf_r;                      // cout fragments (32 channels) of cls_pred / reg_pred
    // candidate sink (y6_nms_sink): key lists [B][cand_cap] and their lengths, or cand_keys == nullptr
    unsigned long long* cand_keys;
    size_t cand_cap;
    int* cand_counts;
    float cand_conf;
    const int* cand_classes;
    int cand_nclasses, cand_ml;
};

__device__ __forceinline__ float side_dist_lds(const PredDecodeArgs& a, const float* r, int side) {
    if (!a.use_dfl) return r[side];
    const int nb = a.reg_max + 1;
    const float* bins = r + side * nb;
    float m = -INFINITY;
    for (int k = 0; k < nb; ++k) m = fmaxf(m, bins[k]);
    float den = 0.f;
    for (int k = 0; k < nb; ++k) den += __expf(bins[k] - m);
    float num = 0.f;
    for (int k = 0; k < nb; ++k) num += y6_round_f16(__expf(bins[k] - m) / den) * a.proj[k];
    return y6_round_f16(num);
}

__global__ __launch_bounds__(256) void head_pred_decode_kernel(const PredDecodeArgs a, int blocks_per_image) {
    extern __shared__ __attribute__((aligned(16))) float s_rows[];   // [PD_TA][no] output rows, then [PD_TA][nreg] reg values, then PD_TA row flags
    __shared__ int s_cnt, s_base, s_gen;
    const int no = a.nc + 5;
    float* s_reg = s_rows + PD_TA * no;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / blocks_per_image;
    const int rb = blockIdx.x - b * blocks_per_image;
    int l = 0;
#pragma unroll
    for (int t = 1; t < Y6_MAX_LEVELS; ++t)
        if (t < a.n_levels && rb >= a.bstart[t]) l = t;
    const int HWl = a.H[l] * a.W[l];
    const int a0 = (rb - a.bstart[l]) * PD_TA;               // first anchor of this block inside its level
    const int na = min(PD_TA, HWl - a0);

    // ---- the two 1x1 convs: this wave's pixel fragment x its cout fragments
    const int pfrag = wave & 1, half = wave >> 1;
    const int p = pfrag * 32 + (lane & 31);                  // anchor slot of this lane inside the block
    const int kh = lane >> 5;
    const int pl = a0 + (p < na ? p : na - 1);               // clamped: rows beyond the level are computed and dropped
    const size_t pix = (size_t)b * HWl + pl;
    const __half* crow = a.cf[l] + pix * a.cf_cs[l] + a.cf_co[l] + kh * 8;
    const __half* rrow = a.rf[l] + pix * a.rf_cs[l] + a.rf_co[l] + kh * 8;
    const int KS = a.C[l] >> 4;
    const int nchunk = (a.C[l] + 31) >> 5;
    const int ntot = a.ncf_c + a.ncf_r;
    for (int t = half; t < ntot; t += 2) {
        const bool is_cls = t < a.ncf_c;
        const int cfi = is_cls ? t : t - a.ncf_c;
        const __half* xrow = is_cls ? crow : rrow;
        const __half* wbase = (is_cls ? a.wc[l] : a.wr[l]) + (size_t)cfi * nchunk * 1024 + lane * 8;   // [cfr][chunk][tap=1][ks][lane][8]
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int ks = 0;
        for (; ks + 4 <= KS; ks += 4) {                      // four k-steps of operands in flight
            h8_t xa[4], wa[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xa[j] = *reinterpret_cast<const h8_t*>(xrow + (ks + j) * 16);
                wa[j] = *reinterpret_cast<const h8_t*>(wbase + (ks + j) * 512);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[j], xa[j], acc, 0, 0, 0);
        }
        for (; ks < KS; ++ks) {
            const h8_t xa = *reinterpret_cast<const h8_t*>(xrow + ks * 16);
            const h8_t wa = *reinterpret_cast<const h8_t*>(wbase + ks * 512);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xa, acc, 0, 0, 0);
        }
        // C/D layout: col = pixel (lane & 31), row = cout = (r & 3) + 8 * (r >> 2) + 4 * kh
        if (is_cls) {
            float* row = s_rows + p * no + 5;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cfi * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                if (c < a.nc) {
                    const float z = y6_round_f16(acc[r] + a.bc[l][c]);                 // the conv's fp16 output
                    row[c] = y6_round_f16(1.f / (1.f + __expf(-z)));                   // torch.sigmoid of an fp16 tensor is fp16
                }
            }
        } else {
            float* row = s_reg + p * a.nreg;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cfi * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                if (c < a.nreg) row[c] = y6_round_f16(acc[r] + a.br[l][c]);
            }
        }
    }
    __syncthreads();
    // ---- boxes: one thread per anchor (the decode kernel's arithmetic)
    if (tid < na) {
        const int local = a0 + tid;
        const int y = local / a.W[l], x = local - y * a.W[l];
        const float* rg = s_reg + tid * a.nreg;
        const float d0 = side_dist_lds(a, rg, 0), d1 = side_dist_lds(a, rg, 1);
        const float d2 = side_dist_lds(a, rg, 2), d3 = side_dist_lds(a, rg, 3);
        const float ax = (float)x + a.cell_offset, ay = (float)y + a.cell_offset;
        const float x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3;
        float* r = s_rows + tid * no;
        r[0] = (x1 + x2) / 2.f * a.stride[l];
        r[1] = (y1 + y2) / 2.f * a.stride[l];
        r[2] = (x2 - x1) * a.stride[l];
        r[3] = (y2 - y1) * a.stride[l];
        r[4] = 1.f;
    }
    __syncthreads();
    // The NMS candidates of these rows, while they are here (nms_cand.hpp: y6_nms's own first stage).  The selection is LDS work
    // and runs BEFORE the row stores; one thread reserves the block's slice of the image's key list, the row stores leave behind
    // that atomic, and its result is only awaited where the keys are written (a wave's memory results return in order: anything
    // that waits behind the 21 KB of row stores of a block pays their acknowledgement, r04q).
    const bool sink = a.cand_keys != nullptr;
    const y6cand::lds_f32* lrows = (const y6cand::lds_f32*)s_rows;
    y6cand::CandSel2 cs;
    cs.fast = 0;
    int cand_total = 0, cand_base = 0;
    if (sink) {
        cs = y6cand::cand_select(lrows, (y6cand::lds_i32*)reinterpret_cast<int*>(s_reg + PD_TA * a.nreg), na, no, a.nc, a.cand_conf, a.cand_classes,
                                 a.cand_nclasses, a.cand_ml, (y6cand::lds_i32*)&s_cnt, (y6cand::lds_i32*)&s_gen);
        cand_total = *(y6cand::lds_i32*)&s_cnt;
        if (tid == 0 && cand_total > 0) cand_base = atomicAdd(a.cand_counts + b * y6cand::kCountStride, cand_total);
    }
    const size_t obase = ((size_t)b * a.A + a.astart[l] + a0) * no;
    const int nfl = na * no;
    if ((obase & 3) == 0) {
        const int nv = nfl >> 2;
        for (int i = tid; i < nv; i += 256)
            reinterpret_cast<float4*>(a.out + obase)[i] = reinterpret_cast<const float4*>(s_rows)[i];
        for (int i = (nv << 2) + tid; i < nfl; i += 256) a.out[obase + i] = s_rows[i];
    } else {
        for (int i = tid; i < nfl; i += 256) a.out[obase + i] = s_rows[i];
    }
    if (sink && cand_total > 0) {
        if (tid == 0) *(y6cand::lds_i32*)&s_base = cand_base;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // LDS only: nobody waits for row stores here
        y6cand::cand_publish(lrows, na, no, a.nc, a.astart[l] + a0, a.cand_ml, cs, a.cand_keys + (size_t)b * a.cand_cap + *(y6cand::lds_i32*)&s_base);
    }
}

}  // namespace

extern "C" int y6_head_decode(const y6_decode_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d && d->out && d->n_levels >= 1 && d->n_levels <= Y6_MAX_LEVELS, "head_decode: bad descriptor");
    Y6_REQUIRE(!d->use_dfl || d->proj, "head_decode: use_dfl needs proj");
    DecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.n_levels = d->n_levels;
    int A = 0;
    const int nreg = 4 * (d->use_dfl ? d->reg_max + 1 : 1);
    for (int l = 0; l < d->n_levels; ++l) {
        const y6_tensor &c = d->cls[l], &r = d->reg[l];
        Y6_REQUIRE(c.data && r.data, "head_decode: level %d null tensor", l);
        Y6_REQUIRE(c.C == d->nc && r.C == nreg, "head_decode: level %d channels cls %d (want %d) reg %d (want %d)", l, c.C,
                   d->nc, r.C, nreg);
        Y6_REQUIRE(c.B == r.B && c.H == r.H && c.W == r.W && c.B == d->cls[0].B, "head_decode: level %d shape mismatch", l);
        a.cls[l] = (const __half*)c.data;
        a.reg[l] = (const __half*)r.data;
        a.cls_cs[l] = c.cstride;
        a.cls_co[l] = c.coff;
        a.reg_cs[l] = r.cstride;
        a.reg_co[l] = r.coff;
        a.H[l] = c.H;
        a.W[l] = c.W;
        a.astart[l] = A;
        a.stride[l] = d->stride[l];
        A += c.H * c.W;
    }
    a.astart[d->n_levels] = A;
    a.use_dfl = d->use_dfl;
    a.reg_max = d->reg_max;
    a.proj = d->proj;
    a.cell_offset = d->grid_cell_offset;
    a.out = d->out;
    a.B = d->cls[0].B;
    a.A = A;
    a.nc = d->nc;
    const size_t total = (size_t)a.B * A * (d->nc + 5);
    Y6_REQUIRE(((uintptr_t)d->out & 15) == 0, "head_decode: output must be 16-byte aligned");
    // tiled path: class logits fetchable as 16-byte pieces on every level
    bool tiled = (d->nc % 8 == 0);
    for (int l = 0; l < d->n_levels; ++l)
        tiled = tiled && (a.cls_cs[l] % 8 == 0) && (a.cls_co[l] % 8 == 0) && (((uintptr_t)a.cls[l] & 15) == 0);
    static const bool no_tiled = getenv("Y6_DECODE_FLAT") != nullptr;   // A/B switch for profiling
    // the tiled kernel keeps DEC_TA fp32 output rows in LDS: class counts whose rows do not fit take the flat kernel
    const size_t lds = (size_t)DEC_TA * (d->nc + 5) * sizeof(float) + 2 * DEC_TA * sizeof(int);
    tiled = tiled && lds <= 64 * 1024;
    if (tiled && !no_tiled) {
        const int bpi = (A + DEC_TA - 1) / DEC_TA;
        hipLaunchKernelGGL(head_decode_tiled_kernel, dim3((unsigned)(a.B * bpi)), dim3(256), lds, (hipStream_t)stream, a, bpi);
        Y6_LAUNCH_CHECK();
        return Y6_OK;
    }
    size_t g = ((total + 3) / 4 + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    hipLaunchKernelGGL(head_decode_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

// ---- fused head tail (descriptor: include/yolov6_hip.h y6_pred_decode_desc)
extern "C" int y6_head_pred_decode_supported(const y6_pred_decode_desc* d) {
    if (!d || d->n_levels < 1 || d->n_levels > Y6_MAX_LEVELS || !d->out) return 0;
    const int nreg = 4 * (d->use_dfl ? d->reg_max + 1 : 1);
    const size_t lds = (size_t)PD_TA * (d->nc + 5 + nreg) * sizeof(float) + PD_TA * sizeof(int);
    if (lds > 64 * 1024) return 0;
    for (int l = 0; l < d->n_levels; ++l) {
        const y6_tensor &c = d->cls_feat[l], &r = d->reg_feat[l];
        if (!c.data || !r.data || !d->w_cls[l] || !d->w_reg[l]) return 0;
        if (c.C != r.C || c.C % 16 != 0 || c.B != r.B || c.H != r.H || c.W != r.W || c.B != d->cls_feat[0].B) return 0;
        if (c.cstride % 8 || c.coff % 8 || r.cstride % 8 || r.coff % 8 || ((uintptr_t)c.data & 15) || ((uintptr_t)r.data & 15)) return 0;
        if (((uintptr_t)d->w_cls[l] & 15) || ((uintptr_t)d->w_reg[l] & 15)) return 0;
    }
    if (d->total_anchors != 0 || d->first_anchor != 0) {   // a subset of the levels, written into a larger tensor
        long A = 0;
        for (int l = 0; l < d->n_levels; ++l) A += (long)d->cls_feat[l].H * d->cls_feat[l].W;
        if (d->first_anchor < 0 || d->total_anchors <= 0 || d->first_anchor + A > d->total_anchors) return 0;
    }
    if (d->cand.workspace != nullptr) {   // candidate sink: the whole tensor in one call, the rows of a block within the selection's task budget
        if (d->total_anchors != 0 || d->first_anchor != 0) return 0;
        if ((long)PD_TA * d->nc > 64 * 256) return 0;   // y6cand::emit_candidates: a 64-bit pass mask per thread
        long A = 0;
        for (int l = 0; l < d->n_levels; ++l) A += (long)d->cls_feat[l].H * d->cls_feat[l].W;
        if (d->cand.workspace_bytes < y6_nms_workspace_bytes(d->cls_feat[0].B, (int)A, d->nc, d->cand.multi_label)) return 0;
        if ((size_t)A * d->nc >= 0xFFFFFFFFull) return 0;
    }
    return ((uintptr_t)d->out & 15) == 0;
}

static int pred_decode_launch(const y6_pred_decode_desc* d, hipStream_t stream) {
    Y6_REQUIRE(y6_head_pred_decode_supported(d), "head_pred_decode: unsupported descriptor (channels %% 16, 16-byte aligned views)");
    Y6_REQUIRE(!d->use_dfl || d->proj, "head_pred_decode: use_dfl needs proj");
    PredDecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.n_levels = d->n_levels;
    int A = 0, nb = 0;
    for (int l = 0; l < d->n_levels; ++l) {
        const y6_tensor &c = d->cls_feat[l], &r = d->reg_feat[l];
        a.cf[l] = (const __half*)c.data;
        a.rf[l] = (const __half*)r.data;
        a.cf_cs[l] = c.cstride;
        a.cf_co[l] = c.coff;
        a.rf_cs[l] = r.cstride;
        a.rf_co[l] = r.coff;
        a.wc[l] = (const __half*)d->w_cls[l];
        a.wr[l] = (const __half*)d->w_reg[l];
        a.bc[l] = d->b_cls[l];
        a.br[l] = d->b_reg[l];
        Y6_REQUIRE(a.bc[l] && a.br[l], "head_pred_decode: level %d bias missing", l);
        a.C[l] = c.C;
        a.H[l] = c.H;
        a.W[l] = c.W;
        a.astart[l] = d->first_anchor + A;
        a.bstart[l] = nb;
        a.stride[l] = d->stride[l];
        A += c.H * c.W;
        nb += (c.H * c.W + PD_TA - 1) / PD_TA;
    }
    a.astart[d->n_levels] = d->first_anchor + A;
    a.bstart[d->n_levels] = nb;
    a.use_dfl = d->use_dfl;
    a.reg_max = d->reg_max;
    a.nreg = 4 * (d->use_dfl ? d->reg_max + 1 : 1);
    a.proj = d->proj;
    a.cell_offset = d->grid_cell_offset;
    a.out = d->out;
    a.B = d->cls_feat[0].B;
    a.A = d->total_anchors > 0 ? d->total_anchors : A;
    a.nc = d->nc;
    a.ncf_c = (d->nc + 31) / 32;
    a.ncf_r = (a.nreg + 31) / 32;
    if (d->cand.workspace != nullptr) {
        int rc = y6_nms_workspace_views(d->cand.workspace, d->cand.workspace_bytes, a.B, a.A, a.nc, d->cand.multi_label, &a.cand_keys, &a.cand_cap,
                                        &a.cand_counts);
        if (rc) return rc;
        a.cand_conf = d->cand.conf_thres;
        a.cand_classes = d->cand.classes;
        a.cand_nclasses = d->cand.n_classes;
        a.cand_ml = d->cand.multi_label && d->nc > 1;
        Y6_HIP(hipMemsetAsync(a.cand_counts, 0, (size_t)a.B * y6cand::kCountStride * sizeof(int), stream));
    }
    const size_t lds = (size_t)PD_TA * (d->nc + 5 + a.nreg) * sizeof(float) + PD_TA * sizeof(int);
    hipLaunchKernelGGL(head_pred_decode_kernel, dim3((unsigned)(a.B * nb)), dim3(256), lds, stream, a, nb);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

extern "C" int y6_head_pred_decode(const y6_pred_decode_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d, "head_pred_decode: null descriptor");
    return pred_decode_launch(d, (hipStream_t)stream);
}

extern "C" int y6_plan_add_pred_decode(y6_plan* p, const y6_pred_decode_desc* d) {
    Y6_REQUIRE(p && d, "plan_add_pred_decode: null argument");
    Y6_REQUIRE(y6_head_pred_decode_supported(d), "plan_add_pred_decode: unsupported descriptor");
    double fl = 0.0, by = 0.0, A = 0.0;
    const int nreg = 4 * (d->use_dfl ? d->reg_max + 1 : 1);
    for (int l = 0; l < d->n_levels; ++l) {
        const y6_tensor& c = d->cls_feat[l];
        const double px = (double)c.B * c.H * c.W;
        fl += 2.0 * px * c.C * (d->nc + nreg);
        by += 2.0 * px * 2.0 * c.C + 2.0 * c.C * (d->nc + nreg);
        A += (double)c.H * c.W;
    }
    by += (double)d->cls_feat[0].B * A * (d->nc + 5) * 4.0;
    int rc = y6_plan_push(p, pred_decode_launch, d, Y6_TOP_PRED_DECODE, fl, by);
    if (rc) return rc;
    return y6_plan_mark_output(p, offsetof(y6_pred_decode_desc, out));
}
