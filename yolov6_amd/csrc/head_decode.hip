// head_decode.hip — EffiDeHead eval epilogue.
// Restates Detect.forward eval branch (reference yolov6/models/effidehead.py:93-139):
//   cls = sigmoid(cls_logits); reg = (use_dfl ? proj . softmax(bins) : reg)
//   anchor_points = (x+0.5, y+0.5), stride per level  (assigners/anchor_generator.py:13-33)
//   box = dist2bbox(reg, anchor_points, 'xywh') * stride   (utils/general.py:32-43)
//   out[b, a, :] = (cx, cy, w, h, 1.0, cls[0..nc))   fp32, levels concatenated along a.
// Four output elements per thread -> 16-byte coalesced fp32 stores; the fp16 NHWC logits of a
// pixel are contiguous so the class reads coalesce too.
#include "common.hpp"

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
