// tal.hip — Task-aligned assigner without the [B,G,topk,A] one-hot temp.
// Restates TaskAlignedAssigner.forward (reference yolov6/assigners/tal_assigner.py:22-106):
//   T1 (per (b,g) block)  get_box_metrics :125-141 + select_candidates_in_gts
//        (assigner_utils.py:25-44) + select_topk_candidates :143-158: the metric row lives in
//        LDS, top-k is k rounds of a block arg-max (value desc, anchor index asc), winners
//        that are inside the gt bump fg_cnt[b,a] and atomicMin first_g[b,a].
//   T2 (per (b,a) thread) select_highest_overlaps (assigner_utils.py:46-67): anchors claimed
//        by >1 gt go to argmax_g IoU over ALL gts (first max), others keep their gt; the
//        chosen (b,g,a) metric / IoU feed atomicMax pos_align[b,g], pos_ov[b,g]  (:76-78).
//   T3a/T3b  get_targets :160-181 + normalisation :76-81:
//        norm = align * pos_ov / (pos_align + eps); target_scores = one_hot(label) * norm.
// Ties: torch.topk's tie order is unspecified; this kernel (and the oracle) define
// "lower anchor index wins" (SURVEY §7 hard parts).  Compile with -ffp-contract=off.
#include "common.hpp"

namespace {

__device__ __forceinline__ float iou_gt_pd(const float4 g, const float4 p, float eps) {
    // iou_calculator(box1=gt, box2=pd)  assigner_utils.py:69-89
    const float x1 = fmaxf(g.x, p.x), y1 = fmaxf(g.y, p.y);
    const float x2 = fminf(g.z, p.z), y2 = fminf(g.w, p.w);
    const float ow = fmaxf(x2 - x1, 0.f), oh = fmaxf(y2 - y1, 0.f);
    const float overlap = ow * oh;
    const float a1 = fmaxf(g.z - g.x, 0.f) * fmaxf(g.w - g.y, 0.f);
    const float a2 = fmaxf(p.z - p.x, 0.f) * fmaxf(p.w - p.y, 0.f);
    const float uni = a1 + a2 - overlap + eps;
    return overlap / uni;
}

__device__ __forceinline__ float align_metric(float score, float iou, float alpha, float beta) {
    const float s = (alpha == 1.f) ? score : powf(score, alpha);
    const float o = powf(iou, beta);
    return s * o;
}

__device__ __forceinline__ bool in_gt(const float2 c, const float4 g, float eps) {
    const float m = fminf(fminf(c.x - g.x, c.y - g.y), fminf(g.z - c.x, g.w - c.y));
    return m > eps;
}

__device__ __forceinline__ int gt_label(const float* gt_labels, size_t bg) {
    return (int)(long long)gt_labels[bg];
}

struct TalArgs {
    const float* pd_scores;
    const float4* pd_bboxes;
    const float2* anc;
    const float* gt_labels;
    const float4* gt_bboxes;
    const float* mask_gt;
    int B, A, C, G, topk;
    float alpha, beta, eps;
    int* fg_cnt;
    int* first_g;
    int* assign;
    float* norm;
    unsigned* pos_am;
    unsigned* pos_ov;
    long long* target_labels;
    float4* target_bboxes;
    float* target_scores;
    unsigned char* fg_mask;
};

__global__ __launch_bounds__(256) void tal_topk_lds_kernel(const TalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* vals = reinterpret_cast<float*>(smem);  // [A]
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const int bg = blockIdx.x;
    const int b = bg / a.G;
    if (!(a.mask_gt[bg] != 0.f)) return;  // padded gt: topk_idxs forced to 0, then de-duplicated to nothing (:152-156)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 g = a.gt_bboxes[bg];
    const int label = gt_label(a.gt_labels, bg);
    const float* sc = a.pd_scores + (size_t)b * a.A * a.C + label;
    const float4* pb = a.pd_bboxes + (size_t)b * a.A;
    // metrics * mask_in_gts (:116).  Only the anchors inside the box (2-3 % of them for a typical COCO box) have a metric that can
    // survive the mask, so only they pay for it: the class score of an anchor is one float out of every C = 80 (a cache line per
    // anchor: 1 MB of lines per box, 1 751 boxes per b64 step - the kernel was 280 us of line traffic), the anchor points are 67 KB
    // of coalesced reads.  For finite metrics 0 is what `0 * m` gave (a NaN metric outside the box no longer stays a NaN here: such
    // an anchor could never be kept anyway - mask_pos needs in_gt below).
    for (int an = tid; an < a.A; an += 256) {
        float v = 0.f;
        if (in_gt(a.anc[an], g, 1e-9f)) {
            const float iou = iou_gt_pd(g, pb[an], 1e-9f);
            v = align_metric(sc[(size_t)an * a.C], iou, a.alpha, a.beta);
        }
        vals[an] = v;
    }
    __syncthreads();
    const int k = a.topk < a.A ? a.topk : a.A;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int an = tid; an < a.A; an += 256) {
            const float v = vals[an];
            if (v > bv) {  // ascending scan: first (lowest index) max per thread
                bv = v;
                bi = an;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_v[wave] = bv;
            s_i[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) {
                    bv = s_v[w];
                    bi = s_i[w];
                }
            if (bi < a.A) {
                vals[bi] = -INFINITY;  // taken
                if (in_gt(a.anc[bi], g, 1e-9f)) {  // mask_pos = mask_topk * mask_in_gts * mask_gt (:121)
                    atomicAdd(&a.fg_cnt[(size_t)b * a.A + bi], 1);
                    atomicMin(&a.first_g[(size_t)b * a.A + bi], bg - b * a.G);
                }
            }
        }
        __syncthreads();
    }
}

// The same selection with the metric row in REGISTERS (A <= 256 * NV: the 8 400 anchors of a 640 x 640 image are 33 per thread): the
// topk rounds compare registers instead of re-reading the whole row from LDS thirteen times (tal_topk_lds_kernel above serves the
// larger maps).  Same comparisons in the same order - per thread ascending, first maximum; across lanes and waves the larger value,
// the lower index on a tie - so the same anchors (bit-exact goldens: tests/test_gpu_nms_tal.py, test_gpu_b64_parity.py).
template <int NV>
__global__ __launch_bounds__(256) void tal_topk_reg_kernel(const TalArgs a) {
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    __shared__ int s_pick;
    const int bg = blockIdx.x;
    const int b = bg / a.G;
    if (!(a.mask_gt[bg] != 0.f)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 g = a.gt_bboxes[bg];
    const int label = gt_label(a.gt_labels, bg);
    const float* sc = a.pd_scores + (size_t)b * a.A * a.C + label;
    const float4* pb = a.pd_bboxes + (size_t)b * a.A;
    float v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int an = tid + j * 256;
        float m = -INFINITY;                   // beyond A: never the maximum (bi stays >= A only if nothing is left)
        if (an < a.A) {
            m = 0.f;
            if (in_gt(a.anc[an], g, 1e-9f)) {
                const float iou = iou_gt_pd(g, pb[an], 1e-9f);
                m = align_metric(sc[(size_t)an * a.C], iou, a.alpha, a.beta);
            }
        }
        v[j] = m;
    }
    const int k = a.topk < a.A ? a.topk : a.A;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int an = tid + j * 256;
            if (an < a.A && v[j] > bv) {
                bv = v[j];
                bi = an;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_v[wave] = bv;
            s_i[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) {
                    bv = s_v[w];
                    bi = s_i[w];
                }
            s_pick = bi;
            if (bi < a.A && in_gt(a.anc[bi], g, 1e-9f)) {
                atomicAdd(&a.fg_cnt[(size_t)b * a.A + bi], 1);
                atomicMin(&a.first_g[(size_t)b * a.A + bi], bg - b * a.G);
            }
        }
        __syncthreads();
        const int pick = s_pick;
        if (pick < a.A && (pick & 255) == tid) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (j == (pick >> 8)) v[j] = -INFINITY;     // taken
        }
    }
}

__global__ __launch_bounds__(256) void tal_resolve_kernel(const TalArgs a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)a.B * a.A) return;
    const int b = (int)(i / a.A), an = (int)(i % a.A);
    const int cnt = a.fg_cnt[i];
    int gsel = -1;
    const float4 p = a.pd_bboxes[i];
    if (cnt == 1) {
        gsel = a.first_g[i];
    } else if (cnt > 1) {
        // overlaps.argmax(axis=1) over ALL gts, first max (assigner_utils.py:60-63)
        float best = -INFINITY;
        for (int g = 0; g < a.G; ++g) {
            const float o = iou_gt_pd(a.gt_bboxes[(size_t)b * a.G + g], p, 1e-9f);
            if (o > best) {
                best = o;
                gsel = g;
            }
        }
    }
    a.assign[i] = gsel;
    if (gsel >= 0) {
        const size_t bg = (size_t)b * a.G + gsel;
        const float iou = iou_gt_pd(a.gt_bboxes[bg], p, 1e-9f);
        const int label = gt_label(a.gt_labels, bg);
        const float m = align_metric(a.pd_scores[i * a.C + label], iou, a.alpha, a.beta);
        atomicMax(&a.pos_am[bg], __float_as_uint(m));    // values are >= 0: uint order == float order
        atomicMax(&a.pos_ov[bg], __float_as_uint(iou));
        a.norm[i] = m;  // raw align metric; normalised in T3a
    }
}

__global__ __launch_bounds__(256) void tal_targets_kernel(const TalArgs a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)a.B * a.A) return;
    const int b = (int)(i / a.A);
    const int gsel = a.assign[i];
    const int g0 = gsel >= 0 ? gsel : 0;  // mask_pos.argmax(-2) of an all-zero column is 0 (:66)
    const size_t bg = (size_t)b * a.G + g0;
    int label = gt_label(a.gt_labels, bg);
    if (label < 0) label = 0;  // target_labels[target_labels<0] = 0 (:172)
    a.target_labels[i] = label;
    a.target_bboxes[i] = a.gt_bboxes[bg];
    a.fg_mask[i] = gsel >= 0 ? 1 : 0;
    float nrm = 0.f;
    if (gsel >= 0) {
        const float am = a.norm[i];
        const float pov = __uint_as_float(a.pos_ov[bg]);
        const float pam = __uint_as_float(a.pos_am[bg]);
        nrm = am * pov / (pam + a.eps);
    }
    a.norm[i] = nrm;
}

__global__ __launch_bounds__(256) void tal_scores_kernel(const TalArgs a) {
    const size_t total = (size_t)a.B * a.A * a.C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t ba = i / a.C;
        const int c = (int)(i - ba * a.C);
        float v = 0.f;
        if (a.assign[ba] >= 0 && (long long)c == a.target_labels[ba]) v = a.norm[ba];
        a.target_scores[i] = v;
    }
}

__global__ void tal_empty_kernel(const TalArgs a) {
    // n_max_boxes == 0 early-out (:48-53): labels = bg_idx, everything else zero
    const size_t total = (size_t)a.B * a.A;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        a.target_labels[i] = a.C;
        a.target_bboxes[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.fg_mask[i] = 0;
    }
}

// ============================================================================================
// ATSS assigner.  Restates ATSSAssigner.forward (reference yolov6/assigners/atss_assigner.py:18-86):
//   A1 (per (b,g) block)  bbox_overlaps(gt, anchor) iou2d_calculator.py:63-241 (union = max(.,1e-6)),
//        dist_calculator assigner_utils.py:4-23, per-level top-k NEAREST anchors
//        (select_topk_candidates :88-115), threshold = mean + unbiased std of the candidates' IoUs
//        (thres_calculator :117-136), positives = candidates with IoU > thr whose centre is inside
//        the gt (:64-71).  The distance row lives in LDS; top-k is k rounds of a block arg-min.
//   A2 (per (b,a))  select_highest_overlaps (assigner_utils.py:46-67) on the ANCHOR-box IoUs,
//        soft label IoU(gt, predicted box) (:80-84).
//   A3  get_targets :138-161: background label = num_classes, one-hot without the bg column.
// Tie rule (torch.topk leaves it open): smaller distance first, then lower anchor index.
struct AtssArgs {
    const float4* anc;         // [A] anchor boxes
    const float* gt_labels;
    const float4* gt_bboxes;
    const float* mask_gt;
    const float4* pd_bboxes;   // [B,A] or null
    int B, A, C, G, topk, n_levels;
    int lvl_start[Y6_MAX_LEVELS + 1];
    int* fg_cnt;
    int* first_g;
    int* assign;
    float* norm;               // soft-label IoU per anchor
    long long* target_labels;
    float4* target_bboxes;
    float* target_scores;
    unsigned char* fg_mask;
};

__device__ __forceinline__ float iou_mmdet(const float4 g, const float4 q) {
    // bbox_overlaps(mode='iou', is_aligned=False, eps=1e-6)  iou2d_calculator.py:186-241
    const float area1 = (g.z - g.x) * (g.w - g.y);
    const float area2 = (q.z - q.x) * (q.w - q.y);
    const float w = fmaxf(fminf(g.z, q.z) - fmaxf(g.x, q.x), 0.f);
    const float h = fmaxf(fminf(g.w, q.w) - fmaxf(g.y, q.y), 0.f);
    const float overlap = w * h;
    const float uni = fmaxf(area1 + area2 - overlap, 1e-6f);
    return overlap / uni;
}

constexpr int ATSS_MAX_CAND = 4 * 16;   // levels x topk held in LDS

__global__ __launch_bounds__(256) void atss_candidates_kernel(const AtssArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dist = reinterpret_cast<float*>(smem);  // [A]
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    __shared__ int s_cand[ATSS_MAX_CAND];
    __shared__ int s_ncand;
    const int bg = blockIdx.x;
    const int b = bg / a.G;
    if (!(a.mask_gt[bg] != 0.f)) return;   // masked gt: candidate indices forced to 0 and de-duplicated away (:104-108)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 g = a.gt_bboxes[bg];
    const float gcx = (g.x + g.z) / 2.0f, gcy = (g.y + g.w) / 2.0f;
    for (int an = tid; an < a.A; an += 256) {
        const float4 q = a.anc[an];
        const float dx = gcx - (q.x + q.z) / 2.0f, dy = gcy - (q.y + q.w) / 2.0f;
        dist[an] = sqrtf(dx * dx + dy * dy);
    }
    if (tid == 0) s_ncand = 0;
    __syncthreads();
    for (int l = 0; l < a.n_levels; ++l) {
        const int lo = a.lvl_start[l], hi = a.lvl_start[l + 1];
        const int k = a.topk < (hi - lo) ? a.topk : (hi - lo);
        for (int r = 0; r < k; ++r) {
            float bv = INFINITY;
            int bi = 0x7fffffff;
            for (int an = lo + tid; an < hi; an += 256) {
                const float v = dist[an];
                if (v < bv) {
                    bv = v;
                    bi = an;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov < bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (lane == 0) {
                s_v[wave] = bv;
                s_i[wave] = bi;
            }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < 4; ++w)
                    if (s_v[w] < bv || (s_v[w] == bv && s_i[w] < bi)) {
                        bv = s_v[w];
                        bi = s_i[w];
                    }
                if (bi < a.A) {
                    dist[bi] = INFINITY;  // taken
                    s_cand[s_ncand++] = bi;
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        const int n = s_ncand;
        float iou[ATSS_MAX_CAND];
        float sum = 0.f;
        for (int i = 0; i < n; ++i) {
            iou[i] = iou_mmdet(g, a.anc[s_cand[i]]);
            sum += iou[i];
        }
        const float mean = sum / (float)n;
        double m2 = 0.0;
        for (int i = 0; i < n; ++i) {
            const double dv = (double)iou[i] - (double)mean;
            m2 += dv * dv;
        }
        const float sd = n > 1 ? (float)sqrt(m2 / (double)(n - 1)) : NAN;   // torch.std: unbiased; NaN for one sample
        const float thr = mean + sd;
        for (int i = 0; i < n; ++i) {
            if (!(iou[i] > thr)) continue;
            const int an = s_cand[i];
            const float4 q = a.anc[an];
            const float2 c = make_float2((q.x + q.z) / 2.0f, (q.y + q.w) / 2.0f);
            if (in_gt(c, g, 1e-9f)) {
                atomicAdd(&a.fg_cnt[(size_t)b * a.A + an], 1);
                atomicMin(&a.first_g[(size_t)b * a.A + an], bg - b * a.G);
            }
        }
    }
}

__global__ __launch_bounds__(256) void atss_targets_kernel(const AtssArgs a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)a.B * a.A) return;
    const int b = (int)(i / a.A), an = (int)(i % a.A);
    const int cnt = a.fg_cnt[i];
    int gsel = -1;
    if (cnt == 1) {
        gsel = a.first_g[i];
    } else if (cnt > 1) {
        const float4 q = a.anc[an];
        float best = -INFINITY;
        for (int g = 0; g < a.G; ++g) {
            const float o = iou_mmdet(a.gt_bboxes[(size_t)b * a.G + g], q);
            if (o > best) {
                best = o;
                gsel = g;
            }
        }
    }
    a.assign[i] = gsel;
    const size_t bg = (size_t)b * a.G + (gsel >= 0 ? gsel : 0);
    a.target_bboxes[i] = a.gt_bboxes[bg];
    a.target_labels[i] = gsel >= 0 ? (long long)a.gt_labels[bg] : (long long)a.C;
    a.fg_mask[i] = gsel >= 0 ? 1 : 0;
    float soft = gsel >= 0 ? 1.f : 0.f;
    if (gsel >= 0 && a.pd_bboxes) soft = iou_gt_pd(a.gt_bboxes[bg], a.pd_bboxes[i], 1e-9f);
    a.norm[i] = soft;
}

__global__ __launch_bounds__(256) void atss_scores_kernel(const AtssArgs a) {
    const size_t total = (size_t)a.B * a.A * a.C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t ba = i / a.C;
        const int c = (int)(i - ba * a.C);
        float v = 0.f;
        if (a.assign[ba] >= 0 && (long long)c == a.target_labels[ba]) v = a.norm[ba];
        a.target_scores[i] = v;
    }
}

__global__ void atss_empty_kernel(const AtssArgs a) {
    const size_t total = (size_t)a.B * a.A;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        a.target_labels[i] = a.C;
        a.target_bboxes[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.fg_mask[i] = 0;
    }
}

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t y6_tal_workspace_bytes(int B, int A, int G) {
    const size_t ba = (size_t)B * A * 4, bgn = (size_t)B * (G > 0 ? G : 1) * 4;
    return 4 * al256(ba) + 2 * al256(bgn);
}

extern "C" int y6_tal_assign(const y6_tal_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d && d->pd_scores && d->pd_bboxes && d->anc_points && d->target_labels && d->target_bboxes &&
                   d->target_scores && d->fg_mask,
               "tal_assign: null argument");
    Y6_REQUIRE(d->B > 0 && d->A > 0 && d->C > 0 && d->G >= 0 && d->topk > 0, "tal_assign: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    TalArgs a;
    memset(&a, 0, sizeof(a));
    a.pd_scores = d->pd_scores;
    a.pd_bboxes = (const float4*)d->pd_bboxes;
    a.anc = (const float2*)d->anc_points;
    a.gt_labels = d->gt_labels;
    a.gt_bboxes = (const float4*)d->gt_bboxes;
    a.mask_gt = d->mask_gt;
    a.B = d->B;
    a.A = d->A;
    a.C = d->C;
    a.G = d->G;
    a.topk = d->topk;
    a.alpha = d->alpha;
    a.beta = d->beta;
    a.eps = d->eps;
    a.target_labels = (long long*)d->target_labels;
    a.target_bboxes = (float4*)d->target_bboxes;
    a.target_scores = d->target_scores;
    a.fg_mask = d->fg_mask;
    const size_t nba = (size_t)d->B * d->A;
    if (d->G == 0) {
        hipLaunchKernelGGL(tal_empty_kernel, dim3(1024), dim3(256), 0, s, a);
        Y6_LAUNCH_CHECK();
        Y6_HIP(hipMemsetAsync(d->target_scores, 0, nba * d->C * sizeof(float), s));
        return Y6_OK;
    }
    Y6_REQUIRE(d->gt_labels && d->gt_bboxes && d->mask_gt && d->workspace, "tal_assign: null gt / workspace");
    Y6_REQUIRE(d->workspace_bytes >= y6_tal_workspace_bytes(d->B, d->A, d->G), "tal_assign: workspace too small");
    Y6_REQUIRE((size_t)d->A * 4 <= 160 * 1024 - 64, "tal_assign: A=%d does not fit the LDS metric row", d->A);
    char* ws = (char*)d->workspace;
    const size_t ba = al256(nba * 4), bgn = al256((size_t)d->B * d->G * 4);
    a.fg_cnt = (int*)ws;
    a.first_g = (int*)(ws + ba);
    a.assign = (int*)(ws + 2 * ba);
    a.norm = (float*)(ws + 3 * ba);
    a.pos_am = (unsigned*)(ws + 4 * ba);
    a.pos_ov = (unsigned*)(ws + 4 * ba + bgn);
    Y6_HIP(hipMemsetAsync(a.fg_cnt, 0, nba * 4, s));
    Y6_HIP(hipMemsetAsync(a.first_g, 0x7f, nba * 4, s));
    Y6_HIP(hipMemsetAsync(a.pos_am, 0, 2 * bgn, s));
    static const bool reg_rows = !(getenv("Y6_TAL_TOPK_REG") && atoi(getenv("Y6_TAL_TOPK_REG")) == 0);   // A/B switch
    if (reg_rows && d->A <= 256 * 34) {
        hipLaunchKernelGGL(tal_topk_reg_kernel<34>, dim3(d->B * d->G), dim3(256), 0, s, a);
    } else {
        const size_t lds = (size_t)d->A * 4;
        if (lds > 64 * 1024)
            Y6_HIP(hipFuncSetAttribute((const void*)tal_topk_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(tal_topk_lds_kernel, dim3(d->B * d->G), dim3(256), lds, s, a);
    }
    Y6_LAUNCH_CHECK();
    const unsigned nb = (unsigned)((nba + 255) / 256);
    hipLaunchKernelGGL(tal_resolve_kernel, dim3(nb), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    hipLaunchKernelGGL(tal_targets_kernel, dim3(nb), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    size_t g = (nba * d->C + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    hipLaunchKernelGGL(tal_scores_kernel, dim3((unsigned)g), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}

extern "C" size_t y6_atss_workspace_bytes(int B, int A, int G) { return y6_tal_workspace_bytes(B, A, G); }

extern "C" int y6_atss_assign(const y6_atss_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d && d->anc_bboxes && d->target_labels && d->target_bboxes && d->target_scores && d->fg_mask,
               "atss_assign: null argument");
    Y6_REQUIRE(d->B > 0 && d->A > 0 && d->C > 0 && d->G >= 0 && d->topk > 0, "atss_assign: bad sizes");
    Y6_REQUIRE(d->n_levels >= 1 && d->n_levels <= Y6_MAX_LEVELS, "atss_assign: 1..%d levels", Y6_MAX_LEVELS);
    Y6_REQUIRE(d->n_levels * d->topk <= ATSS_MAX_CAND, "atss_assign: levels x topk must be <= %d", ATSS_MAX_CAND);
    hipStream_t s = (hipStream_t)stream;
    AtssArgs a;
    memset(&a, 0, sizeof(a));
    a.anc = (const float4*)d->anc_bboxes;
    a.gt_labels = d->gt_labels;
    a.gt_bboxes = (const float4*)d->gt_bboxes;
    a.mask_gt = d->mask_gt;
    a.pd_bboxes = (const float4*)d->pd_bboxes;
    a.B = d->B;
    a.A = d->A;
    a.C = d->C;
    a.G = d->G;
    a.topk = d->topk;
    a.n_levels = d->n_levels;
    int acc = 0;
    for (int l = 0; l < d->n_levels; ++l) {
        a.lvl_start[l] = acc;
        acc += d->n_level_bboxes[l];
    }
    a.lvl_start[d->n_levels] = acc;
    Y6_REQUIRE(acc == d->A, "atss_assign: n_level_bboxes sum %d != A %d", acc, d->A);
    a.target_labels = (long long*)d->target_labels;
    a.target_bboxes = (float4*)d->target_bboxes;
    a.target_scores = d->target_scores;
    a.fg_mask = d->fg_mask;
    const size_t nba = (size_t)d->B * d->A;
    if (d->G == 0) {   // early-out :48-53
        hipLaunchKernelGGL(atss_empty_kernel, dim3(1024), dim3(256), 0, s, a);
        Y6_LAUNCH_CHECK();
        Y6_HIP(hipMemsetAsync(d->target_scores, 0, nba * d->C * sizeof(float), s));
        return Y6_OK;
    }
    Y6_REQUIRE(d->gt_labels && d->gt_bboxes && d->mask_gt && d->workspace, "atss_assign: null gt / workspace");
    Y6_REQUIRE(d->workspace_bytes >= y6_atss_workspace_bytes(d->B, d->A, d->G), "atss_assign: workspace too small");
    Y6_REQUIRE((size_t)d->A * 4 <= 160 * 1024 - 2048, "atss_assign: A=%d does not fit the LDS distance row", d->A);
    char* ws = (char*)d->workspace;
    const size_t ba = al256(nba * 4);
    a.fg_cnt = (int*)ws;
    a.first_g = (int*)(ws + ba);
    a.assign = (int*)(ws + 2 * ba);
    a.norm = (float*)(ws + 3 * ba);
    Y6_HIP(hipMemsetAsync(a.fg_cnt, 0, nba * 4, s));
    Y6_HIP(hipMemsetAsync(a.first_g, 0x7f, nba * 4, s));
    const size_t lds = (size_t)d->A * 4;
    if (lds > 60 * 1024)
        Y6_HIP(hipFuncSetAttribute((const void*)atss_candidates_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(atss_candidates_kernel, dim3(d->B * d->G), dim3(256), lds, s, a);
    Y6_LAUNCH_CHECK();
    const unsigned nb = (unsigned)((nba + 255) / 256);
    hipLaunchKernelGGL(atss_targets_kernel, dim3(nb), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    size_t g = (nba * d->C + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    hipLaunchKernelGGL(atss_scores_kernel, dim3((unsigned)g), dim3(256), 0, s, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
