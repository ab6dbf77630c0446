// nms_cand.hpp - candidate selection of non_max_suppression (reference yolov6/utils/nms.py:48, :69-84) on prediction rows that
// already sit in LDS.  Shared by nms_candidates_kernel (nms.hip: rows fetched from the [B, A, 5 + nc] prediction tensor) and
// head_pred_decode_kernel (head_decode.hip: rows it has just computed - the candidates leave with the decode launch and the 91 MB
// tensor is not read again, y6_nms_sink in include/yolov6_hip.h).  One translation unit's arithmetic, bit for bit, in both.
#pragma once
#include "common.hpp"

namespace y6cand {

typedef unsigned long long u64;
// The per-image lengths of the key lists sit one cache line apart (ints): every block of an image adds to its image's counter,
// and with the 32 counters of a batch in ONE 128-byte line all 4 200 adds of a 640^2 b32 batch queued at one L2 channel, ~12 ns
// each: 50 us - that queue, not the selection and not the fetch, was y6_nms's first stage (r04q).
constexpr int kCountStride = 32;
constexpr int kCandRows = 64;    // most rows of one emit_candidates() call (one lane of wave 0 per row)

__device__ __forceinline__ bool class_ok(int j, const int* classes, int n_classes) {
    if (!classes) return true;
    for (int k = 0; k < n_classes; ++k)
        if (classes[k] == j) return true;
    return false;
}

typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;

// What a thread carries from the selection to the publication of its candidates.
struct CandSel {
    u64 passm;   // multi-label: bit k <=> this thread's k-th element is a candidate; best-class: class + 1 of row `tid`
    int off;     // block-local slot of the thread's first candidate
};

// rows: [nrows][no] fp32 in LDS (xywh, obj, nc class scores), complete and visible to the block; rowflag: nrows ints of LDS scratch;
// s_cnt: an int of LDS.  Called by ALL 256 threads of the block; nrows <= kCandRows and nrows * nc <= 64 * 256 (a 64-bit pass mask
// per thread).  LDS work only (the pointers carry the LDS address space: through generic pointers these reads are FLAT
// instructions, whose results wait - in order - behind every global store the wave has in flight).
//   row flag (nms.py:48): obj > conf AND max cls > conf;  multi-label (:75-77): every class with cls * obj > conf;  else (:79-80):
//   the best class (first maximum) if its cls * obj > conf;  both filtered by `classes` (:83-84).
// Thread t looks at the class scores t, t + 256, ... of the block's row image in (row, class) order: consecutive lanes read
// consecutive LDS words.  Pass 1 raises the row flags, pass 2 finds the candidates and reserves block-local slots with one LDS
// atomic per thread.  On return (behind a barrier) *s_cnt is the block's candidate count.
__device__ __forceinline__ CandSel cand_select_general(const lds_f32* rows, lds_i32* rowflag, int nrows, int no, int nc, float conf_thres,
                                               const int* __restrict__ classes, int n_classes, int multi_label, lds_i32* s_cnt) {
    const int tid = threadIdx.x;
    if (tid == 0) *s_cnt = 0;
    for (int r = tid; r < nrows; r += 256) rowflag[r] = 0;
    __syncthreads();
    const int nel = nrows * nc;
    const int r_step = 256 / nc, j_step = 256 - r_step * nc;   // element t + 256 is (r + r_step, j + j_step), carried into the row
    const int r_first = tid / nc, j_first = tid - r_first * nc;
    {
        int r = r_first, j = j_first;
        for (int e = tid; e < nel; e += 256) {
            const lds_f32* row = rows + r * no;
            if (row[5 + j] > conf_thres && row[4] > conf_thres) rowflag[r] = 1;   // benign race: every writer stores 1
            r += r_step;
            j += j_step;
            if (j >= nc) {
                j -= nc;
                ++r;
            }
        }
    }
    __syncthreads();
    CandSel cs;
    cs.passm = 0ull;
    cs.off = 0;
    if (multi_label) {
        int r = r_first, j = j_first, k = 0;
        for (int e = tid; e < nel; e += 256, ++k) {
            const lds_f32* row = rows + r * no;
            if (rowflag[r] && (row[5 + j] * row[4] > conf_thres) && class_ok(j, classes, n_classes)) cs.passm |= 1ull << k;
            r += r_step;
            j += j_step;
            if (j >= nc) {
                j -= nc;
                ++r;
            }
        }
        if (cs.passm) cs.off = __hip_atomic_fetch_add(s_cnt, __popcll(cs.passm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (tid < nrows && rowflag[tid]) {
        // best class per row: max conf, first (lowest) class index on ties (torch.max semantics, nms.py:79); a thread per row
        // (lanes a row apart: `no` words, conflict-free for odd `no`)
        const lds_f32* row = rows + tid * no;
        const float obj = row[4];
        float bc = -INFINITY;
        int bj = 0;
        for (int j = 0; j < nc; ++j) {
            const float c = row[5 + j] * obj;
            if (c > bc) {
                bc = c;
                bj = j;
            }
        }
        if (bc > conf_thres && class_ok(bj, classes, n_classes)) {
            cs.passm = (u64)bj + 1ull;
            cs.off = __hip_atomic_fetch_add(s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    return cs;
}

// The keys of this thread's candidates into the block's slice kb = (image's key list) + (the block's base, reserved by ONE
// atomicAdd(counts_b, *s_cnt) of one thread).  a0: index (inside the image) of row 0.  A key is (conf bits << 32 | ~flat),
// flat = anchor * nc + cls: a descending key sort is the reference's "score descending, earlier row first" order; the order of the
// list itself does not matter.
__device__ __forceinline__ void cand_publish_general(const lds_f32* rows, int nrows, int no, int nc, int a0, int multi_label, CandSel cs,
                                             u64* __restrict__ kb) {
    const int tid = threadIdx.x;
    if (multi_label) {
        const int nel = nrows * nc;
        const int r_step = 256 / nc, j_step = 256 - r_step * nc;
        int r = tid / nc, j = tid - r * nc;
        int off = cs.off;
        u64 passm = cs.passm;
        for (int e = tid; e < nel && passm; e += 256) {
            if (passm & 1ull) {
                const lds_f32* row = rows + r * no;
                const unsigned flat = (unsigned)(a0 + r) * (unsigned)nc + (unsigned)j;
                kb[off++] = ((u64)__float_as_uint(row[5 + j] * row[4]) << 32) | (u64)(0xFFFFFFFFu - flat);
            }
            passm >>= 1;
            r += r_step;
            j += j_step;
            if (j >= nc) {
                j -= nc;
                ++r;
            }
        }
    } else if (cs.passm) {
        const int bj = (int)cs.passm - 1;
        const lds_f32* row = rows + tid * no;
        const unsigned flat = (unsigned)(a0 + tid) * (unsigned)nc + (unsigned)bj;
        kb[cs.off] = ((u64)__float_as_uint(row[5 + bj] * row[4]) << 32) | (u64)(0xFFFFFFFFu - flat);
    }
}

// ---- the fast form of the multi-label selection ------------------------------------------------------------------------------
// The general form above costs ~150 VALU instructions per class score (two passes, per-element row / class bookkeeping): 21.5 M
// scores of a 640^2 b32 batch = 55 us of the whole chip's vector units - that, not the 91 MB fetch, is what the first stage of
// y6_nms was made of (r04q).  When every objectness of the block is <= 1 (a sigmoid's always is), cls * obj > conf implies
// cls > conf, so the row flag of nms.py:48 is implied by the candidate itself and ONE scan with a quick reject suffices: the row
// image is read as 16-byte pieces in storage order, a piece whose four values are all <= conf (98 % of them) costs its compares;
// only a value above conf is located (row, column), columns 0-3 (the box) dropped, and tested as the reference does
// (obj > conf, cls * obj > conf, class filter).  A block that meets an objectness above 1 (or NaN) raises *s_gen and the caller
// repeats the selection in the general form: same result for every input.
struct CandSelFast {
    u64 passm;   // bit 4 * i + k <=> element 4 * (tid + 256 * i) + k of the row image is a candidate
    int off;
};

__device__ __forceinline__ CandSelFast cand_select_fast(const lds_f32* rows, int nrows, int no, int nc, float conf_thres,
                                                        const int* __restrict__ classes, int n_classes, lds_i32* s_cnt, lds_i32* s_gen) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    if (tid == 0) {
        *s_cnt = 0;
        *s_gen = 0;
    }
    __syncthreads();
    const int nfl = nrows * no;
    // piece q covers elements 4 q .. 4 q + 3; the thread's next piece is 1024 elements on: (row, column) advance by (1024 / no, 1024 % no)
    const int r_step = 1024 / no, c_step = 1024 - r_step * no;
    int r0 = (4 * tid) / no, c0 = 4 * tid - r0 * no;
    CandSelFast cs;
    cs.passm = 0ull;
    cs.off = 0;
    bool general = false;
    int it = 0;
    for (int e0 = 4 * tid; e0 < nfl; e0 += 1024, ++it) {
        const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(rows + e0);
        // objectness column inside this piece?  (column 4 of row r0 is element k = 4 - c0; of row r0 + 1: k = no + 4 - c0)
        {
            const int k4 = c0 <= 4 ? 4 - c0 : no + 4 - c0;
            if (k4 < 4 && e0 + k4 < nfl && !(v[k4] <= 1.0f)) general = true;
        }
        const bool any = (v[0] > conf_thres) || (v[1] > conf_thres) || (v[2] > conf_thres) || (v[3] > conf_thres);
        if (any) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(v[k] > conf_thres) || e0 + k >= nfl) continue;
                int r = r0, c = c0 + k;
                if (c >= no) {
                    c -= no;
                    ++r;
                }
                if (c < 5) continue;   // box / objectness columns
                const float obj = rows[r * no + 4];
                if (obj > conf_thres && v[k] * obj > conf_thres && class_ok(c - 5, classes, n_classes)) cs.passm |= 1ull << (4 * it + k);
            }
        }
        r0 += r_step;
        c0 += c_step;
        if (c0 >= no) {
            c0 -= no;
            ++r0;
        }
    }
    if (general) *s_gen = 1;   // benign race: every writer stores 1
    if (cs.passm) cs.off = __hip_atomic_fetch_add(s_cnt, __popcll(cs.passm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    return cs;
}

__device__ __forceinline__ void cand_publish_fast(const lds_f32* rows, int no, int nc, int a0, CandSelFast cs, u64* __restrict__ kb) {
    const int tid = threadIdx.x;
    int off = cs.off;
    u64 m = cs.passm;
    while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const int e = 4 * (tid + 256 * (bit >> 2)) + (bit & 3);
        const int r = e / no, c = e - r * no;
        const unsigned flat = (unsigned)(a0 + r) * (unsigned)nc + (unsigned)(c - 5);
        kb[off++] = ((u64)__float_as_uint(rows[e] * rows[r * no + 4]) << 32) | (u64)(0xFFFFFFFFu - flat);
    }
}

// ---- what the two kernels call ------------------------------------------------------------------------------------------------
struct CandSel2 {
    CandSel g;
    CandSelFast f;
    int fast;    // block-uniform
};

// s_cnt, s_gen: two ints of LDS; on return *s_cnt is the block's candidate count
__device__ __forceinline__ CandSel2 cand_select(const lds_f32* rows, lds_i32* rowflag, int nrows, int no, int nc, float conf_thres,
                                                const int* __restrict__ classes, int n_classes, int multi_label, lds_i32* s_cnt,
                                                lds_i32* s_gen) {
    CandSel2 cs;
    cs.g.passm = 0ull;
    cs.g.off = 0;
    cs.f.passm = 0ull;
    cs.f.off = 0;
    cs.fast = 0;
    if (multi_label && nrows * no <= 16 * 1024) {
        cs.f = cand_select_fast(rows, nrows, no, nc, conf_thres, classes, n_classes, s_cnt, s_gen);
        cs.fast = *s_gen == 0;
        if (cs.fast) return cs;
        __syncthreads();   // everybody has read *s_gen before the general form resets the counters
    }
    cs.g = cand_select_general(rows, rowflag, nrows, no, nc, conf_thres, classes, n_classes, multi_label, s_cnt);
    return cs;
}

__device__ __forceinline__ void cand_publish(const lds_f32* rows, int nrows, int no, int nc, int a0, int multi_label, const CandSel2& cs,
                                             u64* __restrict__ kb) {
    if (cs.fast)
        cand_publish_fast(rows, no, nc, a0, cs.f, kb);
    else
        cand_publish_general(rows, nrows, no, nc, a0, multi_label, cs.g, kb);
}

}  // namespace y6cand
