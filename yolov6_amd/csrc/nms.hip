// nms.hip — batched, class-aware NMS on the device.
// Restates non_max_suppression (reference yolov6/utils/nms.py:31-105) and the
// torchvision.ops.nms it calls at :96 (torchvision is an un-vendored dependency; its
// published algorithm: stable sort by score descending, greedy sweep, suppress j when
// IoU(i,j) > thr with IoU = inter / (area_i + area_j - inter), fp32, no +1, no eps).
//
// Stage 1 (nms_candidates_kernel): one wave per anchor row; rows that pass
//   obj > conf AND max(cls) > conf (nms.py:48) emit (conf = cls*obj (:69), class) candidates
//   (multi-label: every class with conf > thr (:75-77); else best class (:79-80)), filtered
//   by `classes` (:83-84), appended with one atomic per wave.  A candidate is a 64-bit key
//   (conf bits << 32 | ~flat) with flat = anchor*nc + cls, so a descending key sort IS the
//   reference's "score descending, earlier row first" order.
// Stage 2 (nms_sort_sweep_kernel): one 1024-thread block per image: bitonic sort of the keys
//   (LDS when they fit, workspace otherwise), cap to max_nms (:90-91), then a windowed greedy
//   sweep over the sorted list: xyxy boxes (:72, xywh2xyxy :21-28) offset by cls*max_wh (:94-95)
//   are built 2048 at a time in LDS, tested against the boxes already kept, and swept with a wave
//   ballot over the alive-bitmask, until max_det (:97-98) boxes are kept.
// Compile with -ffp-contract=off: index parity needs the reference's unfused fp32 arithmetic.
#include "common.hpp"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ bool class_ok(int j, const int* classes, int n_classes) {
    if (!classes) return true;
    for (int k = 0; k < n_classes; ++k)
        if (classes[k] == j) return true;
    return false;
}

// One block = `rpb` consecutive anchor rows of ONE image (blockIdx interleaves images so that
// concurrently running blocks append to different per-image counters).  Candidates are staged in
// LDS (one LDS atomic per wave-row), then the block reserves its slice of the image's key list
// with a single global atomic and copies the keys out coalesced.  (v1 did one global atomic per
// row: all resident waves hit the same counter and serialised at ~12 ns each - 4.3 ms per call.)
__global__ __launch_bounds__(256) void nms_candidates_kernel(const float* __restrict__ pred, int B, int A, int nc,
                                                             float conf_thres, const int* __restrict__ classes,
                                                             int n_classes, int multi_label, int rpb,
                                                             u64* __restrict__ keys, size_t cap,
                                                             int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* stage = reinterpret_cast<u64*>(smem);                         // [rpb * (multi_label ? nc : 1)]
    __shared__ int s_cnt, s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x % B, chunk = blockIdx.x / B;
    const int no = nc + 5;
    // the block's rows are one contiguous run of the prediction tensor: fetch it with 16-byte loads, all in
    // flight at once, and work from LDS (v2 walked the rows with dependent 4-byte loads: latency bound)
    float* rows = reinterpret_cast<float*>(smem + (size_t)rpb * (multi_label ? nc : 1) * sizeof(u64));
    const int a0 = chunk * rpb;
    const int nrows = min(rpb, A - a0);
    {
        const size_t gbase = ((size_t)b * A + a0) * no;
        const int nfl = nrows * no;
        const float* src = pred + gbase;
        if (((gbase & 3) == 0) && ((reinterpret_cast<uintptr_t>(pred) & 15) == 0)) {
            const int nv = nfl >> 2;
            for (int i = tid; i < nv; i += 256) reinterpret_cast<float4*>(rows)[i] = reinterpret_cast<const float4*>(src)[i];
            for (int i = (nv << 2) + tid; i < nfl; i += 256) rows[i] = src[i];
        } else {
            for (int i = tid; i < nfl; i += 256) rows[i] = src[i];
        }
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int r = wave; r < nrows; r += 4) {
        const int an = a0 + r;
        const float* row = rows + r * no;
        const float obj = row[4];
        if (!(obj > conf_thres)) continue;
        float mx = -INFINITY;
        for (int j = lane; j < nc; j += 64) mx = fmaxf(mx, row[5 + j]);
        mx = wave_max(mx);
        if (!(mx > conf_thres)) continue;
        if (multi_label) {
            for (int j0 = 0; j0 < nc; j0 += 64) {
                const int j = j0 + lane;
                float c = 0.f;
                bool pass = false;
                if (j < nc) {
                    c = row[5 + j] * obj;
                    pass = (c > conf_thres) && class_ok(j, classes, n_classes);
                }
                const u64 bal = __ballot(pass);
                if (bal == 0) continue;
                const int n = __popcll(bal);
                const int leader = __ffsll((long long)bal) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&s_cnt, n);
                base = __shfl(base, leader, 64);
                if (pass) {
                    const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                    const unsigned flat = (unsigned)an * (unsigned)nc + (unsigned)j;
                    stage[pos] = ((u64)__float_as_uint(c) << 32) | (u64)(0xFFFFFFFFu - flat);
                }
            }
        } else {
            // best class: max conf, first (lowest) class index on ties (torch.max semantics)
            float bc = -INFINITY;
            int bj = 0x7fffffff;
            for (int j = lane; j < nc; j += 64) {
                const float c = row[5 + j] * obj;
                if (c > bc) {
                    bc = c;
                    bj = j;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float oc = __shfl_xor(bc, o, 64);
                const int oj = __shfl_xor(bj, o, 64);
                if (oc > bc || (oc == bc && oj < bj)) {
                    bc = oc;
                    bj = oj;
                }
            }
            if (lane == 0 && bc > conf_thres && class_ok(bj, classes, n_classes)) {
                const int pos = atomicAdd(&s_cnt, 1);
                const unsigned flat = (unsigned)an * (unsigned)nc + (unsigned)bj;
                stage[pos] = ((u64)__float_as_uint(bc) << 32) | (u64)(0xFFFFFFFFu - flat);
            }
        }
    }
    __syncthreads();
    const int total = s_cnt;
    if (total == 0) return;
    if (tid == 0) s_base = atomicAdd(&counts[b], total);
    __syncthreads();
    u64* kb = keys + (size_t)b * cap + s_base;
    for (int i = tid; i < total; i += 256) kb[i] = stage[i];
}

// descending bitonic sort of P (power of two) keys by the whole block; one compare-exchange pair per
// thread-iteration (no idle half).  Force-inlined so that a pointer derived from the LDS array keeps
// its address space (ds_read/ds_write instead of flat accesses).
__device__ __forceinline__ void bitonic_desc(u64* keys, int P) {
    const int T = blockDim.x;
    const int half = P >> 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < half; t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const u64 x = keys[i], y = keys[l];
                const bool desc = (i & k) == 0;
                if (desc ? (x < y) : (x > y)) {
                    keys[i] = y;
                    keys[l] = x;
                }
            }
            __syncthreads();
        }
    }
}

// LDS variant for a 1024-thread block.  Wave w owns the chunk [w*C, (w+1)*C) (C = P/16, at least 128):
// every compare-exchange stage with distance j < C stays inside one chunk, so those stages - 95 of the
// 105 for 16384 keys - run with wave-level ordering only; block barriers remain for j >= C.
__device__ __forceinline__ void bitonic_desc_lds(u64* keys, int P) {
    const int T = blockDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int C = P >> 4;
    if (C < 128) C = 128;
    if (C > P) C = P;
    const int nchunk = P / C;
    const int half = P >> 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < C) {
                if (wave < nchunk) {
                    u64* base = keys + wave * C;
                    for (int jj = j; jj > 0; jj >>= 1) {
                        for (int u = lane; u < (C >> 1); u += 64) {
                            const int i = ((u & ~(jj - 1)) << 1) | (u & (jj - 1));
                            const int l = i | jj;
                            const u64 x = base[i], y = base[l];
                            const bool desc = ((wave * C + i) & k) == 0;
                            if (desc ? (x < y) : (x > y)) {
                                base[i] = y;
                                base[l] = x;
                            }
                        }
                        // same-wave LDS accesses complete in issue order; keep the compiler from reordering them
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                break;   // all remaining distances of this k were handled above
            }
            for (int t = threadIdx.x; t < half; t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const u64 x = keys[i], y = keys[l];
                const bool desc = (i & k) == 0;
                if (desc ? (x < y) : (x > y)) {
                    keys[i] = y;
                    keys[l] = x;
                }
            }
            __syncthreads();
        }
        if (k >= C) __syncthreads();   // the next k starts with a cross-chunk distance (or the sort is over)
    }
    __syncthreads();
}

constexpr int kLdsKeys = 16384;   // 128 KiB of 64-bit keys during the sort
constexpr int kWin = 2048;        // sweep window: sorted candidates resident in LDS at a time
constexpr int kKeptCap = 2048;    // kept boxes resident in LDS (max_det limit)
constexpr size_t kSweepLds = (size_t)(kWin + kKeptCap) * 16 + (kWin / 64) * 8;

__device__ __forceinline__ float nms_iou(const float4 bi, const float4 bj) {
    // torchvision devIoU / cpu kernel arithmetic, fp32, unfused
    const float left = fmaxf(bi.x, bj.x), right = fminf(bi.z, bj.z);
    const float top = fmaxf(bi.y, bj.y), bottom = fminf(bi.w, bj.w);
    const float iw = fmaxf(right - left, 0.f), ih = fmaxf(bottom - top, 0.f);
    const float inter = iw * ih;
    const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
    const float area_j = (bj.z - bj.x) * (bj.w - bj.y);
    return inter / (area_i + area_j - inter);
}

// One 1024-thread block per image.
//  1. bitonic sort of the candidate keys (LDS when <= 16384, else in the global workspace);
//  2. windowed greedy sweep: 2048 sorted candidates at a time become class-offset boxes in LDS, are
//     first tested against the boxes kept so far, then swept greedily: a wave ballot over the alive
//     bitmask finds the next survivor, every later box of the window is tested against it in parallel.
//     Boxes are only ever built for the windows the sweep reaches before max_det boxes are kept.
__global__ __launch_bounds__(1024) void nms_sort_sweep_kernel(const float* __restrict__ pred, int A, int nc,
                                                              float iou_thres, int agnostic, int max_det, int max_nms,
                                                              float max_wh, u64* __restrict__ keys, size_t cap,
                                                              const int* __restrict__ counts,
                                                              float* __restrict__ out_dets, int* __restrict__ out_index,
                                                              int* __restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_batch[64];
    __shared__ int s_nb, s_k0, s_kept, s_last;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    int n = counts[b];
    if (n <= 0) {
        if (tid == 0) out_count[b] = 0;
        return;
    }
    u64* gk = keys + (size_t)b * cap;
    int P = 1;
    while (P < n) P <<= 1;
    if (P <= kLdsKeys) {
        u64* lk = reinterpret_cast<u64*>(smem);
        for (int i = tid; i < P; i += T) lk[i] = i < n ? gk[i] : 0ull;
        __syncthreads();
        bitonic_desc_lds(lk, P);
        for (int i = tid; i < n; i += T) gk[i] = lk[i];   // sorted keys back to global: the LDS is re-used below
    } else {
        for (int i = n + tid; i < P; i += T) gk[i] = 0ull;  // cap is a power of two >= P
        __syncthreads();
        bitonic_desc(gk, P);
    }
    if (n > max_nms) n = max_nms;
    __syncthreads();

    float4* wbox = reinterpret_cast<float4*>(smem);
    float4* kbox = wbox + kWin;
    u64* walive = reinterpret_cast<u64*>(kbox + kKeptCap);
    int* kept_pos = out_index + (size_t)b * max_det;   // sorted positions first; converted to flat ids at the end
    const int no = nc + 5;
    int kept = 0;
    for (int pos = 0; pos < n && kept < max_det; pos += kWin) {
        const int wn = (n - pos) < kWin ? (n - pos) : kWin;
        // window boxes: xywh2xyxy (nms.py:72) + class offset (nms.py:94-95)
        for (int t = tid; t < wn; t += T) {
            const u64 key = gk[pos + t];
            const unsigned flat = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
            const unsigned an = flat / (unsigned)nc, cls = flat - an * (unsigned)nc;
            const float* row = pred + ((size_t)b * A + an) * no;
            const float x = row[0], y = row[1], w = row[2], h = row[3];
            const float off = agnostic ? 0.f : (float)cls * max_wh;
            float4 q;
            q.x = (x - w / 2.f) + off;
            q.y = (y - h / 2.f) + off;
            q.z = (x + w / 2.f) + off;
            q.w = (y + h / 2.f) + off;
            wbox[t] = q;
        }
        __syncthreads();
        // survivors of the boxes kept in earlier windows
#pragma unroll
        for (int r = 0; r < kWin / 1024; ++r) {
            const int t = tid + r * 1024;
            bool alive = t < wn;
            if (alive) {
                const float4 bj = wbox[t];
                for (int k = 0; k < kept; ++k)
                    if (nms_iou(kbox[k], bj) > iou_thres) {
                        alive = false;
                        break;
                    }
            }
            const u64 bal = __ballot(alive);
            if (lane == 0) walive[r * 16 + wave] = bal;   // bit (t & 63) of word (t >> 6)
        }
        __syncthreads();
        // Greedy sweep inside the window, 64 undecided candidates at a time.  Wave 0 takes the first (up to)
        // 64 alive candidates, settles their mutual suppression with lane broadcasts - no block barrier per
        // kept box - and publishes the boxes it kept; then every thread tests the window's later alive
        // candidates against just those new boxes.  Exactly the sequential greedy order: a candidate left
        // outside a batch is alive only if it lies behind every member of that batch.
        int cur = 0;
        while (kept < max_det) {
            if (wave == 0) {
                u64 w = 0ull;
                if (lane < kWin / 64) {
                    w = walive[lane];
                    if (lane < (cur >> 6)) w = 0ull;
                    if (lane == (cur >> 6)) w &= ~((1ull << (cur & 63)) - 1ull);
                }
                const int pc = __popcll(w);
                int incl = pc;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += v;
                }
                const int total = __shfl(incl, 63, 64);
                int p = incl - pc;
                u64 ww = w;
                while (ww != 0ull && p < 64) {
                    s_batch[p++] = (lane << 6) + (__ffsll((long long)ww) - 1);
                    ww &= ww - 1ull;
                }
                if (lane == 0) s_nb = total < 64 ? total : 64;
            }
            __syncthreads();
            const int nb = s_nb;
            if (nb == 0) break;
            if (wave == 0) {
                const int idx = lane < nb ? s_batch[lane] : 0;
                const float4 bx = wbox[idx];
                bool alive = lane < nb;
                u64 keepmask = 0ull;
                int nk = 0;
                for (int i = 0; i < nb; ++i) {
                    const u64 am = __ballot(alive);
                    if (!((am >> i) & 1ull)) continue;
                    if (kept + nk >= max_det) break;
                    keepmask |= 1ull << i;
                    ++nk;
                    float4 bi;
                    bi.x = __shfl(bx.x, i, 64);
                    bi.y = __shfl(bx.y, i, 64);
                    bi.z = __shfl(bx.z, i, 64);
                    bi.w = __shfl(bx.w, i, 64);
                    if (lane > i && alive && nms_iou(bi, bx) > iou_thres) alive = false;
                }
                if ((keepmask >> lane) & 1ull) {
                    const int kp = kept + __popcll(keepmask & ((1ull << lane) - 1ull));
                    kbox[kp] = bx;
                    kept_pos[kp] = pos + idx;
                }
                if (lane < nb) atomicAnd(&walive[idx >> 6], ~(1ull << (idx & 63)));   // decided either way
                if (lane == 0) {
                    s_k0 = kept;
                    s_kept = kept + nk;
                    s_last = s_batch[nb - 1];
                }
            }
            __syncthreads();
            const int k0 = s_k0, last = s_last;
            kept = s_kept;
            if (kept >= max_det) break;
#pragma unroll
            for (int r = 0; r < kWin / 1024; ++r) {
                const int t = tid + r * 1024;
                if (t > last && t < wn && ((walive[t >> 6] >> (t & 63)) & 1ull)) {
                    const float4 bj = wbox[t];
                    for (int k = k0; k < kept; ++k)
                        if (nms_iou(kbox[k], bj) > iou_thres) {
                            atomicAnd(&walive[t >> 6], ~(1ull << (t & 63)));
                            break;
                        }
                }
            }
            __syncthreads();
            cur = last + 1;
            if (cur >= wn) break;
        }
        __syncthreads();
    }
    __syncthreads();
    // detections: (xyxy without the class offset, conf, cls)  nms.py:98 `x[keep_box_idx]`
    for (int k = tid; k < kept; k += T) {
        const u64 key = gk[kept_pos[k]];
        const unsigned flat = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
        const unsigned an = flat / (unsigned)nc, cls = flat - an * (unsigned)nc;
        const float* row = pred + ((size_t)b * A + an) * no;
        const float x = row[0], y = row[1], w = row[2], h = row[3];
        float* o = out_dets + ((size_t)b * max_det + k) * 6;
        o[0] = x - w / 2.f;
        o[1] = y - h / 2.f;
        o[2] = x + w / 2.f;
        o[3] = y + h / 2.f;
        o[4] = __uint_as_float((unsigned)(key >> 32));
        o[5] = (float)cls;
        kept_pos[k] = (int)flat;
    }
    if (tid == 0) out_count[b] = kept;
}

inline size_t next_pow2(size_t v) {
    size_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct NmsWs {
    size_t cap;       // keys per image (power of two)
    size_t off_keys, off_boxes, off_counts, total;
};
NmsWs nms_ws_layout(int B, int A, int nc, int multi_label, int max_nms) {
    NmsWs w;
    const bool ml = multi_label && nc > 1;
    w.cap = next_pow2(ml ? (size_t)A * nc : (size_t)A);
    w.off_keys = 0;
    w.off_boxes = align256(w.off_keys + (size_t)B * w.cap * sizeof(u64));
    w.off_counts = align256(w.off_boxes + (size_t)B * max_nms * sizeof(float4));
    w.total = align256(w.off_counts + (size_t)B * sizeof(int));
    return w;
}

}  // namespace

extern "C" size_t y6_nms_workspace_bytes(int B, int A, int nc, int multi_label) {
    return nms_ws_layout(B, A, nc, multi_label, 30000).total;
}

extern "C" int y6_nms(const y6_nms_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d && d->pred && d->out_dets && d->out_index && d->out_count && d->workspace, "nms: null argument");
    Y6_REQUIRE(d->conf_thres >= 0.f && d->conf_thres <= 1.f, "conf_thresh must be in 0.0 to 1.0, however %g is provided.",
               (double)d->conf_thres);
    Y6_REQUIRE(d->iou_thres >= 0.f && d->iou_thres <= 1.f, "iou_thres must be in 0.0 to 1.0, however %g is provided.",
               (double)d->iou_thres);
    Y6_REQUIRE(d->B > 0 && d->A > 0 && d->nc > 0 && d->max_det > 0, "nms: bad sizes");
    Y6_REQUIRE(d->max_nms > 0 && d->max_nms <= 30000, "nms: max_nms must be in (0, 30000]");
    Y6_REQUIRE(d->max_det <= kKeptCap, "nms: max_det %d exceeds the %d kept boxes held in LDS", d->max_det, kKeptCap);
    Y6_REQUIRE((size_t)d->A * d->nc < 0xFFFFFFFFull, "nms: A*nc overflows the 32-bit candidate id");
    const int ml = d->multi_label && d->nc > 1;
    const NmsWs w = nms_ws_layout(d->B, d->A, d->nc, ml, 30000);
    Y6_REQUIRE(d->workspace_bytes >= w.total, "nms: workspace too small (%zu < %zu)", d->workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d->workspace;
    u64* keys = (u64*)(ws + w.off_keys);
    int* counts = (int*)(ws + w.off_counts);
    Y6_HIP(hipMemsetAsync(counts, 0, (size_t)d->B * sizeof(int), s));
    // rows per block: as many as fit a 48 KiB staging area (64 rows x 80 classes = 40 KiB)
    const int per_row = ml ? d->nc : 1;
    int rpb = 6144 / per_row;
    rpb = rpb > 64 ? 64 : (rpb < 4 ? 4 : (rpb / 4) * 4);
    const size_t stage_bytes = (size_t)rpb * per_row * sizeof(u64) + (size_t)rpb * (d->nc + 5) * sizeof(float);   // keys + row image
    Y6_REQUIRE(stage_bytes <= 160 * 1024 - 1024, "nms: %d classes do not fit the LDS staging area", d->nc);
    static bool cand_attr = false;
    if (stage_bytes > 48 * 1024 && !cand_attr) {
        Y6_HIP(hipFuncSetAttribute((const void*)nms_candidates_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024 - 1024));
        cand_attr = true;
    }
    const unsigned blocks = (unsigned)d->B * (unsigned)((d->A + rpb - 1) / rpb);
    hipLaunchKernelGGL(nms_candidates_kernel, dim3(blocks), dim3(256), stage_bytes, s, d->pred, d->B, d->A, d->nc,
                       d->conf_thres, d->classes, d->n_classes, ml, rpb, keys, w.cap, counts);
    Y6_LAUNCH_CHECK();
    const size_t lds = kSweepLds > (size_t)kLdsKeys * sizeof(u64) ? kSweepLds : (size_t)kLdsKeys * sizeof(u64);
    static bool attr_set = false;
    if (!attr_set) {
        Y6_HIP(hipFuncSetAttribute((const void*)nms_sort_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(nms_sort_sweep_kernel, dim3(d->B), dim3(1024), lds, s, d->pred, d->A, d->nc, d->iou_thres,
                       d->agnostic, d->max_det, d->max_nms, d->max_wh, keys, w.cap, counts, d->out_dets, d->out_index,
                       d->out_count);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
