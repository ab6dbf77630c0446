// nms.hip — batched, class-aware NMS on the device.
// Restates non_max_suppression (reference yolov6/utils/nms.py:31-105) and the
// torchvision.ops.nms it calls at :96 (torchvision is an un-vendored dependency; its
// published algorithm: stable sort by score descending, greedy sweep, suppress j when
// IoU(i,j) > thr with IoU = inter / (area_i + area_j - inter), fp32, no +1, no eps).
//
// Stage 1 (nms_candidates_kernel): a block fetches 64 consecutive anchor rows into LDS; rows that pass
//   obj > conf AND max(cls) > conf (nms.py:48) emit (conf = cls*obj (:69), class) candidates
//   (multi-label: every class with conf > thr (:75-77); else best class (:79-80)), filtered
//   by `classes` (:83-84), appended with one global atomic per block.  A candidate is a 64-bit key
//   (conf bits << 32 | ~flat) with flat = anchor*nc + cls, so a descending key sort IS the
//   reference's "score descending, earlier row first" order.
// Stage 2 (nms_chunk_sort_kernel + nms_merge_rank_kernel): the keys of every image are sorted by the whole
//   chip: 2048-key chunks by bitonic networks in LDS, then a rank merge that also applies the
//   max_nms cap (:90-91).
// Stage 3 (nms_sweep_kernel): one 1024-thread block per image, windowed greedy sweep over the sorted
//   list: xyxy boxes (:72, xywh2xyxy :21-28) offset by cls*max_wh (:94-95) are built 1024 at a time
//   in LDS; only the prefix of the window that can still be needed is tested against the boxes already
//   kept; up to 256 candidates per round are settled by a fixed-point iteration on their suppression
//   matrix (= the sequential greedy order), until max_det (:97-98) boxes are kept.
// Compile with -ffp-contract=off: index parity needs the reference's unfused fp32 arithmetic.
#include "common.hpp"
#include "nms_cand.hpp"

namespace {

typedef unsigned long long u64;

// One block = `rpb` consecutive anchor rows of ONE image (blockIdx interleaves images so that
// concurrently running blocks append to different per-image counters).  The rows are one contiguous run
// of the prediction tensor: they are fetched with 16-byte loads, all in flight at once, and examined in
// LDS with a thread per (row, 8 classes) - no cross-lane traffic.  Pass 1 raises the row flag of
// nms.py:48 (obj > conf AND max cls > conf); pass 2 finds the candidates of flagged rows and reserves
// block-local slots with LDS atomics; one global atomic per block reserves the block's slice of the
// image's key list and the keys go straight to it.  LDS holds only the rows (22 KiB for 80 classes), so
// seven blocks per CU keep ~150 KiB of loads in flight.
// History: v1 one global atomic per row (same-address atomics serialise at ~12 ns: 4.3 ms per call);
// v2 wave per row with shuffles and a 40 KiB LDS key stage (0.17 ms); v3 a thread per (row, 8 classes) 0.05 ms - LDS reads
// with a stride of eight words, 8-way bank conflicts; v4 (nms_cand.hpp, shared with the decode launch): a thread per class score in
// (row, class) order - conflict-free.

__global__ __launch_bounds__(256) void nms_candidates_kernel(const float* __restrict__ pred, int B, int A, int nc,
                                                             float conf_thres, const int* __restrict__ classes,
                                                             int n_classes, int multi_label, int rpb,
                                                             u64* __restrict__ keys, size_t cap,
                                                             int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_cnt, s_base, s_gen;
    const int tid = threadIdx.x;
    const int b = blockIdx.x % B, chunk = blockIdx.x / B;
    const int no = nc + 5;
    float* rows = reinterpret_cast<float*>(smem);
    int* rowflag = reinterpret_cast<int*>(rows + (size_t)rpb * no);
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
    __syncthreads();
    const y6cand::lds_f32* lrows = (const y6cand::lds_f32*)rows;
    const y6cand::CandSel2 cs = y6cand::cand_select(lrows, (y6cand::lds_i32*)rowflag, nrows, no, nc, conf_thres, classes, n_classes, multi_label,
                                                    (y6cand::lds_i32*)&s_cnt, (y6cand::lds_i32*)&s_gen);
    const int total = s_cnt;
    if (total == 0) return;
    if (tid == 0) s_base = atomicAdd(&counts[b * y6cand::kCountStride], total);
    __syncthreads();
    y6cand::cand_publish(lrows, nrows, no, nc, a0, multi_label, cs, keys + (size_t)b * cap + s_base);
}

// descending bitonic sort of P (power of two) keys held in LDS by the whole block.
// LDS variant.  Wave w owns the chunk [w*C, (w+1)*C) (C = P / waves, at least 128): every compare-exchange
// stage with distance j < C stays inside one chunk, so those stages run with wave-level ordering only;
// block barriers remain for j >= C.
__device__ __forceinline__ void bitonic_desc_lds(u64* keys, int P) {
    const int T = blockDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int C = P / (T >> 6);
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

constexpr int kBatch = 256;       // candidates settled per round of the sweep (kBatch / 64 mask words per candidate)
constexpr int kWin = 1024;        // sweep window: sorted candidates resident in LDS at a time (one per thread) when max_det is in the thousands,
constexpr int kWinBig = 4096;     // ... four per thread otherwise: with many same-class overlaps (the bench's decode output keeps ~3 % of the
                                  // candidates) the sweep walks ten 1024-windows to find 300 boxes, each paying box gather + barriers
constexpr int kKeptCap = 8192;    // most kept boxes LDS can hold next to the window (160 KiB per CU): the max_det limit
// LDS of the sweep for a given kept-box capacity (max_det rounded up to 256): window boxes + kept boxes + alive words
// (+ with class chains: class per window candidate, next-of-same-class link per kept box, chain head per class)
static inline size_t sweep_lds(int kept_cap, int nheads, int win = kWin) {
    return (size_t)(win + kept_cap) * 16 + (win / 64) * 8 + (nheads > 0 ? (size_t)(win + 2 * kept_cap + nheads) * 4 : 0);
}

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

// IoU(bi, bj) > thr with the arithmetic above, but the division only runs for boxes that overlap.
// Exact: when right-left (or bottom-top) is <= 0 or NaN the reference's clamped extent is 0, the
// intersection is 0 (or NaN against an infinite extent) and `0/union > thr`, `NaN > thr` are both false.
// With class offsets (nms.py:94-95) almost every pair ends here - the sweep's inner loop is 5x shorter.
__device__ __forceinline__ bool nms_suppresses(const float4 bi, const float4 bj, float thr) {
    const float dw = fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x);
    if (!(dw > 0.f)) return false;
    const float dh = fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y);
    if (!(dh > 0.f)) return false;
    // Overlapping boxes: decide without the division whenever the comparison is not borderline.  inter, union are the
    // reference's own values (dw, dh > 0: the clamps are no-ops); with p = fl(thr * union), inter < p (1 - 2^-20) implies
    // fl(inter / union) < thr and inter > p (1 + 2^-20) implies fl(inter / union) > thr (each fl() is within 2^-24), so only
    // the sliver in between needs the exact quotient.  The sweep's pair loops were VALU-bound on the ~30-instruction
    // division sequence (same-class neighbours overlap without reaching the threshold).
    const float inter = dw * dh;
    const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
    const float area_j = (bj.z - bj.x) * (bj.w - bj.y);
    const float uni = area_i + area_j - inter;
    if (uni > 0.f && thr > 0.f) {
        const float p = thr * uni;
        if (inter < p * 0.99999905f) return false;
        if (inter > p * 1.00000095f) return true;
    }
    return inter / uni > thr;
}

// ---- sort, spread over the whole chip -------------------------------------------------------------
// One image's candidate list used to be sorted by its single sweep block (one CU, LDS-throughput bound:
// ~140 us for 13k keys while 224 CUs idled).  Now:
//   nms_chunk_sort_kernel   every 2048-key chunk of every image is sorted (descending) in LDS by its own
//                           256-thread block, in place;
//   nms_merge_rank_kernel   every key finds its final rank = its position in its own chunk + the number of
//                           larger keys in each other chunk of the image (binary search; keys are unique, so
//                           ranks are a permutation) and is scattered to sorted[b][rank] if rank < max_nms
//                           (the cap of nms.py:90-91).
constexpr int kChunk = 2048;

// v2 (r04r): the chunk lives in REGISTERS - thread t holds keys 8 t .. 8 t + 7.  Of the 66 compare-exchange steps of a 2048-key
// bitonic network 30 have distance < 8 (inside a thread), 33 have distance 8 .. 256 (a lane xor of 1 .. 32: wave shuffles) and only
// 3 cross waves (through LDS).  The first version kept the keys in LDS and paid two dependent LDS round trips plus a wait for every
// one of the 4 exchanges a lane made per step: 50 us per launch, one block per CU with nothing to hide behind.
__device__ __forceinline__ void cmpx(u64& a, u64& b, bool desc) {   // a at the lower position: descending puts the larger there
    const bool sw = desc ? (a < b) : (a > b);
    const u64 t = a;
    a = sw ? b : a;
    b = sw ? t : b;
}

__global__ __launch_bounds__(256) void nms_chunk_sort_kernel(u64* __restrict__ keys, size_t cap,
                                                             const int* __restrict__ counts, int chunks_per_image) {
    __shared__ __attribute__((aligned(16))) u64 lk[kChunk];
    static_assert(kChunk == 2048, "8 keys per thread, 256 threads");
    const int tid = threadIdx.x;
    // the grid holds `chunks_per_image` blocks per image (a few); a block walks the image's chunks with that stride
    const int b = blockIdx.x / chunks_per_image;
    const int n = counts[b * y6cand::kCountStride];
    for (int c = blockIdx.x - b * chunks_per_image; c * kChunk < n; c += chunks_per_image) {
        const int c0 = c * kChunk;
        u64* gk = keys + (size_t)b * cap + c0;
        const int m = min(kChunk, n - c0);
        u64 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (tid * 8 + r < m) ? gk[tid * 8 + r] : 0ull;   // real keys are never 0: the padding sorts last
#pragma unroll
        for (int k = 2; k <= kChunk; k <<= 1) {
            // direction of this thread's keys in the merges of size k: for k < 8 it depends on the register index
            const bool desc_t = ((tid * 8) & k) == 0;
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j < 8) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        if ((r & j) == 0) cmpx(v[r], v[r | j], k < 8 ? ((r & k) == 0) : desc_t);
                } else {
                    const bool lower = ((tid * 8) & j) == 0;
                    const bool take_max = lower == desc_t;
                    if (j < 512) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const u64 o = __shfl_xor(v[r], j >> 3, 64);
                            v[r] = take_max ? (o > v[r] ? o : v[r]) : (o < v[r] ? o : v[r]);
                        }
                    } else {
                        __syncthreads();   // (the previous cross-wave step's reads)
#pragma unroll
                        for (int r = 0; r < 8; ++r) lk[tid * 8 + r] = v[r];
                        __syncthreads();
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const u64 o = lk[((tid * 8) ^ j) + r];
                            v[r] = take_max ? (o > v[r] ? o : v[r]) : (o < v[r] ? o : v[r]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (tid * 8 + r < m) gk[tid * 8 + r] = v[r];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void nms_merge_rank_kernel(const u64* __restrict__ keys, size_t cap,
                                                             const int* __restrict__ counts, int chunks_per_image,
                                                             u64* __restrict__ sorted, int max_nms) {
    // The other chunks of the image pass through a double-buffered LDS stage (coalesced 8-byte loads, the next
    // one in flight while the current one is searched): the 11-12 probes of a binary search cost LDS latency,
    // not a dependent chain of L2 round trips (first version: 0.16 ms, all of it load latency).
    __shared__ __attribute__((aligned(16))) u64 run[2][kChunk];
    constexpr int KPT = kChunk / 256;   // keys per thread
    const int b = blockIdx.x / chunks_per_image;
    const int n = counts[b * y6cand::kCountStride];
    const int tid = threadIdx.x;
    const u64* gk = keys + (size_t)b * cap;
    const int nchunks = (n + kChunk - 1) / kChunk;
    for (int c = blockIdx.x - b * chunks_per_image; c < nchunks; c += chunks_per_image) {
    const int c0 = c * kChunk;
    u64 mykey[KPT];
    int rank[KPT];
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        const int p = tid + q * 256;
        mykey[q] = (c0 + p < n) ? gk[c0 + p] : 0ull;
        rank[q] = p;
    }
    u64 stg[KPT];
    auto fetch = [&](int oc) {
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const int i = oc * kChunk + tid + q * 256;
            stg[q] = i < n ? gk[i] : 0ull;
        }
    };
    auto publish = [&](int buf) {
#pragma unroll
        for (int q = 0; q < KPT; ++q) run[buf][tid + q * 256] = stg[q];
    };
    auto other = [&](int t) { return t < c ? t : t + 1; };   // t-th chunk that is not c
    const int nother = nchunks - 1;
    if (nother > 0) {
        fetch(other(0));
        publish(0);
    }
    __syncthreads();
    for (int t = 0; t < nother; ++t) {
        const int oc = other(t);
        if (t + 1 < nother) fetch(other(t + 1));
        const u64* r = run[t & 1];
        const int len = min(kChunk, n - oc * kChunk);
        int lo[KPT], hi[KPT];
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            lo[q] = 0;
            hi[q] = len;
        }
#pragma unroll
        for (int it = 0; it < 12; ++it) {   // 2^11 = kChunk: 12 halvings settle any length <= kChunk
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                // branch-free so that the KPT probes of one halving are in flight together
                const bool act = lo[q] < hi[q];
                const int mid = (lo[q] + hi[q]) >> 1;
                const bool gt = r[act ? mid : 0] > mykey[q];   // r is descending: lo converges to #entries > key
                lo[q] = (act && gt) ? mid + 1 : lo[q];
                hi[q] = (act && !gt) ? mid : hi[q];
            }
        }
#pragma unroll
        for (int q = 0; q < KPT; ++q) rank[q] += lo[q];
        if (t + 1 < nother) publish((t + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        const int p = tid + q * 256;
        if (c0 + p < n && rank[q] < max_nms) sorted[(size_t)b * max_nms + rank[q]] = mykey[q];
    }
    __syncthreads();   // the LDS stage is re-used by the block's next chunk
    }
}

// One 1024-thread block per image: windowed greedy sweep over the sorted candidate list.  2048 sorted
// candidates at a time become class-offset boxes in LDS, are first tested against the boxes kept so far,
// then settled in batches of 64 (see below).  Boxes are only ever built for the windows the sweep reaches
// before max_det boxes are kept.
template <int WIN>
__global__ __launch_bounds__(1024) void nms_sweep_kernel(const float* __restrict__ pred, int A, int nc,
                                                         float iou_thres, int agnostic, int max_det, int max_nms, int kept_cap,
                                                         int nheads, float max_wh, const u64* __restrict__ sorted,
                                                         const int* __restrict__ counts,
                                                              float* __restrict__ out_dets, int* __restrict__ out_index,
                                                              int* __restrict__ out_count, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // optional s_memtime trace of block 0 / thread 0 (tools/nms_trace.py, env Y6_NMS_TRACE = device address of 512 words)
    int dbg_n = 0;
#define NT_(tag)                                                             \
    do {                                                                     \
        if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && dbg_n < 250) { \
            dbg[2 * dbg_n] = __builtin_amdgcn_s_memtime();                   \
            dbg[2 * dbg_n + 1] = (unsigned long long)(tag);                  \
            ++dbg_n;                                                         \
        }                                                                    \
    } while (0)
    NT_(1);
    __shared__ int s_batch[kBatch];
    __shared__ u64 s_col[kBatch * 4];
    __shared__ float4 s_bbox[kBatch];
    __shared__ u64 s_wm[64];
    __shared__ int s_wpre[64];
    __shared__ int s_nb, s_k0, s_kept, s_last;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    int n = counts[b * y6cand::kCountStride];
    // the outputs need no pre-fill by the caller: rows past the kept count are written here (zeros / -1)
    auto pad_outputs = [&](int from) {
        for (int i = from * 6 + tid; i < max_det * 6; i += T) out_dets[(size_t)b * max_det * 6 + i] = 0.f;
        for (int i = from + tid; i < max_det; i += T) out_index[(size_t)b * max_det + i] = -1;
    };
    if (n <= 0) {
        pad_outputs(0);
        if (tid == 0) out_count[b] = 0;
        return;
    }
    const u64* gk = sorted + (size_t)b * max_nms;   // descending, already cut to max_nms (nms.py:90-91)
    if (n > max_nms) n = max_nms;

    float4* wbox = reinterpret_cast<float4*>(smem);
    float4* kbox = wbox + WIN;
    u64* walive = reinterpret_cast<u64*>(kbox + kept_cap);
    // Class chains (nheads > 0).  With the class offset of nms.py:94-95 only boxes of ONE class can overlap, so a
    // candidate is tested against the kept boxes of its class only: every kept box carries the index of the previously
    // kept box of its class, khead[c] is the newest.  Testing all 1024 window candidates against all (up to 300) kept
    // boxes was the sweep's floor: 300 k box tests on one CU = 36 us of VALU issue; with 80 classes it is 1/80 of that.
    int* wcls = reinterpret_cast<int*>(walive + WIN / 64);
    int* knext = wcls + WIN;
    int* kcls = knext + kept_cap;     // class of every kept box: short linear scans filter on it instead of chasing the chain
    int* khead = kcls + kept_cap;
    const bool chains = nheads > 0;
    if (chains)
        for (int c = tid; c < nheads; c += T) khead[c] = -1;
    int* kept_pos = out_index + (size_t)b * max_det;   // sorted positions first; converted to flat ids at the end
    const int no = nc + 5;
    int kept = 0;
    for (int pos = 0; pos < n && kept < max_det; pos += WIN) {
        const int wn = (n - pos) < WIN ? (n - pos) : WIN;
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
            if (chains) wcls[t] = agnostic ? 0 : (int)cls;
        }
        // every candidate of the window starts alive; alive bits are only brought up to date (tested against the boxes kept
        // so far) for the prefix of the window the sweep actually needs - see below
        for (int w = tid; w < WIN / 64; w += T) {
            const int lo = w * 64;
            walive[w] = wn >= lo + 64 ? ~0ull : (wn > lo ? ((1ull << (wn - lo)) - 1ull) : 0ull);
        }
        __syncthreads();
        NT_(2);
        // Greedy sweep inside the window, kBatch (256) undecided candidates at a time, over a LAZILY tested prefix.
        //   * `tested`: candidates [0, tested) have been tested against every box kept so far.  Before a round the prefix
        //     is extended to cur + need + need/4 + 64 (need = boxes still wanted): the new candidates are tested against
        //     ALL kept boxes of their class.  With well separated boxes (the bench's decode output keeps every one of its
        //     first 300 candidates) that is 375 tests of nothing instead of testing all 1024 window candidates against the
        //     256 boxes of the first round (115 k of the sweep's 216 k cycles, tools/nms_trace.py).
        //   * gather: every thread ranks its own candidate among the alive ones of [cur, tested) from per-word prefix counts
        //     (was: wave 0 peeling set bits one at a time, 10.8 k cycles per round) and copies its box into a compact
        //     batch array, so the suppression matrix reads one LDS location per pair.
        //   * matrix in COLUMN form (bit j of col[i] <=> j < i and IoU(j, i) > thr); wave 0 solves
        //     kept[i] = no kept j < i suppresses i  by fixed-point iteration from "all kept" (the greedy solution is the
        //     unique fixed point: candidate 0 is right from the start, candidate i once 0..i-1 are; in practice the depth
        //     of the longest suppression chain, 2-4 rounds of four ballots) instead of a 64-step scalar scan per 64 candidates.
        //   * then the alive candidates of (last, tested) are tested against the new boxes.
        // Exactly the sequential greedy order: a candidate enters a round only after it has been tested against every
        // box kept before that round.
        int cur = 0, tested = 0;
        // candidates [t0, t1) of the window against kept boxes [ka, kb): spread over all 1024 threads as (candidate, slice of
        // the box range) - with a few hundred candidates and a few hundred boxes a thread per candidate left most of the
        // block idle and every scan was a serial chain of LDS round trips (100 k cycles for 183 x 256 tests)
        auto test_range = [&](int t0, int t1, int ka, int kb) {
            const int ncand = t1 - t0;
            if (ncand <= 0 || kb <= ka) return;
            int cr = 64;
            while (cr < ncand && cr < 1024) cr <<= 1;
            const int slices = 1024 / cr, per = (kb - ka + slices - 1) / slices;
            const int ci = tid & (cr - 1), si = tid / cr;
            const int k_lo = ka + si * per, k_hi = (k_lo + per < kb) ? k_lo + per : kb;
            for (int base = 0; base < ncand; base += cr) {
                const int t = t0 + base + ci;
                if (base + ci < ncand && ((walive[t >> 6] >> (t & 63)) & 1ull)) {
                    const float4 bj = wbox[t];
                    const int c = chains ? wcls[t] : 0;
                    bool hit = false;
                    for (int k = k_lo; k < k_hi && !hit; ++k)
                        if ((!chains || kcls[k] == c) && nms_suppresses(kbox[k], bj, iou_thres)) hit = true;
                    if (hit) atomicAnd(&walive[t >> 6], ~(1ull << (t & 63)));
                }
            }
        };
        NT_(3);
        while (kept < max_det && cur < wn) {
            const int need = max_det - kept;
            int upto = cur + need + (need >> 2) + 64;
            if (upto > wn) upto = wn;
            if (upto > tested) {
                if (chains && kept > 512) {   // long kept lists: walk the class chain (one thread per candidate)
                    for (int t = tested + tid; t < upto; t += T) {
                        const float4 bj = wbox[t];
                        for (int k = khead[wcls[t]]; k >= 0; k = knext[k])
                            if (nms_suppresses(kbox[k], bj, iou_thres)) {
                                atomicAnd(&walive[t >> 6], ~(1ull << (t & 63)));
                                break;
                            }
                    }
                } else {
                    test_range(tested, upto, 0, kept);
                }
                tested = upto;
                __syncthreads();
            }
            if (wave == 0) {   // alive words restricted to [cur, tested), their exclusive prefix counts
                u64 w = 0ull;
                if (lane < WIN / 64) {
                    w = walive[lane];
                    const int lo = lane * 64;
                    if (lo + 64 <= cur || lo >= tested) w = 0ull;
                    if (lo < cur && cur < lo + 64) w &= ~((1ull << (cur - lo)) - 1ull);
                    if (lo < tested && tested < lo + 64) w &= (1ull << (tested - lo)) - 1ull;
                }
                const int pc = __popcll(w);
                int incl = pc;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += v;
                }
                s_wm[lane] = w;
                s_wpre[lane] = incl - pc;
                if (lane == 63) s_nb = incl;   // alive candidates in the range (may exceed kBatch)
            }
            __syncthreads();
            const int total = s_nb;
            // round size: no more candidates than could still be wanted (the suppression matrix costs nb^2 pair tests)
            int cap = need + (need >> 2) + 32;
            if (cap > kBatch) cap = kBatch;
            const int nb = total < cap ? total : cap;
            NT_(10 + (nb << 8));
            if (nb == 0) {   // nothing alive in the tested prefix: move on (extends the prefix, or ends the window)
                cur = tested;
                __syncthreads();
                continue;
            }
#pragma unroll
            for (int r = 0; r < (WIN + 1023) / 1024; ++r) {
                const int t = tid + r * 1024;
                if (t < WIN) {
                    const u64 w = s_wm[t >> 6];
                    if ((w >> (t & 63)) & 1ull) {
                        const int rank = s_wpre[t >> 6] + __popcll(w & ((1ull << (t & 63)) - 1ull));
                        if (rank < nb) {
                            s_batch[rank] = t;
                            s_bbox[rank] = wbox[t];
                        }
                    }
                }
            }
            __syncthreads();
            {   // column masks: thread (i, jq) tests candidate i against the earlier candidates 64*jq .. 64*jq+63
                const int i = tid >> 2, jq = tid & 3;
                if (i < kBatch) {
                    u64 bits = 0ull;
                    if (i < nb && jq * 64 < i) {
                        const float4 bi = s_bbox[i];
                        const int jend = (i < jq * 64 + 64) ? i : jq * 64 + 64;
#pragma unroll 8
                        for (int j = jq * 64; j < jend; ++j)
                            if (nms_suppresses(s_bbox[j], bi, iou_thres)) bits |= 1ull << (j & 63);
                    }
                    s_col[i * 4 + jq] = bits;
                }
            }
            __syncthreads();
            NT_(11);
            if (wave == 0) {
                // lane l owns candidates l, l+64, l+128, l+192
                u64 col[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) col[q][wd] = s_col[(lane + 64 * q) * 4 + wd];
                u64 kv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) kv[q] = __ballot(lane + 64 * q < nb);
                for (int round = 0; round < kBatch; ++round) {
                    u64 nv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool k = (lane + 64 * q < nb) && ((col[q][0] & kv[0]) | (col[q][1] & kv[1]) | (col[q][2] & kv[2]) | (col[q][3] & kv[3])) == 0ull;
                        nv[q] = __ballot(k);
                    }
                    const bool same = nv[0] == kv[0] && nv[1] == kv[1] && nv[2] == kv[2] && nv[3] == kv[3];
#pragma unroll
                    for (int q = 0; q < 4; ++q) kv[q] = nv[q];
                    if (same) break;
                }
                // at most max_det boxes over all: keep the first `room` of this batch
                const int room = max_det - kept;
                int before[4], nk = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    before[q] = nk;
                    const int c = __popcll(kv[q]);
                    if (nk + c > room) {   // cut inside word q: keep its first room - nk set bits
                        u64 x = kv[q], keepw = 0ull;
                        for (int r = nk; r < room; ++r) {
                            const u64 low = x & (~x + 1ull);
                            keepw |= low;
                            x ^= low;
                        }
                        kv[q] = keepw;
#pragma unroll
                        for (int q2 = q + 1; q2 < 4; ++q2) kv[q2] = 0ull;
                    }
                    nk += __popcll(kv[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = lane + 64 * q;
                    if (i < nb) {
                        const int idx = s_batch[i];
                        if ((kv[q] >> lane) & 1ull) {
                            const int kp = kept + before[q] + __popcll(kv[q] & ((1ull << lane) - 1ull));
                            kbox[kp] = s_bbox[i];
                            kept_pos[kp] = pos + idx;
                            if (chains) {
                                knext[kp] = atomicExch(&khead[wcls[idx]], kp);   // newest first
                                kcls[kp] = wcls[idx];
                            }
                        }
                        atomicAnd(&walive[idx >> 6], ~(1ull << (idx & 63)));   // decided either way
                    }
                }
                if (lane == 0) {
                    s_k0 = kept;
                    s_kept = kept + nk;
                    s_last = s_batch[nb - 1];
                }
            }
            __syncthreads();
            const int k0 = s_k0, last = s_last;
            kept = s_kept;
            NT_(12 + (kept << 8));
            if (kept >= max_det) break;
            test_range(last + 1, tested, k0, kept);   // this round's boxes are kbox[k0, kept)
            __syncthreads();
            NT_(13);
            cur = total <= nb ? tested : last + 1;   // everything alive in the prefix was in this round: skip the rest of it
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
    pad_outputs(kept);
    if (tid == 0) out_count[b] = kept;
    NT_(20);
#undef NT_
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
    w.total = align256(w.off_counts + (size_t)B * y6cand::kCountStride * sizeof(int));
    return w;
}

}  // namespace

int y6_nms_workspace_views(void* workspace, size_t bytes, int B, int A, int nc, int multi_label, unsigned long long** keys, size_t* cap, int** counts) {
    const NmsWs w = nms_ws_layout(B, A, nc, multi_label && nc > 1, 30000);
    Y6_REQUIRE(workspace && bytes >= w.total, "nms workspace too small (%zu < %zu)", bytes, w.total);
    *keys = (u64*)((char*)workspace + w.off_keys);
    *cap = w.cap;
    *counts = (int*)((char*)workspace + w.off_counts);
    return Y6_OK;
}

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
    static const int stop0 = getenv("Y6_NMS_STOP_AFTER") ? atoi(getenv("Y6_NMS_STOP_AFTER")) : 99;
    if (stop0 < 1) return Y6_OK;   // host-side cost of a call only
    if (!d->candidates_ready) {
    Y6_HIP(hipMemsetAsync(counts, 0, (size_t)d->B * y6cand::kCountStride * sizeof(int), s));
    // rows per block: up to 64 (y6cand::kCandRows), bounded by the 64 KiB row image
    int rpb = (int)((64 * 1024) / ((size_t)(d->nc + 5) * sizeof(float) + sizeof(int)));
    rpb = rpb > y6cand::kCandRows ? y6cand::kCandRows : rpb;
    if ((long)rpb * d->nc > 64 * 256) rpb = (64 * 256) / d->nc;   // a 64-bit pass mask per thread
    Y6_REQUIRE(rpb >= 1, "nms: %d classes do not fit the LDS row image", d->nc);
    const size_t stage_bytes = (size_t)rpb * (d->nc + 5) * sizeof(float) + (size_t)rpb * sizeof(int);   // row image + row flags
    Y6_REQUIRE(stage_bytes <= 64 * 1024, "nms: %d classes do not fit the LDS row image", d->nc);
    const unsigned blocks = (unsigned)d->B * (unsigned)((d->A + rpb - 1) / rpb);
    hipLaunchKernelGGL(nms_candidates_kernel, dim3(blocks), dim3(256), stage_bytes, s, d->pred, d->B, d->A, d->nc,
                       d->conf_thres, d->classes, d->n_classes, ml, rpb, keys, w.cap, counts);
    Y6_LAUNCH_CHECK();
    }
    static const int stop_after = getenv("Y6_NMS_STOP_AFTER") ? atoi(getenv("Y6_NMS_STOP_AFTER")) : 99;   // stage timing (tools/nms_bench.py)
    if (stop_after < 2) return Y6_OK;
    // sort: chunks, then rank-merge into the (otherwise unused) box area of the workspace
    u64* sorted = (u64*)(ws + w.off_boxes);
    int cpi = (int)(w.cap / kChunk) > 0 ? (int)(w.cap / kChunk) : 1;
    if (cpi > 16) cpi = 16;   // blocks per image in the sort grids (each walks the image's chunks with this stride)
    hipLaunchKernelGGL(nms_chunk_sort_kernel, dim3((unsigned)(d->B * cpi)), dim3(256), 0, s, keys, w.cap, counts, cpi);
    Y6_LAUNCH_CHECK();
    if (stop_after < 3) return Y6_OK;
    hipLaunchKernelGGL(nms_merge_rank_kernel, dim3((unsigned)(d->B * cpi)), dim3(256), 0, s, keys, w.cap, counts, cpi,
                       sorted, d->max_nms);
    Y6_LAUNCH_CHECK();
    if (stop_after < 4) return Y6_OK;
    // kept boxes live in LDS: capacity = max_det rounded up to 256 (2048 boxes = 48 KiB as before; 8192 = 148 KiB, one block per CU)
    int kept_cap = (d->max_det + 255) / 256 * 256;
    if (kept_cap < 2048) kept_cap = 2048;
    int nheads = d->agnostic ? 1 : d->nc;
    if (nheads > 4096 || sweep_lds(kept_cap, nheads) + 16 * 1024 > 160 * 1024) nheads = 0;   // no room (max_det in the thousands): linear scans
    // window size (candidates resident in LDS at a time): A/B switch Y6_NMS_WIN = 256 | 512 | 1024 | 4096
    static const int win_env = getenv("Y6_NMS_WIN") ? atoi(getenv("Y6_NMS_WIN")) : 0;
    int win = win_env ? win_env : kWin;
    if (win != 256 && win != 512 && win != 4096) win = kWin;
    if (sweep_lds(kept_cap, nheads, win) + 16 * 1024 > 160 * 1024) win = kWin;
    const size_t lds = sweep_lds(kept_cap, nheads, win);
    static unsigned long long* const trace = getenv("Y6_NMS_TRACE") ? (unsigned long long*)(uintptr_t)strtoull(getenv("Y6_NMS_TRACE"), nullptr, 10) : nullptr;
    auto launch = [&](auto kern) -> int {
        static size_t attr_lds = 0;   // one per instantiation of this lambda, i.e. per kernel
        if (lds > attr_lds) {
            Y6_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_lds = lds;
        }
        hipLaunchKernelGGL(kern, dim3(d->B), dim3(1024), lds, s, d->pred, d->A, d->nc, d->iou_thres, d->agnostic, d->max_det, d->max_nms,
                           kept_cap, nheads, d->max_wh, sorted, counts, d->out_dets, d->out_index, d->out_count, trace);
        return Y6_OK;
    };
    int rc;
    switch (win) {
        case 256: rc = launch(nms_sweep_kernel<256>); break;
        case 512: rc = launch(nms_sweep_kernel<512>); break;
        case 4096: rc = launch(nms_sweep_kernel<kWinBig>); break;
        default: rc = launch(nms_sweep_kernel<kWin>); break;
    }
    if (rc) return rc;
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
