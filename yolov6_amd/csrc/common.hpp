// Shared helpers for libyolov6_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/yolov6_hip.h"

void y6_set_error(const char* fmt, ...);

#define Y6_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            y6_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return Y6_EHIP;                                                                  \
        }                                                                                    \
    } while (0)

#define Y6_REQUIRE(cond, ...)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            y6_set_error(__VA_ARGS__);                                                       \
            return Y6_EINVAL;                                                                \
        }                                                                                    \
    } while (0)

// hipGetLastError() is sticky per thread: an unrelated earlier failure (e.g. a device probe made by
// another library) would otherwise be reported by the next launch check.  Entry points clear it.
#define Y6_CLEAR_STALE_ERROR() (void)hipGetLastError()

// Y6_SYNC_TRACE=1 (debug): every launch site names itself on stderr and then waits for the device, so that a GPU memory
// fault (which aborts the process from the runtime's event thread) is preceded by the line of the launch that caused it.
static inline bool y6_sync_trace() {
    static const bool v = getenv("Y6_SYNC_TRACE") != nullptr;
    return v;
}

#define Y6_LAUNCH_CHECK()                                                                    \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            y6_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return Y6_EHIP;                                                                  \
        }                                                                                    \
        if (y6_sync_trace()) {                                                               \
            fprintf(stderr, "[y6-sync-trace] launched %s (%s:%d)\n", __func__, __FILE__, __LINE__); \
            fflush(stderr);                                                                  \
            (void)hipDeviceSynchronize();                                                    \
        }                                                                                    \
    } while (0)

static inline int y6_cdiv(int a, int b) { return (a + b - 1) / b; }
// BatchNorm statistics / backward sums (train.hip): at most this many blocks, each leaving its per-channel partial sums in the
// workspace; a second kernel adds them in a fixed order (y6_bn_stats_workspace_bytes / y6_bnact_bwd_workspace_bytes size for it)
constexpr int kBnPartBlocks = 1024;
static inline size_t y6_tensor_elems(const y6_tensor& t) { return (size_t)t.B * t.H * t.W * t.cstride; }

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// The reference's half-precision model rounds to fp16 at every op boundary (conv -> activation -> residual add are
// separate fp16 aten ops).  The fused epilogues compute in fp32 but round at the SAME places, so a fused kernel
// reproduces the reference's fp16 pipeline instead of being "more accurate" by up to 2 ulp per layer (measured:
// tests/test_gpu_parity_bench.py).  ReLU commutes with the rounding and needs none.
__device__ __forceinline__ float y6_round_f16(float v) { return (float)(_Float16)v; }

// n / d for tame operands (no denormals, no overflow: SiLU's d = 1 + exp(-v) with v an fp16 value): reciprocal estimate, one
// residual correction - the core of the IEEE division sequence without its scaling / fix-up instructions (4 VALU
// instructions instead of ~10; the SiLU epilogue of a 64-channel layer was as long as its MFMA loop).
__device__ __forceinline__ float y6_div_tame(float n, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    const float rem = __builtin_fmaf(-q, d, n);
    return __builtin_fmaf(rem, r, q);
}

__device__ __forceinline__ float y6_act(float v, int act) {
    switch (act) {
        case Y6_ACT_RELU: return v > 0.f ? v : 0.f;
        case Y6_ACT_SILU: v = y6_round_f16(v); return y6_div_tame(v, 1.f + __expf(fminf(-v, 80.f)))   /* exp stays finite: v / inf = -0 either way after the fp16 rounding */;
        case Y6_ACT_HARDSWISH: {
            v = y6_round_f16(v);
            float r = v + 3.f;
            r = r < 0.f ? 0.f : (r > 6.f ? 6.f : r);
            return v * r * (1.f / 6.f);
        }
        default: return v;
    }
}

// internal launchers shared between translation units
struct y6_conv_geom;  // conv_mfma.hip
int y6_conv_naive_launch(const y6_conv_desc* d, hipStream_t s);                 // conv_misc.hip
int y6_conv_default_variant(const y6_conv_desc* d);                             // conv_misc.hip: the shape-derived kernel choice (-1: none)
int y6_conv_mfma_launch(const y6_conv_desc* d, int variant, hipStream_t s,      // conv_mfma.hip
                        int up, int updy, int updx);
int y6_conv_mfma_supports(const y6_conv_desc* d, int variant);
int y6_conv_i8_launch(const y6_conv_i8_desc* q, hipStream_t s);             // conv_mfma.hip (int8 kernels)
int y6_conv_i8_variant(const y6_conv_i8_desc* q);
int y6_conv_dma_launch(const void* launch_record, int cf, int pf, int nw, int stages, int interleave, int chunk_channels, int stride,
                       int i8, int resident_weights, hipStream_t s);   // conv_dma.hip
int y6_conv_dma_halo_cap(int block_pixels, int stride);
int y6_conv_wreg_launch(const void* launch_record, int pf, int cout_waves, int pixel_waves, int stride, int i8, hipStream_t s);   // conv_wreg.hip
int y6_conv_wreg_max_pieces(int waves, int stride);
int y6_conv_pw_launch(const void* launch_record, int cout_waves, int pixel_frags, int i8, hipStream_t s);   // conv_pw.hip (1x1 stride 1)
int y6_conv_pw_i8_cin_ok(int cin, int block_pixels);
int y6_conv_pw_cin_ok(int cin, int block_pixels);
// nms.hip: where the key lists / their lengths of a y6_nms workspace live (the candidate sink of head_decode.hip writes them)
int y6_nms_workspace_views(void* workspace, size_t bytes, int B, int A, int nc, int multi_label, unsigned long long** keys, size_t* cap, int** counts);
double y6_conv_flops(const y6_conv_desc* d);
double y6_conv_bytes(const y6_conv_desc* d);
