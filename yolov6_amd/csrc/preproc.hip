// preproc.hip — letterbox on the device: uint8 HWC image -> resized + padded uint8 tile, optionally as the RGB planes the
// uint8 image conv reads (csrc/conv_misc.hip stem kernels, csrc/conv_fused.hip), so that a decoded frame goes from its cv2
// layout to the network's first MFMA without touching the host.
//
// Replaces: letterbox()  yolov6/data/data_augment.py:29-58  (cv2.resize INTER_LINEAR + cv2.copyMakeBorder) and the layout half
// of Inferer.process_image  yolov6/core/inferer.py:162-172  (`transpose((2, 0, 1))[::-1]`: HWC -> CHW, BGR -> RGB); the
// `image /= 255` of that function is folded into the uint8 image conv's load.
//
// The arithmetic is cv2's (opencv-python >= 4.1.2, requirements.txt:7 - un-pinned, not installed in this environment, not
// vendored by the reference: PARITY UNPINNED, restated from imgproc/src/resize.cpp as oracle/letterbox_oracle.py):
//   * source coordinate fx = float((dx + 0.5) * scale - 0.5) in double, scale = 1 / (double(dst) / src); sx = floor(fx); the
//     column weights are zeroed / clamped at the borders (sx < 0: sx = 0, fx = 0; sx >= W - 1: sx = W - 1, fx = 0), the rows are
//     clamped only;
//   * 11-bit fixed point: a = saturate_cast<short>(w * 2048) (round half to even), horizontal pass in int32
//     (S[sx] * a0 + S[sx + 1] * a1), vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
//   * an exact 2 x 2 down-scale takes cv2's INTER_AREA fast path: (s00 + s01 + s10 + s11 + 2) >> 2.
#include "common.hpp"

namespace {

struct LbArgs {
    const uint8_t* src;
    int H, W;                 // source, HWC, 3 channels, row pitch W * 3
    uint8_t* dst;
    int oh, ow;               // padded output size
    int nh, nw;               // resized (un-padded) size
    int top, left;
    long dps, dcs;            // destination pixel stride / channel stride in bytes (HWC: 3 / 1; planes: 1 / oh * ow)
    int crev;                 // 1: channel c goes to plane 2 - c (BGR -> RGB)
    int pad[3];               // border colour per SOURCE channel
    double scale_x, scale_y;
    int mode;                 // 0: copy (no resize), 1: bilinear (fixed point), 2: 2x2 area
};

__device__ __forceinline__ int sat_short_rn(float v) {
    int i = __float2int_rn(v);
    return i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
}

__global__ __launch_bounds__(256) void letterbox_kernel(const LbArgs a) {
    const long total = (long)a.oh * a.ow;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int oy = (int)(i / a.ow), ox = (int)(i - (long)oy * a.ow);
        const int dy = oy - a.top, dx = ox - a.left;
        int v[3];
        if (dy < 0 || dy >= a.nh || dx < 0 || dx >= a.nw) {
            v[0] = a.pad[0];
            v[1] = a.pad[1];
            v[2] = a.pad[2];
        } else if (a.mode == 0) {
            const uint8_t* s = a.src + ((size_t)dy * a.W + dx) * 3;
            v[0] = s[0];
            v[1] = s[1];
            v[2] = s[2];
        } else if (a.mode == 2) {
            const uint8_t* s0 = a.src + ((size_t)(2 * dy) * a.W + 2 * dx) * 3;
            const uint8_t* s1 = s0 + (size_t)a.W * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (s0[c] + s0[3 + c] + s1[c] + s1[3 + c] + 2) >> 2;
        } else {
            float fx = (float)(((double)dx + 0.5) * a.scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= (float)sx;
            if (sx < 0) {
                fx = 0.f;
                sx = 0;
            }
            if (sx >= a.W - 1) {
                fx = 0.f;
                sx = a.W - 1;
            }
            float fy = (float)(((double)dy + 0.5) * a.scale_y - 0.5);
            const int sy = (int)floorf(fy);
            fy -= (float)sy;
            const int a0 = sat_short_rn((1.f - fx) * 2048.f), a1 = sat_short_rn(fx * 2048.f);
            const int b0 = sat_short_rn((1.f - fy) * 2048.f), b1 = sat_short_rn(fy * 2048.f);
            const int y0 = sy < 0 ? 0 : (sy < a.H ? sy : a.H - 1);
            const int y1 = sy + 1 < 0 ? 0 : (sy + 1 < a.H ? sy + 1 : a.H - 1);
            const int x1 = sx + 1 < a.W ? sx + 1 : sx;          // (weight a1 is 0 where sx + 1 would leave the row)
            const uint8_t* r0 = a.src + (size_t)y0 * a.W * 3;
            const uint8_t* r1 = a.src + (size_t)y1 * a.W * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = r0[sx * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
                const int h1 = r1[sx * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
                const int t = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                v[c] = t < 0 ? 0 : (t > 255 ? 255 : t);
            }
        }
        uint8_t* d = a.dst + (size_t)i * a.dps;
#pragma unroll
        for (int c = 0; c < 3; ++c) d[(size_t)(a.crev ? 2 - c : c) * a.dcs] = (uint8_t)v[c];
    }
}

}  // namespace

extern "C" int y6_letterbox(const y6_letterbox_desc* d, void* stream) {
    Y6_CLEAR_STALE_ERROR();
    Y6_REQUIRE(d && d->src && d->dst, "letterbox: null argument");
    Y6_REQUIRE(d->H > 0 && d->W > 0 && d->new_h > 0 && d->new_w > 0 && d->top >= 0 && d->left >= 0 &&
                   d->top + d->new_h <= d->out_h && d->left + d->new_w <= d->out_w,
               "letterbox: the resized image %dx%d at (%d,%d) does not fit the %dx%d output", d->new_h, d->new_w, d->top, d->left,
               d->out_h, d->out_w);
    LbArgs a;
    memset(&a, 0, sizeof(a));
    a.src = (const uint8_t*)d->src;
    a.H = d->H;
    a.W = d->W;
    a.dst = (uint8_t*)d->dst;
    a.oh = d->out_h;
    a.ow = d->out_w;
    a.nh = d->new_h;
    a.nw = d->new_w;
    a.top = d->top;
    a.left = d->left;
    if (d->planar) {
        a.dps = 1;
        a.dcs = (long)d->out_h * d->out_w;
    } else {
        a.dps = 3;
        a.dcs = 1;
    }
    a.crev = d->reverse_channels ? 1 : 0;
    for (int c = 0; c < 3; ++c) a.pad[c] = d->pad[c];
    // cv::resize: inv_scale = double(dsize) / ssize, scale = 1. / inv_scale
    const double inv_x = (double)d->new_w / d->W, inv_y = (double)d->new_h / d->H;
    a.scale_x = 1.0 / inv_x;
    a.scale_y = 1.0 / inv_y;
    if (d->new_h == d->H && d->new_w == d->W)
        a.mode = 0;
    else if (2 * d->new_w == d->W && 2 * d->new_h == d->H)   // is_area_fast with iscale 2 / 2: INTER_LINEAR runs INTER_AREA's fast path
        a.mode = 2;
    else
        a.mode = 1;
    const long total = (long)a.oh * a.ow;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(letterbox_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    Y6_LAUNCH_CHECK();
    return Y6_OK;
}
