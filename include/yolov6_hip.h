/*
 * yolov6_hip.h — C ABI of libyolov6_hip.so, the MI355X (gfx950) hot path behind the
 * YOLOv6 Python module API.
 *
 * The reference (meituan/YOLOv6) has no FFI of its own: its hot path is a composition of
 * aten ops called from Python modules.  Every entry point below therefore names the
 * reference Python interface it replaces (file:line relative to the reference tree), and
 * `INTEGRATION.md` shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HBM) unless the parameter name ends in `_host`.
 *   - Activations are NHWC fp16 ("channels-last"): element (b,y,x,c) of a tensor view lives
 *     at  base[((b*H + y)*W + x) * cstride + coff + c].  `cstride`/`coff` let a producer
 *     write straight into a channel slice of a consumer's buffer (concat-free necks,
 *     reference: torch.cat at yolov6/models/reppan.py:228,232, yolov6/layers/common.py:718).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Nothing here
 *     synchronises the device unless documented.
 *   - Return value: 0 on success, negative Y6_E* code on failure; y6_last_error() returns a
 *     thread-local human-readable message.  The Python host raises RuntimeError from it
 *     (the reference's assigner fallback catches exactly RuntimeError, models/losses/loss.py:105).
 */
#ifndef YOLOV6_HIP_H
#define YOLOV6_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y6_ABI_VERSION 1

enum { Y6_OK = 0, Y6_EINVAL = -1, Y6_EHIP = -2, Y6_EUNSUPPORTED = -3, Y6_ENOMEM = -4 };

/* activation_table of yolov6/layers/common.py:14-17 (+ None) */
enum { Y6_ACT_NONE = 0, Y6_ACT_RELU = 1, Y6_ACT_SILU = 2, Y6_ACT_HARDSWISH = 3 };

/* dtype tags for boundary tensors */
enum { Y6_F16 = 0, Y6_F32 = 1, Y6_U8 = 2 /* stem input only: uint8 pixels, read as imgs.half()/255 (core/evaler.py:121-123) */ };

int y6_abi_version(void);
/* sizeof of a public struct of this header by its name ("y6_conv_desc", ...), 0 if unknown: a hand-written mirror of these
 * structs (ctypes, cgo, JNI) checks itself against it when it loads the library. */
size_t y6_abi_sizeof(const char* name);
const char* y6_last_error(void);
/* number of CUs / arch name of the current device ("gfx950"); fails if no HIP device */
int y6_device_info(int* n_cu, char* arch, size_t arch_len);

/* ------------------------------------------------------------------------------------ */
/* A view of an NHWC fp16 activation tensor inside a (possibly wider) channel buffer.    */
typedef struct y6_tensor {
    void* data;      /* fp16 */
    int32_t B, H, W, C;
    int32_t cstride; /* channels of the underlying buffer (>= coff + C) */
    int32_t coff;    /* first channel of this view */
} y6_tensor;

/* ------------------------------------------------------------------------------------ */
/* Weight packing.
 * Replaces nothing at run time: it is the derived cache SURVEY §8b allows next to the
 * reference's OIHW parameters (`rbr_reparam.weight`, `block.conv.weight`, ...).
 *
 * src      : OIHW weights [Cout][Cin][K][K], fp16 or fp32 (src_dtype)
 * dst      : packed fp16, y6_packed_weight_elems(Cout,Cin,K) elements, layout
 *            [cout/32][cin/32][tap][kstep=2][lane=64][8]  (MFMA 32x32x16 A-fragment order),
 *            zero padded to 32-multiples of Cout and Cin.
 * For ConvTranspose2d(k=2,s=2) (common.py:181-194) use y6_pack_convt2x2_weight: src is
 * IOHW [Cin][Cout][2][2]; dst holds 4 packed 1x1 weights (one per (dy,dx)), each of
 * y6_packed_weight_elems(Cout,Cin,1) elements.                                            */
size_t y6_packed_weight_elems(int Cout, int Cin, int K);
int y6_pack_conv_weight(const void* src, int src_dtype, int Cout, int Cin, int K, void* dst, void* stream);
int y6_pack_convt2x2_weight(const void* src, int src_dtype, int Cin, int Cout, void* dst, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Fused conv + bias (+ post affine) + activation (+ residual), NHWC fp16, fp32 accumulate.
 * Replaces: ConvModule.forward_fuse  yolov6/layers/common.py:51-54
 *           RepVGGBlock.forward (deploy branch)  common.py:247-248
 *           QARepVGGBlock.forward (deploy, post-BN kept)  common.py:338-339
 *           BottleRep.forward residual  common.py:605-608  (res, res_alpha)
 *           nn.Conv2d 1x1 prediction convs  yolov6/models/effidehead.py:169-179
 * groups == 1, dilation == 1, padding == ksize/2 (the only forms the hot path uses).     */
typedef struct y6_conv_desc {
    y6_tensor in, out;
    const void* w_packed;      /* from y6_pack_conv_weight                                */
    const void* w_oihw;        /* optional: original OIHW fp16 weights (used by the naive  */
                               /* cross-check variant only)                               */
    const float* bias;         /* [Cout] or NULL                                          */
    const float* post_scale;   /* [Cout] or NULL : y = y*scale + shift before activation  */
    const float* post_shift;
    y6_tensor res;             /* optional residual (data==NULL: none): out += alpha*res   */
    const float* res_alpha;    /* device scalar or NULL (=> 1.0)                           */
    int32_t ksize;             /* 1 or 3 */
    int32_t stride;            /* 1 or 2 */
    int32_t act;               /* Y6_ACT_*  */
    int32_t variant;           /* -1 auto; >=0 force kernel variant (see y6_conv_variants) */
} y6_conv_desc;

int y6_conv2d(const y6_conv_desc* d, void* stream);
/* number of kernel variants / name of variant i ("mfma_c2p2", "naive", ...) */
int y6_conv_variants(void);
const char* y6_conv_variant_name(int i);
/* 1 if variant i can run desc d */
int y6_conv_variant_supports(const y6_conv_desc* d, int i);
/* The launch geometry variant i would use for desc d: host arithmetic only (nothing is launched, no pointer is dereferenced; the
 * CU count is the device's, or 256 without a device).  For callers that want to know a layer's tiling before they commit to a
 * batch size, and for the property test of the tile choosers (tests/test_host_cpu.py).  Returns Y6_EUNSUPPORTED when the
 * variant does not take the conv.                                                                                         */
typedef struct y6_conv_geometry {
    int32_t tile_h, tile_w;        /* output pixels of one work item                                        */
    int32_t tiles_x, tiles_y;      /* tiles per image                                                       */
    int32_t items;                 /* work items of the launch (tiles x cout blocks, padded to whole groups) */
    int32_t cout_blocks;           /* cout blocks (a block of the grid computes one of them per item)       */
    int32_t block_pixels;          /* pixel slots of a work item (tile_h * tile_w <= block_pixels)          */
    int32_t halo_h, halo_w;        /* input rows / columns a tile reads                                     */
    int32_t halo_pieces;           /* register-fed / LDS-DMA kernels: 1 KiB requests per halo stage, else 0 */
    int32_t halo_pieces_max;       /* ... and what the kernel can issue per stage (0: not applicable)       */
    int32_t row_pitch;             /* register-fed kernels: pixel slots per halo row in LDS, else 0         */
    uint64_t lds_bytes;            /* dynamic LDS of the launch                                             */
    uint64_t lds_limit;            /* what the kernel family may ask for                                    */
} y6_conv_geometry;
int y6_conv_launch_geometry(const y6_conv_desc* d, int variant, y6_conv_geometry* out);
/* Self-test of the LDS-DMA addressing the "dma*" conv variants rely on (csrc/conv_dma.hip): one wave copies 1 KiB from
 * src into LDS at byte offset lds_off (< 160 KiB) by `buffer_load_dwordx4 ... lds` and writes it to dst; lanes whose bit
 * is set in oob_mask request a piece beyond `bytes` and must read zeros.  No reference counterpart (diagnostic).       */
int y6_dma_probe(const void* src, unsigned bytes, unsigned lds_off, unsigned long long oob_mask, void* dst, void* stream);

/* ------------------------------------------------------------------------------------ */
/* int8 convolution (BASELINE configs[4]: YOLOv6-S QARepVGG int8 inference).
 * The reference tree holds NO int8 arithmetic: its int8 numbers come from TensorRT engines / NVIDIA
 * pytorch_quantization fake-quant (deploy/TensorRT/onnx_to_trt.py:62-112, tools/qat/qat_utils.py:61-146:
 * per-channel 8-bit weights, per-tensor 8-bit activations).  This entry point is the arithmetic those describe,
 * defined here and in oracle/int8_oracle.py (PARITY UNPINNED by construction):
 *   weights      symmetric per output channel, quantised on the host: s_w[c] = max|w[c]| / 127,
 *                w_q = clamp(rne(w / s_w[c]), -127, 127)  ->  y6_pack_conv_weight_i8
 *   activations  symmetric per tensor, amax from calibration (y6_absmax over the conv's input view):
 *                a = fp16(amax), inv = fp16(127 / a), x_q = clamp(rne(x * inv), -127, 127) with x the fp16 activation
 *                (exact product, one rounding) - applied while the conv stages its input, or by the PRODUCER's
 *                epilogue into an int8 twin tensor (q_out of the producer = q_in here: half the fill traffic)
 *   accumulate   int8 x int8 -> int32, exact (v_mfma_i32_32x32x32_i8)
 *   epilogue     fp32: y = fp32(acc) * dequant[c] + bias[c], dequant[c] = (fp32(a) / 127) * s_w[c]; then exactly the
 *                fp16 conv's epilogue (kept post-BN affine of QARepVGG common.py:338-339, activation, residual)
 * conv.in / conv.out are fp16 NHWC views as for y6_conv2d (conv.out.data may be NULL when only q_out is wanted);
 * conv.w_packed is the y6_pack_conv_weight_i8 image; conv.variant: 0 = heuristic, 1..6 force a per-tap tile, 7 / 8 / 9 the LDS-DMA kernels (512 / 256 / 512 pixel blocks; need q_in, k3 s1, Cin % 32 == 0; 9: Cin % 64 == 0),
 * 10 / 11 / 12 / 13 the register-fed kernels (csrc/conv_wreg.hip, int8 form: 7 / 4 pixel fragments per wave at stride 1, 3 at
 * stride 2, 13: stride 2 with 64-cout blocks; need q_in, k3, Cin % 64 == 0 (or Cin == 32), Cout % 128 == 0 (13: % 64), no residual,
 * no acc_out, out 16-byte / q_out 4-byte aligned views; anything else is Y6_EINVAL).                                        */
typedef struct y6_conv_i8_desc {
    y6_conv_desc conv;
    const float* dequant;      /* [Cout]                                                          */
    float in_amax;             /* calibrated max|x| of the input view (ignored when q_in is given) */
    y6_tensor q_in;            /* optional int8 NHWC view of the already quantised input (data NULL: none) */
    y6_tensor q_out;           /* optional int8 NHWC twin of the output, quantised with q_out_amax  */
    float q_out_amax;
    void* acc_out;             /* optional int32 [B*Ho*Wo][Cout]: raw accumulators (parity tests)   */
} y6_conv_i8_desc;
int y6_conv2d_i8(const y6_conv_i8_desc* d, void* stream);
/* the kernel variant (1..13, see conv.variant above) y6_conv2d_i8 would run for this descriptor: host arithmetic only, nothing is
 * launched and no pointer is dereferenced (only tested for NULL) - so that a caller planning int8 twins can see whether a conv
 * gets the register-fed / LDS-DMA kernels (which need q_in) before it decides where twins are written */
int y6_conv2d_i8_variant(const y6_conv_i8_desc* d);
/* src: int8 OIHW [Cout][Cin][K][K] -> [cout/32][cin/64][tap][kstep=2][lane=64][16] (zero padded); bytes of the image: */
size_t y6_packed_weight_i8_bytes(int Cout, int Cin, int K);
int y6_pack_conv_weight_i8(const void* src_i8_oihw, int Cout, int Cin, int K, void* dst, void* stream);
/* Calibration: *out = max(*out, max|x|) over the fp16 view (device float, zero it before the first batch).       */
int y6_absmax(const y6_tensor* x, float* out, void* stream);
/* Quantise an fp16 view into an int8 view with the rule above (for tensors whose producer is not an int8 conv).  */
int y6_quantize_i8(const y6_tensor* x, float amax, const y6_tensor* q, void* stream);

/* ConvTranspose2d(k=2, s=2, bias) : out[b,2y+dy,2x+dx,:] = bias + in[b,y,x,:] @ W[:,:,dy,dx]
 * Replaces: Transpose.forward  yolov6/layers/common.py:193-194                           */
typedef struct y6_convt_desc {
    y6_tensor in, out;         /* out.H == 2*in.H, out.W == 2*in.W                        */
    const void* w_packed;      /* from y6_pack_convt2x2_weight                            */
    const float* bias;
} y6_convt_desc;
int y6_convt2x2(const y6_convt_desc* d, void* stream);

/* Stem: 3x3 stride-2 pad-1 conv reading the caller's NCHW image directly (Cin <= 4) and
 * writing NHWC fp16, + bias + activation (+ optional post affine).
 * Replaces: EfficientRep.stem (RepVGGBlock deploy form)  yolov6/models/efficientrep.py:28-33
 * in_nchw: [B][Cin][H][W] fp16, fp32 or uint8 (in_dtype; uint8 pixels enter as the fp16 value of u/255,
 * the `imgs.half(); imgs /= 255` of core/evaler.py:121-123 folded into the load); w: OIHW fp32 [Cout][Cin][3][3]. */
typedef struct y6_stem_desc {
    const void* in_nchw;
    int32_t in_dtype;
    int32_t B, Cin, H, W;
    y6_tensor out;
    const float* w_oihw_f32;
    const float* bias;
    const float* post_scale;
    const float* post_shift;
    int32_t act;
    /* round 6 (int8 plans): optional int8 NHWC twin of the output for a quantised consumer, q = quantise(fp16 output, q_out_amax) -
     * the same rule and bits as y6_quantize_i8 / a y6_conv_i8_desc.q_out; `out.data` may then be NULL (twin only).  Needs the
     * tiled kernel's shapes (y6_stem_twin_supported).  data NULL: none. */
    y6_tensor q_out;
    float q_out_amax;
} y6_stem_desc;
int y6_stem_conv(const y6_stem_desc* d, void* stream);
int y6_stem_twin_supported(const y6_stem_desc* d);   /* 1: y6_stem_conv can write d->q_out (host-only query) */

/* Producer -> 3x3 stride-2 conv pairs as ONE launch (csrc/conv_fused.hip): the producer's output tile stays in LDS, the
 * intermediate tensor is never written to HBM (pw.out / stem.out are IGNORED, their data may be NULL).
 *   y6_fused_pw_s2   : s2(pw(x)), pw = 1x1 conv + bias + act, s2 = 3x3 stride-2 conv + bias + act.
 *     Replaces: BiFusion.forward `self.downsample(self.cv2(x[2]))`  yolov6/layers/common.py:711-716 (ConvBNReLU 1x1 -> ConvBNReLU 3x3 s2)
 *   y6_fused_stem_s2 : s2(stem(image)), stem as y6_stem_conv.
 *     Replaces: EfficientRep.forward `self.ERBlock_2[0](self.stem(x))`  yolov6/models/efficientrep.py:96-99 (RepVGG deploy forms)
 * Rounding points are the unfused graph's (producer output rounded to fp16 before the stride-2 conv reads it).
 * Shapes taken: pw 64->64 / 128->128 into s2 of the same width; stem 3->32 into s2 32->64; no post affine, no residual.
 * y6_fused_*_supported() says whether a pair is taken - callers fall back to the two separate ops.                    */
typedef struct y6_pw_s2_desc {
    y6_conv_desc pw;           /* ksize 1, stride 1 */
    y6_conv_desc s2;           /* ksize 3, stride 2; s2.in = the (virtual) pw.out geometry */
} y6_pw_s2_desc;
typedef struct y6_stem_s2_desc {
    y6_stem_desc stem;
    y6_conv_desc s2;
} y6_stem_s2_desc;
int y6_fused_pw_s2_supported(const y6_pw_s2_desc* d);
int y6_fused_pw_s2(const y6_pw_s2_desc* d, void* stream);
int y6_fused_stem_s2_supported(const y6_stem_s2_desc* d);
int y6_fused_stem_s2(const y6_stem_s2_desc* d, void* stream);

/* Letterbox on the device: uint8 HWC image (cv2.imread layout, 3 channels, dense rows) -> uint8 image resized with cv2's
 * INTER_LINEAR fixed-point arithmetic (or copied, when the size already fits) into the (new_h x new_w) window at (top, left)
 * of an (out_h x out_w) tile filled with `pad`.  planar = 0: the tile is HWC like the source (what letterbox() returns);
 * planar = 1: three [out_h][out_w] planes (CHW), reverse_channels = 1 additionally stores source channel c in plane 2 - c
 * (BGR -> RGB) - the uint8 NCHW image y6_stem_conv / y6_fused_stem_s2 read (in_dtype Y6_U8, `/ 255` folded into their load).
 * Replaces: letterbox()  yolov6/data/data_augment.py:29-58 ; Inferer.process_image (layout half)  yolov6/core/inferer.py:162-172.
 * The host computes new_h / new_w / top / left as data_augment.py:38-56 does.  PARITY UNPINNED: the arithmetic is opencv's
 * (requirements.txt:7, not vendored, not installed here), restated in oracle/letterbox_oracle.py.                      */
typedef struct y6_letterbox_desc {
    const void* src;
    int32_t H, W;
    void* dst;
    int32_t out_h, out_w;
    int32_t new_h, new_w;
    int32_t top, left;
    int32_t planar;
    int32_t reverse_channels;
    int32_t pad[3];
} y6_letterbox_desc;
int y6_letterbox(const y6_letterbox_desc* d, void* stream);

/* Three chained MaxPool2d(5, stride 1, pad 2) (-inf padding) of `x`, written to y1,y2,y3.
 * Replaces: SPPFModule.forward  common.py:106-112 ; CSPSPPFModule.forward  :150-158
 * x,y1,y2,y3 are usually the four channel slices of one 4*C buffer (no torch.cat).        */
int y6_sppf_pool(const y6_tensor* x, const y6_tensor* y1, const y6_tensor* y2, const y6_tensor* y3, void* stream);

/* The same three pools, also leaving the int8 twins of y1,y2,y3 for quantised consumers (int8 plans, BASELINE configs[4]: the
 * 1x1 conv behind the SPPF concat - common.py:112 / :158 `cv2` / `cv5` - then reads one int8 image instead of quantising
 * 4*C fp16 channels on its way in).  q = clamp(rne(y * 127 / amax)) of the fp16 values written to y (the quantiser of
 * y6_conv_i8_desc.q_out); max-pooling commutes with it, so the twin equals quantise-on-load bit for bit.
 * q1,q2,q3: int8 NHWC views (data may be NULL: no twin), channel offsets / strides multiples of 8.                      */
typedef struct y6_sppf_q_desc {
    y6_tensor x, y1, y2, y3;
    y6_tensor q1, q2, q3;
    float q_amax;
    int32_t pad[3];
} y6_sppf_q_desc;
int y6_sppf_pool_q(const y6_sppf_q_desc* d, void* stream);

/* Layout adapters at the module boundary (reference tensors are NCHW). */
int y6_nchw_to_nhwc(const void* src_nchw, int src_dtype, const y6_tensor* dst, void* stream);
int y6_nhwc_to_nchw(const y6_tensor* src, void* dst_nchw, int dst_dtype, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Head epilogue: sigmoid(cls), optional DFL (softmax over reg_max+1 bins . proj),
 * dist2bbox('xywh') * stride, concat -> out[B, A, 5+nc] fp32 (box4, 1.0, cls).
 * Replaces: Detect.forward eval branch  yolov6/models/effidehead.py:93-139,
 *           generate_anchors(is_eval=True)  yolov6/assigners/anchor_generator.py:13-33,
 *           dist2bbox  yolov6/utils/general.py:32-43.
 * Level l: cls[l] view [B,Hl,Wl,nc] logits, reg[l] view [B,Hl,Wl,4*(reg_max+1)] (ltrb bins
 * in the reference's channel order: side-major, bin-minor).                               */
#define Y6_MAX_LEVELS 4
typedef struct y6_decode_desc {
    int32_t n_levels;
    y6_tensor cls[Y6_MAX_LEVELS];
    y6_tensor reg[Y6_MAX_LEVELS];
    float stride[Y6_MAX_LEVELS];
    int32_t use_dfl;           /* 0/1 */
    int32_t reg_max;           /* 0 or 16 */
    const float* proj;         /* [reg_max+1] (Detect.proj) when use_dfl */
    float grid_cell_offset;    /* 0.5 */
    float* out;                /* [B, A, 5+nc] fp32 */
    int32_t nc;
} y6_decode_desc;
int y6_head_decode(const y6_decode_desc* d, void* stream);

/* The same epilogue fused with the two prediction convs in front of it: per level
 *   cls = cls_pred(cls_feat), reg = reg_pred(reg_feat)   (1x1 convs + bias, effidehead.py:100-101)
 * then sigmoid / DFL / dist2bbox / concat as y6_head_decode - ONE launch for all levels, the [B,A,nc+4*(reg_max+1)] fp16
 * logits never reach HBM.  Same rounding points as the unfused ops (conv output fp16, sigmoid fp16, softmax fp16,
 * projection fp16); the result is bit-identical to y6_conv2d x2 per level + y6_head_decode.
 * Replaces: Detect.forward eval branch  yolov6/models/effidehead.py:93-139 from cls_preds / reg_preds on.
 * cls_feat[l] / reg_feat[l]: views [B,Hl,Wl,Cl] (Cl % 16 == 0, 16-byte aligned); w_*: y6_pack_conv_weight images of the
 * [nc,Cl,1,1] / [4*(reg_max+1),Cl,1,1] weights; b_*: fp32 biases (fp16-rounded values).                          */
/* Candidate sink of the fused head tail: with `workspace` set (a y6_nms workspace for the same B, A, nc, multi_label), the decode
 * launch also selects the NMS candidates of its rows (nms.py:48, :69-84, the arithmetic of y6_nms's first stage - same device
 * function) while they sit in LDS and appends their keys to the workspace; y6_nms called with candidates_ready = 1 and the SAME
 * conf_thres / classes / multi_label then starts at the sort and never reads the [B, A, 5 + nc] tensor for the selection
 * (91 MB for YOLOv6-S 640^2 b32).  The tensor itself is still written: it is Detect.forward's result.                        */
typedef struct y6_nms_sink {
    void* workspace;           /* NULL: no sink */
    size_t workspace_bytes;
    float conf_thres;
    const int32_t* classes;    /* optional device list of kept classes */
    int32_t n_classes;
    int32_t multi_label;
} y6_nms_sink;

typedef struct y6_pred_decode_desc {
    int32_t n_levels;
    y6_tensor cls_feat[Y6_MAX_LEVELS];
    y6_tensor reg_feat[Y6_MAX_LEVELS];
    const void* w_cls[Y6_MAX_LEVELS];
    const void* w_reg[Y6_MAX_LEVELS];
    const float* b_cls[Y6_MAX_LEVELS];
    const float* b_reg[Y6_MAX_LEVELS];
    float stride[Y6_MAX_LEVELS];
    int32_t use_dfl;
    int32_t reg_max;
    const float* proj;
    float grid_cell_offset;
    float* out;                /* [B, A, 5+nc] fp32 */
    int32_t nc;
    int32_t first_anchor;      /* with total_anchors > 0: row (inside an image) of `out` where this call's first level starts, */
    int32_t total_anchors;     /* and the rows per image of `out` - a call may then cover a SUBSET of the head's levels (one call per
                                  level lets a level be decoded as soon as its convs are done).  0 / 0: the call covers `out`. */
    y6_nms_sink cand;          /* optional (workspace NULL: off); needs first_anchor / total_anchors = 0 / 0 */
} y6_pred_decode_desc;
int y6_head_pred_decode_supported(const y6_pred_decode_desc* d);   /* 1 if the fused kernel takes this shape */
int y6_head_pred_decode(const y6_pred_decode_desc* d, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Batched NMS.  Replaces: non_max_suppression  yolov6/utils/nms.py:31-105 including the
 * torchvision.ops.nms call at :96 (greedy, IoU > thr strict, fp32, boxes offset by
 * cls*4096 unless agnostic, stable descending score order).
 * pred        : [B, A, 5+nc] fp32 (xywh, obj, cls...)
 * classes     : optional device int32 list of n_classes kept classes (NULL: all)
 * out_dets    : [B, max_det, 6] fp32 (x1,y1,x2,y2,conf,cls); rows >= out_count[b] are written as 0
 * out_index   : [B, max_det] int32, flat candidate id = anchor*nc + cls (parity checks); -1 past the count
 * out_count   : [B] int32 detections per image  (all three outputs are fully written: no pre-fill needed)
 * workspace   : y6_nms_workspace_bytes(B,A,nc,multi_label) bytes of scratch
 * The reference's 10 s wall-clock break (nms.py:56,101-103) has no device analogue and is
 * not reproduced (documented deviation).                                                  */
typedef struct y6_nms_desc {
    const float* pred;
    int32_t B, A, nc;
    float conf_thres, iou_thres;
    const int32_t* classes;
    int32_t n_classes;
    int32_t agnostic, multi_label;
    int32_t max_det;
    int32_t max_nms;           /* 30000 in the reference */
    float max_wh;              /* 4096 in the reference */
    float* out_dets;
    int32_t* out_index;
    int32_t* out_count;
    void* workspace;
    size_t workspace_bytes;
    int32_t candidates_ready;  /* 1: the workspace already holds this call's candidates (y6_nms_sink of the decode launch that
                                  produced `pred`, same conf_thres / classes / multi_label): start at the sort */
} y6_nms_desc;
size_t y6_nms_workspace_bytes(int B, int A, int nc, int multi_label);
int y6_nms(const y6_nms_desc* d, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Task-aligned assigner.  Replaces: TaskAlignedAssigner.forward
 * yolov6/assigners/tal_assigner.py:22-106 (+ get_pos_mask :108-123, get_box_metrics
 * :125-141, select_topk_candidates :143-158, get_targets :160-181) and
 * select_candidates_in_gts / select_highest_overlaps / iou_calculator
 * yolov6/assigners/assigner_utils.py:25-89.  No [B,G,topk,A] one-hot temp is built.
 * pd_scores [B,A,C] f32, pd_bboxes [B,A,4] f32 xyxy, anc_points [A,2] f32,
 * gt_labels [B,G] f32 (class id as float, as the reference passes it), gt_bboxes [B,G,4],
 * mask_gt [B,G] f32 (0/1).
 * outputs: target_labels [B,A] int64, target_bboxes [B,A,4] f32, target_scores [B,A,C] f32,
 * fg_mask [B,A] uint8 (bool).                                                             */
typedef struct y6_tal_desc {
    const float* pd_scores;
    const float* pd_bboxes;
    const float* anc_points;
    const float* gt_labels;
    const float* gt_bboxes;
    const float* mask_gt;
    int32_t B, A, C, G;
    int32_t topk;
    float alpha, beta, eps;
    int64_t* target_labels;
    float* target_bboxes;
    float* target_scores;
    uint8_t* fg_mask;
    void* workspace;
    size_t workspace_bytes;
} y6_tal_desc;
size_t y6_tal_workspace_bytes(int B, int A, int G);
int y6_tal_assign(const y6_tal_desc* d, void* stream);

/* ATSS assigner (warm-up epochs).  Replaces: ATSSAssigner.forward
 * yolov6/assigners/atss_assigner.py:18-86 (+ select_topk_candidates :88-115, thres_calculator
 * :117-136, get_targets :138-161), bbox_overlaps yolov6/assigners/iou2d_calculator.py:63-241 and
 * dist_calculator yolov6/assigners/assigner_utils.py:4-23.
 * anc_bboxes [A,4] f32, n_level_bboxes[n_levels] anchors per level (host array, sums to A),
 * gt_labels [B,G] f32, gt_bboxes [B,G,4], mask_gt [B,G] f32, pd_bboxes [B,A,4] f32 or NULL
 * (soft labels). Outputs as y6_tal_assign; background label = C.                           */
typedef struct y6_atss_desc {
    const float* anc_bboxes;
    int32_t n_level_bboxes[Y6_MAX_LEVELS];
    int32_t n_levels;
    const float* gt_labels;
    const float* gt_bboxes;
    const float* mask_gt;
    const float* pd_bboxes;
    int32_t B, A, C, G;
    int32_t topk;
    int64_t* target_labels;
    float* target_bboxes;
    float* target_scores;
    uint8_t* fg_mask;
    void* workspace;
    size_t workspace_bytes;
} y6_atss_desc;
size_t y6_atss_workspace_bytes(int B, int A, int G);
int y6_atss_assign(const y6_atss_desc* d, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Training-loss forward VALUE.  Replaces, around the assigner call, the arithmetic of
 * ComputeLoss.__call__  yolov6/models/losses/loss.py:52-182:
 *   y6_bbox_decode   bbox_decode :194-198 (softmax over reg_max+1 bins . linspace(0,reg_max), dist2bbox xyxy
 *                    utils/general.py:32-43); anchor_points_s = anchor_points / stride  -> pred_bboxes [B,A,4]
 *   y6_loss_forward  VarifocalLoss :201-211, BboxLoss :214-278 (IOUloss utils/figure_iou.py:7-100 xyxy eps 1e-10,
 *                    DFL :267-278), normalisation by target_scores.sum() when > 1 (:168-169), weights (:171-181)
 * target_* / fg_mask are the outputs of y6_tal_assign / y6_atss_assign (target_bboxes still in pixels: the
 * division by the stride of loss.py:154 happens inside).  out[6] (device, double):
 *   loss, w_iou*loss_iou, w_dfl*loss_dfl, w_class*loss_cls, target_scores_sum, num_pos.
 * Values only - the backward pass is not part of the library yet.                          */
enum { Y6_IOU_GIOU = 0, Y6_IOU_DIOU = 1, Y6_IOU_CIOU = 2, Y6_IOU_SIOU = 3 };
typedef struct y6_loss_desc {
    const float* pred_scores;      /* [B,A,C]  post-sigmoid */
    const float* pred_distri;      /* [B,A,4*(reg_max+1)] DFL logits, or [B,A,4] without DFL */
    const float* pred_bboxes;      /* [B,A,4]  y6_bbox_decode output (stride units) */
    const float* anchor_points_s;  /* [A,2]    anchor points / stride */
    const float* stride;           /* [A] */
    const int64_t* target_labels;  /* [B,A] */
    const float* target_bboxes;    /* [B,A,4]  pixels */
    const float* target_scores;    /* [B,A,C] */
    const uint8_t* fg_mask;        /* [B,A] */
    int32_t B, A, C;
    int32_t use_dfl, reg_max, iou_type;
    float w_class, w_iou, w_dfl;
    double* out;                   /* [6] */
    void* workspace;
    size_t workspace_bytes;
    int32_t box_mode;              /* 0: pred_distri are (l,t,r,b) distances / DFL logits (loss.py:194-198);
                                      1: (dx, dy, w, h) around the anchor point, the anchor-based branch of loss_fuseab.py:75-76
                                         (`pred_distri[..., :2] += anchor_points_s; xywh2xyxy`) - gradient only, use_dfl must be 0 */
    int32_t norm_mode;             /* 0: loss.py's normalisation (terms / target_scores_sum when that is > 1, :168-169, :238-261);
                                      1: the self-distillation losses' (class term when it is > 0, box terms unless it is exactly 0:
                                         loss_distill.py:178-183, :283-330 and loss_distill_ns.py likewise) */
} y6_loss_desc;
int y6_bbox_decode(const float* pred_distri, const float* anchor_points_s, int B, int A, int use_dfl, int reg_max,
                   float* pred_bboxes, void* stream);
size_t y6_loss_workspace_bytes(void);
int y6_loss_forward(const y6_loss_desc* d, void* stream);

/* Self-distillation terms of loss_distill.py / loss_distill_ns.py, as sums the host scales:
 *   acc[0] = sum over ALL anchors of KL(softmax(scores_t / T) || softmax(scores_s / T))      distill_loss_cls :210-221
 *   acc[1] = sum over positive anchors x 4 sides of the same KL over the reg_max + 1 DFL bins   distill_loss_dfl :349-359
 *   acc[2] = sum of the positives' weights (target_scores.sum(-1)),  acc[3] = number of positives
 * (the class "logits" are the post-sigmoid scores of the two heads, as the reference passes them).  y6_distill_backward ADDS
 *   coef[0] * d acc[0] / d scores_s  to dscores  and  coef[1] * d acc[1] / d distri_s  to ddistri
 * (coef: two device floats the host derives from acc - T^2, loss weights, the cosine weight decay, the mean over positives,
 * the incoming gradient - without a host sync).  distri_s / distri_t NULL: class term only.
 * y6_distill_cw: channel-wise feature distillation :222-246 for one level: s_feat / t_feat dense [rows = N*C][hw] fp32; forward
 * (acc given): acc[0] += sum over rows of KL(softmax_hw(t / T) || softmax_hw(s / T)); backward (coef, d_s_feat given):
 * d_s_feat = coef[0] * d acc / d s_feat. */
typedef struct y6_distill_desc {
    const float* scores_s;
    const float* scores_t;
    const float* distri_s;
    const float* distri_t;
    const uint8_t* fg_mask;
    const float* target_scores;
    int32_t BA, C, reg_max;
    float temperature;
    double* acc;
    const float* coef;
    float* dscores;
    float* ddistri;
} y6_distill_desc;
int y6_distill_forward(const y6_distill_desc* d, void* stream);
int y6_distill_backward(const y6_distill_desc* d, void* stream);
int y6_distill_cw(const float* s_feat, const float* t_feat, int rows, int hw, float temperature, double* acc, const float* coef,
                  float* d_s_feat, void* stream);


typedef struct y6_plan y6_plan;

/* ==================================================================================== */
/* TRAINING STEP (config 3): forward in train form with batch-statistics BatchNorm, backward (data + weight
 * gradients), loss gradient, fused SGD.  Replaces what torch autograd does for the reference's
 *   RepVGGBlock.forward train form   yolov6/layers/common.py:250-255
 *   ConvModule.forward               common.py:45-49
 *   Detect.forward training branch   yolov6/models/effidehead.py:72-92
 *   scaler.scale(total_loss).backward() / scaler.step(optimizer)   yolov6/core/engine.py:169-176, :258-266
 * All entry points take plain device pointers; parameters, gradients and BatchNorm buffers are fp32 (master copies),
 * activations and activation gradients NHWC fp16, parameter gradients ACCUMULATE (+=) into fp32 arrays the caller
 * zeroes once per step.  Every op also exists as y6_plan_add_* so a step replays as two native plans.           */

/* tags of training ops inside a plan (y6_plan_op_info reports them in `ksize` for Y6_OP_GENERIC) */
enum { Y6_TOP_BN_STATS = 1, Y6_TOP_BNACT_FWD = 2, Y6_TOP_BNACT_BWD = 3, Y6_TOP_WGRAD_T = 4, Y6_TOP_WGRAD = 5,
       Y6_TOP_PACK = 6, Y6_TOP_POOL_BWD = 7, Y6_TOP_HEAD_PACK = 8, Y6_TOP_HEAD_UNPACK = 9, Y6_TOP_S2D = 10,
       Y6_TOP_BIAS_GRAD = 11, Y6_TOP_FILL = 12, Y6_TOP_ADD = 13,
       Y6_TOP_CONV_I8 = 14, Y6_TOP_ABSMAX = 15, Y6_TOP_QUANT = 16,
       /* fused inference ops (generic plan ops as well) */
       Y6_TOP_PRED_DECODE = 17, Y6_TOP_PW_S2 = 18, Y6_TOP_STEM_S2 = 19,
       Y6_TOP_AVGPOOL3 = 20, Y6_TOP_SPPF_Q = 21, Y6_TOP_DGRAD_S2 = 22 };

/* Batch statistics of a conv output + everything derived from them, on device:
 *   mean, biased var over B*H*W -> invstd = 1/sqrt(var+eps), scale = gamma*invstd, shift = beta - mean*scale;
 *   running_mean/var updated in place with `momentum` (unbiased variance), num_batches_tracked += 1
 * (torch.nn.BatchNorm2d training semantics; eps 1e-3 / momentum 0.03 come from initialize_weights torch_utils.py:38-47). */
typedef struct y6_bn_train_desc {
    y6_tensor x;
    const float* gamma;            /* [C] or NULL (=1) */
    const float* beta;             /* [C] or NULL (=0) */
    float* running_mean;           /* [C] or NULL */
    float* running_var;
    int64_t* num_batches_tracked;  /* scalar or NULL */
    float momentum, eps;
    float* scale;                  /* outputs, [C] each */
    float* shift;
    float* mean;
    float* invstd;
    void* workspace;               /* y6_bn_stats_workspace_bytes_for(C, B*H*W) */
    size_t workspace_bytes;
    int32_t workspace_clean;       /* nonzero: the workspace is all-zero on entry (allocated zeroed, used by these calls only - they
                                      leave it zeroed); no memset launch.  0: a memset is issued first. */
} y6_bn_train_desc;
size_t y6_bn_stats_workspace_bytes(int C);                 /* worst case (1024 partial blocks) */
size_t y6_bn_stats_workspace_bytes_for(int C, long npix);   /* what a tensor of npix = B*H*W pixels needs */
int y6_bn_train_stats(const y6_bn_train_desc* d, void* stream);
/* The statistics of up to three tensors in one pair of launches: the raw outputs of a RepVGG block's 3x3 / 1x1 branches and its
 * identity input (yolov6/layers/common.py:250-255 normalises all three before the sum).  Entries of one shape share the launches
 * (same blocks and addition order as y6_bn_train_stats: same bits); otherwise the entries run one after the other.  The entries
 * must not share outputs, workspaces or running statistics. */
typedef struct y6_bn_train_multi_desc {
    int32_t n;                     /* 1..3 */
    y6_bn_train_desc d[3];
} y6_bn_train_multi_desc;
int y6_bn_train_stats_multi(const y6_bn_train_multi_desc* d, void* stream);

/* out = act( sum_b ( x_b * scale_b[c] + shift_b[c] ) ) [+ alpha * res]   (1..3 branches; NULL scale = 1, NULL shift = 0)
 * - the RepVGG train-form sum (common.py:250-255), ConvModule's BN + act (:45-49), QARepVGG's raw branches (:341-347),
 * BottleRep's shortcut (:605-608). */
typedef struct y6_bnact_desc {
    int32_t n;
    y6_tensor x[3];
    const float* scale[3];
    const float* shift[3];
    y6_tensor res;                 /* data == NULL: none */
    const float* res_alpha;        /* device scalar or NULL (= 1) */
    y6_tensor out;
    int32_t act;
} y6_bnact_desc;
int y6_bnact_forward(const y6_bnact_desc* d, void* stream);

/* Backward of y6_bnact_forward + the BatchNorms that fed it.  With dz = dout * act'(z):
 *   branch with BN:   dx_b = gamma_b*invstd_b * ( dz - mean(dz) - xhat_b * mean(dz*xhat_b) ),  dgamma_b += sum dz*xhat_b,
 *                     dbeta_b += sum dz      (xhat_b = (x_b - mean_b)*invstd_b, batch statistics: torch's batch_norm_backward)
 *   branch without:   dx_b = dz * scale_b (scale NULL = 1)
 *   shortcut:         dres (+)= alpha*dout,  dalpha += sum dout*res
 * z is recomputed from the saved branch inputs.  dx_b views may be `dilated` (dx_dil = 2: element (y,x) is written at
 * (2y,2x) of a pre-zeroed buffer - the zero-inserted gradient a stride-2 conv's data gradient convolves) and may
 * accumulate (+=). */
typedef struct y6_bnact_bwd_desc {
    y6_bnact_desc fwd;             /* the forward op's operands (out unused) */
    const float* mean[3];          /* NULL: branch b has no BatchNorm */
    const float* invstd[3];
    const float* gamma[3];         /* NULL = 1 */
    y6_tensor dout;
    y6_tensor dx[3];               /* data == NULL: not needed */
    int32_t dx_dil[3];             /* 1 or 2 */
    int32_t dx_acc[3];             /* 1: += */
    float* dgamma[3];              /* += ; NULL: skip */
    float* dbeta[3];
    y6_tensor dres;                /* data == NULL: none */
    int32_t dres_acc;
    float* dalpha;                 /* += ; NULL: skip */
    void* workspace;               /* y6_bnact_bwd_workspace_bytes_for(C, B*H*W) */
    size_t workspace_bytes;
    int32_t workspace_clean;       /* as in y6_bn_train_desc: all-zero on entry, left all-zero */
} y6_bnact_bwd_desc;
size_t y6_bnact_bwd_workspace_bytes(int C);
size_t y6_bnact_bwd_workspace_bytes_for(int C, long npix);
int y6_bnact_backward(const y6_bnact_bwd_desc* d, void* stream);

/* Pixel-run-major ("transposed") sampling of an activation for the weight-gradient GEMM:
 *   dst[b][r][q/8][c][q%8] = src(b, r*sy + oy, q*sx + ox, c)   (0 outside the image),  r < R, q < Q (Q % 16 == 0)
 * - 8 consecutive columns of a channel are one 16-byte MFMA operand, the operands of consecutive channels are contiguous.
 * src: an NHWC fp16 view, or (nchw != 0) the caller's NCHW image tensor (fp16 / fp32) - the stem's input. */
typedef struct y6_wgrad_t_desc {
    y6_tensor src;                 /* NHWC view; for nchw: data = NCHW base, C/H/W/B filled, cstride/coff ignored */
    int32_t nchw, src_dtype;       /* src_dtype: Y6_F16 / Y6_F32 (nchw only) */
    int32_t sy, sx, oy, ox, R, Q;
    void* dst;                     /* fp16 [B][R][Q/8][C][8] */
} y6_wgrad_t_desc;
int y6_wgrad_transpose(const y6_wgrad_t_desc* d, void* stream);

/* Weight gradient as a tap-table GEMM over pixels on the matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate):
 *   out[m*sm + n*sn + t*st] += sum_{b, y<rows, q<Q} A(m; b, y, q) * P_t(n; b, y + drow_t, q + shift_t)
 * A and the planes P are y6_wgrad_transpose outputs (q contiguous: both operands are read straight from memory as MFMA
 * fragments, the +-1 column shifts are built in registers; slices of the pixel range are summed by a second small kernel,
 * deterministically, without atomics).  Replaces the weight half of autograd's conv backward for
 *   3x3 s1 (mode 0: one plane with a zero row above/below, 9 taps), 1x1 (mode 1), 3x3 s2 (mode 2: four row/column parity
 *   planes), ConvTranspose2d k2 s2 (mode 3: A = input, planes = the four parities of dout).                          */
enum { Y6_WG_3X3S1 = 0, Y6_WG_1X1 = 1, Y6_WG_3X3S2 = 2, Y6_WG_CONVT = 3 };
typedef struct y6_wgrad_desc {
    int32_t mode;
    const void* a;                 /* y6_wgrad_transpose output with a_channels channels, a_rows rows */
    int32_t M, N, B, Q, rows, a_rows;
    int32_t a_channels, plane_channels;   /* channels the A plane / the B planes were transposed with (>= M / N) */
    const void* plane[6];          /* per stream (mode table in wgrad.hip): y6_wgrad_transpose outputs with plane_rows[s] rows */
    int32_t plane_rows[6];
    int32_t drow[6];
    float* out;
    int32_t sm, sn, st;            /* element strides of the output (OIHW: sm = N*T, sn = T, st = 1) */
    double flops;                  /* algorithmic FLOPs (2*M*N*T*B*Ho*Wo), for the timing table */
    void* workspace;               /* partial sums of the (image,row) slices: >= T*M*N*4 bytes, more allows more slices */
    size_t workspace_bytes;        /* (ops on one stream may share one workspace)                                       */
} y6_wgrad_desc;
int y6_wgrad(const y6_wgrad_desc* d, void* stream);

/* The same weight gradient for 3x3 stride-1 (pad 1) and 1x1 stride-1 convs read STRAIGHT from the NHWC tensors - no
 * y6_wgrad_transpose copies: rows arrive in LDS by DMA and leave it through gfx950's transposing LDS read
 * (ds_read_b64_tr_b16) as the MFMA's pixel-major operands.
 *   out[m*sm + n*sn + t*st] += sum_{b,y,x} dy(b,y,x,m) * x(b, y+ky-1, x+kx-1, n),  t = ky*3 + kx   (1x1: t = 0, no offsets)
 * dy may be wider than M (8-channel padded prediction-conv gradients): channels M.. of the view must then be zero.
 * y6_wgrad_nhwc_supported: 1 when the shapes fit (W <= 208 for 3x3, <= 320 for 1x1; 8-channel aligned fp16 views). */
typedef struct y6_wgrad_nhwc_desc {
    int32_t ksize;                 /* 3 or 1 */
    y6_tensor dy;                  /* [B, H, W, >= M] */
    y6_tensor x;                   /* [B, H, W, >= N] */
    int32_t M, N;
    float* out;
    int32_t sm, sn, st;
    int32_t stride;                /* 0 / 1: stride 1 (dy and x have one spatial shape); 2: x is [B, H, W, .] and dy either
                                      [B, (H-1)/2+1, (W-1)/2+1, .] or the zero-inserted gradient [B, H, W, .] the stride-2 data gradient
                                      reads (values at (2y, 2x)); 3x3 pad 1 or 1x1 pad 0 - the flat-index kernel only (round 6; the
                                      field fills the struct's former padding) */
    double flops;
    void* workspace;               /* as y6_wgrad_desc */
    size_t workspace_bytes;
} y6_wgrad_nhwc_desc;
int y6_wgrad_nhwc_supported(const y6_wgrad_nhwc_desc* d);
int y6_wgrad_nhwc(const y6_wgrad_nhwc_desc* d, void* stream);
/* Planning queries, host arithmetic only (they also run on a machine without a GPU; tests/test_host_cpu.py property-tests them):
 * which kernel a descriptor gets - 0 none (keep y6_wgrad), 1 the flat-index kernel (csrc/wgrad_flat.hip), 2 the row ring - and,
 * for the flat kernel, its geometry: the padded flat index (row pitch W+1, plane (H+1)(W+1) over the OUTPUT grid; 1x1: W, H*W),
 * chunk size and count, block tile, stage images, LDS, slices of the flat range and the bytes of partial tiles they leave. */
typedef struct y6_wgrad_flat_geom {
    int32_t row_pitch, plane, flat_positions;
    int32_t chunk, chunks;
    int32_t tile_m, tile_n, tiles;
    int32_t x_positions, stages;
    int32_t slices, chunks_per_slice;
    uint64_t lds_bytes, partial_bytes;
} y6_wgrad_flat_geom;
int y6_wgrad_nhwc_route(const y6_wgrad_nhwc_desc* d);
int y6_wgrad_flat_geometry(const y6_wgrad_nhwc_desc* d, y6_wgrad_flat_geom* out);

/* The weight gradients of the convs that read the caller's NCHW image - the stem block of EfficientRep / CSPBepBackbone in train form
 * (yolov6/models/efficientrep.py:28-41: RepVGGBlock(3 -> C, k3 s2), forward yolov6/layers/common.py:250-255) - in one pass over the
 * image and the two gradients (csrc/wgrad_stem.hip):
 *   out3[m][c][ky][kx] += sum_{b,y,x} dy3(b,y,x,m) * x(b, c, 2y+ky-1, 2x+kx-1)     3x3 stride 2 pad 1
 *   out1[m][c]         += sum_{b,y,x} dy1(b,y,x,m) * x(b, c, 2y, 2x)               1x1 stride 2 (dy1.data == NULL: none)
 * x: contiguous fp16 NCHW [B][Cin <= 3][H][W], H and W even; dy3 / dy1: compact NHWC views [B, H/2, W/2, >= Cout]; out3 / out1:
 * contiguous fp32 OIHW, accumulated into.  Deterministic (block partials added in a fixed order).
 * y6_wgrad_stem_supported: 1 when the descriptor fits (otherwise the plane-fed route of y6_wgrad serves). */
typedef struct y6_wgrad_stem_desc {
    const void* x;
    int32_t in_dtype;              /* Y6_F16 */
    int32_t B, Cin, H, W;
    int32_t Cout;
    y6_tensor dy3;
    y6_tensor dy1;
    float* out3;
    float* out1;
    void* workspace;               /* y6_wgrad_stem_workspace_bytes(Cout) */
    size_t workspace_bytes;
} y6_wgrad_stem_desc;
size_t y6_wgrad_stem_workspace_bytes(int Cout);
int y6_wgrad_stem_supported(const y6_wgrad_stem_desc* d);
int y6_wgrad_stem(const y6_wgrad_stem_desc* d, void* stream);

/* Per-step weight preparation: every packed fp16 MFMA weight image the step's convs read is rebuilt from the fp32
 * master parameters by ONE launch over a device job table.
 *   kind 0: forward conv    dst = pack(W[Cout][Cin][K][K])
 *   kind 1: data gradient   dst = pack(W'[Cin][Cout][K][K]),  W'[ci][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx]
 *   kind 2: ConvTranspose2d forward (IOHW [Cin][Cout][2][2], the fused four-sub-kernel image of y6_pack_convt2x2_weight)
 *   kind 3: ConvTranspose2d data gradient as a 1x1 conv over space-to-depth(dout): W'[ci][sub*Cout+co] = W[ci][co][sub] */
typedef struct y6_pack_job {
    const float* src;
    void* dst;
    int32_t kind, Cout, Cin, K;
    uint64_t first;                /* index of this job's first packed element in the concatenated work list */
} y6_pack_job;
typedef struct y6_pack_batch_desc {
    const y6_pack_job* jobs;       /* DEVICE array */
    int32_t njobs;
    uint64_t total;                /* packed elements over all jobs */
} y6_pack_batch_desc;
size_t y6_pack_job_elems(int kind, int Cout, int Cin, int K);
int y6_pack_weights_batched(const y6_pack_batch_desc* d, void* stream);

/* Backward of the three chained 5x5 max-pools of SPPFModule / CSPSPPFModule (common.py:106-112, :150-158):
 * gradients flow to the FIRST maximum of each window in row-major order (torch max_pool2d_with_indices).
 *   dx (+)= dcat0 + bwd(x->y1, dy1 + bwd(y1->y2, dy2 + bwd(y2->y3, dy3))) */
typedef struct y6_sppf_bwd_desc {
    y6_tensor x, y1, y2;           /* forward tensors (y3 is not needed) */
    y6_tensor dy1, dy2, dy3;       /* gradients wrt the three pooled slices */
    y6_tensor dx;                  /* gradient wrt x (the cv1 / cv4 output slice): receives the sum */
    int32_t dx_acc;                /* 1: dx already holds the gradient that reached x directly (concat slice 0) */
} y6_sppf_bwd_desc;
int y6_sppf_pool_backward(const y6_sppf_bwd_desc* d, void* stream);

/* Data gradient of a 3x3 stride-2 conv - and of the 1x1 stride-2 conv that reads the same input (RepVGGBlock(k3 s2) in train
 * form, common.py:250-255; ConvBNReLU / ConvBNSiLU(k3 s2), common.py:26-94) - from the COMPACT output gradients: the input half of
 * autograd's conv backward under scaler.scale(loss).backward() (core/engine.py:173).  Four parity classes of dx, nine (ten) tap
 * GEMMs over dy (csrc/dgrad_s2.hip); every element of dx is written once, or added to what dx holds (`accumulate`, the two
 * roundings of the accumulating convs).  M = dy3.C and N = dx.C multiples of 32, dx = [B, 2*Ho, 2*Wo, N], 16-byte aligned views.
 *   w3_packed / w1_packed: the data-gradient images of the 3x3 / 1x1 weight (y6_pack_job kind 1 with K = 3 / 1). */
typedef struct y6_dgrad_s2_desc {
    y6_tensor dy3;                 /* [B, Ho, Wo, M] gradient of the 3x3 conv's output */
    y6_tensor dy1;                 /* gradient of the 1x1 conv's output, same shape (data == NULL: no 1x1 branch) */
    y6_tensor dx;                  /* [B, 2*Ho, 2*Wo, N] */
    const void* w3_packed;
    const void* w1_packed;         /* NULL without dy1 */
    int32_t accumulate;            /* 1: dx += ... */
} y6_dgrad_s2_desc;
int y6_dgrad_s2_supported(const y6_dgrad_s2_desc* d);
int y6_dgrad_s2(const y6_dgrad_s2_desc* d, void* stream);

/* Detect training branch (effidehead.py:72-92): per-level NHWC prediction maps -> cls_scores [B,A,nc] = sigmoid(logits),
 * reg_distri [B,A,nreg], both fp32 (the loss computes in fp32, loss.py:208); and its backward
 * dlogit = dscore * p * (1-p), dreg passes through, written as NHWC fp16 per level. */
typedef struct y6_head_pack_desc {
    int32_t n_levels;
    y6_tensor cls[4], reg[4];      /* forward: inputs; backward: gradient outputs */
    float* scores;                 /* [B,A,nc]   forward: out;  backward: in (the saved probabilities) */
    float* distri;                 /* [B,A,nreg] forward: out;  backward: unused */
    const float* dscores;          /* backward only */
    const float* ddistri;
    int32_t nc, nreg;
} y6_head_pack_desc;
int y6_head_pack(const y6_head_pack_desc* d, void* stream);
int y6_head_unpack_backward(const y6_head_pack_desc* d, void* stream);

/* fuse_ab head, anchor-based auxiliary branch of the training step (yolov6/models/heads/effidehead_fuseab.py:110-124):
 * per level cls_ab [B,H,W,na*nc] -> sigmoid -> scores [B, A_ab, nc] with anchors ordered (level, anchor, pixel);
 * reg_ab [B,H,W,na*4] -> (dx, dy, (2 sigmoid(w))^2 * aw, (2 sigmoid(h))^2 * ah) -> distri [B, A_ab, 4]; and the backward
 * (cls / reg then hold the GRADIENT maps; reg_fwd the forward reg maps the box transform's derivative needs).
 * anchors: HOST values [level][3][2] = anchors_init / stride. */
typedef struct y6_head_ab_desc {
    int32_t n_levels, nc, na;
    y6_tensor cls[4], reg[4];
    y6_tensor reg_fwd[4];          /* backward only */
    float anchors[24];
    float* scores;                 /* [B, A_ab, nc] */
    float* distri;                 /* [B, A_ab, 4] */
    const float* dscores;          /* backward only */
    const float* ddistri;
} y6_head_ab_desc;
int y6_head_ab_pack(const y6_head_ab_desc* d, void* stream);
int y6_head_ab_unpack_backward(const y6_head_ab_desc* d, void* stream);

/* space-to-depth for ConvTranspose2d's data gradient: dst[b,y,x,sub*C + c] = src[b,2y+dy,2x+dx,c], sub = dy*2+dx */
int y6_space_to_depth2(const y6_tensor* src, const y6_tensor* dst, void* stream);
/* dst[b,y,x,:] = src[b,2y,2x,:] - the input sampling of a 1x1 stride-2 conv (RepVGG's rbr_1x1 in the stride-2 blocks,
 * common.py:243), which then runs as a stride-1 GEMM */
int y6_subsample2(const y6_tensor* src, const y6_tensor* dst, void* stream);
/* per-channel sum of an NHWC fp16 view, accumulated into fp32: bias gradients of ConvTranspose2d / prediction convs */
int y6_channel_sum(const y6_tensor* x, float* out_accum, void* workspace, size_t workspace_bytes, void* stream);
/* dst (=|+=) a  for NHWC fp16 views of one shape (gradient fan-in of tensors with several consumers) */
int y6_tensor_add(const y6_tensor* a, const y6_tensor* dst, int accumulate, void* stream);
/* dst (=|+=) [src +] AvgPool2d(kernel 3, stride 1, padding 1)(src) for NHWC fp16 views of one shape: the `rbr_avg` branch of
 * QARepVGGBlockV2's training form merged with its raw identity branch (yolov6/layers/common.py:404, :416-419:
 * `self.rbr_dense(x) + self.rbr_1x1(x) + id_out + self.rbr_avg(x)`); the pooling is its own adjoint, so the backward of the
 * branch is the same op applied to the gradient with accumulate = 1. */
int y6_avgpool3(const y6_tensor* src, const y6_tensor* dst, int with_identity, int accumulate, void* stream);

/* Training loss WITH gradient: y6_loss_forward's value plus d loss / d pred_scores and d loss / d pred_distri
 * (through bbox_decode's DFL projection, dist2bbox, the IoU loss, the DFL cross entropies and VarifocalLoss including
 * its prediction-dependent weight), multiplied by *grad_scale (device scalar, NULL = 1: the GradScaler's loss scale). */
typedef struct y6_loss_grad_desc {
    y6_loss_desc fwd;
    const float* grad_scale;
    float* dpred_scores;           /* [B,A,C] */
    float* dpred_distri;           /* [B,A,4*(reg_max+1)] or [B,A,4] */
} y6_loss_grad_desc;
int y6_loss_forward_backward(const y6_loss_grad_desc* d, void* stream);
/* gradient only: d->fwd.out must hold the result of y6_loss_forward on the same operands */
int y6_loss_backward(const y6_loss_grad_desc* d, void* stream);

/* SGD with (Nesterov) momentum over a flat fp32 arena, unscaling and overflow handling fused
 * (torch.optim.SGD as configured by yolov6/solver/build.py:10-30 + GradScaler.step/update, engine.py:258-266):
 *   y6_grad_finite_check  *found_inf = 1 if any gradient is inf/nan
 *   y6_sgd_step           if !*found_inf:  g = grad * (1/ *scale) + wd*p;  buf = first ? g : mom*buf + g;
 *                                          p -= lr * (nesterov ? g + mom*buf : buf)
 *   y6_scaler_update      scale <- scale*backoff on overflow, *growth after `interval` clean steps; found_inf <- 0 */
int y6_grad_finite_check(const float* grad, size_t n, int32_t* found_inf, void* stream);
int y6_sgd_step(float* param, const float* grad, float* momentum_buf, size_t n, float lr, float momentum, float weight_decay,
                int nesterov, int first_step, const float* scale, const int32_t* found_inf, void* stream);
/* the same step over an arena that mixes the reference's three parameter groups (yolov6/solver/build.py:12-29):
 * group[i] = 0 BatchNorm weight, 1 conv / convT weight (weight decay), 2 bias, >= 3 not optimised;  lr3 / wd3: HOST arrays
 * of three values;  grad_mul multiplies every gradient (1 / world size: the DDP average folded into the step) */
int y6_sgd_step_grouped(float* param, const float* grad, float* momentum_buf, const uint8_t* group, size_t n, const float* lr3_host,
                        const float* wd3_host, float momentum, int nesterov, int first_step, float grad_mul, const float* scale,
                        const int32_t* found_inf, void* stream);
int y6_scaler_update(float* scale, int32_t* found_inf, int32_t* growth_tracker, float growth, float backoff, int interval,
                     void* stream);

/* plan builders for the ops above */
int y6_plan_add_bn_train_stats(y6_plan* p, const y6_bn_train_desc* d);
int y6_plan_add_bn_train_stats_multi(y6_plan* p, const y6_bn_train_multi_desc* d);
int y6_plan_add_bnact_forward(y6_plan* p, const y6_bnact_desc* d);
int y6_plan_add_bnact_backward(y6_plan* p, const y6_bnact_bwd_desc* d);
int y6_plan_add_wgrad_transpose(y6_plan* p, const y6_wgrad_t_desc* d);
int y6_plan_add_wgrad(y6_plan* p, const y6_wgrad_desc* d);
int y6_plan_add_wgrad_nhwc(y6_plan* p, const y6_wgrad_nhwc_desc* d);
int y6_plan_add_wgrad_stem(y6_plan* p, const y6_wgrad_stem_desc* d);
int y6_plan_add_pack_batch(y6_plan* p, const y6_pack_batch_desc* d);
int y6_plan_add_sppf_backward(y6_plan* p, const y6_sppf_bwd_desc* d);
int y6_plan_add_dgrad_s2(y6_plan* p, const y6_dgrad_s2_desc* d);         /* generic op, tag Y6_TOP_DGRAD_S2 */
int y6_plan_add_head_pack(y6_plan* p, const y6_head_pack_desc* d);
int y6_plan_add_head_unpack_backward(y6_plan* p, const y6_head_pack_desc* d);
int y6_plan_add_head_ab_pack(y6_plan* p, const y6_head_ab_desc* d);
int y6_plan_add_head_ab_unpack_backward(y6_plan* p, const y6_head_ab_desc* d);
int y6_plan_add_space_to_depth2(y6_plan* p, const y6_tensor* src, const y6_tensor* dst);
int y6_plan_add_subsample2(y6_plan* p, const y6_tensor* src, const y6_tensor* dst);
int y6_plan_add_channel_sum(y6_plan* p, const y6_tensor* x, float* out_accum, void* workspace, size_t workspace_bytes);
int y6_plan_add_tensor_add(y6_plan* p, const y6_tensor* a, const y6_tensor* dst, int accumulate);
int y6_plan_add_avgpool3(y6_plan* p, const y6_tensor* src, const y6_tensor* dst, int with_identity, int accumulate);
int y6_plan_add_fill_zero(y6_plan* p, void* ptr, size_t bytes);

/* ------------------------------------------------------------------------------------ */
/* Execution plan: an ordered list of the ops above with fixed pointers, replayed with one
 * call per forward (optionally from a captured hipGraph).  This is the native executor
 * behind Model.forward  yolov6/models/yolo.py:33-41.                                      */
enum { Y6_OP_CONV = 1, Y6_OP_CONVT = 2, Y6_OP_STEM = 3, Y6_OP_SPPF = 4, Y6_OP_DECODE = 5,
       Y6_OP_NCHW2NHWC = 6, Y6_OP_NHWC2NCHW = 7, Y6_OP_GENERIC = 8 /* training ops: tag = Y6_TOP_* */ };

y6_plan* y6_plan_create(void);
void y6_plan_destroy(y6_plan* p);
int y6_plan_add_conv(y6_plan* p, const y6_conv_desc* d);
int y6_plan_add_convt(y6_plan* p, const y6_convt_desc* d);
int y6_plan_add_conv_i8(y6_plan* p, const y6_conv_i8_desc* d);            /* generic op, tag Y6_TOP_CONV_I8 */
int y6_plan_add_absmax(y6_plan* p, const y6_tensor* x, float* out);      /* calibration plans */
int y6_plan_add_quantize_i8(y6_plan* p, const y6_tensor* x, float amax, const y6_tensor* q);
int y6_plan_add_stem(y6_plan* p, const y6_stem_desc* d);
int y6_plan_add_sppf(y6_plan* p, const y6_tensor* x, const y6_tensor* y1, const y6_tensor* y2, const y6_tensor* y3);
int y6_plan_add_decode(y6_plan* p, const y6_decode_desc* d);
int y6_plan_add_pw_s2(y6_plan* p, const y6_pw_s2_desc* d);       /* generic op, tag Y6_TOP_PW_S2 */
int y6_plan_add_sppf_q(y6_plan* p, const y6_sppf_q_desc* d);     /* generic op, tag Y6_TOP_SPPF_Q */
int y6_plan_add_stem_s2(y6_plan* p, const y6_stem_s2_desc* d);   /* generic op, tag Y6_TOP_STEM_S2; its image pointer is a rebindable input */
int y6_plan_add_pred_decode(y6_plan* p, const y6_pred_decode_desc* d);   /* generic op, tag Y6_TOP_PRED_DECODE; its `out` is rebindable */
/* Attach (sink->workspace set) or detach (NULL workspace) the candidate sink of the plan's fused head-tail op(s).  Returns the
 * number of ops changed (0: the plan has no such op, or the op cannot serve this sink - the caller keeps y6_nms's own first
 * stage), negative on error. */
int y6_plan_set_nms_sink(y6_plan* p, const y6_nms_sink* sink);
/* Side-stream ops (y6_plan_mark_side) enqueued since the plan's side stream last joined the caller's stream: 0 after every
 * y6_plan_run / y6_plan_run_range (they join before they return). */
int y6_plan_side_pending(const y6_plan* p);
int y6_plan_add_nchw2nhwc(y6_plan* p, const void* src, int src_dtype, const y6_tensor* dst);
int y6_plan_add_nhwc2nchw(y6_plan* p, const y6_tensor* src, void* dst, int dst_dtype);
int y6_plan_num_ops(const y6_plan* p);
/* Choose the conv kernel variant of every conv op by timing (hipEvents on `stream`).  First every supported variant of every
 * conv op by itself (`iters` timed launches each, right behind its predecessor in the plan), then the WHOLE STEP: starting from the
 * better of {per-layer winners, the shape-derived table}, layer groups (equal signature) are switched to their other near-best
 * variants and a switch is kept when the step as a whole gets faster (the first launch of a kernel function that has not run for a
 * while costs 20-35 us: a table that hops between functions loses what the per-layer times promise).  Replaces the reference's
 * reliance on cuDNN's own algorithm search (torch.backends.cudnn.benchmark, tools/train.py:97).  Synchronises the stream.
 * Env: Y6_AUTOTUNE_MODE=layer (per-layer winners only), Y6_AUTOTUNE_CACHE, Y6_AUTOTUNE_LOG, Y6_AUTOTUNE_EXCLUDE. */
int y6_plan_autotune(y6_plan* p, void* stream, int iters);
/* Give `dst` the conv kernel choices of `src`: two plans lowered from the same module for the same input shapes (the slots of
 * yolov6_amd/pipeline.py's InflightRunner) run the same kernels without tuning twice.  Error if the plans differ in ops / shapes. */
int y6_plan_copy_variants(y6_plan* dst, const y6_plan* src);
/* Re-point every op that reads the caller's NCHW boundary tensor at `old_ptr` to `new_ptr`
 * (same shape/dtype). Returns the number of fields changed (>=0) or a negative error.
 * Invalidates a captured graph. */
int y6_plan_rebind(y6_plan* p, const void* old_ptr, const void* new_ptr);
/* Re-point the `index`-th boundary-reading op (stem / NCHW->NHWC adapter ops, in plan order) at `new_ptr`.
 * Rebinding by position is safe when the caller permutes its input tensors. Returns 1 if changed, 0 if equal. */
int y6_plan_rebind_input(y6_plan* p, int index, const void* new_ptr);
/* Re-point every op that writes the boundary tensor at `old_ptr` (the decode output of Detect.forward, effidehead.py:124-139,
 * or an NHWC->NCHW adapter's destination) to `new_ptr` (same shape/dtype): `Model.forward` (models/yolo.py:33-41) returns its
 * result without the copy by alternating between output buffers.  Returns the number of fields changed or a negative error. */
int y6_plan_rebind_output(y6_plan* p, const void* old_ptr, void* new_ptr);
/* Launch all ops in order on `stream` (no sync). */
int y6_plan_run(y6_plan* p, void* stream);
/* Eager launch of ops [first, last) only (teacher-forced per-layer parity tests, partial re-runs). */
int y6_plan_run_range(y6_plan* p, void* stream, int first, int last);
/* Marks the op added last as SIDE-STREAM work for eager runs (y6_plan_run / y6_plan_run_range; Y6_SIDE_STREAM=0 in the
 * environment keeps everything on one stream): it is ordered behind every op before it in plan order, nothing on the main stream waits for it before the end of
 * the run / range, where the side stream is joined.  The caller guarantees that no later op of the run writes what it reads or
 * touches what it writes (the training graph marks weight-gradient work: it feeds only the optimizer step - the reference's
 * autograd engine orders `conv2d_backward`'s weight and input gradients the same way, yolov6/core/engine.py:161-166). */
int y6_plan_mark_side(y6_plan* p);
/* Two-stream schedule of whole-plan eager runs (y6_plan_run only; ranges, timed runs and captured graphs keep plan order on one
 * stream).  order[n]: the ops in the order they are enqueued (a permutation); stream[n], by op index: 0 = the caller's stream,
 * 1 = the plan's side stream; edges[2 * nedges]: (src, dst) pairs on different streams, src enqueued before dst: dst waits for
 * an event recorded behind src.  A run forks the side stream off the caller's stream first and joins it back last.  The caller
 * (yolov6_amd/schedule.py: the ops off the critical path of the forward - BiFusion's lateral convs yolov6/layers/common.py:
 * 695-718, the CSP-SPPF bypass :135-158, the head levels that finish early yolov6/models/effidehead.py:93-139 - as early as
 * their inputs allow) guarantees that every data dependence between ops on different streams is covered by an edge.
 * n = 0 drops the schedule. */
int y6_plan_set_schedule(y6_plan* p, const int32_t* order, const int32_t* stream, int n, const int32_t* edges, int nedges);
/* Live per-op timing: reserve `slots` runs worth of hipEvents, run eagerly with an event between
 * consecutive ops (on `stream`), then - after the caller synchronised - read the per-op sums. */
int y6_plan_timing_begin(y6_plan* p, int slots);
int y6_plan_run_timed(y6_plan* p, void* stream);
int y6_plan_timing_read(y6_plan* p, float* ms_sum, int cap);  /* returns the number of slots summed */
/* kind / conv variant / ksize / stride and algorithmic FLOPs + bytes of op i */
int y6_plan_op_info(const y6_plan* p, int i, int32_t* kind, int32_t* variant, int32_t* ksize, int32_t* stride,
                    double* flops, double* bytes);
/* Capture the op list into a hipGraph once; later y6_plan_run replays the graph. */
int y6_plan_capture(y6_plan* p, void* stream);
/* Per-op profile: runs the plan op by op with hipEvents, `iters` times; fills ms[i] (average
 * milliseconds of op i), kind[i], variant[i] and flops[i]/bytes[i] (algorithmic) for up to
 * `cap` ops.  Returns the number of ops. Synchronises. */
int y6_plan_profile(y6_plan* p, void* stream, int iters, float* ms, int32_t* kind, int32_t* variant,
                    double* flops, double* bytes, int cap);

#ifdef __cplusplus
}
#endif
#endif /* YOLOV6_HIP_H */
