// stem_piece.hpp - one 16-byte-aligned piece (8 consecutive pixels of one NCHW row) of the caller's image, in the three
// dtypes the image conv accepts; uint8 pixels become the fp16 values of `imgs.half() / 255` (core/evaler.py:121-123).
// Shared by the stem kernels (conv_misc.hip) and the fused stem + stride-2 kernel (conv_fused.hip).
#pragma once
#include "common.hpp"

namespace {

template <typename TI>
struct StemPiece;
template <>
struct StemPiece<__half> {
    uint4 v;
    __device__ __forceinline__ void load(const __half* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void zero() { v = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ uint4 as_half8() const { return v; }
};
template <>
struct StemPiece<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void zero() { a = b = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ uint4 as_half8() const {
        h8_t h = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w,
                  (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
        return *reinterpret_cast<const uint4*>(&h);
    }
};

template <>
struct StemPiece<uint8_t> {      // 8 pixels = 8 bytes; converted to the fp16 values of `imgs.half() / 255`
    uint2 v;
    __device__ __forceinline__ void load(const uint8_t* p) { v = *reinterpret_cast<const uint2*>(p); }
    __device__ __forceinline__ void zero() { v = make_uint2(0u, 0u); }
    __device__ __forceinline__ uint4 as_half8() const {
        h8_t h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned b = ((j < 4 ? v.x : v.y) >> (8 * (j & 3))) & 0xffu;
            h[j] = (_Float16)((float)b / 255.f);
        }
        return *reinterpret_cast<const uint4*>(&h);
    }
};

}  // namespace
