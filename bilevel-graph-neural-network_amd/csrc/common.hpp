// Shared host/device helpers for libblgnn (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/blgnn.h"

namespace blgnn {

// ---- error plumbing --------------------------------------------------------
char *err_buf();                       // thread-local, 512 bytes
int fail(int code, const char *fmt, ...);

#define BLGNN_CHECK_ARG(cond, ...)                                   \
    do {                                                             \
        if (!(cond)) return ::blgnn::fail(BLGNN_ERR_ARG, __VA_ARGS__); \
    } while (0)

inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(BLGNN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return BLGNN_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers --------------------------------------------------------
constexpr int WAVE = 64;

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4_fma(float4 &a, float w, const float4 &v) {
    a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y);
    a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w);
}
__device__ __forceinline__ void f4_add(float4 &a, const float4 &v) {
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case BLGNN_ACT_RELU:    return v > 0.f ? v : 0.f;
        case BLGNN_ACT_PRELU:   return v > 0.f ? v : slope * v;
        case BLGNN_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case BLGNN_ACT_TANH:    return tanhf(v);
        default:                return v;
    }
}

// reduce over a power-of-two group of `width` adjacent lanes (width <= 64)
template <typename T>
__device__ __forceinline__ T group_sum(T v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
template <typename T>
__device__ __forceinline__ T group_max(T v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) {
        T t = __shfl_xor(v, o, WAVE);
        v = t > v ? t : v;
    }
    return v;
}

// The same reductions for a COMPILE-TIME group of W adjacent lanes (W = 2 .. 64, aligned), every lane gets the result.
// __shfl_xor compiles to ds_bpermute_b32 on this target -- an LDS-crossbar round trip per step, six dependent ones for a
// 64-lane reduction (~500 cycles).  Inside a 16-lane row the butterfly runs on DPP instead (quad_perm for xor 1 and 2;
// after those every quad is uniform, so row_half_mirror pairs quads and row_mirror pairs half rows exactly as xor 4 and
// xor 8 would), and the four row results of a wavefront are combined from v_readlane scalars -- the association is
// that of the xor butterfly, so sums come out bit for bit the same.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int W, typename OP>
__device__ __forceinline__ float lanes_reduce(float v, OP op) {
    static_assert(W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group width");
#ifdef BLGNN_NO_DPP                                       // (A/B switch: the ds_bpermute butterfly)
    for (int o = 1; o < W; o <<= 1) v = op(v, __shfl_xor(v, o, WAVE));
    return v;
#endif
    if (W >= 2) v = op(v, dpp_f32<0xB1>(v));             // quad_perm [1, 0, 3, 2]
    if (W >= 4) v = op(v, dpp_f32<0x4E>(v));             // quad_perm [2, 3, 0, 1]
    if (W >= 8) v = op(v, dpp_f32<0x141>(v));            // row_half_mirror
    if (W >= 16) v = op(v, dpp_f32<0x140>(v));           // row_mirror
    if (W == 32) v = op(v, __shfl_xor(v, 16, WAVE));
    if (W == 64) {
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
        v = op(op(r0, r1), op(r2, r3));
    }
    return v;
}
template <int W> __device__ __forceinline__ float lanes_max(float v) {
    return lanes_reduce<W>(v, [](float a, float b) { return fmaxf(a, b); });
}
template <int W> __device__ __forceinline__ float lanes_sum(float v) {
    return lanes_reduce<W>(v, [](float a, float b) { return a + b; });
}

// per-channel epilogue shared by the conv kernels and the GEMM:
//   v -> act(v + bias[c]) * bn_scale[c] + bn_shift[c]
struct Epilogue {
    const float *bias;       // nullable
    const float *act_param;  // nullable (PReLU slope per channel)
    const float *bn_scale;   // nullable
    const float *bn_shift;   // nullable
    int act;
    __device__ __forceinline__ float operator()(float v, int c) const {
        if (bias) v += bias[c];
        float slope = (act == BLGNN_ACT_PRELU && act_param) ? act_param[c] : 0.f;
        v = apply_act(v, act, slope);
        if (bn_scale) v = fmaf(v, bn_scale[c], bn_shift ? bn_shift[c] : 0.f);
        return v;
    }
};

// The epilogue of ONE output channel with its parameters already in registers.  Kernels whose
// lanes own a column (the MFMA C/D layout) load this once before their store loop: the
// compiler cannot hoist the parameter loads itself because the stores may alias them, and
// four dependent L1 round trips in front of each of 32-64 stores cost more than the GEMM.
struct EpilogueCol {
    float bias, slope, scale, shift;
    int act;
    bool affine;
    __device__ __forceinline__ EpilogueCol(const Epilogue &E, int c, bool valid) {
        bias = (valid && E.bias) ? E.bias[c] : 0.f;
        slope = (valid && E.act == BLGNN_ACT_PRELU && E.act_param) ? E.act_param[c] : 0.f;
        affine = E.bn_scale != nullptr;
        scale = (valid && affine) ? E.bn_scale[c] : 1.f;
        shift = (valid && affine && E.bn_shift) ? E.bn_shift[c] : 0.f;
        act = E.act;
    }
    __device__ __forceinline__ float operator()(float v) const {
        v = apply_act(v + bias, act, slope);
        return affine ? fmaf(v, scale, shift) : v;
    }
};

}  // namespace blgnn
