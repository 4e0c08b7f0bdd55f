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
