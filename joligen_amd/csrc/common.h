// Shared device helpers for the gfx950 kernels of libjg355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/jg355.h"

typedef _Float16 f16_t;
typedef __bf16 bf16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define JG_WAVE 64

// Dispatch switches: read ONCE from the environment, overridable at run time through the C ABI
// (jg_set_tuning, include/jg355.h) so that tests can force a kernel configuration that the automatic choice would
// not pick at test-sized shapes.  No per-launch getenv().
enum JgTune { JG_TUNE_HALO_CFG = 0, JG_TUNE_WGRAD_HALO_CFG, JG_TUNE_CONV_VARIANT, JG_TUNE_WGRAD_VARIANT, JG_TUNE_SINKHORN_GENERIC,
              JG_TUNE_CONV1X1, JG_TUNE_GN_REVERSE, JG_TUNE_HALO_DBG, JG_TUNE_PERSIST64, JG_TUNE_HALO_PIPE, JG_TUNE_WGRAD_PIPE, JG_TUNE_CONV_SPLITK, JG_TUNE_CONV_SMALL_TILE, JG_TUNE_GN_FUSED, JG_TUNE_GN_FUSED_CAP, JG_TUNE_GN_FUSED_DBG, JG_TUNE_GN_FUSED_SLEEP, JG_TUNE_WGRAD_LDS_PAD,
              JG_TUNE_LN_BWD_CAP, JG_TUNE_DW_BWD_CAP, JG_TUNE_DW_BWD_PPT, JG_TUNE_CONV_KXK, JG_TUNE_CONV_RING, JG_TUNE_WGRAD_DEEP, JG_TUNE_WGRAD_SW, JG_TUNE_DETERMINISTIC, JG_TUNE_WGRAD_GROUP_BLOCKS, JG_TUNE_SGEMM_SPLIT, JG_TUNE_WGRAD_BIG, JG_TUNE_DW_RUN, JG_TUNE_COUNT };
int jg_tune(int which);
// dispatch sites record which kernel instance handled the launch (read back through jg_last_kernel(): bench.py / tools name the
// roofline rows by what actually ran, not by a host-side guess of the dispatch)
void jg_note_kernel(const char* name);

#define JG_CHECK_LAUNCH()                          \
  do {                                             \
    if (hipGetLastError() != hipSuccess) return JG_ERR_LAUNCH; \
  } while (0)

// ---- 16-bit <-> fp32 ----------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

template <typename T> __device__ __forceinline__ T bits_to(uint16_t b) { return __builtin_bit_cast(T, b); }
template <typename T> __device__ __forceinline__ uint16_t to_bits(T v) { return __builtin_bit_cast(uint16_t, v); }

// unpack a 16-byte chunk (8 x T) into floats and back
template <typename T> __device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = to_f32(bits_to<T>((uint16_t)(w[i] & 0xffffu)));
    f[2 * i + 1] = to_f32(bits_to<T>((uint16_t)(w[i] >> 16)));
  }
}
template <typename T> __device__ __forceinline__ uint4 pack8(const float* f) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w[i] = (uint32_t)to_bits<T>(from_f32<T>(f[2 * i])) | ((uint32_t)to_bits<T>(from_f32<T>(f[2 * i + 1])) << 16);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}
template <typename T> __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  uint2 r;
  r.x = (uint32_t)to_bits<T>(from_f32<T>(a)) | ((uint32_t)to_bits<T>(from_f32<T>(b)) << 16);
  r.y = (uint32_t)to_bits<T>(from_f32<T>(c)) | ((uint32_t)to_bits<T>(from_f32<T>(d)) << 16);
  return r;
}
template <typename T> __device__ __forceinline__ void unpack4(const uint2& v, float* f) {
  f[0] = to_f32(bits_to<T>((uint16_t)(v.x & 0xffffu)));
  f[1] = to_f32(bits_to<T>((uint16_t)(v.x >> 16)));
  f[2] = to_f32(bits_to<T>((uint16_t)(v.y & 0xffffu)));
  f[3] = to_f32(bits_to<T>((uint16_t)(v.y >> 16)));
}

// ---- MFMA 16x16x32 (gfx950): D[row][col] += sum_k A[row][k] B[k][col] -----------------
// operand fragments: lane l holds row/col (l & 15) and the 8 k-values of k-group (l >> 4);
// D: col = l & 15, row = (l >> 4) * 4 + reg.
template <typename T> struct Mfma;
template <> struct Mfma<f16_t> {
  static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma<bf16_t> {
  static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

// Predicated 16-byte global load: the address is always a valid one (callers pass the tensor
// base when !ok) so the load is unconditional and the zero fill is four v_cndmask; a ternary on
// the dereference makes hipcc select between the real pointer and a zeroed scratch slot.
template <typename T> __device__ __forceinline__ uint4 ldg16(const T* ptr, bool ok) {
  uint4 v = *reinterpret_cast<const uint4*>(ptr);
  v.x = ok ? v.x : 0u;
  v.y = ok ? v.y : 0u;
  v.z = ok ? v.z : 0u;
  v.w = ok ? v.w : 0u;
  return v;
}

// sigmoid through the hardware reciprocal (v_rcp_f32, 1 ulp): the IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup, ~10
// VALU issues) was a third of the instruction stream of the HBM-bound normalisation passes; results are stored in 16 bits
__device__ __forceinline__ float sigmoid_fast(float u) { return __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }
__device__ __forceinline__ float silu_f(float u) { return u * sigmoid_fast(u); }
__device__ __forceinline__ float silu_grad_f(float u) {
  const float s = sigmoid_fast(u);
  return s * (1.0f + u * (1.0f - s));
}

// activation fused into the normalisation passes: y = act(u); act_grad_f = d act / d u
template <int ACT> __device__ __forceinline__ float act_f(float u) {
  if (ACT == JG_ACT_SILU) return silu_f(u);
  if (ACT == JG_ACT_RELU) return fmaxf(u, 0.f);
  if (ACT == JG_ACT_LRELU) return u > 0.f ? u : 0.2f * u;
  return u;
}
template <int ACT> __device__ __forceinline__ float act_grad_f(float u) {
  if (ACT == JG_ACT_SILU) return silu_grad_f(u);
  if (ACT == JG_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (ACT == JG_ACT_LRELU) return u > 0.f ? 1.f : 0.2f;
  return 1.f;
}
__device__ __forceinline__ float act_grad_rt(float u, int act) {
  return act == JG_ACT_SILU ? silu_grad_f(u) : act == JG_ACT_RELU ? (u > 0.f ? 1.f : 0.f) : act == JG_ACT_LRELU ? (u > 0.f ? 1.f : 0.2f) : 1.f;
}

#define JG_DISPATCH_ACT(act, ...)                          \
  do {                                                     \
    if ((act) == JG_ACT_SILU) { constexpr int ACT = JG_ACT_SILU; __VA_ARGS__ }        \
    else if ((act) == JG_ACT_RELU) { constexpr int ACT = JG_ACT_RELU; __VA_ARGS__ }   \
    else if ((act) == JG_ACT_LRELU) { constexpr int ACT = JG_ACT_LRELU; __VA_ARGS__ } \
    else { constexpr int ACT = JG_ACT_NONE; __VA_ARGS__ }                              \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

#define JG_DISPATCH_DTYPE(dtype, ...)                 \
  do {                                                \
    if ((dtype) == JG_F16) {                          \
      typedef f16_t T;                                \
      __VA_ARGS__                                     \
    } else if ((dtype) == JG_BF16) {                  \
      typedef bf16_t T;                               \
      __VA_ARGS__                                     \
    } else {                                          \
      return JG_ERR_BAD_ARG;                          \
    }                                                 \
  } while (0)
