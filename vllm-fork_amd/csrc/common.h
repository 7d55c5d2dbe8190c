// Shared device helpers for the gfx950 kernels of libmi355q.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mi355q.h"

// banks of library-owned arrival counters: one per stream that uses them (stream_bank); enough for
// the stream pools of a torch process (32 per priority and device)
#define MI355Q_STATE_BANKS 64

// The in-kernel hand-offs (partition merge of the attention kernels, ticketed split-K merges) publish with
// agent-scope relaxed atomic stores -- write-through (sc1) on gfx950 -- followed by s_waitcnt vmcnt(0), a barrier
// and a relaxed ticket, and read with agent-scope atomic loads; no release fence.  That is a property of this
// target's code generation and caches, not of the HIP memory model: refuse to build the device code for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libmi355q device code is written for gfx950 only (fence-free write-through hand-offs, MFMA shapes, LDS-DMA)"
#endif

namespace mi355q {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// error plumbing (capi.cpp)
int set_error(int code, const char* fmt, ...);
int check_launch(const char* what);
// environment knobs, read once per process (capi.cpp)
const char* env_value(const char* name);
bool env_flag(const char* name);
int env_int(const char* name, int dflt);
// bank of library-owned counters for a stream, or -1 (capi.cpp)
int stream_bank(void* stream);

#define MI355Q_REQUIRE(cond, ...)                              \
  do {                                                         \
    if (!(cond)) return set_error(MI355Q_EINVAL, __VA_ARGS__); \
  } while (0)

// ---- 16-bit float <-> f32 -------------------------------------------------
struct BF16 {
  static constexpr int id = MI355Q_BF16;
  static constexpr uint32_t int4_magic = 0x43004300u;  // bf16(128) in both halves
  static constexpr float int4_bias = 128.0f;
  static constexpr uint32_t ones = 0x3F803F80u;
  __host__ __device__ static inline float to_f32(uint16_t h) {
    union { uint32_t u; float f; } c;
    c.u = (uint32_t)h << 16;
    return c.f;
  }
  // round to nearest even; NaN stays NaN
  __host__ __device__ static inline uint16_t from_f32(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(uint16_t, (__bf16)f);   // v_cvt_pk_bf16_f32: the same rounding in one instruction
#endif
    union { float f; uint32_t u; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
};

struct FP16 {
  static constexpr int id = MI355Q_FP16;
  static constexpr uint32_t int4_magic = 0x64006400u;  // fp16(1024)
  static constexpr float int4_bias = 1024.0f;
  static constexpr uint32_t ones = 0x3C003C00u;
  __device__ static inline float to_f32(uint16_t h) {
    return (float)__builtin_bit_cast(_Float16, h);
  }
  __device__ static inline uint16_t from_f32(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
};

template <typename DT>
__device__ inline uint32_t pack2(float lo, float hi) {
  return (uint32_t)DT::from_f32(lo) | ((uint32_t)DT::from_f32(hi) << 16);
}
// bf16 pairs go through the hardware conversion (v_cvt_pk_bf16_f32: round to nearest even, NaN
// stays NaN) -- one instruction instead of the integer rounding sequence, same values
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_cvt;
template <>
__device__ inline uint32_t pack2<BF16>(float lo, float hi) {
  const f32x2 v = f32x2{lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_cvt));
}

template <typename DT>
__device__ inline f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c);
template <>
__device__ inline f32x4 mfma16<BF16>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <>
__device__ inline f32x4 mfma16<FP16>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// ---- OCP e4m3fn (gfx950 hardware converts) ---------------------------------
__device__ inline f32x2 fp8x2_to_f32(uint32_t packed, bool hi) {
  return hi ? __builtin_amdgcn_cvt_pk_f32_fp8(packed, true)
            : __builtin_amdgcn_cvt_pk_f32_fp8(packed, false);
}
// saturating round-to-nearest-even encode of one value
__device__ inline uint8_t f32_to_fp8(float v) {
  if (v != v) return 0x7f;  // NaN stays NaN
  v = __builtin_fminf(__builtin_fmaxf(v, -448.0f), 448.0f);
  uint32_t r = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false);
  return (uint8_t)(r & 0xff);
}

// ---- wave / block reductions ------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Wave sum without LDS permutes: four data-parallel moves inside each row of 16 lanes, then the two
// row-swap instructions of gfx950 (each partner pair is added as own + partner, so every lane ends with the same
// bits).  Deterministic, but associated differently from wave_sum: lanes pair 1, 2, 4, 8, 16, 32 apart in that order.
__device__ inline float wave_sum_dpp(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // lane ^ 1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // lane ^ 2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // the other quad of the half row
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // the other half row
  u2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Sum over a block of NW waves; every thread gets the total.  `scratch` holds NW floats.
template <int NW>
__device__ inline float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; i++) t += scratch[i];
  return t;
}

}  // namespace mi355q
