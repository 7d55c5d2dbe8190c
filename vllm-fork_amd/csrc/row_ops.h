// Device functions shared by kernels of different source files that must produce the same bits:
// the in-register add + RMSNorm of one row, and the packed 16-bit dot product.
#pragma once
#include "common.h"

namespace mi355q {

template <typename DT>
__device__ inline float dot2(uint32_t a, uint32_t b, float c);
template <>
__device__ inline float dot2<BF16>(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
template <>
__device__ inline float dot2<FP16>(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
}

// residual = round(x + residual); x = round(residual * rs * w), one row by one workgroup of NT threads,
// H <= 8 NT VPT elements, 16-byte aligned.  x, residual and the norm weight are all requested before
// anything is consumed, so the row costs one memory round trip, one reduction and one write.  Same
// per-element arithmetic as fused_add_rms_norm_kernel.  `copy` (LDS, H / 8 vectors) also receives the
// normalised row when not null; the caller synchronises before reading it.
template <typename DT, int NT, int VPT>
__device__ __forceinline__ void add_rms_norm_row_vec(uint16_t* __restrict__ xrow, uint16_t* __restrict__ rrow,
                                                     const uint16_t* __restrict__ w, float eps, int H,
                                                     float* scratch, u32x4* copy) {
  u32x4* x = reinterpret_cast<u32x4*>(xrow);
  u32x4* r = reinterpret_cast<u32x4*>(rrow);
  const u32x4* wv = reinterpret_cast<const u32x4*>(w);
  const int nv = H >> 3;
  u32x4 xv[VPT], rv[VPT], ww[VPT];
#pragma unroll
  for (int j = 0; j < VPT; j++) {
    const int i = min((int)threadIdx.x + j * NT, nv - 1);   // clamped: the loads stay unconditional
    xv[j] = x[i];
    rv[j] = r[i];
    ww[j] = w ? wv[i] : u32x4{0, 0, 0, 0};
  }
  float z[VPT][8];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < VPT; j++) {
    const bool live = (int)threadIdx.x + j * NT < nv;
    u32x4 zo;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint16_t z0 = DT::from_f32(DT::to_f32((uint16_t)(xv[j][e] & 0xffff)) + DT::to_f32((uint16_t)(rv[j][e] & 0xffff)));
      const uint16_t z1 = DT::from_f32(DT::to_f32((uint16_t)(xv[j][e] >> 16)) + DT::to_f32((uint16_t)(rv[j][e] >> 16)));
      zo[e] = (uint32_t)z0 | ((uint32_t)z1 << 16);
      z[j][2 * e] = DT::to_f32(z0);
      z[j][2 * e + 1] = DT::to_f32(z1);
      if (live) ss += z[j][2 * e] * z[j][2 * e] + z[j][2 * e + 1] * z[j][2 * e + 1];
    }
    if (live) r[threadIdx.x + j * NT] = zo;
  }
  const float rs = rsqrtf(block_sum<NT / 64>(ss, scratch) / (float)H + eps);
#pragma unroll
  for (int j = 0; j < VPT; j++) {
    if ((int)threadIdx.x + j * NT >= nv) continue;
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float a = z[j][2 * e] * rs, b = z[j][2 * e + 1] * rs;
      if (w) {
        a *= DT::to_f32((uint16_t)(ww[j][e] & 0xffff));
        b *= DT::to_f32((uint16_t)(ww[j][e] >> 16));
      }
      o[e] = (uint32_t)DT::from_f32(a) | ((uint32_t)DT::from_f32(b) << 16);
    }
    x[threadIdx.x + j * NT] = o;
    if (copy) copy[threadIdx.x + j * NT] = o;
  }
}

}  // namespace mi355q
