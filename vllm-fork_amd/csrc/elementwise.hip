// Memory-bound row kernels: RMSNorm family, RoPE, SiLU-mul, KV-cache write, fp8 convert,
// sampler inner loop.  One workgroup per token row, 16-byte vector accesses where the row
// allows it, fp32 arithmetic, wave64 shuffles + one LDS hop for the reductions.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "row_ops.h"

namespace mi355q {

template <typename DT>
__device__ inline float ld(const uint16_t* p, size_t i) { return DT::to_f32(p[i]); }

struct F32 {
  static constexpr int id = MI355Q_FP32;
};

// ---------------------------------------------------------------------------
// RMSNorm
// ---------------------------------------------------------------------------
template <typename DT, int NT>
__global__ __launch_bounds__(NT) void rms_norm_kernel(uint16_t* __restrict__ out,
                                                      const uint16_t* __restrict__ in,
                                                      int64_t in_stride,
                                                      const uint16_t* __restrict__ w, float eps,
                                                      int H) {
  __shared__ float scratch[NT / 64];
  const uint16_t* x = in + (size_t)blockIdx.x * in_stride;
  uint16_t* y = out + (size_t)blockIdx.x * H;
  const bool vec = (H % 8 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 15) == 0) &&
                   (!w || ((uintptr_t)w & 15) == 0);
  float ss = 0.f;
  if (vec) {
    for (int i = threadIdx.x; i < H / 8; i += NT) {
      const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float a = DT::to_f32((uint16_t)(v[e] & 0xffff)), b = DT::to_f32((uint16_t)(v[e] >> 16));
        ss += a * a + b * b;
      }
    }
  } else {
    for (int i = threadIdx.x; i < H; i += NT) {
      const float a = DT::to_f32(x[i]);
      ss += a * a;
    }
  }
  const float rs = rsqrtf(block_sum<NT / 64>(ss, scratch) / (float)H + eps);
  if (vec) {
    for (int i = threadIdx.x; i < H / 8; i += NT) {
      const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
      u32x4 wv = u32x4{0, 0, 0, 0};
      if (w) wv = reinterpret_cast<const u32x4*>(w)[i];
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float a = DT::to_f32((uint16_t)(v[e] & 0xffff)) * rs, b = DT::to_f32((uint16_t)(v[e] >> 16)) * rs;
        if (w) {
          a *= DT::to_f32((uint16_t)(wv[e] & 0xffff));
          b *= DT::to_f32((uint16_t)(wv[e] >> 16));
        }
        o[e] = pack2<DT>(a, b);
      }
      reinterpret_cast<u32x4*>(y)[i] = o;
    }
  } else {
    for (int i = threadIdx.x; i < H; i += NT) {
      float a = DT::to_f32(x[i]) * rs;
      if (w) a *= DT::to_f32(w[i]);
      y[i] = DT::from_f32(a);
    }
  }
}

// residual = round(x + residual [+ c]); x(out) = round(residual * rs * w).
// SPLITS > 0 variant: c = round(sum_s partial[s] + bias) takes the place of x and the
// result goes to `out` (x unused).
template <typename DT, int NT, bool FROM_PARTIAL>
__global__ __launch_bounds__(NT) void fused_add_rms_norm_kernel(
    uint16_t* __restrict__ xio, int64_t x_stride, uint16_t* __restrict__ residual,
    const float* __restrict__ partial, int splits, size_t split_stride,
    const uint16_t* __restrict__ bias, const uint16_t* __restrict__ w, float eps, int H) {
  __shared__ float scratch[NT / 64];
  uint16_t* x = xio + (size_t)blockIdx.x * x_stride;
  uint16_t* r = residual + (size_t)blockIdx.x * H;
  const float* pp = FROM_PARTIAL ? partial + (size_t)blockIdx.x * H : nullptr;
  float ss = 0.f;
  for (int i = threadIdx.x; i < H; i += NT) {
    float c;
    if constexpr (FROM_PARTIAL) {
      float s = pp[i];
      for (int k = 1; k < splits; k++) s += pp[(size_t)k * split_stride + i];
      c = DT::to_f32(DT::from_f32(s));
      if (bias) c = DT::to_f32(DT::from_f32(c + DT::to_f32(bias[i])));
    } else {
      c = DT::to_f32(x[i]);
    }
    const uint16_t z = DT::from_f32(c + DT::to_f32(r[i]));
    r[i] = z;
    const float zf = DT::to_f32(z);
    ss += zf * zf;
  }
  const float rs = rsqrtf(block_sum<NT / 64>(ss, scratch) / (float)H + eps);
  for (int i = threadIdx.x; i < H; i += NT) {
    float a = DT::to_f32(r[i]) * rs;
    if (w) a *= DT::to_f32(w[i]);
    x[i] = DT::from_f32(a);
  }
}

// Rows that fit in registers (H <= 8 NT VPT elements, 16-byte aligned): see add_rms_norm_row_vec.
template <typename DT, int NT, int VPT>
__global__ __launch_bounds__(NT) void fused_add_rms_norm_vec_kernel(
    uint16_t* __restrict__ xio, int64_t x_stride, uint16_t* __restrict__ residual,
    const uint16_t* __restrict__ w, float eps, int H) {
  __shared__ float scratch[NT / 64];
  add_rms_norm_row_vec<DT, NT, VPT>(xio + (size_t)blockIdx.x * x_stride, residual + (size_t)blockIdx.x * H, w, eps, H,
                                    scratch, nullptr);
}

// The same from fp32 K-slice partials: c = round(sum_s partial[s]) (+ bias, rounded again) stands in
// for x, the normalised row goes to `out`.  Four channels per vector, VPT vectors per thread held in
// registers (H <= 4 NT VPT); sixteen waves per row keep enough requests in flight for the slices.
template <typename DT, int NT, int VPT>
__global__ __launch_bounds__(NT) void splitk_add_rms_norm_vec_kernel(
    uint16_t* __restrict__ out, uint16_t* __restrict__ residual, const float* __restrict__ partial, int splits,
    size_t split_stride, const uint16_t* __restrict__ bias, const uint16_t* __restrict__ w, float eps, int H) {
  __shared__ float scratch[NT / 64];
  const int nv = H >> 2;
  u32x2* o = reinterpret_cast<u32x2*>(out + (size_t)blockIdx.x * H);
  u32x2* r = reinterpret_cast<u32x2*>(residual + (size_t)blockIdx.x * H);
  const f32x4* pp = reinterpret_cast<const f32x4*>(partial + (size_t)blockIdx.x * H);
  const size_t sstr = split_stride >> 2;
  auto lo = [](uint32_t u) { return DT::to_f32((uint16_t)(u & 0xffff)); };
  auto hi = [](uint32_t u) { return DT::to_f32((uint16_t)(u >> 16)); };
  int idx[VPT];
  u32x2 rv[VPT], wv[VPT], bv[VPT];
  f32x4 acc[VPT];
#pragma unroll
  for (int j = 0; j < VPT; j++) {
    idx[j] = min((int)threadIdx.x + j * NT, nv - 1);   // clamped: the loads stay unconditional
    rv[j] = r[idx[j]];
    wv[j] = w ? reinterpret_cast<const u32x2*>(w)[idx[j]] : u32x2{0u, 0u};
    bv[j] = bias ? reinterpret_cast<const u32x2*>(bias)[idx[j]] : u32x2{0u, 0u};
    acc[j] = pp[idx[j]];
  }
  // eight slices requested at a time (clamped, so the requests are unconditional), added in slice order
  for (int sp0 = 1; sp0 < splits; sp0 += 8) {
    f32x4 t[8][VPT];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const size_t base = (size_t)min(sp0 + u, splits - 1) * sstr;
#pragma unroll
      for (int j = 0; j < VPT; j++) t[u][j] = pp[base + idx[j]];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (sp0 + u < splits) {
#pragma unroll
        for (int j = 0; j < VPT; j++) acc[j] += t[u][j];
      }
    }
  }
  float z[VPT][4];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < VPT; j++) {
    const bool live = (int)threadIdx.x + j * NT < nv;
    const float rr[4] = {lo(rv[j][0]), hi(rv[j][0]), lo(rv[j][1]), hi(rv[j][1])};
    const float bb[4] = {lo(bv[j][0]), hi(bv[j][0]), lo(bv[j][1]), hi(bv[j][1])};
    uint16_t zb[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float c = DT::to_f32(DT::from_f32(acc[j][e]));
      if (bias) c = DT::to_f32(DT::from_f32(c + bb[e]));
      zb[e] = DT::from_f32(c + rr[e]);
      z[j][e] = DT::to_f32(zb[e]);
      if (live) ss += z[j][e] * z[j][e];
    }
    if (live) r[threadIdx.x + j * NT] = u32x2{(uint32_t)zb[0] | ((uint32_t)zb[1] << 16), (uint32_t)zb[2] | ((uint32_t)zb[3] << 16)};
  }
  const float rs = rsqrtf(block_sum<NT / 64>(ss, scratch) / (float)H + eps);
#pragma unroll
  for (int j = 0; j < VPT; j++) {
    if ((int)threadIdx.x + j * NT >= nv) continue;
    const float ww[4] = {lo(wv[j][0]), hi(wv[j][0]), lo(wv[j][1]), hi(wv[j][1])};
    uint16_t ob[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float a = z[j][e] * rs;
      if (w) a *= ww[e];
      ob[e] = DT::from_f32(a);
    }
    o[threadIdx.x + j * NT] = u32x2{(uint32_t)ob[0] | ((uint32_t)ob[1] << 16), (uint32_t)ob[2] | ((uint32_t)ob[3] << 16)};
  }
}

// ---------------------------------------------------------------------------
// RoPE (in place)
// ---------------------------------------------------------------------------
__device__ inline float load_cache(const void* c, size_t i, int dtype) {
  if (dtype == MI355Q_FP32) return reinterpret_cast<const float*>(c)[i];
  const uint16_t h = reinterpret_cast<const uint16_t*>(c)[i];
  return dtype == MI355Q_BF16 ? BF16::to_f32(h) : FP16::to_f32(h);
}

template <typename DT>
__global__ void rotary_kernel(const int64_t* __restrict__ positions, uint16_t* __restrict__ q,
                              int64_t q_stride, uint16_t* __restrict__ k, int64_t k_stride,
                              int head, int nh, int nkv, const void* __restrict__ cache, int rot,
                              int cache_dtype, int is_neox) {
  const int t = blockIdx.x;
  const int e = rot / 2;
  const size_t cbase = (size_t)positions[t] * rot;
  const int total = (nh + (k ? nkv : 0)) * e;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int h = idx / e, i = idx - h * e;
    uint16_t* arr = h < nh ? q + (size_t)t * q_stride + (size_t)h * head
                           : k + (size_t)t * k_stride + (size_t)(h - nh) * head;
    const int xi = is_neox ? i : 2 * i, yi = is_neox ? e + i : 2 * i + 1;
    const float c = load_cache(cache, cbase + i, cache_dtype);
    const float s = load_cache(cache, cbase + e + i, cache_dtype);
    const float xf = DT::to_f32(arr[xi]), yf = DT::to_f32(arr[yi]);
    arr[xi] = DT::from_f32(xf * c - yf * s);
    arr[yi] = DT::from_f32(yf * c + xf * s);
  }
}

// ---------------------------------------------------------------------------
// SiLU-and-mul
// ---------------------------------------------------------------------------
template <typename DT>
__global__ void silu_mul_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ in, int d) {
  const uint16_t* x = in + (size_t)blockIdx.x * 2 * d;
  uint16_t* y = out + (size_t)blockIdx.x * d;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    const float g = DT::to_f32(x[i]), u = DT::to_f32(x[d + i]);
    const float s = DT::to_f32(DT::from_f32(g / (1.0f + __expf(-g))));
    y[i] = DT::from_f32(s * u);
  }
}

// SiLU-and-multiply over a pair-interleaved row (g0, u0, g1, u1, ...): out[i] = silu(x[2 i]) * x[2 i + 1].
// What silu_mul_kernel gives after the halves have been separated; eight pairs per thread and step.
template <typename DT>
__global__ __launch_bounds__(256) void silu_mul_pairs_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ in,
                                                             int d) {
  const u32x4* x = reinterpret_cast<const u32x4*>(in + (size_t)blockIdx.x * 2 * d);
  u32x4* y = reinterpret_cast<u32x4*>(out + (size_t)blockIdx.x * d);
  for (int v = threadIdx.x; v < d / 8; v += 256) {
    const u32x4 a = x[2 * v], b = x[2 * v + 1];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t p0 = e < 2 ? a[2 * e] : b[2 * e - 4], p1 = e < 2 ? a[2 * e + 1] : b[2 * e - 3];
      const float g0 = DT::to_f32((uint16_t)(p0 & 0xffff)), u0 = DT::to_f32((uint16_t)(p0 >> 16));
      const float g1 = DT::to_f32((uint16_t)(p1 & 0xffff)), u1 = DT::to_f32((uint16_t)(p1 >> 16));
      const float s0 = DT::to_f32(DT::from_f32(g0 / (1.0f + __expf(-g0))));
      const float s1 = DT::to_f32(DT::from_f32(g1 / (1.0f + __expf(-g1))));
      o[e] = (uint32_t)DT::from_f32(s0 * u0) | ((uint32_t)DT::from_f32(s1 * u1) << 16);
    }
    y[v] = o;
  }
}

// ---------------------------------------------------------------------------
// KV-cache write
// ---------------------------------------------------------------------------
template <typename DT, bool FP8>
__global__ void reshape_and_cache_kernel(const uint16_t* __restrict__ key,
                                         const uint16_t* __restrict__ value, int64_t key_stride,
                                         int64_t value_stride, void* __restrict__ kc,
                                         void* __restrict__ vc,
                                         const int64_t* __restrict__ slot_mapping, int kvh,
                                         int head, int bs, int64_t block_stride,
                                         const float* __restrict__ k_scale,
                                         const float* __restrict__ v_scale) {
  const int64_t slot = slot_mapping[blockIdx.x];
  if (slot < 0) return;
  const int64_t b = slot / bs, o = slot % bs;
  constexpr int X = FP8 ? 16 : 8;
  const float ks = FP8 ? *k_scale : 1.f, vs = FP8 ? *v_scale : 1.f;
  for (int idx = threadIdx.x; idx < kvh * head; idx += blockDim.x) {
    const int h = idx / head, d = idx - h * head;
    const uint16_t kb = key[(size_t)blockIdx.x * key_stride + idx];
    const uint16_t vb = value[(size_t)blockIdx.x * value_stride + idx];
    const size_t ki = (size_t)b * block_stride + (((size_t)h * (head / X) + d / X) * bs + o) * X + d % X;
    const size_t vi = (size_t)b * block_stride + ((size_t)h * head + d) * bs + o;
    if constexpr (FP8) {
      reinterpret_cast<uint8_t*>(kc)[ki] = f32_to_fp8(DT::to_f32(kb) / ks);
      reinterpret_cast<uint8_t*>(vc)[vi] = f32_to_fp8(DT::to_f32(vb) / vs);
    } else {
      reinterpret_cast<uint16_t*>(kc)[ki] = kb;
      reinterpret_cast<uint16_t*>(vc)[vi] = vb;
    }
  }
}

// The same write for the flash layouts: both caches [blocks, block_size, heads, head] (NHD) or
// [blocks, heads, block_size, head] (HND), told apart by the strides; one scale per tensor or one per head.
// A head's elements are contiguous in source and cache, so a work item moves 8 of them (16 bytes in, 16 or 8 out).
template <typename DT, bool FP8>
__global__ void reshape_and_cache_flash_kernel(const uint16_t* __restrict__ key, const uint16_t* __restrict__ value,
                                               int64_t key_stride, int64_t value_stride, void* __restrict__ kc,
                                               void* __restrict__ vc, const int64_t* __restrict__ slot_mapping,
                                               int heads, int head, int bs, int64_t block_stride, int64_t page_stride,
                                               int64_t head_stride, const float* __restrict__ k_scale,
                                               const float* __restrict__ v_scale, int scale_stride) {
  const int64_t slot = slot_mapping[blockIdx.x];
  if (slot < 0) return;
  const int64_t b = slot / bs, o = slot % bs;
  const int vph = head >> 3;   // 8-element vectors per head
  for (int idx = threadIdx.x; idx < heads * vph; idx += blockDim.x) {
    const int h = idx / vph, d = (idx - h * vph) * 8;
    const u32x4 kv = *reinterpret_cast<const u32x4*>(key + (size_t)blockIdx.x * key_stride + (size_t)h * head + d);
    const u32x4 vv = *reinterpret_cast<const u32x4*>(value + (size_t)blockIdx.x * value_stride + (size_t)h * head + d);
    const size_t dst = (size_t)b * block_stride + (size_t)o * page_stride + (size_t)h * head_stride + d;
    if constexpr (FP8) {
      const float ks = k_scale[h * scale_stride], vs = v_scale[h * scale_stride];
      auto enc8 = [&](const u32x4 x, float sc) {
        u32x2 r;
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const uint32_t a = x[2 * e], c = x[2 * e + 1];
          r[e] = (uint32_t)f32_to_fp8(DT::to_f32((uint16_t)(a & 0xffff)) / sc) |
                 ((uint32_t)f32_to_fp8(DT::to_f32((uint16_t)(a >> 16)) / sc) << 8) |
                 ((uint32_t)f32_to_fp8(DT::to_f32((uint16_t)(c & 0xffff)) / sc) << 16) |
                 ((uint32_t)f32_to_fp8(DT::to_f32((uint16_t)(c >> 16)) / sc) << 24);
        }
        return r;
      };
      *reinterpret_cast<u32x2*>(reinterpret_cast<uint8_t*>(kc) + dst) = enc8(kv, ks);
      *reinterpret_cast<u32x2*>(reinterpret_cast<uint8_t*>(vc) + dst) = enc8(vv, vs);
    } else {
      *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(kc) + dst) = kv;
      *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(vc) + dst) = vv;
    }
  }
}

// RoPE on q and k (in place) and the KV-cache write of the rotated k and of v in one pass over the
// token: what rotary_kernel followed by reshape_and_cache_kernel leave behind, bit for bit (the cache
// receives k as rounded to the activation dtype by the rotation).  Tokens whose slot is negative are
// rotated and not cached.
template <typename DT, bool FP8>
__global__ void rope_and_cache_kernel(const int64_t* __restrict__ positions, uint16_t* __restrict__ q,
                                      int64_t q_stride, uint16_t* __restrict__ k, int64_t k_stride,
                                      const uint16_t* __restrict__ v, int64_t v_stride, int head, int nh,
                                      int nkv, const void* __restrict__ cache, int rot, int cache_dtype,
                                      int is_neox, void* __restrict__ kc, void* __restrict__ vc,
                                      const int64_t* __restrict__ slot_mapping, int bs, int64_t block_stride,
                                      const float* __restrict__ k_scale, const float* __restrict__ v_scale) {
  const int t = blockIdx.x;
  const int e = rot / 2;
  const size_t cbase = (size_t)positions[t] * rot;
  const int64_t slot = slot_mapping[t];
  const int64_t b = slot / bs, o = slot % bs;
  constexpr int X = FP8 ? 16 : 8;
  const float ks = FP8 ? *k_scale : 1.f, vs = FP8 ? *v_scale : 1.f;
  auto put_k = [&](int h, int d, uint16_t bits) {
    const size_t ki = (size_t)b * block_stride + (((size_t)h * (head / X) + d / X) * bs + o) * X + d % X;
    if constexpr (FP8) reinterpret_cast<uint8_t*>(kc)[ki] = f32_to_fp8(DT::to_f32(bits) / ks);
    else reinterpret_cast<uint16_t*>(kc)[ki] = bits;
  };
  const int total = (nh + nkv) * e;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int h = idx / e, i = idx - h * e;
    const bool is_k = h >= nh;
    uint16_t* arr = is_k ? k + (size_t)t * k_stride + (size_t)(h - nh) * head : q + (size_t)t * q_stride + (size_t)h * head;
    const int xi = is_neox ? i : 2 * i, yi = is_neox ? e + i : 2 * i + 1;
    const float c = load_cache(cache, cbase + i, cache_dtype);
    const float s = load_cache(cache, cbase + e + i, cache_dtype);
    const float xf = DT::to_f32(arr[xi]), yf = DT::to_f32(arr[yi]);
    const uint16_t xo = DT::from_f32(xf * c - yf * s), yo = DT::from_f32(yf * c + xf * s);
    arr[xi] = xo;
    arr[yi] = yo;
    if (is_k && slot >= 0) {
      put_k(h - nh, xi, xo);
      put_k(h - nh, yi, yo);
    }
  }
  if (slot < 0) return;
  // the part of k beyond the rotated dimensions (partial rotary), then v
  const int tail = head - rot;
  for (int idx = threadIdx.x; idx < nkv * tail; idx += blockDim.x) {
    const int h = idx / tail, d = rot + idx - h * tail;
    put_k(h, d, k[(size_t)t * k_stride + (size_t)h * head + d]);
  }
  for (int idx = threadIdx.x; idx < nkv * head; idx += blockDim.x) {
    const int h = idx / head, d = idx - h * head;
    const uint16_t vb = v[(size_t)t * v_stride + idx];
    const size_t vi = (size_t)b * block_stride + ((size_t)h * head + d) * bs + o;
    if constexpr (FP8) reinterpret_cast<uint8_t*>(vc)[vi] = f32_to_fp8(DT::to_f32(vb) / vs);
    else reinterpret_cast<uint16_t*>(vc)[vi] = vb;
  }
}

// NeoX pairing over the whole head, head a multiple of 16: four pairs per work item, 8-byte accesses.
template <typename DT, bool FP8>
__global__ __launch_bounds__(1024) void rope_and_cache_vec_kernel(
    const int64_t* __restrict__ positions, uint16_t* __restrict__ q, int64_t q_stride, uint16_t* __restrict__ k,
    int64_t k_stride, const uint16_t* __restrict__ v, int64_t v_stride, int head, int nh, int nkv,
    const void* __restrict__ cache, int cache_dtype, void* __restrict__ kc, void* __restrict__ vc,
    const int64_t* __restrict__ slot_mapping, int bs, int64_t block_stride, const float* __restrict__ k_scale,
    const float* __restrict__ v_scale) {
  const int t = blockIdx.x;
  const int e = head / 2, q4 = e / 4;
  const size_t cbase = (size_t)positions[t] * head;
  const int64_t slot = slot_mapping[t];
  const int64_t b = slot / bs, o = slot % bs;
  constexpr int X = FP8 ? 16 : 8;
  const float ks = FP8 ? *k_scale : 1.f, vs = FP8 ? *v_scale : 1.f;
  auto lo = [](uint32_t u) { return DT::to_f32((uint16_t)(u & 0xffff)); };
  auto hi = [](uint32_t u) { return DT::to_f32((uint16_t)(u >> 16)); };
  const int rope_items = (nh + nkv) * q4, v_items = nkv * head / 4;
  for (int item = threadIdx.x; item < rope_items + v_items; item += blockDim.x) {
    if (item < rope_items) {
      const int h = item / q4, i4 = (item - h * q4) * 4;
      const bool is_k = h >= nh;
      uint16_t* arr = is_k ? k + (size_t)t * k_stride + (size_t)(h - nh) * head : q + (size_t)t * q_stride + (size_t)h * head;
      const u32x2 xv = *reinterpret_cast<const u32x2*>(arr + i4), yv = *reinterpret_cast<const u32x2*>(arr + e + i4);
      const float x[4] = {lo(xv[0]), hi(xv[0]), lo(xv[1]), hi(xv[1])}, y[4] = {lo(yv[0]), hi(yv[0]), lo(yv[1]), hi(yv[1])};
      uint16_t xo[4], yo[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float c = load_cache(cache, cbase + i4 + j, cache_dtype);
        const float s = load_cache(cache, cbase + e + i4 + j, cache_dtype);
        xo[j] = DT::from_f32(x[j] * c - y[j] * s);
        yo[j] = DT::from_f32(y[j] * c + x[j] * s);
      }
      *reinterpret_cast<u32x2*>(arr + i4) = u32x2{(uint32_t)xo[0] | ((uint32_t)xo[1] << 16), (uint32_t)xo[2] | ((uint32_t)xo[3] << 16)};
      *reinterpret_cast<u32x2*>(arr + e + i4) = u32x2{(uint32_t)yo[0] | ((uint32_t)yo[1] << 16), (uint32_t)yo[2] | ((uint32_t)yo[3] << 16)};
      if (is_k && slot >= 0) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
          const int d = half ? e + i4 : i4;
          const uint16_t* bits = half ? yo : xo;
          const size_t ki = (size_t)b * block_stride + (((size_t)(h - nh) * (head / X) + d / X) * bs + o) * X + d % X;
          if constexpr (FP8) {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) w |= (uint32_t)f32_to_fp8(DT::to_f32(bits[j]) / ks) << (8 * j);
            *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(kc) + ki) = w;
          } else {
            *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(kc) + ki) =
                u32x2{(uint32_t)bits[0] | ((uint32_t)bits[1] << 16), (uint32_t)bits[2] | ((uint32_t)bits[3] << 16)};
          }
        }
      }
    } else if (slot >= 0) {
      const int j4 = (item - rope_items) * 4;
      const int h = j4 / head, d = j4 - h * head;
      const u32x2 vv = *reinterpret_cast<const u32x2*>(v + (size_t)t * v_stride + j4);
      const uint16_t vb[4] = {(uint16_t)(vv[0] & 0xffff), (uint16_t)(vv[0] >> 16), (uint16_t)(vv[1] & 0xffff), (uint16_t)(vv[1] >> 16)};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const size_t vi = (size_t)b * block_stride + ((size_t)h * head + d + j) * bs + o;
        if constexpr (FP8) reinterpret_cast<uint8_t*>(vc)[vi] = f32_to_fp8(DT::to_f32(vb[j]) / vs);
        else reinterpret_cast<uint16_t*>(vc)[vi] = vb[j];
      }
    }
  }
}

// The fused qkv projection finished in one pass over the token: K-slice sums of the fp32 partial tiles
// (slice order, rounded to the activation dtype, bias added to the rounded value), NeoX RoPE over the
// whole head on q and k, the rotated row written to `qkv` and k, v to the paged cache.  Equal, bit for
// bit, to the slice-merge kernel followed by rope_and_cache_kernel.  Four pairs per work item.
template <typename DT, bool FP8>
__global__ __launch_bounds__(1024) void splitk_rope_and_cache_kernel(
    const float* __restrict__ partial, int splits, size_t split_stride, const uint16_t* __restrict__ bias,
    uint16_t* __restrict__ qkv, int N, const int64_t* __restrict__ positions, int head, int nh, int nkv,
    const void* __restrict__ cache, int cache_dtype, void* __restrict__ kc, void* __restrict__ vc,
    const int64_t* __restrict__ slot_mapping, int bs, int64_t block_stride, const float* __restrict__ k_scale,
    const float* __restrict__ v_scale) {
  const int t = blockIdx.x;
  const int e = head / 2, q4 = e / 4;
  const size_t cbase = (size_t)positions[t] * head;
  const int64_t slot = slot_mapping[t];
  const int64_t b = slot / bs, o = slot % bs;
  constexpr int X = FP8 ? 16 : 8;
  const float ks = FP8 ? *k_scale : 1.f, vs = FP8 ? *v_scale : 1.f;
  const float* prow = partial + (size_t)t * N;
  uint16_t* orow = qkv + (size_t)t * N;
  auto lo = [](uint32_t u) { return DT::to_f32((uint16_t)(u & 0xffff)); };
  auto hi = [](uint32_t u) { return DT::to_f32((uint16_t)(u >> 16)); };
  // two columns blocks of four channels summed together (the second is a dummy repeat for v items)
  auto merge2 = [&](int ca, int cb, float (&va)[4], float (&vb)[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(prow + ca), c = *reinterpret_cast<const f32x4*>(prow + cb);
    for (int sp0 = 1; sp0 < splits; sp0 += 4) {
      f32x4 ta[4], tc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const size_t base = (size_t)min(sp0 + u, splits - 1) * split_stride;
        ta[u] = *reinterpret_cast<const f32x4*>(prow + base + ca);
        tc[u] = *reinterpret_cast<const f32x4*>(prow + base + cb);
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (sp0 + u < splits) { a += ta[u]; c += tc[u]; }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      va[j] = DT::to_f32(DT::from_f32(a[j]));
      vb[j] = DT::to_f32(DT::from_f32(c[j]));
    }
    if (bias) {
      const u32x2 ba = *reinterpret_cast<const u32x2*>(bias + ca), bb = *reinterpret_cast<const u32x2*>(bias + cb);
      const float fa[4] = {lo(ba[0]), hi(ba[0]), lo(ba[1]), hi(ba[1])}, fb[4] = {lo(bb[0]), hi(bb[0]), lo(bb[1]), hi(bb[1])};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        va[j] = DT::to_f32(DT::from_f32(va[j] + fa[j]));
        vb[j] = DT::to_f32(DT::from_f32(vb[j] + fb[j]));
      }
    }
  };
  auto put_k4 = [&](int h, int d, const uint16_t (&bits)[4]) {
    const size_t ki = (size_t)b * block_stride + (((size_t)h * (head / X) + d / X) * bs + o) * X + d % X;
    if constexpr (FP8) {
      uint32_t w = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) w |= (uint32_t)f32_to_fp8(DT::to_f32(bits[j]) / ks) << (8 * j);
      *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(kc) + ki) = w;
    } else {
      *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(kc) + ki) =
          u32x2{(uint32_t)bits[0] | ((uint32_t)bits[1] << 16), (uint32_t)bits[2] | ((uint32_t)bits[3] << 16)};
    }
  };
  const int rope_items = (nh + nkv) * q4, v_items = nkv * head / 4;
  for (int item = threadIdx.x; item < rope_items + v_items; item += blockDim.x) {
    if (item < rope_items) {
      const int h = item / q4, i4 = (item - h * q4) * 4;
      const int cx = h * head + i4, cy = cx + e;
      float x[4], y[4];
      merge2(cx, cy, x, y);
      uint16_t xo[4], yo[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float c = load_cache(cache, cbase + i4 + j, cache_dtype);
        const float s = load_cache(cache, cbase + e + i4 + j, cache_dtype);
        xo[j] = DT::from_f32(x[j] * c - y[j] * s);
        yo[j] = DT::from_f32(y[j] * c + x[j] * s);
      }
      *reinterpret_cast<u32x2*>(orow + cx) = u32x2{(uint32_t)xo[0] | ((uint32_t)xo[1] << 16), (uint32_t)xo[2] | ((uint32_t)xo[3] << 16)};
      *reinterpret_cast<u32x2*>(orow + cy) = u32x2{(uint32_t)yo[0] | ((uint32_t)yo[1] << 16), (uint32_t)yo[2] | ((uint32_t)yo[3] << 16)};
      if (h >= nh && slot >= 0) {
        put_k4(h - nh, i4, xo);
        put_k4(h - nh, e + i4, yo);
      }
    } else {
      const int j4 = (item - rope_items) * 4;          // value channel h * head + d
      const int col = (nh + nkv) * head + j4;
      float v[4], dummy[4];
      merge2(col, col, v, dummy);
      uint16_t vb[4];
#pragma unroll
      for (int j = 0; j < 4; j++) vb[j] = DT::from_f32(v[j]);
      *reinterpret_cast<u32x2*>(orow + col) = u32x2{(uint32_t)vb[0] | ((uint32_t)vb[1] << 16), (uint32_t)vb[2] | ((uint32_t)vb[3] << 16)};
      if (slot >= 0) {
        const int h = j4 / head, d = j4 - h * head;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const size_t vi = (size_t)b * block_stride + ((size_t)h * head + d + j) * bs + o;
          if constexpr (FP8) reinterpret_cast<uint8_t*>(vc)[vi] = f32_to_fp8(DT::to_f32(vb[j]) / vs);
          else reinterpret_cast<uint16_t*>(vc)[vi] = vb[j];
        }
      }
    }
  }
}

template <typename DT>
__global__ void convert_fp8_kernel(void* __restrict__ dst, const void* __restrict__ src, int64_t n,
                                   float scale, int to_fp8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (to_fp8) {
      reinterpret_cast<uint8_t*>(dst)[i] =
          f32_to_fp8(DT::to_f32(reinterpret_cast<const uint16_t*>(src)[i]) / scale);
    } else {
      const uint32_t b = reinterpret_cast<const uint8_t*>(src)[i];
      const f32x2 f = __builtin_amdgcn_cvt_pk_f32_fp8(b, false);
      reinterpret_cast<uint16_t*>(dst)[i] = DT::from_f32(f[0] * scale);
    }
  }
}

// ---------------------------------------------------------------------------
// Sampler inner loop
// ---------------------------------------------------------------------------
template <int DTID>
__device__ inline float load_logit(const void* p, size_t i) {
  if constexpr (DTID == MI355Q_FP32) return reinterpret_cast<const float*>(p)[i];
  else if constexpr (DTID == MI355Q_BF16) return BF16::to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
  else return FP16::to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
}

// first index of the maximum; NaNs are ignored
template <int DTID>
__global__ __launch_bounds__(1024) void argmax_kernel(const void* __restrict__ logits,
                                                      int64_t row_stride, int64_t* __restrict__ ids,
                                                      int V) {
  __shared__ float sval[16];
  __shared__ int sidx[16];
  const char* row = reinterpret_cast<const char*>(logits) +
                    (size_t)blockIdx.x * row_stride * (DTID == MI355Q_FP32 ? 4 : 2);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += 1024) {
    const float v = load_logit<DTID>(row, i);
    if (v > best) {  // strided ascending scan keeps the first index per thread
      best = v;
      bi = i;
    }
  }
  if (bi == 0x7fffffff && threadIdx.x == 0) bi = 0;  // row of -inf / NaN: index 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sval[wave] = best;
    sidx[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w2 = 1; w2 < 16; w2++)
      if (sval[w2] > best || (sval[w2] == best && sidx[w2] < bi)) {
        best = sval[w2];
        bi = sidx[w2];
      }
    ids[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;
  }
}

// Wide rows (a vocabulary): the row is cut into segments, one workgroup each, that read 16 bytes
// per lane; a segment's best (value, first index) becomes one 64-bit key -- order-preserving
// image of the float in the upper half, 0x7fffffff - index in the lower -- merged with an atomic
// max per row, and the workgroup that draws the row's last ticket decodes the key.  The per-row
// slots live in device memory owned by the library and return to zero after every call; every stream
// that launches the kernel gets a bank of its own (stream_bank), so calls that overlap on different
// streams do not share slots.  mi355q_reset_library_state zeroes them after an aborted launch.
constexpr int ARGMAX_ROWS = 1024;
constexpr int ARGMAX_SEG_ELEMS = 8192;
static __device__ unsigned long long g_argmax_key[MI355Q_STATE_BANKS * ARGMAX_ROWS];
static __device__ int g_argmax_cnt[MI355Q_STATE_BANKS * ARGMAX_ROWS];

__global__ void argmax_reset_kernel() {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < MI355Q_STATE_BANKS * ARGMAX_ROWS; i += gridDim.x * blockDim.x) {
    g_argmax_key[i] = 0ull;
    g_argmax_cnt[i] = 0;
  }
}
int reset_argmax_state(hipStream_t stream) {
  hipLaunchKernelGGL(argmax_reset_kernel, dim3(4), dim3(256), 0, stream);
  return check_launch("argmax_reset_kernel");
}

// What a greedy decode step does with the sampled id besides storing it: advance the sequence's position and
// length, and fetch the id's embedding row as the next step's first hidden state.  All pointers null: nothing.
struct StepTail {
  int64_t* positions;
  int32_t* seq_lens;
  const uint16_t* embed;
  int64_t embed_stride;
  uint16_t* hidden;
  int64_t hidden_stride;
  int hidden_size;   // 16-bit elements, a multiple of 8
};

__device__ inline void step_tail_row(const StepTail& t, int rowi, int64_t id) {   // the whole workgroup calls this
  if (threadIdx.x == 0) {
    if (t.positions) t.positions[rowi] += 1;
    if (t.seq_lens) t.seq_lens[rowi] += 1;
  }
  if (t.embed && t.hidden) {
    const u32x4* src = reinterpret_cast<const u32x4*>(t.embed + (size_t)id * t.embed_stride);
    u32x4* dst = reinterpret_cast<u32x4*>(t.hidden + (size_t)rowi * t.hidden_stride);
    for (int v = threadIdx.x; v < t.hidden_size / 8; v += blockDim.x) dst[v] = src[v];
  }
}

// the tail as a launch of its own: after the single-workgroup arg-max, one workgroup per row
__global__ void step_tail_kernel(const int64_t* __restrict__ ids, StepTail t) { step_tail_row(t, blockIdx.x, ids[blockIdx.x]); }

template <int DTID>
__global__ __launch_bounds__(256) void argmax_wide_kernel(const void* __restrict__ logits, int64_t row_stride,
                                                          int64_t* __restrict__ ids, int V, int segs, int bank,
                                                          const StepTail tail) {
  constexpr int ES = DTID == MI355Q_FP32 ? 4 : 2, EPV = 16 / ES;
  typedef unsigned long long u64;
  __shared__ u64 skey[4];
  __shared__ long long s_winner;
  const int rowi = blockIdx.x, seg = blockIdx.y;
  const char* row = reinterpret_cast<const char*>(logits) + (size_t)rowi * row_stride * ES;
  const int per = ((V + segs - 1) / segs + EPV - 1) / EPV * EPV;   // elements per segment, whole vectors
  const int e0 = seg * per, e1 = min(V, e0 + per);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  auto take = [&](float v, int i) {
    if (v > best) { best = v; bi = i; }   // ascending scan keeps the first index per thread
  };
  const int nv = max(e1 - e0, 0) / EPV;
  for (int v0 = threadIdx.x; v0 < nv; v0 += 4 * 256) {
    u32x4 t[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (v0 + j * 256 < nv) t[j] = *reinterpret_cast<const u32x4*>(row + ((size_t)e0 + (size_t)(v0 + j * 256) * EPV) * ES);
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (v0 + j * 256 < nv) {
        const int i0 = e0 + (v0 + j * 256) * EPV;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if constexpr (DTID == MI355Q_FP32) take(__uint_as_float(t[j][e]), i0 + e);
          else if constexpr (DTID == MI355Q_BF16) {
            take(BF16::to_f32((uint16_t)(t[j][e] & 0xffff)), i0 + 2 * e);
            take(BF16::to_f32((uint16_t)(t[j][e] >> 16)), i0 + 2 * e + 1);
          } else {
            take(FP16::to_f32((uint16_t)(t[j][e] & 0xffff)), i0 + 2 * e);
            take(FP16::to_f32((uint16_t)(t[j][e] >> 16)), i0 + 2 * e + 1);
          }
        }
      }
  }
  // a thread's vectors are visited in ascending index order, but `take` on equal values must keep
  // the smaller index across vectors too: it does (strict >)
  for (int i = e0 + nv * EPV + threadIdx.x; i < e1; i += 256) take(load_logit<DTID>(row, i), i);
  const uint32_t fb = __float_as_uint(best);
  const uint32_t ord = (fb & 0x80000000u) ? ~fb : (fb | 0x80000000u);
  u64 key = ((u64)ord << 32) | (uint32_t)(0x7fffffff - bi);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 other = __shfl_xor(key, o, 64);
    key = other > key ? other : key;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) skey[wave] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w2 = 1; w2 < 4; w2++) key = skey[w2] > key ? skey[w2] : key;
    const int slot = bank * ARGMAX_ROWS + rowi;
    __hip_atomic_fetch_max(&g_argmax_key[slot], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const int ticket = __hip_atomic_fetch_add(&g_argmax_cnt[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == segs - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const u64 k = __hip_atomic_exchange(&g_argmax_key[slot], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&g_argmax_cnt[slot], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int idx = 0x7fffffff - (int)(uint32_t)(k & 0xffffffffu);
      ids[rowi] = idx == 0x7fffffff ? 0 : idx;   // row of -inf / NaN: index 0
      s_winner = idx == 0x7fffffff ? 0 : idx;
    } else {
      s_winner = -1;
    }
  }
  if (tail.positions || tail.seq_lens || tail.embed) {   // the workgroup that closed the row runs the step's tail
    __syncthreads();
    const long long id = s_winner;
    if (id >= 0) step_tail_row(tail, rowi, id);
  }
}

template <int DTID>
__global__ void repetition_penalty_kernel(void* __restrict__ logits,
                                          const uint8_t* __restrict__ pm,
                                          const uint8_t* __restrict__ om,
                                          const void* __restrict__ pen, int V) {
  const int b = blockIdx.y;
  const float p = load_logit<DTID>(pen, b);
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) {
    const size_t o = (size_t)b * V + v;
    if (pm[o] || om[o]) {
      float l = load_logit<DTID>(logits, o);
      l = l > 0.f ? l / p : l * p;
      if constexpr (DTID == MI355Q_FP32) reinterpret_cast<float*>(logits)[o] = l;
      else if constexpr (DTID == MI355Q_BF16) reinterpret_cast<uint16_t*>(logits)[o] = BF16::from_f32(l);
      else reinterpret_cast<uint16_t*>(logits)[o] = FP16::from_f32(l);
    }
  }
}

static int check_h16(int dtype, const char* what) {
  MI355Q_REQUIRE(dtype == MI355Q_BF16 || dtype == MI355Q_FP16, "%s: dtype must be bf16 or fp16", what);
  return 0;
}

}  // namespace mi355q

using namespace mi355q;

#define DISPATCH_H16(dtype, ...)            \
  if ((dtype) == MI355Q_BF16) {             \
    using DT = BF16;                        \
    __VA_ARGS__;                            \
  } else {                                  \
    using DT = FP16;                        \
    __VA_ARGS__;                            \
  }

extern "C" int mi355q_rms_norm(void* out, const void* input, int64_t input_stride,
                               const void* weight, float epsilon, int num_tokens, int hidden,
                               int dtype, mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "rms_norm")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && input && hidden > 0 && num_tokens >= 0, "rms_norm: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (num_tokens < 256) {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((rms_norm_kernel<DT, 1024>), dim3(num_tokens), dim3(1024), 0, s,
                                           (uint16_t*)out, (const uint16_t*)input, input_stride,
                                           (const uint16_t*)weight, epsilon, hidden));
  } else {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((rms_norm_kernel<DT, 256>), dim3(num_tokens), dim3(256), 0, s,
                                           (uint16_t*)out, (const uint16_t*)input, input_stride,
                                           (const uint16_t*)weight, epsilon, hidden));
  }
  return check_launch("rms_norm_kernel");
}

extern "C" int mi355q_fused_add_rms_norm(void* input, int64_t input_stride, void* residual,
                                         const void* weight, float epsilon, int num_tokens,
                                         int hidden, int dtype, mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "fused_add_rms_norm")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(input && residual && hidden > 0 && num_tokens >= 0, "fused_add_rms_norm: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = hidden % 8 == 0 && input_stride % 8 == 0 && (((uintptr_t)input | (uintptr_t)residual | (uintptr_t)weight) & 15) == 0;
  if (aligned && hidden <= 8 * 256 * 4 && !env_flag("MI355Q_NORM_SCALAR")) {
    if (hidden <= 8 * 256 * 2) {
      DISPATCH_H16(dtype, hipLaunchKernelGGL((fused_add_rms_norm_vec_kernel<DT, 256, 2>), dim3(num_tokens), dim3(256), 0, s,
                                             (uint16_t*)input, input_stride, (uint16_t*)residual,
                                             (const uint16_t*)weight, epsilon, hidden));
    } else {
      DISPATCH_H16(dtype, hipLaunchKernelGGL((fused_add_rms_norm_vec_kernel<DT, 256, 4>), dim3(num_tokens), dim3(256), 0, s,
                                             (uint16_t*)input, input_stride, (uint16_t*)residual,
                                             (const uint16_t*)weight, epsilon, hidden));
    }
    return check_launch("fused_add_rms_norm_vec_kernel");
  }
  DISPATCH_H16(dtype, hipLaunchKernelGGL((fused_add_rms_norm_kernel<DT, 1024, false>), dim3(num_tokens),
                                         dim3(1024), 0, s, (uint16_t*)input, input_stride,
                                         (uint16_t*)residual, (const float*)nullptr, 0, (size_t)0,
                                         (const uint16_t*)nullptr, (const uint16_t*)weight, epsilon, hidden));
  return check_launch("fused_add_rms_norm_kernel");
}

extern "C" int mi355q_splitk_add_rms_norm(void* out, void* residual, const float* partial,
                                          int splits, const void* bias, const void* weight,
                                          float epsilon, int num_tokens, int hidden, int dtype,
                                          mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "splitk_add_rms_norm")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && residual && partial && splits >= 1 && hidden > 0 && num_tokens >= 0,
                 "splitk_add_rms_norm: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = hidden % 4 == 0 && (((uintptr_t)out | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)bias) & 7) == 0 &&
                       ((uintptr_t)partial & 15) == 0;
  if (aligned && hidden <= 4 * 256 * 8 && !env_flag("MI355Q_NORM_SCALAR")) {
    const size_t sstr = (size_t)num_tokens * hidden;
    if (hidden <= 4 * 1024) {
      DISPATCH_H16(dtype, hipLaunchKernelGGL((splitk_add_rms_norm_vec_kernel<DT, 1024, 1>), dim3(num_tokens), dim3(1024), 0, s,
                                             (uint16_t*)out, (uint16_t*)residual, partial, splits, sstr,
                                             (const uint16_t*)bias, (const uint16_t*)weight, epsilon, hidden));
    } else {
      DISPATCH_H16(dtype, hipLaunchKernelGGL((splitk_add_rms_norm_vec_kernel<DT, 1024, 2>), dim3(num_tokens), dim3(1024), 0, s,
                                             (uint16_t*)out, (uint16_t*)residual, partial, splits, sstr,
                                             (const uint16_t*)bias, (const uint16_t*)weight, epsilon, hidden));
    }
    return check_launch("splitk_add_rms_norm_vec_kernel");
  }
  DISPATCH_H16(dtype, hipLaunchKernelGGL((fused_add_rms_norm_kernel<DT, 1024, true>), dim3(num_tokens),
                                         dim3(1024), 0, s, (uint16_t*)out, (int64_t)hidden,
                                         (uint16_t*)residual, partial, splits,
                                         (size_t)num_tokens * hidden, (const uint16_t*)bias,
                                         (const uint16_t*)weight, epsilon, hidden));
  return check_launch("splitk_add_rms_norm_kernel");
}

extern "C" int mi355q_rotary_embedding(const int64_t* positions, void* query, int64_t query_stride,
                                       void* key, int64_t key_stride, int head_size, int num_heads,
                                       int num_kv_heads, const void* cos_sin_cache, int rot_dim,
                                       int cache_dtype, int is_neox, int num_tokens, int dtype,
                                       mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "rotary_embedding")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(positions && query && cos_sin_cache, "rotary_embedding: null pointer");
  MI355Q_REQUIRE(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rotary_embedding: rot_dim %d vs head_size %d", rot_dim, head_size);
  MI355Q_REQUIRE(cache_dtype >= MI355Q_BF16 && cache_dtype <= MI355Q_FP32, "rotary_embedding: bad cache dtype");
  if (num_tokens == 0) return MI355Q_OK;
  int threads = num_heads * rot_dim / 2;
  threads = threads < 64 ? 64 : (threads > 512 ? 512 : (threads + 63) / 64 * 64);
  DISPATCH_H16(dtype, hipLaunchKernelGGL((rotary_kernel<DT>), dim3(num_tokens), dim3(threads), 0,
                                         (hipStream_t)stream, positions, (uint16_t*)query, query_stride,
                                         (uint16_t*)key, key_stride, head_size, num_heads, num_kv_heads,
                                         cos_sin_cache, rot_dim, cache_dtype, is_neox));
  return check_launch("rotary_kernel");
}

extern "C" int mi355q_silu_and_mul(void* out, const void* input, int num_tokens, int d, int dtype,
                                   mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "silu_and_mul")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && input && d > 0, "silu_and_mul: bad arguments");
  DISPATCH_H16(dtype, hipLaunchKernelGGL((silu_mul_kernel<DT>), dim3(num_tokens), dim3(d < 1024 ? (d + 63) / 64 * 64 : 1024),
                                         0, (hipStream_t)stream, (uint16_t*)out, (const uint16_t*)input, d));
  return check_launch("silu_mul_kernel");
}

extern "C" int mi355q_silu_and_mul_pairs(void* out, const void* input, int num_tokens, int d, int dtype,
                                         mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "silu_and_mul_pairs")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && input && d > 0 && d % 8 == 0, "silu_and_mul_pairs: d = %d must be a positive multiple of 8", d);
  MI355Q_REQUIRE((((uintptr_t)out | (uintptr_t)input) & 15) == 0, "silu_and_mul_pairs: tensors must be 16-byte aligned");
  DISPATCH_H16(dtype, hipLaunchKernelGGL((silu_mul_pairs_kernel<DT>), dim3(num_tokens), dim3(256), 0, (hipStream_t)stream,
                                         (uint16_t*)out, (const uint16_t*)input, d));
  return check_launch("silu_mul_pairs_kernel");
}

extern "C" int mi355q_reshape_and_cache(const void* key, const void* value, int64_t key_stride,
                                        int64_t value_stride, void* key_cache, void* value_cache,
                                        const int64_t* slot_mapping, int num_tokens,
                                        int num_kv_heads, int head_size, int block_size,
                                        int64_t block_stride, int dtype, int kv_fp8,
                                        const float* k_scale, const float* v_scale,
                                        mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "reshape_and_cache")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(key && value && key_cache && value_cache && slot_mapping, "reshape_and_cache: null pointer");
  MI355Q_REQUIRE(head_size % (kv_fp8 ? 16 : 8) == 0, "reshape_and_cache: head_size %d", head_size);
  if (kv_fp8) MI355Q_REQUIRE(k_scale && v_scale, "reshape_and_cache: fp8 cache needs k_scale/v_scale");
  if (num_tokens == 0) return MI355Q_OK;
  int threads = num_kv_heads * head_size;
  threads = threads > 512 ? 512 : (threads + 63) / 64 * 64;
  hipStream_t s = (hipStream_t)stream;
  if (kv_fp8) {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((reshape_and_cache_kernel<DT, true>), dim3(num_tokens), dim3(threads), 0, s,
                                           (const uint16_t*)key, (const uint16_t*)value, key_stride, value_stride,
                                           key_cache, value_cache, slot_mapping, num_kv_heads, head_size,
                                           block_size, block_stride, k_scale, v_scale));
  } else {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((reshape_and_cache_kernel<DT, false>), dim3(num_tokens), dim3(threads), 0, s,
                                           (const uint16_t*)key, (const uint16_t*)value, key_stride, value_stride,
                                           key_cache, value_cache, slot_mapping, num_kv_heads, head_size,
                                           block_size, block_stride, k_scale, v_scale));
  }
  return check_launch("reshape_and_cache_kernel");
}

extern "C" int mi355q_reshape_and_cache_flash(const void* key, const void* value, int64_t key_stride,
                                              int64_t value_stride, void* key_cache, void* value_cache,
                                              const int64_t* slot_mapping, int num_tokens, int num_heads,
                                              int head_size, int block_size, int64_t block_stride,
                                              int64_t page_stride, int64_t head_stride, int dtype, int kv_fp8,
                                              const float* k_scale, const float* v_scale, int scales_per_head,
                                              mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "reshape_and_cache_flash")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(key && value && key_cache && value_cache && slot_mapping, "reshape_and_cache_flash: null pointer");
  MI355Q_REQUIRE(head_size > 0 && head_size % 8 == 0, "reshape_and_cache_flash: head_size %d must be a multiple of 8", head_size);
  MI355Q_REQUIRE(num_heads > 0 && block_size > 0, "reshape_and_cache_flash: heads %d, block size %d", num_heads, block_size);
  MI355Q_REQUIRE(key_stride % 8 == 0 && value_stride % 8 == 0 && ((((uintptr_t)key | (uintptr_t)value)) & 15) == 0,
                 "reshape_and_cache_flash: key / value rows must be 16-byte aligned");
  const int eb = kv_fp8 ? 1 : 2, al = kv_fp8 ? 7 : 15;
  MI355Q_REQUIRE((((uintptr_t)key_cache | (uintptr_t)value_cache) & al) == 0 && (block_stride * eb) % (al + 1) == 0 &&
                     (page_stride * eb) % (al + 1) == 0 && (head_stride * eb) % (al + 1) == 0,
                 "reshape_and_cache_flash: cache strides must keep heads %d-byte aligned", al + 1);
  if (kv_fp8) MI355Q_REQUIRE(k_scale && v_scale, "reshape_and_cache_flash: fp8 cache needs k_scale/v_scale");
  int threads = num_heads * (head_size / 8);
  threads = threads > 512 ? 512 : (threads + 63) / 64 * 64;
  hipStream_t s = (hipStream_t)stream;
  const int ss = scales_per_head ? 1 : 0;
  if (kv_fp8) {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((reshape_and_cache_flash_kernel<DT, true>), dim3(num_tokens), dim3(threads), 0, s,
                                           (const uint16_t*)key, (const uint16_t*)value, key_stride, value_stride, key_cache,
                                           value_cache, slot_mapping, num_heads, head_size, block_size, block_stride,
                                           page_stride, head_stride, k_scale, v_scale, ss));
  } else {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((reshape_and_cache_flash_kernel<DT, false>), dim3(num_tokens), dim3(threads), 0, s,
                                           (const uint16_t*)key, (const uint16_t*)value, key_stride, value_stride, key_cache,
                                           value_cache, slot_mapping, num_heads, head_size, block_size, block_stride,
                                           page_stride, head_stride, k_scale, v_scale, ss));
  }
  return check_launch("reshape_and_cache_flash_kernel");
}

extern "C" int mi355q_rope_and_cache(const int64_t* positions, void* query, int64_t query_stride, void* key,
                                     int64_t key_stride, const void* value, int64_t value_stride,
                                     int head_size, int num_heads, int num_kv_heads,
                                     const void* cos_sin_cache, int rot_dim, int cache_dtype, int is_neox,
                                     void* key_cache, void* value_cache, const int64_t* slot_mapping,
                                     int num_tokens, int block_size, int64_t block_stride, int dtype,
                                     int kv_fp8, const float* k_scale, const float* v_scale,
                                     mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "rope_and_cache")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(positions && query && key && value && cos_sin_cache && key_cache && value_cache && slot_mapping,
                 "rope_and_cache: null pointer");
  MI355Q_REQUIRE(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rope_and_cache: rot_dim %d vs head_size %d", rot_dim, head_size);
  MI355Q_REQUIRE(cache_dtype >= MI355Q_BF16 && cache_dtype <= MI355Q_FP32, "rope_and_cache: bad cache dtype");
  MI355Q_REQUIRE(head_size % (kv_fp8 ? 16 : 8) == 0, "rope_and_cache: head_size %d", head_size);
  MI355Q_REQUIRE(num_heads > 0 && num_kv_heads > 0 && block_size > 0, "rope_and_cache: bad sizes");
  if (kv_fp8) MI355Q_REQUIRE(k_scale && v_scale, "rope_and_cache: fp8 cache needs k_scale/v_scale");
  hipStream_t s = (hipStream_t)stream;
  const bool vec = is_neox && rot_dim == head_size && head_size % 16 == 0 && query_stride % 4 == 0 && key_stride % 4 == 0 &&
                   value_stride % 4 == 0 && (((uintptr_t)query | (uintptr_t)key | (uintptr_t)value | (uintptr_t)key_cache) & 7) == 0 &&
                   !env_flag("MI355Q_ROPE_SCALAR");
  if (vec) {
    const int items = (num_heads + num_kv_heads) * head_size / 8 + num_kv_heads * head_size / 4;
    const int vthreads = items >= 1024 ? 1024 : (items + 63) / 64 * 64;
#define MI355Q_RCV(F8)                                                                                              \
  DISPATCH_H16(dtype, hipLaunchKernelGGL((rope_and_cache_vec_kernel<DT, F8>), dim3(num_tokens), dim3(vthreads), 0, s, \
                                         positions, (uint16_t*)query, query_stride, (uint16_t*)key, key_stride,        \
                                         (const uint16_t*)value, value_stride, head_size, num_heads, num_kv_heads,    \
                                         cos_sin_cache, cache_dtype, key_cache, value_cache, slot_mapping, block_size, \
                                         block_stride, k_scale, v_scale))
    if (kv_fp8) { MI355Q_RCV(true); } else { MI355Q_RCV(false); }
#undef MI355Q_RCV
    return check_launch("rope_and_cache_vec_kernel");
  }
  int threads = (num_heads + num_kv_heads) * rot_dim / 2;
  threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : (threads + 63) / 64 * 64);
#define MI355Q_RC(F8)                                                                                            \
  DISPATCH_H16(dtype, hipLaunchKernelGGL((rope_and_cache_kernel<DT, F8>), dim3(num_tokens), dim3(threads), 0, s,   \
                                         positions, (uint16_t*)query, query_stride, (uint16_t*)key, key_stride,     \
                                         (const uint16_t*)value, value_stride, head_size, num_heads, num_kv_heads, \
                                         cos_sin_cache, rot_dim, cache_dtype, is_neox, key_cache, value_cache,      \
                                         slot_mapping, block_size, block_stride, k_scale, v_scale))
  if (kv_fp8) { MI355Q_RC(true); } else { MI355Q_RC(false); }
#undef MI355Q_RC
  return check_launch("rope_and_cache_kernel");
}

extern "C" int mi355q_splitk_rope_and_cache(const float* partial, int splits, const void* bias, void* qkv,
                                            const int64_t* positions, int head_size, int num_heads,
                                            int num_kv_heads, const void* cos_sin_cache, int cache_dtype,
                                            void* key_cache, void* value_cache, const int64_t* slot_mapping,
                                            int num_tokens, int block_size, int64_t block_stride, int dtype,
                                            int kv_fp8, const float* k_scale, const float* v_scale,
                                            mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "splitk_rope_and_cache")) return e;
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(partial && qkv && positions && cos_sin_cache && key_cache && value_cache && slot_mapping && splits >= 1,
                 "splitk_rope_and_cache: bad arguments");
  MI355Q_REQUIRE(head_size % 16 == 0 && num_heads > 0 && num_kv_heads > 0 && block_size > 0,
                 "splitk_rope_and_cache: head_size %d must be a multiple of 16", head_size);
  MI355Q_REQUIRE(cache_dtype >= MI355Q_BF16 && cache_dtype <= MI355Q_FP32, "splitk_rope_and_cache: bad cache dtype");
  MI355Q_REQUIRE((((uintptr_t)partial) & 15) == 0 && (((uintptr_t)qkv | (uintptr_t)bias | (uintptr_t)key_cache) & 7) == 0,
                 "splitk_rope_and_cache: partial tiles must be 16-byte aligned, qkv / bias / cache 8-byte aligned");
  if (kv_fp8) MI355Q_REQUIRE(k_scale && v_scale, "splitk_rope_and_cache: fp8 cache needs k_scale/v_scale");
  const int N = (num_heads + 2 * num_kv_heads) * head_size;
  const int items = (num_heads + num_kv_heads) * head_size / 8 + num_kv_heads * head_size / 4;
  const int threads = items >= 1024 ? 1024 : (items + 63) / 64 * 64;
  hipStream_t s = (hipStream_t)stream;
#define MI355Q_SRC(F8)                                                                                              \
  DISPATCH_H16(dtype, hipLaunchKernelGGL((splitk_rope_and_cache_kernel<DT, F8>), dim3(num_tokens), dim3(threads), 0, s, \
                                         partial, splits, (size_t)num_tokens * N, (const uint16_t*)bias,              \
                                         (uint16_t*)qkv, N, positions, head_size, num_heads, num_kv_heads,            \
                                         cos_sin_cache, cache_dtype, key_cache, value_cache, slot_mapping,             \
                                         block_size, block_stride, k_scale, v_scale))
  if (kv_fp8) { MI355Q_SRC(true); } else { MI355Q_SRC(false); }
#undef MI355Q_SRC
  return check_launch("splitk_rope_and_cache_kernel");
}

extern "C" int mi355q_convert_fp8(void* dst, const void* src, int64_t n, float scale, int dtype,
                                  int to_fp8, mi355q_stream_t stream) {
  if (int e = check_h16(dtype, "convert_fp8")) return e;
  MI355Q_REQUIRE(dst && src && n >= 0, "convert_fp8: bad arguments");
  if (n == 0) return MI355Q_OK;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  DISPATCH_H16(dtype, hipLaunchKernelGGL((convert_fp8_kernel<DT>), dim3(grid), dim3(256), 0,
                                         (hipStream_t)stream, dst, src, n, scale, to_fp8));
  return check_launch("convert_fp8_kernel");
}

static int argmax_with_tail(const void* logits, int64_t row_stride, int64_t* ids, int num_rows,
                            int vocab, int dtype, const StepTail& tail, mi355q_stream_t stream) {
  if (num_rows == 0) return MI355Q_OK;
  MI355Q_REQUIRE(logits && ids && vocab > 0, "argmax: bad arguments");
  MI355Q_REQUIRE(dtype >= MI355Q_BF16 && dtype <= MI355Q_FP32, "argmax: bad dtype");
  if (num_rows == 0) return MI355Q_OK;
  hipStream_t s = (hipStream_t)stream;
  const int es = dtype == MI355Q_FP32 ? 4 : 2;
  const int segs = std::min(vocab / ARGMAX_SEG_ELEMS, 64);
  const int bank = (segs >= 2 && num_rows <= ARGMAX_ROWS && ((uintptr_t)logits & 15) == 0 &&
                    (row_stride * es) % 16 == 0 && !env_flag("MI355Q_ARGMAX_NARROW"))
                       ? stream_bank((void*)stream) : -1;
  if (bank >= 0) {
    const dim3 grid(num_rows, segs);
    if (dtype == MI355Q_FP32)
      hipLaunchKernelGGL((argmax_wide_kernel<MI355Q_FP32>), grid, dim3(256), 0, s, logits, row_stride, ids, vocab, segs, bank, tail);
    else if (dtype == MI355Q_BF16)
      hipLaunchKernelGGL((argmax_wide_kernel<MI355Q_BF16>), grid, dim3(256), 0, s, logits, row_stride, ids, vocab, segs, bank, tail);
    else
      hipLaunchKernelGGL((argmax_wide_kernel<MI355Q_FP16>), grid, dim3(256), 0, s, logits, row_stride, ids, vocab, segs, bank, tail);
    return check_launch("argmax_wide_kernel");
  }
  if (dtype == MI355Q_FP32)
    hipLaunchKernelGGL((argmax_kernel<MI355Q_FP32>), dim3(num_rows), dim3(1024), 0, s, logits, row_stride, ids, vocab);
  else if (dtype == MI355Q_BF16)
    hipLaunchKernelGGL((argmax_kernel<MI355Q_BF16>), dim3(num_rows), dim3(1024), 0, s, logits, row_stride, ids, vocab);
  else
    hipLaunchKernelGGL((argmax_kernel<MI355Q_FP16>), dim3(num_rows), dim3(1024), 0, s, logits, row_stride, ids, vocab);
  if (int e = check_launch("argmax_kernel")) return e;
  if (tail.positions || tail.seq_lens || tail.embed) {
    hipLaunchKernelGGL(step_tail_kernel, dim3(num_rows), dim3(256), 0, s, ids, tail);
    return check_launch("step_tail_kernel");
  }
  return MI355Q_OK;
}

extern "C" int mi355q_argmax(const void* logits, int64_t row_stride, int64_t* ids, int num_rows,
                             int vocab, int dtype, mi355q_stream_t stream) {
  const StepTail none{nullptr, nullptr, nullptr, 0, nullptr, 0, 0};
  return argmax_with_tail(logits, row_stride, ids, num_rows, vocab, dtype, none, stream);
}

extern "C" int mi355q_argmax_decode_step(const void* logits, int64_t row_stride, int64_t* ids, int num_rows, int vocab,
                                         int dtype, int64_t* positions, int32_t* seq_lens, const void* embedding,
                                         int64_t embedding_row_stride, void* hidden, int64_t hidden_row_stride,
                                         int hidden_size, mi355q_stream_t stream) {
  if (embedding || hidden) {
    MI355Q_REQUIRE(embedding && hidden && hidden_size > 0 && hidden_size % 8 == 0 && embedding_row_stride % 8 == 0 &&
                       hidden_row_stride % 8 == 0 && ((((uintptr_t)embedding | (uintptr_t)hidden)) & 15) == 0,
                   "argmax_decode_step: embedding / hidden rows must be 16-byte aligned multiples of 8 elements");
  }
  const StepTail tail{positions, seq_lens, (const uint16_t*)embedding, embedding_row_stride, (uint16_t*)hidden,
                      hidden_row_stride, hidden_size};
  return argmax_with_tail(logits, row_stride, ids, num_rows, vocab, dtype, tail, stream);
}

extern "C" int mi355q_apply_repetition_penalties(void* logits, const uint8_t* prompt_mask,
                                                 const uint8_t* output_mask, const void* penalties,
                                                 int num_seqs, int vocab, int dtype,
                                                 mi355q_stream_t stream) {
  MI355Q_REQUIRE(logits && prompt_mask && output_mask && penalties && vocab > 0, "apply_repetition_penalties: bad arguments");
  MI355Q_REQUIRE(dtype >= MI355Q_BF16 && dtype <= MI355Q_FP32, "apply_repetition_penalties: bad dtype");
  if (num_seqs == 0) return MI355Q_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((vocab + 1023) / 1024 < 64 ? (vocab + 1023) / 1024 : 64, num_seqs);
  if (dtype == MI355Q_FP32)
    hipLaunchKernelGGL((repetition_penalty_kernel<MI355Q_FP32>), grid, dim3(1024), 0, s, logits, prompt_mask, output_mask, penalties, vocab);
  else if (dtype == MI355Q_BF16)
    hipLaunchKernelGGL((repetition_penalty_kernel<MI355Q_BF16>), grid, dim3(1024), 0, s, logits, prompt_mask, output_mask, penalties, vocab);
  else
    hipLaunchKernelGGL((repetition_penalty_kernel<MI355Q_FP16>), grid, dim3(1024), 0, s, logits, prompt_mask, output_mask, penalties, vocab);
  return check_launch("repetition_penalty_kernel");
}
