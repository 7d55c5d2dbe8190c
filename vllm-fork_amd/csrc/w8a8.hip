// fp8 (OCP e4m3fn) activations x fp8 weights: the quantising kernels that produce fp8 activations
// (stand-alone, and fused behind RMSNorm / SiLU-and-multiply) and the scaled matrix product
// C[M,N] = (A[M,K] . W[N,K]^T) * scale_a * scale_b + bias on the fp8 matrix cores.
//
// Quantisers: one workgroup per token row, fp32 arithmetic in the order of the reference kernels
// (which value is rounded to the activation dtype and when decides the fp8 byte at a rounding
// boundary), saturation to +-448 before the hardware round-to-nearest-even conversion.
//
// Scaled product, two kernels:
//   * up to 64 rows (decode): HBM-bound on W.  A workgroup owns 16 R output channels, its eight
//     waves split K in 64-byte units, weights go from memory straight into MFMA operands (a lane's
//     16-byte load is two 16x16x32 fragments), the activation fragments come from L2 the same
//     way; partial sums meet in LDS, scales and bias are applied once.
//   * more rows (prompts): 128 x 128 output tiles, 128-byte K steps staged by LDS-DMA into a
//     double-buffered, XOR-swizzled LDS image, v_mfma_scale_f32_32x32x64_f8f6f4 with unit block
//     scales (twice the bf16 rate).
#include <algorithm>

#include "common.h"

namespace mi355q {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

static __device__ __attribute__((aligned(16))) unsigned char g_zero16[16];   // source of K-tail chunks

// ---- element traits of the tensors that get quantised -------------------------------------------
struct InBF16 {
  typedef uint16_t T;
  static constexpr bool is16 = true;
  __device__ static inline float f(T v) { return BF16::to_f32(v); }
  __device__ static inline T r(float v) { return BF16::from_f32(v); }
};
struct InFP16 {
  typedef uint16_t T;
  static constexpr bool is16 = true;
  __device__ static inline float f(T v) { return FP16::to_f32(v); }
  __device__ static inline T r(float v) { return FP16::from_f32(v); }
};
struct InF32 {
  typedef float T;
  static constexpr bool is16 = false;
  __device__ static inline float f(T v) { return v; }
  __device__ static inline T r(float v) { return v; }
};

// saturate, then round to nearest even (the order of scaled_fp8_conversion,
// csrc/quantization/w8a8/fp8/common.cuh:57-77; a NaN saturates to 448 through fminf)
__device__ inline float sat448(float x) { return __builtin_fmaxf(-448.0f, __builtin_fminf(x, 448.0f)); }
__device__ inline uint8_t enc1(float x) {
  const float r = sat448(x);
  return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(r, r, 0, false) & 0xff);
}
__device__ inline uint32_t enc4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(a), sat448(b), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(c), sat448(d), w, true);
  return (uint32_t)w;
}

__device__ inline float block_max_256(float v, float* scratch) {
  v = wave_max(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float t = scratch[0];
  for (int i = 1; i < nw; i++) t = fmaxf(t, scratch[i]);
  return t;
}
__device__ inline float block_sum_n(float v, float* scratch) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; i++) t += scratch[i];
  return t;
}

template <typename IT>
__device__ inline bool row_vec_ok(const void* in, const void* out, int H) {
  return IT::is16 && H % 8 == 0 && (((uintptr_t)in & 15) == 0) && (((uintptr_t)out & 7) == 0);
}

// eight 16-bit values of one 16-byte vector as floats
template <typename IT>
__device__ inline void unpack8(u32x4 v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; e++) {
    f[2 * e] = IT::f((uint16_t)(v[e] & 0xffff));
    f[2 * e + 1] = IT::f((uint16_t)(v[e] >> 16));
  }
}

// ---------------------------------------------------------------------------
// scaled_fp8_quant: given scales (per tensor / token / channel / 2-D groups)
// ---------------------------------------------------------------------------
// out[t, n] = fp8(x[t, n] * (1 / scale[(t / group_m) * si + (n / group_n) * sj]))
template <typename IT>
__global__ __launch_bounds__(256) void fp8_quant_static_kernel(
    uint8_t* __restrict__ out, int64_t ldo, const typename IT::T* __restrict__ in, int64_t ldi,
    const float* __restrict__ scale, int H, int group_m, int group_n, int64_t si, int64_t sj) {
  const int64_t t = blockIdx.x;
  const typename IT::T* x = in + t * ldi;
  uint8_t* y = out + t * ldo;
  const float* srow = scale + (t / group_m) * si;
  if (row_vec_ok<IT>(x, y, H) && group_n % 8 == 0) {
    for (int i = threadIdx.x; i < H / 8; i += 256) {
      const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
      const float inv = 1.0f / srow[(int64_t)((8 * i) / group_n) * sj];
      float f[8];
      unpack8<IT>(v, f);
      u32x2 o;
      o.x = enc4(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv);
      o.y = enc4(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv);
      reinterpret_cast<u32x2*>(y)[i] = o;
    }
  } else {
    for (int i = threadIdx.x; i < H; i += 256) {
      const float inv = 1.0f / srow[(int64_t)(i / group_n) * sj];
      y[i] = enc1(IT::f(x[i]) * inv);
    }
  }
}

template <typename IT>
__device__ inline float row_absmax(const typename IT::T* x, int H, bool vec) {
  float m = 0.f;
  if (vec) {
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x) {
      float f[8];
      unpack8<IT>(reinterpret_cast<const u32x4*>(x)[i], f);
#pragma unroll
      for (int e = 0; e < 8; e++) m = fmaxf(m, fabsf(f[e]));
    }
  } else {
    for (int i = threadIdx.x; i < H; i += blockDim.x) m = fmaxf(m, fabsf(IT::f(x[i])));
  }
  return m;
}

// scale[0] = max over the tensor of |x| / 448 (scale[0] starts at zero; non-negative floats order as
// their bit patterns, so the integer atomic max is the float max)
template <typename IT>
__global__ __launch_bounds__(256) void fp8_absmax_kernel(float* __restrict__ scale,
                                                         const typename IT::T* __restrict__ in,
                                                         int64_t ldi, int H) {
  __shared__ float scratch[4];
  const typename IT::T* x = in + (int64_t)blockIdx.x * ldi;
  const bool vec = IT::is16 && H % 8 == 0 && (((uintptr_t)x & 15) == 0);
  const float m = block_max_256(row_absmax<IT>(x, H, vec), scratch);
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(scale), __float_as_int(m / 448.0f));
}

// scale[t] = max(min(absmax_t, ub) / 448, 1 / (448 * 512)); out = fp8(x / scale[t])
template <typename IT>
__global__ __launch_bounds__(256) void fp8_quant_per_token_kernel(
    uint8_t* __restrict__ out, int64_t ldo, float* __restrict__ scales,
    const typename IT::T* __restrict__ in, int64_t ldi, const float* __restrict__ scale_ub, int H) {
  __shared__ float scratch[4];
  const int64_t t = blockIdx.x;
  const typename IT::T* x = in + t * ldi;
  uint8_t* y = out + t * ldo;
  const bool vec = row_vec_ok<IT>(x, y, H);
  float m = block_max_256(row_absmax<IT>(x, H, vec), scratch);
  if (scale_ub) m = fminf(m, *scale_ub);
  const float s = fmaxf(m / 448.0f, 1.0f / (448.0f * 512.0f));
  if (threadIdx.x == 0) scales[t] = s;
  if (vec) {
    for (int i = threadIdx.x; i < H / 8; i += 256) {
      float f[8];
      unpack8<IT>(reinterpret_cast<const u32x4*>(x)[i], f);
      u32x2 o;
      o.x = enc4(f[0] / s, f[1] / s, f[2] / s, f[3] / s);
      o.y = enc4(f[4] / s, f[5] / s, f[6] / s, f[7] / s);
      reinterpret_cast<u32x2*>(y)[i] = o;
    }
  } else {
    for (int i = threadIdx.x; i < H; i += 256) y[i] = enc1(IT::f(x[i]) / s);
  }
}

// ---------------------------------------------------------------------------
// RMSNorm -> fp8 with a given per-tensor scale
// ---------------------------------------------------------------------------
// ADD: z = round(x + residual) replaces the residual and is what gets normalised.
// out = fp8(float(round(z * rs * w)) * (1 / scale)): the normalised value passes through the
// activation dtype, as it does between rms_norm and static_scaled_fp8_quant.
template <typename DT, bool ADD, int NT>
__global__ __launch_bounds__(NT) void rms_norm_static_fp8_quant_kernel(
    uint8_t* __restrict__ out, const uint16_t* __restrict__ in, int64_t ldi,
    uint16_t* __restrict__ residual, const uint16_t* __restrict__ w,
    const float* __restrict__ scale, float eps, int H) {
  __shared__ float scratch[NT / 64];
  const uint16_t* x = in + (int64_t)blockIdx.x * ldi;
  uint16_t* r = ADD ? residual + (int64_t)blockIdx.x * H : nullptr;
  uint8_t* y = out + (int64_t)blockIdx.x * H;
  float ss = 0.f;
  for (int i = threadIdx.x; i < H; i += NT) {
    float zf;
    if constexpr (ADD) {
      const uint16_t z = DT::from_f32(DT::to_f32(x[i]) + DT::to_f32(r[i]));
      r[i] = z;
      zf = DT::to_f32(z);
    } else {
      zf = DT::to_f32(x[i]);
    }
    ss += zf * zf;
  }
  const float rs = rsqrtf(block_sum_n(ss, scratch) / (float)H + eps);
  const float inv = 1.0f / *scale;
  const uint16_t* src = ADD ? r : x;
  for (int i = threadIdx.x; i < H; i += NT) {
    const float n = DT::to_f32(DT::from_f32(DT::to_f32(src[i]) * rs * DT::to_f32(w[i])));
    y[i] = enc1(n * inv);
  }
}

// ---------------------------------------------------------------------------
// RMSNorm (optionally after a residual add) -> fp8 with one dynamic scale per token
// ---------------------------------------------------------------------------
// x = float(in) + float(residual) stays unrounded for the statistics and the norm (the rounded sum
// is what is stored as the new residual); y = round(round(x * rms) * w); scale = max(min(max|y|,
// ub) / 448, 1 / (448 * 512)); out = fp8(y / scale).
template <typename DT, bool HAS_RES, int NT>
__global__ __launch_bounds__(NT) void rms_norm_dynamic_per_token_quant_kernel(
    uint8_t* __restrict__ out, float* __restrict__ scales, const uint16_t* __restrict__ in,
    int64_t ldi, uint16_t* __restrict__ residual, const uint16_t* __restrict__ w,
    const float* __restrict__ scale_ub, float eps, int H) {
  __shared__ float scratch[NT / 64];
  const uint16_t* x = in + (int64_t)blockIdx.x * ldi;
  uint16_t* r = HAS_RES ? residual + (int64_t)blockIdx.x * H : nullptr;
  uint8_t* y = out + (int64_t)blockIdx.x * H;
  auto value = [&](int i) {
    float v = DT::to_f32(x[i]);
    if constexpr (HAS_RES) v += DT::to_f32(r[i]);
    return v;
  };
  float ss = 0.f;
  for (int i = threadIdx.x; i < H; i += NT) {
    const float v = value(i);
    ss += v * v;
  }
  const float rms = rsqrtf(block_sum_n(ss, scratch) / (float)H + eps);
  auto normed = [&](float v, int i) {
    return DT::to_f32(DT::from_f32(DT::to_f32(DT::from_f32(v * rms)) * DT::to_f32(w[i])));
  };
  float m = 0.f;
  for (int i = threadIdx.x; i < H; i += NT) m = fmaxf(m, fabsf(normed(value(i), i)));
  m = block_max_256(m, scratch);
  if (scale_ub) m = fminf(m, *scale_ub);
  const float s = fmaxf(m / 448.0f, 1.0f / (448.0f * 512.0f));
  if (threadIdx.x == 0) scales[blockIdx.x] = s;
  for (int i = threadIdx.x; i < H; i += NT) {
    const float v = value(i);
    if constexpr (HAS_RES) r[i] = DT::from_f32(v);
    y[i] = enc1(normed(v, i) / s);
  }
}

// ---------------------------------------------------------------------------
// SiLU(gate) * up -> fp8 with a given per-tensor scale
// ---------------------------------------------------------------------------
// out = fp8(float(round(round(silu(g)) * u)) * (1 / scale)); input rows are [gate(d) | up(d)].
template <typename DT>
__global__ __launch_bounds__(256) void silu_and_mul_quant_kernel(uint8_t* __restrict__ out,
                                                                 const uint16_t* __restrict__ in,
                                                                 const float* __restrict__ scale, int d) {
  const uint16_t* g = in + (int64_t)blockIdx.x * 2 * d;
  const uint16_t* u = g + d;
  uint8_t* y = out + (int64_t)blockIdx.x * d;
  const float inv = 1.0f / *scale;
  auto one = [&](float gv, float uv) {
    const float act = DT::to_f32(DT::from_f32(gv / (1.0f + expf(-gv))));
    return DT::to_f32(DT::from_f32(act * uv)) * inv;
  };
  if (d % 8 == 0 && (((uintptr_t)g | (uintptr_t)u) & 15) == 0 && ((uintptr_t)y & 7) == 0) {
    for (int i = threadIdx.x; i < d / 8; i += 256) {
      const u32x4 gv = reinterpret_cast<const u32x4*>(g)[i], uv = reinterpret_cast<const u32x4*>(u)[i];
      float f[8];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        f[2 * e] = one(DT::to_f32((uint16_t)(gv[e] & 0xffff)), DT::to_f32((uint16_t)(uv[e] & 0xffff)));
        f[2 * e + 1] = one(DT::to_f32((uint16_t)(gv[e] >> 16)), DT::to_f32((uint16_t)(uv[e] >> 16)));
      }
      u32x2 o;
      o.x = enc4(f[0], f[1], f[2], f[3]);
      o.y = enc4(f[4], f[5], f[6], f[7]);
      reinterpret_cast<u32x2*>(y)[i] = o;
    }
  } else {
    for (int i = threadIdx.x; i < d; i += 256) y[i] = enc1(one(DT::to_f32(g[i]), DT::to_f32(u[i])));
  }
}

// ---------------------------------------------------------------------------
// per_token_group_quant_fp8: one scale per (token, group of `gs` consecutive elements), the activations of the
// block-scaled product (csrc/libtorch_stable/quantization/w8a8/fp8/per_token_group_quant.cu:40-160):
//   absmax = max(eps, max |x|);  scale = absmax / 448;  out = fp8(clamp(x / scale, -448, 448))
// Sixteen lanes per group, eight elements per lane and pass (a 128-wide group is one pass); sixteen groups per
// workgroup.  The scale lands at scales[t * ss_t + g * ss_g], so row- and column-major scale tensors both work.
// ---------------------------------------------------------------------------
template <typename IT>
__global__ __launch_bounds__(256) void fp8_quant_per_token_group_kernel(
    uint8_t* __restrict__ out, int64_t ldo, const typename IT::T* __restrict__ in, int64_t ldi,
    float* __restrict__ scales, int64_t ss_t, int64_t ss_g, int64_t groups, int groups_per_row, int gs, float eps) {
  const int sub = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int64_t gid = (int64_t)blockIdx.x * 16 + sub;
  if (gid >= groups) return;                          // whole 16-lane groups leave together
  const int64_t t = gid / groups_per_row;
  const int g = (int)(gid - t * groups_per_row);
  const typename IT::T* x = in + t * ldi + (int64_t)g * gs;
  uint8_t* y = out + t * ldo + (int64_t)g * gs;
  const bool vec = IT::is16 && gs % 8 == 0 && (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 7) == 0);
  float m = eps;
  if (vec) {
    for (int i = l16; i < gs / 8; i += 16) {
      float f[8];
      unpack8<IT>(reinterpret_cast<const u32x4*>(x)[i], f);
#pragma unroll
      for (int e = 0; e < 8; e++) m = fmaxf(m, fabsf(f[e]));
    }
  } else {
    for (int i = l16; i < gs; i += 16) m = fmaxf(m, fabsf(IT::f(x[i])));
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
  const float sc = m / 448.0f;
  if (l16 == 0) scales[t * ss_t + (int64_t)g * ss_g] = sc;
  if (vec) {
    for (int i = l16; i < gs / 8; i += 16) {
      float f[8];
      unpack8<IT>(reinterpret_cast<const u32x4*>(x)[i], f);
      u32x2 o;
      o.x = enc4(f[0] / sc, f[1] / sc, f[2] / sc, f[3] / sc);
      o.y = enc4(f[4] / sc, f[5] / sc, f[6] / sc, f[7] / sc);
      reinterpret_cast<u32x2*>(y)[i] = o;
    }
  } else {
    for (int i = l16; i < gs; i += 16) y[i] = enc1(IT::f(x[i]) / sc);
  }
}

// ---------------------------------------------------------------------------
// scaled product, up to 64 rows
// ---------------------------------------------------------------------------
// Lane (kq = lane / 16, ml = lane % 16) of a wave holds, per 64-byte K unit, bytes 16 kq .. 16 kq + 15
// of weight row n0 + 16 r + ml (r < R) and of activation row 16 t + ml (t < T): each 16-byte load
// is two operands of v_mfma_f32_16x16x32_fp8_fp8 with the same k order on both sides.  Waves take
// units round-robin, so the eight waves of a workgroup sweep adjacent 64-byte pieces of the same
// rows together.  D[channel 4 kq + j][token ml] per (r, t).
template <typename ODT, int R, int T>
__global__ __launch_bounds__(512) void fp8_mm_skinny_kernel(
    const uint8_t* __restrict__ a, int64_t lda, const uint8_t* __restrict__ w, int64_t ldw,
    const float* __restrict__ scale_a, int sa_stride, const float* __restrict__ scale_b, int sb_stride,
    const uint16_t* __restrict__ bias, uint16_t* __restrict__ c, int64_t ldc, int M, int N, int K) {
  constexpr int NW = 8, U = R * T > 4 ? 2 : 4;   // waves per workgroup, K units in flight per wave
  __shared__ __attribute__((aligned(16))) float red[NW][R * T][16][16];   // [wave][tile][token][channel]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = lane >> 4, ml = lane & 15;
  const int n0 = blockIdx.x * 16 * R;
  const uint8_t* wp[R];
  const uint8_t* ap[T];
#pragma unroll
  for (int r = 0; r < R; r++) wp[r] = w + (int64_t)min(n0 + 16 * r + ml, N - 1) * ldw;
#pragma unroll
  for (int t = 0; t < T; t++) ap[t] = a + (int64_t)min(16 * t + ml, M - 1) * lda;
  f32x4 acc[R][T];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int t = 0; t < T; t++) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int units = (K + 63) >> 6;
  // every workgroup starts its sweep of K at a different unit, so that the chip's requests at any one
  // time are spread over K (rows lie a row stride apart, often a power of two)
  const int rot = (int)((blockIdx.x * 37u) % (unsigned)units);
  for (int u0 = wave; u0 < units; u0 += NW * U) {
    u32x4 wv[U][R], av[U][T];
#pragma unroll
    for (int j = 0; j < U; j++) {
      int u = u0 + j * NW;
      const bool in_range = u < units;
      u += rot;
      if (u >= units) u -= units;
      const bool live = in_range && 64 * u + 16 * kq < K;
      const int off = live ? 64 * u + 16 * kq : 0;   // clamped to the row's start: the loads stay unconditional
#pragma unroll
      for (int r = 0; r < R; r++) wv[j][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + off));
#pragma unroll
      for (int t = 0; t < T; t++) av[j][t] = *reinterpret_cast<const u32x4*>(ap[t] + off);
      if (!live) {
#pragma unroll
        for (int t = 0; t < T; t++) av[j][t] = u32x4{0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; r++) wv[j][r] = u32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int j = 0; j < U; j++)
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int t = 0; t < T; t++) {
          const long w0 = (long)(((uint64_t)wv[j][r][1] << 32) | wv[j][r][0]);
          const long w1 = (long)(((uint64_t)wv[j][r][3] << 32) | wv[j][r][2]);
          const long a0 = (long)(((uint64_t)av[j][t][1] << 32) | av[j][t][0]);
          const long a1 = (long)(((uint64_t)av[j][t][3] << 32) | av[j][t][2]);
          acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w0, a0, acc[r][t], 0, 0, 0);
          acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w1, a1, acc[r][t], 0, 0, 0);
        }
  }
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int t = 0; t < T; t++)
      *reinterpret_cast<f32x4*>(&red[wave][r * T + t][ml][4 * kq]) = acc[r][t];
  __syncthreads();
  for (int it = threadIdx.x; it < R * T * 256; it += NW * 64) {
    const int rt = it >> 8, tok = (it >> 4) & 15, ch = it & 15;
    const int r = rt / T, t = rt - r * T;
    const int m = 16 * t + tok, n = n0 + 16 * r + ch;
    if (m >= M || n >= N) continue;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NW; i++) v += red[i][rt][tok][ch];
    v = v * scale_a[(int64_t)m * sa_stride] * scale_b[(int64_t)n * sb_stride];
    if (bias) v += ODT::to_f32(bias[n]);
    c[(int64_t)m * ldc + n] = ODT::from_f32(v);
  }
}

// ---------------------------------------------------------------------------
// scaled product, many rows
// ---------------------------------------------------------------------------
// 128 x 128 output tile per workgroup of four waves (2 x 2, 64 x 64 each = 2 x 2 MFMA tiles of
// 32 x 32), K in steps of 128 bytes.  LDS image of a step: [128 rows][128 bytes] per operand; the
// 16-byte chunk c of row r sits at chunk position c ^ ((r >> 1) & 7), so the 32 rows a
// ds_read_b128 touches at one logical chunk fall on 16 distinct 16-byte slots of the 256-byte bank
// row in each of its lane groups.  LDS-DMA writes lane-linear, so the permutation is applied to
// the source address.  Two buffers: step s + 1 streams in while step s is multiplied.
constexpr int MM_BM = 128, MM_BN = 128, MM_BK = 128;

template <typename ODT>
__global__ __launch_bounds__(256) void fp8_mm_tiled_kernel(
    const uint8_t* __restrict__ a, int64_t lda, const uint8_t* __restrict__ w, int64_t ldw,
    const float* __restrict__ scale_a, int sa_stride, const float* __restrict__ scale_b, int sb_stride,
    const uint16_t* __restrict__ bias, uint16_t* __restrict__ c, int64_t ldc, int M, int N, int K,
    int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2][2][MM_BM * MM_BK];   // [buffer][A, W]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // workgroups that share an XCD (ids equal mod 8) walk consecutive tiles with the row tile fastest,
  // so a weight panel is fetched into that XCD's L2 once
  const int nwg = tiles_m * tiles_n, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * MM_BM, n0 = tn * MM_BN;
  // staging: a wave instruction moves 8 rows x 128 bytes; 16 per operand and step, 4 per wave
  const int srow = lane >> 3, spc = lane & 7;
  const uint8_t* asrc[4];
  const uint8_t* wsrc[4];
  int schunk[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + srow;
    schunk[i] = 16 * (spc ^ ((row >> 1) & 7));
    asrc[i] = a + (int64_t)min(m0 + row, M - 1) * lda + schunk[i];
    wsrc[i] = w + (int64_t)min(n0 + row, N - 1) * ldw + schunk[i];
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool live = k0 + schunk[i] < K;
      const uint8_t* pa = live ? asrc[i] + k0 : g_zero16;
      const uint8_t* pw = live ? wsrc[i] + k0 : g_zero16;
      const int dst = (wave * 4 + i) * 1024;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)pa,
                                       (void __attribute__((address_space(3)))*)(&lds[buf][0][dst]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)pw,
                                       (void __attribute__((address_space(3)))*)(&lds[buf][1][dst]), 16, 0, 0);
    }
  };
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  // fragment of k-step ks (64 bytes) of tile row `row`: bytes 64 ks + 32 fh .. + 31 = chunks 4 ks + 2 fh, + 1
  auto frag = [&](int buf, int op, int row, int ks) {
    const unsigned char* base = &lds[buf][op][row * MM_BK];
    const int sw = (row >> 1) & 7, c0 = 4 * ks + 2 * fh;
    const u32x4 lo = *reinterpret_cast<const u32x4*>(base + 16 * (c0 ^ sw));
    const u32x4 hi = *reinterpret_cast<const u32x4*>(base + 16 * ((c0 + 1) ^ sw));
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  const int steps = (K + MM_BK - 1) / MM_BK;
  stage(0, 0);
  for (int s = 0; s < steps; s++) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // step s has landed for every wave; buffer buf ^ 1 is free
    if (s + 1 < steps) stage(buf ^ 1, (s + 1) * MM_BK);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      i32x8 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) af[i] = frag(buf, 0, wr * 64 + i * 32 + fr, ks);
#pragma unroll
      for (int j = 0; j < 2; j++) wf[j] = frag(buf, 1, wc * 64 + j * 32 + fr, ks);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], wf[j], acc[i][j], 0, 0, 0,
                                                                      0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  // D[row = (e & 3) + 8 (e >> 2) + 4 fh][col = fr]: rows are tokens, columns channels.  Row scales are
  // requested up front with clamped indices, so no store waits on a load of its own.
  float sa[2][16];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
      sa[i][e] = scale_a[(int64_t)min(m, M - 1) * sa_stride];
    }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int n = n0 + wc * 64 + j * 32 + fr;
    const int nc = min(n, N - 1);
    const float sb = scale_b[(int64_t)nc * sb_stride];
    const float bv = bias ? ODT::to_f32(bias[nc]) : 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        float v = acc[i][j][e] * sa[i][e] * sb;
        if (bias) v += bv;
        if (m < M && n < N) c[(int64_t)m * ldc + n] = ODT::from_f32(v);
      }
  }
}

// ---------------------------------------------------------------------------
// block-scaled product (DeepSeek-V3 / Qwen3 style fp8 checkpoints): weights carry one fp32 scale per
// (block_n x 128) tile, activations one per (token, 128-wide group) --
//   C[m, n] = sum over 128-wide k blocks b of (A[m, b] . W[n, b]) * as[m, b] * ws[n / block_n, b]
// (vllm/model_executor/layers/quantization/utils/fp8_utils.py:736-817: the fp32 dot of a block is scaled by the
// activation scale, then by the weight scale, and added to the fp32 accumulator).  The scales are arbitrary fp32
// numbers, not powers of two, so they cannot ride in the MFMA's e8m0 block-scale operands: every 128-wide block
// is multiplied on its own into a fresh accumulator and folded into the running sum in fp32.
//
// Up to 64 rows: the arrangement of fp8_mm_skinny_kernel, with waves taking whole 128-byte blocks of K
// round-robin (two 64-byte units: four 16x16x32 products per (weight tile, row tile)).
// ---------------------------------------------------------------------------
template <typename ODT, int R, int T>
__global__ __launch_bounds__(512) void fp8_block_mm_skinny_kernel(
    const uint8_t* __restrict__ a, int64_t lda, const uint8_t* __restrict__ w, int64_t ldw,
    const float* __restrict__ scale_a, int64_t sa_m, int64_t sa_k, const float* __restrict__ scale_b, int64_t sb_n,
    int64_t sb_k, int block_n, const uint16_t* __restrict__ bias, uint16_t* __restrict__ c, int64_t ldc, int M, int N,
    int K) {
  constexpr int NW = 8, U = R * T > 4 ? 1 : 2;   // waves per workgroup, K blocks in flight per wave
  __shared__ __attribute__((aligned(16))) float red[NW][R * T][16][16];   // [wave][tile][token][channel]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = lane >> 4, ml = lane & 15;
  const int n0 = blockIdx.x * 16 * R;
  const int nblk = (N + block_n - 1) / block_n;
  const uint8_t* wp[R];
  const uint8_t* ap[T];
  const float* sap[T];
  const float* sbp[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    wp[r] = w + (int64_t)min(n0 + 16 * r + ml, N - 1) * ldw;
    sbp[r] = scale_b + (int64_t)min((n0 + 16 * r) / block_n, nblk - 1) * sb_n;   // 16 | block_n: one scale per tile
  }
#pragma unroll
  for (int t = 0; t < T; t++) {
    ap[t] = a + (int64_t)min(16 * t + ml, M - 1) * lda;
    sap[t] = scale_a + (int64_t)min(16 * t + ml, M - 1) * sa_m;
  }
  f32x4 tot[R][T];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int t = 0; t < T; t++) tot[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int blocks = (K + 127) >> 7;
  const int rot = (int)((blockIdx.x * 37u) % (unsigned)blocks);   // workgroups start their sweeps of K apart
  for (int b0 = wave; b0 < blocks; b0 += NW * U) {
    u32x4 wv[U][2][R], av[U][2][T];
    float sa[U][T], sb[U][R];
#pragma unroll
    for (int j = 0; j < U; j++) {
      int b = b0 + j * NW;
      const bool in_range = b < blocks;
      b += rot;
      if (b >= blocks) b -= blocks;
      if (!in_range) b = 0;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const bool live = in_range && 128 * b + 64 * h + 16 * kq < K;
        const int off = live ? 128 * b + 64 * h + 16 * kq : 0;   // clamped: the loads stay unconditional
#pragma unroll
        for (int r = 0; r < R; r++) wv[j][h][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[r] + off));
#pragma unroll
        for (int t = 0; t < T; t++) av[j][h][t] = *reinterpret_cast<const u32x4*>(ap[t] + off);
        if (!live) {
#pragma unroll
          for (int t = 0; t < T; t++) av[j][h][t] = u32x4{0, 0, 0, 0};
#pragma unroll
          for (int r = 0; r < R; r++) wv[j][h][r] = u32x4{0, 0, 0, 0};
        }
      }
#pragma unroll
      for (int t = 0; t < T; t++) sa[j][t] = sap[t][(int64_t)b * sa_k];
#pragma unroll
      for (int r = 0; r < R; r++) sb[j][r] = sbp[r][(int64_t)b * sb_k];
    }
#pragma unroll
    for (int j = 0; j < U; j++)
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int t = 0; t < T; t++) {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const long w0 = (long)(((uint64_t)wv[j][h][r][1] << 32) | wv[j][h][r][0]);
            const long w1 = (long)(((uint64_t)wv[j][h][r][3] << 32) | wv[j][h][r][2]);
            const long a0 = (long)(((uint64_t)av[j][h][t][1] << 32) | av[j][h][t][0]);
            const long a1 = (long)(((uint64_t)av[j][h][t][3] << 32) | av[j][h][t][2]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w0, a0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w1, a1, acc, 0, 0, 0);
          }
          // D[channel 4 kq + e][token ml]: the activation scale is the lane's token's, the weight scale the tile's
#pragma unroll
          for (int e = 0; e < 4; e++) tot[r][t][e] += acc[e] * sa[j][t] * sb[j][r];
        }
  }
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int t = 0; t < T; t++)
      *reinterpret_cast<f32x4*>(&red[wave][r * T + t][ml][4 * kq]) = tot[r][t];
  __syncthreads();
  for (int it = threadIdx.x; it < R * T * 256; it += NW * 64) {
    const int rt = it >> 8, tok = (it >> 4) & 15, ch = it & 15;
    const int r = rt / T, t = rt - r * T;
    const int m = 16 * t + tok, n = n0 + 16 * r + ch;
    if (m >= M || n >= N) continue;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NW; i++) v += red[i][rt][tok][ch];
    if (bias) v += ODT::to_f32(bias[n]);
    c[(int64_t)m * ldc + n] = ODT::from_f32(v);
  }
}

// More rows: the 128 x 128 tile of fp8_mm_tiled_kernel.  Its K step is one scale block: the step's four
// 32 x 32 products start from zero and are folded into the running sums with the row's activation scale (staged
// through LDS a step ahead, one value per tile row) and the 32-channel subtile's weight scale (32 | block_n).
template <typename ODT>
__global__ __launch_bounds__(256) void fp8_block_mm_tiled_kernel(
    const uint8_t* __restrict__ a, int64_t lda, const uint8_t* __restrict__ w, int64_t ldw,
    const float* __restrict__ scale_a, int64_t sa_m, int64_t sa_k, const float* __restrict__ scale_b, int64_t sb_n,
    int64_t sb_k, int block_n, const uint16_t* __restrict__ bias, uint16_t* __restrict__ c, int64_t ldc, int M, int N,
    int K, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2][2][MM_BM * MM_BK];   // [buffer][A, W]
  __shared__ __attribute__((aligned(16))) float s_as[2][MM_BM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwg = tiles_m * tiles_n, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * MM_BM, n0 = tn * MM_BN;
  const int srow = lane >> 3, spc = lane & 7;
  const uint8_t* asrc[4];
  const uint8_t* wsrc[4];
  int schunk[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + srow;
    schunk[i] = 16 * (spc ^ ((row >> 1) & 7));
    asrc[i] = a + (int64_t)min(m0 + row, M - 1) * lda + schunk[i];
    wsrc[i] = w + (int64_t)min(n0 + row, N - 1) * ldw + schunk[i];
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool live = k0 + schunk[i] < K;
      const uint8_t* pa = live ? asrc[i] + k0 : g_zero16;
      const uint8_t* pw = live ? wsrc[i] + k0 : g_zero16;
      const int dst = (wave * 4 + i) * 1024;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)pa,
                                       (void __attribute__((address_space(3)))*)(&lds[buf][0][dst]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)pw,
                                       (void __attribute__((address_space(3)))*)(&lds[buf][1][dst]), 16, 0, 0);
    }
  };
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  f32x16 tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) tot[i][j][e] = 0.f;
  auto frag = [&](int buf, int op, int row, int ks) {
    const unsigned char* base = &lds[buf][op][row * MM_BK];
    const int sw = (row >> 1) & 7, c0 = 4 * ks + 2 * fh;
    const u32x4 lo = *reinterpret_cast<const u32x4*>(base + 16 * (c0 ^ sw));
    const u32x4 hi = *reinterpret_cast<const u32x4*>(base + 16 * ((c0 + 1) ^ sw));
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  const int steps = (K + MM_BK - 1) / MM_BK;
  const int nblk = (N + block_n - 1) / block_n;
  // tile row threadIdx.x's activation scale, a step ahead (rows past M repeat the last row, never stored)
  const float* my_as = scale_a + (int64_t)min(m0 + (int)(threadIdx.x & (MM_BM - 1)), M - 1) * sa_m;
  const float* wsp[2];
#pragma unroll
  for (int j = 0; j < 2; j++) wsp[j] = scale_b + (int64_t)min((n0 + wc * 64 + j * 32) / block_n, nblk - 1) * sb_n;
  if (threadIdx.x < MM_BM) s_as[0][threadIdx.x] = my_as[0];
  stage(0, 0);
  for (int s = 0; s < steps; s++) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // step s has landed for every wave; buffer buf ^ 1 is free
    const bool more = s + 1 < steps;
    // requested in front of the next step's tiles, so that storing it at the end of this step waits for nothing else
    const float as_next = my_as[(int64_t)(more ? s + 1 : s) * sa_k];
    const float ws0 = wsp[0][(int64_t)s * sb_k], ws1 = wsp[1][(int64_t)s * sb_k];
    if (more) stage(buf ^ 1, (s + 1) * MM_BK);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      i32x8 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) af[i] = frag(buf, 0, wr * 64 + i * 32 + fr, ks);
#pragma unroll
      for (int j = 0; j < 2; j++) wf[j] = frag(buf, 1, wc * 64 + j * 32 + fr, ks);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], wf[j], acc[i][j], 0, 0, 0,
                                                                      0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    // D[row = (e & 3) + 8 (e >> 2) + 4 fh][col = fr]: rows are tokens, columns channels
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int e4 = 0; e4 < 4; e4++) {
        const f32x4 as4 = *reinterpret_cast<const f32x4*>(&s_as[buf][wr * 64 + i * 32 + 8 * e4 + 4 * fh]);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          tot[i][0][4 * e4 + e] += acc[i][0][4 * e4 + e] * as4[e] * ws0;
          tot[i][1][4 * e4 + e] += acc[i][1][4 * e4 + e] * as4[e] * ws1;
        }
      }
    if (more && threadIdx.x < MM_BM) s_as[buf ^ 1][threadIdx.x] = as_next;
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int n = n0 + wc * 64 + j * 32 + fr;
    const float bv = bias ? ODT::to_f32(bias[min(n, N - 1)]) : 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        float v = tot[i][j][e];
        if (bias) v += bv;
        if (m < M && n < N) c[(int64_t)m * ldc + n] = ODT::from_f32(v);
      }
  }
}

template <typename ODT>
static int launch_fp8_block_mm(const void* a, int64_t lda, const void* w, int64_t ldw, const float* sa, int64_t sa_m,
                               int64_t sa_k, const float* sb, int64_t sb_n, int64_t sb_k, int block_n,
                               const void* bias, void* c, int64_t ldc, int M, int N, int K, hipStream_t stream) {
  const int skinny_max = env_int("MI355Q_FP8_MM_SKINNY_MAX", 64);
  if ((M <= skinny_max && M <= 64) || block_n % 32) {
    if (M > 64) return set_error(MI355Q_EINVAL, "fp8_block_scaled_mm: more than 64 rows need block_n to be a multiple of 32, got %d", block_n);
    const int T = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    int R = 4;
    while (R > 1 && (R * T > 8 || (N + 16 * R - 1) / (16 * R) < 256)) R >>= 1;
#define MI355Q_BSK(r, t)                                                                                           \
  hipLaunchKernelGGL((fp8_block_mm_skinny_kernel<ODT, r, t>), dim3((N + 16 * r - 1) / (16 * r)), dim3(512), 0, stream, \
                     (const uint8_t*)a, lda, (const uint8_t*)w, ldw, sa, sa_m, sa_k, sb, sb_n, sb_k, block_n,         \
                     (const uint16_t*)bias, (uint16_t*)c, ldc, M, N, K)
    if (T == 1) { if (R == 4) MI355Q_BSK(4, 1); else if (R == 2) MI355Q_BSK(2, 1); else MI355Q_BSK(1, 1); }
    else if (T == 2) { if (R == 4) MI355Q_BSK(4, 2); else if (R == 2) MI355Q_BSK(2, 2); else MI355Q_BSK(1, 2); }
    else { if (R >= 2) MI355Q_BSK(2, 4); else MI355Q_BSK(1, 4); }
#undef MI355Q_BSK
    return check_launch("fp8_block_mm_skinny_kernel");
  }
  const int tiles_m = (M + MM_BM - 1) / MM_BM, tiles_n = (N + MM_BN - 1) / MM_BN;
  hipLaunchKernelGGL((fp8_block_mm_tiled_kernel<ODT>), dim3(tiles_m * tiles_n), dim3(256), 0, stream,
                     (const uint8_t*)a, lda, (const uint8_t*)w, ldw, sa, sa_m, sa_k, sb, sb_n, sb_k, block_n,
                     (const uint16_t*)bias, (uint16_t*)c, ldc, M, N, K, tiles_m, tiles_n);
  return check_launch("fp8_block_mm_tiled_kernel");
}

template <typename ODT>
static int launch_fp8_mm(const void* a, int64_t lda, const void* w, int64_t ldw, const float* sa, int sa_n,
                         const float* sb, int sb_n, const void* bias, void* c, int64_t ldc, int M, int N, int K,
                         hipStream_t stream) {
  const int sas = sa_n == 1 ? 0 : 1, sbs = sb_n == 1 ? 0 : 1;
  const int skinny_max = env_int("MI355Q_FP8_MM_SKINNY_MAX", 64);
  if (M <= skinny_max && M <= 64) {
    const int T = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    int R = 4;
    while (R > 1 && (R * T > 8 || (N + 16 * R - 1) / (16 * R) < 256)) R >>= 1;
#define MI355Q_SK(r, t)                                                                                      \
  hipLaunchKernelGGL((fp8_mm_skinny_kernel<ODT, r, t>), dim3((N + 16 * r - 1) / (16 * r)), dim3(512), 0, stream, \
                     (const uint8_t*)a, lda, (const uint8_t*)w, ldw, sa, sas, sb, sbs, (const uint16_t*)bias,   \
                     (uint16_t*)c, ldc, M, N, K)
    if (T == 1) { if (R == 4) MI355Q_SK(4, 1); else if (R == 2) MI355Q_SK(2, 1); else MI355Q_SK(1, 1); }
    else if (T == 2) { if (R == 4) MI355Q_SK(4, 2); else if (R == 2) MI355Q_SK(2, 2); else MI355Q_SK(1, 2); }
    else { if (R >= 2) MI355Q_SK(2, 4); else MI355Q_SK(1, 4); }
#undef MI355Q_SK
    return check_launch("fp8_mm_skinny_kernel");
  }
  const int tiles_m = (M + MM_BM - 1) / MM_BM, tiles_n = (N + MM_BN - 1) / MM_BN;
  hipLaunchKernelGGL((fp8_mm_tiled_kernel<ODT>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, (const uint8_t*)a,
                     lda, (const uint8_t*)w, ldw, sa, sas, sb, sbs, (const uint16_t*)bias, (uint16_t*)c, ldc, M, N,
                     K, tiles_m, tiles_n);
  return check_launch("fp8_mm_tiled_kernel");
}

}  // namespace mi355q

using namespace mi355q;

#define DISPATCH_IN(dtype, ...)                                          \
  do {                                                                   \
    if ((dtype) == MI355Q_BF16) { using IT = InBF16; __VA_ARGS__; }      \
    else if ((dtype) == MI355Q_FP16) { using IT = InFP16; __VA_ARGS__; } \
    else { using IT = InF32; __VA_ARGS__; }                              \
  } while (0)

#define DISPATCH_H16(dtype, ...)                                  \
  do {                                                            \
    if ((dtype) == MI355Q_BF16) { using DT = BF16; __VA_ARGS__; } \
    else { using DT = FP16; __VA_ARGS__; }                        \
  } while (0)

extern "C" int mi355q_static_scaled_fp8_quant(void* out, int64_t out_stride, const void* input,
                                              int64_t input_stride, const float* scale, int num_tokens,
                                              int hidden, int group_m, int group_n, int64_t scale_stride_m,
                                              int64_t scale_stride_n, int dtype, mi355q_stream_t stream) {
  MI355Q_REQUIRE(dtype >= MI355Q_BF16 && dtype <= MI355Q_FP32, "static_scaled_fp8_quant: dtype must be bf16, fp16 or fp32");
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && input && scale && hidden > 0 && num_tokens > 0, "static_scaled_fp8_quant: bad arguments");
  if (group_m <= 0) group_m = num_tokens;
  if (group_n <= 0) group_n = hidden;
  MI355Q_REQUIRE(num_tokens % group_m == 0 && hidden % group_n == 0,
                 "static_scaled_fp8_quant: group shape (%d, %d) does not divide the input (%d, %d)", group_m,
                 group_n, num_tokens, hidden);
  DISPATCH_IN(dtype, hipLaunchKernelGGL((fp8_quant_static_kernel<IT>), dim3(num_tokens), dim3(256), 0,
                                        (hipStream_t)stream, (uint8_t*)out, out_stride,
                                        (const typename IT::T*)input, input_stride, scale, hidden, group_m,
                                        group_n, scale_stride_m, scale_stride_n));
  return check_launch("fp8_quant_static_kernel");
}

extern "C" int mi355q_dynamic_scaled_fp8_quant(void* out, int64_t out_stride, const void* input,
                                               int64_t input_stride, float* scale, int num_tokens, int hidden,
                                               int dtype, mi355q_stream_t stream) {
  MI355Q_REQUIRE(dtype >= MI355Q_BF16 && dtype <= MI355Q_FP32, "dynamic_scaled_fp8_quant: dtype must be bf16, fp16 or fp32");
  MI355Q_REQUIRE(scale, "dynamic_scaled_fp8_quant: scale is null");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(scale, 0, sizeof(float), s) != hipSuccess) return set_error(MI355Q_EHIP, "dynamic_scaled_fp8_quant: memset failed");
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && input && hidden > 0 && num_tokens > 0, "dynamic_scaled_fp8_quant: bad arguments");
  DISPATCH_IN(dtype, hipLaunchKernelGGL((fp8_absmax_kernel<IT>), dim3(num_tokens), dim3(256), 0, s, scale,
                                        (const typename IT::T*)input, input_stride, hidden));
  if (int e = check_launch("fp8_absmax_kernel")) return e;
  DISPATCH_IN(dtype, hipLaunchKernelGGL((fp8_quant_static_kernel<IT>), dim3(num_tokens), dim3(256), 0, s,
                                        (uint8_t*)out, out_stride, (const typename IT::T*)input, input_stride,
                                        (const float*)scale, hidden, num_tokens, hidden, (int64_t)0, (int64_t)0));
  return check_launch("fp8_quant_static_kernel");
}

extern "C" int mi355q_dynamic_per_token_scaled_fp8_quant(void* out, int64_t out_stride, const void* input,
                                                         int64_t input_stride, float* scales,
                                                         const float* scale_ub, int num_tokens, int hidden,
                                                         int dtype, mi355q_stream_t stream) {
  MI355Q_REQUIRE(dtype >= MI355Q_BF16 && dtype <= MI355Q_FP32, "dynamic_per_token_scaled_fp8_quant: dtype must be bf16, fp16 or fp32");
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && input && scales && hidden > 0 && num_tokens > 0, "dynamic_per_token_scaled_fp8_quant: bad arguments");
  DISPATCH_IN(dtype, hipLaunchKernelGGL((fp8_quant_per_token_kernel<IT>), dim3(num_tokens), dim3(256), 0,
                                        (hipStream_t)stream, (uint8_t*)out, out_stride, scales,
                                        (const typename IT::T*)input, input_stride, scale_ub, hidden));
  return check_launch("fp8_quant_per_token_kernel");
}

static int check_norm_quant(const char* who, int dtype, const void* out, const void* input, const void* weight,
                            int hidden) {
  MI355Q_REQUIRE(dtype == MI355Q_BF16 || dtype == MI355Q_FP16, "%s: dtype must be bf16 or fp16", who);
  MI355Q_REQUIRE(out && input && weight && hidden > 0, "%s: bad arguments", who);
  return MI355Q_OK;
}

extern "C" int mi355q_rms_norm_static_fp8_quant(void* out, const void* input, int64_t input_stride,
                                                const void* weight, const float* scale, float epsilon,
                                                int num_tokens, int hidden, int dtype, mi355q_stream_t stream) {
  if (num_tokens == 0) return MI355Q_OK;
  if (int e = check_norm_quant("rms_norm_static_fp8_quant", dtype, out, input, weight, hidden)) return e;
  MI355Q_REQUIRE(scale, "rms_norm_static_fp8_quant: scale is null");
  hipStream_t s = (hipStream_t)stream;
  if (num_tokens < 256) {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((rms_norm_static_fp8_quant_kernel<DT, false, 1024>), dim3(num_tokens), dim3(1024), 0, s,
                                           (uint8_t*)out, (const uint16_t*)input, input_stride, (uint16_t*)nullptr,
                                           (const uint16_t*)weight, scale, epsilon, hidden));
  } else {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((rms_norm_static_fp8_quant_kernel<DT, false, 256>), dim3(num_tokens), dim3(256), 0, s,
                                           (uint8_t*)out, (const uint16_t*)input, input_stride, (uint16_t*)nullptr,
                                           (const uint16_t*)weight, scale, epsilon, hidden));
  }
  return check_launch("rms_norm_static_fp8_quant_kernel");
}

extern "C" int mi355q_fused_add_rms_norm_static_fp8_quant(void* out, const void* input, int64_t input_stride,
                                                          void* residual, const void* weight,
                                                          const float* scale, float epsilon, int num_tokens,
                                                          int hidden, int dtype, mi355q_stream_t stream) {
  if (num_tokens == 0) return MI355Q_OK;
  if (int e = check_norm_quant("fused_add_rms_norm_static_fp8_quant", dtype, out, input, weight, hidden)) return e;
  MI355Q_REQUIRE(scale && residual, "fused_add_rms_norm_static_fp8_quant: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (num_tokens < 256) {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((rms_norm_static_fp8_quant_kernel<DT, true, 1024>), dim3(num_tokens), dim3(1024), 0, s,
                                           (uint8_t*)out, (const uint16_t*)input, input_stride, (uint16_t*)residual,
                                           (const uint16_t*)weight, scale, epsilon, hidden));
  } else {
    DISPATCH_H16(dtype, hipLaunchKernelGGL((rms_norm_static_fp8_quant_kernel<DT, true, 256>), dim3(num_tokens), dim3(256), 0, s,
                                           (uint8_t*)out, (const uint16_t*)input, input_stride, (uint16_t*)residual,
                                           (const uint16_t*)weight, scale, epsilon, hidden));
  }
  return check_launch("fused_add_rms_norm_static_fp8_quant_kernel");
}

extern "C" int mi355q_rms_norm_dynamic_per_token_quant(void* out, float* scales, const void* input,
                                                       int64_t input_stride, const void* weight,
                                                       const float* scale_ub, void* residual, float epsilon,
                                                       int num_tokens, int hidden, int dtype,
                                                       mi355q_stream_t stream) {
  if (num_tokens == 0) return MI355Q_OK;
  if (int e = check_norm_quant("rms_norm_dynamic_per_token_quant", dtype, out, input, weight, hidden)) return e;
  MI355Q_REQUIRE(scales, "rms_norm_dynamic_per_token_quant: scales is null");
  hipStream_t s = (hipStream_t)stream;
#define MI355Q_DPT(RES, NT)                                                                                      \
  DISPATCH_H16(dtype, hipLaunchKernelGGL((rms_norm_dynamic_per_token_quant_kernel<DT, RES, NT>), dim3(num_tokens), \
                                         dim3(NT), 0, s, (uint8_t*)out, scales, (const uint16_t*)input,          \
                                         input_stride, (uint16_t*)residual, (const uint16_t*)weight, scale_ub,    \
                                         epsilon, hidden))
  if (residual) { if (num_tokens < 256) MI355Q_DPT(true, 1024); else MI355Q_DPT(true, 256); }
  else { if (num_tokens < 256) MI355Q_DPT(false, 1024); else MI355Q_DPT(false, 256); }
#undef MI355Q_DPT
  return check_launch("rms_norm_dynamic_per_token_quant_kernel");
}

extern "C" int mi355q_silu_and_mul_quant(void* out, const void* input, const float* scale, int num_tokens,
                                         int d, int dtype, mi355q_stream_t stream) {
  MI355Q_REQUIRE(dtype == MI355Q_BF16 || dtype == MI355Q_FP16, "silu_and_mul_quant: dtype must be bf16 or fp16");
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && input && scale && d > 0, "silu_and_mul_quant: bad arguments");
  DISPATCH_H16(dtype, hipLaunchKernelGGL((silu_and_mul_quant_kernel<DT>), dim3(num_tokens), dim3(256), 0,
                                         (hipStream_t)stream, (uint8_t*)out, (const uint16_t*)input, scale, d));
  return check_launch("silu_and_mul_quant_kernel");
}

extern "C" int mi355q_fp8_scaled_mm(const void* a, int64_t lda, const void* w, int64_t ldw,
                                    const float* scale_a, int scale_a_numel, const float* scale_b,
                                    int scale_b_numel, const void* bias, void* c, int64_t ldc, int M, int N,
                                    int K, int out_dtype, mi355q_stream_t stream) {
  MI355Q_REQUIRE(out_dtype == MI355Q_BF16 || out_dtype == MI355Q_FP16, "fp8_scaled_mm: output dtype must be bf16 or fp16");
  if (M == 0) return MI355Q_OK;
  MI355Q_REQUIRE(a && w && scale_a && scale_b && c && M > 0 && N > 0 && K > 0, "fp8_scaled_mm: bad arguments");
  MI355Q_REQUIRE(K % 16 == 0 && lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K && ldc >= N,
                 "fp8_scaled_mm: K=%d, lda=%lld, ldw=%lld must be multiples of 16 with rows at least K long", K,
                 (long long)lda, (long long)ldw);
  MI355Q_REQUIRE((((uintptr_t)a | (uintptr_t)w) & 15) == 0, "fp8_scaled_mm: operands must be 16-byte aligned");
  MI355Q_REQUIRE(scale_a_numel == 1 || scale_a_numel == M, "fp8_scaled_mm: scale_a has %d elements, expected 1 or M=%d", scale_a_numel, M);
  MI355Q_REQUIRE(scale_b_numel == 1 || scale_b_numel == N, "fp8_scaled_mm: scale_b has %d elements, expected 1 or N=%d", scale_b_numel, N);
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == MI355Q_BF16)
    return launch_fp8_mm<BF16>(a, lda, w, ldw, scale_a, scale_a_numel, scale_b, scale_b_numel, bias, c, ldc, M, N, K, s);
  return launch_fp8_mm<FP16>(a, lda, w, ldw, scale_a, scale_a_numel, scale_b, scale_b_numel, bias, c, ldc, M, N, K, s);
}

extern "C" int mi355q_per_token_group_fp8_quant(void* out, int64_t out_stride, const void* input,
                                                int64_t input_stride, float* scales, int64_t scale_stride_token,
                                                int64_t scale_stride_group, int num_tokens, int hidden,
                                                int group_size, float eps, int dtype, mi355q_stream_t stream) {
  MI355Q_REQUIRE(dtype >= MI355Q_BF16 && dtype <= MI355Q_FP32, "per_token_group_fp8_quant: dtype must be bf16, fp16 or fp32");
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && input && scales && hidden > 0 && num_tokens > 0, "per_token_group_fp8_quant: bad arguments");
  MI355Q_REQUIRE(group_size > 0 && hidden % group_size == 0,
                 "per_token_group_fp8_quant: the last dimension %d must be divisible by group_size %d", hidden, group_size);
  const int gpr = hidden / group_size;
  const int64_t groups = (int64_t)num_tokens * gpr;
  DISPATCH_IN(dtype, hipLaunchKernelGGL((fp8_quant_per_token_group_kernel<IT>), dim3((unsigned)((groups + 15) / 16)), dim3(256), 0,
                                        (hipStream_t)stream, (uint8_t*)out, out_stride, (const typename IT::T*)input,
                                        input_stride, scales, scale_stride_token, scale_stride_group, groups, gpr,
                                        group_size, eps));
  return check_launch("fp8_quant_per_token_group_kernel");
}

extern "C" int mi355q_fp8_block_scaled_mm(const void* a, int64_t lda, const float* a_scales,
                                          int64_t a_scale_stride_token, int64_t a_scale_stride_block, const void* w,
                                          int64_t ldw, const float* w_scales, int64_t w_scale_stride_n,
                                          int64_t w_scale_stride_k, int block_n, int block_k, const void* bias,
                                          void* c, int64_t ldc, int M, int N, int K, int out_dtype,
                                          mi355q_stream_t stream) {
  MI355Q_REQUIRE(out_dtype == MI355Q_BF16 || out_dtype == MI355Q_FP16, "fp8_block_scaled_mm: output dtype must be bf16 or fp16");
  if (M == 0) return MI355Q_OK;
  MI355Q_REQUIRE(a && w && a_scales && w_scales && c && M > 0 && N > 0 && K > 0, "fp8_block_scaled_mm: bad arguments");
  MI355Q_REQUIRE(block_k == 128, "fp8_block_scaled_mm: block_k = %d (128-wide scale blocks along K only)", block_k);
  MI355Q_REQUIRE(block_n > 0 && block_n % 16 == 0, "fp8_block_scaled_mm: block_n = %d must be a multiple of 16", block_n);
  MI355Q_REQUIRE(K % 16 == 0 && lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K && ldc >= N,
                 "fp8_block_scaled_mm: K=%d, lda=%lld, ldw=%lld must be multiples of 16 with rows at least K long", K,
                 (long long)lda, (long long)ldw);
  MI355Q_REQUIRE((((uintptr_t)a | (uintptr_t)w) & 15) == 0, "fp8_block_scaled_mm: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == MI355Q_BF16)
    return launch_fp8_block_mm<BF16>(a, lda, w, ldw, a_scales, a_scale_stride_token, a_scale_stride_block, w_scales,
                                     w_scale_stride_n, w_scale_stride_k, block_n, bias, c, ldc, M, N, K, s);
  return launch_fp8_block_mm<FP16>(a, lda, w, ldw, a_scales, a_scale_stride_token, a_scale_stride_block, w_scales,
                                   w_scale_stride_n, w_scale_stride_k, block_n, bias, c, ldc, M, N, K, s);
}
