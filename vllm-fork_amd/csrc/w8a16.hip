// fp8 (OCP e4m3fn) weight-only linear for decode: C[M,N] = A[M,K] * (W[N,K] * scale)^T with
// 16-bit activations and fp32 accumulation.  HBM-bound on W (1 byte per weight): every wave
// streams whole weight rows with 16-byte lane loads (16 weights per lane, 1 KiB contiguous per
// wave instruction), converts them with the hardware v_cvt_pk_f32_fp8, multiplies against the
// activations staged in LDS, and a wave64 shuffle tree finishes each row.  Scales are fp32:
// one per output channel, or one per (block_n x block_k) tile of W for block-quantised
// checkpoints (then every lane's 16-element partial product is scaled before it is added).
// The bias joins in fp32 before the single rounding, as in F.linear's GEMM epilogue.
//
// The same kernels serve 8-bit integer weights stored as bytes biased by 128 (GPTQ's uint8b128:
// w = (q - 128) * scale): WT = 1 converts with v_cvt_f32_ubyte and takes the bias out through the
// sum of the lane's activations, so the inner loop stays one convert + one fma per weight.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mi355q {

template <typename DT>
__device__ inline f32x2 unpack2(uint32_t u);
template <>
__device__ inline f32x2 unpack2<BF16>(uint32_t u) {
  f32x2 r;
  r.x = __builtin_bit_cast(float, u << 16);
  r.y = __builtin_bit_cast(float, u & 0xffff0000u);
  return r;
}
template <>
__device__ inline f32x2 unpack2<FP16>(uint32_t u) {
  f32x2 r;
  r.x = FP16::to_f32((uint16_t)(u & 0xffffu));
  r.y = FP16::to_f32((uint16_t)(u >> 16));
  return r;
}

// weights of one dword as two pairs of floats: fp8 values (WT = 0) or raw bytes 0..255 (WT = 1)
template <int WT>
__device__ inline void w4_to_f32(uint32_t w, f32x2& lo, f32x2& hi) {
  if constexpr (WT == 0) {
    lo = fp8x2_to_f32(w, false);
    hi = fp8x2_to_f32(w, true);
  } else {
    lo.x = (float)(w & 0xffu); lo.y = (float)((w >> 8) & 0xffu);
    hi.x = (float)((w >> 16) & 0xffu); hi.y = (float)(w >> 24);
  }
}

// sum of 16 16-bit activations (two u32x4)
template <typename DT>
__device__ inline float sum16(u32x4 x0, u32x4 x1) {
  f32x2 acc = {0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; e++) acc += unpack2<DT>(x0[e]) + unpack2<DT>(x1[e]);
  return acc.x + acc.y;
}

// dot product of 16 8-bit weights (one u32x4) with 16 16-bit activations (two u32x4)
template <typename DT, int WT>
__device__ inline float dot16(u32x4 w, u32x4 x0, u32x4 x1) {
  f32x2 acc = {0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; e++) {
    f32x2 wl, wh;
    w4_to_f32<WT>(w[e], wl, wh);
    const uint32_t xa = e < 2 ? x0[2 * e] : x1[2 * e - 4];
    const uint32_t xb = e < 2 ? x0[2 * e + 1] : x1[2 * e - 3];
    acc += wl * unpack2<DT>(xa);
    acc += wh * unpack2<DT>(xb);
  }
  return acc.x + acc.y;
}

template <typename DT, int MB, bool BLOCK_SCALE, int WT>
__global__ __launch_bounds__(256) void w8a16_skinny_kernel(
    const uint16_t* __restrict__ a, int lda, const uint8_t* __restrict__ w,
    const float* __restrict__ scales, const uint16_t* __restrict__ bias, uint16_t* __restrict__ c,
    int M, int N, int K, int block_n, int block_k, int scale_cols, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);  // [MB][K/8]
  constexpr int R = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.y * MB;
  const int mb = min(MB, M - m0);
  const int kv8 = K >> 3;   // 16-byte activation vectors per row
  const int kv16 = K >> 4;  // 16-byte weight vectors per row

  for (int idx = threadIdx.x; idx < MB * kv8; idx += 256) {
    const int m = idx / kv8, v = idx - m * kv8;
    const int mr = m0 + min(m, mb - 1);
    xs[idx] = reinterpret_cast<const u32x4*>(a + (size_t)mr * lda)[v];
  }
  __syncthreads();

  const int n_begin = blockIdx.x * rows_per_block;
  const int n_end = min(N, n_begin + rows_per_block);
  for (int n = n_begin + wave * R; n < n_end; n += 4 * R) {
    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int m = 0; m < MB; m++) acc[r][m] = 0.f;
    const bool has1 = n + 1 < n_end;
    const int n1 = has1 ? n + 1 : n;
    const u32x4* w0 = reinterpret_cast<const u32x4*>(w + (size_t)n * K);
    const u32x4* w1 = reinterpret_cast<const u32x4*>(w + (size_t)n1 * K);
    const float* s0 = scales + (size_t)(BLOCK_SCALE ? n / block_n : n) * scale_cols;
    const float* s1 = scales + (size_t)(BLOCK_SCALE ? n1 / block_n : n1) * scale_cols;
#pragma unroll 2
    for (int v = lane; v < kv16; v += 64) {
      const u32x4 wa = __builtin_nontemporal_load(w0 + v);
      const u32x4 wb = __builtin_nontemporal_load(w1 + v);
      float sa = 1.f, sb = 1.f;
      if (BLOCK_SCALE) {
        const int kb = (v << 4) / block_k;
        sa = s0[kb];
        sb = s1[kb];
      }
#pragma unroll
      for (int m = 0; m < MB; m++) {
        const u32x4 x0 = xs[m * kv8 + 2 * v], x1 = xs[m * kv8 + 2 * v + 1];
        const float off = WT == 1 ? 128.0f * sum16<DT>(x0, x1) : 0.f;   // sum a (q - 128) = sum a q - 128 sum a
        if (BLOCK_SCALE) {
          acc[0][m] += sa * (dot16<DT, WT>(wa, x0, x1) - off);
          acc[1][m] += sb * (dot16<DT, WT>(wb, x0, x1) - off);
        } else {
          acc[0][m] += dot16<DT, WT>(wa, x0, x1) - off;
          acc[1][m] += dot16<DT, WT>(wb, x0, x1) - off;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int m = 0; m < MB; m++) acc[r][m] = wave_sum(acc[r][m]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (r == 1 && !has1) break;
        const float s = BLOCK_SCALE ? 1.f : (r == 0 ? s0[0] : s1[0]);
        const float b = bias ? DT::to_f32(bias[n + r]) : 0.f;
#pragma unroll
        for (int m = 0; m < MB; m++)
          if (m < mb) c[(size_t)(m0 + m) * N + n + r] = DT::from_f32(acc[r][m] * s + b);
      }
    }
  }
}

// 16 8-bit weights (one u32x4, consecutive k of one output channel) -> two MFMA operand fragments
// of 8 16-bit values each (k 0..7 and k 8..15).  fp8 values and the raw bytes 0..255 are exactly
// representable in both activation dtypes, so nothing is rounded here (the bias of 128 of the
// integer format is taken out afterwards through the sum of the activations).
template <typename DT, int WT>
__device__ inline void w16_to_frags(u32x4 w, u32x4& f0, u32x4& f1) {
#pragma unroll
  for (int e = 0; e < 4; e++) {
    f32x2 lo, hi;
    w4_to_f32<WT>(w[e], lo, hi);
    uint32_t a, b;
    if constexpr (DT::id == MI355Q_BF16) {   // exact: the upper halves of the fp32 words
      a = (__float_as_uint(lo.x) >> 16) | (__float_as_uint(lo.y) & 0xffff0000u);
      b = (__float_as_uint(hi.x) >> 16) | (__float_as_uint(hi.y) & 0xffff0000u);
    } else {
      a = pack2<DT>(lo.x, lo.y);
      b = pack2<DT>(hi.x, hi.y);
    }
    if (e < 2) { f0[2 * e] = a; f0[2 * e + 1] = b; }
    else { f1[2 * e - 4] = a; f1[2 * e - 3] = b; }
  }
}

// 2..16 activation rows per pass: a workgroup owns 16 output channels and its waves split K.  Lane
// (kq, ml) loads the 16 weights k0 + 16 kq .. of channel n0 + ml (one 16-byte load per 64 k),
// expands them once into two MFMA A-operand fragments, and the activation rows come from LDS as B
// operands with the matching k order, so the cost per weight byte does not grow with the row count.
// Block scales are applied to the group's partial sums in fp32; the waves' sums meet in LDS;
// per-channel scales and the bias at the end, one rounding.
template <typename DT, bool BLOCK_SCALE, int WT>
__global__ __launch_bounds__(512) void w8a16_mfma_kernel(
    const uint16_t* __restrict__ a, int lda, const uint8_t* __restrict__ w,
    const float* __restrict__ scales, const uint16_t* __restrict__ bias, uint16_t* __restrict__ c,
    int M, int N, int K, int block_n, int block_k, int scale_cols, int rows_pp, int nw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                            // [nw][16 rows][16 channels]
  float* sc_l = red + (size_t)nw * 256;                                   // [16 channels][scale_cols] (block scales)
  const int sc_pad = BLOCK_SCALE ? (16 * scale_cols + 3) & ~3 : 0;
  uint16_t* xs = reinterpret_cast<uint16_t*>(sc_l + sc_pad);              // [rows_pp][K + 8]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = lane >> 4, ml = lane & 15;
  const int nthreads = nw * 64;
  const int m0 = blockIdx.y * rows_pp;
  const int mb = min(rows_pp, M - m0);
  const int kv = K >> 3, astride = K + 8;
  // four requests in flight per thread: one at a time would pay an L2 round trip per iteration
  for (int base = threadIdx.x; base < mb * kv; base += 4 * nthreads) {
    u32x4 t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = base + j * nthreads;
      if (idx < mb * kv) {
        const int m = idx / kv, v = idx - m * kv;
        t[j] = *reinterpret_cast<const u32x4*>(a + (size_t)(m0 + m) * lda + 8 * v);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = base + j * nthreads;
      if (idx < mb * kv) {
        const int m = idx / kv, v = idx - m * kv;
        *reinterpret_cast<u32x4*>(xs + (size_t)m * astride + 8 * v) = t[j];
      }
    }
  }
  const int n0 = blockIdx.x * 16;
  if (BLOCK_SCALE) {   // the scales of the workgroup's channels wait in LDS: a flush must not wait for memory
    for (int idx = threadIdx.x; idx < 16 * scale_cols; idx += nthreads) {
      const int nl = idx / scale_cols, kb = idx - nl * scale_cols;
      sc_l[idx] = scales[(size_t)(min(n0 + nl, N - 1) / block_n) * scale_cols + kb];
    }
  }
  __syncthreads();
  const int kw = K / nw, kbeg = wave * kw, kend = kbeg + kw;   // this wave's slice: whole 64-k steps
  const uint8_t* wrow = w + (size_t)min(n0 + ml, N - 1) * K + 16 * kq;
  const uint16_t* arow = xs + (size_t)min(ml, mb - 1) * astride + 16 * kq;
  f32x4 tot = f32x4{0.f, 0.f, 0.f, 0.f}, acc = f32x4{0.f, 0.f, 0.f, 0.f};
  // this lane's outputs: activation row ml, channels n0 + 4 kq + r
  const float* srow = sc_l + (size_t)(4 * kq) * scale_cols;
  // WT == 1: acc1 = sum of the activations of the group (all-ones A operand), acc - 128 acc1 = sum a (q - 128)
  f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = u32x4{DT::ones, DT::ones, DT::ones, DT::ones};
  auto centred = [&]() {
    if constexpr (WT == 1) return acc - 128.0f * acc1;
    else return acc;
  };
  auto flush = [&](int kb) {
    const f32x4 d = centred();
#pragma unroll
    for (int r = 0; r < 4; r++) tot[r] = __builtin_fmaf(srow[r * scale_cols + kb], d[r], tot[r]);
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  int k0 = kbeg;
  for (; k0 + 256 <= kend; k0 += 256) {
    u32x4 wv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) wv[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + k0 + 64 * j));
#pragma unroll
    for (int j = 0; j < 4; j++) {
      u32x4 f0, f1;
      w16_to_frags<DT, WT>(wv[j], f0, f1);
      const uint16_t* ap = arow + k0 + 64 * j;
      const u32x4 b0 = *reinterpret_cast<const u32x4*>(ap), b1 = *reinterpret_cast<const u32x4*>(ap + 8);
      acc = mfma16<DT>(f0, b0, acc);
      acc = mfma16<DT>(f1, b1, acc);
      if constexpr (WT == 1) { acc1 = mfma16<DT>(ones, b0, acc1); acc1 = mfma16<DT>(ones, b1, acc1); }
      if (BLOCK_SCALE && (k0 + 64 * j + 64) % block_k == 0) flush((k0 + 64 * j) / block_k);
    }
  }
  for (; k0 < kend; k0 += 64) {
    u32x4 f0, f1;
    w16_to_frags<DT, WT>(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + k0)), f0, f1);
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(arow + k0), b1 = *reinterpret_cast<const u32x4*>(arow + k0 + 8);
    acc = mfma16<DT>(f0, b0, acc);
    acc = mfma16<DT>(f1, b1, acc);
    if constexpr (WT == 1) { acc1 = mfma16<DT>(ones, b0, acc1); acc1 = mfma16<DT>(ones, b1, acc1); }
    if (BLOCK_SCALE && (k0 + 64) % block_k == 0) flush(k0 / block_k);
  }
  if (BLOCK_SCALE && kend % block_k != 0) flush((kend - 1) / block_k);   // a scale block the slice ends inside
  *reinterpret_cast<f32x4*>(red + ((size_t)wave * 16 + ml) * 16 + 4 * kq) = BLOCK_SCALE ? tot : centred();
  __syncthreads();
  for (int it = threadIdx.x; it < mb * 16; it += nthreads) {
    const int m = it >> 4, nl = it & 15, n = n0 + nl;
    if (n >= N) continue;
    float v = 0.f;
    for (int i = 0; i < nw; i++) v += red[((size_t)i * 16 + m) * 16 + nl];
    if (!BLOCK_SCALE) v *= scales[n];
    c[(size_t)(m0 + m) * N + n] = DT::from_f32(v + (bias ? DT::to_f32(bias[n]) : 0.f));
  }
}

// W16[N,K] = round(fp8(W) * scale): the weight the dense-GEMM route multiplies with.
template <typename DT, int WT>
__global__ __launch_bounds__(256) void w8a16_dequantize_kernel(const uint8_t* __restrict__ w,
                                                               const float* __restrict__ scales,
                                                               uint16_t* __restrict__ out, int N, int K,
                                                               int block_n, int block_k, int scale_cols) {
  const size_t vec = (size_t)blockIdx.x * 256 + threadIdx.x;  // 4 weights per thread
  const size_t total = (size_t)N * K / 4;
  if (vec >= total) return;
  const size_t e = vec * 4;
  const int n = (int)(e / K), k = (int)(e - (size_t)n * K);
  const float s = scales[(size_t)(n / block_n) * scale_cols + k / block_k];
  const uint32_t p = reinterpret_cast<const uint32_t*>(w)[vec];
  f32x2 lo, hi;
  w4_to_f32<WT>(p, lo, hi);
  const float b = WT == 1 ? 128.0f : 0.f;
  u32x2 o;
  o.x = pack2<DT>((lo.x - b) * s, (lo.y - b) * s);
  o.y = pack2<DT>((hi.x - b) * s, (hi.y - b) * s);
  reinterpret_cast<u32x2*>(out)[vec] = o;
}

template <typename DT, int WT>
static int launch_w8a16(const void* a, int lda, const void* w, const float* scales, const void* bias,
                        void* c, int M, int N, int K, int block_n, int block_k, hipStream_t stream) {
  const bool block_scale = block_k < K;
  const int scale_cols = block_scale ? (K + block_k - 1) / block_k : 1;
  // several rows: matrix cores (whole 64-k steps per wave; scale blocks must end on 64-k steps)
  if (M >= 2 && K % 64 == 0 && (!block_scale || block_k % 64 == 0) && !env_flag("MI355Q_W8_NO_MFMA")) {
    int nw = 8;
    while (nw > 1 && K % (64 * nw)) nw >>= 1;
    int rows_pp = std::min(M, 16);
    const size_t red_bytes = (size_t)nw * 1024 + (block_scale ? (size_t)((16 * scale_cols + 3) & ~3) * 4 : 0);
    while (rows_pp > 1 && red_bytes + (size_t)rows_pp * (K + 8) * 2 > 144 * 1024) rows_pp--;
    const size_t lds_m = red_bytes + (size_t)rows_pp * (K + 8) * 2;
    if (lds_m <= 144 * 1024) {
      dim3 grid((N + 15) / 16, (M + rows_pp - 1) / rows_pp), block(nw * 64);
      if (block_scale)
        hipLaunchKernelGGL((w8a16_mfma_kernel<DT, true, WT>), grid, block, lds_m, stream, (const uint16_t*)a, lda,
                           (const uint8_t*)w, scales, (const uint16_t*)bias, (uint16_t*)c, M, N, K, block_n, block_k,
                           scale_cols, rows_pp, nw);
      else
        hipLaunchKernelGGL((w8a16_mfma_kernel<DT, false, WT>), grid, block, lds_m, stream, (const uint16_t*)a, lda,
                           (const uint8_t*)w, scales, (const uint16_t*)bias, (uint16_t*)c, M, N, K, 1, block_k,
                           scale_cols, rows_pp, nw);
      return check_launch("w8a16_mfma_kernel");
    }
  }
  int MB = M >= 4 ? 4 : (M >= 2 ? 2 : 1);
  while (MB > 1 && (size_t)MB * K * 2 > 64 * 1024) MB >>= 1;
  const size_t lds = (size_t)MB * K * 2;
  MI355Q_REQUIRE(lds <= 64 * 1024, "w8a16_gemm: K=%d too large for one LDS pass", K);
  int rows_per_block = 64;
  while (rows_per_block > 8 && (N + rows_per_block - 1) / rows_per_block < 2048) rows_per_block >>= 1;
  dim3 grid((N + rows_per_block - 1) / rows_per_block, (M + MB - 1) / MB), block(256);
#define MI355Q_W8(mb, bs)                                                                          \
  hipLaunchKernelGGL((w8a16_skinny_kernel<DT, mb, bs, WT>), grid, block, lds, stream,                  \
                     (const uint16_t*)a, lda, (const uint8_t*)w, scales, (const uint16_t*)bias,    \
                     (uint16_t*)c, M, N, K, block_scale ? block_n : 1, block_k, scale_cols,        \
                     rows_per_block)
  if (block_scale) {
    switch (MB) {
      case 4: MI355Q_W8(4, true); break;
      case 2: MI355Q_W8(2, true); break;
      default: MI355Q_W8(1, true); break;
    }
  } else {
    switch (MB) {
      case 4: MI355Q_W8(4, false); break;
      case 2: MI355Q_W8(2, false); break;
      default: MI355Q_W8(1, false); break;
    }
  }
#undef MI355Q_W8
  return check_launch("w8a16_skinny_kernel");
}

static int check_w8_scales(const char* who, int N, int K, int block_n, int block_k) {
  MI355Q_REQUIRE(block_n >= 1 && block_k >= 16 && block_k % 16 == 0,
                 "%s: scale block (%d, %d) must be >= (1, 16) with block_k a multiple of 16", who, block_n, block_k);
  MI355Q_REQUIRE(block_k < K || block_n == 1,
                 "%s: a scale that spans all of K must be per output channel (block_n == 1), got block_n=%d",
                 who, block_n);
  (void)N;
  return MI355Q_OK;
}

}  // namespace mi355q

using namespace mi355q;

static int w8_gemm_impl(int wt, const void* a, int lda, const void* w, const float* scales,
                        const void* bias, void* c, int M, int N, int K, int block_n,
                        int block_k, int act_dtype, mi355q_stream_t stream) {
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w8a16_gemm: dtype must be bf16 or fp16");
  if (M == 0) return MI355Q_OK;
  MI355Q_REQUIRE(a && w && scales && c && N > 0 && K > 0 && M >= 0, "w8a16_gemm: bad arguments");
  MI355Q_REQUIRE(K % 16 == 0 && lda % 8 == 0 && lda >= K, "w8a16_gemm: K=%d must be a multiple of 16 and lda=%d of 8", K, lda);
  MI355Q_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0, "w8a16_gemm: pointers must be 16-byte aligned");
  if (block_k <= 0 || block_k > K) block_k = K;
  if (block_n <= 0) block_n = 1;
  if (int st = check_w8_scales("w8a16_gemm", N, K, block_n, block_k)) return st;
  hipStream_t st_ = (hipStream_t)stream;
  if (wt == 0) {
    if (act_dtype == MI355Q_BF16) return launch_w8a16<BF16, 0>(a, lda, w, scales, bias, c, M, N, K, block_n, block_k, st_);
    return launch_w8a16<FP16, 0>(a, lda, w, scales, bias, c, M, N, K, block_n, block_k, st_);
  }
  if (act_dtype == MI355Q_BF16) return launch_w8a16<BF16, 1>(a, lda, w, scales, bias, c, M, N, K, block_n, block_k, st_);
  return launch_w8a16<FP16, 1>(a, lda, w, scales, bias, c, M, N, K, block_n, block_k, st_);
}

extern "C" int mi355q_w8a16_gemm(const void* a, int lda, const void* w, const float* scales,
                                 const void* bias, void* c, int M, int N, int K, int block_n,
                                 int block_k, int act_dtype, mi355q_stream_t stream) {
  return w8_gemm_impl(0, a, lda, w, scales, bias, c, M, N, K, block_n, block_k, act_dtype, stream);
}

extern "C" int mi355q_wi8a16_gemm(const void* a, int lda, const void* w, const float* scales,
                                  const void* bias, void* c, int M, int N, int K, int group_k,
                                  int act_dtype, mi355q_stream_t stream) {
  return w8_gemm_impl(1, a, lda, w, scales, bias, c, M, N, K, 1, group_k, act_dtype, stream);
}

static int w8_dequant_impl(int wt, const void* w, const float* scales, void* out, int N, int K,
                           int block_n, int block_k, int act_dtype, mi355q_stream_t stream) {
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w8a16_dequantize: dtype must be bf16 or fp16");
  MI355Q_REQUIRE(w && scales && out && N > 0 && K > 0 && K % 16 == 0, "w8a16_dequantize: bad arguments");
  if (block_k <= 0 || block_k > K) block_k = K;
  if (block_n <= 0) block_n = 1;
  if (int st = check_w8_scales("w8a16_dequantize", N, K, block_n, block_k)) return st;
  const int scale_cols = (K + block_k - 1) / block_k;
  const size_t total = (size_t)N * K / 4;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define MI355Q_DQ(DTT, WTT)                                                                                  \
  hipLaunchKernelGGL((w8a16_dequantize_kernel<DTT, WTT>), grid, block, 0, (hipStream_t)stream, (const uint8_t*)w, \
                     scales, (uint16_t*)out, N, K, block_n, block_k, scale_cols)
  if (wt == 0) { if (act_dtype == MI355Q_BF16) MI355Q_DQ(BF16, 0); else MI355Q_DQ(FP16, 0); }
  else { if (act_dtype == MI355Q_BF16) MI355Q_DQ(BF16, 1); else MI355Q_DQ(FP16, 1); }
#undef MI355Q_DQ
  return check_launch("w8a16_dequantize_kernel");
}

extern "C" int mi355q_w8a16_dequantize(const void* w, const float* scales, void* out, int N, int K,
                                       int block_n, int block_k, int act_dtype,
                                       mi355q_stream_t stream) {
  return w8_dequant_impl(0, w, scales, out, N, K, block_n, block_k, act_dtype, stream);
}

extern "C" int mi355q_wi8a16_dequantize(const void* w, const float* scales, void* out, int N, int K,
                                        int group_k, int act_dtype, mi355q_stream_t stream) {
  return w8_dequant_impl(1, w, scales, out, N, K, 1, group_k, act_dtype, stream);
}
