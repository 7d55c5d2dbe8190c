// Unquantized skinny linear for decode: C[M,N] = A[M,K] * W[N,K]^T with M small
// (lm_head, and the 16-bit comparator of the W4A16 path).  HBM-bound on W: every wave
// streams whole weight rows with 16-byte lane loads (1 KiB contiguous per wave
// instruction), the activations sit in LDS, products run on v_dot2_f32 with fp32
// accumulation, and a wave64 shuffle tree finishes each row.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "row_ops.h"

namespace mi355q {

// MB = activation rows handled per pass (1, 2, 4, 8); R = weight rows per wave iteration.
template <typename DT, int MB>
__global__ __launch_bounds__(256) void linear_skinny_kernel(const uint16_t* __restrict__ a, int lda,
                                                            const uint16_t* __restrict__ w,
                                                            const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ c, int M, int N,
                                                            int K, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);  // [MB][K/8]
  constexpr int R = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.y * MB;
  const int mb = min(MB, M - m0);
  const int kv = K >> 3;  // 16-byte vectors per row

  for (int idx = threadIdx.x; idx < MB * kv; idx += 256) {
    const int m = idx / kv, v = idx - m * kv;
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
    const u32x4* w0 = reinterpret_cast<const u32x4*>(w + (size_t)n * K);
    const bool has1 = n + 1 < n_end;
    const u32x4* w1 = reinterpret_cast<const u32x4*>(w + (size_t)(has1 ? n + 1 : n) * K);
#pragma unroll 4
    for (int v = lane; v < kv; v += 64) {
      const u32x4 wa = __builtin_nontemporal_load(w0 + v);
      const u32x4 wb = __builtin_nontemporal_load(w1 + v);
#pragma unroll
      for (int m = 0; m < MB; m++) {
        const u32x4 x = xs[m * kv + v];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          acc[0][m] = dot2<DT>(wa[e], x[e], acc[0][m]);
          acc[1][m] = dot2<DT>(wb[e], x[e], acc[1][m]);
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
        const float b = bias ? DT::to_f32(bias[n + r]) : 0.f;
#pragma unroll
        for (int m = 0; m < MB; m++)
          if (m < mb)
            c[(size_t)(m0 + m) * N + n + r] =
                bias ? DT::from_f32(DT::to_f32(DT::from_f32(acc[r][m])) + b) : DT::from_f32(acc[r][m]);
      }
    }
  }
}

// 2..16 activation rows per pass: the weight rows are the MFMA A operand straight from global
// memory (lane (kq, ml) loads the 16 bytes k0 + 8 kq .. of weight row n0 + ml; a lane's
// consecutive k-steps are consecutive 64-byte pieces of its row), the activation rows are the B
// operand from LDS, so the cost per weight byte no longer grows with the row count.
template <typename DT>
__global__ __launch_bounds__(512) void linear_mfma_kernel(const uint16_t* __restrict__ a, int lda,
                                                          const uint16_t* __restrict__ w,
                                                          const uint16_t* __restrict__ bias,
                                                          uint16_t* __restrict__ c, int M, int N, int K,
                                                          int rows_pp, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem);  // [rows_pp][K + 8]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = lane >> 4, ml = lane & 15;
  const int m0 = blockIdx.y * rows_pp;
  const int mb = min(rows_pp, M - m0);
  const int kv = K >> 3, astride = K + 8;
  const int n_begin = blockIdx.x * rows_per_block;
  const int n_end = min(N, n_begin + rows_per_block);
  // the wave's first weights are requested before the activations are staged: they depend on nothing
  u32x4 wf[8], wnx[8];
  const int n_first = n_begin + wave * 16;
  if (K >= 256) {
    const uint16_t* wrow = w + (size_t)min(n_first + ml, N - 1) * K + 8 * kq;
#pragma unroll
    for (int j = 0; j < 8; j++) wf[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + 32 * j));
  }
  // four requests in flight per thread: one at a time would pay an L2 round trip per iteration
  for (int base = threadIdx.x; base < mb * kv; base += 4 * 512) {
    u32x4 t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = base + j * 512;
      if (idx < mb * kv) {
        const int m = idx / kv, v = idx - m * kv;
        t[j] = *reinterpret_cast<const u32x4*>(a + (size_t)(m0 + m) * lda + 8 * v);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = base + j * 512;
      if (idx < mb * kv) {
        const int m = idx / kv, v = idx - m * kv;
        *reinterpret_cast<u32x4*>(xs + (size_t)m * astride + 8 * v) = t[j];
      }
    }
  }
  __syncthreads();
  const uint16_t* arow = xs + (size_t)min(ml, mb - 1) * astride + 8 * kq;
  for (int n0 = n_first; n0 < n_end; n0 += 8 * 16) {
    const uint16_t* wrow = w + (size_t)min(n0 + ml, N - 1) * K + 8 * kq;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    int k0 = 0;
    // the next 256 k of weights are requested before the current ones are multiplied (two register
    // sets); the scheduling barrier keeps the requests ahead of the arithmetic
    if (K >= 256 && n0 != n_first) {
#pragma unroll
      for (int j = 0; j < 8; j++) wf[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + 32 * j));
    }
    for (; k0 + 512 <= K; k0 += 256) {   // every block but the last: request the next one first
#pragma unroll
      for (int j = 0; j < 8; j++) wnx[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + k0 + 256 + 32 * j));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        acc0 = mfma16<DT>(wf[j], *reinterpret_cast<const u32x4*>(arow + k0 + 32 * j), acc0);
        acc1 = mfma16<DT>(wf[j + 1], *reinterpret_cast<const u32x4*>(arow + k0 + 32 * j + 32), acc1);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) wf[j] = wnx[j];
    }
    if (k0 + 256 <= K) {                 // the last whole block
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        acc0 = mfma16<DT>(wf[j], *reinterpret_cast<const u32x4*>(arow + k0 + 32 * j), acc0);
        acc1 = mfma16<DT>(wf[j + 1], *reinterpret_cast<const u32x4*>(arow + k0 + 32 * j + 32), acc1);
      }
      k0 += 256;
    }
    for (; k0 < K; k0 += 32)
      acc0 = mfma16<DT>(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + k0)),
                        *reinterpret_cast<const u32x4*>(arow + k0), acc0);
    // lane (kq, ml) holds outputs of activation row ml, weight rows n0 + 4 kq + r
    if (ml < mb) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = n0 + 4 * kq + r;
        if (n < n_end) {
          const float v = acc0[r] + acc1[r];
          c[(size_t)(m0 + ml) * N + n] =
              bias ? DT::from_f32(DT::to_f32(DT::from_f32(v)) + DT::to_f32(bias[n])) : DT::from_f32(v);
        }
      }
    }
  }
}

// Rows whose whole K does not fit in LDS beside each other (9..16 rows of a long K, 17..32 rows): as above with
// RB 16-row blocks per weight fragment and K walked in phases of `kp` channels.  The rows' channels of the phase
// are staged, every wave runs its 16 weight rows over them (its accumulators live across the phases), and the
// weights of the next phase's first block are already requested when the workgroup meets to restage.  The
// phase length is chosen so that two workgroups share a CU: one covers the other's restaging.
// One 16-row weight tile per wave (128 weight rows per workgroup); K and kp multiples of 256.
template <typename DT, int RB>
__global__ __launch_bounds__(512) void linear_mfma_phases_kernel(const uint16_t* __restrict__ a, int lda,
                                                                 const uint16_t* __restrict__ w,
                                                                 const uint16_t* __restrict__ bias,
                                                                 uint16_t* __restrict__ c, int M, int N, int K, int kp_len) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem);  // [16 RB][kp_len + 8]
  const int astride = kp_len + 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = lane >> 4, ml = lane & 15;
  const int m0 = blockIdx.y * (16 * RB);
  const int mb = min(16 * RB, M - m0);
  const int n0 = blockIdx.x * 128 + wave * 16;
  const uint16_t* wrow = w + (size_t)min(n0 + ml, N - 1) * K + 8 * kq;
  const uint16_t* arow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; rb++) arow[rb] = xs + (size_t)min(16 * rb + ml, mb - 1) * astride + 8 * kq;
  f32x4 acc[RB][2];
#pragma unroll
  for (int i = 0; i < RB; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 wf[8], wnx[8];
#pragma unroll
  for (int j = 0; j < 8; j++) wf[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + 32 * j));
  for (int kp = 0; kp < K; kp += kp_len) {
    const int klen = min(kp_len, K - kp), kv = klen >> 3;
    if (kp) __syncthreads();   // everyone has finished with the previous phase's rows
    for (int base = threadIdx.x; base < mb * kv; base += 4 * 512) {
      u32x4 t[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int idx = min(base + j * 512, mb * kv - 1);
        const int m = idx / kv, v = idx - m * kv;
        t[j] = *reinterpret_cast<const u32x4*>(a + (size_t)(m0 + m) * lda + kp + 8 * v);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int idx = base + j * 512;
        if (idx < mb * kv) {
          const int m = idx / kv, v = idx - m * kv;
          *reinterpret_cast<u32x4*>(xs + (size_t)m * astride + 8 * v) = t[j];
        }
      }
    }
    __syncthreads();
    for (int kk = 0; kk < klen; kk += 256) {
      // the block after this one (the last block asks for itself again: the request stays unconditional)
      const int knext = min(kp + kk + 256, K - 256);
#pragma unroll
      for (int j = 0; j < 8; j++) wnx[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + knext + 32 * j));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; j++)
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
          acc[rb][j & 1] = mfma16<DT>(wf[j], *reinterpret_cast<const u32x4*>(arow[rb] + kk + 32 * j), acc[rb][j & 1]);
#pragma unroll
      for (int j = 0; j < 8; j++) wf[j] = wnx[j];
    }
  }
  // lane (kq, ml) holds outputs of activation rows 16 rb + ml, weight rows n0 + 4 kq + r
#pragma unroll
  for (int rb = 0; rb < RB; rb++) {
    const int m = 16 * rb + ml;
    if (m < mb) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = n0 + 4 * kq + r;
        if (n < N) {
          const float v = acc[rb][0][r] + acc[rb][1][r];
          c[(size_t)(m0 + m) * N + n] =
              bias ? DT::from_f32(DT::to_f32(DT::from_f32(v)) + DT::to_f32(bias[n])) : DT::from_f32(v);
        }
      }
    }
  }
}

template <typename DT>
static int launch_linear(const void* a, int lda, const void* w, const void* bias, void* c, int M,
                         int N, int K, hipStream_t stream) {
  // rows that do not fit in half a CU's LDS with their whole K: phases of K, two workgroups per CU.  (The lm_head
  // of Llama-3-8B, 128256 x 4096: 8 rows unphased 187 us; 16 rows 210 us, against 385 us as two passes of 8 rows
  // over the weight; 32 rows 231 us.)
  if (M >= 3 && K % 256 == 0 && (M > 16 || (size_t)M * (K + 8) * 2 > 80 * 1024) && !env_flag("MI355Q_LINEAR_NO_MFMA")) {
    const int rb = M > 16 ? 2 : 1;
    const int kp = std::min(K, env_int("MI355Q_LINEAR_KP", rb == 2 ? 1024 : 2048));
    MI355Q_REQUIRE(kp >= 256 && kp % 256 == 0 && (size_t)16 * rb * (kp + 8) * 2 <= 150 * 1024, "linear_skinny: bad phase length %d", kp);
    const size_t lds = (size_t)16 * rb * (kp + 8) * 2;
    dim3 grid((N + 127) / 128, (M + 16 * rb - 1) / (16 * rb)), block(512);
    if (rb == 2)
      hipLaunchKernelGGL((linear_mfma_phases_kernel<DT, 2>), grid, block, lds, stream, (const uint16_t*)a, lda,
                         (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)c, M, N, K, kp);
    else
      hipLaunchKernelGGL((linear_mfma_phases_kernel<DT, 1>), grid, block, lds, stream, (const uint16_t*)a, lda,
                         (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)c, M, N, K, kp);
    return check_launch("linear_mfma_phases_kernel");
  }
  // two rows still go on the dot-product kernel (lm_head of Llama-3-8B: 172 us against 187 on the matrix cores; from
  // four rows on its vector work shows: 186 / 187, 346 at eight)
  if (M >= 3 && K % 32 == 0 && !env_flag("MI355Q_LINEAR_NO_MFMA")) {
    int rows_pp = std::min(M, 16);
    while (rows_pp > 1 && (size_t)rows_pp * (K + 8) * 2 > 144 * 1024) rows_pp--;
    const size_t lds_m = (size_t)rows_pp * (K + 8) * 2;
    if (lds_m <= 144 * 1024) {
      const int rpb = 128;
      dim3 grid((N + rpb - 1) / rpb, (M + rows_pp - 1) / rows_pp), block(512);
      hipLaunchKernelGGL((linear_mfma_kernel<DT>), grid, block, lds_m, stream, (const uint16_t*)a, lda,
                         (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)c, M, N, K, rows_pp, rpb);
      return check_launch("linear_mfma_kernel");
    }
  }
  // rows of A per pass: bounded by 64 KiB of LDS for the staged activations
  int MB = M >= 8 ? 8 : (M >= 4 ? 4 : (M >= 2 ? 2 : 1));
  while (MB > 1 && (size_t)MB * K * 2 > 64 * 1024) MB >>= 1;
  const size_t lds = (size_t)MB * K * 2;
  MI355Q_REQUIRE(lds <= 64 * 1024, "linear_skinny: K=%d too large for one LDS pass", K);
  // enough blocks to fill the chip a few times over, each amortising its activation staging
  int rows_per_block = 64;
  while (rows_per_block > 8 && (N + rows_per_block - 1) / rows_per_block < 2048) rows_per_block >>= 1;
  dim3 grid((N + rows_per_block - 1) / rows_per_block, (M + MB - 1) / MB), block(256);
#define MI355Q_LL(mb)                                                                            \
  hipLaunchKernelGGL((linear_skinny_kernel<DT, mb>), grid, block, lds, stream, (const uint16_t*)a, \
                     lda, (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)c, M, N, K,      \
                     rows_per_block)
  switch (MB) {
    case 8: MI355Q_LL(8); break;
    case 4: MI355Q_LL(4); break;
    case 2: MI355Q_LL(2); break;
    default: MI355Q_LL(1); break;
  }
#undef MI355Q_LL
  return check_launch("linear_skinny_kernel");
}

}  // namespace mi355q

using namespace mi355q;

extern "C" int mi355q_linear_skinny(const void* a, int lda, const void* w, const void* bias,
                                    void* c, int M, int N, int K, int act_dtype,
                                    mi355q_stream_t stream) {
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "linear_skinny: dtype must be bf16 or fp16");
  if (M == 0) return MI355Q_OK;
  MI355Q_REQUIRE(a && w && c && N > 0 && K > 0 && M >= 0, "linear_skinny: bad arguments");
  MI355Q_REQUIRE(K % 8 == 0 && lda % 8 == 0 && lda >= K, "linear_skinny: K=%d and lda=%d must be multiples of 8", K, lda);
  MI355Q_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0, "linear_skinny: pointers must be 16-byte aligned");
  if (act_dtype == MI355Q_BF16) return launch_linear<BF16>(a, lda, w, bias, c, M, N, K, (hipStream_t)stream);
  return launch_linear<FP16>(a, lda, w, bias, c, M, N, K, (hipStream_t)stream);
}
