// Mixture-of-experts routing and combine for decode.  The expert GEMMs themselves are the
// W4A16 skinny kernel with per-row expert selection (mi355q_moe_w4a16_gemm in w4a16.hip);
// here: router softmax + top-k, and the weighted sum of the selected experts' outputs.
#include "common.h"

namespace mi355q {

template <int DTID>
__device__ inline float load_as_f32(const void* p, size_t i) {
  if constexpr (DTID == MI355Q_FP32) return reinterpret_cast<const float*>(p)[i];
  else if constexpr (DTID == MI355Q_BF16) return BF16::to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
  else return FP16::to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
}

// One wave per token.  probs = softmax(gating[t, :]) in fp32; the top-k are taken by k rounds
// of arg-max (ties -> lowest expert index, as torch.topk on a descending stable order);
// renormalize divides the selected probabilities by their sum.
template <int DTID, int PER_LANE>
__global__ __launch_bounds__(64) void topk_softmax_kernel(float* __restrict__ topk_weights,
                                                          int32_t* __restrict__ topk_ids,
                                                          const void* __restrict__ gating, int E,
                                                          int topk, int renormalize) {
  const int t = blockIdx.x, lane = threadIdx.x;
  float v[PER_LANE];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER_LANE; i++) {
    const int e = lane + 64 * i;
    v[i] = e < E ? load_as_f32<DTID>(gating, (size_t)t * E + e) : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; i++) {
    v[i] = (lane + 64 * i) < E ? __expf(v[i] - mx) : -1.f;  // -1 marks "not a candidate"
    if (v[i] > 0.f) sum += v[i];
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  float picked_sum = 0.f;
  float my_w = 0.f;
  int my_id = 0;
  for (int j = 0; j < topk; j++) {
    float best = -1.f;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PER_LANE; i++) {
      const int e = lane + 64 * i;
      if (v[i] > best) { best = v[i]; best_e = e; }  // ascending e within a lane: first maximum kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oe = __shfl_xor(best_e, o, 64);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
    }
#pragma unroll
    for (int i = 0; i < PER_LANE; i++)
      if (lane + 64 * i == best_e) v[i] = -1.f;
    const float w = best * inv;
    picked_sum += w;
    if (lane == j) { my_w = w; my_id = best_e; }
  }
  if (lane < topk) {
    topk_weights[(size_t)t * topk + lane] = renormalize ? my_w / picked_sum : my_w;
    topk_ids[(size_t)t * topk + lane] = my_id;
  }
}

// out[t, h] = round(sum_j w[t, j] * y[t * topk + j, h]) in fp32, j ascending (w == null: plain sum)
template <typename DT>
__global__ __launch_bounds__(256) void moe_sum_kernel(uint16_t* __restrict__ out,
                                                      const uint16_t* __restrict__ y,
                                                      const float* __restrict__ w, int topk, int H) {
  const int t = blockIdx.y;
  const int v = blockIdx.x * 256 + threadIdx.x;  // 8-element vectors
  if (v * 8 >= H) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < topk; j++) {
    const float wj = w ? w[(size_t)t * topk + j] : 1.f;
    const u32x4 x = *reinterpret_cast<const u32x4*>(y + ((size_t)t * topk + j) * H + 8 * v);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      acc[2 * e] += wj * DT::to_f32((uint16_t)(x[e] & 0xffff));
      acc[2 * e + 1] += wj * DT::to_f32((uint16_t)(x[e] >> 16));
    }
  }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; e++) o[e] = pack2<DT>(acc[2 * e], acc[2 * e + 1]);
  *reinterpret_cast<u32x4*>(out + (size_t)t * H + 8 * v) = o;
}

}  // namespace mi355q

using namespace mi355q;

extern "C" int mi355q_topk_softmax(float* topk_weights, int32_t* topk_ids, const void* gating_output,
                                   int num_tokens, int num_experts, int topk, int renormalize,
                                   int dtype, mi355q_stream_t stream) {
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(topk_weights && topk_ids && gating_output, "topk_softmax: null pointer");
  MI355Q_REQUIRE(dtype >= MI355Q_BF16 && dtype <= MI355Q_FP32, "topk_softmax: bad dtype");
  MI355Q_REQUIRE(num_experts >= 1 && num_experts <= 512, "topk_softmax: %d experts (1..512 supported)", num_experts);
  MI355Q_REQUIRE(topk >= 1 && topk <= num_experts && topk <= 64, "topk_softmax: topk=%d out of range", topk);
  hipStream_t s = (hipStream_t)stream;
  const int per = (num_experts + 63) / 64;
#define MI355Q_TK(DTID, PL)                                                                             \
  hipLaunchKernelGGL((topk_softmax_kernel<DTID, PL>), dim3(num_tokens), dim3(64), 0, s, topk_weights, \
                     topk_ids, gating_output, num_experts, topk, renormalize)
#define MI355Q_TKD(PL)                                        \
  do {                                                        \
    if (dtype == MI355Q_FP32) MI355Q_TK(MI355Q_FP32, PL);     \
    else if (dtype == MI355Q_BF16) MI355Q_TK(MI355Q_BF16, PL);\
    else MI355Q_TK(MI355Q_FP16, PL);                          \
  } while (0)
  if (per <= 1) MI355Q_TKD(1);
  else if (per <= 2) MI355Q_TKD(2);
  else if (per <= 4) MI355Q_TKD(4);
  else MI355Q_TKD(8);
#undef MI355Q_TKD
#undef MI355Q_TK
  return check_launch("topk_softmax_kernel");
}

extern "C" int mi355q_moe_sum(void* out, const void* expert_out, const float* topk_weights,
                              int num_tokens, int topk, int hidden, int dtype,
                              mi355q_stream_t stream) {
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(out && expert_out && topk >= 1 && hidden > 0 && hidden % 8 == 0, "moe_sum: bad arguments");
  MI355Q_REQUIRE(dtype == MI355Q_BF16 || dtype == MI355Q_FP16, "moe_sum: dtype must be bf16 or fp16");
  MI355Q_REQUIRE(((uintptr_t)out & 15) == 0 && ((uintptr_t)expert_out & 15) == 0, "moe_sum: pointers must be 16-byte aligned");
  dim3 grid((hidden / 8 + 255) / 256, num_tokens), block(256);
  if (dtype == MI355Q_BF16)
    hipLaunchKernelGGL((moe_sum_kernel<BF16>), grid, block, 0, (hipStream_t)stream, (uint16_t*)out,
                       (const uint16_t*)expert_out, topk_weights, topk, hidden);
  else
    hipLaunchKernelGGL((moe_sum_kernel<FP16>), grid, block, 0, (hipStream_t)stream, (uint16_t*)out,
                       (const uint16_t*)expert_out, topk_weights, topk, hidden);
  return check_launch("moe_sum_kernel");
}
