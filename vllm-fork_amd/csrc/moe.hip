// Mixture-of-experts routing and combine for decode.  The expert GEMMs themselves are the
// W4A16 skinny kernel with per-row expert selection (mi355q_moe_w4a16_gemm in w4a16.hip);
// here: router softmax + top-k, and the weighted sum of the selected experts' outputs.
#include <algorithm>

#include "common.h"
#include "row_ops.h"

namespace mi355q {

template <int DTID>
__device__ inline float load_as_f32(const void* p, size_t i) {
  if constexpr (DTID == MI355Q_FP32) return reinterpret_cast<const float*>(p)[i];
  else if constexpr (DTID == MI355Q_BF16) return BF16::to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
  else return FP16::to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
}

// One wave, one token.  v holds the wave's logits (lane + 64 i; -inf past E).  probs = softmax in fp32;
// the top-k are taken by k rounds of arg-max (ties -> lowest expert index, as torch.topk on a descending
// stable order); renormalize divides the selected probabilities by their sum.
template <int PER_LANE>
__device__ __forceinline__ void topk_softmax_wave(float (&v)[PER_LANE], int lane, int E, int topk, int renormalize,
                                                  float* __restrict__ weights_row, int32_t* __restrict__ ids_row) {
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER_LANE; i++) mx = fmaxf(mx, v[i]);
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
    weights_row[lane] = renormalize ? my_w / picked_sum : my_w;
    ids_row[lane] = my_id;
  }
}

template <int DTID, int PER_LANE>
__global__ __launch_bounds__(64) void topk_softmax_kernel(float* __restrict__ topk_weights,
                                                          int32_t* __restrict__ topk_ids,
                                                          const void* __restrict__ gating, int E,
                                                          int topk, int renormalize) {
  const int t = blockIdx.x, lane = threadIdx.x;
  float v[PER_LANE];
#pragma unroll
  for (int i = 0; i < PER_LANE; i++) {
    const int e = lane + 64 * i;
    v[i] = e < E ? load_as_f32<DTID>(gating, (size_t)t * E + e) : -INFINITY;
  }
  topk_softmax_wave<PER_LANE>(v, lane, E, topk, renormalize, topk_weights + (size_t)t * topk, topk_ids + (size_t)t * topk);
}

// The router of a decoder layer at decode sizes in one launch, one workgroup per token:
//   residual = round(x + residual); x = RMSNorm(residual) * norm_w      (as fused_add_rms_norm)
//   logits = round(x . router_w[e, :])                                   (as the 16-bit skinny linear)
//   softmax + top-k of the logits                                        (as topk_softmax)
// with every stage's arithmetic in the order of the stand-alone kernel, so the results are the same
// bits.  The router rows of a wave's first two experts are requested before the norm starts: the
// launch then costs one memory round trip, two reductions and the selection.  E <= 64,
// H <= 2048 VPT.
template <typename DT, int VPT>
__global__ __launch_bounds__(256) void moe_router_topk_kernel(uint16_t* __restrict__ x, int64_t x_stride,
                                                              uint16_t* __restrict__ residual,
                                                              const uint16_t* __restrict__ norm_w, float eps,
                                                              const uint16_t* __restrict__ router_w,
                                                              uint16_t* __restrict__ logits_out,
                                                              float* __restrict__ topk_weights,
                                                              int32_t* __restrict__ topk_ids, int E, int H,
                                                              int topk, int renormalize) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);                         // [H / 8] the normalised row
  uint16_t* lg = reinterpret_cast<uint16_t*>(smem + (size_t)H * 2);   // [64] logits
  float* scratch = reinterpret_cast<float*>(lg + 64);                 // [4]
  const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kv = H >> 3;
  constexpr int PV = 4 * VPT;   // vectors of a router row per lane
  u32x4 wa[PV], wb[PV];
  auto request = [&](int n) {
    const u32x4* w0 = reinterpret_cast<const u32x4*>(router_w + (size_t)min(n, E - 1) * H);
    const u32x4* w1 = reinterpret_cast<const u32x4*>(router_w + (size_t)min(n + 1, E - 1) * H);
#pragma unroll
    for (int i = 0; i < PV; i++) {
      const int v = min(lane + 64 * i, kv - 1);
      wa[i] = w0[v];
      wb[i] = w1[v];
    }
  };
  request(2 * wave);
  add_rms_norm_row_vec<DT, 256, VPT>(x + (size_t)t * x_stride, residual + (size_t)t * H, norm_w, eps, H, scratch, xs);
  __syncthreads();
  for (int n = 2 * wave; n < E; n += 8) {
    if (n != 2 * wave) request(n);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < PV; i++) {
      const int v = lane + 64 * i;
      if (v < kv) {
        const u32x4 xv = xs[v];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          a0 = dot2<DT>(wa[i][e], xv[e], a0);
          a1 = dot2<DT>(wb[i][e], xv[e], a1);
        }
      }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane == 0) {
      lg[n] = DT::from_f32(a0);
      if (n + 1 < E) lg[n + 1] = DT::from_f32(a1);
    }
  }
  __syncthreads();
  if (logits_out && threadIdx.x < E) logits_out[(size_t)t * E + threadIdx.x] = lg[threadIdx.x];
  if (wave == 0) {
    float v[1] = {lane < E ? DT::to_f32(lg[lane]) : -INFINITY};
    topk_softmax_wave<1>(v, lane, E, topk, renormalize, topk_weights + (size_t)t * topk, topk_ids + (size_t)t * topk);
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
      // (one fused multiply-add per term, spelled out: the projection that takes this sum in its prologue repeats it)
      acc[2 * e] = __builtin_fmaf(wj, DT::to_f32((uint16_t)(x[e] & 0xffff)), acc[2 * e]);
      acc[2 * e + 1] = __builtin_fmaf(wj, DT::to_f32((uint16_t)(x[e] >> 16)), acc[2 * e + 1]);
    }
  }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    // (the sums exist as fp32 values before they are rounded to 16 bits: no fused multiply-add-convert)
    asm volatile("" : "+v"(acc[2 * e]), "+v"(acc[2 * e + 1]));
    o[e] = pack2<DT>(acc[2 * e], acc[2 * e + 1]);
  }
  *reinterpret_cast<u32x4*>(out + (size_t)t * H + 8 * v) = o;
}

// Token sort for the grouped expert GEMM (moe_align_block_size of the reference): the (token, expert)
// slots are grouped by expert, every expert's run is padded to whole row tiles of `block` slots, and
// each tile is labelled with its expert.  One workgroup; counts, offsets and cursors in LDS.  The
// order of an expert's slots inside its run depends on arrival order, which no result depends on: a
// row of the product is a function of its own input row and its expert's weights only.
__global__ __launch_bounds__(1024) void moe_align_kernel(const int* __restrict__ ids, int slots, int num_experts,
                                                         int block, int* __restrict__ sorted_ids,
                                                         int* __restrict__ tile_expert, int* __restrict__ num_tiles,
                                                         int* __restrict__ slot_pos, int max_tiles) {
  extern __shared__ int sm[];          // counts[E], offsets[E + 1], cursor[E]
  int* counts = sm;
  int* offsets = sm + num_experts;
  int* cursor = offsets + num_experts + 1;
  for (int e = threadIdx.x; e < num_experts; e += blockDim.x) { counts[e] = 0; cursor[e] = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < slots; i += blockDim.x) {
    const int e = ids[i];
    if (e >= 0 && e < num_experts) atomicAdd(&counts[e], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int off = 0;
    for (int e = 0; e < num_experts; e++) {
      offsets[e] = off;
      const int tiles = (counts[e] + block - 1) / block;
      for (int t = 0; t < tiles; t++) tile_expert[off / block + t] = e;
      off += tiles * block;
    }
    offsets[num_experts] = off;
    *num_tiles = off / block;
    for (int t = off / block; t < max_tiles; t++) tile_expert[t] = -1;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < slots; i += blockDim.x) {
    const int e = ids[i];
    if (e >= 0 && e < num_experts) {
      const int pos = offsets[e] + atomicAdd(&cursor[e], 1);
      sorted_ids[pos] = i;
      slot_pos[i] = pos;
    } else {
      slot_pos[i] = -1;
    }
  }
  // padding: the tail of every expert's run and everything past the last run name no slot
  for (int e = 0; e < num_experts; e++) {
    const int from = offsets[e] + counts[e], to = offsets[e + 1];
    for (int i = from + threadIdx.x; i < to; i += blockDim.x) sorted_ids[i] = slots;
  }
  for (int i = offsets[num_experts] + threadIdx.x; i < max_tiles * block; i += blockDim.x) sorted_ids[i] = slots;
}

}  // namespace mi355q

using namespace mi355q;

extern "C" int mi355q_moe_align_block_size(const int32_t* expert_ids, int slots, int num_experts, int block,
                                           int32_t* sorted_ids, int32_t* tile_expert, int32_t* num_tiles,
                                           int32_t* slot_pos, int max_tiles, mi355q_stream_t stream) {
  MI355Q_REQUIRE(expert_ids && sorted_ids && tile_expert && num_tiles && slot_pos, "moe_align_block_size: null pointer");
  MI355Q_REQUIRE(slots >= 0 && num_experts >= 1 && num_experts <= 4096 && block >= 1, "moe_align_block_size: bad arguments");
  const long need = ((long)slots + (long)std::min(slots, num_experts) * (block - 1) + block - 1) / block;
  MI355Q_REQUIRE(max_tiles >= need, "moe_align_block_size: room for %d tiles, %ld may be needed", max_tiles, need);
  hipLaunchKernelGGL(moe_align_kernel, dim3(1), dim3(1024), (3 * (size_t)num_experts + 1) * sizeof(int), (hipStream_t)stream,
                     expert_ids, slots, num_experts, block, sorted_ids, tile_expert, num_tiles, slot_pos, max_tiles);
  return check_launch("moe_align_kernel");
}

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

extern "C" int mi355q_moe_router_topk_supported(int num_experts, int hidden) {
  return num_experts >= 1 && num_experts <= 64 && hidden >= 8 && hidden % 8 == 0 && hidden <= 8192;
}

extern "C" int mi355q_moe_router_topk(void* input, int64_t input_stride, void* residual, const void* norm_weight,
                                      float epsilon, const void* router_weight, void* logits, float* topk_weights,
                                      int32_t* topk_ids, int num_tokens, int num_experts, int hidden, int topk,
                                      int renormalize, int dtype, mi355q_stream_t stream) {
  if (num_tokens == 0) return MI355Q_OK;
  MI355Q_REQUIRE(input && residual && router_weight && topk_weights && topk_ids, "moe_router_topk: null pointer");
  MI355Q_REQUIRE(dtype == MI355Q_BF16 || dtype == MI355Q_FP16, "moe_router_topk: dtype must be bf16 or fp16");
  MI355Q_REQUIRE(mi355q_moe_router_topk_supported(num_experts, hidden),
                 "moe_router_topk: %d experts over %d channels (up to 64 experts, hidden a multiple of 8 up to 8192)",
                 num_experts, hidden);
  MI355Q_REQUIRE(topk >= 1 && topk <= num_experts, "moe_router_topk: topk=%d out of range", topk);
  MI355Q_REQUIRE(input_stride >= hidden && input_stride % 8 == 0, "moe_router_topk: bad input stride");
  MI355Q_REQUIRE((((uintptr_t)input | (uintptr_t)residual | (uintptr_t)norm_weight | (uintptr_t)router_weight) & 15) == 0,
                 "moe_router_topk: pointers must be 16-byte aligned");
  const size_t lds = (size_t)hidden * 2 + 64 * 2 + 4 * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define MI355Q_RT(DT, VPT)                                                                                          \
  hipLaunchKernelGGL((moe_router_topk_kernel<DT, VPT>), dim3(num_tokens), dim3(256), lds, s, (uint16_t*)input,     \
                     input_stride, (uint16_t*)residual, (const uint16_t*)norm_weight, epsilon,                     \
                     (const uint16_t*)router_weight, (uint16_t*)logits, topk_weights, topk_ids, num_experts, hidden, \
                     topk, renormalize)
  if (dtype == MI355Q_BF16) {
    if (hidden <= 4096) MI355Q_RT(BF16, 2); else MI355Q_RT(BF16, 4);
  } else {
    if (hidden <= 4096) MI355Q_RT(FP16, 2); else MI355Q_RT(FP16, 4);
  }
#undef MI355Q_RT
  return check_launch("moe_router_topk_kernel");
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
