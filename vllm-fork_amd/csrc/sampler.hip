// Random sampling with top-k / top-p for gfx950: one workgroup per row, no sort.
//
// What it replaces (vllm/v1/sample/ops/topk_topp_sampler.py): apply_top_k_top_p_pytorch (:367-407) sorts
// the whole vocabulary, masks everything below the k-th largest logit, masks the longest ascending
// prefix whose cumulative probability stays <= 1 - p (never the largest token), scatters back;
// random_sample (:441-466) then takes argmax(softmax(masked) / q) with q ~ Exp(1).
//
// Here both masks are thresholds on the logit value, found by radix selection over an order-preserving
// 32-bit key (11 + 11 + 10 bits, histograms in LDS):
//   top-k : descend from the top digit on COUNTS until the k-th largest key is fixed;
//   top-p : descend from the bottom digit on PROBABILITY MASS until the key is fixed at which the
//           ascending cumulative mass first exceeds (1 - p) Z.  Masses are 2^40 fixed point, summed with
//           integer atomics, so the result does not depend on the order in which lanes arrive.
// A last pass takes argmax over the kept keys of exp(l - max) / Z' / q (first index wins ties).
// The row is read once per pass (1 + 3 + 3 + 1 passes at most, from L2 after the first).
// Tokens with EQUAL logits at a threshold are kept or dropped together; the reference orders them by
// its sort, which is unspecified for ties.
#include "common.h"

namespace mi355q {

namespace {

constexpr int SAMPLE_THREADS = 1024;
constexpr int SAMPLE_BINS = 2048;

__device__ inline uint32_t order_key(float f) {   // a < b  <=>  key(a) < key(b); -0 < +0
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SampleParams {
  const float* logits;      // [B, stride]
  int64_t stride;
  const float* temperature; // [B] or null (1)
  const int32_t* top_k;     // [B] or null
  const float* top_p;       // [B] or null
  const float* noise;       // [B, V]: Exp(1) draws
  int64_t* ids;             // [B]
  int V;
};

// digit of `key` for the round that owns bits [shift, shift + bits)
__device__ inline uint32_t digit(uint32_t key, int shift, int bits) { return (key >> shift) & ((1u << bits) - 1u); }

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_topk_topp_kernel(const SampleParams p) {
  __shared__ uint32_t h_cnt[SAMPLE_BINS];
  __shared__ unsigned long long h_mass[SAMPLE_BINS];
  __shared__ float red_f[SAMPLE_THREADS / 64];
  __shared__ unsigned long long red_k[SAMPLE_THREADS / 64];
  __shared__ uint32_t s_sel;                 // digit chosen in a round
  __shared__ unsigned long long s_carry;     // count / mass carried into the next round
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* lg = p.logits + (size_t)row * p.stride;
  const float* q = p.noise + (size_t)row * p.V;
  const float temp = p.temperature ? p.temperature[row] : 1.0f;
  const int V = p.V;
  auto scaled = [&](int i) { return lg[i] / temp; };   // logits.div_(temperature), sampler.py:263-272

  // ---- max ---------------------------------------------------------------------------------
  float m = -INFINITY;
  for (int i = tid; i < V; i += SAMPLE_THREADS) m = fmaxf(m, scaled(i));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red_f[wave] = m;
  __syncthreads();
  m = red_f[0];
  for (int w = 1; w < SAMPLE_THREADS / 64; w++) m = fmaxf(m, red_f[w]);
  __syncthreads();

  const int rounds_shift[3] = {21, 10, 0}, rounds_bits[3] = {11, 11, 10};

  // ---- top-k: key of the k-th largest logit ---------------------------------------------------
  uint32_t key_k = 0u;   // keep keys >= key_k
  const int k = p.top_k ? p.top_k[row] : 0;
  if (k > 0 && k < V) {
    uint32_t prefix = 0u, prefix_mask = 0u;
    unsigned long long want = (unsigned long long)k;   // rank among the keys that share the prefix, from the top
    for (int r = 0; r < 3; r++) {
      const int shift = rounds_shift[r], bits = rounds_bits[r], nb = 1 << bits;
      for (int b = tid; b < nb; b += SAMPLE_THREADS) h_cnt[b] = 0u;
      __syncthreads();
      for (int i = tid; i < V; i += SAMPLE_THREADS) {
        const uint32_t key = order_key(scaled(i));
        if ((key & prefix_mask) == prefix) atomicAdd(&h_cnt[digit(key, shift, bits)], 1u);
      }
      __syncthreads();
      if (wave == 0) {   // suffix sums over the bins, 64 lanes x (nb / 64) bins, top bin first
        const int per = nb / 64;
        const int hi = nb - 1 - lane * per;           // this lane's bins: hi, hi - 1, ..., hi - per + 1
        unsigned long long mine = 0;
        for (int j = 0; j < per; j++) mine += h_cnt[hi - j];
        unsigned long long above = mine;               // inclusive scan over lanes 0..lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned long long t = __shfl_up(above, o, 64);
          if (lane >= o) above += t;
        }
        above -= mine;                                 // keys in bins above this lane's
        if (above < want && want <= above + mine) {
          unsigned long long acc = above;
          for (int j = 0; j < per; j++) {
            const unsigned long long c = h_cnt[hi - j];
            if (acc + c >= want) { s_sel = (uint32_t)(hi - j); s_carry = want - acc; break; }
            acc += c;
          }
        }
      }
      __syncthreads();
      prefix |= s_sel << shift;
      prefix_mask |= ((1u << bits) - 1u) << shift;
      want = s_carry;
      __syncthreads();
    }
    key_k = prefix;
  }

  // ---- top-p: smallest key whose ascending cumulative mass exceeds (1 - p) Z ---------------------
  uint32_t key_p = 0u;
  const float scale40 = 1099511627776.0f;   // 2^40
  auto mass_of = [&](float l) { return (unsigned long long)(__expf(l - m) * scale40); };
  unsigned long long z_kept = 0;            // mass of the finally kept keys, fixed point
  const float top_p = p.top_p ? p.top_p[row] : 1.0f;
  const bool use_p = p.top_p != nullptr && top_p < 1.0f;
  {
    uint32_t prefix = 0u, prefix_mask = 0u;
    unsigned long long below = 0;           // mass of kept-by-k keys smaller than the prefix range
    unsigned long long target = 0, ztot = 0;
    for (int r = 0; r < (use_p ? 3 : 1); r++) {
      const int shift = rounds_shift[r], bits = rounds_bits[r], nb = 1 << bits;
      for (int b = tid; b < nb; b += SAMPLE_THREADS) h_mass[b] = 0ull;
      __syncthreads();
      for (int i = tid; i < V; i += SAMPLE_THREADS) {
        const float l = scaled(i);
        const uint32_t key = order_key(l);
        if (key >= key_k && (key & prefix_mask) == prefix) atomicAdd(&h_mass[digit(key, shift, bits)], mass_of(l));
      }
      __syncthreads();
      if (wave == 0) {
        const int per = nb / 64;
        const int lo = lane * per;                    // this lane's bins, ascending
        unsigned long long mine = 0;
        for (int j = 0; j < per; j++) mine += h_mass[lo + j];
        unsigned long long upto = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned long long t = __shfl_up(upto, o, 64);
          if (lane >= o) upto += t;
        }
        if (r == 0) {
          const unsigned long long tot = __shfl(upto, 63, 64);
          if (lane == 0) {
            red_k[0] = tot;
            // drop the ascending prefix whose mass stays <= (1 - p) Z (topk_topp_sampler.py:396-401)
            red_k[1] = use_p ? (unsigned long long)((double)tot * (double)(1.0f - top_p)) : 0ull;
          }
        }
        const unsigned long long tgt = r == 0 ? (use_p ? (unsigned long long)((double)__shfl(upto, 63, 64) * (double)(1.0f - top_p)) : 0ull)
                                              : target;
        const unsigned long long before = below + upto - mine;   // mass below this lane's bins
        // first bin (ascending) at which the cumulative mass exceeds the target
        const bool here = before <= tgt && before + mine > tgt;
        const unsigned long long found = __ballot(here);
        if (found == 0ull) {
          // nothing exceeds it (p <= 0, or rounding at the very top): the largest non-empty bin stays
          const unsigned long long nonempty = __ballot(mine > 0);
          const int top_lane = 63 - __builtin_clzll(nonempty | 1ull);
          if (lane == top_lane) {
            unsigned long long acc = before;
            int sel = lo;
            for (int j = 0; j < per; j++) {
              if (h_mass[lo + j] > 0) { sel = lo + j; s_carry = acc; }
              acc += h_mass[lo + j];
            }
            s_sel = (uint32_t)sel;
          }
        } else if (here && (found & ((1ull << lane) - 1ull)) == 0ull) {
          unsigned long long acc = before;
          for (int j = 0; j < per; j++) {
            const unsigned long long c = h_mass[lo + j];
            if (acc + c > tgt) { s_sel = (uint32_t)(lo + j); s_carry = acc; break; }
            acc += c;
          }
        }
      }
      __syncthreads();
      if (r == 0) { ztot = red_k[0]; target = red_k[1]; }
      if (use_p) {
        prefix |= s_sel << shift;
        prefix_mask |= ((1u << bits) - 1u) << shift;
        below = s_carry;
      }
      __syncthreads();
    }
    if (use_p) key_p = prefix;
    z_kept = ztot - (use_p ? below : 0ull);
  }
  const uint32_t key_min = key_k > key_p ? key_k : key_p;
  const float z = (float)((double)z_kept / (double)scale40);

  // ---- argmax of softmax(masked) / q, first index on ties -----------------------------------------
  float best = -1.0f;
  int best_i = 0x7fffffff;
  for (int i = tid; i < V; i += SAMPLE_THREADS) {
    const float l = scaled(i);
    if (order_key(l) >= key_min) {
      const float s = (__expf(l - m) / z) / q[i];
      if (s > best) { best = s; best_i = i; }   // ascending i per thread: the first maximum stays
    }
  }
  // (score, -index) packed so that one 64-bit max does both comparisons; scores are >= 0
  unsigned long long pk = best >= 0.0f ? ((unsigned long long)__float_as_uint(best) << 32) | (uint32_t)(0x7fffffff - best_i) : 0ull;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(pk, o, 64);
    pk = t > pk ? t : pk;
  }
  if (lane == 0) red_k[wave] = pk;
  __syncthreads();
  if (tid == 0) {
    unsigned long long bestk = red_k[0];
    for (int w = 1; w < SAMPLE_THREADS / 64; w++) bestk = red_k[w] > bestk ? red_k[w] : bestk;
    p.ids[row] = (int64_t)(0x7fffffff - (int)(uint32_t)(bestk & 0xffffffffull));
  }
}

}  // namespace

}  // namespace mi355q

extern "C" int mi355q_sample_topk_topp(const float* logits, int64_t row_stride, const float* temperature,
                                       const int32_t* top_k, const float* top_p, const float* noise, int64_t* ids,
                                       int num_rows, int vocab, mi355q_stream_t stream) {
  using namespace mi355q;
  if (num_rows == 0) return MI355Q_OK;
  MI355Q_REQUIRE(logits && noise && ids && num_rows > 0 && vocab > 0, "sample_topk_topp: bad arguments");
  MI355Q_REQUIRE(row_stride >= vocab, "sample_topk_topp: row stride %lld < vocabulary %d", (long long)row_stride, vocab);
  MI355Q_REQUIRE(vocab <= (1 << 24), "sample_topk_topp: vocabulary %d too large for the 2^40 fixed-point masses", vocab);
  SampleParams p;
  p.logits = logits; p.stride = row_stride; p.temperature = temperature; p.top_k = top_k; p.top_p = top_p;
  p.noise = noise; p.ids = ids; p.V = vocab;
  hipLaunchKernelGGL(sample_topk_topp_kernel, dim3(num_rows), dim3(SAMPLE_THREADS), 0, (hipStream_t)stream, p);
  return check_launch("sample_topk_topp_kernel");
}
