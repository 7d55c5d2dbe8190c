// W4A16 dequant-GEMM for gfx950: tiled int4 layout, skinny (M <= 16 per pass)
// MFMA kernel with deterministic split-K, load-time repack, dequantize.
//
// Packed layout ("tile64x32"): the [K, N] int4 matrix is cut into tiles of 64 output
// channels x 32 input channels (1 KiB); tiles of one 64-channel strip are consecutive
// along K.  Inside a tile lane l of a wave owns bytes [16 l, 16 l + 16): four dwords,
// dword i holding the 8 nibbles of
//     n = 64 strip + 16 ((l & 15) >> 2) + 4 i + (l & 3),   k = 32 kstep + 8 (l >> 4) + j
// with nibble j = 2p at bits [4p, 4p+4) and j = 2p+1 at bits [16 + 4p, 16 + 4p + 4).
// One 16-byte load per lane therefore gives the A fragments of four
// v_mfma_f32_16x16x32 tiles, and ((w >> 4p) & 0x000F000F) | magic is directly the pair of
// 16-bit floats (bias + q) for fragment register p.
//
// Arithmetic: per scale group, acc = sum_k a_k (bias + q_k) on the matrix cores (exact
// products, fp32 accumulation), a1 = sum_k a_k from one extra MFMA with an all-ones A
// operand, and the group contributes s * (acc - (bias + z) * a1) to the fp32 total.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace mi355q {

// ---------------------------------------------------------------------------
// repack (shared host/device arithmetic)
// ---------------------------------------------------------------------------
__host__ __device__ static inline int awq_pos(int n) {
  // logical column j of 8 sits at nibble position [0,4,1,5,2,6,3,7][j]
  const int j = n & 7;
  return (j >> 1) + ((j & 1) << 2);
}

__host__ __device__ static inline uint32_t src_nibble(const int32_t* qw, int layout, int N, int k,
                                                      int n) {
  if (layout == MI355Q_W_GPTQ)
    return ((uint32_t)qw[(size_t)(k >> 3) * N + n] >> (4 * (k & 7))) & 0xfu;
  const int pos = layout == MI355Q_W_AWQ ? awq_pos(n) : (n & 7);
  return ((uint32_t)qw[(size_t)k * (N >> 3) + (n >> 3)] >> (4 * pos)) & 0xfu;
}

// index of the packed dword holding (k, n) and the bit offset of its nibble
__host__ __device__ static inline void packed_pos(int K, int k, int n, size_t* dword, int* shift) {
  const int strip = n >> 6, nl = n & 63;
  const int q = nl >> 4, i = (nl >> 2) & 3, r = nl & 3;  // row rho = 4 q + r of MFMA tile i
  const int t = k >> 5, kq = (k >> 3) & 3, j = k & 7;
  const int lane = (kq << 4) | (q << 2) | r;
  *dword = (((size_t)strip * (K >> 5) + t) * 64 + lane) * 4 + i;
  *shift = ((j & 1) << 4) + ((j >> 1) << 2);
}

__host__ __device__ static inline uint32_t repack_dword(const int32_t* qw, int layout, int K, int N,
                                                        size_t d) {
  (void)K;
  const int i = (int)(d & 3);
  const int lane = (int)((d >> 2) & 63);
  const size_t tile = d >> 8;
  const int ksteps = K >> 5;
  const int t = (int)(tile % ksteps), strip = (int)(tile / ksteps);
  const int rho = lane & 15, kq = lane >> 4;
  const int n = 64 * strip + 16 * (rho >> 2) + 4 * i + (rho & 3);
  const int k0 = 32 * t + 8 * kq;
  uint32_t w = 0;
  if (layout == MI355Q_W_GPTQ) {
    const uint32_t s = (uint32_t)qw[(size_t)(k0 >> 3) * N + n];
#pragma unroll
    for (int j = 0; j < 8; j++)
      w |= ((s >> (4 * j)) & 0xfu) << (((j & 1) << 4) + ((j >> 1) << 2));
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++)
      w |= src_nibble(qw, layout, N, k0 + j, n) << (((j & 1) << 4) + ((j >> 1) << 2));
  }
  return w;
}

__host__ __device__ static inline uint16_t repack_zc(const int32_t* qz, int zp_mode, int N, int g,
                                                     int n, int act_dtype) {
  const int pos = zp_mode == MI355Q_ZP_AWQ ? awq_pos(n) : (n & 7);
  uint32_t z = ((uint32_t)qz[(size_t)g * (N >> 3) + (n >> 3)] >> (4 * pos)) & 0xfu;
  if (zp_mode == MI355Q_ZP_GPTQ_V1) z = (z + 1) & 0x1fu;
  // bias + z is exactly representable: bf16(128 + z) = 0x4300 | z, fp16(1024 + z) = 0x6400 | z
  return act_dtype == MI355Q_BF16 ? (uint16_t)(0x4300u + z) : (uint16_t)(0x6400u + z);
}

__global__ void repack_weight_kernel(const int32_t* __restrict__ qw, uint32_t* __restrict__ out,
                                     int K, int N, int layout, size_t ndwords) {
  for (size_t d = blockIdx.x * (size_t)blockDim.x + threadIdx.x; d < ndwords;
       d += (size_t)gridDim.x * blockDim.x)
    out[d] = repack_dword(qw, layout, K, N, d);
}

__global__ void repack_zc_kernel(const int32_t* __restrict__ qz, uint16_t* __restrict__ zc, int N,
                                 int groups, int zp_mode, int act_dtype) {
  const size_t total = (size_t)groups * N;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x)
    zc[e] = repack_zc(qz, zp_mode, N, (int)(e / N), (int)(e % N), act_dtype);
}

static int check_shape(int K, int N, int& G) {
  if (G <= 0) G = K;
  MI355Q_REQUIRE(K > 0 && N > 0, "w4a16: K=%d N=%d must be positive", K, N);
  MI355Q_REQUIRE(K % 32 == 0, "w4a16: K=%d must be a multiple of 32", K);
  MI355Q_REQUIRE(N % 64 == 0, "w4a16: N=%d must be a multiple of 64", N);
  MI355Q_REQUIRE(G % 32 == 0 && K % G == 0, "w4a16: group size %d must divide K=%d and be a multiple of 32", G, K);
  return 0;
}

static int check_enums(int src_layout, int zp_mode, int act_dtype, const void* qzeros,
                       const void* zc) {
  MI355Q_REQUIRE(src_layout >= MI355Q_W_GPTQ && src_layout <= MI355Q_W_AWQ, "w4a16: bad src_layout %d", src_layout);
  MI355Q_REQUIRE(zp_mode >= MI355Q_ZP_CONST && zp_mode <= MI355Q_ZP_GPTQ_V1, "w4a16: bad zp_mode %d", zp_mode);
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w4a16: act_dtype must be bf16 or fp16");
  if (zp_mode != MI355Q_ZP_CONST)
    MI355Q_REQUIRE(qzeros != nullptr && zc != nullptr, "w4a16: zero-point mode %d needs qzeros and zc", zp_mode);
  return 0;
}

}  // namespace mi355q

using namespace mi355q;

extern "C" size_t mi355q_w4a16_packed_bytes(int K, int N) { return (size_t)K * N / 2; }

extern "C" int mi355q_w4a16_repack(const int32_t* qweight, const int32_t* qzeros, void* packed,
                                   void* zc, int K, int N, int G, int src_layout, int zp_mode,
                                   int act_dtype, mi355q_stream_t stream) {
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_enums(src_layout, zp_mode, act_dtype, qzeros, zc)) return e;
  MI355Q_REQUIRE(qweight && packed, "w4a16_repack: null pointer");
  const size_t nd = (size_t)K * N / 8;
  const int grid = (int)((nd + 255) / 256 < 8192 ? (nd + 255) / 256 : 8192);
  hipLaunchKernelGGL(repack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, qweight,
                     (uint32_t*)packed, K, N, src_layout, nd);
  if (int e = check_launch("repack_weight_kernel")) return e;
  if (zp_mode != MI355Q_ZP_CONST) {
    const size_t nz = (size_t)(K / G) * N;
    const int gz = (int)((nz + 255) / 256 < 4096 ? (nz + 255) / 256 : 4096);
    hipLaunchKernelGGL(repack_zc_kernel, dim3(gz), dim3(256), 0, (hipStream_t)stream, qzeros,
                       (uint16_t*)zc, N, K / G, zp_mode, act_dtype);
    if (int e = check_launch("repack_zc_kernel")) return e;
  }
  return MI355Q_OK;
}

extern "C" int mi355q_w4a16_repack_host(const int32_t* qweight, const int32_t* qzeros,
                                        void* packed, void* zc, int K, int N, int G,
                                        int src_layout, int zp_mode, int act_dtype) {
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_enums(src_layout, zp_mode, act_dtype, qzeros, zc)) return e;
  MI355Q_REQUIRE(qweight && packed, "w4a16_repack_host: null pointer");
  const size_t nd = (size_t)K * N / 8;
  uint32_t* out = (uint32_t*)packed;
  for (size_t d = 0; d < nd; d++) out[d] = repack_dword(qweight, src_layout, K, N, d);
  if (zp_mode != MI355Q_ZP_CONST) {
    uint16_t* z = (uint16_t*)zc;
    for (int g = 0; g < K / G; g++)
      for (int n = 0; n < N; n++) z[(size_t)g * N + n] = repack_zc(qzeros, zp_mode, N, g, n, act_dtype);
  }
  return MI355Q_OK;
}

extern "C" int mi355q_w4a16_unpack_host(const void* packed, int32_t* q, int K, int N) {
  int G = K;
  if (int e = check_shape(K, N, G)) return e;
  const uint32_t* p = (const uint32_t*)packed;
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) {
      size_t d;
      int sh;
      packed_pos(K, k, n, &d, &sh);
      q[(size_t)k * N + n] = (int32_t)((p[d] >> sh) & 0xfu);
    }
  return MI355Q_OK;
}

// ---------------------------------------------------------------------------
// skinny GEMM kernel
// ---------------------------------------------------------------------------
namespace mi355q {

struct SkinnyParams {
  const uint16_t* x;       // [M, K] (or [M, 2K] gate|up when SILU)
  int ldx;
  const u32x4* wp;         // packed tiles
  const uint16_t* scales;  // [K/G, N]
  const uint16_t* zc;      // [K/G, N] or null
  const uint16_t* bias;    // [N] or null
  uint16_t* out;           // [M, N] (written when splits == 1)
  float* partial;          // [splits, M, N] (written when splits > 1)
  int M, N, K, G;
  int chunks;              // K / (32 * CS)
  int splits;
};

template <typename DT>
__device__ inline u32x4 dequant_frag(uint32_t w) {
  u32x4 r;
  r[0] = (w & 0x000F000Fu) | DT::int4_magic;
  r[1] = ((w >> 4) & 0x000F000Fu) | DT::int4_magic;
  r[2] = ((w >> 8) & 0x000F000Fu) | DT::int4_magic;
  r[3] = ((w >> 12) & 0x000F000Fu) | DT::int4_magic;
  return r;
}

__device__ inline float silu_f(float g) { return g / (1.0f + __expf(-g)); }

// CS = k-steps (of 32) per flush chunk; a chunk never crosses a scale group.
template <typename DT, int CS, bool HAS_ZC, bool SILU>
__global__ __launch_bounds__(256) void w4a16_skinny_kernel(const SkinnyParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                     // [4][16][64] fp32 = 16 KiB
  uint16_t* act = reinterpret_cast<uint16_t*>(smem + 16384);       // SILU: [Mt][krange]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x, sp = blockIdx.y, mt = blockIdx.z;
  const int kq = lane >> 4, ml = lane & 15;
  const int ksteps = p.K >> 5;
  const int c0 = (int)(((long long)p.chunks * sp) / p.splits);
  const int c1 = (int)(((long long)p.chunks * (sp + 1)) / p.splits);
  const int Mt = min(16, p.M - mt * 16);
  const int mrow = mt * 16 + min(ml, Mt - 1);

  const u32x4* wstrip = p.wp + (size_t)strip * ksteps * 64 + lane;
  const uint16_t* xrow = p.x + (size_t)mrow * p.ldx + 8 * kq;
  const int kbase = c0 * CS * 32;  // first k of this block's range

  if constexpr (SILU) {
    // stage act[m][k - kbase] = silu(gate) * up for the block's k range, rows < Mt
    const int krange = (c1 - c0) * CS * 32;
    const int vecs = krange >> 3;
    for (int idx = threadIdx.x; idx < Mt * vecs; idx += 256) {
      const int m = idx / vecs, v = idx - m * vecs;
      const uint16_t* src = p.x + (size_t)(mt * 16 + m) * p.ldx + kbase + 8 * v;
      const u32x4 g = *reinterpret_cast<const u32x4*>(src);
      const u32x4 u = *reinterpret_cast<const u32x4*>(src + p.K);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float g0 = DT::to_f32((uint16_t)(g[e] & 0xffff)), g1 = DT::to_f32((uint16_t)(g[e] >> 16));
        const float u0 = DT::to_f32((uint16_t)(u[e] & 0xffff)), u1 = DT::to_f32((uint16_t)(u[e] >> 16));
        // silu rounded to the activation dtype before the multiply, as the reference op does
        const float s0 = DT::to_f32(DT::from_f32(silu_f(g0))), s1 = DT::to_f32(DT::from_f32(silu_f(g1)));
        o[e] = pack2<DT>(s0 * u0, s1 * u1);
      }
      *reinterpret_cast<u32x4*>(act + (size_t)m * krange + 8 * v) = o;
    }
    __syncthreads();
  }
  const int krange_l = SILU ? (c1 - c0) * CS * 32 : 0;
  const uint16_t* arow = act + (size_t)min(ml, Mt - 1) * krange_l + 8 * kq;

  f32x4 tot[4];
#pragma unroll
  for (int i = 0; i < 4; i++) tot[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const u32x4 ones = u32x4{DT::ones, DT::ones, DT::ones, DT::ones};
  const size_t sc_off = (size_t)strip * 64 + 16 * kq;

  auto issue = [&](int c, u32x4(&w)[CS], u32x4(&b)[CS]) {
#pragma unroll
    for (int u = 0; u < CS; u++) {
      const int t = c * CS + u;
      w[u] = __builtin_nontemporal_load(wstrip + (size_t)t * 64);
      if constexpr (SILU)
        b[u] = *reinterpret_cast<const u32x4*>(arow + (t * 32 - kbase));
      else
        b[u] = *reinterpret_cast<const u32x4*>(xrow + t * 32);
    }
  };

  auto consume = [&](int c, const u32x4(&w)[CS], const u32x4(&b)[CS]) {
    const int g = (c * CS * 32) / p.G;
    const u32x4* sp_ = reinterpret_cast<const u32x4*>(p.scales + (size_t)g * p.N + sc_off);
    const u32x4 s_lo = sp_[0], s_hi = sp_[1];
    u32x4 z_lo, z_hi;
    if constexpr (HAS_ZC) {
      const u32x4* zp_ = reinterpret_cast<const u32x4*>(p.zc + (size_t)g * p.N + sc_off);
      z_lo = zp_[0];
      z_hi = zp_[1];
    }
    f32x4 acc[4], acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CS; u++) {
      acc1 = mfma16<DT>(ones, b[u], acc1);
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = mfma16<DT>(dequant_frag<DT>(w[u][i]), b[u], acc[i]);
    }
    // flush: output channel e = 4 i + r of this lane's 16 (n = 64 strip + 16 kq + e)
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int e = 4 * i + r;
        const uint32_t sw = e < 8 ? s_lo[e >> 1] : s_hi[(e - 8) >> 1];
        const float s = DT::to_f32((uint16_t)((e & 1) ? (sw >> 16) : (sw & 0xffff)));
        float z;
        if constexpr (HAS_ZC) {
          const uint32_t zw = e < 8 ? z_lo[e >> 1] : z_hi[(e - 8) >> 1];
          z = DT::to_f32((uint16_t)((e & 1) ? (zw >> 16) : (zw & 0xffff)));
        } else {
          z = DT::int4_bias + 8.0f;
        }
        const float d = __builtin_fmaf(-z, acc1[r], acc[i][r]);
        tot[i][r] = __builtin_fmaf(s, d, tot[i][r]);
      }
    }
  };

  {
    u32x4 w0[CS], b0[CS], w1[CS], b1[CS];
    int c = c0 + wave;
    if (c < c1) {
      issue(c, w0, b0);
      while (true) {
        int cn = c + 4;
        bool more = cn < c1;
        if (more) issue(cn, w1, b1);
        consume(c, w0, b0);
        if (!more) break;
        c = cn;
        cn = c + 4;
        more = cn < c1;
        if (more) issue(cn, w0, b0);
        consume(c, w1, b1);
        if (!more) break;
        c = cn;
      }
    }
  }

  // cross-wave reduction through LDS, then one coalesced store per thread
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) red[(wave * 16 + 4 * i + r) * 64 + lane] = tot[i][r];
  __syncthreads();

  const int m_l = threadIdx.x >> 4, nq = threadIdx.x & 15;
  if (m_l < Mt) {
    const int src_lane = 16 * (nq >> 2) + m_l, e0 = (nq & 3) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; w++) s += red[(w * 16 + e0 + j) * 64 + src_lane];
      v[j] = s;
    }
    const int m = mt * 16 + m_l;
    const int n = strip * 64 + 4 * nq;
    if (p.splits == 1) {
      if (p.bias) {  // bias is added to the rounded GEMM result, as output.add_(bias) does
        const u32x2 bb = *reinterpret_cast<const u32x2*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = DT::to_f32(DT::from_f32(v[j]));
        v[0] += DT::to_f32((uint16_t)(bb[0] & 0xffff));
        v[1] += DT::to_f32((uint16_t)(bb[0] >> 16));
        v[2] += DT::to_f32((uint16_t)(bb[1] & 0xffff));
        v[3] += DT::to_f32((uint16_t)(bb[1] >> 16));
      }
      u32x2 o;
      o[0] = pack2<DT>(v[0], v[1]);
      o[1] = pack2<DT>(v[2], v[3]);
      *reinterpret_cast<u32x2*>(p.out + (size_t)m * p.N + n) = o;
    } else {
      *reinterpret_cast<f32x4*>(p.partial + ((size_t)sp * p.M + m) * p.N + n) =
          f32x4{v[0], v[1], v[2], v[3]};
    }
  }
}

// out[m][n] = round(sum_s partial[s][m][n] + bias[n]); fixed summation order
template <typename DT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial,
                                                            const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ out, int M,
                                                            int N, int splits) {
  const size_t total4 = (size_t)M * N / 4;
  for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < total4;
       v += (size_t)gridDim.x * blockDim.x) {
    f32x4 s = reinterpret_cast<const f32x4*>(partial)[v];
    for (int k = 1; k < splits; k++) {
      const f32x4 t = reinterpret_cast<const f32x4*>(partial + (size_t)k * M * N)[v];
      s += t;
    }
    if (bias) {
      const int n = (int)((v * 4) % N);
      const u32x2 bb = *reinterpret_cast<const u32x2*>(bias + n);
#pragma unroll
      for (int j = 0; j < 4; j++) s[j] = DT::to_f32(DT::from_f32(s[j]));
      s[0] += DT::to_f32((uint16_t)(bb[0] & 0xffff));
      s[1] += DT::to_f32((uint16_t)(bb[0] >> 16));
      s[2] += DT::to_f32((uint16_t)(bb[1] & 0xffff));
      s[3] += DT::to_f32((uint16_t)(bb[1] >> 16));
    }
    u32x2 o;
    o[0] = pack2<DT>(s[0], s[1]);
    o[1] = pack2<DT>(s[2], s[3]);
    reinterpret_cast<u32x2*>(out)[v] = o;
  }
}

static inline int chunk_steps(int G) { return (G % 128 == 0) ? 4 : (G % 64 == 0) ? 2 : 1; }

// Split-K heuristic: a block is 4 waves, each streaming whole chunks; about 4 blocks
// are resident per CU (1024 slots).  Cost model in units of "one chunk per wave".
int choose_splits(int M, int N, int K, int G, bool silu) {
  const int CS = chunk_steps(G);
  const int chunks = K / (32 * CS);
  const int strips = N / 64, mts = (M + 15) / 16;
  const char* env = getenv("MI355Q_SPLITK");
  if (env && atoi(env) > 0) return std::min(atoi(env), chunks);
  int best = 1;
  float best_cost = 1e30f;
  for (int s = 1; s <= chunks && s <= 64; s++) {
    const int per_block = (chunks + s - 1) / s;
    if (silu && (size_t)per_block * CS * 32 * std::min(M, 16) * 2 > 64 * 1024) continue;
    const int per_wave = (per_block + 3) / 4;
    const long blocks = (long)strips * mts * s;
    const int rounds = (int)((blocks + 1023) / 1024);
    const float reduce = s == 1 ? 0.f : 0.6f + 0.08f * s * (float)std::min(M, 16) / 4.0f;
    const float cost = (float)rounds * per_wave * (CS / 4.0f) + reduce + 0.5f;
    if (cost < best_cost) {
      best_cost = cost;
      best = s;
    }
  }
  return best;
}

template <typename DT, bool SILU>
static int launch_skinny(const SkinnyParams& p, int CS, hipStream_t stream, size_t lds) {
  dim3 grid(p.N / 64, p.splits, (p.M + 15) / 16), block(256);
#define MI355Q_LAUNCH(cs, zc)                                                                    \
  hipLaunchKernelGGL((w4a16_skinny_kernel<DT, cs, zc, SILU>), grid, block, lds, stream, p)
  const bool zc = p.zc != nullptr;
  if (CS == 4) { if (zc) MI355Q_LAUNCH(4, true); else MI355Q_LAUNCH(4, false); }
  else if (CS == 2) { if (zc) MI355Q_LAUNCH(2, true); else MI355Q_LAUNCH(2, false); }
  else { if (zc) MI355Q_LAUNCH(1, true); else MI355Q_LAUNCH(1, false); }
#undef MI355Q_LAUNCH
  if (int e = check_launch("w4a16_skinny_kernel")) return e;
  if (p.splits > 1) {
    const size_t total4 = (size_t)p.M * p.N / 4;
    const int g = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    hipLaunchKernelGGL((splitk_reduce_kernel<DT>), dim3(g), dim3(256), 0, stream, p.partial, p.bias,
                       p.out, p.M, p.N, p.splits);
    if (int e = check_launch("splitk_reduce_kernel")) return e;
  }
  return MI355Q_OK;
}

static int gemm_impl(const void* x, int ldx, const void* packed, const void* scales, const void* zc,
                     const void* bias, void* c, int M, int N, int K, int G, int act_dtype,
                     void* workspace, size_t workspace_bytes, int split_k, hipStream_t stream,
                     bool silu) {
  if (int e = check_shape(K, N, G)) return e;
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w4a16_gemm: act_dtype must be bf16 or fp16");
  if (M == 0) return MI355Q_OK;
  MI355Q_REQUIRE(x && packed && scales && c, "w4a16_gemm: null pointer");
  MI355Q_REQUIRE(M > 0, "w4a16_gemm: M=%d", M);
  MI355Q_REQUIRE(ldx % 8 == 0 && ldx >= (silu ? 2 * K : K), "w4a16_gemm: row stride %d must be a multiple of 8 and cover the row", ldx);
  MI355Q_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)scales & 15) == 0 &&
                     ((uintptr_t)c & 7) == 0 && ((uintptr_t)zc & 15) == 0 && ((uintptr_t)bias & 7) == 0,
                 "w4a16_gemm: pointers must be 16-byte aligned");
  const int CS = chunk_steps(G);
  SkinnyParams p;
  p.x = (const uint16_t*)x;
  p.ldx = ldx;
  p.wp = (const u32x4*)packed;
  p.scales = (const uint16_t*)scales;
  p.zc = (const uint16_t*)zc;
  p.bias = (const uint16_t*)bias;
  p.out = (uint16_t*)c;
  p.partial = (float*)workspace;
  p.M = M; p.N = N; p.K = K; p.G = G;
  p.chunks = K / (32 * CS);
  int splits = split_k > 0 ? std::min(split_k, p.chunks) : choose_splits(M, N, K, G, silu);
  size_t lds = 16384;
  if (silu) {
    const size_t row_bytes = (size_t)CS * 32 * std::min(M, 16) * 2;  // LDS bytes per chunk
    while (splits < p.chunks && (size_t)((p.chunks + splits - 1) / splits) * row_bytes > 64 * 1024) splits++;
    const int pb = (p.chunks + splits - 1) / splits;
    lds += (size_t)(pb + 1) * CS * 32 * std::min(M, 16) * 2;
    MI355Q_REQUIRE(lds <= 160 * 1024 - 1024, "w4a16_gemm_silu_mul: K range per block does not fit LDS");
  }
  p.splits = splits;
  if (splits > 1) {
    MI355Q_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "w4a16_gemm: split-K needs a 16-byte aligned workspace");
    if (workspace_bytes < (size_t)splits * M * N * sizeof(float))
      return set_error(MI355Q_ENOSPACE, "w4a16_gemm: workspace %zu B < %zu B", workspace_bytes,
                       (size_t)splits * M * N * sizeof(float));
  }
  if (act_dtype == MI355Q_BF16)
    return silu ? launch_skinny<BF16, true>(p, CS, stream, lds) : launch_skinny<BF16, false>(p, CS, stream, lds);
  return silu ? launch_skinny<FP16, true>(p, CS, stream, lds) : launch_skinny<FP16, false>(p, CS, stream, lds);
}

// ---- dequantize ---------------------------------------------------------------
template <typename DT>
__global__ void dequantize_kernel(const uint32_t* __restrict__ wp, const uint16_t* __restrict__ scales,
                                  const uint16_t* __restrict__ zc, uint16_t* __restrict__ W, int K,
                                  int N, int G) {
  const size_t nd = (size_t)K * N / 8;
  const int ksteps = K >> 5;
  for (size_t d = blockIdx.x * (size_t)blockDim.x + threadIdx.x; d < nd;
       d += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(d & 3), lane = (int)((d >> 2) & 63);
    const size_t tile = d >> 8;
    const int t = (int)(tile % ksteps), strip = (int)(tile / ksteps);
    const int rho = lane & 15, kq = lane >> 4;
    const int n = 64 * strip + 16 * (rho >> 2) + 4 * i + (rho & 3);
    const int k0 = 32 * t + 8 * kq;
    const int g = k0 / G;
    const float s = DT::to_f32(scales[(size_t)g * N + n]);
    const float z = zc ? DT::to_f32(zc[(size_t)g * N + n]) - DT::int4_bias : 8.0f;
    const uint32_t w = wp[d];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float q = (float)((w >> (((j & 1) << 4) + ((j >> 1) << 2))) & 0xfu);
      W[(size_t)(k0 + j) * N + n] = DT::from_f32((q - z) * s);
    }
  }
}

}  // namespace mi355q

extern "C" size_t mi355q_w4a16_workspace_bytes(int M, int N, int K, int G) {
  if (G <= 0) G = K;
  if (M <= 0 || N <= 0 || K <= 0 || G % 32 != 0) return 0;
  const int CS = chunk_steps(G);
  const int chunks = K / (32 * CS);
  const int smax = chunks < 64 ? chunks : 64;
  return (size_t)smax * M * N * sizeof(float);
}

extern "C" int mi355q_w4a16_gemm(const void* a, int lda, const void* packed, const void* scales,
                                 const void* zc, const void* bias, void* c, int M, int N, int K,
                                 int G, int act_dtype, void* workspace, size_t workspace_bytes,
                                 int split_k, mi355q_stream_t stream) {
  return gemm_impl(a, lda, packed, scales, zc, bias, c, M, N, K, G, act_dtype, workspace,
                   workspace_bytes, split_k, (hipStream_t)stream, false);
}

extern "C" int mi355q_w4a16_gemm_silu_mul(const void* x, int ldx, const void* packed,
                                          const void* scales, const void* zc, const void* bias,
                                          void* c, int M, int N, int K, int G, int act_dtype,
                                          void* workspace, size_t workspace_bytes, int split_k,
                                          mi355q_stream_t stream) {
  return gemm_impl(x, ldx, packed, scales, zc, bias, c, M, N, K, G, act_dtype, workspace,
                   workspace_bytes, split_k, (hipStream_t)stream, true);
}

extern "C" int mi355q_w4a16_dequantize(const void* packed, const void* scales, const void* zc,
                                       void* w, int K, int N, int G, int act_dtype,
                                       mi355q_stream_t stream) {
  if (int e = check_shape(K, N, G)) return e;
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w4a16_dequantize: act_dtype must be bf16 or fp16");
  MI355Q_REQUIRE(packed && scales && w, "w4a16_dequantize: null pointer");
  const size_t nd = (size_t)K * N / 8;
  const int grid = (int)((nd + 255) / 256 < 8192 ? (nd + 255) / 256 : 8192);
  if (act_dtype == MI355Q_BF16)
    hipLaunchKernelGGL((dequantize_kernel<BF16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)packed, (const uint16_t*)scales, (const uint16_t*)zc,
                       (uint16_t*)w, K, N, G);
  else
    hipLaunchKernelGGL((dequantize_kernel<FP16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)packed, (const uint16_t*)scales, (const uint16_t*)zc,
                       (uint16_t*)w, K, N, G);
  return check_launch("dequantize_kernel");
}
