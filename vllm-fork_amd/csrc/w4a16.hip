// W4A16 dequant-GEMM for gfx950: tiled int4 layouts, skinny (M <= 16 per pass) MFMA
// kernel with in-block K split (and an optional deterministic cross-block split),
// load-time repack, dequantize.
//
// Packed layout: the [K, N] int4 matrix is cut into strips of SW output channels
// (SW = 64, 32 or 16, chosen per layer so that even narrow layers give every CU a strip)
// and each strip into 1 KiB tiles of SW channels x (2048 / SW) input channels, consecutive
// along K.  With NS = SW / 16 and KT = 4 / NS, lane l of a wave owns bytes [16 l, 16 l + 16)
// of a tile: four dwords, dword i = ks * NS + sub holding the 8 nibbles of
//     n = SW strip + (SW/4) ((l & 15) >> 2) + 4 sub + (l & 3)
//     k = 32 (KT tile + ks) + 8 (l >> 4) + j,   j = 0..7
// with nibble j = 2p at bits [4p, 4p+4) and j = 2p+1 at bits [16 + 4p, 16 + 4p + 4).
// One 16-byte load per lane therefore gives the A fragments of four
// v_mfma_f32_16x16x32 tiles, and ((w >> 4p) & 0x000F000F) | magic is directly the pair of
// 16-bit floats (bias + q) for fragment register p.
//
// Arithmetic: per scale group, acc = sum_k a_k (bias + q_k) on the matrix cores (exact
// products, fp32 accumulation), a1 = sum_k a_k from one extra MFMA with an all-ones A
// operand, and the group contributes s * (acc - (bias + z) * a1) to the fp32 total.
//
// Latency shape: a kernel on this chip costs ~1.6 us to launch plus ~1 us per dependent
// global-memory round trip, so the kernel requests weights, scales and activations of its
// first chunks at entry, splits K over up to 16 waves of one workgroup (combined through
// LDS) and writes the final 16-bit result itself; the cross-block split (ticket + last
// arriver) is only used when a layer has too few strips to occupy the chip.
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

__host__ __device__ static inline int nibble_shift(int j) { return ((j & 1) << 4) + ((j >> 1) << 2); }

// Strip widths that are not 16, 32 or 64 (NS = sw / 16 not a divisor of 4) use the same rule
// written per 128-input-channel chunk: a chunk of a strip is NS consecutive 1 KiB planes and
// dword e = ks * NS + sub of a lane's 4 NS dwords sits in plane e / 4 at position e % 4.  For
// NS = 1, 2, 4 and K % 128 == 0 both rules give the same bytes.
__host__ __device__ static inline bool chunk_rule(int ns) { return 4 % ns != 0; }

// index of the packed dword holding (k, n) and the bit offset of its nibble
__host__ __device__ static inline void packed_pos(int K, int sw, int k, int n, size_t* dword,
                                                  int* shift) {
  const int ns = sw >> 4, qw = sw >> 2;
  const int strip = n / sw, nl = n % sw;
  const int q = nl / qw, sub = (nl % qw) >> 2, r = nl & 3;
  const int kstep = k >> 5;
  const int lane = (((k >> 3) & 3) << 4) | (q << 2) | r;
  if (chunk_rule(ns)) {
    const int c = kstep >> 2, e = (kstep & 3) * ns + sub;
    *dword = ((((size_t)strip * (K >> 7) + c) * ns + (e >> 2)) * 64 + lane) * 4 + (e & 3);
  } else {
    const int kt = 4 / ns, tile = kstep / kt, ks = kstep % kt;
    *dword = (((size_t)strip * (K / (32 * kt)) + tile) * 64 + lane) * 4 + ks * ns + sub;
  }
  *shift = nibble_shift(k & 7);
}

// (k0, n) of packed dword d: its nibbles are k0 .. k0 + 7 of channel n
__host__ __device__ static inline void dword_pos(int K, int sw, size_t d, int* k0, int* n) {
  const int ns = sw >> 4;
  const int i = (int)(d & 3), lane = (int)((d >> 2) & 63);
  const size_t tile_id = d >> 8;   // 1 KiB plane index
  const int rho = lane & 15, kq = lane >> 4;
  int strip, sub, kstep;
  if (chunk_rule(ns)) {
    const int chunks = K >> 7;
    const int plane = (int)(tile_id % ns);
    const size_t cid = tile_id / ns;
    const int c = (int)(cid % chunks), e = plane * 4 + i;
    strip = (int)(cid / chunks);
    sub = e % ns;
    kstep = 4 * c + e / ns;
  } else {
    const int kt = 4 / ns, tiles = K / (32 * kt);
    const int tile = (int)(tile_id % tiles);
    strip = (int)(tile_id / tiles);
    sub = i % ns;
    kstep = kt * tile + i / ns;
  }
  *n = sw * strip + (sw >> 2) * (rho >> 2) + 4 * sub + (rho & 3);
  *k0 = 32 * kstep + 8 * kq;
}

// With MI355Q_W_PAIR_HALVES, packed column n holds checkpoint column n/2 of the first half
// (n even) or of the second half (n odd): gate_i and up_i of a merged gate_up projection
// become neighbours, so one workgroup can finish silu(gate_i) * up_i in its epilogue.
__host__ __device__ static inline int pair_src(int n, int N, int layout) {
  return (layout & MI355Q_W_PAIR_HALVES) ? (n >> 1) + ((n & 1) ? (N >> 1) : 0) : n;
}

__host__ __device__ static inline uint32_t repack_dword(const int32_t* qw, int layout, int K, int N,
                                                        int sw, size_t d) {
  int k0, n;
  dword_pos(K, sw, d, &k0, &n);
  n = pair_src(n, N, layout);
  layout &= 0xff;
  uint32_t w = 0;
  if (layout == MI355Q_W_GPTQ) {
    const uint32_t s = (uint32_t)qw[(size_t)(k0 >> 3) * N + n];
#pragma unroll
    for (int j = 0; j < 8; j++) w |= ((s >> (4 * j)) & 0xfu) << nibble_shift(j);
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) w |= src_nibble(qw, layout, N, k0 + j, n) << nibble_shift(j);
  }
  return w;
}

// Scales and zero points are stored strip-major, [N / sw][K / G][sw]: everything one workgroup
// needs is one contiguous run.
__host__ __device__ static inline size_t scale_idx(int g, int n, int groups, int sw) {
  return ((size_t)(n / sw) * groups + g) * sw + n % sw;
}

__host__ __device__ static inline uint16_t repack_zc(const int32_t* qz, int zp_mode, int N, int g,
                                                     int n, int act_dtype, int layout) {
  n = pair_src(n, N, layout);
  const int pos = zp_mode == MI355Q_ZP_AWQ ? awq_pos(n) : (n & 7);
  uint32_t z = ((uint32_t)qz[(size_t)g * (N >> 3) + (n >> 3)] >> (4 * pos)) & 0xfu;
  if (zp_mode == MI355Q_ZP_GPTQ_V1) z = (z + 1) & 0x1fu;
  // bias + z is exactly representable: bf16(128 + z) = 0x4300 | z, fp16(1024 + z) = 0x6400 | z
  return act_dtype == MI355Q_BF16 ? (uint16_t)(0x4300u + z) : (uint16_t)(0x6400u + z);
}

__global__ void repack_weight_kernel(const int32_t* __restrict__ qw, uint32_t* __restrict__ out,
                                     int K, int N, int layout, int sw, size_t ndwords) {
  for (size_t d = blockIdx.x * (size_t)blockDim.x + threadIdx.x; d < ndwords;
       d += (size_t)gridDim.x * blockDim.x)
    out[d] = repack_dword(qw, layout, K, N, sw, d);
}

__global__ void repack_zc_kernel(const int32_t* __restrict__ qz, uint16_t* __restrict__ zc, int N,
                                 int groups, int zp_mode, int act_dtype, int layout, int sw) {
  const size_t total = (size_t)groups * N;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x)
    zc[scale_idx((int)(e / N), (int)(e % N), groups, sw)] =
        repack_zc(qz, zp_mode, N, (int)(e / N), (int)(e % N), act_dtype, layout);
}

static inline int chunk_steps(int G) { return (G % 128 == 0) ? 4 : (G % 64 == 0) ? 2 : 1; }

static int check_shape(int K, int N, int& G) {
  if (G <= 0) G = K;
  MI355Q_REQUIRE(K > 0 && N > 0, "w4a16: K=%d N=%d must be positive", K, N);
  MI355Q_REQUIRE(K % 32 == 0, "w4a16: K=%d must be a multiple of 32", K);
  MI355Q_REQUIRE(N % 64 == 0, "w4a16: N=%d must be a multiple of 64", N);
  MI355Q_REQUIRE(G % 32 == 0 && K % G == 0, "w4a16: group size %d must divide K=%d and be a multiple of 32", G, K);
  return 0;
}

// Strip widths the split-K capable kernel (w4a16_skinny_kernel) can run: one of its three widths,
// and a 1 KiB tile (32 * KT input channels) must not straddle a scale group.
static bool legacy_kernel_ok(int sw, int K, int G) {
  if (sw != 64 && sw != 32 && sw != 16) return false;
  const int kt = 4 / (sw >> 4);
  return chunk_steps(G) % kt == 0 && K % (32 * kt) == 0;
}
// Is `sw` a usable strip width for a [K, N] layer with group size G?  Whole 128-k chunks with
// groups of 32, 64 or a multiple of 128 channels suit every width (the decode kernel folds a group
// inside a chunk); other shapes need what the split-K capable kernel needs.
static bool tile_ok(int sw, int K, int G) {
  if (sw < 16 || sw > 128 || sw % 16) return false;
  if (K % 128 == 0 && (G % 128 == 0 || G == 64 || G == 32)) return true;
  return legacy_kernel_ok(sw, K, G);
}
// strip widths the split-K capable kernel (w4a16_skinny_kernel) is built for
static bool legacy_tile(int sw) { return sw == 64 || sw == 32 || sw == 16; }

static int check_tile(int sw, int K, int G, int N = 0) {
  MI355Q_REQUIRE(tile_ok(sw, K, G), "w4a16: tile width %d is not usable with K=%d, group size %d", sw, K, G);
  MI355Q_REQUIRE(N % sw == 0, "w4a16: N=%d is not a multiple of the tile width %d", N, sw);
  return 0;
}

static int check_enums(int src_layout, int zp_mode, int act_dtype, const void* qzeros,
                       const void* zc) {
  MI355Q_REQUIRE((src_layout & 0xff) >= MI355Q_W_GPTQ && (src_layout & 0xff) <= MI355Q_W_AWQ &&
                     (src_layout & ~0x1ff) == 0, "w4a16: bad src_layout %d", src_layout);
  MI355Q_REQUIRE(zp_mode >= MI355Q_ZP_CONST && zp_mode <= MI355Q_ZP_GPTQ_V1, "w4a16: bad zp_mode %d", zp_mode);
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w4a16: act_dtype must be bf16 or fp16");
  if (zp_mode != MI355Q_ZP_CONST)
    MI355Q_REQUIRE(qzeros != nullptr && zc != nullptr, "w4a16: zero-point mode %d needs qzeros and zc", zp_mode);
  return 0;
}

}  // namespace mi355q

using namespace mi355q;

extern "C" size_t mi355q_w4a16_packed_bytes(int K, int N) { return (size_t)K * N / 2; }

// Strip width for a layer: the narrowest strips that still leave each of the 256 CUs about
// one workgroup, so that no cross-block split of K is needed.
extern "C" int mi355q_w4a16_choose_tile(int K, int N, int G) {
  if (G <= 0) G = K;
  const char* env = env_value("MI355Q_TILE");
  if (env && tile_ok(atoi(env), K, G)) return atoi(env);
  int sw = N >= 16384 ? 64 : (N >= 6144 ? 32 : 16);
  while (sw < 64 && !legacy_kernel_ok(sw, K, G)) sw <<= 1;   // the widths the split-K capable kernel runs
  return sw;
}

extern "C" int mi355q_w4a16_repack(const int32_t* qweight, const int32_t* qzeros, void* packed,
                                   void* zc, int K, int N, int G, int tile, int src_layout,
                                   int zp_mode, int act_dtype, mi355q_stream_t stream) {
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_tile(tile, K, G, N)) return e;
  if (int e = check_enums(src_layout, zp_mode, act_dtype, qzeros, zc)) return e;
  MI355Q_REQUIRE(qweight && packed, "w4a16_repack: null pointer");
  const size_t nd = (size_t)K * N / 8;
  const int grid = (int)((nd + 255) / 256 < 8192 ? (nd + 255) / 256 : 8192);
  hipLaunchKernelGGL(repack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, qweight,
                     (uint32_t*)packed, K, N, src_layout, tile, nd);
  if (int e = check_launch("repack_weight_kernel")) return e;
  if (zp_mode != MI355Q_ZP_CONST) {
    const size_t nz = (size_t)(K / G) * N;
    const int gz = (int)((nz + 255) / 256 < 4096 ? (nz + 255) / 256 : 4096);
    hipLaunchKernelGGL(repack_zc_kernel, dim3(gz), dim3(256), 0, (hipStream_t)stream, qzeros,
                       (uint16_t*)zc, N, K / G, zp_mode, act_dtype, src_layout, tile);
    if (int e = check_launch("repack_zc_kernel")) return e;
  }
  return MI355Q_OK;
}

extern "C" int mi355q_w4a16_repack_host(const int32_t* qweight, const int32_t* qzeros,
                                        void* packed, void* zc, int K, int N, int G, int tile,
                                        int src_layout, int zp_mode, int act_dtype) {
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_tile(tile, K, G, N)) return e;
  if (int e = check_enums(src_layout, zp_mode, act_dtype, qzeros, zc)) return e;
  MI355Q_REQUIRE(qweight && packed, "w4a16_repack_host: null pointer");
  const size_t nd = (size_t)K * N / 8;
  uint32_t* out = (uint32_t*)packed;
  for (size_t d = 0; d < nd; d++) out[d] = repack_dword(qweight, src_layout, K, N, tile, d);
  if (zp_mode != MI355Q_ZP_CONST) {
    uint16_t* z = (uint16_t*)zc;
    for (int g = 0; g < K / G; g++)
      for (int n = 0; n < N; n++)
        z[scale_idx(g, n, K / G, tile)] = repack_zc(qzeros, zp_mode, N, g, n, act_dtype, src_layout);
  }
  return MI355Q_OK;
}

extern "C" int mi355q_w4a16_unpack_host(const void* packed, int32_t* q, int K, int N, int tile) {
  int G = K;
  if (int e = check_shape(K, N, G)) return e;
  MI355Q_REQUIRE(tile_ok(tile, K, G) && N % tile == 0, "w4a16_unpack_host: bad tile width %d", tile);
  const uint32_t* p = (const uint32_t*)packed;
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) {
      size_t d;
      int sh;
      packed_pos(K, tile, k, n, &d, &sh);
      q[(size_t)k * N + n] = (int32_t)((p[d] >> sh) & 0xfu);
    }
  return MI355Q_OK;
}

// ---------------------------------------------------------------------------
// skinny GEMM kernel
// ---------------------------------------------------------------------------
namespace mi355q {

// PRO_MIXNORM (the one-workgroup-per-strip kernel only): PRO_ADDNORM whose input row is the weighted sum of a routed
// layer's two selected expert outputs, x = round(w0 y0 + w1 y1) in fp32 in slot order -- the combine of moe_sum_kernel
enum { PRO_NONE = 0, PRO_SILU = 1, PRO_ADDNORM = 2, PRO_MIXNORM = 3 };

struct SkinnyParams {
  const uint16_t* x;       // [M, K]; PRO_SILU: [M, 2K] gate|up; PRO_ADDNORM: the un-normed input
  int ldx;
  const u32x4* wp;         // packed tiles
  const uint16_t* scales;  // [K/G, N]
  const uint16_t* zc;      // [K/G, N] or null
  const uint16_t* bias;    // [N] or null
  uint16_t* out;           // [M, N]
  float* partial;          // [splits, M, N] split-K partial sums (splits > 1)
  int* counters;           // [mtiles * strips] arrival tickets, zero between launches
  // PRO_ADDNORM
  const uint16_t* res_in;  // [M, K] or null
  uint16_t* res_out;       // [M, K]
  const uint16_t* norm_w;  // [K] or null
  float eps;
  int M, N, K, G;
  int chunks;              // K / (32 * CS)
  int splits;
  int nw;                  // waves per workgroup
  // Row tiling: blockIdx.z owns output rows [z * mrows, z * mrows + mrows).  Dense GEMMs use 16
  // (one MFMA tile of rows).  Mixture-of-experts launches use 1: output row z is one
  // (token, selected expert) slot, its weights are those of expert_ids[z] (negative: the row is
  // zero-filled) and its input row is z / x_row_div.
  int mrows;
  int x_row_div;
  const int* expert_ids;   // [M] or null
  long long wp_estride;    // u32x4 units between experts
  long long sc_estride;    // 16-bit elements between experts (scales and zc)
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

template <int ND>
struct Dwords {
  uint32_t v[ND];
};
// ND consecutive dwords from a 4*ND-byte aligned address, as the widest loads
template <int ND>
__device__ inline Dwords<ND> load_dwords(const uint16_t* p) {
  Dwords<ND> r;
  if constexpr (ND == 8) {
    const u32x4 a = reinterpret_cast<const u32x4*>(p)[0], b = reinterpret_cast<const u32x4*>(p)[1];
#pragma unroll
    for (int i = 0; i < 4; i++) { r.v[i] = a[i]; r.v[4 + i] = b[i]; }
  } else if constexpr (ND == 4) {
    const u32x4 a = reinterpret_cast<const u32x4*>(p)[0];
#pragma unroll
    for (int i = 0; i < 4; i++) r.v[i] = a[i];
  } else {
    const u32x2 a = reinterpret_cast<const u32x2*>(p)[0];
    r.v[0] = a[0];
    r.v[1] = a[1];
  }
  return r;
}

// SW = strip width, CS = k-steps (of 32) per flush chunk (a chunk never crosses a scale
// group).  Dynamic LDS: [0, nw * NS KiB) cross-wave reduction / prologue scratch, then 16
// bytes of flags, then the staged activations of the LDS prologues.
// Workgroups of the 64-wide tile hold at most 8 waves (256-register budget per lane).
template <typename DT, int SW, int CS, bool HAS_ZC, int PRO>
__global__ __launch_bounds__(SW == 64 ? 512 : 1024) void w4a16_skinny_kernel(const SkinnyParams p) {
  constexpr int NS = SW / 16;      // 16-channel MFMA subtiles per k-step
  constexpr int KT = 4 / NS;       // k-steps per 1 KiB tile
  constexpr int CT = CS / KT;      // tiles per chunk
  constexpr int NE = NS * 4;       // outputs per lane
  constexpr int ND = SW / 8;       // dwords of 16-bit scales per lane and group
  constexpr int PF = SW == 16 ? 4 : 2;  // weight chunks in flight per wave (even)
  constexpr bool LDS_B = PRO != PRO_NONE;
  static_assert(CS % KT == 0, "chunk must hold whole tiles");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nw = p.nw;
  const int red_bytes = nw * NS * 1024;
  float* red = reinterpret_cast<float*>(smem);                          // [nw][NE][64] fp32
  int* flag = reinterpret_cast<int*>(smem + red_bytes);
  uint16_t* act = reinterpret_cast<uint16_t*>(smem + red_bytes + 16);   // [Mt][krange]

  const int nthreads = nw * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x, sp = blockIdx.y, mt = blockIdx.z;
  const int kq = lane >> 4, ml = lane & 15;
  const int tiles = p.K / (32 * KT);
  const int c0 = (int)(((long long)p.chunks * sp) / p.splits);
  const int c1 = (int)(((long long)p.chunks * (sp + 1)) / p.splits);
  const int row0 = mt * p.mrows;
  const int Mt = min(p.mrows, p.M - row0);
  const int mrow = row0 + min(ml, Mt - 1);

  const u32x4* wexp = p.wp;
  const uint16_t* sc_e = p.scales;
  const uint16_t* zc_e = p.zc;
  if (p.expert_ids) {
    const int expert = p.expert_ids[mt];
    if (expert < 0) {  // slot routed to an expert this rank does not hold: the row is zero
      if (sp == 0 && threadIdx.x < SW / 4)
        *reinterpret_cast<u32x2*>(p.out + (size_t)row0 * p.N + strip * SW + 4 * threadIdx.x) = u32x2{0u, 0u};
      return;
    }
    wexp += (size_t)expert * p.wp_estride;
    sc_e += (size_t)expert * p.sc_estride;
    if (zc_e) zc_e += (size_t)expert * p.sc_estride;
  }
  const u32x4* wstrip = wexp + (size_t)strip * tiles * 64 + lane;
  const uint16_t* xrow = p.x + (size_t)(mrow / p.x_row_div) * p.ldx + 8 * kq;
  const int kbase = c0 * CS * 32;  // first k of this block's range
  const int krange = (c1 - c0) * CS * 32;
  const uint16_t* arow = act + (size_t)min(ml, Mt - 1) * krange + 8 * kq;
  const int groups = p.K / p.G;
  const size_t sc_off = (size_t)strip * groups * SW + (SW / 4) * kq;  // + g * SW: strip-major scales

  u32x4 w[PF][CT], b[2][CS];  // activations: one chunk ahead
  Dwords<ND> sc[PF];

  auto issue_w = [&](int c, int slot) {
#pragma unroll
    for (int t = 0; t < CT; t++) w[slot][t] = __builtin_nontemporal_load(wstrip + (size_t)(c * CT + t) * 64);
    const int g = (c * CS * 32) / p.G;
    sc[slot] = load_dwords<ND>(sc_e + (size_t)g * SW + sc_off);
  };
  auto issue_b = [&](int c, int slot) {
#pragma unroll
    for (int u = 0; u < CS; u++) {
      const int t = c * CS + u;
      if constexpr (LDS_B)
        b[slot][u] = *reinterpret_cast<const u32x4*>(arow + (t * 32 - kbase));
      else
        b[slot][u] = *reinterpret_cast<const u32x4*>(xrow + t * 32);
    }
  };

  // everything this wave needs first is requested before the prologue
  const int cw = c0 + wave;
#pragma unroll
  for (int s = 0; s < PF; s++) {
    const int c = cw + s * nw;
    if (c < c1) issue_w(c, s);
  }
  if constexpr (!LDS_B) {
    if (cw < c1) issue_b(cw, 0);
  }

  if constexpr (PRO == PRO_SILU) {
    // stage act[m][k - kbase] = silu(gate) * up for the block's k range, rows < Mt
    const int vecs = krange >> 3;
    for (int idx = threadIdx.x; idx < Mt * vecs; idx += nthreads) {
      const int m = idx / vecs, v = idx - m * vecs;
      const uint16_t* src = p.x + (size_t)((row0 + m) / p.x_row_div) * p.ldx + kbase + 8 * v;
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

  if constexpr (PRO == PRO_ADDNORM) {
    // z = round(x + res_in) over the whole row gives the variance; the block stages
    // round(z * rs * w) for its own k range and strip 0 writes z to res_out for that range.
    // Rows are handled 4 at a time.
    const int vecs_row = p.K >> 3;
    for (int mb = 0; mb < Mt; mb += 4) {
      const int nm = min(4, Mt - mb);
      float ss[4] = {0.f, 0.f, 0.f, 0.f};
      for (int v = threadIdx.x; v < vecs_row; v += nthreads) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
          if (m < nm) {
            const size_t off = (size_t)(row0 + mb + m);
            const u32x4 xv = *reinterpret_cast<const u32x4*>(p.x + off * p.ldx + 8 * v);
            u32x4 rv = u32x4{0u, 0u, 0u, 0u};
            if (p.res_in) rv = *reinterpret_cast<const u32x4*>(p.res_in + off * p.K + 8 * v);
#pragma unroll
            for (int e = 0; e < 4; e++) {
              float a = DT::to_f32((uint16_t)(xv[e] & 0xffff)), c = DT::to_f32((uint16_t)(xv[e] >> 16));
              if (p.res_in) {
                a = DT::to_f32(DT::from_f32(a + DT::to_f32((uint16_t)(rv[e] & 0xffff))));
                c = DT::to_f32(DT::from_f32(c + DT::to_f32((uint16_t)(rv[e] >> 16))));
              }
              ss[m] += a * a + c * c;
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 4; m++) ss[m] = wave_sum(ss[m]);
      __syncthreads();
      if (lane == 0) {
#pragma unroll
        for (int m = 0; m < 4; m++) red[wave * 4 + m] = ss[m];
      }
      __syncthreads();
      float rs[4];
#pragma unroll
      for (int m = 0; m < 4; m++) {
        float t = 0.f;
        for (int i = 0; i < nw; i++) t += red[i * 4 + m];
        rs[m] = rsqrtf(t / (float)p.K + p.eps);
      }
      const int vecs = krange >> 3;
      for (int idx = threadIdx.x; idx < nm * vecs; idx += nthreads) {
        const int m = idx / vecs, v = idx - m * vecs;
        const size_t off = (size_t)(row0 + mb + m);
        const int k = kbase + 8 * v;
        const u32x4 xv = *reinterpret_cast<const u32x4*>(p.x + off * p.ldx + k);
        u32x4 rv = u32x4{0u, 0u, 0u, 0u}, wv = u32x4{0u, 0u, 0u, 0u};
        if (p.res_in) rv = *reinterpret_cast<const u32x4*>(p.res_in + off * p.K + k);
        if (p.norm_w) wv = *reinterpret_cast<const u32x4*>(p.norm_w + k);
        const float r_m = m == 0 ? rs[0] : (m == 1 ? rs[1] : (m == 2 ? rs[2] : rs[3]));
        u32x4 zo, o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float a = DT::to_f32((uint16_t)(xv[e] & 0xffff)), c = DT::to_f32((uint16_t)(xv[e] >> 16));
          if (p.res_in) {
            a = DT::to_f32(DT::from_f32(a + DT::to_f32((uint16_t)(rv[e] & 0xffff))));
            c = DT::to_f32(DT::from_f32(c + DT::to_f32((uint16_t)(rv[e] >> 16))));
          }
          zo[e] = pack2<DT>(a, c);
          a *= r_m;
          c *= r_m;
          if (p.norm_w) {
            a *= DT::to_f32((uint16_t)(wv[e] & 0xffff));
            c *= DT::to_f32((uint16_t)(wv[e] >> 16));
          }
          o[e] = pack2<DT>(a, c);
        }
        *reinterpret_cast<u32x4*>(act + (size_t)(mb + m) * krange + 8 * v) = o;
        if (strip == 0 && p.res_out) *reinterpret_cast<u32x4*>(p.res_out + off * p.K + k) = zo;
      }
      __syncthreads();
    }
  }

  f32x4 tot[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) tot[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = u32x4{DT::ones, DT::ones, DT::ones, DT::ones};

  auto consume = [&](int c, int slot, int bslot) {
    Dwords<ND> zz;
    if constexpr (HAS_ZC) zz = load_dwords<ND>(zc_e + (size_t)((c * CS * 32) / p.G) * SW + sc_off);
    f32x4 acc[NS], acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NS; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < CT; t++) {
#pragma unroll
      for (int ks = 0; ks < KT; ks++) {
        const u32x4 bf = b[bslot][t * KT + ks];
        acc1 = mfma16<DT>(ones, bf, acc1);
#pragma unroll
        for (int sub = 0; sub < NS; sub++)
          acc[sub] = mfma16<DT>(dequant_frag<DT>(w[slot][t][ks * NS + sub]), bf, acc[sub]);
      }
    }
    // flush: output channel e = 4 sub + r of this lane's NE (n = SW strip + (SW/4) kq + e)
#pragma unroll
    for (int sub = 0; sub < NS; sub++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int e = 4 * sub + r;
        const uint32_t sw_ = sc[slot].v[e >> 1];
        const float s = DT::to_f32((uint16_t)((e & 1) ? (sw_ >> 16) : (sw_ & 0xffff)));
        float z;
        if constexpr (HAS_ZC) {
          const uint32_t zw = zz.v[e >> 1];
          z = DT::to_f32((uint16_t)((e & 1) ? (zw >> 16) : (zw & 0xffff)));
        } else {
          z = DT::int4_bias + 8.0f;
        }
        const float d = __builtin_fmaf(-z, acc1[r], acc[sub][r]);
        tot[sub][r] = __builtin_fmaf(s, d, tot[sub][r]);
      }
    }
  };

  for (int base = cw; base < c1; base += PF * nw) {
#pragma unroll
    for (int s = 0; s < PF; s++) {
      const int c = base + s * nw;
      if (c < c1) {
        if constexpr (LDS_B) {
          issue_b(c, s & 1);
        } else {
          if (c + nw < c1) issue_b(c + nw, (s + 1) & 1);
        }
        consume(c, s, s & 1);
        const int cn = c + PF * nw;
        if (cn < c1) issue_w(cn, s);
      }
    }
  }

  // cross-wave reduction through LDS
#pragma unroll
  for (int sub = 0; sub < NS; sub++)
#pragma unroll
    for (int r = 0; r < 4; r++) red[(wave * NE + 4 * sub + r) * 64 + lane] = tot[sub][r];
  __syncthreads();

  // thread tau < 4 SW finishes 4 consecutive channels of one activation row
  constexpr int QW = SW / 4;
  const int m_l = threadIdx.x / QW, nq = threadIdx.x % QW;
  const bool active = threadIdx.x < 4 * SW && m_l < Mt;
  const int m = row0 + m_l;
  const int n = strip * SW + 4 * nq;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const int nl = 4 * nq;
    const int src_lane = 16 * (nl / QW) + m_l, e0 = nl % QW;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float s = 0.f;
      for (int i = 0; i < nw; i++) s += red[(i * NE + e0 + j) * 64 + src_lane];
      v[j] = s;
    }
  }

  if (p.splits > 1) {
    // Deterministic cross-block split: every block publishes its fp32 tile with write-through
    // (agent-scope) stores, draws a ticket, and the block that draws the last ticket of the
    // tile sums all partial tiles in split order.
    typedef unsigned long long u64;
    if (active) {
      u64* dst = reinterpret_cast<u64*>(p.partial + ((size_t)sp * p.M + m) * p.N + n);
      const u64 lo = (u64)__float_as_uint(v[0]) | ((u64)__float_as_uint(v[1]) << 32);
      const u64 hi = (u64)__float_as_uint(v[2]) | ((u64)__float_as_uint(v[3]) << 32);
      __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      int* cnt = p.counters + (size_t)mt * gridDim.x + strip;
      // No release fence in front of the ticket: every published word was written by an agent-scope atomic
      // store (write-through) and waited for (vmcnt(0) + barrier above).  A fence here writes the L2 back in
      // every workgroup: measured -28 % on decode at 32 x 1024 tokens and +36 % on TTFT(2048).
      const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == p.splits - 1;
      if (last) {
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    if (active) {
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = 0.f;
      for (int s = 0; s < p.splits; s++) {
        const u64* src = reinterpret_cast<const u64*>(p.partial + ((size_t)s * p.M + m) * p.N + n);
        const u64 lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[0] += __uint_as_float((uint32_t)lo);
        v[1] += __uint_as_float((uint32_t)(lo >> 32));
        v[2] += __uint_as_float((uint32_t)hi);
        v[3] += __uint_as_float((uint32_t)(hi >> 32));
      }
    }
  }

  if (active) {
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
  }
}

// ---------------------------------------------------------------------------
// Burst kernel: the decode-time GEMV for scale groups that are multiples of 128.
//
// One workgroup per strip of SW = 16 NS output channels, strips sized so that a layer gives
// about one workgroup per CU (e.g. 112 channels for N = 28672 on 256 CUs).  A CU ingests
// HBM data at ~10 B/cycle however it is asked, so the only lever is to have every CU asking
// all the time: each wave requests its first CPW chunks (NS KiB each; for most layers that is
// everything it will ever consume) at kernel entry, before anything else can stall it,
// consumes them in arrival order and refills a slot the moment it is consumed.  What the prologue needs (the activation row, its residual, the norm
// weight, the strip's scales) is requested BEFORE the weights, because vector loads return
// in issue order; its arithmetic then overlaps the weight stream.  Scales (and zero points)
// of the whole strip are staged in LDS once.  Rows are processed rows_pp at a time (grid z).
// ---------------------------------------------------------------------------
struct BurstParams {
  const uint16_t* x;
  int ldx;
  const u32x4* wp;
  const uint16_t* scales;
  const uint16_t* zc;
  const uint16_t* bias;
  uint16_t* out;
  const uint16_t* res_in;
  uint16_t* res_out;
  const uint16_t* norm_w;
  float eps;
  int M, N, K;
  int chunks;   // K / 128
  int groups;   // K / G
  int kpg;      // 32-k steps per scale group (G / 32)
  int rows_pp;  // activation rows per pass
  int nw;       // waves per workgroup; chunks == nw * CPW * bursts
  int bursts;
  int epi_silu;  // 1: channels are (gate, up) pairs; write silu(gate) * up, N / 2 values per row
  // mixture of experts (rows_pp == 1): output row z is one (token, selected expert) slot; its
  // weights are those of expert_ids[z] (negative: the row is zero-filled), its input row z / x_row_div
  const int* expert_ids;
  int x_row_div;
  long long wp_estride;   // u32x4 units between experts
  long long sc_estride;   // 16-bit elements between experts (scales and zc)
  unsigned long long* stamps;  // diagnostics: 4 x 100 MHz timestamps per workgroup, or null
  const float* mix_w;          // PRO_MIXNORM: [M, 2] routing weights; x holds [M, 2, K] expert outputs (ldx = K)
};

static unsigned long long* g_burst_stamps = nullptr;

constexpr int BURST_XV = 2;  // 16-byte activation vectors a thread may hold in the single-row prologue

// threads a workgroup of the burst kernel may have: the deep rings need the registers of 512-thread workgroups
constexpr int burst_threads(int ns, int cpw) { return (ns * cpw >= 7 && !(ns == 7 && cpw == 1)) ? 512 : 1024; }

// SMALLG: scale groups of 32 or 64 input channels (one or two k-steps): the group sum is folded into
// the total inside the k-step loop instead of once per 128-k chunk.
template <typename DT, int NS, int CPW, bool HAS_ZC, int PRO, bool SMALLG = false>
__global__ __launch_bounds__(burst_threads(NS, CPW)) void w4a16_burst_kernel(const BurstParams p) {
  constexpr int SW = 16 * NS, QW = 4 * NS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nw = p.nw, nthreads = nw * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x;
  const int kq = lane >> 4, ml = lane & 15;
  const int row0 = blockIdx.z * p.rows_pp;
  const int Mt = min(p.rows_pp, p.M - row0);
  const int groups = p.groups;
  const int astride = p.K + 8;  // +16 B per row: rows land on different LDS banks

  float* red = reinterpret_cast<float*>(smem);                                 // [nw][rows_pp][SW]
  uint16_t* sc_l = reinterpret_cast<uint16_t*>(smem + (size_t)nw * p.rows_pp * SW * 4);  // [groups][SW]
  uint16_t* zc_l = sc_l + (size_t)groups * SW;                                 // [groups][SW] (HAS_ZC)
  uint16_t* act = zc_l + (HAS_ZC ? (size_t)groups * SW : 0);                   // [rows_pp][astride]

  const u32x4* wstrip = p.wp + (size_t)strip * p.chunks * NS * 64;   // wave-uniform: the base of a buffer descriptor
  const uint16_t* scales_e = p.scales;
  const uint16_t* zc_e = p.zc;
  int xrow0 = row0;
  if (p.expert_ids) {
    const int e = __builtin_amdgcn_readfirstlane(p.expert_ids[row0]);
    if (e < 0) {   // expert held by another rank: the slot contributes nothing
      const int ncols = p.epi_silu ? SW / 2 : SW;
      uint16_t* o = p.out + (size_t)row0 * (p.epi_silu ? p.N >> 1 : p.N) + (size_t)strip * ncols;
      for (int i = threadIdx.x; i < ncols; i += nthreads) o[i] = 0;
      return;
    }
    wstrip += (size_t)e * p.wp_estride;
    scales_e += (size_t)e * p.sc_estride;
    if (HAS_ZC) zc_e += (size_t)e * p.sc_estride;
    xrow0 = row0 / p.x_row_div;
  }
  const int cpw_total = CPW * p.bursts;
  const int cfirst = wave * cpw_total;  // this wave's chunks: [cfirst, cfirst + cpw_total)

  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
  // ---- requests, oldest first: single-row prologue inputs, scales, then the weight burst.
  // Every load here is unconditional (indices clamped into range, absent operands pointed at the input
  // row): a load under a lane- or argument-dependent branch makes the compiler merge its result with a
  // constant at the join and wait for ALL outstanding loads there -- before the weight burst below has
  // even been requested, which costs the kernel a whole memory round trip.
  const int vecs_row = p.K >> 3;
  constexpr bool NORM = PRO == PRO_ADDNORM || PRO == PRO_MIXNORM;
  u32x4 xa[BURST_XV], xb[BURST_XV], xw[BURST_XV], xm[PRO == PRO_MIXNORM ? BURST_XV : 1];
  const bool single = Mt == 1 && vecs_row <= BURST_XV * nthreads;
  float mw0 = 0.f, mw1 = 0.f;
  if constexpr (PRO == PRO_MIXNORM) {   // wave-uniform: scalar loads
    mw0 = p.mix_w[2 * row0];
    mw1 = p.mix_w[2 * row0 + 1];
  }
  {
    const uint16_t* xr = p.x + (size_t)(PRO == PRO_MIXNORM ? 2 * xrow0 : xrow0) * p.ldx;
    const uint16_t* rr = (NORM && p.res_in) ? p.res_in + (size_t)row0 * p.K : xr;
    const uint16_t* wr = (NORM && p.norm_w) ? p.norm_w : xr;
#pragma unroll
    for (int i = 0; i < BURST_XV; i++) {
      const int v = min((int)threadIdx.x + i * nthreads, vecs_row - 1);
      xa[i] = *reinterpret_cast<const u32x4*>(xr + 8 * v);
      if constexpr (PRO == PRO_MIXNORM) xm[i] = *reinterpret_cast<const u32x4*>(xr + p.ldx + 8 * v);
      if constexpr (PRO == PRO_SILU) xb[i] = *reinterpret_cast<const u32x4*>(xr + p.K + 8 * v);
      if constexpr (NORM) {
        xb[i] = *reinterpret_cast<const u32x4*>(rr + 8 * v);
        xw[i] = *reinterpret_cast<const u32x4*>(wr + 8 * v);
      }
    }
  }
  // strip scales: groups x (SW / 2) dwords, up to SR per thread ahead of the weights
  constexpr int SD = SW / 2;
  constexpr int SR = 4;
  const int sdw = groups * SD;
  uint32_t s_reg[SR], z_reg[SR];
#pragma unroll
  for (int i = 0; i < SR; i++) {
    const int d = min((int)threadIdx.x + i * nthreads, sdw - 1);   // strip-major: the strip's groups x SW scales are contiguous
    const size_t off = (size_t)strip * sdw * 2 + 2 * (size_t)d;
    s_reg[i] = *reinterpret_cast<const uint32_t*>(scales_e + off);
    if constexpr (HAS_ZC) z_reg[i] = *reinterpret_cast<const uint32_t*>(zc_e + off);
    else z_reg[i] = 0u;
  }

  // The strip's weights are read through a buffer descriptor that spans exactly the strip: lane offset in a
  // vector register, plane offset in a scalar one, non-temporal.  A request past the strip returns zeros and
  // moves no data, which lets the refill of the ring below be UNCONDITIONAL: a refill under `if (more bursts)`
  // makes every wait behind it count as if the refill had not been issued, i.e. the next chunk's first plane
  // then waits for one plane of the refill as well -- a whole memory round trip in the middle of the stream.
  u32x4 w[CPW][NS];
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int cfirst_s = wave_s * cpw_total;
  const __amdgpu_buffer_rsrc_t wdesc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(wstrip), 0, p.chunks * NS * 1024, 0x00020000);
  const int wlane = lane * 16;
  constexpr int W_NT = 2;                  // cache policy bits of the load: non-temporal
  auto issue_chunk = [&](int burst, int i, int voff) {
    const int c = cfirst_s + burst * CPW + i;
#pragma unroll
    for (int pl = 0; pl < NS; pl++) w[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(wdesc, voff, (c * NS + pl) * 1024, W_NT);
  };
  // Metered start.  Requesting the whole ring (up to 112 KiB per CU, 28 MB over the chip) the moment the
  // kernel starts floods the memory system's queues: every later request of the chip -- the prologue
  // reads above of workgroups dispatched a little later (dispatch skew is 0.3-0.7 us), the scales --
  // then waits behind megabytes of weights and arrives microseconds late, which holds back consumption
  // everywhere.  So: a short pause that lets the prologue reads of the late starters get out first,
  // then one chunk per wave (NS KiB), and the rest of the ring after the barrier that ends the prologue:
  // a wave that asks for more than the CU's miss queue holds stalls AT THE ISSUE, and with the late
  // requests in front of that barrier the whole workgroup waited ~1 us for its slowest issuer before
  // consuming data that had long arrived (gate_up 14.7 -> 14.2 us).  Measured on Llama-3-8B decode
  // (tok/s): all at once 568, one chunk first 593, with the pause of 12 x 64 clocks 620, longer pauses
  // (20, 30) 615 and 603, a wait for the prologue inputs instead of the pause 583.  The macros are
  // build-time knobs for such sweeps (MI355Q_BURST_LATE: 0 = rest right after the first planes' wait,
  // 1 = in front of the barrier, 2 = behind it).
#ifndef MI355Q_BURST_FP
#define MI355Q_BURST_FP -1
#endif
#ifndef MI355Q_BURST_LATE
#define MI355Q_BURST_LATE 2
#endif
#ifndef MI355Q_BURST_DELAY
#define MI355Q_BURST_DELAY 12
#endif
#if MI355Q_BURST_DELAY > 0
  __builtin_amdgcn_s_sleep(MI355Q_BURST_DELAY);
#endif
  constexpr int RING = CPW * NS;                                         // planes per wave
  constexpr int FPW = NS * CPW >= 7 ? 4 : 2;                             // 8 x 4 or 16 x 2 KiB per CU
#ifndef MI355Q_BURST_FP_SMALL
#define MI355Q_BURST_FP_SMALL -1
#endif
  constexpr int FPB = MI355Q_BURST_FP == 0 ? RING : (MI355Q_BURST_FP == 1 ? FPW : (MI355Q_BURST_FP < 0 ? NS : MI355Q_BURST_FP));
  // 32-wide strips on 8 waves have 32 KiB per CU in their whole ring (4 KiB per wave): asked for at once, qkv_proj
  // 5.9 -> 5.6 us; the 16-wide strips of o_proj, 16 waves, lose with the same change (4.35 -> 4.65)
  constexpr int FPS = MI355Q_BURST_FP_SMALL >= 0 ? MI355Q_BURST_FP_SMALL : (NS == 2 ? 0 : -1);
  constexpr int FPQ = (NS * CPW < 7 && FPS >= 0) ? (FPS == 0 ? RING : FPS) : FPB;
  constexpr int FP = FPQ < RING ? FPQ : RING;
  auto issue_plane = [&](int f) {   // plane f of the first ring fill
    const int i = f / NS, pl = f % NS;
    w[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(wdesc, wlane, ((cfirst_s + i) * NS + pl) * 1024, W_NT);
  };
#pragma unroll
  for (int f = 0; f < FP; f++) issue_plane(f);
#if MI355Q_BURST_LATE == 0
  if constexpr (FP < RING) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FP) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = FP; f < RING; f++) issue_plane(f);
    __builtin_amdgcn_sched_barrier(0);
  }
#endif

  // ---- scales -> LDS (strips with more than SR dwords per thread loop here, behind the burst)
#pragma unroll
  for (int i = 0; i < SR; i++) {
    const int d = threadIdx.x + i * nthreads;
    if (d < sdw) {
      reinterpret_cast<uint32_t*>(sc_l)[d] = s_reg[i];
      if constexpr (HAS_ZC) reinterpret_cast<uint32_t*>(zc_l)[d] = z_reg[i];
    }
  }
  for (int d = threadIdx.x + SR * nthreads; d < sdw; d += nthreads) {
    const size_t off = (size_t)strip * sdw * 2 + 2 * (size_t)d;
    reinterpret_cast<uint32_t*>(sc_l)[d] = *reinterpret_cast<const uint32_t*>(scales_e + off);
    if constexpr (HAS_ZC) reinterpret_cast<uint32_t*>(zc_l)[d] = *reinterpret_cast<const uint32_t*>(zc_e + off);
  }

  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
  // ---- prologue: activations of this pass -> LDS in the activation dtype
  auto lo = [](uint32_t u) { return DT::to_f32((uint16_t)(u & 0xffff)); };
  auto hi = [](uint32_t u) { return DT::to_f32((uint16_t)(u >> 16)); };
  if (single) {
    if constexpr (NORM) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < BURST_XV; i++) {
        const int v = threadIdx.x + i * nthreads;
        if (v < vecs_row) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if constexpr (PRO == PRO_MIXNORM) {   // the routed layer's combine, as moe_sum_kernel does it
              float m0 = __builtin_fmaf(mw1, lo(xm[i][e]), __builtin_fmaf(mw0, lo(xa[i][e]), 0.f));
              float m1 = __builtin_fmaf(mw1, hi(xm[i][e]), __builtin_fmaf(mw0, hi(xa[i][e]), 0.f));
              // the fp32 sums are rounded to fp32 first, as the stand-alone kernel's are: without this the compiler
              // folds the last multiply-add and the conversion into one instruction with a single rounding (fp16)
              asm volatile("" : "+v"(m0), "+v"(m1));
              xa[i][e] = pack2<DT>(m0, m1);
            }
            float a = lo(xa[i][e]), c = hi(xa[i][e]);
            if (p.res_in) {
              a = DT::to_f32(DT::from_f32(a + lo(xb[i][e])));
              c = DT::to_f32(DT::from_f32(c + hi(xb[i][e])));
            }
            xa[i][e] = pack2<DT>(a, c);  // z, rounded
            ss += a * a + c * c;
          }
          if (strip == 0 && p.res_out) *reinterpret_cast<u32x4*>(p.res_out + (size_t)row0 * p.K + 8 * v) = xa[i];
        }
      }
      ss = wave_sum_dpp(ss);
      if (lane == 0) red[wave] = ss;
      if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
      __syncthreads();
      if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
      float t = 0.f;
      for (int i = 0; i < nw; i++) t += red[i];
      const float rs = rsqrtf(t / (float)p.K + p.eps);
#pragma unroll
      for (int i = 0; i < BURST_XV; i++) {
        const int v = threadIdx.x + i * nthreads;
        if (v < vecs_row) {
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float a = lo(xa[i][e]) * rs, c = hi(xa[i][e]) * rs;
            if (p.norm_w) { a *= lo(xw[i][e]); c *= hi(xw[i][e]); }
            o[e] = pack2<DT>(a, c);
          }
          *reinterpret_cast<u32x4*>(act + 8 * v) = o;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < BURST_XV; i++) {
        const int v = threadIdx.x + i * nthreads;
        if (v < vecs_row) {
          u32x4 o = xa[i];
          if constexpr (PRO == PRO_SILU) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
              // silu rounded to the activation dtype before the multiply, as the reference op does
              const float s0 = DT::to_f32(DT::from_f32(silu_f(lo(xa[i][e])))), s1 = DT::to_f32(DT::from_f32(silu_f(hi(xa[i][e]))));
              o[e] = pack2<DT>(s0 * lo(xb[i][e]), s1 * hi(xb[i][e]));
            }
          }
          *reinterpret_cast<u32x4*>(act + 8 * v) = o;
        }
      }
    }
  } else {
    // several rows (or a very long row): straightforward loops; their loads queue behind the burst
    for (int mb = 0; mb < Mt; mb += 4) {
      const int nm = min(4, Mt - mb);
      float rs[4] = {1.f, 1.f, 1.f, 1.f};
      if constexpr (NORM) {   // (PRO_MIXNORM never gets here: the host admits it for single rows only)
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
        for (int v = threadIdx.x; v < vecs_row; v += nthreads) {
#pragma unroll
          for (int m = 0; m < 4; m++) {
            if (m < nm) {
              const size_t off = (size_t)(row0 + mb + m);
              const u32x4 xv = *reinterpret_cast<const u32x4*>(p.x + (size_t)(xrow0 + mb + m) * p.ldx + 8 * v);
              u32x4 rv = u32x4{0u, 0u, 0u, 0u};
              if (p.res_in) rv = *reinterpret_cast<const u32x4*>(p.res_in + off * p.K + 8 * v);
#pragma unroll
              for (int e = 0; e < 4; e++) {
                float a = lo(xv[e]), c = hi(xv[e]);
                if (p.res_in) {
                  a = DT::to_f32(DT::from_f32(a + lo(rv[e])));
                  c = DT::to_f32(DT::from_f32(c + hi(rv[e])));
                }
                ss[m] += a * a + c * c;
              }
            }
          }
        }
#pragma unroll
        for (int m = 0; m < 4; m++) ss[m] = wave_sum(ss[m]);
        __syncthreads();
        if (lane == 0) {
#pragma unroll
          for (int m = 0; m < 4; m++) red[wave * 4 + m] = ss[m];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; m++) {
          float t = 0.f;
          for (int i = 0; i < nw; i++) t += red[i * 4 + m];
          rs[m] = rsqrtf(t / (float)p.K + p.eps);
        }
      }
      for (int idx = threadIdx.x; idx < nm * vecs_row; idx += nthreads) {
        const int m = idx / vecs_row, v = idx - m * vecs_row;
        const size_t off = (size_t)(row0 + mb + m);
        const uint16_t* src = p.x + (size_t)(xrow0 + mb + m) * p.ldx + 8 * v;
        const u32x4 xv = *reinterpret_cast<const u32x4*>(src);
        u32x4 o = xv;
        if constexpr (PRO == PRO_SILU) {
          const u32x4 u = *reinterpret_cast<const u32x4*>(src + p.K);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float s0 = DT::to_f32(DT::from_f32(silu_f(lo(xv[e])))), s1 = DT::to_f32(DT::from_f32(silu_f(hi(xv[e]))));
            o[e] = pack2<DT>(s0 * lo(u[e]), s1 * hi(u[e]));
          }
        }
        if constexpr (NORM) {
          u32x4 rv = u32x4{0u, 0u, 0u, 0u}, wv = u32x4{0u, 0u, 0u, 0u}, zo;
          if (p.res_in) rv = *reinterpret_cast<const u32x4*>(p.res_in + off * p.K + 8 * v);
          if (p.norm_w) wv = *reinterpret_cast<const u32x4*>(p.norm_w + 8 * v);
          const float r_m = m == 0 ? rs[0] : (m == 1 ? rs[1] : (m == 2 ? rs[2] : rs[3]));
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float a = lo(xv[e]), c = hi(xv[e]);
            if (p.res_in) {
              a = DT::to_f32(DT::from_f32(a + lo(rv[e])));
              c = DT::to_f32(DT::from_f32(c + hi(rv[e])));
            }
            zo[e] = pack2<DT>(a, c);
            a *= r_m;
            c *= r_m;
            if (p.norm_w) { a *= lo(wv[e]); c *= hi(wv[e]); }
            o[e] = pack2<DT>(a, c);
          }
          if (strip == 0 && p.res_out) *reinterpret_cast<u32x4*>(p.res_out + off * p.K + 8 * v) = zo;
        }
        *reinterpret_cast<u32x4*>(act + (size_t)(mb + m) * astride + 8 * v) = o;
      }
      if constexpr (NORM) __syncthreads();  // red is reused by the next row group
    }
  }
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
#if MI355Q_BURST_LATE == 1
#pragma unroll
  for (int f = FP; f < RING; f++) issue_plane(f);
#endif
  __syncthreads();
#if MI355Q_BURST_LATE == 2
#pragma unroll
  for (int f = FP; f < RING; f++) issue_plane(f);
#endif
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();

  // ---- consume the weight planes in arrival order
  f32x4 tot[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) tot[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = u32x4{DT::ones, DT::ones, DT::ones, DT::ones};
  const uint16_t* arow = act + (size_t)min(ml, Mt - 1) * astride + 8 * kq;
#ifndef MI355Q_BURST_ANDOR
#define MI355Q_BURST_ANDOR 1
#endif
#ifndef MI355Q_BURST_PF_BF
#define MI355Q_BURST_PF_BF 1
#endif
#ifndef MI355Q_BURST_PF_SC
#define MI355Q_BURST_PF_SC 1
#endif
  // (x & mask) | magic as ONE instruction (v_and_or_b32) per pair of weights: that form takes no literal
  // operands, so both constants are put in registers behind the compiler's back; left to itself it emits
  // v_and + v_or with literals, 11 instead of 7 instructions per packed dword
  uint32_t magic_v = DT::int4_magic, nib_mask = 0x000F000Fu;
#if MI355Q_BURST_ANDOR
  asm volatile("" : "+v"(magic_v), "+s"(nib_mask));
#endif
  auto expand = [&](uint32_t wd) {
    u32x4 r;
    r[0] = (wd & nib_mask) | magic_v;
    r[1] = ((wd >> 4) & nib_mask) | magic_v;
    r[2] = ((wd >> 8) & nib_mask) | magic_v;
    r[3] = ((wd >> 12) & nib_mask) | magic_v;
    return r;
  };
  // What a chunk takes from LDS -- the activation fragments of its four k-steps, the scales of its group -- is
  // requested before the first product: behind the scheduling fences below each read would otherwise be issued
  // where its value is needed and the wave would stand through the LDS latency five times per chunk.  The
  // widest variant with zero points has no registers left for that and keeps the reads in place.
  constexpr bool ROOM = !(HAS_ZC && NS * CPW >= 14);
  constexpr bool PF_BF = MI355Q_BURST_PF_BF && ROOM;
  constexpr bool PF_SC = MI355Q_BURST_PF_SC && ROOM && !SMALLG;

  for (int burst = 0; burst < p.bursts; burst++) {
    // lane offset of this burst's refills: past the strip once there is nothing left to ask for
    const int refill_off = wlane + (burst + 1 < p.bursts ? 0 : 0x40000000);
#pragma unroll
    for (int i = 0; i < CPW; i++) {
      const int c = cfirst + burst * CPW + i;
      f32x4 acc[NS], acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; s++) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x4 bfr[4];
      u32x2 svr[NS], zvr[NS];
      if constexpr (PF_BF) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) bfr[ks] = *reinterpret_cast<const u32x4*>(arow + (c * 4 + ks) * 32);
      }
      if constexpr (PF_SC) {
        const int g = (c * 4 + 3) / p.kpg;
        const uint16_t* sg = sc_l + (size_t)g * SW + QW * kq;
        const uint16_t* zg = zc_l + (size_t)g * SW + QW * kq;
#pragma unroll
        for (int sub = 0; sub < NS; sub++) {
          svr[sub] = *reinterpret_cast<const u32x2*>(sg + 4 * sub);
          if constexpr (HAS_ZC) zvr[sub] = *reinterpret_cast<const u32x2*>(zg + 4 * sub);
        }
      }
      // fold one scale group's sums into the totals: this lane's outputs are channels
      // QW kq + 4 sub + r of the strip
      auto flush = [&](int g) {
        const uint16_t* sg = sc_l + (size_t)g * SW + QW * kq;
        const uint16_t* zg = zc_l + (size_t)g * SW + QW * kq;
#pragma unroll
        for (int sub = 0; sub < NS; sub++) {
          u32x2 sv, zv = u32x2{0u, 0u};
          if constexpr (PF_SC) {
            sv = svr[sub];
            if constexpr (HAS_ZC) zv = zvr[sub];
          } else {
            sv = *reinterpret_cast<const u32x2*>(sg + 4 * sub);
            if constexpr (HAS_ZC) zv = *reinterpret_cast<const u32x2*>(zg + 4 * sub);
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const uint32_t sw_ = sv[r >> 1];
            const float s = (r & 1) ? hi(sw_) : lo(sw_);
            float z;
            if constexpr (HAS_ZC) z = (r & 1) ? hi(zv[r >> 1]) : lo(zv[r >> 1]);
            else z = DT::int4_bias + 8.0f;
            const float d = __builtin_fmaf(-z, acc1[r], acc[sub][r]);
            tot[sub][r] = __builtin_fmaf(s, d, tot[sub][r]);
          }
        }
      };
#ifdef MI355Q_BURST_NOCOMPUTE   // diagnostics: the weight stream alone (results are wrong)
#pragma unroll
      for (int pl = 0; pl < NS; pl++) acc1[0] += __uint_as_float((w[i][pl][0] ^ w[i][pl][1] ^ w[i][pl][2] ^ w[i][pl][3]) & 0x3fffffffu);
      constexpr int KSTEPS = 0;
#else
      constexpr int KSTEPS = 4;
#endif
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        u32x4 bf;
        if constexpr (PF_BF) bf = bfr[ks];
        else bf = *reinterpret_cast<const u32x4*>(arow + (c * 4 + ks) * 32);
#ifndef MI355Q_DIAG_NOONES
        acc1 = mfma16<DT>(ones, bf, acc1);
#endif
#pragma unroll
        for (int sub = 0; sub < NS; sub++) {
          const int e = ks * NS + sub;
#ifdef MI355Q_DIAG_NODQ      // diagnostics: the products without the 4-bit expansion (results are wrong)
          const uint32_t raw = w[i][e >> 2][e & 3];
          acc[sub] = mfma16<DT>(u32x4{raw, raw, raw, raw}, bf, acc[sub]);
#else
          acc[sub] = mfma16<DT>(expand(w[i][e >> 2][e & 3]), bf, acc[sub]);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);  // dequantised fragments of one k-step live at a time
        if constexpr (SMALLG) {
          if (((ks + 1) % p.kpg) == 0) {   // kpg is 1 or 2 here: a group ends inside the chunk
            flush((c * 4 + ks) / p.kpg);
            acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; s++) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#ifdef MI355Q_DIAG_NOFLUSH   // diagnostics: no scale step (results are wrong)
#pragma unroll
      for (int sub = 0; sub < NS; sub++) tot[sub] += acc[sub];
#else
      if constexpr (!SMALLG) flush((c * 4 + 3) / p.kpg);
#endif
      // the slot just consumed is refilled at once: CPW chunks stay in flight per wave to the end
      issue_chunk(burst + 1, i, refill_off);
      // keep the chunks' MFMA chains apart: interleaving them would hold several accumulator sets live
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- cross-wave reduction through LDS: lanes of real rows publish their 4 NS sums (channels
  // QW kq + 4 sub + r) as float4s; every (row, 4 channels) item is then finished by one thread
  if (ml < Mt) {
    float* mine = red + ((size_t)wave * p.rows_pp + ml) * SW + QW * kq;
#pragma unroll
    for (int sub = 0; sub < NS; sub++) *reinterpret_cast<f32x4*>(mine + 4 * sub) = tot[sub];
  }
  __syncthreads();
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime();
  for (int it = threadIdx.x; it < Mt * QW; it += nthreads) {
    const int m_l = it / QW, nq = it % QW;
    const int nl = 4 * nq;
    f32x4 acc4 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nw; i++) acc4 += *reinterpret_cast<const f32x4*>(red + ((size_t)i * p.rows_pp + m_l) * SW + nl);
    float v[4] = {acc4[0], acc4[1], acc4[2], acc4[3]};
    const int n = strip * SW + nl;
    if (p.bias) {  // bias is added to the rounded GEMM result, as output.add_(bias) does
      const u32x2 bb = *reinterpret_cast<const u32x2*>(p.bias + n);
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = DT::to_f32(DT::from_f32(v[j]));
      v[0] += lo(bb[0]); v[1] += hi(bb[0]); v[2] += lo(bb[1]); v[3] += hi(bb[1]);
    }
    if (p.epi_silu) {
      // SiluAndMul on the rounded projection outputs: round(round(silu(g)) * u), as the separate op does
      const float g0 = DT::to_f32(DT::from_f32(v[0])), u0 = DT::to_f32(DT::from_f32(v[1]));
      const float g1 = DT::to_f32(DT::from_f32(v[2])), u1 = DT::to_f32(DT::from_f32(v[3]));
      const float s0 = DT::to_f32(DT::from_f32(silu_f(g0))), s1 = DT::to_f32(DT::from_f32(silu_f(g1)));
      *reinterpret_cast<uint32_t*>(p.out + (size_t)(row0 + m_l) * (p.N >> 1) + (n >> 1)) = pack2<DT>(s0 * u0, s1 * u1);
    } else {
      u32x2 o;
      o[0] = pack2<DT>(v[0], v[1]);
      o[1] = pack2<DT>(v[2], v[3]);
      *reinterpret_cast<u32x2*>(p.out + (size_t)(row0 + m_l) * p.N + n) = o;
    }
  }
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
}

// (NS, CPW) pairs the burst kernel is built for
static bool burst_built(int ns, int cpw) {
  switch (ns) {
    case 1: return cpw == 2 || cpw == 4 || cpw == 7;
    case 2: return cpw == 1 || cpw == 2 || cpw == 4;
    case 4: return cpw == 2 || cpw == 4;
    case 7: return cpw == 1 || cpw == 2;
    default: return false;
  }
}

struct BurstPlan {
  bool ok;
  int nw, cpw, bursts, rows_pp;
  size_t lds;
};

// Waves and burst depth for a strip of NS subtiles over K: the most waves (<= 16, <= 8 for the
// wide strips whose register budget needs 512-thread workgroups) that divide the chunk count
// with a per-wave share the kernel is built for; rows per pass from what is left of the LDS.
static BurstPlan plan_burst(int M, int K, int G, int ns, bool has_zc) {
  BurstPlan bp{false, 0, 0, 0, 0, 0};
  if ((G % 128 && G != 64 && G != 32) || K % 128) return bp;
  const int chunks = K / 128, groups = K / G;
  // experiments: MI355Q_BURST_NW / MI355Q_BURST_CPW force the waves per workgroup and the ring depth of
  // strips of MI355Q_BURST_NS subtiles (0: every width) where the shape allows them
  const int f_nw = env_int("MI355Q_BURST_NW", 0), f_cpw = env_int("MI355Q_BURST_CPW", 0), f_ns = env_int("MI355Q_BURST_NS", 0);
  if (f_nw > 0 && f_cpw > 0 && (f_ns == 0 || f_ns == ns) && chunks % (f_nw * f_cpw) == 0 && burst_built(ns, f_cpw) &&
      f_nw * 64 <= burst_threads(ns, f_cpw)) {
    bp.ok = true; bp.nw = f_nw; bp.cpw = f_cpw; bp.bursts = chunks / (f_nw * f_cpw);
  }
  // 32-wide strips run fastest on 8 waves with two ring fills each (Llama-3-8B qkv_proj with the add + RMSNorm
  // prologue, us: 16 waves x 2 chunks 6.6, 8 x 2 x 2 fills 6.1, 8 x 4 6.3, 4 x 4 x 2 6.5, 16 x 1 x 2 6.8); the other
  // widths measured fastest with the most waves their ring depth allows
  const int nw_max = (ns == 2 && chunks % 16 == 0 && !env_flag("MI355Q_BURST_WIDE_NS2")) ? 8 : 16;
  for (int nw = nw_max; nw >= 2 && !bp.ok; nw--) {
    if (chunks % nw) continue;
    const int per = chunks / nw;
    if (nw_max == 8 && nw == 8 && per % 2 == 0 && per >= 4) {   // two chunks in flight, refilled as they retire
      bp.ok = true; bp.nw = 8; bp.cpw = 2; bp.bursts = per / 2;
      break;
    }
    for (int cpw = 8; cpw >= 2; cpw--) {
      if (per % cpw || !burst_built(ns, cpw)) continue;
      if (nw * 64 > burst_threads(ns, cpw)) continue;   // deep rings run on 512-thread workgroups
      bp.ok = true; bp.nw = nw; bp.cpw = cpw; bp.bursts = per / cpw;
      break;
    }
  }
  if (!bp.ok) return bp;
  // LDS: reduction scratch [nw][rows][SW] fp32, strip scales (and zero points), staged rows
  const size_t fixed = (size_t)groups * ns * 16 * 2 * (has_zc ? 2 : 1);
  const size_t row = (size_t)(K + 8) * 2 + (size_t)bp.nw * ns * 16 * 4;
  if (fixed + row > 160 * 1024) { bp.ok = false; return bp; }
  bp.rows_pp = (int)std::min<size_t>(16, (160 * 1024 - fixed) / row);
  bp.rows_pp = std::min(bp.rows_pp, M);
  bp.lds = fixed + (size_t)bp.rows_pp * row;
  return bp;
}

template <typename DT, int NS, int CPW, int PRO>
static int launch_burst2(const BurstParams& p, size_t lds, hipStream_t stream) {
  dim3 grid(p.N / (16 * NS), 1, (p.M + p.rows_pp - 1) / p.rows_pp), block(p.nw * 64);
  if (p.kpg < 4) {   // 32- or 64-wide scale groups (built without the SiLU prologue)
    if constexpr (PRO != PRO_SILU) {
      if (p.zc)
        hipLaunchKernelGGL((w4a16_burst_kernel<DT, NS, CPW, true, PRO, true>), grid, block, lds, stream, p);
      else
        hipLaunchKernelGGL((w4a16_burst_kernel<DT, NS, CPW, false, PRO, true>), grid, block, lds, stream, p);
      return check_launch("w4a16_burst_kernel");
    } else {
      return set_error(MI355Q_EINVAL, "w4a16_gemm: the SiLU prologue needs a group size that is a multiple of 128");
    }
  }
  if (p.zc)
    hipLaunchKernelGGL((w4a16_burst_kernel<DT, NS, CPW, true, PRO>), grid, block, lds, stream, p);
  else
    hipLaunchKernelGGL((w4a16_burst_kernel<DT, NS, CPW, false, PRO>), grid, block, lds, stream, p);
  return check_launch("w4a16_burst_kernel");
}

template <typename DT, int PRO>
static int launch_burst(const BurstParams& p, int ns, int cpw, size_t lds, hipStream_t stream) {
#define MI355Q_B(n, c) if (ns == n && cpw == c) return launch_burst2<DT, n, c, PRO>(p, lds, stream)
  MI355Q_B(1, 2); MI355Q_B(1, 4); MI355Q_B(1, 7);
  MI355Q_B(2, 1); MI355Q_B(2, 2); MI355Q_B(2, 4);
  MI355Q_B(4, 2); MI355Q_B(4, 4);
  MI355Q_B(7, 1); MI355Q_B(7, 2);
#undef MI355Q_B
  return set_error(MI355Q_EINVAL, "w4a16_gemm: no burst kernel for strip %d, depth %d", 16 * ns, cpw);
}

// The ticket counters occupy a fixed block at the start of the workspace, so that partial
// tiles of one GEMM shape can never land on the counters of another.
constexpr size_t COUNTER_BYTES = 64 * 1024;
static inline bool counters_fit(int M, int N, int sw) {
  return (size_t)(N / sw) * ((M + 15) / 16) * sizeof(int) <= COUNTER_BYTES;
}

// LDS bytes one chunk of staged activations takes (LDS prologues)
static inline size_t act_bytes_per_chunk(int CS, int M) { return (size_t)CS * 32 * std::min(M, 16) * 2; }
constexpr size_t ACT_LDS_LIMIT = 96 * 1024;

struct Plan {
  int splits, nw;
};

// One workgroup per strip splits K over its waves; a cross-block split is added only when
// the strips alone leave most CUs idle and K is long enough to share.
static Plan make_plan(int M, int N, int K, int G, int sw, bool lds_prologue, int split_k) {
  const int CS = chunk_steps(G);
  const int chunks = K / (32 * CS);
  const int strips = N / sw, mts = (M + 15) / 16;
  int s = 1;
  const char* env = env_value("MI355Q_SPLITK");
  if (split_k > 0) s = std::min(split_k, chunks);
  else if (env && atoi(env) > 0) s = std::min(atoi(env), chunks);
  else {
    while ((long)strips * mts * s < 160 && chunks / (s * 2) >= 16) s *= 2;
  }
  if (lds_prologue)
    while (s < chunks && (size_t)((chunks + s - 1) / s) * act_bytes_per_chunk(CS, M) > ACT_LDS_LIMIT) s++;
  if (!counters_fit(M, N, sw)) s = 1;
  const int per_block = (chunks + s - 1) / s;
  // 16 waves only pay off when every wave still gets several chunks of a long K
  int nw = (per_block >= 64 && sw != 64) ? 16 : 8;
  const char* envw = env_value("MI355Q_WAVES");
  if (envw && atoi(envw) > 0) nw = std::min(sw == 64 ? 8 : 16, atoi(envw));
  while (nw > 4 && nw / 2 >= per_block) nw /= 2;
  return Plan{s, nw};
}

template <typename DT, int SW, int PRO>
static int launch_sw(const SkinnyParams& p, int CS, hipStream_t stream, size_t lds) {
  dim3 grid(p.N / SW, p.splits, (p.M + p.mrows - 1) / p.mrows), block(p.nw * 64);
  const bool zc = p.zc != nullptr;
#define MI355Q_LAUNCH(cs, z) \
  hipLaunchKernelGGL((w4a16_skinny_kernel<DT, SW, cs, z, PRO>), grid, block, lds, stream, p)
  constexpr int KT = 4 / (SW / 16);
  if (CS == 4) { if (zc) MI355Q_LAUNCH(4, true); else MI355Q_LAUNCH(4, false); }
  else if constexpr (PRO == PRO_NONE && KT <= 2) {
    if (CS == 2) { if (zc) MI355Q_LAUNCH(2, true); else MI355Q_LAUNCH(2, false); }
    else if constexpr (KT == 1) { if (zc) MI355Q_LAUNCH(1, true); else MI355Q_LAUNCH(1, false); }
    else return set_error(MI355Q_EINVAL, "w4a16_gemm: group size needs a wider tile");
  } else {
    return set_error(MI355Q_EINVAL, "w4a16_gemm: fused prologues need a group size that is a multiple of 128");
  }
#undef MI355Q_LAUNCH
  return check_launch("w4a16_skinny_kernel");
}

template <typename DT, int PRO>
static int launch_skinny(const SkinnyParams& p, int sw, int CS, hipStream_t stream, size_t lds) {
  if (sw == 64) return launch_sw<DT, 64, PRO>(p, CS, stream, lds);
  if (sw == 32) return launch_sw<DT, 32, PRO>(p, CS, stream, lds);
  return launch_sw<DT, 16, PRO>(p, CS, stream, lds);
}

// ---------------------------------------------------------------------------
// Rows kernel: several activation rows (5 <= M, 16 per pass) against a 16-channel-strip layout.
//
// With M rows a workgroup must take in M K 2 bytes of activations next to its weights, and a CU
// ingests ~10 B/cycle whatever the source.  One workgroup per 16-channel strip over the whole
// of K (the single-row shape) would make every CU read the entire activation matrix, several
// times its weights.  Here a workgroup owns SPB adjacent strips (their runs are read side by side:
// same layout, no repack) and 1/splits of K, so both its weights and its activation slice are
// K N / 2 / 256-ish bytes; the K-slices meet through fp32 partial tiles in the workspace, and the
// workgroup that draws the last arrival ticket of a tile sums them in slice order (deterministic).
// ---------------------------------------------------------------------------
struct RowsParams {
  const uint16_t* x;
  int ldx;
  const u32x4* wp;
  const uint16_t* scales;
  const uint16_t* zc;
  const uint16_t* bias;
  uint16_t* out;
  float* partial;   // [splits][M][N]
  int* counters;    // [row passes][N / (16 SPB)], zero between launches
  int M, N, K;
  int chunks;       // K / 128
  int gchunks;      // chunks per scale group
  int cps;          // chunks per K-slice = nw * CPW * bursts
  int splits, nw, bursts, rows_pp;
  int partial_only;  // 1: leave the fp32 K-slice tiles in `partial` for a consumer kernel that sums them
};

template <typename DT, int SPB, int CPW, bool HAS_ZC>
__global__ __launch_bounds__(512) void w4a16_rows_kernel(const RowsParams p) {
  constexpr int BW = 16 * SPB, QB = BW / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nw = p.nw, nthreads = nw * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vstrip = blockIdx.x, sp = blockIdx.y;
  const int kq = lane >> 4, ml = lane & 15;
  const int row0 = blockIdx.z * p.rows_pp;
  const int Mt = min(p.rows_pp, p.M - row0);
  const int kc0 = sp * p.cps;
  const int Ks = p.cps * 128, astride = Ks + 8;
  const int groups = p.chunks / p.gchunks;
  const int g_first = kc0 / p.gchunks;
  const int gps = (kc0 + p.cps - 1) / p.gchunks - g_first + 1;

  float* red = reinterpret_cast<float*>(smem);                                   // [nw][rows_pp][BW]
  uint16_t* sc_l = reinterpret_cast<uint16_t*>(smem + (size_t)nw * p.rows_pp * BW * 4);  // [SPB][gps][16]
  uint16_t* zc_l = sc_l + (size_t)SPB * gps * 16;
  int* flag = reinterpret_cast<int*>(zc_l + (HAS_ZC ? (size_t)SPB * gps * 16 : 0));
  uint16_t* act = reinterpret_cast<uint16_t*>(flag + 4);                         // [rows_pp][astride]

  // ---- this wave's weight planes first (for most shapes: everything it will consume) ...
  const int per_wave = CPW * p.bursts;
  const int cfirst = kc0 + wave * per_wave;
  const u32x4* wbase = p.wp + (size_t)vstrip * SPB * p.chunks * 64 + lane;
  u32x4 w[CPW][SPB];
  auto issue_chunk = [&](int burst, int i) {
    const size_t c = (size_t)(cfirst + burst * CPW + i);
#pragma unroll
    for (int sub = 0; sub < SPB; sub++)
      w[i][sub] = __builtin_nontemporal_load(wbase + ((size_t)sub * p.chunks + c) * 64);
  };
#pragma unroll
  for (int i = 0; i < CPW; i++) issue_chunk(0, i);

  // ---- ... then the slice's scales and activation rows -> LDS
  for (int d = threadIdx.x; d < SPB * gps * 8; d += nthreads) {
    const int sub = d / (gps * 8), r = d - sub * gps * 8;     // r: dword within the strip's slice run
    const size_t off = ((size_t)(vstrip * SPB + sub) * groups + g_first) * 16 + 2 * (size_t)r;
    reinterpret_cast<uint32_t*>(sc_l)[d] = *reinterpret_cast<const uint32_t*>(p.scales + off);
    if constexpr (HAS_ZC) reinterpret_cast<uint32_t*>(zc_l)[d] = *reinterpret_cast<const uint32_t*>(p.zc + off);
  }
  const int vecs = Ks >> 3, total = Mt * vecs;
  for (int base = threadIdx.x; base < total; base += 4 * nthreads) {
    u32x4 t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = base + j * nthreads;
      if (idx < total) {
        const int m = idx / vecs, v = idx - m * vecs;
        t[j] = *reinterpret_cast<const u32x4*>(p.x + (size_t)(row0 + m) * p.ldx + (size_t)kc0 * 128 + 8 * v);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = base + j * nthreads;
      if (idx < total) {
        const int m = idx / vecs, v = idx - m * vecs;
        *reinterpret_cast<u32x4*>(act + (size_t)m * astride + 8 * v) = t[j];
      }
    }
  }
  __syncthreads();

  auto lo = [](uint32_t u) { return DT::to_f32((uint16_t)(u & 0xffff)); };
  auto hi = [](uint32_t u) { return DT::to_f32((uint16_t)(u >> 16)); };
  f32x4 tot[SPB];
#pragma unroll
  for (int i = 0; i < SPB; i++) tot[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = u32x4{DT::ones, DT::ones, DT::ones, DT::ones};
  // (x & mask) | magic in one instruction per 16-bit pair: both constants live in registers (see the ring kernel)
  uint32_t magic_v = DT::int4_magic, nib_mask = 0x000F000Fu;
  asm volatile("" : "+v"(magic_v), "+s"(nib_mask));
  auto expand = [&](uint32_t wd) {
    u32x4 r;
    r[0] = (wd & nib_mask) | magic_v;
    r[1] = ((wd >> 4) & nib_mask) | magic_v;
    r[2] = ((wd >> 8) & nib_mask) | magic_v;
    r[3] = ((wd >> 12) & nib_mask) | magic_v;
    return r;
  };
  const uint16_t* arow = act + (size_t)min(ml, Mt - 1) * astride + 8 * kq;

  for (int burst = 0; burst < p.bursts; burst++) {
#pragma unroll
    for (int i = 0; i < CPW; i++) {
      const int c = cfirst + burst * CPW + i;
      f32x4 acc[SPB], acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < SPB; s++) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const u32x4 bf = *reinterpret_cast<const u32x4*>(arow + ((c - kc0) * 4 + ks) * 32);
        acc1 = mfma16<DT>(ones, bf, acc1);
#pragma unroll
        for (int sub = 0; sub < SPB; sub++) acc[sub] = mfma16<DT>(expand(w[i][sub][ks]), bf, acc[sub]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // this lane's outputs: channels 16 sub + 4 kq + r of the workgroup's BW
      const int gl = c / p.gchunks - g_first;
#pragma unroll
      for (int sub = 0; sub < SPB; sub++) {
        const u32x2 sv = *reinterpret_cast<const u32x2*>(sc_l + ((size_t)sub * gps + gl) * 16 + 4 * kq);
        u32x2 zv = u32x2{0u, 0u};
        if constexpr (HAS_ZC) zv = *reinterpret_cast<const u32x2*>(zc_l + ((size_t)sub * gps + gl) * 16 + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float s = (r & 1) ? hi(sv[r >> 1]) : lo(sv[r >> 1]);
          float z;
          if constexpr (HAS_ZC) z = (r & 1) ? hi(zv[r >> 1]) : lo(zv[r >> 1]);
          else z = DT::int4_bias + 8.0f;
          const float d = __builtin_fmaf(-z, acc1[r], acc[sub][r]);
          tot[sub][r] = __builtin_fmaf(s, d, tot[sub][r]);
        }
      }
      if (burst + 1 < p.bursts) issue_chunk(burst + 1, i);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- cross-wave reduction through LDS
  if (ml < Mt) {
    float* mine = red + ((size_t)wave * p.rows_pp + ml) * BW + 4 * kq;
#pragma unroll
    for (int sub = 0; sub < SPB; sub++) *reinterpret_cast<f32x4*>(mine + 16 * sub) = tot[sub];
  }
  __syncthreads();

  typedef unsigned long long u64;
  auto block_sum = [&](int m_l, int nl) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nw; i++) a += *reinterpret_cast<const f32x4*>(red + ((size_t)i * p.rows_pp + m_l) * BW + nl);
    return a;
  };
  auto finish = [&](int m, int n, f32x4 a) {
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (p.bias) {  // bias is added to the rounded GEMM result, as output.add_(bias) does
      const u32x2 bb = *reinterpret_cast<const u32x2*>(p.bias + n);
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = DT::to_f32(DT::from_f32(v[j]));
      v[0] += lo(bb[0]); v[1] += hi(bb[0]); v[2] += lo(bb[1]); v[3] += hi(bb[1]);
    }
    u32x2 o;
    o[0] = pack2<DT>(v[0], v[1]);
    o[1] = pack2<DT>(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p.out + (size_t)m * p.N + n) = o;
  };

  if (p.partial_only) {   // the next kernel on the stream merges the slices: plain stores, no ticket
    for (int it = threadIdx.x; it < Mt * QB; it += nthreads) {
      const int m_l = it / QB, nl = 4 * (it % QB);
      *reinterpret_cast<f32x4*>(p.partial + ((size_t)sp * p.M + row0 + m_l) * p.N + vstrip * BW + nl) = block_sum(m_l, nl);
    }
    return;
  }
  if (p.splits == 1) {
    for (int it = threadIdx.x; it < Mt * QB; it += nthreads) {
      const int m_l = it / QB, nl = 4 * (it % QB);
      finish(row0 + m_l, vstrip * BW + nl, block_sum(m_l, nl));
    }
    return;
  }
  // publish this slice's tile with write-through (agent-scope) stores, then draw a ticket
  for (int it = threadIdx.x; it < Mt * QB; it += nthreads) {
    const int m_l = it / QB, nl = 4 * (it % QB);
    const f32x4 a = block_sum(m_l, nl);
    u64* dst = reinterpret_cast<u64*>(p.partial + ((size_t)sp * p.M + row0 + m_l) * p.N + vstrip * BW + nl);
    __hip_atomic_store(dst, (u64)__float_as_uint(a[0]) | ((u64)__float_as_uint(a[1]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, (u64)__float_as_uint(a[2]) | ((u64)__float_as_uint(a[3]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    int* cnt = p.counters + (size_t)blockIdx.z * gridDim.x + vstrip;
    // No release fence in front of the ticket: every published word was written by an agent-scope atomic
    // store (write-through) and waited for (vmcnt(0) + barrier above).  A fence here writes the L2 back in
    // every workgroup: measured -28 % on decode at 32 x 1024 tokens and +36 % on TTFT(2048).
    const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == p.splits - 1;
    if (last) {
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  for (int it = threadIdx.x; it < Mt * QB; it += nthreads) {
    const int m_l = it / QB, nl = 4 * (it % QB);
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splits; s++) {
      const u64* src = reinterpret_cast<const u64*>(p.partial + ((size_t)s * p.M + row0 + m_l) * p.N + vstrip * BW + nl);
      const u64 l = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u64 h = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a[0] += __uint_as_float((uint32_t)l);
      a[1] += __uint_as_float((uint32_t)(l >> 32));
      a[2] += __uint_as_float((uint32_t)h);
      a[3] += __uint_as_float((uint32_t)(h >> 32));
    }
    finish(row0 + m_l, vstrip * BW + nl, a);
  }
}

constexpr int ROWS_SPB = 4;
constexpr int ROWS_MIN_M = 5;

struct RowsPlan {
  bool ok;
  int splits, nw, cpw, bursts, rows_pp;
  size_t lds;
};

// K-slices so that (N / 64) x slices is about one workgroup per CU, each slice an exact
// nw x cpw x bursts of 128-k chunks; rows per pass from what the LDS holds.
static RowsPlan plan_rows(int M, int N, int K, int G, bool has_zc) {
  RowsPlan rp{false, 0, 0, 0, 0, 0, 0};
  if (G % 128 || K % 128 || N % (16 * ROWS_SPB)) return rp;
  const int chunks = K / 128, gchunks = G / 128, vstrips = N / (16 * ROWS_SPB);
  int best_score = -1;
  const int max_wgs = env_int("MI355Q_ROWS_MAX_WGS", 320), fill_cap = env_int("MI355Q_ROWS_FILL", 256);
  for (int s = 1; s <= 16; s++) {
    if (chunks % s || vstrips * s > max_wgs) continue;
    const int cps = chunks / s;
    for (int nw = 8; nw >= 1; nw--) {
      if (cps % nw) continue;
      const int per = cps / nw;
      for (int cpw = 4; cpw >= 2; cpw -= 2) {
        if (per % cpw) continue;
        // prefer the slice count that fills the chip, then bytes in flight per CU, then waves
        const int fill = std::min(vstrips * s, fill_cap);
        const int score = fill * 1000 + std::min(nw * cpw, 32) * 10 + nw;
        if (score > best_score) {
          best_score = score;
          rp.ok = true; rp.splits = s; rp.nw = nw; rp.cpw = cpw; rp.bursts = per / cpw;
        }
      }
    }
  }
  if (!rp.ok) return rp;
  const int cps = chunks / rp.splits;
  const int gps = (cps + gchunks - 1) / gchunks + 1;   // upper bound on the groups a slice touches
  const size_t fixed = (size_t)ROWS_SPB * gps * 16 * 2 * (has_zc ? 2 : 1) + 16;
  const size_t row = (size_t)(cps * 128 + 8) * 2 + (size_t)rp.nw * 16 * ROWS_SPB * 4;
  if (fixed + row > 160 * 1024) { rp.ok = false; return rp; }
  rp.rows_pp = (int)std::min<size_t>(16, (160 * 1024 - fixed) / row);
  rp.rows_pp = std::min(rp.rows_pp, M);
  rp.lds = fixed + (size_t)rp.rows_pp * row;
  return rp;
}

template <typename DT>
static int launch_rows(const RowsParams& p, int cpw, size_t lds, hipStream_t stream) {
  dim3 grid(p.N / (16 * ROWS_SPB), p.splits, (p.M + p.rows_pp - 1) / p.rows_pp), block(p.nw * 64);
#define MI355Q_R(c, z) hipLaunchKernelGGL((w4a16_rows_kernel<DT, ROWS_SPB, c, z>), grid, block, lds, stream, p)
  if (cpw == 4) { if (p.zc) MI355Q_R(4, true); else MI355Q_R(4, false); }
  else { if (p.zc) MI355Q_R(2, true); else MI355Q_R(2, false); }
#undef MI355Q_R
  return check_launch("w4a16_rows_kernel");
}

// ---------------------------------------------------------------------------
// Many-row kernel (prompt processing, 17+ rows): a workgroup of four waves computes a
// [16 RT rows] x [4 waves x CT 16-channel subtiles] tile over one K-slice.  The activation
// chunk [rows x 128 k] is staged in LDS (double-buffered, shared by the four waves); every wave
// streams the planes of its own SPW strips straight into registers -- they already are MFMA
// operand fragments -- one chunk ahead of use.  Same arithmetic as the decode kernels: per
// 128-k chunk acc = sum a (bias + q) and a1 = sum a on the matrix cores, then
// tot += s (acc - (bias + z) a1) in fp32; one rounding at the end.  K-slices (layers too narrow
// to occupy the chip otherwise) leave fp32 partial tiles that a second kernel sums in slice order.
// ---------------------------------------------------------------------------
struct MmParams {
  const uint16_t* x;
  int ldx;
  const u32x4* wp;
  const uint16_t* scales;
  const uint16_t* zc;
  const uint16_t* bias;
  uint16_t* out;
  float* partial;   // [splits][M][N] when splits > 1
  int M, N, K;
  int chunks, gchunks, cps, splits;
  int epi_silu;
  // grouped (token-sorted) expert GEMM: row tile t multiplies with the weights of expert tile_expert[t]
  // (negative: none) while t < *num_tiles; wp / scales / zc then hold all experts, these strides apart
  const int* tile_expert;
  const int* num_tiles;
  size_t wp_estride, sc_estride;   // in u32x4 / in scale elements
};

template <typename DT>
__device__ inline void mm_store(const MmParams& p, int row, int n, f32x4 a) {
  auto lo = [](uint32_t u) { return DT::to_f32((uint16_t)(u & 0xffff)); };
  auto hi = [](uint32_t u) { return DT::to_f32((uint16_t)(u >> 16)); };
  float v[4] = {a[0], a[1], a[2], a[3]};
  if (p.bias) {  // bias is added to the rounded GEMM result, as output.add_(bias) does
    const u32x2 bb = *reinterpret_cast<const u32x2*>(p.bias + n);
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = DT::to_f32(DT::from_f32(v[j]));
    v[0] += lo(bb[0]); v[1] += hi(bb[0]); v[2] += lo(bb[1]); v[3] += hi(bb[1]);
  }
  if (p.epi_silu) {
    const float g0 = DT::to_f32(DT::from_f32(v[0])), u0 = DT::to_f32(DT::from_f32(v[1]));
    const float g1 = DT::to_f32(DT::from_f32(v[2])), u1 = DT::to_f32(DT::from_f32(v[3]));
    const float s0 = DT::to_f32(DT::from_f32(silu_f(g0))), s1 = DT::to_f32(DT::from_f32(silu_f(g1)));
    *reinterpret_cast<uint32_t*>(p.out + (size_t)row * (p.N >> 1) + (n >> 1)) = pack2<DT>(s0 * u0, s1 * u1);
  } else {
    u32x2 o;
    o[0] = pack2<DT>(v[0], v[1]);
    o[1] = pack2<DT>(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p.out + (size_t)row * p.N + n) = o;
  }
}

// (32-row tiles are held to 256 registers; measured + 1 % on decode batches of 24 / 32)
template <typename DT, int NS, int SPW, int RT, bool HAS_ZC>
__global__ __launch_bounds__(256, RT == 2 ? 2 : 1) void w4a16_mm_kernel(const MmParams p) {
  constexpr int CT = NS * SPW, SW = 16 * NS, QW = 4 * NS, BM = 16 * RT;
  constexpr int ASTR = 128 + 16;  // LDS row stride (elements): 32 bytes of padding keep ds_read_b128 conflict-free
  __shared__ __attribute__((aligned(16))) uint16_t abuf[2][BM * ASTR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = lane >> 4, ml = lane & 15;
  const int strip0 = (blockIdx.x * 4 + wave) * SPW;
  const int row0 = blockIdx.y * BM;
  const int c0 = blockIdx.z * p.cps, c1 = c0 + p.cps;
  const int groups = p.chunks / p.gchunks;
  const int srow = threadIdx.x >> 4, sseg = threadIdx.x & 15;
  const u32x4* wp_e = p.wp;
  const uint16_t* sc_e = p.scales;
  const uint16_t* zc_e = p.zc;
  if (p.tile_expert) {   // uniform over the workgroup: whole tiles leave before any barrier
    if ((int)blockIdx.y >= *p.num_tiles) return;
    const int e = p.tile_expert[blockIdx.y];
    if (e < 0) return;
    wp_e += (size_t)e * p.wp_estride;
    sc_e += (size_t)e * p.sc_estride;
    if constexpr (HAS_ZC) zc_e += (size_t)e * p.sc_estride;
  }

  auto load_acts = [&](int c, u32x4 (&t)[RT]) {
#pragma unroll
    for (int i = 0; i < RT; i++) {
      const int r = min(row0 + srow + 16 * i, p.M - 1);
      t[i] = *reinterpret_cast<const u32x4*>(p.x + (size_t)r * p.ldx + (size_t)c * 128 + 8 * sseg);
    }
  };
  auto store_acts = [&](int buf, const u32x4 (&t)[RT]) {
#pragma unroll
    for (int i = 0; i < RT; i++) *reinterpret_cast<u32x4*>(&abuf[buf][(srow + 16 * i) * ASTR + 8 * sseg]) = t[i];
  };
  auto load_w = [&](int c, u32x4 (&w)[CT]) {
#pragma unroll
    for (int j = 0; j < SPW; j++)
#pragma unroll
      for (int pl = 0; pl < NS; pl++)
        w[j * NS + pl] = wp_e[(((size_t)(strip0 + j) * p.chunks + c) * NS + pl) * 64 + lane];
  };
  auto load_s = [&](int c, u32x2 (&sv)[CT], u32x2 (&zv)[CT]) {
    const int g = c / p.gchunks;
#pragma unroll
    for (int j = 0; j < SPW; j++)
#pragma unroll
      for (int sub = 0; sub < NS; sub++) {
        const size_t off = ((size_t)(strip0 + j) * groups + g) * SW + QW * kq + 4 * sub;
        sv[j * NS + sub] = *reinterpret_cast<const u32x2*>(sc_e + off);
        if constexpr (HAS_ZC) zv[j * NS + sub] = *reinterpret_cast<const u32x2*>(zc_e + off);
      }
  };
  auto lo = [](uint32_t u) { return DT::to_f32((uint16_t)(u & 0xffff)); };
  auto hi = [](uint32_t u) { return DT::to_f32((uint16_t)(u >> 16)); };

  u32x4 wc[CT], wn[CT], an[RT];
  u32x2 sc[CT], sn[CT], zv[CT], zn[CT];
  load_w(c0, wc);
  load_s(c0, sc, zv);
  load_acts(c0, an);
  store_acts(0, an);
  __syncthreads();

  f32x4 tot[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int ct = 0; ct < CT; ct++) tot[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = u32x4{DT::ones, DT::ones, DT::ones, DT::ones};
  // (x & mask) | magic in one instruction per 16-bit pair: both constants live in registers (see the ring kernel)
  uint32_t magic_v = DT::int4_magic, nib_mask = 0x000F000Fu;
  asm volatile("" : "+v"(magic_v), "+s"(nib_mask));
  auto expand = [&](uint32_t wd) {
    u32x4 r;
    r[0] = (wd & nib_mask) | magic_v;
    r[1] = ((wd >> 4) & nib_mask) | magic_v;
    r[2] = ((wd >> 8) & nib_mask) | magic_v;
    r[3] = ((wd >> 12) & nib_mask) | magic_v;
    return r;
  };

  for (int c = c0; c < c1; c++) {
    const int cur = (c - c0) & 1;
    // The next chunk's operands are requested unconditionally (the last iteration re-requests its own
    // chunk): with the loads behind a branch the compiler cannot count what is outstanding and waits
    // for all of it at the first consumer, which exposes a whole memory round trip in every chunk.
    const int cn = min(c + 1, c1 - 1);
    load_w(cn, wn);
    load_s(cn, sn, zn);
    // ... and they must be ISSUED here: left alone, the scheduler sinks them below the arithmetic
    // (shorter live ranges), where the next iteration then waits for them at once.  The 112-wide
    // strips have no registers left for the activation chunk as well: there only the weights (the
    // HBM round trip) are pinned, the activations (an L2 round trip) may sink.
    if constexpr (NS == 7) __builtin_amdgcn_sched_barrier(0);
    load_acts(cn, an);
    if constexpr (NS != 7) __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[RT][CT], acc1[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
      acc1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < CT; ct++) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      u32x4 bf[RT];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        bf[rt] = *reinterpret_cast<const u32x4*>(&abuf[cur][(16 * rt + ml) * ASTR + 32 * ks + 8 * kq]);
        acc1[rt] = mfma16<DT>(ones, bf[rt], acc1[rt]);
      }
#pragma unroll
      for (int j = 0; j < SPW; j++)
#pragma unroll
        for (int sub = 0; sub < NS; sub++) {
          const int e = ks * NS + sub;
          const u32x4 wf = expand(wc[j * NS + (e >> 2)][e & 3]);
#pragma unroll
          for (int rt = 0; rt < RT; rt++) acc[rt][j * NS + sub] = mfma16<DT>(wf, bf[rt], acc[rt][j * NS + sub]);
        }
    }
    // lane (kq, ml): rows 16 rt + ml, channels QW kq + 4 sub + r of each strip
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float s = (r & 1) ? hi(sc[ct][r >> 1]) : lo(sc[ct][r >> 1]);
        float z;
        if constexpr (HAS_ZC) z = (r & 1) ? hi(zv[ct][r >> 1]) : lo(zv[ct][r >> 1]);
        else z = DT::int4_bias + 8.0f;
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
          tot[rt][ct][r] = __builtin_fmaf(s, __builtin_fmaf(-z, acc1[rt][r], acc[rt][ct][r]), tot[rt][ct][r]);
      }
    }
    store_acts(cur ^ 1, an);
#pragma unroll
    for (int i = 0; i < CT; i++) { wc[i] = wn[i]; sc[i] = sn[i]; if constexpr (HAS_ZC) zv[i] = zn[i]; }
    __syncthreads();
  }

#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    const int row = row0 + 16 * rt + ml;
    if (row >= p.M) continue;
#pragma unroll
    for (int j = 0; j < SPW; j++)
#pragma unroll
      for (int sub = 0; sub < NS; sub++) {
        const int n = SW * (strip0 + j) + QW * kq + 4 * sub;
        if (p.splits > 1)
          *reinterpret_cast<f32x4*>(p.partial + ((size_t)blockIdx.z * p.M + row) * p.N + n) = tot[rt][j * NS + sub];
        else
          mm_store<DT>(p, row, n, tot[rt][j * NS + sub]);
      }
  }
}

// ---------------------------------------------------------------------------
// The same tile ([64 rows] x [4 waves x CT subtiles], one K-slice) with every operand of a chunk --
// activations, weight planes, scales -- brought in by LDS-DMA (global_load ... lds) into a ring of
// S stages, S - 1 chunks ahead of use.  The register-prefetching kernel above holds one chunk in
// flight per wave at one wave per SIMD, and a chunk's arithmetic (about 1.5 us) is shorter than a
// loaded memory round trip, so it runs at the speed of the memory latency; the ring holds 100+ KB
// in flight per CU without a single register.
//   stage image: A [64 rows][256 B], the 16-byte piece p of row r at position p ^ (r & 15) (conflict-
//   free ds_read_b128 of MFMA fragments; LDS-DMA writes lane-linear, so the source address carries
//   the permutation); W [wave][CT planes][1 KiB] lane-linear, read back as it was written; scales
//   [wave][strip][256 B] (one 4-byte piece per lane), read as the 8 bytes of a lane's 4 channels.
// LDS-DMA completes in issue order with the other vector-memory operations of the wave, so "stage c
// has landed" is s_waitcnt vmcnt(number of operations issued after it); there are no other
// vector-memory operations inside the loop.  Arithmetic and rounding as in the kernel above.
// ---------------------------------------------------------------------------
// What the two wave halves (waves 0-3, 4-7) of a workgroup split between them
constexpr int RING_ROWS = 0;   // the two 64-row halves of a 128-row tile
constexpr int RING_SUB = 1;    // subtiles 0-3 / 4-6 of a 112-wide strip, same 64 rows
constexpr int RING_K = 2;      // even / odd chunks of the same 64-row tile, summed through LDS at the end
constexpr int RING_SUBK = 3;   // 112-wide strips, few rows: two strips per workgroup; wave = (strip, even / odd chunk, subtile half)

template <int NS, int SPW, bool HAS_ZC, int MODE>
struct MmRing {
  static constexpr int CT = NS * SPW;
  static constexpr bool SUBK = MODE == RING_SUBK;
  static constexpr bool SPLIT_SUB = MODE == RING_SUB || SUBK, SPLIT_K = MODE == RING_K || SUBK;
  static constexpr int BM = MODE == RING_ROWS ? 128 : 64;
  static constexpr int QN = SUBK ? 2 : 4;             // strip groups (of SPW strips) per workgroup
  static constexpr int A_OPS = SPLIT_K ? 4 : BM / 32, W_OPS = MODE == RING_K ? CT : (CT + 1) / 2, SC_OPS = HAS_ZC ? 2 : 1;
  static constexpr int OPS = A_OPS + W_OPS + SC_OPS;  // LDS-DMA operations per wave and stage
  static constexpr int A_BYTES = BM * 256, W_BYTES = QN * CT * 1024, SC_BYTES = QN * 256;
  static constexpr int HALF = A_BYTES + W_BYTES + SC_BYTES * (HAS_ZC ? 2 : 1);   // one chunk's image
  static constexpr int STAGE = SPLIT_K ? 2 * HALF : HALF;                       // a stage of the K-split form holds two chunks
  static constexpr int S = (160 * 1024) / STAGE >= 4 ? 4 : (160 * 1024) / STAGE;
  static constexpr int LDS = S * STAGE;
  static_assert(!SPLIT_K || 4 * 16 * 64 * 16 <= LDS, "the final sum of the two K halves goes through the ring");
};

template <int N>
__device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Eight waves, two per SIMD, so that one wave's vector work (4-bit expansion, scale step) overlaps the
// other's matrix instructions: waves 0-3 and 4-7 take the same strips and either the two 64-row halves of
// a 128-row tile, or subtiles 0-3 and 4-6 of the same 64 rows (112-wide strips), or the even and the odd
// chunks of the same 64-row tile (few rows: half as many K-slices, i.e. partial tiles, fill the chip).
template <typename DT, int NS, int SPW, bool HAS_ZC, int MODE>
__global__ __launch_bounds__(512, 1) void w4a16_mm_ring_kernel(const MmParams p) {
  using R = MmRing<NS, SPW, HAS_ZC, MODE>;
  constexpr int CT = R::CT, SW = 16 * NS, RT = 4, BM = R::BM, S = R::S;
  constexpr int CTW = R::SPLIT_SUB ? 4 : CT;          // subtiles a wave owns (the upper half of a 112-wide strip: 3)
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: LDS-DMA targets stay in SGPRs
  const int wq = wave & 3, wh = wave >> 2;
  const int sq = R::SUBK ? (wq & 1) : wq;             // this wave's strip group
  const int kh = R::SUBK ? (wq >> 1) : (MODE == RING_K ? wh : 0);   // this wave's chunk of a two-chunk stage
  const int kq = lane >> 4, ml = lane & 15;
  const int strip0 = (blockIdx.x * R::QN + sq) * SPW;
  const int row0 = blockIdx.y * BM;
  const int rbase = MODE == RING_ROWS ? 64 * wh : 0;  // this wave's rows inside the tile
  const int sub0 = R::SPLIT_SUB ? 4 * wh : 0;         // this wave's first subtile
  const int nsub = R::SPLIT_SUB ? (wh ? 3 : 4) : CT;
  constexpr int CSTEP = R::SPLIT_K ? 2 : 1;           // chunks per stage
  const int cofs = kh;                                // this wave's chunk of a stage
  // which quarter (eighth) of the chunk's activation rows this wave fetches
  const int aw = R::SUBK ? 2 * (wq & 1) + wh : (MODE == RING_K ? wq : wave);
  const int c0 = blockIdx.z * p.cps, c1 = c0 + p.cps;
  const int groups = p.chunks / p.gchunks;
  const u32x4* wp_e = p.wp;
  const uint16_t* sc_e = p.scales;
  const uint16_t* zc_e = p.zc;
  if (p.tile_expert) {   // uniform over the workgroup: whole tiles leave before any barrier
    if ((int)blockIdx.y >= *p.num_tiles) return;
    const int e = p.tile_expert[blockIdx.y];
    if (e < 0) return;
    wp_e += (size_t)e * p.wp_estride;
    sc_e += (size_t)e * p.sc_estride;
    if constexpr (HAS_ZC) zc_e += (size_t)e * p.sc_estride;
  }
  // sources of this lane's pieces of a stage (chunk 0; a chunk is 256 bytes further along a row)
  const uint16_t* asrc[R::A_OPS];
#pragma unroll
  for (int i = 0; i < R::A_OPS; i++) {
    const int r = 4 * (R::A_OPS * aw + i) + (lane >> 4);
    asrc[i] = p.x + (size_t)min(row0 + r, p.M - 1) * p.ldx + 8 * ((lane & 15) ^ (r & 15));
  }
  // scales of the quad's SPW strips in ONE operation: lane l fetches 4-byte piece l % (SW/2) of strip l / (SW/2)
  const int sj = min(lane / (SW / 2), SPW - 1), sdw = lane % (SW / 2);
  auto issue = [&](int c_in, int slot) {
    const int c = c_in + cofs;
    unsigned char* st = ring + slot * R::STAGE + cofs * R::HALF;
#pragma unroll
    for (int i = 0; i < R::A_OPS; i++)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(asrc[i] + (size_t)c * 128),
                                       (void __attribute__((address_space(3)))*)(st + (R::A_OPS * aw + i) * 1024), 16, 0, 0);
    // the planes of the quad's strips are shared out between its two waves (an odd count: the last twice);
    // in the K-split form every wave fetches all planes of its own chunk
#pragma unroll
    for (int i = 0; i < R::W_OPS; i++) {
      const int pi = MODE == RING_K ? i : min(2 * i + wh, CT - 1), j = pi / NS, pl = pi % NS;
      __builtin_amdgcn_global_load_lds(
          (const void __attribute__((address_space(1)))*)(wp_e + (((size_t)(strip0 + j) * p.chunks + c) * NS + pl) * 64 + lane),
          (void __attribute__((address_space(3)))*)(st + R::A_BYTES + (sq * CT + pi) * 1024), 16, 0, 0);
    }
    const int g = c / p.gchunks;
    {   // both waves of a quad (same bytes, same place)
      const size_t off = ((size_t)(strip0 + sj) * groups + g) * SW + 2 * sdw;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(sc_e + off),
                                       (void __attribute__((address_space(3)))*)(st + R::A_BYTES + R::W_BYTES + sq * 256), 4, 0, 0);
      if constexpr (HAS_ZC)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(zc_e + off),
                                         (void __attribute__((address_space(3)))*)(st + R::A_BYTES + R::W_BYTES + R::SC_BYTES + sq * 256), 4, 0, 0);
    }
  };
  auto lo = [](uint32_t u) { return DT::to_f32((uint16_t)(u & 0xffff)); };
  auto hi = [](uint32_t u) { return DT::to_f32((uint16_t)(u >> 16)); };

#pragma unroll
  for (int i = 0; i < S - 1; i++)
    if (c0 + CSTEP * i < c1) issue(c0 + CSTEP * i, i);

  f32x4 tot[RT][CTW];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int ct = 0; ct < CTW; ct++) tot[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 ones = u32x4{DT::ones, DT::ones, DT::ones, DT::ones};
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // (x & mask) | magic as ONE instruction needs both constants in registers (gfx9 three-operand encodings
  // take no literal; with literals the compiler emits an AND and an OR): 7 instead of 11 per packed dword
  uint32_t magic_v = DT::int4_magic, nib_mask = 0x000F000Fu;
  asm volatile("" : "+v"(magic_v), "+s"(nib_mask));
  auto expand = [&](uint32_t w) {
    u32x4 r;
    r[0] = (w & nib_mask) | magic_v;
    r[1] = ((w >> 4) & nib_mask) | magic_v;
    r[2] = ((w >> 8) & nib_mask) | magic_v;
    r[3] = ((w >> 12) & nib_mask) | magic_v;
    return r;
  };

  int slot = 0;
  for (int c = c0; c < c1; c += CSTEP) {
    // stages after this one already issued: min(S - 2, stages left); each is OPS operations per wave
    const int ahead = min(S - 2, (c1 - CSTEP - c) / CSTEP);
    if (S >= 4 && ahead >= 2) wait_vm<2 * R::OPS>();
    else if (S >= 3 && ahead == 1) wait_vm<R::OPS>();
    else wait_vm<0>();
    // A bare barrier: __syncthreads() carries a fence that waits for EVERY outstanding memory operation,
    // the younger stages included.  This wave's pieces of stage c have landed (the wait above), its LDS
    // reads of chunk c - 1 have returned (their data was consumed), so the barrier alone makes stage c
    // complete for every wave and chunk c - 1's slot free.
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int free_slot = slot == 0 ? S - 1 : slot - 1;
    if (c + CSTEP * (S - 1) < c1) issue(c + CSTEP * (S - 1), free_slot);
    const unsigned char* st = ring + slot * R::STAGE + cofs * R::HALF;
    slot = slot == S - 1 ? 0 : slot + 1;

    u32x4 bf[4][RT];
    f32x4 acc1[RT];
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        bf[ks][rt] = *reinterpret_cast<const u32x4*>(st + (rbase + 16 * rt + ml) * 256 + 16 * ((4 * ks + kq) ^ ml));
        acc1[rt] = mfma16<DT>(ones, bf[ks][rt], ks == 0 ? zero4 : acc1[rt]);
      }
    // 112-wide strips on 128-row tiles: a wave's 7 subtiles x 4 row tiles of running sums leave no room for the
    // strip's 7 planes in registers as well; there every packed dword is read from LDS where it is used
    constexpr bool W_FROM_LDS = NS == 7 && MODE == RING_ROWS;
    u32x4 wc[W_FROM_LDS ? 1 : CT];
    if constexpr (!W_FROM_LDS) {
#pragma unroll
      for (int i = 0; i < CT; i++)
        wc[i] = *reinterpret_cast<const u32x4*>(st + R::A_BYTES + (sq * CT + i) * 1024 + 16 * lane);
    }
    // symmetric weights: the products start from -(bias + 8) a1 instead of zero, which leaves one
    // multiply-add per output for the scale step (same value as s (acc - z a1) up to the order of two roundings)
    f32x4 cinit[RT];
    if constexpr (!HAS_ZC) {
#pragma unroll
      for (int rt = 0; rt < RT; rt++) cinit[rt] = acc1[rt] * (-(DT::int4_bias + 8.0f));
    }
    // lane (kq, ml): rows 16 rt + ml, channels QW kq + 4 sub + r of each strip
#pragma unroll
    for (int ct = 0; ct < CTW; ct++) {
      if (R::SPLIT_SUB && ct >= nsub) break;           // wave-uniform
      const int j = R::SPLIT_SUB ? 0 : ct / NS;
      const int sub = R::SPLIT_SUB ? sub0 + ct : ct % NS;   // run-time only for the split strips
      const int soff = sq * 256 + j * (2 * SW) + 8 * (NS * kq + sub);
      const u32x2 sv = *reinterpret_cast<const u32x2*>(st + R::A_BYTES + R::W_BYTES + soff);
      u32x2 zv = u32x2{0u, 0u};
      if constexpr (HAS_ZC) zv = *reinterpret_cast<const u32x2*>(st + R::A_BYTES + R::W_BYTES + R::SC_BYTES + soff);
      f32x4 acc[RT];
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        uint32_t w;
        if constexpr (R::SPLIT_SUB) {   // dword ks * 7 + sub of the strip's 28; wh selects between two fixed ones
          const int e0 = ks * NS + ct, e1 = ks * NS + 4 + ct;
          const uint32_t w0 = wc[e0 >> 2][e0 & 3], w1 = wc[min(e1, 27) >> 2][min(e1, 27) & 3];
          w = wh ? w1 : w0;
        } else if constexpr (W_FROM_LDS) {
          const int e = ks * NS + sub;
          w = *reinterpret_cast<const uint32_t*>(st + R::A_BYTES + (sq * CT + (e >> 2)) * 1024 + 16 * lane + 4 * (e & 3));
        } else {
          const int e = ks * NS + sub;
          w = wc[j * NS + (e >> 2)][e & 3];
        }
        const u32x4 wf = expand(w);
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
          acc[rt] = mfma16<DT>(wf, bf[ks][rt], ks == 0 ? (HAS_ZC ? zero4 : cinit[rt]) : acc[rt]);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float s = (r & 1) ? hi(sv[r >> 1]) : lo(sv[r >> 1]);
        if constexpr (HAS_ZC) {
          const float z = (r & 1) ? hi(zv[r >> 1]) : lo(zv[r >> 1]);
#pragma unroll
          for (int rt = 0; rt < RT; rt++)
            tot[rt][ct][r] = __builtin_fmaf(s, __builtin_fmaf(-z, acc1[rt][r], acc[rt][r]), tot[rt][ct][r]);
        } else {
#pragma unroll
          for (int rt = 0; rt < RT; rt++) tot[rt][ct][r] = __builtin_fmaf(s, acc[rt][r], tot[rt][ct][r]);
        }
      }
    }
  }

  if constexpr (R::SPLIT_K) {   // odd-chunk sums join the even-chunk sums through the (now idle) ring
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int rslot = R::SUBK ? 2 * sq + wh : wq;     // the pair of waves that hold the same outputs
    f32x4* red = reinterpret_cast<f32x4*>(ring) + (size_t)rslot * (RT * CTW) * 64 + lane;
    if (kh) {
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < CTW; ct++) red[(rt * CTW + ct) * 64] = tot[rt][ct];
    }
    __syncthreads();
    if (kh) return;
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ct = 0; ct < CTW; ct++) tot[rt][ct] += red[(rt * CTW + ct) * 64];
  }
  constexpr int QW = 4 * NS;
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    const int row = row0 + rbase + 16 * rt + ml;
    if (row >= p.M) continue;
#pragma unroll
    for (int ct = 0; ct < CTW; ct++) {
      if (R::SPLIT_SUB && ct >= nsub) break;
      const int j = R::SPLIT_SUB ? 0 : ct / NS;
      const int sub = R::SPLIT_SUB ? sub0 + ct : ct % NS;
      const int n = SW * (strip0 + j) + QW * kq + 4 * sub;
      if (p.splits > 1)
        *reinterpret_cast<f32x4*>(p.partial + ((size_t)blockIdx.z * p.M + row) * p.N + n) = tot[rt][ct];
      else
        mm_store<DT>(p, row, n, tot[rt][ct]);
    }
  }
}

// ---------------------------------------------------------------------------
// 17..32 rows per pass (decode batches), one workgroup per strip: the weights of a strip are streamed
// ONCE, by eight waves that split K, exactly as the decode kernel above streams them -- the tiled
// kernel re-reads every strip per 32-row block and needs K-slices plus a second kernel to fill the
// chip, which costs 2-3x the streaming time at these sizes.  The activation rows do not fit in LDS
// (32 rows x K), so every lane reads its own MFMA operand fragments straight from global memory (L2
// resident: M x K x 2 bytes), one chunk ahead (XLDS: whole rows, turned into fragments through LDS,
// see below).  The per-chunk fp32 results tot += s (acc - (bias + z)
// acc1) are accumulated in the wave's own slice of LDS instead of registers (two accumulator sets per
// row tile do not fit beside the weight ring and the activation fragments at two waves per SIMD); the
// slices are summed across waves at the end.  Same arithmetic and rounding as the other kernels.
// ---------------------------------------------------------------------------
// REGACC: the running sums stay in registers (the per-chunk read-modify-write of the LDS slice is gone; same
// operations in the same order, so the same bits) -- where ring, fragments and two sets of sums fit in 256 registers.
template <typename DT, int NS, int CPW, bool HAS_ZC, bool REGACC = false, bool XLDS = false>
__global__ __launch_bounds__(512) void w4a16_stream_rows_kernel(const MmParams p, int bursts) {
  static_assert(!XLDS || REGACC, "the staging areas lie over the slices of the final sum");
  constexpr int SW = 16 * NS, QW = 4 * NS, RT = 2, NW = 8;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [NW][32][SW]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = lane >> 4, ml = lane & 15;
  const int strip = blockIdx.x, row0 = blockIdx.y * 32;
  const int Mt = min(32, p.M - row0);
  const int groups = p.chunks / p.gchunks;
  const int cfirst = wave * CPW * bursts;     // this wave's chunks: [cfirst, cfirst + CPW * bursts)
  const u32x4* wstrip = p.wp + (size_t)strip * p.chunks * NS * 64 + lane;
  const uint16_t* xr[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) xr[rt] = p.x + (size_t)min(row0 + 16 * rt + ml, p.M - 1) * p.ldx + 8 * kq;
  auto lo = [](uint32_t u) { return DT::to_f32((uint16_t)(u & 0xffff)); };
  auto hi = [](uint32_t u) { return DT::to_f32((uint16_t)(u >> 16)); };

  u32x4 w[CPW][NS];
  auto issue_chunk = [&](int burst, int i) {
    const size_t c = (size_t)(cfirst + burst * CPW + i);
#pragma unroll
    for (int pl = 0; pl < NS; pl++) w[i][pl] = __builtin_nontemporal_load(wstrip + (c * NS + pl) * 64);
  };
  // The activation chunk [32 rows][256 B] of a wave.  Fetched as operand fragments, one wave instruction gathers
  // 64-byte pieces of 16 rows and occupies the address path about four times as long as one that reads whole rows
  // (measured: the second row tile of a 32-row pass cost gate_up 3.5 us and qkv 3.1 us; a tile of one distinct
  // row costs next to nothing).  XLDS: the rows are fetched whole (4 rows x 256 B per instruction), written to the
  // wave's own 8 KiB of LDS (piece p of row r at position p ^ (r & 15): conflict-free both ways) and read back as
  // fragments -- same values in the same lanes.  The staging areas lie over the slices of the final sum, which the
  // running sums in registers leave unused until the end.  The plan says from which row count on it pays for a strip
  // width (at 32 rows, gathered -> staged: qkv 12.3 -> 9.3 us, gate_up 20.0 -> 19.0).
  unsigned char* xst = reinterpret_cast<unsigned char*>(red) + wave * 8192;
  const unsigned char* xbytes = reinterpret_cast<const unsigned char*>(p.x);
  uint32_t xoff[8];   // byte offset of piece ml of row 4 t + kq
#pragma unroll
  for (int t = 0; t < 8; t++) xoff[t] = (uint32_t)min(row0 + 4 * t + kq, p.M - 1) * (uint32_t)(2 * p.ldx) + 16u * ml;
  auto load_x = [&](int c, u32x4 (&xf)[4][RT]) {
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        if constexpr (XLDS) xf[ks][rt] = *reinterpret_cast<const u32x4*>(xbytes + xoff[2 * ks + rt] + (size_t)c * 256);
        else xf[ks][rt] = *reinterpret_cast<const u32x4*>(xr[rt] + (size_t)c * 128 + 32 * ks);
      }
  };
  // rows as fetched -> the wave's staging area (fragments are read from there one k-step ahead of their use);
  // gathered fragments are used as they are
  auto stage_rows = [&](const u32x4 (&xrow)[4][RT]) {
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int r = 4 * t + kq;
      *reinterpret_cast<u32x4*>(xst + r * 256 + 16 * (ml ^ (r & 15))) = xrow[t >> 1][t & 1];
    }
  };
  auto read_fragments = [&](int ks, u32x4 (&xf)[RT]) {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
      xf[rt] = *reinterpret_cast<const u32x4*>(xst + (16 * rt + ml) * 256 + 16 * ((4 * ks + kq) ^ ml));
  };
  u32x4 xc[XLDS ? 1 : 4][RT], xn[4][RT];
  if constexpr (XLDS) {
    load_x(cfirst, xn);
#pragma unroll
    for (int i = 0; i < CPW; i++) issue_chunk(0, i);
    stage_rows(xn);
  } else {
    load_x(cfirst, xc);
#pragma unroll
    for (int i = 0; i < CPW; i++) issue_chunk(0, i);
  }

  const u32x4 ones = u32x4{DT::ones, DT::ones, DT::ones, DT::ones};
  // (x & mask) | magic in one instruction per 16-bit pair, constants in registers as in the tiled kernel below
  uint32_t magic_v = DT::int4_magic, nib_mask = 0x000F000Fu;
  asm volatile("" : "+v"(magic_v), "+s"(nib_mask));
  auto expand = [&](uint32_t wd) {
    u32x4 r;
    r[0] = (wd & nib_mask) | magic_v;
    r[1] = ((wd >> 4) & nib_mask) | magic_v;
    r[2] = ((wd >> 8) & nib_mask) | magic_v;
    r[3] = ((wd >> 12) & nib_mask) | magic_v;
    return r;
  };
  float* mine = red + (size_t)wave * 32 * SW + ml * SW + QW * kq;   // + 16 rt SW + 4 sub
  const int last = cfirst + CPW * bursts - 1;
  f32x4 tot[REGACC ? RT : 1][REGACC ? NS : 1];
  if constexpr (REGACC) {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int sub = 0; sub < NS; sub++) tot[rt][sub] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int burst = 0; burst < bursts; burst++) {
#pragma unroll
    for (int i = 0; i < CPW; i++) {
      const int c = cfirst + burst * CPW + i;
      // requests of this chunk's scales and of the next chunk's activation fragments (the last chunk
      // re-requests itself: loads stay unconditional)
      u32x2 sv[NS], zv[NS];
      {
        const int g = c / p.gchunks;
#pragma unroll
        for (int sub = 0; sub < NS; sub++) {
          const size_t off = ((size_t)strip * groups + g) * SW + QW * kq + 4 * sub;
          sv[sub] = *reinterpret_cast<const u32x2*>(p.scales + off);
          if constexpr (HAS_ZC) zv[sub] = *reinterpret_cast<const u32x2*>(p.zc + off);
        }
      }
      load_x(min(c + 1, last), xn);
      __builtin_amdgcn_sched_barrier(0);
      // With the fragments in LDS a 112-wide strip is taken in two passes over the chunk (subtiles 0-3, then 4-6):
      // 8 instead of 14 accumulators per chunk are live beside the running sums, and the fragments are cheap to read
      // twice.  Every accumulator still sums its k-steps in the same order.
      constexpr int NH = (XLDS && NS == 7) ? 2 : 1, HS = NH == 2 ? 4 : NS;
      f32x4 acc1[RT];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) acc1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < NH; h++) {
        const int s0 = h * HS;
        f32x4 acc[RT][HS];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
          for (int sl = 0; sl < HS; sl++) acc[rt][sl] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 xf[2][RT];
        if constexpr (XLDS) read_fragments(0, xf[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
          if constexpr (XLDS) { if (ks < 3) read_fragments(ks + 1, xf[(ks + 1) & 1]); }
          u32x4 (&xk)[RT] = XLDS ? xf[ks & 1] : xc[XLDS ? 0 : ks];
          if (h == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) acc1[rt] = mfma16<DT>(ones, xk[rt], acc1[rt]);
          }
#pragma unroll
          for (int sl = 0; sl < HS; sl++) {
            if (s0 + sl >= NS) continue;
            const int e = ks * NS + s0 + sl;
            const u32x4 wf = expand(w[i][e >> 2][e & 3]);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) acc[rt][sl] = mfma16<DT>(wf, xk[rt], acc[rt][sl]);
          }
          __builtin_amdgcn_sched_barrier(0);  // dequantised fragments of one k-step live at a time
        }
        // the slot just consumed is refilled at once
        if (h == NH - 1 && burst + 1 < bursts) issue_chunk(burst + 1, i);
        // lane (kq, ml): rows 16 rt + ml, channels QW kq + 4 sub + r of the strip
#pragma unroll
        for (int sl = 0; sl < HS; sl++) {
          if (s0 + sl >= NS) continue;
          const int sub = s0 + sl;
          f32x4 sc, zz;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            sc[r] = (r & 1) ? hi(sv[sub][r >> 1]) : lo(sv[sub][r >> 1]);
            if constexpr (HAS_ZC) zz[r] = (r & 1) ? hi(zv[sub][r >> 1]) : lo(zv[sub][r >> 1]);
            else zz[r] = DT::int4_bias + 8.0f;
          }
#pragma unroll
          for (int rt = 0; rt < RT; rt++) {
            if constexpr (REGACC) {
#pragma unroll
              for (int r = 0; r < 4; r++)
                tot[rt][sub][r] = __builtin_fmaf(sc[r], __builtin_fmaf(-zz[r], acc1[rt][r], acc[rt][sl][r]), tot[rt][sub][r]);
            } else {
              f32x4* dst = reinterpret_cast<f32x4*>(mine + 16 * rt * SW + 4 * sub);
              f32x4 t = (burst == 0 && i == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : *dst;
#pragma unroll
              for (int r = 0; r < 4; r++) t[r] = __builtin_fmaf(sc[r], __builtin_fmaf(-zz[r], acc1[rt][r], acc[rt][sl][r]), t[r]);
              *dst = t;
            }
          }
        }
      }
      if constexpr (XLDS) {
        stage_rows(xn);
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
          for (int rt = 0; rt < RT; rt++) xc[ks][rt] = xn[ks][rt];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (REGACC) {
    if constexpr (XLDS) __syncthreads();   // the slices lie over the staging areas: every wave has read its last fragments
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int sub = 0; sub < NS; sub++) *reinterpret_cast<f32x4*>(mine + 16 * rt * SW + 4 * sub) = tot[rt][sub];
  }
  __syncthreads();
  for (int it = threadIdx.x; it < Mt * QW; it += 512) {
    const int m_l = it / QW, nl = 4 * (it % QW);
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NW; i++) a += *reinterpret_cast<const f32x4*>(red + ((size_t)i * 32 + m_l) * SW + nl);
    mm_store<DT>(p, row0 + m_l, strip * SW + nl, a);
  }
}

// (NS, CPW) pairs the streaming rows kernel is built for
static bool stream_rows_built(int ns, int cpw) {
  switch (ns) {
    case 1: return cpw == 1 || cpw == 2 || cpw == 4 || cpw == 7 || cpw == 8;
    case 2: return cpw == 1 || cpw == 2 || cpw == 4;
    case 4: return cpw == 1 || cpw == 2 || cpw == 4;
    case 7: return cpw == 1 || cpw == 2;
    default: return false;
  }
}

struct StreamRowsPlan {
  bool ok;
  int ns, cpw, bursts;
  size_t lds;
  bool regacc;
  bool xlds;   // activation rows fetched whole and turned into fragments through LDS (register-sum kernels only)
};

// rows 17..32, strips that alone occupy the chip, K that eight waves divide.  The kernel itself takes up to 64
// rows in two passes over the weights (MI355Q_STREAM_ROWS_MAX=64), but from 33 rows on the LDS-ring form of the
// tiled kernel reads them once and wins: gate_up at 64 rows 42.0 -> 33.0 us, qkv 23.0 -> 19.5; whole decode step
// at batch 48 4.27 -> 3.80 ms, batch 64 4.82 -> 4.29 ms.
static StreamRowsPlan plan_stream_rows(int M, int N, int K, int G, int sw) {
  StreamRowsPlan sp{false, 0, 0, 0, 0, false, false};
  const int max_rows = env_int("MI355Q_STREAM_ROWS_MAX", 32);   // looked up per call: mi355q_set_option switches it
  // 112-wide strips from 13 rows on: the single-pass GEMV kernel holds 12 activation rows of K = 4096 and takes a
  // second pass over the weights beyond (gate_up + SiLU at 14 / 16 rows: 27.3 / 27.5 us, streamed 16.7 / 16.0; at 12
  // rows 15.5 against 16.6).  16-wide strips on a short K likewise (o_proj at 13 / 16 rows, with the stand-alone
  // add-and-norm: 10.9 / 11.2 -> 10.1 / 9.8 us).  32-wide strips gain nothing below 17 rows (qkv at 16 rows 9.05
  // against 9.13 gathered, 8.65 with the whole-row fetch; 8.7 at 13-14 rows either way).
  const int min_rows = env_int("MI355Q_STREAM_ROWS_MIN", (sw == 112 || sw == 16) ? 13 : 17);
  if (M < min_rows || M > max_rows || G % 128 || K % 128 || sw % 16 || N % sw) return sp;
  const int ns = sw / 16, chunks = K / 128;
  // Every workgroup reads all M x K activations, so the activation traffic is (N / sw) x M x K x 2 bytes:
  // on 16-wide strips that is 8x the weights at 32 rows (measured: down_proj 14336x4096 31 us against 21 us
  // for the tiled kernel with its K-slice merge, o_proj 12.4 + norm against 7.4 + merge-and-norm 6) -- the
  // tiled kernel shares one activation tile between 256 channels and keeps those layers.  On 32- and
  // 112-wide strips streaming wins (qkv 12.7 against 11.7 + 6.4, gate_up 23.5 against 23.2 + 5.6).
  // With the whole-row fetch a short K changes that for 16-wide strips: o_proj 4096 x 4096 streamed + stand-alone
  // add-and-norm 10.5 / 10.2 / 10.2 us at 32 / 24 / 17 rows against 12.4 / 11.7 / 11.4 for the tiled kernel with its
  // merge-and-norm (gathered fragments: 14.4 / 12.7 at 32 / 25); down_proj (K = 14336) 33 us against 19.
  const int regacc = env_int("MI355Q_STREAM_ROWS_REGACC", 2);
  // whole-row activation fetch (looked up per call): 16-wide strips always, 32-wide from 17 rows on (with the two-chunk ring qkv
  // 9.4 / 9.8 / 10.4 -> 9.0 / 8.9 / 8.9 us at 17 / 20 / 24 rows), 112-wide strips from 23 (gate_up 16.1 / 16.6 / 17.8 ->
  // 17.8 / 17.5 / 17.3: the two subtile passes cost more than the gather of a second row tile that is mostly one row)
  const bool want_xlds = M >= env_int("MI355Q_STREAM_ROWS_XLDS_MIN", ns == 7 ? 23 : (ns == 1 ? 13 : 17));
  const bool narrow_ok = ns == 1 && want_xlds && regacc >= 1 && chunks <= 32;
  const int min_ns = env_int("MI355Q_STREAM_ROWS_MIN_NS", 2);
  if ((ns < min_ns && !narrow_ok) || chunks % 8 || (long)(N / sw) < 64) return sp;
  const int per = chunks / 8;
  // Running sums in registers: 32-wide strips, and 112-wide strips with a one-chunk ring (two chunks of ring and two
  // sets of sums do not fit in 256 registers).  gate_up at 17 / 32 rows 19.6 -> 16.7 / 21.8 -> 20.3 us; qkv does not
  // move (12.2 us: bound by the activation fragments every workgroup reads from L2, 5 x its weights).
  // MI355Q_STREAM_ROWS_REGACC = 0 keeps the sums in LDS, 1 leaves 112-wide strips on the two-chunk ring.
  const int cands[5] = {8, 7, 4, 2, 1};
  for (int cpw : cands) {
    if (per % cpw || cpw * ns > 16 || !stream_rows_built(ns, cpw)) continue;
    // a four-chunk ring beside the whole-row fetch spills registers (qkv at 32 rows: 10.9 us, 9.3 with two chunks)
    if (want_xlds && regacc >= 1 && ns == 2 && cpw > 2) continue;
    if (regacc == 2 && ns == 7 && cpw != 1) continue;
    if (narrow_ok && cpw > 4) continue;   // the register-sum kernels of 16-wide strips are built for rings of 1, 2, 4 chunks
    sp.regacc = (regacc >= 1 && ns == 2) || (regacc == 2 && ns == 7 && cpw == 1) || narrow_ok;
    sp.ok = true; sp.ns = ns; sp.cpw = cpw; sp.bursts = per / cpw;
    // the eight 8 KiB staging areas lie over the slices
    sp.xlds = sp.regacc && want_xlds;
    const size_t slices = (size_t)8 * 32 * sw * sizeof(float), staging = (size_t)8 * 8192;
    sp.lds = sp.xlds && staging > slices ? staging : slices;
    return sp;
  }
  return sp;
}

template <typename DT>
static int launch_stream_rows(const MmParams& p, const StreamRowsPlan& sp, hipStream_t stream) {
  dim3 grid(p.N / (16 * sp.ns), (p.M + 31) / 32), block(512);
#define MI355Q_SR(NS_, CPW_)                                                                                             \
  if (sp.ns == NS_ && sp.cpw == CPW_) {                                                                                  \
    if (p.zc) hipLaunchKernelGGL((w4a16_stream_rows_kernel<DT, NS_, CPW_, true>), grid, block, sp.lds, stream, p, sp.bursts);  \
    else hipLaunchKernelGGL((w4a16_stream_rows_kernel<DT, NS_, CPW_, false>), grid, block, sp.lds, stream, p, sp.bursts);      \
    return check_launch("w4a16_stream_rows_kernel");                                                                     \
  }
#define MI355Q_SRR(NS_, CPW_)                                                                                            \
  if (sp.regacc && sp.ns == NS_ && sp.cpw == CPW_) {                                                                     \
    if (sp.xlds) {                                                                                                       \
      if constexpr (NS_ * CPW_ <= 7) { /* the plan never pairs it with a longer ring */                                  \
        if (p.zc) hipLaunchKernelGGL((w4a16_stream_rows_kernel<DT, NS_, CPW_, true, true, true>), grid, block, sp.lds, stream, p, sp.bursts);  \
        else hipLaunchKernelGGL((w4a16_stream_rows_kernel<DT, NS_, CPW_, false, true, true>), grid, block, sp.lds, stream, p, sp.bursts);      \
      } else {                                                                                                           \
        return set_error(MI355Q_EINVAL, "w4a16_stream_rows: no whole-row kernel for %d subtiles x %d chunks", sp.ns, sp.cpw);  \
      }                                                                                                                  \
    } else {                                                                                                             \
      if (p.zc) hipLaunchKernelGGL((w4a16_stream_rows_kernel<DT, NS_, CPW_, true, true>), grid, block, sp.lds, stream, p, sp.bursts);  \
      else hipLaunchKernelGGL((w4a16_stream_rows_kernel<DT, NS_, CPW_, false, true>), grid, block, sp.lds, stream, p, sp.bursts);      \
    }                                                                                                                    \
    return check_launch("w4a16_stream_rows_kernel");                                                                     \
  }
  MI355Q_SRR(2, 1) MI355Q_SRR(2, 2) MI355Q_SRR(2, 4) MI355Q_SRR(7, 1) MI355Q_SRR(1, 1) MI355Q_SRR(1, 2) MI355Q_SRR(1, 4)
#undef MI355Q_SRR
  MI355Q_SR(1, 1) MI355Q_SR(1, 2) MI355Q_SR(1, 4) MI355Q_SR(1, 7) MI355Q_SR(1, 8)
  MI355Q_SR(2, 1) MI355Q_SR(2, 2) MI355Q_SR(2, 4)
  MI355Q_SR(4, 1) MI355Q_SR(4, 2) MI355Q_SR(4, 4)
  MI355Q_SR(7, 1) MI355Q_SR(7, 2)
#undef MI355Q_SR
  return set_error(MI355Q_EINVAL, "w4a16_stream_rows: no kernel for %d subtiles x %d chunks", sp.ns, sp.cpw);
}

template <typename DT>
__global__ __launch_bounds__(256) void w4a16_mm_reduce_kernel(const MmParams p) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;   // 4 consecutive channels of one row
  const size_t nq = (size_t)p.N >> 2;
  if (q >= (size_t)p.M * nq) return;
  const int row = (int)(q / nq), n = 4 * (int)(q - (size_t)row * nq);
  f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < p.splits; sp++) a += *reinterpret_cast<const f32x4*>(p.partial + ((size_t)sp * p.M + row) * p.N + n);
  mm_store<DT>(p, row, n, a);
}

constexpr int MM_RT = 4;
// 32 rows or fewer (decode batches beyond the GEMV kernels' reach): 32-row tiles, half the matrix-core
// and accumulator work per weight chunk of the 64-row tile
static inline int mm_rt(int M) {
  const int rt2_max = env_int("MI355Q_MM_RT2_MAX", 32);
  return M <= rt2_max ? 2 : MM_RT;
}

// The LDS-ring form of the tiled kernel serves 33+ rows on the full-width workgroup forms (4+ subtiles per
// wave quad); its row tile is 128 except on 112-wide strips.  The grouped expert GEMM sorts rows into blocks
// of 64, so there it applies to 112-wide strips only.  Measured on the 8B layers (us, register-prefetch form
// -> ring form): 2048 rows down 353 -> 271, o 105 -> 82, gate_up 665 -> 629, qkv 150 -> 154; 512 rows down
// 105 -> 86; 128 rows gate_up 63.6 -> 57.6, but qkv 33.9 -> 37.7 and o 23.0 -> 24.2 on a single 128-row tile.
// Returns the form (RING_ROWS / RING_SUB / RING_K / RING_SUBK) or -1 for the register-prefetch kernel (the
// knobs are looked up per call so that mi355q_set_option can switch forms inside one process).  cps = chunks per
// K-slice when known (the K-split form takes them in pairs), else 0; colblocks = N / workgroup width.
// Rows or K between the wave halves: a 128-row tile takes about 1.85 x the time of a K-split 64-row tile
// (measured on the 8B layers at 2048 rows), so the choice is which grid wastes less of its last round of 256
// workgroups -- 2048 rows: down / o 16 x 16 tiles of 128 rows = one round (255 us against 278 K-split), qkv 24 x 16
// = 1.5 rounds (146 us against 121 K-split in three full rounds).
static inline int mm_ring_form(int ns, int spw, bool grouped, int M, int cps, int colblocks) {
  const int use_ring = env_int("MI355Q_MM_RING", 1);
  const int use_ksplit = env_int("MI355Q_MM_RING_K", 1);
  const int use_subk = env_int("MI355Q_MM_RING_SUBK", 1);
  const int force_rows_min = env_int("MI355Q_MM_RING_MIN", 0);   // experiments: 128-row tiles from this many rows on
  if (!use_ring || ns * spw < 4) return -1;
  // 112-wide strips, up to 128 rows: two strips per workgroup with K split inside it, so that 4096 x 28672 fills
  // the chip without K-slices (no partial tiles, SiLU in the epilogue)
  if (ns == 7) {
    if (use_subk && !grouped && M <= 128 && cps % 2 == 0) return RING_SUBK;
    // 128-row tiles (each wave a whole strip for 64 rows: no second copy of the activation sums, balanced halves)
    // take about 1.7 x the time of a subtile-split 64-row tile: worth it when their grid fills its rounds
    const long tr = (long)colblocks * ((M + 127) / 128), lr = tr % 256;
    const long rounds_s = ((long)colblocks * ((M + 63) / 64) + 255) / 256;
    const bool rows7 = env_int("MI355Q_MM_RING_ROWS7", 1) && !grouped && M > 128 &&
                       (force_rows_min > 0 ? M >= force_rows_min
                                           : ((tr + 255) / 256) * 170 < rounds_s * 100 && (lr == 0 || lr >= 230));
    return rows7 ? RING_ROWS : RING_SUB;
  }
  const bool k_ok = use_ksplit && cps % 2 == 0;
  if (grouped) return k_ok ? RING_K : -1;
  const long tiles_r = (long)colblocks * ((M + 127) / 128);
  const long rounds_r = (tiles_r + 255) / 256, last_r = tiles_r % 256;
  const long rounds_k = ((long)colblocks * ((M + 63) / 64) + 255) / 256;
  // ... and only when the 128-row grid's last round is (nearly) full: qkv at 1024 rows, 192 tiles, 92 us against 80
  const bool rows = force_rows_min > 0 ? M >= force_rows_min
                                       : (M > 128 && rounds_r * 185 < rounds_k * 100 && (last_r == 0 || last_r >= 230));
  if (rows || (!k_ok && M > 128)) return RING_ROWS;
  return k_ok ? RING_K : -1;
}

struct MmPlan {
  bool ok;
  int ns, spw, splits;
  size_t ws;   // bytes of fp32 partial tiles
};

static MmPlan plan_mm(int M, int N, int K, int G, int sw, bool grouped = false) {
  MmPlan mp{false, 0, 0, 1, 0};
  if (G % 128 || K % 128 || sw % 16) return mp;
  const int ns = sw / 16;
  if (ns != 1 && ns != 2 && ns != 4 && ns != 7) return mp;
  // Strips per wave: a workgroup covers 4 spw strips.  Narrower workgroups (MI355Q_MM_SPW) need fewer
  // K-slices to fill the chip and measured 10-25 % faster as stand-alone GEMMs on the short-K 8B layers
  // up to about a thousand rows, but inside the prompt pass, where the slice sums are merged by the fused
  // consumers (residual add + RMSNorm, RoPE + cache write), the whole pass measured slower with them
  // (TTFT(128) 5.36 -> 5.56 ms), so the wide forms stay the default.
  int spw = ns == 1 ? 4 : (ns == 2 ? 2 : 1);
  // decode batches (32-row tiles) on a short K: 128-wide workgroups need half the K-slices to fill the chip, and
  // with 32 chunks the slices are what the launch costs (o_proj 4096 x 4096 at 32 rows, with its merge-and-norm:
  // 15.2 -> 12.3 us; down_proj, 112 chunks, 19.7 -> 21.1 and keeps the wide form)
  // A long K (down_proj, 112 chunks) takes them too, with K-slices until two workgroups share a CU: the 256-wide form
  // holds one chunk in flight per SIMD (1 workgroup of 224 registers per lane on a CU), the 128-wide one (112
  // registers) two; down_proj + merge-and-norm at 32 / 24 / 17 rows 19.7 / 18.6 / 18.6 -> 18.2 / 17.8 / 17.5 us with
  // 16 slices of 7 chunks on 512 workgroups (8 slices of 14 chunks: 21.1)
  const bool rt2_narrow = mm_rt(M) == 2 && ns == 1 && !grouped;
  if (rt2_narrow) spw = 2;
  while (spw > 1 && N % (4 * spw * sw)) spw >>= 1;
  if (const int o = env_int("MI355Q_MM_SPW", 0); o > 0 && o <= 4 / ns && (o & (o - 1)) == 0) spw = o;   // experiments
  const int bw = 4 * spw * sw;
  if (N % bw) return mp;
  const int chunks = K / 128;
  const int bm = (mm_rt(M) == MM_RT && mm_ring_form(ns, spw, grouped, M, 0, N / bw) == RING_ROWS) ? 128 : 16 * mm_rt(M);
  const int form0 = mm_rt(M) == MM_RT ? mm_ring_form(ns, spw, grouped, M, 0, N / bw) : -1;
  const long tiles = (long)(N / bw) * (form0 == RING_SUBK ? 2 : 1) * ((M + bm - 1) / bm);   // two strips per workgroup there
  // K-slices until about 200 workgroups exist.  The K-split ring form runs eight waves on a tile: from 120
  // workgroups on it prefers slices of 8+ chunks to more workgroups (measured at 128 rows: qkv 30.2 -> 23.8 us
  // with 4 instead of 8 slices, o 21.4 -> 20.7; down_proj, whose slices stay long, keeps its 8: 33.4 vs 43.4).
  const bool ksplit = mm_rt(M) == MM_RT && mm_ring_form(ns, spw, grouped, M, 0, N / bw) == RING_K;
  const long want = rt2_narrow && spw == 2 && chunks > 32 ? 400 : 200;
  int s = 1;
  while (tiles * s < want && s < 16 && chunks % (s * 2) == 0 && chunks / (s * 2) >= 2) {
    if (ksplit && tiles * s >= 120 && chunks / (s * 2) < 8) break;
    if (want > 200 && tiles * s >= 200 && chunks / (s * 2) < 7) break;   // the second workgroup per CU only on slices of 7+ chunks
    s *= 2;
  }
  if (tiles * s < want) {   // odd factors of the chunk count (K = 14336 = 128 * 112 = 128 * 16 * 7)
    for (int f = 7; f >= 3; f -= 2)
      if (chunks % (s * f) == 0 && chunks / (s * f) >= 2 && tiles * s * f <= 512) { s *= f; break; }
  }
  if (const int o = env_int("MI355Q_MM_SPLITS", 0); o > 0 && chunks % o == 0) s = o;                  // experiments
  mp.ok = true; mp.ns = ns; mp.spw = spw; mp.splits = s;
  mp.ws = s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
  return mp;
}

template <typename DT, int NS, int SPW, int MODE>
static void launch_mm_ring_mode(const MmParams& p, int bw, hipStream_t stream) {
  using R0 = MmRing<NS, SPW, false, MODE>;
  using R1 = MmRing<NS, SPW, true, MODE>;
  dim3 grid(p.N / (R0::QN * SPW * 16 * NS), (p.M + R0::BM - 1) / R0::BM, p.splits), block(512);
  if (p.zc) hipLaunchKernelGGL((w4a16_mm_ring_kernel<DT, NS, SPW, true, MODE>), grid, block, R1::LDS, stream, p);
  else hipLaunchKernelGGL((w4a16_mm_ring_kernel<DT, NS, SPW, false, MODE>), grid, block, R0::LDS, stream, p);
}

template <typename DT, int NS, int SPW>
static void launch_mm_ring(const MmParams& p, int bw, int form, hipStream_t stream) {
  if constexpr (NS * SPW >= 4) {
    if constexpr (NS == 7) {
      if (form == RING_SUBK) launch_mm_ring_mode<DT, NS, SPW, RING_SUBK>(p, bw, stream);
      else if (form == RING_ROWS) launch_mm_ring_mode<DT, NS, SPW, RING_ROWS>(p, bw, stream);
      else launch_mm_ring_mode<DT, NS, SPW, RING_SUB>(p, bw, stream);
    } else {
      if (form == RING_K) launch_mm_ring_mode<DT, NS, SPW, RING_K>(p, bw, stream);
      else launch_mm_ring_mode<DT, NS, SPW, RING_ROWS>(p, bw, stream);
    }
  }
}

template <typename DT>
static int launch_mm(const MmParams& p, const MmPlan& mp, hipStream_t stream, bool reduce = true) {
  const int bw = 4 * mp.spw * 16 * mp.ns;
  const int rt = mm_rt(p.M);
  dim3 grid(p.N / bw, (p.M + 16 * rt - 1) / (16 * rt), p.splits), block(256);
  const int ring = mm_ring_form(mp.ns, mp.spw, p.tile_expert != nullptr, p.M, p.cps, p.N / bw);
#define MI355Q_MM_RT(NS_, SPW_, RT_)                                                                       \
  {                                                                                                       \
    if (p.zc) hipLaunchKernelGGL((w4a16_mm_kernel<DT, NS_, SPW_, RT_, true>), grid, block, 0, stream, p);    \
    else hipLaunchKernelGGL((w4a16_mm_kernel<DT, NS_, SPW_, RT_, false>), grid, block, 0, stream, p);        \
  }
#define MI355Q_MM(NS_, SPW_)                                                                              \
  if (mp.ns == NS_ && mp.spw == SPW_) {                                                                   \
    if (rt == 2) MI355Q_MM_RT(NS_, SPW_, 2)                                                                \
    else if (ring >= 0 && NS_ * SPW_ >= 4) launch_mm_ring<DT, NS_, SPW_>(p, bw, ring, stream);             \
    else MI355Q_MM_RT(NS_, SPW_, MM_RT)                                                                    \
  }
  MI355Q_MM(1, 4) else MI355Q_MM(1, 2) else MI355Q_MM(1, 1) else MI355Q_MM(2, 2) else MI355Q_MM(2, 1)
  else MI355Q_MM(4, 1) else MI355Q_MM(7, 1)
#undef MI355Q_MM
#undef MI355Q_MM_RT
  if (int e = check_launch("w4a16_mm_kernel")) return e;
  if (p.splits > 1 && reduce) {
    const size_t items = (size_t)p.M * (p.N >> 2);
    hipLaunchKernelGGL((w4a16_mm_reduce_kernel<DT>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, p);
    return check_launch("w4a16_mm_reduce_kernel");
  }
  return MI355Q_OK;
}

struct NormArgs {
  const void* res_in;
  void* res_out;
  const void* norm_w;
  float eps;
  const float* mix_w = nullptr;   // PRO_MIXNORM: [M, 2] routing weights
};

struct MoeArgs {
  const int* expert_ids;  // [M] expert of every output row, negative = not on this rank
  int x_row_div;          // input row of output row r is r / x_row_div
  int num_experts;
};

static int gemm_impl(const void* x, int ldx, const void* packed, const void* scales, const void* zc,
                     const void* bias, void* c, int M, int N, int K, int G, int sw, int act_dtype,
                     void* workspace, size_t workspace_bytes, int split_k, hipStream_t stream,
                     int pro, const NormArgs* na, const MoeArgs* moe = nullptr, int epi_silu = 0) {
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_tile(sw, K, G, N)) return e;
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w4a16_gemm: act_dtype must be bf16 or fp16");
  if (M == 0) return MI355Q_OK;
  MI355Q_REQUIRE(x && packed && scales && c, "w4a16_gemm: null pointer");
  MI355Q_REQUIRE(M > 0, "w4a16_gemm: M=%d", M);
  MI355Q_REQUIRE(ldx % 8 == 0 && ldx >= (pro == PRO_SILU ? 2 * K : K), "w4a16_gemm: row stride %d must be a multiple of 8 and cover the row", ldx);
  MI355Q_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)scales & 15) == 0 &&
                     ((uintptr_t)c & 7) == 0 && ((uintptr_t)zc & 15) == 0 && ((uintptr_t)bias & 7) == 0,
                 "w4a16_gemm: pointers must be 16-byte aligned");
  // several rows against the 16-channel-strip layout: strips side by side, K sliced over workgroups
  if (!moe && pro == PRO_NONE && !epi_silu && sw == 16 && M >= ROWS_MIN_M && !env_flag("MI355Q_NO_ROWS")) {
    RowsPlan rp = plan_rows(M, N, K, G, zc != nullptr);
    // short K: the single-strip kernel still holds every row of a pass in LDS and needs no K-slices
    const BurstPlan sb = plan_burst(M, K, G, 1, zc != nullptr);
    if (sb.ok && sb.rows_pp >= std::min(M, 16) && (size_t)std::min(M, 16) * K * 2 <= 96 * 1024) rp.ok = false;
    const size_t need = COUNTER_BYTES + (size_t)rp.splits * M * N * sizeof(float);
    const bool ws_ok = rp.splits == 1 || (workspace && ((uintptr_t)workspace & 255) == 0 && workspace_bytes >= need &&
                                          (size_t)(N / (16 * ROWS_SPB)) * ((M + rp.rows_pp - 1) / std::max(rp.rows_pp, 1)) * sizeof(int) <= COUNTER_BYTES);
    if (rp.ok && ws_ok) {
      RowsParams r;
      r.x = (const uint16_t*)x; r.ldx = ldx; r.wp = (const u32x4*)packed;
      r.scales = (const uint16_t*)scales; r.zc = (const uint16_t*)zc; r.bias = (const uint16_t*)bias;
      r.out = (uint16_t*)c;
      r.counters = (int*)workspace;
      r.partial = workspace ? (float*)((char*)workspace + COUNTER_BYTES) : nullptr;
      r.M = M; r.N = N; r.K = K; r.chunks = K / 128; r.gchunks = G / 128;
      r.splits = rp.splits; r.cps = r.chunks / rp.splits; r.nw = rp.nw; r.bursts = rp.bursts; r.rows_pp = rp.rows_pp;
      r.partial_only = 0;
      return act_dtype == MI355Q_BF16 ? launch_rows<BF16>(r, rp.cpw, rp.lds, stream)
                                      : launch_rows<FP16>(r, rp.cpw, rp.lds, stream);
    }
  }
  // decode path: one burst workgroup per strip whenever the strips alone occupy the chip
  {
    const int ns = sw / 16;
    BurstPlan bp = plan_burst(moe ? 1 : M, K, G, ns, zc != nullptr);
    const char* nb = env_value("MI355Q_NO_BURST");
    const long blocks = moe ? (long)(N / sw) * M : (long)(N / sw) * ((M + 15) / 16);
    const int split_below = env_int("MI355Q_SPLIT_BELOW", 64);   // fewer strips than this: K-sliced kernel (80 strips of 8192 k: 5.9 us here, 7.6 K-sliced)
    const bool want_split = !epi_silu && (split_k > 1 || blocks < split_below);
    if (moe) {
      bp.ok = bp.ok && pro != PRO_ADDNORM && bias == nullptr && moe->expert_ids && moe->x_row_div >= 1;
      bp.rows_pp = 1;
    }
    if (G % 128 && pro == PRO_SILU) bp.ok = false;   // that prologue is built for 128-multiples only
    if (bp.ok && !(nb && atoi(nb)) && (!legacy_kernel_ok(sw, K, G) || !want_split)) {
      BurstParams b;
      b.expert_ids = nullptr; b.x_row_div = 1; b.wp_estride = 0; b.sc_estride = 0;
      if (moe) {
        b.expert_ids = moe->expert_ids; b.x_row_div = moe->x_row_div;
        b.wp_estride = (long long)K * N / 32;    // K*N/2 bytes per expert
        b.sc_estride = (long long)(K / G) * N;
      }
      b.x = (const uint16_t*)x; b.ldx = ldx; b.wp = (const u32x4*)packed;
      b.scales = (const uint16_t*)scales; b.zc = (const uint16_t*)zc; b.bias = (const uint16_t*)bias;
      b.out = (uint16_t*)c;
      b.res_in = nullptr; b.res_out = nullptr; b.norm_w = nullptr; b.eps = 0.f; b.mix_w = nullptr;
      if (pro == PRO_MIXNORM) {
        MI355Q_REQUIRE(na && na->mix_w && M == 1 && !moe && K / 8 <= BURST_XV * bp.nw * 64 && ldx == K,
                       "w4a16_gemm_moe_sum_add_rms_norm: one token, rows of K <= %d channels", 8 * BURST_XV * bp.nw * 64);
        b.mix_w = na->mix_w;
      }
      if (pro == PRO_ADDNORM || pro == PRO_MIXNORM) {
        MI355Q_REQUIRE(na && na->res_out, "w4a16_gemm_add_rms_norm: residual_out is required");
        MI355Q_REQUIRE(((uintptr_t)na->res_in & 15) == 0 && ((uintptr_t)na->res_out & 15) == 0 && ((uintptr_t)na->norm_w & 15) == 0,
                       "w4a16_gemm_add_rms_norm: pointers must be 16-byte aligned");
        MI355Q_REQUIRE(na->res_out != na->res_in && na->res_out != x, "w4a16_gemm_add_rms_norm: residual_out must not alias an input");
        b.res_in = (const uint16_t*)na->res_in; b.res_out = (uint16_t*)na->res_out;
        b.norm_w = (const uint16_t*)na->norm_w; b.eps = na->eps;
      }
      b.M = M; b.N = N; b.K = K; b.chunks = K / 128; b.groups = K / G; b.kpg = G / 32;
      b.rows_pp = bp.rows_pp; b.nw = bp.nw; b.bursts = bp.bursts;
      b.stamps = g_burst_stamps;
      b.epi_silu = epi_silu;
      MI355Q_REQUIRE(!epi_silu || bias == nullptr, "w4a16_gemm: the SiLU-pair epilogue takes no bias");
#define MI355Q_BPRO(DTT)                                                                       \
  switch (pro) {                                                                               \
    case PRO_SILU: return launch_burst<DTT, PRO_SILU>(b, ns, bp.cpw, bp.lds, stream);          \
    case PRO_ADDNORM: return launch_burst<DTT, PRO_ADDNORM>(b, ns, bp.cpw, bp.lds, stream);    \
    case PRO_MIXNORM: return launch_burst<DTT, PRO_MIXNORM>(b, ns, bp.cpw, bp.lds, stream);    \
    default: return launch_burst<DTT, PRO_NONE>(b, ns, bp.cpw, bp.lds, stream);                \
  }
      if (act_dtype == MI355Q_BF16) { MI355Q_BPRO(BF16) }
      MI355Q_BPRO(FP16)
#undef MI355Q_BPRO
    }
  }
  MI355Q_REQUIRE(!epi_silu, "w4a16_gemm: the SiLU-pair epilogue needs a layer the burst kernel serves "
                            "(group size a multiple of 128, at least 128 strips)");
  MI355Q_REQUIRE(pro != PRO_MIXNORM, "w4a16_gemm_moe_sum_add_rms_norm: layer shape not served by the one-workgroup-per-strip kernel");
  MI355Q_REQUIRE(legacy_kernel_ok(sw, K, G), "w4a16_gemm: strip width %d with group size %d needs K = 128 * waves * {2,4,7} and at least 128 strips", sw, G);
  const int CS = chunk_steps(G);
  SkinnyParams p;
  p.x = (const uint16_t*)x;
  p.ldx = ldx;
  p.wp = (const u32x4*)packed;
  p.scales = (const uint16_t*)scales;
  p.zc = (const uint16_t*)zc;
  p.bias = (const uint16_t*)bias;
  p.out = (uint16_t*)c;
  p.res_in = nullptr; p.res_out = nullptr; p.norm_w = nullptr; p.eps = 0.f;
  if (pro == PRO_ADDNORM) {
    MI355Q_REQUIRE(na && na->res_out, "w4a16_gemm_add_rms_norm: residual_out is required");
    MI355Q_REQUIRE(((uintptr_t)na->res_in & 15) == 0 && ((uintptr_t)na->res_out & 15) == 0 && ((uintptr_t)na->norm_w & 15) == 0,
                   "w4a16_gemm_add_rms_norm: pointers must be 16-byte aligned");
    MI355Q_REQUIRE(na->res_out != na->res_in && na->res_out != x, "w4a16_gemm_add_rms_norm: residual_out must not alias an input");
    p.res_in = (const uint16_t*)na->res_in;
    p.res_out = (uint16_t*)na->res_out;
    p.norm_w = (const uint16_t*)na->norm_w;
    p.eps = na->eps;
  }
  p.M = M; p.N = N; p.K = K; p.G = G;
  p.mrows = 16; p.x_row_div = 1; p.expert_ids = nullptr; p.wp_estride = 0; p.sc_estride = 0;
  if (moe) {
    MI355Q_REQUIRE(moe->expert_ids && moe->x_row_div >= 1 && moe->num_experts >= 1, "moe_w4a16_gemm: bad expert arguments");
    MI355Q_REQUIRE(pro != PRO_ADDNORM && bias == nullptr, "moe_w4a16_gemm: no bias or norm prologue");
    p.mrows = 1;
    p.x_row_div = moe->x_row_div;
    p.expert_ids = moe->expert_ids;
    p.wp_estride = (long long)K * N / 32;   // K*N/2 bytes per expert
    p.sc_estride = (long long)(K / G) * N;
    split_k = 1;                            // rows of different experts never share a tile
  }
  p.chunks = K / (32 * CS);
  const bool lds_pro = pro != PRO_NONE;
  const Plan plan = make_plan(moe ? 1 : M, N, K, G, sw, lds_pro, split_k);
  p.splits = plan.splits;
  p.nw = plan.nw;
  size_t lds = (size_t)p.nw * (sw / 16) * 1024 + 16;
  if (lds_pro) {
    const int pb = (p.chunks + p.splits - 1) / p.splits;
    lds += (size_t)pb * act_bytes_per_chunk(CS, moe ? 1 : M);
  }
  MI355Q_REQUIRE(lds <= 160 * 1024, "w4a16_gemm: %zu bytes of LDS needed (K range per block too long)", lds);
  p.counters = nullptr;
  p.partial = nullptr;
  if (p.splits > 1) {
    MI355Q_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 255) == 0, "w4a16_gemm: split-K needs a 256-byte aligned workspace");
    const size_t need = COUNTER_BYTES + (size_t)p.splits * M * N * sizeof(float);
    if (workspace_bytes < need)
      return set_error(MI355Q_ENOSPACE, "w4a16_gemm: workspace %zu B < %zu B", workspace_bytes, need);
    p.counters = (int*)workspace;
    p.partial = (float*)((char*)workspace + COUNTER_BYTES);
  }
#define MI355Q_PRO(DTT)                                                                \
  switch (pro) {                                                                       \
    case PRO_SILU: return launch_skinny<DTT, PRO_SILU>(p, sw, CS, stream, lds);        \
    case PRO_ADDNORM: return launch_skinny<DTT, PRO_ADDNORM>(p, sw, CS, stream, lds);  \
    default: return launch_skinny<DTT, PRO_NONE>(p, sw, CS, stream, lds);              \
  }
  if (act_dtype == MI355Q_BF16) { MI355Q_PRO(BF16) }
  MI355Q_PRO(FP16)
#undef MI355Q_PRO
}

// ---- dequantize ---------------------------------------------------------------
// One workgroup dequantises a [32 k x 64 n] tile: each thread expands one packed dword (8 k
// of one channel) into LDS, then the tile leaves as 32 rows of 128 contiguous bytes.
template <typename DT>
__global__ __launch_bounds__(256) void dequantize_kernel(const uint32_t* __restrict__ wp,
                                                         const uint16_t* __restrict__ scales,
                                                         const uint16_t* __restrict__ zc,
                                                         uint16_t* __restrict__ W, int K, int N, int G,
                                                         int sw) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[32][64 + 8];
  const int nb = blockIdx.x * 64, kb = blockIdx.y * 32;
  const int t = threadIdx.x;
  {
    const int kq = t >> 6, nl = t & 63;
    const int n = nb + nl, k0 = kb + 8 * kq;
    size_t d;
    int sh;
    packed_pos(K, sw, k0, n, &d, &sh);
    const int g = k0 / G;
    const size_t si = scale_idx(g, n, K / G, sw);
    const float s = DT::to_f32(scales[si]);
    const float z = zc ? DT::to_f32(zc[si]) - DT::int4_bias : 8.0f;
    const uint32_t w = wp[d];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float q = (float)((w >> nibble_shift(j)) & 0xfu);
      tile[8 * kq + j][nl] = DT::from_f32((q - z) * s);
    }
  }
  __syncthreads();
  const int row = t >> 3, v = t & 7;
  *reinterpret_cast<u32x4*>(W + (size_t)(kb + row) * N + nb + 8 * v) = *reinterpret_cast<const u32x4*>(&tile[row][8 * v]);
}

}  // namespace mi355q

extern "C" size_t mi355q_w4a16_workspace_bytes(int M, int N, int K, int G) {
  if (G <= 0) G = K;
  if (M <= 0 || N <= 0 || K <= 0 || G % 32 != 0) return 0;
  const int CS = chunk_steps(G);
  const int chunks = K / (32 * CS);
  const int smax = chunks < 64 ? chunks : 64;
  return COUNTER_BYTES + (size_t)smax * M * N * sizeof(float);
}

extern "C" int mi355q_workspace_init(void* workspace, size_t workspace_bytes, mi355q_stream_t stream) {
  MI355Q_REQUIRE(workspace != nullptr, "workspace_init: null workspace");
  hipError_t e = hipMemsetAsync(workspace, 0, workspace_bytes, (hipStream_t)stream);
  if (e != hipSuccess) return set_error(MI355Q_EHIP, "workspace_init: %s", hipGetErrorString(e));
  return MI355Q_OK;
}

extern "C" int mi355q_w4a16_gemm(const void* a, int lda, const void* packed, const void* scales,
                                 const void* zc, const void* bias, void* c, int M, int N, int K,
                                 int G, int tile, int act_dtype, void* workspace,
                                 size_t workspace_bytes, int split_k, mi355q_stream_t stream) {
  return gemm_impl(a, lda, packed, scales, zc, bias, c, M, N, K, G, tile, act_dtype, workspace,
                   workspace_bytes, split_k, (hipStream_t)stream, PRO_NONE, nullptr);
}

extern "C" int mi355q_w4a16_gemm_silu_mul(const void* x, int ldx, const void* packed,
                                          const void* scales, const void* zc, const void* bias,
                                          void* c, int M, int N, int K, int G, int tile,
                                          int act_dtype, void* workspace, size_t workspace_bytes,
                                          int split_k, mi355q_stream_t stream) {
  return gemm_impl(x, ldx, packed, scales, zc, bias, c, M, N, K, G, tile, act_dtype, workspace,
                   workspace_bytes, split_k, (hipStream_t)stream, PRO_SILU, nullptr);
}

extern "C" int mi355q_w4a16_gemm_add_rms_norm(const void* x, int ldx, const void* residual_in,
                                              void* residual_out, const void* norm_weight,
                                              float epsilon, const void* packed, const void* scales,
                                              const void* zc, const void* bias, void* c, int M, int N,
                                              int K, int G, int tile, int act_dtype, void* workspace,
                                              size_t workspace_bytes, int split_k,
                                              mi355q_stream_t stream) {
  NormArgs na{residual_in, residual_out, norm_weight, epsilon};
  return gemm_impl(x, ldx, packed, scales, zc, bias, c, M, N, K, G, tile, act_dtype, workspace,
                   workspace_bytes, split_k, (hipStream_t)stream, PRO_ADDNORM, &na);
}

extern "C" int mi355q_w4a16_gemm_moe_sum_add_rms_norm_supported(int num_tokens, int topk, int N, int K, int G, int tile) {
  if (G <= 0) G = K;
  if (num_tokens != 1 || topk != 2 || K <= 0 || N <= 0 || tile <= 0 || tile % 16 || N % tile || K % 128) return 0;
  if (!tile_ok(tile, K, G)) return 0;
  const BurstPlan bp = plan_burst(1, K, G, tile / 16, false);
  const long blocks = N / tile;
  return bp.ok && K / 8 <= BURST_XV * bp.nw * 64 && blocks >= env_int("MI355Q_SPLIT_BELOW", 64) ? 1 : 0;
}

extern "C" int mi355q_w4a16_gemm_moe_sum_add_rms_norm(const void* expert_out, const float* topk_weights, int topk,
                                                       const void* residual_in, void* residual_out,
                                                       const void* norm_weight, float epsilon, const void* packed,
                                                       const void* scales, const void* zc, const void* bias, void* c,
                                                       int num_tokens, int N, int K, int G, int tile, int act_dtype,
                                                       mi355q_stream_t stream) {
  MI355Q_REQUIRE(expert_out && topk_weights, "w4a16_gemm_moe_sum_add_rms_norm: null pointer");
  MI355Q_REQUIRE(mi355q_w4a16_gemm_moe_sum_add_rms_norm_supported(num_tokens, topk, N, K, G, tile),
                 "w4a16_gemm_moe_sum_add_rms_norm: one token, two selected experts, a layer the one-workgroup-per-strip "
                 "kernel serves (tokens %d, top-%d, %d x %d)", num_tokens, topk, K, N);
  NormArgs na{residual_in, residual_out, norm_weight, epsilon, topk_weights};
  return gemm_impl(expert_out, K, packed, scales, zc, bias, c, num_tokens, N, K, G, tile, act_dtype, nullptr, 0, 1,
                   (hipStream_t)stream, PRO_MIXNORM, &na);
}

extern "C" int mi355q_w4a16_dequantize(const void* packed, const void* scales, const void* zc,
                                       void* w, int K, int N, int G, int tile, int act_dtype,
                                       mi355q_stream_t stream) {
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_tile(tile, K, G, N)) return e;
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w4a16_dequantize: act_dtype must be bf16 or fp16");
  MI355Q_REQUIRE(packed && scales && w, "w4a16_dequantize: null pointer");
  MI355Q_REQUIRE(((uintptr_t)w & 15) == 0, "w4a16_dequantize: output must be 16-byte aligned");
  const dim3 grid(N / 64, K / 32);
  if (act_dtype == MI355Q_BF16)
    hipLaunchKernelGGL((dequantize_kernel<BF16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)packed, (const uint16_t*)scales, (const uint16_t*)zc,
                       (uint16_t*)w, K, N, G, tile);
  else
    hipLaunchKernelGGL((dequantize_kernel<FP16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)packed, (const uint16_t*)scales, (const uint16_t*)zc,
                       (uint16_t*)w, K, N, G, tile);
  return check_launch("dequantize_kernel");
}

/* ---- mixture of experts ------------------------------------------------------ */
extern "C" int mi355q_moe_w4a16_gemm(const void* x, int ldx, int x_row_div, const void* packed,
                                     const void* scales, const void* zc, const int32_t* expert_ids,
                                     void* c, int slots, int num_experts, int N, int K, int G,
                                     int tile, int act_dtype, int fuse_silu_mul,
                                     mi355q_stream_t stream) {
  MoeArgs moe{expert_ids, x_row_div, num_experts};
  MI355Q_REQUIRE(fuse_silu_mul >= 0 && fuse_silu_mul <= 2, "moe_w4a16_gemm: fuse_silu_mul must be 0, 1 or 2");
  return gemm_impl(x, ldx, packed, scales, zc, nullptr, c, slots, N, K, G, tile, act_dtype, nullptr, 0, 1,
                   (hipStream_t)stream, fuse_silu_mul == 1 ? PRO_SILU : PRO_NONE, nullptr, &moe, fuse_silu_mul == 2);
}

// Strip width for decode: strips of 16 NS channels such that the layer gives about one
// workgroup per CU in as few rounds as possible (estimate: rounds x (fixed cost + strip bytes at
// the per-CU ingest rate)).  Only widths the burst kernel is built for are considered; layers
// with too few strips to occupy the chip keep the widths of the split-K capable kernel.
// rows: the rows a decode launch runs at once with a workgroup each -- 1 for a dense layer, the (token, expert) slots
// of a routed layer at batch 1 (top-k): two slots on 16-wide strips of a 4096-wide layer are two rounds of 256
// workgroups, on 32-wide strips one
static int choose_strip_for(int K, int N, int G, int rows) {
  if (G <= 0) G = K;
  const int legacy = mi355q_w4a16_choose_tile(K, N, G);
  if (env_flag("MI355Q_TILE") || (G % 128 && G != 64 && G != 32) || K % 128) return legacy;
  int best = 0;
  double best_cost = 1e30;
  const int cands[4] = {1, 2, 4, 7};
  for (int ns : cands) {
    const int sw = 16 * ns;
    if (N % sw || !tile_ok(sw, K, G) || !plan_burst(1, K, G, ns, false).ok) continue;
    const int blocks = N / sw * rows;
    if (blocks < 128) continue;
    const int rounds = (blocks + 255) / 256;
    const double cost = rounds * (3.0 + (double)K * sw / 2 / 24e3);  // us; 24 GB/s per CU ~ 10 B/cycle
    if (cost < best_cost) { best_cost = cost; best = sw; }
  }
  return best ? best : legacy;
}

extern "C" int mi355q_w4a16_choose_strip(int K, int N, int G) { return choose_strip_for(K, N, G, 1); }

extern "C" int mi355q_moe_w4a16_choose_strip(int K, int N, int G, int decode_slots) {
  return choose_strip_for(K, N, G, decode_slots < 1 ? 1 : decode_slots);
}

extern "C" int mi355q_w4a16_gate_up_silu(const void* x, int ldx, const void* residual_in,
                                         void* residual_out, const void* norm_weight, float epsilon,
                                         const void* packed, const void* scales, const void* zc, void* c,
                                         int M, int N, int K, int G, int tile, int act_dtype,
                                         mi355q_stream_t stream) {
  if (residual_out == nullptr)   // no norm in front: plain GEMM + SiLU-pair epilogue
    return gemm_impl(x, ldx, packed, scales, zc, nullptr, c, M, N, K, G, tile, act_dtype, nullptr, 0, 0,
                     (hipStream_t)stream, PRO_NONE, nullptr, nullptr, 1);
  NormArgs na{residual_in, residual_out, norm_weight, epsilon};
  return gemm_impl(x, ldx, packed, scales, zc, nullptr, c, M, N, K, G, tile, act_dtype, nullptr, 0, 0,
                   (hipStream_t)stream, PRO_ADDNORM, &na, nullptr, 1);
}

extern "C" int mi355q_w4a16_gate_up_silu_supported(int K, int N, int G, int tile) {
  if (G <= 0) G = K;
  if (K <= 0 || N <= 0 || N % 64 || tile <= 0 || N % tile || !tile_ok(tile, K, G)) return 0;
  return plan_burst(1, K, G, tile / 16, false).ok ? 1 : 0;
}

/* ---- 2- and 3-bit GPTQ checkpoints ---------------------------------------------------
 * AutoGPTQ packs b-bit values as one little-endian bit string per column (weights, along K) or per row
 * (zero points, along N): value i sits at bits [b i, b i + b) of the string, 32-bit words in order -- for
 * b = 3 that is the 32-values-in-3-words layout of qdq_3.cuh:21-23.  The values are widened to the 4-bit
 * layout the kernels read (plus `add`), so these checkpoints run on the W4A16 path unchanged: symmetric
 * weights use add = 8 - 2^(b-1) with the constant zero point 8, zero-point checkpoints add = 0. */
namespace mi355q {
__global__ void gptq_expand_bits_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int out_rows,
                                        int out_cols, int src_cols, int bits, int add, int axis) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)out_rows * out_cols) return;
  const int r = (int)(idx / out_cols), c = (int)(idx - (long)r * out_cols);
  const uint32_t mask = (1u << bits) - 1u;
  uint32_t out = 0u;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    // value index along the packed axis, its place in the source bit string
    const long v = axis == 0 ? (long)r * 8 + j : (long)c * 8 + j;
    const long bit = v * bits;
    const int word = (int)(bit >> 5), sh = (int)(bit & 31);
    const size_t at = axis == 0 ? (size_t)word * src_cols + c : (size_t)r * src_cols + word;
    uint32_t val = src[at] >> sh;
    if (sh + bits > 32) val |= src[axis == 0 ? at + src_cols : at + 1] << (32 - sh);
    out |= (((val & mask) + (uint32_t)add) & 0xfu) << (4 * j);
  }
  dst[idx] = out;
}
}  // namespace mi355q

extern "C" int mi355q_gptq_expand_bits(const void* src, void* dst, int values, int other, int bits, int add, int axis,
                                       mi355q_stream_t stream) {
  MI355Q_REQUIRE(src && dst, "gptq_expand_bits: null pointer");
  MI355Q_REQUIRE(bits == 2 || bits == 3, "gptq_expand_bits: %d-bit values (2 or 3 expected; 4 needs no widening)", bits);
  MI355Q_REQUIRE(axis == 0 || axis == 1, "gptq_expand_bits: axis %d", axis);
  MI355Q_REQUIRE(values > 0 && other > 0 && values % 32 == 0, "gptq_expand_bits: %d values along the packed axis (multiple of 32 expected)", values);
  MI355Q_REQUIRE(add >= 0 && add + (1 << bits) - 1 <= 15, "gptq_expand_bits: add = %d leaves the nibble", add);
  const int out_rows = axis == 0 ? values / 8 : other, out_cols = axis == 0 ? other : values / 8;
  const int src_cols = axis == 0 ? other : values * bits / 32;
  const long n = (long)out_rows * out_cols;
  hipLaunchKernelGGL(gptq_expand_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)src, (uint32_t*)dst, out_rows, out_cols, src_cols, bits, add, axis);
  return check_launch("gptq_expand_bits_kernel");
}

/* Diagnostics: when set, every burst-kernel workgroup writes four 100 MHz timestamps (entry,
 * activations staged, weights consumed, done) at stamps[4 * strip ..].  Not part of the public ABI. */
extern "C" void mi355q_debug_burst_stamps(void* stamps) { g_burst_stamps = (unsigned long long*)stamps; }

/* ---- many rows (prompt processing) ------------------------------------------- */
extern "C" size_t mi355q_w4a16_mm_workspace_bytes(int M, int N, int K, int G, int tile) {
  if (G <= 0) G = K;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const MmPlan mp = plan_mm(M, N, K, G, tile);
  return mp.ok ? mp.ws : 0;
}

extern "C" int mi355q_w4a16_mm_min_rows(int N, int K, int G, int tile) {
  if (G <= 0) G = K;
  if (N <= 0 || K <= 0 || tile <= 0 || N % tile || !tile_ok(tile, K, G)) return 17;
  for (int m = 2; m < 17; m++)
    if (plan_stream_rows(m, N, K, G, tile).ok) return m;
  return 17;
}

extern "C" int mi355q_w4a16_mm_streams(int M, int N, int K, int G, int tile) {
  if (G <= 0) G = K;
  if (M <= 0 || N <= 0 || K <= 0 || tile <= 0 || N % tile || !tile_ok(tile, K, G)) return 0;
  return plan_stream_rows(M, N, K, G, tile).ok ? 1 : 0;
}

extern "C" int mi355q_w4a16_mm_supported(int N, int K, int G, int tile) {
  if (G <= 0) G = K;
  if (N <= 0 || K <= 0 || tile <= 0 || N % tile || !tile_ok(tile, K, G)) return 0;
  return plan_mm(64, N, K, G, tile).ok ? 1 : 0;
}

extern "C" int mi355q_w4a16_mm(const void* a, int lda, const void* packed, const void* scales, const void* zc,
                               const void* bias, void* c, int M, int N, int K, int G, int tile, int act_dtype,
                               int silu_pairs, void* workspace, size_t workspace_bytes, mi355q_stream_t stream) {
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_tile(tile, K, G, N)) return e;
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w4a16_mm: act_dtype must be bf16 or fp16");
  if (M == 0) return MI355Q_OK;
  MI355Q_REQUIRE(a && packed && scales && c && M > 0, "w4a16_mm: bad arguments");
  MI355Q_REQUIRE(lda % 8 == 0 && lda >= K, "w4a16_mm: row stride %d must be a multiple of 8 and cover the row", lda);
  MI355Q_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)scales & 15) == 0 &&
                     ((uintptr_t)c & 7) == 0 && ((uintptr_t)zc & 15) == 0 && ((uintptr_t)bias & 7) == 0,
                 "w4a16_mm: pointers must be 16-byte aligned");
  MI355Q_REQUIRE(!silu_pairs || bias == nullptr, "w4a16_mm: the SiLU-pair epilogue takes no bias");
  MmParams p{};
  p.x = (const uint16_t*)a; p.ldx = lda; p.wp = (const u32x4*)packed;
  p.scales = (const uint16_t*)scales; p.zc = (const uint16_t*)zc; p.bias = (const uint16_t*)bias;
  p.out = (uint16_t*)c; p.partial = nullptr;
  p.M = M; p.N = N; p.K = K; p.chunks = K / 128; p.gchunks = G / 128;
  p.epi_silu = silu_pairs ? 1 : 0;
  // decode batches of 17..32 rows: stream every strip once
  if (const StreamRowsPlan sp = plan_stream_rows(M, N, K, G, tile); sp.ok) {
    p.splits = 1; p.cps = p.chunks;
    return act_dtype == MI355Q_BF16 ? launch_stream_rows<BF16>(p, sp, (hipStream_t)stream)
                                    : launch_stream_rows<FP16>(p, sp, (hipStream_t)stream);
  }
  const MmPlan mp = plan_mm(M, N, K, G, tile);
  MI355Q_REQUIRE(mp.ok, "w4a16_mm: needs a group size that is a multiple of 128 and N a multiple of %d for strip width %d",
                 tile == 112 ? 448 : 256, tile);
  p.splits = mp.splits; p.cps = p.chunks / mp.splits;
  if (mp.splits > 1) {
    MI355Q_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "w4a16_mm: K-slices need a 16-byte aligned workspace");
    if (workspace_bytes < mp.ws) return set_error(MI355Q_ENOSPACE, "w4a16_mm: workspace %zu B < %zu B", workspace_bytes, mp.ws);
    p.partial = (float*)workspace;
  }
  return act_dtype == MI355Q_BF16 ? launch_mm<BF16>(p, mp, (hipStream_t)stream) : launch_mm<FP16>(p, mp, (hipStream_t)stream);
}

/* ---- grouped expert GEMM over token-sorted rows (MoE prompt batches) -------------- */
extern "C" int mi355q_moe_grouped_block_rows(void) { return 16 * MM_RT; }

extern "C" size_t mi355q_moe_w4a16_grouped_mm_workspace_bytes(int rows, int N, int K, int G, int tile) {
  if (G <= 0) G = K;
  if (rows <= 16 * 2 || N <= 0 || K <= 0) return 0;
  const MmPlan mp = plan_mm(rows, N, K, G, tile, true);
  return mp.ok ? mp.ws : 0;
}

extern "C" int mi355q_moe_w4a16_grouped_mm(const void* a, int lda, const void* packed, const void* scales,
                                           const void* zc, const int* tile_expert, const int* num_tiles, void* c,
                                           int rows, int num_experts, int N, int K, int G, int tile, int act_dtype,
                                           int silu_pairs, void* workspace, size_t workspace_bytes,
                                           mi355q_stream_t stream) {
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_tile(tile, K, G, N)) return e;
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "moe_w4a16_grouped_mm: act_dtype must be bf16 or fp16");
  if (rows == 0) return MI355Q_OK;
  const int bm = 16 * MM_RT;
  MI355Q_REQUIRE(a && packed && scales && c && tile_expert && num_tiles && num_experts > 0, "moe_w4a16_grouped_mm: bad arguments");
  MI355Q_REQUIRE(rows > 0 && rows % bm == 0 && mm_rt(rows) == MM_RT,
                 "moe_w4a16_grouped_mm: %d rows; whole tiles of %d rows (at least one) are expected", rows, bm);
  MI355Q_REQUIRE(lda % 8 == 0 && lda >= K, "moe_w4a16_grouped_mm: row stride %d must be a multiple of 8 and cover the row", lda);
  MI355Q_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)scales & 15) == 0 &&
                     ((uintptr_t)c & 7) == 0 && ((uintptr_t)zc & 15) == 0,
                 "moe_w4a16_grouped_mm: pointers must be 16-byte aligned");
  const MmPlan mp = plan_mm(rows, N, K, G, tile, true);
  MI355Q_REQUIRE(mp.ok, "moe_w4a16_grouped_mm: needs a group size that is a multiple of 128 and N a multiple of %d for strip width %d",
                 tile == 112 ? 448 : 256, tile);
  MmParams p{};
  p.x = (const uint16_t*)a; p.ldx = lda; p.wp = (const u32x4*)packed;
  p.scales = (const uint16_t*)scales; p.zc = (const uint16_t*)zc; p.bias = nullptr;
  p.out = (uint16_t*)c; p.partial = nullptr;
  p.M = rows; p.N = N; p.K = K; p.chunks = K / 128; p.gchunks = G / 128;
  p.epi_silu = silu_pairs ? 1 : 0;
  p.tile_expert = tile_expert; p.num_tiles = num_tiles;
  p.wp_estride = (size_t)K * N / 32;            // K N / 2 bytes per expert
  p.sc_estride = (size_t)(K / G) * N;
  p.splits = mp.splits; p.cps = p.chunks / mp.splits;
  if (mp.splits > 1) {
    MI355Q_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "moe_w4a16_grouped_mm: K-slices need a 16-byte aligned workspace");
    if (workspace_bytes < mp.ws) return set_error(MI355Q_ENOSPACE, "moe_w4a16_grouped_mm: workspace %zu B < %zu B", workspace_bytes, mp.ws);
    p.partial = (float*)workspace;
  }
  return act_dtype == MI355Q_BF16 ? launch_mm<BF16>(p, mp, (hipStream_t)stream) : launch_mm<FP16>(p, mp, (hipStream_t)stream);
}

// 5..16 rows on 16-channel strips whose activations do not fit one workgroup's LDS: the K-sliced rows
// kernel serves them (same choice as mi355q_w4a16_gemm makes)
static bool rows_route(int M, int N, int K, int G, int tile, bool has_zc, RowsPlan* out) {
  if (tile != 16 || M < ROWS_MIN_M || M >= 17 || env_flag("MI355Q_NO_ROWS")) return false;
  RowsPlan rp = plan_rows(M, N, K, G, has_zc);
  if (!rp.ok) return false;
  const BurstPlan sb = plan_burst(M, K, G, 1, has_zc);
  if (sb.ok && sb.rows_pp >= std::min(M, 16) && (size_t)std::min(M, 16) * K * 2 <= 96 * 1024) return false;
  *out = rp;
  return true;
}

extern "C" int mi355q_w4a16_mm_add_rms_norm_supported(int M, int N, int K, int G, int tile, int has_zc) {
  if (G <= 0) G = K;
  if (M <= 0 || N <= 0 || K <= 0 || tile <= 0 || N % tile || !tile_ok(tile, K, G)) return 0;
  RowsPlan rp;
  if (rows_route(M, N, K, G, tile, has_zc != 0, &rp)) return 1;
  return M >= 17 && plan_mm(M, N, K, G, tile).ok ? 1 : 0;
}

extern "C" size_t mi355q_w4a16_mm_add_rms_norm_workspace_bytes(int M, int N, int K, int G, int tile, int has_zc) {
  if (G <= 0) G = K;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  RowsPlan rp;
  if (rows_route(M, N, K, G, tile, has_zc != 0, &rp)) return (size_t)rp.splits * M * N * sizeof(float);
  const MmPlan mp = plan_mm(M, N, K, G, tile);
  return mp.ok ? mp.ws : 0;
}

/* c = a W (+ bias), then fused_add_rms_norm(c, residual): residual = round(c + residual),
 * out = round(residual * rsqrt(mean(residual^2) + eps) * norm_weight).  With K-slices the slice sums,
 * the residual add and the norm are one kernel over the fp32 partial tiles. */
extern "C" int mi355q_w4a16_mm_add_rms_norm(const void* a, int lda, const void* packed, const void* scales,
                                            const void* zc, const void* bias, void* residual,
                                            const void* norm_weight, float epsilon, void* out, int M, int N,
                                            int K, int G, int tile, int act_dtype, void* workspace,
                                            size_t workspace_bytes, mi355q_stream_t stream) {
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_tile(tile, K, G, N)) return e;
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w4a16_mm_add_rms_norm: act_dtype must be bf16 or fp16");
  if (M == 0) return MI355Q_OK;
  MI355Q_REQUIRE(a && packed && scales && residual && out && M > 0, "w4a16_mm_add_rms_norm: bad arguments");
  MI355Q_REQUIRE(lda % 8 == 0 && lda >= K, "w4a16_mm_add_rms_norm: row stride %d must be a multiple of 8 and cover the row", lda);
  MI355Q_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)scales & 15) == 0 &&
                     ((uintptr_t)out & 15) == 0 && ((uintptr_t)residual & 15) == 0 && ((uintptr_t)zc & 15) == 0 &&
                     ((uintptr_t)bias & 7) == 0 && ((uintptr_t)norm_weight & 15) == 0,
                 "w4a16_mm_add_rms_norm: pointers must be 16-byte aligned");
  RowsPlan rp;
  if (rows_route(M, N, K, G > 0 ? G : K, tile, zc != nullptr, &rp)) {
    const size_t need = (size_t)rp.splits * M * N * sizeof(float);
    MI355Q_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "w4a16_mm_add_rms_norm: K-slices need a 16-byte aligned workspace");
    if (workspace_bytes < need) return set_error(MI355Q_ENOSPACE, "w4a16_mm_add_rms_norm: workspace %zu B < %zu B", workspace_bytes, need);
    if (G <= 0) G = K;
    RowsParams r;
    r.x = (const uint16_t*)a; r.ldx = lda; r.wp = (const u32x4*)packed;
    r.scales = (const uint16_t*)scales; r.zc = (const uint16_t*)zc; r.bias = nullptr;
    r.out = (uint16_t*)out; r.counters = nullptr; r.partial = (float*)workspace;
    r.M = M; r.N = N; r.K = K; r.chunks = K / 128; r.gchunks = G / 128;
    r.splits = rp.splits; r.cps = r.chunks / rp.splits; r.nw = rp.nw; r.bursts = rp.bursts; r.rows_pp = rp.rows_pp;
    r.partial_only = 1;
    const int e = act_dtype == MI355Q_BF16 ? launch_rows<BF16>(r, rp.cpw, rp.lds, (hipStream_t)stream)
                                           : launch_rows<FP16>(r, rp.cpw, rp.lds, (hipStream_t)stream);
    if (e) return e;
    return mi355q_splitk_add_rms_norm(out, residual, r.partial, rp.splits, bias, norm_weight, epsilon, M, N, act_dtype, stream);
  }
  MmParams p{};
  p.x = (const uint16_t*)a; p.ldx = lda; p.wp = (const u32x4*)packed;
  p.scales = (const uint16_t*)scales; p.zc = (const uint16_t*)zc; p.bias = (const uint16_t*)bias;
  p.out = (uint16_t*)out; p.partial = nullptr;
  p.M = M; p.N = N; p.K = K; p.chunks = K / 128; p.gchunks = G / 128;
  p.epi_silu = 0;
  if (const StreamRowsPlan sp = plan_stream_rows(M, N, K, G, tile); sp.ok) {
    p.splits = 1; p.cps = p.chunks;
    const int e = act_dtype == MI355Q_BF16 ? launch_stream_rows<BF16>(p, sp, (hipStream_t)stream)
                                           : launch_stream_rows<FP16>(p, sp, (hipStream_t)stream);
    if (e) return e;
    return mi355q_fused_add_rms_norm(out, N, residual, norm_weight, epsilon, M, N, act_dtype, stream);
  }
  const MmPlan mp = plan_mm(M, N, K, G, tile);
  MI355Q_REQUIRE(mp.ok, "w4a16_mm_add_rms_norm: needs a group size that is a multiple of 128 and N a multiple of %d for strip width %d",
                 tile == 112 ? 448 : 256, tile);
  p.splits = mp.splits; p.cps = p.chunks / mp.splits;
  if (mp.splits > 1) {
    MI355Q_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "w4a16_mm_add_rms_norm: K-slices need a 16-byte aligned workspace");
    if (workspace_bytes < mp.ws) return set_error(MI355Q_ENOSPACE, "w4a16_mm_add_rms_norm: workspace %zu B < %zu B", workspace_bytes, mp.ws);
    p.partial = (float*)workspace;
  }
  const int e = act_dtype == MI355Q_BF16 ? launch_mm<BF16>(p, mp, (hipStream_t)stream, false)
                                         : launch_mm<FP16>(p, mp, (hipStream_t)stream, false);
  if (e) return e;
  if (mp.splits > 1)
    return mi355q_splitk_add_rms_norm(out, residual, p.partial, mp.splits, bias, norm_weight, epsilon, M, N, act_dtype, stream);
  return mi355q_fused_add_rms_norm(out, N, residual, norm_weight, epsilon, M, N, act_dtype, stream);
}

/* The fused qkv projection of prompt tokens: c = a W (+ bias), NeoX RoPE on the q and k heads of c (in
 * place), rotated k and v written to the paged cache.  With K-slices, the slice sums, the rotation and
 * the cache write are one kernel; otherwise (or for GPT-J pairing / partial rotary) the plain sequence
 * mi355q_w4a16_mm -> mi355q_rope_and_cache runs. */
extern "C" int mi355q_w4a16_mm_rope_and_cache(const void* a, int lda, const void* packed, const void* scales,
                                              const void* zc, const void* bias, void* qkv, int M, int N, int K,
                                              int G, int tile, int act_dtype, void* workspace,
                                              size_t workspace_bytes, const int64_t* positions, int head_size,
                                              int num_heads, int num_kv_heads, const void* cos_sin_cache,
                                              int rot_dim, int cache_dtype, int is_neox, void* key_cache,
                                              void* value_cache, const int64_t* slot_mapping, int block_size,
                                              int64_t block_stride, int kv_fp8, const float* k_scale,
                                              const float* v_scale, mi355q_stream_t stream) {
  MI355Q_REQUIRE(head_size > 0 && num_heads > 0 && num_kv_heads > 0 && N == (num_heads + 2 * num_kv_heads) * head_size,
                 "w4a16_mm_rope_and_cache: N = %d is not (heads + 2 kv_heads) * head_size", N);
  if (G <= 0) G = K;
  const MmPlan mp = plan_mm(M > 0 ? M : 1, N, K, G, tile);
  const bool fused = mp.ok && mp.splits > 1 && is_neox && rot_dim == head_size && head_size % 16 == 0;
  if (!fused) {
    if (int e = mi355q_w4a16_mm(a, lda, packed, scales, zc, bias, qkv, M, N, K, G, tile, act_dtype, 0, workspace,
                                workspace_bytes, stream)) return e;
    uint16_t* q = (uint16_t*)qkv;
    return mi355q_rope_and_cache(positions, q, N, q + (size_t)num_heads * head_size, N,
                                 q + (size_t)(num_heads + num_kv_heads) * head_size, N, head_size, num_heads,
                                 num_kv_heads, cos_sin_cache, rot_dim, cache_dtype, is_neox, key_cache, value_cache,
                                 slot_mapping, M, block_size, block_stride, act_dtype, kv_fp8, k_scale, v_scale, stream);
  }
  if (int e = check_shape(K, N, G)) return e;
  if (int e = check_tile(tile, K, G, N)) return e;
  MI355Q_REQUIRE(act_dtype == MI355Q_BF16 || act_dtype == MI355Q_FP16, "w4a16_mm_rope_and_cache: act_dtype must be bf16 or fp16");
  if (M == 0) return MI355Q_OK;
  MI355Q_REQUIRE(a && packed && scales && qkv && M > 0, "w4a16_mm_rope_and_cache: bad arguments");
  MI355Q_REQUIRE(lda % 8 == 0 && lda >= K, "w4a16_mm_rope_and_cache: row stride %d must be a multiple of 8 and cover the row", lda);
  MI355Q_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)scales & 15) == 0 &&
                     ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)zc & 15) == 0 && ((uintptr_t)bias & 7) == 0,
                 "w4a16_mm_rope_and_cache: pointers must be 16-byte aligned");
  MI355Q_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "w4a16_mm_rope_and_cache: K-slices need a 16-byte aligned workspace");
  if (workspace_bytes < mp.ws) return set_error(MI355Q_ENOSPACE, "w4a16_mm_rope_and_cache: workspace %zu B < %zu B", workspace_bytes, mp.ws);
  MmParams p{};
  p.x = (const uint16_t*)a; p.ldx = lda; p.wp = (const u32x4*)packed;
  p.scales = (const uint16_t*)scales; p.zc = (const uint16_t*)zc; p.bias = (const uint16_t*)bias;
  p.out = (uint16_t*)qkv; p.partial = (float*)workspace;
  p.M = M; p.N = N; p.K = K; p.chunks = K / 128; p.gchunks = G / 128;
  p.splits = mp.splits; p.cps = p.chunks / mp.splits; p.epi_silu = 0;
  const int e = act_dtype == MI355Q_BF16 ? launch_mm<BF16>(p, mp, (hipStream_t)stream, false)
                                         : launch_mm<FP16>(p, mp, (hipStream_t)stream, false);
  if (e) return e;
  return mi355q_splitk_rope_and_cache(p.partial, mp.splits, bias, qkv, positions, head_size, num_heads, num_kv_heads,
                                      cos_sin_cache, cache_dtype, key_cache, value_cache, slot_mapping, M, block_size,
                                      block_stride, act_dtype, kv_fp8, k_scale, v_scale, stream);
}
