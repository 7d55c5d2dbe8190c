// Paged attention over the ROCM_ATTN cache layout for gfx950.
//   K cache [blocks, kv_heads, head/x, block_size, x]  (x = 8 for 16-bit, 16 for fp8)
//   V cache [blocks, kv_heads, head, block_size]
// One workgroup (4 waves) handles one (query block, 256-token partition, kv head); each
// wave owns 64 tokens of the partition.  The 16 columns of every MFMA tile are
// (query token, query head of the kv head's group) pairs, so the same kernel serves
// decode (one query token) and prompt rows (several query tokens with per-column causal
// limits) reading K/V back from the cache.
//
//   S^T = K * Q^T      v_mfma_f32_16x16x32: rows = tokens, columns = query columns
//   O^T = V^T * P^T    rows = head dims, columns = query columns; the k index of this
//                      product is permuted so that each lane's P fragment is exactly the
//                      S^T registers it already holds (no cross-lane traffic)
// Softmax statistics are combined across waves through LDS; per-partition results go to
// tmp_out / exp_sums / max_logits and a second kernel merges partitions.
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "common.h"

// No fused multiply-add unless the source asks for one: the kernels below repeat the same fp32 expressions (one
// workgroup per item, workgroups that walk several items, fused and unfused decode steps, in-kernel and separate
// partition merges) and promise the same bits.  Left to itself the compiler contracts a * b + c in one of them
// and not in another, depending on what surrounds the expression, and the results then differ in the last bit
// now and then (seen: one output element in tens of thousands).
#pragma clang fp contract(off)

namespace mi355q {

struct PAParams {
  void* out;
  float* exp_sums;
  float* max_logits;
  void* tmp_out;
  const uint16_t* q;
  int64_t q_stride;
  const void* kc;
  const void* vc;
  int64_t kv_block_stride, kv_head_stride;
  int num_seqs, num_heads, num_kv_heads;
  float scale;
  const int32_t* block_tables;
  int max_blocks;
  const int32_t* seq_lens;
  const int32_t* qsl;  // query_start_loc or null
  int bs;
  int max_parts;
  const float* alibi;
  const float* k_scale;
  const float* v_scale;
  int varlen;          // 0: decode rows only, 1: every row, several query tokens per block
  int qblocks_per_seq; // varlen: ceil(max_query_len / QB)
  int gp;              // columns per query token (power of two >= group size)
  int g;               // query heads per kv head
  int bs_shift;        // log2(bs) when the block size is a power of two, else -1
  // fused decode step (FUSED kernels): q points into the qkv projection output, the new
  // token's k/v sit k_off / v_off elements after the row start; NeoX RoPE over the whole head
  const int64_t* positions;
  const uint16_t* cos_sin;  // [max_pos, HEAD] in the activation dtype
  unsigned long long* stamps;  // diagnostics: 10 timestamps per workgroup (8 at 100 MHz, 2 shader-clock), or null
  int k_off, v_off;
  int inline_reduce;  // 1: the workgroup that finishes a query block's last partition merges the partitions
  int cnt_bank;       // bank of arrival counters this launch's stream owns (inline_reduce)
};

// Arrival counters of the in-kernel partition merge, one per (query block, kv head); library-owned,
// zero between launches (the last arriver resets its counter).  Every stream that launches attention
// gets a bank of its own (stream_bank), so launches that overlap on different streams do not meet on a
// counter; a process with more streams than banks, or a grid larger than a bank, takes the separate
// merge launch instead.  mi355q_reset_library_state zeroes them after an aborted launch.
//
// Hand-off form (gfx950): the partial tiles and statistics are stored write-through (agent-scope
// atomic stores = `sc1`), every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup
// meets at a barrier, ONE lane adds to the counter with an agent-scope atomic, and the workgroup whose
// add returned the last ticket reads every handed-off byte with agent-scope (`sc1`) loads after a
// barrier behind that add -- stores, signal and loads all bypass the non-coherent L1 / L2 levels, so the
// producers need no release fence; the last arriver still runs one agent-scope acquire before its loads.
constexpr int PA_BANK = 16384;
static __device__ int g_pa_cnt[MI355Q_STATE_BANKS * PA_BANK];

__global__ void pa_reset_kernel() {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < MI355Q_STATE_BANKS * PA_BANK; i += gridDim.x * blockDim.x)
    g_pa_cnt[i] = 0;
}
int reset_attention_state(hipStream_t stream) {
  hipLaunchKernelGGL(pa_reset_kernel, dim3(256), dim3(256), 0, stream);
  return check_launch("pa_reset_kernel");
}

// four e4m3 bytes -> four 16-bit values.  Every e4m3 value (subnormals included) is a bf16 and an fp16 value, so the
// conversion is exact; gfx950 converts a pair straight to the 16-bit type (v_cvt_scalef32_pk_{bf16,f16}_fp8 with a
// unit scale): two instructions per word instead of two conversions to fp32 and the repacking of four words.
typedef __attribute__((ext_vector_type(2))) __bf16 pa_bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 pa_f16x2;
template <typename DT>
__device__ inline u32x2 fp8x4_to_h16(uint32_t v) {
  u32x2 r;
  if constexpr (DT::id == MI355Q_BF16) {
    r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, false));
    r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, true));
  } else {
    r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v, 1.0f, false));
    r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v, 1.0f, true));
  }
  return r;
}

// rotate 8 (x, y) pairs held as packed 16-bit values: x' = x c - y s, y' = y c + x s,
// fp32 arithmetic, one rounding (csrc/libtorch_stable/pos_encoding_kernels.cu:29-32 of the reference)
template <typename DT>
__device__ inline void rope8(u32x4& x, u32x4& y, const u32x4 c, const u32x4 s) {
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const float x0 = DT::to_f32((uint16_t)(x[e] & 0xffff)), x1 = DT::to_f32((uint16_t)(x[e] >> 16));
    const float y0 = DT::to_f32((uint16_t)(y[e] & 0xffff)), y1 = DT::to_f32((uint16_t)(y[e] >> 16));
    const float c0 = DT::to_f32((uint16_t)(c[e] & 0xffff)), c1 = DT::to_f32((uint16_t)(c[e] >> 16));
    const float s0 = DT::to_f32((uint16_t)(s[e] & 0xffff)), s1 = DT::to_f32((uint16_t)(s[e] >> 16));
    x[e] = pack2<DT>(x0 * c0 - y0 * s0, x1 * c1 - y1 * s1);
    y[e] = pack2<DT>(y0 * c0 + x0 * s0, y1 * c1 + x1 * s1);
  }
}

#define PA_STAMP(i)                                                                              \
  if (p.stamps && threadIdx.x == 0)                                                              \
    p.stamps[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * 10 + (i)] = ((i) >= 9 ? __builtin_amdgcn_s_memtime() : __builtin_amdgcn_s_memrealtime())

static unsigned long long* g_pa_stamps = nullptr;

// The variant that serves prompt rows asks for three waves per SIMD (168 registers, a handful of spills):
// its workgroups are bound by the latency of their own K/V reads, so one more resident workgroup per CU
// is worth more than the spills cost (TTFT at 2048 tokens 40.4 -> 38.8 ms; four waves spill 79 and lose).
// The fused decode step with an fp8 cache fits 160 registers without spilling and takes the same hint;
// with a 16-bit cache it would spill 48 and lose 4 % of a decode step, so it keeps its 224 registers.
template <typename DT, int HEAD, bool KVFP8, bool FUSED>
__global__ __launch_bounds__(256, (FUSED && !KVFP8) ? 1 : 3) void paged_attention_kernel(const PAParams p) {
  __shared__ float s_max[4][16];
  __shared__ float s_sum[4][16];
  __shared__ __attribute__((aligned(16))) float s_out[4][16][HEAD];
  // fused decode step: the new token's rotated key and its value, as the cache will hold them
  __shared__ __attribute__((aligned(16))) uint16_t s_knew[FUSED ? HEAD : 8];
  __shared__ __attribute__((aligned(16))) uint16_t s_vnew[FUSED ? HEAD : 8];
  __shared__ int s_last;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = lane >> 4, col = lane & 15;
  const int g = p.num_heads / p.num_kv_heads;
  const int gp = p.gp, QB = 16 / gp;
  PA_STAMP(0);
  PA_STAMP(8);

  int seq, q0, qlen, qstart;
  if constexpr (FUSED) {   // the fused decode step only exists for one new token per sequence
    seq = blockIdx.x;
    q0 = 0;
    qlen = 1;
    qstart = seq;
  } else if (p.varlen) {
    seq = blockIdx.x / p.qblocks_per_seq;
    q0 = (blockIdx.x % p.qblocks_per_seq) * QB;
    qstart = p.qsl[seq];
    qlen = p.qsl[seq + 1] - qstart;
    if (q0 >= qlen) return;
  } else {
    seq = blockIdx.x;
    q0 = 0;
    qlen = 1;
    qstart = seq;
    if (p.qsl) {
      qstart = p.qsl[seq];
      if (p.qsl[seq + 1] - qstart != 1) return;  // prompt rows are not handled here
    }
  }
  // The kernel is latency-bound at decode sizes, so its memory requests are arranged in two dependent
  // groups and nothing else: (1) everything addressed by the block index alone -- sequence length,
  // position, this wave's block-table entries, the query fragments, the new token's key and value --
  // and (2) everything addressed by what group 1 returned -- cos/sin of the position, the K and V
  // tiles.  Every request is unconditional (addresses are clamped into valid memory and the results
  // masked afterwards): a load under a lane-dependent branch makes the compiler drain the whole
  // vector-memory queue at the join, which turns each such branch into one more round trip.
  constexpr int DS = HEAD / 32;
  const bool plain = FUSED || (!p.varlen && !p.qsl);
  int phys_pre[4] = {0, 0, 0, 0};
  u32x4 qf[DS];
  if (plain) {
    const int32_t* tb = p.block_tables + (size_t)seq * p.max_blocks;
#pragma unroll
    for (int T = 0; T < 4; T++) {
      const int t0 = blockIdx.y * MI355Q_PA_PARTITION + wave * 64 + T * 16;
      phys_pre[T] = tb[min(t0 / p.bs, p.max_blocks - 1)];
    }
    const int hd = blockIdx.z * g + min((col % gp), g - 1);
    const uint16_t* qp0 = p.q + (size_t)seq * p.q_stride + (size_t)hd * HEAD;
#pragma unroll
    for (int ds = 0; ds < DS; ds++) {
      const int d0 = KVFP8 ? 16 * (4 * (ds >> 1) + kq) + 8 * (ds & 1) : 32 * ds + 8 * kq;
      qf[ds] = *reinterpret_cast<const u32x4*>(qp0 + d0);
    }
  }
  // new token: thread t < HEAD/2 rotates key pair (t, t + HEAD/2), thread 128 + d carries value d; the
  // other threads read the same few lines again and ignore them
  uint16_t nk_x = 0, nk_y = 0, nk_c = 0, nk_s = 0, nv = 0;
  if constexpr (FUSED) {
    const uint16_t* rowp = p.q + (size_t)seq * p.q_stride;
    const uint16_t* kp = rowp + p.k_off + blockIdx.z * HEAD;
    const int tk = threadIdx.x & (HEAD / 2 - 1);
    nk_x = kp[tk];
    nk_y = kp[tk + HEAD / 2];
    nv = rowp[p.v_off + blockIdx.z * HEAD + (threadIdx.x & (HEAD - 1))];
  }
  int64_t pos = FUSED ? p.positions[seq] : 0;
  const float kscale = KVFP8 ? *p.k_scale : 1.0f, vscale = KVFP8 ? *p.v_scale : 1.0f;  // requested with the rest
  int seq_len = p.seq_lens[seq];
  // Pin the first group in front of the early exit below.  Left alone, the compiler sinks every load
  // whose result the exit path does not need behind that branch, i.e. behind the round trip of the
  // sequence length.  Touching one register of each load here costs nothing: they were requested
  // together and return together.
  if (plain) {
    uint32_t t0 = nk_x, t1 = nk_y, t2 = nv;
    int64_t ps = pos;
    asm volatile("" : "+v"(seq_len), "+v"(ps), "+v"(phys_pre[0]), "+v"(phys_pre[1]), "+v"(phys_pre[2]), "+v"(phys_pre[3]),
                      "+v"(qf[0][0]), "+v"(qf[DS / 2][0]), "+v"(qf[DS - 1][0]), "+v"(t0), "+v"(t1), "+v"(t2));
    nk_x = (uint16_t)t0; nk_y = (uint16_t)t1; nv = (uint16_t)t2;
    pos = ps;
  }
  const int q_in_blk = min(QB, qlen - q0);
  const int blk_ctx = seq_len - qlen + q0 + q_in_blk;  // tokens visible to the last column
  const int part = blockIdx.y;
  const int p0 = part * MI355Q_PA_PARTITION;
  if (p0 >= blk_ctx) return;
  PA_STAMP(6);
  const int kvh = blockIdx.z;

  // this lane's query column
  const int qi = col / gp, hi = col % gp;
  const bool col_valid = hi < g && qi < q_in_blk;
  const int ctx_c = col_valid ? seq_len - qlen + q0 + qi + 1 : 0;
  const int qrow = qstart + q0 + min(qi, q_in_blk - 1);
  const int head = kvh * g + min(hi, g - 1);

  const float scale = p.scale * kscale;
  const float slope = p.alibi ? p.alibi[head] : 0.f;
  const int32_t* table = p.block_tables + (size_t)seq * p.max_blocks;
  const int wave_t0 = p0 + wave * 64;

  // ---- Q fragments (B operand): column = this lane's query column --------------------
  // 16-bit cache: k index of step ds is d = 32 ds + 8 kq + j
  // fp8 cache   : one 16-byte K load covers d = 16 (4 (ds/2) + kq) + 8 (ds&1) + j
  if (!plain) {
    const uint16_t* qp = p.q + (size_t)qrow * p.q_stride + (size_t)head * HEAD;
#pragma unroll
    for (int ds = 0; ds < DS; ds++) {
      const int d0 = KVFP8 ? 16 * (4 * (ds >> 1) + kq) + 8 * (ds & 1) : 32 * ds + 8 * kq;
      qf[ds] = *reinterpret_cast<const u32x4*>(qp + d0);
    }
  }

  // ---- second group: cos/sin of the position, the cache block of the new token, K and V tiles ------
  u32x4 cfrag[FUSED ? DS / 2 : 1], sfrag[FUSED ? DS / 2 : 1];
  const int tnew = seq_len - 1;
  const bool owns_new = FUSED && part == tnew / MI355Q_PA_PARTITION;
  int blk_new = 0;
  if constexpr (FUSED) {
    const uint16_t* cs = p.cos_sin + (size_t)pos * HEAD;
#pragma unroll
    for (int ds = 0; ds < DS / 2; ds++) {
      const int d0 = KVFP8 ? 16 * (4 * (ds >> 1) + kq) + 8 * (ds & 1) : 32 * ds + 8 * kq;
      cfrag[ds] = *reinterpret_cast<const u32x4*>(cs + d0);
      sfrag[ds] = *reinterpret_cast<const u32x4*>(cs + HEAD / 2 + d0);
    }
    const int tk = threadIdx.x & (HEAD / 2 - 1);
    nk_c = cs[tk];
    nk_s = cs[HEAD / 2 + tk];
    blk_new = table[min(tnew / p.bs, p.max_blocks - 1)];
  }

  // K: A operand rows = tokens; V^T: rows = head dims, k = tokens in the permuted order of the
  // P fragment (element j <-> token 4 kq + j of the even tile, j >= 4: of the odd tile).
  // A tile past the visible context reads cache block 0 (always mapped) and is masked out below:
  // block-table entries out there are not the caller's to keep valid.
  constexpr int DTILES = HEAD / 16;
  constexpr int KR = KVFP8 ? DS / 2 : DS;
  int phys[4];
  u32x4 kreg[4][KR];
  u32x2 vreg[4][DTILES];  // 16-bit cache: 4 tokens x 16 bit; fp8: 4 tokens in [0], [1] unused
#pragma unroll
  for (int T = 0; T < 4; T++) {
    const int t0 = wave_t0 + T * 16;
    const int entry = plain ? phys_pre[T] : table[min(t0 / p.bs, p.max_blocks - 1)];
    phys[T] = t0 < blk_ctx ? entry : -1;
  }
#pragma unroll
  for (int T = 0; T < 4; T++) {
    const int blk = max(phys[T], 0);
    const int t0 = wave_t0 + T * 16;
    const int off = t0 % p.bs + col;  // this lane's token row of the tile
    if constexpr (!KVFP8) {
      const uint16_t* kb = reinterpret_cast<const uint16_t*>(p.kc) +
                           (size_t)blk * p.kv_block_stride + (size_t)kvh * p.kv_head_stride;
#pragma unroll
      for (int ds = 0; ds < DS; ds++)
        kreg[T][ds] = *reinterpret_cast<const u32x4*>(kb + ((size_t)(ds * 4 + kq) * p.bs + off) * 8);
    } else {
      const uint8_t* kb = reinterpret_cast<const uint8_t*>(p.kc) +
                          (size_t)blk * p.kv_block_stride + (size_t)kvh * p.kv_head_stride;
#pragma unroll
      for (int sp = 0; sp < DS / 2; sp++)
        kreg[T][sp] = *reinterpret_cast<const u32x4*>(kb + ((size_t)(sp * 4 + kq) * p.bs + off) * 16);
    }
  }
#pragma unroll
  for (int T = 0; T < 4; T++) {
    const int blk = max(phys[T], 0);
    const int offv = (wave_t0 + T * 16) % p.bs + 4 * kq;
    if constexpr (!KVFP8) {
      const uint16_t* vb = reinterpret_cast<const uint16_t*>(p.vc) +
                           (size_t)blk * p.kv_block_stride + (size_t)kvh * p.kv_head_stride;
#pragma unroll
      for (int dt = 0; dt < DTILES; dt++)
        vreg[T][dt] = *reinterpret_cast<const u32x2*>(vb + (size_t)(dt * 16 + col) * p.bs + offv);
    } else {
      const uint8_t* vb = reinterpret_cast<const uint8_t*>(p.vc) +
                          (size_t)blk * p.kv_block_stride + (size_t)kvh * p.kv_head_stride;
#pragma unroll
      for (int dt = 0; dt < DTILES; dt++) {
        vreg[T][dt][0] = *reinterpret_cast<const uint32_t*>(vb + (size_t)(dt * 16 + col) * p.bs + offv);
        vreg[T][dt][1] = 0u;
      }
    }
  }

  if constexpr (FUSED) {
    // cos/sin of the new key are first touched HERE, after every cache read has been requested:
    // otherwise the compiler converts them right behind their loads and the wait for that round
    // trip lands in front of the K/V requests (loads return in order, so waiting for these older
    // ones here leaves the K/V reads in flight).
    {
      uint32_t t2 = nk_c, t3 = nk_s;
      asm volatile("" : "+v"(t2), "+v"(t3));
      nk_c = (uint16_t)t2; nk_s = (uint16_t)t3;
    }
    // RoPE on the query fragments: the partner d + HEAD/2 of every element sits in fragment
    // ds + DS/2 of the same lane
#pragma unroll
    for (int ds = 0; ds < DS / 2; ds++) rope8<DT>(qf[ds], qf[ds + DS / 2], cfrag[ds], sfrag[ds]);
    PA_STAMP(7);
    // The workgroup whose partition holds the newest position rotates the new key, sends key and
    // value to the cache (nobody waits for those stores: this step's reads of that slot are
    // replaced below) and shares them through LDS in the form the cache holds them.
    if (owns_new) {
      const int blk = blk_new, off = tnew % p.bs;
      constexpr int X = KVFP8 ? 16 : 8;
      const size_t base = (size_t)blk * p.kv_block_stride + (size_t)kvh * p.kv_head_stride;
      const int t = threadIdx.x;
      if (t < HEAD / 2) {
        const float x = DT::to_f32(nk_x), y = DT::to_f32(nk_y);
        const float c = DT::to_f32(nk_c), sn = DT::to_f32(nk_s);
        const uint16_t xr = DT::from_f32(x * c - y * sn), yr = DT::from_f32(y * c + x * sn);
        const int d1 = t, d2 = t + HEAD / 2;
        const size_t i1 = base + ((size_t)(d1 / X) * p.bs + off) * X + d1 % X;
        const size_t i2 = base + ((size_t)(d2 / X) * p.bs + off) * X + d2 % X;
        if constexpr (KVFP8) {
          const float ks = kscale;
          const uint8_t b1 = f32_to_fp8(DT::to_f32(xr) / ks), b2 = f32_to_fp8(DT::to_f32(yr) / ks);
          reinterpret_cast<uint8_t*>(const_cast<void*>(p.kc))[i1] = b1;
          reinterpret_cast<uint8_t*>(const_cast<void*>(p.kc))[i2] = b2;
          reinterpret_cast<uint8_t*>(s_knew)[d1] = b1;
          reinterpret_cast<uint8_t*>(s_knew)[d2] = b2;
        } else {
          reinterpret_cast<uint16_t*>(const_cast<void*>(p.kc))[i1] = xr;
          reinterpret_cast<uint16_t*>(const_cast<void*>(p.kc))[i2] = yr;
          s_knew[d1] = xr;
          s_knew[d2] = yr;
        }
      } else if (t >= 128 && t < 128 + HEAD) {
        const int d = t - 128;
        const size_t vi = base + (size_t)d * p.bs + off;
        if constexpr (KVFP8) {
          const uint8_t b = f32_to_fp8(DT::to_f32(nv) / vscale);
          reinterpret_cast<uint8_t*>(const_cast<void*>(p.vc))[vi] = b;
          reinterpret_cast<uint8_t*>(s_vnew)[d] = b;
        } else {
          reinterpret_cast<uint16_t*>(const_cast<void*>(p.vc))[vi] = nv;
          s_vnew[d] = nv;
        }
      }
      __syncthreads();
      // patch the fragments of the tile that holds the new position (it belongs to one wave)
      const int tl = tnew - wave_t0;
      if (tl >= 0 && tl < 64) {
#pragma unroll
        for (int T = 0; T < 4; T++) {
          if ((tl >> 4) == T) {
            if (col == (tl & 15)) {
              if constexpr (!KVFP8) {
#pragma unroll
                for (int ds = 0; ds < DS; ds++) kreg[T][ds] = *reinterpret_cast<const u32x4*>(s_knew + 32 * ds + 8 * kq);
              } else {
#pragma unroll
                for (int sp = 0; sp < DS / 2; sp++)
                  kreg[T][sp] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(s_knew) + 16 * (sp * 4 + kq));
              }
            }
            if (kq == ((tl & 15) >> 2)) {
              const int j = tl & 3;
#pragma unroll
              for (int dt = 0; dt < DTILES; dt++) {
                if constexpr (!KVFP8) {
                  const uint32_t val = s_vnew[dt * 16 + col];
                  const uint32_t old = (j >> 1) ? vreg[T][dt][1] : vreg[T][dt][0];
                  const uint32_t upd = (j & 1) ? ((old & 0x0000ffffu) | (val << 16)) : ((old & 0xffff0000u) | val);
                  if (j >> 1) vreg[T][dt][1] = upd; else vreg[T][dt][0] = upd;
                } else {
                  const uint32_t val = reinterpret_cast<const uint8_t*>(s_vnew)[dt * 16 + col];
                  vreg[T][dt][0] = (vreg[T][dt][0] & ~(0xffu << (8 * j))) | (val << (8 * j));
                }
              }
            }
          }
        }
      }
    }
  }

  PA_STAMP(1);
  // ---- S^T = K Q^T ---------------------------------------------------------------------------
  f32x4 st[4];
#pragma unroll
  for (int T = 0; T < 4; T++) {
    st[T] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (phys[T] >= 0) {
      if constexpr (!KVFP8) {
#pragma unroll
        for (int ds = 0; ds < DS; ds++) st[T] = mfma16<DT>(kreg[T][ds], qf[ds], st[T]);
      } else {
#pragma unroll
        for (int sp = 0; sp < DS / 2; sp++) {
          const u32x4 raw = kreg[T][sp];
          const u32x2 a0 = fp8x4_to_h16<DT>(raw[0]), a1 = fp8x4_to_h16<DT>(raw[1]);
          const u32x2 b0 = fp8x4_to_h16<DT>(raw[2]), b1 = fp8x4_to_h16<DT>(raw[3]);
          st[T] = mfma16<DT>(u32x4{a0[0], a0[1], a1[0], a1[1]}, qf[2 * sp], st[T]);
          st[T] = mfma16<DT>(u32x4{b0[0], b0[1], b1[0], b1[1]}, qf[2 * sp + 1], st[T]);
        }
      }
    }
  }

  PA_STAMP(2);
  // ---- logits, partition max / sum per column ------------------------------------------
  // Each wave normalises against its own maximum; one exchange of (max, sum) per wave through
  // LDS then gives the partition maximum, the rescaling factor of this wave and the total.
  float mw = -INFINITY;
#pragma unroll
  for (int T = 0; T < 4; T++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int tok = wave_t0 + T * 16 + 4 * kq + r;
      float v = st[T][r] * scale + slope * (float)(tok - ctx_c + 1);
      v = tok < ctx_c ? v : -INFINITY;
      st[T][r] = v;
      mw = fmaxf(mw, v);
    }
  mw = fmaxf(mw, __shfl_xor(mw, 16, 64));
  mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
  const float mwref = mw == -INFINITY ? 0.f : mw;
  float sw = 0.f;
#pragma unroll
  for (int T = 0; T < 4; T++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float e = __expf(st[T][r] - mwref);
      st[T][r] = e;
      sw += e;
    }
  sw += __shfl_xor(sw, 16, 64);
  sw += __shfl_xor(sw, 32, 64);
  if (kq == 0) {
    s_max[wave][col] = mw;
    s_sum[wave][col] = sw;
  }
  __syncthreads();
  float mx = fmaxf(fmaxf(s_max[0][col], s_max[1][col]), fmaxf(s_max[2][col], s_max[3][col]));
  const float mref = mx == -INFINITY ? 0.f : mx;
  float sum = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < 4; w2++) {
    const float m2 = s_max[w2][col];
    sum += m2 == -INFINITY ? 0.f : s_sum[w2][col] * __expf(m2 - mref);
  }
  // this wave's probabilities relative to the partition maximum, normalised
  const float inv = (mw == -INFINITY ? 0.f : __expf(mwref - mref)) / (sum + 1e-6f);

  PA_STAMP(3);
  // ---- O^T += V^T P^T ---------------------------------------------------------------------
  f32x4 o[DTILES];
#pragma unroll
  for (int dt = 0; dt < DTILES; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s2 = 0; s2 < 2; s2++) {
    const int Te = 2 * s2, To = 2 * s2 + 1;
    if (phys[Te] >= 0) {  // wave-uniform: tiles are visited in token order
      u32x4 pf;
      pf[0] = pack2<DT>(st[Te][0] * inv, st[Te][1] * inv);
      pf[1] = pack2<DT>(st[Te][2] * inv, st[Te][3] * inv);
      pf[2] = pack2<DT>(st[To][0] * inv, st[To][1] * inv);
      pf[3] = pack2<DT>(st[To][2] * inv, st[To][3] * inv);
      // tokens at or past seq_len hold unwritten cache memory: force their V to zero
      const int te0 = wave_t0 + Te * 16 + 4 * kq, to0 = wave_t0 + To * 16 + 4 * kq;
      const int ne = min(max(seq_len - te0, 0), 4), no = min(max(seq_len - to0, 0), 4);
      const uint32_t me0 = ne >= 2 ? 0xffffffffu : (ne == 1 ? 0x0000ffffu : 0u);
      const uint32_t me1 = ne >= 4 ? 0xffffffffu : (ne == 3 ? 0x0000ffffu : 0u);
      const uint32_t mo0 = no >= 2 ? 0xffffffffu : (no == 1 ? 0x0000ffffu : 0u);
      const uint32_t mo1 = no >= 4 ? 0xffffffffu : (no == 3 ? 0x0000ffffu : 0u);
      const bool odd_ok = phys[To] >= 0;
#pragma unroll
      for (int dt = 0; dt < DTILES; dt++) {
        u32x2 ve, vo = u32x2{0u, 0u};
        if constexpr (!KVFP8) {
          ve = vreg[Te][dt];
          if (odd_ok) vo = vreg[To][dt];
        } else {
          ve = fp8x4_to_h16<DT>(vreg[Te][dt][0]);
          if (odd_ok) vo = fp8x4_to_h16<DT>(vreg[To][dt][0]);
        }
        const u32x4 vf = u32x4{ve[0] & me0, ve[1] & me1, vo[0] & mo0, vo[1] & mo1};
        o[dt] = mfma16<DT>(vf, pf, o[dt]);
      }
    }
  }

  PA_STAMP(4);
  // ---- cross-wave sum of O, store ---------------------------------------------------------
  const float vs = vscale;
#pragma unroll
  for (int dt = 0; dt < DTILES; dt++)
    *reinterpret_cast<f32x4*>(&s_out[wave][col][dt * 16 + 4 * kq]) = o[dt] * vs;
  __syncthreads();

  // one partition covers the whole context of this query block: the result is final
  const int nactive = (blk_ctx + MI355Q_PA_PARTITION - 1) / MI355Q_PA_PARTITION;
  const bool single = p.max_parts == 1 || (p.inline_reduce && nactive == 1);
  typedef unsigned long long u64;
  for (int idx = threadIdx.x; idx < 16 * HEAD / 8; idx += 256) {   // 8 consecutive head dims per thread
    const int c = idx / (HEAD / 8), d = (idx - c * (HEAD / 8)) * 8;
    const int cqi = c / gp, chi = c % gp;
    if (chi >= g || cqi >= q_in_blk) continue;
    f32x4 lo4 = f32x4{0.f, 0.f, 0.f, 0.f}, hi4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < 4; w2++) {
      lo4 += *reinterpret_cast<const f32x4*>(&s_out[w2][c][d]);
      hi4 += *reinterpret_cast<const f32x4*>(&s_out[w2][c][d + 4]);
    }
    u32x4 o8;
    o8[0] = pack2<DT>(lo4[0], lo4[1]); o8[1] = pack2<DT>(lo4[2], lo4[3]);
    o8[2] = pack2<DT>(hi4[0], hi4[1]); o8[3] = pack2<DT>(hi4[2], hi4[3]);
    const size_t row = (size_t)(qstart + q0 + cqi);
    const int h = kvh * g + chi;
    if (single) {
      *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + (row * p.num_heads + h) * HEAD + d) = o8;
    } else {
      uint16_t* dst = reinterpret_cast<uint16_t*>(p.tmp_out) + ((row * p.num_heads + h) * p.max_parts + part) * HEAD + d;
      if (p.inline_reduce) {   // write-through: another workgroup reads these in this launch
        __hip_atomic_store(reinterpret_cast<u64*>(dst), (u64)o8[0] | ((u64)o8[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<u64*>(dst) + 1, (u64)o8[2] | ((u64)o8[3] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *reinterpret_cast<u32x4*>(dst) = o8;
      }
    }
  }
  PA_STAMP(5);
  PA_STAMP(9);
  if (single) return;
  if (wave == 0 && kq == 0 && col_valid) {
    const size_t row = (size_t)(qstart + q0 + qi);
    const size_t o2 = (row * p.num_heads + (kvh * g + hi)) * p.max_parts + part;
    if (p.inline_reduce) {
      __hip_atomic_store(reinterpret_cast<uint32_t*>(p.exp_sums + o2), __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<uint32_t*>(p.max_logits + o2), __float_as_uint(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      p.exp_sums[o2] = sum;
      p.max_logits[o2] = mx;
    }
  }
  if (!p.inline_reduce) return;

  // ---- merge of the partitions by the last workgroup to arrive --------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    int* cnt = g_pa_cnt + (size_t)p.cnt_bank * PA_BANK + (size_t)blockIdx.x * gridDim.z + blockIdx.z;
    // No release fence in front of the ticket: every published word was written by an agent-scope atomic
    // store (write-through) and waited for (vmcnt(0) + barrier above).  A fence here writes the L2 back in
    // every workgroup: measured -28 % on decode at 32 x 1024 tokens and +36 % on TTFT(2048).
    const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == nactive - 1;
    if (last) {
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  for (int idx = threadIdx.x; idx < 16 * HEAD / 8; idx += 256) {
    const int c = idx / (HEAD / 8), d = (idx - c * (HEAD / 8)) * 8;
    const int cqi = c / gp, chi = c % gp;
    if (chi >= g || cqi >= q_in_blk) continue;
    const size_t row = (size_t)(qstart + q0 + cqi);
    const int h = kvh * g + chi;
    const int ctx_r = seq_len - qlen + q0 + cqi + 1;
    const int nparts = (ctx_r + MI355Q_PA_PARTITION - 1) / MI355Q_PA_PARTITION;
    const size_t base = (row * p.num_heads + h) * p.max_parts;
    // Eight partitions per round trip: every load of a batch is requested before the first is used (indices past
    // the last partition repeat it; the sums below skip them).  One partition per iteration made the merge of a
    // 2048-token context sixteen dependent round trips -- longer than the attention itself.
    constexpr int MB = 8;
    auto ld32 = [](const float* q) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
    float gmax = -INFINITY;
    for (int i0 = 0; i0 < nparts; i0 += MB) {
      float m8[MB];
#pragma unroll
      for (int u = 0; u < MB; u++) m8[u] = ld32(p.max_logits + base + min(i0 + u, nparts - 1));
#pragma unroll
      for (int u = 0; u < MB; u++) gmax = fmaxf(gmax, m8[u]);
    }
    float wsum = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint16_t* tmp = reinterpret_cast<const uint16_t*>(p.tmp_out) + base * HEAD + d;
    for (int i0 = 0; i0 < nparts; i0 += MB) {
      float es8[MB], ml8[MB];
      u64 a8[MB], b8[MB];
#pragma unroll
      for (int u = 0; u < MB; u++) {
        const int i = min(i0 + u, nparts - 1);
        es8[u] = ld32(p.exp_sums + base + i);
        ml8[u] = ld32(p.max_logits + base + i);
        a8[u] = __hip_atomic_load(reinterpret_cast<const u64*>(tmp + (size_t)i * HEAD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b8[u] = __hip_atomic_load(reinterpret_cast<const u64*>(tmp + (size_t)i * HEAD) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < MB; u++) {
        if (i0 + u < nparts) {   // in partition order, as the stand-alone merge kernel sums them
          const float w = es8[u] * __expf(ml8[u] - gmax);
          wsum += w;
          const uint32_t wd[4] = {(uint32_t)a8[u], (uint32_t)(a8[u] >> 32), (uint32_t)b8[u], (uint32_t)(b8[u] >> 32)};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            acc[2 * e] += DT::to_f32((uint16_t)(wd[e] & 0xffff)) * w;
            acc[2 * e + 1] += DT::to_f32((uint16_t)(wd[e] >> 16)) * w;
          }
        }
      }
    }
    const float den = wsum + 1e-6f;   // divisions, as the stand-alone merge kernel: identical results
    u32x4 o8;
#pragma unroll
    for (int e = 0; e < 4; e++) o8[e] = pack2<DT>(acc[2 * e] / den, acc[2 * e + 1] / den);
    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + (row * p.num_heads + h) * HEAD + d) = o8;
  }
}

// merge partitions: out = sum_p tmp_p w_p / (sum_p w_p + 1e-6), w_p = exp_sum_p exp(max_p - max)
template <typename DT, int HEAD>
__global__ __launch_bounds__(HEAD) void paged_attention_reduce_kernel(const PAParams p, int max_qlen) {
  const int h = blockIdx.x;
  int seq, qi, qstart, qlen;
  if (p.varlen) {
    seq = blockIdx.y / max_qlen;
    qi = blockIdx.y % max_qlen;
    qstart = p.qsl[seq];
    qlen = p.qsl[seq + 1] - qstart;
    if (qi >= qlen) return;
  } else {
    seq = blockIdx.y;
    qi = 0;
    qlen = 1;
    qstart = seq;
    if (p.qsl) {
      qstart = p.qsl[seq];
      if (p.qsl[seq + 1] - qstart != 1) return;
    }
  }
  const int ctx = p.seq_lens[seq] - qlen + qi + 1;
  const int nparts = (ctx + MI355Q_PA_PARTITION - 1) / MI355Q_PA_PARTITION;
  const size_t row = (size_t)(qstart + qi);
  const size_t base = (row * p.num_heads + h) * p.max_parts;
  constexpr int MB = 8;   // partitions per round trip (indices past the last repeat it and are skipped in the sums)
  float gmax = -INFINITY;
  for (int i0 = 0; i0 < nparts; i0 += MB) {
    float m8[MB];
#pragma unroll
    for (int u = 0; u < MB; u++) m8[u] = p.max_logits[base + min(i0 + u, nparts - 1)];
#pragma unroll
    for (int u = 0; u < MB; u++) gmax = fmaxf(gmax, m8[u]);
  }
  float wsum = 0.f, acc = 0.f;
  const uint16_t* tmp = reinterpret_cast<const uint16_t*>(p.tmp_out) + base * HEAD + threadIdx.x;
  for (int i0 = 0; i0 < nparts; i0 += MB) {
    float es8[MB], ml8[MB];
    uint16_t t8[MB];
#pragma unroll
    for (int u = 0; u < MB; u++) {
      const int i = min(i0 + u, nparts - 1);
      es8[u] = p.exp_sums[base + i];
      ml8[u] = p.max_logits[base + i];
      t8[u] = tmp[(size_t)i * HEAD];
    }
#pragma unroll
    for (int u = 0; u < MB; u++) {
      if (i0 + u < nparts) {
        const float w = es8[u] * __expf(ml8[u] - gmax);
        wsum += w;
        acc += DT::to_f32(t8[u]) * w;
      }
    }
  }
  reinterpret_cast<uint16_t*>(p.out)[(row * p.num_heads + h) * HEAD + threadIdx.x] =
      DT::from_f32(acc / (wsum + 1e-6f));
}

// ---------------------------------------------------------------------------
// The decode step's attention launch: RoPE (NeoX, whole head) on the query heads and on the new key, the append
// of the new key / value to the paged cache, and the attention of ONE new token per sequence, as one kernel
// (the sequence rotary_embedding -> reshape_and_cache -> paged_attention of
// vllm/v1/attention/backends/rocm_attn.py:367-517, csrc/rocm/attention.cu:317-914 for the arithmetic).
//
// Same decomposition and the same fp32 operations in the same order as paged_attention_kernel above -- one
// workgroup of 4 waves per (sequence, 256-token partition, kv head), wave w owns tokens 64 w .. 64 w + 63,
// probabilities normalised inside the partition and rounded to the activation dtype -- so its outputs equal that
// kernel's bit for bit.  What differs is the order in which things are asked for, because at decode sizes the
// launch is nothing but a chain of memory round trips:
//   * everything that only the STEP defines (sequence length, position, this wave's block-table entries) is read
//     through the scalar cache: wave-uniform, counted apart from the vector loads, and back long before the
//     operands the previous launch has just written (q, new k, new v), which are requested in the same breath;
//   * the K / V tiles and the position's cos / sin row are requested the moment the scalar data is back -- they
//     do not wait for q -- so the two round trips of a dependent chain overlap into one;
//   * the new key is fetched in MFMA-fragment order like q, rotated in registers by every wave and patched into
//     the K fragments of the tile that holds the newest position; the new value likewise: no LDS stage, no
//     barrier in front of the first product.  The wave whose tile holds the new position writes both to the cache;
//   * integer divisions are gone from the prelude (block size and group padding are powers of two here; other
//     block sizes take paged_attention_kernel's fused form);
//   * the two wave-level reductions of the softmax use v_permlane16_swap / v_permlane32_swap instead of LDS
//     permutes, and only the columns of real query heads travel through LDS in the cross-wave sum.
// Columns of the 16-wide MFMA tile beyond the group's heads repeat the last head: finite everywhere, never stored.
// ---------------------------------------------------------------------------
typedef const int32_t __attribute__((address_space(4)))* pa_ci32p;   // read through the scalar cache
typedef const int64_t __attribute__((address_space(4)))* pa_ci64p;
typedef const float __attribute__((address_space(4)))* pa_cf32p;
typedef __attribute__((ext_vector_type(2))) unsigned pa_u2;

// max / sum over the four lanes (kq = 0..3) that share a column
__device__ inline float quad_rows_max(float v) {
  pa_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// the pairing of __shfl_xor(v, 16) then (v, 32): (own + partner) each time, so the sum is the same bits
__device__ inline float quad_rows_sum(float v) {
  const int kq = (threadIdx.x & 63) >> 4;
  pa_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  // r[0]: rows (0,0,2,2), r[1]: rows (1,1,3,3); own + partner
  v = (kq & 1) ? __uint_as_float(r[1]) + __uint_as_float(r[0]) : __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return (kq & 2) ? __uint_as_float(r[1]) + __uint_as_float(r[0]) : __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <typename DT, int HEAD, bool KVFP8>
__global__ __launch_bounds__(256, 2) void decode_step_kernel(const PAParams p) {
  constexpr int DS = HEAD / 32, DTILES = HEAD / 16, KR = KVFP8 ? DS / 2 : DS;
  constexpr int OPAD = HEAD + 4;   // floats per column of the cross-wave sum: columns land on different banks
  __shared__ float s_max[4][16];
  __shared__ float s_sum[4][16];
  __shared__ __attribute__((aligned(16))) float s_out[4][16][OPAD];
  __shared__ int s_last;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kq = lane >> 4, col = lane & 15;
  const int seq = blockIdx.x, part = blockIdx.y, kvh = blockIdx.z;
  const int g = p.g, gp = p.gp, bsh = p.bs_shift, bs = p.bs;
  PA_STAMP(0);

  // ---- what the step defines: scalar loads ---------------------------------------------------------------
  const int seq_len = ((pa_ci32p)p.seq_lens)[seq];
  const int64_t pos = ((pa_ci64p)p.positions)[seq];
  const float kscale = KVFP8 ? *(pa_cf32p)p.k_scale : 1.0f, vscale = KVFP8 ? *(pa_cf32p)p.v_scale : 1.0f;
  const int p0 = part * MI355Q_PA_PARTITION;
  const int wave_t0 = p0 + wave * 64;
  pa_ci32p table = (pa_ci32p)p.block_tables + (size_t)seq * p.max_blocks;
  int ent[4];
#pragma unroll
  for (int T = 0; T < 4; T++) ent[T] = table[min((wave_t0 + T * 16) >> bsh, p.max_blocks - 1)];

  // ---- what the previous launch wrote: q, the new key (both in fragment order), the new value ----------------
  // fragment ds of a lane covers head dims d0(ds) .. d0(ds) + 7
  //   16-bit cache: d0 = 32 ds + 8 kq;   fp8 cache: d0 = 16 (4 (ds / 2) + kq) + 8 (ds & 1)
  const uint16_t* rowp = p.q + (size_t)seq * p.q_stride;
  const int hq = min(col & (gp - 1), g - 1);
  const uint16_t* qp = rowp + (size_t)(kvh * g + hq) * HEAD;
  const uint16_t* kp = rowp + p.k_off + kvh * HEAD;
  const uint16_t* vp = rowp + p.v_off + kvh * HEAD + col;
  u32x4 qf[DS], kf[DS];
  uint16_t vn[DTILES];
#pragma unroll
  for (int ds = 0; ds < DS; ds++) {
    const int d0 = KVFP8 ? 16 * (4 * (ds >> 1) + kq) + 8 * (ds & 1) : 32 * ds + 8 * kq;
    qf[ds] = *reinterpret_cast<const u32x4*>(qp + d0);
    kf[ds] = *reinterpret_cast<const u32x4*>(kp + d0);
  }
#pragma unroll
  for (int dt = 0; dt < DTILES; dt++) vn[dt] = vp[dt * 16];
  __builtin_amdgcn_sched_barrier(0);

  if (p0 >= seq_len) {
    // a partition past the context.  The operands above are named here so that their requests stay in front of
    // this branch for the workgroups that go on (a load whose result only the other side needs is moved there).
    uint32_t t0 = vn[0], t1 = vn[DTILES - 1];
    asm volatile("" ::"v"(qf[0][0]), "v"(qf[DS - 1][3]), "v"(kf[0][0]), "v"(kf[DS - 1][3]), "v"(t0), "v"(t1));
    asm volatile("" ::"s"(ent[0]), "s"(ent[1]), "s"(ent[2]), "s"(ent[3]), "s"(pos), "s"(kscale), "s"(vscale));
    return;
  }
  PA_STAMP(6);

  // ---- second group: cos / sin of the position, K and V tiles ----------------------------------------------
  u32x4 cfrag[DS / 2], sfrag[DS / 2];
  {
    const uint16_t* cs = p.cos_sin + (size_t)pos * HEAD;
#pragma unroll
    for (int ds = 0; ds < DS / 2; ds++) {
      const int d0 = KVFP8 ? 16 * (4 * (ds >> 1) + kq) + 8 * (ds & 1) : 32 * ds + 8 * kq;
      cfrag[ds] = *reinterpret_cast<const u32x4*>(cs + d0);
      sfrag[ds] = *reinterpret_cast<const u32x4*>(cs + HEAD / 2 + d0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // requests in the order of use: vector loads return in issue order
  // A tile past the context reads cache block 0 (always mapped) and is skipped or masked below.
  bool tile_on[4];
  u32x4 kreg[4][KR];
  u32x2 vreg[4][DTILES];
#pragma unroll
  for (int T = 0; T < 4; T++) tile_on[T] = wave_t0 + T * 16 < seq_len;
  // Addresses: one buffer descriptor per tile (block, kv head, the tile's first row; wave-uniform, in scalar
  // registers) plus ONE 32-bit lane offset shared by all of the lane's requests; the fragment index is a scalar
  // offset.  A tile past the context gets a descriptor of size zero: its loads stay in the instruction stream
  // (no branch, so every later wait counts exactly) but every lane is out of range, which returns zeros and sends
  // nothing to the cache -- the L1 path of the CU moves 64 bytes per clock, and at a 170-token context a third of
  // the sixteen tiles would be wasted on it.
  constexpr int EBY = KVFP8 ? 1 : 2;                                   // bytes per cache element
  __amdgpu_buffer_rsrc_t kdesc[4], vdesc[4];
#pragma unroll
  for (int T = 0; T < 4; T++) {
    const int blk = tile_on[T] ? ent[T] : 0;
    const int row0 = (wave_t0 + T * 16) & (bs - 1);                    // the tile's first token row in its block
    const size_t base = ((size_t)blk * p.kv_block_stride + (size_t)kvh * p.kv_head_stride) * EBY;
    const int span = tile_on[T] ? 0x7fffffff : 0;
    kdesc[T] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.kc)) + base + (size_t)row0 * 16,
                                                 0, span, 0x00020000);
    vdesc[T] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.vc)) + base + (size_t)row0 * EBY,
                                                 0, span, 0x00020000);
  }
  const int koff = (kq * bs + col) * 16;                  // K: 16-byte groups [head dims / x][row]
  const int voff = (col * bs + 4 * kq) * EBY;             // V: [head dim][row]
#pragma unroll
  for (int T = 0; T < 4; T++) {
#pragma unroll
    for (int r = 0; r < KR; r++)
      kreg[T][r] = __builtin_amdgcn_raw_buffer_load_b128(kdesc[T], koff, r * 64 * bs, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  // The V tiles are requested further down, in two halves between the arithmetic: the CU's L1 path takes in
  // 64 bytes per clock and a wave cannot run ahead of its own requests, so a wave that asks for everything at
  // once stands at the last request until the path has drained -- with the arithmetic on q and K still to do.
  auto request_v = [&](int T) {
#pragma unroll
    for (int dt = 0; dt < DTILES; dt++) {
      if constexpr (!KVFP8) {
        vreg[T][dt] = __builtin_amdgcn_raw_buffer_load_b64(vdesc[T], voff, dt * 16 * bs * EBY, 0);
      } else {
        vreg[T][dt][0] = __builtin_amdgcn_raw_buffer_load_b32(vdesc[T], voff, dt * 16 * bs * EBY, 0);
        vreg[T][dt][1] = 0u;
      }
    }
  };
  PA_STAMP(8);

  // ---- RoPE on q and on the new key: the partner d + HEAD / 2 of every element is fragment ds + DS / 2 ----------
#pragma unroll
  for (int ds = 0; ds < DS / 2; ds++) {
    rope8<DT>(qf[ds], qf[ds + DS / 2], cfrag[ds], sfrag[ds]);
    rope8<DT>(kf[ds], kf[ds + DS / 2], cfrag[ds], sfrag[ds]);
  }
  __builtin_amdgcn_sched_barrier(0);
  request_v(0);
  request_v(1);
  __builtin_amdgcn_sched_barrier(0);
  PA_STAMP(7);

  // ---- the new token: into the cache, and into the fragments of the tile that holds it ---------------------------
  // (the key here, in front of the first product; the value further down, in front of the second: its patch
  // needs the V tile, the youngest of the requests above)
  const int tnew = seq_len - 1;
  const int tl = tnew - wave_t0;          // wave-uniform
  const bool owns_new = tl >= 0 && tl < 64;
  const int Tn = tl >> 4, cn = tl & 15;
  uint32_t vnew[DTILES];                  // the value's cache form, one element per head-dim tile
  if constexpr (!KVFP8) {
#pragma unroll
    for (int dt = 0; dt < DTILES; dt++) vnew[dt] = vn[dt];
  }
  if (owns_new) {
    u32x4 knew[KR];                       // the key's cache form
    if constexpr (!KVFP8) {
#pragma unroll
      for (int ds = 0; ds < DS; ds++) knew[ds] = kf[ds];
    } else {
      // value / scale is a true division (the arithmetic of the cache-write kernel), a dozen instructions each, and
      // this wave stands between every wave of the workgroup and the softmax barrier: the 16 lanes of a kq row hold
      // the same 32 key values and the 4 lanes of a column the same 8 value elements, so every lane quantises one
      // dword of the key (4 values) and two elements of the value instead of all 40, and the rows / columns exchange
      // the results (6 divisions per lane instead of 40)
      auto enc4 = [&](uint32_t lo, uint32_t hi) {   // four 16-bit values -> four e4m3 bytes of value / k_scale
        const float a = DT::to_f32((uint16_t)(lo & 0xffff)) / kscale, b = DT::to_f32((uint16_t)(lo >> 16)) / kscale;
        const float c = DT::to_f32((uint16_t)(hi & 0xffff)) / kscale, d = DT::to_f32((uint16_t)(hi >> 16)) / kscale;
        return (uint32_t)f32_to_fp8(a) | ((uint32_t)f32_to_fp8(b) << 8) | ((uint32_t)f32_to_fp8(c) << 16) |
               ((uint32_t)f32_to_fp8(d) << 24);
      };
      static_assert(DS == 4 && DTILES == 8, "the exchange below is laid out for head size 128");
      // dword j = 4 sp + e of the key's 32 bytes comes from fragment 2 sp + (e >> 1), words 2 (e & 1) and 2 (e & 1) + 1
      const int j = col & 7, jf = 2 * (j >> 2) + ((j >> 1) & 1), jh = j & 1;
      uint32_t lo = 0u, hi = 0u;
#pragma unroll
      for (int f = 0; f < DS; f++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const bool m = jf == f && jh == h;
          lo = m ? kf[f][2 * h] : lo;
          hi = m ? kf[f][2 * h + 1] : hi;
        }
      const uint32_t kmine = enc4(lo, hi);
#pragma unroll
      for (int sp = 0; sp < DS / 2; sp++)
#pragma unroll
        for (int e = 0; e < 4; e++) knew[sp][e] = (uint32_t)__shfl((int)kmine, (lane & 48) | (4 * sp + e), 64);
      // value elements 2 kq and 2 kq + 1 of this column
      uint32_t v0 = 0u, v1 = 0u;
#pragma unroll
      for (int dt = 0; dt < DTILES; dt++) {
        v0 = dt == 2 * kq ? vn[dt] : v0;
        v1 = dt == 2 * kq + 1 ? vn[dt] : v1;
      }
      const uint32_t q0 = f32_to_fp8(DT::to_f32((uint16_t)v0) / vscale), q1 = f32_to_fp8(DT::to_f32((uint16_t)v1) / vscale);
#pragma unroll
      for (int dt = 0; dt < DTILES; dt++)
        vnew[dt] = (uint32_t)__shfl((int)((dt & 1) ? q1 : q0), ((dt >> 1) << 4) | col, 64);
    }
    // cache append: the block is this wave's entry of the tile
    const int blk = Tn == 0 ? ent[0] : (Tn == 1 ? ent[1] : (Tn == 2 ? ent[2] : ent[3]));
    const int off = tnew & (bs - 1);
    const size_t base = (size_t)blk * p.kv_block_stride + (size_t)kvh * p.kv_head_stride;
    if (col == 0) {     // four lanes hold the whole key, 16 bytes (one x-group of the K layout) per fragment
      if constexpr (!KVFP8) {
        uint16_t* kb = reinterpret_cast<uint16_t*>(const_cast<void*>(p.kc)) + base;
#pragma unroll
        for (int ds = 0; ds < DS; ds++) *reinterpret_cast<u32x4*>(kb + ((size_t)(ds * 4 + kq) * bs + off) * 8) = knew[ds];
      } else {
        uint8_t* kb = reinterpret_cast<uint8_t*>(const_cast<void*>(p.kc)) + base;
#pragma unroll
        for (int sp = 0; sp < DS / 2; sp++) *reinterpret_cast<u32x4*>(kb + ((size_t)(sp * 4 + kq) * bs + off) * 16) = knew[sp];
      }
    }
    if (kq == 0) {      // sixteen lanes hold the value, head dims 16 dt + col
      if constexpr (!KVFP8) {
        uint16_t* vb = reinterpret_cast<uint16_t*>(const_cast<void*>(p.vc)) + base;
#pragma unroll
        for (int dt = 0; dt < DTILES; dt++) vb[(size_t)(dt * 16 + col) * bs + off] = (uint16_t)vnew[dt];
      } else {
        uint8_t* vb = reinterpret_cast<uint8_t*>(const_cast<void*>(p.vc)) + base;
#pragma unroll
        for (int dt = 0; dt < DTILES; dt++) vb[(size_t)(dt * 16 + col) * bs + off] = (uint8_t)vnew[dt];
      }
    }
    // this step's reads of that slot are replaced in registers (nobody waits for the stores)
    const bool krow = col == cn;                 // the lane's K row is the new token
#pragma unroll
    for (int T = 0; T < 4; T++) {
      if (T == Tn) {
#pragma unroll
        for (int r = 0; r < KR; r++)
#pragma unroll
          for (int e = 0; e < 4; e++) kreg[T][r][e] = krow ? knew[r][e] : kreg[T][r][e];
      }
    }
  }

  PA_STAMP(1);
  // ---- S^T = K Q^T ---------------------------------------------------------------------------------------
  f32x4 st[4];
  auto qk_tile = [&](int T) {
    st[T] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tile_on[T]) {
      if constexpr (!KVFP8) {
#pragma unroll
        for (int ds = 0; ds < DS; ds++) st[T] = mfma16<DT>(kreg[T][ds], qf[ds], st[T]);
      } else {
#pragma unroll
        for (int sp = 0; sp < DS / 2; sp++) {
          const u32x4 raw = kreg[T][sp];
          const u32x2 a0 = fp8x4_to_h16<DT>(raw[0]), a1 = fp8x4_to_h16<DT>(raw[1]);
          const u32x2 b0 = fp8x4_to_h16<DT>(raw[2]), b1 = fp8x4_to_h16<DT>(raw[3]);
          st[T] = mfma16<DT>(u32x4{a0[0], a0[1], a1[0], a1[1]}, qf[2 * sp], st[T]);
          st[T] = mfma16<DT>(u32x4{b0[0], b0[1], b1[0], b1[1]}, qf[2 * sp + 1], st[T]);
        }
      }
    }
  };
  qk_tile(0);
  qk_tile(1);
  __builtin_amdgcn_sched_barrier(0);
  request_v(2);
  request_v(3);
  __builtin_amdgcn_sched_barrier(0);
  qk_tile(2);
  qk_tile(3);
  PA_STAMP(2);

  // ---- logits, partition max / sum per column (the arithmetic of paged_attention_kernel) ----------------------
  const float scale = p.scale * kscale;
  float mw = -INFINITY;
#pragma unroll
  for (int T = 0; T < 4; T++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int tok = wave_t0 + T * 16 + 4 * kq + r;
      float v = st[T][r] * scale;
      v = tok < seq_len ? v : -INFINITY;
      st[T][r] = v;
      mw = fmaxf(mw, v);
    }
  mw = quad_rows_max(mw);
  const float mwref = mw == -INFINITY ? 0.f : mw;
  float sw = 0.f;
#pragma unroll
  for (int T = 0; T < 4; T++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float e = __expf(st[T][r] - mwref);
      st[T][r] = e;
      sw += e;
    }
  sw = quad_rows_sum(sw);
  if (kq == 0) {
    s_max[wave][col] = mw;
    s_sum[wave][col] = sw;
  }
  __syncthreads();
  float m4[4], s4[4];
#pragma unroll
  for (int w2 = 0; w2 < 4; w2++) {
    m4[w2] = s_max[w2][col];
    s4[w2] = s_sum[w2][col];
  }
  float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
  const float mref = mx == -INFINITY ? 0.f : mx;
  float sum = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < 4; w2++)   // a wave past the context published (-inf, 0): its term is exp(-inf) * 0 = 0
    sum += s4[w2] * __expf(m4[w2] - mref);
  const float inv = (mw == -INFINITY ? 0.f : __expf(mwref - mref)) / (sum + 1e-6f);
  PA_STAMP(3);

  // the new value into the V fragments: bit-field inserts, the position inside the tile is a run-time value
  if (owns_new) {
    const bool vrow = kq == (cn >> 2);           // the lane's four V tokens hold it
    constexpr int EB = KVFP8 ? 8 : 16;           // bits per cache element
    const int j = cn & 3;
    const int sh = (EB * j) & 31;
    const uint32_t m = (KVFP8 ? 0xffu : 0xffffu) << sh;
    const uint32_t m0 = (vrow && (KVFP8 || j < 2)) ? m : 0u;     // element j lives in word 0 ...
    const uint32_t m1 = (vrow && !KVFP8 && j >= 2) ? m : 0u;     // ... or word 1 of the lane's 64 bits
#pragma unroll
    for (int T = 0; T < 4; T++) {
      if (T == Tn) {
#pragma unroll
        for (int dt = 0; dt < DTILES; dt++) {
          const uint32_t val = vnew[dt] << sh;
          vreg[T][dt][0] = (vreg[T][dt][0] & ~m0) | (val & m0);
          if constexpr (!KVFP8) vreg[T][dt][1] = (vreg[T][dt][1] & ~m1) | (val & m1);
        }
      }
    }
  }

  // ---- O^T += V^T P^T ---------------------------------------------------------------------------------------
  f32x4 o[DTILES];
#pragma unroll
  for (int dt = 0; dt < DTILES; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s2 = 0; s2 < 2; s2++) {
    const int Te = 2 * s2, To = 2 * s2 + 1;
    if (tile_on[Te]) {
      u32x4 pf;
      pf[0] = pack2<DT>(st[Te][0] * inv, st[Te][1] * inv);
      pf[1] = pack2<DT>(st[Te][2] * inv, st[Te][3] * inv);
      pf[2] = pack2<DT>(st[To][0] * inv, st[To][1] * inv);
      pf[3] = pack2<DT>(st[To][2] * inv, st[To][3] * inv);
      // tokens at or past seq_len hold unwritten cache memory: force their V to zero
      const int te0 = wave_t0 + Te * 16 + 4 * kq, to0 = wave_t0 + To * 16 + 4 * kq;
      const int ne = min(max(seq_len - te0, 0), 4), no = min(max(seq_len - to0, 0), 4);
      const uint32_t me0 = ne >= 2 ? 0xffffffffu : (ne == 1 ? 0x0000ffffu : 0u);
      const uint32_t me1 = ne >= 4 ? 0xffffffffu : (ne == 3 ? 0x0000ffffu : 0u);
      const uint32_t mo0 = no >= 2 ? 0xffffffffu : (no == 1 ? 0x0000ffffu : 0u);
      const uint32_t mo1 = no >= 4 ? 0xffffffffu : (no == 3 ? 0x0000ffffu : 0u);
      const bool odd_ok = tile_on[To];
      if (wave_t0 + (To + 1) * 16 <= seq_len) {   // wave-uniform: every token of the pair is cached, no mask needed
#pragma unroll
        for (int dt = 0; dt < DTILES; dt++) {
          u32x2 ve, vo;
          if constexpr (!KVFP8) {
            ve = vreg[Te][dt];
            vo = vreg[To][dt];
          } else {
            ve = fp8x4_to_h16<DT>(vreg[Te][dt][0]);
            vo = fp8x4_to_h16<DT>(vreg[To][dt][0]);
          }
          o[dt] = mfma16<DT>(u32x4{ve[0], ve[1], vo[0], vo[1]}, pf, o[dt]);
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < DTILES; dt++) {
          u32x2 ve, vo = u32x2{0u, 0u};
          if constexpr (!KVFP8) {
            ve = vreg[Te][dt];
            if (odd_ok) vo = vreg[To][dt];
          } else {
            ve = fp8x4_to_h16<DT>(vreg[Te][dt][0]);
            if (odd_ok) vo = fp8x4_to_h16<DT>(vreg[To][dt][0]);
          }
          const u32x4 vf = u32x4{ve[0] & me0, ve[1] & me1, vo[0] & mo0, vo[1] & mo1};
          o[dt] = mfma16<DT>(vf, pf, o[dt]);
        }
      }
    }
  }
  PA_STAMP(4);

  // ---- cross-wave sum of O (columns of real heads only), store ---------------------------------------------------
  if (col < g) {
#pragma unroll
    for (int dt = 0; dt < DTILES; dt++)
      *reinterpret_cast<f32x4*>(&s_out[wave][col][dt * 16 + 4 * kq]) = o[dt] * vscale;
  }
  __syncthreads();

  const int nactive = (seq_len + MI355Q_PA_PARTITION - 1) / MI355Q_PA_PARTITION;
  const bool single = p.max_parts == 1 || (p.inline_reduce && nactive == 1);
  typedef unsigned long long u64;
  for (int idx = threadIdx.x; idx < g * (HEAD / 8); idx += 256) {   // 8 consecutive head dims per thread
    const int c = idx / (HEAD / 8), d = (idx - c * (HEAD / 8)) * 8;
    f32x4 lo4 = f32x4{0.f, 0.f, 0.f, 0.f}, hi4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < 4; w2++) {
      lo4 += *reinterpret_cast<const f32x4*>(&s_out[w2][c][d]);
      hi4 += *reinterpret_cast<const f32x4*>(&s_out[w2][c][d + 4]);
    }
    u32x4 o8;
    o8[0] = pack2<DT>(lo4[0], lo4[1]); o8[1] = pack2<DT>(lo4[2], lo4[3]);
    o8[2] = pack2<DT>(hi4[0], hi4[1]); o8[3] = pack2<DT>(hi4[2], hi4[3]);
    const int h = kvh * g + c;
    if (single) {
      *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + ((size_t)seq * p.num_heads + h) * HEAD + d) = o8;
    } else {
      uint16_t* dst = reinterpret_cast<uint16_t*>(p.tmp_out) + (((size_t)seq * p.num_heads + h) * p.max_parts + part) * HEAD + d;
      if (p.inline_reduce) {   // write-through: another workgroup reads these in this launch
        __hip_atomic_store(reinterpret_cast<u64*>(dst), (u64)o8[0] | ((u64)o8[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<u64*>(dst) + 1, (u64)o8[2] | ((u64)o8[3] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *reinterpret_cast<u32x4*>(dst) = o8;
      }
    }
  }
  PA_STAMP(5);
  if (single) return;
  if (wave == 0 && kq == 0 && col < g) {
    const size_t o2 = ((size_t)seq * p.num_heads + (kvh * g + col)) * p.max_parts + part;
    if (p.inline_reduce) {
      __hip_atomic_store(reinterpret_cast<uint32_t*>(p.exp_sums + o2), __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<uint32_t*>(p.max_logits + o2), __float_as_uint(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      p.exp_sums[o2] = sum;
      p.max_logits[o2] = mx;
    }
  }
  if (!p.inline_reduce) return;

  // ---- merge of the partitions by the last workgroup to arrive (the hand-off of paged_attention_kernel) ----------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    int* cnt = g_pa_cnt + (size_t)p.cnt_bank * PA_BANK + (size_t)blockIdx.x * gridDim.z + blockIdx.z;
    const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == nactive - 1;
    if (last) {
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  for (int idx = threadIdx.x; idx < g * (HEAD / 8); idx += 256) {
    const int c = idx / (HEAD / 8), d = (idx - c * (HEAD / 8)) * 8;
    const int h = kvh * g + c;
    const int nparts = nactive;
    const size_t base = ((size_t)seq * p.num_heads + h) * p.max_parts;
    constexpr int MB = 8;   // partitions per round trip (indices past the last repeat it and are skipped in the sums)
    auto ld32 = [](const float* q) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
    float gmax = -INFINITY;
    for (int i0 = 0; i0 < nparts; i0 += MB) {
      float m8[MB];
#pragma unroll
      for (int u = 0; u < MB; u++) m8[u] = ld32(p.max_logits + base + min(i0 + u, nparts - 1));
#pragma unroll
      for (int u = 0; u < MB; u++) gmax = fmaxf(gmax, m8[u]);
    }
    float wsum = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint16_t* tmp = reinterpret_cast<const uint16_t*>(p.tmp_out) + base * HEAD + d;
    for (int i0 = 0; i0 < nparts; i0 += MB) {
      float es8[MB], ml8[MB];
      u64 a8[MB], b8[MB];
#pragma unroll
      for (int u = 0; u < MB; u++) {
        const int i = min(i0 + u, nparts - 1);
        es8[u] = ld32(p.exp_sums + base + i);
        ml8[u] = ld32(p.max_logits + base + i);
        a8[u] = __hip_atomic_load(reinterpret_cast<const u64*>(tmp + (size_t)i * HEAD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b8[u] = __hip_atomic_load(reinterpret_cast<const u64*>(tmp + (size_t)i * HEAD) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < MB; u++) {
        if (i0 + u < nparts) {   // in partition order, as the stand-alone merge kernel sums them
          const float w = es8[u] * __expf(ml8[u] - gmax);
          wsum += w;
          const uint32_t wd[4] = {(uint32_t)a8[u], (uint32_t)(a8[u] >> 32), (uint32_t)b8[u], (uint32_t)(b8[u] >> 32)};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            acc[2 * e] += DT::to_f32((uint16_t)(wd[e] & 0xffff)) * w;
            acc[2 * e + 1] += DT::to_f32((uint16_t)(wd[e] >> 16)) * w;
          }
        }
      }
    }
    const float den = wsum + 1e-6f;
    u32x4 o8;
#pragma unroll
    for (int e = 0; e < 4; e++) o8[e] = pack2<DT>(acc[2 * e] / den, acc[2 * e + 1] / den);
    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + ((size_t)seq * p.num_heads + h) * HEAD + d) = o8;
  }
}

// ---------------------------------------------------------------------------
// Prompt rows over a 16-bit cache, long queries: one workgroup = 16 column tiles (256 / gp query tokens x gp
// heads: 64 tokens at four query heads per kv head) of one sequence x one kv head, walking the visible context
// in tiles of 64 tokens with a running softmax (no partitions, no scratch, no merge).  The kernel above gives every 16 MFMA columns (16 / gp query
// tokens x gp heads) their own copy of each K/V tile; here a tile is brought into LDS once per
// workgroup by LDS-DMA (double-buffered) and serves 16 column tiles: 8 waves x 2.
//   K image [4 token tiles][DS][64 lanes x 16 B]: the piece lane (row i, k-chunk) of an MFMA A operand,
//   exactly as the cache holds it (16 bytes = 8 head dims of one token).  Row i of token tile T is
//   token 32 (T / 2) + 8 (i / 4) + 4 (T % 2) + i % 4 of the 64: with that order the 8 probabilities a lane
//   holds after S^T = K Q^T for tiles (2 s, 2 s + 1) belong to 8 CONSECUTIVE tokens, so they are the
//   B operand of O^T += V^T P^T as they stand and V is fetched in its natural order, 16 bytes (8 tokens
//   of one head dim) per lane: V image [2 groups of 32 tokens][DTILES][64 lanes x 16 B].
// Probabilities are rounded to the activation dtype before the second product, as above -- but relative to the
// running maximum, not normalised: results agree with the partitioned kernel to the output's last bit or so
// (the reference's own prompt kernel is held to atol 1e-4 on values of magnitude 1e-3,
// tests/kernels/attention/test_prefix_prefill.py:159-163,322-323).
// ---------------------------------------------------------------------------
constexpr int PF_KV = 64;         // context tokens per step
constexpr int PF_NW = 8;          // waves per workgroup (two per SIMD)

// fp8 cache (KVFP8): a 16-byte piece holds 16 head dims of a token (K) or 16 tokens of a head dim (V), i.e. two
// MFMA k-steps after widening.  K: step ds takes dims 16 (4 (ds / 2) + kq) + 8 (ds & 1) + j, the query fragments are
// loaded to match.  V: a lane's piece is tokens 16 kq .. 16 kq + 15 of the tile, its halves feed the two 32-token
// products, so row i of token tile T is token 16 (i / 4) + 8 (T / 2) + 4 (T % 2) + i % 4 there.
// PF_CTL = column tiles per wave: 2 (64 query tokens x 4 heads per workgroup, one workgroup per CU) or 1 (32 tokens,
// two workgroups per CU: the second launch bound is waves per SIMD, four of them fit 128 registers each).
template <typename DT, int HEAD, bool KVFP8, int PF_CTL>
__global__ __launch_bounds__(64 * PF_NW, PF_CTL == 1 ? 4 : 2) void prefill_flash_kernel(const PAParams p) {
  constexpr int DS = HEAD / 32, DTILES = HEAD / 16;
  constexpr int KI = KVFP8 ? 4 * DS / 2 : 4 * DS, VI = KVFP8 ? DTILES : 2 * DTILES;     // LDS-DMA operations per stage
  constexpr int K_BYTES = KI * 1024, V_BYTES = VI * 1024, STAGE = K_BYTES + V_BYTES;
  constexpr int EB = KVFP8 ? 1 : 2;               // bytes per cache element
  static_assert(KI % PF_NW == 0 && VI % PF_NW == 0, "pieces are shared out evenly");
  __shared__ __attribute__((aligned(1024))) unsigned char kv[2 * STAGE];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kq = lane >> 4, col = lane & 15;
  const int g = p.num_heads / p.num_kv_heads, gp = p.gp, QB = 16 / gp;
  const int QW = PF_NW * PF_CTL * QB;              // query tokens per workgroup
  // Workgroups are numbered kv head fastest: with as many kv heads as XCDs (consecutive workgroups go to
  // consecutive XCDs) every XCD's L2 holds the K/V of one head only.
  const int kvh = blockIdx.x % p.num_kv_heads;
  const int bx = blockIdx.x / p.num_kv_heads;
  const int seq = bx / p.qblocks_per_seq;
  // a sequence's last query block (the longest context under the causal mask) is dispatched first
  const int q0 = (p.qblocks_per_seq - 1 - bx % p.qblocks_per_seq) * QW;
  const int qstart = p.qsl[seq];
  const int qlen = p.qsl[seq + 1] - qstart;
  if (q0 >= qlen) return;
  const int seq_len = p.seq_lens[seq];
  const int q_in_wg = min(QW, qlen - q0);
  const int wg_ctx = seq_len - qlen + q0 + q_in_wg;          // tokens visible to the workgroup's last query
  const int ntiles = (wg_ctx + PF_KV - 1) / PF_KV;
  const int32_t* table = p.block_tables + (size_t)seq * p.max_blocks;
  const unsigned char* kc = reinterpret_cast<const unsigned char*>(p.kc) + (size_t)kvh * p.kv_head_stride * EB;
  const unsigned char* vc = reinterpret_cast<const unsigned char*>(p.vc) + (size_t)kvh * p.kv_head_stride * EB;
  // token of row i of token tile T / first token of a lane's V piece, relative to the tile
  auto row_tok = [&](int T, int i) {
    return KVFP8 ? 16 * (i >> 2) + 8 * (T >> 1) + 4 * (T & 1) + (i & 3) : 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3);
  };
  const int last_tok = seq_len - 1;

  // this lane's query columns
  const int qi_l = col / gp, hi = col % gp;
  int ctx_c[PF_CTL];
  u32x4 qf[PF_CTL][DS];
  bool col_ok[PF_CTL];
#pragma unroll
  for (int c = 0; c < PF_CTL; c++) {
    const int qi = q0 + (wave * PF_CTL + c) * QB + qi_l;
    col_ok[c] = hi < g && qi < qlen;
    ctx_c[c] = col_ok[c] ? seq_len - qlen + qi + 1 : 0;
    const uint16_t* qp = p.q + (size_t)(qstart + min(qi, qlen - 1)) * p.q_stride + (size_t)(kvh * g + min(hi, g - 1)) * HEAD;
#pragma unroll
    for (int ds = 0; ds < DS; ds++) {
      const int d0 = KVFP8 ? 16 * (4 * (ds >> 1) + kq) + 8 * (ds & 1) : 32 * ds + 8 * kq;
      qf[c][ds] = *reinterpret_cast<const u32x4*>(qp + d0);
    }
  }
  const float kscale = KVFP8 ? *p.k_scale : 1.0f, vscale = KVFP8 ? *p.v_scale : 1.0f;
  const float c2 = p.scale * kscale * 1.44269504088896f;   // logits in the base-2 domain
  const float slope2 = (p.alibi ? p.alibi[kvh * g + min(hi, g - 1)] : 0.f) * 1.44269504088896f;
  // ALiBi slope per position step, in units of the raw logit (it is multiplied by c2 with the logit)
  const float slope_raw = p.alibi ? p.alibi[kvh * g + min(hi, g - 1)] / (p.scale * kscale) : 0.f;
  // context of the wave's first query token (0: the wave has none)
  const int wq0 = q0 + wave * PF_CTL * QB;
  const int wave_min_ctx = wq0 < qlen ? seq_len - qlen + wq0 + 1 : 0;

  // LDS-DMA pieces of a stage: K pieces idx = T * DS + ds (this lane: row i = col, k-chunk kq), V pieces
  // idx = s2 * DTILES + dt (this lane: head dim 16 dt + col, tokens 8 kq .. 8 kq + 7 of group s2); a wave takes
  // idx = wave, wave + PF_NW, ...  Tokens past the sequence read the piece that holds its last token (valid
  // memory; their logits are masked, their V forced to zero).
  // block index of a token: a shift -- this kernel is launched for power-of-two block sizes only (an integer division
  // costs about twenty vector instructions, four times per wave and tile)
  const int bs_shift = p.bs_shift, bs_mask = p.bs - 1;
  auto blk_of = [&](int tok) { return tok >> bs_shift; };
  auto stage = [&](int tile, int buf) {
    unsigned char* st = kv + buf * STAGE;
    const int t0 = tile * PF_KV;
#pragma unroll
    for (int n = 0; n < KI / PF_NW; n++) {
      const int idx = wave + PF_NW * n;
      const int T = KVFP8 ? idx / (DS / 2) : idx / DS, dc = KVFP8 ? idx % (DS / 2) : idx % DS;   // token tile, 16-byte dim piece
      const int tok = min(t0 + row_tok(T, col), last_tok);
      const int bi = blk_of(tok), bo = tok & bs_mask;
      const int blk = table[bi];
      const unsigned char* src = kc + ((size_t)blk * p.kv_block_stride) * EB + ((size_t)(dc * 4 + kq) * p.bs + bo) * 16;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)(st + idx * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < VI / PF_NW; n++) {
      const int idx = wave + PF_NW * n;
      const int s2 = KVFP8 ? 0 : idx / DTILES, dt = idx % DTILES;
      const int tok = KVFP8 ? min(t0 + 16 * kq, last_tok & ~15) : min(t0 + 32 * s2 + 8 * kq, last_tok & ~7);
      const int bi = blk_of(tok), bo = tok & bs_mask;
      const int blk = table[bi];
      const unsigned char* src = vc + ((size_t)blk * p.kv_block_stride + (size_t)(dt * 16 + col) * p.bs + bo) * EB;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)(st + K_BYTES + idx * 1024), 16, 0, 0);
    }
  };

  f32x4 o[PF_CTL][DTILES];
  float mrun[PF_CTL], lrun[PF_CTL];   // running maximum (per column) and this lane's share of the running sum
#pragma unroll
  for (int c = 0; c < PF_CTL; c++) {
    mrun[c] = -INFINITY;
    lrun[c] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DTILES; dt++) o[c][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  stage(0, 0);
  for (int tile = 0; tile < ntiles; tile++) {
    const int buf = tile & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // tile has landed for every wave; the other buffer is free
    asm volatile("" ::: "memory");
    if (tile + 1 < ntiles) stage(tile + 1, buf ^ 1);
    const unsigned char* st = kv + buf * STAGE;
    const int t0 = tile * PF_KV;

    // ---- S^T = K Q^T ------------------------------------------------------------------------
    f32x4 sv[PF_CTL][4];
#pragma unroll
    for (int T = 0; T < 4; T++) {
      if constexpr (!KVFP8) {
#pragma unroll
        for (int ds = 0; ds < DS; ds++) {
          const u32x4 kf = *reinterpret_cast<const u32x4*>(st + (T * DS + ds) * 1024 + 16 * lane);
#pragma unroll
          for (int c = 0; c < PF_CTL; c++)
            sv[c][T] = mfma16<DT>(kf, qf[c][ds], ds == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sv[c][T]);
        }
      } else {
#pragma unroll
        for (int sp = 0; sp < DS / 2; sp++) {
          const u32x4 raw = *reinterpret_cast<const u32x4*>(st + (T * (DS / 2) + sp) * 1024 + 16 * lane);
          const u32x2 a0 = fp8x4_to_h16<DT>(raw[0]), a1 = fp8x4_to_h16<DT>(raw[1]);
          const u32x2 b0 = fp8x4_to_h16<DT>(raw[2]), b1 = fp8x4_to_h16<DT>(raw[3]);
#pragma unroll
          for (int c = 0; c < PF_CTL; c++) {
            sv[c][T] = mfma16<DT>(u32x4{a0[0], a0[1], a1[0], a1[1]}, qf[c][2 * sp], sp == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sv[c][T]);
            sv[c][T] = mfma16<DT>(u32x4{b0[0], b0[1], b1[0], b1[1]}, qf[c][2 * sp + 1], sv[c][T]);
          }
        }
      }
    }
    // The 32-token form keeps masks and tail handling as selects inside one straight-line block (124 registers:
    // two workgroups per CU); the 64-token form takes them out of the common path with wave-uniform branches.
    if constexpr (PF_CTL == 1) {
      // ---- running softmax per column, in the base-2 domain (scale * log2 e folded into one multiply-add) -----
      // Tiles wholly inside the context of the wave's first query token need no mask, and without ALiBi the
      // logit, the shift by the maximum and the exponent's scaling are one fma + one v_exp per element.
      u32x4 pf[PF_CTL][2];
      const bool plain = !p.alibi && t0 + PF_KV <= wave_min_ctx;   // wave-uniform
      bool rescale = false;
      float alpha[PF_CTL];
#pragma unroll
      for (int c = 0; c < PF_CTL; c++) {
        float tmax = -INFINITY;
        if (plain) {
#pragma unroll
          for (int T = 0; T < 4; T++)
            tmax = fmaxf(tmax, fmaxf(fmaxf(sv[c][T][0], sv[c][T][1]), fmaxf(sv[c][T][2], sv[c][T][3])));
          tmax *= c2;
        } else {
#pragma unroll
          for (int T = 0; T < 4; T++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int tok = t0 + row_tok(T, 4 * kq) + r;
              float v = sv[c][T][r] * c2 + slope2 * (float)(tok - ctx_c[c] + 1);
              v = tok < ctx_c[c] ? v : -INFINITY;
              sv[c][T][r] = v;
              tmax = fmaxf(tmax, v);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun[c], tmax);
        const float mref = mnew == -INFINITY ? 0.f : mnew;
        alpha[c] = mrun[c] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mrun[c] - mref);
        rescale |= mnew != mrun[c];
        mrun[c] = mnew;
        float ls = 0.f;
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float e = plain ? __builtin_amdgcn_exp2f(__builtin_fmaf(sv[c][T][r], c2, -mref)) : __builtin_amdgcn_exp2f(sv[c][T][r] - mref);
            sv[c][T][r] = e;
            ls += e;
          }
        lrun[c] = lrun[c] * alpha[c] + ls;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          pf[c][s2][0] = pack2<DT>(sv[c][2 * s2][0], sv[c][2 * s2][1]);
          pf[c][s2][1] = pack2<DT>(sv[c][2 * s2][2], sv[c][2 * s2][3]);
          pf[c][s2][2] = pack2<DT>(sv[c][2 * s2 + 1][0], sv[c][2 * s2 + 1][1]);
          pf[c][s2][3] = pack2<DT>(sv[c][2 * s2 + 1][2], sv[c][2 * s2 + 1][3]);
        }
      }
      if (__builtin_amdgcn_ballot_w64(rescale) != 0) {   // some column's maximum moved: rescale what has been summed
#pragma unroll
        for (int c = 0; c < PF_CTL; c++)
#pragma unroll
          for (int dt = 0; dt < DTILES; dt++) o[c][dt] *= alpha[c];
      }
      // ---- O^T += V^T P^T --------------------------------------------------------------------------
      const bool ragged = t0 + PF_KV > seq_len;   // the tile reaches past the sequence: unwritten cache memory
      auto tail_mask = [&](int first_tok, uint32_t (&vm)[4]) {   // the lane's 8 tokens from first_tok on
        const int nv = min(max(seq_len - first_tok, 0), 8);
#pragma unroll
        for (int e = 0; e < 4; e++) vm[e] = nv >= 2 * e + 2 ? 0xffffffffu : (nv == 2 * e + 1 ? 0x0000ffffu : 0u);
      };
      if constexpr (!KVFP8) {
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          uint32_t vm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
          if (ragged) tail_mask(t0 + 32 * s2 + 8 * kq, vm);
#pragma unroll
          for (int dt = 0; dt < DTILES; dt++) {
            u32x4 vf = *reinterpret_cast<const u32x4*>(st + K_BYTES + (s2 * DTILES + dt) * 1024 + 16 * lane);
            if (ragged) {
              // a group that starts past the sequence was fetched from the piece of the last token: all zero here
              vf[0] &= vm[0]; vf[1] &= vm[1]; vf[2] &= vm[2]; vf[3] &= vm[3];
            }
#pragma unroll
            for (int c = 0; c < PF_CTL; c++) o[c][dt] = mfma16<DT>(vf, pf[c][s2], o[c][dt]);
          }
        }
      } else {
        uint32_t vm0[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        uint32_t vm1[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (ragged) {
          tail_mask(t0 + 16 * kq, vm0);
          tail_mask(t0 + 16 * kq + 8, vm1);
        }
#pragma unroll
        for (int dt = 0; dt < DTILES; dt++) {
          const u32x4 raw = *reinterpret_cast<const u32x4*>(st + K_BYTES + dt * 1024 + 16 * lane);
          const u32x2 a0 = fp8x4_to_h16<DT>(raw[0]), a1 = fp8x4_to_h16<DT>(raw[1]);
          const u32x2 b0 = fp8x4_to_h16<DT>(raw[2]), b1 = fp8x4_to_h16<DT>(raw[3]);
          u32x4 v0 = u32x4{a0[0], a0[1], a1[0], a1[1]}, v1 = u32x4{b0[0], b0[1], b1[0], b1[1]};
          if (ragged) {
            v0[0] &= vm0[0]; v0[1] &= vm0[1]; v0[2] &= vm0[2]; v0[3] &= vm0[3];
            v1[0] &= vm1[0]; v1[1] &= vm1[1]; v1[2] &= vm1[2]; v1[3] &= vm1[3];
          }
#pragma unroll
          for (int c = 0; c < PF_CTL; c++) {
            o[c][dt] = mfma16<DT>(v0, pf[c][0], o[c][dt]);
            o[c][dt] = mfma16<DT>(v1, pf[c][1], o[c][dt]);
          }
        }
      }
    } else {
      // ---- running softmax per column, in the base-2 domain (scale * log2 e folded into one multiply-add) -----
      // Tiles wholly inside the context of the wave's first query token need no mask, and without ALiBi the
      // logit, the shift by the maximum and the exponent's scaling are one fma + one v_exp per element.
      u32x4 pf[PF_CTL][2];
      const bool plain = !p.alibi && t0 + PF_KV <= wave_min_ctx;   // wave-uniform
      bool rescale = false;
      float alpha[PF_CTL];
      // Tiles that touch the causal diagonal (or carry ALiBi) are masked in the raw-logit domain first, under a branch
      // the whole wave takes or skips (the empty asm keeps the compiler from turning it into selects that every tile
      // would execute); after that one arithmetic serves both kinds of tile: masked logits are -inf, and
      // exp2(-inf * c2 - m) = 0.
      if (!plain) {
        asm volatile("");
#pragma unroll
        for (int c = 0; c < PF_CTL; c++)
#pragma unroll
          for (int T = 0; T < 4; T++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int tok = t0 + row_tok(T, 4 * kq) + r;
              const float v = sv[c][T][r] + slope_raw * (float)(tok - ctx_c[c] + 1);
              sv[c][T][r] = tok < ctx_c[c] ? v : -INFINITY;
            }
      }
#pragma unroll
      for (int c = 0; c < PF_CTL; c++) {
        float tmax = -INFINITY;
#pragma unroll
        for (int T = 0; T < 4; T++)
          tmax = fmaxf(tmax, fmaxf(fmaxf(sv[c][T][0], sv[c][T][1]), fmaxf(sv[c][T][2], sv[c][T][3])));
        tmax *= c2;
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun[c], tmax);
        const float mref = mnew == -INFINITY ? 0.f : mnew;
        alpha[c] = mrun[c] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mrun[c] - mref);
        rescale |= mnew != mrun[c];
        mrun[c] = mnew;
        float ls = 0.f;
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[c][T][r], c2, -mref));
            sv[c][T][r] = e;
            ls += e;
          }
        lrun[c] = lrun[c] * alpha[c] + ls;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          pf[c][s2][0] = pack2<DT>(sv[c][2 * s2][0], sv[c][2 * s2][1]);
          pf[c][s2][1] = pack2<DT>(sv[c][2 * s2][2], sv[c][2 * s2][3]);
          pf[c][s2][2] = pack2<DT>(sv[c][2 * s2 + 1][0], sv[c][2 * s2 + 1][1]);
          pf[c][s2][3] = pack2<DT>(sv[c][2 * s2 + 1][2], sv[c][2 * s2 + 1][3]);
        }
      }
      if (__builtin_amdgcn_ballot_w64(rescale) != 0) {   // some column's maximum moved: rescale what has been summed
#pragma unroll
        for (int c = 0; c < PF_CTL; c++)
#pragma unroll
          for (int dt = 0; dt < DTILES; dt++) o[c][dt] *= alpha[c];
      }
      // ---- O^T += V^T P^T --------------------------------------------------------------------------
      const bool ragged = t0 + PF_KV > seq_len;   // the tile reaches past the sequence: unwritten cache memory
      auto tail_mask = [&](int first_tok, uint32_t (&vm)[4]) {   // the lane's 8 tokens from first_tok on
        const int nv = min(max(seq_len - first_tok, 0), 8);
#pragma unroll
        for (int e = 0; e < 4; e++) vm[e] = nv >= 2 * e + 2 ? 0xffffffffu : (nv == 2 * e + 1 ? 0x0000ffffu : 0u);
      };
      auto pv = [&](auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        if constexpr (!KVFP8) {
#pragma unroll
          for (int s2 = 0; s2 < 2; s2++) {
            uint32_t vm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            if constexpr (MASKED) tail_mask(t0 + 32 * s2 + 8 * kq, vm);
#pragma unroll
            for (int dt = 0; dt < DTILES; dt++) {
              u32x4 vf = *reinterpret_cast<const u32x4*>(st + K_BYTES + (s2 * DTILES + dt) * 1024 + 16 * lane);
              if constexpr (MASKED) {
                // a group that starts past the sequence was fetched from the piece of the last token: all zero here
                vf[0] &= vm[0]; vf[1] &= vm[1]; vf[2] &= vm[2]; vf[3] &= vm[3];
              }
#pragma unroll
              for (int c = 0; c < PF_CTL; c++) o[c][dt] = mfma16<DT>(vf, pf[c][s2], o[c][dt]);
            }
          }
        } else {
          uint32_t vm0[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
          uint32_t vm1[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
          if constexpr (MASKED) {
            tail_mask(t0 + 16 * kq, vm0);
            tail_mask(t0 + 16 * kq + 8, vm1);
          }
#pragma unroll
          for (int dt = 0; dt < DTILES; dt++) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(st + K_BYTES + dt * 1024 + 16 * lane);
            const u32x2 a0 = fp8x4_to_h16<DT>(raw[0]), a1 = fp8x4_to_h16<DT>(raw[1]);
            const u32x2 b0 = fp8x4_to_h16<DT>(raw[2]), b1 = fp8x4_to_h16<DT>(raw[3]);
            u32x4 v0 = u32x4{a0[0], a0[1], a1[0], a1[1]}, v1 = u32x4{b0[0], b0[1], b1[0], b1[1]};
            if constexpr (MASKED) {
              v0[0] &= vm0[0]; v0[1] &= vm0[1]; v0[2] &= vm0[2]; v0[3] &= vm0[3];
              v1[0] &= vm1[0]; v1[1] &= vm1[1]; v1[2] &= vm1[2]; v1[3] &= vm1[3];
            }
#pragma unroll
            for (int c = 0; c < PF_CTL; c++) {
              o[c][dt] = mfma16<DT>(v0, pf[c][0], o[c][dt]);
              o[c][dt] = mfma16<DT>(v1, pf[c][1], o[c][dt]);
            }
          }
        }
      };
      if (ragged) {   // the sequence's last tile only: a branch of its own, so that no other tile pays for the masks
        asm volatile("");
        pv(std::true_type{});
      } else {
        pv(std::false_type{});
      }
    }
  }

  // ---- normalise and store: lane (kq, col) holds head dims 16 dt + 4 kq .. + 3 of its columns ------------
#pragma unroll
  for (int c = 0; c < PF_CTL; c++) {
    float l = lrun[c];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (!col_ok[c]) continue;
    const float inv = vscale / (l + 1e-6f);
    const int qi = q0 + (wave * PF_CTL + c) * QB + qi_l;
    uint16_t* dst = reinterpret_cast<uint16_t*>(p.out) + ((size_t)(qstart + qi) * p.num_heads + (kvh * g + hi)) * HEAD + 4 * kq;
#pragma unroll
    for (int dt = 0; dt < DTILES; dt++) {
      u32x2 w;
      w[0] = pack2<DT>(o[c][dt][0] * inv, o[c][dt][1] * inv);
      w[1] = pack2<DT>(o[c][dt][2] * inv, o[c][dt][3] * inv);
      *reinterpret_cast<u32x2*>(dst + 16 * dt) = w;
    }
  }
}

template <typename DT, int CTL>
static int launch_prefill_flash_ctl(const PAParams& p_in, int head_size, int kv_fp8, int max_qlen, hipStream_t stream) {
  PAParams p = p_in;
  const int QW = PF_NW * CTL * (16 / p.gp);
  p.qblocks_per_seq = (max_qlen + QW - 1) / QW;
  dim3 grid(p.num_seqs * p.qblocks_per_seq * p.num_kv_heads), block(64 * PF_NW);
  if (kv_fp8) hipLaunchKernelGGL((prefill_flash_kernel<DT, 128, true, CTL>), grid, block, 0, stream, p);
  else if (head_size == 128) hipLaunchKernelGGL((prefill_flash_kernel<DT, 128, false, CTL>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((prefill_flash_kernel<DT, 64, false, CTL>), grid, block, 0, stream, p);
  return check_launch("prefill_flash_kernel");
}

// 32 or 64 query tokens per workgroup.  The narrow form reads every K/V tile twice as often but brings twice the
// workgroups, two resident per CU, and halves the longest one under the causal mask.  Taken where the wide form
// leaves CUs idle, for short prompts in batches beyond one round, and for a single prompt that makes at most one
// round of the wide form; measurements in the table of the selection below (MI355Q_PREFILL_FLASH_CTL = 1 / 2 forces).
template <typename DT>
static int launch_prefill_flash(const PAParams& p_in, int head_size, int kv_fp8, int max_qlen, hipStream_t stream) {
  const int qw2 = PF_NW * 2 * (16 / p_in.gp);
  const long wgs2 = (long)p_in.num_seqs * ((max_qlen + qw2 - 1) / qw2) * p_in.num_kv_heads;
  const int forced = env_int("MI355Q_PREFILL_FLASH_CTL", 0);
  const bool narrow = forced ? forced == 1
                             : (wgs2 < 256 || (max_qlen <= 512 && wgs2 > 256) || (p_in.num_seqs == 1 && wgs2 <= 256));
  if (narrow) return launch_prefill_flash_ctl<DT, 1>(p_in, head_size, kv_fp8, max_qlen, stream);
  return launch_prefill_flash_ctl<DT, 2>(p_in, head_size, kv_fp8, max_qlen, stream);
}

template <typename DT>
static int launch_pa(const PAParams& p_in, int head_size, int kv_fp8, int qblocks, int max_qlen,
                     hipStream_t stream, bool fused = false) {
  PAParams p = p_in;
  // partitions are merged inside the kernel whenever this stream's bank of arrival counters covers the grid
  p.cnt_bank = 0;
  p.inline_reduce = 0;
  if (p.max_parts > 1 && (long)qblocks * p.num_kv_heads <= PA_BANK && !env_flag("MI355Q_PA_REDUCE_KERNEL")) {
    const int bank = stream_bank((void*)stream);
    if (bank >= 0) { p.cnt_bank = bank; p.inline_reduce = 1; }
  }
  dim3 grid(qblocks, p.max_parts, p.num_kv_heads), block(256);
  // the decode step: its own kernel (power-of-two block sizes; MI355Q_DECODE_STEP_GENERIC=1 keeps the fused form
  // of the general kernel, for comparisons)
  const bool step_kernel = fused && p.bs_shift >= 0 && !env_flag("MI355Q_DECODE_STEP_GENERIC");
#define MI355Q_PA(H, F8, FU) hipLaunchKernelGGL((paged_attention_kernel<DT, H, F8, FU>), grid, block, 0, stream, p)
  if (step_kernel) {
    if (head_size == 128) {
      if (kv_fp8) hipLaunchKernelGGL((decode_step_kernel<DT, 128, true>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((decode_step_kernel<DT, 128, false>), grid, block, 0, stream, p);
    } else {
      hipLaunchKernelGGL((decode_step_kernel<DT, 64, false>), grid, block, 0, stream, p);
    }
  } else if (fused) {
    if (head_size == 128) { if (kv_fp8) MI355Q_PA(128, true, true); else MI355Q_PA(128, false, true); }
    else MI355Q_PA(64, false, true);
  } else if (head_size == 128) { if (kv_fp8) MI355Q_PA(128, true, false); else MI355Q_PA(128, false, false); }
  else { if (kv_fp8) MI355Q_PA(64, true, false); else MI355Q_PA(64, false, false); }
#undef MI355Q_PA
  if (int e = check_launch("paged_attention_kernel")) return e;
  if (p.max_parts > 1 && !p.inline_reduce) {
    dim3 rgrid(p.num_heads, p.num_seqs * max_qlen);
    if (head_size == 128)
      hipLaunchKernelGGL((paged_attention_reduce_kernel<DT, 128>), rgrid, dim3(128), 0, stream, p, max_qlen);
    else
      hipLaunchKernelGGL((paged_attention_reduce_kernel<DT, 64>), rgrid, dim3(64), 0, stream, p, max_qlen);
    if (int e = check_launch("paged_attention_reduce_kernel")) return e;
  }
  return MI355Q_OK;
}

static int pa_common(PAParams& p, int head_size, int block_size, int max_seq_len, int dtype,
                     int kv_fp8, const char* what) {
  p.stamps = g_pa_stamps;
  MI355Q_REQUIRE(dtype == MI355Q_BF16 || dtype == MI355Q_FP16, "%s: dtype must be bf16 or fp16", what);
  MI355Q_REQUIRE(head_size == 64 || head_size == 128, "%s: head_size %d not in {64,128}", what, head_size);
  MI355Q_REQUIRE(block_size > 0 && block_size % 16 == 0, "%s: block_size %d must be a multiple of 16", what, block_size);
  MI355Q_REQUIRE(p.num_kv_heads > 0 && p.num_heads % p.num_kv_heads == 0, "%s: heads %d / kv heads %d", what, p.num_heads, p.num_kv_heads);
  const int g = p.num_heads / p.num_kv_heads;
  MI355Q_REQUIRE(g >= 1 && g <= 16, "%s: unsupported gqa ratio %d", what, g);
  MI355Q_REQUIRE(p.out && p.q && p.kc && p.vc && p.block_tables && p.seq_lens, "%s: null pointer", what);
  if (kv_fp8) MI355Q_REQUIRE(p.k_scale && p.v_scale, "%s: fp8 cache needs k_scale/v_scale", what);
  MI355Q_REQUIRE(p.q_stride % 8 == 0 && ((uintptr_t)p.q & 15) == 0, "%s: query must be 16-byte aligned", what);
  MI355Q_REQUIRE(max_seq_len > 0, "%s: max_seq_len %d", what, max_seq_len);
  p.max_parts = (max_seq_len + MI355Q_PA_PARTITION - 1) / MI355Q_PA_PARTITION;
  if (p.max_parts > 1)
    MI355Q_REQUIRE(p.exp_sums && p.max_logits && p.tmp_out, "%s: partition scratch required for max_seq_len > 256", what);
  int gp = 1;
  while (gp < g) gp <<= 1;
  p.gp = gp;
  p.g = g;
  p.bs_shift = (block_size & (block_size - 1)) == 0 ? __builtin_ctz((unsigned)block_size) : -1;
  return 0;
}

}  // namespace mi355q

using namespace mi355q;

extern "C" int mi355q_paged_attention(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                                      const void* query, int64_t q_stride, const void* key_cache,
                                      const void* value_cache, int64_t kv_block_stride,
                                      int64_t kv_head_stride, int num_seqs, int num_heads,
                                      int num_kv_heads, int head_size, float scale,
                                      const int32_t* block_tables, int max_blocks_per_seq,
                                      const int32_t* seq_lens, const int32_t* query_start_loc,
                                      int block_size, int max_seq_len, const float* alibi_slopes,
                                      int dtype, int kv_fp8, const float* k_scale,
                                      const float* v_scale, mi355q_stream_t stream) {
  PAParams p;
  p.out = out; p.exp_sums = exp_sums; p.max_logits = max_logits; p.tmp_out = tmp_out;
  p.q = (const uint16_t*)query; p.q_stride = q_stride; p.kc = key_cache; p.vc = value_cache;
  p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.num_kv_heads = num_kv_heads; p.scale = scale;
  p.block_tables = block_tables; p.max_blocks = max_blocks_per_seq; p.seq_lens = seq_lens;
  p.qsl = query_start_loc; p.bs = block_size; p.alibi = alibi_slopes; p.k_scale = k_scale;
  p.v_scale = v_scale; p.varlen = 0; p.qblocks_per_seq = 1;
  p.positions = nullptr; p.cos_sin = nullptr; p.k_off = 0; p.v_off = 0;
  if (num_seqs == 0) return MI355Q_OK;
  if (int e = pa_common(p, head_size, block_size, max_seq_len, dtype, kv_fp8, "paged_attention")) return e;
  if (dtype == MI355Q_BF16) return launch_pa<BF16>(p, head_size, kv_fp8, num_seqs, 1, (hipStream_t)stream);
  return launch_pa<FP16>(p, head_size, kv_fp8, num_seqs, 1, (hipStream_t)stream);
}

extern "C" int mi355q_paged_prefill_attention(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
    int64_t q_stride, const void* key_cache, const void* value_cache, int64_t kv_block_stride,
    int64_t kv_head_stride, int num_seqs, int num_heads, int num_kv_heads, int head_size,
    float scale, const int32_t* block_tables, int max_blocks_per_seq, const int32_t* seq_lens,
    const int32_t* query_start_loc, int max_query_len, int block_size, int max_seq_len,
    const float* alibi_slopes, int dtype, int kv_fp8, const float* k_scale, const float* v_scale,
    mi355q_stream_t stream) {
  PAParams p;
  p.out = out; p.exp_sums = exp_sums; p.max_logits = max_logits; p.tmp_out = tmp_out;
  p.q = (const uint16_t*)query; p.q_stride = q_stride; p.kc = key_cache; p.vc = value_cache;
  p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.num_kv_heads = num_kv_heads; p.scale = scale;
  p.block_tables = block_tables; p.max_blocks = max_blocks_per_seq; p.seq_lens = seq_lens;
  p.qsl = query_start_loc; p.bs = block_size; p.alibi = alibi_slopes; p.k_scale = k_scale;
  p.v_scale = v_scale; p.varlen = 1;
  p.positions = nullptr; p.cos_sin = nullptr; p.k_off = 0; p.v_off = 0;
  if (num_seqs == 0) return MI355Q_OK;
  MI355Q_REQUIRE(query_start_loc != nullptr && max_query_len > 0, "paged_prefill_attention: query_start_loc and max_query_len are required");
  if (int e = pa_common(p, head_size, block_size, max_seq_len, dtype, kv_fp8, "paged_prefill_attention")) return e;
  // long queries over a 16-bit cache: the running-softmax kernel (K/V tiles shared by 8 column tiles through LDS)
  // ... when the query is long or the batch brings enough workgroups: a single short prompt is a handful of
  // workgroups walking their tiles one after the other, and the partitioned kernel's many small workgroups
  // finish sooner (us, partitioned / running in the form launch_prefill_flash picks: 1 x 128 tokens 13.5 / 11-15,
  // 1 x 256 13.3 / 14.6, 1 x 384 27 / 17.7, 1 x 512 37 / 22.3, 1 x 1024 94 / 38.8, 1 x 2048 295 / 83.6,
  // 1 x 4096 1140 / 197; 2 x 128 12.2 / 11.5, 4 x 128 17.2 / 11.5, 32 x 128 95 / 28.4, 8 x 512 188 / 56,
  // 32 x 1024 3120 / 574.  The two forms of the running-softmax kernel, 64- / 32-token: 1 x 512 29.3 / 22.6,
  // 1 x 1024 49.6 / 39.7, 1 x 2048 94.8 / 86.8, 1 x 3072 138 / 142, 1 x 4096 201 / 230, 2 x 1024 52.4 / 60.8,
  // 4 x 512 31.8 / 36.3, 8 x 512 60.0 / 56.0, 32 x 128 34.6 / 28.4, 32 x 256 81.3 / 67.7, 32 x 1024 572 / 621)
  const int flash_qw = 8 * 2 * (16 / p.gp);
  const long flash_wgs = (long)num_seqs * ((max_query_len + flash_qw - 1) / flash_qw) * num_kv_heads;
  if ((!kv_fp8 || head_size == 128) && max_query_len >= 32 && p.bs_shift >= 0 &&
      (max_query_len >= env_int("MI355Q_PREFILL_FLASH_MIN", 320) || flash_wgs >= env_int("MI355Q_PREFILL_FLASH_WGS", 64))) {
    if (dtype == MI355Q_BF16) return launch_prefill_flash<BF16>(p, head_size, kv_fp8, max_query_len, (hipStream_t)stream);
    return launch_prefill_flash<FP16>(p, head_size, kv_fp8, max_query_len, (hipStream_t)stream);
  }
  const int QB = 16 / p.gp;
  p.qblocks_per_seq = (max_query_len + QB - 1) / QB;
  const int qblocks = num_seqs * p.qblocks_per_seq;
  if (dtype == MI355Q_BF16) return launch_pa<BF16>(p, head_size, kv_fp8, qblocks, max_query_len, (hipStream_t)stream);
  return launch_pa<FP16>(p, head_size, kv_fp8, qblocks, max_query_len, (hipStream_t)stream);
}

extern "C" int mi355q_decode_attention_fused(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* qkv, int64_t qkv_stride,
    const int64_t* positions, const void* cos_sin_cache, void* key_cache, void* value_cache,
    int64_t kv_block_stride, int64_t kv_head_stride, int num_seqs, int num_heads, int num_kv_heads,
    int head_size, float scale, const int32_t* block_tables, int max_blocks_per_seq,
    const int32_t* seq_lens, int block_size, int max_seq_len, int dtype, int kv_fp8,
    const float* k_scale, const float* v_scale, mi355q_stream_t stream) {
  PAParams p;
  p.out = out; p.exp_sums = exp_sums; p.max_logits = max_logits; p.tmp_out = tmp_out;
  p.q = (const uint16_t*)qkv; p.q_stride = qkv_stride; p.kc = key_cache; p.vc = value_cache;
  p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.num_kv_heads = num_kv_heads; p.scale = scale;
  p.block_tables = block_tables; p.max_blocks = max_blocks_per_seq; p.seq_lens = seq_lens;
  p.qsl = nullptr; p.bs = block_size; p.alibi = nullptr; p.k_scale = k_scale;
  p.v_scale = v_scale; p.varlen = 0; p.qblocks_per_seq = 1;
  p.positions = positions; p.cos_sin = (const uint16_t*)cos_sin_cache;
  p.k_off = num_heads * head_size; p.v_off = (num_heads + num_kv_heads) * head_size;
  if (num_seqs == 0) return MI355Q_OK;
  if (int e = pa_common(p, head_size, block_size, max_seq_len, dtype, kv_fp8, "decode_attention_fused")) return e;
  MI355Q_REQUIRE(positions && cos_sin_cache, "decode_attention_fused: null pointer");
  MI355Q_REQUIRE(head_size == 128 || !kv_fp8, "decode_attention_fused: head_size 64 with an fp8 cache is not supported");
  MI355Q_REQUIRE(((uintptr_t)cos_sin_cache & 15) == 0, "decode_attention_fused: cos_sin_cache must be 16-byte aligned");
  if (dtype == MI355Q_BF16) return launch_pa<BF16>(p, head_size, kv_fp8, num_seqs, 1, (hipStream_t)stream, true);
  return launch_pa<FP16>(p, head_size, kv_fp8, num_seqs, 1, (hipStream_t)stream, true);
}

// ---------------------------------------------------------------------------
// Entry point that allocates nothing: the partition scratch of mi355q_paged_attention lives in a
// device array of the library.  Each GPU holds its own copy of the array at its own address, looked
// up once per device.
// ---------------------------------------------------------------------------
namespace mi355q {
constexpr size_t NA_ITEMS = MI355Q_PA_NOALLOC_MAX_ITEMS;   // (sequence, head, partition) triples
constexpr size_t NA_TMP_BYTES = NA_ITEMS * 128 * 2;         // tmp_out: head size <= 128, 16-bit
constexpr size_t NA_STAT_BYTES = NA_ITEMS * 4;              // exp_sums, max_logits: fp32 each
static __device__ __align__(256) unsigned char scratch[NA_TMP_BYTES + 2 * NA_STAT_BYTES];

constexpr int NA_MAX_DEVICES = 64;
static unsigned char* g_scratch_ptr[NA_MAX_DEVICES];
static std::mutex g_scratch_mu;

// device address of this GPU's copy of `scratch` (the array's name is not an address host code may use)
static int scratch_address(unsigned char** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NA_MAX_DEVICES)
    return set_error(MI355Q_EHIP, "paged_attention_noalloc: cannot identify the current device");
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
      return set_error(MI355Q_EHIP, "paged_attention_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
    g_scratch_ptr[dev] = (unsigned char*)ptr;
  }
  *out = g_scratch_ptr[dev];
  return MI355Q_OK;
}
}  // namespace mi355q

extern "C" int mi355q_paged_attention_noalloc(
    void* out, const void* query, int64_t q_stride, const void* key_cache, const void* value_cache,
    int64_t kv_block_stride, int64_t kv_head_stride, int num_seqs, int num_heads, int num_kv_heads,
    int head_size, float scale, const int32_t* block_tables, int max_blocks_per_seq,
    const int32_t* seq_lens, const int32_t* query_start_loc, int block_size, int max_seq_len,
    const float* alibi_slopes, int dtype, int kv_fp8, const float* k_scale, const float* v_scale,
    mi355q_stream_t stream) {
  if (num_seqs == 0) return MI355Q_OK;
  MI355Q_REQUIRE(num_seqs > 0 && num_heads > 0 && max_seq_len > 0, "paged_attention_noalloc: bad sizes");
  const size_t parts = ((size_t)max_seq_len + MI355Q_PA_PARTITION - 1) / MI355Q_PA_PARTITION;
  const size_t items = (size_t)num_seqs * (size_t)num_heads * parts;
  if (items > NA_ITEMS)
    return set_error(MI355Q_ENOSPACE,
                     "paged_attention_noalloc: %d sequences x %d heads x %zu partitions = %zu exceeds the built-in "
                     "scratch of %zu (use mi355q_paged_attention with caller-allocated scratch)",
                     num_seqs, num_heads, parts, items, (size_t)NA_ITEMS);
  unsigned char* base = nullptr;
  if (int e = scratch_address(&base)) return e;
  void* tmp_out = base;
  float* exp_sums = reinterpret_cast<float*>(base + NA_TMP_BYTES);
  float* max_logits = reinterpret_cast<float*>(base + NA_TMP_BYTES + NA_STAT_BYTES);
  return mi355q_paged_attention(out, exp_sums, max_logits, tmp_out, query, q_stride, key_cache, value_cache,
                                kv_block_stride, kv_head_stride, num_seqs, num_heads, num_kv_heads, head_size, scale,
                                block_tables, max_blocks_per_seq, seq_lens, query_start_loc, block_size, max_seq_len,
                                alibi_slopes, dtype, kv_fp8, k_scale, v_scale, stream);
}

/* Diagnostics (not part of the public ABI): per-workgroup 100 MHz timestamps of the attention kernel. */
extern "C" void mi355q_debug_attention_stamps(void* stamps) { g_pa_stamps = (unsigned long long*)stamps; }
