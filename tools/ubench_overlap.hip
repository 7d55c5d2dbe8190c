// Microbenchmark: can dependent weight-streaming kernels overlap their launch boundary on gfx950?
// A decode layer is a chain of kernels, each of which streams its own weights once and needs the small
// output vector of its predecessor.  Stream order makes kernel i+1 wait for the END of kernel i before it
// even starts, so the memory system idles for the boundary, the prologue and the first round trip of
// every launch.  Here consecutive kernels go to two alternating streams (two parallel chains in one
// graph) and the data dependency is carried by a counter in memory instead: kernel i+1 starts while
// kernel i is still running, requests its first ring of weights, and only then waits for the counter.
//   build: hipcc -O3 --offload-arch=gfx950 tools/ubench_overlap.hip -o tools/ubench_overlap
// Every spin loop gives up after a bounded time and records an error, so a runtime that serialises the
// two chains in the wrong order ends with a wrong result instead of a hang.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Link {
  const unsigned* flag_in;  // counter the predecessor's workgroups bump when their output is visible (null: none)
  unsigned need_in;         // its value once every one of them has
  unsigned* flag_out;       // this kernel's own counter (null: none)
  unsigned* err;            // set to 1 by a workgroup whose wait ran out
};

template <int DEPTH>
__global__ __launch_bounds__(512, 2) void ovl_k(const u32x4* __restrict__ w, long vecs_per_wg, const unsigned* x_in, int xwords,
                                                unsigned* x_out, Link lk, unsigned long long spin_limit) {
  __shared__ unsigned red[16];
  __shared__ unsigned s_x[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long per_wave = vecs_per_wg / nw;
  const u32x4* p = w + (long)blockIdx.x * vecs_per_wg + (long)wave * per_wave + lane;
  const long steps = per_wave / 64;
  u32x4 buf[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; i++) buf[i] = __builtin_nontemporal_load(p + (long)(i < steps ? i : 0) * 64);
  // wait for the predecessor, then read its output past this CU's L1 and this XCD's stale L2 lines
  if (lk.flag_in) {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(lk.flag_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lk.need_in) {
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > spin_limit) { *lk.err = 1; break; }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 256) s_x[threadIdx.x] = __hip_atomic_load(x_in + (threadIdx.x % xwords), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  unsigned acc = s_x[lane] ^ s_x[lane + 64] ^ s_x[lane + 128] ^ s_x[lane + 192];
  for (long s = 0; s < steps; s += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; i++) {
      u32x4 v = buf[i];
      if (s + i < steps) acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
      if (s + DEPTH + i < steps) buf[i] = __builtin_nontemporal_load(p + (s + DEPTH + i) * 64);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc ^= __shfl_xor(acc, o, 64);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int i = 0; i < nw; i++) t ^= red[i];
    __hip_atomic_store(x_out + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
    if (lk.flag_out) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(lk.flag_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The same chain as ONE persistent kernel per layer: every workgroup walks the stages; before it waits for
// the previous stage's counter it has already requested the first DEPTH KiB per wave of its slice of this
// stage's weights, so the memory system keeps streaming through the hand-off.  Dynamic LDS (unused) pins
// one workgroup per CU.
struct PStage { const u32x4* w; long vecs_per_wg; int wgs; };
struct PLayer { PStage st[5]; const unsigned* x_in; unsigned* xbuf; unsigned* flags; unsigned* err; unsigned long long spin_limit; };

template <int DEPTH>
__global__ __launch_bounds__(512) void persist_k(const PLayer L) {
  extern __shared__ unsigned pin_lds[];
  __shared__ unsigned red[16];
  __shared__ unsigned s_x[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int s = 0; s < 5; s++) {
    const PStage st = L.st[s];
    const bool active = (int)blockIdx.x < st.wgs;
    const long per_wave = st.vecs_per_wg / nw;
    const u32x4* p = st.w + (long)blockIdx.x * st.vecs_per_wg + (long)wave * per_wave + lane;
    const long steps = per_wave / 64;
    u32x4 buf[DEPTH];
    if (active) {
#pragma unroll
      for (int i = 0; i < DEPTH; i++) buf[i] = __builtin_nontemporal_load(p + (long)(i < steps ? i : 0) * 64);
    }
    const unsigned* xin = s == 0 ? L.x_in : L.xbuf + 256 * (s - 1);
    if (s > 0) {
      if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(L.flags + s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)L.st[s - 1].wgs) {
          __builtin_amdgcn_s_sleep(2);
          if (__builtin_amdgcn_s_memrealtime() - t0 > L.spin_limit) { *L.err = 1; break; }
        }
      }
      __syncthreads();
    }
    if (!active) continue;      // uniform per workgroup; it still took part in the wait above
    if (threadIdx.x < 256) s_x[threadIdx.x] = __hip_atomic_load(xin + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    unsigned acc = s_x[lane] ^ s_x[lane + 64] ^ s_x[lane + 128] ^ s_x[lane + 192];
    for (long k = 0; k < steps; k += DEPTH) {
#pragma unroll
      for (int i = 0; i < DEPTH; i++) {
        u32x4 v = buf[i];
        if (k + i < steps) acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        if (k + DEPTH + i < steps) buf[i] = __builtin_nontemporal_load(p + (k + DEPTH + i) * 64);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc ^= __shfl_xor(acc, o, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int i = 0; i < nw; i++) t ^= red[i];
      __hip_atomic_store(L.xbuf + 256 * s + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(L.flags + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // red and s_x are reused by the next stage
  }
}

// ---------------------------------------------------------------------------
// The persistent layer once more, with the hand-off and loader forms that cost least on this chip:
//   * one LOADER wave per CU streams the weights of every stage, one after the other and without regard to the
//     hand-offs, into a ring of 8 x 16 KiB LDS slots by LDS-DMA (16 x 1 KiB non-temporal loads per slot), three
//     fills in flight; a slot is published (FULL word in LDS) once a counted vmcnt says it has landed and reused
//     once its three consumers have released it (FREE word);
//   * three CONSUMER waves per CU reduce the slots (here: xor) -- the stand-in for the dequantise + MFMA work;
//   * the stage's result of a workgroup is ONE 8-byte granule {data, tag} written with a single write-through
//     store; the next stage's input is gathered by one consumer wave per CU that sweeps all granules with
//     write-through loads until every tag is the expected one (no counter, no fence, no second round trip);
//   * while its CU gathers, the loader keeps a single fill outstanding.
// One workgroup of 256 threads per CU (the ring pins it), one launch per layer.
// ---------------------------------------------------------------------------
struct EStage { const u32x4* w; int wgs, base_slots, extra; };   // workgroup b streams base_slots + (b < extra) slots
struct ELayer {
  EStage st[5];
  const unsigned long long* g_in;   // granules of the previous layer's last stage (null: none)
  int g_in_count;
  unsigned g_in_tag;
  unsigned long long* g_out;        // [5][256] granules of this layer
  unsigned tag0;                    // stage s publishes tag0 + s
  unsigned* err;
  unsigned long long spin_limit;
  int no_handoff;                   // diagnostics: consumers do not wait for their input (streaming rate of the ring alone)
};

#ifndef UB_INFLIGHT
#define UB_INFLIGHT 3
#endif
#ifndef UB_AUX
#define UB_AUX 2
#endif
constexpr int E_SLOTS = 8, E_SLOT_BYTES = 16384, E_INFLIGHT = UB_INFLIGHT;

__global__ __launch_bounds__(256) void engine_k(const ELayer L) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];   // E_SLOTS x 16 KiB
  __shared__ unsigned s_full[E_SLOTS], s_free[E_SLOTS];
  __shared__ unsigned s_x[256], s_red[4], s_done[5], s_ready[5], s_gather, s_abort;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  if (threadIdx.x < E_SLOTS) { s_full[threadIdx.x] = 0; s_free[threadIdx.x] = 0; }
  if (threadIdx.x < 5) { s_done[threadIdx.x] = 0; s_ready[threadIdx.x] = 0; }
  if (threadIdx.x == 0) { s_gather = 0; s_abort = 0; }
  __syncthreads();
  volatile unsigned* v_full = s_full;
  volatile unsigned* v_free = s_free;
  volatile unsigned* v_done = s_done;
  volatile unsigned* v_ready = s_ready;
  volatile unsigned* v_gather = &s_gather;
  volatile unsigned* v_abort = &s_abort;
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  auto expired = [&]() {
    if (*v_abort) return true;
    if (__builtin_amdgcn_s_memrealtime() - t_start > L.spin_limit) { *v_abort = 1; *L.err = 1; return true; }
    return false;
  };

  if (wave == 0) {
    // ---- loader ---------------------------------------------------------------------------------------------
    unsigned q = 0;          // slots issued so far by this workgroup
    unsigned published = 0;  // slots published so far
    for (int s = 0; s < 5; s++) {
      const EStage st = L.st[s];
      if (b >= st.wgs) continue;
      const int slots = st.base_slots + (b < st.extra ? 1 : 0);
      const size_t first = (size_t)b * st.base_slots + (b < st.extra ? b : st.extra);
      const u32x4* src = st.w + first * (E_SLOT_BYTES / 16) + lane;
      for (int i = 0; i < slots; i++, q++) {
        const unsigned slot = q % E_SLOTS, round = q / E_SLOTS;
        // the slot's previous contents must have been released by all three consumers
        while (v_free[slot] < 3 * round) { __builtin_amdgcn_s_sleep(1); if (expired()) return; }
        unsigned char* dst = ring + (size_t)slot * E_SLOT_BYTES;
#pragma unroll
        for (int r = 0; r < 16; r++)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)i * 1024 + r * 64),
                                           (void __attribute__((address_space(3)))*)(dst + r * 1024), 16, 0, UB_AUX);
        // publish what has certainly landed: all but the newest fills
        if (*v_gather) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          while (published <= q) { v_full[published % E_SLOTS] = published / E_SLOTS + 1; published++; }
        } else if (q + 1 >= E_INFLIGHT) {
          // 16 loads per fill: all but the newest E_INFLIGHT - 1 fills have landed
          if constexpr (E_INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else if constexpr (E_INFLIGHT == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
          while (published + (E_INFLIGHT - 1) <= q) { v_full[published % E_SLOTS] = published / E_SLOTS + 1; published++; }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    while (published < q) { v_full[published % E_SLOTS] = published / E_SLOTS + 1; published++; }
    return;
  }

  // ---- consumers ----------------------------------------------------------------------------------------------
  const int c = wave - 1;
  unsigned q = 0;
  for (int s = 0; s < 5; s++) {
    const EStage st = L.st[s];
    if (b >= st.wgs) continue;
    // input of the stage: the granules of the stage before (of the previous launch for stage 0)
    const unsigned long long* gin = L.no_handoff ? nullptr : (s == 0 ? L.g_in : L.g_out + 256 * (s - 1));
    const int gcount = s == 0 ? L.g_in_count : L.st[s - 1].wgs;
    const unsigned gtag = s == 0 ? L.g_in_tag : L.tag0 + s - 1;
    if (c == 0) {
      unsigned d[4] = {0, 0, 0, 0};
      if (gin) {
        *v_gather = 1;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int gi = lane + 64 * j;
            const unsigned long long g = __hip_atomic_load(gin + (gi < gcount ? gi : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d[j] = gi < gcount ? (unsigned)g : 0u;
            ok = ok && (unsigned)(g >> 32) == gtag;
          }
          if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
          __builtin_amdgcn_s_sleep(1);
          if (expired()) break;
        }
        *v_gather = 0;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) s_x[lane + 64 * j] = d[j];
      __builtin_amdgcn_s_waitcnt(0xc07f);   // the LDS writes above are done (lgkmcnt(0))
      if (lane == 0) v_ready[s] = 1;
    } else {
      while (v_ready[s] == 0) { __builtin_amdgcn_s_sleep(1); if (expired()) break; }
    }
    unsigned acc = c == 0 ? (s_x[lane] ^ s_x[lane + 64] ^ s_x[lane + 128] ^ s_x[lane + 192]) : 0u;
    const int slots = st.base_slots + (b < st.extra ? 1 : 0);
    for (int i = 0; i < slots; i++, q++) {
      const unsigned slot = q % E_SLOTS, round = q / E_SLOTS;
      while (v_full[slot] < round + 1) { __builtin_amdgcn_s_sleep(1); if (expired()) break; }
      const u32x4* src = reinterpret_cast<const u32x4*>(ring + (size_t)slot * E_SLOT_BYTES) + lane;
      u32x4 v[6];
#pragma unroll
      for (int r = 0; r < 6; r++) {
        const int row = c + 3 * r;
        v[r] = row < 16 ? src[row * 64] : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int r = 0; r < 6; r++) acc ^= v[r][0] ^ v[r][1] ^ v[r][2] ^ v[r][3];
      asm volatile("" :: "v"(acc) : "memory");   // the slot's words are in registers before it is released
      if (lane == 0) atomicAdd(&s_free[slot], 1u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc ^= __shfl_xor(acc, o, 64);
    if (lane == 0) {
      s_red[c] = acc;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      atomicAdd(&s_done[s], 1u);
    }
    if (c == 0) {
      while (v_done[s] < 3) { __builtin_amdgcn_s_sleep(1); if (expired()) break; }
      if (lane == 0) {
        const unsigned t = ((volatile unsigned*)s_red)[0] ^ ((volatile unsigned*)s_red)[1] ^ ((volatile unsigned*)s_red)[2];
        const unsigned long long g = ((unsigned long long)(L.tag0 + s) << 32) | t;
        __hip_atomic_store(L.g_out + 256 * s + b, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one write-through store
      }
    }
    // s_red is reused by the next stage: consumers 1 and 2 wait for that stage's ready word first, which consumer 0
    // sets after it has read s_red and published
  }
}

// persist_k with the tagged-granule hand-off instead of arrival counters: every wave streams through its register
// ring as there, workgroup outputs are 8-byte {data, tag} granules, wave 0 sweeps the previous stage's granules.
struct GLayer { PStage st[5]; int out_g[5]; /* granules a stage publishes in all (0: one per workgroup) */ const unsigned long long* g_in; int g_in_count; unsigned g_in_tag; unsigned long long* g_out; unsigned tag0;
                unsigned* err; unsigned long long spin_limit; };

template <int DEPTH>
__global__ __launch_bounds__(512) void persist_granule_k(const GLayer L) {
  extern __shared__ unsigned pin_lds[];
  __shared__ unsigned red[16];
  __shared__ unsigned s_x[256];
  __shared__ unsigned s_big[7168];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int s = 0; s < 5; s++) {
    const PStage st = L.st[s];
    const bool active = (int)blockIdx.x < st.wgs;
    if (!active) continue;      // nobody waits for a workgroup that produces nothing
    const long per_wave = st.vecs_per_wg / nw;
    const u32x4* p = st.w + (long)blockIdx.x * st.vecs_per_wg + (long)wave * per_wave + lane;
    const long steps = per_wave / 64;
    u32x4 buf[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; i++) buf[i] = __builtin_nontemporal_load(p + (long)(i < steps ? i : 0) * 64);
    const unsigned long long* gin = s == 0 ? L.g_in : L.g_out + (L.out_g[0] > 0 ? 7168 : 256) * (s - 1);
    const int gcount = s == 0 ? L.g_in_count : L.st[s - 1].wgs;
    const unsigned gtag = s == 0 ? L.g_in_tag : L.tag0 + s - 1;
    if (L.out_g[0] > 0) {
      // payloads of real size: every thread sweeps its share of the previous stage's granules until their tags match
      const int total = gin ? (s == 0 ? L.out_g[4] : L.out_g[s - 1]) : 0;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int base = 0; base < total; base += 512 * 4) {
        unsigned long long g[4];
        bool ok = false;
        while (!ok) {
          ok = true;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int gi = base + threadIdx.x + 512 * j;
            g[j] = __hip_atomic_load(gin + (gi < total ? gi : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (unsigned)(g[j] >> 32) == gtag;
          }
          if (!ok) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > L.spin_limit) { *L.err = 1; break; }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int gi = base + threadIdx.x + 512 * j;
          if (gi < total) s_big[gi] = (unsigned)g[j];
        }
      }
      if (!gin && threadIdx.x < 256) s_big[threadIdx.x] = 0;
      __syncthreads();
      if (threadIdx.x < 256) {
        unsigned v = 0;
        for (int i = threadIdx.x; i < (total ? total : 256); i += 256) v ^= s_big[i];
        s_x[threadIdx.x] = v;
      }
    } else
    if (wave == 0) {
      unsigned d[4] = {0, 0, 0, 0};
      if (gin) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int gi = lane + 64 * j;
            const unsigned long long g = __hip_atomic_load(gin + (gi < gcount ? gi : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d[j] = gi < gcount ? (unsigned)g : 0u;
            ok = ok && (unsigned)(g >> 32) == gtag;
          }
          if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
          __builtin_amdgcn_s_sleep(1);
          if (__builtin_amdgcn_s_memrealtime() - t0 > L.spin_limit) { *L.err = 1; break; }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) s_x[lane + 64 * j] = d[j];
    }
    __syncthreads();
    unsigned acc = s_x[lane] ^ s_x[lane + 64] ^ s_x[lane + 128] ^ s_x[lane + 192];
    for (long k = 0; k < steps; k += DEPTH) {
#pragma unroll
      for (int i = 0; i < DEPTH; i++) {
        u32x4 v = buf[i];
        if (k + i < steps) acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        if (k + DEPTH + i < steps) buf[i] = __builtin_nontemporal_load(p + (k + DEPTH + i) * 64);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc ^= __shfl_xor(acc, o, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int i = 0; i < nw; i++) t ^= red[i];
      if (L.out_g[0] == 0)
        __hip_atomic_store(L.g_out + 256 * s + blockIdx.x, ((unsigned long long)(L.tag0 + s) << 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      red[0] = t;
    }
    __syncthreads();
    if (L.out_g[0] > 0) {
      const int per = L.out_g[s] / st.wgs;     // this workgroup's share of the stage's output
      if ((int)threadIdx.x < per)
        __hip_atomic_store(L.g_out + 7168 * s + (size_t)blockIdx.x * per + threadIdx.x,
                           ((unsigned long long)(L.tag0 + s) << 32) | (red[0] + threadIdx.x * 2654435761u), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // red and s_x are reused by the next stage
  }
}

struct Stage { double mb; int wgs; };

// mode 0: one stream, no counters (stream order carries the dependency)
// mode 1: one stream, counters (what the protocol itself costs when nothing overlaps)
// mode 2: two alternating streams, counters
template <int DEPTH>
static float run(hipStream_t sa, hipStream_t sb, const u32x4* weights, size_t total_vecs, unsigned* xbuf, unsigned* flags, unsigned* err,
                 const std::vector<Stage>& stages, int layers, int mode, double* checksum) {
  hipGraph_t g; hipGraphExec_t ge;
  hipEvent_t fork, join; hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreateWithFlags(&join, hipEventDisableTiming);
  const int nk = layers * (int)stages.size();
  hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal);
  hipMemsetAsync(flags, 0, sizeof(unsigned) * (nk + 1), sa);
  if (mode == 2) { hipEventRecord(fork, sa); hipStreamWaitEvent(sb, fork, 0); }
  size_t off = 0;
  int k = 0, prev_wgs = 0;
  for (int l = 0; l < layers; l++)
    for (const Stage& s : stages) {
      long vpw = (long)(s.mb * 1e6 / 16 / s.wgs) / (8 * 64) * (8 * 64);
      if (vpw == 0) vpw = 8 * 64;
      if (off + (size_t)vpw * s.wgs > total_vecs) off = 0;
      hipStream_t st = (mode == 2 && (k & 1)) ? sb : sa;
      Link lk = {nullptr, 0, nullptr, err};
      if (mode >= 1) {
        lk.flag_in = k ? flags + k - 1 : nullptr;
        lk.need_in = prev_wgs;
        lk.flag_out = flags + k;
      }
      // x ping-pongs over three buffers: a kernel two places down the chain may start while the reader of
      // the buffer it writes is still running only if they share a stream, which two buffers would allow
      unsigned* xin = xbuf + 256 * (k % 3);
      unsigned* xout = xbuf + 256 * ((k + 1) % 3);
      hipLaunchKernelGGL((ovl_k<DEPTH>), dim3(s.wgs), dim3(512), 0, st, weights + off, vpw, xin, 256, xout, lk, 100ull * 100000ull /* 100 ms */);
      off += (size_t)vpw * s.wgs;
      prev_wgs = s.wgs;
      k++;
    }
  if (mode == 2) { hipEventRecord(join, sb); hipStreamWaitEvent(sa, join, 0); }
  hipStreamEndCapture(sa, &g);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return -1.f; }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; i++) hipGraphLaunch(ge, sa);
  hipStreamSynchronize(sa);
  hipEventRecord(a, sa);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, sa);
  hipEventRecord(b, sa);
  hipStreamSynchronize(sa);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned h[768]; hipMemcpy(h, xbuf, sizeof(h), hipMemcpyDeviceToHost);
  double cs = 0; for (int i = 0; i < 768; i++) cs += h[i] * (i + 1.0);
  *checksum = cs;
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * layers);
}

template <int DEPTH>
static float run_persist(hipStream_t sa, const u32x4* weights, size_t total_vecs, unsigned* xbuf, unsigned* flags, unsigned* err,
                         const std::vector<Stage>& stages, int layers, double* checksum) {
  hipGraph_t g; hipGraphExec_t ge;
  hipFuncSetAttribute((const void*)persist_k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal);
  hipMemsetAsync(flags, 0, sizeof(unsigned) * 8 * layers, sa);
  size_t off = 0;
  for (int l = 0; l < layers; l++) {
    PLayer L;
    int maxw = 0;
    for (int k = 0; k < 5; k++) {
      long vpw = (long)(stages[k].mb * 1e6 / 16 / stages[k].wgs) / (8 * 64) * (8 * 64);
      if (vpw == 0) vpw = 8 * 64;
      if (off + (size_t)vpw * stages[k].wgs > total_vecs) off = 0;
      L.st[k] = PStage{weights + off, vpw, stages[k].wgs};
      off += (size_t)vpw * stages[k].wgs;
      maxw = stages[k].wgs > maxw ? stages[k].wgs : maxw;
    }
    // layer l reads the last stage's output of layer l-1 (a kernel boundary lies between them)
    L.x_in = xbuf + 2048 * ((l + 1) & 1) + 256 * 4;
    L.xbuf = xbuf + 2048 * (l & 1);
    L.flags = flags + 8 * l;
    L.err = err;
    L.spin_limit = 100ull * 100000ull;
    hipLaunchKernelGGL((persist_k<DEPTH>), dim3(maxw), dim3(512), 96 * 1024, sa, L);
  }
  hipStreamEndCapture(sa, &g);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return -1.f; }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; i++) hipGraphLaunch(ge, sa);
  hipStreamSynchronize(sa);
  hipEventRecord(a, sa);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, sa);
  hipEventRecord(b, sa);
  hipStreamSynchronize(sa);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned h[4096]; hipMemcpy(h, xbuf, sizeof(h), hipMemcpyDeviceToHost);
  double cs = 0; for (int i = 0; i < 4096; i++) cs += h[i] * (i + 1.0);
  *checksum = cs;
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * layers);
}

template <int DEPTH>
static float run_persist_granule(hipStream_t sa, const u32x4* weights, size_t total_vecs, unsigned long long* gran, unsigned* err,
                                 const std::vector<Stage>& stages, int layers, double* checksum, bool real = false) {
  // outputs of real size, in granules of two 16-bit values: qkv 6144, attention 4096, o 4096, gate_up (after SiLU) 14336, down 4096
  const int real_g[5] = {3072, 2048, 2048, 7168, 2048};
  const size_t lstride = real ? 5 * 7168 : 5 * 256;
  hipGraph_t g; hipGraphExec_t ge;
  hipFuncSetAttribute((const void*)persist_granule_k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal);
  hipMemsetAsync(gran, 0, sizeof(unsigned long long) * lstride * layers, sa);
  size_t off = 0;
  for (int l = 0; l < layers; l++) {
    GLayer L;
    for (int k = 0; k < 5; k++) L.out_g[k] = real ? real_g[k] : 0;
    int maxw = 0;
    for (int k = 0; k < 5; k++) {
      long vpw = (long)(stages[k].mb * 1e6 / 16 / stages[k].wgs) / (8 * 64) * (8 * 64);
      if (vpw == 0) vpw = 8 * 64;
      if (off + (size_t)vpw * stages[k].wgs > total_vecs) off = 0;
      L.st[k] = PStage{weights + off, vpw, stages[k].wgs};
      off += (size_t)vpw * stages[k].wgs;
      maxw = stages[k].wgs > maxw ? stages[k].wgs : maxw;
    }
    L.g_in = l ? gran + (size_t)(l - 1) * lstride + 4 * (real ? 7168 : 256) : nullptr;
    L.g_in_count = stages[4].wgs;
    L.g_in_tag = l ? (unsigned)(8 * (l - 1) + 1 + 4) : 0u;
    L.g_out = gran + (size_t)l * lstride;
    L.tag0 = (unsigned)(8 * l + 1);
    L.err = err;
    L.spin_limit = 100ull * 20000ull;
    hipLaunchKernelGGL((persist_granule_k<DEPTH>), dim3(maxw), dim3(512), 96 * 1024, sa, L);
  }
  hipStreamEndCapture(sa, &g);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return -1.f; }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; i++) hipGraphLaunch(ge, sa);
  hipStreamSynchronize(sa);
  hipEventRecord(a, sa);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, sa);
  hipEventRecord(b, sa);
  hipStreamSynchronize(sa);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(lstride * layers);
  hipMemcpy(h.data(), gran, h.size() * 8, hipMemcpyDeviceToHost);
  double cs = 0; for (size_t i = 0; i < h.size(); i++) cs += (double)(unsigned)h[i] * ((i % 4096) + 1.0);
  *checksum = cs;
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * layers);
}

static float run_engine(hipStream_t sa, const u32x4* weights, size_t total_vecs, unsigned long long* gran, unsigned* err,
                        const std::vector<Stage>& stages, int layers, double* checksum, int no_handoff = 0) {
  hipGraph_t g; hipGraphExec_t ge;
  const size_t lds = (size_t)E_SLOTS * E_SLOT_BYTES;
  if (hipFuncSetAttribute((const void*)engine_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("lds attr failed\n"); return -1.f; }
  hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal);
  hipMemsetAsync(gran, 0, sizeof(unsigned long long) * 5 * 256 * layers, sa);
  size_t off = 0;
  for (int l = 0; l < layers; l++) {
    ELayer L;
    int maxw = 0;
    for (int k = 0; k < 5; k++) {
      const long total_slots = (long)(stages[k].mb * 1e6 / E_SLOT_BYTES + 0.5);
      const int wgs = stages[k].wgs;
      const size_t vecs = (size_t)total_slots * (E_SLOT_BYTES / 16);
      if (off + vecs > total_vecs) off = 0;
      L.st[k] = EStage{weights + off, wgs, (int)(total_slots / wgs), (int)(total_slots % wgs)};
      off += vecs;
      maxw = wgs > maxw ? wgs : maxw;
    }
    L.g_in = l ? gran + (size_t)(l - 1) * 5 * 256 + 4 * 256 : nullptr;
    L.g_in_count = stages[4].wgs;
    L.g_in_tag = l ? (unsigned)(8 * (l - 1) + 1 + 4) : 0u;
    L.g_out = gran + (size_t)l * 5 * 256;
    L.tag0 = (unsigned)(8 * l + 1);
    L.err = err;
    L.spin_limit = 100ull * 20000ull;   // 20 ms
    L.no_handoff = no_handoff;
    hipLaunchKernelGGL(engine_k, dim3(maxw), dim3(256), lds, sa, L);
  }
  hipStreamEndCapture(sa, &g);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return -1.f; }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; i++) hipGraphLaunch(ge, sa);
  hipStreamSynchronize(sa);
  hipEventRecord(a, sa);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, sa);
  hipEventRecord(b, sa);
  hipStreamSynchronize(sa);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h((size_t)5 * 256 * layers);
  hipMemcpy(h.data(), gran, h.size() * 8, hipMemcpyDeviceToHost);
  double cs = 0; for (size_t i = 0; i < h.size(); i++) cs += (double)(unsigned)h[i] * ((i % 4096) + 1.0);
  *checksum = cs;
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * layers);
}

int main() {
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const size_t bytes = (size_t)3600 << 20;
  u32x4* w; CK(hipMalloc(&w, bytes));
  // distinct words, so that a kernel that ran on a stale input vector changes the checksum
  {
    std::vector<unsigned> h(bytes / 4 / 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)(i * 2654435761u) ^ 0x5bd1e995u;
    for (int r = 0; r < 64; r++) CK(hipMemcpy((char*)w + (size_t)r * h.size() * 4, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  unsigned *xbuf, *flags, *err;
  CK(hipMalloc(&xbuf, 16384)); CK(hipMemset(xbuf, 0, 16384));
  CK(hipMalloc(&flags, 8192)); CK(hipMemset(flags, 0, 8192)); err = flags + 1500;
  CK(hipDeviceSynchronize());
  const size_t total_vecs = bytes / 16;
  const int layers = 32;
  std::vector<Stage> layer = {{13.0, 192}, {0.9, 8}, {8.65, 256}, {60.6, 256}, {30.3, 256}};
  const char* names[3] = {"one stream, stream order", "one stream, counters", "two streams, counters"};
#define RUN(D)                                                                                                           \
  for (int mode = 0; mode < 2; mode++) {                                                                                 \
    double cs;                                                                                                           \
    float t = run<D>(sa, sb, w, total_vecs, xbuf, flags, err, layer, layers, mode, &cs);                                 \
    unsigned ran_out = 0; CK(hipMemcpy(&ran_out, err, 4, hipMemcpyDeviceToHost));                                                        \
    printf("depth %2d  %-26s layer %6.2f us  checksum %.0f  wait-ran-out %u\n", D, names[mode], t, cs, ran_out);               \
    fflush(stdout);                                                                                                      \
    if (ran_out) { fflush(stdout); _exit(2); }                                                                                                   \
  }
  RUN(4);
  RUN(14);
#define RUNP(D)                                                                                              \
  {                                                                                                          \
    double cs;                                                                                               \
    float t = run_persist<D>(sa, w, total_vecs, xbuf, flags, err, layer, layers, &cs);                        \
    unsigned ran_out = 0; CK(hipMemcpy(&ran_out, err, 4, hipMemcpyDeviceToHost));                            \
    printf("depth %2d  one persistent kernel per layer  layer %6.2f us  checksum %.0f  wait-ran-out %u\n", D, t, cs, ran_out); \
    fflush(stdout);                                                                                          \
    if (ran_out) { _exit(2); }                                                                               \
  }
  RUNP(4);
  RUNP(8);
  RUNP(14);
  {
    unsigned long long* gran; CK(hipMalloc(&gran, sizeof(unsigned long long) * 5 * 7168 * layers));
    double cs, cs2;
    float t = run_engine(sa, w, total_vecs, gran, err, layer, layers, &cs);
    unsigned ran_out = 0; CK(hipMemcpy(&ran_out, err, 4, hipMemcpyDeviceToHost));
    printf("LDS-DMA ring (1 loader + 3 consumer waves per CU), 8-byte tagged granules, one launch per layer  layer %6.2f us  checksum %.0f  wait-ran-out %u\n", t, cs, ran_out);
    fflush(stdout);
    if (ran_out) _exit(2);
    t = run_engine(sa, w, total_vecs, gran, err, layer, layers, &cs2);
    printf("  again: layer %6.2f us  checksum %s\n", t, cs2 == cs ? "equal" : "DIFFERENT");
#define RUNG(D)                                                                                                      \
    {                                                                                                                \
      double c3;                                                                                                     \
      float tg = run_persist_granule<D>(sa, w, total_vecs, gran, err, layer, layers, &c3);                            \
      CK(hipMemcpy(&ran_out, err, 4, hipMemcpyDeviceToHost));                                                         \
      printf("depth %2d  one persistent kernel per layer, register rings, tagged granules  layer %6.2f us  checksum %.0f  wait-ran-out %u\n", D, tg, c3, ran_out); \
      fflush(stdout);                                                                                                \
      if (ran_out) _exit(2);                                                                                         \
    }
    RUNG(4);
    RUNG(8);
    RUNG(14);
#define RUNGR(D)                                                                                                     \
    {                                                                                                                \
      double c3;                                                                                                     \
      float tg = run_persist_granule<D>(sa, w, total_vecs, gran, err, layer, layers, &c3, true);                      \
      CK(hipMemcpy(&ran_out, err, 4, hipMemcpyDeviceToHost));                                                         \
      printf("depth %2d  the same with stage outputs of real size (24 / 16 / 16 / 57 / 16 KB of granules, every workgroup sweeps all)  layer %6.2f us  checksum %.0f  wait-ran-out %u\n", D, tg, c3, ran_out); \
      fflush(stdout);                                                                                                \
      if (ran_out) _exit(2);                                                                                         \
    }
    RUNGR(4);
    RUNGR(14);
    t = run_engine(sa, w, total_vecs, gran, err, layer, layers, &cs2, 1);
    printf("  the ring alone (consumers do not wait for their inputs): layer %6.2f us\n", t);
  }
  return 0;
}
