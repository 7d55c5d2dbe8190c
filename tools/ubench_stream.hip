// Microbenchmark: the floor of a decode layer built from dependent weight-streaming launches on gfx950.
// Every kernel streams its own slice of weights once (nt loads, 16 B per lane), depends on a small
// vector written by the previous kernel, and writes a small vector for the next one.  No arithmetic
// beyond an XOR per load, so what is measured is launch boundary + first-byte latency + stream + tail.
//   build: hipcc -O3 --offload-arch=gfx950 tools/ubench_stream.hip -o tools/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// DEPTH: 1 KiB loads in flight per wave.  FIRST: how many of them are requested before x has landed.
// XDEP: 1 = nothing is consumed before x has landed (GEMV semantics), 0 = x only joins at the end.
template <int DEPTH, int FIRST, int XDEP, int WORK>
__global__ __launch_bounds__(512) void stream_k(const u32x4* __restrict__ w, long vecs_per_wg, const u32x4* x_in, int xvecs,
                                                unsigned* x_out, int delay) {
  __shared__ unsigned red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long per_wave = vecs_per_wg / nw;            // 16-byte vectors per wave, multiple of 64 * DEPTH
  // delay bit 1: the waves of a workgroup read adjacent 1 KiB pieces (piece s of wave v at (s nw + v) KiB)
  // instead of one contiguous range each
  const bool ilv = delay & 2;
  const long sstride = ilv ? nw : 1;
  const u32x4* p = w + (long)blockIdx.x * vecs_per_wg + (ilv ? (long)wave * 64 : (long)wave * per_wave) + lane;
  const long steps = per_wave / 64;                   // 1 KiB loads per wave
  u32x4 xv = x_in[(threadIdx.x) % xvecs];             // the small dependent operand, requested first
  u32x4 buf[DEPTH];
  if (delay & 1) __builtin_amdgcn_s_sleep(12);
#pragma unroll
  for (int i = 0; i < FIRST; i++) buf[i] = __builtin_nontemporal_load(p + (long)(i < steps ? i : 0) * sstride * 64);
  unsigned acc = 0;
  if (XDEP) {
    acc = xv[0] ^ xv[1] ^ xv[2] ^ xv[3];              // waits for x (and nothing newer)
    asm volatile("" : "+v"(acc));
  }
#pragma unroll
  for (int i = FIRST; i < DEPTH; i++) buf[i] = __builtin_nontemporal_load(p + (long)(i < steps ? i : 0) * sstride * 64);
  for (long s = 0; s < steps; s += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; i++) {
      u32x4 v = buf[i];
      unsigned t = v[0] ^ v[1] ^ v[2] ^ v[3];
#pragma unroll
      for (int k = 0; k < WORK; k++) t = t * 2654435761u + acc;   // stand-in for the dequant arithmetic
      if (s + i < steps) acc ^= t;
      if (s + DEPTH + i < steps) buf[i] = __builtin_nontemporal_load(p + (s + DEPTH + i) * sstride * 64);
    }
  }
  if (!XDEP) acc ^= xv[0] ^ xv[1] ^ xv[2] ^ xv[3];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc ^= __shfl_xor(acc, o, 64);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int i = 0; i < nw; i++) t ^= red[i];
    x_out[blockIdx.x * 4] = t;
  }
}

// The same stream through buffer loads, whose cache-policy bits can be chosen freely:
// AUX bit 0 = sc0, bit 1 = nt, bit 4 = sc1.
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int DEPTH, int AUX>
__global__ __launch_bounds__(512) void stream_buf_k(const u32x4* __restrict__ w, long vecs_per_wg, const u32x4* x_in, int xvecs,
                                                    unsigned* x_out, int delay) {
  __shared__ unsigned red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long per_wave = vecs_per_wg / nw;
  const u32x4* base = w + (long)blockIdx.x * vecs_per_wg + (long)wave * per_wave;
  const long steps = per_wave / 64;
  // buffer resource over this wave's range: base, stride 0, num_records in bytes, raw 32-bit data format
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(per_wave * 16), 0x00020000);
  u32x4 xv = x_in[(threadIdx.x) % xvecs];
  u32x4 buf[DEPTH];
  const int voff = lane * 16;
#pragma unroll
  for (int i = 0; i < DEPTH; i++) {
    i32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (int)((i < steps ? i : 0) * 1024), AUX);
    buf[i] = u32x4{(unsigned)t[0], (unsigned)t[1], (unsigned)t[2], (unsigned)t[3]};
  }
  unsigned acc = xv[0] ^ xv[1] ^ xv[2] ^ xv[3];
  asm volatile("" : "+v"(acc));
  for (long s = 0; s < steps; s += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; i++) {
      u32x4 v = buf[i];
      if (s + i < steps) acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
      if (s + DEPTH + i < steps) {
        i32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (int)((s + DEPTH + i) * 1024), AUX);
        buf[i] = u32x4{(unsigned)t[0], (unsigned)t[1], (unsigned)t[2], (unsigned)t[3]};
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc ^= __shfl_xor(acc, o, 64);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int i = 0; i < nw; i++) t ^= red[i];
    x_out[blockIdx.x * 4] = t;
  }
}

template <int DEPTH, int AUX>
static float run_buf(hipStream_t st, const u32x4* weights, size_t total_vecs, unsigned* xa, unsigned* xb, double mb, int wgs, int layers) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  size_t off = 0;
  int flip = 0;
  for (int l = 0; l < layers; l++) {
    long vecs_per_wg = (long)(mb * 1e6 / 16 / wgs) / (8 * 64) * (8 * 64);
    if (off + (size_t)vecs_per_wg * wgs > total_vecs) off = 0;
    hipLaunchKernelGGL((stream_buf_k<DEPTH, AUX>), dim3(wgs), dim3(512), 0, st, weights + off, vecs_per_wg,
                       (const u32x4*)(flip ? xb : xa), 256, flip ? xa : xb, 0);
    off += (size_t)vecs_per_wg * wgs;
    flip ^= 1;
  }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, st);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * layers);
}

// A latency-bound stage (few workgroups, like decode attention at short contexts) whose launch carries
// extra workgroups that do nothing but read the weights of the FOLLOWING stages into the cache hierarchy:
// HBM would otherwise idle for the whole stage.  Consumer workgroup j of a following stage reads the
// contiguous slice j of its weights and shares an XCD (and its L2) with every block b = j mod 8 of the
// same launch; SHIFT rotates that assignment to tell L2 hits from memory-side cache hits.
__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u; }  // HW_REG_XCC_ID, 4 bits

struct PfTarget { const u32x4* base; long vecs_per_consumer; int consumers; long vecs_each; };

template <int DEPTH>
__global__ __launch_bounds__(512) void small_pf_k(const u32x4* __restrict__ w, long vecs_per_wg, int nsmall, const u32x4* x_in, int xvecs,
                                                  unsigned* x_out, PfTarget pf0, PfTarget pf1, int shift, unsigned* xcc_log) {
  __shared__ unsigned red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (xcc_log && threadIdx.x == 0 && blockIdx.x < 16) xcc_log[blockIdx.x] = xcc_id();
  unsigned acc = 0;
  if ((int)blockIdx.x >= nsmall) {
    // prefetch role: block b serves the consumers j = (b + shift) mod 8, in turn
    const int b = blockIdx.x - nsmall, npf = gridDim.x - nsmall;
    const int cls = (blockIdx.x + shift) & 7, ord = b >> 3, per_cls = npf >> 3;
    for (int t = 0; t < 2; t++) {
      const PfTarget pf = t ? pf1 : pf0;
      for (int j = cls + 8 * ord; j < pf.consumers; j += 8 * per_cls) {
        const u32x4* src = pf.base + (long)j * pf.vecs_per_consumer;
        for (long v = threadIdx.x; v < pf.vecs_each; v += 4 * 512) {
          u32x4 a = src[v], b2 = v + 512 < pf.vecs_each ? src[v + 512] : a, c = v + 1024 < pf.vecs_each ? src[v + 1024] : a,
                d = v + 1536 < pf.vecs_each ? src[v + 1536] : a;
          acc ^= a[0] ^ b2[1] ^ c[2] ^ d[3];
        }
      }
    }
    if (acc == 0x9e3779b9u) x_out[4096 + blockIdx.x] = acc;   // never true in practice: keeps the loads alive
    return;
  }
  const long per_wave = vecs_per_wg / nw;
  const u32x4* p = w + (long)blockIdx.x * vecs_per_wg + (long)wave * per_wave + lane;
  const long steps = per_wave / 64;
  u32x4 xv = x_in[(threadIdx.x) % xvecs];
  acc = xv[0] ^ xv[1] ^ xv[2] ^ xv[3];
  asm volatile("" : "+v"(acc));
  u32x4 buf[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH; i++) buf[i] = __builtin_nontemporal_load(p + (long)(i < steps ? i : 0) * 64);
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
    x_out[blockIdx.x * 4] = t;
  }
}

// One layer = qkv, small(+prefetch), o, gate_up, down with DEPTH 1-KiB loads in flight per wave.
// pf_o / pf_gu: MB of the o and gate_up weights that the small stage's extra workgroups read ahead.
template <int DEPTH>
static float run_pf(hipStream_t st, const u32x4* weights, size_t total_vecs, unsigned* xa, unsigned* xb, int layers, int small_wgs,
                    double small_mb, int pf_wgs, double pf_o_frac, double pf_gu_frac, int shift, unsigned* xcc_log) {
  struct S { double mb; int wgs; };
  const S st_[5] = {{13.0, 192}, {small_mb, small_wgs}, {8.65, 256}, {60.6, 256}, {30.3, 256}};
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  size_t off = 0;
  int flip = 0;
  for (int l = 0; l < layers; l++) {
    long vpw[5]; size_t offs[5];
    for (int k = 0; k < 5; k++) {
      vpw[k] = (long)(st_[k].mb * 1e6 / 16 / st_[k].wgs) / (8 * 64) * (8 * 64);
      if (vpw[k] == 0) vpw[k] = 8 * 64;
    }
    size_t need = 0;
    for (int k = 0; k < 5; k++) need += (size_t)vpw[k] * st_[k].wgs;
    if (off + need > total_vecs) off = 0;
    for (int k = 0; k < 5; k++) { offs[k] = off; off += (size_t)vpw[k] * st_[k].wgs; }
    for (int k = 0; k < 5; k++) {
      const u32x4* xin = (const u32x4*)(flip ? xb : xa);
      unsigned* xout = flip ? xa : xb;
      if (k == 1) {
        const PfTarget t0 = {weights + offs[2], vpw[2], st_[2].wgs, (long)(vpw[2] * pf_o_frac)};
        const PfTarget t1 = {weights + offs[3], vpw[3], st_[3].wgs, (long)(vpw[3] * pf_gu_frac)};
        hipLaunchKernelGGL((small_pf_k<DEPTH>), dim3(st_[1].wgs + (pf_o_frac > 0 || pf_gu_frac > 0 ? pf_wgs : 0)), dim3(512), 0, st,
                           weights + offs[1], vpw[1], st_[1].wgs, xin, 256, xout, t0, t1, shift, l == 0 ? xcc_log : nullptr);
      } else {
        hipLaunchKernelGGL((stream_k<DEPTH, DEPTH, 1, 0>), dim3(st_[k].wgs), dim3(512), 0, st, weights + offs[k], vpw[k], xin, 256, xout, 0);
      }
      flip ^= 1;
    }
  }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, st);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * layers);
}

struct Stage { const char* name; double mb; int wgs; };

template <int DEPTH, int FIRST, int XDEP, int WORK>
static float run(hipStream_t st, const u32x4* weights, size_t total_vecs, unsigned* xa, unsigned* xb, const std::vector<Stage>& stages,
                 int layers, int delay, float* per_stage_us) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  size_t off = 0;
  int flip = 0;
  for (int l = 0; l < layers; l++)
    for (const Stage& s : stages) {
      long vecs_per_wg = (long)(s.mb * 1e6 / 16 / s.wgs);
      vecs_per_wg = vecs_per_wg / (8 * 64) * (8 * 64);
      if (vecs_per_wg == 0) vecs_per_wg = 8 * 64;
      if (off + (size_t)vecs_per_wg * s.wgs > total_vecs) off = 0;
      hipLaunchKernelGGL((stream_k<DEPTH, FIRST, XDEP, WORK>), dim3(s.wgs), dim3(512), 0, st, weights + off, vecs_per_wg,
                         (const u32x4*)(flip ? xb : xa), 256, flip ? xa : xb, delay);
      off += (size_t)vecs_per_wg * s.wgs;
      flip ^= 1;
    }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, st);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * layers);
}

int main(int argc, char** argv) {
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t bytes = (size_t)3600 << 20;   // as much distinct weight data as a 32-layer model streams
  u32x4* w; CK(hipMalloc(&w, bytes)); CK(hipMemset(w, 1, bytes));
  unsigned *xa, *xb; CK(hipMalloc(&xa, 1 << 16)); CK(hipMalloc(&xb, 1 << 16)); CK(hipMemset(xa, 0, 1 << 16)); CK(hipMemset(xb, 0, 1 << 16));
  const size_t total_vecs = bytes / 16;
  const int layers = 32;
  std::vector<Stage> layer = {{"qkv", 13.0, 192}, {"attn", 0.9, 32}, {"o", 8.65, 256}, {"gate_up", 60.6, 256}, {"down", 30.3, 256}};
  std::vector<Stage> only_gu = {{"gate_up", 60.6, 256}};
  std::vector<Stage> only_dn = {{"down", 30.3, 256}};
  std::vector<Stage> only_qkv = {{"qkv", 13.0, 192}};
  std::vector<Stage> only_o = {{"o", 8.65, 256}};
  std::vector<Stage> only_small = {{"attn", 0.9, 32}};
#define RUN(D, F, X, W, DL, tag)                                                                                      \
  printf("%-34s layer %6.2f us | gate_up %5.2f down %5.2f qkv %5.2f o %5.2f small %5.2f\n", tag,                       \
         run<D, F, X, W>(st, w, total_vecs, xa, xb, layer, layers, DL, nullptr),                                        \
         run<D, F, X, W>(st, w, total_vecs, xa, xb, only_gu, 64, DL, nullptr), run<D, F, X, W>(st, w, total_vecs, xa, xb, only_dn, 64, DL, nullptr), \
         run<D, F, X, W>(st, w, total_vecs, xa, xb, only_qkv, 64, DL, nullptr), run<D, F, X, W>(st, w, total_vecs, xa, xb, only_o, 64, DL, nullptr), \
         run<D, F, X, W>(st, w, total_vecs, xa, xb, only_small, 64, DL, nullptr))
#define RUNB(D, AUX, tag)                                                                                   \
  printf("%-34s gate_up %5.2f down %5.2f o %5.2f\n", tag, run_buf<D, AUX>(st, w, total_vecs, xa, xb, 60.6, 256, 64), \
         run_buf<D, AUX>(st, w, total_vecs, xa, xb, 30.3, 256, 64), run_buf<D, AUX>(st, w, total_vecs, xa, xb, 8.65, 256, 64))
  RUNB(4, 0, "buffer depth4 plain");
  RUNB(4, 2, "buffer depth4 nt");
  RUNB(4, 16, "buffer depth4 sc1");
  RUNB(4, 17, "buffer depth4 sc0 sc1");
  RUNB(4, 18, "buffer depth4 nt sc1");
  RUNB(4, 19, "buffer depth4 nt sc0 sc1");
  RUNB(7, 0, "buffer depth7 plain");
  RUNB(7, 2, "buffer depth7 nt");
  RUNB(7, 16, "buffer depth7 sc1");
  RUNB(7, 17, "buffer depth7 sc0 sc1");
  RUNB(7, 18, "buffer depth7 nt sc1");
  RUNB(7, 19, "buffer depth7 nt sc0 sc1");
  RUNB(14, 0, "buffer depth14 plain");
  RUNB(14, 2, "buffer depth14 nt");
  RUNB(14, 16, "buffer depth14 sc1");
  RUNB(14, 17, "buffer depth14 sc0 sc1");
  RUNB(14, 18, "buffer depth14 nt sc1");
  RUNB(2, 2, "buffer depth2 nt");
  RUNB(3, 2, "buffer depth3 nt");
  RUNB(5, 2, "buffer depth5 nt");
  RUN(14, 14, 1, 0, 0, "depth14 all xdep");
  RUN(14, 14, 1, 0, 2, "depth14 all xdep interleaved");
  RUN(8, 8, 1, 0, 0, "depth8 all xdep");
  RUN(8, 8, 1, 0, 2, "depth8 all xdep interleaved");
  RUN(7, 7, 1, 0, 0, "depth7 all xdep");
  RUN(7, 7, 1, 0, 2, "depth7 all xdep interleaved");
  RUN(6, 6, 1, 0, 0, "depth6 all xdep");
  RUN(6, 6, 1, 0, 2, "depth6 all xdep interleaved");
  RUN(4, 4, 1, 0, 0, "depth4 all xdep");
  RUN(4, 4, 1, 0, 2, "depth4 all xdep interleaved");
  RUN(3, 3, 1, 0, 2, "depth3 all xdep interleaved");
  RUN(4, 4, 1, 8, 0, "depth4 all xdep work8");
  RUN(4, 4, 1, 24, 0, "depth4 all xdep work24");
  RUN(7, 7, 1, 24, 0, "depth7 all xdep work24");
  unsigned* xlog; CK(hipMalloc(&xlog, 64)); CK(hipMemset(xlog, 0xff, 64));
  struct E { const char* tag; int small_wgs; double small_mb; int pf_wgs; double fo, fg; int shift; };
  const E exps[] = {
      {"small 8 WG, no prefetch", 8, 0.9, 0, 0, 0, 0},
      {"small 8 WG + 248 pf: o", 8, 0.9, 248, 1.0, 0, 0},
      {"small 8 WG + 248 pf: o, shifted", 8, 0.9, 248, 1.0, 0, 3},
      {"small 8 WG + 248 pf: o + 15% gu", 8, 0.9, 248, 1.0, 0.15, 0},
  };
  for (const E& e : exps) {
    float t = run_pf<4>(st, w, total_vecs, xa, xb, layers, e.small_wgs, e.small_mb, e.pf_wgs, e.fo, e.fg, e.shift, xlog);
    unsigned h[16]; CK(hipMemcpy(h, xlog, 64, hipMemcpyDeviceToHost));
    printf("%-44s layer %6.2f us | xcc of blocks 0..9:", e.tag, t);
    for (int i = 0; i < 10; i++) printf(" %u", h[i]);
    printf("\n");
  }
  return 0;
}
