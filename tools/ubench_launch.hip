// Microbenchmark: cost of dependent kernels in a HIP graph on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_chain1(const int* a, int* b) { if (threadIdx.x == 0) b[blockIdx.x] = a[blockIdx.x] + 1; }
__global__ void k_chain2(const int* idx, const int* a, int* b) { if (threadIdx.x == 0) b[blockIdx.x] = a[idx[blockIdx.x]] + 1; }
__global__ void k_atomic_tail(int* cnt, const float* part, float* out, int n) {
  __shared__ int last;
  if (threadIdx.x == 0) {
    int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = t == (int)gridDim.x - 1;
    if (last) { __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
  }
  __syncthreads();
  if (!last) return;
  float s = 0; for (int i = 0; i < n; i++) s += part[i * 256 + threadIdx.x];
  out[threadIdx.x] = s;
}

template <typename F> float time_graph(hipStream_t st, int nk, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nk; i++) launch(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  for (int i = 0; i < 10; i++) hipGraphLaunch(ge, st);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / (10 * nk);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  int *a, *b, *idx, *cnt; float *part, *out;
  CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMalloc(&idx, 1 << 20)); CK(hipMalloc(&cnt, 256));
  CK(hipMalloc(&part, 1 << 20)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20)); CK(hipMemset(idx, 0, 1 << 20)); CK(hipMemset(cnt, 0, 256));
  const int NK = 200;
  for (int grid : {1, 256, 1024}) {
    for (int wg : {64, 256, 1024}) {
      float t0 = time_graph(st, NK, [&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(wg), 0, st); });
      float t1 = time_graph(st, NK, [&](int i) { hipLaunchKernelGGL(k_chain1, dim3(grid), dim3(wg), 0, st, (i & 1) ? b : a, (i & 1) ? a : b); });
      float t2 = time_graph(st, NK, [&](int i) { hipLaunchKernelGGL(k_chain2, dim3(grid), dim3(wg), 0, st, idx, (i & 1) ? b : a, (i & 1) ? a : b); });
      printf("grid %4d wg %4d : empty %.2f us/kernel, load->store %.2f, load->load->store %.2f\n", grid, wg, t0, t1, t2);
    }
  }
  float t3 = time_graph(st, NK, [&](int) { hipLaunchKernelGGL(k_atomic_tail, dim3(512), dim3(256), 0, st, cnt, part, out, 8); });
  printf("ticket + acquire + 8 dependent loads tail (512 blocks): %.2f us/kernel\n", t3);
  // eager
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st);
  hipEventRecord(e1, st); hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("eager empty: %.2f us/kernel\n", ms * 1000 / 2000);
  return 0;
}
