// One-shot all-reduce of small tensors over peer memory (xGMI), for the two [tokens, hidden] sums per
// decoder layer of tensor-parallel decode: 16 KiB at batch 1, where a library collective costs several
// times the 1-2 us the wires need.
//
// Every rank owns a region of uncached device memory that all peers map through IPC handles.  A call
// PUSHES this rank's input into every peer's region (and its own) as 8-byte granules {32 bits of data,
// 32-bit tag}: the tag is the call's epoch, so a granule is valid the moment its tag matches -- one store
// carries the data and its arrival, and no fence, flag or second round trip is needed (8-byte stores are
// not torn).  The rank then sweeps its OWN region (local polls: nothing spins over the links) until the
// granules of all ranks carry the epoch, adds them in rank order in fp32 (every rank computes the same
// bits) and writes the result.  The epoch lives in device memory and is advanced by the kernel, so a
// captured call replays correctly.  Two sets of slots alternate by epoch parity: a rank can be at most
// one call ahead of the slowest (it cannot finish call k + 1 before every peer has pushed its part of
// k + 1, which a peer does only after finishing call k), so the set of call k is not overwritten before
// every rank has read it.  Every wait is bounded: on a timeout the kernel counts an error and leaves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace mi355q {

constexpr int PEER_AR_MAX_WORLD = 8;
constexpr int PEER_AR_THREADS = 256;
constexpr int PEER_AR_MAX_BLOCKS = 64;

struct PeerArParams {
  uint64_t* regions[PEER_AR_MAX_WORLD];   // every rank's region as mapped here; [rank] is this rank's own
  const uint32_t* in;                     // n / 2 granules of input
  uint32_t* out;
  uint32_t* state;                        // [0] epoch, [1] workgroups done, [2] timeouts
  int rank, world;
  int granules;                           // of this call (per rank)
  int row_granules;                       // all-gather: granules per row of this rank's shard
  int max_granules;                       // slots per (parity, rank)
  long long timeout_ticks;                // of the 100 MHz wall clock
};

template <typename DT>
__device__ inline void add2(float& a, float& b, uint32_t v) {
  a += DT::to_f32((uint16_t)(v & 0xffff));
  b += DT::to_f32((uint16_t)(v >> 16));
}

// GATHER: instead of the sum, every rank's granules are laid side by side: row m of the result is the ranks' rows m
// in rank order (the concatenation along the last dimension of [rows, row_granules] shards).
template <typename DT, bool GATHER>
__global__ __launch_bounds__(PEER_AR_THREADS) void peer_all_reduce_kernel(const PeerArParams p) {
  const uint32_t epoch = __hip_atomic_load(&p.state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const size_t set = (size_t)(epoch & 1u) * p.world * p.max_granules;
  const int stride = gridDim.x * PEER_AR_THREADS;
  // ---- push: this rank's granules into slot [set][rank] of every region
  for (int i = blockIdx.x * PEER_AR_THREADS + threadIdx.x; i < p.granules; i += stride) {
    const uint64_t g = (uint64_t)p.in[i] | ((uint64_t)epoch << 32);
    const size_t at = set + (size_t)p.rank * p.max_granules + i;
#pragma unroll
    for (int r = 0; r < PEER_AR_MAX_WORLD; r++)
      if (r < p.world) __hip_atomic_store(p.regions[r] + at, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- sweep this rank's own region, add in rank order
  const uint64_t* mine = p.regions[p.rank] + set;
  const long long t0 = wall_clock64();
  bool timed_out = false;
  for (int i = blockIdx.x * PEER_AR_THREADS + threadIdx.x; i < p.granules; i += stride) {
    uint64_t g[PEER_AR_MAX_WORLD];
    unsigned pending = (1u << p.world) - 1u;
    while (pending) {
#pragma unroll
      for (int r = 0; r < PEER_AR_MAX_WORLD; r++)
        if (pending & (1u << r))
          g[r] = __hip_atomic_load(mine + (size_t)r * p.max_granules + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
      for (int r = 0; r < PEER_AR_MAX_WORLD; r++)
        if ((pending & (1u << r)) && (uint32_t)(g[r] >> 32) == epoch) pending &= ~(1u << r);
      if (pending && wall_clock64() - t0 > p.timeout_ticks) { timed_out = true; break; }
    }
    if constexpr (GATHER) {
      const int m = i / p.row_granules, j = i - m * p.row_granules;
      uint32_t* o = p.out + (size_t)m * p.world * p.row_granules + j;
#pragma unroll
      for (int r = 0; r < PEER_AR_MAX_WORLD; r++)
        if (r < p.world) o[(size_t)r * p.row_granules] = (uint32_t)g[r];
    } else {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int r = 0; r < PEER_AR_MAX_WORLD; r++)
        if (r < p.world) add2<DT>(a, b, (uint32_t)g[r]);
      p.out[i] = pack2<DT>(a, b);
    }
    if (timed_out) break;
  }
  if (timed_out) atomicAdd(&p.state[2], 1u);
  // ---- the last workgroup to finish advances the epoch (0 is the tag of never-written slots: skipped)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&p.state[1], 1u) == gridDim.x - 1) {
      p.state[1] = 0;
      uint32_t next = epoch + 1u;
      if (next == 0u) next = 2u;   // keeps the parity sequence: ... 0xffffffff (odd), 2 (even)
      __hip_atomic_store(&p.state[0], next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct PeerArComm {
  int rank = 0, world = 1, device = 0;
  size_t max_bytes = 0;
  int max_granules = 0;
  size_t region_bytes = 0;
  uint64_t* region = nullptr;                 // own, uncached
  uint64_t* regions[PEER_AR_MAX_WORLD] = {};  // as mapped in this process
  bool opened[PEER_AR_MAX_WORLD] = {};
  uint32_t* state = nullptr;
  bool ready = false;
  long long timeout_ticks = 500000000LL;      // five seconds: ranks may reach their first exchange far apart
};

}  // namespace mi355q

using namespace mi355q;

extern "C" int mi355q_peer_ar_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int mi355q_peer_ar_init(int rank, int world_size, size_t max_bytes, void** comm_out) {
  MI355Q_REQUIRE(comm_out, "peer_ar_init: null comm_out");
  *comm_out = nullptr;
  // (one rank: the exchange with itself, for tests of the host side on a single device)
  MI355Q_REQUIRE(world_size >= 1 && world_size <= PEER_AR_MAX_WORLD && rank >= 0 && rank < world_size,
                 "peer_ar_init: rank %d of %d (up to %d ranks)", rank, world_size, PEER_AR_MAX_WORLD);
  MI355Q_REQUIRE(max_bytes >= 16 && max_bytes % 16 == 0 && max_bytes <= (size_t)8 << 20,
                 "peer_ar_init: max_bytes %zu (a multiple of 16 up to 8 MiB)", max_bytes);
  PeerArComm* c = new PeerArComm();
  c->rank = rank;
  c->world = world_size;
  c->max_bytes = max_bytes;
  c->max_granules = (int)(max_bytes / 4);
  c->region_bytes = (size_t)2 * world_size * c->max_granules * sizeof(uint64_t);
  hipError_t e = hipGetDevice(&c->device);
  if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&c->region, c->region_bytes, hipDeviceMallocUncached);
  if (e == hipSuccess) e = hipMemset(c->region, 0, c->region_bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&c->state, 4 * sizeof(uint32_t));
  if (e == hipSuccess) {
    const uint32_t init[4] = {2u, 0u, 0u, 0u};   // epoch starts at 2: tag 0 marks slots never written
    e = hipMemcpy(c->state, init, sizeof(init), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (c->region) (void)hipFree(c->region);
    if (c->state) (void)hipFree(c->state);
    delete c;
    return set_error(MI355Q_EHIP, "peer_ar_init: %s", hipGetErrorString(e));
  }
  c->regions[rank] = c->region;
  if (const char* t = getenv("MI355Q_PEER_AR_TIMEOUT_MS")) {
    const long long ms = atoll(t);
    if (ms > 0) c->timeout_ticks = ms * 100000LL;
  }
  *comm_out = c;
  return MI355Q_OK;
}

extern "C" int mi355q_peer_ar_get_handle(void* comm, void* handle_out) {
  PeerArComm* c = (PeerArComm*)comm;
  MI355Q_REQUIRE(c && handle_out, "peer_ar_get_handle: null argument");
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, c->region);
  if (e != hipSuccess) return set_error(MI355Q_EHIP, "peer_ar_get_handle: %s", hipGetErrorString(e));
  memcpy(handle_out, &h, sizeof(h));
  return MI355Q_OK;
}

extern "C" int mi355q_peer_ar_open_handles(void* comm, const void* handles) {
  PeerArComm* c = (PeerArComm*)comm;
  MI355Q_REQUIRE(c && handles, "peer_ar_open_handles: null argument");
  MI355Q_REQUIRE(!c->ready, "peer_ar_open_handles: already opened");
  for (int r = 0; r < c->world; r++) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)r * sizeof(h), sizeof(h));
    const hipError_t e = hipIpcOpenMemHandle((void**)&c->regions[r], h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return set_error(MI355Q_EHIP, "peer_ar_open_handles: rank %d: %s", r, hipGetErrorString(e));
    c->opened[r] = true;
  }
  c->ready = true;
  return MI355Q_OK;
}

extern "C" size_t mi355q_peer_ar_max_bytes(void* comm) { return comm ? ((PeerArComm*)comm)->max_bytes : 0; }

static int peer_launch(PeerArComm* c, const void* input, void* output, size_t num_elements, size_t row_elements,
                       int dtype, bool gather, hipStream_t stream) {
  PeerArParams p;
  for (int r = 0; r < PEER_AR_MAX_WORLD; r++) p.regions[r] = r < c->world ? c->regions[r] : nullptr;
  p.in = (const uint32_t*)input;
  p.out = (uint32_t*)output;
  p.state = c->state;
  p.rank = c->rank;
  p.world = c->world;
  p.granules = (int)(num_elements / 2);
  p.row_granules = (int)(row_elements / 2);
  p.max_granules = c->max_granules;
  p.timeout_ticks = c->timeout_ticks;
  const int blocks = std::min(PEER_AR_MAX_BLOCKS, (p.granules + PEER_AR_THREADS - 1) / PEER_AR_THREADS);
  if (gather)
    hipLaunchKernelGGL((peer_all_reduce_kernel<BF16, true>), dim3(blocks), dim3(PEER_AR_THREADS), 0, stream, p);
  else if (dtype == MI355Q_BF16)
    hipLaunchKernelGGL((peer_all_reduce_kernel<BF16, false>), dim3(blocks), dim3(PEER_AR_THREADS), 0, stream, p);
  else
    hipLaunchKernelGGL((peer_all_reduce_kernel<FP16, false>), dim3(blocks), dim3(PEER_AR_THREADS), 0, stream, p);
  return check_launch("peer_all_reduce_kernel");
}

extern "C" int mi355q_peer_ar_all_reduce(void* comm, const void* input, void* output, size_t num_elements, int dtype,
                                         mi355q_stream_t stream) {
  PeerArComm* c = (PeerArComm*)comm;
  MI355Q_REQUIRE(c && c->ready, "peer_ar_all_reduce: communicator not ready (init, exchange handles, open them)");
  MI355Q_REQUIRE(dtype == MI355Q_BF16 || dtype == MI355Q_FP16, "peer_ar_all_reduce: dtype must be bf16 or fp16");
  if (num_elements == 0) return MI355Q_OK;
  MI355Q_REQUIRE(input && output, "peer_ar_all_reduce: null pointer");
  MI355Q_REQUIRE(num_elements % 2 == 0 && num_elements * 2 <= c->max_bytes,
                 "peer_ar_all_reduce: %zu elements (an even count, at most %zu bytes)", num_elements, c->max_bytes);
  MI355Q_REQUIRE((((uintptr_t)input | (uintptr_t)output) & 3) == 0, "peer_ar_all_reduce: pointers must be 4-byte aligned");
  return peer_launch(c, input, output, num_elements, num_elements, dtype, false, (hipStream_t)stream);
}

extern "C" int mi355q_peer_ar_all_gather(void* comm, const void* input, void* output, int rows, size_t row_elements,
                                         int dtype, mi355q_stream_t stream) {
  PeerArComm* c = (PeerArComm*)comm;
  MI355Q_REQUIRE(c && c->ready, "peer_ar_all_gather: communicator not ready (init, exchange handles, open them)");
  MI355Q_REQUIRE(dtype == MI355Q_BF16 || dtype == MI355Q_FP16, "peer_ar_all_gather: dtype must be bf16 or fp16");
  if (rows == 0 || row_elements == 0) return MI355Q_OK;
  MI355Q_REQUIRE(input && output && input != output && rows > 0, "peer_ar_all_gather: bad arguments (no gather in place)");
  MI355Q_REQUIRE(row_elements % 2 == 0 && (size_t)rows * row_elements * 2 <= c->max_bytes,
                 "peer_ar_all_gather: %d rows of %zu elements (an even count, at most %zu bytes per rank)", rows,
                 row_elements, c->max_bytes);
  MI355Q_REQUIRE((((uintptr_t)input | (uintptr_t)output) & 3) == 0, "peer_ar_all_gather: pointers must be 4-byte aligned");
  return peer_launch(c, input, output, (size_t)rows * row_elements, row_elements, dtype, true, (hipStream_t)stream);
}

extern "C" int mi355q_peer_ar_timeouts(void* comm, int* timeouts_out) {
  PeerArComm* c = (PeerArComm*)comm;
  MI355Q_REQUIRE(c && timeouts_out, "peer_ar_timeouts: null argument");
  uint32_t st[4];
  const hipError_t e = hipMemcpy(st, c->state, sizeof(st), hipMemcpyDeviceToHost);   // waits for the work issued so far
  if (e != hipSuccess) return set_error(MI355Q_EHIP, "peer_ar_timeouts: %s", hipGetErrorString(e));
  *timeouts_out = (int)st[2];
  return MI355Q_OK;
}

extern "C" int mi355q_peer_ar_destroy(void* comm) {
  PeerArComm* c = (PeerArComm*)comm;
  if (!c) return MI355Q_OK;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; r++)
    if (c->opened[r]) (void)hipIpcCloseMemHandle(c->regions[r]);
  if (c->region) (void)hipFree(c->region);
  if (c->state) (void)hipFree(c->state);
  delete c;
  return MI355Q_OK;
}
