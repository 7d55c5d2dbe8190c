// Error plumbing and version of libmi355q (host only).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace mi355q {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// hipGetLastError after a launch reports bad launch configurations without synchronising.
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(MI355Q_EHIP, "%s: %s", what, hipGetErrorString(e));
  return MI355Q_OK;
}

// Environment knobs are read once per process: launch paths ask this table, never getenv().
namespace {
struct Knob { const char* name; bool read; const char* value; };
std::mutex g_env_mu;
constexpr int MAX_KNOBS = 256;
Knob g_knobs[MAX_KNOBS];
int g_nknobs = 0;
}  // namespace

const char* env_value(const char* name) {
  std::lock_guard<std::mutex> lock(g_env_mu);
  for (int i = 0; i < g_nknobs; i++)
    if (!strcmp(g_knobs[i].name, name)) return g_knobs[i].value;
  const char* v = getenv(name);
  if (g_nknobs < MAX_KNOBS) g_knobs[g_nknobs++] = Knob{name, true, v ? strdup(v) : nullptr};
  return v;
}
bool env_override(const char* name, const char* value) {
  std::lock_guard<std::mutex> lock(g_env_mu);
  for (int i = 0; i < g_nknobs; i++)
    if (!strcmp(g_knobs[i].name, name)) {
      g_knobs[i].value = value ? strdup(value) : nullptr;   // old strings are left alone: readers may hold them
      return true;
    }
  if (g_nknobs == MAX_KNOBS) return false;
  g_knobs[g_nknobs++] = Knob{strdup(name), true, value ? strdup(value) : nullptr};
  return true;
}
bool env_flag(const char* name) { return env_value(name) != nullptr; }
int env_int(const char* name, int dflt) {
  const char* v = env_value(name);
  return v ? atoi(v) : dflt;
}

// Streams are given banks of the library-owned arrival counters in order of first use, so that
// launches on different streams of one process never share a counter.  -1: no bank left (the
// caller takes its stateless route).
namespace {
std::mutex g_bank_mu;
void* g_bank_stream[MI355Q_STATE_BANKS];
int g_banks_used = 0;
}  // namespace

int stream_bank(void* stream) {
  std::lock_guard<std::mutex> lock(g_bank_mu);
  for (int i = 0; i < g_banks_used; i++)
    if (g_bank_stream[i] == stream) return i;
  if (g_banks_used == MI355Q_STATE_BANKS) return -1;
  g_bank_stream[g_banks_used] = stream;
  return g_banks_used++;
}

int reset_attention_state(hipStream_t stream);   // attention.hip
int reset_argmax_state(hipStream_t stream);      // elementwise.hip

}  // namespace mi355q

extern "C" int mi355q_set_option(const char* name, const char* value) {
  if (!name || strncmp(name, "MI355Q_", 7) != 0)
    return mi355q::set_error(MI355Q_EINVAL, "set_option: option names start with MI355Q_");
  if (!mi355q::env_override(name, value))
    return mi355q::set_error(MI355Q_ENOSPACE, "set_option: the option table is full (%s not set)", name);
  return MI355Q_OK;
}

extern "C" int mi355q_reset_library_state(mi355q_stream_t stream) {
  if (int e = mi355q::reset_attention_state((hipStream_t)stream)) return e;
  return mi355q::reset_argmax_state((hipStream_t)stream);
}

extern "C" const char* mi355q_last_error(void) { return mi355q::g_err; }
extern "C" int mi355q_version(void) { return 2; }
