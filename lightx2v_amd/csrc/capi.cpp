// Error plumbing and option switches shared by every entry point of liblx2v_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"

static thread_local char g_err[512] = "";

void lx2v_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* lx2v_last_error(void) { return g_err; }

extern "C" int lx2v_abi_version(void) { return 4; }

namespace {

Lx2vOptions g_opt;
std::once_flag g_opt_once;

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e == nullptr ? dflt : atoi(e);
}

void init_options() {
  const char* k = getenv("LX2V_ATTN_KERNEL");
  g_opt.attn_kernel = k == nullptr ? 16 : strcmp(k, "w8") == 0 ? 8 : strcmp(k, "w4") == 0 ? 4 : 16;
  const int qb = env_int("LX2V_ATTN_QB", 0);
  g_opt.attn_qb = (qb == 4 || qb == 5) ? qb : 0;
  g_opt.attn_tail = env_int("LX2V_ATTN_TAIL", 1) != 0;
  g_opt.attn_round_sync = env_int("LX2V_ATTN_ROUND_SYNC", 1) != 0;
  g_opt.attn_variant = env_int("LX2V_ATTN_VARIANT", 1) != 0;
  const char* f8 = getenv("LX2V_ATTN_FP8_KERNEL");
  g_opt.attn_fp8_kernel = (f8 != nullptr && strcmp(f8, "w8") == 0) ? 8 : 4;
  const char* f8p = getenv("LX2V_ATTN_FP8_P");
  g_opt.attn_fp8_p = (f8p != nullptr && strcmp(f8p, "linear") == 0) ? 1 : 0;
  const char* g = getenv("LX2V_GEMM_KERNEL");
  g_opt.gemm_kernel = g == nullptr ? 0 : strcmp(g, "simple") == 0 ? 1 : strcmp(g, "pingpong") == 0 ? 2 : 0;
}

struct OptName {
  const char* name;
  int Lx2vOptions::*field;
};
const OptName kOptNames[] = {
    {"attn_kernel", &Lx2vOptions::attn_kernel}, {"attn_qb", &Lx2vOptions::attn_qb},
    {"attn_tail", &Lx2vOptions::attn_tail},     {"attn_round_sync", &Lx2vOptions::attn_round_sync},
    {"attn_variant", &Lx2vOptions::attn_variant}, {"gemm_kernel", &Lx2vOptions::gemm_kernel},
    {"attn_fp8_kernel", &Lx2vOptions::attn_fp8_kernel}, {"attn_fp8_p", &Lx2vOptions::attn_fp8_p},
};

}  // namespace

const Lx2vOptions& lx2v_options() {
  std::call_once(g_opt_once, init_options);
  return g_opt;
}

extern "C" int lx2v_set_option(const char* name, int value) {
  LX2V_REQUIRE(name != nullptr, LX2V_EINVAL, "set_option: null name");
  lx2v_options();
  for (const OptName& o : kOptNames)
    if (strcmp(name, o.name) == 0) {
      if (o.field == &Lx2vOptions::attn_kernel) LX2V_REQUIRE(value == 16 || value == 4 || value == 8, LX2V_EINVAL, "set_option: attn_kernel must be 16, 4 or 8");
      if (o.field == &Lx2vOptions::attn_fp8_kernel) LX2V_REQUIRE(value == 4 || value == 8, LX2V_EINVAL, "set_option: attn_fp8_kernel must be 4 or 8");
      if (o.field == &Lx2vOptions::attn_fp8_p) LX2V_REQUIRE(value == 0 || value == 1, LX2V_EINVAL, "set_option: attn_fp8_p must be 0 (exact) or 1 (linear)");
      if (o.field == &Lx2vOptions::attn_qb) LX2V_REQUIRE(value == 0 || value == 4 || value == 5, LX2V_EINVAL, "set_option: attn_qb must be 0, 4 or 5");
      g_opt.*(o.field) = value;
      return LX2V_OK;
    }
  lx2v_set_error("set_option: unknown option '%s'", name);
  return LX2V_EINVAL;
}

extern "C" int lx2v_get_option(const char* name, int* value) {
  LX2V_REQUIRE(name != nullptr && value != nullptr, LX2V_EINVAL, "get_option: null argument");
  const Lx2vOptions& opt = lx2v_options();
  for (const OptName& o : kOptNames)
    if (strcmp(name, o.name) == 0) {
      *value = opt.*(o.field);
      return LX2V_OK;
    }
  lx2v_set_error("get_option: unknown option '%s'", name);
  return LX2V_EINVAL;
}
