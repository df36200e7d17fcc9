// Pieces shared by the one-wave-per-SIMD attention kernels (attn_fwd_w4.hip, attn_fwd_x16.hip):
// the round alignment of workgroups on an XCD, the softmax VALU instructions as asm statements,
// and the launch helper that owns the per-device round counters.
#pragma once

#include <stdlib.h>

#include <mutex>

#include "attn_common.h"

namespace {

// Round alignment.  Workgroup ids go to the 8 XCDs round-robin and, with one workgroup per CU,
// run there in rounds of `cus_per_xcd`; the workgroups of one round share a head (or two) and
// sweep its K/V together, so each tile is fetched into the XCD's L2 once per round -- as long as
// they stay within the L2's reach of each other.  Left alone they drift apart over a few dozen
// rounds (deep prefetch hides the misses that would hold a leader back) and every CU ends up
// streaming K/V from memory on its own.  So a workgroup of round r waits, with a time limit, until
// the XCD's previous round has finished.  Only a locality hint: any counter value is safe.
//
// The counters only ever grow.  A launch gets one of kRoundSlots counter sets and the values they
// will hold when it starts (`base`, tracked on the host: a launch of n workgroups adds exactly
// n/8 (+1) to each XCD's counter), so nothing is reset between launches and launches that overlap
// on different streams do not share counters.
struct RoundSync {
  unsigned* done;    // [8] finished workgroups per XCD of this launch's slot; null = off
  unsigned base[8];  // their values before this launch's first workgroup
  int cus_per_xcd;
};
constexpr long long kRoundWaitTicks = 10000;  // 100 us of the 100 MHz wall clock
constexpr int kRoundSlots = 64;

// Entry of a workgroup: wait (bounded) for the XCD's previous round.  Ends in a workgroup barrier
// when alignment is on.
LX2V_DEVICE void round_enter(const RoundSync& rs) {
  if (rs.done != nullptr) {
    const unsigned round = (blockIdx.x >> 3) / rs.cus_per_xcd;
    if (round > 0 && threadIdx.x == 0) {
      const unsigned want = rs.base[blockIdx.x & 7] + round * rs.cus_per_xcd;
      const long long t_end = wall_clock64() + kRoundWaitTicks;
      while ((int)(__hip_atomic_load(rs.done + (blockIdx.x & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0 && wall_clock64() < t_end)
        __builtin_amdgcn_s_sleep(16);
    }
    __syncthreads();
  }
}
LX2V_DEVICE void round_leave(const RoundSync& rs) {
  if (rs.done != nullptr && threadIdx.x == 0) __hip_atomic_fetch_add(rs.done + (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The softmax VALU instructions are asm statements: their ORDER inside each MFMA gap is part of
// the design (a lone wave issues in program order), and the compiler would otherwise hoist exps
// above the MFMA they are meant to follow and gather the row-sum adds into one tree at the end of
// the period.  None of them reads a register written by the statement just before it.
LX2V_DEVICE float exp2_op(float x) {
  float r;
  asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
LX2V_DEVICE void add_op(float& acc, float x) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x)); }
LX2V_DEVICE uint32_t cvt_pk_op(float lo, float hi) {
  uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__device__ __attribute__((aligned(256))) unsigned g_round_done[kRoundSlots][8];  // one copy per translation unit and device

// Host bookkeeping of the counter sets of one device (per translation unit, like the array).
struct RoundHost {
  unsigned* dev_ptr = nullptr;  // address of this device's copy of g_round_done
  int cus_per_xcd = 0;          // 0 = alignment unavailable on this device (not 8 XCDs' worth of CUs)
  unsigned next = 0;            // next slot
  unsigned base[kRoundSlots][8] = {};
};

// Launch `kshort` (skv <= 2048) or `klong` over the first `nwg` of the nqb * heads workgroups
// (head-major order) with `threads` threads each, with round alignment when a head's K/V
// (`kv_bytes_per_head` as staged) outgrows an XCD's L2 (4 MiB) and there are many rounds.
template <class KS, class KL, class P>
int launch_with_rounds(KS kshort, KL klong, int threads, const P& p, int64_t kv_bytes_per_head, int64_t nwg, hipStream_t st, const char* what);

template <class KS, class KL>
int launch_with_rounds(KS kshort, KL klong, int threads, const lx2v_attn::AttnParams& p, int64_t nwg, hipStream_t st, const char* what) {
  return launch_with_rounds(kshort, klong, threads, p, (int64_t)p.skv * lx2v_attn::HD * 4, nwg, st, what);
}

template <class KS, class KL, class P>
int launch_with_rounds(KS kshort, KL klong, int threads, const P& p, int64_t kv_bytes_per_head, int64_t nwg, hipStream_t st, const char* what) {
  RoundSync rs{};
  rs.done = nullptr;
  rs.cus_per_xcd = 32;
  static std::mutex mu;
  static RoundHost host[64];
  RoundHost* h = nullptr;
  unsigned slot = 0;
  // A launch recorded into a graph would replay with the counter values of its capture baked in while the
  // device counters move on with every replay: alignment is left off under capture (it is a locality hint only).
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (st != nullptr && hipStreamIsCapturing(st, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
  if (lx2v_options().attn_round_sync != 0 && kv_bytes_per_head > (8 << 20) && capturing == hipStreamCaptureStatusNone) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { lx2v_set_error("%s: hipGetDevice failed", what); return LX2V_EHIP; }
    std::lock_guard<std::mutex> lock(mu);
    h = &host[dev];
    if (h->dev_ptr == nullptr) {
      int cus = 0;
      if (hipGetSymbolAddress((void**)&h->dev_ptr, HIP_SYMBOL(g_round_done)) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        h->dev_ptr = nullptr;
        lx2v_set_error("%s: round counters unavailable", what);
        return LX2V_EHIP;
      }
      h->cus_per_xcd = (cus >= 64 && cus % 8 == 0) ? cus / 8 : 0;
    }
    if (h->cus_per_xcd > 0 && nwg > 16 * (int64_t)h->cus_per_xcd) {
      slot = h->next++ % kRoundSlots;
      rs.done = h->dev_ptr + slot * 8;
      rs.cus_per_xcd = h->cus_per_xcd;
      for (int x = 0; x < 8; ++x) {
        rs.base[x] = h->base[slot][x];
        h->base[slot][x] += (unsigned)(nwg / 8 + (x < nwg % 8 ? 1 : 0));
      }
    } else {
      h = nullptr;
    }
  }
  if (p.skv <= 2048) hipLaunchKernelGGL(kshort, dim3((unsigned)nwg), dim3(threads), 0, st, p, rs);
  else hipLaunchKernelGGL(klong, dim3((unsigned)nwg), dim3(threads), 0, st, p, rs);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    if (h != nullptr) {  // nothing ran: the counters keep their old values
      std::lock_guard<std::mutex> lock(mu);
      for (int x = 0; x < 8; ++x) h->base[slot][x] = rs.base[x];
    }
    lx2v_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return LX2V_EHIP;
  }
  return LX2V_OK;
}

}  // namespace
