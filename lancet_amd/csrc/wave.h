// wave.h -- execution-model shims for the window kernels.
//
// The kernels are written in a "phase" style: a workgroup owns one window; data-parallel phases are
// WG_FOR loops (work item i handled by lane i % WG_SIZE), single-lane sections are WG_LANE0, and phases are
// separated by WG_SYNC().  All control state that decides uniform branches lives in LDS (struct WinShared),
// written by lane 0 and read by everyone after a WG_SYNC.
//
// On gfx950 these map to the obvious HIP constructs.  When LANCET_WAVE_EMU is defined (ONLY by the test-side
// build tests/emu/Makefile; never by the product build) the same source compiles as plain C++ that runs the
// lanes of a phase one after the other, which lets the graph logic be debugged against the oracle on a
// GPU-less development box.  The emulation build is test infrastructure: the shipped library is the HIP one
// and has no CPU path.
#pragma once
#include <stdint.h>
#include "layout.h"

#ifndef LANCET_WAVE_EMU
#include <hip/hip_runtime.h>
#define DEV __device__ __forceinline__
#define DEVNI __device__ __noinline__
#ifndef LANCET_FAT
#define WG_FOR(i, n) for (int i = (int)threadIdx.x; i < (int)(n); i += (int)blockDim.x)
#define XG_FOR(i, n) WG_FOR(i, n)
#define XG_LANES 64
#else
// The fat form of the window kernel (window_fat.hip; the re-run tier, where a coverage pile-up is one window's build over
// a million k-mer occurrences): the same source on a workgroup of several waves.  Wave 0 runs every phase as before;
// the other waves ("helpers") start every WG_FOR past its bound, take part in every barrier, and share the loops marked
// XG_FOR -- the streaming passes of the general build, whose iterations are independent.
DEV int wg_thin_lane() { const int t = (int)threadIdx.x; return t < 64 ? t : 0x3FFFFF00; }
#define WG_FOR(i, n) for (int i = wg_thin_lane(); i < (int)(n); i += 64)
#define XG_FOR(i, n) for (int i = (int)threadIdx.x; i < (int)(n); i += (int)blockDim.x)
#define XG_LANES LC_FAT_LANES
#endif
// A data-parallel loop whose body EVERY lane runs: a lane past the end works on the last item again and `act` is false for it (the body
// guards its stores with it).  For loops with a heavy body: no EXEC-masked region around it, so whatever the register allocator moves
// inside the body is moved in every lane (see lc_sgpr below for what happened otherwise).
#define WG_FULL_BEGIN(i, act, n) for (int _b = 0; _b < (int)(n); _b += (int)blockDim.x) { const bool act = _b + (int)threadIdx.x < (int)(n); const int i = act ? _b + (int)threadIdx.x : (int)(n) - 1;
#define WG_FULL_END }
// A data-parallel loop that hands every lane LC_ILP independent items per trip (item u of a trip is i0 + u * LC_ILP_STRIDE, valid while
// < n): for loops that are a chain of dependent global round trips per item -- the loads of the items of one trip are issued together.
//   WG_FOR_ILP(i0, n) { idx[u] = i0 + u * LC_ILP_STRIDE ... }
#define LC_ILP 4
#ifndef LANCET_FAT
#define LC_ILP_STRIDE ((int)blockDim.x)
#define WG_FOR_ILP(i0, n) for (int i0 = (int)threadIdx.x; i0 < (int)(n); i0 += LC_ILP * (int)blockDim.x)
#else
#define LC_ILP_STRIDE 64
#define WG_FOR_ILP(i0, n) for (int i0 = wg_thin_lane(); i0 < (int)(n); i0 += LC_ILP * 64)
#endif
// Agent-scope fence before the barrier: the phases communicate through HBM with a mix of atomics (performed
// at L2) and plain loads (which may hit the CU's vector L1), so the L1 has to be invalidated at phase boundaries.
#define WG_SYNC() __syncthreads()
// Used at the phase boundaries of the table build, where atomics (executed at L2) are followed by plain loads of the
// same words (which may hit the vector L1): agent-scope fence = L1 invalidate.
#define WG_SYNC_FENCE() do { __threadfence(); __syncthreads(); } while (0)
// Barrier between phases that hand data over through LDS only: waits for the LDS operations, not for the global stores in flight (a
// __syncthreads() waits for their acknowledgement too -- a round trip to HBM per barrier in a latency-bound loop).
#define WG_SYNC_LDS() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
// lane-0 predicate, opaque to the optimiser (two consecutive lane-0 sections must not be merged or threaded)
DEV bool wg_is_lane0() { uint32_t t = threadIdx.x; asm volatile("" : "+v"(t)); return t == 0; }
DEV void wg_sync_fn() { __syncthreads(); }
// every lane-0 section is followed by a barrier executed by all lanes
#define WG_LANE0 for (int _wg_once = 1; _wg_once; _wg_once = 0, wg_sync_fn()) if (wg_is_lane0())
#define WG_SHARED __shared__
// atomics / L2 loads on pointers of any address space (generic, LC_GLOBAL, LC_LDS)
template <class P> DEV uint32_t dev_atomic_min(P p, uint32_t v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class P> DEV uint32_t dev_atomic_add(P p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class P> DEV uint32_t dev_atomic_max(P p, uint32_t v) { return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class P> DEV uint32_t dev_atomic_or(P p, uint32_t v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class P> DEV unsigned long long dev_atomic_cas64(P p, unsigned long long cmp, unsigned long long v) {
  __hip_atomic_compare_exchange_strong(p, &cmp, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return cmp;
}
template <class P> DEV uint32_t dev_atomic_cas32(P p, uint32_t cmp, uint32_t v) {
  __hip_atomic_compare_exchange_strong(p, &cmp, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return cmp;
}
// load that bypasses the CU's vector L1 (performed at L2): for words that other lanes updated with atomics
template <class P> DEV auto ld2(P p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// acquire load / release store at agent scope: hand-over of plain data between workgroups of different kernels (build service)
template <class P> DEV uint32_t ld_acq(P p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
template <class P> DEV void st_rel(P p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
template <class P> DEV uint32_t add_rel(P p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
DEV void dev_sleep() { __builtin_amdgcn_s_sleep(127); }
// A pointer / integer that is the same in every lane, moved to scalar registers.  Arguments of a non-inlined device function arrive in
// VGPRs whatever they hold; a value the compiler knows to be uniform needs no VGPR live range -- and cannot be hit by the live-range
// split ROCm 7.2's register allocator was seen to place in front of an EXEC restore (round 4: the copy of the EngineCaps pointer sat
// BEFORE `s_or_b64 exec` of the join block of a loop that waves without work skip, so those waves went on with a stale register and
// faulted on the next load through it; tools/check_exec_copies.py looks for that shape in the compiled kernels).
DEV uint32_t lc_sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
DEV int lc_sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> DEV T *lc_sgpr(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned long long r = (unsigned long long)lc_sgpr((uint32_t)v) | ((unsigned long long)lc_sgpr((uint32_t)(v >> 32)) << 32);
  return (T *)r;
}
typedef uint4 lc_u4;
typedef uint32_t lc_v4 __attribute__((ext_vector_type(4)));
// 16-byte load / store through a global pointer (HIP's uint4 has no copy from an address-space reference)
DEV lc_u4 ldg4(LC_GLOBAL const uint32_t *p) { const lc_v4 t = *(LC_GLOBAL const lc_v4 *)p; lc_u4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w; return r; }
DEV void stg4(LC_GLOBAL uint32_t *p, const lc_u4 v) { lc_v4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(LC_GLOBAL lc_v4 *)p = t; }
typedef uint32_t lc_v2 __attribute__((ext_vector_type(2)));
DEV void stg2(LC_GLOBAL uint32_t *p, uint32_t a, uint32_t b) { lc_v2 t; t.x = a; t.y = b; *(LC_GLOBAL lc_v2 *)p = t; }      // 8-byte store (8-byte aligned)
DEV int dev_popc(uint32_t x) { return __popc(x); }
DEV unsigned long long dev_brev64(unsigned long long x) { return __brevll(x); }
DEV int dev_popcll(unsigned long long x) { return __popcll(x); }
#else
#include <cstring>
#include <cmath>
#define DEV static inline
#define DEVNI static
#define WG_FOR(i, n) for (int i = 0; i < (int)(n); ++i)
#define LC_ILP 4
#define LC_ILP_STRIDE 1
#define WG_FOR_ILP(i0, n) for (int i0 = 0; i0 < (int)(n); i0 += LC_ILP)
#define WG_FULL_BEGIN(i, act, n) for (int i = 0; i < (int)(n); ++i) { const bool act = true;
#define WG_FULL_END }
#define XG_FOR(i, n) WG_FOR(i, n)
#define XG_LANES 64
static thread_local unsigned long lc_emu_syncs = 0;          /* barriers a workgroup would execute (tuning aid: tests/emu LANCET_EMU_SYNCS) */
#define WG_SYNC() ((void)++lc_emu_syncs)
#define WG_SYNC_FENCE() ((void)++lc_emu_syncs)
#define WG_SYNC_LDS() ((void)++lc_emu_syncs)
#define WG_LANE0 if (true)
#define WG_SHARED static thread_local
DEV uint32_t dev_atomic_min(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
DEV uint32_t dev_atomic_add(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
DEV uint32_t dev_atomic_max(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
DEV uint32_t dev_atomic_or(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
DEV unsigned long long dev_atomic_cas64(unsigned long long *p, unsigned long long cmp, unsigned long long v) {
  unsigned long long o = *p; if (o == cmp) *p = v; return o;
}
DEV uint32_t dev_atomic_cas32(uint32_t *p, uint32_t cmp, uint32_t v) { uint32_t o = *p; if (o == cmp) *p = v; return o; }
DEV uint32_t ld2(const uint32_t *p) { return *p; }
DEV unsigned long long ld2(const unsigned long long *p) { return *p; }
DEV uint32_t ld_acq(const uint32_t *p) { return *p; }
DEV void st_rel(uint32_t *p, uint32_t v) { *p = v; }
DEV uint32_t add_rel(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
DEV void dev_sleep() {}
template <class T> DEV T lc_sgpr(T v) { return v; }
struct alignas(16) lc_u4 { uint32_t x, y, z, w; };
typedef lc_u4 lc_v4;
DEV lc_u4 ldg4(const uint32_t *p) { lc_u4 v; memcpy(&v, p, 16); return v; }
DEV void stg4(uint32_t *p, const lc_u4 v) { memcpy(p, &v, 16); }
DEV void stg2(uint32_t *p, uint32_t a, uint32_t b) { memcpy(p, &a, 4); memcpy(p + 1, &b, 4); }
DEV int dev_popc(uint32_t x) { return __builtin_popcount(x); }
DEV unsigned long long dev_brev64(unsigned long long x) {
  x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  return __builtin_bswap64(x);
}
DEV int dev_popcll(unsigned long long x) { return __builtin_popcountll(x); }
#endif
