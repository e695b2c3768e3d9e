// engine.hip -- C-ABI of the MI355X micro-assembly engine (include/lancet_engine.h) on top of the kernels.
//
// Host responsibilities only: device memory, the upload/prep step (reads are trimmed and packed to
// 2 bit/base + 1 quality bit/base on the GPU), the persistent-kernel launch, result read-back.
// There is no CPU path: without a HIP device lancet_engine_create fails with LANCET_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "build_lds.h"
#include "host_common.h"

// window_fat.hip
extern "C" int lancet_engine_submit(lancet_engine *e);
int lc_launch_window_fat(int slots, hipStream_t stream, const lancet_params *P, const DevBatch *B, const EngineCaps *C, Work *works, DevOut *OUT);

#define HIPCHK(e, call)                                                                           \
  do { hipError_t _r = (call); if (_r != hipSuccess) { (e)->err = std::string(#call) + ": " + hipGetErrorString(_r); return LANCET_E_HIP; } } while (0)

// launch_bounds is deliberately 2x the launched size: with a provably single-wave workgroup the compiler turns
// s_barrier into a no-op and then threads the lane-0 sections of consecutive phases together, which lets lane 0
// run ahead of the other lanes (observed on gfx950: lanes 1..63 skipped whole phases).
#ifndef LC_W_EU
#define LC_W_EU 4               /* waves per SIMD the window kernel is compiled for: 128 VGPRs, 16 slots per CU (5: 96 VGPRs and 2.4x the spills for the same kernel time; tools/variant.sh) */
#endif
__global__ void __launch_bounds__(LANCET_WG * 2) __attribute__((amdgpu_waves_per_eu(LC_W_EU, LC_W_EU))) window_kernel(const lancet_params *P, const DevBatch *B, const EngineCaps *C, Work *works, DevOut *OUT) {
  window_kernel_body((LC_GLOBAL const lancet_params *)P, (LC_GLOBAL const DevBatch *)B, (LC_GLOBAL const EngineCaps *)C, (LC_GLOBAL Work *)works, (LC_GLOBAL DevOut *)OUT, (LC_WS *)&lc_shared, (int)blockIdx.x);
}

// The first graph of every window in LDS (build_lds.h): 256 lanes per window, 2 workgroups per CU (80 KB of LDS each)
__global__ void __launch_bounds__(bl_small::WG) __attribute__((amdgpu_waves_per_eu(BL_SMALL_EU, BL_SMALL_EU))) build_kernel(const lancet_params *P, const DevBatch *B, const EngineCaps *C, uint8_t *pre, uint8_t *scratch, uint32_t *queue,
                                                      unsigned long long *phase, uint8_t *pool, uint32_t pool_cap, int depth, uint32_t *biglist, int wait_all, int cycle, uint32_t *slotmap, int nslots, int first_gen) {
  int slot = (int)blockIdx.x;
  bl_small::BL_S &S = *(bl_small::BL_S *)&bl_small::bl_shared;
  const int slots_on = cycle;                                      // (scratch slots by the bitmap for every workgroup of such a launch)
  if ((int)blockIdx.x >= first_gen) cycle = 0;                     // (the workgroups behind the first generation stay)
  if (slots_on > 0) {            // (LANCET_BUILD_CYCLE: a workgroup leaves after `cycle` windows and the grid is that much larger; scratch slots by a bitmap)
    if (threadIdx.x == 0) {
      int got = -1;
      while (got < 0) {
        for (int i = 0; i < (nslots + 31) / 32 && got < 0; ++i) {
          uint32_t m = __hip_atomic_load(&slotmap[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while (~m) {
            const int b = __builtin_ctz(~m);
            if (i * 32 + b >= nslots) break;
            const uint32_t old = atomicOr(&slotmap[i], 1u << b);
            if (!(old & (1u << b))) { got = i * 32 + b; break; }
            m = old | (1u << b);
          }
        }
        if (got < 0) __builtin_amdgcn_s_sleep(127);
      }
      S.w = got;
    }
    __syncthreads(); slot = S.w; __syncthreads();
  }
  bl_small::build_kernel_body((LC_GLOBAL const lancet_params *)P, (LC_GLOBAL const DevBatch *)B, (LC_GLOBAL const EngineCaps *)C, (LC_GLOBAL uint8_t *)pre, (LC_GLOBAL uint8_t *)scratch,
                    (LC_GLOBAL uint32_t *)queue, S, slot, (LC_GLOBAL unsigned long long *)phase, (LC_GLOBAL uint8_t *)pool, pool_cap, depth,
                    (LC_GLOBAL uint32_t *)biglist, false, wait_all != 0, cycle);
  if (slots_on > 0) { __syncthreads(); if (threadIdx.x == 0) atomicAnd(&slotmap[slot >> 5], ~(1u << (slot & 31))); }
}
// The same for the windows the 512-lane configuration turned away for their size (60x/60x: ~360 reads, 58 k bases): 1024 lanes,
// one workgroup per CU (~100 KB of LDS), off the list the first kernel left.
__global__ void __launch_bounds__(bl_large::WG) __attribute__((amdgpu_waves_per_eu(4, 4))) build_kernel_large(const lancet_params *P, const DevBatch *B, const EngineCaps *C, uint8_t *pre, uint8_t *scratch, uint32_t *queue,
                                                      unsigned long long *phase, uint8_t *pool, uint32_t pool_cap, int depth, uint32_t *biglist) {
  bl_large::build_kernel_body((LC_GLOBAL const lancet_params *)P, (LC_GLOBAL const DevBatch *)B, (LC_GLOBAL const EngineCaps *)C, (LC_GLOBAL uint8_t *)pre, (LC_GLOBAL uint8_t *)scratch,
                    (LC_GLOBAL uint32_t *)queue, *(bl_large::BL_S *)&bl_large::bl_shared, (int)blockIdx.x, (LC_GLOBAL unsigned long long *)phase, (LC_GLOBAL uint8_t *)pool, pool_cap, depth,
                    (LC_GLOBAL uint32_t *)biglist, true);
}

// Build service (build_lds_impl.h svc_kernel_body): a few resident 512-lane workgroups that build, while the window kernel runs, the
// graphs of later k attempts it asks for.  Launched on its own stream before the batch's kernels, leaves when svc_done_kernel
// (enqueued behind the window kernel) has set SvcCtl::done.
__global__ void __launch_bounds__(bl_small::WG) __attribute__((amdgpu_waves_per_eu(BL_SMALL_EU, BL_SMALL_EU))) svc_kernel(const lancet_params *P, const DevBatch *B, const EngineCaps *C, uint8_t *pre, uint8_t *scratch, uint32_t *queue,
                                                      uint8_t *pool, uint32_t pool_cap, int depth, SvcCtl *sv, const uint32_t *wqueue, uint32_t *biglist, int help_depth, unsigned long long *phase) {
  bl_small::svc_kernel_body((LC_GLOBAL const lancet_params *)P, (LC_GLOBAL const DevBatch *)B, (LC_GLOBAL const EngineCaps *)C, (LC_GLOBAL uint8_t *)pre, (LC_GLOBAL uint8_t *)scratch,
                    (LC_GLOBAL uint32_t *)queue, *(bl_small::BL_S *)&bl_small::bl_shared, (int)blockIdx.x, (LC_GLOBAL uint8_t *)pool, pool_cap, depth, (LC_GLOBAL SvcCtl *)sv,
                    (LC_GLOBAL const uint32_t *)wqueue, (LC_GLOBAL uint32_t *)biglist, help_depth, (LC_GLOBAL unsigned long long *)phase);
}
// (the control block is set up by a kernel, not by a copy: an asynchronous copy from pageable host memory blocks the caller until the
//  stream has caught up, and with another engine's persistent kernels on the device that is the rest of their batch)
// ... and the same in the 1024-lane configuration, for a batch with windows beyond the 512-lane one's limits (one workgroup per CU)
__global__ void __launch_bounds__(bl_large::WG) __attribute__((amdgpu_waves_per_eu(4, 4))) svc_kernel_large(const lancet_params *P, const DevBatch *B, const EngineCaps *C, uint8_t *pre, uint8_t *scratch, uint32_t *queue,
                                                                 uint8_t *pool, uint32_t pool_cap, int depth, SvcCtl *sv, const uint32_t *wqueue) {
  bl_large::svc_kernel_body((LC_GLOBAL const lancet_params *)P, (LC_GLOBAL const DevBatch *)B, (LC_GLOBAL const EngineCaps *)C, (LC_GLOBAL uint8_t *)pre, (LC_GLOBAL uint8_t *)scratch,
                            (LC_GLOBAL uint32_t *)queue, *(bl_large::BL_S *)&bl_large::bl_shared, (int)blockIdx.x, (LC_GLOBAL uint8_t *)pool, pool_cap, depth, (LC_GLOBAL SvcCtl *)sv, (LC_GLOBAL const uint32_t *)wqueue);
}
__global__ void svc_init_kernel(SvcCtl *sv, SvcCtl v) { *sv = v; }
// Gate of lancet_engine_submit_after (LANCET_GATE): one wave, no LDS, at the head of this engine's stream.  It leaves when the OTHER engine's window
// kernel has taken the last window off its list (`head` >= n) -- from then on that kernel's slots leave their CUs one after the other while a few dozen
// windows that wait for the build service keep the launch alive for another ~2 ms: this engine's build kernel is dispatched onto the CUs as they empty.
// It must not become launchable earlier: whatever of it is resident when the window kernel is dispatched takes that kernel's slots away for the whole
// launch (LANCET_STAGGER, profiles/r6_stagger_rejected.txt).  No progress of the head for 2 ms (a profiler that serialises kernels, a launch that failed): leave.
__global__ void gate_kernel(const uint32_t *head, uint32_t n) {
  if (threadIdx.x != 0) return;
  unsigned long long t_prog = wall_clock64(); uint32_t last = 0xFFFFFFFFu;
  while (true) {
    const uint32_t h = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (h >= n) break;
    const unsigned long long now = wall_clock64();
    if (h != last) { last = h; t_prog = now; } else if (now - t_prog > 200000ull) break;       // 2 ms at 100 MHz
    __builtin_amdgcn_s_sleep(127);
  }
}
__global__ void svc_done_kernel(SvcCtl *sv) { __hip_atomic_store(&sv->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// Processing order of the window kernel: longest expected first.  A slot takes the next window of the list when it is through with
// one, so the launch ends one window's time after the list runs dry: what is taken last has to be short, and what may run long has to be
// under way early.  What the build kernel left in the hand-off area says how long a window will take (tools/window_cost.py,
// profiles/r5_window_order.txt; eighths of a millisecond of a slot at 30x-60x):
//   not built (general build phases in the window kernel)      very long              first
//   a read repeats a k-mer (`heavy`: k will climb, 2+ builds)   2.1 ms, up to 6
//   first compress not done here (several components)           1.3 ms .. 2.9 ms by the number of survivors
//   first compress done: nodes left in the table                3 .. 5: 0.35 ms (no bubble: nothing to align), 8: 1.1 ms, 11: 1.7 ms, 16: 2.6 ms
//   no reads / no k                                             0.1 ms                 last
// Only the order changes; results are sorted by (window, emission) afterwards.
#define ORD_CLASSES 32
#define LC_COUNTER_BYTES 512            /* u32: 0..3 results, 2 the window kernel's queue head, 8..15 the build kernels', 32..63 windows per class, 64..95 places given out per class */
__device__ uint32_t order_class(const uint8_t *area, uint32_t chdr) {
  const PreHdr *H = (const PreHdr *)(area + PRE_OFF_HDR);
  if (H->status != PB_BUILT) return (H->why == BLW_NOREADS || H->why == BLW_K) ? 0u : 31u;
  if (H->heavy) return 26u;
  const PreCmp *Cm = (const PreCmp *)(area + chdr);
  if (Cm->done) {
    const uint32_t m = Cm->m_live;
    return m <= 5u ? 3u : std::min(23u, (3u * m) / 2u - 3u); }
  const uint32_t ns = H->nsurv;
  return ns <= 520u ? 10u : std::min(23u, (ns - 400u) / 12u);
}
// (a workgroup counts its windows per class in LDS and adds once per class: 9000 windows of one class are 9000 atomics on one word otherwise, 0.17 ms)
__global__ void __launch_bounds__(256) order_class_kernel(const uint8_t *pre, uint32_t stride, uint32_t chdr, int n_windows, uint8_t *cls, uint32_t *cnt, int two) {
  __shared__ uint32_t h[ORD_CLASSES];
  if (threadIdx.x < ORD_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (w < n_windows) {
    uint32_t c = order_class(pre + (size_t)w * stride, chdr);
    if (two && c) c = c >= 26u ? 31u : 1u;          // (LANCET_ORDER=two: what will run long first, the rest as they come -- the order of round 4)
    cls[w] = (uint8_t)c;
    atomicAdd(&h[c], 1u);
  }
  __syncthreads();
  if (threadIdx.x < ORD_CLASSES && h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) order_place_kernel(int n_windows, const uint8_t *cls, const uint32_t *cnt, uint32_t *cur, uint32_t *list) {
  __shared__ uint32_t h[ORD_CLASSES], base[ORD_CLASSES];
  if (threadIdx.x < ORD_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  uint32_t c = 0, r = 0;
  if (w < n_windows) { c = cls[w]; r = atomicAdd(&h[c], 1u); }
  __syncthreads();
  if (threadIdx.x < ORD_CLASSES) {                  // the class's first place (the classes above it come before), then this workgroup's stretch of it
    const uint32_t q = threadIdx.x;
    uint32_t b = 0;
    for (uint32_t x = ORD_CLASSES - 1u; x > q; --x) b += cnt[x];
    base[q] = b + (h[q] ? atomicAdd(&cur[q], h[q]) : 0u);
  }
  __syncthreads();
  if (w < n_windows) list[base[c] + r] = (uint32_t)w;
}

// Graph_t::trim (reference src/Graph.cc:355-384) + 2-bit packing + quality mask, one wave per read: the lanes look at consecutive
// bases (a wave instruction reads 64 consecutive bytes), first / last base that is DNA with quality >= MIN_QUAL_TRIM by wave
// reduction, junk test (a non-ACGT base inside the kept part) by ballot, then one output word per lane.
__global__ void __launch_bounds__(256) prep_kernel(const lancet_params *P, int n_reads, const char *seq, const char *qual, const uint32_t *seq_off,
                            const uint8_t *label, const uint8_t *strand, const uint8_t *mate, const uint8_t *mapped,
                            uint32_t *rinfo, uint32_t *bases, const uint32_t *bw, uint32_t *good, const uint32_t *gw) {
  const int r = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63u);
  if (r >= n_reads) return;
  const uint32_t off = seq_off[r];
  const int len = (int)(seq_off[r + 1] - off);
  const char *sq = seq + off, *ql = qual + off;
  const int qtrim = P->min_qual_trim, qcall = P->min_qual_call;
  int fg = 0x7FFFFFFF, lg = -1;
  for (int p = lane; p < len; p += 64) { if (is_dna(sq[p]) && !(ql[p] < qtrim)) { if (p < fg) fg = p; if (p > lg) lg = p; } }
  for (int d = 32; d > 0; d >>= 1) { const int a = __shfl_xor(fg, d, 64), b = __shfl_xor(lg, d, 64); if (a < fg) fg = a; if (b > lg) lg = b; }
  bool junk = lg < 0;
  if (!junk) {
    bool bad = false;
    for (int p = fg + lane; p <= lg; p += 64) if (!is_dna(sq[p])) bad = true;
    junk = __ballot(bad) != 0;
  }
  const int trim5 = junk ? 0 : fg;
  int tlen = junk ? 0 : lg - fg + 1;
  if (tlen > 0xFFFF) tlen = 0xFFFF;
  if (lane == 0) rinfo[r] = (uint32_t)tlen | ((label[r] == LANCET_NML ? 1u : 0u) << 16) | ((strand[r] == LANCET_REV ? 1u : 0u) << 17) | ((uint32_t)(mate[r] & 3) << 18) | ((mapped[r] ? 1u : 0u) << 20);
  const uint32_t b0 = bw[r], g0 = gw[r];
  for (int wv = lane; wv < (tlen + 15) / 16; wv += 64) {
    uint32_t v = 0;
    for (int j = 0; j < 16 && wv * 16 + j < tlen; ++j) v |= (uint32_t)(base_code(sq[trim5 + wv * 16 + j]) & 3) << (2 * j);
    bases[b0 + wv] = v;
  }
  for (int wv = lane; wv < (tlen + 31) / 32; wv += 64) {
    uint32_t v = 0;
    for (int j = 0; j < 32 && wv * 32 + j < tlen; ++j) if (ql[trim5 + wv * 32 + j] >= qcall) v |= 1u << j;
    good[g0 + wv] = v;
  }
}
// test hook: global_align_aff alone (align_fill + align_traceback) on one pair of strings
__global__ void __launch_bounds__(LANCET_WG * 2) align_test_kernel(const EngineCaps *C, Work *work, const uint8_t *Sx, int n, const uint8_t *Tx, int m, int *out_len, int mode) {
  LC_WS &S = *(LC_WS *)&lc_shared;
  Ctx c; c.P = nullptr; c.B = nullptr; c.C = (LC_GLOBAL const EngineCaps *)C; c.W = (LC_GLOBAL Work *)work; c.OUT = nullptr; c.S = &S;
  LC_CTX_PUBLISH(c);
  WG_LANE0 { S.overflow = 0; }
  WG_SYNC();
  if (mode == 1 || !align_fill_band(c, (LC_GLOBAL const uint8_t *)Sx, n, (LC_GLOBAL const uint8_t *)Tx, m)) {     // mode 1: the full matrix only
    if (mode == 2) { WG_LANE0 { *out_len = -2; } return; }                                                          // mode 2: the band only (-2: not certified)
    align_fill(c, (LC_GLOBAL const uint8_t *)Sx, n, (LC_GLOBAL const uint8_t *)Tx, m);
  }
  { const int L = align_traceback_wg(c, (LC_GLOBAL const uint8_t *)Sx, n, (LC_GLOBAL const uint8_t *)Tx, m); WG_LANE0 { S.tmp0 = L; *out_len = S.overflow ? -1 : L; } }
  align_traceback_fill(c, (LC_GLOBAL const uint8_t *)Sx, (LC_GLOBAL const uint8_t *)Tx, wg_bcast(&S.tmp0));
}

__global__ void ref_code_kernel(const char *ref, uint8_t *codes, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) codes[i] = (uint8_t)base_code(ref[i]);
}

struct DevBuf {
  void *p = nullptr; size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = n + n / 4 + 256;
    if (hipMalloc(&p, want) != hipSuccess) {             // no room for the slack: the exact size
      (void)hipGetLastError(); want = n;
      if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return -1; }
    }
    cap = want; return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct lancet_engine {
  lancet_params params;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_done = nullptr;     // recorded by every submit behind the batch's kernels (ev1 is re-recorded by the re-run tier in lancet_engine_wait): what lancet_engine_submit_after waits for
  hipEvent_t wait_ev = nullptr;     // lancet_engine_submit_after: the event this submit's kernels wait for (set around the call only)
  hipEvent_t ev_a_done = nullptr;   // recorded behind the build kernels (and the ordering kernels) of a submit: LANCET_STAGGER=1 lets the next engine's kernels start there
  bool stagger = false;
  hipEvent_t gate_prev_done = nullptr;
  bool gate = false; const uint32_t *gate_head = nullptr; uint32_t gate_n = 0;      // LANCET_GATE: gate_kernel above (set around a submit_after only)
  int build_cycle = 3;            // windows a first-generation build workgroup takes before it leaves, in a launch behind the gate (LANCET_BUILD_CYCLE; 0: persistent as ever)
  bool cycle_now = false;         // this submit's build kernel is such a launch
  int slots2_pred = 0;              // re-run tier of the windows started early: laid out by submit before it waits for `wait_ev`
  std::string err;
  // device buffers
  DevBuf d_params, d_batch, d_caps, d_out, d_works;
  DevBuf d_chr, d_refstart, d_refoff, d_refasc, d_refcodes, d_readbegin, d_seqoff, d_seq, d_qual, d_label, d_strand, d_mate, d_mapped,
      d_rinfo, d_name, d_bw, d_gw, d_bases, d_good, d_bx, d_hp, d_varlr, d_bxblob;
  DevBuf d_variants, d_blob, d_counters, d_stats, d_evtlen, d_evt, d_workmem, d_phase;
  std::vector<unsigned long long> phase;
  EngineCaps caps;      // tier 1
  EngineCaps caps2;     // tier 2 (re-run of overflowed windows)
  DevBuf d_caps2, d_works2, d_workmem2, d_out2, d_winlist, d_skip;
  int n_slots2 = 0;
  uint32_t node_cap1 = 16384;   // tier-1 node limit per window when LANCET_NODE_CAP1 sets it (else from the batch, lc_upload)
  bool node_cap1_env = false;
  uint32_t debug_stop = 0;   // LANCET_STOP_PHASE (profiling only)
  uint32_t table_start = 0;  // LANCET_TABLE_START (testing only: exercises the table-doubling path)
  int n_windows = 0, n_reads = 0, n_slots = 0;
  int n_rerun = 0;
  // windows the host can tell will not fit tier 1 (more reads than its slots hold: coverage pile-ups) go to the re-run tier at
  // once, on a second stream, while tier 1 works through the rest
  std::vector<uint32_t> pred;
  std::vector<uint8_t> is_pred;
  hipStream_t stream2 = nullptr;
  hipEvent_t evf0 = nullptr, evf1 = nullptr, ev_ready = nullptr;
  bool fat_inflight = false;
  float ms_fat = 0;
  // LDS build kernel: hand-off areas (one per window), per-workgroup scratch
  DevBuf d_pre, d_blscratch, d_blphase, d_order, d_prepool, d_blscratch_large, d_biglist;
  int n_bslots_large = 0, n_biglist = -1;          // n_biglist: windows the last batch handed to the 1024-lane configuration (-1: no batch yet)
  uint32_t pool_cap = 0; int ahead_depth = 6;      // graphs built ahead for windows whose k will climb (build_lds.h, build_kernel_body)
  int n_ahead_built = 0, n_ahead_used = 0;
  bool heavy_first = true;    // LANCET_NO_HEAVY_FIRST=1: windows in batch order
  int order_mode = 0;         // LANCET_ORDER=two: two classes only (see order_class)
  unsigned long long blphase[16] = {0};
  int n_bslots = 0, n_prebuilt = 0;
  bool prebuild = true;       // LANCET_NO_PREBUILD=1: every window through the general build phases (comparison / debugging)
  hipEvent_t evb0 = nullptr, evb1 = nullptr;
  // build service (svc_kernel): control block + request / ready / continuation arrays in one buffer, scratch of its workgroups
  DevBuf d_svc, d_svcscratch;
  hipStream_t stream3 = nullptr; hipEvent_t ev_svc = nullptr;
  bool svc = true, svc_running = false, svc_large = false;
  bool svc_help = true;                      // the service's workgroups take windows off the build kernel's queue until the window kernel runs (LANCET_SVC_HELP=0: they only wait)      // LANCET_NO_SVC=1 (at create): every later graph of a window by the general build
  int n_svc_wgs = 32; uint32_t svc_cap = 0; int svc_depth = 6;      // (round 6: 32 -- with the window kernel's slot time down by an eighth its last millisecond is windows waiting for the service: 11.9 -> 11.6 ms; beyond 32 their LDS costs it more slots than the shorter waits give back: 36 -> 12.0, 48 -> 13.3 ms)
  int svc_cus = 0, n_cus = 256;                // LANCET_SVC_CUS=n: CUs set aside for the service (CU masks on the two streams), so that its workgroups are resident
                                             // whatever the batch's kernels -- or another engine's -- occupy; 0 = no masks
  SvcCtl svc_host;
  uint32_t svc_counts[5] = {0, 0, 0, 0, 0};  // posted, built, failed, stolen of the last run ; service workgroups that gave up waiting
  float ms_build = 0, ms_window = 0;
  bool uploaded = false, ran = false, submitted = false;
  // upload: reads are trimmed and packed by host threads into a pinned staging buffer that mirrors one device buffer (one DMA);
  // LANCET_PREP=device keeps the round-2 path (ASCII bases + qualities to the device, prep_kernel there)
  void *h_stage = nullptr; size_t h_stage_cap = 0; DevBuf d_stage;
  bool host_prep = true; int prep_threads = 0;
  int prep_threads_auto = 1;       // hardware threads, at most 96 -- and at most twice the container's CPU quota (cgroup cpu.max): measured on a 16-CPU quota, 32 threads pack a batch in 26 ms, 16 in 39, 96 in 31
  int exact_need_large = -1;       // host trim: windows that exceed the 512-lane build configuration (exact: from the trimmed lengths); -1 unknown
  float ms_pack = 0;
  bool phase_times = false;        // LANCET_PHASE_TIMES=1: lancet_engine_phase_times / _build_phase_times return ticks (else zeros)
  bool up_timing = false;          // LANCET_UPLOAD_TIMING=1: where lancet_engine_upload spends its time (allocations, packing, the copy), on stderr
  bool dbg = false, no_fat = false, no_early_rerun = false, no_large_build = false;     // LANCET_DEBUG / LANCET_NO_FAT / ... read once, at create
  int build_slots_env = 0, ahead_depth_env = -1;
  int pre_wide_env = -1;           // LANCET_PRE_WIDE=0 / 1: the narrow / wide form of the hand-off areas whatever the batch looks like
  uint32_t evt_cap = 0;
  size_t mem_budget = (size_t)96 << 30;
  int max_slots = 5120;      // work-space slots = resident single-wave workgroups: 5 per SIMD (96 VGPRs, < 8 KB LDS each) x 4 SIMDs x CUs
  uint32_t max_nodes_limit = 65536; bool max_nodes_env = false;
  // host results
  std::vector<lancet_variant> variants;
  std::vector<char> blob;
  std::vector<lancet_window_stats> stats;
  std::vector<uint32_t> evt_len, evt;
  std::vector<lancet_variant_lr> variants_lr;   // lr_mode: parallel to variants
  std::vector<uint32_t> bx_blob;
  float ms_all = 0, ms_kernel = 0;
};


// Copies of an engine go through ITS stream and wait for that stream only: hipMemcpy on the legacy default stream would also wait
// for the kernels of every other engine on the device (two engines take turns on one GPU).
static hipError_t lc_copy(lancet_engine *e, void *dst, const void *src, size_t bytes, hipMemcpyKind kind);

// The stream of the build service.  Its kernel is resident for the whole batch and must run BESIDE the batch's kernels; the runtime maps
// streams onto a few hardware queues (four by default), and two streams that land on one queue run their kernels one after the other --
// the service would then hold up the very kernels it waits for, until its 300 ms give-up timer.  A stream of another priority gets its
// queue from another pool, so the service's never shares one with an ordinary stream (LANCET_SVC_PRIO=0: an ordinary stream).
static hipError_t lc_service_stream(hipStream_t *s) {
  const char *env = getenv("LANCET_SVC_PRIO");
  if (!env || atoi(env) != 0) {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo && hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    *s = nullptr;
  }
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

static hipError_t lc_copy(lancet_engine *e, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  hipError_t r = hipMemcpyAsync(dst, src, bytes, kind, e->stream);
  if (r != hipSuccess) return r;
  return hipStreamSynchronize(e->stream);
}

#include "host_pack.h"      // lc_prep_read_host: Graph_t::trim + 2-bit packing + quality mask of one read on the host (the scalar twin of prep_kernel)
template <class F> static void lc_parallel(int threads, size_t n, F body) {       // body(lo, hi, t) over a static split of [0, n)
  if (threads <= 1 || n < 4096) { body((size_t)0, n, 0); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t) th.emplace_back([=] { body(n * (size_t)t / threads, n * (size_t)(t + 1) / threads, t); });
  for (auto &x : th) x.join();
}

extern "C" {

void lancet_params_default(lancet_params *p) {
  memset(p, 0, sizeof(*p));
  p->min_k = 11; p->max_k = 101; p->max_tip_len = 11; p->cov_threshold = 5; p->low_cov_threshold = 1; p->dfs_limit = 1000000;
  p->max_indel_len = 500; p->max_mismatch = 2; p->min_qual_trim = 10 + 33; p->min_qual_call = 17 + 33; p->max_unit_len = 4;
  p->min_report_units = 3; p->min_report_len = 7; p->dist_from_str = 1; p->lr_mode = 0; p->min_cov_ratio = 0.01;
}

int lancet_engine_create(const lancet_params *p, int device, lancet_engine **out) {
  if (!p || !out) return LANCET_E_ARG;
  *out = nullptr;
  if (p->max_k > 127 || p->min_k < 3 || p->max_unit_len > 8) return LANCET_E_UNSUPPORTED;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return LANCET_E_NO_DEVICE;
  lancet_engine *e = new lancet_engine();
  e->params = *p; e->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete e; return LANCET_E_HIP; }
  if (getenv("LANCET_NO_SVC") || getenv("LANCET_NO_PREBUILD") || (p->lr_mode && getenv("LANCET_LR_PREBUILD") && atoi(getenv("LANCET_LR_PREBUILD")) == 0)) e->svc = false;
  if (const char *s = getenv("LANCET_SVC_CUS")) e->svc_cus = std::max(0, std::min(64, atoi(s)));
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->n_cus = cus; }
  if (!e->svc || e->svc_cus * 4 > e->n_cus) e->svc_cus = 0;
  if (e->svc_cus) {
    // Build service on its own CUs (spread over the device: every n_cus / svc_cus-th one); the batch's kernels on the others.
    const uint32_t words = (uint32_t)((e->n_cus + 31) / 32);
    std::vector<uint32_t> m_main(words, 0), m_svc(words, 0);
    const int step = e->n_cus / e->svc_cus;
    for (int c = 0; c < e->n_cus; ++c) { const bool sv = (c % step) == step / 2 && c / step < e->svc_cus; (sv ? m_svc : m_main)[(size_t)c >> 5] |= 1u << (c & 31); }
    if (hipExtStreamCreateWithCUMask(&e->stream, words, m_main.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&e->stream3, words, m_svc.data()) != hipSuccess) {
      (void)hipGetLastError();
      if (e->stream) { (void)hipStreamDestroy(e->stream); e->stream = nullptr; }
      if (e->stream3) { (void)hipStreamDestroy(e->stream3); e->stream3 = nullptr; }
      e->svc_cus = 0;
    }
  }
  if ((!e->stream && hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) || hipEventCreate(&e->ev0) != hipSuccess ||
      hipEventCreate(&e->ev1) != hipSuccess || hipEventCreate(&e->evb0) != hipSuccess || hipEventCreate(&e->evb1) != hipSuccess ||
      hipEventCreate(&e->ev_done) != hipSuccess || hipEventCreate(&e->ev_a_done) != hipSuccess) { delete e; return LANCET_E_HIP; }
  e->stagger = getenv("LANCET_STAGGER") && atoi(getenv("LANCET_STAGGER")) != 0;
  e->gate = !(getenv("LANCET_GATE") && atoi(getenv("LANCET_GATE")) == 0) && !e->stagger;       // (on unless LANCET_GATE=0)
  if (const char *s = getenv("LANCET_BUILD_CYCLE")) e->build_cycle = std::max(0, atoi(s));
  if (getenv("LANCET_NO_PREBUILD")) e->prebuild = false;
  if (const char *s = getenv("LANCET_PREP")) e->host_prep = strcmp(s, "device") != 0;
  if (const char *s = getenv("LANCET_PREP_THREADS")) e->prep_threads = std::max(1, atoi(s));
  {
    unsigned hw = std::max(1u, std::min(96u, std::thread::hardware_concurrency()));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0}; long per = 0;
      if (fscanf(f, "%31s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { const long cpus = 2 * ((atol(q) + per - 1) / per); if (cpus >= 2 && (unsigned long)cpus < hw) hw = (unsigned)cpus; }
      fclose(f);
    }
    e->prep_threads_auto = (int)hw;
  }
  e->up_timing = getenv("LANCET_UPLOAD_TIMING") != nullptr;
  e->phase_times = getenv("LANCET_PHASE_TIMES") != nullptr && atoi(getenv("LANCET_PHASE_TIMES")) != 0;      // per-phase tick accounting of both kernels (tools/quick_gpu.py ...): clock reads per phase and contended atomics per window otherwise
  e->dbg = getenv("LANCET_DEBUG") != nullptr; e->no_fat = getenv("LANCET_NO_FAT") != nullptr; e->no_early_rerun = getenv("LANCET_NO_EARLY_RERUN") != nullptr;
  e->no_large_build = getenv("LANCET_NO_LARGE_BUILD") != nullptr; e->heavy_first = getenv("LANCET_NO_HEAVY_FIRST") == nullptr;
  if (const char *s = getenv("LANCET_ORDER")) e->order_mode = strcmp(s, "two") == 0 ? 1 : 0;
  if (const char *s = getenv("LANCET_BUILD_SLOTS")) e->build_slots_env = std::max(1, atoi(s));
  if (const char *s = getenv("LANCET_AHEAD_DEPTH")) e->ahead_depth_env = std::max(0, std::min(16, atoi(s)));
  if (const char *s = getenv("LANCET_SVC_HELP")) e->svc_help = atoi(s) != 0;
  if (const char *s = getenv("LANCET_PRE_WIDE")) e->pre_wide_env = atoi(s) != 0 ? 1 : 0;
  if (const char *s = getenv("LANCET_SVC_WGS")) e->n_svc_wgs = std::max(0, std::min(256, atoi(s)));
  if (const char *s = getenv("LANCET_SVC_DEPTH")) e->svc_depth = std::max(0, std::min(16, atoi(s)));
  // --linked-reads: the build kernel builds a window's graphs as for any other batch (graphs ahead, the build service) and hands the tracked
  // nodes' occurrences over with each; the window kernel replays barcodes and haplotypes over them (kernels.h load_prebuilt_lr).
  // LANCET_LR_PREBUILD=0: the general build for every window (the route until round 5).
  if (p->lr_mode) { if (const char *s = getenv("LANCET_LR_PREBUILD")) { if (atoi(s) == 0) e->prebuild = false; } }
  if (const char *s = getenv("LANCET_TRACE_WORDS")) e->evt_cap = (uint32_t)atoi(s);
  e->max_slots = (e->n_cus - e->svc_cus) * 4 * LC_W_EU - 4 * LC_W_EU;       // (one CU's worth short of the device: see n_bslots)
  if (e->svc_cus) e->n_svc_wgs = 2 * e->svc_cus;
  if (const char *s = getenv("LANCET_MAX_SLOTS")) e->max_slots = atoi(s);
  if (const char *s = getenv("LANCET_MEM_GB")) e->mem_budget = (size_t)atoi(s) << 30;
  if (const char *s = getenv("LANCET_MAX_NODES")) { e->max_nodes_limit = (uint32_t)atoi(s); e->max_nodes_env = true; }
  if (e->max_nodes_limit > (1u << 22)) e->max_nodes_limit = 1u << 22;        // (sequence descriptors keep the k-mer node in 23 bits: layout.h SD_KMER)
  if (e->max_nodes_limit < 1024u) e->max_nodes_limit = 1024u;
  if (const char *s = getenv("LANCET_NODE_CAP1")) { e->node_cap1 = (uint32_t)atoi(s); e->node_cap1_env = true; }
  if (const char *s = getenv("LANCET_STOP_PHASE")) e->debug_stop = (uint32_t)atoi(s);
  if (const char *s = getenv("LANCET_TABLE_START")) e->table_start = lc_pow2_ge((uint32_t)atoi(s));
  *out = e;
  return LANCET_OK;
}

void lancet_engine_destroy(lancet_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  DevBuf *all[] = {&e->d_params, &e->d_batch, &e->d_caps, &e->d_out, &e->d_works, &e->d_chr, &e->d_refstart, &e->d_refoff, &e->d_refasc,
                   &e->d_refcodes, &e->d_readbegin, &e->d_seqoff, &e->d_seq, &e->d_qual, &e->d_label, &e->d_strand, &e->d_mate, &e->d_mapped,
                   &e->d_rinfo, &e->d_name, &e->d_bw, &e->d_gw, &e->d_bases, &e->d_good, &e->d_variants, &e->d_blob, &e->d_counters,
                   &e->d_stats, &e->d_evtlen, &e->d_evt, &e->d_workmem, &e->d_phase, &e->d_caps2, &e->d_works2, &e->d_workmem2, &e->d_out2, &e->d_winlist, &e->d_skip, &e->d_bx, &e->d_hp, &e->d_varlr, &e->d_bxblob, &e->d_pre, &e->d_blscratch, &e->d_blphase, &e->d_order, &e->d_prepool, &e->d_blscratch_large, &e->d_biglist, &e->d_svc, &e->d_svcscratch, &e->d_stage};
  for (DevBuf *b : all) b->release();
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  if (e->evb0) (void)hipEventDestroy(e->evb0);
  if (e->evb1) (void)hipEventDestroy(e->evb1);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->ev_done) (void)hipEventDestroy(e->ev_done);
  if (e->ev_a_done) (void)hipEventDestroy(e->ev_a_done);
  if (e->evf0) (void)hipEventDestroy(e->evf0);
  if (e->evf1) (void)hipEventDestroy(e->evf1);
  if (e->ev_ready) (void)hipEventDestroy(e->ev_ready);
  if (e->ev_svc) (void)hipEventDestroy(e->ev_svc);
  if (e->stream3) (void)hipStreamDestroy(e->stream3);
  if (e->stream2) (void)hipStreamDestroy(e->stream2);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

const char *lancet_engine_last_error(const lancet_engine *e) { return e ? e->err.c_str() : "null engine"; }

// debug knob used by the tests: number of 32-bit trace words kept per window (0 = off)
int lancet_engine_set_trace(lancet_engine *e, uint32_t words_per_window) { if (!e) return LANCET_E_ARG; e->evt_cap = words_per_window; return LANCET_OK; }

static int up(lancet_engine *e, DevBuf &b, const void *src, size_t bytes) {
  if (b.ensure(bytes ? bytes : 1)) { e->err = "hipMalloc failed"; return LANCET_E_OOM; }
  if (bytes) HIPCHK(e, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, e->stream));
  return LANCET_OK;
}
#define DBG(msg) do { if (e->dbg) { fprintf(stderr, "[lancet] %s:%d %s\n", __func__, __LINE__, msg); fflush(stderr); } } while (0)
#define UP(buf, src, bytes) do { DBG(#buf); int _rc = up(e, buf, src, bytes); if (_rc) return _rc; } while (0)
#define ENS(buf, bytes) do { if ((buf).ensure((bytes) ? (bytes) : 1)) { e->err = "hipMalloc failed"; return LANCET_E_OOM; } } while (0)

static int lc_upload(lancet_engine *e, const lancet_window_batch *b, const lancet_packed_reads *pk);
int lancet_engine_upload(lancet_engine *e, const lancet_window_batch *b) { return lc_upload(e, b, nullptr); }
// The reads arrive trimmed and packed (lancet_pack_read with THIS engine's parameters: the caller's threads did what the upload's own do
// from the ASCII arrays); b->seq / qual / label / strand / mate / mapped are not looked at, b->seq_off still gives the untrimmed lengths.
int lancet_engine_upload_packed(lancet_engine *e, const lancet_window_batch *b, const lancet_packed_reads *pk) {
  if (e && pk && pk->struct_size != (uint32_t)sizeof(lancet_packed_reads)) {
    e->err = "lancet_packed_reads: struct_size " + std::to_string(pk->struct_size) + " is not this library's " + std::to_string(sizeof(lancet_packed_reads)) + " (caller built against another header)";
    return LANCET_E_ARG;
  }
  if (!e || !pk || !pk->rinfo || !pk->base_woff || !pk->good_woff || !pk->bases || !pk->good) { if (e) e->err = "packed reads missing"; return LANCET_E_ARG; }
  if (!e->host_prep) { e->err = "packed upload with LANCET_PREP=device"; return LANCET_E_STATE; }
  if (pk->min_qual_trim != e->params.min_qual_trim || pk->min_qual_call != e->params.min_qual_call) {
    e->err = "packed reads were trimmed / masked with other thresholds (min_qual_trim " + std::to_string(pk->min_qual_trim) + ", min_qual_call " + std::to_string(pk->min_qual_call) +
             ") than this engine's (" + std::to_string(e->params.min_qual_trim) + ", " + std::to_string(e->params.min_qual_call) + ")";
    return LANCET_E_ARG;
  }
  return lc_upload(e, b, pk);
}
void lancet_pack_read(const lancet_params *P, const char *seq, const char *qual, int len, uint8_t label, uint8_t strand, uint8_t mate, uint8_t mapped,
                      uint32_t *rinfo, uint32_t *bases, uint32_t *good) {
  lc_prep_read_host(*P, seq, qual, len, label, strand, mate, mapped, rinfo, bases, good);
  const uint32_t tl = RI_TLEN(*rinfo);
  for (uint32_t wv = (tl + 15) / 16; wv < ((uint32_t)len + 15) / 16; ++wv) bases[wv] = 0;
  for (uint32_t wv = (tl + 31) / 32; wv < ((uint32_t)len + 31) / 32; ++wv) good[wv] = 0;
}
static int lc_upload(lancet_engine *e, const lancet_window_batch *b, const lancet_packed_reads *pk) {
  if (!e || !b || b->n_windows < 0) return LANCET_E_ARG;
  if (e->submitted) { e->err = "upload while a batch is in flight"; return LANCET_E_STATE; }
  HIPCHK(e, hipSetDevice(e->device));
  e->uploaded = false; e->ran = false;
  auto t_up0 = std::chrono::steady_clock::now();
  auto tick = [&](const char *what, size_t bytes = 0) {
    if (!e->up_timing) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[lancet upload] %-34s %8.1f ms", what, std::chrono::duration<double, std::milli>(t - t_up0).count());
    if (bytes) fprintf(stderr, "  (%.2f GB)", bytes / 1073741824.0);
    fprintf(stderr, "\n");
    t_up0 = t;
  };
  const int nw = b->n_windows;
  if (e->params.lr_mode && nw > 0 && b->read_begin[nw] > 0 && (!b->bx_rank || !b->hp)) { e->err = "lr_mode needs bx_rank and hp"; return LANCET_E_ARG; }
  const uint32_t R = nw ? b->read_begin[nw] : 0;
  const uint32_t nbases = R ? b->seq_off[R] : 0;
  const uint32_t nref = nw ? b->ref_off[nw] : 0;
  e->n_windows = nw; e->n_reads = (int)R;
  if (nw == 0) { e->uploaded = true; return LANCET_OK; }
  // (a window longer than LC_MAXW does not fail the batch: process_window reports it LANCET_W_OVERFLOW on its own and every other
  //  window is assembled.  A window of more than 65 535 reads -- the reference takes up to MAX_AVG_COV = 10 000x per sample,
  //  src/Microassembler.cc:491-496 -- overflows the one-wave kernel's 16-bit read ids and runs in the re-run tier, whose csr words and
  //  mate-name records keep the read in 32 bits (layout.h cs_t); its node tables are then sized for up to 2^20 distinct k-mers
  //  instead of 65 536, unless LANCET_MAX_NODES says otherwise.)
  // Tier-1 node limit: from the batch unless LANCET_NODE_CAP1 says otherwise -- a quarter of the mean window's bases as distinct k-mers
  // (30x/30x: ~2.1 k nodes of 27 k bases), a power of two between 8192 and 16384 (16384 for every batch until round 5: the node-sized
  // arrays are a third of a slot), and never below what a hand-off area of the LDS build kernel may hold (set further down).
  uint32_t node_cap1 = e->node_cap1;
  if (!e->node_cap1_env && nw > 0) {
    const uint64_t mean_bases = ((uint64_t)b->seq_off[b->read_begin[nw]] + b->ref_off[nw]) / (uint64_t)nw;
    node_cap1 = mean_bases / 4 <= 8192 ? 8192u : 16384u;
  }
  e->caps = lc_caps_for_batch(b, &e->params, e->evt_cap, node_cap1, 1);
  uint32_t nodes2 = e->max_nodes_limit;
  if (!e->max_nodes_env) for (int w = 0; w < nw; ++w) if (b->read_begin[w + 1] - b->read_begin[w] >= 0xFFFFu) { nodes2 = std::max(nodes2, 1u << 20); break; }
  e->caps2 = lc_caps_for_batch(b, &e->params, e->evt_cap, nodes2, 2);
  e->caps2.wide_ids = e->no_fat ? 0u : 1u;                     // (LANCET_NO_FAT: the one-wave kernel re-runs, with its 16-bit limit)
  e->caps2.var_cap = e->caps.var_cap; e->caps2.blob_cap = e->caps.blob_cap; e->caps2.bx_cap = e->caps.bx_cap;
  e->caps.debug_stop = e->debug_stop;
  e->caps.table_start = e->caps2.table_start = e->table_start;
  {   // hand-off areas of the LDS build kernel: narrow or wide (host_common.h lc_pre_layout_for_batch), one per window + the pool
    const bool use_svc0 = e->svc && e->n_svc_wgs > 0 && !e->debug_stop;
    const int depth0 = e->ahead_depth_env >= 0 ? e->ahead_depth_env : 6;
    const size_t n_areas = (size_t)nw + (depth0 > 0 ? (size_t)std::max(64, nw / 4) : 0) + (use_svc0 ? (size_t)std::max(64, nw / 8) : 0);
    e->caps.pl = e->caps2.pl = lc_pre_layout_for_batch(b, n_areas, (size_t)20 << 30, e->pre_wide_env, e->params.lr_mode != 0);
    if (!e->node_cap1_env && e->caps.pl.ncap + 64u > node_cap1) {          // (wide hand-off areas: 14 336 nodes may arrive from the build kernel)
      const PreLayout pl = e->caps.pl;
      e->caps = lc_caps_for_batch(b, &e->params, e->evt_cap, 16384u, 1);
      e->caps.debug_stop = e->debug_stop; e->caps.table_start = e->table_start; e->caps.pl = pl;
      e->caps2.var_cap = e->caps.var_cap; e->caps2.blob_cap = e->caps.blob_cap; e->caps2.bx_cap = e->caps.bx_cap;
    }
  }
  // ---- inputs
  UP(e->d_params, &e->params, sizeof(lancet_params));
  DevBatch db;
  db.n_windows = nw;
  if (e->host_prep) {
    // One staging buffer, same layout on both sides: window arrays, per-read words, packed bases / quality masks, reference codes.
    const int T = e->prep_threads > 0 ? e->prep_threads : e->prep_threads_auto;
    const auto t_pack0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> tb((size_t)T + 1, 0), tg((size_t)T + 1, 0);
    std::vector<char> bad((size_t)T, 0);
    lc_parallel(T, (size_t)R, [&](size_t lo, size_t hi, int t) {
      uint64_t bo = 0, go = 0; char bd = 0;
      for (size_t r = lo; r < hi; ++r) {
        const uint32_t len = b->seq_off[r + 1] - b->seq_off[r]; bo += (len + 15) / 16; go += (len + 31) / 32;
        if (b->name_rank[r] > 0xFFFFu) {              // fine in a window of more than 65 535 reads (the re-run tier's), if it is a dense rank
          const uint32_t *ub = std::upper_bound(b->read_begin, b->read_begin + nw + 1, (uint32_t)r);
          const size_t w = (size_t)(ub - b->read_begin) - 1;
          if (b->name_rank[r] >= b->read_begin[w + 1] - b->read_begin[w]) bd = 1;
        }
        if (e->params.lr_mode && b->hp[r] > 2) bd = 2;
        if (!pk && ((b->label[r] != LANCET_TMR && b->label[r] != LANCET_NML) || (b->strand[r] != LANCET_FWD && b->strand[r] != LANCET_REV) || b->mate[r] > 2)) bd = 3;
      }
      tb[(size_t)t + 1] = bo; tg[(size_t)t + 1] = go; bad[(size_t)t] = bd;
    });
    for (int t = 0; t < T; ++t) {
      if (bad[(size_t)t] == 1) { e->err = "name_rank must be the dense per-window rank (< the window's reads)"; return LANCET_E_ARG; }
      if (bad[(size_t)t] == 2) { e->err = "hp must be 0, 1 or 2"; return LANCET_E_ARG; }      // Node_t::addHP indexes a 3-array (src/Node.cc:54-57)
      if (bad[(size_t)t] == 3) { e->err = "label must be LANCET_TMR / LANCET_NML, strand LANCET_FWD / LANCET_REV, mate 0, 1 or 2"; return LANCET_E_ARG; }
      tb[(size_t)t + 1] += tb[(size_t)t]; tg[(size_t)t + 1] += tg[(size_t)t];
    }
    // packed reads stored ONCE per batch (pk->read_index: a read of a window -> one of pk->n_distinct reads; an alignment lies in ~6 of the
    // overlapping windows): the per-read words of the device batch point into the one copy, whose words are staged and copied once
    const bool shared = pk && pk->read_index;
    if (shared && pk->n_distinct == 0 && R > 0) { e->err = "packed reads: read_index without distinct reads"; return LANCET_E_ARG; }
    const uint64_t bo_all = shared ? (uint64_t)pk->base_woff[pk->n_distinct] : tb[(size_t)T], go_all = shared ? (uint64_t)pk->good_woff[pk->n_distinct] : tg[(size_t)T];
    if (pk && !shared && ((uint64_t)pk->base_woff[R] != bo_all || (uint64_t)pk->good_woff[R] != go_all)) { e->err = "packed reads: word offsets do not match the read lengths"; return LANCET_E_ARG; }
    if (bo_all + 4 > 0xFFFFFFFFull) { e->err = "batch too large"; return LANCET_E_ARG; }
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_chr = take(4 * (size_t)nw), o_rs = take(4 * (size_t)nw), o_ro = take(4 * ((size_t)nw + 1)), o_rb = take(4 * ((size_t)nw + 1));
    const size_t o_ri = take(4 * ((size_t)R + 1)), o_nm = take(4 * (size_t)R), o_bw = take(4 * ((size_t)R + 1)), o_gw = take(4 * ((size_t)R + 1));
    const size_t o_ba = take(4 * ((size_t)bo_all + 4)), o_go = take(4 * ((size_t)go_all + 1)), o_rc = take((size_t)nref + 1);
    const size_t o_bx = e->params.lr_mode ? take(4 * (size_t)R) : 0, o_hp = e->params.lr_mode ? take((size_t)R) : 0;
    const size_t total = off;
    if (total > e->h_stage_cap) {
      if (e->h_stage) (void)hipHostFree(e->h_stage);
      e->h_stage = nullptr; e->h_stage_cap = 0;
      const size_t want = total + total / 8;
      if (hipHostMalloc(&e->h_stage, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); e->err = "hipHostMalloc failed (staging)"; return LANCET_E_OOM; }
      e->h_stage_cap = want;
    }
    ENS(e->d_stage, total);
    char *H = (char *)e->h_stage;
    memcpy(H + o_chr, b->chr_id, 4 * (size_t)nw); memcpy(H + o_rs, b->ref_start, 4 * (size_t)nw);
    memcpy(H + o_ro, b->ref_off, 4 * ((size_t)nw + 1)); memcpy(H + o_rb, b->read_begin, 4 * ((size_t)nw + 1));
    uint32_t *h_ri = (uint32_t *)(H + o_ri), *h_nm = (uint32_t *)(H + o_nm), *h_bw = (uint32_t *)(H + o_bw), *h_gw = (uint32_t *)(H + o_gw);
    uint32_t *h_ba = (uint32_t *)(H + o_ba), *h_go = (uint32_t *)(H + o_go);
    const lancet_params P = e->params;
    lc_parallel(T, (size_t)R, [&](size_t lo, size_t hi, int t) {
      uint64_t bo = tb[(size_t)t], go = tg[(size_t)t];
      if (shared) {                                           // every read of a window names its distinct read; the distinct reads' words are copied below
        const uint32_t nd = pk->n_distinct;
        for (size_t r = lo; r < hi; ++r) {
          const uint32_t len = b->seq_off[r + 1] - b->seq_off[r], u = pk->read_index[r];
          if (u >= nd || pk->base_woff[u + 1] < pk->base_woff[u] || pk->good_woff[u + 1] < pk->good_woff[u] || (uint64_t)pk->base_woff[u + 1] > bo_all || (uint64_t)pk->good_woff[u + 1] > go_all ||
              pk->base_woff[u + 1] - pk->base_woff[u] != (len + 15) / 16 || pk->good_woff[u + 1] - pk->good_woff[u] != (len + 31) / 32 || RI_TLEN(pk->rinfo[u]) > len) { bad[(size_t)t] = 3; return; }
          h_bw[r] = pk->base_woff[u]; h_gw[r] = pk->good_woff[u]; h_nm[r] = b->name_rank[r]; h_ri[r] = pk->rinfo[u];
        }
        return;
      }
      if (pk) {                                               // packed by the caller: copied as they are (the offsets were checked above in total, here per read)
        for (size_t r = lo; r < hi; ++r) {
          const uint32_t len = b->seq_off[r + 1] - b->seq_off[r];
          if (pk->base_woff[r] != (uint32_t)bo || pk->good_woff[r] != (uint32_t)go || RI_TLEN(pk->rinfo[r]) > len) { bad[(size_t)t] = 3; return; }
          h_bw[r] = (uint32_t)bo; h_gw[r] = (uint32_t)go; h_nm[r] = b->name_rank[r]; h_ri[r] = pk->rinfo[r];
          bo += (len + 15) / 16; go += (len + 31) / 32;
        }
        const uint64_t b0 = tb[(size_t)t], g0 = tg[(size_t)t];
        memcpy(h_ba + b0, pk->bases + b0, 4 * (size_t)(bo - b0)); memcpy(h_go + g0, pk->good + g0, 4 * (size_t)(go - g0));
        return;
      }
      for (size_t r = lo; r < hi; ++r) {
        const uint32_t o = b->seq_off[r], len = b->seq_off[r + 1] - o;
        h_bw[r] = (uint32_t)bo; h_gw[r] = (uint32_t)go; h_nm[r] = b->name_rank[r];
        lc_prep_read_host(P, b->seq + o, b->qual + o, (int)len, b->label[r], b->strand[r], b->mate[r], b->mapped[r], &h_ri[r], h_ba + bo, h_go + go);
        // (the words a trimmed read does not fill stay unread: every consumer goes by the trimmed length; zero them anyway so that the staging buffer is deterministic)
        const uint32_t tl = RI_TLEN(h_ri[r]);
        for (uint32_t wv = (tl + 15) / 16; wv < (len + 15) / 16; ++wv) h_ba[bo + wv] = 0;
        for (uint32_t wv = (tl + 31) / 32; wv < (len + 31) / 32; ++wv) h_go[go + wv] = 0;
        bo += (len + 15) / 16; go += (len + 31) / 32;
      }
    });
    for (int t = 0; t < T; ++t) if (bad[(size_t)t] == 3) { e->err = "packed reads: word offsets / trimmed lengths / read indices do not match the read lengths"; return LANCET_E_ARG; }
    if (shared) {                                             // the one copy of the words
      lc_parallel(T, (size_t)bo_all, [&](size_t lo, size_t hi, int) { memcpy(h_ba + lo, pk->bases + lo, 4 * (hi - lo)); });
      lc_parallel(T, (size_t)go_all, [&](size_t lo, size_t hi, int) { memcpy(h_go + lo, pk->good + lo, 4 * (hi - lo)); });
    }
    h_ri[R] = 0; h_bw[R] = (uint32_t)bo_all; h_gw[R] = (uint32_t)go_all;
    for (int i = 0; i < 4; ++i) h_ba[bo_all + (uint64_t)i] = 0;
    h_go[go_all] = 0;
    uint8_t *h_rc = (uint8_t *)(H + o_rc);
    lc_parallel(T, (size_t)nref, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) h_rc[i] = (uint8_t)lc_code(b->ref_bases[i]); });
    h_rc[nref] = 0;
    if (e->params.lr_mode && R) { memcpy(H + o_bx, b->bx_rank, 4 * (size_t)R); memcpy(H + o_hp, b->hp, (size_t)R); }
    {   // the 512-lane build configuration's size test (build_lds_impl.h: nbw / ngw against BL_BASES), window by window on the trimmed lengths
      std::vector<int> need((size_t)T, 0);
      lc_parallel(T, (size_t)nw, [&](size_t lo, size_t hi, int t) {
        int cnt = 0;
        for (size_t w = lo; w < hi; ++w) {
          const uint32_t r0 = b->read_begin[w], r1 = b->read_begin[w + 1], rl = b->ref_off[w + 1] - b->ref_off[w];
          uint64_t wb = (rl + 15) / 16, wg = (rl + 31) / 32;
          for (uint32_t r = r0; r < r1; ++r) { const uint32_t tl = RI_TLEN(h_ri[r]); wb += (tl + 15) / 16; wg += (tl + 31) / 32; }
          if (r1 - r0 > bl_small::LDS_READS || wb > bl_small::LDS_BASES / 16u || wg > bl_small::LDS_BASES / 32u) ++cnt;
        }
        need[(size_t)t] = cnt;
      });
      e->exact_need_large = 0;
      for (int t = 0; t < T; ++t) e->exact_need_large += need[(size_t)t];
    }
    e->ms_pack = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_pack0).count();
    if (e->dbg) fprintf(stderr, "[lancet] trim + pack on %d host threads: %.1f ms, %.1f MB to the device\n", T, e->ms_pack, total / 1048576.0);
    HIPCHK(e, hipMemcpyAsync(e->d_stage.p, e->h_stage, total, hipMemcpyHostToDevice, e->stream));
    char *D = (char *)e->d_stage.p;
    db.chr_id = (LC_GLOBAL const int32_t *)(D + o_chr); db.ref_start = (LC_GLOBAL const int32_t *)(D + o_rs);
    db.ref_off = (LC_GLOBAL const uint32_t *)(D + o_ro); db.ref_codes = (LC_GLOBAL const uint8_t *)(D + o_rc); db.read_begin = (LC_GLOBAL const uint32_t *)(D + o_rb);
    db.rinfo = (LC_GLOBAL const uint32_t *)(D + o_ri); db.name_rank = (LC_GLOBAL const uint32_t *)(D + o_nm); db.base_woff = (LC_GLOBAL const uint32_t *)(D + o_bw);
    db.good_woff = (LC_GLOBAL const uint32_t *)(D + o_gw); db.bases = (LC_GLOBAL const uint32_t *)(D + o_ba); db.good = (LC_GLOBAL const uint32_t *)(D + o_go);
    db.bx_rank = e->params.lr_mode ? (LC_GLOBAL const uint32_t *)(D + o_bx) : nullptr; db.hp = e->params.lr_mode ? (LC_GLOBAL const uint8_t *)(D + o_hp) : nullptr;
  } else {
  for (uint32_t r = 0; r < R; ++r) if (b->name_rank[r] > 0xFFFFu) {
    const uint32_t *ub = std::upper_bound(b->read_begin, b->read_begin + nw + 1, r);
    const size_t w = (size_t)(ub - b->read_begin) - 1;
    if (b->name_rank[r] >= b->read_begin[w + 1] - b->read_begin[w]) { e->err = "name_rank must be the dense per-window rank (< the window's reads)"; return LANCET_E_ARG; }
  }
  if (e->params.lr_mode) for (uint32_t r = 0; r < R; ++r) if (b->hp[r] > 2) { e->err = "hp must be 0, 1 or 2"; return LANCET_E_ARG; }   // Node_t::addHP indexes a 3-array (src/Node.cc:54-57)
  if (!pk) for (uint32_t r = 0; r < R; ++r) if ((b->label[r] != LANCET_TMR && b->label[r] != LANCET_NML) || (b->strand[r] != LANCET_FWD && b->strand[r] != LANCET_REV) || b->mate[r] > 2) {
    e->err = "label must be LANCET_TMR / LANCET_NML, strand LANCET_FWD / LANCET_REV, mate 0, 1 or 2"; return LANCET_E_ARG; }
  UP(e->d_chr, b->chr_id, sizeof(int32_t) * nw);
  UP(e->d_refstart, b->ref_start, sizeof(int32_t) * nw);
  UP(e->d_refoff, b->ref_off, sizeof(uint32_t) * (nw + 1));
  UP(e->d_refasc, b->ref_bases, nref);
  UP(e->d_readbegin, b->read_begin, sizeof(uint32_t) * (nw + 1));
  UP(e->d_seqoff, b->seq_off, sizeof(uint32_t) * (R + 1));
  UP(e->d_seq, b->seq, nbases);
  UP(e->d_qual, b->qual, nbases);
  UP(e->d_label, b->label, R); UP(e->d_strand, b->strand, R); UP(e->d_mate, b->mate, R); UP(e->d_mapped, b->mapped, R);
  UP(e->d_name, b->name_rank, sizeof(uint32_t) * R);
  if (e->params.lr_mode && R) { UP(e->d_bx, b->bx_rank, sizeof(uint32_t) * R); UP(e->d_hp, b->hp, R); }
  std::vector<uint32_t> bw(R + 1), gw(R + 1);
  uint32_t bo = 0, go = 0;
  for (uint32_t r = 0; r < R; ++r) { uint32_t len = b->seq_off[r + 1] - b->seq_off[r]; bw[r] = bo; gw[r] = go; bo += (len + 15) / 16; go += (len + 31) / 32; }
  bw[R] = bo; gw[R] = go;
  UP(e->d_bw, bw.data(), sizeof(uint32_t) * (R + 1));
  UP(e->d_gw, gw.data(), sizeof(uint32_t) * (R + 1));
  ENS(e->d_bases, sizeof(uint32_t) * (bo + 4));   /* (the occurrence-major insert reads up to two words past a k-mer) */ ENS(e->d_good, sizeof(uint32_t) * (go + 1)); ENS(e->d_rinfo, sizeof(uint32_t) * (R + 1));
  ENS(e->d_refcodes, nref + 1);
  DBG("prep launch");
  // ---- prep on the device: Graph_t::trim + packing, reference -> codes
  if (R) hipLaunchKernelGGL(prep_kernel, dim3((R + 3) / 4), dim3(256), 0, e->stream, (const lancet_params *)e->d_params.p, (int)R,
                            (const char *)e->d_seq.p, (const char *)e->d_qual.p, (const uint32_t *)e->d_seqoff.p, (const uint8_t *)e->d_label.p,
                            (const uint8_t *)e->d_strand.p, (const uint8_t *)e->d_mate.p, (const uint8_t *)e->d_mapped.p, (uint32_t *)e->d_rinfo.p,
                            (uint32_t *)e->d_bases.p, (const uint32_t *)e->d_bw.p, (uint32_t *)e->d_good.p, (const uint32_t *)e->d_gw.p);
  hipLaunchKernelGGL(ref_code_kernel, dim3((nref + 255) / 256), dim3(256), 0, e->stream, (const char *)e->d_refasc.p, (uint8_t *)e->d_refcodes.p, nref);
  HIPCHK(e, hipGetLastError());
  DBG("prep launched");
  db.chr_id = (LC_GLOBAL const int32_t *)e->d_chr.p; db.ref_start = (LC_GLOBAL const int32_t *)e->d_refstart.p;
  db.ref_off = (LC_GLOBAL const uint32_t *)e->d_refoff.p; db.ref_codes = (LC_GLOBAL const uint8_t *)e->d_refcodes.p; db.read_begin = (LC_GLOBAL const uint32_t *)e->d_readbegin.p;
  db.rinfo = (LC_GLOBAL const uint32_t *)e->d_rinfo.p; db.name_rank = (LC_GLOBAL const uint32_t *)e->d_name.p; db.base_woff = (LC_GLOBAL const uint32_t *)e->d_bw.p;
  db.good_woff = (LC_GLOBAL const uint32_t *)e->d_gw.p; db.bases = (LC_GLOBAL const uint32_t *)e->d_bases.p; db.good = (LC_GLOBAL const uint32_t *)e->d_good.p;
  db.bx_rank = e->params.lr_mode ? (LC_GLOBAL const uint32_t *)e->d_bx.p : nullptr; db.hp = e->params.lr_mode ? (LC_GLOBAL const uint8_t *)e->d_hp.p : nullptr;
  }
  tick("inputs (pack + stage + copy)");
  UP(e->d_batch, &db, sizeof(db));
  UP(e->d_caps, &e->caps, sizeof(e->caps));
  UP(e->d_caps2, &e->caps2, sizeof(e->caps2));
  // ---- work space: as many slots as fit (and are useful)
  size_t slot_bytes = lc_work_carve(nullptr, nullptr, e->caps);
  DBG("carve measured");
  int slots = e->max_slots;
  if (slots > nw) slots = nw;
  if ((size_t)slots * slot_bytes > e->mem_budget) slots = (int)std::max<size_t>(1, e->mem_budget / slot_bytes);   // as many as the budget holds
  // a GPU that is shared (or a second engine on it): fewer windows in flight rather than no run
  while (e->d_workmem.ensure((size_t)slots * slot_bytes)) {
    if (slots <= 64) { e->err = "hipMalloc failed (work space)"; return LANCET_E_OOM; }
    slots = slots * 3 / 4;
  }
  e->n_slots = slots;
  tick("work space of the window kernel", (size_t)slots * slot_bytes);
  std::vector<Work> works(slots);
  for (int s = 0; s < slots; ++s) lc_work_carve(&works[s], (char *)e->d_workmem.p + (size_t)s * slot_bytes, e->caps);
  UP(e->d_works, works.data(), sizeof(Work) * slots);
  // ---- outputs
  ENS(e->d_variants, sizeof(lancet_variant) * e->caps.var_cap);
  ENS(e->d_blob, e->caps.blob_cap);
  if (e->caps.lr_mode) { ENS(e->d_varlr, sizeof(lancet_variant_lr) * e->caps.var_cap); ENS(e->d_bxblob, sizeof(uint32_t) * e->caps.bx_cap); }
  ENS(e->d_counters, LC_COUNTER_BYTES);
  ENS(e->d_stats, sizeof(lancet_window_stats) * nw);
  ENS(e->d_evtlen, sizeof(uint32_t) * nw);
  ENS(e->d_phase, sizeof(unsigned long long) * 16 * nw);
  ENS(e->d_evt, sizeof(uint32_t) * (size_t)nw * (e->caps.evt_cap ? e->caps.evt_cap : 1));
  DevOut o;
  o.variants = (LC_GLOBAL lancet_variant *)e->d_variants.p; o.blob = (LC_GLOBAL char *)e->d_blob.p;
  o.n_variants = (LC_GLOBAL uint32_t *)e->d_counters.p; o.n_blob = (LC_GLOBAL uint32_t *)e->d_counters.p + 1; o.queue_head = (LC_GLOBAL uint32_t *)e->d_counters.p + 2;
  o.n_bx = (LC_GLOBAL uint32_t *)e->d_counters.p + 3; o.variants_lr = (LC_GLOBAL lancet_variant_lr *)e->d_varlr.p; o.bx_blob = (LC_GLOBAL uint32_t *)e->d_bxblob.p;
  o.stats = (LC_GLOBAL lancet_window_stats *)e->d_stats.p; o.evt_len = (LC_GLOBAL uint32_t *)e->d_evtlen.p; o.evt_out = (LC_GLOBAL uint32_t *)e->d_evt.p; o.phase = e->phase_times ? (LC_GLOBAL unsigned long long *)e->d_phase.p : nullptr; o.win_list = nullptr; o.n_list = 0;
  o.pre = nullptr; o.pre_pool = nullptr; o.n_ahead_used = nullptr; o.skip = nullptr; o.svc = nullptr;
  e->svc_cap = 0;
  e->pred.clear(); e->is_pred.assign(nw, 0);
  if (!e->debug_stop && !e->no_fat && !e->no_early_rerun) {
    for (int w = 0; w < nw; ++w) {
      const uint32_t r0 = b->read_begin[w], r1 = b->read_begin[w + 1];
      bool big = r1 - r0 + 1 > e->caps.reads_cap ||      // process_window's first tests: more reads, or more bases, than a tier-1 slot holds
                 (uint64_t)(b->seq_off[r1] - b->seq_off[r0]) + (b->ref_off[w + 1] - b->ref_off[w]) + 64u > (uint64_t)e->caps.occ_cap;
      // ... or more than the larger configuration of the LDS build kernel takes (131 040 bases with every read padded to 16, 1024 reads):
      // its graphs would all come from the general build on ONE wave (tens of ms: the tail of the launch); the several-wave kernel of the
      // re-run tier builds them in a few ms, next to everything else
      if (!big && e->prebuild && !e->no_large_build) {
        const uint64_t raw = (uint64_t)(b->seq_off[r1] - b->seq_off[r0]) + 8ull * (r1 - r0) + (b->ref_off[w + 1] - b->ref_off[w]) + 16u;    // (8: the mean padding)
        big = raw > (uint64_t)bl_large::LDS_BASES || r1 - r0 > bl_large::LDS_READS;
      }
      if (big) { e->pred.push_back((uint32_t)w); e->is_pred[w] = 1; }
    }
    if (e->pred.size() > 4096 || e->pred.size() * 4 > (size_t)nw) { e->pred.clear(); e->is_pred.assign(nw, 0); }     // (a batch of nothing but such windows: tier 1 keeps them)
  }
  if (!e->pred.empty()) {
    UP(e->d_skip, e->is_pred.data(), (size_t)nw);
    o.skip = (LC_GLOBAL const uint8_t *)e->d_skip.p;
    if (!e->stream2 && (hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e->evf0) != hipSuccess ||
                        hipEventCreate(&e->evf1) != hipSuccess || hipEventCreate(&e->ev_ready) != hipSuccess)) { e->err = "second stream"; return LANCET_E_HIP; }
  }
  if (e->prebuild) {
    const int cus = e->n_cus - e->svc_cus;
    // Two 512-lane workgroups fit a CU.  The grid stays one CU's worth short of the device (the service's workgroups, which work the same
    // queue first, counted in): the runtime carries out small copies, and copies to or from pageable memory, with copy KERNELS, and with
    // two engines taking turns on one GPU the read-back and the upload of one batch otherwise waited until the other batch's build kernel
    // -- which holds the registers and the LDS of every CU -- had left (38 ms per step; value_e2e 0.7 of value, now 0.9).
    const bool svc_helps = e->svc && e->n_svc_wgs > 0 && !e->debug_stop && e->svc_help;
    e->n_bslots = std::max(1, std::min(nw, cus * 2 - 2 - (svc_helps ? std::min(e->n_svc_wgs, cus) : 0)));
    if (e->build_slots_env) e->n_bslots = std::min(nw, e->build_slots_env);
    tick("outputs");
    ENS(e->d_pre, (size_t)nw * e->caps.pl.stride);
    tick("hand-off areas", (size_t)nw * e->caps.pl.stride);
    ENS(e->d_blscratch, (size_t)((e->build_cycle && e->gate) ? std::max(e->n_bslots, 2 * e->n_cus) : e->n_bslots) * bl_small::SCRATCH_BYTES);
    // Can any window be too big for the 512-lane configuration?  Trimming only shortens reads, so the untrimmed lengths bound the
    // LDS footprint (reads padded to 16 bases + the reference); when none can, the 1024-lane kernel is not launched at all.
    bool may_need_large = false;
    int n_need_large = 0;
    if (e->host_prep && e->exact_need_large >= 0) n_need_large = e->exact_need_large;
    else
    for (int w = 0; w < nw; ++w) {
      const uint32_t r0 = b->read_begin[w], r1 = b->read_begin[w + 1];
      const uint64_t raw = (uint64_t)(b->seq_off[r1] - b->seq_off[r0]) + 15ull * (r1 - r0) + (b->ref_off[w + 1] - b->ref_off[w]) + 16u;
      if (r1 - r0 > bl_small::LDS_READS || raw > (uint64_t)bl_small::LDS_BASES) ++n_need_large;
    }
    may_need_large = n_need_large > 0 || e->caps.pl.kw > 1;      // (wide hand-off areas: windows whose first k is above 31 go to the 1024-lane configuration too)
    e->n_bslots_large = (e->no_large_build || !may_need_large) ? 0 : std::min(nw, cus);
    if (e->n_bslots_large) { ENS(e->d_blscratch_large, (size_t)e->n_bslots_large * bl_large::SCRATCH_BYTES); ENS(e->d_biglist, sizeof(uint32_t) * (size_t)nw); }
    ENS(e->d_blphase, 16 * sizeof(unsigned long long));
    if (e->heavy_first) { ENS(e->d_order, (sizeof(uint32_t) + 1u) * (size_t)nw); o.win_list = (LC_GLOBAL const uint32_t *)e->d_order.p; o.n_list = (uint32_t)nw; }
    o.pre = (LC_GLOBAL const uint8_t *)e->d_pre.p;
    e->ahead_depth = e->ahead_depth_env >= 0 ? e->ahead_depth_env : 6;
    e->pool_cap = e->ahead_depth > 0 ? (uint32_t)std::max(64, nw / 4) : 0u;
    const bool use_svc = e->svc && e->n_svc_wgs > 0 && !e->debug_stop;
    if (use_svc) e->pool_cap += (uint32_t)std::max(64, nw / 8);
    if (use_svc) {
      if ((!e->stream3 && lc_service_stream(&e->stream3) != hipSuccess) || (!e->ev_svc && hipEventCreate(&e->ev_svc) != hipSuccess)) { e->err = "service stream"; return LANCET_E_HIP; }
      e->svc_cap = 2u * (uint32_t)nw + 1024u;
      const size_t off_req = 256, off_rdy = off_req + sizeof(SvcReq) * (size_t)e->svc_cap, off_cont = (off_rdy + 4u * (size_t)e->svc_cap + 63) & ~(size_t)63;
      ENS(e->d_svc, off_cont + sizeof(SvcCont) * (size_t)e->svc_cap);
      // Which configuration the service runs: the 1024-lane one (a whole CU per workgroup) when a good part of the batch is beyond the
      // 512-lane configuration's limits; else the 512-lane one, and the few deep windows build their later graphs themselves.
      e->svc_large = e->n_bslots_large > 0 && n_need_large * 8 > nw;
      ENS(e->d_svcscratch, (size_t)e->n_svc_wgs * (e->svc_large ? bl_large::SCRATCH_BYTES : bl_small::SCRATCH_BYTES));
      memset(&e->svc_host, 0, sizeof(SvcCtl));
      e->svc_host.cap = e->svc_cap; e->svc_host.large = e->svc_large ? 1u : 0u;
      e->svc_host.req = (LC_GLOBAL SvcReq *)((char *)e->d_svc.p + off_req); e->svc_host.rdy = (LC_GLOBAL uint32_t *)((char *)e->d_svc.p + off_rdy);
      e->svc_host.cont = (LC_GLOBAL SvcCont *)((char *)e->d_svc.p + off_cont);
      o.svc = (LC_GLOBAL SvcCtl *)e->d_svc.p;
    }
    if (e->pool_cap) { ENS(e->d_prepool, (size_t)e->pool_cap * e->caps.pl.stride); tick("pool of hand-off areas", (size_t)e->pool_cap * e->caps.pl.stride); o.pre_pool = (LC_GLOBAL const uint8_t *)e->d_prepool.p; o.n_ahead_used = (LC_GLOBAL uint32_t *)e->d_counters.p + 14; }
  }
  UP(e->d_out, &o, sizeof(o));
  DBG("sync");
  HIPCHK(e, hipStreamSynchronize(e->stream));
  tick("rest + stream sync");
  DBG("uploaded");
  e->uploaded = true;
  return LANCET_OK;
}

// The re-run tier for a list of windows: worst-case work space, window_fat.hip (or the one-wave kernel with LANCET_NO_FAT),
// queue head in counters[qword].  lc_prepare_rerun lays out the work space and the list (returns the number of slots),
// lc_launch_rerun_kernel launches on `st`; the caller brackets it with events.
// early: the launch runs next to build_kernel of the same batch -- the hand-off areas are being written (or still hold the previous
// batch's graphs), so it must not look at them: every graph by the general build.
static int lc_prepare_rerun(lancet_engine *e, const std::vector<uint32_t> &list, int qword, bool early) {
  size_t slot2 = lc_work_carve(nullptr, nullptr, e->caps2);
  int slots2 = (int)std::min<size_t>(list.size(), 128);
  while (slots2 > 1 && (size_t)slots2 * slot2 > ((size_t)16 << 30)) slots2 /= 2;
  if (e->d_workmem2.ensure((size_t)slots2 * slot2) || e->d_works2.ensure(sizeof(Work) * slots2) || e->d_winlist.ensure(sizeof(uint32_t) * list.size()) ||
      e->d_out2.ensure(sizeof(DevOut))) { e->err = "hipMalloc failed (tier 2)"; return LANCET_E_OOM; }
  std::vector<Work> works2(slots2);
  for (int s2 = 0; s2 < slots2; ++s2) lc_work_carve(&works2[s2], (char *)e->d_workmem2.p + (size_t)s2 * slot2, e->caps2);
  HIPCHK(e, lc_copy(e, e->d_works2.p, works2.data(), sizeof(Work) * slots2, hipMemcpyHostToDevice));
  HIPCHK(e, lc_copy(e, e->d_winlist.p, list.data(), sizeof(uint32_t) * list.size(), hipMemcpyHostToDevice));
  DevOut o2;
  HIPCHK(e, lc_copy(e, &o2, e->d_out.p, sizeof(o2), hipMemcpyDeviceToHost));
  o2.win_list = (LC_GLOBAL const uint32_t *)e->d_winlist.p; o2.n_list = (uint32_t)list.size(); o2.skip = nullptr; o2.svc = nullptr;
  if (early) { o2.pre = nullptr; o2.pre_pool = nullptr; o2.n_ahead_used = nullptr; }
  o2.queue_head = (LC_GLOBAL uint32_t *)e->d_counters.p + qword;
  HIPCHK(e, lc_copy(e, e->d_out2.p, &o2, sizeof(o2), hipMemcpyHostToDevice));
  return slots2;
}
static int lc_launch_rerun_kernel(lancet_engine *e, int slots2, hipStream_t st) {
  if (e->no_fat) {
    hipLaunchKernelGGL(window_kernel, dim3(slots2), dim3(LANCET_WG), 0, st, (const lancet_params *)e->d_params.p,
                       (const DevBatch *)e->d_batch.p, (const EngineCaps *)e->d_caps2.p, (Work *)e->d_works2.p, (DevOut *)e->d_out2.p);
    HIPCHK(e, hipGetLastError());
  } else {      // several waves per window (window_fat.hip)
    HIPCHK(e, (hipError_t)lc_launch_window_fat(slots2, st, (const lancet_params *)e->d_params.p, (const DevBatch *)e->d_batch.p,
                                               (const EngineCaps *)e->d_caps2.p, (Work *)e->d_works2.p, (DevOut *)e->d_out2.p));
  }
  return LANCET_OK;
}

// lancet_engine_run = lancet_engine_submit (launches the kernels of the uploaded batch on the engine's stream, returns at once) +
// lancet_engine_wait (waits for them, re-runs what overflowed the small work space, reads the results back).  Two engines on one
// GPU, submitted in turn, overlap the tail of one batch (a few multi-build windows) with the bulk of the next.
// Everything after the launch of the build service goes through lc_submit_body: whatever it returns, svc_done_kernel is enqueued
// behind it, so that the service's workgroups never outlive a submit that failed half way.
static int lc_submit_body(lancet_engine *e) {
  if (!e->pred.empty()) {          // the pile-ups start now, next to everything else (their work space was laid out by lancet_engine_submit)
    const int slots2 = e->slots2_pred;
    HIPCHK(e, hipEventRecord(e->ev_ready, e->stream));            // counters and statistics are cleared
    HIPCHK(e, hipStreamWaitEvent(e->stream2, e->ev_ready, 0));
    HIPCHK(e, hipEventRecord(e->evf0, e->stream2));
    int rc = lc_launch_rerun_kernel(e, slots2, e->stream2);
    if (rc) return rc;
    HIPCHK(e, hipEventRecord(e->evf1, e->stream2));
    e->fat_inflight = true;
  }
  if (e->prebuild) {
    HIPCHK(e, hipEventRecord(e->evb0, e->stream));
    // (behind the gate the first workgroups of this launch land on CUs the other batch's window kernel is still leaving: wherever its slots left their 80 KB, for as long
    //  as they stay -- so they stay for `build_cycle` windows only, and a second generation, dispatched as they leave, is the persistent one; profiles/r6h_gate.txt)
    const int cyc = e->cycle_now ? e->build_cycle : 0;
    const int bl_nslots = std::max(e->n_bslots, 2 * e->n_cus), bl_grid = cyc ? 2 * e->n_bslots : e->n_bslots;
    hipLaunchKernelGGL(build_kernel, dim3(bl_grid), dim3(bl_small::WG), 0, e->stream, (const lancet_params *)e->d_params.p, (const DevBatch *)e->d_batch.p,
                       (const EngineCaps *)e->d_caps.p, (uint8_t *)e->d_pre.p, (uint8_t *)e->d_blscratch.p, (uint32_t *)e->d_counters.p + 8,
                       (unsigned long long *)(e->phase_times ? e->d_blphase.p : nullptr), (uint8_t *)(e->pool_cap ? e->d_prepool.p : nullptr), e->pool_cap, e->ahead_depth,
                       (uint32_t *)(e->n_bslots_large ? e->d_biglist.p : nullptr), (e->svc_running && e->svc_help && !e->svc_large) ? 1 : 0,
                       cyc, (uint32_t *)e->d_counters.p + 96, bl_nslots, e->n_bslots);
    HIPCHK(e, hipGetLastError());
    if (e->dbg) { HIPCHK(e, hipStreamSynchronize(e->stream)); DBG("build_kernel done"); }
    if (e->n_bslots_large) {
      // (grid sized from what the previous batch left on the list -- batches of one scan are alike; the kernel works any list
      //  off whatever its grid, and a full grid of idle 1024-lane workgroups costs 0.4 ms per batch)
      const int glarge = e->n_biglist < 0 ? e->n_bslots_large : std::max(8, std::min(e->n_bslots_large, e->n_biglist));
      hipLaunchKernelGGL(build_kernel_large, dim3(glarge), dim3(bl_large::WG), 0, e->stream, (const lancet_params *)e->d_params.p, (const DevBatch *)e->d_batch.p,
                         (const EngineCaps *)e->d_caps.p, (uint8_t *)e->d_pre.p, (uint8_t *)e->d_blscratch_large.p, (uint32_t *)e->d_counters.p + 8,
                         (unsigned long long *)(e->phase_times ? e->d_blphase.p : nullptr), (uint8_t *)(e->pool_cap ? e->d_prepool.p : nullptr), e->pool_cap, e->ahead_depth, (uint32_t *)e->d_biglist.p);
      HIPCHK(e, hipGetLastError());
      if (e->dbg) { HIPCHK(e, hipStreamSynchronize(e->stream)); DBG("build_kernel_large done"); }
    }
    if (e->heavy_first) {
      uint8_t *cls = (uint8_t *)e->d_order.p + sizeof(uint32_t) * (size_t)e->n_windows;
      uint32_t *cnt = (uint32_t *)e->d_counters.p + 32;
      hipLaunchKernelGGL(order_class_kernel, dim3((e->n_windows + 255) / 256), dim3(256), 0, e->stream, (const uint8_t *)e->d_pre.p, e->caps.pl.stride, e->caps.pl.chdr, e->n_windows, cls, cnt, e->order_mode);
      hipLaunchKernelGGL(order_place_kernel, dim3((e->n_windows + 255) / 256), dim3(256), 0, e->stream, e->n_windows, (const uint8_t *)cls, (const uint32_t *)cnt, cnt + ORD_CLASSES, (uint32_t *)e->d_order.p);
      HIPCHK(e, hipGetLastError());
    }
    HIPCHK(e, hipEventRecord(e->evb1, e->stream));
  }
  HIPCHK(e, hipEventRecord(e->ev_a_done, e->stream));
  HIPCHK(e, hipEventRecord(e->ev0, e->stream));
  hipLaunchKernelGGL(window_kernel, dim3(e->n_slots), dim3(LANCET_WG), 0, e->stream, (const lancet_params *)e->d_params.p,
                     (const DevBatch *)e->d_batch.p, (const EngineCaps *)e->d_caps.p, (Work *)e->d_works.p, (DevOut *)e->d_out.p);
  HIPCHK(e, hipGetLastError());
  if (e->dbg && !e->svc_running) { HIPCHK(e, hipStreamSynchronize(e->stream)); DBG("window_kernel done"); }
  HIPCHK(e, hipEventRecord(e->ev1, e->stream));
  return LANCET_OK;
}

// Two engines on one device, submitted in turn: the kernels of `e` start only when `prev`'s kernels are through.  Both of a batch's
// kernels are persistent and sized for the whole device (LDS of every CU); side by side with the other batch's they only slow each other
// down (measured: 61-68 ms per step overlapped, 57 one batch at a time, 54-55 back to back like this).  What the second engine buys is
// that upload, launch and read-back of one batch run while the other's kernels do.
// `prev` must have been submitted (its lancet_engine_submit[_after] has returned) by the time of this call -- its done-event is recorded
// by then; a caller that submits from several threads orders them itself (lancet_main.cc submits on one thread and waits on others).
int lancet_engine_submit_after(lancet_engine *e, lancet_engine *prev) {
  if (!e) return LANCET_E_ARG;
  const bool gate_ok = e->gate && prev && !prev->params.lr_mode && !e->params.lr_mode;
  e->wait_ev = (prev && prev != e && prev->device == e->device) ? ((e->stagger || gate_ok) ? prev->ev_a_done : prev->ev_done) : nullptr;      // (an event never recorded, or long since reached, does not hold anything up)
  // (not with --linked-reads: there the window kernel's launch is three times the build kernel's and ends on many busy slots, not on a few waiting windows --
  //  the next batch's build workgroups trickle in between them for milliseconds: config 5 232 k instead of 243 k windows/s, profiles/r6h_gate.txt)
  if (e->wait_ev && e->gate && prev->submitted && prev->n_windows > 0 && !prev->params.lr_mode && !e->params.lr_mode) { e->gate_head = (const uint32_t *)prev->d_counters.p + 2; e->gate_n = (uint32_t)std::max(1, prev->n_windows - prev->n_windows / 4); e->gate_prev_done = prev->ev_done; }
  const int rc = lancet_engine_submit(e);
  e->wait_ev = nullptr; e->gate_head = nullptr; e->gate_prev_done = nullptr;
  return rc;
}

int lancet_engine_submit(lancet_engine *e) {
  if (!e) return LANCET_E_ARG;
  if (!e->uploaded) { e->err = "run before upload"; return LANCET_E_STATE; }
  if (e->submitted) { e->err = "submit while a batch is in flight"; return LANCET_E_STATE; }
  HIPCHK(e, hipSetDevice(e->device));
  e->ran = false;
  e->variants.clear(); e->blob.clear(); e->stats.clear(); e->evt_len.clear(); e->evt.clear(); e->variants_lr.clear(); e->bx_blob.clear();
  if (e->n_windows == 0) { e->submitted = true; return LANCET_OK; }
  // The copies that lay out the early re-run tier synchronise this engine's stream: they go first, so that the host thread is not held
  // until the other engine's kernels are through (the wait below is only enqueued, on the device).
  e->slots2_pred = 0;
  if (!e->pred.empty()) { e->slots2_pred = lc_prepare_rerun(e, e->pred, 4, true); if (e->slots2_pred < 0) return e->slots2_pred; }
  if (e->wait_ev) HIPCHK(e, hipStreamWaitEvent(e->stream, e->wait_ev, 0));
  e->cycle_now = e->wait_ev && e->gate_head && e->build_cycle > 0;
  if (e->wait_ev && e->gate_head) { hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, e->stream, e->gate_head, e->gate_n); HIPCHK(e, hipGetLastError()); }
  HIPCHK(e, hipMemsetAsync(e->d_counters.p, 0, LC_COUNTER_BYTES, e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_stats.p, 0, sizeof(lancet_window_stats) * e->n_windows, e->stream));
  if (e->prebuild) HIPCHK(e, hipMemsetAsync(e->d_blphase.p, 0, 16 * sizeof(unsigned long long), e->stream));      // (before the service starts: its workgroups add to it too)
  e->ms_build = 0; e->n_prebuilt = 0;
  e->fat_inflight = false; e->ms_fat = 0;
  e->svc_running = false;
  if (e->svc_cap) {                // the build service takes its place on the GPU before the batch's kernels
    HIPCHK(e, hipMemsetAsync(e->d_svc.p, 0, 256 + (sizeof(SvcReq) + 4u) * (size_t)e->svc_cap, e->stream));
    hipLaunchKernelGGL(svc_init_kernel, dim3(1), dim3(1), 0, e->stream, (SvcCtl *)e->d_svc.p, e->svc_host);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipEventRecord(e->ev_svc, e->stream));
    HIPCHK(e, hipStreamWaitEvent(e->stream3, e->ev_svc, 0));
    if (e->gate_prev_done) HIPCHK(e, hipStreamWaitEvent(e->stream3, e->gate_prev_done, 0));      // (the service's workgroups stay for two kernels: not into a CU half full of the other batch's slots)
    if (e->svc_large)
      hipLaunchKernelGGL(svc_kernel_large, dim3(e->n_svc_wgs), dim3(bl_large::WG), 0, e->stream3, (const lancet_params *)e->d_params.p, (const DevBatch *)e->d_batch.p,
                         (const EngineCaps *)e->d_caps.p, (uint8_t *)e->d_pre.p, (uint8_t *)e->d_svcscratch.p, (uint32_t *)e->d_counters.p + 8,
                         (uint8_t *)e->d_prepool.p, e->pool_cap, e->svc_depth, (SvcCtl *)e->d_svc.p, (const uint32_t *)e->d_counters.p + 2);
    else
    hipLaunchKernelGGL(svc_kernel, dim3(e->n_svc_wgs), dim3(bl_small::WG), 0, e->stream3, (const lancet_params *)e->d_params.p, (const DevBatch *)e->d_batch.p,
                       (const EngineCaps *)e->d_caps.p, (uint8_t *)e->d_pre.p, (uint8_t *)e->d_svcscratch.p, (uint32_t *)e->d_counters.p + 8,
                       (uint8_t *)e->d_prepool.p, e->pool_cap, e->svc_depth, (SvcCtl *)e->d_svc.p, (const uint32_t *)e->d_counters.p + 2,
                       (uint32_t *)(e->n_bslots_large ? e->d_biglist.p : nullptr), (e->svc_help && e->prebuild) ? e->ahead_depth : -1, (unsigned long long *)(e->phase_times ? e->d_blphase.p : nullptr));
    HIPCHK(e, hipGetLastError());
    e->svc_running = true;
  }
  const int rc = lc_submit_body(e);
  if (e->svc_running) {
    hipLaunchKernelGGL(svc_done_kernel, dim3(1), dim3(1), 0, e->stream, (SvcCtl *)e->d_svc.p);
    if (hipGetLastError() != hipSuccess || rc) {               // make sure the service leaves before the error is reported
      (void)hipStreamSynchronize(e->stream);
      const unsigned one = 1;
      (void)hipMemcpy((char *)e->d_svc.p + offsetof(SvcCtl, done), &one, sizeof(one), hipMemcpyHostToDevice);
      (void)hipStreamSynchronize(e->stream3);
      e->svc_running = false;
      if (!rc) { e->err = "svc_done_kernel launch"; return LANCET_E_HIP; }
    }
  }
  if (rc) { (void)hipStreamSynchronize(e->stream); if (e->stream2) (void)hipStreamSynchronize(e->stream2); e->fat_inflight = false; return rc; }
  HIPCHK(e, hipEventRecord(e->ev_done, e->stream));             // the batch's kernels (build, window, the service's leave signal) are behind this
  e->submitted = true;
  return LANCET_OK;
}

int lancet_engine_wait(lancet_engine *e) {
  if (!e) return LANCET_E_ARG;
  if (!e->submitted) { e->err = "wait without submit"; return LANCET_E_STATE; }
  HIPCHK(e, hipSetDevice(e->device));
  e->submitted = false;
  if (e->n_windows == 0) { e->ran = true; return LANCET_OK; }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (e->svc_running) {
    HIPCHK(e, hipStreamSynchronize(e->stream3));
    e->svc_running = false;
    SvcCtl ct;
    HIPCHK(e, lc_copy(e, &ct, e->d_svc.p, sizeof(ct), hipMemcpyDeviceToHost));
    e->svc_counts[0] = std::min(ct.req_alloc, ct.cap); e->svc_counts[1] = ct.n_built; e->svc_counts[2] = ct.n_failed; e->svc_counts[3] = ct.n_stolen; e->svc_counts[4] = ct.n_gaveup;
  }
  HIPCHK(e, hipEventElapsedTime(&e->ms_window, e->ev0, e->ev1));
  if (e->fat_inflight) {
    HIPCHK(e, hipStreamSynchronize(e->stream2));
    HIPCHK(e, hipEventElapsedTime(&e->ms_fat, e->evf0, e->evf1));
    e->fat_inflight = false;
  }
  if (e->prebuild) {
    HIPCHK(e, hipEventElapsedTime(&e->ms_build, e->evb0, e->evb1));
    uint32_t bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(e, lc_copy(e, bq, (uint32_t *)e->d_counters.p + 8, sizeof(bq), hipMemcpyDeviceToHost));
    e->n_prebuilt = (int)bq[1]; e->n_ahead_built = (int)bq[3]; e->n_ahead_used = (int)bq[6]; e->n_biglist = (int)bq[4];
    if (e->phase_times) HIPCHK(e, lc_copy(e, e->blphase, e->d_blphase.p, sizeof(e->blphase), hipMemcpyDeviceToHost)); else memset(e->blphase, 0, sizeof(e->blphase));
  }
  if (e->ms_fat > e->ms_window + e->ms_build) e->ms_window = e->ms_fat - e->ms_build;    // both started together: the window kernels' time is the longer of the two
  e->ms_all = e->ms_window + e->ms_build;
  e->ms_kernel = e->ms_all;
  // ---- tier 2: windows that did not fit the small work space are re-run with the worst-case one
  e->stats.resize(e->n_windows);
  HIPCHK(e, lc_copy(e, e->stats.data(), e->d_stats.p, sizeof(lancet_window_stats) * e->n_windows, hipMemcpyDeviceToHost));
  uint32_t nv_tier1 = 0;
  HIPCHK(e, lc_copy(e, &nv_tier1, e->d_counters.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::vector<uint32_t> rerun;
  std::vector<char> ok1(e->n_windows, 1);
  for (int w = 0; w < e->n_windows; ++w) if (e->stats[w].status == LANCET_W_OVERFLOW && !e->is_pred[w]) { rerun.push_back((uint32_t)w); ok1[w] = 0; }
  e->n_rerun = (int)(rerun.size() + e->pred.size());
  if (!rerun.empty() && !e->debug_stop) {
    int slots2 = lc_prepare_rerun(e, rerun, 2, false);
    if (slots2 < 0) return slots2;
    HIPCHK(e, hipMemsetAsync((uint32_t *)e->d_counters.p + 2, 0, sizeof(uint32_t), e->stream));      // queue head
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    { int rc = lc_launch_rerun_kernel(e, slots2, e->stream); if (rc) return rc; }
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    float ms2 = 0;
    HIPCHK(e, hipEventElapsedTime(&ms2, e->ev0, e->ev1));
    e->ms_all += ms2; e->ms_kernel += ms2; e->ms_window += ms2;
  }
  // ---- read back
  uint32_t counters[4];
  HIPCHK(e, lc_copy(e, counters, e->d_counters.p, sizeof(counters), hipMemcpyDeviceToHost));
  HIPCHK(e, lc_copy(e, e->stats.data(), e->d_stats.p, sizeof(lancet_window_stats) * e->n_windows, hipMemcpyDeviceToHost));
  bool global_overflow = counters[0] > e->caps.var_cap || counters[1] > e->caps.blob_cap || (e->caps.lr_mode && counters[3] > e->caps.bx_cap);
  uint32_t nv = std::min(counters[0], e->caps.var_cap), nb = std::min(counters[1], e->caps.blob_cap);
  std::vector<lancet_variant> raw(nv);
  std::vector<char> rawblob(nb);
  if (nv) HIPCHK(e, lc_copy(e, raw.data(), e->d_variants.p, sizeof(lancet_variant) * nv, hipMemcpyDeviceToHost));
  if (nb) HIPCHK(e, lc_copy(e, rawblob.data(), e->d_blob.p, nb, hipMemcpyDeviceToHost));
  std::vector<lancet_variant_lr> rawlr; std::vector<uint32_t> rawbx;
  if (e->caps.lr_mode) {
    uint32_t nx = std::min(counters[3], e->caps.bx_cap);
    rawlr.resize(nv); rawbx.resize(nx);
    if (nv) HIPCHK(e, lc_copy(e, rawlr.data(), e->d_varlr.p, sizeof(lancet_variant_lr) * nv, hipMemcpyDeviceToHost));
    if (nx) HIPCHK(e, lc_copy(e, rawbx.data(), e->d_bxblob.p, sizeof(uint32_t) * nx, hipMemcpyDeviceToHost));
  }
  e->phase.clear();                                   // (profiling data: fetched when lancet_engine_phase_times asks for it)
  if (e->caps.evt_cap) {
    e->evt_len.resize(e->n_windows); e->evt.resize((size_t)e->n_windows * e->caps.evt_cap);
    HIPCHK(e, lc_copy(e, e->evt_len.data(), e->d_evtlen.p, sizeof(uint32_t) * e->n_windows, hipMemcpyDeviceToHost));
    HIPCHK(e, lc_copy(e, e->evt.data(), e->d_evt.p, sizeof(uint32_t) * e->evt.size(), hipMemcpyDeviceToHost));
  }
  // variants of windows that overflowed are dropped; order by (window, emission order)
  std::vector<uint32_t> idx;
  for (uint32_t i = 0; i < nv; ++i) {
    const lancet_variant &v = raw[i];
    if (v.window < 0 || v.window >= e->n_windows) continue;
    if (e->stats[v.window].status < 0) continue;
    if (i < nv_tier1 && !ok1[v.window]) continue;           // partial output of a window that was re-run in tier 2
    if ((size_t)v.str_off + v.str_len > nb) continue;
    idx.push_back(i);
  }
  std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
    if (raw[a].window != raw[b].window) return raw[a].window < raw[b].window;
    return raw[a].seq_in_window < raw[b].seq_in_window;
  });
  for (uint32_t i : idx) {
    lancet_variant v = raw[i];
    uint32_t o = (uint32_t)e->blob.size();
    e->blob.insert(e->blob.end(), rawblob.begin() + v.ref_off, rawblob.begin() + v.ref_off + v.ref_len);
    e->blob.insert(e->blob.end(), rawblob.begin() + v.alt_off, rawblob.begin() + v.alt_off + v.alt_len);
    e->blob.insert(e->blob.end(), rawblob.begin() + v.str_off, rawblob.begin() + v.str_off + v.str_len);
    v.ref_off = o; v.alt_off = o + v.ref_len; v.str_off = o + v.ref_len + v.alt_len;
    e->variants.push_back(v);
    if (e->caps.lr_mode) {
      lancet_variant_lr l = rawlr[i];
      for (int q = 0; q < 4; ++q) {
        if ((size_t)l.bx_off[q] + l.bx_len[q] > rawbx.size()) { l.bx_len[q] = 0; l.bx_off[q] = 0; }
        uint32_t no = (uint32_t)e->bx_blob.size();
        e->bx_blob.insert(e->bx_blob.end(), rawbx.begin() + l.bx_off[q], rawbx.begin() + l.bx_off[q] + l.bx_len[q]);
        l.bx_off[q] = no;
      }
      e->variants_lr.push_back(l);
    }
  }
  if (global_overflow) {   // some window could not write its records: mark every window that lost some
    std::vector<int> got(e->n_windows, 0);
    for (auto &v : e->variants) ++got[v.window];
    for (int w = 0; w < e->n_windows; ++w) if (e->stats[w].status >= 0 && got[w] != e->stats[w].n_variants) e->stats[w].status = LANCET_W_OVERFLOW;
    if (e->caps.lr_mode) {
      std::vector<lancet_variant_lr> keep;
      for (size_t i = 0; i < e->variants.size(); ++i) if (e->stats[e->variants[i].window].status >= 0) keep.push_back(e->variants_lr[i]);
      e->variants_lr.swap(keep);
    }
    e->variants.erase(std::remove_if(e->variants.begin(), e->variants.end(), [&](const lancet_variant &v) { return e->stats[v.window].status < 0; }), e->variants.end());
  }
  e->ran = true;
  return LANCET_OK;
}

int lancet_engine_run(lancet_engine *e) {
  int rc = lancet_engine_submit(e);
  if (rc) return rc;
  return lancet_engine_wait(e);
}

int lancet_engine_process(lancet_engine *e, const lancet_window_batch *b) {
  int rc = lancet_engine_upload(e, b);
  if (rc) return rc;
  return lancet_engine_run(e);
}

int lancet_engine_results(lancet_engine *e, const lancet_variant **variants, uint32_t *n_variants, const char **blob, uint32_t *blob_len,
                          const lancet_window_stats **stats) {
  if (!e) return LANCET_E_ARG;
  if (!e->ran) { e->err = "results before run"; return LANCET_E_STATE; }
  if (variants) *variants = e->variants.data();
  if (n_variants) *n_variants = (uint32_t)e->variants.size();
  if (blob) *blob = e->blob.data();
  if (blob_len) *blob_len = (uint32_t)e->blob.size();
  if (stats) *stats = e->stats.data();
  return LANCET_OK;
}

int lancet_engine_results_lr(lancet_engine *e, const lancet_variant_lr **lr, const uint32_t **bx_blob, uint32_t *bx_blob_len) {
  if (!e) return LANCET_E_ARG;
  if (!e->ran) { e->err = "results before run"; return LANCET_E_STATE; }
  if (lr) *lr = e->params.lr_mode ? e->variants_lr.data() : nullptr;
  if (bx_blob) *bx_blob = e->bx_blob.data();
  if (bx_blob_len) *bx_blob_len = (uint32_t)e->bx_blob.size();
  return LANCET_OK;
}

int lancet_engine_last_timing(lancet_engine *e, float out[2]) {
  if (!e || !e->ran) return LANCET_E_STATE;
  out[0] = e->ms_all; out[1] = e->ms_kernel;
  return LANCET_OK;
}

// debug: trace events of the last run (see lancet_amd/trace.py); *words_per_window = 0 when tracing is off
int lancet_engine_trace(lancet_engine *e, const uint32_t **evt_len, const uint32_t **evt, uint32_t *words_per_window) {
  if (!e || !e->ran) return LANCET_E_STATE;
  *evt_len = e->evt_len.data(); *evt = e->evt.data(); *words_per_window = e->caps.evt_cap;
  return LANCET_OK;
}

// test hook: runs the device alignment on (S, T) (ACGT strings, |S| <= LC_MAXW); writes the aligned strings.
// mode 0: band first, full matrix when the band is not certified (what the window kernel does); 1: full matrix only; 2: band only
// (LANCET_E_STATE when the band could not be certified)
int lancet_debug_align_mode(lancet_engine *e, const char *S, const char *T, char *S_aln, char *T_aln, int cap, int mode);
int lancet_debug_align(lancet_engine *e, const char *S, const char *T, char *S_aln, char *T_aln, int cap) { return lancet_debug_align_mode(e, S, T, S_aln, T_aln, cap, 0); }
int lancet_debug_align_mode(lancet_engine *e, const char *S, const char *T, char *S_aln, char *T_aln, int cap, int mode) {
  if (!e || !S || !T) return LANCET_E_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  int n = (int)strlen(S), m = (int)strlen(T);
  if (n < 1 || m < 1 || n > LC_MAXW) return LANCET_E_ARG;
  EngineCaps caps; memset(&caps, 0, sizeof(caps));
  caps.reads_cap = 4; caps.occ_cap = 64; caps.node_cap = 16; caps.table_cap = 32; caps.bucket_cap = 32; caps.special_cap = 4; caps.surv_cap = 4;
  caps.seq_cap = 64; caps.queue_cap = 4; caps.path_cap = (uint32_t)m + 8; caps.max_k = 16; caps.qv_cap = 64;
  caps.max_w = std::max<uint32_t>(LC_MAXW_DEFAULT, ((uint32_t)n + 63u) & ~63u);
  size_t bytes = lc_work_carve(nullptr, nullptr, caps);
  DevBuf mem, dcaps, dwork, ds, dt, dl;
  std::vector<uint8_t> sc(n), tc(m);
  auto code = [](char b) -> uint8_t { switch (b) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; } return 4; };
  for (int i = 0; i < n; ++i) sc[i] = code(S[i]);
  for (int i = 0; i < m; ++i) tc[i] = code(T[i]);
  if (mem.ensure(bytes) || dcaps.ensure(sizeof(caps)) || dwork.ensure(sizeof(Work)) || ds.ensure(n) || dt.ensure(m) || dl.ensure(4)) return LANCET_E_OOM;
  Work w; lc_work_carve(&w, (char *)mem.p, caps);
  HIPCHK(e, lc_copy(e, dcaps.p, &caps, sizeof(caps), hipMemcpyHostToDevice));
  HIPCHK(e, lc_copy(e, dwork.p, &w, sizeof(w), hipMemcpyHostToDevice));
  HIPCHK(e, lc_copy(e, ds.p, sc.data(), n, hipMemcpyHostToDevice));
  HIPCHK(e, lc_copy(e, dt.p, tc.data(), m, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(align_test_kernel, dim3(1), dim3(LANCET_WG), 0, e->stream, (const EngineCaps *)dcaps.p, (Work *)dwork.p, (const uint8_t *)ds.p, n,
                     (const uint8_t *)dt.p, m, (int *)dl.p, mode);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  int L = 0;
  HIPCHK(e, lc_copy(e, &L, dl.p, 4, hipMemcpyDeviceToHost));
  int rc = LANCET_OK;
  if (L == -2) rc = LANCET_E_STATE;
  else if (L < 0 || L + 1 > cap) rc = LANCET_E_UNSUPPORTED;
  else {
    const int acap = (int)caps.max_w + (int)caps.path_cap + 2;
    HIPCHK(e, lc_copy(e, S_aln, w.aln, L, hipMemcpyDeviceToHost));
    HIPCHK(e, lc_copy(e, T_aln, w.aln + acap, L, hipMemcpyDeviceToHost));
    S_aln[L] = 0; T_aln[L] = 0;
  }
  mem.release(); dcaps.release(); dwork.release(); ds.release(); dt.release(); dl.release();
  return rc;
}

// profiling aid: per-window, per-phase time in 10 ns ticks (16 phases per window, see PHASE() in kernels.h)
int lancet_engine_phase_times(lancet_engine *e, const unsigned long long **ticks) {
  if (!e || !e->ran) return LANCET_E_STATE;
  if (e->phase.empty() && e->n_windows > 0) {
    if (e->submitted) { e->err = "phase times while a batch is in flight"; return LANCET_E_STATE; }
    HIPCHK(e, hipSetDevice(e->device));
    e->phase.resize((size_t)e->n_windows * 16);
    if (e->phase_times) HIPCHK(e, lc_copy(e, e->phase.data(), e->d_phase.p, sizeof(unsigned long long) * e->phase.size(), hipMemcpyDeviceToHost));      // (else zeros: LANCET_PHASE_TIMES was not set when the engine was made)
  }
  *ticks = e->phase.data();
  return LANCET_OK;
}

// HIP-event durations (ms) of the kernels of the last run, in the order of lancet_engine_kernel_name(0..); returns how many
static const char *const lc_kernel_names[] = {"build_kernel", "window_kernel"};
const char *lancet_engine_kernel_name(int i) { return (i >= 0 && i < (int)(sizeof(lc_kernel_names) / sizeof(lc_kernel_names[0]))) ? lc_kernel_names[i] : nullptr; }
int lancet_engine_kernel_times(lancet_engine *e, float *ms, int cap) {
  if (!e || !e->ran) return LANCET_E_STATE;
  if (cap < 2 || !ms) return LANCET_E_ARG;
  ms[0] = e->ms_build; ms[1] = e->ms_window;
  return 2;
}
// profiling aid: workgroup-seconds of the LDS build kernel per phase (16 values, 10 ns ticks; BLP() markers of build_lds.h)
int lancet_engine_build_phase_times(lancet_engine *e, const unsigned long long **ticks) {
  if (!e || !e->ran) return LANCET_E_STATE;
  *ticks = e->blphase;
  return LANCET_OK;
}
// windows of the last run whose first graph came from the LDS build kernel
int lancet_engine_prebuilt_count(lancet_engine *e) { return e ? e->n_prebuilt : -1; }

// test/tuning hook: the hand-off headers of the last run (status | why << 8 | table order and components came along << 16, K, heavy, N, nsurv, numcomp, ncand, next per window)
int lancet_debug_pre_headers(lancet_engine *e, uint32_t *out) {
  if (!e || !e->uploaded || !e->d_pre.p) return LANCET_E_STATE;
  std::vector<PreHdr> h(1);
  for (int w = 0; w < e->n_windows; ++w) {
    if (hipMemcpy(h.data(), (const uint8_t *)e->d_pre.p + (size_t)w * e->caps.pl.stride + PRE_OFF_HDR, sizeof(PreHdr), hipMemcpyDeviceToHost) != hipSuccess) return LANCET_E_STATE;
    out[8 * w] = h[0].status | (h[0].why << 8) | (h[0].have_order == 1u ? 1u << 16 : 0u); out[8 * w + 1] = h[0].K; out[8 * w + 2] = h[0].heavy; out[8 * w + 3] = h[0].N;
    out[8 * w + 4] = h[0].nsurv; out[8 * w + 5] = h[0].numcomp; out[8 * w + 6] = h[0].ncand; out[8 * w + 7] = h[0].next;
  }
  return LANCET_OK;
}

// test / tuning hook: per window of the last run, what the build kernel's first compress left (done, nodes in the table, k-mer nodes merged away, words of sequence)
int lancet_debug_pre_cmp(lancet_engine *e, uint32_t *out) {
  if (!e || !e->uploaded || !e->d_pre.p) return LANCET_E_STATE;
  PreCmp h;
  for (int w = 0; w < e->n_windows; ++w) {
    if (hipMemcpy(&h, (const uint8_t *)e->d_pre.p + (size_t)w * e->caps.pl.stride + e->caps.pl.chdr, sizeof(PreCmp), hipMemcpyDeviceToHost) != hipSuccess) return LANCET_E_STATE;
    out[4 * w] = h.done; out[4 * w + 1] = h.m_live; out[4 * w + 2] = h.dead; out[4 * w + 3] = h.seqn;
  }
  return LANCET_OK;
}

// graphs the build kernel built ahead (a later k of a window whose first k was going to be rejected) / how many of them the
// window kernel took
int lancet_engine_ahead_counts(lancet_engine *e, int32_t *built, int32_t *used) {
  if (!e) return LANCET_E_ARG;
  if (built) *built = e->n_ahead_built;
  if (used) *used = e->n_ahead_used;
  return LANCET_OK;
}

// build service of the last run: requests posted by the window kernel, served (graph built in LDS), not buildable there, taken back
int lancet_engine_svc_counts(lancet_engine *e, uint32_t out[4]) {
  if (!e || !out) return LANCET_E_ARG;
  if (e->dbg && e->svc_counts[4]) fprintf(stderr, "[lancet] %u service workgroups gave up waiting\n", e->svc_counts[4]);
  for (int i = 0; i < 4; ++i) out[i] = e->svc_counts[i];
  return LANCET_OK;
}

// number of windows of the last run that needed the worst-case work space (tier 2)
int lancet_engine_rerun_count(lancet_engine *e) { return e ? e->n_rerun : -1; }

// introspection used by bench.py: slots in flight and bytes of work space per slot
int lancet_engine_geometry(lancet_engine *e, int32_t *n_slots, uint64_t *slot_bytes) {
  if (!e || !e->uploaded) return LANCET_E_STATE;
  *n_slots = e->n_slots; *slot_bytes = lc_work_carve(nullptr, nullptr, e->caps);
  return LANCET_OK;
}

}  // extern "C"
