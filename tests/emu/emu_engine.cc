// emu_engine.cc -- TEST INFRASTRUCTURE ONLY.
//
// Compiles lancet_amd/csrc/kernels.h with LANCET_WAVE_EMU (lanes of a phase run one after the other on the
// host) so that the kernel logic can be exercised against the oracle on a machine without a GPU.  This is a
// debugging aid for the authoring container; it is not a backend: the product library (liblancet_engine.so)
// contains only the HIP build and refuses to run without a device.
#define LANCET_WAVE_EMU 1
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../lancet_amd/csrc/kernels.h"
#include "../../lancet_amd/csrc/build_lds.h"
#include "../../lancet_amd/csrc/host_common.h"

struct EmuResult {
  std::vector<lancet_variant> variants;
  std::vector<lancet_variant_lr> lr;
  std::vector<uint32_t> bx_blob;
  std::vector<char> blob;
  std::vector<lancet_window_stats> stats;
  std::vector<uint32_t> evt_len, evt;
  uint32_t evt_cap;
  uint32_t n_variants, n_blob;
  uint32_t n_prebuilt, n_ahead_built, n_ahead_used, n_biglist;
  uint32_t n_svc_built, n_svc_stolen, n_svc_posted;
  uint32_t n_cmp_done = 0;
};

extern "C" void *lancet_emu_run(const lancet_params *P, const lancet_window_batch *b, uint32_t evt_cap) {
  EngineCaps C = getenv("LANCET_EMU_TIER1") ? lc_caps_for_batch(b, P, evt_cap, 16384, 1)      // (the engine's tier-1 work space: to see which limit a window hits there)
                                            : lc_caps_for_batch(b, P, evt_cap, 65536);
  C.pl = lc_pre_layout_for_batch(b, 1, (size_t)1 << 40, getenv("LANCET_PRE_WIDE") ? atoi(getenv("LANCET_PRE_WIDE")) : -1, P->lr_mode != 0);      // (engine.hip lc_upload)
  C.wide_ids = LC_WIDE_IDS;                        // (libemu_fat.so: the re-run tier's 64-bit csr words, as engine.hip lays its tier 2 out)
  if (const char *ts = getenv("LANCET_TABLE_START")) C.table_start = lc_pow2_ge((uint32_t)atoi(ts));
  if (const char *st = getenv("LANCET_STOP_PHASE")) C.debug_stop = (uint32_t)atoi(st);
  const uint32_t R = b->read_begin[b->n_windows];
  // ---- device batch (host memory here)
  std::vector<uint8_t> ref_codes(b->ref_off[b->n_windows]);
  for (size_t i = 0; i < ref_codes.size(); ++i) ref_codes[i] = (uint8_t)base_code(b->ref_bases[i]);
  std::vector<uint32_t> rinfo(R), bw(R + 1), gw(R + 1);
  uint32_t bo = 0, go = 0;
  for (uint32_t r = 0; r < R; ++r) { uint32_t len = b->seq_off[r + 1] - b->seq_off[r]; bw[r] = bo; gw[r] = go; bo += (len + 15) / 16; go += (len + 31) / 32; }
  std::vector<uint32_t> bases(bo + 4), good(go + 1);
  for (uint32_t r = 0; r < R; ++r)
    prep_read(P, b->seq, b->qual, b->seq_off[r], (int)(b->seq_off[r + 1] - b->seq_off[r]), b->label[r], b->strand[r], b->mate[r], b->mapped[r],
              &rinfo[r], bases.data(), bw[r], good.data(), gw[r]);
  DevBatch B;
  B.n_windows = b->n_windows; B.chr_id = b->chr_id; B.ref_start = b->ref_start; B.ref_off = b->ref_off; B.ref_codes = ref_codes.data();
  B.read_begin = b->read_begin; B.rinfo = rinfo.data(); B.name_rank = b->name_rank; B.base_woff = bw.data(); B.good_woff = gw.data();
  B.bases = bases.data(); B.good = good.data();
  B.bx_rank = P->lr_mode ? b->bx_rank : nullptr; B.hp = P->lr_mode ? b->hp : nullptr;
  // ---- one work slot
  size_t wbytes = lc_work_carve(nullptr, nullptr, C);
  std::vector<char> wmem(wbytes + 256);
  memset(wmem.data(), getenv("LANCET_EMU_POISON") ? atoi(getenv("LANCET_EMU_POISON")) : 0xCD, wmem.size());   // HBM is not zeroed: make stale-memory bugs show up here
  Work work; lc_work_carve(&work, wmem.data(), C);
  auto *res = new EmuResult();
  res->variants.resize(C.var_cap); res->blob.resize(C.blob_cap); res->stats.resize(b->n_windows);
  res->evt_len.assign(b->n_windows, 0); res->evt.assign((size_t)b->n_windows * (evt_cap ? evt_cap : 1), 0); res->evt_cap = evt_cap;
  uint32_t nv = 0, nb = 0, qh = 0, nx = 0;
  res->lr.resize(C.var_cap); res->bx_blob.resize(C.bx_cap + 1);
  DevOut O; memset(&O, 0, sizeof(O)); O.variants = res->variants.data(); O.blob = res->blob.data(); O.n_variants = &nv; O.n_blob = &nb; O.stats = res->stats.data();
  O.variants_lr = res->lr.data(); O.bx_blob = res->bx_blob.data(); O.n_bx = &nx;
  O.queue_head = &qh; O.phase = nullptr; O.win_list = nullptr; O.n_list = 0; O.evt_len = res->evt_len.data(); O.evt_out = res->evt.data();
  // ---- the LDS build kernel first (one emulated workgroup), unless switched off: LANCET_NO_PREBUILD=1 runs the general build for every window
  std::vector<uint8_t> pre, blscr, pool;
  static thread_local bl_small::BlShared BS;
  uint32_t bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t pool_cap = 0; int depth = 0;
  static thread_local bl_large::BlShared BSL;
  O.pre = nullptr; O.pre_pool = nullptr; O.n_ahead_used = &res->n_ahead_used;
  res->n_prebuilt = 0; res->n_ahead_built = 0; res->n_ahead_used = 0; res->n_biglist = 0;
  const bool lr_pre = !(getenv("LANCET_LR_PREBUILD") && atoi(getenv("LANCET_LR_PREBUILD")) == 0);      // (engine.hip: --linked-reads windows through the build kernel unless switched off)
  if ((!P->lr_mode || lr_pre) && !getenv("LANCET_NO_PREBUILD") && b->n_windows > 0) {
    pre.assign((size_t)b->n_windows * C.pl.stride, 0xCD); blscr.assign(bl_large::SCRATCH_BYTES + 256, 0xCD);
    memset(&BS, 0xCD, sizeof(BS)); memset(&BSL, 0xCD, sizeof(BSL));
    std::vector<uint32_t> biglist((size_t)b->n_windows + 1, 0);
    const bool large = getenv("LANCET_NO_LARGE_BUILD") == nullptr;
    depth = getenv("LANCET_AHEAD_DEPTH") ? atoi(getenv("LANCET_AHEAD_DEPTH")) : 6;
    pool_cap = (uint32_t)(b->n_windows / 4 + (depth > 0 ? 8 : 0) + (getenv("LANCET_NO_SVC") ? 0 : (b->n_windows < 128 ? 4 * b->n_windows + 24 : 536)));
    if (pool_cap) pool.assign((size_t)pool_cap * C.pl.stride, 0xCD);
    if (getenv("LANCET_EMU_FORCE_LARGE")) {                     // (test hook: every window through the 1024-lane configuration)
      for (int w = 0; w < b->n_windows; ++w) { biglist[(size_t)w] = (uint32_t)w; PreHdr *H = (PreHdr *)(pre.data() + (size_t)w * C.pl.stride); H->status = PB_NOT_BUILT; H->why = 0; H->have_rep = 0; H->heavy = 0; H->next = 0; }
      bq[4] = (uint32_t)b->n_windows;
    } else
    bl_small::build_kernel_body(P, &B, &C, pre.data(), blscr.data(), bq, BS, 0, nullptr, pool_cap ? pool.data() : nullptr, pool_cap, depth, large ? biglist.data() : nullptr, false);
    if (large) bl_large::build_kernel_body(P, &B, &C, pre.data(), blscr.data(), bq, BSL, 0, nullptr, pool_cap ? pool.data() : nullptr, pool_cap, depth, biglist.data(), true);
    res->n_biglist = bq[4];
    if (getenv("LANCET_EMU_SYNCS")) {
      unsigned long tot = 0; for (int i = 0; i < 16; ++i) tot += bl_emu_sync_acc[i];
      fprintf(stderr, "[emu] build kernel: %.1f barriers per window;", (double)tot / b->n_windows);
      for (int i = 0; i < 16; ++i) fprintf(stderr, " %d:%.1f", i, (double)bl_emu_sync_acc[i] / b->n_windows);
      fprintf(stderr, "\n");
    }
    O.pre = pre.data(); O.pre_pool = pool_cap ? pool.data() : nullptr;
    res->n_prebuilt = bq[1]; res->n_ahead_built = bq[3];
    res->n_cmp_done = 0;
    for (int w = 0; w < b->n_windows; ++w) { const PreHdr *H = (const PreHdr *)(pre.data() + (size_t)w * C.pl.stride); const PreCmp *CH = (const PreCmp *)(pre.data() + (size_t)w * C.pl.stride + C.pl.chdr); if (H->status == PB_BUILT && CH->done == 1u) ++res->n_cmp_done; }
    if (getenv("LANCET_EMU_CMP")) fprintf(stderr, "[emu] first compress done by the build kernel: %u of %u built windows\n", res->n_cmp_done, bq[1]);
    if (getenv("LANCET_EMU_HEAVY")) for (int w = 0; w < b->n_windows; ++w) { const PreHdr *H = (const PreHdr *)(pre.data() + (size_t)w * C.pl.stride); if (H->status == PB_BUILT && H->heavy) fprintf(stderr, "[emu] heavy %d K %d\n", w, H->K); }
    if (getenv("LANCET_EMU_HDR")) for (int w = 0; w < b->n_windows; ++w) { const PreHdr *H = (const PreHdr *)(pre.data() + (size_t)w * C.pl.stride); fprintf(stderr, "[emu] hdr %d status %u K %d N %u order %u refE %d refM %d heavy %u next %u\n", w, H->status, H->K, H->N, H->have_order, H->refE, H->refM, H->heavy, H->next); }
    if (getenv("LANCET_EMU_WHY")) for (int w = 0; w < b->n_windows; ++w) { const PreHdr *H = (const PreHdr *)(pre.data() + (size_t)w * C.pl.stride); if (H->status != PB_BUILT) fprintf(stderr, "[emu] window %d not prebuilt: why %u\n", w, H->why); }
  }
  static thread_local WinShared S;
  memset(&S, 0xCD, sizeof(S));
  // ---- the build service (engine.hip svc_kernel): LANCET_NO_SVC=1 off, LANCET_SVC_DEAD=1 nobody serves (the slots take their requests back)
  SvcCtl sv; memset(&sv, 0, sizeof(sv));
  std::vector<SvcReq> sreq; std::vector<uint32_t> srdy; std::vector<SvcCont> scont;
  if (O.pre && O.pre_pool && !getenv("LANCET_NO_SVC")) {
    sv.cap = (uint32_t)b->n_windows * 4u + 16u;
    sreq.assign(sv.cap, SvcReq{0, 0, SV_EMPTY, 0}); srdy.assign(sv.cap, 0u); scont.resize(sv.cap);
    sv.req = sreq.data(); sv.rdy = srdy.data(); sv.cont = scont.data();
    sv.alive = getenv("LANCET_SVC_DEAD") ? 0u : 1u;
    sv.large = (res->n_biglist * 8u > (uint32_t)b->n_windows || getenv("LANCET_EMU_SVC_LARGE")) ? 1u : 0u;   // (engine.hip: a batch with many windows beyond the 512-lane configuration runs the service in the 1024-lane one)
    O.svc = &sv;
  }
  while (window_kernel_body(P, &B, &C, &work, &O, &S, 0) != 0)
    if (sv.large) bl_large::svc_kernel_body(P, &B, &C, pre.data(), blscr.data(), bq, BSL, 0, pool.data(), pool_cap, getenv("LANCET_SVC_DEPTH") ? atoi(getenv("LANCET_SVC_DEPTH")) : depth, &sv);
    else bl_small::svc_kernel_body(P, &B, &C, pre.data(), blscr.data(), bq, BS, 0, pool.data(), pool_cap, getenv("LANCET_SVC_DEPTH") ? atoi(getenv("LANCET_SVC_DEPTH")) : depth, &sv);
  res->n_svc_built = sv.n_built; res->n_svc_stolen = sv.n_stolen; res->n_svc_posted = sv.req_alloc < sv.cap ? sv.req_alloc : sv.cap;
  if (getenv("LANCET_EMU_SVC")) fprintf(stderr, "[emu] svc posted %u built %u failed %u stolen %u\n", sv.req_alloc, sv.n_built, sv.n_failed, sv.n_stolen);
  res->n_variants = nv < C.var_cap ? nv : C.var_cap; res->n_blob = nb;
  return res;
}
extern "C" uint32_t lancet_emu_n_variants(void *h) { return ((EmuResult *)h)->n_variants; }
extern "C" const lancet_variant *lancet_emu_variants(void *h) { return ((EmuResult *)h)->variants.data(); }
extern "C" const char *lancet_emu_blob(void *h) { return ((EmuResult *)h)->blob.data(); }
extern "C" const lancet_variant_lr *lancet_emu_variants_lr(void *h) { return ((EmuResult *)h)->lr.data(); }
extern "C" const uint32_t *lancet_emu_bx_blob(void *h) { return ((EmuResult *)h)->bx_blob.data(); }
extern "C" uint32_t lancet_emu_blob_len(void *h) { return ((EmuResult *)h)->n_blob; }
extern "C" const lancet_window_stats *lancet_emu_stats(void *h) { return ((EmuResult *)h)->stats.data(); }
extern "C" const uint32_t *lancet_emu_evt_len(void *h) { return ((EmuResult *)h)->evt_len.data(); }
extern "C" const uint32_t *lancet_emu_evt(void *h) { return ((EmuResult *)h)->evt.data(); }
extern "C" uint32_t lancet_emu_n_prebuilt(void *h) { return ((EmuResult *)h)->n_prebuilt; }
extern "C" uint32_t lancet_emu_n_biglist(void *h) { return ((EmuResult *)h)->n_biglist; }
extern "C" uint32_t lancet_emu_n_ahead_built(void *h) { return ((EmuResult *)h)->n_ahead_built; }
extern "C" uint32_t lancet_emu_n_ahead_used(void *h) { return ((EmuResult *)h)->n_ahead_used; }
extern "C" uint32_t lancet_emu_n_cmp_done(void *h) { return ((EmuResult *)h)->n_cmp_done; }
extern "C" uint32_t lancet_emu_n_svc_posted(void *h) { return ((EmuResult *)h)->n_svc_posted; }
extern "C" uint32_t lancet_emu_n_svc_built(void *h) { return ((EmuResult *)h)->n_svc_built; }
extern "C" uint32_t lancet_emu_n_svc_stolen(void *h) { return ((EmuResult *)h)->n_svc_stolen; }
extern "C" void lancet_emu_free(void *h) { delete (EmuResult *)h; }

// repeat_scan both ways (bit-parallel LDS version vs the byte-wise restatement) for the unit test
extern "C" void lancet_emu_repeat_scan(const uint8_t *s, int len, int mm, int bitparallel, int *outE, int *outM) {
  static WinShared S;
  volatile int e = 0, m = 0;
  if (bitparallel) repeat_scan(S.rs, s, len, mm, &e, &m); else repeat_scan_bytes(s, len, mm, &e, &m);
  *outE = e; *outM = m;
}
// the long-matches-only scan: results below lminE / lminM may be reported smaller
// form 0: 4 bits per base; 1: 2 bits per base staged from the bytes (falls back to 4 when it meets an N); 2: 2 bits per base from
// a packed copy of the string (the LDS build kernel's call; no N allowed)
extern "C" void lancet_emu_repeat_scan_min(const uint8_t *s, int len, int mm, int lminE, int lminM, int *outE, int *outM, int form) {
  static WinShared S;
  volatile int e = 0, m = 0, bad = 0;
  if (form == 0) repeat_scan_min(S.rs, s, len, mm, lminE, lminM, &e, &m);
  else if (form == 1) repeat_scan_min(S.rs, s, len, mm, lminE, lminM, &e, &m, nullptr, &bad);
  else {
    static uint32_t packed[LC_MAXW / 16 + 64];
    memset(packed, 0, sizeof packed);
    for (int i = 0; i < len; ++i) packed[i >> 4] |= (uint32_t)(s[i] & 3u) << (2 * (i & 15));
    repeat_scan_min(S.rs, s, len, mm, lminE, lminM, &e, &m, packed, &bad);
  }
  *outE = e; *outM = m;
}

// the path form of the scan (kernels.h repeats_in_graph_paths): only the windows that overlap the positions [ra, rb) in either copy
extern "C" void lancet_emu_repeat_scan_range(const uint8_t *s, int len, int mm, int lminM, int ra, int rb, int *outM) {
  static WinShared S;
  volatile int e = 0, m = 0, bad = 0;
  repeat_scan_min(S.rs, s, len, mm, 0x7FFF, lminM, &e, &m, nullptr, &bad, ra, rb);
  *outM = m;
}

// find_tandems_local (test hook; must equal the oracle's restatement of src/util.cc:574-758; `local` is kept in the signature, ignored)
extern "C" int lancet_emu_find_tandems(const uint8_t *codes, int n, int pos, int max_unit_len, int min_report_units, int min_report_len, int dist_from_str,
                                       int local, int *len, uint8_t *motif, int *motif_len) {
  lancet_params P; memset(&P, 0, sizeof(P));
  P.max_unit_len = max_unit_len; P.min_report_units = min_report_units; P.min_report_len = min_report_len; P.dist_from_str = dist_from_str;
  static WinShared S;
  Ctx c; c.P = &P; c.B = nullptr; c.C = nullptr; c.W = nullptr; c.OUT = nullptr; c.S = &S;
  *len = 0;
  (void)local;
  bool a = find_tandems_local(c, codes, n, pos, len, motif, motif_len);
  return a ? 1 : 0;
}

// global_align_aff through the emulated kernels: mode 0 band with fall-back, 1 full matrix, 2 band only (returns -2 when not certified)
extern "C" int lancet_emu_align(const char *Sa, const char *Ta, char *S_aln, char *T_aln, int cap, int mode) {
  int n = (int)strlen(Sa), m = (int)strlen(Ta);
  if (n < 1 || m < 1 || n > LC_MAXW) return -1;
  EngineCaps caps; memset(&caps, 0, sizeof(caps));
  caps.reads_cap = 4; caps.occ_cap = 64; caps.node_cap = 16; caps.table_cap = 32; caps.bucket_cap = 32; caps.special_cap = 4; caps.surv_cap = 4;
  caps.seq_cap = 64; caps.queue_cap = 4; caps.path_cap = (uint32_t)m + 8; caps.max_k = 16; caps.qv_cap = 64;
  caps.max_w = LC_MAXW_DEFAULT > (((uint32_t)n + 63u) & ~63u) ? LC_MAXW_DEFAULT : (((uint32_t)n + 63u) & ~63u);
  size_t bytes = lc_work_carve(nullptr, nullptr, caps);
  std::vector<char> mem(bytes + 256, 0);
  Work w; lc_work_carve(&w, mem.data(), caps);
  auto code = [](char b) -> uint8_t { switch (b) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; } return 4; };
  std::vector<uint8_t> sc(n), tc(m);
  for (int i = 0; i < n; ++i) sc[i] = code(Sa[i]);
  for (int i = 0; i < m; ++i) tc[i] = code(Ta[i]);
  static WinShared S; memset(&S, 0, sizeof(S));
  Ctx c; c.P = nullptr; c.B = nullptr; c.C = &caps; c.W = &w; c.OUT = nullptr; c.S = &S;
  if (mode == 1 || !align_fill_band(c, sc.data(), n, tc.data(), m)) { if (mode == 2) return -2; align_fill(c, sc.data(), n, tc.data(), m); }
  int L = align_traceback(c, sc.data(), n, tc.data(), m);
  if (S.overflow || L + 1 > cap) return -1;
  align_traceback_fill(c, sc.data(), tc.data(), L);
  const int acap = (int)caps.max_w + (int)caps.path_cap + 2;
  memcpy(S_aln, w.aln, L); memcpy(T_aln, w.aln + acap, L); S_aln[L] = 0; T_aln[L] = 0;
  return L;
}
