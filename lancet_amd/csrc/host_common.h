// host_common.h -- host-side sizing of the HBM work space (shared by engine.hip and the test-side emulator).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "layout.h"
#include "../../include/lancet_engine.h"

static inline uint32_t lc_pow2_ge(uint32_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }
static inline uint32_t lc_bucket_cap_for(uint32_t nodes) {
  const uint32_t chain[] = {13u, 29u, 59u, 127u, 257u, 541u, 1109u, 2357u, 5087u, 10273u, 20753u, 42043u,
                            85229u, 172933u, 351061u, 712697u, 1447153u, 2938679u};
  for (int i = 0; i < 18; ++i) if (chain[i] >= nodes + 2) return chain[i] + 1;
  return 2938680u;
}

// Layout of one hand-off area of the LDS build kernel (layout.h PreLayout): `ncap` distinct k-mers, `qvcap` (candidate, position)
// quality rows, `kw` words per candidate key.
static inline PreLayout lc_pre_layout(uint32_t ncap, uint32_t qvcap, uint32_t kw, uint32_t maxw = LC_MAXW_DEFAULT, uint32_t lrcap = 0) {
  PreLayout L; memset(&L, 0, sizeof(L));
  L.ncap = ncap; L.qvcap = qvcap; L.kw = kw; L.maxw = maxw;
  uint32_t o = PRE_OFF_OCCREF + 4u * maxw;
  auto take = [&](uint32_t bytes, uint32_t align) { o = (o + align - 1u) & ~(align - 1u); const uint32_t at = o; o += bytes; return at; };
  L.refcov = take(8u * maxw, 64u);
  L.nhash = take(8u * ncap, 64u); L.surv = take(ncap, 64u);
  L.snode = take(4u * PB_CCAP, 64u); L.skey = take(8u * kw * PB_CCAP, 64u); L.sid = take(4u * PB_SCAP, 64u);
  L.pgr = take(128u * PB_SCAP, 128u); L.order = take(4u * PB_SCAP, 64u); L.qv = take(8u * qvcap, 64u);
  L.chdr = take(64u, 64u); L.clive = take(4u * (PB_CMAX + 2u), 64u); L.cseq = take(4u * PB_CSEQ, 64u);
  L.cord = take(8u * PB_CMAX, 64u); L.chl = take(12u * PB_CHEADS, 64u);
  if (lrcap) { L.lrnocc = take(4u * (ncap + 2u), 64u); L.lrcsr = take(4u * lrcap, 64u); L.lrcap = lrcap; }      // --linked-reads
  L.stride = (o + 255u) & ~255u;
  return L;
}
// Which form a batch gets: the wide one (PB_NCAP_WIDE k-mers, keys of 4 words, PB_QVCAP_WIDE rows: windows of 100x / 40x build at
// k = 31..101 with 9-12 k distinct k-mers) when its windows are deep (more than 400 reads on average: 60x / 60x windows, ~360 reads, all fit the narrow form) and `n_areas` of them stay
// within `budget` bytes; else the narrow one.  LANCET_PRE_WIDE=0 / 1 forces either (read by the caller, passed as `force`).
static inline uint32_t lc_max_w_for_batch(const lancet_window_batch *b) {      // EngineCaps::max_w
  uint32_t m = LC_MAXW_DEFAULT;
  for (int w = 0; w < b->n_windows; ++w) { const uint32_t l = b->ref_off[w + 1] - b->ref_off[w]; if (l > m && l <= LC_MAXW) m = l; }
  m = (m + 63u) & ~63u;
  return m > LC_MAXW ? (uint32_t)LC_MAXW : m;
}
static inline PreLayout lc_pre_layout_for_batch(const lancet_window_batch *b, size_t n_areas, size_t budget, int force = -1, bool lr = false) {
  const uint32_t mw = lc_max_w_for_batch(b);
  const PreLayout narrow = lc_pre_layout(PB_NCAP, PB_QVCAP, 1u, mw, lr ? PB_LRCAP : 0u), wide = lc_pre_layout(PB_NCAP_WIDE, PB_QVCAP_WIDE, LC_NWMAX, mw, lr ? PB_LRCAP_WIDE : 0u);
  if (force == 0 || b->n_windows <= 0) return narrow;
  if (force == 1) return wide;
  const uint32_t R = b->read_begin[b->n_windows];
  const bool deep = R / (uint32_t)b->n_windows > 400u;
  return (deep && n_areas * (size_t)wide.stride <= budget) ? wide : narrow;
}

// Work-space caps for a batch: the largest window decides.
// tier 1 = the common case (small tables: cheap to clear, cache/TLB friendly); tier 2 = worst case for the window
// shapes in the batch, used to re-run the windows that overflowed tier 1.
static inline EngineCaps lc_caps_for_batch(const lancet_window_batch *b, const lancet_params *p, uint32_t evt_cap,
                                           uint32_t max_nodes_limit, int tier = 2) {
  EngineCaps c; memset(&c, 0, sizeof(c));
  uint32_t max_reads = 0; uint64_t max_bases = 0;
  for (int w = 0; w < b->n_windows; ++w) {
    uint32_t r0 = b->read_begin[w], r1 = b->read_begin[w + 1];
    if (r1 - r0 > max_reads) max_reads = r1 - r0;
    uint64_t bases = (uint64_t)(b->seq_off[r1] - b->seq_off[r0]) + (b->ref_off[w + 1] - b->ref_off[w]);
    if (bases > max_bases) max_bases = bases;
  }
  if (tier == 1 && b->n_windows > 0) {
    // coverage pile-ups: a window far above the batch's typical size would size every slot's work space; tier 1 is laid
    // out for windows up to 4x the mean (they overflow at once and run in tier 2, whose limits are the true maxima)
    const uint32_t R = b->read_begin[b->n_windows];
    const uint64_t mean_reads = R / (uint32_t)b->n_windows, mean_bases = ((uint64_t)b->seq_off[R] + b->ref_off[b->n_windows]) / (uint64_t)b->n_windows;
    // (round 5: bases 2x the mean + 16 Ki instead of 4x + 32 Ki -- the occurrence-sized arrays are a third of a slot, and what lies between
    //  is a handful of windows per scan that run in the re-run tier like the pile-ups above them)
    const uint64_t lim_reads = 4 * mean_reads + 256, lim_bases = 2 * mean_bases + 16384;
    if (max_reads > lim_reads) max_reads = (uint32_t)lim_reads;
    if (max_bases > lim_bases) max_bases = lim_bases;
  }
  c.reads_cap = max_reads + 2;
  c.occ_cap = (uint32_t)max_bases + 64;           // every base starts at most one k-mer
  uint32_t nodes = c.occ_cap;
  if (nodes > max_nodes_limit) nodes = max_nodes_limit;
  c.node_cap = nodes;
  c.table_cap = lc_pow2_ge(2 * nodes);
  c.bucket_cap = lc_bucket_cap_for(nodes);
  c.special_cap = tier == 1 ? 64u : 4096u;     /* source + sink per component that touches the reference, per build */
  uint32_t maxk = (uint32_t)(p->max_k > 0 ? p->max_k : 101);
  c.max_k = maxk;
  if (tier == 1) {
    c.surv_cap = nodes < 4096 ? nodes : 4096;
    c.qv_cap = 4096u * 32u;                    // (survivor, position) entries: 4096 survivors at k<=32, 1297 at k=101
    c.queue_cap = 8192;
  } else {
    c.surv_cap = nodes;                        // every node may survive (--low-cov 0 on noisy reads: fuzz case 10083, 17 k survivors of 17.4 k nodes)
    c.qv_cap = c.surv_cap * maxk;
    // Graph_t::bfs keeps whole partial paths in its FIFO until DFS_LIMIT dequeues (reference src/Graph.cc:1299-1425); every
    // dequeue enqueues at most the node's out-degree (a few after compaction).  The worst-case tier holds 4 entries per
    // permitted dequeue, capped at 4 Mi entries (96 MB per slot); beyond that the window reports an overflow.
    { uint64_t q = 4ull * (uint64_t)(p->dfs_limit > 0 ? p->dfs_limit : 1000000) + 4096; if (q < 65536) q = 65536; if (q > (4ull << 20)) q = 4ull << 20; c.queue_cap = (uint32_t)q; }
  }
  c.seq_cap = 3 * c.qv_cap + 65536;
  c.max_w = lc_max_w_for_batch(b);              // (a window above LC_MAXW is reported LANCET_W_OVERFLOW on its own)
  c.path_cap = c.max_w + (uint32_t)p->max_indel_len + 256;
  c.evt_cap = evt_cap;
  // Records of the whole batch.  A scan emits ~0.7 per window; permissive settings (--low-cov 0 on noisy reads) reach several
  // hundred per window (fuzz case 7122: 355), and a small batch has no other windows to average that out: 64 per window and
  // a floor of 64 Ki records (lancet_variant is 64 bytes: 140 MB for a 32768-window batch).
  c.var_cap = (uint32_t)b->n_windows * 64u + 65536u;
  c.blob_cap = (uint32_t)b->n_windows * 4096u + (4u << 20);
  c.lr_mode = p->lr_mode ? 1u : 0u;
  c.bx_cap = c.lr_mode ? (uint32_t)b->n_windows * 8192u + (1u << 20) : 0u;
  c.pl = lc_pre_layout(PB_NCAP, PB_QVCAP, 1u, c.max_w);      // (the engine replaces it per upload: lc_pre_layout_for_batch)
  return c;
}

struct LcCarver {
  char *base; size_t off;
  template <class T> LC_GLOBAL T *take(size_t n) {
    off = (off + 63) & ~(size_t)63;
    LC_GLOBAL T *p = base ? (LC_GLOBAL T *)(base + off) : (LC_GLOBAL T *)nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// Lays one Work slot out at `base` (may be null to just measure).  Returns bytes used.
static inline size_t lc_work_carve(Work *w, char *base, const EngineCaps &c) {
  LcCarver k{base, 0};
  const size_t nodes = (size_t)c.node_cap + c.special_cap;
  const size_t MW = c.max_w ? c.max_w : LC_MAXW_DEFAULT;
  Work t; memset(&t, 0, sizeof(t));
  t.occ_base = k.take<uint32_t>(c.reads_cap + 1);
  t.rd = k.take<uint32_t>(4 * (size_t)c.reads_cap);
  t.cand = k.take<uint8_t>(c.reads_cap);
  t.mate_of = k.take<uint32_t>(c.reads_cap);
  t.items = k.take<uint32_t>(2 * ((size_t)c.reads_cap + MW / LC_SEG + 2));
  t.chunk = k.take<uint32_t>(2 * (((size_t)c.reads_cap + MW / LC_SEG + 2) / 64 + 2));
  t.occ = k.take<uint32_t>(c.occ_cap);
  t.slots = k.take<uint32_t>(4 * (size_t)c.table_cap);
  t.todo = k.take<uint32_t>(c.table_cap);
  t.mv = k.take<uint32_t>((c.wide_ids ? 8 : 4) * (size_t)c.occ_cap);
  t.slot_key = k.take<unsigned long long>((size_t)c.table_cap * LC_NWMAX);
  t.bitmap = k.take<uint32_t>((c.occ_cap + c.special_cap) / 32 + 4);   /* also the visited set of the component search (node ids) */
  t.bitpre = k.take<uint32_t>(c.occ_cap / 32 + 2);
  t.csr = k.take<uint32_t>((c.wide_ids ? 2 : 1) * (size_t)c.occ_cap);
  t.nkey = k.take<unsigned long long>(nodes * LC_NWMAX);
  t.nhash = k.take<unsigned long long>(nodes);
  t.nfill = k.take<uint32_t>((c.wide_ids ? 2 : 1) * (nodes + 1));
  t.gr = k.take<NodeGr>(nodes + 1);          /* + the stand-in record of reference k-mers whose node is gone (prebuilt windows) */
  t.cmp = k.take<CmpRec>(nodes);
  t.nocc = k.take<uint32_t>(nodes + 1);
  t.qv = k.take<uint16_t>((size_t)c.qv_cap * (c.lr_mode ? 10 : 4));
  t.qv_own = t.qv;
  t.khp = k.take<uint16_t>(c.lr_mode ? nodes * 6 : 1);
  t.refhp = k.take<uint16_t>(c.lr_mode ? MW * 6 : 1);
  t.bxbuf = k.take<uint32_t>(c.lr_mode ? c.reads_cap : 1);
  t.lr_refnode = k.take<uint32_t>(c.lr_mode ? MW : 1);
  t.seq = k.take<uint32_t>(c.seq_cap);
  t.ht_next = k.take<uint32_t>(nodes);
  t.ht_bucket = k.take<uint32_t>(c.bucket_cap);
  t.ht_cnt = k.take<uint32_t>(c.bucket_cap);
  t.ht_start = k.take<uint32_t>(c.bucket_cap);
  t.order = k.take<uint32_t>(nodes + 1);
  t.scratch = k.take<uint32_t>(2 * nodes > c.occ_cap ? 2 * nodes : c.occ_cap);
  t.refcov = k.take<uint16_t>(MW * 4);
  t.queue = k.take<BfsEntry>(c.queue_cap);
  t.pnodes = k.take<uint32_t>(nodes); t.pedges = k.take<uint32_t>(nodes);
  t.pdesc = k.take<uint32_t>(c.path_cap);
  t.pseq = k.take<uint8_t>(c.path_cap);
  t.tb = k.take<uint8_t>((size_t)(MW + c.path_cap + 4) * (MW + 2));   /* anti-diagonal-major: (n+m+1) diagonals of n+1 cells */
  t.dp = k.take<int32_t>(7 * (MW + 2));
  t.aln = k.take<uint8_t>(2 * (size_t)(MW + c.path_cap + 2));
  t.evt = k.take<uint32_t>(c.evt_cap + 8);
  t.survb = k.take<uint8_t>(nodes + 1);
  if (w) *w = t;
  return (k.off + 255) & ~(size_t)255;
}
