// kernels.h -- the per-window micro-assembly kernel for gfx950 (single source, see wave.h).
//
// One workgroup assembles one window at a time, start to finish: reference repeat scan, the self-tuning k
// loop, node-table build, graph clean-up, path enumeration, alignment and variant extraction -- the path the
// reference runs inside Microassembler::processGraph (reference src/Microassembler.cc:73-249).  Function
// headers cite the reference code whose observable behaviour they reproduce; the implementation is not a
// translation: k-mers are 2-bit packed integers, the node table is an open-addressing hash over HBM, nodes are
// dense ids in first-insertion order, unitigs are descriptor deques in an arena, libstdc++'s unordered_map
// iteration order (which the reference's results depend on) is computed explicitly.
#pragma once
#include "wave.h"
#include "layout.h"
#include "../../include/lancet_engine.h"

#ifndef LANCET_WG
#define LANCET_WG 64
#endif
// the per-position quality counts stage LC_QSTAGE occurrences per round, three per lane: all the lanes of the fat form
#ifdef LANCET_FAT
#define LC_QLANES LC_FAT_LANES
#define LC_PF_G 6                       /* 512-word sets in the staging area */
#define LC_QSTAGE (3 * LC_FAT_LANES)
#else
#define LC_QLANES LANCET_WG
#define LC_QSTAGE LC_STAGE
#endif

// ---------------------------------------------------------------------------------------------------------
// trace events (optional; host side turns them into the reference's `-v` text, tests compare with goldens)
// ---------------------------------------------------------------------------------------------------------
enum {
  EV_PROCESS = 1, EV_REPEAT_REF, EV_NEAR_REF, EV_READS, EV_STATS, EV_MARKREF, EV_LOWCOV, EV_CLEANDEAD,
  EV_COMPRESS, EV_CC, EV_CCID, EV_CCEND, EV_TRIM, EV_AMBIG_SRC, EV_NOMATCH_SRC, EV_AMBIG_SNK, EV_NOMATCH_SNK,
  EV_CYCLE, EV_TIPS_ROUND, EV_TIPS_REMOVED, EV_LINKS, EV_LOOKREP, EV_MISSING, EV_SEARCH, EV_NEAR_QRY,
  EV_DFSLIMIT, EV_PATH, EV_TS, EV_PATH_END, EV_EKA_END, EV_FOUND, EV_END
};

struct WinShared {
  int w, K, NW, R, reflen;
  uint32_t O, N, nspecial, M;          // occurrences, k-mer nodes, special nodes, live order length
  int overflow;
  int seq_t5, seq_len;                 // Ref_t::seq as a window of rawseq (carried across k)
  int trim5, trim3;                    // Ref_t::trim5/trim3 (unsigned short in the reference)
  int totalreadbp;
  uint32_t ht_bc, ht_next_resize, ht_elt, ht_head;
  uint32_t source, sink;
  int numcomp, refcomp;
  int repE, repM;
  uint32_t seq_top, qv_top;
  int emit_seq;
  uint32_t evt_len;
  unsigned long long n_kmers;
  uint32_t max_nodes, sum_nodes;
  int n_builds, final_k, status;
  int tmp0, tmp1, tmp2, tmp3, hasN;
  int rs_bad;                          // repeat_scan_min: the 2-bit staging met a code above 3
  // the paths findRepeatsInGraphPaths enumerated, kept for eka (W.mv): see pcache_store
  int pc_ok, pc_n, pc_i, pc_bits, pc_dfs, pc_end_dfs, bfs_dfs;
  uint32_t pc_top, pc_rd, pc_base;
  int nitems;                          // work items of the per-occurrence passes (build_items)
  uint32_t part[LANCET_WG + 1];
  uint32_t part2[LANCET_WG + 1];                 // second scan scratch (part[0..7] carry the path loop's state)
  int wk[4];                                     // walk_prepare: match / snp / ins / del columns
  int ps_first, ps_len, ps_hd, ps_tl;                 // path_string_wg: first real node, length ; Hamming distance to the reference
  int wk_n;                                      // walk_prepare: number of non-match columns
  int wk_stop, wk_nts, wk_last, wk_code, wk_tend, wk_tref;   // process_path_walk_wg: lane 0's state between the chunks of columns
  // (8 KB of LDS per workgroup = 20 single-wave workgroups per CU, the fifth wave per SIMD: two pairs of buffers that are
  //  never live together share their space)
  union alignas(16) {
    uint32_t mk[LC_QSTAGE][4];                   // staged quality masks of up to LC_QSTAGE occurrences (read back as 16-byte vectors)
    unsigned long long rs[LC_RS_WORDS];          // repeat_scan (window start): the string at 4 bits per base
    uint8_t lbytes[LC_QSTAGE * 16];              // alignment: the two strings (band fill) ; transcript walk: the transcripts (LC_TS_LDS of them)
  };
  uint32_t mmeta[LC_QSTAGE];
#ifdef LANCET_FAT
  uint32_t pacc[(LC_FAT_LANES / 2) * 10];        // step 2 split over lane groups: running counts per (candidate, position), 4 classes + 6 lr
  uint32_t pf_o0[LC_PF_G], pf_nk[LC_PF_G], pf_m0[LC_PF_G], pf_mnk[LC_PF_G], pf_r[LC_PF_G], pf_all[LC_PF_G];   // mate-overlap prefilter: LC_PF_G candidate reads per trip
#endif
  uint32_t g_n[LC_PACK], g_lo[LC_PACK], g_cnt[LC_PACK], g_es[LC_PACK], g_min[LC_PACK], g_N;   // the candidates of the current group
  uint32_t g_fl[LC_PACK]; float g_tt[LC_PACK], g_tn[LC_PACK];      // their flags and tumor / normal coverage (fetched while the occurrences are staged)
  uint32_t g_row[LC_PACK];                                         // build_qcounts in rows-only mode: the quality row of each candidate of the group
  union {
    struct { uint32_t cq_n[64], cq_lo[64], cq_cnt[64], cq_row[64]; };   // the next 64 candidates (node, csr start, occurrences; rows-only mode: the row to write), fetched together
    uint16_t acc[128][10];                       // a candidate with more occurrences than the staging area, between its rounds: per k-mer
                                                 // position running counts Tf Tr Nf Nr (+ lr_mode: T hp0-2, N hp0-2 minqv); the candidate list is re-fetched after it
  };
  uint32_t tmask, N_last; int tfull;                 // open-addressing table of this build: size - 1, filled up, nodes of the window's previous build
  int cmp_ok;                                    // compress_prepare: the component qualifies for compress_fast
  uint32_t cmp_nh;                               // compress_rank: heads found
  int cmp_done;                                  // the graph came with markRefEnds and the first compress done (build_lds_impl.h bl_compress_first)
  uint32_t cmp_dead, cmp_edges0, cmp_nsurv;      // ... its cleanDead count, component 1's edge total before markRefEnds (trace), the survivors
  uint32_t cmp_edges_all, cmp_n1, cmp_refmask;   // ... all survivors' edge total, the survivors in component 1, which components hold a reference k-mer (round 6: graphs of several components)
  int seq_lazy;                                  // graph from the LDS build kernel: the k-mer nodes' descriptors are not written yet (seq_materialize)
  unsigned long long lz_area;                    // ... its hand-off area
  unsigned long long lz_skey; uint32_t lz_kw;    // ... the candidate keys there, words per key
  int mr_src, mr_snk, mr_ambs, mr_ambk;          // mark_ref_scan: first / last qualifying reference offset, ambiguity flags
  int QS, LR;                                    // counters per (survivor, position): 4, or 10 with --linked-reads ; lr_mode
  unsigned long long t_last, phase_acc[16];
  int phase_cur;
  int prebuilt;                                  // this build came from the LDS build kernel (build_lds.h): survb[] instead of the stale node records
  uint32_t pre_edges, pre_refn;                  // its trace aggregates
  int pre_order;                                 // ... and it came with the survivors' table order and the components
  int items_ready;                               // build_items ran for this window
  int al_band, al_lo, al_score;                  // alignment: traceback bytes in band layout (offsets j - i from al_lo, two per lane), score at (n, m)
  int al_L;                                      // align_traceback_wg: alignment length
  int nosusp_k;                                  // build service: the k this window was resumed for (no second request for it)
  uint32_t svc_i;                                // ... the request just posted
  int act, act_arg;                              // what the slot does next (window_kernel_body)
  int gc_ok;                                     // graph_cache_wg: the live nodes' edge lists are in LDS (the staging area)
  uint32_t cn_mn, cn_mq;                         // compress_node_wv: minima over the descriptors a merge appends
};

// The one WinShared of the workgroup.  Functions reach it by name rather than through the pointer in Ctx: a pointer
// loaded from a structure is generic to the compiler (FLAT instructions for every control word), the variable is LDS.
typedef volatile LC_LDS WinShared LC_WS;
#ifndef LANCET_WAVE_EMU
static __shared__ WinShared lc_shared;
#define LC_SREF(c) (*(LC_WS *)&lc_shared)
#else
#define LC_SREF(c) (*(c).S)
#endif

struct Ctx {
  LC_GLOBAL const lancet_params *P;
  LC_GLOBAL const DevBatch *B;
  LC_GLOBAL const EngineCaps *C;
  LC_GLOBAL Work *W;
  LC_GLOBAL DevOut *OUT;
  LC_WS *S;                // volatile: control words written by lane 0 and read by every lane after a barrier must be
                           // real LDS accesses (the optimiser was observed to drop/sink such stores across the barrier)
};
// The context pointers are read in every function; Ctx itself travels by reference through the separately compiled
// phase functions, i.e. it lives in the kernel's private memory and each `c.W` there is a FLAT load from scratch.  The
// workgroup keeps one copy in LDS, reached by name (DS loads, no aliasing with the global stores).
#ifndef LANCET_WAVE_EMU
static __shared__ Ctx lc_ctx;
#define LC_CTX(c) (*(LC_LDS Ctx *)&lc_ctx)
#define LC_CTX_PUBLISH(c) do { if (threadIdx.x == 0) { LC_CTX(c).P = (c).P; LC_CTX(c).B = (c).B; LC_CTX(c).C = (c).C; LC_CTX(c).W = (c).W; LC_CTX(c).OUT = (c).OUT; } __syncthreads(); } while (0)
#else
#define LC_CTX(c) (c)
#define LC_CTX_PUBLISH(c) ((void)0)
#endif


#ifdef LANCET_WAVE_EMU   /* (emulator build: LANCET_OVF_DEBUG=1 names the limit that was hit) */
#define OVF(c) do { LC_SREF(c).overflow = 1; if (getenv("LANCET_OVF_DEBUG")) fprintf(stderr, "[emu] work-space limit hit at kernels.h:%d\n", __LINE__); } while (0)
#else
#define OVF(c) do { LC_SREF(c).overflow = 1; } while (0)
#endif
// profiling only (EngineCaps::debug_stop): abandon the window after a phase marker, as an overflow
#define STOP_SET(c, id) do { if (LC_CTX(c).C->debug_stop == (uint32_t)(id)) { WG_LANE0 { OVF(c); } } } while (0)
#define STOP_RET(c, id) do { if (LC_CTX(c).C->debug_stop == (uint32_t)(id)) { WG_LANE0 { OVF(c); } return; } } while (0)

// per-phase wall-clock accounting (lane 0; 100 MHz constant counter), read back through lancet_engine_phase_times
#ifndef LANCET_WAVE_EMU
#define PHASE(c, id) do { if (threadIdx.x == 0 && (c).S->phase_cur >= 0) { unsigned long long _t = wall_clock64(); (c).S->phase_acc[(c).S->phase_cur] += _t - (c).S->t_last; (c).S->t_last = _t; (c).S->phase_cur = (id); } } while (0)
#else
#define PHASE(c, id) ((void)0)
#endif
// profiling builds only (tools/subphase.sh: -DLANCET_PROF=<group>): the steps inside ONE of the coarse phases, accounted in the slots of
// the general build's phases (2..7, idle on windows whose graphs all come from the LDS build kernel); SUBEND returns to the coarse phase
#ifdef LANCET_PROF
#define SUBPHASE(c, group, id) do { if (LANCET_PROF == (group)) PHASE(c, id); } while (0)
#else
#define SUBPHASE(c, group, id) ((void)0)
#endif

// Uniform read of a control word: barrier, everybody reads, barrier (so that the next writer cannot race a
// slow reader).  Every branch that contains a WG_SYNC must be decided through this.
// The value is passed through readfirstlane so that the compiler sees a wave-uniform (scalar) branch condition:
// with a lane-varying condition it builds EXEC-masked control flow around the phases and, on gfx950, was observed
// to let lanes 1..63 evaluate a loop condition before lane 0 had executed the preceding lane-0 section.
DEV int wg_uniform(int v) {
#ifndef LANCET_WAVE_EMU
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}
// The pointer is laundered through an empty asm so that the optimiser cannot correlate the value with a store
// made by the preceding lane-0 section (observed on gfx950/ROCm 7.2: the lane-0 path was threaded straight into
// a following `while (wg_bcast(..))` loop, lanes 1..63 then evaluated the loop condition before lane 0's store).
template <class P> DEV int wg_bcast(P p) {
  WG_SYNC();
#ifndef LANCET_WAVE_EMU
  asm volatile("" : "+v"(p) : : "memory");
#endif
  int v = wg_uniform((int)*p);
  WG_SYNC();
  return v;
}
template <class P> DEV uint32_t wg_bcastu(P p) { return (uint32_t)wg_bcast(p); }

// (the record is written out of line: inlined at its ~60 call sites the eight operands were what the window kernel spilled -- 142 VGPRs,
//  all on paths that run with tracing on only)
DEVNI void evt_write(Ctx &c, uint32_t code, uint32_t a, uint32_t b, uint32_t d, uint32_t e, uint32_t f, uint32_t g, uint32_t h) {
  LC_WS &S = LC_SREF(c);
  if (S.evt_len + 8 > LC_CTX(c).C->evt_cap) return;
  uint32_t *p = LC_CTX(c).W->evt + S.evt_len;
  p[0] = code; p[1] = a; p[2] = b; p[3] = d; p[4] = e; p[5] = f; p[6] = g; p[7] = h;
  S.evt_len += 8;
}
DEV void evt(Ctx &c, uint32_t code, uint32_t a = 0, uint32_t b = 0, uint32_t d = 0, uint32_t e = 0, uint32_t f = 0,
             uint32_t g = 0, uint32_t h = 0) {
  if (!LC_CTX(c).C->evt_cap) return;
  evt_write(c, code, a, b, d, e, f, g, h);
}
DEV void evt_bytes(Ctx &c, const uint8_t *s, uint32_t n) {   // raw bytes appended after an event, padded to 8 words
  if (!LC_CTX(c).C->evt_cap) return;
  LC_WS &S = LC_SREF(c);
  uint32_t words = ((n + 3) / 4 + 7) / 8 * 8;
  if (S.evt_len + words > LC_CTX(c).C->evt_cap) return;
  uint8_t *p = (uint8_t *)(LC_CTX(c).W->evt + S.evt_len);
  for (uint32_t i = 0; i < words * 4; ++i) p[i] = i < n ? s[i] : 0;
  S.evt_len += words;
}

// ---------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------
DEV int rd_base(const uint32_t *bases, uint32_t woff, int i) { return (bases[woff + (i >> 4)] >> ((i & 15) * 2)) & 3; }
DEV int rd_good(const uint32_t *good, uint32_t woff, int i) { return (good[woff + (i >> 5)] >> (i & 31)) & 1; }
// number of quality-passing bases in [a, b)
DEV int good_count(const uint32_t *good, uint32_t woff, int a, int b) {
  int n = 0;
  for (int i = a; i < b;) {
    int w = i >> 5, lo = i & 31;
    int take = 32 - lo; if (take > b - i) take = b - i;
    uint32_t m = good[woff + w] >> lo;
    if (take < 32) m &= ((1u << take) - 1u);
    n += dev_popc(m);
    i += take;
  }
  return n;
}

DEV bool is_dir(uint32_t edir, char dir) {                        // Edge_t::isDir, reference src/Edge.cc:25-31
  return dir == 'F' ? (edir == 0 || edir == 1) : (edir == 3 || edir == 2);
}
DEV char dir_start(uint32_t d) { return (d == 0 || d == 1) ? 'F' : 'R'; }    // reference src/Edge.hh:71-75
DEV char dir_dest(uint32_t d) { return (d == 0 || d == 2) ? 'F' : 'R'; }     // :77-81
DEV uint32_t flipme(uint32_t d) { return d == 0 ? 2u : d == 1 ? 3u : d == 2 ? 0u : 1u; }   // :89-97
DEV uint32_t fliplink(uint32_t d) { return d == 0 ? 3u : d == 3 ? 0u : d; }                 // :99-107

// 64-bit finaliser used for the open-addressing table (not observable)
DEV unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// libstdc++ std::hash<std::string> == _Hash_bytes(ptr, len, 0xc70f6907) (64-bit MurmurHash2 variant,
// SURVEY.md Appendix A.1).  `get(i)` returns byte i of the string.
template <class G>
DEV unsigned long long std_hash_bytes(G get, int len) {
  const unsigned long long mul = (0xc6a4a793ULL << 32) + 0x5bd1e995ULL;
  unsigned long long hash = 0xc70f6907ULL ^ ((unsigned long long)len * mul);
  int nblk = len & ~7;
  for (int i = 0; i < nblk; i += 8) {
    unsigned long long d = 0;
    for (int j = 7; j >= 0; --j) d = (d << 8) | (unsigned long long)get(i + j);
    d *= mul; d ^= d >> 47; d *= mul;
    hash ^= d; hash *= mul;
  }
  if (len & 7) {
    unsigned long long d = 0;
    for (int j = (len & 7) - 1; j >= 0; --j) d = (d << 8) | (unsigned long long)get(nblk + j);
    hash ^= d; hash *= mul;
  }
  hash ^= hash >> 47; hash *= mul; hash ^= hash >> 47;
  return hash;
}

// k-mer keys: 2*K bits right-aligned in NW little-endian 64-bit words, first base in the most significant
// position, A<C<G<T = 0..3, so integer order == std::string order (reference src/Mer.hh:57-71).
DEV int key_base(const unsigned long long *k, int K, int j) {      // j-th character
  int bit = 2 * (K - 1 - j);
  return (int)((k[bit >> 6] >> (bit & 63)) & 3ULL);
}
DEV void key_push_fw(unsigned long long *k, int NW, int K, int b) { // k = (k << 2 | b) mod 4^K
  for (int w = NW - 1; w > 0; --w) k[w] = (k[w] << 2) | (k[w - 1] >> 62);
  k[0] = (k[0] << 2) | (unsigned long long)b;
  int top = 2 * K - 64 * (NW - 1);
  if (top < 64) k[NW - 1] &= ((1ULL << top) - 1ULL);
}
DEV void key_push_rc(unsigned long long *k, int NW, int K, int b) { // k = (k >> 2) | (3-b) << 2(K-1)
  for (int w = 0; w < NW - 1; ++w) k[w] = (k[w] >> 2) | (k[w + 1] << 62);
  k[NW - 1] >>= 2;
  int bit = 2 * (K - 1);
  k[bit >> 6] |= ((unsigned long long)(3 - b)) << (bit & 63);
}
DEV bool key_less(const unsigned long long *a, const unsigned long long *b, int NW) {
  for (int w = NW - 1; w >= 0; --w) { if (a[w] != b[w]) return a[w] < b[w]; }
  return false;
}
// j-th character of candidate ci's key in a hand-off area of the LDS build kernel (layout.h PRE_OFF_SKEY: `kw` words per candidate)
DEV int pre_key_base(LC_GLOBAL const unsigned long long *skey, uint32_t kw, uint32_t ci, int K, int j) {
  const int bit = 2 * (K - 1 - j);
  return (int)((skey[(size_t)ci * kw + (uint32_t)(bit >> 6)] >> (bit & 63)) & 3ULL);
}
#define LC_PL(c) (LC_CTX(c).C->pl)

// ---------------------------------------------------------------------------------------------------------
// K0: reference repeat scan.  isRepeat / isAlmostRepeat (reference src/util.cc:295-360) for every k at once:
//   isRepeat(seq,k)          <=> E >= k      E = longest exact self-match between offsets a<b with b+k-1 <= len-2
//   isAlmostRepeat(seq,k,mm) <=> Mm >= k+1   Mm = longest self-match window with <= mm mismatches, b+L-1 <= len-1
// (the reference skips the last k-mer in both loops).  One shift d = b-a per lane.
// ---------------------------------------------------------------------------------------------------------
DEV void repeat_scan_bytes(LC_GLOBAL const uint8_t *s, int len, int mm, volatile LC_LDS int *outE, volatile LC_LDS int *outM) {
  WG_LANE0 { *outE = 0; *outM = 0; }
  WG_SYNC();
  WG_FOR(dd, len > 1 ? len - 1 : 0) {
    int d = dd + 1;
    int run = 0, bestE = 0;
    int lenE = len - 1 - d;                         // positions p with p + d <= len-2
    for (int p = 0; p < lenE; ++p) { if (s[p] == s[p + d]) { ++run; if (run > bestE) bestE = run; } else run = 0; }
    int lenM = len - d, lo = 0, mis = 0, bestM = 0;  // positions p with p + d <= len-1
    for (int p = 0; p < lenM; ++p) {
      if (s[p] != s[p + d]) ++mis;
      while (mis > mm) { if (s[lo] != s[lo + d]) --mis; ++lo; }
      if (p - lo + 1 > bestM) bestM = p - lo + 1;
    }
    if (bestE > 0) dev_atomic_max((LC_LDS uint32_t *)outE, (uint32_t)bestE);
    if (bestM > 0) dev_atomic_max((LC_LDS uint32_t *)outM, (uint32_t)bestM);
  }
  WG_SYNC();
}
// Same result, bit-parallel: the string is staged in LDS at 4 bits per base; a lane owns a shift d and pulls the
// mismatch flags of 16 positions out of two unaligned 64-bit words.  Only mismatch positions are visited:
// with last[j] = position of the (j+1)-th most recent mismatch, the longest exact run ending before a mismatch q is
// q-1-last[0] and the longest window with <= mm mismatches ending there is q-1-last[mm] (the two-pointer window).
DEVNI void repeat_scan(volatile LC_LDS unsigned long long *rsbuf, LC_GLOBAL const uint8_t *s, int len, int mm, volatile LC_LDS int *outE, volatile LC_LDS int *outM) {
  if (mm > 7 || mm < 0 || len + 48 > 16 * LC_RS_WORDS) { repeat_scan_bytes(s, len, mm, outE, outM); return; }
  WG_LANE0 { *outE = 0; *outM = 0; }
  const int nwords = len / 16 + 3;
  WG_FOR(w, nwords) {
    unsigned long long v = 0;
    for (int j = 0; j < 16; ++j) { int idx = 16 * w + j; if (idx < len) v |= (unsigned long long)(s[idx] & 15u) << (4 * j); }
    rsbuf[w] = v;
  }
  WG_SYNC();
  const LC_LDS unsigned long long *rs = (const LC_LDS unsigned long long *)rsbuf;
  WG_FOR(dd, len > 1 ? len - 1 : 0) {
    const int d = dd + 1;
    const int lenE = len - 1 - d, lenM = len - d;
    int l0 = -1, l1 = -1, l2 = -1, l3 = -1, l4 = -1, l5 = -1, l6 = -1, l7 = -1, lm = -1;
    int bestE = 0, bestM = 0;
    for (int p0 = 0; p0 < lenM; p0 += 16) {
      const int wa = p0 >> 4, wb = (p0 + d) >> 4, sb = ((p0 + d) & 15) * 4;
      const unsigned long long a = rs[wa];
      const unsigned long long b = sb ? ((rs[wb] >> sb) | (rs[wb + 1] << (64 - sb))) : rs[wb];
      unsigned long long x = a ^ b;
      unsigned long long ne = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x1111111111111111ULL;
      if (lenM - p0 < 16) ne &= (1ULL << (4 * (lenM - p0))) - 1ULL;
      while (ne) {
        const int q = p0 + (__builtin_ctzll(ne) >> 2);
        ne &= ne - 1ULL;
        const int cm = q - 1 - lm; if (cm > bestM) bestM = cm;
        const int ce = (q < lenE ? q : lenE) - 1 - l0; if (ce > bestE) bestE = ce;
        l7 = l6; l6 = l5; l5 = l4; l4 = l3; l3 = l2; l2 = l1; l1 = l0; l0 = q;
        lm = mm == 0 ? l0 : mm == 1 ? l1 : mm == 2 ? l2 : mm == 3 ? l3 : mm == 4 ? l4 : mm == 5 ? l5 : mm == 6 ? l6 : l7;
      }
    }
    { const int cm = lenM - 1 - lm; if (cm > bestM) bestM = cm;
      const int ce = lenE - 1 - l0; if (ce > bestE) bestE = ce; }
    if (bestE > 0) dev_atomic_max((LC_LDS uint32_t *)outE, (uint32_t)bestE);
    if (bestM > 0) dev_atomic_max((LC_LDS uint32_t *)outM, (uint32_t)bestM);
  }
  WG_SYNC();
}

// The same operands where only long matches count: the callers compare E with k and M with k + 1 for k >= some k0, so a result
// below lminE (resp. lminM) may be reported as anything smaller.  A window of >= lminM positions with <= mm mismatches holds a
// run of >= (lminM - mm) / (mm + 1) matching positions, so the lane of a shift walks its mismatch mask a word (16 positions)
// at a time looking only for the END of a match run of >= rmin positions (on random sequence one word in a hundred), and only
// there looks up the mm + 1 mismatches on either side (backward / forward scans of the mask) to size the windows around the run:
//   windows with i mismatches left of the run and mm - i right of it: (prev_i, next_{mm-i}) exclusive, i = 0..mm.
// 20 ALU operations per word instead of 15 per mismatch position (12 of 16 positions mismatch on random sequence).
// `packed2`: when non-null the string is taken from there (2 bits per base, 16 bases per word, no N) instead of from s.
// B = bits per base in the staged copy: 4 (any code, 16 positions per 64-bit word) or 2 (codes 0..3 only, 32 positions per word:
// half the words to walk).  The 2-bit form is used whenever the string holds no N; `bad2` (LDS) is raised by the staging pass
// of the 2-bit form when it meets a code above 3, and the caller then runs the 4-bit form.
// [ra, rb): when given (the repeat test of a PATH, repeats_in_graph_paths), only windows that overlap these positions -- in either copy of
// the string -- are looked at: every shift walks the mask words of the positions [ra, rb) and of [ra - d, rb - d) instead of the whole
// string.  Reporting a window outside the range as well is harmless (it is a near-repeat of the string all the same); E is not exact then.
template <int B>
DEV void repeat_scan_min_t(volatile LC_LDS unsigned long long *rsbuf, LC_GLOBAL const uint8_t *s, int len, int mm, int rmin,
                           volatile LC_LDS int *outE, volatile LC_LDS int *outM, const LC_LDS uint32_t *packed2, volatile LC_LDS int *bad2,
                           int ra = 0, int rb = 0x7FFFFFFF) {
  constexpr int PW = 64 / B, SH = B == 4 ? 4 : 5, LB = B == 4 ? 2 : 1;
  constexpr unsigned long long M1 = B == 4 ? 0x1111111111111111ULL : 0x5555555555555555ULL;
  const int nwords = len / PW + 3;
  const bool al4 = (((size_t)s) & 3u) == 0;
  WG_FOR(w, nwords) {
    unsigned long long v = 0;
    if (B == 2) {
      if (packed2) {
        const int nw32 = (len + 15) >> 4;
        const unsigned long long lo = 2 * w < nw32 ? (unsigned long long)packed2[2 * w] : 0ULL, hi = 2 * w + 1 < nw32 ? (unsigned long long)packed2[2 * w + 1] : 0ULL;
        v = lo | (hi << 32);
        if (len - PW * w < PW) v &= len - PW * w > 0 ? (1ULL << (2 * (len - PW * w))) - 1ULL : 0ULL;
      } else if (al4 && PW * w + PW <= len) {                      // eight aligned 4-byte loads, issued together
        LC_GLOBAL const uint32_t *q = (LC_GLOBAL const uint32_t *)(s + PW * w);
        const uint32_t a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4], a5 = q[5], a6 = q[6], a7 = q[7];
        const uint32_t aa[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
        uint32_t any = 0;
        for (int t = 0; t < 8; ++t) { any |= aa[t]; for (int j = 0; j < 4; ++j) v |= (unsigned long long)((aa[t] >> (8 * j)) & 3u) << (2 * (4 * t + j)); }
        if (any & 0xFCFCFCFCu) *bad2 = 1;
      } else {
        for (int j = 0; j < PW; ++j) { const int idx = PW * w + j; if (idx < len) { const uint32_t cde = s[idx]; if (cde > 3u) *bad2 = 1; v |= (unsigned long long)(cde & 3u) << (2 * j); } }
      }
    } else {
      if (packed2) {                                               // 2-bit groups spread to nibbles
        unsigned long long x = 16 * w < len ? (unsigned long long)packed2[w] : 0ULL;
        x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL; x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
        x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL; x = (x | (x << 2)) & 0x3333333333333333ULL;
        if (len - 16 * w < 16 && len - 16 * w > 0) x &= (1ULL << (4 * (len - 16 * w))) - 1ULL;
        v = x;
      } else if (al4 && 16 * w + 16 <= len) {                      // four aligned 4-byte loads, issued together
        LC_GLOBAL const uint32_t *q = (LC_GLOBAL const uint32_t *)(s + 16 * w);
        const uint32_t a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
        const uint32_t aa[4] = {a0, a1, a2, a3};
        for (int t = 0; t < 4; ++t) for (int j = 0; j < 4; ++j) v |= (unsigned long long)((aa[t] >> (8 * j)) & 15u) << (4 * (4 * t + j));
      } else {
        for (int j = 0; j < 16; ++j) { int idx = 16 * w + j; if (idx < len) v |= (unsigned long long)(s[idx] & 15u) << (4 * j); }
      }
    }
    rsbuf[w] = v;
  }
  WG_SYNC();
  if (B == 2 && wg_bcast(bad2)) return;                           // a code above 3: the caller runs the 4-bit form
  const LC_LDS unsigned long long *rs = (const LC_LDS unsigned long long *)rsbuf;
  const int nsh = len > 1 ? len - 1 : 0;
  // shift d = item + 1: a lane that gets a second item gets a short one (the longest shifts are the first items)
  WG_FOR(it, nsh) {
    const int d = it + 1;
    const int lenE = len - 1 - d, lenM = len - d;
    // mismatch mask of word w of this shift: bit B j set <=> position PW w + j mismatches (positions >= lenM read as mismatches)
    auto NE = [&](int w) -> unsigned long long {
      const int p0 = w << SH;
      const int wb = (p0 + d) >> SH, sb = ((p0 + d) & (PW - 1)) * B;
      const unsigned long long a = rs[w];
      const unsigned long long b = sb ? ((rs[wb] >> sb) | (rs[wb + 1] << (64 - sb))) : rs[wb];
      const unsigned long long x = a ^ b;
      unsigned long long ne = B == 4 ? ((x | (x >> 1) | (x >> 2) | (x >> 3)) & M1) : ((x | (x >> 1)) & M1);
      if (lenM - p0 < PW) ne |= (lenM - p0 <= 0) ? M1 : (M1 << (B * (lenM - p0)));
      return ne;
    };
    auto prev_mis = [&](int pos) -> int {                        // largest mismatch position < pos, or -1
      int w = (pos - 1) >> SH;
      if (pos <= 0) return -1;
      unsigned long long m = NE(w);
      const int hi = pos - (w << SH);                            // positions [0, hi) of the word
      if (hi < PW) m &= (1ULL << (B * hi)) - 1ULL;
      while (true) {
        if (m) return (w << SH) + ((63 - __builtin_clzll(m)) >> LB);
        if (--w < 0) return -1;
        m = NE(w);
      }
    };
    auto next_mis = [&](int pos) -> int {                        // smallest mismatch position > pos, or lenM
      int w = (pos + 1) >> SH;
      if (pos + 1 >= lenM) return lenM;
      unsigned long long m = NE(w);
      const int lo = pos + 1 - (w << SH);
      if (lo > 0) m &= ~((1ULL << (B * lo)) - 1ULL);
      while (true) {
        if (m) { const int q = (w << SH) + (__builtin_ctzll(m) >> LB); return q < lenM ? q : lenM; }
        if ((++w << SH) >= lenM) return lenM;
        m = NE(w);
      }
    };
    int bestE = 0, bestM = 0;
    auto run_end = [&](int a, int b) {                             // maximal match run [a, b), b - a >= rmin
      { const int e = (b < lenE ? b : lenE) - a; if (e > bestE) bestE = e; }
      int pv[8];                                                   // pv[i]: the (i+1)-th mismatch left of a (-1: none)
      int q = a;  for (int i = 0; i <= mm; ++i) { q = q >= 0 ? prev_mis(q) : -1; pv[i] = q; }
      q = b - 1;                                                   // the (j+1)-th mismatch at or right of b closes the windows with mm - j mismatches on the left
      for (int j = 0; j <= mm; ++j) { q = q < lenM ? next_mis(q) : lenM; const int L = q - pv[mm - j] - 1; if (L > bestM) bestM = L; }
    };
    // The walk only NOTES the long runs (a << 16 | b, up to six per shift in registers); they are sized afterwards, all lanes of
    // the wave together -- sizing a run where it is found would serialise the wave on the lane that found it.
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0; int nc = 0;
    auto note = [&](int a, int b) {
      const uint32_t v = ((uint32_t)a << 16) | (uint32_t)b;
      if (nc == 0) c0 = v; else if (nc == 1) c1 = v; else if (nc == 2) c2 = v; else if (nc == 3) c3 = v; else if (nc == 4) c4 = v; else if (nc == 5) c5 = v;
      else run_end(a, b);
      ++nc;
    };
    const int nw = (lenM + PW - 1) >> SH;
    // the words to walk: all of them, or (a range was given) those of [ra, rb) and of [ra - d, rb - d), the lower stretch first
    int wlo[2] = {0, 0}, whi[2] = {0, nw};
    if (ra > 0 || rb < lenM) {
      const int a0 = ra > 0 ? ra : 0, a1 = rb < lenM ? rb : lenM, b0 = ra - d > 0 ? ra - d : 0, b1 = rb - d < lenM ? rb - d : lenM;
      wlo[1] = a0 >> SH; whi[1] = a1 > a0 ? (a1 + PW - 1) >> SH : wlo[1];
      wlo[0] = b0 >> SH; whi[0] = b1 > b0 ? (b1 + PW - 1) >> SH : wlo[0];
      if (whi[0] >= wlo[1]) { if (wlo[0] < wlo[1]) wlo[1] = wlo[0]; if (whi[0] > whi[1]) whi[1] = whi[0]; whi[0] = wlo[0]; }      // (they touch: one stretch)
      if (whi[1] > nw) whi[1] = nw;
      if (whi[0] > nw) whi[0] = nw;
    }
    for (int part = 0; part < 2; ++part) {
    int run = 0;                                                   // matches ending just before the current word (a run that began before the
                                                                   // stretch counts from the stretch's start: the windows looked for lie inside it)
    for (int w = wlo[part]; w < whi[part]; ++w) {
      const unsigned long long ne = NE(w);
      const int p0 = w << SH;
      if (ne == 0) { run += PW; continue; }
      const int q1 = __builtin_ctzll(ne) >> LB, ql = (63 - __builtin_clzll(ne)) >> LB;
      if (run + q1 >= rmin) note(p0 - run, p0 + q1);
      if (rmin <= PW - 2 && ql - q1 > rmin) {                      // a run of >= rmin matches between two mismatches of this word?
        const unsigned long long mt = (~ne) & M1;
        unsigned long long t = mt;
        for (int i = 1; i < rmin; ++i) t &= (mt >> (B * i));
        t &= ~((2ULL << (B * q1)) - 1ULL);                         // starts after the first mismatch ...
        t &= (1ULL << (B * ql)) - 1ULL;                            // ... and before the last one
        // every set bit of t starts rmin matching positions; the lowest one of a group is where a long run begins, the next
        // mismatch above it is where it ends.  One trip per LONG run (walking all the word's mismatches instead made every
        // lane of the wave wait ~24 trips whenever one lane had a long run in its word -- which is nearly always).
        while (t) {
          const int sb = __builtin_ctzll(t) >> LB;                          // run [sb, e)
          const unsigned long long above = ne >> (B * sb);                 // (the word's last mismatch is above sb: not zero)
          const int e = sb + (__builtin_ctzll(above) >> LB);
          note(p0 + sb, p0 + e);
          t = e >= PW ? 0ULL : (t & ~((1ULL << (B * e)) - 1ULL));
        }
      }
      run = PW - 1 - ql;
    }
    if (run >= rmin) { const int e = (whi[part] << SH) < lenM ? (whi[part] << SH) : lenM; note(e - run, e); }   // (a stretch that ends inside a run; the whole string: lenM a multiple of PW)
    }
    for (int j = 0; j < 6; ++j) {
      if (j < nc) { const uint32_t v = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : j == 3 ? c3 : j == 4 ? c4 : c5; run_end((int)(v >> 16), (int)(v & 0xFFFFu)); }
    }
    if (bestE > 0) dev_atomic_max((LC_LDS uint32_t *)outE, (uint32_t)bestE);
    if (bestM > 0) dev_atomic_max((LC_LDS uint32_t *)outM, (uint32_t)bestM);
  }
  WG_SYNC();
}
// The same scan for a string without N (codes 0..3), on two BIT PLANES of the string (bit j of word w of plane q = bit q of the code of
// base 64 w + j): a lane's shift then walks 64 positions per 64-bit word -- mismatch mask = (P0 ^ P0 >> d) | (P1 ^ P1 >> d), one bit per
// position, no gaps to skip in ctz / clz -- instead of 32 (the 2-bit form this replaces) or 16, and a long run is sized without loops:
// the mm + 1 mismatches on either side of it are the top / bottom bits of the 64 positions left / right of the run, cut out of the mask
// words once (the walk back / forward over the words is kept for the run whose neighbourhood holds fewer: a match run of 60).
// MM: max_mismatch at compile time (0..2: the filter walk described inside, the lists of mismatches in registers), or -1: any mm <= 7, the
// walk over match runs of rmin positions (what repeat_scan_min_t does 16 positions per word) on the planes.
// The staging pass raises *bad2 when it meets a code above 3 (string taken from bytes); the caller then runs the 4-bit form
// (repeat_scan_min_t<4>; its B == 2 branches are no longer instantiated).
DEV uint32_t rs_even16(uint32_t v) {                               // the 16 even bits of v, packed
  v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0F0F0F0Fu; v = (v | (v >> 4)) & 0x00FF00FFu; v = (v | (v >> 8)) & 0x0000FFFFu;
  return v;
}
DEV int rs_ctz64(unsigned long long x) { return __builtin_ctzll(x); }
DEV int rs_top64(unsigned long long x) { return 63 - __builtin_clzll(x); }
template <int MM>
DEV void repeat_scan_planes(volatile LC_LDS unsigned long long *rsbuf, LC_GLOBAL const uint8_t *s, int len, int mm_, int rmin, int lf,
                            volatile LC_LDS int *outE, volatile LC_LDS int *outM, const LC_LDS uint32_t *packed2, volatile LC_LDS int *bad2,
                            int ra = 0, int rb = 0x7FFFFFFF) {
  const int mm = MM >= 0 ? MM : mm_;
  const int nwords = len / 64 + 3;
  const bool al4 = (((size_t)s) & 3u) == 0;
  volatile LC_LDS unsigned long long *pl0 = rsbuf, *pl1 = rsbuf + nwords;
  WG_FOR(w, nwords) {
    unsigned long long v0 = 0, v1 = 0;
    if (packed2) {
      const int nw32 = (len + 15) >> 4;
      for (int t = 0; t < 4; ++t) {
        const uint32_t x = 4 * w + t < nw32 ? packed2[4 * w + t] : 0u;
        v0 |= (unsigned long long)rs_even16(x) << (16 * t); v1 |= (unsigned long long)rs_even16(x >> 1) << (16 * t);
      }
    } else if (al4 && 64 * w + 64 <= len) {                        // sixteen aligned 4-byte loads, issued together
      LC_GLOBAL const uint32_t *q = (LC_GLOBAL const uint32_t *)(s + 64 * w);
      uint32_t aa[16]; for (int t = 0; t < 16; ++t) aa[t] = q[t];
      uint32_t any = 0;
      for (int t = 0; t < 16; ++t) {
        any |= aa[t];
        v0 |= (unsigned long long)((((aa[t] & 0x01010101u) * 0x00204081u) >> 21) & 0xFu) << (4 * t);
        v1 |= (unsigned long long)(((((aa[t] >> 1) & 0x01010101u) * 0x00204081u) >> 21) & 0xFu) << (4 * t);
      }
      if (any & 0xFCFCFCFCu) *bad2 = 1;
    } else {
      for (int j = 0; j < 64; ++j) { const int idx = 64 * w + j; if (idx < len) { const uint32_t cde = s[idx]; if (cde > 3u) *bad2 = 1; v0 |= (unsigned long long)(cde & 1u) << j; v1 |= (unsigned long long)((cde >> 1) & 1u) << j; } }
    }
    if (len - 64 * w < 64) { const unsigned long long m = len - 64 * w > 0 ? (1ULL << (len - 64 * w)) - 1ULL : 0ULL; v0 &= m; v1 &= m; }
    pl0[w] = v0; pl1[w] = v1;
  }
  WG_SYNC();
  if (!packed2 && wg_bcast(bad2)) return;                           // a code above 3: the caller runs the 4-bit form
  const LC_LDS unsigned long long *P0 = (const LC_LDS unsigned long long *)pl0, *P1 = (const LC_LDS unsigned long long *)pl1;
  const int nsh = len > 1 ? len - 1 : 0;
  WG_FOR(it, nsh) {
    const int d = it + 1;
    const int lenE = len - 1 - d, lenM = len - d;
    const int dw = d >> 6, sb = d & 63;
    // mismatch mask of word w of this shift: bit j set <=> position 64 w + j mismatches (positions >= lenM read as mismatches)
    auto NE = [&](int w) -> unsigned long long {
      const int p0 = w << 6;
      if (p0 >= lenM) return ~0ULL;
      const int wb = w + dw;
      const unsigned long long b0 = sb ? ((P0[wb] >> sb) | (P0[wb + 1] << (64 - sb))) : P0[wb];
      const unsigned long long b1 = sb ? ((P1[wb] >> sb) | (P1[wb + 1] << (64 - sb))) : P1[wb];
      unsigned long long ne = (P0[w] ^ b0) | (P1[w] ^ b1);
      if (lenM - p0 < 64) ne |= ~0ULL << (lenM - p0);
      return ne;
    };
    // the mask of the 64 positions [pos, pos + 64), pos >= -64: bit i = position pos + i (positions below 0: clear)
    auto ctx = [&](int pos) -> unsigned long long {
      const int w = pos >> 6, sh = pos & 63;
      const unsigned long long lo = w < 0 ? 0ULL : NE(w);
      if (!sh) return lo;
      return (lo >> sh) | (NE(w + 1) << (64 - sh));
    };
    auto prev_mis = [&](int pos) -> int {                        // largest mismatch position < pos, or -1
      int w = (pos - 1) >> 6;
      if (pos <= 0) return -1;
      unsigned long long m = NE(w);
      const int hi = pos - (w << 6);                             // positions [0, hi) of the word
      if (hi < 64) m &= (1ULL << hi) - 1ULL;
      while (true) {
        if (m) return (w << 6) + rs_top64(m);
        if (--w < 0) return -1;
        m = NE(w);
      }
    };
    auto next_mis = [&](int pos) -> int {                        // smallest mismatch position > pos, or lenM
      int w = (pos + 1) >> 6;
      if (pos + 1 >= lenM) return lenM;
      unsigned long long m = NE(w);
      const int lo = pos + 1 - (w << 6);
      if (lo > 0) m &= ~((1ULL << lo) - 1ULL);
      while (true) {
        if (m) { const int q = (w << 6) + rs_ctz64(m); return q < lenM ? q : lenM; }
        if ((++w << 6) >= lenM) return lenM;
        m = NE(w);
      }
    };
    int bestE = 0, bestM = 0;
    auto run_end = [&](int a, int b) {                             // match run [a, b) (MM >= 0: a == b, the start of a window the filter let through)
      constexpr int NP = MM >= 0 ? MM + 1 : 8;
      int pv[NP];                                                  // pv[i]: the (i+1)-th mismatch left of a (-1: none)
      {
        unsigned long long Lm = ctx(a - 64);
        const bool at_start = a - 64 <= 0;
        int q = a; bool inctx = true;
        for (int i = 0; i < NP; ++i) {
          if (i <= mm) {
            if (q >= 0) {
              if (inctx) {
                if (Lm) { const int hb = rs_top64(Lm); Lm &= ~(1ULL << hb); q = a - 64 + hb; }
                else if (at_start) q = -1;
                else { inctx = false; q = prev_mis(a - 64); }
              } else q = prev_mis(q);
            }
            pv[i] = q;
          } else pv[i] = -1;
        }
      }
      {
        unsigned long long Rm = ctx(b);
        const bool at_end = b + 64 >= lenM;
        int q = b - 1; bool inctx = true;
        for (int j = 0; j < NP; ++j) {
          if (j <= mm) {
            if (q < lenM) {
              if (inctx) {
                if (Rm) { const int lb = rs_ctz64(Rm); Rm &= Rm - 1ULL; q = b + lb; if (q > lenM) q = lenM; }
                else if (at_end) q = lenM;
                else { inctx = false; q = next_mis(b + 63); }
              } else q = next_mis(q);
            } else q = lenM;
            if (j == 0) { const int e = (q < lenE ? q : lenE) - (pv[0] + 1); if (e > bestE) bestE = e; }      // the match run between the nearest mismatches either side
            int left = -1;                                         // pv[mm - j]
            for (int i = 0; i < NP; ++i) if (i + j == mm) left = pv[i];
            const int L = q - left - 1; if (L > bestM) bestM = L;
          }
        }
      }
    };
    // The walk only NOTES the long runs (a << 16 | b, up to six per shift in registers); they are sized afterwards, all lanes of
    // the wave together -- sizing a run where it is found would serialise the wave on the lane that found it.
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0; int nc = 0;
    auto note = [&](int a, int b) {
      const uint32_t v = ((uint32_t)a << 16) | (uint32_t)b;
      if (nc == 0) c0 = v; else if (nc == 1) c1 = v; else if (nc == 2) c2 = v; else if (nc == 3) c3 = v; else if (nc == 4) c4 = v; else if (nc == 5) c5 = v;
      else run_end(a, b);
      ++nc;
    };
    if (MM >= 0) {
      // ---- The filter walk.  What counts is a window of >= lminM positions with <= mm mismatches or a match run of >= lminE: both hold a
      // window of lf = min(lminE, lminM, 32) positions with <= mm mismatches.  The number of mismatches among the lf positions from p on is
      // counted for 64 - lf + 1 values of p at a time, bit-sliced: T[k] bit p = "at least k + 1 mismatches in the window at p", windows of
      // 1, 2, 4 ... positions doubled and put together along the bits of lf.  A start p that passes is looked at only if position p - 1
      // mismatches (or p is the first start of the stretch): every maximal window begins behind a mismatch, and inside a long repeat
      // (mismatches are rare there) the starts would otherwise be counted by the hundred.  On random sequence one shift in fifteen has such
      // a start (lf = 11, mm = 2); the run walk below met two or three runs of four matches in every shift and sized each of them.
      constexpr int NL = MM >= 0 ? MM + 1 : 1;
      constexpr int TOP = MM >= 0 ? MM : 0;
      const int SV = 64 - lf + 1;                                    // starts per chunk
      int slo[2] = {0, 0}, shi[2] = {0, lenM};
      if (ra > 0 || rb < lenM) {                                     // starts whose window of lf positions overlaps [ra, rb) in either copy, the lower stretch first
        const int a0 = ra - lf + 1 > 0 ? ra - lf + 1 : 0, a1 = rb < lenM ? rb : lenM, b0 = ra - d - lf + 1 > 0 ? ra - d - lf + 1 : 0, b1 = rb - d < lenM ? rb - d : lenM;
        slo[1] = a0; shi[1] = a1 > a0 ? a1 : a0; slo[0] = b0; shi[0] = b1 > b0 ? b1 : b0;
        if (shi[0] >= slo[1]) { if (slo[0] < slo[1]) slo[1] = slo[0]; if (shi[0] > shi[1]) shi[1] = shi[0]; shi[0] = slo[0]; }      // (they touch: one stretch)
      }
      for (int part = 0; part < 2; ++part) {
        unsigned long long prevbit = 1ULL;                           // does position base - 1 mismatch?  (the first start of a stretch is always looked at)
        for (int base = slo[part]; base < shi[part]; base += SV) {
          unsigned long long x;
          {
            const int wa = base >> 6, sa = base & 63, pb = base + d, wb = pb >> 6, sb2 = pb & 63;
            const unsigned long long a0l = P0[wa], a0h = P0[wa + 1], a1l = P1[wa], a1h = P1[wa + 1], b0l = P0[wb], b0h = P0[wb + 1], b1l = P1[wb], b1h = P1[wb + 1];
            const unsigned long long a0 = sa ? ((a0l >> sa) | (a0h << (64 - sa))) : a0l, a1 = sa ? ((a1l >> sa) | (a1h << (64 - sa))) : a1l;
            const unsigned long long b0 = sb2 ? ((b0l >> sb2) | (b0h << (64 - sb2))) : b0l, b1 = sb2 ? ((b1l >> sb2) | (b1h << (64 - sb2))) : b1l;
            x = (a0 ^ b0) | (a1 ^ b1);
            if (lenM - base < 64) x |= ~0ULL << (lenM - base);
          }
          unsigned long long Pw[NL], Ac[NL];
          auto combine = [&](unsigned long long (&R)[NL], const unsigned long long (&A)[NL], const unsigned long long (&B)[NL], int sh) {
            unsigned long long Bs[NL], O[NL];
            for (int k = 0; k < NL; ++k) Bs[k] = B[k] >> sh;
            for (int k = 0; k < NL; ++k) { unsigned long long r = A[k] | Bs[k]; for (int i = 0; i < k; ++i) r |= A[i] & Bs[k - 1 - i]; O[k] = r; }
            for (int k = 0; k < NL; ++k) R[k] = O[k];
          };
          for (int k = 0; k < NL; ++k) Pw[k] = 0;
          Pw[0] = x;
          bool have = false; int alen = 0;
          for (int bit = 1; ; bit <<= 1) {
            if (lf & bit) { if (!have) { for (int k = 0; k < NL; ++k) Ac[k] = Pw[k]; have = true; alen = bit; } else { combine(Ac, Ac, Pw, alen); alen += bit; } }
            if ((bit << 1) > lf) break;
            combine(Pw, Pw, Pw, bit);
          }
          const int left = shi[part] - base, nv = left < SV ? left : SV;
          unsigned long long cand = ~Ac[TOP] & ((x << 1) | prevbit) & (nv >= 64 ? ~0ULL : ((1ULL << nv) - 1ULL));
          prevbit = (x >> (SV - 1)) & 1ULL;
          while (cand) { const int p = base + rs_ctz64(cand); cand &= cand - 1ULL; note(p, p); }
        }
      }
    }
    const int nw = MM >= 0 ? 0 : (lenM + 63) >> 6;
    // the words to walk: all of them, or (a range was given) those of [ra, rb) and of [ra - d, rb - d), the lower stretch first
    int wlo[2] = {0, 0}, whi[2] = {0, nw};
    if (MM < 0 && (ra > 0 || rb < lenM)) {
      const int a0 = ra > 0 ? ra : 0, a1 = rb < lenM ? rb : lenM, b0 = ra - d > 0 ? ra - d : 0, b1 = rb - d < lenM ? rb - d : lenM;
      wlo[1] = a0 >> 6; whi[1] = a1 > a0 ? (a1 + 63) >> 6 : wlo[1];
      wlo[0] = b0 >> 6; whi[0] = b1 > b0 ? (b1 + 63) >> 6 : wlo[0];
      if (whi[0] >= wlo[1]) { if (wlo[0] < wlo[1]) wlo[1] = wlo[0]; if (whi[0] > whi[1]) whi[1] = whi[0]; whi[0] = wlo[0]; }      // (they touch: one stretch)
      if (whi[1] > nw) whi[1] = nw;
      if (whi[0] > nw) whi[0] = nw;
    }
    int s1 = 1; while (2 * s1 <= rmin) s1 *= 2;                     // (the in-word test for a run of rmin matches: doubling, then the rest)
    for (int part = 0; part < 2; ++part) {
      int run = 0;                                                 // matches ending just before the current word (a run that began before the
                                                                   // stretch counts from the stretch's start: the windows looked for lie inside it)
      for (int w = wlo[part]; w < whi[part]; ++w) {
        const unsigned long long ne = NE(w);
        const int p0 = w << 6;
        if (ne == 0) { run += 64; continue; }
        const int q1 = rs_ctz64(ne), ql = rs_top64(ne);
        if (run + q1 >= rmin) note(p0 - run, p0 + q1);
        if (rmin <= 62 && ql - q1 > rmin) {                        // a run of >= rmin matches between two mismatches of this word?
          unsigned long long t = ~ne;
          for (int h = 1; h < s1; h *= 2) t &= t >> h;
          if (s1 < rmin) t &= t >> (rmin - s1);
          t &= ~((2ULL << q1) - 1ULL);                             // starts after the first mismatch ...
          t &= (1ULL << ql) - 1ULL;                                // ... and before the last one
          // every set bit of t starts rmin matching positions; the lowest one of a group is where a long run begins, the next
          // mismatch above it is where it ends: one trip per LONG run
          while (t) {
            const int sbit = rs_ctz64(t);                                  // run [sbit, e)
            const int e = sbit + rs_ctz64(ne >> sbit);                     // (the word's last mismatch is above sbit: not zero)
            note(p0 + sbit, p0 + e);
            t = e >= 64 ? 0ULL : (t & ~((1ULL << e) - 1ULL));
          }
        }
        run = 63 - ql;
      }
      if (MM < 0 && run >= rmin) { const int e = (whi[part] << 6) < lenM ? (whi[part] << 6) : lenM; note(e - run, e); }   // (a stretch that ends inside a run)
    }
    for (int j = 0; j < 6; ++j) {
      if (j < nc) { const uint32_t v = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : j == 3 ? c3 : j == 4 ? c4 : c5; run_end((int)(v >> 16), (int)(v & 0xFFFFu)); }
    }
    if (bestE > 0) dev_atomic_max((LC_LDS uint32_t *)outE, (uint32_t)bestE);
    if (bestM > 0) dev_atomic_max((LC_LDS uint32_t *)outM, (uint32_t)bestM);
  }
  WG_SYNC();
}
// `bits2`: LDS word the 2-bit form may use as its "met an N" flag; null: 4-bit form at once (a string known to hold N)
DEVNI void repeat_scan_min(volatile LC_LDS unsigned long long *rsbuf, LC_GLOBAL const uint8_t *s, int len, int mm, int lminE, int lminM,
                           volatile LC_LDS int *outE, volatile LC_LDS int *outM, const LC_LDS uint32_t *packed2 = nullptr, volatile LC_LDS int *bits2 = nullptr,
                           int ra = 0, int rb = 0x7FFFFFFF) {
  int rmin = lminE;
  { const int rm = (lminM - mm + mm) / (mm + 1); if (rm < rmin) rmin = rm; }       // ceil((lminM - mm) / (mm + 1))
  if (mm > 7 || mm < 0 || len + 48 > 16 * LC_RS_WORDS || rmin < 3) { repeat_scan(rsbuf, s, len, mm, outE, outM); return; }
  WG_LANE0 { *outE = 0; *outM = 0; if (bits2) *bits2 = 0; }
  WG_SYNC();
  if (bits2) {
    int lf = lminE < lminM ? lminE : lminM; if (lf > 32) lf = 32;
    if (mm == 2 && lf >= 4) repeat_scan_planes<2>(rsbuf, s, len, mm, rmin, lf, outE, outM, packed2, bits2, ra, rb);      // (MAX_MISMATCH's default: the filter walk)
    else if (mm == 1 && lf >= 4) repeat_scan_planes<1>(rsbuf, s, len, mm, rmin, lf, outE, outM, packed2, bits2, ra, rb);
    else if (mm == 0 && lf >= 4) repeat_scan_planes<0>(rsbuf, s, len, mm, rmin, lf, outE, outM, packed2, bits2, ra, rb);
    else repeat_scan_planes<-1>(rsbuf, s, len, mm, rmin, lf, outE, outM, packed2, bits2, ra, rb);
    if (!wg_bcast(bits2)) return;
    WG_LANE0 { *outE = 0; *outM = 0; }
    WG_SYNC();
  }
  repeat_scan_min_t<4>(rsbuf, s, len, mm, rmin, outE, outM, packed2, nullptr, ra, rb);
}

// exclusive prefix sum of a[0..n) in place; returns total in part[LANCET_WG] (default: S.part)
DEV void wg_scan(LC_GLOBAL uint32_t *a, int n, LC_WS &S, volatile LC_LDS uint32_t *part = nullptr) {
  if (!part) part = S.part;
  int chunk = (n + LANCET_WG - 1) / LANCET_WG;
#ifndef LANCET_WAVE_EMU
  // one wave: each lane sums its chunk, the 64 partial sums are scanned with wave shuffles (no LDS round trips)
  WG_SYNC();
  {
    const int l = (int)threadIdx.x;
    int lo = l * chunk, hi = lo + chunk; if (hi > n) hi = n;
    uint32_t s = 0;
    int i4 = lo;                                                      // (four loads in flight per trip: a lane's chunk is a serial walk otherwise)
    for (; i4 + 4 <= hi; i4 += 4) { const uint32_t x0 = ld2(&a[i4]), x1 = ld2(&a[i4 + 1]), x2 = ld2(&a[i4 + 2]), x3 = ld2(&a[i4 + 3]); s += x0 + x1 + x2 + x3; }
    for (int i = i4; i < hi; ++i) s += ld2(&a[i]);
    uint32_t inc = s;
    for (int d = 1; d < LANCET_WG; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, d, LANCET_WG); if (l >= d) inc += t; }
    const uint32_t total = (uint32_t)__shfl((int)inc, LANCET_WG - 1, LANCET_WG);
    uint32_t run = inc - s;
    i4 = lo;
    for (; i4 + 4 <= hi; i4 += 4) {
      const uint32_t x0 = ld2(&a[i4]), x1 = ld2(&a[i4 + 1]), x2 = ld2(&a[i4 + 2]), x3 = ld2(&a[i4 + 3]);
      a[i4] = run; a[i4 + 1] = run + x0; a[i4 + 2] = run + x0 + x1; a[i4 + 3] = run + x0 + x1 + x2; run += x0 + x1 + x2 + x3;
    }
    for (int i = i4; i < hi; ++i) { const uint32_t t = ld2(&a[i]); a[i] = run; run += t; }
    if (l == 0) part[LANCET_WG] = total;
  }
  WG_SYNC();
#else
  WG_FOR(l, LANCET_WG) {
    uint32_t s = 0;
    int lo = l * chunk, hi = lo + chunk; if (hi > n) hi = n;
    for (int i = lo; i < hi; ++i) s += ld2(&a[i]);
    part[l] = s;
  }
  WG_SYNC();
  WG_LANE0 { uint32_t s = 0; for (int l = 0; l < LANCET_WG; ++l) { uint32_t t = part[l]; part[l] = s; s += t; } part[LANCET_WG] = s; }
  WG_SYNC();
  WG_FOR(l, LANCET_WG) {
    uint32_t s = part[l];
    int lo = l * chunk, hi = lo + chunk; if (hi > n) hi = n;
    for (int i = lo; i < hi; ++i) { uint32_t t = ld2(&a[i]); a[i] = s; s += t; }
  }
  WG_SYNC();
#endif
}

// ---------------------------------------------------------------------------------------------------------
// node accessors
// ---------------------------------------------------------------------------------------------------------
DEV bool n_special(const Ctx &c, uint32_t n) { return (LC_CTX(c).W->gr[n].flags & NF_SPECIAL) != 0; }
DEV bool n_dead(const Ctx &c, uint32_t n) { return (LC_CTX(c).W->gr[n].flags & NF_DEAD) != 0; }
DEV int n_len(const Ctx &c, uint32_t n) { return (int)(LC_CTX(c).W->gr[n].seq_hi - LC_CTX(c).W->gr[n].seq_lo); }   // str_m.length()
DEV int n_strlen(const Ctx &c, uint32_t n) { return n_special(c, n) ? 0 : n_len(c, n); }       // Node_t::strlen
DEV float n_totcov(const Ctx &c, uint32_t n) { const float *f = LC_CTX(c).W->gr[n].cov; return f[0] + f[1] + f[2] + f[3]; }

DEV int get_buddy(const Ctx &c, uint32_t n, char dir) {             // Node_t::getBuddy, reference src/Node.cc:235-266
  if (n_special(c, n)) return -1;
  int ret = -1;
  const uint32_t *e = LC_CTX(c).W->gr[n].edges;
  int cnt = (int)LC_CTX(c).W->gr[n].necnt;
  for (int i = 0; i < cnt; ++i) if (is_dir(ED_DIR(e[i]), dir)) { if (ret != -1) return -1; ret = i; }
  if (ret != -1 && ED_TO(e[ret]) == n) return -1;
  return ret;
}
DEV bool is_tandem(const Ctx &c, uint32_t n) {                      // reference src/Node.cc:123-134
  const uint32_t *e = LC_CTX(c).W->gr[n].edges;
  for (int i = 0; i < (int)LC_CTX(c).W->gr[n].necnt; ++i) if (ED_TO(e[i]) == n) return true;
  return false;
}
DEV void add_edge(Ctx &c, uint32_t n, uint32_t to, uint32_t dir) {  // reference src/Node.cc:140-175
  uint32_t *e = LC_CTX(c).W->gr[n].edges;
  int cnt = (int)LC_CTX(c).W->gr[n].necnt;
  for (int i = 0; i < cnt; ++i) if (ED_TO(e[i]) == to && ED_DIR(e[i]) == dir) return;
  if (cnt >= LC_EMAX) { OVF(c); return; }
  e[cnt] = ED_MAKE(to, dir);
  LC_CTX(c).W->gr[n].necnt = cnt + 1;
}
DEV void erase_edge_at(Ctx &c, uint32_t n, int idx) {
  uint32_t *e = LC_CTX(c).W->gr[n].edges;
  int cnt = (int)LC_CTX(c).W->gr[n].necnt;
  for (int i = idx; i + 1 < cnt; ++i) e[i] = e[i + 1];
  LC_CTX(c).W->gr[n].necnt = cnt - 1;
}
DEV void remove_edge(Ctx &c, uint32_t n, uint32_t to, uint32_t dir) {   // reference src/Node.cc:209-229
  uint32_t *e = LC_CTX(c).W->gr[n].edges;
  for (int i = 0; i < (int)LC_CTX(c).W->gr[n].necnt; ++i)
    if (ED_TO(e[i]) == to && ED_DIR(e[i]) == dir) { erase_edge_at(c, n, i); return; }
}
DEV void update_edge(Ctx &c, uint32_t n, uint32_t oldto, uint32_t olddir, uint32_t newto, uint32_t newdir) {   // :181-204
  uint32_t *e = LC_CTX(c).W->gr[n].edges;
  for (int i = 0; i < (int)LC_CTX(c).W->gr[n].necnt; ++i)
    if (ED_TO(e[i]) == oldto && ED_DIR(e[i]) == olddir) { e[i] = ED_MAKE(newto, newdir) | (e[i] & (1u << 30)); return; }
}
DEV void remove_node(Ctx &c, uint32_t n) {                          // Graph_t::removeNode, reference src/Graph.cc:2768-2784
  LC_CTX(c).W->gr[n].flags |= NF_DEAD;
  const uint32_t *e = LC_CTX(c).W->gr[n].edges;
  for (int i = 0; i < (int)LC_CTX(c).W->gr[n].necnt; ++i) {
    uint32_t nn = ED_TO(e[i]);
    if (nn != n) remove_edge(c, nn, n, fliplink(ED_DIR(e[i])));
  }
}

// position data behind a sequence descriptor (cov_t of the reference, src/Ref.hh:41-53)
DEV void desc_cov(const Ctx &c, uint32_t d, int sampleT, uint16_t *fwd, uint16_t *rev, uint16_t *qf, uint16_t *qr) {
  uint32_t km = SD_KMER(d);
  const uint16_t *cn = LC_CTX(c).W->gr[km].kc;
  int o = sampleT ? 0 : 2;
  *fwd = (uint16_t)cn[o]; *rev = (uint16_t)cn[o + 1];
  uint32_t q = LC_CTX(c).W->gr[km].nqv;
  if (q == LC_NIL) { *qf = 0; *qr = 0; return; }
  const uint16_t *qq = LC_CTX(c).W->qv + ((size_t)q * LC_SREF(c).K + SD_OFF(d)) * LC_SREF(c).QS;
  *qf = qq[o]; *qr = qq[o + 1];
}
// lr_mode: hp0 hp1 hp2 and hp0/1/2_minqv behind a descriptor (cov_t::hp*, reference src/Ref.hh:47-52)
DEV void desc_hp(const Ctx &c, uint32_t d, int sampleT, uint16_t *hp3, uint16_t *hpm3) {
  uint32_t km = SD_KMER(d);
  const int o = sampleT ? 0 : 3;
  const uint16_t *h = LC_CTX(c).W->khp + 6 * (size_t)km + o;
  hp3[0] = h[0]; hp3[1] = h[1]; hp3[2] = h[2];
  uint32_t q = LC_CTX(c).W->gr[km].nqv;
  if (q == LC_NIL) { hpm3[0] = hpm3[1] = hpm3[2] = 0; return; }
  const uint16_t *qq = LC_CTX(c).W->qv + ((size_t)q * LC_SREF(c).K + SD_OFF(d)) * LC_SREF(c).QS + 4 + o;
  hpm3[0] = qq[0]; hpm3[1] = qq[1]; hpm3[2] = qq[2];
}
DEV void desc_tot(const Ctx &c, uint32_t d, int *tot, int *totqv) {  // operands of Node_t::computeMinCov
  uint32_t km = SD_KMER(d);
  const uint16_t *cn = LC_CTX(c).W->gr[km].kc;
  *tot = (int)(uint16_t)cn[0] + (int)(uint16_t)cn[1] + (int)(uint16_t)cn[2] + (int)(uint16_t)cn[3];
  uint32_t q = LC_CTX(c).W->gr[km].nqv;
  if (q == LC_NIL) { *totqv = 0; return; }
  const uint16_t *qq = LC_CTX(c).W->qv + ((size_t)q * LC_SREF(c).K + SD_OFF(d)) * LC_SREF(c).QS;
  *totqv = (int)qq[0] + (int)qq[1] + (int)qq[2] + (int)qq[3];
}

// ---------------------------------------------------------------------------------------------------------
// libstdc++ unordered_map order: exact replay of _M_insert_bucket_begin / _M_rehash_aux / _Prime_rehash_policy
// (hashtable.h:1888-1912, 2380-2420, hashtable_policy.h / hashtable_c++0x.cc).
// The prime chain is the subset of __prime_list reachable by growth_factor 2 from the initial 13 buckets.
// ---------------------------------------------------------------------------------------------------------
DEV uint32_t ht_next_prime(uint32_t n) {   // _M_next_bkt(n) for the values that can occur (n = 2*bucket_count or <= 13)
  const uint32_t chain[] = {13u, 29u, 59u, 127u, 257u, 541u, 1109u, 2357u, 5087u, 10273u, 20753u, 42043u,
                            85229u, 172933u, 351061u, 712697u, 1447153u, 2938679u};
  for (int i = 0; i < 18; ++i) if (chain[i] >= n) return chain[i];
  return 0;
}
// hash % bucket_count for the bucket counts of that chain: a 64-bit remainder by a run-time divisor is a ~100-instruction loop
// on this hardware, by a compile-time constant a few multiplies -- and the replay takes one per node and growth stage.
DEV uint32_t ht_mod(unsigned long long h, uint32_t bc) {
  switch (bc) {
    case 13u: return (uint32_t)(h % 13u);       case 29u: return (uint32_t)(h % 29u);       case 59u: return (uint32_t)(h % 59u);
    case 127u: return (uint32_t)(h % 127u);     case 257u: return (uint32_t)(h % 257u);     case 541u: return (uint32_t)(h % 541u);
    case 1109u: return (uint32_t)(h % 1109u);   case 2357u: return (uint32_t)(h % 2357u);   case 5087u: return (uint32_t)(h % 5087u);
    case 10273u: return (uint32_t)(h % 10273u); case 20753u: return (uint32_t)(h % 20753u); case 42043u: return (uint32_t)(h % 42043u);
    default: return (uint32_t)(h % bc);
  }
}
DEV void ht_reset(Ctx &c) { LC_WS &S = LC_SREF(c); S.ht_bc = 1; S.ht_next_resize = 0; S.ht_elt = 0; S.ht_head = LC_NIL; LC_CTX(c).W->ht_bucket[0] = LC_NIL; }
DEVNI void ht_rehash(Ctx &c, uint32_t nb) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  if (nb == 0 || nb > LC_CTX(c).C->bucket_cap) { OVF(c); return; }
  for (uint32_t i = 0; i < nb; ++i) W.ht_bucket[i] = LC_NIL;
  uint32_t p = S.ht_head; S.ht_head = LC_NIL;
  uint32_t bbegin = 0;
  while (p != LC_NIL) {
    uint32_t nxt = W.ht_next[p];
    uint32_t b = ht_mod(W.nhash[p], nb);
    if (W.ht_bucket[b] == LC_NIL) {
      W.ht_next[p] = S.ht_head; S.ht_head = p; W.ht_bucket[b] = LC_BB;
      if (W.ht_next[p] != LC_NIL) W.ht_bucket[bbegin] = p;
      bbegin = b;
    } else {
      uint32_t prev = W.ht_bucket[b];
      if (prev == LC_BB) { W.ht_next[p] = S.ht_head; S.ht_head = p; }
      else { W.ht_next[p] = W.ht_next[prev]; W.ht_next[prev] = p; }
    }
    p = nxt;
  }
  S.ht_bc = nb;
}
DEVNI void ht_insert(Ctx &c, uint32_t n) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  if (S.ht_elt + 1 > S.ht_next_resize) {                      // _Prime_rehash_policy::_M_need_rehash
    unsigned long long mn = S.ht_elt + 1;
    if (S.ht_next_resize == 0 && mn < 11) mn = 11;
    if (mn >= S.ht_bc) {
      unsigned long long want = mn + 1; if (want < 2ULL * S.ht_bc) want = 2ULL * S.ht_bc;
      uint32_t nb = ht_next_prime((uint32_t)want);
      S.ht_next_resize = nb;
      ht_rehash(c, nb);
      if (S.overflow) return;
    } else S.ht_next_resize = S.ht_bc;
  }
  uint32_t b = (uint32_t)(W.nhash[n] % S.ht_bc);
  uint32_t prev = W.ht_bucket[b];
  if (prev != LC_NIL) {
    if (prev == LC_BB) { W.ht_next[n] = S.ht_head; S.ht_head = n; }
    else { W.ht_next[n] = W.ht_next[prev]; W.ht_next[prev] = n; }
  } else {
    W.ht_next[n] = S.ht_head; S.ht_head = n;
    if (W.ht_next[n] != LC_NIL) W.ht_bucket[(uint32_t)(W.nhash[W.ht_next[n]] % S.ht_bc)] = n;
    W.ht_bucket[b] = LC_BB;
  }
  ++S.ht_elt;
}
// Insert into the (array form of the) live table: unordered_map::insert after erasures.  Erase never moves
// other nodes and keeps each bucket's run contiguous, so "bucket empty" == no live node hashes to it.
// unordered_map::insert of a node that the table order has to show (the two special nodes per component), whole wave:
// the place is in front of the first element of the same bucket (else at the head); finding it is a scan of the whole
// order with a 64-bit modulo per element -- one lane took ~1 ms for it.  Call with all lanes; n is uniform.
DEVNI void order_insert(Ctx &c, uint32_t n) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  WG_LANE0 {
    if (S.ht_elt + 1 > S.ht_next_resize) {
      unsigned long long mn = S.ht_elt + 1;
      if (S.ht_next_resize == 0 && mn < 11) mn = 11;
      if (mn >= S.ht_bc) {   // rebuild the linked form from the array, rehash, and read the array back
        unsigned long long want = mn + 1; if (want < 2ULL * S.ht_bc) want = 2ULL * S.ht_bc;
        uint32_t nb = ht_next_prime((uint32_t)want);
        S.ht_head = S.M ? W.order[0] : LC_NIL;
        for (uint32_t i = 0; i < S.M; ++i) W.ht_next[W.order[i]] = (i + 1 < S.M) ? W.order[i + 1] : LC_NIL;
        S.ht_next_resize = nb;
        ht_rehash(c, nb);
        if (!S.overflow) { uint32_t m = 0; for (uint32_t p = S.ht_head; p != LC_NIL; p = W.ht_next[p]) W.order[m++] = p; }
      } else S.ht_next_resize = S.ht_bc;
    }
    S.tmp3 = 0x7FFFFFFF;
  }
  if (wg_bcast(&S.overflow)) return;
  const uint32_t bc = wg_bcastu(&S.ht_bc), M = wg_bcastu(&S.M);
  const uint32_t b = ht_mod(W.nhash[n], bc);
  WG_FOR(i, M) { if (ht_mod(W.nhash[W.order[i]], bc) == b) dev_atomic_min((LC_LDS uint32_t *)&S.tmp3, (uint32_t)i); }
  WG_SYNC();
  uint32_t at = (uint32_t)wg_bcast(&S.tmp3);
  if (at == 0x7FFFFFFFu) at = 0;
  LC_GLOBAL uint32_t *tmp = W.scratch;                     // order[at..M) moves up by one: out, barrier, back
  WG_FOR(i, M - at) { tmp[i] = W.order[at + (uint32_t)i]; }
  WG_SYNC();
  WG_FOR(i, M - at) { W.order[at + 1 + (uint32_t)i] = tmp[i]; }
  WG_LANE0 { W.order[at] = n; ++S.M; ++S.ht_elt; }
}
// cleanDead (reference src/Graph.cc:2737-2762): erase every dead node from the table
DEVNI uint32_t clean_dead(Ctx &c, bool quiet = false) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  uint32_t m = 0, dead = 0;
  for (uint32_t i = 0; i < S.M; ++i) { uint32_t n = W.order[i]; if (W.gr[n].flags & NF_DEAD) ++dead; else W.order[m++] = n; }
  S.M = m; S.ht_elt -= dead;
  if (!quiet) evt(c, EV_CLEANDEAD, dead);
  return dead;
}
DEVNI void print_stats(Ctx &c, int comp, bool every = false) {        // Graph_t::printStats, reference src/Graph.cc:3674-3691
  if (!LC_CTX(c).C->evt_cap) return;                                   // (every: before the components are numbered every node carries 0)
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  int edgecnt = 0, span = 0;
  for (uint32_t i = 0; i < S.M; ++i) { uint32_t n = W.order[i]; if (every || W.gr[n].comp == comp) { edgecnt += W.gr[n].necnt; span += n_strlen(c, n); } }
  evt(c, EV_STATS, comp, S.M, edgecnt, span);
}

// ---------------------------------------------------------------------------------------------------------
// sequence deques
// ---------------------------------------------------------------------------------------------------------
DEV bool seq_reserve(Ctx &c, uint32_t n, uint32_t front, uint32_t back) {   // room for `front` more before, `back` after
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_WS &S = LC_SREF(c);
  uint32_t lo = W.gr[n].seq_lo, hi = W.gr[n].seq_hi;
  if (lo - W.gr[n].seq_clo >= front && W.gr[n].seq_chi - hi >= back) return true;
  uint32_t len = hi - lo;
  uint32_t cap = 2 * (len + front + back) + 16;
  if (S.seq_top + cap > LC_CTX(c).C->seq_cap) { OVF(c); return false; }
  uint32_t nclo = S.seq_top; S.seq_top += cap;
  uint32_t nlo = nclo + (cap - len - front - back) / 2 + front;
  for (uint32_t i = 0; i < len; ++i) W.seq[nlo + i] = W.seq[lo + i];
  W.gr[n].seq_clo = nclo; W.gr[n].seq_chi = nclo + cap; W.gr[n].seq_lo = nlo; W.gr[n].seq_hi = nlo + len;
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// buildgraph (reference src/Graph.cc:530-589 + loadSequence :119-349 + Node.cc / Ref.cc counters)
// ---------------------------------------------------------------------------------------------------------
DEV void read_geom(const Ctx &c, int r, uint32_t *rinfo, uint32_t *bw, uint32_t *gw, int *tlen, bool *isref) {
  LC_GLOBAL const DevBatch &B = *LC_CTX(c).B; const LC_WS &S = LC_SREF(c);
  if (r == S.R - 1) { *isref = true; *tlen = S.reflen; *rinfo = 0; *bw = 0; *gw = 0; return; }
  uint32_t g = B.read_begin[S.w] + (uint32_t)r;
  *isref = false; *rinfo = B.rinfo[g]; *bw = B.base_woff[g]; *gw = B.good_woff[g]; *tlen = (int)RI_TLEN(*rinfo);
}
DEV int read_base(const Ctx &c, bool isref, uint32_t bw, int i) {
  if (isref) return LC_CTX(c).B->ref_codes[LC_CTX(c).B->ref_off[LC_SREF(c).w] + i];
  return rd_base(LC_CTX(c).B->bases, bw, i);
}

// ---- reference k-mers that contain N.  Only the reference pseudo-read can have them; they become
// nodes like any other (they shift the libstdc++ rehash points and bucket order) but never gain coverage.  Their
// identity is the ASCII string, 'N' sorting between 'G' and 'T', rrc('N') == 'N' (reference src/util.cc:204-217).
DEV int n_ord(int code) { return code == 3 ? 4 : (code == 4 ? 3 : code); }          // A C G N T
DEV int n_comp(int code) { return code == 4 ? 4 : 3 - code; }
DEV int nk_char(const uint8_t *ref, int p, int K, bool isR, int j) {                 // j-th char (code) of the canonical string
  return isR ? n_comp(ref[p + K - 1 - j]) : ref[p + j];
}
DEV bool nk_is_forward(const uint8_t *ref, int p, int K) {                           // mer < rc(mer)
  for (int j = 0; j < K; ++j) { int a = n_ord(ref[p + j]), b = n_ord(n_comp(ref[p + K - 1 - j])); if (a != b) return a < b; }
  return false;
}

// ---- the k-mer table: open addressing, one 16-byte slot per entry = tag (u64), first occurrence (u32), node id (u32), so that
// a probe brings in everything the insert needs with one line access
#define SL_TAG(W, i) (*(unsigned long long *)((W).slots + 4 * (size_t)(i)))
#define SL_FIRST(W, i) ((W).slots[4 * (size_t)(i) + 2])
#define SL_NODE(W, i) ((W).slots[4 * (size_t)(i) + 3])

// ---- work items of the per-occurrence passes: one per read, the (long) reference pseudo-read cut into segments of
// LC_SEG k-mer starts so that no lane trails the wave.  items[2i] = read | first k-mer start << IT_RBITS,
// items[2i+1] = sweep offset | number of k-mer starts << 16.  chunk[2c], chunk[2c+1] = sweep origin / length of the
// c-th group of LANCET_WG items.  The sweep offset lines the lanes of a group up on the same genome position
// (reads arrive in coordinate order per sample, so rank * (W - len) / n is a fair estimate of a read's start): at a
// given step the lanes then touch the same few table slots / nodes, which the L2 can coalesce.
DEVNI void build_items(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  WG_LANE0 {
    const uint32_t g0 = LC_CTX(c).B->read_begin[S.w];
    const int nr = S.R - 1;
    int n = 0;
    for (int r = 0; r < nr; ++r) {
      int tlen = (int)RI_TLEN(LC_CTX(c).B->rinfo[g0 + r]);
      if (tlen <= 0) continue;
      W.items[2 * n] = (uint32_t)r; W.items[2 * n + 1] = ((uint32_t)tlen << 16); ++n;
    }
    for (int b0 = 0; b0 < S.reflen; b0 += LC_SEG) {
      int len = S.reflen - b0 < LC_SEG ? S.reflen - b0 : LC_SEG;
      W.items[2 * n] = (uint32_t)nr | ((uint32_t)b0 << IT_RBITS); W.items[2 * n + 1] = ((uint32_t)len << 16); ++n;
    }
    S.nitems = n;
    for (int cb = 0; cb < n; cb += LANCET_WG) {
      uint32_t lo = 0xFFFFu, hi = 0;
      for (int j = cb; j < n && j < cb + LANCET_WG; ++j) { uint32_t e = W.items[2 * j + 1] & 0xFFFFu, l = W.items[2 * j + 1] >> 16; if (e < lo) lo = e; if (e + l > hi) hi = e + l; }
      W.chunk[2 * (cb / LANCET_WG)] = lo; W.chunk[2 * (cb / LANCET_WG) + 1] = hi - lo;
    }
  }
}
#define ITEMS_BEGIN(c, S, W) \
  for (int _cb = 0, _ni = wg_uniform((S).nitems); _cb < _ni; _cb += LANCET_WG) { \
    const int _tlo = (int)(W).chunk[2 * (_cb / LANCET_WG)], _span = (int)(W).chunk[2 * (_cb / LANCET_WG) + 1]; \
    WG_FOR(_j, (_ni - _cb < LANCET_WG ? _ni - _cb : LANCET_WG)) { \
      const uint32_t _i0 = (W).items[2 * (_cb + _j)], _i1 = (W).items[2 * (_cb + _j) + 1]; \
      const int r = (int)(_i0 & ((1u << IT_RBITS) - 1u)), _b0 = (int)(_i0 >> IT_RBITS), _e = (int)(_i1 & 0xFFFFu) - _tlo; \
      uint32_t rinfo, bw, gw; int tlen; bool isref; \
      read_geom(c, r, &rinfo, &bw, &gw, &tlen, &isref); \
      const int nk = tlen - (S).K > 0 ? tlen - (S).K + 1 : 0;   /* a read of exactly K bases has no k-mer: loadSequence needs len > K (Graph.cc:121-124), and occ_base[] allots it no occurrence */ \
      const int pbeg = _b0, pend = (_b0 + (int)(_i1 >> 16) < nk) ? _b0 + (int)(_i1 >> 16) : nk; \
      (void)_e; (void)_span; \
      if (pend <= pbeg) continue;
#define ITEMS_SWEEP(p) for (int _t = 0; _t < _span; ++_t) { const int p = pbeg + _t - _e; if (p < pbeg || p >= pend) continue;
#define ITEMS_END } } }
#define ITEMS_END_NOSWEEP } }

template <int NW>
DEVNI void build_insert_pass(Ctx &c, bool verify) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = S.K;
  const uint32_t mask = S.tmask;
  const uint32_t plimit = mask + 1 < LC_CTX(c).C->table_cap ? 48u : mask;   // a growable table is doubled rather than probed at length
  ITEMS_BEGIN(c, S, W)
    unsigned long long fw[NW], rc[NW];
    for (int w = 0; w < NW; ++w) { fw[w] = 0; rc[w] = 0; }
    const uint32_t obase = W.occ_base[r];
    LC_GLOBAL const uint8_t *refc = LC_CTX(c).B->ref_codes + LC_CTX(c).B->ref_off[S.w];
    int nN = 0;                                                  // N's among the last K bases (reference read only)
    for (int i = pbeg; i < pbeg + K - 1; ++i) {                  // the K-1 bases before the first k-mer's last one
      int b = read_base(c, isref, bw, i);
      if (isref && b > 3) ++nN;
      key_push_fw(fw, NW, K, b & 3);
      key_push_rc(rc, NW, K, b & 3);
    }
    ITEMS_SWEEP(p)
      {
        int b = read_base(c, isref, bw, p + K - 1);
        if (isref) { if (b > 3) ++nN; if (p > pbeg && refc[p - 1] > 3) --nN; }
        key_push_fw(fw, NW, K, b & 3);
        key_push_rc(rc, NW, K, b & 3);
      }
      const uint32_t o = obase + (uint32_t)p;
      const bool nk_ = nN > 0;                                   // this k-mer contains N
      bool isF = nk_ ? nk_is_forward(refc, p, K) : key_less(fw, rc, NW);   // CanonicalMer_t::set: mer < rmer -> F, tie -> R
      const unsigned long long *ck = isF ? fw : rc;
      if (!verify) {
        unsigned long long h = 0;
        uint32_t idx;
        if (nk_) {
          for (int j = 0; j < K; ++j) h = mix64(h * 131ULL + (unsigned long long)(nk_char(refc, p, K, !isF, j) + 1));
          h |= 1ULL << 63;                                       // tag space disjoint from ordinary k-mers
          idx = (uint32_t)mix64(h) & mask;
        } else if (NW == 1 && K <= 31) { h = ck[0] + 1ULL; idx = (uint32_t)mix64(h) & mask; }   // tag == key: exact, no verify pass
        else {
          for (int w = 0; w < NW; ++w) h = mix64(h ^ (ck[w] + 0x9e3779b97f4a7c15ULL * (unsigned long long)(w + 1)));
          h &= ~(1ULL << 63);
          if (h == 0) h = 1;
          idx = (uint32_t)h & mask;
        }
        uint32_t probes = 0;
        while (true) {
          unsigned long long cur = ld2(&SL_TAG(W, idx));
          if (cur == h) break;
          if (cur == 0) {
            unsigned long long old = dev_atomic_cas64(&SL_TAG(W, idx), 0ULL, h);
            if (old == 0) {
              if (nk_) W.slot_key[(size_t)idx * LC_NWMAX] = ((unsigned long long)p << 1) | (isF ? 0ULL : 1ULL);   // where the string lives
              else if (!(NW == 1 && K <= 31)) for (int w = 0; w < NW; ++w) W.slot_key[(size_t)idx * LC_NWMAX + w] = ck[w];   // (k <= 31: the tag is the key + 1)
              break;
            }
            if (old == h) break;
          }
          idx = (idx + 1) & mask;
          if (++probes > plimit) { if (plimit < mask) S.tfull = 1; else OVF(c); break; }
        }
        dev_atomic_min(&SL_FIRST(W, idx), o);
        W.occ[o] = idx | (isF ? 0u : 0x80000000u);
      } else {
        uint32_t idx = W.occ[o] & 0x3FFFFFFFu;
        if (nk_) {
          unsigned long long sk = W.slot_key[(size_t)idx * LC_NWMAX];
          int p2 = (int)(sk >> 1); bool r2 = (sk & 1ULL) != 0;
          for (int j = 0; j < K; ++j) if (nk_char(refc, p, K, !isF, j) != nk_char(refc, p2, K, r2, j)) { OVF(c); break; }
        } else if (K > 31) {
          for (int w = 0; w < NW; ++w) if (W.slot_key[(size_t)idx * LC_NWMAX + w] != ck[w]) OVF(c);   // 64-bit tag collision
        }
      }
  ITEMS_END
  WG_SYNC();
}

// The common case (k <= 31, no N in the window reference) occurrence-major: lane = consecutive occurrence index, the
// k-mer is cut out of the read's packed bases instead of being rolled along the read, so that neighbouring lanes read
// the same words and occ[] is written in whole cache lines.  Same table protocol as build_insert_pass (exact tags).
//   packed bases are little-endian (base j of the k-mer at bits 2j): the reverse-complement key of key_push_rc is the
//   complement of exactly that; the forward key of key_push_fw is the same bases with the 2-bit groups reversed.
DEVNI void build_insert_occ_major(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = S.K;
  const uint32_t mask = S.tmask;
  const uint32_t plimit = mask + 1 < LC_CTX(c).C->table_cap ? 48u : mask;   // a growable table is doubled rather than probed at length
  const unsigned long long kmask = (K == 32) ? ~0ULL : ((1ULL << (2 * K)) - 1ULL);
  const uint32_t refr = (uint32_t)(S.R - 1);
  const uint32_t g0 = LC_CTX(c).B->read_begin[S.w];
  LC_GLOBAL const uint8_t *refc = LC_CTX(c).B->ref_codes + LC_CTX(c).B->ref_off[S.w];
  LC_GLOBAL const uint32_t *occ_base = W.occ_base, *bases = LC_CTX(c).B->bases, *base_woff = LC_CTX(c).B->base_woff + g0;
  LC_GLOBAL uint32_t *slots = W.slots, *occ = W.occ;
  // LC_INS occurrences per lane and trip: keys first, then the first probe of each (one 16-byte load) issued together, then
  // the rest of each probe sequence (the loads of different occurrences overlap; one occurrence is a dependent chain)
  constexpr int LC_INS = 4;
  XG_FOR(l, XG_LANES) {
    const int O = (int)S.O;
    uint32_t rcur = 0;
    for (int o0 = l; o0 < O; o0 += LC_INS * XG_LANES) {
      unsigned long long hh[LC_INS]; uint32_t ix[LC_INS]; bool fF[LC_INS]; lc_u4 sv0[LC_INS];
      for (int u = 0; u < LC_INS; ++u) {
        const int o = o0 + u * XG_LANES;
        hh[u] = 0; ix[u] = 0; fF[u] = false;
        if (o >= O) continue;
        while (rcur < refr && (uint32_t)o >= occ_base[rcur + 1]) ++rcur;
        const int p = (int)((uint32_t)o - occ_base[rcur]);
        unsigned long long v = 0;                                    // base j of the k-mer at bits 2j
        if (rcur == refr) { for (int j = 0; j < K; ++j) v |= (unsigned long long)(refc[p + j] & 3) << (2 * j); }
        else {
          LC_GLOBAL const uint32_t *bp = bases + base_woff[rcur] + (uint32_t)(p >> 4);
          const int sh = (p & 15) * 2;
          const unsigned long long lo = (unsigned long long)bp[0] | ((unsigned long long)bp[1] << 32);
          v = sh ? ((lo >> sh) | ((unsigned long long)bp[2] << (64 - sh))) : lo;
          v &= kmask;
        }
        const unsigned long long rc = (~v) & kmask;
        unsigned long long fw = dev_brev64(v);
        fw = ((fw >> 1) & 0x5555555555555555ULL) | ((fw & 0x5555555555555555ULL) << 1);
        fw >>= (64 - 2 * K);
        fF[u] = fw < rc;                                             // CanonicalMer_t::set: mer < rmer -> F, tie -> R
        hh[u] = (fF[u] ? fw : rc) + 1ULL;                            // tag == key + 1: exact
        ix[u] = (uint32_t)mix64(hh[u]) & mask;
      }
      // A plain (L1-cacheable) 16-byte load: what it returns may be older than the atomics of other lanes, never wrong --
      // tags do not change once set (a stale 0 just sends us into the CAS, which answers with the real tag) and the
      // first-occurrence field only ever decreases (a stale, larger value costs a superfluous atomicMin at worst).
      for (int u = 0; u < LC_INS; ++u) sv0[u] = ldg4(slots + 4 * (size_t)ix[u]);
      for (int u = 0; u < LC_INS; ++u) {
        const int o = o0 + u * XG_LANES;
        if (o >= O) continue;
        const unsigned long long h = hh[u];
        uint32_t idx = ix[u];
        uint32_t probes = 0;
        uint32_t seen_first = LC_NIL;
        lc_u4 sv = sv0[u];
        while (true) {
          const unsigned long long cur = (unsigned long long)sv.x | ((unsigned long long)sv.y << 32);
          if (cur == h) { seen_first = sv.z; break; }
          if (cur == 0) {
            const unsigned long long old = dev_atomic_cas64((LC_GLOBAL unsigned long long *)(slots + 4 * (size_t)idx), 0ULL, h);
            if (old == 0 || old == h) break;
          }
          idx = (idx + 1) & mask;
          if (++probes > plimit) { if (plimit < mask) S.tfull = 1; else OVF(c); break; }
          sv = ldg4(slots + 4 * (size_t)idx);
        }
        if ((uint32_t)o < seen_first) dev_atomic_min(slots + 4 * (size_t)idx + 2, (uint32_t)o);     // most occurrences are not the first one
        occ[o] = idx | (fF[u] ? 0u : 0x80000000u);
      }
    }
  }
  WG_SYNC();
}

// tumor flag condition of loadSequence (reference src/Graph.cc:209-217): step s of a read qualifies when all
// K quals of u and of v pass, i.e. bases s..s+K are all >= MIN_QUAL_CALL.
DEV bool step_all_good(const Ctx &c, bool isref, uint32_t gw, int s, int tlen, int K) {
  if (s < 0 || s >= tlen - K) return false;
  if (isref) return true;                                        // reference quality string is 'K' everywhere
  return good_count(LC_CTX(c).B->good, gw, s, s + K + 1) == K + 1;
}

// --linked-reads: what loadSequence does to ONE node, replayed over the node's occurrences in the order the reference
// visits them (read, then position; reference src/Graph.cc:239-317).  Every occurrence is one "LR event"
// (Node_t::hasBX / addBX / addHP) and, when it is counted (not the reference read, not an overlapping mate), one
// "coverage event" that writes the CURRENT barcode count of the read's strand and the current haplotype counts
// (last writer wins) and bumps hpX_minqv where the stored count had grown.  Only at offset 0 two LR events precede
// the coverage events (u at position 0 and v at position 1): that matters when both are this node.
//   out[0..3] = cov_distr fwd/rev of tumor, normal (barcode counts) ; out[4..6] / out[7..9] = hp0 hp1 hp2 tumor / normal
//   csr bits 29..31 of a counted occurrence = "hpX had grown" (feeds hpX_minqv in the per-position pass)
#define LC_PK(x, i) ((uint32_t)(((x) >> (16 * (i))) & 0xFFFFULL))
#define W_CSR(W) ((LC_GLOBAL cs_t *)(W).csr)       /* layout.h cs_t: 32-bit words, 64-bit in the re-run tier */
DEVNI void lr_node_replay(Ctx &c, uint32_t lo, uint32_t hi, uint32_t *out) {
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_GLOBAL const DevBatch &B = *LC_CTX(c).B; LC_WS &S = LC_SREF(c);
  const uint32_t g0 = B.read_begin[S.w];
  const uint32_t refr = (uint32_t)(S.R - 1);
  LC_GLOBAL cs_t *csr = W_CSR(W);
  for (uint32_t i = lo + 1; i < hi; ++i) {                      // order of the visits
    const cs_t v = csr[i]; const cs_key_t kv = CS_KEY(v);
    uint32_t j = i;
    while (j > lo) { const cs_t u = csr[j - 1]; if (CS_KEY(u) <= kv) break; csr[j] = u; --j; }
    csr[j] = v;
  }
  unsigned long long bxc = 0, covw = 0, hpcT = 0, hpcN = 0, hpwT = 0, hpwN = 0;      // 16-bit fields
  // hasBX is "did an earlier occurrence of this node carry the same barcode in the same sample": one word per occurrence (barcode << 1 |
  // sample; no barcode / the reference read: never equal to anything) written once into the node's stretch of W.mv (idle by now, indexed
  // like the csr), so that the quadratic search is one load per step instead of three dependent ones (round 4: 56 % of a linked-read
  // window's time was spent here).
  LC_GLOBAL uint32_t *bxk = W.mv;
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t r = CS_READ(csr[i]);
    uint32_t key = 0xFFFFFFFEu;
    if (r != refr) { const uint32_t bx = B.bx_rank[g0 + r]; key = bx == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((bx << 1) | RI_NML(B.rinfo[g0 + r])); }
    bxk[i] = key;
  }
  auto lr_event = [&](uint32_t i) {
    const cs_t e = csr[i]; const uint32_t g = g0 + CS_READ(e);
    const uint32_t ri = B.rinfo[g];
    const uint32_t s = RI_NML(ri), d = RI_REV(ri), bx = B.bx_rank[g];
    uint32_t h = B.hp[g]; if (h > 2) h = 2;
    bool seen = false;
    if (bx != 0xFFFFFFFFu) {
      const uint32_t key = (bx << 1) | s;
      uint32_t j = lo;
      for (; j + 4 <= i && !seen; j += 4) { const uint32_t k0 = bxk[j], k1 = bxk[j + 1], k2 = bxk[j + 2], k3 = bxk[j + 3]; seen = k0 == key || k1 == key || k2 == key || k3 == key; }
      for (; j < i && !seen; ++j) seen = bxk[j] == key;
    }
    if (!seen) {
      if (bx != 0xFFFFFFFFu) bxc += 1ULL << (16 * (2 * s + d));
      if (s) hpcN += 1ULL << (16 * h); else hpcT += 1ULL << (16 * h);
    }
  };
  auto cov_event = [&](uint32_t i) {
    const cs_t e = csr[i];
    if (CS_ST(e) != 0) return;
    const uint32_t ri = B.rinfo[g0 + CS_READ(e)];
    const uint32_t s = RI_NML(ri), d = RI_REV(ri);
    const unsigned long long cur = s ? hpcN : hpcT, old = s ? hpwN : hpwT;
    uint32_t grow = 0;
    for (int j = 0; j < 3; ++j) if (LC_PK(old, j) < LC_PK(cur, j)) grow |= 1u << j;
    const int f = (int)(2 * s + d);
    covw = (covw & ~(0xFFFFULL << (16 * f))) | ((unsigned long long)LC_PK(bxc, f) << (16 * f));
    if (s) hpwN = cur; else hpwT = cur;
    csr[i] = e | (cs_t)(grow << 29);
  };
  for (uint32_t i = lo; i < hi; ++i) {
    const cs_t e = csr[i];
    if (CS_READ(e) == refr) continue;                            // BX "null", label REF: no effect (Graph.cc:243-262)
    bool pair = false;
    if (CS_POS(e) == 0 && i + 1 < hi) { const cs_t e2 = csr[i + 1]; pair = CS_READ(e2) == CS_READ(e) && CS_POS(e2) == 1; }
    lr_event(i);
    if (pair) lr_event(i + 1);
    cov_event(i);
    if (pair) { cov_event(i + 1); ++i; }
  }
  for (int q = 0; q < 4; ++q) out[q] = LC_PK(covw, q);
  for (int j = 0; j < 3; ++j) { out[4 + j] = LC_PK(hpwT, j); out[7 + j] = LC_PK(hpwN, j); }
}

// The same replay by the whole wave, for the nodes of LR_COOP_MIN..LR_COOP_MAX occurrences, several nodes at a time (round 4: with one lane
// per node the replay was 44 % of a linked-read window -- a chain of dependent loads per occurrence and a search over the node's
// earlier occurrences for every one of them, 72 nodes per lane; one node at a time by the wave was no better: ~14 us of exposed
// round trips per node).  A batch = consecutive nodes of perm[0..n_many) whose eligible runs fit LR_COOP_MAX staged occurrences; lane j
// takes the j-th staged occurrence, everything a lane needs from the others goes through LDS, all loops stay inside the node's segment:
//   1. the run is sorted into visiting order by rank (keys (read, position) are distinct),
//   2. per occurrence: hasBX key (barcode << 1 | sample), class bits; "counted" = not the reference read, not an overlapping mate,
//   3. seen_j = an earlier occurrence of the node carries the same key (Node_t::hasBX); an occurrence that is not seen ADDS (barcode
//      count of its (sample, strand), haplotype count of its sample),
//   4. a counted occurrence j takes the counts over the occurrences up to E(j), E(j) = j + 1 for the first of a (position 0,
//      position 1) pair of one read -- both LR events of loadSequence's offset 0 precede the coverage events -- else j (counts fit
//      8 bits here); the last counted occurrence of a (sample, strand) / of a sample leaves the node's values,
//   5. "grown" bits against the previous counted occurrence of the same sample.
// Same results as lr_node_replay: the ten values, the csr run sorted, grown bits in csr bits 29..31.
#define LR_COOP_MIN 3u
#define LR_COOP_MAX 192u
#define LR_COOP_SEGS 32u
static_assert(LR_COOP_MAX <= LC_QSTAGE && LR_COOP_MAX < 256u, "staging area ; 8-bit counts");
DEVNI void lr_replay_batches(Ctx &c, uint32_t n_list) {
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_GLOBAL const DevBatch &B = *LC_CTX(c).B; LC_WS &S = LC_SREF(c);
  const uint32_t g0 = B.read_begin[S.w], refr = (uint32_t)(S.R - 1);
  LC_GLOBAL cs_t *csr = W_CSR(W);
  LC_GLOBAL const uint32_t *list = W.scratch;                      // (node, csr start | occurrences << 24) of the nodes build_gather left for the wave
  LC_LDS cs_t *raw = (LC_LDS cs_t *)S.mmeta;                       // [LR_COOP_MAX] the runs as stored ; afterwards win[seg][6]: what the last counted occurrence
  LC_LDS uint32_t *win = (LC_LDS uint32_t *)S.mmeta;               //   of a (sample, strand) [0..3] / of a sample [4..5] leaves: (its place + 1) << 24 | value(s), by atomic max
  LC_LDS cs_t *ev = (LC_LDS cs_t *)S.lbytes;                       // [MAX] in visiting order
  LC_LDS uint32_t *key = (LC_LDS uint32_t *)(ev + LR_COOP_MAX);    // [MAX] hasBX key
  LC_LDS uint32_t *cur = key + LR_COOP_MAX;                        // [MAX] counted: the haplotype counts of its sample at its coverage event (3 x 8 bit) | previous counted one << 24
  LC_LDS uint8_t *meta = (LC_LDS uint8_t *)(cur + LR_COOP_MAX);    // [MAX] 1 adds | 2 barcode present | class f << 2 | haplotype << 4 | 64 counted | 128 first of a pair
  LC_LDS uint8_t *segof = meta + LR_COOP_MAX;                      // [MAX] segment of a staged occurrence
  LC_LDS uint32_t *seg = (LC_LDS uint32_t *)(segof + LR_COOP_MAX); // [SEGS][3] node, csr start, first staged slot | occurrences << 16 | run not in visiting order << 31
  LC_LDS uint32_t *cand_n = (LC_LDS uint32_t *)S.part2;            // the next 64 entries of the list
  LC_LDS uint32_t *cand_lm = (LC_LDS uint32_t *)S.part;
  static_assert(sizeof(cs_t) * LR_COOP_MAX <= sizeof(uint32_t) * LC_QSTAGE && 24u * LR_COOP_SEGS <= sizeof(uint32_t) * LC_QSTAGE &&
                (sizeof(cs_t) + 8u + 2u) * LR_COOP_MAX + 12u * LR_COOP_SEGS <= 16u * LC_QSTAGE, "LDS arrays of the replay");
#define LR_SEG_B0(sg) (seg[3 * (sg) + 2] & 0xFFFFu)
#define LR_SEG_M(sg) ((seg[3 * (sg) + 2] >> 16) & 0x7FFFu)
  uint32_t li = 0, lbase = 0, lend = 0;                            // next entry ; the entries in LDS are [lbase, lend)
  while (li < n_list) {
    if (li >= lend || (lend < n_list && li + 16u > lend)) {        // refill (also when fewer than 16 are left: a batch rarely takes more)
      WG_FOR(t, 64) { if (li + (uint32_t)t < n_list) { cand_n[t] = list[2 * (size_t)(li + (uint32_t)t)]; cand_lm[t] = list[2 * (size_t)(li + (uint32_t)t) + 1]; } }
      lbase = li; lend = li + 64u < n_list ? li + 64u : n_list;
      WG_SYNC();
    }
    WG_LANE0 {                                                     // the batch: entries li .. li + S.tmp2 - 1 = S.tmp3 segments, S.tmp1 staged occurrences
      uint32_t tot = 0, ns = 0, t = li - lbase;
      for (; lbase + t < lend; ++t) {
        const uint32_t mm = cand_lm[t] >> 24;
        if (tot + mm > LR_COOP_MAX || ns == LR_COOP_SEGS) break;
        seg[3 * ns] = cand_n[t]; seg[3 * ns + 1] = cand_lm[t] & 0xFFFFFFu; seg[3 * ns + 2] = tot | (mm << 16);
        ++ns; tot += mm;
      }
      S.tmp1 = (int)tot; S.tmp2 = (int)(lbase + t - li); S.tmp3 = (int)ns;
    }
    const uint32_t tot = (uint32_t)wg_bcast(&S.tmp1), adv = (uint32_t)wg_bcast(&S.tmp2), ns = (uint32_t)wg_bcast(&S.tmp3);
    WG_FOR(t, ns) { const uint32_t b0 = LR_SEG_B0(t), mm = LR_SEG_M(t); for (uint32_t q = 0; q < mm; ++q) segof[b0 + q] = (uint8_t)t; }
    WG_SYNC_LDS();
    WG_FOR(j, tot) {                                               // the runs as stored; a run that is not in visiting order yet is marked (arrival order: nearly always sorted)
      const uint32_t sg = segof[j], b0 = LR_SEG_B0(sg);
      LC_GLOBAL const cs_t *src = csr + seg[3 * sg + 1] + ((uint32_t)j - b0);
      cs_t v = src[0];
      if ((uint32_t)j > b0) { const cs_t vp = src[-1]; if (CS_KEY(vp) > CS_KEY(v)) dev_atomic_or(&seg[3 * sg + 2], 0x80000000u); }
      if (LC_CTX(c).C->debug_stop == 141u) {                       // (test knob: every run arrives backwards -- the emulated lanes fill them in visiting order)
        v = csr[seg[3 * sg + 1] + (LR_SEG_M(sg) - 1u - ((uint32_t)j - b0))]; dev_atomic_or(&seg[3 * sg + 2], 0x80000000u);
      }
      raw[j] = v;
    }
    WG_SYNC_LDS();
    WG_FOR(j, tot) {
      const uint32_t sg = segof[j], b0 = LR_SEG_B0(sg), e0 = b0 + LR_SEG_M(sg);
      const cs_t v = raw[j]; const cs_key_t kv = CS_KEY(v);
      const uint32_t r = CS_READ(v);                               // (the three loads are in flight during the loop below)
      uint32_t k = 0xFFFFFFFEu, mt = 0;
      if (r != refr) {
        const uint32_t g = g0 + r, ri = B.rinfo[g], sm = RI_NML(ri), d = RI_REV(ri), bx = B.bx_rank[g];
        uint32_t h = B.hp[g]; if (h > 2) h = 2;
        k = bx == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((bx << 1) | sm);
        mt = (bx != 0xFFFFFFFFu ? 2u : 0u) | ((2u * sm + d) << 2) | (h << 4) | (CS_ST(v) == 0 ? 64u : 0u);
      }
      uint32_t rank = (uint32_t)j - b0;
      if (seg[3 * sg + 2] >> 31) {
        rank = 0;
#ifndef LANCET_WAVE_EMU
#pragma unroll 4
#endif
        for (uint32_t i = b0; i < e0; ++i) rank += CS_KEY(raw[i]) < kv ? 1u : 0u;
      }
      ev[b0 + rank] = v; key[b0 + rank] = k; meta[b0 + rank] = (uint8_t)mt;
    }
    WG_SYNC_LDS();
    WG_FOR(t, LR_COOP_SEGS * 6u) { win[t] = 0; }                   // (raw[] is done with)
    WG_FOR(j, tot) {
      const uint32_t sg = segof[j], b0 = LR_SEG_B0(sg), e0 = b0 + LR_SEG_M(sg);
      const uint32_t k = key[j];
      uint32_t same = 0;
#ifndef LANCET_WAVE_EMU
#pragma unroll 4
#endif
      for (uint32_t i = b0; i < (uint32_t)j; ++i) same |= key[i] == k ? 1u : 0u;
      uint32_t mt = meta[j];
      if (k != 0xFFFFFFFEu && (k == 0xFFFFFFFFu || !same)) mt |= 1u;           // (the reference read: no LR event at all ; no barcode: hasBX is never true)
      if (k != 0xFFFFFFFEu) { const cs_t e = ev[j]; if (CS_POS(e) == 0 && (uint32_t)j + 1 < e0) { const cs_t e2 = ev[j + 1]; if (CS_READ(e2) == CS_READ(e) && CS_POS(e2) == 1) mt |= 128u; } }
      meta[j] = (uint8_t)mt;
    }
    WG_SYNC_LDS();
    WG_FOR(j, tot) {
      const uint32_t mt = meta[j];
      if (mt & 64u) {
        const uint32_t sg = segof[j], b0 = LR_SEG_B0(sg);
        const uint32_t f = (mt >> 2) & 3u, sm = f >> 1, E = (uint32_t)j + ((mt & 128u) ? 1u : 0u);
        // what is added up to E: barcodes of THIS (sample, strand), haplotypes of THIS sample (3 x 8 bit) ; the counted occurrence of the sample before this one
        uint32_t bxc = 0, hpc = 0, prev = 0xFFu;
        const uint32_t want_b = 3u | (f << 2), want_s = 1u | (sm << 3), cnt_s = 64u | (sm << 3);
#ifndef LANCET_WAVE_EMU
#pragma unroll 4
#endif
        for (uint32_t i = b0; i <= E; ++i) {
          const uint32_t x = meta[i];
          bxc += (x & 15u) == want_b ? 1u : 0u;
          hpc += (x & 9u) == want_s ? (1u << (8u * ((x >> 4) & 3u))) : 0u;
          prev = ((x & 72u) == cnt_s && i < (uint32_t)j) ? i - b0 : prev;
        }
        cur[j] = hpc | (prev << 24);
        const uint32_t place = ((uint32_t)j - b0 + 1u) << 24;      // the last one wins
        dev_atomic_max(&win[6 * sg + f], place | bxc);
        dev_atomic_max(&win[6 * sg + 4 + sm], place | hpc);
      }
    }
    WG_SYNC_LDS();
    WG_FOR(j, tot) {
      const uint32_t mt = meta[j], sg = segof[j], b0 = LR_SEG_B0(sg);
      cs_t e = ev[j];
      if (mt & 64u) {
        const uint32_t now = cur[j], pv = now >> 24;
        const uint32_t old = pv == 0xFFu ? 0u : cur[b0 + pv];
        uint32_t grow = 0;
        for (uint32_t q = 0; q < 3; ++q) if (((old >> (8u * q)) & 0xFFu) < ((now >> (8u * q)) & 0xFFu)) grow |= 1u << q;
        e |= (cs_t)(grow << 29);
      }
      csr[seg[3 * sg + 1] + ((uint32_t)j - b0)] = e;
    }
    WG_FOR(t, ns) {
      const uint32_t n = seg[3 * t];
      LC_GLOBAL NodeGr &G = W.gr[n];
      uint32_t sum = 0;
      for (int q = 0; q < 4; ++q) { const uint32_t v = win[6 * t + q] & 0xFFu; G.kc[q] = (uint16_t)v; sum += v; }
      for (int sm = 0; sm < 2; ++sm) for (int q = 0; q < 3; ++q) W.khp[6 * (size_t)n + 3 * sm + q] = (uint16_t)((win[6 * t + 4 + sm] >> (8 * q)) & 0xFFu);
      G.mincov = (int)sum;
    }
    WG_SYNC_LDS();
    li += adv;
  }
#undef LR_SEG_B0
#undef LR_SEG_M
  WG_SYNC();
}

// buildgraph is cut into separately compiled pieces (DEVNI): one register allocation per phase instead of one for the
// whole window program, which kept values of later phases alive (and spilled) across the hot loops of earlier ones.
DEVNI void build_tables(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W; LC_GLOBAL const EngineCaps &C = *LC_CTX(c).C;
  const int K = S.K;
  (void)C; (void)K; (void)W;
  // ---- occurrence index space: read r owns [occ_base[r], occ_base[r+1]) = its k-mers p = 0..tlen-K
  //      (k-mers per read by all lanes, exclusive scan, one 16-byte record per read for the per-occurrence passes)
  WG_LANE0 { S.tmp1 = 0; S.tmp2 = 0; }
  WG_FOR(r, S.R) {
    uint32_t rinfo, bw, gw; int tlen; bool isref;
    read_geom(c, r, &rinfo, &bw, &gw, &tlen, &isref);
    W.rd[4 * r] = rinfo; W.rd[4 * r + 1] = bw; W.rd[4 * r + 2] = gw;
    W.occ_base[r] = tlen - K > 0 ? (uint32_t)(tlen - K + 1) : 0u;
    if (tlen - K + 1 > 1024) OVF(c);                             // k-mer positions are 10 bits in the occurrence words (layout.h CS_POS): reads of up to 1023 + k bases
    if (tlen > 0 && !isref) dev_atomic_add((LC_LDS uint32_t *)&S.tmp1, (uint32_t)tlen);          // totalreadbp_m (Graph.cc:121-124)
    if (tlen - K > 0) dev_atomic_add((LC_LDS uint32_t *)&S.tmp2, (uint32_t)(tlen - K));
  }
  WG_LANE0 { W.occ_base[S.R] = 0; }
  wg_scan(W.occ_base, S.R + 1, S);
  WG_FOR(r, S.R) { W.rd[4 * r + 3] = W.occ_base[r]; }
  WG_LANE0 {
    const uint32_t o = S.part[LANCET_WG]; const int bp = S.tmp1;
    S.n_kmers += (unsigned long long)(uint32_t)S.tmp2;
    S.O = o; S.totalreadbp = bp;
    if (o > C.occ_cap) OVF(c);
    ++S.n_builds;
    // table size of this build: every slot is cleared and swept twice per build, so it follows the expected number of
    // distinct k-mers (previous build of the window + 25 %, else an eighth of the occurrences + the reference's) at a
    // load <= 2/3 instead of the worst case; a table that fills up is doubled and the insert redone (below).
    const uint32_t est = S.N_last ? S.N_last + S.N_last / 4 : o / 8 + (uint32_t)S.reflen;
    uint32_t tcap = 1024;
    while (tcap < est + est / 2 && tcap < C.table_cap) tcap <<= 1;
    if (C.table_start) tcap = C.table_start < C.table_cap ? C.table_start : C.table_cap;
    S.tmask = tcap - 1; S.tfull = 0;
  }
  if (wg_bcast(&S.overflow)) return;
 again:
  XG_FOR(i, (int)(S.tmask + 1)) { lc_u4 z; z.x = 0; z.y = 0; z.z = LC_NIL; z.w = 0; *(lc_u4 *)(W.slots + 4 * (size_t)i) = z; }
  XG_FOR(i, (int)(S.O / 32 + 2)) { W.bitmap[i] = 0; }
  WG_SYNC();
  PHASE(c, 2);
  // ---- pass 1: canonical k-mers -> open-addressing slots
  if (K <= 31 && !S.hasN) build_insert_occ_major(c);
  else switch (S.NW) { case 1: build_insert_pass<1>(c, false); break; case 2: build_insert_pass<2>(c, false); break;
                       case 3: build_insert_pass<3>(c, false); break; default: build_insert_pass<4>(c, false); break; }
  if (wg_bcast(&S.tfull)) { WG_LANE0 { S.tmask = S.tmask * 2 + 1; S.tfull = 0; } goto again; }
  if (K > 31 || S.hasN) {   // 64-bit tags of longer keys / of N k-mers can collide: compare the full keys
    switch (S.NW) { case 1: build_insert_pass<1>(c, true); break; case 2: build_insert_pass<2>(c, true); break;
                    case 3: build_insert_pass<3>(c, true); break; default: build_insert_pass<4>(c, true); break; }
  }
  if (wg_bcast(&S.overflow)) return;
  PHASE(c, 3);
  STOP_RET(c, 3);
  // ---- dense node ids in first-insertion order (= order of first occurrence, Graph.cc:163-197)
  WG_SYNC_FENCE();   // the insert pass read slots through the L1 while atomics changed them at L2: drop those lines, then whole-slot plain loads
  XG_FOR(l, XG_LANES) {                      // four slots per lane and trip, loads together
    const int T = (int)(S.tmask + 1);
    for (int i0 = l; i0 < T; i0 += 4 * XG_LANES) {
      lc_u4 sv[4];
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * XG_LANES; if (i < T) sv[u] = ldg4(W.slots + 4 * (size_t)i); else { sv[u].x = 0; sv[u].y = 0; sv[u].z = 0; sv[u].w = 0; } }
      for (int u = 0; u < 4; ++u) if ((sv[u].x | sv[u].y) != 0) dev_atomic_or(&W.bitmap[sv[u].z >> 5], 1u << (sv[u].z & 31));
    }
  }
  WG_SYNC();
  int nwords = (int)(S.O / 32 + 1);
  XG_FOR(i, nwords) { W.bitpre[i] = (uint32_t)dev_popc(ld2(&W.bitmap[i])); }
  WG_SYNC();
  wg_scan(W.bitpre, nwords, S);
  WG_LANE0 { S.N = S.part[LANCET_WG]; S.N_last = (uint32_t)S.N; if (S.N > C.node_cap) OVF(c); if (S.N > S.max_nodes) S.max_nodes = S.N; S.sum_nodes += (uint32_t)S.N; S.nspecial = 0; }
  if (wg_bcast(&S.overflow)) return;
  XG_FOR(l, XG_LANES) {
    const int T = (int)(S.tmask + 1), NW = S.NW; const bool hasN = S.hasN != 0;
    for (int i0 = l; i0 < T; i0 += 4 * XG_LANES) {
      lc_u4 sv[4]; uint32_t bp[4], bm[4];
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * XG_LANES; if (i < T) sv[u] = ldg4(W.slots + 4 * (size_t)i); else { sv[u].x = 0; sv[u].y = 0; sv[u].z = 0; sv[u].w = 0; } }
      for (int u = 0; u < 4; ++u) { const bool occd = (sv[u].x | sv[u].y) != 0; const uint32_t f = occd ? sv[u].z : 0u; bp[u] = W.bitpre[f >> 5]; bm[u] = ld2(&W.bitmap[f >> 5]); }
      for (int u = 0; u < 4; ++u) {
        if ((sv[u].x | sv[u].y) == 0) continue;
        const int i = i0 + u * XG_LANES;
        const uint32_t f = sv[u].z;
        const uint32_t id = bp[u] + (uint32_t)dev_popc(bm[u] & ((1u << (f & 31)) - 1u));
        SL_NODE(W, i) = id;
        W.todo[i] = id;                                              // compact copy for the slot -> node pass (todo[] is idle until the mate prefilter)
        const unsigned long long tg = (unsigned long long)sv[u].x | ((unsigned long long)sv[u].y << 32);
        if (NW == 1 && K <= 31 && !(tg >> 63)) W.nkey[(size_t)id * LC_NWMAX] = tg - 1ULL;            // exact tag: key + 1
        else for (int w = 0; w < NW; ++w) W.nkey[(size_t)id * LC_NWMAX + w] = W.slot_key[(size_t)i * LC_NWMAX + w];
        if (hasN) W.gr[id].flags = (tg >> 63) ? NF_NKMER : 0u;        // (without N in the window the gather pass writes the flags whole)
      }
    }
  }
  WG_SYNC();
  // ---- per node: std::hash of the ASCII k-mer, zeroed occurrence counters
  XG_FOR(n, S.N) {
    LC_GLOBAL const unsigned long long *k = W.nkey + (size_t)n * LC_NWMAX;
    if (S.hasN && (W.gr[n].flags & NF_NKMER)) {
      LC_GLOBAL const uint8_t *refc = LC_CTX(c).B->ref_codes + LC_CTX(c).B->ref_off[S.w];
      int p = (int)(k[0] >> 1); bool isR = (k[0] & 1ULL) != 0;
      W.nhash[n] = std_hash_bytes([&](int j) -> int { return (int)"ACGTN"[nk_char(refc, p, K, isR, j)]; }, K);
    } else W.nhash[n] = std_hash_bytes([&](int j) -> int { return (int)"ACGT"[key_base(k, K, j)]; }, K);
    W.nocc[n] = 0; W.nfill[n] = 0;
  }
  // reads whose opposite mate (same name) comes earlier in the window: only these can ever see
  // hasOverlappingMate()==true (reference src/Node.cc:638-661); everything else is counted directly.
  // cand: 0 none, 1 exactly one such earlier read (its index in mate_of), 2 several (same name seen more than once)
  // (the smallest and second smallest read index per (name, mate number), by atomicMin over the name ranks -- a scan over the
  //  earlier reads per read is quadratic in the window's reads: 120 ms for a 12 000-read pile-up)
  {
    const int R1 = S.R - 1;
    const uint32_t g0 = LC_CTX(c).B->read_begin[S.w];
    LC_GLOBAL uint32_t *f1 = W.mv, *f2 = W.mv + 2 * (size_t)S.R;        // (mv[] is idle until build_csr; name ranks are < R)
    WG_FOR(i, 4 * S.R) { W.mv[i] = LC_NIL; }
    WG_SYNC();
    WG_FOR(r, R1) {
      const uint32_t mi = RI_MATE(LC_CTX(c).B->rinfo[g0 + r]);
      if (mi == 1 || mi == 2) dev_atomic_min(&f1[2 * (size_t)LC_CTX(c).B->name_rank[g0 + r] + (mi - 1u)], (uint32_t)r);
    }
    WG_SYNC();
    WG_FOR(r, R1) {
      const uint32_t mi = RI_MATE(LC_CTX(c).B->rinfo[g0 + r]);
      if (mi == 1 || mi == 2) { const size_t sl = 2 * (size_t)LC_CTX(c).B->name_rank[g0 + r] + (mi - 1u); if (ld2(&f1[sl]) != (uint32_t)r) dev_atomic_min(&f2[sl], (uint32_t)r); }
    }
    WG_SYNC();
    WG_FOR(r, S.R) {
      uint8_t cd = 0; uint32_t mo = LC_NIL;
      if (r != R1) {
        const uint32_t mi = RI_MATE(LC_CTX(c).B->rinfo[g0 + r]);
        if (mi == 1 || mi == 2) {
          const size_t sl = 2 * (size_t)LC_CTX(c).B->name_rank[g0 + r] + (2u - mi);    // the opposite mate number's entry
          const uint32_t a1 = ld2(&f1[sl]), a2 = ld2(&f2[sl]);
          if (a1 < (uint32_t)r) { cd = 1; mo = a1; if (a2 < (uint32_t)r) cd = 2; }
        }
      }
      W.cand[r] = cd; W.mate_of[r] = mo;
    }
  }
  WG_LANE0 { S.tmp1 = 0; W.nocc[S.N] = 0; }
  WG_SYNC();
}
DEVNI void build_csr(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W; LC_GLOBAL const EngineCaps &C = *LC_CTX(c).C;
  const int K = S.K;
  (void)C; (void)K; (void)W;
  // ---- pass 2a: occurrences slot -> node id; occurrences per node (the only per-occurrence atomic, on a compact array)
  //      occurrence-major (lane = consecutive occurrence index): occ[] is read and rewritten in whole cache lines
  //      four occurrences per lane and trip, each step issued for all four before the next (the steps of one occurrence
  //      are a chain of dependent round trips; GLOBAL loads of different occurrences overlap)
  XG_FOR(l, XG_LANES) {
    const int O = (int)S.O;
    for (int o0 = l; o0 < O; o0 += 4 * XG_LANES) {
      uint32_t oc[4], X[4], rk[4];
      for (int u = 0; u < 4; ++u) { const int o = o0 + u * XG_LANES; oc[u] = o < O ? W.occ[o] : 0u; }
      for (int u = 0; u < 4; ++u) { const int o = o0 + u * XG_LANES; X[u] = o < O ? W.todo[oc[u] & 0x3FFFFFFFu] : 0u; }   // 4-byte copies of the node ids: a quarter of the slots' footprint
      for (int u = 0; u < 4; ++u) { const int o = o0 + u * XG_LANES; rk[u] = o < O ? dev_atomic_add(&W.nocc[X[u]], 1u) : 0u; }   // arrival rank on the node = place in its csr run
      for (int u = 0; u < 4; ++u) { const int o = o0 + u * XG_LANES; if (o < O) { W.mv[o] = rk[u]; W.occ[o] = X[u] | (oc[u] & 0x80000000u); } }   // (mv[] is idle until the replay)
    }
  }
  WG_SYNC();
  // ---- mate-overlap prefilter: an occurrence of a candidate read can only be suppressed if the node also holds an
  //      occurrence of its earlier opposite mate.  One candidate read at a time, whole wave: the mate's slot ids go into
  //      an open-addressing set in LDS (whole-line reads of its occurrence run), then the read's own occurrences are
  //      probed against it; hits get bit 30 in occ[] and are decided by the exact replay after the csr is built.
#ifdef LANCET_FAT
  // several waves: the candidate reads are listed first (any order: the hits are a set), then LC_PF_G of them share every
  // trip -- one set each in the staging area -- so that a pile-up's ~6000 candidate reads cost ~1000 barrier rounds, not 18000
  {
    LC_LDS uint32_t *set = (LC_LDS uint32_t *)S.mk;
    static_assert(LC_PF_G * 512 <= LC_QSTAGE * 4, "sets live in the staging area");
    const uint32_t g0 = LC_CTX(c).B->read_begin[S.w];
    const int nreads = S.R - 1;
    LC_GLOBAL uint32_t *clist = W.pnodes;                                  // (idle until the replay below)
    WG_LANE0 { S.tmp2 = 0; }
    XG_FOR(r, nreads) {
      if (W.cand[r] && (int)RI_TLEN(LC_CTX(c).B->rinfo[g0 + r]) - K > 0) {
        const uint32_t at = dev_atomic_add((LC_LDS uint32_t *)&S.tmp2, 1u);
        if (at < C.node_cap) clist[at] = (uint32_t)r; else OVF(c);
      }
    }
    if (wg_bcast(&S.overflow)) return;
    const int ncl = wg_bcast(&S.tmp2);
    for (int c0 = 0; c0 < ncl; c0 += LC_PF_G) {
      const int ng = ncl - c0 < LC_PF_G ? ncl - c0 : LC_PF_G;
      WG_FOR(g, ng) {
        const uint32_t r = clist[c0 + g];
        const int tlen = (int)RI_TLEN(LC_CTX(c).B->rinfo[g0 + r]);
        bool all = W.cand[r] == 2;
        uint32_t m0 = 0; int mnk = 0;
        if (!all) {
          const uint32_t mo = W.mate_of[r];
          const int mtl = (int)RI_TLEN(LC_CTX(c).B->rinfo[g0 + mo]);
          mnk = mtl - K > 0 ? mtl - K + 1 : 0;
          if (mnk > 150) { all = true; mnk = 0; } else m0 = W.occ_base[mo];
        }
        S.pf_r[g] = r; S.pf_o0[g] = W.occ_base[r]; S.pf_nk[g] = (uint32_t)(tlen - K + 1); S.pf_m0[g] = m0; S.pf_mnk[g] = (uint32_t)mnk; S.pf_all[g] = all ? 1u : 0u;
      }
      XG_FOR(i, LC_PF_G * 512) { set[i] = 0; }
      WG_SYNC();
      XG_FOR(x, ng * 160) {
        const int g = x / 160, j = x - g * 160;
        if ((uint32_t)j < S.pf_mnk[g]) {
          LC_LDS uint32_t *sg = set + 512 * g;
          const uint32_t id = (W.occ[S.pf_m0[g] + (uint32_t)j] & 0x3FFFFFFFu) + 1u;
          uint32_t h = (id * 2654435761u) >> 23;
          while (true) { const uint32_t old = dev_atomic_cas32(&sg[h], 0u, id); if (old == 0u || old == id) break; h = (h + 1) & 511u; }
        }
      }
      WG_SYNC();
      for (int g = 0; g < ng; ++g) {
        const uint32_t r = S.pf_r[g], o0 = S.pf_o0[g]; const int nk = (int)S.pf_nk[g]; const bool all = S.pf_all[g] != 0;
        volatile LC_LDS uint32_t *sg = (volatile LC_LDS uint32_t *)(set + 512 * g);
        XG_FOR(p, nk) {
          const uint32_t oc = W.occ[o0 + p];
          bool hit = all;
          if (!all) {
            const uint32_t id = (oc & 0x3FFFFFFFu) + 1u;
            uint32_t h = (id * 2654435761u) >> 23;
            while (true) { const uint32_t v = sg[h]; if (v == 0u) break; if (v == id) { hit = true; break; } h = (h + 1) & 511u; }
          }
          if (hit) {
            W.occ[o0 + p] = oc | 0x40000000u;
            uint32_t t = dev_atomic_add((LC_LDS uint32_t *)&S.tmp1, 1u);
            if (t < LC_CTX(c).C->table_cap) W.todo[t] = (r << 10) | (uint32_t)p; else OVF(c);
          }
        }
      }
      WG_SYNC();
    }
  }
#else
  {
    LC_LDS uint32_t *set = (LC_LDS uint32_t *)S.mk;                       // 512 words of the (idle) staging area
    const uint32_t g0 = LC_CTX(c).B->read_begin[S.w];
    const int nreads = S.R - 1;
    for (int r = 0; r < nreads; ++r) {
      const int cd = wg_uniform((int)W.cand[r]);
      if (!cd) continue;
      const int tlen = wg_uniform((int)RI_TLEN(LC_CTX(c).B->rinfo[g0 + r]));
      if (tlen - K <= 0) continue;
      const uint32_t o0 = W.occ_base[r];
      const int nk = tlen - K + 1;
      bool all = cd == 2;
      uint32_t m0 = 0; int mnk = 0;
      if (!all) {
        const uint32_t mo = (uint32_t)wg_uniform((int)W.mate_of[r]);
        const int mtl = wg_uniform((int)RI_TLEN(LC_CTX(c).B->rinfo[g0 + mo]));
        mnk = mtl - K > 0 ? mtl - K + 1 : 0;
        if (mnk > 150) all = true; else m0 = W.occ_base[mo];
      }
      if (!all) {
        WG_FOR(i, 512) { set[i] = 0; }
        WG_SYNC();
        WG_FOR(j, mnk) {
          const uint32_t id = (W.occ[m0 + j] & 0x3FFFFFFFu) + 1u;
          uint32_t h = (id * 2654435761u) >> 23;
          while (true) { const uint32_t old = dev_atomic_cas32(&set[h], 0u, id); if (old == 0u || old == id) break; h = (h + 1) & 511u; }
        }
        WG_SYNC();
      }
      WG_FOR(p, nk) {
        const uint32_t oc = W.occ[o0 + p];
        bool hit = all;
        if (!all) {
          const uint32_t id = (oc & 0x3FFFFFFFu) + 1u;
          uint32_t h = (id * 2654435761u) >> 23;
          while (true) { const uint32_t v = ((volatile LC_LDS uint32_t *)set)[h]; if (v == 0u) break; if (v == id) { hit = true; break; } h = (h + 1) & 511u; }
        }
        if (hit) {
          W.occ[o0 + p] = oc | 0x40000000u;
          uint32_t t = dev_atomic_add((LC_LDS uint32_t *)&S.tmp1, 1u);
          if (t < LC_CTX(c).C->table_cap) W.todo[t] = ((uint32_t)r << 10) | (uint32_t)p; else OVF(c);
        }
      }
      WG_SYNC();
    }
  }
#endif
  WG_SYNC();
  // ---- csr of occurrences by node
  wg_scan(W.nocc, (int)S.N + 1, S);
  {
    // occurrence-major again; the read of an occurrence index is found by walking occ_base[] forward (a lane's indices
    // only grow), its position in the read is the offset from the read's first occurrence
    const uint32_t refr = (uint32_t)(S.R - 1);
    XG_FOR(l, XG_LANES) {
      const int O = (int)S.O;
      uint32_t rcur = 0;
      for (int o0 = l; o0 < O; o0 += 4 * XG_LANES) {        // four occurrences per lane and trip, as above
        uint32_t oc[4], rk[4], at[4], rr[4], pp[4];
        for (int u = 0; u < 4; ++u) { const int o = o0 + u * XG_LANES; oc[u] = o < O ? W.occ[o] : 0u; rk[u] = o < O ? W.mv[o] : 0u; }
        for (int u = 0; u < 4; ++u) {
          const int o = o0 + u * XG_LANES;
          if (o < O) { while (rcur < refr && (uint32_t)o >= W.occ_base[rcur + 1]) ++rcur; rr[u] = rcur; pp[u] = (uint32_t)o - W.occ_base[rcur]; } else { rr[u] = 0; pp[u] = 0; }
          at[u] = o < O ? W.nocc[oc[u] & 0x3FFFFFFFu] : 0u;
        }
        for (int u = 0; u < 4; ++u) {
          const int o = o0 + u * XG_LANES;
          if (o < O) {
            const uint32_t st = rr[u] == refr ? 2u : ((oc[u] & 0x40000000u) ? 1u : 0u);     // the reference read never counts (Graph.cc:265)
            W_CSR(W)[at[u] + rk[u]] = CS_MAKE(rr[u], pp[u], oc[u] >> 31, st);
          }
        }
      }
    }
  }
  WG_SYNC();
  // ---- exact replay for what is left: reproduces std::binary_search over the unsorted vector of opposite-mate names
  //      pushed so far on the node (reference src/Node.cc:638-671).  Per node that holds a flagged
  //      occurrence, once: its csr run sorted into visiting order (read, position) and its two name vectors written out
  //      (mv[]: read << 16 | name rank per push; mate-1 pushes at 4*lo, mate-2 pushes at 4*lo + 2*len).  Per flagged
  //      occurrence: the vector it sees is the prefix pushed by earlier reads (binary search on the read field), then the
  //      reference's lower_bound over the names of that prefix.
  {
    const uint32_t ntodo0 = (uint32_t)wg_bcast(&S.tmp1);
    const uint32_t ntodo = ntodo0 > LC_CTX(c).C->table_cap ? LC_CTX(c).C->table_cap : ntodo0;
    if (ntodo) {
      const uint32_t g0 = LC_CTX(c).B->read_begin[S.w];
      LC_GLOBAL uint32_t *mark = W.bitmap;
      XG_FOR(i, (int)(S.N / 32 + 1)) { mark[i] = 0; }
      WG_SYNC();
      XG_FOR(ti, ntodo) {
        const uint32_t X = W.occ[W.occ_base[W.todo[ti] >> 10] + (W.todo[ti] & 1023u)] & 0x3FFFFFFFu;
        dev_atomic_or(&mark[X >> 5], 1u << (X & 31));
      }
      WG_SYNC();
      WG_LANE0 { S.tmp2 = 0; }
      XG_FOR(n, S.N) { if (ld2(&mark[n >> 5]) & (1u << (n & 31))) W.pnodes[dev_atomic_add((LC_LDS uint32_t *)&S.tmp2, 1u)] = (uint32_t)n; }   // the marked nodes, densely
      WG_SYNC();
      const int nmarked = wg_bcast(&S.tmp2);
      XG_FOR(li, nmarked) {
        const uint32_t n = W.pnodes[li];
        const uint32_t lo = W.nocc[n], hi = W.nocc[n + 1], len = hi - lo;
        LC_GLOBAL cs_t *csr = W_CSR(W);
        for (uint32_t i = lo + 1; i < hi; ++i) {                 // visiting order (the run is nearly sorted already)
          const cs_t v = csr[i]; const cs_key_t kv = CS_KEY(v);
          uint32_t j = i;
          while (j > lo) { const cs_t u = csr[j - 1]; if (CS_KEY(u) <= kv) break; csr[j] = u; --j; }
          csr[j] = v;
        }
        uint32_t n1 = 0, n2 = 0;
        LC_GLOBAL mv_t *v1 = (LC_GLOBAL mv_t *)W.mv + 4 * (size_t)lo, *v2 = v1 + 2 * (size_t)len;      // (mv_t: layout.h; each occurrence pushes at most twice)
        for (uint32_t i = lo; i < hi; ++i) {
          const cs_t e = csr[i]; const uint32_t er = CS_READ(e);
          if ((int)er == S.R - 1) continue;
          const uint32_t ri = LC_CTX(c).B->rinfo[g0 + er], mt = RI_MATE(ri);
          if (mt != 1 && mt != 2) continue;
          const int ep = (int)CS_POS(e), etl = (int)RI_TLEN(ri);
          const mv_t rec = MV_REC(er, LC_CTX(c).B->name_rank[g0 + er]);                      // (ranks < number of reads)
          const uint32_t pushes = (ep >= 1 ? 1u : 0u) + (ep <= etl - K - 1 ? 1u : 0u);   // as v of step p-1, as u of step p
          for (uint32_t q = 0; q < pushes; ++q) { if (mt == 1) v1[n1++] = rec; else v2[n2++] = rec; }
        }
#if LC_WIDE_IDS
        W.nfill[2 * (size_t)n] = n1; W.nfill[2 * (size_t)n + 1] = n2;
#else
        W.nfill[n] = n1 | (n2 << 16);
#endif
      }
      WG_SYNC();
      XG_FOR(ti, ntodo) {
        const uint32_t r = W.todo[ti] >> 10, p = W.todo[ti] & 1023u;
        const uint32_t mi = RI_MATE(LC_CTX(c).B->rinfo[g0 + r]), nm = MV_NAME(MV_REC(0u, LC_CTX(c).B->name_rank[g0 + r]));
        const uint32_t X = W.occ[W.occ_base[r] + p] & 0x3FFFFFFFu;
        const uint32_t lo = W.nocc[X], len = W.nocc[X + 1] - lo;
        LC_GLOBAL const mv_t *vec = (LC_GLOBAL const mv_t *)W.mv + 4 * (size_t)lo + (mi == 1 ? 2 * (size_t)len : 0);   // the OTHER mate's pushes
#if LC_WIDE_IDS
        const uint32_t nv = mi == 1 ? W.nfill[2 * (size_t)X + 1] : W.nfill[2 * (size_t)X];
#else
        const uint32_t nv = mi == 1 ? (W.nfill[X] >> 16) : (W.nfill[X] & 0xFFFFu);
#endif
        uint32_t total = 0;                                       // pushes of reads before r
        { uint32_t f = 0, l = nv; while (l > 0) { const uint32_t h = l >> 1; if (MV_READ(vec[f + h]) < r) { f += h + 1; l -= h + 1; } else l = h; } total = f; }
        uint32_t first = 0, l2 = total;                           // std::lower_bound over the names, as pushed
        while (l2 > 0) { const uint32_t h = l2 >> 1, mid = first + h; if (MV_NAME(vec[mid]) < nm) { first = mid + 1; l2 = l2 - h - 1; } else l2 = h; }
        const bool ovl = (first != total) && !(nm < MV_NAME(vec[first]));
        uint32_t f = lo, l = len;                                 // the occurrence's own entry in the sorted run
        const cs_key_t key = CS_KEY_OF(r, p);
        LC_GLOBAL cs_t *csr = W_CSR(W);
        while (l > 0) { const uint32_t h = l >> 1; const cs_t u = csr[f + h]; if (CS_KEY(u) < key) { f += h + 1; l -= h + 1; } else l = h; }
        const cs_t e = csr[f];
        csr[f] = CS_MAKE(CS_READ(e), CS_POS(e), CS_ORI(e), ovl ? 2u : 0u);
      }
    }
    WG_LANE0 { S.seq_top = 0; S.qv_top = 0; }
  }
  WG_SYNC();
}
// Even k: a k-mer can be its own reverse complement, and CanonicalMer_t::set gives such a k-mer the orientation R wherever it
// occurs (tie -> R, reference src/Mer.hh:57-71).  For an ordinary neighbour the (side, extension base) slot of a node fixes the
// edge (target, direction) whichever strand the step was read on -- the target then appears reverse-complemented and the
// reciprocal edge of loadSequence flips its orientation back (reference src/Graph.cc:320-347).  A self-complementary target
// shows R on both strands, so the two ways of meeting it (this node as the step's u, or as its v) give two different edges
// (e.g. RR and RF) that Node_t::addEdge keeps apart (it compares node and direction, src/Node.cc:140-175).  Hence for even k
// the first-seen stamps are kept per (slot, role of this node in the step), the edges are resolved, and equal
// (target, direction) pairs are merged under their earliest stamp.  Lane-local; only called when K is even.
DEVNI void gather_edges_even(Ctx &c, uint32_t n, uint32_t lo, uint32_t hi) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = S.K;
  const int Rref = S.R - 1, reflen_ = S.reflen;
  uint32_t ef[20];
  for (int j = 0; j < 20; ++j) ef[j] = LC_NIL;
  for (uint32_t q = lo; q < hi; ++q) {
    const cs_t e = W_CSR(W)[q];
    const int r = (int)CS_READ(e), p = (int)CS_POS(e);
    const uint32_t ori = CS_ORI(e);
    const bool isref = r == Rref;
    const lc_u4 rdv = *(const lc_u4 *)(W.rd + 4 * (size_t)r);
    const uint32_t bw = rdv.y;
    const int tlen = isref ? reflen_ : (int)RI_TLEN(rdv.x);
    const int nk = tlen - K + 1;
    const uint32_t o0 = rdv.w;
    if (p + 1 < nk) {
      const int bnew = read_base(c, isref, bw, p + K);
      const int eb = (ori == 0) ? bnew : n_comp(bnew);
      const uint32_t side = (ori == 0) ? 0u : 1u;
      const uint32_t sl = eb < 4 ? side * 4u + (uint32_t)eb : 8u + side, st = 2u * (o0 + (uint32_t)p);
      if (st < ef[sl]) ef[sl] = st;
    }
    if (p > 0) {
      const int bold = read_base(c, isref, bw, p - 1);
      const int eb = (ori == 0) ? bold : n_comp(bold);
      const uint32_t side = (ori == 0) ? 1u : 0u;
      const uint32_t sl = 10u + (eb < 4 ? side * 4u + (uint32_t)eb : 8u + side), st = 2u * (o0 + (uint32_t)p - 1u) + 1u;
      if (st < ef[sl]) ef[sl] = st;
    }
  }
  uint32_t stamp[20], edge[20]; int ne = 0;
  for (int j = 0; j < 20; ++j) {
    if (ef[j] == LC_NIL) continue;
    const uint32_t s = ef[j] >> 1;
    const uint32_t a = W.occ[s], b = W.occ[s + 1];
    const uint32_t ua = a >> 31, ub = b >> 31;
    const uint32_t ew = (ef[j] & 1u) == 0 ? ED_MAKE(b & 0x3FFFFFFFu, ua == 0 ? (ub == 0 ? 0u : 1u) : (ub == 0 ? 2u : 3u))
                                          : ED_MAKE(a & 0x3FFFFFFFu, ua == 0 ? (ub == 0 ? 3u : 1u) : (ub == 0 ? 2u : 0u));
    int at = -1;
    for (int i = 0; i < ne; ++i) if (edge[i] == ew) { at = i; break; }
    if (at >= 0) { if (ef[j] < stamp[at]) stamp[at] = ef[j]; }
    else { edge[ne] = ew; stamp[ne] = ef[j]; ++ne; }
  }
  for (int i = 1; i < ne; ++i) { const uint32_t s = stamp[i], ew = edge[i]; int j = i; while (j > 0 && stamp[j - 1] > s) { stamp[j] = stamp[j - 1]; edge[j] = edge[j - 1]; --j; } stamp[j] = s; edge[j] = ew; }
  if (ne > LC_EMAX) { OVF(c); ne = LC_EMAX; }
  LC_GLOBAL NodeGr &G = W.gr[n];
  for (int i = 0; i < ne; ++i) G.edges[i] = edge[i];
  G.necnt = (uint32_t)ne;
}
DEVNI void build_gather(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W; LC_GLOBAL const EngineCaps &C = *LC_CTX(c).C;
  const int K = S.K;
  (void)C; (void)K; (void)W;
  // ---- per node, gathering over its occurrences (reference src/Graph.cc:163-349): colours, counted occurrences per
  //      strand/sample, edges in first-seen order (stamp = 2*occurrence index of the step's u, +1 for the v side),
  //      float coverages.  Everything lands in the node's own record: no scattered updates.
  const double avgcov = ((double)S.totalreadbp) / ((double)S.reflen);
  // A lane walks all occurrences of its node, and a wave step lasts as long as its busiest lane: visit the nodes grouped
  // by occurrence count (> 24, 3..24, <= 2) so that the ~85 % one-occurrence (sequencing error) k-mers do not wait for
  // the high-coverage ones.  perm[] = node ids in that order (two exclusive scans).
  {
    LC_GLOBAL uint32_t *fa = W.order, *fb = W.scratch, *perm = W.ht_next;
    XG_FOR(n, S.N) { const uint32_t cn = W.nocc[n + 1] - W.nocc[n]; fa[n] = cn > 24u; fb[n] = (cn > 2u && cn <= 24u); }
    WG_LANE0 { fa[S.N] = 0; fb[S.N] = 0; }
    wg_scan(fa, (int)S.N + 1, S);
    wg_scan(fb, (int)S.N + 1, S);
    XG_FOR(n, S.N) {
      const uint32_t a = fa[n], b = fb[n], ta = fa[S.N], tb = fb[S.N];
      const bool ia = fa[n + 1] != a, ib = fb[n + 1] != b;
      perm[ia ? a : (ib ? ta + b : ta + tb + ((uint32_t)n - a - b))] = (uint32_t)n;
    }
    WG_SYNC();
    WG_LANE0 { S.tmp2 = 0; }                                      // linked reads: nodes listed for lr_replay_batches
  }
  SUBPHASE(c, 5, 11);
  XG_FOR(pi, S.N) {
    const int n = (int)W.ht_next[pi];
    uint32_t ef0 = LC_NIL, ef1 = LC_NIL, ef2 = LC_NIL, ef3 = LC_NIL, ef4 = LC_NIL, ef5 = LC_NIL, ef6 = LC_NIL, ef7 = LC_NIL, ef8 = LC_NIL, ef9 = LC_NIL;
#define LC_EFMIN(sl, st) do { uint32_t _s = (sl), _v = (st); \
      ef0 = (_s == 0 && _v < ef0) ? _v : ef0; ef1 = (_s == 1 && _v < ef1) ? _v : ef1; ef2 = (_s == 2 && _v < ef2) ? _v : ef2; \
      ef3 = (_s == 3 && _v < ef3) ? _v : ef3; ef4 = (_s == 4 && _v < ef4) ? _v : ef4; ef5 = (_s == 5 && _v < ef5) ? _v : ef5; \
      ef6 = (_s == 6 && _v < ef6) ? _v : ef6; ef7 = (_s == 7 && _v < ef7) ? _v : ef7; ef8 = (_s == 8 && _v < ef8) ? _v : ef8; \
      ef9 = (_s == 9 && _v < ef9) ? _v : ef9; } while (0)
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, fl = 0;
    bool onref = false;                                            // a k-mer of the reference pseudo-read (Ref_t's tables ask for its counts)
    const uint32_t lo = W.nocc[n], hi = W.nocc[n + 1];
    const int Rref = S.R - 1, reflen_ = S.reflen;
#ifndef LANCET_WAVE_EMU
#pragma unroll 4
#endif
    for (uint32_t q = lo; q < hi; ++q) {
      const cs_t e = W_CSR(W)[q];
      const int r = (int)CS_READ(e), p = (int)CS_POS(e);
      const uint32_t ori = CS_ORI(e), st = CS_ST(e);
      // (read_geom without its volatile LDS reads, so that the loads of the unrolled iterations can be issued together)
      const bool isref = r == Rref;
      onref = onref || isref;
      const lc_u4 rdv = *(const lc_u4 *)(W.rd + 4 * (size_t)r);                 // rinfo, packed-base offset, quality-mask offset, first occurrence
      const uint32_t rinfo = rdv.x, bw = rdv.y, gw = rdv.z;
      const int tlen = isref ? reflen_ : (int)RI_TLEN(rinfo);
      const int nk = tlen - K + 1;
      const uint32_t o0 = rdv.w;
      if (!isref) {
        if (RI_NML(rinfo)) fl |= NF_NORMAL;
        else if (!(fl & NF_TUMOR) && (step_all_good(c, isref, gw, p, tlen, K) || step_all_good(c, isref, gw, p - 1, tlen, K))) fl |= NF_TUMOR;
        if (st == 0) { const int ctr = (RI_NML(rinfo) ? 2 : 0) + (RI_REV(rinfo) ? 1 : 0); c0 += ctr == 0; c1 += ctr == 1; c2 += ctr == 2; c3 += ctr == 3; }
      }
      // edge slots: (side, extension base in the node's canonical orientation); F side = right extension.
      // index: ACGT -> side*4 + base, N -> 8 + side            (Graph.cc:320-347 for the directions)
      if (p + 1 < nk) {            // step p: this node is u, the next k-mer v
        int bnew = read_base(c, isref, bw, p + K);            // base that v adds after u
        int eb = (ori == 0) ? bnew : n_comp(bnew);
        uint32_t side = (ori == 0) ? 0u : 1u;
        LC_EFMIN(eb < 4 ? side * 4u + (uint32_t)eb : 8u + side, 2u * (o0 + (uint32_t)p));
      }
      if (p > 0) {                 // step p-1: the previous k-mer is u, this node v
        int bold = read_base(c, isref, bw, p - 1);           // base that u has before v
        int eb = (ori == 0) ? bold : n_comp(bold);
        uint32_t side = (ori == 0) ? 1u : 0u;
        LC_EFMIN(eb < 4 ? side * 4u + (uint32_t)eb : 8u + side, 2u * (o0 + (uint32_t)p - 1u) + 1u);
      }
    }
#undef LC_EFMIN
    uint32_t stamp[10]; int ne = 0;
    { const uint32_t ef[10] = {ef0, ef1, ef2, ef3, ef4, ef5, ef6, ef7, ef8, ef9};
      for (int j = 0; j < 10; ++j) if (ef[j] != LC_NIL) stamp[ne++] = ef[j]; }
    for (int i = 1; i < ne; ++i) { uint32_t s = stamp[i]; int j = i; while (j > 0 && stamp[j - 1] > s) { stamp[j] = stamp[j - 1]; --j; } stamp[j] = s; }
    LC_GLOBAL NodeGr &G = W.gr[n];
    // first removeLowCov predicate (reference src/Graph.cc:2790-2827, docompression=false, compid=0): minqv <= T.
    // minqv cannot exceed the number of counted occurrences, so most nodes (sequencing-error k-mers) are decided here;
    // the others get their per-position counts from the whole wave below.  A node that is certain to go needs no edge
    // targets (only the edge COUNT shows up, in the trace's graph statistics): skip the look-ups of its neighbours.
    const uint32_t counted = c0 + c1 + c2 + c3;
    const float tt = (float)c0 + (float)c1, tn = (float)c2 + (float)c3;
    const bool low = ((int)counted <= LC_CTX(c).P->low_cov_threshold) || ((double)counted <= (LC_CTX(c).P->min_cov_ratio * avgcov)) || (tt == 1.0f && tn == 1.0f);
    if (!low) for (int i = 0; i < ne; ++i) {
      const uint32_t s = stamp[i] >> 1;
      const uint32_t a = W.occ[s], b = W.occ[s + 1];           // u and v of that step
      const uint32_t ua = a >> 31, ub = b >> 31;
      if ((stamp[i] & 1u) == 0) G.edges[i] = ED_MAKE(b & 0x3FFFFFFFu, ua == 0 ? (ub == 0 ? 0u : 1u) : (ub == 0 ? 2u : 3u));   // FF FR RF RR
      else G.edges[i] = ED_MAKE(a & 0x3FFFFFFFu, ua == 0 ? (ub == 0 ? 3u : 1u) : (ub == 0 ? 2u : 0u));                          // RR FR RF FF
    }
    G.necnt = (uint32_t)ne; G.comp = 0; G.color = 0; G.onref = 0; G.nkm = 1; G.nkmT = 0; G.nqv = LC_NIL;
    if ((K & 1) == 0) gather_edges_even(c, (uint32_t)n, lo, hi);   // self-complementary neighbours: edges per (slot, role), merged by (target, direction)
    G.flags = (S.hasN ? (G.flags & NF_NKMER) : 0u) | fl;
    uint16_t *kc = G.kc;                     // counts per strand/sample as the cov_t fields hold them (unsigned short)
    kc[0] = (uint16_t)c0; kc[1] = (uint16_t)c1; kc[2] = (uint16_t)c2; kc[3] = (uint16_t)c3;
    G.cov[0] = (float)c0; G.cov[1] = (float)c1; G.cov[2] = (float)c2; G.cov[3] = (float)c3;
    if (S.LR && low && !onref) {       // removed by the first removeLowCov and not a reference k-mer: nobody reads its barcode / haplotype counts
      for (int q = 0; q < 6; ++q) W.khp[6 * (size_t)n + q] = 0;
    } else if (S.LR) {        // cov_distr holds barcode counts instead of read counts; the float coverages stay read counts
      if (hi - lo >= LR_COOP_MIN && hi - lo <= LR_COOP_MAX && lo < (1u << 24)) {     // by the whole wave, below: listed
        const uint32_t at = dev_atomic_add((LC_LDS uint32_t *)&S.tmp2, 1u);
        W.scratch[2 * (size_t)at] = (uint32_t)n; W.scratch[2 * (size_t)at + 1] = lo | ((hi - lo) << 24);
      } else {
        uint32_t lrv[10];
        lr_node_replay(c, lo, hi, lrv);
        for (int q = 0; q < 4; ++q) kc[q] = (uint16_t)lrv[q];
        for (int q = 0; q < 6; ++q) W.khp[6 * (size_t)n + q] = (uint16_t)lrv[4 + q];
      }
    }
    G.mincov = (int)(uint16_t)kc[0] + (int)(uint16_t)kc[1] + (int)(uint16_t)kc[2] + (int)(uint16_t)kc[3];
    G.mincovqv = 0;
    W.order[n] = low ? 0u : 1u;
  }
  WG_LANE0 { W.order[S.N] = 0; }
  WG_SYNC();
  SUBPHASE(c, 5, 12);
  if (S.LR) lr_replay_batches(c, (uint32_t)wg_bcast(&S.tmp2));     // (the nodes listed above)
  SUBPHASE(c, 5, 13);
  wg_scan(W.order, (int)S.N + 1, S);
  // the candidates with their csr range next to them (compact arrays for the group formation of the per-position pass)
  XG_FOR(n, S.N) { if (W.order[n + 1] != W.order[n]) { const uint32_t at = W.order[n]; W.pnodes[at] = (uint32_t)n; W.pedges[at] = W.nocc[n]; W.ht_bucket[at] = W.nocc[n + 1] - W.nocc[n]; } }
  WG_SYNC();
}
// rows_n > 0 ("rows only", kernels.h load_prebuilt_lr): the list holds rows_n nodes whose fate is known -- the survivors of a graph the LDS
// build kernel made, in --linked-reads mode -- and W.ht_start[x] names the quality row of entry x; only the rows are written (all ten
// counters), no predicate, no record, no descriptor.
DEVNI void build_qcounts(Ctx &c, uint32_t rows_n = 0) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W; LC_GLOBAL const EngineCaps &C = *LC_CTX(c).C;
  const int K = S.K;
  (void)C; (void)K; (void)W;
  const bool rows = rows_n != 0;
  const double avgcov = ((double)S.totalreadbp) / ((double)S.reflen);
  // ---- undecided nodes: number of counted reads whose base passes MIN_QUAL_CALL, per k-mer position and strand/sample
  //      (Node_t::updateCovDistr minqv_fwd/minqv_rev, reference src/Node.cc:470-497).  Up to LC_PACK consecutive
  //      candidates are handled together as long as their occurrences fit the LDS staging area (a bigger one goes alone,
  //      in rounds): the dependent chain occurrence -> read info -> quality words is paid once per group.
  //      Step 1, lane = occurrence (two per lane, loads issued together): K quality bits + class of the occurrence -> LDS.
  //      Step 2, lane = (candidate, k-mer position): count over the candidate's staged occurrences out of LDS.
  const uint32_t ncand = rows ? rows_n : wg_bcastu(&S.part[LANCET_WG]);
  const int QS = S.QS; const bool LR = S.LR != 0;
  uint32_t ci = 0, cqb = 0;
  bool cq_valid = false;
  while (ci < ncand) {
    if (!cq_valid || ci + LC_PACK > cqb + 64) {  // candidate list a line at a time into LDS: group formation is a lane-0 chain
      WG_FOR(t, 64) {
        const uint32_t x = ci + (uint32_t)t;
        if (x < ncand) { S.cq_n[t] = W.pnodes[x]; S.cq_lo[t] = W.pedges[x]; S.cq_cnt[t] = W.ht_bucket[x]; S.cq_row[t] = rows ? W.ht_start[x] : 0u; }
      }
      WG_SYNC();
      cqb = ci; cq_valid = true;
    }
    WG_LANE0 {                                   // group formation
      uint32_t gN = 0, tot = 0;
      for (uint32_t k = 0; k < LC_PACK && ci + k < ncand; ++k) {
        const uint32_t q = ci - cqb + k;
        const uint32_t n = S.cq_n[q], lo = S.cq_lo[q], cnt = S.cq_cnt[q];
        // (a candidate with more than 65 535 occurrences: the reference's per-position counters are unsigned short and wrap -- src/Ref.hh:43-52,
        //  ++ in Node_t::updateCovDistr -- and so do these: its rounds below add into 16-bit fields, round by round; until round 5 such a
        //  window was reported LANCET_W_OVERFLOW)
        if (gN > 0 && tot + cnt > LC_QSTAGE) break;
        S.g_n[gN] = n; S.g_lo[gN] = lo; S.g_cnt[gN] = cnt; S.g_es[gN] = tot; S.g_min[gN] = 0x7FFFFFFFu;
        S.g_row[gN] = rows ? S.cq_row[q] : S.qv_top + gN;
        if (rows && ((size_t)S.cq_row[q] + 1u) * (size_t)K > (size_t)LC_CTX(c).C->qv_cap) OVF(c);
        ++gN; tot += cnt;
        if (tot >= LC_QSTAGE) break;
      }
      S.g_N = gN;
      if (!rows && (S.qv_top + gN > LC_CTX(c).C->surv_cap || ((size_t)S.qv_top + gN) * (size_t)K > (size_t)LC_CTX(c).C->qv_cap)) OVF(c);
    }
    if (wg_bcast(&S.overflow)) return;
    const int gN = wg_uniform((int)S.g_N);
    const uint32_t qi0 = (uint32_t)wg_uniform((int)S.qv_top);
    const uint32_t cnt0 = (uint32_t)wg_uniform((int)S.g_cnt[0]);
    const bool big = cnt0 > LC_QSTAGE;            // then gN == 1: rounds over its occurrences, counts carried in S.acc
    const uint32_t total = big ? cnt0 : (uint32_t)wg_uniform((int)(S.g_es[gN - 1] + S.g_cnt[gN - 1]));
    for (uint32_t r0 = 0; r0 < total; r0 += LC_QSTAGE) {
      const int cnt = (int)(total - r0 < LC_QSTAGE ? total - r0 : LC_QSTAGE);
      XG_FOR(ln, LC_QLANES) {   // ---- step 1: entries ln, ln + lanes, ... of the staging area
        constexpr int U = LC_QSTAGE / LC_QLANES;
        float gcov[4] = {0.f, 0.f, 0.f, 0.f}; uint32_t gfl = 0;
        const bool gfetch = r0 == 0 && ln < gN;                  // lane k: candidate k's record for the tail of this group
        if (gfetch) { LC_GLOBAL const NodeGr &G = W.gr[S.g_n[ln]]; gcov[0] = G.cov[0]; gcov[1] = G.cov[1]; gcov[2] = G.cov[2]; gcov[3] = G.cov[3]; gfl = G.flags; }
        cs_t e[U]; uint32_t m[U][4], meta[U]; bool act[U];
        const uint32_t *gd[U]; uint32_t ri[U];
        for (int u = 0; u < U; ++u) {
          const int j = ln + u * LC_QLANES;
          act[u] = j < cnt;
          e[u] = 0;
          if (act[u]) {
            uint32_t src;
            if (big) src = S.g_lo[0] + r0 + (uint32_t)j;
            else { int k = 0; while (k + 1 < gN && (uint32_t)j >= S.g_es[k + 1]) ++k; src = S.g_lo[k] + ((uint32_t)j - S.g_es[k]); }
            e[u] = W_CSR(W)[src];
          }
        }
        for (int u = 0; u < U; ++u) {
          act[u] = act[u] && CS_ST(e[u]) == 0;
          ri[u] = 0; gd[u] = LC_CTX(c).B->good;
          if (act[u]) { const lc_u4 rdv = *(const lc_u4 *)(W.rd + 4 * (size_t)CS_READ(e[u])); ri[u] = rdv.x; gd[u] = LC_CTX(c).B->good + rdv.z; }
        }
        for (int u = 0; u < U; ++u) {
          m[u][0] = m[u][1] = m[u][2] = m[u][3] = 0; meta[u] = 0;
          if (act[u]) {
            const int p0 = (int)CS_POS(e[u]), sh = p0 & 31, wv = p0 >> 5;
            // bits [p0, p0+K) of the read's mask, 32 at a time; the word after is only touched when it holds needed bits
            #define LC_TAKE(t) ((32 * (t) < K) ? ((gd[u][wv + (t)] >> sh) | ((sh && 32 * (t) + 32 - sh < K) ? (gd[u][wv + (t) + 1] << (32 - sh)) : 0u)) : 0u)
            m[u][0] = LC_TAKE(0); m[u][1] = LC_TAKE(1); m[u][2] = LC_TAKE(2); m[u][3] = LC_TAKE(3);
            #undef LC_TAKE
            meta[u] = 1u | (((RI_NML(ri[u]) ? 2u : 0u) + (RI_REV(ri[u]) ? 1u : 0u)) << 1) | (CS_ORI(e[u]) << 3) | (((uint32_t)e[u] >> 29) << 4);
          }
        }
        for (int u = 0; u < U; ++u) {
          const int j = ln + u * LC_QLANES;
          // (k <= 96 leaves the fourth mask word free: the class word rides along, one 16-byte LDS read per entry in step 2)
          if (u == 0 && gfetch) { S.g_tt[ln] = gcov[0] + gcov[1]; S.g_tn[ln] = gcov[2] + gcov[3]; S.g_fl[ln] = gfl; }
          if (j < cnt) { S.mk[j][0] = m[u][0]; S.mk[j][1] = m[u][1]; S.mk[j][2] = m[u][2]; S.mk[j][3] = K <= 96 ? meta[u] : m[u][3]; S.mmeta[j] = meta[u]; }
        }
      }
      const bool first = (r0 == 0), last = (r0 + LC_QSTAGE >= total);
      const int T = gN * K;
#ifdef LANCET_FAT
      // few (candidate, position) pairs -- a pile-up's candidates have thousands of occurrences each, so a group is one or two
      // of them: P lane groups take a slice of the staged entries each and add their counts up in LDS
      const int P = (2 * T <= LC_QLANES) ? LC_QLANES / T : 1;
      if (P > 1 && first) { XG_FOR(x, T * 10) { S.pacc[x] = 0; } }
#else
      constexpr int P = 1; (void)P;
#endif
      WG_SYNC();
#ifdef LANCET_FAT
      if (P > 1) {
        XG_FOR(x, T * P) {
          const int part = x / T, t = x - part * T;
          const int k = big ? 0 : t / K, i = big ? t : t - k * K;
          const int es = big ? 0 : (int)S.g_es[k], ee = big ? cnt : es + (int)S.g_cnt[k];
          const int chunk = (ee - es + P - 1) / P;
          const int js = es + part * chunk, je = js + chunk < ee ? js + chunk : ee;
          unsigned long long a = 0;
          uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0;
          const int iR = K - 1 - i;
          const int wF = i >> 5, sF = i & 31, wR = iR >> 5, sR = iR & 31;
          const lc_u4 *mk4 = (const lc_u4 *)S.mk;
          const uint32_t *mm = (const uint32_t *)S.mmeta;
          for (int j = js; j < je; ++j) {
            const lc_u4 v = mk4[j]; const uint32_t meta = mm[j];
            const bool rev = (meta & 8u) != 0;
            const int wsel = rev ? wR : wF, sh = rev ? sR : sF;
            const uint32_t word = K <= 32 ? v.x : (wsel == 0 ? v.x : wsel == 1 ? v.y : wsel == 2 ? v.z : v.w);
            const uint32_t bit = (word >> sh) & meta & 1u;
            const uint32_t cls = (meta >> 1) & 3u;
            a += (unsigned long long)bit << (16 * cls);
            if (LR) {
              const uint32_t gT = (cls < 2) ? bit : 0u, gNm = (cls >= 2) ? bit : 0u, gr3 = meta >> 4;
              h0 += gT & gr3; h1 += gT & (gr3 >> 1); h2 += gT & (gr3 >> 2);
              h3 += gNm & gr3; h4 += gNm & (gr3 >> 1); h5 += gNm & (gr3 >> 2);
            }
          }
          LC_LDS uint32_t *pa = (LC_LDS uint32_t *)&S.pacc[10 * t];
          if (js < je) {
            for (int q = 0; q < 4; ++q) { const uint32_t aq = (uint32_t)((a >> (16 * q)) & 0xFFFFu); if (aq) dev_atomic_add(pa + q, aq); }
            if (LR) {
              const uint32_t hh[6] = {h0, h1, h2, h3, h4, h5};
              for (int q = 0; q < 6; ++q) if (hh[q]) dev_atomic_add(pa + 4 + q, hh[q]);
            }
          }
        }
        WG_SYNC();
        if (last) {
          XG_FOR(t, T) {
            const int k = big ? 0 : t / K, i = big ? t : t - k * K;
            const uint32_t a0 = S.pacc[10 * t], a1 = S.pacc[10 * t + 1], a2 = S.pacc[10 * t + 2], a3 = S.pacc[10 * t + 3];
            LC_GLOBAL uint16_t *qq = W.qv + (size_t)S.g_row[k] * K * QS;
            qq[QS * i] = (uint16_t)a0; qq[QS * i + 1] = (uint16_t)a1; qq[QS * i + 2] = (uint16_t)a2; qq[QS * i + 3] = (uint16_t)a3;
            if (LR) { uint16_t *qh = qq + QS * i + 4; for (int q = 0; q < 6; ++q) qh[q] = (uint16_t)S.pacc[10 * t + 4 + q]; }
            const int sq = (int)(uint16_t)a0 + (int)(uint16_t)a1 + (int)(uint16_t)a2 + (int)(uint16_t)a3;
            dev_atomic_min((LC_LDS uint32_t *)&S.g_min[k], (uint32_t)sq);
          }
          WG_SYNC();
        }
        continue;
      }
#endif
      XG_FOR(t, T) {   // ---- step 2
        const int k = big ? 0 : t / K, i = big ? t : t - k * K;
        const int es = big ? 0 : (int)S.g_es[k], ee = big ? cnt : es + (int)S.g_cnt[k];
        const uint32_t *mm = (const uint32_t *)S.mmeta;
        unsigned long long a = 0;                                      // four 16-bit counters: class c at bits 16c -- of THIS round (at most LC_QSTAGE entries: no field
                                                                       // carries into its neighbour); what the rounds before counted is added below, modulo 65 536 as the
                                                                       // reference's unsigned short counters count
        uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0;       // lr_mode: hp0/1/2_minqv tumor, normal
        uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        if (!first) { p0 = S.acc[i][0]; p1 = S.acc[i][1]; p2 = S.acc[i][2]; p3 = S.acc[i][3]; }
        if (!first && LR) { h0 = S.acc[i][4]; h1 = S.acc[i][5]; h2 = S.acc[i][6]; h3 = S.acc[i][7]; h4 = S.acc[i][8]; h5 = S.acc[i][9]; }
        // bit of k-mer position i in an entry: position i of a forward occurrence, K-1-i of a reverse one
        const int iR = K - 1 - i;
        const int wF = i >> 5, sF = i & 31, wR = iR >> 5, sR = iR & 31;
        auto add_entry = [&](const lc_u4 v, const uint32_t meta) {
          const bool rev = (meta & 8u) != 0;
          const int wsel = rev ? wR : wF, sh = rev ? sR : sF;
          const uint32_t word = K <= 32 ? v.x : (wsel == 0 ? v.x : wsel == 1 ? v.y : wsel == 2 ? v.z : v.w);
          const uint32_t bit = (word >> sh) & meta & 1u;
          const uint32_t cls = (meta >> 1) & 3u;
          a += (unsigned long long)bit << (16 * cls);
          if (LR) {   // Node_t::updateHPCovDistr: quality ok and the stored count had grown
            const uint32_t gT = (cls < 2) ? bit : 0u, gNm = (cls >= 2) ? bit : 0u, gr3 = meta >> 4;
            h0 += gT & gr3; h1 += gT & (gr3 >> 1); h2 += gT & (gr3 >> 2);
            h3 += gNm & gr3; h4 += gNm & (gr3 >> 1); h5 += gNm & (gr3 >> 2);
          }
        };
        const lc_u4 *mk4 = (const lc_u4 *)S.mk;                       // plain LDS reads: staged before the barrier above
        if (K <= 96) {   // the class word is the entry's fourth word; four entries in flight per trip (the loop is LDS-latency bound)
          int j = es;
          for (; j + 4 <= ee; j += 4) {
            const lc_u4 v0 = mk4[j], v1 = mk4[j + 1], v2 = mk4[j + 2], v3 = mk4[j + 3];
            add_entry(v0, v0.w); add_entry(v1, v1.w); add_entry(v2, v2.w); add_entry(v3, v3.w);
          }
          for (; j < ee; ++j) { const lc_u4 v = mk4[j]; add_entry(v, v.w); }
        } else {
          for (int j = es; j < ee; ++j) add_entry(mk4[j], mm[j]);
        }
        const uint32_t a0 = (p0 + (uint32_t)(a & 0xFFFFu)) & 0xFFFFu, a1 = (p1 + (uint32_t)((a >> 16) & 0xFFFFu)) & 0xFFFFu,
                       a2 = (p2 + (uint32_t)((a >> 32) & 0xFFFFu)) & 0xFFFFu, a3 = (p3 + (uint32_t)(a >> 48)) & 0xFFFFu;
        if (!last) {
          S.acc[i][0] = (uint16_t)a0; S.acc[i][1] = (uint16_t)a1; S.acc[i][2] = (uint16_t)a2; S.acc[i][3] = (uint16_t)a3;
          if (LR) { S.acc[i][4] = (uint16_t)h0; S.acc[i][5] = (uint16_t)h1; S.acc[i][6] = (uint16_t)h2; S.acc[i][7] = (uint16_t)h3; S.acc[i][8] = (uint16_t)h4; S.acc[i][9] = (uint16_t)h5; }
        } else {
          LC_GLOBAL uint16_t *qq = W.qv + (size_t)S.g_row[k] * K * QS;
          qq[QS * i] = (uint16_t)a0; qq[QS * i + 1] = (uint16_t)a1; qq[QS * i + 2] = (uint16_t)a2; qq[QS * i + 3] = (uint16_t)a3;
          if (LR) { uint16_t *qh = qq + QS * i + 4; qh[0] = (uint16_t)h0; qh[1] = (uint16_t)h1; qh[2] = (uint16_t)h2; qh[3] = (uint16_t)h3; qh[4] = (uint16_t)h4; qh[5] = (uint16_t)h5; }
          const int sq = (int)(uint16_t)a0 + (int)(uint16_t)a1 + (int)(uint16_t)a2 + (int)(uint16_t)a3;
          dev_atomic_min((LC_LDS uint32_t *)&S.g_min[k], (uint32_t)sq);
        }
      }
      WG_SYNC();
    }
    if (rows) { ci += (uint32_t)gN; if (big) cq_valid = false; continue; }
    // ---- the first removeLowCov predicate per candidate; survivors keep their counts (slot qi0 + k; a non-survivor
    //      leaves its slot unused) and start their sequence-descriptor deque
    WG_FOR(t, gN * K) {
      const int k = t / K, i = t - k * K;
      const uint32_t n = S.g_n[k];
      const int minqv = (int)S.g_min[k];
      const float tt = S.g_tt[k], tn = S.g_tn[k];
      const bool low = (minqv <= LC_CTX(c).P->low_cov_threshold) || ((double)minqv <= (LC_CTX(c).P->min_cov_ratio * avgcov)) || (tt == 1.0f && tn == 1.0f);
      if (!low) {
        const uint32_t base = (qi0 + (uint32_t)k) * (uint32_t)K;
        LC_GLOBAL const unsigned long long *kk = W.nkey + (size_t)n * LC_NWMAX;
        W.seq[base + i] = SD_MAKE(n, i, key_base(kk, K, i));
      }
    }
    WG_FOR(k, gN) {                              // one lane per candidate of the group: stores only
      {
        const uint32_t n = S.g_n[k];
        LC_GLOBAL NodeGr &G = W.gr[n];
        const int minqv = (int)S.g_min[k];
        const float tt = S.g_tt[k], tn = S.g_tn[k];
        const bool low = (minqv <= LC_CTX(c).P->low_cov_threshold) || ((double)minqv <= (LC_CTX(c).P->min_cov_ratio * avgcov)) || (tt == 1.0f && tn == 1.0f);
        G.mincovqv = minqv;
        if (!low) {
          const uint32_t qi = qi0 + (uint32_t)k, base = qi * (uint32_t)K;
          G.seq_clo = base; G.seq_lo = base; G.seq_hi = base + K; G.seq_chi = base + K;
          const uint32_t f = S.g_fl[k];
          G.nkmT = ((f & NF_TUMOR) && !(f & NF_NORMAL)) ? 1u : 0u;      // cov_status == 'T'
          G.nqv = qi;
          G.flags = f | NF_SURV;
        }
      }
    }
    WG_LANE0 { S.qv_top = qi0 + (uint32_t)gN; }
    ci += (uint32_t)gN;
    if (big) cq_valid = false;                   // its running counts lived where the candidate list does
  }
  if (!rows) WG_LANE0 { S.seq_top = S.qv_top * (uint32_t)K; }
  WG_SYNC();
}
DEVNI void build_refcov(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W; LC_GLOBAL const EngineCaps &C = *LC_CTX(c).C;
  const int K = S.K;
  (void)C; (void)K; (void)W;
  // ---- Ref_t::mertable membership (indexMers over the possibly trimmed seq, reference src/Ref.cc:40-64)
  {
    uint32_t ro = W.occ_base[S.R - 1];
    int nrefk = S.reflen - K + 1;                          // k-mers of the reference pseudo-read
    bool loaded = (S.reflen - K > 0);
    WG_FOR(i, S.seq_len - K > 0 ? S.seq_len - K : 0) {     // i + K < seq.length()
      int p = S.seq_t5 + i;
      if (loaded && p < nrefk) dev_atomic_or(&W.gr[W.occ[ro + p] & 0x3FFFFFFFu].flags, NF_INMER);
    }
    WG_SYNC();
    // ---- Ref_t::computeCoverage (reference src/Ref.cc:173-250): per rawseq position, Tf Tr Nf Nr
    WG_FOR(j, S.reflen) { for (int q = 0; q < 4; ++q) W.refcov[4 * j + q] = 0; }
    WG_SYNC();
    WG_FOR(i, S.reflen - K > 0 ? S.reflen - K : 0) {       // i + K < rawseq.length()
      uint32_t X = W.occ[ro + i] & 0x3FFFFFFFu;
      uint16_t v[4] = {0, 0, 0, 0};
      if (ld2(&W.gr[X].flags) & NF_INMER) for (int q = 0; q < 4; ++q) v[q] = W.gr[X].kc[q];
      if (i == 0) { for (int j = 0; j < K; ++j) for (int q = 0; q < 4; ++q) W.refcov[4 * j + q] = v[q]; }
      else { for (int q = 0; q < 4; ++q) W.refcov[4 * (i + K - 1) + q] = v[q]; }
    }
    if (S.LR) {        // hp0 hp1 hp2 of Ref_t::computeCoverage (reference src/Ref.cc:192-196, 216-240): same pattern
      WG_FOR(j, S.reflen) { for (int q = 0; q < 6; ++q) W.refhp[6 * j + q] = 0; }
      WG_SYNC();
      WG_FOR(i, S.reflen - K > 0 ? S.reflen - K : 0) {
        uint32_t X = W.occ[ro + i] & 0x3FFFFFFFu;
        uint16_t v[6] = {0, 0, 0, 0, 0, 0};
        if (ld2(&W.gr[X].flags) & NF_INMER) for (int q = 0; q < 6; ++q) v[q] = W.khp[6 * (size_t)X + q];
        if (i == 0) { for (int j = 0; j < K; ++j) for (int q = 0; q < 6; ++q) W.refhp[6 * j + q] = v[q]; }
        else { for (int q = 0; q < 6; ++q) W.refhp[6 * (i + K - 1) + q] = v[q]; }
      }
    }
    WG_SYNC_FENCE();   // from here on the node arrays are only touched with plain loads/stores: one L1 invalidate
  }
}
DEV void build_graph(Ctx &c) {
  WG_LANE0 { LC_CTX(c).W->qv = LC_CTX(c).W->qv_own; }
  if (!wg_bcast(&LC_SREF(c).items_ready)) { build_items(c); WG_LANE0 { LC_SREF(c).items_ready = 1; } }     // (a window whose graphs all come from the LDS build kernel never needs them)
  build_tables(c);
  if (wg_bcast(&LC_SREF(c).overflow)) return;
  PHASE(c, 4);
  STOP_RET(c, 4);
  build_csr(c);
  if (wg_bcast(&LC_SREF(c).overflow)) return;
  PHASE(c, 5);
  STOP_RET(c, 5);
  build_gather(c);
  if (wg_bcast(&LC_SREF(c).overflow)) return;
  PHASE(c, 6);
  build_qcounts(c);
  if (wg_bcast(&LC_SREF(c).overflow)) return;
  build_refcov(c);
  STOP_RET(c, 6);
}

// ---------------------------------------------------------------------------------------------------------
// libstdc++ iteration order of the node table after the build (nodes were inserted in id order 0..N-1).
// The sequential replay (ht_insert) costs ~3N dependent memory round trips; the same order has a closed form per
// growth stage.  While the bucket count is B, every insertion AND every rehash step applies one rule to an element x
// of a sequence Q: "bucket(x) empty -> x becomes the list head; else x goes to the front of its bucket's run".
// Hence after processing Q the list is: runs ordered by the position of their first element in Q, latest first;
// inside a run, latest first.  Q of a stage = (list of the previous stage, then the node ids inserted before the next
// rehash).  Each stage is a counting sort by (first position of the bucket desc, position desc): all parallel.
// The first 29 insertions (2 small stages) are replayed sequentially.
// ---------------------------------------------------------------------------------------------------------
DEVNI void order_stage(Ctx &c, LC_GLOBAL const uint32_t *Q, LC_GLOBAL uint32_t *Qn, int n, uint32_t B) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  LC_GLOBAL uint32_t *bkt = W.scratch, *tmp = W.scratch + LC_CTX(c).C->node_cap;
  LC_GLOBAL uint32_t *first = W.ht_bucket, *cnt = W.ht_cnt, *start = W.ht_start, *out = W.ht_next;
  LC_GLOBAL const unsigned long long *nhash = W.nhash;
  // Every pass below is a chain of dependent global round trips per element (index -> hash -> bucket words); LC_ILP elements per lane are
  // in flight per trip (round 4: a table of 10 k nodes -- 100x / 40x windows at k = 31..101 -- spent 8 ms here on one wave).
  // (an element past the end is clamped for the loads and skipped for the stores)
  WG_FOR(b, (int)B) { cnt[b] = 0; first[b] = LC_NIL; }
  WG_SYNC();
  WG_FOR_ILP(i0, n) {
    uint32_t q[LC_ILP]; unsigned long long h[LC_ILP];
    for (int u = 0; u < LC_ILP; ++u) { const int i = i0 + u * LC_ILP_STRIDE; q[u] = Q[i < n ? i : n - 1]; }
    for (int u = 0; u < LC_ILP; ++u) h[u] = nhash[q[u]];
    for (int u = 0; u < LC_ILP; ++u) {
      const int i = i0 + u * LC_ILP_STRIDE;
      if (i >= n) continue;
      const uint32_t b = ht_mod(h[u], B);
      bkt[i] = b;
      dev_atomic_add(&cnt[b], 1u);
      dev_atomic_min(&first[b], (uint32_t)i);
    }
  }
  WG_SYNC();
  WG_FOR_ILP(i0, n) {
    uint32_t b[LC_ILP], f[LC_ILP], m[LC_ILP];
    for (int u = 0; u < LC_ILP; ++u) { const int i = i0 + u * LC_ILP_STRIDE; b[u] = bkt[i < n ? i : n - 1]; }
    for (int u = 0; u < LC_ILP; ++u) { f[u] = ld2(&first[b[u]]); m[u] = ld2(&cnt[b[u]]); }
    for (int u = 0; u < LC_ILP; ++u) { const int i = i0 + u * LC_ILP_STRIDE; if (i < n) tmp[n - 1 - i] = (f[u] == (uint32_t)i) ? m[u] : 0u; }
  }
  WG_SYNC();
  wg_scan(tmp, n, S);                                   // elements in front of the run that starts with position i
  WG_FOR_ILP(i0, n) {
    uint32_t b[LC_ILP], f[LC_ILP], t[LC_ILP];
    for (int u = 0; u < LC_ILP; ++u) { const int i = i0 + u * LC_ILP_STRIDE; const int ic = i < n ? i : n - 1; b[u] = bkt[ic]; t[u] = tmp[n - 1 - ic]; }
    for (int u = 0; u < LC_ILP; ++u) f[u] = ld2(&first[b[u]]);
    for (int u = 0; u < LC_ILP; ++u) { const int i = i0 + u * LC_ILP_STRIDE; if (i < n && f[u] == (uint32_t)i) start[b[u]] = t[u]; }
  }
  WG_SYNC();
  WG_FOR(b, (int)B) { first[b] = 0; }                   // from here: fill cursor of the run
  WG_SYNC();
  WG_FOR_ILP(i0, n) {
    uint32_t b[LC_ILP], s0[LC_ILP];
    for (int u = 0; u < LC_ILP; ++u) { const int i = i0 + u * LC_ILP_STRIDE; b[u] = bkt[i < n ? i : n - 1]; }
    for (int u = 0; u < LC_ILP; ++u) s0[u] = start[b[u]];
    for (int u = 0; u < LC_ILP; ++u) { const int i = i0 + u * LC_ILP_STRIDE; if (i < n) { const uint32_t at = s0[u] + dev_atomic_add(&first[b[u]], 1u); out[at] = (uint32_t)i; } }
  }
  WG_SYNC();
  WG_FOR(b, (int)B) {
    uint32_t m = ld2(&cnt[b]);
    if (m > 1) {
      uint32_t *o = out + start[b];
      for (uint32_t i = 1; i < m; ++i) { uint32_t v = o[i]; uint32_t j = i; while (j > 0 && o[j - 1] < v) { o[j] = o[j - 1]; --j; } o[j] = v; }
    }
  }
  WG_SYNC();
  WG_FOR_ILP(j0, n) {
    uint32_t o[LC_ILP], v[LC_ILP];
    for (int u = 0; u < LC_ILP; ++u) { const int j = j0 + u * LC_ILP_STRIDE; o[u] = out[j < n ? j : n - 1]; }
    for (int u = 0; u < LC_ILP; ++u) v[u] = Q[o[u]];
    for (int u = 0; u < LC_ILP; ++u) { const int j = j0 + u * LC_ILP_STRIDE; if (j < n) Qn[j] = v[u]; }
  }
  WG_SYNC();
}

// the live table in libstdc++ iteration order -> order[0..M)
DEVNI void first_lowcov(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const uint32_t SEQ = 29u;                             // a value of the growth chain
  WG_LANE0 {
    ht_reset(c);
    uint32_t lim = S.N < SEQ ? S.N : SEQ;
    for (uint32_t n = 0; n < lim && !S.overflow; ++n) ht_insert(c, n);
    uint32_t m = 0;
    for (uint32_t p = S.ht_head; p != LC_NIL; p = W.ht_next[p]) W.order[m++] = p;
    S.M = m;
  }
  const uint32_t N = wg_bcastu(&S.N);
  if (N <= SEQ) return;
  LC_GLOBAL uint32_t *Q = W.order, *Qn = W.pnodes;
  uint32_t nprev = SEQ, B = SEQ;
  while (true) {
    B = ht_next_prime(2u * B);
    if (B == 0 || B > LC_CTX(c).C->bucket_cap) { WG_LANE0 { OVF(c); } return; }
    const uint32_t n = N < B ? N : B;
    WG_FOR(j, (int)(n - nprev)) { Q[nprev + j] = nprev + (uint32_t)j; }
    WG_SYNC();
    order_stage(c, Q, Qn, (int)n, B);
    LC_GLOBAL uint32_t *t = Q; Q = Qn; Qn = t;
    if (N <= B) break;
    nprev = B;
  }
  if (Q != W.order) { WG_FOR(j, (int)N) { W.order[j] = Q[j]; } WG_SYNC(); }
  WG_LANE0 { S.M = N; S.ht_bc = B; S.ht_elt = N; S.ht_next_resize = B; S.ht_head = LC_NIL; }
}

// cleanDead over the whole table (reference src/Graph.cc:2737-2762), parallel: compaction of order[] keeping the order
DEVNI void clean_dead_wg(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int M = (int)wg_bcastu(&S.M);
  LC_GLOBAL uint32_t *keep = W.scratch;
  if (wg_uniform(S.prebuilt)) { WG_FOR(i, M) { keep[i] = W.survb[W.order[i]] ? 1u : 0u; } }     // (no node is dead yet; the records of the non-survivors were never written)
  else { WG_FOR(i, M) { const uint32_t f = W.gr[W.order[i]].flags; keep[i] = ((f & NF_SURV) && !(f & NF_DEAD)) ? 1u : 0u; } }
  WG_LANE0 { keep[M] = 0; }
  wg_scan(keep, M + 1, S);
  WG_FOR(i, M) { if (keep[i + 1] != keep[i]) W.pnodes[keep[i]] = W.order[i]; }
  WG_SYNC();
  const int live = (int)wg_bcastu(&S.part[LANCET_WG]);
  WG_FOR(i, live) { W.order[i] = W.pnodes[i]; }
  WG_LANE0 { S.M = (uint32_t)live; S.ht_elt -= (uint32_t)(M - live); evt(c, EV_CLEANDEAD, (uint32_t)(M - live)); }
}

DEV void recompute_after_append(Ctx &c, uint32_t n, uint32_t from, uint32_t to) {
  // Node_t::computeMinCov over the merged arrays == min(old minima, minima over the appended descriptors)
  LC_GLOBAL Work &W = *LC_CTX(c).W;
  int mn = W.gr[n].mincov, mq = W.gr[n].mincovqv;
  for (uint32_t i = from; i < to; ++i) { int t, tq; desc_tot(c, W.seq[i], &t, &tq); if (t < mn) mn = t; if (tq < mq) mq = tq; }
  W.gr[n].mincov = mn; W.gr[n].mincovqv = mq;
}

DEVNI void compress_node(Ctx &c, uint32_t node, char dir) {          // Graph_t::compressNode, reference src/Graph.cc:2486-2706
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_WS &S = LC_SREF(c);
  const int K = S.K;
  while (!S.overflow) {
    int uid = get_buddy(c, node, dir);
    if (uid == -1) return;
    if (is_tandem(c, node)) return;
    uint32_t ew = W.gr[node].edges[uid];
    uint32_t edir = ED_DIR(ew);
    char bdir = (edir == 0 || edir == 2) ? 'R' : 'F';
    uint32_t buddy = ED_TO(ew);
    if (is_tandem(c, buddy)) return;
    int buid = get_buddy(c, buddy, bdir);
    if (buid == -1) return;
    bool brev = dir_dest(edir) == 'R';
    int alen = n_len(c, node), blen = n_len(c, buddy);
    uint32_t tail = (uint32_t)(blen - (K - 1));
    // merged = astr + bstr[K-1:]  (dir F: append ; dir R: prepend the reverse complement)
    if (dir == 'F') {
      if (!seq_reserve(c, node, 0, tail)) return;
      uint32_t hi = W.gr[node].seq_hi, blo = W.gr[buddy].seq_lo, bhi = W.gr[buddy].seq_hi;
      for (uint32_t t = 0; t < tail; ++t) {
        uint32_t d = brev ? W.seq[bhi - 1 - ((uint32_t)(K - 1) + t)] : W.seq[blo + (uint32_t)(K - 1) + t];
        if (brev) d ^= 3u;
        W.seq[hi + t] = d;
      }
      W.gr[node].seq_hi = hi + tail;
      recompute_after_append(c, node, hi, hi + tail);
    } else {
      if (!seq_reserve(c, node, tail, 0)) return;
      uint32_t lo = W.gr[node].seq_lo, blo = W.gr[buddy].seq_lo, bhi = W.gr[buddy].seq_hi;
      // element t of B'[K-1:] lands at lo-1-t, complemented
      for (uint32_t t = 0; t < tail; ++t) {
        uint32_t d = brev ? W.seq[bhi - 1 - ((uint32_t)(K - 1) + t)] : W.seq[blo + (uint32_t)(K - 1) + t];
        if (brev) d ^= 3u;
        W.seq[lo - 1 - t] = d ^ 3u;
      }
      W.gr[node].seq_lo = lo - tail;
      recompute_after_append(c, node, lo - tail, lo);
    }
    W.gr[node].nkm += W.gr[buddy].nkm; W.gr[node].nkmT += W.gr[buddy].nkmT;
    int amer = alen - K + 1, bmer = blen - K + 1;
    float *nc = W.gr[node].cov; const float *bc = W.gr[buddy].cov;
    for (int q = 0; q < 4; ++q) nc[q] = ((nc[q] * amer) + (bc[q] * bmer)) / (amer + bmer);      // Graph.cc:2632-2636
    W.gr[buddy].flags |= NF_DEAD;
    W.gr[node].flags |= (W.gr[buddy].flags & (NF_TUMOR | NF_NORMAL));
    erase_edge_at(c, node, uid);
    int bcnt = (int)W.gr[buddy].necnt;
    for (int i = 0; i < bcnt; ++i) {
      if (i == buid) continue;
      uint32_t be = W.gr[buddy].edges[i];
      uint32_t ndir = ED_DIR(be);
      if (edir == 1 || edir == 2) ndir = flipme(ndir);
      uint32_t other = ED_TO(be);
      int cnt = (int)W.gr[node].necnt;
      if (cnt >= LC_EMAX) { OVF(c); return; }
      if (other == buddy) { W.gr[node].edges[cnt] = ED_MAKE(node, ndir) | (be & (1u << 30)); W.gr[node].necnt = cnt + 1; }
      else {
        W.gr[node].edges[cnt] = ED_MAKE(other, ndir) | (be & (1u << 30)); W.gr[node].necnt = cnt + 1;
        update_edge(c, other, buddy, fliplink(ED_DIR(be)), node, fliplink(ndir));
      }
    }
  }
}
// ---------------------------------------------------------------------------------------------------------
// First compress of a component (every node still is one k-mer): the same merges in the same order as
// compress()/compress_node() above, but each merge costs ONE dependent memory access instead of a dozen.
//   compress_prepare (all lanes): per live node a 64-byte record with everything a merge reads from the absorbed
//     node -- its two mergeable links (the getBuddy / isTandem / reciprocal-getBuddy tests of compressNode are
//     properties of the untouched graph), coverages, flags, the descriptor and the minima the appended base brings.
//   compress_fast (lane 0): for each head in table order, walk the chain of records read-only into a list
//     (a ring or anything irregular falls back to compress_node on the untouched state), then replay the list:
//     descriptors, float averages in merge order, minima, flags; the last absorbed node's outward edges are
//     attached exactly like compress_node does.
// ---------------------------------------------------------------------------------------------------------
#define CL_VALID 0x80000000u
#define CL_TO(l) ((l) & 0x0FFFFFFFu)
#define CL_DIR(l) (((l) >> 28) & 3u)
DEV uint32_t cmp_link(const Ctx &c, uint32_t n, char dir, bool *irregular) {
  int uid = get_buddy(c, n, dir);
  if (uid == -1 || is_tandem(c, n)) return 0u;
  const uint32_t ew = LC_CTX(c).W->gr[n].edges[uid], edir = ED_DIR(ew), b = ED_TO(ew);
  if (is_tandem(c, b)) return 0u;
  const char bdir = (edir == 0 || edir == 2) ? 'R' : 'F';
  const int buid = get_buddy(c, b, bdir);
  if (buid == -1) return 0u;
  if (ED_TO(LC_CTX(c).W->gr[b].edges[buid]) != n) { *irregular = true; return 0u; }
  return CL_VALID | (edir << 28) | b;
}
// First half of a node record (flags, degree, component, colour, the 12 edge words) in registers: the link tests of compress_prepare
// look at a node's and its neighbours' edges a dozen times each; field by field that was ~100 dependent (L1-hit) loads per node.
// The loops run over all 12 slots with a predicate, so that the edge words stay in registers (an array indexed at run time would
// live in scratch memory).
struct GrLine0 { uint32_t flags, necnt; int comp; uint32_t color; uint32_t e0, e1, e2, e3, e4, e5, e6, e7, e8, e9, e10, e11; };
DEV GrLine0 gr_line0(LC_GLOBAL const NodeGr *g) {
  LC_GLOBAL const uint32_t *p = (LC_GLOBAL const uint32_t *)g;
  const lc_u4 a = ldg4(p), b = ldg4(p + 4), c = ldg4(p + 8), d = ldg4(p + 12);
  GrLine0 r; r.flags = a.x; r.necnt = a.y; r.comp = (int)a.z; r.color = a.w;
  r.e0 = b.x; r.e1 = b.y; r.e2 = b.z; r.e3 = b.w; r.e4 = c.x; r.e5 = c.y; r.e6 = c.z; r.e7 = c.w; r.e8 = d.x; r.e9 = d.y; r.e10 = d.z; r.e11 = d.w;
  return r;
}
#define LC_L0_EACH(g, X) do { X(0, (g).e0); X(1, (g).e1); X(2, (g).e2); X(3, (g).e3); X(4, (g).e4); X(5, (g).e5); X(6, (g).e6); X(7, (g).e7); X(8, (g).e8); X(9, (g).e9); X(10, (g).e10); X(11, (g).e11); } while (0)
// Node_t::getBuddy on the registers: the one edge in direction dir (LC_NIL: none, several, a special node, or a self loop)
DEV uint32_t l0_buddy(const GrLine0 &g, uint32_t self, char dir) {
  if (g.flags & NF_SPECIAL) return LC_NIL;
  uint32_t ew = LC_NIL; int cnt = 0;
#define LC_X(i, e) do { if ((uint32_t)(i) < g.necnt && is_dir(ED_DIR(e), dir)) { ew = (e); ++cnt; } } while (0)
  LC_L0_EACH(g, LC_X);
#undef LC_X
  if (cnt != 1 || ED_TO(ew) == self) return LC_NIL;
  return ew;
}
DEV bool l0_tandem(const GrLine0 &g, uint32_t self) {
  bool t = false;
#define LC_X(i, e) do { if ((uint32_t)(i) < g.necnt && ED_TO(e) == self) t = true; } while (0)
  LC_L0_EACH(g, LC_X);
#undef LC_X
  return t;
}
// descriptor i of candidate ci's k-mer while the descriptors are not materialised (load_prebuilt)
DEV uint32_t seq_desc_lazy(const LC_WS &S, uint32_t node, uint32_t ci, int K, int i) {
  return SD_MAKE(node, i, pre_key_base((LC_GLOBAL const unsigned long long *)(uintptr_t)S.lz_skey, S.lz_kw, ci, K, i));
}
// all lanes: write every survivor's descriptors (what load_prebuilt used to do), for the routes that read W.seq of k-mer nodes directly
DEVNI void seq_materialize_all(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  if (!wg_bcast(&S.seq_lazy)) return;
  LC_GLOBAL const PreLayout &PL = LC_PL(c);
  LC_GLOBAL const uint8_t *area = (LC_GLOBAL const uint8_t *)(uintptr_t)S.lz_area;
  LC_GLOBAL const PreHdr *H = (LC_GLOBAL const PreHdr *)(area + PRE_OFF_HDR);
  LC_GLOBAL const uint32_t *snode = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_SNODE);
  LC_GLOBAL const unsigned long long *skey = (LC_GLOBAL const unsigned long long *)(area + PRE_OFF_SKEY);
  const int K = wg_uniform(S.K);
  const uint32_t ncand = H->ncand, kw = PL.kw;
  WG_FOR(t, ncand * (uint32_t)K) {
    const uint32_t ci = (uint32_t)t / (uint32_t)K; const int i = (int)((uint32_t)t % (uint32_t)K);
    const uint32_t n = snode[ci];
    if (n != LC_NIL) W.seq[t] = SD_MAKE(n, i, pre_key_base(skey, kw, ci, K, i));
  }
  WG_LANE0 { S.seq_lazy = 0; }
  WG_SYNC();
}
DEVNI void compress_prepare(Ctx &c, int comp) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = S.K, QS = S.QS;
  LC_GLOBAL NodeGr *gr = W.gr; LC_GLOBAL CmpRec *cmp = W.cmp; LC_GLOBAL uint32_t *todo = W.todo; LC_GLOBAL const uint32_t *seq = W.seq; LC_GLOBAL const uint16_t *qv = W.qv;
  WG_LANE0 { S.cmp_ok = 1; }
  const bool lazy = wg_bcast(&S.seq_lazy) != 0;
  WG_FOR(i, S.M) {
    const uint32_t n = W.order[i];
    // round trip 1: the node's own record
    const GrLine0 G = gr_line0(&gr[n]);
    LC_GLOBAL const uint32_t *g1 = (LC_GLOBAL const uint32_t *)&gr[n] + 16;
    const lc_u4 h0 = ldg4(g1), h1 = ldg4(g1 + 4), h2 = ldg4(g1 + 8), h3 = ldg4(g1 + 12);      // cov[4] | mincov mincovqv seq_lo seq_hi | seq_clo seq_chi nkm nkmT | nqv onref kc[4]
    lc_u4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
    todo[n] = 0;                                                 // "absorbed" mark of compress_fast / compress_rank (todo[] is idle after the build)
    if (G.comp != comp || (G.flags & (NF_DEAD | NF_SPECIAL))) { cmp[n].lnk[0] = 0; cmp[n].lnk[1] = 0; continue; }
    const uint32_t seq_lo = h1.z, seq_hi = h1.w, nkm = h2.z, nkmT = h2.w, nqv = h3.x;
    if ((int)(seq_hi - seq_lo) != K || nkm != 1 || nqv == LC_NIL) { cmp[n].lnk[0] = 0; cmp[n].lnk[1] = 0; S.cmp_ok = 0; continue; }
    const bool tand = l0_tandem(G, n);
    const uint32_t ewF = tand ? LC_NIL : l0_buddy(G, n, 'F'), ewR = tand ? LC_NIL : l0_buddy(G, n, 'R');
    // round trip 2: the two neighbours' records, the first / last descriptor, their per-position quality counts
    const uint32_t bF = ewF != LC_NIL ? ED_TO(ewF) : n, bR = ewR != LC_NIL ? ED_TO(ewR) : n;
    const GrLine0 BF = gr_line0(&gr[bF]), BR = gr_line0(&gr[bR]);
    uint32_t d0, dK;
    if (lazy && seq_lo == nqv * (uint32_t)K) {                  // (a k-mer node of a graph from the LDS build kernel: its descriptors follow from its key)
      d0 = seq_desc_lazy(S, n, nqv, K, 0); dK = seq_desc_lazy(S, n, nqv, K, K - 1);
    } else { d0 = seq[seq_lo]; dK = seq[seq_lo + (uint32_t)(K - 1)]; }
    LC_GLOBAL const uint16_t *q0p = qv + ((size_t)nqv * K + 0) * QS, *qKp = qv + ((size_t)nqv * K + (size_t)(K - 1)) * QS;
    const int tq0 = (int)q0p[0] + (int)q0p[1] + (int)q0p[2] + (int)q0p[3], tqK = (int)qKp[0] + (int)qKp[1] + (int)qKp[2] + (int)qKp[3];
    bool irr = false;
    uint32_t lnk[2] = {0u, 0u};
    for (int sd = 0; sd < 2; ++sd) {
      const uint32_t ew = sd == 0 ? ewF : ewR;
      if (ew == LC_NIL) continue;
      const uint32_t edir = ED_DIR(ew), bn = ED_TO(ew);
      const GrLine0 &Bq = sd == 0 ? BF : BR;
      if (l0_tandem(Bq, bn)) continue;
      const char bdir = (edir == 0 || edir == 2) ? 'R' : 'F';
      const uint32_t bew = l0_buddy(Bq, bn, bdir);
      if (bew == LC_NIL) continue;
      if (ED_TO(bew) != n) { irr = true; continue; }
      lnk[sd] = CL_VALID | (edir << 28) | bn;
    }
    // (the descriptors of a single k-mer node belong to the node itself: desc_tot's `tot` is the sum of its own four counts)
    const int tot = (int)(h3.z & 0xFFFFu) + (int)(h3.z >> 16) + (int)(h3.w & 0xFFFFu) + (int)(h3.w >> 16);
    lc_u4 r0, r1, r2;
    r0.x = lnk[0]; r0.y = lnk[1]; r0.z = h0.x; r0.w = h0.y;                          // lnk[2], cov[0..1]
    r1.x = h0.z; r1.y = h0.w; r1.z = G.flags; r1.w = nkmT;                           // cov[2..3], flags, nkmT
    r2.x = d0; r2.y = dK; r2.z = (uint32_t)tot; r2.w = (uint32_t)tq0;                // d0 dK tot tq0
    LC_GLOBAL uint32_t *rp = (LC_GLOBAL uint32_t *)&cmp[n];
    stg4(rp, r0); stg4(rp + 4, r1); stg4(rp + 8, r2); rp[12] = (uint32_t)tqK;
    (void)z; (void)h2;
    if (irr) S.cmp_ok = 0;
  }
  WG_SYNC();
}
DEVNI uint32_t compress_fast(Ctx &c, int comp, bool quiet = false) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = S.K;
  if (!quiet) evt(c, EV_COMPRESS);
  LC_GLOBAL uint32_t *list = W.scratch;                                   // (absorbed node, edge dir it is entered by) per merge
  const uint32_t lcap = LC_CTX(c).C->node_cap;
  // Which nodes are still heads is read off the merge records (both links of a node outside the component, of a special
  // node and of every node absorbed so far are 0), one 8-byte load per table position, fetched one position ahead; the
  // absorbed nodes are also marked in todo[] for the compaction of the table order, which the whole wave does afterwards
  // (compact_absorbed_wg; S.tmp2 says whether it is due).  After a ring (literal replay, which marks its merges in the
  // node records only) both fall back to the node records.
  const uint32_t M = S.M;
  bool ringed = false;
  uint32_t nabs = 0;
  uint32_t Hn = M ? W.order[0] : 0;
  uint32_t ln0 = M ? W.cmp[Hn].lnk[0] : 0u, ln1 = M ? W.cmp[Hn].lnk[1] : 0u;
  for (uint32_t oi = 0; oi < M && !S.overflow; ++oi) {
    const uint32_t H = Hn;
    const uint32_t lh = ln0 | ln1;
    if (oi + 1 < M) { Hn = W.order[oi + 1]; ln0 = W.cmp[Hn].lnk[0]; ln1 = W.cmp[Hn].lnk[1]; }
    if (!ringed) {
      if (!(lh & CL_VALID)) continue;                                         // nothing to merge here (or not a head)
      if (!((W.cmp[H].lnk[0] | W.cmp[H].lnk[1]) & CL_VALID)) continue;        // (the early fetch may predate its absorption)
    } else {
      if (W.gr[H].comp != comp) continue;
      if (W.gr[H].flags & (NF_DEAD | NF_SPECIAL)) continue;
    }
    for (int pass = 0; pass < 2 && !S.overflow; ++pass) {
      const char dir = pass == 0 ? 'F' : 'R';
      // ---- read-only walk.  The head's own link comes from its record; after a merge the head's edge in `dir` is the
      //      absorbed node's onward edge, its direction flipped when the entering edge was FR or RF (flipme).
      uint32_t l = W.cmp[H].lnk[pass];
      uint32_t cnt = 0; bool ring = false;
      uint32_t edir = CL_DIR(l), B = CL_TO(l);
      bool valid = (l & CL_VALID) != 0;
      while (valid) {
        if (B == H || cnt >= lcap) { ring = true; break; }
        LC_GLOBAL const CmpRec &rb = W.cmp[B];                               // the one dependent access of this merge
        list[2 * cnt] = B; list[2 * cnt + 1] = edir; ++cnt;
        const char bdir = (edir == 0 || edir == 2) ? 'R' : 'F';
        const uint32_t on = rb.lnk[bdir == 'R' ? 0 : 1];          // the absorbed node's link away from the head
        valid = (on & CL_VALID) != 0;
        if (valid) { uint32_t nd = CL_DIR(on); if (edir == 1 || edir == 2) nd = flipme(nd); edir = nd; B = CL_TO(on); }
      }
      if (ring) { ringed = true; compress_node(c, H, dir); continue; }           // untouched so far: the literal replay handles it
      if (cnt == 0) continue;
      // ---- replay
      LC_GLOBAL NodeGr &G = W.gr[H];
      {                                                            // the head's edge to the first absorbed node goes away
        const int uid = get_buddy(c, H, dir);
        if (uid == -1) { OVF(c); return 0; }
        erase_edge_at(c, H, uid);
      }
      if (!seq_reserve(c, H, dir == 'F' ? 0u : cnt, dir == 'F' ? cnt : 0u)) return 0;
      uint32_t lo = G.seq_lo, hi = G.seq_hi;
      int mn = G.mincov, mq = G.mincovqv;
      float nc0 = G.cov[0], nc1 = G.cov[1], nc2 = G.cov[2], nc3 = G.cov[3];
      uint32_t fl = G.flags, nkm = G.nkm, nkmT = G.nkmT;
      int alen = (int)(hi - lo);
      // one merge per trip; the record of the next absorbed node is fetched before this trip's stores (a store orders
      // every later load behind it for the compiler, and each record fetch is a full memory round trip)
      uint32_t Bn = list[0], en = list[1];
      CmpRec rn = W.cmp[Bn];
      for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t Bj = Bn, ed = en;
        const CmpRec rb = rn;
        if (j + 1 < cnt) { Bn = list[2 * j + 2]; en = list[2 * j + 3]; rn = W.cmp[Bn]; }
        const bool brev = dir_dest(ed) == 'R';
        uint32_t d = brev ? (rb.d0 ^ 3u) : rb.dK;
        if (dir == 'F') W.seq[hi++] = d; else W.seq[--lo] = d ^ 3u;
        if (rb.tot < mn) mn = rb.tot;
        { const int tq = brev ? rb.tq0 : rb.tqK; if (tq < mq) mq = tq; }
        const int amer = alen - K + 1, bmer = 1;                  // Graph.cc:2632-2636, same expression, same order
        nc0 = ((nc0 * amer) + (rb.cov[0] * bmer)) / (amer + bmer);
        nc1 = ((nc1 * amer) + (rb.cov[1] * bmer)) / (amer + bmer);
        nc2 = ((nc2 * amer) + (rb.cov[2] * bmer)) / (amer + bmer);
        nc3 = ((nc3 * amer) + (rb.cov[3] * bmer)) / (amer + bmer);
        ++alen; nkm += 1; nkmT += rb.nkmT;
        fl |= rb.flags & (NF_TUMOR | NF_NORMAL);
        W.gr[Bj].flags = rb.flags | NF_DEAD;
        W.cmp[Bj].lnk[0] = 0; W.cmp[Bj].lnk[1] = 0; W.todo[Bj] = 1; ++nabs;
      }
      G.seq_lo = lo; G.seq_hi = hi; G.mincov = mn; G.mincovqv = mq;
      G.cov[0] = nc0; G.cov[1] = nc1; G.cov[2] = nc2; G.cov[3] = nc3;
      G.flags = fl; G.nkm = nkm; G.nkmT = nkmT;
      // ---- the last absorbed node's other edges move to the head (compress_node's tail, unchanged)
      const uint32_t buddy = list[2 * (cnt - 1)], ed = list[2 * (cnt - 1) + 1];
      const char bdir = (ed == 0 || ed == 2) ? 'R' : 'F';
      const int buid = get_buddy(c, buddy, bdir);
      const int bcnt = (int)W.gr[buddy].necnt;
      for (int i = 0; i < bcnt; ++i) {
        if (i == buid) continue;
        const uint32_t be = W.gr[buddy].edges[i];
        uint32_t ndir = ED_DIR(be);
        if (ed == 1 || ed == 2) ndir = flipme(ndir);
        const uint32_t other = ED_TO(be);
        const int ec = (int)G.necnt;
        if (ec >= LC_EMAX) { OVF(c); return 0; }
        if (other == buddy) { G.edges[ec] = ED_MAKE(H, ndir) | (be & (1u << 30)); G.necnt = ec + 1; }
        else {
          G.edges[ec] = ED_MAKE(other, ndir) | (be & (1u << 30)); G.necnt = ec + 1;
          update_edge(c, other, buddy, fliplink(ED_DIR(be)), H, fliplink(ndir));
        }
      }
    }
  }
  if (ringed || S.overflow) { S.tmp2 = 0; return clean_dead(c, quiet); }
  S.tmp2 = 1;                                                     // cleanDead is due: compact_absorbed_wg, by the whole wave
  if (!quiet) evt(c, EV_CLEANDEAD, nabs);
  return nabs;
}
// ---------------------------------------------------------------------------------------------------------
// The first compress of a component by the whole wave (same result as compress_fast, which pays one dependent memory round
// trip per merge on one lane -- a clean 600-bp window is ONE chain of ~590 merges).
//   * The mergeable links of compress_prepare make every chain a path; a node has two "ports" (leave it through its F or
//     its R side), and following the links from a port is a linked list over ports.  Pointer jumping over the ports
//     (log2 rounds, one 16-byte record per port and round) gives every port: how many nodes lie beyond it, the smallest
//     table position among them (the chain's head is the member the reference's loop over the table reaches first:
//     Graph.cc:2712-2732), the port the list ends in, and the parity of orientation-changing links (FR / RF) along it.
//   * From those every absorbed node knows its head, the side of the head it hangs on, its place j in the merge order
//     and the direction it is entered by (compressNode's dir bookkeeping, Graph.cc:2486-2706), and writes its
//     descriptor into the head's new deque and its merge operands into the head's slice of a list in merge order.
//   * One lane per head replays the float averaging (Graph.cc:2632-2636) over its slice -- the only sequential part, on
//     contiguous memory -- and builds the head's edge list: its own edges without the two merged links, then the outward
//     edges of the F-side end, then those of the R-side end (the erase / push_back order of compressNode).
//   * All live nodes redirect edges that point at an absorbed node to its head (Node_t::updateEdge keeps the position).
// A ring (no end to start from) or an asymmetric link leaves everything untouched and returns false: compress_fast runs.
// ---------------------------------------------------------------------------------------------------------
// port record of compress_rank in 8 bytes: next port [12:0] (8191 = none) | nodes beyond [25:13] | smallest table position [38:26] |
// last port [51:39] | parity [52]   <->   x = next (LC_NIL none), y = nodes beyond | parity << 31, z = smallest position, w = last port
DEV lc_u4 pr_unpack(unsigned long long v) {
  lc_u4 r; const uint32_t nx = (uint32_t)(v & 0x1FFFu);
  r.x = nx == 0x1FFFu ? LC_NIL : nx; r.y = (uint32_t)((v >> 13) & 0x1FFFu) | ((uint32_t)((v >> 52) & 1u) << 31); r.z = (uint32_t)((v >> 26) & 0x1FFFu); r.w = (uint32_t)((v >> 39) & 0x1FFFu);
  return r;
}
DEV unsigned long long pr_pack(const lc_u4 r) {
  return (unsigned long long)(r.x == LC_NIL ? 0x1FFFu : r.x) | ((unsigned long long)(r.y & 0x1FFFu) << 13) | ((unsigned long long)(r.z & 0x1FFFu) << 26) | ((unsigned long long)(r.w & 0x1FFFu) << 39) | ((unsigned long long)(r.y >> 31) << 52);
}
DEVNI bool compress_rank(Ctx &c, int comp) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = S.K;
  SUBPHASE(c, 1, 2);
  const uint32_t M = wg_bcastu(&S.M);
  if (M == 0 || 2 * M > 8190u || (size_t)44 * M + 64 > (size_t)4 * LC_CTX(c).C->occ_cap) return false;
  LC_GLOBAL uint32_t *pos = W.pnodes;                                       // node -> table position
  // port records, 8 bytes: next port [12:0] (8191 = none) | nodes beyond [25:13] | smallest table position [38:26] | last port [51:39] | parity [52]
  LC_GLOBAL unsigned long long *PA = (LC_GLOBAL unsigned long long *)W.mv, *PB = PA + 2 * (size_t)M;      // ping-pong
  LC_GLOBAL uint32_t *ord = W.mv + 16 * (size_t)M;                         // the absorbed nodes' coverages in merge order (4 floats each)
  LC_GLOBAL uint32_t *hacc = W.mv + 20 * (size_t)M;                        // [heads * 4] min tot, min totqv, colour bits, tumor-only k-mers of the absorbed nodes
  LC_GLOBAL uint32_t *hs = W.mv + 24 * (size_t)M;                          // [M+1] absorbed nodes per head -> slice start
  LC_GLOBAL uint32_t *al = hs + (M + 1);                                   // [M+1] deque length per head -> arena offset
  LC_GLOBAL uint32_t *ne = al + (M + 1);                                   // [M * 13] new edge lists of the heads (count + 12)
  LC_GLOBAL uint32_t *hl = ne + 13 * (size_t)M;                            // [M] table positions of the heads (any order)
  WG_FOR(i, M) { pos[W.order[i]] = (uint32_t)i; }
  WG_LANE0 { S.tmp0 = 0; S.tmp3 = 0; S.cmp_nh = 0; }
  WG_SYNC();
  WG_FOR(i, M) {
    const uint32_t n = W.order[i];
    for (uint32_t sd = 0; sd < 2; ++sd) {
      const uint32_t l = W.cmp[n].lnk[sd];
      lc_u4 r; r.x = LC_NIL; r.y = 0; r.z = (uint32_t)i; r.w = 2u * (uint32_t)i + sd;
      if (l & CL_VALID) {
        const uint32_t B = CL_TO(l), ed = CL_DIR(l);
        const uint32_t sb = (ed == 0 || ed == 2) ? 1u : 0u;                 // the side of B the link arrives at ('R' -> 1)
        const uint32_t back = W.cmp[B].lnk[sb];
        if (!(back & CL_VALID) || CL_TO(back) != n || CL_DIR(back) != fliplink(ed)) S.tmp3 = 1;      // not the mirror image: leave it to the literal replay
        r.x = 2u * pos[B] + (1u - sb); r.y = 1u | (((ed == 1 || ed == 2) ? 1u : 0u) << 31);
      }
      PA[2u * (uint32_t)i + sd] = pr_pack(r);
    }
  }
  if (wg_bcast(&S.tmp3)) return false;
  SUBPHASE(c, 1, 3);
  // ---- pointer jumping
  LC_GLOBAL unsigned long long *P = PA, *Q = PB;
  for (int round = 0; round < 15; ++round) {
    WG_LANE0 { S.tmp0 = 0; }
    WG_FOR(l, LANCET_WG) {                                                  // four ports per lane and trip: the records first, then what they point at
      const int NP = (int)(2 * M);
      for (int p0 = l; p0 < NP; p0 += 4 * LANCET_WG) {
        lc_u4 r[4], q[4];
        for (int u = 0; u < 4; ++u) { const int p = p0 + u * LANCET_WG; r[u] = pr_unpack(P[p < NP ? p : l]); }
        for (int u = 0; u < 4; ++u) q[u] = pr_unpack(P[r[u].x != LC_NIL ? r[u].x : (uint32_t)l]);
        for (int u = 0; u < 4; ++u) {
          const int p = p0 + u * LANCET_WG;
          if (p >= NP) continue;
          if (r[u].x != LC_NIL) {
            r[u].y = (((r[u].y & 0x7FFFFFFFu) + (q[u].y & 0x7FFFFFFFu)) & 0x7FFFFFFFu) | ((r[u].y ^ q[u].y) & 0x80000000u);
            if (q[u].z < r[u].z) r[u].z = q[u].z;
            r[u].w = q[u].w; r[u].x = q[u].x;
            if (r[u].x != LC_NIL) S.tmp0 = 1;
          }
          Q[p] = pr_pack(r[u]);
        }
      }
    }
    WG_SYNC();
    { LC_GLOBAL unsigned long long *t = P; P = Q; Q = t; }
    if (!wg_bcast(&S.tmp0)) break;
    if (round == 14) return false;                                        // a ring: no port ever reaches an end
  }
  // ---- heads, slices, arena space
  SUBPHASE(c, 1, 4);
  WG_FOR(i, M) {
    const lc_u4 a = pr_unpack(P[2 * (size_t)i]), b = pr_unpack(P[2 * (size_t)i + 1]);
    const uint32_t mF = a.y & 0x7FFFFFFFu, mR = b.y & 0x7FFFFFFFu;
    const uint32_t cmin = a.z < b.z ? a.z : b.z;
    const bool head = (mF + mR > 0) && cmin == (uint32_t)i;
    hs[i] = head ? mF + mR : 0u; al[i] = head ? (uint32_t)K + mF + mR : 0u;
    if (head) { lc_u4 z; z.x = 0x7FFFFFFFu; z.y = 0x7FFFFFFFu; z.z = 0; z.w = 0; stg4(hacc + 4 * (size_t)i, z); hl[dev_atomic_add((LC_LDS uint32_t *)&S.cmp_nh, 1u)] = (uint32_t)i; }
  }
  WG_LANE0 { hs[M] = 0; al[M] = 0; }
  wg_scan(hs, (int)M + 1, S);
  const uint32_t nabs = wg_bcastu(&S.part[LANCET_WG]);
  wg_scan(al, (int)M + 1, S);
  const uint32_t need = wg_bcastu(&S.part[LANCET_WG]);
  const uint32_t top = wg_bcastu(&S.seq_top);
  if (top + need > LC_CTX(c).C->seq_cap) { WG_LANE0 { OVF(c); S.tmp1 = 0; S.tmp2 = 0; } return true; }
  // ---- every absorbed node: head, side, place, entering direction; descriptor into the head's deque, operands into its slice
  SUBPHASE(c, 1, 5);
  WG_FOR(i, M) {
    const lc_u4 a = pr_unpack(P[2 * (size_t)i]), b = pr_unpack(P[2 * (size_t)i + 1]);
    const uint32_t mF = a.y & 0x7FFFFFFFu, mR = b.y & 0x7FFFFFFFu;
    if (mF + mR == 0) continue;
    const uint32_t cmin = a.z < b.z ? a.z : b.z;
    const uint32_t n = W.order[i];
    if (cmin == (uint32_t)i) { W.cmp[n].pad[0] = n; W.cmp[n].pad[1] = 0; continue; }       // a head
    const uint32_t sH = (a.z == cmin) ? 0u : 1u;                            // the head lies beyond this side
    const lc_u4 away = sH ? a : b;                                          // the port that leads away from the head
    const lc_u4 hF = pr_unpack(P[2 * (size_t)cmin]), hR = pr_unpack(P[2 * (size_t)cmin + 1]);
    const bool onF = away.w == hF.w && (hF.y & 0x7FFFFFFFu) > 0;
    const lc_u4 hp = onF ? hF : hR;
    const uint32_t mdir = hp.y & 0x7FFFFFFFu, hmF = hF.y & 0x7FFFFFFFu, hmR = hR.y & 0x7FFFFFFFu;
    const uint32_t j = mdir - (away.y & 0x7FFFFFFFu);                       // 1 .. mdir
    const uint32_t st_j = ((hp.y ^ away.y) >> 31) & 1u;
    LC_GLOBAL const CmpRec &r = W.cmp[n];
    const uint32_t raw = fliplink(CL_DIR(r.lnk[sH]));                      // the link that enters this node, in its predecessor's frame
    const uint32_t st_p = st_j ^ ((raw == 1 || raw == 2) ? 1u : 0u);
    const uint32_t edir = st_p ? flipme(raw) : raw;
    const bool brev = (edir == 1 || edir == 3);
    const uint32_t H = W.order[cmin];
    const uint32_t nb = top + al[cmin];                                     // the head's new deque: [nb, nb + K + hmF + hmR), its own k-mer at nb + hmR
    const uint32_t d = brev ? (r.d0 ^ 3u) : r.dK;
    if (onF) W.seq[nb + hmR + (uint32_t)K + (j - 1u)] = d; else W.seq[nb + hmR - j] = d ^ 3u;
    const uint32_t t = hs[cmin] + (onF ? j - 1u : hmF + j - 1u);
    lc_u4 o0;
    o0.x = __builtin_bit_cast(uint32_t, r.cov[0]); o0.y = __builtin_bit_cast(uint32_t, r.cov[1]); o0.z = __builtin_bit_cast(uint32_t, r.cov[2]); o0.w = __builtin_bit_cast(uint32_t, r.cov[3]);
    stg4(ord + 4 * (size_t)t, o0);
    {                                                                       // what does not depend on the merge order goes straight to the head
      LC_GLOBAL uint32_t *ha = hacc + 4 * (size_t)cmin;
      dev_atomic_min(&ha[0], (uint32_t)r.tot); dev_atomic_min(&ha[1], (uint32_t)(brev ? r.tq0 : r.tqK));
      if (r.flags & (NF_TUMOR | NF_NORMAL)) dev_atomic_or(&ha[2], r.flags & (NF_TUMOR | NF_NORMAL));
      if (r.nkmT) dev_atomic_add(&ha[3], r.nkmT);
    }
    W.cmp[n].pad[0] = H; W.cmp[n].pad[1] = 1u | (st_j << 1) | (edir << 2);
    W.gr[n].flags = r.flags | NF_DEAD; W.todo[n] = 1;
  }
  // (Barriers, not agent-scope fences, between these passes: plain stores and loads of ONE wave through its CU's L1 stay ordered;
  //  only words updated by atomics -- performed at L2 -- must then be read past the L1 (ld2).  A __threadfence() writes back /
  //  invalidates across the XCDs' L2s and was measured at ~4 % of the kernel per fence and window.)
  WG_SYNC();
  // ---- one lane per head: the new deque, the new edge list  (the heads are few: a list of them instead of a look at every position)
  SUBPHASE(c, 1, 6);
  const uint32_t nheads = wg_bcastu(&S.cmp_nh);
  const bool lazy = wg_bcast(&S.seq_lazy) != 0;
  WG_FOR(hx, nheads) {
    const uint32_t i = hl[hx];
    const uint32_t cnt = hs[i + 1] - hs[i];
    if (cnt == 0) continue;
    const uint32_t H = W.order[i];
    LC_GLOBAL NodeGr &G = W.gr[H];
    const lc_u4 a = pr_unpack(P[2 * (size_t)i]), b = pr_unpack(P[2 * (size_t)i + 1]);
    const uint32_t mF = a.y & 0x7FFFFFFFu, mR = b.y & 0x7FFFFFFFu;
    const uint32_t nb = top + al[i], olo = G.seq_lo;
    if (lazy && olo == G.nqv * (uint32_t)K) { for (int t = 0; t < K; ++t) W.seq[nb + mR + (uint32_t)t] = seq_desc_lazy(S, H, G.nqv, K, t); }
    else for (int t = 0; t < K; ++t) W.seq[nb + mR + (uint32_t)t] = W.seq[olo + (uint32_t)t];
    int mn = G.mincov, mq = G.mincovqv;
    uint32_t fl = G.flags, nkmT = G.nkmT;
    {
      lc_u4 ha; ha.x = ld2(&hacc[4 * (size_t)i]); ha.y = ld2(&hacc[4 * (size_t)i + 1]); ha.z = ld2(&hacc[4 * (size_t)i + 2]); ha.w = ld2(&hacc[4 * (size_t)i + 3]);
      if ((int)ha.x < mn) mn = (int)ha.x;
      if ((int)ha.y < mq) mq = (int)ha.y;
      fl |= ha.z; nkmT += ha.w;
    }
    // edges: own ones without the merged links, then the outward edges of the F-side end, then of the R-side end
    uint32_t el[LC_EMAX + 1]; int m = 0; bool bad = false;
    const int uF = mF ? get_buddy(c, H, 'F') : -1, uR = mR ? get_buddy(c, H, 'R') : -1;
    if ((mF && uF < 0) || (mR && uR < 0)) bad = true;
    for (int e = 0; e < (int)G.necnt; ++e) { if (e == uF || e == uR) continue; el[m++] = G.edges[e]; }
    for (int side = 0; side < 2 && !bad; ++side) {
      if (!(side == 0 ? mF : mR)) continue;
      const uint32_t E = W.order[(side == 0 ? a.w : b.w) >> 1];            // the end of the list in that direction
      const uint32_t info = W.cmp[E].pad[1], ed = (info >> 2) & 3u, st = (info >> 1) & 1u;
      const char bdir = (ed == 0 || ed == 2) ? 'R' : 'F';
      const int buid = get_buddy(c, E, bdir);
      const int bcnt = (int)W.gr[E].necnt;
      for (int e = 0; e < bcnt; ++e) {
        if (e == buid) continue;
        const uint32_t be = W.gr[E].edges[e];
        uint32_t ndir = ED_DIR(be);
        if (st) ndir = flipme(ndir);
        const uint32_t other = ED_TO(be);
        if (m >= LC_EMAX) { bad = true; break; }
        el[m++] = ED_MAKE(other == E ? H : other, ndir) | (be & (1u << 30));
      }
    }
    if (bad) { OVF(c); continue; }
    ne[13 * (size_t)i] = (uint32_t)m;
    for (int e = 0; e < m; ++e) ne[13 * (size_t)i + 1 + e] = el[e];
    G.seq_clo = nb; G.seq_lo = nb; G.seq_hi = nb + (uint32_t)K + mF + mR; G.seq_chi = G.seq_hi;
    G.mincov = mn; G.mincovqv = mq;
    G.flags = fl; G.nkm += cnt; G.nkmT = nkmT;
  }
  // ---- the float averaging of the merges (Graph.cc:2632-2636), in merge order: the one strictly sequential piece -- a clean window is
  //      ONE chain of ~590 merges, four IEEE divisions each.  The four coverages are independent recurrences: one lane per (head,
  //      coverage) instead of one per head, the next four operands already on their way.
  // Round 6: the division by the integer t + 2 as (float)((double)numerator * RCP[t]), RCP[t] = 1.0 / (double)(t + 2): that IS the correctly
  // rounded float quotient -- the double product is off by less than 2^-52 relative (two roundings of 2^-53), while the exact quotient of two
  // 24-bit floats is either a float or more than 2^-49 relative away from every midpoint between two floats (|A * 2^24 - M * B| >= 1 for an
  // odd M: B < 2^24 cannot divide it out), so rounding the product to float rounds as the quotient does (no overflow / subnormals: coverages
  // below 65 536, divisors below 4097).  Five dependent instructions per merge instead of the IEEE division's twelve.  The table is made by
  // all lanes in the (idle) scratch array.
  LC_GLOBAL double *RCP = (LC_GLOBAL double *)W.scratch;
  static_assert(sizeof(double) == 8, "reciprocal table");
  WG_FOR(t, nabs) { RCP[t] = 1.0 / (double)((uint32_t)t + 2u); }
  WG_SYNC();
  WG_FOR(x, 4 * nheads) {
    const uint32_t i = hl[(uint32_t)x >> 2], q = (uint32_t)x & 3u;
    const uint32_t cnt = hs[i + 1] - hs[i];
    if (cnt == 0) continue;
    LC_GLOBAL NodeGr &G = W.gr[W.order[i]];
    float nc = G.cov[q];
    LC_GLOBAL const uint32_t *sl = ord + 4 * (size_t)hs[i] + q;
#define LC_MERGE_STEP(cv, rv, t) do { if ((t) < cnt) { const int amer = (int)(t) + 1, bmer = 1;   /* Graph.cc:2632-2636: same numerator, same order; the division as above */ \
      const float num = (nc * amer) + (__builtin_bit_cast(float, (cv)) * bmer); nc = (float)((double)num * (rv)); } } while (0)
#define LC_MERGE_IDX(u, t0) ((size_t)((t0) + (u) < cnt ? (t0) + (u) : cnt - 1))
    uint32_t n0 = sl[4 * LC_MERGE_IDX(0u, 0u)], n1 = sl[4 * LC_MERGE_IDX(1u, 0u)], n2 = sl[4 * LC_MERGE_IDX(2u, 0u)], n3 = sl[4 * LC_MERGE_IDX(3u, 0u)];
    double r0 = RCP[LC_MERGE_IDX(0u, 0u)], r1 = RCP[LC_MERGE_IDX(1u, 0u)], r2 = RCP[LC_MERGE_IDX(2u, 0u)], r3 = RCP[LC_MERGE_IDX(3u, 0u)];
    for (uint32_t t0 = 0; t0 < cnt; t0 += 4) {
      const uint32_t c0 = n0, c1 = n1, c2 = n2, c3 = n3; const double d0 = r0, d1 = r1, d2 = r2, d3 = r3;
      if (t0 + 4 < cnt) {
        n0 = sl[4 * LC_MERGE_IDX(0u, t0 + 4)]; n1 = sl[4 * LC_MERGE_IDX(1u, t0 + 4)]; n2 = sl[4 * LC_MERGE_IDX(2u, t0 + 4)]; n3 = sl[4 * LC_MERGE_IDX(3u, t0 + 4)];
        r0 = RCP[LC_MERGE_IDX(0u, t0 + 4)]; r1 = RCP[LC_MERGE_IDX(1u, t0 + 4)]; r2 = RCP[LC_MERGE_IDX(2u, t0 + 4)]; r3 = RCP[LC_MERGE_IDX(3u, t0 + 4)];
      }
      LC_MERGE_STEP(c0, d0, t0); LC_MERGE_STEP(c1, d1, t0 + 1); LC_MERGE_STEP(c2, d2, t0 + 2); LC_MERGE_STEP(c3, d3, t0 + 3);
    }
#undef LC_MERGE_STEP
#undef LC_MERGE_IDX
    G.cov[q] = nc;
  }
  WG_SYNC();
  SUBPHASE(c, 1, 7);
  // ---- every live node of the component: the head's new list / its own one, edges into an absorbed node redirected to its head
  //      (the orientation the edge arrives in flips when the absorbed node's frame was flipped against the head's: updateEdge)
  WG_FOR(i, M) {
    const uint32_t n = W.order[i];
    LC_GLOBAL NodeGr &G = W.gr[n];
    if (G.comp != comp || (G.flags & NF_DEAD)) continue;
    const bool head = hs[i + 1] != hs[i];
    if (lazy && !head && !(G.flags & NF_SPECIAL) && G.nqv != LC_NIL && G.seq_lo == G.nqv * (uint32_t)K && G.seq_hi == G.seq_lo + (uint32_t)K) {
      const uint32_t b0 = G.seq_lo, q = G.nqv;                  // a k-mer node that stays on its own: its descriptors are due now
      for (int t = 0; t < K; ++t) W.seq[b0 + (uint32_t)t] = seq_desc_lazy(S, n, q, K, t);
    }
    const int cnt = head ? (int)ne[13 * (size_t)i] : (int)G.necnt;
    for (int e = 0; e < cnt; ++e) {
      uint32_t ew = head ? ne[13 * (size_t)i + 1 + e] : G.edges[e];
      const uint32_t to = ED_TO(ew);
      if (!(W.gr[to].flags & NF_SPECIAL) && W.todo[to]) {
        const uint32_t info = W.cmp[to].pad[1];
        ew = ED_MAKE(W.cmp[to].pad[0], ED_DIR(ew) ^ ((info >> 1) & 1u)) | (ew & (1u << 30));
      }
      G.edges[e] = ew;
    }
    G.necnt = (uint32_t)cnt;
  }
  WG_LANE0 { S.seq_top = top + need; S.tmp1 = (int)nabs; S.tmp2 = 1; }
  WG_SYNC();
  return true;
}

// cleanDead after compress_fast: the table order without the nodes marked in todo[] (scan-compaction, no node record touched)
DEVNI void compact_absorbed_wg(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int M = (int)wg_bcastu(&S.M);
  LC_GLOBAL uint32_t *keep = W.scratch;
  WG_FOR(i, M) { keep[i] = W.todo[W.order[i]] ? 0u : 1u; }
  WG_LANE0 { keep[M] = 0; }
  wg_scan(keep, M + 1, S);
  WG_FOR(i, M) { if (keep[i + 1] != keep[i]) W.pnodes[keep[i]] = W.order[i]; }
  WG_SYNC();
  const int live = (int)wg_bcastu(&S.part[LANCET_WG]);
  WG_FOR(i, live) { W.order[i] = W.pnodes[i]; }
  WG_LANE0 { S.M = (uint32_t)live; S.ht_elt -= (uint32_t)(M - live); }
}
DEVNI uint32_t compress(Ctx &c, int comp, bool quiet = false) {       // reference src/Graph.cc:2712-2732
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  S.tmp2 = 0;                                                    // (no compaction pending, see compress_fast)
  if (!quiet) evt(c, EV_COMPRESS);
  for (uint32_t i = 0; i < S.M && !S.overflow; ++i) {
    uint32_t n = W.order[i];
    if (W.gr[n].comp != comp) continue;
    if (W.gr[n].flags & NF_DEAD) continue;
    if (W.gr[n].flags & NF_SPECIAL) continue;
    compress_node(c, n, 'F');
    compress_node(c, n, 'R');
  }
  return clean_dead(c, quiet);
}
DEVNI void remove_low_cov(Ctx &c, int comp) {                         // reference src/Graph.cc:2790-2827 (docompression=true)
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const double avgcov = ((double)S.totalreadbp) / ((double)S.reflen);
  uint32_t low = 0;
  for (uint32_t i = 0; i < S.M; ++i) {
    uint32_t n = W.order[i];
    if (W.gr[n].comp != comp) continue;
    if (W.gr[n].flags & NF_SPECIAL) continue;
    int mq = W.gr[n].mincovqv;
    float tt = W.gr[n].cov[0] + W.gr[n].cov[1], tn = W.gr[n].cov[2] + W.gr[n].cov[3];
    if ((mq <= LC_CTX(c).P->low_cov_threshold) || ((double)mq <= (LC_CTX(c).P->min_cov_ratio * avgcov)) || (tt == 1.0f && tn == 1.0f)) { ++low; remove_node(c, n); }
  }
  evt(c, EV_LOWCOV, low);
  clean_dead(c);
  // Nothing removed: the component was compressed to the end just before (compress is idempotent, hasCycle in between changes
  // nothing), so the reference's unconditional compress() has nothing to merge -- only its trace lines are owed.
  if (low == 0) { evt(c, EV_COMPRESS); evt(c, EV_CLEANDEAD, 0); S.tmp2 = 0; }
  else compress(c, comp);
  print_stats(c, comp);
}

// findTandems (reference src/util.cc:574-758) on a code string (0..3): answered from the neighbourhood of `pos` alone; returns ans,
// len, motif (codes).  The reference scans the whole string and keeps, per unit length and phase, the start of the current
// stretch of equal consecutive units, and reports a stretch when it ends (at position i, having matched j more characters) if
// pos lies within delta of [start, i + j].  A stretch that ends at i < pos - delta - MAXU cannot reach pos, so the literal loop
// is entered at i0 = pos - delta - MAXU with the per-(unit, phase) stretch starts found by walking back over equal units (before
// i0 every unit is a whole one and the end-of-string rule of the reference cannot fire), and left as soon as every stretch in
// progress starts right of pos + delta.  ~200 character comparisons instead of ~10 000 for a 600-base path; same order of the
// reports (the reference appends every reported motif and keeps the last length).
DEVNI bool find_tandems_local(const Ctx &c, LC_GLOBAL const uint8_t *seq, int n, int pos, int *len, uint8_t *motif, int *motif_len,
                               const volatile LC_LDS uint8_t *stg = nullptr, int stg_lo = 0, int stg_hi = 0) {      // stg: seq[stg_lo .. stg_hi) in LDS (the caller's copy; every character is read a dozen times)
  const unsigned MAXU = (unsigned)LC_CTX(c).P->max_unit_len, MRU = (unsigned)LC_CTX(c).P->min_report_units, MRL = (unsigned)LC_CTX(c).P->min_report_len;
  const int delta = LC_CTX(c).P->dist_from_str;
  bool ans = false;
  *motif_len = 0;
  const unsigned MU = MAXU < 8 ? MAXU : 8;
  int i0 = pos - delta - (int)MU; if (i0 < 0) i0 = 0;
  if (i0 > n) i0 = n;
#define LC_SEQ(i) (((int)(i) >= stg_lo && (int)(i) < stg_hi) ? (int)stg[(int)(i) - stg_lo] : (int)seq[(i)])
  // (the stretch starts are indexed at run time: in LDS -- the staging area of the per-position pass is idle in the graph phases -- an
  //  indexed local array would live in scratch memory)
#ifndef LANCET_WAVE_EMU
  static_assert(sizeof(LC_SREF(c).mmeta) >= 9 * 8 * sizeof(int), "stretch starts of find_tandems_local");
  LC_LDS int (*offs)[8] = (LC_LDS int (*)[8])((LC_LDS uint32_t *)LC_SREF(c).mmeta);      // (not the staging area itself: the transcripts of the walk live there)
#else
  int offs[9][8];
#endif
  for (unsigned ml = 1; ml <= MU; ++ml) for (unsigned ph = 0; ph < ml; ++ph) {
    int t = (int)ph;
    if (i0 > (int)ph) {
      int u = i0 - 1 - (int)(((unsigned)(i0 - 1) - ph) % ml);            // last position < i0 of this phase
      // the stretch that holds u: back over equal neighbouring units
      t = u;
      while (t - (int)ml >= (int)ph) {
        bool eq = true;
        for (unsigned m = 0; m < ml; ++m) if (LC_SEQ(t - (int)ml + (int)m) != LC_SEQ(t + (int)m)) { eq = false; break; }
        if (!eq) break;
        t -= (int)ml;
      }
    }
    offs[ml][ph] = t;
  }
  // is [st, st + span) a whole number of copies of a unit shorter than `u`?  (the reference reports a stretch under its SMALLEST unit only)
  auto shorter_unit = [&](int st, unsigned span, unsigned u) -> bool {
    for (unsigned v = 1; v < u; ++v) {
      const unsigned reps = span / v;
      bool same = true;
      for (unsigned q = v; same && q < reps * v; ++q) same = LC_SEQ(st + (int)q) == LC_SEQ(st + (int)(q % v));
      if (same) return true;
    }
    return false;
  };
  auto byte_or_none = [&](int at) -> int { return (at < 0 || at >= n) ? 255 : LC_SEQ(at); };      // (the byte before a libstdc++ string is 0, never a code)
  for (unsigned i = (unsigned)i0; i < (unsigned)n; ++i) {
    bool live = false;                                           // a stretch in progress that still starts at or left of pos + delta
    for (unsigned u = 1; u <= MU; ++u) {
      LC_LDS int &slot = offs[u][i % u];
      const int st = slot;
      unsigned j = 0;                                            // characters of the unit at i that repeat the stretch's first unit
      while (j < u && i + j < (unsigned)n && LC_SEQ(i + j) == LC_SEQ(st + (int)j)) ++j;
      const bool ends = j != u || i + j + 1 == (unsigned)n;
      if (!ends) continue;
      const unsigned run = i - (unsigned)st;
      // a stretch counts when it cannot be extended to the left by one character, is long enough, and `u` is its smallest unit
      if (byte_or_none(st - 1) != byte_or_none(st + (int)u - 1) && run / u >= MRU && run >= MRL && !shorter_unit(st, run + j, u)) {
        const int stop = (int)(i + j);
        if (pos >= st - delta && pos <= stop + delta) {
          ans = true; *len = stop - st;
          for (unsigned z = 0; z < u; ++z) if (*motif_len < 60) motif[(*motif_len)++] = (uint8_t)LC_SEQ(st + (int)z);
        }
      }
      slot = (int)i;
    }
    for (unsigned ml = 1; ml <= MU && !live; ++ml) for (unsigned ph = 0; ph < ml; ++ph) if (offs[ml][ph] - delta <= pos) { live = true; break; }
    if (!live) break;
  }
  return ans;
#undef LC_SEQ
}

DEV void node_string(const Ctx &c, uint32_t n, LC_GLOBAL uint8_t *out) {      // str_m as codes
  const LC_GLOBAL Work &W = *LC_CTX(c).W;
  uint32_t lo = W.gr[n].seq_lo, hi = W.gr[n].seq_hi;
  for (uint32_t i = lo; i < hi; ++i) out[i - lo] = (uint8_t)SD_BASE(W.seq[i]);
}

DEVNI void remove_tips(Ctx &c, int comp) {                            // reference src/Graph.cc:2885-2926
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  int tips = 0, round = 0;
  do {
    ++round; tips = 0;
    evt(c, EV_TIPS_ROUND, round);
    for (uint32_t i = 0; i < S.M; ++i) {
      uint32_t n = W.order[i];
      if (W.gr[n].comp != comp) continue;
      if (W.gr[n].flags & NF_SPECIAL) continue;
      int deg = (int)W.gr[n].necnt, len = n_strlen(c, n) - S.K + 1;
      if (deg <= 1 && len < LC_CTX(c).P->max_tip_len) { remove_node(c, n); ++tips; }
    }
    evt(c, EV_TIPS_REMOVED, tips);
    if (tips) compress(c, comp);
  } while (tips && !S.overflow);
  print_stats(c, comp);
}
DEVNI void remove_short_links(Ctx &c, int comp) {                     // reference src/Graph.cc:2833-2880
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const double avgcov = ((double)S.totalreadbp) / ((double)S.reflen);
  const int max_link_len = S.K / 2;                                  // setK: floor(K/2.0)
  const double thr = floor(sqrt(avgcov));
  int links = 0;
  for (uint32_t i = 0; i < S.M; ++i) {
    uint32_t n = W.order[i];
    if (W.gr[n].comp != comp) continue;
    if (W.gr[n].flags & NF_SPECIAL) continue;
    int deg = (int)W.gr[n].necnt, len = n_len(c, n) - S.K + 1;
    if (deg >= 2 && len < max_link_len && (double)W.gr[n].mincov <= thr) {
      int L = 0, ml = 0; uint8_t motif[64];
      int sl = n_len(c, n);
      if (sl > (int)LC_CTX(c).C->path_cap) { OVF(c); return; }
      node_string(c, n, W.pseq);
      if (!find_tandems_local(c, W.pseq, sl, S.K - 1, &L, motif, &ml)) { remove_node(c, n); ++links; }
    }
  }
  evt(c, EV_LINKS, links);
  if (links) compress(c, comp);
  print_stats(c, comp);
}

// ---------------------------------------------------------------------------------------------------------
// removeLowCov / removeTips / removeShortLinks / compress of the cleaned component with the whole wave doing the looking.
// The reference walks the whole table per pass and removes / merges as it goes, so what a pass does to a node can depend
// on what it did earlier in the same walk (a removal lowers the neighbours' degrees).  The part of each pass's test that
// CANNOT change during the walk (component, special node, string length, coverage minima -- and for compress: which
// links are mergeable, because a merge only redirects edges and never makes another link mergeable) is evaluated by all
// lanes, the nodes that pass are compacted in table order, and lane 0 then walks only those with the live part of the test
// (the degree; compress_node's own checks).  A table walk is ~2 dependent round trips per node; the lists are a few
// nodes long.  cleanDead after each pass is a scan-compaction by the wave.
// ---------------------------------------------------------------------------------------------------------
enum { GP_LOWCOV = 0, GP_TIPS = 1, GP_LINKS = 2, GP_MERGE = 3 };
DEVNI int pass_candidates_wg(Ctx &c, int comp, int pass) {            // list -> W.pedges (idle outside the path phases), count returned
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int M = (int)wg_bcastu(&S.M);
  const int K = wg_bcast(&S.K);
  LC_GLOBAL uint32_t *fl = W.scratch;
  const double avgcov = ((double)S.totalreadbp) / ((double)S.reflen);
  const double thr = floor(sqrt(avgcov));
  const int max_link_len = K / 2, max_tip_len = LC_CTX(c).P->max_tip_len;
  LC_GLOBAL const NodeGr *gr = W.gr;
  WG_FOR(i, M) {
    const uint32_t n = W.order[i];
    const GrLine0 G = gr_line0(&gr[n]);
    LC_GLOBAL const uint32_t *g1 = (LC_GLOBAL const uint32_t *)&gr[n] + 16;
    const lc_u4 h0 = ldg4(g1), h1 = ldg4(g1 + 4);                  // cov[4] | mincov mincovqv seq_lo seq_hi
    bool ok = G.comp == comp && !(G.flags & (NF_SPECIAL | NF_DEAD));
    if (ok) {
      const int len = (int)(h1.w - h1.z);
      if (pass == GP_LOWCOV) {
        const int mq = (int)h1.y;
        const float tt = __builtin_bit_cast(float, h0.x) + __builtin_bit_cast(float, h0.y), tn = __builtin_bit_cast(float, h0.z) + __builtin_bit_cast(float, h0.w);
        ok = (mq <= LC_CTX(c).P->low_cov_threshold) || ((double)mq <= (LC_CTX(c).P->min_cov_ratio * avgcov)) || (tt == 1.0f && tn == 1.0f);
      } else if (pass == GP_TIPS) ok = (len - K + 1) < max_tip_len;
      else if (pass == GP_LINKS) ok = (len - K + 1) < max_link_len && (double)(int)h1.x <= thr;
      else {                                                       // a mergeable link on either side (compress_node's conditions)
        ok = false;
        if (!l0_tandem(G, n)) {
          const uint32_t ewF = l0_buddy(G, n, 'F'), ewR = l0_buddy(G, n, 'R');
          const uint32_t bF = ewF != LC_NIL ? ED_TO(ewF) : n, bR = ewR != LC_NIL ? ED_TO(ewR) : n;
          const GrLine0 BF = gr_line0(&gr[bF]), BR = gr_line0(&gr[bR]);
          for (int sd = 0; sd < 2; ++sd) {
            const uint32_t ew = sd == 0 ? ewF : ewR;
            if (ew == LC_NIL) continue;
            const uint32_t edir = ED_DIR(ew), bn = ED_TO(ew);
            const GrLine0 &Bq = sd == 0 ? BF : BR;
            if (l0_tandem(Bq, bn)) continue;
            if (l0_buddy(Bq, bn, (edir == 0 || edir == 2) ? 'R' : 'F') != LC_NIL) ok = true;
          }
        }
      }
    }
    fl[i] = ok ? 1u : 0u;
  }
  WG_LANE0 { fl[M] = 0; }
  wg_scan(fl, M + 1, S);
  WG_FOR(i, M) { if (fl[i + 1] != fl[i]) W.pedges[fl[i]] = W.order[i]; }
  WG_SYNC();
  return (int)wg_bcastu(&S.part[LANCET_WG]);
}
// cleanDead (reference src/Graph.cc:2737-2763) by the wave: the table order without the nodes marked dead
DEVNI void clean_dead_flags_wg(Ctx &c, bool quiet) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int M = (int)wg_bcastu(&S.M);
  LC_GLOBAL uint32_t *keep = W.scratch;
  WG_SYNC();                                                       // (lane 0 has just set the flags; plain stores and loads of ONE wave through its CU's L1 stay ordered: no cache maintenance -- an agent-scope fence per pass cost 25 % of the kernel)
  WG_FOR(i, M) { keep[i] = (W.gr[W.order[i]].flags & NF_DEAD) ? 0u : 1u; }
  WG_LANE0 { keep[M] = 0; }
  wg_scan(keep, M + 1, S);
  WG_FOR(i, M) { if (keep[i + 1] != keep[i]) W.pnodes[keep[i]] = W.order[i]; }
  WG_SYNC();
  const int live = (int)wg_bcastu(&S.part[LANCET_WG]);
  WG_FOR(i, live) { W.order[i] = W.pnodes[i]; }
  WG_LANE0 { S.M = (uint32_t)live; S.ht_elt -= (uint32_t)(M - live); if (!quiet) evt(c, EV_CLEANDEAD, (uint32_t)(M - live)); }
  WG_SYNC();
}
// Graph_t::compressNode (reference src/Graph.cc:2486-2706) by the whole wave (round 6; the single-wave window kernel only).  The merges of
// the passes that follow the first compress join unitigs of a hundred k-mers and more: compress_node copies the absorbed node's
// descriptors one by one and looks up the coverage minima behind each -- a load, a store and three dependent loads per descriptor on
// one lane, ~100 us of the ~120 us removeTips took per window.  Here every lane runs the node-level logic on the same values (uniform
// loads of the two records; only lane 0 stores), and the loops over descriptors -- the copy, the minima, the move of a deque that has
// no room left -- are shared out over the lanes.  Same merges in the same order, same float operations per merge.
DEV int l0_buddy_idx(const GrLine0 &g, uint32_t self, char dir) {      // get_buddy on the registers: index of the one edge in direction dir, or -1
  if (g.flags & NF_SPECIAL) return -1;
  int ret = -1, cnt = 0; uint32_t ew = 0;
#define LC_X(i, e) do { if ((uint32_t)(i) < g.necnt && is_dir(ED_DIR(e), dir)) { if (cnt == 0) { ret = (i); ew = (e); } ++cnt; } } while (0)
  LC_L0_EACH(g, LC_X);
#undef LC_X
  if (cnt != 1 || ED_TO(ew) == self) return -1;
  return ret;
}
DEV uint32_t l0_edge(const GrLine0 &g, int idx) {
  uint32_t r = 0;
#define LC_X(i, e) do { if ((i) == idx) r = (e); } while (0)
  LC_L0_EACH(g, LC_X);
#undef LC_X
  return r;
}
#ifndef LANCET_FAT
DEVNI void compress_node_wv(Ctx &c, uint32_t node, char dir) {
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_WS &S = LC_SREF(c);
  const int K = wg_uniform(S.K);
#ifndef LANCET_WAVE_EMU
  const bool l0 = wg_is_lane0();
#else
  const bool l0 = true;
#endif
  while (!wg_uniform(S.overflow)) {
    LC_GLOBAL NodeGr *gn = &W.gr[node];
    const GrLine0 G = gr_line0(gn);
    const int uid = wg_uniform(l0_buddy_idx(G, node, dir));
    if (uid == -1) return;
    if (wg_uniform((int)l0_tandem(G, node))) return;
    const uint32_t ew = (uint32_t)wg_uniform((int)l0_edge(G, uid));
    const uint32_t edir = ED_DIR(ew);
    const char bdir = (edir == 0 || edir == 2) ? 'R' : 'F';
    const uint32_t buddy = ED_TO(ew);
    LC_GLOBAL NodeGr *gb = &W.gr[buddy];
    const GrLine0 B = gr_line0(gb);
    if (wg_uniform((int)l0_tandem(B, buddy))) return;
    const int buid = wg_uniform(l0_buddy_idx(B, buddy, bdir));
    if (buid == -1) return;
    const bool brev = dir_dest(edir) == 'R';
    // second halves of the two records: cov[4] | mincov mincovqv seq_lo seq_hi | seq_clo seq_chi nkm nkmT
    const lc_u4 nc4 = ldg4((LC_GLOBAL const uint32_t *)gn + 16), nm4 = ldg4((LC_GLOBAL const uint32_t *)gn + 20), ns4 = ldg4((LC_GLOBAL const uint32_t *)gn + 24);
    const lc_u4 bc4 = ldg4((LC_GLOBAL const uint32_t *)gb + 16), bm4 = ldg4((LC_GLOBAL const uint32_t *)gb + 20), bs4 = ldg4((LC_GLOBAL const uint32_t *)gb + 24);
    uint32_t lo = (uint32_t)wg_uniform((int)nm4.z), hi = (uint32_t)wg_uniform((int)nm4.w), clo = (uint32_t)wg_uniform((int)ns4.x), chi = (uint32_t)wg_uniform((int)ns4.y);
    const uint32_t blo = (uint32_t)wg_uniform((int)bm4.z), bhi = (uint32_t)wg_uniform((int)bm4.w);
    const int alen = (int)(hi - lo), blen = (int)(bhi - blo);
    const uint32_t tail = (uint32_t)(blen - (K - 1));
    // seq_reserve: room for `tail` more descriptors behind (dir F) / in front (dir R); else the deque moves to the top of the arena
    {
      const uint32_t front = dir == 'F' ? 0u : tail, back = dir == 'F' ? tail : 0u;
      if (!(lo - clo >= front && chi - hi >= back)) {
        const uint32_t len = hi - lo, cap = 2 * (len + front + back) + 16;
        const uint32_t top = (uint32_t)wg_uniform((int)S.seq_top);
        if (top + cap > LC_CTX(c).C->seq_cap) { if (l0) OVF(c); return; }
        const uint32_t nlo = top + (cap - len - front - back) / 2 + front;
        WG_FOR(i, len) { W.seq[nlo + (uint32_t)i] = W.seq[lo + (uint32_t)i]; }
        if (l0) { S.seq_top = top + cap; gn->seq_clo = top; gn->seq_chi = top + cap; }
        clo = top; chi = top + cap; lo = nlo; hi = nlo + len;
      }
    }
    // merged = astr + bstr[K-1:]  (dir F: append ; dir R: prepend the reverse complement), and Node_t::computeMinCov over what is appended
    if (l0) { S.cn_mn = nm4.x; S.cn_mq = nm4.y; }
    {
      uint32_t mn = 0x7FFFFFFFu, mq = 0x7FFFFFFFu;
      WG_FOR(t, tail) {
        uint32_t d = brev ? W.seq[bhi - 1 - ((uint32_t)(K - 1) + (uint32_t)t)] : W.seq[blo + (uint32_t)(K - 1) + (uint32_t)t];
        if (brev) d ^= 3u;
        if (dir == 'F') W.seq[hi + (uint32_t)t] = d; else { d ^= 3u; W.seq[lo - 1 - (uint32_t)t] = d; }
        int tt, tq; desc_tot(c, d, &tt, &tq);
        if ((uint32_t)tt < mn) mn = (uint32_t)tt;
        if ((uint32_t)tq < mq) mq = (uint32_t)tq;
      }
      WG_FOR(l, LANCET_WG) { if (mn != 0x7FFFFFFFu) { dev_atomic_min((LC_LDS uint32_t *)&S.cn_mn, mn); dev_atomic_min((LC_LDS uint32_t *)&S.cn_mq, mq); mn = 0x7FFFFFFFu; } }
    }
    const uint32_t nmn = (uint32_t)wg_uniform((int)S.cn_mn), nmq = (uint32_t)wg_uniform((int)S.cn_mq);
    const int amer = alen - K + 1, bmer = blen - K + 1;
    if (l0) {
      if (dir == 'F') gn->seq_hi = hi + tail; else gn->seq_lo = lo - tail;
      if (dir == 'F') { if (lo != nm4.z) gn->seq_lo = lo; } else { if (hi != nm4.w) gn->seq_hi = hi; }      // (the deque moved)
      gn->mincov = (int)nmn; gn->mincovqv = (int)nmq;
      gn->nkm = ns4.z + bs4.z; gn->nkmT = ns4.w + bs4.w;
      const float ncv[4] = {__builtin_bit_cast(float, nc4.x), __builtin_bit_cast(float, nc4.y), __builtin_bit_cast(float, nc4.z), __builtin_bit_cast(float, nc4.w)};
      const float bcv[4] = {__builtin_bit_cast(float, bc4.x), __builtin_bit_cast(float, bc4.y), __builtin_bit_cast(float, bc4.z), __builtin_bit_cast(float, bc4.w)};
      for (int q = 0; q < 4; ++q) gn->cov[q] = ((ncv[q] * amer) + (bcv[q] * bmer)) / (amer + bmer);      // Graph.cc:2632-2636
      gb->flags = B.flags | NF_DEAD;
      gn->flags = G.flags | (B.flags & (NF_TUMOR | NF_NORMAL));
      erase_edge_at(c, node, uid);
      const int bcnt = (int)B.necnt;
      for (int i = 0; i < bcnt; ++i) {
        if (i == buid) continue;
        const uint32_t be = l0_edge(B, i);
        uint32_t ndir = ED_DIR(be);
        if (edir == 1 || edir == 2) ndir = flipme(ndir);
        const uint32_t other = ED_TO(be);
        const int cnt = (int)gn->necnt;
        if (cnt >= LC_EMAX) { OVF(c); break; }
        if (other == buddy) { gn->edges[cnt] = ED_MAKE(node, ndir) | (be & (1u << 30)); gn->necnt = cnt + 1; }
        else {
          gn->edges[cnt] = ED_MAKE(other, ndir) | (be & (1u << 30)); gn->necnt = cnt + 1;
          update_edge(c, other, buddy, fliplink(ED_DIR(be)), node, fliplink(ndir));
        }
      }
    }
  }
}
#endif
DEVNI void compress_wg(Ctx &c, int comp) {                            // Graph_t::compress, reference src/Graph.cc:2712-2732
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int nc = pass_candidates_wg(c, comp, GP_MERGE);
#ifndef LANCET_FAT
  WG_LANE0 { S.tmp2 = 0; evt(c, EV_COMPRESS); }
  for (int j = 0; j < nc && !wg_uniform(S.overflow); ++j) {          // (every lane runs the merges: compress_node_wv)
    const uint32_t n = (uint32_t)wg_uniform((int)W.pedges[j]);
    if (wg_uniform((int)W.gr[n].flags) & NF_DEAD) continue;
    compress_node_wv(c, n, 'F');
    compress_node_wv(c, n, 'R');
  }
  WG_SYNC();
#else
  WG_LANE0 {
    S.tmp2 = 0;
    evt(c, EV_COMPRESS);
    for (int j = 0; j < nc && !S.overflow; ++j) {
      const uint32_t n = W.pedges[j];
      if (W.gr[n].flags & NF_DEAD) continue;
      compress_node(c, n, 'F');
      compress_node(c, n, 'R');
    }
  }
#endif
  clean_dead_flags_wg(c, false);
}
DEVNI void remove_low_cov_wg(Ctx &c, int comp) {                      // reference src/Graph.cc:2790-2827 (docompression=true)
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int nc = pass_candidates_wg(c, comp, GP_LOWCOV);
  WG_LANE0 { for (int j = 0; j < nc; ++j) remove_node(c, W.pedges[j]); evt(c, EV_LOWCOV, (uint32_t)nc); }
  if (nc) clean_dead_flags_wg(c, false); else { WG_LANE0 { evt(c, EV_CLEANDEAD, 0); } }
  // Nothing removed: the component was compressed to the end just before (compress is idempotent, hasCycle in between changes
  // nothing), so the reference's unconditional compress() has nothing to merge -- only its trace lines are owed.
  if (nc == 0) { WG_LANE0 { evt(c, EV_COMPRESS); evt(c, EV_CLEANDEAD, 0); S.tmp2 = 0; } }
  else compress_wg(c, comp);
  WG_LANE0 { print_stats(c, comp); }
}
DEVNI void remove_tips_wg(Ctx &c, int comp) {                         // reference src/Graph.cc:2885-2926
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  int round = 0;
  while (true) {
    ++round;
    const int nc = pass_candidates_wg(c, comp, GP_TIPS);
    WG_LANE0 {
      int tips = 0;
      evt(c, EV_TIPS_ROUND, round);
      for (int j = 0; j < nc; ++j) { const uint32_t n = W.pedges[j]; if ((int)W.gr[n].necnt <= 1) { remove_node(c, n); ++tips; } }
      evt(c, EV_TIPS_REMOVED, tips);
      S.tmp3 = tips;
    }
    if (wg_bcast(&S.tmp3) == 0) break;
    compress_wg(c, comp);
    if (wg_bcast(&S.overflow)) break;
  }
  WG_LANE0 { print_stats(c, comp); }
}
DEVNI void remove_short_links_wg(Ctx &c, int comp) {                  // reference src/Graph.cc:2833-2880
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int nc = pass_candidates_wg(c, comp, GP_LINKS);
  WG_LANE0 {
    int links = 0;
    for (int j = 0; j < nc && !S.overflow; ++j) {
      const uint32_t n = W.pedges[j];
      if ((int)W.gr[n].necnt < 2) continue;
      int L = 0, ml = 0; uint8_t motif[64];
      const int sl = n_len(c, n);
      if (sl > (int)LC_CTX(c).C->path_cap) { OVF(c); break; }
      node_string(c, n, W.pseq);
      if (!find_tandems_local(c, W.pseq, sl, S.K - 1, &L, motif, &ml)) { remove_node(c, n); ++links; }
    }
    evt(c, EV_LINKS, links);
    S.tmp3 = links;
  }
  if (wg_bcast(&S.tmp3) && !wg_bcast(&S.overflow)) compress_wg(c, comp);
  WG_LANE0 { print_stats(c, comp); }
}

// markConnectedComponents (reference src/Graph.cc:2252-2336), whole wave.  Only the partition, the numbering of the
// components (in order of their first node in table order) and which of them hold a reference k-mer are observable,
// not the breadth-first order the reference finds them in, so the partition is computed by min-label hooking with
// pointer jumping (O(log n) rounds, every lane busy) instead of a one-lane queue walk that pays a full memory round trip
// per node.  parent[] (node ids; a node's label only ever decreases) is updated with atomics and read at L2.
DEVNI void mark_connected_components_wg(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int M = (int)wg_bcastu(&S.M);
  const uint32_t nodes = LC_CTX(c).C->node_cap + LC_CTX(c).C->special_cap;
  LC_GLOBAL uint32_t *parent = W.scratch, *minpos = W.scratch + nodes, *touch = W.pnodes, *first = W.pedges, *cid = W.nfill;
  WG_LANE0 { evt(c, EV_CC, S.M); }
  // one pass over the 128-byte node records: neighbours as a 16-byte record per table position (degree, three ids; the
  // rare node of higher degree keeps using its record), so that a hooking round reads whole lines
  LC_GLOBAL uint32_t *adj = W.mv;                                          // idle after the build
  WG_FOR(i, M) {
    const uint32_t n = W.order[i];
    LC_GLOBAL const NodeGr &G = W.gr[n];
    const int ne = (int)G.necnt;
    lc_u4 a; a.x = (uint32_t)ne; a.y = ne > 0 ? ED_TO(G.edges[0]) : n; a.z = ne > 1 ? ED_TO(G.edges[1]) : n; a.w = ne > 2 ? ED_TO(G.edges[2]) : n;
    *(lc_u4 *)(adj + 4 * (size_t)i) = a;
    parent[n] = n; minpos[n] = LC_NIL; touch[n] = (G.flags & NF_INMER) ? 2u : 0u;
  }
  WG_SYNC();
  while (true) {
    WG_LANE0 { S.tmp0 = 0; }
    // hook: the smallest label around a node goes to the node and to its current parent
    WG_FOR(i, M) {
      const uint32_t u = W.order[i];
      const lc_u4 a = *(const lc_u4 *)(adj + 4 * (size_t)i);
      const uint32_t pu = ld2(&parent[u]);
      const uint32_t p1 = ld2(&parent[a.y]), p2 = ld2(&parent[a.z]), p3 = ld2(&parent[a.w]);   // (absent neighbours stand in as the node itself)
      uint32_t m = pu;
      if (p1 < m) m = p1;
      if (p2 < m) m = p2;
      if (p3 < m) m = p3;
      if (a.x > 3) { LC_GLOBAL const NodeGr &G = W.gr[u]; for (int e = 3; e < (int)a.x; ++e) { const uint32_t pv = ld2(&parent[ED_TO(G.edges[e])]); if (pv < m) m = pv; } }
      if (m < pu) { dev_atomic_min(&parent[pu], m); dev_atomic_min(&parent[u], m); S.tmp0 = 1; }
    }
    WG_SYNC();
    // jump: every node to its root
    while (true) {
      WG_LANE0 { S.tmp1 = 0; }
      WG_FOR(i, M) {
        const uint32_t u = W.order[i];
        const uint32_t pu = ld2(&parent[u]), gp = ld2(&parent[pu]);
        if (gp != pu) { dev_atomic_min(&parent[u], gp); S.tmp1 = 1; }
      }
      WG_SYNC();
      if (!wg_bcast(&S.tmp1)) break;
    }
    if (!wg_bcast(&S.tmp0)) break;
  }
  // numbering: a component's place is that of its first node in table order
  WG_FOR(i, M) {
    const uint32_t u = W.order[i], r = ld2(&parent[u]);
    dev_atomic_min(&minpos[r], (uint32_t)i);
    if (ld2(&touch[u]) & 2u) dev_atomic_or(&touch[r], 1u);
  }
  WG_SYNC();
  WG_FOR(i, M) { const uint32_t r = ld2(&parent[W.order[i]]); first[i] = (ld2(&minpos[r]) == (uint32_t)i) ? 1u : 0u; }
  WG_LANE0 { first[M] = 0; }
  wg_scan(first, M + 1, S);
  const int numcomp = (int)wg_bcastu(&S.part[LANCET_WG]);
  WG_LANE0 { S.tmp1 = 0; }
  WG_FOR(i, M) {
    const uint32_t r = ld2(&parent[W.order[i]]);
    if (ld2(&minpos[r]) == (uint32_t)i) { cid[r] = first[i] + 1u; if (ld2(&touch[r]) & 1u) dev_atomic_add((LC_LDS uint32_t *)&S.tmp1, 1u); }
  }
  WG_SYNC();
  WG_FOR(i, M) { const uint32_t u = W.order[i]; W.gr[u].comp = (int)cid[ld2(&parent[u])]; }
  WG_SYNC();
  WG_LANE0 {
    S.refcomp = S.tmp1; S.numcomp = numcomp;
    if (LC_CTX(c).C->evt_cap) {
      for (int i = 0; i < M; ++i) { const uint32_t r = ld2(&parent[W.order[i]]); if (ld2(&minpos[r]) == (uint32_t)i && (ld2(&touch[r]) & 1u)) evt(c, EV_CCID, cid[r]); }
    }
    evt(c, EV_CCEND, (uint32_t)numcomp, (uint32_t)S.refcomp);
  }
}

// markRefEnds (reference src/Graph.cc:2028-2228).  The node of the reference k-mer at `offset` is the node
// of the reference pseudo-read's occurrence at that offset (if it is still in the table).
DEV uint32_t special_new(Ctx &c, bool issource, int comp) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  if (S.nspecial >= LC_CTX(c).C->special_cap) { OVF(c); return LC_NIL; }
  uint32_t id = LC_CTX(c).C->node_cap + S.nspecial++;
  char name[24]; int L = 0;
  const char *pre = issource ? "source" : "sink";
  while (*pre) name[L++] = *pre++;
  char digs[12]; int nd = 0; int v = comp; do { digs[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (nd) name[L++] = digs[--nd];
  W.nhash[id] = std_hash_bytes([&](int j) -> int { return (int)(unsigned char)name[j]; }, L);
  for (int q = 0; q < 4; ++q) { W.gr[id].kc[q] = 0; W.gr[id].cov[q] = 0.0f; }
  W.gr[id].flags = issource ? NF_SOURCE : NF_SINK;
  W.gr[id].necnt = 0; W.gr[id].comp = comp; W.gr[id].mincov = 0; W.gr[id].mincovqv = 0; W.gr[id].nqv = LC_NIL; W.gr[id].color = 0;
  W.gr[id].seq_lo = W.gr[id].seq_hi = W.gr[id].seq_clo = W.gr[id].seq_chi = 0; W.gr[id].nkm = 0; W.gr[id].nkmT = 0; W.gr[id].onref = 0;
  return id;
}
// The two scans of markRefEnds (reference src/Graph.cc:2060-2110) over all reference offsets, in parallel:
// mr_src / mr_snk = first / last offset whose node is live, has getTotCov() >= COV_THRESHOLD and is in the component
// (-1 if none); mr_ambs / mr_ambk = the same node qualifies again further on (the reference then gives up).
DEVNI void mark_ref_scan(Ctx &c, int comp) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = S.K;
  const uint32_t ro = W.occ_base[S.R - 1];
  const int nrefk = (S.reflen - K > 0) ? S.reflen - K + 1 : 0;
  WG_LANE0 { S.mr_src = 0x7FFFFFFF; S.mr_snk = 0; S.mr_ambs = 0; S.mr_ambk = 0; }
  {
    uint32_t lmin = 0x7FFFFFFFu, lmax = 0;                    // per lane; one pair of LDS atomics per lane at the end
    const float thr = (float)LC_CTX(c).P->cov_threshold;
    WG_FOR(off, nrefk) {
      const uint32_t t = W.occ[ro + off] & 0x3FFFFFFFu;
      LC_GLOBAL const NodeGr &G = W.gr[t];
      const uint32_t f = G.flags;
      if ((f & NF_DEAD) || !(f & NF_SURV)) continue;
      if (G.cov[0] + G.cov[1] + G.cov[2] + G.cov[3] >= thr && G.comp == comp) {
        if ((uint32_t)off < lmin) lmin = (uint32_t)off;
        if ((uint32_t)(off + 1) > lmax) lmax = (uint32_t)(off + 1);
      }
    }
    WG_FOR(l, LANCET_WG) {
      if (lmax) { dev_atomic_min((LC_LDS uint32_t *)&S.mr_src, lmin); dev_atomic_max((LC_LDS uint32_t *)&S.mr_snk, lmax); }
    }
  }
  WG_SYNC();
  const int so = wg_uniform(S.mr_src), ko = wg_uniform(S.mr_snk) - 1;
  if (so == 0x7FFFFFFF) { WG_LANE0 { S.mr_src = -1; S.mr_snk = -1; } return; }
  const uint32_t sn = W.occ[ro + so] & 0x3FFFFFFFu, kn = W.occ[ro + ko] & 0x3FFFFFFFu;
  WG_FOR(off, nrefk) {
    const uint32_t t = W.occ[ro + off] & 0x3FFFFFFFu;
    if (off > so && t == sn) S.mr_ambs = 1;
    if (off < ko && t == kn) S.mr_ambk = 1;
  }
  WG_LANE0 { S.mr_snk = ko; }
}
// markRefEnds (reference src/Graph.cc): source / sink creation by lane 0, their insertion into the table order by the wave.
DEVNI void mark_ref_ends(Ctx &c, int comp) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = S.K;
  WG_LANE0 {
    S.tmp2 = (int)LC_NIL;                                      // node to insert next (LC_NIL: stop)
    S.trim5 = 0xFFFF; S.trim3 = 0xFFFF;
    S.source = LC_NIL; S.sink = LC_NIL;
    // the scans over the reference offsets ran in mark_ref_scan (same predicates, same outcome order)
    if (S.mr_src >= 0 && S.mr_ambs) evt(c, EV_AMBIG_SRC);
    else if (S.mr_src < 0) evt(c, EV_NOMATCH_SRC);
    else if (S.mr_ambk || S.mr_snk < 0) {
      // the reference has its source_m pointing at the matching k-mer by now and leaves it there (src/Graph.cc:2060-2138):
      // nothing is searched without a sink, but countRefPath still reports (" Found 0 on ref path")
      evt(c, S.mr_ambk ? EV_AMBIG_SNK : EV_NOMATCH_SNK);
      S.source = W.occ[W.occ_base[S.R - 1] + (uint32_t)S.mr_src] & 0x3FFFFFFFu;
    }
    else {
      const uint32_t ro = W.occ_base[S.R - 1];
      const int src_off = S.mr_src, snk_off = S.mr_snk;
      const uint32_t soc = W.occ[ro + src_off];
      const uint32_t src = soc & 0x3FFFFFFFu, src_ori = soc >> 31;
      int ref_dist = snk_off - src_off + K;
      int t3 = S.reflen - snk_off - K;
      S.seq_t5 = src_off; S.seq_len = ref_dist;                           // ref_m->seq = rawseq.substr(source_offset, ref_dist)
      S.trim5 = src_off & 0xFFFF; S.trim3 = t3 & 0xFFFF;
      evt(c, EV_TRIM, src_off, t3, ref_dist);
      // fake source
      uint32_t ns = special_new(c, true, comp);
      if (ns != LC_NIL) {
        uint32_t sourcedir = src_ori ? 1u : 0u;                              // FF, or FR when the k-mer is reversed
        char flip = src_ori ? 'F' : 'R';                                     // Edge_t::flipdir(source_mer.ori_m)
        for (int i = (int)W.gr[src].necnt - 1; i >= 0; --i) {
          uint32_t e = W.gr[src].edges[i];
          if (dir_start(ED_DIR(e)) == flip) {
            uint32_t other = ED_TO(e);
            if (other != src) { remove_edge(c, other, src, fliplink(ED_DIR(e))); erase_edge_at(c, src, i); }
          }
        }
        add_edge(c, ns, src, sourcedir);
        add_edge(c, src, ns, fliplink(sourcedir));
        S.source = ns;
        S.tmp2 = (int)ns;
      }
    }
  }
  uint32_t ins = (uint32_t)wg_bcast(&S.tmp2);
  if (ins == LC_NIL) return;
  order_insert(c, ins);
  if (wg_bcast(&S.overflow)) return;
  WG_LANE0 {
    S.tmp2 = (int)LC_NIL;
    const uint32_t ro = W.occ_base[S.R - 1];
    const uint32_t koc = W.occ[ro + S.mr_snk];
    const uint32_t snk = koc & 0x3FFFFFFFu, snk_ori = koc >> 31;
    // fake sink
    uint32_t nk = special_new(c, false, comp);
    if (nk != LC_NIL) {
      uint32_t sinkdir = snk_ori ? 0u : 3u;                                // RR, or FF when reversed
      char same = snk_ori ? 'R' : 'F';
      for (int i = (int)W.gr[snk].necnt - 1; i >= 0; --i) {
        uint32_t e = W.gr[snk].edges[i];
        if (dir_start(ED_DIR(e)) == same) {
          uint32_t other = ED_TO(e);
          if (other != snk) { remove_edge(c, other, snk, fliplink(ED_DIR(e))); erase_edge_at(c, snk, i); }
        }
      }
      add_edge(c, nk, snk, sinkdir);
      add_edge(c, snk, nk, fliplink(sinkdir));
      S.sink = nk;
      S.tmp2 = (int)nk;
    }
  }
  ins = (uint32_t)wg_bcast(&S.tmp2);
  if (ins == LC_NIL) return;
  order_insert(c, ins);
}

DEVNI bool has_cycle(Ctx &c, bool colored = false) {                                         // reference src/Graph.cc:593-681
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  if (S.source == LC_NIL || S.sink == LC_NIL) return false;
  if (!colored) for (uint32_t i = 0; i < S.M; ++i) { uint32_t n = W.order[i]; if (!(W.gr[n].flags & NF_SPECIAL)) W.gr[n].color = 1; }
  bool ans = false;
  // explicit stack of (node, next edge index, dir)
  LC_GLOBAL uint32_t *st = W.scratch; uint32_t cap = (2 * (LC_CTX(c).C->node_cap + LC_CTX(c).C->special_cap)) / 3;
  for (int pass = 0; pass < 2 && !ans; ++pass) {
    uint32_t sp = 0;
    st[0] = S.source; st[1] = 0; st[2] = (uint32_t)(pass == 0 ? 'F' : 'R'); sp = 1;
    W.gr[S.source].color = 2;
    while (sp && !ans) {
      uint32_t *fr = st + 3 * (sp - 1);
      uint32_t node = fr[0]; char dir = (char)fr[2];
      bool descended = false;
      while (fr[1] < W.gr[node].necnt) {
        uint32_t e = W.gr[node].edges[fr[1]]; ++fr[1];
        if (!is_dir(ED_DIR(e), dir)) continue;
        uint32_t other = ED_TO(e);
        if (W.gr[other].flags & NF_SPECIAL) continue;
        if (W.gr[other].color == 2) { ans = true; break; }
        if (W.gr[other].color == 1) {
          if (sp >= cap) { OVF(c); return false; }
          W.gr[other].color = 2;
          uint32_t *nf = st + 3 * sp; nf[0] = other; nf[1] = 0; nf[2] = (uint32_t)dir_dest(ED_DIR(e)); ++sp;
          descended = true; break;
        }
      }
      if (ans) break;
      if (!descended) { W.gr[node].color = 3; --sp; }
    }
  }
  if (ans) evt(c, EV_CYCLE, S.K);
  return ans;
}

// ---------------------------------------------------------------------------------------------------------
// The cleaned graph in LDS (round 6).  After the first compress a component is a dozen unitigs, and hasCycle and the path search walk it
// on one lane: every step a chain of dependent loads from the node records in HBM (flags, degree, an edge word, the neighbour's flags
// and colour ...), ~35 us per call for a graph that fits a few cache lines.  graph_cache_wg (all lanes) copies what those walks read --
// per table position: node id, degree, the edges as (position of the neighbour, direction, "used by an earlier path" flag), string
// length, the special-node bit -- into the staging area of the per-position pass (idle in the graph phases) with an id -> position
// table beside it; has_cycle_cached / bfs_cached are has_cycle / bfs on that copy: the same visits in the same order.  A table of more
// than GC_MAX live nodes (or an edge to a node that is not in the table) takes the HBM walk as before.  The copy is made right before
// each walk: the passes in between rewrite edge lists, and the repeat scan of a path uses the same LDS.
// ---------------------------------------------------------------------------------------------------------
#define GC_MAX 48
#define GC_HASH 128
#define GC_OFF_ID 0
#define GC_OFF_HASH (GC_OFF_ID + 4 * GC_MAX)
#define GC_OFF_E (GC_OFF_HASH + 4 * GC_HASH)
#define GC_OFF_LEN (GC_OFF_E + 2 * 12 * GC_MAX)
#define GC_OFF_NE (GC_OFF_LEN + 2 * GC_MAX)
#define GC_OFF_FL (GC_OFF_NE + GC_MAX)
#define GC_OFF_COL (GC_OFF_FL + GC_MAX)
#define GC_OFF_ST (GC_OFF_COL + GC_MAX)
#define GC_BYTES (GC_OFF_ST + 3 * (GC_MAX + 2))
#define GC_TPOS(e) ((uint32_t)(e) & 63u)
#define GC_DIR(e) (((uint32_t)(e) >> 6) & 3u)
#define GC_FLAG(e) (((uint32_t)(e) >> 8) & 1u)
DEV uint32_t gc_lookup(const volatile LC_LDS uint32_t *ids, const volatile LC_LDS uint32_t *hash, uint32_t id) {   // position of node id in the table order, or LC_NIL
  uint32_t h = (id * 0x9E3779B1u) >> 25;
  for (int t = 0; t < GC_HASH; ++t) {
    const uint32_t v = hash[h];
    if (v == 0u) return LC_NIL;
    if (ids[v - 1u] == id) return v - 1u;
    h = (h + 1u) & (GC_HASH - 1u);
  }
  return LC_NIL;
}
DEVNI bool graph_cache_wg(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  static_assert(GC_BYTES <= (int)sizeof(S.lbytes) && GC_MAX <= 64 && (GC_HASH & (GC_HASH - 1)) == 0 && GC_HASH >= 2 * GC_MAX, "graph cache in the staging area");
  const int M = (int)wg_bcastu(&S.M);
  if (M > GC_MAX || M < 1) return false;
  volatile LC_LDS uint8_t *base = &S.lbytes[0];
  volatile LC_LDS uint32_t *ids = (volatile LC_LDS uint32_t *)(base + GC_OFF_ID), *hash = (volatile LC_LDS uint32_t *)(base + GC_OFF_HASH);
  volatile LC_LDS uint16_t *ge = (volatile LC_LDS uint16_t *)(base + GC_OFF_E), *glen = (volatile LC_LDS uint16_t *)(base + GC_OFF_LEN);
  volatile LC_LDS uint8_t *gne = base + GC_OFF_NE, *gfl = base + GC_OFF_FL;
  WG_FOR(i, GC_HASH) { hash[i] = 0u; }
  WG_LANE0 { S.gc_ok = 1; }
  WG_FOR(i, M) {
    const uint32_t id = W.order[i];
    ids[i] = id;
    uint32_t h = (id * 0x9E3779B1u) >> 25;
    while (dev_atomic_cas32((LC_LDS uint32_t *)&hash[h], 0u, (uint32_t)i + 1u) != 0u) h = (h + 1u) & (GC_HASH - 1u);
  }
  WG_SYNC();
  WG_FOR(i, M) {
    const uint32_t id = ids[i];
    LC_GLOBAL const NodeGr *g = &W.gr[id];
    const GrLine0 G = gr_line0(g);
    const lc_u4 h1 = ldg4((LC_GLOBAL const uint32_t *)g + 20);          // mincov mincovqv seq_lo seq_hi
    const bool sp = (G.flags & NF_SPECIAL) != 0;
    const uint32_t len = sp ? 0u : h1.w - h1.z;
    bool ok = G.necnt <= 12u && len <= 0xFFFFu;
#define LC_X(j, e) do { if ((uint32_t)(j) < G.necnt) { const uint32_t tp = gc_lookup(ids, hash, ED_TO(e)); if (tp == LC_NIL) ok = false; \
                        ge[12 * i + (j)] = (uint16_t)((tp & 63u) | (ED_DIR(e) << 6) | (ED_FLAG(e) << 8)); } } while (0)
    LC_L0_EACH(G, LC_X);
#undef LC_X
    gne[i] = (uint8_t)G.necnt; gfl[i] = sp ? 1 : 0; glen[i] = (uint16_t)len;
    if (!ok) S.gc_ok = 0;
  }
  return wg_bcast(&S.gc_ok) != 0;
}
// has_cycle on the copy (lane 0): colours and the stack of (position, next edge, direction) in LDS too
DEVNI bool has_cycle_cached(Ctx &c) {
  LC_WS &S = LC_SREF(c);
  if (S.source == LC_NIL || S.sink == LC_NIL) return false;
  volatile LC_LDS uint8_t *base = &S.lbytes[0];
  const volatile LC_LDS uint32_t *ids = (const volatile LC_LDS uint32_t *)(base + GC_OFF_ID), *hash = (const volatile LC_LDS uint32_t *)(base + GC_OFF_HASH);
  const volatile LC_LDS uint16_t *ge = (const volatile LC_LDS uint16_t *)(base + GC_OFF_E);
  const volatile LC_LDS uint8_t *gne = base + GC_OFF_NE, *gfl = base + GC_OFF_FL;
  volatile LC_LDS uint8_t *col = base + GC_OFF_COL, *st = base + GC_OFF_ST;
  const int M = (int)S.M;
  for (int i = 0; i < M; ++i) col[i] = gfl[i] ? 0 : 1;
  const uint32_t src = gc_lookup(ids, hash, S.source);
  if (src == LC_NIL) return false;                                  // (the source is in the table whenever it exists)
  bool ans = false;
  for (int pass = 0; pass < 2 && !ans; ++pass) {
    int sp = 0;
    st[0] = (uint8_t)src; st[1] = 0; st[2] = (uint8_t)(pass == 0 ? 'F' : 'R'); sp = 1;
    col[src] = 2;
    while (sp && !ans) {
      volatile LC_LDS uint8_t *fr = st + 3 * (sp - 1);
      const uint32_t node = fr[0]; const char dir = (char)fr[2];
      bool descended = false;
      uint32_t ei = fr[1]; const uint32_t ne = gne[node];
      while (ei < ne) {
        const uint32_t e = ge[12 * node + ei]; ++ei;
        if (!is_dir(GC_DIR(e), dir)) continue;
        const uint32_t other = GC_TPOS(e);
        if (gfl[other]) continue;
        const uint32_t oc = col[other];
        if (oc == 2) { ans = true; break; }
        if (oc == 1) {
          col[other] = 2;
          fr[1] = (uint8_t)ei;
          volatile LC_LDS uint8_t *nf = st + 3 * sp; nf[0] = (uint8_t)other; nf[1] = 0; nf[2] = (uint8_t)dir_dest(GC_DIR(e)); ++sp;
          descended = true; break;
        }
      }
      if (ans) break;
      if (!descended) { col[node] = 3; --sp; }
    }
  }
  if (ans) evt(c, EV_CYCLE, S.K);
  return ans;
}

// ---------------------------------------------------------------------------------------------------------
// path enumeration: Graph_t::bfs (reference src/Graph.cc:1299-1425), a FIFO over whole paths.  Paths are
// queue entries with parent links; the winner is the first-dequeued path with the most unflagged edges.
// Returns queue index of the best path or LC_NIL.
// ---------------------------------------------------------------------------------------------------------
DEV bool path_has_node(const Ctx &c, uint32_t idx, uint32_t node) {
  const BfsEntry *Q = LC_CTX(c).W->queue;
  for (uint32_t i = idx; i != LC_NIL; i = Q[i].parent) if (Q[i].node == node) return true;
  return false;
}
DEVNI uint32_t bfs(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  LC_GLOBAL BfsEntry *Q = W.queue;
  const uint32_t cap = LC_CTX(c).C->queue_cap;
  int reflen = S.seq_len;
  uint32_t qh = 0, qt = 0;
  const bool tracing = LC_CTX(c).C->evt_cap != 0;
  S.bfs_dfs = 0;
  Q[qt].parent = LC_NIL; Q[qt].node = S.source; Q[qt].edge = LC_NIL; Q[qt].len = S.K; Q[qt].score = 0; Q[qt].dir = 'F'; Q[qt].bits = 1; ++qt;
  uint32_t best = LC_NIL; int complete = 0; int visit = 0;
  while (qh < qt) {
    ++visit;
    if (LC_CTX(c).P->dfs_limit && visit > LC_CTX(c).P->dfs_limit) { evt(c, EV_DFSLIMIT); S.bfs_dfs = 1; break; }
    uint32_t idx = qh++;
    BfsEntry cur = Q[idx];
    if (cur.node == S.sink && (cur.bits & 1) == 0) {
      ++complete;
      if (best == LC_NIL) best = idx; else if (cur.score > Q[best].score) best = idx;
    } else if (cur.len > reflen + LC_CTX(c).P->max_indel_len) {
    } else {
      int cnt = (int)W.gr[cur.node].necnt;
      for (int i = 0; i < cnt; ++i) {
        uint32_t e = W.gr[cur.node].edges[i];
        if (!is_dir(ED_DIR(e), (char)cur.dir)) continue;
        uint32_t other = ED_TO(e);
        // Path_t::hasCycle: only ever printed (the -v trace: per path and in the search summary) -- a walk up the whole partial
        // path per expansion, i.e. quadratic in the path's nodes, so it is evaluated only when the trace is being recorded
        if (tracing && !(Q[idx].bits & 2) && path_has_node(c, idx, other)) Q[idx].bits |= 2;
        if (qt >= cap) { OVF(c); return LC_NIL; }
        BfsEntry ne;
        ne.parent = idx; ne.node = other; ne.edge = (cur.node << 4) | (uint32_t)i; ne.dir = (uint8_t)dir_dest(ED_DIR(e));
        ne.len = cur.len + n_strlen(c, other) - S.K + 1;
        uint32_t ef = ED_FLAG(e);
        ne.bits = (uint8_t)(((cur.bits & 1) & ef) | (Q[idx].bits & 2));
        ne.score = (uint16_t)(cur.score + (ef == 0 ? 1 : 0));
        Q[qt++] = ne;
      }
    }
  }
  if (complete == 0) best = LC_NIL;
  return best;
}
// bfs on the copy graph_cache_wg made (lane 0): the queue stays where it is, what an expansion reads of the graph comes out of LDS
DEVNI uint32_t bfs_cached(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  LC_GLOBAL BfsEntry *Q = W.queue;
  const uint32_t cap = LC_CTX(c).C->queue_cap;
  volatile LC_LDS uint8_t *base = &S.lbytes[0];
  const volatile LC_LDS uint32_t *ids = (const volatile LC_LDS uint32_t *)(base + GC_OFF_ID), *hash = (const volatile LC_LDS uint32_t *)(base + GC_OFF_HASH);
  const volatile LC_LDS uint16_t *ge = (const volatile LC_LDS uint16_t *)(base + GC_OFF_E), *glen = (const volatile LC_LDS uint16_t *)(base + GC_OFF_LEN);
  const volatile LC_LDS uint8_t *gne = base + GC_OFF_NE;
  const int reflen = S.seq_len, K = S.K;
  const int max_indel = LC_CTX(c).P->max_indel_len, dfs_limit = LC_CTX(c).P->dfs_limit;
  const uint32_t sink = S.sink;
  uint32_t qh = 0, qt = 0;
  const bool tracing = LC_CTX(c).C->evt_cap != 0;
  S.bfs_dfs = 0;
  Q[qt].parent = LC_NIL; Q[qt].node = S.source; Q[qt].edge = LC_NIL; Q[qt].len = K; Q[qt].score = 0; Q[qt].dir = 'F'; Q[qt].bits = 1; ++qt;
  uint32_t best = LC_NIL; int complete = 0; int visit = 0; uint32_t best_score = 0;
  while (qh < qt) {
    ++visit;
    if (dfs_limit && visit > dfs_limit) { evt(c, EV_DFSLIMIT); S.bfs_dfs = 1; break; }
    const uint32_t idx = qh++;
    const BfsEntry cur = Q[idx];
    if (cur.node == sink && (cur.bits & 1) == 0) {
      ++complete;
      if (best == LC_NIL || cur.score > best_score) { best = idx; best_score = cur.score; }
    } else if (cur.len > reflen + max_indel) {
    } else {
      const uint32_t pos = gc_lookup(ids, hash, cur.node);
      if (pos == LC_NIL) { OVF(c); return LC_NIL; }                   // (cannot happen: every queue entry's node came out of the table)
      const int cnt = (int)gne[pos];
      uint8_t bits2 = cur.bits & 2;
      for (int i = 0; i < cnt; ++i) {
        const uint32_t e = ge[12 * pos + (uint32_t)i];
        if (!is_dir(GC_DIR(e), (char)cur.dir)) continue;
        const uint32_t tp = GC_TPOS(e), other = ids[tp];
        // Path_t::hasCycle: only ever printed (see bfs)
        if (tracing && !bits2 && path_has_node(c, idx, other)) { bits2 = 2; Q[idx].bits |= 2; }
        if (qt >= cap) { OVF(c); return LC_NIL; }
        BfsEntry ne;
        ne.parent = idx; ne.node = other; ne.edge = (cur.node << 4) | (uint32_t)i; ne.dir = (uint8_t)dir_dest(GC_DIR(e));
        ne.len = cur.len + (int)glen[tp] - K + 1;
        const uint32_t ef = GC_FLAG(e);
        ne.bits = (uint8_t)(((cur.bits & 1) & ef) | bits2);
        ne.score = (uint16_t)(cur.score + (ef == 0 ? 1 : 0));
        Q[qt++] = ne;
      }
    }
  }
  if (complete == 0) best = LC_NIL;
  return best;
}
// unpack the best path into W.pnodes / W.pedges ; returns number of nodes
DEVNI int path_unpack(Ctx &c, uint32_t best) {
  LC_GLOBAL Work &W = *LC_CTX(c).W; const BfsEntry *Q = W.queue;
  int n = 0;
  for (uint32_t i = best; i != LC_NIL; i = Q[i].parent) ++n;
  int k = n;
  for (uint32_t i = best; i != LC_NIL; i = Q[i].parent) { --k; W.pnodes[k] = Q[i].node; W.pedges[k] = Q[i].edge; }
  return n;   // pedges[j] (j>=1) is the edge leading into node j
}
DEV void path_flag_edges(Ctx &c, int n, uint32_t v) {
  LC_GLOBAL Work &W = *LC_CTX(c).W;
  for (int j = 1; j < n; ++j) {
    uint32_t owner = W.pedges[j] >> 4, ei = W.pedges[j] & 15u;
    LC_GLOBAL uint32_t *e = &W.gr[owner].edges[ei];
    *e = (*e & ~(1u << 30)) | (v << 30);
  }
}
// Path_t::str + covDistr (reference src/Path.cc:69-175): string codes + descriptor per base ; returns length
DEVNI int path_string(Ctx &c, int n) {
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_WS &S = LC_SREF(c);
  const int K = S.K;
  int len = 0;
  uint32_t e1 = W.gr[(W.pedges[1] >> 4)].edges[(W.pedges[1] & 15u)];
  char dir = dir_start(ED_DIR(e1));
  for (int i = 0; i < n; ++i) {
    uint32_t nd = W.pnodes[i];
    if (!(W.gr[nd].flags & NF_SPECIAL)) {
      uint32_t lo = W.gr[nd].seq_lo, hi = W.gr[nd].seq_hi;
      int L = (int)(hi - lo);
      int from = len > 0 ? K - 1 : 0;
      if (len + L - from > (int)LC_CTX(c).C->path_cap) { OVF(c); return 0; }
      for (int j = from; j < L; ++j) {
        uint32_t d = (dir == 'R') ? (W.seq[hi - 1 - (uint32_t)j] ^ 3u) : W.seq[lo + (uint32_t)j];
        W.pdesc[len] = d; W.pseq[len] = (uint8_t)SD_BASE(d); ++len;
      }
    }
    if (i + 1 < n) {
      uint32_t e = W.gr[(W.pedges[i + 1] >> 4)].edges[(W.pedges[i + 1] & 15u)];
      dir = dir_dest(ED_DIR(e));
    }
  }
  return len;
}
// The same string and descriptors with all lanes: contribution length of every path node (special nodes none, the first
// real node whole, the others without the K-1 overlap), exclusive scan, then one lane per output base.
DEVNI int path_string_wg(Ctx &c, int n) {
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_WS &S = LC_SREF(c);
  const int K = S.K;
  const int dcap = 7 * ((int)LC_CTX(c).C->max_w + 2);
  if (2 * (n + 1) > dcap || n < 2) { WG_LANE0 { S.ps_len = path_string(c, n); } return wg_bcast(&S.ps_len); }
  if (n <= 64) {
    // (round 6) a path of up to 64 nodes -- all but the odd one: what the per-base loop looks up per node (offset in the string, deque bounds,
    // direction) is staged in LDS (the staging area of the per-position pass, idle here) by one lane per node; a base then costs a binary search
    // in LDS and ONE load from the descriptor arena instead of the ~8 dependent loads from HBM of the form below.  Same string, same descriptors.
    volatile LC_LDS uint8_t *lb = &S.lbytes[0];
    volatile LC_LDS uint32_t *loff = (volatile LC_LDS uint32_t *)lb, *llen = loff + 66, *lslo = llen + 64, *lshi = lslo + 64;
    volatile LC_LDS uint8_t *ldir = (volatile LC_LDS uint8_t *)(lshi + 64);
    static_assert(4 * (66 + 3 * 64) + 64 <= (int)sizeof(S.lbytes), "path nodes staged in LDS");
    WG_LANE0 { S.ps_first = 0x7FFFFFFF; }
    WG_FOR(i, n) {
      const uint32_t nd = W.pnodes[i];
      const uint32_t pe = W.pedges[i == 0 ? 1 : i];
      LC_GLOBAL const uint32_t *g = (LC_GLOBAL const uint32_t *)&W.gr[nd];
      const uint32_t fl = g[0]; const lc_u4 h1 = ldg4(g + 20);         // mincov mincovqv seq_lo seq_hi
      const uint32_t e = W.gr[pe >> 4].edges[pe & 15u];
      const bool sp = (fl & NF_SPECIAL) != 0;
      if (!sp) dev_atomic_min((LC_LDS uint32_t *)&S.ps_first, (uint32_t)i);
      ldir[i] = (uint8_t)(i == 0 ? dir_start(ED_DIR(e)) : dir_dest(ED_DIR(e)));
      lslo[i] = h1.z; lshi[i] = h1.w; llen[i] = sp ? 0xFFFFFFFFu : h1.w - h1.z;
    }
    WG_SYNC();
    const int f = wg_uniform(S.ps_first);
    WG_FOR(i, n + 1) {
      uint32_t sum = 0;
      for (int j = 0; j < i; ++j) { const uint32_t L = llen[j]; if (L != 0xFFFFFFFFu) sum += (j == f ? L : L - (uint32_t)(K - 1)); }
      loff[i] = sum;
    }
    WG_SYNC();
    const int plen = (int)wg_uniform((int)loff[n]);
    if (plen > (int)LC_CTX(c).C->path_cap) { WG_LANE0 { OVF(c); } return 0; }
    WG_FOR(x, plen) {
      int lo = 0, hi = n - 1;                                    // last node whose offset is <= x and that contributes
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)loff[mid] <= x) lo = mid; else hi = mid - 1; }
      while (lo > 0 && loff[lo + 1] == loff[lo]) --lo;           // (special nodes and empty contributions share an offset)
      const uint32_t slo = lslo[lo], shi = lshi[lo];
      const uint32_t j = (uint32_t)(x - (int)loff[lo]) + (lo == f ? 0u : (uint32_t)(K - 1));
      const uint32_t d = (ldir[lo] == (uint8_t)'R') ? (W.seq[shi - 1 - j] ^ 3u) : W.seq[slo + j];
      W.pdesc[x] = d; W.pseq[x] = (uint8_t)SD_BASE(d);
    }
    WG_SYNC();
    return plen;
  }
  LC_GLOBAL uint32_t *off = (LC_GLOBAL uint32_t *)W.dp, *pdir = off + (n + 1);
  WG_LANE0 { S.ps_first = 0x7FFFFFFF; off[n] = 0; }
  WG_FOR(i, n) {
    const uint32_t nd = W.pnodes[i];
    if (!(W.gr[nd].flags & NF_SPECIAL)) dev_atomic_min((LC_LDS uint32_t *)&S.ps_first, (uint32_t)i);
    const uint32_t pe = W.pedges[i == 0 ? 1 : i];
    const uint32_t e = W.gr[pe >> 4].edges[pe & 15u];
    pdir[i] = (uint32_t)(i == 0 ? dir_start(ED_DIR(e)) : dir_dest(ED_DIR(e)));
  }
  WG_SYNC();
  const int f = wg_uniform(S.ps_first);
  WG_FOR(i, n) {
    const uint32_t nd = W.pnodes[i];
    const int L = n_len(c, nd);
    off[i] = (W.gr[nd].flags & NF_SPECIAL) ? 0u : (uint32_t)(i == f ? L : L - (K - 1));
  }
  WG_SYNC();
  wg_scan(off, n + 1, S, S.part2);
  const int plen = (int)wg_uniform((int)S.part2[LANCET_WG]);
  if (plen > (int)LC_CTX(c).C->path_cap) { WG_LANE0 { OVF(c); } return 0; }
  WG_FOR(x, plen) {
    int lo = 0, hi = n - 1;                                    // last node whose offset is <= x and that contributes
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)off[mid] <= x) lo = mid; else hi = mid - 1; }
    while (lo > 0 && off[lo + 1] == off[lo]) --lo;             // (special nodes and empty contributions share an offset)
    const uint32_t nd = W.pnodes[lo];
    const uint32_t slo = W.gr[nd].seq_lo, shi = W.gr[nd].seq_hi;
    const uint32_t j = (uint32_t)(x - (int)off[lo]) + (lo == f ? 0u : (uint32_t)(K - 1));
    const uint32_t d = (pdir[lo] == (uint32_t)'R') ? (W.seq[shi - 1 - j] ^ 3u) : W.seq[slo + j];
    W.pdesc[x] = d; W.pseq[x] = (uint8_t)SD_BASE(d);
  }
  WG_SYNC();
  return plen;
}
DEV uint32_t path_contig(const Ctx &c, int n, int pos) {            // Path_t::pathcontig, reference src/Path.cc:291-314
  const LC_GLOBAL Work &W = *LC_CTX(c).W;
  int cur = 0;
  for (int i = 0; i < n; ++i) {
    uint32_t nd = W.pnodes[i];
    if (W.gr[nd].flags & NF_SPECIAL) continue;
    int span = n_len(c, nd);
    if (cur + span >= pos) return nd;
    cur += span - LC_SREF(c).K + 1;
  }
  return LC_NIL;
}
DEV bool status_cnt_T(const Ctx &c, uint32_t n) {                   // Node_t::isStatusCnt('T'), reference src/Node.cc:423-440
  double pr = (double)LC_CTX(c).W->gr[n].nkmT / (double)LC_CTX(c).W->gr[n].nkm;
  return pr > 0.8;
}

// ---------------------------------------------------------------------------------------------------------
// global_align_aff (reference src/align.cc:235-364): full Gotoh, MATCH 2 / MISMATCH -4 / OPEN -8 / EXTEND -1,
// the reference's tie rules; anti-diagonal wavefront across the workgroup, traceback by lane 0.
// The traceback matrix is stored anti-diagonal-major: cell (i, j) at (i + j) * (n + 1) + i.  The systolic fill works on one
// anti-diagonal per step with lane = row, so a wave writes 64 consecutive bytes (one request) instead of 64 bytes in 64
// different lines -- these stores were more than half of the kernel's write requests.
#define LC_TB(i, j, n) ((size_t)((i) + (j)) * (size_t)((n) + 1) + (size_t)(i))
// tb byte: M.tb [1:0] (0 '\\', 1 '<', 2 '^', 3 '*') ; X.tb [3:2] (0 '<', 1 '-', 2 '*') ; Y.tb [5:4] (0 '^', 1 '|', 2 '*')
// ---------------------------------------------------------------------------------------------------------
DEV void align_fill_arrays(Ctx &c, LC_GLOBAL const uint8_t *Sx, int n, LC_GLOBAL const uint8_t *Tx, int m) {
  LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int A = (int)LC_CTX(c).C->max_w + 2;
  int32_t *Mb[3] = {W.dp, W.dp + A, W.dp + 2 * A};
  int32_t *Xb[2] = {W.dp + 3 * A, W.dp + 4 * A};
  int32_t *Yb[2] = {W.dp + 5 * A, W.dp + 6 * A};
  // boundary rows/cols of the traceback
  WG_FOR(j, m + 1) { W.tb[LC_TB(0, j, n)] = (uint8_t)((j == 0 ? 3 : 2) | (0 << 2) | (2 << 4)); }       // M[0][j] '^' ; Y[0][j] '*'
  WG_FOR(i, n + 1) { if (i > 0) W.tb[LC_TB(i, 0, n)] = (uint8_t)(1 | (2 << 2) | (0 << 4)); }   // M[i][0] '<' ; X[i][0] '*'
  WG_LANE0 { Mb[0][0] = 0; Mb[1][0] = -9; Xb[1][0] = -9; Mb[1][1] = -9; Yb[1][1] = -9; }   // diagonals 0 and 1
  WG_SYNC();
  for (int d = 2; d <= n + m; ++d) {
    int32_t *Mc = Mb[d % 3], *Mp = Mb[(d + 2) % 3], *Mpp = Mb[(d + 1) % 3];
    int32_t *Xc = Xb[d & 1], *Xp = Xb[(d + 1) & 1], *Yc = Yb[d & 1], *Yp = Yb[(d + 1) & 1];
    int ilo = d - m; if (ilo < 1) ilo = 1;
    int ihi = d - 1; if (ihi > n) ihi = n;
    WG_FOR(t, ihi - ilo + 1) {
      int i = ilo + t, j = d - i;
      int xa = Xp[i - 1] + (-1), xb = Mp[i - 1] + (-8);
      int xs, xt; if (xa > xb) { xs = xa; xt = 1; } else { xs = xb; xt = 0; }
      int ya = Yp[i] + (-1), yb = Mp[i] + (-8);
      int ys, yt; if (ya > yb) { ys = ya; yt = 1; } else { ys = yb; yt = 0; }
      int ms = Mpp[i - 1] + (Sx[i - 1] == Tx[j - 1] ? 2 : -4), mt = 0;
      if (xs > ms) { ms = xs; mt = 1; }
      if (ys > ms) { ms = ys; mt = 2; }
      Mc[i] = ms; Xc[i] = xs; Yc[i] = ys;
      W.tb[LC_TB(i, j, n)] = (uint8_t)(mt | (xt << 2) | (yt << 4));
    }
    WG_LANE0 {
      if (d <= m) { Mc[0] = -8 - d; Xc[0] = -8 - d; }      // (0, d)
      if (d <= n) { Mc[d] = -8 - d; Yc[d] = -8 - d; }      // (d, 0)
    }
    WG_SYNC();
  }
}
#ifndef LANCET_WAVE_EMU
// GPU form: systolic over the wave.  Lane l owns rows l+1, l+65, ... (GMAX of them: 10 for a window of up to 640 bases); cell (i,j)
// is computed at step t = i + j, so everything a cell needs was produced one or two steps earlier by the lane above
// (wave shuffles) or by the lane itself.  The score diagonals never touch memory; only the traceback bytes do.
template <int GMAX>
DEVNI void align_fill_g(Ctx &c, LC_GLOBAL const uint8_t *Sx, int n, LC_GLOBAL const uint8_t *Tx, int m) {
  static_assert(LANCET_WG == 64, "one wave per window");
  LC_GLOBAL Work &W = *LC_CTX(c).W;
  if (threadIdx.x == 0) LC_SREF(c).al_band = 0;
  const int lane = (int)threadIdx.x;
  if (lane < 64) {                                                   // (helper waves of the fat form only wait at the barrier below)
  for (int j = lane; j < m + 1; j += 64) W.tb[LC_TB(0, j, n)] = (uint8_t)((j == 0 ? 3 : 2) | (0 << 2) | (2 << 4));           // M[0][j] '^' ; Y[0][j] '*'
  for (int i = lane + 1; i < n + 1; i += 64) W.tb[LC_TB(i, 0, n)] = (uint8_t)(1 | (2 << 2) | (0 << 4));     // M[i][0] '<' ; X[i][0] '*'
  const int G = (n + 63) / 64;
  // per row: M, X (packed 16+16 in A) and Y, M(i-1,j-1) (packed in B); all scores fit 16 bits (|score| < 4*1280)
  int A[GMAX], Bv[GMAX], sg[GMAX];
  auto lo16 = [](int v) -> int { return (int)(short)(v & 0xFFFF); };
  auto hi16 = [](int v) -> int { return v >> 16; };
  auto pack = [](int lo, int hi) -> int { return (lo & 0xFFFF) | (hi << 16); };
#pragma unroll
  for (int g = 0; g < GMAX; ++g) {
    int i = g * 64 + lane + 1;
    A[g] = pack(-8 - i, 0);                                          // M(i,0) = GAP_OPEN + i*GAP_EXTEND ; X(i,0) unused
    Bv[g] = pack(-8 - i, (i - 1 == 0) ? 0 : -8 - (i - 1));           // Y(i,0) ; M(i-1,0)
    sg[g] = (i <= n) ? (int)Sx[i - 1] : 255;
  }
  for (int t = 2; t <= n + m; ++t) {
    // groups are visited from the last row block to the first, so that the values of block g-1 (and of the lane
    // above) are still those of step t-1 when block g needs them
#pragma unroll
    for (int gg = 0; gg < GMAX; ++gg) {
      const int g = GMAX - 1 - gg;
      if (g < G) {
        int i = g * 64 + lane + 1, j = t - i;
        int nb = __shfl_up(A[g], 1);
        if (g > 0) { int e = __shfl(A[g > 0 ? g - 1 : 0], 63); if (lane == 0) nb = e; }
        else if (lane == 0) nb = pack(-8 - j, -8 - j);                 // row 0: M(0,j) = X(0,j)
        int nbM = lo16(nb), nbX = hi16(nb);
        bool active = (i <= n) && (j >= 1) && (j <= m);
        int bv = Bv[g];
        if (active) {
          int xa = nbX - 1, xb = nbM - 8;
          int xs, xt; if (xa > xb) { xs = xa; xt = 1; } else { xs = xb; xt = 0; }
          int ya = lo16(bv) - 1, yb = lo16(A[g]) - 8;
          int ys, yt; if (ya > yb) { ys = ya; yt = 1; } else { ys = yb; yt = 0; }
          int ms = hi16(bv) + (sg[g] == (int)Tx[j - 1] ? 2 : -4), mt = 0;
          if (xs > ms) { ms = xs; mt = 1; }
          if (ys > ms) { ms = ys; mt = 2; }
          A[g] = pack(ms, xs);
          bv = pack(ys, hi16(bv));
          W.tb[LC_TB(i, j, n)] = (uint8_t)(mt | (xt << 2) | (yt << 4));
        }
        if (j >= 1) bv = pack(lo16(bv), nbM);                            // M(i-1,j) is next step's M(i-1,j-1)
        Bv[g] = bv;
      }
    }
  }
  }
  WG_SYNC();
}
DEV void align_fill(Ctx &c, LC_GLOBAL const uint8_t *Sx, int n, LC_GLOBAL const uint8_t *Tx, int m) {
  if (n <= LC_MAXW_DEFAULT) { align_fill_g<(LC_MAXW_DEFAULT + 63) / 64>(c, Sx, n, Tx, m); return; }     // ten rows per lane, in registers
  // A longer window (--window-size above 640) whose band could not be certified: the diagonals in the slot's work space instead of
  // 16 rows per lane in registers (which cost the whole kernel a wave per SIMD: 136 VGPRs).  Rare twice over, and exact all the same.
  WG_LANE0 { LC_SREF(c).al_band = 0; }
  align_fill_arrays(c, Sx, n, Tx, m);
}
#else
DEV void align_fill(Ctx &c, LC_GLOBAL const uint8_t *Sx, int n, LC_GLOBAL const uint8_t *Tx, int m) { LC_SREF(c).al_band = 0; align_fill_arrays(c, Sx, n, Tx, m); }
#endif
// ---------------------------------------------------------------------------------------------------------
// The same dynamic programme restricted to a band of 128 diagonals (offsets j - i in [lo, lo + 127]) around the two corners:
// an anti-diagonal t = i + j meets 64 cells of the band (the offsets of t's parity), one per lane; the cell's three
// neighbours were computed one or two steps earlier by the lane itself or by the lane next to it.  65 bytes of traceback per
// anti-diagonal instead of n + 1, ~10x fewer cell updates for a 600 x 600 problem.
// The result is the full matrix's whenever the best alignment stays inside the band, and that is CERTIFIED: an alignment
// that leaves the band needs gaps of at least hi + 1 in one direction and hi + 1 - d back (resp. 1 - lo and 1 - lo + d),
// which bounds its score; if the score found in the band is strictly above that bound, no alignment outside can equal it, so
// every cell on an optimal path -- and every tie the reference's rules break -- has the same value in both programmes.
// Returns false (nothing usable) when the band cannot be certified or does not hold both corners: align_fill then runs.
// ---------------------------------------------------------------------------------------------------------
#define LC_BNEG (-(1 << 24))
template <bool B> struct LcBool { static constexpr bool value = B; };
DEV void band_cell(int i, int j, int sch, int tch, int dM, int uM, int uX, int lM, int lY, int *oM, int *oX, int *oY, uint8_t *otb) {
  const int xa = uX - 1, xb = uM - 8;
  int xs, xt; if (xa > xb) { xs = xa; xt = 1; } else { xs = xb; xt = 0; }
  const int ya = lY - 1, yb = lM - 8;
  int ys, yt; if (ya > yb) { ys = ya; yt = 1; } else { ys = yb; yt = 0; }
  int ms = dM + (sch == tch ? 2 : -4), mt = 0;
  if (xs > ms) { ms = xs; mt = 1; }
  if (ys > ms) { ms = ys; mt = 2; }
  (void)i; (void)j;
  *oM = ms; *oX = xs; *oY = ys; *otb = (uint8_t)(mt | (xt << 2) | (yt << 4));
}
DEVNI bool align_fill_band(Ctx &c, LC_GLOBAL const uint8_t *Sx, int n, LC_GLOBAL const uint8_t *Tx, int m) {
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_WS &S = LC_SREF(c);
  const int d = m - n;
  const int w = (127 - (d < 0 ? -d : d)) / 2;
  if (w < 8 || n < 1 || m < 1) return false;
  const int lo = (d < 0 ? d : 0) - w, hi = lo + 127;
  if ((size_t)(n + m + 2) * 64 > (size_t)(LC_CTX(c).C->max_w + LC_CTX(c).C->path_cap + 4) * (LC_CTX(c).C->max_w + 2)) return false;
  WG_LANE0 { S.al_band = 1; S.al_lo = lo; S.al_score = LC_BNEG; }
#ifndef LANCET_WAVE_EMU
  {
    const int lane = (int)threadIdx.x;
    LC_GLOBAL uint8_t *tbp = W.tb;
    // A lane owns two neighbouring offsets, E = lo + 2 * lane (met on the steps of one parity) and O = E + 1 (the other parity), and
    // keeps the last cell of each in registers: the cell above a new E cell is the lane's own O cell, the one to its left the O cell
    // of the lane below (one DPP wave shift, no LDS round trip); for an O cell it is the other way round; the diagonal predecessor is
    // the previous cell of the same offset.  Two steps per trip, so the parity is a compile-time constant.
    int ME = LC_BNEG, XE = LC_BNEG, YE = LC_BNEG, MO = LC_BNEG, XO = LC_BNEG, YO = LC_BNEG;
    int fin = LC_BNEG;
    // The two strings in LDS: every step of every lane reads one character of each, and a global load there (~1 us under load)
    // was the whole cost of a step (1200 steps for a 600 x 600 problem).
    LC_LDS uint8_t *ls = (LC_LDS uint8_t *)&S.lbytes[0];
    const bool staged = (size_t)(n + m) <= sizeof(S.lbytes);
    if (staged) { WG_FOR(x, n) { ls[x] = Sx[x]; } WG_FOR(x, m) { ls[n + x] = Tx[x]; } }
    WG_SYNC();
    // (two instances of the loop: with a global load anywhere in it, every step's wait for that load would also wait for the
    //  previous step's traceback store -- a full memory round trip per step; the LDS instance leaves the stores in flight)
    auto run = [&](auto STG) {
      constexpr bool stg = decltype(STG)::value;
      const int oE = lo + 2 * lane, oO = oE + 1;
      const bool okE = oE <= hi, okO = oO <= hi;
      auto cell = [&](const int t, const int o, const bool ok, const int dM0, const int uM0, const int uX0, const int lM0, const int lY0, int &oM, int &oX, int &oY) {
        const int i = (t - o) >> 1, j = t - i;
        const bool valid = ok && i >= 1 && i <= n && j >= 1 && j <= m;
        int dM = dM0, uM = uM0, uX = uX0, lM = lM0, lY = lY0;
        if (i == 1) { uM = -8 - j; uX = -8 - j; dM = (j == 1) ? 0 : -8 - (j - 1); }
        if (j == 1) { lM = -8 - i; lY = -8 - i; if (i != 1) dM = -8 - (i - 1); }
        const int ic = valid ? i - 1 : 0, jc = valid ? j - 1 : 0;
        int sch, tch;
        if constexpr (stg) { sch = (int)ls[ic]; tch = (int)ls[n + jc]; } else { sch = (int)Sx[ic]; tch = (int)Tx[jc]; }
        const int xa = uX - 1, xb = uM - 8;
        const int xs = xa > xb ? xa : xb, xt = xa > xb ? 1 : 0;
        const int ya = lY - 1, yb = lM - 8;
        const int ys = ya > yb ? ya : yb, yt = ya > yb ? 1 : 0;
        int ms = dM + (sch == tch ? 2 : -4), mt = 0;
        if (xs > ms) { ms = xs; mt = 1; }
        if (ys > ms) { ms = ys; mt = 2; }
        if (valid) {
          tbp[(size_t)t * 64 + (size_t)lane] = (uint8_t)(mt | (xt << 2) | (yt << 4));
          if (i == n && j == m) fin = ms;
        }
        oM = valid ? ms : LC_BNEG; oX = valid ? xs : LC_BNEG; oY = valid ? ys : LC_BNEG;
      };
      // wave_shr:1 -- lane l reads lane l - 1 (lane 0 keeps `old`); wave_shl:1 -- lane l reads lane l + 1 (lane 63 keeps `old`)
      auto stepE = [&](const int t) {      // new cell at offset E: above = own O, left = O of lane - 1
        const int lM = __builtin_amdgcn_update_dpp(LC_BNEG, MO, 0x138, 0xf, 0xf, false);
        const int lY = __builtin_amdgcn_update_dpp(LC_BNEG, YO, 0x138, 0xf, 0xf, false);
        int cM, cX, cY; cell(t, oE, okE, ME, MO, XO, lM, lY, cM, cX, cY);
        ME = cM; XE = cX; YE = cY;
      };
      auto stepO = [&](const int t) {      // new cell at offset O: above = E of lane + 1, left = own E
        const int uM = __builtin_amdgcn_update_dpp(LC_BNEG, ME, 0x130, 0xf, 0xf, false);
        const int uX = __builtin_amdgcn_update_dpp(LC_BNEG, XE, 0x130, 0xf, 0xf, false);
        int cM, cX, cY; cell(t, oO, okO, MO, uM, uX, ME, YE, cM, cX, cY);
        MO = cM; XO = cX; YO = cY;
      };
      int t = 2;
      if (((t - lo) & 1) != 0) { stepO(t); ++t; }
      for (; t + 1 <= n + m; t += 2) { stepE(t); stepO(t + 1); }
      if (t <= n + m) stepE(t);
    };
    if (lane < 64) { if (staged) run(LcBool<true>{}); else run(LcBool<false>{}); }          // (not the helper waves of the fat form)
    if (fin != LC_BNEG) S.al_score = fin;                              // (exactly one lane holds the corner)
  }
#else
  {
    int M1[64], X1[64], Y1[64], M2[64];
    for (int l = 0; l < 64; ++l) { M1[l] = X1[l] = Y1[l] = M2[l] = LC_BNEG; }
    for (int t = 2; t <= n + m; ++t) {
      const int pt = (t - lo) & 1;
      int nM_[64], nX_[64], nY_[64];
      for (int lane = 0; lane < 64; ++lane) {
        const int o = lo + 2 * lane + pt;
        const int i = (t - o) >> 1, j = t - i;
        const bool valid = o <= hi && i >= 1 && i <= n && j >= 1 && j <= m;
        const int nl = pt ? lane + 1 : lane - 1;
        int nM = (nl < 0 || nl > 63) ? LC_BNEG : M1[nl];
        int nXY = (nl < 0 || nl > 63) ? LC_BNEG : (pt ? X1[nl] : Y1[nl]);
        int uM = pt ? nM : M1[lane], uX = pt ? nXY : X1[lane];
        int lM = pt ? M1[lane] : nM, lY = pt ? Y1[lane] : nXY;
        int dM = M2[lane];
        if (i == 1) { uM = -8 - j; uX = -8 - j; dM = (j == 1) ? 0 : -8 - (j - 1); }
        if (j == 1) { lM = -8 - i; lY = -8 - i; if (i != 1) dM = -8 - (i - 1); }
        int cM = LC_BNEG, cX = LC_BNEG, cY = LC_BNEG; uint8_t tb = 0;
        if (valid) {
          band_cell(i, j, (int)Sx[i - 1], (int)Tx[j - 1], dM, uM, uX, lM, lY, &cM, &cX, &cY, &tb);
          W.tb[(size_t)t * 64 + (size_t)lane] = tb;
          if (i == n && j == m) S.al_score = cM;
        }
        nM_[lane] = cM; nX_[lane] = cX; nY_[lane] = cY;
      }
      for (int l = 0; l < 64; ++l) { M2[l] = M1[l]; M1[l] = nM_[l]; X1[l] = nX_[l]; Y1[l] = nY_[l]; }
    }
  }
#endif
  WG_SYNC();
  // ---- is the band enough?
  const int score = wg_bcast(&S.al_score);
  const int gh = hi + 1, gv_h = hi + 1 - d;                            // leave through the upper edge: >= gh columns skipped, >= gv_h rows to come back
  const int gv = 1 - lo, gh_l = 1 - lo + d;                            // through the lower edge
  const int out_hi = 2 * (m - gh) - (7 + gh) - (7 + gv_h);
  const int out_lo = 2 * (n - gv) - (7 + gv) - (7 + gh_l);
  const int bound = out_hi > out_lo ? out_hi : out_lo;
  if (score == LC_BNEG || score <= bound) { WG_LANE0 { S.al_band = 0; } return false; }
  return true;
}

// traceback into W.aln: returns alignment length; ref_aln at aln[0..L), path_aln at aln[cap..cap+L) (ASCII)
// Lane 0 follows the traceback bytes and only NOTES every column (i << 16 | j << 2 | kind: 0 both characters, 1 reference character
// against '-', 2 '-' against path character) in W.scratch; align_traceback_fill (all lanes) writes the two aligned strings from the
// notes -- the walk itself then has no loads but the traceback bytes, eight cells of a diagonal at a time.
DEVNI int align_traceback(Ctx &c, LC_GLOBAL const uint8_t *Sx, int n, LC_GLOBAL const uint8_t *Tx, int m) {
  LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int cap = (int)LC_CTX(c).C->max_w + (int)LC_CTX(c).C->path_cap + 2;
  LC_GLOBAL uint32_t *cols = W.scratch;
  (void)Sx; (void)Tx;
  int i = n, j = m, L = 0;
  bool forcex = false, forcey = false;
  unsigned long long pre = 0; int bi = -1, bj = -1, used = 8;
  while (i > 0 || j > 0) {
    if (i < 0 || j < 0 || L >= cap) { OVF(c); return 0; }          // the reference would read out of bounds here
    // the path mostly runs along a diagonal: fetch the next 8 cells of the current diagonal in one go (independent
    // loads, one wait) and fall back to a new batch whenever a gap leaves the diagonal
    if (used >= 8 || i != bi - used || j != bj - used) {
      bi = i; bj = j; used = 0;
      pre = 0;
      if (!LC_SREF(c).al_band) { for (int q = 0; q < 8; ++q) pre |= (unsigned long long)((i - q >= 0 && j - q >= 0) ? W.tb[LC_TB(i - q, j - q, n)] : (uint8_t)0) << (8 * q); }
      else {
        const int blo = LC_SREF(c).al_lo;
        for (int q = 0; q < 8; ++q) {
          const int ii = i - q, jj = j - q;
          uint8_t bb = 0;
          if (ii >= 0 && jj >= 0) {
            if (ii == 0) bb = (uint8_t)((jj == 0 ? 3 : 2) | (0 << 2) | (2 << 4));        // the borders are not stored in the band layout
            else if (jj == 0) bb = (uint8_t)(1 | (2 << 2) | (0 << 4));
            else { const int tt = ii + jj, oo = jj - ii; bb = W.tb[(size_t)tt * 64 + (size_t)((oo - blo - ((tt - blo) & 1)) >> 1)]; }
          }
          pre |= (unsigned long long)bb << (8 * q);
        }
      }
    }
    uint8_t b = (uint8_t)(pre >> (8 * used)); ++used;
    int t = b & 3, x = (b >> 2) & 3, y = (b >> 4) & 3;
    if (t == 3) break;
    else if (forcex) { if (i < 1) { OVF(c); return 0; } cols[L++] = ((uint32_t)i << 16) | ((uint32_t)j << 2) | 1u; if (x == 0) forcex = false; --i; }
    else if (t == 1) { cols[L++] = ((uint32_t)i << 16) | ((uint32_t)j << 2) | 1u; if (x == 1) forcex = true; --i; }
    else if (forcey) { if (j < 1) { OVF(c); return 0; } cols[L++] = ((uint32_t)i << 16) | ((uint32_t)j << 2) | 2u; if (y == 0) forcey = false; --j; }
    else if (t == 2) { cols[L++] = ((uint32_t)i << 16) | ((uint32_t)j << 2) | 2u; if (y == 1) forcey = true; --j; }
    else { cols[L++] = ((uint32_t)i << 16) | ((uint32_t)j << 2); --i; --j; }
  }
  return L;
}
// The same walk by the whole wave (all lanes: returns the length, uniform).  Host emulation: the one-lane walk above.
// GPU: every lane follows the walk redundantly (the state is wave-uniform); the traceback bytes are fetched 64 at a time, one per
// lane, along the line the walk is on -- the diagonal, or the column / row while a gap is being extended -- and handed around with a
// wave shuffle, so the walk pays one memory round trip per 64 cells (or per turn) instead of one per 8; the notes are kept one per
// lane and stored 64 at a time.
DEVNI int align_traceback_wg(Ctx &c, LC_GLOBAL const uint8_t *Sx, int n, LC_GLOBAL const uint8_t *Tx, int m) {
  LC_WS &S = LC_SREF(c);
#ifdef LANCET_WAVE_EMU
  WG_LANE0 { S.al_L = align_traceback(c, Sx, n, Tx, m); }
  return wg_bcast(&S.al_L);
#else
  LC_GLOBAL Work &W = *LC_CTX(c).W;
  (void)Sx; (void)Tx;
  if (threadIdx.x < 64) {
    const int lane = (int)threadIdx.x;
    const int cap = (int)LC_CTX(c).C->max_w + (int)LC_CTX(c).C->path_cap + 2;
    LC_GLOBAL uint32_t *cols = W.scratch;
    LC_GLOBAL const uint8_t *tbp = W.tb;
    const int band = S.al_band, blo = S.al_lo;
    int i = n, j = m, L = 0;
    bool forcex = false, forcey = false, bad = false;
    int ci = 0, cj = 0, cdir = -1;
    uint32_t mine = 0, note = 0;
    while (i > 0 || j > 0) {
      if (i < 0 || j < 0 || L >= cap) { bad = true; break; }              // the reference would read out of bounds here
      const int want = forcex ? 1 : (forcey ? 2 : 0);
      int q = -1;
      if (cdir == want) {
        if (want == 0) { if (ci - i == cj - j) q = ci - i; }
        else if (want == 1) { if (cj == j) q = ci - i; }
        else { if (ci == i) q = cj - j; }
      }
      if (q < 0 || q > 63) {
        cdir = want; ci = i; cj = j; q = 0;
        const int ii = want == 2 ? i : i - lane, jj = want == 1 ? j : j - lane;
        uint32_t bb = 0;
        if (ii >= 0 && jj >= 0) {
          if (!band) bb = tbp[LC_TB(ii, jj, n)];
          else if (ii == 0) bb = (uint32_t)((jj == 0 ? 3 : 2) | (0 << 2) | (2 << 4));        // the borders are not stored in the band layout
          else if (jj == 0) bb = (uint32_t)(1 | (2 << 2) | (0 << 4));
          else { const int tt = ii + jj, oo = jj - ii; const int li = (oo - blo - ((tt - blo) & 1)) >> 1; bb = (li >= 0 && li < 64) ? tbp[(size_t)tt * 64 + (size_t)li] : 0u; }
        }
        mine = bb;
      }
      const uint32_t b = (uint32_t)__shfl((int)mine, q, 64);
      const int t = (int)(b & 3u), x = (int)((b >> 2) & 3u), y = (int)((b >> 4) & 3u);
      uint32_t v;
      if (t == 3) break;
      else if (forcex) { if (i < 1) { bad = true; break; } v = ((uint32_t)i << 16) | ((uint32_t)j << 2) | 1u; if (x == 0) forcex = false; --i; }
      else if (t == 1) { v = ((uint32_t)i << 16) | ((uint32_t)j << 2) | 1u; if (x == 1) forcex = true; --i; }
      else if (forcey) { if (j < 1) { bad = true; break; } v = ((uint32_t)i << 16) | ((uint32_t)j << 2) | 2u; if (y == 0) forcey = false; --j; }
      else if (t == 2) { v = ((uint32_t)i << 16) | ((uint32_t)j << 2) | 2u; if (y == 1) forcey = true; --j; }
      else { v = ((uint32_t)i << 16) | ((uint32_t)j << 2); --i; --j; }
      if (lane == (L & 63)) note = v;
      ++L;
      if ((L & 63) == 0) cols[L - 64 + lane] = note;
    }
    if (!bad && lane < (L & 63)) cols[(L & ~63) + lane] = note;
    if (lane == 0) { if (bad) { OVF(c); L = 0; } S.al_L = L; }
  }
  return wg_bcast(&S.al_L);
#endif
}
// all lanes: the aligned strings from the noted columns (noted from the end of the alignment backwards)
DEVNI void align_traceback_fill(Ctx &c, LC_GLOBAL const uint8_t *Sx, LC_GLOBAL const uint8_t *Tx, int L) {
  LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int cap = (int)LC_CTX(c).C->max_w + (int)LC_CTX(c).C->path_cap + 2;
  LC_GLOBAL uint8_t *ra = W.aln, *pa = W.aln + cap;
  LC_GLOBAL const uint32_t *cols = W.scratch;
  WG_FOR(x, L) {
    const uint32_t v = cols[L - 1 - x];
    const int i = (int)(v >> 16), j = (int)((v >> 2) & 0x3FFFu), kind = (int)(v & 3u);
    ra[x] = kind == 2 ? (uint8_t)'-' : (uint8_t)"ACGTN"[Sx[i - 1]];
    pa[x] = kind == 1 ? (uint8_t)'-' : (uint8_t)"ACGT"[Tx[j - 1]];
  }
  WG_SYNC();
}

// ---------------------------------------------------------------------------------------------------------
// processPath (reference src/Graph.cc:788-1220) + Transcript_t::computeStats (reference src/Transcript.hh:123-226)
// ---------------------------------------------------------------------------------------------------------
struct Acc {   // one cov_t field of one of the four coverage vectors of a transcript
  uint16_t first, mn, mnz, sum, sumnz, nnz; uint32_t n;
};
template <class A> DEV void acc_init(A &a, uint16_t v) { a.first = v; a.mn = v; a.mnz = v; a.sum = 0; a.sumnz = 0; a.nnz = 0; a.n = 0; a.sum = (uint16_t)(a.sum + v); if (v != 0) { a.sumnz = (uint16_t)(a.sumnz + v); ++a.nnz; } a.n = 1; }
template <class A> DEV void acc_push(A &a, uint16_t v) {
  a.sum = (uint16_t)(a.sum + v); if (v != 0) { a.sumnz = (uint16_t)(a.sumnz + v); a.nnz = (uint16_t)(a.nnz + 1); }
  if (v < a.mn) a.mn = v;
  if (v < a.mnz && v != 0) a.mnz = v;
  ++a.n;
}
template <class A> DEV uint16_t acc_mean(const A &a) { return a.n > 0 ? (uint16_t)((float)a.sum / (float)a.n) : (uint16_t)0; }

struct TS {
  uint32_t pos, ref_pos, start_pos, end_pos, ref_end_pos;
  int col0, col1;          // alignment columns [col0, col1] -> transcript.ref / .qry
  char code, prev_bp_ref, prev_bp_alt; bool somatic;
  Acc aN[4], aT[4], rN[2], rT[2];   // alt: fwd rev minqv_fwd minqv_rev ; ref: fwd rev
  // lr_mode, hp0 hp1 hp2: ref minimum and (u16-wrapping) sum per sample ; alt minimum of hpX and of hpX_minqv
  uint16_t hrmnN[3], hrmnT[3], hrsumN[3], hrsumT[3], hamnN[3], hamnT[3], haqN[3], haqT[3];
};
struct HPc { uint16_t nh[3], nq[3], th[3], tq[3]; };   // hp0-2 and hp0-2_minqv of one position, normal / tumor
template <class T> DEV void ts_hp_init(T &t, const HPc &a, const HPc &r) {
  for (int j = 0; j < 3; ++j) {
    t.hrmnN[j] = r.nh[j]; t.hrmnT[j] = r.th[j]; t.hrsumN[j] = r.nh[j]; t.hrsumT[j] = r.th[j];
    t.hamnN[j] = a.nh[j]; t.hamnT[j] = a.th[j]; t.haqN[j] = a.nq[j]; t.haqT[j] = a.tq[j];
  }
}
template <class T> DEV void ts_hp_add_alt(T &t, const HPc &a) {
  for (int j = 0; j < 3; ++j) {
    if (a.nh[j] < t.hamnN[j]) t.hamnN[j] = a.nh[j];
    if (a.th[j] < t.hamnT[j]) t.hamnT[j] = a.th[j];
    if (a.nq[j] < t.haqN[j]) t.haqN[j] = a.nq[j];
    if (a.tq[j] < t.haqT[j]) t.haqT[j] = a.tq[j];
  }
}
template <class T> DEV void ts_hp_add_ref(T &t, const HPc &r) {
  for (int j = 0; j < 3; ++j) {
    if (r.nh[j] < t.hrmnN[j]) t.hrmnN[j] = r.nh[j];
    if (r.th[j] < t.hrmnT[j]) t.hrmnT[j] = r.th[j];
    t.hrsumN[j] = (uint16_t)(t.hrsumN[j] + r.nh[j]); t.hrsumT[j] = (uint16_t)(t.hrsumT[j] + r.th[j]);
  }
}
DEV void ref_hp_at(const Ctx &c, uint32_t pos, HPc &r) {       // hp0-2 of Ref_t::getCovStructAt (the minqv fields of the reference stay 0)
  for (int j = 0; j < 3; ++j) { r.nh[j] = r.th[j] = r.nq[j] = r.tq[j] = 0; }
  if (!LC_SREF(c).LR || (int)pos >= LC_SREF(c).reflen) return;
  const uint16_t *h = LC_CTX(c).W->refhp + 6 * pos;
  for (int j = 0; j < 3; ++j) { r.th[j] = h[j]; r.nh[j] = h[3 + j]; }
}
DEV void path_hp_at(const Ctx &c, int P, HPc &a) {
  for (int j = 0; j < 3; ++j) { a.nh[j] = a.th[j] = a.nq[j] = a.tq[j] = 0; }
  if (!LC_SREF(c).LR) return;
  uint32_t d = LC_CTX(c).W->pdesc[P];
  desc_hp(c, d, 0, a.nh, a.nq);
  desc_hp(c, d, 1, a.th, a.tq);
}
template <class T> DEV void ts_add_alt(T &t, const uint16_t *n4, const uint16_t *t4) { for (int q = 0; q < 4; ++q) { acc_push(t.aN[q], n4[q]); acc_push(t.aT[q], t4[q]); } }
template <class T> DEV void ts_add_ref(T &t, const uint16_t *n2, const uint16_t *t2) { for (int q = 0; q < 2; ++q) { acc_push(t.rN[q], n2[q]); acc_push(t.rT[q], t2[q]); } }

DEV void ref_cov_at(const Ctx &c, uint32_t pos, uint16_t *n2, uint16_t *t2) {   // Ref_t::getCovStructAt, reference src/Ref.cc:253-267
  if ((int)pos < LC_SREF(c).reflen) { const uint16_t *r = LC_CTX(c).W->refcov + 4 * pos; t2[0] = r[0]; t2[1] = r[1]; n2[0] = r[2]; n2[1] = r[3]; }
  else { n2[0] = n2[1] = t2[0] = t2[1] = 0; }
}
DEV void path_cov_at(const Ctx &c, int P, uint16_t *n4, uint16_t *t4) {         // coverageN[P], coverageT[P]
  uint32_t d = LC_CTX(c).W->pdesc[P];
  desc_cov(c, d, 0, &n4[0], &n4[1], &n4[2], &n4[3]);
  desc_cov(c, d, 1, &t4[0], &t4[1], &t4[2], &t4[3]);
}

// ---- --linked-reads: barcode sets of a variant (Graph_t::getBXsetAt / Ref_t::getBXsetAt, reference src/Graph.cc:83-114,
// src/Ref.cc:96-125).  bx_table[mer] = barcodes of all reads of the sample that contain the k-mer = the node's csr list.
DEV void bx_add_node(Ctx &c, uint32_t X, uint32_t nml, uint32_t *n) {
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_GLOBAL const DevBatch &B = *LC_CTX(c).B; LC_WS &S = LC_SREF(c);
  const uint32_t g0 = B.read_begin[S.w], refr = (uint32_t)(S.R - 1);
  LC_GLOBAL uint32_t *buf = W.bxbuf;
  for (uint32_t i = W.nocc[X]; i < W.nocc[X + 1]; ++i) {
    const uint32_t r = CS_READ(W_CSR(W)[i]);
    if (r == refr) continue;
    if (RI_NML(B.rinfo[g0 + r]) != nml) continue;
    const uint32_t bx = B.bx_rank[g0 + r];
    if (bx == 0xFFFFFFFFu) continue;
    uint32_t lo = 0, hi = *n;                                    // sorted, distinct (std::set<string> order == rank order)
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (buf[mid] < bx) lo = mid + 1; else hi = mid; }
    if (lo < *n && buf[lo] == bx) continue;
    if (*n >= LC_CTX(c).C->reads_cap) { OVF(c); return; }
    for (uint32_t j = *n; j > lo; --j) buf[j] = buf[j - 1];
    buf[lo] = bx; ++*n;
  }
}
// node of the k-mer codes[0..K) (2-bit codes) or LC_NIL: the open-addressing table of the current build
DEVNI uint32_t kmer_lookup(Ctx &c, const uint8_t *codes) {
  LC_GLOBAL Work &W = *LC_CTX(c).W; LC_WS &S = LC_SREF(c);
  const int K = S.K, NW = S.NW;
  unsigned long long fw[LC_NWMAX], rc[LC_NWMAX];
  for (int w = 0; w < LC_NWMAX; ++w) { fw[w] = 0; rc[w] = 0; }
  for (int i = 0; i < K; ++i) { key_push_fw(fw, NW, K, codes[i] & 3); key_push_rc(rc, NW, K, codes[i] & 3); }
  const unsigned long long *ck = key_less(fw, rc, NW) ? fw : rc;
  const uint32_t mask = S.tmask;
  unsigned long long h = 0; uint32_t idx;
  if (NW == 1 && K <= 31) { h = ck[0] + 1ULL; idx = (uint32_t)mix64(h) & mask; }
  else {
    for (int w = 0; w < NW; ++w) h = mix64(h ^ (ck[w] + 0x9e3779b97f4a7c15ULL * (unsigned long long)(w + 1)));
    h &= ~(1ULL << 63);
    if (h == 0) h = 1;
    idx = (uint32_t)h & mask;
  }
  for (uint32_t probes = 0; probes <= mask; ++probes) {
    const unsigned long long cur = SL_TAG(W, idx);
    if (cur == 0) return LC_NIL;
    if (cur == h) {
      bool same = true;
      if (!(NW == 1 && K <= 31)) for (int w = 0; w < NW; ++w) if (W.slot_key[(size_t)idx * LC_NWMAX + w] != ck[w]) same = false;
      if (same) return SL_NODE(W, idx);
    }
    idx = (idx + 1) & mask;
  }
  return LC_NIL;
}
DEVNI void emit_variant_lr(Ctx &c, uint32_t vi, const TS &t, const uint16_t hp12[12], int plen) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W; DevOut &O = *LC_CTX(c).OUT;
  lancet_variant_lr &l = O.variants_lr[vi];
  for (int q = 0; q < 12; ++q) l.hp[q] = hp12[q];
  l.reserved[0] = l.reserved[1] = 0;
  const int K = S.K;
  const uint32_t ro = W.occ_base[S.R - 1];
  for (int q = 0; q < 4; ++q) {          // bxset_ref_N, bxset_ref_T, bxset_alt_N, bxset_alt_T  (Graph.cc:1177-1182)
    const uint32_t nml = (q == 0 || q == 2) ? 1u : 0u;
    uint32_t n = 0;
    if (q < 2) {                           // k-mers of Ref_t::seq at [ref_pos-1, ref_end_pos-1] that are in the reference's mer table
      for (int i = (int)t.ref_pos - 1; i <= (int)t.ref_end_pos - 1; ++i) {
        if (i < 0 || i + K > S.seq_len) continue;            // shorter substr: not a key of the table
        // (a graph from the LDS build kernel: W.occ names a stand-in for a k-mer whose node did not survive, but Ref_t's barcode table
        //  holds the reads of every k-mer of its mer table -- the node and the membership bit come from lr_refnode, load_prebuilt_lr)
        const uint32_t rv = S.prebuilt ? W.lr_refnode[S.seq_t5 + i] : 0u;
        const uint32_t X = S.prebuilt ? (rv & 0x3FFFFFFFu) : (W.occ[ro + (uint32_t)(S.seq_t5 + i)] & 0x3FFFFFFFu);
        if (S.prebuilt ? (rv >> 31) != 0u : (W.gr[X].flags & NF_INMER) != 0u) bx_add_node(c, X, nml, &n);
      }
    } else {                               // k-mers of the path string at [start_pos-2, end_pos-1]
      for (int i = (int)t.start_pos - 2; i <= (int)t.end_pos - 1; ++i) {
        if (i < 0 || i + K > plen) continue;
        const uint32_t X = kmer_lookup(c, W.pseq + i);
        if (X != LC_NIL) bx_add_node(c, X, nml, &n);
      }
    }
    const uint32_t off = dev_atomic_add(O.n_bx, n);
    if (off + n > LC_CTX(c).C->bx_cap) { OVF(c); l.bx_off[q] = 0; l.bx_len[q] = 0; continue; }
    l.bx_off[q] = off; l.bx_len[q] = n;
    for (uint32_t j = 0; j < n; ++j) O.bx_blob[off + j] = W.bxbuf[j];
  }
}

DEVNI void emit_variant(Ctx &c, const TS &t, const uint16_t cov[8], int strLen, const uint8_t *motif, int motifLen, bool hasStr,
                      const uint8_t *ra, const uint8_t *pa, const uint16_t hp12[12], int plen) {
  LC_WS &S = LC_SREF(c); DevOut &O = *LC_CTX(c).OUT;
  uint32_t vi = dev_atomic_add(O.n_variants, 1u);
  int rl = t.col1 - t.col0 + 1;
  char sbuf[80]; int sl = 0;
  if (hasStr) {
    char digs[12]; int nd = 0; int v = strLen; do { digs[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (nd) sbuf[sl++] = digs[--nd];
    for (int i = 0; i < motifLen; ++i) sbuf[sl++] = "ACGT"[motif[i]];
  }
  uint32_t need = (uint32_t)(2 * rl + sl);
  uint32_t bo = dev_atomic_add(O.n_blob, need);
  if (vi >= LC_CTX(c).C->var_cap || bo + need > LC_CTX(c).C->blob_cap) { OVF(c); return; }
  lancet_variant &v = O.variants[vi];
  v.window = S.w; v.seq_in_window = S.emit_seq++; v.chr_id = LC_CTX(c).B->chr_id[S.w]; v.pos = (int32_t)t.pos - 1;
  v.code = (uint8_t)t.code; v.prev_bp_ref = (uint8_t)t.prev_bp_ref; v.prev_bp_alt = (uint8_t)t.prev_bp_alt; v.reserved = 0;
  v.kmer = (uint16_t)S.K; for (int q = 0; q < 8; ++q) v.cov[q] = cov[q]; v.reserved2 = 0;
  v.ref_off = bo; v.ref_len = (uint32_t)rl; v.alt_off = bo + (uint32_t)rl; v.alt_len = (uint32_t)rl; v.str_off = bo + 2u * (uint32_t)rl; v.str_len = (uint32_t)sl;
  for (int i = 0; i < rl; ++i) { O.blob[bo + i] = (char)ra[t.col0 + i]; O.blob[bo + rl + i] = (char)pa[t.col0 + i]; }
  for (int i = 0; i < sl; ++i) O.blob[bo + 2 * rl + i] = sbuf[i];
  if (S.LR) emit_variant_lr(c, vi, t, hp12, plen);
}

// All lanes, before the transcript walk: per alignment column the reference position before it and the path position
// after it (prefix counts of non-gap characters), the list of the columns that are not matches, and the four
// column-type counts.  The walk (lane 0) then visits only the non-match columns instead of all of them.
//   scratch[0..L) = pos_in_ref, scratch[L+1..2L+1) = pathpos - (column consumes a path base), scratch[2L+2..] = column list
DEVNI void walk_prepare(Ctx &c, int L) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int cap = (int)LC_CTX(c).C->max_w + (int)LC_CTX(c).C->path_cap + 2;
  LC_GLOBAL const uint8_t *ra = W.aln, *pa = W.aln + cap;
  LC_GLOBAL uint32_t *E1 = W.scratch, *E2 = W.scratch + (L + 1), *F = W.scratch + 2 * (L + 1), *cols = W.scratch + 3 * (L + 1);
  WG_LANE0 { S.wk[0] = S.wk[1] = S.wk[2] = S.wk[3] = 0; E1[L] = 0; E2[L] = 0; F[L] = 0; }
  WG_FOR(i, L) {
    const uint8_t r = ra[i], p = pa[i];
    E1[i] = r != '-'; E2[i] = p != '-'; F[i] = r != p;
    const int t = r == p ? 0 : (r == '-' ? 2 : (p == '-' ? 3 : 1));
    dev_atomic_add((LC_LDS uint32_t *)&S.wk[t], 1u);
  }
  WG_SYNC();
  wg_scan(E1, L + 1, S, S.part2);
  wg_scan(E2, L + 1, S, S.part2);
  wg_scan(F, L + 1, S, S.part2);
  WG_FOR(i, L) { if (F[i + 1] != F[i]) cols[F[i]] = (uint32_t)i; }
  WG_LANE0 { S.wk_n = (int)F[L]; }
}

// lane 0.  `np` = nodes in path, `plen` = path string length, aligned strings in W.aln (length L).
DEVNI void process_path_walk(Ctx &c, int np, int plen, int L, int complete) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = S.K;
  const int cap = (int)LC_CTX(c).C->max_w + (int)LC_CTX(c).C->path_cap + 2;
  LC_GLOBAL const uint8_t *ra = W.aln, *pa = W.aln + cap;
  const int refstart = LC_CTX(c).B->ref_start[S.w];
  TS *ts = (TS *)(void *)W.tb;                 // the traceback matrix is dead by now: reuse it for the transcripts
  auto ts_at = [&](int idx) -> TS & { return ts[idx]; };
  int nts = 0;
  unsigned pos_in_ref = 0, pathpos = 0;
  char code = '?', prev_code = '?';
  const int match_bp = S.wk[0], snp_bp = S.wk[1], ins_bp = S.wk[2], del_bp = S.wk[3];      // walk_prepare
  LC_GLOBAL const uint32_t *E1 = W.scratch, *E2 = W.scratch + (L + 1), *cols = W.scratch + 3 * (L + 1);
  int pc_i = 0, pc_cur = 0;
  int last_col = -2;
  for (int ci = 0; ci < S.wk_n; ++ci) {
    const int i = (int)cols[ci];
    prev_code = (last_col == i - 1) ? code : '=';       // the column before is a match unless it is the previous listed one
    last_col = i;
    if (ra[i] == '-') code = '^'; else if (pa[i] == '-') code = 'v'; else code = 'x';
    pos_in_ref = E1[i];
    pathpos = E2[i] + (pa[i] != '-' ? 1u : 0u);
    // Path_t::pathcontig(pathpos): pathpos never decreases, so the scan over the path's nodes resumes where it stopped
    uint32_t spanner = LC_NIL;
    for (; pc_i < np; ++pc_i) {
      const uint32_t nd = W.pnodes[pc_i];
      if (W.gr[nd].flags & NF_SPECIAL) continue;
      const int span = n_len(c, nd);
      if (pc_cur + span >= (int)pathpos) { spanner = nd; break; }
      pc_cur += span - K + 1;
    }
    if (spanner == LC_NIL) break;
    bool within_tumor = status_cnt_T(c, spanner);
    int P = (int)pathpos - 1;
    if (P < 0 || P >= plen) { OVF(c); return; }      // the reference reads coverageN[-1] here (undefined)
    {
      uint16_t cn4[4], ct4[4], rn2[2], rt2[2];
      path_cov_at(c, P, cn4, ct4);
      ref_cov_at(c, pos_in_ref + (uint32_t)S.trim5, rn2, rt2);
      HPc ha, hr;
      path_hp_at(c, P, ha);
      ref_hp_at(c, pos_in_ref + (uint32_t)S.trim5, hr);
      unsigned rrpos = pos_in_ref + (unsigned)refstart + (unsigned)S.trim5;
      int pr = i - 1, pq = i - 1;
      while (pr >= 0 && ra[pr] != 'A' && ra[pr] != 'C' && ra[pr] != 'G' && ra[pr] != 'T') --pr;
      while (pq >= 0 && pa[pq] != 'A' && pa[pq] != 'C' && pa[pq] != 'G' && pa[pq] != 'T') --pq;
      if (pr < 0 || pq < 0) { OVF(c); return; }        // reference: assert(pr >= 0)
      if (nts > 0 && prev_code != '=') {
        TS &t = ts_at(nts - 1);
        if (within_tumor) t.somatic = true;
        int reflen_before = t.col1 - t.col0 + 1;        // transcript.ref.length() before the append
        t.col1 = i; t.end_pos = (uint32_t)P; t.ref_end_pos = pos_in_ref;
        if (code == '^' && t.code == code && t.pos == rrpos) { ts_add_alt(t, cn4, ct4); ts_hp_add_alt(t, ha); }
        else if (code == 'v' && t.code == code && (t.pos + (unsigned)(reflen_before + 1)) == rrpos) { ts_add_ref(t, rn2, rt2); ts_hp_add_ref(t, hr); }
        else if (code == 'x' || t.code != code) { t.code = 'c'; ts_add_alt(t, cn4, ct4); ts_add_ref(t, rn2, rt2); ts_hp_add_alt(t, ha); ts_hp_add_ref(t, hr); }
      } else {
        if (nts >= LC_MAXTS) { OVF(c); return; }
        TS &t = ts_at(nts++);
        t.pos = rrpos; t.ref_pos = pos_in_ref; t.start_pos = (uint32_t)(P + 1); t.code = code; t.end_pos = (uint32_t)P; t.ref_end_pos = pos_in_ref;
        t.col0 = i; t.col1 = i; t.somatic = within_tumor; t.prev_bp_ref = (char)ra[pr]; t.prev_bp_alt = (char)pa[pq];
        for (int q = 0; q < 4; ++q) { acc_init(t.aN[q], cn4[q]); acc_init(t.aT[q], ct4[q]); }
        for (int q = 0; q < 2; ++q) { acc_init(t.rN[q], rn2[q]); acc_init(t.rT[q], rt2[q]); }
        ts_hp_init(t, ha, hr);
      }
    }
  }
  // bit 1 of the BFS entry = Path_t::hasCycle_m
  evt(c, EV_PATH, (uint32_t)complete, (uint32_t)S.tmp3, (uint32_t)match_bp, (uint32_t)snp_bp, (uint32_t)ins_bp, (uint32_t)del_bp);
  // Path_t::pathcontig over positions that mostly increase: the scan over the path's nodes resumes where the last one stopped
  // (from the start again when a position lies before the last one) instead of walking the path from its first node per position
  int ct_i = 0, ct_cur = 0, ct_last = -1;
  auto contig_at = [&](int pos) -> uint32_t {
    if (pos < ct_last) { ct_i = 0; ct_cur = 0; }
    ct_last = pos;
    for (; ct_i < np; ++ct_i) {
      const uint32_t nd = W.pnodes[ct_i];
      if (W.gr[nd].flags & NF_SPECIAL) continue;
      const int span = n_len(c, nd);
      if (ct_cur + span >= pos) return nd;
      ct_cur += span - K + 1;
    }
    return LC_NIL;
  };
  for (int ti = 0; ti < nts; ++ti) {
    TS &t = ts_at(ti);
    if (t.code != 'x') {
      for (int j = 0; j <= K; ++j) {
        unsigned idx1 = t.end_pos + (unsigned)j;
        if (idx1 < (unsigned)plen) {
          uint32_t sp = contig_at((int)idx1);
          if (sp == LC_NIL) break;
          if (status_cnt_T(c, sp)) t.somatic = true;
          uint16_t cn4[4], ct4[4];
          path_cov_at(c, (int)idx1, cn4, ct4);
          ts_add_alt(t, cn4, ct4);
          if (S.LR) { HPc ha; path_hp_at(c, (int)idx1, ha); ts_hp_add_alt(t, ha); }
        }
        unsigned idx2 = t.ref_end_pos + (unsigned)S.trim5 + (unsigned)j;
        uint16_t rn2[2], rt2[2];
        ref_cov_at(c, idx2, rn2, rt2);
        ts_add_ref(t, rn2, rt2);
        if (S.LR) { HPc hr; ref_hp_at(c, idx2, hr); ts_hp_add_ref(t, hr); }
      }
    }
    bool x = t.code == 'x';
    uint16_t RCNF = t.rN[0].mn, RCNR = t.rN[1].mn, RCTF = t.rT[0].mn, RCTR = t.rT[1].mn;
    uint16_t ACNF = x ? t.aN[2].mn : t.aN[0].mn, ACNR = x ? t.aN[3].mn : t.aN[1].mn;
    if (!x) { ACNF = t.aN[0].mnz; ACNR = t.aN[1].mnz; }     // getMinNon0Cov*: code != 'x' -> .fwd/.rev
    uint16_t ACTF = x ? t.aT[2].mn : t.aT[0].mn, ACTR = x ? t.aT[3].mn : t.aT[1].mn;
    if (t.somatic) { RCNF = acc_mean(t.rN[0]); RCNR = acc_mean(t.rN[1]); RCTF = acc_mean(t.rT[0]); RCTR = acc_mean(t.rT[1]); ACNF = 0; ACNR = 0; }
    uint16_t cov[8] = {RCNF, RCNR, RCTF, RCTR, ACNF, ACNR, ACTF, ACTR};
    // haplotype counts (Graph.cc:1091-1128): min of the ref / alt counts, means of the ref counts when somatic
    uint16_t hp12[12];                                  // HPRN HPRT HPAN HPAT as {hp1, hp2, hp0} (Graph.cc:1166-1169)
    {
      const uint32_t nref = t.rN[0].n;
      auto mean = [&](uint16_t sum) -> uint16_t { return nref > 0 ? (uint16_t)((float)sum / (float)nref) : (uint16_t)0; };
      uint16_t RN[3], RT[3], AN[3], AT[3];
      for (int j = 0; j < 3; ++j) {
        RN[j] = t.hrmnN[j]; RT[j] = t.hrmnT[j];
        AN[j] = x ? t.haqN[j] : t.hamnN[j]; AT[j] = x ? t.haqT[j] : t.hamnT[j];
        if (t.somatic) { RT[j] = mean(t.hrsumT[j]); RN[j] = mean(t.hrsumN[j]); AN[j] = 0; }
        if (!S.LR) { RN[j] = RT[j] = AN[j] = AT[j] = 0; }
      }
      const uint16_t v[12] = {RN[1], RN[2], RN[0], RT[1], RT[2], RT[0], AN[1], AN[2], AN[0], AT[1], AT[2], AT[0]};
      for (int q = 0; q < 12; ++q) hp12[q] = v[q];
    }
    if (LC_CTX(c).C->evt_cap) {
      evt(c, EV_TS, t.pos, (uint32_t)(t.col1 - t.col0 + 1), ((uint32_t)RCNF << 16) | RCNR, ((uint32_t)RCTF << 16) | RCTR,
          ((uint32_t)ACNF << 16) | ACNR, ((uint32_t)ACTF << 16) | ACTR, ((uint32_t)(uint8_t)t.prev_bp_ref << 8) | (uint8_t)t.prev_bp_alt);
      evt_bytes(c, ra + t.col0, (uint32_t)(t.col1 - t.col0 + 1));
      evt_bytes(c, pa + t.col0, (uint32_t)(t.col1 - t.col0 + 1));
      evt_bytes(c, (const uint8_t *)hp12, 24);
    }
    if (ACNF > 0 || ACNR > 0 || ACTF > 0 || ACTR > 0) {
      int LEN = 0, ml = 0; uint8_t motif[64];
      bool ans = find_tandems_local(c, W.pseq, plen, (int)t.start_pos, &LEN, motif, &ml);
      emit_variant(c, t, cov, LEN, motif, ml, ans, ra, pa, hp12, plen);
    }
  }
  evt(c, EV_PATH_END);
  for (int i = 0; i < np; ++i) ++W.gr[W.pnodes[i]].onref;
  // counters of eka (perfect / withsnps / withindel / withmix) ride in tmp0..tmp2 + part[0]
  if ((snp_bp + ins_bp + del_bp) == 0) ++S.tmp0; else if (snp_bp == 0) ++S.tmp1; else if ((ins_bp + del_bp) == 0) ++S.tmp2; else ++S.part[0];
}

// The same walk with the looking-up done by all lanes (short reads; --linked-reads keeps the one-lane form above).
// What a column contributes -- the path's and the reference's coverage at its position, the node that spans it -- does not
// depend on the columns before it, and in HBM it is a chain of ~15 dependent loads per column (and per position of the K+1
// positions every indel is extended by), which made this the longest phase of the windows that align.  So: the lanes gather
// 64 columns at a time into LDS records (the `acc` area of the build phases, idle here), lane 0 then runs the reference's
// sequential rules (which transcript a column opens / extends / turns complex) over LDS.  Path_t::pathcontig becomes a
// binary search over the path's node ends (they grow along the path).
//   record (10 words): alt N fwd|rev, N qf|qr, alt T fwd|rev, T qf|qr, ref N fwd|rev, ref T fwd|rev, column / position j, P, pos_in_ref,
//                      flags (1 spanner within tumor, 2 no spanner, 4 P outside the path) | column code << 8
// (lr: words 10..18 of the record take the haplotype counts of the position too -- path hp0-2 and hp0-2_minqv, normal then tumor; reference hp0-2)
DEV void walk_gather(Ctx &c, volatile LC_LDS uint32_t *rec, int pathpos, int P, uint32_t refpos, int plen, int np, LC_GLOBAL const uint32_t *pend, uint32_t w6, uint32_t w8, uint32_t code, bool lr = false) {
  LC_GLOBAL Work &W = *LC_CTX(c).W;
  uint32_t fl = 0;
  // first node whose end reaches pathpos (special nodes carry the running position: never the answer unless nothing else is)
  int lo = 0, hi = np;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)(pend[2 * mid] & 0x7FFFFFFFu) >= pathpos) hi = mid; else lo = mid + 1; }
  while (lo < np && (pend[2 * lo] & 0x80000000u)) ++lo;
  if (lo >= np) fl |= 2u; else if (pend[2 * lo + 1] & 1u) fl |= 1u;
  uint16_t cn4[4] = {0, 0, 0, 0}, ct4[4] = {0, 0, 0, 0}, rn2[2], rt2[2];
  if (P < 0 || P >= plen) fl |= 4u; else path_cov_at(c, P, cn4, ct4);
  ref_cov_at(c, refpos, rn2, rt2);
  rec[0] = (uint32_t)cn4[0] | ((uint32_t)cn4[1] << 16); rec[1] = (uint32_t)cn4[2] | ((uint32_t)cn4[3] << 16);
  rec[2] = (uint32_t)ct4[0] | ((uint32_t)ct4[1] << 16); rec[3] = (uint32_t)ct4[2] | ((uint32_t)ct4[3] << 16);
  rec[4] = (uint32_t)rn2[0] | ((uint32_t)rn2[1] << 16); rec[5] = (uint32_t)rt2[0] | ((uint32_t)rt2[1] << 16);
  rec[6] = w6; rec[7] = (uint32_t)P; rec[8] = w8; rec[9] = fl | (code << 8);
  if (lr) {
    HPc ha, hr;
    for (int j = 0; j < 3; ++j) { ha.nh[j] = ha.th[j] = ha.nq[j] = ha.tq[j] = 0; }
    if (!(P < 0 || P >= plen)) path_hp_at(c, P, ha);
    ref_hp_at(c, refpos, hr);
    rec[10] = (uint32_t)ha.nh[0] | ((uint32_t)ha.nh[1] << 16); rec[11] = (uint32_t)ha.nh[2] | ((uint32_t)ha.nq[0] << 16); rec[12] = (uint32_t)ha.nq[1] | ((uint32_t)ha.nq[2] << 16);
    rec[13] = (uint32_t)ha.th[0] | ((uint32_t)ha.th[1] << 16); rec[14] = (uint32_t)ha.th[2] | ((uint32_t)ha.tq[0] << 16); rec[15] = (uint32_t)ha.tq[1] | ((uint32_t)ha.tq[2] << 16);
    rec[16] = (uint32_t)hr.nh[0] | ((uint32_t)hr.nh[1] << 16); rec[17] = (uint32_t)hr.nh[2] | ((uint32_t)hr.th[0] << 16); rec[18] = (uint32_t)hr.th[1] | ((uint32_t)hr.th[2] << 16);
  }
  (void)W;
}
DEV void walk_unpack_hp(volatile LC_LDS uint32_t *rc, HPc &ha, HPc &hr) {
  ha.nh[0] = (uint16_t)rc[10]; ha.nh[1] = (uint16_t)(rc[10] >> 16); ha.nh[2] = (uint16_t)rc[11]; ha.nq[0] = (uint16_t)(rc[11] >> 16); ha.nq[1] = (uint16_t)rc[12]; ha.nq[2] = (uint16_t)(rc[12] >> 16);
  ha.th[0] = (uint16_t)rc[13]; ha.th[1] = (uint16_t)(rc[13] >> 16); ha.th[2] = (uint16_t)rc[14]; ha.tq[0] = (uint16_t)(rc[14] >> 16); ha.tq[1] = (uint16_t)rc[15]; ha.tq[2] = (uint16_t)(rc[15] >> 16);
  hr.nh[0] = (uint16_t)rc[16]; hr.nh[1] = (uint16_t)(rc[16] >> 16); hr.nh[2] = (uint16_t)rc[17]; hr.th[0] = (uint16_t)(rc[17] >> 16); hr.th[1] = (uint16_t)rc[18]; hr.th[2] = (uint16_t)(rc[18] >> 16);
  for (int j = 0; j < 3; ++j) { hr.nq[j] = hr.tq[j] = 0; }
}
DEVNI void process_path_walk_wg(Ctx &c, int np, int plen, int L, int complete) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  const int K = wg_uniform(S.K);
  const int cap = (int)LC_CTX(c).C->max_w + (int)LC_CTX(c).C->path_cap + 2;
  LC_GLOBAL const uint8_t *ra = W.aln, *pa = W.aln + cap;
  const int refstart = LC_CTX(c).B->ref_start[S.w];
  const int trim5 = wg_uniform(S.trim5);
  // the transcripts of the path in LDS (the staging area of the build phases); a path with more of them than fit there takes the one-lane form
#ifndef LANCET_WAVE_EMU
  LC_LDS TS *ts = (LC_LDS TS *)&lc_shared.lbytes[0];
#else
  TS *ts = (TS *)(void *)&S.lbytes[0];
#endif
  constexpr int LC_TS_LDS = (int)(sizeof(S.lbytes) / sizeof(TS));
  volatile LC_LDS uint32_t *recs = (volatile LC_LDS uint32_t *)&S.acc[0][0];
  static_assert(sizeof(S.acc) >= 64 * 10 * sizeof(uint32_t) && sizeof(S.acc) >= 32 * 19 * sizeof(uint32_t), "64 column records of 10 words, or 32 of 19 (--linked-reads: + the haplotype counts)");
  const bool LR = wg_uniform(S.LR) != 0;
  const int RS = LR ? 19 : 10, CW = LR ? 32 : 64;                  // words per record ; columns per round
  LC_GLOBAL const uint32_t *E1 = W.scratch, *E2 = W.scratch + (L + 1), *cols = W.scratch + 3 * (L + 1);
  // ---- the path's nodes: end position (Path_t::pathcontig's cur + span) | special << 31, isStatusCnt('T')  (W.cmp is idle after the first compress)
  LC_GLOBAL uint32_t *pend = (LC_GLOBAL uint32_t *)W.cmp;
  LC_GLOBAL uint32_t *pstep = W.pdesc + plen;                     // [np + 1] span - K + 1 per node (behind the path's descriptors) -> running position
  const bool fits = (size_t)plen + (size_t)np + 2 <= (size_t)LC_CTX(c).C->path_cap && (size_t)np * 8 <= (size_t)(LC_CTX(c).C->node_cap + LC_CTX(c).C->special_cap) * sizeof(CmpRec);
  if (!fits) { WG_LANE0 { process_path_walk(c, np, plen, L, complete); } return; }
  SUBPHASE(c, 4, 3);
  WG_FOR(i, np) {
    const uint32_t nd = W.pnodes[i];
    pstep[i] = (W.gr[nd].flags & NF_SPECIAL) ? 0u : (uint32_t)(n_len(c, nd) - K + 1);
  }
  WG_LANE0 { pstep[np] = 0; }
  wg_scan(pstep, np + 1, S, S.part2);
  WG_FOR(i, np) {
    const uint32_t nd = W.pnodes[i];
    const bool sp = (W.gr[nd].flags & NF_SPECIAL) != 0;
    // (a special node -- source first, sink last -- carries the end of the real node before it, so that the ends never decrease
    //  along the path and the binary search of walk_gather is sound: running position + K - 1 is exactly that end)
    pend[2 * i] = sp ? ((pstep[i] + (uint32_t)(K - 1)) | 0x80000000u) : (pstep[i] + (uint32_t)n_len(c, nd));
    pend[2 * i + 1] = (!sp && status_cnt_T(c, nd)) ? 1u : 0u;
  }
  WG_LANE0 { S.wk_stop = 0; S.wk_nts = 0; S.wk_last = -2; S.wk_code = '?'; }
  WG_SYNC();
  const int ncols = wg_bcast(&S.wk_n);
  // ---- the non-match columns
  SUBPHASE(c, 4, 4);
  for (int c0 = 0; c0 < ncols; c0 += CW) {
    WG_FOR(l, LANCET_WG) {
      const int ci = c0 + l;
      if (l < CW && ci < ncols) {
        const int i = (int)cols[ci];
        const uint8_t r = ra[i], p = pa[i];
        const uint32_t code = r == '-' ? (uint32_t)'^' : (p == '-' ? (uint32_t)'v' : (uint32_t)'x');
        const uint32_t pos_in_ref = E1[i];
        const int pathpos = (int)(E2[i] + (p != '-' ? 1u : 0u));
        walk_gather(c, recs + RS * l, pathpos, pathpos - 1, pos_in_ref + (uint32_t)trim5, plen, np, pend, (uint32_t)i, pos_in_ref, code, LR);
      }
    }
    WG_SYNC();
    WG_LANE0 {
      const int cnt = ncols - c0 < CW ? ncols - c0 : CW;
      int nts = S.wk_nts, last_col = S.wk_last; char code = (char)S.wk_code, prev_code;
      for (int l = 0; l < cnt && !S.wk_stop; ++l) {
        volatile LC_LDS uint32_t *rc = recs + RS * l;
        const int i = (int)rc[6], P = (int)rc[7]; const uint32_t pos_in_ref = rc[8], fl = rc[9] & 0xFFu;
        prev_code = (last_col == i - 1) ? code : '=';
        last_col = i; code = (char)(rc[9] >> 8);
        if (fl & 2u) { S.wk_stop = 1; break; }                           // no node spans the position: the reference's loop ends here
        if (fl & 4u) { OVF(c); S.wk_stop = 2; break; }                   // the reference reads coverageN[-1] here (undefined)
        const bool within_tumor = (fl & 1u) != 0;
        uint16_t cn4[4] = {(uint16_t)rc[0], (uint16_t)(rc[0] >> 16), (uint16_t)rc[1], (uint16_t)(rc[1] >> 16)};
        uint16_t ct4[4] = {(uint16_t)rc[2], (uint16_t)(rc[2] >> 16), (uint16_t)rc[3], (uint16_t)(rc[3] >> 16)};
        uint16_t rn2[2] = {(uint16_t)rc[4], (uint16_t)(rc[4] >> 16)}, rt2[2] = {(uint16_t)rc[5], (uint16_t)(rc[5] >> 16)};
        const unsigned rrpos = pos_in_ref + (unsigned)refstart + (unsigned)trim5;
        HPc ha, hr;
        if (LR) walk_unpack_hp(rc, ha, hr); else { for (int j = 0; j < 3; ++j) { ha.nh[j] = ha.th[j] = ha.nq[j] = ha.tq[j] = 0; hr.nh[j] = hr.th[j] = hr.nq[j] = hr.tq[j] = 0; } }
        if (nts > 0 && prev_code != '=') {
          auto &t = ts[nts - 1];
          if (within_tumor) t.somatic = true;
          const int reflen_before = t.col1 - t.col0 + 1;
          t.col1 = i; t.end_pos = (uint32_t)P; t.ref_end_pos = pos_in_ref;
          if (code == '^' && t.code == code && t.pos == rrpos) { ts_add_alt(t, cn4, ct4); if (LR) ts_hp_add_alt(t, ha); }
          else if (code == 'v' && t.code == code && (t.pos + (unsigned)(reflen_before + 1)) == rrpos) { ts_add_ref(t, rn2, rt2); if (LR) ts_hp_add_ref(t, hr); }
          else if (code == 'x' || t.code != code) { t.code = 'c'; ts_add_alt(t, cn4, ct4); ts_add_ref(t, rn2, rt2); if (LR) { ts_hp_add_alt(t, ha); ts_hp_add_ref(t, hr); } }
        } else {
          int pr = i - 1, pq = i - 1;
          while (pr >= 0 && ra[pr] != 'A' && ra[pr] != 'C' && ra[pr] != 'G' && ra[pr] != 'T') --pr;
          while (pq >= 0 && pa[pq] != 'A' && pa[pq] != 'C' && pa[pq] != 'G' && pa[pq] != 'T') --pq;
          if (pr < 0 || pq < 0) { OVF(c); S.wk_stop = 2; break; }        // reference: assert(pr >= 0)
          if (nts >= LC_MAXTS) { OVF(c); S.wk_stop = 2; break; }
          if (nts >= LC_TS_LDS) { S.wk_stop = 3; break; }                 // more transcripts than LDS holds: the one-lane form, from the start
          auto &t = ts[nts++];
          t.pos = rrpos; t.ref_pos = pos_in_ref; t.start_pos = (uint32_t)(P + 1); t.code = code; t.end_pos = (uint32_t)P; t.ref_end_pos = pos_in_ref;
          t.col0 = i; t.col1 = i; t.somatic = within_tumor; t.prev_bp_ref = (char)ra[pr]; t.prev_bp_alt = (char)pa[pq];
          for (int q = 0; q < 4; ++q) { acc_init(t.aN[q], cn4[q]); acc_init(t.aT[q], ct4[q]); }
          for (int q = 0; q < 2; ++q) { acc_init(t.rN[q], rn2[q]); acc_init(t.rT[q], rt2[q]); }
          ts_hp_init(t, ha, hr);                                          // (zeros without --linked-reads)
        }
      }
      S.wk_nts = nts; S.wk_last = last_col; S.wk_code = (int)code;
    }
    if (wg_bcast(&S.wk_stop)) break;
  }
  if (wg_bcast(&S.wk_stop) == 2) return;                                  // work-space limit / undefined read: the window is given up
  if (wg_bcast(&S.wk_stop) == 3) { WG_LANE0 { process_path_walk(c, np, plen, L, complete); } return; }
  WG_LANE0 { evt(c, EV_PATH, (uint32_t)complete, (uint32_t)S.tmp3, (uint32_t)S.wk[0], (uint32_t)S.wk[1], (uint32_t)S.wk[2], (uint32_t)S.wk[3]); }
  // ---- the transcripts: K + 1 positions past the end of every indel / complex one, then the record
  const int nts = wg_bcast(&S.wk_nts);
  SUBPHASE(c, 4, 5);
  for (int ti = 0; ti < nts; ++ti) {
    WG_LANE0 { const auto &t = ts[ti]; S.wk_code = (int)t.code; S.wk_tend = (int)t.end_pos; S.wk_tref = (int)t.ref_end_pos; S.wk_stop = 0; }
    if ((char)wg_bcast(&S.wk_code) != 'x') {
      const int tend = wg_bcast(&S.wk_tend), tref = wg_bcast(&S.wk_tref);
      for (int j0 = 0; j0 <= K; j0 += CW) {
        WG_FOR(l, LANCET_WG) {
          const int j = j0 + l;
          if (l < CW && j <= K) {
            const unsigned idx1 = (unsigned)tend + (unsigned)j;
            // (idx1 >= plen: the path contributes nothing at this position, the reference still does)
            // contig_at(idx1) and coverage[idx1]: the node search is made for idx1 itself here (not idx1 + 1 as in the column walk)
            walk_gather(c, recs + RS * l, (int)idx1, idx1 < (unsigned)plen ? (int)idx1 : -1, (uint32_t)tref + (uint32_t)trim5 + (uint32_t)j, plen, np, pend, (uint32_t)j, idx1 < (unsigned)plen ? 1u : 0u, 0u, LR);
          }
        }
        WG_SYNC();
        WG_LANE0 {
          auto &t = ts[ti];
          const int cnt = K + 1 - j0 < CW ? K + 1 - j0 : CW;
          for (int l = 0; l < cnt; ++l) {
            volatile LC_LDS uint32_t *rc = recs + RS * l;
            HPc ha, hr;
            if (LR) walk_unpack_hp(rc, ha, hr);
            if (rc[8]) {                                                   // idx1 < plen
              if (rc[9] & 2u) { S.wk_stop = 1; break; }                    // contig_at == NIL: the reference leaves the loop over j
              if (rc[9] & 1u) t.somatic = true;
              uint16_t cn4[4] = {(uint16_t)rc[0], (uint16_t)(rc[0] >> 16), (uint16_t)rc[1], (uint16_t)(rc[1] >> 16)};
              uint16_t ct4[4] = {(uint16_t)rc[2], (uint16_t)(rc[2] >> 16), (uint16_t)rc[3], (uint16_t)(rc[3] >> 16)};
              ts_add_alt(t, cn4, ct4);
              if (LR) ts_hp_add_alt(t, ha);
            }
            uint16_t rn2[2] = {(uint16_t)rc[4], (uint16_t)(rc[4] >> 16)}, rt2[2] = {(uint16_t)rc[5], (uint16_t)(rc[5] >> 16)};
            ts_add_ref(t, rn2, rt2);
            if (LR) ts_hp_add_ref(t, hr);
          }
        }
        if (wg_bcast(&S.wk_stop)) break;
      }
    }
    SUBPHASE(c, 4, 6);
    // (round 6) the path string around the transcript in LDS for findTandems: lane 0 compares every character there a dozen times, one load
    // from HBM each until now -- 2.4 of the walk's 5.8 slot-seconds per 32768 windows.  The column records (S.acc) are read out by now.
    volatile LC_LDS uint8_t *stg = (volatile LC_LDS uint8_t *)&S.acc[0][0];
    int stg_lo = (int)ts[ti].start_pos - 1024; if (stg_lo < 0) stg_lo = 0;
    stg_lo = wg_uniform(stg_lo);
    const int stg_hi = stg_lo + 2048 < plen ? stg_lo + 2048 : plen;
    static_assert(sizeof(S.acc) >= 2048, "path string staged for findTandems");
    WG_FOR(i, stg_hi - stg_lo) { stg[i] = W.pseq[stg_lo + i]; }
    WG_SYNC();
    WG_LANE0 {
      auto &t = ts[ti];
      const bool x = t.code == 'x';
      uint16_t RCNF = t.rN[0].mn, RCNR = t.rN[1].mn, RCTF = t.rT[0].mn, RCTR = t.rT[1].mn;
      uint16_t ACNF = x ? t.aN[2].mn : t.aN[0].mn, ACNR = x ? t.aN[3].mn : t.aN[1].mn;
      if (!x) { ACNF = t.aN[0].mnz; ACNR = t.aN[1].mnz; }     // getMinNon0Cov*: code != 'x' -> .fwd/.rev
      uint16_t ACTF = x ? t.aT[2].mn : t.aT[0].mn, ACTR = x ? t.aT[3].mn : t.aT[1].mn;
      if (t.somatic) { RCNF = acc_mean(t.rN[0]); RCNR = acc_mean(t.rN[1]); RCTF = acc_mean(t.rT[0]); RCTR = acc_mean(t.rT[1]); ACNF = 0; ACNR = 0; }
      uint16_t cov[8] = {RCNF, RCNR, RCTF, RCTR, ACNF, ACNR, ACTF, ACTR};
      uint16_t hp12[12]; for (int q = 0; q < 12; ++q) hp12[q] = 0;       // HPRN HPRT HPAN HPAT as {hp1, hp2, hp0} (Graph.cc:1091-1128, 1166-1169; process_path_walk has the same lines)
      if (LR) {
        const uint32_t nref = t.rN[0].n;
        auto mean = [&](uint16_t sum) -> uint16_t { return nref > 0 ? (uint16_t)((float)sum / (float)nref) : (uint16_t)0; };
        uint16_t RN[3], RT[3], AN[3], AT[3];
        for (int j = 0; j < 3; ++j) {
          RN[j] = t.hrmnN[j]; RT[j] = t.hrmnT[j];
          AN[j] = x ? t.haqN[j] : t.hamnN[j]; AT[j] = x ? t.haqT[j] : t.hamnT[j];
          if (t.somatic) { RT[j] = mean(t.hrsumT[j]); RN[j] = mean(t.hrsumN[j]); AN[j] = 0; }
        }
        const uint16_t v[12] = {RN[1], RN[2], RN[0], RT[1], RT[2], RT[0], AN[1], AN[2], AN[0], AT[1], AT[2], AT[0]};
        for (int q = 0; q < 12; ++q) hp12[q] = v[q];
      }
      if (LC_CTX(c).C->evt_cap) {
        evt(c, EV_TS, t.pos, (uint32_t)(t.col1 - t.col0 + 1), ((uint32_t)RCNF << 16) | RCNR, ((uint32_t)RCTF << 16) | RCTR,
            ((uint32_t)ACNF << 16) | ACNR, ((uint32_t)ACTF << 16) | ACTR, ((uint32_t)(uint8_t)t.prev_bp_ref << 8) | (uint8_t)t.prev_bp_alt);
        evt_bytes(c, ra + t.col0, (uint32_t)(t.col1 - t.col0 + 1));
        evt_bytes(c, pa + t.col0, (uint32_t)(t.col1 - t.col0 + 1));
        evt_bytes(c, (const uint8_t *)hp12, 24);
      }
      if (ACNF > 0 || ACNR > 0 || ACTF > 0 || ACTR > 0) {
        int LEN = 0, ml = 0; uint8_t motif[64];
        bool ans = find_tandems_local(c, W.pseq, plen, (int)t.start_pos, &LEN, motif, &ml, stg, stg_lo, stg_hi);
        TS tc; tc.pos = t.pos; tc.ref_pos = t.ref_pos; tc.start_pos = t.start_pos; tc.end_pos = t.end_pos; tc.ref_end_pos = t.ref_end_pos; tc.col0 = t.col0; tc.col1 = t.col1;
        tc.code = t.code; tc.prev_bp_ref = t.prev_bp_ref; tc.prev_bp_alt = t.prev_bp_alt; tc.somatic = t.somatic;      // (what emit_variant reads; short reads: no haplotype fields)
        emit_variant(c, tc, cov, LEN, motif, ml, ans, ra, pa, hp12, plen);
      }
    }
    SUBPHASE(c, 4, 5);
  }
  SUBPHASE(c, 4, 7);
  WG_LANE0 { evt(c, EV_PATH_END); }
  WG_FOR(i, np) { dev_atomic_add(&W.gr[W.pnodes[i]].onref, 1u); }
  // counters of eka (perfect / withsnps / withindel / withmix) ride in tmp0..tmp2 + part[0]
  WG_LANE0 { const int snp_bp = S.wk[1], ins_bp = S.wk[2], del_bp = S.wk[3]; if ((snp_bp + ins_bp + del_bp) == 0) ++S.tmp0; else if (snp_bp == 0) ++S.tmp1; else if ((ins_bp + del_bp) == 0) ++S.tmp2; else ++S.part[0]; }
}

// ---------------------------------------------------------------------------------------------------------
// per-component driver pieces that need the whole workgroup (alignment, repeat scan of a path)
// ---------------------------------------------------------------------------------------------------------
// returns true when a near-perfect repeat is found in a source->sink path (Graph_t::findRepeatsInGraphPaths,
// reference src/Graph.cc:686-730)
// findRepeatsInGraphPaths (reference src/Graph.cc:1218-1296) and eka (:1430-1501) enumerate the same paths: both start from
// cleared edge flags on the same graph, take the best path of a breadth-first search over whole partial paths, flag its edges
// and search again.  The search is the longest one-lane stretch left in the window kernel, so the first enumeration keeps what
// the second needs of every path -- nodes, edges, string, per-base descriptors, the has-cycle bit, whether DFS_LIMIT cut the
// search short -- in W.mv (idle in the graph phases) and eka replays them instead of searching again.  Only a first search
// that ran to its natural end (no near-repeat found, nothing overflowed, everything fitted) is replayed.
DEVNI void pcache_store(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  LC_GLOBAL uint32_t *pc = W.mv;
  const uint32_t capw = 4u * LC_CTX(c).C->occ_cap;
  WG_LANE0 {
    const uint32_t n = (uint32_t)S.tmp2, pl = (uint32_t)S.tmp3, need = 4u + 2u * n + pl + (pl + 3u) / 4u;
    if (S.pc_top + need > capw) S.pc_ok = 0;
    else { pc[S.pc_top] = n; pc[S.pc_top + 1] = pl; pc[S.pc_top + 2] = (uint32_t)S.pc_bits; pc[S.pc_top + 3] = (uint32_t)S.pc_dfs; S.pc_base = S.pc_top; S.pc_top += need; ++S.pc_n; }
  }
  if (!wg_bcast(&S.pc_ok)) return;
  const uint32_t base = wg_bcastu(&S.pc_base), n = (uint32_t)wg_bcast(&S.tmp2), pl = (uint32_t)wg_bcast(&S.tmp3);
  WG_FOR(i, n) { pc[base + 4u + (uint32_t)i] = W.pnodes[i]; pc[base + 4u + n + (uint32_t)i] = W.pedges[i]; }
  WG_FOR(i, pl) { pc[base + 4u + 2u * n + (uint32_t)i] = W.pdesc[i]; }
  WG_FOR(i, (pl + 3u) / 4u) { pc[base + 4u + 2u * n + pl + (uint32_t)i] = ((LC_GLOBAL const uint32_t *)W.pseq)[i]; }
  WG_SYNC();
}
// the next path of the replay into W.pnodes / W.pedges / W.pdesc / W.pseq; returns its string length
DEVNI int pcache_load(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  LC_GLOBAL const uint32_t *pc = W.mv;
  const uint32_t base = wg_bcastu(&S.pc_base);
  const uint32_t n = pc[base], pl = pc[base + 1];
  WG_FOR(i, n) { W.pnodes[i] = pc[base + 4u + (uint32_t)i]; W.pedges[i] = pc[base + 4u + n + (uint32_t)i]; }
  WG_FOR(i, pl) { W.pdesc[i] = pc[base + 4u + 2u * n + (uint32_t)i]; }
  WG_FOR(i, (pl + 3u) / 4u) { ((LC_GLOBAL uint32_t *)W.pseq)[i] = pc[base + 4u + 2u * n + pl + (uint32_t)i]; }
  WG_SYNC();
  return (int)pl;
}
DEVNI bool repeats_in_graph_paths(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  WG_LANE0 {
    evt(c, EV_LOOKREP);
    S.tmp0 = 0;                                  // 0 continue, 1 stop:false, 2 stop:true
    if (S.source == LC_NIL || S.sink == LC_NIL) { evt(c, EV_MISSING); S.tmp0 = 1; }
    else evt(c, EV_SEARCH, (uint32_t)W.gr[S.source].comp);
    S.tmp1 = 0;                                  // number of flagged-edge records kept in scratch
    S.pc_ok = 1; S.pc_n = 0; S.pc_top = 0; S.pc_end_dfs = 0;
  }
  while (wg_bcast(&S.tmp0) == 0) {
    SUBPHASE(c, 3, 2);
    const bool gcached = graph_cache_wg(c);                           // (again for every path: the path before flagged its edges, and the scan below uses the same LDS)
    WG_LANE0 {
      uint32_t best = gcached ? bfs_cached(c) : bfs(c);
      if (best == LC_NIL || S.overflow) { S.tmp0 = 1; S.pc_end_dfs = S.bfs_dfs; }
      else { S.tmp2 = path_unpack(c, best); S.pc_bits = (W.queue[best].bits & 2) ? 1 : 0; S.pc_dfs = S.bfs_dfs; }
    }
    if (wg_bcast(&S.tmp0) != 0) break;
    SUBPHASE(c, 3, 3);
    { const int pl = path_string_wg(c, wg_bcast(&S.tmp2)); WG_LANE0 { S.tmp3 = pl; if (S.overflow) S.tmp0 = 1; } }
    SUBPHASE(c, 3, 4);
    if (wg_bcast(&S.tmp0) != 0) break;
    {
      // isAlmostRepeat(path->str(), K, MAX_MISMATCH): only M >= K + 1 is asked.  A path that spells the reference between the anchors
      // needs no scan: its longest near-repeat cannot exceed that of the whole window reference, which passed this very test for
      // this k before the graph was built (Microassembler.cc:124-131; the test is only made when reflen > k).
      // Round 5: a path that DIFFERS from the reference between the anchors needs only the windows that overlap what differs.  With the
      // common prefix and suffix taken off, the path is ref[0, a) + something + ref[b', end): a window of k + 1 positions that lies inside
      // the prefix or the suffix, compared with another such window, is a pair of windows of the window reference at two DIFFERENT places
      // (the one in the suffix maps to a position >= b' >= a, past the one in the prefix) -- which the reference passed; so a new
      // near-repeat has a window of its pair overlapping [a, b).  The scan walks, per shift, only the mask words around it.
      const int pl = wg_bcast(&S.tmp3);
      LC_GLOBAL const uint8_t *rs = LC_CTX(c).B->ref_codes + LC_CTX(c).B->ref_off[S.w] + S.seq_t5;
      const int rl = wg_uniform(S.seq_len), ml = pl < rl ? pl : rl;
      const bool tested = S.reflen - S.K > 0;                       // (the window reference went through this test for this k)
      WG_LANE0 { S.ps_hd = 0x7FFFFFFF; S.ps_tl = 0x7FFFFFFF; }
      WG_SYNC();
      if (tested) {
        WG_FOR(i, ml) {
          if (rs[i] != W.pseq[i]) dev_atomic_min((LC_LDS uint32_t *)&S.ps_hd, (uint32_t)i);
          if (rs[rl - 1 - i] != W.pseq[pl - 1 - i]) dev_atomic_min((LC_LDS uint32_t *)&S.ps_tl, (uint32_t)i);
        }
      }
      const int hd = wg_bcast(&S.ps_hd), tl = wg_bcast(&S.ps_tl);
      if (tested && pl == rl && hd == 0x7FFFFFFF) { WG_LANE0 { S.repE = 0; S.repM = 0; } }          // the reference itself
      else if (!tested) repeat_scan_min(S.rs, W.pseq, pl, LC_CTX(c).P->max_mismatch, 0x7FFF, wg_uniform(S.K) + 1, &S.repE, &S.repM, nullptr, &S.rs_bad);
      else {
        const int a = hd < ml ? hd : ml;                              // common prefix
        int sf = tl < ml ? tl : ml; if (a + sf > ml) sf = ml - a;     // common suffix, not overlapping the prefix in either string
        const int b = pl - sf, K1 = wg_uniform(S.K) + 1;
        repeat_scan_min(S.rs, W.pseq, pl, LC_CTX(c).P->max_mismatch, 0x7FFF, K1, &S.repE, &S.repM, nullptr, &S.rs_bad, a - K1, b + K1);
      }
    }
    SUBPHASE(c, 3, 5);
    WG_LANE0 {
      // NB isAlmostRepeat only looks at windows that end before the last base: handled inside repeat_scan
      if (S.tmp3 - S.K > 0 && S.repM >= S.K + 1) { evt(c, EV_NEAR_QRY, S.K); S.tmp0 = 2; }
      else {
        path_flag_edges(c, S.tmp2, 1u);
        for (int j = 1; j < S.tmp2; ++j) { if ((uint32_t)S.tmp1 < LC_CTX(c).C->node_cap) W.scratch[S.tmp1++] = W.pedges[j]; else { OVF(c); S.tmp0 = 1; } }
      }
    }
    SUBPHASE(c, 3, 6);
    if (wg_bcast(&S.tmp0) == 0 && wg_bcast(&S.pc_ok)) pcache_store(c);
  }
  SUBPHASE(c, 3, 10);
  WG_LANE0 {
    for (int j = 0; j < S.tmp1; ++j) { uint32_t owner = W.scratch[j] >> 4, ei = W.scratch[j] & 15u; W.gr[owner].edges[ei] &= ~(1u << 30); }
    if (S.tmp0 != 1 || S.overflow) S.pc_ok = 0;          // (only a search that ran to its natural end is replayed by eka)
  }
  return wg_bcast(&S.tmp0) == 2;
}

// eka (reference src/Graph.cc:1430-1501) via countRefPath (:2420-2445)
DEVNI void count_ref_path(Ctx &c) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  if (wg_bcastu(&S.source) == LC_NIL) return;
  if (wg_bcastu(&S.sink) != LC_NIL) {
    WG_LANE0 { evt(c, EV_SEARCH, (uint32_t)W.gr[S.source].comp); S.tmp0 = S.tmp1 = S.tmp2 = 0; S.part[0] = 0; S.part[1] = 0; S.part[2] = 0; S.part[3] = 0; S.pc_i = 0; S.pc_rd = 0; }
    const bool replay = wg_bcast(&S.pc_ok) != 0;             // the paths are the ones findRepeatsInGraphPaths found (pcache_store)
    // part[1] = complete, part[2] = allcycles, part[3] = loop state (0 run, 1 stop)
    while (wg_bcastu(&S.part[3]) == 0) {
      WG_LANE0 {
        if (replay) {
          if (S.pc_i < S.pc_n) {
            LC_GLOBAL const uint32_t *pc = W.mv + S.pc_rd;
            const uint32_t n = pc[0], pl = pc[1];
            if (pc[3]) evt(c, EV_DFSLIMIT);
            S.tmp3 = (int)pc[2];
            if (S.tmp3) ++S.part[2];
            ++S.part[1];
            S.part[4] = n;
            S.pc_base = S.pc_rd; S.pc_rd += 4u + 2u * n + pl + (pl + 3u) / 4u; ++S.pc_i;
          } else { if (S.pc_end_dfs) evt(c, EV_DFSLIMIT); S.part[3] = 1; }
        } else {
          uint32_t best = bfs(c);
          if (best == LC_NIL || S.overflow) S.part[3] = 1;
          else {
            S.tmp3 = (W.queue[best].bits & 2) ? 1 : 0;
            if (S.tmp3) ++S.part[2];
            ++S.part[1];
            S.part[4] = (uint32_t)path_unpack(c, best);
          }
        }
      }
      if (wg_bcastu(&S.part[3]) != 0) break;
      {
        const int m = replay ? pcache_load(c) : path_string_wg(c, (int)wg_bcastu(&S.part[4]));
        // Hamming short-cut (reference src/Graph.cc:818-826)
        const int n = wg_bcast(&S.seq_len);
        LC_GLOBAL const uint8_t *rs = LC_CTX(c).B->ref_codes + LC_CTX(c).B->ref_off[S.w] + S.seq_t5;
        WG_LANE0 { S.ps_hd = 0; }
        if (n == m) { WG_FOR(i, n) { if (rs[i] != W.pseq[i]) dev_atomic_add((LC_LDS uint32_t *)&S.ps_hd, 1u); } }
        WG_SYNC();
        const int hd = (n == m) ? wg_uniform(S.ps_hd) : -1;
        const bool need_align = (hd == -1 || hd > 5);
        if (!need_align) {
          const int cap = (int)LC_CTX(c).C->max_w + (int)LC_CTX(c).C->path_cap + 2;
          WG_FOR(i, n) { W.aln[i] = "ACGTN"[rs[i]]; W.aln[cap + i] = "ACGT"[W.pseq[i]]; }
        }
        WG_LANE0 {
          S.part[5] = (uint32_t)m; S.part[6] = need_align ? 1u : 0u;
          if (!need_align) S.part[7] = (uint32_t)n;
          if (n > (int)LC_CTX(c).C->max_w || n < 1 || m < 1) OVF(c);
          if (S.overflow) S.part[3] = 1;
        }
      }
      if (wg_bcastu(&S.part[3]) != 0) break;
      if (wg_bcastu(&S.part[6])) {
        LC_GLOBAL const uint8_t *rs = LC_CTX(c).B->ref_codes + LC_CTX(c).B->ref_off[S.w] + S.seq_t5;
        const int pl = (int)wg_bcastu(&S.part[5]);
        PHASE(c, 12);
        if (!align_fill_band(c, rs, S.seq_len, W.pseq, pl)) align_fill(c, rs, S.seq_len, W.pseq, pl);
        PHASE(c, 13);
        { const int aL = align_traceback_wg(c, rs, S.seq_len, W.pseq, pl); WG_LANE0 { S.part[7] = (uint32_t)aL; } }
        WG_SYNC();
        align_traceback_fill(c, rs, W.pseq, (int)wg_bcastu(&S.part[7]));
      }
      PHASE(c, 14);
      SUBPHASE(c, 4, 2);
      walk_prepare(c, (int)wg_bcastu(&S.part[7]));
      SUBPHASE(c, 4, 14);
#ifdef LANCET_WAVE_EMU
      const bool old_walk = getenv("LANCET_OLD_WALK") != nullptr;      // (emulator only: the one-lane form on every window, to compare the two)
#else
      const bool old_walk = false;
#endif
#ifdef LANCET_LR_WALK_LANE0                /* (tuning builds: --linked-reads windows on the one-lane walk, as until round 6) */
      if (old_walk || wg_uniform(S.LR)) {
#else
      if (old_walk) {
#endif
        WG_LANE0 { if (!S.overflow) process_path_walk(c, (int)S.part[4], (int)S.part[5], (int)S.part[7], (int)S.part[1]); }
      } else if (!wg_bcast(&S.overflow)) {
        process_path_walk_wg(c, (int)wg_bcastu(&S.part[4]), (int)wg_bcastu(&S.part[5]), (int)wg_bcastu(&S.part[7]), (int)wg_bcastu(&S.part[1]));
      }
      WG_LANE0 {
        if (!S.overflow) path_flag_edges(c, (int)S.part[4], 1u);
        if (S.overflow) S.part[3] = 1;
      }
      PHASE(c, 11);
    }
    WG_LANE0 { evt(c, EV_EKA_END, (uint32_t)S.refcomp, S.part[1], S.part[2], (uint32_t)S.tmp0, (uint32_t)S.tmp2, (uint32_t)S.tmp1, S.part[0]); }
  }
  WG_LANE0 {
    if (LC_CTX(c).C->evt_cap) { uint32_t n = 0; for (uint32_t i = 0; i < S.M; ++i) if (W.gr[W.order[i]].onref) ++n; evt(c, EV_FOUND, n); }
  }
  WG_SYNC();
}

// ---------------------------------------------------------------------------------------------------------
// --linked-reads, a graph from the LDS build kernel.  What a node carries in this mode beyond the ordinary counts -- barcode counts per
// strand and sample in place of read counts (cov_distr), haplotype counts, the per-position hpX_minqv counters, Ref_t's coverage
// tables from the barcode counts, the barcode sets of a variant -- is the result of a replay over the node's occurrences in visiting order
// (Node_t::hasBX / addBX / addHP, reference src/Graph.cc:239-317, src/Node.cc:30-118, 502-520; src/Ref.cc:75-170).  The build kernel
// handed the occurrences of every tracked node over as csr runs (layout.h PRE_OFF_LRNOCC / PRE_OFF_LRCSR); load_prebuilt has loaded
// the graph.  Here, for the survivors and for the nodes of reference k-mers:
//   1. per-read words, the runs (copied: the replay sorts them and sets the "grown" bits),
//   2. the node of every reference offset with its mer-table bit (emit_variant_lr: Ref_t::getBXsetAt asks nodes that did not survive too),
//   3. the replay (lr_node_replay / lr_replay_batches, as build_gather runs them) -> kc[], mincov, khp[],
//   4. the survivors' quality rows with all ten counters (build_qcounts, rows only) -- W.qv is the slot's own array, stride 10,
//   5. Ref_t::computeCoverage from the barcode counts (build_refcov's rule, over Ref_t::seq as the k attempts before this one left it),
//   6. the open-addressing table of the survivors' k-mers for Graph_t::getBXsetAt (kmer_lookup; every k-mer of a path string is a live node).
// ---------------------------------------------------------------------------------------------------------
DEVNI void load_prebuilt_lr(Ctx &c, LC_GLOBAL const uint8_t *area) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  LC_GLOBAL const PreLayout &PL = LC_PL(c);
  LC_GLOBAL const PreHdr *H = (LC_GLOBAL const PreHdr *)(area + PRE_OFF_HDR);
  const uint32_t N = H->N, ncand = H->ncand, nsurv = H->nsurv, total = H->lr_total;
  const int K = S.K, NW = S.NW, reflen = S.reflen;
  LC_GLOBAL const uint32_t *lrnocc = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_LRNOCC), *lrcsr = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_LRCSR);
  LC_GLOBAL const uint32_t *occ_ref = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_OCCREF);
  LC_GLOBAL const uint32_t *snode = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_SNODE), *sid = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_SID);
  LC_GLOBAL const unsigned long long *skey = (LC_GLOBAL const unsigned long long *)(area + PRE_OFF_SKEY);
  LC_GLOBAL const uint8_t *surv = (LC_GLOBAL const uint8_t *)(area + PRE_OFF_SURV);
  const int nrefk = reflen - K > 0 ? reflen - K + 1 : 0;
  const int t5 = wg_bcast(&S.seq_t5), L = wg_bcast(&S.seq_len);
  PHASE(c, 2);
  // ---- 1
  WG_LANE0 { W.qv = W.qv_own; }
  WG_FOR(r, S.R) {
    uint32_t rinfo, bw, gw; int tlen; bool isref;
    read_geom(c, r, &rinfo, &bw, &gw, &tlen, &isref);
    W.rd[4 * r] = rinfo; W.rd[4 * r + 1] = bw; W.rd[4 * r + 2] = gw; W.rd[4 * r + 3] = 0;
  }
  WG_FOR(n, N + 1) { W.nocc[n] = lrnocc[n]; }
  WG_FOR(i, total) { const uint32_t v = lrcsr[i]; W_CSR(W)[i] = CS_MAKE(v & 0xFFFFu, (v >> 16) & 0x3FFu, (v >> 26) & 1u, (v >> 27) & 3u); }
  // ---- 2
  WG_FOR(i, (N + 31u) / 32u) { W.bitmap[i] = 0; W.bitpre[i] = 0; }
  WG_SYNC();
  WG_FOR(i, nrefk) {
    const uint32_t n = occ_ref[i] & 0x3FFFFFFFu & ~PB_GONE;
    dev_atomic_or(&W.bitmap[n >> 5], 1u << (n & 31u));
    // Ref_t::mertable (indexMers, src/Ref.cc:40-64): the k-mers at i + K < seq.length() of Ref_t::seq -- all of rawseq for the window's first
    // graph; what an earlier, rejected k of this window trimmed it to (markRefEnds) for a graph built ahead or by the service
    if (i >= t5 && i - t5 < L - K) dev_atomic_or(&W.bitpre[n >> 5], 1u << (n & 31u));
  }
  WG_LANE0 { S.tmp2 = 0; }
  WG_SYNC();
  WG_FOR(i, nrefk) {
    const uint32_t n = occ_ref[i] & 0x3FFFFFFFu & ~PB_GONE;
    W.lr_refnode[i] = n | (((ld2(&W.bitpre[n >> 5]) >> (n & 31u)) & 1u) << 31);
  }
  PHASE(c, 4);
  // ---- 3
  WG_FOR(n, N) {
    const bool onr = ((ld2(&W.bitmap[(uint32_t)n >> 5]) >> ((uint32_t)n & 31u)) & 1u) != 0;
    if (!surv[n] && !onr) continue;
    const uint32_t lo = lrnocc[n], hi = lrnocc[n + 1];
    if (hi - lo >= LR_COOP_MIN && hi - lo <= LR_COOP_MAX && lo < (1u << 24)) {        // by the whole wave, below: listed
      const uint32_t at = dev_atomic_add((LC_LDS uint32_t *)&S.tmp2, 1u);
      W.scratch[2 * (size_t)at] = (uint32_t)n; W.scratch[2 * (size_t)at + 1] = lo | ((hi - lo) << 24);
    } else {
      uint32_t lrv[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (hi > lo) lr_node_replay(c, lo, hi, lrv);                                      // (no run: the reference pseudo-read is the node's only occurrence)
      LC_GLOBAL NodeGr &G = W.gr[n];
      for (int q = 0; q < 4; ++q) G.kc[q] = (uint16_t)lrv[q];
      for (int q = 0; q < 6; ++q) W.khp[6 * (size_t)n + q] = (uint16_t)lrv[4 + q];
      G.mincov = (int)(uint16_t)lrv[0] + (int)(uint16_t)lrv[1] + (int)(uint16_t)lrv[2] + (int)(uint16_t)lrv[3];
    }
  }
  WG_SYNC();
  PHASE(c, 5);
  lr_replay_batches(c, (uint32_t)wg_bcast(&S.tmp2));
  PHASE(c, 6);
  // ---- 4
  WG_FOR(si, nsurv) {
    const uint32_t n = sid[si];
    W.pnodes[si] = n; W.pedges[si] = lrnocc[n]; W.ht_bucket[si] = lrnocc[n + 1] - lrnocc[n]; W.ht_start[si] = W.gr[n].nqv;
  }
  WG_SYNC();
  if (nsurv) build_qcounts(c, nsurv);
  if (wg_bcast(&S.overflow)) return;
  PHASE(c, 3);
  // ---- 5
  WG_FOR(j, reflen) { for (int q = 0; q < 4; ++q) W.refcov[4 * j + q] = 0; for (int q = 0; q < 6; ++q) W.refhp[6 * j + q] = 0; }
  WG_SYNC();
  WG_FOR(i, reflen - K > 0 ? reflen - K : 0) {
    const uint32_t rv = W.lr_refnode[i], X = rv & 0x3FFFFFFFu;
    uint16_t v[4] = {0, 0, 0, 0}, h[6] = {0, 0, 0, 0, 0, 0};
    if (rv >> 31) {                                                  // (computeCoverage reads 0 for a k-mer that is not in the table)
      for (int q = 0; q < 4; ++q) v[q] = W.gr[X].kc[q];
      for (int q = 0; q < 6; ++q) h[q] = W.khp[6 * (size_t)X + q];
    }
    if (i == 0) { for (int j = 0; j < K; ++j) { for (int q = 0; q < 4; ++q) W.refcov[4 * j + q] = v[q]; for (int q = 0; q < 6; ++q) W.refhp[6 * j + q] = h[q]; } }
    else { for (int q = 0; q < 4; ++q) W.refcov[4 * (i + K - 1) + q] = v[q]; for (int q = 0; q < 6; ++q) W.refhp[6 * (i + K - 1) + q] = h[q]; }
  }
  // ---- 6
  WG_LANE0 { uint32_t tcap = 1024; while (tcap < 2u * nsurv + 2u && tcap < LC_CTX(c).C->table_cap) tcap <<= 1; S.tmask = tcap - 1u; S.tfull = 0; }
  const uint32_t mask = wg_bcastu(&S.tmask);
  WG_FOR(i, mask + 1u) { lc_u4 z; z.x = 0; z.y = 0; z.z = LC_NIL; z.w = 0; *(lc_u4 *)(W.slots + 4 * (size_t)i) = z; }
  WG_SYNC_FENCE();
  WG_FOR(ci, ncand) {
    const uint32_t n = snode[ci];
    if (n == LC_NIL) continue;
    unsigned long long ck[LC_NWMAX];
    for (int w = 0; w < LC_NWMAX; ++w) ck[w] = w < NW ? skey[(size_t)ci * PL.kw + (uint32_t)w] : 0ULL;
    unsigned long long h = 0; uint32_t idx;
    if (NW == 1 && K <= 31) { h = ck[0] + 1ULL; idx = (uint32_t)mix64(h) & mask; }      // (kmer_lookup's hashing)
    else {
      for (int w = 0; w < NW; ++w) h = mix64(h ^ (ck[w] + 0x9e3779b97f4a7c15ULL * (unsigned long long)(w + 1)));
      h &= ~(1ULL << 63);
      if (h == 0) h = 1;
      idx = (uint32_t)h & mask;
    }
    for (uint32_t probes = 0; probes <= mask; ++probes) {
      const unsigned long long old = dev_atomic_cas64(&SL_TAG(W, idx), 0ULL, h);
      if (old == 0ULL) { SL_NODE(W, idx) = n; if (!(NW == 1 && K <= 31)) for (int w = 0; w < NW; ++w) W.slot_key[(size_t)idx * LC_NWMAX + w] = ck[w]; break; }
      idx = (idx + 1) & mask;                                      // (an equal 64-bit tag of another long k-mer: the next slot, as the look-up probes on)
    }
  }
  WG_SYNC_FENCE();
}

// ---------------------------------------------------------------------------------------------------------
// The window's first graph as the LDS build kernel left it (build_lds.h, layout.h PreHdr ...): the slot's arrays are
// brought to the state build_graph() leaves them in, as far as the graph phases read them -- std::hash of every node,
// survivor flags, the survivors' records / per-position counts / sequence descriptors, the reference pseudo-read's
// nodes, the reference coverage.  Whole wave, coalesced copies.  Returns false when the window has no such graph for k.
// ---------------------------------------------------------------------------------------------------------
DEVNI bool load_prebuilt(Ctx &c, int k) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  LC_GLOBAL const uint8_t *pre = LC_CTX(c).OUT->pre;
  if (!pre) return false;
  LC_GLOBAL const PreLayout &PL = LC_PL(c);
  LC_GLOBAL const uint8_t *area = pre + (size_t)S.w * PRE_STRIDE;
  LC_GLOBAL const PreHdr *H = (LC_GLOBAL const PreHdr *)(area + PRE_OFF_HDR);
  // the window's first graph, or one the build kernel built ahead for a later k of this loop (PreHdr::next)
  for (int hop = 0; hop < 16 && H->status == PB_BUILT && H->K < k && H->next != 0u && LC_CTX(c).OUT->pre_pool; ++hop) {
    area = LC_CTX(c).OUT->pre_pool + (size_t)(H->next - 1u) * PRE_STRIDE;
    H = (LC_GLOBAL const PreHdr *)(area + PRE_OFF_HDR);
  }
  WG_LANE0 { S.tmp1 = (H->status == PB_BUILT && H->K == k && H->N <= LC_CTX(c).C->node_cap && (size_t)H->ncand * (size_t)k <= (size_t)LC_CTX(c).C->qv_cap &&
                       H->ncand <= LC_CTX(c).C->surv_cap && (size_t)H->ncand * (size_t)k <= (size_t)LC_CTX(c).C->seq_cap) ? 1 : 0;
             // --linked-reads: only with the tracked nodes' occurrences handed over, and room for the runs
             if (S.LR && !(H->lr == 1u && H->have_order == 1u && H->lr_total <= LC_CTX(c).C->occ_cap)) S.tmp1 = 0; }
  if (!wg_bcast(&S.tmp1)) return false;
  const uint32_t N = H->N, ncand = H->ncand, nsurv = H->nsurv;
  const int K = k;
  WG_LANE0 {
    if (S.n_builds > 0 && LC_CTX(c).OUT->n_ahead_used) dev_atomic_add(LC_CTX(c).OUT->n_ahead_used, 1u);
    S.N = N; S.N_last = N; S.O = H->O; S.totalreadbp = (int)H->totalreadbp; S.n_kmers += (unsigned long long)H->n_kmers; ++S.n_builds;
    if (N > S.max_nodes) S.max_nodes = N;
    S.sum_nodes += N;
    S.nspecial = 0; S.qv_top = ncand; S.seq_top = ncand * (uint32_t)K; S.tmask = 0; S.tfull = 0; S.prebuilt = 1;
    S.pre_edges = H->edges_total; S.pre_refn = H->refn;
    W.occ_base[S.R - 1] = 0;                                     // the reference pseudo-read's occurrences are the only ones the graph phases visit
    S.pre_order = H->have_order == 1u ? 1 : 0;
    if (S.pre_order) {                                           // first_lowcov + cleanDead + markConnectedComponents came along
      S.M = nsurv; S.ht_bc = H->ht_bc; S.ht_elt = nsurv; S.ht_next_resize = H->ht_next_resize; S.ht_head = LC_NIL;
      S.numcomp = (int)H->numcomp; S.refcomp = (int)H->refcomp;
    }
  }
  const bool pre_order = H->have_order == 1u;
  LC_GLOBAL const unsigned long long *nhash = (LC_GLOBAL const unsigned long long *)(area + PRE_OFF_NHASH);
  LC_GLOBAL const uint8_t *surv = (LC_GLOBAL const uint8_t *)(area + PRE_OFF_SURV);
  LC_GLOBAL const uint32_t *snode = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_SNODE), *sid = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_SID);
  LC_GLOBAL const unsigned long long *skey = (LC_GLOBAL const unsigned long long *)(area + PRE_OFF_SKEY);
  LC_GLOBAL const uint32_t *occ_ref = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_OCCREF);
  LC_GLOBAL const lc_v4 *pgr = (LC_GLOBAL const lc_v4 *)(area + PRE_OFF_PGR);
  LC_GLOBAL const lc_v4 *qsrc = (LC_GLOBAL const lc_v4 *)(area + PRE_OFF_QV);
  const uint32_t dummy = LC_CTX(c).C->node_cap + LC_CTX(c).C->special_cap;       // stand-in for reference k-mers whose node is gone
  // markRefEnds and the first compress came along too (a single-component first graph): only the ~20 unitigs are loaded, the window
  // kernel goes on at hasCycle (process_window).  The special nodes get this slot's ids (PB_SPECIAL + k in the hand-off).
  LC_GLOBAL const PreCmp *CH = (LC_GLOBAL const PreCmp *)(area + PRE_OFF_CHDR);
  WG_LANE0 { S.tmp1 = (S.seq_t5 != 0 || S.seq_len != S.reflen) ? 1 : 0; }       // Ref_t::seq trimmed by an earlier k of this window (see below)
  const bool trimmed_before = wg_bcast(&S.tmp1) != 0;
  const int t5_before = wg_uniform(S.seq_t5), len_before = wg_uniform(S.seq_len);     // (Ref_t::seq as the k attempts before this one left it: markRefEnds of THIS graph, when it came along, sets S.seq_* anew below)
  // (round 6: also when an earlier k has trimmed Ref_t::seq -- the build service's graphs: neither markRefEnds nor the compress looks at
  //  Ref_t::seq, and what does is set right below, `trimmed_before`)
  const bool cdone = pre_order && CH->done == 1u && LC_CTX(c).C->special_cap >= 2u && CH->m_live <= LC_CTX(c).C->node_cap &&
                     (size_t)ncand * (size_t)K + CH->seqn <= (size_t)LC_CTX(c).C->seq_cap;
  WG_LANE0 { S.cmp_done = cdone ? 1 : 0; }
  if (cdone) {
    const uint32_t mlive = CH->m_live, seqn = CH->seqn, ncap = LC_CTX(c).C->node_cap;
    LC_GLOBAL const uint32_t *clive = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_CLIVE);
    LC_GLOBAL const uint32_t *cseq = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_CSEQ);
    WG_LANE0 {
      const int so = CH->src_off, ko = CH->snk_off;
      S.M = mlive; S.ht_elt = mlive; S.nspecial = 2; S.source = ncap; S.sink = ncap + 1u;
      S.seq_t5 = so; S.seq_len = ko - so + K; S.trim5 = so & 0xFFFF; S.trim3 = (S.reflen - ko - K) & 0xFFFF;
      S.seq_top = ncand * (uint32_t)K + seqn; S.cmp_dead = CH->dead; S.cmp_edges0 = CH->edges0; S.cmp_nsurv = nsurv; S.cmp_edges_all = CH->edges_all; S.cmp_n1 = CH->n_c1; S.cmp_refmask = CH->refmask;
    }
    // what the descriptors of merged k-mers still point at: the k-mer's counts and its quality-count row (first and last 16 bytes of a record)
    WG_FOR(si, nsurv) {
      const uint32_t n = sid[si];
      LC_GLOBAL lc_v4 *dst = (LC_GLOBAL lc_v4 *)&W.gr[n];
      lc_v4 a = pgr[8 * (size_t)si]; const lc_v4 z = pgr[8 * (size_t)si + 7];
      a.x |= NF_DEAD;                                                // (merged into a unitig -- unless it is in the list of live nodes, whose records follow)
      dst[0] = a; dst[7] = z;
    }
    WG_SYNC();
    WG_FOR(i, mlive) {                                              // the live nodes: whole records, in table order
      const uint32_t idx = clive[i];
      const bool sp = (idx & 0x80000000u) != 0;
      const uint32_t ri = sp ? nsurv + (idx & 0xFFu) : idx;
      const uint32_t id = sp ? ncap + (idx & 0xFFu) : sid[ri];
      W.order[i] = id;
      W.nhash[id] = sp ? CH->spec_hash[idx & 0xFFu] : nhash[id];
      lc_v4 r[8];
      for (int q = 0; q < 8; ++q) r[q] = pgr[8 * (size_t)ri + q];
      for (int q = 1; q < 4; ++q) {                                 // edges[12]: the special nodes' ids
        if (ED_TO(r[q].x) >= PB_SPECIAL) r[q].x = (r[q].x & 0xF0000000u) | (ncap + (ED_TO(r[q].x) - PB_SPECIAL));
        if (ED_TO(r[q].y) >= PB_SPECIAL) r[q].y = (r[q].y & 0xF0000000u) | (ncap + (ED_TO(r[q].y) - PB_SPECIAL));
        if (ED_TO(r[q].z) >= PB_SPECIAL) r[q].z = (r[q].z & 0xF0000000u) | (ncap + (ED_TO(r[q].z) - PB_SPECIAL));
        if (ED_TO(r[q].w) >= PB_SPECIAL) r[q].w = (r[q].w & 0xF0000000u) | (ncap + (ED_TO(r[q].w) - PB_SPECIAL));
      }
      LC_GLOBAL lc_v4 *dst = (LC_GLOBAL lc_v4 *)&W.gr[id];
      for (int q = 0; q < 8; ++q) dst[q] = r[q];
      // a k-mer node that stayed on its own: its K descriptors (the merged ones are in the heads' deques, the heads' own k-mers too)
      const uint32_t nqv = r[7].x, slo = r[5].z, shi = r[5].w;
      if (!sp && nqv != LC_NIL && slo == nqv * (uint32_t)K && shi == slo + (uint32_t)K) {
        for (int t = 0; t < K; ++t) W.seq[slo + (uint32_t)t] = SD_MAKE(id, t, pre_key_base(skey, PL.kw, nqv, K, t));
      }
    }
    { const uint32_t base = ncand * (uint32_t)K; WG_FOR(i, seqn) { W.seq[base + (uint32_t)i] = cseq[i]; } }
    WG_LANE0 { S.seq_lazy = 0; }
    // ---- the coverage of the unitig heads: the float averaging of the merges (reference src/Graph.cc:2632-2636) in merge order, which the
    //      build kernel leaves to this wave (round 6: there it held a 512-lane workgroup for 25 us per window on four lanes).  One lane per
    //      (head, coverage): nc starts as the head's own k-mer's count and takes the merged k-mers' counts one by one,
    //          nc = ((nc * amer) + (bc * bmer)) / (amer + bmer),  amer = t + 1, bmer = 1   (the reference's expression, IEEE float division).
    //      The operands come in chunks of LC_COV_CHUNK through LDS (the staging area, idle here): straight from HBM every fourth merge
    //      waited a round trip -- 80 us per window instead of 25.  A head whose slice spans chunks keeps its value in its record in between.
    const uint32_t nch = CH->cov_heads;
    if (nch) {
      LC_GLOBAL const unsigned long long *cord = (LC_GLOBAL const unsigned long long *)(area + PRE_OFF_CORD);
      LC_GLOBAL const uint32_t *chl = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_CHL);
      constexpr uint32_t LC_COV_CHUNK = 256u;
      static_assert(8u * LC_COV_CHUNK <= sizeof(S.lbytes), "merge operands staged in LDS");
      volatile LC_LDS unsigned long long *ops = (volatile LC_LDS unsigned long long *)&S.lbytes[0];
      const uint32_t nabs = CH->dead;
      WG_SYNC();                                                    // (the heads' records are in place: their cov[] is rewritten below)
      for (uint32_t c0 = 0; c0 < nabs; c0 += LC_COV_CHUNK) {
        const uint32_t c1 = c0 + LC_COV_CHUNK < nabs ? c0 + LC_COV_CHUNK : nabs;
        WG_FOR(i, c1 - c0) { ops[i] = cord[c0 + (uint32_t)i]; }
        WG_SYNC();
        WG_FOR(x, 4 * nch) {
          const uint32_t hx = (uint32_t)x >> 2, q = (uint32_t)x & 3u;
          const uint32_t st = chl[3 * hx + 1], cnt = chl[3 * hx + 2];
          const uint32_t lo = st > c0 ? st : c0, hi = st + cnt < c1 ? st + cnt : c1;
          if (lo >= hi) continue;                                   // (nothing of this head's slice in the chunk)
          LC_GLOBAL NodeGr &G = W.gr[sid[chl[3 * hx]]];
          float nc = lo == st ? (float)(uint32_t)G.kc[q] : G.cov[q];
#ifndef LANCET_WAVE_EMU
          __builtin_amdgcn_s_setprio(3);                            // (a chain of dependent operations: every issue slot it can get)
#endif
          float famer = (float)((int)(lo - st) + 1);                // (amer and amer + bmer as floats: small integers, exactly what the conversions give)
          for (uint32_t g = lo; g < hi; ++g) {
            const float bc = (float)(uint32_t)((ops[g - c0] >> (16 * q)) & 0xFFFFu);
            nc = ((nc * famer) + bc) / (famer + 1.0f);
            famer += 1.0f;
          }
#ifndef LANCET_WAVE_EMU
          __builtin_amdgcn_s_setprio(0);
#endif
          G.cov[q] = nc;
        }
        WG_SYNC();
      }
    }
  } else {
  if (pre_order) {                                               // only the survivors' hashes are looked at again (unordered_map::insert of the special nodes)
    LC_GLOBAL const uint32_t *ord = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_ORDER);
    WG_FOR(i, nsurv) { const uint32_t n = ord[i]; W.order[i] = n; W.nhash[n] = nhash[n]; }
  } else { WG_FOR(n, N) { W.nhash[n] = nhash[n]; } }
  WG_FOR(l, LANCET_WG) {                                          // four 16-byte pieces in flight per lane (the loads first, then the stores)
    const int total = (int)(nsurv * 8u);
    for (int i0 = l; i0 < total; i0 += 4 * LANCET_WG) {
      lc_v4 v[4]; uint32_t dn[4];
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * LANCET_WG; const int ic = i < total ? i : total - 1; v[u] = pgr[(size_t)ic]; dn[u] = sid[(uint32_t)ic >> 3]; }
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * LANCET_WG; if (i < total) ((LC_GLOBAL lc_v4 *)&W.gr[dn[u]])[(uint32_t)i & 7u] = v[u]; }
    }
  }
  // The survivors' sequence descriptors (K words each: 32 KB per window) are NOT written here: almost every k-mer node is merged
  // into a unitig by the first compress, which needs only the first and the last descriptor of a node -- both follow from its
  // key.  compress_prepare / compress_rank derive what they need (seq_desc_lazy) and write out the K descriptors of the few
  // nodes that stay on their own; any other route into the graph phases calls seq_materialize_all first.
  WG_LANE0 { S.seq_lazy = 1; S.lz_area = (unsigned long long)(uintptr_t)area; S.lz_skey = (unsigned long long)(uintptr_t)skey; S.lz_kw = PL.kw; }
  (void)snode; (void)skey;
  }
  WG_FOR(i, (N + 3) / 4) { ((LC_GLOBAL uint32_t *)W.survb)[i] = ((LC_GLOBAL const uint32_t *)surv)[i]; }
  WG_LANE0 { W.qv = (LC_GLOBAL uint16_t *)(area + PRE_OFF_QV); }  // (read in place: 73 KB per window not copied; build_graph points qv back at the slot's array)
  (void)qsrc;
  const int nrefk = S.reflen - K > 0 ? S.reflen - K + 1 : 0;
  WG_FOR(i, nrefk) { const uint32_t e = occ_ref[i]; W.occ[i] = (e & PB_GONE) ? (dummy | (e & 0x80000000u)) : e; }
  WG_FOR(i, S.reflen * 2) { ((LC_GLOBAL uint32_t *)W.refcov)[i] = ((LC_GLOBAL const uint32_t *)(area + PRE_OFF_REFCOV))[i]; }
  WG_LANE0 { W.gr[dummy].flags = 0; }
  // A graph built ahead was built as if it were the window's first: Ref_t::mertable indexed over the whole rawseq.  By now an
  // earlier k of this window may have trimmed Ref_t::seq to its anchors (markRefEnds, reference src/Graph.cc:2200-2228) before it
  // was rejected, and indexMers (src/Ref.cc:40-64) runs over that shorter seq at every k: the k-mers outside it
  // are not in the table, their nodes are not reference nodes (unless another k-mer inside is the same node) and
  // computeCoverage (src/Ref.cc:173-250) reads 0 for them.  Same rule as build_refcov, applied to what came along.
  if (trimmed_before) {
    const int t5 = t5_before, L = len_before;
    WG_FOR(i, (N + 31u) / 32u) { W.bitmap[i] = 0; }
    WG_SYNC();
    WG_FOR(i, L - K > 0 ? L - K : 0) {
      const int p = t5 + i;
      if (p < nrefk) { const uint32_t n = occ_ref[p] & 0x3FFFFFFFu; dev_atomic_or(&W.bitmap[n >> 5], 1u << (n & 31u)); }
    }
    WG_SYNC();                                                     // (the bitmap is read past the L1 below: ld2)
    WG_LANE0 { S.tmp1 = 0; }
    WG_FOR(si, nsurv) {
      const uint32_t n = sid[si]; const uint32_t f = W.gr[n].flags;
      const bool in = ((ld2(&W.bitmap[n >> 5]) >> (n & 31u)) & 1u) != 0;
      W.gr[n].flags = in ? (f | NF_INMER) : (f & ~(uint32_t)NF_INMER);
      if (in) S.tmp1 = 1;
    }
    WG_SYNC();
    WG_LANE0 { if (S.cmp_done) S.refcomp = S.tmp1 ? 1 : 0; }          // (the one component holds a reference k-mer of the trimmed seq or not: markConnectedComponents' count)
    WG_FOR(i, S.reflen - K > 0 ? S.reflen - K : 0) {
      const uint32_t n = occ_ref[i] & 0x3FFFFFFFu;
      if (!((ld2(&W.bitmap[n >> 5]) >> (n & 31u)) & 1u)) {
        if (i == 0) { for (int j = 0; j < K; ++j) for (int q = 0; q < 4; ++q) W.refcov[4 * j + q] = 0; }
        else { for (int q = 0; q < 4; ++q) W.refcov[4 * (i + K - 1) + q] = 0; }
      }
    }
    if (LC_CTX(c).C->evt_cap) {                                   // trace only: nodes that hold a reference k-mer (markRefNodes' count)
      WG_LANE0 { uint32_t cnt = 0; for (uint32_t i = 0; i < (N + 31u) / 32u; ++i) cnt += (uint32_t)dev_popc(ld2(&W.bitmap[i])); S.pre_refn = cnt; }
    }
  }
  WG_SYNC();
  if (wg_uniform(S.LR)) load_prebuilt_lr(c, area);
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// Build service (layout.h SvcCtl): the loop needs a graph at k that nobody built in LDS.  If the window is one the LDS build
// kernel can take (its first graph came from the configuration the service runs) the state the reference carries from one k to the
// next (Ref_t::seq / trim of the last markRefEnds, the variants emitted so far) goes into a continuation record,
// a request is posted and the slot is free for another window.  Returns true when the window was suspended.
// ---------------------------------------------------------------------------------------------------------
#define LANCET_W_SUSPENDED 0x5355
DEVNI bool try_suspend(Ctx &c, int k) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W;
  LC_GLOBAL SvcCtl *sv = LC_CTX(c).OUT->svc;
  if (!sv) return false;
  const int w = wg_uniform(S.w);
  LC_GLOBAL const PreLayout &PL = LC_PL(c);
  WG_LANE0 {
    S.tmp1 = 0;
    LC_GLOBAL const PreHdr *H0 = (LC_GLOBAL const PreHdr *)(LC_CTX(c).OUT->pre + (size_t)w * PRE_STRIDE + PRE_OFF_HDR);
    if (k != S.nosusp_k && (k & 1) && !ld2(&sv->nosvc) && (k <= 31 || (sv->large && 2 * k <= 64 * (int)PL.kw)) && !S.overflow && LC_CTX(c).OUT->pre_pool && H0->status == PB_BUILT && (!H0->big || sv->large)) {
      const uint32_t i = dev_atomic_add(&sv->req_alloc, 1u);
      if (i < sv->cap) { S.tmp1 = 1; S.svc_i = i; }
    }
  }
  if (!wg_bcast(&S.tmp1)) return false;
  const uint32_t i = wg_bcastu(&S.svc_i);
  const uint32_t ecap = LC_CTX(c).C->evt_cap;
  if (ecap) { WG_FOR(j, S.evt_len) { LC_CTX(c).OUT->evt_out[(size_t)w * ecap + (uint32_t)j] = W.evt[j]; } }
  PHASE(c, 0);
  WG_LANE0 {
    LC_GLOBAL SvcCont &ct = sv->cont[i];
    ct.k = k; ct.seq_t5 = S.seq_t5; ct.seq_len = S.seq_len; ct.trim5 = S.trim5; ct.trim3 = S.trim3; ct.emit_seq = S.emit_seq;
    ct.n_builds = S.n_builds; ct.final_k = S.final_k; ct.max_nodes = S.max_nodes; ct.evt_len = S.evt_len; ct.N_last = S.N_last; ct.sum_nodes = S.sum_nodes;
    ct.n_kmers = S.n_kmers;
    sv->req[i].w = (uint32_t)w; sv->req[i].k = k;
#ifdef LANCET_PROF_TIMELINE
    S.phase_acc[4] = wall_clock64();                          // (when it was put aside)
#endif
    if (LC_CTX(c).OUT->phase) for (int q = 0; q < 16; ++q) LC_CTX(c).OUT->phase[(size_t)w * 16 + q] = S.phase_acc[q];
    S.status = LANCET_W_SUSPENDED;
  }
  WG_SYNC();
  WG_LANE0 { st_rel(&sv->req[i].state, SV_POSTED); }
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// the window: Microassembler::processGraph (reference src/Microassembler.cc:73-249)
// rq: -1, or the request (SvcCtl::req) whose graph is ready -- the window resumes at that k
// ---------------------------------------------------------------------------------------------------------
DEV void process_window(Ctx &c, int w, int rq = -1) {
  LC_WS &S = LC_SREF(c); LC_GLOBAL Work &W = *LC_CTX(c).W; LC_GLOBAL const DevBatch &B = *LC_CTX(c).B;
  LC_GLOBAL const PreLayout &PL = LC_PL(c);
  WG_LANE0 {
    S.items_ready = 0;
    S.w = w; S.overflow = 0; S.evt_len = 0; S.emit_seq = 0; S.n_kmers = 0; S.max_nodes = 0; S.sum_nodes = 0; S.n_builds = 0; S.N_last = 0; S.final_k = 0;
    S.status = LANCET_W_OK;
    S.LR = LC_CTX(c).C->lr_mode ? 1 : 0; S.QS = S.LR ? 10 : 4;
    S.reflen = (int)(B.ref_off[w + 1] - B.ref_off[w]);
    int nr = (int)(B.read_begin[w + 1] - B.read_begin[w]);
    S.R = nr + 1;                                   // + the reference pseudo-read, appended last (Graph.cc:535-540)
    S.seq_t5 = 0; S.seq_len = S.reflen; S.trim5 = 0; S.trim3 = 0;
    S.tmp0 = 0;
    if ((uint32_t)S.R > LC_CTX(c).C->reads_cap || (uint32_t)S.R > LC_READS_MAX || S.reflen > (int)LC_CTX(c).C->max_w || S.reflen < 1) S.overflow = 1;   // csr / item words keep the read in 16 bits (in 32 / 21 in the re-run tier: layout.h cs_t)
    if (LC_WIDE_IDS && !LC_CTX(c).C->wide_ids) S.overflow = 1;                                  // (a work space laid out for the 32-bit words)
    S.hasN = 0;
    S.nosusp_k = -1;
    if (rq >= 0) {                                  // what the window carried when it was suspended
      LC_GLOBAL const SvcCont &ct = LC_CTX(c).OUT->svc->cont[rq];
      S.seq_t5 = ct.seq_t5; S.seq_len = ct.seq_len; S.trim5 = ct.trim5; S.trim3 = ct.trim3; S.emit_seq = ct.emit_seq;
      S.n_builds = ct.n_builds; S.final_k = ct.final_k; S.max_nodes = ct.max_nodes; S.sum_nodes = ct.sum_nodes; S.evt_len = ct.evt_len; S.N_last = ct.N_last;
      S.n_kmers = ct.n_kmers; S.nosusp_k = ct.k;
    }
  }
  if (rq >= 0 && LC_CTX(c).C->evt_cap) {
    const uint32_t ecap = LC_CTX(c).C->evt_cap, el = wg_bcastu(&S.evt_len);
    WG_FOR(j, el) { W.evt[j] = LC_CTX(c).OUT->evt_out[(size_t)w * ecap + (uint32_t)j]; }
    WG_SYNC();
  }
  LC_GLOBAL const PreHdr *H0 = LC_CTX(c).OUT->pre ? (LC_GLOBAL const PreHdr *)(LC_CTX(c).OUT->pre + (size_t)w * PRE_STRIDE + PRE_OFF_HDR) : nullptr;
  WG_LANE0 { S.tmp1 = (H0 && H0->have_rep == 1u) ? 1 : 0; if (S.tmp1) { S.tmp0 = (int)H0->mapped; S.hasN = 0; } }     // (the build kernel counted them; it scans no window with N)
  if (!wg_bcast(&S.tmp1)) {                                     // mapped reads, N in the window reference: all lanes
    WG_SYNC();
    const uint32_t r0 = B.read_begin[w]; const int nr = S.R - 1, rl = S.reflen;
    uint32_t mine = 0;
    WG_FOR(r, nr) { if (RI_MAPPED(B.rinfo[r0 + (uint32_t)r])) ++mine; }
    WG_FOR(l, LANCET_WG) { if (mine) dev_atomic_add((LC_LDS uint32_t *)&S.tmp0, mine); mine = 0; }
    const uint32_t f0 = B.ref_off[w];
    WG_FOR(i, rl) { if (B.ref_codes[f0 + (uint32_t)i] > 3) S.hasN = 1; }
    WG_SYNC();
  }
  WG_LANE0 { if (S.tmp0 > 0 && rq < 0) evt(c, EV_PROCESS, (uint32_t)(S.R - 1), (uint32_t)S.tmp0); }
  if (wg_bcast(&S.tmp0) <= 0) { WG_LANE0 { S.status = LANCET_W_NO_READS; } WG_SYNC(); return; }     // Microassembler.cc:83
  if (wg_bcast(&S.overflow)) { WG_LANE0 { S.status = LANCET_W_OVERFLOW; } WG_SYNC(); return; }
  const int reflen = wg_bcast(&S.reflen);
  PHASE(c, 1);
  {
    // isRepeat / isAlmostRepeat operands of the window reference: taken from the LDS build kernel when it scanned this window
    LC_GLOBAL const uint8_t *pre = LC_CTX(c).OUT->pre;
    LC_GLOBAL const PreHdr *H = pre ? (LC_GLOBAL const PreHdr *)(pre + (size_t)w * PRE_STRIDE + PRE_OFF_HDR) : nullptr;
#ifdef BL_DBG_NOREP
    WG_LANE0 { S.tmp1 = 0; (void)H; }
#else
    WG_LANE0 { S.tmp1 = (H && H->have_rep == 1u) ? 1 : 0; if (S.tmp1) { S.repE = H->refE; S.repM = H->refM; } }
#endif
    if (!wg_bcast(&S.tmp1)) repeat_scan_min(S.rs, B.ref_codes + B.ref_off[w], reflen, LC_CTX(c).P->max_mismatch, LC_CTX(c).P->min_k, LC_CTX(c).P->min_k + 1, &S.repE, &S.repM, nullptr, &S.rs_bad);
  }
  PHASE(c, 0);
  const int refE = wg_bcast(&S.repE), refM = wg_bcast(&S.repM);
  int rptInRef = 0, rptInQry = 0, cycleInGraph = 0;
  bool processed = false;
  const int k_first = rq >= 0 ? wg_uniform(LC_CTX(c).OUT->svc->cont[rq].k) : LC_CTX(c).P->min_k;
  for (int k = k_first; k <= LC_CTX(c).P->max_k; k += 2) {
    rptInRef = rptInQry = cycleInGraph = 0;
    // isRepeat / isAlmostRepeat on rawseq (Microassembler.cc:118-131)
    if (reflen - k > 0 && refE >= k) { WG_LANE0 { evt(c, EV_REPEAT_REF, k); } rptInRef = 1; continue; }
    if (reflen - k > 0 && refM >= k + 1) { WG_LANE0 { evt(c, EV_NEAR_REF, k); } rptInRef = 1; continue; }
    WG_LANE0 { S.K = k; S.NW = (2 * k + 63) / 64; S.final_k = k; S.source = LC_NIL; S.sink = LC_NIL; if (k > 127 || S.NW > LC_NWMAX) S.overflow = 1; }
    if (wg_bcast(&S.overflow)) break;
    WG_LANE0 { S.prebuilt = 0; S.pre_order = 0; S.seq_lazy = 0; if (LC_CTX(c).OUT->svc) dev_atomic_add(&LC_CTX(c).OUT->svc->beat, 1u); }
    if (!load_prebuilt(c, k)) {
      if (try_suspend(c, k)) return;
      build_graph(c);
    }
    if (wg_bcast(&S.overflow)) break;
    PHASE(c, 7);
    const bool pre_order = wg_bcast(&S.pre_order) != 0;
    if (!pre_order) first_lowcov(c);
    STOP_SET(c, 7);
    if (wg_bcast(&S.overflow)) break;
    PHASE(c, 8);
    // ---- everything below is the (small) cleaned graph
    WG_LANE0 {
      evt(c, EV_READS, (uint32_t)S.R, (uint32_t)S.reflen, (uint32_t)S.totalreadbp);
      // trace: printStats(0) over the full table, markRefNodes, removeLowCov(false,0)
      if (LC_CTX(c).C->evt_cap) {
        uint32_t edges = 0, refn = 0, low = 0;
        if (S.prebuilt) { edges = S.pre_edges; refn = S.pre_refn; for (uint32_t n = 0; n < S.N; ++n) if (!W.survb[n]) ++low; }
        else for (uint32_t n = 0; n < S.N; ++n) { edges += W.gr[n].necnt; if (W.gr[n].flags & NF_INMER) ++refn; if (!(W.gr[n].flags & NF_SURV)) ++low; }
        evt(c, EV_STATS, 0, S.N, edges, S.N * (uint32_t)S.K);
        evt(c, EV_MARKREF, S.N, refn);
        evt(c, EV_LOWCOV, low);
      }
    }
    // removeNode for every non-survivor: drop the reciprocal edges, then erase from the table.  A node without NF_SURV is
    // gone from here on (it is in no edge list and not in order[]); it does not need a NF_DEAD mark of its own.
    if (!wg_uniform(S.prebuilt)) {                          // (a prebuilt graph arrives with the edges filtered)
      WG_FOR(n, S.N) {
        if (!(W.gr[n].flags & NF_SURV)) continue;
        LC_GLOBAL uint32_t *e = W.gr[n].edges; int cnt = (int)W.gr[n].necnt, m = 0;
        for (int i = 0; i < cnt; ++i) if (W.gr[ED_TO(e[i])].flags & NF_SURV) e[m++] = e[i];
        W.gr[n].necnt = m;
      }
    }
    WG_SYNC();
    const bool cdone = pre_order && wg_bcast(&S.cmp_done) != 0;    // markRefEnds and the first compress came along too: the trace lines of the stages before them from the counts
    if (cdone) {
      WG_LANE0 {
        if (LC_CTX(c).C->evt_cap) {
          evt(c, EV_CLEANDEAD, S.N - S.cmp_nsurv);
          evt(c, EV_STATS, 0, S.cmp_nsurv, S.cmp_edges_all, S.cmp_nsurv * (uint32_t)S.K);
          evt(c, EV_CC, S.cmp_nsurv);
          if (S.numcomp == 1) { if (S.refcomp) evt(c, EV_CCID, 1u); }                     // (refcomp: after an earlier k's trim it is load_prebuilt's, not the mask's)
          else for (int q = 1; q <= S.numcomp && q <= 32; ++q) if ((S.cmp_refmask >> (q - 1)) & 1u) evt(c, EV_CCID, (uint32_t)q);
          evt(c, EV_CCEND, (uint32_t)S.numcomp, (uint32_t)S.refcomp);
        }
      }
    } else if (pre_order) {
      WG_LANE0 {
        evt(c, EV_CLEANDEAD, S.N - S.M);
        print_stats(c, 0, true);
        if (LC_CTX(c).C->evt_cap) {                             // markConnectedComponents' trace lines from the numbering that came along
          evt(c, EV_CC, S.M);
          LC_GLOBAL uint32_t *tc = W.scratch;
          for (int q = 0; q <= S.numcomp; ++q) tc[q] = 0;
          for (uint32_t i = 0; i < S.M; ++i) { const uint32_t n = W.order[i]; if (W.gr[n].flags & NF_INMER) tc[W.gr[n].comp] = 1; }
          int rc = 0;
          for (int q = 1; q <= S.numcomp; ++q) if (tc[q]) { evt(c, EV_CCID, (uint32_t)q); ++rc; }
          S.refcomp = rc;                                       // (= the count that came along, unless Ref_t::seq has been trimmed since)
          evt(c, EV_CCEND, (uint32_t)S.numcomp, (uint32_t)S.refcomp);
        }
      }
    } else {
      clean_dead_wg(c);
      WG_LANE0 { print_stats(c, 0); }
      mark_connected_components_wg(c);
    }
    int numcomp = wg_bcast(&S.numcomp);
    bool brk = false;
    PHASE(c, 9);
    STOP_SET(c, 8);
    if (wg_bcast(&S.overflow)) break;
    for (int comp = 1; comp <= numcomp; ++comp) {
      if (cdone && comp == 1) {                                       // (the build kernel's markRefEnds + compress are component 1's; the others go the whole way here)
        WG_LANE0 {
          if (LC_CTX(c).C->evt_cap) {
            evt(c, EV_STATS, 1u, S.cmp_nsurv, S.cmp_edges0, S.cmp_n1 * (uint32_t)S.K);
            evt(c, EV_TRIM, (uint32_t)S.seq_t5, (uint32_t)(S.reflen - (S.seq_t5 + S.seq_len)), (uint32_t)S.seq_len);
          }
          S.tmp1 = (int)S.cmp_dead; S.tmp2 = 0;
        }
      } else {
      SUBPHASE(c, 2, 2);
      mark_ref_scan(c, comp);
      WG_LANE0 { print_stats(c, comp); }
      SUBPHASE(c, 2, 3);
      mark_ref_ends(c, comp);
      SUBPHASE(c, 2, 9);
      if (wg_bcast(&S.overflow)) break;
      PHASE(c, 15);
      compress_prepare(c, comp);
      }
      const bool ranked = (cdone && comp == 1) || (wg_bcast(&S.cmp_ok) && compress_rank(c, comp));     // (whole wave; false: a ring or an irregular link, nothing touched)
      if (!ranked) seq_materialize_all(c);
      if (!ranked) WG_LANE0 {
        // The reference runs hasCycle on the k-mer graph and compresses only if there is none.  Unitig compaction merges
        // nodes across links that are the only edge on both sides, which neither creates nor removes a walk that comes
        // back to a node on the DFS stack, so the answer is the same on the compacted graph -- with ~30x fewer nodes to
        // visit.  The graph of a rejected k is thrown away, so compacting first is unobservable; only the trace lines
        // of the compaction are held back until the cycle check has passed.
        S.tmp1 = (int)(S.cmp_ok ? compress_fast(c, comp, true) : compress(c, comp, true));
      }
      SUBPHASE(c, 1, 15);
      if (wg_bcast(&S.tmp2)) compact_absorbed_wg(c);
      PHASE(c, 9);
      SUBPHASE(c, 2, 4);
      const bool gcached = graph_cache_wg(c);                         // (all lanes: the component's edge lists into LDS for the walk below)
      WG_LANE0 {
        const uint32_t dead = (uint32_t)S.tmp1;
        S.tmp0 = 0;
        if (!S.overflow && (gcached ? has_cycle_cached(c) : has_cycle(c))) S.tmp0 = 1;
        if (!S.tmp0 && !S.overflow) {
          evt(c, EV_COMPRESS); evt(c, EV_CLEANDEAD, dead);
          print_stats(c, comp);
        }
        S.tmp3 = (!S.tmp0 && !S.overflow) ? 1 : 0;
      }
      if (wg_bcast(&S.tmp3)) {
        SUBPHASE(c, 2, 5);
        remove_low_cov_wg(c, comp);
        SUBPHASE(c, 2, 6);
        if (!wg_bcast(&S.overflow)) remove_tips_wg(c, comp);
        SUBPHASE(c, 2, 7);
        if (!wg_bcast(&S.overflow)) remove_short_links_wg(c, comp);
        SUBPHASE(c, 2, 4);
        { const bool gc2 = graph_cache_wg(c); WG_LANE0 { if (!S.overflow && (gc2 ? has_cycle_cached(c) : has_cycle(c))) S.tmp0 = 1; } }
        SUBPHASE(c, 2, 9);
      }
      if (wg_bcast(&S.overflow)) break;
      if (wg_bcast(&S.tmp0)) { cycleInGraph = 1; brk = true; break; }
      PHASE(c, 10);
      if (repeats_in_graph_paths(c)) { rptInQry = 1; brk = true; break; }
      PHASE(c, 11);
      if (wg_bcast(&S.overflow)) break;
      count_ref_path(c);
      PHASE(c, 9);
      if (wg_bcast(&S.overflow)) break;
    }
    PHASE(c, 0);
    if (wg_bcast(&S.overflow)) break;
    if (brk) continue;
    processed = true;
    break;
  }
  WG_LANE0 {
    evt(c, EV_END, (uint32_t)rptInRef, (uint32_t)rptInQry, (uint32_t)cycleInGraph);
    S.status = S.overflow ? LANCET_W_OVERFLOW : (processed ? LANCET_W_OK : LANCET_W_K_EXHAUSTED);
  }
  WG_SYNC();
}

// entry: persistent workgroup pulling windows off the batch queue -- and, with the build service, suspended windows whose graph
// is ready off the ready list.  Returns 1 only in the host emulation: nothing to do until the service has run.
DEV int window_kernel_body(LC_GLOBAL const lancet_params *P, LC_GLOBAL const DevBatch *B, LC_GLOBAL const EngineCaps *C, LC_GLOBAL Work *works, LC_GLOBAL DevOut *OUT, LC_WS *S, int slot) {
  Ctx c; c.P = P; c.B = B; c.C = C; c.W = works + slot; c.OUT = OUT; c.S = S;
  LC_CTX_PUBLISH(c);
  LC_GLOBAL SvcCtl *sv = OUT->svc;
  int idle = 0;                                             // consecutive waits (a slot that has waited ~50 ms takes a request back whatever the service does)
  bool waiting = false;                                     // this slot is counted in SvcCtl::n_waiting (uniform)
  while (true) {
    WG_LANE0 {
      int a = 3, g = 0; bool got = false, just_registered = false;   // 0 new window, 1 resume (graph ready), 2 resume (request taken back), 3 leave, 4 wait, 5 wait (newly counted)
      if (sv) {
        const uint32_t h = ld2(&sv->rdy_head);
        if (h < sv->cap) {
          uint32_t v = ld2(&sv->rdy[h]);
          if (v != 0u) v = ld_acq(&sv->rdy[h]);                  // (the acquire -- an L1 / L2 invalidate -- only when there is something to take)
          if (v != 0u && dev_atomic_cas32(&sv->rdy_head, h, h + 1u) == h) { a = 1; g = (int)(v - 1u); got = true; dev_atomic_add(&sv->n_resumed, 1u); }
        }
      }
      if (!got) {
        const uint32_t limit = OUT->win_list ? OUT->n_list : (uint32_t)B->n_windows;
        if (ld2(OUT->queue_head) < limit) { const uint32_t q = dev_atomic_add(OUT->queue_head, 1u); if (q < limit) { a = 0; g = (int)q; got = true; } }
      }
      if (!got && sv) {
        uint32_t posted = ld2(&sv->req_alloc); if (posted > sv->cap) posted = sv->cap;
        const uint32_t resumed = ld2(&sv->n_resumed);
        if (resumed < posted) {
          a = 4;
          // Only as many slots wait as there are requests out: the rest leave, so that their share of the CU (LDS, wave slots) is
          // free for the next batch's kernels while the last few windows of this one finish.
          if (!waiting) { const uint32_t wv = dev_atomic_add(&sv->n_waiting, 1u); if (wv >= posted - resumed) { dev_atomic_add(&sv->n_waiting, 0xFFFFFFFFu); a = 3; } else { a = 5; just_registered = true; } }
          if (a != 3 && (ld2(&sv->alive) == 0u || idle > 20000)) {      // no service workgroup runs: the general build after all
            for (uint32_t i = 0; i < posted; ++i)
              if (ld2(&sv->req[i].state) == SV_POSTED && dev_atomic_cas32(&sv->req[i].state, SV_POSTED, SV_STOLEN) == SV_POSTED) {
                (void)ld_acq(&sv->req[i].state);
                a = 2; g = (int)i; dev_atomic_add(&sv->n_resumed, 1u); dev_atomic_add(&sv->n_stolen, 1u); break;
              }
          }
        }
      }
      if (sv && ((waiting && a != 4) || (!waiting && a == 2 && just_registered))) dev_atomic_add(&sv->n_waiting, 0xFFFFFFFFu);   // (got work, or leaves)
      S->act = a; S->act_arg = g;
    }
    int act = wg_bcast(&S->act); const int arg = wg_bcast(&S->act_arg);
    if (act == 5) { waiting = true; act = 4; } else if (act != 4) waiting = false;
    if (act == 3) break;
    if (act == 4) {
#ifdef LANCET_WAVE_EMU
      if (waiting) sv->n_waiting -= 1u;                      // (the emulated slot comes back after the service has run)
      return 1;
#else
      dev_sleep(); ++idle; continue;
#endif
    }
    idle = 0;
    WG_LANE0 { if (sv) dev_atomic_add(&sv->beat, 1u); }
    int w = arg, rq = -1;
    if (act == 0) {
      if (OUT->win_list) w = (int)OUT->win_list[w];
      if (OUT->skip && OUT->skip[w]) continue;      // a coverage pile-up the host sent straight to the re-run tier (engine.hip)
    } else { rq = arg; w = (int)sv->req[rq].w; }
#ifndef LANCET_WAVE_EMU
    if (threadIdx.x == 0 && !OUT->phase) S->phase_cur = -1;            // (nobody asked for phase times: PHASE does nothing)
    if (threadIdx.x == 0 && OUT->phase) {
      for (int i = 0; i < 16; ++i) S->phase_acc[i] = rq >= 0 ? OUT->phase[(size_t)w * 16 + i] : 0ull;
      S->phase_cur = 0; S->t_last = wall_clock64();
#ifdef LANCET_PROF_TIMELINE   /* profiling builds only (tools/timeline.py): when the window was taken, resumed and finished, in the slots of the general build's phases */
      if (rq < 0) { S->phase_acc[2] = S->t_last; S->phase_acc[4] = 0; S->phase_acc[5] = 0; S->phase_acc[6] = 0; S->phase_acc[7] = (unsigned long long)slot; } else { S->phase_acc[5] = S->t_last; S->phase_acc[6] += 1; }
#endif
    }
#endif
    process_window(c, w, rq);
    if (wg_bcast(&S->status) == LANCET_W_SUSPENDED) continue;
    PHASE(c, 0);
#ifdef LANCET_PROF_TIMELINE
    WG_LANE0 { S->phase_acc[3] = wall_clock64(); }
#endif
    WG_LANE0 {
      lancet_window_stats &st = OUT->stats[w];
      st.status = S->status; st.final_k = S->final_k; st.n_builds = S->n_builds; st.n_variants = S->emit_seq;
      st.n_kmers = S->n_kmers; st.max_nodes = S->max_nodes; st.sum_nodes = S->sum_nodes;
      if (OUT->phase) for (int i = 0; i < 16; ++i) OUT->phase[(size_t)w * 16 + i] = S->phase_acc[i];
      if (C->evt_cap) { OUT->evt_len[w] = S->evt_len; for (uint32_t i = 0; i < S->evt_len; ++i) OUT->evt_out[(size_t)w * C->evt_cap + i] = LC_CTX(c).W->evt[i]; }
    }
    WG_SYNC();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// prep: Graph_t::trim (reference src/Graph.cc:355-384) + 2-bit packing ; one read per lane
// ---------------------------------------------------------------------------------------------------------
DEV bool is_dna(char b) { return b == 'A' || b == 'a' || b == 'C' || b == 'c' || b == 'G' || b == 'g' || b == 'T' || b == 't'; }
DEV int base_code(char b) { switch (b) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; } return 4; }
DEV void prep_read(const lancet_params *P, const char *seq, const char *qual, uint32_t off, int len, uint8_t label, uint8_t strand, uint8_t mate,
                   uint8_t mapped, uint32_t *rinfo, uint32_t *bases, uint32_t bw, uint32_t *good, uint32_t gw) {
  int trim5 = 0, trim3 = 0; bool junk = false;
  while (trim5 < len && (!is_dna(seq[off + trim5]) || (qual[off + trim5] < P->min_qual_trim))) ++trim5;
  if (trim5 < len) {
    while (trim3 < len && (!is_dna(seq[off + len - 1 - trim3]) || (qual[off + len - 1 - trim3] < P->min_qual_trim))) ++trim3;
    for (int i = trim5; i < len - trim3; ++i) if (!is_dna(seq[off + i])) { junk = true; break; }
  } else junk = true;
  int tlen = junk ? 0 : len - trim5 - trim3;
  if (tlen > 0xFFFF) tlen = 0xFFFF;
  *rinfo = (uint32_t)tlen | ((label == LANCET_NML ? 1u : 0u) << 16) | ((strand == LANCET_REV ? 1u : 0u) << 17) | ((uint32_t)(mate & 3) << 18) | ((mapped ? 1u : 0u) << 20);
  for (int wv = 0; wv < (tlen + 15) / 16; ++wv) {
    uint32_t v = 0;
    for (int j = 0; j < 16 && wv * 16 + j < tlen; ++j) v |= (uint32_t)(base_code(seq[off + trim5 + wv * 16 + j]) & 3) << (2 * j);
    bases[bw + wv] = v;
  }
  for (int wv = 0; wv < (tlen + 31) / 32; ++wv) {
    uint32_t v = 0;
    for (int j = 0; j < 32 && wv * 32 + j < tlen; ++j) if (qual[off + trim5 + wv * 32 + j] >= P->min_qual_call) v |= 1u << j;
    good[gw + wv] = v;
  }
}
