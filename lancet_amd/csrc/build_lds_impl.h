// build_lds_impl.h -- body of the LDS build kernel, included by build_lds.h once per size configuration (BL_NS, BL_WG, BL_BASES,
// BL_RMAX, BL_SLOTS, BL_TCAP, BL_BIG, BL_OFFBITS, BL_FLAGCAP, BL_LDS_LIMIT).  No include guard on purpose.
namespace BL_NS {

// An LDS offset (a base of the packed reads) has BL_OFFBITS bits; the k-mer table keeps it under a fingerprint of the remaining bits.
#if BL_OFFBITS > 16
typedef uint32_t bl_occ_t;          /* an occurrence index (the window may hold more than 65 535 k-mer starts) */
typedef uint32_t bl_off_t;          /* an LDS offset kept in HBM scratch */
#else
typedef uint16_t bl_occ_t;
typedef uint16_t bl_off_t;
#endif
static constexpr uint32_t BL_OFFMASK = (1u << BL_OFFBITS) - 1u;
// The per-occurrence word in HBM scratch: table slot, then node id, with the orientation and the "overlapping mate" mark on top.
#if BL_WIDE
typedef uint32_t bl_on_t;
static constexpr uint32_t ON_ID = 0xFFFFFu, ON_OVL = 0x40000000u, ON_ORI = 0x80000000u, ON_ORISH = 31u;
#else
typedef uint16_t bl_on_t;
static constexpr uint32_t ON_ID = 0x1FFFu, ON_OVL = 0x4000u, ON_ORI = 0x8000u, ON_ORISH = 15u;
#endif
static_assert(BL_NCAP <= ON_ID + 1u && BL_SLOTS <= ON_ID + 1u && BL_NCAP < 65536u, "node ids / slots in a per-occurrence word");
static constexpr uint32_t LDS_BASES = BL_BASES, LDS_READS = BL_RMAX;       /* this configuration's limits, for the host */
static_assert(BL_BASES < (1u << BL_OFFBITS) && BL_BASES / 16 + 4 < 65536, "offsets of the packed reads");

struct BlShared {
  uint32_t bases[BL_BASES / 16 + 4];
  uint32_t goodm[BL_BASES / 32 + 4];
  uint16_t rdo[BL_RMAX + 4];        /* first 16-base word of read r                                             */
  uint16_t gwo[BL_RMAX + 4];        /* first quality-mask word of read r                                        */
  bl_occ_t cbase[BL_RMAX + 4];      /* first chunk (eight k-mer starts, bl_chunk) of read r; [R] = all chunks   */
  uint16_t c2r[BL_BASES / 128 + 2]; /* read that holds chunk 16 * j                                             */
  uint32_t rinfo[BL_RMAX + 4];
  uint8_t pidx[BL_RMAX + 4];        /* mate-pair signature bit of the read (0xFF none)                          */
  uint8_t prole[BL_RMAX + 4];       /* 1 = earlier mate of a pair, 2 = the later one                            */
#if !BL_WIDE
  uint16_t idoff[BL_NCAP];          /* node id -> LDS offset of its first occurrence (BL_WIDE: in the HBM scratch, BlScratch::idoff) */
#endif
  alignas(8) uint16_t cidx[BL_NCAP];/* node id -> tracked index, later survivor index (0xFFFF none)             */
  uint16_t t2c[BL_TCAP];            /* tracked index -> candidate index (0xFFFF none)                           */
  alignas(16) uint32_t big[BL_BIG / 4];   /* (64-bit LDS atomics on it: must be 8-byte aligned) */
  uint32_t wsum[BL_WG / 64 + 1];
  uint32_t scan_total;
  int w, R, reflen, K, hasN, mapped;
  uint32_t O, N, T, ncand, nsurv, nbw, ngw;
  uint32_t totalreadbp, n_kmers;
  int repE, repM;
  int why;
  uint32_t npairs, flagged, edges_total, refn;
  uint32_t g0, g1;                  /* group bounds of the current pass                                          */
  uint32_t hint;                    /* the graph will very likely have a cycle at this k (scheduling hint, see the insert pass) */
  uint32_t ndup;                    /* occurrences noted in BlScratch::dupo                                        */
  unsigned long long t_last, ph_acc[16]; int ph_cur;   /* profiling: wall-clock ticks per phase (lane 0)                 */
};

/* per-workgroup scratch in HBM (streamed, never shared between windows in flight) */
struct BlScratch {
  LC_GLOBAL bl_on_t *occn;          /* [BL_BASES] slot, then node id | ON_ORI, per k-mer start offset            */
  LC_GLOBAL uint32_t *idoff;        /* [BL_NCAP] BL_WIDE: node id -> LDS offset of its first occurrence          */
  LC_GLOBAL unsigned long long *tcc;/* [BL_TCAP] counted occurrences Tf Tr Nf Nr (4 x 16 bit)                    */
  LC_GLOBAL uint32_t *tfl;          /* [BL_TCAP] NF_TUMOR | NF_NORMAL                                            */
  LC_GLOBAL uint32_t *c_id;         /* [PB_CCAP] node of candidate ci                                            */
  LC_GLOBAL uint32_t *c_ti;         /* [PB_CCAP] its tracked index                                               */
  LC_GLOBAL uint32_t *c_minqv;      /* [PB_CCAP]                                                                 */
  LC_GLOBAL uint32_t *s_ci;         /* [PB_SCAP] candidate of survivor si                                        */
  LC_GLOBAL uint32_t *s_edges;      /* [PB_SCAP * 9] resolved edges + count                                      */
  LC_GLOBAL bl_off_t *dupo;         /* [BL_DUPCAP] occurrences that met their k-mer in the other orientation, or twice in one read */
  LC_GLOBAL uint32_t *ord;          /* [4 * PB_CMAX] (spare) */
  LC_GLOBAL uint32_t *pq;           /* [BL_PQCAP] per-position counts: the occurrences of the candidates beyond the first LDS group (read | position << 10 | candidate << 20 | reversed << 31) */
};
static constexpr uint32_t BL_PQCAP = 16384u;
static constexpr uint32_t SCRATCH_BYTES = (((uint32_t)sizeof(bl_on_t) * BL_BASES + 64u + (BL_WIDE ? 4u * BL_NCAP + 64u : 64u) + 8u * BL_TCAP + 4u * BL_TCAP + 12u * PB_CCAP + 4u * PB_SCAP + 36u * PB_SCAP + (uint32_t)sizeof(bl_off_t) * BL_DUPCAP + 16u * PB_CMAX + 4u * 16384u + 768u) + 255u) & ~255u;   /* (a multiple of 256: the strips' 16-byte loads stay aligned in every slot) */
static constexpr int WG = BL_WG;
DEV void bl_scratch_carve(BlScratch *s, LC_GLOBAL uint8_t *base) {
  size_t o = 0;
  auto take = [&](size_t bytes) { LC_GLOBAL uint8_t *p = base + o; o = (o + bytes + 63) & ~(size_t)63; return p; };
  s->occn = (LC_GLOBAL bl_on_t *)take(sizeof(bl_on_t) * BL_BASES + 64u);
  s->idoff = (LC_GLOBAL uint32_t *)take(BL_WIDE ? 4u * BL_NCAP : 4u);
  s->tcc = (LC_GLOBAL unsigned long long *)take(8u * BL_TCAP);
  s->tfl = (LC_GLOBAL uint32_t *)take(4u * BL_TCAP);
  s->c_id = (LC_GLOBAL uint32_t *)take(4u * PB_CCAP);
  s->c_ti = (LC_GLOBAL uint32_t *)take(4u * PB_CCAP);
  s->c_minqv = (LC_GLOBAL uint32_t *)take(4u * PB_CCAP);
  s->s_ci = (LC_GLOBAL uint32_t *)take(4u * PB_SCAP);
  s->s_edges = (LC_GLOBAL uint32_t *)take(36u * PB_SCAP);
  s->dupo = (LC_GLOBAL bl_off_t *)take(sizeof(bl_off_t) * BL_DUPCAP);
  s->ord = (LC_GLOBAL uint32_t *)take(16u * PB_CMAX);
  s->pq = (LC_GLOBAL uint32_t *)take(4u * BL_PQCAP);
}

static_assert(sizeof(BlShared) <= BL_LDS_LIMIT, "LDS of the build kernel: 2 x 80 KB (512 lanes) or 1 x 160 KB (1024 lanes) per CU");
typedef LC_LDS BlShared BL_S;
#ifndef LANCET_WAVE_EMU
static __shared__ BlShared bl_shared;
template <class P> DEV unsigned long long dev_atomic_add64(P p, unsigned long long v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class P> DEV unsigned long long dev_atomic_or64(P p, unsigned long long v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#else
DEV unsigned long long dev_atomic_add64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
DEV unsigned long long dev_atomic_or64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o | v; return o; }
#endif

#define bl_bcast(p) wg_bcastu(p)

// exclusive prefix sum of an LDS array in place, whole workgroup; the total lands in S.scan_total
template <class T> DEV void bl_scan_t(LC_LDS T *a, int n, BL_S &S) {
  WG_SYNC();
#ifndef LANCET_WAVE_EMU
  const int t = (int)threadIdx.x, chunk = (n + BL_WG - 1) / BL_WG;
  int lo = t * chunk, hi = lo + chunk; if (lo > n) lo = n; if (hi > n) hi = n;
  uint32_t s = 0;
  for (int i = lo; i < hi; ++i) s += a[i];
  uint32_t inc = s; const int lane = t & 63;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += x; }
  if (lane == 63) S.wsum[t >> 6] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int i = 0; i < (t >> 6); ++i) woff += S.wsum[i];
  uint32_t run = woff + inc - s;
  for (int i = lo; i < hi; ++i) { const uint32_t x = a[i]; a[i] = (T)run; run += x; }
  if (t == BL_WG - 1) S.scan_total = run;
  __syncthreads();
#else
  uint32_t run = 0;
  for (int i = 0; i < n; ++i) { const uint32_t x = a[i]; a[i] = (T)run; run += x; }
  S.scan_total = run;
  lc_emu_syncs += 2;
#endif
}
DEV void bl_scan32(LC_LDS uint32_t *a, int n, BL_S &S) { bl_scan_t<uint32_t>(a, n, S); }
// the same scan over run sizes, which also sets bit `start` of `heads` for every run that is not empty (the table-order stages: a run's
// first position, known here for free, is what the pass after the fill would otherwise work out through four dependent look-ups)
DEV void bl_scan32_heads(LC_LDS uint32_t *a, int n, BL_S &S, LC_LDS uint32_t *heads) {
  WG_SYNC();
#ifndef LANCET_WAVE_EMU
  const int t = (int)threadIdx.x, chunk = (n + BL_WG - 1) / BL_WG;
  int lo = t * chunk, hi = lo + chunk; if (lo > n) lo = n; if (hi > n) hi = n;
  uint32_t s = 0;
  for (int i = lo; i < hi; ++i) s += a[i];
  uint32_t inc = s; const int lane = t & 63;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += x; }
  if (lane == 63) S.wsum[t >> 6] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int i = 0; i < (t >> 6); ++i) woff += S.wsum[i];
  uint32_t run = woff + inc - s;
  for (int i = lo; i < hi; ++i) { const uint32_t x = a[i]; a[i] = run; if (x) dev_atomic_or(&heads[run >> 5], 1u << (run & 31u)); run += x; }
  if (t == BL_WG - 1) S.scan_total = run;
  __syncthreads();
#else
  uint32_t run = 0;
  for (int i = 0; i < n; ++i) { const uint32_t x = a[i]; a[i] = run; if (x) heads[run >> 5] |= 1u << (run & 31u); run += x; }
  S.scan_total = run;
  lc_emu_syncs += 2;
#endif
}
// 16-bit halves of LDS words under 32-bit atomics (the table order of large tables: twice the elements in the same LDS)
DEV void bl_min16(LC_LDS uint16_t *a, uint32_t i, uint32_t v) {
  LC_LDS uint32_t *w = (LC_LDS uint32_t *)a + (i >> 1); const uint32_t sh = (i & 1u) * 16u;
  uint32_t old = *(volatile LC_LDS uint32_t *)w;
  while (((old >> sh) & 0xFFFFu) > v) {
    const uint32_t want = (old & ~(0xFFFFu << sh)) | (v << sh);
    const uint32_t got = dev_atomic_cas32(w, old, want);
    if (got == old) break;
    old = got;
  }
}
DEV uint32_t bl_add16(LC_LDS uint16_t *a, uint32_t i, uint32_t inc) {        // returns the half before the addition (no carry: the callers' sums stay below 65 536)
  LC_LDS uint32_t *w = (LC_LDS uint32_t *)a + (i >> 1); const uint32_t sh = (i & 1u) * 16u;
  return (dev_atomic_add(w, inc << sh) >> sh) & 0xFFFFu;
}

// the k-mer that starts at LDS offset `boff`: base j at bits 2j (k <= 31)
DEV unsigned long long bl_kmer(const LC_LDS uint32_t *bases, uint32_t boff, unsigned long long kmask) {
  const uint32_t w = boff >> 4, sh = (boff & 15u) * 2u;
  const unsigned long long lo = (unsigned long long)bases[w] | ((unsigned long long)bases[w + 1] << 32);
  const unsigned long long v = (lo >> sh) | (((unsigned long long)bases[w + 2] << 1) << (63u - sh));     // (no branch on sh == 0: three reads always, in flight together)
  return v & kmask;
}
// canonical form as kernels.h holds it (first base most significant; CanonicalMer_t::set, reference src/Mer.hh:57-71: tie -> R)
DEV unsigned long long bl_canon(unsigned long long v, int K, unsigned long long kmask, bool *isF) {
  const unsigned long long rc = (~v) & kmask;
  unsigned long long fw = dev_brev64(v);
  fw = ((fw >> 1) & 0x5555555555555555ULL) | ((fw & 0x5555555555555555ULL) << 1);
  fw >>= (64 - 2 * K);
  *isF = fw < rc;
  return *isF ? fw : rc;
}
DEV unsigned long long bl_canon2(unsigned long long v, int K, unsigned long long kmask, bool *isF, unsigned long long *fwd) {
  const unsigned long long rc = (~v) & kmask;
  unsigned long long fw = dev_brev64(v);
  fw = ((fw >> 1) & 0x5555555555555555ULL) | ((fw & 0x5555555555555555ULL) << 1);
  fw >>= (64 - 2 * K);
  *fwd = fw;
  *isF = fw < rc;
  return *isF ? fw : rc;
}
template <class SS, class XX> DEV uint32_t bl_idoff(SS &S, XX &X, uint32_t n) {     // LDS offset of node n's first occurrence
#if BL_WIDE
  (void)S; return X.idoff[n];
#else
  (void)X; return (uint32_t)S.idoff[n];
#endif
}
// ---- k-mers of more than one 64-bit word (BL_KW > 1: the 1024-lane configuration, k <= 127).  `NW` = words this k needs.
struct BlKm { unsigned long long w[BL_KW]; };
DEV unsigned long long bl_rev2(unsigned long long x) {             // the 32 two-bit groups of a word in reverse order
  x = dev_brev64(x);
  return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
}
// the k-mer that starts at LDS offset `boff`, as it lies there: base j at bits 2j of the word string (low word first)
DEV void bl_kmer_x(const LC_LDS uint32_t *bases, uint32_t boff, int NW, int K, BlKm &o) {
  const uint32_t w0 = boff >> 4, sh = (boff & 15u) * 2u;
  const int tb = 2 * K - 64 * (NW - 1);
  for (int i = 0; i < BL_KW; ++i) {
    unsigned long long v = 0;
    if (i < NW) {
      const uint32_t w = w0 + 2u * (uint32_t)i;
      const unsigned long long lo = (unsigned long long)bases[w] | ((unsigned long long)bases[w + 1] << 32);
      v = sh ? ((lo >> sh) | ((unsigned long long)bases[w + 2] << (64u - sh))) : lo;
      if (i == NW - 1 && tb < 64) v &= (1ULL << tb) - 1ULL;
    }
    o.w[i] = v;
  }
}
// canonical form as kernels.h holds it (first base most significant, right-aligned in NW words), and `alt`: what an EARLIER occurrence
// of the same node looks like in LDS when it lies there in the other orientation (see the insert pass)
template <int NW> DEV void bl_canon_n(const BlKm &v, int K, BlKm &ck, BlKm &alt, bool *isF) {
  unsigned long long R[NW + 1], fw[NW], rc[NW];
  for (int i = 0; i < NW; ++i) R[i] = bl_rev2(v.w[NW - 1 - i]);
  R[NW] = 0;
  const int s = 64 * NW - 2 * K, tb = 64 - s;                       // 0 <= s < 64 ; bits of the top word in use
  for (int i = 0; i < NW; ++i) fw[i] = s ? ((R[i] >> s) | (R[i + 1] << (64 - s))) : R[i];
  for (int i = 0; i < NW; ++i) rc[i] = ~v.w[i];
  if (tb < 64) rc[NW - 1] &= (1ULL << tb) - 1ULL;
  bool less = false, decided = false;
  for (int i = NW - 1; i >= 0; --i) if (!decided && fw[i] != rc[i]) { less = fw[i] < rc[i]; decided = true; }
  *isF = less;
  for (int i = 0; i < BL_KW; ++i) { ck.w[i] = i < NW ? (less ? fw[i] : rc[i]) : 0ULL; alt.w[i] = i < NW ? ~fw[i] : 0ULL; }
  if (tb < 64) alt.w[NW - 1] &= (1ULL << tb) - 1ULL;
}
DEV void bl_canon_x(const BlKm &v, int NW, int K, BlKm &ck, BlKm &alt, bool *isF) {
#if BL_KW >= 4
  if (NW == 4) { bl_canon_n<4>(v, K, ck, alt, isF); return; }
  if (NW == 3) { bl_canon_n<3>(v, K, ck, alt, isF); return; }
#endif
#if BL_KW >= 2
  if (NW == 2) { bl_canon_n<2>(v, K, ck, alt, isF); return; }
#endif
  bl_canon_n<1>(v, K, ck, alt, isF);
}
// libstdc++ std::hash of the canonical k-mer's string (kernels.h std_hash_bytes), characters taken off the top of the key eight at a
// time: the key is shifted to the top of its NW words once and then moves left by 16 bits per block -- constant register indices
// only (a character picked by a run-time word index put the key in scratch memory: one dependent load per character).
DEV int bl_acgt(int code) { return (int)((0x54474341u >> (8 * code)) & 0xFFu); }      // 'A' 'C' 'G' 'T' without a look-up in constant memory
DEV unsigned long long bl_std_hash_km(const BlKm &ck, int NW, int K) {
  unsigned long long w[BL_KW];
  {   // left-align: shift the NW-word value left by 64 * NW - 2 * K bits (< 64), then move it to the top of the BL_KW words
    const int s = 64 * NW - 2 * K;
    unsigned long long t[BL_KW];
    for (int i = 0; i < BL_KW; ++i) t[i] = i < NW ? ((ck.w[i] << s) | ((i > 0 && s) ? (ck.w[i - 1] >> (64 - s)) : 0ULL)) : 0ULL;
    for (int i = 0; i < BL_KW; ++i) { unsigned long long x = 0; for (int j = 0; j < BL_KW; ++j) if (j + (BL_KW - NW) == i) x = t[j]; w[i] = x; }
  }
  const unsigned long long mul = (0xc6a4a793ULL << 32) + 0x5bd1e995ULL;
  unsigned long long hash = 0xc70f6907ULL ^ ((unsigned long long)K * mul);
  auto take = [&](int nch) -> unsigned long long {                    // the next nch (<= 8) characters as a little-endian block of bytes
    const uint32_t top = (uint32_t)(w[BL_KW - 1] >> 48);
    unsigned long long d = 0;
    for (int j = 0; j < 8; ++j) if (j < nch) d |= (unsigned long long)bl_acgt((int)((top >> (14 - 2 * j)) & 3u)) << (8 * j);
    for (int i = BL_KW - 1; i > 0; --i) w[i] = (w[i] << 16) | (w[i - 1] >> 48);
    w[0] <<= 16;
    return d;
  };
  const int nblk = K & ~7;
  for (int i = 0; i < nblk; i += 8) { unsigned long long d = take(8); d *= mul; d ^= d >> 47; d *= mul; hash ^= d; hash *= mul; }
  if (K & 7) { const unsigned long long d = take(K & 7); hash ^= d; hash *= mul; }
  hash ^= hash >> 47; hash *= mul; hash ^= hash >> 47;
  return hash;
}
DEV bool bl_km_eq(const BlKm &a, const BlKm &b) { bool e = true; for (int i = 0; i < BL_KW; ++i) e = e && a.w[i] == b.w[i]; return e; }
DEV int bl_base(const LC_LDS uint32_t *bases, uint32_t boff) { return (int)((bases[boff >> 4] >> ((boff & 15u) * 2u)) & 3u); }
// quality-mask bits [a, b) of a read whose mask starts at word gw: all set?
DEV bool bl_all_good(const LC_LDS uint32_t *goodm, uint32_t gw, int a, int b) {
  for (int i = a; i < b;) {
    const int w = i >> 5, lo = i & 31;
    int take = 32 - lo; if (take > b - i) take = b - i;
    uint32_t m = goodm[gw + w] >> lo;
    const uint32_t full = take < 32 ? ((1u << take) - 1u) : 0xFFFFFFFFu;
    if ((m & full) != full) return false;
    i += take;
  }
  return true;
}

// a growth stage of the table order, first pass: bucket of every element under B buckets, smallest position per bucket (stamped: see the stages);
// U = elements per lane (n <= U * BL_WG <= 4096), all of a lane's look-ups in flight together
template <int U> DEV void bl_stage_buckets(const LC_LDS uint16_t *Q, LC_GLOBAL const unsigned long long *nhash, uint32_t n, uint32_t B, uint32_t st,
                                           LC_LDS uint16_t *bkt, LC_LDS uint32_t *first) {
  WG_FOR(_t, BL_WG) {
    uint32_t qi[U]; unsigned long long hv[U];
    BL_UNROLL for (int u = 0; u < U; ++u) { const uint32_t i = (uint32_t)_t + (uint32_t)u * BL_WG; qi[u] = Q[i < n ? i : 0u]; }
    BL_UNROLL for (int u = 0; u < U; ++u) hv[u] = nhash[qi[u]];
    BL_UNROLL for (int u = 0; u < U; ++u) {
      const uint32_t i = (uint32_t)_t + (uint32_t)u * BL_WG;
      if (i < n) { const uint32_t b = ht_mod(hv[u], B); bkt[i] = (uint16_t)b; dev_atomic_min(&first[b], st | i); }
    }
  }
}
// 64 bits of a read's quality mask (first word gw) from bit s on (bits past the read's last base belong to whatever follows: the callers
// only look at bits inside the read)
DEV unsigned long long bl_good64(const LC_LDS uint32_t *goodm, uint32_t gw, int s) {
  const uint32_t w = gw + ((uint32_t)s >> 5), sh = (uint32_t)s & 31u;
  const unsigned long long lo = (unsigned long long)goodm[w] | ((unsigned long long)goodm[w + 1] << 32);
  return sh ? ((lo >> sh) | ((unsigned long long)goodm[w + 2] << (64u - sh))) : lo;
}
// ---- The walk over the window's k-mer occurrences, eight at a time.
// A "chunk" is eight consecutive k-mer starts: the bases [8c, 8c + 8) of the LDS copy of the reads.  A read starts on a 16-base word, so
// a chunk lies inside ONE read: the read (w2r + rdo), its trimmed length and the number of its k-mers are looked up once per eight
// occurrences instead of once per occurrence, the chunk's per-occurrence words in HBM scratch are one 16-byte load (BL_WIDE: two), and the
// bases / quality bits of all eight k-mers come out of the same three LDS words.  (Until round 5 every lane took single occurrences,
// lane-strided over a dense occurrence index: ~25 of the ~110-150 VALU instructions per occurrence and pass were that look-up -- and the
// occurrence passes are VALU-bound, profiles/r5_sq_counters.txt.)  Chunks past a read's last k-mer are skipped; no pass depends on the
// order the occurrences are visited in (first occurrences and edge stamps by atomicMin on offsets, counts by atomic adds, lists sorted).
struct BlChunk {
  int r, p0, nv;                    /* read, k-mer start of the chunk's first occurrence in it, occurrences in the chunk (1..8) */
  uint32_t boff0;                   /* LDS offset (base index) of that first occurrence = index of its per-occurrence word       */
  int tlen;                         /* the read's trimmed length (the window reference: its length)                              */
};
// chunk q of the window (chunks are numbered densely: read r owns ceil(k-mers / 8) of them from cbase[r] on)
DEV void bl_chunk(BL_S &S, int q, int nr, int reflen, int K, BlChunk &ch) {
  uint32_t r = S.c2r[(uint32_t)q >> 4];
  while ((uint32_t)q >= (uint32_t)S.cbase[r + 1]) ++r;
  const int tlen = (int)r < nr ? (int)RI_TLEN(S.rinfo[r]) : reflen;
  const int p0 = 8 * (q - (int)S.cbase[r]);
  ch.r = (int)r; ch.p0 = p0; ch.tlen = tlen; ch.boff0 = 16u * (uint32_t)S.rdo[r] + (uint32_t)p0;
  const int left = tlen - K + 1 - p0;                                 // (>= 1: the read has a k-mer start in this chunk)
  ch.nv = left < 8 ? left : 8;
}
// the eight per-occurrence words of a chunk (HBM scratch), as loaded
#if BL_WIDE
struct BlOccW { lc_u4 a, b; };
DEV BlOccW bl_occw_load(LC_GLOBAL const bl_on_t *occn, uint32_t boff0) { BlOccW w; w.a = ldg4((LC_GLOBAL const uint32_t *)(occn + boff0)); w.b = ldg4((LC_GLOBAL const uint32_t *)(occn + boff0 + 4u)); return w; }
DEV void bl_occw_store(LC_GLOBAL bl_on_t *occn, uint32_t boff0, const BlOccW &w) { stg4((LC_GLOBAL uint32_t *)(occn + boff0), w.a); stg4((LC_GLOBAL uint32_t *)(occn + boff0 + 4u), w.b); }
DEV uint32_t bl_occw_get(const BlOccW &w, int j) {
  const uint32_t lo = (j & 2) ? ((j & 1) ? w.a.w : w.a.z) : ((j & 1) ? w.a.y : w.a.x), hi = (j & 2) ? ((j & 1) ? w.b.w : w.b.z) : ((j & 1) ? w.b.y : w.b.x);
  return (j & 4) ? hi : lo;
}
DEV void bl_occw_set(BlOccW &w, int j, uint32_t v) {
  if (j == 0) w.a.x = v; else if (j == 1) w.a.y = v; else if (j == 2) w.a.z = v; else if (j == 3) w.a.w = v;
  else if (j == 4) w.b.x = v; else if (j == 5) w.b.y = v; else if (j == 6) w.b.z = v; else w.b.w = v;
}
#else
struct BlOccW { lc_u4 a; };
DEV BlOccW bl_occw_load(LC_GLOBAL const bl_on_t *occn, uint32_t boff0) { BlOccW w; w.a = ldg4((LC_GLOBAL const uint32_t *)(occn + boff0)); return w; }
DEV void bl_occw_store(LC_GLOBAL bl_on_t *occn, uint32_t boff0, const BlOccW &w) { stg4((LC_GLOBAL uint32_t *)(occn + boff0), w.a); }
DEV uint32_t bl_occw_get(const BlOccW &w, int j) {
  const uint32_t x = (j & 4) ? ((j & 2) ? w.a.w : w.a.z) : ((j & 2) ? w.a.y : w.a.x);
  return (j & 1) ? (x >> 16) : (x & 0xFFFFu);
}
DEV void bl_occw_set(BlOccW &w, int j, uint32_t v) {                 // (v < 65 536)
  const uint32_t sh = (uint32_t)(j & 1) * 16u, m = ~(0xFFFFu << sh), x = v << sh;
  if ((j >> 1) == 0) w.a.x = (w.a.x & m) | x; else if ((j >> 1) == 1) w.a.y = (w.a.y & m) | x; else if ((j >> 1) == 2) w.a.z = (w.a.z & m) | x; else w.a.w = (w.a.w & m) | x;
}
#endif
// every chunk of the window that holds k-mer starts: body(chunk, its eight per-occurrence words).  The words of the lane's NEXT chunk are
// loaded before this one is worked off (unconditionally, index clamped: a conditional load would wait for every earlier store).
template <class F> DEV void bl_for_chunk(BL_S &S, LC_GLOBAL const bl_on_t *occn, F body) {
  const int nr = S.R - 1, NC = (int)S.cbase[nr + 1], reflen = S.reflen, K = S.K;
  WG_FOR(_t, BL_WG) {
    if (_t < NC) {
      BlChunk ch; bl_chunk(S, _t, nr, reflen, K, ch);
      BlOccW cur = bl_occw_load(occn, ch.boff0);
      for (int q = _t; q < NC; q += BL_WG) {
        BlChunk nx = ch;
        if (q + BL_WG < NC) bl_chunk(S, q + BL_WG, nr, reflen, K, nx);        // (LDS look-ups only; the load below stays unconditional)
        const BlOccW nxt = bl_occw_load(occn, nx.boff0);
        body(ch, cur);
        cur = nxt; ch = nx;
      }
    }
  }
}
// the same walk, one call of `body(r, p, boff, word)` per occurrence
template <class F> DEV void bl_for_occ(BL_S &S, LC_GLOBAL const bl_on_t *occn, F body) {
  bl_for_chunk(S, occn, [&](const BlChunk &ch, const BlOccW &w) {
    for (int j = 0; j < ch.nv; ++j) body(ch.r, ch.p0 + j, ch.boff0 + (uint32_t)j, bl_occw_get(w, j));
  });
}
// the chunk's bases as a bit string: base i of the chunk (i = 0 .. 8 + K - 2 <= 37 for k <= 31) at bits 2i of (lo, hi)
DEV void bl_chunk_bases(const LC_LDS uint32_t *bases, int c, unsigned long long &lo, unsigned long long &hi) {
  const uint32_t wd = (uint32_t)c >> 1;
  lo = (unsigned long long)bases[wd] | ((unsigned long long)bases[wd + 1] << 32); hi = (unsigned long long)bases[wd + 2];
  if (c & 1) { lo = (lo >> 16) | (hi << 48); hi >>= 16; }
}


// ---------------------------------------------------------------------------------------------------------
// markRefEnds (reference src/Graph.cc:2028-2228) and the first Graph_t::compress (:2486-2732) of a single-component first graph,
// in LDS by the whole workgroup -- what the window kernel otherwise does on one wave over HBM (kernels.h mark_ref_scan /
// mark_ref_ends / compress_prepare / compress_rank, whose results this reproduces field for field).  Nodes are addressed by their
// position in the table order after cleanDead (0 .. nsurv-1; the source is nsurv, the sink nsurv + 1); an edge is 16 bits:
// neighbour position [9:0], direction [11:10].  Returns with PreCmp::done = 1 and
//   * the records of the live nodes rewritten in the hand-off area (unitig heads: coverage by the reference's float recurrence in merge
//     order, minima, flags, new edge list, new descriptor deque in CSEQ; the others: edges redirected to the heads; the two special nodes),
//   * CLIVE = the table order after the two insertions and cleanDead,
// or with done = 0 and nothing touched (no unambiguous source / sink, a ring, an irregular link, more than 8 edges on a node, a
// table that would be rehashed by the insertions, too many heads / descriptors): the window kernel then does all of it itself.
// ---------------------------------------------------------------------------------------------------------
// profiling builds only (-DLANCET_PROF_A): the steps of bl_compress_first accounted in the build kernel's phase slots 1..13
#ifdef LANCET_PROF_A
#define BLPA(S, id) BLP(S, id)
#else
#define BLPA(S, id) ((void)0)
#endif
// -DLANCET_PROF_ORDER: the steps of a growth stage of the table order in the phase slots 1..6 (1 buckets + minima, 2 run sizes, 3 scan, 4 fill,
// 5 run order + move, 6 the rest of the tail up to the components), the components in 7, what follows in 15
#ifdef LANCET_PROF_ORDER
#define BLPO(S, id) BLP(S, id)
#else
#define BLPO(S, id) ((void)0)
#endif
#define BLC_TO(e) ((uint32_t)(e) & 0x3FFu)
#define BLC_DIR(e) (((uint32_t)(e) >> 10) & 3u)
#define BLC_MAKE(to, dir) ((uint16_t)((to) | ((dir) << 10)))
DEVNI void bl_compress_first(LC_GLOBAL const lancet_params *P, LC_GLOBAL const EngineCaps *C, BL_S &S, BlScratch &X, LC_GLOBAL uint8_t *area, const int K_,
                             const uint32_t N_, const uint32_t nsurv_, const uint32_t ncand_, const int reflen_, const uint32_t ht_bc_, const uint32_t refmask_) {
  P = lc_sgpr(P); C = lc_sgpr(C); area = lc_sgpr(area);                        // (uniform arguments: scalar registers, wave.h lc_sgpr)
  const uint32_t N = lc_sgpr(N_), nsurv = lc_sgpr(nsurv_), ncand = lc_sgpr(ncand_), ht_bc = lc_sgpr(ht_bc_); const int reflen = lc_sgpr(reflen_);
  const int K = lc_sgpr(K_); const uint32_t refmask = lc_sgpr(refmask_);
  LC_GLOBAL const PreLayout &PL = C->pl;
  LC_GLOBAL PreCmp *CH = (LC_GLOBAL PreCmp *)(area + PRE_OFF_CHDR);
  LC_GLOBAL const uint32_t *occ_ref = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_OCCREF);
  LC_GLOBAL const unsigned long long *nhash = (LC_GLOBAL const unsigned long long *)(area + PRE_OFF_NHASH);
  LC_GLOBAL const unsigned long long *skey = (LC_GLOBAL const unsigned long long *)(area + PRE_OFF_SKEY);
  LC_GLOBAL const uint32_t *sidv = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_SID);
  LC_GLOBAL NodeGr *pgr = (LC_GLOBAL NodeGr *)(area + PRE_OFF_PGR);
  LC_GLOBAL const uint16_t *qv = (LC_GLOBAL const uint16_t *)(area + PRE_OFF_QV);
  LC_GLOBAL uint32_t *clive = (LC_GLOBAL uint32_t *)(area + PRE_OFF_CLIVE);
  LC_GLOBAL uint32_t *cseq = (LC_GLOBAL uint32_t *)(area + PRE_OFF_CSEQ);
  const uint32_t Pn = nsurv + 2u, SRC = nsurv, SNK = nsurv + 1u;
  // ---- LDS arena (the reads are done with; pos2si of the component search stays where it is, at byte 70304)
  LC_LDS uint8_t *arena = (LC_LDS uint8_t *)&S.bases[0];
  LC_LDS uint16_t *pos2si = (LC_LDS uint16_t *)(arena + 70304);
  LC_LDS uint16_t *E = (LC_LDS uint16_t *)arena;                               // [Pn][8]
  LC_LDS uint8_t *NE = arena + 13376;                                          // [Pn]
  LC_LDS uint8_t *FL = arena + 14272;                                          // [Pn] 1 tumor, 2 normal, 4 source, 8 sink, 16 absorbed
  LC_LDS uint16_t *SI2POS = (LC_LDS uint16_t *)(arena + 15168);                // [nsurv]
  LC_LDS uint32_t *LNK = (LC_LDS uint32_t *)(arena + 16896);                   // [Pn][2]
  LC_LDS unsigned long long *PT = (LC_LDS unsigned long long *)(arena + 23616);   // [2 Pn] port records (kernels.h pr_pack)
  LC_LDS uint32_t *HS = (LC_LDS uint32_t *)(arena + 36992);                    // [Pn + 1]
  LC_LDS uint32_t *AL = (LC_LDS uint32_t *)(arena + 40384);                    // [Pn + 1]
  LC_LDS uint16_t *HEADOF = (LC_LDS uint16_t *)(arena + 43776);                // [Pn] head (position) of an absorbed node
  LC_LDS uint8_t *INFO = arena + 45504;                                        // [Pn] 1 | frame flipped << 1 | entering direction << 2
  LC_LDS uint16_t *HX = (LC_LDS uint16_t *)(arena + 46400);                    // [Pn] index of a head in the head list
  LC_LDS uint16_t *HL = (LC_LDS uint16_t *)(arena + 48128);                    // [PB_CHEADS]
  LC_LDS uint32_t *HACC = (LC_LDS uint32_t *)(arena + 48640);                  // [PB_CHEADS][4]
  LC_LDS uint16_t *NEWE = (LC_LDS uint16_t *)(arena + 52736);                  // [PB_CHEADS][13]
  LC_LDS uint16_t *NPOS = (LC_LDS uint16_t *)(arena + 59392);                  // [N <= 4096] node id -> position (0xFFFF: not a survivor)
  LC_LDS unsigned long long *ORD = (LC_LDS unsigned long long *)(arena + 59392);  // [merged k-mers] their counted occurrences (4 x 16 bit) in merge order -- over NPOS, once markRefEnds is through
  LC_LDS unsigned long long *TCC = (LC_LDS unsigned long long *)(arena + 74400);  // [nsurv] counted occurrences Tf Tr Nf Nr of the k-mer at a position (behind pos2si)
  LC_LDS uint16_t *CI = (LC_LDS uint16_t *)(arena + 67584);                      // [nsurv] its candidate index (between NPOS and pos2si)
  static_assert(74400 + 8 * PB_CMAX <= offsetof(BlShared, big) + BL_BIG && 8 * PB_CMAX <= 2 * 4096 && 67584 + 2 * PB_CMAX <= 70304, "compress arena (2)");
  static_assert(13376 >= (PB_CMAX + 2) * 16 && 14272 - 13376 >= PB_CMAX + 2 && 16896 - 15168 >= 2 * PB_CMAX && 23616 - 16896 >= 8 * (PB_CMAX + 2) &&
                36992 - 23616 >= 16 * (PB_CMAX + 2) && 40384 - 36992 >= 4 * (PB_CMAX + 3) && 45504 - 43776 >= 2 * (PB_CMAX + 2) && 48128 - 46400 >= 2 * (PB_CMAX + 2) &&
                52736 - 48640 >= 16 * PB_CHEADS && 59392 - 52736 >= 26 * PB_CHEADS && 59392 + 2 * 4096 <= 70304, "compress arena");
  // Round 6: a graph of several components comes here too -- for its component 1 (the component of the table's first node: with a second,
  // small component beside the one that holds the window that is the large one 99 times in 100).  The window kernel works the components
  // off in turn, component 1 first; the others stay as they are (FL bit 32: not in component 1 -- no anchor, no mergeable link).
  WG_LANE0 { CH->done = 0; S.why = 0; S.flagged = 0; S.g0 = 0x7FFFFFFFu; S.g1 = 0; S.ndup = 0; S.wsum[0] = 0; S.wsum[1] = 0; }      // (wsum: the scans' partial sums, idle until the first scan below)
  BLPA(S, 1);
  WG_FOR(n, N) { NPOS[n] = (uint16_t)0xFFFFu; }
  WG_SYNC();
  WG_FOR(u, nsurv) {
    const uint32_t si = pos2si[u]; SI2POS[si] = (uint16_t)u; NPOS[sidv[si]] = (uint16_t)u;
    const uint32_t ci = X.s_ci[si]; CI[u] = (uint16_t)ci; TCC[u] = X.tcc[X.c_ti[ci]];
  }
  WG_SYNC();
  WG_FOR(u, Pn) {
    uint32_t ne = 0, fl = 0;
    if ((uint32_t)u < nsurv) {
      const uint32_t si = pos2si[u];
      LC_GLOBAL const NodeGr &G = pgr[si];
      ne = G.necnt; fl = G.flags & 3u;
      if (ne > 8u) { S.why = 1; ne = 8; }
      for (uint32_t e = 0; e < ne; ++e) E[8 * (uint32_t)u + e] = BLC_MAKE((uint32_t)SI2POS[X.s_edges[9 * (size_t)si + e]], ED_DIR(G.edges[e]));
      dev_atomic_add((LC_LDS uint32_t *)&S.wsum[0], ne);                      // (trace: edges before markRefEnds, all survivors ...)
      if (G.comp == 1) { dev_atomic_add((LC_LDS uint32_t *)&S.ndup, ne); dev_atomic_add((LC_LDS uint32_t *)&S.wsum[1], 1u); }      // (... component 1's, its survivors)
      else fl |= 32u;
    } else fl = (uint32_t)u == SRC ? 4u : 8u;
    NE[u] = (uint8_t)ne; FL[u] = (uint8_t)fl;
  }
  WG_SYNC();                                                                // (the scans read FL bit 32 of other lanes' positions)
  BLPA(S, 2);
  // ---- markRefEnds' two scans (kernels.h mark_ref_scan): first / last reference offset whose node survives with getTotCov() >= COV_THRESHOLD
  const int nrefk = reflen - K > 0 ? reflen - K + 1 : 0;
  WG_FOR(off, nrefk) {
    const uint32_t e = occ_ref[off];
    if (e & PB_GONE) continue;
    const uint32_t u = NPOS[e & ON_ID];
    const unsigned long long c4 = TCC[u];
    const uint32_t tot = (uint32_t)(c4 & 0xFFFFu) + (uint32_t)((c4 >> 16) & 0xFFFFu) + (uint32_t)((c4 >> 32) & 0xFFFFu) + (uint32_t)(c4 >> 48);
    if ((float)tot >= (float)P->cov_threshold && !(FL[u] & 32u)) { dev_atomic_min((LC_LDS uint32_t *)&S.g0, (uint32_t)off); dev_atomic_max((LC_LDS uint32_t *)&S.g1, (uint32_t)off + 1u); }      // (kernels.h mark_ref_scan: ... and is in the component)
  }
  WG_SYNC();
  const uint32_t why_a = S.why, g0_a = S.g0, g1_a = S.g1, edges0 = S.ndup, edges_all = S.wsum[0], n_c1 = S.wsum[1];     // (one pair of barriers for the six words; S.ndup serves as a flag further down)
  WG_SYNC();
  if (why_a || g0_a == 0x7FFFFFFFu) return;
  const int so = (int)g0_a, ko = (int)g1_a - 1;
  const uint32_t sn = occ_ref[so] & 0xFFFFFu, kn = occ_ref[ko] & 0xFFFFFu;
  WG_FOR(off, nrefk) {
    const uint32_t e = occ_ref[off];
    if (e & PB_GONE) continue;                                              // (a node that is gone is not "the same node again")
    const uint32_t t = e & 0xFFFFFu;
    if ((off > so && t == sn) || (off < ko && t == kn)) S.why = 2;             // ambiguous source / sink: no anchors
  }
  if (bl_bcast(&S.why)) return;
  BLPA(S, 3);
  // ---- markRefEnds proper (kernels.h mark_ref_ends), lane 0: cut the edges that leave the source k-mer backwards / the sink k-mer forwards, hang the special nodes on
  WG_LANE0 {
    auto erase_at = [&](uint32_t u, int idx) { const int cnt = (int)NE[u]; for (int i = idx; i + 1 < cnt; ++i) E[8 * u + i] = E[8 * u + i + 1]; NE[u] = (uint8_t)(cnt - 1); };
    auto remove_edge_l = [&](uint32_t u, uint32_t to, uint32_t dir) { for (int i = 0; i < (int)NE[u]; ++i) if (BLC_TO(E[8 * u + i]) == to && BLC_DIR(E[8 * u + i]) == dir) { erase_at(u, i); return; } };
    auto add_edge_l = [&](uint32_t u, uint32_t to, uint32_t dir) {
      const int cnt = (int)NE[u];
      for (int i = 0; i < cnt; ++i) if (BLC_TO(E[8 * u + i]) == to && BLC_DIR(E[8 * u + i]) == dir) return;
      if (cnt >= 8) { S.why = 3; return; }
      E[8 * u + cnt] = BLC_MAKE(to, dir); NE[u] = (uint8_t)(cnt + 1);
    };
    for (int which = 0; which < 2; ++which) {
      const uint32_t oc = occ_ref[which == 0 ? so : ko];
      const uint32_t x = NPOS[oc & ON_ID], ori = oc >> 31;
      const uint32_t sdir = which == 0 ? (ori ? 1u : 0u) : (ori ? 0u : 3u);     // source: FF, or FR when the k-mer is reversed ; sink: RR, or FF
      const char cut = which == 0 ? (ori ? 'F' : 'R') : (ori ? 'R' : 'F');       // the edges that start in this direction go
      for (int i = (int)NE[x] - 1; i >= 0; --i) {
        const uint32_t e = E[8 * x + i];
        if (dir_start(BLC_DIR(e)) == cut) { const uint32_t other = BLC_TO(e); if (other != x) { remove_edge_l(other, x, fliplink(BLC_DIR(e))); erase_at(x, i); } }
      }
      const uint32_t sp = which == 0 ? SRC : SNK;
      add_edge_l(sp, x, sdir);
      add_edge_l(x, sp, fliplink(sdir));
    }
  }
  if (bl_bcast(&S.why)) return;
  BLPA(S, 4);
  // ---- where unordered_map::insert puts the two special nodes (kernels.h order_insert; no rehash: the caller checked): in front of the first
  //      element of the same bucket, else at the head.  S.g0 / S.g1 = position of the source / the sink in the table after both insertions.
  unsigned long long hsrc, hsnk;
  {
    const char ns[] = "source1", nk[] = "sink1";
    hsrc = std_hash_bytes([&](int j) -> int { return (int)(unsigned char)ns[j]; }, 7);
    hsnk = std_hash_bytes([&](int j) -> int { return (int)(unsigned char)nk[j]; }, 5);
  }
  const uint32_t bsrc = ht_mod(hsrc, ht_bc), bsnk = ht_mod(hsnk, ht_bc);
  WG_LANE0 { S.g0 = 0x7FFFFFFFu; S.g1 = 0x7FFFFFFFu; }
  WG_SYNC();
  WG_FOR(u, nsurv) {
    const uint32_t b = ht_mod(nhash[sidv[pos2si[u]]], ht_bc);
    if (b == bsrc) dev_atomic_min((LC_LDS uint32_t *)&S.g0, (uint32_t)u);
    if (b == bsnk) dev_atomic_min((LC_LDS uint32_t *)&S.g1, (uint32_t)u);
  }
  WG_SYNC();
  uint32_t at_s = S.g0, at_k = S.g1;
  WG_SYNC();
  if (at_s == 0x7FFFFFFFu) at_s = 0;
  // (the sink is inserted into the order that holds the source already)
  if (at_k != 0x7FFFFFFFu) at_k = at_k >= at_s ? at_k + 1u : at_k;
  if (bsnk == bsrc && (at_k == 0x7FFFFFFFu || at_s < at_k)) at_k = at_s;
  if (at_k == 0x7FFFFFFFu) at_k = 0;
  if (at_k <= at_s) at_s += 1u;                                                // final index of the source once the sink is in
  // final table index of position u (before cleanDead): real nodes shift by the insertions in front of them
  auto final_index = [&](uint32_t u) -> uint32_t {
    if (u == SRC) return at_s;
    if (u == SNK) return at_k;
    uint32_t i = u; const uint32_t s0 = at_k <= at_s ? at_s - 1u : at_s;      // source's index before the sink went in
    if (i >= s0) ++i;
    if (i >= at_k) ++i;
    return i;
  };
  BLPA(S, 5);
  // ---- compress_prepare: the mergeable link of every node in either direction
  auto buddy = [&](uint32_t u, char dir) -> int {                               // Node_t::getBuddy
    if (FL[u] & 12u) return -1;
    int ret = -1; const int cnt = (int)NE[u];
    for (int i = 0; i < cnt; ++i) if (is_dir(BLC_DIR(E[8 * u + i]), dir)) { if (ret != -1) return -1; ret = i; }
    if (ret != -1 && BLC_TO(E[8 * u + ret]) == u) return -1;
    return ret;
  };
  auto tandem = [&](uint32_t u) -> bool { const int cnt = (int)NE[u]; for (int i = 0; i < cnt; ++i) if (BLC_TO(E[8 * u + i]) == u) return true; return false; };
  WG_FOR(u, Pn) {
    uint32_t l0 = 0, l1 = 0;
    if (!(FL[u] & (12u | 32u)) && !tandem((uint32_t)u)) {
      for (int sd = 0; sd < 2; ++sd) {
        const int uid = buddy((uint32_t)u, sd == 0 ? 'F' : 'R');
        if (uid < 0) continue;
        const uint32_t ew = E[8 * (uint32_t)u + (uint32_t)uid], edir = BLC_DIR(ew), b = BLC_TO(ew);
        if ((FL[b] & 12u) || tandem(b)) continue;
        const int buid = buddy(b, (edir == 0 || edir == 2) ? 'R' : 'F');
        if (buid < 0) continue;
        if (BLC_TO(E[8 * b + (uint32_t)buid]) != (uint32_t)u) { S.why = 4; continue; }      // an irregular link: the literal replay's business
        (sd == 0 ? l0 : l1) = CL_VALID | (edir << 28) | b;
      }
    }
    LNK[2 * u] = l0; LNK[2 * u + 1] = l1;
  }
  if (bl_bcast(&S.why)) return;
  BLPA(S, 6);
  // ---- compress_rank: ports, pointer jumping
  WG_FOR(u, Pn) {
    for (uint32_t sd = 0; sd < 2; ++sd) {
      const uint32_t l = LNK[2 * u + sd];
      lc_u4 r; r.x = LC_NIL; r.y = 0; r.z = (uint32_t)u; r.w = 2u * (uint32_t)u + sd;
      if (l & CL_VALID) {
        const uint32_t B = CL_TO(l), ed = CL_DIR(l);
        const uint32_t sb = (ed == 0 || ed == 2) ? 1u : 0u;
        const uint32_t back = LNK[2 * B + sb];
        if (!(back & CL_VALID) || CL_TO(back) != (uint32_t)u || CL_DIR(back) != fliplink(ed)) S.why = 5;
        r.x = 2u * B + (1u - sb); r.y = 1u | (((ed == 1 || ed == 2) ? 1u : 0u) << 31);
      }
      PT[2u * (uint32_t)u + sd] = pr_pack(r);
    }
  }
  if (bl_bcast(&S.why)) return;
  {
    const int NP = (int)(2u * Pn);
    // (two change flags taken in turn -- S.flagged / S.ndup -- so that a round is two barriers: read, barrier, write + flag, barrier, look)
    WG_LANE0 { S.flagged = 0; S.ndup = 0; }
    WG_SYNC();
    for (int round = 0; ; ++round) {
      if (round == 15) { return; }                                             // a ring: no port ever reaches an end
      LC_LDS uint32_t *flag = (round & 1) ? (LC_LDS uint32_t *)&S.ndup : (LC_LDS uint32_t *)&S.flagged;
      LC_LDS uint32_t *other = (round & 1) ? (LC_LDS uint32_t *)&S.flagged : (LC_LDS uint32_t *)&S.ndup;
      lc_u4 r[4]; bool have[4], chg = false;
#ifndef LANCET_WAVE_EMU
      const int l = (int)threadIdx.x;
      for (int q = 0; q < 4; ++q) {
        const int p = l + q * BL_WG; have[q] = p < NP;
        if (have[q]) { r[q] = pr_unpack(PT[p]); if (r[q].x != LC_NIL) { const lc_u4 t = pr_unpack(PT[r[q].x]);
          r[q].y = (((r[q].y & 0x7FFFFFFFu) + (t.y & 0x7FFFFFFFu)) & 0x7FFFFFFFu) | ((r[q].y ^ t.y) & 0x80000000u);
          if (t.z < r[q].z) r[q].z = t.z;
          r[q].w = t.w; r[q].x = t.x; if (r[q].x != LC_NIL) chg = true; } }
      }
      __syncthreads();
      for (int q = 0; q < 4; ++q) if (have[q]) PT[l + q * BL_WG] = pr_pack(r[q]);
      if (chg) *flag = 1;
      if (l == 0) *other = 0;
      __syncthreads();
      if (!*flag) break;
#else
      {                                                                       // (lanes one after the other: all reads of a round see the round's start)
        std::vector<unsigned long long> nxt((size_t)NP);
        for (int p = 0; p < NP; ++p) {
          lc_u4 a = pr_unpack(PT[p]);
          if (a.x != LC_NIL) { const lc_u4 t = pr_unpack(PT[a.x]);
            a.y = (((a.y & 0x7FFFFFFFu) + (t.y & 0x7FFFFFFFu)) & 0x7FFFFFFFu) | ((a.y ^ t.y) & 0x80000000u);
            if (t.z < a.z) a.z = t.z;
            a.w = t.w; a.x = t.x; if (a.x != LC_NIL) chg = true; }
          nxt[(size_t)p] = pr_pack(a);
        }
        for (int p = 0; p < NP; ++p) PT[p] = nxt[(size_t)p];
        *flag = chg ? 1u : 0u; *other = 0;
        (void)r; (void)have;
        if (!*flag) break;
      }
#endif
    }
    static_assert(4 * BL_WG >= 2 * (PB_CMAX + 2), "ports per lane");
  }
  WG_SYNC();
  BLPA(S, 7);
  // ---- heads, slices of the merge-order list, arena space of the new deques
  WG_LANE0 { S.flagged = 0; }
  WG_SYNC();
  WG_FOR(u, Pn + 1) {
    uint32_t hs = 0, al = 0;
    if ((uint32_t)u < Pn) {
      const lc_u4 a = pr_unpack(PT[2 * (size_t)u]), b = pr_unpack(PT[2 * (size_t)u + 1]);
      const uint32_t mF = a.y & 0x7FFFFFFFu, mR = b.y & 0x7FFFFFFFu, cmin = a.z < b.z ? a.z : b.z;
      if (mF + mR > 0 && cmin == (uint32_t)u) {
        hs = mF + mR; al = (uint32_t)K + mF + mR;
        const uint32_t hx = dev_atomic_add((LC_LDS uint32_t *)&S.flagged, 1u);
        if (hx < PB_CHEADS) { HL[hx] = (uint16_t)u; HX[u] = (uint16_t)hx; HACC[4 * hx] = 0x7FFFFFFFu; HACC[4 * hx + 1] = 0x7FFFFFFFu; HACC[4 * hx + 2] = 0; HACC[4 * hx + 3] = 0; }
      }
    }
    HS[u] = hs; AL[u] = al;
  }
  bl_scan32(HS, (int)Pn + 1, S);                                              // (ends in a barrier: scan_total is readable)
  const uint32_t nabs = S.scan_total;
  bl_scan32(AL, (int)Pn + 1, S);
  const uint32_t need = S.scan_total, nheads = S.flagged;
  WG_SYNC();
  if (nheads > PB_CHEADS || need > PB_CSEQ || nabs > PB_CMAX) return;
  // per node: its k-mer's figures (what compress_prepare keeps in a CmpRec)
  auto node_key = [&](uint32_t u, uint32_t *ci_out) -> unsigned long long { const uint32_t ci = CI[u]; *ci_out = ci; return skey[(size_t)ci * PL.kw]; };      // (K <= 31 here: one word)
  const uint32_t top = ncand * (uint32_t)K;                                     // the window kernel's arena top for a graph from here
  BLPA(S, 8);
  // ---- every merged k-mer: its head, side, place in the merge order; descriptor into the head's deque, coverage into its slice
  WG_FOR(u, Pn) {
    const lc_u4 a = pr_unpack(PT[2 * (size_t)u]), b = pr_unpack(PT[2 * (size_t)u + 1]);
    const uint32_t mF = a.y & 0x7FFFFFFFu, mR = b.y & 0x7FFFFFFFu;
    if (mF + mR == 0) continue;
    const uint32_t cmin = a.z < b.z ? a.z : b.z;
    if (cmin == (uint32_t)u) continue;                                         // a head
    const uint32_t sH = (a.z == cmin) ? 0u : 1u;
    const lc_u4 away = sH ? a : b;
    const lc_u4 hF = pr_unpack(PT[2 * (size_t)cmin]), hR = pr_unpack(PT[2 * (size_t)cmin + 1]);
    const bool onF = away.w == hF.w && (hF.y & 0x7FFFFFFFu) > 0;
    const lc_u4 hp = onF ? hF : hR;
    const uint32_t mdir = hp.y & 0x7FFFFFFFu, hmF = hF.y & 0x7FFFFFFFu, hmR = hR.y & 0x7FFFFFFFu;
    const uint32_t j = mdir - (away.y & 0x7FFFFFFFu);
    const uint32_t st_j = ((hp.y ^ away.y) >> 31) & 1u;
    const uint32_t raw = fliplink(CL_DIR(LNK[2 * u + sH]));
    const uint32_t st_p = st_j ^ ((raw == 1 || raw == 2) ? 1u : 0u);
    const uint32_t edir = st_p ? flipme(raw) : raw;
    const bool brev = (edir == 1 || edir == 3);
    uint32_t ci; const unsigned long long kk = node_key((uint32_t)u, &ci);
    const uint32_t n = sidv[pos2si[u]];
    const uint32_t d0 = SD_MAKE(n, 0, key_base(&kk, K, 0)), dK = SD_MAKE(n, K - 1, key_base(&kk, K, K - 1));
    const uint32_t nb = AL[cmin];
    const uint32_t d = brev ? (d0 ^ 3u) : dK;
    if (onF) cseq[nb + hmR + (uint32_t)K + (j - 1u)] = d; else cseq[nb + hmR - j] = d ^ 3u;
    const uint32_t t = HS[cmin] + (onF ? j - 1u : hmF + j - 1u);
    const unsigned long long c4 = TCC[u];
    const uint32_t c0 = (uint32_t)(c4 & 0xFFFFu), c1 = (uint32_t)((c4 >> 16) & 0xFFFFu), c2 = (uint32_t)((c4 >> 32) & 0xFFFFu), c3 = (uint32_t)(c4 >> 48);
    ORD[t] = c4;
    LC_GLOBAL const uint16_t *q0p = qv + ((size_t)ci * K + 0) * 4, *qKp = qv + ((size_t)ci * K + (size_t)(K - 1)) * 4;
    const uint32_t tq0 = (uint32_t)q0p[0] + q0p[1] + q0p[2] + q0p[3], tqK = (uint32_t)qKp[0] + qKp[1] + qKp[2] + qKp[3];
    const uint32_t fl = FL[u] & 3u, hx = HX[cmin];
    dev_atomic_min(&HACC[4 * hx], c0 + c1 + c2 + c3); dev_atomic_min(&HACC[4 * hx + 1], brev ? tq0 : tqK);
    if (fl) dev_atomic_or(&HACC[4 * hx + 2], fl);
    if (fl == 1u) dev_atomic_add(&HACC[4 * hx + 3], 1u);
    HEADOF[u] = (uint16_t)cmin; INFO[u] = (uint8_t)(1u | (st_j << 1) | (edir << 2));
    FL[u] = (uint8_t)(FL[u] | 16u);
  }
  WG_SYNC();
  BLPA(S, 9);
  // ---- the heads: own k-mer's descriptors, minima / flags, the new edge list (own edges without the merged links, then the outward
  //      edges of the F-side end, then of the R-side end: the erase / push_back order of compressNode)
  WG_FOR(hx, nheads) {
    const uint32_t u = HL[hx];
    const lc_u4 a = pr_unpack(PT[2 * (size_t)u]), b = pr_unpack(PT[2 * (size_t)u + 1]);
    const uint32_t mF = a.y & 0x7FFFFFFFu, mR = b.y & 0x7FFFFFFFu;
    uint32_t ci; const unsigned long long kk = node_key(u, &ci);
    const uint32_t si = pos2si[u], n = sidv[si];
    const uint32_t nb = AL[u];
    for (int t = 0; t < K; ++t) cseq[nb + mR + (uint32_t)t] = SD_MAKE(n, t, key_base(&kk, K, t));
    uint16_t el[13]; int m = 0; bool bad = false;
    const int uF = mF ? buddy(u, 'F') : -1, uR = mR ? buddy(u, 'R') : -1;
    if ((mF && uF < 0) || (mR && uR < 0)) bad = true;
    for (int e = 0; e < (int)NE[u]; ++e) { if (e == uF || e == uR) continue; el[m++] = E[8 * u + (uint32_t)e]; }
    for (int side = 0; side < 2 && !bad; ++side) {
      if (!(side == 0 ? mF : mR)) continue;
      const uint32_t En = ((side == 0 ? a.w : b.w) >> 1);                     // the end of the list in that direction
      const uint32_t info = INFO[En], ed = (info >> 2) & 3u, st = (info >> 1) & 1u;
      const int buid = buddy(En, (ed == 0 || ed == 2) ? 'R' : 'F');
      for (int e = 0; e < (int)NE[En]; ++e) {
        if (e == buid) continue;
        const uint32_t be = E[8 * En + (uint32_t)e];
        uint32_t ndir = BLC_DIR(be); if (st) ndir = flipme(ndir);
        const uint32_t other = BLC_TO(be);
        if (m >= 12) { bad = true; break; }
        el[m++] = BLC_MAKE(other == En ? u : other, ndir);
      }
    }
    if (bad) { S.why = 6; continue; }
    NEWE[13 * (size_t)hx] = (uint16_t)m;
    for (int e = 0; e < m; ++e) NEWE[13 * (size_t)hx + 1 + (size_t)e] = el[e];
  }
  if (bl_bcast(&S.why)) return;                                                // (nothing in the hand-off records has been touched so far)
  BLPA(S, 10);
  // ---- from here on the records change: no way back
  WG_FOR(hx, nheads) {                                                         // record fields that do not depend on the merge order
    const uint32_t u = HL[hx];
    const uint32_t cnt = HS[u + 1] - HS[u];
    uint32_t ci; (void)node_key(u, &ci);
    const uint32_t si = pos2si[u];
    const uint32_t nb = AL[u];
    LC_GLOBAL NodeGr &G = pgr[si];
    const unsigned long long c4 = TCC[u];
    const uint32_t own = (uint32_t)(c4 & 0xFFFFu) + (uint32_t)((c4 >> 16) & 0xFFFFu) + (uint32_t)((c4 >> 32) & 0xFFFFu) + (uint32_t)(c4 >> 48);
    int mn = (int)own, mq = (int)X.c_minqv[ci];
    if ((int)HACC[4 * hx] < mn) mn = (int)HACC[4 * hx];
    if ((int)HACC[4 * hx + 1] < mq) mq = (int)HACC[4 * hx + 1];
    const uint32_t f0 = FL[u] & 3u;
    G.flags = (G.flags | HACC[4 * hx + 2]);
    G.nkm = 1u + cnt; G.nkmT = (f0 == 1u ? 1u : 0u) + HACC[4 * hx + 3];
    G.mincov = mn; G.mincovqv = mq;
    G.seq_clo = top + nb; G.seq_lo = top + nb; G.seq_hi = top + nb + (uint32_t)K + cnt; G.seq_chi = G.seq_hi;
  }
  BLPA(S, 11);
  // ---- the float averaging of the merges (Graph.cc:2632-2636) in merge order is the one strictly sequential piece of the build -- a clean
  //      window is ONE chain of ~500 merges, a chain of dependent float operations on four lanes while 508 wait: it was 0.8 of this kernel's
  //      14.5 workgroup-seconds per batch.  Round 6: it is left to the window kernel (load_prebuilt), where the same 25 us per window are
  //      one wave's of sixteen per CU.  What it needs leaves here: the merged k-mers' counts in merge order (CORD) and, per unitig head, its
  //      record, its slice of that list and the number of merges (CHL); the head's record keeps cov[] = its own k-mer's counts until then.
  {
    LC_GLOBAL unsigned long long *cord = (LC_GLOBAL unsigned long long *)(area + PRE_OFF_CORD);
    LC_GLOBAL uint32_t *chl = (LC_GLOBAL uint32_t *)(area + PRE_OFF_CHL);
    WG_FOR(t, nabs) { cord[t] = ORD[t]; }
    WG_FOR(hx, nheads) { const uint32_t u = HL[hx]; chl[3 * hx] = (uint32_t)pos2si[u]; chl[3 * hx + 1] = HS[u]; chl[3 * hx + 2] = HS[u + 1] - HS[u]; }
  }
  BLPA(S, 12);
  // ---- every live node: edges into a merged k-mer go to its head (frame flipped with it); the records' edge lists; dead flags
  WG_FOR(u, Pn) {
    if (FL[u] & 16u) continue;                     // (a merged k-mer: the window kernel marks it dead itself -- its record is not touched again here)
    const bool head = HS[u + 1] != HS[u];
    const uint32_t hx = head ? (uint32_t)HX[u] : 0u;
    const int cnt = head ? (int)NEWE[13 * (size_t)hx] : (int)NE[u];
    LC_GLOBAL NodeGr &G = pgr[(uint32_t)u < nsurv ? (uint32_t)pos2si[u] : nsurv + ((uint32_t)u - nsurv)];
    if ((uint32_t)u >= nsurv) {                                                // a special node: the record special_new makes
      G.flags = (uint32_t)u == SRC ? NF_SOURCE : NF_SINK; G.comp = 1; G.color = 0; G.mincov = 0; G.mincovqv = 0; G.nqv = LC_NIL;
      G.seq_lo = G.seq_hi = G.seq_clo = G.seq_chi = 0; G.nkm = 0; G.nkmT = 0; G.onref = 0;
      for (int q = 0; q < 4; ++q) { G.kc[q] = 0; G.cov[q] = 0.0f; }
    }
    for (int e = 0; e < cnt; ++e) {
      uint32_t ew = head ? NEWE[13 * (size_t)hx + 1 + (size_t)e] : E[8 * (uint32_t)u + (uint32_t)e];
      uint32_t to = BLC_TO(ew), dir = BLC_DIR(ew);
      if (FL[to] & 16u) { dir ^= (INFO[to] >> 1) & 1u; to = HEADOF[to]; }
      G.edges[e] = ED_MAKE(to < nsurv ? sidv[pos2si[to]] : PB_SPECIAL + (to - nsurv), dir);
      if (head) NEWE[13 * (size_t)hx + 1 + (size_t)e] = BLC_MAKE(to, dir); else E[8 * (uint32_t)u + (uint32_t)e] = BLC_MAKE(to, dir);      // (the final list by position too: the cycle check below)
    }
    for (int e = cnt; e < LC_EMAX; ++e) G.edges[e] = 0;
    G.necnt = (uint32_t)cnt;
  }
  BLPA(S, 13);
  // ---- cleanDead: the table order without the merged k-mers, the two special nodes in their places
  WG_SYNC();
  // ---- the scheduling hint made exact (round 6).  S.hint says "a surviving node was met twice in one read / in both orientations: this k
  //      will very likely be rejected for a cycle" and makes the workgroup build the next k at once; a third of the graphs built ahead that
  //      way were never asked for (3032 built, 2037 taken per 32 768 windows: 0.4 workgroup-seconds).  The graph the window kernel's first
  //      hasCycle will walk (reference src/Graph.cc:593-681: the DFS from the source in either direction) lies here, compressed, in LDS:
  //      lane 0 walks it when the hint is set and takes the hint back when there is no cycle.  Scheduling only -- a graph that is then
  //      missing (a cycle that only appears after removeTips, a near-repeat in a path) is built by the service on request, as for any
  //      window without a hint; results never depend on what was built ahead.
  WG_LANE0 {
    if (S.hint) {
      LC_LDS uint8_t *col = INFO;                                                 // (INFO is done with) 0 special / absorbed, 1 unvisited, 2 on the stack, 3 done
      LC_LDS uint16_t *stk = (LC_LDS uint16_t *)LNK;                              // (the links are done with) frames of (position, next edge, direction)
      static_assert(8 * (PB_CMAX + 2) >= 6 * (PB_CMAX + 3), "DFS stack in the link array");
      for (uint32_t u = 0; u < Pn; ++u) col[u] = (FL[u] & (12u | 16u)) ? 0 : 1;
      bool cyc = false;
      for (int pass = 0; pass < 2 && !cyc; ++pass) {
        uint32_t sp = 1; stk[0] = (uint16_t)SRC; stk[1] = 0; stk[2] = (uint16_t)(pass == 0 ? 'F' : 'R');
        col[SRC] = 2;
        while (sp && !cyc) {
          LC_LDS uint16_t *fr = stk + 3 * (sp - 1);
          const uint32_t node = fr[0]; const char dir = (char)fr[2];
          const bool hd = HS[node + 1] != HS[node];
          const uint32_t hxn = hd ? (uint32_t)HX[node] : 0u;
          const uint32_t ne = hd ? (uint32_t)NEWE[13 * (size_t)hxn] : (uint32_t)NE[node];
          bool descended = false;
          uint32_t ei = fr[1];
          while (ei < ne) {
            const uint32_t e = hd ? NEWE[13 * (size_t)hxn + 1 + ei] : E[8 * node + ei]; ++ei;
            if (!is_dir(BLC_DIR(e), dir)) continue;
            const uint32_t other = BLC_TO(e);
            if (FL[other] & 12u) continue;
            const uint32_t oc = col[other];
            if (oc == 2) { cyc = true; break; }
            if (oc == 1) {
              col[other] = 2; fr[1] = (uint16_t)ei;
              LC_LDS uint16_t *nf = stk + 3 * sp; nf[0] = (uint16_t)other; nf[1] = 0; nf[2] = (uint16_t)dir_dest(BLC_DIR(e)); ++sp;
              descended = true; break;
            }
          }
          if (cyc) break;
          if (!descended) { col[node] = 3; --sp; }
        }
      }
      if (!cyc) S.hint = 0;
    }
  }
  WG_FOR(u, Pn + 1) { AL[u] = ((uint32_t)u < Pn && !(FL[u] & 16u)) ? 1u : 0u; }   // (AL is done with: keep flags by FINAL index)
  WG_SYNC();
  {
    LC_LDS uint32_t *keep = HS;                                                 // (HS is done with too) keep[final index]
    WG_FOR(i, Pn + 1) { keep[i] = 0; }
    WG_SYNC();
    WG_FOR(u, Pn) { if (AL[u]) keep[final_index((uint32_t)u)] = 1u; }
    bl_scan32(keep, (int)Pn + 1, S);
    WG_FOR(u, Pn) { if (AL[u]) clive[keep[final_index((uint32_t)u)]] = (uint32_t)u < nsurv ? (uint32_t)pos2si[u] : (0x80000000u | ((uint32_t)u - nsurv)); }
  }
  WG_LANE0 {
    CH->m_live = Pn - nabs; CH->dead = nabs; CH->seqn = need; CH->src_off = so; CH->snk_off = ko; CH->edges0 = edges0; CH->cov_heads = nheads; CH->edges_all = edges_all; CH->n_c1 = n_c1; CH->refmask = refmask; CH->pad1 = 0;
    CH->spec_hash[0] = hsrc; CH->spec_hash[1] = hsnk;
    CH->done = 1;
  }
  WG_SYNC();
}

// One window.  Returns with the hand-off area of the window written (PB_BUILT) or marked PB_NOT_BUILT.
// kmin: the loop over k starts there (min_k for the window's first graph; the k after a rejected one for a graph built ahead).
// rep: the window's isRepeat / isAlmostRepeat operands when an earlier call scanned the reference already, else null.
// cmp_later: a graph at a later k of the window's loop (rep != null) gets markRefEnds and the first compress here too -- the build service's
//            graphs: a window that was put aside for one is on the critical path of the launch when it comes back, and the service has the
//            time; graphs built ahead by the build kernel itself leave that to the window kernel (a workgroup-second there is worth six
//            slot-seconds here).  Neither step depends on what an earlier k left of Ref_t::seq (they scan every offset of rawseq for live
//            nodes: kernels.h mark_ref_scan); what does -- the mer-table flags, the reference coverage -- load_prebuilt sets right.
DEVNI void bl_build_window(LC_GLOBAL const lancet_params *P, LC_GLOBAL const DevBatch *Bp, LC_GLOBAL const EngineCaps *C, BL_S &S, LC_GLOBAL uint8_t *xbase,
                           LC_GLOBAL uint8_t *area, int w, int kmin, LC_GLOBAL const PreHdr *rep, bool cmp_later = false) {
  P = lc_sgpr(P); Bp = lc_sgpr(Bp); C = lc_sgpr(C); xbase = lc_sgpr(xbase); area = lc_sgpr(area); rep = lc_sgpr(rep);      // (uniform arguments: scalar registers, wave.h lc_sgpr)
  w = lc_sgpr(w); kmin = lc_sgpr(kmin);
  LC_GLOBAL const DevBatch &B = *Bp;
  LC_GLOBAL const PreLayout &PL = C->pl;
  BlScratch X; bl_scratch_carve(&X, xbase);                       // (a local of this function: its pointers live in registers)
  LC_GLOBAL PreHdr *H = (LC_GLOBAL PreHdr *)(area + PRE_OFF_HDR);
  const uint32_t g0 = B.read_begin[w];
  const int nr = (int)(B.read_begin[w + 1] - g0);
  const int reflen = (int)(B.ref_off[w + 1] - B.ref_off[w]);
  LC_GLOBAL const uint8_t *refc = B.ref_codes + B.ref_off[w];
  WG_LANE0 { S.hint = 0; S.ndup = 0; S.w = w; S.why = BLW_NONE; S.R = nr + 1; S.reflen = reflen; S.hasN = 0; S.mapped = 0; S.flagged = 0; S.npairs = 0; S.edges_total = 0; S.refn = 0;
             H->status = PB_NOT_BUILT; H->why = 0; H->have_rep = 0; H->heavy = 0; H->next = 0; H->lr = 0; H->lr_total = 0;
             ((LC_GLOBAL PreCmp *)(area + PRE_OFF_CHDR))->done = 0;
             if (nr > BL_RMAX || reflen > (int)PL.maxw || reflen < 1) S.why = BLW_SIZE; }
  if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
  // ---- mapped reads, N in the window reference, per-read geometry
  LC_LDS uint32_t *tmpA = S.big, *tmpB = S.big + (BL_RMAX + 8), *tmpC = S.big + 2 * (BL_RMAX + 8);
  WG_FOR(r, nr + 1) {
    uint32_t ri = 0; int tlen = reflen;
    if (r < nr) { ri = B.rinfo[g0 + (uint32_t)r]; tlen = (int)RI_TLEN(ri); if (RI_MAPPED(ri)) dev_atomic_add((LC_LDS uint32_t *)&S.mapped, 1u);
                  if (tlen > 1023) S.flagged = 1; }                         // (k-mer positions are 10 bits here: such a window is the general build's)
    S.rinfo[r] = ri;
    tmpA[r] = (uint32_t)((tlen + 15) / 16); tmpB[r] = (uint32_t)((tlen + 31) / 32);
  }
  WG_FOR(i, reflen) { if (refc[i] > 3) S.hasN = 1; }
  WG_LANE0 { tmpA[nr + 1] = 0; tmpB[nr + 1] = 0; }
  bl_scan32(tmpA, nr + 2, S);
  const uint32_t nbw = lc_sgpr((uint32_t)S.scan_total);                           // (a scan ends in a barrier, and the next one overwrites the total only behind two of its own)
  bl_scan32(tmpB, nr + 2, S);
  const uint32_t ngw = lc_sgpr((uint32_t)S.scan_total);
  WG_SYNC();
  WG_LANE0 {
    S.nbw = nbw; S.ngw = ngw;
    if (S.mapped <= 0) S.why = BLW_NOREADS;                    // the window kernel reports LANCET_W_NO_READS itself
    else if (S.hasN) S.why = BLW_HASN;
    else if (nbw > BL_BASES / 16 || ngw > BL_BASES / 32 || S.flagged) S.why = BLW_SIZE;
    S.flagged = 0;
  }
  if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
  WG_FOR(r, nr + 2) { S.rdo[r] = (uint16_t)tmpA[r]; S.gwo[r] = (uint16_t)tmpB[r]; }
  WG_SYNC();
  BLP(S, 1);
  if (C->debug_stop == 101u) { WG_LANE0 { H->why = 99; } return; }
  // ---- the window's packed reads and quality masks into LDS (whole words; a read starts on a word)
  WG_FOR(r, nr) {
    const uint32_t ri = S.rinfo[r]; const int tlen = (int)RI_TLEN(ri);
    LC_GLOBAL const uint32_t *bsrc = B.bases + B.base_woff[g0 + (uint32_t)r], *gsrc = B.good + B.good_woff[g0 + (uint32_t)r];
    const int nb = (tlen + 15) / 16, ng = (tlen + 31) / 32;
    for (int i = 0; i < nb; ++i) S.bases[S.rdo[r] + i] = bsrc[i];
    for (int i = 0; i < ng; ++i) S.goodm[S.gwo[r] + i] = gsrc[i];
  }
  WG_FOR(wd, (reflen + 15) / 16) {
    uint32_t v = 0;
    for (int j = 0; j < 16 && wd * 16 + j < reflen; ++j) v |= (uint32_t)(refc[wd * 16 + j] & 3u) << (2 * j);
    S.bases[S.rdo[nr] + wd] = v;
  }
  WG_FOR(i, 4) { S.bases[nbw + (uint32_t)i] = 0; }
  WG_SYNC();
  BLP(S, 2);
  if (C->debug_stop == 102u) { WG_LANE0 { H->why = 99; } return; }
  // ---- reference repeat scan -> the first k of the loop that reaches buildgraph (Microassembler.cc:118-131)
  if (rep) { WG_LANE0 { S.repE = rep->refE; S.repM = rep->refM; } WG_SYNC(); }
  else repeat_scan_min((volatile LC_LDS unsigned long long *)(S.big + 4 * (BL_RMAX + 8)), refc, reflen, P->max_mismatch, P->min_k, P->min_k + 1, (volatile LC_LDS int *)&S.repE, (volatile LC_LDS int *)&S.repM,
                  (const LC_LDS uint32_t *)&S.bases[S.rdo[nr]], (volatile LC_LDS int *)&S.flagged);     // (the reference is in LDS already, 2 bits per base: no N here)
  WG_LANE0 {
    int K = 0;
    for (int k = kmin; k <= P->max_k; k += 2) {
      if (reflen - k > 0 && S.repE >= k) continue;
      if (reflen - k > 0 && S.repM >= k + 1) continue;
      K = k; break;
    }
    S.K = K;
    H->refE = S.repE; H->refM = S.repM; H->mapped = (uint32_t)S.mapped; H->have_rep = 1;          // (the window kernel does not repeat the scan)
    if (K == 0 || (K & 1) == 0) S.why = BLW_K;
    else if (2 * K > 64 * (int)PL.kw) S.why = BLW_K;                     // (the batch's hand-off areas hold one-word keys)
    else if (K > BL_KMAX) S.why = BLW_KBIG;                              // (the 1024-lane configuration takes it off the list)
  }
  WG_SYNC();
  const int why_k = lc_sgpr((int)S.why); const int K = lc_sgpr((int)S.K);                    // (both words between one pair of barriers)
  WG_SYNC();
  if (why_k) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
  const int NW = (2 * K + 63) / 64;                                      // 64-bit words of a k-mer
  const unsigned long long kmask = K < 32 ? (1ULL << (2 * K)) - 1ULL : ~0ULL;   // (one-word form: K <= 31)
  (void)kmask; (void)NW;
  const int R = nr + 1;
  BLP(S, 3);
  if (C->debug_stop == 103u) { WG_LANE0 { H->why = 99; } return; }
  // ---- occurrence index space (read r owns its k-mers p = 0..tlen-K; a read of exactly K bases has none, Graph.cc:142-143)
  WG_LANE0 { S.totalreadbp = 0; S.n_kmers = 0; }
  WG_SYNC();
  WG_FOR(r, R) {
    const int tlen = r < nr ? (int)RI_TLEN(S.rinfo[r]) : reflen;
    tmpC[r] = tlen - K > 0 ? (uint32_t)(tlen - K + 1) : 0u;
    if (tlen > 0 && r < nr) dev_atomic_add((LC_LDS uint32_t *)&S.totalreadbp, (uint32_t)tlen);
    if (tlen - K > 0) dev_atomic_add((LC_LDS uint32_t *)&S.n_kmers, (uint32_t)(tlen - K));
  }
  WG_LANE0 { tmpC[R] = 0; }
  // (one scan for two prefix sums: k-mer starts in the low 17 bits -- at most BL_BASES < 2^17 of them -- and chunks of eight above)
  WG_FOR(r, R) { tmpC[r] |= ((tmpC[r] + 7u) >> 3) << 17; }
  bl_scan32(tmpC, R + 1, S);
  WG_LANE0 { S.O = S.scan_total & 0x1FFFFu; }                  // (all chunks of the window: cbase[R], below)
  WG_FOR(r, R + 1) { S.cbase[r] = (bl_occ_t)(tmpC[r] >> 17); }
  WG_LANE0 { S.cbase[R + 1] = (bl_occ_t)~(bl_occ_t)0; }
  WG_SYNC();
  WG_FOR(r, R) {                                                 // read r holds the chunks [cbase[r], cbase[r+1]): the multiples of 16 among them
    const uint32_t a = S.cbase[r], b = S.cbase[r + 1];
    for (uint32_t j = (a + 15u) >> 4; (j << 4) < b; ++j) S.c2r[j] = (uint16_t)r;
  }
  WG_SYNC();
  // ---- mate pairs: a read with exactly one earlier read of the same name and the opposite mate number is the later mate of
  //      a pair (kernels.h build_tables: cand / mate_of); several such reads -> general path
  {
    LC_LDS uint32_t *first = S.big;                              // [2 * (BL_RMAX + 1)] smallest read index per (name, mate)
    LC_LDS uint32_t *cnt = S.big + 2 * (BL_RMAX + 1);
    WG_FOR(i, 4 * (BL_RMAX + 1)) { S.big[i] = i < 2 * (BL_RMAX + 1) ? 0xFFFFFFFFu : 0u; }
    WG_FOR(r, R) { S.pidx[r] = 0xFF; S.prole[r] = 0; }
    WG_SYNC();
    WG_FOR(r, nr) {
      const uint32_t mi = RI_MATE(S.rinfo[r]);
      if (mi == 1 || mi == 2) {
        const uint32_t nm = B.name_rank[g0 + (uint32_t)r];
        if (nm > BL_RMAX) S.why = BLW_NAMES;
        else { dev_atomic_min(&first[2 * nm + (mi - 1)], (uint32_t)r); dev_atomic_add(&cnt[2 * nm + (mi - 1)], 1u); }
      }
    }
    WG_SYNC();
    LC_LDS uint32_t *isl = S.big + 4 * (BL_RMAX + 1);          // later-mate flag per read -> pair index by scan
    WG_FOR(r, nr + 1) {
      uint32_t later = 0;
      if (r < nr && !S.why) {
        const uint32_t mi = RI_MATE(S.rinfo[r]);
        if (mi == 1 || mi == 2) {
          const uint32_t nm = B.name_rank[g0 + (uint32_t)r];
          const uint32_t oc = cnt[2 * nm + (2 - mi)], of = first[2 * nm + (2 - mi)];
          if (oc == 1 && of < (uint32_t)r) { later = 1; if (cnt[2 * nm + (mi - 1)] > 1) S.why = BLW_NAMES; }   // (two reads of this name and mate number would share the earlier mate)
          else if (oc > 1 && of < (uint32_t)r) S.why = BLW_NAMES;   // (several earlier mates: the general path sorts it out)
        }
      }
      isl[r] = later;
    }
    bl_scan32(isl, nr + 1, S);
    WG_LANE0 { S.npairs = S.scan_total; if (S.scan_total > 254u) S.why = BLW_PAIRS; }
    WG_SYNC();
    WG_FOR(r, nr) {
      if (!S.why && isl[r + 1] != isl[r]) {
        const uint32_t mi = RI_MATE(S.rinfo[r]), nm = B.name_rank[g0 + (uint32_t)r];
        const uint32_t q = first[2 * nm + (2 - mi)];
        S.pidx[r] = (uint8_t)isl[r]; S.prole[r] = 2;
        S.pidx[q] = (uint8_t)isl[r]; S.prole[q] = 1;               // (q is the earlier mate of exactly this read: its name has one read per mate number before r)
      }
    }
  }
  if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
  BLP(S, 4);
  if (C->debug_stop == 104u) { WG_LANE0 { H->why = 99; } return; }
  // ---- pass 1: every k-mer into the table (first occurrence kept), slot per occurrence to HBM
  LC_LDS uint32_t *tab = S.big;
  // (a batch with narrow hand-off areas holds at most PB_NCAP nodes per window: 8192 slots do, as in the 512-lane configuration)
  const uint32_t nslots = (PL.ncap > PB_NCAP || BL_SLOTS < 8192u) ? (uint32_t)BL_SLOTS : 8192u;
  WG_FOR(i, nslots) { tab[i] = BL_EMPTY; }
  WG_SYNC();
  // A lane takes a chunk of eight consecutive k-mer starts (bl_chunk) in two batches of BL_INS = 4: the batch's k-mers are shifted out of the
  // chunk's three LDS words, hashed and their first table words read before the first probe starts (a probe is a chain of dependent LDS
  // round trips).  A table word read early can be out of date by the time its probe looks at it: an empty word is then settled by the
  // compare-and-swap, an occupied one only ever changes to an earlier occurrence of the same k-mer.
  // (Measured and dropped in round 5: the four probes of a batch in lock step -- one loop that advances all four, their LDS reads in flight
  // together -- 2.62 against 2.36 workgroup-seconds per 32 768 windows: the loop runs until the slowest of 4 x 64 probes is through.)
  static_assert(BL_INS == 4, "two batches of four per chunk");
  const int NC = (int)S.cbase[R];                                 // (written before the last barrier)
  if (BL_KW == 1 || NW == 1) {       // one-word k-mers (k <= 31): the form below; else the general one behind it
  WG_FOR(_t, BL_WG) {
    for (int q = _t; q < NC; q += BL_WG) {
      BlChunk ch; bl_chunk(S, q, nr, reflen, K, ch);
      const int r = ch.r; const uint32_t rb = ch.boff0 - (uint32_t)ch.p0;          // LDS offset of the read's first base
      unsigned long long lo, hi; bl_chunk_bases(S.bases, (int)(ch.boff0 >> 3), lo, hi);
      for (int hb = 0; hb < 2; ++hb) {
        if (4 * hb >= ch.nv) break;
        uint32_t ix[BL_INS], fpv[BL_INS], cu[BL_INS], sw[BL_INS]; unsigned long long kv[BL_INS], al[BL_INS]; bool fF[BL_INS];
        BL_UNROLL for (int u = 0; u < BL_INS; ++u) kv[u] = (u ? ((lo >> (2 * u)) | (hi << (64 - 2 * u))) : lo) & kmask;
        lo = (lo >> 8) | (hi << 56); hi >>= 8;                        // (the next batch starts four bases on)
        BL_UNROLL for (int u = 0; u < BL_INS; ++u) {
          unsigned long long fw1;
          const unsigned long long ck = bl_canon2(kv[u], K, kmask, &fF[u], &fw1);
          // An earlier occurrence v2 (as it lies in LDS: first base in the low bits) is the same node iff it is this k-mer or its
          // reverse complement.  Read first-base-high, v2's reverse complement is ~v2 & mask (bl_canon); that equals this k-mer's
          // forward form fw1 iff v2 == ~fw1 & mask.  So a fingerprint hit is confirmed with one k-mer cut out of LDS and two
          // compares, without canonicalising the earlier occurrence (k is odd here: no k-mer is its own reverse complement).
          al[u] = (~fw1) & kmask;
          // table hash on 32-bit words (the table is private to this pass: node ids come from first-occurrence offsets, not slots)
          uint32_t hh = (uint32_t)ck * 0x9E3779B1u ^ (((uint32_t)(ck >> 32)) ^ ((uint32_t)ck >> 15)) * 0x85EBCA77u;
          hh ^= hh >> 15; hh *= 0x2C1B3C6Du; hh ^= hh >> 12;
          ix[u] = hh & (nslots - 1);
          uint32_t fp = hh >> BL_OFFBITS; if (fp == (0xFFFFFFFFu >> BL_OFFBITS)) fp -= 1u;
          fpv[u] = fp;
        }
        BL_UNROLL for (int u = 0; u < BL_INS; ++u) cu[u] = ld2(&tab[ix[u]]);
        BL_UNROLL for (int u = 0; u < BL_INS; ++u) {
          sw[u] = 0;
          if (4 * hb + u < ch.nv) {
            const uint32_t boff = ch.boff0 + (uint32_t)(4 * hb + u), fp = fpv[u]; const bool isF = fF[u]; const unsigned long long v1 = kv[u], alt = al[u];
            const uint32_t mine = (fp << BL_OFFBITS) | boff;
            // (the loop only finds the slot; what a hit has to do is done behind it -- every exit out of a loop body costs the compiled
            //  loop a set of EXEC-mask bookkeeping per trip)
            uint32_t idx = ix[u], cur = cu[u], probes = 0; bool hit = false; unsigned long long v2 = 0;
            while (true) {
              if (cur == BL_EMPTY) { cur = dev_atomic_cas32(&tab[idx], BL_EMPTY, mine); if (cur == BL_EMPTY) break; }
              if ((cur >> BL_OFFBITS) == fp) {
                v2 = bl_kmer(S.bases, cur & BL_OFFMASK, kmask);
                if (v2 == v1 || v2 == alt) { hit = true; break; }
              }
              idx = (idx + 1) & (nslots - 1);
              if (++probes > 256u) break;
              cur = ld2(&tab[idx]);
            }
            if (hit) {
              const bool f2 = (v2 == v1) ? isF : !isF;
              if (mine < cur) dev_atomic_min(&tab[idx], mine);
              // Scheduling hint (PreHdr::heavy).  All reads are in reference orientation and the loop over k only builds at a k
              // above the window reference's longest repeat, so a node normally meets its k-mer once per read and always in the
              // same orientation.  The same k-mer twice in one read (a duplication in the sample), or in both orientations (an
              // inverted repeat, which isRepeat does not look for), means a walk comes back to the node: if that node survives
              // removeLowCov the graph has a cycle and this k is rejected (Microassembler.cc:198-206).  The occurrence is noted
              // here and looked at again once the survivors are known.
              if (cur != mine && ((f2 != isF) || (r < nr && (cur & BL_OFFMASK) - rb < (uint32_t)ch.tlen))) {
                const uint32_t di = dev_atomic_add((LC_LDS uint32_t *)&S.ndup, 1u);
                if (di < BL_DUPCAP) X.dupo[di] = (bl_off_t)boff;
              }
            } else if (probes > 256u) S.why = BLW_TABLE;
            sw[u] = idx | (isF ? 0u : ON_ORI);
          }
        }
        // the batch's four words (table slot | orientation) in one store; a word past the read's last k-mer is 0
#if BL_WIDE
        { lc_u4 qv4; qv4.x = sw[0]; qv4.y = sw[1]; qv4.z = sw[2]; qv4.w = sw[3]; stg4((LC_GLOBAL uint32_t *)(X.occn + ch.boff0 + 4u * (uint32_t)hb), qv4); }
#else
        stg2((LC_GLOBAL uint32_t *)(X.occn + ch.boff0 + 4u * (uint32_t)hb), sw[0] | (sw[1] << 16), sw[2] | (sw[3] << 16));
#endif
      }
    }
  }
  }
#if BL_KW > 1
  else {
  // The general form (k-mers of NW <= BL_KW words): the same probe; only offsets, table positions and fingerprints are kept per occurrence in
  // flight -- on a fingerprint hit both k-mers are cut out of LDS again and compared word by word.
  WG_FOR(_t, BL_WG) {
    for (int q = _t; q < NC; q += BL_WG) {
      BlChunk ch; bl_chunk(S, q, nr, reflen, K, ch);
      const int r = ch.r; const uint32_t rb = ch.boff0 - (uint32_t)ch.p0;
      for (int hb = 0; hb < 2; ++hb) {
        if (4 * hb >= ch.nv) break;
        uint32_t ix[BL_INS], fpv[BL_INS], cu[BL_INS], sw[BL_INS]; bool fF[BL_INS];
        for (int u = 0; u < BL_INS; ++u) {
          const int jj = 4 * hb + u < ch.nv ? 4 * hb + u : ch.nv - 1;      // (clamped: a k-mer past the read's end is not cut out)
          BlKm v, ck, alt; bl_kmer_x(S.bases, ch.boff0 + (uint32_t)jj, NW, K, v); bl_canon_x(v, NW, K, ck, alt, &fF[u]);
          unsigned long long acc = ck.w[0];
          for (int i = 1; i < BL_KW; ++i) if (i < NW) acc = (acc ^ (acc >> 29)) * 0x9E3779B97F4A7C15ULL + ck.w[i];
          uint32_t hh = (uint32_t)acc * 0x9E3779B1u ^ (((uint32_t)(acc >> 32)) ^ ((uint32_t)acc >> 15)) * 0x85EBCA77u;
          hh ^= hh >> 15; hh *= 0x2C1B3C6Du; hh ^= hh >> 12;
          ix[u] = hh & (nslots - 1);
          uint32_t fp = hh >> BL_OFFBITS; if (fp == (0xFFFFFFFFu >> BL_OFFBITS)) fp -= 1u;
          fpv[u] = fp;
        }
        for (int u = 0; u < BL_INS; ++u) cu[u] = ld2(&tab[ix[u]]);
        for (int u = 0; u < BL_INS; ++u) {
          sw[u] = 0;
          if (4 * hb + u < ch.nv) {
          const uint32_t boff = ch.boff0 + (uint32_t)(4 * hb + u), fp = fpv[u]; const bool isF = fF[u];
          const uint32_t mine = (fp << BL_OFFBITS) | boff;
          uint32_t idx = ix[u], cur = cu[u], probes = 0;
          bool have = false; BlKm v1, alt;
          while (true) {
            if (cur == BL_EMPTY) { cur = dev_atomic_cas32(&tab[idx], BL_EMPTY, mine); if (cur == BL_EMPTY) break; }
            if ((cur >> BL_OFFBITS) == fp) {
              if (!have) { BlKm ck; bool f; bl_kmer_x(S.bases, boff, NW, K, v1); bl_canon_x(v1, NW, K, ck, alt, &f); have = true; }
              BlKm v2; bl_kmer_x(S.bases, cur & BL_OFFMASK, NW, K, v2);
              const bool same = bl_km_eq(v2, v1);
              if (same || bl_km_eq(v2, alt)) {
                const bool f2 = same ? isF : !isF;
                if (mine < cur) dev_atomic_min(&tab[idx], mine);
                if (cur != mine && ((f2 != isF) || (r < nr && (cur & BL_OFFMASK) - rb < (uint32_t)ch.tlen))) {     // (the hint: see the one-word form)
                  const uint32_t di = dev_atomic_add((LC_LDS uint32_t *)&S.ndup, 1u);
                  if (di < BL_DUPCAP) X.dupo[di] = (bl_off_t)boff;
                }
                break;
              }
            }
            idx = (idx + 1) & (nslots - 1);
            if (++probes > 256u) { S.why = BLW_TABLE; break; }
            cur = ld2(&tab[idx]);
          }
          sw[u] = idx | (isF ? 0u : ON_ORI);
          }
        }
#if BL_WIDE
        { lc_u4 q; q.x = sw[0]; q.y = sw[1]; q.z = sw[2]; q.w = sw[3]; stg4((LC_GLOBAL uint32_t *)(X.occn + ch.boff0 + 4u * (uint32_t)hb), q); }
#else
        stg2((LC_GLOBAL uint32_t *)(X.occn + ch.boff0 + 4u * (uint32_t)hb), sw[0] | (sw[1] << 16), sw[2] | (sw[3] << 16));
#endif
      }
    }
  }
  }
#endif
  if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
  BLP(S, 5);
  if (C->debug_stop == 105u) { WG_LANE0 { H->why = 99; } return; }
  // ---- node ids in first-insertion order: rank of the slot's first-occurrence offset among the occupied slots
  {
    // bitmap over the LDS offsets of first occurrences (cidx is idle here) + its popcount prefix per PAIR of words (t2c is idle too)
    constexpr int NBW = BL_BASES / 32 + 1, NPW = (NBW + 1) / 2;
    static_assert((size_t)(NBW + NPW) * 4 <= sizeof(S.cidx), "node-id bitmap + prefix in cidx");
    LC_LDS uint32_t *bm = (LC_LDS uint32_t *)S.cidx;
    LC_LDS uint32_t *pre = bm + NBW;
    WG_FOR(i, NBW) { bm[i] = 0; }
    WG_SYNC();
    WG_FOR(i, nslots) { const uint32_t e = tab[i]; if (e != BL_EMPTY) dev_atomic_or(&bm[(e & BL_OFFMASK) >> 5], 1u << (e & 31u)); }
    WG_SYNC();
    WG_FOR(i, NPW) { pre[i] = (uint32_t)dev_popc(bm[2 * i]) + (2 * i + 1 < NBW ? (uint32_t)dev_popc(bm[2 * i + 1]) : 0u); }
    bl_scan32(pre, NPW, S);
    WG_LANE0 { S.N = S.scan_total; if (S.scan_total > BL_NCAP || S.scan_total > PL.ncap) S.why = BLW_NODES; }
    if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
    WG_FOR(i, nslots) {
      const uint32_t e = tab[i];
      if (e != BL_EMPTY) {
        const uint32_t off = e & BL_OFFMASK, wd = off >> 5;
        const uint32_t id = pre[wd >> 1] + ((wd & 1u) ? (uint32_t)dev_popc(bm[wd - 1u]) : 0u) + (uint32_t)dev_popc(bm[wd] & ((1u << (off & 31u)) - 1u));
        tab[i] = id;                                             // slot -> node id, occurrence count in the upper half (below)
#if BL_WIDE
        X.idoff[id] = off;
#else
        S.idoff[id] = (uint16_t)off;
#endif
      }
    }
    WG_SYNC();
  }
  const uint32_t N = bl_bcast(&S.N);
  BLP(S, 6);
  if (C->debug_stop == 106u) { WG_LANE0 { H->why = 99; } return; }
  // ---- pass 2: slot -> node id per occurrence (HBM, streamed), occurrences per node
  // (the eight atomics of a chunk are issued together -- a word past the read's last k-mer adds 0 to slot 0 -- and waited for once)
  bl_for_chunk(S, X.occn, [&](const BlChunk &ch, const BlOccW &w) {
    uint32_t old[8];
    BL_UNROLL for (int j = 0; j < 8; ++j) {
      const bool act = j < ch.nv;
      const uint32_t e = bl_occw_get(w, j);
      old[j] = dev_atomic_add(&tab[act ? (e & (nslots - 1)) : 0u], act ? (1u << 16) : 0u);
    }
    BlOccW o = {};
    BL_UNROLL for (int j = 0; j < 8; ++j) {
      if (BL_BASES > 65535u && j < ch.nv && (old[j] >> 16) >= 0xFFFEu) S.why = BLW_SIZE;       // (a k-mer with 65 535 occurrences -- reads that are one long repeat, by the hundred -- would wrap the 16-bit counts: the general build's)
      bl_occw_set(o, j, j < ch.nv ? ((old[j] & 0xFFFFu) | (bl_occw_get(w, j) & ON_ORI)) : 0u);
    }
    bl_occw_store(X.occn, ch.boff0, o);
  });
  WG_SYNC();
  if (BL_BASES > 65535u) { if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; } }
  BLP(S, 7);
  if (C->debug_stop == 107u) { WG_LANE0 { H->why = 99; } return; }
  // ---- std::hash of every node's k-mer (libstdc++ table order), survivor bytes cleared
  {
    LC_GLOBAL unsigned long long *nhash = (LC_GLOBAL unsigned long long *)(area + PRE_OFF_NHASH);
    LC_GLOBAL uint8_t *surv = (LC_GLOBAL uint8_t *)(area + PRE_OFF_SURV);
    WG_FULL_BEGIN(n, act, N)                                     // (every lane hashes: wave.h WG_FULL_BEGIN)
      unsigned long long h;
#if BL_KW > 1
      if (NW > 1) {
        BlKm v, ck, alt; bool f; bl_kmer_x(S.bases, bl_idoff(S, X, (uint32_t)n), NW, K, v); bl_canon_x(v, NW, K, ck, alt, &f);
        h = bl_std_hash_km(ck, NW, K);
      } else
#endif
      { bool f; const unsigned long long ck = bl_canon(bl_kmer(S.bases, bl_idoff(S, X, (uint32_t)n), kmask), K, kmask, &f);
        h = std_hash_bytes([&](int j) -> int { return bl_acgt((int)((ck >> (2 * (K - 1 - j))) & 3ULL)); }, K); }
      if (act) { nhash[n] = h; surv[n] = 0; }
    WG_FULL_END
  }
  // ---- tracked nodes: those whose occurrence count leaves the first removeLowCov test open, and every node with a read AND
  //      a reference occurrence (their counts feed Ref_t::computeCoverage).  A node with one occurrence is decided: its
  //      counted occurrences are <= 1 (removeLowCov: mincovQV <= max(LOW_COV_THRESHOLD, MIN_COV_RATIO * avgcov)).
  const double avgcov = ((double)S.totalreadbp) / ((double)reflen);
  int tmin = 1;
  while ((tmin <= P->low_cov_threshold) || ((double)tmin <= (P->min_cov_ratio * avgcov))) ++tmin;
  const uint32_t tthr = tmin < 2 ? (uint32_t)tmin : 2u;
  WG_FOR(i, nslots) { const uint32_t e = tab[i]; if (e != BL_EMPTY) S.cidx[e & 0xFFFFu] = (uint16_t)(e >> 16); }     // count by node id
  WG_SYNC();
  {   // the hint's occurrences: only nodes with real coverage matter (a hairpin in one erroneous read survives removeLowCov
      // with coverage 2 and is trimmed as a tip later: k is not rejected for it)
    const uint32_t nd = S.ndup < BL_DUPCAP ? S.ndup : BL_DUPCAP;
    const uint32_t cthr = (uint32_t)(avgcov / 4.0) > 4u ? (uint32_t)(avgcov / 4.0) : 4u;
    WG_FOR(i, nd) { if ((uint32_t)S.cidx[X.occn[X.dupo[i]] & ON_ID] < cthr) X.dupo[i] = (bl_off_t)~(bl_off_t)0; }
  }
  {
    LC_LDS uint32_t *fl = S.big;                                 // (the table is no longer needed: occn holds node ids)
    WG_FOR(n, N + 1) { fl[n] = (n < (int)N && S.cidx[n] >= tthr) ? 1u : 0u; }
    bl_scan32(fl, (int)N + 1, S);
    WG_LANE0 { S.T = S.scan_total; if (S.scan_total > BL_TCAP) S.why = BLW_TRACKED; }
    if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
    WG_FOR(n, N) { S.cidx[n] = (fl[n + 1] != fl[n]) ? (uint16_t)fl[n] : (uint16_t)0xFFFFu; }
    WG_SYNC();
  }
  const uint32_t T = bl_bcast(&S.T);
  BLP(S, 8);
  if (C->debug_stop == 108u) { WG_LANE0 { H->why = 99; } return; }
  // ---- pass 3 (tracked nodes): counted occurrences per strand / sample, colours (Graph.cc:200-217), pair signatures
  //      S.big: cc[T] (u64) | sig[T * SW] (u64, earlier mates present, one bit per pair) ... todo[1024] | mk[64] at the end
  {
    const uint32_t npairs = bl_bcast(&S.npairs);
    const uint32_t SW = (npairs + 63u) / 64u;
    LC_LDS unsigned long long *cc = (LC_LDS unsigned long long *)S.big;
    LC_LDS unsigned long long *sig = cc + T;
    LC_LDS uint32_t *mk = S.big + BL_BIG / 4 - 64;                 // marked tracked nodes (hold a flagged occurrence)
    LC_LDS uint32_t *todo = mk - BL_FLAGCAP;                       // flagged occurrences: read << 10 | position
    WG_LANE0 { S.flagged = 0; if (2u * T * (1u + SW) > (uint32_t)(BL_BIG / 4 - 64 - BL_FLAGCAP)) S.why = BLW_PAIRS; }
    if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
    WG_FOR(t, T) { cc[t] = 0; S.t2c[t] = 0; }
    WG_FOR(t, T * SW) { sig[t] = 0; }
    WG_FOR(i, 64) { mk[i] = 0; }
    WG_SYNC();
    bl_for_chunk(S, X.occn, [&](const BlChunk &ch, const BlOccW &w) {
      const int r = ch.r;
      if (r == nr) return;                                         // the reference read: no colour, never counted (Graph.cc:265)
      const uint32_t ri = S.rinfo[r];
      const bool nml = RI_NML(ri) != 0;
      const unsigned long long inc = 1ULL << (16u * ((nml ? 2u : 0u) + (RI_REV(ri) ? 1u : 0u)));
      const int tlen = ch.tlen;
      const bool early = S.prole[r] == 1; const uint32_t px = S.pidx[r];
      // tumor reads: "the step's u and v both pass MIN_QUAL_CALL at every base" needs the bases s..s+K of the read, for step s = p or p - 1:
      // the mask bits from position p0 - 1 on (bit 0 = position p0 - 1; p0 = 0: bit 0 is not a base), 1 + 8 + K <= 40 of them for k <= 31
      const bool fastq = !nml && K <= 31;
      const unsigned long long gq = fastq ? (ch.p0 > 0 ? bl_good64(S.goodm, S.gwo[r], ch.p0 - 1) : (bl_good64(S.goodm, S.gwo[r], 0) << 1)) : 0ULL;
      const unsigned long long need = K <= 31 ? (1ULL << (K + 1)) - 1ULL : 0ULL;
      // eight look-ups in flight, then the atomics: a wave spends most of its time waiting for LDS round trips (profiles/r5_sq_counters.txt),
      // and one occurrence after the other is two dependent round trips each (node -> tracked index -> colour flags)
      uint32_t ti[8], t2[8];
      BL_UNROLL for (int j = 0; j < 8; ++j) ti[j] = S.cidx[bl_occw_get(w, j) & ON_ID];       // (a word past the read's last k-mer is 0: node 0)
      BL_UNROLL for (int j = 0; j < 8; ++j) { if (j >= ch.nv) ti[j] = 0xFFFFu; t2[j] = S.t2c[ti[j] != 0xFFFFu ? ti[j] : 0u]; }
      BL_UNROLL for (int j = 0; j < 8; ++j) {
        if (ti[j] == 0xFFFFu) continue;
        const int p = ch.p0 + j;
        dev_atomic_add64(&cc[ti[j]], inc);
        if (nml) { if (!(t2[j] & 2u)) dev_atomic_or((LC_LDS uint32_t *)&S.t2c[ti[j] & ~1u], (ti[j] & 1u) ? (2u << 16) : 2u); }
        else if (!(t2[j] & 1u)) {
          bool ok;
          if (fastq) ok = (p < tlen - K && ((gq >> (j + 1)) & need) == need) || (p - 1 >= 0 && p - 1 < tlen - K && ((gq >> j) & need) == need);
          else { const uint32_t gw = S.gwo[r];
                 ok = (p < tlen - K && bl_all_good(S.goodm, gw, p, p + K + 1)) || (p - 1 >= 0 && p - 1 < tlen - K && bl_all_good(S.goodm, gw, p - 1, p + K)); }
          if (ok) dev_atomic_or((LC_LDS uint32_t *)&S.t2c[ti[j] & ~1u], (ti[j] & 1u) ? (1u << 16) : 1u);
        }
        if (early) dev_atomic_or64(&sig[ti[j] * SW + (px >> 6)], 1ULL << (px & 63u));
      }
    });
    WG_SYNC();
    if (C->debug_stop == 120u) { WG_LANE0 { H->why = 99; } return; }
    // the later mates: an occurrence on a node that also holds one of the earlier mate can be an "overlapping mate"
    // (Node_t::hasOverlappingMate, src/Node.cc:638-661: a hit needs the name to BE in the other mate's vector); these are replayed
    if (npairs) bl_for_chunk(S, X.occn, [&](const BlChunk &ch, const BlOccW &w) {
      const int r = ch.r;
      if (S.prole[r] != 2) return;                                 // (only the later mate of a pair can meet its earlier mate's pushes)
      const uint32_t px = S.pidx[r];
      uint32_t ti[8]; unsigned long long sg[8];
      BL_UNROLL for (int j = 0; j < 8; ++j) ti[j] = S.cidx[bl_occw_get(w, j) & ON_ID];
      BL_UNROLL for (int j = 0; j < 8; ++j) { if (j >= ch.nv) ti[j] = 0xFFFFu; sg[j] = sig[(ti[j] != 0xFFFFu ? ti[j] : 0u) * SW + (px >> 6)]; }
      BL_UNROLL for (int j = 0; j < 8; ++j) {
        if (ti[j] != 0xFFFFu && ((sg[j] >> (px & 63u)) & 1ULL)) {
          const uint32_t at = dev_atomic_add((LC_LDS uint32_t *)&S.flagged, 1u);
          if (at < BL_FLAGCAP) todo[at] = ((uint32_t)r << 10) | (uint32_t)(ch.p0 + j);
          dev_atomic_or(&mk[ti[j] >> 5], 1u << (ti[j] & 31u));
        }
      }
    });
    if (C->debug_stop == 121u) { WG_LANE0 { H->why = 99; } return; }
    WG_SYNC();                                                     // (every wave's flagged occurrences are counted before lane 0 looks)
    WG_LANE0 { if (S.flagged > BL_FLAGCAP) { S.why = BLW_MATE; BL_DBG("[emu] window %d: %u flagged mate occurrences (%u)\n", w, S.flagged, (uint32_t)BL_FLAGCAP); } }
    WG_SYNC();
    const int why_f = lc_sgpr((int)S.why); const uint32_t nflag = lc_sgpr((uint32_t)S.flagged);
    WG_SYNC();
    if (why_f) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
    if (nflag) {
      // ---- exact replay (kernels.h build_csr): std::binary_search over the names the OTHER mate number pushed on the node before
      //      this read, in push order (unsorted: SURVEY.md H3).  The occurrences of the marked nodes are listed as
      //      node << 20 | read << 10 | position and sorted, which is the order loadSequence visited them in.
      LC_LDS uint32_t *list = S.big + 2 * T;                       // (sig is done with)
      const uint32_t lcap0 = (uint32_t)(BL_BIG / 4 - 64 - BL_FLAGCAP) - 2u * T;
      uint32_t lcap = 1; while (lcap * 2u <= lcap0 && lcap < (uint32_t)BL_BIG / 8u) lcap *= 2u;
      if (C->debug_stop == 131u) lcap = 512u;                       // (test knob: the ranges below on ordinary windows)
      // The list holds lcap entries; when the marked nodes have more occurrences than that, they are taken in ranges [t_lo, t_hi) of
      // tracked indices (a node's run is complete within its range), the range halved until its occurrences fit.
      uint32_t t_lo = 0;
      while (t_lo < T) {
      uint32_t t_hi = T, nl = 0;
      while (true) {
        WG_LANE0 { S.g0 = 0; }
        WG_FOR(i, lcap) { list[i] = 0xFFFFFFFFu; }
        WG_SYNC();
        bl_for_occ(S, X.occn, [&](int r, int p, uint32_t boff, uint32_t e) {
          (void)boff;
          if (r == nr) return;
          const uint32_t ti = S.cidx[e & ON_ID];
          if (ti == 0xFFFFu || ti < t_lo || ti >= t_hi || !((mk[ti >> 5] >> (ti & 31u)) & 1u)) return;
          const uint32_t mt = RI_MATE(S.rinfo[r]);
          if (mt != 1 && mt != 2) return;
          const uint32_t at = dev_atomic_add((LC_LDS uint32_t *)&S.g0, 1u);
          if (at < lcap) list[at] = (ti << 20) | ((uint32_t)r << 10) | (uint32_t)p;
        });
        WG_SYNC();                                                 // (the count is complete only once every wave is through: without this barrier
                                                                   //  lane 0 could pass the test on a partial count, and the sort below then ran
                                                                   //  over the flagged-occurrence list next to it -- found on the 1024-lane configuration)
        nl = bl_bcast(&S.g0);
        if (nl <= lcap) break;
        if (t_hi - t_lo <= 1u) {                                    // one node with more occurrences than the list holds
          WG_LANE0 { S.why = BLW_MATE; BL_DBG("[emu] window %d: %u occurrences on one marked node (%u), %u tracked\n", w, S.g0, lcap, T); }
          break;
        }
        t_hi = t_lo + (t_hi - t_lo) / 2u;
      }
      if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
      if (C->debug_stop == 123u) { WG_LANE0 { H->why = 99; } return; }
      uint32_t n2 = 1; while (n2 < nl) n2 *= 2u;
      for (uint32_t kk = 2; kk <= n2; kk <<= 1)                    // bitonic sort, ascending (the padding sorts last)
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
          WG_FOR(i, n2) {
            const uint32_t l = (uint32_t)i ^ j;
            if (l > (uint32_t)i) {
              const uint32_t a = list[i], b2 = list[l];
              const bool up = (((uint32_t)i & kk) == 0);
              if ((a > b2) == up) { list[i] = b2; list[l] = a; }
            }
          }
          WG_SYNC();
        }
      if (C->debug_stop == 124u) { WG_LANE0 { H->why = 99; } return; }
      WG_FOR(fi, nflag) {
        const uint32_t r = todo[fi] >> 10, p = todo[fi] & 1023u;
        const uint32_t boff = 16u * S.rdo[r] + p;
        const uint32_t e = X.occn[boff];
        const uint32_t ti = S.cidx[e & ON_ID];
        if (ti < t_lo || ti >= t_hi) continue;
        const uint32_t ri = S.rinfo[r];
        const uint32_t mi = RI_MATE(ri), nm = B.name_rank[g0 + r] & 0xFFFFu;
        uint32_t lo = 0, len = nl;                                   // start of the node's run
        while (len > 0) { const uint32_t h = len >> 1; if ((list[lo + h] >> 20) < ti) { lo += h + 1; len -= h + 1; } else len = h; }
        // the other mate's pushes of reads before r, in order: one per step the occurrence takes part in (as v of step p-1, as u of step p)
        auto pushes_of = [&](uint32_t ent) -> uint32_t {
          const uint32_t er = (ent >> 10) & 1023u, ep = ent & 1023u;
          const uint32_t eri = S.rinfo[er];
          if (RI_MATE(eri) != 3u - mi) return 0u;
          const int etl = (int)RI_TLEN(eri);
          return (ep >= 1u ? 1u : 0u) + ((int)ep <= etl - K - 1 ? 1u : 0u);
        };
        uint32_t total = 0;
        for (uint32_t i = lo; i < nl && (list[i] >> 20) == ti && ((list[i] >> 10) & 1023u) < r; ++i) total += pushes_of(list[i]);
        auto name_at = [&](uint32_t idx) -> uint32_t {               // name rank of push number idx of that vector
          uint32_t acc = 0;
          for (uint32_t i = lo; i < nl; ++i) { const uint32_t pc = pushes_of(list[i]); if (idx < acc + pc) return B.name_rank[g0 + ((list[i] >> 10) & 1023u)] & 0xFFFFu; acc += pc; }
          return 0xFFFFFFFFu;
        };
        uint32_t first = 0, l2 = total;                              // std::lower_bound over the names, as pushed
        while (l2 > 0) { const uint32_t h = l2 >> 1, mid = first + h; if (name_at(mid) < nm) { first = mid + 1; l2 = l2 - h - 1; } else l2 = h; }
        const bool ovl = (first != total) && !(nm < name_at(first));
        if (ovl) {                                                   // do not update coverage for overlapping mates (Graph.cc:267-271)
          BL_DBG("[emu] window %d: overlapping mate, read %u position %u (nodes %u..%u of %u)\n", w, r, p, t_lo, t_hi, T);
          const uint32_t cls = (RI_NML(ri) ? 2u : 0u) + (RI_REV(ri) ? 1u : 0u);
          dev_atomic_add64(&cc[ti], 0ULL - (1ULL << (16 * cls)));
          X.occn[boff] = (bl_on_t)(e | ON_OVL);
        }
      }
      WG_SYNC();
      t_lo = t_hi;
      }
    }
    if (C->debug_stop == 122u) { WG_LANE0 { H->why = 99; } return; }
    WG_FOR(t, T) { X.tcc[t] = cc[t]; X.tfl[t] = ((S.t2c[t] & 1u) ? NF_TUMOR : 0u) | ((S.t2c[t] & 2u) ? NF_NORMAL : 0u); }
    WG_SYNC();
  }
  BLP(S, 9);
  if (C->debug_stop == 109u) { WG_LANE0 { H->why = 99; } return; }
  // ---- candidates: tracked nodes the count-based predicate leaves undecided (kernels.h build_gather: `low`), in node order
  {
    LC_LDS uint32_t *fl = S.big + 2 * BL_TCAP;                   // (cc still in the first 2 T words of S.big)
    LC_LDS const unsigned long long *cc = (LC_LDS const unsigned long long *)S.big;
    WG_FOR(t, T + 1) {
      uint32_t cand = 0;
      if (t < (int)T) {
        const unsigned long long c4 = cc[t];
        const uint32_t c0 = (uint32_t)(c4 & 0xFFFFu), c1 = (uint32_t)((c4 >> 16) & 0xFFFFu), c2 = (uint32_t)((c4 >> 32) & 0xFFFFu), c3 = (uint32_t)(c4 >> 48);
        const uint32_t counted = c0 + c1 + c2 + c3;
        const float tt = (float)c0 + (float)c1, tn = (float)c2 + (float)c3;
        const bool low = ((int)counted <= P->low_cov_threshold) || ((double)counted <= (P->min_cov_ratio * avgcov)) || (tt == 1.0f && tn == 1.0f);
        cand = low ? 0u : 1u;
      }
      fl[t] = cand;
    }
    bl_scan32(fl, (int)T + 1, S);
    WG_LANE0 { S.ncand = S.scan_total; if (S.scan_total > PB_CCAP) S.why = BLW_CAND; else if ((size_t)S.scan_total * (size_t)K > (size_t)PL.qvcap) S.why = BLW_QV; }
    if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
    WG_FOR(t, T) { S.t2c[t] = (fl[t + 1] != fl[t]) ? (uint16_t)fl[t] : (uint16_t)0xFFFFu; }
    WG_SYNC();
    WG_FOR(n, N) { const uint32_t ti = S.cidx[n]; if (ti != 0xFFFFu) { const uint32_t ci = S.t2c[ti]; if (ci != 0xFFFFu) { X.c_id[ci] = (uint32_t)n; X.c_ti[ci] = ti; } } }
    WG_SYNC();
  }
  const uint32_t ncand = bl_bcast(&S.ncand);
  // ---- --linked-reads: the occurrences of the tracked nodes leave the workgroup as well, as csr runs by node id (layout.h PRE_OFF_LRNOCC /
  //      PRE_OFF_LRCSR; entries = the window kernel's cs_t words: read, position, orientation, state 2 = not counted -- the reference pseudo-read,
  //      an overlapping mate).  Barcode and haplotype bookkeeping (Node_t::addBX / addHP / hasBX, reference src/Graph.cc:239-263,
  //      src/Node.cc:30-118) is a replay over a node's occurrences in visiting order: the window kernel does it (kernels.h load_prebuilt_lr)
  //      over these runs instead of building the whole window in HBM to get at them.  What this kernel computes does not depend on barcodes:
  //      the four per-position counters, the first removeLowCov (on mincovQV of those), edges, table order, components.  A node with
  //      one occurrence is not tracked and has no run: it is removed and no reference k-mer with a read on it (those are tracked).
  if (P->lr_mode) {
    LC_GLOBAL uint32_t *lrnocc = (LC_GLOBAL uint32_t *)(area + PRE_OFF_LRNOCC);
    LC_GLOBAL uint32_t *lrcsr = (LC_GLOBAL uint32_t *)(area + PRE_OFF_LRCSR);
    LC_LDS uint32_t *cur = S.big;                                  // [N + 1] occurrences per node -> run starts -> fill cursors
    static_assert(4u * (BL_NCAP + 2u) <= (uint32_t)BL_BIG, "run starts of every node in the phase area");
    WG_FOR(n, N + 1) { cur[n] = 0; }
    WG_SYNC();
    bl_for_chunk(S, X.occn, [&](const BlChunk &ch, const BlOccW &w) {
      uint32_t nd[8];
      BL_UNROLL for (int j = 0; j < 8; ++j) nd[j] = bl_occw_get(w, j) & ON_ID;
      BL_UNROLL for (int j = 0; j < 8; ++j) { if (j < ch.nv && S.cidx[nd[j]] != 0xFFFFu) dev_atomic_add(&cur[nd[j]], 1u); }
    });
    bl_scan32(cur, (int)N + 1, S);
    WG_LANE0 { if (PL.lrcap == 0u || S.scan_total > PL.lrcap) S.why = BLW_SIZE; }
    if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
    WG_FOR(n, N + 1) { lrnocc[n] = cur[n]; }
    WG_SYNC();
    bl_for_chunk(S, X.occn, [&](const BlChunk &ch, const BlOccW &w) {
      const uint32_t st_r = ch.r == nr ? 2u : 0u;                   // (the reference pseudo-read never counts: Graph.cc:265)
      uint32_t ev[8], at[8]; bool tk[8];
      BL_UNROLL for (int j = 0; j < 8; ++j) { ev[j] = bl_occw_get(w, j); tk[j] = j < ch.nv && S.cidx[ev[j] & ON_ID] != 0xFFFFu; }
      BL_UNROLL for (int j = 0; j < 8; ++j) at[j] = tk[j] ? dev_atomic_add(&cur[ev[j] & ON_ID], 1u) : 0u;
      BL_UNROLL for (int j = 0; j < 8; ++j) {
        if (!tk[j]) continue;
        const uint32_t st = st_r ? st_r : ((ev[j] & ON_OVL) ? 2u : 0u);
        lrcsr[at[j]] = (uint32_t)ch.r | ((uint32_t)(ch.p0 + j) << 16) | (((ev[j] >> ON_ORISH) & 1u) << 26) | (st << 27);
      }
    });
    WG_SYNC();
    WG_LANE0 { H->lr = 1u; H->lr_total = S.scan_total; }
  }
  BLP(S, 10);
  if (C->debug_stop == 110u) { WG_LANE0 { H->why = 99; } return; }
  // ---- per-position quality counts of the candidates (Node_t::updateCovDistr minqv_fwd / minqv_rev, src/Node.cc:470-497) as
  //      counted - (occurrences whose base at that k-mer position is below MIN_QUAL_CALL); groups of candidates that fit LDS
  {
    LC_GLOBAL uint16_t *qv = (LC_GLOBAL uint16_t *)(area + PRE_OFF_QV);
    // Counters per (candidate, k-mer position): four classes.  When no candidate has more than 255 counted occurrences in a class
    // (the usual case below ~100x) they are bytes of one 32-bit word -- twice as many candidates per group, i.e. fewer passes
    // over the window's occurrences, and 32-bit LDS atomics -- else 16-bit fields of a 64-bit word.
    WG_LANE0 { S.g0 = 0; }
    WG_SYNC();
    WG_FOR(ci, ncand) {
      const unsigned long long c4 = X.tcc[X.c_ti[ci]];
      if ((c4 & 0xFF00FF00FF00FF00ULL) != 0ULL) S.g0 = 1;
    }
    const bool wide = bl_bcast(&S.g0) != 0 || C->debug_stop == 130u;      // (130: test knob, the 16-bit form on any window)
    const uint32_t esz = wide ? 8u : 4u;
    const uint32_t gmax = ((uint32_t)BL_BIG - 8u) / ((uint32_t)K * esz + 8u);
    LC_LDS unsigned long long *bad64 = (LC_LDS unsigned long long *)S.big;      // wide: [group][K] four 16-bit counters
    LC_LDS uint32_t *bad32 = (LC_LDS uint32_t *)S.big;                          // else: [group][K] four 8-bit counters
    LC_LDS unsigned long long *gcc = (LC_LDS unsigned long long *)S.big + ((size_t)gmax * K * esz + 7u) / 8u;   // [group] the candidates' counted occurrences
    for (uint32_t c0 = 0; c0 < ncand; c0 += gmax) {
      const uint32_t c1 = c0 + gmax < ncand ? c0 + gmax : ncand;
      if (wide) { WG_FOR(i, (c1 - c0) * (uint32_t)K) { bad64[i] = 0; } } else { WG_FOR(i, (c1 - c0) * (uint32_t)K) { bad32[i] = 0; } }
      WG_FOR(i, c1 - c0) { gcc[i] = X.tcc[X.c_ti[c0 + (uint32_t)i]]; }
      WG_SYNC();
      // one occurrence of candidate ci: the bases of its k-mer below MIN_QUAL_CALL
      auto count_occ = [&](int r, int p, uint32_t ci, bool rev) {
        const uint32_t ri = S.rinfo[r];
        const uint32_t cls = (RI_NML(ri) ? 2u : 0u) + (RI_REV(ri) ? 1u : 0u);
        const uint32_t gw = S.gwo[r];
        // bits p .. p+K-1 of the read's mask; a clear bit at read position p + j is k-mer position j (forward) or K-1-j (reverse)
        for (int j0 = 0; j0 < K;) {
          const int pos = p + j0, wv = pos >> 5, lo = pos & 31;
          int take = 32 - lo; if (take > K - j0) take = K - j0;
          uint32_t m = ~(S.goodm[gw + wv] >> lo);
          if (take < 32) m &= (1u << take) - 1u;
          while (m) {
            const int j = j0 + (int)__builtin_ctz(m); m &= m - 1u;
            const int i = rev ? K - 1 - j : j;
            if (wide) dev_atomic_add64(&bad64[(ci - c0) * (uint32_t)K + (uint32_t)i], 1ULL << (16 * cls));
            else dev_atomic_add(&bad32[(ci - c0) * (uint32_t)K + (uint32_t)i], 1u << (8 * cls));
          }
          j0 += take;
        }
      };
      // The first group walks every occurrence of the window and notes those of the candidates that did not fit (usually a few dozen
      // of ~600); the later groups work that list off instead of walking all occurrences again.
      const bool listed = c0 > 0 && bl_bcast(&S.g1) <= BL_PQCAP;
      if (c0 == 0) { WG_LANE0 { S.g1 = 0; } WG_SYNC(); }
      if (listed) {
        const uint32_t nq = S.g1;
        WG_FOR(i, nq) {
          const uint32_t v = X.pq[i], ci = (v >> 20) & 0x7FFu;
          if (ci >= c0 && ci < c1) count_occ((int)(v & 0x3FFu), (int)((v >> 10) & 0x3FFu), ci, (v >> 31) != 0);
        }
      } else
      bl_for_chunk(S, X.occn, [&](const BlChunk &ch, const BlOccW &w) {
        const int r = ch.r;
        if (r == nr) return;
        const uint32_t ri = S.rinfo[r];
        const uint32_t cls = (RI_NML(ri) ? 2u : 0u) + (RI_REV(ri) ? 1u : 0u);
        // the bases of the chunk's k-mers below MIN_QUAL_CALL: bit i = read position p0 + i (8 + K - 1 <= 38 of them for k <= 31)
        const bool fastq = K <= 31;
        const unsigned long long bq = fastq ? ~bl_good64(S.goodm, S.gwo[r], ch.p0) : 0ULL;
        const uint32_t km = fastq ? (uint32_t)((1ULL << K) - 1ULL) : 0u;
        // eight look-ups in flight (node -> tracked index, then -> candidate); an occurrence none of whose bases is below the threshold has
        // nothing to count and is not looked up at all (k = 13 at 7 % low-quality bases: 4 of 10)
        uint32_t ti[8], ci[8], mm[8];
        BL_UNROLL for (int j = 0; j < 8; ++j) {
          const uint32_t e = bl_occw_get(w, j);
          mm[j] = fastq ? ((uint32_t)(bq >> j) & km) : 1u;             // bit i: base i of this k-mer (as the read has it) is below the threshold
          if (j >= ch.nv || (e & ON_OVL)) mm[j] = 0;                   // (an overlapping mate's occurrence: not counted)
          ti[j] = S.cidx[e & ON_ID];
        }
        BL_UNROLL for (int j = 0; j < 8; ++j) { if (!mm[j]) ti[j] = 0xFFFFu; ci[j] = S.t2c[ti[j] != 0xFFFFu ? ti[j] : 0u]; }
        BL_UNROLL for (int j = 0; j < 8; ++j) {
          if (ti[j] == 0xFFFFu || ci[j] == 0xFFFFu) continue;         // not tracked / not a candidate
          const int p = ch.p0 + j; const uint32_t e = bl_occw_get(w, j);
          if (ci[j] >= c1 && c0 == 0) {                                // a later group's: note it
            if (p > 1023 || r > 1023) { S.g1 = BL_PQCAP + 1u; continue; }     // (does not fit an entry: the later groups walk everything)
            const uint32_t at = dev_atomic_add((LC_LDS uint32_t *)&S.g1, 1u);
            if (at < BL_PQCAP) X.pq[at] = (uint32_t)r | ((uint32_t)p << 10) | (ci[j] << 20) | ((e & ON_ORI) ? 0x80000000u : 0u);
            continue;
          }
          if (ci[j] < c0 || ci[j] >= c1) continue;
          if (!fastq) { count_occ(r, p, ci[j], (e & ON_ORI) != 0); continue; }
          uint32_t m = mm[j];
          const uint32_t base = (ci[j] - c0) * (uint32_t)K; const bool rev = (e & ON_ORI) != 0;
          while (m) {
            const int jb = (int)__builtin_ctz(m); m &= m - 1u;
            const uint32_t i = (uint32_t)(rev ? K - 1 - jb : jb);
            if (wide) dev_atomic_add64(&bad64[base + i], 1ULL << (16 * cls));
            else dev_atomic_add(&bad32[base + i], 1u << (8 * cls));
          }
        }
      });
      WG_SYNC();
      auto bad_of = [&](uint32_t t, int cl) -> uint32_t { return wide ? (uint32_t)((bad64[t] >> (16 * cl)) & 0xFFFFu) : ((bad32[t] >> (8 * cl)) & 0xFFu); };
      WG_FOR(t, (c1 - c0) * (uint32_t)K) {
        const uint32_t ci = c0 + (uint32_t)t / (uint32_t)K;
        const unsigned long long c4 = gcc[(uint32_t)t / (uint32_t)K];
        LC_GLOBAL uint16_t *q = qv + ((size_t)ci * K + ((uint32_t)t % (uint32_t)K)) * 4;
        for (int cl = 0; cl < 4; ++cl) q[cl] = (uint16_t)(((c4 >> (16 * cl)) & 0xFFFFu) - bad_of((uint32_t)t, cl));
      }
      WG_FOR(cc_, c1 - c0) {                                          // mincovQV of the candidate
        const uint32_t ci = c0 + (uint32_t)cc_;
        const unsigned long long c4 = gcc[cc_];
        uint32_t mn = 0x7FFFFFFFu;
        for (int i = 0; i < K; ++i) {
          uint32_t sq = 0;
          for (int cl = 0; cl < 4; ++cl) sq += (uint32_t)((c4 >> (16 * cl)) & 0xFFFFu) - bad_of((uint32_t)cc_ * (uint32_t)K + (uint32_t)i, cl);
          if (sq < mn) mn = sq;
        }
        X.c_minqv[ci] = mn;
      }
      WG_SYNC();
    }
  }
  BLP(S, 11);
  if (C->debug_stop == 111u) { WG_LANE0 { H->why = 99; } return; }
  // ---- survivors of the first removeLowCov (the predicate on mincovQV), dense in node order
  {
    LC_LDS uint32_t *fl = S.big;
    WG_FOR(ci, ncand + 1) {
      uint32_t sv = 0;
      if (ci < (int)ncand) {
        const unsigned long long c4 = X.tcc[X.c_ti[ci]];
        const float tt = (float)(uint32_t)(c4 & 0xFFFFu) + (float)(uint32_t)((c4 >> 16) & 0xFFFFu), tn = (float)(uint32_t)((c4 >> 32) & 0xFFFFu) + (float)(uint32_t)(c4 >> 48);
        const int minqv = (int)X.c_minqv[ci];
        const bool low = (minqv <= P->low_cov_threshold) || ((double)minqv <= (P->min_cov_ratio * avgcov)) || (tt == 1.0f && tn == 1.0f);
        sv = low ? 0u : 1u;
      }
      fl[ci] = sv;
    }
    bl_scan32(fl, (int)ncand + 1, S);
    WG_LANE0 { S.nsurv = S.scan_total; if (S.scan_total > PB_SCAP) S.why = BLW_SURV; }
    if (bl_bcast(&S.why)) { WG_LANE0 { H->why = (uint32_t)S.why; } return; }
    LC_GLOBAL uint32_t *snode = (LC_GLOBAL uint32_t *)(area + PRE_OFF_SNODE);
    LC_GLOBAL unsigned long long *skey = (LC_GLOBAL unsigned long long *)(area + PRE_OFF_SKEY);
    LC_GLOBAL uint32_t *sid = (LC_GLOBAL uint32_t *)(area + PRE_OFF_SID);
    LC_GLOBAL uint8_t *surv = (LC_GLOBAL uint8_t *)(area + PRE_OFF_SURV);
    WG_FOR(ci, ncand) {
      const bool sv = fl[ci + 1] != fl[ci];
      const uint32_t n = X.c_id[ci];
      snode[ci] = sv ? n : LC_NIL;
      if (sv) {
        const uint32_t si = fl[ci];
#if BL_KW > 1
        if (NW > 1) {
        BlKm v, ck, alt; bool f; bl_kmer_x(S.bases, bl_idoff(S, X, (uint32_t)n), NW, K, v); bl_canon_x(v, NW, K, ck, alt, &f);
        for (int i = 0; i < BL_KW; ++i) if (i < NW) skey[(size_t)ci * PL.kw + (uint32_t)i] = ck.w[i];
        } else
#endif
        { bool f; skey[(size_t)ci * PL.kw] = bl_canon(bl_kmer(S.bases, bl_idoff(S, X, (uint32_t)n), kmask), K, kmask, &f); }
        sid[si] = n; surv[n] = 1; X.s_ci[si] = (uint32_t)ci;
      }
    }
    WG_SYNC();
  }
  const uint32_t nsurv = bl_bcast(&S.nsurv);
  BLP(S, 12);
  if (C->debug_stop == 112u) { WG_LANE0 { H->why = 99; } return; }
  // ---- Ref_t::mertable membership, Ref_t::computeCoverage (src/Ref.cc:40-64, 173-250) from the counts of the tracked nodes, the
  //      reference pseudo-read's node per offset.  First build of the window: Ref_t::seq is still the whole rawseq.
  {
    LC_GLOBAL uint32_t *occ_ref = (LC_GLOBAL uint32_t *)(area + PRE_OFF_OCCREF);
    LC_GLOBAL uint16_t *refcov = (LC_GLOBAL uint16_t *)(area + PRE_OFF_REFCOV);
    LC_LDS uint32_t *inmer = S.big;                                // bitmap over node ids
    WG_FOR(i, BL_NCAP / 32) { inmer[i] = 0; }
    WG_FOR(j, reflen) { for (int q = 0; q < 4; ++q) refcov[4 * j + q] = 0; }
    WG_SYNC();
    const int nrefk = reflen - K > 0 ? reflen - K + 1 : 0;
    const uint32_t rb = 16u * S.rdo[nr];
    WG_FOR(i, reflen - K > 0 ? reflen - K : 0) {                   // i + K < seq.length()
      const uint32_t n = X.occn[rb + (uint32_t)i] & ON_ID;
      dev_atomic_or(&inmer[n >> 5], 1u << (n & 31u));
    }
    WG_SYNC();
    WG_FOR(i, reflen - K > 0 ? reflen - K : 0) {                   // i + K < rawseq.length(): every one of them is in the table here
      const uint32_t n = X.occn[rb + (uint32_t)i] & ON_ID;
      const uint32_t ti = S.cidx[n];
      unsigned long long c4 = 0;
      if (ti != 0xFFFFu) c4 = X.tcc[ti];
      uint16_t v[4]; for (int q = 0; q < 4; ++q) v[q] = (uint16_t)((c4 >> (16 * q)) & 0xFFFFu);
      if (i == 0) { for (int j = 0; j < K; ++j) for (int q = 0; q < 4; ++q) refcov[4 * j + q] = v[q]; }
      else { for (int q = 0; q < 4; ++q) refcov[4 * (i + K - 1) + q] = v[q]; }
    }
    WG_LANE0 { S.refn = 0; }
    WG_SYNC();
    WG_FOR(i, BL_NCAP / 32) { const uint32_t m = inmer[i]; if (m) dev_atomic_add((LC_LDS uint32_t *)&S.refn, (uint32_t)dev_popc(m)); }
    // survivor index per node from here on (the tracked index is in c_ti for every candidate)
    WG_SYNC();
    LC_LDS uint32_t *inm2 = S.big + BL_NCAP / 32;                  // INMER per survivor: keep a copy of the bitmap while cidx changes meaning
    (void)inm2;
    WG_FOR(n, N) { S.cidx[n] = 0xFFFFu; }
    WG_SYNC();
    WG_FOR(si, nsurv) { S.cidx[X.c_id[X.s_ci[si]]] = (uint16_t)si; }
    WG_SYNC();
    {   // the hint: does a node that was met twice in a read / in both orientations survive?
      const uint32_t nd = S.ndup < BL_DUPCAP ? S.ndup : BL_DUPCAP;
      WG_FOR(i, nd) { const bl_off_t o = X.dupo[i]; if (o != (bl_off_t)~(bl_off_t)0 && S.cidx[X.occn[o] & ON_ID] != 0xFFFFu) S.hint = 1; }
      WG_LANE0 { if (S.ndup > BL_DUPCAP) S.hint = 1; }
    }
    WG_SYNC();
    WG_FOR(i, nrefk) {
      const uint32_t e = X.occn[rb + (uint32_t)i];
      const uint32_t n = e & ON_ID;
      occ_ref[i] = n | (S.cidx[n] != 0xFFFFu ? 0u : PB_GONE) | ((e & ON_ORI) ? 0x80000000u : 0u);
    }
  }
  BLP(S, 13);
  if (C->debug_stop == 113u) { WG_LANE0 { H->why = 99; } return; }
  // ---- trace only: edge count of every node before the filter (printStats over the whole table): distinct (side, base) slots
  if (C->evt_cap) {
    LC_LDS uint32_t *msk = S.big + BL_NCAP / 32;                   // one byte per node, four nodes per word
    WG_FOR(i, BL_NCAP / 4) { msk[i] = 0; }
    WG_LANE0 { S.edges_total = 0; }
    WG_SYNC();
    bl_for_occ(S, X.occn, [&](int r, int p, uint32_t boff, uint32_t e) {
      const uint32_t n = e & ON_ID, ori = (e >> ON_ORISH) & 1u;
      const int tlen = r < nr ? (int)RI_TLEN(S.rinfo[r]) : reflen;
      const int nk = tlen - K + 1;
      uint32_t m = 0;
      if (p + 1 < nk) { const int b = bl_base(S.bases, boff + (uint32_t)K); m |= 1u << ((ori == 0 ? 0 : 4) + (ori == 0 ? b : 3 - b)); }
      if (p > 0) { const int b = bl_base(S.bases, boff - 1u); m |= 1u << ((ori == 0 ? 4 : 0) + (ori == 0 ? b : 3 - b)); }
      if (m) dev_atomic_or(&msk[n >> 2], m << (8u * (n & 3u)));
    });
    WG_SYNC();
    WG_FOR(i, (N + 3) / 4) { const uint32_t m = msk[i]; if (m) dev_atomic_add((LC_LDS uint32_t *)&S.edges_total, (uint32_t)dev_popc(m)); }
    WG_SYNC();
  }
  BLP(S, 14);
  if (C->debug_stop == 114u) { WG_LANE0 { H->why = 99; } return; }
  // ---- edges of the survivors in first-seen order (Node_t::addEdge order over the reads, Graph.cc:320-347): earliest step per
  //      (side, extension base) slot with LDS atomicMin, groups of survivors that fit; stamp = 2 * offset of the step's u (+1 for
  //      the v side), offsets grow with (read, position) like the occurrence index of kernels.h
  {
    LC_LDS uint32_t *inmer = S.big;                                // (kept: BL_NCAP / 32 words)
    LC_LDS uint32_t *E = S.big + BL_NCAP / 32 + BL_NCAP / 4;       // [group][8]
    const uint32_t gmax = (BL_BIG / 4u - BL_NCAP / 32u - BL_NCAP / 4u) / 8u;
    LC_GLOBAL NodeGr *pgr = (LC_GLOBAL NodeGr *)(area + PRE_OFF_PGR);
    for (uint32_t s0 = 0; s0 < nsurv; s0 += gmax) {
      const uint32_t s1 = s0 + gmax < nsurv ? s0 + gmax : nsurv;
      WG_FOR(i, (s1 - s0) * 8u) { E[i] = LC_NIL; }
      WG_SYNC();
      bl_for_chunk(S, X.occn, [&](const BlChunk &ch, const BlOccW &w) {
        const int nk = ch.tlen - K + 1;
        // the extension bases out of the chunk's own words (k <= 31): base i of the chunk at bits 2i of (lo, hi); the base in front of the
        // chunk's first k-mer is read on its own
        const bool fastb = K <= 31;
        unsigned long long lo = 0, hi = 0;
        if (fastb) bl_chunk_bases(S.bases, (int)(ch.boff0 >> 3), lo, hi);
        uint32_t sv[8];
        BL_UNROLL for (int j = 0; j < 8; ++j) sv[j] = S.cidx[bl_occw_get(w, j) & ON_ID];
        BL_UNROLL for (int j = 0; j < 8; ++j) {
          const uint32_t si = sv[j];
          if (j >= ch.nv || si < s0 || si >= s1) continue;
          const int p = ch.p0 + j; const uint32_t boff = ch.boff0 + (uint32_t)j;
          const uint32_t ori = (bl_occw_get(w, j) >> ON_ORISH) & 1u;
          if (p + 1 < nk) {                                            // step p: this node is u
            const int i = j + K;
            const int b = fastb ? (int)((i < 32 ? (lo >> (2 * i)) : (hi >> (2 * i - 64))) & 3ULL) : bl_base(S.bases, boff + (uint32_t)K);
            const uint32_t sl = (ori == 0 ? 0u : 4u) + (uint32_t)(ori == 0 ? b : 3 - b);
            dev_atomic_min(&E[(si - s0) * 8u + sl], 2u * boff);
          }
          if (p > 0) {                                                 // step p-1: this node is v
            const int b = (fastb && j > 0) ? (int)((lo >> (2 * (j - 1))) & 3ULL) : bl_base(S.bases, boff - 1u);
            const uint32_t sl = (ori == 0 ? 4u : 0u) + (uint32_t)(ori == 0 ? b : 3 - b);
            dev_atomic_min(&E[(si - s0) * 8u + sl], 2u * (boff - 1u) + 1u);
          }
        }
      });
      WG_SYNC();
      WG_FOR(sg, s1 - s0) {                                          // one lane per survivor: its record
        const uint32_t si = s0 + (uint32_t)sg, ci = X.s_ci[si], n = X.c_id[ci], ti = X.c_ti[ci];
        uint32_t stamp[8]; int ne = 0;
        for (int j = 0; j < 8; ++j) { const uint32_t v = E[(uint32_t)sg * 8u + (uint32_t)j]; if (v != LC_NIL) stamp[ne++] = v; }
        for (int i = 1; i < ne; ++i) { const uint32_t s = stamp[i]; int j = i; while (j > 0 && stamp[j - 1] > s) { stamp[j] = stamp[j - 1]; --j; } stamp[j] = s; }
        LC_GLOBAL NodeGr &G = pgr[si];
        int m = 0;
        for (int i = 0; i < ne; ++i) {
          const uint32_t s = stamp[i] >> 1;
          const uint32_t a = X.occn[s], b = X.occn[s + 1];           // u and v of that step
          const uint32_t ua = (a >> ON_ORISH) & 1u, ub = (b >> ON_ORISH) & 1u;
          uint32_t to, dir;
          if ((stamp[i] & 1u) == 0) { to = b & ON_ID; dir = ua == 0 ? (ub == 0 ? 0u : 1u) : (ub == 0 ? 2u : 3u); }   // FF FR RF RR
          else { to = a & ON_ID; dir = ua == 0 ? (ub == 0 ? 3u : 1u) : (ub == 0 ? 2u : 0u); }                         // RR FR RF FF
          if (S.cidx[to] == 0xFFFFu) continue;                       // removeNode of a non-survivor took the edge with it
          X.s_edges[9 * (size_t)si + (uint32_t)m] = S.cidx[to];     // (neighbour by survivor index: the component search below)
          G.edges[m++] = ED_MAKE(to, dir);
        }
        for (int i = m; i < 8; ++i) X.s_edges[9 * (size_t)si + (uint32_t)i] = 0xFFFFu;
        X.s_edges[9 * (size_t)si + 8] = ((inmer[n >> 5] >> (n & 31u)) & 1u);
        for (int i = m; i < LC_EMAX; ++i) G.edges[i] = 0;
        const unsigned long long c4 = X.tcc[ti];
        const uint32_t c0 = (uint32_t)(c4 & 0xFFFFu), c1 = (uint32_t)((c4 >> 16) & 0xFFFFu), c2 = (uint32_t)((c4 >> 32) & 0xFFFFu), c3 = (uint32_t)(c4 >> 48);
        const uint32_t f = X.tfl[ti];
        G.flags = f | NF_SURV | ((inmer[n >> 5] >> (n & 31u)) & 1u ? NF_INMER : 0u);
        G.necnt = (uint32_t)m; G.comp = 0; G.color = 0; G.onref = 0; G.nkm = 1;
        G.nkmT = ((f & NF_TUMOR) && !(f & NF_NORMAL)) ? 1u : 0u;
        G.cov[0] = (float)c0; G.cov[1] = (float)c1; G.cov[2] = (float)c2; G.cov[3] = (float)c3;
        G.kc[0] = (uint16_t)c0; G.kc[1] = (uint16_t)c1; G.kc[2] = (uint16_t)c2; G.kc[3] = (uint16_t)c3;
        G.mincov = (int)(c0 + c1 + c2 + c3); G.mincovqv = (int)X.c_minqv[ci];
        const uint32_t base = ci * (uint32_t)K;
        G.seq_clo = base; G.seq_lo = base; G.seq_hi = base + (uint32_t)K; G.seq_chi = base + (uint32_t)K;
        G.nqv = ci;
      }
      WG_SYNC();
    }
  }
  BLP(S, 15);
  BL_TAIL_PRIO_SET(BL_TAIL_PRIO);
  // ---- libstdc++ iteration order of the node table after the N inserts (kernels.h first_lowcov / order_stage),
  //      reduced to the survivors (cleanDead), and markConnectedComponents over them -- all in LDS: the reads are done with, the
  //      whole arena from S.bases to the end of S.big is laid out anew.  Tables of more than 4096 nodes: the 1024-lane configuration, whose
  //      arena holds the arrays for its BL_NCAP nodes once the bucket minima and the run counters are 16-bit halves (below); else the window kernel.
  WG_LANE0 { H->have_order = 0; }
  const bool wide_tab = BL_WIDE != 0 && N > 4096u;
  if ((N <= 4096u || (wide_tab && N <= BL_NCAP)) && nsurv > 0) {
    LC_LDS uint8_t *arena = (LC_LDS uint8_t *)&S.bases[0];
    LC_LDS uint32_t *first = (LC_LDS uint32_t *)arena;                       // [5120] smallest position of a bucket's elements
    LC_LDS uint32_t *tmp = first + 5120;                                     // [4104] run sizes -> run starts -> fill cursors
    LC_LDS uint16_t *Qa = (LC_LDS uint16_t *)(tmp + 4104), *Qb = Qa + 4096, *bkt = Qb + 4096, *out = bkt + 4096;
    LC_LDS unsigned long long *nh = (LC_LDS unsigned long long *)(out + 4096);   // [32] hashes of the first inserts
    LC_LDS uint32_t *nx = (LC_LDS uint32_t *)(nh + 32), *bk = nx + 32;           // [32] list links, [64] buckets of the sequential prefix
    LC_LDS uint16_t *pos2si = (LC_LDS uint16_t *)(bk + 64);                      // [PB_SCAP] survivor index of the node at a position
    LC_LDS uint32_t *heads = (LC_LDS uint32_t *)(pos2si + PB_SCAP);              // [128] bit x: position x is the first of a run (growth stages)
    static_assert(5120 * 4 + 4104 * 4 + 4 * 4096 * 2 + 32 * 8 + 32 * 4 + 64 * 4 + PB_SCAP * 2 + 128 * 4 <= offsetof(BlShared, big) + BL_BIG, "order arena");
    // what the steps after the stages use, wherever the layout puts it: the scan array (N + 1 words; then the components' parent / touch words),
    // node -> position among the survivors, the neighbour lists, the component numbers
    LC_LDS uint32_t *tmpP = tmp, *numP = first; LC_LDS uint16_t *nposP = (LC_LDS uint16_t *)first, *adjP = Qa;
#if BL_WIDE
    LC_LDS uint16_t *first16 = (LC_LDS uint16_t *)arena, *tmp16 = first16 + 20768, *bktw = nullptr;       // [20 753 + 1] bucket minima ; [BL_NCAP + 2] run counters
    static_assert(BL_NCAP <= 14336 && BL_NCAP < 16384, "positions / run sizes in 16 bits, bucket counts up to 20 753");
    if (wide_tab) {
      Qa = tmp16 + 14352; Qb = Qa + 14336; bktw = Qb + 14336;
      nh = (LC_LDS unsigned long long *)(bktw + 14336); nx = (LC_LDS uint32_t *)(nh + 32); bk = nx + 32; pos2si = (LC_LDS uint16_t *)(bk + 64);
      tmpP = (LC_LDS uint32_t *)arena;                                       // [14 337] over first16 / tmp16 (57 348 of their 70 240 bytes) ...
      numP = tmpP + 14352;                                                   // ... [PB_SCAP + 1] behind it
      nposP = bktw;                                                          // [N] (the buckets are done with)
      adjP = (LC_LDS uint16_t *)(tmpP + 4104);                               // [8 * PB_SCAP] behind parent / touch, inside the scan array (read out by then)
    }
    static_assert(2 * (20768 + 14352 + 3 * 14336) + 32 * 8 + 32 * 4 + 64 * 4 + PB_SCAP * 2 <= offsetof(BlShared, big) + BL_BIG && 4 * (14352 + PB_SCAP + 1) <= 2 * (20768 + 14352) &&
                  4 * 4104 + 2 * 8 * PB_SCAP <= 4 * 14337, "order arena of the large tables");
#endif
    LC_GLOBAL const unsigned long long *nhash = (LC_GLOBAL const unsigned long long *)(area + PRE_OFF_NHASH);
    const uint32_t SEQ = 13u;
    const uint32_t n0 = N < SEQ ? N : SEQ;
    WG_SYNC();
    WG_FOR(i, n0) { nh[i] = nhash[i]; }
    WG_LANE0 {                                                               // the first inserts literally (_M_insert_bucket_begin / _M_rehash_aux)
      uint32_t bc = 1, next_resize = 0, elt = 0, head = LC_NIL;
      bk[0] = LC_NIL;
      for (uint32_t n = 0; n < n0; ++n) {
        if (elt + 1 > next_resize) {
          unsigned long long mn = elt + 1;
          if (next_resize == 0 && mn < 11) mn = 11;
          if (mn >= bc) {
            unsigned long long want = mn + 1; if (want < 2ULL * bc) want = 2ULL * bc;
            const uint32_t nb = ht_next_prime((uint32_t)want);
            next_resize = nb;
            for (uint32_t i = 0; i < nb; ++i) bk[i] = LC_NIL;
            uint32_t pp = head; head = LC_NIL; uint32_t bbegin = 0;
            while (pp != LC_NIL) {
              const uint32_t nxt = nx[pp], b = ht_mod(nh[pp], nb);
              if (bk[b] == LC_NIL) { nx[pp] = head; head = pp; bk[b] = LC_BB; if (nx[pp] != LC_NIL) bk[bbegin] = pp; bbegin = b; }
              else { const uint32_t prev = bk[b]; if (prev == LC_BB) { nx[pp] = head; head = pp; } else { nx[pp] = nx[prev]; nx[prev] = pp; } }
              pp = nxt;
            }
            bc = nb;
          } else next_resize = bc;
        }
        const uint32_t b = ht_mod(nh[n], bc), prev = bk[b];
        if (prev != LC_NIL) { if (prev == LC_BB) { nx[n] = head; head = n; } else { nx[n] = nx[prev]; nx[prev] = n; } }
        else { nx[n] = head; head = n; if (nx[n] != LC_NIL) bk[ht_mod(nh[nx[n]], bc)] = n; bk[b] = LC_BB; }
        ++elt;
      }
      uint32_t m = 0;
      for (uint32_t pp = head; pp != LC_NIL; pp = nx[pp]) Qa[m++] = (uint16_t)pp;
      S.g0 = bc; S.g1 = next_resize;
    }
    LC_LDS uint16_t *Q = Qa, *Qn = Qb;
    uint32_t nprev = SEQ, B = SEQ;
    // Six barriers per growth stage (eight until round 5 -- these stages are barriers with little work between them, ~100 of a window's ~300):
    // the bucket minima carry a stamp that falls from stage to stage above the position, so that a stage's atomicMin overrides what the
    // stages before left and `first` is cleared once, not per stage; the stage's last pass -- the lane at the head of a run orders the run
    // and moves its nodes into the other order array -- also clears the run counters and appends the next stage's new nodes.
    if (N > SEQ && !wide_tab) {
      {
        const uint32_t B1 = ht_next_prime(2u * SEQ), n1 = N < B1 ? N : B1;
        WG_FOR(b, 5120) { first[b] = LC_NIL; }
        WG_FOR(i, 128) { heads[i] = 0; }
        WG_FOR(i, n1 + 1) { tmp[i] = 0; }
        WG_FOR(j, n1 - SEQ) { Q[SEQ + (uint32_t)j] = (uint16_t)(SEQ + (uint32_t)j); }
        WG_SYNC();
      }
      uint32_t stamp = 0xFFFEu;
      while (true) {
        B = ht_next_prime(2u * B);
        const uint32_t n = N < B ? N : B;
        const uint32_t st = stamp << 16; --stamp;
        BLPO(S, 1);
        WG_FOR(i, 128) { heads[i] = 0; }                                        // (set by this stage's scan, read in its last pass: two barriers from here)
        // (a lane's elements together: position -> node -> hash is an LDS and a global round trip, one after the other per element otherwise)
        static_assert(8 * BL_WG >= 4096, "bl_stage_buckets<8> covers a growth stage of up to 4096 elements");
        if (n <= 2u * BL_WG) bl_stage_buckets<2>(Q, nhash, n, B, st, bkt, first);
        else if (n <= 4u * BL_WG) bl_stage_buckets<4>(Q, nhash, n, B, st, bkt, first);
        else bl_stage_buckets<8>(Q, nhash, n, B, st, bkt, first);
        WG_SYNC();
        BLPO(S, 2);
        WG_FOR(i, n) { dev_atomic_add(&tmp[n - 1 - (first[bkt[i]] & 0xFFFFu)], 1u); }      // elements per run, runs indexed by their first position, latest first
        BLPO(S, 3);
        bl_scan32_heads(tmp, (int)n, S, heads);
        BLPO(S, 4);
        WG_FOR(i, n) { const uint32_t at = dev_atomic_add(&tmp[n - 1 - (first[bkt[i]] & 0xFFFFu)], 1u); out[at] = (uint16_t)i; }
        WG_SYNC();
        BLPO(S, 5);
        const bool last = N <= B;
        const uint32_t Bn = ht_next_prime(2u * B), nn = N < Bn ? N : Bn;
        WG_FOR(x, n) {                                                          // inside a run: latest first; then the run's nodes into the other array
          const uint32_t hw = heads[(uint32_t)x >> 5];
          if (!((hw >> ((uint32_t)x & 31u)) & 1u)) continue;                     // (not the first position of a run)
          uint32_t e = n;                                                       // the next run's first position
          { uint32_t wd = (uint32_t)x >> 5, m = ((uint32_t)x & 31u) == 31u ? 0u : (hw >> (((uint32_t)x & 31u) + 1u)) << (((uint32_t)x & 31u) + 1u);
            while (true) { if (m) { const uint32_t c = 32u * wd + (uint32_t)__builtin_ctz(m); if (c < e) e = c; break; } if (32u * (++wd) >= n) break; m = heads[wd]; } }
          for (uint32_t i = (uint32_t)x + 1; i < e; ++i) { const uint16_t v = out[i]; uint32_t j = i; while (j > (uint32_t)x && out[j - 1] < v) { out[j] = out[j - 1]; --j; } out[j] = v; }
          for (uint32_t i = (uint32_t)x; i < e; ++i) Qn[i] = Q[out[i]];
        }
        if (!last) {
          WG_FOR(i, nn + 1) { tmp[i] = 0; }
          WG_FOR(j, nn - n) { Qn[n + (uint32_t)j] = (uint16_t)(n + (uint32_t)j); }
        }
        WG_SYNC();
        BLPO(S, 6);
        { LC_LDS uint16_t *t = Q; Q = Qn; Qn = t; }
        if (last) break;
        nprev = B;
      }
    }
#if BL_WIDE
    // the same stages for a table of up to BL_NCAP nodes (bucket counts up to 20 753): first / run counters as 16-bit halves of LDS words
    // (positions and run sizes stay below 16 384), the sorted positions written straight into the other order array and converted in place
    if (N > SEQ && wide_tab) while (true) {
      B = ht_next_prime(2u * B);
      const uint32_t n = N < B ? N : B;
      WG_FOR(j, n - nprev) { Q[nprev + (uint32_t)j] = (uint16_t)(nprev + (uint32_t)j); }
      WG_FOR(b, (B + 2u) / 2u) { ((LC_LDS uint32_t *)first16)[b] = 0xFFFFFFFFu; }
      WG_FOR(i, (n + 3u) / 2u) { ((LC_LDS uint32_t *)tmp16)[i] = 0; }
      WG_SYNC();
      WG_FOR(i, n) { const uint32_t b = ht_mod(nhash[Q[i]], B); bktw[i] = (uint16_t)b; bl_min16(first16, b, (uint32_t)i); }
      WG_SYNC();
      WG_FOR(i, n) { (void)bl_add16(tmp16, n - 1u - (uint32_t)first16[bktw[i]], 1u); }     // elements per run, runs indexed by their first position, latest first
      bl_scan_t<uint16_t>(tmp16, (int)n, S);
      WG_FOR(i, n) { const uint32_t at = bl_add16(tmp16, n - 1u - (uint32_t)first16[bktw[i]], 1u); Qn[at] = (uint16_t)i; }
      WG_SYNC();
      WG_FOR(x, n) {                                                          // inside a run: latest first
        if (x > 0 && bktw[Qn[x - 1]] == bktw[Qn[x]]) continue;
        const uint32_t b = bktw[Qn[x]];
        uint32_t e = (uint32_t)x + 1; while (e < n && bktw[Qn[e]] == b) ++e;
        for (uint32_t i = (uint32_t)x + 1; i < e; ++i) { const uint16_t v = Qn[i]; uint32_t j = i; while (j > (uint32_t)x && Qn[j - 1] < v) { Qn[j] = Qn[j - 1]; --j; } Qn[j] = v; }
      }
      WG_SYNC();
      WG_FOR(j, n) { Qn[j] = Q[Qn[j]]; }                                     // (positions -> nodes: every lane its own entries)
      WG_SYNC();
      { LC_LDS uint16_t *t = Q; Q = Qn; Qn = t; }
      if (N <= B) break;
      nprev = B;
    }
#endif
    if (N > SEQ) { WG_LANE0 { S.g0 = B; S.g1 = B; } }
    // ---- cleanDead: the survivors in that order; position of every survivor
    LC_GLOBAL const uint8_t *surv = (LC_GLOBAL const uint8_t *)(area + PRE_OFF_SURV);
    LC_GLOBAL uint32_t *order_s = (LC_GLOBAL uint32_t *)(area + PRE_OFF_ORDER);
    LC_GLOBAL const uint32_t *sidv = (LC_GLOBAL const uint32_t *)(area + PRE_OFF_SID);
    WG_FOR(j, N + 1) { tmpP[j] = (j < (int)N && surv[Q[j]]) ? 1u : 0u; }
    bl_scan32(tmpP, (int)N + 1, S);
    WG_FOR(j, N) { if (tmpP[j + 1] != tmpP[j]) { order_s[tmpP[j]] = Q[j]; nposP[Q[j]] = (uint16_t)tmpP[j]; } }    // nposP[]: node -> position among the survivors
    WG_SYNC();
    // ---- markConnectedComponents (Graph.cc:2252-2336): min-label hooking + pointer jumping over the survivors' positions; the label of a
    //      component is the position of its first node in table order, which is also what numbers the components
    BLPA(S, 14);                                                             // (profiling builds: the components apart from the table order)
    BLPO(S, 7);
    LC_LDS uint32_t *parent = tmpP;                                          // [nsurv]
    LC_LDS uint32_t *touch = tmpP + 2052;                                    // [nsurv] bit 0: component holds a reference k-mer ; later: component number
    LC_LDS uint16_t *adj = adjP;                                             // [nsurv * 8] neighbours as positions (small tables: Qa .. out, 32 KB)
    WG_FOR(si, nsurv) {
      const uint32_t ppos = nposP[sidv[si]];
      for (int e = 0; e < 8; ++e) { const uint32_t t = X.s_edges[9 * (size_t)si + (uint32_t)e]; adj[8 * ppos + (uint32_t)e] = t == 0xFFFFu ? (uint16_t)ppos : nposP[sidv[t]]; }
      parent[ppos] = ppos; touch[ppos] = X.s_edges[9 * (size_t)si + 8]; pos2si[ppos] = (uint16_t)si;
    }
    WG_LANE0 { S.flagged = 0; S.ndup = 0; }                                  // (the rounds' two change flags, cleared in front of the barrier every wave passes before its first hooking pass)
    WG_SYNC();
#ifdef BL_CC_ROUNDS
    // (a round = one hooking pass + three pointer-jumping passes, one barrier each, then ONE look at the change flag: testing
    //  for convergence after every pass cost two more barriers of the 512-lane workgroup per pass, and these passes do little
    //  else than wait at barriers.  A round that changed nothing leaves every parent a root with no smaller neighbour label.
    //  Two change flags taken in turn -- S.flagged / S.ndup, both idle here -- so that a round is its four barriers and nothing else: the
    //  flag of the round after is cleared by lane 0 in this round's last pass, and every lane looks at this round's flag behind the last barrier.)
    for (int round = 0; ; ++round) {
      LC_LDS uint32_t *flag = (round & 1) ? (LC_LDS uint32_t *)&S.ndup : (LC_LDS uint32_t *)&S.flagged;
      LC_LDS uint32_t *other = (round & 1) ? (LC_LDS uint32_t *)&S.flagged : (LC_LDS uint32_t *)&S.ndup;
      WG_FOR(u, nsurv) {
        const uint32_t pu = ld2(&parent[u]);
        uint32_t m = pu;
        for (int e = 0; e < 8; ++e) { const uint32_t pv = ld2(&parent[adj[8 * (uint32_t)u + (uint32_t)e]]); if (pv < m) m = pv; }
        if (m < pu) { dev_atomic_min(&parent[pu], m); dev_atomic_min(&parent[u], m); *flag = 1; }
      }
      WG_SYNC();
      for (int jp = 0; jp < 3; ++jp) {
        WG_FOR(u, nsurv) { const uint32_t pu = ld2(&parent[u]), gp = ld2(&parent[pu]); if (gp != pu) { dev_atomic_min(&parent[u], gp); *flag = 1; } }
        if (jp == 2) { WG_FOR(z, 1) { *other = 0; } }
        WG_SYNC();
      }
      if (!*(volatile LC_LDS uint32_t *)flag) break;
    }
#else
    // (round 6: a lock-free union-find over the positions instead of rounds of hooking and pointer jumping -- 7 to 10 rounds of four passes
    //  and four barriers each on a graph that is mostly long chains in hash order.  A link always goes from the larger root to the smaller
    //  (compare-and-swap on a node that is still its own parent), finds halve the paths they walk (atomic min: a parent only ever moves towards
    //  the root), so the forest stays acyclic under any interleaving and the one root a component is left with is its smallest position -- the same
    //  labels the rounds converged to.  Two passes, two barriers.)
    auto uf_find = [&](uint32_t x) -> uint32_t {
      while (true) {
        const uint32_t p = ld2(&parent[x]);
        if (p == x) return x;
        const uint32_t gp = ld2(&parent[p]);
        if (gp == p) return p;
        dev_atomic_min(&parent[x], gp);
        x = gp;
      }
    };
    WG_FOR(u, nsurv) {
      for (int e = 0; e < 8; ++e) {
        uint32_t a = (uint32_t)u, b = adj[8 * (uint32_t)u + (uint32_t)e];
        if (b == a) continue;
        while (true) {
          a = uf_find(a); b = uf_find(b);
          if (a == b) break;
          const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
          if (dev_atomic_cas32(&parent[hi], hi, lo) == hi) break;
        }
      }
    }
    WG_SYNC();
    WG_FOR(u, nsurv) { const uint32_t r = uf_find((uint32_t)u); dev_atomic_min(&parent[u], r); }
    WG_SYNC();
#endif
    WG_FOR(u, nsurv) { if (touch[u] & 1u) dev_atomic_or(&touch[parent[u]], 2u); }
    WG_SYNC();
    LC_LDS uint32_t *num = numP;                                             // (small tables: over nposP -- positions are no longer looked up by node)
    WG_FOR(u, nsurv + 1) { num[u] = (u < (int)nsurv && parent[u] == (uint32_t)u) ? 1u : 0u; }
    bl_scan32(num, (int)nsurv + 1, S);
    WG_LANE0 { S.nbw = S.scan_total; S.ngw = 0; S.flagged = 0; }             // (nbw / ngw are free by now: components, components on the reference; flagged: which of the first 32 those are)
    WG_SYNC();
    WG_FOR(u, nsurv) { if (parent[u] == (uint32_t)u && (touch[u] & 2u)) { dev_atomic_add((LC_LDS uint32_t *)&S.ngw, 1u); if (num[u] < 32u) dev_atomic_or((LC_LDS uint32_t *)&S.flagged, 1u << num[u]); } }
    LC_GLOBAL NodeGr *pgr2 = (LC_GLOBAL NodeGr *)(area + PRE_OFF_PGR);
    WG_FOR(u, nsurv) { pgr2[pos2si[u]].comp = (int)(num[parent[u]] + 1u); }   // numbered by the position of the component's first node
    WG_LANE0 { H->have_order = 1; H->ht_bc = S.g0; H->ht_next_resize = S.g1; H->numcomp = S.nbw; H->refcomp = S.ngw; }
    WG_SYNC();
    BLPA(S, 15); BLPO(S, 15);
    // ---- the window's first graph with a single component: markRefEnds and the first compress here too (bl_compress_first)
    {
      WG_SYNC();
      const uint32_t hbc = lc_sgpr((uint32_t)S.g0), hnr = lc_sgpr((uint32_t)S.g1), ncomp = lc_sgpr((uint32_t)S.nbw), refmask = lc_sgpr((uint32_t)S.flagged);
      WG_SYNC();
      // (--linked-reads: a node's counts are barcode counts the window kernel has yet to replay -- no compress here)
      // (round 6: several components too -- for component 1; only for a window's first graph: load_prebuilt's fix-up after an earlier k's trim
      //  knows one component)
      if ((!rep || cmp_later) && (ncomp == 1u || (!rep && ncomp <= 32u)) && nsurv <= PB_CMAX && nsurv + 2u <= hnr && K <= 31 && !wide_tab && C->debug_stop != 140u && !P->lr_mode)
        bl_compress_first(P, C, S, X, area, K, N, nsurv, ncand, reflen, hbc, refmask);
    }
  }
  BL_TAIL_PRIO_SET(0);
  WG_LANE0 {
    H->K = K; H->refE = S.repE; H->refM = S.repM; H->N = S.N; H->O = S.O; H->totalreadbp = S.totalreadbp; H->n_kmers = S.n_kmers;
    H->ncand = S.ncand; H->nsurv = S.nsurv; H->edges_total = S.edges_total; H->refn = S.refn; H->why = 0; H->heavy = S.hint;
    H->big = BL_WG > 512 ? 1u : 0u;
    H->status = PB_BUILT;
  }
  WG_SYNC();
}

// entry: persistent workgroups pull windows off the batch queue (`queue` is a counter block of its own)
// Building ahead: a window whose graph holds the same k-mer twice in one read (PreHdr::heavy) will almost always be rejected at
// this k (the duplication is a cycle), and the window kernel would then build the next graph with its general, HBM-resident
// phases on one wave -- ~30x the latency of this kernel, and the few windows that climb through several k are the critical
// path of the whole launch.  So the workgroup goes on to the next k of the window's loop right away (up to `depth` graphs
// ahead), into an area of `pool` (pool_cap areas, handed out by queue[2]); PreHdr::next links them.  A graph built ahead that
// the window kernel does not ask for is wasted work, nothing else: results never depend on what was built ahead.
// (queue[0] = next window, queue[1] = windows built, queue[2] = pool areas handed out, queue[3] = graphs built ahead,
//  queue[4] = windows on `biglist`, queue[5] = next entry of it, queue[7] = windows finished -- built, listed or turned away)
// The workgroups of the build service (svc_kernel_body) work the same queue off until it is empty, before they serve their first
// request; the kernel that the window kernel is ordered behind is this one, so its workgroups leave only when every window handed
// out -- to either kernel -- is finished (`helpers`: queue[7], released by whoever finished the window).
// `biglist`: the small configuration appends the windows it turns away for their size; the large one (from_list) works that list
// off instead of the batch.
#ifndef BL_END_G
#define BL_END_G 512
#endif
DEV void build_kernel_body(LC_GLOBAL const lancet_params *P, LC_GLOBAL const DevBatch *B, LC_GLOBAL const EngineCaps *C, LC_GLOBAL uint8_t *pre,
                           LC_GLOBAL uint8_t *scratch, LC_GLOBAL uint32_t *queue, BL_S &S, int slot, LC_GLOBAL unsigned long long *phase = nullptr,
                           LC_GLOBAL uint8_t *pool = nullptr, uint32_t pool_cap = 0, int depth = 0, LC_GLOBAL uint32_t *biglist = nullptr, bool from_list = false,
                           bool wait_all = false, int cycle = 0) {
  LC_GLOBAL const PreLayout &PL = C->pl;
  LC_GLOBAL uint8_t *xbase = scratch + (size_t)slot * SCRATCH_BYTES;
  int done_here = 0;
  if (from_list && queue[4] == 0u) return;                       // (nothing was turned away for its size: the usual case at 30x)
  while (true) {
    WG_LANE0 { S.w = (int)dev_atomic_add(queue + (from_list ? 5 : 0), 1u); }
    int w = (int)bl_bcast(&S.w);
    if (from_list) { if ((uint32_t)w >= queue[4]) break; w = (int)biglist[w]; }
    else if (w >= B->n_windows) {
#ifndef LANCET_WAVE_EMU
      if (wait_all) { WG_LANE0 { while (ld_acq(queue + 7) < (uint32_t)B->n_windows) dev_sleep(); } WG_SYNC(); }   // (the service's workgroups may still be at their last windows)
#endif
      break;
    }
#ifndef LANCET_WAVE_EMU
    if (threadIdx.x == 0) { if (phase) { for (int i = 0; i < 16; ++i) S.ph_acc[i] = 0; S.ph_cur = 0; S.t_last = wall_clock64(); } else S.ph_cur = -1; }
#endif
    bl_build_window(P, B, C, S, xbase, pre + (size_t)w * PRE_STRIDE, w, P->min_k, nullptr);
    {
      LC_GLOBAL PreHdr *H0 = (LC_GLOBAL PreHdr *)(pre + (size_t)w * PRE_STRIDE), *cur = H0;
      // (the end of the queue: a chain of graphs built ahead for one of the last windows taken is what the whole launch then waits for -- every other
      //  workgroup has left.  The last BL_END_G windows of the batch get no graph ahead, the BL_END_G before them one, ...: what they lack the build
      //  service builds at the START of the window kernel's launch, where it has nothing else to do -- these windows are taken first there, PreHdr::heavy)
      const int dmax = from_list ? depth : (B->n_windows - 1 - w) / BL_END_G;
      for (int lvl = 0; pool && lvl < depth && lvl < dmax; ++lvl) {
        WG_SYNC();
        WG_LANE0 {
          S.scan_total = 0xFFFFFFFFu;
          if (cur->status == PB_BUILT && cur->heavy && cur->K + 2 <= P->max_k) { const uint32_t a = dev_atomic_add(queue + 2, 1u); if (a < pool_cap) S.scan_total = a; }
        }
        const uint32_t a = bl_bcast(&S.scan_total);
        if (a == 0xFFFFFFFFu) break;
        LC_GLOBAL uint8_t *nx = pool + (size_t)a * PRE_STRIDE;
        bl_build_window(P, B, C, S, xbase, nx, w, cur->K + 2, H0, true);
        WG_SYNC();
        WG_LANE0 { if (((LC_GLOBAL PreHdr *)nx)->status == PB_BUILT) { cur->next = a + 1u; dev_atomic_add(queue + 3, 1u); } }
        if (((LC_GLOBAL PreHdr *)nx)->status != PB_BUILT) break;
        cur = (LC_GLOBAL PreHdr *)nx;
      }
    }
    BLP(S, 15);
#ifndef LANCET_WAVE_EMU
    if (threadIdx.x == 0 && phase) for (int i = 0; i < 16; ++i) if (S.ph_acc[i]) atomicAdd((unsigned long long *)&phase[i], S.ph_acc[i]);
#endif
    WG_LANE0 {
      LC_GLOBAL const PreHdr *Hw = (LC_GLOBAL const PreHdr *)(pre + (size_t)w * PRE_STRIDE);
      if (Hw->status == PB_BUILT) dev_atomic_add(queue + 1, 1u);
      else if (!from_list && biglist && (Hw->why == (uint32_t)BLW_SIZE || Hw->why == (uint32_t)BLW_KBIG)) biglist[dev_atomic_add(queue + 4, 1u)] = (uint32_t)w;
    }
    WG_SYNC();                                                     // (every lane's stores to the hand-off area are issued ...)
    // (... and released with the count -- by the service's workgroups: their kernel stays resident, nothing else makes what they wrote visible
    //  to the window kernel; the build kernel's own stores are written back when it ends, and an agent-scope release per window is an L2
    //  write-back per window)
    if (!from_list) { WG_LANE0 { if (wait_all) dev_atomic_add(queue + 7, 1u); else add_rel(queue + 7, 1u); } }
    if (cycle > 0 && ++done_here >= cycle) break;                  // (LANCET_BUILD_CYCLE: the grid holds a workgroup for every `cycle` windows)
  }
}

// Build service (layout.h SvcCtl): resident workgroups that build, on request of the window kernel, the graph of a window at a
// later k of its loop -- the window was suspended when its k was rejected and nobody had built the next graph ahead.  A
// workgroup takes a ticket, waits for that request to be posted (or for `done`), builds into a pool area (and on into the
// following k while the hint says that one will be rejected too), chains the area to the window's hand-off and puts the
// request on the ready list.  Host emulation: one call serves what is posted and returns.
// A waiting workgroup also leaves when NOTHING of its batch has made progress for ~300 ms (another engine's batch may hold the device for tens of ms first) (the build kernel's and the window
// kernel's queue heads, the slots' heartbeat, the request counter): under a profiler that serialises kernels (rocprofv3 --pmc) the
// service would otherwise wait for a window kernel that cannot start before it has left.  The slots then find no service
// (SvcCtl::alive == 0) and build those graphs themselves.
DEV void svc_kernel_body(LC_GLOBAL const lancet_params *P, LC_GLOBAL const DevBatch *B, LC_GLOBAL const EngineCaps *C, LC_GLOBAL uint8_t *pre,
                         LC_GLOBAL uint8_t *scratch, LC_GLOBAL uint32_t *queue, BL_S &S, int slot, LC_GLOBAL uint8_t *pool, uint32_t pool_cap, int depth,
                         LC_GLOBAL SvcCtl *sv, LC_GLOBAL const uint32_t *wqueue = nullptr, LC_GLOBAL uint32_t *biglist = nullptr, int help_depth = -1,
                         LC_GLOBAL unsigned long long *phase = nullptr) {
  LC_GLOBAL const PreLayout &PL = C->pl;
  LC_GLOBAL uint8_t *xbase = scratch + (size_t)slot * SCRATCH_BYTES;
  // Until the window kernel runs there is nothing to serve: the workgroup takes windows off the build kernel's queue like that kernel's own
  // (help_depth >= 0: graphs built ahead per window there; the build kernel waits for the windows taken here, build_kernel_body `wait_all`).
  if (help_depth >= 0) build_kernel_body(P, B, C, pre, scratch, queue, S, slot, phase, pool, pool_cap, help_depth, biglist, false, false);
#ifndef LANCET_WAVE_EMU
  WG_LANE0 { dev_atomic_add(&sv->alive, 1u); }
  unsigned long long t_prog = wall_clock64(); uint32_t last_sum = 0xFFFFFFFFu;      // (lane 0's)
#endif
  while (true) {
    WG_LANE0 { S.w = (int)dev_atomic_add(&sv->ticket, 1u); }
    const uint32_t t = bl_bcast(&S.w);
    if (t >= sv->cap) break;
    WG_LANE0 {
      uint32_t st;
      while (true) {
        st = ld_acq(&sv->req[t].state);
        if (st != SV_EMPTY) break;
#ifdef LANCET_WAVE_EMU
        sv->ticket = t; st = 0xFFFFFFFFu; break;                 // (nothing more is posted: hand the ticket back)
#else
        if (ld2(&sv->done)) { st = 0xFFFFFFFFu; break; }
        {
          const uint32_t sum = ld2(queue) + ld2(&sv->beat) + ld2(&sv->req_alloc) + (wqueue ? ld2(wqueue) : 0u);
          const unsigned long long now = wall_clock64();
          if (sum != last_sum) { last_sum = sum; t_prog = now; }
          else if (now - t_prog > 30000000ull) { dev_atomic_add(&sv->n_gaveup, 1u); st_rel(&sv->nosvc, 1u); st = 0xFFFFFFFFu; break; }      // 300 ms at 100 MHz (the ticket it holds is never served: no window posts a request from here on)
        }
        dev_sleep();
#endif
      }
      if (st == SV_POSTED) st = dev_atomic_cas32(&sv->req[t].state, SV_POSTED, SV_CLAIMED) == SV_POSTED ? SV_CLAIMED : SV_STOLEN;
      S.g0 = st;
    }
    const uint32_t st = bl_bcast(&S.g0);
    if (st == 0xFFFFFFFFu) break;
    if (st != SV_CLAIMED) continue;                                 // a window slot took it back
    const int w = (int)sv->req[t].w, k = sv->req[t].k;
#ifndef LANCET_WAVE_EMU
    if (threadIdx.x == 0) { if (phase) { for (int i = 0; i < 16; ++i) S.ph_acc[i] = 0; S.ph_cur = 0; S.t_last = wall_clock64(); } else S.ph_cur = -1; }
#endif
    LC_GLOBAL PreHdr *H0 = (LC_GLOBAL PreHdr *)(pre + (size_t)w * PRE_STRIDE), *cur = H0;
    for (int hop = 0; hop < 64 && cur->next != 0u; ++hop) cur = (LC_GLOBAL PreHdr *)(pool + (size_t)(cur->next - 1u) * PRE_STRIDE);
    bool any = false;
    if (cur->status == PB_BUILT && cur->K < k) {
      for (int lvl = 0; lvl <= depth; ++lvl) {
        WG_SYNC();
        WG_LANE0 {
          S.scan_total = 0xFFFFFFFFu;
          if (lvl == 0 || (cur->heavy && cur->K + 2 <= P->max_k)) { const uint32_t a = dev_atomic_add(queue + 2, 1u); if (a < pool_cap) S.scan_total = a; }
        }
        const uint32_t a = bl_bcast(&S.scan_total);
        if (a == 0xFFFFFFFFu) break;
        LC_GLOBAL uint8_t *nx = pool + (size_t)a * PRE_STRIDE;
        bl_build_window(P, B, C, S, xbase, nx, w, lvl == 0 ? k : cur->K + 2, H0, true);
        WG_SYNC();
        if (((LC_GLOBAL PreHdr *)nx)->status != PB_BUILT) break;
        WG_LANE0 { cur->next = a + 1u; dev_atomic_add(queue + 3, 1u); }
        any = true;
        cur = (LC_GLOBAL PreHdr *)nx;
      }
    }
    WG_SYNC();
    WG_LANE0 {
      dev_atomic_add(any ? &sv->n_built : &sv->n_failed, 1u);
      sv->req[t].state = SV_DONE;
      const uint32_t r = dev_atomic_add(&sv->rdy_alloc, 1u);
      st_rel(&sv->rdy[r], t + 1u);
    }
  }
#ifndef LANCET_WAVE_EMU
  WG_LANE0 { dev_atomic_add(&sv->alive, 0xFFFFFFFFu); }
#endif
}

}  // namespace BL_NS
