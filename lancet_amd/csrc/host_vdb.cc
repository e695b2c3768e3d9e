// host_vdb.cc -- host side below the seam: Variant_t normalisation, VariantDB_t and the VCF writer
// (SURVEY.md §8(f) N3).  Double-precision Fisher scores are computed with the C library's lgamma/exp/log10
// exactly as the reference does, so that the printed 6-digit values agree.
//
//   Variant_t ctor        reference src/Variant.hh:106-172
//   getSignature          reference src/Variant.cc:339-344
//   VariantDB_t::addVar   reference src/VariantDB.cc:28-91   (std::map keyed by the sha256 hex of the signature)
//   printHeader/printToVCF reference src/VariantDB.cc:93-179 ; byPos src/VariantDB.hh:37-54 (std::sort, unstable)
//   Variant_t::printVCF   reference src/Variant.cc:39-223
//   FET_t                 reference src/FET.hh:36-128 (Heng Li's kt_fisher_exact)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#if defined(__x86_64__) || defined(__i386__)
#define LANCET_X86_SHA 1
#include <cpuid.h>
#include <immintrin.h>
#endif

#include "../../include/lancet_engine.h"

namespace {

// ---- SHA-256 (FIPS 180-4), hex digest: only its ordering matters (map iteration order of the final merge)
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

#ifdef LANCET_X86_SHA
// One 64-byte block with the x86 SHA extensions (sha256rnds2 / sha256msg1 / sha256msg2), used when the host CPU has them:
// rank 0 hashes every record of every rank when it replays them, and the scalar rounds were its largest single cost.
__attribute__((target("sha,sse4.1,ssse3"))) void sha256_block_ni(uint32_t st[8], const uint8_t *p) {
  const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
  __m128i t = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)&st[0]), 0xB1);   // CDAB
  __m128i s1 = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)&st[4]), 0x1B);  // EFGH
  __m128i s0 = _mm_alignr_epi8(t, s1, 8);                                          // ABEF
  s1 = _mm_blend_epi16(s1, t, 0xF0);                                               // CDGH
  const __m128i s0_in = s0, s1_in = s1;
  __m128i m[4];
  for (int r = 0; r < 4; ++r) m[r] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(p + 16 * r)), bswap);
  for (int r = 0; r < 16; ++r) {                                                   // four rounds per turn
    if (r >= 4) {
      __m128i w = _mm_sha256msg1_epu32(m[r & 3], m[(r + 1) & 3]);
      w = _mm_add_epi32(w, _mm_alignr_epi8(m[(r + 3) & 3], m[(r + 2) & 3], 4));
      m[r & 3] = _mm_sha256msg2_epu32(w, m[(r + 3) & 3]);
    }
    __m128i x = _mm_add_epi32(m[r & 3], _mm_loadu_si128((const __m128i *)&SHA_K[4 * r]));
    s1 = _mm_sha256rnds2_epu32(s1, s0, x);
    s0 = _mm_sha256rnds2_epu32(s0, s1, _mm_shuffle_epi32(x, 0x0E));
  }
  s0 = _mm_add_epi32(s0, s0_in); s1 = _mm_add_epi32(s1, s1_in);
  t = _mm_shuffle_epi32(s0, 0x1B);                                                 // FEBA
  s1 = _mm_shuffle_epi32(s1, 0xB1);                                                // DCHG
  _mm_storeu_si128((__m128i *)&st[0], _mm_blend_epi16(t, s1, 0xF0));               // DCBA
  _mm_storeu_si128((__m128i *)&st[4], _mm_alignr_epi8(s1, t, 8));                  // HGFE
}
bool sha_ni_detect() {
  unsigned a, b, c, d;
  if (getenv("LANCET_NO_SHA_NI")) return false;
  if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
  bool sha = (b >> 29) & 1;
  if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
  return sha && ((c >> 19) & 1) && ((c >> 9) & 1);                                 // + sse4.1, ssse3
}
#endif
struct Sha256 {
  uint32_t h[8]; uint8_t buf[64]; uint64_t len; size_t fill;
  static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void init() {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(h, iv, sizeof(h)); len = 0; fill = 0;
  }
  static bool use_ni();
  void block_scalar(const uint8_t *p) {
    const uint32_t *k = SHA_K;
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      uint32_t S1 = ror(e, 6) ^ ror(e, 11) ^ ror(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + k[i] + w[i];
      uint32_t S0 = ror(a, 2) ^ ror(a, 13) ^ ror(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void block(const uint8_t *p) {
#ifdef LANCET_X86_SHA
    if (use_ni()) { sha256_block_ni(h, p); return; }
#endif
    block_scalar(p);
  }
  void update(const uint8_t *p, size_t n) {
    len += n;
    while (n) { size_t t = std::min(n, 64 - fill); memcpy(buf + fill, p, t); fill += t; p += t; n -= t; if (fill == 64) { block(buf); fill = 0; } }
  }
  void digest(uint8_t out[32]) {
    uint64_t bits = len * 8;
    uint8_t tail[72]; size_t nt = (fill < 56 ? 56 : 120) - fill;
    memset(tail, 0, sizeof(tail)); tail[0] = 0x80;
    update(tail, nt);
    uint8_t l[8]; for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(l, 8);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
  }
};
// The SHA-NI rounds are used only when the CPU has them AND they reproduce the scalar rounds on one block (checked once, at first use).
bool Sha256::use_ni() {
#ifdef LANCET_X86_SHA
  static const bool ok = [] {
    if (!sha_ni_detect()) return false;
    uint8_t blk[64]; for (int i = 0; i < 64; ++i) blk[i] = (uint8_t)(37 * i + 11);
    Sha256 a, b; a.init(); b.init();
    sha256_block_ni(a.h, blk); b.block_scalar(blk);
    return memcmp(a.h, b.h, sizeof(a.h)) == 0;
  }();
  return ok;
#else
  return false;
#endif
}
// The map key.  The reference keys its std::map by the 64-character hex digest; hex encoding keeps byte order
// ('0'..'9' < 'a'..'f'), so ordering the 32 raw bytes with memcmp IS the reference's iteration order.
struct Dig {
  uint8_t b[32];
  bool operator<(const Dig &o) const { return memcmp(b, o.b, 32) < 0; }
  bool operator==(const Dig &o) const { return memcmp(b, o.b, 32) == 0; }
};
struct DigHash { size_t operator()(const Dig &d) const { uint64_t x; memcpy(&x, d.b + 8, 8); return (size_t)x; } };   // (a digest is its own hash; byte 0 picks the shard)
void sha256_bin(const char *p, size_t n, Dig &d) { Sha256 c; c.init(); c.update((const uint8_t *)p, n); c.digest(d.b); }

// ---- FET_t
double lbinom(int n, int k) { if (k == 0 || n == k) return 0; return lgamma(n + 1) - lgamma(k + 1) - lgamma(n - k + 1); }
double hypergeo(int n11, int n1_, int n_1, int n) { return exp(lbinom(n1_, n11) + lbinom(n - n1_, n_1 - n11) - lbinom(n, n_1)); }
// only q (the probability of the observed table) is consumed by the reference; the tails do not feed back into it
double kt_fisher_exact_q(int n11, int n12, int n21, int n22) {
  int n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
  int mx = (n_1 < n1_) ? n_1 : n1_;
  int mn = n1_ + n_1 - n; if (mn < 0) mn = 0;
  if (mn == mx) return 1.;
  return hypergeo(n11, n1_, n_1, n);
}

std::string dtos(double d) { std::ostringstream s; s << d; return s.str(); }
std::string itos(int i) { return std::to_string(i); }

struct Variant {
  unsigned short kmer = 0; std::string chr; int pos = 0; char type = '?'; unsigned short len = 0;
  std::string ref, alt, str;
  unsigned short rcn_f = 0, rcn_r = 0, rct_f = 0, rct_r = 0, acn_f = 0, acn_r = 0, act_f = 0, act_r = 0;

  // Variant_t's constructor normalisation (src/Variant.hh:106-172) on caller-owned strings, so that the replay can
  // reuse two per-thread buffers for the records it only needs the signature of.
  static void normalise(const lancet_variant &v, const char *blob, std::string &ref, std::string &alt, int &pos, char &type, unsigned short &len) {
    ref.assign(blob + v.ref_off, v.ref_len); alt.assign(blob + v.alt_off, v.alt_len);
    pos = v.pos; type = '?'; len = 0;
    char code = (char)v.code;
    if (code == '^') { type = 'I'; ref.clear(); len = (unsigned short)alt.length(); }
    if (code == 'v') { type = 'D'; alt.clear(); len = (unsigned short)ref.length(); }
    if (code == 'x') { type = 'S'; pos++; }
    if (code == 'c') {
      type = 'C';
      ref.erase(std::remove(ref.begin(), ref.end(), '-'), ref.end());
      alt.erase(std::remove(alt.begin(), alt.end(), '-'), alt.end());
      unsigned short rl = (unsigned short)ref.length(), al = (unsigned short)alt.length();
      if (rl == al) len = al; else if (rl > al) len = rl - al; else len = al - rl;
    }
    if (type != 'S') { ref.insert(ref.begin(), (char)v.prev_bp_alt); alt.insert(alt.begin(), (char)v.prev_bp_alt); }
    else len = 1;
  }
  Variant() {}
  Variant(const std::string &chr_, const lancet_variant &v, const char *blob) {
    kmer = v.kmer; chr = chr_;
    str.assign(blob + v.str_off, v.str_len);
    normalise(v, blob, ref, alt, pos, type, len);
    rcn_f = v.cov[0]; rcn_r = v.cov[1]; rct_f = v.cov[2]; rct_r = v.cov[3];
    acn_f = v.cov[4]; acn_r = v.cov[5]; act_f = v.cov[6]; act_r = v.cov[7];
  }
  static int tot_of(const lancet_variant &v) {
    int t = 0;
    for (int q = 0; q < 8; ++q) t += (unsigned short)v.cov[q];
    return t;
  }
  // --linked-reads members (Variant.hh:72-104): HPRN HPRT HPAN HPAT as {hp1, hp2, hp0}, barcode sets as printed
  bool lr = false;
  unsigned short hp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  std::string bx[4];                                   // bxset_ref_N, bxset_ref_T, bxset_alt_N, bxset_alt_T
  void set_lr(const lancet_variant_lr &l, const uint32_t *bx_blob, const char *const *bx_names, uint32_t n_bx) {
    lr = true;
    for (int i = 0; i < 12; ++i) hp[i] = l.hp[i];
    for (int q = 0; q < 4; ++q) {                      // Graph_t::getBXsetAt / Ref_t::getBXsetAt: ';'-joined, "." when empty
      std::string r;
      for (uint32_t j = 0; j < l.bx_len[q]; ++j) { uint32_t id = bx_blob[l.bx_off[q] + j]; if (j) r += ";"; r += id < n_bx ? bx_names[id] : "?"; }
      bx[q] = r.empty() ? "." : r;
    }
  }
  static double hp_score(int hpr1, int hpr2, int hpa1, int hpa2) {     // Variant_t::compute_HP_score, src/Variant.cc:281-298
    double prob = kt_fisher_exact_q(hpr1, hpr2, hpa1, hpa2);
    if (prob == 1) return 0.0;
    return -10.0 * log10(prob);
  }
  // getSignature (src/Variant.cc:339-344): chr:pos:type:len:ref:alt
  static void signature_into(std::string &out, const char *chr, int pos, char type, unsigned short len, const std::string &ref, const std::string &alt) {
    auto dec = [&out](int x) {                            // "%d"
      char num[12]; int k = 12; unsigned u = x < 0 ? 0u - (unsigned)x : (unsigned)x;
      do { num[--k] = (char)('0' + u % 10); u /= 10; } while (u);
      if (x < 0) num[--k] = '-';
      out.append(num + k, (size_t)(12 - k));
    };
    out.assign(chr); out += ':';
    dec(pos); out += ':';
    out += type; out += ':';
    dec((int)len); out += ':';
    out += ref; out += ':'; out += alt;
  }
  int tot() const { return rcn_f + rcn_r + rct_f + rct_r + acn_f + acn_r + act_f + act_r; }
  double fet_score() const {
    double prob = kt_fisher_exact_q(rcn_f + rcn_r, rct_f + rct_r, acn_f + acn_r, act_f + act_r);
    if (prob == 1.0) return 0.0;
    if (prob == 0.0) return -10.0 * log10(1 / std::numeric_limits<double>::max());
    return -10.0 * log10(prob);
  }
  double sb_score() const {
    double prob = kt_fisher_exact_q(rct_f, rct_r, act_f, act_r);
    if (prob == 1) return 0.0;
    return -10.0 * log10(prob);
  }
  static std::string genotype(int R, int A) { if (R > 0 && A > 0) return "0/1"; if (R > 0 && A == 0) return "0/0"; if (R == 0 && A > 0) return "1/1"; return "."; }
  std::string vcf(const lancet_filters &fs) const {
    int tr_t = rct_f + rct_r, ta_t = act_f + act_r, tr_n = rcn_f + rcn_r, ta_n = acn_f + acn_r;
    double fet = fet_score(), sb = sb_score();
    std::string status; bool somatic = false;
    if (ta_n > 0 && ta_t > 0) status = "SHARED"; else if (ta_n == 0 && ta_t > 0) { status = "SOMATIC"; somatic = true; } else if (ta_n > 0 && ta_t == 0) status = "NORMAL"; else return "";
    std::string INFO = status + ";FETS=" + dtos(fet);
    if (type == 'I') INFO += ";TYPE=ins";
    if (type == 'D') INFO += ";TYPE=del";
    if (type == 'S') INFO += ";TYPE=snv";
    if (type == 'C') INFO += ";TYPE=complex";
    INFO += ";LEN=" + itos(len) + ";KMERSIZE=" + itos(kmer) + ";SB=" + dtos(sb);
    const unsigned short *HPRN = hp, *HPRT = hp + 3, *HPAN = hp + 6, *HPAT = hp + 9;
    if (lr) {                                            // src/Variant.cc:56-60, 78-80
      double hpsn = hp_score(HPRN[0], HPRN[1], HPAN[0], HPAN[1]);
      double hpst = hp_score(HPRT[0], HPRT[1], HPAT[0], HPAT[1]);
      double hps = hp_score(HPRN[0] + HPAN[0], HPRN[1] + HPAN[1], HPRT[0] + HPAT[0], HPRT[1] + HPAT[1]);
      INFO += ";HPS=" + dtos(hps) + ";HPSN=" + dtos(hpsn) + ";HPST=" + dtos(hpst);
    }
    if (!str.empty()) INFO += ";MS=" + str;
    int tumor_cov = tr_t + ta_t; double tumor_vaf = (tumor_cov == 0) ? 0 : ((double)ta_t / (double)tumor_cov);
    int normal_cov = tr_n + ta_n; double normal_vaf = (normal_cov == 0) ? 0 : ((double)ta_n / (double)normal_cov);
    std::string F;
    auto add = [&](const char *s) { if (F.empty()) F = s; else { F += ";"; F += s; } };
    if (!str.empty()) { if (fet < fs.min_phred_fisher_str) add("LowFisherSTR"); }
    else if (fet < fs.min_phred_fisher) add("LowFisherScore");
    if (normal_cov < fs.min_cov_normal) add("LowCovNormal");
    if (normal_cov > fs.max_cov_normal) add("HighCovNormal");
    if (tumor_cov < fs.min_cov_tumor) add("LowCovTumor");
    if (tumor_cov > fs.max_cov_tumor) add("HighCovTumor");
    if (tumor_vaf < fs.min_vaf_tumor) add("LowVafTumor");
    if (normal_vaf > fs.max_vaf_normal) add("HighVafNormal");
    if (ta_t < fs.min_alt_cnt_tumor) add("LowAltCntTumor");
    if (ta_n > fs.max_alt_cnt_normal) add("HighAltCntNormal");
    if ((act_f < fs.min_strand_bias) || (act_r < fs.min_strand_bias)) add("StrandBias");
    if (lr && somatic && HPAT[0] > 0 && HPAT[1] > 0) add("MultiHP");   // src/Variant.cc:172-177
    if (F.empty()) F = "PASS";
    std::string NORMAL = genotype(tr_n, ta_n) + ":" + itos(tr_n) + "," + itos(ta_n) + ":" + itos(rcn_f) + "," + itos(rcn_r) + ":" + itos(acn_f) + "," +
                         itos(acn_r) + ":" + itos(tr_n + ta_n);
    std::string TUMOR = genotype(tr_t, ta_t) + ":" + itos(tr_t) + "," + itos(ta_t) + ":" + itos(rct_f) + "," + itos(rct_r) + ":" + itos(act_f) + "," +
                        itos(act_r) + ":" + itos(tr_t + ta_t);
    std::string FORMAT = "GT:AD:SR:SA:DP";
    if (lr) {                                            // src/Variant.cc:204-215
      FORMAT += ":HPR:HPA:BX";
      auto c3 = [](const unsigned short *h) { return itos(h[0]) + "," + itos(h[1]) + "," + itos(h[2]); };
      NORMAL += ":" + c3(HPRN) + ":" + c3(HPAN) + ":" + bx[0] + "," + bx[2];
      TUMOR += ":" + c3(HPRT) + ":" + c3(HPAT) + ":" + bx[1] + "," + bx[3];
    }
    std::ostringstream line;
    line << chr << "\t" << pos << "\t.\t" << ref << "\t" << alt << "\t" << fet << "\t" << F << "\t" << INFO << "\t" << FORMAT << "\t" << NORMAL << "\t" << TUMOR << std::endl;
    return line.str();
  }
};

struct byPos {   // reference src/VariantDB.hh:37-54 (pairs by value there; same ordering)
  bool operator()(const std::pair<const Dig, Variant> *a, const std::pair<const Dig, Variant> *b) const {
    int cmp = a->second.chr.compare(b->second.chr);
    if (cmp == 0) return a->second.pos < b->second.pos;
    return cmp < 0;
  }
};

}  // namespace

// The reference keeps one std::map, of which only two things are observable: addVar's per-key rule and the iteration order at the
// end (it feeds printToVCF's unstable sort).  Here the table is cut into 16 by the first hex digit of the key (the top four bits of
// the digest) and each shard is a HASH table (round 4: rank 0 replays 180 k records per 8-GPU step, and a red-black tree of
// 32-byte keys cost it ~0.15 us per record); lancet_vdb_vcf sorts every shard's entries by key -- shard 0..15 in turn is then the
// single map's iteration order.  A batch of records is inserted one shard per thread; each shard sees its records in arrival
// order: addVar's "larger total coverage replaces, first wins ties" is decided per key, and a key lives in exactly one shard.
static const int VDB_SHARDS = 16;
struct lancet_vdb {
  lancet_filters fs;
  bool lr = false;             // VariantDB_t::LR_MODE
  struct alignas(128) Shard : std::unordered_map<Dig, Variant, DigHash> {};      // (a shard per cache line pair: neighbouring shards are filled by different threads)
  Shard db[VDB_SHARDS];
  size_t size() const { size_t n = 0; for (int s = 0; s < VDB_SHARDS; ++s) n += db[s].size(); return n; }
};
static int vdb_threads() {
  static const int n = [] {
    const char *e = getenv("LANCET_VDB_THREADS");
    int t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return t < 1 ? 1 : (t > (e ? VDB_SHARDS : 4) ? (e ? VDB_SHARDS : 4) : t);   // 4 unless asked: a record costs ~0.2 us, waking more threads costs more
  }();
  return n;
}

extern "C" {

void lancet_filters_default(lancet_filters *f) {   // reference src/Lancet.cc:627-637
  memset(f, 0, sizeof(*f));
  f->min_phred_fisher_str = 25; f->min_phred_fisher = 5; f->min_cov_normal = 10; f->max_cov_normal = 1000000; f->min_cov_tumor = 4;
  f->max_cov_tumor = 1000000; f->min_vaf_tumor = 0.04; f->max_vaf_normal = 0; f->min_alt_cnt_tumor = 3; f->max_alt_cnt_normal = 0; f->min_strand_bias = 1;
}
lancet_vdb *lancet_vdb_create(const lancet_filters *f) { lancet_vdb *d = new lancet_vdb(); if (f) d->fs = *f; else lancet_filters_default(&d->fs); return d; }
void lancet_vdb_destroy(lancet_vdb *db) { delete db; }
uint32_t lancet_vdb_size(const lancet_vdb *db) { return db ? (uint32_t)db->size() : 0; }

// keys of n records (step 1 of addVar: Variant_t's normalisation, getSignature, sha256), threads [0, T)
static void vdb_keys(const lancet_variant *v, uint32_t n, const char *blob, const char *const *chr_names, Dig *key, int T) {
  auto body = [&](int t) {
    std::string ref, alt, sig; int pos; char type; unsigned short len;
    for (uint32_t i = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T); i < hi; ++i) {
      Variant::normalise(v[i], blob, ref, alt, pos, type, len);
      Variant::signature_into(sig, chr_names[v[i].chr_id], pos, type, len, ref, alt);
      sha256_bin(sig.data(), sig.size(), key[i]);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(body, t);
  body(0);
  for (auto &x : th) x.join();
}

static int vdb_add(lancet_vdb *db, const lancet_variant *v, const lancet_variant_lr *lr, uint32_t n, const char *blob, const uint32_t *bx_blob,
                   const char *const *bx_names, uint32_t n_bx, const char *const *chr_names, int32_t n_chr, const uint8_t *prekeys = nullptr) {
  if (!db || (n && (!v || !blob))) return LANCET_E_ARG;
  if (lr) db->lr = true;
  for (uint32_t i = 0; i < n; ++i) if (v[i].chr_id < 0 || v[i].chr_id >= n_chr) return LANCET_E_ARG;
  const int T = n >= 32768 ? vdb_threads() : 1;
  auto run = [&](auto &&fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(fn, t);
    fn(0);
    for (auto &x : th) x.join();
  };
  // 1. the key of every record: normalise into per-thread buffers, sha256 of the signature (no allocation per record)
  static const bool timing = getenv("LANCET_VDB_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t0 = now();
  std::vector<Dig> own;
  const Dig *key = (const Dig *)prekeys;                          // (the records came with their keys: lancet_vdb_keys on the rank that made them)
  if (!key) { own.resize(n); vdb_keys(v, n, blob, chr_names, own.data(), T); key = own.data(); }
  auto t1 = now();
  // 2. addVar (src/VariantDB.cc:28-91), shard s on thread s mod T, records in arrival order.  Large batches: the records' indices are
  //    dealt out by shard first (a counting sort, arrival order kept), so that a thread walks its own records only.
  std::vector<uint32_t> by_shard; uint32_t sh_at[VDB_SHARDS + 1] = {0};
  if (T > 1) {
    uint32_t cnt[VDB_SHARDS] = {0};
    for (uint32_t i = 0; i < n; ++i) ++cnt[key[i].b[0] >> 4];
    for (int q = 0; q < VDB_SHARDS; ++q) sh_at[q + 1] = sh_at[q] + cnt[q];
    by_shard.resize(n);
    uint32_t fill[VDB_SHARDS]; memcpy(fill, sh_at, sizeof(fill));
    for (uint32_t i = 0; i < n; ++i) by_shard[fill[key[i].b[0] >> 4]++] = i;
    for (int q = 0; q < VDB_SHARDS; ++q) db->db[q].reserve(db->db[q].size() + cnt[q] / 3 + 16);      // (a variant is met by ~4 overlapping windows)
  }
  auto t1b = now();
  run([&](int t) {
    for (int shq = T > 1 ? t : 0; shq < (T > 1 ? VDB_SHARDS : 1); shq += T)
    for (uint32_t j = T > 1 ? sh_at[shq] : 0, je = T > 1 ? sh_at[shq + 1] : n; j < je; ++j) {
      const uint32_t i = T > 1 ? by_shard[j] : j;
      const int sh = key[i].b[0] >> 4;
      auto &m = db->db[sh];
      auto it = m.find(key[i]);
      if (it != m.end()) {
        if (it->second.tot() < Variant::tot_of(v[i])) {          // keep the entry with the larger total coverage; first wins ties
          Variant &o = it->second;
          o.kmer = v[i].kmer;
          o.rcn_f = v[i].cov[0]; o.rcn_r = v[i].cov[1]; o.rct_f = v[i].cov[2]; o.rct_r = v[i].cov[3];
          o.acn_f = v[i].cov[4]; o.acn_r = v[i].cov[5]; o.act_f = v[i].cov[6]; o.act_r = v[i].cov[7];
          if (lr) {
            Variant nv; nv.set_lr(lr[i], bx_blob, bx_names, n_bx);
            for (int q = 0; q < 12; ++q) o.hp[q] = nv.hp[q];                       // src/VariantDB.cc:73-76
            for (int q = 0; q < 4; ++q) o.bx[q] = nv.bx[q];                        // :78-83
          } else {
            for (int q = 0; q < 12; ++q) o.hp[q] = 0;
            if (db->lr) for (int q = 0; q < 4; ++q) o.bx[q].clear();
          }
        }
      } else {
        auto ins = m.emplace(std::piecewise_construct, std::forward_as_tuple(key[i]), std::forward_as_tuple(chr_names[v[i].chr_id], v[i], blob)).first;
        if (lr) ins->second.set_lr(lr[i], bx_blob, bx_names, n_bx);
      }
    }
  });
  if (timing) fprintf(stderr, "  (dealing the records out by shard: %.2f ms)\n", std::chrono::duration<double, std::milli>(t1b - t1).count());
  if (timing) fprintf(stderr, "lancet_vdb_add: %u records%s, %d threads: keys %.2f ms, insert %.2f ms\n", n, prekeys ? " (keyed)" : "", T, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(now() - t1).count());
  return LANCET_OK;
}
int lancet_vdb_add(lancet_vdb *db, const lancet_variant *v, uint32_t n, const char *blob, const char *const *chr_names, int32_t n_chr) {
  return vdb_add(db, v, nullptr, n, blob, nullptr, nullptr, 0, chr_names, n_chr);
}
int lancet_vdb_add_lr(lancet_vdb *db, const lancet_variant *v, const lancet_variant_lr *lr, uint32_t n, const char *blob, const uint32_t *bx_blob,
                      const char *const *bx_names, uint32_t n_bx, const char *const *chr_names, int32_t n_chr) {
  if (n && !lr) return LANCET_E_ARG;
  if (!db) return LANCET_E_ARG;
  db->lr = true;
  return vdb_add(db, v, lr, n, blob, bx_blob, bx_names, n_bx, chr_names, n_chr);
}

/* The 32-byte key of every record as VariantDB_t::addVar computes it (reference src/VariantDB.cc:36-40: sha256 of getSignature() of the
 * normalised Variant_t), 32 raw digest bytes per record (the hex string's order is their memcmp order).  Computed where the records are
 * made -- every rank of a multi-GPU run -- so that the rank that owns the VariantDB only inserts (lancet_vdb_add_keyed). */
int lancet_vdb_keys(const lancet_variant *v, uint32_t n, const char *blob, const char *const *chr_names, int32_t n_chr, uint8_t *keys) {
  if (n && (!v || !blob || !keys || !chr_names)) return LANCET_E_ARG;
  for (uint32_t i = 0; i < n; ++i) if (v[i].chr_id < 0 || v[i].chr_id >= n_chr) return LANCET_E_ARG;
  vdb_keys(v, n, blob, chr_names, (Dig *)keys, n >= 32768 ? vdb_threads() : 1);
  return LANCET_OK;
}
/* Which of n keyed records (in the order they would be added) can change a VariantDB at all: per key, addVar keeps the strings of the
 * FIRST record and the k-mer size / coverages (/ haplotype counts / barcode sets) of the first record that reaches the key's largest
 * total coverage (src/VariantDB.cc:45-88: a strictly larger total replaces, the first wins ties) -- every other record of the key is a
 * no-op whatever comes before or after it, in this stream or in another rank's.  keep[i] = 1 for those one or two records per key.
 * A rank applies this to its own records before they travel: a variant is met by ~4.5 overlapping windows, so ~2.5x fewer records
 * reach the rank that merges, and the merged database is the same (the argument holds for any split of the windows over the ranks
 * that keeps each rank's records in their global order). */
int lancet_vdb_reduce(const lancet_variant *v, const uint8_t *keys, uint32_t n, uint8_t *keep) {
  if (n && (!v || !keys || !keep)) return LANCET_E_ARG;
  struct Ent { uint32_t first, best; int best_tot; };
  std::unordered_map<Dig, Ent, DigHash> m;
  m.reserve(n / 3 + 16);
  const Dig *key = (const Dig *)keys;
  for (uint32_t i = 0; i < n; ++i) {
    keep[i] = 0;
    const int tot = Variant::tot_of(v[i]);
    auto it = m.find(key[i]);
    if (it == m.end()) m.emplace(key[i], Ent{i, i, tot});
    else if (it->second.best_tot < tot) { it->second.best = i; it->second.best_tot = tot; }
  }
  for (auto &kv : m) { keep[kv.second.first] = 1; keep[kv.second.best] = 1; }
  return LANCET_OK;
}
int lancet_vdb_add_keyed(lancet_vdb *db, const lancet_variant *v, const lancet_variant_lr *lr, const uint8_t *keys, uint32_t n, const char *blob,
                         const uint32_t *bx_blob, const char *const *bx_names, uint32_t n_bx, const char *const *chr_names, int32_t n_chr) {
  if (!db || (n && !keys)) return LANCET_E_ARG;
  if (lr) db->lr = true;
  return vdb_add(db, v, lr, n, blob, bx_blob, bx_names, n_bx, chr_names, n_chr, keys);
}

char *lancet_vdb_vcf(lancet_vdb *db, const char *version, const char *cmdline, const char *reference, const char *date_line,
                     const char *sample_normal, const char *sample_tumor) {
  if (!db) return nullptr;
  const lancet_filters &fs = db->fs;
  std::ostringstream hdr;
  hdr << "##fileformat=VCFv4.2\n";
  if (date_line) hdr << "##fileDate=" << date_line;
  hdr << "##source=lancet " << (version ? version : "1.1.0, October 18 2019") << "\n";
  if (cmdline) hdr << "##cmdline=" << cmdline << "\n";
  if (reference) hdr << "##reference=" << reference << "\n";
  hdr << "##INFO=<ID=FETS,Number=1,Type=Float,Description=\"Phred-scaled p-value of the Fisher's exact test for tumor-normal allele counts\">\n"
         "##INFO=<ID=SOMATIC,Number=0,Type=Flag,Description=\"Somatic mutation\">\n"
         "##INFO=<ID=SHARED,Number=0,Type=Flag,Description=\"Shared mutation betweem tumor and normal\">\n"
         "##INFO=<ID=NORMAL,Number=0,Type=Flag,Description=\"Mutation present only in the normal\">\n"
         "##INFO=<ID=NONE,Number=0,Type=Flag,Description=\"Mutation not supported by data\">\n"
         "##INFO=<ID=KMERSIZE,Number=1,Type=Integer,Description=\"K-mer size used to assemble the locus\">\n"
         "##INFO=<ID=SB,Number=1,Type=Float,Description=\"Strand bias score: phred-scaled p-value of the Fisher's exact test for the forward/reverse read counts in the tumor\">\n"
         "##INFO=<ID=MS,Number=1,Type=String,Description=\"Microsatellite mutation (format: #LEN#MOTIF)\">\n"
         "##INFO=<ID=LEN,Number=1,Type=Integer,Description=\"Variant size in base pairs\">\n"
         "##INFO=<ID=TYPE,Number=1,Type=String,Description=\"Variant type (snv, del, ins, complex)\">\n";
  if (db->lr)
    hdr << "##INFO=<ID=HPS,Number=1,Type=Float,Description=\"Haplotype score for the T/N pair: phred-scaled p-value of the Fisher's exact test of the total counts of the two haplotype in the tumor-normal pair\">\n"
           "##INFO=<ID=HPSN,Number=1,Type=Float,Description=\"Normal haplotype score: phred-scaled p-value of the Fisher's exact test for ref/alt haplotype counts in the normal\">\n"
           "##INFO=<ID=HPST,Number=1,Type=Float,Description=\"Tumor haplotype score: phred-scaled p-value of the Fisher's exact test for ref/alt haplotype counts in the tumor\">\n";
  hdr << "##FILTER=<ID=LowCovNormal,Description=\"Low coverage in the normal (<" << fs.min_cov_normal << ")\">\n"
         "##FILTER=<ID=HighCovNormal,Description=\"High coverage in the normal (>" << fs.max_cov_normal << ")\">\n"
         "##FILTER=<ID=LowCovTumor,Description=\"Low coverage in the tumor (<" << fs.min_cov_tumor << ")\">\n"
         "##FILTER=<ID=HighCovTumor,Description=\"High coverage in the tumor (>" << fs.max_cov_tumor << ")\">\n"
         "##FILTER=<ID=LowVafTumor,Description=\"Low variant allele frequency in the tumor (<" << fs.min_vaf_tumor << ")\">\n"
         "##FILTER=<ID=HighVafNormal,Description=\"High variant allele frequency in the normal (>" << fs.max_vaf_normal << ")\">\n"
         "##FILTER=<ID=LowAltCntTumor,Description=\"Low alternative allele count in the tumor (<" << fs.min_alt_cnt_tumor << ")\">\n"
         "##FILTER=<ID=HighAltCntNormal,Description=\"High alternative allele count in the normal (>" << fs.max_alt_cnt_normal << ")\">\n"
         "##FILTER=<ID=LowFisherScore,Description=\"Low Fisher's exact test score for tumor-normal allele counts (<" << fs.min_phred_fisher << ")\">\n"
         "##FILTER=<ID=LowFisherSTR,Description=\"Low Fisher's exact test score for tumor-normal STR allele counts (<" << fs.min_phred_fisher_str << ")\">\n"
         "##FILTER=<ID=StrandBias,Description=\"Strand bias: # of non-reference reads in either forward or reverse strand below threshold (<" << fs.min_strand_bias << ")\">\n"
         "##FILTER=<ID=STR,Description=\"Microsatellite mutation\">\n";
  if (db->lr) hdr << "##FILTER=<ID=MultiHP,Description=\"Supporting reads from multiple haplotypes based on linked-reads analysis\">\n";
  hdr << "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
         "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Depth\">\n"
         "##FORMAT=<ID=AD,Number=.,Type=Integer,Description=\"Allele depth: # of supporting ref,alt reads at the site\">\n"
         "##FORMAT=<ID=SR,Number=.,Type=Integer,Description=\"Strand counts for ref: # of supporting forward,reverse reads for reference allele\">\n"
         "##FORMAT=<ID=SA,Number=.,Type=Integer,Description=\"Strand counts for alt: # of supporting forward,reverse reads for alterantive allele\">\n";
  if (db->lr)
    hdr << "##FORMAT=<ID=BX,Number=.,Type=String,Description=\"Barcodes supporting ref and alt alleles\">\n"
           "##FORMAT=<ID=HPR,Number=.,Type=Integer,Description=\"Haplotype counts for ref: # of reads supporting reference allele in haplotype 1, 2, and 0 respectively (0 = unassigned)\">\n"
           "##FORMAT=<ID=HPA,Number=.,Type=Integer,Description=\"Haplotype counts for alt: # of reads supporting alternative allele in haplotype 1, 2, and 0 respectively (0 = unassigned)\">\n";
  hdr << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" << (sample_normal ? sample_normal : "NORMAL") << "\t" << (sample_tumor ? sample_tumor : "TUMOR") << "\n";
  // printToVCF copies the map into a vector in key order and std::sorts it by (chr, pos) -- an unstable sort, so the
  // order of equal positions depends on that starting order: shard 0..15 in turn reproduces it.
  std::vector<const std::pair<const Dig, Variant> *> vec;
  vec.reserve(db->size());
  for (int s = 0; s < VDB_SHARDS; ++s) {                       // (a shard's entries in key order = that stretch of the reference's map)
    const size_t at = vec.size();
    for (auto &kv : db->db[s]) vec.push_back(&kv);
    std::sort(vec.begin() + (ptrdiff_t)at, vec.end(), [](const std::pair<const Dig, Variant> *a, const std::pair<const Dig, Variant> *b) { return a->first < b->first; });
  }
  std::sort(vec.begin(), vec.end(), byPos());
  std::string out = hdr.str();
  for (auto *kv : vec) out += kv->second.vcf(fs);
  char *r = (char *)malloc(out.size() + 1);
  if (!r) return nullptr;
  memcpy(r, out.c_str(), out.size() + 1);
  return r;
}
void lancet_free(void *p) { free(p); }

}  // extern "C"
