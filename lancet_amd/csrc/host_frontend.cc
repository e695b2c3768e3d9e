// host_frontend.cc -- native host side in front of the engine: BGZF/BAM + FASTA input, window tiling, the per-window
// filters and read selection of Microassembler::processReads, SoA batch assembly (include/lancet_host.h).
// Pure CPU code (std::thread over windows); nothing here computes what the engine computes.
//
// reference: src/Lancet.cc:189-316 (loadRefs), src/Microassembler.cc:253-432 (isActiveRegion), :436-655 (extractReads),
//            :779-842 (processReads), src/util.cc:295-315 (isRepeat), :428-483 (parseMD), :167-187 (isAmbiguos).
#include "../../include/lancet_host.h"

#include <sys/types.h>
#include <zlib.h>
#include <dlfcn.h>

#include <algorithm>
#include <memory>
#include <cerrno>
#include <atomic>
#include <cctype>
#include <chrono>
#include <sys/stat.h>
#include "host_pack.h"
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Read {
  int32_t pos0;               // 0-based leftmost position (BamAlignment::Position)
  int32_t ref_len;            // reference bases covered (M D N = X): GetEndPosition() - Position
  uint16_t flag;
  uint8_t mapq;
  uint8_t has_md, has_xa, xt_is_R, has_qual;
  int32_t hp;                 // HP:i (0 when absent)
  float as, xs;               // -1 when absent (extractReads initialises them so)
  uint32_t cig_off, n_cig;    // into Sample::cigar (BAM encoding: len << 4 | op)
  uint32_t seq_off, l_seq;    // into Sample::seq / qual (ASCII, phred+33)
  uint32_t name_off;          // into Sample::text (NUL terminated)
  uint32_t md_off, bx_off;    // into Sample::text (NUL terminated; bx "null" when absent)
  uint32_t rg_off;            // RG:Z value in Sample::text, 0xFFFFFFFF when the alignment has no RG tag
  int32_t chr;                // index into lancet_host::chroms
};

// A byte buffer that grows without initialising what it grows by: the bases / qualities / text of a scan are tens of MB per slab, a
// std::string would zero-fill them (and first touch every page) on the one thread that resizes, before the decoding threads write
// every byte anyway.  realloc moves large blocks by remapping.
struct Bytes {
  char *p = nullptr; size_t n = 0, cap = 0;
  Bytes() {}
  Bytes(const Bytes &) = delete; Bytes &operator=(const Bytes &) = delete;
  Bytes(Bytes &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
  Bytes &operator=(Bytes &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
  ~Bytes() { free(p); }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  void reserve(size_t want) { if (want > cap) { size_t c = cap + cap / 2; if (c < want) c = want; if (c < 4096) c = 4096; char *q = (char *)realloc(p, c); if (!q) throw std::bad_alloc(); p = q; cap = c; } }
  void resize(size_t want) { reserve(want); n = want; }
  void append(const char *s, size_t len) { reserve(n + len + 1); memcpy(p + n, s, len); n += len; }
  void append(const std::string &s) { append(s.data(), s.size()); }
  void push_back(char c) { reserve(n + 1); p[n++] = c; }
  void clear() { n = 0; }
  void erase_front(size_t k) { if (k >= n) { n = 0; return; } memmove(p, p + k, n - k); n -= k; }
  char *data() { return p; }
  const char *data() const { return p; }
  const char *c_str() const { return p; }        // (every string in it is NUL terminated by its writer)
  char &operator[](size_t i) { return p[i]; }
  const char &operator[](size_t i) const { return p[i]; }
};

struct Sample {
  std::vector<Read> reads;
  std::vector<int32_t> starts;      // pos0 per read (region seek)
  std::vector<uint32_t> cigar;
  Bytes seq, qual, text;
  std::string sample_name = "NA";
  std::vector<std::pair<std::string, int32_t>> refs;
  std::string path;
  int first_has_md = 1;             // MD on the first alignment of the file
  // lancet_host_batch_packed: every alignment trimmed and packed ONCE (it is in about six windows), for the thresholds pk_qtrim / pk_qcall
  std::vector<uint32_t> pk_tlen, pk_bw, pk_gw;        // trimmed length; first word of its bases / quality mask in pk_bases / pk_good
  Bytes pk_bases, pk_good;
  int pk_qtrim = -1, pk_qcall = -1; size_t pk_n = 0;
  std::unique_ptr<std::atomic<uint8_t>[]> pk_state;   // per alignment: 0 not packed yet, 1 a thread is packing it, 2 packed (pack_read_once)
  size_t u_lo = 1, u_hi = 0;                           // [u_lo, u_hi]: the entries of uidx the last batch wrote (all others are 0xFFFFFFFF)
  std::vector<uint32_t> uidx;                          // lancet_host_batch_packed, reads stored once: per alignment its place among the batch's distinct reads of this sample (0xFFFFFFFF: in no window of the batch)
  std::vector<std::pair<size_t, size_t>> span;   // per contig of the tiling: its reads [first, last) (file order = coordinate order)
};

struct Window { std::string hdr; int32_t start, end; std::string seq; int chr; };
struct Sel { uint32_t idx; uint8_t mate, strand, mapped; };
struct RSel { uint8_t smp; Sel s; };   // a selected read: sample (1 tumor, 0 normal) + its per-window attributes

bool read_file(const std::string &path, std::string *out, std::string *err) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) { *err = "cannot open " + path; return false; }
  std::string buf;
  char tmp[1 << 16];
  size_t n;
  while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
  fclose(f);
  out->swap(buf);
  return true;
}

unsigned host_threads(int items);

inline int32_t rd_i32(const unsigned char *p) { int32_t v; memcpy(&v, p, 4); return v; }
inline uint32_t rd_u32(const unsigned char *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd_u16(const unsigned char *p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint64_t rd_u64(const unsigned char *p) { uint64_t v; memcpy(&v, p, 8); return v; }

// BGZF (SAM spec 4.1): a series of gzip members, each with the BC extra subfield holding the block size.  The file is
// read in slabs of whole blocks from a compressed offset onwards; the blocks of a slab are independent deflate streams
// and are inflated on the host threads.  Memory in flight is one slab, whatever the size of the file.
// libdeflate (when the loader finds libdeflate.so.0; no header needed for three entry points) inflates a BGZF block two to three times faster
// than zlib's inflate -- and inflating is what the input decoding waits for.  Raw deflate streams of known output size; zlib otherwise.
struct LibDeflate {
  void *(*alloc)() = nullptr;
  int (*inflate)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;      // 0 = LIBDEFLATE_SUCCESS
  void (*release)(void *) = nullptr;
  LibDeflate() {
    if (const char *e = getenv("LANCET_HOST_ZLIB")) { if (atoi(e) != 0) return; }         // (LANCET_HOST_ZLIB=1: zlib, for comparison)
    void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
    inflate = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
    release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
    if (!alloc || !inflate || !release) { alloc = nullptr; inflate = nullptr; release = nullptr; }
  }
};
static const LibDeflate &libdeflate() { static const LibDeflate L; return L; }

class BgzfReader {
 public:
  ~BgzfReader() { if (f_) fclose(f_); }
  bool open(const std::string &path, std::string *err) {
    f_ = fopen(path.c_str(), "rb");
    if (!f_) { *err = "cannot open " + path; return false; }
    return true;
  }
  // continue at a BAM virtual offset (compressed offset << 16 | offset inside the inflated block)
  bool seek(uint64_t voff, std::string *err) {
    coff_ = voff >> 16; ubuf_.clear(); upos_ = 0; eof_ = false; slab_ = slab_max() < (1u << 20) ? slab_max() : (1u << 20);
    const size_t in_block = (size_t)(voff & 0xFFFFu);
    if (in_block == 0) return true;
    if (!fill(err)) { if (err->empty()) *err = "index points past the end of the file"; return false; }
    if (in_block > ubuf_.size()) { *err = "index points outside a BGZF block"; return false; }
    upos_ = in_block;
    return true;
  }
  // at least n bytes at the cursor; false at the end of the file (err stays empty) or on an error (err set)
  bool need(size_t n, std::string *err) {
    while (ubuf_.size() - upos_ < n) if (!fill(err)) return false;
    return true;
  }
  bool have(size_t n) const { return ubuf_.size() - upos_ >= n; }     // n bytes at the cursor without reading on (a read-on moves the buffer)
  const unsigned char *cur() const { return (const unsigned char *)ubuf_.data() + upos_; }
  const unsigned char *base() const { return (const unsigned char *)ubuf_.data(); }
  size_t offset() const { return upos_; }
  void advance(size_t n) { upos_ += n; }
  uint64_t compressed_pos() const { return coff_; }
  uint64_t inflated_total() const { return inflated_; }
  double fill_seconds() const { return fill_s_; }

 private:
  bool fill(std::string *err) {
    struct Acc { double *t; std::chrono::steady_clock::time_point t0; ~Acc() { *t += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } acc{&fill_s_, std::chrono::steady_clock::now()};
    if (eof_) return false;
    if (upos_ > 0) { ubuf_.erase_front(upos_); upos_ = 0; }
    cbuf_.resize(slab_);
    if (fseeko(f_, (off_t)coff_, SEEK_SET) != 0) { *err = "seek failed"; return false; }
    const size_t got = fread(&cbuf_[0], 1, slab_, f_);
    if (got == 0) { eof_ = true; return false; }
    struct Blk { size_t cdata, clen, opos; unsigned isize; };
    std::vector<Blk> blks;
    size_t p = 0, total = 0;
    const unsigned char *raw = (const unsigned char *)cbuf_.data();
    while (p + 18 <= got) {
      const unsigned char *h = raw + p;
      if (h[0] != 31 || h[1] != 139) { *err = "not a BGZF block"; return false; }
      const unsigned xlen = h[10] | (h[11] << 8);
      unsigned bsize = 0; bool found = false;
      for (unsigned q = 12; q + 4 <= 12 + xlen && p + q + 6 <= got;) {
        const unsigned slen = h[q + 2] | (h[q + 3] << 8);
        if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2) { bsize = (h[q + 4] | (h[q + 5] << 8)) + 1u; found = true; }
        q += 4 + slen;
      }
      if (!found) { if (p + 12 + xlen > got) break; *err = "not a BGZF block"; return false; }
      if (bsize < 12 + xlen + 8) { *err = "truncated BGZF block"; return false; }
      if (p + bsize > got) break;                       // the rest of this block is in the next slab
      const unsigned isize = h[bsize - 4] | (h[bsize - 3] << 8) | (h[bsize - 2] << 16) | ((unsigned)h[bsize - 1] << 24);
      if (isize > 65536u) { *err = "corrupt BGZF block"; return false; }
      blks.push_back(Blk{p + 12 + xlen, (size_t)bsize - 12 - xlen - 8, total, isize});
      total += isize;
      p += bsize;
    }
    if (blks.empty()) { *err = "truncated BGZF block"; return false; }   // fewer bytes than one block: the file ends inside a block
    const size_t base = ubuf_.size();
    ubuf_.resize(base + total);
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    const LibDeflate &LD = libdeflate();
    auto work = [&]() {
      void *dec = LD.alloc ? LD.alloc() : nullptr;
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= blks.size()) break;
        const Blk &b = blks[i];
        if (b.isize == 0) continue;
        if (dec) {
          size_t got_out = 0;
          if (LD.inflate(dec, raw + b.cdata, b.clen, &ubuf_[base + b.opos], b.isize, &got_out) != 0 || got_out != b.isize) bad = 1;
          continue;
        }
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; continue; }
        zs.next_in = (Bytef *)raw + b.cdata; zs.avail_in = (uInt)b.clen;
        zs.next_out = (Bytef *)&ubuf_[base + b.opos]; zs.avail_out = b.isize;
        const int rc = inflate(&zs, Z_FINISH);
        if (rc != Z_STREAM_END || zs.total_out != b.isize) bad = 1;
        inflateEnd(&zs);
      }
      if (dec) LD.release(dec);
    };
    const unsigned nt = host_threads((int)(blks.size() / 8 + 1));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (bad) { *err = "corrupt BGZF block"; return false; }
    coff_ += p; inflated_ += total;
    if (slab_ < slab_max()) slab_ *= 4;
    return true;
  }
  // compressed bytes per read: grows from 256 KB to 16 MB; LANCET_HOST_SLAB_KB caps it (tests use it to make a small file span many slabs)
  static size_t slab_max() { if (const char *e = getenv("LANCET_HOST_SLAB_KB")) { const long v = atol(e); if (v >= 64) return (size_t)v << 10; } return 16u << 20; }
  FILE *f_ = nullptr;
  uint64_t coff_ = 0, inflated_ = 0;
  double fill_s_ = 0;
  Bytes cbuf_, ubuf_;
  size_t upos_ = 0, slab_ = slab_max() < (1u << 18) ? slab_max() : (1u << 18);
  bool eof_ = false;
};

// A .bai (SAM spec 5.2), reduced to what finding "the alignments that START at or after a position" needs.  Per reference
// sequence: the linear index (for every 16 kb stretch the smallest virtual offset of an alignment overlapping it) when the
// indexer wrote one (samtools does; `bamtools index` leaves it empty), the first chunk of every 16 kb leaf bin
// (4681 + stretch: alignments that lie inside that stretch), and the first chunk of all.  Each gives a lower bound on the
// file position of the first alignment starting at or after `lo`; the reader streams from there and skips what starts
// before `lo`.  The bins' chunk lists are not needed because only alignment starts inside a range are wanted.
struct RefIndex {
  std::vector<uint64_t> linear;
  std::unordered_map<uint32_t, uint64_t> leaf_first;
  uint64_t first = 0;                                   // 0: no alignment on this reference sequence
  uint64_t start_for(int32_t lo) const {
    const size_t w = (size_t)(lo > 0 ? lo : 0) >> 14;
    if (!linear.empty()) { for (size_t i = w; i < linear.size(); ++i) if (linear[i]) return linear[i]; return 0; }
    for (size_t i = w; i-- > 0;) { auto it = leaf_first.find(4681u + (uint32_t)i); if (it != leaf_first.end()) return it->second; }   // a leaf before lo's: its alignments start earlier
    return first;
  }
};
bool load_bai(const std::string &bam, std::vector<RefIndex> *idx) {
  std::string raw, err;
  std::string cand[2] = {bam + ".bai", std::string()};
  if (bam.size() > 4 && bam.compare(bam.size() - 4, 4, ".bam") == 0) cand[1] = bam.substr(0, bam.size() - 4) + ".bai";
  bool got = false;
  for (const std::string &c : cand) if (!got && !c.empty()) { FILE *f = fopen(c.c_str(), "rb"); if (f) { fclose(f); got = read_file(c, &raw, &err); } }
  if (!got || raw.size() < 8 || memcmp(raw.data(), "BAI\1", 4) != 0) return false;
  const unsigned char *b = (const unsigned char *)raw.data();
  const size_t n = raw.size();
  size_t p = 4;
  const int32_t n_ref = rd_i32(b + p); p += 4;
  if (n_ref < 0 || (size_t)n_ref > n) return false;
  idx->assign((size_t)n_ref, RefIndex());
  for (int32_t r = 0; r < n_ref; ++r) {
    RefIndex &R = (*idx)[(size_t)r];
    if (p + 4 > n) return false;
    const int32_t n_bin = rd_i32(b + p); p += 4;
    for (int32_t i = 0; i < n_bin; ++i) {
      if (p + 8 > n) return false;
      const uint32_t bin = rd_u32(b + p);
      const int32_t n_chunk = rd_i32(b + p + 4); p += 8;
      if (n_chunk < 0 || p + 16 * (size_t)n_chunk > n) return false;
      if (bin < 37450u) for (int32_t c = 0; c < n_chunk; ++c) {          // (37450: samtools' metadata pseudo-bin)
        const uint64_t beg = rd_u64(b + p + 16 * (size_t)c);
        if (R.first == 0 || beg < R.first) R.first = beg;
        if (bin >= 4681u) { auto it = R.leaf_first.find(bin); if (it == R.leaf_first.end()) R.leaf_first.emplace(bin, beg); else if (beg < it->second) it->second = beg; }
      }
      p += 16 * (size_t)n_chunk;
    }
    if (p + 4 > n) return false;
    const int32_t n_intv = rd_i32(b + p); p += 4;
    if (n_intv < 0 || p + 8 * (size_t)n_intv > n) return false;
    R.linear.resize((size_t)n_intv);
    for (int32_t i = 0; i < n_intv; ++i) R.linear[(size_t)i] = rd_u64(b + p + 8 * (size_t)i);
    p += 8 * (size_t)n_intv;
  }
  return true;
}

// One alignment record (the block after block_size; SAM spec 4.2) -> Sample arrays.  Every length in the record is
// checked against the record's end before it is used.
bool decode_record(const unsigned char *b, size_t bs, int chr, Sample *S, std::string *err) {
  static const char SEQ[] = "=ACMGRSVTWYHKDBN";
  const size_t end = bs;
  const int32_t pos = rd_i32(b + 4);
  const unsigned l_name = b[8], mapq = b[9];
  const unsigned n_cig = rd_u16(b + 12), flag = rd_u16(b + 14);
  const int32_t l_seq = rd_i32(b + 16);
  if (l_seq < 0 || 32 + (size_t)l_name + 4 * (size_t)n_cig + (size_t)((l_seq + 1) / 2) + (size_t)l_seq > end) { *err = "alignment record with fields past its end"; return false; }
  // (offsets into the sample's arrays are 32 bits: ~28 M reads of 150 bases per sample and tiling)
  if ((uint64_t)S->seq.size() + (uint64_t)l_seq > 0xFFFFFFFFull || (uint64_t)S->text.size() + (uint64_t)bs > 0xFFFFFFFFull || (uint64_t)S->cigar.size() + n_cig > 0xFFFFFFFFull) {
    *err = "more than 4 GB of alignments in one tiling (tile fewer windows at a time)"; return false; }
  Read r; memset(&r, 0, sizeof r);
  r.pos0 = pos; r.flag = (uint16_t)flag; r.mapq = (uint8_t)mapq; r.as = -1.f; r.xs = -1.f; r.chr = chr;
  size_t q = 32;
  r.name_off = (uint32_t)S->text.size(); S->text.append((const char *)b + q, l_name ? strnlen((const char *)b + q, l_name - 1) : 0); S->text.push_back('\0'); q += l_name;
  r.cig_off = (uint32_t)S->cigar.size(); r.n_cig = n_cig;
  int32_t rl = 0;
  for (unsigned i = 0; i < n_cig; ++i) {
    const uint32_t c = rd_u32(b + q + 4 * i); S->cigar.push_back(c);
    const unsigned op = c & 15u;                      // MIDNSHP=X
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += (int32_t)(c >> 4);
  }
  r.ref_len = rl; q += 4 * (size_t)n_cig;
  r.seq_off = (uint32_t)S->seq.size(); r.l_seq = (uint32_t)l_seq;
  {
    const size_t o0 = S->seq.size();
    S->seq.resize(o0 + (size_t)l_seq); S->qual.resize(o0 + (size_t)l_seq);
    char *sp = &S->seq[0] + o0, *qp = &S->qual[0] + o0;
    for (int32_t i = 0; i + 1 < l_seq; i += 2) { const unsigned v = b[q + (size_t)(i >> 1)]; sp[i] = SEQ[v >> 4]; sp[i + 1] = SEQ[v & 15u]; }
    if (l_seq & 1) sp[l_seq - 1] = SEQ[b[q + (size_t)(l_seq >> 1)] >> 4];
    q += (size_t)((l_seq + 1) / 2);
    r.has_qual = (l_seq > 0 && b[q] != 0xFF) ? 1 : 0;
    if (r.has_qual) for (int32_t i = 0; i < l_seq; ++i) qp[i] = (char)(b[q + (size_t)i] + 33);
    else memset(qp, 0, (size_t)l_seq);       // unstored: bamtools fills (char)0xFF, negative = below every quality threshold; byte 0 is that for unsigned compares
    q += (size_t)l_seq;
  }
  r.md_off = 0; r.bx_off = 0; r.rg_off = 0xFFFFFFFFu;
  std::string bx = "null";
  while (q + 3 <= end) {                              // tags
    const char k0 = (char)b[q], k1 = (char)b[q + 1], t = (char)b[q + 2];
    q += 3;
    double num = 0; bool isnum = false; std::string sval; bool isstr = false;
    size_t sz = 0;
    switch (t) {
      case 'A':   // bamtools' GetTag(tag, std::string&) on a one-character tag: strlen over the raw tag block, i.e. the character
                  // and every byte after it up to the next NUL -- the bare character only when it is the record's last tag
                  // (src/api/BamAlignment.cpp); extractReads compares that string with "R" (src/Microassembler.cc:549-559)
        sval.assign((const char *)b + q, strnlen((const char *)b + q, end - q)); isstr = true; sz = 1; break;
      case 'c': sz = 1; if (q + sz <= end) { num = (int8_t)b[q]; isnum = true; } break;
      case 'C': sz = 1; if (q + sz <= end) { num = b[q]; isnum = true; } break;
      case 's': sz = 2; if (q + sz <= end) { int16_t v; memcpy(&v, b + q, 2); num = v; isnum = true; } break;
      case 'S': sz = 2; if (q + sz <= end) { num = rd_u16(b + q); isnum = true; } break;
      case 'i': sz = 4; if (q + sz <= end) { num = rd_i32(b + q); isnum = true; } break;
      case 'I': sz = 4; if (q + sz <= end) { num = rd_u32(b + q); isnum = true; } break;
      case 'f': sz = 4; if (q + sz <= end) { float v; memcpy(&v, b + q, 4); num = v; isnum = true; } break;
      case 'Z': case 'H': { const size_t l = strnlen((const char *)b + q, end - q); sval.assign((const char *)b + q, l); isstr = true; sz = l + 1; break; }
      case 'B': {
        if (q + 5 > end) { sz = 5; break; }
        const char st = (char)b[q]; const int32_t cnt = rd_i32(b + q + 1);
        const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
        if (cnt < 0) { *err = "BAM array tag with a negative count"; return false; }
        sz = 5 + es * (size_t)cnt; break;
      }
      default: *err = "unknown BAM tag type"; return false;
    }
    if (q + sz > end && !(t == 'Z' || t == 'H')) { *err = "BAM tag past the end of its record"; return false; }
    q += sz;
    if (k0 == 'A' && k1 == 'S' && isnum) r.as = (float)num;
    else if (k0 == 'X' && k1 == 'S' && isnum) r.xs = (float)num;
    else if (k0 == 'X' && k1 == 'T' && isstr) r.xt_is_R = (sval == "R");
    else if (k0 == 'X' && k1 == 'A' && isstr) r.has_xa = !sval.empty();
    else if (k0 == 'M' && k1 == 'D' && isstr) { r.has_md = 1; r.md_off = (uint32_t)S->text.size(); S->text.append(sval); S->text.push_back('\0'); }
    else if (k0 == 'B' && k1 == 'X' && isstr) { if (!sval.empty()) bx = sval; }
    else if (k0 == 'R' && k1 == 'G' && isstr) { r.rg_off = (uint32_t)S->text.size(); S->text.append(sval); S->text.push_back('\0'); }
    else if (k0 == 'H' && k1 == 'P' && isnum) r.hp = num > 0 ? (int32_t)num : 0;
  }
  r.bx_off = (uint32_t)S->text.size(); S->text.append(bx); S->text.push_back('\0');
  S->reads.push_back(r);
  S->starts.push_back(pos);
  return true;
}

// true when the record carries an MD tag (checkPresenceOfMDtag, reference src/util.cc:416-427, looks at the first alignment)
bool record_has_md(const unsigned char *b, size_t bs) {
  const unsigned l_name = b[8]; const unsigned n_cig = rd_u16(b + 12); const int32_t l_seq = rd_i32(b + 16);
  if (l_seq < 0) return false;
  size_t q = 32 + (size_t)l_name + 4 * (size_t)n_cig + (size_t)((l_seq + 1) / 2) + (size_t)l_seq;
  while (q + 3 <= bs) {
    const char k0 = (char)b[q], k1 = (char)b[q + 1], t = (char)b[q + 2];
    q += 3;
    if (k0 == 'M' && k1 == 'D') return true;
    size_t sz;
    switch (t) {
      case 'A': case 'c': case 'C': sz = 1; break;
      case 's': case 'S': sz = 2; break;
      case 'i': case 'I': case 'f': sz = 4; break;
      case 'Z': case 'H': sz = strnlen((const char *)b + q, bs - q) + 1; break;
      case 'B': { if (q + 5 > bs) return false; const char st = (char)b[q]; const int32_t cnt = rd_i32(b + q + 1); if (cnt < 0) return false;
                  sz = 5 + ((st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4) * (size_t)cnt; break; }
      default: return false;
    }
    q += sz;
  }
  return false;
}

// What the windows of a tiling can select from one contig: alignments that START in [lo, hi] (0-based), ranges sorted and disjoint.
struct Want { std::string chrom; int chr; std::vector<std::pair<int32_t, int32_t>> iv; };

// Decodes, in file order, the alignments the tiling can select.  With a .bai next to the BAM the reader jumps to each
// range through the linear index; without one it streams the file once and stops after the last range.
struct BaiCache { bool tried = false, have = false; std::vector<RefIndex> index; };
bool load_bam(const std::string &path, std::vector<Want> wants, Sample *S, std::string *err, BaiCache *bai = nullptr) {
  const bool timing = getenv("LANCET_HOST_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  BgzfReader in;
  if (!in.open(path, err)) return false;
  auto fail = [&](const std::string &m) { *err = path + ": " + (err->empty() ? m : *err); return false; };
  {   // room for what a file of this size can hold (address space only until it is written: growing a 25 MB block later costs more than decoding it)
    struct stat st;
    if (stat(path.c_str(), &st) == 0 && st.st_size > 0) {
      const size_t fs = (size_t)st.st_size;
      try { S->seq.reserve(std::min<size_t>(3 * fs, (size_t)1 << 30)); S->qual.reserve(std::min<size_t>(3 * fs, (size_t)1 << 30)); S->text.reserve(std::min<size_t>(fs, (size_t)1 << 29));
            S->reads.reserve(std::min<size_t>(fs / 48, (size_t)1 << 24)); S->starts.reserve(std::min<size_t>(fs / 48, (size_t)1 << 24)); S->cigar.reserve(std::min<size_t>(fs / 24, (size_t)1 << 25)); }
      catch (const std::bad_alloc &) {}
    }
  }
  if (!in.need(12, err)) return fail("not a BAM file");
  if (memcmp(in.cur(), "BAM\1", 4) != 0) { err->clear(); return fail("not a BAM file"); }
  const int32_t l_text = rd_i32(in.cur() + 4);
  if (l_text < 0 || !in.need(12 + (size_t)l_text, err)) { return fail("truncated BAM header"); }
  std::string text((const char *)in.cur() + 8, strnlen((const char *)in.cur() + 8, (size_t)l_text));
  const int32_t n_ref = rd_i32(in.cur() + 8 + (size_t)l_text);
  in.advance(12 + (size_t)l_text);
  if (n_ref < 0) return fail("truncated BAM header");
  std::unordered_map<std::string, int32_t> tid_of;
  for (int32_t i = 0; i < n_ref; ++i) {
    if (!in.need(4, err)) return fail("truncated BAM header");
    const int32_t ln = rd_i32(in.cur());
    if (ln < 1 || ln > (1 << 20) || !in.need(8 + (size_t)ln, err)) return fail("truncated BAM header");
    std::string name((const char *)in.cur() + 4, strnlen((const char *)in.cur() + 4, (size_t)ln - 1));
    const int32_t len = rd_i32(in.cur() + 4 + (size_t)ln);
    in.advance(8 + (size_t)ln);
    tid_of.emplace(name, i);
    S->refs.emplace_back(name, len);
  }
  {   // SM of the first @RG line
    size_t q = 0;
    bool got = false;
    while (q < text.size() && !got) {
      size_t e = text.find('\n', q); if (e == std::string::npos) e = text.size();
      if (text.compare(q, 3, "@RG") == 0) {
        size_t f = q;
        while (f < e) {
          size_t t = text.find('\t', f); if (t == std::string::npos || t > e) t = e;
          if (t - f > 3 && text.compare(f, 3, "SM:") == 0) { S->sample_name = text.substr(f + 3, t - f - 3); got = true; break; }
          f = t + 1;
        }
      }
      q = e + 1;
    }
  }
  // the first alignment of the file: does it carry MD?  (no alignment at all counts as yes, as in the reference)
  S->first_has_md = 1;
  if (in.need(4, err)) {
    const int32_t bs = rd_i32(in.cur());
    if (bs < 32 || bs > (256 << 20) || !in.need(4 + (size_t)bs, err)) return fail("truncated alignment record");
    S->first_has_md = record_has_md(in.cur() + 4, (size_t)bs) ? 1 : 0;
  } else if (!err->empty()) return fail("");
  // ranges in file order: by reference id, then position
  struct Rng { int32_t tid; int chr; int32_t lo, hi; };
  std::vector<Rng> rngs;
  size_t max_chr = 0;
  for (const Want &w : wants) {
    if ((size_t)w.chr + 1 > max_chr) max_chr = (size_t)w.chr + 1;
    auto it = tid_of.find(w.chrom);
    if (it == tid_of.end()) continue;                   // contig absent from this BAM: no reads (bamtools' SetRegion fails, the windows stay empty)
    for (auto &iv : w.iv) rngs.push_back(Rng{it->second, w.chr, iv.first, iv.second});
  }
  std::sort(rngs.begin(), rngs.end(), [](const Rng &a, const Rng &b) { return a.tid != b.tid ? a.tid < b.tid : a.lo < b.lo; });
  S->span.assign(max_chr, std::pair<size_t, size_t>(0, 0));
  std::vector<RefIndex> own;
  if (bai && !bai->tried) { bai->have = load_bai(path, &bai->index); bai->tried = true; }      // (lazy mode reads the index once, not once per batch)
  const bool indexed = bai ? bai->have : load_bai(path, &own);
  const std::vector<RefIndex> &index = bai ? bai->index : own;
  uint64_t seeks = 0;
  int32_t prev_tid = 0, prev_pos = -1; bool have_prev = false;
  bool at_end = false;
  // Records that pass the range test are only NOTED (offset in the inflated buffer, size); they are decoded, by several threads into
  // pieces that are then appended in order, before the buffer moves (a read-on, a seek) and at the end of a range.  Walking the chain of
  // block sizes is all that stays serial: with the inflate on the host threads already, decoding one record after the other was 90 % of
  // the load on a 64-thread host.
  struct Pend { size_t off; uint32_t bs; int chr; };
  std::vector<Pend> pend;
  double t_decode = 0; int n_flush = 0;
  auto flush = [&]() -> bool {
    if (pend.empty()) return true;
    const double tf0 = timing ? now() : 0; ++n_flush;
    struct Acc { double *t; double t0; bool on; std::function<double()> clk; ~Acc() { if (on) *t += clk() - t0; } } acc{&t_decode, tf0, timing, now};
    const unsigned nt = host_threads((int)(pend.size() / 2048 + 1));
    bool ok = true;
    if (nt <= 1) {
      for (const Pend &q : pend) if (!decode_record(in.base() + q.off, (size_t)q.bs, q.chr, S, err)) { ok = false; break; }
    } else {
      std::vector<Sample> piece(nt);
      std::vector<std::string> perr(nt);
      std::vector<char> pok(nt, 1);
      const unsigned char *base = in.base();
      auto work = [&](unsigned t) {
        const size_t lo = pend.size() * t / nt, hi = pend.size() * (t + 1) / nt;
        Sample &P = piece[t];
        P.reads.reserve(hi - lo); P.starts.reserve(hi - lo);
        { size_t bytes = 0; for (size_t i = lo; i < hi; ++i) bytes += pend[i].bs;          // (a record's bases / qualities / text are each shorter than the record)
          P.seq.reserve(bytes); P.qual.reserve(bytes); P.text.reserve(bytes / 2 + 64); P.cigar.reserve((hi - lo) * 4); }
        for (size_t i = lo; i < hi; ++i) if (!decode_record(base + pend[i].off, (size_t)pend[i].bs, pend[i].chr, &P, &perr[t])) { pok[t] = 0; return; }
      };
      std::vector<std::thread> th;
      for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
      work(0);
      for (auto &t : th) t.join();
      for (unsigned t = 0; t < nt && ok; ++t) if (!pok[t]) { *err = perr[t]; ok = false; }
      // append the pieces in order: sizes first, then every piece copies itself into place and shifts its reads' offsets
      std::vector<size_t> r0(nt + 1), c0(nt + 1), s0(nt + 1), t0(nt + 1);
      r0[0] = S->reads.size(); c0[0] = S->cigar.size(); s0[0] = S->seq.size(); t0[0] = S->text.size();
      for (unsigned t = 0; t < nt; ++t) { r0[t + 1] = r0[t] + piece[t].reads.size(); c0[t + 1] = c0[t] + piece[t].cigar.size(); s0[t + 1] = s0[t] + piece[t].seq.size(); t0[t + 1] = t0[t] + piece[t].text.size(); }
      if (ok && (s0[nt] > 0xFFFFFFFFull || t0[nt] > 0xFFFFFFFFull || c0[nt] > 0xFFFFFFFFull)) { *err = "more than 4 GB of alignments in one tiling (tile fewer windows at a time)"; ok = false; }
      if (ok) {
        S->reads.resize(r0[nt]); S->starts.resize(r0[nt]); S->cigar.resize(c0[nt]); S->seq.resize(s0[nt]); S->qual.resize(s0[nt]); S->text.resize(t0[nt]);
        auto place = [&](unsigned t) {
          const Sample &P = piece[t];
          if (!P.cigar.empty()) memcpy(&S->cigar[c0[t]], P.cigar.data(), 4 * P.cigar.size());
          if (!P.seq.empty()) { memcpy(&S->seq[s0[t]], P.seq.data(), P.seq.size()); memcpy(&S->qual[s0[t]], P.qual.data(), P.qual.size()); }
          if (!P.text.empty()) memcpy(&S->text[t0[t]], P.text.data(), P.text.size());
          for (size_t i = 0; i < P.reads.size(); ++i) {
            Read r = P.reads[i];
            r.cig_off += (uint32_t)c0[t]; r.seq_off += (uint32_t)s0[t]; r.name_off += (uint32_t)t0[t]; r.bx_off += (uint32_t)t0[t];
            if (r.has_md) r.md_off += (uint32_t)t0[t];
            if (r.rg_off != 0xFFFFFFFFu) r.rg_off += (uint32_t)t0[t];
            S->reads[r0[t] + i] = r; S->starts[r0[t] + i] = P.starts[i];
          }
        };
        std::vector<std::thread> th2;
        for (unsigned t = 1; t < nt; ++t) th2.emplace_back(place, t);
        place(0);
        for (auto &t : th2) t.join();
      }
    }
    pend.clear();
    return ok;
  };
  // an error of the walk is reported only after the records noted before it have been decoded: a damaged record earlier in the file wins
  auto fail_after = [&](const char *m) { if (!flush()) return fail(""); err->clear(); return fail(m); };
  for (size_t ri = 0; ri < rngs.size() && !at_end; ++ri) {
    const Rng &g = rngs[ri];
    if (indexed && (size_t)g.tid < index.size()) {
      const uint64_t voff = index[(size_t)g.tid].start_for(g.lo);
      if (voff == 0) continue;                          // no alignment on this contig from lo onwards
      if ((voff >> 16) > in.compressed_pos()) { if (!in.seek(voff, err)) return fail(""); ++seeks; have_prev = false; }
    }
    const size_t first = S->reads.size();              // (nothing is pending here: every range ends with a flush)
    for (;;) {
      if (!in.have(4)) { if (!flush()) return fail(""); if (!in.need(4, err)) { if (!err->empty()) return fail(""); at_end = true; break; } }
      const int32_t bs = rd_i32(in.cur());
      if (bs < 32) return fail_after("truncated alignment record");
      if (bs > (256 << 20)) return fail_after("alignment record of more than 256 MB (damaged block_size?)");     // (need() would buffer the rest of the file first)
      if (!in.have(4 + (size_t)bs)) { if (!flush()) return fail(""); if (!in.need(4 + (size_t)bs, err)) return fail("truncated alignment record"); }
      const unsigned char *rec = in.cur() + 4;
      const int32_t tid = rd_i32(rec), pos = rd_i32(rec + 4);
      if (tid >= 0) {
        if (have_prev && (tid < prev_tid || (tid == prev_tid && pos < prev_pos))) return fail_after("not coordinate sorted");
        prev_tid = tid; prev_pos = pos; have_prev = true;
      }
      if (tid < 0 || tid > g.tid || (tid == g.tid && pos > g.hi)) { if (tid < 0) at_end = true; break; }   // left for the next range
      if (tid == g.tid && pos >= g.lo) pend.push_back(Pend{in.offset() + 4, (uint32_t)bs, g.chr});
      in.advance(4 + (size_t)bs);
    }
    if (!flush()) return fail("");
    std::pair<size_t, size_t> &sp = S->span[(size_t)g.chr];
    if (S->reads.size() > first) { if (sp.second == sp.first) sp.first = first; sp.second = S->reads.size(); }
  }
  S->path = path;
  if (timing) fprintf(stderr, "[lancet_host] %s: %s, %zu ranges, %llu seeks, %llu MB inflated, %zu alignments kept, %.3f s (read + inflate %.3f s, decode %.3f s in %d steps)\n", path.c_str(),
                      indexed ? "indexed (.bai)" : "no .bai: streamed from the start", rngs.size(), (unsigned long long)seeks,
                      (unsigned long long)(in.inflated_total() >> 20), S->reads.size(), now() - t0, in.fill_seconds(), t_decode, n_flush);
  return true;
}

// FASTA: through the .fai when there is one (only the tiled stretches are read), else the whole file once.
struct FaiEnt { long len, off, linebases, linewidth; };
struct Fasta {
  std::string path;
  bool have_fai = false, whole_loaded = false;
  std::map<std::string, FaiEnt> fai;
  std::map<std::string, std::string> whole;
  bool open(const std::string &p, std::string *err) {
    path = p;
    std::string idx;
    { FILE *f = fopen((p + ".fai").c_str(), "rb"); if (f) { fclose(f); std::string e2; have_fai = read_file(p + ".fai", &idx, &e2); } }
    if (have_fai) {
      size_t q = 0;
      while (q < idx.size()) {
        size_t e = idx.find('\n', q); if (e == std::string::npos) e = idx.size();
        const std::string line = idx.substr(q, e - q);
        q = e + 1;
        if (line.empty()) continue;
        std::vector<std::string> t; size_t a = 0;
        while (a <= line.size()) { size_t b = line.find('\t', a); if (b == std::string::npos) b = line.size(); t.push_back(line.substr(a, b - a)); a = b + 1; }
        if (t.size() < 5) { have_fai = false; fai.clear(); break; }
        FaiEnt en{atol(t[1].c_str()), atol(t[2].c_str()), atol(t[3].c_str()), atol(t[4].c_str())};
        if (en.linebases <= 0 || en.linewidth < en.linebases) { have_fai = false; fai.clear(); break; }
        fai.emplace(t[0], en);
      }
      FILE *f = fopen(p.c_str(), "rb"); if (!f) { *err = "cannot open " + p; return false; } fclose(f);
    }
    if (!have_fai) return load_whole(err);
    return true;
  }
  // the whole file, once: no .fai, or a contig the .fai does not list (an index older than the file)
  bool load_whole(std::string *err) {
    if (whole_loaded) return true;
    {
      std::string buf;
      if (!read_file(path, &buf, err)) return false;
      std::string *cur = nullptr;
      size_t q = 0;
      while (q < buf.size()) {
        size_t e = buf.find('\n', q); if (e == std::string::npos) e = buf.size();
        size_t le = e; while (le > q && (buf[le - 1] == '\r' || buf[le - 1] == '\n')) --le;
        if (le > q && buf[q] == '>') {
          size_t w = q + 1; while (w < le && !isspace((unsigned char)buf[w])) ++w;
          cur = &whole[buf.substr(q + 1, w - q - 1)]; cur->clear();
        } else if (cur) cur->append(buf, q, le - q);
        q = e + 1;
      }
    }
    whole_loaded = true;
    return true;
  }
  long length(const std::string &name) {
    if (have_fai) { auto it = fai.find(name); if (it != fai.end()) return it->second.len; std::string e; if (!load_whole(&e)) return -1; }
    auto it = whole.find(name); return it == whole.end() ? -1 : (long)it->second.size();
  }
  // bases [sp, ep] (1-based, inclusive, already clipped to the contig)
  bool fetch(const std::string &name, long sp, long ep, std::string *out, std::string *err) const {
    out->clear();
    if (ep < sp) return true;
    if (!have_fai || fai.find(name) == fai.end()) { *out = whole.at(name).substr((size_t)(sp - 1), (size_t)(ep - sp + 1)); return true; }
    const FaiEnt &en = fai.at(name);
    const long p0 = sp - 1, p1 = ep - 1;
    const long o0 = en.off + (p0 / en.linebases) * en.linewidth + p0 % en.linebases;
    const long o1 = en.off + (p1 / en.linebases) * en.linewidth + p1 % en.linebases;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { *err = "cannot open " + path; return false; }
    std::string raw((size_t)(o1 - o0 + 1), '\0');
    const bool ok = fseeko(f, (off_t)o0, SEEK_SET) == 0 && fread(&raw[0], 1, raw.size(), f) == raw.size();
    fclose(f);
    if (!ok) { *err = path + ": short read (stale .fai?)"; return false; }
    out->reserve((size_t)(ep - sp + 1));
    for (char c : raw) if (c != '\n' && c != '\r') out->push_back(c);
    if ((long)out->size() != ep - sp + 1) { *err = path + ": .fai does not match the file"; return false; }
    return true;
  }
};

}  // namespace

struct lancet_host {
  std::string err;
  Sample smp[4];                    // 0 normal, 1 tumor; 2, 3: reads an earlier window left in the graph, kept across a reload (lazy mode)
  // Lazy mode: the tiling only builds the window table; every lancet_host_batch call loads the alignments ITS windows can select (through
  // the .bai: a run of windows in processing order is a few stretches of the contig).  Memory is then one batch's reads instead of the whole
  // region's -- a whole chromosome at 60x does not fit the 32-bit offsets of one load.  LANCET_HOST_LAZY=1 / 0 forces it on / off; by default
  // it is on above 400 000 windows when both BAMs have an index.
  bool lazy = false;
  BaiCache bai[2];
  Fasta fa;
  std::vector<std::string> chroms;     // contigs of the tiling, in order of first appearance (chr_id of the batches)
  std::vector<const char *> chrom_ptrs;
  std::vector<Window> windows;      // processing order
  std::vector<RSel> leak;           // reads of a window without a mapped read: the reference's processGraph returns before g.clear()
                                    // (src/Microassembler.cc:83), so they are still in the graph when the next window is loaded
  int next_w = 0;                   // where the last batch call ended: a call that starts elsewhere works `leak` out again (prime_leak)
  int loaded_lo = 0, loaded_hi = 0; // lazy mode: the windows whose alignments are loaded (lancet_host_load_range); a batch inside them does not reload
  // last batch
  std::vector<int32_t> b_chr, b_refstart;
  std::vector<uint32_t> b_refoff, b_readbegin, b_seqoff, b_namerank, b_bxrank;
  std::string b_ref;
  // bases / qualities of the batch: grow-only raw buffers (a std::string would zero-fill ~270 MB per batch on one thread before the
  // host threads overwrite every byte of it; here the pages are first touched by the threads that fill them)
  struct RawBuf {
    char *p = nullptr; size_t cap = 0;
    ~RawBuf() { free(p); }
    void need(size_t n) { if (n > cap) { free(p); cap = n + n / 8 + 4096; p = (char *)malloc(cap); } }
    char *data() { return p; }
  } b_seq, b_qual, b_pbases, b_pgood;      // ASCII bases / qualities, or (lancet_host_batch_packed) their packed form
  std::vector<uint32_t> b_rinfo, b_bw, b_gw, b_ridx;   // (b_ridx: reads stored once -- which distinct read each read of a window is; the three others then describe the distinct reads)
  std::vector<uint8_t> b_label, b_strand, b_mate, b_mapped, b_hp;
  std::vector<std::string> bx_names;
  std::vector<const char *> bx_ptrs;
  // --rg-file (Microassembler::loadRG, reference src/Microassembler.cc:29-48): the read groups to keep; {"null"} keeps everything (:716-721)
  std::set<std::string> readgroups{"null"};
};

namespace {

inline bool md_valid(char c) { return c != 0 && strchr("acgtumrwsykvhdbxnACGTUMRWSYKVHDBXN^", c) != nullptr; }

// parseMD (reference src/util.cc:428-483) with its quirks: the quality looked at is that of the base after the mismatch
// in MD coordinates (rpos is incremented first; std::string::operator[] at size() is '\0')
void parse_md(const char *md, std::unordered_map<int, int> &M, int start, const char *qual, int lqual, int min_qv) {
  const int n = (int)strlen(md);
  auto ffo = [&](int from) { for (int i = from; i < n; ++i) if (md_valid(md[i])) return i; return -1; };
  auto ffno = [&](int from) { for (int i = from; i < n; ++i) if (!md_valid(md[i])) return i; return -1; };
  int p = ffo(0), p_old = -1, pos = start, rpos = 0;
  while (p != -1) {
    const std::string num(md + p_old + 1, (size_t)(p - p_old - 1));
    const int step = atoi(num.c_str());
    pos += step; rpos += step;
    if (md[p] == '^') {
      const int p2 = ffno(p + 1);
      pos += (p2 != -1) ? p2 - (p + 1) : n - (p + 1);
      if (p2 == -1) break;
      p = ffo(p2); p_old = p2 - 1;
    } else {
      pos += 1; rpos += 1;
      const int q = rpos < lqual ? (unsigned char)qual[rpos] : 0;
      if (q >= min_qv) ++M[pos];
      p_old = p; p = ffo(p_old + 1);
    }
  }
}

bool any_ge(const std::unordered_map<int, int> &m, int thr) { for (auto &kv : m) if (kv.second >= thr) return true; return false; }

// isActiveRegion for one sample (label TMR / NML)
bool is_active_region(const Sample &S, const Window &win, bool normal, const lancet_host_opts &o, const std::set<std::string> &rgs) {
  const int mq = normal ? 0 : o.min_map_qual;
  const bool rg_all = rgs.count("null") != 0;
  std::string rg;                       // NOT cleared per alignment in the reference (:296): an alignment without RG keeps the previous one's
  std::unordered_map<int, int> mapX, mapI, mapD, mapSC;
  const std::pair<size_t, size_t> sp = (size_t)win.chr < S.span.size() ? S.span[(size_t)win.chr] : std::pair<size_t, size_t>(0, 0);
  size_t lo = (size_t)(std::lower_bound(S.starts.begin() + (long)sp.first, S.starts.begin() + (long)sp.second, win.start) - S.starts.begin());
  for (size_t i = lo; i < sp.second; ++i) {
    const Read &r = S.reads[i];
    const int alstart = r.pos0;
    if (alstart > win.end) break;
    const int alend = alstart + r.ref_len;
    if (alstart < win.start || alend > win.end) continue;
    if (!(r.mapq >= mq && !(r.flag & 0x400))) continue;
    if (r.l_seq == 0) continue;        // QueryBases / Qualities empty (:293); unstored qualities (0xFF) are NOT empty in bamtools:
                                       // such reads still count with their CIGAR evidence, their MD mismatches never pass the quality test
    if (r.rg_off != 0xFFFFFFFFu) rg = S.text.c_str() + r.rg_off;       // al.GetTag("RG", rg); if (rg.empty()) rg = "null";  (:296-297)
    if (rg.empty()) rg = "null";
    if (!(rg_all || rgs.count(rg))) continue;                            // :302 (everything below is inside that branch)
    if (r.has_md) parse_md(S.text.c_str() + r.md_off, mapX, alstart, S.qual.data() + r.seq_off, (int)r.l_seq, o.min_qual_call);
    int pos = alstart, refpos = alstart;
    for (uint32_t c = 0; c < r.n_cig; ++c) {
      const uint32_t cg = S.cigar[r.cig_off + c]; const unsigned op = cg & 15u; const int len = (int)(cg >> 4);
      if (op != 1) pos += len;
      if (op == 8) ++mapX[pos];
      if (op == 1) ++mapI[pos];
      if (op == 2) ++mapD[pos];
      // BamAlignment::GetSoftClips (bamtools 2.5.2 src/api/BamAlignment.cpp:536-606): genome position of every S op
      if (op == 2 || op == 0 || op == 8 || op == 3 || op == 7) refpos += len;
      else if (op == 4) ++mapSC[refpos];
    }
  }
  return any_ge(mapX, o.min_evidence) || any_ge(mapI, o.min_evidence) || any_ge(mapD, o.min_evidence) || any_ge(mapSC, o.min_evidence);
}

// extractReads for one sample; returns false when the window is to be skipped (coverage above --max-avg-cov)
bool extract_reads(const Sample &S, const Window &win, bool normal, const lancet_host_opts &o, const std::set<std::string> &rgs, std::vector<Sel> *out) {
  const bool rg_all = rgs.count("null") != 0;
  int mq = o.min_map_qual; double min_delta = o.max_delta_as_xs;
  if (normal) { mq = 0; min_delta = -1; }
  long totalbp = 0;
  const size_t rawlen = win.seq.size();
  const std::pair<size_t, size_t> sp = (size_t)win.chr < S.span.size() ? S.span[(size_t)win.chr] : std::pair<size_t, size_t>(0, 0);
  size_t lo = (size_t)(std::lower_bound(S.starts.begin() + (long)sp.first, S.starts.begin() + (long)sp.second, win.start) - S.starts.begin());
  for (size_t i = lo; i < sp.second; ++i) {
    const Read &r = S.reads[i];
    const int alstart = r.pos0;
    if (alstart > win.end) break;
    if (rawlen > 0 && ((double)totalbp / (double)rawlen) > (double)o.max_avg_cov) return false;      // :491-496
    const int alend = alstart + r.ref_len;
    if (alstart < win.start || alend > win.end) continue;                                           // :498-500
    if (o.primary_alignment_only && (r.flag & 0x100)) continue;
    if (!(r.mapq >= mq && !(r.flag & 0x400))) continue;                                             // :504
    uint8_t mate = (r.flag & 0x40) ? 1 : ((r.flag & 0x80) ? 2 : 0);
    if ((r.flag & 0x40) && (r.flag & 0x80)) mate = 2;                                               // :510-511
    const uint8_t strand = (r.flag & 0x10) ? LANCET_REV : LANCET_FWD;
    if (std::fabs((double)r.as - (double)r.xs) <= min_delta && r.as != -1.f && r.xs != -1.f) continue;   // :535
    if (r.xt_is_R && !normal) continue;                                                             // :554-559
    if (r.has_xa && !normal && o.xa_filter) continue;                                               // :573-579
    if (!rg_all) {                                                                                  // :611-616 (rg = ""; GetTag; empty -> "null")
      const char *rg = r.rg_off != 0xFFFFFFFFu ? S.text.c_str() + r.rg_off : "";
      if (!rgs.count(*rg ? rg : "null")) continue;
    }
    out->push_back(Sel{(uint32_t)i, mate, strand, (uint8_t)((r.flag & 0x4) ? 0 : 1)});
    totalbp += (long)r.l_seq;
  }
  return true;
}

unsigned host_threads(int items) {
  unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 1; if (nt > 64) nt = 64;
  if (const char *e = getenv("LANCET_HOST_THREADS")) { const int v = atoi(e); if (v > 0) nt = (unsigned)v; }
  if ((int)nt > items) nt = (unsigned)(items > 0 ? items : 1);
  return nt;
}

// isRepeat (reference src/util.cc:295-315): a k-mer seen twice among offsets [0, len-K).  The reference sorts nothing either -- it
// inserts every k-mer into a std::set and looks at the size; here: a rolling 64-bit hash per offset into an open-addressing table, a hit
// confirmed on the characters (sorting the ~500 offsets of a window by 101-character compares was the largest single cost of read
// selection: 60 of 70 us per window).
bool is_repeat(const std::string &s, int k) {
  const int n = (int)s.size() - k;
  if (n <= 1 || k <= 0) return false;
  unsigned bits = 4; while ((1u << bits) < 2u * (unsigned)n) ++bits;
  const unsigned cap = 1u << bits;
  static thread_local std::vector<int> tab; static thread_local std::vector<uint64_t> hs;
  tab.assign(cap, -1); hs.resize(cap);
  const uint64_t B = 0x100000001B3ULL;
  uint64_t pw = 1; for (int i = 0; i + 1 < k; ++i) pw *= B;
  uint64_t h = 0; for (int i = 0; i < k; ++i) h = h * B + (unsigned char)s[(size_t)i];
  for (int i = 0; i < n; ++i) {
    if (i) h = (h - (uint64_t)(unsigned char)s[(size_t)i - 1] * pw) * B + (unsigned char)s[(size_t)(i + k - 1)];
    uint64_t x = h ^ (h >> 31); x *= 0x9E3779B97F4A7C15ULL;
    unsigned idx = (unsigned)(x >> (64 - bits));
    while (tab[idx] >= 0) {
      if (hs[idx] == h && memcmp(s.data() + tab[idx], s.data() + i, (size_t)k) == 0) return true;
      idx = (idx + 1) & (cap - 1);
    }
    tab[idx] = i; hs[idx] = h;
  }
  return false;
}

}  // namespace

extern "C" {

void lancet_host_opts_default(lancet_host_opts *o) {
  o->padding = 250; o->window_size = 600; o->min_map_qual = 15; o->max_delta_as_xs = 5; o->primary_alignment_only = 0;
  o->xa_filter = 0; o->max_avg_cov = 10000; o->max_k = 101; o->linked = 0; o->active_region = 1; o->min_evidence = 3;
  o->min_qual_call = 17 + 33;
}

lancet_host *lancet_host_open(const char *tumor_bam, const char *normal_bam, const char *ref_fasta, char *err, size_t errlen) {
  lancet_host *h = new lancet_host();
  h->smp[0].path = normal_bam ? normal_bam : ""; h->smp[1].path = tumor_bam ? tumor_bam : "";
  bool ok = tumor_bam && normal_bam && ref_fasta && h->fa.open(ref_fasta, &h->err);
  if (ok) for (int s = 0; s < 2 && ok; ++s) { FILE *f = fopen(h->smp[s].path.c_str(), "rb"); if (!f) { h->err = "cannot open " + h->smp[s].path; ok = false; } else fclose(f); }
  if (!ok) {
    if (h->err.empty()) h->err = "missing argument";
    if (err && errlen) { strncpy(err, h->err.c_str(), errlen - 1); err[errlen - 1] = 0; }
    delete h; return nullptr;
  }
  return h;
}
void lancet_host_close(lancet_host *h) { delete h; }
const char *lancet_host_last_error(const lancet_host *h) { return h ? h->err.c_str() : "null host"; }
const char *lancet_host_sample(const lancet_host *h, int which) { return h->smp[which ? 1 : 0].sample_name.c_str(); }
const char *lancet_host_chrom(const lancet_host *h) { return h->chroms.empty() ? "" : h->chroms[0].c_str(); }
const char *const *lancet_host_chroms(const lancet_host *h, int *n) { if (n) *n = (int)h->chrom_ptrs.size(); return h->chrom_ptrs.data(); }
// --rg-file: Microassembler::loadRG (reference src/Microassembler.cc:29-48) -- whitespace-separated read-group names; an empty file
// keeps everything ("null").  Call before the windows are batched.
int lancet_host_set_rg_file(lancet_host *h, const char *path) {
  if (!h) return LANCET_E_ARG;
  h->readgroups.clear();
  if (path && *path) {
    FILE *fp = fopen(path, "r");
    if (!fp) { h->err = std::string("cannot open read-group file '") + path + "'"; h->readgroups.insert("null"); return LANCET_E_ARG; }
    char buf[4096];
    while (fscanf(fp, "%4095s", buf) == 1) h->readgroups.insert(buf);
    fclose(fp);
  }
  if (h->readgroups.empty()) h->readgroups.insert("null");
  return LANCET_OK;
}
int lancet_host_debug_is_repeat(const char *seq, int k) { return is_repeat(std::string(seq ? seq : ""), k) ? 1 : 0; }     // (tests)
int lancet_host_first_has_md(const lancet_host *h, int which) { return h->smp[which ? 1 : 0].first_has_md; }
const char *lancet_host_window_hdr(const lancet_host *h, int w) { return (w >= 0 && (size_t)w < h->windows.size()) ? h->windows[(size_t)w].hdr.c_str() : ""; }
int lancet_host_window_chrom(const lancet_host *h, int w) { return (w >= 0 && (size_t)w < h->windows.size()) ? h->windows[(size_t)w].chr : -1; }
int lancet_host_window_span(const lancet_host *h, int w, int32_t *start, int32_t *end) {
  if (w < 0 || (size_t)w >= h->windows.size()) return LANCET_E_ARG;
  if (start) *start = h->windows[(size_t)w].start;
  if (end) *end = h->windows[(size_t)w].end;
  return LANCET_OK;
}

}  // extern "C"

namespace {

// loadRefs for one region (reference src/Lancet.cc:189-316): padding, clipping, windows of window_size every 100 bp, last
// window LEN = len - offset - 1, upper case + IUPAC -> N.
// Known divergence: the padded end is clipped to the contig length of the FASTA index; the reference clips to the tumor BAM header's
// RefLength (src/Lancet.cc:233-249).  The two agree whenever the BAM was aligned to this FASTA (the only supported use).  Appends to h->windows and notes which alignment starts the
// windows can select.
int tile_one(lancet_host *h, const std::string &reg, const lancet_host_opts *o, std::map<std::string, std::vector<std::pair<int32_t, int32_t>>> *want) {
  const size_t x = reg.find(':');
  const std::string chrom = reg.substr(0, x);
  const long clen = h->fa.length(chrom);
  if (clen < 0) { h->err = "contig '" + chrom + "' not in the reference"; return LANCET_E_ARG; }
  long sp = 1, ep = clen;
  if (x != std::string::npos) {
    const size_t y = reg.find('-', x);
    if (y == std::string::npos) { h->err = "region must be chr:start-end"; return LANCET_E_ARG; }
    sp = atol(reg.substr(x + 1, y - x - 1).c_str()) - o->padding;
    ep = atol(reg.substr(y + 1).c_str()) + o->padding;
    if (sp < 1) sp = 1;
    if (ep > clen) ep = clen;
  }
  std::string s;
  if (!h->fa.fetch(chrom, sp, ep, &s, &h->err)) return LANCET_E_ARG;
  for (char &c : s) { c = (char)toupper((unsigned char)c); if (strchr("MRWSYKVHDBX", c)) c = 'N'; }
  int chr = -1;
  for (size_t i = 0; i < h->chroms.size(); ++i) if (h->chroms[i] == chrom) chr = (int)i;
  if (chr < 0) { chr = (int)h->chroms.size(); h->chroms.push_back(chrom); }
  const long delta = 100, wsz = o->window_size;
  long end = (long)s.size(), offset = 0;
  while (offset < end) {
    long ln = wsz;
    if (offset + wsz >= (long)s.size()) { ln = (long)s.size() - offset - 1; end = offset; }
    Window w;
    w.seq = ln > 0 ? s.substr((size_t)offset, (size_t)ln) : std::string();
    w.start = (int32_t)(sp + offset); w.end = (int32_t)(sp + offset + ln); w.chr = chr;
    w.hdr = chrom + ":" + std::to_string(w.start) + "-" + std::to_string(w.end);
    h->windows.push_back(std::move(w));
    offset += delta;
  }
  // the alignments a window can select start inside [sp, ep] (1-based window coordinates compared with 0-based
  // alignment starts, as the reference does)
  if (ep >= sp) (*want)[chrom].emplace_back((int32_t)sp - 1, (int32_t)ep + 1);
  return LANCET_OK;
}

// processing order = iteration order of the reference's std::map<string, Ref_t*> keyed by the window header
// (src/Microassembler.cc:779); a header tiled twice (overlapping BED lines) is kept once, like map::insert.  Then the
// alignments of both samples are decoded for the union of the tiled stretches.
int finish_tiling(lancet_host *h, std::map<std::string, std::vector<std::pair<int32_t, int32_t>>> &want) {
  std::stable_sort(h->windows.begin(), h->windows.end(), [](const Window &a, const Window &b) { return a.hdr < b.hdr; });
  h->windows.erase(std::unique(h->windows.begin(), h->windows.end(), [](const Window &a, const Window &b) { return a.hdr == b.hdr; }), h->windows.end());
  h->chrom_ptrs.clear();
  for (auto &c : h->chroms) h->chrom_ptrs.push_back(c.c_str());
  std::vector<Want> wants;
  for (auto &kv : want) {
    Want w; w.chrom = kv.first; w.chr = 0;
    for (size_t i = 0; i < h->chroms.size(); ++i) if (h->chroms[i] == kv.first) w.chr = (int)i;
    std::sort(kv.second.begin(), kv.second.end());
    for (auto &iv : kv.second) {
      if (!w.iv.empty() && iv.first <= w.iv.back().second + 1) { if (iv.second > w.iv.back().second) w.iv.back().second = iv.second; }
      else w.iv.push_back(iv);
    }
    wants.push_back(std::move(w));
  }
  {
    const char *e = getenv("LANCET_HOST_LAZY");
    auto has_bai = [](const std::string &bam) {
      std::string c[2] = {bam + ".bai", bam.size() > 4 && bam.compare(bam.size() - 4, 4, ".bam") == 0 ? bam.substr(0, bam.size() - 4) + ".bai" : std::string()};
      for (const std::string &x : c) if (!x.empty()) { FILE *f = fopen(x.c_str(), "rb"); if (f) { fclose(f); return true; } }
      return false;
    };
    bool indexed = has_bai(h->smp[0].path) && has_bai(h->smp[1].path);
    const bool want_lazy = e ? atoi(e) != 0 : h->windows.size() > 400000;      // (lancet_host.h: above 400 000 windows a tiling loads batch by batch)
    h->smp[2] = Sample(); h->smp[3] = Sample();
    h->bai[0] = BaiCache(); h->bai[1] = BaiCache();
    if (want_lazy && indexed) {
      // lazy only with both indexes PARSED (a file that merely exists would make every batch stream its BAM from the start)
      for (int smp = 0; smp < 2; ++smp) { h->bai[smp].have = load_bai(h->smp[smp].path, &h->bai[smp].index); h->bai[smp].tried = true; }
      indexed = h->bai[0].have && h->bai[1].have;
    }
    h->lazy = want_lazy && indexed;
    if (want_lazy && !h->lazy)
      fprintf(stderr, "[lancet_host] batch-by-batch loading %s but not possible: both BAMs need a readable .bai (%s); the whole tiling's alignments are loaded at once\n",
              e ? "asked for (LANCET_HOST_LAZY)" : "wanted for this many windows", has_bai(h->smp[0].path) && has_bai(h->smp[1].path) ? "an index could not be parsed" : "an index is missing");
    if (h->lazy) wants.clear();                      // header, sample name, MD on the first alignment: no alignments yet
  }
  std::string errs[2]; bool ok[2] = {false, false};
  auto load = [&](int smp) {
    Sample &S = h->smp[smp];
    const std::string path = S.path;
    S = Sample(); S.path = path;
    ok[smp] = load_bam(path, wants, &S, &errs[smp]);
  };
  std::thread other(load, 0);                         // the two samples side by side
  load(1);
  other.join();
  for (int smp = 0; smp < 2; ++smp) if (!ok[smp]) { h->err = errs[smp]; return LANCET_E_ARG; }
  return (int)h->windows.size();
}

}  // namespace

namespace {

// Lazy mode: the alignments the windows [w_begin, w_end) can select, loaded in place of the previous batch's.  Reads an earlier window left
// in the graph (lancet_host::leak) are copied into the side store first: their indices would not survive the reload.
int reload_for(lancet_host *h, int w_begin, int w_end) {
  if (!h->leak.empty()) {
    Sample keep[2];
    for (RSel &r : h->leak) {
      const Sample &S = h->smp[r.smp];
      const Read &src = S.reads[r.s.idx];
      Sample &D = keep[r.smp & 1];
      Read d = src;
      d.seq_off = (uint32_t)D.seq.size(); D.seq.append(S.seq.data() + src.seq_off, src.l_seq); D.qual.append(S.qual.data() + src.seq_off, src.l_seq);
      d.cig_off = (uint32_t)D.cigar.size(); for (uint32_t c = 0; c < src.n_cig; ++c) D.cigar.push_back(S.cigar[src.cig_off + c]);
      auto put = [&](uint32_t off) { const char *z = S.text.c_str() + off; const uint32_t o2 = (uint32_t)D.text.size(); D.text.append(z, strlen(z)); D.text.push_back('\0'); return o2; };
      d.name_off = put(src.name_off); d.bx_off = put(src.bx_off);
      if (src.has_md) d.md_off = put(src.md_off);
      if (src.rg_off != 0xFFFFFFFFu) d.rg_off = put(src.rg_off);
      r.s.idx = (uint32_t)D.reads.size(); r.smp = (uint8_t)(2 + (r.smp & 1));
      D.reads.push_back(d); D.starts.push_back(d.pos0);
    }
    h->smp[2] = std::move(keep[0]); h->smp[3] = std::move(keep[1]);
  } else { h->smp[2] = Sample(); h->smp[3] = Sample(); }
  // the stretches of each contig these windows cover: a window selects alignments that start inside [start, end] (tile_one)
  std::vector<Want> wants(h->chroms.size());
  for (size_t c = 0; c < h->chroms.size(); ++c) { wants[c].chrom = h->chroms[c]; wants[c].chr = (int)c; }
  for (int w = w_begin; w < w_end; ++w) {
    const Window &win = h->windows[(size_t)w];
    if (win.end >= win.start) wants[(size_t)win.chr].iv.emplace_back(win.start - 1, win.end + 1);
  }
  for (Want &w : wants) {
    std::sort(w.iv.begin(), w.iv.end());
    std::vector<std::pair<int32_t, int32_t>> m;
    for (auto &iv : w.iv) { if (!m.empty() && iv.first <= m.back().second + 1) { if (iv.second > m.back().second) m.back().second = iv.second; } else m.push_back(iv); }
    w.iv.swap(m);
  }
  wants.erase(std::remove_if(wants.begin(), wants.end(), [](const Want &w) { return w.iv.empty(); }), wants.end());
  std::string errs[2]; bool ok[2] = {false, false};
  auto load = [&](int smp) {
    Sample &S = h->smp[smp];
    const std::string path = S.path;
    Sample fresh; fresh.path = path;
    ok[smp] = load_bam(path, wants, &fresh, &errs[smp], &h->bai[smp]);
    S = std::move(fresh);
  };
  std::thread other(load, 0);
  load(1);
  other.join();
  for (int smp = 0; smp < 2; ++smp) if (!ok[smp]) { h->err = errs[smp]; return LANCET_E_ARG; }
  return LANCET_OK;
}

}  // namespace

extern "C" {

int lancet_host_tile(lancet_host *h, const char *region, const lancet_host_opts *o) {
  const char *regs[1] = {region ? region : ""};
  return lancet_host_tile_regions(h, nullptr, regs, 1, o);
}

// main() of the reference: loadBed first, then loadRefs on --reg, into one table (src/Lancet.cc:852-857).  loadBed hands
// "chr:(start-PADDING)-(end+PADDING)" to loadRefs, which pads again (src/Lancet.cc:319-351, :233-249): a BED interval is
// padded twice.  Lines starting with '#' are skipped; columns are tab separated.
int lancet_host_tile_regions(lancet_host *h, const char *bed_path, const char *const *regions, int n_regions, const lancet_host_opts *o) {
  if (!h || !o) return LANCET_E_ARG;
  h->windows.clear(); h->leak.clear(); h->chroms.clear(); h->next_w = 0; h->loaded_lo = h->loaded_hi = 0;
  std::map<std::string, std::vector<std::pair<int32_t, int32_t>>> want;
  if (bed_path && *bed_path) {
    std::string bed;
    if (!read_file(bed_path, &bed, &h->err)) return LANCET_E_ARG;
    size_t q = 0; int line_no = 0;
    while (q < bed.size()) {
      size_t e = bed.find('\n', q); if (e == std::string::npos) e = bed.size();
      std::string line = bed.substr(q, e - q);
      q = e + 1; ++line_no;
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (!line.empty() && line[0] == '#') continue;
      std::vector<std::string> t; size_t a = 0;
      while (a <= line.size()) { size_t b = line.find('\t', a); if (b == std::string::npos) b = line.size(); t.push_back(line.substr(a, b - a)); a = b + 1; }
      // std::stoi as the reference uses it (src/Lancet.cc:343-344): leading blanks, a sign, then digits; anything after them is ignored
      auto stoi_like = [](const std::string &x, long *v) { char *end = nullptr; errno = 0; *v = strtol(x.c_str(), &end, 10); return end != x.c_str() && errno == 0 && *v >= INT32_MIN && *v <= INT32_MAX; };
      long b0 = 0, b1 = 0;
      if (t.size() < 3 || !stoi_like(t[1], &b0) || !stoi_like(t[2], &b1)) {
        h->err = std::string(bed_path) + ": line " + std::to_string(line_no) + " is not chrom<TAB>start<TAB>end"; return LANCET_E_ARG; }   // (the reference's stoi throws here)
      long sp = b0 - o->padding, ep = b1 + o->padding;
      if (sp < 1) sp = 1;
      const int rc = tile_one(h, t[0] + ":" + std::to_string(sp) + "-" + std::to_string(ep), o, &want);
      if (rc != LANCET_OK) return rc;
    }
  }
  for (int i = 0; i < n_regions; ++i) if (regions && regions[i] && *regions[i]) { const int rc = tile_one(h, regions[i], o, &want); if (rc != LANCET_OK) return rc; }
  if (h->chroms.empty() && h->windows.empty() && !(bed_path && *bed_path)) { h->err = "no region given"; return LANCET_E_ARG; }
  return finish_tiling(h, want);
}

// Trims and packs every alignment of a sample once (host_pack.h), on the host threads; a read that is not usable keeps length 0.
// The packed form of a load's alignments: laid out here (word offsets, room for every alignment), filled by pack_read_once when a window
// first SELECTS an alignment -- duplicates, low-MAPQ reads and reads no window takes are never trimmed or packed, and their pages of the
// two arrays are never touched (round 4; it used to pack the whole load on the first packed batch).
static void ensure_pack_cache(Sample &S, const lancet_params &P) {
  const size_t n = S.reads.size();
  if (S.pk_qtrim == P.min_qual_trim && S.pk_qcall == P.min_qual_call && S.pk_n == n && S.pk_tlen.size() == n && S.pk_state) return;
  S.pk_tlen.assign(n, 0); S.pk_bw.assign(n + 1, 0); S.pk_gw.assign(n + 1, 0);
  uint64_t bw = 0, gw = 0;
  for (size_t i = 0; i < n; ++i) { S.pk_bw[i] = (uint32_t)bw; S.pk_gw[i] = (uint32_t)gw; bw += (S.reads[i].l_seq + 15) / 16; gw += (S.reads[i].l_seq + 31) / 32; }
  S.pk_bw[n] = (uint32_t)bw; S.pk_gw[n] = (uint32_t)gw;          // (< 2^32: at most one word per 16 of the < 4 G bases)
  S.pk_bases.resize(4 * (size_t)bw + 4); S.pk_good.resize(4 * (size_t)gw + 4);
  S.pk_state.reset(new std::atomic<uint8_t>[n + 1]);
  for (size_t i = 0; i <= n; ++i) S.pk_state[i].store(0, std::memory_order_relaxed);
  S.pk_qtrim = P.min_qual_trim; S.pk_qcall = P.min_qual_call; S.pk_n = n;
}
// Trim + pack alignment i of the load, once: the first thread to ask does it, a thread that asks meanwhile waits for it (the batch is
// assembled by several threads, and an alignment is in about six windows).
static inline void pack_read_once(Sample &S, const lancet_params &P, size_t i) {
  std::atomic<uint8_t> &st = S.pk_state[i];
  if (st.load(std::memory_order_acquire) == 2) return;
  uint8_t expect = 0;
  if (st.compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) {
    const Read &r = S.reads[i];
    uint32_t ri = 0, *pb = (uint32_t *)S.pk_bases.data() + S.pk_bw[i], *pg = (uint32_t *)S.pk_good.data() + S.pk_gw[i];
    lc_prep_read_host(P, S.seq.data() + r.seq_off, S.qual.data() + r.seq_off, (int)r.l_seq, 0, 0, 0, 0, &ri, pb, pg);
    const uint32_t tl = ri & 0xFFFFu;
    for (uint32_t wv = (tl + 15) / 16; wv < (r.l_seq + 15) / 16; ++wv) pb[wv] = 0;
    for (uint32_t wv = (tl + 31) / 32; wv < (r.l_seq + 31) / 32; ++wv) pg[wv] = 0;
    S.pk_tlen[i] = tl;
    st.store(2, std::memory_order_release);
  } else while (st.load(std::memory_order_acquire) != 2) std::this_thread::yield();
}

// What the windows before w_begin left in the graph, for a batch call that does not continue the previous one (an N-process run deals the
// batches out: lancet_main.cc --ranks).  `leak` is sequential state of the reference's window loop: a window that passes the filters
// but has no mapped read returns from processGraph before g.clear() (src/Microassembler.cc:83), so its reads are still there when the next
// window is loaded; a window with a mapped read, or one skipped for coverage (g.clear(true), :836), ends the run; a window the filters
// turn away (:799-820) does not touch the graph.  So the leak entering w_begin is the reads of the trailing run of such windows: walk
// backwards until one ends it -- nearly always the very first step.
static int prime_leak(lancet_host *h, int w_begin, const lancet_host_opts *o) {
  h->leak.clear();
  for (int w = w_begin - 1; w >= 0; --w) {
    const Window &win = h->windows[(size_t)w];
    if (win.seq.empty() || is_repeat(win.seq, o->max_k)) continue;
    if (h->lazy) {                                    // this window's alignments (what was collected so far moves to the side store)
      const int rc = reload_for(h, w, w + 1); if (rc != LANCET_OK) return rc;
      h->loaded_lo = w; h->loaded_hi = w + 1;
    }
    if (o->active_region && !(is_active_region(h->smp[1], win, false, *o, h->readgroups) || is_active_region(h->smp[0], win, true, *o, h->readgroups))) continue;
    std::vector<Sel> sT, sN;
    const bool okT = extract_reads(h->smp[1], win, false, *o, h->readgroups, &sT);
    const bool okN = extract_reads(h->smp[0], win, true, *o, h->readgroups, &sN);
    if (!(okT && okN)) break;                         // skipped for coverage: the graph was cleared
    uint8_t mp = 0;
    for (const Sel &x : sT) mp |= x.mapped;
    for (const Sel &x : sN) mp |= x.mapped;
    if (mp) break;                                    // processed: cleared
    std::vector<RSel> mine;                           // this window's reads come BEFORE what the later windows of the run added
    for (const Sel &x : sT) mine.push_back(RSel{1, x});
    for (const Sel &x : sN) mine.push_back(RSel{0, x});
    mine.insert(mine.end(), h->leak.begin(), h->leak.end());
    h->leak.swap(mine);
  }
  return LANCET_OK;
}

static int host_batch_impl(lancet_host *h, int w_begin, int w_end, const lancet_host_opts *o, lancet_window_batch *out,
                           int32_t *kept, int32_t *n_kept, const lancet_params *P, lancet_packed_reads *pk);
// Lazy mode: loads, once, the alignments the windows [w_begin, w_end) can select; batch calls inside that range then load nothing.  (A rank
// of an N-process run owns one contiguous range of the window table; in the table's order -- header strings -- a batch of windows is
// scattered over its contig, so loading batch by batch touches the same stretches again and again.)  Without lazy mode: nothing to do.
int lancet_host_load_range(lancet_host *h, int w_begin, int w_end, const lancet_host_opts *o) {
  if (!h || !o || w_begin < 0 || w_end < w_begin || (size_t)w_end > h->windows.size()) { if (h) h->err = "bad window range"; return LANCET_E_ARG; }
  if (!h->lazy || w_end == w_begin) return LANCET_OK;
  // what the windows before the range left in the graph first (it loads single windows), then the range: reload_for moves those reads to the
  // side store, and the batch call that starts at w_begin continues from here
  if (w_begin != h->next_w) { const int rc = prime_leak(h, w_begin, o); if (rc != LANCET_OK) return rc; }
  const int rc = reload_for(h, w_begin, w_end);
  if (rc != LANCET_OK) return rc;
  h->loaded_lo = w_begin; h->loaded_hi = w_end; h->next_w = w_begin;
  return LANCET_OK;
}
int lancet_host_batch(lancet_host *h, int w_begin, int w_end, const lancet_host_opts *o, lancet_window_batch *out,
                      int32_t *kept, int32_t *n_kept) { return host_batch_impl(h, w_begin, w_end, o, out, kept, n_kept, nullptr, nullptr); }
// The same batch with the reads trimmed and packed for lancet_engine_upload_packed (host_pack.h: the engine's own routine) instead of
// copied as ASCII: out->seq / qual are NULL, `pk` points at the packed arrays (owned by the host object like the batch's).
int lancet_host_batch_packed(lancet_host *h, int w_begin, int w_end, const lancet_host_opts *o, const lancet_params *P, lancet_window_batch *out,
                             lancet_packed_reads *pk, int32_t *kept, int32_t *n_kept) {
  if (!P || !pk) { if (h) h->err = "lancet_host_batch_packed: parameters and packed-reads struct are required"; return LANCET_E_ARG; }
  return host_batch_impl(h, w_begin, w_end, o, out, kept, n_kept, P, pk);
}
static int host_batch_impl(lancet_host *h, int w_begin, int w_end, const lancet_host_opts *o, lancet_window_batch *out,
                           int32_t *kept, int32_t *n_kept, const lancet_params *P, lancet_packed_reads *pk) {
  if (!h || !o || !out || w_begin < 0 || w_end < w_begin || (size_t)w_end > h->windows.size()) { if (h) h->err = "bad window range"; return LANCET_E_ARG; }
  const int nwin = w_end - w_begin;
  const bool timing = getenv("LANCET_HOST_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  if (w_begin != h->next_w) { const int rc = prime_leak(h, w_begin, o); if (rc != LANCET_OK) return rc; }      // (not the continuation of the call before)
  if (h->lazy && !(w_begin >= h->loaded_lo && w_end <= h->loaded_hi)) {
    const int rc = reload_for(h, w_begin, w_end); if (rc != LANCET_OK) return rc;
    h->loaded_lo = w_begin; h->loaded_hi = w_end;
  }
  h->next_w = w_end;
  const double t0 = now();
  std::vector<std::vector<Sel>> selT((size_t)nwin), selN((size_t)nwin);
  std::vector<uint8_t> keep((size_t)nwin, 0), wmapped((size_t)nwin, 0);
  std::vector<uint64_t> wbases((size_t)nwin, 0), wbw((size_t)nwin, 0), wgw((size_t)nwin, 0);      // bases; 16-base / 32-base words of the packed form
  {   // per-window filters and read selection: independent windows, one chunk of windows per host thread at a time
    std::atomic<int> next(0);
    auto work = [&]() {
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= nwin) break;
        const Window &win = h->windows[(size_t)(w_begin + i)];
        if (win.seq.empty()) continue;          // isNseq (:799, src/util.cc:259-273): its test `!= 'N' || != 'n'` holds for every character, so only an EMPTY window is "all N"
        if (is_repeat(win.seq, o->max_k)) continue;                                                   // :800
        if (o->active_region && !(is_active_region(h->smp[1], win, false, *o, h->readgroups) || is_active_region(h->smp[0], win, true, *o, h->readgroups))) continue;   // :817-820
        const bool okT = extract_reads(h->smp[1], win, false, *o, h->readgroups, &selT[(size_t)i]);
        const bool okN = extract_reads(h->smp[0], win, true, *o, h->readgroups, &selN[(size_t)i]);      // (both samples are read before the skip test, :833-836)
        keep[(size_t)i] = (okT && okN) ? 1 : 2;                                        // 2: skipped for coverage -> g.clear(true)
        uint64_t nb = 0, bw = 0, gw = 0; uint8_t mp = 0;
        for (const Sel &s : selT[(size_t)i]) { const uint32_t l = h->smp[1].reads[s.idx].l_seq; nb += l; bw += (l + 15) / 16; gw += (l + 31) / 32; mp |= s.mapped; }
        for (const Sel &s : selN[(size_t)i]) { const uint32_t l = h->smp[0].reads[s.idx].l_seq; nb += l; bw += (l + 15) / 16; gw += (l + 31) / 32; mp |= s.mapped; }
        wbases[(size_t)i] = nb; wbw[(size_t)i] = bw; wgw[(size_t)i] = gw; wmapped[(size_t)i] = mp;
      }
    };
    unsigned nt = host_threads(nwin);
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
  }
  const double t1 = now();
  // ---- SoA assembly, windows in processing order; per window tumor reads then normal reads (:833-834).
  //      Sizes first (prefix sums), then every window fills its own slices on the host threads.
  std::vector<int> kw;                                 // kept windows (index into [w_begin, w_end))
  std::vector<std::vector<RSel>> pre;                  // per kept window: what an earlier window left in the graph (nearly always empty);
                                                       // its own reads follow: selT then selN -- no per-window copies
  for (int i = 0; i < nwin; ++i) {
    if (keep[(size_t)i] == 2) { h->leak.clear(); continue; }
    if (keep[(size_t)i] != 1) continue;
    kw.push_back(i); pre.push_back(h->leak);
    if (wmapped[(size_t)i]) h->leak.clear();           // (left-over reads are unmapped by construction: the window's own reads decide)
    else {                                             // countMappedReads() <= 0: processGraph returns, nothing is cleared
      for (const Sel &s2 : selT[(size_t)i]) h->leak.push_back(RSel{1, s2});
      for (const Sel &s2 : selN[(size_t)i]) h->leak.push_back(RSel{0, s2});
    }
  }
  const int nk = (int)kw.size();
  h->b_chr.assign((size_t)nk, 0); h->b_refstart.resize((size_t)nk);
  h->b_refoff.assign((size_t)nk + 1, 0); h->b_readbegin.assign((size_t)nk + 1, 0);
  std::vector<uint64_t> base0((size_t)nk + 1, 0), bw0((size_t)nk + 1, 0), gw0((size_t)nk + 1, 0);
  for (int k = 0; k < nk; ++k) {
    const int i = kw[(size_t)k];
    const Window &win = h->windows[(size_t)(w_begin + i)];
    if (kept) kept[k] = w_begin + i;
    uint64_t nb = wbases[(size_t)i], bw = wbw[(size_t)i], gw = wgw[(size_t)i];
    for (const RSel &r : pre[(size_t)k]) { const uint32_t l = h->smp[r.smp].reads[r.s.idx].l_seq; nb += l; bw += (l + 15) / 16; gw += (l + 31) / 32; }   // (smp 2 / 3: the store of left-over reads)
    base0[(size_t)k + 1] = base0[(size_t)k] + nb; bw0[(size_t)k + 1] = bw0[(size_t)k] + bw; gw0[(size_t)k + 1] = gw0[(size_t)k] + gw;
    h->b_readbegin[(size_t)k + 1] = h->b_readbegin[(size_t)k] + (uint32_t)(pre[(size_t)k].size() + selT[(size_t)i].size() + selN[(size_t)i].size());
    h->b_refoff[(size_t)k + 1] = h->b_refoff[(size_t)k] + (uint32_t)win.seq.size();
  }
  if (base0[(size_t)nk] > 0xFFFFFFFFull) { h->err = "batch holds more than 4 Gi bases: use fewer windows per batch"; return LANCET_E_ARG; }
  const size_t R = h->b_readbegin[(size_t)nk], NB = (size_t)base0[(size_t)nk];
  h->b_ref.resize(h->b_refoff[(size_t)nk]);
  const bool packed = P != nullptr;
  // Reads stored once (LANCET_HOST_SHARED=0: one copy per window, as until round 5): windows every 100 bases of 600 put an alignment into ~6
  // of them.  The batch's DISTINCT alignments are numbered (per sample in file order: tumor, normal, then the two stores of left-over reads),
  // trimmed + packed on first use as before, and their words go into the batch once; a read of a window is an index into them.
  const bool shared = packed && !(getenv("LANCET_HOST_SHARED") && atoi(getenv("LANCET_HOST_SHARED")) == 0);
  uint32_t ubase[4] = {0, 0, 0, 0}; size_t U = 0; uint64_t ubw_all = 0, ugw_all = 0;
  std::vector<std::pair<uint8_t, uint32_t>> ulist;                  // distinct read u -> (sample, alignment)
  if (packed) {
    for (int smp = 0; smp < 4; ++smp) ensure_pack_cache(h->smp[smp], *P);
    if (shared) {
      // (only the index range the batch's windows touch is visited: a scan over every loaded alignment per batch was O(total reads) per
      //  batch, single-threaded -- millions of alignments per sample in a scan that is loaded at once)
      for (int smp = 0; smp < 4; ++smp) {
        Sample &S = h->smp[smp];
        if (S.uidx.size() != S.reads.size()) { S.uidx.assign(S.reads.size(), 0xFFFFFFFFu); S.u_lo = 1; S.u_hi = 0; }
        for (size_t i = S.u_lo; i <= S.u_hi && i < S.uidx.size(); ++i) S.uidx[i] = 0xFFFFFFFFu;      // what the batch before wrote
        S.u_lo = 1; S.u_hi = 0;
      }
      {   // which alignments the batch's windows hold (the same word from several threads: relaxed stores)
        std::atomic<int> next(0);
        std::mutex mm;
        auto mark = [&]() {
          size_t lo[4] = {SIZE_MAX, SIZE_MAX, SIZE_MAX, SIZE_MAX}, hi[4] = {0, 0, 0, 0}; bool any[4] = {false, false, false, false};
          auto touch = [&](int smp, size_t idx) { __atomic_store_n(&h->smp[smp].uidx[idx], 0u, __ATOMIC_RELAXED); if (idx < lo[smp]) lo[smp] = idx; if (idx > hi[smp]) hi[smp] = idx; any[smp] = true; };
          for (;;) {
            const int k = next.fetch_add(1);
            if (k >= nk) break;
            const int i = kw[(size_t)k];
            for (const RSel &rs : pre[(size_t)k]) touch(rs.smp, rs.s.idx);
            for (const Sel &s2 : selT[(size_t)i]) touch(1, s2.idx);
            for (const Sel &s2 : selN[(size_t)i]) touch(0, s2.idx);
          }
          std::lock_guard<std::mutex> g(mm);
          for (int smp = 0; smp < 4; ++smp) if (any[smp]) {
            Sample &S = h->smp[smp];
            if (S.u_lo > S.u_hi) { S.u_lo = lo[smp]; S.u_hi = hi[smp]; } else { if (lo[smp] < S.u_lo) S.u_lo = lo[smp]; if (hi[smp] > S.u_hi) S.u_hi = hi[smp]; }
          }
        };
        unsigned nt = host_threads(nk);
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(mark);
        mark();
        for (auto &t : th) t.join();
      }
      const int order[4] = {1, 0, 2, 3};
      for (int q = 0; q < 4; ++q) {
        Sample &S = h->smp[order[q]];
        ubase[order[q]] = (uint32_t)U;
        uint32_t c = 0;
        for (size_t i = S.u_lo; i <= S.u_hi && i < S.uidx.size(); ++i) if (S.uidx[i] != 0xFFFFFFFFu) { S.uidx[i] = c++; ulist.emplace_back((uint8_t)order[q], (uint32_t)i); }
        U += c;
      }
      h->b_rinfo.resize(U + 1); h->b_bw.resize(U + 1); h->b_gw.resize(U + 1); h->b_ridx.resize(R);
      for (size_t u = 0; u < U; ++u) {
        const uint32_t l = h->smp[ulist[u].first].reads[ulist[u].second].l_seq;
        h->b_bw[u] = (uint32_t)ubw_all; h->b_gw[u] = (uint32_t)ugw_all;
        ubw_all += (l + 15) / 16; ugw_all += (l + 31) / 32;
      }
      if (ubw_all + 4 > 0xFFFFFFFFull) { h->err = "batch too large"; return LANCET_E_ARG; }
      h->b_bw[U] = (uint32_t)ubw_all; h->b_gw[U] = (uint32_t)ugw_all; h->b_rinfo[U] = 0;
      h->b_pbases.need(4 * ((size_t)ubw_all + 4)); h->b_pgood.need(4 * ((size_t)ugw_all + 1));
      if (!h->b_pbases.p || !h->b_pgood.p) { h->err = "out of memory for the batch"; return LANCET_E_ARG; }
      {   // the distinct reads: trimmed + packed when first met (pack_read_once), their words into the batch, once
        std::atomic<size_t> next(0);
        auto packu = [&]() {
          for (;;) {
            const size_t u0 = next.fetch_add(256);
            if (u0 >= U) break;
            const size_t u1 = u0 + 256 < U ? u0 + 256 : U;
            for (size_t u = u0; u < u1; ++u) {
              const int smp = ulist[u].first; const uint32_t idx = ulist[u].second;
              Sample &S = h->smp[smp];
              const Read &rd = S.reads[idx];
              pack_read_once(S, *P, idx);
              memcpy((uint32_t *)h->b_pbases.p + h->b_bw[u], (const uint32_t *)S.pk_bases.data() + S.pk_bw[idx], 4 * (size_t)((rd.l_seq + 15) / 16));
              memcpy((uint32_t *)h->b_pgood.p + h->b_gw[u], (const uint32_t *)S.pk_good.data() + S.pk_gw[idx], 4 * (size_t)((rd.l_seq + 31) / 32));
              const uint8_t mate = (rd.flag & 0x40) ? 1 : ((rd.flag & 0x80) ? 2 : 0);
              h->b_rinfo[u] = lc_rinfo_word(S.pk_tlen[idx], (smp & 1) ? LANCET_TMR : LANCET_NML, (rd.flag & 0x10) ? LANCET_REV : LANCET_FWD,
                                            ((rd.flag & 0x40) && (rd.flag & 0x80)) ? 2 : mate, (uint8_t)((rd.flag & 0x4) ? 0 : 1));      // (as extract_reads derives them: the alignment's own, whatever the window)
            }
          }
        };
        unsigned nt = host_threads((int)std::min<size_t>(U / 256 + 1, 1u << 20));
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(packu);
        packu();
        for (auto &t : th) t.join();
      }
    } else {
    if (bw0[(size_t)nk] + 4 > 0xFFFFFFFFull) { h->err = "batch too large"; return LANCET_E_ARG; }
    h->b_pbases.need(4 * ((size_t)bw0[(size_t)nk] + 4)); h->b_pgood.need(4 * ((size_t)gw0[(size_t)nk] + 1));
    if (!h->b_pbases.p || !h->b_pgood.p) { h->err = "out of memory for the batch"; return LANCET_E_ARG; }
    h->b_rinfo.resize(R + 1); h->b_bw.resize(R + 1); h->b_gw.resize(R + 1);
    }
  } else {
  h->b_seq.need(NB + 1); h->b_qual.need(NB + 1);
  if (!h->b_seq.p || !h->b_qual.p) { h->err = "out of memory for the batch"; return LANCET_E_ARG; }
  }
  h->b_seqoff.resize(R + 1); h->b_seqoff[0] = 0;
  h->b_label.resize(R); h->b_strand.resize(R); h->b_mate.resize(R); h->b_mapped.resize(R); h->b_namerank.resize(R);
  h->b_hp.resize(o->linked ? R : 0);
  std::vector<const char *> bx_of(o->linked ? R : 0);   // per read (linked)
  const double t2 = now();
  {
    std::atomic<int> next(0);
    auto fill = [&]() {
      std::vector<const char *> names; std::vector<uint32_t> ord;
      for (;;) {
        const int k = next.fetch_add(1);
        if (k >= nk) break;
        const int i = kw[(size_t)k];
        const Window &win = h->windows[(size_t)(w_begin + i)];
        h->b_refstart[(size_t)k] = win.start; h->b_chr[(size_t)k] = win.chr;
        memcpy(&h->b_ref[h->b_refoff[(size_t)k]], win.seq.data(), win.seq.size());
        size_t r = h->b_readbegin[(size_t)k]; const size_t r0 = r; size_t bo = (size_t)base0[(size_t)k];
        size_t pbo = (size_t)bw0[(size_t)k], pgo = (size_t)gw0[(size_t)k];
        names.clear();
        {
          auto one = [&](int smp, const Sel &s) {
            Sample &S = h->smp[smp];
            const Read &rd = S.reads[s.idx];
            h->b_label[r] = (smp & 1) ? LANCET_TMR : LANCET_NML; h->b_strand[r] = s.strand; h->b_mate[r] = s.mate; h->b_mapped[r] = s.mapped;
            if (shared) h->b_ridx[r] = ubase[smp] + S.uidx[s.idx];
            else if (packed) {
              uint32_t *pb = (uint32_t *)h->b_pbases.p + pbo, *pg = (uint32_t *)h->b_pgood.p + pgo;
              h->b_bw[r] = (uint32_t)pbo; h->b_gw[r] = (uint32_t)pgo;
              const uint32_t nbw = (rd.l_seq + 15) / 16, ngw = (rd.l_seq + 31) / 32;
              pack_read_once(S, *P, s.idx);                                             // (packed once per alignment, when a window first takes it)
              memcpy(pb, (const uint32_t *)S.pk_bases.data() + S.pk_bw[s.idx], 4 * (size_t)nbw);
              memcpy(pg, (const uint32_t *)S.pk_good.data() + S.pk_gw[s.idx], 4 * (size_t)ngw);
              h->b_rinfo[r] = lc_rinfo_word(S.pk_tlen[s.idx], h->b_label[r], s.strand, s.mate, s.mapped);
              pbo += nbw; pgo += ngw;
            } else { memcpy(h->b_seq.p + bo, S.seq.data() + rd.seq_off, rd.l_seq); memcpy(h->b_qual.p + bo, S.qual.data() + rd.seq_off, rd.l_seq); }
            bo += rd.l_seq;
            h->b_seqoff[r + 1] = (uint32_t)bo;
            names.push_back(S.text.c_str() + rd.name_off);
            if (o->linked) { bx_of[r] = S.text.c_str() + rd.bx_off; h->b_hp[r] = (uint8_t)(rd.hp > 255 ? 255 : rd.hp); }
            ++r;
          };
          for (const RSel &rs : pre[(size_t)k]) one(rs.smp, rs.s);
          for (const Sel &s2 : selT[(size_t)i]) one(1, s2);
          for (const Sel &s2 : selN[(size_t)i]) one(0, s2);
        }
        // dense rank of the read name among the window's names under std::string operator<
        ord.resize(names.size());
        for (size_t j = 0; j < ord.size(); ++j) ord[j] = (uint32_t)j;
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return strcmp(names[a], names[b]) < 0; });
        uint32_t rank = 0;
        for (size_t j = 0; j < ord.size(); ++j) {
          if (j > 0 && strcmp(names[ord[j]], names[ord[j - 1]]) != 0) ++rank;
          h->b_namerank[r0 + ord[j]] = rank;
        }
      }
    };
    unsigned nt = host_threads(nk);
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(fill);
    fill();
    for (auto &t : th) t.join();
  }
  const double t3 = now();
  h->bx_names.clear(); h->bx_ptrs.clear(); h->b_bxrank.clear();
  if (o->linked) {
    std::unordered_set<std::string> uniq;
    for (const char *b : bx_of) if (strcmp(b, "null") != 0) uniq.emplace(b);
    std::vector<std::string> all(uniq.begin(), uniq.end());
    std::sort(all.begin(), all.end());
    h->bx_names.swap(all);
    std::unordered_map<std::string, uint32_t> rk;
    for (size_t j = 0; j < h->bx_names.size(); ++j) rk.emplace(h->bx_names[j], (uint32_t)j);
    h->b_bxrank.reserve(bx_of.size());
    for (const char *b : bx_of) h->b_bxrank.push_back(strcmp(b, "null") != 0 ? rk[b] : LANCET_NO_BX);
    for (auto &s2 : h->bx_names) h->bx_ptrs.push_back(s2.c_str());
  }
  if (timing) fprintf(stderr, "[lancet_host] threads %u; %d windows: select %.3f s, size %.3f s, fill %.3f s, barcodes %.3f s\n", host_threads(nwin), nwin, t1 - t0, t2 - t1, t3 - t2, now() - t3);
  memset(out, 0, sizeof *out);
  out->n_windows = nk;
  out->chr_id = h->b_chr.data(); out->ref_start = h->b_refstart.data(); out->ref_off = h->b_refoff.data(); out->ref_bases = h->b_ref.data();
  out->read_begin = h->b_readbegin.data(); out->seq_off = h->b_seqoff.data(); out->seq = packed ? nullptr : h->b_seq.data(); out->qual = packed ? nullptr : h->b_qual.data();
  if (packed) {
    uint32_t *pb = (uint32_t *)h->b_pbases.p, *pg = (uint32_t *)h->b_pgood.p;
    memset(pk, 0, sizeof *pk);
    if (shared) {
      for (int q = 0; q < 4; ++q) pb[ubw_all + (uint64_t)q] = 0;
      pg[ugw_all] = 0;
      pk->read_index = h->b_ridx.data(); pk->n_distinct = (uint32_t)U;
    } else {
    for (int q = 0; q < 4; ++q) pb[bw0[(size_t)nk] + (uint64_t)q] = 0;
    pg[gw0[(size_t)nk]] = 0;
    h->b_rinfo[R] = 0; h->b_bw[R] = (uint32_t)bw0[(size_t)nk]; h->b_gw[R] = (uint32_t)gw0[(size_t)nk];
    }
    pk->struct_size = (uint32_t)sizeof(lancet_packed_reads); pk->reserved = 0;
    pk->rinfo = h->b_rinfo.data(); pk->base_woff = h->b_bw.data(); pk->good_woff = h->b_gw.data(); pk->bases = pb; pk->good = pg;
    pk->min_qual_trim = P->min_qual_trim; pk->min_qual_call = P->min_qual_call;
  }
  out->label = h->b_label.data(); out->strand = h->b_strand.data(); out->mate = h->b_mate.data(); out->mapped = h->b_mapped.data();
  out->name_rank = h->b_namerank.data();
  out->bx_rank = o->linked ? h->b_bxrank.data() : nullptr; out->hp = o->linked ? h->b_hp.data() : nullptr;
  if (n_kept) *n_kept = nk;
  return LANCET_OK;
}

const char *const *lancet_host_bx_names(const lancet_host *h, uint32_t *n) {
  if (n) *n = (uint32_t)h->bx_ptrs.size();
  return h->bx_ptrs.data();
}

}  // extern "C"
