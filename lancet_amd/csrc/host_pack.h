// host_pack.h -- one read of a batch, trimmed and packed on the host: what the engine keeps of it in HBM.
// Shared by lancet_engine_upload (engine.hip: packs the caller's ASCII arrays on host threads), lancet_pack_read (the same routine for
// a caller that packs itself) and lancet_host_batch_packed (host_frontend.cc: packs while it assembles the batch).  Host code only.
#pragma once
#include <stdint.h>
#include <string.h>
#include "../../include/lancet_engine.h"

// Graph_t::trim (reference src/Graph.cc:355-384) + 2-bit packing + quality mask on the host: the scalar twin of prep_kernel, one read
struct LcCodeTab { uint8_t c[256]; LcCodeTab() { for (int i = 0; i < 256; ++i) c[i] = 4; c['A'] = c['a'] = 0; c['C'] = c['c'] = 1; c['G'] = c['g'] = 2; c['T'] = c['t'] = 3; } };
static const LcCodeTab lc_tab;
static inline uint32_t lc_code(char b) { return lc_tab.c[(uint8_t)b]; }
// the per-read word of the resident batch: trimmed length, sample, strand, mate number, mapped (layout.h RI_*)
static inline uint32_t lc_rinfo_word(uint32_t tlen, uint8_t label, uint8_t strand, uint8_t mate, uint8_t mapped) {
  return tlen | ((label == LANCET_NML ? 1u : 0u) << 16) | ((strand == LANCET_REV ? 1u : 0u) << 17) | ((uint32_t)(mate & 3) << 18) | ((mapped ? 1u : 0u) << 20);
}
static inline void lc_prep_read_host(const lancet_params &P, const char *sq, const char *ql, int len, uint8_t label, uint8_t strand, uint8_t mate, uint8_t mapped,
                              uint32_t *rinfo, uint32_t *bases, uint32_t *good) {
  const uint8_t *tab = lc_tab.c;
  const int qtrim = P.min_qual_trim, qcall = P.min_qual_call;
  int fg = 0; while (fg < len && !(tab[(uint8_t)sq[fg]] < 4 && !(ql[fg] < qtrim))) ++fg;
  int lg = len - 1; while (lg >= fg && !(tab[(uint8_t)sq[lg]] < 4 && !(ql[lg] < qtrim))) --lg;
  bool junk = fg >= len || lg < fg;
  if (!junk) { uint32_t any = 0; for (int p = fg; p <= lg; ++p) any |= tab[(uint8_t)sq[p]]; junk = (any & 4u) != 0; }
  const int trim5 = junk ? 0 : fg;
  int tlen = junk ? 0 : lg - fg + 1;
  if (tlen > 0xFFFF) tlen = 0xFFFF;
  *rinfo = lc_rinfo_word((uint32_t)tlen, label, strand, mate, mapped);
  const uint8_t *s = (const uint8_t *)sq + trim5; const char *q = ql + trim5;
  // Bases: inside the trimmed range every character is one of ACGTacgt (else the read is junk, above), and for those
  // ((c >> 1) ^ (c >> 2)) & 3 is the code 0..3; four characters of a 32-bit word are gathered into eight bits by one multiplication
  // (t * 0x01041040 puts character i's two bits at 24 + 2i; the cross terms fall below bit 24 or out of the word).
  const int nfull = tlen / 16;
  for (int wv = 0; wv < nfull; ++wv) {
    const uint8_t *x = s + wv * 16; uint32_t v = 0;
    for (int g = 0; g < 4; ++g) {
      uint32_t c4; memcpy(&c4, x + 4 * g, 4);
      const uint32_t t = ((c4 >> 1) ^ (c4 >> 2)) & 0x03030303u;
      v |= ((t * 0x01041040u) >> 24) << (8 * g);
    }
    bases[wv] = v;
  }
  if (tlen & 15) { uint32_t v = 0; for (int j = 0; j < (tlen & 15); ++j) v |= (uint32_t)(tab[s[nfull * 16 + j]] & 3u) << (2 * j); bases[nfull] = v; }
  // Quality mask: `q >= qcall` on (signed) characters, eight at a time: with the top bit masked off, adding 128 - qcall carries into bit 7
  // exactly when the character reaches qcall; a character with its top bit set is negative and never does.  (A threshold outside 1..127
  // takes the plain loop.)
  const int gfull = tlen / 32;
  const bool swar = qcall >= 1 && qcall <= 127;
  const uint64_t kadd = 0x0101010101010101ULL * (uint64_t)(128 - (swar ? qcall : 1));
  for (int wv = 0; wv < gfull; ++wv) {
    const char *x = q + wv * 32; uint32_t v = 0;
    if (swar) {
      for (int g = 0; g < 4; ++g) {
        uint64_t c8; memcpy(&c8, x + 8 * g, 8);
        const uint64_t hit = (((c8 & 0x7F7F7F7F7F7F7F7FULL) + kadd) & ~c8) & 0x8080808080808080ULL;
        v |= (uint32_t)(((hit >> 7) * 0x0102040810204080ULL) >> 56) << (8 * g);     // bit 0 of byte i -> bit i of the top byte
      }
    } else for (int j = 0; j < 32; ++j) v |= (uint32_t)(x[j] >= qcall) << j;
    good[wv] = v;
  }
  if (tlen & 31) { uint32_t v = 0; for (int j = 0; j < (tlen & 31); ++j) v |= (uint32_t)(q[gfull * 32 + j] >= qcall) << j; good[gfull] = v; }
}
