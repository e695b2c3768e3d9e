// host_trace.cc -- the engine's per-window trace events (kernels.h: EV_*) as the text the reference prints with -v
// (reference src/Microassembler.cc:87-246 and the verbose blocks of src/Graph.cc), so that a parity break can be
// localised to one graph stage by diffing against a reference trace.  Host code.
#include "../../include/lancet_engine.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
enum { EV_PROCESS = 1, EV_REPEAT_REF, EV_NEAR_REF, EV_READS, EV_STATS, EV_MARKREF, EV_LOWCOV, EV_CLEANDEAD, EV_COMPRESS, EV_CC,
       EV_CCID, EV_CCEND, EV_TRIM, EV_AMBIG_SRC, EV_NOMATCH_SRC, EV_AMBIG_SNK, EV_NOMATCH_SNK, EV_CYCLE, EV_TIPS_ROUND,
       EV_TIPS_REMOVED, EV_LINKS, EV_LOOKREP, EV_MISSING, EV_SEARCH, EV_NEAR_QRY, EV_DFSLIMIT, EV_PATH, EV_TS, EV_PATH_END,
       EV_EKA_END, EV_FOUND, EV_END };

void put(std::string &o, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void put(std::string &o, const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt);
  const int n = vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (n > 0) o.append(buf, (size_t)(n < (int)sizeof buf ? n : (int)sizeof buf - 1));
}
// raw bytes appended after an event, padded to 8 words
std::string take_bytes(const uint32_t *w, uint32_t nw, uint32_t *at, uint32_t n) {
  const uint32_t nwords = ((n + 3) / 4 + 7) / 8 * 8;
  std::string s;
  for (uint32_t i = 0; i < n && *at + i / 4 < nw; ++i) s.push_back((char)((w[*at + i / 4] >> (8 * (i % 4))) & 0xFF));
  *at += nwords;
  return s;
}
}  // namespace

extern "C" char *lancet_trace_format(const uint32_t *words, uint32_t n_words, int32_t idx1, const char *hdr, const char *chrom,
                                     int32_t start, int32_t end, int32_t dfs_limit) {
  std::string out;
  std::vector<uint32_t> ccids;
  uint32_t cc_nodes = 0;
  uint32_t i = 0;
  while (i + 8 <= n_words) {
    const uint32_t code = words[i], a = words[i + 1], b = words[i + 2], c = words[i + 3], d = words[i + 4], e = words[i + 5], f = words[i + 6], g = words[i + 7];
    i += 8;
    switch (code) {
      case EV_PROCESS: put(out, "== Processing %d: %s numsequences: %u mapped: %u bastards: %u\n", idx1, hdr, a, b, a - b); out.append(53, '='); out.push_back('\n'); break;
      case EV_REPEAT_REF: put(out, "Repeat in reference sequence for kmer %u\n", a); break;
      case EV_NEAR_REF: put(out, "Near-perfect repeat in reference sequence for kmer %u\n", a); break;
      case EV_READS: put(out, "reads: %u reflen: %u readlen: %u cov: %.1f\n", a, b, c, (double)c / (double)b); break;
      case EV_STATS: put(out, "  %u: nodes: %u edges: %u span: %u\n", a, b, c, d); break;
      case EV_MARKREF: put(out, "\nmark refnodes\n nodes: %u refnodes: %u\n", a, b); break;
      case EV_LOWCOV: put(out, "\nremoving low coverage: found %u", a); break;
      case EV_CLEANDEAD: put(out, "  removing %u dead nodes\n", a); break;
      case EV_COMPRESS: out += "compressing graph:"; break;
      case EV_CC: out += "\nconnected components\n"; ccids.clear(); cc_nodes = a; break;
      case EV_CCID: ccids.push_back(a); break;
      case EV_CCEND:
        put(out, " nodes: %u refnodes: 0 comp: %u refcomp: %u refcompids: ", cc_nodes, a, b);
        for (uint32_t x : ccids) put(out, " %u", x);
        out.push_back('\n');
        break;
      case EV_TRIM: put(out, "ref trim5: %u trim3: %u uncovered: %u ref_dist: %u\n", a, b, a + b, c); break;
      case EV_AMBIG_SRC: out += "Ambiguous match to reference for source\n"; break;
      case EV_NOMATCH_SRC: out += "No match to reference for source\n"; break;
      case EV_AMBIG_SNK: out += "Ambiguous match to reference for sink\n"; break;
      case EV_NOMATCH_SNK: out += "No match to reference for sink\n"; break;
      case EV_CYCLE: put(out, "Cycle found in the graph (kmer = %u)!\n", a); break;
      case EV_TIPS_ROUND: put(out, "\nremove tips round: %u", a); break;
      case EV_TIPS_REMOVED: put(out, " removed: %u\n", a); break;
      case EV_LINKS: put(out, "\nremove short links:  removed links: %u\n", a); break;
      case EV_LOOKREP: out += "\nlooking for near-perfect repeats:\n"; break;
      case EV_MISSING: out += "Missing source or sink\n"; break;
      case EV_SEARCH: put(out, "\nsearching from source%u to sink%u dir: F\n", a, a); break;
      case EV_NEAR_QRY: put(out, "Near-perfect repeat in assembled sequence for kmer %u\n", a); break;
      case EV_DFSLIMIT: put(out, "WARNING: DFS_LIMIT (%d) exceeded\n", dfs_limit); break;
      case EV_PATH: put(out, ">p_%s:%d-%d_%u cycle: %u match: %u snp: %u ins: %u del: %u", chrom, start, end, a, b, c, d, e, f); break;
      case EV_TS: {
        const std::string ref = take_bytes(words, n_words, &i, b), qry = take_bytes(words, n_words, &i, b);
        uint32_t h[12];                                  // 12 x u16: HPRN HPRT HPAN HPAT, each {hp1, hp2, hp0}
        for (int q = 0; q < 12; ++q) h[q] = (i + (uint32_t)q / 2 < n_words) ? (words[i + (uint32_t)q / 2] >> (16 * (q % 2))) & 0xFFFFu : 0u;
        i += 8;
        // printed as hp0,hp1,hp2 (reference src/Graph.cc:1138-1141)
        put(out, " %u:", a); out += ref; out.push_back('|'); out += qry;
        put(out, "|R:(%u+,%u-)n,(%u+,%u-)t|A:(%u+,%u-)n,(%u+,%u-)t|HPref(%u,%u,%u)n,(%u,%u,%u)t|HPalt(%u,%u,%u)n,(%u,%u,%u)t|%c|%c",
            c >> 16, c & 0xFFFFu, d >> 16, d & 0xFFFFu, e >> 16, e & 0xFFFFu, f >> 16, f & 0xFFFFu,
            h[2], h[0], h[1], h[5], h[3], h[4], h[8], h[6], h[7], h[11], h[9], h[10], (int)(g >> 8), (int)(g & 0xFF));
        break;
      }
      case EV_PATH_END: out.push_back('\n'); break;
      case EV_EKA_END:
        put(out, " refcomp: %u refnodes: -2 complete: %u allcycles: %u\n perfect: %u withsnps: %u withindel: %u withmix: %u withmixindel: %u\n",
            a, b, c, d, e, f, g, g + f);
        break;
      case EV_FOUND: put(out, " Found %u on ref path\n", a); break;
      case EV_END:
        if (a) out += " Found repeat in reference\n";
        if (b) out += " Found repeat in assembly\n";
        if (c) out += " Found cycle in assembly\n";
        out += "FINISHED\n";
        break;
      default: break;
    }
  }
  char *r = (char *)malloc(out.size() + 1);
  if (!r) return nullptr;
  memcpy(r, out.c_str(), out.size() + 1);
  return r;
}
