// extern "C" entry point around the reference's own global_align_aff (reference src/align.hh, src/align.cc:235).
// Test infrastructure only: lets tests/ call the real reference alignment through ctypes.
#include <cstring>
#include <string>
#include "align.hh"

extern "C" int lancet_ref_align(const char *S, const char *T, char *S_aln, char *T_aln, int cap) {
  std::string a, b;
  global_align_aff(std::string(S), std::string(T), a, b, 0, 0);
  if ((int)a.size() + 1 > cap) return -1;
  memcpy(S_aln, a.c_str(), a.size() + 1);
  memcpy(T_aln, b.c_str(), b.size() + 1);
  return (int)a.size();
}
