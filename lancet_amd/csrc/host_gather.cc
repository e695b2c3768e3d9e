// host_gather.cc -- include/lancet_gather.h: the records of an N-process run packed, gathered to rank 0 over RCCL and replayed into
// the VariantDB in (global window, emission) order.  Host code; what it replaces is the reference's merge of per-thread databases
// (reference src/Lancet.cc:940-959) -- see the header.  The byte format is lancet_amd/dist.py's (pack_records / unpack_records):
//   8 x u64  n records, blob bytes, has_lr, barcode ids, bytes of barcode names, bytes of contig names, bytes of keys, 0
//   records | blob | [lr records | barcode ids (u32, indices into the names) | names joined by NUL] | contig names joined by NUL | keys
#include <dlfcn.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../include/lancet_gather.h"

namespace {

struct Hdr { uint64_t n, blob, has_lr, nids, names, chrs, keys, zero; };
static_assert(sizeof(Hdr) == 64 && sizeof(lancet_variant) == 64 && sizeof(lancet_variant_lr) == 64, "wire format");

std::string join0(const char *const *names, size_t n) {
  std::string s;
  for (size_t i = 0; i < n; ++i) { if (i) s.push_back('\0'); s += names[i] ? names[i] : ""; }
  return s;
}
std::vector<std::string> split0(const uint8_t *p, size_t len) {
  std::vector<std::string> out;
  if (!len) return out;
  size_t a = 0;
  for (size_t i = 0; i <= len; ++i) if (i == len || p[i] == 0) { out.emplace_back((const char *)p + a, i - a); a = i + 1; }
  return out;
}
bool in_replay_order(const lancet_variant *v, size_t n) {
  for (size_t i = 1; i < n; ++i) if (!(v[i].window > v[i - 1].window || (v[i].window == v[i - 1].window && v[i].seq_in_window >= v[i - 1].seq_in_window))) return false;
  return true;
}

}  // namespace

extern "C" int lancet_records_pack(const lancet_variant *v, uint32_t n, const char *blob, uint32_t blob_len,
                                   const lancet_variant_lr *lr, const uint32_t *bx_blob, const char *const *bx_names, uint32_t n_bx,
                                   const char *const *chr_names, int32_t n_chr, const int64_t *window_index, uint32_t n_windows,
                                   int reduce, uint8_t **out, size_t *out_len) {
  if (!out || !out_len || !chr_names || n_chr <= 0 || (n && (!v || !blob))) return LANCET_E_ARG;
  *out = nullptr; *out_len = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (v[i].chr_id < 0 || v[i].chr_id >= n_chr) return LANCET_E_ARG;
    if (window_index && (v[i].window < 0 || (uint32_t)v[i].window >= n_windows)) return LANCET_E_ARG;
    if ((uint64_t)v[i].ref_off + v[i].ref_len > blob_len || (uint64_t)v[i].alt_off + v[i].alt_len > blob_len || (uint64_t)v[i].str_off + v[i].str_len > blob_len) return LANCET_E_ARG;
  }
  bool mono = true;
  if (window_index) for (uint32_t w = 1; w < n_windows; ++w) if (window_index[w] <= window_index[w - 1]) mono = false;
  // keys where the records are made; the reduction only on records that are in the order rank 0 replays them in
  std::vector<uint8_t> keys((size_t)n * 32), keep((size_t)n, 1);
  if (n) {
    std::string z(blob, blob_len); z.push_back('\0');
    const int rc = lancet_vdb_keys(v, n, z.c_str(), chr_names, n_chr, keys.data());
    if (rc != LANCET_OK) return rc;
    if (reduce && mono && in_replay_order(v, n)) { const int r2 = lancet_vdb_reduce(v, keys.data(), n, keep.data()); if (r2 != LANCET_OK) return r2; }
  }
  const bool has_lr = lr != nullptr;
  std::vector<uint32_t> ids; std::string names;
  if (has_lr && n) {
    uint64_t nbx = 0;
    for (uint32_t i = 0; i < n; ++i) for (int q = 0; q < 4; ++q) nbx = std::max<uint64_t>(nbx, (uint64_t)lr[i].bx_off[q] + lr[i].bx_len[q]);
    if (nbx && (!bx_blob || !bx_names)) return LANCET_E_ARG;
    std::vector<uint32_t> used(bx_blob, bx_blob + nbx);
    std::sort(used.begin(), used.end()); used.erase(std::unique(used.begin(), used.end()), used.end());
    for (uint32_t u : used) if (u >= n_bx) return LANCET_E_ARG;
    ids.resize(nbx);
    for (uint64_t i = 0; i < nbx; ++i) ids[i] = (uint32_t)(std::lower_bound(used.begin(), used.end(), bx_blob[i]) - used.begin());     // only the barcodes that occur travel
    for (size_t i = 0; i < used.size(); ++i) { if (i) names.push_back('\0'); names += bx_names[used[i]]; }
  }
  const std::string chrs = join0(chr_names, (size_t)n_chr);
  uint32_t nk = 0; for (uint32_t i = 0; i < n; ++i) nk += keep[i] ? 1u : 0u;
  Hdr h = {nk, blob_len, has_lr ? 1u : 0u, ids.size(), names.size(), chrs.size(), (uint64_t)nk * 32, 0};
  const size_t total = sizeof(Hdr) + (size_t)nk * 64 + blob_len + (has_lr ? (size_t)nk * 64 + 4 * ids.size() + names.size() : 0) + chrs.size() + (size_t)nk * 32;
  uint8_t *b = (uint8_t *)malloc(total ? total : 1);
  if (!b) return LANCET_E_OOM;
  size_t o = 0;
  memcpy(b, &h, sizeof h); o += sizeof h;
  for (uint32_t i = 0; i < n; ++i) if (keep[i]) {
    lancet_variant r = v[i];
    if (window_index) r.window = (int32_t)window_index[r.window];
    memcpy(b + o, &r, 64); o += 64;
  }
  if (blob_len) memcpy(b + o, blob, blob_len);
  o += blob_len;
  if (has_lr) {
    for (uint32_t i = 0; i < n; ++i) if (keep[i]) { memcpy(b + o, &lr[i], 64); o += 64; }
    if (!ids.empty()) memcpy(b + o, ids.data(), 4 * ids.size());
    o += 4 * ids.size();
    if (!names.empty()) memcpy(b + o, names.data(), names.size());
    o += names.size();
  }
  if (!chrs.empty()) memcpy(b + o, chrs.data(), chrs.size());
  o += chrs.size();
  for (uint32_t i = 0; i < n; ++i) if (keep[i]) { memcpy(b + o, keys.data() + (size_t)i * 32, 32); o += 32; }
  *out = b; *out_len = o;
  return LANCET_OK;
}

extern "C" int lancet_records_merge(lancet_vdb *db, const uint8_t *const *parts, const size_t *lens, int n_parts, uint32_t *n_added) {
  if (n_added) *n_added = 0;
  if (!db || n_parts < 0 || (n_parts && (!parts || !lens))) return LANCET_E_ARG;
  struct Part { Hdr h; const uint8_t *recs, *blob, *lr, *ids, *names, *chrs, *keys; };
  std::vector<Part> ps;
  for (int i = 0; i < n_parts; ++i) {
    if (!parts[i] || lens[i] == 0) continue;
    if (lens[i] < sizeof(Hdr)) return LANCET_E_ARG;
    Part p; memcpy(&p.h, parts[i], sizeof(Hdr));
    const Hdr &h = p.h;
    if (h.n > 0x7FFFFFFFull || h.blob > 0xFFFFFFFFull || (h.keys != 0 && h.keys != h.n * 32)) return LANCET_E_ARG;
    const uint64_t need = sizeof(Hdr) + h.n * 64 + h.blob + (h.has_lr ? h.n * 64 + 4 * h.nids + h.names : 0) + h.chrs + h.keys;
    if (need > lens[i]) return LANCET_E_ARG;
    const uint8_t *q = parts[i] + sizeof(Hdr);
    p.recs = q; q += h.n * 64; p.blob = q; q += h.blob;
    p.lr = p.ids = p.names = nullptr;
    if (h.has_lr) { p.lr = q; q += h.n * 64; p.ids = q; q += 4 * h.nids; p.names = q; q += h.names; }
    p.chrs = q; q += h.chrs; p.keys = h.keys ? q : nullptr;
    if (h.n) ps.push_back(p);
  }
  if (ps.empty()) return LANCET_OK;
  const bool lr_mode = ps[0].h.has_lr != 0;
  bool keyed = true; uint64_t total = 0;
  for (const Part &p : ps) { if ((p.h.has_lr != 0) != lr_mode) return LANCET_E_ARG; if (!p.keys) keyed = false; total += p.h.n; }
  if (total > 0xFFFFFFFFull) return LANCET_E_ARG;
  std::vector<lancet_variant> recs((size_t)total);
  std::vector<lancet_variant_lr> lrs(lr_mode ? (size_t)total : 0);
  std::vector<uint8_t> keys(keyed ? (size_t)total * 32 : 0);
  std::vector<std::string> chr_names, bx_names;
  std::map<std::string, uint32_t> bx_index;
  std::vector<uint32_t> ids_all;
  std::string blob;
  size_t o = 0;
  for (const Part &p : ps) {
    const size_t n = (size_t)p.h.n;
    memcpy(&recs[o], p.recs, n * 64);
    if (keyed) memcpy(&keys[o * 32], p.keys, n * 32);
    const std::vector<std::string> cn = split0(p.chrs, (size_t)p.h.chrs);
    std::vector<int32_t> cmap(std::max<size_t>(1, cn.size()), 0);
    for (size_t i = 0; i < cn.size(); ++i) {
      size_t at = std::find(chr_names.begin(), chr_names.end(), cn[i]) - chr_names.begin();
      if (at == chr_names.size()) chr_names.push_back(cn[i]);
      cmap[i] = (int32_t)at;
    }
    const uint32_t blob_base = (uint32_t)blob.size();
    if ((uint64_t)blob.size() + p.h.blob > 0xFFFFFFF0ull) return LANCET_E_ARG;
    for (size_t i = 0; i < n; ++i) {
      lancet_variant &r = recs[o + i];
      if (r.chr_id < 0 || (size_t)r.chr_id >= cn.size()) return LANCET_E_ARG;
      if ((uint64_t)r.ref_off + r.ref_len > p.h.blob || (uint64_t)r.alt_off + r.alt_len > p.h.blob || (uint64_t)r.str_off + r.str_len > p.h.blob) return LANCET_E_ARG;
      r.chr_id = cmap[(size_t)r.chr_id]; r.ref_off += blob_base; r.alt_off += blob_base; r.str_off += blob_base;
    }
    blob.append((const char *)p.blob, (size_t)p.h.blob);
    if (lr_mode) {
      memcpy(&lrs[o], p.lr, n * 64);
      const std::vector<std::string> bn = split0(p.names, (size_t)p.h.names);
      std::vector<uint32_t> gmap(std::max<size_t>(1, bn.size()), 0);
      for (size_t i = 0; i < bn.size(); ++i) {
        auto it = bx_index.find(bn[i]);
        if (it == bx_index.end()) { it = bx_index.emplace(bn[i], (uint32_t)bx_names.size()).first; bx_names.push_back(bn[i]); }
        gmap[i] = it->second;
      }
      const uint32_t id_base = (uint32_t)ids_all.size();
      for (size_t i = 0; i < n; ++i) for (int q = 0; q < 4; ++q) {
        if ((uint64_t)lrs[o + i].bx_off[q] + lrs[o + i].bx_len[q] > p.h.nids) return LANCET_E_ARG;
        lrs[o + i].bx_off[q] += id_base;
      }
      for (uint64_t i = 0; i < p.h.nids; ++i) { uint32_t id; memcpy(&id, p.ids + 4 * i, 4); if (id >= bn.size()) return LANCET_E_ARG; ids_all.push_back(gmap[id]); }
    }
    o += n;
  }
  // ranks that hold runs of windows arrive in (window, emission) order already: look before sorting
  if (!in_replay_order(recs.data(), recs.size())) {
    std::vector<uint32_t> ord(recs.size());
    std::iota(ord.begin(), ord.end(), 0u);
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
      return recs[a].window != recs[b].window ? recs[a].window < recs[b].window : recs[a].seq_in_window < recs[b].seq_in_window; });
    std::vector<lancet_variant> r2(recs.size());
    for (size_t i = 0; i < ord.size(); ++i) r2[i] = recs[ord[i]];
    recs.swap(r2);
    if (keyed) { std::vector<uint8_t> k2(keys.size()); for (size_t i = 0; i < ord.size(); ++i) memcpy(&k2[i * 32], &keys[(size_t)ord[i] * 32], 32); keys.swap(k2); }
    if (lr_mode) { std::vector<lancet_variant_lr> l2(lrs.size()); for (size_t i = 0; i < ord.size(); ++i) l2[i] = lrs[ord[i]]; lrs.swap(l2); }
  }
  std::vector<const char *> cn; for (const std::string &s : chr_names) cn.push_back(s.c_str());
  int rc;
  if (lr_mode) {
    // ids become ranks in the union's name order: a set stays sorted by name (std::set<string>) because every part's ids were ranks by name
    std::vector<uint32_t> by_name(bx_names.size()); std::iota(by_name.begin(), by_name.end(), 0u);
    std::sort(by_name.begin(), by_name.end(), [&](uint32_t a, uint32_t b) { return bx_names[a] < bx_names[b]; });
    std::vector<uint32_t> rank_of(bx_names.size());
    std::vector<const char *> sorted_names(bx_names.size());
    for (size_t i = 0; i < by_name.size(); ++i) { rank_of[by_name[i]] = (uint32_t)i; sorted_names[i] = bx_names[by_name[i]].c_str(); }
    for (uint32_t &x : ids_all) x = rank_of[x];
    if (ids_all.empty()) ids_all.push_back(0);
    if (keyed) rc = lancet_vdb_add_keyed(db, recs.data(), lrs.data(), keys.data(), (uint32_t)recs.size(), blob.c_str(), ids_all.data(), sorted_names.data(), (uint32_t)sorted_names.size(), cn.data(), (int32_t)cn.size());
    else rc = lancet_vdb_add_lr(db, recs.data(), lrs.data(), (uint32_t)recs.size(), blob.c_str(), ids_all.data(), sorted_names.data(), (uint32_t)sorted_names.size(), cn.data(), (int32_t)cn.size());
  } else if (keyed) rc = lancet_vdb_add_keyed(db, recs.data(), nullptr, keys.data(), (uint32_t)recs.size(), blob.c_str(), nullptr, nullptr, 0, cn.data(), (int32_t)cn.size());
  else rc = lancet_vdb_add(db, recs.data(), (uint32_t)recs.size(), blob.c_str(), cn.data(), (int32_t)cn.size());
  if (rc == LANCET_OK && n_added) *n_added = (uint32_t)recs.size();
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// transport
// ---------------------------------------------------------------------------------------------------------------------------------
struct lancet_comm {
  int rank = 0, world = 1, device = 0;
  bool files = false;
  std::string path, err;
  void *lib = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  uint64_t *d_sizes = nullptr;
  int round = 0;
  decltype(&ncclGetUniqueId) p_uid = nullptr;
  decltype(&ncclCommInitRank) p_init = nullptr;
  decltype(&ncclAllGather) p_allgather = nullptr;
  decltype(&ncclSend) p_send = nullptr;
  decltype(&ncclRecv) p_recv = nullptr;
  decltype(&ncclGroupStart) p_gstart = nullptr;
  decltype(&ncclGroupEnd) p_gend = nullptr;
  decltype(&ncclCommDestroy) p_destroy = nullptr;
  decltype(&ncclGetErrorString) p_errstr = nullptr;
};

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool write_file_atomic(const std::string &path, const void *p, size_t n) {
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = (n == 0 || fwrite(p, 1, n, f) == n);
  if (fclose(f) != 0 || !ok) { unlink(tmp.c_str()); return false; }
  return rename(tmp.c_str(), path.c_str()) == 0;
}
bool read_file_when_there(const std::string &path, std::vector<uint8_t> &out, double deadline) {
  while (true) {
    struct stat st;
    if (stat(path.c_str(), &st) == 0) {
      FILE *f = fopen(path.c_str(), "rb");
      if (f) { out.resize((size_t)st.st_size); const size_t got = st.st_size ? fread(out.data(), 1, (size_t)st.st_size, f) : 0; fclose(f); if (got == (size_t)st.st_size) return true; }
    }
    if (now_s() > deadline) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
}
int fail(lancet_comm *c, const std::string &m, int code = LANCET_E_HIP) { c->err = m; return code; }
#define NCHK(c, call) do { const ncclResult_t _r = (call); if (_r != ncclSuccess) return fail(c, std::string(#call) + ": " + (c->p_errstr ? c->p_errstr(_r) : "rccl error")); } while (0)
#define HCHK(c, call) do { const hipError_t _e = (call); if (_e != hipSuccess) return fail(c, std::string(#call) + ": " + hipGetErrorString(_e)); } while (0)

int comm_open_rccl(lancet_comm *c, double timeout_s) {
  c->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!c->lib) c->lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!c->lib) return fail(c, std::string("cannot load librccl.so.1: ") + dlerror(), LANCET_E_NO_DEVICE);
#define SYM(field, name) do { c->field = (decltype(c->field))dlsym(c->lib, name); if (!c->field) return fail(c, std::string("librccl has no ") + name, LANCET_E_NO_DEVICE); } while (0)
  SYM(p_uid, "ncclGetUniqueId"); SYM(p_init, "ncclCommInitRank"); SYM(p_allgather, "ncclAllGather"); SYM(p_send, "ncclSend"); SYM(p_recv, "ncclRecv");
  SYM(p_gstart, "ncclGroupStart"); SYM(p_gend, "ncclGroupEnd"); SYM(p_destroy, "ncclCommDestroy"); SYM(p_errstr, "ncclGetErrorString");
#undef SYM
  HCHK(c, hipSetDevice(c->device));
  ncclUniqueId id;
  if (c->rank == 0) {
    NCHK(c, c->p_uid(&id));
    if (!write_file_atomic(c->path, &id, sizeof id)) return fail(c, "cannot write the rendezvous file " + c->path + ": " + strerror(errno), LANCET_E_ARG);
  } else {
    std::vector<uint8_t> b;
    if (!read_file_when_there(c->path, b, now_s() + timeout_s) || b.size() != sizeof id) return fail(c, "rank 0's id did not appear at " + c->path, LANCET_E_STATE);
    memcpy(&id, b.data(), sizeof id);
  }
  {
    // RCCL prints a version banner on STDOUT when a communicator comes up -- the stream rank 0 writes the VCF to.  For the time of the call
    // file descriptor 1 points at stderr.
    fflush(stdout);
    const int saved = dup(1);
    if (saved >= 0) dup2(2, 1);
    const ncclResult_t r = c->p_init(&c->comm, c->world, id, c->rank);
    fflush(nullptr);
    if (saved >= 0) { dup2(saved, 1); close(saved); }
    if (r != ncclSuccess) return fail(c, std::string("ncclCommInitRank: ") + c->p_errstr(r) + " (NCCL_DEBUG=WARN says why; two ranks on one GPU are refused)");
  }
  HCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HCHK(c, hipMalloc((void **)&c->d_sizes, 8 * ((size_t)c->world + 1)));
  return LANCET_OK;
}
}  // namespace

extern "C" lancet_comm *lancet_comm_create(int rank, int world, int device, const char *rendezvous, double timeout_s, char *err, size_t errlen) {
  auto say = [&](const std::string &m) { if (err && errlen) { snprintf(err, errlen, "%s", m.c_str()); } };
  if (world < 1 || rank < 0 || rank >= world || !rendezvous || !*rendezvous) { say("lancet_comm_create: rank / world / rendezvous path"); return nullptr; }
  lancet_comm *c = new lancet_comm();
  c->rank = rank; c->world = world; c->device = device; c->path = rendezvous;
  const char *t = getenv("LANCET_COMM_TEST_FILES");
  c->files = t && *t && strcmp(t, "0") != 0;
  if (!c->files) {
    const int rc = comm_open_rccl(c, timeout_s > 0 ? timeout_s : 120.0);
    if (rc != LANCET_OK) { say(c->err); lancet_comm_destroy(c); return nullptr; }
  }
  return c;
}

extern "C" int lancet_comm_gather(lancet_comm *c, const uint8_t *payload, size_t len, uint8_t **all, size_t *lens) {
  if (!c || !all || (len && !payload) || (c->rank == 0 && !lens)) return LANCET_E_ARG;
  *all = nullptr;
  const int W = c->world;
  const int round = c->round++;
  if (c->files) {                                   // test transport: one file per (round, rank) next to the rendezvous path
    const std::string base = c->path + ".r" + std::to_string(round) + ".";
    if (c->rank != 0) return write_file_atomic(base + std::to_string(c->rank), payload, len) ? LANCET_OK : fail(c, "cannot write " + base, LANCET_E_ARG);
    std::vector<std::vector<uint8_t>> got((size_t)W);
    size_t total = len;
    for (int r = 1; r < W; ++r) {
      if (!read_file_when_there(base + std::to_string(r), got[(size_t)r], now_s() + 600.0)) return fail(c, "rank " + std::to_string(r) + " sent nothing", LANCET_E_STATE);
      unlink((base + std::to_string(r)).c_str());
      total += got[(size_t)r].size();
    }
    uint8_t *b = (uint8_t *)malloc(total ? total : 1);
    if (!b) return fail(c, "out of memory", LANCET_E_OOM);
    size_t o = 0;
    for (int r = 0; r < W; ++r) {
      const uint8_t *src = r == 0 ? payload : got[(size_t)r].data(); const size_t n = r == 0 ? len : got[(size_t)r].size();
      if (n) memcpy(b + o, src, n);
      lens[r] = n; o += n;
    }
    *all = b;
    return LANCET_OK;
  }
  HCHK(c, hipSetDevice(c->device));
  // sizes: one all-gather of 8 bytes per rank
  const uint64_t mine = (uint64_t)len;
  HCHK(c, hipMemcpyAsync(c->d_sizes + W, &mine, 8, hipMemcpyHostToDevice, c->stream));
  NCHK(c, c->p_allgather(c->d_sizes + W, c->d_sizes, 1, ncclUint64, c->comm, c->stream));
  std::vector<uint64_t> sizes((size_t)W);
  HCHK(c, hipMemcpyAsync(sizes.data(), c->d_sizes, 8 * (size_t)W, hipMemcpyDeviceToHost, c->stream));
  HCHK(c, hipStreamSynchronize(c->stream));
  if (sizes[(size_t)c->rank] != mine) return fail(c, "the size exchange returned another size for this rank", LANCET_E_STATE);
  // payloads: point to point, to rank 0 only
  if (c->rank != 0) {
    if (len) {
      uint8_t *d = nullptr;
      HCHK(c, hipMalloc((void **)&d, len));
      HCHK(c, hipMemcpyAsync(d, payload, len, hipMemcpyHostToDevice, c->stream));
      const ncclResult_t r = c->p_send(d, len, ncclUint8, 0, c->comm, c->stream);
      const hipError_t e = hipStreamSynchronize(c->stream);
      (void)hipFree(d);
      if (r != ncclSuccess) return fail(c, std::string("ncclSend: ") + c->p_errstr(r));
      if (e != hipSuccess) return fail(c, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    return LANCET_OK;
  }
  size_t total = 0, remote = 0;
  for (int r = 0; r < W; ++r) { lens[r] = (size_t)sizes[(size_t)r]; total += lens[r]; if (r) remote += lens[r]; }
  uint8_t *b = (uint8_t *)malloc(total ? total : 1);
  if (!b) return fail(c, "out of memory", LANCET_E_OOM);
  if (len) memcpy(b, payload, len);
  if (remote) {                                     // ONE receive buffer, every receive posted in one group, one copy to the host
    uint8_t *d = nullptr;
    if (hipMalloc((void **)&d, remote) != hipSuccess) { free(b); return fail(c, "hipMalloc (receive buffer)", LANCET_E_OOM); }
    ncclResult_t r = c->p_gstart();
    size_t o = 0;
    for (int q = 1; q < W && r == ncclSuccess; ++q) if (lens[q]) { r = c->p_recv(d + o, lens[q], ncclUint8, q, c->comm, c->stream); o += lens[q]; }
    const ncclResult_t r2 = c->p_gend();
    hipError_t e = hipMemcpyAsync(b + len, d, remote, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (r != ncclSuccess || r2 != ncclSuccess) { free(b); return fail(c, std::string("ncclRecv: ") + c->p_errstr(r != ncclSuccess ? r : r2)); }
    if (e != hipSuccess) { free(b); return fail(c, std::string("receive copy: ") + hipGetErrorString(e)); }
  }
  *all = b;
  return LANCET_OK;
}

extern "C" const char *lancet_comm_last_error(const lancet_comm *c) { return c ? c->err.c_str() : "no communicator"; }
extern "C" const char *lancet_comm_transport(const lancet_comm *c) { return c && c->files ? "files" : "rccl"; }

extern "C" void lancet_comm_destroy(lancet_comm *c) {
  if (!c) return;
  if (c->d_sizes) (void)hipFree(c->d_sizes);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->comm && c->p_destroy) (void)c->p_destroy(c->comm);
  if (c->rank == 0 && !c->path.empty()) unlink(c->path.c_str());
  // (librccl stays loaded: unloading a library that owns device state at exit is asking for trouble)
  delete c;
}
