// lancet_main.cc -- `lancet_gpu`: the reference's command line (reference src/Lancet.cc:655-800) on the native host side
// (include/lancet_host.h) and the MI355X engine (include/lancet_engine.h).
//
//     lancet_gpu --tumor T.bam --normal N.bam --ref ref.fa --reg chr22:1000-5000 > out.vcf
//
// There is no CPU path: without a gfx950 device engine creation fails and the program stops with an error.
// Beyond the reference's options: --device N | --devices a,b,... (one engine per entry; batches of --batch-windows windows
// go to the engines in turn while the host prepares the next one; records reach the VariantDB in window order whatever
// the number of engines), --strict (no VCF at all when a window exceeded the engine's work space; by default the run
// finishes, the windows are listed on stderr and the exit code is 3).
// --ranks N: N processes, one per GPU (rank r on --devices[r mod #devices], default device r): the batches of --batch-windows windows are
// dealt out to the ranks in contiguous runs, every rank keys and reduces its records (lancet_records_pack) and they are gathered to rank 0 over RCCL
// (lancet_comm_gather: sizes by one all-gather, payloads by send / recv to rank 0 only) and replayed into the VariantDB in window order
// (lancet_records_merge) -- what the reference does with its per-thread databases at the end of main() (reference src/Lancet.cc:940-959).
// Without --rank the program starts the N ranks itself (children of this process, --rank r --rendezvous <path> appended) and waits for
// them; a job launcher may start the ranks itself with --rank / --rendezvous.  The VCF (rank 0's stdout) does not depend on N.
// Not offered: --kmer-recovery, --print-graph; --num-threads is accepted and ignored (windows are
// batched on the GPU); -v prints the reference's per-window stage trace to stderr.
#include "../../include/lancet_host.h"
#include "../../include/lancet_gather.h"

#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <future>
#include <map>
#include <string>
#include <vector>

namespace {
struct Opt { const char *lng; char shrt; int has_arg; };
const Opt OPTS[] = {
  {"tumor", 't', 1}, {"normal", 'n', 1}, {"ref", 'r', 1}, {"reg", 'p', 1}, {"bed", 'B', 1}, {"rg-file", 'g', 1}, {"min-k", 'k', 1}, {"max-k", 'K', 1},
  {"trim-lowqual", 'q', 1}, {"min-base-qual", 'C', 1}, {"quality-range", 'Q', 1}, {"min-map-qual", 'b', 1},
  {"max-as-xs-diff", 'Z', 1}, {"tip-len", 'l', 1}, {"cov-thr", 'c', 1}, {"cov-ratio", 'x', 1}, {"low-cov", 'd', 1},
  {"max-avg-cov", 'u', 1}, {"window-size", 'w', 1}, {"padding", 'P', 1}, {"dfs-limit", 'F', 1}, {"max-indel-len", 'T', 1},
  {"max-mismatch", 'M', 1}, {"num-threads", 'X', 1}, {"min-alt-count-tumor", 'a', 1}, {"max-alt-count-normal", 'm', 1},
  {"min-vaf-tumor", 'e', 1}, {"max-vaf-normal", 'i', 1}, {"min-coverage-tumor", 'o', 1}, {"max-coverage-tumor", 'y', 1},
  {"min-coverage-normal", 'z', 1}, {"max-coverage-normal", 'j', 1}, {"min-phred-fisher", 's', 1},
  {"min-phred-fisher-str", 'E', 1}, {"min-strand-bias", 'f', 1}, {"max-unit-length", 'U', 1}, {"min-report-unit", 'N', 1},
  {"min-report-len", 'Y', 1}, {"dist-from-str", 'D', 1}, {"linked-reads", 'J', 0}, {"primary-alignment-only", 'I', 0},
  {"XA-tag-filter", 'O', 0}, {"active-region-off", 'W', 0}, {"verbose", 'v', 0}, {"device", 0, 1}, {"devices", 0, 1}, {"batch-windows", 0, 1}, {"date-line", 0, 1}, {"strict", 0, 0}, {"ranks", 0, 1}, {"rank", 0, 1}, {"rendezvous", 0, 1}, {"help", 'h', 0},
};
int die(const std::string &m) { fprintf(stderr, "lancet_gpu: %s\n", m.c_str()); return 1; }
void usage() {
  fputs("Usage: lancet_gpu --tumor T.bam --normal N.bam --ref ref.fa (--reg chr:start-end | --bed regions.bed) [options] > out.vcf\n"
        "The reference's options with the reference's defaults (lancet --help), except:\n"
        "   --window-size, -w  <int>   : at most 1024 bp (the engine's per-window tables; the reference default is 600)\n"
        "   --max-k, -K        <int>   : at most 127; --min-k at least 3; --max-unit-length at most 8\n"
        "   --num-threads, -X  <int>   : accepted and ignored (windows are batched on the GPU)\n"
        "   --kmer-recovery, --print-graph, --node-str-len, --more-verbose, --print-config-file: not offered\n"
        "Additional options:\n"
        "   --device <n> | --devices a,b,...  : GPU(s) to use; an entry may repeat (two engines on one GPU overlap the upload of a\n"
        "                                       batch with the kernels of the previous one)\n"
        "   --batch-windows <n>               : windows per engine batch [8192: a scan waits for the host side, and smaller batches keep two engines\n"
        "                                       on the GPU busy while the next batch is assembled; 32768 is what the kernels alone like best]\n"
        "   --strict                          : write no VCF when a window exceeded the engine's work space (default: finish,\n"
        "                                       list those windows on stderr, exit code 3)\n"
        "   --ranks <n>                       : n processes, one per GPU; records gathered to rank 0 over RCCL, same VCF for every n\n"
        "   --rank <r> --rendezvous <path>    : this process is rank r of --ranks (a launcher starts the ranks; without --rank the\n"
        "                                       program starts them itself)\n"
        "BAM input: <bam>.bai / <stem>.bai is used when present (one seek per tiled stretch); otherwise the BAM is streamed once.\n"
        "There is no CPU path: a gfx950 device is required.\n", stderr);
}
}  // namespace

int main(int argc, char **argv) {
  std::string tumor, normal, ref, reg, bed, qrange = "!", date_line, devices, rg_file;
  int min_k = 11, max_k = 101, trim_lowqual = 10, min_base_qual = 17, tip_len = 11, cov_thr = 5, low_cov = 1, dfs_limit = 1000000;
  int max_indel_len = 500, max_mismatch = 2, max_unit_length = 4, min_report_unit = 3, min_report_len = 7, dist_from_str = 1;
  int device = 0, batch_windows = 8192, verbose = 0, strict = 0, ranks = 0, rank = -1;
  std::string rendezvous;
  double cov_ratio = 0.01;
  lancet_host_opts ho; lancet_host_opts_default(&ho);
  lancet_filters flt; lancet_filters_default(&flt);
  for (int i = 1; i < argc; ++i) {
    const char *a = argv[i];
    const Opt *o = nullptr;
    if (a[0] == '-' && a[1] == '-') { for (const Opt &c : OPTS) if (strcmp(a + 2, c.lng) == 0) o = &c; }
    else if (a[0] == '-' && a[1] && !a[2]) { for (const Opt &c : OPTS) if (c.shrt && c.shrt == a[1]) o = &c; }
    if (!o) return die(std::string("unknown option ") + a);
    const char *v = "";
    if (o->has_arg) { if (i + 1 >= argc) return die(std::string("option ") + a + " needs a value"); v = argv[++i]; }
    const std::string L = o->lng;
    if (L == "tumor") tumor = v; else if (L == "normal") normal = v; else if (L == "ref") ref = v; else if (L == "reg") reg = v;
    else if (L == "min-k") min_k = atoi(v); else if (L == "max-k") max_k = atoi(v); else if (L == "trim-lowqual") trim_lowqual = atoi(v);
    else if (L == "min-base-qual") min_base_qual = atoi(v); else if (L == "quality-range") qrange = v; else if (L == "min-map-qual") ho.min_map_qual = atoi(v);
    else if (L == "max-as-xs-diff") { /* accepted and without effect, as in the reference: its main() parses -Z but never hands it to the
                                         assemblers (src/Lancet.cc:865-918 lacks the MAX_DELTA_AS_XS line of :496), so the filter always uses 5 */ } else if (L == "tip-len") tip_len = atoi(v); else if (L == "cov-thr") cov_thr = atoi(v);
    else if (L == "cov-ratio") cov_ratio = atof(v); else if (L == "low-cov") low_cov = atoi(v); else if (L == "max-avg-cov") ho.max_avg_cov = atoi(v);
    else if (L == "window-size") ho.window_size = atoi(v); else if (L == "padding") ho.padding = atoi(v); else if (L == "dfs-limit") dfs_limit = atoi(v);
    else if (L == "max-indel-len") max_indel_len = atoi(v); else if (L == "max-mismatch") max_mismatch = atoi(v); else if (L == "num-threads") {}
    else if (L == "min-alt-count-tumor") flt.min_alt_cnt_tumor = atoi(v); else if (L == "max-alt-count-normal") flt.max_alt_cnt_normal = atoi(v);
    else if (L == "min-vaf-tumor") flt.min_vaf_tumor = atof(v); else if (L == "max-vaf-normal") flt.max_vaf_normal = atof(v);
    else if (L == "min-coverage-tumor") flt.min_cov_tumor = atoi(v); else if (L == "max-coverage-tumor") flt.max_cov_tumor = atoi(v);
    else if (L == "min-coverage-normal") flt.min_cov_normal = atoi(v); else if (L == "max-coverage-normal") flt.max_cov_normal = atoi(v);
    else if (L == "min-phred-fisher") flt.min_phred_fisher = atof(v); else if (L == "min-phred-fisher-str") flt.min_phred_fisher_str = atof(v);
    else if (L == "min-strand-bias") flt.min_strand_bias = (int)atof(v); else if (L == "max-unit-length") max_unit_length = atoi(v);
    else if (L == "min-report-unit") min_report_unit = atoi(v); else if (L == "min-report-len") min_report_len = atoi(v); else if (L == "dist-from-str") dist_from_str = atoi(v);
    else if (L == "linked-reads") ho.linked = 1; else if (L == "primary-alignment-only") ho.primary_alignment_only = 1; else if (L == "XA-tag-filter") ho.xa_filter = 1;
    else if (L == "active-region-off") ho.active_region = 0; else if (L == "verbose") verbose = 1; else if (L == "device") device = atoi(v); else if (L == "devices") devices = v; else if (L == "batch-windows") batch_windows = atoi(v);
    else if (L == "date-line") date_line = v; else if (L == "bed") bed = v; else if (L == "rg-file") rg_file = v; else if (L == "strict") strict = 1;
    else if (L == "ranks") ranks = atoi(v); else if (L == "rank") rank = atoi(v); else if (L == "rendezvous") rendezvous = v;
    else if (L == "help") { usage(); return 0; }
  }
  if (tumor.empty() || normal.empty() || ref.empty() || (reg.empty() && bed.empty())) { usage(); return die("--tumor, --normal, --ref and a region (--reg) or BED file (--bed) are required"); }
  if (ranks < 0 || (ranks == 0 && rank >= 0) || (ranks > 0 && rank >= ranks)) return die("--ranks must be >= 1 and --rank one of 0 .. ranks-1");
  if (ranks > 0 && rank < 0) {
    // the launcher: rank r = this program again with --rank r --rendezvous <path>; rank 0 writes the VCF to the stdout they all inherit
    if (rendezvous.empty()) {
      const char *shm = access("/dev/shm", W_OK) == 0 ? "/dev/shm" : "/tmp";
      rendezvous = std::string(shm) + "/lancet_gpu_" + std::to_string((long)getpid()) + ".id";
    }
    unlink(rendezvous.c_str());
    std::vector<pid_t> kids;
    for (int r = 0; r < ranks; ++r) {
      const pid_t pid = fork();
      if (pid < 0) return die("fork failed");
      if (pid == 0) {
        std::vector<std::string> av(argv, argv + argc);
        av.push_back("--rank"); av.push_back(std::to_string(r)); av.push_back("--rendezvous"); av.push_back(rendezvous);
        std::vector<char *> cv; for (std::string &x : av) cv.push_back(&x[0]); cv.push_back(nullptr);
        execv("/proc/self/exe", cv.data());
        _exit(127);
      }
      kids.push_back(pid);
    }
    // a rank that fails (exit code other than 0 / 3: bad input, no device, a communicator that does not come up) takes the others with it --
    // they would wait for it at the rendezvous or in the gather
    int worst = 0; size_t left = kids.size();
    while (left > 0) {
      int st = 0;
      const pid_t k = waitpid(-1, &st, 0);
      if (k < 0) { worst = worst ? worst : 1; break; }
      if (std::find(kids.begin(), kids.end(), k) == kids.end()) continue;
      --left;
      const int c = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
      if (c && (!worst || c == 1 || worst == 3)) worst = c;
      if (c != 0 && c != 3) for (pid_t o : kids) if (o != k) kill(o, SIGTERM);
    }
    unlink(rendezvous.c_str());
    return worst;
  }
  if (ranks > 0 && rendezvous.empty()) return die("--rank needs --rendezvous <path> (the same path for every rank)");
  const int qoff = qrange.empty() ? 33 : (unsigned char)qrange[0];
  lancet_params P; lancet_params_default(&P);
  P.min_k = min_k; P.max_k = max_k; P.max_tip_len = tip_len; P.cov_threshold = cov_thr; P.low_cov_threshold = low_cov; P.dfs_limit = dfs_limit;
  P.max_indel_len = max_indel_len; P.max_mismatch = max_mismatch; P.min_qual_trim = trim_lowqual + qoff; P.min_qual_call = min_base_qual + qoff;
  P.max_unit_len = max_unit_length; P.min_report_units = min_report_unit; P.min_report_len = min_report_len; P.dist_from_str = dist_from_str;
  P.lr_mode = ho.linked; P.min_cov_ratio = cov_ratio;
  ho.max_k = max_k; ho.min_evidence = flt.min_alt_cnt_tumor; ho.min_qual_call = min_base_qual + qoff;

  const bool timing = getenv("LANCET_HOST_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_batch = 0, t_engine = 0, t_vdb = 0; float t_kernel = 0;
  const double t_start = now();
  // one engine per entry of --devices (default: --device); an entry may repeat a device (two engines on one GPU overlap the
  // upload of a batch with the kernels of the previous one).  Batches go to the engines in turn; there is no CPU path:
  // without a GPU engine creation fails.
  std::vector<int> devs;
  if (devices.empty()) devs.push_back(ranks > 0 ? rank : device);
  else { size_t p0 = 0; while (p0 <= devices.size()) { size_t q = devices.find(',', p0); if (q == std::string::npos) q = devices.size(); if (q > p0) devs.push_back(atoi(devices.substr(p0, q - p0).c_str())); p0 = q + 1; } }
  if (devs.empty()) return die("--devices is empty");
  if (ranks > 0) { const int d = devs[(size_t)rank % devs.size()]; devs.assign(1, d); }     // one GPU per rank
  // The communicator comes up on a thread of its own while this one decodes, tiles and assembles (loading librccl and ncclCommInitRank take
  // about a second and a half, and nothing needs the communicator before the gather at the end); a rank that cannot join reports it there --
  // the others then fail in ncclCommInitRank or time out at the rendezvous.
  const bool ranked = ranks > 0;
  lancet_comm *comm = nullptr;
  char cerr[512] = "";
  std::future<lancet_comm *> comm_fut;
  if (ranked) {
    if (ranks > 1 && !getenv("LANCET_HOST_LAZY")) setenv("LANCET_HOST_LAZY", "1", 0);     // a rank loads what ITS windows select, once (lancet_host_load_range; needs both .bai, else a notice and everything is loaded)
    const int cdev = devs[0];                                                                // (the environment is settled before the thread starts: it reads it)
    comm_fut = std::async(std::launch::async, [=, &cerr]() { return lancet_comm_create(rank, ranks, cdev, rendezvous.c_str(), 300.0, cerr, sizeof cerr); });
  }
  std::vector<lancet_engine *> engs;
  for (int d : devs) {
    lancet_engine *e = nullptr;
    const int rc = lancet_engine_create(&P, d, &e);
    if (rc == LANCET_E_UNSUPPORTED) return die("unsupported parameters: --max-k must be <= 127, --min-k >= 3, --max-unit-length <= 8");
    if (rc != LANCET_OK) return die(std::string("cannot create the MI355X engine on device ") + std::to_string(d) + " (code " + std::to_string(rc) + "): " + (e ? lancet_engine_last_error(e) : "no gfx950 device / HIP runtime"));
    if (verbose) lancet_engine_set_trace(e, 1u << 17);            // -v: the reference's per-window stage trace, to stderr
    engs.push_back(e);
  }
  char err[512] = "";
  lancet_host *H = lancet_host_open(tumor.c_str(), normal.c_str(), ref.c_str(), err, sizeof err);
  if (H && !rg_file.empty() && lancet_host_set_rg_file(H, rg_file.c_str()) != LANCET_OK) return die(lancet_host_last_error(H));
  if (!H) return die(err);
  const double t_tile0 = now();
  const char *regs[1] = {reg.c_str()};
  const int nwin = lancet_host_tile_regions(H, bed.empty() ? nullptr : bed.c_str(), regs, reg.empty() ? 0 : 1, &ho);
  const double t_tile = now() - t_tile0;
  if (nwin < 0) return die(lancet_host_last_error(H));
  if (ho.active_region && !lancet_host_first_has_md(H, 1) && !lancet_host_first_has_md(H, 0)) {      // reference src/Lancet.cc:817-825
    fputs("\n--------WARNING--------\nThe MD tag is required to select the active regions, but is missing from the alignments in the BAM(s) file(s).\n"
          "To avoid unpredictable behavior, the active region module has been automatically turned off (--active-region-off)\n-----------------------\n\n", stderr);
    ho.active_region = 0;
  }
  lancet_vdb *db = lancet_vdb_create(&flt);
  int n_chr = 0;
  const char *const *chr_names = lancet_host_chroms(H, &n_chr);
  const int step = batch_windows > 0 ? batch_windows : 1;
  const int nchunks = (nwin + step - 1) / step;
  // a scan of several batches on one GPU: a second engine on the same device, so that the upload / trim of batch i+1 and the tail
  // of its kernels overlap with batch i (what `--devices 0,0` asks for explicitly; bench.py --in-flight 2 measures it)
  if (devices.empty() && nchunks > 1 && engs.size() == 1) {
    lancet_engine *e2 = nullptr;
    if (lancet_engine_create(&P, devs[0], &e2) == LANCET_OK) { if (verbose) lancet_engine_set_trace(e2, 1u << 17); engs.push_back(e2); }
  }
  struct Job { bool have = false; std::string trace; std::vector<lancet_variant> v; std::string blob; std::vector<lancet_variant_lr> lr; std::vector<uint32_t> bx; std::vector<std::string> bxn;
               std::vector<int64_t> widx; /* --ranks: tiled (= global) number of the batch's window w */ };
  std::vector<std::vector<uint8_t>> parts;      // --ranks: this rank's batches, packed (lancet_records_pack), in batch order
  struct Slot { lancet_engine *e = nullptr; std::future<int> fut; int chunk = -1; int nk = 0; long base = 0; std::vector<int32_t> kept; std::vector<std::string> bxn; };
  std::vector<Job> jobs((size_t)nchunks);
  std::vector<Slot> slots(engs.size());
  for (size_t k = 0; k < engs.size(); ++k) { slots[k].e = engs[k]; slots[k].kept.resize((size_t)step); }
  long done = 0;
  std::map<int, lancet_engine *> last_on_dev;      // device -> the engine submitted last on it
  int next_add = 0;
  std::string fail;
  std::vector<std::string> overflowed;
  // results of a finished run are copied out (the engine's buffers live until its next upload) and added to the
  // VariantDB strictly in chunk order: addVar order is window order
  auto finish = [&](Slot &sl) -> bool {
    const double t0 = now();
    const int rc = sl.fut.get();
    t_engine += now() - t0;
    if (rc != LANCET_OK) { fail = std::string("engine: ") + lancet_engine_last_error(sl.e); return false; }
    { float tm[2] = {0, 0}; lancet_engine_last_timing(sl.e, tm); t_kernel += tm[0]; }
    const lancet_variant *v; uint32_t nv, blen; const char *blob; const lancet_window_stats *st;
    if (lancet_engine_results(sl.e, &v, &nv, &blob, &blen, &st) != LANCET_OK) { fail = std::string("engine: ") + lancet_engine_last_error(sl.e); return false; }
    for (int w = 0; w < sl.nk; ++w) if (st[w].status < 0) {
      if (strict) { fail = std::string("work-space overflow in window ") + lancet_host_window_hdr(H, sl.kept[(size_t)w]) + ": results withheld (--strict)"; return false; }
      overflowed.push_back(lancet_host_window_hdr(H, sl.kept[(size_t)w]));      // the window emitted nothing (the engine drops a window's records when it overflows)
    }
    Job &j = jobs[(size_t)sl.chunk];
    j.v.assign(v, v + nv); j.blob.assign(blob, blen);
    if (ranked) j.widx.assign(sl.kept.begin(), sl.kept.begin() + sl.nk);
    if (ho.linked) {
      const lancet_variant_lr *lr; const uint32_t *bxb; uint32_t bxl;
      if (lancet_engine_results_lr(sl.e, &lr, &bxb, &bxl) != LANCET_OK) { fail = std::string("engine: ") + lancet_engine_last_error(sl.e); return false; }
      j.lr.assign(lr, lr + nv); j.bx.assign(bxb, bxb + bxl); j.bxn.swap(sl.bxn);
    }
    if (verbose) {
      const uint32_t *elen, *evt; uint32_t wpw = 0;
      if (lancet_engine_trace(sl.e, &elen, &evt, &wpw) == LANCET_OK && wpw) for (int w = 0; w < sl.nk; ++w) {
        const int tw = sl.kept[(size_t)w];
        const char *hdr = lancet_host_window_hdr(H, tw);
        int32_t start = 0, end = 0; lancet_host_window_span(H, tw, &start, &end);
        char *t = lancet_trace_format(evt + (size_t)w * wpw, elen[w], (int32_t)(sl.base + w + 1), hdr, chr_names[lancet_host_window_chrom(H, tw)], start, end, dfs_limit);
        if (t) { j.trace += t; lancet_free(t); }
      }
    }
    j.have = true; sl.chunk = -1;
    return true;
  };
  auto flush = [&]() -> bool {
    const double t0 = now();
    while (next_add < nchunks && jobs[(size_t)next_add].have) {
      Job &j = jobs[(size_t)next_add];
      if (!j.trace.empty()) fputs(j.trace.c_str(), stderr);
      int arc = LANCET_OK;
      if (ranked) {                                  // the records travel: keyed, reduced, window numbers of the whole tiling
        if (!j.v.empty()) {
          std::vector<const char *> names; for (auto &n : j.bxn) names.push_back(n.c_str());
          uint8_t *pb = nullptr; size_t pl = 0;
          arc = lancet_records_pack(j.v.data(), (uint32_t)j.v.size(), j.blob.data(), (uint32_t)j.blob.size(), ho.linked ? j.lr.data() : nullptr, ho.linked ? j.bx.data() : nullptr,
                                    ho.linked ? names.data() : nullptr, (uint32_t)names.size(), chr_names, n_chr, j.widx.data(), (uint32_t)j.widx.size(), 1, &pb, &pl);
          if (arc == LANCET_OK) { parts.emplace_back(pb, pb + pl); lancet_free(pb); }
        }
      } else
      if (!j.v.empty()) {
        if (ho.linked) {
          std::vector<const char *> names; for (auto &n : j.bxn) names.push_back(n.c_str());
          arc = lancet_vdb_add_lr(db, j.v.data(), j.lr.data(), (uint32_t)j.v.size(), j.blob.c_str(), j.bx.data(), names.data(), (uint32_t)names.size(), chr_names, (uint32_t)n_chr);
        } else arc = lancet_vdb_add(db, j.v.data(), (uint32_t)j.v.size(), j.blob.c_str(), chr_names, (uint32_t)n_chr);
      }
      if (arc != LANCET_OK) { fail = "VariantDB rejected the records"; return false; }
      j = Job(); j.have = true;
      ++next_add;
    }
    t_vdb += now() - t0;
    return true;
  };
  // The host threads that assemble a batch also trim and pack its reads (lancet_host_batch_packed -> lancet_engine_upload_packed: 3 bits per
  // base handed over, one pass over the reads instead of two); LANCET_GPU_ASCII=1, or trim + pack on the device (LANCET_PREP=device), keep
  // the ASCII hand-over.
  const bool packed = getenv("LANCET_GPU_ASCII") == nullptr && !(getenv("LANCET_PREP") && strcmp(getenv("LANCET_PREP"), "device") == 0);
  // --ranks: rank r owns one contiguous run of batches -- in the table's order (header strings) that is a run of contigs, whose alignments
  // the rank loads once; the records then reach rank 0 in window order, rank by rank
  const int c_lo = ranked ? (int)((long)nchunks * rank / ranks) : 0, c_hi = ranked ? (int)((long)nchunks * (rank + 1) / ranks) : nchunks;
  if (ranked && c_hi > c_lo && lancet_host_load_range(H, c_lo * step, c_hi * step < nwin ? c_hi * step : nwin, &ho) != LANCET_OK) return die(lancet_host_last_error(H));
  for (int c = 0; c < nchunks; ++c) {
    const int lo = c * step, hi = lo + step < nwin ? lo + step : nwin;
    if (ranked && (c < c_lo || c >= c_hi)) { jobs[(size_t)c].have = true; if (!flush()) return die(fail); continue; }      // another rank's batch
    Slot &sl = slots[(size_t)c % slots.size()];
    lancet_window_batch B; lancet_packed_reads PK; int32_t nk = 0;
    memset(&PK, 0, sizeof(PK));
    double t0 = now();
    if (sl.fut.valid() && !finish(sl)) return die(fail);          // its kept[] / engine must be free before they are reused
    if ((packed ? lancet_host_batch_packed(H, lo, hi, &ho, &P, &B, &PK, sl.kept.data(), &nk) : lancet_host_batch(H, lo, hi, &ho, &B, sl.kept.data(), &nk)) != LANCET_OK)
      return die(lancet_host_last_error(H));                      // (overlaps the other engines' kernels)
    t_batch += now() - t0;
    if (nk == 0) { jobs[(size_t)c].have = true; if (!flush()) return die(fail); continue; }
    t0 = now();
    if ((packed ? lancet_engine_upload_packed(sl.e, &B, &PK) : lancet_engine_upload(sl.e, &B)) != LANCET_OK) return die(std::string("engine: ") + lancet_engine_last_error(sl.e));
    t_engine += now() - t0;
    sl.chunk = c; sl.nk = nk; sl.base = done; done += nk; sl.bxn.clear();
    if (ho.linked) { uint32_t nbx = 0; const char *const *bxn = lancet_host_bx_names(H, &nbx); for (uint32_t i = 0; i < nbx; ++i) sl.bxn.emplace_back(bxn[i]); }
    lancet_engine *e = sl.e;
    // The launch happens HERE, on the one thread that submits (a few hundred microseconds: everything is asynchronous), behind the kernels
    // of the last engine submitted on the same GPU -- the two batches' kernels run back to back, not side by side; its done-event is
    // recorded by then because its own submit has returned.  Only the wait (re-run tier, read-back) goes to another thread.
    const int dev = devs.size() > 1 ? devs[(size_t)c % slots.size() % devs.size()] : devs[0];
    lancet_engine *prev = last_on_dev.count(dev) ? last_on_dev[dev] : nullptr;
    t0 = now();
    if (lancet_engine_submit_after(e, prev) != LANCET_OK) return die(std::string("engine: ") + lancet_engine_last_error(e));
    t_engine += now() - t0;
    last_on_dev[dev] = e;
    sl.fut = std::async(std::launch::async, [e]() { return lancet_engine_wait(e); });
    if (!flush()) return die(fail);
  }
  for (Slot &sl : slots) if (sl.fut.valid() && !finish(sl)) return die(fail);
  if (!flush()) return die(fail);
  double t_gather = 0;
  if (ranked) {
    comm = comm_fut.get();
    if (!comm) return die(std::string("rank ") + std::to_string(rank) + ": cannot join the communicator: " + cerr);
    // one payload per rank: u64 number of parts, their lengths, the parts; rank 0 replays every rank's parts into the database
    const double tg0 = now();
    std::vector<uint8_t> payload(8 * (1 + parts.size()));
    { uint64_t x = parts.size(); memcpy(payload.data(), &x, 8); for (size_t i = 0; i < parts.size(); ++i) { x = parts[i].size(); memcpy(payload.data() + 8 * (1 + i), &x, 8); } }
    for (auto &pt : parts) payload.insert(payload.end(), pt.begin(), pt.end());
    uint8_t *all = nullptr; std::vector<size_t> lens((size_t)ranks, 0);
    if (lancet_comm_gather(comm, payload.data(), payload.size(), &all, lens.data()) != LANCET_OK) return die(std::string("rank ") + std::to_string(rank) + ": gather: " + lancet_comm_last_error(comm));
    uint32_t added = 0;
    if (rank == 0) {
      std::vector<const uint8_t *> pp; std::vector<size_t> pl;
      size_t o = 0;
      for (int r = 0; r < ranks; ++r) {
        const uint8_t *b = all + o; const size_t len = lens[(size_t)r]; o += len;
        if (len < 8) return die("gather: a rank sent a damaged payload");
        uint64_t np = 0; memcpy(&np, b, 8);
        if (8 * (1 + np) > len) return die("gather: a rank sent a damaged payload");
        size_t q = 8 * (1 + (size_t)np);
        for (uint64_t i = 0; i < np; ++i) { uint64_t x; memcpy(&x, b + 8 * (1 + i), 8); if (q + x > len) return die("gather: a rank sent a damaged payload"); pp.push_back(b + q); pl.push_back((size_t)x); q += (size_t)x; }
      }
      if (lancet_records_merge(db, pp.data(), pl.data(), (int)pp.size(), &added) != LANCET_OK) return die("VariantDB rejected the gathered records");
    }
    t_gather = now() - tg0;
    fprintf(stderr, "[lancet_gpu] rank %d of %d on GPU %d (%s): %ld windows assembled, %zu batches, %zu bytes to rank 0%s\n", rank, ranks, devs[0], lancet_comm_transport(comm), done, parts.size(),
            payload.size(), rank == 0 ? (std::string("; ") + std::to_string(added) + " records replayed").c_str() : "");
    if (all) lancet_free(all);
    lancet_comm_destroy(comm);
  }
  if (ranks > 0 && rank != 0) {
    lancet_vdb_destroy(db); lancet_host_close(H); for (lancet_engine *e : engs) lancet_engine_destroy(e);
    if (!overflowed.empty()) {
      fprintf(stderr, "lancet_gpu: rank %d: %zu window(s) exceeded the engine's work space and contributed NO variants:\n", rank, overflowed.size());
      for (const std::string &w : overflowed) fprintf(stderr, "lancet_gpu:   %s\n", w.c_str());
      return 3;
    }
    return 0;
  }
  std::string cmdline = "lancet";
  for (int i = 1; i < argc; ++i) {
    if (strcmp(argv[i], "--date-line") == 0 || strcmp(argv[i], "--rank") == 0 || strcmp(argv[i], "--rendezvous") == 0) { ++i; continue; }     // (what names the run / the process, not the job)
    cmdline += " "; cmdline += argv[i];
  }
  if (date_line.empty()) { time_t t = time(nullptr); date_line = ctime(&t); }
  else if (date_line.back() != '\n') date_line += "\n";
  char *vcf = lancet_vdb_vcf(db, nullptr, cmdline.c_str(), ref.c_str(), date_line.c_str(), lancet_host_sample(H, 0), lancet_host_sample(H, 1));
  if (!vcf) return die("VCF rendering failed");
  fputs(vcf, stdout);
  fprintf(stderr, "[lancet_gpu] %d windows tiled, %ld assembled on %zu engine(s), first GPU %d, %u variants\n", nwin, done, engs.size(), devs[0], lancet_vdb_size(db));
  if (timing) fprintf(stderr, "[lancet_gpu] wall %.3f s: input decode + tiling %.3f, window filters + batches %.3f, engine (upload + kernels + results) %.3f (kernels %.3f), VariantDB %.3f\n",
                      now() - t_start, t_tile, t_batch, t_engine, t_kernel / 1000.0, t_vdb + t_gather);
  lancet_free(vcf);
  lancet_vdb_destroy(db); lancet_host_close(H); for (lancet_engine *e : engs) lancet_engine_destroy(e);
  if (!overflowed.empty()) {
    fprintf(stderr, "lancet_gpu: %zu window(s) exceeded the engine's work space (tier-2 limits, DESIGN.md section 9) and contributed NO variants:\n", overflowed.size());
    for (const std::string &w : overflowed) fprintf(stderr, "lancet_gpu:   %s\n", w.c_str());
    return 3;
  }
  return 0;
}
