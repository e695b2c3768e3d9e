/* lancet_engine.h -- C-ABI of the MI355X micro-assembly engine (drop-in for Lancet's per-window hot path).
 *
 * The reference has no plugin / FFI interface.  The seam this library sits behind is the
 * C++ member call
 *      numreads = processGraph(g, graphref, minK, maxK);          reference src/Microassembler.cc:837
 *                                                        (declared reference src/Microassembler.hh:224)
 * whose inputs are, at that point, fully materialised:
 *      g.readid2info  -- reads filled by Graph_t::addAlignment     reference src/Graph.cc:487-501
 *                        (fields: reference src/ReadInfo.hh:44-67)
 *      Ref_t          -- window reference string + coordinates     reference src/Ref.hh:55-97
 *      knobs          -- Graph_t setters                           reference src/Microassembler.cc:726-753
 * and whose only output is the stream of
 *      vDB->addVar(Variant_t(...))                                 reference src/Graph.cc:1184-1188
 *                                                (ctor signature: reference src/Variant.hh:106-112)
 *
 * One `lancet_engine_run` == that call, for a whole batch of independent windows at once.
 * Plain pointers and sizes only; no C++ / torch types.  Return 0 on success, <0 on error (never aborts).
 * Outputs are owned by the engine and stay valid until the next upload/run/destroy on that engine.
 * One engine per (host thread, GPU); an engine is not thread-safe.
 */
#ifndef LANCET_ENGINE_H
#define LANCET_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sample labels / strands as the reference encodes them */
#define LANCET_TMR 4 /* reference src/Ref.hh:36 */
#define LANCET_NML 5 /* reference src/Ref.hh:37 */
#define LANCET_FWD 1 /* reference src/ReadInfo.hh:30 */
#define LANCET_REV 2 /* reference src/ReadInfo.hh:31 */

/* Knobs copied into Graph_t before processGraph (reference src/Microassembler.cc:726-753, defaults
 * reference src/Lancet.hh:33-81).  lancet_params_default() fills the reference defaults. */
typedef struct lancet_params {
  int32_t min_k;             /* minK 11                      */
  int32_t max_k;             /* maxK 101                     */
  int32_t max_tip_len;       /* MAX_TIP_LEN 11               */
  int32_t cov_threshold;     /* COV_THRESHOLD 5              */
  int32_t low_cov_threshold; /* LOW_COV_THRESHOLD 1          */
  int32_t dfs_limit;         /* DFS_LIMIT 1000000            */
  int32_t max_indel_len;     /* MAX_INDEL_LEN 500            */
  int32_t max_mismatch;      /* MAX_MISMATCH 2               */
  int32_t min_qual_trim;     /* MIN_QV_TRIM 10 + '!'         */
  int32_t min_qual_call;     /* MIN_QV_CALL 17 + '!'         */
  int32_t max_unit_len;      /* MAX_UNIT_LEN 4   (STR)       */
  int32_t min_report_units;  /* MIN_REPORT_UNITS 3           */
  int32_t min_report_len;    /* MIN_REPORT_LEN 7             */
  int32_t dist_from_str;     /* DIST_FROM_STR 1              */
  int32_t lr_mode;           /* --linked-reads (LR_MODE false): barcode/haplotype aware coverage   */
  int32_t reserved;
  double  min_cov_ratio;     /* MIN_COV_RATIO 0.01           */
} lancet_params;

/* A batch of independent windows, SoA, caller-owned, read-only during the call.
 * Reads of window w are read_begin[w] .. read_begin[w+1]-1, in the order Graph_t::readid2info is filled:
 * tumor reads in BAM order, then normal reads in BAM order (reference src/Microassembler.cc:833-834).
 * Bases / qualities are ASCII exactly as BamAlignment::QueryBases / Qualities hold them (phred+33). */
typedef struct lancet_window_batch {
  int32_t         n_windows;
  const int32_t  *chr_id;      /* [n_windows] caller's chromosome id, echoed in lancet_variant        */
  const int32_t  *ref_start;   /* [n_windows] Ref_t::refstart (1-based)                               */
  const uint32_t *ref_off;     /* [n_windows+1] offsets into ref_bases                                 */
  const char     *ref_bases;   /* Ref_t::rawseq of every window, upper-case ACGTN, concatenated        */
  const uint32_t *read_begin;  /* [n_windows+1]                                                        */
  const uint32_t *seq_off;     /* [n_reads+1] offsets into seq / qual                                  */
  const char     *seq;         /* ReadInfo_t::seq_m                                                    */
  const char     *qual;        /* ReadInfo_t::qv_m                                                     */
  const uint8_t  *label;       /* [n_reads] LANCET_TMR | LANCET_NML   (ReadInfo_t::label_m); label / strand / mate with
                                   any other value are refused at upload (LANCET_E_ARG)                  */
  const uint8_t  *strand;      /* [n_reads] LANCET_FWD | LANCET_REV   (ReadInfo_t::strand)             */
  const uint8_t  *mate;        /* [n_reads] 0 | 1 | 2                 (ReadInfo_t::mate_order_m)       */
  const uint8_t  *mapped;      /* [n_reads] 1 = CODE_MAPPED, 0 = CODE_BASTARD (ReadInfo_t::code_m)     */
  const uint32_t *name_rank;   /* [n_reads] dense rank of ReadInfo_t::readname_m among the window's
                                  read names under std::string operator< (equal names = equal rank)   */
  /* --linked-reads only (lancet_params::lr_mode != 0); may be NULL otherwise.
   * reference: ReadInfo_t::BX / HP (src/ReadInfo.hh:60-61), filled by extractReads (src/Microassembler.cc:581-593) */
  const uint32_t *bx_rank;     /* [n_reads] dense rank of the BX:Z barcode among the batch's barcodes under
                                  std::string operator< ; LANCET_NO_BX for a read without barcode ("null")   */
  const uint8_t  *hp;          /* [n_reads] HP:i haplotype 0 (unassigned / absent) | 1 | 2                    */
} lancet_window_batch;
#define LANCET_NO_BX 0xFFFFFFFFu

/* One addVar(Variant_t(...)) call, arguments as passed at reference src/Graph.cc:1184-1188. */
typedef struct lancet_variant {
  int32_t  window;          /* index of the window in the batch                                      */
  int32_t  seq_in_window;   /* emission order within the window (0,1,2,...)                          */
  int32_t  chr_id;
  int32_t  pos;             /* transcript.pos - 1  (Variant_t ctor arg pos_)                          */
  uint8_t  code;            /* 'x' snv, '^' ins, 'v' del, 'c' complex                                 */
  uint8_t  prev_bp_ref;
  uint8_t  prev_bp_alt;
  uint8_t  reserved;
  uint16_t kmer;            /* K                                                                      */
  uint16_t cov[8];          /* RCN fwd,rev  RCT fwd,rev  ACN fwd,rev  ACT fwd,rev                     */
  uint16_t reserved2;
  uint32_t ref_off, ref_len;   /* transcript.ref  (may contain '-') in the blob                       */
  uint32_t alt_off, alt_len;   /* transcript.qry  (may contain '-') in the blob                       */
  uint32_t str_off, str_len;   /* STR annotation "<len><motif>" or empty                              */
} lancet_variant;

/* Per-window outcome. */
/* Linked-read annotations of variant i (same index as the lancet_variant array); all zero / empty unless lr_mode.
 * Replaces the HPRN/HPRT/HPAN/HPAT and bxset_* arguments of the Variant_t constructor (reference src/Graph.cc:1166-1188). */
typedef struct lancet_variant_lr {
  uint16_t hp[12];          /* HPRN, HPRT, HPAN, HPAT ; each {hp1, hp2, hp0}                                  */
  uint32_t bx_off[4];       /* bxset_ref_N, bxset_ref_T, bxset_alt_N, bxset_alt_T : offset (in u32) into bx_blob */
  uint32_t bx_len[4];       /* number of barcodes; they are bx_rank values in increasing order (== the ';'-joined
                               std::set<string> order of the reference); len 0 prints as "."                   */
  uint32_t reserved[2];
} lancet_variant_lr;

typedef struct lancet_window_stats {
  int32_t  status;        /* LANCET_W_* below                                                        */
  int32_t  final_k;       /* k of the last graph build (0 if none)                                   */
  int32_t  n_builds;      /* k-attempts that reached buildgraph                                       */
  int32_t  n_variants;
  uint64_t n_kmers;       /* sum over builds of the loadSequence trip count         */
  uint32_t max_nodes;     /* largest node table over the builds                                       */
  uint32_t sum_nodes;     /* sum over the builds of their node tables' sizes (40 B per node of every build) */
} lancet_window_stats;

#define LANCET_W_OK            0   /* processed (possibly with no variants)                          */
#define LANCET_W_NO_READS      1   /* countMappedReads()<=0 -> returned early (Microassembler.cc:83)  */
#define LANCET_W_K_EXHAUSTED   2   /* every k up to max_k was rejected                                */
#define LANCET_W_OVERFLOW     -1   /* a device work-space limit was hit; results for this window are
                                      NOT valid (reported loudly, never silently wrong).  Limits per window: a
                                      reference of at most 1024 bases, reads of at most 1023 + k bases (per-position
                                      counts wrap at 65 536 as the reference's unsigned short counters do), 2^21 reads
                                      (more than 65 535 reads: assembled by the re-run tier), 2^20 distinct k-mers
                                      unless LANCET_MAX_NODES says otherwise                            */

/* error codes */
#define LANCET_OK              0
#define LANCET_E_ARG          -1
#define LANCET_E_NO_DEVICE    -2
#define LANCET_E_HIP          -3
#define LANCET_E_UNSUPPORTED  -4
#define LANCET_E_OOM          -5
#define LANCET_E_STATE        -6

typedef struct lancet_engine lancet_engine;

void lancet_params_default(lancet_params *p);

/* device >= 0 : HIP device ordinal.  There is no CPU backend: creation fails with LANCET_E_NO_DEVICE
 * when no gfx950-compatible device is present.  LANCET_E_UNSUPPORTED: max_k > 127, min_k < 3 or max_unit_len > 8.  Even k is
 * supported (k-mers that are their own reverse complement: CanonicalMer_t::set ties -> R, reference src/Mer.hh:57-71). */
int  lancet_engine_create(const lancet_params *p, int device, lancet_engine **out);
void lancet_engine_destroy(lancet_engine *e);
const char *lancet_engine_last_error(const lancet_engine *e);

/* Copies a batch to HBM (packs bases to 2 bit, sizes the per-window work space).  After this returns the
 * batch is device-resident and the caller's buffers are no longer referenced. */
int lancet_engine_upload(lancet_engine *e, const lancet_window_batch *b);

/* The same with the reads already trimmed and packed by the caller -- what lancet_engine_upload does itself, on host threads, from the
 * ASCII arrays (Graph_t::trim, reference src/Graph.cc:355-384, then 2 bit per base and one bit per base for `quality >= MIN_QUAL_CALL`):
 * 3 bits per base handed over instead of 16, and no second pass over the reads for a caller that assembles its batch from alignments anyway
 * (lancet_host_batch_packed does).  `b` as for lancet_engine_upload, but seq / qual / label / strand / mate / mapped are not looked at
 * (seq_off still gives the UNTRIMMED length of every read).  Read r owns (len + 15) / 16 words of `bases` from base_woff[r] and
 * (len + 31) / 32 words of `good` from good_woff[r], len its untrimmed length; both offset arrays have n_reads + 1 entries, `bases` has
 * 4 readable words past the last read's, `good` one.  Every read must have been packed by lancet_pack_read with the parameters this
 * engine was created with: the trim and the mask depend on min_qual_trim / min_qual_call, which the producer records in the struct
 * (lancet_host_batch_packed does; a caller of lancet_pack_read copies them from the lancet_params it packed with) -- an engine created
 * with other thresholds refuses the batch (LANCET_E_ARG) instead of building graphs from reads trimmed for another run. */
typedef struct lancet_packed_reads {
  uint32_t struct_size;        /* sizeof(lancet_packed_reads) as the CALLER's header has it: the first field, so that a caller built against
                                  another layout of this struct is refused (LANCET_E_ARG) instead of being read past its end.  The producers
                                  in this library (lancet_host_batch_packed) fill it in */
  uint32_t reserved;           /* 0 */
  const uint32_t *rinfo;       /* [n_reads] trimmed length and the read's flags, as lancet_pack_read leaves them */
  const uint32_t *base_woff;   /* [n_reads + 1] */
  const uint32_t *good_woff;   /* [n_reads + 1] */
  const uint32_t *bases;       /* 2 bit per base of the trimmed read, first base in the low bits */
  const uint32_t *good;        /* 1 bit per base of the trimmed read */
  int32_t min_qual_trim;       /* the thresholds the reads were trimmed / masked with (lancet_params of the packing side) */
  int32_t min_qual_call;
  /* Reads stored once (round 5).  Windows every 100 bases of 600 put an alignment into ~6 windows (reference src/Microassembler.cc:436-655
   * fetches it for each of them).  read_index != NULL: the five arrays above describe n_distinct DISTINCT reads (offsets: n_distinct + 1
   * entries) and read_index[r], r < read_begin[n_windows], says which of them read r of the batch is -- the words of an alignment are packed,
   * staged and copied to the device once per batch instead of once per window that holds it.  label / strand / mate / mapped are the
   * alignment's (in rinfo); only the name rank is per window (lancet_window_batch::name_rank).  NULL: one entry per read of the batch. */
  const uint32_t *read_index;
  uint32_t n_distinct;
  uint32_t reserved2;          /* 0 */
} lancet_packed_reads;
int lancet_engine_upload_packed(lancet_engine *e, const lancet_window_batch *b, const lancet_packed_reads *p);
/* One read, trimmed and packed: rinfo[0], (len + 15) / 16 words of bases and (len + 31) / 32 words of good are written (zero past the
 * trimmed length).  label / strand / mate / mapped as in lancet_window_batch. */
void lancet_pack_read(const lancet_params *P, const char *seq, const char *qual, int len, uint8_t label, uint8_t strand, uint8_t mate, uint8_t mapped,
                      uint32_t *rinfo, uint32_t *bases, uint32_t *good);

/* Runs the whole hot path (self-tuning k loop included) for every uploaded window.  Blocking. */
int lancet_engine_run(lancet_engine *e);

/* upload + run in one call, the exact equivalent of the reference seam. */
int lancet_engine_process(lancet_engine *e, const lancet_window_batch *b);

/* lancet_engine_run in two halves: submit launches the kernels of the uploaded batch on the engine's stream and returns at once,
 * wait blocks until they are done (and re-runs what overflowed the small work space) and reads the results back.  One batch in
 * flight per engine; two engines on the same device, submitted in turn, overlap the tail of one batch with the bulk of the next. */
int lancet_engine_submit(lancet_engine *e);
int lancet_engine_wait(lancet_engine *e);
/* submit for the second of two engines that take turns on one device: e's kernels start when `prev`'s kernels are through (a batch's
 * kernels are sized for the whole device; side by side with another batch's they only slow each other down), while the host side of
 * the two batches -- upload, launch, read-back -- overlaps the other's kernels.  prev == NULL: lancet_engine_submit.
 * "Through" is, unless LANCET_GATE=0 is in the environment of the process that made the engines (and never in lr_mode): prev's window
 * kernel has taken the last quarter of its windows in hand -- e's build kernel is dispatched onto the CUs as that launch leaves them
 * (profiles/r6h_gate.txt).  Results do not depend on it. */
int lancet_engine_submit_after(lancet_engine *e, lancet_engine *prev);

/* Results of the last run: variants ordered by (window, seq_in_window) so that the caller can replay
 * addVar in reference order. */
int lancet_engine_results(lancet_engine *e, const lancet_variant **variants, uint32_t *n_variants,
                          const char **blob, uint32_t *blob_len, const lancet_window_stats **stats);

/* Timing of the last run as measured with HIP events on the engine's stream (milliseconds):
 * out[0] = all kernels of the run, out[1] = the assembly kernel only. */
/* Linked-read annotations of the last run (lr_mode engines; otherwise *lr = NULL): lr[i] belongs to variants[i]. */
int lancet_engine_results_lr(lancet_engine *e, const lancet_variant_lr **lr, const uint32_t **bx_blob, uint32_t *bx_blob_len);
int lancet_engine_last_timing(lancet_engine *e, float out[2]);
/* The kernels one run launches, back to back on the engine's stream: lancet_engine_kernel_name(i) for i = 0, 1, ... (NULL past
 * the last) and the HIP-event duration of each in the last run (milliseconds; returns how many were written, <= cap). */
const char *lancet_engine_kernel_name(int i);
int lancet_engine_kernel_times(lancet_engine *e, float *ms, int cap);
/* Windows of the last run whose first graph was assembled by the LDS build kernel (the others took the general build phases). */
int lancet_engine_prebuilt_count(lancet_engine *e);
/* Graphs the build kernel built AHEAD in the last run -- the next k of a window whose graph at the current k holds a k-mer twice
 * in one read and will be rejected (reference src/Microassembler.cc:198-206: cycle -> next k) -- and how many of them the window
 * kernel took instead of building that graph itself.  Scheduling only: results never depend on what was built ahead. */
int lancet_engine_ahead_counts(lancet_engine *e, int32_t *built, int32_t *used);
/* Build service of the last run.  A window whose k was rejected and whose next graph was not built ahead is suspended; a few
 * resident workgroups of the LDS build kernel build that graph while the window kernel goes on with other windows, then the window
 * resumes.  out = requests posted, served, not buildable in LDS, taken back by the window kernel (general build).  Scheduling
 * only: results never depend on who built a graph. */
int lancet_engine_svc_counts(lancet_engine *e, uint32_t out[4]);
/* Profiling aid: wall-clock ticks (10 ns) the LDS build kernel's workgroups spent per phase in the last run (16 values).
 * Accounted only in an engine made with LANCET_PHASE_TIMES set in the environment (zeros otherwise; the same holds for
 * lancet_engine_phase_times): the accounting costs both kernels a clock read per phase and the build kernel contended atomics per window. */
int lancet_engine_build_phase_times(lancet_engine *e, const unsigned long long **ticks);

/* ---- the reference's -v stage trace (reference src/Microassembler.cc:87-246, Graph.cc verbose blocks) ----
 * lancet_engine_set_trace: before an upload, reserve words_per_window 32-bit words of trace events per window (0 = off).
 * lancet_engine_trace: after a run, evt_len[w] words of window w's events start at evt[w * words_per_window].
 * lancet_trace_format: those events as the text the reference prints (malloc'd, free with lancet_free); idx1 = the running
 * number of the window ("== Processing N:"), hdr / chrom / start / end = the window's name and coordinates. */
int lancet_engine_set_trace(lancet_engine *e, uint32_t words_per_window);
int lancet_engine_trace(lancet_engine *e, const uint32_t **evt_len, const uint32_t **evt, uint32_t *words_per_window);
char *lancet_trace_format(const uint32_t *words, uint32_t n_words, int32_t idx1, const char *hdr, const char *chrom,
                          int32_t start, int32_t end, int32_t dfs_limit);

/* ---- host side of the seam: Variant_t normalisation + VariantDB + VCF ----------
 * reference src/Variant.hh:106-172 (ctor), src/VariantDB.cc:28-91 (addVar), :93-179 (VCF),
 * src/Variant.cc:39-223 (printVCF). */
typedef struct lancet_filters {     /* reference src/Variant.hh:42-56, defaults src/Lancet.cc:627-637 */
  double  min_phred_fisher_str, min_phred_fisher, max_vaf_normal, min_vaf_tumor;
  int32_t min_cov_normal, max_cov_normal, min_cov_tumor, max_cov_tumor;
  int32_t min_alt_cnt_tumor, max_alt_cnt_normal, min_strand_bias, reserved;
} lancet_filters;

typedef struct lancet_vdb lancet_vdb;

void lancet_filters_default(lancet_filters *f);
lancet_vdb *lancet_vdb_create(const lancet_filters *f);
void lancet_vdb_destroy(lancet_vdb *db);
/* addVar for records [0,n) in the given order; chr_names[chr_id] gives the chromosome string. */
int  lancet_vdb_add(lancet_vdb *db, const lancet_variant *v, uint32_t n, const char *blob,
                    const char *const *chr_names, int32_t n_chr);
/* --linked-reads flavour of lancet_vdb_add (VariantDB_t(LR_MODE=true)): lr / bx_blob from lancet_engine_results_lr,
 * bx_names[rank] = the barcode strings the batch's bx_rank values refer to.  A database is either fed with
 * lancet_vdb_add only or with lancet_vdb_add_lr only. */
int  lancet_vdb_add_lr(lancet_vdb *db, const lancet_variant *v, const lancet_variant_lr *lr, uint32_t n, const char *blob,
                       const uint32_t *bx_blob, const char *const *bx_names, uint32_t n_bx,
                       const char *const *chr_names, int32_t n_chr);
/* Keys where the records are made, inserts where the VariantDB lives (a multi-GPU run: every rank keys its own records, rank 0 merges):
 * lancet_vdb_keys writes the 32 raw sha256 bytes addVar keys its map with (reference src/VariantDB.cc:36-40, Variant_t::getSignature
 * src/Variant.cc:339-344) per record; lancet_vdb_add_keyed is lancet_vdb_add / lancet_vdb_add_lr (lr, bx_blob, bx_names may be null)
 * with those keys given.  Keys that do not belong to the records are the caller's error (not checked). */
int  lancet_vdb_keys(const lancet_variant *v, uint32_t n, const char *blob, const char *const *chr_names, int32_t n_chr, uint8_t *keys);
/* keep[i] = 1 for the records of a keyed stream that can change a VariantDB (per key: the first, and the first that reaches the key's
 * largest total coverage; reference src/VariantDB.cc:45-88); a rank drops the others before its records travel to the merging rank. */
int  lancet_vdb_reduce(const lancet_variant *v, const uint8_t *keys, uint32_t n, uint8_t *keep);
int  lancet_vdb_add_keyed(lancet_vdb *db, const lancet_variant *v, const lancet_variant_lr *lr, const uint8_t *keys, uint32_t n, const char *blob,
                          const uint32_t *bx_blob, const char *const *bx_names, uint32_t n_bx, const char *const *chr_names, int32_t n_chr);
uint32_t lancet_vdb_size(const lancet_vdb *db);
/* Writes the VCF (header + sorted body) into a malloc'd string the caller frees with lancet_free.
 * date_line: text after "##fileDate=" (ctime() format incl. trailing newline), may be NULL -> omitted. */
char *lancet_vdb_vcf(lancet_vdb *db, const char *version, const char *cmdline, const char *reference,
                     const char *date_line, const char *sample_normal, const char *sample_tumor);
void lancet_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* LANCET_ENGINE_H */
