/* lancet_host.h -- C-ABI of the native host side around the engine.
 *
 * What the reference does between its command line and the processGraph call, restated for the batch interface of
 * lancet_engine.h:
 *      input decoding      BamReader + faidx                    reference src/Microassembler.cc:436-655, src/Lancet.cc:189-316
 *      window tiling       loadRefs                             reference src/Lancet.cc:189-316
 *      window order        std::map<string, Ref_t*> iteration   reference src/Microassembler.cc:779
 *      per-window filters  isRepeat on the window reference     reference src/Microassembler.cc:800, src/util.cc:295-315
 *                          isActiveRegion                       reference src/Microassembler.cc:253-432
 *                          extractReads                         reference src/Microassembler.cc:436-655
 * The output is a `lancet_window_batch` (arrays owned by the host object, valid until the next lancet_host_batch or
 * lancet_host_close) ready for lancet_engine_upload.  Pure CPU code: no device, no engine involved.
 * The bamtools / htslib libraries the reference reads its inputs with are not part of /root/reference; the readers
 * here follow the published BGZF / BAM (SAM specification v1 sections 4.1-4.2) and FASTA formats.
 */
#ifndef LANCET_HOST_H
#define LANCET_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "lancet_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* command-line knobs of the host side (defaults: reference src/Lancet.hh:33-81, src/Lancet.cc:627-637) */
typedef struct lancet_host_opts {
  int32_t padding;                 /* --padding 250                                    */
  int32_t window_size;             /* --window-size 600                                */
  int32_t min_map_qual;            /* --min-map-qual 15                                */
  int32_t max_delta_as_xs;         /* --max-as-xs-diff 5                               */
  int32_t primary_alignment_only;  /* --primary-alignment-only                         */
  int32_t xa_filter;               /* --XA-tag-filter                                  */
  int32_t max_avg_cov;             /* --max-avg-cov 10000                              */
  int32_t max_k;                   /* --max-k 101 (window reference repeat test)       */
  int32_t linked;                  /* --linked-reads: fill bx_rank / hp                */
  int32_t active_region;           /* 1 unless --active-region-off                     */
  int32_t min_evidence;            /* active regions: filters.minAltCntTumor (3)       */
  int32_t min_qual_call;           /* active regions: MIN_QUAL_CALL, ASCII (17 + '!')  */
} lancet_host_opts;

typedef struct lancet_host lancet_host;

void lancet_host_opts_default(lancet_host_opts *o);

/* Opens the inputs: the FASTA (through its .fai when there is one, so only tiled stretches are read) and the two BAMs
 * (existence only; their alignments are decoded by the tile calls).  NULL on failure with a message in err. */
lancet_host *lancet_host_open(const char *tumor_bam, const char *normal_bam, const char *ref_fasta, char *err, size_t errlen);
void lancet_host_close(lancet_host *h);
const char *lancet_host_last_error(const lancet_host *h);

/* --rg-file (Microassembler::loadRG, reference src/Microassembler.cc:29-48; applied at :302 and :616): only alignments whose RG tag is
 * one of the whitespace-separated names in the file are used ("null" stands for an alignment without RG; an empty file keeps all).
 * NULL / "": keep everything (the default).  Call before lancet_host_batch. */
int lancet_host_set_rg_file(lancet_host *h, const char *path);

/* SM of the first @RG line of the normal (which = 0) / tumor (which = 1) BAM, "NA" if there is none
 * (Microassembler::retriveSampleName, reference src/Microassembler.cc:52-67).  Valid after a tile call. */
const char *lancet_host_sample(const lancet_host *h, int which);
/* checkPresenceOfMDtag (reference src/util.cc:416-427): 1 when the first alignment of the BAM carries MD (or the BAM has
 * no alignment), else 0.  main() turns the active-region module off when neither BAM has it (src/Lancet.cc:817-825).
 * Valid after a tile call. */
int lancet_host_first_has_md(const lancet_host *h, int which);
/* isRepeat (reference src/util.cc:295-315: a k-mer seen twice among the offsets [0, len - k)) on a NUL-terminated sequence: 1 or 0.
 * The window filter of lancet_host_batch (src/Microassembler.cc:800) uses it; exported for the tests. */
int lancet_host_debug_is_repeat(const char *seq, int k);

/* Tiles "chr:start-end" (or "chr") into windows (loadRefs, reference src/Lancet.cc:189-316), puts them in processing
 * order and decodes the alignments of both BAMs the windows can select: through the .bai linear index when
 * <bam>.bai / <stem>.bai exists (one seek per tiled stretch), else by streaming the BGZF blocks once and stopping
 * after the last stretch.  Memory: the kept alignments + one slab of blocks, never the whole file.
 * Returns the number of windows, < 0 on error. */
int lancet_host_tile(lancet_host *h, const char *region, const lancet_host_opts *o);
/* The same for a BED file (--bed; loadBed, reference src/Lancet.cc:319-351 -- its intervals are padded twice, as there)
 * and/or several regions; contigs may differ.  All windows go into ONE table ordered by header string, as in the
 * reference's main() (src/Lancet.cc:852-857); chr_id of the batches indexes lancet_host_chroms(). */
int lancet_host_tile_regions(lancet_host *h, const char *bed_path, const char *const *regions, int n_regions, const lancet_host_opts *o);
const char *lancet_host_chrom(const lancet_host *h);                        /* first contig of the tiling */
const char *const *lancet_host_chroms(const lancet_host *h, int *n);        /* all of them, indexed by chr_id */
const char *lancet_host_window_hdr(const lancet_host *h, int w);            /* "chr:start-end" of tiled window w */
int lancet_host_window_chrom(const lancet_host *h, int w);                  /* its chr_id */
int lancet_host_window_span(const lancet_host *h, int w, int32_t *start, int32_t *end);

/* Windows [w_begin, w_end) of the tiling through the per-window filters and read selection.  `out` points into
 * arrays owned by h; kept[i] = tiled index of batch window i (kept has room for w_end - w_begin entries). */
int lancet_host_batch(lancet_host *h, int w_begin, int w_end, const lancet_host_opts *o, lancet_window_batch *out,
                      int32_t *kept, int32_t *n_kept);
/* Loading: a tiling normally decodes every alignment its windows can select before the first batch.  Above 400 000 windows (or with
 * LANCET_HOST_LAZY=1; =0 switches it off) and when BOTH BAMs have a .bai that parses, the tiling only builds the window table and every
 * lancet_host_batch[_packed] call loads what ITS windows can select through the index (same batches; a run of neighbouring windows per
 * call is what this is made for -- far-apart ranges cost a seek each).  When that mode is wanted but an index is missing or unreadable a
 * notice goes to stderr and the tiling loads everything at once.
 * The same batch with the reads trimmed and packed on the host threads that assemble it (lancet_pack_read's routine, with the engine's
 * parameters `P`), for lancet_engine_upload_packed: out->seq / out->qual are NULL, `pk` receives the packed arrays (owned by the host
 * object like the batch's, valid until the next batch call). */
int lancet_host_batch_packed(lancet_host *h, int w_begin, int w_end, const lancet_host_opts *o, const lancet_params *P, lancet_window_batch *out,
                             lancet_packed_reads *pk, int32_t *kept, int32_t *n_kept);
/* Batch calls need not follow each other: a call that does not start where the previous one ended works out again what the windows before
 * it left in the graph (the reference's read leak across windows, src/Microassembler.cc:83) -- an N-process run deals the window table out.
 * lancet_host_load_range: lazy mode only (else a no-op) -- loads, once, what the windows [w_begin, w_end) can select, so that the batch calls
 * inside that range do not load again (the range a rank owns). */
int lancet_host_load_range(lancet_host *h, int w_begin, int w_end, const lancet_host_opts *o);
/* barcode strings of the last batch by bx_rank (linked reads) */
const char *const *lancet_host_bx_names(const lancet_host *h, uint32_t *n);

#ifdef __cplusplus
}
#endif
#endif /* LANCET_HOST_H */
