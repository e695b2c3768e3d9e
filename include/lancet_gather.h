/* lancet_gather.h -- C-ABI of the multi-process record gather.
 *
 * Windows are independent, so an N-GPU run is N processes (one per GPU) that each assemble their share of the windows; the only
 * exchange is the gather of the Variant records into the VariantDB on rank 0.  This replaces the reference's merge of its
 * per-thread databases into one (reference src/Lancet.cc:940-959: `vDB.addVar(...)` over every thread's map, in thread order) and
 * keeps what that merge guarantees for a single thread: addVar sees the records in (window, emission) order, whatever N is
 * (src/VariantDB.cc:28-91 keeps the FIRST record of a key and the counts of the first record that reaches the key's largest total
 * coverage -- SURVEY.md H7).
 *
 *   lancet_records_pack    a rank's records of one batch -> bytes (records with GLOBAL window numbers, their strings, with
 *                          --linked-reads the barcode sets by NAME, the contig names, the 32-byte addVar keys; optionally only
 *                          the records that can change a database: lancet_vdb_reduce)
 *   lancet_records_merge   rank 0: any number of such parts, from any ranks, in any order -> replayed into a lancet_vdb in
 *                          (global window, emission) order
 *   lancet_comm_*          the transport: RCCL (librccl, loaded on first use).  Sizes by one ncclAllGather of 8 bytes per rank,
 *                          payloads by grouped ncclSend / ncclRecv to rank 0 only -- a gatherv over xGMI; the ncclUniqueId
 *                          travels through a file that rank 0 writes (rendezvous path, e.g. on /dev/shm).
 * The byte format is the one lancet_amd/dist.py (the harness under bench.py) packs and unpacks; tests hold the two against each other.
 */
#ifndef LANCET_GATHER_H
#define LANCET_GATHER_H

#include <stddef.h>
#include <stdint.h>
#include "lancet_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One batch's records as a malloc'd byte string (*out, *out_len; free with lancet_free).
 *   v, n, blob, blob_len             as lancet_engine_results returns them (records in (window, emission) order)
 *   lr, bx_blob, bx_names, n_bx      --linked-reads: lancet_engine_results_lr + the batch's barcode names by bx_rank; else NULL / 0
 *   chr_names, n_chr                 chr_names[chr_id] for the records' chr_id
 *   window_index, n_windows          window_index[w] = GLOBAL number of the batch's window w (strictly increasing; the tiled index
 *                                    lancet_host_batch reports in kept[]); NULL: the records keep their numbers
 *   reduce                           != 0: only the records that can change a database travel (their keys are computed either way)
 * Returns LANCET_OK, LANCET_E_ARG (a chr_id / window / barcode id out of range) or LANCET_E_OOM. */
int lancet_records_pack(const lancet_variant *v, uint32_t n, const char *blob, uint32_t blob_len,
                        const lancet_variant_lr *lr, const uint32_t *bx_blob, const char *const *bx_names, uint32_t n_bx,
                        const char *const *chr_names, int32_t n_chr, const int64_t *window_index, uint32_t n_windows,
                        int reduce, uint8_t **out, size_t *out_len);

/* Replays n_parts packed parts into db in (global window, emission) order (parts that arrive in that order are not sorted again).
 * Linked-read parts and ordinary parts cannot be mixed.  *n_added (may be NULL) = records handed to the database. */
int lancet_records_merge(lancet_vdb *db, const uint8_t *const *parts, const size_t *lens, int n_parts, uint32_t *n_added);

typedef struct lancet_comm lancet_comm;

/* Joins the communicator of `world` processes as `rank` on GPU `device`.  rendezvous: a path all ranks agree on and can reach; rank 0
 * creates it (the ncclUniqueId), the others wait for it (timeout_s), rank 0 removes it in lancet_comm_destroy.
 * NULL on failure with a message in err.  (Transport "files" -- set by LANCET_COMM_TEST_FILES=1, test use only: payloads travel through
 * files next to the rendezvous path, for driving the N-process path on a box with one GPU, where RCCL refuses two ranks per device.) */
lancet_comm *lancet_comm_create(int rank, int world, int device, const char *rendezvous, double timeout_s, char *err, size_t errlen);
/* Variable-size gather to rank 0.  On rank 0: *all = malloc'd concatenation of every rank's payload in rank order (free with
 * lancet_free), lens[r] = bytes of rank r (lens has room for `world` entries).  Elsewhere: *all = NULL. */
int lancet_comm_gather(lancet_comm *c, const uint8_t *payload, size_t len, uint8_t **all, size_t *lens);
const char *lancet_comm_last_error(const lancet_comm *c);
const char *lancet_comm_transport(const lancet_comm *c);           /* "rccl" | "files" */
void lancet_comm_destroy(lancet_comm *c);

#ifdef __cplusplus
}
#endif
#endif /* LANCET_GATHER_H */
