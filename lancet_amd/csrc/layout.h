// layout.h -- HBM data layout shared by the host side (engine.hip) and the kernels (kernels.h).
//
// Everything a window needs while it is being assembled lives in one "slot" of work space in HBM; a
// persistent workgroup owns one slot and pulls windows from a queue.  Sizes are fixed per engine (EngineCaps);
// a window that does not fit is reported with status LANCET_W_OVERFLOW, never processed approximately.
#pragma once
#include <stdint.h>

/* Address spaces of the device pointers.  A pointer that comes out of a structure in memory is generic to the compiler
 * and gets FLAT instructions, whose completion is tracked by the LDS counter as well (every wait for an LDS read then
 * also waits for all memory loads in flight, and LDS control words reached through a generic pointer are FLAT too).
 * Typed pointers give GLOBAL / DS instructions.  Device pass only: empty in hipcc's host pass (where named address
 * spaces do not convert to generic) and for the host-compiler builds (wave emulator). */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LANCET_WAVE_EMU)
#define LC_GLOBAL __attribute__((address_space(1)))
#define LC_LDS __attribute__((address_space(3)))
#else
#define LC_GLOBAL
#define LC_LDS
#endif

#define LC_NWMAX 4            /* 64-bit words per k-mer key: k <= 128 (reference max k = 101)            */
#define LC_EMAX 12            /* edges per node (8 possible k-mer extensions + source/sink + slack)       */
#define LC_NIL 0xFFFFFFFFu
#define LC_BB 0xFFFFFFFEu     /* libstdc++ _M_before_begin sentinel in the bucket array                   */
#define LC_MAXW 1024          /* hard cap of a window's length (k-mer positions of the reference pseudo-read are 10 bits; registers of the
                                 full-matrix alignment); the work space is laid out for EngineCaps::max_w, the longest window of the batch */
#define LC_MAXW_DEFAULT 640   /* ... which is at least this (reference WINDOW_SIZE = 600): the layouts of ordinary batches do not depend on the batch */
#define LC_RS_WORDS 96        /* 64-bit words of the LDS copy of a string in repeat_scan (16 bases each) */
#define LC_STAGE 192          /* occurrences staged in LDS per round of the per-position quality counts     */
#define LC_PACK 8             /* candidates handled together in that pass                                   */
#define LC_FAT_LANES 512     /* lanes per window of the re-run tier's kernel (window_fat.hip)                  */
#define LC_SEG 128            /* k-mer starts per work item of the reference pseudo-read                    */
#define LC_MAXTS 64           /* transcripts per path                                                     */

/* per-read info word (DevBatch::rinfo) */
#define RI_TLEN(x) ((x) & 0xFFFFu)      /* trimmed length, 0 for junk reads (reference Graph_t::trim)     */
#define RI_NML(x) (((x) >> 16) & 1u)    /* label == NML                                                   */
#define RI_REV(x) (((x) >> 17) & 1u)    /* strand == REV                                                  */
#define RI_MATE(x) (((x) >> 18) & 3u)   /* mate_order 0|1|2                                               */
#define RI_MAPPED(x) (((x) >> 20) & 1u)

/* node flag bits */
#define NF_TUMOR 1u
#define NF_NORMAL 2u
#define NF_DEAD 4u
#define NF_SOURCE 8u
#define NF_SINK 16u
#define NF_INMER 32u       /* canonical k-mer is in Ref_t::mertable (reference src/Ref.cc:40-64)          */
#define NF_SURV 64u        /* survived the first removeLowCov: per-position quality counts are stored     */
#define NF_NKMER 128u      /* reference k-mer containing N: identified by its string, not by a 2-bit key */
#define NF_SPECIAL (NF_SOURCE | NF_SINK)

/* edge word: target node [27:0], dir [29:28] (FF=0 FR=1 RF=2 RR=3, reference src/Edge.hh:37), flag [30]  */
#define ED_TO(e) ((e) & 0x0FFFFFFFu)
#define ED_DIR(e) (((e) >> 28) & 3u)
#define ED_FLAG(e) (((e) >> 30) & 1u)
#define ED_MAKE(to, dir) (((uint32_t)(to)) | ((uint32_t)(dir) << 28))

/* sequence descriptor: one per base of a (compressed) node string:
 *   base [1:0], offset inside the owning k-mer [8:2], owning k-mer node id [31:9]                        */
#define SD_BASE(d) ((d) & 3u)
#define SD_OFF(d) (((d) >> 2) & 0x7Fu)
#define SD_KMER(d) ((d) >> 9)
#define SD_MAKE(kmer, off, base) (((uint32_t)(kmer) << 9) | ((uint32_t)(off) << 2) | (uint32_t)(base))

/* csr entry (cs_t): read [15:0], k-mer position in read [25:16], ori R [26], state [28:27], lr_mode "hpX had grown" [31:29].
 * The re-run tier (window_fat.hip, LANCET_FAT) keeps the read in the upper half of a 64-bit word instead: windows of more than
 * 65 535 reads (the reference assembles up to MAX_AVG_COV = 10 000x per sample, src/Microassembler.cc:491-496) run there.   */
#define CS_POS(c) (((uint32_t)(c) >> 16) & 0x3FFu)
#define CS_ORI(c) (((uint32_t)(c) >> 26) & 1u)
#define CS_ST(c) (((uint32_t)(c) >> 27) & 3u)   /* 0 counted, 1 candidate (needs the mate-overlap replay), 2 suppressed */
#ifdef LANCET_FAT
#define LC_WIDE_IDS 1
typedef unsigned long long cs_t;
typedef unsigned long long cs_key_t;
#define CS_READ(c) ((uint32_t)((c) >> 32))
#define CS_MAKE(r, p, ori, st) (((cs_t)(uint32_t)(r) << 32) | (cs_t)(((uint32_t)(p) << 16) | ((uint32_t)(ori) << 26) | ((uint32_t)(st) << 27)))
#define CS_KEY_OF(r, p) (((cs_key_t)(uint32_t)(r) << 10) | (cs_key_t)(uint32_t)(p))
#define MV_REC(r, name) (((unsigned long long)(uint32_t)(r) << 32) | (unsigned long long)(uint32_t)(name))
#define MV_READ(x) ((uint32_t)((x) >> 32))
#define MV_NAME(x) ((uint32_t)(x))
typedef unsigned long long mv_t;
#define IT_RBITS 21           /* items[]: read [20:0], first k-mer start [31:21]                                              */
#define LC_READS_MAX 0x1FFFFFu
#else
#define LC_WIDE_IDS 0
typedef uint32_t cs_t;
typedef uint32_t cs_key_t;
#define CS_READ(c) ((c) & 0xFFFFu)
#define CS_MAKE(r, p, ori, st) ((uint32_t)(r) | ((uint32_t)(p) << 16) | ((uint32_t)(ori) << 26) | ((uint32_t)(st) << 27))
#define CS_KEY_OF(r, p) (((uint32_t)(r) << 10) | (uint32_t)(p))
#define MV_REC(r, name) (((uint32_t)(r) << 16) | ((uint32_t)(name) & 0xFFFFu))
#define MV_READ(x) ((x) >> 16)
#define MV_NAME(x) ((x) & 0xFFFFu)
typedef uint32_t mv_t;
#define IT_RBITS 16           /* items[]: read [15:0], first k-mer start [31:16]                                              */
#define LC_READS_MAX 0xFFFFu
#endif
#define CS_KEY(c) CS_KEY_OF(CS_READ(c), CS_POS(c))

struct PreLayout {
  uint32_t ncap, qvcap, kw, maxw;
  uint32_t refcov, nhash, surv, snode, skey, sid, pgr, order, qv, chdr, clive, cseq, cord, chl;
  uint32_t lrnocc, lrcsr, lrcap;   /* --linked-reads only (else 0): csr offsets by node id, the csr entries of the tracked nodes, room for them */
  uint32_t stride;
};
struct EngineCaps {
  uint32_t reads_cap;    /* reads per window (incl. the reference pseudo-read)            */
  uint32_t occ_cap;      /* k-mer occurrences per (window, k)                              */
  uint32_t node_cap;     /* distinct k-mers per (window, k)                                */
  uint32_t table_cap;    /* open-addressing slots, power of two >= 2*node_cap              */
  uint32_t bucket_cap;   /* libstdc++ bucket-array simulation                              */
  uint32_t special_cap;  /* source/sink pseudo nodes                                       */
  uint32_t surv_cap;     /* nodes surviving the first low-coverage filter                  */
  uint32_t seq_cap;      /* sequence-descriptor arena (u32)                                */
  uint32_t queue_cap;    /* path-BFS queue entries                                         */
  uint32_t path_cap;     /* longest path string                                            */
  uint32_t evt_cap;      /* trace events (u32 words) per window, 0 = tracing off           */
  uint32_t var_cap;      /* variant records for the whole batch                            */
  uint32_t blob_cap;     /* variant string bytes for the whole batch                       */
  uint32_t max_k;        /* largest k the engine was created for                           */
  uint32_t qv_cap;       /* (survivor, k-mer position) entries of per-position quality counts */
  uint32_t debug_stop;   /* profiling only: abandon every window after this phase marker (0 = off)  */
  uint32_t table_start;  /* testing only: first table size of every build (power of two; 0 = estimated)       */
  uint32_t lr_mode;      /* --linked-reads: 10 instead of 4 counters per (survivor, position), barcode outputs */
  uint32_t bx_cap;       /* barcode ids (u32) of the variants' barcode sets, whole batch (lr_mode)   */
  uint32_t wide_ids;     /* work space laid out for the re-run tier's 64-bit csr words / mate-name records (read ids above 16 bits) */
  uint32_t max_w;        /* longest window reference of the batch, rounded up (>= LC_MAXW_DEFAULT, <= LC_MAXW): sizes the alignment / coverage arrays */
  struct PreLayout pl;   /* hand-off areas of the LDS build kernel (below)                           */
};

/* device-resident batch (after upload + prep) */
struct DevBatch {
  int32_t n_windows;
  LC_GLOBAL const int32_t *chr_id, *ref_start;
  LC_GLOBAL const uint32_t *ref_off;      /* [n+1] */
  LC_GLOBAL const uint8_t *ref_codes;     /* A,C,G,T -> 0..3 ; anything else 4 */
  LC_GLOBAL const uint32_t *read_begin;   /* [n+1] */
  LC_GLOBAL const uint32_t *rinfo;        /* [R] */
  LC_GLOBAL const uint32_t *name_rank;    /* [R] */
  LC_GLOBAL const uint32_t *base_woff;    /* [R] word offset of the read's packed bases (16 bases / u32) */
  LC_GLOBAL const uint32_t *good_woff;    /* [R] word offset of the read's quality mask (32 bases / u32) */
  LC_GLOBAL const uint32_t *bases;        /* 2-bit packed, trimmed reads only */
  LC_GLOBAL const uint32_t *good;         /* bit = (qual >= MIN_QUAL_CALL) */
  LC_GLOBAL const uint32_t *bx_rank;      /* [R] barcode rank or LANCET_NO_BX (lr_mode only, else null) */
  LC_GLOBAL const uint8_t *hp;            /* [R] haplotype 0|1|2 (lr_mode only, else null) */
};

struct BfsEntry {
  uint32_t parent;      /* queue index of the path this one extends, LC_NIL for the root   */
  uint32_t node;        /* last node of the path                                            */
  uint32_t edge;        /* (owner node << 4) | edge index : the Edge_t* of the reference    */
  int32_t len;
  uint16_t score;
  uint8_t dir;          /* 'F' / 'R' travel direction at `node`                             */
  uint8_t bits;         /* bit0 flag, bit1 hasCycle                                         */
};

/* Node state the graph passes walk: one 128-byte record (line 0 = flags, degree, component, colour, edges).
 * It is written once per build by the lane that gathers the node's occurrences (kernels.h build_graph). */
struct NodeGr {
  uint32_t flags, necnt;   /* NF_* , number of edges                                                               */
  int32_t comp;
  uint32_t color;
  uint32_t edges[LC_EMAX]; /* ordered (first-seen order, then the reference's erase/push_back edits)               */
  float cov[4];            /* Tf Tr Nf Nr (float, as the reference)                                                */
  int32_t mincov, mincovqv;
  uint32_t seq_lo, seq_hi, seq_clo, seq_chi;   /* descriptor deque in the seq arena                                */
  uint32_t nkm, nkmT;      /* constituent k-mers (cov_status entries >= K-1) / of which status == 'T'              */
  uint32_t nqv;            /* index into qv (LC_NIL if not stored)                                                 */
  uint32_t onref;
  uint16_t kc[4];          /* the k-mer's counted occurrences Tf Tr Nf Nr (cov_t::fwd/rev are unsigned short); never merged */
};

/* compress_prepare -> compress_fast: what one merge reads from the absorbed (single k-mer) node, in one line */
struct CmpRec {
  uint32_t lnk[2];         /* mergeable link in direction F / R: valid [31], edge dir [29:28], neighbour [27:0]          */
  float cov[4];
  uint32_t flags, nkmT;
  uint32_t d0, dK;         /* descriptors of the k-mer's first / last base                                            */
  int32_t tot, tq0, tqK;   /* computeMinCov operands of those positions                                               */
  uint32_t pad[3];
};


/* ---- hand-off of the LDS build kernel (build_lds.h) to the window kernel: one area per window of the batch ----
 * The build kernel assembles the FIRST graph of a window (the smallest k that passes the reference-repeat tests) in LDS and
 * leaves here exactly what the graph phases read: the node table in first-insertion order reduced to what is observable
 * (std::hash of every k-mer for the libstdc++ iteration order, survivor flags), dense records of the nodes that survive the
 * first removeLowCov, their per-position quality counts, the reference pseudo-read's node per offset and the reference
 * coverage.  A window the LDS limits do not hold (or that needs the mate-overlap replay, N handling, even k, k > 31) is marked
 * PB_NOT_BUILT and takes the general path (build phases of kernels.h in HBM). */
#define PB_NCAP 5632          /* distinct k-mers per window of the narrow hand-off (512-lane build: LDS table of 8192 slots) */
#define PB_NCAP_WIDE 14336    /* ... of the wide one (1024-lane build: 16384 slots)                             */
#define PB_CCAP 2048          /* candidates: nodes not decided by their occurrence count alone                 */
#define PB_SCAP 2048          /* survivors of the first removeLowCov                                          */
#define PB_QVCAP 32768        /* (candidate, k-mer position) entries of per-position quality counts, narrow hand-off */
#define PB_QVCAP_WIDE 196608  /* ... wide (2048 candidates at k = 96)                                           */
#define PB_LRCAP 40960u       /* --linked-reads: csr entries a narrow hand-off holds (= the 512-lane build's bases in LDS) */
#define PB_LRCAP_WIDE 126976u /* ... a wide one (the 1024-lane build's)                                           */
#define PB_GONE 0x40000000u   /* occ_ref: the node of this reference k-mer did not survive the first removeLowCov (its id is still in the low bits) */
#define PB_NOT_BUILT 0u
#define PB_BUILT 1u
struct PreHdr {
  uint32_t status;            /* PB_BUILT / PB_NOT_BUILT                                                        */
  uint32_t why;               /* PB_NOT_BUILT: which limit (diagnostic)                                        */
  int32_t K, refE, refM;
  uint32_t N, O, totalreadbp, n_kmers, ncand, nsurv;
  uint32_t edges_total, refn; /* trace only: sum of the edge counts of all N nodes, nodes that hold a reference k-mer */
  uint32_t have_rep;          /* refE / refM are valid (also when the graph was not built)                      */
  uint32_t have_order;        /* the survivors' table order, the hashtable state and the components are here too */
  uint32_t ht_bc, ht_next_resize;   /* libstdc++ bucket count / next rehash threshold after the N inserts        */
  uint32_t numcomp, refcomp;  /* markConnectedComponents: components, components that hold a reference k-mer     */
  uint32_t heavy;             /* scheduling hint only: a read holds the same k-mer twice (tandem duplication: the graph will have a cycle and k will climb) */
  uint32_t mapped;            /* countMappedReads of the window (valid with have_rep)                            */
  uint32_t next;              /* 1 + index of the pool area that holds this window's graph at the next k of its loop (built ahead
                                 because `heavy` says this k will be rejected, or on request of the window kernel), 0: none */
  uint32_t big;               /* built by the 1024-lane configuration (the build service runs the 512-lane one)   */
  uint32_t lr;                /* --linked-reads: the occurrences of the tracked nodes came along (PRE_OFF_LRNOCC / PRE_OFF_LRCSR): the window kernel replays
                                 barcodes and haplotypes over them (kernels.h load_prebuilt_lr) instead of building the window in HBM    */
  uint32_t lr_total;          /* ... csr entries                                                                    */
  uint32_t pad[7];
};
/* The area's layout is fixed per engine upload, not per build: the front (header, reference arrays) is the same everywhere, the
 * arrays behind it are sized by three numbers the host picks for the batch (host_common.h lc_pre_layout) -- `ncap` distinct k-mers
 * a hand-off may hold, `qvcap` (candidate, position) quality rows, `kw` 64-bit words per candidate key (1: k <= 31; 4: k <= 127).
 * A batch of deep windows (100x / 40x: 9-12 k distinct k-mers at k = 31..101) gets the wide form, the bulk of a 30x scan the narrow
 * one (660 KB per window instead of 1.7 MB).  Kernel code reads the offsets through `PL` (= EngineCaps::pl). */
#define PRE_OFF_HDR 0u
#define PRE_OFF_OCCREF 128u                                   /* u32[maxw]      node | ori << 31 per reference offset   */
#define PRE_OFF_REFCOV (PL.refcov)                            /* u16[maxw*4]                                            */
#define PRE_OFF_NHASH (PL.nhash)                              /* u64[ncap]      std::hash of every node, by node id     */
#define PRE_OFF_SURV (PL.surv)                                /* u8[ncap]       1 = survivor                            */
#define PRE_OFF_SNODE (PL.snode)                              /* u32[PB_CCAP]   node of candidate ci (LC_NIL: not a survivor) */
#define PRE_OFF_SKEY (PL.skey)                                /* u64[PB_CCAP * kw] its canonical k-mer, `kw` words each (right-aligned, low word first) */
#define PRE_OFF_SID (PL.sid)                                  /* u32[PB_SCAP]   node of survivor si                     */
#define PRE_OFF_PGR (PL.pgr)                                  /* NodeGr[PB_SCAP] records of the survivors, dense        */
#define PRE_OFF_ORDER (PL.order)                              /* u32[PB_SCAP]   the survivors in libstdc++ table order  */
#define PRE_OFF_QV (PL.qv)                                    /* u16[qvcap*4]   rows of K positions per candidate       */
#define PRE_OFF_CHDR (PL.chdr)                                /* PreCmp                                                 */
#define PRE_OFF_CLIVE (PL.clive)                              /* u32[PB_CMAX + 2]  record index (survivor index, or nsurv + k for special k) per table position */
#define PRE_OFF_CSEQ (PL.cseq)                                /* u32[PB_CSEQ]                                           */
#define PRE_OFF_CORD (PL.cord)                                /* u64[PB_CMAX]   the merged k-mers' counted occurrences (4 x 16 bit) in merge order: operands of the coverage recurrence */
#define PRE_OFF_CHL (PL.chl)                                  /* u32[3 * PB_CHEADS]  per unitig head: survivor index, first entry of its slice of CORD, merges */
#define PRE_OFF_LRNOCC (PL.lrnocc)                            /* u32[ncap + 2]  --linked-reads: csr run of node n = [lrnocc[n], lrnocc[n+1]) (empty for a node with one occurrence) */
#define PRE_OFF_LRCSR (PL.lrcsr)                              /* u32[lrcap]     ... the runs: cs_t words (read, position, orientation, state), unsorted inside a run */
#define PRE_STRIDE (PL.stride)
/* ---- first compress done by the build kernel (build_lds_impl.h bl_compress_first): single-component first graphs ----
 * markRefEnds + the first Graph_t::compress of the component in LDS; the window kernel then loads the ~20 unitigs instead of ~600
 * k-mer nodes and starts at hasCycle.  PreCmp::done == 0: nothing here, the window kernel does both itself. */
#define PB_CMAX 832u          /* survivors a graph may have for this                                                  */
#define PB_CHEADS 256u        /* unitigs with merged k-mers                                                           */
#define PB_CSEQ 12288u        /* descriptor words of their new deques                                                 */
#define PB_SPECIAL 0x0FFFFFF0u /* edge target: special node k of the component is PB_SPECIAL + k (the window kernel knows its node ids) */
struct PreCmp {
  uint32_t done;
  uint32_t m_live;            /* nodes in the table after cleanDead (two special nodes included), in CLIVE             */
  uint32_t dead;              /* k-mer nodes merged away (the cleanDead count of the trace)                            */
  uint32_t seqn;              /* words used in CSEQ: the window kernel's arena top moves by this                       */
  int32_t src_off, snk_off;   /* markRefEnds: offsets of the source / sink k-mer in the window reference              */
  uint32_t edges0;            /* trace only: edges of the survivors before markRefEnds                                 */
  uint32_t cov_heads;         /* unitig heads whose coverage (the float recurrence over the merges) is left to the window kernel: entries of CHL */
  unsigned long long spec_hash[2];   /* std::hash of "source1" / "sink1"                                               */
  uint32_t edges_all;         /* trace only: edges of ALL survivors before markRefEnds (edges0: those of component 1)         */
  uint32_t n_c1;              /* survivors in component 1 -- the component markRefEnds and the compress were done for          */
  uint32_t refmask;           /* bit q - 1: component q holds a reference k-mer (graphs of up to 32 components come along)     */
  uint32_t pad1;
};

/* One slot of work space.  All pointers are device pointers into one big allocation. */
struct Work {
  /* ---- build ---- */
  LC_GLOBAL uint32_t *occ_base;     /* [reads_cap+1] first occurrence index of each read                 */
  LC_GLOBAL uint32_t *rd;           /* [reads_cap*4] per read: info word, packed-base offset, quality-mask offset, first occurrence (one 16-byte load) */
  LC_GLOBAL uint8_t *cand;          /* [reads_cap]   read has an earlier opposite mate of the same name  */
  LC_GLOBAL uint32_t *mate_of;      /* [reads_cap]   index of that earlier mate (when unique)            */
  LC_GLOBAL uint32_t *items;        /* [2*(reads_cap + max_w/LC_SEG + 2)] work items of the per-occurrence passes */
  LC_GLOBAL uint32_t *chunk;        /* [2*((reads_cap + max_w/LC_SEG + 2)/64 + 2)] sweep origin/length per group of items */
  LC_GLOBAL uint32_t *occ;          /* [occ_cap]     slot (then node) | ori<<31                           */
  LC_GLOBAL uint32_t *slots;        /* [4*table_cap] k-mer table, 16 bytes per slot: tag (u64), first occurrence, node id */
  LC_GLOBAL uint32_t *mv;           /* [4*occ_cap] mate-name vectors of the nodes with flagged occurrences (read << 16 | name rank); wide_ids: [8*occ_cap], 64-bit records */
  LC_GLOBAL uint32_t *todo;         /* [table_cap] occurrences flagged by the mate-overlap prefilter (read << 10 | position)  */
  LC_GLOBAL unsigned long long *slot_key; /* [table_cap * LC_NWMAX]                                    */
  LC_GLOBAL uint32_t *bitmap;       /* [occ_cap/32 + 2]                                                  */
  LC_GLOBAL uint32_t *bitpre;       /* [occ_cap/32 + 2]                                                  */
  LC_GLOBAL uint32_t *csr;          /* [occ_cap] cs_t words (wide_ids: [2*occ_cap], 64-bit words)          */
  /* ---- nodes: index < node_cap are k-mers in first-insertion order; then special nodes ---- */
  LC_GLOBAL unsigned long long *nkey;     /* [nodes * LC_NWMAX] right-aligned 2-bit canonical k-mer     */
  LC_GLOBAL unsigned long long *nhash;    /* [nodes] libstdc++ std::hash<std::string> of the node id     */
  LC_GLOBAL uint32_t *nfill;        /* [nodes+1] csr fill cursors (wide_ids: [2*(nodes+1)])               */
  LC_GLOBAL NodeGr *gr;             /* [nodes]                                                           */
  LC_GLOBAL CmpRec *cmp;            /* [nodes] compress_prepare records                                   */
  LC_GLOBAL uint32_t *nocc;         /* [nodes+1] csr offsets                                             */
  LC_GLOBAL uint16_t *qv_own;       /* the slot's own array; `qv` below points at it, or -- for a graph the LDS build kernel made -- straight
                             at the rows in the window's hand-off area (they are only read after the build)                */
  LC_GLOBAL uint16_t *qv;           /* [qv_cap * QS] per-position min-quality counts Tf Tr Nf Nr (QS = 4), in lr_mode followed by
                             hp0/hp1/hp2_minqv of the tumor and of the normal (QS = 10)        */
  LC_GLOBAL uint16_t *khp;          /* [nodes*6] lr_mode: last-written hp0 hp1 hp2 of the k-mer, tumor then normal */
  LC_GLOBAL uint16_t *refhp;        /* [max_w*6] lr_mode: the same per rawseq position (Ref_t coverage)  */
  LC_GLOBAL uint32_t *bxbuf;        /* [reads_cap] lr_mode: sorted distinct barcodes of the set being collected */
  LC_GLOBAL uint32_t *lr_refnode;   /* [max_w] lr_mode, graph from the LDS build kernel: node of the reference k-mer at an offset (also when it did not survive),
                                       bit 31 = the k-mer is in Ref_t::mertable (kernels.h load_prebuilt_lr, emit_variant_lr)          */
  LC_GLOBAL uint32_t *seq;          /* [seq_cap] descriptor arena                                        */
  /* ---- libstdc++ node-table order ---- */
  LC_GLOBAL uint32_t *ht_next;      /* [nodes]                                                           */
  LC_GLOBAL uint32_t *ht_bucket;    /* [bucket_cap]                                                      */
  LC_GLOBAL uint32_t *ht_cnt;       /* [bucket_cap] parallel order stages: elements per bucket           */
  LC_GLOBAL uint32_t *ht_start;     /* [bucket_cap] parallel order stages: first list position of the run */
  LC_GLOBAL uint32_t *order;        /* [nodes] iteration order of the live table                         */
  LC_GLOBAL uint32_t *scratch;      /* [nodes*2] stacks / queues of the graph passes                     */
  /* ---- reference coverage ---- */
  LC_GLOBAL uint16_t *refcov;       /* [max_w*4] Tf Tr Nf Nr per rawseq position                       */
  /* ---- paths ---- */
  LC_GLOBAL BfsEntry *queue;        /* [queue_cap]                                                       */
  LC_GLOBAL uint32_t *pnodes;       /* [nodes] nodes of the current path                                 */
  LC_GLOBAL uint32_t *pedges;       /* [nodes] edge refs of the current path                             */
  LC_GLOBAL uint32_t *pdesc;        /* [path_cap] descriptor per path base                               */
  LC_GLOBAL uint8_t *pseq;          /* [path_cap] path string (codes 0..3)                               */
  LC_GLOBAL uint8_t *tb;            /* [(max_w+2)*(path_cap+2)] traceback bits                         */
  LC_GLOBAL int32_t *dp;            /* [7*(max_w+2)] alignment diagonals                               */
  LC_GLOBAL uint8_t *aln;           /* [2*(max_w+path_cap+2)] aligned strings (ASCII)                  */
  LC_GLOBAL uint32_t *evt;          /* [evt_cap] trace events                                            */
  LC_GLOBAL uint8_t *survb;         /* [nodes] prebuilt window: survivor flag per node (stands in for the stale node records) */
};

/* ---- build service (engine.hip svc_kernel): graphs of later k attempts built in LDS on request ----
 * A window whose k was rejected and whose next graph nobody built ahead used to build it with the general (HBM) phases on its one
 * wave: ~9 ms, the critical path of the launch.  Instead the window is SUSPENDED: its carried state (SvcCont) and a request
 * (SvcReq) are posted, the slot goes on with another window; a few resident workgroups of the LDS build kernel serve the
 * requests into pool areas (chained by PreHdr::next like graphs built ahead) and put them on a ready list, from which any slot
 * resumes the window.  Scheduling only: the records never depend on who built a graph. */
#define SV_EMPTY 0u
#define SV_POSTED 1u
#define SV_CLAIMED 2u   /* a service workgroup is building it                      */
#define SV_DONE 3u      /* on the ready list                                       */
#define SV_STOLEN 4u    /* taken back by a window slot (general build)             */
struct SvcReq { uint32_t w; int32_t k; uint32_t state; uint32_t pad; };
struct SvcCont {          /* WinShared fields that live across the k attempts of a window */
  int32_t k, seq_t5, seq_len, trim5, trim3, emit_seq, n_builds, final_k;
  uint32_t max_nodes, evt_len, N_last, sum_nodes;
  unsigned long long n_kmers;
};
struct SvcCtl {
  uint32_t req_alloc;     /* requests posted (may run past `cap`: those were not posted)      */
  uint32_t ticket;        /* next request a service workgroup waits for                       */
  uint32_t rdy_alloc, rdy_head;
  uint32_t n_resumed;     /* requests taken off the ready list or stolen                      */
  uint32_t alive;         /* service workgroups running                                       */
  uint32_t done;          /* set after the window kernel: the service workgroups leave        */
  uint32_t n_built, n_stolen, n_failed;
  uint32_t cap;
  uint32_t beat;          /* bumped by the window slots at every window / k attempt: the service's sign that somebody else runs */
  uint32_t n_gaveup;      /* service workgroups that left because nothing else made progress (kernels serialised by a profiler)  */
  uint32_t n_waiting;     /* window slots that wait for a graph: never more than there are requests out (the others leave and free their CU) */
  uint32_t large;         /* the service workgroups are the 1024-lane configuration: windows of any size the build kernels take may ask */
  uint32_t nosvc;         /* a service workgroup gave up waiting (kernels serialised by a profiler): windows stop suspending and build their later graphs themselves */
  LC_GLOBAL SvcReq *req;          /* [cap] */
  LC_GLOBAL uint32_t *rdy;        /* [cap] request index + 1 */
  LC_GLOBAL SvcCont *cont;        /* [cap] */
};

/* batch-level outputs */
struct DevOut {
  LC_GLOBAL struct lancet_variant *variants;
  LC_GLOBAL char *blob;
  LC_GLOBAL uint32_t *n_variants;   /* atomic */
  LC_GLOBAL uint32_t *n_blob;       /* atomic */
  LC_GLOBAL struct lancet_window_stats *stats;
  LC_GLOBAL uint32_t *queue_head;   /* atomic window queue */
  LC_GLOBAL struct lancet_variant_lr *variants_lr;   /* lr_mode: parallel to variants */
  LC_GLOBAL uint32_t *bx_blob;      /* lr_mode: barcode ids of the variants' barcode sets */
  LC_GLOBAL uint32_t *n_bx;         /* atomic */
  LC_GLOBAL uint32_t *evt_len;      /* [n_windows] words used in the window's trace (slot evt copied out)  */
  LC_GLOBAL uint32_t *evt_out;      /* [n_windows * evt_cap] */
  LC_GLOBAL unsigned long long *phase; /* [n_windows * 16] per-phase time (100 MHz ticks), may be null */
  LC_GLOBAL const uint32_t *win_list;  /* when non-null: the windows to process (re-run of overflowed windows)   */
  uint32_t n_list;
  LC_GLOBAL const uint8_t *pre;        /* hand-off areas of the LDS build kernel (PRE_STRIDE bytes per window), or null */
  LC_GLOBAL const uint8_t *pre_pool;   /* areas of graphs built ahead at later k (PreHdr::next chains into it), or null */
  LC_GLOBAL uint32_t *n_ahead_used;    /* atomic: window builds that took a graph built ahead                           */
  LC_GLOBAL const uint8_t *skip;       /* [n_windows] or null; non-zero: not this launch's window (a concurrent launch of the re-run tier has it) */
  LC_GLOBAL SvcCtl *svc;               /* build service of this launch, or null                                          */
};
