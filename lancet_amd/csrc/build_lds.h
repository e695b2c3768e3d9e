// build_lds.h -- the first graph of every window, assembled in LDS (one 256-lane workgroup per window).
//
// What buildgraph / loadSequence (reference src/Graph.cc:119-349, 530-589) and the first removeLowCov(false, 0)
// (src/Graph.cc:2790-2827) leave behind, computed with the window's working set in LDS instead of HBM:
//   * the window's trimmed reads + the reference pseudo-read, 2 bit / base, and their quality masks: copied into LDS once;
//   * the k-mer table: 8192 open-addressing slots of 4 bytes = 16-bit fingerprint | offset (in the LDS copy of the reads) of
//     the k-mer's first occurrence.  A fingerprint match is confirmed by cutting that earlier k-mer out of the LDS reads, so the
//     table needs no key storage and atomicMin on the word keeps the FIRST occurrence (= first-insertion order of
//     std::unordered_map, which fixes the node ids);
//   * per node: occurrence count, then -- only for the nodes whose count leaves the first low-coverage test open ("tracked") --
//     the four strand/sample counts, colours and mate-pair signatures with LDS atomics (no csr, no per-occurrence HBM traffic);
//   * per-position quality counts of the candidates as "counted - bases below MIN_QUAL_CALL" (about one LDS atomic per
//     occurrence instead of k), in groups of candidates that fit LDS;
//   * first-seen edge stamps of the survivors (atomicMin per (node, side, base) slot), resolved to edges in first-seen order.
// HBM sees the packed reads once (coalesced), a 2-byte node id per occurrence (written once, streamed by the later passes) and
// the hand-off area of the window (layout.h: PreHdr ...), which holds only what the graph phases of kernels.h read.
//
// The kernel builds the graph for the smallest k that passes the reference-repeat tests (Microassembler.cc:118-131) -- the
// first buildgraph of the self-tuning-k loop; later k attempts of a window (8 % of the windows of a 30x/30x scan) and every
// window outside the limits below run the general build phases of kernels.h.  Nothing is approximated: a window either gets
// the identical node table here or is marked PB_NOT_BUILT.
//
// Not handled here (-> PB_NOT_BUILT, general path): --linked-reads, N in the window reference, even k or k > 31, windows
// above the LDS limits, windows with an occurrence the mate-overlap prefilter flags (hasOverlappingMate needs the replay of
// build_csr), reads whose name occurs more than once with the same mate number.
#pragma once
#include <stddef.h>
#include "kernels.h"

#define BL_EMPTY 0xFFFFFFFFu
#ifndef BL_INS
#define BL_INS 4                  /* occurrences a lane inserts at a time (their LDS reads overlap)                              */
#endif
#ifndef BL_INFLIGHT
#define BL_INFLIGHT 4             /* occurrence words a lane has in flight in a pass over the window's occurrences (bl_for_occ) */
#endif
#define BL_DUPCAP 1024u
#ifndef LANCET_WAVE_EMU
#define BL_UNROLL _Pragma("unroll")
#else
#define BL_UNROLL
#endif
// The tail of a window's build (table order, components, first compress: ~165 workgroup barriers with little work between them) at a raised
// wave priority: its waves mostly wait, and what they issue between two barriers is on the critical path of the window, while the other
// workgroup of the CU is, most of the time, in an occurrence pass that is bound by VALU throughput and fills whatever slots are left.
#ifndef BL_TAIL_PRIO
#define BL_TAIL_PRIO 0
#endif
#if !defined(LANCET_WAVE_EMU)
#define BL_TAIL_PRIO_SET(p) do { if (BL_TAIL_PRIO) __builtin_amdgcn_s_setprio(p); } while (0)
#else
#define BL_TAIL_PRIO_SET(p) ((void)0)
#endif
enum { BLW_NONE = 0, BLW_NOREADS, BLW_SIZE, BLW_HASN, BLW_K, BLW_TABLE, BLW_NODES, BLW_TRACKED, BLW_CAND, BLW_QV, BLW_SURV, BLW_MATE, BLW_NAMES, BLW_PAIRS,
       BLW_KBIG /* k > 31: the 1024-lane configuration's (keys of up to four words) */ };

#ifdef LANCET_WAVE_EMU
#define BL_DBG(...) do { if (getenv("LANCET_EMU_WHY")) fprintf(stderr, __VA_ARGS__); } while (0)
#else
#define BL_DBG(...) ((void)0)
#endif
#ifndef LANCET_WAVE_EMU
#define BLP(S, id) do { if (threadIdx.x == 0 && (S).ph_cur >= 0) { const unsigned long long _t = wall_clock64();      /* (ph_cur < 0: nobody asked for phase times, LANCET_PHASE_TIMES) */ (S).ph_acc[(S).ph_cur] += _t - (S).t_last; (S).t_last = _t; (S).ph_cur = (id); } } while (0)
#else
static thread_local unsigned long bl_emu_sync_acc[17], bl_emu_sync_last = 0; static thread_local int bl_emu_sync_cur = 16;   /* barriers per phase (tuning aid) */
#define BLP(S, id) do { bl_emu_sync_acc[bl_emu_sync_cur] += lc_emu_syncs - bl_emu_sync_last; bl_emu_sync_last = lc_emu_syncs; bl_emu_sync_cur = (id); } while (0)
#endif
// uniform read of a control word: barrier, read, barrier (kernels.h wg_bcast)

// Two size configurations of the same code:
//   bl_small  512 lanes, 40 960 bases, 512 reads, 80 KB of LDS: two workgroups per CU -- the 30x/30x windows;
//   bl_large 1024 lanes, 126 976 bases (17-bit offsets under a 15-bit fingerprint), 1024 reads, a 64 KB phase area (room for the
//            mate-overlap replay of 8192 occurrences), ~158 KB of LDS: one workgroup per CU -- windows the small one turns away
//            for their size (60x/60x: ~360 reads, 58 k bases), taken off the list the small kernel leaves.
//            Round 4: it is also the configuration for k > 31 (k-mers of up to four 64-bit words cut out of the LDS reads, BL_KW) and
//            for large graphs: a 16384-slot table (all of its phase area), 14 336 distinct k-mers (node -> first-occurrence offset in
//            its HBM scratch instead of LDS, 32-bit per-occurrence words) -- windows of 100x / 40x build at k = 31..101 with 9-12 k nodes.
// Same limits on what leaves the workgroup (PB_CCAP candidates, PB_SCAP survivors; BL_NCAP nodes within the hand-off's PreLayout::ncap).
#ifndef BL_SMALL_WG
#define BL_SMALL_WG 512           /* lanes of the small configuration (tuning builds: tools/variant.sh)          */
#define BL_SMALL_EU 4             /* waves per SIMD it is compiled for: 2 workgroups per CU                      */
#endif
#ifndef BL_SMALL_BASES
#define BL_SMALL_BASES 40960
#endif
#define BL_NS bl_small
#define BL_WG BL_SMALL_WG
#define BL_BASES BL_SMALL_BASES   /* bases in LDS (reads padded to 16, + the reference)                       */
#define BL_RMAX 512               /* reads per window                                                          */
#define BL_SLOTS 8192
#define BL_NCAP PB_NCAP           /* distinct k-mers                                                           */
#define BL_KW 1                   /* 64-bit words of a k-mer: k <= 31                                          */
#define BL_KMAX 31
#define BL_WIDE 0                 /* 16-bit per-occurrence words, node -> first-occurrence offset in LDS       */
#define BL_TCAP 2048              /* tracked nodes                                                             */
#define BL_BIG 32768              /* bytes of the phase-dependent LDS area                                     */
#define BL_OFFBITS 16             /* bits of an LDS base offset                                                */
#define BL_FLAGCAP 1024u          /* occurrences the mate-overlap prefilter may flag                           */
#define BL_LDS_LIMIT (80u * 1024u)
#include "build_lds_impl.h"
#undef BL_NS
#undef BL_WG
#undef BL_BASES
#undef BL_RMAX
#undef BL_BIG
#undef BL_OFFBITS
#undef BL_FLAGCAP
#undef BL_LDS_LIMIT
#undef BL_SLOTS
#undef BL_NCAP
#undef BL_KW
#undef BL_KMAX
#undef BL_WIDE
#define BL_NS bl_large
#define BL_WG 1024
#define BL_BASES 126976
#define BL_RMAX 1024
#define BL_SLOTS 16384
#define BL_NCAP PB_NCAP_WIDE
#define BL_KW 4
#define BL_KMAX 127
#define BL_WIDE 1
#define BL_BIG 65536
#define BL_OFFBITS 17
#define BL_FLAGCAP 2048u
#define BL_LDS_LIMIT (160u * 1024u)
#include "build_lds_impl.h"
#undef BL_NS
#undef BL_WG
#undef BL_BASES
#undef BL_RMAX
#undef BL_SLOTS
#undef BL_NCAP
#undef BL_KW
#undef BL_KMAX
#undef BL_WIDE
#undef BL_TCAP
#undef BL_BIG
#undef BL_OFFBITS
#undef BL_FLAGCAP
#undef BL_LDS_LIMIT
