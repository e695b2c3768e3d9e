"""Turns the engine's per-window trace events (lancet_amd/csrc/kernels.h: EV_*) into the text the reference
prints with `-v` (reference src/Microassembler.cc:87-246, src/Graph.cc verbose blocks), so that a parity break
can be localised to one graph stage by diffing against a reference trace."""
from __future__ import annotations

from typing import List, Sequence

(EV_PROCESS, EV_REPEAT_REF, EV_NEAR_REF, EV_READS, EV_STATS, EV_MARKREF, EV_LOWCOV, EV_CLEANDEAD, EV_COMPRESS, EV_CC,
 EV_CCID, EV_CCEND, EV_TRIM, EV_AMBIG_SRC, EV_NOMATCH_SRC, EV_AMBIG_SNK, EV_NOMATCH_SNK, EV_CYCLE, EV_TIPS_ROUND,
 EV_TIPS_REMOVED, EV_LINKS, EV_LOOKREP, EV_MISSING, EV_SEARCH, EV_NEAR_QRY, EV_DFSLIMIT, EV_PATH, EV_TS, EV_PATH_END,
 EV_EKA_END, EV_FOUND, EV_END) = range(1, 33)


def _bytes(words: Sequence[int], at: int, n: int):
    nwords = ((n + 3) // 4 + 7) // 8 * 8
    raw = b"".join(int(w).to_bytes(4, "little") for w in words[at:at + nwords])
    return raw[:n].decode(), at + nwords


def format_window(words: Sequence[int], idx1: int, hdr: str, chrom: str, start: int, end: int, dfs_limit: int = 1000000) -> str:
    out: List[str] = []
    i = 0
    ccids: List[int] = []
    while i + 8 <= len(words):
        code, a, b, c, d, e, f, g = (int(x) for x in words[i:i + 8])
        i += 8
        if code == EV_PROCESS:
            out.append(f"== Processing {idx1}: {hdr} numsequences: {a} mapped: {b} bastards: {a - b}\n" + "=" * 53 + "\n")
        elif code == EV_REPEAT_REF:
            out.append(f"Repeat in reference sequence for kmer {a}\n")
        elif code == EV_NEAR_REF:
            out.append(f"Near-perfect repeat in reference sequence for kmer {a}\n")
        elif code == EV_READS:
            out.append(f"reads: {a} reflen: {b} readlen: {c} cov: {c / b:.1f}\n")
        elif code == EV_STATS:
            out.append(f"  {a}: nodes: {b} edges: {c} span: {d}\n")
        elif code == EV_MARKREF:
            out.append(f"\nmark refnodes\n nodes: {a} refnodes: {b}\n")
        elif code == EV_LOWCOV:
            out.append(f"\nremoving low coverage: found {a}")
        elif code == EV_CLEANDEAD:
            out.append(f"  removing {a} dead nodes\n")
        elif code == EV_COMPRESS:
            out.append("compressing graph:")
        elif code == EV_CC:
            out.append("\nconnected components\n")
            ccids = []
            cc_nodes = a
        elif code == EV_CCID:
            ccids.append(a)
        elif code == EV_CCEND:
            out.append(f" nodes: {cc_nodes} refnodes: 0 comp: {a} refcomp: {b} refcompids: " + "".join(f" {x}" for x in ccids) + "\n")
        elif code == EV_TRIM:
            out.append(f"ref trim5: {a} trim3: {b} uncovered: {a + b} ref_dist: {c}\n")
        elif code == EV_AMBIG_SRC:
            out.append("Ambiguous match to reference for source\n")
        elif code == EV_NOMATCH_SRC:
            out.append("No match to reference for source\n")
        elif code == EV_AMBIG_SNK:
            out.append("Ambiguous match to reference for sink\n")
        elif code == EV_NOMATCH_SNK:
            out.append("No match to reference for sink\n")
        elif code == EV_CYCLE:
            out.append(f"Cycle found in the graph (kmer = {a})!\n")
        elif code == EV_TIPS_ROUND:
            out.append(f"\nremove tips round: {a}")
        elif code == EV_TIPS_REMOVED:
            out.append(f" removed: {a}\n")
        elif code == EV_LINKS:
            out.append(f"\nremove short links:  removed links: {a}\n")
        elif code == EV_LOOKREP:
            out.append("\nlooking for near-perfect repeats:\n")
        elif code == EV_MISSING:
            out.append("Missing source or sink\n")
        elif code == EV_SEARCH:
            out.append(f"\nsearching from source{a} to sink{a} dir: F\n")
        elif code == EV_NEAR_QRY:
            out.append(f"Near-perfect repeat in assembled sequence for kmer {a}\n")
        elif code == EV_DFSLIMIT:
            out.append(f"WARNING: DFS_LIMIT ({dfs_limit}) exceeded\n")
        elif code == EV_PATH:
            out.append(f">p_{chrom}:{start}-{end}_{a} cycle: {b} match: {c} snp: {d} ins: {e} del: {f}")
        elif code == EV_TS:
            ref, i = _bytes(words, i, b)
            qry, i = _bytes(words, i, b)
            hw = [int(x) for x in words[i:i + 6]]            # 12 x u16: HPRN HPRT HPAN HPAT, each {hp1, hp2, hp0}
            i += 8
            h = [(hw[q // 2] >> (16 * (q % 2))) & 0xFFFF for q in range(12)]
            hp = lambda o: f"{h[o + 2]},{h[o]},{h[o + 1]}"     # printed as hp0,hp1,hp2 (reference src/Graph.cc:1138-1141)
            out.append(f" {a}:{ref}|{qry}|R:({c >> 16}+,{c & 0xFFFF}-)n,({d >> 16}+,{d & 0xFFFF}-)t|A:({e >> 16}+,{e & 0xFFFF}-)n,"
                       f"({f >> 16}+,{f & 0xFFFF}-)t|HPref({hp(0)})n,({hp(3)})t|HPalt({hp(6)})n,({hp(9)})t|{chr(g >> 8)}|{chr(g & 0xFF)}")
        elif code == EV_PATH_END:
            out.append("\n")
        elif code == EV_EKA_END:
            out.append(f" refcomp: {a} refnodes: -2 complete: {b} allcycles: {c}\n"
                       f" perfect: {d} withsnps: {e} withindel: {f} withmix: {g} withmixindel: {g + f}\n")
        elif code == EV_FOUND:
            out.append(f" Found {a} on ref path\n")
        elif code == EV_END:
            if a:
                out.append(" Found repeat in reference\n")
            if b:
                out.append(" Found repeat in assembly\n")
            if c:
                out.append(" Found cycle in assembly\n")
            out.append("FINISHED\n")
    return "".join(out)
