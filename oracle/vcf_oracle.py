"""TEST INFRASTRUCTURE ONLY -- Python restatement of the host side below the seam:
Variant_t normalisation, VariantDB_t::addVar and VCF emission.

  Variant_t ctor          reference src/Variant.hh:106-172
  getSignature            reference src/Variant.cc:339-344
  VariantDB_t::addVar     reference src/VariantDB.cc:28-91      (map keyed by sha256 hex of the signature)
  printHeader/printToVCF  reference src/VariantDB.cc:93-179, byPos src/VariantDB.hh:37-54
  Variant_t::printVCF     reference src/Variant.cc:39-223
  FET_t                   reference src/FET.hh:36-128 (lgamma/exp taken from the C library through ctypes so
                          the doubles match the reference's libm bit for bit)
  std::sort               libstdc++ introsort (bits/stl_algo.h), restated because it is unstable and the
                          reference sorts records that tie on (chr, pos)

Pinned by tests/test_oracle_golden.py against tests/golden/*.vcf (reference output)."""
from __future__ import annotations

import ctypes
import ctypes.util
import hashlib
from typing import Dict, List

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _f in ("lgamma", "exp", "log10"):
    getattr(_libm, _f).restype = ctypes.c_double
    getattr(_libm, _f).argtypes = [ctypes.c_double]

DBL_MAX = 1.7976931348623157e308

DEFAULT_FILTERS = dict(minPhredFisherSTR=25.0, minPhredFisher=5.0, minCovNormal=10, maxCovNormal=1000000,
                       minCovTumor=4, maxCovTumor=1000000, minVafTumor=0.04, maxVafNormal=0.0, minAltCntTumor=3,
                       maxAltCntNormal=0, minStrandBias=1)     # reference src/Lancet.cc:627-637


def _g(x: float) -> str:
    """operator<<(ostream, double) with default precision == printf("%g")."""
    return "%g" % x


# ---------------- FET_t -------------------------------------------------------------------------------------
def _lbinom(n, k):
    if k == 0 or n == k:
        return 0.0
    return _libm.lgamma(float(n + 1)) - _libm.lgamma(float(k + 1)) - _libm.lgamma(float(n - k + 1))


def _hypergeo(n11, n1_, n_1, n):
    return _libm.exp(_lbinom(n1_, n11) + _lbinom(n - n1_, n_1 - n11) - _lbinom(n, n_1))


class _Acc:
    __slots__ = ("n11", "n1_", "n_1", "n", "p")


def _hypergeo_acc(n11, n1_, n_1, n, aux: _Acc):
    if n1_ or n_1 or n:
        aux.n11, aux.n1_, aux.n_1, aux.n = n11, n1_, n_1, n
    else:
        if n11 % 11 and n11 + aux.n - aux.n1_ - aux.n_1:
            if n11 == aux.n11 + 1:
                aux.p *= float(aux.n1_ - aux.n11) / n11 * (aux.n_1 - aux.n11) / (n11 + aux.n - aux.n1_ - aux.n_1)
                aux.n11 = n11
                return aux.p
            if n11 == aux.n11 - 1:
                aux.p *= float(aux.n11) / (aux.n1_ - n11) * (aux.n11 + aux.n - aux.n1_ - aux.n_1) / (aux.n_1 - n11)
                aux.n11 = n11
                return aux.p
        aux.n11 = n11
    aux.p = _hypergeo(aux.n11, aux.n1_, aux.n_1, aux.n)
    return aux.p


def kt_fisher_exact(n11, n12, n21, n22) -> float:
    """Returns q, the probability of the observed table (that is what the reference uses)."""
    n1_ = n11 + n12
    n_1 = n11 + n21
    n = n11 + n12 + n21 + n22
    mx = n_1 if n_1 < n1_ else n1_
    mn = n1_ + n_1 - n
    if mn < 0:
        mn = 0
    if mn == mx:
        return 1.0
    aux = _Acc()
    q = _hypergeo_acc(n11, n1_, n_1, n, aux)
    # the tails are computed by the reference but do not influence q
    return q


# ---------------- Variant_t ---------------------------------------------------------------------------------
class Variant:
    def __init__(self, chrom: str, rec: dict, lr: bool = False, bx_names=None):
        """rec: one lancet_variant as dict (abi.variants_to_py [+ variants_lr_to_py]).  lr = Variant_t::LR_MODE."""
        self.lr = lr
        self.hp = tuple(rec.get("hp", (0,) * 12))             # HPRN HPRT HPAN HPAT, each {hp1, hp2, hp0}
        self.bx = ["", "", "", ""]                              # bxset_ref_N, bxset_ref_T, bxset_alt_N, bxset_alt_T
        if lr:
            self.bx = [";".join(bx_names[i] for i in ids) or "." for ids in rec["bx"]]
        self.kmer = rec["kmer"]
        self.str = rec["str"]
        self.chr = chrom
        self.pos = rec["pos"]
        ref_, alt_ = rec["ref"], rec["alt"]
        code = rec["code"]
        self.type = "?"
        self.len = 0
        if code == "^":
            self.type = "I"; ref_ = ""; self.len = len(alt_) & 0xFFFF
        if code == "v":
            self.type = "D"; alt_ = ""; self.len = len(ref_) & 0xFFFF
        if code == "x":
            self.type = "S"; self.pos += 1
        if code == "c":
            self.type = "C"
            ref_ = ref_.replace("-", "")
            alt_ = alt_.replace("-", "")
            rl, al = len(ref_) & 0xFFFF, len(alt_) & 0xFFFF
            self.len = al if rl == al else (rl - al if rl > al else al - rl)
        if self.type != "S":
            self.ref = rec["prev_bp_alt"] + ref_
            self.alt = rec["prev_bp_alt"] + alt_
        else:
            self.alt = alt_; self.ref = ref_; self.len = 1
        (self.rcn_f, self.rcn_r, self.rct_f, self.rct_r, self.acn_f, self.acn_r, self.act_f, self.act_r) = rec["cov"]

    def signature(self) -> str:
        return f"{self.chr}:{self.pos}:{self.type}:{self.len}:{self.ref}:{self.alt}"

    def tot(self) -> int:
        return (self.rcn_f + self.rcn_r + self.rct_f + self.rct_r + self.acn_f + self.acn_r + self.act_f + self.act_r)

    # scores
    def fet_score(self) -> float:
        prob = kt_fisher_exact(self.rcn_f + self.rcn_r, self.rct_f + self.rct_r, self.acn_f + self.acn_r,
                               self.act_f + self.act_r)
        if prob == 1.0:
            return 0.0
        if prob == 0.0:
            return -10.0 * _libm.log10(1 / DBL_MAX)
        return -10.0 * _libm.log10(prob)

    def sb_score(self) -> float:
        prob = kt_fisher_exact(self.rct_f, self.rct_r, self.act_f, self.act_r)
        if prob == 1:
            return 0.0
        return -10.0 * _libm.log10(prob) if prob > 0 else float("inf")

    @staticmethod
    def hp_score(hpr1, hpr2, hpa1, hpa2) -> float:            # Variant_t::compute_HP_score, src/Variant.cc:281-298
        prob = kt_fisher_exact(hpr1, hpr2, hpa1, hpa2)
        if prob == 1:
            return 0.0
        return -10.0 * _libm.log10(prob) if prob > 0 else float("inf")

    @staticmethod
    def genotype(R, A) -> str:
        if R > 0 and A > 0:
            return "0/1"
        if R > 0 and A == 0:
            return "0/0"
        if R == 0 and A > 0:
            return "1/1"
        return "."

    def vcf_line(self, fs: dict) -> str:
        tr_t = self.rct_f + self.rct_r
        ta_t = self.act_f + self.act_r
        tr_n = self.rcn_f + self.rcn_r
        ta_n = self.acn_f + self.acn_r
        fet = self.fet_score()
        sb = self.sb_score()
        if ta_n > 0 and ta_t > 0:
            status = "SHARED"; flag = "S"
        elif ta_n == 0 and ta_t > 0:
            status = "SOMATIC"; flag = "T"
        elif ta_n > 0 and ta_t == 0:
            status = "NORMAL"; flag = "N"
        else:
            return ""
        info = status + ";FETS=" + _g(fet)
        info += {"I": ";TYPE=ins", "D": ";TYPE=del", "S": ";TYPE=snv", "C": ";TYPE=complex"}.get(self.type, "")
        info += f";LEN={self.len};KMERSIZE={self.kmer};SB=" + _g(sb)
        hprn, hprt, hpan, hpat = self.hp[0:3], self.hp[3:6], self.hp[6:9], self.hp[9:12]
        if self.lr:                                               # src/Variant.cc:56-60, 78-80
            hpsn = self.hp_score(hprn[0], hprn[1], hpan[0], hpan[1])
            hpst = self.hp_score(hprt[0], hprt[1], hpat[0], hpat[1])
            hps = self.hp_score(hprn[0] + hpan[0], hprn[1] + hpan[1], hprt[0] + hpat[0], hprt[1] + hpat[1])
            info += ";HPS=" + _g(hps) + ";HPSN=" + _g(hpsn) + ";HPST=" + _g(hpst)
        if self.str:
            info += ";MS=" + self.str
        tumor_cov = tr_t + ta_t
        tumor_vaf = 0 if tumor_cov == 0 else ta_t / tumor_cov
        normal_cov = tr_n + ta_n
        normal_vaf = 0 if normal_cov == 0 else ta_n / normal_cov
        F: List[str] = []
        if self.str:
            if fet < fs["minPhredFisherSTR"]:
                F.append("LowFisherSTR")
        elif fet < fs["minPhredFisher"]:
            F.append("LowFisherScore")
        if normal_cov < fs["minCovNormal"]:
            F.append("LowCovNormal")
        if normal_cov > fs["maxCovNormal"]:
            F.append("HighCovNormal")
        if tumor_cov < fs["minCovTumor"]:
            F.append("LowCovTumor")
        if tumor_cov > fs["maxCovTumor"]:
            F.append("HighCovTumor")
        if tumor_vaf < fs["minVafTumor"]:
            F.append("LowVafTumor")
        if normal_vaf > fs["maxVafNormal"]:
            F.append("HighVafNormal")
        if ta_t < fs["minAltCntTumor"]:
            F.append("LowAltCntTumor")
        if ta_n > fs["maxAltCntNormal"]:
            F.append("HighAltCntNormal")
        if self.act_f < fs["minStrandBias"] or self.act_r < fs["minStrandBias"]:
            F.append("StrandBias")
        if self.lr and flag == "T" and hpat[0] > 0 and hpat[1] > 0:   # src/Variant.cc:172-177
            F.append("MultiHP")
        flt = ";".join(F) if F else "PASS"
        normal = f"{self.genotype(tr_n, ta_n)}:{tr_n},{ta_n}:{self.rcn_f},{self.rcn_r}:{self.acn_f},{self.acn_r}:{tr_n + ta_n}"
        tumor = f"{self.genotype(tr_t, ta_t)}:{tr_t},{ta_t}:{self.rct_f},{self.rct_r}:{self.act_f},{self.act_r}:{tr_t + ta_t}"
        fmt = "GT:AD:SR:SA:DP"
        if self.lr:                                               # src/Variant.cc:204-215
            fmt += ":HPR:HPA:BX"
            c = lambda t: ",".join(str(x) for x in t)
            normal += ":" + c(hprn) + ":" + c(hpan) + ":" + self.bx[0] + "," + self.bx[2]
            tumor += ":" + c(hprt) + ":" + c(hpat) + ":" + self.bx[1] + "," + self.bx[3]
        return "\t".join([self.chr, str(self.pos), ".", self.ref, self.alt, _g(fet), flt, info, fmt, normal, tumor]) + "\n"


# ---------------- VariantDB_t -------------------------------------------------------------------------------
class VariantDB:
    def __init__(self, filters: dict | None = None, lr: bool = False):
        self.lr = lr
        self.db: Dict[str, Variant] = {}
        self.filters = dict(DEFAULT_FILTERS, **(filters or {}))

    def add(self, v: Variant) -> None:
        key = hashlib.sha256(v.signature().encode()).hexdigest()
        old = self.db.get(key)
        if old is not None:
            if old.tot() < v.tot():
                old.kmer = v.kmer
                (old.rcn_f, old.rcn_r, old.rct_f, old.rct_r, old.acn_f, old.acn_r, old.act_f, old.act_r) = (
                    v.rcn_f, v.rcn_r, v.rct_f, v.rct_r, v.acn_f, v.acn_r, v.act_f, v.act_r)
                old.hp = v.hp
                if self.lr:
                    old.bx = list(v.bx)
        else:
            self.db[key] = v

    def header(self, version: str, sample_n: str, sample_t: str) -> str:
        fs = self.filters
        h = ("##fileformat=VCFv4.2\n"
             f"##source=lancet {version}\n"
             "##INFO=<ID=FETS,Number=1,Type=Float,Description=\"Phred-scaled p-value of the Fisher's exact test for tumor-normal allele counts\">\n"
             "##INFO=<ID=SOMATIC,Number=0,Type=Flag,Description=\"Somatic mutation\">\n"
             "##INFO=<ID=SHARED,Number=0,Type=Flag,Description=\"Shared mutation betweem tumor and normal\">\n"
             "##INFO=<ID=NORMAL,Number=0,Type=Flag,Description=\"Mutation present only in the normal\">\n"
             "##INFO=<ID=NONE,Number=0,Type=Flag,Description=\"Mutation not supported by data\">\n"
             "##INFO=<ID=KMERSIZE,Number=1,Type=Integer,Description=\"K-mer size used to assemble the locus\">\n"
             "##INFO=<ID=SB,Number=1,Type=Float,Description=\"Strand bias score: phred-scaled p-value of the Fisher's exact test for the forward/reverse read counts in the tumor\">\n"
             "##INFO=<ID=MS,Number=1,Type=String,Description=\"Microsatellite mutation (format: #LEN#MOTIF)\">\n"
             "##INFO=<ID=LEN,Number=1,Type=Integer,Description=\"Variant size in base pairs\">\n"
             "##INFO=<ID=TYPE,Number=1,Type=String,Description=\"Variant type (snv, del, ins, complex)\">\n")
        if self.lr:
            h += ("##INFO=<ID=HPS,Number=1,Type=Float,Description=\"Haplotype score for the T/N pair: phred-scaled p-value of the Fisher's exact test of the total counts of the two haplotype in the tumor-normal pair\">\n"
                  "##INFO=<ID=HPSN,Number=1,Type=Float,Description=\"Normal haplotype score: phred-scaled p-value of the Fisher's exact test for ref/alt haplotype counts in the normal\">\n"
                  "##INFO=<ID=HPST,Number=1,Type=Float,Description=\"Tumor haplotype score: phred-scaled p-value of the Fisher's exact test for ref/alt haplotype counts in the tumor\">\n")
        h += (f"##FILTER=<ID=LowCovNormal,Description=\"Low coverage in the normal (<{fs['minCovNormal']})\">\n"
              f"##FILTER=<ID=HighCovNormal,Description=\"High coverage in the normal (>{fs['maxCovNormal']})\">\n"
              f"##FILTER=<ID=LowCovTumor,Description=\"Low coverage in the tumor (<{fs['minCovTumor']})\">\n"
              f"##FILTER=<ID=HighCovTumor,Description=\"High coverage in the tumor (>{fs['maxCovTumor']})\">\n"
              f"##FILTER=<ID=LowVafTumor,Description=\"Low variant allele frequency in the tumor (<{_g(fs['minVafTumor'])})\">\n"
              f"##FILTER=<ID=HighVafNormal,Description=\"High variant allele frequency in the normal (>{_g(fs['maxVafNormal'])})\">\n"
              f"##FILTER=<ID=LowAltCntTumor,Description=\"Low alternative allele count in the tumor (<{fs['minAltCntTumor']})\">\n"
              f"##FILTER=<ID=HighAltCntNormal,Description=\"High alternative allele count in the normal (>{fs['maxAltCntNormal']})\">\n"
              f"##FILTER=<ID=LowFisherScore,Description=\"Low Fisher's exact test score for tumor-normal allele counts (<{_g(fs['minPhredFisher'])})\">\n"
              f"##FILTER=<ID=LowFisherSTR,Description=\"Low Fisher's exact test score for tumor-normal STR allele counts (<{_g(fs['minPhredFisherSTR'])})\">\n"
              f"##FILTER=<ID=StrandBias,Description=\"Strand bias: # of non-reference reads in either forward or reverse strand below threshold (<{fs['minStrandBias']})\">\n"
              "##FILTER=<ID=STR,Description=\"Microsatellite mutation\">\n")
        if self.lr:
            h += "##FILTER=<ID=MultiHP,Description=\"Supporting reads from multiple haplotypes based on linked-reads analysis\">\n"
        h += ("##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
              "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Depth\">\n"
              "##FORMAT=<ID=AD,Number=.,Type=Integer,Description=\"Allele depth: # of supporting ref,alt reads at the site\">\n"
              "##FORMAT=<ID=SR,Number=.,Type=Integer,Description=\"Strand counts for ref: # of supporting forward,reverse reads for reference allele\">\n"
              "##FORMAT=<ID=SA,Number=.,Type=Integer,Description=\"Strand counts for alt: # of supporting forward,reverse reads for alterantive allele\">\n")
        if self.lr:
            h += ("##FORMAT=<ID=BX,Number=.,Type=String,Description=\"Barcodes supporting ref and alt alleles\">\n"
                  "##FORMAT=<ID=HPR,Number=.,Type=Integer,Description=\"Haplotype counts for ref: # of reads supporting reference allele in haplotype 1, 2, and 0 respectively (0 = unassigned)\">\n"
                  "##FORMAT=<ID=HPA,Number=.,Type=Integer,Description=\"Haplotype counts for alt: # of reads supporting alternative allele in haplotype 1, 2, and 0 respectively (0 = unassigned)\">\n")
        h += f"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t{sample_n}\t{sample_t}\n"
        return h

    def vcf(self, version="1.1.0, October 18 2019", sample_n="NORMAL", sample_t="TUMOR") -> str:
        """Header (without the ##fileDate / ##cmdline / ##reference lines) + sorted body."""
        recs = [self.db[k] for k in sorted(self.db)]          # std::map<string,...> iteration order
        _std_sort(recs, lambda a, b: _by_pos(a, b))
        return self.header(version, sample_n, sample_t) + "".join(v.vcf_line(self.filters) for v in recs)


def _by_pos(a: Variant, b: Variant) -> bool:              # reference src/VariantDB.hh:37-54
    ca, cb = a.chr.encode(), b.chr.encode()
    if ca == cb:
        return a.pos < b.pos
    return ca < cb


# ---------------- libstdc++ std::sort (introsort), bits/stl_algo.h ------------------------------------------
def _std_sort(v: list, comp) -> None:
    n = len(v)
    if n == 0:
        return
    _introsort_loop(v, 0, n, 2 * (n.bit_length() - 1), comp)
    if n > 16:
        _insertion_sort(v, 0, 16, comp)
        for i in range(16, n):
            _unguarded_linear_insert(v, i, comp)
    else:
        _insertion_sort(v, 0, n, comp)


def _introsort_loop(v, first, last, depth, comp):
    while last - first > 16:
        if depth == 0:
            raise NotImplementedError("heap-sort fallback of std::sort not restated")
        depth -= 1
        mid = first + (last - first) // 2
        _move_median_to_first(v, first, first + 1, mid, last - 1, comp)
        cut = _unguarded_partition(v, first + 1, last, first, comp)
        _introsort_loop(v, cut, last, depth, comp)
        last = cut


def _move_median_to_first(v, result, a, b, c, comp):
    def sw(i, j):
        v[i], v[j] = v[j], v[i]
    if comp(v[a], v[b]):
        if comp(v[b], v[c]):
            sw(result, b)
        elif comp(v[a], v[c]):
            sw(result, c)
        else:
            sw(result, a)
    elif comp(v[a], v[c]):
        sw(result, a)
    elif comp(v[b], v[c]):
        sw(result, c)
    else:
        sw(result, b)


def _unguarded_partition(v, first, last, pivot, comp):
    while True:
        while comp(v[first], v[pivot]):
            first += 1
        last -= 1
        while comp(v[pivot], v[last]):
            last -= 1
        if not (first < last):
            return first
        v[first], v[last] = v[last], v[first]
        first += 1


def _insertion_sort(v, first, last, comp):
    for i in range(first + 1, last):
        if comp(v[i], v[first]):
            val = v[i]
            v[first + 1:i + 1] = v[first:i]
            v[first] = val
        else:
            _unguarded_linear_insert(v, i, comp)


def _unguarded_linear_insert(v, last, comp):
    val = v[last]
    nxt = last - 1
    while comp(val, v[nxt]):
        v[last] = v[nxt]
        last = nxt
        nxt -= 1
    v[last] = val
