"""Seeded synthetic stand-ins for the genomes BASELINE.json names.

The real assemblies are not available offline; these have matching size, contig
structure, GC content, soft-mask fraction and N content.  Every contig is
generated from its own seed, so a rank can build just the contigs it owns.

A contig is returned as the CHARACTER STRING the reference would scan for a
single-token FASTA header in the re-formatted path: a leading
quote, the bases, and the trailing  '),  (or  ')]  for the last contig).
"""
import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _base_lut(gc):
    """256-entry byte -> base table with P(G)=P(C)=gc/2."""
    n_gc = int(round(gc / 2 * 256))
    n_at = 128 - n_gc
    lut = np.empty(256, dtype=np.uint8)
    lut[:n_gc] = ord("G")
    lut[n_gc:2 * n_gc] = ord("C")
    lut[2 * n_gc:2 * n_gc + n_at] = ord("A")
    lut[2 * n_gc + n_at:] = ord("T")
    return lut


class ContigSpec:
    def __init__(self, name, length, seed, twin_of=None):
        self.name = name
        self.length = int(length)
        self.seed = seed          # tuple fed to SeedSequence
        self.twin_of = twin_of    # spec whose sequence this one is a 3 %-substituted copy of


class Workload:
    """A list of ContigSpec + composition parameters."""

    def __init__(self, name, specs, gc, soft_mask, n_frac, mask_run=2000):
        self.name = name
        self.specs = specs
        self.gc = gc
        self.soft_mask = soft_mask
        self.n_frac = n_frac
        self.mask_run = mask_run

    @property
    def n_bases(self):
        return sum(s.length for s in self.specs)

    def _raw(self, spec):
        rng = np.random.default_rng(np.random.SeedSequence(spec.seed))
        n = spec.length
        a = _base_lut(self.gc)[rng.integers(0, 256, n, dtype=np.uint8)]
        if self.n_frac > 0:
            n_runs = max(1, int(self.n_frac * n / 5050))
            starts = rng.integers(0, max(1, n - 100), n_runs)
            lens = rng.integers(100, 10001, n_runs)
            for st, ln in zip(starts.tolist(), lens.tolist()):
                a[st:st + ln] = ord("N")
        if self.soft_mask > 0:
            mean_masked = self.mask_run
            mean_clear = mean_masked * (1 - self.soft_mask) / self.soft_mask
            n_pairs = int(n / (mean_masked + mean_clear) * 1.3) + 16
            runs = np.empty(2 * n_pairs, dtype=np.int64)
            runs[0::2] = rng.geometric(1.0 / mean_clear, n_pairs)
            runs[1::2] = rng.geometric(1.0 / mean_masked, n_pairs)
            flags = np.zeros(2 * n_pairs, dtype=np.uint8)
            flags[1::2] = 0x20
            mask = np.repeat(flags, runs)[:n]
            if mask.size < n:
                mask = np.concatenate([mask, np.zeros(n - mask.size, dtype=np.uint8)])
            letters = a != ord("N")
            a |= mask * letters  # lower-case = soft-masked
        return a

    def bases(self, spec):
        """uint8 bases of one contig (no decoration)."""
        if spec.twin_of is None:
            a = self._raw(spec)
            self._last = (spec, a.copy())  # its homeolog usually comes next
            return a
        last = getattr(self, "_last", None)
        if last is not None and last[0] is spec.twin_of:
            a = last[1]
            self._last = None
        else:
            a = self._raw(spec.twin_of)
        rng = np.random.default_rng(np.random.SeedSequence(spec.seed))
        idx = np.nonzero(rng.integers(0, 256, a.size, dtype=np.uint8) < 8)[0]  # ~3 % of positions
        sub = _BASES[rng.integers(0, 4, idx.size)]
        keep_case = a[idx] & 0x20
        isn = a[idx] == ord("N")
        a[idx] = np.where(isn, a[idx], sub | keep_case)
        return a

    def contig_string(self, k):
        """The string the reference scans for contig k (decorated)."""
        spec = self.specs[k]
        tail = b"')]" if k == len(self.specs) - 1 else b"'),"
        return np.concatenate([np.frombuffer(b"'", dtype=np.uint8), self.bases(spec),
                               np.frombuffer(tail, dtype=np.uint8)])


def _lengths(rng, n_chr, lo, hi, n_scaf, scaf_lo, scaf_hi):
    chrom = rng.integers(lo, hi, n_chr)
    scaf = np.exp(rng.uniform(np.log(scaf_lo), np.log(scaf_hi), n_scaf)).astype(np.int64)
    return chrom.tolist(), scaf.tolist()


def switchgrass_like(genome=0, scale=1.0):
    """cfg 5 stand-in: 18 chromosomes as 9 homeologous pairs (40-78 Mb) + 626
    scaffolds (1-500 kb, log-uniform), ~1.13 Gb, GC 0.46, 55 % soft-masked, 3 % N."""
    rng = np.random.default_rng(np.random.SeedSequence([5, 0]))  # lengths shared by all genomes
    chrom, scaf = _lengths(rng, 9, 40_000_000, 78_000_000, 626, 1_000, 500_000)
    # nudge the chromosome total so the genome is ~1.13 Gb like P. virgatum v5
    target = 1.13e9 - sum(scaf)
    f = target / (2 * sum(chrom))
    chrom = [int(c * f * scale) for c in chrom]
    scaf = [max(64, int(s * scale)) for s in scaf]
    specs = []
    for i, n in enumerate(chrom):
        a = ContigSpec("Chr%02dK" % (i + 1), n, (5, genome, 2 * i))
        specs.append(a)
        specs.append(ContigSpec("Chr%02dN" % (i + 1), n, (5, genome, 2 * i + 1), twin_of=a))
    for i, n in enumerate(scaf):
        specs.append(ContigSpec("scaffold_%d" % (i + 1), n, (5, genome, 1000 + i)))
    return Workload("switchgrass-like-1.13Gb" if scale == 1.0 else "switchgrass-like-x%g" % scale,
                    specs, gc=0.46, soft_mask=0.55, n_frac=0.03)


def sorghum_like():
    """cfg 4 stand-in: 10 chromosomes (52-81 Mb) + 857 scaffolds (1-200 kb), ~730 Mb, GC 0.44,
    60 % soft-masked, 1 % N."""
    rng = np.random.default_rng(np.random.SeedSequence([4, 0]))
    chrom, scaf = _lengths(rng, 10, 52_000_000, 81_000_000, 857, 1_000, 200_000)
    f = (730e6 - sum(scaf)) / sum(chrom)
    chrom = [int(c * f) for c in chrom]
    specs = [ContigSpec("Chr%02d" % (i + 1), n, (4, 0, i)) for i, n in enumerate(chrom)]
    specs += [ContigSpec("super_%d" % (i + 11), n, (4, 0, 1000 + i)) for i, n in enumerate(scaf)]
    return Workload("sorghum-like-730Mb", specs, gc=0.44, soft_mask=0.60, n_frac=0.01)


def maize_like():
    """A genome beyond one arena (2^31 characters): 10 chromosomes (150-310 Mb) + 300 scaffolds (1-300 kb), ~2.4 Gb like
    Zea mays B73 v4, GC 0.47, 85 % soft-masked, 0.5 % N.  Not a BASELINE.json config: the size the node handle's
    several-arenas-per-device path is held to (the reference reads a genome of any size whole, CROPSR.py:59)."""
    rng = np.random.default_rng(np.random.SeedSequence([6, 0]))
    chrom, scaf = _lengths(rng, 10, 150_000_000, 310_000_000, 300, 1_000, 300_000)
    f = (2.4e9 - sum(scaf)) / sum(chrom)
    chrom = [int(c * f) for c in chrom]
    specs = [ContigSpec("chr%d" % (i + 1), n, (6, 0, i)) for i, n in enumerate(chrom)]
    specs += [ContigSpec("B73V4_ctg%d" % (i + 1), n, (6, 0, 1000 + i)) for i, n in enumerate(scaf)]
    return Workload("maize-like-2.4Gb", specs, gc=0.47, soft_mask=0.85, n_frac=0.005)


def ecoli_like():
    """cfg 2 stand-in: one 4 641 652-base contig, GC 0.508, upper case, no N."""
    return Workload("ecoli-like-4.6Mb", [ContigSpec("NC_000913", 4_641_652, (2, 0, 0))],
                    gc=0.508, soft_mask=0.0, n_frac=0.0)


def tair10_like():
    """cfg 3 stand-in: 5 chromosomes + chloroplast + mitochondrion, ~119.7 Mb, GC 0.36, 0.2 % N."""
    lens = [30_427_671, 19_698_289, 23_459_830, 18_585_056, 26_975_502, 366_924, 154_478]
    specs = [ContigSpec("Chr%d" % (i + 1), n, (3, 0, i)) for i, n in enumerate(lens)]
    return Workload("tair10-like-120Mb", specs, gc=0.36, soft_mask=0.0, n_frac=0.002)


def synthetic_annotation(wl, gff_path, info_path=None, n_genes=30000, seed=11):
    """A seeded Phytozome-style GFF3 (+ annotation_info.txt) for a Workload -- the stand-in for the annotation files
    BASELINE.json configs[2] / [3] name (the real ones are not available offline).  n_genes gene models are dealt to
    the contigs in proportion to their length (every contig of >= 20 kb gets at least one): gene + mRNA + exon +
    CDS (+ UTR) rows, 1-based closed coordinates, 1-9 CDS per gene; ~6 % of the genes start inside their predecessor
    (overlap, often on the other strand), ~2 % lie nested inside it, some CDS rows carry no ID (only Parent).  Two
    thirds of the genes get a line in the annotation_info file.  Returns the number of gene and CDS rows written."""
    rng = np.random.default_rng(np.random.SeedSequence([seed, len(wl.specs), n_genes]))
    lens = np.array([s.length for s in wl.specs], dtype=np.int64)
    share = np.maximum((lens >= 20000).astype(np.int64), np.floor(n_genes * lens / lens.sum()).astype(np.int64))
    n_rows = [0, 0]
    info = None
    if info_path:
        info = open(info_path, "w")
        info.write("#pacId\tlocusName\ttranscriptName\tpeptideName\tPfam\tPanther\tKOG\tKEGG/ec\tKO\tGO\t"
                   "Best-hit-arabi-name\tarabi-symbol\tarabi-defline\n")
    deflines = ["protein kinase superfamily protein", "NAC domain containing protein 1", "", "F-box family protein",
                "Leucine-rich repeat (LRR) family protein", "RING/U-box superfamily protein, putative"]
    with open(gff_path, "w") as f:
        f.write("##gff-version 3\n##annot-version synthetic\n")
        pac = 37000000
        for ci, (spec, n_here) in enumerate(zip(wl.specs, share.tolist())):
            if n_here == 0 or spec.length < 2000:
                continue
            starts = np.sort(rng.integers(1, max(2, spec.length - 1500), n_here))
            glen = np.minimum(np.exp(rng.normal(7.8, 0.7, n_here)).astype(np.int64) + 300, 40000)
            prev = None
            out = []
            for g in range(n_here):
                a, b = int(starts[g]), int(min(spec.length, starts[g] + glen[g]))
                u = rng.random()
                if prev is not None and prev[1] - prev[0] > 900:
                    if u < 0.06:    # starts inside the previous gene
                        a = int(rng.integers(prev[0] + 1, prev[1]))
                        b = int(min(spec.length, max(b, a + 400)))
                    elif u < 0.08:  # nested in the previous gene
                        a = int(rng.integers(prev[0] + 1, prev[1] - 400))
                        b = int(rng.integers(a + 200, prev[1]))
                if b - a < 200:
                    continue
                prev = (a, b)
                strand = "+" if rng.random() < 0.5 else "-"
                locus = "%s.%03dG%06d" % (wl.name[:5].replace("-", ""), ci + 1, (g + 1) * 100)
                gid, tid = locus + ".v1.1", locus + ".1.v1.1"
                pac += 1
                out.append("%s\tphytozomev12\tgene\t%d\t%d\t.\t%s\t.\tID=%s;Name=%s\n" % (spec.name, a, b, strand, gid, locus))
                out.append("%s\tphytozomev12\tmRNA\t%d\t%d\t.\t%s\t.\tID=%s;Name=%s.1;pacid=%d;longest=1;Parent=%s\n"
                           % (spec.name, a, b, strand, tid, locus, pac, gid))
                n_rows[0] += 1
                n_ex = int(min(9, 1 + rng.geometric(0.3)))
                cuts = np.sort(rng.choice(np.arange(a, b + 1), size=min(2 * n_ex, b - a + 1), replace=False))
                for e in range(len(cuts) // 2):
                    ea, eb = int(cuts[2 * e]), int(cuts[2 * e + 1])
                    out.append("%s\tphytozomev12\texon\t%d\t%d\t.\t%s\t.\tID=%s.exon.%d;Parent=%s;pacid=%d\n"
                               % (spec.name, ea, eb, strand, tid, e + 1, tid, pac))
                    if e == 0 and eb - ea > 60 and rng.random() < 0.5:
                        out.append("%s\tphytozomev12\tfive_prime_UTR\t%d\t%d\t.\t%s\t.\tID=%s.five_prime_UTR.1;Parent=%s;pacid=%d\n"
                                   % (spec.name, ea, ea + 30, strand, tid, tid, pac))
                        ea += 31
                    ident = "ID=%s.CDS.%d;" % (tid, e + 1) if rng.random() < 0.97 else ""
                    out.append("%s\tphytozomev12\tCDS\t%d\t%d\t.\t%s\t%d\t%sParent=%s;pacid=%d\n"
                               % (spec.name, ea, eb, strand, e % 3, ident, tid, pac))
                    n_rows[1] += 1
                if info is not None and rng.random() < 0.67:
                    at = "AT%dG%05d.1" % (int(rng.integers(1, 6)), int(rng.integers(1000, 80000)))
                    info.write("%d\t%s\t%s.1\t%s.1.p\tPF%05d\tPTHR%05d\t\t\t\tGO:%07d\t%s\t%s\t%s\n"
                               % (pac, locus, locus, locus, int(rng.integers(1, 20000)), int(rng.integers(10000, 48000)),
                                  int(rng.integers(1, 99999)), at, "SYM%d" % g if g % 3 else "", deflines[g % len(deflines)]))
            f.write("".join(out))
    if info is not None:
        info.close()
    return tuple(n_rows)
