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


def ecoli_like():
    """cfg 2 stand-in: one 4 641 652-base contig, GC 0.508, upper case, no N."""
    return Workload("ecoli-like-4.6Mb", [ContigSpec("NC_000913", 4_641_652, (2, 0, 0))],
                    gc=0.508, soft_mask=0.0, n_frac=0.0)


def tair10_like():
    """cfg 3 stand-in: 5 chromosomes + chloroplast + mitochondrion, ~119.7 Mb, GC 0.36, 0.2 % N."""
    lens = [30_427_671, 19_698_289, 23_459_830, 18_585_056, 26_975_502, 366_924, 154_478]
    specs = [ContigSpec("Chr%d" % (i + 1), n, (3, 0, i)) for i, n in enumerate(lens)]
    return Workload("tair10-like-120Mb", specs, gc=0.36, soft_mask=0.0, n_frac=0.002)
