"""--annotate: the GFF / Phytozome annotation join the reference never performs.

The reference parses the GFF into a DataFrame (CROPSR.py:77-95, called at :375), drops it, and
writes the constant '' into the `features` column (CROPSR.py:466, :468); `-p` is only echoed in
the banner (:364).  A real join changes the CSV, so it is OPT-IN here and the default output
stays byte-identical.  Being absent from the reference, the join has no reference oracle
("parity unpinned"); its definition is this module's, and tests/test_annotate.py checks it
against a brute-force restatement.

Definition.  A row of the CSV that has a cut site (the 12-field rows; cutsite = end_pos - 3,
CROPSR.py:155-158) gets, in `features`, the ';'-joined labels of every GFF row whose type is `gene`
or `CDS`, whose seqid is the contig's FASTA name and whose [start, end] (GFF: 1-based, closed)
contains the cut site's 1-based genome coordinate, in GFF file order, each label once.
  label = "<type>:<ID>"  (ID attribute; else Name, else Parent, else "."), and for a gene whose Name
  (or ID) is a locusName of the Phytozome annotation_info file: + "|<Best-hit-arabi-name>" +
  "|<arabi-defline>" (fields that are empty are left out).
Coordinates: the reference's positions index the string it scans; in the re-formatted path that
string starts with one decoration character, so genome coordinate (1-based) =
cutsite - dec + 1 with dec = 1 there and 0 for an unformatted FASTA.

Method: per seqid the feature intervals cut the coordinate axis into elementary intervals with a
constant label set (a sweep over the 2F interval ends builds one string per distinct set); a hit
then needs one binary search (numpy searchsorted over the sorted cut points).
"""
import re

import numpy as np

NO_FEATURE = 0xFFFFFFFF
TYPES = ("gene", "CDS")
_ATTR = re.compile(r"(?:^|;)\s*(ID|Name|Parent)=([^;]*)")


def contig_name(key):
    """FASTA name of a contig from the reference's dict key: "[('Chr01'," / "('c2',"
    in the re-formatted path, ">name" otherwise."""
    if key.startswith(">"):
        return key[1:]
    k = key[1:] if key.startswith("[") else key
    if k.startswith("('"):
        k = k[2:]
    if k.endswith("',"):
        k = k[:-2]
    return k


def parse_gff(path):
    """[(seqid, type, start, end, attributes)] of the gene / CDS rows, in file order."""
    out = []
    with open(path, "r") as f:
        for line in f:
            if not line or line[0] == "#":
                continue
            cols = line.rstrip("\n").split("\t")
            if len(cols) < 9 or cols[2] not in TYPES:
                continue
            try:
                start, end = int(cols[3]), int(cols[4])
            except ValueError:
                continue
            out.append((cols[0], cols[2], start, end, cols[8]))
    return out


def parse_annotation_info(path):
    """Phytozome annotation_info.txt -> {locusName: (best-hit-arabi-name, arabi-defline)}.  Columns are
    taken by name from a '#pacId ...' header when there is one, else by the usual positions."""
    names = ["pacId", "locusName", "transcriptName", "peptideName", "Pfam", "Panther", "KOG", "KEGG/ec", "KO", "GO",
             "Best-hit-arabi-name", "arabi-symbol", "arabi-defline"]
    info = {}
    with open(path, "r") as f:
        for line in f:
            cols = line.rstrip("\n").split("\t")
            if line.startswith("#"):
                head = [c.lstrip("#") for c in cols]
                if "locusName" in head:
                    names = head
                continue
            if len(cols) < 2:
                continue
            rec = dict(zip(names, cols))
            locus = rec.get("locusName", "")
            if locus and locus not in info:
                info[locus] = (rec.get("Best-hit-arabi-name", ""), rec.get("arabi-defline", ""))
    return info


def label_of(ftype, attributes, info=None):
    attrs = dict(_ATTR.findall(attributes))
    ident = attrs.get("ID") or attrs.get("Name") or attrs.get("Parent") or "."
    label = "%s:%s" % (ftype, ident)
    if info and ftype == "gene":
        hit = info.get(attrs.get("Name", "")) or info.get(attrs.get("ID", ""))
        if hit:
            label += "".join("|" + x for x in hit if x)
    return label


class _Track:
    """One seqid: sorted cut points and the string id of every elementary interval."""

    def __init__(self, feats, strings, string_ids):
        # feats: [(start, end, order, label)]
        points = sorted(set([s for s, _, _, _ in feats] + [e + 1 for _, e, _, _ in feats]))
        self.points = np.asarray(points, dtype=np.int64)
        by_start = sorted(feats, key=lambda t: t[0])
        ids = np.full(len(points), NO_FEATURE, dtype=np.uint32)
        active, nxt = [], 0
        for i, x in enumerate(points):
            while nxt < len(by_start) and by_start[nxt][0] <= x:
                active.append(by_start[nxt])
                nxt += 1
            active = [t for t in active if t[1] >= x]
            if active:
                labels = []
                for t in sorted(active, key=lambda t: t[2]):  # GFF file order
                    if t[3] not in labels:
                        labels.append(t[3])
                text = ";".join(labels)
                k = string_ids.get(text)
                if k is None:
                    k = string_ids[text] = len(strings)
                    strings.append(text)
                ids[i] = k
        self.ids = ids

    def lookup(self, x):
        """string id per coordinate (numpy int64 array)."""
        i = np.searchsorted(self.points, x, "right") - 1
        out = np.full(x.shape, NO_FEATURE, dtype=np.uint32)
        ok = i >= 0
        out[ok] = self.ids[i[ok]]
        return out


class Annotation:
    def __init__(self, gff_path, phytozome_path=None):
        info = parse_annotation_info(phytozome_path) if phytozome_path else None
        per_seq = {}
        for order, (seqid, ftype, start, end, attrs) in enumerate(parse_gff(gff_path)):
            per_seq.setdefault(seqid, []).append((start, end, order, label_of(ftype, attrs, info)))
        self.strings, ids = [], {}
        self.tracks = {seqid: _Track(feats, self.strings, ids) for seqid, feats in per_seq.items()}

    def for_contig(self, key, hits, guide_len, dec, contig_len):
        """(strings, idx): idx[k] = entry of `strings` for row k of the contig (rows in the reference's
        order: '+' hits, then '-' hits), NO_FEATURE for rows without a feature or without a cut site."""
        ip = np.asarray(hits["pos_plus"]).astype(np.int64)
        jm = np.asarray(hits["pos_minus"]).astype(np.int64)
        l = int(guide_len)
        # a row has a cut site iff its long_sequence has 30 characters (CROPSR.py:466): Python clamps
        # the slice at the end of the string
        full_p = np.minimum(ip + 5, contig_len) - (ip - l - 5) == 30
        full_m = np.minimum(jm + 3 + l + 5, contig_len) - (jm - 2) == 30
        cut = np.concatenate([ip - 3, jm])            # end_pos - 3 (CROPSR.py:157): '+' end = i, '-' end = j + 3
        full = np.concatenate([full_p, full_m])
        idx = np.full(cut.shape, NO_FEATURE, dtype=np.uint32)
        track = self.tracks.get(contig_name(key))
        if track is not None and cut.size:
            idx[full] = track.lookup(cut[full] - dec + 1)
        return self.strings, idx
