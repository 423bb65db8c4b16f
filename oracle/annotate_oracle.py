"""Brute-force restatement of the opt-in annotation join (cropsr_amd/annotate.py).

TEST INFRASTRUCTURE ONLY.  Parity unpinned: the reference parses the GFF and never uses it
(CROPSR.py:77-95, :375; `features` is the constant '' at :466, :468), so there is no reference output to
compare with.  This file states the join's definition as directly as possible -- per CSV row, a loop
over every line of the GFF -- so that the product's sweep (csrc/crp_annotation.cpp) + device look-up
(csrc/crp_annotate.hip) can be checked against it.  host_join() is a second, numpy statement of the look-up
alone, on the product's own interval tables: what the CPU tests' oracle backend uses in place of the GPU, and
what the whole-genome digests of the GPU tests are compared with.
"""
import re

import numpy as np

NO_FEATURE = 0xFFFFFFFF


def gff_rows(path):
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                continue
            c = line.rstrip("\n").split("\t")
            if len(c) >= 9 and c[2] in ("gene", "CDS") and _DIGITS(c[3]) and _DIGITS(c[4]):  # rows with unreadable coordinates join nothing
                yield c[0], c[2], int(c[3]), int(c[4]), c[8]


# The definition is stated on ASCII: "stripped" removes blank, tab, newline, carriage return, vertical tab and form feed
# (not str.strip()'s \x1c-\x1f or Unicode blanks), "digits" are 1 to 18 of 0-9 (not str.isdigit()'s Unicode digits) -- the
# same as cropsr_amd/csrc/crp_annotation.cpp and tests/native/annotation_driver.cpp.
_SPACE = " \t\n\r\v\f"
_DIGITS = re.compile(r"[0-9]{1,18}").fullmatch


def label(ftype, attrs, info=None):
    d = {}
    for part in attrs.split(";"):
        k, _, v = part.strip(_SPACE).partition("=")
        if k in ("ID", "Name", "Parent") and k not in d:
            d[k] = v
    ident = d.get("ID") or d.get("Name") or d.get("Parent") or "."
    out = ftype + ":" + ident
    if info and ftype == "gene":
        hit = info.get(d.get("Name", "")) or info.get(d.get("ID", ""))
        if hit:
            out += "".join("|" + x for x in hit if x)
    return out


def features_of_rows(csv_rows, fasta_name_of, dec, gff_path, info=None):
    """csv_rows: the reference's row tuples as lists of strings (12 or 11 fields).  fasta_name_of:
    chromosome column -> FASTA name.  Returns the `features` string of every row."""
    feats = list(gff_rows(gff_path))
    out = []
    for r in csv_rows:
        if len(r) != 12:          # no cut site
            out.append("")
            continue
        x = int(r[7]) - dec + 1   # cutsite column -> 1-based genome coordinate
        name = fasta_name_of(r[4])
        labels = []
        for seqid, ftype, start, end, attrs in feats:
            if seqid == name and start <= x <= end:
                lab = label(ftype, attrs, info)
                if lab not in labels:
                    labels.append(lab)
        out.append(";".join(labels))
    return out


def parse_info(path):
    """Phytozome annotation_info.txt -> {locusName: (Best-hit-arabi-name, arabi-defline)}; columns by name from a
    '#' header line that names locusName, else the usual 13 positions; the first line of a locus wins."""
    names = ["pacId", "locusName", "transcriptName", "peptideName", "Pfam", "Panther", "KOG", "KEGG/ec", "KO", "GO",
             "Best-hit-arabi-name", "arabi-symbol", "arabi-defline"]
    info = {}
    with open(path) as f:
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


def host_join(annotation, name, start, dec, hits, guide_len, text_len):
    """(feat_plus, feat_minus) for the hit dict of ONE text (positions local to the text): the label-set id of the
    elementary interval of seqid `name` each row's cut site falls in -- numpy searchsorted over the product's
    interval table (annotation.seq_track).  start: index of the text's first character inside its contig string
    (0 unless the text is a piece of a cut contig); dec: SURVEY.md A.1.  A row has a cut site iff its long_sequence
    has 30 characters (CROPSR.py:466; Python clamps the slice at the end of the string)."""
    l = int(guide_len)
    ip = np.asarray(hits["pos_plus"]).astype(np.int64)
    jm = np.asarray(hits["pos_minus"]).astype(np.int64)
    full = [np.minimum(ip + 5, text_len) - (ip - l - 5) == 30, np.minimum(jm + 3 + l + 5, text_len) - (jm - 2) == 30]
    cut = [ip - 3, jm]  # end_pos - 3 (CROPSR.py:157): '+' end = i, '-' end = j + 3
    track = annotation.seq_track(name)
    out = []
    for c, ok in zip(cut, full):
        idx = np.full(c.shape, NO_FEATURE, dtype=np.uint32)
        if track is not None and track[0].size and c.size:
            points, ids = track
            k = np.searchsorted(points, c + start - dec + 1, "right") - 1
            hit = ok & (k >= 0)
            idx[hit] = ids[k[hit]]
        out.append(idx)
    return out[0], out[1]
