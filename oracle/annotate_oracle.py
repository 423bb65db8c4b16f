"""Brute-force restatement of the opt-in annotation join (cropsr_amd/annotate.py).

TEST INFRASTRUCTURE ONLY.  Parity unpinned: the reference parses the GFF and never uses it
(CROPSR.py:77-95, :375; `features` is the constant '' at :466, :468), so there is no reference output to
compare with.  This file states the join's definition as directly as possible -- per CSV row, a loop
over every line of the GFF -- so that the product's sweep + binary search can be checked against it.
"""


def gff_rows(path):
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                continue
            c = line.rstrip("\n").split("\t")
            if len(c) >= 9 and c[2] in ("gene", "CDS") and c[3].isdigit() and c[4].isdigit():  # rows with unreadable coordinates join nothing
                yield c[0], c[2], int(c[3]), int(c[4]), c[8]


def label(ftype, attrs, info=None):
    d = {}
    for part in attrs.split(";"):
        k, _, v = part.strip().partition("=")
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
