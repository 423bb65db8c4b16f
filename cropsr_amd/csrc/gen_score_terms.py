#!/usr/bin/env python3
"""Build-time generator: doench_weights.def -> score_terms.inc (+ exp_table.inc).

Emits the fully unrolled gated-FMA body of the on-target score for the HIP
kernels (crp_score.h).  No weight table exists at run time: every non-zero
weight becomes one `v_and_b32` + one `v_fma_f64`.

Why an FMA, and why it is bit-exact.  The reference computes the score with two
numpy matmuls of one-hot rows against the weight vectors (CROPSR.py:304-311).
OpenBLAS evaluates each as four running sums, one per (flat index mod 4), i.e.
one per alphabet letter of the (second) base, adding terms in ascending index
order, and combines them as (l0+l2)+(l1+l3).  Adding a zero
term is exact, so only the 39+31 non-zero weights matter, each as
`acc = acc + (bit ? w : 0)`.  We evaluate that as fma(g, w', acc) where g is a
power of two or zero made directly from the mask bit: the masked bit is placed
in the exponent field of a double's high word (bits 20..30), giving
g = 2^(E-1023) with E = 1 << (bitpos-20), and w' = w * 2^(1023-E) is a
compile-time constant.  g*w' == w exactly (powers of two), the FMA rounds
once, so the result equals round(acc + w); when the bit is clear g = +0 and
the FMA returns acc unchanged.

Each 30-bit mask is pre-shifted into three copies so that every position p lands
on a high-word bit in 20..30:  a = m << 20 (p 0..10), b = m << 9 (p 11..21),
c = m >> 2 (p 22..29).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
IDX = {"A": 0, "T": 1, "C": 2, "G": 3}


def parse(path):
    consts, first, second = {}, [], []
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].split()
            if not line:
                continue
            if line[0] == "CONST":
                consts[line[1]] = float(line[2])
            elif line[0] == "FIRST":
                first.append((line[1], int(line[2]), float(line[3]), line[3]))
            elif line[0] == "SECOND":
                second.append((line[1], line[2], int(line[3]), float(line[4]), line[4]))
            else:
                raise ValueError(line)
    return consts, first, second


def dense(first, second):
    """Dense 120 / 464 tables in the reference's flat layout (for the tests)."""
    w1 = [0.0] * 120
    w2 = [0.0] * 464
    for b, pos, w, _ in first:
        w1[(pos - 1) * 4 + IDX[b]] = w
    for b1, b2, pos, w, _ in second:
        w2[(pos - 1) * 16 + IDX[b1] * 4 + IDX[b2]] = w
    return w1, w2


def copy_of(p):
    """(copy letter, high-word bit) for 0-based position p."""
    if p <= 10:
        return "a", p + 20
    if p <= 21:
        return "b", p + 9
    return "c", p - 2


def scaled(w, bit):
    e = 1 << (bit - 20)
    v = w * 2.0 ** (1023 - e)  # exact: power-of-two scaling, no overflow since |w| < 2
    assert v == v and abs(v) != float("inf") and v != 0.0
    return v.hex()


# ---------------------------------------------------------------------------------------------
# Table form of the PAM-variant scorer.  The value of an accumulation chain after its first k
# terms is a function of k gate bits only, so it is looked up instead of summed: the table entry
# for a bit pattern is init + (the weights of the set bits, added one by one in chain order, each
# addition rounded to double) -- exactly what the k gated FMAs would have produced.  The k gate
# bits are gathered into a table index with one multiplication: for the right constant the top k
# bits of (bits * MAGIC) are different for every one of the 2^k patterns (any one-to-one mapping
# will do, the table is laid out to match).  Terms past the first k of a chain stay gated FMAs.
# (sT: its three live terms as an 8-entry table; sG: 8 of its 14 terms.  Measured with the 1 024-word tile, where the LDS
# left over by three workgroups per CU is split between these tables and the hit list: profiles/EXPERIMENTS.md, round 3)
TABLE_BITS = {"fA": 8, "fT": 8, "fG": 7, "fC": 7, "sG": 8, "sC": 7, "sA": 6, "sT": 3}
# experiments: CRP_TABLE_BITS="fA=7,fT=7" trades table size (LDS) against gated FMAs
for _kv in os.environ.get("CRP_TABLE_BITS", "").split(","):
    if "=" in _kv:
        TABLE_BITS[_kv.split("=")[0].strip()] = int(_kv.split("=")[1])


def find_gather(qs, rng_seed):
    """bits at positions qs (chain order) of a 32-bit word -> (lo, magic, W): the table index of a
    bit pattern is the top k bits of ((word >> lo) * magic) mod 2^W, one-to-one on all 2^k patterns.
    W = 24 when the bits span fewer than 24 positions (v_mul_u32_u24, full rate), else 32."""
    import random

    import numpy as np
    k = len(qs)
    lo = min(qs)
    rel = np.array([q - lo for q in qs], dtype=np.uint64)
    W = 24 if int(rel.max()) < 24 else 32
    modmask = np.uint64((1 << W) - 1)
    rng = random.Random(rng_seed)
    patterns = np.array([[(s >> i) & 1 for i in range(k)] for s in range(1 << k)], dtype=np.uint64).T
    batch = 4096
    for _ in range(2000):
        magic = np.array([rng.getrandbits(W) for _ in range(batch)], dtype=np.uint64)
        contrib = (magic[:, None] << rel[None, :]) & modmask
        idx = ((contrib @ patterns) & modmask) >> np.uint64(W - k)
        idx.sort(axis=1)
        ok = (np.diff(idx.astype(np.int64), axis=1) != 0).all(axis=1) & (magic != 0)
        if ok.any():
            return lo, int(magic[np.nonzero(ok)[0][0]]), W
    raise RuntimeError("no gather constant found for bit positions %r" % (qs,))


def emit_table_scorer(out, live, init, wtable, emit_copies):
    import collections
    chains = collections.OrderedDict()
    for t in live:
        chains.setdefault(t[5], []).append(t)
    lines = []
    data = []  # table entries (doubles), all chains back to back
    leftover = []
    lines.append("const uint32_t nA = (mA) >> 1, nT = (mT) >> 1, nC = (mC) >> 1, nG = (mG) >> 1; (void)nA; (void)nT; (void)nC; (void)nG;")
    for cname in ("fA", "fT", "fC", "fG", "sA", "sT", "sC", "sG"):
        terms = chains.get(cname, [])
        k = min(len(terms), TABLE_BITS[cname])
        head, tail = terms[:k], terms[k:]
        leftover.extend(tail)
        if k == 0:
            continue  # the chain keeps its start value and is summed by gated FMAs only
        # where each head term's gate bit sits in the source word
        if cname[0] == "f":
            b = cname[1]
            qs = [t[1] // 4 for t in head]
            sel = sum(1 << q for q in qs)
            src = "((m%s) & 0x%xu)" % (b, sel)
        else:
            x = cname[1]
            by_b1 = collections.OrderedDict()
            for t in head:
                p, b1 = t[1] // 16, "ATCG"[(t[1] % 16) // 4]
                by_b1.setdefault(b1, []).append(p)
            taken, shift_of, qs_of = set(), {}, {}
            for b1, ps in by_b1.items():  # move a whole part up when one of its positions is taken
                for sh in range(0, 32):
                    moved = [p + sh for p in ps]
                    if max(moved) < 32 and not (set(moved) & taken):
                        break
                else:
                    raise RuntimeError("no room for the %s terms of chain %s" % (b1, cname))
                shift_of[b1] = sh
                taken |= set(moved)
                for p in ps:
                    qs_of[(p, b1)] = p + sh
            qs = [qs_of[(t[1] // 16, "ATCG"[(t[1] % 16) // 4])] for t in head]
            parts = []
            for b1, ps in by_b1.items():
                part = "((m%s) & n%s & 0x%xu)" % (b1, x, sum(1 << p for p in ps))
                parts.append("(%s << %d)" % (part, shift_of[b1]) if shift_of[b1] else part)
            src = "(" + " | ".join(parts) + ")"
        lo, magic, W = find_gather(qs, "%s-%d" % (cname, k))
        # table: entry for every bit pattern, at the index the kernel will compute for it
        base = len(data)
        entries = [None] * (1 << k)
        for pattern in range(1 << k):
            word = sum(1 << (qs[i] - lo) for i in range(k) if (pattern >> i) & 1)
            idx = ((word * magic) & ((1 << W) - 1)) >> (W - k)
            v = init[cname]
            for i in range(k):
                if (pattern >> i) & 1:
                    v = v + head[i][6]  # one rounding per addition, chain order
            assert entries[idx] is None
            entries[idx] = v
        data.extend(entries)
        word = src if lo == 0 else "(%s >> %d)" % (src, lo)
        mul = "__umul24(%s, 0x%xu)" % (word, magic) if W == 24 else "(%s * 0x%xu)" % (word, magic)
        lines.append("%s = crp_tab_at(score_tab, %d, (%s >> %d) & 0x%xu); /* %d terms: %s */"
                     % (cname, 8 * base, mul, W - k - 3, ((1 << k) - 1) << 3, k,
                        " ".join(t[2].split("/*")[1].split()[0] for t in head)))
    body = emit_copies(leftover) + lines
    for t in leftover:
        body.append(t[2].replace("@W@", "CRP_WS(%d)" % wtable.index(t[3])))
    out.append("/* PAM variant with chain-prefix tables: %d of %d terms looked up, %d gated FMAs left */"
               % (len(live) - len(leftover), len(live), len(leftover)))
    out.append("#define CRP_SCORE_TAB_N %d" % len(data))
    out.append("#define CRP_SCORE_TAB_DATA { \\")
    out.extend("    %s, \\" % float(v).hex() for v in data)
    out.append("    }")
    out.append("#define CRP_SCORE_BODY_PAM_TABLES(mA, mT, mC, mG) \\")
    out.extend("    %s \\" % b for b in body)
    out.append("    /* end */")


def generate(def_path):
    consts, first, second = parse(def_path)
    out = []
    used = set()
    terms = []  # (flat index, text)
    for b, pos, w, lit in first:
        p = pos - 1
        cp, bit = copy_of(p)
        used.add(("m" + b, cp))
        terms.append((0, p * 4 + IDX[b],
                      "CRP_TERM(f%s, m%s_%s, %2d, @W@) /* %s%02d %s */" % (b, b, cp, bit, b, pos, lit), scaled(w, bit),
                      {("m" + b, cp)}, "f" + b, w))
    # second order: pair mask per (pair, copy) = mB1_copy & nB2_copy, nY = mY shifted one
    # position down.  Measured on gfx950: a VOP2 v_and_b32 (also with a literal) issues in 2
    # cycles, any VOP3 (v_bitop3_b32) and any shift in 4 -- so two VOP2 ANDs, the first
    # shared by the terms of one (pair, copy), beat one three-input AND per term.
    pair_copies = set()
    for b1, b2, pos, w, lit in second:
        p = pos - 1
        cp, bit = copy_of(p)
        used.add(("m" + b1, cp))
        used.add(("n" + b2, cp))
        pair_copies.add((b1, b2, cp))
        terms.append((1, p * 16 + IDX[b1] * 4 + IDX[b2],
                      "CRP_TERM(s%s, p%s%s_%s, %2d, @W@) /* %s%s%02d %s */"
                      % (b2, b1, b2, cp, bit, b1, b2, pos, lit), scaled(w, bit),
                      {("m" + b1, cp), ("n" + b2, cp), (b1, b2, cp)}, "s" + b2, w))
    out.append("/* generated by gen_score_terms.py from doench_weights.def -- do not edit */")
    out.append("#define CRP_INTERSECT %s" % consts["intersect"].hex())
    out.append("#define CRP_LOW_GC %s" % consts["low_gc"].hex())
    m_shift = {"a": "<< 20", "b": "<< 9", "c": ">> 2"}
    n_shift = {"a": "<< 19", "b": "<< 8", "c": ">> 3"}  # (m >> 1) moved like the m copy; stray low bit lands below bit 20
    ordered = sorted(terms, key=lambda t: (t[1] // (16 if t[0] else 4), t[0], t[1]))

    def emit_copies(live):
        """shifted mask copies and pair masks the gated terms in `live` read"""
        body = []
        need = set()
        for t in live:
            need.update(t[4])
        for name, cp in sorted(n for n in need if len(n) == 2):
            body.append("const uint32_t %s_%s = (m%s) %s;" % (name, cp, name[1], (m_shift if name[0] == "m" else n_shift)[cp]))
        for b1, b2, cp in sorted(n for n in need if len(n) == 3):
            body.append("const uint32_t p%s%s_%s = m%s_%s & n%s_%s;" % (b1, b2, cp, b1, cp, b2, cp))
        return body

    def emit_body(macro, keep, table):
        live = [t for t in ordered if keep(t)]
        body = emit_copies(live)
        # interleave first- and second-order terms by position: eight independent chains
        for t in live:
            body.append(t[2].replace("@W@", "CRP_WS(%d)" % table.index(t[3])))
        out.append("#define %s(mA, mT, mC, mG) \\" % macro)
        out.extend("    %s \\" % b for b in body)
        out.append("    /* end */")

    table = [t[3] for t in ordered]
    emit_body("CRP_SCORE_BODY", lambda t: True, table)

    # Variant for windows found by the PAM scan with l = 20: both strands bring the
    # window to [2 flank][C C N][protospacer][5 flank], so t[2] == t[3] == 'C' always
    # (SURVEY.md A.3).  Terms that test position 3 or 4 (1-based) are therefore constant:
    # the ones on 'C' always fire and are folded into the start value of their chain
    # (same additions in the same order, done here in IEEE double), the others never fire.
    init = {c: 0.0 for c in ("fA", "fT", "fC", "fG", "sA", "sT", "sC", "sG")}
    settled = {}

    def pam_state(t):
        order, flat = t[0], t[1]
        if order == 0:
            p, b = flat // 4, "ATCG"[flat % 4]
            if p in (2, 3):
                return "always" if b == "C" else "never"
            return "live"
        p, b1, b2 = flat // 16, "ATCG"[(flat % 16) // 4], "ATCG"[flat % 4]
        known = {2: "C", 3: "C"}
        r1 = None if p not in known else (known[p] == b1)
        r2 = None if p + 1 not in known else (known[p + 1] == b2)
        if r1 is False or r2 is False:
            return "never"
        if r1 is True and r2 is True:
            return "always"
        if r1 is None and r2 is None:
            return "live"
        raise ValueError("half-determined pair term: not handled by the generator")

    seen_live = set()
    for t in ordered:
        chain = t[5]
        st = pam_state(t)
        settled[id(t)] = st
        if st == "always":
            assert chain not in seen_live, "constant term after a live one would reorder the chain"
            init[chain] = init[chain] + t[6]
        elif st == "live":
            seen_live.add(chain)
    # (the all-gated-FMA form of the PAM variant -- what the tables are derived from -- is no longer emitted: it lives on
    # as the executable specification in tests/test_host.py::test_score_tables_equal_sequential_sums)
    emit_table_scorer(out, [t for t in ordered if settled[id(t)] == "live"], init, table, emit_copies)
    for c in sorted(init):
        out.append("#define CRP_PAM_INIT_%s %s" % (c, float(init[c]).hex()))
    # the pre-scaled weights in evaluation order
    out.append("#define CRP_WS_TABLE { \\")
    out.extend("    %s, \\" % w for w in table)
    out.append("    }")
    out.append("#define CRP_WS_COUNT %d" % len(table))
    return "\n".join(out) + "\n", dense(first, second), consts


def main():
    text, (w1, w2), _ = generate(os.path.join(HERE, "doench_weights.def"))
    with open(os.path.join(HERE, "score_terms.inc"), "w") as f:
        f.write(text)
    # dense copies in the reference's flat layout, for the generic-order scorer
    # (crp_score_generic.h) that reproduces the BLAS tail / single-row orders
    with open(os.path.join(HERE, "dense_weights.inc"), "w") as f:
        f.write("/* generated by gen_score_terms.py from doench_weights.def -- do not edit */\n")
        f.write("__device__ const double CRP_W1[120] = {\n%s};\n" % "".join("    %s,\n" % v.hex() for v in w1))
        f.write("__device__ const double CRP_W2[464] = {\n%s};\n" % "".join("    %s,\n" % v.hex() for v in w2))
    with open(os.path.join(HERE, "exp_table.inc"), "w") as f:
        f.write("/* generated by gen_score_terms.py -- 2^(k/128), k = 0..127, as {tail, scale-bits} pairs */\n")
        for t, h in exp2_table():
            f.write("0x%016xULL, 0x%016xULL,\n" % (t, h))


def exp2_table(n=128):
    """The constants of the table-driven exp the kernel evaluates (crp_score.h):
    2^(k/n) = H*(1+T) with H the nearest double and T the nearest double to the
    relative remainder; stored as bits(T), bits(H) - (k << 52)/n.  Computed with
    80-digit decimals (mathematical constants, no binary is read)."""
    import struct
    from decimal import Decimal, getcontext
    getcontext().prec = 80
    ln2 = Decimal(2).ln()
    bits = lambda x: struct.unpack("<Q", struct.pack("<d", x))[0]
    rows = []
    for k in range(n):
        v = (ln2 * Decimal(k) / Decimal(n)).exp()
        h = float(v)
        t = float((v - Decimal(h)) / Decimal(h))
        rows.append((bits(t), (bits(h) - ((k << 52) // n)) & (2 ** 64 - 1)))
    return rows


if __name__ == "__main__":
    main()
