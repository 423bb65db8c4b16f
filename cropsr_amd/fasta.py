"""FASTA text -> the contig table the reference iterates over.

Host-side mirror of import_fasta_file (reference CROPSR.py:54-74) and the two
helpers it calls, formatted (cropsr_functions.py:221-229) and
generate_dictionary (cropsr_functions.py:190-196).  The reference does not parse
FASTA records: it prints a Python list of (header, sequence) tuples with str(),
splits that text on whitespace and pairs the tokens up.  Every coordinate and the
`chromosome` column of the output depend on the by-products of that:

  * keys look like  [('Chr01',   ('c2',   and values like  'ACGT...'),  'ACGT...')]
    -- the quote/paren/comma characters are part of the strings that get scanned
    (positions are therefore 1 higher than 0-based genome offsets, and the
    decoration leaks into long_sequence near contig ends);
  * a header containing blanks shifts the pairing;
  * a FASTA that is already "two lines per record, no final newline" skips the
    re-formatting and is split as is (keys keep their '>').

This module reproduces that table for any input by building the same printed
form.  (A streaming packer that avoids the printed copy for Gb genomes is the
f2 row of SURVEY.md section 8.)
"""
from itertools import zip_longest


def needs_formatting(text):
    """CROPSR.py:62-63: anything but 2 lines per record without a final newline."""
    return 2 * text.count(">") != text.count("\n") + 1


def printed_records(text):
    """The text cropsr_functions.formatted returns: str() of the record list.

    A record is everything between two '>' (empty pieces dropped); it is split
    once at its first newline into (header, body) and newlines are removed from
    both parts.  A record without any newline yields a 1-tuple.
    """
    records = []
    for piece in text.split(">"):
        if piece:
            records.append(tuple(part.replace("\n", "") for part in piece.split("\n", 1)))
    return str(records)


def contig_table(text):
    """dict name_token -> sequence_string, insertion-ordered (CROPSR.py:70-71).

    Tokens are paired (1st, 2nd), (3rd, 4th), ...; an odd token count leaves the
    last key with "" as its value; a repeated key keeps its first position and
    takes the later value (dict semantics).
    """
    if needs_formatting(text):
        text = printed_records(text)
    tokens = text.split()
    return dict(zip_longest(tokens[0::2], tokens[1::2], fillvalue=""))


def load(path):
    with open(path, "r") as f:
        return contig_table(f.read())
