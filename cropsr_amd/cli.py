"""`python -m cropsr_amd` -- the CROPSR command line on top of the MI355X engine.

Same flags, defaults, stdout lines, time.txt and CSV bytes as the reference's
CROPSR.py (v1.11b); the PAM scan and the scoring run in libcropsr_hip.so
instead of re/str.replace/numpy.  Reference anchors are given per step.

Differences that are NOT observable in the outputs:
  * all contigs are uploaded and scanned in ONE arena pass before the per-contig
    output loop starts (the reference scans inside the loop, CROPSR.py:413-434);
  * the 5 s sleep per contig (CROPSR.py:478) is kept only with --reference-sleep.
"""
import argparse
import sys
import time
from multiprocessing import cpu_count

from . import fasta, rows

__version__ = "1.11b"  # the CROPSR version whose behaviour is reproduced

BANNER = r"""
################################################################################
##                                                                            ##
##                                                                            ##
##          .o88b.   d8888b.    .d88b.    d8888b.   .d8888.   d8888b.         ##
##         d8P  Y8   88  `8D   .8P  Y8.   88  `8D   88'  YP   88  `8D         ##
##         8P        88oobY'   88    88   88oodD'   `8bo.     88oobY'         ##
##         8b        88`8b     88    88   88ººº       `Y8b.   88`8b           ##
##         Y8b  d8   88 `88.   `8b  d8'   88        db   8D   88 `88.         ##
##          `Y88P'   88   YD    `Y88P'    88        `8888Y'   88   YD         ##
##                                                                            ##
##                                                                            ##
################################################################################
U.S. Dept. of Energy's Center for Advanced Bioenergy and Bioproducts Innovation
University of Illinois at Urbana-Champaign
"""


def build_parser():
    """The reference's flags (CROPSR.py:24-49) plus engine options that default
    to reference behaviour."""
    p = argparse.ArgumentParser(prog="CROPSR.py")
    p.add_argument("-f", "--fasta", metavar="", required=True, dest="f",
                   help="[required] path to input file in FASTA format")
    p.add_argument("-g", "--gff", metavar="", dest="g", help="path to input file in GFF format")
    p.add_argument("-p", "--phytozome", metavar="", dest="p", default=None,
                   help="path to input annotation info file in TXT format, default = None")
    p.add_argument("-o", "--output", metavar="", dest="o", default="data.csv",
                   help="path to output file, default = data.csv")
    p.add_argument("-l", "--length", metavar="", dest="l", type=int, default=20,
                   help="length of the gRNA se3quence, default = 20")
    p.add_argument("-L", "--flanking", metavar="", dest="L", type=int, default=200,
                   help="length of flanking region for verification, default = 200")
    p.add_argument("--cas9", action="store_true",
                   help="specifies that design will be made for the Cas9 CRISPR system")
    p.add_argument("-v", "--verbose", action="store_true",
                   help="prints visual indicators for each iteration")
    eng = p.add_argument_group("MI355X engine")
    eng.add_argument("--device", type=int, default=None,
                     help="HIP device index, default = 0 (LOCAL_RANK under torch.distributed.run)")
    eng.add_argument("--seed", type=int, default=None,
                     help="seed numpy's global RNG so crispr_id is reproducible (reference: unseeded)")
    eng.add_argument("--csv-writer", choices=["native", "python"], default="native",
                     help="native: multi-threaded C++ row formatter (same bytes); python: csv module like the reference")
    eng.add_argument("--each-contig-once", action="store_true",
                     help="NOT reference behaviour: write every contig's rows once instead of re-writing all "
                          "earlier contigs on every pass (the reference never clears Complete_dataset, "
                          "CROPSR.py:407, which makes multi-contig genomes quadratic)")
    eng.add_argument("--reference-sleep", action="store_true",
                     help="also reproduce the reference's 5 s pause per contig")
    return p


def import_gff_file(gff, verbose, out=sys.stdout):
    """Side effects of CROPSR.py:77-95.  The table is parsed exactly as the
    reference parses it and, exactly as there, never used."""
    import pandas as pd
    start_index = 0
    with open(gff, "r") as raw:
        if verbose:
            print(f"Annotation file {gff} successfully imported", file=out)
        lines = raw.readlines()
        for index, line in enumerate(lines):
            if "##" not in line:
                start_index = index
                break
    cols = ["chromosome", "source", "feature", "start", "end", "score", "strand", "phase", "attributes"]
    table = pd.read_csv(gff, sep="\t", skiprows=start_index, header=None, names=cols)
    if verbose:
        print("Annotation database successfully generated", file=out)
    return table


class EngineBackend:
    """The product's hit provider: libcropsr_hip.so on one MI355X."""

    def __init__(self, device=0):
        from .engine import Engine
        self.engine = Engine(device)

    def scan(self, contig_strings, guide_len):
        """One arena pass on the GPU for all contig strings (seam 1 + 2)."""
        genome = self.engine.genome(contig_strings)  # as many arenas as the genome needs
        hits = genome.scan_score(guide_len, want_pre=False)
        out = [hits.contig(k) for k in range(len(contig_strings))]
        genome.close()
        return out

    def scan_tables(self, texts, guide_len):
        """The same scan with the tables left in HBM, for parallel.sharded_scan: torch views of the
        library's device tables (zero copy), the arena layout, and a release callback."""
        from . import _native as nat
        from . import parallel
        L = nat.lib()
        need = sum(int(L.crp_arena_words_for(len(t))) for t in texts)
        if int(L.crp_arena_words_total(need)) > int(L.crp_arena_max_words()):
            raise ValueError("this rank's share of the genome (%d characters) does not fit one arena of 2^31 "
                             "characters: run on more GPUs" % sum(len(t) for t in texts))
        arena = self.engine.arena(texts)
        n_plus, n_minus = arena.scan_score_device(guide_len, want_pre=False)
        tables = parallel.device_tables_as_tensors(arena, n_plus, n_minus)
        return tables, list(zip(arena.offsets, arena.lengths)), arena.close

    def rescore(self, rows_u8, order):
        """Seam 2 on a few rows in one of the BLAS tail orders (rows.Dataset.rows)."""
        return self.engine.score_30mers(rows_u8, order)[1]

    def close(self):
        self.engine.close()


def _distributed():
    """(rank, world) when launched by torch.distributed.run with more than one process, else None.
    One process per GPU; the process group is RCCL (backend "nccl") unless CROPSR_DIST_BACKEND
    names another one (gloo: CPU rehearsals and the tests)."""
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return None
    import torch  # noqa: F401  (before the engine: PyTorch-ROCm brings its own HIP runtime)
    import torch.distributed as dist
    if not dist.is_initialized():
        name = os.environ.get("CROPSR_DIST_BACKEND", "nccl")
        if name == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(name)
    return dist.get_rank(), dist.get_world_size()


def run(args, backend=None, out=sys.stdout):
    """main() of the reference (CROPSR.py:333-486) with the hot path swapped out.

    `backend` provides scan(contig_strings, guide_len) -> [hits dict per contig]
    and rescore(rows_u8, order) -> scores; it defaults to the HIP engine.  The CPU
    test-suite injects the oracle here to pin this host logic against the golden
    CSVs without a GPU.
    """
    begin = time.time()
    if not args.cas9:
        sys.exit("Please select at least one CRISPR system: Cas9")  # CROPSR.py:335-336
    shard = _distributed()
    if shard is not None and shard[0] != 0:
        # Ranks other than 0 of a multi-GPU run (python -m torch.distributed.run ... -m cropsr_amd):
        # read the same FASTA, scan their share of the contigs, hand the tables to rank 0 -- which
        # alone prints, draws ids and writes files -- and leave.
        import os
        from . import parallel
        table = fasta.table_from_bytes(fasta.read_text_bytes(args.f))
        own_backend = backend is None
        if own_backend:
            device = getattr(args, "device", None)
            backend = EngineBackend(int(os.environ.get("LOCAL_RANK", "0")) if device is None else device)
        parallel.sharded_scan(backend, [v for _, v in table], args.l,
                              max_piece=int(os.environ.get("CROPSR_DIST_MAX_PIECE", "0")) or None)
        if own_backend:
            backend.close()
        return
    verbose = args.verbose
    if verbose:
        print(BANNER + f"""
        You are currently utilizing the following settings:

        CROPSR version:                                 {__version__}
        Path to genome file in FASTA format:            {args.f}
        Path to output file:                            {args.o}
        Length of the gRNA sequence:                    {args.l}
        Length of flanking region for verification:     {args.L}
        Number of available CPUs:                       {cpu_count()}
        Path to annotation file in GFF format:          {args.g}
        Path to annotation_info file in TXT format:     {args.p}
        Designing for CRISPR system:
            Streptococcus pyogenes Cas9                 {args.cas9}
        """, file=out)

    timing = open("time.txt", "w")  # CROPSR.py:371 (CWD side effect, kept)

    # CROPSR.py:374, 54-74 -- read as text mode would (universal newlines), kept as bytes
    data = fasta.read_text_bytes(args.f)
    if verbose:
        print(f"Genome file {args.f} successfully imported", file=out)
        if 2 * data.count(b">") != data.count(b"\n") + 1:
            print("formatting genome", file=out)
            print(f"Genome file {args.f} successfully formatted", file=out)
    table = fasta.table_from_bytes(data)  # == fasta.contig_table(text).items(), without printing the genome
    del data
    if verbose:
        print("The genome was successfully converted to a dictionary", file=out)
    import_gff_file(args.g, verbose, out)  # CROPSR.py:375 (raises like the reference if -g is missing)

    if verbose:
        # (the blank line in the middle carries the reference's 12 blanks of indentation)
        print("\n            Initiating PAM site detection.\n            \n"
              "            Please wait, this may take a while...\n            ", file=out)

    if getattr(args, "seed", None) is not None:
        import numpy as np
        np.random.seed(args.seed)

    rows.write_header(args.o)  # CROPSR.py:402-405

    names = [k for k, _ in table]
    strings = [v for _, v in table]  # contig strings as bytes, one byte per character
    own_backend = backend is None
    if own_backend:
        import os
        device = getattr(args, "device", None)
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if shard is not None else 0
        backend = EngineBackend(device)
    if shard is None:
        all_hits = backend.scan(strings, args.l)  # seam 1 + 2 for every contig, one GPU pass
    else:  # contigs (cut where longer than a rank's share) over all GPUs, tables gathered here
        import os
        from . import parallel
        all_hits = parallel.sharded_scan(backend, strings, args.l,
                                         max_piece=int(os.environ.get("CROPSR_DIST_MAX_PIECE", "0")) or None)

    native = getattr(args, "csv_writer", "native") == "native"
    once = getattr(args, "each_contig_once", False)
    dataset = rows.NativeDataset() if native else rows.Dataset()  # Complete_dataset, CROPSR.py:407
    ids = None
    if native:
        # every pass draws its ids from one RNG stream, in order; the pass sizes are known now,
        # so a worker draws pass k+1's ids while pass k is formatted and written
        per_contig = [int(h["pos_plus"].size + h["pos_minus"].size) for h in all_hits]
        import numpy as np
        sizes = per_contig if once else np.cumsum(per_contig).tolist()
        ids = rows.IdStream(sizes, reverse=True)
    for name, s, hits in zip(names, strings, all_hits):
        print("Searching on Chromosome: ", name[:25], file=out)  # CROPSR.py:410-411
        print("With start of sequence: ", bytes(s[:25]).decode("latin-1"), file=out)
        block = rows.ContigTable(name, s, hits, args.l) if native else rows.ContigRows(name, bytes(s).decode("latin-1"), hits, args.l)
        if once:
            dataset = rows.NativeDataset() if native else rows.Dataset()  # opt-in fix of CROPSR.py:407
        dataset.append(block)
        if verbose:
            # CROPSR.py:436-439 counts regex matches BEFORE the keep-filter; the
            # engine reports kept hits, so re-count only for this message.
            import re
            n_sites = sum(1 for _ in re.finditer(rb"(?=.GG)", s)) + sum(1 for _ in re.finditer(rb"(?=CC.)", s))
            print(f"""
                {n_sites:n} Cas9 PAM sites were found on {name[1::]}
                """, file=out)
        if native:  # CROPSR.py:442-474
            rows.write_pass_native(args.o, dataset, backend.rescore, ids)
        else:
            rows.write_pass(args.o, dataset, backend.rescore)
        end = time.time()
        timing.write("Total runtime of the program is " + str(end - begin))  # CROPSR.py:477
        if getattr(args, "reference_sleep", False):
            time.sleep(5)  # CROPSR.py:478
    timing.close()
    if ids is not None:
        ids.close()
    if own_backend:
        backend.close()
    if verbose:
        print(f"The output file has been generated at {args.o}", file=out)


def main(argv=None):
    args = build_parser().parse_args(argv)
    run(args)


if __name__ == "__main__":
    main()
