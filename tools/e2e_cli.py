"""End-to-end timing of `python -m cropsr_amd` on a synthetic genome written as FASTA.

usage: python tools/e2e_cli.py {ecoli|tair10|sorghum|switchgrass[:scale]} [--reference-behaviour] [--profile] [--annotate N_GENES]
       python tools/e2e_cli.py real --fasta PATH [--gff PATH [--phytozome PATH]]      (a REAL genome: nothing is generated)
Writes the FASTA and the CSV under $TMPDIR (default /tmp), prints one JSON line.  --devices 0,1,..: the CLI's one-process
multi-GPU mode (the library's node handle).
"""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_fasta(workload, path, width=60):
    import numpy as np
    with open(path, "wb") as f:
        for spec in workload.specs:
            a = workload.bases(spec)
            f.write(b">" + spec.name.encode() + b"\n")
            full = (a.size // width) * width
            body = np.empty((a.size // width, width + 1), dtype=np.uint8)
            body[:, :width] = a[:full].reshape(-1, width)
            body[:, width] = 10
            f.write(body.tobytes())
            if a.size > full:
                f.write(a[full:].tobytes() + b"\n")


def file_md5(path):
    import hashlib
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--reference-behaviour", action="store_true", help="re-emit earlier contigs like the reference")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--procs", type=int, default=1,
                    help="run the CLI as that many processes under torch.distributed.run (the launcher only; all on "
                         "device 0, tables over the host transport: a rehearsal of the multi-GPU mode on a one-GPU box)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--md5", action="store_true", help="print the CSV's md5 (to compare runs with different --procs)")
    ap.add_argument("--ablate-ids", action="store_true",
                    help="timing only (the CSV's id column is wrong): ids cost nothing -- how much of the CSV stage is the "
                         "one sequential MT19937 stream the reference's ids come from")
    ap.add_argument("--cli-flag", action="append", default=[], metavar="FLAG",
                    help="extra flag for the CLI, e.g. --cli-flag=--offtarget (repeatable)")
    ap.add_argument("--annotate", type=int, default=0, metavar="N_GENES",
                    help="write a seeded synthetic Phytozome-style GFF3 with that many gene models and an annotation_info "
                         "file (bench_workload.synthetic_annotation) and run the CLI with -g / -p / --annotate")
    ap.add_argument("--fasta", default=None, metavar="PATH", help="workload `real`: the genome to run on (SURVEY.md 8d)")
    ap.add_argument("--gff", default=None, metavar="PATH", help="with --fasta: its GFF3; the run then uses --annotate")
    ap.add_argument("--phytozome", default=None, metavar="PATH", help="with --gff: the annotation_info file")
    ap.add_argument("--devices", default=None, metavar="LIST", help="pass --devices LIST to the CLI (one process, several GPUs)")
    a = ap.parse_args()
    import bench_workload as bw
    name, _, scale = a.workload.partition(":")
    tmp = os.environ.get("TMPDIR", "/tmp")
    gff = os.path.join(tmp, "e2e.gff")
    out_csv = a.out or os.path.join(tmp, "e2e_out.csv")
    real = a.fasta is not None
    if real:
        class _Real:
            name = os.path.basename(a.fasta)
            n_bases = None
        wl, fa = _Real(), a.fasta
    else:
        wl = {"ecoli": bw.ecoli_like, "tair10": bw.tair10_like, "sorghum": bw.sorghum_like}.get(name)
        wl = wl() if wl else bw.switchgrass_like(scale=float(scale or 1.0))
        fa = os.path.join(tmp, "e2e_%s.fa" % wl.name)
    t0 = time.time()
    if not real:
        write_fasta(wl, fa)
    gff_rows = None
    extra = []
    if real and a.gff:
        gff = a.gff
        extra = (["-p", a.phytozome] if a.phytozome else []) + ["--annotate"]
    elif a.annotate and not real:
        info = os.path.join(tmp, "e2e_annotation_info.txt")
        gff_rows = bw.synthetic_annotation(wl, gff, info, n_genes=a.annotate)
        extra = ["-p", info, "--annotate"]
    else:
        with open(gff, "w") as f:
            f.write("##gff-version 3\n")
    t_gen = time.time() - t0
    argv = ["-f", fa, "-g", gff, "-o", out_csv, "--cas9", "--seed", "1"] + extra
    if not a.reference_behaviour:
        argv.append("--each-contig-once")
    argv += a.cli_flag
    if a.devices:
        argv += ["--devices", a.devices]
    if a.procs > 1:
        import subprocess
        # the CLI starts its own ranks (cropsr_amd/launch.py); all of them on device 0 here, tables over the host transport
        # (RCCL cannot put two ranks on one GPU).  Rank 0 writes the stage timings (--bench-json).
        stages_json = os.path.join(tmp, "stages_procs.json")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CROPSR_LAUNCHED")}
        env.update(CROPSR_GATHER="host", PYTHONPATH=ROOT)
        cmd = [sys.executable, "-m", "cropsr_amd", "--gpus", str(a.procs)] + argv + ["--device", "0", "--bench-json", stages_json]
        t0 = time.time()
        p = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True)
        wall = time.time() - t0
        if p.returncode != 0:
            sys.exit(p.stderr[-3000:])
        size = os.path.getsize(out_csv)
        print(json.dumps({"workload": wl.name, "procs": a.procs, "transport": "host sockets (ranks share GPU 0)", "csv_bytes": size,
                          "wall_incl_process_start_s": round(wall, 3), "phases": json.load(open(stages_json)),
                          "gff_gene_cds_rows": gff_rows, "md5": file_md5(out_csv) if a.md5 else None}))
        os.remove(out_csv)
        if not real:
            os.remove(fa)
        return
    from cropsr_amd import cli
    if a.ablate_ids:
        import numpy as np
        from cropsr_amd import rows as _rows
        _rows.draw_ids = lambda size, piece=1 << 20, reverse=False: np.full((size, 7), 65, dtype=np.uint8)
    stages_json = os.path.join(tmp, "stages.json")
    args = cli.build_parser().parse_args(argv + ["--bench-json", stages_json])
    os.chdir(tmp)
    sink = io.StringIO()
    prof = cProfile.Profile() if a.profile else None
    t0 = time.time()
    if prof:
        prof.enable()
    cli.run(args, out=sink)
    if prof:
        prof.disable()
    wall = time.time() - t0
    size = os.path.getsize(out_csv)
    with open(out_csv, "rb") as f:
        rows = sum(chunk.count(b"\n") for chunk in iter(lambda: f.read(1 << 24), b"")) - 1
    print(json.dumps({"workload": wl.name, "data": "real" if real else "synthetic", "bases": wl.n_bases, "rows": rows, "csv_bytes": size,
                      "fasta_write_s": round(t_gen, 2), "cli_wall_s": round(wall, 3),
                      "rows_per_s": round(rows / wall), "phases": json.load(open(stages_json)), "gff_gene_cds_rows": gff_rows,
                      "gff_bytes": os.path.getsize(gff),
                      "md5": file_md5(out_csv) if a.md5 else None}))
    if prof:
        s = io.StringIO()
        pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(30)
        print(s.getvalue())
    os.remove(out_csv)
    if not real:
        os.remove(fa)


if __name__ == "__main__":
    main()
