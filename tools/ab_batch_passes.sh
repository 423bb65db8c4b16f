#!/bin/bash
# f1, round 6: the CLI's format + write stage under --each-contig-once with the passes handed to the formatter one call per
# pass (CROPSR_BATCH_PASSES=0) or together up to CROPSR_BATCH_ROWS rows -- format_write_s, the CLI's total and the CSV's md5
# (equal in every mode).  GPU box, repo root:  bash tools/ab_batch_passes.sh   (profiles/EXPERIMENTS.md round 6)
run() { python3 tools/e2e_cli.py switchgrass --md5 | python3 -c "
import sys,json; e=json.loads(sys.stdin.readline()); p=e['phases']; print('$1', e['cli_wall_s'], round(p['format_write_s'],3), round(p['total_s'],3), e.get('md5'))"; }
for i in 1 2; do
CROPSR_BATCH_PASSES=0 run one_call_per_pass
CROPSR_BATCH_ROWS=1000000 run together_1M
run together_4M_default
CROPSR_BATCH_ROWS=8000000 run together_8M
CROPSR_BATCH_ROWS=100000000 run everything_in_one_call
done
