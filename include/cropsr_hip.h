/*
 * cropsr_hip.h -- C ABI of libcropsr_hip.so, the MI355X (gfx950) PAM-scan +
 * on-target-score engine that replaces CROPSR's inner loop.
 *
 * The reference (H2muller/CROPSR) is one Python script and has no FFI of its
 * own; the two seams this library slots into are
 *
 *   seam 1  CROPSR.py:413-434  per contig string: regex (?=.GG) / (?=CC.) scan,
 *           window slicing and keep-filter that append to Complete_dataset
 *           -> crp_arena_* + crp_scan_score + crp_fetch_hits
 *   seam 2  CROPSR.py:285-313, called at :461  rs1_score(ndarray[n,30] uint8)
 *           -> ndarray[n] float64      -> crp_score_30mers
 *   seam 3  (opt-in) CROPSR.py:77-95 parses the GFF, :375 drops the table, :466-468 write '' into `features`: the join
 *           those lines stop short of -> crp_annotation_build / _track (host) + crp_annotate_set_track / _lookup (GPU)
 *
 * The binding a CROPSR maintainer would add (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every function returns 0 (CRP_OK) or a negative
 *     crp_status; nothing throws, aborts or prints.
 *   - the caller owns every host buffer; the library owns every device buffer.
 *   - one crp_ctx per GPU; calls on one handle are serialised by the caller.  Two ways to the whole node: one
 *     process per GPU, each with its own crp_ctx and an RCCL communicator across them (crp_comm_*), or ONE process
 *     with a crp_node over N devices (crp_node_*: the fan-out, the cut of the genome and the gatherv happen inside
 *     the library) -- what the single-process reference script binds.
 *   - there is NO CPU fallback: without a usable HIP device crp_init fails.
 *
 * Coordinates.  A contig is handed over as the exact character string the
 * reference scans (CROPSR.py:409 `sequence`, including the decoration left by
 * cropsr_functions.py:221-229 when the FASTA was re-formatted), one byte per
 * character.  Hit positions come back as indices into that string:
 *   '+' table: i = re match index of (?=.GG)   start_pos=i-l end_pos=i cutsite=i-3
 *   '-' table: j = re match index of (?=CC.)   start_pos=j+3+l end_pos=j+3 cutsite=j
 * shifted by the contig's arena offset (see crp_arena_add_contig_*).
 */
#ifndef CROPSR_HIP_H
#define CROPSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRP_ABI_VERSION 6

typedef enum crp_status {
    CRP_OK = 0,
    CRP_ERR_INVALID = -1,     /* bad argument */
    CRP_ERR_NO_DEVICE = -2,   /* no usable HIP device: none, or not a gfx950 (never falls back to CPU) */
    CRP_ERR_HIP = -3,         /* a HIP runtime call failed; see crp_last_error */
    CRP_ERR_NOMEM = -4,       /* host or device allocation failed */
    CRP_ERR_STATE = -5,       /* call out of order (e.g. scan before seal) */
    CRP_ERR_CAPACITY = -6,    /* arena capacity exceeded */
    CRP_ERR_UNSUPPORTED = -7, /* e.g. guide length outside [0, 50] */
    CRP_ERR_IO = -8,          /* write(2) on the caller's descriptor failed; errno is left set */
    CRP_ERR_COMM = -9,        /* an RCCL call failed or librccl.so could not be loaded; see crp_last_error */
    CRP_ERR_PEER = -10        /* a collective call was abandoned ON EVERY RANK because another rank reported an error
                               * before the exchange started (crp_gather_hits); this rank itself was fine */
} crp_status;

typedef struct crp_ctx crp_ctx;
typedef struct crp_arena crp_arena;

/* ---- library ----------------------------------------------------------- */
int crp_abi_version(void);
const char *crp_strerror(int status);

/* ---- context ----------------------------------------------------------- */
/* Opens HIP device `device_id` (index as seen by this process). */
int crp_init(int device_id, crp_ctx **out);
int crp_destroy(crp_ctx *ctx);
/* Text of the last failing HIP call on this context ("" if none). */
const char *crp_last_error(const crp_ctx *ctx);
/* Name / CU count / HBM bytes of the device, for logs and bench output. */
int crp_device_info(const crp_ctx *ctx, char *name, int name_cap, int *n_cu, uint64_t *hbm_bytes);

/* ---- host-side packing (no GPU needed) --------------------------------- */
/* Number of 64-base words one contig of `len` characters occupies in an arena
 * (its bases, padded to a word, plus one separator word). */
uint64_t crp_arena_words_for(uint64_t len);
/* Arena words needed for contigs whose crp_arena_words_for() sum is `sum`
 * (adds the leading and trailing separator words). */
uint64_t crp_arena_words_total(uint64_t sum);
/* Largest capacity_words crp_arena_create accepts (arena positions stay below 2^31);
 * a bigger genome is spread over several arenas, contig by contig. */
uint64_t crp_arena_max_words(void);
/* Classify `len` characters into the four bit-planes the kernels read, 64
 * characters per word, bit k of word w = character 64*w + k:
 *   hi,lo  2-bit base code, alphabet order of CROPSR.py:300 (A=00 T=01 C=10 G=11)
 *   up     upper-case base: complemented on the '+' strand (CROPSR.py:128 only
 *          maps upper case) and eligible for the PAM (regexes match 'G'/'C' only)
 *   ac     a scoring base (acgtACGT; 'U' counts as 'A')
 * Characters past `len` in the last word are encoded as "void" (outside any
 * contig).  Each plane needs ceil(len/64) words.  n_threads <= 1: serial. */
int crp_pack_ascii(const uint8_t *text, uint64_t len, uint64_t *hi, uint64_t *lo,
                   uint64_t *up, uint64_t *ac, int n_threads);

/* ---- arena: the device-resident genome --------------------------------- */
/* An arena holds any number of contigs as four bit-planes in HBM, separated by
 * void words, so one kernel launch scans all of them.  capacity_words bounds
 * the total (use crp_arena_words_total). */
int crp_arena_create(crp_ctx *ctx, uint64_t capacity_words, crp_arena **out);
int crp_arena_destroy(crp_arena *arena);
/* Append one contig given as characters; packing runs on the GPU.  Returns the
 * arena offset (in characters) of its first character: a hit at arena position
 * P belongs to the contig with the largest offset <= P, at index P - offset. */
int crp_arena_add_contig_ascii(crp_arena *arena, const uint8_t *text, uint64_t len,
                               uint64_t *arena_offset);
/* n contigs in one call, in order (arena_offsets: n values, may be NULL).  Same result as n calls of
 * crp_arena_add_contig_ascii; small contigs (an assembly's scaffolds) share one host-to-device copy and one pack
 * launch per ~31 MiB instead of paying a copy, a launch and an event each.  On an error the contigs before the failing
 * one have been added. */
int crp_arena_add_contigs_ascii(crp_arena *arena, const uint8_t *const *texts, const uint64_t *lens, uint64_t n,
                                uint64_t *arena_offsets);
/* Same, from planes packed on the host with crp_pack_ascii. */
int crp_arena_add_contig_packed(crp_arena *arena, const uint64_t *hi, const uint64_t *lo,
                                const uint64_t *up, const uint64_t *ac, uint64_t len,
                                uint64_t *arena_offset);
/* No more contigs; waits for the uploads. */
int crp_arena_seal(crp_arena *arena);
/* The tile shape the arena was sealed with (1 LARGE, 2 SMALL: CRP_OPT_TILE_GEOMETRY), its number of tiles
 * (= workgroups per scan) and the words of one tile.  Any pointer may be NULL. */
int crp_arena_tiles(const crp_arena *arena, int *geometry, uint64_t *n_tiles, uint64_t *tile_words);
/* Totals: contigs, characters, words used. */
int crp_arena_stats(const crp_arena *arena, uint64_t *n_contigs, uint64_t *n_chars, uint64_t *n_words);

/* What the arena's characters are, counted on the GPU from the planes: upper-case A/C/G/T, and everything else
 * (lower case, N, IUPAC codes, the decoration characters of cropsr_functions.py:221-229).  An arena whose n_other is
 * nothing but decoration (<= 4 per contig) is the "entirely upper-case ACGT" input for which SURVEY.md 8(d) leaves the
 * two 1-bit planes out of the algorithmic bytes (bench.py).  Either pointer may be NULL. */
int crp_arena_composition(crp_arena *arena, uint64_t *n_plain, uint64_t *n_other);

/* ---- seam 1 + 2 over a whole arena -------------------------------------- */
/* Scan both strands of every contig, keep what CROPSR.py:419/:430 keep for
 * guide length `guide_len` (0..50; the reference takes any integer: for lengths outside that range scan with the
 * nearer end of it -- a superset of the reference's hits, none of them scored at those lengths -- and apply
 * CROPSR.py:419/:430 to the positions on the host, as cropsr_amd/cli.py refilter_hits does), score every kept hit whose long_sequence has exactly
 * 30 characters (guide_len == 20: complete windows; > 20: only windows the end of
 * the string cuts to 30; < 20: none -- the others get -1 like CROPSR.py:466-468), leave
 * the tables in HBM.  Tables are ascending in arena position per strand, i.e.
 * per contig in the reference's own order.  flags: CRP_SCAN_PRE also keeps the
 * pre-sigmoid sum (CROPSR.py:312). */
#define CRP_SCAN_PRE 1    /* (flags == 1 is what callers of ABI version 2 passed as `want_pre`) */
/* CRP_SCAN_SEEDS: the scan also writes, per kept hit, the 12 characters of `sequence` next to the PAM that the
 * off-target seed scan (below) works on -- the emit kernel holds every hit's window in registers anyway.
 * crp_offtarget_add then takes them from there instead of reading the planes a second time.  Honoured for
 * guide_len == 20; for other lengths the flag is ignored and crp_offtarget_add derives the seeds itself. */
#define CRP_SCAN_SEEDS 2
int crp_scan_score(crp_arena *arena, int guide_len, int flags,
                   uint64_t *n_plus, uint64_t *n_minus);
/* Copy the tables of the last crp_scan_score to host arrays sized n_plus /
 * n_minus.  Any pointer may be NULL to skip that column. */
int crp_fetch_hits(crp_arena *arena, uint32_t *pos_plus, double *pre_plus, double *score_plus,
                   uint32_t *pos_minus, double *pre_minus, double *score_minus);
/* Rows of the '+' and '-' tables of the last crp_scan_score (what it returned in *n_plus / *n_minus). */
int crp_hits_counts(const crp_arena *arena, uint64_t *n_plus, uint64_t *n_minus);
/* Device addresses of the same tables (valid until the next crp_scan_score or
 * crp_arena_destroy) for a device-to-device gather such as RCCL send/recv. */
int crp_hits_device(crp_arena *arena, void **pos_plus, void **score_plus,
                    void **pos_minus, void **score_minus);

/* ---- seam 2 alone -------------------------------------------------------- */
/* Floating-point accumulation order of the two matmuls inside rs1_score
 * (CROPSR.py:305,311).  With the reference's BLAS the order a row is summed in
 * depends on its place in the batch of n rows handed to rs1_score:
 *   BODY4  rows 0 .. 4*floor(n/4)-1, and the last row if n%4 is 1 or 3
 *   TAIL2  rows 4*floor(n/4) and 4*floor(n/4)+1 if n%4 is 2 or 3
 *   DOT1   the single row of a batch with n == 1
 * crp_scan_score always uses BODY4; a host that wants the reference's CSV bytes
 * re-scores the <= 2 TAIL2 rows (or the DOT1 row) of each written chunk. */
#define CRP_ORDER_BODY4 0
#define CRP_ORDER_TAIL2 1
#define CRP_ORDER_DOT1 2
/* rs1_score (CROPSR.py:285-313) on n rows of 30 bytes, row-major, exactly the
 * array CROPSR.py:458-461 builds: bytes equal to 'A','T','C','G' select weights,
 * every other byte selects none.  Every row is summed in `order`.  pre may be
 * NULL. */
int crp_score_30mers(crp_ctx *ctx, const uint8_t *rows, uint64_t n, int order, double *pre, double *score);

/* ---- output side: native CSV rows (host code, no GPU needed) ---------------- */
/* Formats n_rows rows of ONE contig exactly as the reference's row tuples go through
 * csv.writer.writerows (CROPSR.py:463-474; strings as at :420-421 / :431-432):
 *   contig_text/contig_len  the contig string (1 byte per character)
 *   chrom/chrom_len         the chromosome column text (CROPSR.py:422 chromosome[1::])
 *   pos[r], minus[r]        regex match index and strand (0 '+', 1 '-') of row r
 *   score[r]                on_site_score of row r (ignored when long_sequence != 30 chars)
 *   ids                     n_rows x 7 bytes, the crispr_id of each row
 * Output: the CSV bytes ("\r\n" line ends, minimal quoting, repr() floats, 11-field rows
 * with -1 where the reference writes them).  Returns CRP_ERR_CAPACITY with the needed size
 * in *out_len when out_cap is too small. */
int crp_format_rows(const uint8_t *contig_text, uint64_t contig_len, const uint8_t *chrom, uint64_t chrom_len,
                    int guide_len, const uint32_t *pos, const uint8_t *minus, const double *score,
                    const uint8_t *ids, uint64_t n_rows, uint8_t *out, uint64_t out_cap, uint64_t *out_len,
                    int n_threads);

/* The same rows appended to an open file descriptor instead of a buffer: the stand-in for
 * csv.writer(file).writerows(rows) at CROPSR.py:471-474 when the rows of a chunk are not
 * wanted in memory.  n_threads workers format blocks of 16384 rows and commit them with
 * write(2) in row order (a descriptor opened with O_APPEND, like Python's mode "a", appends).
 * *bytes_written (may be NULL) receives the number of bytes written; on CRP_ERR_IO the output
 * stops at a block boundary.  Flush any buffered writer on the same file before the call. */
int crp_write_rows(int fd, const uint8_t *contig_text, uint64_t contig_len, const uint8_t *chrom, uint64_t chrom_len,
                   int guide_len, const uint32_t *pos, const uint8_t *minus, const double *score,
                   const uint8_t *ids, uint64_t n_rows, uint64_t *bytes_written, int n_threads);

/* The reference's own crispr ids (CROPSR.py:316-318: np.random.choice(alphanum, [n_rows, 7]) on
 * numpy's global legacy generator), drawn natively: mt_key[624] / *mt_pos are the MT19937 state
 * from np.random.get_state(); on return they hold the state after the draws, for set_state().
 * Same values, same state afterwards as numpy; reverse != 0 stores the rows last-first. */
int crp_legacy_ids(uint32_t *mt_key, int32_t *mt_pos, uint8_t *ids, uint64_t n_rows, int reverse);

/* ---- input side: FASTA bytes -> contig strings (host code, no GPU needed) ----- */
/* The contig table import_fasta_file builds (CROPSR.py:54-74 with cropsr_functions.py:221-229
 * and :190-196) for a file in the "re-formatted" path whose records all have a header line and
 * whose headers and bodies are plain (printable ASCII without blank, quote, backslash):
 *   data/n        the file's bytes as text mode hands them over (newlines already '\n')
 *   out_text      receives the VALUE strings back to back: for record k
 *                 ' + body without newlines + ') + (',' or, for the last record, ']')
 *                 -- the exact character string CROPSR.py:412-434 scans for that contig
 *   records       4 x uint64 per record: header offset and length in `data`, value offset and
 *                 length in `out_text` (the host prepends  [('  or  ('  and appends  ',  to the
 *                 header to get the dict key, and applies dict semantics to repeated keys)
 * *plain = 1 when the table was produced.  *plain = 0 (with CRP_OK) means the input is outside
 * this fast path -- already two lines per record, a header with a blank or quote, a record
 * without a newline, no record at all -- and the caller must build the table the literal way.
 * CRP_ERR_CAPACITY reports the needed sizes in *n_records / *out_len (out_cap >= n + 4 * records
 * always suffices).  Work is spread over n_threads threads in 4 MiB pieces of the input. */
int crp_fasta_table(const uint8_t *data, uint64_t n, uint8_t *out_text, uint64_t out_cap, uint64_t *records,
                    uint64_t records_cap, uint64_t *n_records, uint64_t *out_len, int *plain, int n_threads);

/* The same rows with the two OPT-IN extensions of this engine (both absent from the reference, whose
 * default output stays byte-identical when they are not asked for):
 *   features    the `features` column (always '' in the reference, CROPSR.py:466-468): a table of
 *               strings, entry t = feat_blob[feat_off[t] : feat_off[t+1]], and per row the entry it
 *               gets, feat_idx[r] (0xFFFFFFFF: ''); csv-quoted as needed.  feat_idx NULL = ''.
 *               Rows without a cutsite (the 11-field rows) keep ''.
 *   offtarget   4 x uint32 per row appended as four more columns (crp_offtarget_counts);
 *               0xFFFFFFFF prints as -1.  NULL = no extra columns.
 * With both NULL this is crp_write_rows. */
int crp_write_rows_ex(int fd, const uint8_t *contig_text, uint64_t contig_len, const uint8_t *chrom, uint64_t chrom_len,
                      int guide_len, const uint32_t *pos, const uint8_t *minus, const double *score,
                      const uint8_t *ids, uint64_t n_rows, const uint8_t *feat_blob, const uint64_t *feat_off,
                      const uint32_t *feat_idx, const uint32_t *offtarget, uint64_t *bytes_written, int n_threads);

/* Rows of several SEGMENTS -- a segment is what crp_write_rows_ex takes: consecutive rows of one contig with their columns --
 * appended to fd in segment order by ONE call: the rows a write pass of the reference sends through csv.writer.writerows
 * (CROPSR.py:442-474; with the reference's accumulating Complete_dataset, :407, a pass holds the rows of every contig so far),
 * or the passes of several short contigs together.  A worker's block (16384 rows, one write(2)) may span segments, so a run of
 * short contigs costs what one contig of their total size costs.  Bytes: exactly the concatenation of the segments'
 * crp_write_rows_ex output.  feat_idx / offtarget may be NULL per segment like there; guide_len is the call's. */
typedef struct crp_row_segment {
    const uint8_t *contig_text;
    uint64_t contig_len;
    const uint8_t *chrom;
    uint64_t chrom_len;
    const uint32_t *pos;
    const uint8_t *minus;
    const double *score;
    const uint8_t *ids;
    uint64_t n_rows;
    const uint8_t *feat_blob;
    const uint64_t *feat_off;
    const uint32_t *feat_idx;
    const uint32_t *offtarget;
} crp_row_segment;
int crp_write_segments(int fd, int guide_len, const crp_row_segment *segs, uint64_t n_segs, uint64_t *bytes_written, int n_threads);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ------------------------ */
/* The reference has no parallelism (its only hint is the dead cropsr_functions.py:256-273).  The path
 * shards by contig with no collective on the data path; the one exchange is the final gatherv of the
 * per-rank hit tables to a root.  RCCL has no gatherv: crp_gather_hits does an all-gather of the two
 * counts, then ncclGroupStart / ncclSend|ncclRecv per column / ncclGroupEnd -- every peer->root
 * transfer on its own xGMI link.  librccl.so is loaded on the first crp_comm_* call, never before.
 * The 128-byte id comes from crp_comm_unique_id on one rank and reaches the others through the
 * launcher (cropsr_amd/rendezvous.py: a socket of its own, not torch). */
#define CRP_COMM_ID_BYTES 128
int crp_comm_unique_id(uint8_t id[CRP_COMM_ID_BYTES]);
int crp_comm_init(crp_ctx *ctx, const uint8_t id[CRP_COMM_ID_BYTES], int rank, int world);
int crp_comm_destroy(crp_ctx *ctx);
/* All ranks: returns when every rank has called it and the library's stream has drained. */
int crp_comm_barrier(crp_ctx *ctx);
/* Element-wise reduction of n (<= 64) host doubles over all ranks, result on every rank. */
#define CRP_REDUCE_SUM 0
#define CRP_REDUCE_MAX 1
int crp_comm_allreduce_f64(crp_ctx *ctx, double *values, int n, int op);
/* All ranks, after crp_scan_score on `arena` (NULL: this rank contributes empty tables): gathers the
 * '+' and '-' tables (pos u32, score f64) of every rank into root's HBM.  flags (the same on every
 * rank): CRP_GATHER_OFFTARGET also gathers the per-hit off-target counts (crp_offtarget_counts must
 * have run on `arena`).  counts_all (2 x world values, may be NULL) receives every rank's
 * {n_plus, n_minus} on every rank.
 * Failure is collective: whatever can go wrong on ONE rank before the tables move -- its arena has no
 * (matching) tables: CRP_ERR_STATE; the root cannot size its receive buffers: CRP_ERR_NOMEM -- travels with
 * the counts, and every rank returns from the same call without starting the exchange: the rank at fault with
 * its own status, the others with CRP_ERR_PEER (crp_last_error names the rank).  The communicator stays
 * usable.  CRP_ERR_COMM / CRP_ERR_HIP from inside the exchange are NOT agreed on: the caller must end the run. */
#define CRP_GATHER_OFFTARGET 1
/* CRP_GATHER_PRE: the f64 column that travels is the pre-sigmoid sum (crp_scan_score with want_pre)
 * instead of the score -- for a root that applies its own host's exp (cli --score-finalize=host). */
#define CRP_GATHER_PRE 2
/* CRP_GATHER_FEATURES: also gathers the per-hit label-set ids of the annotation join (crp_annotate_lookup must
 * have run on `arena` since its last scan; else CRP_ERR_STATE, agreed on like the others). */
#define CRP_GATHER_FEATURES 4
/* CRP_GATHER_POS16: positions cross the links as 16 bits per hit plus one uint32 per 65 536 arena positions (the index of
 * the first hit at or after it) instead of 32 bits per hit -- tables are ascending, so this is exact and needs no escape:
 * 10 B per hit instead of 12.  A small kernel packs on the peers, another expands into the root's table
 * (crp_gather.hip); what crp_gathered_fetch returns is bit for bit the same. */
#define CRP_GATHER_POS16 8
int crp_gather_hits(crp_ctx *ctx, crp_arena *arena, int root, int flags, uint64_t *counts_all);
/* Root only: copy the tables rank `rank` contributed to the last crp_gather_hits to host arrays
 * (sizes from counts_all; ot_*: 4 x uint32 per hit; any pointer may be NULL). */
int crp_gathered_fetch(crp_ctx *ctx, int rank, uint32_t *pos_plus, double *score_plus, uint32_t *ot_plus,
                       uint32_t *pos_minus, double *score_minus, uint32_t *ot_minus);

/* Root only, after a crp_gather_hits with CRP_GATHER_FEATURES: the ids rank `rank` contributed (uint32 per hit). */
int crp_gathered_fetch_features(crp_ctx *ctx, int rank, uint32_t *feat_plus, uint32_t *feat_minus);

/* ---- seam 1 as a pipeline: upload | scan | fetch side by side ---------------- */
/* The reference's loop produces and consumes contig by contig (CROPSR.py:409-474).  crp_arena_* + crp_scan_score +
 * crp_fetch_hits do the three steps of seam 1 one after the other; crp_scan_stream does them as a pipeline over slices
 * of the genome (slice_chars characters each, 0 = 64 Mi; whole contigs while they fit, a longer contig cut with
 * CRP_HALO characters of context either side and every hit owned by the piece its match index falls in): while slice k is
 * scanned, slice k + 1 crosses the host link upwards and the tables of slice k - 1 cross it downwards.  One blocking
 * call; the caller's strings are read, not kept.
 *   texts / lens / n   the contig strings CROPSR.py:409 iterates over, in order (each at most 2^32 - 1 characters)
 *   flags              CRP_SCAN_PRE: the f64 column is the pre-sigmoid sum (CROPSR.py:312) instead of the score
 *   pos_* / score_*    the caller's tables, cap_* rows each (a pointer may be NULL: that column is not copied): ONE table
 *                      per strand, contig after contig, ascending inside a contig, positions LOCAL to the contig string
 *                      (the regex match indices of CROPSR.py:418 / :429) -- what crp_node_fetch returns.  Pinned memory
 *                      (crp_host_alloc) is written by DMA directly; pageable memory through the staging buffers.
 *   per_contig         2 x n values {plus, minus} (may be NULL); *n_plus / *n_minus: the totals
 *   stats              12 doubles (may be NULL): wall seconds, uploader busy, drainer busy (incl. waiting for the scans),
 *                      slices, lanes, seconds until the first slice's tables were on the host, uploader waiting for a
 *                      lane, drainer waiting for a staging buffer, copier busy (staging -> the caller's pageable tables),
 *                      copier waiting for the D2H copies, bytes the copier moved, 1 if the tables were pinned
 * CRP_ERR_CAPACITY: a table was too small -- nothing was written beyond cap_*, *n_plus / *n_minus hold the sizes to
 * come back with.  crp_scan_stream_prepare opens the pipeline's lanes (further contexts on the device -- four lanes in all,
 * environment CRP_STREAM_LANES 2..8 -- each with an arena of slice_chars and its tables) ahead of time, e.g. while the
 * FASTA is still being read. */
int crp_scan_stream(crp_ctx *ctx, const uint8_t *const *texts, const uint64_t *lens, uint64_t n, int guide_len, int flags,
                    uint64_t slice_chars, uint32_t *pos_plus, double *score_plus, uint64_t cap_plus, uint32_t *pos_minus,
                    double *score_minus, uint64_t cap_minus, uint64_t *per_contig, uint64_t *n_plus, uint64_t *n_minus,
                    double *stats);
int crp_scan_stream_prepare(crp_ctx *ctx, uint64_t slice_chars);
/* Host memory the GPU reaches by DMA (hipHostMalloc): tables allocated here are filled by crp_fetch_hits,
 * crp_node_fetch and crp_scan_stream without a staging copy and without first-touch page faults, and contig strings kept
 * here are uploaded straight off their pages (they must then stay valid until crp_arena_seal / the end of
 * crp_scan_stream).  For callers that keep their buffers from genome to genome: pinning costs more than one copy saves. */
int crp_host_alloc(uint64_t bytes, void **out);
int crp_host_free(void *p);

/* ---- multi-GPU: ONE process over N GPUs, the node handle -------------------- */
/* The reference is one process with one contig loop (CROPSR.py:333, :409).  A crp_node lets that one process drive N
 * GPUs through one handle: crp_node_load cuts the genome into N contiguous equal shares (a contig that straddles a
 * share boundary is cut there; every piece carries CRP_HALO characters of context either side, a hit belongs to the
 * piece that contains its match index) and uploads every share to its device; crp_node_scan_score launches the scan on
 * every device from a host thread of that device's own, so the kernels start side by side; crp_node_gather is the path's one exchange, the gatherv of the per-device tables
 * to a root device -- RCCL in one process (ncclCommInitAll; per peer ncclSend, at the root ncclRecv, all inside one
 * ncclGroupStart / ncclGroupEnd) or, where RCCL cannot run (the same device listed twice: rehearsals on one GPU) or is
 * not wanted, device-to-device copies the root pulls over its own streams, one per peer.  The root ends up with ONE
 * table per strand, in contig order, positions LOCAL to their contig string: exactly the regex match indices the
 * reference iterates over, bit for bit what one GPU alone produces.  The per-device crp_ctx / crp_arena stay
 * reachable (crp_node_ctx / crp_node_arena) for crp_configure, crp_profile_*, crp_device_info and the like. */
#define CRP_HALO 128
typedef struct crp_node crp_node;
/* Host side (no GPU needed): the cut.  lens[0..n): contig string lengths, in order.  Device r gets the characters
 * [r, r + 1) * total / world of the concatenation; a contig that straddles a boundary is cut there unless one side
 * would be shorter than min_piece (0: 4096) -- then it stays whole on the side that holds most of it.  pieces: 4 x uint64
 * per piece {contig, start, end, device}, in contig order; CRP_ERR_CAPACITY with the needed count in *n_pieces when cap
 * (in pieces) is too small.  At most n + world - 1 pieces. */
int crp_plan_shares(const uint64_t *lens, uint64_t n, int world, uint64_t min_piece, uint64_t *pieces, uint64_t cap,
                    uint64_t *n_pieces);
/* Opens the listed HIP devices (indices as this process sees them; the same index may appear more than once: each
 * entry is a logical device with a context of its own). */
int crp_node_init(int n_devices, const int *device_ids, crp_node **out);
int crp_node_destroy(crp_node *node);
/* Text of the last failure on the node or on one of its devices. */
const char *crp_node_last_error(const crp_node *node);
int crp_node_size(const crp_node *node);           /* number of logical devices (negative status on a NULL handle) */
crp_ctx *crp_node_ctx(crp_node *node, int k);      /* logical device k's context (NULL: out of range) */
crp_arena *crp_node_arena(crp_node *node, int k);  /* its (first) arena after crp_node_load (NULL: none -- the device got no piece) */
/* A device's share is packed into as many arenas as it needs (one arena addresses fewer than 2^31 characters; the
 * reference reads a genome of any size whole, CROPSR.py:59): a share beyond one arena goes on in the next, and a piece that
 * no arena can hold is cut again, with halos like every other cut.  Scan, the two opt-in steps, the ownership cuts and the
 * gatherv run arena by arena inside the library; the caller sees one table per strand as before.  crp_node_arenas: how
 * many arenas logical device k holds (negative status: bad argument); crp_node_arena_at: its j-th one. */
int crp_node_arenas(const crp_node *node, int k);
crp_arena *crp_node_arena_at(crp_node *node, int k, int j);
/* Knobs of a node (the environment sets the defaults at crp_node_init):
 *   CRP_NODE_OPT_ARENA_WORDS            most 64-character words one arena of the node may hold; 0 = crp_arena_max_words().
 *                                       Smaller values only make more arenas (tests; a way to bound one allocation).
 *   CRP_NODE_OPT_COMM_INIT_TIMEOUT_MS   how long ncclCommInitAll may take (it runs on a helper thread; default 180 000,
 *                                       environment CRP_NODE_COMM_INIT_TIMEOUT_S): a bootstrap that has not returned by then
 *                                       is left behind (crp_node_comm_stuck) and the node goes on as device-to-device
 *                                       copies, or fails with CRP_ERR_COMM if RCCL was asked for by name
 *                                       (CRP_NODE_TRANSPORT=rccl).  <= 0: no bound.
 *   CRP_NODE_OPT_COLLECTIVE_TIMEOUT_MS  how long the grouped send/recv of crp_node_gather and the histogram all-reduce
 *                                       of crp_node_offtarget may take (default 300 000, environment
 *                                       CRP_NODE_COLLECTIVE_TIMEOUT_S): RCCL has no time-out of its own, so the streams are
 *                                       awaited by polling an event.  When the bound runs out the communicators are
 *                                       aborted (ncclCommAbort) and never used again by this node; the call then starts over
 *                                       on the device-to-device transport from fresh state and returns CRP_OK (crp_node_last_error
 *                                       and crp_node_transport_note say what happened) -- or, with CRP_NODE_TRANSPORT=rccl,
 *                                       returns CRP_ERR_COMM naming the stage.  <= 0: no bound. */
#define CRP_NODE_OPT_ARENA_WORDS 1
#define CRP_NODE_OPT_COMM_INIT_TIMEOUT_MS 2
#define CRP_NODE_OPT_COLLECTIVE_TIMEOUT_MS 3
int crp_node_set_option(crp_node *node, int option, int64_t value);
/* Why RCCL is not (or no longer) this node's transport -- "" while it is, or was never wanted. */
const char *crp_node_transport_note(const crp_node *node);
/* Communicator bootstraps of this PROCESS that never returned: their helper threads still sit inside RCCL, and the
 * runtime's tear-down at a normal exit may wait for them -- a program that sees a non-zero value should flush its output
 * and leave through _exit (cropsr_amd/cli.py does). */
int crp_node_comm_stuck(void);
/* The genome: n contig strings, in order (the strings CROPSR.py:409 iterates over).  Cuts (crp_plan_shares), uploads
 * share r to device r (one host thread per device), seals.  The caller's strings are read, not kept.  A second call
 * replaces the genome.  Any total size; a single contig string may have at most 2^32 - 1 characters (the tables carry
 * contig-local positions as 32 bits): CRP_ERR_CAPACITY names a longer one. */
int crp_node_load(crp_node *node, const uint8_t *const *texts, const uint64_t *lens, uint64_t n);
/* The cut that was made: 7 x uint64 per piece {contig, start, end, device, arena offset of the piece's text in its
 * arena, index of `start` inside that text (the left halo's length), which of the device's arenas}.  Same capacity
 * protocol as crp_plan_shares; at most n + world - 1 pieces plus one per arena beyond a device's first. */
int crp_node_plan(const crp_node *node, uint64_t *pieces, uint64_t cap, uint64_t *n_pieces);
/* crp_scan_score on every device at once (same guide_len / flags).  *n_plus / *n_minus (may be NULL): rows of all
 * devices' tables together, INCLUDING the few hits inside halos (at most 2 x CRP_HALO positions per device); the
 * exact totals come from crp_node_gather. */
int crp_node_scan_score(crp_node *node, int guide_len, int flags, uint64_t *n_plus, uint64_t *n_minus);
/* The two opt-in steps over the node's resident tables, between crp_node_scan_score and crp_node_gather:
 *   crp_node_offtarget   the genome-wide off-target seed scan (see crp_offtarget_* below): every device adds the sites it
 *                        OWNS to its histogram (scan with CRP_SCAN_SEEDS to hand the seed words over), the histograms
 *                        are summed over the devices (RCCL all-reduce in one group; through device 0 when RCCL cannot
 *                        run), every device solves and looks up its own hits.  *n_sites (may be NULL): sites in all.
 *   crp_node_annotate    the annotation join (see crp_annotate_* below) on every device with the track of ITS pieces:
 *                        seqid_of_contig[k] = index of contig k's FASTA name among the annotation's seqids
 *                        (crp_annotation_seqid order; any value >= their number: no features), dec as in
 *                        crp_annotation_track.
 * The counts / ids stay in HBM and travel with crp_node_gather (CRP_GATHER_OFFTARGET / CRP_GATHER_FEATURES). */
struct crp_annotation;
int crp_node_offtarget(crp_node *node, int guide_len, uint64_t *n_sites);
int crp_node_annotate(crp_node *node, const struct crp_annotation *annotation, const uint64_t *seqid_of_contig, int dec);
/* The gatherv: every device's OWNED rows to logical device `root`.  flags: CRP_GATHER_PRE (the f64 column is the
 * pre-sigmoid sum), CRP_GATHER_POS16 (10 B per hit on the links instead of 12, see above), CRP_GATHER_OFFTARGET /
 * CRP_GATHER_FEATURES (the columns of the two steps above travel too),
 * CRP_NODE_PEER_COPY (device-to-device copies instead of RCCL for this call; always the case when a device is
 * listed twice, or when the environment says CRP_NODE_TRANSPORT=peer).  CRP_NODE_TRANSPORT=rccl: RCCL or CRP_ERR_COMM (no
 * falling back; also with a device listed twice, where the real RCCL refuses the clique); CRP_NODE_TRANSPORT=try: RCCL
 * is attempted even with a device listed twice and given up for device-to-device copies like in the default mode. */
#define CRP_NODE_PEER_COPY 16
/* CRP_NODE_HOST_GATHER: for a consumer that lives on the HOST (the reference's consumer does: csv.writer, CROPSR.py:471-474).
 * Nothing crosses xGMI and no table is built on `root`: the call makes the ownership cuts and the counts, every device
 * rebases the positions of its owned rows where they lie, and crp_node_fetch (/_fetch_offtarget /_fetch_features) then
 * copies every device's rows over THAT device's own PCIe link, side by side, straight into their place in the caller's
 * arrays -- N host links instead of a gather into one GPU followed by one link.  Same rows, same order.
 * crp_node_tables_device has nothing to hand out after such a gather (CRP_ERR_STATE). */
#define CRP_NODE_HOST_GATHER 32
int crp_node_gather(crp_node *node, int root, int flags);
/* After crp_node_gather: kept hits per contig and strand (2 x n contigs: {plus, minus}) and in all; any pointer may
 * be NULL. */
int crp_node_counts(const crp_node *node, uint64_t *per_contig, uint64_t *n_plus, uint64_t *n_minus);
/* Rows of the gathered tables that carry a real score (not -1), counted on the root: the unit of "gRNAs scored". */
int crp_node_count_scored(crp_node *node, uint64_t *n_scored);
/* The gathered tables to host arrays (n_plus / n_minus rows; contig order, ascending inside a contig, positions local
 * to the contig string; any pointer may be NULL), and their addresses in the root's HBM (valid until the next
 * crp_node_gather / crp_node_load). */
int crp_node_fetch(crp_node *node, uint32_t *pos_plus, double *score_plus, uint32_t *pos_minus, double *score_minus);
int crp_node_tables_device(crp_node *node, void **pos_plus, void **score_plus, void **pos_minus, void **score_minus);
/* After a gather with CRP_GATHER_OFFTARGET / CRP_GATHER_FEATURES: the off-target counts (4 x uint32 per row) and the
 * label-set ids (uint32 per row) of the same rows, same order.  Either pointer may be NULL. */
int crp_node_fetch_offtarget(crp_node *node, uint32_t *ot_plus, uint32_t *ot_minus);
int crp_node_fetch_features(crp_node *node, uint32_t *feat_plus, uint32_t *feat_minus);
/* The last crp_node_gather in numbers: wall time of the whole call and of its exchange step alone (ms), bytes that
 * crossed from peers to the root, transport used (1 RCCL, 2 device-to-device copies, 3 none: host gather).  Any pointer may be NULL. */
#define CRP_TRANSPORT_RCCL 1
#define CRP_TRANSPORT_PEER_COPY 2
#define CRP_TRANSPORT_HOST_LINKS 3   /* CRP_NODE_HOST_GATHER: nothing moved between devices */
int crp_node_gather_stats(const crp_node *node, double *ms_total, double *ms_exchange, uint64_t *bytes_to_root, int *transport);

/* ---- annotation join (opt-in; a no-op in the reference) -------------------- */
/* BASELINE.json configs[2], [3] name a GFF and a Phytozome annotation_info file.  The reference parses the GFF
 * (CROPSR.py:77-95, called at :375), drops the table and writes '' into `features` (:466, :468), so the join is
 * this engine's own, opt-in, and the default CSV stays byte-identical (definition: cropsr_amd/annotate.py,
 * DESIGN.md section 11; parity unpinned).  The host builds, per arena, a TRACK: strictly ascending arena positions
 * points[k], each opening an elementary interval [points[k], points[k+1]) whose set of gene / CDS labels is
 * constant and named by ids[k] (an index into the caller's string table, or CRP_NO_FEATURE for "none"; every
 * contig starts with a point of its own, so nothing leaks from its predecessor).  The library keeps a copy in HBM
 * together with a bucket index over the arena's positions. */
#define CRP_NO_FEATURE 0xFFFFFFFFu
/* Host side (no GPU needed).  GFF3 bytes (+ the bytes of a Phytozome annotation_info file, or NULL) as text mode hands
 * them over -> the annotation: the distinct label-set strings and, per seqid, the elementary intervals of its gene /
 * CDS rows.  The definition (which rows, which label, which order) is stated at the top of
 * cropsr_amd/csrc/crp_annotation.cpp and restated as a brute-force loop in oracle/annotate_oracle.py. */
typedef struct crp_annotation crp_annotation;
int crp_annotation_build(const uint8_t *gff, uint64_t gff_len, const uint8_t *info_text, uint64_t info_len,
                         crp_annotation **out);
int crp_annotation_destroy(crp_annotation *annotation);
/* Sizes: seqids with at least one gene / CDS row, distinct label-set strings, bytes of all strings, gene and CDS
 * rows read (any pointer may be NULL). */
int crp_annotation_stats(const crp_annotation *annotation, uint64_t *n_seqids, uint64_t *n_strings, uint64_t *blob_bytes,
                         uint64_t *n_genes, uint64_t *n_cds);
/* The string table in the form crp_write_rows_ex takes: blob (blob_bytes) and n_strings + 1 offsets. */
int crp_annotation_strings(const crp_annotation *annotation, uint8_t *blob, uint64_t *offsets);
/* Seqid k (order of first appearance in the GFF): its name and its intervals in 1-based genome coordinates --
 * interval j = [points[j], points[j+1]) carries string ids[j] (CRP_NO_FEATURE: none).  The pointers stay valid
 * until crp_annotation_destroy. */
int crp_annotation_seqid(const crp_annotation *annotation, uint64_t k, const uint8_t **name, uint64_t *name_len,
                         const int64_t **points, const uint32_t **ids, uint64_t *n_points);
/* The track of one arena for crp_annotate_set_track.  entries: 4 x uint64 per text of the arena, in arena order --
 * {seqid index (any value >= n_seqids: the text has no features), index of the text's first character inside its
 * contig string (0 unless the text is a piece of a cut contig), length of the text, arena offset of the text}.
 * dec: the contig strings start with `dec` decoration characters before the first base (1 in the reference's
 * re-formatted path, 0 otherwise): string index = 1-based genome coordinate + dec - 1.
 * Every text opens with a point of its own.  Returns CRP_ERR_CAPACITY with the needed size in *n_out when cap is
 * too small (cap 0: sizing call). */
int crp_annotation_track(const crp_annotation *annotation, const uint64_t *entries, uint64_t n_entries, int dec,
                         uint32_t *points, uint32_t *ids, uint64_t cap, uint64_t *n_out);
/* Device side. */
int crp_annotate_set_track(crp_arena *arena, const uint32_t *points, const uint32_t *ids, uint64_t n_points);
/* After crp_scan_score on `arena` (whose tables it reads where they lie): per kept hit the id of the interval its
 * cut site falls in -- cut site = end_pos - 3 (CROPSR.py:155-158): i - 3 on the '+' table, j on the '-' table --
 * or CRP_NO_FEATURE for a hit before the first point and for a row WITHOUT a cut site (long_sequence is not 30
 * characters, CROPSR.py:466-468: exactly the rows whose score is -1).  One streaming pass over the position and
 * score columns (12 B in, 4 B out per hit).  The ids stay in HBM for crp_gather_hits(CRP_GATHER_FEATURES) and
 * are copied to feat_plus / feat_minus (n_plus / n_minus values, table order; either may be NULL); they are what
 * crp_write_rows_ex takes as feat_idx. */
int crp_annotate_lookup(crp_arena *arena, uint32_t *feat_plus, uint32_t *feat_minus);

/* ---- off-target seed scan (opt-in; absent from the reference) ------------- */
/* BASELINE.json configs[4]: genome-wide off-target <=3-mismatch seed scan.  The reference has no such
 * step (its only alignment code is the dead bowtie2 shell-out of prmrdsgn2.py:139-160), so the
 * definition is this engine's own, stated on the reference's own strings:
 *   site   every kept hit of the scan (either strand, CROPSR.py:419/:430) whose `sequence` column
 *          (CROPSR.py:420/:431) has >= 12 characters of which the first 12 -- the PAM-proximal seed --
 *          are bases after the scoring transform of CROPSR.py:458 (.replace('U','T').upper() in ACGT)
 *   count  for a site g and k = 0..3: the number of OTHER sites whose seed differs from g's in
 *          exactly k of the 12 positions
 * Genome-wide means: over every arena added between crp_offtarget_reset and crp_offtarget_solve, on
 * every rank of the communicator (crp_offtarget_reduce).  Method: a histogram of the 4^12 seeds (built
 * by partitioning the sites by seed prefix, no atomic per site), then
 * the exact Hamming-ball sums of that histogram for all seeds at once by a position-wise recurrence
 * (three passes of four positions through LDS), then one 16-byte look-up per hit -- no pairwise
 * comparison anywhere. */
#define CRP_OT_SEED_LEN 12
#define CRP_OT_MAX_MM 3
/* Allocates (first call) and zeroes the site histogram of this context. */
int crp_offtarget_reset(crp_ctx *ctx);
/* After crp_scan_score on `arena`: derives the seed of every kept hit and adds the sites to the
 * histogram.  own_ranges (n_ranges pairs [begin, end) of arena positions, ascending, may be NULL =
 * everything): only hits whose match position lies inside count as sites -- a contig cut into pieces
 * with halos (multi-GPU) must not count its halo hits twice.  *n_sites (may be NULL): sites added. */
int crp_offtarget_add(crp_arena *arena, int guide_len, const uint64_t *own_ranges, uint64_t n_ranges,
                      uint64_t *n_sites);
/* Multi-GPU: RCCL all-reduce (sum) of the histogram over the communicator -- the one bandwidth-heavy
 * xGMI collective of this engine (64 MiB per rank).  Without a communicator: no-op. */
int crp_offtarget_reduce(crp_ctx *ctx);
/* The site histogram (4^12 uint32) to / from host memory: what a host-side exchange sums instead of
 * crp_offtarget_reduce (ranks sharing one GPU, where RCCL cannot run), and what the tests check. */
int crp_offtarget_hist_get(crp_ctx *ctx, uint32_t *hist);
int crp_offtarget_hist_set(crp_ctx *ctx, const uint32_t *hist);
/* Hamming-ball sums for all 4^12 seeds. */
int crp_offtarget_solve(crp_ctx *ctx);
/* Per-hit counts of `arena` (which must have been added): 4 x uint32 per hit, table order, k = 0..3;
 * 0xFFFFFFFF x 4 for a hit that is not a site.  Host pointers may be NULL (counts stay in HBM). */
int crp_offtarget_counts(crp_arena *arena, uint32_t *counts_plus, uint32_t *counts_minus);
/* The seed codes of the same hits (uint32 per hit: sum of code_k << 2k, A=0 T=1 C=2 G=3, k = 0 next
 * to the PAM; 0xFFFFFFFF not a site, 0xFFFFFFFE a site outside own_ranges), for tests. */
int crp_offtarget_seeds(crp_arena *arena, uint32_t *seeds_plus, uint32_t *seeds_minus);

/* ---- options -------------------------------------------------------------- */
/* CRP_OPT_TWO_PASS (value 0/1, default 0): with 0 crp_scan_score is ONE kernel launch; the
 * table offsets come from a chained scan across workgroups inside it (decoupled look-back
 * over per-tile descriptors; every wait is bounded in wall time).  With 1 it runs the count /
 * tile-scan / emit+score launch sequence (one more pass over the packed planes).  Results
 * are identical; the single launch is ~20 % faster per scan on MI355X.
 * A scan in which a look-back ran out of its allowance is repeated with the three-launch
 * sequence (see CRP_OPT_CHAIN_TIMEOUT_US); the setting itself is not changed by that. */
#define CRP_OPT_TWO_PASS 1
/* CRP_OPT_CHAIN_TIMEOUT_US (default 20000): how long a workgroup of the single-launch scan waits for
 * a predecessor's counts before it gives up (wall time, read from the GPU's 100 MHz real-time
 * counter).  A scan in which a wait ran out is repeated with the three-launch sequence; the next scan
 * tries the single launch again.  Only after three such scans in a row does the context stay with
 * three launches (crp_query reports both). */
#define CRP_OPT_CHAIN_TIMEOUT_US 2
/* CRP_OPT_TILE_GEOMETRY (default 0): the tile shape crp_arena_seal gives the arenas sealed AFTER the call.  The scan works
 * tile by tile, one workgroup per tile; results do not depend on the shape, time does:
 *   0  by the arena's size against the GPU: SMALL when the arena holds fewer than 1.5 LARGE tiles per CU (E. coli-like:
 *      one tile's life is the whole kernel), LARGE otherwise
 *   1  LARGE   512 threads on 1 024 words (65 536 positions), two words per lane: the throughput shape
 *   2  SMALL   512 threads on 512 words, one word per lane: half the rows per lane, half the tile's life */
#define CRP_OPT_TILE_GEOMETRY 3
int crp_configure(crp_ctx *ctx, int option, int64_t value);

/* Read-only state, for logs and bench output. */
#define CRP_Q_CHAIN_TIMEOUTS 1   /* single-launch scans that had to be repeated with three launches */
#define CRP_Q_TWO_PASS_ACTIVE 2  /* 1 when scans currently run as three launches (configured or latched) */
#define CRP_Q_COMM_WORLD 3       /* ranks of the RCCL communicator (0: none) */
#define CRP_Q_COMM_RANK 4
#define CRP_Q_HBM_FREE 5         /* bytes of device memory free right now, as hipMemGetInfo sees the whole device (all processes) */
#define CRP_Q_HBM_TOTAL 6
#define CRP_Q_GATHER_BYTES 7     /* last crp_gather_hits: bytes this rank sent to the root (a peer) or received from all peers (the root) */
int crp_query(const crp_ctx *ctx, int what, int64_t *value);
/* Identifies the build: a hash of the library's sources taken by the Makefile ("unknown" otherwise).
 * profiles/traffic.json carries the id of the build it was measured on; bench.py refuses another. */
const char *crp_build_id(void);

/* ---- measurement --------------------------------------------------------- */
/* on = 1: the emit+score kernel of every crp_scan_score is bracketed by HIP events
 * on the stream it is launched on; on = 2: all three kernels; 0: off. */
int crp_profile_enable(crp_ctx *ctx, int on);
/* Sum of durations (ms) and launch count per kernel since the last reset:
 * index 0 = count pass, 1 = tile-offset scan (both only with CRP_OPT_TWO_PASS = 1),
 * 2 = emit+score pass (the whole scan with CRP_OPT_TWO_PASS = 0). */
int crp_profile_read(crp_ctx *ctx, double ms[3], uint64_t launches[3], int reset);
/* The same for any kernel kind (ms and launches since the last reset of that kind). */
#define CRP_K_COUNT 0
#define CRP_K_TILE_SCAN 1
#define CRP_K_EMIT 2
#define CRP_K_OT_SEED 3      /* off-target: seeds of the kept hits + site histogram */
#define CRP_K_OT_BALL 4      /* off-target: the three Hamming-ball passes together */
#define CRP_K_OT_LOOKUP 5    /* off-target: per-hit counts */
#define CRP_K_GATHER 6       /* RCCL gatherv of the hit tables (count all-gather + grouped send/recv) */
#define CRP_K_OT_REDUCE 7    /* RCCL all-reduce of the site histogram */
#define CRP_K_ANNOTATE 8     /* annotation join: the look-up kernel of one crp_annotate_lookup (both tables, one launch) */
#define CRP_K_KINDS 9
int crp_profile_read_kind(crp_ctx *ctx, int kind, double *ms, uint64_t *launches, int reset);
/* Blocks until everything queued on the library's stream has finished. */
int crp_synchronize(crp_ctx *ctx);
/* Number of rows of the last crp_scan_score that carry a real score (not -1), counted on the GPU:
 * the unit of the "gRNAs scored" metric. */
int crp_count_scored(crp_arena *arena, uint64_t *n_scored);

#ifdef __cplusplus
}
#endif
#endif /* CROPSR_HIP_H */
