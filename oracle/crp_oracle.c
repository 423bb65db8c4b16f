/*
 * crp_oracle.c -- CPU restatement of CROPSR's PAM-scan + on-target-score path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: it may be used by
 * tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and by
 * nothing else.  The product (cropsr_amd/) never links, loads or calls it.
 *
 * Pinned: yes.  tests/test_oracle.py checks this restatement against golden
 * vectors produced by running the real reference (tests/golden/make_golden.py):
 * every row of the sample genome CSV and of the quirk probes, the seam-2
 * rs1_score vectors, and exp() against the host libm.
 *
 * It follows the reference one function at a time (file:line into
 * /root/reference):
 *   orc_scan            CROPSR.py:98-104 (find_PAM_site), :415-434 (window/filter)
 *   orc_rna / orc_revc  CROPSR.py:124-129 / :116-121 (chained str.replace + [::-1])
 *   orc_window          CROPSR.py:420-421, :431-432 (slices, Python truncation) and
 *                       :458 (replace('U','T').upper())
 *   orc_score30         CROPSR.py:285-313 (rs1_score) with the weight constants of
 *                       CROPSR.py:161-283
 *   orc_exp             numpy exp -> glibc 2.35 exp() (pinned `libm` environment)
 *
 * Third-party arithmetic the reference reaches through numpy (not under
 * /root/reference, versions not pinned by the reference; this container:
 * numpy 2.2.6 + OpenBLAS 0.3.29 SkylakeX/Haswell dgemv_t kernel, glibc 2.35):
 * the two np.matmul calls accumulate in four lanes by (flat index mod 4) and
 * combine as (l0+l2)+(l1+l3); exp is the table-driven N=128
 * algorithm with FMA contraction as glibc's x86-64 FMA ifunc variant performs
 * it.  Both restated below and checked against the real thing.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ weights */
/* Independent transcription of the non-zero entries of first_matrix
 * (CROPSR.py:165-188) and second_matrix (CROPSR.py:190-283).  Key = base(s) +
 * 1-based position in the 30-mer; flat index = (pos-1)*4 + b or
 * (pos-1)*16 + b1*4 + b2 with the alphabet order A,T,C,G of CROPSR.py:300-302. */
typedef struct { const char *key; double w; } orc_term;

static const orc_term ORC_FIRST[] = {
    {"G02", -0.2753771},  {"A03", -0.3238875},  {"C03", 0.17212887},  {"C04", -0.1006662},
    {"C05", -0.2018029},  {"G05", 0.24595663},  {"A06", 0.03644004},  {"C06", 0.09837684},
    {"C07", -0.7411813},  {"G07", -0.3932644},  {"A12", -0.466099},   {"A15", 0.08537695},
    {"C15", -0.013814},   {"A16", 0.27262051},  {"C16", 0.1190226},   {"T16", -0.2859442},
    {"A17", 0.09745459},  {"G17", -0.1755462},  {"C18", -0.3457955},  {"G18", -0.6780964},
    {"A19", 0.22508903},  {"C19", -0.5077941},  {"G20", -0.4173736},  {"T20", -0.054307},
    {"G21", 0.37989937},  {"T21", -0.0907126},  {"C22", 0.05782332},  {"T22", -0.5305673},
    {"T23", -0.8770074},  {"C24", -0.8762358},  {"G24", 0.27891626},  {"T24", -0.4031022},
    {"A25", -0.0773007},  {"C25", 0.28793562},  {"T25", -0.2216372},  {"G28", -0.6890167},
    {"T28", 0.11787758},  {"C29", -0.1604453},  {"G30", 0.38634258},
};
static const orc_term ORC_SECOND[] = {
    {"GT02", -0.6257787}, {"GC05", 0.30004332}, {"AA06", -0.8348362}, {"TA06", 0.76062777},
    {"GG07", -0.4908167}, {"GG12", -1.5169074}, {"TA12", 0.7092612},  {"TC12", 0.49629861},
    {"TT12", -0.5868739}, {"GG13", -0.3345637}, {"GA14", 0.76384993}, {"GC14", -0.5370252},
    {"TG17", -0.7981461}, {"GG19", -0.6668087}, {"TC19", 0.35318325}, {"CC20", 0.74807209},
    {"TG20", -0.3672668}, {"AC21", 0.56820913}, {"CG21", 0.32907207}, {"GA21", -0.8364568},
    {"GG21", -0.7822076}, {"TC22", -1.029693},  {"CG23", 0.85619782}, {"CT23", -0.4632077},
    {"AA24", -0.5794924}, {"AG24", 0.64907554}, {"AG25", -0.0773007}, {"CG25", 0.28793562},
    {"TG25", -0.2216372}, {"GT27", 0.11787758}, {"GG29", -0.69774},
};
static const double ORC_INTERSECT = 0.59763615; /* CROPSR.py:161 */
static const double ORC_LOW_GC = -0.2026259;    /* CROPSR.py:162; added unconditionally at :312 */

static double W1[120];
static double W2[464];
static int weights_ready;

static int base_index(char c)
{
    switch (c) { case 'A': return 0; case 'T': return 1; case 'C': return 2; case 'G': return 3; }
    return -1;
}

static void build_weights(void)
{
    if (weights_ready) return;
    memset(W1, 0, sizeof W1);
    memset(W2, 0, sizeof W2);
    for (size_t k = 0; k < sizeof ORC_FIRST / sizeof ORC_FIRST[0]; ++k) {
        const char *s = ORC_FIRST[k].key;
        int pos = (s[1] - '0') * 10 + (s[2] - '0');
        W1[(pos - 1) * 4 + base_index(s[0])] = ORC_FIRST[k].w;
    }
    for (size_t k = 0; k < sizeof ORC_SECOND / sizeof ORC_SECOND[0]; ++k) {
        const char *s = ORC_SECOND[k].key;
        int pos = (s[2] - '0') * 10 + (s[3] - '0');
        W2[(pos - 1) * 16 + base_index(s[0]) * 4 + base_index(s[1])] = ORC_SECOND[k].w;
    }
    weights_ready = 1;
}

/* Dense tables for the tests to compare with tests/golden/weights.npz. */
void orc_weights(double *first120, double *second464, double *consts2)
{
    build_weights();
    memcpy(first120, W1, sizeof W1);
    memcpy(second464, W2, sizeof W2);
    consts2[0] = ORC_INTERSECT;
    consts2[1] = ORC_LOW_GC;
}

/* --------------------------------------------------------------------- exp */
static const uint64_t EXP_TAB[256] = {
#include "exp_table.inc"
};

static inline uint64_t as_u64(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double as_f64(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* glibc 2.35 exp(), x86-64 FMA variant, for |x| in [2^-54, 512) (the score path
 * stays within (-10, 18)); outside that range this defers to libm. */
double orc_exp(double x)
{
    const double inv_ln2_n = 0x1.71547652b82fep0 * 128;
    const double neg_ln2_hi_n = -0x1.62e42fefa0000p-8;
    const double neg_ln2_lo_n = -0x1.cf79abc9e3b3ap-47;
    const double shift = 0x1.8p52;
    const double c2 = 0x1.ffffffffffdbdp-2, c3 = 0x1.555555555543cp-3;
    const double c4 = 0x1.55555cf172b91p-5, c5 = 0x1.1111167a4d017p-7;
    double ax = fabs(x);
    if (!(ax >= 0x1p-54 && ax < 512.0)) return exp(x);
    double z = inv_ln2_n * x;
    double kd = z + shift;
    uint64_t ki = as_u64(kd);
    kd -= shift;
    double r = fma(kd, neg_ln2_lo_n, fma(kd, neg_ln2_hi_n, x));
    uint64_t idx = 2 * (ki % 128);
    uint64_t top = ki << (52 - 7);
    double tail = as_f64(EXP_TAB[idx]);
    uint64_t sbits = EXP_TAB[idx + 1] + top;
    double r2 = r * r;
    double p_lo = fma(r, c3, c2);
    double p_hi = fma(r, c5, c4);
    double tmp = fma(r2, p_lo, tail + r);
    tmp = fma(r2 * r2, p_hi, tmp);
    double scale = as_f64(sbits);
    return fma(scale, tmp, scale);
}

void orc_exp_many(const double *x, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = orc_exp(x[i]);
}

/* The table path alone, for |x| < 512, WITHOUT glibc's early return for |x| < 2^-54: the HIP
 * scorer (cropsr_amd/csrc/crp_score.h crp_exp) runs exactly this, on the argument that the
 * early return only protects the exception flags and never changes the value.
 * tests/test_oracle.py checks that claim against libm on tiny, denormal and zero arguments. */
void orc_exp_table_path_many(const double *x, int64_t n, double *out)
{
    const double inv_ln2_n = 0x1.71547652b82fep0 * 128;
    const double neg_ln2_hi_n = -0x1.62e42fefa0000p-8;
    const double neg_ln2_lo_n = -0x1.cf79abc9e3b3ap-47;
    const double shift = 0x1.8p52;
    const double c2 = 0x1.ffffffffffdbdp-2, c3 = 0x1.555555555543cp-3;
    const double c4 = 0x1.55555cf172b91p-5, c5 = 0x1.1111167a4d017p-7;
    for (int64_t i = 0; i < n; ++i) {
        const double v = x[i];
        double z = inv_ln2_n * v;
        double kd = z + shift;
        uint64_t ki = as_u64(kd);
        kd -= shift;
        double r = fma(kd, neg_ln2_lo_n, fma(kd, neg_ln2_hi_n, v));
        uint64_t idx = 2 * (ki % 128);
        uint64_t top = ki << (52 - 7);
        double tail = as_f64(EXP_TAB[idx]);
        uint64_t sbits = EXP_TAB[idx + 1] + top;
        double r2 = r * r;
        double p_lo = fma(r, c3, c2);
        double p_hi = fma(r, c5, c4);
        double tmp = fma(r2, p_lo, tail + r);
        tmp = fma(r2 * r2, p_hi, tmp);
        double scale = as_f64(sbits);
        out[i] = fma(scale, tmp, scale);
    }
}

/* ------------------------------------------------------------------- score */
/* rs1_score on ONE row of 30 bytes (CROPSR.py:285-313).  One-hot compare with
 * 'A','T','C','G' (65,84,67,71) only: any other byte selects no column. */
static void score_row(const uint8_t *t, double *pre_out, double *score_out)
{
    double l1[4] = {0.0, 0.0, 0.0, 0.0}, l2[4] = {0.0, 0.0, 0.0, 0.0};
    int code[30];
    for (int p = 0; p < 30; ++p) code[p] = base_index((char)t[p]);
    /* matmul(matrix1, first_matrix): flat index p*4+b, lane = b, ascending */
    for (int p = 0; p < 30; ++p)
        for (int b = 0; b < 4; ++b)
            l1[b] += (code[p] == b ? 1.0 : 0.0) * W1[p * 4 + b];
    double s1 = (l1[0] + l1[2]) + (l1[1] + l1[3]);
    /* matmul(matrix2, second_matrix): flat index p*16+b1*4+b2, lane = b2 */
    for (int p = 0; p < 29; ++p)
        for (int b1 = 0; b1 < 4; ++b1)
            for (int b2 = 0; b2 < 4; ++b2)
                l2[b2] += ((code[p] == b1 && code[p + 1] == b2) ? 1.0 : 0.0) * W2[p * 16 + b1 * 4 + b2];
    double s2 = (l2[0] + l2[2]) + (l2[1] + l2[3]);
    /* CROPSR.py:312  (score_first + score_second + intersect + low_gc) * -1 */
    double pre = (((s1 + s2) + ORC_INTERSECT) + ORC_LOW_GC) * -1.0;
    *pre_out = pre;
    /* CROPSR.py:313 */
    *score_out = 1.0 / (1.0 + orc_exp(pre));
}

void orc_score30(const uint8_t *rows, int64_t n, double *pre, double *score)
{
    build_weights();
    for (int64_t i = 0; i < n; ++i) score_row(rows + 30 * i, pre + i, score + i);
}

/* ---- the other accumulation orders of the reference's BLAS --------------------
 * np.matmul(one_hot[n,K], w[K]) reaches OpenBLAS dgemv_t (0.3.29, SkylakeX
 * runtime core, one thread).  Its driver (kernel/x86_64/dgemv_t_4.c) hands rows
 * four at a time to the AVX2 4x4 micro-kernel (order above), then a leftover
 * PAIR of rows to dgemv_kernel_4x2 (SSE2: one 2-lane accumulator per row, so the
 * lanes are flat index mod 2, combined by haddpd = l0 + l1), then a leftover
 * single row to dgemv_kernel_4x1 (two 2-lane accumulators = the 4-lane order
 * again).  A batch of exactly one row never reaches dgemv: numpy calls ddot,
 * whose SkylakeX kernel (kernel/x86_64/ddot_microk_skylakex-2.c + ddot.c) runs
 * 4 x 8 lanes over blocks of 32, folds them to 4 x 4, does the remaining block
 * of 16 in 4 x 4 lanes, adds the four accumulators left to right, folds halves,
 * hadds, and finishes the last K mod 16 elements one by one.
 * Measured against the real reference on thousands of rows per case.
 */
static void one_hot(const uint8_t *t, double *v1, double *v2)
{
    int code[30];
    for (int p = 0; p < 30; ++p) code[p] = base_index((char)t[p]);
    for (int p = 0; p < 30; ++p)
        for (int b = 0; b < 4; ++b) v1[p * 4 + b] = code[p] == b ? 1.0 : 0.0;
    for (int p = 0; p < 29; ++p)
        for (int b1 = 0; b1 < 4; ++b1)
            for (int b2 = 0; b2 < 4; ++b2)
                v2[p * 16 + b1 * 4 + b2] = (code[p] == b1 && code[p + 1] == b2) ? 1.0 : 0.0;
}

static double dot_tail2(const double *v, const double *w, int k)
{
    double l[2] = {0.0, 0.0};
    for (int i = 0; i < k; ++i) l[i & 1] += v[i] * w[i];
    return l[0] + l[1];
}

static double dot_skylakex(const double *v, const double *w, int k)
{
    const int k16 = k & -16, k32 = k16 & ~31;
    double wide[4][8] = {{0}}, acc[4][4], col[4], d;
    int i = 0;
    for (; i < k32; i += 32)
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 8; ++j) wide[q][j] += v[i + 8 * q + j] * w[i + 8 * q + j];
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 4; ++j) acc[q][j] = wide[q][j] + wide[q][j + 4];
    for (; i < k16; i += 16)
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 4; ++j) acc[q][j] += v[i + 4 * q + j] * w[i + 4 * q + j];
    for (int j = 0; j < 4; ++j) col[j] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
    d = (col[0] + col[2]) + (col[1] + col[3]);
    for (; i < k; ++i) d += v[i] * w[i];
    return d;
}

/* order: 0 = 4-lane body (same as orc_score30), 1 = 2-lane pair tail, 2 = single-row ddot */
void orc_score30_order(const uint8_t *rows, int64_t n, int order, double *pre, double *score)
{
    build_weights();
    for (int64_t i = 0; i < n; ++i) {
        if (order == 0) { score_row(rows + 30 * i, pre + i, score + i); continue; }
        double v1[120], v2[464], s1, s2;
        one_hot(rows + 30 * i, v1, v2);
        if (order == 1) { s1 = dot_tail2(v1, W1, 120); s2 = dot_tail2(v2, W2, 464); }
        else { s1 = dot_skylakex(v1, W1, 120); s2 = dot_skylakex(v2, W2, 464); }
        pre[i] = (((s1 + s2) + ORC_INTERSECT) + ORC_LOW_GC) * -1.0;
        score[i] = 1.0 / (1.0 + orc_exp(pre[i]));
    }
}

/* rs1_score on a whole batch exactly as the reference computes it: the order of
 * each row follows from its position in the batch. */
void orc_rs1_batch(const uint8_t *rows, int64_t n, double *pre, double *score)
{
    if (n == 1) { orc_score30_order(rows, 1, 2, pre, score); return; }
    const int64_t body = 4 * (n / 4);
    orc_score30_order(rows, body, 0, pre, score);
    int64_t k = body;
    if ((n & 3) >= 2) { orc_score30_order(rows + 30 * k, 2, 1, pre + k, score + k); k += 2; }
    if (k < n) orc_score30_order(rows + 30 * k, 1, 0, pre + k, score + k);
}

/* ------------------------------------------------------- string transforms */
/* A chain of str.replace(a, b) calls acts on every character independently, so
 * it is a byte map obtained by pushing each byte through the chain. */
static uint8_t MAP_RNA[256];  /* get_gRNA_sequence, CROPSR.py:128, before [::-1] */
static uint8_t MAP_REVC[256]; /* get_reverse_complement, CROPSR.py:120, before [::-1] */
static int maps_ready;

static uint8_t push(uint8_t c, const char *chain)
{
    for (const char *p = chain; p[0]; p += 2)
        if (c == (uint8_t)p[0]) c = (uint8_t)p[1];
    return c;
}

static void build_maps(void)
{
    if (maps_ready) return;
    for (int c = 0; c < 256; ++c) {
        MAP_RNA[c] = push((uint8_t)c, "AUCZGCZGTA");
        MAP_REVC[c] = push((uint8_t)c, "AUCZGCZGTAUT");
    }
    maps_ready = 1;
}

static void map_reverse(const uint8_t *in, int64_t n, const uint8_t *map, uint8_t *out)
{
    for (int64_t k = 0; k < n; ++k) out[k] = map[in[n - 1 - k]];
}

/* Python slice s[a:b] with 0 <= a: clamps b to len. */
static int64_t py_slice(const uint8_t *s, int64_t len, int64_t a, int64_t b, uint8_t *out)
{
    if (a < 0) a = 0; /* never negative on kept hits; guard only */
    if (b > len) b = len;
    if (b <= a) return 0;
    memcpy(out, s + a, (size_t)(b - a));
    return b - a;
}

/* long_sequence of one kept hit, as bytes.  strand '+': CROPSR.py:421; '-': :432.
 * Returns its length (30 when complete). */
int64_t orc_long_sequence(const uint8_t *s, int64_t len, int64_t pos, int strand_minus, int l, uint8_t *out)
{
    uint8_t buf[512], tmp[512];
    build_maps();
    if (!strand_minus) {
        int64_t a = pos - l, b = pos; /* pam_location, CROPSR.py:418 */
        int64_t n = py_slice(s, len, a - 5, b + 5, buf);
        map_reverse(buf, n, MAP_RNA, out);
        return n;
    }
    int64_t a = pos + 3, b = pos + 3 + l; /* CROPSR.py:429 */
    int64_t n = py_slice(s, len, a - 5, b + 5, buf);
    map_reverse(buf, n, MAP_REVC, tmp);
    map_reverse(tmp, n, MAP_RNA, out);
    return n;
}

/* sequence (short guide) of one kept hit: CROPSR.py:420 / :431. */
int64_t orc_short_sequence(const uint8_t *s, int64_t len, int64_t pos, int strand_minus, int l, uint8_t *out)
{
    uint8_t buf[512], tmp[512];
    build_maps();
    if (!strand_minus) {
        int64_t n = py_slice(s, len, pos - l, pos, buf);
        map_reverse(buf, n, MAP_RNA, out);
        return n;
    }
    int64_t n = py_slice(s, len, pos + 3, pos + 3 + l, buf);
    map_reverse(buf, n, MAP_REVC, tmp);
    map_reverse(tmp, n, MAP_RNA, out);
    return n;
}

/* -------------------------------------------------------------------- scan */
/* Kept hits of one contig string, in the reference's order (all '+' ascending,
 * then all '-' ascending).  pos is the regex match index (target[0]):
 *   '+'  (?=.GG)  CROPSR.py:415-423   start=pos-l, end=pos
 *   '-'  (?=CC.)  CROPSR.py:426-434   start=pos+3+l, end=pos+3
 * Either output pointer may be NULL (count only).  l <= 200. */
void orc_scan(const uint8_t *s, int64_t len, int l, uint32_t *plus, int64_t *n_plus,
              uint32_t *minus, int64_t *n_minus)
{
    int64_t np = 0, nm = 0;
    for (int64_t i = 0; i + 2 < len; ++i) {
        /* '.' matches any character except newline; contig strings hold none,
         * but keep the regex's rule. */
        if (s[i] != '\n' && s[i + 1] == 'G' && s[i + 2] == 'G') {
            int64_t a = i - l, b = i;
            if (a >= 5 && a + 5 <= len + 10 && b >= 5 && b <= len + 10) { /* :419 */
                if (plus) plus[np] = (uint32_t)i;
                ++np;
            }
        }
    }
    for (int64_t j = 0; j + 2 < len; ++j) {
        if (s[j] == 'C' && s[j + 1] == 'C' && s[j + 2] != '\n') {
            int64_t a = j + 3, b = j + 3 + l;
            if (a >= 5 && a + 5 <= len + 10 && b >= 5 && b <= len + 10) { /* :430 */
                if (minus) minus[nm] = (uint32_t)j;
                ++nm;
            }
        }
    }
    *n_plus = np;
    *n_minus = nm;
}

/* Score the kept hits of one contig string.  Rows whose long_sequence is not
 * 30 characters get score = pre = -1 (CROPSR.py:466-468 writes -1 for them). */
void orc_score_hits(const uint8_t *s, int64_t len, int l, const uint32_t *pos, int64_t n,
                    int strand_minus, double *pre, double *score)
{
    uint8_t lng[512], t[512];
    build_weights();
    for (int64_t k = 0; k < n; ++k) {
        int64_t m = orc_long_sequence(s, len, pos[k], strand_minus, l, lng);
        if (m != 30) { pre[k] = -1.0; score[k] = -1.0; continue; }
        /* CROPSR.py:458: replace('U','T') then upper() (ASCII letters) */
        for (int p = 0; p < 30; ++p) {
            uint8_t c = lng[p];
            if (c == 'U') c = 'T';
            if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
            t[p] = c;
        }
        score_row(t, pre + k, score + k);
    }
}

/* ---------------------------------------------------------------- off-target
 * The off-target seed scan has NO counterpart in the reference: parity
 * is unpinned by construction.  What is pinned is the INPUT of the definition: it is stated on the
 * reference's own `sequence` column (CROPSR.py:420 / :431), which this file restates literally
 * (orc_short_sequence) and which tests/test_offtarget.py also takes from the reference's golden CSVs.
 *
 *   seed(g)  = the first 12 characters of sequence(g) after .replace('U','T').upper() (the scoring
 *              transform of CROPSR.py:458), defined iff there are 12 and all are in ACGT
 *   c_k(g)   = #{ sites s != g : hamming(seed(g), seed(s)) == k },  k = 0..3
 *
 * orc_seed_codes     seed -> 24-bit code, sum of code_k << 2k, alphabet order A,T,C,G of CROPSR.py:300
 * orc_offtarget_pairs    the definition, all pairs: O(n^2)
 * orc_offtarget_enum     an independent second method (histogram + enumeration of the 6 571 seeds
 *                        within distance 3 of each distinct seed): for inputs too big for all pairs;
 *                        tests/test_offtarget.py checks it against orc_offtarget_pairs first
 */
#define ORC_NOT_A_SITE 0xffffffffu

void orc_seed_codes(const uint8_t *s, int64_t len, int l, const uint32_t *pos, int64_t n, int strand_minus,
                    uint32_t *codes)
{
    uint8_t buf[512];
    for (int64_t i = 0; i < n; ++i) {
        const int64_t m = orc_short_sequence(s, len, pos[i], strand_minus, l, buf);
        uint32_t code = 0;
        int ok = m >= 12;
        for (int k = 0; ok && k < 12; ++k) {
            uint8_t c = buf[k];
            if (c == 'U') c = 'T';                       /* .replace('U','T') */
            if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32); /* .upper() */
            const int b = base_index((char)c);
            if (b < 0) ok = 0;
            else code |= (uint32_t)b << (2 * k);
        }
        codes[i] = ok ? code : ORC_NOT_A_SITE;
    }
}

static inline int seed_distance(uint32_t a, uint32_t b)
{
    const uint32_t x = a ^ b;
    return __builtin_popcount((x | (x >> 1)) & 0x555555u);
}

/* counts: n x 4 */
void orc_offtarget_pairs(const uint32_t *codes, int64_t n, uint32_t *counts)
{
    for (int64_t i = 0; i < n; ++i) {
        uint32_t c[4] = {0, 0, 0, 0};
        if (codes[i] == ORC_NOT_A_SITE) {
            for (int k = 0; k < 4; ++k) counts[4 * i + k] = ORC_NOT_A_SITE;
            continue;
        }
        for (int64_t j = 0; j < n; ++j) {
            if (j == i || codes[j] == ORC_NOT_A_SITE) continue;
            const int d = seed_distance(codes[i], codes[j]);
            if (d <= 3) c[d]++;
        }
        for (int k = 0; k < 4; ++k) counts[4 * i + k] = c[k];
    }
}

/* hist: 4^12 site counts (the caller may have summed several inputs into it).  counts of the codes
 * given, minus the guide itself. */
void orc_offtarget_hist_add(const uint32_t *codes, int64_t n, uint32_t *hist)
{
    for (int64_t i = 0; i < n; ++i)
        if (codes[i] != ORC_NOT_A_SITE) hist[codes[i]]++;
}

static void ball_of(uint32_t seed, const uint32_t *hist, uint32_t c[4])
{
    c[0] = hist[seed];
    c[1] = c[2] = c[3] = 0;
    for (int p1 = 0; p1 < 12; ++p1)
        for (uint32_t b1 = 1; b1 < 4; ++b1) {
            const uint32_t s1 = seed ^ (b1 << (2 * p1));
            c[1] += hist[s1];
            for (int p2 = p1 + 1; p2 < 12; ++p2)
                for (uint32_t b2 = 1; b2 < 4; ++b2) {
                    const uint32_t s2 = s1 ^ (b2 << (2 * p2));
                    c[2] += hist[s2];
                    for (int p3 = p2 + 1; p3 < 12; ++p3)
                        for (uint32_t b3 = 1; b3 < 4; ++b3) c[3] += hist[s2 ^ (b3 << (2 * p3))];
                }
        }
}

void orc_offtarget_enum(const uint32_t *codes, int64_t n, const uint32_t *hist, uint32_t *counts)
{
    for (int64_t i = 0; i < n; ++i) {
        if (codes[i] == ORC_NOT_A_SITE) {
            for (int k = 0; k < 4; ++k) counts[4 * i + k] = ORC_NOT_A_SITE;
            continue;
        }
        uint32_t c[4];
        ball_of(codes[i], hist, c);
        c[0] -= 1;
        for (int k = 0; k < 4; ++k) counts[4 * i + k] = c[k];
    }
}
