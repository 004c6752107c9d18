/*
 * oracle/oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See oracle.h for scope, provenance and the parity pin.
 *
 * Everything here is written from scratch. Reference citations are
 * `path:line` relative to the deciphon-old tree; "imm" marks behaviour of the
 * absent third-party library EBI-Metagenomics/imm v2.0.3 restated from its
 * published algorithm and validated by goldens G1-G3.
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORC_F64
#define O_LOG(x) log(x)
#define O_EXP(x) exp(x)
#define O_LOG1P(x) log1p(x)
#else
#define O_LOG(x) logf(x)
#define O_EXP(x) expf(x)
#define O_LOG1P(x) log1pf(x)
#endif

#define NEG_INF ((ofloat)(-INFINITY))

int orc_float_bytes(void) { return (int)sizeof(ofloat); }

/* ======================================================================== */
/* RNG (imm_rnd): xoshiro256+ seeded through splitmix64, doubles from the    */
/* top 53 bits. Identified by search against golden G1                        */
/* (test/protein_profile.c:41): this generator reproduces it to 3e-11; the   */
/* other candidates tried (PCG, xorshift128+, xoroshiro128+ and 128-starstar, */
/* xoshiro256-starstar, splitmix64) miss by >= 4e-3.                          */
/* ======================================================================== */
static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static inline uint64_t rotl64(uint64_t x, int k)
{
    return (x << k) | (x >> (64 - k));
}

void orc_rnd_seed(struct orc_rnd *r, uint64_t seed)
{
    for (int i = 0; i < 4; ++i)
        r->s[i] = splitmix64(&seed);
}

uint64_t orc_rnd_u64(struct orc_rnd *r)
{
    uint64_t *s = r->s;
    uint64_t const result = s[0] + s[3];
    uint64_t const t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

double orc_rnd_dbl(struct orc_rnd *r)
{
    return (double)(orc_rnd_u64(r) >> 11) * 0x1.0p-53;
}

/* ======================================================================== */
/* lprob helpers (imm: imm_lprob_add = logaddexp, chained sums)              */
/* ======================================================================== */
ofloat orc_logaddexp(ofloat x, ofloat y)
{
    if (x == y) return x + (ofloat)0.69314718055994530942;
    ofloat tmp = x - y;
    if (tmp > 0) return x + O_LOG1P(O_EXP(-tmp));
    if (tmp <= 0) return y + O_LOG1P(O_EXP(tmp));
    return tmp; /* NaN */
}

static ofloat lse3(ofloat a, ofloat b, ofloat c)
{
    return orc_logaddexp(orc_logaddexp(a, b), c);
}

static ofloat lprob_sum(unsigned n, ofloat const *arr)
{
    ofloat r = NEG_INF;
    for (unsigned i = 0; i < n; ++i)
        r = orc_logaddexp(r, arr[i]);
    return r;
}

void orc_lprob_normalize(unsigned n, ofloat *arr)
{
    ofloat s = lprob_sum(n, arr);
    for (unsigned i = 0; i < n; ++i)
        arr[i] -= s;
}

void orc_lprob_sample(struct orc_rnd *r, unsigned n, ofloat *arr)
{
    for (unsigned i = 0; i < n; ++i)
        arr[i] = (ofloat)O_LOG((ofloat)orc_rnd_dbl(r));
}

/* ======================================================================== */
/* Alphabets and genetic code (imm: imm_dna_iupac "ACGT", imm_amino_iupac    */
/* "ACDEFGHIKLMNPQRSTVWY", imm_gc table 1 = NCBI standard code).             */
/* ======================================================================== */
static char const amino_symbols[] = "ACDEFGHIKLMNPQRSTVWY";
static char const gc_base1[] =
    "TTTTTTTTTTTTTTTTCCCCCCCCCCCCCCCCAAAAAAAAAAAAAAAAGGGGGGGGGGGGGGGG";
static char const gc_base2[] =
    "TTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGG";
static char const gc_base3[] =
    "TCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAG";
static char const gc_aa1[] =
    "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";

static unsigned nuclt_idx(char c)
{
    switch (c)
    {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    default: return 3; /* 'T' */
    }
}

/* ======================================================================== */
/* setup_nuclt_dist: protein_model.c:396-408 with codon_lprob :361-394,      */
/* nuclt_lprob :342-359 and imm_codon_marg (imm).                            */
/* ======================================================================== */
void orc_setup_nuclt_dist(struct orc_nuclt_dist *d, ofloat const aa_lprobs[20])
{
    /* codon_lprob(): split each amino acid's lprob over its synonymous codons */
    unsigned count[256];
    memset(count, 0, sizeof count);
    for (unsigned i = 0; i < 64; ++i)
        count[(unsigned char)gc_aa1[i]] += 1;

    ofloat aa_lp[256];
    for (unsigned i = 0; i < 256; ++i)
        aa_lp[i] = NEG_INF;
    for (unsigned i = 0; i < ORC_AMINO_SIZE; ++i)
    {
        unsigned char aa = (unsigned char)amino_symbols[i];
        ofloat norm = (ofloat)O_LOG((ofloat)count[aa]);
        aa_lp[aa] = aa_lprobs[i] - norm;
    }

    ofloat codonp[64]; /* [a][b][c] in ACGT ids */
    for (unsigned i = 0; i < 64; ++i)
        codonp[i] = NEG_INF;
    for (unsigned i = 0; i < 64; ++i)
    {
        unsigned a = nuclt_idx(gc_base1[i]);
        unsigned b = nuclt_idx(gc_base2[i]);
        unsigned c = nuclt_idx(gc_base3[i]);
        codonp[a * 16 + b * 4 + c] = aa_lp[(unsigned char)gc_aa1[i]];
    }
    /* imm_codon_lprob_normalize */
    orc_lprob_normalize(64, codonp);

    /* nuclt_lprob(): each codon gives lprob - log 3 to each of its bases */
    ofloat np[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
    ofloat const norm = (ofloat)O_LOG((ofloat)3);
    for (unsigned i = 0; i < 64; ++i)
    {
        unsigned a = nuclt_idx(gc_base1[i]);
        unsigned b = nuclt_idx(gc_base2[i]);
        unsigned c = nuclt_idx(gc_base3[i]);
        ofloat lp = codonp[a * 16 + b * 4 + c];
        np[a] = orc_logaddexp(np[a], lp - norm);
        np[b] = orc_logaddexp(np[b], lp - norm);
        np[c] = orc_logaddexp(np[c], lp - norm);
    }
    memcpy(d->nucltp, np, sizeof np);

    /* imm_codon_marg: wildcard (index 4) = logsumexp over matching codons */
    for (unsigned a = 0; a < 5; ++a)
        for (unsigned b = 0; b < 5; ++b)
            for (unsigned c = 0; c < 5; ++c)
            {
                ofloat acc = NEG_INF;
                unsigned a0 = a == 4 ? 0 : a, a1 = a == 4 ? 4 : a + 1;
                unsigned b0 = b == 4 ? 0 : b, b1 = b == 4 ? 4 : b + 1;
                unsigned c0 = c == 4 ? 0 : c, c1 = c == 4 ? 4 : c + 1;
                for (unsigned i = a0; i < a1; ++i)
                    for (unsigned j = b0; j < b1; ++j)
                        for (unsigned k = c0; k < c1; ++k)
                            acc = orc_logaddexp(acc,
                                                codonp[i * 16 + j * 4 + k]);
                d->codonm[a * 25 + b * 5 + c] = acc;
            }
}

/* ======================================================================== */
/* Frame-state emission (imm frame_state)                 */
/* ======================================================================== */
#define CM(a, b, c) (cm[(a)*25 + (b)*5 + (c)])
enum { W_ = 4 };

static ofloat frame_lprob(ofloat const *b, ofloat const *cm, ofloat le,
                          ofloat l1, unsigned char const *x, unsigned len)
{
    ofloat const LOG2 = (ofloat)O_LOG((ofloat)2);
    ofloat const LOG3 = (ofloat)O_LOG((ofloat)3);
    ofloat const LOG4 = (ofloat)O_LOG((ofloat)4);
    ofloat const LOG9 = (ofloat)O_LOG((ofloat)9);
    ofloat const LOG10 = (ofloat)O_LOG((ofloat)10);

    if (len == 1)
    {
        unsigned x1 = x[0];
        ofloat c = 2 * le + 2 * l1;
        return c + lse3(CM(x1, W_, W_), CM(W_, x1, W_), CM(W_, W_, x1)) - LOG3;
    }
    if (len == 2)
    {
        unsigned x1 = x[0], x2 = x[1];
        ofloat c0 = LOG2 + le + l1 * 3 - LOG3;
        ofloat v0 =
            c0 + lse3(CM(W_, x1, x2), CM(x1, W_, x2), CM(x1, x2, W_));
        ofloat c1 = 3 * le + l1 - LOG3;
        ofloat v1 = c1 +
                    lse3(CM(x1, W_, W_), CM(W_, x1, W_), CM(W_, W_, x1)) +
                    b[x2];
        ofloat v2 = c1 +
                    lse3(CM(x2, W_, W_), CM(W_, x2, W_), CM(W_, W_, x2)) +
                    b[x1];
        return lse3(v0, v1, v2);
    }
    if (len == 3)
    {
        unsigned x1 = x[0], x2 = x[1], x3 = x[2];
        ofloat v0 = 4 * l1 + CM(x1, x2, x3);
        ofloat c1 = LOG4 + 2 * le + 2 * l1 - LOG9;
        ofloat v1 = c1 +
                    lse3(CM(W_, x2, x3), CM(x2, W_, x3), CM(x2, x3, W_)) +
                    b[x1];
        ofloat v2 = c1 +
                    lse3(CM(W_, x1, x3), CM(x1, W_, x3), CM(x1, x3, W_)) +
                    b[x2];
        ofloat v3 = c1 +
                    lse3(CM(W_, x1, x2), CM(x1, W_, x2), CM(x1, x2, W_)) +
                    b[x3];
        ofloat c2 = 4 * le - LOG9;
        ofloat v4 = c2 +
                    lse3(CM(x3, W_, W_), CM(W_, x3, W_), CM(W_, W_, x3)) +
                    b[x1] + b[x2];
        ofloat v5 = c2 +
                    lse3(CM(x2, W_, W_), CM(W_, x2, W_), CM(W_, W_, x2)) +
                    b[x1] + b[x3];
        ofloat v6 = c2 +
                    lse3(CM(x1, W_, W_), CM(W_, x1, W_), CM(W_, W_, x1)) +
                    b[x2] + b[x3];
        ofloat r = orc_logaddexp(v0, v1);
        r = orc_logaddexp(r, v2);
        r = orc_logaddexp(r, v3);
        r = orc_logaddexp(r, v4);
        r = orc_logaddexp(r, v5);
        r = orc_logaddexp(r, v6);
        return r;
    }
    if (len == 4)
    {
        unsigned x1 = x[0], x2 = x[1], x3 = x[2], x4 = x[3];
        ofloat v0 = orc_logaddexp(
            orc_logaddexp(b[x1] + CM(x2, x3, x4), b[x2] + CM(x1, x3, x4)),
            orc_logaddexp(b[x3] + CM(x1, x2, x4), b[x4] + CM(x1, x2, x3)));
        v0 += le + 3 * l1 - LOG2;
        ofloat c = 3 * le + l1 - LOG9;
        /* pairs (a<b) inserted, (c,d) = the two remaining in order */
        unsigned xs[4] = {x1, x2, x3, x4};
        ofloat acc = NEG_INF;
        for (unsigned i = 0; i < 4; ++i)
            for (unsigned j = i + 1; j < 4; ++j)
            {
                unsigned rest[2], n = 0;
                for (unsigned k = 0; k < 4; ++k)
                    if (k != i && k != j) rest[n++] = xs[k];
                ofloat t = b[xs[i]] + b[xs[j]] +
                           lse3(CM(W_, rest[0], rest[1]),
                                CM(rest[0], W_, rest[1]),
                                CM(rest[0], rest[1], W_));
                acc = orc_logaddexp(acc, t);
            }
        return orc_logaddexp(v0, c + acc);
    }
    /* len == 5 */
    {
        ofloat c = 2 * le + 2 * l1 - LOG10;
        ofloat acc = NEG_INF;
        for (unsigned i = 0; i < 5; ++i)
            for (unsigned j = i + 1; j < 5; ++j)
            {
                unsigned rest[3], n = 0;
                for (unsigned k = 0; k < 5; ++k)
                    if (k != i && k != j) rest[n++] = x[k];
                ofloat t =
                    b[x[i]] + b[x[j]] + CM(rest[0], rest[1], rest[2]);
                acc = orc_logaddexp(acc, t);
            }
        return c + acc;
    }
}

static unsigned const code_off[6] = {0, 0, 4, 20, 84, 340};

unsigned orc_word_code(unsigned char const *x, unsigned len)
{
    unsigned v = 0;
    for (unsigned i = 0; i < len; ++i)
        v = v * 4 + x[i];
    return code_off[len] + v;
}

void orc_frame_table(struct orc_nuclt_dist const *d, ofloat eps, ofloat *tbl)
{
    ofloat le = (ofloat)O_LOG(eps);
    ofloat l1 = (ofloat)O_LOG((ofloat)1 - eps);
    for (unsigned len = 1; len <= 5; ++len)
    {
        unsigned n = 1u << (2 * len);
        for (unsigned v = 0; v < n; ++v)
        {
            unsigned char x[5];
            for (unsigned i = 0; i < len; ++i)
                x[i] = (unsigned char)((v >> (2 * (len - 1 - i))) & 3);
            tbl[code_off[len] + v] =
                frame_lprob(d->nucltp, d->codonm, le, l1, x, len);
        }
    }
}

/* ======================================================================== */
/* Generic HMM graph (imm_hmm / imm_dp restatement)                           */
/* ======================================================================== */
struct ostate
{
    unsigned id;       /* protein_state id */
    int emitting;      /* frame state (span 1..5) vs mute (span 0) */
    ofloat const *tbl; /* 1364-entry emission table or NULL */
};

struct otrans
{
    int src, dst;
    ofloat lp;
};

struct ohmm
{
    int nstates;
    struct ostate *states;
    int ntrans, cap_trans;
    struct otrans *trans;
    int start;
    ofloat start_lp;
    int end;
    /* compiled ("imm_hmm_reset_dp") */
    int *order;    /* emitting states first, then mute in topological order */
    int n_emit;
    int *in_begin; /* CSR over incoming transitions, by dst */
    int *in_idx;   /* indices into trans[] */
};

static int hmm_add_state(struct ohmm *h, unsigned id, int emitting,
                         ofloat const *tbl)
{
    h->states = realloc(h->states, (size_t)(h->nstates + 1) * sizeof *h->states);
    h->states[h->nstates] = (struct ostate){id, emitting, tbl};
    return h->nstates++;
}

/* imm_hmm_set_trans: creates or overwrites */
static void hmm_set_trans(struct ohmm *h, int src, int dst, ofloat lp)
{
    for (int i = 0; i < h->ntrans; ++i)
        if (h->trans[i].src == src && h->trans[i].dst == dst)
        {
            h->trans[i].lp = lp;
            return;
        }
    if (h->ntrans == h->cap_trans)
    {
        h->cap_trans = h->cap_trans ? h->cap_trans * 2 : 64;
        h->trans = realloc(h->trans, (size_t)h->cap_trans * sizeof *h->trans);
    }
    h->trans[h->ntrans++] = (struct otrans){src, dst, lp};
}

static int hmm_trans_idx(struct ohmm const *h, int src, int dst)
{
    for (int i = 0; i < h->ntrans; ++i)
        if (h->trans[i].src == src && h->trans[i].dst == dst) return i;
    return -1;
}

static void hmm_compile(struct ohmm *h)
{
    int n = h->nstates;
    h->order = malloc((size_t)n * sizeof(int));
    h->in_begin = calloc((size_t)n + 1, sizeof(int));
    h->in_idx = malloc((size_t)(h->ntrans ? h->ntrans : 1) * sizeof(int));
    for (int i = 0; i < h->ntrans; ++i)
        h->in_begin[h->trans[i].dst + 1]++;
    for (int i = 0; i < n; ++i)
        h->in_begin[i + 1] += h->in_begin[i];
    int *fill = calloc((size_t)n, sizeof(int));
    for (int i = 0; i < h->ntrans; ++i)
    {
        int d = h->trans[i].dst;
        h->in_idx[h->in_begin[d] + fill[d]++] = i;
    }
    free(fill);

    int k = 0;
    for (int i = 0; i < n; ++i)
        if (h->states[i].emitting) h->order[k++] = i;
    h->n_emit = k;
    /* Kahn topological sort over the mute sub-graph, stable in add order */
    int *indeg = calloc((size_t)n, sizeof(int));
    for (int i = 0; i < h->ntrans; ++i)
    {
        struct otrans const *t = &h->trans[i];
        if (!h->states[t->src].emitting && !h->states[t->dst].emitting)
            indeg[t->dst]++;
    }
    char *done = calloc((size_t)n, 1);
    int placed = k;
    while (placed < n)
    {
        int progressed = 0;
        for (int i = 0; i < n; ++i)
        {
            if (h->states[i].emitting || done[i] || indeg[i] != 0) continue;
            done[i] = 1;
            h->order[placed++] = i;
            progressed = 1;
            for (int j = 0; j < h->ntrans; ++j)
                if (h->trans[j].src == i && !h->states[h->trans[j].dst].emitting)
                    indeg[h->trans[j].dst]--;
        }
        if (!progressed)
        {
            fprintf(stderr, "oracle: mute cycle in HMM\n");
            abort();
        }
    }
    free(indeg);
    free(done);
}

static void hmm_free(struct ohmm *h)
{
    free(h->states);
    free(h->trans);
    free(h->order);
    free(h->in_begin);
    free(h->in_idx);
}

/* ======================================================================== */
/* Profile = protein_model + protein_profile restatement                     */
/* ======================================================================== */
struct orc_profile
{
    unsigned M;
    int entry_dist;
    ofloat eps;
    ofloat null_lprobs[20];
    struct orc_nuclt_dist null_d, insert_d, *match_d;
    ofloat *trans; /* [M+1][7] */
    ofloat *locc;
    ofloat *tbl_null, *tbl_insert, *tbl_match; /* [1364], [1364], [M][1364] */
    struct ohmm null, alt;
    int R, S, N, B, E, J, C, T;
    int *Mi, *Ii, *Di; /* state indices per node */
    /* transition indices changed by protein_profile_setup */
    int t_RR, t_SB, t_SN, t_NN, t_NB, t_ET, t_EC, t_CC, t_CT, t_EB, t_EJ, t_JJ,
        t_JB;
};

/* log(1 - p) given log(p): protein_model.c:18 (double libm on imm_float) */
static ofloat log1_p(ofloat logp) { return (ofloat)log1p(-exp((double)logp)); }

/* calculate_occupancy: protein_model.c:258-283 */
static void calculate_occupancy(struct orc_profile *p)
{
    unsigned M = p->M;
    ofloat const *t = p->trans; /* trans[i] = t + 7*i : MM MI MD IM II DM DD */
    p->locc[0] = orc_logaddexp(t[1], t[0]);
    for (unsigned i = 1; i < M; ++i)
    {
        t += 7;
        ofloat v0 = p->locc[i - 1] + orc_logaddexp(t[0], t[1]);
        ofloat v1 = log1_p(p->locc[i - 1]) + t[5];
        p->locc[i] = orc_logaddexp(v0, v1);
    }
    ofloat logZ = NEG_INF;
    for (unsigned i = 0; i < M; ++i)
        logZ = orc_logaddexp(logZ, p->locc[i] + (ofloat)O_LOG((ofloat)(M - i)));
    for (unsigned i = 0; i < M; ++i)
        p->locc[i] -= logZ;
}

static void build_models(struct orc_profile *p)
{
    unsigned M = p->M;
    /* emission tables (imm precomputes them per state at imm_hmm_reset_dp) */
    p->tbl_null = malloc(sizeof(ofloat) * ORC_NCODES);
    p->tbl_insert = malloc(sizeof(ofloat) * ORC_NCODES);
    p->tbl_match = malloc(sizeof(ofloat) * ORC_NCODES * (size_t)M);
    orc_frame_table(&p->null_d, p->eps, p->tbl_null);
    orc_frame_table(&p->insert_d, p->eps, p->tbl_insert);
    for (unsigned k = 0; k < M; ++k)
        orc_frame_table(&p->match_d[k], p->eps,
                        p->tbl_match + (size_t)k * ORC_NCODES);

    /* null model: add_xnodes protein_model.c:223-225, init_null_xtrans :316 */
    struct ohmm *h = &p->null;
    memset(h, 0, sizeof *h);
    p->R = hmm_add_state(h, ORC_R_STATE, 1, p->tbl_null);
    h->start = p->R;
    h->start_lp = 0;
    h->end = p->R;
    hmm_set_trans(h, p->R, p->R, 0);
    p->t_RR = hmm_trans_idx(h, p->R, p->R);
    hmm_compile(h);

    /* alt model: add_xnodes :227-234, add_node :49-82 */
    h = &p->alt;
    memset(h, 0, sizeof *h);
    p->S = hmm_add_state(h, ORC_S_STATE, 0, NULL);
    p->N = hmm_add_state(h, ORC_N_STATE, 1, p->tbl_null);
    p->B = hmm_add_state(h, ORC_B_STATE, 0, NULL);
    p->E = hmm_add_state(h, ORC_E_STATE, 0, NULL);
    p->J = hmm_add_state(h, ORC_J_STATE, 1, p->tbl_null);
    p->C = hmm_add_state(h, ORC_C_STATE, 1, p->tbl_null);
    p->T = hmm_add_state(h, ORC_T_STATE, 0, NULL);
    h->start = p->S;
    h->start_lp = 0;
    h->end = p->T;
    p->Mi = malloc(sizeof(int) * M);
    p->Ii = malloc(sizeof(int) * M);
    p->Di = malloc(sizeof(int) * M);
    for (unsigned k = 0; k < M; ++k)
    {
        p->Mi[k] = hmm_add_state(h, ORC_MATCH_STATE | (k + 1), 1,
                                 p->tbl_match + (size_t)k * ORC_NCODES);
        p->Ii[k] = hmm_add_state(h, ORC_INSERT_STATE | (k + 1), 1,
                                 p->tbl_insert);
        p->Di[k] = hmm_add_state(h, ORC_DELETE_STATE | (k + 1), 0, NULL);
    }

    /* setup_transitions: protein_model.c:460-500 */
    ofloat const *tr = p->trans;
    hmm_set_trans(h, p->B, p->Mi[0], tr[0] /* trans[0].MM */);
    for (unsigned i = 0; i + 1 < M; ++i)
    {
        ofloat const *t = tr + 7 * (i + 1);
        hmm_set_trans(h, p->Mi[i], p->Ii[i], t[1]);     /* MI */
        hmm_set_trans(h, p->Ii[i], p->Ii[i], t[4]);     /* II */
        hmm_set_trans(h, p->Mi[i], p->Mi[i + 1], t[0]); /* MM */
        hmm_set_trans(h, p->Ii[i], p->Mi[i + 1], t[3]); /* IM */
        hmm_set_trans(h, p->Mi[i], p->Di[i + 1], t[2]); /* MD */
        hmm_set_trans(h, p->Di[i], p->Di[i + 1], t[6]); /* DD */
        hmm_set_trans(h, p->Di[i], p->Mi[i + 1], t[5]); /* DM */
    }
    hmm_set_trans(h, p->Mi[M - 1], p->E, tr[7 * M] /* trans[M].MM */);

    /* setup_entry_trans: :410-439 */
    if (p->entry_dist == ORC_ENTRY_DIST_UNIFORM)
    {
        ofloat Mf = (ofloat)M;
        ofloat cost = (ofloat)O_LOG((ofloat)(2.0 / (Mf * (Mf + 1)))) * Mf;
        for (unsigned i = 0; i < M; ++i)
            hmm_set_trans(h, p->B, p->Mi[i], cost);
    }
    else
    {
        calculate_occupancy(p);
        for (unsigned i = 0; i < M; ++i)
            hmm_set_trans(h, p->B, p->Mi[i], p->locc[i]);
    }
    /* setup_exit_trans: :441-458 */
    for (unsigned i = 0; i < M; ++i)
        hmm_set_trans(h, p->Mi[i], p->E, 0);
    for (unsigned i = 1; i < M; ++i)
        hmm_set_trans(h, p->Di[i], p->E, 0);
    /* init_alt_xtrans: :322-340 (all LOG1 until protein_profile_setup) */
    hmm_set_trans(h, p->S, p->B, 0);
    hmm_set_trans(h, p->S, p->N, 0);
    hmm_set_trans(h, p->N, p->N, 0);
    hmm_set_trans(h, p->N, p->B, 0);
    hmm_set_trans(h, p->E, p->T, 0);
    hmm_set_trans(h, p->E, p->C, 0);
    hmm_set_trans(h, p->C, p->C, 0);
    hmm_set_trans(h, p->C, p->T, 0);
    hmm_set_trans(h, p->E, p->B, 0);
    hmm_set_trans(h, p->E, p->J, 0);
    hmm_set_trans(h, p->J, p->J, 0);
    hmm_set_trans(h, p->J, p->B, 0);

    p->t_SB = hmm_trans_idx(h, p->S, p->B);
    p->t_SN = hmm_trans_idx(h, p->S, p->N);
    p->t_NN = hmm_trans_idx(h, p->N, p->N);
    p->t_NB = hmm_trans_idx(h, p->N, p->B);
    p->t_ET = hmm_trans_idx(h, p->E, p->T);
    p->t_EC = hmm_trans_idx(h, p->E, p->C);
    p->t_CC = hmm_trans_idx(h, p->C, p->C);
    p->t_CT = hmm_trans_idx(h, p->C, p->T);
    p->t_EB = hmm_trans_idx(h, p->E, p->B);
    p->t_EJ = hmm_trans_idx(h, p->E, p->J);
    p->t_JJ = hmm_trans_idx(h, p->J, p->J);
    p->t_JB = hmm_trans_idx(h, p->J, p->B);
    hmm_compile(h);
}

struct orc_profile *orc_profile_new(unsigned core_size, int entry_dist,
                                    ofloat epsilon,
                                    ofloat const null_lprobs[20],
                                    ofloat const *match_lprobs,
                                    ofloat const *trans)
{
    if (core_size == 0 || core_size > ORC_CORE_SIZE_MAX) return NULL;
    struct orc_profile *p = calloc(1, sizeof *p);
    unsigned M = core_size;
    p->M = M;
    p->entry_dist = entry_dist;
    p->eps = epsilon;
    memcpy(p->null_lprobs, null_lprobs, sizeof p->null_lprobs);
    p->match_d = malloc(sizeof *p->match_d * M);
    p->trans = malloc(sizeof(ofloat) * 7 * (M + 1));
    p->locc = malloc(sizeof(ofloat) * M);
    memcpy(p->trans, trans, sizeof(ofloat) * 7 * (M + 1));

    /* protein_model_init: :105-137 */
    orc_setup_nuclt_dist(&p->null_d, null_lprobs);
    ofloat zeros[20] = {0};
    orc_setup_nuclt_dist(&p->insert_d, zeros);
    /* protein_model_add_node: :60-66 (lodds = lprob - null) */
    for (unsigned k = 0; k < M; ++k)
    {
        ofloat lodds[20];
        for (unsigned i = 0; i < 20; ++i)
            lodds[i] = match_lprobs[k * 20 + i] - null_lprobs[i];
        orc_setup_nuclt_dist(&p->match_d[k], lodds);
    }
    build_models(p);
    return p;
}

struct orc_profile *orc_profile_sample(unsigned seed, unsigned core_size,
                                       int entry_dist, ofloat epsilon)
{
    if (core_size < 2) return NULL;
    struct orc_rnd rnd;
    orc_rnd_seed(&rnd, seed);
    unsigned M = core_size;
    ofloat null_lp[20];
    orc_lprob_sample(&rnd, 20, null_lp);
    orc_lprob_normalize(20, null_lp);
    ofloat *match = malloc(sizeof(ofloat) * 20 * M);
    for (unsigned k = 0; k < M; ++k)
    {
        orc_lprob_sample(&rnd, 20, match + 20 * k);
        orc_lprob_normalize(20, match + 20 * k);
    }
    ofloat *trans = malloc(sizeof(ofloat) * 7 * (M + 1));
    for (unsigned i = 0; i < M + 1; ++i)
    {
        ofloat *t = trans + 7 * i;
        orc_lprob_sample(&rnd, 7, t);
        if (i == 0) t[6] = NEG_INF; /* DD */
        if (i == M)
        {
            t[2] = NEG_INF; /* MD */
            t[6] = NEG_INF; /* DD */
        }
        orc_lprob_normalize(7, t);
    }
    struct orc_profile *p =
        orc_profile_new(M, entry_dist, epsilon, null_lp, match, trans);
    free(match);
    free(trans);
    return p;
}

void orc_profile_del(struct orc_profile *p)
{
    if (!p) return;
    hmm_free(&p->null);
    hmm_free(&p->alt);
    free(p->match_d);
    free(p->trans);
    free(p->locc);
    free(p->tbl_null);
    free(p->tbl_insert);
    free(p->tbl_match);
    free(p->Mi);
    free(p->Ii);
    free(p->Di);
    free(p);
}

unsigned orc_profile_core_size(struct orc_profile const *p) { return p->M; }
unsigned orc_profile_nstates(struct orc_profile const *p, int alt)
{
    return (unsigned)(alt ? p->alt.nstates : p->null.nstates);
}

/* protein_profile_setup: protein_profile.c:155-216.  The 13 values it writes, in the order
 * RR, SB, SN, NN, NB, ET, EC, CC, CT, EB, EJ, JJ, JB. */
int orc_xtrans(unsigned seq_size, int multi_hits, int hmmer3_compat, ofloat xt[13])
{
    if (seq_size == 0) return ORC_EINVAL;
    ofloat L = (ofloat)seq_size;
    ofloat q = 0;
    ofloat log_q = NEG_INF;
    if (multi_hits)
    {
        q = (ofloat)0.5;
        log_q = (ofloat)O_LOG((ofloat)0.5);
    }
    ofloat lp = (ofloat)O_LOG(L) - (ofloat)O_LOG(L + 2 + q / (1 - q));
    ofloat l1p = (ofloat)O_LOG(2 + q / (1 - q)) -
                 (ofloat)O_LOG(L + 2 + q / (1 - q));
    ofloat lr = (ofloat)O_LOG(L) - (ofloat)O_LOG(L + 1);

    ofloat NN, CC, JJ, NB, CT, JB, RR, EJ, EC;
    NN = CC = JJ = lp;
    NB = CT = JB = l1p;
    RR = lr;
    EJ = log_q;
    EC = (ofloat)O_LOG(1 - q);
    if (hmmer3_compat) NN = CC = JJ = 0;
    xt[0] = RR;
    xt[1] = NB;      /* S->B */
    xt[2] = NN;      /* S->N */
    xt[3] = NN;      /* N->N */
    xt[4] = NB;      /* N->B */
    xt[5] = EC + CT; /* E->T */
    xt[6] = EC + CC; /* E->C */
    xt[7] = CC;      /* C->C */
    xt[8] = CT;      /* C->T */
    xt[9] = EJ + JB; /* E->B */
    xt[10] = EJ + JJ; /* E->J */
    xt[11] = JJ;     /* J->J */
    xt[12] = JB;     /* J->B */
    return ORC_OK;
}

int orc_profile_setup(struct orc_profile *p, unsigned seq_size, int multi_hits,
                      int hmmer3_compat)
{
    ofloat xt[13];
    if (orc_xtrans(seq_size, multi_hits, hmmer3_compat, xt)) return ORC_EINVAL;
    p->null.trans[p->t_RR].lp = xt[0];
    struct otrans *t = p->alt.trans;
    t[p->t_SB].lp = xt[1];
    t[p->t_SN].lp = xt[2];
    t[p->t_NN].lp = xt[3];
    t[p->t_NB].lp = xt[4];
    t[p->t_ET].lp = xt[5];
    t[p->t_EC].lp = xt[6];
    t[p->t_CC].lp = xt[7];
    t[p->t_CT].lp = xt[8];
    t[p->t_EB].lp = xt[9];
    t[p->t_EJ].lp = xt[10];
    t[p->t_JJ].lp = xt[11];
    t[p->t_JB].lp = xt[12];
    return ORC_OK;
}

static ofloat alt_lp(struct orc_profile const *p, int src, int dst)
{
    int i = hmm_trans_idx(&p->alt, src, dst);
    return i < 0 ? NEG_INF : p->alt.trans[i].lp;
}

void orc_profile_export(struct orc_profile const *p, ofloat *trans8,
                        ofloat *emis_match, ofloat *emis_insert,
                        ofloat *emis_null, ofloat *xtrans)
{
    unsigned M = p->M;
    if (trans8)
    {
        for (unsigned k = 0; k < M; ++k)
        {
            trans8[0 * M + k] = alt_lp(p, p->B, p->Mi[k]);
            trans8[1 * M + k] = k ? alt_lp(p, p->Mi[k - 1], p->Mi[k]) : NEG_INF;
            trans8[2 * M + k] = k ? alt_lp(p, p->Ii[k - 1], p->Mi[k]) : NEG_INF;
            trans8[3 * M + k] = k ? alt_lp(p, p->Di[k - 1], p->Mi[k]) : NEG_INF;
            trans8[4 * M + k] = k ? alt_lp(p, p->Mi[k - 1], p->Di[k]) : NEG_INF;
            trans8[5 * M + k] = k ? alt_lp(p, p->Di[k - 1], p->Di[k]) : NEG_INF;
            trans8[6 * M + k] = alt_lp(p, p->Mi[k], p->Ii[k]);
            trans8[7 * M + k] = alt_lp(p, p->Ii[k], p->Ii[k]);
        }
    }
    if (emis_match)
        for (unsigned c = 0; c < ORC_NCODES; ++c)
            for (unsigned k = 0; k < M; ++k)
                emis_match[(size_t)c * M + k] =
                    p->tbl_match[(size_t)k * ORC_NCODES + c];
    if (emis_insert)
        memcpy(emis_insert, p->tbl_insert, sizeof(ofloat) * ORC_NCODES);
    if (emis_null) memcpy(emis_null, p->tbl_null, sizeof(ofloat) * ORC_NCODES);
    if (xtrans)
    {
        struct otrans const *t = p->alt.trans;
        xtrans[0] = p->null.trans[p->t_RR].lp;
        xtrans[1] = t[p->t_SB].lp;
        xtrans[2] = t[p->t_SN].lp;
        xtrans[3] = t[p->t_NN].lp;
        xtrans[4] = t[p->t_NB].lp;
        xtrans[5] = t[p->t_ET].lp;
        xtrans[6] = t[p->t_EC].lp;
        xtrans[7] = t[p->t_CC].lp;
        xtrans[8] = t[p->t_CT].lp;
        xtrans[9] = t[p->t_EB].lp;
        xtrans[10] = t[p->t_EJ].lp;
        xtrans[11] = t[p->t_JJ].lp;
        xtrans[12] = t[p->t_JB].lp;
    }
}

void orc_profile_dists(struct orc_profile const *p,
                       struct orc_nuclt_dist *null_d,
                       struct orc_nuclt_dist *insert_d,
                       struct orc_nuclt_dist *match_d)
{
    if (null_d) *null_d = p->null_d;
    if (insert_d) *insert_d = p->insert_d;
    if (match_d) memcpy(match_d, p->match_d, sizeof *match_d * p->M);
}

/* ======================================================================== */
/* Generic Viterbi (imm_dp_viterbi restatement, end-indexed bookkeeping)      */
/* ======================================================================== */
#define START_MARK (-2)

static int viterbi_generic(struct ohmm const *h, unsigned char const *seq,
                           unsigned L, ofloat *loglik, uint16_t *path_state,
                           uint8_t *path_len, unsigned *nsteps)
{
    int const n = h->nstates;
    int const want_path = path_state != NULL && path_len != NULL;
    size_t rows = want_path ? (size_t)L + 1 : 6;
    /* V[j][s]: best score of a path ending in s having consumed j symbols.
     * P[j][s]: max over incoming (V[j][src] + t) (and the start lprob) --
     * "best_trans_score" of imm, evaluated once per (j, s). */
    ofloat *V = malloc(sizeof(ofloat) * rows * (size_t)n);
    ofloat *P = malloc(sizeof(ofloat) * rows * (size_t)n);
    int32_t *argP = NULL, *argV = NULL;
    uint8_t *lenV = NULL;
    if (!V || !P)
    {
        free(V);
        free(P);
        return ORC_ENOMEM;
    }
    if (want_path)
    {
        argP = malloc(sizeof(int32_t) * rows * (size_t)n);
        argV = malloc(sizeof(int32_t) * rows * (size_t)n);
        lenV = malloc(rows * (size_t)n);
        if (!argP || !argV || !lenV)
        {
            free(V), free(P), free(argP), free(argV), free(lenV);
            return ORC_ENOMEM;
        }
    }
#define ROW(j) ((want_path ? (size_t)(j) : (size_t)(j) % 6) * (size_t)n)

    unsigned w = 0; /* rolling base-4 word of the last 5 symbols */
    for (unsigned j = 0; j <= L; ++j)
    {
        if (j > 0) w = ((w << 2) | seq[j - 1]) & 1023u;
        ofloat *Vj = V + ROW(j);
        /* emitting states */
        for (int oi = 0; oi < h->n_emit; ++oi)
        {
            int s = h->order[oi];
            ofloat best = NEG_INF;
            int bl = 0;
            unsigned maxl = j < 5 ? j : 5;
            for (unsigned l = 1; l <= maxl; ++l)
            {
                unsigned code = code_off[l] + (w & ((1u << (2 * l)) - 1));
                ofloat sc = P[ROW(j - l) + s] + h->states[s].tbl[code];
                if (sc > best)
                {
                    best = sc;
                    bl = (int)l;
                }
            }
            Vj[s] = best;
            if (want_path)
            {
                lenV[ROW(j) + s] = (uint8_t)bl;
                argV[ROW(j) + s] = -1;
            }
        }
        /* mute states, topological order */
        for (int oi = h->n_emit; oi < n; ++oi)
        {
            int s = h->order[oi];
            ofloat best = NEG_INF;
            int arg = -1;
            if (s == h->start && j == 0)
            {
                best = h->start_lp;
                arg = START_MARK;
            }
            for (int e = h->in_begin[s]; e < h->in_begin[s + 1]; ++e)
            {
                struct otrans const *t = &h->trans[h->in_idx[e]];
                ofloat v = Vj[t->src] + t->lp;
                if (v > best)
                {
                    best = v;
                    arg = t->src;
                }
            }
            Vj[s] = best;
            if (want_path)
            {
                argV[ROW(j) + s] = arg;
                lenV[ROW(j) + s] = 0;
            }
        }
        /* predecessor maxima for emitting states leaving row j */
        ofloat *Pj = P + ROW(j);
        for (int oi = 0; oi < h->n_emit; ++oi)
        {
            int s = h->order[oi];
            ofloat best = NEG_INF;
            int arg = -1;
            if (s == h->start && j == 0)
            {
                best = h->start_lp;
                arg = START_MARK;
            }
            for (int e = h->in_begin[s]; e < h->in_begin[s + 1]; ++e)
            {
                struct otrans const *t = &h->trans[h->in_idx[e]];
                ofloat v = Vj[t->src] + t->lp;
                if (v > best)
                {
                    best = v;
                    arg = t->src;
                }
            }
            Pj[s] = best;
            if (want_path) argP[ROW(j) + s] = arg;
        }
    }
    *loglik = V[ROW(L) + h->end];

    int rc = ORC_OK;
    if (want_path)
    {
        unsigned cap = *nsteps, cnt = 0;
        /* count + write reversed, then flip */
        int s = h->end;
        unsigned j = L;
        if (!(*loglik > NEG_INF))
        {
            *nsteps = 0;
        }
        else
        {
            while (s != START_MARK && s >= 0)
            {
                unsigned l = lenV[ROW(j) + s];
                if (cnt < cap)
                {
                    path_state[cnt] = (uint16_t)h->states[s].id;
                    path_len[cnt] = (uint8_t)l;
                }
                cnt++;
                if (h->states[s].emitting)
                {
                    j -= l;
                    s = argP[ROW(j) + s];
                }
                else
                    s = argV[ROW(j) + s];
            }
            if (cnt > cap) rc = ORC_ENOMEM;
            unsigned m = cnt < cap ? cnt : cap;
            for (unsigned i = 0; i < m / 2; ++i)
            {
                uint16_t ts = path_state[i];
                path_state[i] = path_state[m - 1 - i];
                path_state[m - 1 - i] = ts;
                uint8_t tl = path_len[i];
                path_len[i] = path_len[m - 1 - i];
                path_len[m - 1 - i] = tl;
            }
            *nsteps = cnt;
        }
    }
#undef ROW
    free(V), free(P), free(argP), free(argV), free(lenV);
    return rc;
}

int orc_viterbi(struct orc_profile const *p, int alt, unsigned char const *seq,
                unsigned L, ofloat *loglik, uint16_t *path_state,
                uint8_t *path_len, unsigned *nsteps)
{
    for (unsigned i = 0; i < L; ++i)
        if (seq[i] > 3) return ORC_EINVAL;
    return viterbi_generic(alt ? &p->alt : &p->null, seq, L, loglik, path_state,
                           path_len, nsteps);
}

/* Score of a GIVEN path in the model: start lprob + transitions + emissions, accumulated in
 * the DP's own order ((prev + trans) + emis).  NaN if the path is not a path of the graph
 * (unknown state, missing transition, wrong start/end, fragments not covering the sequence). */
ofloat orc_path_score(struct orc_profile const *p, int alt, unsigned char const *seq, unsigned L,
                      uint16_t const *path_state, uint8_t const *path_len, unsigned nsteps)
{
    struct ohmm const *h = alt ? &p->alt : &p->null;
    ofloat score = 0;
    int prev = -1;
    unsigned pos = 0;
    for (unsigned i = 0; i < nsteps; ++i)
    {
        int cur = -1;
        for (int s = 0; s < h->nstates; ++s)
            if (h->states[s].id == path_state[i])
            {
                cur = s;
                break;
            }
        if (cur < 0) return (ofloat)NAN;
        unsigned len = path_len[i];
        if (h->states[cur].emitting ? (len < 1 || len > 5) : len != 0) return (ofloat)NAN;
        if (pos + len > L) return (ofloat)NAN;
        if (i == 0)
        {
            if (cur != h->start) return (ofloat)NAN;
            score = h->start_lp;
        }
        else
        {
            int t = hmm_trans_idx(h, prev, cur);
            if (t < 0) return (ofloat)NAN;
            score = score + h->trans[t].lp;
        }
        if (len) score = score + h->states[cur].tbl[orc_word_code(seq + pos, len)];
        pos += len;
        prev = cur;
    }
    if (prev != h->end || pos != L) return (ofloat)NAN;
    return score;
}

/* ======================================================================== */
/* End-indexed score-only recursion                       */
/* ======================================================================== */
static inline ofloat omax(ofloat a, ofloat b) { return a > b ? a : b; }

int orc_dp_tables(unsigned M, unsigned ldk, ofloat const *trans8,
                  ofloat const *emis_match, ofloat const *emis_insert,
                  ofloat const *emis_null, ofloat const *xt, unsigned char const *seq,
                  unsigned L, ofloat *null_loglik, ofloat *alt_loglik)
{
    if (L == 0) return ORC_EINVAL;
    for (unsigned i = 0; i < L; ++i)
        if (seq[i] > 3) return ORC_EINVAL;
    ofloat const *ENT = trans8 + 0 * (size_t)ldk, *MM = trans8 + 1 * (size_t)ldk,
                 *IM = trans8 + 2 * (size_t)ldk, *DM = trans8 + 3 * (size_t)ldk,
                 *MD = trans8 + 4 * (size_t)ldk, *DD = trans8 + 5 * (size_t)ldk,
                 *MI = trans8 + 6 * (size_t)ldk, *II = trans8 + 7 * (size_t)ldk;
    ofloat const RR = xt[0], SB = xt[1], SN = xt[2], NN = xt[3], NB = xt[4],
                 ET = xt[5], EC = xt[6], CC = xt[7], CT = xt[8], EB = xt[9],
                 EJ = xt[10], JJ = xt[11], JB = xt[12];

    ofloat *PM = malloc(sizeof(ofloat) * 6 * (size_t)M);
    ofloat *QI = malloc(sizeof(ofloat) * 6 * (size_t)M);
    ofloat *Mr = malloc(sizeof(ofloat) * (size_t)M);
    ofloat *Ir = malloc(sizeof(ofloat) * (size_t)M);
    ofloat *Dr = malloc(sizeof(ofloat) * (size_t)M);
    ofloat PN[6], PJ[6], PC[6], PR[6];

    /* row 0 */
    {
        ofloat S = 0, B = S + SB; /* N,J,E = -inf */
        for (unsigned k = 0; k < M; ++k)
        {
            PM[k] = B + ENT[k];
            QI[k] = NEG_INF;
        }
        PN[0] = S + SN;
        PJ[0] = NEG_INF;
        PC[0] = NEG_INF;
        PR[0] = 0; /* start lprob of R */
    }
    ofloat E = NEG_INF, Cc = NEG_INF, Rr = NEG_INF;
    unsigned w = 0;
    for (unsigned j = 1; j <= L; ++j)
    {
        w = ((w << 2) | seq[j - 1]) & 1023u;
        unsigned maxl = j < 5 ? j : 5;
        ofloat N = NEG_INF, J = NEG_INF;
        Cc = NEG_INF;
        Rr = NEG_INF;
        for (unsigned k = 0; k < M; ++k)
            Mr[k] = Ir[k] = NEG_INF;
        for (unsigned l = 1; l <= maxl; ++l)
        {
            unsigned code = code_off[l] + (w & ((1u << (2 * l)) - 1));
            unsigned r = (j - l) % 6;
            ofloat eN = emis_null[code], eI = emis_insert[code];
            ofloat const *eM = emis_match + (size_t)code * ldk;
            N = omax(N, PN[r] + eN);
            J = omax(J, PJ[r] + eN);
            Cc = omax(Cc, PC[r] + eN);
            Rr = omax(Rr, PR[r] + eN);
            ofloat const *pm = PM + (size_t)r * M, *qi = QI + (size_t)r * M;
            for (unsigned k = 0; k < M; ++k)
            {
                Mr[k] = omax(Mr[k], pm[k] + eM[k]);
                Ir[k] = omax(Ir[k], qi[k] + eI);
            }
        }
        Dr[0] = NEG_INF;
        for (unsigned k = 1; k < M; ++k)
            Dr[k] = omax(Mr[k - 1] + MD[k], Dr[k - 1] + DD[k]);
        E = NEG_INF;
        for (unsigned k = 0; k < M; ++k)
        {
            E = omax(E, Mr[k]);
            if (k) E = omax(E, Dr[k]);
        }
        ofloat B = omax(omax(N + NB, E + EB), J + JB); /* S(j>0) = -inf */
        unsigned r = j % 6;
        ofloat *pm = PM + (size_t)r * M, *qi = QI + (size_t)r * M;
        for (unsigned k = 0; k < M; ++k)
        {
            ofloat v = B + ENT[k];
            if (k)
            {
                v = omax(v, Mr[k - 1] + MM[k]);
                v = omax(v, Ir[k - 1] + IM[k]);
                v = omax(v, Dr[k - 1] + DM[k]);
            }
            pm[k] = v;
            qi[k] = omax(Mr[k] + MI[k], Ir[k] + II[k]);
        }
        PN[r] = N + NN;
        PJ[r] = omax(E + EJ, J + JJ);
        PC[r] = omax(E + EC, Cc + CC);
        PR[r] = Rr + RR;
    }
    *null_loglik = Rr;
    *alt_loglik = omax(E + ET, Cc + CT);
    free(PM), free(QI), free(Mr), free(Ir), free(Dr);
    return ORC_OK;
}

int orc_viterbi_fast(struct orc_profile const *p, unsigned char const *seq,
                     unsigned L, ofloat *null_loglik, ofloat *alt_loglik)
{
    unsigned M = p->M;
    ofloat *t8 = malloc(sizeof(ofloat) * 8 * (size_t)M);
    ofloat *em = malloc(sizeof(ofloat) * ORC_NCODES * (size_t)M);
    ofloat xt[13];
    orc_profile_export(p, t8, em, NULL, NULL, xt);
    int rc = orc_dp_tables(M, M, t8, em, p->tbl_insert, p->tbl_null, xt, seq, L,
                           null_loglik, alt_loglik);
    free(t8), free(em);
    return rc;
}

ofloat orc_lrt(ofloat null_loglik, ofloat alt_loglik)
{
    return -2 * (null_loglik - alt_loglik);
}

/* ======================================================================== */
/* Codon decode (imm_frame_cond_decode restatement)                           */
/* ======================================================================== */
ofloat orc_profile_decode(struct orc_profile const *p, unsigned char const *frag,
                          unsigned len, unsigned state_id,
                          unsigned char codon[3])
{
    struct orc_nuclt_dist const *d;
    unsigned msb = state_id & (3u << 14);
    if (msb == ORC_INSERT_STATE)
        d = &p->insert_d;
    else if (msb == ORC_MATCH_STATE)
        d = &p->match_d[(state_id & 0x3FFF) - 1];
    else
        d = &p->null_d;
    ofloat le = (ofloat)O_LOG(p->eps), l1 = (ofloat)O_LOG((ofloat)1 - p->eps);
    ofloat best = NEG_INF;
    codon[0] = codon[1] = codon[2] = 4;
    if (len < 1 || len > 5) return (ofloat)NAN;
    for (unsigned a = 0; a < 4; ++a)
        for (unsigned b = 0; b < 4; ++b)
            for (unsigned c = 0; c < 4; ++c)
            {
                /* joint p(frag, codon): the marginal formula with the codon
                 * marginals restricted to this one codon */
                ofloat lp = d->codonm[a * 25 + b * 5 + c];
                ofloat cm[125];
                for (unsigned i = 0; i < 5; ++i)
                    for (unsigned j = 0; j < 5; ++j)
                        for (unsigned k = 0; k < 5; ++k)
                        {
                            int ok = (i == 4 || i == a) && (j == 4 || j == b) &&
                                     (k == 4 || k == c);
                            cm[i * 25 + j * 5 + k] = ok ? lp : NEG_INF;
                        }
                ofloat v = frame_lprob(d->nucltp, cm, le, l1, frag, len);
                if (v >= best && !(v != v))
                {
                    best = v;
                    codon[0] = (unsigned char)a;
                    codon[1] = (unsigned char)b;
                    codon[2] = (unsigned char)c;
                }
            }
    return best;
}

/* joint log p(fragment, codon) under the distribution protein_profile_decode picks for state_id */
ofloat orc_profile_codon_lprob(struct orc_profile const *p, unsigned char const *frag,
                               unsigned len, unsigned state_id, unsigned char const codon[3])
{
    struct orc_nuclt_dist const *d;
    unsigned msb = state_id & (3u << 14);
    if (msb == ORC_INSERT_STATE)
        d = &p->insert_d;
    else if (msb == ORC_MATCH_STATE)
        d = &p->match_d[(state_id & 0x3FFF) - 1];
    else
        d = &p->null_d;
    ofloat le = (ofloat)O_LOG(p->eps), l1 = (ofloat)O_LOG((ofloat)1 - p->eps);
    if (len < 1 || len > 5) return (ofloat)NAN;
    unsigned a = codon[0], b = codon[1], c = codon[2];
    ofloat lp = d->codonm[a * 25 + b * 5 + c];
    ofloat cm[125];
    for (unsigned i = 0; i < 5; ++i)
        for (unsigned j = 0; j < 5; ++j)
            for (unsigned k = 0; k < 5; ++k)
            {
                int ok = (i == 4 || i == a) && (j == 4 || j == b) && (k == 4 || k == c);
                cm[i * 25 + j * 5 + k] = ok ? lp : NEG_INF;
            }
    return frame_lprob(d->nucltp, cm, le, l1, frag, len);
}

/* protein_state_name: protein_state.c:5-39 */
unsigned orc_state_name(unsigned id, char name[8])
{
    unsigned msb = id & (3u << 14);
    if (msb == ORC_EXT_STATE)
    {
        static char const ext[] = "RSNBEJCT";
        unsigned i = id & 0x3FFF;
        name[0] = i < 8 ? ext[i] : '?';
        name[1] = '\0';
        return 1;
    }
    name[0] = msb == ORC_MATCH_STATE ? 'M' : msb == ORC_INSERT_STATE ? 'I' : 'D';
    return (unsigned)snprintf(name + 1, 7, "%u", id & 0x3FFF) + 1;
}

/* ======================================================================== */
/* thread_run restatement = CPU baseline                                     */
/* ======================================================================== */
static unsigned ceildiv(unsigned x, unsigned y) { return (x + y - 1) / y; }

long orc_scan(struct orc_profile *const *profiles, unsigned nprofiles,
              unsigned char const *seqs, uint32_t const *seq_off,
              unsigned nseqs, int multi_hits, int hmmer3_compat, double lrt_thr,
              int nthreads, int mode, ofloat *out_null, ofloat *out_alt)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64; /* limits.h NUM_THREADS */
    unsigned nparts = (unsigned)nthreads < nprofiles ? (unsigned)nthreads : nprofiles;
    unsigned psize = ceildiv(nprofiles, nparts); /* xmath_partition_size */
    long hits = 0;
    /* scan.c:227-258: serial over sequences, parallel over partitions */
    for (unsigned q = 0; q < nseqs; ++q)
    {
        unsigned char const *seq = seqs + seq_off[q];
        unsigned L = seq_off[q + 1] - seq_off[q];
#pragma omp parallel for schedule(static, 1) num_threads(nthreads) reduction(+ : hits)
        for (unsigned part = 0; part < nparts; ++part)
        {
            unsigned lo = part * psize;
            unsigned hi = lo + psize < nprofiles ? lo + psize : nprofiles;
            for (unsigned pi = lo; pi < hi; ++pi)
            {
                struct orc_profile *p = profiles[pi];
                ofloat nl = NEG_INF, al = NEG_INF;
                if (orc_profile_setup(p, L, multi_hits, hmmer3_compat)) continue;
                if (mode == 0)
                {
                    unsigned ns = 0;
                    viterbi_generic(&p->null, seq, L, &nl, NULL, NULL, &ns);
                    viterbi_generic(&p->alt, seq, L, &al, NULL, NULL, &ns);
                }
                else
                    orc_viterbi_fast(p, seq, L, &nl, &al);
                if (out_null) out_null[(size_t)q * nprofiles + pi] = nl;
                if (out_alt) out_alt[(size_t)q * nprofiles + pi] = al;
                ofloat lrt = orc_lrt(nl, al);
                if (isfinite((double)lrt) && !(lrt < (ofloat)lrt_thr)) hits++;
            }
        }
    }
    return hits;
}

/* ======================================================================== */
/* Optimised CPU variant (SURVEY.md 8d "second CPU figure"): DB resident in   */
/* RAM, tables exported ONCE per profile, null score once per (sequence,      */
/* distinct null table), no allocation inside the pair loop, no barrier per   */
/* sequence: each partition's thread runs its profiles over all sequences.    */
/* Same arithmetic as orc_dp_tables (bit-identical scores).                   */
/* ======================================================================== */
struct resident_prof
{
    unsigned M;
    ofloat *t8; /* [8][M] */
    ofloat *em; /* [1364][M] */
    ofloat const *ei, *en;
};

static ofloat null_score_tables(ofloat const *emis_null, ofloat RR, unsigned char const *seq, unsigned L)
{
    ofloat PR[6], Rr = NEG_INF;
    unsigned w = 0;
    PR[0] = 0;
    for (unsigned j = 1; j <= L; ++j)
    {
        w = ((w << 2) | seq[j - 1]) & 1023u;
        unsigned maxl = j < 5 ? j : 5;
        Rr = NEG_INF;
        for (unsigned l = 1; l <= maxl; ++l)
            Rr = omax(Rr, PR[(j - l) % 6] + emis_null[code_off[l] + (w & ((1u << (2 * l)) - 1))]);
        PR[j % 6] = Rr + RR;
    }
    return Rr;
}

/* alt model only, caller-provided work area of 15*M floats */
static ofloat alt_score_tables(struct resident_prof const *rp, ofloat const *xt, unsigned char const *seq,
                               unsigned L, ofloat *work)
{
    unsigned const M = rp->M;
    ofloat const *ENT = rp->t8, *MM = rp->t8 + M, *IM = rp->t8 + 2 * (size_t)M, *DM = rp->t8 + 3 * (size_t)M,
                 *MD = rp->t8 + 4 * (size_t)M, *DD = rp->t8 + 5 * (size_t)M, *MI = rp->t8 + 6 * (size_t)M,
                 *II = rp->t8 + 7 * (size_t)M;
    ofloat const SB = xt[1], SN = xt[2], NN = xt[3], NB = xt[4], ET = xt[5], EC = xt[6], CC = xt[7],
                 CT = xt[8], EB = xt[9], EJ = xt[10], JJ = xt[11], JB = xt[12];
    ofloat *PM = work, *QI = work + 6 * (size_t)M, *Mr = work + 12 * (size_t)M, *Ir = work + 13 * (size_t)M,
           *Dr = work + 14 * (size_t)M;
    ofloat PN[6], PJ[6], PC[6];
    {
        ofloat B = 0 + SB;
        for (unsigned k = 0; k < M; ++k)
        {
            PM[k] = B + ENT[k];
            QI[k] = NEG_INF;
        }
        PN[0] = 0 + SN;
        PJ[0] = PC[0] = NEG_INF;
    }
    ofloat E = NEG_INF, Cc = NEG_INF;
    unsigned w = 0;
    for (unsigned j = 1; j <= L; ++j)
    {
        w = ((w << 2) | seq[j - 1]) & 1023u;
        unsigned maxl = j < 5 ? j : 5;
        ofloat N = NEG_INF, J = NEG_INF;
        Cc = NEG_INF;
        for (unsigned k = 0; k < M; ++k)
            Mr[k] = Ir[k] = NEG_INF;
        for (unsigned l = 1; l <= maxl; ++l)
        {
            unsigned code = code_off[l] + (w & ((1u << (2 * l)) - 1));
            unsigned r = (j - l) % 6;
            ofloat eN = rp->en[code], eI = rp->ei[code];
            ofloat const *restrict eM = rp->em + (size_t)code * M;
            N = omax(N, PN[r] + eN);
            J = omax(J, PJ[r] + eN);
            Cc = omax(Cc, PC[r] + eN);
            ofloat const *restrict pm = PM + (size_t)r * M, *restrict qi = QI + (size_t)r * M;
            for (unsigned k = 0; k < M; ++k)
            {
                Mr[k] = omax(Mr[k], pm[k] + eM[k]);
                Ir[k] = omax(Ir[k], qi[k] + eI);
            }
        }
        Dr[0] = NEG_INF;
        for (unsigned k = 1; k < M; ++k)
            Dr[k] = omax(Mr[k - 1] + MD[k], Dr[k - 1] + DD[k]);
        E = Mr[0];
        for (unsigned k = 1; k < M; ++k)
            E = omax(E, omax(Mr[k], Dr[k]));
        ofloat B = omax(omax(N + NB, E + EB), J + JB);
        unsigned r = j % 6;
        ofloat *restrict pm = PM + (size_t)r * M, *restrict qi = QI + (size_t)r * M;
        pm[0] = B + ENT[0];
        for (unsigned k = 1; k < M; ++k)
            pm[k] = omax(omax(B + ENT[k], Mr[k - 1] + MM[k]), omax(Ir[k - 1] + IM[k], Dr[k - 1] + DM[k]));
        for (unsigned k = 0; k < M; ++k)
            qi[k] = omax(Mr[k] + MI[k], Ir[k] + II[k]);
        PN[r] = N + NN;
        PJ[r] = omax(E + EJ, J + JJ);
        PC[r] = omax(E + EC, Cc + CC);
    }
    return omax(E + ET, Cc + CT);
}

long orc_scan_resident(struct orc_profile *const *profiles, unsigned nprofiles,
                       unsigned char const *seqs, uint32_t const *seq_off, unsigned nseqs,
                       int multi_hits, int hmmer3_compat, double lrt_thr, int nthreads,
                       ofloat *out_null, ofloat *out_alt, double *prepare_seconds, double *dp_seconds)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    for (unsigned q = 0; q < nseqs; ++q)
        if (seq_off[q + 1] <= seq_off[q]) return -1;
    double t0 = omp_get_wtime();
    struct resident_prof *rp = calloc(nprofiles, sizeof *rp);
    ofloat *xts = malloc(sizeof(ofloat) * 13 * (size_t)nseqs);
    if (!rp || !xts)
    {
        free(rp), free(xts);
        return -1;
    }
    unsigned maxM = 0;
    /* once per DB: the tables every pair of this profile uses */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(max : maxM)
    for (unsigned pi = 0; pi < nprofiles; ++pi)
    {
        struct orc_profile const *p = profiles[pi];
        unsigned M = p->M;
        rp[pi].M = M;
        rp[pi].t8 = malloc(sizeof(ofloat) * 8 * (size_t)M);
        rp[pi].em = malloc(sizeof(ofloat) * ORC_NCODES * (size_t)M);
        rp[pi].ei = p->tbl_insert;
        rp[pi].en = p->tbl_null;
        orc_profile_export(p, rp[pi].t8, rp[pi].em, NULL, NULL, NULL);
        if (M > maxM) maxM = M;
    }
    /* once per sequence: the 13 length-dependent transitions (protein_profile_setup) */
    for (unsigned q = 0; q < nseqs; ++q)
        orc_xtrans(seq_off[q + 1] - seq_off[q], multi_hits, hmmer3_compat, xts + 13 * (size_t)q);
    double t1 = omp_get_wtime();
    unsigned nparts = (unsigned)nthreads < nprofiles ? (unsigned)nthreads : nprofiles;
    unsigned psize = ceildiv(nprofiles, nparts);
    long hits = 0;
#pragma omp parallel num_threads(nthreads) reduction(+ : hits)
    {
        ofloat *work = malloc(sizeof(ofloat) * 15 * (size_t)maxM);
        ofloat *nullq = malloc(sizeof(ofloat) * (size_t)nseqs);
#pragma omp for schedule(static, 1)
        for (unsigned part = 0; part < nparts; ++part)
        {
            unsigned lo = part * psize;
            unsigned hi = lo + psize < nprofiles ? lo + psize : nprofiles;
            ofloat const *null_of = NULL; /* table the cached null scores were computed with */
            for (unsigned pi = lo; pi < hi; ++pi)
            {
                /* null score: once per sequence for every run of profiles sharing one null table */
                if (!null_of || memcmp(null_of, rp[pi].en, sizeof(ofloat) * ORC_NCODES))
                {
                    for (unsigned q = 0; q < nseqs; ++q)
                        nullq[q] = null_score_tables(rp[pi].en, xts[13 * (size_t)q + 0], seqs + seq_off[q],
                                                     seq_off[q + 1] - seq_off[q]);
                    null_of = rp[pi].en;
                }
                for (unsigned q = 0; q < nseqs; ++q)
                {
                    ofloat nl = nullq[q];
                    ofloat al = alt_score_tables(&rp[pi], xts + 13 * (size_t)q, seqs + seq_off[q],
                                                 seq_off[q + 1] - seq_off[q], work);
                    if (out_null) out_null[(size_t)q * nprofiles + pi] = nl;
                    if (out_alt) out_alt[(size_t)q * nprofiles + pi] = al;
                    ofloat lrt = orc_lrt(nl, al);
                    if (isfinite((double)lrt) && !(lrt < (ofloat)lrt_thr)) hits++;
                }
            }
        }
        free(work), free(nullq);
    }
    double t2 = omp_get_wtime();
    if (prepare_seconds) *prepare_seconds = t1 - t0;
    if (dp_seconds) *dp_seconds = t2 - t1;
    for (unsigned pi = 0; pi < nprofiles; ++pi)
        free(rp[pi].t8), free(rp[pi].em);
    free(rp), free(xts);
    return hits;
}
