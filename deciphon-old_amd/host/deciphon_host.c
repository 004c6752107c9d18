/*
 * deciphon_host.c -- C11 host layer, part 1: the imm subset and the model objects
 * (include/deciphon_host.h sections "core", "imm subset", "model").
 *
 * Keeps the reference's names, argument meaning and return codes; every score and path is computed
 * on the device through dcp_gpu.h.  Reference files followed (behaviour, not code):
 * src/model/protein_profile.c, src/model/profile.c, src/model/protein_model.c:49-184,
 * src/model/protein_state.c, src/model/protein_codec.c, src/model/protein_h3reader.c:18-72,
 * src/model/nuclt_dist.c, src/model/standard_profile.c.
 */
#include "deciphon_host.h"
#include "host_internal.h"

#include <pthread.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

/* ---- logging: errors are reported where they are detected and returned (logging.h:32-72) ------- */
enum rc dcp_host_fail(enum rc rc, char const *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    fputs("deciphon_host: ", stderr);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    return rc;
}
#define fail dcp_host_fail

char const *rc_string(enum rc rc)
{
    static char const *const names[] = {"RC_OK",     "RC_END",    "RC_EFAIL", "RC_EINVAL", "RC_EIO",
                                        "RC_ENOMEM", "RC_EPARSE", "RC_EAPI",  "RC_EHTTP"};
    return (unsigned)rc < sizeof names / sizeof names[0] ? names[rc] : "invalid return code";
}

/* ---- alphabets / sequences --------------------------------------------------------------------- */
#define ABC(sz, syms, tid) {sz, syms, sz, 'X', {tid, NULL}}
struct imm_dna const imm_dna_iupac = {{ABC(4, "ACGT", IMM_DNA)}};
struct imm_rna const imm_rna_iupac = {{ABC(4, "ACGU", IMM_RNA)}};
struct imm_amino const imm_amino_iupac = {ABC(20, "ACDEFGHIKLMNPQRSTVWY", IMM_AMINO)};

void imm_nuclt_code_init(struct imm_nuclt_code *code, struct imm_nuclt const *nuclt)
{
    code->super.abc = &nuclt->super;
    code->nuclt = nuclt;
}

int imm_abc_symbol_idx(struct imm_abc const *abc, char symbol)
{
    if (symbol == '\0') return -1;
    if (symbol == abc->any_symbol) return (int)abc->any_symbol_id;
    char const *p = memchr(abc->symbols, symbol, abc->size);
    return p ? (int)(p - abc->symbols) : -1;
}

char const *imm_abc_typeid_name(enum imm_abc_typeid typeid)
{
    static char const *const names[] = {"null_abc", "abc", "amino", "nuclt", "dna", "rna"};
    return (unsigned)typeid < 6 ? names[typeid] : "unknown";
}

struct imm_str imm_str(char const *cstr) { return (struct imm_str){(unsigned)strlen(cstr), cstr}; }

struct imm_seq imm_seq(struct imm_str str, struct imm_abc const *abc)
{
    return (struct imm_seq){str.size, str.data, abc};
}

struct imm_seq imm_subseq(struct imm_seq const *seq, unsigned start, unsigned size)
{
    assert(start + size <= seq->size);
    return (struct imm_seq){size, seq->str + start, seq->abc};
}

struct imm_codon imm_codon(struct imm_nuclt const *nuclt, unsigned a, unsigned b, unsigned c)
{
    return (struct imm_codon){nuclt, a, b, c};
}

struct imm_codon imm_codon_any(struct imm_nuclt const *nuclt)
{
    unsigned const any = imm_abc_any_symbol_id(&nuclt->super);
    return imm_codon(nuclt, any, any, any);
}

static unsigned sym_or_any(struct imm_abc const *abc, char c)
{
    int const i = imm_abc_symbol_idx(abc, c);
    return i < 0 ? abc->any_symbol_id : (unsigned)i;
}

struct imm_codon imm_codon_from_symbols(struct imm_nuclt const *nuclt, char const sym[3])
{
    struct imm_abc const *abc = &nuclt->super;
    return imm_codon(nuclt, sym_or_any(abc, sym[0]), sym_or_any(abc, sym[1]), sym_or_any(abc, sym[2]));
}

static char codon_sym(struct imm_codon const *codon, unsigned id)
{
    struct imm_abc const *abc = &codon->nuclt->super;
    return id < abc->size ? abc->symbols[id] : abc->any_symbol;
}
char imm_codon_asym(struct imm_codon const *codon) { return codon_sym(codon, codon->a); }
char imm_codon_bsym(struct imm_codon const *codon) { return codon_sym(codon, codon->b); }
char imm_codon_csym(struct imm_codon const *codon) { return codon_sym(codon, codon->c); }

char imm_gc_decode(unsigned table, struct imm_codon codon)
{
    assert(table == 1);
    (void)table;
    uint8_t ids[3] = {(uint8_t)codon.a, (uint8_t)codon.b, (uint8_t)codon.c};
    return dcp_gc_decode(ids);
}

/* ---- rnd / lprob --------------------------------------------------------------------------------- */
struct imm_rnd imm_rnd(uint64_t seed)
{
    struct imm_rnd r;
    dcp_rnd_seed(r.data, seed);
    return r;
}

double imm_rnd_dbl(struct imm_rnd *rnd) { return dcp_rnd_next(rnd->data); }

void imm_lprob_sample(struct imm_rnd *rnd, unsigned len, imm_float *lprobs)
{
    for (unsigned i = 0; i < len; ++i)
        lprobs[i] = (imm_float)log(imm_rnd_dbl(rnd));
}

void imm_lprob_normalize(unsigned len, imm_float *lprobs) { dcp_lprob_normalize(len, lprobs); }

struct imm_frame_epsilon imm_frame_epsilon(imm_float epsilon)
{
    return (struct imm_frame_epsilon){imm_log(epsilon), imm_log(1 - epsilon)};
}

/* ---- task / prod / path ---------------------------------------------------------------------------- */
struct imm_task *imm_task_new(struct imm_dp const *dp)
{
    struct imm_task *t = malloc(sizeof *t);
    if (!t) return NULL;
    t->dp = dp;
    t->seq = NULL;
    return t;
}

enum imm_rc imm_task_reset(struct imm_task *task, struct imm_dp const *dp)
{
    task->dp = dp;
    task->seq = NULL;
    return IMM_OK;
}

enum imm_rc imm_task_setup(struct imm_task *task, struct imm_seq const *seq)
{
    if (!task || !seq) return IMM_FAILURE;
    task->seq = seq;
    return IMM_OK;
}

void imm_task_del(struct imm_task const *task) { free((void *)task); }

struct imm_prod imm_prod(void) { return (struct imm_prod){{NULL, 0, 0}, NAN, 0}; }

void imm_prod_reset(struct imm_prod *prod)
{
    prod->path.nsteps = 0;
    prod->loglik = NAN;
    prod->mseconds = 0;
}

void imm_prod_del(struct imm_prod const *prod) { free(prod->path.steps); }

enum rc dcp_host_path_assign(struct imm_path *path, struct dcp_step const *steps, unsigned n)
{
    if (path->capacity < n)
    {
        struct imm_step *p = realloc(path->steps, (size_t)n * sizeof *p);
        if (!p) return fail(RC_ENOMEM, "alloc path");
        path->steps = p;
        path->capacity = n;
    }
    for (unsigned i = 0; i < n; ++i)
        path->steps[i] = (struct imm_step){steps[i].state_id, steps[i].seqlen};
    path->nsteps = n;
    return RC_OK;
}

/* symbol ids 0..3 of a sequence over its own 4-letter alphabet (DNA or RNA); NULL + EINVAL for a symbol
 * outside it.  (What imm does with the any-symbol in a frame state's emission is not pinned by anything in
 * the reference tree -- SURVEY.md Appendix C -- so it is rejected, not guessed.) */
uint8_t *dcp_host_seq_ids(struct imm_seq const *seq, enum rc *rc)
{
    uint8_t *ids = malloc(seq->size ? seq->size : 1);
    if (!ids)
    {
        *rc = fail(RC_ENOMEM, "alloc sequence");
        return NULL;
    }
    for (unsigned i = 0; i < seq->size; ++i)
    {
        int const id = imm_abc_symbol_idx(seq->abc, seq->str[i]);
        if (id < 0 || id > 3 || seq->abc->size != 4)
        {
            *rc = fail(RC_EINVAL, "symbol '%c' at %u is not one of \"%s\"", seq->str[i], i, seq->abc->symbols);
            free(ids);
            return NULL;
        }
        ids[i] = (uint8_t)id;
    }
    *rc = RC_OK;
    return ids;
}

/* one shared device context for single-pair imm_dp_viterbi calls (library-level use, as the
 * reference's tests do); thread_run owns one context per partition instead */
static dcp_gpu_ctx *g_ctx;
static dcp_profile *g_ctx_db;
/* the reference calls imm_dp_viterbi from every thread of its OpenMP team (scan.c:239-249): the shared
 * context serialises such callers instead of racing (thread_run has a context per partition and never
 * comes here) */
static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;

static void release_shared_ctx(void)
{
    if (g_ctx) dcp_gpu_ctx_del(g_ctx);
    g_ctx = NULL;
    g_ctx_db = NULL;
}

static dcp_gpu_ctx *shared_ctx(void)
{
    if (!g_ctx)
    {
        g_ctx = dcp_gpu_ctx_new(0);
        if (g_ctx) atexit(release_shared_ctx);
    }
    return g_ctx;
}

void dcp_host_forget_profile(dcp_profile *impl)
{
    if (!impl) return;
    pthread_mutex_lock(&g_ctx_lock);
    if (g_ctx_db == impl) g_ctx_db = NULL;
    pthread_mutex_unlock(&g_ctx_lock);
}

static enum imm_rc viterbi_locked(struct imm_dp const *dp, struct imm_task *task, struct imm_prod *prod);

enum imm_rc imm_dp_viterbi(struct imm_dp const *dp, struct imm_task *task, struct imm_prod *prod)
{
    pthread_mutex_lock(&g_ctx_lock);
    enum imm_rc rc = viterbi_locked(dp, task, prod);
    pthread_mutex_unlock(&g_ctx_lock);
    return rc;
}

static enum imm_rc viterbi_locked(struct imm_dp const *dp, struct imm_task *task, struct imm_prod *prod)
{
    if (!dp || !task || !prod || !task->seq) return IMM_FAILURE;
    struct protein_profile *prof = dp->owner;
    struct imm_seq const *seq = task->seq;
    if (!prof || !prof->impl)
    {
        fail(RC_EINVAL, "dp holds no model (protein_profile_sample / _absorb / _unpack first)");
        return IMM_FAILURE;
    }
    if (seq->size == 0)
    {
        fail(RC_EINVAL, "sequence cannot be empty");
        return IMM_FAILURE;
    }
    dcp_gpu_ctx *ctx = shared_ctx();
    if (!ctx)
    {
        fail(RC_EFAIL, "no HIP device: imm_dp_viterbi has no CPU implementation here");
        return IMM_FAILURE;
    }
    enum rc rc = RC_OK;
    uint8_t *ids = dcp_host_seq_ids(seq, &rc);
    if (!ids) return IMM_FAILURE;
    if (g_ctx_db != prof->impl)
    {
        dcp_profile *one[1] = {prof->impl};
        g_ctx_db = NULL;
        if (dcp_gpu_db_upload(ctx, one, 1, 0))
        {
            free(ids);
            return IMM_FAILURE;
        }
        g_ctx_db = prof->impl;
    }
    uint32_t off[2] = {0, seq->size};
    int drc = dcp_gpu_seqs_upload(ctx, ids, off, 1);
    free(ids);
    if (drc) return IMM_FAILURE;
    /* the transitions the profile holds NOW: protein_profile_setup's, or the LOG1 defaults */
    if (dcp_gpu_seqs_set_xtrans(ctx, prof->xtrans, 1)) return IMM_FAILURE;
    struct dcp_scan_params prm = {1, 0, 10.0f, 1, 1 /* row sweep */};
    if (dcp_gpu_scan(ctx, &prm) || dcp_gpu_sync(ctx)) return IMM_FAILURE;
    float nul = NAN, alt = NAN;
    if (dcp_gpu_fetch_scores(ctx, &nul, &alt)) return IMM_FAILURE;

    struct dcp_hit pair = {0, 0, nul, alt};
    unsigned cap = 2 * seq->size + 2 * prof->core_size + 16;
    struct dcp_step *steps = malloc((size_t)cap * sizeof *steps);
    if (!steps) return IMM_FAILURE;
    uint32_t soff[2] = {0, 0};
    float traced = NAN;
    drc = dcp_gpu_trace_paths(ctx, &pair, 1, 1, 0, dp->null_model, steps, cap, soff, &traced);
    enum imm_rc out = IMM_FAILURE;
    if (!drc && !dcp_host_path_assign(&prod->path, steps, soff[1]))
    {
        prod->loglik = dp->null_model ? nul : alt;
        out = traced == prod->loglik ? IMM_OK : IMM_FAILURE; /* the trace recomputes the score bit for bit */
    }
    free(steps);
    return out;
}

/* ---- imm_dp: a handle on the owning profile ------------------------------------------------------- */
void imm_dp_init(struct imm_dp *dp, struct imm_code const *code)
{
    dp->owner = NULL;
    dp->null_model = 0;
    dp->code = code;
}

void imm_dp_del(struct imm_dp const *dp) { (void)dp; }

/* The 13 special transitions, addressed like protein_profile_setup addresses them
 * (src/model/protein_profile.c:186-214): by (source, destination) state index. */
unsigned imm_dp_trans_idx(struct imm_dp *dp, unsigned src, unsigned dst)
{
    struct protein_profile const *p = dp->owner;
    if (!p) return UINT_MAX;
    if (dp->null_model) return src == p->null.R && dst == p->null.R ? DCP_HOST_X_RR : UINT_MAX;
    unsigned const S = p->alt.S, N = p->alt.N, B = p->alt.B, E = p->alt.E, J = p->alt.J, C = p->alt.C, T = p->alt.T;
    if (src == S && dst == B) return DCP_HOST_X_SB;
    if (src == S && dst == N) return DCP_HOST_X_SN;
    if (src == N && dst == N) return DCP_HOST_X_NN;
    if (src == N && dst == B) return DCP_HOST_X_NB;
    if (src == E && dst == T) return DCP_HOST_X_ET;
    if (src == E && dst == C) return DCP_HOST_X_EC;
    if (src == C && dst == C) return DCP_HOST_X_CC;
    if (src == C && dst == T) return DCP_HOST_X_CT;
    if (src == E && dst == B) return DCP_HOST_X_EB;
    if (src == E && dst == J) return DCP_HOST_X_EJ;
    if (src == J && dst == J) return DCP_HOST_X_JJ;
    if (src == J && dst == B) return DCP_HOST_X_JB;
    return UINT_MAX;
}

void imm_dp_change_trans(struct imm_dp *dp, unsigned trans_idx, imm_float lprob)
{
    assert(dp->owner && trans_idx < DCP_NXTRANS);
    if (dp->owner && trans_idx < DCP_NXTRANS) dp->owner->xtrans[trans_idx] = lprob;
}

void imm_dp_write_dot(struct imm_dp const *dp, FILE *fp, imm_state_name *name)
{
    /* the alt model's core as a Graphviz digraph: enough for a human to look at, nothing parses it */
    struct protein_profile const *p = dp->owner;
    fprintf(fp, "digraph hmm {\n");
    if (p && p->impl && !dp->null_model)
    {
        float const *t8 = dcp_profile_trans8(p->impl);
        unsigned const M = p->core_size;
        char a[IMM_STATE_NAME_SIZE], b[IMM_STATE_NAME_SIZE];
        for (unsigned k = 0; k < M; ++k)
        {
            name(PROTEIN_MATCH_STATE | (k + 1), a);
            fprintf(fp, "B -> %s [label=%.4f];\n", a, (double)t8[0 * M + k]);
            if (k + 1 < M)
            {
                name(PROTEIN_MATCH_STATE | (k + 2), b);
                fprintf(fp, "%s -> %s [label=%.4f];\n", a, b, (double)t8[1 * M + k + 1]);
            }
        }
    }
    else if (p && dp->null_model)
        fprintf(fp, "R -> R [label=%.4f];\n", (double)p->xtrans[DCP_HOST_X_RR]);
    fprintf(fp, "}\n");
}

/* ---- protein state --------------------------------------------------------------------------------- */
static unsigned state_msb(unsigned id) { return id & (3U << (BITS_PER_PROFILE_TYPEID - 2)); }
bool protein_state_is_match(unsigned id) { return state_msb(id) == PROTEIN_MATCH_STATE; }
bool protein_state_is_insert(unsigned id) { return state_msb(id) == PROTEIN_INSERT_STATE; }
bool protein_state_is_delete(unsigned id) { return state_msb(id) == PROTEIN_DELETE_STATE; }
bool protein_state_is_mute(unsigned id)
{
    if (state_msb(id) == PROTEIN_EXT_STATE)
        return id == PROTEIN_S_STATE || id == PROTEIN_B_STATE || id == PROTEIN_E_STATE || id == PROTEIN_T_STATE;
    return state_msb(id) == PROTEIN_DELETE_STATE;
}
unsigned protein_state_idx(unsigned id) { return (id & (0xFFFF >> 2)) - 1; }
unsigned protein_state_name(unsigned id, char name[IMM_STATE_NAME_SIZE]) { return dcp_state_name(id, name); }

char const *profile_typeid_name(enum profile_typeid typeid)
{
    static char const *const names[] = {"null", "standard", "protein"};
    return (unsigned)typeid < 3 ? names[typeid] : "unknown";
}

/* ---- protein_model: collects the amino-acid level parameters ------------------------------------------ */
void protein_model_init(struct protein_model *m, struct imm_amino const *amino, struct imm_nuclt_code const *code,
                        struct protein_cfg cfg, imm_float const null_lprobs[IMM_AMINO_SIZE])
{
    m->amino = amino;
    m->code = code;
    m->cfg = cfg;
    m->core_size = 0;
    m->consensus[0] = '\0';
    memcpy(m->null.lprobs, null_lprobs, sizeof m->null.lprobs);
    m->alt.node_idx = UINT_MAX; /* "setup not called yet" (protein_model.c:131,134) */
    m->alt.match_lprobs = NULL;
    m->alt.trans_idx = UINT_MAX;
    m->alt.trans = NULL;
    protein_xtrans_init(&m->xtrans);
}

enum rc protein_model_setup(struct protein_model *m, unsigned core_size)
{
    if (core_size == 0) return fail(RC_EINVAL, "`core_size` cannot be zero");
    if (core_size > PROTEIN_MODEL_CORE_SIZE_MAX) return fail(RC_EINVAL, "`core_size` is too big");
    void *nodes = realloc(m->alt.match_lprobs, (size_t)core_size * sizeof *m->alt.match_lprobs);
    if (!nodes) return fail(RC_ENOMEM, "failed to alloc nodes");
    m->alt.match_lprobs = nodes;
    void *trans = realloc(m->alt.trans, ((size_t)core_size + 1) * sizeof *m->alt.trans);
    if (!trans) return fail(RC_ENOMEM, "failed to alloc trans");
    m->alt.trans = trans;
    m->core_size = core_size;
    m->consensus[core_size] = '\0';
    m->alt.node_idx = 0;
    m->alt.trans_idx = 0;
    return RC_OK;
}

static bool model_setup_called(struct protein_model const *m) { return m->core_size > 0; }
static bool model_complete(struct protein_model const *m)
{
    return model_setup_called(m) && m->alt.node_idx == m->core_size && m->alt.trans_idx == m->core_size + 1;
}

enum rc protein_model_add_node(struct protein_model *m, imm_float const lp[IMM_AMINO_SIZE], char consensus)
{
    if (!model_setup_called(m)) return fail(RC_EFAIL, "must call protein_model_setup first");
    if (m->alt.node_idx == m->core_size) return fail(RC_EFAIL, "reached limit of nodes");
    m->consensus[m->alt.node_idx] = consensus;
    memcpy(m->alt.match_lprobs[m->alt.node_idx], lp, sizeof m->alt.match_lprobs[0]);
    m->alt.node_idx++;
    return RC_OK;
}

enum rc protein_model_add_trans(struct protein_model *m, struct protein_trans trans)
{
    if (!model_setup_called(m)) return fail(RC_EFAIL, "must call protein_model_setup first");
    if (m->alt.trans_idx == m->core_size + 1) return fail(RC_EFAIL, "reached limit of transitions");
    m->alt.trans[m->alt.trans_idx++] = trans;
    return RC_OK;
}

void protein_model_del(struct protein_model const *m)
{
    free(m->alt.match_lprobs);
    free(m->alt.trans);
}

struct imm_amino const *protein_model_amino(struct protein_model const *m) { return m->amino; }
struct imm_nuclt const *protein_model_nuclt(struct protein_model const *m) { return m->code->nuclt; }

/* ---- profile ------------------------------------------------------------------------------------------ */
void profile_init(struct profile *prof, char const *accession, struct imm_code const *code,
                  struct profile_vtable vtable, imm_state_name *state_name)
{
    prof->vtable = vtable;
    snprintf(prof->accession, sizeof prof->accession, "%s", accession ? accession : "");
    prof->state_name = state_name;
    prof->code = code;
}

void profile_del(struct profile *prof)
{
    if (prof && prof->vtable.del) prof->vtable.del(prof);
}
enum rc profile_unpack(struct profile *prof, struct lip_file *file) { return prof->vtable.unpack(prof, file); }
int profile_typeid(struct profile const *prof) { return prof->vtable.typeid; }
struct imm_dp const *profile_null_dp(struct profile const *prof) { return prof->vtable.null_dp(prof); }
struct imm_dp const *profile_alt_dp(struct profile const *prof) { return prof->vtable.alt_dp(prof); }

/* ---- nuclt_dist <-> the compact profile's 129-float rows ------------------------------------------------ */
static void ndist_from_row(struct nuclt_dist *d, float const row[DCP_NDIST])
{
    memcpy(d->nucltp.lprobs, row, sizeof d->nucltp.lprobs);
    memcpy(d->codonm.lprobs, row + IMM_NUCLT_SIZE, sizeof d->codonm.lprobs);
}

static void ndist_to_row(struct nuclt_dist const *d, float row[DCP_NDIST])
{
    memcpy(row, d->nucltp.lprobs, sizeof d->nucltp.lprobs);
    memcpy(row + IMM_NUCLT_SIZE, d->codonm.lprobs, sizeof d->codonm.lprobs);
}

enum rc nuclt_dist_pack(struct nuclt_dist const *ndist, struct lip_file *file)
{
    if (!lip_write_array_size(file, 2)) return RC_EFAIL;
    if (imm_nuclt_lprob_pack(&ndist->nucltp, file)) return RC_EFAIL;
    if (imm_codon_marg_pack(&ndist->codonm, file)) return RC_EFAIL;
    return RC_OK;
}

enum rc nuclt_dist_unpack(struct nuclt_dist *ndist, struct lip_file *file)
{
    unsigned size = 0;
    if (!lip_read_array_size(file, &size) || size != 2) return RC_EFAIL;
    if (imm_nuclt_lprob_unpack(&ndist->nucltp, file)) return RC_EFAIL;
    if (imm_codon_marg_unpack(&ndist->codonm, file)) return RC_EFAIL;
    return RC_OK;
}

/* ---- protein_profile ------------------------------------------------------------------------------------- */
static void protein_del(struct profile *prof)
{
    if (!prof) return;
    struct protein_profile *p = (struct protein_profile *)prof;
    free(p->alt.match_ndists);
    p->alt.match_ndists = NULL;
    imm_dp_del(&p->null.dp);
    imm_dp_del(&p->alt.dp);
    dcp_host_forget_profile(p->impl);
    dcp_profile_del(p->impl);
    p->impl = NULL;
}
static struct imm_dp const *protein_null_dp(struct profile const *prof)
{
    return &((struct protein_profile const *)prof)->null.dp;
}
static struct imm_dp const *protein_alt_dp(struct profile const *prof)
{
    return &((struct protein_profile const *)prof)->alt.dp;
}
static enum rc protein_unpack(struct profile *prof, struct lip_file *file)
{
    return protein_profile_unpack((struct protein_profile *)prof, file);
}

void protein_profile_init(struct protein_profile *p, char const *accession, struct imm_amino const *amino,
                          struct imm_nuclt_code const *code, struct protein_cfg cfg)
{
    struct profile_vtable const vtable = {PROFILE_PROTEIN, protein_del, protein_unpack, protein_null_dp,
                                          protein_alt_dp};
    profile_init(&p->super, accession, &code->super, vtable, protein_state_name);
    p->amino = amino;
    p->code = code;
    p->cfg = cfg;
    p->eps = imm_frame_epsilon(cfg.epsilon);
    p->core_size = 0;
    p->consensus[0] = '\0';
    imm_dp_init(&p->null.dp, &code->super);
    imm_dp_init(&p->alt.dp, &code->super);
    p->null.dp.owner = p->alt.dp.owner = p;
    p->null.dp.null_model = 1;
    nuclt_dist_init(&p->null.ndist, code->nuclt);
    nuclt_dist_init(&p->alt.insert_ndist, code->nuclt);
    p->alt.match_ndists = NULL;
    /* state indices as imm_state_idx reports them after the HMM -> DP compile: the null model has the
     * single state R; the alt model's specials were added first (protein_model.c:227-233) */
    p->null.R = 0;
    p->alt.S = 0, p->alt.N = 1, p->alt.B = 2, p->alt.E = 3, p->alt.J = 4, p->alt.C = 5, p->alt.T = 6;
    p->impl = NULL;
    for (int i = 0; i < DCP_NXTRANS; ++i)
        p->xtrans[i] = IMM_LPROB_ONE;
}

enum rc protein_profile_setup(struct protein_profile *prof, unsigned seq_size, bool multi_hits, bool hmmer3_compat)
{
    /* the 13 values protein_profile.c:155-216 computes, written through imm_dp_trans_idx /
     * imm_dp_change_trans like there */
    float xt[DCP_NXTRANS];
    if (dcp_xtrans(seq_size, multi_hits, hmmer3_compat, xt)) return fail(RC_EINVAL, "sequence cannot be empty");
    struct imm_dp *dp = &prof->null.dp;
    unsigned const R = prof->null.R;
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, R, R), xt[DCP_HOST_X_RR]);
    dp = &prof->alt.dp;
    unsigned const S = prof->alt.S, N = prof->alt.N, B = prof->alt.B, E = prof->alt.E, J = prof->alt.J,
                   C = prof->alt.C, T = prof->alt.T;
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, S, B), xt[DCP_HOST_X_SB]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, S, N), xt[DCP_HOST_X_SN]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, N, N), xt[DCP_HOST_X_NN]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, N, B), xt[DCP_HOST_X_NB]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, E, T), xt[DCP_HOST_X_ET]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, E, C), xt[DCP_HOST_X_EC]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, C, C), xt[DCP_HOST_X_CC]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, C, T), xt[DCP_HOST_X_CT]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, E, B), xt[DCP_HOST_X_EB]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, E, J), xt[DCP_HOST_X_EJ]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, J, J), xt[DCP_HOST_X_JJ]);
    imm_dp_change_trans(dp, imm_dp_trans_idx(dp, J, B), xt[DCP_HOST_X_JB]);
    return RC_OK;
}

/* take ownership of a freshly built compact profile and mirror it into the reference's members */
enum rc dcp_host_adopt(struct protein_profile *p, dcp_profile *impl, int rc)
{
    if (!impl) return fail((enum rc)rc, "failed to build the profile");
    unsigned const M = dcp_profile_core_size(impl);
    struct nuclt_dist *nd = realloc(p->alt.match_ndists, (size_t)M * sizeof *nd);
    if (!nd)
    {
        dcp_profile_del(impl);
        return fail(RC_ENOMEM, "alloc nuclt dists");
    }
    p->alt.match_ndists = nd;
    if (p->impl)
    {
        dcp_host_forget_profile(p->impl);
        dcp_profile_del(p->impl);
    }
    p->impl = impl;
    p->core_size = M;
    memcpy(p->consensus, dcp_profile_consensus(impl), (size_t)M + 1);
    ndist_from_row(&p->null.ndist, dcp_profile_null_dist(impl));
    ndist_from_row(&p->alt.insert_ndist, dcp_profile_insert_dist(impl));
    float const *md = dcp_profile_match_dist(impl);
    for (unsigned k = 0; k < M; ++k)
    {
        nuclt_dist_init(nd + k, p->code->nuclt);
        ndist_from_row(nd + k, md + (size_t)k * DCP_NDIST);
    }
    /* a fresh DP: specials back at LOG1 (protein_model.c:322-340) until protein_profile_setup */
    for (int i = 0; i < DCP_NXTRANS; ++i)
        p->xtrans[i] = IMM_LPROB_ONE;
    return RC_OK;
}

enum rc protein_profile_absorb(struct protein_profile *p, struct protein_model const *m)
{
    if (p->code->nuclt != protein_model_nuclt(m)) return fail(RC_EINVAL, "Different nucleotide alphabets.");
    if (p->amino != protein_model_amino(m)) return fail(RC_EINVAL, "Different amino alphabets.");
    if (!model_complete(m)) return fail(RC_EINVAL, "model is incomplete: %u of %u nodes, %u of %u transitions",
                                        m->alt.node_idx == UINT_MAX ? 0 : m->alt.node_idx, m->core_size,
                                        m->alt.trans_idx == UINT_MAX ? 0 : m->alt.trans_idx, m->core_size + 1);
    int rc = 0;
    dcp_profile *impl = dcp_profile_new(p->super.accession, m->core_size, (int)m->cfg.entry_dist, m->cfg.epsilon,
                                        m->null.lprobs, &m->alt.match_lprobs[0][0], &m->alt.trans[0].data[0],
                                        m->consensus, &rc);
    return dcp_host_adopt(p, impl, rc);
}

enum rc protein_profile_sample(struct protein_profile *p, unsigned seed, unsigned core_size)
{
    int rc = 0;
    dcp_profile *impl = dcp_profile_sample(p->super.accession, seed, core_size, (int)p->cfg.entry_dist,
                                           p->cfg.epsilon, &rc);
    return dcp_host_adopt(p, impl, rc);
}

enum rc protein_profile_from_params(struct protein_profile *p, unsigned core_size, imm_float const *null_lprobs,
                                    imm_float const *match_lprobs, imm_float const *trans)
{
    int rc = 0;
    dcp_profile *impl = dcp_profile_new(p->super.accession, core_size, (int)p->cfg.entry_dist, p->cfg.epsilon,
                                        null_lprobs, match_lprobs, trans, NULL, &rc);
    return dcp_host_adopt(p, impl, rc);
}

enum rc protein_profile_decode(struct protein_profile const *prof, struct imm_seq const *seq, unsigned state_id,
                               struct imm_codon *codon)
{
    assert(!protein_state_is_mute(state_id));
    uint8_t frag[5], out[3];
    if (seq->size < 1 || seq->size > 5) return fail(RC_EINVAL, "failed to decode sequence");
    for (unsigned i = 0; i < seq->size; ++i)
    {
        int const id = imm_abc_symbol_idx(seq->abc, seq->str[i]);
        if (id < 0 || id > 3) return fail(RC_EINVAL, "failed to decode sequence");
        frag[i] = (uint8_t)id;
    }
    if (dcp_profile_decode(prof->impl, frag, seq->size, state_id, out))
        return fail(RC_EINVAL, "failed to decode sequence");
    codon->nuclt = prof->code->nuclt;
    codon->a = out[0], codon->b = out[1], codon->c = out[2];
    return RC_OK;
}

void protein_profile_write_dot(struct protein_profile const *p, FILE *fp)
{
    imm_dp_write_dot(&p->alt.dp, fp, protein_state_name);
}

/* ---- pack / unpack: the map(16) of src/model/protein_profile.c:38-117,338-400 ------------------------------
 * Key order and value types are the reference's.  The two imm_dp values are imm's own serialisation
 * there (its layout lives in the absent imm library); this library writes its own dp value
 *     map(3) {"fmt": "dcp-dp-1", "xtrans": 1darray f32 [13 | 1], "trans8": 1darray f32 [8 * M | 0]}
 * (alt: all 13 specials + the core transitions; null: RR alone) and on reading accepts exactly
 * that.  A dp value in any other layout -- a file pressed by the reference -- is skipped as one
 * MessagePack object, everything else of the profile is still parsed and validated, and the
 * unpack ends with RC_EPARSE "transitions live in imm's dp serialisation: unsupported". */
static char const kDpFormat[] = "dcp-dp-1";

enum imm_rc imm_dp_pack(struct imm_dp const *dp, struct lip_file *file)
{
    struct protein_profile const *p = dp->owner;
    if (!lip_write_map_size(file, 3)) return IMM_FAILURE;
    if (!lip_write_cstr(file, "fmt") || !lip_write_cstr(file, kDpFormat)) return IMM_FAILURE;
    bool const alt = p && !dp->null_model;
    unsigned const nx = alt ? DCP_NXTRANS : (p ? 1u : 0u);
    unsigned const nt = alt && p->impl ? 8u * p->core_size : 0u;
    if (!lip_write_cstr(file, "xtrans") || !lip_write_1darray_size_type(file, nx, LIP_1DARRAY_F32)) return IMM_FAILURE;
    if (nx && !lip_write_1darray_f32_data(file, nx, p->xtrans)) return IMM_FAILURE;
    if (!lip_write_cstr(file, "trans8") || !lip_write_1darray_size_type(file, nt, LIP_1DARRAY_F32)) return IMM_FAILURE;
    if (nt && !lip_write_1darray_f32_data(file, nt, dcp_profile_trans8(p->impl))) return IMM_FAILURE;
    return IMM_OK;
}

struct dp_blob
{
    bool foreign;
    unsigned nx, nt;
    float xtrans[DCP_NXTRANS];
    float *trans8;
};

static enum imm_rc dp_blob_read(struct dp_blob *b, struct lip_file *file)
{
    memset(b, 0, sizeof *b);
    long const at = ftell(file->fp);
    unsigned n = 0;
    char fmt[16] = {0};
    if (at >= 0 && lip_read_map_size(file, &n) && n == 3 && expect_map_key(file, "fmt") &&
        lip_read_cstr(file, sizeof fmt, fmt) && !strcmp(fmt, kDpFormat))
    {
        enum lip_1darray_type ty;
        if (!expect_map_key(file, "xtrans") || !lip_read_1darray_size_type(file, &b->nx, &ty) ||
            ty != LIP_1DARRAY_F32 || b->nx > DCP_NXTRANS || !lip_read_1darray_f32_data(file, b->nx, b->xtrans))
            return IMM_FAILURE;
        if (!expect_map_key(file, "trans8") || !lip_read_1darray_size_type(file, &b->nt, &ty) ||
            ty != LIP_1DARRAY_F32 || b->nt > 8u * PROTEIN_MODEL_CORE_SIZE_MAX)
            return IMM_FAILURE;
        if (b->nt)
        {
            b->trans8 = malloc((size_t)b->nt * sizeof(float));
            if (!b->trans8 || !lip_read_1darray_f32_data(file, b->nt, b->trans8))
            {
                free(b->trans8);
                b->trans8 = NULL;
                return IMM_FAILURE;
            }
        }
        return IMM_OK;
    }
    /* not this library's dp value: rewind and step over it as one opaque object */
    if (at < 0 || fseek(file->fp, at, SEEK_SET)) return IMM_FAILURE;
    file->error = false;
    b->foreign = true;
    return lip_skip_object(file) ? IMM_OK : IMM_FAILURE;
}

enum imm_rc imm_dp_unpack(struct imm_dp *dp, struct lip_file *file)
{
    /* stand-alone form (standard_profile_unpack): consume and check the value */
    struct dp_blob b;
    (void)dp;
    enum imm_rc rc = dp_blob_read(&b, file);
    free(b.trans8);
    return rc;
}

static enum rc read_state_idx(struct lip_file *file, char const *key, unsigned *out)
{
    if (!expect_map_key(file, key)) return fail(RC_EIO, "skip key");
    if (!lip_read_unsigned(file, out)) return fail(RC_EIO, "read %s state", key);
    return RC_OK;
}

enum rc protein_profile_unpack(struct protein_profile *p, struct lip_file *file)
{
    struct profile *prof = &p->super;
    unsigned size = 0;
    if (!lip_read_map_size(file, &size)) return fail(RC_EIO, "read profile map size");
    if (size != 16) return fail(RC_EPARSE, "profile map has %u keys, expected 16", size);

    if (!expect_map_key(file, "accession")) return fail(RC_EIO, "read key");
    if (!lip_read_cstr(file, PROFILE_ACC_SIZE, prof->accession)) return fail(RC_EIO, "read accession");

    struct dp_blob nul, alt;
    if (!expect_map_key(file, "null")) return fail(RC_EIO, "skip key");
    if (dp_blob_read(&nul, file)) return fail(RC_EFAIL, "read null dp");
    free(nul.trans8);
    nul.trans8 = NULL;
    if (!expect_map_key(file, "alt") || dp_blob_read(&alt, file))
    {
        return fail(file->error ? RC_EIO : RC_EFAIL, "read alt dp");
    }
    enum rc rc = RC_OK;
    float *rows = NULL;
    char consensus[PROTEIN_MODEL_CORE_SIZE_MAX + 1];

    if (!expect_map_key(file, "core_size") || !lip_read_unsigned(file, &size))
    {
        rc = fail(RC_EIO, "read core size");
        goto done;
    }
    if (size == 0 || size > PROTEIN_MODEL_CORE_SIZE_MAX)
    {
        rc = fail(RC_EIO, "profile is too long");
        goto done;
    }
    unsigned const M = size;
    if (!expect_map_key(file, "consensus") || !lip_read_cstr(file, sizeof consensus, consensus))
    {
        rc = fail(RC_EIO, "read consensus");
        goto done;
    }
    unsigned st[8];
    static char const *const keys[8] = {"R", "S", "N", "B", "E", "J", "C", "T"};
    for (int i = 0; i < 8 && !rc; ++i)
        rc = read_state_idx(file, keys[i], &st[i]);
    if (rc) goto done;

    rows = malloc(((size_t)M + 2) * DCP_NDIST * sizeof(float)); /* null, insert, match[M] */
    if (!rows)
    {
        rc = fail(RC_ENOMEM, "alloc nuclt dists");
        goto done;
    }
    struct nuclt_dist nd;
    nuclt_dist_init(&nd, p->code->nuclt);
    if (!expect_map_key(file, "null_ndist") || nuclt_dist_unpack(&nd, file))
    {
        rc = fail(RC_EIO, "read null_ndist");
        goto done;
    }
    ndist_to_row(&nd, rows);
    if (!expect_map_key(file, "alt_insert_ndist") || nuclt_dist_unpack(&nd, file))
    {
        rc = fail(RC_EIO, "read alt_insert_ndist");
        goto done;
    }
    ndist_to_row(&nd, rows + DCP_NDIST);
    if (!expect_map_key(file, "alt_match_ndist") || !lip_read_array_size(file, &size))
    {
        rc = fail(RC_EIO, "read size");
        goto done;
    }
    if (size != M)
    {
        rc = fail(RC_EPARSE, "alt_match_ndist has %u entries for core_size %u", size, M);
        goto done;
    }
    for (unsigned i = 0; i < M; ++i)
    {
        if (nuclt_dist_unpack(&nd, file))
        {
            rc = fail(RC_EIO, "read alt_match_ndist");
            goto done;
        }
        ndist_to_row(&nd, rows + ((size_t)i + 2) * DCP_NDIST);
    }

    /* everything but the transitions is in hand; they are in the alt dp value */
    if (alt.foreign || nul.foreign)
    {
        rc = fail(RC_EPARSE,
                  "profile '%s' (core_size %u): its transitions live in imm's dp serialisation, which this "
                  "library cannot read (framing, consensus and %u nuclt_dists parsed and valid)",
                  prof->accession, M, M + 2);
        goto done;
    }
    if (alt.nt != 8u * M || alt.nx != DCP_NXTRANS || nul.nx != 1)
    {
        rc = fail(RC_EPARSE, "dp value does not match core_size %u", M);
        goto done;
    }
    int drc = 0;
    dcp_profile *impl = dcp_profile_from_parts(prof->accession, M, (int)p->cfg.entry_dist, p->cfg.epsilon, consensus,
                                               alt.trans8, rows, rows + DCP_NDIST, rows + 2 * DCP_NDIST, &drc);
    rc = dcp_host_adopt(p, impl, drc);
    if (rc) goto done;
    p->null.R = st[0];
    p->alt.S = st[1], p->alt.N = st[2], p->alt.B = st[3], p->alt.E = st[4], p->alt.J = st[5], p->alt.C = st[6],
    p->alt.T = st[7];
    /* the dp carries whatever specials it was packed with (LOG1 for a pressed DB) */
    memcpy(p->xtrans, alt.xtrans, sizeof p->xtrans);
    p->xtrans[DCP_HOST_X_RR] = nul.xtrans[0];

done:
    free(rows);
    free(alt.trans8);
    return rc;
}

enum rc protein_profile_pack(struct protein_profile const *prof, struct lip_file *file)
{
    if (!prof->impl) return fail(RC_EINVAL, "profile holds no model");
    if (!lip_write_map_size(file, 16)) return fail(RC_EIO, "write profile map size");
    if (!lip_write_cstr(file, "accession") || !lip_write_cstr(file, prof->super.accession))
        return fail(RC_EIO, "write accession");
    if (!lip_write_cstr(file, "null") || imm_dp_pack(&prof->null.dp, file)) return fail(RC_EFAIL, "write null dp");
    if (!lip_write_cstr(file, "alt") || imm_dp_pack(&prof->alt.dp, file)) return fail(RC_EFAIL, "write alt dp");
    if (!lip_write_cstr(file, "core_size") || !lip_write_int(file, prof->core_size)) return fail(RC_EIO, "write core_size");
    if (!lip_write_cstr(file, "consensus") || !lip_write_cstr(file, prof->consensus)) return fail(RC_EIO, "write consensus");
    unsigned const st[8] = {prof->null.R, prof->alt.S, prof->alt.N, prof->alt.B,
                            prof->alt.E,  prof->alt.J, prof->alt.C, prof->alt.T};
    static char const *const keys[8] = {"R", "S", "N", "B", "E", "J", "C", "T"};
    for (int i = 0; i < 8; ++i)
        if (!lip_write_cstr(file, keys[i]) || !lip_write_int(file, st[i])) return fail(RC_EIO, "write %s state", keys[i]);
    enum rc rc;
    if (!lip_write_cstr(file, "null_ndist")) return fail(RC_EIO, "write null_ndist key");
    if ((rc = nuclt_dist_pack(&prof->null.ndist, file))) return rc;
    if (!lip_write_cstr(file, "alt_insert_ndist")) return fail(RC_EIO, "write alt_insert_ndist key");
    if ((rc = nuclt_dist_pack(&prof->alt.insert_ndist, file))) return rc;
    if (!lip_write_cstr(file, "alt_match_ndist") || !lip_write_array_size(file, prof->core_size))
        return fail(RC_EIO, "write array length");
    for (unsigned i = 0; i < prof->core_size; ++i)
        if ((rc = nuclt_dist_pack(prof->alt.match_ndists + i, file))) return rc;
    return RC_OK;
}

/* ---- imm value serialisations the db code calls ------------------------------------------------------------
 * imm_abc: map {"symbols": str, "any_symbol": uint, "typeid": uint}; the reader takes these keys from a
 * map in any order and skips keys it does not know (imm's own layout carries more). */
enum imm_rc imm_abc_pack(struct imm_abc const *abc, struct lip_file *file)
{
    bool ok = lip_write_map_size(file, 3) && lip_write_cstr(file, "symbols") && lip_write_cstr(file, abc->symbols) &&
              lip_write_cstr(file, "any_symbol") && lip_write_uint(file, (unsigned char)abc->any_symbol) &&
              lip_write_cstr(file, "typeid") && lip_write_uint(file, (uint64_t)abc->vtable.typeid);
    return ok ? IMM_OK : IMM_FAILURE;
}

enum imm_rc imm_abc_unpack(struct imm_abc *abc, struct lip_file *file)
{
    unsigned n = 0;
    if (!lip_read_map_size(file, &n)) return IMM_FAILURE;
    bool have_symbols = false;
    abc->any_symbol = 'X';
    abc->vtable.typeid = IMM_ABC;
    abc->vtable.derived = NULL;
    for (unsigned i = 0; i < n; ++i)
    {
        char key[32] = {0};
        if (!lip_read_cstr(file, sizeof key, key)) return IMM_FAILURE;
        if (!strcmp(key, "symbols"))
        {
            if (!lip_read_cstr(file, sizeof abc->symbols, abc->symbols)) return IMM_FAILURE;
            have_symbols = true;
        }
        else if (!strcmp(key, "any_symbol"))
        {
            unsigned v = 0;
            if (!lip_read_unsigned(file, &v) || v == 0 || v > 127) return IMM_FAILURE;
            abc->any_symbol = (char)v;
        }
        else if (!strcmp(key, "typeid"))
        {
            unsigned v = 0;
            if (!lip_read_unsigned(file, &v) || v > IMM_RNA) return IMM_FAILURE;
            abc->vtable.typeid = (enum imm_abc_typeid)v;
        }
        else if (!lip_skip_object(file))
            return IMM_FAILURE;
    }
    if (!have_symbols) return IMM_FAILURE;
    abc->size = (unsigned)strlen(abc->symbols);
    abc->any_symbol_id = abc->size;
    return abc->size ? IMM_OK : IMM_FAILURE;
}

enum imm_rc imm_nuclt_lprob_pack(struct imm_nuclt_lprob const *nucltp, struct lip_file *file)
{
    return lip_write_1darray_size_type(file, IMM_NUCLT_SIZE, LIP_1DARRAY_F32) &&
                   lip_write_1darray_f32_data(file, IMM_NUCLT_SIZE, nucltp->lprobs)
               ? IMM_OK
               : IMM_FAILURE;
}

static enum imm_rc read_f32_block(struct lip_file *file, unsigned want, float *out)
{
    unsigned n = 0;
    enum lip_1darray_type ty;
    if (!lip_read_1darray_size_type(file, &n, &ty) || ty != LIP_1DARRAY_F32 || n != want) return IMM_FAILURE;
    if (!lip_read_1darray_f32_data(file, n, out)) return IMM_FAILURE;
    for (unsigned i = 0; i < n; ++i)
        if (isnan(out[i])) return IMM_FAILURE;
    return IMM_OK;
}

enum imm_rc imm_nuclt_lprob_unpack(struct imm_nuclt_lprob *nucltp, struct lip_file *file)
{
    return read_f32_block(file, IMM_NUCLT_SIZE, nucltp->lprobs);
}

enum imm_rc imm_codon_marg_pack(struct imm_codon_marg const *codonm, struct lip_file *file)
{
    unsigned const n = sizeof codonm->lprobs / sizeof(imm_float);
    return lip_write_1darray_size_type(file, n, LIP_1DARRAY_F32) &&
                   lip_write_1darray_f32_data(file, n, &codonm->lprobs[0][0][0])
               ? IMM_OK
               : IMM_FAILURE;
}

enum imm_rc imm_codon_marg_unpack(struct imm_codon_marg *codonm, struct lip_file *file)
{
    return read_f32_block(file, sizeof codonm->lprobs / sizeof(imm_float), &codonm->lprobs[0][0][0]);
}

/* ---- standard_profile: typed shell (src/model/standard_profile.c, standard_state.c:5-10) ------------------ */
static void standard_del(struct profile *prof) { (void)prof; }
static struct imm_dp const *standard_null_dp(struct profile const *prof)
{
    return &((struct standard_profile const *)prof)->dp.null;
}
static struct imm_dp const *standard_alt_dp(struct profile const *prof)
{
    return &((struct standard_profile const *)prof)->dp.alt;
}
static enum rc standard_unpack(struct profile *prof, struct lip_file *file)
{
    return standard_profile_unpack((struct standard_profile *)prof, file);
}
unsigned standard_state_name(unsigned id, char name[IMM_STATE_NAME_SIZE])
{
    assert(id <= UINT16_MAX);
    return (unsigned)snprintf(name, IMM_STATE_NAME_SIZE, "S%u", id);
}
void standard_profile_init(struct standard_profile *p, char const *accession, struct imm_code const *code)
{
    struct profile_vtable const vtable = {PROFILE_STANDARD, standard_del, standard_unpack, standard_null_dp,
                                          standard_alt_dp};
    profile_init(&p->super, accession, code, vtable, standard_state_name);
    imm_dp_init(&p->dp.null, code);
    imm_dp_init(&p->dp.alt, code);
    p->dp.null.null_model = 1;
}
enum rc standard_profile_pack(struct standard_profile const *prof, struct lip_file *file)
{
    if (!lip_write_map_size(file, 2)) return fail(RC_EIO, "write map size");
    if (!lip_write_cstr(file, "null") || imm_dp_pack(&prof->dp.null, file)) return fail(RC_EFAIL, "write null dp");
    if (!lip_write_cstr(file, "alt") || imm_dp_pack(&prof->dp.alt, file)) return fail(RC_EFAIL, "write alt dp");
    return RC_OK;
}
enum rc standard_profile_unpack(struct standard_profile *prof, struct lip_file *file)
{
    if (!expect_map_size(file, 2)) return fail(RC_EIO, "read map size");
    if (!expect_map_key(file, "null") || imm_dp_unpack(&prof->dp.null, file)) return fail(RC_EFAIL, "read null dp");
    if (!expect_map_key(file, "alt") || imm_dp_unpack(&prof->dp.alt, file)) return fail(RC_EFAIL, "read alt dp");
    return RC_OK;
}

/* ---- protein_codec (src/model/protein_codec.c) ---------------------------------------------------------------- */
enum rc protein_codec_next(struct protein_codec *codec, struct imm_seq const *seq, struct imm_codon *codon)
{
    /* advance to the next emitting step; mute steps (S, B, D_k, E, T) carry no fragment */
    unsigned const n = imm_path_nsteps(codec->path);
    while (codec->idx < n && protein_state_is_mute(imm_path_step(codec->path, codec->idx)->state_id))
        codec->idx++;
    if (codec->idx >= n) return RC_END;
    struct imm_step const *step = imm_path_step(codec->path, codec->idx++);
    if (codec->start + step->seqlen > imm_seq_size(seq)) return fail(RC_EINVAL, "path does not fit the sequence");
    struct imm_seq frag = imm_subseq(seq, codec->start, step->seqlen);
    codec->start += step->seqlen;
    return protein_profile_decode(codec->prof, &frag, step->state_id, codon);
}

/* ---- protein_h3reader (src/model/protein_h3reader.c:18-72) over dcp_h3reader_* ------------------------------------ */
void protein_h3reader_init(struct protein_h3reader *reader, struct imm_amino const *amino,
                           struct imm_nuclt_code const *code, struct protein_cfg cfg, FILE *fp)
{
    reader->impl = dcp_h3reader_open_fp(fp, (int)cfg.entry_dist, cfg.epsilon);
    dcp_swissprot_null_lprobs(reader->null_lprobs); /* protein_h3reader.c:79-103 */
    protein_model_init(&reader->model, amino, code, cfg, reader->null_lprobs);
    reader->name[0] = reader->acc[0] = '\0';
}

enum rc protein_h3reader_next(struct protein_h3reader *reader)
{
    if (!reader->impl) return fail(RC_EFAIL, "reader is not open");
    struct dcp_h3params prm;
    int drc = dcp_h3reader_next_params(reader->impl, &prm);
    if (drc == DCP_END) return RC_END;
    if (drc == DCP_EINVAL) return fail(RC_EINVAL, "%s", dcp_h3reader_error(reader->impl));
    if (drc) return fail(RC_EFAIL, "%s", dcp_h3reader_error(reader->impl));
    enum rc rc = protein_model_setup(&reader->model, prm.core_size);
    if (rc) return rc;
    struct protein_trans t;
    memcpy(t.data, prm.trans, sizeof t.data);
    if ((rc = protein_model_add_trans(&reader->model, t))) return rc;
    for (unsigned k = 0; k < prm.core_size; ++k)
    {
        if ((rc = protein_model_add_node(&reader->model, prm.match_lprobs + 20 * (size_t)k, prm.consensus[k])))
            return rc;
        memcpy(t.data, prm.trans + 7 * ((size_t)k + 1), sizeof t.data);
        if ((rc = protein_model_add_trans(&reader->model, t))) return rc;
    }
    snprintf(reader->name, sizeof reader->name, "%s", prm.name);
    snprintf(reader->acc, sizeof reader->acc, "%s", prm.acc);
    return RC_OK;
}

void protein_h3reader_del(struct protein_h3reader const *reader)
{
    if (reader->impl) dcp_h3reader_close(reader->impl);
    protein_model_del(&reader->model);
}
