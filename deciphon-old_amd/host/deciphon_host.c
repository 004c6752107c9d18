/*
 * deciphon_host.c -- C11 host orchestration of the scan path over the HIP C-ABI.
 *
 * Implements include/deciphon_host.h: the reference's model/db/server entry points
 * for this path (names, argument meaning, return codes) with every score and path
 * computed on the device through dcp_gpu.h.  Reference files followed (behaviour,
 * not code): src/model/protein_profile.c:134-331, src/model/profile.c,
 * src/model/protein_state.c, src/model/protein_codec.c, src/db/profile_reader.c:54-168,
 * src/server/scan_thread.c:9-135, src/server/prod.c:13-41,153-181.
 */
#include "deciphon_host.h"

#include <assert.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

/* ---- logging: errors are reported where they are detected and returned -------------------- */
static enum rc fail(enum rc rc, char const *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    fputs("deciphon_host: ", stderr);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    return rc;
}

/* ---- core ----------------------------------------------------------------------------------- */
unsigned xmath_partition_size(unsigned nelems, unsigned nparts, unsigned idx)
{
    unsigned size = (nelems + nparts - 1) / nparts;
    assert(nelems >= size * idx);
    return size < nelems - size * idx ? size : nelems - size * idx;
}

float xmath_lrt_f32(float null_loglik, float alt_loglik) { return dcp_lrt(null_loglik, alt_loglik); }

/* ---- alphabets / sequences -------------------------------------------------------------------- */
struct imm_nuclt const imm_dna_iupac = {{IMM_DNA, "ACGT", 'X'}};
struct imm_amino const imm_amino_iupac = {{IMM_AMINO, "ACDEFGHIKLMNPQRSTVWY", 'X'}};

void imm_nuclt_code_init(struct imm_nuclt_code *code, struct imm_nuclt const *nuclt)
{
    code->abc = &nuclt->super;
    code->nuclt = nuclt;
}

unsigned imm_abc_any_symbol_id(struct imm_abc const *abc) { return (unsigned)strlen(abc->symbols); }

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

unsigned imm_seq_size(struct imm_seq const *seq) { return seq->size; }

struct imm_seq imm_subseq(struct imm_seq const *seq, unsigned start, unsigned size)
{
    assert(start + size <= seq->size);
    return (struct imm_seq){size, seq->str + start, seq->abc};
}

static unsigned symbol_id(struct imm_abc const *abc, char c)
{
    char const *p = strchr(abc->symbols, c);
    return p && c ? (unsigned)(p - abc->symbols) : imm_abc_any_symbol_id(abc);
}

struct imm_codon imm_codon(struct imm_nuclt const *nuclt, unsigned a, unsigned b, unsigned c)
{
    return (struct imm_codon){nuclt, a, b, c};
}

struct imm_codon imm_codon_from_symbols(struct imm_nuclt const *nuclt, char const sym[3])
{
    return imm_codon(nuclt, symbol_id(&nuclt->super, sym[0]), symbol_id(&nuclt->super, sym[1]),
                     symbol_id(&nuclt->super, sym[2]));
}

static char codon_sym(struct imm_codon const *codon, unsigned id)
{
    struct imm_abc const *abc = &codon->nuclt->super;
    return id < imm_abc_any_symbol_id(abc) ? abc->symbols[id] : abc->any_symbol;
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

/* ---- task / prod / path ------------------------------------------------------------------------ */
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

struct imm_prod imm_prod(void) { return (struct imm_prod){{NULL, 0, 0}, NAN}; }

void imm_prod_reset(struct imm_prod *prod)
{
    prod->path.nsteps = 0;
    prod->loglik = NAN;
}

void imm_prod_del(struct imm_prod const *prod) { free(prod->path.steps); }

unsigned imm_path_nsteps(struct imm_path const *path) { return path->nsteps; }

struct imm_step const *imm_path_step(struct imm_path const *path, unsigned idx)
{
    assert(idx < path->nsteps);
    return path->steps + idx;
}

bool imm_lprob_is_finite(imm_float x) { return isfinite(x); }

/* one shared device context for single-pair imm_dp_viterbi calls (library-level use, as the
 * reference's tests do); thread_run owns one context per partition instead */
static dcp_gpu_ctx *g_ctx;
static dcp_profile *g_ctx_db;

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

static enum rc path_assign(struct imm_path *path, struct dcp_step const *steps, unsigned n)
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

enum imm_rc imm_dp_viterbi(struct imm_dp const *dp, struct imm_task *task, struct imm_prod *prod)
{
    if (!dp || !task || !prod || !task->seq) return IMM_FAILURE;
    struct protein_profile *prof = dp->owner;
    struct imm_seq const *seq = task->seq;
    if (!prof || !prof->impl)
    {
        fail(RC_EINVAL, "profile holds no model (call protein_profile_sample / _from_params first)");
        return IMM_FAILURE;
    }
    if (prof->seq_size == 0 || prof->seq_size != seq->size)
    {
        /* imm keeps whatever transitions the last protein_profile_setup() wrote; the device path
         * derives them from the sequence length, so the two must agree */
        fail(RC_EINVAL, "protein_profile_setup(%u) does not match the sequence length %u", prof->seq_size,
             seq->size);
        return IMM_FAILURE;
    }
    dcp_gpu_ctx *ctx = shared_ctx();
    if (!ctx)
    {
        fail(RC_EFAIL, "no HIP device: imm_dp_viterbi has no CPU implementation here");
        return IMM_FAILURE;
    }
    if (g_ctx_db != prof->impl)
    {
        dcp_profile *one[1] = {prof->impl};
        if (dcp_gpu_db_upload(ctx, one, 1, 0)) return IMM_FAILURE;
        g_ctx_db = prof->impl;
    }
    uint32_t off[2] = {0, seq->size};
    if (dcp_gpu_seqs_upload_text(ctx, seq->str, off, 1)) return IMM_FAILURE;
    struct dcp_scan_params prm = {prof->multi_hits, prof->hmmer3_compat, 10.0f, 1, 1 /* row sweep */};
    if (dcp_gpu_scan(ctx, &prm) || dcp_gpu_sync(ctx)) return IMM_FAILURE;
    float nul = NAN, alt = NAN;
    if (dcp_gpu_fetch_scores(ctx, &nul, &alt)) return IMM_FAILURE;

    struct dcp_hit pair = {0, 0, nul, alt};
    unsigned cap = 2 * seq->size + 2 * prof->core_size + 16;
    struct dcp_step *steps = malloc((size_t)cap * sizeof *steps);
    if (!steps) return IMM_FAILURE;
    uint32_t soff[2] = {0, 0};
    float traced = NAN;
    int rc = dcp_gpu_trace_paths(ctx, &pair, 1, prof->multi_hits, prof->hmmer3_compat, dp->null_model, steps, cap,
                                 soff, &traced);
    enum imm_rc out = IMM_FAILURE;
    if (!rc && !path_assign(&prod->path, steps, soff[1]))
    {
        prod->loglik = dp->null_model ? nul : alt;
        out = traced == prod->loglik ? IMM_OK : IMM_FAILURE; /* the trace recomputes the score bit for bit */
    }
    free(steps);
    return out;
}

/* ---- protein state ----------------------------------------------------------------------------- */
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

/* ---- profile / protein_profile ------------------------------------------------------------------ */
struct protein_cfg protein_cfg(enum entry_dist entry_dist, imm_float epsilon)
{
    assert(epsilon >= 0.0f && epsilon <= 1.0f);
    return (struct protein_cfg){entry_dist, epsilon};
}

static void protein_del(struct profile *prof)
{
    if (!prof) return;
    struct protein_profile *p = (struct protein_profile *)prof;
    if (g_ctx_db == p->impl) g_ctx_db = NULL;
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

void profile_del(struct profile *prof)
{
    if (prof && prof->vtable.del) prof->vtable.del(prof);
}
int profile_typeid(struct profile const *prof) { return prof->vtable.typeid; }
struct imm_dp const *profile_null_dp(struct profile const *prof) { return prof->vtable.null_dp(prof); }
struct imm_dp const *profile_alt_dp(struct profile const *prof) { return prof->vtable.alt_dp(prof); }

/* ---- standard_profile: typed shell (src/model/standard_profile.c:45-51, standard_state.c:5-10) --- */
static void standard_del(struct profile *prof) { (void)prof; }
static struct imm_dp const *standard_null_dp(struct profile const *prof)
{
    return &((struct standard_profile const *)prof)->dp.null;
}
static struct imm_dp const *standard_alt_dp(struct profile const *prof)
{
    return &((struct standard_profile const *)prof)->dp.alt;
}
unsigned standard_state_name(unsigned id, char name[IMM_STATE_NAME_SIZE])
{
    assert(id <= UINT16_MAX);
    return (unsigned)snprintf(name, IMM_STATE_NAME_SIZE, "S%u", id);
}
void standard_profile_init(struct standard_profile *p, char const *accession, struct imm_nuclt_code const *code)
{
    memset(p, 0, sizeof *p);
    p->super.vtable = (struct profile_vtable){PROFILE_STANDARD, standard_del, standard_null_dp, standard_alt_dp};
    snprintf(p->super.accession, sizeof p->super.accession, "%s", accession ? accession : "");
    p->super.state_name = standard_state_name;
    p->super.code = code;
    p->dp.null = (struct imm_dp){NULL, 1};
    p->dp.alt = (struct imm_dp){NULL, 0};
}

void protein_profile_init(struct protein_profile *p, char const *accession, struct imm_amino const *amino,
                          struct imm_nuclt_code const *code, struct protein_cfg cfg)
{
    memset(p, 0, sizeof *p);
    p->super.vtable = (struct profile_vtable){PROFILE_PROTEIN, protein_del, protein_null_dp, protein_alt_dp};
    snprintf(p->super.accession, sizeof p->super.accession, "%s", accession ? accession : "");
    p->super.state_name = protein_state_name;
    p->super.code = code;
    p->amino = amino;
    p->code = code;
    p->cfg = cfg;
    p->null.dp = (struct imm_dp){p, 1};
    p->alt.dp = (struct imm_dp){p, 0};
    /* state indices as imm_state_idx reports them after the HMM -> DP compile: the null model has
     * the single state R; the alt model's specials were added first (protein_model.c:227-233) */
    p->null.R = 0;
    p->alt.S = 0, p->alt.N = 1, p->alt.B = 2, p->alt.E = 3, p->alt.J = 4, p->alt.C = 5, p->alt.T = 6;
}

enum rc protein_profile_setup(struct protein_profile *prof, unsigned seq_size, bool multi_hits, bool hmmer3_compat)
{
    float xt[DCP_NXTRANS];
    if (dcp_xtrans(seq_size, multi_hits, hmmer3_compat, xt)) return fail(RC_EINVAL, "sequence cannot be empty");
    prof->seq_size = seq_size;
    prof->multi_hits = multi_hits;
    prof->hmmer3_compat = hmmer3_compat;
    return RC_OK;
}

static enum rc adopt(struct protein_profile *p, dcp_profile *impl, int rc)
{
    if (!impl) return fail((enum rc)rc, "failed to build the profile");
    if (p->impl)
    {
        if (g_ctx_db == p->impl) g_ctx_db = NULL;
        dcp_profile_del(p->impl);
    }
    p->impl = impl;
    p->core_size = dcp_profile_core_size(impl);
    p->seq_size = 0;
    return RC_OK;
}

enum rc protein_profile_sample(struct protein_profile *p, unsigned seed, unsigned core_size)
{
    int rc = 0;
    dcp_profile *impl = dcp_profile_sample(p->super.accession, seed, core_size, (int)p->cfg.entry_dist,
                                           p->cfg.epsilon, &rc);
    return adopt(p, impl, rc);
}

enum rc protein_profile_from_params(struct protein_profile *p, unsigned core_size, imm_float const *null_lprobs,
                                    imm_float const *match_lprobs, imm_float const *trans)
{
    int rc = 0;
    dcp_profile *impl = dcp_profile_new(p->super.accession, core_size, (int)p->cfg.entry_dist, p->cfg.epsilon,
                                        null_lprobs, match_lprobs, trans, NULL, &rc);
    return adopt(p, impl, rc);
}

enum rc protein_profile_decode(struct protein_profile const *prof, struct imm_seq const *seq, unsigned state_id,
                               struct imm_codon *codon)
{
    assert(!protein_state_is_mute(state_id));
    uint8_t frag[5], out[3];
    if (seq->size < 1 || seq->size > 5) return fail(RC_EINVAL, "failed to decode sequence");
    for (unsigned i = 0; i < seq->size; ++i)
    {
        unsigned id = symbol_id(seq->abc, seq->str[i]);
        if (id > 3) return fail(RC_EINVAL, "failed to decode sequence");
        frag[i] = (uint8_t)id;
    }
    if (dcp_profile_decode(prof->impl, frag, seq->size, state_id, out))
        return fail(RC_EINVAL, "failed to decode sequence");
    codon->nuclt = prof->code->nuclt;
    codon->a = out[0], codon->b = out[1], codon->c = out[2];
    return RC_OK;
}

struct protein_codec protein_codec_init(struct protein_profile const *prof, struct imm_path const *path)
{
    return (struct protein_codec){0, 0, prof, path};
}

enum rc protein_codec_next(struct protein_codec *codec, struct imm_seq const *seq, struct imm_codon *codon)
{
    struct imm_step const *step = NULL;
    while (codec->idx < imm_path_nsteps(codec->path))
    {
        step = imm_path_step(codec->path, codec->idx);
        if (!protein_state_is_mute(step->state_id)) break;
        codec->idx++;
    }
    if (codec->idx >= imm_path_nsteps(codec->path)) return RC_END;
    if (codec->start + step->seqlen > imm_seq_size(seq)) return fail(RC_EINVAL, "path does not fit the sequence");
    struct imm_seq frag = imm_subseq(seq, codec->start, step->seqlen);
    codec->start += step->seqlen;
    codec->idx++;
    return protein_profile_decode(codec->prof, &frag, step->state_id, codon);
}

/* ---- profile_reader: count-balanced contiguous partitions (profile_reader.c:54-72) ----------------- */
enum rc profile_reader_setup(struct profile_reader *reader, struct protein_db const *db, unsigned npartitions)
{
    if (npartitions == 0) return fail(RC_EINVAL, "can't have zero partitions");
    if (npartitions > NUM_THREADS) return fail(RC_EINVAL, "too many partitions");
    memset(reader, 0, sizeof *reader);
    reader->db = db;
    unsigned sizes[DCP_NUM_THREADS];
    reader->npartitions = dcp_partition_by_count(db->nprofiles, npartitions, sizes);
    for (unsigned i = 0; i < reader->npartitions; ++i)
    {
        reader->partition_size[i] = sizes[i];
        reader->partition_begin[i + 1] = reader->partition_begin[i] + sizes[i];
    }
    return profile_reader_rewind_all(reader);
}

unsigned profile_reader_npartitions(struct profile_reader const *reader) { return reader->npartitions; }
unsigned profile_reader_partition_size(struct profile_reader const *reader, unsigned partition)
{
    return reader->partition_size[partition];
}
unsigned profile_reader_nprofiles(struct profile_reader const *reader)
{
    unsigned n = 0;
    for (unsigned i = 0; i < reader->npartitions; ++i)
        n += reader->partition_size[i];
    return n;
}
enum rc profile_reader_rewind_all(struct profile_reader *reader)
{
    for (unsigned i = 0; i < reader->npartitions; ++i)
        reader->cursor[i] = reader->partition_begin[i];
    return RC_OK;
}
enum rc profile_reader_rewind(struct profile_reader *reader, unsigned partition)
{
    reader->cursor[partition] = reader->partition_begin[partition];
    return RC_OK;
}
enum rc profile_reader_next(struct profile_reader *reader, unsigned partition, struct profile **profile)
{
    if (reader->cursor[partition] == reader->partition_begin[partition + 1]) return RC_END;
    *profile = &reader->db->profiles[reader->cursor[partition]++]->super; /* borrowed */
    return RC_OK;
}

/* ---- scan thread ------------------------------------------------------------------------------------ */
void thread_init(struct scan_thread *t, unsigned id, struct profile_reader *reader, bool multi_hits,
                 bool hmmer3_compat, double lrt_threshold)
{
    memset(t, 0, sizeof *t);
    t->id = id;
    t->reader = reader;
    t->multi_hits = multi_hits;
    t->hmmer3_compat = hmmer3_compat;
    t->lrt_threshold = lrt_threshold;
}

void thread_setup_job(struct scan_thread *t, enum imm_abc_typeid abc_typeid, enum profile_typeid typeid,
                      int64_t scan_id)
{
    snprintf(t->prod.abc_name, sizeof t->prod.abc_name, "%s", imm_abc_typeid_name(abc_typeid));
    snprintf(t->prod.profile_typeid, sizeof t->prod.profile_typeid, "%s", profile_typeid_name(typeid));
    snprintf(t->prod.version, sizeof t->prod.version, "%s", "0.1.0");
    t->prod.scan_id = scan_id;
}

void thread_setup_seq(struct scan_thread *t, struct imm_seq *seq, int64_t seq_id)
{
    t->seq = seq;
    t->prod.seq_id = seq_id;
}

void thread_cleanup(struct scan_thread *t)
{
    if (t->gpu) dcp_gpu_ctx_del(t->gpu);
    free(t->rows);
    t->gpu = NULL;
    t->rows = NULL;
    t->rows_len = t->rows_cap = 0;
    t->db_resident = false;
}

static enum rc rows_reserve(struct scan_thread *t, size_t extra)
{
    if (t->rows_len + extra + 1 <= t->rows_cap) return RC_OK;
    size_t cap = t->rows_cap ? t->rows_cap * 2 : 4096;
    while (cap < t->rows_len + extra + 1)
        cap *= 2;
    char *p = realloc(t->rows, cap);
    if (!p) return fail(RC_ENOMEM, "alloc product rows");
    t->rows = p;
    t->rows_cap = cap;
    return RC_OK;
}

static enum rc thread_prepare(struct scan_thread *t, int tid)
{
    struct profile_reader *reader = t->reader;
    unsigned const n = reader->partition_size[t->id];
    if (!t->gpu)
    {
        int ndev = dcp_gpu_device_count();
        if (ndev <= 0) return fail(RC_EFAIL, "no HIP device: thread_run has no CPU implementation here");
        t->gpu = dcp_gpu_ctx_new(tid % ndev);
        if (!t->gpu) return fail(RC_EFAIL, "failed to create the device context");
    }
    /* the partition's profiles are uploaded once and stay resident between sequences
     * (the reference re-reads and re-unpacks them for every sequence: scan_thread.c:96-99) */
    if (!t->db_resident)
    {
        dcp_profile **impls = malloc((size_t)n * sizeof *impls);
        if (!impls) return fail(RC_ENOMEM, "alloc");
        enum rc rc = profile_reader_rewind(reader, t->id);
        struct profile *prof = NULL;
        unsigned i = 0;
        while (!rc && (rc = profile_reader_next(reader, t->id, &prof)) == RC_OK)
            impls[i++] = ((struct protein_profile *)prof)->impl;
        if (rc == RC_END) rc = RC_OK;
        if (!rc && dcp_gpu_db_upload(t->gpu, impls, n, 0)) rc = fail(RC_EFAIL, "%s", dcp_gpu_last_error(t->gpu));
        free(impls);
        if (rc) return rc;
        t->db_resident = true;
    }
    return RC_OK;
}

enum rc thread_run_batch(struct scan_thread *t, int tid, struct imm_seq const *seqs, int64_t const *seq_ids,
                         unsigned nseqs)
{
    struct profile_reader *reader = t->reader;
    if (!reader || !seqs || nseqs == 0) return fail(RC_EINVAL, "thread has no reader or sequence");
    unsigned const first = reader->partition_begin[t->id];
    unsigned const n = reader->partition_size[t->id];
    if (n == 0) return RC_OK;
    enum rc rc = thread_prepare(t, tid);
    if (rc) return rc;

    /* protein_profile_setup(pp, size, ...) for every profile rejects the empty sequence (:112) */
    size_t total = 0;
    for (unsigned q = 0; q < nseqs; ++q)
    {
        if (seqs[q].size == 0) return fail(RC_EINVAL, "sequence cannot be empty");
        total += seqs[q].size;
    }
    /* the caller's buffers may be overwritten by the next fetch (scan.c:227-229): copy now */
    char *text = malloc(total + 1);
    uint32_t *off = malloc(((size_t)nseqs + 1) * sizeof *off);
    if (!text || !off)
    {
        free(text), free(off);
        return fail(RC_ENOMEM, "alloc sequence batch");
    }
    off[0] = 0;
    for (unsigned q = 0; q < nseqs; ++q)
    {
        memcpy(text + off[q], seqs[q].str, seqs[q].size);
        off[q + 1] = off[q] + seqs[q].size;
    }
    int drc = dcp_gpu_seqs_upload_text(t->gpu, text, off, nseqs);
    if (drc) rc = fail((enum rc)drc, "%s", dcp_gpu_last_error(t->gpu));
    struct dcp_scan_params prm = {t->multi_hits, t->hmmer3_compat, (float)t->lrt_threshold, 0, 0};
    if (!rc && ((drc = dcp_gpu_scan(t->gpu, &prm)) || (drc = dcp_gpu_sync(t->gpu))))
        rc = fail((enum rc)drc, "failed to run viterbi: %s", dcp_gpu_last_error(t->gpu));

    /* lrt filter ran on the device (scan_thread.c:121-123): only hits come back, sorted by (seq, profile) */
    unsigned nhits = 0;
    size_t hit_cap = (size_t)n * nseqs;
    struct dcp_hit *hits = NULL;
    struct dcp_step *steps = NULL;
    uint32_t *soff = NULL;
    uint8_t *ids = NULL;
    if (!rc)
    {
        drc = dcp_gpu_fetch_hits(t->gpu, NULL, 0, &nhits); /* count first */
        if (drc && drc != DCP_ENOMEM) rc = fail((enum rc)drc, "fetch hits");
        if (!rc && nhits > hit_cap) rc = fail(RC_EFAIL, "more hits than pairs");
    }
    if (!rc && nhits)
    {
        hits = malloc((size_t)nhits * sizeof *hits);
        if (!hits) rc = fail(RC_ENOMEM, "alloc hits");
        if (!rc && (drc = dcp_gpu_fetch_hits(t->gpu, hits, nhits, &nhits))) rc = fail((enum rc)drc, "fetch hits");
        size_t cap = 0;
        for (unsigned h = 0; !rc && h < nhits; ++h)
            cap += 2 * (size_t)seqs[hits[h].seq_idx].size +
                   2 * (size_t)reader->db->profiles[first + hits[h].profile_idx]->core_size + 16;
        if (!rc)
        {
            steps = malloc(cap * sizeof *steps);
            soff = malloc(((size_t)nhits + 1) * sizeof *soff);
            ids = malloc(total);
            if (!steps || !soff || !ids) rc = fail(RC_ENOMEM, "alloc paths");
        }
        if (!rc && (drc = dcp_gpu_trace_paths(t->gpu, hits, nhits, t->multi_hits, t->hmmer3_compat, 0, steps,
                                              (unsigned)cap, soff, NULL)))
            rc = fail((enum rc)drc, "%s", dcp_gpu_last_error(t->gpu));
        for (size_t i = 0; !rc && i < total; ++i)
            ids[i] = (uint8_t)symbol_id(seqs[0].abc, text[i]);
        for (unsigned h = 0; !rc && h < nhits; ++h)
        {
            unsigned const q = hits[h].seq_idx;
            struct protein_profile const *pp = reader->db->profiles[first + hits[h].profile_idx];
            /* strcpy(t->prod.profile_name, prof->accession); match_setup; write_product (:125-128) */
            snprintf(t->prod.profile_name, sizeof t->prod.profile_name, "%s", pp->super.accession);
            t->prod.seq_id = seq_ids ? seq_ids[q] : t->prod.seq_id;
            t->prod.null_loglik = (double)hits[h].null_loglik;
            t->prod.alt_loglik = (double)hits[h].alt_loglik;
            unsigned ns = soff[h + 1] - soff[h];
            size_t need = 512 + 64 * ((size_t)ns + 1) + 2 * (size_t)seqs[q].size;
            if ((rc = rows_reserve(t, need))) break;
            long w = dcp_prod_format_row(t->rows + t->rows_len, t->rows_cap - t->rows_len, t->prod.scan_id,
                                         t->prod.seq_id, t->prod.profile_name, t->prod.abc_name, t->prod.alt_loglik,
                                         t->prod.null_loglik, t->prod.profile_typeid, t->prod.version, pp->impl,
                                         ids + off[q], seqs[q].size, steps + soff[h], ns);
            if (w < 0)
            {
                rc = fail(RC_EIO, "failed to write prod");
                break;
            }
            t->rows_len += (size_t)w;
            t->nprods++;
        }
    }
    free(ids);
    free(soff);
    free(steps);
    free(hits);
    free(off);
    free(text);
    return rc;
}

enum rc thread_run(struct scan_thread *t, int tid)
{
    if (!t->reader || !t->seq) return fail(RC_EINVAL, "thread has no reader or sequence");
    int64_t id = t->prod.seq_id;
    return thread_run_batch(t, tid, t->seq, &id, 1);
}

enum rc scan_run_local(struct protein_db const *db, struct scan_seq const *seqs, unsigned nseqs,
                       unsigned num_threads, bool multi_hits, bool hmmer3_compat, double lrt_threshold,
                       int64_t scan_id, unsigned batch, FILE *prods)
{
    if (!db || !seqs || !prods || batch == 0) return fail(RC_EINVAL, "bad scan arguments");
    struct profile_reader reader;
    enum rc rc = profile_reader_setup(&reader, db, num_threads); /* prepare_readers: scan.c:45-74 */
    if (rc) return rc;
    unsigned const nparts = profile_reader_npartitions(&reader);
    struct scan_thread *th = calloc(nparts, sizeof *th);
    struct imm_seq *bseq = malloc((size_t)batch * sizeof *bseq);
    int64_t *bid = malloc((size_t)batch * sizeof *bid);
    if (!th || !bseq || !bid)
    {
        free(th), free(bseq), free(bid);
        return fail(RC_ENOMEM, "alloc scan");
    }
    for (unsigned i = 0; i < nparts; ++i)
    {
        thread_init(&th[i], i, &reader, multi_hits, hmmer3_compat, lrt_threshold);
        thread_setup_job(&th[i], IMM_DNA, PROFILE_PROTEIN, scan_id);
    }
    for (unsigned b0 = 0; b0 < nseqs && !rc; b0 += batch)
    {
        unsigned const nb = nseqs - b0 < batch ? nseqs - b0 : batch;
        for (unsigned q = 0; q < nb; ++q)
        {
            bseq[q] = imm_seq(imm_str(seqs[b0 + q].data), &imm_dna_iupac.super);
            bid[q] = seqs[b0 + q].id;
        }
        enum rc shared = RC_OK;
#pragma omp parallel for schedule(static, 1)
        for (unsigned i = 0; i < nparts; ++i)
        {
            enum rc r = thread_run_batch(&th[i], (int)i, bseq, bid, nb);
            if (r)
            {
#pragma omp atomic write
                shared = r; /* scan.c:246-248: first failing partition fails the scan */
            }
        }
        rc = shared;
    }
    if (!rc)
    {
        /* prod_fclose: header, then every thread's rows in thread order (prod.c:119-134) */
        if (fputs(prod_header(), prods) < 0) rc = fail(RC_EIO, "fail to finish product");
        for (unsigned i = 0; !rc && i < nparts; ++i)
            if (th[i].rows_len && fwrite(th[i].rows, 1, th[i].rows_len, prods) != th[i].rows_len)
                rc = fail(RC_EIO, "fail to finish product");
    }
    for (unsigned i = 0; i < nparts; ++i)
        thread_cleanup(&th[i]);
    free(th), free(bseq), free(bid);
    return rc;
}

char const *prod_header(void) { return dcp_prod_header(); }
