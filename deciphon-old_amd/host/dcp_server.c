/*
 * dcp_server.c -- C11 host layer, part 3: products, scan threads and the scan loop
 * (include/deciphon_host.h section "server").
 *
 * Reference files followed (behaviour, not code): src/server/prod.c, src/server/protein_match.c:21-56,
 * src/server/scan_thread.c:9-135, src/server/scan.c:45-74,215-269.  One scan thread = one database
 * partition = one device context; the partition's profiles are unpacked and uploaded ONCE and stay
 * resident, where the reference re-reads and re-unpacks them for every sequence.
 */
#include "deciphon_host.h"
#include "host_internal.h"

#include <omp.h>

#include <inttypes.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

/* scan_last_stats(): where the last scan_run_source spent its host time */
static struct scan_stats g_stats;
static double stat_now(void) { return omp_get_wtime(); }
static void stat_add(double *field, double dt)
{
#pragma omp atomic
    *field += dt;
}
void scan_last_stats(struct scan_stats *out)
{
    if (out) *out = g_stats;
}

#define fail dcp_host_fail

/* ============================== products (src/server/prod.c) ================================== */
struct tmp_file
{
    FILE *fp;
    char path[64];
};
static unsigned num_threads;
static struct tmp_file prod_file[NUM_THREADS];
static struct tmp_file final_file;

static enum rc tmp_open(struct tmp_file *t)
{
    snprintf(t->path, sizeof t->path, "%s/dcp_prod_XXXXXX", P_tmpdir);
    int fd = mkstemp(t->path);
    if (fd < 0) return fail(RC_EIO, "mkstemp");
    t->fp = fdopen(fd, "wb+");
    if (!t->fp)
    {
        close(fd);
        unlink(t->path);
        return fail(RC_EIO, "fdopen");
    }
    return RC_OK;
}

static void tmp_del(struct tmp_file *t)
{
    if (t->fp)
    {
        fclose(t->fp);
        unlink(t->path);
    }
    t->fp = NULL;
    t->path[0] = '\0';
}

enum rc prod_fopen(unsigned nthreads)
{
    assert(nthreads <= NUM_THREADS);
    prod_fcleanup();
    for (num_threads = 0; num_threads < nthreads; ++num_threads)
        if (tmp_open(prod_file + num_threads))
        {
            prod_fcleanup();
            return fail(RC_EFAIL, "begin prod submission");
        }
    return RC_OK;
}

void prod_setup_job(struct prod *prod, char const *abc_name, char const *prof_typeid, int64_t scan_id)
{
    snprintf(prod->abc_name, sizeof prod->abc_name, "%s", abc_name);
    snprintf(prod->profile_typeid, sizeof prod->profile_typeid, "%s", prof_typeid);
    snprintf(prod->version, sizeof prod->version, "%s", DECIPHON_VERSION);
    prod->scan_id = scan_id;
}

void prod_setup_seq(struct prod *prod, int64_t seq_id) { prod->seq_id = seq_id; }

void prod_fcleanup(void)
{
    for (unsigned i = 0; i < NUM_THREADS; ++i)
        tmp_del(prod_file + i);
    num_threads = 0;
}

char const *prod_header(void) { return dcp_prod_header(); }

enum rc prod_fclose(void)
{
    enum rc rc = RC_OK;
    tmp_del(&final_file);
    if (tmp_open(&final_file))
    {
        rc = fail(RC_EIO, "fail to finish product");
        goto cleanup;
    }
    fputs(prod_header(), final_file.fp);
    for (unsigned i = 0; i < num_threads && !rc; ++i)
    {
        char buf[1 << 16];
        size_t n;
        if (!prod_file[i].fp) continue; /* a thread that was never opened (on-demand use) wrote nothing */
        if (fflush(prod_file[i].fp)) rc = fail(RC_EIO, "failed to flush");
        rewind(prod_file[i].fp);
        while (!rc && (n = fread(buf, 1, sizeof buf, prod_file[i].fp)) > 0)
            if (fwrite(buf, 1, n, final_file.fp) != n) rc = fail(RC_EIO, "failed to copy products");
    }
    if (!rc && fflush(final_file.fp)) rc = fail(RC_EIO, "failed to flush");
    if (rc) tmp_del(&final_file);
    else rewind(final_file.fp);

cleanup:
    prod_fcleanup();
    return rc;
}

FILE *prod_final_fp(void) { return final_file.fp; }
char const *prod_final_path(void) { return final_file.path; }
void prod_final_cleanup(void) { tmp_del(&final_file); }

/* One products row: eight tab-separated fields, then the match column -- one "frag,state,codon,amino"
 * item per path step, ';'-separated, each written by the caller's fwrite_match (prod.c:13-41,153-181). */
static enum rc prod_thread_fp(unsigned thread_num, FILE **fp)
{
    if (thread_num >= NUM_THREADS) return fail(RC_EINVAL, "thread number out of range");
    if (!prod_file[thread_num].fp)
    {
        /* thread_run driven without scan_run's prod_fopen (library-level use): open on demand */
        if (tmp_open(prod_file + thread_num)) return fail(RC_EIO, "failed to write prod");
        if (num_threads <= thread_num) num_threads = thread_num + 1;
    }
    *fp = prod_file[thread_num].fp;
    return RC_OK;
}

static enum rc prod_fwrite_fp(FILE *fp, struct prod const *prod, struct imm_seq const *seq, struct imm_path const *path,
                             prod_fwrite_match_func_t fwrite_match, struct match *match);

enum rc prod_fwrite(struct prod const *prod, struct imm_seq const *seq, struct imm_path const *path,
                    unsigned thread_num, prod_fwrite_match_func_t fwrite_match, struct match *match)
{
    FILE *fp = NULL;
    enum rc rc = prod_thread_fp(thread_num, &fp);
    return rc ? rc : prod_fwrite_fp(fp, prod, seq, path, fwrite_match, match);
}

static enum rc prod_fwrite_fp(FILE *fp, struct prod const *prod, struct imm_seq const *seq, struct imm_path const *path,
                             prod_fwrite_match_func_t fwrite_match, struct match *match)
{
    if (fprintf(fp, "%" PRId64 "\t%" PRId64 "\t%s\t%s\t%.17g\t%.17g\t%s\t%s\t", prod->scan_id, prod->seq_id,
                prod->profile_name, prod->abc_name, prod->alt_loglik, prod->null_loglik, prod->profile_typeid,
                prod->version) < 0)
        return fail(RC_EIO, "failed to write prod");
    unsigned start = 0;
    unsigned const n = imm_path_nsteps(path);
    for (unsigned idx = 0; idx < n; idx++)
    {
        match->step = imm_path_step(path, idx);
        if (start + match->step->seqlen > seq->size) return fail(RC_EINVAL, "path does not fit the sequence");
        struct imm_seq frag = imm_subseq(seq, start, match->step->seqlen);
        match->frag = &frag;
        if (idx > 0 && fputc(';', fp) == EOF) return fail(RC_EIO, "failed to write prod");
        if (fwrite_match(fp, match)) return fail(RC_EIO, "write prod");
        start += match->step->seqlen;
    }
    match->frag = NULL;
    if (fputc('\n', fp) == EOF) return fail(RC_EIO, "failed to write prod");
    return RC_OK;
}

/* src/server/protein_match.c:21-56 */
enum rc protein_match_write_func(FILE *fp, void const *match)
{
    struct match const *m = match;
    struct protein_profile const *prof = (struct protein_profile const *)m->profile;
    struct imm_step const *step = m->step;
    struct imm_seq const *f = m->frag;
    struct imm_codon codon = imm_codon_any(prof->code->nuclt);
    char state[IMM_STATE_NAME_SIZE] = {0};
    m->profile->state_name(step->state_id, state);
    char ccodon[4] = {0}, camino[2] = {0};
    if (!protein_state_is_mute(step->state_id))
    {
        if (protein_profile_decode(prof, f, step->state_id, &codon)) return fail(RC_EIO, "failed to write match");
        ccodon[0] = imm_codon_asym(&codon);
        ccodon[1] = imm_codon_bsym(&codon);
        ccodon[2] = imm_codon_csym(&codon);
        camino[0] = imm_gc_decode(1, codon);
    }
    if (fprintf(fp, "%.*s,%s,%s,%s", (int)f->size, f->str, state, ccodon, camino) < 0)
        return fail(RC_EIO, "failed to write match");
    return RC_OK;
}

/* ============================== scan thread (src/server/scan_thread.c) ========================= */
void thread_init(struct scan_thread *t, unsigned id, struct profile_reader *reader, bool multi_hits,
                 bool hmmer3_compat, double lrt_threshold, prod_fwrite_match_func_t write_match_func)
{
    memset(t, 0, sizeof *t);
    t->id = id;
    t->reader = reader;
    t->multi_hits = multi_hits;
    t->hmmer3_compat = hmmer3_compat;
    t->lrt_threshold = lrt_threshold;
    t->write_match_func = write_match_func;
    t->null.prod = imm_prod();
    t->alt.prod = imm_prod();
}

void thread_setup_job(struct scan_thread *t, enum imm_abc_typeid abc_typeid, enum profile_typeid typeid,
                      int64_t scan_id)
{
    prod_setup_job(&t->prod, imm_abc_typeid_name(abc_typeid), profile_typeid_name(typeid), scan_id);
}

void thread_setup_seq(struct scan_thread *t, struct imm_seq *seq, int64_t seq_id)
{
    t->seq = seq;
    prod_setup_seq(&t->prod, seq_id);
}

void thread_cleanup(struct scan_thread *t)
{
    if (t->gpu) dcp_gpu_ctx_del(t->gpu);
    for (unsigned i = 0; i < t->nimpls; ++i)
        dcp_profile_del(t->impls[i]);
    free(t->impls);
    imm_prod_del(&t->null.prod);
    imm_prod_del(&t->alt.prod);
    imm_task_del(t->null.task);
    imm_task_del(t->alt.task);
    t->gpu = NULL;
    t->impls = NULL;
    t->nimpls = 0;
    t->null.task = t->alt.task = NULL;
    t->null.prod = t->alt.prod = imm_prod();
    t->db_resident = false;
}

/* Unpack the partition once (the reference does this per sequence: scan_thread.c:96-99), keep the
 * compact profiles on the host for decoding and upload them to the thread's device context. */
static enum rc thread_prepare(struct scan_thread *t, int tid)
{
    struct profile_reader *reader = t->reader;
    unsigned const n = reader->partition_size[t->id];
    if (!t->gpu)
    {
        int ndev = dcp_gpu_device_count();
        if (ndev <= 0) return fail(RC_EFAIL, "no HIP device: thread_run has no CPU implementation here");
        t->gpu = dcp_gpu_ctx_new((tid < 0 ? 0 : tid) % ndev);
        if (!t->gpu) return fail(RC_EFAIL, "failed to create the device context");
    }
    if (t->db_resident) return RC_OK;
    /* a previous attempt that failed half way left profiles behind: start over */
    for (unsigned i = 0; i < t->nimpls; ++i)
        dcp_profile_del(t->impls[i]);
    free(t->impls);
    t->nimpls = 0;
    t->impls = calloc(n ? n : 1, sizeof *t->impls);
    if (!t->impls) return fail(RC_ENOMEM, "alloc");
    /* The partition is unpacked by several host threads at once: profile_sizes[] gives every profile's byte
     * offset, each helper reads its own contiguous share through a FILE* of its own into a profile object of
     * its own (the reference parses one profile at a time, per sequence: profile_reader_next inside
     * thread_run).  A Pfam-sized partition is ~2 GB of MessagePack: 1.15 s on one core. */
    enum rc rc = RC_OK;
    int helpers = omp_get_num_procs() / (int)(reader->npartitions ? reader->npartitions : 1);
    if (helpers > 16) helpers = 16;
    if ((unsigned)helpers > n / 32u + 1u) helpers = (int)(n / 32u + 1u);
    if (helpers < 1 || !reader->profile_sizes) helpers = 1;
    unsigned const first = reader->partition_first[t->id];
    int const levels = omp_get_max_active_levels();
    if (helpers > 1 && levels < 2) omp_set_max_active_levels(2); /* this runs inside scan_run_source's team */
#pragma omp parallel num_threads(helpers) if (helpers > 1)
    {
        unsigned const me = (unsigned)omp_get_thread_num(), team = (unsigned)omp_get_num_threads();
        unsigned const lo = (unsigned)((uint64_t)n * me / team), hi = (unsigned)((uint64_t)n * (me + 1u) / team);
        enum rc r = RC_OK;
        FILE *fp = lo < hi ? dcp_host_reopen(reader->file[t->id].fp) : NULL;
        if (lo < hi && !fp) r = fail(RC_EIO, "failed to open file");
        if (fp)
        {
            int64_t at = reader->partition_offset[t->id];
            for (unsigned j = 0; j < lo; ++j)
                at += reader->profile_sizes ? reader->profile_sizes[first + j] : 0;
            if (!reader->profile_sizes && lo) r = fail(RC_EFAIL, "no profile sizes");
            if (!r && fseek(fp, (long)at, SEEK_SET)) r = fail(RC_EIO, "failed to fseek");
            struct lip_file file;
            lip_file_init(&file, fp);
            struct protein_profile const *tmpl = &reader->profiles[t->id].pro;
            struct protein_profile one;
            protein_profile_init(&one, "", tmpl->amino, tmpl->code, tmpl->cfg);
            for (unsigned j = lo; j < hi && !r; ++j)
            {
                r = profile_unpack(&one.super, &file);
                if (r) break;
                /* the profile object is reused for the next one: move its compact form out */
                t->impls[j] = one.impl;
                dcp_host_forget_profile(one.impl);
                one.impl = NULL;
            }
            /* every helper must end exactly where the next one began (and the last at the partition's end) */
            if (!r && reader->profile_sizes)
            {
                int64_t want = at;
                for (unsigned j = lo; j < hi; ++j)
                    want += reader->profile_sizes[first + j];
                if (ftell(fp) != (long)want) r = fail(RC_EPARSE, "partition %u: profiles %u..%u do not fill their bytes", t->id, lo, hi);
            }
            profile_del(&one.super);
            fclose(fp);
        }
        if (r)
        {
#pragma omp critical(dcp_thread_prepare)
            rc = rc ? rc : r;
        }
    }
    if (helpers > 1 && levels < 2) omp_set_max_active_levels(levels);
    t->nimpls = n; /* thread_cleanup frees what was unpacked, also after an error (missing ones are NULL) */
    if (!rc)
    {
        int64_t end = reader->partition_offset[t->id];
        for (unsigned j = 0; j < n && reader->profile_sizes; ++j)
            end += reader->profile_sizes[first + j];
        if (reader->profile_sizes && end != reader->partition_offset[t->id + 1])
            rc = fail(RC_EPARSE, "partition %u: %u profiles do not end at the partition's end offset", t->id, n);
    }
    if (!rc && dcp_gpu_db_upload(t->gpu, t->impls, n, 0)) rc = fail(RC_EFAIL, "%s", dcp_gpu_last_error(t->gpu));
    if (rc) return rc;
    t->db_resident = true;
    return RC_OK;
}

/* a borrowed view of resident profile i, complete enough for write_match_func callbacks */
static enum rc profile_view(struct scan_thread *t, unsigned i, struct protein_profile *view)
{
    struct protein_profile const *tmpl = &t->reader->profiles[t->id].pro;
    dcp_profile *impl = t->impls[i];
    protein_profile_init(view, dcp_profile_accession(impl), tmpl->amino, tmpl->code, tmpl->cfg);
    view->impl = impl; /* borrowed: never profile_del() a view */
    view->core_size = dcp_profile_core_size(impl);
    memcpy(view->consensus, dcp_profile_consensus(impl), (size_t)view->core_size + 1);
    memcpy(view->null.ndist.nucltp.lprobs, dcp_profile_null_dist(impl), sizeof view->null.ndist.nucltp.lprobs);
    memcpy(view->null.ndist.codonm.lprobs, dcp_profile_null_dist(impl) + IMM_NUCLT_SIZE,
           sizeof view->null.ndist.codonm.lprobs);
    memcpy(view->alt.insert_ndist.nucltp.lprobs, dcp_profile_insert_dist(impl), sizeof view->alt.insert_ndist.nucltp.lprobs);
    memcpy(view->alt.insert_ndist.codonm.lprobs, dcp_profile_insert_dist(impl) + IMM_NUCLT_SIZE,
           sizeof view->alt.insert_ndist.codonm.lprobs);
    view->alt.match_ndists = malloc((size_t)view->core_size * sizeof *view->alt.match_ndists);
    if (!view->alt.match_ndists) return fail(RC_ENOMEM, "alloc nuclt dists");
    float const *md = dcp_profile_match_dist(impl);
    for (unsigned k = 0; k < view->core_size; ++k)
    {
        struct nuclt_dist *d = view->alt.match_ndists + k;
        nuclt_dist_init(d, view->code->nuclt);
        memcpy(d->nucltp.lprobs, md + (size_t)k * DCP_NDIST, sizeof d->nucltp.lprobs);
        memcpy(d->codonm.lprobs, md + (size_t)k * DCP_NDIST + IMM_NUCLT_SIZE, sizeof d->codonm.lprobs);
    }
    return RC_OK;
}

/* A batch goes through three phases, so that scan_run_source can run one batch's row formatting (host) under
 * the next batch's scan (device):
 *   batch_submit  unpack + upload the partition if it is not resident, encode and upload the sequences,
 *                 enqueue the scan                                                   (returns at once)
 *   batch_trace   wait for the scan, fetch the hit list, recover the hits' paths on the device
 *   batch_rows    one product row per hit (decode every step against its state's codon distribution), appended
 *                 to the partition's product file in (sequence, profile) order
 * batch_trace must precede the next batch_submit on the same context (the traceback reads the resident
 * sequences); batch_rows only needs the host copies. */
struct batch_result
{
    unsigned nhits;
    struct dcp_hit *hits;
    struct dcp_step *steps;
    uint32_t *soff;
};

static void batch_result_free(struct batch_result *r)
{
    free(r->hits), free(r->steps), free(r->soff);
    memset(r, 0, sizeof *r);
}

static enum rc batch_check(struct scan_thread *t, struct imm_seq const *seqs, unsigned nseqs, unsigned *nprofiles)
{
    struct profile_reader *reader = t->reader;
    if (!reader || !seqs || nseqs == 0) return fail(RC_EINVAL, "thread has no reader or sequence");
    if (t->id >= reader->npartitions) return fail(RC_EINVAL, "thread %u has no partition", t->id);
    *nprofiles = reader->partition_size[t->id];
    if (*nprofiles && !t->write_match_func) return fail(RC_EINVAL, "thread has no write_match_func");
    return RC_OK;
}

static enum rc batch_submit(struct scan_thread *t, int tid, struct imm_seq const *seqs, unsigned nseqs)
{
    double const t_in = stat_now();
    enum rc rc = thread_prepare(t, tid);
    double const t_loaded = stat_now();
    stat_add(&g_stats.load_s, t_loaded - t_in);
    if (rc) return rc;
    /* protein_profile_setup(pp, size, ...) rejects the empty sequence for every profile (:112) */
    size_t total = 0;
    for (unsigned q = 0; q < nseqs; ++q)
    {
        if (seqs[q].size == 0) return fail(RC_EINVAL, "sequence cannot be empty");
        total += seqs[q].size;
    }
    /* the batch's offsets are 32-bit (dcp_gpu_seqs_upload): a batch of 4 Gi symbols or more would wrap them */
    if (total > UINT32_MAX) return fail(RC_EINVAL, "sequence batch exceeds 2^32 - 1 symbols: scan it in smaller batches");
    /* encode once per batch (imm_task_setup does it per pair: scan_thread.c:51-55); the caller's
     * buffers may be overwritten by its next fetch (scan.c:227-229), nothing here keeps them */
    uint8_t *ids = malloc(total);
    uint32_t *off = malloc(((size_t)nseqs + 1) * sizeof *off);
    if (!ids || !off)
    {
        free(ids), free(off);
        return fail(RC_ENOMEM, "alloc sequence batch");
    }
    off[0] = 0;
    for (unsigned q = 0; q < nseqs && !rc; ++q)
    {
        uint8_t *one = dcp_host_seq_ids(&seqs[q], &rc);
        if (one)
        {
            memcpy(ids + off[q], one, seqs[q].size);
            free(one);
        }
        off[q + 1] = off[q] + seqs[q].size;
    }
    int drc = 0;
    if (!rc && (drc = dcp_gpu_seqs_upload(t->gpu, ids, off, nseqs))) rc = fail((enum rc)drc, "%s", dcp_gpu_last_error(t->gpu));
    free(ids);
    free(off);
    struct dcp_scan_params prm = {t->multi_hits, t->hmmer3_compat, (float)t->lrt_threshold, 0, 0};
    if ((double)prm.lrt_threshold != t->lrt_threshold && !rc)
    {
        /* the device filter compares in float32 like xmath_lrt's imm_float (scan_thread.c:121-123); a
         * threshold that is not a float32 value would be rounded: keep every candidate at the next
         * lower float and re-apply the caller's double when the rows are written */
        prm.lrt_threshold = nextafterf(prm.lrt_threshold, -INFINITY);
    }
    if (!rc && (drc = dcp_gpu_scan(t->gpu, &prm))) rc = fail((enum rc)drc, "failed to run viterbi: %s", dcp_gpu_last_error(t->gpu));
    stat_add(&g_stats.submit_s, stat_now() - t_loaded);
    return rc;
}

static enum rc batch_trace(struct scan_thread *t, struct imm_seq const *seqs, unsigned nseqs, unsigned nprofiles,
                           struct batch_result *res)
{
    memset(res, 0, sizeof *res);
    double const t_in = stat_now();
    int drc = dcp_gpu_sync(t->gpu);
    double const t_scanned = stat_now();
    stat_add(&g_stats.scan_wait_s, t_scanned - t_in);
    if (drc) return fail((enum rc)drc, "failed to run viterbi: %s", dcp_gpu_last_error(t->gpu));
    /* the LRT filter ran on the device: only hits come back, sorted by (seq, profile) */
    enum rc rc = RC_OK;
    unsigned nhits = 0;
    drc = dcp_gpu_fetch_hits(t->gpu, NULL, 0, &nhits); /* count first */
    if (drc && drc != DCP_ENOMEM) return fail((enum rc)drc, "fetch hits");
    if ((uint64_t)nhits > (uint64_t)nprofiles * nseqs) return fail(RC_EFAIL, "more hits than pairs");
    if (nhits == 0) return RC_OK;
    res->hits = malloc((size_t)nhits * sizeof *res->hits);
    if (!res->hits) rc = fail(RC_ENOMEM, "alloc hits");
    if (!rc && (drc = dcp_gpu_fetch_hits(t->gpu, res->hits, nhits, &nhits))) rc = fail((enum rc)drc, "fetch hits");
    uint64_t cap = 0;
    for (unsigned h = 0; !rc && h < nhits; ++h)
        cap += 2 * (uint64_t)seqs[res->hits[h].seq_idx].size +
               2 * (uint64_t)dcp_profile_core_size(t->impls[res->hits[h].profile_idx]) + 16;
    if (!rc && cap > UINT32_MAX) rc = fail(RC_ENOMEM, "too many path steps in one batch");
    if (!rc)
    {
        res->steps = malloc((size_t)cap * sizeof *res->steps);
        res->soff = malloc(((size_t)nhits + 1) * sizeof *res->soff);
        if (!res->steps || !res->soff) rc = fail(RC_ENOMEM, "alloc paths");
    }
    if (!rc && (drc = dcp_gpu_trace_paths(t->gpu, res->hits, nhits, t->multi_hits, t->hmmer3_compat, 0, res->steps,
                                          (unsigned)cap, res->soff, NULL)))
        rc = fail((enum rc)drc, "%s", dcp_gpu_last_error(t->gpu));
    if (rc) batch_result_free(res);
    else
    {
        res->nhits = nhits;
#pragma omp atomic
        g_stats.hits += nhits;
#pragma omp atomic
        g_stats.steps += res->soff[nhits];
    }
    stat_add(&g_stats.trace_s, stat_now() - t_scanned);
    return rc;
}

static enum rc batch_rows_timed(struct scan_thread *t, struct imm_seq const *seqs, int64_t const *seq_ids, struct batch_result *res);
static enum rc batch_rows(struct scan_thread *t, struct imm_seq const *seqs, int64_t const *seq_ids, struct batch_result *res)
{
    double const t_in = stat_now();
    enum rc rc = batch_rows_timed(t, seqs, seq_ids, res);
    stat_add(&g_stats.rows_s, stat_now() - t_in);
    return rc;
}

static enum rc batch_rows_timed(struct scan_thread *t, struct imm_seq const *seqs, int64_t const *seq_ids, struct batch_result *res)
{
    enum rc rc = RC_OK;
    unsigned const nhits = res->nhits;
    struct dcp_hit const *hits = res->hits;
    if (nhits)
    {
        /* Product rows: one per hit, each formatted into its own memory stream -- every emitting step of a
         * path is decoded against its state's codon distribution, a few hundred per row -- by however many
         * host threads OpenMP gives this call (none extra when the caller already runs it inside a parallel
         * region), then appended to the partition's file in hit order: the file is what the serial loop wrote. */
        char **row = NULL;
        size_t *row_len = NULL;
        FILE *out = NULL;
        rc = prod_thread_fp(t->id, &out);
        if (!rc)
        {
            row = calloc(nhits, sizeof *row);
            row_len = calloc(nhits, sizeof *row_len);
            if (!row || !row_len) rc = fail(RC_ENOMEM, "alloc product rows");
        }
        if (!rc)
        {
            enum rc shared = RC_OK;
            /* inside scan_run_source this is the second level: the cores are shared by the partitions' threads */
            int const team = omp_in_parallel() ? omp_get_num_procs() / (omp_get_num_threads() > 0 ? omp_get_num_threads() : 1) : 0;
#pragma omp parallel for schedule(dynamic, 8) if (nhits >= 64) num_threads(team > 0 ? (team > 32 ? 32 : team) : omp_get_max_threads())
            for (unsigned h = 0; h < nhits; ++h)
            {
                enum rc cur;
#pragma omp atomic read
                cur = shared;
                if (cur) continue;
                unsigned const q = hits[h].seq_idx;
                imm_float const lrt = xmath_lrt_f32(hits[h].null_loglik, hits[h].alt_loglik);
                if (!imm_lprob_is_finite(lrt) || lrt < t->lrt_threshold) continue; /* scan_thread.c:123 */
                struct protein_profile view;
                struct imm_path path = {0};
                struct protein_match pm;
                struct prod pr = t->prod; /* job fields; the per-hit ones follow (:125-128) */
                FILE *ms = NULL;
                enum rc r = profile_view(t, hits[h].profile_idx, &view);
                if (!r)
                {
                    snprintf(pr.profile_name, sizeof pr.profile_name, "%s", view.super.accession);
                    if (seq_ids) pr.seq_id = seq_ids[q];
                    pr.null_loglik = (double)hits[h].null_loglik;
                    pr.alt_loglik = (double)hits[h].alt_loglik;
                    r = dcp_host_path_assign(&path, res->steps + res->soff[h], res->soff[h + 1] - res->soff[h]);
                    if (!r && !(ms = open_memstream(&row[h], &row_len[h]))) r = fail(RC_ENOMEM, "alloc product row");
                    if (!r)
                    {
                        match_setup(&pm.match, &view.super);
                        r = prod_fwrite_fp(ms, &pr, &seqs[q], &path, t->write_match_func, &pm.match);
                    }
                    if (ms && fclose(ms) && !r) r = fail(RC_EIO, "failed to write prod");
                    free(view.alt.match_ndists);
                }
                free(path.steps);
                if (r)
                {
#pragma omp atomic write
                    shared = r;
                }
            }
            rc = shared;
        }
        for (unsigned h = 0; row && h < nhits; ++h)
        {
            if (!rc && row[h] && row_len && fwrite(row[h], 1, row_len[h], out) != row_len[h]) rc = fail(RC_EIO, "failed to write prod");
            free(row[h]);
        }
        free(row);
        free(row_len);
        /* the thread's own prod / match keep what the serial loop left in them: the last hit's fields */
        if (!rc)
        {
            unsigned const h = nhits - 1u;
            t->prod.null_loglik = (double)hits[h].null_loglik;
            t->prod.alt_loglik = (double)hits[h].alt_loglik;
            t->null.prod.loglik = hits[h].null_loglik;
            t->alt.prod.loglik = hits[h].alt_loglik;
            if (seq_ids) t->prod.seq_id = seq_ids[hits[h].seq_idx];
        }
    }
    batch_result_free(res);
    return rc;
}

enum rc thread_run_batch(struct scan_thread *t, int tid, struct imm_seq const *seqs, int64_t const *seq_ids,
                         unsigned nseqs)
{
    unsigned n = 0;
    enum rc rc = batch_check(t, seqs, nseqs, &n);
    if (rc || n == 0) return rc;
    struct batch_result res;
    if ((rc = batch_submit(t, tid, seqs, nseqs))) return rc;
    if ((rc = batch_trace(t, seqs, nseqs, n, &res))) return rc;
    return batch_rows(t, seqs, seq_ids, &res);
}

enum rc thread_run(struct scan_thread *t, int tid)
{
    if (!t->reader || !t->seq) return fail(RC_EINVAL, "thread has no reader or sequence");
    int64_t id = t->prod.seq_id;
    return thread_run_batch(t, tid, t->seq, &id, 1);
}

/* ============================== scan loop (src/server/scan.c:215-269) ========================== */
/* The resident database a scan with cfg.keep_resident left behind (one scan at a time per process, as in the
 * reference: scan.c:41-43 keeps its state in file-scope statics too). */
static struct
{
    bool valid;
    dev_t dev;
    ino_t ino;
    off_t size;
    struct timespec mtime;
    unsigned nparts;
    bool by_cells;
    struct
    {
        dcp_gpu_ctx *gpu;
        dcp_profile **impls;
        unsigned nimpls;
    } part[NUM_THREADS];
} g_resident;

void scan_resident_release(void)
{
    if (!g_resident.valid) return;
    for (unsigned i = 0; i < g_resident.nparts; ++i)
    {
        if (g_resident.part[i].gpu) dcp_gpu_ctx_del(g_resident.part[i].gpu);
        for (unsigned k = 0; k < g_resident.part[i].nimpls; ++k)
            dcp_profile_del(g_resident.part[i].impls[k]);
        free(g_resident.part[i].impls);
    }
    memset(&g_resident, 0, sizeof g_resident);
}

static bool resident_matches(struct stat const *st, unsigned nparts, bool by_cells)
{
    return g_resident.valid && g_resident.dev == st->st_dev && g_resident.ino == st->st_ino &&
           g_resident.size == st->st_size && g_resident.mtime.tv_sec == st->st_mtim.tv_sec &&
           g_resident.mtime.tv_nsec == st->st_mtim.tv_nsec && g_resident.nparts == nparts && g_resident.by_cells == by_cells;
}

enum rc scan_run_source(char const *db_filename, struct scan_cfg cfg, unsigned nthreads, scan_next_seq_func_t next_seq,
                        void *arg)
{
    if (!db_filename || !next_seq) return fail(RC_EINVAL, "bad scan arguments");
    if (nthreads == 0 || nthreads > NUM_THREADS) return fail(RC_EINVAL, "bad number of threads");
    unsigned const batch = cfg.batch ? cfg.batch : 1;
    /* A pass may grow to one and a half times its target when the source ends inside it (below): the host copies
     * of a pass and the look-ahead queue are sized for that. */
    unsigned const limit_n = batch + batch / 2u;
    unsigned long const limit_s = cfg.batch_symbols + cfg.batch_symbols / 2u;
    unsigned const qcap = limit_n + 1u;
    memset(&g_stats, 0, sizeof g_stats);
    /* the partitions' host threads (one per device) each fan out once more: unpacking a partition and formatting a
     * batch's product rows are many-core jobs of their own */
    int const omp_levels = omp_get_max_active_levels();
    if (omp_levels < 2) omp_set_max_active_levels(2);

    /* prepare_readers (scan.c:45-74) */
    FILE *fp = fopen(db_filename, "rb");
    if (!fp)
    {
        if (omp_levels < 2) omp_set_max_active_levels(omp_levels);
        return fail(RC_EIO, "failed to open database");
    }
    struct protein_db_reader *db = calloc(1, sizeof *db);
    struct profile_reader *reader = calloc(1, sizeof *reader);
    struct scan_thread *th = NULL;
    struct batch_result *pend = NULL;
    struct imm_seq *bseq[2] = {NULL, NULL};
    int64_t *bid[2] = {NULL, NULL};
    char **btext[2] = {NULL, NULL};
    char **qtext = NULL; /* look-ahead queue: sequences fetched from the source, not yet in a pass */
    int64_t *qid = NULL;
    unsigned long *qlen = NULL;
    unsigned qn = 0;
    unsigned long qs = 0;
    bool db_open = false, reader_open = false, have_stat = false;
    struct stat st;
    unsigned nparts = 0;
    enum rc rc = RC_OK;
    if (!db || !reader)
    {
        rc = fail(RC_ENOMEM, "alloc scan");
        goto cleanup;
    }
    if ((rc = protein_db_reader_open(db, fp))) goto cleanup;
    db_open = true;
    rc = cfg.balance_by_cells ? profile_reader_setup_balanced(reader, &db->super, nthreads)
                              : profile_reader_setup(reader, &db->super, nthreads);
    if (rc) goto cleanup;
    reader_open = true;
    nparts = profile_reader_npartitions(reader);
    struct imm_abc const *abc = &db->nuclt.super;

    if ((rc = prod_fopen(nparts))) goto cleanup;
    th = calloc(nparts ? nparts : 1, sizeof *th);
    /* two host copies of a batch (A/B): while batch i is scanned, batch i-1's rows are still being written */
    for (int k = 0; k < 2; ++k)
    {
        bseq[k] = malloc((size_t)limit_n * sizeof *bseq[k]);
        bid[k] = malloc((size_t)limit_n * sizeof *bid[k]);
        btext[k] = calloc(limit_n, sizeof *btext[k]);
    }
    pend = calloc(nparts ? nparts : 1, sizeof *pend);
    qtext = calloc(qcap, sizeof *qtext);
    qid = malloc((size_t)qcap * sizeof *qid);
    qlen = malloc((size_t)qcap * sizeof *qlen);
    if (!qtext || !qid || !qlen || !th || !pend || !bseq[0] || !bseq[1] || !bid[0] || !bid[1] || !btext[0] || !btext[1])
    {
        rc = fail(RC_ENOMEM, "alloc scan");
        goto cleanup;
    }
    for (unsigned i = 0; i < nparts; ++i)
    {
        thread_init(&th[i], i, reader, cfg.multi_hits, cfg.hmmer3_compat, cfg.lrt_threshold, protein_match_write_func);
        thread_setup_job(&th[i], imm_abc_typeid(abc), reader->profile_typeid, cfg.scan_id);
    }
    have_stat = fstat(fileno(fp), &st) == 0;
    if (have_stat && resident_matches(&st, nparts, cfg.balance_by_cells))
    {
        /* the previous scan's resident partitions: theirs to this scan's threads (thread_prepare finds
         * db_resident set and goes straight to the sequences) */
        for (unsigned i = 0; i < nparts; ++i)
        {
            th[i].gpu = g_resident.part[i].gpu;
            th[i].impls = g_resident.part[i].impls;
            th[i].nimpls = g_resident.part[i].nimpls;
            th[i].db_resident = th[i].gpu != NULL && th[i].nimpls == reader->partition_size[i];
            if (!th[i].db_resident && th[i].gpu) /* never expected: fall back to a fresh load */
            {
                dcp_gpu_ctx_del(th[i].gpu);
                for (unsigned k = 0; k < th[i].nimpls; ++k)
                    dcp_profile_del(th[i].impls[k]);
                free(th[i].impls);
                th[i].gpu = NULL, th[i].impls = NULL, th[i].nimpls = 0;
            }
        }
        memset(&g_resident, 0, sizeof g_resident);
    }
    else
        scan_resident_release(); /* another database (or layout): its memory goes first */

    /* The loop of scan.c:224-258, software-pipelined per partition: batch i is submitted to the device, THEN
     * batch i-1's product rows are formatted and written by the host while the device scans, then batch
     * i's hits and paths are collected.  Rows reach the product files in the same order as without the
     * overlap; with one batch in flight per partition a sequence buffer is reused two batches later. */
    unsigned npend = 0;  /* sequences of the batch whose rows are still to be written */
    int cur = 0;         /* host copy the next batch is fetched into */
    /* Passes are cut from a look-ahead queue.  A pass's target is `batch` sequences or cfg.batch_symbols bases,
     * whichever comes first; the queue is filled until it exceeds one and a half targets or the source ends.  If the
     * source ends while the queue is within one and a half targets, the whole queue is ONE pass -- a job never ends in
     * a sliver of a pass, which costs a mixed-length job as much device time as a full one (the pass's longest
     * sequence bounds it: DESIGN.md 4.3c) -- otherwise the pass is the shortest prefix that reaches a target. */
    bool src_end = false;
    for (; !rc;)
    {
        /* each sequence is copied, the source may reuse its buffer (scan.c:227-229) */
        while (!src_end && qn <= limit_n && (cfg.batch_symbols == 0 || qs <= limit_s) && qn < qcap)
        {
            struct scan_seq s = {0, NULL};
            enum rc r = next_seq(arg, &s);
            if (r == RC_END)
            {
                src_end = true;
                break;
            }
            if (r || !s.data)
            {
                rc = r ? r : fail(RC_EINVAL, "sequence source returned no data");
                break;
            }
            qtext[qn] = strdup(s.data);
            if (!qtext[qn])
            {
                rc = fail(RC_ENOMEM, "alloc sequence");
                break;
            }
            qid[qn] = s.id;
            qlen[qn] = (unsigned long)strlen(qtext[qn]);
            qs += qlen[qn];
            ++qn;
        }
        if (rc || qn == 0) break;
        unsigned nb = 0;
        if (src_end && qn <= limit_n && (cfg.batch_symbols == 0 || qs <= limit_s)) nb = qn; /* the job's last pass */
        else
        {
            unsigned long symbols = 0;
            while (nb < qn && nb < batch && (cfg.batch_symbols == 0 || nb == 0 || symbols < cfg.batch_symbols))
                symbols += qlen[nb++];
        }
        for (unsigned i = 0; i < nb; ++i)
        {
            free(btext[cur][i]);
            btext[cur][i] = qtext[i];
            bseq[cur][i] = imm_seq(imm_str(btext[cur][i]), abc);
            bid[cur][i] = qid[i];
            qs -= qlen[i];
        }
        memmove(qtext, qtext + nb, (size_t)(qn - nb) * sizeof *qtext);
        memmove(qid, qid + nb, (size_t)(qn - nb) * sizeof *qid);
        memmove(qlen, qlen + nb, (size_t)(qn - nb) * sizeof *qlen);
        qn -= nb;
        enum rc shared = RC_OK;
        ++g_stats.passes;
        if (nparts == 0) continue; /* an empty database: every sequence is consumed, nothing is scored */
        int const prev = cur ^ 1;
#pragma omp parallel for schedule(static, 1) num_threads(nparts)
        for (unsigned i = 0; i < nparts; ++i)
        {
            unsigned n = 0;
            enum rc r = batch_check(&th[i], bseq[cur], nb, &n);
            bool const work = !r && n != 0;
            if (work) r = batch_submit(&th[i], (int)i, bseq[cur], nb);
            /* the device is busy with batch i: now the rows of batch i-1 */
            if (npend && pend[i].nhits)
            {
                enum rc r2 = batch_rows(&th[i], bseq[prev], bid[prev], &pend[i]);
                if (!r) r = r2;
            }
            if (work && !r) r = batch_trace(&th[i], bseq[cur], nb, n, &pend[i]);
            if (r)
            {
#pragma omp atomic write
                shared = r; /* scan.c:246-248: a failing partition fails the scan */
            }
            else if (cfg.progress)
                cfg.progress((unsigned long)nb * reader->partition_size[i], cfg.progress_arg);
        }
        rc = shared;
        npend = nb;
        cur ^= 1;
    }
    /* the last batch's rows */
    if (!rc && npend && nparts)
    {
        enum rc shared = RC_OK;
        int const prev = cur ^ 1;
#pragma omp parallel for schedule(static, 1) num_threads(nparts)
        for (unsigned i = 0; i < nparts; ++i)
        {
            enum rc r = pend[i].nhits ? batch_rows(&th[i], bseq[prev], bid[prev], &pend[i]) : RC_OK;
            if (r)
            {
#pragma omp atomic write
                shared = r;
            }
        }
        rc = shared;
    }
    if (!rc) rc = prod_fclose(); /* work_finishup: header + every thread's rows, in thread order */

cleanup:
    if (rc) prod_fcleanup();
    if (!rc && cfg.keep_resident && th && have_stat && nparts > 0)
    {
        bool all = true;
        for (unsigned i = 0; i < nparts; ++i)
            all = all && th[i].db_resident;
        if (all)
        {
            g_resident.valid = true;
            g_resident.dev = st.st_dev, g_resident.ino = st.st_ino, g_resident.size = st.st_size;
            g_resident.mtime = st.st_mtim;
            g_resident.nparts = nparts, g_resident.by_cells = cfg.balance_by_cells;
            for (unsigned i = 0; i < nparts; ++i)
            {
                g_resident.part[i].gpu = th[i].gpu;
                g_resident.part[i].impls = th[i].impls;
                g_resident.part[i].nimpls = th[i].nimpls;
                th[i].gpu = NULL, th[i].impls = NULL, th[i].nimpls = 0; /* thread_cleanup leaves them alone */
            }
        }
    }
    for (unsigned i = 0; th && i < nparts; ++i)
        thread_cleanup(&th[i]);
    for (unsigned i = 0; pend && i < nparts; ++i)
        batch_result_free(&pend[i]);
    for (int k = 0; k < 2; ++k)
    {
        for (unsigned i = 0; btext[k] && i < limit_n; ++i)
            free(btext[k][i]);
        free(btext[k]), free(bid[k]), free(bseq[k]);
    }
    for (unsigned i = 0; qtext && i < qn; ++i)
        free(qtext[i]);
    free(qtext), free(qid), free(qlen);
    free(pend), free(th);
    if (reader_open) profile_reader_del(reader);
    if (db_open) db_reader_close(&db->super);
    free(reader), free(db);
    fclose(fp);
    if (omp_levels < 2) omp_set_max_active_levels(omp_levels);
    return rc;
}

struct list_source
{
    struct scan_seq const *seqs;
    unsigned n, next;
};

static enum rc list_next(void *arg, struct scan_seq *seq)
{
    struct list_source *src = arg;
    if (src->next >= src->n) return RC_END;
    *seq = src->seqs[src->next++];
    return RC_OK;
}

enum rc scan_run_local(char const *db_filename, struct scan_seq const *seqs, unsigned nseqs, unsigned nthreads,
                       bool multi_hits, bool hmmer3_compat, double lrt_threshold, int64_t scan_id, unsigned batch,
                       FILE *prods)
{
    if (!seqs || !prods || batch == 0) return fail(RC_EINVAL, "bad scan arguments");
    struct list_source src = {seqs, nseqs, 0};
    struct scan_cfg cfg = {scan_id, multi_hits, hmmer3_compat, lrt_threshold, batch, true, false, NULL, NULL, 0};
    enum rc rc = scan_run_source(db_filename, cfg, nthreads, list_next, &src);
    if (rc) return rc;
    char buf[1 << 16];
    size_t n;
    FILE *fp = prod_final_fp();
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0)
        if (fwrite(buf, 1, n, prods) != n) rc = fail(RC_EIO, "fail to finish product");
    prod_final_cleanup();
    return rc;
}
