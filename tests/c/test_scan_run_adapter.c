/*
 * integration/scan_run_adapter.c, compiled UNCHANGED against stub declarations of the scheduler interfaces
 * (tests/c/stubs/) and driven by an in-memory scheduler: the job's database and sequences come through
 * api_get_scan_by_job_id / api_get_db / api_download_db / api_scan_next_seq, progress goes to
 * api_increment_job_progress, the products file to api_upload_prods_file, the final state to
 * api_set_job_state -- the calls src/server/scan.c:215-269 makes.  The uploaded products must equal
 * scan_run_local's on the same database and sequences, row for row.  Needs a GPU (no CPU fallback).
 */
#include "../../integration/scan_run_adapter.c"

#include <stdarg.h>
#include <stdlib.h>
#include <unistd.h>

static int failed;
#define CHECK(cond)                                                                                                \
    do                                                                                                             \
    {                                                                                                              \
        if (!(cond))                                                                                               \
        {                                                                                                          \
            fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond);                               \
            failed++;                                                                                              \
        }                                                                                                          \
    } while (0)

/* ---- the in-memory scheduler ------------------------------------------------------------------------- */
enum { NSEQ = 9, NPROF = 6 };
static struct
{
    char remote_db[64]; /* where api_download_db reads from */
    char local_db[64];  /* what api_get_db names: absent until file_ensure_local fetches it */
    char text[NSEQ][640];
    int fail_at_seq;   /* api_scan_next_seq returns RC_EAPI at this sequence (0 = never) */
    /* what the job did */
    int downloads, progress_sum, progress_calls, uploads, state, state_calls, next_calls;
    char fail_msg[256];
    char *uploaded;
} sch;

void sched_seq_init(struct sched_seq *seq) { memset(seq, 0, sizeof *seq); }

enum rc api_get_scan_by_job_id(int64_t job_id, struct sched_scan *scan, struct api_rc *arc)
{
    arc->rc = 0;
    if (job_id != 42) return RC_EAPI;
    *scan = (struct sched_scan){.id = 77, .db_id = 5, .multi_hits = true, .hmmer3_compat = false, .job_id = 42};
    return RC_OK;
}
enum rc api_get_db(int64_t id, struct sched_db *db, struct api_rc *arc)
{
    arc->rc = 0;
    if (id != 5) return RC_EAPI;
    memset(db, 0, sizeof *db);
    db->id = 5, db->xxh3 = 1234, db->hmm_id = 1;
    snprintf(db->filename, sizeof db->filename, "%s", sch.local_db);
    return RC_OK;
}
enum rc api_download_db(int64_t id, FILE *fp, struct api_rc *arc)
{
    arc->rc = 0;
    sch.downloads++;
    if (id != 5) return RC_EAPI;
    FILE *src = fopen(sch.remote_db, "rb");
    if (!src) return RC_EIO;
    char buf[1 << 14];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, src)) > 0)
        if (fwrite(buf, 1, n, fp) != n) return RC_EIO;
    fclose(src);
    return RC_OK;
}
enum rc file_ensure_local(char const *filename, int64_t xxh3, enum rc (*fetch)(char const *, int64_t))
{
    return access(filename, R_OK) == 0 ? RC_OK : fetch(filename, xxh3);
}
enum rc api_scan_num_seqs(int64_t scan_id, unsigned *n, struct api_rc *arc)
{
    arc->rc = 0;
    *n = NSEQ;
    return scan_id == 77 ? RC_OK : RC_EAPI;
}
/* sequence ids 1001..: the one AFTER seq_id (0 = the first) */
enum rc api_scan_next_seq(int64_t scan_id, int64_t seq_id, struct sched_seq *seq, struct api_rc *arc)
{
    arc->rc = 0;
    sch.next_calls++;
    if (scan_id != 77) return RC_EAPI;
    int const idx = seq_id == 0 ? 0 : (int)(seq_id - 1000);
    if (idx >= NSEQ) return RC_END;
    if (sch.fail_at_seq && idx + 1 == sch.fail_at_seq)
    {
        arc->rc = 7;
        snprintf(arc->msg, sizeof arc->msg, "scheduler said no");
        return RC_EAPI;
    }
    seq->id = 1001 + idx, seq->scan_id = 77;
    snprintf(seq->name, sizeof seq->name, "seq%d", idx);
    snprintf(seq->data, sizeof seq->data, "%s", sch.text[idx]);
    return RC_OK;
}
enum rc api_increment_job_progress(int64_t job_id, int inc, struct api_rc *arc)
{
    arc->rc = 0;
    CHECK(job_id == 42 && inc > 0);
    sch.progress_sum += inc, sch.progress_calls++;
    return RC_OK;
}
static char *slurp_path(char const *path)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    rewind(fp);
    char *t = calloc((size_t)n + 1, 1);
    if (fread(t, 1, (size_t)n, fp) != (size_t)n) t[0] = 0;
    fclose(fp);
    return t;
}
enum rc api_upload_prods_file(char const *filepath, struct api_rc *arc)
{
    arc->rc = 0;
    sch.uploads++;
    free(sch.uploaded);
    sch.uploaded = slurp_path(filepath);
    return sch.uploaded ? RC_OK : RC_EIO;
}
enum rc api_set_job_state(int64_t job_id, enum sched_job_state st, char const *msg, struct api_rc *arc)
{
    arc->rc = 0;
    CHECK(job_id == 42);
    (void)msg;
    sch.state = (int)st, sch.state_calls++;
    return RC_OK;
}
void job_set_fail(int64_t job_id, char const *fmt, ...)
{
    CHECK(job_id == 42);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(sch.fail_msg, sizeof sch.fail_msg, fmt, ap);
    va_end(ap);
    sch.state = SCHED_FAIL;
}

/* ---- a pressed database with profiles that have real hits --------------------------------------------- */
static char g_domain[NPROF][3 * 64 + 1];
static void press(char const *path)
{
    struct imm_nuclt const *nuclt = imm_super(&imm_dna_iupac);
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, nuclt);
    FILE *fp = fopen(path, "wb");
    struct protein_db_writer db = {0};
    CHECK(fp && protein_db_writer_open(&db, fp, &imm_amino_iupac, nuclt, PROTEIN_CFG_DEFAULT) == RC_OK);
    static char const amino[] = "ACDEFGHIKLMNPQRSTVWY";
    for (unsigned p = 0; p < NPROF; ++p)
    {
        unsigned const M = 18 + 7 * p;
        imm_float null[20], *match = malloc(sizeof(imm_float) * 20 * M), *trans = malloc(sizeof(imm_float) * 7 * (M + 1));
        for (int a = 0; a < 20; ++a)
            null[a] = logf(0.05f);
        for (unsigned k = 0; k < M; ++k)
        {
            char const fav = ((k * (p + 2) + p) % 5 < 2) ? 'W' : 'M'; /* Trp = TGG, Met = ATG: one codon each */
            for (int a = 0; a < 20; ++a)
                match[20 * k + a] = logf(amino[a] == fav ? 0.81f : 0.01f);
            memcpy(g_domain[p] + 3 * k, fav == 'W' ? "TGG" : "ATG", 3);
        }
        g_domain[p][3 * M] = 0;
        for (unsigned i = 0; i <= M; ++i)
        {
            imm_float *t = trans + 7 * i;
            t[0] = logf(0.95f), t[1] = t[2] = logf(0.025f), t[3] = t[5] = logf(0.6f), t[4] = t[6] = logf(0.4f);
            if (i == 0) t[6] = -INFINITY, t[5] = 0;
            if (i == M) t[2] = t[6] = -INFINITY, t[0] = logf(0.975f), t[5] = 0;
        }
        struct protein_profile prof;
        char acc[16];
        snprintf(acc, sizeof acc, "PF%05u", p);
        protein_profile_init(&prof, acc, &imm_amino_iupac, &code, PROTEIN_CFG_DEFAULT);
        CHECK(protein_profile_from_params(&prof, M, null, match, trans) == RC_OK);
        CHECK(protein_db_writer_pack_profile(&db, &prof) == RC_OK);
        profile_del(&prof.super);
        free(match), free(trans);
    }
    CHECK(db_writer_close((struct db_writer *)&db, true) == RC_OK);
    fclose(fp);
}

static int cmp_str(void const *a, void const *b) { return strcmp(*(char *const *)a, *(char *const *)b); }
static unsigned sorted_rows(char *text, char **rows, unsigned cap)
{
    unsigned n = 0;
    for (char *l = strtok(text, "\n"); l && n < cap; l = strtok(NULL, "\n"))
        rows[n++] = l;
    qsort(rows, n, sizeof *rows, cmp_str);
    return n;
}

int main(void)
{
    snprintf(sch.remote_db, sizeof sch.remote_db, "/tmp/dcp_adapter_remote_%d.dcp", (int)getpid());
    snprintf(sch.local_db, sizeof sch.local_db, "/tmp/dcp_adapter_local_%d.dcp", (int)getpid());
    press(sch.remote_db);
    unlink(sch.local_db);
    static char const *const flank[NSEQ] = {"ACGTTGCAAGGCTTAACC", "TTGACCA", "GGGCATCATCAGGAC", "AC", "CCGTA",
                                            "GATTACAGATTACA",     "TGCATGCAAT", "GGA",          "CATTAG"};
    struct scan_seq seqs[NSEQ];
    for (unsigned q = 0; q < NSEQ; ++q)
    {
        char const *dom = q % 2 == 0 ? g_domain[(q / 2) % NPROF] : "";
        snprintf(sch.text[q], sizeof sch.text[q], "%s%s%s", flank[q], dom, flank[(q + 4) % NSEQ]);
        seqs[q] = (struct scan_seq){1001 + q, sch.text[q]};
    }

    /* the job, through the adapter */
    enum rc rc = scan_run(42, 2);
    CHECK(rc == RC_OK);
    CHECK(sch.downloads == 1 && access(sch.local_db, R_OK) == 0); /* fetched once, then on disk */
    CHECK(sch.uploads == 1 && sch.uploaded != NULL);
    CHECK(sch.state == SCHED_DONE && sch.state_calls == 1 && sch.fail_msg[0] == 0);
    CHECK(sch.progress_sum == 100 && sch.progress_calls >= 1 && sch.progress_calls <= 100);
    CHECK(sch.next_calls == NSEQ + 1); /* every sequence once, then RC_END */

    /* the same database and sequences through scan_run_local */
    FILE *want_fp = tmpfile();
    CHECK(scan_run_local(sch.remote_db, seqs, NSEQ, 2, true, false, 10.0, 77, 4, want_fp) == RC_OK);
    fflush(want_fp);
    fseek(want_fp, 0, SEEK_END);
    long wn = ftell(want_fp);
    rewind(want_fp);
    char *want = calloc((size_t)wn + 1, 1);
    CHECK(fread(want, 1, (size_t)wn, want_fp) == (size_t)wn);
    fclose(want_fp);
    if (sch.uploaded)
    {
        size_t const hl = strlen(prod_header());
        CHECK(strncmp(sch.uploaded, prod_header(), hl) == 0 && strncmp(want, prod_header(), hl) == 0);
        char *a[256], *b[256];
        unsigned na = sorted_rows(sch.uploaded + hl, a, 256), nb = sorted_rows(want + hl, b, 256);
        CHECK(na == nb && na >= 5); /* five planted domains at least */
        for (unsigned i = 0; i < na && i < nb; ++i)
            CHECK(strcmp(a[i], b[i]) == 0);
        fprintf(stderr, "adapter: %u product rows, progress calls %d\n", na, sch.progress_calls);
    }
    free(want);

    /* a second job on the same database: already local (no download), resident on the device */
    sch.uploads = sch.state_calls = sch.progress_sum = sch.next_calls = 0;
    CHECK(scan_run(42, 2) == RC_OK);
    CHECK(sch.downloads == 1 && sch.uploads == 1 && sch.progress_sum == 100 && sch.state == SCHED_DONE);

    /* the scheduler fails in the middle: the job is failed with its message, nothing is uploaded */
    sch.uploads = 0, sch.fail_at_seq = 5, sch.fail_msg[0] = 0;
    CHECK(scan_run(42, 2) == RC_EAPI);
    CHECK(sch.uploads == 0 && sch.state == SCHED_FAIL && strstr(sch.fail_msg, "scheduler said no") != NULL);
    sch.fail_at_seq = 0;
    /* an unknown job */
    CHECK(scan_run(43, 2) == RC_EAPI);

    scan_resident_release();
    unlink(sch.remote_db);
    unlink(sch.local_db);
    free(sch.uploaded);
    if (failed) fprintf(stderr, "%d check(s) failed\n", failed);
    else printf("all checks passed\n");
    return failed;
}
