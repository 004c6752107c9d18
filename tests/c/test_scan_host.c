/*
 * C-level parity test of the host layer (include/deciphon_host.h) on the GPU.
 *
 * Part 1 follows what the reference checks in test/protein_profile.c (sampled M=2
 * profile, eps 0.1, 32-nt query): setup(0) is RC_EINVAL; null loglik, path length and
 * end steps; alt loglik for both entry distributions, path ends, decoded codons.
 * Part 2 presses a small database into a .dcp file (protein_db_writer), reads it back
 * (protein_db_reader, profile_reader) and drives thread_run / scan_run_local over it with a planted
 * domain, checking the product rows.
 *
 * Expected values are the reference's goldens (float32 tolerance 5e-5 relative,
 * test/hope_support.h:26).  Exit status = number of failed checks.
 */
#include "deciphon_host.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int failed;
#define CHECK(cond)                                                            \
    do                                                                         \
    {                                                                          \
        if (!(cond))                                                           \
        {                                                                      \
            fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond); \
            failed++;                                                          \
        }                                                                      \
    } while (0)
#define NEAR(a, b) CHECK(fabs((double)(a) - (double)(b)) <= 5e-5 * fabs((double)(b)))

static char const query[] = "ATGAAACGCATTAGCACCACCATTACCACCAC";
static char const *const want_codons[10] = {"ATG", "AAA", "CGC", "ATA", "GCA", "CCA", "CCT", "TAC", "CAC", "CAC"};

static void golden_profile(enum entry_dist entry, double want_alt)
{
    struct imm_nuclt const *nuclt = imm_super(&imm_dna_iupac);
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, nuclt);
    struct protein_profile prof;
    protein_profile_init(&prof, "accession", &imm_amino_iupac, &code, protein_cfg(entry, 0.1f));
    CHECK(protein_profile_sample(&prof, 1, 2) == RC_OK);

    struct imm_seq seq = imm_seq(IMM_STR(query), prof.super.code->abc);
    CHECK(protein_profile_setup(&prof, 0, true, false) == RC_EINVAL);
    CHECK(protein_profile_setup(&prof, imm_seq_size(&seq), true, false) == RC_OK);

    /* null model */
    struct imm_prod prod = imm_prod();
    struct imm_task *task = imm_task_new(&prof.null.dp);
    CHECK(task != NULL);
    CHECK(imm_task_setup(task, &seq) == IMM_OK);
    CHECK(imm_dp_viterbi(&prof.null.dp, task, &prod) == IMM_OK);
    NEAR(prod.loglik, -48.9272687711);
    CHECK(imm_path_nsteps(&prod.path) == 11);
    char name[IMM_STATE_NAME_SIZE];
    CHECK(imm_path_step(&prod.path, 0)->seqlen == 3);
    CHECK(imm_path_step(&prod.path, 0)->state_id == PROTEIN_R_STATE);
    protein_state_name(imm_path_step(&prod.path, 0)->state_id, name);
    CHECK(strcmp(name, "R") == 0);
    CHECK(imm_path_step(&prod.path, 10)->seqlen == 2);
    CHECK(imm_path_step(&prod.path, 10)->state_id == PROTEIN_R_STATE);
    imm_prod_reset(&prod);
    imm_task_del(task);

    /* alt model */
    struct imm_dp const *dp = profile_alt_dp(&prof.super);
    task = imm_task_new(dp);
    CHECK(imm_task_setup(task, &seq) == IMM_OK);
    CHECK(imm_dp_viterbi(dp, task, &prod) == IMM_OK);
    NEAR(prod.loglik, want_alt);
    CHECK(imm_path_nsteps(&prod.path) == 14);
    CHECK(imm_path_step(&prod.path, 0)->seqlen == 0);
    CHECK(imm_path_step(&prod.path, 0)->state_id == PROTEIN_S_STATE);
    protein_state_name(imm_path_step(&prod.path, 0)->state_id, name);
    CHECK(strcmp(name, "S") == 0);
    CHECK(imm_path_step(&prod.path, 13)->seqlen == 0);
    CHECK(imm_path_step(&prod.path, 13)->state_id == PROTEIN_T_STATE);
    protein_state_name(imm_path_step(&prod.path, 13)->state_id, name);
    CHECK(strcmp(name, "T") == 0);

    /* codons of the emitting steps */
    struct protein_codec codec = protein_codec_init(&prof, &prod.path);
    unsigned any = imm_abc_any_symbol_id(imm_super(nuclt));
    struct imm_codon codon = imm_codon(nuclt, any, any, any);
    enum rc rc;
    unsigned i = 0;
    while (!(rc = protein_codec_next(&codec, &seq, &codon)))
    {
        if (i < 10)
        {
            struct imm_codon want = IMM_CODON(nuclt, want_codons[i]);
            CHECK(want.a == codon.a && want.b == codon.b && want.c == codon.c);
        }
        ++i;
    }
    CHECK(rc == RC_END);
    CHECK(i == 10);

    /* imm keeps whatever transitions the last protein_profile_setup wrote: a shorter sequence is scored
     * with the 32-nt transitions (not refused), and differs from the score after a matching setup */
    struct imm_seq shorter = imm_subseq(&seq, 0, 10);
    CHECK(imm_task_setup(task, &shorter) == IMM_OK);
    CHECK(imm_dp_viterbi(dp, task, &prod) == IMM_OK);
    imm_float const stale = prod.loglik;
    CHECK(isfinite(stale));
    CHECK(protein_profile_setup(&prof, 10, true, false) == RC_OK);
    CHECK(imm_dp_viterbi(dp, task, &prod) == IMM_OK);
    CHECK(isfinite(prod.loglik) && prod.loglik != stale);

    /* a profile that never saw protein_profile_setup runs with the LOG1 = 0 defaults
     * (protein_model.c:322-340; test/protein_db.c:73 does exactly this): scores are finite and at
     * least as high as with any real (<= 0) special transitions */
    struct protein_profile fresh;
    protein_profile_init(&fresh, "fresh", &imm_amino_iupac, &code, protein_cfg(entry, 0.1f));
    CHECK(protein_profile_sample(&fresh, 1, 2) == RC_OK);
    CHECK(imm_task_reset(task, &fresh.alt.dp) == IMM_OK);
    CHECK(imm_task_setup(task, &seq) == IMM_OK);
    CHECK(imm_dp_viterbi(&fresh.alt.dp, task, &prod) == IMM_OK);
    CHECK(isfinite(prod.loglik) && prod.loglik > (imm_float)want_alt);
    CHECK(imm_path_step(&prod.path, 0)->state_id == PROTEIN_S_STATE);
    profile_del(&fresh.super);

    /* RNA sequences are scored over their own alphabet; the any-symbol is rejected */
    struct imm_nuclt_code rcode;
    imm_nuclt_code_init(&rcode, imm_super(&imm_rna_iupac));
    struct protein_profile rprof;
    protein_profile_init(&rprof, "rna", &imm_amino_iupac, &rcode, protein_cfg(entry, 0.1f));
    CHECK(protein_profile_sample(&rprof, 1, 2) == RC_OK);
    char rna[sizeof query];
    for (size_t c = 0; c < sizeof query; ++c)
        rna[c] = query[c] == 'T' ? 'U' : query[c];
    struct imm_seq rseq = imm_seq(imm_str(rna), rprof.super.code->abc);
    CHECK(protein_profile_setup(&rprof, imm_seq_size(&rseq), true, false) == RC_OK);
    CHECK(imm_task_reset(task, &rprof.alt.dp) == IMM_OK && imm_task_setup(task, &rseq) == IMM_OK);
    CHECK(imm_dp_viterbi(&rprof.alt.dp, task, &prod) == IMM_OK);
    NEAR(prod.loglik, want_alt);
    struct imm_seq xseq = imm_seq(imm_str("ACGUXACGU"), rprof.super.code->abc);
    CHECK(imm_task_setup(task, &xseq) == IMM_OK);
    CHECK(imm_dp_viterbi(&rprof.alt.dp, task, &prod) != IMM_OK);
    profile_del(&rprof.super);

    profile_del(&prof.super);
    imm_del(&prod);
    imm_del(task);
}

/* A profile whose node k strongly prefers Met (ATG) or Trp (TGG): both have one codon only. */
static void peaked_profile(struct protein_profile *prof, struct imm_nuclt_code const *code, char const *acc,
                           unsigned M, unsigned phase, char *domain /* 3*M + 1 */)
{
    protein_profile_init(prof, acc, &imm_amino_iupac, code, PROTEIN_CFG_DEFAULT);
    static char const amino[] = "ACDEFGHIKLMNPQRSTVWY";
    imm_float null[20], *match = malloc(sizeof(imm_float) * 20 * M), *trans = malloc(sizeof(imm_float) * 7 * (M + 1));
    for (int a = 0; a < 20; ++a)
        null[a] = logf(1.0f / 20);
    for (unsigned k = 0; k < M; ++k)
    {
        char fav = ((k + phase) % 3 == 0) ? 'W' : 'M';
        for (int a = 0; a < 20; ++a)
            match[20 * k + a] = logf(amino[a] == fav ? 0.81f : 0.01f);
        memcpy(domain + 3 * k, fav == 'W' ? "TGG" : "ATG", 3);
    }
    domain[3 * M] = '\0';
    for (unsigned i = 0; i <= M; ++i)
    {
        imm_float *t = trans + 7 * i; /* MM MI MD IM II DM DD */
        t[0] = logf(0.95f), t[1] = logf(0.025f), t[2] = logf(0.025f), t[3] = logf(0.6f), t[4] = logf(0.4f);
        t[5] = logf(0.6f), t[6] = logf(0.4f);
        if (i == 0) t[6] = -INFINITY, t[5] = 0.0f;
        if (i == M) t[2] = -INFINITY, t[6] = -INFINITY, t[0] = logf(0.975f), t[5] = 0.0f;
    }
    CHECK(protein_profile_from_params(prof, M, null, match, trans) == RC_OK);
    free(match);
    free(trans);
}

/* press `n` peaked profiles into a .dcp file the way hmm_press does: protein_db_writer_open,
 * protein_db_writer_pack_profile per profile, db_writer_close (test/protein_db.c:18-50) */
enum { NPROF = 5 };
static unsigned const kSizes[NPROF] = {20, 33, 41, 57, 60};
static char g_domain[NPROF][3 * 60 + 1];
static char g_db_path[64];

static void press_db(void)
{
    struct imm_nuclt const *nuclt = imm_super(&imm_dna_iupac);
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, nuclt);
    snprintf(g_db_path, sizeof g_db_path, "/tmp/dcp_test_db_XXXXXX");
    int fd = mkstemp(g_db_path);
    CHECK(fd >= 0);
    FILE *fp = fdopen(fd, "wb");
    CHECK(fp != NULL);
    struct protein_db_writer db = {0};
    CHECK(protein_db_writer_open(&db, fp, &imm_amino_iupac, nuclt, PROTEIN_CFG_DEFAULT) == RC_OK);
    for (unsigned p = 0; p < NPROF; ++p)
    {
        struct protein_profile prof;
        char acc[16];
        snprintf(acc, sizeof acc, "PF%05u", p);
        peaked_profile(&prof, &code, acc, kSizes[p], p, g_domain[p]);
        CHECK(protein_db_writer_pack_profile(&db, &prof) == RC_OK);
        profile_del(&prof.super);
    }
    /* a profile pressed with another cfg does not belong in this database */
    struct protein_profile odd;
    protein_profile_init(&odd, "odd", &imm_amino_iupac, &code, protein_cfg(ENTRY_DIST_UNIFORM, 0.01f));
    CHECK(protein_profile_sample(&odd, 3, 4) == RC_OK);
    CHECK(protein_db_writer_pack_profile(&db, &odd) == RC_EINVAL);
    profile_del(&odd.super);
    CHECK(db_writer_close((struct db_writer *)&db, true) == RC_OK);
    CHECK(fclose(fp) == 0);
}

static char *slurp(FILE *fp)
{
    fflush(fp);
    fseek(fp, 0, SEEK_END);
    long len = ftell(fp);
    rewind(fp);
    char *text = calloc((size_t)len + 1, 1);
    CHECK(fread(text, 1, (size_t)len, fp) == (size_t)len);
    return text;
}

static unsigned g_custom_steps;
static enum rc count_match_func(FILE *fp, void const *match)
{
    struct match const *m = match;
    g_custom_steps++;
    return fprintf(fp, "%u", (unsigned)m->step->seqlen) < 0 ? RC_EIO : RC_OK;
}

static void scan_threads(void)
{
    FILE *fp = fopen(g_db_path, "rb");
    CHECK(fp != NULL);
    struct protein_db_reader db = {0};
    CHECK(protein_db_reader_open(&db, fp) == RC_OK);
    CHECK(db.super.nprofiles == NPROF);
    CHECK(db.super.profile_typeid == PROFILE_PROTEIN);
    CHECK(imm_abc_typeid(imm_super(&db.nuclt)) == IMM_DNA);
    CHECK(imm_abc_typeid(imm_super(&db.amino)) == IMM_AMINO);
    CHECK(db.cfg.entry_dist == ENTRY_DIST_OCCUPANCY && db.cfg.epsilon == 0.01f);

    static struct profile_reader reader; /* 64 profiles inside: not for the stack */
    CHECK(profile_reader_setup(&reader, (struct db_reader *)&db, 0) == RC_EINVAL);
    CHECK(profile_reader_setup(&reader, (struct db_reader *)&db, NUM_THREADS + 1) == RC_EINVAL);
    CHECK(profile_reader_setup(&reader, (struct db_reader *)&db, 2) == RC_OK);
    CHECK(profile_reader_npartitions(&reader) == 2);
    CHECK(profile_reader_partition_size(&reader, 0) == 3 && profile_reader_partition_size(&reader, 1) == 2);
    CHECK(profile_reader_nprofiles(&reader) == NPROF);
    /* byte offsets: partition i+1 starts where partition i's profile_sizes end */
    CHECK(reader.partition_offset[1] - reader.partition_offset[0] ==
          (int64_t)db.super.profile_sizes[0] + db.super.profile_sizes[1] + db.super.profile_sizes[2]);
    struct profile *it = NULL;
    unsigned seen = 0;
    while (profile_reader_next(&reader, 1, &it) == RC_OK)
    {
        CHECK(profile_typeid(it) == PROFILE_PROTEIN);
        CHECK(it == (struct profile *)&reader.profiles[1]); /* borrowed, reused */
        char acc[16];
        snprintf(acc, sizeof acc, "PF%05u", 3 + seen);
        CHECK(strcmp(it->accession, acc) == 0);
        CHECK(((struct protein_profile *)it)->core_size == kSizes[3 + seen]);
        seen++;
    }
    CHECK(seen == 2);
    CHECK(profile_reader_next(&reader, 1, &it) == RC_END);
    CHECK(profile_reader_end(&reader, 1));

    /* an unpacked profile scores without protein_profile_setup, as test/protein_db.c:66-80 does */
    CHECK(profile_reader_rewind(&reader, 0) == RC_OK);
    CHECK(profile_reader_next(&reader, 0, &it) == RC_OK);
    {
        struct imm_prod prod = imm_prod();
        struct imm_task *task = imm_task_new(profile_alt_dp(it));
        struct imm_seq seq = imm_seq(imm_str(g_domain[0]), imm_super(&db.nuclt));
        CHECK(imm_task_setup(task, &seq) == IMM_OK);
        CHECK(imm_dp_viterbi(profile_alt_dp(it), task, &prod) == IMM_OK);
        CHECK(isfinite(prod.loglik));
        imm_del(task);
        imm_del(&prod);
    }

    /* query 0 carries profile 3's domain between random-looking flanks; query 1 is flank only */
    char q0[512], q1[] = "ACGTTGCAAGGCTTAACCGGTTACGATCGATTAGC";
    snprintf(q0, sizeof q0, "ACGTTGCAAGGCTTAACC%sGGTTACGATCGATTAGC", g_domain[3]);
    struct scan_thread th[2];
    CHECK(prod_fopen(2) == RC_OK);
    for (unsigned i = 0; i < 2; ++i)
    {
        thread_init(&th[i], i, &reader, true, false, 10.0, protein_match_write_func);
        thread_setup_job(&th[i], IMM_DNA, PROFILE_PROTEIN, 77);
    }
    char const *queries[2] = {q0, q1};
    for (int64_t s = 0; s < 2; ++s)
    {
        struct imm_seq seq = imm_seq(imm_str(queries[s]), imm_super(&db.nuclt));
        for (unsigned i = 0; i < 2; ++i)
        {
            thread_setup_seq(&th[i], &seq, 100 + s);
            CHECK(thread_run(&th[i], (int)i) == RC_OK);
        }
    }
    /* an empty sequence is RC_EINVAL, as protein_profile_setup reports it */
    struct imm_seq empty = imm_seq(imm_str(""), imm_super(&db.nuclt));
    thread_setup_seq(&th[0], &empty, 102);
    CHECK(thread_run(&th[0], 0) == RC_EINVAL);
    /* a symbol outside the alphabet is rejected too */
    struct imm_seq bad = imm_seq(imm_str("ACGTNACGT"), imm_super(&db.nuclt));
    thread_setup_seq(&th[0], &bad, 103);
    CHECK(thread_run(&th[0], 0) == RC_EINVAL);

    /* prod_fclose: header + thread 0's rows + thread 1's rows */
    CHECK(prod_fclose() == RC_OK);
    CHECK(prod_final_fp() != NULL && prod_final_path()[0] == '/');
    char *text = slurp(prod_final_fp());
    CHECK(strncmp(text, prod_header(), strlen(prod_header())) == 0);
    CHECK(strncmp(prod_header(), "scan_id\tseq_id\tprofile_name", 27) == 0);
    char *rows = text + strlen(prod_header());
    /* profile 3 lives in partition 1 and must be the hit for query 0; the flank-only query hits nothing */
    char *hit = strstr(rows, "77\t100\tPF00003\tdna\t");
    CHECK(hit != NULL);
    if (hit)
    {
        CHECK(strstr(hit, "\tprotein\t" DECIPHON_VERSION "\t,S,,;") != NULL);
        CHECK(strstr(hit, ";,T,,\n") != NULL);
        CHECK(strstr(hit, "ATG,M") != NULL && strstr(hit, ",ATG,M") != NULL);
        /* the fragments of the match column tile the query */
        char *row = strdup(hit);
        char *nl = strchr(row, '\n');
        if (nl) *nl = '\0';
        char *match = strrchr(row, '\t');
        char rebuilt[512] = "";
        for (char *m = strtok(match + 1, ";"); m; m = strtok(NULL, ";"))
            strncat(rebuilt, m, strcspn(m, ","));
        CHECK(strcmp(rebuilt, q0) == 0);
        free(row);
    }
    CHECK(strstr(rows, "\t101\t") == NULL); /* nothing reported for the second query */
    free(text);
    prod_final_cleanup();

    /* the caller's write_match_func is what writes the match column: one call per path step */
    struct scan_thread custom;
    thread_init(&custom, 1, &reader, true, false, 10.0, count_match_func);
    thread_setup_job(&custom, IMM_DNA, PROFILE_PROTEIN, 78);
    struct imm_seq seq0 = imm_seq(imm_str(q0), imm_super(&db.nuclt));
    thread_setup_seq(&custom, &seq0, 200);
    g_custom_steps = 0;
    CHECK(thread_run(&custom, 1) == RC_OK);
    CHECK(prod_fclose() == RC_OK);
    text = slurp(prod_final_fp());
    hit = strstr(text, "78\t200\tPF00003\t");
    CHECK(hit != NULL && g_custom_steps > 10);
    {
        /* one call per path step over all rows: steps = ';' separators + one per row */
        unsigned seps = 0, nrows = 0;
        for (char *c = text + strlen(prod_header()); *c; ++c)
            seps += *c == ';', nrows += *c == '\n';
        CHECK(nrows >= 1 && seps + nrows == g_custom_steps);
    }
    free(text);
    prod_final_cleanup();
    thread_cleanup(&custom);

    for (unsigned i = 0; i < 2; ++i)
        thread_cleanup(&th[i]);
    profile_reader_del(&reader);
    db_reader_close((struct db_reader *)&db);
    fclose(fp);
}

/* scan_run_local (batched dispatch, 2 partitions) must write the rows per-sequence thread_run writes */
static int cmp_str(void const *a, void const *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

static unsigned split_lines(char *text, char **lines, unsigned cap)
{
    unsigned n = 0;
    for (char *l = strtok(text, "\n"); l && n < cap; l = strtok(NULL, "\n"))
        lines[n++] = l;
    qsort(lines, n, sizeof *lines, cmp_str);
    return n;
}

/* cfg.keep_resident: the next scan of the same file picks the resident partitions up (no unpack, no upload)
 * and writes the same products; a scan of a different layout, or scan_resident_release(), drops them. */
struct list_src
{
    struct scan_seq const *seqs;
    unsigned n, at;
};
static enum rc list_src_next(void *arg, struct scan_seq *seq)
{
    struct list_src *l = arg;
    if (l->at == l->n) return RC_END;
    *seq = l->seqs[l->at++];
    return RC_OK;
}
static char *run_source(struct scan_seq const *seqs, unsigned n, unsigned nthreads, bool keep)
{
    struct list_src src = {seqs, n, 0};
    struct scan_cfg cfg = {.scan_id = 9, .multi_hits = true, .hmmer3_compat = false, .lrt_threshold = 10.0, .batch = 3,
                           .balance_by_cells = true, .keep_resident = keep};
    CHECK(scan_run_source(g_db_path, cfg, nthreads, list_src_next, &src) == RC_OK);
    char *text = slurp(prod_final_fp());
    prod_final_cleanup();
    return text;
}
static void resident_reuse(void)
{
    enum { NSEQ = 4 };
    char text[NSEQ][512];
    struct scan_seq seqs[NSEQ];
    for (unsigned q = 0; q < NSEQ; ++q)
    {
        snprintf(text[q], sizeof text[q], "GATTACA%sTTGACCAGG", q == 1 ? g_domain[2] : q == 3 ? g_domain[0] : "ACGT");
        seqs[q] = (struct scan_seq){500 + q, text[q]};
    }
    char *fresh = run_source(seqs, NSEQ, 2, false);
    char *first = run_source(seqs, NSEQ, 2, true);   /* loads, leaves resident */
    char *again = run_source(seqs, NSEQ, 2, true);   /* picks up */
    char *other = run_source(seqs, NSEQ, 1, true);   /* another partition count: reloads */
    char *third = run_source(seqs + 1, NSEQ - 1, 1, false); /* picks up, does not keep */
    scan_resident_release();                         /* nothing left: a no-op */
    CHECK(strcmp(fresh, first) == 0 && strcmp(fresh, again) == 0);
    CHECK(strstr(fresh, "\tPF00002\t") != NULL && strstr(fresh, "\tPF00000\t") != NULL);
    /* one partition: the same rows (one thread's file instead of two joined) */
    char *a[64], *b[64];
    unsigned na = split_lines(fresh + strlen(prod_header()), a, 64), nb = split_lines(other + strlen(prod_header()), b, 64);
    CHECK(na == nb && na >= 2);
    unsigned same = 0;
    for (unsigned i = 0; i < na; ++i)
        for (unsigned k = 0; k < nb; ++k)
            same += strcmp(a[i], b[k]) == 0;
    CHECK(same == na);
    CHECK(strstr(third, "9\t500\t") == NULL && strstr(third, "9\t501\t") != NULL);
    free(fresh), free(first), free(again), free(other), free(third);
}

/* cfg.batch_symbols: a device pass closes at a number of SYMBOLS as well as at a number of sequences (a pass is
 * sized by its work).  Products do not depend on where the passes are cut; the progress callback is called once per
 * pass and partition, so it counts the passes. */
static unsigned long g_passes, g_pairs;
static void count_pass(unsigned long pairs, void *arg)
{
    (void)arg;
#pragma omp critical(count_pass)
    {
        ++g_passes;
        g_pairs += pairs;
    }
}
static char *run_sized(struct scan_seq const *seqs, unsigned n, unsigned batch, unsigned long symbols)
{
    struct list_src src = {seqs, n, 0};
    struct scan_cfg cfg = {.scan_id = 9, .multi_hits = true, .hmmer3_compat = false, .lrt_threshold = 10.0, .batch = batch,
                           .balance_by_cells = true, .keep_resident = false, .progress = count_pass, .batch_symbols = symbols};
    g_passes = g_pairs = 0;
    CHECK(scan_run_source(g_db_path, cfg, 1, list_src_next, &src) == RC_OK);
    char *text = slurp(prod_final_fp());
    prod_final_cleanup();
    return text;
}
/* the rule of scan_run_source (include/deciphon_host.h, struct scan_cfg), restated: passes are cut from a look-ahead queue
 * that is filled until it EXCEEDS one and a half targets (batch sequences, batch_symbols bases) or the source ends; if the
 * source has ended and the queue is within one and a half targets, it is one pass; else the pass is the shortest prefix
 * that reaches a target */
static unsigned long expected_passes(unsigned long const *len, unsigned n, unsigned batch, unsigned long symbols)
{
    unsigned const limit_n = batch + batch / 2u;
    unsigned long const limit_s = symbols + symbols / 2u;
    unsigned next = 0, qn = 0;
    unsigned long qs = 0, passes = 0;
    unsigned long q[64];
    bool end = false;
    for (;;)
    {
        while (!end && qn <= limit_n && (symbols == 0 || qs <= limit_s))
        {
            if (next == n)
            {
                end = true;
                break;
            }
            q[qn++] = len[next], qs += len[next++];
        }
        if (qn == 0) return passes;
        unsigned nb = 0;
        if (end && qn <= limit_n && (symbols == 0 || qs <= limit_s)) nb = qn;
        else
        {
            unsigned long s = 0;
            while (nb < qn && nb < batch && (symbols == 0 || nb == 0 || s < symbols))
                s += q[nb++];
        }
        for (unsigned i = 0; i < nb; ++i)
            qs -= q[i];
        memmove(q, q + nb, (qn - nb) * sizeof *q);
        qn -= nb;
        ++passes;
    }
}
static void passes_sized_by_symbols(void)
{
    enum { NSEQ = 9 };
    static char text[NSEQ][700];
    struct scan_seq seqs[NSEQ];
    unsigned long len[NSEQ], total = 0;
    for (unsigned q = 0; q < NSEQ; ++q)
    {
        /* lengths from 4 to a few hundred symbols, two of them with a domain */
        char const *dom = q == 2 ? g_domain[1] : q == 7 ? g_domain[3] : "";
        size_t at = 0;
        for (unsigned r = 0; r < 1u + 7u * (q % 4u); ++r)
            at += (size_t)snprintf(text[q] + at, sizeof text[q] - at, "ACGT");
        snprintf(text[q] + at, sizeof text[q] - at, "%s%s", dom, q & 1 ? "GATTACA" : "");
        seqs[q] = (struct scan_seq){3000 + q, text[q]};
        total += len[q] = strlen(text[q]);
    }
    char *by_count = run_sized(seqs, NSEQ, 100, 0); /* one pass of nine sequences */
    unsigned long const pairs = g_pairs;
    CHECK(g_passes == 1 && pairs > 0);
    char *by_work = run_sized(seqs, NSEQ, 100, 60); /* a pass closes once it holds >= 60 symbols */
    unsigned long const want = expected_passes(len, NSEQ, 100, 60);
    CHECK(want >= 3 && want < NSEQ && g_passes == want && g_pairs == pairs);
    CHECK(strcmp(by_count, by_work) == 0); /* same rows, same order */
    char *one_each = run_sized(seqs, NSEQ, 100, 1); /* every pass holds at least one sequence */
    CHECK(g_passes == NSEQ && expected_passes(len, NSEQ, 100, 1) == NSEQ && g_pairs == pairs && strcmp(by_count, one_each) == 0);
    /* the count bound closes the passes first: 2, 2, 2 -- and the last three sequences are ONE pass: the source ends
     * within one and a half targets, so the job does not finish on a sliver */
    char *count_first = run_sized(seqs, NSEQ, 2, total);
    CHECK(g_passes == 4 && expected_passes(len, NSEQ, 2, total) == 4 && strcmp(by_count, count_first) == 0);
    /* batch = 1 is still the reference's loop shape: one sequence per pass */
    char *ref_shape = run_sized(seqs, NSEQ, 1, 0);
    CHECK(g_passes == NSEQ && strcmp(by_count, ref_shape) == 0);
    CHECK(strstr(by_count, "9\t3002\t") != NULL && strstr(by_count, "9\t3007\t") != NULL);
    free(by_count), free(by_work), free(one_each), free(count_first), free(ref_shape);
}

static void scan_run_batched(void)
{
    enum { NSEQ = 7 };
    char text[NSEQ][512];
    char const *flank[NSEQ] = {"ACGTTGCAAGGCTTAACC", "TTGACCA", "GGGCATCATCAGGAC", "A", "CCGTA", "GATTACAGATTACA", "TGCATGCAAT"};
    struct scan_seq seqs[NSEQ];
    for (unsigned q = 0; q < NSEQ; ++q)
    {
        char const *dom = q == 0 ? g_domain[3] : q == 5 ? g_domain[1] : q == 6 ? g_domain[4] : "";
        snprintf(text[q], sizeof text[q], "%s%s%s", flank[q], dom, flank[(q + 3) % NSEQ]);
        seqs[q] = (struct scan_seq){1000 + q, text[q]};
    }
    FILE *out = tmpfile();
    CHECK(out != NULL);
    CHECK(scan_run_local(g_db_path, seqs, NSEQ, 2, true, false, 10.0, 9, 3, out) == RC_OK);
    char *got = slurp(out);
    fclose(out);
    CHECK(strncmp(got, prod_header(), strlen(prod_header())) == 0);

    /* reference flow: one sequence at a time through thread_run, per (count-balanced) partition */
    FILE *fp = fopen(g_db_path, "rb");
    struct protein_db_reader db = {0};
    CHECK(protein_db_reader_open(&db, fp) == RC_OK);
    static struct profile_reader reader;
    CHECK(profile_reader_setup(&reader, (struct db_reader *)&db, 2) == RC_OK);
    struct scan_thread th[2];
    CHECK(prod_fopen(2) == RC_OK);
    for (unsigned i = 0; i < 2; ++i)
    {
        thread_init(&th[i], i, &reader, true, false, 10.0, protein_match_write_func);
        thread_setup_job(&th[i], IMM_DNA, PROFILE_PROTEIN, 9);
    }
    for (unsigned q = 0; q < NSEQ; ++q)
    {
        struct imm_seq seq = imm_seq(imm_str(text[q]), imm_super(&db.nuclt));
        for (unsigned i = 0; i < 2; ++i)
        {
            thread_setup_seq(&th[i], &seq, seqs[q].id);
            CHECK(thread_run(&th[i], (int)i) == RC_OK);
        }
    }
    CHECK(prod_fclose() == RC_OK);
    char *want = slurp(prod_final_fp());
    prod_final_cleanup();
    char *gl[64], *wlines[64];
    unsigned ng = split_lines(got + strlen(prod_header()), gl, 64);
    unsigned nw = split_lines(want + strlen(prod_header()), wlines, 64);
    CHECK(ng == nw && ng >= 3);
    for (unsigned i = 0; i < ng && i < nw; ++i)
        CHECK(strcmp(gl[i], wlines[i]) == 0);
    unsigned found = 0;
    for (unsigned i = 0; i < ng; ++i)
        found += strstr(gl[i], "9\t1000\tPF00003\t") == gl[i] || strstr(gl[i], "9\t1005\tPF00001\t") == gl[i] ||
                 strstr(gl[i], "9\t1006\tPF00004\t") == gl[i];
    CHECK(found == 3);
    /* error propagation: an empty sequence fails the whole scan with RC_EINVAL */
    struct scan_seq bad[2] = {{1, "ACGTACGTACGT"}, {2, ""}};
    out = tmpfile();
    CHECK(scan_run_local(g_db_path, bad, 2, 2, true, false, 10.0, 9, 2, out) == RC_EINVAL);
    fclose(out);
    /* a missing database is RC_EIO (scan.c:47-52) */
    out = tmpfile();
    CHECK(scan_run_local("/nonexistent/db.dcp", seqs, NSEQ, 2, true, false, 10.0, 9, 3, out) == RC_EIO);
    fclose(out);
    free(got);
    free(want);
    for (unsigned i = 0; i < 2; ++i)
        thread_cleanup(&th[i]);
    profile_reader_del(&reader);
    db_reader_close((struct db_reader *)&db);
    fclose(fp);
}

/* The reference calls imm_dp_viterbi from every thread of an OpenMP team (scan.c:239-249, through its own
 * thread_run): callers that share the library's single-pair context must be serialised, not race. */
static void concurrent_viterbi(void)
{
    enum { NT = 8 };
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, imm_super(&imm_dna_iupac));
    static struct protein_profile prof[NT];
    imm_float serial[NT], parallel[NT];
    for (unsigned t = 0; t < NT; ++t)
    {
        protein_profile_init(&prof[t], "p", &imm_amino_iupac, &code, PROTEIN_CFG_DEFAULT);
        CHECK(protein_profile_sample(&prof[t], 50 + t, 5 + 3 * t) == RC_OK);
        CHECK(protein_profile_setup(&prof[t], sizeof query - 1, true, false) == RC_OK);
    }
    for (int pass = 0; pass < 2; ++pass)
    {
#pragma omp parallel for num_threads(NT) schedule(static, 1) if (pass == 1)
        for (unsigned t = 0; t < NT; ++t)
        {
            struct imm_seq seq = imm_seq(IMM_STR(query), prof[t].super.code->abc);
            struct imm_prod prod = imm_prod();
            struct imm_task *task = imm_task_new(&prof[t].alt.dp);
            enum imm_rc rc = imm_task_setup(task, &seq);
            if (!rc) rc = imm_dp_viterbi(&prof[t].alt.dp, task, &prod);
            (pass ? parallel : serial)[t] = rc ? NAN : prod.loglik;
            imm_del(task);
            imm_del(&prod);
        }
    }
    for (unsigned t = 0; t < NT; ++t)
    {
        CHECK(isfinite(serial[t]) && serial[t] == parallel[t]);
        profile_del(&prof[t].super);
    }
}

/* One process per GPU, in C: shard map, communicator through the file rendezvous, scan, RCCL gather.
 * One rank is all a single-GPU box allows (RCCL refuses two ranks on one device); it still runs
 * dlopen(librccl), ncclCommInitRank, the counts all-gather, the local leg of the gather-v and the merge. */
static void one_process_per_gpu(void)
{
    enum { NP = 6 };
    unsigned const sizes[NP] = {40, 25, 70, 33, 90, 12};
    unsigned b = 9, e = 9;
    dcp_dist_shard(sizes, NP, 1, 0, &b, &e);
    CHECK(b == 0 && e == NP);
    dcp_dist_shard(sizes, NP, 3, 1, &b, &e);
    CHECK(b >= 1 && e > b && e < NP); /* the middle rank of three */

    char idfile[64];
    snprintf(idfile, sizeof idfile, "/tmp/dcp_dist_id_XXXXXX");
    int fd = mkstemp(idfile);
    CHECK(fd >= 0);
    close(fd);
    remove(idfile); /* rank 0 creates it */
    dcp_dist *comm = dcp_dist_init_from_file(idfile, 0, 1, 0, 10.0);
    CHECK(comm != NULL);
    if (!comm) return;
    CHECK(dcp_dist_rank(comm) == 0 && dcp_dist_nranks(comm) == 1);

    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, imm_super(&imm_dna_iupac));
    dcp_profile *impls[NP];
    char domain[NP][3 * 90 + 1];
    for (unsigned p = 0; p < NP; ++p)
    {
        struct protein_profile prof;
        char acc[16];
        snprintf(acc, sizeof acc, "PF%05u", p);
        peaked_profile(&prof, &code, acc, sizes[p], p, domain[p]);
        impls[p] = prof.impl;
        prof.impl = NULL; /* moved out */
        profile_del(&prof.super);
    }
    dcp_gpu_ctx *ctx = dcp_gpu_ctx_new(0);
    CHECK(ctx != NULL);
    CHECK(dcp_gpu_db_upload(ctx, impls, NP, 0) == DCP_OK);
    char text[1024];
    uint32_t off[4] = {0, 0, 0, 0};
    int n = snprintf(text, sizeof text, "ACGTTGCAAGGCTTAACC%sGGTTACG", domain[2]);
    off[1] = (uint32_t)n;
    n += snprintf(text + n, sizeof text - (size_t)n, "TTGACCAGGGCATCATCAGGACCCGTA");
    off[2] = (uint32_t)n;
    n += snprintf(text + n, sizeof text - (size_t)n, "GATTACA%sTGCATGCAAT", domain[4]);
    off[3] = (uint32_t)n;
    CHECK(dcp_gpu_seqs_upload_text(ctx, text, off, 3) == DCP_OK);
    struct dcp_scan_params prm = {1, 0, 10.0f, 0, 0};
    CHECK(dcp_gpu_scan(ctx, &prm) == DCP_OK && dcp_gpu_sync(ctx) == DCP_OK);
    struct dcp_hit want[64];
    unsigned nwant = 0;
    CHECK(dcp_gpu_fetch_hits(ctx, want, 64, &nwant) == DCP_OK && nwant >= 2);
    void *hits_dev = NULL, *nhits_dev = NULL;
    unsigned cap = 0;
    CHECK(dcp_gpu_hit_buffer(ctx, &hits_dev, &nhits_dev, &cap) == DCP_OK && hits_dev && nhits_dev && cap >= nwant);
    struct dcp_hit *all = NULL;
    unsigned nall = 0;
    CHECK(dcp_dist_gather_hits(comm, hits_dev, nhits_dev, cap, 100 /* shard starts at profile 100 */, 0,
                               dcp_gpu_stream(ctx), &all, &nall) == DCP_OK);
    CHECK(nall == nwant && all != NULL);
    for (unsigned h = 0; all && h < nall && h < nwant; ++h)
        CHECK(all[h].seq_idx == want[h].seq_idx && all[h].profile_idx == want[h].profile_idx + 100 &&
              all[h].alt_loglik == want[h].alt_loglik && all[h].null_loglik == want[h].null_loglik);
    dcp_dist_free_hits(all);
    dcp_gpu_ctx_del(ctx);
    for (unsigned p = 0; p < NP; ++p)
        dcp_profile_del(impls[p]);
    dcp_dist_free(comm);
    remove(idfile);
}

/* test/standard_profile.c:5-31 is a smoke test (no numeric golden): a standard profile is a typed
 * shell around two imm_dp, and the scan path never takes one (profile_reader.c:95-98) */
static void standard_profile_shell(void)
{
    struct imm_nuclt const *nuclt = imm_super(&imm_dna_iupac);
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, nuclt);
    struct standard_profile prof;
    standard_profile_init(&prof, "accession", &code.super);
    CHECK(profile_typeid(&prof.super) == PROFILE_STANDARD);
    CHECK(strcmp(profile_typeid_name(PROFILE_STANDARD), "standard") == 0);
    CHECK(strcmp(prof.super.accession, "accession") == 0);
    CHECK(profile_null_dp(&prof.super) == &prof.dp.null);
    CHECK(profile_alt_dp(&prof.super) == &prof.dp.alt);
    char name[IMM_STATE_NAME_SIZE];
    CHECK(prof.super.state_name(513, name) == 4);
    CHECK(strcmp(name, "S513") == 0);
    /* no model behind its dps: scoring is refused, not faked */
    struct imm_seq seq = imm_seq(IMM_STR(query), prof.super.code->abc);
    struct imm_task *task = imm_task_new(&prof.dp.alt);
    struct imm_prod prod = imm_prod();
    CHECK(task != NULL);
    CHECK(imm_task_setup(task, &seq) == IMM_OK);
    CHECK(imm_dp_viterbi(&prof.dp.alt, task, &prod) != IMM_OK);
    imm_task_del(task);
    imm_prod_del(&prod);
    profile_del(&prof.super);
}

int main(void)
{
    standard_profile_shell();
    golden_profile(ENTRY_DIST_UNIFORM, -55.59428153448);
    golden_profile(ENTRY_DIST_OCCUPANCY, -54.35543421312);
    press_db();
    scan_threads();
    scan_run_batched();
    passes_sized_by_symbols();
    resident_reuse();
    remove(g_db_path);
    one_process_per_gpu();
    concurrent_viterbi();
    CHECK(xmath_partition_size(20000, 8, 7) == 2500);
    CHECK(fabsf(xmath_lrt(-48.927f, -54.355f) - (-10.856f)) < 1e-3f);
    if (failed) fprintf(stderr, "%d check(s) failed\n", failed);
    else puts("test_scan_host: all checks passed");
    return failed;
}
