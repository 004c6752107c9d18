/* End-to-end timing of the drop-in path a deciphon server would drive, through the C host layer only:
 * press a database file (protein_profile_sample + protein_db_writer), then scan_run_source() = open the file,
 * make it resident, batch the sequences through thread_run_batch (upload, scan, hits, device traceback,
 * product rows into the per-thread tmp file), join the products file.
 *
 *   gcc -std=gnu11 -O2 -I include profiles/host_scan_probe.c -o /tmp/host_scan_probe \
 *       -L deciphon-old_amd -ldeciphon_host -ldcp_hip -lm -fopenmp -Wl,-rpath,$PWD/deciphon-old_amd
 *   /tmp/host_scan_probe [nprofiles=2000] [nseqs=2000] [seq_len=1000] [batch=1000] [lrt_threshold=10] [batch_symbols=0]
 *   seq_len = 0: mixed lengths, log-uniform on 100 .. 10 000 nt (BASELINE configs[4]'s queries); batch_symbols: a device
 *   pass also closes at that many bases (scan_cfg.batch_symbols; the scan_run adapter asks for 2 Mi)
 *
 * Core sizes: the lognormal draw of BASELINE config C3 restated with this file's own generator (median 150,
 * sigma 0.6, clipped to 30..2000); sequences uniform over ACGT.  Prints one line per phase. */
#include "deciphon_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64(void)
{
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double next_unit(void) { return ((double)(next_u64() >> 11) + 0.5) / 9007199254740992.0; }
static double next_normal(void) { return sqrt(-2.0 * log(next_unit())) * cos(6.283185307179586 * next_unit()); }

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

int main(int argc, char **argv)
{
    unsigned const nprof = argc > 1 ? (unsigned)atoi(argv[1]) : 2000u;
    unsigned const nseqs = argc > 2 ? (unsigned)atoi(argv[2]) : 2000u;
    unsigned const len = argc > 3 ? (unsigned)atoi(argv[3]) : 1000u;
    unsigned const batch = argc > 4 ? (unsigned)atoi(argv[4]) : 1000u;
    double const threshold = argc > 5 ? atof(argv[5]) : 10.0; /* 1e30: no hits, i.e. no traceback and no product rows */
    unsigned long const batch_symbols = argc > 6 ? strtoul(argv[6], NULL, 10) : 0ul;

    struct imm_nuclt const *nuclt = imm_super(&imm_dna_iupac);
    struct imm_nuclt_code code;
    imm_nuclt_code_init(&code, nuclt);

    char path[64];
    snprintf(path, sizeof path, "/tmp/dcp_probe_db_XXXXXX");
    int fd = mkstemp(path);
    if (fd < 0) return perror("mkstemp"), 1;
    FILE *fp = fdopen(fd, "wb");
    double t0 = now();
    struct protein_db_writer db = {0};
    if (protein_db_writer_open(&db, fp, &imm_amino_iupac, nuclt, PROTEIN_CFG_DEFAULT) != RC_OK) return 2;
    double sum_core = 0;
    for (unsigned p = 0; p < nprof; ++p)
    {
        double m = exp(log(150.0) + 0.6 * next_normal());
        unsigned const M = (unsigned)(m < 30 ? 30 : m > 2000 ? 2000 : m + 0.5);
        struct protein_profile prof;
        char acc[16];
        snprintf(acc, sizeof acc, "PF%05u", p);
        protein_profile_init(&prof, acc, &imm_amino_iupac, &code, PROTEIN_CFG_DEFAULT);
        if (protein_profile_sample(&prof, 0xDEC1F0u + p, M) != RC_OK) return 3;
        if (protein_db_writer_pack_profile(&db, &prof) != RC_OK) return 4;
        profile_del(&prof.super);
        sum_core += M;
    }
    if (db_writer_close((struct db_writer *)&db, true) != RC_OK) return 5;
    long const db_bytes = ftell(fp);
    fclose(fp);
    printf("press: %u profiles (sum M = %.0f), %.1f MB .dcp, %.2f s\n", nprof, sum_core, db_bytes / 1e6, now() - t0);

    /* the host half of making the DB resident, alone: open + header + unpack every profile once */
    {
        t0 = now();
        FILE *rf = fopen(path, "rb");
        struct protein_db_reader rdb;
        struct profile_reader reader;
        if (!rf || protein_db_reader_open(&rdb, rf) != RC_OK) return 7;
        if (profile_reader_setup(&reader, (struct db_reader *)&rdb, 1) != RC_OK) return 8;
        if (profile_reader_rewind_all(&reader) != RC_OK) return 9;
        unsigned n = 0;
        struct profile *prof = NULL;
        enum rc rc;
        while ((rc = profile_reader_next(&reader, 0, &prof)) == RC_OK)
            ++n;
        if (rc != RC_END || n != nprof) return 10;
        profile_reader_del(&reader);
        db_reader_close((struct db_reader *)&rdb);
        fclose(rf);
        printf("read + unpack: %u profiles, %.2f s (%.0f MB/s)\n", n, now() - t0, db_bytes / 1e6 / (now() - t0));
    }
    if (nseqs == 0) return remove(path), 0;

    struct scan_seq *seqs = calloc(nseqs, sizeof *seqs);
    unsigned *lens = malloc((size_t)nseqs * sizeof *lens);
    size_t total_len = 0;
    for (unsigned q = 0; q < nseqs; ++q)
    {
        lens[q] = len ? len : (unsigned)(exp(log(100.0) + next_unit() * log(100.0)) + 0.5); /* 100 .. 10 000, log-uniform */
        total_len += lens[q];
    }
    char *text = malloc(total_len + nseqs);
    size_t at = 0;
    for (unsigned q = 0; q < nseqs; ++q)
    {
        char *s = text + at;
        for (unsigned i = 0; i < lens[q]; i += 32)
        {
            uint64_t r = next_u64();
            for (unsigned k = 0; k < 32 && i + k < lens[q]; ++k, r >>= 2)
                s[i + k] = "ACGT"[r & 3u];
        }
        s[lens[q]] = '\0';
        seqs[q].id = (int64_t)q + 1;
        seqs[q].data = s;
        at += lens[q] + 1u;
    }

    double const cells = sum_core * (double)total_len;
    /* job 0 includes the first-use costs (module load); jobs 0 and 1 load the database like the reference's
     * scan_run does per job; job 1 leaves it resident (cfg.keep_resident) and job 2 picks it up */
    for (int job = 0; job < 3; ++job)
    {
        struct list_src src = {seqs, nseqs, 0};
        struct scan_cfg cfg = {.scan_id = 1, .multi_hits = true, .hmmer3_compat = false, .lrt_threshold = threshold,
                               .batch = batch, .balance_by_cells = true, .keep_resident = job >= 1, .batch_symbols = batch_symbols};
        t0 = now();
        enum rc rc = scan_run_source(path, cfg, 1, list_src_next, &src);
        double const dt = now() - t0;
        if (rc != RC_OK) return fprintf(stderr, "scan_run_source: rc %d\n", (int)rc), 6;
        long rows = -1; /* header line */
        FILE *prods = prod_final_fp();
        for (int ch; (ch = fgetc(prods)) != EOF;)
            rows += ch == '\n';
        prod_final_cleanup();
        printf("scan_run_source job %d%s: %u seqs x %s nt (%zu bases), passes of <= %u seqs / %lu bases: %.3f s wall = %.1f Gcell/s end to "
               "end (file -> resident DB -> products), %ld product rows, %.1f seqs/s\n",
               job, job == 2 ? " (database resident from job 1)" : "", nseqs, len ? "fixed" : "100-10000", total_len, batch,
               batch_symbols, dt, cells / dt / 1e9, rows, nseqs / dt);
        struct scan_stats ss;
        scan_last_stats(&ss);
        printf("    host seconds: load %.3f, submit (encode + upload + enqueue) %.3f, waiting for the scans %.3f, hit fetch + traceback %.3f, "
               "product rows %.3f; %u passes, %lu hits, %lu path steps\n",
               ss.load_s, ss.submit_s, ss.scan_wait_s, ss.trace_s, ss.rows_s, ss.passes, ss.hits, ss.steps);
    }
    scan_resident_release();
    remove(path);
    free(text);
    free(lens);
    free(seqs);
    return 0;
}
