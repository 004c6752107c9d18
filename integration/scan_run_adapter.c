/*
 * integration/scan_run_adapter.c -- the one function src/server/job.c:18 dispatches scan jobs to,
 *
 *     enum rc scan_run(int64_t job_id, unsigned num_threads);          (src/server/scan.h:6)
 *
 * implemented on the MI355X host layer (include/deciphon_host.h: scan_run_source) with exactly the
 * scheduler calls the reference's own scan_run makes (src/server/scan.c:215-269 and its scan_init /
 * work_finishup, :104-213), in the same order and with the same failure reporting:
 *
 *     api_get_scan_by_job_id  -> the scan's id, database id and flags               scan.c:219
 *     api_get_db              -> the database's file name and checksum               scan.c:118-127
 *     file_ensure_local + api_download_db -> the pressed database on local disk      scan.c:76-95,130-134
 *     api_scan_num_seqs       -> number of tasks = sequences x profiles (progress)    scan.c:104-113,163-172
 *     api_scan_next_seq       -> one sequence per call until RC_END                  scan.c:227
 *     api_increment_job_progress -> percent units as pairs complete                  scan.c:97-102
 *     api_upload_prods_file   -> the joined products file                            scan.c:193
 *     api_set_job_state(DONE) / job_set_fail                                         scan.c:199, job.c:43-56
 *
 * This file is the SEAM only.  It is compiled inside the reference tree in place of src/server/scan.c
 * (INTEGRATION.md): the scheduler client (src/sched/api.c), its structs (the third-party
 * deciphon-sched "sched/structs.h"), job.c and file.c stay the reference's.  What differs from the
 * reference loop is inside scan_run_source: `SCAN_RUN_BATCH_SYMBOLS` bases (at most `SCAN_RUN_BATCH` sequences) are fetched per device pass
 * instead of one, partitions are device contexts (one host thread per GPU), and progress is consumed per
 * pass instead of per pair.  tests/c/test_scan_run_adapter.c compiles this file unchanged against stub
 * declarations of those interfaces and checks its products against scan_run_local's, row for row.
 */
#include "deciphon/sched/api.h" /* struct api_rc, api_*; pulls in sched/structs.h (sched_scan, sched_seq, sched_db) */
#include "deciphon_host.h"      /* in the reference tree: the deciphon/... headers forward here (include/compat) */
#include "file.h"               /* file_ensure_local (src/server/file.h) */
#include "job.h"                /* job_set_fail (src/server/job.h) */
#include "scan.h"

#include <string.h>

/* A device pass is sized by work (BASELINE configs[4] "dynamic batching"): its target is
 * SCAN_RUN_BATCH_SYMBOLS bases -- about 4 000 sequences of 1 kbp, or 2 000 of a 100 nt .. 10 kbp mix, some ten seconds
 * of device time against a Pfam-sized partition -- or SCAN_RUN_BATCH sequences, whichever comes first; a job's last
 * pass takes up to one and a half targets rather than leaving a sliver (scan_cfg.batch_symbols, deciphon_host.h).
 * Inside a pass the device packs the sequences by length into its wavefront slots; the larger the pass, the less its
 * longest sequence weighs (profiles/r04/host_scan_probe.txt). */
#ifndef SCAN_RUN_BATCH
#define SCAN_RUN_BATCH 32768
#endif
#ifndef SCAN_RUN_BATCH_SYMBOLS
#define SCAN_RUN_BATCH_SYMBOLS (4ul << 20)
#endif

/* one scan at a time per process, like the reference's file-scope `scan`, `api_rc`, `db` (scan.c:41-43) */
static struct
{
    struct sched_scan sched;
    struct sched_seq seq;
    struct sched_db db;
    struct api_rc api_rc;
    int64_t job_id;
    int64_t last_seq_id;
    unsigned long total, done;
    int percent_sent;
} ad;

static enum rc fetch_db(char const *filename, int64_t xxh3)
{
    (void)xxh3;
    FILE *fp = fopen(filename, "wb");
    if (!fp) return RC_EIO;
    struct api_rc arc = {0};
    enum rc rc = api_download_db(ad.sched.db_id, fp, &arc);
    if (rc)
    {
        job_set_fail(ad.sched.job_id, "failed to download database");
        fclose(fp);
        return rc;
    }
    if (fclose(fp))
    {
        job_set_fail(ad.sched.job_id, "failed to close database file");
        return RC_EIO;
    }
    return RC_OK;
}

/* scan_run_source's sequence source: the scheduler's cursor (scan.c:224-227: seq_id = the id of the last
 * sequence received, 0 to start).  The text stays valid until the next call: scan_run_source copies it
 * before it asks again, as the ownership note of SURVEY 8b requires. */
static enum rc next_seq(void *arg, struct scan_seq *out)
{
    (void)arg;
    enum rc rc = api_scan_next_seq(ad.sched.id, ad.last_seq_id, &ad.seq, &ad.api_rc);
    if (rc) return rc; /* RC_END ends the scan; anything else fails it */
    ad.last_seq_id = ad.seq.id;
    out->id = ad.seq.id;
    out->data = ad.seq.data;
    return RC_OK;
}

/* progress_setup(npartitions, total, 100, send_progress) of scan.c:176-178: the job's progress is sent
 * in whole percent units.  Called by the partitions' host threads, one at a time here. */
static void on_progress(unsigned long pairs, void *arg)
{
    (void)arg;
#pragma omp critical(scan_run_adapter_progress)
    {
        ad.done += pairs;
        int const percent = ad.total ? (int)(ad.done * 100ul / ad.total) : 100;
        if (percent > ad.percent_sent)
        {
            struct api_rc arc = {0};
            api_increment_job_progress(ad.job_id, percent - ad.percent_sent, &arc);
            ad.percent_sent = percent;
        }
    }
}

static unsigned long count_profiles(char const *filename)
{
    FILE *fp = fopen(filename, "rb");
    if (!fp) return 0;
    struct protein_db_reader db;
    unsigned long n = 0;
    if (!protein_db_reader_open(&db, fp))
    {
        n = db.super.nprofiles;
        db_reader_close(&db.super);
    }
    fclose(fp);
    return n;
}

enum rc scan_run(int64_t job_id, unsigned num_threads)
{
    enum rc rc = RC_OK;
    memset(&ad, 0, sizeof ad);
    ad.job_id = job_id;
    if ((rc = api_get_scan_by_job_id(job_id, &ad.sched, &ad.api_rc))) return rc;

    /* scan_init (scan.c:104-182) */
    if ((rc = api_get_db(ad.sched.db_id, &ad.db, &ad.api_rc)))
    {
        job_set_fail(ad.sched.job_id, "failed to get database");
        return rc;
    }
    if (ad.api_rc.rc)
    {
        job_set_fail(ad.sched.job_id, "%s", ad.api_rc.msg);
        return rc;
    }
    if ((rc = file_ensure_local(ad.db.filename, ad.db.xxh3, fetch_db)))
    {
        job_set_fail(ad.sched.job_id, "failed to have database on disk");
        return rc;
    }
    unsigned nseqs = 0;
    if ((rc = api_scan_num_seqs(ad.sched.id, &nseqs, &ad.api_rc)))
    {
        job_set_fail(ad.sched.job_id, "failed to compute number of tasks");
        return rc;
    }
    ad.total = (unsigned long)nseqs * count_profiles(ad.db.filename);

    /* the loop of scan.c:224-258, a pass of `SCAN_RUN_BATCH_SYMBOLS` bases (at most `SCAN_RUN_BATCH` sequences); lrt threshold 10 as scan.c:221 */
    sched_seq_init(&ad.seq);
    ad.last_seq_id = ad.seq.id;
    struct scan_cfg cfg = {.scan_id = ad.sched.id,
                           .multi_hits = ad.sched.multi_hits,
                           .hmmer3_compat = ad.sched.hmmer3_compat,
                           .lrt_threshold = 10.,
                           .batch = SCAN_RUN_BATCH,
                           .balance_by_cells = true, /* partitions = GPUs: balance by work, not by count */
                           .keep_resident = true,    /* the next job on the same database starts at the sequences */
                           .progress = on_progress,
                           .progress_arg = NULL,
                           .batch_symbols = SCAN_RUN_BATCH_SYMBOLS};
    rc = scan_run_source(ad.db.filename, cfg, num_threads, next_seq, NULL);
    if (rc)
    {
        if (rc == RC_EAPI) job_set_fail(job_id, "%s", ad.api_rc.msg);
        else job_set_fail(job_id, "thread_run error (%s)", RC_STRING(rc));
        return rc;
    }

    /* work_finishup (scan.c:184-213): scan_run_source has already joined the threads' rows (prod_fclose) */
    if ((rc = api_upload_prods_file(prod_final_path(), &ad.api_rc)))
    {
        job_set_fail(job_id, "failed to submit prods_file");
        prod_final_cleanup();
        return rc;
    }
    prod_final_cleanup();
    rc = api_set_job_state(job_id, SCHED_DONE, "", &ad.api_rc);
    if (rc) return rc;
    return ad.api_rc.rc ? RC_EAPI : RC_OK;
}
