/* TEST STUB of the third-party deciphon-sched "sched/structs.h" (absent from the reference tree): only the
 * members integration/scan_run_adapter.c and the reference's src/server/scan.c touch, with the names
 * src/sched/sched.c:20-230 parses into them.  Sizes are this test's, not deciphon-sched's. */
#ifndef STUB_SCHED_STRUCTS_H
#define STUB_SCHED_STRUCTS_H
#include <stdbool.h>
#include <stdint.h>

enum
{
    SCHED_JOB_ERROR_SIZE = 256,
    SCHED_PATH_SIZE = 4096,
    SCHED_FILENAME_SIZE = 128,
    SCHED_SEQ_NAME_SIZE = 256,
    SCHED_SEQ_SIZE = 1 << 20,
};
enum sched_job_state
{
    SCHED_PEND,
    SCHED_RUN,
    SCHED_DONE,
    SCHED_FAIL
};
struct sched_db
{
    int64_t id, xxh3;
    char filename[SCHED_PATH_SIZE];
    int64_t hmm_id;
};
struct sched_scan
{
    int64_t id, db_id;
    bool multi_hits, hmmer3_compat;
    int64_t job_id;
};
struct sched_seq
{
    int64_t id, scan_id;
    char name[SCHED_SEQ_NAME_SIZE];
    char data[SCHED_SEQ_SIZE];
};
#endif
