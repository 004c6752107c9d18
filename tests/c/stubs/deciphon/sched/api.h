/* TEST STUB of include/deciphon/sched/api.h: the declarations of the scheduler calls a scan job makes
 * (the reference's signatures, include/deciphon/sched/api.h:10-53); tests/c/test_scan_run_adapter.c
 * implements them over memory. */
#ifndef STUB_SCHED_API_H
#define STUB_SCHED_API_H
#include "deciphon_host.h" /* enum rc */
#include "sched/structs.h"
#include <stdint.h>
#include <stdio.h>

struct api_rc
{
    int rc;
    char msg[SCHED_JOB_ERROR_SIZE];
};
void sched_seq_init(struct sched_seq *);
enum rc api_get_scan_by_job_id(int64_t job_id, struct sched_scan *, struct api_rc *);
enum rc api_get_db(int64_t id, struct sched_db *, struct api_rc *);
enum rc api_download_db(int64_t id, FILE *fp, struct api_rc *);
enum rc api_scan_num_seqs(int64_t scan_id, unsigned *num_seqs, struct api_rc *);
enum rc api_scan_next_seq(int64_t scan_id, int64_t seq_id, struct sched_seq *, struct api_rc *);
enum rc api_increment_job_progress(int64_t job_id, int increment, struct api_rc *);
enum rc api_upload_prods_file(char const *filepath, struct api_rc *);
enum rc api_set_job_state(int64_t job_id, enum sched_job_state, char const *msg, struct api_rc *);
#endif
