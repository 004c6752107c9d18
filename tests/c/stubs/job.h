/* TEST STUB of src/server/job.h */
#ifndef STUB_JOB_H
#define STUB_JOB_H
#include <stdint.h>
void job_set_fail(int64_t job_id, char const *fmt, ...) __attribute__((format(printf, 2, 3)));
#endif
