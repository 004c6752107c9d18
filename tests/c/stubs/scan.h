/* TEST STUB of src/server/scan.h:6 */
#ifndef STUB_SCAN_H
#define STUB_SCAN_H
#include "deciphon_host.h"
#include <stdint.h>
enum rc scan_run(int64_t job_id, unsigned num_threads);
#endif
