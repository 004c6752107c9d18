/* TEST STUB of src/server/file.h: file_ensure_local(filename, xxh3, fetch) calls `fetch` when the file is
 * not on disk (the reference also compares its xxh3). */
#ifndef STUB_FILE_H
#define STUB_FILE_H
#include "deciphon_host.h"
#include <stdint.h>
enum rc file_ensure_local(char const *filename, int64_t xxh3, enum rc (*fetch)(char const *filename, int64_t xxh3));
#endif
