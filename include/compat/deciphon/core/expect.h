/* Forwarding header: code written against the reference's `include/deciphon/core/expect.h` builds against
 * this library (-Iinclude -Iinclude/compat). Everything is declared in deciphon_host.h. */
#include "deciphon_host.h"
