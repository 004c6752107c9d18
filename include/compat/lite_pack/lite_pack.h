/* Forwarding header: code written against the reference's `lite_pack/lite_pack.h (EBI-Metagenomics/lite-pack 0.3.0: only the calls deciphon's db code makes)` builds against
 * this library (-Iinclude -Iinclude/compat). Everything is declared in deciphon_host.h. */
#include "deciphon_host.h"
