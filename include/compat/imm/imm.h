/* Forwarding header: code written against the reference's `imm/imm.h (EBI-Metagenomics/imm 2.0.3: only the calls deciphon makes on the scan path)` builds against
 * this library (-Iinclude -Iinclude/compat). Everything is declared in deciphon_host.h. */
#include "deciphon_host.h"
