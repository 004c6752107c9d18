/* host_internal.h -- shared by the C files of the host layer (not installed). */
#ifndef DCP_HOST_INTERNAL_H
#define DCP_HOST_INTERNAL_H

#include "deciphon_host.h"

/* positions in protein_profile.xtrans / dcp_xtrans() output (the values protein_profile_setup writes,
 * src/model/protein_profile.c:190-214) */
enum
{
    DCP_HOST_X_RR = 0,
    DCP_HOST_X_SB = 1,
    DCP_HOST_X_SN = 2,
    DCP_HOST_X_NN = 3,
    DCP_HOST_X_NB = 4,
    DCP_HOST_X_ET = 5,
    DCP_HOST_X_EC = 6,
    DCP_HOST_X_CC = 7,
    DCP_HOST_X_CT = 8,
    DCP_HOST_X_EB = 9,
    DCP_HOST_X_EJ = 10,
    DCP_HOST_X_JJ = 11,
    DCP_HOST_X_JB = 12,
};

/* log at the point of detection, return the code (include/deciphon/core/logging.h:32-72) */
enum rc dcp_host_fail(enum rc rc, char const *fmt, ...) __attribute__((format(printf, 2, 3)));
enum rc dcp_host_path_assign(struct imm_path *path, struct dcp_step const *steps, unsigned n);
uint8_t *dcp_host_seq_ids(struct imm_seq const *seq, enum rc *rc);
void dcp_host_forget_profile(dcp_profile *impl);
enum rc dcp_host_adopt(struct protein_profile *p, dcp_profile *impl, int rc);
/* another FILE* on the file behind fp, with a position of its own (NULL on failure) */
FILE *dcp_host_reopen(FILE *fp);

#endif
