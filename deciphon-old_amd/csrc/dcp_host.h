// dcp_host.h -- internal declarations shared by the host model code and the
// HIP side. Public C-ABI lives in include/dcp_gpu.h.
#ifndef DCP_HOST_H
#define DCP_HOST_H

#include "dcp_gpu.h"

// Rows of the per-profile transition matrix trans8[8][ldk]: the edges INTO
// node k (from node k-1 and from B) and node k's own insert edges.
// Reference wiring: src/model/protein_model.c:469-489 (core), :410-439 (entry).
enum
{
    DCP_T_ENTRY = 0, // B      -> M_k
    DCP_T_MM = 1,    // M_{k-1} -> M_k
    DCP_T_IM = 2,    // I_{k-1} -> M_k
    DCP_T_DM = 3,    // D_{k-1} -> M_k
    DCP_T_MD = 4,    // M_{k-1} -> D_k
    DCP_T_DD = 5,    // D_{k-1} -> D_k
    DCP_T_MI = 6,    // M_k     -> I_k
    DCP_T_II = 7,    // I_k     -> I_k
};

// Special transitions as protein_profile_setup writes them
// (src/model/protein_profile.c:190-214).
enum
{
    DCP_X_RR = 0,
    DCP_X_SB = 1,
    DCP_X_SN = 2,
    DCP_X_NN = 3,
    DCP_X_NB = 4,
    DCP_X_ET = 5,
    DCP_X_EC = 6,
    DCP_X_CC = 7,
    DCP_X_CT = 8,
    DCP_X_EB = 9,
    DCP_X_EJ = 10,
    DCP_X_JJ = 11,
    DCP_X_JB = 12,
    DCP_XSTRIDE = 16, // floats per sequence in the device xtrans array
};

#endif
