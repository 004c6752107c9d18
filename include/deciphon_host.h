/*
 * deciphon_host.h -- host orchestration in C over the HIP C-ABI (dcp_gpu.h).
 *
 * One header for the two call surfaces SURVEY.md §8(b) names: deciphon-old's own model / db /
 * server API for the scan path, and the subset of the (absent, third-party) imm API that
 * deciphon code and its tests call on that path.  Names, signatures, struct members that callers
 * touch, return codes and ownership follow the reference; the arithmetic (Viterbi scores and
 * paths) runs on the MI355X through dcp_gpu.h -- there is no CPU Viterbi behind these functions.
 *
 * Code written against the reference's headers builds against this one through the forwarding
 * headers in include/compat/ (`-Iinclude -Iinclude/compat`): the reference's own
 * test/protein_profile.c and test/protein_model.c compile unchanged and pass on the GPU
 * (tests/test_reference_tests.py).  Reference header -> section here:
 *
 *   include/deciphon/core/rc.h, limits.h, xmath.h, expect.h         -> "core"
 *   imm/imm.h (abc, seq, codon, rnd, lprob, dp, task, prod, path)    -> "imm subset"
 *   include/deciphon/core/lite_pack.h (lip_file, lip_read/write_*)   -> "lite_pack subset"
 *   include/deciphon/model/entry_dist.h, protein_cfg.h, protein_state.h, profile_typeid.h,
 *       nuclt_dist.h, protein_trans.h, protein_xtrans.h, protein_model.h, profile.h,
 *       protein_profile.h, standard_profile.h, standard_state.h, protein_codec.h,
 *       protein_h3reader.h                                           -> "model"
 *   include/deciphon/db/types.h, reader.h, writer.h, protein_reader.h, protein_writer.h,
 *       profile_reader.h                                             -> "db"
 *   src/server/prod.h, match.h, protein_match.h, hypothesis.h, scan_thread.h, scan.h
 *                                                                    -> "server"
 *
 * Deliberate differences: struct imm_dp / imm_task / imm_hmm internals are this
 * library's own (imm's are private to imm); a profile's emission tables live on the device, the
 * host object keeps the compact form (transitions + nuclt_dists); scan_run() without the REST
 * scheduler is scan_run_source() (sequences from a callback).
 */
#ifndef DECIPHON_HOST_H
#define DECIPHON_HOST_H

#include "dcp_gpu.h"
#include <assert.h>
#include <limits.h>
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ============================== core ========================================================= */
/* include/deciphon/core/rc.h:4-15 */
enum rc
{
    RC_OK,
    RC_END,
    RC_EFAIL,
    RC_EINVAL,
    RC_EIO,
    RC_ENOMEM,
    RC_EPARSE,
    RC_EAPI,
    RC_EHTTP,
};
char const *rc_string(enum rc rc);
#define RC_STRING(rc) rc_string(rc)

/* include/deciphon/core/limits.h */
enum limits
{
    BITS_PER_PROFILE_TYPEID = 16,
    MAX_NPROFILES = (1 << 20),
    NUM_THREADS = 64,
    PROFILE_ACC_SIZE = 32,
    PROTEIN_MODEL_CORE_SIZE_MAX = 4096,
};

/* include/deciphon/core/xmath.h (inline there, inline here) */
static inline unsigned xmath_min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned xmath_max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline unsigned xmath_ceildiv(unsigned x, unsigned y)
{
    assert(y > 0 && y - 1 <= UINT_MAX - x);
    return (x + y - 1) / y;
}
static inline unsigned xmath_partition_size(unsigned nelems, unsigned nparts, unsigned idx)
{
    unsigned const size = xmath_ceildiv(nelems, nparts);
    assert(nelems >= size * idx);
    return xmath_min(size, nelems - size * idx);
}
static inline float xmath_lrt_f32(float null_loglik, float alt_loglik) { return -2 * (null_loglik - alt_loglik); }
static inline double xmath_lrt_f64(double null_loglik, double alt_loglik) { return -2 * (null_loglik - alt_loglik); }
#ifndef __cplusplus
#define xmath_lrt(null, alt) _Generic((null), float : xmath_lrt_f32, double : xmath_lrt_f64)(null, alt)
#endif

/* ============================== imm subset ================================================== */
typedef float imm_float; /* the reference's default build (IMM_FLOAT_BYTES == 4) */
#define IMM_FLOAT_BYTES 4
#define IMM_STATE_NAME_SIZE 8
#define IMM_AMINO_SIZE 20
#define IMM_NUCLT_SIZE 4
#define IMM_LPROB_ZERO ((imm_float)-INFINITY)
#define IMM_LPROB_ONE ((imm_float)0)
#define IMM_ABC_MAX_SIZE 31
enum imm_rc
{
    IMM_OK = 0,
    IMM_FAILURE = 1,
};
static inline imm_float imm_log(imm_float x) { return logf(x); }
static inline bool imm_lprob_is_nan(imm_float x) { return isnan(x); }
static inline bool imm_lprob_is_finite(imm_float x) { return isfinite(x); }
static inline bool imm_lprob_is_zero(imm_float x) { return isinf(x) && x < 0; }

/* ---- alphabets ---- */
enum imm_abc_typeid
{
    IMM_NULL_ABC = 0,
    IMM_ABC = 1,
    IMM_AMINO = 2,
    IMM_NUCLT = 3,
    IMM_DNA = 4,
    IMM_RNA = 5,
};
struct imm_abc_vtable
{
    enum imm_abc_typeid typeid;
    void *derived;
};
struct imm_abc
{
    unsigned size;
    char symbols[IMM_ABC_MAX_SIZE + 1]; /* NUL-terminated */
    unsigned any_symbol_id;             /* == size */
    char any_symbol;
    struct imm_abc_vtable vtable;
};
struct imm_nuclt
{
    struct imm_abc super;
};
struct imm_amino
{
    struct imm_abc super;
};
struct imm_dna
{
    struct imm_nuclt super;
};
struct imm_rna
{
    struct imm_nuclt super;
};
extern struct imm_dna const imm_dna_iupac;     /* "ACGT", any 'X' */
extern struct imm_rna const imm_rna_iupac;     /* "ACGU", any 'X' */
extern struct imm_amino const imm_amino_iupac; /* "ACDEFGHIKLMNPQRSTVWY", any 'X' */
#define imm_super(x) (&(x)->super)
static inline enum imm_abc_typeid imm_abc_typeid(struct imm_abc const *abc) { return abc->vtable.typeid; }
static inline unsigned imm_abc_size(struct imm_abc const *abc) { return abc->size; }
static inline char const *imm_abc_symbols(struct imm_abc const *abc) { return abc->symbols; }
static inline unsigned imm_abc_any_symbol_id(struct imm_abc const *abc) { return abc->any_symbol_id; }
/* id of `symbol` in abc; any_symbol_id for the any-symbol; -1 if it is not in the alphabet */
int imm_abc_symbol_idx(struct imm_abc const *abc, char symbol);
char const *imm_abc_typeid_name(enum imm_abc_typeid typeid);

/* imm_code: the (position, length) -> word index of a sequence; here it only carries the alphabet
 * (words are coded on the device from 2-bit bases) */
struct imm_code
{
    struct imm_abc const *abc;
};
struct imm_nuclt_code
{
    struct imm_code super;
    struct imm_nuclt const *nuclt;
};
void imm_nuclt_code_init(struct imm_nuclt_code *code, struct imm_nuclt const *nuclt);

/* ---- strings / sequences ---- */
struct imm_str
{
    unsigned size;
    char const *data;
};
struct imm_str imm_str(char const *cstr);
#define IMM_STR(lit) ((struct imm_str){(unsigned)sizeof(lit) - 1, (lit)})
struct imm_seq
{
    unsigned size;
    char const *str;
    struct imm_abc const *abc;
};
struct imm_seq imm_seq(struct imm_str str, struct imm_abc const *abc);
static inline unsigned imm_seq_size(struct imm_seq const *seq) { return seq->size; }
struct imm_seq imm_subseq(struct imm_seq const *seq, unsigned start, unsigned size);

/* ---- codons / genetic code ---- */
struct imm_codon
{
    struct imm_nuclt const *nuclt;
    unsigned a, b, c; /* symbol ids; any_symbol_id = any */
};
struct imm_codon imm_codon(struct imm_nuclt const *nuclt, unsigned a, unsigned b, unsigned c);
struct imm_codon imm_codon_any(struct imm_nuclt const *nuclt);
struct imm_codon imm_codon_from_symbols(struct imm_nuclt const *nuclt, char const sym[3]);
#define IMM_CODON(nuclt, lit) imm_codon_from_symbols((nuclt), (lit))
char imm_codon_asym(struct imm_codon const *codon);
char imm_codon_bsym(struct imm_codon const *codon);
char imm_codon_csym(struct imm_codon const *codon);
char imm_gc_decode(unsigned table, struct imm_codon codon);

/* ---- random numbers / log-probability helpers (test/protein_model.c, protein_profile_sample) ---- */
struct imm_rnd
{
    uint64_t data[4];
};
struct imm_rnd imm_rnd(uint64_t seed);
double imm_rnd_dbl(struct imm_rnd *rnd);
void imm_lprob_sample(struct imm_rnd *rnd, unsigned len, imm_float *lprobs);
void imm_lprob_normalize(unsigned len, imm_float *lprobs);

/* ---- nucleotide / codon distributions (nuclt_dist members) ---- */
struct imm_nuclt_lprob
{
    struct imm_nuclt const *nuclt;
    imm_float lprobs[IMM_NUCLT_SIZE];
};
struct imm_codon_marg
{
    struct imm_nuclt const *nuclt;
    imm_float lprobs[IMM_NUCLT_SIZE + 1][IMM_NUCLT_SIZE + 1][IMM_NUCLT_SIZE + 1]; /* index 4 = any */
};
struct imm_frame_epsilon
{
    imm_float loge;
    imm_float log1e;
};
struct imm_frame_epsilon imm_frame_epsilon(imm_float epsilon);

/* ---- dp / task / prod / path ---- */
struct protein_profile;
struct lip_file;
/* The reference's imm_dp is imm's compiled HMM.  Here it is a handle on the profile that owns it:
 * the model's tables live in the profile (host: compact form; device: expanded), the 13
 * length-dependent special transitions are addressed through imm_dp_trans_idx /
 * imm_dp_change_trans exactly as protein_profile_setup does (src/model/protein_profile.c:186-214). */
struct imm_dp
{
    struct protein_profile *owner;
    int null_model; /* 1: null (R), 0: alt */
    struct imm_code const *code;
};
void imm_dp_init(struct imm_dp *dp, struct imm_code const *code);
void imm_dp_del(struct imm_dp const *dp);
/* index of the transition src -> dst among the special states' (state indices as the profile's
 * R / S,N,B,E,J,C,T members hold them); UINT_MAX if it is not one of the 13 */
unsigned imm_dp_trans_idx(struct imm_dp *dp, unsigned src_idx, unsigned dst_idx);
void imm_dp_change_trans(struct imm_dp *dp, unsigned trans_idx, imm_float lprob);
enum imm_rc imm_dp_pack(struct imm_dp const *dp, struct lip_file *file);
enum imm_rc imm_dp_unpack(struct imm_dp *dp, struct lip_file *file);
typedef unsigned imm_state_name(unsigned id, char name[IMM_STATE_NAME_SIZE]);
void imm_dp_write_dot(struct imm_dp const *dp, FILE *fp, imm_state_name *name);

struct imm_step
{
    uint16_t state_id;
    uint8_t seqlen;
};
struct imm_path
{
    struct imm_step *steps;
    unsigned nsteps, capacity;
};
struct imm_prod
{
    struct imm_path path;
    imm_float loglik;
    uint64_t mseconds;
};
struct imm_task
{
    struct imm_dp const *dp;
    struct imm_seq const *seq;
};
struct imm_task *imm_task_new(struct imm_dp const *dp);
enum imm_rc imm_task_reset(struct imm_task *task, struct imm_dp const *dp);
enum imm_rc imm_task_setup(struct imm_task *task, struct imm_seq const *seq);
void imm_task_del(struct imm_task const *task);
struct imm_prod imm_prod(void);
void imm_prod_reset(struct imm_prod *prod);
void imm_prod_del(struct imm_prod const *prod);
/* Viterbi score and path of task->seq under dp, computed on the MI355X with whatever special
 * transitions the profile currently holds (protein_profile_setup's, or the LOG1 = 0 defaults of
 * a profile that never saw a setup: src/model/protein_model.c:322-340, test/protein_db.c:73).
 * IMM_FAILURE without a HIP device, for an empty sequence, or for symbols outside the alphabet. */
enum imm_rc imm_dp_viterbi(struct imm_dp const *dp, struct imm_task *task, struct imm_prod *prod);
static inline unsigned imm_path_nsteps(struct imm_path const *path) { return path->nsteps; }
static inline struct imm_step const *imm_path_step(struct imm_path const *path, unsigned idx)
{
    assert(idx < path->nsteps);
    return path->steps + idx;
}
/* imm_del(x): imm's type-generic destructor */
#ifndef __cplusplus
#define imm_del(x)                                                                                                   \
    _Generic((x), struct imm_task * : imm_task_del, struct imm_task const * : imm_task_del,                           \
             struct imm_prod * : imm_prod_del, struct imm_prod const * : imm_prod_del,                                \
             struct imm_dp * : imm_dp_del, struct imm_dp const * : imm_dp_del)(x)
#endif

/* ============================== lite_pack subset (MessagePack file I/O) ======================= */
/* What deciphon's db code calls of EBI-Metagenomics/lite-pack 0.3.0 (absent): a MessagePack
 * stream over a FILE*.  Writers return false on failure; readers set file->error and return false.
 * Integers are written in the smallest MessagePack form; floats as float32; "1darray" is a
 * MessagePack ext (type = element type below) whose payload is the elements, big-endian. */
struct lip_file
{
    FILE *fp;
    bool error;
};
static inline void lip_file_init(struct lip_file *file, FILE *fp)
{
    file->fp = fp;
    file->error = false;
}
static inline FILE *lip_file_ptr(struct lip_file *file) { return file->fp; }
enum lip_1darray_type
{
    LIP_1DARRAY_UINT8 = 0x11,
    LIP_1DARRAY_UINT16 = 0x12,
    LIP_1DARRAY_UINT32 = 0x13,
    LIP_1DARRAY_F32 = 0x21,
};
bool lip_write_map_size(struct lip_file *file, unsigned size);
bool lip_write_array_size(struct lip_file *file, unsigned size);
bool lip_write_cstr(struct lip_file *file, char const *str);
bool lip_write_uint(struct lip_file *file, uint64_t val);
bool lip_write_f32(struct lip_file *file, float val);
bool lip_write_1darray_size_type(struct lip_file *file, unsigned size, uint8_t type);
bool lip_write_1darray_u32_item(struct lip_file *file, uint32_t item);
bool lip_write_1darray_f32_data(struct lip_file *file, unsigned size, float const *data);
bool lip_write_1darray_u8_data(struct lip_file *file, unsigned size, uint8_t const *data);
bool lip_read_map_size(struct lip_file *file, unsigned *size);
bool lip_read_array_size(struct lip_file *file, unsigned *size);
bool lip_read_str_size(struct lip_file *file, unsigned *size);
bool lip_read_str_data(struct lip_file *file, unsigned size, char *str);
bool lip_read_cstr(struct lip_file *file, unsigned size, char *str); /* str[size]: at most size-1 chars + NUL */
bool lip_read_uint(struct lip_file *file, uint64_t *val);
bool lip_read_f32(struct lip_file *file, float *val);
bool lip_read_1darray_size_type(struct lip_file *file, unsigned *size, enum lip_1darray_type *type);
bool lip_read_1darray_u32_data(struct lip_file *file, unsigned size, uint32_t *data);
bool lip_read_1darray_f32_data(struct lip_file *file, unsigned size, float *data);
bool lip_read_1darray_u8_data(struct lip_file *file, unsigned size, uint8_t *data);
/* skip one whole MessagePack object of any type (nested maps / arrays included) */
bool lip_skip_object(struct lip_file *file);
/* lip_write_int / lip_read_int / lip_write_float / lip_read_float: lite-pack's type-generic forms */
bool lip_read_unsigned(struct lip_file *file, unsigned *val);
bool lip_read_int_as_int(struct lip_file *file, int *val);
#ifndef __cplusplus
#define lip_write_int(file, val) lip_write_uint((file), (uint64_t)(val))
#define lip_write_float(file, val) lip_write_f32((file), (float)(val))
#define lip_read_int(file, ptr)                                                                                      \
    _Generic((ptr), unsigned * : lip_read_unsigned, default : lip_read_int_as_int)((file), (void *)(ptr))
#define lip_read_float(file, ptr) lip_read_f32((file), (ptr))
#endif
/* include/deciphon/core/expect.h */
bool expect_map_size(struct lip_file *file, unsigned size);
bool expect_map_key(struct lip_file *file, char const key[]);

enum imm_rc imm_abc_pack(struct imm_abc const *abc, struct lip_file *file);
enum imm_rc imm_abc_unpack(struct imm_abc *abc, struct lip_file *file);
enum imm_rc imm_nuclt_lprob_pack(struct imm_nuclt_lprob const *nucltp, struct lip_file *file);
enum imm_rc imm_nuclt_lprob_unpack(struct imm_nuclt_lprob *nucltp, struct lip_file *file);
enum imm_rc imm_codon_marg_pack(struct imm_codon_marg const *codonm, struct lip_file *file);
enum imm_rc imm_codon_marg_unpack(struct imm_codon_marg *codonm, struct lip_file *file);

/* ============================== model ======================================================= */
/* include/deciphon/model/entry_dist.h */
enum entry_dist
{
    ENTRY_DIST_NULL,
    ENTRY_DIST_UNIFORM,
    ENTRY_DIST_OCCUPANCY,
};
/* include/deciphon/model/protein_cfg.h */
struct protein_cfg
{
    enum entry_dist entry_dist;
    imm_float epsilon;
};
static inline struct protein_cfg protein_cfg(enum entry_dist entry_dist, imm_float epsilon)
{
    assert(epsilon >= 0.0f && epsilon <= 1.0f);
    struct protein_cfg cfg = {entry_dist, epsilon};
    return cfg;
}
#define DEFAULT_EPSILON ((imm_float)0.01)
#define PROTEIN_CFG_DEFAULT ((struct protein_cfg){ENTRY_DIST_OCCUPANCY, DEFAULT_EPSILON})

/* include/deciphon/model/protein_state.h:7-55 */
enum protein_state_id
{
    PROTEIN_MATCH_STATE = (0 << (BITS_PER_PROFILE_TYPEID - 2)),
    PROTEIN_INSERT_STATE = (1 << (BITS_PER_PROFILE_TYPEID - 2)),
    PROTEIN_DELETE_STATE = (2 << (BITS_PER_PROFILE_TYPEID - 2)),
    PROTEIN_EXT_STATE = (3 << (BITS_PER_PROFILE_TYPEID - 2)),
    PROTEIN_R_STATE = (PROTEIN_EXT_STATE | 0),
    PROTEIN_S_STATE = (PROTEIN_EXT_STATE | 1),
    PROTEIN_N_STATE = (PROTEIN_EXT_STATE | 2),
    PROTEIN_B_STATE = (PROTEIN_EXT_STATE | 3),
    PROTEIN_E_STATE = (PROTEIN_EXT_STATE | 4),
    PROTEIN_J_STATE = (PROTEIN_EXT_STATE | 5),
    PROTEIN_C_STATE = (PROTEIN_EXT_STATE | 6),
    PROTEIN_T_STATE = (PROTEIN_EXT_STATE | 7),
};
bool protein_state_is_match(unsigned id);
bool protein_state_is_insert(unsigned id);
bool protein_state_is_delete(unsigned id);
bool protein_state_is_mute(unsigned id);
unsigned protein_state_idx(unsigned id);
unsigned protein_state_name(unsigned id, char name[IMM_STATE_NAME_SIZE]);

/* include/deciphon/model/profile_typeid.h */
enum profile_typeid
{
    PROFILE_NULL,
    PROFILE_STANDARD,
    PROFILE_PROTEIN,
};
char const *profile_typeid_name(enum profile_typeid typeid);

/* include/deciphon/model/nuclt_dist.h */
struct nuclt_dist
{
    struct imm_nuclt_lprob nucltp;
    struct imm_codon_marg codonm;
};
static inline void nuclt_dist_init(struct nuclt_dist *nucltd, struct imm_nuclt const *nuclt)
{
    nucltd->nucltp.nuclt = nuclt;
    nucltd->codonm.nuclt = nuclt;
}
enum rc nuclt_dist_pack(struct nuclt_dist const *ndist, struct lip_file *file);
enum rc nuclt_dist_unpack(struct nuclt_dist *ndist, struct lip_file *file);

/* include/deciphon/model/protein_trans.h */
#define PROTEIN_TRANS_SIZE 7
struct protein_trans
{
    union
    {
        struct
        {
            imm_float MM, MI, MD, IM, II, DM, DD;
        };
        imm_float data[PROTEIN_TRANS_SIZE];
    };
};
/* include/deciphon/model/protein_xtrans.h */
struct protein_xtrans
{
    imm_float NN, CC, JJ, NB, CT, JB, RR, EJ, EC;
};
static inline void protein_xtrans_init(struct protein_xtrans *t)
{
    t->NN = t->NB = t->EC = t->CC = t->CT = t->EJ = t->JJ = t->JB = t->RR = IMM_LPROB_ONE;
}

/* include/deciphon/model/protein_model.h: the builder the HMMER3 reader, protein_profile_sample and
 * test/protein_model.c drive.  It collects the amino-acid level parameters; protein_profile_absorb
 * turns them into the scan-time profile (src/model/protein_model.c:49-500 semantics: lodds = match
 * - null, codon/nucleotide marginals, occupancy / uniform entry, exit overrides). */
struct protein_model
{
    struct imm_amino const *amino;
    struct imm_nuclt_code const *code;
    struct protein_cfg cfg;
    unsigned core_size;
    struct protein_xtrans xtrans;
    char consensus[PROTEIN_MODEL_CORE_SIZE_MAX + 1];
    struct
    {
        imm_float lprobs[IMM_AMINO_SIZE];
    } null;
    struct
    {
        unsigned node_idx;
        imm_float (*match_lprobs)[IMM_AMINO_SIZE]; /* [core_size] as given to add_node */
        unsigned trans_idx;
        struct protein_trans *trans; /* [core_size + 1] */
    } alt;
};
void protein_model_init(struct protein_model *m, struct imm_amino const *amino, struct imm_nuclt_code const *code,
                        struct protein_cfg cfg, imm_float const null_lprobs[IMM_AMINO_SIZE]);
enum rc protein_model_setup(struct protein_model *m, unsigned core_size);
enum rc protein_model_add_node(struct protein_model *m, imm_float const lp[IMM_AMINO_SIZE], char consensus);
enum rc protein_model_add_trans(struct protein_model *m, struct protein_trans trans);
void protein_model_del(struct protein_model const *m);
struct imm_amino const *protein_model_amino(struct protein_model const *m);
struct imm_nuclt const *protein_model_nuclt(struct protein_model const *m);

/* include/deciphon/model/profile.h */
struct profile;
struct profile_vtable
{
    int typeid;
    void (*del)(struct profile *prof);
    enum rc (*unpack)(struct profile *prof, struct lip_file *file);
    struct imm_dp const *(*null_dp)(struct profile const *prof);
    struct imm_dp const *(*alt_dp)(struct profile const *prof);
};
struct profile
{
    struct profile_vtable vtable;
    char accession[PROFILE_ACC_SIZE];
    imm_state_name *state_name;
    struct imm_code const *code;
};
void profile_init(struct profile *prof, char const *accession, struct imm_code const *code,
                  struct profile_vtable vtable, imm_state_name *state_name);
void profile_del(struct profile *prof);
enum rc profile_unpack(struct profile *prof, struct lip_file *file);
int profile_typeid(struct profile const *prof);
struct imm_dp const *profile_null_dp(struct profile const *prof);
struct imm_dp const *profile_alt_dp(struct profile const *prof);

/* include/deciphon/model/protein_profile.h:12-43.  The members before `impl` are the reference's;
 * the rest is this library's (device-facing compact profile, current special transitions). */
struct protein_profile
{
    struct profile super;
    struct imm_amino const *amino;
    struct imm_nuclt_code const *code;
    struct protein_cfg cfg;
    struct imm_frame_epsilon eps;
    unsigned core_size;
    char consensus[PROTEIN_MODEL_CORE_SIZE_MAX + 1];
    struct
    {
        struct nuclt_dist ndist;
        struct imm_dp dp;
        unsigned R;
    } null;
    struct
    {
        struct nuclt_dist *match_ndists;
        struct nuclt_dist insert_ndist;
        struct imm_dp dp;
        unsigned S, N, B, E, J, C, T;
    } alt;
    dcp_profile *impl;           /* owned */
    imm_float xtrans[DCP_NXTRANS]; /* RR, SB, SN, NN, NB, ET, EC, CC, CT, EB, EJ, JJ, JB */
};
void protein_profile_init(struct protein_profile *prof, char const *accession, struct imm_amino const *amino,
                          struct imm_nuclt_code const *code, struct protein_cfg cfg);
enum rc protein_profile_setup(struct protein_profile *prof, unsigned seq_size, bool multi_hits, bool hmmer3_compat);
enum rc protein_profile_absorb(struct protein_profile *prof, struct protein_model const *model);
enum rc protein_profile_sample(struct protein_profile *prof, unsigned seed, unsigned core_size);
enum rc protein_profile_decode(struct protein_profile const *prof, struct imm_seq const *seq, unsigned state_id,
                               struct imm_codon *codon);
void protein_profile_write_dot(struct protein_profile const *prof, FILE *fp);
enum rc protein_profile_unpack(struct protein_profile *prof, struct lip_file *file);
enum rc protein_profile_pack(struct protein_profile const *prof, struct lip_file *file);
/* protein_model_{init,setup,add_node,add_trans} + protein_profile_absorb in one call:
 * null_lprobs[20], match_lprobs[core_size][20], trans[core_size+1][7] (MM MI MD IM II DM DD). */
enum rc protein_profile_from_params(struct protein_profile *prof, unsigned core_size,
                                    imm_float const *null_lprobs, imm_float const *match_lprobs,
                                    imm_float const *trans);

/* include/deciphon/model/standard_profile.h, standard_state.h: dead in the reference's scan path
 * (profile_reader_setup accepts PROFILE_PROTEIN only, src/db/profile_reader.c:95-98); kept as the
 * typed shell the API promises.  Its imm_dp hold no model: imm_dp_viterbi on them fails; pack /
 * unpack write / read the map(2) {"null", "alt"} of src/model/standard_profile.c:20-43 with empty
 * dp values. */
struct standard_profile
{
    struct profile super;
    struct
    {
        struct imm_dp null;
        struct imm_dp alt;
    } dp;
};
void standard_profile_init(struct standard_profile *prof, char const *accession, struct imm_code const *code);
enum rc standard_profile_unpack(struct standard_profile *prof, struct lip_file *file);
enum rc standard_profile_pack(struct standard_profile const *prof, struct lip_file *file);
unsigned standard_state_name(unsigned id, char name[IMM_STATE_NAME_SIZE]);

/* include/deciphon/model/protein_codec.h */
struct protein_codec
{
    unsigned idx;
    unsigned start;
    struct protein_profile const *prof;
    struct imm_path const *path;
};
static inline struct protein_codec protein_codec_init(struct protein_profile const *prof, struct imm_path const *path)
{
    struct protein_codec codec = {0, 0, prof, path};
    return codec;
}
enum rc protein_codec_next(struct protein_codec *codec, struct imm_seq const *seq, struct imm_codon *codon);

/* include/deciphon/model/protein_h3reader.h over this library's own HMMER3 ASCII parser
 * (dcp_h3reader_*; the reference parses with the absent `hmr`): every next() leaves the profile's
 * parameters in reader->model, ready for protein_profile_absorb (src/server/hmm.c:120-178). */
struct protein_h3reader
{
    dcp_h3reader *impl;
    imm_float null_lprobs[IMM_AMINO_SIZE];
    struct protein_model model;
    char name[64], acc[PROFILE_ACC_SIZE];
};
void protein_h3reader_init(struct protein_h3reader *reader, struct imm_amino const *amino,
                           struct imm_nuclt_code const *code, struct protein_cfg cfg, FILE *fp);
enum rc protein_h3reader_next(struct protein_h3reader *reader);
void protein_h3reader_del(struct protein_h3reader const *reader);

/* ============================== db ========================================================== */
/* include/deciphon/db/types.h */
enum db_typeid
{
    DB_NULL,
    DB_STANDARD,
    DB_PROTEIN,
};
#define MAGIC_NUMBER 0xC6F0

/* include/deciphon/db/reader.h */
struct db_reader
{
    unsigned nprofiles;
    uint32_t *profile_sizes;
    enum profile_typeid profile_typeid;
    struct lip_file file;
};
enum rc db_reader_open(struct db_reader *db, FILE *fp);
void db_reader_close(struct db_reader *db);
enum rc db_reader_unpack_magic_number(struct db_reader *db);
enum rc db_reader_unpack_profile_typeid(struct db_reader *db, enum profile_typeid typeid);
enum rc db_reader_unpack_float_size(struct db_reader *db);
enum rc db_reader_unpack_profile_sizes(struct db_reader *db);

/* include/deciphon/db/protein_reader.h */
struct protein_db_reader
{
    struct db_reader super;
    struct imm_amino amino;
    struct imm_nuclt nuclt;
    struct imm_nuclt_code code;
    struct protein_cfg cfg;
};
enum rc protein_db_reader_open(struct protein_db_reader *db, FILE *fp);

/* include/deciphon/db/writer.h */
typedef enum rc (*pack_profile_func_t)(struct lip_file *file, void const *arg);
typedef enum rc (*pack_header_item_func_t)(struct lip_file *file, void const *arg);
struct db_writer
{
    unsigned nprofiles;
    unsigned header_size;
    struct lip_file file;
    struct
    {
        struct lip_file header;
        struct lip_file profile_sizes;
        struct lip_file profiles;
    } tmp;
};
enum rc db_writer_open(struct db_writer *db, FILE *fp);
enum rc db_writer_close(struct db_writer *db, bool successfully);
enum rc db_writer_pack_magic_number(struct db_writer *db);
enum rc db_writer_pack_profile_typeid(struct db_writer *db, int profile_typeid);
enum rc db_writer_pack_float_size(struct db_writer *db);
enum rc db_writer_pack_profile(struct db_writer *db, pack_profile_func_t pack_profile, void const *arg);
enum rc db_writer_pack_header_item(struct db_writer *db, pack_header_item_func_t pack_header_item, void const *arg);

/* include/deciphon/db/protein_writer.h */
struct protein_db_writer
{
    struct db_writer super;
    struct imm_amino amino;
    struct imm_nuclt nuclt;
    struct imm_nuclt_code code;
    struct protein_cfg cfg;
};
enum rc protein_db_writer_open(struct protein_db_writer *db, FILE *fp, struct imm_amino const *amino,
                               struct imm_nuclt const *nuclt, struct protein_cfg cfg);
enum rc protein_db_writer_pack_profile(struct protein_db_writer *db, struct protein_profile const *profile);

/* include/deciphon/db/profile_reader.h:9-37 */
struct profile_reader
{
    unsigned npartitions;
    unsigned partition_size[NUM_THREADS];
    int64_t partition_offset[NUM_THREADS + 1];
    struct lip_file file[NUM_THREADS];
    enum profile_typeid profile_typeid;
    union
    {
        struct protein_profile pro;
    } profiles[NUM_THREADS];
    /* this library's: first profile index of each partition (hit records -> DB index) */
    unsigned partition_first[NUM_THREADS + 1];
    /* this library's: the database's profile_sizes[] (borrowed from the db_reader, which outlives the reader
     * as in the reference) -- byte offsets of single profiles, so that a partition can be unpacked by several
     * host threads at once */
    uint32_t const *profile_sizes;
};
enum rc profile_reader_setup(struct profile_reader *reader, struct db_reader *db, unsigned npartitions);
/* Same, but the contiguous partitions are balanced by profile BYTES (proportional to core size, i.e.
 * to DP cells) instead of by profile count: one partition per GPU wants equal work, and the
 * reference's count balance is a known imbalance for mixed sizes.  Offsets and sizes
 * obey the same invariants as profile_reader_setup's. */
enum rc profile_reader_setup_balanced(struct profile_reader *reader, struct db_reader *db, unsigned npartitions);
unsigned profile_reader_npartitions(struct profile_reader const *reader);
unsigned profile_reader_partition_size(struct profile_reader const *reader, unsigned partition);
unsigned profile_reader_nprofiles(struct profile_reader const *reader);
enum rc profile_reader_rewind_all(struct profile_reader *reader);
enum rc profile_reader_rewind(struct profile_reader *reader, unsigned partition);
enum rc profile_reader_next(struct profile_reader *reader, unsigned partition, struct profile **profile);
bool profile_reader_end(struct profile_reader *reader, unsigned partition);
void profile_reader_del(struct profile_reader *reader);

/* ============================== server ====================================================== */
/* sched/structs.h sizes used by src/server/prod.h (deciphon-sched 0.4.2, absent: values from
 * SURVEY.md Appendix C; only the field widths of struct prod depend on them) */
enum
{
    SCHED_PROFILE_NAME_SIZE = 64,
    SCHED_ABC_NAME_SIZE = 16,
    SCHED_PROFILE_TYPEID_SIZE = 16,
    SCHED_VERSION_SIZE = 16,
};
#define DECIPHON_VERSION "0.1.0"

/* src/server/match.h, protein_match.h */
struct match
{
    struct imm_step const *step;
    struct imm_seq const *frag;
    struct profile const *profile;
};
static inline void match_setup(struct match *match, struct profile const *profile)
{
    match->step = 0;
    match->frag = 0;
    match->profile = profile;
}
struct protein_match
{
    struct match match;
};
/* "frag,state,codon,amino" of one path step (src/server/protein_match.c:21-56) */
enum rc protein_match_write_func(FILE *fp, void const *match);

/* src/server/prod.h.  The match column is written straight to the file, as the reference does. */
struct prod
{
    int64_t id;
    int64_t scan_id;
    int64_t seq_id;
    char profile_name[SCHED_PROFILE_NAME_SIZE];
    char abc_name[SCHED_ABC_NAME_SIZE];
    double alt_loglik;
    double null_loglik;
    char profile_typeid[SCHED_PROFILE_TYPEID_SIZE];
    char version[SCHED_VERSION_SIZE];
};
typedef enum rc (*prod_fwrite_match_func_t)(FILE *fp, void const *match);
enum rc prod_fopen(unsigned nthreads);
void prod_setup_job(struct prod *prod, char const *abc_name, char const *prof_typeid, int64_t scan_id);
void prod_setup_seq(struct prod *prod, int64_t seq_id);
enum rc prod_fwrite(struct prod const *prod, struct imm_seq const *seq, struct imm_path const *path,
                    unsigned thread_num, prod_fwrite_match_func_t fwrite_match, struct match *match);
void prod_fcleanup(void);
enum rc prod_fclose(void);
FILE *prod_final_fp(void);
char const *prod_final_path(void);
void prod_final_cleanup(void);
/* Header line of the products file (src/server/prod.c:119-121). */
char const *prod_header(void);

/* src/server/hypothesis.h */
struct hypothesis
{
    struct imm_task *task;
    struct imm_prod prod;
};

/* src/server/scan_thread.h:8-43.  One scan thread = one partition = one device context; the members
 * after write_match_func are this library's. */
struct scan_thread
{
    unsigned id;
    int64_t job_id;
    struct imm_seq const *seq;
    struct profile_reader *reader;
    bool multi_hits;
    bool hmmer3_compat;
    double lrt_threshold;
    struct prod prod;
    struct hypothesis null;
    struct hypothesis alt;
    union
    {
        struct protein_match pro;
    } match;
    prod_fwrite_match_func_t write_match_func;
    /* device side: the partition's profiles are unpacked once and stay resident between sequences */
    dcp_gpu_ctx *gpu;
    bool db_resident;
    dcp_profile **impls; /* [nimpls]: the partition's compact profiles (decode / product rows) */
    unsigned nimpls;
};
void thread_init(struct scan_thread *t, unsigned id, struct profile_reader *reader, bool multi_hits,
                 bool hmmer3_compat, double lrt_threshold, prod_fwrite_match_func_t write_match_func);
void thread_setup_job(struct scan_thread *t, enum imm_abc_typeid abc_typeid, enum profile_typeid profile_typeid,
                      int64_t scan_id);
void thread_setup_seq(struct scan_thread *t, struct imm_seq *seq, int64_t seq_id);
/* For the thread's sequence and partition: every profile scored null + alt on the device, LRT
 * filter, and for each hit the alt path and its product row through prod_fwrite
 * (src/server/scan_thread.c:86-135).  tid selects the HIP device (tid % device count). */
enum rc thread_run(struct scan_thread *t, int tid);
/* The same for `nseqs` prefetched sequences in ONE device pass: the reference
 * fetches and scans one sequence at a time (src/server/scan.c:227-258), which cannot fill a GPU.
 * Products come out ordered by (sequence, profile). seq_ids[i] is the id of seqs[i]. */
enum rc thread_run_batch(struct scan_thread *t, int tid, struct imm_seq const *seqs, int64_t const *seq_ids,
                         unsigned nseqs);
void thread_cleanup(struct scan_thread *t);

/* src/server/scan.h: scan_run(job_id, num_threads) pulls its scan, database and sequences from the
 * REST scheduler (src/sched, out of scope).  scan_run_source is the same loop (src/server/scan.c:215-269)
 * with those three supplied by the caller: the pressed database file, the scan's flags, and a
 * callback that yields the next sequence (RC_OK + *seq, RC_END when exhausted -- api_scan_next_seq's
 * contract).  `batch` sequences are prefetched per device pass (1 = the reference's loop shape).
 * The products file (header + every thread's rows, thread by thread: prod.c:106-145) is left in
 * prod_final_fp() / prod_final_path(); release it with prod_final_cleanup(). */
struct scan_seq
{
    int64_t id;
    char const *data; /* NUL-terminated; valid until the next call of the callback */
};
typedef enum rc (*scan_next_seq_func_t)(void *arg, struct scan_seq *seq);
struct scan_cfg
{
    int64_t scan_id;
    bool multi_hits;
    bool hmmer3_compat;
    double lrt_threshold;  /* scan.c:221 passes 10.0 */
    unsigned batch;        /* sequences per device pass; 0 = 1 */
    bool balance_by_cells; /* partitions by profile bytes (one per GPU) instead of by count */
    /* Leave the database resident (device contexts + unpacked profiles of every partition) when the scan
     * ends: the next scan_run_source of the SAME file (device, inode, size, mtime), partition count and
     * balance picks it up and skips unpacking and upload -- what a server that polls jobs in one process
     * wants (20k profiles: 1.5 s per job).  Device memory stays allocated until scan_resident_release()
     * or a scan of another database. */
    bool keep_resident;
    /* Batched progress (the reference consumes one unit per (profile, sequence) pair inside thread_run,
     * src/core/progress.c:69-94, scan.c:97-102): called once per device pass and partition with the number of
     * pairs that pass completed, possibly from several host threads at once -- the callee serialises
     * (integration/scan_run_adapter.c turns it into api_increment_job_progress).  NULL = no reporting. */
    void (*progress)(unsigned long pairs, void *arg);
    void *progress_arg;
    /* A device pass is sized by WORK, not only by count: its target is `batch` sequences or batch_symbols bases
     * (0 = no such bound), whichever comes first -- at least one sequence.  The cells of a pass are (sum of the
     * partition's core sizes) x (symbols of the pass), so with sequences of 100 nt .. 10 kbp a count-sized pass varies
     * a hundredfold in device time and a short one cannot fill the query-lane kernels' lanes (BASELINE configs[4]
     * "dynamic batching").  Passes are cut from a look-ahead queue filled until it exceeds one and a half targets or
     * the source ends: if the source ends within one and a half targets the queue is ONE pass (a job never ends in a
     * sliver of a pass -- with mixed lengths that costs as much device time as a full one), otherwise the pass is the
     * shortest prefix that reaches a target.  batch = 1 keeps the reference's loop shape.  Within a pass the device
     * packs the sequences by length itself (include/dcp_gpu.h, dcp_plan_query_slots); product rows keep the source's
     * order. */
    unsigned long batch_symbols;
};
/* Where the last scan_run_source spent its time (not in the reference; profiles/host_scan_probe.c prints it): host
 * seconds summed over the partitions' threads.  scan_wait_s is the wait for the device scans (the host has nothing
 * else to do then); trace_s the hit fetch + the traceback of the hits' paths (device, the scan stream idle); rows_s the
 * product rows (host; all but the last pass's run under the next pass's scan). */
struct scan_stats
{
    unsigned passes;
    unsigned long hits, steps;
    double load_s, submit_s, scan_wait_s, trace_s, rows_s;
};
void scan_last_stats(struct scan_stats *out);
void scan_resident_release(void);
enum rc scan_run_source(char const *db_filename, struct scan_cfg cfg, unsigned num_threads,
                        scan_next_seq_func_t next_seq, void *arg);
/* scan_run_source over an in-memory sequence list, products copied to `prods`. */
enum rc scan_run_local(char const *db_filename, struct scan_seq const *seqs, unsigned nseqs, unsigned num_threads,
                       bool multi_hits, bool hmmer3_compat, double lrt_threshold, int64_t scan_id, unsigned batch,
                       FILE *prods);

#ifdef __cplusplus
}
#endif
#endif
