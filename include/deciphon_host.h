/*
 * deciphon_host.h -- host orchestration in C over the HIP C-ABI (dcp_gpu.h).
 *
 * Keeps the reference's own names, argument meaning and error behaviour for the
 * scan path, so code written against deciphon-old's model/db/server headers (and
 * tests shaped like test/protein_profile.c) compiles against this header and
 * runs its Viterbi on the MI355X:
 *
 *   include/deciphon/core/rc.h, limits.h, xmath.h          -> enum rc, limits, xmath_*
 *   include/deciphon/model/protein_cfg.h, entry_dist.h     -> struct protein_cfg, protein_cfg()
 *   include/deciphon/model/protein_state.h                 -> PROTEIN_*_STATE, protein_state_*
 *   include/deciphon/model/profile.h, profile_typeid.h     -> struct profile, profile_*()
 *   include/deciphon/model/protein_profile.h               -> protein_profile_{init,sample,setup,decode}
 *   include/deciphon/model/protein_codec.h                 -> protein_codec_{init,next}
 *   include/deciphon/db/profile_reader.h                   -> profile_reader_* (resident DB, not a file)
 *   src/server/scan_thread.h, hypothesis.h, prod.h         -> thread_{init,setup_job,setup_seq,run}
 *   imm (third party, absent): only the calls deciphon makes on this path
 *       imm_seq / imm_str / imm_subseq / imm_task_* / imm_dp_viterbi / imm_prod* / imm_path_*
 *
 * Everything that computes a score or a path goes to the device through
 * dcp_gpu.h; there is no CPU Viterbi behind these functions.  Differences from the
 * reference are listed in DESIGN.md §1 (no lite_pack/.dcp I/O: profiles come from
 * protein_profile_sample or protein_profile_from_params; products are collected in
 * memory instead of per-thread tmp files).
 */
#ifndef DECIPHON_HOST_H
#define DECIPHON_HOST_H

#include "dcp_gpu.h"
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- core ----------------------------------------------------------------- */
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

enum limits
{
    BITS_PER_PROFILE_TYPEID = 16,
    MAX_NPROFILES = (1 << 20),
    NUM_THREADS = 64,
    PROFILE_ACC_SIZE = 32,
    PROTEIN_MODEL_CORE_SIZE_MAX = 4096,
};

typedef float imm_float; /* IMM_FLOAT_BYTES == 4: the reference's default build */
#define IMM_STATE_NAME_SIZE 8
enum imm_rc
{
    IMM_OK = 0,
    IMM_FAILURE = 1,
};

unsigned xmath_partition_size(unsigned nelems, unsigned nparts, unsigned idx);
float xmath_lrt_f32(float null_loglik, float alt_loglik);
#define xmath_lrt(null, alt) xmath_lrt_f32((float)(null), (float)(alt))

/* ---- alphabets / sequences (imm subset) --------------------------------------------- */
enum imm_abc_typeid
{
    IMM_NULL_ABC = 0,
    IMM_ABC = 1,
    IMM_AMINO = 2,
    IMM_NUCLT = 3,
    IMM_DNA = 4,
    IMM_RNA = 5,
};
struct imm_abc
{
    enum imm_abc_typeid typeid;
    char const *symbols;
    char any_symbol;
};
struct imm_nuclt
{
    struct imm_abc super;
};
struct imm_amino
{
    struct imm_abc super;
};
struct imm_nuclt_code
{
    struct imm_abc const *abc;   /* what profile.code->abc gives the tests */
    struct imm_nuclt const *nuclt;
};
extern struct imm_nuclt const imm_dna_iupac;   /* "ACGT", any 'X' */
extern struct imm_amino const imm_amino_iupac; /* "ACDEFGHIKLMNPQRSTVWY", any 'X' */
#define imm_super(x) (&(x)->super)
void imm_nuclt_code_init(struct imm_nuclt_code *code, struct imm_nuclt const *nuclt);
unsigned imm_abc_any_symbol_id(struct imm_abc const *abc);
char const *imm_abc_typeid_name(enum imm_abc_typeid typeid);

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
unsigned imm_seq_size(struct imm_seq const *seq);
struct imm_seq imm_subseq(struct imm_seq const *seq, unsigned start, unsigned size);

struct imm_codon
{
    struct imm_nuclt const *nuclt;
    unsigned a, b, c; /* symbol ids; 4 = any */
};
struct imm_codon imm_codon(struct imm_nuclt const *nuclt, unsigned a, unsigned b, unsigned c);
struct imm_codon imm_codon_from_symbols(struct imm_nuclt const *nuclt, char const sym[3]);
#define IMM_CODON(nuclt, lit) imm_codon_from_symbols((nuclt), (lit))
char imm_codon_asym(struct imm_codon const *codon);
char imm_codon_bsym(struct imm_codon const *codon);
char imm_codon_csym(struct imm_codon const *codon);
char imm_gc_decode(unsigned table, struct imm_codon codon);

/* ---- dp / task / prod (imm subset; the arithmetic is on the device) -------------------- */
struct protein_profile;
struct imm_dp
{
    struct protein_profile *owner;
    int null_model; /* 1: null (R), 0: alt */
};
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
/* Viterbi score and path of task->seq under dp, computed on the MI355X
 * (src/server/scan_thread.c:69-76). Fails if no HIP device is present. */
enum imm_rc imm_dp_viterbi(struct imm_dp const *dp, struct imm_task *task, struct imm_prod *prod);
unsigned imm_path_nsteps(struct imm_path const *path);
struct imm_step const *imm_path_step(struct imm_path const *path, unsigned idx);
bool imm_lprob_is_finite(imm_float x);

/* ---- protein model layer ------------------------------------------------------------- */
enum entry_dist
{
    ENTRY_DIST_NULL,
    ENTRY_DIST_UNIFORM,
    ENTRY_DIST_OCCUPANCY,
};
struct protein_cfg
{
    enum entry_dist entry_dist;
    imm_float epsilon;
};
struct protein_cfg protein_cfg(enum entry_dist entry_dist, imm_float epsilon);
#define DEFAULT_EPSILON ((imm_float)0.01)
#define PROTEIN_CFG_DEFAULT ((struct protein_cfg){ENTRY_DIST_OCCUPANCY, DEFAULT_EPSILON})

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

enum profile_typeid
{
    PROFILE_NULL,
    PROFILE_STANDARD,
    PROFILE_PROTEIN,
};
char const *profile_typeid_name(enum profile_typeid typeid);

typedef unsigned imm_state_name(unsigned id, char name[IMM_STATE_NAME_SIZE]);
struct profile;
struct profile_vtable
{
    int typeid;
    void (*del)(struct profile *prof);
    struct imm_dp const *(*null_dp)(struct profile const *prof);
    struct imm_dp const *(*alt_dp)(struct profile const *prof);
};
struct profile
{
    struct profile_vtable vtable;
    char accession[PROFILE_ACC_SIZE];
    imm_state_name *state_name;
    struct imm_nuclt_code const *code;
};
void profile_del(struct profile *prof);
int profile_typeid(struct profile const *prof);
struct imm_dp const *profile_null_dp(struct profile const *prof);
struct imm_dp const *profile_alt_dp(struct profile const *prof);

struct protein_profile
{
    struct profile super;
    struct imm_amino const *amino;
    struct imm_nuclt_code const *code;
    struct protein_cfg cfg;
    unsigned core_size;
    struct
    {
        struct imm_dp dp;
        unsigned R;
    } null;
    struct
    {
        struct imm_dp dp;
        unsigned S, N, B, E, J, C, T;
    } alt;
    /* device-facing compact profile and the last protein_profile_setup() */
    dcp_profile *impl;
    unsigned seq_size;
    bool multi_hits, hmmer3_compat;
};

/* include/deciphon/model/standard_profile.h:10-23, standard_state.h:6: the generic (non-protein)
 * profile is just two imm_dp.  It is dead in the reference's scan path -- profile_reader_setup
 * accepts PROFILE_PROTEIN only (src/db/profile_reader.c:95-98) and the union member is commented
 * out (include/deciphon/db/profile_reader.h:18) -- so it is kept as the typed shell the API
 * promises: init, typeid, state names.  Its imm_dp hold no model: imm_dp_viterbi on them fails.
 * (standard_profile_pack writes lite_pack, which is outside this path.) */
struct standard_profile
{
    struct profile super;
    struct
    {
        struct imm_dp null;
        struct imm_dp alt;
    } dp;
};
void standard_profile_init(struct standard_profile *prof, char const *accession,
                           struct imm_nuclt_code const *code);
unsigned standard_state_name(unsigned id, char name[IMM_STATE_NAME_SIZE]);

void protein_profile_init(struct protein_profile *prof, char const *accession,
                          struct imm_amino const *amino, struct imm_nuclt_code const *code,
                          struct protein_cfg cfg);
enum rc protein_profile_setup(struct protein_profile *prof, unsigned seq_size, bool multi_hits,
                              bool hmmer3_compat);
enum rc protein_profile_sample(struct protein_profile *prof, unsigned seed, unsigned core_size);
/* protein_model_{init,setup,add_node,add_trans} + protein_profile_absorb in one call:
 * null_lprobs[20], match_lprobs[core_size][20], trans[core_size+1][7] (MM MI MD IM II DM DD). */
enum rc protein_profile_from_params(struct protein_profile *prof, unsigned core_size,
                                    imm_float const *null_lprobs, imm_float const *match_lprobs,
                                    imm_float const *trans);
enum rc protein_profile_decode(struct protein_profile const *prof, struct imm_seq const *seq,
                               unsigned state_id, struct imm_codon *codon);

struct protein_codec
{
    unsigned idx;
    unsigned start;
    struct protein_profile const *prof;
    struct imm_path const *path;
};
struct protein_codec protein_codec_init(struct protein_profile const *prof, struct imm_path const *path);
enum rc protein_codec_next(struct protein_codec *codec, struct imm_seq const *seq, struct imm_codon *codon);

/* ---- db: partitioned reader over a resident profile set --------------------------------- */
struct protein_db
{
    unsigned nprofiles;
    struct protein_profile **profiles;
};
struct profile_reader
{
    unsigned npartitions;
    unsigned partition_size[NUM_THREADS];
    unsigned partition_begin[NUM_THREADS + 1];
    unsigned cursor[NUM_THREADS];
    struct protein_db const *db;
};
enum rc profile_reader_setup(struct profile_reader *reader, struct protein_db const *db, unsigned npartitions);
unsigned profile_reader_npartitions(struct profile_reader const *reader);
unsigned profile_reader_partition_size(struct profile_reader const *reader, unsigned partition);
unsigned profile_reader_nprofiles(struct profile_reader const *reader);
enum rc profile_reader_rewind_all(struct profile_reader *reader);
enum rc profile_reader_rewind(struct profile_reader *reader, unsigned partition);
enum rc profile_reader_next(struct profile_reader *reader, unsigned partition, struct profile **profile);

/* ---- server: one scan thread = one partition = one device context -------------------------- */
struct prod
{
    int64_t scan_id, seq_id;
    char profile_name[64];
    char abc_name[16];
    double alt_loglik, null_loglik;
    char profile_typeid[16];
    char version[16];
};
struct scan_thread
{
    unsigned id;
    struct imm_seq const *seq;
    struct profile_reader *reader;
    bool multi_hits;
    bool hmmer3_compat;
    double lrt_threshold;
    struct prod prod;
    /* device side: the partition's profiles stay resident between sequences */
    dcp_gpu_ctx *gpu;
    bool db_resident;
    /* product rows of this thread (prod_fwrite output), grown as needed */
    char *rows;
    size_t rows_len, rows_cap;
    unsigned nprods;
};
void thread_init(struct scan_thread *t, unsigned id, struct profile_reader *reader, bool multi_hits,
                 bool hmmer3_compat, double lrt_threshold);
void thread_setup_job(struct scan_thread *t, enum imm_abc_typeid abc_typeid,
                      enum profile_typeid profile_typeid, int64_t scan_id);
void thread_setup_seq(struct scan_thread *t, struct imm_seq *seq, int64_t seq_id);
/* For the thread's sequence and partition: every profile scored null + alt on the device, LRT
 * filter, and for each hit the alt path and its product row (src/server/scan_thread.c:86-135).
 * tid selects the HIP device (tid % device count). */
enum rc thread_run(struct scan_thread *t, int tid);
/* The same for a batch of `nseqs` prefetched sequences in ONE device pass: the
 * reference fetches and scans one sequence at a time (src/server/scan.c:227-258), which cannot fill
 * a GPU. Products come out ordered by (sequence, profile). seq_ids[i] is the id of seqs[i]. */
enum rc thread_run_batch(struct scan_thread *t, int tid, struct imm_seq const *seqs, int64_t const *seq_ids,
                         unsigned nseqs);
void thread_cleanup(struct scan_thread *t);

/* scan_run without the scheduler (src/server/scan.c:215-269): the sequences come from the caller instead of
 * api_scan_next_seq, the products file is written to `prods` (header + every thread's rows, thread by
 * thread, as prod_fclose concatenates them: src/server/prod.c:106-145). num_threads partitions = host
 * threads (OpenMP, schedule(static,1): scan.c:239) = device contexts; `batch` sequences per device pass. */
struct scan_seq
{
    int64_t id;
    char const *data;
};
enum rc scan_run_local(struct protein_db const *db, struct scan_seq const *seqs, unsigned nseqs,
                       unsigned num_threads, bool multi_hits, bool hmmer3_compat, double lrt_threshold,
                       int64_t scan_id, unsigned batch, FILE *prods);
/* Header line of the products file (src/server/prod.c:119-121). */
char const *prod_header(void);

#ifdef __cplusplus
}
#endif
#endif
