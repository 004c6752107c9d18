/*
 * oracle/oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A from-scratch, plain-C restatement of the reference's profile-HMM scan path:
 *   deciphon-old: src/model/protein_model.c, src/model/protein_profile.c,
 *                 src/server/scan_thread.c:86-135, include/deciphon/core/xmath.h
 *   third-party:  EBI-Metagenomics/imm v2.0.3 (pinned at CMakeLists.txt:16 of the
 *                 reference; NOT present in /root/reference) -- its published
 *                 algorithm (frame-state emission, generic max-plus Viterbi,
 *                 logaddexp, RNG) is restated here from its public description.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library, and only as the checker. The product (deciphon-old_amd/) never
 * links or calls it.
 *
 * PARITY PIN: this oracle reproduces the reference's own known-answer tests
 * test/protein_profile.c:41,65,157 (goldens G1-G3 of SURVEY.md §8c) -- see
 * tests/test_oracle_goldens.py.  G4/G5 need inputs that are not in the tree.
 *
 * Precision: compile with -DORC_F64 for imm's IMM_DOUBLE_PRECISION build
 * (double), default is float32 (imm's default imm_float).
 */
#ifndef DCP_ORACLE_H
#define DCP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef ORC_F64
typedef double ofloat;
#else
typedef float ofloat;
#endif

#ifdef __cplusplus
extern "C" {
#endif

enum
{
    ORC_AMINO_SIZE = 20,
    ORC_NUCLT_SIZE = 4,
    ORC_TRANS_SIZE = 7,   /* MM MI MD IM II DM DD: protein_trans.h:8-27 */
    ORC_NCODES = 1364,    /* words of length 1..5 over ACGT */
    ORC_CORE_SIZE_MAX = 4096, /* limits.h:11 */
};

/* enum rc mirror: include/deciphon/core/rc.h:4-15 */
enum
{
    ORC_OK = 0,
    ORC_END = 1,
    ORC_EFAIL = 2,
    ORC_EINVAL = 3,
    ORC_EIO = 4,
    ORC_ENOMEM = 5,
};

/* entry_dist.h:4-9 */
enum
{
    ORC_ENTRY_DIST_NULL = 0,
    ORC_ENTRY_DIST_UNIFORM = 1,
    ORC_ENTRY_DIST_OCCUPANCY = 2,
};

/* protein_state.h:7-21 */
enum
{
    ORC_MATCH_STATE = (0 << 14),
    ORC_INSERT_STATE = (1 << 14),
    ORC_DELETE_STATE = (2 << 14),
    ORC_EXT_STATE = (3 << 14),
    ORC_R_STATE = (3 << 14) | 0,
    ORC_S_STATE = (3 << 14) | 1,
    ORC_N_STATE = (3 << 14) | 2,
    ORC_B_STATE = (3 << 14) | 3,
    ORC_E_STATE = (3 << 14) | 4,
    ORC_J_STATE = (3 << 14) | 5,
    ORC_C_STATE = (3 << 14) | 6,
    ORC_T_STATE = (3 << 14) | 7,
};

int orc_float_bytes(void);

/* ---- RNG (imm_rnd restatement: xoshiro256+ / splitmix64, pinned by G1) -- */
struct orc_rnd
{
    uint64_t s[4];
};
void orc_rnd_seed(struct orc_rnd *r, uint64_t seed);
uint64_t orc_rnd_u64(struct orc_rnd *r);
double orc_rnd_dbl(struct orc_rnd *r);

/* ---- lprob helpers ------------------------------------------------------- */
ofloat orc_logaddexp(ofloat a, ofloat b);
void orc_lprob_normalize(unsigned n, ofloat *arr);
void orc_lprob_sample(struct orc_rnd *r, unsigned n, ofloat *arr);

/* ---- nucleotide distribution of one state (nuclt_dist.h:7-11) ------------ */
struct orc_nuclt_dist
{
    ofloat nucltp[4];   /* base log-probs */
    ofloat codonm[125]; /* 5x5x5 codon marginals, index 4 = wildcard */
};
/* protein_model.c:396-408 */
void orc_setup_nuclt_dist(struct orc_nuclt_dist *d, ofloat const aa_lprobs[20]);
/* imm frame-state emission table over all 1364 words */
void orc_frame_table(struct orc_nuclt_dist const *d, ofloat eps, ofloat *tbl);
/* code of a word: offsets 0,4,20,84,340 (length 1..5) + base-4 value,
 * first base most significant. x[] holds symbol ids 0..3. */
unsigned orc_word_code(unsigned char const *x, unsigned len);

/* ---- profile ------------------------------------------------------------- */
struct orc_profile;

/* protein_profile_sample (protein_profile.c:259-304) */
struct orc_profile *orc_profile_sample(unsigned seed, unsigned core_size,
                                       int entry_dist, ofloat epsilon);
/* protein_model_init/setup/add_node/add_trans + protein_profile_absorb given
 * explicit parameters. match_lprobs: [M][20], trans: [M+1][7]. */
struct orc_profile *orc_profile_new(unsigned core_size, int entry_dist,
                                    ofloat epsilon,
                                    ofloat const null_lprobs[20],
                                    ofloat const *match_lprobs,
                                    ofloat const *trans);
void orc_profile_del(struct orc_profile *p);
unsigned orc_profile_core_size(struct orc_profile const *p);
unsigned orc_profile_nstates(struct orc_profile const *p, int alt);

/* protein_profile_setup (protein_profile.c:155-216). Returns ORC_EINVAL for
 * seq_size == 0. */
int orc_profile_setup(struct orc_profile *p, unsigned seq_size, int multi_hits,
                      int hmmer3_compat);

/* Parameter read-back, used to feed the SAME numbers to the device path in
 * DP-level (bit-exact) tests.
 *   trans8: [8][M] rows = entry, MM, IM, DM, MD, DD, MI, II as seen by node k
 *           (k's incoming edges from node k-1, and k's own MI/II);
 *   emis_match: [1364][M]; emis_insert, emis_null: [1364];
 *   xtrans: 13 scalars RR, SB, SN, NN, NB, ET, EC, CC, CT, EB, EJ, JJ, JB. */
void orc_profile_export(struct orc_profile const *p, ofloat *trans8,
                        ofloat *emis_match, ofloat *emis_insert,
                        ofloat *emis_null, ofloat *xtrans);
void orc_profile_dists(struct orc_profile const *p,
                       struct orc_nuclt_dist *null_d,
                       struct orc_nuclt_dist *insert_d,
                       struct orc_nuclt_dist *match_d /* [M] */);

/* ---- Viterbi ------------------------------------------------------------- */
/* seq: symbol ids 0..3 (A,C,G,T), length L.
 * Generic (imm_dp_viterbi-like) max-plus Viterbi over the state graph with
 * traceback. alt=0 -> null model (end state R), alt=1 -> alt model (end T).
 * path_state/path_len may be NULL; otherwise capacity *nsteps on input. */
int orc_viterbi(struct orc_profile const *p, int alt, unsigned char const *seq,
                unsigned L, ofloat *loglik, uint16_t *path_state,
                uint8_t *path_len, unsigned *nsteps);

/* Score of a given path (validates it is a path of the model graph); NaN if it is not. */
ofloat orc_path_score(struct orc_profile const *p, int alt, unsigned char const *seq, unsigned L,
                      uint16_t const *path_state, uint8_t const *path_len, unsigned nsteps);

/* Score-only, end-indexed recursion of SURVEY Appendix B (the formulation the
 * HIP kernels implement); same float association as orc_viterbi. */
int orc_viterbi_fast(struct orc_profile const *p, unsigned char const *seq,
                     unsigned L, ofloat *null_loglik, ofloat *alt_loglik);

/* Score-only DP on raw tables (no orc_profile): what the device computes from
 * the product's own tables. All arrays float32/64 per build.
 *   trans8 [8][ldk], emis_match [1364][ldk] (ldk >= M). */
int orc_dp_tables(unsigned M, unsigned ldk, ofloat const *trans8,
                  ofloat const *emis_match, ofloat const *emis_insert,
                  ofloat const *emis_null, ofloat const *xtrans,
                  unsigned char const *seq, unsigned L, ofloat *null_loglik,
                  ofloat *alt_loglik);

/* xmath_lrt (xmath.h:32-43) */
ofloat orc_lrt(ofloat null_loglik, ofloat alt_loglik);

/* imm_frame_cond_decode restatement: most likely codon of a 1..5 nt fragment.
 * state_id picks the distribution as protein_profile_decode does
 * (protein_profile.c:306-331). Returns lprob, writes codon[3] (ids 0..3). */
ofloat orc_profile_decode(struct orc_profile const *p, unsigned char const *frag,
                          unsigned len, unsigned state_id,
                          unsigned char codon[3]);

/* joint log p(fragment, codon) for one given codon (what the decode maximises) */
ofloat orc_profile_codon_lprob(struct orc_profile const *p, unsigned char const *frag,
                               unsigned len, unsigned state_id, unsigned char const codon[3]);

/* protein_state_name (protein_state.c:5-39) */
unsigned orc_state_name(unsigned id, char name[8]);

/* thread_run restatement (scan_thread.c:86-135) used as the CPU baseline:
 * for every (sequence, profile) pair: setup -> null Viterbi -> alt Viterbi ->
 * LRT. Profiles split over `nthreads` contiguous count-balanced partitions
 * (profile_reader.c:54-72) under OpenMP (scan.c:239). mode 0 = generic graph
 * Viterbi (reference-faithful), mode 1 = end-indexed fast recursion.
 * out_null/out_alt: [nseqs][nprofiles]. Returns number of hits (lrt >= thr). */
long orc_scan(struct orc_profile *const *profiles, unsigned nprofiles,
              unsigned char const *seqs, uint32_t const *seq_off,
              unsigned nseqs, int multi_hits, int hmmer3_compat, double lrt_thr,
              int nthreads, int mode, ofloat *out_null, ofloat *out_alt);

/* The 13 special transitions protein_profile_setup writes (protein_profile.c:155-216), in the
 * order RR, SB, SN, NN, NB, ET, EC, CC, CT, EB, EJ, JJ, JB. ORC_EINVAL for seq_size == 0. */
int orc_xtrans(unsigned seq_size, int multi_hits, int hmmer3_compat, ofloat xt[13]);

/* Optimised CPU variant: DB resident -- tables exported once per profile --,
 * null score once per (sequence, distinct null table), no allocation in the pair loop, every
 * partition's thread runs its profiles over all sequences (no barrier per sequence).  Scores are
 * bit-identical to orc_dp_tables / orc_scan mode 1.  prepare_seconds = the per-DB table export,
 * dp_seconds = the pair loop (what a resident engine pays per batch). Returns hits or -1. */
long orc_scan_resident(struct orc_profile *const *profiles, unsigned nprofiles,
                       unsigned char const *seqs, uint32_t const *seq_off, unsigned nseqs,
                       int multi_hits, int hmmer3_compat, double lrt_thr, int nthreads,
                       ofloat *out_null, ofloat *out_alt, double *prepare_seconds, double *dp_seconds);

/* ---- file readers (oracle_io.c): the oracle's OWN parsers of the formats either side of the scan path,
 * so that parity tests compare the HIP path with the oracle reading the same bytes --- */
/* Swiss-Prot 50.8 background, src/model/protein_h3reader.c:79-103 */
void orc_swissprot_null(ofloat out[20]);
/* HMMER3/f ASCII -> profiles (protein_h3reader_next + protein_profile_absorb) */
struct orc_h3;
struct orc_h3 *orc_h3_open(char const *path, int entry_dist, ofloat eps);
/* ORC_OK + *out (caller frees with orc_profile_del), ORC_END, ORC_EFAIL (malformed), ORC_EINVAL (core size) */
int orc_h3_next(struct orc_h3 *, struct orc_profile **out);
char const *orc_h3_error(struct orc_h3 const *);
char const *orc_h3_acc(struct orc_h3 const *);       /* ACC, or NAME without one: of the last profile read */
char const *orc_h3_consensus(struct orc_h3 const *); /* CONS column of the last profile read */
void orc_h3_close(struct orc_h3 *);
/* MessagePack ".dcp" database (src/db/writer.c:95-117, src/model/protein_profile.c:338-400) */
struct orc_dcp;
struct orc_dcp *orc_dcp_open(char const *path, char *err, size_t errcap);
void orc_dcp_close(struct orc_dcp *);
unsigned orc_dcp_nprofiles(struct orc_dcp const *);
int orc_dcp_entry_dist(struct orc_dcp const *);
ofloat orc_dcp_epsilon(struct orc_dcp const *);
int orc_dcp_profile(struct orc_dcp const *, unsigned i, unsigned *core_size, char *acc, ofloat *trans8,
                    ofloat *xtrans, struct orc_nuclt_dist *null_d, struct orc_nuclt_dist *insert_d,
                    struct orc_nuclt_dist *match_d, char *consensus);
int orc_dcp_score(struct orc_dcp const *, unsigned i, unsigned char const *seq, unsigned L, int multi_hits,
                  int hmmer3_compat, ofloat *null_loglik, ofloat *alt_loglik);

#ifdef __cplusplus
}
#endif
#endif
