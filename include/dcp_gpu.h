/*
 * dcp_gpu.h -- C-ABI of the MI355X-native profile-HMM scan engine.
 *
 * This is the drop-in boundary for deciphon-old's scan hot path
 * (SURVEY.md §8b). Plain C: pointers, sizes, POD structs; no C++/torch types.
 * Citations are `path:line` in the reference tree (EBI-Metagenomics/deciphon-old).
 *
 * What each group replaces:
 *   dcp_profile_*        protein_profile_{init,sample,absorb,setup} and the model
 *                        builder it calls   src/model/protein_profile.c:134-304,
 *                                           src/model/protein_model.c:49-500
 *   dcp_gpu_db_*         profile_reader_{setup,rewind,next} + protein_profile.unpack:
 *                        the DB is uploaded ONCE and stays resident in HBM instead
 *                        of being re-read per sequence
 *                                           src/db/profile_reader.c:74-168,
 *                                           src/model/protein_profile.c:38-117
 *   dcp_gpu_seqs_*       imm_seq()/imm_task_setup(): sequence encoding, once per
 *                        batch instead of once per (seq, profile) pair
 *                                           src/server/scan.c:229,
 *                                           src/server/scan_thread.c:51-55
 *   dcp_gpu_scan*        thread_run's per-pair body: protein_profile_setup,
 *                        imm_dp_viterbi(null), imm_dp_viterbi(alt), xmath_lrt filter
 *                                           src/server/scan_thread.c:99-123,
 *                                           include/deciphon/core/xmath.h:32-43
 *
 * Errors: every function that can fail returns `enum dcp_rc`, numerically equal
 * to the reference's `enum rc` (include/deciphon/core/rc.h:4-15); HIP failures map
 * to DCP_EFAIL and the message is kept in dcp_gpu_last_error().
 * There is NO CPU fallback: without a HIP device dcp_gpu_ctx_new() fails.
 */
#ifndef DCP_GPU_H
#define DCP_GPU_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/deciphon/core/rc.h:4-15 */
enum dcp_rc
{
    DCP_OK = 0,
    DCP_END = 1,
    DCP_EFAIL = 2,
    DCP_EINVAL = 3,
    DCP_EIO = 4,
    DCP_ENOMEM = 5,
    DCP_EPARSE = 6,
    DCP_EAPI = 7,
    DCP_EHTTP = 8,
};

/* include/deciphon/model/entry_dist.h:4-9 */
enum dcp_entry_dist
{
    DCP_ENTRY_DIST_NULL = 0,
    DCP_ENTRY_DIST_UNIFORM = 1,
    DCP_ENTRY_DIST_OCCUPANCY = 2,
};

enum
{
    DCP_AMINO_SIZE = 20,          /* IMM_AMINO_SIZE */
    DCP_TRANS_SIZE = 7,           /* PROTEIN_TRANS_SIZE, protein_trans.h:6 */
    DCP_NCODES = 1364,            /* nucleotide words of length 1..5 */
    DCP_NDIST = 129,              /* nuclt_dist: 4 base lprobs + 5x5x5 codon marg */
    DCP_CORE_SIZE_MAX = 4096,     /* PROTEIN_MODEL_CORE_SIZE_MAX, limits.h:11 */
    DCP_PROFILE_ACC_SIZE = 32,    /* limits.h:9 */
    DCP_NUM_THREADS = 64,         /* limits.h:8: max partitions */
    DCP_NXTRANS = 13,
};

/* ------------------------------------------------------------------------ */
/* Host-side compact profile ("dcpx"): everything protein_profile carries     */
/* that the scan needs, without the per-state emission tables (those are      */
/* expanded on the device at upload).                                         */
/* ------------------------------------------------------------------------ */
typedef struct dcp_profile dcp_profile;

/* protein_profile_init + protein_model_init/setup/add_node/add_trans +
 * protein_profile_absorb (protein_profile.c:134-153,218-257;
 * protein_model.c:49-137,155-184).
 *   null_lprobs [20], match_lprobs [core_size][20] amino log-probs in
 *   imm_amino_iupac order "ACDEFGHIKLMNPQRSTVWY"; trans [core_size+1][7] in
 *   protein_trans order MM MI MD IM II DM DD (protein_trans.h:8-27).
 * Returns NULL (and DCP_EINVAL in *rc) for core_size == 0 or > 4096,
 * as protein_model_setup does (protein_model.c:157-160). */
dcp_profile *dcp_profile_new(char const *accession, unsigned core_size,
                             int entry_dist, float epsilon,
                             float const *null_lprobs,
                             float const *match_lprobs, float const *trans,
                             char const *consensus, int *rc);

/* protein_profile_sample (protein_profile.c:259-304): imm_rnd(seed) stream,
 * log-uniform lprobs, normalised; DD=-inf at i=0, MD=DD=-inf at i=core_size. */
dcp_profile *dcp_profile_sample(char const *accession, unsigned seed,
                                unsigned core_size, int entry_dist,
                                float epsilon, int *rc);
/* A profile from its scan-time parts (what a pressed DB stores: protein_profile.c:338-400):
 * trans8 [8][core_size] rows entry(B->Mk), MM, IM, DM, MD, DD (edges INTO node k), MI, II;
 * null / insert dists [129], match dists [core_size][129] (4 base lprobs + 5x5x5 codon marginals).
 * NULL + DCP_EINVAL for core_size outside 1..4096, epsilon outside [0,1] or a NaN value. */
dcp_profile *dcp_profile_from_parts(char const *accession, unsigned core_size, int entry_dist, float epsilon,
                                    char const *consensus, float const *trans8, float const *null_dist,
                                    float const *insert_dist, float const *match_dist, int *rc);
void dcp_profile_del(dcp_profile *);
unsigned dcp_profile_core_size(dcp_profile const *);
int dcp_profile_entry_dist(dcp_profile const *);
float dcp_profile_epsilon(dcp_profile const *);
/* imm_rnd(seed) / its next double in [0,1) (xoshiro256+ seeded by splitmix64: pinned by the
 * reference's golden test/protein_profile.c:41), and imm_lprob_normalize. */
void dcp_rnd_seed(uint64_t state[4], uint64_t seed);
double dcp_rnd_next(uint64_t state[4]);
void dcp_lprob_normalize(unsigned n, float *lprobs);
char const *dcp_profile_accession(dcp_profile const *);

/* ---- HMMER3 ASCII reader: .hmm -> profiles, what protein_h3reader_next +
 * protein_profile_absorb do in hmm_press (src/model/protein_h3reader.c:18-72,
 * src/server/hmm.c:120-178). The reference parses with the third-party `hmr` library (absent);
 * this is an own parser of the published HMMER3/f text format: per profile the node-0 transition
 * line gives trans[0]; every node line gives 20 match -ln(p) values (`*` = probability 0) and its
 * CONS letter, the next line (insert emissions) is ignored as the reference ignores it, the third
 * gives trans[k]. The null model is the hard-coded Swiss-Prot 50.8 amino frequencies
 * (protein_h3reader.c:79-103); the profile's accession is the ACC field (hmm.c:113-117), NAME if
 * there is none. */
typedef struct dcp_h3reader dcp_h3reader;
dcp_h3reader *dcp_h3reader_open(char const *path, int entry_dist, float epsilon);
/* Same over a stream the caller keeps owning (protein_h3reader_init takes a FILE*). */
dcp_h3reader *dcp_h3reader_open_fp(FILE *fp, int entry_dist, float epsilon);
/* The next profile's raw parameters instead of a built profile (what protein_h3reader_next leaves in
 * its protein_model): match_lprobs [core_size][20], trans [core_size+1][7]; pointers stay valid
 * until the next call on this reader. Same return codes as dcp_h3reader_next. */
struct dcp_h3params
{
    unsigned core_size;
    float const *match_lprobs;
    float const *trans;
    char const *consensus;
    char const *name;
    char const *acc;
};
int dcp_h3reader_next_params(dcp_h3reader *, struct dcp_h3params *out);
/* DCP_OK and *out (caller owns it), DCP_END at end of file, DCP_EPARSE on malformed input
 * (message in dcp_h3reader_error), DCP_EINVAL for core sizes outside 1..4096. */
int dcp_h3reader_next(dcp_h3reader *, dcp_profile **out);
char const *dcp_h3reader_error(dcp_h3reader const *);
void dcp_h3reader_close(dcp_h3reader *);
/* log of the Swiss-Prot 50.8 background frequencies, imm_amino_iupac order */
void dcp_swissprot_null_lprobs(float out[DCP_AMINO_SIZE]);
char const *dcp_profile_consensus(dcp_profile const *);

/* ---- dcpx: compact on-disk profile DB (press once, scan many) ------------------------------
 * Own little-endian container for what the scan needs (transitions + nuclt_dists; no per-state
 * emission tables: the device expands them at upload). It is NOT the reference's .dcp: that is
 * MessagePack with opaque imm_dp blobs whose format lives in the absent imm library. The header
 * carries the reference's fields and its checks (src/db/protein_reader.c:40-82, src/db/reader.c:25-79):
 * magic 0xC6F0 (include/deciphon/db/types.h:11), profile_typeid == PROFILE_PROTEIN, float_size == 4,
 * entry_dist in {UNIFORM, OCCUPANCY}, 0 <= epsilon <= 1, alphabets, profile_sizes[] (u32 bytes per
 * profile). One (entry_dist, epsilon) per DB, as protein_db_writer_open fixes it
 * (src/db/protein_writer.c:56-96). */
typedef struct dcp_db dcp_db;
int dcp_db_write(char const *path, dcp_profile *const *profiles, unsigned nprofiles);
/* NULL + *rc: DCP_EIO (cannot open / truncated), DCP_EINVAL (bad magic, typeid, float size,
 * entry_dist, epsilon, too many profiles) */
dcp_db *dcp_db_open(char const *path, int *rc);
void dcp_db_close(dcp_db *);
unsigned dcp_db_nprofiles(dcp_db const *);
int dcp_db_entry_dist(dcp_db const *);
float dcp_db_epsilon(dcp_db const *);
uint32_t const *dcp_db_profile_sizes(dcp_db const *);
/* profile_reader_setup's partition table for this file (src/db/profile_reader.c:45-72): count-balanced
 * contiguous partitions, part_offset[i] = byte offset in the file where partition i starts,
 * the end of the last non-empty partition = end of the profiles. ceil-sized partitions can leave
 * trailing empty ones whose end offset the reference never writes (0): reproduced as is.
 * Returns npart = min(npartitions, nprofiles); 0 on EINVAL. */
unsigned dcp_db_partitions(dcp_db const *, unsigned npartitions, unsigned part_size[DCP_NUM_THREADS],
                           int64_t part_offset[DCP_NUM_THREADS + 1]);
/* Profiles [begin, end) (caller owns them). */
int dcp_db_read(dcp_db *, unsigned begin, unsigned end, dcp_profile **out);

/* Read-only views (host memory owned by the profile):
 *   trans8 [8][core_size]: rows entry(B->Mk), MM, IM, DM, MD, DD (edges INTO
 *   node k from node k-1) and MI, II (node k's own insert edges);
 *   dists: null [129], insert [129], match [core_size][129]. */
float const *dcp_profile_trans8(dcp_profile const *);
float const *dcp_profile_null_dist(dcp_profile const *);
float const *dcp_profile_insert_dist(dcp_profile const *);
float const *dcp_profile_match_dist(dcp_profile const *);

/* Frame-state emission table of one nuclt_dist over all 1364 words (host).
 * out[code], code = {0,4,20,84,340}[len-1] + base-4 value of the word, first
 * base most significant. Same formula the device expansion kernel evaluates. */
void dcp_frame_table_host(float const dist[DCP_NDIST], float epsilon,
                          float out[DCP_NCODES]);

/* protein_profile_setup (protein_profile.c:155-216): the 13 length-dependent
 * special transitions RR, SB, SN, NN, NB, ET, EC, CC, CT, EB, EJ, JJ, JB.
 * DCP_EINVAL for seq_size == 0 (:158). */
int dcp_xtrans(unsigned seq_size, int multi_hits, int hmmer3_compat,
               float out[DCP_NXTRANS]);

/* xmath_lrt_f32 (xmath.h:32-35) */
float dcp_lrt(float null_loglik, float alt_loglik);

/* profile_reader partitioning (profile_reader.c:54-72, xmath.h:24-30):
 * contiguous count-balanced partitions. part_size[npart], returns npart =
 * min(npartitions, nprofiles) or 0 on EINVAL (0 or > 64 partitions). */
unsigned dcp_partition_by_count(unsigned nprofiles, unsigned npartitions,
                                unsigned part_size[DCP_NUM_THREADS]);
/* MI355X shard map: contiguous partitions balanced by sum of core sizes
 * (cells), one per GPU. part_begin[npart+1]. */
void dcp_partition_by_cells(unsigned const *core_sizes, unsigned nprofiles,
                            unsigned npartitions, unsigned *part_begin);

/* ------------------------------------------------------------------------ */
/* Device context                                                            */
/* ------------------------------------------------------------------------ */
typedef struct dcp_gpu_ctx dcp_gpu_ctx;

int dcp_gpu_device_count(void);
/* Fails (NULL) when no HIP device is present: there is no CPU fallback. */
dcp_gpu_ctx *dcp_gpu_ctx_new(int device);
void dcp_gpu_ctx_del(dcp_gpu_ctx *);
char const *dcp_gpu_last_error(dcp_gpu_ctx const *);
/* The HIP stream every launch of this context goes to (hipStream_t). */
void *dcp_gpu_stream(dcp_gpu_ctx *);

/* Upload `nprofiles` profiles; expands every match/insert/null frame-state
 * emission table into HBM (layout: DESIGN.md §3).  `flags` (0 for a server):
 *   DCP_DB_EXPAND_ON_HOST  the tables are computed with dcp_frame_table_host and
 *                          copied (slow; parity tests).  (The argument used to be
 *                          `int expand_on_host`: 0 / 1 keep their meaning.)
 *   DCP_DB_ONE_LAYOUT      keep ONE layout of the match tables resident -- the row
 *                          sweep's [1364][ldk] -- and let the query-lane kernels
 *                          gather their 8-node tile images from it, instead of
 *                          holding those images as a second copy: half the
 *                          footprint (20.6 instead of 40.2 GB per 20 000 Pfam-like
 *                          profiles), same bits.  Chosen by the library itself when
 *                          both layouts would not leave 16 GiB of the device's
 *                          free memory.
 * Replaces any previously uploaded DB.  (The reference re-reads the profiles from
 * the .dcp file for every sequence: src/server/scan.c:227-258, profile_reader.c.) */
#define DCP_DB_EXPAND_ON_HOST 1
#define DCP_DB_ONE_LAYOUT 2
int dcp_gpu_db_upload(dcp_gpu_ctx *, dcp_profile *const *profiles,
                      unsigned nprofiles, int flags);
unsigned dcp_gpu_db_nprofiles(dcp_gpu_ctx const *);
/* 1 when the resident DB holds one table layout (asked for or chosen). */
int dcp_gpu_db_one_layout(dcp_gpu_ctx const *);
/* Bytes of expanded match tables resident now (the row-sweep layout of a
 * two-layout DB appears with the first scan that needs it). */
uint64_t dcp_gpu_db_table_bytes(dcp_gpu_ctx const *);
/* Copy profile p's expanded match table back: out [1364][core_size]. */
int dcp_gpu_db_fetch_match_table(dcp_gpu_ctx *, unsigned p, float *out);

/* Upload a batch of sequences. seqs: concatenated symbol ids 0..3 (A,C,G,T);
 * seq_off[nseqs+1]. Any id > 3 or an empty sequence -> DCP_EINVAL
 * (protein_profile.c:158; imm rejects symbols outside the alphabet). */
int dcp_gpu_seqs_upload(dcp_gpu_ctx *, uint8_t const *seqs,
                        uint32_t const *seq_off, unsigned nseqs);
/* Same from ASCII "ACGT" text (what scan.c:229 hands to imm_seq). */
int dcp_gpu_seqs_upload_text(dcp_gpu_ctx *, char const *text,
                             uint32_t const *seq_off, unsigned nseqs);
unsigned dcp_gpu_nseqs(dcp_gpu_ctx const *);
/* Explicit special transitions for the resident sequences: xt [nseqs][13] in dcp_xtrans order.
 * By default a scan derives them from each sequence's length and the scan's flags (what
 * protein_profile_setup does per pair); imm_dp_viterbi on a profile whose transitions were set some
 * other way -- or never: the LOG1 defaults of protein_model.c:322-340, as test/protein_db.c:73 runs
 * it -- passes them here.  They stay in force until the next sequence upload; the scan flags
 * multi_hits / hmmer3_compat are then ignored.  NaN values are rejected (DCP_EINVAL). */
int dcp_gpu_seqs_set_xtrans(dcp_gpu_ctx *, float const *xt, unsigned nseqs);

struct dcp_scan_params
{
    int multi_hits;      /* scan_thread.h:17 */
    int hmmer3_compat;   /* scan_thread.h:18 */
    float lrt_threshold; /* scan.c:221 passes 10.0 */
    int keep_scores;     /* also keep dense null/alt score matrices */
    int kernel;          /* 0 = choose by a cost model (DB size x batch size); 1 = row sweep (one wavefront
                          * group per pair: any batch size); 2 = query lane (one lane per query, tiles in
                          * LDS: throughput path); 3 = query lane, two-stage blocks (even / odd tiles of a
                          * profile pipelined through an LDS ring: half the scratch traffic) */
};

/* scan_thread.c:121-123 keeps a pair iff lrt is finite and >= threshold */
struct dcp_hit
{
    uint32_t seq_idx;
    uint32_t profile_idx;
    float null_loglik;
    float alt_loglik;
};

/* Enqueue the scan of all resident sequences against all resident profiles on
 * the context's stream (asynchronous). */
int dcp_gpu_scan(dcp_gpu_ctx *, struct dcp_scan_params const *);
/* One scan is outstanding per context: results are those of the last scan enqueued; enqueueing
 * another one first completes the previous (it does not queue behind unread results).
 * Same for the resident sequences [q_begin, q_end): scan.c:227-258 hands
 * thread_run one sequence at a time; a caller that prefetched N sequences scans
 * them batch by batch without re-uploading. Hits and scores keep the indices
 * of the resident batch. */
int dcp_gpu_scan_range(dcp_gpu_ctx *, struct dcp_scan_params const *,
                       unsigned q_begin, unsigned q_end);
/* Let the scan write its hit records and hit count into caller-owned DEVICE
 * memory (cap records of struct dcp_hit, one uint32 counter) -- e.g. a buffer
 * that RCCL then gathers. NULL, 0, NULL restores the internal
 * buffer. */
int dcp_gpu_set_hit_buffer(dcp_gpu_ctx *, void *hits_dev, unsigned cap,
                           void *nhits_dev);
/* The device memory the last scan wrote its hit records and counter to (the context's own buffer, or the
 * caller's from dcp_gpu_set_hit_buffer) -- what a C host hands to dcp_dist_gather_hits without touching
 * the HIP API itself.  Call dcp_gpu_sync first: a query-lane scan completes its redo pairs there. */
int dcp_gpu_hit_buffer(dcp_gpu_ctx *, void **hits_dev, void **nhits_dev, unsigned *cap);
#ifdef DCP_TEST_HOOKS
/* NOT in the shipped library: exists only in libdcp_hip_testhooks.so (the same sources compiled with
 * -DDCP_TEST_HOOKS, loaded by the tests alone).  Shrinks the per-size-class capacity of the redo lists
 * (default 2^26 pairs; 0 restores it) so a test can reach the overflow path.  Results are unaffected:
 * an overflowed scan is repeated with the row-sweep kernel. */
int dcp_gpu_test_set_redo_cap(dcp_gpu_ctx *, unsigned cap);
/* Same build only.  on != 0: in the next two-stage query-lane scans, stage 0 of the first task sits out its first
 * step, so its partner stage runs into the bound of the LDS ring's hand-shake: the kernel must drain and
 * dcp_gpu_sync must return DCP_EFAIL (never a hang). */
int dcp_gpu_test_set_ring_stall(dcp_gpu_ctx *, int on);
/* Same build only.  Cap in bytes on the boundary columns the segmented row sweep keeps per size class (0 restores the
 * default, 6 GiB): a small cap makes it sweep a class's queries chunk by chunk. */
int dcp_gpu_test_set_seg_col_bytes(dcp_gpu_ctx *, unsigned long long bytes);
/* Same build only.  own_forward != 0: dcp_gpu_trace_paths fills the hits' work areas with the trace kernel's own
 * one-wavefront forward loop (rounds 1-3) instead of the row-sweep kernels' -- the tests' second implementation of
 * the same rows; budget_floats != 0: floats of work area per round of launches (default 2^31), so that a handful of
 * hits already takes several rounds. */
int dcp_gpu_test_set_trace_mode(dcp_gpu_ctx *, int own_forward, unsigned long long budget_floats);
/* Same build only.  Forces the grid-mode row-sweep kernel variant -- leading emission rows a block stages in
 * LDS (0, 20 or 84) and, in `block_waves`: bits 0..7 wavefronts per block (0: the default), bits 8..15 KiB of
 * unused LDS per block (an occupancy experiment), bit 16 the two-rows-ahead prefetch variant, bits 20..23 the
 * one size class (nodes per lane) to force, 0 = all, bits 24..25 the segmented sweep of the classes of more than
 * 512 nodes (1 never, 2 always, 0 the library's rule), bits 26..27 likewise the K-profiles-per-wavefront kernel of
 * the classes of at most 128 nodes -- instead of the one the library picks per size class and
 * batch size; stage < 0 restores the automatic choice.  A class without such a kernel keeps the automatic
 * one.  Every variant computes the same scores: tests/test_gpu_parity.py runs them all against the oracle. */
int dcp_gpu_test_set_rowsweep_variant(dcp_gpu_ctx *, int stage_rows, unsigned block_waves);
#endif
/* Wait for the stream (and, after a query-lane scan, check its redo lists:
 * see dcp_gpu_last_scan_redo_pairs). */
int dcp_gpu_sync(dcp_gpu_ctx *);
/* Query-lane scans with multi_hits score a pair under B(j) = N(j) + NB and
 * verify that against the E -> B / J -> B feedback; pairs that fail the check
 * are re-scored exactly by the row-sweep kernel behind it on the same stream.
 * Their number in the last scan (0 for row-sweep and uni-hit scans);
 * synchronises. */
int dcp_gpu_last_scan_redo_pairs(dcp_gpu_ctx *, unsigned *npairs);
/* Milliseconds between HIP events recorded on the context's stream around the
 * kernels of the LAST dcp_gpu_scan (valid after dcp_gpu_sync). */
float dcp_gpu_last_scan_ms(dcp_gpu_ctx *);
/* Number of DP kernel launches of the last scan (row sweep: one per profile
 * size class; query lane: one, plus one redo launch per size class). */
unsigned dcp_gpu_last_scan_launches(dcp_gpu_ctx const *);
/* The kernel the last scan ran with, as dcp_scan_params.kernel names it (1, 2 or 3): what the cost
 * model chose when the scan asked for 0.  0 before any scan. */
int dcp_gpu_last_scan_kernel(dcp_gpu_ctx const *);
/* Launch i of the last scan: its kernel shape, HIP-event duration on the
 * context's stream, DP cells and algorithmic bytes.  The cells
 * of a query-lane scan are all counted in its launch 0; its redo launches
 * report 0 cells. */
struct dcp_launch_info
{
    int nodes_per_lane; /* row sweep: R; query lane: nodes per tile */
    int waves_per_pair; /* row sweep: W; query lane: 0 */
    unsigned nprofiles;
    float ms;
    uint64_t cells;
    uint64_t algorithmic_bytes;
};
int dcp_gpu_last_scan_launch_info(dcp_gpu_ctx *, unsigned i,
                                  struct dcp_launch_info *out);

/* Results of the last scan (synchronises).
 * scores: null_out/alt_out [nseqs][nprofiles] (need keep_scores).
 * hits: sorted by (seq_idx, profile_idx); returns count in *nhits; DCP_ENOMEM
 * if cap is too small (nhits still set). */
int dcp_gpu_fetch_scores(dcp_gpu_ctx *, float *null_out, float *alt_out);
int dcp_gpu_fetch_hits(dcp_gpu_ctx *, struct dcp_hit *hits, unsigned cap,
                       unsigned *nhits);
/* ------------------------------------------------------------------------ */
/* Hits -> paths -> product rows                           */
/* ------------------------------------------------------------------------ */
/* imm_step {state_id, seqlen}: state ids as include/deciphon/model/protein_state.h:7-21 */
struct dcp_step
{
    uint16_t state_id;
    uint8_t seqlen;
    uint8_t reserved;
};

/* Viterbi paths (what imm_dp_viterbi leaves in prod.path,
 * src/server/scan_thread.c:115-117) of `nhits` (seq_idx, profile_idx) pairs of the
 * resident batch / DB, computed on the device: null_model == 0 -> alt model
 * (S ... T), != 0 -> null model (R steps). Flags as in the scan that found
 * them. steps_out receives the paths back to back; step_off[nhits+1] their
 * offsets; alt_out (may be NULL) the log-likelihood the trace recomputed (equal
 * to the scan's). DCP_ENOMEM if cap_steps is too small (step_off[nhits] then
 * holds the needed total), DCP_EFAIL if a pair has no finite path.
 * Device work: the rows of every hit are swept once more by its size class's row-sweep kernel, which parks every
 * row's M, I, D, N, B, E, J, C in a work area (12 x 64 R W + 20 bytes per row; at most 8 GiB per round of launches,
 * kept by the context between calls), then one wavefront per hit walks back through it. */
int dcp_gpu_trace_paths(dcp_gpu_ctx *, struct dcp_hit const *hits, unsigned nhits,
                        int multi_hits, int hmmer3_compat, int null_model,
                        struct dcp_step *steps_out, unsigned cap_steps, uint32_t *step_off,
                        float *alt_out);

/* protein_state_name (src/model/protein_state.c:5-39): "M12", "I3", "N"... */
unsigned dcp_state_name(unsigned state_id, char name[8]);
/* protein_profile_decode (src/model/protein_profile.c:306-331) = imm_frame_cond_decode:
 * most likely codon (ids 0..3) of a 1..5-nt fragment emitted by `state_id`. */
int dcp_profile_decode(dcp_profile const *, uint8_t const *frag, unsigned len,
                       unsigned state_id, uint8_t codon[3]);
/* imm_gc_decode(1, codon): amino acid letter of a codon, '*' for stops */
char dcp_gc_decode(uint8_t const codon[3]);
/* One product row exactly as prod_fwrite + protein_match_write_func write it
 * (src/server/prod.c:13-41,153-181; src/server/protein_match.c:21-56), without
 * the trailing newline handling changed: returns the number of bytes written to
 * buf (incl. the final '\n'), or -1 if cap is too small. seq: symbol ids. */
long dcp_prod_format_row(char *buf, size_t cap, int64_t scan_id, int64_t seq_id,
                         char const *profile_name, char const *abc_name,
                         double alt_loglik, double null_loglik,
                         char const *profile_typeid, char const *version,
                         dcp_profile const *prof, uint8_t const *seq, unsigned seq_len,
                         struct dcp_step const *steps, unsigned nsteps);
/* The header line prod_fclose writes (src/server/prod.c:119-121). */
char const *dcp_prod_header(void);

/* ------------------------------------------------------------------------ */
/* One process per GPU: profile shards + the hit gather over RCCL */
/* ------------------------------------------------------------------------ */
/* Pairs (profile, query) are independent: every rank keeps a contiguous profile shard (balanced by sum
 * of core sizes = DP cells) resident and scans ALL queries against it; the only exchange of the path
 * is the gather of the 16-byte hit records.  librccl.so is loaded on first use. */
typedef struct dcp_dist dcp_dist;
enum { DCP_DIST_ID_BYTES = 128 }; /* NCCL_UNIQUE_ID_BYTES */
enum { DCP_DIST_META_WORDS = 3 }; /* per rank in the meta all-gather: records held, profile offset, records found */
/* "records found" of a rank whose scan FAILED (it holds 0 records): not a count, a status every rank reads */
#define DCP_DIST_FOUND_FAILED 0xFFFFFFFFu
/* Rank 0 creates the communicator id (ncclGetUniqueId) and hands it to the other ranks through
 * whatever channel the launcher has (bench.py: torch.distributed; a C launcher: the file variant). */
int dcp_dist_unique_id(unsigned char id[DCP_DIST_ID_BYTES]);
dcp_dist *dcp_dist_init(unsigned char const id[DCP_DIST_ID_BYTES], int rank, int nranks, int device);
/* Rendezvous through a file: rank 0 writes the id to `path`, the others wait up to timeout_s for it.
 * Ranks holding different ids would block in ncclCommInitRank, so a peer must tell this run's file from a
 * left-over one.  _run: the launcher gives every rank the same non-zero `run_nonce` (its pid and start time,
 * say); rank 0 writes it into the file and a peer takes ONLY a file carrying it -- safe for back-to-back runs
 * at one path.  Without a nonce (the second form = run_nonce 0) `path` must be FRESH for every run: rank 0
 * removes what lies there before creating its id, and the other ranks refuse a file of another layout or rank
 * count or one written more than 120 s before they arrived. */
dcp_dist *dcp_dist_init_from_file_run(char const *path, uint64_t run_nonce, int rank, int nranks, int device,
                                      double timeout_s);
dcp_dist *dcp_dist_init_from_file(char const *path, int rank, int nranks, int device, double timeout_s);
void dcp_dist_free(dcp_dist *);
int dcp_dist_rank(dcp_dist const *);
int dcp_dist_nranks(dcp_dist const *);
/* Diagnostics for the launcher: the rank count RCCL itself reports for the communicator (ncclCommCount; must
 * equal dcp_dist_nranks; -1 without one), and the wall time of this rank's last gather in ms. */
int dcp_dist_comm_count(dcp_dist const *);
double dcp_dist_last_gather_ms(dcp_dist const *);
char const *dcp_dist_last_error(dcp_dist const *);
/* [begin, end) of rank's shard: dcp_partition_by_cells over nranks. */
void dcp_dist_shard(unsigned const *core_sizes, unsigned nprofiles, int nranks, int rank, unsigned *begin,
                    unsigned *end);
/* Gather the hit records of every rank's last scan -- the form hosts call: completes the scan of `ctx`
 * first (dcp_gpu_sync: a query-lane scan finishes its redo pairs there, and a scan whose redo lists
 * overflowed is repeated with the row sweep, so the list that travels is final), then gathers the buffer
 * that scan wrote.  profile_offset: first global profile index of this rank's shard (records carry
 * shard-local indices).  {held, offset, found} of every rank travel in one 3-word all-gather, the records
 * in one grouped ncclSend/ncclRecv (gather-v).  root >= 0: only that rank receives; root < 0: every rank
 * does.  On a receiving rank *out is a malloc'ed array (dcp_dist_free_hits) of *nout records with GLOBAL
 * profile indices, ordered by (seq_idx, profile_idx); elsewhere *out = NULL and *nout = the global total.
 * If ANY rank's scan found more hits than its buffer holds, every rank still completes both exchanges
 * with the records there are (nobody is left waiting) and then EVERY rank returns DCP_ENOMEM -- on a
 * receiving rank with *out set to the incomplete list.  A rank whose scan FAILED (dcp_gpu_sync /
 * dcp_gpu_hit_buffer returned an error) takes part in both exchanges holding nothing and marks its meta
 * words (found = DCP_DIST_FOUND_FAILED): it returns its scan's error, and EVERY other rank returns DCP_EFAIL
 * -- a root never gets DCP_OK for a list that lacks one shard's hits (*out, if set, is that incomplete list;
 * free it).  More than 2^32 - 1 records in all: DCP_EINVAL everywhere. */
int dcp_dist_gather_scan_hits(dcp_dist *, dcp_gpu_ctx *ctx, unsigned profile_offset, int root,
                              struct dcp_hit **out, unsigned *nout);
/* The same for an explicit device buffer (hits_dev / nhits_dev / cap as given to dcp_gpu_set_hit_buffer).
 * The caller must have completed the scan with dcp_gpu_sync(ctx) -- synchronising the stream alone (what
 * a non-NULL scan_stream does here) does not check the redo lists of a query-lane scan. */
int dcp_dist_gather_hits(dcp_dist *, void const *hits_dev, void const *nhits_dev, unsigned cap,
                         unsigned profile_offset, int root, void *scan_stream, struct dcp_hit **out,
                         unsigned *nout);
/* What every rank derives from the gathered meta words ({held, profile_offset, found} x nranks): counts,
 * offsets, 64-bit displacements displ[nranks + 1], whether any rank overflowed (found > held), whether any
 * rank's scan failed (found = DCP_DIST_FOUND_FAILED, held = 0), the total.  Host only.
 * DCP_EINVAL when the total exceeds 2^32 - 1 or a rank holds more than it found. */
int dcp_dist_gather_plan(uint32_t const *meta, int nranks, unsigned *counts, unsigned *profile_offset,
                         uint64_t *displ, int *any_overflow, int *any_failed, uint64_t *total);
void dcp_dist_free_hits(struct dcp_hit *hits);
/* The bookkeeping of the gather alone (host, no device, no RCCL): counts[r] records of rank r lie back
 * to back in `records`; out receives them with profile_idx += profile_offset[r], ordered by
 * (seq_idx, profile_idx).  Returns the total, -1 if cap is too small. */
long dcp_dist_merge_hits(unsigned const *counts, unsigned const *profile_offset, int nranks,
                         struct dcp_hit const *records, struct dcp_hit *out, unsigned cap);

/* Dynamic batching of the query-lane kernels, host side only (what dcp_gpu_scan does with a batch whose automatic
 * or forced kernel is a query-lane one): the queries in ascending length order are cut into groups of 64 -- one
 * wavefront's lanes -- and the groups are packed into wavefront slots (4 per 256-lane block, 1 for the
 * 64-lane variant) so that the slots of every block carry about the same number of DP rows; a slot sweeps its
 * groups one after the other.  len_sorted: ascending lengths.  Out: the block count, the rows a tile costs summed
 * over the blocks (each block's longest slot: the kernel-choice model's figure), the longest slot's rows; optionally
 * the groups as 4 words each {first query (index into the sorted order), queries, first plane row, longest member}
 * in slot order and slot_first[nblocks * slots + 1].  DCP_EINVAL for unsorted input, DCP_ENOMEM if a capacity
 * (group_cap in groups, slot_cap in words) is too small. */
int dcp_plan_query_slots(unsigned const *len_sorted, unsigned nq, unsigned slots_per_block, unsigned *nblocks,
                         unsigned long long *sum_block_rows, unsigned *plane_rows, unsigned *groups4,
                         unsigned group_cap, unsigned *slot_first, unsigned slot_cap);

/* Work accounting of the last scan (or of a full scan if none ran yet):
 * alt-model DP cells = sum over pairs of core_size * L (the Gcell/s numerator)
 * and the algorithmic bytes of SURVEY.md §8(d): sum over pairs of
 * 20*M*L + 32*(M+1) + L + 8. */
uint64_t dcp_gpu_scan_cells(dcp_gpu_ctx const *);
uint64_t dcp_gpu_scan_algorithmic_bytes(dcp_gpu_ctx const *);

#ifdef __cplusplus
}
#endif
#endif
