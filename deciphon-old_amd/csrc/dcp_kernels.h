// dcp_kernels.h -- POD argument blocks shared by the kernels and their launchers.
#ifndef DCP_KERNELS_H
#define DCP_KERNELS_H

#include "dcp_host.h"
#include <stdint.h>

// One resident profile (device array, sorted by size class then by index).
struct dcp_prof_meta
{
    uint64_t emis_off;  // float offset of this profile's match table in emis_match
    uint32_t trans_off; // float offset of its trans8[8][ldk] block
    uint32_t core_size; // M
    uint32_t ldk;       // padded node count = lanes * R of its size class
    uint32_t pidx;      // index in the caller's profile order
    uint32_t flags;     // DCP_PROF_EXACT_E
    uint32_t width;     // this profile's own columns of a row: core_size + padding of -inf (= ldk unless the profile
                        // shares its rows with others: dcp_mp_group)
};
// a finite MD or DD > 0: E(j) is not the maximum of the match states alone (a delete state may exceed every
// match state before it), so the kernels take the delete states into E(j) for this profile
#define DCP_PROF_EXACT_E 1u

// Profiles of at most 128 nodes share their table rows with their neighbours (round 4): K of them side by side in one
// table [1364][ldk] (and one transition block [8][ldk]), each followed by at least 8 columns of -inf, so that ONE
// wavefront scores K (profile, query) pairs with the same query at once -- 64 / K lanes of four nodes each
// (viterbi_mp_kernel).  A member's dcp_prof_meta is a column view of the group's table: emis_off / trans_off point at
// its first column, ldk is the group's row length, width its own columns.  in_off: the group's [1364][K] table of
// {insert, background} emissions (float2), one per-lane load per word instead of two scalar tables per profile.
struct dcp_mp_group
{
    uint64_t emis_off;  // float offset of the group's table in emis_match
    uint32_t trans_off; // float offset of its trans8 block
    uint32_t ldk;       // row length (multiple of 4)
    uint32_t in_off;    // float2 offset of its {eI, eN} table in mp_in
    uint32_t nparts;    // members, 1..K
    uint32_t col0[4];      // first column of each member (absent members: columns of -inf)
    uint32_t core_size[4]; // 0 for an absent member
    uint32_t pidx[4];      // caller's profile index
    uint32_t slot[4];      // entry of profs[]
};

// a (query, profile) pair the query-lane kernel hands to the row-sweep kernel
struct dcp_pair
{
    uint32_t q;    // sequence index relative to the scan's first
    uint32_t slot; // entry of profs[] (row-sweep order)
};
#define DCP_MAX_CLASSES 16

struct dcp_scan_args
{
    dcp_prof_meta const *profs;
    float const *emis_match;  // per profile [1364][ldk], -inf in padding columns
    float const *emis_insert; // [nprof_total][1364]
    float const *emis_null;   // [nprof_total][1364]
    float const *trans8;      // per profile [8][ldk], -inf in padding columns
    uint32_t const *seq_words; // 2-bit packed bases, 16 per word, per-seq aligned
    uint32_t const *seq_woff;  // [nseqs] first word of each sequence
    uint32_t const *seq_len;   // [nseqs]
    float const *xtrans;       // [nseqs][DCP_XSTRIDE]
    float *out_null;           // [nseqs][nprof_total] or NULL
    float *out_alt;            // [nseqs][nprof_total] or NULL
    dcp_hit *hits;
    unsigned *nhits;
    unsigned hit_cap;
    float lrt_threshold;
    unsigned first_prof; // first entry of profs[] in this launch's size class
    unsigned nprof;      // profiles in this launch
    unsigned nprof_total;
    unsigned nseqs;  // sequences in this launch (arrays already offset to the first)
    unsigned q_base; // index of the first one in the resident batch (hit records)
    unsigned qchunk;  // queries per task
    unsigned nchunks; // ceil(nseqs / qchunk)
    // pair mode (pairs != NULL): task i = pairs[i], i < min(*npairs, pair_cap); the grid is
    // persistent and strides over the list (its length is only known on the device)
    dcp_pair const *pairs;
    unsigned const *npairs;
    unsigned pair_cap;
    // segmented sweep of a multi-wavefront class (viterbi_segment_kernel, one segment per launch): the pairs' boundary
    // columns -- pair (profile s_rel of the class, query q) owns column s_rel * nseqs + q of seg_stride float4 in each
    // of the two buffers; segment s reads the one segment s - 1 wrote (col0 for odd s) and writes the other -- and
    // the list of the pairs the last segment hands to the exact kernel
    float *seg_col0, *seg_col1; // 16 bytes per row
    unsigned seg_stride; // rows (of 16 bytes) per column, >= lmax + 2
    unsigned seg_index;  // this launch's segment
    // profiles of at most 128 nodes, grid mode (viterbi_mp_kernel): first_prof / nprof then count GROUPS
    dcp_mp_group const *mp_groups;
    float const *mp_in; // {eI, eN} pairs: [group][1364][K]
    unsigned seg_q0, seg_nq; // the queries of this launch: q = seg_q0 .. seg_q0 + seg_nq - 1 (a chunk of the scan's, sized so that the columns fit)
    dcp_pair *seg_redo;
    unsigned *seg_redo_n;
    unsigned seg_redo_cap;
    // traceback's forward pass (dcp_launch_trace_forward, pair mode): pair i's work area starts at trace_work +
    // trace_woff[i] -- M, I, D as [L + 1][64 R W] each, then N, B, E, J, C as [L + 1] each -- and its alt score
    // goes to trace_alt[i]
    float *trace_work;
    uint64_t const *trace_woff;
    float *trace_alt;
};

// One 64-column tile of the expansion kernel.
struct dcp_expand_tile
{
    uint64_t out_off;  // float offset of column 0 of this tile in `out`
    uint32_t dist_row; // first row of dists[][129] (one row per column)
    uint32_t ncols;    // columns backed by a dist row
    uint32_t nstore;   // columns to write (>= ncols: the rest is -inf padding)
    uint32_t ld_code;  // stride between consecutive codes
    uint32_t ld_col;   // stride between consecutive columns
    // kt != 0: write the query-lane kernel's LDS tile image instead: column
    // (= node) k of the profile goes to [k / kt][(k % kt) / 4][code][k % 4];
    // out_off is then the profile's image base and col0 this tile's first node.
    uint32_t kt;
    uint32_t col0;
};

// ---- alt-path traceback for hits (viterbi_trace_kernel) ----------------------
struct dcp_trace_args
{
    dcp_prof_meta const *profs;
    uint32_t const *slot_of_pidx; // caller's profile index -> entry of profs[]
    float const *emis_match;
    float const *emis_insert;
    float const *emis_null;
    float const *trans8;
    uint32_t const *seq_words;
    uint32_t const *seq_woff;
    uint32_t const *seq_len;
    float const *xtrans;
    dcp_hit const *hits;   // pairs to trace (seq_idx, profile_idx of the resident batch / DB)
    unsigned nhits;
    float *work;           // per hit: 3 x [L+1][ldk] + 5 x [L+1] floats
    uint64_t const *work_off; // [nhits] float offset of each hit's work area
    dcp_step *steps;
    uint32_t const *step_off; // [nhits+1] capacity slices of steps[]
    uint32_t *nsteps;         // [nhits] steps written; 0xffffffff = no path
    float *alt_out;           // [nhits] log-likelihood recomputed by the trace
    int null_model;           // 0: alt model path (S..T); 1: null model path (R steps)
    int skip_forward;         // the work areas are filled (dcp_launch_trace_forward): walk back only
    uint32_t const *work_ld;  // [nhits] row length of each hit's M / I / D matrices, or NULL = the profile's own columns
};

// ---- query-lane kernel (dcp_qlane.hip) --------------------------------------
struct dcp_ql_prof
{
    uint64_t tile_off;   // float offset of the profile's tile images in emis_tiles
    uint32_t ttrans_off; // float offset of its per-tile transitions [T][KT+1][8]
    uint32_t core_size;
    uint32_t ntiles;     // T = ceil(core_size / KT)
    uint32_t pidx;
    uint32_t rs_slot;    // this profile's entry in the row-sweep kernel's profs[]
    uint32_t cls;        // its row-sweep size class (redo list to append to)
    uint32_t needs_exact_e; // a finite MD or DD > 0: E(j) is not the match states' maximum -> row sweep (redo lists)
    uint32_t ldk;        // DCP_DB_ONE_LAYOUT: row length of its row-sweep table (tile_off = its first column in emis_match)
};

// A group = up to 64 queries, consecutive in the length order, that ONE wavefront sweeps together (one lane each).
// A wavefront slot of a block holds a list of groups, swept one after the other per tile: the host packs groups into
// slots so that the slots of a block finish together whatever the batch's length mix ("dynamic batching": BASELINE
// configs[4]).  A group's rows are rowbase + 1 .. rowbase + L (L = its longest member) of the slot's column of the
// block's planes; its region is L + 10 rows (rounded up to even).
struct dcp_ql_group
{
    uint32_t qfirst;  // first entry of qorder[]
    uint32_t nq;      // queries (lanes in use), 1..64
    uint32_t rowbase; // first row of the group's region, minus one... row r of the group is plane row rowbase + r
    uint32_t lmax;    // its longest member
};

struct dcp_qlane_args
{
    dcp_ql_prof const *profs; // sorted by ascending size
    float const *emis_tiles;  // per profile [T][G][1364][4]: the LDS image of each tile (tiles_from_rows: emis_match)
    float const *emis_insert; // [nprof_total][1364]
    float const *emis_null;   // [nprof_total][1364]
    float const *ttrans;      // per profile [T][KT+1][8] (row KT = edges into the next tile)
    uint32_t const *seq_words;
    uint32_t const *seq_woff;
    uint32_t const *seq_len;
    float const *xtrans;
    uint32_t const *qorder;   // [nseqs] query indices sorted by length
    dcp_ql_group const *groups;  // [ngroups]
    uint32_t const *slot_first;  // [nqblocks * slots per block + 1]: a slot's groups are groups[slot_first[s] .. slot_first[s + 1])
    // window plane of each block of query slots: uint16 [plane rows][lanes per block], entry (r, t) = (the window of the
    // row r of lane t's column) << 4 -- one coalesced load per row; block qb's plane starts at word wt_off[qb]
    uint32_t *words_t;
    uint32_t const *wt_off;   // [nqblocks + 1]
    float *scratch;           // [nblocks][3 (or 4) planes][plane_rows][lanes per block]
    unsigned *task_counter;
    // the two-stage kernel's ring hand-shake gives up after a bounded number of polls: it sets *ring_error and both
    // stages drain (the scan's results are then invalid and dcp_gpu_sync says so).  ring_stall: TEST ONLY (set through
    // the -DDCP_TEST_HOOKS build's setter, always 0 in the shipped library): stage 0 of the first task's first step
    // never sweeps, so its partner stage runs into the bound.
    unsigned *ring_error;
    unsigned ring_stall;
    // redo lists, one per row-sweep size class: pairs whose B0(j) = N(j) + NB was beaten by the
    // E -> B / J -> B feedback (dcp_qlane.hip header)
    dcp_pair *redo;
    unsigned *redo_n;        // [DCP_MAX_CLASSES] appended so far (may exceed the capacity)
    unsigned *redo_overflow; // set when a list overflowed
    unsigned redo_base[DCP_MAX_CLASSES];
    unsigned redo_cap[DCP_MAX_CLASSES];
    float *out_null;
    float *out_alt;
    dcp_hit *hits;
    unsigned *nhits;
    unsigned hit_cap;
    float lrt_threshold;
    unsigned nprof;
    unsigned nprof_total;
    unsigned nseqs;
    unsigned q_base;
    unsigned plane_rows; // rows of a block's scratch planes: the longest slot's rows + 8 (prefetch runs past the end)
    unsigned ntasks;   // nprof * nqblocks
    unsigned nqblocks; // blocks of query slots
    unsigned tiles_from_rows; // DCP_DB_ONE_LAYOUT: the launchers take the kernels that gather a tile's image from emis_match
};

struct dcp_expand_args
{
    dcp_expand_tile const *tiles;
    float const *dists; // [rows][129]
    float const *eps;   // [rows] frame epsilon of each row
    float *out;
};

#ifdef __cplusplus
extern "C" {
#endif
void dcp_launch_expand(dcp_expand_args const *a, unsigned ntiles, void *stream);
int dcp_launch_rowsweep(int R, int W, dcp_scan_args const *a, unsigned nblocks,
                        void *stream);
unsigned dcp_rowsweep_tasks_per_block(int W);
// segment a->seg_index of every pair of the launch's profiles x queries, one wavefront per pair, segments of 64 x R
// nodes (R = 5..8); != 0: no such kernel
int dcp_launch_segsweep(int R, dcp_scan_args const *a, unsigned nblocks, void *stream);
// K = 2 or 4 profiles per wavefront (groups a->first_prof .. + a->nprof of a->mp_groups) x all queries; != 0: no such kernel
int dcp_launch_mp(int K, dcp_scan_args const *a, void *stream);
// the groups' {insert, background} tables: groups [0, n_first) hold k_first members per wavefront, the rest k_rest
void dcp_launch_mp_in(dcp_mp_group const *groups, unsigned ngroups, unsigned n_first, unsigned k_first, unsigned k_rest,
                      float const *emis_insert, float const *emis_null, float *out, void *stream);
unsigned dcp_segsweep_blocks(unsigned nprof, unsigned nq); // grid of one segment launch over nprof profiles x nq queries
// grid mode (all chunks x the profiles of one size class): stg = leading emission rows a block stages in LDS
// (0, 20 or 84), bw = wavefronts per staged block; != 0 if there is no such kernel or the grid is too large
int dcp_launch_rowsweep_grid(int R, int W, dcp_scan_args const *a, int stg, unsigned bw, void *stream,
                             unsigned pad_lds, // unused dynamic LDS per block (occupancy experiments), normally 0
                             int prefetch2);   // != 0: the variant that fetches its global rows two DP rows ahead
unsigned dcp_rowsweep_max_block_waves(int R, int W, int stg); // 0: no kernel stages `stg` rows for this class
unsigned dcp_rowsweep_stage_bytes(int R, int stg);
int dcp_launch_qlane(dcp_qlane_args const *a, unsigned nblocks, unsigned nt, void *stream); // nt: 64, 128 or 256 queries per block
// two-stage variant: 512-thread blocks, one per CU; != 0 if the kernel cannot be configured
int dcp_launch_qlane2(dcp_qlane_args const *a, unsigned nblocks, void *stream);
unsigned dcp_qlane2_lds_bytes(void);
// small batches: three independent 64-query wavefronts per 192-thread block, one block per CU
int dcp_launch_qlane_w3(dcp_qlane_args const *a, unsigned nblocks, void *stream);
int dcp_launch_qlane_transpose(dcp_qlane_args const *a, unsigned nt, void *stream);
void dcp_launch_trace(dcp_trace_args const *a, unsigned nhits, void *stream);
int dcp_launch_trace_forward(int R, int W, dcp_scan_args const *a, unsigned nblocks, void *stream);
unsigned dcp_qlane_block_size(void);
unsigned dcp_qlane_tile_nodes(void);
unsigned dcp_qlane_scratch_planes(void);
unsigned dcp_qlane_diag_build(void); // != 0: a -DDCP_QLANE_DIAG timing build (wrong results)
unsigned dcp_qlane_group_rows(unsigned lmax); // plane rows of a group's region (its longest member + 10, even)
unsigned dcp_qlane_window_planes(void);       // != 0: the plane holds per-row windows (uint16) instead of packed words
unsigned dcp_qlane_exact_e_by_redo(void);     // != 0: profiles with a positive MD / DD must go through the redo lists
#ifdef __cplusplus
}
#endif

#endif
