// dcp_gpu.hip -- C-ABI of the device path (include/dcp_gpu.h): context, resident
// profile DB, resident sequence batch, scan launch, result fetch.
//
// Replaces, for the hot path only: profile_reader_* + protein_profile.unpack
// (src/db/profile_reader.c:74-168, src/model/protein_profile.c:38-117),
// imm_task_setup (src/server/scan_thread.c:51-55) and the per-pair body of
// thread_run (src/server/scan_thread.c:99-123).  No CPU fallback exists.
#include "dcp_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <new>
#include <string>
#include <queue>
#include <vector>

namespace
{

// Size classes: a profile with core_size M runs in the smallest class whose
// capacity 64*R*W holds it (R nodes per lane, W wavefronts per pair).
struct SizeClass
{
    int R, W;
    unsigned cap() const { return 64u * (unsigned)R * (unsigned)W; }
};
// One wavefront with up to 8 nodes per lane covers core sizes up to 512 (97 % of a Pfam-like DB) without
// any cross-wavefront exchange; measured on the C3 DB, R = 5..8 in one wavefront take 0.52 / 0.66 of the
// time of the two-wavefront classes R = 3, 4 they replace, while two wavefronts of 6 or 8 nodes per lane
// (two per SIMD: too few to hide the block barriers) are slower than four of 3 or 4
// (profiles/r02/rowsweep_tuning.txt).
constexpr SizeClass kClasses[] = {{1, 1}, {2, 1}, {3, 1}, {4, 1}, {5, 1}, {6, 1}, {7, 1}, {8, 1},
                                  {3, 4}, {4, 4}, {3, 8}, {4, 8}, {3, 16}, {4, 16}};
// cells/s of each class on a full grid, measured on the C3 DB (profiles/r03/rowsweep_variants.txt; round 2:
// profiles/r02/rowsweep_tuning.txt): the kernel choice prices a row-sweep scan with them
constexpr double kClassRate[] = {880e9, 930e9, 1120e9, 1250e9, 1240e9, 1320e9, 1220e9, 1230e9,
                                 830e9, 960e9, 820e9, 600e9, 600e9, 600e9}; // multi-wavefront classes: the segmented sweep (round 4: profiles/r04/rowsweep_multiwave.txt)
constexpr int kNumClasses = (int)(sizeof kClasses / sizeof kClasses[0]);
static_assert(sizeof kClassRate / sizeof kClassRate[0] == sizeof kClasses / sizeof kClasses[0], "one rate per class");
static_assert(kNumClasses <= DCP_MAX_CLASSES, "redo lists are sized for DCP_MAX_CLASSES size classes");

// profiles per wavefront of the two smallest classes in grid mode (viterbi_mp_kernel<K>: 64 / K lanes x 4 nodes each)
constexpr unsigned kMpParts[2] = {4u, 2u};
constexpr unsigned kMpClass1MaxQueries = 96u; // the 65 .. 128-node class runs two profiles per wavefront below this many queries

int class_of(unsigned M)
{
    for (int c = 0; c < kNumClasses; ++c)
        if (M <= kClasses[c].cap()) return c;
    return -1;
}

// Nodes per lane of the segments a profile of more than 512 nodes is cut into by the segmented row sweep
// (viterbi_segment_kernel): the fewest segments of at most 512 nodes, and the narrowest lanes that cover the profile
// with them -- R = ceil(M / (64 x ceil(M / 512))), 5..8 -- so that a segment's lanes are at least 80 % full (with a
// fixed 384 / 512 nodes per class they were 67 .. 100 %).  0: a one-wavefront profile.
unsigned seg_r_of(unsigned M)
{
    if (M <= 512u) return 0u;
    unsigned const nseg = (M + 511u) / 512u;
    return (M + 64u * nseg - 1u) / (64u * nseg);
}

template <class T> struct DevBuf
{
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        else p = nullptr;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

} // namespace

struct dcp_gpu_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::string err;

    // resident DB
    unsigned nprof = 0;
    std::vector<dcp_prof_meta> metas;   // sorted by class
    std::vector<unsigned> core_sizes;   // by pidx
    unsigned class_first[kNumClasses + 1] = {0};
    DevBuf<dcp_prof_meta> d_metas;
    DevBuf<uint32_t> d_slot_of_pidx;
    std::vector<uint32_t> slot_of_pidx;
    DevBuf<float> d_emis_match, d_emis_insert, d_emis_null, d_trans8;
    // the row-sweep layout of the match tables (d_emis_match) is expanded on first use: the
    // throughput path only needs the tile images, and both together double the DB's footprint
    DevBuf<float> d_dists, d_eps;
    std::vector<dcp_expand_tile> rs_tiles;
    uint64_t rs_floats = 0;
    bool rs_ready = false;
    // DCP_DB_ONE_LAYOUT: no tile images; the query-lane kernels gather a tile's LDS image from d_emis_match (always resident)
    bool one_layout = false;
    uint64_t table_bytes = 0; // expanded match tables resident after the upload (either or both layouts)
    // query-lane kernel layout (dcp_qlane.hip)
    int ql_G = 2; // nodes per tile = 4 * G (KT = 8: the tile transitions fit in SGPRs)
    std::vector<dcp_ql_prof> ql_metas; // same order as metas
    uint64_t sum_core = 0, sum_tiles = 0; // over the resident DB (kernel choice)
    uint64_t class_core[DCP_MAX_CLASSES] = {0}; // sum of core sizes per row-sweep size class
    unsigned max_tiles = 0;
    DevBuf<dcp_ql_prof> d_ql_metas;
    DevBuf<float> d_emis_tiles, d_ttrans, d_scratch;
    DevBuf<uint32_t> d_qorder;
    DevBuf<uint32_t> d_words_t, d_wt_off; // query-lane kernel: per-block window planes
    // profiles of at most 128 nodes: groups of K sharing table rows (dcp_mp_group), per row-sweep class 0 / 1
    std::vector<dcp_mp_group> mp_groups;
    unsigned mp_first[3] = {0, 0, 0};        // groups of class 0: [mp_first[0], mp_first[1]), class 1: [mp_first[1], mp_first[2])
    unsigned mp_flagged_first[2] = {0, 0};   // first flagged (ungrouped) profile of class 0 / 1 in metas order
    DevBuf<dcp_mp_group> d_mp_groups;
    DevBuf<float> d_mp_in;                   // [group][1364][K] {eI, eN}
    bool mp_in_ready = false;
    DevBuf<dcp_ql_group> d_ql_groups;     // groups of up to 64 queries and the wavefront slots they are packed into
    DevBuf<uint32_t> d_slot_first;
    unsigned ql_nqb = 0, ql_plane_rows = 0; // the cached plan: blocks of slots, rows of a block's scratch planes
    unsigned ring_stall = 0;                // test hook: dcp_qlane_args::ring_stall
    bool ring_check_pending = false;        // a two-stage scan's error word has not been looked at yet
    // the length-sorted query order and the plane offsets travel through pinned host memory: the copies are
    // truly asynchronous and dcp_gpu_scan_range returns without waiting for the stream
    uint32_t *h_qstage = nullptr;
    size_t h_qstage_n = 0;
    hipEvent_t ev_qstage = nullptr;       // recorded behind the copies: the buffer may be rewritten after it
    int rs_force_stg = -1;                // test hook: row-sweep variant (rows staged, wavefronts per block)
    unsigned rs_force_bw = 0;
    unsigned rs_pad_lds = 0;
    int rs_force_pf = 0;
    int rs_force_R = 0; // 0: every class
    bool any_exact_e = false;             // some profile has a positive MD / DD (dcp_ql_prof::needs_exact_e)
    DevBuf<unsigned> d_task_counter;
    // redo lists of the query-lane kernel (pairs handed to the row sweep), one per size class;
    // d_redo_n = [DCP_MAX_CLASSES counters][overflow flag][ring hand-shake error word]
    DevBuf<dcp_pair> d_redo;
    DevBuf<unsigned> d_redo_n;
    bool redo_pending = false;      // the last scan's redo counters have not been checked yet
    unsigned last_redo_pairs = 0;
    struct dcp_scan_params last_prm = {};
    unsigned qorder_q0 = ~0u, qorder_q1 = ~0u, qorder_lmax = 0;
    unsigned num_cus = 0;
    int last_kernel = 0; // 1 row sweep, 2 query lane
    unsigned qorder_nt = 0;         // block size the cached query order / transposed words were built for
    int last_kernel_variant = 0; // as dcp_scan_params.kernel names it: 1, 2 or 3 (two-stage query lane)
    unsigned redo_cap_limit = 1u << 26; // only the -DDCP_TEST_HOOKS build can change it
    float last_ql_ms = 0;

    // resident sequences
    unsigned nseqs = 0;
    uint64_t total_len = 0;
    std::vector<uint32_t> seq_len;
    DevBuf<uint32_t> d_seq_words, d_seq_woff, d_seq_len;
    DevBuf<float> d_xtrans;
    int xt_multi = -1, xt_h3 = -1;
    bool xt_explicit = false; // dcp_gpu_seqs_set_xtrans: the caller's transitions, not the length-derived ones

    // results
    DevBuf<float> d_null, d_alt;
    DevBuf<dcp_hit> d_hits;
    DevBuf<unsigned> d_nhits;
    unsigned hit_cap = 0;
    bool have_scores = false;
    unsigned last_launches = 0;
    bool scanned = false;
    unsigned last_q0 = 0, last_q1 = 0;
    // caller-owned hit buffer (e.g. a torch tensor that RCCL gathers)
    dcp_hit *ext_hits = nullptr;
    unsigned *ext_nhits = nullptr;
    unsigned ext_cap = 0;
    // one HIP event after each size-class launch of the last scan
    static constexpr int kMaxLaunches = 2 * kNumClasses + 1; // a segmented class is two launches
    hipEvent_t ev_class[kMaxLaunches] = {nullptr};
    // small row-sweep scans (the reference's one-sequence-at-a-time mode) are bound by the latency
    // of one pair's row chain, not by throughput: their size-class launches run side by side
    hipStream_t class_stream[kNumClasses] = {nullptr};
    hipEvent_t ev_trace[kNumClasses + 1] = {nullptr}; // traceback: the fork point and one event per class's forward launch
    bool last_overlapped = false;
    int launched_class[kMaxLaunches] = {0};
    bool launched_redo[kMaxLaunches] = {false}; // the exact kernel behind a segmented sweep: its cells are counted there
    // segmented sweep of the multi-wavefront classes (grid mode): per-class scratch columns and redo lists
    DevBuf<float> d_trace_work;   // the traceback's work areas (kept between calls, grows to the largest round)
    int trace_mode = 0;            // test hook: 1 = the trace kernel's own forward loop instead of the row sweep's
    uint64_t trace_budget = 0;     // test hook: floats of work area per round of launches (0 = 2^31)
    DevBuf<float> d_seg_scratch;
    DevBuf<dcp_pair> d_seg_redo;
    DevBuf<unsigned> d_seg_redo_n; // [DCP_MAX_CLASSES]
    int seg_mode = -1;             // test hook: 0 never, 1 always where a kernel exists, -1 automatic
    int mp_mode = -1;              // test hook: 0 never K profiles per wavefront, 1 always, -1 automatic
    uint64_t seg_col_bytes = (uint64_t)6 << 30; // cap on one class's boundary columns (test hook: shrink it to reach the chunked path)
    unsigned n_launched = 0;

    int fail(int rc, char const *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        fprintf(stderr, "dcp_gpu: %s\n", buf);
        return rc;
    }
};

#define HIP_TRY(ctx, call)                                                     \
    do                                                                         \
    {                                                                          \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess)                                                  \
            return (ctx)->fail(e_ == hipErrorOutOfMemory ? DCP_ENOMEM : DCP_EFAIL, \
                               "%s: %s", #call, hipGetErrorString(e_));        \
    } while (0)

extern "C" {

int dcp_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

dcp_gpu_ctx *dcp_gpu_ctx_new(int device)
{
    int n = dcp_gpu_device_count();
    if (n <= 0 || device < 0 || device >= n)
    {
        fprintf(stderr,
                "dcp_gpu: no HIP device %d (found %d); this engine has no CPU "
                "fallback\n",
                device, n);
        return nullptr;
    }
    dcp_gpu_ctx *c = new (std::nothrow) dcp_gpu_ctx();
    if (!c) return nullptr;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev_start) != hipSuccess ||
        hipEventCreate(&c->ev_stop) != hipSuccess)
    {
        fprintf(stderr, "dcp_gpu: failed to create stream/events on device %d\n", device);
        delete c;
        return nullptr;
    }
    if (dcp_qlane_diag_build())
        fprintf(stderr, "dcp_gpu: WARNING: this library is a -DDCP_QLANE_DIAG=%u timing build; its scores are WRONG\n",
                dcp_qlane_diag_build());
    {
        hipDeviceProp_t prop;
        c->num_cus = hipGetDeviceProperties(&prop, device) == hipSuccess ? (unsigned)prop.multiProcessorCount : 256u;
        c->ql_G = (int)dcp_qlane_tile_nodes() / 4;
    }
    bool ok = true;
    for (int k = 0; k < dcp_gpu_ctx::kMaxLaunches; ++k)
        ok = ok && hipEventCreate(&c->ev_class[k]) == hipSuccess;
    for (int k = 0; k < kNumClasses; ++k)
        ok = ok && hipStreamCreateWithFlags(&c->class_stream[k], hipStreamNonBlocking) == hipSuccess;
    if (!ok)
    {
        fprintf(stderr, "dcp_gpu: failed to create stream/events on device %d\n", device);
        delete c;
        return nullptr;
    }
    return c;
}

void dcp_gpu_ctx_del(dcp_gpu_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->ev_qstage) (void)hipEventDestroy(c->ev_qstage);
    if (c->h_qstage) (void)hipHostFree(c->h_qstage);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    for (int k = 0; k < dcp_gpu_ctx::kMaxLaunches; ++k)
        if (c->ev_class[k]) (void)hipEventDestroy(c->ev_class[k]);
    for (int k = 0; k <= kNumClasses; ++k)
        if (c->ev_trace[k]) (void)hipEventDestroy(c->ev_trace[k]);
    for (int k = 0; k < kNumClasses; ++k)
        if (c->class_stream[k]) (void)hipStreamDestroy(c->class_stream[k]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

char const *dcp_gpu_last_error(dcp_gpu_ctx const *c) { return c ? c->err.c_str() : "no context"; }
void *dcp_gpu_stream(dcp_gpu_ctx *c) { return (void *)c->stream; }
unsigned dcp_gpu_db_nprofiles(dcp_gpu_ctx const *c) { return c->nprof; }
unsigned dcp_gpu_nseqs(dcp_gpu_ctx const *c) { return c->nseqs; }

// Host -> device stream of floats in device order through two pinned buffers (dcp_gpu_db_upload).
struct StagedUpload
{
    static constexpr size_t kCap = (size_t)16 << 20; // floats per buffer (64 MB)
    dcp_gpu_ctx *c = nullptr;
    float *dst = nullptr, *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    size_t off = 0, fill = 0;
    int cur = 0;
    int begin(dcp_gpu_ctx *ctx, float *device_dst)
    {
        c = ctx, dst = device_dst;
        for (int i = 0; i < 2; ++i)
        {
            HIP_TRY(c, hipHostMalloc((void **)&buf[i], kCap * sizeof(float), hipHostMallocDefault));
            HIP_TRY(c, hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        return DCP_OK;
    }
    int flush()
    {
        if (fill == 0) return DCP_OK;
        HIP_TRY(c, hipMemcpyAsync(dst + off, buf[cur], fill * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(ev[cur], c->stream));
        busy[cur] = true;
        off += fill, fill = 0, cur ^= 1;
        if (busy[cur]) // the buffer about to be refilled: its copy must have left it
        {
            HIP_TRY(c, hipEventSynchronize(ev[cur]));
            busy[cur] = false;
        }
        return DCP_OK;
    }
    int append(float const *src, size_t n)
    {
        while (n)
        {
            size_t const k = std::min(n, kCap - fill);
            std::memcpy(buf[cur] + fill, src, k * sizeof(float));
            fill += k, src += k, n -= k;
            if (fill == kCap)
                if (int rc = flush()) return rc;
        }
        return DCP_OK;
    }
    int finish()
    {
        if (int rc = flush()) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        busy[0] = busy[1] = false;
        return DCP_OK;
    }
    ~StagedUpload()
    {
        if (c && (busy[0] || busy[1])) (void)hipStreamSynchronize(c->stream);
        for (int i = 0; i < 2; ++i)
        {
            if (buf[i]) (void)hipHostFree(buf[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
    }
};

// ---------------------------------------------------------------------------
// DB upload
// ---------------------------------------------------------------------------
int dcp_gpu_db_upload(dcp_gpu_ctx *c, dcp_profile *const *profiles,
                      unsigned nprofiles, int flags)
{
    if (!c) return DCP_EINVAL;
    if (flags & ~(DCP_DB_EXPAND_ON_HOST | DCP_DB_ONE_LAYOUT)) return c->fail(DCP_EINVAL, "unknown upload flags %d", flags);
    int const expand_on_host = flags & DCP_DB_EXPAND_ON_HOST;
    if (!profiles || nprofiles == 0) return c->fail(DCP_EINVAL, "empty profile list");
    if (nprofiles > (1u << 20)) return c->fail(DCP_EINVAL, "too many profiles"); // MAX_NPROFILES limits.h:7
    HIP_TRY(c, hipSetDevice(c->device));
    c->scanned = false;
    c->redo_pending = false;

    // classify and order: by size class, then by caller index
    std::vector<int> cls(nprofiles);
    c->core_sizes.assign(nprofiles, 0);
    for (unsigned p = 0; p < nprofiles; ++p)
    {
        if (!profiles[p]) return c->fail(DCP_EINVAL, "null profile %u", p);
        unsigned M = dcp_profile_core_size(profiles[p]);
        int k = class_of(M);
        if (k < 0) return c->fail(DCP_EINVAL, "profile %u: core_size %u too big", p, M);
        cls[p] = k;
        c->core_sizes[p] = M;
    }
    // E(j) as the match states' maximum (both kernels) needs MD, DD <= 0: true of every profile whose
    // transitions are log-probabilities.  Any other profile is flagged: the row sweep then takes the
    // delete states into E(j), the query-lane kernels hand its pairs to the row sweep (redo lists).
    std::vector<uint8_t> flagged(nprofiles, 0);
    for (unsigned p = 0; p < nprofiles; ++p)
    {
        float const *t8 = dcp_profile_trans8(profiles[p]);
        unsigned const M = c->core_sizes[p];
        bool pos = false;
        for (unsigned k = 1; k < M && !pos; ++k) // edges into node 0 do not exist (-inf below)
            pos = t8[(size_t)DCP_T_MD * M + k] > 0.0f || t8[(size_t)DCP_T_DD * M + k] > 0.0f;
        flagged[p] = pos ? 1 : 0;
    }
    std::vector<unsigned> order(nprofiles);
    for (unsigned p = 0; p < nprofiles; ++p)
        order[p] = p;
    // by size class; inside a class of several wavefronts by the segmented sweep's lane width (seg_r_of), so that the
    // profiles of one segment kernel are a contiguous range; inside the two smallest classes the flagged profiles last
    // (the others are grouped K to a wavefront: kMpParts); then by the caller's index
    std::stable_sort(order.begin(), order.end(), [&](unsigned a, unsigned b) {
        if (cls[a] != cls[b]) return cls[a] < cls[b];
        if (cls[a] < 2) return flagged[a] < flagged[b];
        return seg_r_of(c->core_sizes[a]) < seg_r_of(c->core_sizes[b]);
    });

    c->metas.assign(nprofiles, dcp_prof_meta{});
    std::vector<uint32_t> dist_row(nprofiles);
    uint64_t emis_floats = 0, trans_floats = 0, match_rows = 0;
    for (int k = 0; k <= kNumClasses; ++k)
        c->class_first[k] = 0;
    c->mp_groups.clear();
    c->mp_first[0] = c->mp_first[1] = c->mp_first[2] = 0;
    c->mp_flagged_first[0] = c->mp_flagged_first[1] = 0;
    c->mp_in_ready = false;
    uint64_t mp_in_pairs = 0;
    for (unsigned i = 0; i < nprofiles;)
    {
        unsigned const p0 = order[i];
        int const k0 = cls[p0];
        SizeClass const sc = kClasses[k0];
        // The two smallest classes (at most 128 nodes), unflagged: kMpParts[class] consecutive profiles share one
        // table -- their columns side by side, each followed by >= 8 columns of -inf -- and one wavefront scores
        // them against a query together (viterbi_mp_kernel).  Each member's meta is a column view of that table.
        unsigned members = 1;
        if (k0 < 2 && !flagged[p0])
            while (members < kMpParts[k0] && i + members < nprofiles && cls[order[i + members]] == k0 && !flagged[order[i + members]])
                ++members;
        bool const grouped = k0 < 2 && !flagged[p0];
        unsigned widths[4] = {0, 0, 0, 0}, ldk = 0;
        for (unsigned j = 0; j < members; ++j)
        {
            unsigned const M = c->core_sizes[order[i + j]];
            if (grouped) widths[j] = (M + 8u + 3u) & ~3u;
            else
            {
                // Row length of a stand-alone profile's tables.  Several wavefronts per pair: the class capacity
                // (every lane has columns, the padding ones -inf), or core_size + 8 if that is shorter.  One
                // wavefront: core_size + R columns rounded up to 4 -- every row ends in at least R columns of -inf,
                // which all lanes past the last node read instead of owning padding columns: 15 % fewer bytes per
                // row on a Pfam-like size distribution, fewer HBM and L2 lines per DP row.
                unsigned w = sc.cap();
                if (sc.W == 1 && M <= 63u * (unsigned)sc.R) w = (M + (unsigned)sc.R + 3u) & ~3u;
                if (sc.W > 1) w = std::min(w, (M + 8u + 3u) & ~3u);
                widths[j] = w;
            }
            ldk += widths[j];
        }
        dcp_mp_group g{};
        if (grouped)
        {
            g.emis_off = emis_floats, g.trans_off = (uint32_t)trans_floats, g.ldk = ldk;
            g.in_off = (uint32_t)mp_in_pairs, g.nparts = members;
            mp_in_pairs += (uint64_t)DCP_NCODES * kMpParts[k0];
        }
        unsigned col0 = 0;
        for (unsigned j = 0; j < members; ++j)
        {
            unsigned const p = order[i + j];
            dcp_prof_meta &m = c->metas[i + j];
            m.emis_off = emis_floats + col0;
            m.trans_off = (uint32_t)(trans_floats + col0);
            m.core_size = c->core_sizes[p];
            m.ldk = ldk;
            m.width = widths[j];
            m.pidx = p;
            m.flags = flagged[p] ? DCP_PROF_EXACT_E : 0u;
            dist_row[i + j] = (uint32_t)match_rows;
            match_rows += m.core_size;
            if (grouped) g.col0[j] = col0, g.core_size[j] = m.core_size, g.pidx[j] = p, g.slot[j] = i + j;
            col0 += widths[j];
        }
        if (grouped)
        {
            // an absent member's lanes read -inf: the padding behind the last member
            for (unsigned j = members; j < 4; ++j)
                g.col0[j] = g.col0[members - 1] + ((g.core_size[members - 1] + 3u) & ~3u), g.core_size[j] = 0;
            c->mp_groups.push_back(g);
            c->mp_first[k0 + 1] = (unsigned)c->mp_groups.size();
        }
        else if (k0 < 2 && c->mp_flagged_first[k0] == 0 && flagged[p0])
            c->mp_flagged_first[k0] = i + 1u; // (+1: 0 means none)
        emis_floats += (uint64_t)DCP_NCODES * ldk;
        trans_floats += 8ull * ldk;
        for (unsigned j = 0; j < members; ++j)
            c->class_first[cls[order[i + j]] + 1] = i + j + 1;
        i += members;
    }
    if (c->mp_first[1] < c->mp_first[0]) c->mp_first[1] = c->mp_first[0];
    if (c->mp_first[2] < c->mp_first[1]) c->mp_first[2] = c->mp_first[1];
    if (mp_in_pairs > 0xffffffffull) return c->fail(DCP_EINVAL, "DB too large for 32-bit table offsets");
    for (int k = 1; k <= kNumClasses; ++k) // empty classes inherit the boundary
        if (c->class_first[k] < c->class_first[k - 1]) c->class_first[k] = c->class_first[k - 1];
    if (trans_floats > 0xffffffffull || match_rows > 0xffffffffull)
        return c->fail(DCP_EINVAL, "DB too large for 32-bit row offsets");

    HIP_TRY(c, c->d_metas.alloc(nprofiles));
    c->slot_of_pidx.assign(nprofiles, 0);
    for (unsigned i = 0; i < nprofiles; ++i)
        c->slot_of_pidx[c->metas[i].pidx] = i;
    HIP_TRY(c, c->d_slot_of_pidx.alloc(nprofiles));
    HIP_TRY(c, hipMemcpy(c->d_slot_of_pidx.p, c->slot_of_pidx.data(), nprofiles * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->rs_floats = emis_floats;
    c->rs_ready = false;
    c->rs_tiles.clear();
    c->d_emis_match.release();
    c->d_emis_tiles.release();
    c->d_dists.release();
    c->d_eps.release();
    // One table layout or two.  The tile images of the query-lane kernels (19.6 GB for 20 000 Pfam-like profiles) repeat
    // the numbers of the row-sweep tables (20.6 GB), which a scan with hits needs anyway (redo lists, traceback).  With
    // DCP_DB_ONE_LAYOUT -- or by itself when both would not leave 16 GiB of this device's free memory for planes,
    // sequences and the traceback's work areas -- only the row-sweep tables are built, at once, and the query-lane
    // kernels gather each tile's image from them (stage_tile_image<G, true>, dcp_qlane.hip).
    uint64_t tile_floats_needed = 0;
    for (unsigned i = 0; i < nprofiles; ++i)
        tile_floats_needed += (uint64_t)((c->metas[i].core_size + 4u * (unsigned)c->ql_G - 1u) / (4u * (unsigned)c->ql_G)) *
                              (4u * (unsigned)c->ql_G) * DCP_NCODES;
    c->one_layout = (flags & DCP_DB_ONE_LAYOUT) != 0;
    if (!c->one_layout)
    {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
        uint64_t const both = (tile_floats_needed + emis_floats + (match_rows + 2ull * nprofiles) * (DCP_NDIST + 1)) * sizeof(float);
        if (both + (16ull << 30) > (uint64_t)free_b) c->one_layout = true;
    }
    c->table_bytes = (c->one_layout ? emis_floats : tile_floats_needed + (expand_on_host ? emis_floats : 0)) * sizeof(float);
    if (expand_on_host || c->one_layout) HIP_TRY(c, c->d_emis_match.alloc(emis_floats));
    HIP_TRY(c, c->d_trans8.alloc(trans_floats));
    HIP_TRY(c, c->d_emis_insert.alloc((size_t)nprofiles * DCP_NCODES));
    HIP_TRY(c, c->d_emis_null.alloc((size_t)nprofiles * DCP_NCODES));
    HIP_TRY(c, hipMemcpyAsync(c->d_metas.p, c->metas.data(), nprofiles * sizeof(dcp_prof_meta),
                              hipMemcpyHostToDevice, c->stream));

    // transitions, padded with -inf (unreachable padding nodes)
    {
        float const ninf = -std::numeric_limits<float>::infinity();
        std::vector<float> t8(trans_floats, ninf);
        for (unsigned i = 0; i < nprofiles; ++i)
        {
            dcp_prof_meta const &m = c->metas[i];
            float const *src = dcp_profile_trans8(profiles[m.pidx]);
            for (int row = 0; row < 8; ++row)
                std::memcpy(&t8[m.trans_off + (size_t)row * m.ldk], src + (size_t)row * m.core_size,
                            sizeof(float) * m.core_size); // (a view: row stride = the group's ldk)
            // The first node has no predecessor node: whatever a caller-built profile (dcp_profile_from_parts)
            // holds there, the edges into it are -inf.  The row sweep relies on it (lane 0 adds them to 0).
            for (int row : {DCP_T_MM, DCP_T_IM, DCP_T_DM, DCP_T_MD, DCP_T_DD})
                t8[m.trans_off + (size_t)row * m.ldk] = ninf;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_trans8.p, t8.data(), t8.size() * sizeof(float),
                                  hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }

    // query-lane layout: per profile T = ceil(M / KT) tile images and per-tile transitions
    unsigned const KT = 4u * (unsigned)c->ql_G;
    uint64_t tile_floats = 0, ttrans_floats = 0;
    c->ql_metas.assign(nprofiles, dcp_ql_prof{});
    c->any_exact_e = false;
    c->sum_core = c->sum_tiles = 0;
    c->max_tiles = 0;
    for (uint64_t &v : c->class_core)
        v = 0;
    for (unsigned i = 0; i < nprofiles; ++i)
    {
        dcp_prof_meta const &m = c->metas[i];
        dcp_ql_prof &qm = c->ql_metas[i];
        qm.core_size = m.core_size;
        qm.pidx = m.pidx;
        qm.rs_slot = i; // ql_metas and metas share one order
        qm.cls = (uint32_t)class_of(m.core_size);
        qm.ntiles = (m.core_size + KT - 1) / KT;
        c->sum_core += m.core_size;
        c->class_core[qm.cls] += m.core_size;
        c->sum_tiles += qm.ntiles;
        c->max_tiles = std::max(c->max_tiles, qm.ntiles);
        qm.needs_exact_e = (m.flags & DCP_PROF_EXACT_E) ? 1u : 0u;
        c->any_exact_e = c->any_exact_e || qm.needs_exact_e != 0u;
        qm.tile_off = c->one_layout ? m.emis_off : tile_floats; // one layout: its first column in emis_match
        qm.ldk = c->one_layout ? m.ldk : 0u;
        qm.ttrans_off = (uint32_t)ttrans_floats;
        tile_floats += (uint64_t)qm.ntiles * KT * DCP_NCODES;
        ttrans_floats += (uint64_t)qm.ntiles * (KT + 1) * 8;
    }
    if (ttrans_floats > 0xffffffffull) return c->fail(DCP_EINVAL, "DB too large for 32-bit transition offsets");
    HIP_TRY(c, c->d_ql_metas.alloc(nprofiles));
    if (!c->one_layout) HIP_TRY(c, c->d_emis_tiles.alloc(tile_floats));
    HIP_TRY(c, c->d_ttrans.alloc(ttrans_floats));
    HIP_TRY(c, hipMemcpy(c->d_ql_metas.p, c->ql_metas.data(), nprofiles * sizeof(dcp_ql_prof), hipMemcpyHostToDevice));
    {
        float const ninf = -std::numeric_limits<float>::infinity();
        std::vector<float> tt(ttrans_floats, ninf);
        for (unsigned i = 0; i < nprofiles; ++i)
        {
            dcp_ql_prof const &qm = c->ql_metas[i];
            float const *src = dcp_profile_trans8(profiles[qm.pidx]);
            unsigned const M = qm.core_size;
            for (unsigned t = 0; t < qm.ntiles; ++t)
                for (unsigned kk = 0; kk <= KT; ++kk)
                {
                    unsigned node = t * KT + kk;
                    if (node >= M) break;
                    float *dst = &tt[qm.ttrans_off + ((size_t)t * (KT + 1) + kk) * 8];
                    for (int row = 0; row < 8; ++row)
                        dst[row] = src[(size_t)row * M + node];
                }
        }
        HIP_TRY(c, hipMemcpy(c->d_ttrans.p, tt.data(), tt.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    c->qorder_q0 = c->qorder_q1 = ~0u;

    if (expand_on_host)
    {
        // parity path: tables from dcp_frame_table_host, profile by profile
        std::vector<float> col(DCP_NCODES), tab;
        std::vector<float> ins((size_t)nprofiles * DCP_NCODES), nul((size_t)nprofiles * DCP_NCODES);
        float const ninf = -std::numeric_limits<float>::infinity();
        std::vector<float> emis_host((size_t)emis_floats, ninf);
        for (unsigned i = 0; i < nprofiles; ++i)
        {
            dcp_prof_meta const &m = c->metas[i];
            dcp_profile const *pr = profiles[m.pidx];
            float eps = dcp_profile_epsilon(pr);
            tab.assign((size_t)DCP_NCODES * m.width, ninf);
            dcp_ql_prof const &qm = c->ql_metas[i];
            std::vector<float> img((size_t)qm.ntiles * KT * DCP_NCODES, ninf);
            float const *md = dcp_profile_match_dist(pr);
            for (unsigned k = 0; k < m.core_size; ++k)
            {
                dcp_frame_table_host(md + (size_t)k * DCP_NDIST, eps, col.data());
                size_t const grp = (size_t)(k / KT) * (KT / 4) + (k % KT) / 4;
                for (unsigned code = 0; code < DCP_NCODES; ++code)
                {
                    tab[(size_t)code * m.width + k] = col[code];
                    img[(grp * DCP_NCODES + code) * 4 + (k & 3u)] = col[code];
                }
            }
            // its own columns of rows it may share with other profiles: into the host image of all tables, which goes
            // to the device in one piece below (this path is the tests': small databases)
            for (unsigned code = 0; code < DCP_NCODES; ++code)
                std::memcpy(&emis_host[m.emis_off + (size_t)code * m.ldk], &tab[(size_t)code * m.width], sizeof(float) * m.width);
            if (!c->one_layout)
                HIP_TRY(c, hipMemcpy(c->d_emis_tiles.p + qm.tile_off, img.data(), img.size() * sizeof(float),
                                     hipMemcpyHostToDevice));
            dcp_frame_table_host(dcp_profile_insert_dist(pr), eps, &ins[(size_t)m.pidx * DCP_NCODES]);
            dcp_frame_table_host(dcp_profile_null_dist(pr), eps, &nul[(size_t)m.pidx * DCP_NCODES]);
        }
        HIP_TRY(c, hipMemcpy(c->d_emis_match.p, emis_host.data(), emis_host.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_emis_insert.p, ins.data(), ins.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_emis_null.p, nul.data(), nul.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    else
    {
        // dists rows: all match nodes (sorted profile order), then insert dists
        // by pidx, then null dists by pidx
        // The compact DB (129 floats per node: 1.9 GB for 20k profiles) goes to the device in the order
        // the rows have there, through two pinned 64 MB buffers: the copy of one overlaps the filling of
        // the other, and no host-side image of the whole array is built.
        size_t const rows = (size_t)match_rows + 2 * (size_t)nprofiles;
        std::vector<float> eps(rows);
        HIP_TRY(c, c->d_dists.alloc(rows * DCP_NDIST));
        StagedUpload up;
        if (int rc = up.begin(c, c->d_dists.p)) return rc;
        for (unsigned i = 0; i < nprofiles; ++i)
        {
            dcp_prof_meta const &m = c->metas[i];
            dcp_profile const *pr = profiles[m.pidx];
            if (int rc = up.append(dcp_profile_match_dist(pr), (size_t)DCP_NDIST * m.core_size)) return rc;
            std::fill(eps.begin() + dist_row[i], eps.begin() + dist_row[i] + m.core_size,
                      dcp_profile_epsilon(pr));
        }
        for (unsigned p = 0; p < nprofiles; ++p)
        {
            if (int rc = up.append(dcp_profile_insert_dist(profiles[p]), DCP_NDIST)) return rc;
            eps[(size_t)match_rows + p] = eps[(size_t)match_rows + nprofiles + p] =
                dcp_profile_epsilon(profiles[p]);
        }
        for (unsigned p = 0; p < nprofiles; ++p)
            if (int rc = up.append(dcp_profile_null_dist(profiles[p]), DCP_NDIST)) return rc;
        if (int rc = up.finish()) return rc;
        std::vector<dcp_expand_tile> tiles;
        for (unsigned i = 0; i < nprofiles; ++i)
        {
            dcp_prof_meta const &m = c->metas[i];
            for (unsigned k0 = 0; k0 < m.width; k0 += 64) // its own columns of the (possibly shared) rows
            {
                dcp_expand_tile t{};
                t.out_off = m.emis_off + k0;
                t.ncols = k0 < m.core_size ? std::min(64u, m.core_size - k0) : 0u;
                t.dist_row = t.ncols ? dist_row[i] + k0 : 0u;
                t.nstore = std::min(64u, m.width - k0);
                t.ld_code = m.ldk;
                t.ld_col = 1;
                tiles.push_back(t);
            }
        }
        size_t const n_match_tiles = tiles.size();
        for (unsigned i = 0; i < nprofiles && !c->one_layout; ++i)
        {
            dcp_prof_meta const &m = c->metas[i];
            dcp_ql_prof const &qm = c->ql_metas[i];
            unsigned const ncol = qm.ntiles * KT; // incl. -inf padding nodes
            for (unsigned k0 = 0; k0 < ncol; k0 += 64)
            {
                dcp_expand_tile t{};
                t.out_off = qm.tile_off;
                t.ncols = k0 < m.core_size ? std::min(64u, m.core_size - k0) : 0u;
                t.dist_row = t.ncols ? dist_row[i] + k0 : 0u;
                t.nstore = std::min(64u, ncol - k0);
                t.kt = KT;
                t.col0 = k0;
                tiles.push_back(t);
            }
        }
        size_t const n_image_tiles = tiles.size() - n_match_tiles;
        for (int which = 0; which < 2; ++which)
            for (unsigned p0 = 0; p0 < nprofiles; p0 += 64)
            {
                dcp_expand_tile t{};
                t.out_off = (uint64_t)p0 * DCP_NCODES;
                t.ncols = t.nstore = std::min(64u, nprofiles - p0);
                t.dist_row = (uint32_t)(match_rows + (which ? nprofiles : 0) + p0);
                t.ld_code = 1;
                t.ld_col = DCP_NCODES;
                tiles.push_back(t);
            }
        size_t const n_special_tiles = (tiles.size() - n_match_tiles - n_image_tiles) / 2;

        DevBuf<float> &d_dists = c->d_dists, &d_eps = c->d_eps; // kept for the lazy row-sweep layout
        DevBuf<dcp_expand_tile> d_tiles;
        HIP_TRY(c, d_eps.alloc(eps.size()));
        HIP_TRY(c, d_tiles.alloc(tiles.size()));
        HIP_TRY(c, hipMemcpy(d_eps.p, eps.data(), eps.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(d_tiles.p, tiles.data(), tiles.size() * sizeof(dcp_expand_tile), hipMemcpyHostToDevice));
        c->rs_tiles.assign(tiles.begin(), tiles.begin() + (long)n_match_tiles);
        dcp_expand_args ea{d_tiles.p, d_dists.p, d_eps.p, nullptr};
        if (c->one_layout)
        {
            // the row-sweep tables now, not on first use
            ea.tiles = d_tiles.p;
            ea.out = c->d_emis_match.p;
            dcp_launch_expand(&ea, (unsigned)n_match_tiles, c->stream);
        }
        else
        {
            ea.tiles = d_tiles.p + n_match_tiles;
            ea.out = c->d_emis_tiles.p;
            dcp_launch_expand(&ea, (unsigned)n_image_tiles, c->stream);
        }
        ea.tiles = d_tiles.p + n_match_tiles + n_image_tiles;
        ea.out = c->d_emis_insert.p;
        dcp_launch_expand(&ea, (unsigned)n_special_tiles, c->stream);
        ea.tiles = d_tiles.p + n_match_tiles + n_image_tiles + n_special_tiles;
        ea.out = c->d_emis_null.p;
        dcp_launch_expand(&ea, (unsigned)n_special_tiles, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->one_layout)
        {
            // nothing is rebuilt later: the compact dists go
            c->rs_tiles.clear();
            c->d_dists.release();
            c->d_eps.release();
        }
    }
    // the grouped profiles' descriptors and their merged {insert, background} tables
    if (!c->mp_groups.empty())
    {
        HIP_TRY(c, c->d_mp_groups.alloc(c->mp_groups.size()));
        HIP_TRY(c, hipMemcpy(c->d_mp_groups.p, c->mp_groups.data(), c->mp_groups.size() * sizeof(dcp_mp_group), hipMemcpyHostToDevice));
        HIP_TRY(c, c->d_mp_in.alloc((size_t)mp_in_pairs * 2u));
        dcp_launch_mp_in(c->d_mp_groups.p, (unsigned)c->mp_groups.size(), c->mp_first[1], kMpParts[0], kMpParts[1],
                         c->d_emis_insert.p, c->d_emis_null.p, c->d_mp_in.p, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->rs_ready = expand_on_host != 0 || c->one_layout;
    c->nprof = nprofiles;
    return DCP_OK;
}

// Expand the row-sweep layout of the match tables ([1364][ldk] per profile) if it is not resident yet.
static int ensure_rowsweep_layout(dcp_gpu_ctx *c)
{
    if (c->rs_ready) return DCP_OK;
    if (c->rs_tiles.empty() || !c->d_dists.p) return c->fail(DCP_EFAIL, "row-sweep layout cannot be rebuilt");
    HIP_TRY(c, c->d_emis_match.alloc(c->rs_floats));
    DevBuf<dcp_expand_tile> d_tiles;
    HIP_TRY(c, d_tiles.alloc(c->rs_tiles.size()));
    HIP_TRY(c, hipMemcpy(d_tiles.p, c->rs_tiles.data(), c->rs_tiles.size() * sizeof(dcp_expand_tile), hipMemcpyHostToDevice));
    dcp_expand_args ea{d_tiles.p, c->d_dists.p, c->d_eps.p, c->d_emis_match.p};
    dcp_launch_expand(&ea, (unsigned)c->rs_tiles.size(), c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->rs_ready = true;
    return DCP_OK;
}

int dcp_gpu_db_one_layout(dcp_gpu_ctx const *c) { return c && c->one_layout ? 1 : 0; }
uint64_t dcp_gpu_db_table_bytes(dcp_gpu_ctx const *c)
{
    if (!c || c->nprof == 0) return 0;
    // the lazily expanded row-sweep tables count once they are there
    bool const lazy_there = !c->one_layout && c->rs_ready && c->d_dists.p;
    return c->table_bytes + (lazy_there ? c->rs_floats * sizeof(float) : 0);
}

int dcp_gpu_db_fetch_match_table(dcp_gpu_ctx *c, unsigned p, float *out)
{
    if (!c || !out || p >= c->nprof) return DCP_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure_rowsweep_layout(c)) return rc;
    for (dcp_prof_meta const &m : c->metas)
        if (m.pidx == p)
        {
            // rows of (possibly shared) length ldk from the profile's first column to its last row's last node
            std::vector<float> tab((size_t)(DCP_NCODES - 1) * m.ldk + m.core_size);
            HIP_TRY(c, hipMemcpy(tab.data(), c->d_emis_match.p + m.emis_off, tab.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (unsigned code = 0; code < DCP_NCODES; ++code)
                std::memcpy(out + (size_t)code * m.core_size, &tab[(size_t)code * m.ldk], sizeof(float) * m.core_size);
            return DCP_OK;
        }
    return DCP_EINVAL;
}

// ---------------------------------------------------------------------------
// Sequences
// ---------------------------------------------------------------------------
static int upload_seqs(dcp_gpu_ctx *c, uint8_t const *seqs, uint32_t const *seq_off,
                       unsigned nseqs, bool text)
{
    if (!c) return DCP_EINVAL;
    if (!seqs || !seq_off || nseqs == 0) return c->fail(DCP_EINVAL, "empty sequence batch");
    HIP_TRY(c, hipSetDevice(c->device));
    c->scanned = false;
    c->redo_pending = false;           // the previous batch's scan is void
    c->ring_check_pending = false;
    c->qorder_q0 = c->qorder_q1 = ~0u; // also when this upload fails half way
    std::vector<uint32_t> woff(nseqs), len(nseqs);
    uint64_t nwords = 0, total = 0;
    for (unsigned q = 0; q < nseqs; ++q)
    {
        if (seq_off[q + 1] <= seq_off[q])
            return c->fail(DCP_EINVAL, "sequence cannot be empty"); // protein_profile.c:158
        uint32_t L = seq_off[q + 1] - seq_off[q];
        woff[q] = (uint32_t)nwords;
        len[q] = L;
        nwords += L / 16 + 3; // pad: the kernels look ahead by up to one word + two bases
        total += L;
        if (nwords > 0xffffffffull) return c->fail(DCP_EINVAL, "sequence batch too large");
    }
    std::vector<uint32_t> words(nwords, 0u);
    for (unsigned q = 0; q < nseqs; ++q)
    {
        uint8_t const *s = seqs + seq_off[q];
        uint32_t *w = &words[woff[q]];
        for (uint32_t i = 0; i < len[q]; ++i)
        {
            unsigned b = s[i];
            if (text)
            {
                switch (s[i])
                {
                case 'A': b = 0; break;
                case 'C': b = 1; break;
                case 'G': b = 2; break;
                case 'T': b = 3; break;
                default: b = 255; break;
                }
            }
            if (b > 3)
                return c->fail(DCP_EINVAL, "sequence %u: symbol at %u is outside ACGT", q, i);
            w[i >> 4] |= b << ((i & 15u) * 2u);
        }
    }
    // a failed (re)allocation below must not leave the old batch's size paired with new buffers
    c->nseqs = 0;
    c->total_len = 0;
    c->seq_len.clear();
    HIP_TRY(c, c->d_seq_words.alloc(nwords));
    HIP_TRY(c, c->d_seq_woff.alloc(nseqs));
    HIP_TRY(c, c->d_seq_len.alloc(nseqs));
    HIP_TRY(c, c->d_xtrans.alloc((size_t)nseqs * DCP_XSTRIDE));
    HIP_TRY(c, hipMemcpy(c->d_seq_words.p, words.data(), nwords * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_seq_woff.p, woff.data(), nseqs * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_seq_len.p, len.data(), nseqs * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->seq_len = len;
    c->nseqs = nseqs;
    c->total_len = total;
    c->xt_multi = c->xt_h3 = -1;
    c->xt_explicit = false;
    // the length order and the transposed word planes belong to the batch that was resident
    c->qorder_q0 = c->qorder_q1 = ~0u;
    return DCP_OK;
}

int dcp_gpu_seqs_upload(dcp_gpu_ctx *c, uint8_t const *seqs, uint32_t const *seq_off, unsigned nseqs)
{
    return upload_seqs(c, seqs, seq_off, nseqs, false);
}

int dcp_gpu_seqs_upload_text(dcp_gpu_ctx *c, char const *text, uint32_t const *seq_off, unsigned nseqs)
{
    return upload_seqs(c, (uint8_t const *)text, seq_off, nseqs, true);
}

// ---------------------------------------------------------------------------
// Scan
// ---------------------------------------------------------------------------
static int finish_scan(dcp_gpu_ctx *c);
// protein_profile_setup once per sequence (length) instead of once per pair
static int ensure_xtrans(dcp_gpu_ctx *c, int multi_hits, int hmmer3_compat)
{
    if (c->xt_explicit) return DCP_OK;
    if (c->xt_multi == !!multi_hits && c->xt_h3 == !!hmmer3_compat) return DCP_OK;
    std::vector<float> xt((size_t)c->nseqs * DCP_XSTRIDE, 0.0f);
    std::map<uint32_t, std::vector<float>> by_len;
    for (unsigned q = 0; q < c->nseqs; ++q)
    {
        auto it = by_len.find(c->seq_len[q]);
        if (it == by_len.end())
        {
            std::vector<float> v(DCP_XSTRIDE, 0.0f);
            int rc = dcp_xtrans(c->seq_len[q], multi_hits, hmmer3_compat, v.data());
            if (rc) return c->fail(rc, "sequence cannot be empty");
            it = by_len.emplace(c->seq_len[q], std::move(v)).first;
        }
        std::memcpy(&xt[(size_t)q * DCP_XSTRIDE], it->second.data(), sizeof(float) * DCP_XSTRIDE);
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_xtrans.p, xt.data(), xt.size() * sizeof(float),
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->xt_multi = !!multi_hits;
    c->xt_h3 = !!hmmer3_compat;
    return DCP_OK;
}

int dcp_gpu_seqs_set_xtrans(dcp_gpu_ctx *c, float const *xt, unsigned nseqs)
{
    if (!c) return DCP_EINVAL;
    if (!xt || nseqs == 0 || nseqs != c->nseqs) return c->fail(DCP_EINVAL, "xtrans must cover the resident sequences");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<float> buf((size_t)nseqs * DCP_XSTRIDE, 0.0f);
    for (unsigned q = 0; q < nseqs; ++q)
        for (int i = 0; i < DCP_NXTRANS; ++i)
        {
            float const v = xt[(size_t)q * DCP_NXTRANS + i];
            uint32_t bits; // this file is built with -fno-honor-nans: test the encoding, not v != v
            std::memcpy(&bits, &v, sizeof bits);
            if ((bits & 0x7fffffffu) > 0x7f800000u) return c->fail(DCP_EINVAL, "NaN special transition");
            buf[(size_t)q * DCP_XSTRIDE + i] = v;
        }
    HIP_TRY(c, hipMemcpy(c->d_xtrans.p, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
    c->xt_explicit = true;
    c->xt_multi = c->xt_h3 = -1;
    return DCP_OK;
}

int dcp_gpu_scan(dcp_gpu_ctx *c, struct dcp_scan_params const *prm)
{
    if (!c) return DCP_EINVAL;
    return dcp_gpu_scan_range(c, prm, 0, c->nseqs);
}

int dcp_gpu_set_hit_buffer(dcp_gpu_ctx *c, void *hits_dev, unsigned cap, void *nhits_dev)
{
    if (!c) return DCP_EINVAL;
    if ((hits_dev == nullptr) != (nhits_dev == nullptr) || (hits_dev && cap == 0))
        return c->fail(DCP_EINVAL, "hit buffer and counter must be given together");
    c->ext_hits = (dcp_hit *)hits_dev;
    c->ext_nhits = (unsigned *)nhits_dev;
    c->ext_cap = hits_dev ? cap : 0;
    return DCP_OK;
}

int dcp_gpu_hit_buffer(dcp_gpu_ctx *c, void **hits_dev, void **nhits_dev, unsigned *cap)
{
    if (!c || !hits_dev || !nhits_dev || !cap) return DCP_EINVAL;
    if (!c->scanned) return c->fail(DCP_EINVAL, "no scan yet");
    *hits_dev = c->ext_hits ? (void *)c->ext_hits : (void *)c->d_hits.p;
    *nhits_dev = c->ext_hits ? (void *)c->ext_nhits : (void *)c->d_nhits.p;
    *cap = c->ext_hits ? c->ext_cap : c->hit_cap;
    return DCP_OK;
}

// ---- dynamic batching of the query-lane kernels (BASELINE configs[4]) ----------------------------
// The queries of a scan, sorted by length, are cut into GROUPS of 64 consecutive ones -- what one wavefront sweeps
// together, one lane each, for as many rows as its longest member has.  A block has `slots` wavefront slots (4: the
// 256 lanes of a stage; 1 for the three-independent-wavefronts variant) and sits through a tile until its slowest
// slot is done, so a block of four CONSECUTIVE groups (rounds 1-3) wastes (longest - mean) of its slots' time:
// a 1 000-query batch of 100 nt .. 10 kbp ran at 0.6 of the rate of a uniform one.  Here each slot holds a LIST of
// groups, swept one after the other per tile, and the groups are packed into slots so that all slots of all blocks
// carry about the same number of rows (longest-processing-time-first; the block count is the one that wastes least).
// Uniform batches come out as before: one group per slot.
struct QlPlan
{
    std::vector<dcp_ql_group> groups;  // in slot order
    std::vector<uint32_t> slot_first;  // [nqb * slots + 1]
    std::vector<uint32_t> block_rows;  // [nqb] rows of the block's planes: its longest slot
    unsigned nqb = 0;
    unsigned plane_rows = 0;           // max over blocks
    uint64_t sum_block_rows = 0;       // what a tile costs, summed over the blocks: the cost model's figure
    unsigned busy_slots = 0;           // slots of the first block that hold a group (cost model)
};
// len_sorted: the lengths in ascending order (entry i belongs to qorder[i]).
static QlPlan plan_query_groups(unsigned const *len_sorted, unsigned nq, unsigned slots)
{
    QlPlan pl;
    unsigned const ng = (nq + 63u) / 64u;
    struct G
    {
        unsigned first, n, lmax, rows;
    };
    std::vector<G> gs(ng);
    uint64_t sum_rows = 0;
    unsigned max_rows = 0;
    // full groups from the LONG end: if nq is not a multiple of 64 the partial group is the one of the shortest
    // queries -- idle lanes then cost the fewest rows (the other way round, a 965-query pass of 100 nt .. 10 kbp spent
    // a slot's 10 000 rows on the five longest queries: profiles/r04/host_scan_probe.txt)
    unsigned const rem = nq % 64u;
    for (unsigned g = 0; g < ng; ++g)
    {
        unsigned const first = rem == 0u ? g * 64u : (g == 0u ? 0u : rem + (g - 1u) * 64u);
        unsigned const n = rem != 0u && g == 0u ? rem : 64u;
        unsigned const lmax = len_sorted[first + n - 1u];
        gs[g] = G{first, n, lmax, dcp_qlane_group_rows(lmax)};
        sum_rows += gs[g].rows;
        max_rows = std::max(max_rows, gs[g].rows);
    }
    // longest first into the least loaded slot; slots sorted by load, `slots` consecutive ones make a block
    auto pack = [&](unsigned nqb, std::vector<std::vector<unsigned>> &slot_groups, std::vector<uint64_t> &load) {
        unsigned const S = nqb * slots;
        slot_groups.assign(S, {});
        load.assign(S, 0);
        typedef std::pair<uint64_t, unsigned> LS; // (load, slot): least loaded on top, lowest slot first on ties
        std::priority_queue<LS, std::vector<LS>, std::greater<LS>> heap;
        for (unsigned i = 0; i < S; ++i)
            heap.push(LS(0, i));
        for (unsigned g = ng; g-- > 0;) // ascending lengths: the last group is the longest
        {
            LS top = heap.top();
            heap.pop();
            slot_groups[top.second].push_back(g);
            load[top.second] += gs[g].rows;
            heap.push(LS(load[top.second], top.second));
        }
        std::vector<unsigned> order(S);
        for (unsigned i = 0; i < S; ++i)
            order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return load[x] > load[y]; });
        std::vector<std::vector<unsigned>> sg(S);
        std::vector<uint64_t> ld(S);
        for (unsigned i = 0; i < S; ++i)
            sg[i] = std::move(slot_groups[order[i]]), ld[i] = load[order[i]];
        slot_groups.swap(sg), load.swap(ld);
        uint64_t cost = 0;
        for (unsigned b = 0; b < nqb; ++b)
            cost += load[(size_t)b * slots]; // the block's longest slot
        return cost;
    };
    // candidate block counts around (all rows) / (slots x the longest group): fewer blocks balance better, more blocks
    // keep more tasks in flight; take the cheapest, the larger count on ties
    unsigned const cap = (ng + slots - 1u) / slots; // one group per slot: more blocks than that only adds empty slots
    uint64_t const denom = (uint64_t)slots * max_rows;
    unsigned const lo = (unsigned)std::min<uint64_t>(std::max<uint64_t>(sum_rows / denom, 1), cap);
    unsigned const hi = (unsigned)std::min<uint64_t>(std::max<uint64_t>((sum_rows + denom - 1u) / denom, 1), cap);
    std::vector<std::vector<unsigned>> best_sg, sg;
    std::vector<uint64_t> best_ld, ld;
    uint64_t best_cost = ~0ull;
    unsigned best_nqb = 0;
    for (unsigned nqb = lo; nqb <= hi; ++nqb)
    {
        uint64_t const cost = pack(nqb, sg, ld);
        if (cost <= best_cost) best_cost = cost, best_nqb = nqb, best_sg.swap(sg), best_ld.swap(ld);
    }
    pl.nqb = best_nqb;
    pl.sum_block_rows = best_cost;
    pl.slot_first.assign((size_t)best_nqb * slots + 1u, 0);
    pl.block_rows.assign(best_nqb, 0);
    for (unsigned sidx = 0; sidx < best_nqb * slots; ++sidx)
    {
        pl.slot_first[sidx] = (uint32_t)pl.groups.size();
        uint32_t rowbase = 0;
        for (unsigned g : best_sg[sidx])
        {
            pl.groups.push_back(dcp_ql_group{gs[g].first, gs[g].n, rowbase, gs[g].lmax});
            rowbase += gs[g].rows;
        }
        pl.block_rows[sidx / slots] = std::max<uint32_t>(pl.block_rows[sidx / slots], rowbase);
        if (sidx < slots && rowbase) ++pl.busy_slots;
    }
    pl.slot_first.back() = (uint32_t)pl.groups.size();
    for (unsigned b = 0; b < best_nqb; ++b)
        pl.plane_rows = std::max(pl.plane_rows, pl.block_rows[b]);
    return pl;
}

// The plan as plain arrays (host only; tests/test_query_slots.py checks its invariants on the CPU).
extern "C" int dcp_plan_query_slots(unsigned const *len_sorted, unsigned nq, unsigned slots_per_block, unsigned *nblocks,
                                    unsigned long long *sum_block_rows, unsigned *plane_rows, unsigned *groups4,
                                    unsigned group_cap, unsigned *slot_first, unsigned slot_cap)
{
    if (!len_sorted || nq == 0 || (slots_per_block != 1u && slots_per_block != 4u) || !nblocks) return DCP_EINVAL;
    for (unsigned i = 1; i < nq; ++i)
        if (len_sorted[i] < len_sorted[i - 1]) return DCP_EINVAL;
    QlPlan const pl = plan_query_groups(len_sorted, nq, slots_per_block);
    *nblocks = pl.nqb;
    if (sum_block_rows) *sum_block_rows = pl.sum_block_rows;
    if (plane_rows) *plane_rows = pl.plane_rows;
    if (groups4)
    {
        if (group_cap < pl.groups.size()) return DCP_ENOMEM;
        std::memcpy(groups4, pl.groups.data(), pl.groups.size() * sizeof(dcp_ql_group));
    }
    if (slot_first)
    {
        if (slot_cap < pl.slot_first.size()) return DCP_ENOMEM;
        std::memcpy(slot_first, pl.slot_first.data(), pl.slot_first.size() * sizeof(uint32_t));
    }
    return DCP_OK;
}

// Which grid-mode row-sweep kernel scores one size class against `nchunks` queries: rows of the emission table
// each block stages in LDS and wavefronts per block (dcp_kernels.hip).  Measured on the 20 000-profile DB
// (profiles/r03/rowsweep_variants.txt, ms per scan for 1 .. 1 000 queries):
//   * small batches stream the tables from HBM (an XCD's 512 wavefront slots work on far more profiles than its
//     4 MB L2 holds tables of), so what a block keeps in LDS is traffic saved: with the three-base rows staged
//     as well, 8 queries take 49 ms instead of 57, 16: 80 instead of 99, 32: 142 instead of 160 -- provided the
//     block is wide enough that the 21.5 KB x R image does not cost wavefront slots (one profile per block);
//   * from about 40 queries on the L2 serves the rows and the 20-row image with narrower blocks wins (64 queries:
//     259 vs 270 ms); wide blocks start and drain together, which a long launch pays for (1 000 queries: 3.65 s
//     with 4 wavefronts per block, 3.83 with 8, 4.14 with 16).
static void rowsweep_variant(dcp_gpu_ctx const *c, int R, int W, unsigned nchunks, int *stg, unsigned *bw, int *pf)
{
    *pf = 0;
    if (W != 1)
    {
        *stg = 0, *bw = 1;
        return;
    }
    // blocks of equal width: 24 queries are two blocks of 12 wavefronts, not 16 + 8
    auto balanced = [&](unsigned maxw) {
        unsigned const nb = (nchunks + maxw - 1u) / maxw;
        return (nchunks + nb - 1u) / nb;
    };
    int g = 20;
    unsigned w = balanced(nchunks <= 56u ? 8u : 4u); // 48 queries: 198 ms with 8 wavefronts per block, 205 with 4; 64: 256 / 252
    unsigned const max84 = dcp_rowsweep_max_block_waves(R, W, 84);
    if (nchunks >= 6u && nchunks <= 36u && max84 != 0u)
    {
        unsigned const w84 = balanced(max84);
        unsigned const slots = dcp_rowsweep_max_block_waves(R, W, 84); // wavefronts of this class a CU runs at the kernel's register count
        unsigned const blocks = (160u * 1024u) / dcp_rowsweep_stage_bytes(R, 84);
        // the image may cost up to half of them: such a batch waits for HBM, not for issue slots (8 queries with
        // the R = 4 class at 8 of 16 wavefronts per CU: 49 ms; with that class on the 20-row image: 51)
        if (std::min(blocks * w84, slots) * 2u >= slots) g = 84, w = w84;
    }
    bool const forced = c->rs_force_stg >= 0 && (c->rs_force_R == 0 || c->rs_force_R == R) &&
                        dcp_rowsweep_max_block_waves(R, W, c->rs_force_stg) != 0u;
    if (forced)
    {
        g = c->rs_force_stg;
        w = g == 0 ? 4u : std::min(std::min(c->rs_force_bw ? c->rs_force_bw : 4u, dcp_rowsweep_max_block_waves(R, W, g)), nchunks);
    }
    *stg = g, *bw = std::max(1u, w);
    // two rows of prefetch: the batches that wait for HBM (the 84-row variants' range, and below it)
    *pf = g > 0 && nchunks <= 36u;
    if (forced && g > 0) *pf = c->rs_force_pf;
}

int dcp_gpu_scan_range(dcp_gpu_ctx *c, struct dcp_scan_params const *prm, unsigned q_begin,
                       unsigned q_end)
{
    if (!c || !prm) return DCP_EINVAL;
    if (c->nprof == 0) return c->fail(DCP_EINVAL, "no profile DB resident");
    if (c->nseqs == 0) return c->fail(DCP_EINVAL, "no sequences resident");
    if (q_begin >= q_end || q_end > c->nseqs) return c->fail(DCP_EINVAL, "bad sequence range");
    HIP_TRY(c, hipSetDevice(c->device));
    // One scan is outstanding per context: results (hits, scores, redo lists) are those of the LAST
    // scan.  A scan enqueued while the previous one still has unchecked redo lists first completes
    // that one (its overflow re-run included), so nothing of it is silently half done.
    if (c->redo_pending || c->ring_check_pending)
        if (int rc = finish_scan(c)) return rc;

    if (int rc = ensure_xtrans(c, prm->multi_hits, prm->hmmer3_compat)) return rc;

    size_t const npairs = (size_t)c->nseqs * c->nprof;
    if (prm->keep_scores)
    {
        if (c->d_null.n != npairs)
        {
            HIP_TRY(c, c->d_null.alloc(npairs));
            HIP_TRY(c, c->d_alt.alloc(npairs));
        }
    }
    c->have_scores = prm->keep_scores != 0;
    unsigned want_cap = (unsigned)std::min<size_t>(npairs, (size_t)1 << 22);
    if (!c->ext_hits && c->hit_cap < want_cap)
    {
        HIP_TRY(c, c->d_hits.alloc(want_cap));
        c->hit_cap = want_cap;
    }
    if (!c->d_nhits.p) HIP_TRY(c, c->d_nhits.alloc(1));
    dcp_hit *const hits_p = c->ext_hits ? c->ext_hits : c->d_hits.p;
    unsigned *const nhits_p = c->ext_hits ? c->ext_nhits : c->d_nhits.p;
    unsigned const hits_cap = c->ext_hits ? c->ext_cap : c->hit_cap;

    dcp_scan_args a{};
    a.profs = c->d_metas.p;
    a.emis_match = c->d_emis_match.p;
    a.emis_insert = c->d_emis_insert.p;
    a.emis_null = c->d_emis_null.p;
    a.trans8 = c->d_trans8.p;
    a.seq_words = c->d_seq_words.p;
    // the kernels index sequences relative to q_begin
    unsigned const nq = q_end - q_begin;
    a.seq_woff = c->d_seq_woff.p + q_begin;
    a.seq_len = c->d_seq_len.p + q_begin;
    a.xtrans = c->d_xtrans.p + (size_t)q_begin * DCP_XSTRIDE;
    a.out_null = c->have_scores ? c->d_null.p + (size_t)q_begin * c->nprof : nullptr;
    a.out_alt = c->have_scores ? c->d_alt.p + (size_t)q_begin * c->nprof : nullptr;
    a.hits = hits_p;
    a.nhits = nhits_p;
    a.hit_cap = hits_cap;
    a.lrt_threshold = prm->lrt_threshold;
    a.nprof_total = c->nprof;
    a.nseqs = nq;
    a.q_base = q_begin;
    // Queries a row-sweep wavefront scores one after the other with the profile's transitions in registers:
    // ONE.  More (8 per task in round 1) saves the reload of 8 transitions per node and nothing else, but
    // leaves fewer concurrent tasks per profile, so that an XCD works on more profiles at a time than its
    // 4 MB L2 holds tables of (C3 step: 848 -> 879 Gcell/s from 8 to 1; 25 profiles x 256 queries: 8.8 -> 1.8 ms).
    a.qchunk = 1u;
    a.nchunks = (nq + a.qchunk - 1) / a.qchunk;
    c->last_q0 = q_begin;
    c->last_q1 = q_end;

    // Queries per block of the single-stage query-lane kernel.  (Narrow 64- / 128-query blocks were tried for
    // small batches: hipOccupancyMaxActiveBlocksPerMultiprocessor reports three 54.5 KB blocks per CU, the
    // measured rate is that of two -- profiles/r02/latency_probe_narrow_blocks_attempt.txt -- so they are gone.)
    unsigned ql_nt = dcp_qlane_block_size(), ql_blocks_per_cu = 2; // 2 x 54.5 KB of LDS, 2 x 4 wavefronts of 256 VGPRs
    // up to 64 queries: the three-independent-wavefronts variant (64 queries per task, 3 busy wavefronts per CU)
    // (since the row sweep went from 370 to 850 Gcell/s the automatic choice no longer reaches it:
    // 64 queries 283 ms there against 330 ms here; it is what kernel = 2 runs for such batches)
    bool const w3 = nq <= 64u;
    if (w3) ql_nt = 64u, ql_blocks_per_cu = 3; // counted in wavefront slots
    // kernel choice: the query-lane kernel needs enough queries to fill its lanes
    int kernel = prm->kernel;
    if (kernel == 0)
    {
        {
            // Cost model fitted to profiles/latency_probe.py and profiles/smalldb_probe.py:
            //   row sweep    cells of each size class / that class's rate (kClassRate: 0.5-1.05 Tcell/s with one
            //                wavefront per pair, 0.4-0.6 with 4-16) + one pass over the emission tables at
            //                3 TB/s (a small batch streams them from HBM: 20k profiles, 16 queries: 103 ms =
            //                67 + 36) + the last task's row chain
            //   query lane   max(longest task, all tile rows / resident blocks); a tile row of a block
            //                takes 0.52 us with one busy wavefront, 0.73 us with four.
            // On the 20k-profile DB the switch comes at about 150 queries since round 3 (128 queries: 484 ms in
            // the row sweep, 480 in the single-stage query-lane kernel; round 2: at about 115); a DB of a few
            // hundred profiles stays with the row sweep up to several hundred queries (its tasks cannot fill the grid).
            unsigned const NTq = ql_nt;
            std::vector<unsigned> len(c->seq_len.begin() + q_begin, c->seq_len.begin() + q_end);
            std::sort(len.begin(), len.end());
            double sum_len = 0;
            for (unsigned L : len)
                sum_len += L;
            // rows a tile costs, summed over the blocks of wavefront slots the batch is packed into (plan_query_groups)
            QlPlan const plan = plan_query_groups(len.data(), nq, NTq / 64u);
            unsigned const nqb = plan.nqb;
            double const sum_block_lmax = (double)plan.sum_block_rows;
            unsigned const lmax = len.back();
            double t_rs = (double)c->sum_core * (DCP_NCODES * 4.0) / 3e12 + a.qchunk * lmax * 1.2e-6; // 1.2 us per row
            // (the class rates are those of a 1 000-query step; between 64 and 256 queries the row sweep runs 5 %
            // below them -- 128 queries 450 ms, 160: 558 -- profiles/r03/switch_probe.txt)
            for (int k = 0; k < kNumClasses; ++k)
                t_rs += 1.05 * (double)c->class_core[k] * sum_len / kClassRate[k];
            unsigned const waves = std::max(1u, std::min(4u, plan.busy_slots));
            // (round 3, 20k-profile DB: 96 and 128 queries 477 / 480 ms, 256 queries 659 ms in the single-stage kernel)
            static double const kTrow[4] = {0.52, 0.54, 0.62, 0.73}; // 144..191 queries: 544-555 ms
            double const trow = (w3 ? 0.56 : kTrow[waves - 1u]) * 1e-6; // w3: + one add per gather
            double const resident = (double)std::min<uint64_t>((uint64_t)c->nprof * nqb, (uint64_t)ql_blocks_per_cu * c->num_cus);
            double const t_ql = std::max((double)c->max_tiles * lmax * 0.52e-6,
                                         (double)c->sum_tiles * sum_block_lmax * trow / resident) +
                                1e-4; // its redo launches
            // The two-stage variant (512-thread blocks, two tiles of a profile in flight) wins once the
            // query blocks are mostly full -- measured on the C3 DB (profiles/r02/latency_probe.txt):
            // 256 queries 641 vs 678 ms, 1024 queries 2508 vs 2657 ms (0.94); below that its idle
            // wavefronts still sit through every barrier (128 queries 583 vs 514 ms).  With fewer tasks
            // than blocks fit on the chip it takes 0.66 of the single-stage time (profiles/smalldb_probe.py).
            bool const stage2 = nq > 192u; // 192 queries (three full wavefronts per block): 555 ms single-stage, 569 two-stage; 224: 654 / 628
            double const fill = std::min(1.0, (double)c->nprof * nqb / (4.0 * c->num_cus));
            double const t_q = stage2 ? t_ql * (0.66 + 0.28 * fill) : t_ql;
            kernel = t_q < t_rs ? (stage2 ? 3 : 2) : 1;
            // a batch with very long sequences cannot keep enough blocks resident: row sweep instead
            if (lmax > 200000u) kernel = 1;
        }
    }
    if (kernel < 1 || kernel > 3) return c->fail(DCP_EINVAL, "unknown kernel %d", kernel);
    bool const two_stage = kernel == 3; // the query-lane kernel's two-stage variant (dcp_qlane.hip)
    c->last_kernel_variant = kernel;
    if (two_stage) kernel = 2;
    c->last_kernel = kernel;
    if (two_stage) ql_nt = dcp_qlane_block_size(), ql_blocks_per_cu = 1; // 2 x 256 queries' wavefronts, one block per CU
    bool const use_w3 = w3 && !two_stage;
    if (kernel == 2)
    {
        // queries sorted by length, cut into 64-query groups, the groups packed into wavefront slots (plan_query_groups)
        if (c->qorder_q0 != q_begin || c->qorder_q1 != q_end || c->qorder_nt != ql_nt)
        {
            unsigned const NTq = ql_nt, slots = NTq / 64u;
            std::vector<uint32_t> ord(nq);
            for (unsigned i = 0; i < nq; ++i)
                ord[i] = i;
            std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) {
                return c->seq_len[q_begin + x] < c->seq_len[q_begin + y];
            });
            std::vector<unsigned> len_sorted(nq);
            for (unsigned i = 0; i < nq; ++i)
                len_sorted[i] = c->seq_len[q_begin + ord[i]];
            QlPlan const plan = plan_query_groups(len_sorted.data(), nq, slots);
            unsigned const nqb = plan.nqb;
            if ((uint64_t)plan.plane_rows * NTq * 4u > 0xffffffffull) // 32-bit byte offsets into a block's planes
                return c->fail(DCP_ENOMEM, "sequences too long for the query-lane kernel (%u plane rows): use kernel = 1", plan.plane_rows);
            // pinned staging: [nq] order, [nqb + 1] plane offsets, [nslots + 1] slot table, 4 words per group
            size_t const nslot_words = (size_t)nqb * slots + 1u, ngroup_words = plan.groups.size() * 4u;
            size_t const need_stage = (size_t)nq + nqb + 1u + nslot_words + ngroup_words;
            if (!c->ev_qstage) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_qstage, hipEventDisableTiming));
            else HIP_TRY(c, hipEventSynchronize(c->ev_qstage)); // the previous layout's copies have left the buffer
            if (c->h_qstage_n < need_stage)
            {
                if (c->h_qstage) (void)hipHostFree(c->h_qstage);
                c->h_qstage = nullptr, c->h_qstage_n = 0;
                HIP_TRY(c, hipHostMalloc((void **)&c->h_qstage, need_stage * sizeof(uint32_t), hipHostMallocDefault));
                c->h_qstage_n = need_stage;
            }
            uint32_t *const h_ord = c->h_qstage, *const wt_off = h_ord + nq, *const h_slots = wt_off + nqb + 1u,
                           *const h_groups = h_slots + nslot_words;
            static_assert(sizeof(dcp_ql_group) == 4 * sizeof(uint32_t), "group records travel as four words");
            std::memcpy(h_ord, ord.data(), nq * sizeof(uint32_t));
            std::memcpy(h_slots, plan.slot_first.data(), nslot_words * sizeof(uint32_t));
            std::memcpy(h_groups, plan.groups.data(), ngroup_words * sizeof(uint32_t));
            // per block: its window plane, uint16 [block rows + 8][NTq] (the prefetch runs a few rows past the end)
            uint64_t tot = 0;
            wt_off[0] = 0u;
            for (unsigned b = 0; b < nqb; ++b)
            {
                tot += ((uint64_t)plan.block_rows[b] + 8u) * NTq / 2u;
                if (tot > 0xffffffffull) return c->fail(DCP_EINVAL, "sequence batch too large");
                wt_off[b + 1u] = (uint32_t)tot;
            }
            if (c->d_qorder.n < nq) HIP_TRY(c, c->d_qorder.alloc(nq));
            if (c->d_wt_off.n < nqb + 1u) HIP_TRY(c, c->d_wt_off.alloc(nqb + 1u));
            if (c->d_slot_first.n < nslot_words) HIP_TRY(c, c->d_slot_first.alloc(nslot_words));
            if (c->d_ql_groups.n < plan.groups.size()) HIP_TRY(c, c->d_ql_groups.alloc(plan.groups.size()));
            if (c->d_words_t.n < tot) HIP_TRY(c, c->d_words_t.alloc((size_t)tot));
            HIP_TRY(c, hipMemcpyAsync(c->d_qorder.p, h_ord, nq * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->d_wt_off.p, wt_off, (nqb + 1u) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->d_slot_first.p, h_slots, nslot_words * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->d_ql_groups.p, h_groups, ngroup_words * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipEventRecord(c->ev_qstage, c->stream));
            dcp_qlane_args ta{};
            ta.seq_words = a.seq_words, ta.seq_woff = a.seq_woff, ta.seq_len = a.seq_len;
            ta.qorder = c->d_qorder.p, ta.words_t = c->d_words_t.p, ta.wt_off = c->d_wt_off.p;
            ta.groups = c->d_ql_groups.p, ta.slot_first = c->d_slot_first.p;
            ta.nseqs = nq, ta.nqblocks = nqb;
            if (dcp_launch_qlane_transpose(&ta, NTq, c->stream)) return c->fail(DCP_EFAIL, "no kernel for %u-query blocks", NTq);
            HIP_TRY(c, hipGetLastError());
            c->qorder_q0 = q_begin, c->qorder_q1 = q_end, c->qorder_nt = NTq;
            c->ql_nqb = nqb, c->ql_plane_rows = plan.plane_rows + 8u;
        }
        if (!c->d_task_counter.p) HIP_TRY(c, c->d_task_counter.alloc(1));
    }

    HIP_TRY(c, hipMemsetAsync(nhits_p, 0, sizeof(unsigned), c->stream));
    c->last_launches = 0;
    c->n_launched = 0;
    if (kernel == 2)
    {
        dcp_qlane_args qa{};
        qa.profs = c->d_ql_metas.p;
        qa.emis_tiles = c->one_layout ? c->d_emis_match.p : c->d_emis_tiles.p;
        qa.tiles_from_rows = c->one_layout ? 1u : 0u;
        qa.emis_insert = c->d_emis_insert.p;
        qa.emis_null = c->d_emis_null.p;
        qa.ttrans = c->d_ttrans.p;
        qa.seq_words = a.seq_words;
        qa.seq_woff = a.seq_woff;
        qa.seq_len = a.seq_len;
        qa.xtrans = a.xtrans;
        qa.qorder = c->d_qorder.p;
        qa.words_t = c->d_words_t.p;
        qa.wt_off = c->d_wt_off.p;
        qa.groups = c->d_ql_groups.p;
        qa.slot_first = c->d_slot_first.p;
        qa.ring_stall = c->ring_stall;
        qa.task_counter = c->d_task_counter.p;
        qa.out_null = a.out_null;
        qa.out_alt = a.out_alt;
        qa.hits = hits_p;
        qa.nhits = nhits_p;
        qa.hit_cap = hits_cap;
        qa.lrt_threshold = prm->lrt_threshold;
        qa.nprof = c->nprof;
        qa.nprof_total = c->nprof;
        qa.nseqs = nq;
        qa.q_base = q_begin;
        qa.plane_rows = c->ql_plane_rows;
        unsigned const NT = ql_nt;
        qa.nqblocks = c->ql_nqb;
        uint64_t const ntasks = (uint64_t)c->nprof * qa.nqblocks;
        if (ntasks > 0xffffffffull) return c->fail(DCP_EINVAL, "scan too large for one launch");
        qa.ntasks = (unsigned)ntasks;
        // redo lists: pairs whose multi-hit feedback beat B0 go to the row-sweep kernel, which
        // runs right behind the query-lane kernel on this stream (uni-hit scans have no feedback)
        // (explicit transitions may carry E->B feedback; flagged profiles leave the kernel through the lists too)
        bool const redo = prm->multi_hits != 0 || c->xt_explicit || (dcp_qlane_exact_e_by_redo() && c->any_exact_e);
        unsigned redo_grid[kNumClasses] = {0};
        if (redo)
        {
            if (int rc = ensure_rowsweep_layout(c)) return rc;
            a.emis_match = c->d_emis_match.p;
            uint64_t tot = 0;
            uint64_t const cap_limit = c->redo_cap_limit; // per size class (2^26: 512 MB of pairs at most)
            for (int k = 0; k < kNumClasses; ++k)
            {
                uint64_t const pairs = (uint64_t)nq * (c->class_first[k + 1] - c->class_first[k]);
                unsigned const cap = (unsigned)std::min<uint64_t>(pairs, cap_limit);
                qa.redo_base[k] = (unsigned)tot;
                qa.redo_cap[k] = cap;
                tot += cap;
                uint64_t const tpb = dcp_rowsweep_tasks_per_block(kClasses[k].W);
                uint64_t g = std::min<uint64_t>((cap + tpb - 1) / tpb, 8ull * c->num_cus);
                redo_grid[k] = (unsigned)((g + 7) / 8 * 8);
            }
            if (c->d_redo.n < tot) HIP_TRY(c, c->d_redo.alloc((size_t)tot));
        }
        // counters, overflow flag and the ring's error word start every query-lane scan at zero (without redo the
        // lists' capacities are 0: the counters are never written)
        if (!c->d_redo_n.p) HIP_TRY(c, c->d_redo_n.alloc(DCP_MAX_CLASSES + 2));
        HIP_TRY(c, hipMemsetAsync(c->d_redo_n.p, 0, (DCP_MAX_CLASSES + 2) * sizeof(unsigned), c->stream));
        qa.redo = c->d_redo.p;
        qa.redo_n = c->d_redo_n.p;
        qa.redo_overflow = c->d_redo_n.p + DCP_MAX_CLASSES;
        qa.ring_error = c->d_redo_n.p + DCP_MAX_CLASSES + 1;
        // scratch = 3 planes x plane rows x NT floats per resident block; long sequences
        // (SCHED_SEQ_SIZE allows 1 MiB) get fewer resident blocks so the planes stay within budget
        uint64_t const per_block = (uint64_t)dcp_qlane_scratch_planes() * (uint64_t)qa.plane_rows * NT; // floats
        uint64_t const budget = (uint64_t)64 << 28;                      // 64 GiB of floats / 4
        uint64_t fit = per_block ? budget / per_block : 0;
        // one 512-thread block per CU (two-stage) or two 256-thread blocks (single-stage)
        uint64_t const resident_blocks = (uint64_t)ql_blocks_per_cu * c->num_cus; // w3: wavefront slots, 3 per block
        unsigned nblocks = (unsigned)std::min<uint64_t>(std::min<uint64_t>(ntasks, resident_blocks), fit);
        if (use_w3) nblocks = (nblocks + 2u) / 3u * 3u; // whole 3-slot blocks (scratch is sized for every slot)
        if (nblocks == 0)
            return c->fail(DCP_ENOMEM, "sequences too long for the query-lane kernel (%u plane rows): use kernel = 1", qa.plane_rows);
        size_t const need = (size_t)nblocks * per_block;
        if (c->d_scratch.n < need) HIP_TRY(c, c->d_scratch.alloc(need));
        qa.scratch = c->d_scratch.p;
        HIP_TRY(c, hipMemsetAsync(c->d_task_counter.p, 0, sizeof(unsigned), c->stream));
        HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
        if (two_stage ? dcp_launch_qlane2(&qa, nblocks, c->stream)
                      : use_w3 ? dcp_launch_qlane_w3(&qa, nblocks / 3u, c->stream)
                               : dcp_launch_qlane(&qa, nblocks, NT, c->stream))
            return c->fail(DCP_EFAIL, "query-lane launch failed");
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->ev_class[c->n_launched], c->stream));
        c->launched_class[c->n_launched++] = -1;
        c->last_launches = 1;
        for (int k = 0; redo && k < kNumClasses; ++k)
        {
            if (redo_grid[k] == 0) continue;
            a.pairs = c->d_redo.p + qa.redo_base[k];
            a.npairs = c->d_redo_n.p + k;
            a.pair_cap = qa.redo_cap[k];
            a.first_prof = 0;
            a.nprof = c->nprof;
            SizeClass const sc = kClasses[k];
            if (dcp_launch_rowsweep(sc.R, sc.W, &a, redo_grid[k], c->stream))
                return c->fail(DCP_EFAIL, "no kernel for class R=%d W=%d", sc.R, sc.W);
            HIP_TRY(c, hipEventRecord(c->ev_class[c->n_launched], c->stream));
            c->launched_class[c->n_launched++] = k;
            c->last_launches++;
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
        c->scanned = true;
        c->redo_pending = redo;
        c->ring_check_pending = two_stage;
        c->last_redo_pairs = 0;
        c->last_prm = *prm;
        return DCP_OK;
    }
    if (int rc = ensure_rowsweep_layout(c)) return rc;
    a.emis_match = c->d_emis_match.p;
    c->redo_pending = false;
    c->last_redo_pairs = 0;
    HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    // fork / join around the class launches when no single one fills the chip for long (2^22 pairs: 20 000 profiles x 209
    // queries, i.e. every batch the automatic choice gives the row sweep; 112 .. 160 queries: 1 % faster than one launch
    // after the other, profiles/r04/class_launch_order.txt)
    bool const overlap = (uint64_t)nq * c->nprof < ((uint64_t)1 << 22);
    c->last_overlapped = overlap;
    // Multi-wavefront classes with a segmented-sweep kernel (dcp_kernels.hip): one wavefront per pair, one SEGMENT of
    // the profile per launch, B(j) = N(j) + NB, the pairs with feedback finished by the exact kernel behind it.
    // Per class: two boundary columns of lmax + 2 rows of 16 bytes per pair, and a redo list of all its pairs.  The
    // columns of a class are capped at kSegColBytes: beyond, its queries are swept chunk by chunk.
    unsigned lmax_scan = 0;
    for (unsigned q = q_begin; q < q_end; ++q)
        lmax_scan = std::max(lmax_scan, c->seq_len[q]);
    unsigned const seg_stride = lmax_scan + 2u; // rows of 16 bytes per column
    uint64_t const kSegColBytes = c->seg_col_bytes;
    unsigned seg_blocks[kNumClasses] = {0}, seg_chunk[kNumClasses] = {0};
    uint64_t seg_redo_off[kNumClasses] = {0}, seg_col_rows = 0, seg_redo_tot = 0;
    for (int k = 0; k < kNumClasses; ++k)
    {
        unsigned const np = c->class_first[k + 1] - c->class_first[k];
        SizeClass const sc = kClasses[k];
        uint64_t const pairs = (uint64_t)np * nq;
        // (once the class has pairs enough to fill the chip's wavefront slots twice over: below, a pair's segments one
        // after the other are a longer serial chain than the exact kernel's wavefronts side by side -- C3 DB, 1 query
        // 11.8 -> 14.4 ms with round 3's kernel; C5 DB, where 37 % of the profiles are in these classes: 16 queries
        // 291 ms with the exact kernel, profiles/r04/latency_probe_c5_db.txt)
        bool const want = c->seg_mode == 1 || (c->seg_mode < 0 && pairs >= 16ull * c->num_cus);
        bool const have = sc.W > 1 && want && np > 0 && pairs <= ((uint64_t)1 << 26);
        if (!have) continue;
        // queries per chunk: np x chunk pairs x 2 columns x seg_stride x 16 bytes within the cap
        uint64_t const per_query = (uint64_t)np * 2u * seg_stride * 16u;
        uint64_t const chunk = std::min<uint64_t>(nq, kSegColBytes / per_query);
        if (chunk == 0) continue; // very long sequences: the exact kernel
        if ((uint64_t)np * ((chunk + 3u) / 4u) > 0x7ffffff0ull) continue;
        seg_blocks[k] = dcp_segsweep_blocks(np, (unsigned)chunk);
        seg_chunk[k] = (unsigned)chunk;
        // (classes run one after the other on a stream, or side by side on their own streams when the scan is small:
        // every class gets columns of its own)
        seg_col_rows += (uint64_t)np * chunk * 2u * seg_stride;
        seg_redo_off[k] = seg_redo_tot, seg_redo_tot += pairs;
    }
    if (seg_col_rows)
    {
        if (c->d_seg_scratch.n < seg_col_rows * 4u) HIP_TRY(c, c->d_seg_scratch.alloc((size_t)seg_col_rows * 4u));
        if (c->d_seg_redo.n < seg_redo_tot) HIP_TRY(c, c->d_seg_redo.alloc((size_t)seg_redo_tot));
        if (!c->d_seg_redo_n.p) HIP_TRY(c, c->d_seg_redo_n.alloc(DCP_MAX_CLASSES));
        HIP_TRY(c, hipMemsetAsync(c->d_seg_redo_n.p, 0, DCP_MAX_CLASSES * sizeof(unsigned), c->stream));
        HIP_TRY(c, hipEventRecord(c->ev_start, c->stream)); // the forked streams wait for the counters' reset too
    }
    uint64_t seg_col_at = 0; // floats
    for (int kk = 0; kk < kNumClasses; ++kk)
    {
        // Forked launches (small batches): the classes of the largest profiles first.  A pair takes rows x the row's
        // latency, which grows with the wavefronts the pair spans, and the few pairs of the largest classes, started
        // last, ran on alone at the end: C3 DB, 1 / 2 / 4 / 16 queries 10.3 / 16.8 / 26.5 / 74.2 -> 9.4 / 15.4 / 25.2 /
        // 70.8 ms; C5 DB 4 .. 32 queries 1-5 % less, 1 and 64 queries 1-2 % more (profiles/r04/class_launch_order.txt;
        // the chip-filling classes first and only then the sparse ones, or the sparse ones first: in between)
        int const k = overlap ? kNumClasses - 1 - kk : kk;
        unsigned first = c->class_first[k], last = c->class_first[k + 1];
        if (last <= first) continue;
        hipStream_t const ls = overlap ? c->class_stream[k] : c->stream;
        if (overlap) HIP_TRY(c, hipStreamWaitEvent(ls, c->ev_start, 0));
        a.first_prof = first;
        a.nprof = last - first;
        a.pairs = nullptr, a.npairs = nullptr, a.pair_cap = 0;
        SizeClass const sc = kClasses[k];
        uint64_t const ntasks = (uint64_t)a.nprof * a.nchunks;
        if (ntasks > 0xffffffffull) return c->fail(DCP_EINVAL, "scan too large for one launch");
        if (seg_blocks[k])
        {
            unsigned const np = a.nprof, chunk = seg_chunk[k];
            float *const col_base = c->d_seg_scratch.p + seg_col_at;
            seg_col_at += (uint64_t)np * chunk * 2u * seg_stride * 4u;
            a.seg_stride = seg_stride;
            a.seg_redo = c->d_seg_redo.p + seg_redo_off[k];
            a.seg_redo_n = c->d_seg_redo_n.p + k;
            a.seg_redo_cap = (unsigned)ntasks;
            // Segment-major: launch s sweeps segment s of every pair of the chunk; the kernel boundary is the hand-off.
            // The class's profiles are ordered by their segments' lane width: one run of launches per width.
            for (unsigned sub = first; sub < last;)
            {
                unsigned const segR = seg_r_of(c->metas[sub].core_size);
                unsigned sub_end = sub, max_m = 0;
                while (sub_end < last && seg_r_of(c->metas[sub_end].core_size) == segR)
                    max_m = std::max(max_m, c->metas[sub_end].core_size), ++sub_end;
                unsigned const nsub = sub_end - sub, nseg_max = (max_m + 64u * segR - 1u) / (64u * segR);
                a.first_prof = sub, a.nprof = nsub;
                a.seg_col0 = col_base + (size_t)(sub - first) * chunk * 2u * seg_stride * 4u;
                a.seg_col1 = a.seg_col0 + (size_t)nsub * chunk * seg_stride * 4u;
                for (unsigned qc = 0; qc < nq; qc += chunk)
                {
                    a.seg_q0 = qc, a.seg_nq = std::min(chunk, nq - qc);
                    for (unsigned sg = 0; sg < nseg_max; ++sg)
                    {
                        a.seg_index = sg;
                        if (dcp_launch_segsweep((int)segR, &a, dcp_segsweep_blocks(nsub, a.seg_nq), ls))
                            return c->fail(DCP_EFAIL, "no segmented kernel for %u nodes per lane", segR);
                    }
                }
                sub = sub_end;
            }
            a.first_prof = first, a.nprof = np;
            HIP_TRY(c, hipEventRecord(c->ev_class[c->n_launched], ls));
            c->launched_redo[c->n_launched] = false;
            c->launched_class[c->n_launched++] = k;
            c->last_launches++;
            // the pairs it could not finish: the exact kernel in pair mode, right behind
            a.pairs = a.seg_redo, a.npairs = a.seg_redo_n, a.pair_cap = a.seg_redo_cap;
            a.first_prof = 0, a.nprof = c->nprof;
            uint64_t g = std::min<uint64_t>(ntasks, 8ull * c->num_cus);
            g = (g + 7u) / 8u * 8u;
            if (dcp_launch_rowsweep(sc.R, sc.W, &a, (unsigned)g, ls))
                return c->fail(DCP_EFAIL, "no kernel for class R=%d W=%d", sc.R, sc.W);
            HIP_TRY(c, hipEventRecord(c->ev_class[c->n_launched], ls));
            if (overlap) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_class[c->n_launched], 0));
            c->launched_redo[c->n_launched] = true;
            c->launched_class[c->n_launched++] = k;
            c->last_launches++;
            continue;
        }
        // K profiles per wavefront: the class of at most 64 nodes always (855-890 Gcell/s against 600 with one node
        // per lane); the class of 65 .. 128 nodes while the batch is small -- its one-profile kernels at six or seven
        // wavefronts per SIMD are faster on a full grid (936 against 860 Gcell/s at 1 000 queries), slower below
        // (profiles/r04/rowsweep_small_classes.txt)
        bool const use_mp = k < 2 && c->mp_first[k + 1] > c->mp_first[k] &&
                            (c->mp_mode == 1 || (c->mp_mode < 0 && (k == 0 || nq < kMpClass1MaxQueries)));
        if (use_mp)
        {
            // The two smallest classes: K profiles per wavefront (viterbi_mp_kernel) for the grouped ones, the
            // unstaged one-profile kernel on the column views of the flagged rest.
            unsigned const g0 = c->mp_first[k], g1 = c->mp_first[k + 1];
            if (g1 > g0)
            {
                a.mp_groups = c->d_mp_groups.p, a.mp_in = c->d_mp_in.p;
                a.first_prof = g0, a.nprof = g1 - g0;
                if (int lrc = dcp_launch_mp((int)kMpParts[k], &a, ls))
                    return lrc == -2 ? c->fail(DCP_EINVAL, "scan too large for one launch") : c->fail(DCP_EFAIL, "no %u-profile kernel", kMpParts[k]);
            }
            if (c->mp_flagged_first[k])
            {
                a.first_prof = c->mp_flagged_first[k] - 1u, a.nprof = last - a.first_prof;
                if (int lrc = dcp_launch_rowsweep_grid(sc.R, sc.W, &a, 0, 1u, ls, 0u, 0))
                    return lrc == -2 ? c->fail(DCP_EINVAL, "scan too large for one launch") : c->fail(DCP_EFAIL, "no kernel for class R=%d W=%d", sc.R, sc.W);
            }
            a.first_prof = first, a.nprof = last - first;
            HIP_TRY(c, hipEventRecord(c->ev_class[c->n_launched], ls));
            if (overlap) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_class[c->n_launched], 0));
            c->launched_redo[c->n_launched] = false;
            c->launched_class[c->n_launched++] = k;
            c->last_launches++;
            continue;
        }
        int stg, pf;
        unsigned bw;
        rowsweep_variant(c, sc.R, sc.W, a.nchunks, &stg, &bw, &pf);
        if (int lrc = dcp_launch_rowsweep_grid(sc.R, sc.W, &a, stg, bw, ls, c->rs_pad_lds, pf))
            return lrc == -2 ? c->fail(DCP_EINVAL, "scan too large for one launch")
                             : c->fail(DCP_EFAIL, "no kernel for class R=%d W=%d (stage %d, %u wavefronts)", sc.R, sc.W, stg, bw);
        HIP_TRY(c, hipEventRecord(c->ev_class[c->n_launched], ls));
        if (overlap) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_class[c->n_launched], 0));
        c->launched_redo[c->n_launched] = false;
        c->launched_class[c->n_launched++] = k;
        c->last_launches++;
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->scanned = true;
    return DCP_OK;
}

// (declared above dcp_gpu_scan_range)
// Wait for the stream; after a query-lane scan also look at the redo counters.  A redo list
// that overflowed (> 2^26 pairs of one size class needed the row sweep) lost pairs: the scan is
// repeated with the row-sweep kernel, which needs no list.
static int finish_scan(dcp_gpu_ctx *c)
{
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!c->redo_pending && !c->ring_check_pending) return DCP_OK;
    bool const redo = c->redo_pending;
    c->redo_pending = false;
    c->ring_check_pending = false;
    unsigned n[DCP_MAX_CLASSES + 2];
    HIP_TRY(c, hipMemcpy(n, c->d_redo_n.p, sizeof n, hipMemcpyDeviceToHost));
    // the two-stage kernel's ring hand-shake ran into its poll bound (dcp_qlane.hip, ring_wait): the stages drained,
    // but rows were computed from values their partner never wrote -- nothing of this scan can be used
    if (n[DCP_MAX_CLASSES + 1])
    {
        c->scanned = false;
        return c->fail(DCP_EFAIL, "query-lane kernel: the LDS ring hand-shake between the two stages timed out; the scan's results are invalid");
    }
    if (!redo) return DCP_OK;
    uint64_t tot = 0;
    for (int k = 0; k < kNumClasses; ++k)
        tot += n[k];
    c->last_redo_pairs = (unsigned)std::min<uint64_t>(tot, 0xffffffffull);
    if (n[DCP_MAX_CLASSES] == 0) return DCP_OK;
    struct dcp_scan_params prm = c->last_prm;
    prm.kernel = 1;
    if (int rc = dcp_gpu_scan_range(c, &prm, c->last_q0, c->last_q1)) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DCP_OK;
}

#ifdef DCP_TEST_HOOKS
int dcp_gpu_test_set_rowsweep_variant(dcp_gpu_ctx *c, int stg, unsigned bw)
{
    if (!c) return DCP_EINVAL;
    c->rs_force_stg = stg;
    c->rs_force_bw = bw & 0xffu;
    c->rs_pad_lds = ((bw >> 8) & 0xffu) * 1024u; // bits 8..15: KiB of unused LDS per block (fewer blocks per CU)
    c->rs_force_pf = (int)((bw >> 16) & 1u);      // bit 16: the two-rows-ahead prefetch variant
    c->rs_force_R = (int)((bw >> 20) & 15u);      // bits 20..23: only the class with this many nodes per lane (0: all)
    c->seg_mode = ((bw >> 24) & 3u) == 1u ? 0 : ((bw >> 24) & 3u) == 2u ? 1 : -1; // bits 24..25: 1 = never the segmented sweep, 2 = always
    c->mp_mode = ((bw >> 26) & 3u) == 1u ? 0 : ((bw >> 26) & 3u) == 2u ? 1 : -1;  // bits 26..27: 1 = never K profiles per wavefront, 2 = always
    return DCP_OK;
}
int dcp_gpu_test_set_seg_col_bytes(dcp_gpu_ctx *c, unsigned long long bytes)
{
    if (!c) return DCP_EINVAL;
    c->seg_col_bytes = bytes ? bytes : (uint64_t)6 << 30;
    return DCP_OK;
}
int dcp_gpu_test_set_ring_stall(dcp_gpu_ctx *c, int on)
{
    if (!c) return DCP_EINVAL;
    c->ring_stall = on ? 1u : 0u;
    return DCP_OK;
}
int dcp_gpu_test_set_trace_mode(dcp_gpu_ctx *c, int own_forward, unsigned long long budget_floats)
{
    if (!c) return DCP_EINVAL;
    c->trace_mode = own_forward ? 1 : 0;
    c->trace_budget = budget_floats;
    return DCP_OK;
}
int dcp_gpu_test_set_redo_cap(dcp_gpu_ctx *c, unsigned cap)
{
    if (!c) return DCP_EINVAL;
    c->redo_cap_limit = cap ? cap : 1u << 26;
    return DCP_OK;
}
#endif

int dcp_gpu_sync(dcp_gpu_ctx *c)
{
    if (!c) return DCP_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    return finish_scan(c);
}

int dcp_gpu_last_scan_redo_pairs(dcp_gpu_ctx *c, unsigned *npairs)
{
    if (!c || !npairs) return DCP_EINVAL;
    if (!c->scanned) return c->fail(DCP_EINVAL, "no scan yet");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = finish_scan(c)) return rc;
    *npairs = c->last_redo_pairs;
    return DCP_OK;
}

float dcp_gpu_last_scan_ms(dcp_gpu_ctx *c)
{
    if (!c || !c->scanned) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(c->ev_stop) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) != hipSuccess) return -1.0f;
    return ms;
}

unsigned dcp_gpu_last_scan_launches(dcp_gpu_ctx const *c) { return c ? c->last_launches : 0; }
int dcp_gpu_last_scan_kernel(dcp_gpu_ctx const *c) { return c && c->scanned ? c->last_kernel_variant : 0; }

int dcp_gpu_last_scan_launch_info(dcp_gpu_ctx *c, unsigned i, struct dcp_launch_info *out)
{
    if (!c || !out || !c->scanned) return DCP_EINVAL;
    if (i >= c->n_launched) return DCP_EINVAL;
    int const k = c->launched_class[i];
    // overlapped row-sweep launches all start at ev_start
    hipEvent_t const before = i == 0 || (c->last_kernel == 1 && c->last_overlapped) ? c->ev_start : c->ev_class[i - 1];
    if (hipEventSynchronize(c->ev_class[i]) != hipSuccess) return DCP_EFAIL;
    float ms = 0;
    if (hipEventElapsedTime(&ms, before, c->ev_class[i]) != hipSuccess) return DCP_EFAIL;
    if (k < 0) // the query-lane launch: every pair of the scan
    {
        out->nodes_per_lane = 4 * c->ql_G; // KT nodes per tile, one query per lane
        out->waves_per_pair = 0;
        out->nprofiles = c->nprof;
        out->ms = ms;
        out->cells = dcp_gpu_scan_cells(c);
        out->algorithmic_bytes = dcp_gpu_scan_algorithmic_bytes(c);
        return DCP_OK;
    }
    if (c->last_kernel == 2 || c->launched_redo[i]) // a redo launch: its pairs are counted in the launch before it
    {
        out->nodes_per_lane = kClasses[k].R;
        out->waves_per_pair = kClasses[k].W;
        out->nprofiles = c->class_first[k + 1] - c->class_first[k];
        out->ms = ms;
        out->cells = 0;
        out->algorithmic_bytes = 0;
        return DCP_OK;
    }
    uint64_t sumM = 0, np = 0, len = 0;
    for (unsigned j = c->class_first[k]; j < c->class_first[k + 1]; ++j, ++np)
        sumM += c->metas[j].core_size;
    for (unsigned q = c->last_q0; q < c->last_q1; ++q)
        len += c->seq_len[q];
    uint64_t const nq = c->last_q1 - c->last_q0;
    out->nodes_per_lane = kClasses[k].R;
    out->waves_per_pair = kClasses[k].W;
    out->nprofiles = (unsigned)np;
    out->ms = ms;
    out->cells = sumM * len;
    out->algorithmic_bytes = 20ull * sumM * len + 32ull * (sumM + np) * nq + len * np + 8ull * np * nq;
    return DCP_OK;
}

int dcp_gpu_fetch_scores(dcp_gpu_ctx *c, float *null_out, float *alt_out)
{
    if (!c) return DCP_EINVAL;
    if (!c->scanned || !c->have_scores) return c->fail(DCP_EINVAL, "no dense scores kept by the last scan");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = finish_scan(c)) return rc;
    size_t const bytes = (size_t)c->nseqs * c->nprof * sizeof(float);
    if (null_out) HIP_TRY(c, hipMemcpy(null_out, c->d_null.p, bytes, hipMemcpyDeviceToHost));
    if (alt_out) HIP_TRY(c, hipMemcpy(alt_out, c->d_alt.p, bytes, hipMemcpyDeviceToHost));
    return DCP_OK;
}

int dcp_gpu_fetch_hits(dcp_gpu_ctx *c, struct dcp_hit *hits, unsigned cap, unsigned *nhits)
{
    if (!c || !nhits) return DCP_EINVAL;
    if (!c->scanned) return c->fail(DCP_EINVAL, "no scan to fetch hits from");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = finish_scan(c)) return rc;
    unsigned n = 0;
    dcp_hit const *const hits_p = c->ext_hits ? c->ext_hits : c->d_hits.p;
    unsigned const hits_cap = c->ext_hits ? c->ext_cap : c->hit_cap;
    HIP_TRY(c, hipMemcpy(&n, c->ext_hits ? c->ext_nhits : c->d_nhits.p, sizeof n, hipMemcpyDeviceToHost));
    *nhits = n;
    if (n > hits_cap) return c->fail(DCP_ENOMEM, "device hit buffer overflow: %u > %u", n, hits_cap);
    if (n > cap || (n && !hits)) return DCP_ENOMEM;
    if (n == 0) return DCP_OK;
    HIP_TRY(c, hipMemcpy(hits, hits_p, (size_t)n * sizeof(dcp_hit), hipMemcpyDeviceToHost));
    std::sort(hits, hits + n, [](dcp_hit const &x, dcp_hit const &y) {
        return x.seq_idx != y.seq_idx ? x.seq_idx < y.seq_idx : x.profile_idx < y.profile_idx;
    });
    return DCP_OK;
}


// ---------------------------------------------------------------------------
// Hits -> alt paths (device traceback)
// ---------------------------------------------------------------------------
int dcp_gpu_trace_paths(dcp_gpu_ctx *c, struct dcp_hit const *hits, unsigned nhits,
                        int multi_hits, int hmmer3_compat, int null_model,
                        struct dcp_step *steps_out, unsigned cap_steps, uint32_t *step_off,
                        float *alt_out)
{
    if (!c || !step_off || (nhits && !hits)) return DCP_EINVAL;
    if (c->nprof == 0 || c->nseqs == 0) return c->fail(DCP_EINVAL, "no DB / sequences resident");
    HIP_TRY(c, hipSetDevice(c->device));
    step_off[0] = 0;
    if (nhits == 0) return DCP_OK;
    for (unsigned h = 0; h < nhits; ++h)
        if (hits[h].seq_idx >= c->nseqs || hits[h].profile_idx >= c->nprof)
            return c->fail(DCP_EINVAL, "hit %u is outside the resident batch / DB", h);
    if (int rc = ensure_xtrans(c, multi_hits, hmmer3_compat)) return rc;
    if (int rc = ensure_rowsweep_layout(c)) return rc;

    // Forward pass: the row-sweep kernel of the profile's size class in pair mode, every row's values written to the
    // hit's work area (viterbi_rowsweep_kernel<R, W, 0, false, TRACE>: the scan's own rows, ~0.1 us each); then one
    // wavefront per hit walks back (viterbi_trace_kernel).  Until round 4 the forward pass was that kernel's own
    // one-wavefront loop over rows held in global memory (5-7 us per row: a job of 3 000 sequences of up to 10 kbp spent
    // 5 of its 23 s there, profiles/r04/host_scan_probe.txt); it is kept as the tests' second implementation
    // (trace_mode) and for the null model's one-state path.
    bool const sweep_forward = !null_model && c->trace_mode != 1;
    // per-hit work area and step capacity
    std::vector<uint64_t> need(nhits);
    std::vector<uint32_t> cap(nhits), wld(nhits);
    std::vector<int> cls(nhits);
    for (unsigned h = 0; h < nhits; ++h)
    {
        unsigned const slot = c->slot_of_pidx[hits[h].profile_idx];
        dcp_prof_meta const &m = c->metas[slot];
        uint64_t const L = c->seq_len[hits[h].seq_idx];
        int k = 0;
        while (k + 1 < kNumClasses && slot >= c->class_first[k + 1])
            ++k;
        cls[h] = k;
        wld[h] = sweep_forward ? 64u * (unsigned)kClasses[k].R * (unsigned)kClasses[k].W : m.width;
        need[h] = 3ull * (L + 1) * wld[h] + 5ull * (L + 1);
        cap[h] = (uint32_t)(2 * L + 2ull * m.core_size + 16);
    }
    uint64_t const budget = c->trace_budget ? c->trace_budget : 1ull << 31; // floats (8 GiB) of work area per round of launches
    int rc = DCP_OK;
    uint64_t total_steps = 0;
    std::vector<dcp_step> host_steps;
    for (unsigned h0 = 0; h0 < nhits;)
    {
        unsigned h1 = h0;
        uint64_t work = 0, scap = 0;
        while (h1 < nhits && (h1 == h0 || work + need[h1] <= budget))
            work += need[h1], scap += cap[h1], ++h1;
        unsigned const n = h1 - h0;
        // this round's hits in size-class order (the forward launches take contiguous pair lists); results go back
        // to the caller's order on the host
        std::vector<unsigned> ord(n);
        for (unsigned i = 0; i < n; ++i)
            ord[i] = h0 + i;
        if (sweep_forward) std::stable_sort(ord.begin(), ord.end(), [&](unsigned x, unsigned y) { return cls[x] < cls[y]; });
        std::vector<uint64_t> woff(n);
        std::vector<uint32_t> soff(n + 1, 0), ld(n);
        std::vector<dcp_hit> shits(n);
        std::vector<dcp_pair> pairs(n);
        unsigned cfirst[kNumClasses + 1] = {0};
        uint64_t acc = 0;
        for (unsigned i = 0; i < n; ++i)
        {
            unsigned const h = ord[i];
            woff[i] = acc;
            acc += need[h];
            soff[i + 1] = soff[i] + cap[h];
            ld[i] = wld[h];
            shits[i] = hits[h];
            pairs[i] = dcp_pair{hits[h].seq_idx, c->slot_of_pidx[hits[h].profile_idx]}; // {q, slot}
            cfirst[cls[h] + 1] = i + 1u;
        }
        for (int k = 1; k <= kNumClasses; ++k)
            if (cfirst[k] < cfirst[k - 1]) cfirst[k] = cfirst[k - 1];
        if (c->d_trace_work.n < work) HIP_TRY(c, c->d_trace_work.alloc(work));
        DevBuf<float> d_alt;
        DevBuf<uint64_t> d_woff;
        DevBuf<uint32_t> d_soff, d_nsteps, d_ld, d_counts;
        DevBuf<dcp_step> d_steps;
        DevBuf<dcp_hit> d_hits;
        DevBuf<dcp_pair> d_pairs;
        HIP_TRY(c, d_alt.alloc(n));
        HIP_TRY(c, d_woff.alloc(n));
        HIP_TRY(c, d_soff.alloc(n + 1));
        HIP_TRY(c, d_nsteps.alloc(n));
        HIP_TRY(c, d_ld.alloc(n));
        HIP_TRY(c, d_steps.alloc(scap));
        HIP_TRY(c, d_hits.alloc(n));
        HIP_TRY(c, hipMemcpy(d_woff.p, woff.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(d_soff.p, soff.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(d_ld.p, ld.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(d_hits.p, shits.data(), n * sizeof(dcp_hit), hipMemcpyHostToDevice));
        if (sweep_forward)
        {
            unsigned counts[kNumClasses];
            for (int k = 0; k < kNumClasses; ++k)
                counts[k] = cfirst[k + 1] - cfirst[k];
            HIP_TRY(c, d_pairs.alloc(n));
            HIP_TRY(c, d_counts.alloc(kNumClasses));
            HIP_TRY(c, hipMemcpy(d_pairs.p, pairs.data(), n * sizeof(dcp_pair), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(d_counts.p, counts, sizeof counts, hipMemcpyHostToDevice));
            dcp_scan_args fa{};
            fa.profs = c->d_metas.p;
            fa.emis_match = c->d_emis_match.p;
            fa.emis_insert = c->d_emis_insert.p;
            fa.emis_null = c->d_emis_null.p;
            fa.trans8 = c->d_trans8.p;
            fa.seq_words = c->d_seq_words.p;
            fa.seq_woff = c->d_seq_woff.p;
            fa.seq_len = c->d_seq_len.p;
            fa.xtrans = c->d_xtrans.p;
            fa.nprof_total = c->nprof;
            fa.nprof = c->nprof;
            fa.nseqs = c->nseqs;
            fa.qchunk = 1u;
            fa.nchunks = c->nseqs;
            fa.trace_work = c->d_trace_work.p;
            // the classes' launches side by side on their own streams, largest profiles first: a class's launch lasts as
            // long as its longest hit and seldom fills the chip (one after the other they took 0.31 of a job's 0.40 s of
            // traceback, profiles/r04/host_scan_probe_mixed_kernel_stats.csv)
            for (int k = 0; k <= kNumClasses; ++k)
                if (!c->ev_trace[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_trace[k], hipEventDisableTiming));
            HIP_TRY(c, hipEventRecord(c->ev_trace[kNumClasses], c->stream));
            for (int k = kNumClasses - 1; k >= 0; --k)
            {
                if (counts[k] == 0u) continue;
                SizeClass const sc = kClasses[k];
                hipStream_t const ls = c->class_stream[k];
                HIP_TRY(c, hipStreamWaitEvent(ls, c->ev_trace[kNumClasses], 0));
                fa.pairs = d_pairs.p + cfirst[k];
                fa.npairs = d_counts.p + k;
                fa.pair_cap = counts[k];
                fa.trace_woff = d_woff.p + cfirst[k];
                fa.trace_alt = d_alt.p + cfirst[k];
                unsigned const tpb = dcp_rowsweep_tasks_per_block(sc.W);
                unsigned const nb = ((counts[k] + tpb - 1u) / tpb + 7u) / 8u * 8u;
                if (dcp_launch_trace_forward(sc.R, sc.W, &fa, nb, ls))
                    return c->fail(DCP_EFAIL, "no traceback kernel for class R=%d W=%d", sc.R, sc.W);
                HIP_TRY(c, hipEventRecord(c->ev_trace[k], ls));
                HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_trace[k], 0));
            }
            HIP_TRY(c, hipGetLastError());
        }
        dcp_trace_args ta{};
        ta.profs = c->d_metas.p;
        ta.slot_of_pidx = c->d_slot_of_pidx.p;
        ta.emis_match = c->d_emis_match.p;
        ta.emis_insert = c->d_emis_insert.p;
        ta.emis_null = c->d_emis_null.p;
        ta.trans8 = c->d_trans8.p;
        ta.seq_words = c->d_seq_words.p;
        ta.seq_woff = c->d_seq_woff.p;
        ta.seq_len = c->d_seq_len.p;
        ta.xtrans = c->d_xtrans.p;
        ta.hits = d_hits.p;
        ta.nhits = n;
        ta.work = c->d_trace_work.p;
        ta.work_off = d_woff.p;
        ta.steps = d_steps.p;
        ta.step_off = d_soff.p;
        ta.nsteps = d_nsteps.p;
        ta.alt_out = d_alt.p;
        ta.null_model = null_model ? 1 : 0;
        ta.skip_forward = sweep_forward ? 1 : 0;
        ta.work_ld = d_ld.p;
        dcp_launch_trace(&ta, n, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::vector<uint32_t> ns(n);
        std::vector<dcp_step> st(scap);
        std::vector<float> alts(n);
        HIP_TRY(c, hipMemcpy(ns.data(), d_nsteps.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(st.data(), d_steps.p, scap * sizeof(dcp_step), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(alts.data(), d_alt.p, n * sizeof(float), hipMemcpyDeviceToHost));
        // back to the caller's order
        std::vector<unsigned> at(n);
        for (unsigned i = 0; i < n; ++i)
            at[ord[i] - h0] = i;
        for (unsigned j = 0; j < n; ++j)
        {
            unsigned const i = at[j], h = h0 + j;
            if (alt_out) alt_out[h] = alts[i];
            if (ns[i] == 0xffffffffu)
            {
                rc = c->fail(DCP_EFAIL, "pair (seq %u, profile %u) has no finite alt path", hits[h].seq_idx, hits[h].profile_idx);
                ns[i] = 0;
            }
            else if (ns[i] > cap[h])
            {
                rc = c->fail(DCP_EFAIL, "path of hit %u exceeds its step capacity", h);
                ns[i] = 0;
            }
            host_steps.insert(host_steps.end(), st.begin() + soff[i], st.begin() + soff[i] + ns[i]);
            total_steps += ns[i];
            step_off[h + 1] = (uint32_t)total_steps;
        }
        h0 = h1;
    }
    if (rc) return rc;
    if (total_steps > cap_steps || (total_steps && !steps_out)) return DCP_ENOMEM;
    if (total_steps) std::memcpy(steps_out, host_steps.data(), total_steps * sizeof(dcp_step));
    return DCP_OK;
}

static void last_range(dcp_gpu_ctx const *c, uint64_t *sumM, uint64_t *len, uint64_t *nq)
{
    *sumM = *len = 0;
    for (unsigned m : c->core_sizes)
        *sumM += m;
    unsigned q0 = c->scanned ? c->last_q0 : 0, q1 = c->scanned ? c->last_q1 : c->nseqs;
    for (unsigned q = q0; q < q1; ++q)
        *len += c->seq_len[q];
    *nq = q1 - q0;
}

uint64_t dcp_gpu_scan_cells(dcp_gpu_ctx const *c)
{
    if (!c) return 0;
    uint64_t sumM, len, nq;
    last_range(c, &sumM, &len, &nq);
    return sumM * len;
}

uint64_t dcp_gpu_scan_algorithmic_bytes(dcp_gpu_ctx const *c)
{
    // per pair 20*M*L + 32*(M+1) + L + 8
    if (!c) return 0;
    uint64_t sumM, len, Q, P = c->core_sizes.size();
    last_range(c, &sumM, &len, &Q);
    return 20ull * sumM * len + 32ull * (sumM + P) * Q + len * P + 8ull * P * Q;
}

} // extern "C"
