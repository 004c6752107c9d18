// dcp_dist.cpp -- multi-GPU layer of the scan path in C-ABI form (include/dcp_gpu.h, "One process
// per GPU"): contiguous profile shards balanced by cells, and the ONLY collective of the path, the
// gather of hit records over RCCL (counts all-gather, then a grouped send/recv
// gather-v).  Pairs are independent, so the data path itself needs no exchange.
//
// librccl.so is loaded on first use (dlopen), not linked: a single-GPU user never pays for it, and a
// box without RCCL can still run everything else.  The pure bookkeeping (counts -> displacements ->
// concatenation -> global profile indices -> (seq, profile) order) is a separate host function,
// dcp_dist_merge_hits, so that it is covered on CPU by the gloo world_size-2 test.
#include "dcp_gpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace
{
struct Rccl
{
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

Rccl &rccl()
{
    static Rccl r;
    return r;
}

bool rccl_load()
{
    Rccl &r = rccl();
    if (r.handle) return true;
    for (char const *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
        if ((r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.handle)
    {
        r.err = std::string("cannot load librccl.so: ") + dlerror();
        return false;
    }
    bool ok = true;
    auto sym = [&](char const *name) {
        void *p = dlsym(r.handle, name);
        if (!p)
        {
            ok = false;
            r.err = std::string("librccl.so lacks ") + name;
        }
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok)
    {
        dlclose(r.handle);
        r.handle = nullptr;
    }
    return ok;
}
} // namespace

struct dcp_dist
{
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    hipStream_t stream = nullptr;
    // device staging: {records held, profile_offset, records found} of every rank; all ranks' records back to back
    uint32_t *d_meta_mine = nullptr, *d_meta_all = nullptr;
    dcp_hit *d_recv = nullptr;
    size_t recv_cap = 0;
    double last_gather_ms = 0.0; // wall time of the last gather on this rank (both exchanges + merge)
    std::string err;

    int fail(int rc, char const *what, char const *detail)
    {
        err = std::string(what) + ": " + detail;
        std::fprintf(stderr, "dcp_dist[%d/%d]: %s\n", rank, nranks, err.c_str());
        return rc;
    }
};

#define DIST_HIP(d, call)                                                                         \
    do                                                                                            \
    {                                                                                             \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return (d)->fail(DCP_EFAIL, #call, hipGetErrorString(e_));          \
    } while (0)
#define DIST_NCCL(d, call)                                                                        \
    do                                                                                            \
    {                                                                                             \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) return (d)->fail(DCP_EFAIL, #call, rccl().GetErrorString(r_));     \
    } while (0)

extern "C" {

int dcp_dist_unique_id(unsigned char id[DCP_DIST_ID_BYTES])
{
    static_assert(DCP_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) return DCP_EINVAL;
    if (!rccl_load())
    {
        std::fprintf(stderr, "dcp_dist: %s\n", rccl().err.c_str());
        return DCP_EFAIL;
    }
    ncclUniqueId u;
    ncclResult_t r = rccl().GetUniqueId(&u);
    if (r != ncclSuccess)
    {
        std::fprintf(stderr, "dcp_dist: ncclGetUniqueId: %s\n", rccl().GetErrorString(r));
        return DCP_EFAIL;
    }
    std::memcpy(id, u.internal, DCP_DIST_ID_BYTES);
    return DCP_OK;
}

dcp_dist *dcp_dist_init(unsigned char const id[DCP_DIST_ID_BYTES], int rank, int nranks, int device)
{
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return nullptr;
    if (!rccl_load())
    {
        std::fprintf(stderr, "dcp_dist: %s\n", rccl().err.c_str());
        return nullptr;
    }
    dcp_dist *d = new (std::nothrow) dcp_dist();
    if (!d) return nullptr;
    d->rank = rank, d->nranks = nranks, d->device = device;
    ncclUniqueId u;
    std::memcpy(u.internal, id, DCP_DIST_ID_BYTES);
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&d->d_meta_mine, DCP_DIST_META_WORDS * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc((void **)&d->d_meta_all, DCP_DIST_META_WORDS * sizeof(uint32_t) * (size_t)nranks) == hipSuccess;
    if (ok)
    {
        ncclResult_t r = rccl().CommInitRank(&d->comm, nranks, u, rank);
        if (r != ncclSuccess)
        {
            std::fprintf(stderr, "dcp_dist[%d/%d]: ncclCommInitRank: %s\n", rank, nranks, rccl().GetErrorString(r));
            ok = false;
        }
    }
    if (!ok)
    {
        dcp_dist_free(d);
        return nullptr;
    }
    return d;
}

// Rendezvous through a file for launchers without any other channel (a plain C program started once
// per GPU): rank 0 writes {magic, nranks, run nonce, id} to `path` (atomically, via rename), the others wait
// for it.  A left-over id would give ranks different communicators and ncclCommInitRank would block without
// a timeout, so a peer must be able to tell this run's file from a previous run's:
//   run_nonce != 0 (dcp_dist_init_from_file_run): the launcher hands every rank the same non-zero number
//     (its pid and start time, say); a peer takes ONLY a file carrying it -- a previous run's file at the
//     same path, however young, is ignored until rank 0 has replaced it;
//   run_nonce == 0 (dcp_dist_init_from_file): the path itself must be fresh for every run; rank 0 removes
//     whatever lies there first, and the other ranks refuse a file of another layout or rank count or one
//     written more than kStaleSeconds before they arrived.  A peer that arrives before rank 0 has removed
//     a young left-over file still takes it: use the nonce form where runs may follow each other at one path.
namespace
{
constexpr uint32_t kIdFileMagic = 0xDC9D1574u;
constexpr double kStaleSeconds = 120.0;
struct IdFile
{
    uint32_t magic, nranks;
    uint64_t nonce;
    unsigned char id[DCP_DIST_ID_BYTES];
};
} // namespace

dcp_dist *dcp_dist_init_from_file_run(char const *path, uint64_t run_nonce, int rank, int nranks, int device,
                                      double timeout_s)
{
    if (!path || nranks < 1 || rank < 0 || rank >= nranks) return nullptr;
    IdFile f;
    if (rank == 0)
    {
        (void)::unlink(path); // a previous run's id must not be picked up by a fast peer
        if (dcp_dist_unique_id(f.id)) return nullptr;
        f.magic = kIdFileMagic, f.nranks = (uint32_t)nranks, f.nonce = run_nonce;
        std::string tmp = std::string(path) + ".tmp";
        FILE *fp = std::fopen(tmp.c_str(), "wb");
        if (!fp) return nullptr;
        bool ok = std::fwrite(&f, 1, sizeof f, fp) == sizeof f;
        ok = std::fclose(fp) == 0 && ok;
        if (!ok || std::rename(tmp.c_str(), path) != 0) return nullptr;
    }
    else
    {
        auto const t0 = std::chrono::steady_clock::now();
        time_t const arrived = ::time(nullptr);
        for (;;)
        {
            FILE *fp = std::fopen(path, "rb");
            if (fp)
            {
                size_t n = std::fread(&f, 1, sizeof f, fp);
                struct stat sb;
                bool const fresh = ::fstat(fileno(fp), &sb) == 0 && difftime(arrived, sb.st_mtime) <= kStaleSeconds;
                std::fclose(fp);
                bool const this_run = run_nonce ? f.nonce == run_nonce : (f.nonce == 0 && fresh);
                if (n == sizeof f && f.magic == kIdFileMagic && f.nranks == (uint32_t)nranks && this_run) break;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
            {
                std::fprintf(stderr, "dcp_dist[%d/%d]: no id of this run for %d ranks in %s after %.0f s\n", rank, nranks,
                             nranks, path, timeout_s);
                return nullptr;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    return dcp_dist_init(f.id, rank, nranks, device);
}

dcp_dist *dcp_dist_init_from_file(char const *path, int rank, int nranks, int device, double timeout_s)
{
    return dcp_dist_init_from_file_run(path, 0, rank, nranks, device, timeout_s);
}

void dcp_dist_free(dcp_dist *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    if (d->comm) (void)rccl().CommDestroy(d->comm);
    if (d->d_meta_mine) (void)hipFree(d->d_meta_mine);
    if (d->d_meta_all) (void)hipFree(d->d_meta_all);
    if (d->d_recv) (void)hipFree(d->d_recv);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

int dcp_dist_rank(dcp_dist const *d) { return d ? d->rank : -1; }
int dcp_dist_nranks(dcp_dist const *d) { return d ? d->nranks : 0; }
// Ranks RCCL itself reports for the communicator (ncclCommCount) -- a launcher's diagnostic: it must equal
// dcp_dist_nranks, and says so in bench.py's JSON line.  -1 without a communicator.
int dcp_dist_comm_count(dcp_dist const *d)
{
    int n = -1;
    if (!d || !d->comm || !rccl().CommCount || rccl().CommCount(d->comm, &n) != ncclSuccess) return -1;
    return n;
}
double dcp_dist_last_gather_ms(dcp_dist const *d) { return d ? d->last_gather_ms : 0.0; }
char const *dcp_dist_last_error(dcp_dist const *d) { return d ? d->err.c_str() : "no communicator"; }

void dcp_dist_shard(unsigned const *core_sizes, unsigned nprofiles, int nranks, int rank, unsigned *begin,
                    unsigned *end)
{
    std::vector<unsigned> b((size_t)nranks + 1);
    dcp_partition_by_cells(core_sizes, nprofiles, (unsigned)nranks, b.data());
    *begin = b[(size_t)rank];
    *end = b[(size_t)rank + 1];
}

// counts[r] records of rank r lie back to back in `records` (rank order).  Shard-local profile indices
// become global (+ profile_offset[r]); the result is ordered by (seq_idx, profile_idx) -- the order
// dcp_gpu_fetch_hits gives for one device.  Returns the total, or -1 if cap is too small.
long dcp_dist_merge_hits(unsigned const *counts, unsigned const *profile_offset, int nranks,
                         struct dcp_hit const *records, struct dcp_hit *out, unsigned cap)
{
    if (!counts || !profile_offset || nranks < 1 || (!records && !out)) return -1;
    uint64_t total = 0;
    for (int r = 0; r < nranks; ++r)
        total += counts[r];
    if (total > cap) return -1;
    size_t at = 0;
    for (int r = 0; r < nranks; ++r)
        for (unsigned i = 0; i < counts[r]; ++i, ++at)
        {
            out[at] = records[at];
            out[at].profile_idx += profile_offset[r];
        }
    std::sort(out, out + total, [](dcp_hit const &x, dcp_hit const &y) {
        return x.seq_idx != y.seq_idx ? x.seq_idx < y.seq_idx : x.profile_idx < y.profile_idx;
    });
    return (long)total;
}

// The decisions every rank takes from the gathered meta words {held, profile_offset, found} x nranks:
// counts and offsets per rank, 64-bit displacements, whether ANY rank found more records than it holds
// (its buffer overflowed: the global list would be truncated, so every rank must report it), whether ANY
// rank's scan FAILED (found = DCP_DIST_FOUND_FAILED, held = 0: that shard's hits are missing from the list,
// which "no hits" -- {0, off, 0} -- must not be mistaken for), and the total.  Pure host code, covered on
// CPU.  DCP_EINVAL if the total does not fit the 32-bit record count of the interface or a rank claims to
// hold more than it found.
int dcp_dist_gather_plan(uint32_t const *meta, int nranks, unsigned *counts, unsigned *profile_offset,
                         uint64_t *displ, int *any_overflow, int *any_failed, uint64_t *total)
{
    if (!meta || nranks < 1 || !counts || !profile_offset || !displ || !any_overflow || !any_failed || !total)
        return DCP_EINVAL;
    *any_overflow = 0;
    *any_failed = 0;
    displ[0] = 0;
    for (int r = 0; r < nranks; ++r)
    {
        uint32_t const held = meta[(size_t)DCP_DIST_META_WORDS * r], found = meta[(size_t)DCP_DIST_META_WORDS * r + 2];
        if (held > found) return DCP_EINVAL;
        if (found == DCP_DIST_FOUND_FAILED)
        {
            if (held) return DCP_EINVAL; // a failed rank sends nothing
            *any_failed = 1;
        }
        else if (found > held)
            *any_overflow = 1;
        counts[r] = held;
        profile_offset[r] = meta[(size_t)DCP_DIST_META_WORDS * r + 1];
        displ[r + 1] = displ[r] + held;
    }
    *total = displ[nranks];
    return *total > 0xffffffffull ? DCP_EINVAL : DCP_OK;
}

// All ranks call this after their scan is COMPLETE (dcp_gpu_sync: a query-lane scan finishes its redo
// pairs there -- dcp_dist_gather_scan_hits does that itself).
// hits_dev / nhits_dev: the device hit buffer and counter the scan wrote (dcp_gpu_set_hit_buffer).
// root >= 0: only that rank receives (gather-v); root < 0: every rank receives (all-gather-v).
// On a receiving rank *out is a malloc'ed array of *nout records (caller frees), global profile
// indices, ordered by (seq_idx, profile_idx); elsewhere *out = NULL, *nout = the global total.
} // extern "C"

namespace
{
// my_scan_failed: this rank has no valid hit list (its scan failed).  It still takes part in both exchanges
// (leaving would hang its peers inside the collective), holds nothing, and says so in the meta words.
int gather_impl(dcp_dist *d, void const *hits_dev, void const *nhits_dev, unsigned cap, unsigned profile_offset,
                int root, void *scan_stream, bool my_scan_failed, struct dcp_hit **out, unsigned *nout)
{
    if (!d || !out || !nout || root >= d->nranks || (!my_scan_failed && (!hits_dev || !nhits_dev))) return DCP_EINVAL;
    *out = nullptr;
    *nout = 0;
    auto const t_begin = std::chrono::steady_clock::now();
    struct Stamp
    {
        dcp_dist *d;
        std::chrono::steady_clock::time_point t0;
        ~Stamp() { d->last_gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    } stamp{d, t_begin};
    DIST_HIP(d, hipSetDevice(d->device));
    int const R = d->nranks;
    // the scan must have finished writing its records and counter
    if (scan_stream) DIST_HIP(d, hipStreamSynchronize((hipStream_t)scan_stream));

    // 1. {records held, profile_offset, records found} of every rank: one all-gather of 3 words
    uint32_t mine[DCP_DIST_META_WORDS] = {0, profile_offset, DCP_DIST_FOUND_FAILED};
    if (!my_scan_failed)
    {
        DIST_HIP(d, hipMemcpy(&mine[2], nhits_dev, sizeof(uint32_t), hipMemcpyDeviceToHost));
        // the sentinel is not a count a scan can report: 2^32 - 1 hits cannot be told from a failure, and is one
        if (mine[2] == DCP_DIST_FOUND_FAILED) my_scan_failed = true;
    }
    // A rank whose buffer overflowed still takes part in both exchanges (leaving here would hang its
    // peers inside the collective): it contributes the records it holds, and the third word tells EVERY
    // rank that the global list is incomplete -- all of them return DCP_ENOMEM.
    mine[0] = my_scan_failed ? 0 : (mine[2] > cap ? cap : mine[2]);
    DIST_HIP(d, hipMemcpyAsync(d->d_meta_mine, mine, sizeof mine, hipMemcpyHostToDevice, d->stream));
    DIST_NCCL(d, rccl().AllGather(d->d_meta_mine, d->d_meta_all, DCP_DIST_META_WORDS, ncclUint32, d->comm, d->stream));
    std::vector<uint32_t> meta((size_t)DCP_DIST_META_WORDS * R);
    DIST_HIP(d, hipMemcpyAsync(meta.data(), d->d_meta_all, meta.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
    DIST_HIP(d, hipStreamSynchronize(d->stream));
    std::vector<unsigned> counts((size_t)R), offs((size_t)R);
    std::vector<uint64_t> displ((size_t)R + 1, 0);
    int any_overflow = 0, any_failed = 0;
    uint64_t total = 0;
    // every rank sees the same meta words, so every rank takes the same branch here: nobody is left
    // waiting in the record exchange below
    if (dcp_dist_gather_plan(meta.data(), R, counts.data(), offs.data(), displ.data(), &any_overflow, &any_failed, &total))
        return d->fail(DCP_EINVAL, "hit gather", "more than 2^32 - 1 records in all, or inconsistent counts");
    *nout = (unsigned)total;
    bool const receiver = root < 0 || root == d->rank;

    // 2. gather-v of the 16-byte records: grouped send / recv, peer to peer over xGMI
    if (receiver && d->recv_cap < total)
    {
        if (d->d_recv) (void)hipFree(d->d_recv);
        d->d_recv = nullptr;
        d->recv_cap = 0;
        size_t const want = std::max<size_t>((size_t)total, 4096);
        DIST_HIP(d, hipMalloc((void **)&d->d_recv, want * sizeof(dcp_hit)));
        d->recv_cap = want;
    }
    size_t const words = sizeof(dcp_hit) / sizeof(uint32_t);
    if (receiver && counts[(size_t)d->rank]) // my own records: a local copy, no self send
        DIST_HIP(d, hipMemcpyAsync(d->d_recv + displ[(size_t)d->rank], hits_dev,
                                   (size_t)counts[(size_t)d->rank] * sizeof(dcp_hit), hipMemcpyDeviceToDevice, d->stream));
    if (R > 1)
    {
        DIST_NCCL(d, rccl().GroupStart());
        ncclResult_t gr = ncclSuccess;
        for (int peer = 0; peer < R && gr == ncclSuccess; ++peer)
        {
            if (peer == d->rank) continue;
            bool const peer_receives = root < 0 || root == peer;
            if (peer_receives && counts[(size_t)d->rank])
                gr = rccl().Send(hits_dev, (size_t)counts[(size_t)d->rank] * words, ncclUint32, peer, d->comm, d->stream);
            if (gr == ncclSuccess && receiver && counts[(size_t)peer])
                gr = rccl().Recv(d->d_recv + displ[(size_t)peer], (size_t)counts[(size_t)peer] * words, ncclUint32, peer,
                                 d->comm, d->stream);
        }
        ncclResult_t const ge = rccl().GroupEnd();
        if (gr != ncclSuccess) return d->fail(DCP_EFAIL, "ncclSend/ncclRecv", rccl().GetErrorString(gr));
        if (ge != ncclSuccess) return d->fail(DCP_EFAIL, "ncclGroupEnd", rccl().GetErrorString(ge));
    }
    char const *const ovf = "a rank found more hits than its device buffer holds: the gathered list is incomplete";
    char const *const flr = "a rank's scan failed: its shard's hits are missing from the gathered list";
    // ONE verdict on every rank, after both exchanges: a failed scan anywhere outranks an overflow
    auto verdict = [&]() {
        if (any_failed) return d->fail(DCP_EFAIL, "scan failed on some rank", flr);
        return any_overflow ? d->fail(DCP_ENOMEM, "hit buffer overflow", ovf) : (int)DCP_OK;
    };
    if (!receiver)
    {
        DIST_HIP(d, hipStreamSynchronize(d->stream));
        return verdict();
    }
    // 3. to the host; global indices; (seq, profile) order
    std::vector<dcp_hit> raw((size_t)total);
    if (total)
        DIST_HIP(d, hipMemcpyAsync(raw.data(), d->d_recv, (size_t)total * sizeof(dcp_hit), hipMemcpyDeviceToHost, d->stream));
    DIST_HIP(d, hipStreamSynchronize(d->stream));
    dcp_hit *res = (dcp_hit *)std::malloc(std::max<size_t>((size_t)total, 1) * sizeof(dcp_hit));
    if (!res) return d->fail(DCP_ENOMEM, "malloc", "hit list");
    if (dcp_dist_merge_hits(counts.data(), offs.data(), R, raw.data(), res, (unsigned)total) < 0)
    {
        std::free(res);
        return d->fail(DCP_EFAIL, "merge", "inconsistent counts");
    }
    *out = res;
    return verdict();
}
} // namespace

extern "C" {

int dcp_dist_gather_hits(dcp_dist *d, void const *hits_dev, void const *nhits_dev, unsigned cap,
                         unsigned profile_offset, int root, void *scan_stream, struct dcp_hit **out,
                         unsigned *nout)
{
    return gather_impl(d, hits_dev, nhits_dev, cap, profile_offset, root, scan_stream, false, out, nout);
}

// The gather for a scan context: completes the scan first (dcp_gpu_sync -- after a query-lane scan that
// is where the redo lists are checked and an overflowed scan is repeated with the row sweep, so the hit
// list is final), then gathers the buffer the scan wrote.  This is the form hosts should call.
int dcp_dist_gather_scan_hits(dcp_dist *d, dcp_gpu_ctx *ctx, unsigned profile_offset, int root,
                              struct dcp_hit **out, unsigned *nout)
{
    if (!d || !ctx || !out || !nout) return DCP_EINVAL;
    *out = nullptr;
    *nout = 0;
    // A rank whose scan failed must still enter the collective, or its peers would wait for it forever: it holds
    // nothing and puts DCP_DIST_FOUND_FAILED into its meta words, so EVERY rank -- a root that receives the list
    // included -- returns an error instead of a list that silently lacks one shard.  This rank reports its own
    // scan's error code and message.
    int const src = dcp_gpu_sync(ctx);
    void *hits_dev = nullptr, *nhits_dev = nullptr;
    unsigned cap = 0;
    int const brc = src ? src : dcp_gpu_hit_buffer(ctx, &hits_dev, &nhits_dev, &cap);
    if (brc)
    {
        std::string const why = dcp_gpu_last_error(ctx);
        (void)gather_impl(d, nullptr, nullptr, 0, profile_offset, root, nullptr, true, out, nout);
        return d->fail(brc, "scan", why.c_str());
    }
    return gather_impl(d, hits_dev, nhits_dev, cap, profile_offset, root, nullptr, false, out, nout);
}

void dcp_dist_free_hits(struct dcp_hit *hits) { std::free(hits); }

} // extern "C"
