// dcp_qlane.hip -- throughput kernel of the scan path: one LANE per query.
//
// viterbi_qlane_kernel<G>: a block is 256 queries against ONE profile.  The
// profile is cut into tiles of KT = 4*G consecutive nodes; a tile's match
// emission tables (KT x 1364 floats) sit in LDS and every lane gathers the five
// rows its own sequence window selects.  A tile is swept over all rows j with
// the 5-row history of its KT nodes in registers; what the next tile needs from
// row j -- D of its first node, the M/I/D part of that node's predecessor
// maximum, the running E -- goes through per-block scratch planes in HBM
// (coalesced, 12 B read + 12 B written per row, tile and lane).  No cross-lane operation exists: the
// delete chain is sequential in k inside the lane.
//
// Multi-hit feedback (B(j) needs E(j) of the same row, which needs every tile):
// the sweep runs with B0(j) = N(j) + NB only; the last tile then knows E(j), J(j)
// and checks whether max(E(j)+EB, J(j)+JB) exceeds the B(j) that was used.  If it
// never does, B0 IS the solution of the recurrence (a forward recurrence in j, so
// its solution is unique) and the lane's scores are exact -- bit-identical to the
// row sweep of dcp_kernels.hip and to the oracle.  A lane where it does (a pair
// with a local match good enough to re-enter the core: 1 % of the C3 pairs)
// publishes nothing and appends its (query, profile) to the redo list of the
// profile's size class; the row-sweep kernel, which has the whole row in one
// wavefront group and therefore the exact B(j), scores those pairs right after
// this kernel on the same stream.  (Re-sweeping in place until nothing changes
// is also exact, but one such lane costs its whole wavefront another full pass:
// 13 % of the scan time on the C3 workload.)
// Reference recursion: imm_dp_viterbi as driven by src/server/scan_thread.c:99-123;
// model wiring src/model/protein_model.c:410-500.
#include "dcp_kernels.h"

#include <hip/hip_runtime.h>
#include <type_traits>

namespace
{
constexpr int NC = DCP_NCODES;

// Every tile needs B0(j) = N(j) + NB.  RECOMPUTE_B: every tile re-derives N(j) from the
// background table (12 VALU per row; the gathers are shared with the insert table) instead
// of reading a fourth scratch plane written by the first tile (4 more bytes per row and tile).
#ifndef DCP_QLANE_RECOMPUTE_B
#define DCP_QLANE_RECOMPUTE_B 1
#endif
constexpr bool kRecomputeB = DCP_QLANE_RECOMPUTE_B != 0;
#ifndef DCP_QLANE_WD
#define DCP_QLANE_WD 5 // rows of sequence-word prefetch (1..5)
#endif
constexpr unsigned kWD = DCP_QLANE_WD;
constexpr unsigned kPlanes = kRecomputeB ? 3u : 4u; // scratch planes per block: Xm, Xd, Em (, B0)
// Timing diagnostics exist only as a separate COMPILE-TIME build (-DDCP_QLANE_DIAG=1|2|3, results are
// wrong): bit 0 makes every lane gather table row 0 (no LDS bank conflicts), bit 1 collapses the
// scratch planes to one row (no HBM traffic).  The shipped library is built without it; nothing at
// run time (no argument, no environment variable) can switch it on.
#ifndef DCP_QLANE_DIAG
#define DCP_QLANE_DIAG 0
#endif
// (DCP_QLANE_DIAG bit 0 is applied in gather_off)

// Round-3 trims of the row's bookkeeping instructions.  The row is VALU-bound -- 817 SIMD cycles per
// wavefront-row against 773 for its arithmetic alone (profiles/r03/row_valu.txt) -- so what is left is
// instruction count:
//   WPLANE  the sequence window of every row comes ready-made (w << 4, 16 bits) from a per-block plane
//           [row][lane] built once per scan, instead of being shifted together from packed words in every
//           tile's sweep: -5 VALU per row (stage 1 of the two-stage kernel: -4, it ORs its image base in);
//           2 bytes per row and lane instead of a quarter byte (+0.9 TB of the 6.3 TB a C3 launch moves)
//   IN16    the insert / background table has 16-byte rows like the match images, so its gathers reuse the
//           match table's byte offsets: -5 v_lshrrev per row, +10.9 KB of LDS per table.  Single-stage
//           kernel only: the two-stage kernel's 160 KiB are spoken for -- two images, the table and a
//           16-row ring -- and the ring cannot shrink: with the hand-shake once per five-row group the
//           producer may start a group only when the consumer has taken row p + 4 - kRD (+ hysteresis),
//           the consumer only when row c + 6 (+ hysteresis) is written, so fewer than 12 rows can leave
//           both waiting (an 8-row ring did: a hung parity test, round 3).
//   EM      E(j) is the maximum over the MATCH states only: with MD, DD <= 0 -- every model whose
//           transitions are log-probabilities -- D_k <= max_{i<k} M_i, so the delete states never decide
//           it (exactly: an add of a non-positive number never rounds up); -4 v_max3 per row.  A profile
//           with a positive MD or DD is flagged at upload and its pairs go to the row sweep (redo lists).
#ifndef DCP_QL_WPLANE
#define DCP_QL_WPLANE 1
#endif
#ifndef DCP_QL_IN16
#define DCP_QL_IN16 1
#endif
#ifndef DCP_QL_EM
#define DCP_QL_EM 1
#endif
constexpr bool kWPlane = DCP_QL_WPLANE != 0, kIn16 = DCP_QL_IN16 != 0, kEM = DCP_QL_EM != 0;

// Read-only tables are accessed through the constant address space: the data
// never changes during the kernel, and loads at wave-uniform addresses then
// become scalar (SMEM) loads instead of per-lane VMEM loads -- the compiler
// cannot prove that on its own because the kernel also stores to global memory.
typedef float const __attribute__((address_space(4))) cfloat;
__device__ __forceinline__ cfloat *as_const(float const *p)
{
    return (cfloat *)(unsigned long long)p;
}

__device__ __forceinline__ float ninf() { return -__builtin_inff(); }
__device__ __forceinline__ float mx3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float mx5(float a, float b, float c, float d, float e)
{
    return mx3(mx3(a, b, c), d, e); // two v_max3_f32
}

template <int G> struct QState
{
    float P[5][4 * G]; // predecessor maxima of M_k leaving row j', slot j' % 5
    float Q[5][4 * G]; // same for I_k
    float PN[5], PR[5]; // N (alt) and R (null): first tile only
    float PJ[5], PC[5]; // J and C: last tile only
};

// the lane's 13 special transitions (protein_profile_setup, per query length)
struct LaneXt
{
    float RR, SB, SN, NN, NB, ET, EC, CC, CT, EB, EJ, JJ, JB;
};

struct SweepOut
{
    float E, C, Rn;
};

// The tile's transitions, loaded once per sweep.  They are wave-uniform, so the
// compiler keeps them in SGPRs (free second operand of v_add_f32).  Re-loading
// them inside the row loop would cost nothing on the VALU, but SMEM shares the
// lgkmcnt counter with LDS and returns out of order: every use would force
// s_waitcnt lgkmcnt(0) and drain the LDS gathers in flight.
template <int G> struct TileTrans
{
    float ent[4 * G], mi[4 * G], ii[4 * G];
    float mm[4 * G + 1], im[4 * G + 1], dm[4 * G + 1], md[4 * G + 1], dd[4 * G + 1]; // [k]: edges INTO node k
};

template <int G> __device__ __forceinline__ void load_tile_trans(TileTrans<G> &t, cfloat *tt)
{
    constexpr int KT = 4 * G;
#pragma unroll
    for (int k = 0; k < KT; ++k)
    {
        t.ent[k] = tt[k * 8 + DCP_T_ENTRY];
        t.mi[k] = tt[k * 8 + DCP_T_MI];
        t.ii[k] = tt[k * 8 + DCP_T_II];
    }
#pragma unroll
    for (int k = 1; k <= KT; ++k)
    {
        t.mm[k] = tt[k * 8 + DCP_T_MM];
        t.im[k] = tt[k * 8 + DCP_T_IM];
        t.dm[k] = tt[k * 8 + DCP_T_DM];
        t.md[k] = tt[k * 8 + DCP_T_MD];
        t.dd[k] = tt[k * 8 + DCP_T_DD];
    }
    t.mm[0] = t.im[0] = t.dm[0] = t.md[0] = t.dd[0] = 0.0f; // unused: node 0 takes Xm / Xd
}

// What a row needs from memory, fetched one row ahead (software pipeline):
// gather group 0 of the match tables, the insert / background emissions and the
// previous tile's boundary values.  Groups 1.. are fetched at the top of the row
// and land while group 0 is being computed.
struct RowIn
{
    float4 e0[5];
    float eI[5], eN[5];
};

// Boundary values of the previous tile, fetched D (= 3) rows
// ahead into a register ring indexed by j % 5: a row's compute time (~0.6 us) is
// shorter than loaded-HBM latency, and with two waves per SIMD a late load stalls
// the SIMD.
// scratch plane access: wave-uniform base pointer + 32-bit BYTE offset, the form that maps to
// A tile's LDS image [G][1364 codes][4 nodes].  ROWS = false: a contiguous copy from emis_tiles (the DB holds both table
// layouts).  ROWS = true (DCP_DB_ONE_LAYOUT: only the row-sweep tables [1364][ldk] are resident, half the
// footprint): a.emis_tiles is emis_match, pm.tile_off the profile's first column there, and entry (group, code) is the
// 16 aligned bytes at row `code`, columns 8 t + 4 group .. + 3 -- inside the profile's own columns whenever the group's
// first node exists (a row ends in >= 1 column of -inf rounded up to 4; the columns behind core_size are -inf there as
// they are in the image); a group past the last node is -inf.  One strided 16-byte load per entry instead of a
// coalesced one: 2 728 per tile and block, against the ~10^6 cells the block then sweeps on it.
template <int G, bool ROWS>
__device__ __forceinline__ void stage_tile_image(float *tabM, dcp_qlane_args const &a, dcp_ql_prof const &pm, unsigned t,
                                                 unsigned first, unsigned stride)
{
    constexpr unsigned N4 = (unsigned)(G * NC);
    float4 *dst = reinterpret_cast<float4 *>(tabM);
    if constexpr (!ROWS)
    {
        float4 const *__restrict__ src = reinterpret_cast<float4 const *>(a.emis_tiles + pm.tile_off + (size_t)t * (N4 * 4u));
        for (unsigned i = first; i < N4; i += stride)
            dst[i] = src[i];
    }
    else
    {
        float const *__restrict__ rows = a.emis_tiles + pm.tile_off + t * (4u * (unsigned)G);
        unsigned const ldk = pm.ldk, M = pm.core_size;
        for (unsigned i = first; i < N4; i += stride)
        {
            unsigned const grp = i / (unsigned)NC, code = i - grp * (unsigned)NC;
            float4 v = make_float4(ninf(), ninf(), ninf(), ninf());
            if (t * (4u * (unsigned)G) + grp * 4u < M)
                v = *reinterpret_cast<float4 const *>(rows + (size_t)code * ldk + grp * 4u);
            dst[i] = v;
        }
    }
}

// global_load/store with an SGPR base and a VGPR offset (no 64-bit address arithmetic per access)
__device__ __forceinline__ float ld_off(float const *base, unsigned boff)
{
    return *reinterpret_cast<float const *>(reinterpret_cast<char const *>(base) + boff);
}
typedef uint32_t const __attribute__((address_space(1))) *gu32_ptr; // explicitly global: survives an asm pin
typedef uint16_t const __attribute__((address_space(1))) *gu16_ptr;
typedef char const __attribute__((address_space(1))) *gchar_ptr;
__device__ __forceinline__ void st_off(float *base, unsigned boff, float v)
{
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + boff) = v;
}

struct Ring
{
    float B[5], Xm[5], Xd[5], Em[5];
};

// Where a tile's boundary values (Xm, Xd, Em per row and lane) come from / go to:
//   IO_HBM  the per-block scratch planes in global memory (the single-stage kernel, and the
//           odd -> even tile boundary of the two-stage kernel)
//   IO_LDS  a ring of kRD rows in LDS between the two stages of a block of the two-stage kernel
//           (viterbi_qlane2_kernel): stage 0 sweeps the even tiles, stage 1 the odd ones a few rows
//           behind, so half of all boundaries never leave the CU
enum
{
    IO_HBM = 1,
    IO_LDS = 2
};
constexpr unsigned kRD = 16;      // rows of the LDS ring (power of two; at least 12: see IN16 above)
constexpr unsigned kRLanes = 256; // lanes (queries) per stage
constexpr unsigned kRingPlaneBytes = kRD * kRLanes * 4u;
// Tuning knobs of the ring hand-shake, measured on the C3 step (profiles/r02/qlane2_tuning.txt; ms of the
// launch): waiting for more rows than needed is what costs -- hysteresis 4: 2603-2908, 1: 2520-2535,
// 0: 2531; the initial skew (2..8 rows) and the poll sleep (0..4) do not matter; stages by wavefront
// halves (0-3 / 4-7) are a little better than even / odd wavefronts (2520 vs 2535).
#ifndef DCP_Q2_SKEW
#define DCP_Q2_SKEW 4
#endif
#ifndef DCP_Q2_HYST
#define DCP_Q2_HYST 1
#endif
#ifndef DCP_Q2_SLEEP
#define DCP_Q2_SLEEP 2
#endif
#ifndef DCP_Q2_PRIO
#define DCP_Q2_PRIO 0 // 1: consumer stage at s_setprio 1; 2: producer stage
#endif
#ifndef DCP_Q2_STAGEMAP
#define DCP_Q2_STAGEMAP 1 // 0: even / odd wavefronts; 1: wavefronts 0-3 / 4-7
#endif
constexpr unsigned kRingSkew = DCP_Q2_SKEW; // rows the consumer stage starts behind the producer
constexpr unsigned kRingHyst = DCP_Q2_HYST; // a side that has to wait waits for this many rows beyond what it needs
typedef float __attribute__((address_space(3))) lds_float;
typedef unsigned __attribute__((address_space(3))) lds_uint;

typedef char __attribute__((address_space(3))) lds_char;

// One wavefront's end of the ring.  Producer lane i and consumer lane i hold the same query, so the
// hand-off is wave to wave: flagP[lane] = last row the producer has written, flagC[lane] = last row the
// consumer has taken; a wavefront polls lane 0 of its partner (a broadcast read).  LDS executes one
// wavefront's operations in order: data first, flag second is all the release a producer needs; the
// consumer's acquire is the wait for the flag's value before it issues its data reads.
// No per-lane register is spent on the ring: a row's slot is (row - 1) % kRD, and the scratch-plane
// byte offset `off` = (row - 1) * 1024 + lane * 4 that every sweep keeps anyway yields both the slot
// address (off & 0x3fff) and the lane's flag address (off & 0x3ff); the ring sits at the start of the
// block's LDS so that planes and flags are immediate offsets of the ds instructions.
struct LdsLink
{
    lds_char *base;        // the block's LDS (a compile-time constant: the kernel's only LDS object)
    unsigned my_flag;      // byte offset of this stage's flag array (wave-uniform)
    unsigned peer_flag;    // byte offset of the partner wavefront's lane-0 flag (wave-uniform)
    unsigned seen;         // wave-uniform: the partner's progress as last read
};
// Block LDS of the two-stage kernel: all 160 KiB, laid out so that EVERY access keeps an immediate
// offset.  Gathers address a tile image as (window bits) + 16-bit immediate: image 0 sits at 0, image 1
// at 64 KiB with that base carried in the window register (gather_off<BASE>); the insert/background
// table is reached from (window >> 1), i.e. from base/2 = 32 KiB for stage 1, so ONE copy at 43 648
// serves both stages (immediates 43 648.. and 10 880..).  The ring's planes start at 7 x 16 KiB.
constexpr unsigned kL2TabIN = 43648u;                 // = one tile image: 2 groups x 1364 codes x 16 B
constexpr unsigned kL2FlagP = kL2TabIN + 2u * 1364u * 4u; // 54 560: producer flags [256]
constexpr unsigned kL2FlagC = kL2FlagP + kRLanes * 4u;
constexpr unsigned kL2Null = kL2FlagC + kRLanes * 4u;    // (round 2-3: null scores, stage 0 -> final stage; now parked in the planes)
constexpr unsigned kL2Task = kL2Null + kRLanes * 4u;     // the task word
constexpr unsigned kL2Abort = kL2Task + 4u;              // != 0: a wavefront of this block ran into the ring's poll bound during this task
constexpr unsigned kL2Err = kL2Task + 8u;                // the address of the scan's error word (8 bytes), for ring_wait
constexpr unsigned kL2Tab1 = 65536u;                   // image of the odd tile
constexpr unsigned kL2Ring = 7u * kRingPlaneBytes;     // 114 688 .. 163 840
constexpr unsigned kL2Bytes = kL2Ring + 3u * kRingPlaneBytes;
static_assert(kL2Task + 16u <= kL2Tab1 && kL2Tab1 + kL2TabIN <= kL2Ring && kL2Bytes == 160u * 1024u, "LDS layout");
static_assert((kL2Ring & (kRingPlaneBytes - 1u)) == 0u, "ring slots are addressed by OR");
static_assert(kRD >= 12u, "a shorter ring can deadlock the two stages (five-row hand-shake groups)");

// The ring's flags are relaxed workgroup-scope atomics and its data plain LDS accesses: `volatile` would
// make the backend's memory legalizer put s_waitcnt vmcnt(0) lgkmcnt(0) behind every access, which
// drains the row's whole software pipeline (first version: 35 % slower than the single-stage kernel).
// What is needed is ORDER only, and LDS gives it in hardware (one wavefront's operations execute in
// order); the compiler is held to it by empty asm statements that clobber "memory".
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
// The same fence, leaving a comment in the emitted ISA: tests/test_isa_order.py assembles this file and checks
// that between the marks the LDS operations of the ring hand-off come out in program order -- data stores,
// THEN the flag store; flag load, THEN the data loads; data loads, THEN the "taken" flag -- which is what
// the protocol relies on.  HARDWARE ASSUMPTION: the LDS executes one wavefront's DS
// operations in issue order, so a partner that observes the flag also observes the data written before it.
// (The HIP memory model gives no such edge for relaxed atomics; release / acquire atomics would, at the price
// of an s_waitcnt lgkmcnt(0) that drains the row's gathers in flight.)
#define DCP_ISA_MARK(name) asm volatile("; " name ::: "memory")
__device__ __forceinline__ unsigned flag_load(lds_uint *flag)
{
    return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void flag_store(lds_uint *flag, unsigned v)
{
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Polls are BOUNDED: the only hang this kernel has ever produced came from this loop (an
// experimental 8-row ring, round 3), and a hung GPU costs a whole lease.  A legitimate wait is a few rows of the
// partner's work -- microseconds; kRingSpinBound polls of >= 0.1 us each are a second.  A wavefront that runs into the
// bound sets the scan's error word (dcp_gpu_sync then fails the scan as a whole) and POISONS the flag it polls with
// a row number no sweep reaches: every later wait of this step passes at its first read -- the state "dead" lives in
// the LDS word, not in a register of the row loops.  (A first form kept it in the LdsLink and carried the error
// word's address with it: three more scalar registers live through six inlined sweeps cost the C3 launch 1.7 %; a
// second form read the global error word at every step, in front of the step's barrier: 0.9 % --
// profiles/r04/ring_bound_ab.txt.)  The address of the error word waits in LDS (kL2Err) for the slow path; the
// wavefront also sets the block's kL2Abort word, which every step checks behind its barrier (an LDS read), so the
// block's task winds down without sweeping; and no block takes another task once the global word is set (read
// together with the task counter's atomic): the grid drains.
#ifndef DCP_Q2_BOUND
#define DCP_Q2_BOUND 1 // 0: the unbounded loop of rounds 2-3 (pricing builds only)
#endif
constexpr unsigned kRingSpinBound = 1u << 23;
constexpr unsigned kRingPoison = 0x7fffff00u;
__device__ __forceinline__ unsigned ring_wait(LdsLink const &lk, unsigned need)
{
    lds_uint *flag = (lds_uint *)(lk.base + lk.peer_flag);
    unsigned v = __builtin_amdgcn_readfirstlane(flag_load(flag));
#if DCP_Q2_BOUND
    unsigned spins = 0;
#pragma nounroll
    while (v < need)
    {
        __builtin_amdgcn_s_sleep(DCP_Q2_SLEEP);
        v = __builtin_amdgcn_readfirstlane(flag_load(flag));
        if (__builtin_expect(++spins == kRingSpinBound, 0))
        {
            typedef unsigned __attribute__((address_space(1))) *gerr_ptr;
            gerr_ptr const err = *(gerr_ptr __attribute__((address_space(3))) *)(lk.base + kL2Err);
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag_store((lds_uint *)(lk.base + kL2Abort), 1u); // this block's task winds down at its next step
            flag_store(flag, kRingPoison);
            v = kRingPoison;
            // Nothing of this block may be pending where it joins the row loop again: the compiler's waitcnt pass
            // merges the two paths' counters, and with this path's LDS stores outstanding the row's first counted
            // wait -- lgkmcnt(5), which lets the five group-1 gathers just issued fly -- became lgkmcnt(0): 1 % of
            // the C3 launch for a block that never runs (profiles/r04/ring_bound_ab.txt).
            __builtin_amdgcn_s_waitcnt(0x0070); // vmcnt(0) lgkmcnt(0)
        }
    }
#else
    while (v < need)
    {
        __builtin_amdgcn_s_sleep(DCP_Q2_SLEEP);
        v = __builtin_amdgcn_readfirstlane(flag_load(flag));
    }
#endif
    DCP_ISA_MARK("DCP_RING_ACQUIRED"); // acquire: the ring reads that follow stay behind the flag read
    return v;
}
// boundary values of the row whose scratch-plane byte offset is `rowoff`
__device__ __forceinline__ float ring_ld(LdsLink const &lk, unsigned rowoff, unsigned plane)
{
    return *(lds_float *)(lk.base + ((rowoff & (kRingPlaneBytes - 1u)) | kL2Ring) + plane * kRingPlaneBytes);
}
__device__ __forceinline__ void ring_st(LdsLink const &lk, unsigned rowoff, unsigned plane, float v)
{
    *(lds_float *)(lk.base + ((rowoff & (kRingPlaneBytes - 1u)) | kL2Ring) + plane * kRingPlaneBytes) = v;
}
__device__ __forceinline__ void ring_publish(LdsLink const &lk, unsigned rowoff, unsigned row)
{
    flag_store((lds_uint *)(lk.base + lk.my_flag + (rowoff & (kRLanes * 4u - 1u))), row);
}

template <bool FIRST>
__device__ __forceinline__ void ring_fetch_x(Ring &r, int slot, float const *pXm, float const *pXd,
                                             float const *pEm, unsigned off)
{
    if constexpr (!FIRST)
    {
        r.Xm[slot] = ld_off(pXm, off);
        r.Xd[slot] = ld_off(pXd, off);
        r.Em[slot] = ld_off(pEm, off);
    }
}
template <bool FIRST>
__device__ __forceinline__ void ring_fetch_b(Ring &r, int slot, float const *pB, unsigned off)
{
    if constexpr (!kRecomputeB && !FIRST) r.B[slot] = ld_off(pB, off);
}
template <bool FIRST>
__device__ __forceinline__ void ring_fetch(Ring &r, int slot, float const *pB, float const *pXm,
                                           float const *pXd, float const *pEm, unsigned off)
{
    ring_fetch_x<FIRST>(r, slot, pXm, pXd, pEm, off);
    ring_fetch_b<FIRST>(r, slot, pB, off);
}

// Gather addresses.  Row `code` of a table group is 16 bytes (4 nodes), so the byte offset of
// the word of length l+1 ending at the window w is  (w * 16 & mask_l * 16) + first_l * 16: one
// v_and per length on the pre-shifted window, the constant part (and the group's base) folds
// into the ds_read offset field.  The insert / background table has 8-byte rows: half of it.
struct GatherOff
{
    unsigned a[5]; // (w & (4^(l+1) - 1)) * 16
};
// BASE: LDS byte address of the tile image when it does not start at the block's LDS base (stage 1 of
// the two-stage kernel: 64 KiB; a multiple of 16 KiB so that it does not overlap the window's 14 bits).
// It rides in the window's high bits -- one v_lshl_or instead of the shift, and masks that keep it --
// so that every gather keeps its 16-bit immediate offset.  The base is made opaque to the optimiser
// (an SGPR that passed through an empty asm): knowing the constant, InstCombine rewrites
// (w | B) & (m | B) into (w & m) | B and every gather pays a v_or (measured: +13 VALU per row).
template <unsigned BASE> struct GatherMasks
{
    unsigned base, m[4];
    __device__ __forceinline__ GatherMasks()
    {
        static_assert((BASE & 0x3fffu) == 0u, "the base must not overlap the window's 14 bits");
        base = BASE;
        if constexpr (BASE != 0u) asm volatile("" : "+s"(base));
#pragma unroll
        for (int l = 0; l < 4; ++l)
            m[l] = (((1u << (2 * l + 2)) - 1u) << 4) | base;
    }
};
template <unsigned BASE> __device__ __forceinline__ GatherOff gather_off(unsigned w, GatherMasks<BASE> const &gm)
{
    // kWPlane: the plane already holds w << 4; the second image's base is ORed in (an opaque SGPR: see GatherMasks)
    unsigned const wp = (DCP_QLANE_DIAG & 1) ? 0u : w;
    unsigned const w16 = kWPlane ? (BASE != 0u ? (wp | gm.base) : wp) : ((w << 4) | gm.base);
    GatherOff g;
#pragma unroll
    for (int l = 0; l < 4; ++l)
        g.a[l] = w16 & gm.m[l];
    // the window is 10 bits: all of it.  With window planes w16 IS the loaded value: the offset must live in
    // a register of its own, or the slot cannot be refilled in place while this row's gathers still need the
    // old value -- the allocator then refills another register and copies it into place at the loop's back
    // edge, and a copy of a value loaded a moment ago is a s_waitcnt vmcnt(0) (one v_mov: 2.3 cycles).
    if constexpr (kWPlane && BASE == 0u && !(DCP_QLANE_DIAG & 1)) asm volatile("v_mov_b32 %0, %1" : "=v"(g.a[4]) : "v"(w16));
    else g.a[4] = w16;
    return g;
}
template <int GROUP> __device__ __forceinline__ float4 gather_match(float const *tabM, GatherOff const &g, int l)
{
    constexpr unsigned first[5] = {0u, 4u, 20u, 84u, 340u};
    return *reinterpret_cast<float4 const *>(reinterpret_cast<char const *>(tabM) +
                                             (g.a[l] + (first[l] + (unsigned)GROUP * NC) * 16u));
}

template <int G, bool FIRST, bool LAST, bool IN16>
__device__ __forceinline__ void ql_fetch(RowIn &in, float const *tabM, float2 const *tabIN, GatherOff const &g)
{
    constexpr unsigned first[5] = {0u, 4u, 20u, 84u, 340u};
#pragma unroll
    for (int l = 0; l < 5; ++l)
    {
        in.e0[l] = gather_match<0>(tabM, g, l);
        // insert and background emissions of a word sit side by side: one ds_read_b64
        // IN16: the same byte offset as the match gather (the image base in it is the table's base too)
        float2 const v = *reinterpret_cast<float2 const *>(
            reinterpret_cast<char const *>(tabIN) + (IN16 ? g.a[l] + first[l] * 16u : (g.a[l] >> 1) + first[l] * 8u));
        in.eI[l] = v.x;
        in.eN[l] = v.y;
    }
}

template <int R> __device__ __forceinline__ float comp(float4 const &v)
{
    return R == 0 ? v.x : R == 1 ? v.y : R == 2 ? v.z : v.w;
}

// One row of one tile for this lane's query.  PH = j % 5 (compile time).
// `in` holds this row's prefetched inputs and is refilled for row j+1 (window
// wn) as soon as group 0 has consumed it.
template <int G, bool FIRST, bool LAST, int PH, int NT, int D, int IN = IO_HBM, int OUT = IO_HBM, unsigned TBASE = 0u,
          bool IN16 = kIn16>
__device__ __forceinline__ void ql_row(QState<G> &s, TileTrans<G> const &tr, float const *tabM,
                                       float2 const *tabIN, GatherOff &go,
                                       unsigned wn, RowIn &in, Ring &ring, float *pB, float *pXm,
                                       float *pXd, float *pEm, unsigned &off, LaneXt const &xt,
                                       bool live, bool at_end, bool &dirty, SweepOut &o,
                                       LdsLink const &lk, unsigned jrow, GatherMasks<TBASE> const &gm
#if DCP_QLANE_DIAG & 4
                                       , unsigned &off2, unsigned din
#endif
                                       )
{
    constexpr int KT = 4 * G;
    constexpr int s1 = (PH + 4) % 5, s2 = (PH + 3) % 5, s3 = (PH + 2) % 5, s4 = (PH + 1) % 5, s5 = PH;
    float const ni = ninf();

    // gather groups 1.. of this row now; they land while group 0 computes
    // (`go` = this row's gather offsets, computed when the previous row prefetched group 0)
    static_assert(G <= 2, "gather_match<GROUP> is instantiated for groups 0 and 1");
    float4 e[G > 1 ? G - 1 : 1][5];
    if constexpr (G > 1)
    {
#pragma unroll
        for (int l = 0; l < 5; ++l)
            e[0][l] = gather_match<1>(tabM, go, l);
    }

    // one counted wait for the whole prefetched `in` (the five group-1 gathers just issued stay in
    // flight) instead of the compiler's wait before every use: an s_waitcnt costs an issue slot
    // The same s_waitcnt covers what this row takes from global memory -- its ring slot and its
    // sequence word, issued D and five rows ago: everything but the VMEM operations of the D - 1
    // rows in between (loads and stores count alike, in issue order).
    {
        constexpr int per_row = ((FIRST || IN != IO_HBM) ? 0 : 3) + ((LAST || OUT != IO_HBM) ? 0 : 3) +
                                ((!kRecomputeB && !FIRST) ? 1 : 0) + ((!kRecomputeB && FIRST) ? 1 : 0) + 1;
        constexpr int vm = ((D < (int)kWD ? D : (int)kWD) - 1) * per_row; // <= 32
        constexpr int lgkm = G > 1 ? 5 : 0;
        __builtin_amdgcn_s_waitcnt((vm & 15) | ((vm >> 4) << 14) | (7 << 4) | (lgkm << 8));
    }
    __builtin_amdgcn_sched_barrier(0); // keep the gathers above: the scheduler would sink them to their use
    float Xm = ni, Xd = ni, E = ni, Bj = ring.B[PH];
    if constexpr (!FIRST)
    {
        Xm = ring.Xm[PH];
        Xd = ring.Xd[PH];
        E = ring.Em[PH];
    }
    float eI[5], eN[5];
#pragma unroll
    for (int l = 0; l < 5; ++l)
    {
        eI[l] = in.eI[l];
        eN[l] = in.eN[l];
    }

    if constexpr (FIRST || kRecomputeB)
    {
        // N(j); B0(j) = N(j) + NB  (S(j>0) = -inf).  The first tile also runs the null
        // model R(j).
        float const N = mx5(s.PN[s1] + eN[0], s.PN[s2] + eN[1], s.PN[s3] + eN[2],
                            s.PN[s4] + eN[3], s.PN[s5] + eN[4]);
        s.PN[PH] = N + xt.NN;
        Bj = N + xt.NB;
        if constexpr (FIRST)
        {
            float const Rn = mx5(s.PR[s1] + eN[0], s.PR[s2] + eN[1], s.PR[s3] + eN[2],
                                 s.PR[s4] + eN[3], s.PR[s5] + eN[4]);
            s.PR[PH] = Rn + xt.RR;
            o.Rn = at_end ? Rn : o.Rn;
            if constexpr (!kRecomputeB) st_off(pB, off, Bj);
        }
    }

    float pm = ni, pi = ni, pd = ni; // node k-1 of this row
    float m_even = ni;               // kEM: M of the even node, folded into E together with the odd one's
    auto node = [&](int k, float e0, float e1, float e2, float e3, float e4) {
        float const m = mx5(s.P[s1][k] + e0, s.P[s2][k] + e1, s.P[s3][k] + e2, s.P[s4][k] + e3,
                            s.P[s5][k] + e4);
        float const iv = mx5(s.Q[s1][k] + eI[0], s.Q[s2][k] + eI[1], s.Q[s3][k] + eI[2],
                             s.Q[s4][k] + eI[3], s.Q[s5][k] + eI[4]);
        float d, pin;
        if (k == 0)
        {
            d = Xd;
            pin = Xm;
        }
        else
        {
            d = fmaxf(pm + tr.md[k], pd + tr.dd[k]);
            pin = mx3(pm + tr.mm[k], pi + tr.im[k], pd + tr.dm[k]);
        }
        if constexpr (kEM)
        {
            // E(j) = max over the match states: D_k <= max_{i<k} M_i when MD, DD <= 0 (checked at upload)
            if (k & 1) E = mx3(E, m_even, m);
            else m_even = m;
        }
        else E = mx3(E, m, d);
        s.P[PH][k] = fmaxf(Bj + tr.ent[k], pin);
        s.Q[PH][k] = fmaxf(m + tr.mi[k], iv + tr.ii[k]);
        pm = m, pi = iv, pd = d;
    };

    // group 0 from the prefetched registers
    node(0, in.e0[0].x, in.e0[1].x, in.e0[2].x, in.e0[3].x, in.e0[4].x);
    node(1, in.e0[0].y, in.e0[1].y, in.e0[2].y, in.e0[3].y, in.e0[4].y);
    node(2, in.e0[0].z, in.e0[1].z, in.e0[2].z, in.e0[3].z, in.e0[4].z);
    node(3, in.e0[0].w, in.e0[1].w, in.e0[2].w, in.e0[3].w, in.e0[4].w);

    // fetch row j+D's boundary into its ring slot (slot PH itself when D == 5).  The slot's
    // old values died in node 0; the empty asm ties the load's address to group 0's E so the
    // scheduler cannot hoist the load above that point -- while old and new overlap the
    // allocator gives the new value another register and copies it into place at the loop's
    // back edge, and that copy waits for a load issued half a row earlier.  (B is in use
    // until the last node: its slot is refilled at the end of the row.)
    // (The row offset D rows ahead is folded into the plane pointers; the asm works on `off`
    // itself, so no copy of it is made.)
    if constexpr (!FIRST && IN == IO_HBM)
    {
        asm volatile("" : "+v"(off) : "v"(E));
#if DCP_QLANE_DIAG & 4
        ring_fetch_x<FIRST>(ring, (PH + D) % 5, pXm + din, pXd + din, pEm + din, off);
#else
        ring_fetch_x<FIRST>(ring, (PH + D) % 5, pXm + D * NT, pXd + D * NT, pEm + D * NT, off);
#endif
    }

    // `in` is consumed: refill it for row j+1 while the other groups compute
    // group 1's gathers (issued at the top of the row) are the only LDS reads in flight: one wait
    __builtin_amdgcn_s_waitcnt(0xC07F | (0 << 8)); // lgkmcnt(0)
    go = gather_off<TBASE>(wn, gm);
    ql_fetch<G, FIRST, LAST, IN16>(in, tabM, tabIN, go);
    if constexpr (!FIRST && IN == IO_LDS)
    {
        // row j+1's boundary from the LDS ring (the sweep loop made sure the producer has written it);
        // then tell the producer the slot is taken -- LDS runs these in order
        constexpr int sl = (PH + 1) % 5;
        DCP_ISA_MARK("DCP_RING_TAKE"); // not above the availability check of this row
        unsigned const nxt = off + kRLanes * 4u; // row j+1
        ring.Xm[sl] = ring_ld(lk, nxt, 0);
        ring.Xd[sl] = ring_ld(lk, nxt, 1);
        ring.Em[sl] = ring_ld(lk, nxt, 2);
        DCP_ISA_MARK("DCP_RING_TAKEN");
        ring_publish(lk, off, jrow + 1u);
        DCP_ISA_MARK("DCP_RING_TAKE_END");
    }

#pragma unroll
    for (int g = 1; g < G; ++g)
    {
        float4 const *eg = e[g - 1];
        node(4 * g + 0, eg[0].x, eg[1].x, eg[2].x, eg[3].x, eg[4].x);
        node(4 * g + 1, eg[0].y, eg[1].y, eg[2].y, eg[3].y, eg[4].y);
        node(4 * g + 2, eg[0].z, eg[1].z, eg[2].z, eg[3].z, eg[4].z);
        node(4 * g + 3, eg[0].w, eg[1].w, eg[2].w, eg[3].w, eg[4].w);
    }

    if constexpr (!kRecomputeB && !FIRST)
    {
        asm volatile("" : "+v"(off) : "v"(pm));
        ring_fetch_b<FIRST>(ring, (PH + D) % 5, pB + D * NT, off);
    }

    if constexpr (!LAST)
    {
        // edges into the next tile's first node
        float const oXm = mx3(pm + tr.mm[KT], pi + tr.im[KT], pd + tr.dm[KT]);
        float const oXd = fmaxf(pm + tr.md[KT], pd + tr.dd[KT]);
        if constexpr (OUT == IO_LDS)
        {
            DCP_ISA_MARK("DCP_RING_DATA"); // not above the free-slot check of this row
            ring_st(lk, off, 0, oXm);
            ring_st(lk, off, 1, oXd);
            ring_st(lk, off, 2, E);
            DCP_ISA_MARK("DCP_RING_FLAG");
            ring_publish(lk, off, jrow); // after the data: LDS keeps this wavefront's order
            DCP_ISA_MARK("DCP_RING_DATA_END");
        }
        else
        {
#if DCP_QLANE_DIAG & 4
            unsigned const offw = off2;
#else
            unsigned const offw = off;
#endif
            st_off(pXm, offw, oXm);
            st_off(pXd, offw, oXd);
            st_off(pEm, offw, E);
        }
    }
    else
    {
        float const J = mx5(s.PJ[s1] + eN[0], s.PJ[s2] + eN[1], s.PJ[s3] + eN[2], s.PJ[s4] + eN[3],
                            s.PJ[s5] + eN[4]);
        float const C = mx5(s.PC[s1] + eN[0], s.PC[s2] + eN[1], s.PC[s3] + eN[2], s.PC[s4] + eN[3],
                            s.PC[s5] + eN[4]);
        // did E(j) -> B(j) or J(j) -> B(j) beat the B(j) this sweep ran with?  Then this
        // pair goes to the redo list (nothing computed from here on is used)
        float const B1 = fmaxf(E + xt.EB, J + xt.JB);
        dirty = dirty || (live && B1 > Bj);
        s.PJ[PH] = fmaxf(E + xt.EJ, J + xt.JJ);
        s.PC[PH] = fmaxf(E + xt.EC, C + xt.CC);
        o.E = at_end ? E : o.E;
        o.C = at_end ? C : o.C;
    }
}

// Sweep one tile over rows 1..L of this lane's query.  Scratch planes are
// addressed as (wave-uniform plane base) + (32-bit lane/row offset).
template <int G, bool FIRST, bool LAST, int NT, int D, int IN = IO_HBM, int OUT = IO_HBM, unsigned TBASE = 0u,
          bool IN16 = kIn16>
__device__ __forceinline__ void ql_sweep(cfloat *tt, float const *tabM, float2 const *tabIN,
                                         uint32_t const *__restrict__ wordsT, unsigned rowbase,
                                         unsigned L, unsigned Lwave, bool active, float *sc,
                                         size_t plane, unsigned tid, LaneXt const &xt, bool &dirty,
                                         SweepOut &o, LdsLink lk
#if DCP_QLANE_DIAG & 4
                                         , unsigned tile_odd
#endif
                                         )
{
    constexpr int KT = 4 * G;
#if DCP_QLANE_DIAG & 4
    // timing build: what a 2-stage tile pipeline would save -- the even -> odd tile boundary goes through
    // ONE plane row (stays in L2, like an LDS ring would keep it on chip), the odd -> even one through HBM
    unsigned const rowstep = tile_odd ? 0u : (unsigned)NT;      // input side of this tile
    unsigned const rowstep_out = tile_odd ? (unsigned)NT : 0u;  // output side
    unsigned const din = (unsigned)D * rowstep;
    unsigned off2 = tid * 4u;
#else
    constexpr unsigned rowstep = (DCP_QLANE_DIAG & 2) ? 0u : (unsigned)NT;
#endif
    float const ni = ninf();
    TileTrans<G> tr;
    load_tile_trans<G>(tr, tt);
    QState<G> s;
#pragma unroll
    for (int h = 0; h < 5; ++h)
    {
#pragma unroll
        for (int k = 0; k < KT; ++k)
            s.P[h][k] = ni, s.Q[h][k] = ni;
        s.PN[h] = ni, s.PR[h] = ni, s.PJ[h] = ni, s.PC[h] = ni;
    }
    // row 0: S = 0, B(0) = S + SB; every tile's nodes can be entered from B(0)
    {
        float const B0 = 0.0f + xt.SB;
#pragma unroll
        for (int k = 0; k < KT; ++k)
            s.P[0][k] = B0 + tr.ent[k];
        s.PN[0] = 0.0f + xt.SN;
        s.PR[0] = 0.0f;
    }
    // plane 3 (B0) exists only when the tiles do not recompute N themselves
    float *pXm = sc, *pXd = sc + plane, *pEm = sc + 2 * plane, *pB = sc + (kRecomputeB ? 0 : 3) * plane;
    // byte offset of ((rowbase + j - 1) * NT + tid): the group's rows are rows rowbase + 1 .. rowbase + L of the slot's
    // column (rowbase is wave-uniform; the first group of a slot has rowbase 0)
    unsigned off = (rowbase * (unsigned)NT + tid) * 4u;

    // sequence window: w = row j, wn = row j+1 (the word past the last base is padding)
    // Every lane of the wavefront executes every row up to Lwave -- no per-lane
    // branch around the row body: values loaded inside a divergent `if` would reach
    // the next row through phi copies, and a copy needs its load complete, which
    // drains every prefetch at the end of each row.  A lane past its own length
    // (or without a query) computes garbage into its own scratch column and its
    // own registers; its results were captured at row L (`at_end`), `live` guards
    // the only externally visible effects.
    // Sequence words come from the block's transposed plane (word index uniform, lane =
    // column): every row issues ONE coalesced load, five rows ahead of its use like the
    // boundary ring, so the in-order vmcnt wait for it never drains younger prefetches.
    // (A load every 16th row inside `if ((pos & 15) == 0)` reaches the next row through a
    // phi copy and costs a vmcnt(0) drain each time.)
    unsigned j = 1;
    static_assert(kWPlane, "the packed-word sequence planes are gone (round 4): a slot's groups are addressed by plane row");
    typedef uint16_t wq_t;
    wq_t wq[5];
    unsigned w;
    unsigned lane2 = tid * 2u; // byte offset of this lane in a window-plane row
    // this group's rows of the block's window plane [row][lane] of uint16: row r holds (window of row r) << 4
    // (row 0 unused, the rows past the longest member up to the region's end are there to be prefetched).
    // wq[r % 5] = row r's window, r = 2..6: a slot is used by exactly one phase (row j reads slot (j + 1) % 5 for its
    // gather prefetch and then refills it for row j + 6), so no value ever moves between registers.
    gu16_ptr const wpl = (gu16_ptr)wordsT + (size_t)rowbase * (unsigned)NT;
    // (lane2 is loop-invariant over a whole task: the optimiser computes it once per kernel and the allocator spills
    // it; the reload must come HERE, in front of the wait below -- a scratch_load pending on the way into the row
    // loop is a static vmcnt(0) in the loop body -- so the value is re-defined by an asm the reload has to feed)
    asm volatile("" : "+v"(lane2));
    {
        w = wpl[NT + tid];
#pragma unroll
        for (int r = 2; r < 7; ++r)
            wq[r % 5] = wpl[r * NT + tid];
        __builtin_amdgcn_sched_barrier(0); // all seven are issued in front of the wait
        // all seven are waited for here, once per sweep: a load still pending on the way into the row loop
        // would make the compiler's (static) wait for its first use drain the loop's own prefetches every
        // fifth row -- the back-edge path has 16+ younger operations behind such a value, the entry path none
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    }
    RowIn in;
    Ring ring;
#pragma unroll
    for (int l = 0; l < 5; ++l)
    {
        in.e0[l] = float4{ni, ni, ni, ni}, in.eI[l] = ni, in.eN[l] = ni;
        ring.B[l] = ring.Xm[l] = ring.Xd[l] = ring.Em[l] = ni;
    }
    GatherMasks<TBASE> const gm;
    GatherOff go = gather_off<TBASE>(w, gm);
    ql_fetch<G, FIRST, LAST, IN16>(in, tabM, tabIN, go);
    if constexpr (!FIRST && IN == IO_LDS)
    {
        // start kRingSkew rows behind the producer, then row 1 -> slot 1
        static_assert(NT == (int)kRLanes, "the ring's row stride is the scratch planes' row stride");
        unsigned const need0 = rowbase + (Lwave < kRingSkew ? Lwave : kRingSkew);
        lk.seen = ring_wait(lk, need0);
        DCP_ISA_MARK("DCP_RING_TAKE");
        ring.Xm[1] = ring_ld(lk, off, 0); // row 1
        ring.Xd[1] = ring_ld(lk, off, 1);
        ring.Em[1] = ring_ld(lk, off, 2);
        DCP_ISA_MARK("DCP_RING_TAKEN");
        ring_publish(lk, off, rowbase + 1u);
        DCP_ISA_MARK("DCP_RING_TAKE_END");
    }
    else
    {
        lk.seen = 0u;
#pragma unroll
        for (int r = 0; r < D; ++r) // rows 1..D -> slots 1..D (mod 5)
            ring_fetch<(FIRST || IN != IO_HBM)>(ring, (r + 1) % 5, pB, pXm, pXd, pEm, off + (unsigned)r * rowstep * 4u);
    }

// timing build (bit 3): the scratch planes wrap after DCP_QLANE_DIAG_ROWS rows -- a block's planes then are
// DCP_QLANE_DIAG_ROWS x 3 KB instead of (L + 8) x 3 KB: what keeping them in the L2 / Infinity Cache would buy
#if DCP_QLANE_DIAG & 8
#ifndef DCP_QLANE_DIAG_ROWS
#define DCP_QLANE_DIAG_ROWS 256u
#endif
#define QL_DIAG8_WRAP off &= (DCP_QLANE_DIAG_ROWS * (unsigned)NT * 4u - 1u);
#else
#define QL_DIAG8_WRAP
#endif
#if DCP_QLANE_DIAG & 4
#define QL_DIAG4_ARGS , off2, din
#define QL_DIAG4_STEP off2 += rowstep_out * 4u;
#else
#define QL_DIAG4_ARGS
#define QL_DIAG4_STEP
#endif
#define QL_ROW(PH)                                                                         \
    {                                                                                      \
        /* base of row j+2 sits at position j+1 */                                         \
        unsigned const pos = j + 1u;                                                       \
        ql_row<G, FIRST, LAST, PH, NT, D, IN, OUT, TBASE, IN16>(s, tr, tabM, tabIN, go,              \
                                   (unsigned)wq[(PH + 1) % 5], in, ring, pB, pXm,                    \
                                   pXd, pEm, off, xt, active && j <= L, active && j == L,  \
                                   dirty, o, lk, rowbase + j, gm QL_DIAG4_ARGS);           \
        {   /* the slot just read (row j+1's window, loaded five rows ago) takes row j+6's: one coalesced */ \
            /* load, SGPR row pointer + the lane's byte offset re-derived from `off`, as below            */ \
            /* wq[] is uint16: the zero-extension then happens where the value is USED (folded into the   */ \
            /* gather masks); as `unsigned` it is a v_and at the loop's back edge, on a value loaded a      */ \
            /* moment ago -- s_waitcnt vmcnt(0) in every fifth row                                          */ \
            gu16_ptr wrow = wpl + (pos + 5u) * (unsigned)NT;                                     \
            asm volatile("" : "+s"(wrow));                                                   \
            asm volatile("" : "+v"(lane2)); /* keeps the 32-bit lane offset next to the load */ \
            wq[(PH + 1) % 5] = *(gu16_ptr)((gchar_ptr)wrow + lane2);                          \
        }                                                                                    \
        off += rowstep * 4u;                                                                     \
        QL_DIAG8_WRAP                                                                            \
        QL_DIAG4_STEP                                                                            \
        ++j;                                                                               \
        /* keep the scheduler from pulling the next row's loads up here: the only   */     \
        /* cross-row traffic is the explicit prefetch above (register budget);      */     \
        /* mask 6: VALU / SALU arithmetic may still move across (+0.3 %)            */     \
        __builtin_amdgcn_sched_barrier(6);                                                 \
    }
    // Ring hand-shake once per group of rows, not per row: the five-row body stays one basic block.
    // A consumer row j prefetches ring row j+1, so a group starting at j needs rows <= j + n written; a
    // producer group writes rows j .. j+n-1, whose slots must have been taken (row - kRD).  A side that
    // has to wait waits for kRingHyst rows more than it needs, so that it polls once per burst.
    // (flags count PLANE rows, rowbase + j: a slot's groups follow each other in one step without the counters restarting)
    auto ring_sync = [&](unsigned n) {
        if constexpr (!FIRST && IN == IO_LDS)
        {
            unsigned const need = rowbase + (j + n < Lwave ? j + n : Lwave);
            if (lk.seen < need) lk.seen = ring_wait(lk, need + kRingHyst < rowbase + Lwave ? need + kRingHyst : rowbase + Lwave);
        }
        if constexpr (!LAST && OUT == IO_LDS)
        {
            unsigned const last = rowbase + j + n - 1u;
            if (last > lk.seen + kRD) lk.seen = ring_wait(lk, last - kRD + kRingHyst);
        }
    };
    while (j + 4 <= Lwave)
    {
        ring_sync(5u);
        QL_ROW(1) QL_ROW(2) QL_ROW(3) QL_ROW(4) QL_ROW(0)
    }
    ring_sync(4u);
    if (j <= Lwave) QL_ROW(1)
    if (j <= Lwave) QL_ROW(2)
    if (j <= Lwave) QL_ROW(3)
    if (j <= Lwave) QL_ROW(4)
#undef QL_ROW
}

// [code] = {insert, background} emissions of the task's profile, 8-byte rows or (IN16) 16-byte rows
template <bool IN16>
__device__ __forceinline__ void fill_tab_in(float *tab, float const *__restrict__ gi, float const *__restrict__ gn, unsigned t,
                                            unsigned nthreads)
{
    for (unsigned i = t; i < (unsigned)NC; i += nthreads)
    {
        if constexpr (IN16) *reinterpret_cast<float4 *>(tab + 4u * i) = float4{gi[i], gn[i], 0.0f, 0.0f};
        else *reinterpret_cast<float2 *>(tab + 2u * i) = float2{gi[i], gn[i]};
    }
}

// ---- a wavefront slot's groups (dcp_ql_group, dcp_kernels.h) ---------------------------------------------------
typedef dcp_ql_group const __attribute__((address_space(4))) cgroup;
typedef uint32_t const __attribute__((address_space(4))) cu32;

// What a lane holds of its group's query while a tile is swept.
struct GroupLane
{
    unsigned q, L, Lwave, rowbase;
    bool has;
    LaneXt xt;
};
// lane: 0..63 within the wavefront.  Everything about the group is wave-uniform (scalar loads).
__device__ __forceinline__ void load_group(dcp_qlane_args const &a, unsigned gi, unsigned lane, GroupLane &g)
{
    cgroup *gd = (cgroup *)(unsigned long long)(a.groups + gi);
    unsigned const qfirst = gd->qfirst, nq = gd->nq;
    g.rowbase = gd->rowbase;
    g.has = lane < nq;
    g.q = g.has ? a.qorder[qfirst + lane] : 0u;
    g.L = g.has ? a.seq_len[g.q] : 0u;
    g.Lwave = gd->lmax; // = the wavefront's maximum of L: the host sorted by length
    float const *__restrict__ x = a.xtrans + (size_t)g.q * DCP_XSTRIDE;
    g.xt.RR = x[DCP_X_RR], g.xt.SB = x[DCP_X_SB], g.xt.SN = x[DCP_X_SN], g.xt.NN = x[DCP_X_NN];
    g.xt.NB = x[DCP_X_NB], g.xt.ET = x[DCP_X_ET], g.xt.EC = x[DCP_X_EC], g.xt.CC = x[DCP_X_CC];
    g.xt.CT = x[DCP_X_CT], g.xt.EB = x[DCP_X_EB], g.xt.EJ = x[DCP_X_EJ], g.xt.JJ = x[DCP_X_JJ];
    g.xt.JB = x[DCP_X_JB];
    // Everything loaded above is complete -- and known to the compiler's waitcnt pass to be complete -- before a sweep
    // starts: a load still pending on the way into a row loop (a transition only the epilogue uses, say) makes the
    // pass put a static s_waitcnt vmcnt(0) into the loop body, which drains the boundary prefetch every fifth row
    // (profiles/tools/isa_loops.py shows it).
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
}
// Per-lane values that outlive a sweep (the null score from the first tile's sweep to the last one's; in the
// two-stage kernel also alt score and feedback flag from the last sweep to the block's epilogue) are parked in a
// spare row of the group's region of the block's scratch planes: rows rowbase + L + 1 .. + 10 are prefetched from but
// never written by a sweep.  Byte offset of that row's entry for lane column `col`:
__device__ __forceinline__ unsigned park_off(GroupLane const &g, unsigned NT, unsigned col)
{
    return ((g.rowbase + g.Lwave + 8u) * NT + col) * 4u;
}

// The pair's results: scores (dense matrices if kept), LRT filter + hit record, or the redo list.
__device__ __forceinline__ void ql_publish(dcp_qlane_args const &a, dcp_ql_prof const &pm, unsigned q, float nul, float alt,
                                           bool redo)
{
    if (redo)
    {
        // B0 was not the solution for this pair (or its profile has a positive MD / DD): the row-sweep kernel scores it
        unsigned const cls = pm.cls;
        unsigned const i = atomicAdd(a.redo_n + cls, 1u);
        if (i < a.redo_cap[cls]) a.redo[a.redo_base[cls] + i] = dcp_pair{q, pm.rs_slot};
        else *a.redo_overflow = 1u; // host falls back to the row sweep for the whole scan
        return;
    }
    size_t const oi = (size_t)q * a.nprof_total + pm.pidx;
    if (a.out_null) a.out_null[oi] = nul;
    if (a.out_alt) a.out_alt[oi] = alt;
    // xmath_lrt_f32 + filter of scan_thread.c:121-123
    float const lrt = -2 * (nul - alt);
    if (__builtin_isfinite(lrt) && !(lrt < a.lrt_threshold))
    {
        unsigned const h = atomicAdd(a.nhits, 1u);
        if (h < a.hit_cap) a.hits[h] = dcp_hit{a.q_base + q, pm.pidx, nul, alt};
    }
}

} // namespace

template <int G, int NT, int D, bool ROWS = false>
__global__ __launch_bounds__(NT, NT / 128) void viterbi_qlane_kernel(dcp_qlane_args a)
{
    constexpr int KT = 4 * G;
    constexpr int TAB_FLOATS = G * NC * 4;
    __shared__ __attribute__((aligned(16))) float lds[TAB_FLOATS + (kIn16 ? 4 : 2) * NC];
    __shared__ unsigned s_task;
    float *tabM = lds;
    float2 *tabIN = reinterpret_cast<float2 *>(lds + TAB_FLOATS); // [code] = {insert, background}
    unsigned const tid = threadIdx.x;
    size_t const plane = (size_t)a.plane_rows * (unsigned)NT;
    float *const sc = a.scratch + (size_t)blockIdx.x * kPlanes * plane; // wave-uniform base
    constexpr unsigned SLOTS = (unsigned)NT / 64u;

    for (;;)
    {
        if (tid == 0) s_task = atomicAdd(a.task_counter, 1u);
        __syncthreads();
        unsigned const task = __builtin_amdgcn_readfirstlane(s_task); // uniform: transitions via SMEM
        __syncthreads();
        if (task >= a.ntasks) break;
        // biggest profiles first (metas are sorted by ascending size)
        unsigned const slot = a.nprof - 1u - task / a.nqblocks;
        unsigned const qb = task % a.nqblocks;
        dcp_ql_prof const pm = a.profs[slot];
        unsigned const T = pm.ntiles;
        // this wavefront's slot of the block: its groups are swept one after the other, tile by tile
        unsigned const sidx = __builtin_amdgcn_readfirstlane(qb * SLOTS + (tid >> 6));
        cu32 *sf = (cu32 *)(unsigned long long)a.slot_first;
        unsigned const g0 = sf[sidx], g1 = sf[sidx + 1u];
        // wave-uniform (readfirstlane: the compiler cannot see that through the load)
        uint32_t const *__restrict__ wordsT = a.words_t + __builtin_amdgcn_readfirstlane(a.wt_off[qb]);

        // the profile's insert and background tables stay in LDS for the whole task
        {
            float const *__restrict__ gi = a.emis_insert + (size_t)pm.pidx * NC;
            float const *__restrict__ gn = a.emis_null + (size_t)pm.pidx * NC;
            fill_tab_in<kIn16>(lds + TAB_FLOATS, gi, gn, tid, (unsigned)NT);
        }

        GroupLane g;
        bool const single = g1 - g0 == 1u;
        bool t_loaded = false;
        for (unsigned t = 0; t < T; ++t)
        {
            __syncthreads(); // previous tile's readers are done with tabM
            stage_tile_image<G, ROWS>(tabM, a, pm, t, tid, (unsigned)NT);
            __syncthreads();
            cfloat *tt = as_const(a.ttrans + pm.ttrans_off + (size_t)t * (KT + 1) * 8);
            bool const first = t == 0, last = t + 1 == T;
            for (unsigned gi = g0; gi < g1; ++gi)
            {
                // a slot with ONE group (every uniform batch) keeps its lane state in registers over the task's tiles
                if (!(single && t_loaded)) load_group(a, gi, tid & 63u, g);
                t_loaded = true;
                if (g.Lwave == 0u) continue;
                SweepOut o{ninf(), ninf(), ninf()};
                bool dirty = false;
#if DCP_QLANE_DIAG & 4
#define QL_DIAG4_TILE , (t & 1u)
#else
#define QL_DIAG4_TILE
#endif
#define QL_SWEEP(F, L_)                                                                          \
    ql_sweep<G, F, L_, NT, D>(tt, tabM, tabIN, wordsT, g.rowbase, g.L, g.Lwave, g.has, sc, plane, tid, g.xt, dirty, o, LdsLink{} QL_DIAG4_TILE)
                if (first && last) QL_SWEEP(true, true);
                else if (first) QL_SWEEP(true, false);
                else if (last) QL_SWEEP(false, true);
                else QL_SWEEP(false, false);
#undef QL_SWEEP
                // the null score waits for the last tile in the group's spare plane row (written and read by this lane)
                unsigned const po = park_off(g, (unsigned)NT, tid);
                if (first && !last) st_off(sc, po, o.Rn);
                if (last && g.has)
                {
                    float const nul = first ? o.Rn : ld_off(sc, po);
                    ql_publish(a, pm, g.q, nul, fmaxf(o.E + g.xt.ET, o.C + g.xt.CT), dirty || (kEM && pm.needs_exact_e));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Two-stage variant: a block is 512 threads = 2 stages x 256 queries on ONE profile.  Stage 0 (waves
// 0-3) sweeps the even tiles, stage 1 (waves 4-7) the odd tiles a few rows behind; wave w of stage 1
// holds the same 64 queries as wave w of stage 0 and shares its SIMD.  The boundary stage 0 -> stage 1
// (tile 2k -> 2k+1) goes through a 16-row LDS ring with per-wavefront progress flags; only the
// boundary stage 1 -> stage 0 (tile 2k+1 -> 2k+2) still goes through the HBM scratch planes: half of
// the single-stage kernel's HBM traffic.  Both tile images live in LDS (2 x 43.6 KB + the ring's 48 KB:
// one block per CU, still two wavefronts per SIMD).  A step = one tile per stage between two block
// barriers; the arithmetic of a row is ql_row's, unchanged, so results are bit-identical.
// Measured motive (profiles/r02/diag_builds.txt): with the planes collapsed to one row (no HBM traffic)
// the single-stage kernel runs 26 % faster, with every other boundary collapsed 12 % faster.
// ---------------------------------------------------------------------------------------------------
template <int G, int D, bool ROWS = false>
__global__ __launch_bounds__(512, 2) void viterbi_qlane2_kernel(dcp_qlane_args a)
{
    constexpr int NT = (int)kRLanes; // lanes per stage
    constexpr int KT = 4 * G;
    __shared__ __attribute__((aligned(16))) float lds[kL2Bytes / 4u]; // the kernel's only LDS object: at address 0
    // Stage 0 = wavefronts 0-3, stage 1 = wavefronts 4-7; wavefront w of stage 1 holds the same 64 queries as
    // wavefront w of stage 0.  (The other map -- even / odd wavefronts -- measured 0.6 % slower.)
    unsigned const wv = threadIdx.x >> 6;
#if DCP_Q2_STAGEMAP == 0
    unsigned const stage = __builtin_amdgcn_readfirstlane(wv & 1u);
    unsigned const tid = ((wv >> 1) << 6) | (threadIdx.x & 63u); // lane within the stage = query slot of the block
#else
    unsigned const stage = __builtin_amdgcn_readfirstlane(wv >> 2);
    unsigned const tid = threadIdx.x & 255u;
#endif
    unsigned *const s_task_p = reinterpret_cast<unsigned *>(lds + kL2Task / 4u);
    float *const tabM = lds + (stage ? kL2Tab1 / 4u : 0u);
    size_t const plane = (size_t)a.plane_rows * (unsigned)NT;
    float *const sc = a.scratch + (size_t)blockIdx.x * kPlanes * plane;
    constexpr unsigned SLOTS = (unsigned)NT / 64u;

#if DCP_Q2_PRIO == 1
    if (stage == 1u) __builtin_amdgcn_s_setprio(1);
#elif DCP_Q2_PRIO == 2
    if (stage == 0u) __builtin_amdgcn_s_setprio(1);
#endif
    LdsLink lk;
    lk.base = (lds_char *)lds;
    lk.my_flag = stage == 0u ? kL2FlagP : kL2FlagC;
    lk.peer_flag = (stage == 0u ? kL2FlagC : kL2FlagP) + (tid & ~63u) * 4u;
    lk.seen = 0u;
    if (threadIdx.x == 0) *(unsigned **)(lds + kL2Err / 4u) = a.ring_error; // (a global address: ring_wait reads it back as one)
    unsigned *const s_abort_p = reinterpret_cast<unsigned *>(lds + kL2Abort / 4u);

    for (;;)
    {
        // (once the error word is set no block takes another task: the grid drains; load and atomic are in flight together)
        if (threadIdx.x == 0)
        {
            unsigned const failed = __hip_atomic_load(a.ring_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned const next = atomicAdd(a.task_counter, 1u);
            *s_task_p = failed ? 0xffffffffu : next;
            *s_abort_p = 0u;
        }
        __syncthreads();
        unsigned const task = __builtin_amdgcn_readfirstlane(*s_task_p);
        __syncthreads();
        if (task >= a.ntasks) break;
        unsigned const slot = a.nprof - 1u - task / a.nqblocks; // biggest profiles first
        unsigned const qb = task % a.nqblocks;
        dcp_ql_prof const pm = a.profs[slot];
        unsigned const T = pm.ntiles;
        // wavefront w of either stage sweeps slot w of the block: the same groups in the same order
        unsigned const sidx = __builtin_amdgcn_readfirstlane(qb * SLOTS + (tid >> 6));
        cu32 *sf = (cu32 *)(unsigned long long)a.slot_first;
        unsigned const g0 = sf[sidx], g1 = sf[sidx + 1u];
        uint32_t const *__restrict__ wordsT = a.words_t + __builtin_amdgcn_readfirstlane(a.wt_off[qb]);
        {
            float const *__restrict__ gi = a.emis_insert + (size_t)pm.pidx * NC;
            float const *__restrict__ gn = a.emis_null + (size_t)pm.pidx * NC;
            fill_tab_in<false>(lds + kL2TabIN / 4u, gi, gn, threadIdx.x, 512u);
        }
        // TEST ONLY (dcp_qlane_args::ring_stall, 0 in the shipped library): stage 0 sits out the first step of the
        // first task, so stage 1 runs into ring_wait's bound -- the path that must end in an error, not in a hang
        bool const stalled = a.ring_stall != 0u && task == 0u && stage == 0u;

        unsigned const nsteps = (T + 1u) / 2u;
        unsigned const final_stage = (T - 1u) & 1u; // the stage that sweeps the last tile
        GroupLane g;
        bool const single = g1 - g0 == 1u;
        bool t_loaded = false;
        for (unsigned st = 0; st < nsteps; ++st)
        {
            unsigned const t = 2u * st + stage;
            bool const mine = t < T; // the last step of an odd profile has no odd tile
            __syncthreads();       // previous step: both stages are done with their images and with the ring
            if (mine) stage_tile_image<G, ROWS>(tabM, a, pm, t, tid, (unsigned)NT);
            flag_store((lds_uint *)(lk.base + lk.my_flag + tid * 4u), 0u); // row counters restart with every step
            __syncthreads();
            // (the scan has failed -- some wavefront's hand-shake ran into its bound: no more sweeps, the task winds down)
            if (!mine || (stalled && st == 0u) || __builtin_amdgcn_readfirstlane(*s_abort_p) != 0u) continue;
            cfloat *tt = as_const(a.ttrans + pm.ttrans_off + (size_t)t * (KT + 1) * 8);
            bool const first = t == 0u, last = t + 1u == T;
#if DCP_QLANE_DIAG & 4
#error "the two-stage kernel has no DIAG=4 build"
#endif
            for (unsigned gi = g0; gi < g1; ++gi)
            {
                // a slot with ONE group (every uniform batch) keeps its lane state in registers over the task's tiles
                if (!(single && t_loaded)) load_group(a, gi, tid & 63u, g);
                t_loaded = true;
                if (g.Lwave == 0u) continue;
                SweepOut o{ninf(), ninf(), ninf()};
                bool dirty = false;
    /* tabIN as seen from the gather offsets: its 8-byte rows are reached from (window >> 1), i.e. from half the image base */
#define QL2_SWEEP(F, L_, IN_, OUT_, TB_)                                                                  \
    ql_sweep<G, F, L_, NT, D, IN_, OUT_, TB_, false>(tt, lds, reinterpret_cast<float2 const *>(lds + (kL2TabIN - TB_ / 2u) / 4u), \
                                              wordsT, g.rowbase, g.L, g.Lwave, g.has, sc, plane, tid, g.xt, dirty, o, lk)
                if (stage == 0u)
                {
                    if (first && last) QL2_SWEEP(true, true, IO_HBM, IO_HBM, 0u);
                    else if (first) QL2_SWEEP(true, false, IO_HBM, IO_LDS, 0u);
                    else if (last) QL2_SWEEP(false, true, IO_HBM, IO_HBM, 0u);
                    else QL2_SWEEP(false, false, IO_HBM, IO_LDS, 0u);
                }
                else
                {
                    if (last) QL2_SWEEP(false, true, IO_LDS, IO_HBM, kL2Tab1);
                    else QL2_SWEEP(false, false, IO_LDS, IO_HBM, kL2Tab1);
                }
#undef QL2_SWEEP
                // What outlives the sweep is parked in the group's spare plane row (park_off): the null score by the
                // stage that swept the first tile, alt score and feedback flag by the one that swept the last; the
                // epilogue below reads them after the block barrier (the two may be different wavefronts, and with
                // T = 2 they run in the same step).
                unsigned const po = park_off(g, (unsigned)NT, tid);
                if (first) st_off(sc, po, o.Rn);
                if (last)
                {
                    st_off(sc + plane, po, fmaxf(o.E + g.xt.ET, o.C + g.xt.CT));
                    st_off(sc + 2 * plane, po, dirty ? 1.0f : 0.0f);
                }
            }
        }
        __syncthreads(); // the parked values are visible to the final stage; nobody is still reading tabIN

        if (stage != final_stage) continue;
        for (unsigned gi = g0; gi < g1; ++gi)
        {
            cgroup *gd = (cgroup *)(unsigned long long)(a.groups + gi);
            GroupLane ge;
            ge.rowbase = gd->rowbase, ge.Lwave = gd->lmax;
            if (ge.Lwave == 0u || (tid & 63u) >= gd->nq) continue;
            unsigned const q = a.qorder[gd->qfirst + (tid & 63u)];
            unsigned const po = park_off(ge, (unsigned)NT, tid);
            float const nul = ld_off(sc, po), alt = ld_off(sc + plane, po);
            bool const dirty = ld_off(sc + 2 * plane, po) != 0.0f;
            ql_publish(a, pm, q, nul, alt, dirty || (kEM && pm.needs_exact_e));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Small batches (<= 64 queries): three INDEPENDENT wavefronts per block, each with its own 54.5 KB slice
// of the CU's LDS (tile image + insert/background table) and its own (profile, 64-query block) task
// stream.  With so few queries the 256-query kernel has one busy wavefront per block and LDS lets two
// blocks share a CU: two busy wavefronts per CU.  Three 64-thread blocks do not fit (allocation
// granularity), ONE 160 KiB block with three slices does: three busy wavefronts per CU.  No barrier is
// needed anywhere -- a wavefront's LDS operations execute in order and nobody else touches its slice.
// The slices' bases are not compile-time constants of the row code, so each gather address pays one add
// (+10 VALU per row, 4 %).  Same rows (ql_row), same results.
// ---------------------------------------------------------------------------------------------------
template <int G, int D, bool ROWS = false>
__global__ __launch_bounds__(192, 1) void viterbi_qlane_w3_kernel(dcp_qlane_args a)
{
    constexpr int NT = 64;
    constexpr int KT = 4 * G;
    constexpr int TAB_FLOATS = G * NC * 4;
    constexpr int SLICE = TAB_FLOATS + 2 * NC;
    __shared__ __attribute__((aligned(16))) float lds[3 * SLICE];
    unsigned const wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned const tid = threadIdx.x & 63u;
    float *const tabM = lds + wv * SLICE;
    float2 *const tabIN = reinterpret_cast<float2 *>(tabM + TAB_FLOATS);
    size_t const plane = (size_t)a.plane_rows * (unsigned)NT;
    float *const sc = a.scratch + ((size_t)blockIdx.x * 3u + wv) * kPlanes * plane;

    for (;;)
    {
        unsigned t0 = 0;
        if (tid == 0) t0 = atomicAdd(a.task_counter, 1u);
        unsigned const task = __builtin_amdgcn_readfirstlane(t0); // lane 0's value
        if (task >= a.ntasks) break;
        unsigned const slot = a.nprof - 1u - task / a.nqblocks; // biggest profiles first
        unsigned const qb = task % a.nqblocks;                  // = the slot: one wavefront slot per 64-lane block
        dcp_ql_prof const pm = a.profs[slot];
        unsigned const T = pm.ntiles;
        cu32 *sf = (cu32 *)(unsigned long long)a.slot_first;
        unsigned const g0 = sf[qb], g1 = sf[qb + 1u];
        uint32_t const *__restrict__ wordsT = a.words_t + __builtin_amdgcn_readfirstlane(a.wt_off[qb]);
        {
            float const *__restrict__ gi = a.emis_insert + (size_t)pm.pidx * NC;
            float const *__restrict__ gn = a.emis_null + (size_t)pm.pidx * NC;
            for (unsigned i = tid; i < (unsigned)NC; i += (unsigned)NT)
                tabIN[i] = float2{gi[i], gn[i]};
        }
        GroupLane g;
        bool const single = g1 - g0 == 1u;
        bool t_loaded = false;
        for (unsigned t = 0; t < T; ++t)
        {
            stage_tile_image<G, ROWS>(tabM, a, pm, t, tid, (unsigned)NT);
            compiler_fence(); // the sweep's gathers stay behind the image's stores (same wavefront: LDS keeps the order)
            cfloat *tt = as_const(a.ttrans + pm.ttrans_off + (size_t)t * (KT + 1) * 8);
            bool const first = t == 0, last = t + 1 == T;
            for (unsigned gi = g0; gi < g1; ++gi)
            {
                // a slot with ONE group (every uniform batch) keeps its lane state in registers over the task's tiles
                if (!(single && t_loaded)) load_group(a, gi, tid, g);
                t_loaded = true;
                if (g.Lwave == 0u) continue;
                SweepOut o{ninf(), ninf(), ninf()};
                bool dirty = false;
#define QLW_SWEEP(F, L_)                                                                                  \
    ql_sweep<G, F, L_, NT, D, IO_HBM, IO_HBM, 0u, false>(tt, tabM, tabIN, wordsT, g.rowbase, g.L, g.Lwave, g.has, sc, plane, tid, g.xt, dirty, o, LdsLink{})
                if (first && last) QLW_SWEEP(true, true);
                else if (first) QLW_SWEEP(true, false);
                else if (last) QLW_SWEEP(false, true);
                else QLW_SWEEP(false, false);
#undef QLW_SWEEP
                unsigned const po = park_off(g, (unsigned)NT, tid);
                if (first && !last) st_off(sc, po, o.Rn);
                if (last && g.has)
                {
                    float const nul = first ? o.Rn : ld_off(sc, po);
                    ql_publish(a, pm, g.q, nul, fmaxf(o.E + g.xt.ET, o.C + g.xt.CT), dirty || (kEM && pm.needs_exact_e));
                }
            }
            compiler_fence(); // ... and the next image's stores behind this sweep's gathers
        }
    }
}

// The block's window plane: uint16 [rows][NT] starting at word wt_off[qb] of words_t.  Lane t of wavefront slot s
// holds, for every group of the slot, rows rowbase + 0 .. rowbase + lmax + 9 of column t: entry (rowbase + r, t) =
// (window of row r of the group's query in lane t) << 4, the window being the base-4 value of the last five bases up
// to position r (zeros before the start and past the end).  Rows no group owns are zeroed: every entry is a valid
// gather offset whatever a prefetch reads.
template <int NT>
__global__ __launch_bounds__(NT) void transpose_words_kernel(dcp_qlane_args a)
{
    unsigned const qb = blockIdx.x, tid = threadIdx.x;
    unsigned const rows = (a.wt_off[qb + 1] - a.wt_off[qb]) * 2u / (unsigned)NT;
    uint16_t *dst = reinterpret_cast<uint16_t *>(a.words_t + a.wt_off[qb]);
    unsigned const sidx = qb * ((unsigned)NT / 64u) + (tid >> 6);
    unsigned const g0 = a.slot_first[sidx], g1 = a.slot_first[sidx + 1u];
    unsigned r = 0; // next plane row of this lane's column to write
    for (unsigned gi = g0; gi < g1; ++gi)
    {
        dcp_ql_group const g = a.groups[gi];
        bool const has = (tid & 63u) < g.nq;
        unsigned const q = has ? a.qorder[g.qfirst + (tid & 63u)] : 0u;
        uint32_t const *__restrict__ src = a.seq_words + a.seq_woff[q];
        unsigned const len = has ? a.seq_len[q] : 0u;
        for (; r < g.rowbase; ++r)
            dst[r * (unsigned)NT + tid] = 0;
        unsigned w = 0, word = 0;
        unsigned const gr = g.lmax + 10u;
        for (unsigned k = 0; k < gr && r < rows; ++k, ++r)
        {
            if (k >= 1u)
            {
                unsigned const pos = k - 1u; // row k ends with base k - 1
                if ((pos & 15u) == 0u) word = pos < len ? src[pos >> 4] : 0u;
                w = ((w << 2) | ((pos < len ? word >> ((pos & 15u) * 2u) : 0u) & 3u)) & 1023u;
            }
            dst[r * (unsigned)NT + tid] = (uint16_t)(w << 4);
        }
    }
    for (; r < rows; ++r)
        dst[r * (unsigned)NT + tid] = 0;
}

// (dcp_qlane_args::tiles_from_rows picks the ROWS instantiation: the default kernels' code is the same with and
// without it in the library)
template <int G, int NT, int D>
static void launch_ql(dcp_qlane_args const *a, unsigned nblocks, hipStream_t s)
{
    if (a->tiles_from_rows) hipLaunchKernelGGL((viterbi_qlane_kernel<G, NT, D, true>), dim3(nblocks), dim3(NT), 0, s, *a);
    else hipLaunchKernelGGL((viterbi_qlane_kernel<G, NT, D>), dim3(nblocks), dim3(NT), 0, s, *a);
}

// One configuration is built: KT = 8 nodes per tile (G = 2: the tile's transitions fit in
// SGPRs), 256 queries per block at 2 wavefronts per SIMD, boundary prefetch 3 rows deep.
// Measured alternatives: KT = 12 spills at the 256-VGPR cap; 384-thread
// blocks at 3 wavefronts per SIMD (168 VGPRs) spill ~840 registers and run 3x slower.
#ifndef DCP_QLANE_NT
#define DCP_QLANE_NT 256
#endif
#ifndef DCP_QLANE_D
#define DCP_QLANE_D 3 // rows of boundary prefetch (1..5); measured 1: -30 %, 2: -1 %, 3: best, 4: -0.3 %, 5: -0.7 %
#endif
extern "C" unsigned dcp_qlane_block_size(void) { return DCP_QLANE_NT; }
extern "C" unsigned dcp_qlane_tile_nodes(void) { return 8u; }
extern "C" unsigned dcp_qlane_scratch_planes(void) { return kPlanes; }
extern "C" unsigned dcp_qlane_diag_build(void) { return DCP_QLANE_DIAG; }
// plane rows of a group's region for a longest member of `lmax` bases: rows 0 .. lmax + 9, an even number (the window
// plane holds two uint16 rows per 32-bit word and lane)
extern "C" unsigned dcp_qlane_group_rows(unsigned lmax) { return (lmax + 11u) & ~1u; }
extern "C" unsigned dcp_qlane_window_planes(void) { return kWPlane ? 1u : 0u; }
extern "C" unsigned dcp_qlane_exact_e_by_redo(void) { return kEM ? 1u : 0u; }

extern "C" int dcp_launch_qlane_transpose(dcp_qlane_args const *a, unsigned nt, void *stream)
{
    if (nt == 64) hipLaunchKernelGGL((transpose_words_kernel<64>), dim3(a->nqblocks), dim3(64), 0, (hipStream_t)stream, *a);
    else if (nt == DCP_QLANE_NT)
        hipLaunchKernelGGL((transpose_words_kernel<DCP_QLANE_NT>), dim3(a->nqblocks), dim3(DCP_QLANE_NT), 0, (hipStream_t)stream, *a);
    else return 1;
    return 0;
}

extern "C" unsigned dcp_qlane2_lds_bytes(void) { return kL2Bytes; }

// three independent 64-query wavefronts per block (small batches): nblocks blocks of 192 threads
extern "C" int dcp_launch_qlane_w3(dcp_qlane_args const *a, unsigned nblocks, void *stream)
{
    if (a->tiles_from_rows)
        hipLaunchKernelGGL((viterbi_qlane_w3_kernel<2, DCP_QLANE_D, true>), dim3(nblocks), dim3(192), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((viterbi_qlane_w3_kernel<2, DCP_QLANE_D>), dim3(nblocks), dim3(192), 0, (hipStream_t)stream, *a);
    return 0;
}

extern "C" int dcp_launch_qlane2(dcp_qlane_args const *a, unsigned nblocks, void *stream)
{
    if (a->tiles_from_rows)
        hipLaunchKernelGGL((viterbi_qlane2_kernel<2, DCP_QLANE_D, true>), dim3(nblocks), dim3(512), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((viterbi_qlane2_kernel<2, DCP_QLANE_D>), dim3(nblocks), dim3(512), 0, (hipStream_t)stream, *a);
    return 0;
}

extern "C" int dcp_launch_qlane(dcp_qlane_args const *a, unsigned nblocks, unsigned nt, void *stream)
{
    if (nt == DCP_QLANE_NT) launch_ql<2, DCP_QLANE_NT, DCP_QLANE_D>(a, nblocks, (hipStream_t)stream);
    else return 1;
    return 0;
}
